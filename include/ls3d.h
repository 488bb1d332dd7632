/*
 * include/ls3d.h — C ABI of libls3d.so: the MI355X (gfx950) kernels of the MSeg3D / SDSeg3D
 * segmentation forward path of jialeli1/lidarseg3d.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in _host;
 *   - the caller owns every buffer (incl. workspaces, sized by the *_workspace_bytes queries);
 *   - every call is stream-ordered on `stream` (a hipStream_t passed as void*), never synchronises
 *     and never allocates;
 *   - return value: LS3D_OK (0) or a negative LS3D_ERR_* code; no exit(), no exceptions
 *     (the reference's pointnet2 launcher calls exit(-1) on a CUDA error,
 *     det3d/ops/pointnet2_batch/src/interpolate_gpu.cu:76-80 — deliberately not reproduced);
 *   - row counts come as a host capacity `n` plus an optional device count `n_dev` (int32*): when
 *     n_dev != NULL the kernels process min(*n_dev, n) rows, so producer -> consumer chains need no
 *     host round trip.
 *
 * Each entry point cites the reference interface (file:line in the jialeli1/lidarseg3d tree) it replaces.
 */
#ifndef LS3D_H
#define LS3D_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LS3D_OK 0
#define LS3D_ERR_ARG (-1)
#define LS3D_ERR_LAUNCH (-2)
#define LS3D_ERR_UNSUPPORTED (-3)
#define LS3D_ERR_WORKSPACE (-4)

/* arithmetic of ls3d_gather_gemm.  F32: exact f32 products and accumulation (v_mfma_f32_32x32x2_f32).
 * BF16X3: f32 operands split into bf16 head + tail, a*b = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
 * v_mfma_f32_32x32x16_bf16 with f32 accumulation: ~1e-5 relative error per layer, 5.3x less matrix-pipe time.
 * BF16X6: operands split EXACTLY into three bf16 planes (8+8+8 mantissa bits), the six partial products of weight
 * >= 2^-16 are accumulated in f32; the dropped ones are <= 2^-24 relative, the size of an f32 rounding error, so results
 * agree with F32 to f32 rounding (measured against float64: tests/probes/bwd_dbg.py), at 2.7x less matrix-pipe time. */
#define LS3D_PRECISION_F32 0
#define LS3D_PRECISION_BF16X3 1
#define LS3D_PRECISION_BF16X6 2
/* (the 8-product arithmetic of the same split, "bf16x8", exists on the tile-halo path only: ls3d_tile_conv, products = 8) */

typedef void *ls3d_stream_t;

/* library / build identification: returns e.g. "ls3d 0.1 gfx950" */
const char *ls3d_version(void);

/* ------------------------------------------------------------------------------------------------
 * Voxelization
 * ---------------------------------------------------------------------------------------------- */

/* Geometry of a voxel grid.  grid = round((hi-lo)/vs) per axis, computed by the caller in f32 exactly as
 * det3d/ops/point_cloud/point_cloud_ops.py:26-29 / det3d/ops/voxel/src/voxelization_cpu.cpp:118-121. */
typedef struct {
  float vs[3];    /* voxel size x,y,z */
  float lo[3];    /* range minimum x,y,z */
  int32_t grid[3];/* cells x,y,z */
} ls3d_grid_t;

/* Layout of a point table: row i starts at points + i*stride (floats); xyz at columns xyz_col..+2;
 * batch index (as float, collate_kitti's column 0, det3d/torchie/parallel/collate.py:141-150) at
 * batch_col or -1 for a single frame; the n_feat columns copied into `voxels` start at feat_col. */
typedef struct {
  int32_t stride, xyz_col, batch_col, feat_col, n_feat;
} ls3d_points_layout_t;

/* dynamic voxelization: coors[n,3] = (z,y,x) int32, (-1,-1,-1) for points outside the range.
 * Replaces voxel_layer.dynamic_voxelize (det3d/ops/voxel/src/voxelization.h:77-88,
 * CUDA kernel det3d/ops/voxel/src/voxelization_cuda.cu:25-61).  f32 subtraction, f32 division, floor. */
int ls3d_voxelize_dynamic(const float *points, int n, const ls3d_points_layout_t *lay_host,
                          const ls3d_grid_t *grid_host, int32_t *coors, ls3d_stream_t stream);

size_t ls3d_voxelize_hard_workspace_bytes(int n, int max_points, int max_voxels);

/* hard voxelization, bit-exact with the reference's serial algorithms:
 *   overflow_mode 0 = numba kernel, det3d/ops/point_cloud/point_cloud_ops.py:7-55 (what the MSeg3D/SDSeg3D
 *                     configs run in the dataloader; `continue` once max_voxels is reached);
 *   overflow_mode 1 = voxel_layer.hard_voxelize, det3d/ops/voxel/src/voxelization.h:46-61,
 *                     voxelization_cpu.cpp:42-99 (`break`).
 * Voxel ids are first-appearance order over the point order, the first max_points points are kept.
 * Outputs: voxels[max_voxels,max_points,n_feat] (zero padded), coors[max_voxels,coors_cols] with
 * coors_cols==3 -> (z,y,x), ==4 -> (batch,z,y,x); num_points[max_voxels]; *num_voxels_dev.
 * With batch_col >= 0 the frames must be concatenated in batch order; max_voxels then bounds the TOTAL
 * (callers needing a per-frame cap below the frame's point count voxelise frame by frame). */
int ls3d_voxelize_hard(const float *points, int n, const ls3d_points_layout_t *lay_host,
                       const ls3d_grid_t *grid_host, int max_points, int max_voxels, int overflow_mode,
                       void *workspace, size_t workspace_bytes, float *voxels, int32_t *coors, int coors_cols,
                       int32_t *num_points, int32_t *num_voxels_dev, ls3d_stream_t stream);

size_t ls3d_dynamic_scatter_workspace_bytes(int n);

/* DynamicScatter (det3d/ops/voxel/scatter_points.py:68-129 over dynamic_point_to_voxel_forward,
 * det3d/ops/voxel/src/voxelization.h:90-100): group points of equal coordinate (coors[n,coors_cols],
 * rows with -1 dropped), first-appearance voxel order, reduce ALL points of a voxel:
 * mode 0 = mean, 1 = max, 2 = sum (point order; DynamicScatterWithDistance, scatter_points.py:132-213).  Outputs feats[n,n_feat] (first *num_voxels_dev rows valid), voxel_coors[n,coors_cols],
 * point2voxel[n] (may be NULL). */
int ls3d_dynamic_scatter(const float *feats_in, int n, int n_feat, const int32_t *coors, int coors_cols,
                         const int32_t shape_zyx_host[3], int mode, void *workspace, size_t workspace_bytes,
                         float *feats_out, int32_t *voxel_coors, int32_t *point2voxel,
                         int32_t *num_voxels_dev, ls3d_stream_t stream);

size_t ls3d_dynamic_scatter_backward_workspace_bytes(int n, int n_feat);

/* Backward of ls3d_dynamic_scatter: grad_points[n,n_feat] from grad_voxels[V,n_feat] and the forward's point2voxel[n].
 * Replaces voxel_layer.dynamic_point_to_voxel_backward (det3d/ops/voxel/src/voxelization.h:95-110,
 * scatter_points_cuda.cu:247-282) TOGETHER WITH torch's backward of the mean / max over the padded [V,M,C] tensor
 * (scatter_points.py:34-50,89-98): mode 0: grad/count to every point of the voxel; mode 1: grad to the first point
 * (lowest index) whose value equals the reduced value, nothing if only the zero padding attains it (needs the forward's
 * feats_in[n,n_feat] and feats_out[V,n_feat]; both may be NULL for mode 0).  Points outside (point2voxel < 0) get zeros. */
int ls3d_dynamic_scatter_backward(const float *grad_voxels, const int32_t *point2voxel, int n, int n_feat, int mode,
                                  const float *feats_in, const float *feats_out, void *workspace, size_t workspace_bytes,
                                  float *grad_points, ls3d_stream_t stream);

/* voxel_layer.dynamic_point_to_voxel_forward / _backward with the reference's own argument shapes (det3d/ops/voxel/src/voxelization.h:63-111,
 * bound at voxelization.cpp:6-11; CUDA: scatter_points_cuda.cu:142-282), for a maintainer who replaces `voxel_layer` function by function
 * (the module path above - ls3d_dynamic_scatter - never materialises the padded tensor):
 *   _index   : voxel_mapping[n, ndim] (ndim 3 = (z,y,x) or 4 = (batch,z,y,x); rows with a negative coordinate are outside) ->
 *              point_to_voxelidx[n] (slot of the point inside its voxel, point order; -1 outside), coor_to_voxelidx[n] (voxel of the point,
 *              first-appearance order; -1 outside), num_points_per_voxel[n] (first voxel_num entries), voxel_coors[n, ndim],
 *              counts_dev[2] = (voxel_num, max_points = the fullest voxel's count).  The reference copies the same two counts to the host
 *              between its kernels (scatter_points_cuda.cu:218-221) to size `voxels`; here the caller does, then calls
 *   _forward : voxels[voxel_num, max_points, n_feat] = zero padded scatter of points[n, n_feat] (scatter_point_to_voxel_kernel, :20-48);
 *   _backward: grad_input_points[i] = grad_output_voxels[coor_to_voxelidx[i], point_to_voxelidx[i]] for the points inside, other rows
 *              untouched (map_voxel_to_point_kernel, :50-69; the caller passes zeros, scatter_points.py:57-63). */
size_t ls3d_dynamic_point_to_voxel_workspace_bytes(int n);
int ls3d_dynamic_point_to_voxel_index(const int32_t *voxel_mapping, int n, int ndim, const int32_t shape_zyx_host[3], void *workspace,
                                      size_t workspace_bytes, int32_t *point_to_voxelidx, int32_t *coor_to_voxelidx,
                                      int32_t *num_points_per_voxel, int32_t *voxel_coors, int32_t *counts_dev, ls3d_stream_t stream);
int ls3d_dynamic_point_to_voxel_forward(const float *points, int n, int n_feat, const int32_t *point_to_voxelidx, const int32_t *coor_to_voxelidx,
                                        int voxel_num, int max_points, float *voxels, ls3d_stream_t stream);
int ls3d_dynamic_point_to_voxel_backward(float *grad_input_points, const float *grad_output_voxels, const int32_t *point_to_voxelidx,
                                         const int32_t *coor_to_voxelidx, int n, int n_feat, int max_points, ls3d_stream_t stream);

size_t ls3d_segment_reduce_workspace_bytes(int n, int n_seg);

/* Segment mean / max over dim 0: out[n_seg,n_feat] from src[n,n_feat] and one int64 segment id per row (mode 0 = mean,
 * 1 = max with arg_out[n_seg,n_feat] = lowest row index attaining it, may be NULL).  Segments without rows give 0 (arg = n).
 * Replaces torch_scatter.scatter_mean / scatter_max as the dynamic readers call them with torch.unique's inverse
 * (det3d/models/readers/voxel_encoder.py:366-372,451-456,594-600,682-686; torch_scatter itself is a third-party
 * dependency absent from the reference tree).  Ids outside [0, n_seg) are skipped. */
int ls3d_segment_reduce(const float *src, const int64_t *index, int n, int n_feat, int n_seg, int mode, void *workspace,
                        size_t workspace_bytes, float *out, int64_t *arg_out, ls3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * The dynamic (point-wise) readers and the Cylinder3D blocks: SURVEY.md 8f rank 4
 * ---------------------------------------------------------------------------------------------- */

/* cart2cylind + voxelize of PolarNetDynamicVoxelFeatureExtractor / Cylinder3DDynamicVoxelFeatureExtractor
 * (det3d/models/readers/voxel_encoder.py:11-18,333-360,563-590): points[n, stride] = (batch, x, y, z, features...);
 *   cyl5[n, 5]   = (rho = sqrt(x*x + y*y), phi = atan2(y, x), z, x, y) - the five columns the readers centre on their voxel mean;
 *   vcoors[n, 4] = (batch, cell columns) int64, cell_j = clamp(int(floor((cyl_j - lo_j) / vs_j)), 0, grid_j - 1) (f32 subtraction and
 *                  division; the reference clamps BEFORE its range test, so every point is inside), columns (c0, c1, c2) or - reverse != 0,
 *                  Cylinder3D - (c2, c1, c0);
 *   keys[n]      = the row linearised in that column order (collapse_last != 0, PolarNet: last column = grid[2] / 2 for every point),
 *                  so that sorting the keys is torch.unique(dim=0)'s lexicographic row order.  LS3D_ERR_UNSUPPORTED when
 *                  batch * grid cells >= 2^32.
 * phi is atan2 evaluated in double and rounded once; torch's f32 atan2 may differ from it by an ulp, which moves a point that sits within
 * ~1e-5 of a cell boundary into the neighbouring cell (the reference's CPU and CUDA builds differ from each other in the same way). */
int ls3d_cyl_voxelize(const float *points, int n, int stride, const ls3d_grid_t *grid_host, int reverse, int collapse_last, int batch, float *cyl5,
                      int64_t *vcoors, uint32_t *keys, ls3d_stream_t stream);

/* torch.unique(rows, return_inverse=True, return_counts=True, dim=0) (voxel_encoder.py:441,670) from the SORTED keys of ls3d_cyl_voxelize
 * (ls3d_radix_sort: keys_sorted, perm = source positions): inverse[n] (int64, indexed by the unsorted position), unique_rows[n, 4]
 * (int64, the key decoded with dims_host = the sizes of its last three columns), counts[n] (int64), *n_unique_dev; the first
 * *n_unique_dev rows of unique_rows / counts are valid.  workspace: ls3d_unique_sorted_workspace_bytes(n). */
size_t ls3d_unique_sorted_workspace_bytes(int n);
int ls3d_unique_sorted(const uint32_t *keys_sorted, const int32_t *perm, int n, const int32_t dims_host[3], void *workspace, size_t workspace_bytes,
                       int64_t *inverse, int64_t *unique_rows, int64_t *counts, int32_t *n_unique_dev, ls3d_stream_t stream);

/* prepare_input_feature (voxel_encoder.py:362-386,592-616) for every point, then x * scale + shift (PPmodel's leading BatchNorm1d in
 * eval mode; NULL = none): out[n, out_ld] = [cyl5 (5), points[:, 4:] (stride - 4), cyl5 - mean5[inverse] (5), cyl - voxel centre (3)],
 * zero padded to out_ld >= stride + 9 columns; mean5[V, 5] = ls3d_segment_reduce(cyl5, inverse, mean); the voxel centre is
 * common_utils.get_voxel_centers on the three stored cell columns of vcoors, whatever their order (core/utils/common_utils.py:74-90). */
int ls3d_dyn_point_features(const float *points, int n, int stride, const float *cyl5, const int64_t *vcoors, const int64_t *inverse,
                            const float *mean5, const ls3d_grid_t *grid_host, const float *scale, const float *shift, float *out, int out_ld,
                            ls3d_stream_t stream);

/* Test-time-augmentation merge of PointSegBatchlossHead.predict / PointSegMSeg3DHead.predict (det3d/models/point_heads/
 * point_seg_batchloss_head.py:190-245, point_seg_mseg3d_head.py:398-453, merge_type "ArithmeticMean"): the k augmented variants of one sample
 * are k frames of n points each inside the collated batch, variant t starting at row variant_first_row_host[t] of logits[*, ld];
 * probs_out[n, num_class] (may be NULL) = mean over t of softmax(logits[first_t + p]), labels_out[n] (int64) = its argmax (first index on
 * ties).  One pass, no boolean masks, no [k, n, C] intermediate.  k <= 16; num_class <= 64 (LS3D_ERR_UNSUPPORTED beyond). */
int ls3d_tta_merge(const float *logits, int ld, int num_class, int n, const int32_t *variant_first_row_host, int k, float *probs_out,
                   int64_t *labels_out, ls3d_stream_t stream);

/* y[r, c] = post(pre(x[r, c]) * scale[c] + shift[c]) [+ add[r, c]] [* mul[r, c]] over n rows (n_dev: device count, optional) of c columns:
 * the tails of the Cylinder3D blocks - convolution, LeakyReLU, BatchNorm1d in THAT order, the residual sums, ReconBlock's sigmoid gates
 * times its input (det3d/models/backbones/scn_unet_cylinder3d.py:52-252, cylinder3d_backbone.py:50-252) - which the BN -> ReLU epilogue
 * of the gather-GEMM does not cover.  pre_act / post_act: 0 none, 1 ReLU, 2 LeakyReLU(slope), 3 sigmoid; scale / shift / add / mul
 * may be NULL; y may alias x.  float4 accesses when c and every row stride are multiples of 4 and the pointers 16-byte aligned. */
int ls3d_act_affine(const float *x, int x_ld, int n, const int32_t *n_dev, int c, int pre_act, int post_act, float slope, const float *scale,
                    const float *shift, const float *add, int add_ld, const float *mul, int mul_ld, float *y, int y_ld, ls3d_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Voxel feature extractors (readers)
 * ---------------------------------------------------------------------------------------------- */

/* MeanVoxelFeatureExtractor.forward, det3d/models/readers/voxel_encoder.py:51-58. out[v,c] ld = out_ld */
int ls3d_vfe_mean(const float *voxels, const int32_t *num_points, int n, const int32_t *n_dev, int max_points,
                  int n_feat, float *out, int out_ld, ls3d_stream_t stream);

/* ImprovedMeanVoxelFeatureExtractor.forward, voxel_encoder.py:74-124:
 * [mean_xyz, max_xyz, min_xyz, mean_other, density, std] -> out[v, 0..n_feat+8); columns up to out_ld are
 * zero filled (so the 13-channel descriptor can feed a 16-wide GEMM). */
int ls3d_vfe_improved_mean(const float *voxels, const int32_t *num_points, int n, const int32_t *n_dev,
                           int max_points, int n_feat, float *out, int out_ld, ls3d_stream_t stream);

/* TransformerVoxelFeatureExtractor token assembly, voxel_encoder.py:210-252: row (v*max_points+t) of
 * tokens = [point features (n_feat), descriptor (n_feat+8), zeros up to tok_ld]. */
int ls3d_vfe_tokens(const float *voxels, const int32_t *num_points, int n, const int32_t *n_dev, int max_points,
                    int n_feat, float *tokens, int tok_ld, ls3d_stream_t stream);

/* multi-head self attention core over groups of `seq` consecutive rows (nn.MultiheadAttention, eval, no masks;
 * used with seq = 5 points of a voxel, voxel_encoder.py:155, and seq = 2*num_class memory tokens,
 * det3d/models/point_heads/context_module.py:231).  qkv[rows, 3*embed] = (q | k | v); out[rows, embed]. */
int ls3d_mha_core(const float *qkv, int groups, const int32_t *groups_dev, int seq, int embed, int heads,
                  float *out, ls3d_stream_t stream);

/* max over the `seq` rows of every group: in[groups*seq, c] -> out[groups, c] (voxel_encoder.py:265) */
int ls3d_group_max(const float *in, int groups, const int32_t *groups_dev, int seq, int c, float *out,
                   ls3d_stream_t stream);

/* TransformerVoxelFeatureExtractor.forward (voxel_encoder.py:202-270) as ONE kernel: tokens -> embedding -> num_layers x
 * TransformerEncoderLayerPreNorm (:149-163) -> max over the point slots -> Linear + ReLU compression; out[n, num_compressed]
 * (or [n, embed] without compression).  Matrix weights in the layout of ls3d_gather_gemm_pack(nt = 2, F32) of the (in, out)
 * matrices (embedding K padded to token_ld), w_compress: the PLAIN nn.Linear weight [num_compressed][embed].
 * Supported: embed 64, heads 4, ffn 128, token_ld 32 (2C + 8 <= 32), P <= 32, num_layers <= 4; anything else returns
 * LS3D_ERR_UNSUPPORTED and the caller composes the layer from ls3d_vfe_tokens / ls3d_gather_gemm / ls3d_mha_core / ls3d_group_max. */
typedef struct {
  const float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2, *n1_gamma, *n1_beta, *n2_gamma, *n2_beta;
  float n1_eps, n2_eps;
} ls3d_transvfe_layer_t;
typedef struct {
  const float *w_embed, *b_embed, *w_compress, *b_compress;
  const ls3d_transvfe_layer_t *layers; /* host array [num_layers] */
  int num_layers, num_compressed, embed, heads, ffn, token_ld;
  int planes; /* 0: f32 MFMA, the five GEMM weights (w_embed, wqkv, wo, w1, w2) in the packed nt = 2 layout of ls3d_gather_gemm_pack;
               * 6 | 8: the exact 3-plane bf16 split (f32-grade, see ls3d_tile_conv), the five weights converted by ls3d_transvfe_pack_planes */
  int flags;  /* bit 0 (planes == 0 only): experimental variant with the B fragments straight from the L2-resident weights instead of
               * LDS staging, no workgroup barriers - same results, measured slower (profiles/round2_experiments.md);
               * bit 1: no token deduplication even when ls3d_transvfe gets a workspace (A/B) */
} ls3d_transvfe_t;
size_t ls3d_transvfe_planes_bytes(int K, int N);
int ls3d_transvfe_pack_planes(const float *w_packed_nt2 /*K x N*/, int K, int N, void *out, ls3d_stream_t stream);
/* workspace (optional, ls3d_transvfe_workspace_bytes(n, P) bytes, 16-byte aligned): enables the token deduplication - the zero-padded
 * slots of a voxel are identical tokens in the reference's unmasked transformer (voxel_encoder.py:154-161), so a voxel with k < P
 * points is computed on k + 1 rows, the padding row's key weighted P - k in the softmax; voxels are grouped by k with one stable
 * radix pass.  Same result up to the summation order inside the softmax; 2.1x fewer 32-row tiles on LiDAR data.  Voxels whose slots
 * beyond num_points are not all zero keep all P rows. */
size_t ls3d_transvfe_workspace_bytes(int n, int P);
int ls3d_transvfe(const float *voxels /*[n,P,C]*/, const int32_t *num_points, int n, const int32_t *n_dev, int P, int C,
                  const ls3d_transvfe_t *model, float *out, int out_ld, void *workspace, size_t workspace_bytes, ls3d_stream_t stream);

/* y = LayerNorm(x (+ res)) * gamma + beta over the last dim c (<= 256); ld = c for all */
int ls3d_layernorm(const float *x, const float *res, const float *gamma, const float *beta, float eps, int rows,
                   const int32_t *rows_dev, int c, float *y, ls3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sparse convolution: coordinate index + rulebooks + gather-GEMM
 * (third-party spconv v1.x @ fad3000, call sites det3d/models/backbones/scn_unet.py:15-24,205;
 *  semantics SURVEY.md §2.3)
 * ---------------------------------------------------------------------------------------------- */

/* open-addressing hash (64-bit linearised (b,z,y,x) keys -> row).  cap must be a power of two >= 2*n.
 * keys[cap] (uint64), vals[cap] (int32). */
int ls3d_index_build(const int32_t *coords /*[n,4]*/, int n, const int32_t *n_dev, const int32_t shape_zyx_host[3],
                     uint64_t *keys, int32_t *vals, int cap, ls3d_stream_t stream);

/* SubMConv3d rulebook: nbr[n, kvol] with kvol = kz*ky*kx, nbr[v,k] = input row at coords[v] + k - ksize/2
 * (kernel offset index row-major over (kz,ky,kx)) or -1. */
int ls3d_rulebook_subm(const int32_t *coords, int n, const int32_t *n_dev, const int32_t shape_zyx_host[3],
                       const int32_t ksize_host[3], const uint64_t *keys, const int32_t *vals, int cap,
                       int32_t *nbr, ls3d_stream_t stream);

size_t ls3d_rulebook_conv_workspace_bytes(int batch, const int32_t out_shape_zyx_host[3]);

/* SparseConv3d rulebook (+ the transposed table SparseInverseConv3d needs).
 *   out_shape = (in + 2*pad - k)/stride + 1;  output sites in ascending linear index (spconv's CUDA order);
 *   nbr_out[out_cap, kvol]: nbr_out[o,k] = input row at o*stride - pad + k, or -1;
 *   nbr_inv[n_in, kvol]:    nbr_inv[i,k] = output row fed by input i through offset k, or -1;
 *   out_coords[out_cap,4]; *n_out_dev = number of output sites (rows beyond out_cap are dropped and
 *   *overflow_dev is set to 1). */
int ls3d_rulebook_conv(const int32_t *coords_in, int n_in, const int32_t *n_in_dev, int batch,
                       const int32_t in_shape_zyx_host[3], const int32_t ksize_host[3],
                       const int32_t stride_host[3], const int32_t pad_host[3], void *workspace,
                       size_t workspace_bytes, int32_t *out_coords, int out_cap, int32_t *n_out_dev,
                       int32_t *nbr_out, int32_t *nbr_inv, int32_t *overflow_dev, ls3d_stream_t stream);

/* mask[r] = bitmask of the kernel offsets with an active neighbour in row r of a rulebook table (kvol <= 31).
 * A permutation of the rows sorted by this key, passed as `row_order` to ls3d_gather_gemm, makes the rows that a
 * wave processes together share their empty offsets, so whole MFMA groups / weight chunks are skipped (the same
 * idea as spconv v2's mask-sorted implicit GEMM).  Results are written to the original rows: only the processing
 * order changes. */
int ls3d_rulebook_masks(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, int32_t *mask, ls3d_stream_t stream);

/* The same ordering for SEVERAL tables with one sort: keys[r] = (segment << 27) | mask(r) (descending != 0: mask
 * complemented, densest rows first), kvol <= 27, segment <= 15.  The caller concatenates the keys of its tables, sorts
 * them once (ascending, keeping the source positions) and converts the positions with ls3d_segment_local_index:
 * out[i] = perm[i] - seg_offsets[s] for i in [seg_offsets[s], seg_offsets[s+1]) (seg_offsets: nseg + 1 host ints). */
int ls3d_rulebook_sort_keys(const int32_t *tbl, int n, const int32_t *n_dev, int kvol, int segment, int descending, int32_t *keys,
                            ls3d_stream_t stream);
/* Sort keys for the rows of the TRANSPOSED table of a strided convolution (nbr_inv of ls3d_rulebook_conv: SparseInverseConv3d, dgrad of
 * SparseConv3d) from the input coordinates alone: a row's offset mask is a function of the residues (c + pad) mod stride, so
 * keys[r] = (segment << class_bits) | class(r), class < stride_z * stride_y * stride_x ordered densest first, and rows beyond *n_dev get
 * the spare class 2^class_bits - 1.  One radix pass over class_bits + segment bits replaces the four passes of the 27-bit mask keys; use
 * with ls3d_radix_sort + ls3d_segment_local_index32 exactly like ls3d_rulebook_sort_keys.  stride > 8 or more classes than
 * 2^class_bits - 1: LS3D_ERR_UNSUPPORTED (sort by ls3d_rulebook_sort_keys instead). */
int ls3d_rulebook_parity_keys(const int32_t *coords_in, int n, const int32_t *n_dev, const int32_t ksize_host[3], const int32_t stride_host[3],
                              const int32_t pad_host[3], int segment, int class_bits, int32_t *keys, ls3d_stream_t stream);
int ls3d_segment_local_index(const int64_t *perm, int n, const int32_t *seg_offsets_host, int nseg, int32_t *out, ls3d_stream_t stream);
int ls3d_segment_local_index32(const int32_t *perm, int n, const int32_t *seg_offsets_host, int nseg, int32_t *out, ls3d_stream_t stream);

/* (The library has no process-global switches: every A/B knob is an argument of the call or a field of the model descriptor it
 * applies to, so two models or two threads in one process never see each other's settings.) */

/* Fused epilogue of the gather-GEMM (all optional):
 *   v = acc * scale[c] + shift[c]           (folded eval BatchNorm / bias)
 *   v += res_pre[r*res_pre_ld + c]          (SparseBasicBlock identity, scn_unet.py:66)
 *   v = relu ? max(v,0) : v
 *   v += pair[r*pair_ld + 2c] + pair[r*pair_ld + 2c+1]   (channel_reduction + add, scn_unet.py:168-169)
 *   v = LayerNorm_row(v) * ln_gamma[c] + ln_beta[c]      (post-/pre-norm transformer layers; needs nt*wc*32 >= cout,
 *                                                          i.e. the whole row in one workgroup; exclusive with `pair`) */
typedef struct {
  const float *scale, *shift;
  const float *res_pre;
  int32_t res_pre_ld;
  const float *pair;
  int32_t pair_ld;
  int32_t relu;
  const float *ln_gamma, *ln_beta;
  float ln_eps;
} ls3d_epilogue_t;

/* Weight packing for ls3d_gather_gemm.  Input: plain row-major W[kvol][cin_src][cout] (= spconv's
 * (kD,kH,kW,Cin,Cout) flattened, or a transposed nn.Linear weight with kvol = 1).  Output (w_packed, holding
 * ls3d_gather_gemm_packed_floats(kvol,cin_pad,cout) floats): zero padded to cin_pad (multiple of 16) rows and
 * roundup(cout,32) columns, laid out [kvol][slab][cin_pad][32][NT] so that a workgroup's weight chunk is one
 * contiguous copy into LDS and a lane's MFMA B operands for one k-step are one LDS read.
 * NT (1..4, must divide roundup(cout,32)/32) is the number of 32-column blocks one workgroup computes: large NT
 * = fewer, fatter workgroups (best MFMA/byte ratio), small NT = more workgroups (needed to fill 256 CUs when
 * the row count is small).  nt == 0 selects ls3d_gather_gemm_default_nt(cout).  The same nt must be passed
 * to ls3d_gather_gemm. */
size_t ls3d_gather_gemm_packed_floats(int kvol, int cin_pad, int cout);
int ls3d_gather_gemm_default_nt(int cout);
int ls3d_gather_gemm_pack(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, int nt, int precision,
                          float *w_packed, ls3d_stream_t stream);

/* out[r, 0..cout) = epilogue( sum_k W[k]^T * in[tbl[r,k]] ), tbl == NULL means the identity table with
 * kvol == 1 (a dense Linear layer).  in[*, cin] row stride in_ld, `w` = weights packed by
 * ls3d_gather_gemm_pack with cin_pad == cin, out row stride out_ld.  cin must be a multiple of 16, in_ld of 4.
 * f32 MFMA (v_mfma_f32_32x32x2_f32): exact f32 products and accumulation.
 * Workgroup geometry: 4 waves as a (4/wc) x wc grid, each wave 32 rows x 32*nt columns, so a workgroup covers
 * 32*(4/wc) rows x 32*nt*wc columns (nt*wc <= 4, must divide roundup(cout,32)/32; wc == 0 means 1).  The weights must
 * have been packed with the same nt.
 * row_order (optional, int32[n_rows]): tile slot i processes output row row_order[i] (see ls3d_rulebook_masks).
 * flags (per call; results are identical for every value): bits 0-1 = workgroup -> (tile, column slab) mapping, 0 (default): the
 * slabs of a tile run on one XCD, tiles interleaved over the 8 XCDs; bit 0: each XCD takes a contiguous range of tiles; bit 1:
 * slab-major dispatch (every slab re-gathers its rows from HBM); bit 2: sparse BF16X6 layers on the one-stage pipeline of round 3 instead of
 * the one that gathers two stages ahead (k_gather_gemm_x6, csrc/spconv.hip; kvol <= 27).
 * One kernel serves SubMConv3d (tbl = subm nbr), SparseConv3d (tbl = nbr_out), SparseInverseConv3d
 * (tbl = nbr_inv) and every nn.Linear on the path. */
int ls3d_gather_gemm(const float *in, int in_ld, const int32_t *tbl, const int32_t *row_order, int kvol, const float *w, int nt, int wc,
                     int precision /* must equal the packing's; BF16X3 / BF16X6 need cin % 32 == 0 and wc == 1 */, int cin, int cout,
                     int n_rows, const int32_t *n_rows_dev, const ls3d_epilogue_t *epi_host, float *out,
                     int out_ld, int flags, ls3d_stream_t stream);

/* ---- Tile-halo sparse convolution: the same operator as ls3d_gather_gemm with a rulebook table (spconv v1.x SubMConv3d /
 * SparseConv3d / SparseInverseConv3d, call sites det3d/models/backbones/scn_unet.py:15-24,34-69), organised for locality:
 * output rows are ordered along a space-filling key and cut into tiles of 128; each tile's UNIQUE input rows (its halo) are
 * staged once per 16-channel chunk in LDS as three bf16 planes (exact 8+8+8-bit split of the f32 mantissa) and all kernel
 * offsets run out of LDS on v_mfma_f32_32x32x16_bf16 with f32 accumulation.
 *   ls3d_tile_keys : key[r] = (batch << 2m) | Morton(y >> s, x >> s) of an output site (s = 2 unless the key would not
 *                    fit 31 bits); rows beyond *n_dev get 0x7FFFFFFF.  The caller sorts the keys (stable, ascending) and passes
 *                    the resulting row permutation as `spatial_order`.
 *   ls3d_tile_build: plan for one table tbl[n_rows, kvol] (kvol <= 32): per tile the output rows (sorted by neighbour mask,
 *                    densest first), the sorted unique input rows, the local table tloc[k][slot] -> position in that list,
 *                    the tile's / waves' offset masks.  `plan` holds ls3d_tile_plan_bytes(n_rows, kvol) bytes, 16-byte aligned.
 *                    Deterministic.  Any table and any order are valid; only speed depends on their locality.
 *   ls3d_tile_conv_pack: plain W[kvol][cin_src][cout] f32 -> [kvol][cin_pad/16][cout32/32][plane 3][2][32] x 8 bf16
 *                    (ls3d_tile_conv_packed_bytes bytes); more than 128 output columns: slab after slab of 128 columns, each in
 *                    that layout.  ls3d_tile_conv_pack_bf16 / ls3d_tile_conv_packed_bytes_bf16: the head plane only, for products = 1
 *                    (a third of the bytes: a 12 KB step of the kernel's weight stream then carries 3 / 6 / 12 kernel offsets).
 *   ls3d_tile_conv : out[r, 0..cout) = epilogue(sum_k W[k]^T in[tbl[r,k]]) for the rows of the plan.  products = 8: every
 *                    plane product except tail x tail (2^-32 relative) — f32-grade: the result differs from exact f32
 *                    arithmetic by less than f32 summation-order noise; products = 6: the BF16X6 arithmetic; products = 1: plain
 *                    bf16 operands (the head plane of both), f32 accumulation - NOT f32-grade (BASELINE configs[4]); w_packed
 *                    is then the single-plane layout of ls3d_tile_conv_pack_bf16.
 *                    cin % 16 == 0, in_ld % 4 == 0.  cout > 128 (SCALING_RATIO > 2 of scn_unet.py:88-123): one launch per slab of
 *                    128 columns on the same plan, workspace and counters (no LayerNorm epilogue then: LS3D_ERR_UNSUPPORTED).
 *                    Summation order per output row is fixed by the plan.
 *                    Workgroups are dispatched in the plan's most-expensive-tile-first order (ls3d_tile_build's last step).
 *                    workspace (optional, ls3d_tile_conv_workspace_bytes(n_rows, cout) bytes, 16-byte aligned, per call) +
 *                    counters (optional, ls3d_tile_conv_counter_bytes() bytes): let layers of cin >= 64 in launches of <= 512
 *                    tiles (fewer than the 2 x 256 workgroup slots of the chip) run TWO work units per tile, each over half of
 *                    the input channels; the unit that finishes second adds the other's partial sums (a + b == b + a: results do
 *                    not depend on which) and runs the epilogue.  The counters are caller-owned: zeroed ONCE, then handed to
 *                    any number of calls that are ordered one after the other (one array per stream); a completed call leaves
 *                    every counter even, nothing is reset.  A plan is read-only for ls3d_tile_conv: one plan may serve
 *                    concurrent launches on different streams, each with its own workspace and counters.  Which tiles are
 *                    split is a function of the plan and of `flags` only, so results are bit-reproducible.
 *                    flags (per call, 0 = defaults): bit 0: the plain offset loop instead of the software-pipelined one (products
 *                    == 6, cout > 32: fragments of the next offset are read under the current one's MFMAs, the step barrier sits in
 *                    the middle of an offset) and bit 1: the general two-pass epilogue instead of the single-pass one - both for
 *                    A/B runs, bit-identical results; bits 2-4 timing ablations for profiling (skip the MFMAs / the weight DMA /
 *                    the halo staging: results invalid); bits 6-7 channel split (0 = the rule above, 1 = never, 2 = as many
 *                    tiles as the workspace allows, 3 = the rule above + in launches of more than 512 tiles the [live tiles mod 512]
 *                    tiles at the end of the dispatch order: measured -1 % on the 120k frame, off by default); bits 8-19: number of split tiles (the last ones of the dispatch order) + 1
 *                    for any launch; bits 20-29: number of split tiles + 1 for launches of more than 512 tiles (default: none -
 *                    measured without gain); bit 30: halo planes in LDS without the bank swizzle.  Every flag value gives the
 *                    same bits for the rows of unsplit tiles.  bit 5 (products == 6): run the tracing build of the kernel - same
 *                    results; every wave of every work unit writes a 16-word record (100 MHz wall clock at start / end, HW_ID,
 *                    XCC_ID, tile, unit kind, halo rows, offset counts, shader cycles: total / prologue / halo staging / waits at
 *                    the step barriers / epilogue, steps) into the workspace behind the partial sums, at byte offset
 *                    ls3d_tile_conv_workspace_bytes(n_rows, cout): the workspace must then hold that many bytes +
 *                    ls3d_tile_conv_trace_bytes(n_rows) (records [unit][wave][16] uint32, unit = blockIdx; tools/trace_tile.py).
 *   ls3d_tile_build / ls3d_tile_plan flags: bit 0 = dispatch the tiles in plan (spatial) order instead of most expensive first;
 *                    bit 1 (ls3d_tile_plan on a 3x3x3 table whose input sites are its output sites - SubM; ignored otherwise): the
 *                    COLOURED halo layout - a tile's rows are dealt by a linear colour of their coordinates (mod 16) to the lane groups
 *                    of the kernel's LDS reads and its halo rows get LDS slots = colour (mod 16), so that the 16 rows a group gathers
 *                    at any offset sit in different bank columns (csrc/tileconv.hip: tc_color) instead of the neighbour-mask order.
 *                    Placement only: ls3d_tile_conv results are bit-identical with either layout. */
int ls3d_tile_keys(const int32_t *coords /*[n,4] b,z,y,x*/, int n, const int32_t *n_dev, const int32_t shape_zyx_host[3], int batch,
                   uint32_t *keys, ls3d_stream_t stream);
size_t ls3d_tile_plan_bytes(int n_rows, int kvol);
int ls3d_tile_build(const int32_t *tbl, int n_rows, const int32_t *n_rows_dev, int kvol, const int32_t *spatial_order, void *plan,
                    size_t plan_bytes, int flags, ls3d_stream_t stream);
/* keys -> sort -> plan in one call (what a caller does for every table of a frame): ls3d_tile_keys, the in-library stable radix
 * sort, ls3d_tile_build back to back on `stream`.  workspace: ls3d_tile_plan_workspace_bytes(n_rows) bytes, 16-byte aligned; it
 * holds the spatial order at byte offset roundup(4 n_rows, 256) afterwards. */
size_t ls3d_tile_plan_workspace_bytes(int n_rows);
int ls3d_tile_plan(const int32_t *tbl, const int32_t *coords, int n_rows, const int32_t *n_rows_dev, int kvol,
                   const int32_t shape_zyx_host[3], int batch, void *workspace, size_t workspace_bytes, void *plan, size_t plan_bytes,
                   int flags, ls3d_stream_t stream);
/* stable LSD radix sort of (uint32 key, int32 value) pairs by the low `bits` key bits, ascending; vals == NULL: the values are
 * the positions 0..n-1 (the result is the sorting permutation).  keys_out may be NULL.  Not in place.  n_dev (optional): only the
 * first min(*n_dev, n) pairs are sorted, into the first positions of the outputs (the rest is unspecified).  Replaces torch.argsort
 * for the row orders of the sparse convolutions (spatial tile keys, neighbour-mask keys). */
size_t ls3d_radix_sort_workspace_bytes(int n);
int ls3d_radix_sort(const uint32_t *keys, const int32_t *vals, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                    void *workspace, size_t workspace_bytes, ls3d_stream_t stream);
size_t ls3d_tile_conv_packed_bytes(int kvol, int cin_pad, int cout);
int ls3d_tile_conv_pack(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, void *w_packed, ls3d_stream_t stream);
size_t ls3d_tile_conv_packed_bytes_bf16(int kvol, int cin_pad, int cout);
int ls3d_tile_conv_pack_bf16(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, void *w_packed, ls3d_stream_t stream);
size_t ls3d_tile_conv_workspace_bytes(int n_rows, int cout);
size_t ls3d_tile_conv_counter_bytes(void);
size_t ls3d_tile_conv_trace_bytes(int n_rows);
int ls3d_tile_conv(const float *in, int in_ld, const void *plan, int n_rows, int kvol, const void *w_packed, int cin, int cout,
                   int products, const ls3d_epilogue_t *epi_host, float *out, int out_ld, void *workspace, size_t workspace_bytes,
                   int32_t *counters, int flags, ls3d_stream_t stream);

/* Chained launch: up to 8 consecutive SubM layers that share ONE plan (the convolutions of the SparseBasicBlocks of a UNet level,
 * scn_unet.py:34-69, 189-249: each layer reads the previous layer's output on the same sites) in ONE persistent launch.  Work units
 * (layer, tile) are taken from a ticket counter in layer-major order; a tile of layer l + 1 starts as soon as the tiles that own its halo
 * rows (the plan's producer lists, built by ls3d_tile_build) have finished layer l - so the tail of a layer (677 tiles on 512 workgroup
 * slots) is filled with the next layer's tiles instead of idle CUs.  A unit waits only for units with smaller tickets, so the launch cannot
 * deadlock whatever the number of resident workgroups; a wait of more than ~1 s sets state[1] (int32) and traps (the process aborts: no rows computed from a missing halo are published).
 *   layers[l]: the operands of ls3d_tile_conv for layer l (epilogue by value).  A layer may read anything an EARLIER layer of the chain wrote
 *              on the plan's rows (its input, res_pre, pair) and anything written before the launch.  All layers: cout in the same class
 *              (<= 32, <= 64, <= 128), cout % 4 == 0, float4-aligned leading dimensions, no LayerNorm epilogue, products == 6,
 *              n_rows * ld * 4 < 4 GiB - else LS3D_ERR_UNSUPPORTED (the caller launches the layers one by one).
 *   state:     ls3d_tile_chain_state_bytes(n_rows) bytes of per-call scratch, 16-byte aligned (tickets, completion counters, the layer
 *              table; initialised by the call).  workspace / counters / flags bits 6-7, 30: as for ls3d_tile_conv (the split over the
 *              input channels is decided layer by layer by the same rule).
 * Results are bit-identical to n_layers calls of ls3d_tile_conv: a unit runs the same code on the same tile in the same summation order;
 * only the data movement between layers (coherent sc1 accesses across the XCDs' L2s) and the dispatch order of the tiles (spatial) differ. */
typedef struct {
  const float *in;
  int32_t in_ld;
  const void *w_packed;
  int32_t cin, cout;
  ls3d_epilogue_t epi;
  float *out;
  int32_t out_ld;
} ls3d_tile_chain_layer_t;
size_t ls3d_tile_chain_state_bytes(int n_rows);
int ls3d_tile_conv_chain(const void *plan, int n_rows, int kvol, const ls3d_tile_chain_layer_t *layers, int n_layers, int products, void *state,
                         size_t state_bytes, void *workspace, size_t workspace_bytes, int32_t *counters, int flags, ls3d_stream_t stream);

/* Backward of the sparse convolutions (spconv v1.x indice_conv_backward).
 *   grad_in : ls3d_gather_gemm on grad_out with the TRANSPOSED table (SubM: the same table; SparseConv3d: nbr_inv;
 *             SparseInverseConv3d: nbr_out) and weights W'[k] = W[k]^T (SubM: W'[k] = W[kvol-1-k]^T) - no extra entry point;
 *   grad_w  : grad_w[k][ci][co] = sum over rows o with tbl[o][k] >= 0 of in[tbl[o][k]][ci] * grad_out[o][co], tbl = the table
 *             the FORWARD launch used, plain [kvol][cin][cout] layout (the layout of the module's weight); any cout (slabs of 128
 *             columns of grad_out over one transposed table).
 * Deterministic (fixed summation order).  products: 0 = exact f32 MFMA (two rows per v_mfma_f32_32x32x2_f32); 6 | 8 = the exact 3-plane
 * bf16 split of both operands (16 rows per v_mfma_f32_32x32x16_bf16, operands transposed through LDS, head x head in its own accumulator):
 * f32-grade like ls3d_tile_conv; used for layers with >= 4 output blocks of 32 x 32 (64 -> 64 and wider), narrower ones run the exact-f32 kernel either way. */
size_t ls3d_spconv_wgrad_workspace_bytes(int kvol, int cin, int cout, int n_rows);
int ls3d_spconv_wgrad(const float *in, int in_ld, const float *grad_out, int grad_out_ld, const int32_t *tbl, const int32_t *row_order,
                      int kvol, int cin, int cout, int n_rows, const int32_t *n_rows_dev, int products, void *workspace,
                      size_t workspace_bytes, float *grad_w, ls3d_stream_t stream);
/* The same in two steps, for layers that share one table (the SubM layers of a UNet level share their indice_key's table): the compacted
 * (input row, output row) lists per kernel offset are built once - ls3d_spconv_pairs, an opaque 16-byte-aligned device buffer of
 * ls3d_spconv_pairs_bytes(kvol, n_rows) - and every layer's weight gradient runs on them.  ls3d_spconv_wgrad is exactly
 * ls3d_spconv_pairs into its own workspace followed by ls3d_spconv_wgrad_on_pairs: the two give bit-identical gradients.
 * workspace of ls3d_spconv_wgrad_on_pairs: ls3d_spconv_wgrad_workspace_bytes (only its partial-sum part is used). */
size_t ls3d_spconv_pairs_bytes(int kvol, int n_rows);
/* identity lists (kvol = 1: the tall-skinny weight gradient of a Linear layer) for a row capacity: build != 0 writes the lists of n_rows_cap
 * rows, every call sets the number of valid rows to n_rows <= n_rows_cap; pass n_rows_cap as n_rows of ls3d_spconv_wgrad_on_pairs / its
 * workspace query.  pairs: ls3d_spconv_pairs_bytes(1, n_rows_cap) bytes. */
int ls3d_spconv_identity_pairs(int n_rows_cap, int n_rows, int build, void *pairs, size_t pairs_bytes, ls3d_stream_t stream);
int ls3d_spconv_pairs(const int32_t *tbl, const int32_t *row_order, int n_rows, const int32_t *n_rows_dev, int kvol, void *pairs,
                      size_t pairs_bytes, ls3d_stream_t stream);
int ls3d_spconv_wgrad_on_pairs(const float *in, int in_ld, const float *grad_out, int grad_out_ld, const void *pairs, int kvol, int cin,
                               int cout, int n_rows, int products, void *workspace, size_t workspace_bytes, float *grad_w,
                               ls3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Devoxelization
 * ---------------------------------------------------------------------------------------------- */

/* Row offsets of the frames of a frame-sorted table (points [n,stride] f32 or coordinates [n,stride] int32 with the
 * batch index in column `col`): off[b] = first row with batch index >= b, b = 0..batch (off[batch] = n).  Replaces the
 * reference's per-frame boolean masks (`coords[:, 0] == i`, point_utils.py:20-21, context_module.py:39,350). */
int ls3d_frame_offsets(const void *table, int is_float, int stride, int col, int n, const int32_t *n_dev, int batch, int32_t *off,
                       ls3d_stream_t stream);

/* voxel centres: out[v] = (b, (x+.5)*vx+x0, (y+.5)*vy+y0, (z+.5)*vz+z0), f32 mul then add (unfused),
 * det3d/core/utils/common_utils.py:74-90 + scn_unet.py:243-247. */
int ls3d_voxel_centers(const int32_t *coords, int n, const int32_t *n_dev, const float vs_host[3],
                       const float lo_host[3], float *out, ls3d_stream_t stream);

/* three_nn_wrapper(b,n,m,unknown,known,dist2,idx), det3d/ops/pointnet2_batch/src/pointnet2_api.cpp:21,
 * interpolate_gpu.cu:16-59: exact brute-force 3 nearest, squared f32 distance evaluated as
 * fma(dz,dz,fma(dy,dy,dx*dx)), strict '<' (lowest index wins ties), dist2 = +inf / idx = 0 for m < 3.
 * unknown (b,n,3), known (b,m,3) contiguous. */
int ls3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                  ls3d_stream_t stream);

/* three_interpolate_wrapper(b,c,m,n,points,idx,weight,out), pointnet2_api.cpp:22, interpolate_gpu.cu:84-104:
 * channel-major points (b,c,m) -> out (b,c,n). */
int ls3d_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx,
                           const float *weight, float *out, ls3d_stream_t stream);

/* three_interpolate_grad_wrapper, pointnet2_api.cpp:23, interpolate_gpu.cu:127-149 (training only). */
int ls3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx,
                                const float *weight, float *grad_points, ls3d_stream_t stream);

/* The whole of point_utils.three_interpolate_wrap (det3d/models/point_heads/point_utils.py:8-52) for a
 * collated batch: per frame f, for the points rows [pt_off[f], pt_off[f+1]) find the 3 nearest voxel
 * centres among rows [vx_off[f], vx_off[f+1]) of centers[*,4] (col 0 = batch), weights
 * w_j = (1/(sqrt(d2_j)+1e-8))/sum, out[p, 0..c) = sum_j w_j * feat[vx_off[f]+idx_j].
 * points[*, pt_stride] with xyz at columns 1..3.  idx_out[n,3] (frame-relative, may be NULL). */
int ls3d_devoxelize(const float *points, int pt_stride, int n_points, const int32_t *pt_off /*[batch+1] dev*/,
                    const float *centers, const int32_t *vx_off /*[batch+1] dev*/, int batch, int max_frame_points,
                    const float *feat, int feat_ld, int c, float *out, int out_ld, int32_t *idx_out,
                    ls3d_stream_t stream);

size_t ls3d_devoxelize_grid_workspace_bytes(int n_points, int n_voxels, int batch, const int32_t grid_xyz_host[3]);

/* Same result as ls3d_devoxelize (bit-identical neighbours and weights) when the known points are voxel
 * centres on a regular lattice: the centres are binned into coarse cells (counting sort) and each point
 * searches expanding shells of coarse cells with an exact stopping bound, instead of scanning every centre
 * (the reference kernel is O(N*V), interpolate_gpu.cu:36-57).  coords[V,4] = integer (batch,z,y,x) of the
 * centres, centers[V,4] as written by ls3d_voxel_centers, grid_xyz = fine cells per axis. */
int ls3d_devoxelize_grid(const float *points, int pt_stride, int n_points, const int32_t *pt_off /*[batch+1] dev*/,
                         int max_frame_points, const int32_t *coords, const float *centers,
                         int n_voxels, const int32_t *n_voxels_dev, const int32_t *vx_off /*[batch+1] dev*/, int batch,
                         const float vs_host[3], const float lo_host[3], const int32_t grid_xyz_host[3], const float *feat,
                         int feat_ld, int c, float *out, int out_ld, int32_t *idx_out, float *weight_out /*[n,3] or NULL*/,
                         void *workspace, size_t workspace_bytes, ls3d_stream_t stream);
/* feat == NULL: search only (idx_out and weight_out required) - the neighbour search depends on geometry alone, so it can run
 * on another stream while the features are still being computed; ls3d_interpolate_rows then finishes the job:
 * out[p] = sum_j weight[p][j] * feat[vx_off[frame(p)] + idx[p][j]] (same arithmetic as the fused call). */
int ls3d_interpolate_rows(const float *feat, int feat_ld, int c, const int32_t *idx, const float *weight, const float *points,
                          int pt_stride, const int32_t *vx_off, int n_points, float *out, int out_ld, ls3d_stream_t stream);
/* Backward of ls3d_interpolate_rows with respect to the voxel features: grad_feat[v] = sum over the (point, neighbour) entries that refer to row v
 * of weight * grad_out[point] (three_interpolate_grad, interpolate_gpu.cu:127-149, which scatters with atomicAdd: run-to-run different sums).
 * Here the entries are sorted by voxel row (stable radix sort) and summed in entry order: bit-reproducible.  Every row of grad_feat[n_voxels,
 * c] is written.  workspace: ls3d_interpolate_rows_backward_workspace_bytes(n_points, n_voxels) bytes, 16-byte aligned. */
size_t ls3d_interpolate_rows_backward_workspace_bytes(int n_points, int n_voxels);
int ls3d_interpolate_rows_backward(const float *grad_out, int go_ld, int c, const int32_t *idx, const float *weight, const float *points,
                                   int pt_stride, const int32_t *vx_off, int n_points, int n_voxels, void *workspace, size_t workspace_bytes,
                                   float *grad_feat, int gf_ld, ls3d_stream_t stream);

/* The per-point tail of PointSegBatchlossHead at inference (det3d/models/point_heads/point_seg_batchloss_head.py:141-168) in ONE launch:
 *   x[p] = sum_j weight[p][j] * feat[vx_off[frame(p)] + idx[p][j]]   (the arithmetic of ls3d_interpolate_rows; idx == NULL: x[p] = feat[p])
 *   for every layer: x = [relu](x W * scale + shift)                  (Linear + BatchNorm(eval) + ReLU folded; scale / shift may be NULL)
 *   out[p, 0:cout_last] = x;  labels[p] = argmax over the classes with torch.argmax's tie rule (labels may be NULL).
 * W: plain [cin][cout] f32 row-major on the device.  c_in and the hidden widths are 32 or 64, the last layer's cout <= 64, <= 6 layers;
 * anything else: LS3D_ERR_UNSUPPORTED (compose it from ls3d_interpolate_rows + ls3d_gather_gemm).  Exact f32 products (f32 MFMA). */
typedef struct {
  const float *w, *scale, *shift;
  int32_t cin, cout, relu;
} ls3d_point_mlp_layer_t;
int ls3d_point_mlp(const float *feat, int feat_ld, int c_in, const int32_t *idx, const float *weight, const float *points, int pt_stride,
                   const int32_t *vx_off, int n, int num_layers, const ls3d_point_mlp_layer_t *layers_host, float *out, int out_ld,
                   int64_t *labels, ls3d_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LiDAR-camera fusion (MSeg3D point head)
 * ---------------------------------------------------------------------------------------------- */

/* get_points_image_feature (det3d/models/point_heads/point_seg_mseg3d_head.py:200-236): 5-D grid_sample of
 * image_features[batch,ncam,c,h,w] at points_cuv[n,4] = (valid, cam, h, w in [-1,1]), trilinear over
 * (cam,h,w), zeros padding, align_corners=True.  Rows with valid != 1 get zeros.  out[n, c] ld out_ld. */
/* channels_last != 0: image_features is [batch,ncam,h,w,c] (ls3d_nchw_to_nhwc of the reference layout): a point's c channels
 * are contiguous per corner; same arithmetic and results. */
int ls3d_grid_gather(const float *image_features, int batch, int ncam, int c, int h, int w, int channels_last,
                     const float *points_cuv, const float *points, int pt_stride, int n, float *out, int out_ld,
                     ls3d_stream_t stream);
/* [planes][c][hw] -> [planes][hw][c] */
int ls3d_nchw_to_nhwc(const float *in, int planes, int c, int hw, float *out, ls3d_stream_t stream);

/* feature completion (point_seg_mseg3d_head.py:314-334) fused with the concat of :341:
 * lc[p, 0..c_l) = lidar[p], lc[p, c_l..c_l+c_c) = valid[p] ? camera[p] : pseudo[p].
 * pseudo == NULL means zeros: that is what the reference's forward produces, because it evaluates the mimic
 * layer on the VALID points only and zero-pads the rest before the torch.where (:305,:320-334). */
int ls3d_complete_concat(const float *lidar, int c_l, const float *camera, const float *pseudo, int c_c,
                         const float *points_cuv, int n, float *lc, ls3d_stream_t stream);

/* LiDARSemanticFeatureAggregationModule.forward (context_module.py:25-53): per frame softmax of
 * logits[V,cls] over the voxels of the frame, emb[f, cls, c] = sum_v p[v,cls] * feats[v,c].
 * workspace: 2*batch*cls floats. */
int ls3d_sfam(const float *feats, int feat_ld, int c, const float *logits, int cls, const int32_t *vx_off, int batch,
              int max_frame_voxels, float *workspace, float *emb /*[batch,cls,c]*/, ls3d_stream_t stream);

/* The point side of the SF-Phase decoder as ONE kernel (SemanticFeatureFusionModule.forward, context_module.py:89-117, over
 * TransformerDecoderLayer.forward_post :211-250 and SparsePointCorssAttention :320-376): out[n, d_model] = norm_tgt(decoder(
 * input_proj_point(x))) with the points attending to the L class embeddings of their frame.  The embedding side does not depend on
 * the points: the caller passes, per layer, k and v exactly as ls3d_cross_attn takes them: kv = [num_layers][2 (k, v)][batch][d_model][L].
 * Every weight matrix is an (in, out) matrix of 96 output columns in the layout of ls3d_gather_gemm_pack(nt = 3, F32): w_in
 * [d_in x 96]; per layer wq, wo [96 x 96], the FFN as column halves of linear1 (w1a, w1b: [96 x 96] each, b1: all 192 biases) and
 * row halves of linear2 (w2a, w2b).  Supported: d_model 96, 4 heads, ffn 192, L <= 64 (matrix-pipe attention for L <= 36), d_in in
 * {32, 64, 96}, num_layers <= 8;
 * anything else returns LS3D_ERR_UNSUPPORTED and the caller composes the decoder from ls3d_gather_gemm / ls3d_cross_attn. */
typedef struct {
  const float *wq, *bq, *wo, *bo, *w1a, *w1b, *b1, *w2a, *w2b, *b2;
  const float *n2_gamma, *n2_beta, *n3_gamma, *n3_beta;
  float n2_eps, n3_eps;
  /* gemm_products == 6: the six matrices again as three exact bf16 planes - ls3d_tile_conv_pack(plain [in][out], kvol 1, cin 96, cin_pad 96,
   * cout 96) of each with the INPUT channels of every 16-block reordered for the transposed product: packed row 16 c + 8 kk + q = plain row
   * 16 c + 8 (q / 4) + 4 kk + q % 4 (csrc/sffm.hip, k_sffm_decoder_rt); NULL otherwise */
  const void *wq_planes, *wo_planes, *w1a_planes, *w1b_planes, *w2a_planes, *w2b_planes;
} ls3d_sffm_layer_t;
typedef struct {
  const float *w_in, *b_in;
  const ls3d_sffm_layer_t *layers; /* HOST array */
  int32_t num_layers, d_in, d_model, heads, ffn;
  const float *norm_gamma, *norm_beta; /* decoder.norm_tgt, or NULL */
  float norm_eps;
  int32_t attention; /* arithmetic of the decoder's point -> class-embedding attention (QK^T and PV): 0 (default) exact f32 on
                      * v_mfma_f32_32x32x2_f32, 1 operands rounded to bf16 on v_mfma_f32_32x32x16_bf16, 3 operands rounded to OCP e4m3 on
                      * v_mfma_f32_32x32x16_fp8_fp8 - both with f32 accumulation and f32 softmax (BASELINE configs[4]); 2 the vector pipe (A/B) */
  const void *w_in_planes; /* gemm_products == 6: the input projection as three bf16 planes (ls3d_tile_conv_pack, cin_pad = d_in) */
  const int32_t *pt_off;   /* gemm_products == 6: DEVICE int32 [batch + 1], first row of every frame in the frame-sorted x (ls3d_frame_offsets):
                            * the register-resident kernel cuts its 128-point tiles per frame */
  int32_t gemm_products;   /* arithmetic of the decoder's GEMMs: 0 exact f32 on v_mfma_f32_32x32x2_f32; 6 both operands split exactly into three
                            * round-to-nearest bf16 planes, the six plane products of weight >= 2^-16 on v_mfma_f32_32x32x16_bf16, f32 accumulation,
                            * head x head in its own accumulator (the f32-grade arithmetic of ls3d_tile_conv) - the decoder then runs
                            * transposed with its activations in registers (needs pt_off, attention == 0, L <= 64) */
} ls3d_sffm_t;
int ls3d_sffm_decoder(const float *x, int x_ld, int n, const float *points, int pt_stride, const float *kv, int L, int batch,
                      const ls3d_sffm_t *model_host, float *out, int out_ld, ls3d_stream_t stream);

/* The class-embedding ("memory") side of the same decoder for all layers in one launch (context_module.py:147-171, :211-250, :320-338): per
 * frame the L <= 64 memory tokens [L][96] go through num_layers x { self-attention (4 heads) + residual + norm1 } and every layer's cross-attention
 * k_proj / v_proj (Conv1d, k = 1) is written as kv[2 l + {0: k, 1: v}][batch][96][L] - the `kv` input of ls3d_sffm_decoder.  Weights are the
 * modules' own matrices TRANSPOSED to [in][out] row-major f32 (in_proj_weight^T [96][288], out_proj.weight^T, k_proj / v_proj weight^T [96][96]).
 * GEMMs on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), one 16-wave workgroup per frame.  mem_out (optional): the memory after the last layer, [batch * L][96].  Other shapes
 * (embed != 96, heads != 4, L > 64, > 8 layers): LS3D_ERR_UNSUPPORTED, the caller composes it from ls3d_gather_gemm / ls3d_mha_core. */
typedef struct {
  const float *wqkv_t, *bqkv, *wo_t, *bo, *n1_gamma, *n1_beta, *wk_t, *bk, *wv_t, *bv;
  float n1_eps;
} ls3d_sffm_memory_layer_t;
int ls3d_sffm_memory(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers_host,
                     float *kv, float *mem_out, ls3d_stream_t stream);
/* diagnostics (tools/trace_memory.py): the same launch in a tracing build - trace[2 + 4 num_layers] shader-clock values of frame 0's first thread at the
 * phase boundaries (tokens staged; per layer: q | k | v projections, self-attention, out-projection, norm1; the last k / v projections) */
int ls3d_sffm_memory_trace(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers,
                           float *kv, unsigned long long *trace, ls3d_stream_t stream);

/* diagnostics (tools/probe_graph_timeline.py): *dst = the GPU's 100 MHz wall clock when `stream` reaches this launch (a one-lane kernel: inside a captured
 * frame it is a graph node, so a replay can be timed stream by stream) */
int ls3d_stamp(unsigned long long *dst, ls3d_stream_t stream);

/* SparsePointCorssAttention core (context_module.py:339-372): q[n,embed] (already projected), per-frame
 * k,v[batch, heads, embed/heads, L] (Conv1d outputs reshaped as the reference does), softmax(q.k*scale) v
 * -> out[n, embed].  frame of point p = (int)points[p*pt_stride]. */
int ls3d_cross_attn(const float *q, const float *k, const float *v, int batch, int heads, int embed, int L,
                    const float *points, int pt_stride, int n, float *out, ls3d_stream_t stream);


/* ------------------------------------------------------------------------------------------------
 * Camera-branch input step on the GPU
 * ---------------------------------------------------------------------------------------------- */

/* points_cp (det3d/datasets/pipelines/loading.py:384-413): per camera c (in order, later cameras win):
 *   X = cams_from_global[c] * (ref_to_global * [x y z 1]);  (u, v) = (K[c] X)_{0,1} / (K[c] X)_2    (float64)
 *   visible iff X_z > 0 and 1 < u < im_w - 1 and 1 < v < im_h - 1
 * points_cp[n,3] = (cam_id + 1, u, v) as float32, (-100,-100,-100) when no camera sees the point.
 * Matrices are HOST arrays, row-major: ref_to_global[16], cams_from_global[ncam*16], intrinsics[ncam*9]; ncam <= 8. */
int ls3d_points_cp(const float *points, int pt_stride, int xyz_col, int n, const double *ref_to_global,
                   const double *cams_from_global, const double *intrinsics, int ncam, int im_h, int im_w,
                   float *points_cp, ls3d_stream_t stream);

/* points_cuv (det3d/datasets/pipelines/segpreprocess.py:649-671): points_cuv[n,4] = (valid = cam_id > 0,
 * (cam_id-1)/(ncam-1)*2-1 (0 if ncam == 1), v/(res_h-1)*2-1, u/(res_w-1)*2-1), float32 in that operation order;
 * points_cp is expected in the coordinates of the res_h x res_w feature-map input. */
int ls3d_points_cuv(const float *points_cp, int n, int ncam, int res_h, int res_w, float *points_cuv, ls3d_stream_t stream);

/* Segmentation loss of the point heads: cross entropy with an ignored label + Lovasz-Softmax over the classes present, forward and
 * backward (det3d/core/utils/loss_utils.py:217-291 lovasz_softmax(classes='present', ignore) + F.cross_entropy(ignore_index), as
 * point_seg_batchloss_head.py:77-121 / point_seg_mseg3d_head.py:137 call them on flat [P, C] logits).  One softmax pass, ONE batched
 * radix sort of the C x P class errors, per-class scans; no host synchronisation, deterministic.  num_classes <= 32; labels outside
 * [0, num_classes) count as ignored (torch raises for them).
 *   forward : out2[0] = mean_i(-log softmax(logits_i)[label_i]) over label_i != ignore_index (nan when there is none, as torch),
 *             out2[1] = mean over the classes c present of sum_j e_(j) g_j, e = |[label == c] - softmax_c| sorted descending, g = the
 *             increments of the Jaccard index (Lovasz gradient); the workspace keeps what the backward needs.
 *   backward: grad_logits[P, ld] = grad_ce * d ce / d logits + grad_lovasz * d lovasz / d logits (device scalars; NULL = 1), from the
 *             workspace the forward filled for the SAME logits / labels.  It reads the first ls3d_seg_loss_saved_bytes() bytes only (softmax,
 *             Lovasz gradient, counts): a caller that holds the workspace until the backward may keep that prefix and release the rest
 *             (5 of the 7 [P, C] arrays and the sort histograms). */
size_t ls3d_seg_loss_workspace_bytes(int n_points, int num_classes);
size_t ls3d_seg_loss_saved_bytes(int n_points, int num_classes);
int ls3d_seg_loss_forward(const float *logits, int ld, const int32_t *labels, int n_points, int num_classes, int ignore_index, void *workspace,
                          size_t workspace_bytes, float *out2, ls3d_stream_t stream);
int ls3d_seg_loss_backward(const int32_t *labels, int n_points, int num_classes, int ignore_index, const void *workspace, size_t workspace_bytes,
                           const float *grad_ce, const float *grad_lovasz, float *grad_logits, int ld, ls3d_stream_t stream);

/* BatchNorm1d over [n, c] rows in training mode (batch statistics) with the ReLU / residual add that follow it in the UNet fused in
 * (det3d/models/backbones/scn_unet.py:11-69; nn.BatchNorm1d semantics: biased variance for the normalisation), c % 4 == 0, c <= 256, 256 % (c / 4) == 0:
 *   ls3d_batch_norm_stats         : mean_m2[2 c + 1] = per-column mean and sum of squared deviations of the n rows, then n itself as a float (one pass
 *                                   over x, per-block partials merged with Chan's update in a fixed-shape tree: deterministic, no cancellation);
 *                                   workspace = ls3d_batch_norm_workspace_bytes;
 *   ls3d_batch_norm_finalize      : parts[world][2 c + 1] (the triples of `world` ranks; world == 1 and local_n >= 0: parts[0 .. 2 c) with n = local_n)
 *                                   -> mean[c], var[c] (biased), rstd[c] = 1 / sqrt(var + eps), count_out[1] = total rows, and - running_mean /
 *                                   running_var != NULL - nn.BatchNorm's running update (momentum, unbiased variance; num_batches_tracked += 1 when given);
 *   ls3d_batch_norm_apply         : y = [relu]((x - mean) * rstd * gamma + beta [+ res]);
 *   ls3d_batch_norm_backward_sums : sums[2 c] = column sums of g and of g * xhat, g = dy * [y > 0] (y_or_null = NULL: g = dy) - exposed so that a
 *                                   data-parallel step can all-reduce them (lidarseg3d_amd/syncbn.py);
 *   ls3d_batch_norm_backward_apply: dx = gamma rstd (g - sums[0..c) * inv_count - xhat * sums[c..2c) * inv_count), dres = g (dres may be NULL);
 *                                   count_dev != NULL: inv_count = 1 / max(count_dev[0], 1) read on the device (ls3d_batch_norm_finalize's count_out). */
size_t ls3d_batch_norm_workspace_bytes(int n, int c);
int ls3d_batch_norm_stats(const float *x, int ld, int n, int c, void *workspace, size_t workspace_bytes, float *mean_m2, ls3d_stream_t stream);
int ls3d_batch_norm_finalize(const float *parts, int world, int c, int local_n, float eps, float momentum, float *running_mean, float *running_var,
                             int64_t *num_batches_tracked, float *mean, float *var, float *rstd, float *count_out, ls3d_stream_t stream);
int ls3d_batch_norm_apply(const float *x, int ld, int n, int c, const float *mean, const float *rstd, const float *gamma, const float *beta,
                          const float *res, int res_ld, int relu, float *y, int y_ld, ls3d_stream_t stream);
int ls3d_batch_norm_backward_sums(const float *x, int ld, const float *dy, const float *y_or_null, int n, int c, const float *mean, const float *rstd,
                                  void *workspace, size_t workspace_bytes, float *sums, ls3d_stream_t stream);
int ls3d_batch_norm_backward_apply(const float *x, int ld, const float *dy, const float *y_or_null, int n, int c, const float *mean, const float *rstd,
                                   const float *gamma, const float *sums, float inv_count, const float *count_dev, float *dx, float *dres,
                                   ls3d_stream_t stream);

/* Point -> class-token attention of the SF-Phase decoder under autograd (context_module.py:222-257): n point queries q[n, heads * hd] against the
 * frame's L tokens, k / v [heads, hd, L] (the [E, L] layout of the reference's k_proj / v_proj outputs), per point and head
 *   forward : out = softmax(scale q_h K_h) V_h                                                        -> out[n, heads * hd]
 *   backward: dq[n, heads * hd], dk / dv [heads, hd, L] (the sums over the points: row blocks on the matrix pipe, two points per
 *             v_mfma_f32_32x32x2_f32 step, then the blocks' partials in order - deterministic, no atomics); the probabilities are recomputed from q,
 *             the forward keeps nothing.  workspace: ls3d_token_attention_workspace_bytes(n, heads, L) bytes, 16-byte aligned.
 * Exact f32; one thread per (point, head), K / V through the scalar cache.  hd == 24, L in {34, 38, 40, 46} (2 x 17 / 19 / 20 / 23 classes):
 * LS3D_ERR_UNSUPPORTED otherwise (the caller composes it from GEMMs).  q / dout / out / dq 16-byte aligned. */
int ls3d_token_attention_forward(const float *q, int n, int heads, int hd, const float *k, const float *v, int L, float scale, float *out,
                                 ls3d_stream_t stream);
size_t ls3d_token_attention_workspace_bytes(int n, int heads, int L);
int ls3d_token_attention_backward(const float *q, const float *dout, int n, int heads, int hd, const float *k, const float *v, int L, float scale,
                                  float *dq, float *dk, float *dv, void *workspace, size_t workspace_bytes, ls3d_stream_t stream);

/* out[c] = column sums of x[n, c] (row stride ld floats; c <= 256; float4 loads when c, ld are multiples of 4 and x is 16-byte aligned): the bias gradient of a
 * Linear layer over 10^5 - 10^6 point rows (torch.nn.functional.linear's backward, sum of grad_out over the rows).  Row blocks + a fixed tree:
 * deterministic.  workspace: ls3d_column_sums_workspace_bytes(n, c). */
size_t ls3d_column_sums_workspace_bytes(int n, int c);
int ls3d_column_sums(const float *x, int ld, int n, int c, void *workspace, size_t workspace_bytes, float *out, ls3d_stream_t stream);

/* Row LayerNorm over [n, c] (rows contiguous, c % 4 == 0, c <= 256), forward and backward, for the training step: the LayerNorms of the
 * reader (voxel_encoder.py:149-163) and of the SF-Phase decoder (context_module.py:319-376) over 10^5 - 10^6 token rows.
 *   forward : y = (x - mean) * rstd * gamma + beta per row, biased variance + eps as torch.nn.LayerNorm; stats[n][2] = (mean, rstd) for
 *             the backward (may be NULL).
 *   backward: dx, dgamma[c], dbeta[c] from x, dy, gamma and the forward's stats; workspace = ls3d_layer_norm_workspace_bytes(n, c)
 *             bytes (per-block column sums, reduced in a fixed order: deterministic). */
size_t ls3d_layer_norm_workspace_bytes(int n, int c);
int ls3d_layer_norm_forward(const float *x, int n, int c, const float *gamma, const float *beta, float eps, float *y, float *stats,
                            ls3d_stream_t stream);
int ls3d_layer_norm_backward(const float *x, const float *dy, const float *gamma, const float *stats, int n, int c, float *dx, float *dgamma,
                             float *dbeta, void *workspace, size_t workspace_bytes, ls3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LS3D_H */
