// pointmlp.hip — the per-point tail of PointSegBatchlossHead as ONE kernel (det3d/models/point_heads/point_seg_batchloss_head.py:141-168 at
// inference): 3-NN interpolation of the voxel features -> conv_align_layers -> out_cls_layers -> logits (+ argmax).  Layer by layer that is
// ls3d_interpolate_rows + four dense ls3d_gather_gemm launches + torch.argmax on 120 000 rows: seven latency-bound launches (10 - 22 us each)
// with every intermediate [N, 64] through HBM.  Here a wave keeps its 32 points as the COLUMNS of the MFMA (as k_sffm_decoder_rt does):
//     Y^T[channel][point] = W^T X^T,   weights = A operand (from LDS, staged once per workgroup), activations = B operand (registers).
// In the C layout of v_mfma_f32_32x32x2_f32 lane (point, kk) holds channels 32 n + 8 i + 4 kk + j (register 4 i + j of block n) of ITS point, and
// the next product wants from lane (point, kk) one K value per step: step (n, i, j) takes register 4 i + j of block n, i.e. K index 32 n + 8 i + 4 kk
// + j - the K order of an MFMA is free as long as both operands agree, so a layer's output registers are the next layer's input registers.
// Exact f32 products, f32 accumulation.  BatchNorm(eval) / bias as per-channel scale / shift, ReLU, the classifier's argmax with torch.argmax's
// tie rule (lowest index; a NaN wins).
#include "common.h"

typedef float pm_f32x16 __attribute__((ext_vector_type(16)));
constexpr int PM_MAX_LAYERS = 6;
constexpr int PM_WS = 72;  // LDS row stride of a weight matrix (64 columns + 8: the two lane halves read rows 4 apart - other banks)

struct PmLayer { const float *w, *scale, *shift; int cin, cout, relu, w_off, v_off; };
struct PmParams { int num_layers, c_in; PmLayer layer[PM_MAX_LAYERS]; };

// out[m] = sum_k W[k][32 m + col] x[k]: IB input blocks of 32 channels, OB output blocks
template <int IB, int OB>
__device__ __forceinline__ void pm_gemm(const pm_f32x16 (&x)[2], const float *Ws, pm_f32x16 (&y)[2], int col, int kk) {
#pragma unroll
  for (int m = 0; m < OB; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[m][r] = 0.0f;
#pragma unroll
  for (int n = 0; n < IB; ++n)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float *wr = Ws + (32 * n + 8 * i + 4 * kk + j) * PM_WS + col;
#pragma unroll
        for (int m = 0; m < OB; ++m) y[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(wr[32 * m], x[n][4 * i + j], y[m], 0, 0, 0);
      }
}

template <int OB>
__device__ __forceinline__ void pm_epilogue(pm_f32x16 (&y)[2], const float *vec, int relu, int kk) {
#pragma unroll
  for (int m = 0; m < OB; ++m)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 sc = *(const float4 *)(vec + 32 * m + 8 * i + 4 * kk), sh = *(const float4 *)(vec + 64 + 32 * m + 8 * i + 4 * kk);
      float v0 = y[m][4 * i + 0] * sc.x + sh.x, v1 = y[m][4 * i + 1] * sc.y + sh.y, v2 = y[m][4 * i + 2] * sc.z + sh.z, v3 = y[m][4 * i + 3] * sc.w + sh.w;
      if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
      y[m][4 * i + 0] = v0; y[m][4 * i + 1] = v1; y[m][4 * i + 2] = v2; y[m][4 * i + 3] = v3;
    }
}

__global__ __launch_bounds__(256, 2) void k_point_mlp(const float *__restrict__ feat, int feat_ld, const int32_t *__restrict__ idx,
                                                      const float *__restrict__ wgt, const float *__restrict__ points, int pt_stride,
                                                      const int32_t *__restrict__ vx_off, int n, PmParams prm, float *__restrict__ out, int out_ld,
                                                      long long *__restrict__ labels) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 31, kk = lane >> 5;
  // ---- the layers' weights [cin][PM_WS] and (scale | shift) vectors [128] -> LDS, once per workgroup
  for (int l = 0; l < prm.num_layers; ++l) {
    const PmLayer &L = prm.layer[l];
    float *Ws = smem + L.w_off, *vec = smem + L.v_off;
    const int cpad = (L.cout + 31) & ~31, sh = cpad == 32 ? 5 : 6;  // 32 or 64 columns: no division, eight loads in flight per trip
    const int total = L.cin * cpad;
    for (int t0 = tid; t0 < total; t0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256, tc = t < total ? t : 0, k = tc >> sh, c = tc & (cpad - 1);
        v[u] = L.w[(size_t)k * L.cout + (c < L.cout ? c : 0)];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = t0 + u * 256, k = t >> sh, c = t & (cpad - 1);
        if (t < total) Ws[k * PM_WS + c] = c < L.cout ? v[u] : 0.0f;
      }
    }
    for (int c = tid; c < 64; c += 256) {
      vec[c] = (c < L.cout && L.scale) ? L.scale[c] : 1.0f;
      vec[64 + c] = (c < L.cout && L.shift) ? L.shift[c] : 0.0f;
    }
  }
  __syncthreads();
  const int ntiles = (n + 127) >> 7;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int p = tile * 128 + wave * 32 + col;  // both lane halves of a column hold the same point
    const bool live = p < n;
    const int pc = live ? p : n - 1;
    pm_f32x16 x[2], y[2];
    // ---- input: the interpolated row (or the row itself) in the C layout
    {
      const int nb = prm.c_in >> 5;
      if (idx) {
        const int f = (int)points[(size_t)pc * pt_stride];
        const int v0 = vx_off[f], cnt = vx_off[f + 1] - v0;
        const float w0 = wgt[(size_t)pc * 3], w1 = wgt[(size_t)pc * 3 + 1], w2 = wgt[(size_t)pc * 3 + 2];
        const float *ra = feat + (size_t)(v0 + idx[(size_t)pc * 3]) * feat_ld, *rb = feat + (size_t)(v0 + idx[(size_t)pc * 3 + 1]) * feat_ld,
                    *rc = feat + (size_t)(v0 + idx[(size_t)pc * 3 + 2]) * feat_ld;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (b < nb && cnt > 0) {  // the arithmetic of k_interp_rows
              const int c0 = 32 * b + 8 * i + 4 * kk;
              const float4 a = *(const float4 *)(ra + c0), bb = *(const float4 *)(rb + c0), c = *(const float4 *)(rc + c0);
              o.x = fmaf(w2, c.x, fmaf(w1, bb.x, w0 * a.x)); o.y = fmaf(w2, c.y, fmaf(w1, bb.y, w0 * a.y));
              o.z = fmaf(w2, c.z, fmaf(w1, bb.z, w0 * a.z)); o.w = fmaf(w2, c.w, fmaf(w1, bb.w, w0 * a.w));
            }
            x[b][4 * i + 0] = o.x; x[b][4 * i + 1] = o.y; x[b][4 * i + 2] = o.z; x[b][4 * i + 3] = o.w;
          }
      } else {
        const float *row = feat + (size_t)pc * feat_ld;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 o = b < nb ? *(const float4 *)(row + 32 * b + 8 * i + 4 * kk) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            x[b][4 * i + 0] = o.x; x[b][4 * i + 1] = o.y; x[b][4 * i + 2] = o.z; x[b][4 * i + 3] = o.w;
          }
      }
    }
    // ---- the layers
    for (int l = 0; l < prm.num_layers; ++l) {
      const PmLayer &L = prm.layer[l];
      const float *Ws = smem + L.w_off, *vec = smem + L.v_off;
      const int ib = L.cin >> 5, ob = (L.cout + 31) >> 5;  // wave-uniform: four straight-line instances
      if (ib == 1 && ob == 1) { pm_gemm<1, 1>(x, Ws, y, col, kk); pm_epilogue<1>(y, vec, L.relu, kk); }
      else if (ib == 1) { pm_gemm<1, 2>(x, Ws, y, col, kk); pm_epilogue<2>(y, vec, L.relu, kk); }
      else if (ob == 1) { pm_gemm<2, 1>(x, Ws, y, col, kk); pm_epilogue<1>(y, vec, L.relu, kk); }
      else { pm_gemm<2, 2>(x, Ws, y, col, kk); pm_epilogue<2>(y, vec, L.relu, kk); }
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[0][r] = y[0][r]; x[1][r] = ob > 1 ? y[1][r] : 0.0f; }
    }
    // ---- logits out, argmax
    const int cout = prm.layer[prm.num_layers - 1].cout;
    float bv = 0.0f;
    int bi = -1;
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 32 * b + 8 * i + 4 * kk + j;
          if (c < cout) {
            const float v = x[b][4 * i + j];
            if (live) out[(size_t)p * out_ld + c] = v;
            if (bi < 0 || v > bv || (v != v && bv == bv)) { bv = v; bi = c; }  // ascending c per lane: the first maximum stays
          }
        }
    if (labels) {
      const float ov = __shfl_xor(bv, 32);
      const int oi = __shfl_xor(bi, 32);
      // the other half's candidate: it wins when it is larger, or equal with a lower class index, or a NaN met before ours
      const bool onan = ov != ov, mnan = bv != bv;
      bool take = oi >= 0 && (bi < 0 || (onan && (!mnan || oi < bi)) || (!mnan && !onan && (ov > bv || (ov == bv && oi < bi))));
      if (take) { bv = ov; bi = oi; }
      if (live && kk == 0) labels[p] = (long long)bi;
    }
  }
}

static size_t pm_layout(PmParams &prm) {
  int off = 0;
  for (int l = 0; l < prm.num_layers; ++l) {
    prm.layer[l].w_off = off;
    off += prm.layer[l].cin * PM_WS;
    prm.layer[l].v_off = off;
    off += 128;
  }
  return (size_t)off * sizeof(float);
}

extern "C" int ls3d_point_mlp(const float *feat, int feat_ld, int c_in, const int32_t *idx, const float *weight, const float *points, int pt_stride,
                              const int32_t *vx_off, int n, int num_layers, const ls3d_point_mlp_layer_t *layers_host, float *out, int out_ld,
                              int64_t *labels, ls3d_stream_t stream) {
  if (!feat || !out || !layers_host || n < 0 || num_layers < 1) return LS3D_ERR_ARG;
  if (idx && (!weight || !points || !vx_off || pt_stride < 1)) return LS3D_ERR_ARG;
  if (num_layers > PM_MAX_LAYERS || (c_in != 32 && c_in != 64)) return LS3D_ERR_UNSUPPORTED;
  if ((feat_ld % 4) || feat_ld < c_in || ((uintptr_t)feat & 15)) return LS3D_ERR_ARG;
  PmParams prm;
  prm.num_layers = num_layers;
  prm.c_in = c_in;
  int c = c_in;
  for (int l = 0; l < num_layers; ++l) {
    const ls3d_point_mlp_layer_t &s = layers_host[l];
    if (!s.w || s.cin != c || s.cout < 1) return LS3D_ERR_ARG;
    const bool last = l + 1 == num_layers;
    if (s.cout > 64 || (!last && s.cout != 32 && s.cout != 64)) return LS3D_ERR_UNSUPPORTED;  // hidden widths are whole 32-channel blocks
    prm.layer[l] = PmLayer{s.w, s.scale, s.shift, s.cin, s.cout, s.relu, 0, 0};
    c = s.cout;
  }
  if (out_ld < c) return LS3D_ERR_ARG;
  const size_t lds = pm_layout(prm);
  if (lds > 80 * 1024) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) return LS3D_OK;
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_point_mlp, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  const int ntiles = (n + 127) / 128;
  const int grid = ntiles < 512 ? ntiles : 512;  // two workgroups per CU, each stages the weights once and walks its tiles
  hipLaunchKernelGGL(k_point_mlp, dim3(grid), dim3(256), lds, (hipStream_t)stream, feat, feat_ld, idx, weight, points, pt_stride, vx_off, n, prm, out,
                     out_ld, (long long *)labels);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
