// dynreader.hip - what the "other backbones" row needs beyond the sparse convolutions:
//   * ls3d_act_affine          the activation -> BatchNorm(eval) (-> add, -> multiply) tails of the Cylinder3D blocks
//                              (det3d/models/backbones/scn_unet_cylinder3d.py:52-252: conv, LeakyReLU, BatchNorm1d in THAT order, which
//                              the gather-GEMM's BN -> ReLU epilogue cannot express; ReconBlock's sigmoid gates)
//   * ls3d_cyl_voxelize        cart2cylind + voxelize of the PolarNet / Cylinder3D dynamic readers
//                              (det3d/models/readers/voxel_encoder.py:11-18,333-360,563-590)
//   * ls3d_unique_sorted       torch.unique(rows, return_inverse, return_counts, dim=0) on sorted linearised keys (:441,670)
//   * ls3d_dyn_point_features  prepare_input_feature (:362-386,592-616) fused with the leading BatchNorm1d of PPmodel
//   * ls3d_tta_merge           test-time-augmentation merge of the point heads' predict(): mean over the variants of the softmax,
//                              argmax (det3d/models/point_heads/point_seg_batchloss_head.py:190-245)
// All of it is HBM-streaming work of a few bytes per point: one pass, coalesced rows, no atomics.
#include "common.h"

namespace {

__device__ __forceinline__ float dr_act(float v, int kind, float slope) {
  switch (kind) {
    case 1: return fmaxf(v, 0.f);
    case 2: return v >= 0.f ? v : __fmul_rn(v, slope);             // nn.LeakyReLU: x if x >= 0 else slope * x
    case 3: return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v)));       // torch.sigmoid in f32
    default: return v;
  }
}

// y[r, c] = post(pre(x[r, c]) * scale[c] + shift[c]) (+ add[r, c]) (* mul[r, c]); one thread per 4 columns when everything is 16-byte
// aligned (VEC), else per element
template <bool VEC>
__global__ __launch_bounds__(256) void k_act_affine(const float *__restrict__ x, int x_ld, int n, const int32_t *n_dev, int c, int pre, int post, float slope,
                                                    const float *__restrict__ scale, const float *__restrict__ shift, const float *__restrict__ add, int add_ld,
                                                    const float *__restrict__ mul, int mul_ld, float *__restrict__ y, int y_ld) {
  const int N = ls3d_count(n, n_dev);
  constexpr int W = VEC ? 4 : 1;
  const int cg = c / W;
  const long long work = (long long)N * cg;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(t / cg), c0 = (int)(t % cg) * W;
    float v[W], s[W], h[W], a[W], m[W];
    if constexpr (VEC) {
      *(float4 *)v = *(const float4 *)(x + (size_t)r * x_ld + c0);
      if (scale) *(float4 *)s = *(const float4 *)(scale + c0);
      if (shift) *(float4 *)h = *(const float4 *)(shift + c0);
      if (add) *(float4 *)a = *(const float4 *)(add + (size_t)r * add_ld + c0);
      if (mul) *(float4 *)m = *(const float4 *)(mul + (size_t)r * mul_ld + c0);
    } else {
      v[0] = x[(size_t)r * x_ld + c0];
      if (scale) s[0] = scale[c0];
      if (shift) h[0] = shift[c0];
      if (add) a[0] = add[(size_t)r * add_ld + c0];
      if (mul) m[0] = mul[(size_t)r * mul_ld + c0];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float u = dr_act(v[j], pre, slope);
      if (scale) u = __fmul_rn(u, s[j]);
      if (shift) u = __fadd_rn(u, h[j]);
      u = dr_act(u, post, slope);
      if (add) u = __fadd_rn(u, a[j]);
      if (mul) u = __fmul_rn(u, m[j]);
      v[j] = u;
    }
    if constexpr (VEC) *(float4 *)(y + (size_t)r * y_ld + c0) = *(const float4 *)v;
    else y[(size_t)r * y_ld + c0] = v[0];
  }
}

// one thread per point: (rho, phi, z) in f32 as torch evaluates cart2cylind (x * x + y * y, two roundings, sqrt; atan2 evaluated in
// double and rounded once: the correctly rounded f32 value), cell = clamp(int(floor((c - lo) / vs)), 0, grid - 1) with an f32
// subtraction and a true f32 division (voxel_encoder.py:337-345: the clamp comes BEFORE the range test, so no point is ever dropped).
__global__ __launch_bounds__(256) void k_cyl_voxelize(const float *__restrict__ points, int n, int stride, float lo0, float lo1, float lo2, float vs0, float vs1,
                                                      float vs2, int g0, int g1, int g2, int reverse, int collapse, int batch, float *__restrict__ cyl,
                                                      int64_t *__restrict__ vcoors, uint32_t *__restrict__ keys) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float *p = points + (size_t)i * stride;
    const float x = p[1], y = p[2], z = p[3];
    const float rho = __fsqrt_rn(__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y)));
    const float phi = (float)atan2((double)y, (double)x);
    { float *q = cyl + 5 * (size_t)i; q[0] = rho, q[1] = phi, q[2] = z, q[3] = x, q[4] = y; }  // the five columns the readers normalise by their voxel mean
    int c0 = (int)floorf(__fdiv_rn(__fsub_rn(rho, lo0), vs0)), c1 = (int)floorf(__fdiv_rn(__fsub_rn(phi, lo1), vs1)),
        c2 = (int)floorf(__fdiv_rn(__fsub_rn(z, lo2), vs2));
    c0 = min(max(c0, 0), g0 - 1), c1 = min(max(c1, 0), g1 - 1), c2 = min(max(c2, 0), g2 - 1);
    int b = (int)p[0];
    int64_t *o = vcoors + 4 * (size_t)i;
    o[0] = b;
    if (reverse) o[1] = c2, o[2] = c1, o[3] = c0;
    else o[1] = c0, o[2] = c1, o[3] = c2;
    // the key the rows are grouped by, in the column order torch.unique(dim=0) sorts lexicographically; PolarNet (collapse) groups a
    // whole (rho, phi) column: its last column is the constant grid[2] // 2 (voxel_encoder.py:438-440)
    b = min(max(b, 0), batch - 1);
    const unsigned k1 = (unsigned)(reverse ? c2 : c0), k2 = (unsigned)c1, k3 = (unsigned)(collapse ? g2 / 2 : (reverse ? c0 : c2));
    const unsigned d1 = (unsigned)(reverse ? g2 : g0), d2 = (unsigned)g1, d3 = (unsigned)(reverse ? g0 : g2);
    keys[i] = (((unsigned)b * d1 + k1) * d2 + k2) * d3 + k3;
  }
}

__global__ __launch_bounds__(256) void k_us_flags(const uint32_t *__restrict__ skeys, int n, int32_t *__restrict__ flag) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) flag[i] = (i == 0 || skeys[i] != skeys[i - 1]) ? 1 : 0;
}

__global__ __launch_bounds__(256) void k_us_assign(const uint32_t *__restrict__ skeys, const int32_t *__restrict__ perm, const int32_t *__restrict__ flag,
                                                   const int32_t *__restrict__ rank, int n, int d1, int d2, int d3, int64_t *__restrict__ inverse,
                                                   int64_t *__restrict__ rows, int32_t *__restrict__ start) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int id = rank[i] + flag[i] - 1;  // rank = exclusive scan of the head flags
    inverse[perm[i]] = id;
    if (flag[i]) {
      unsigned k = skeys[i];
      int64_t *o = rows + 4 * (size_t)id;
      o[3] = k % (unsigned)d3; k /= (unsigned)d3;
      o[2] = k % (unsigned)d2; k /= (unsigned)d2;
      o[1] = k % (unsigned)d1; o[0] = k / (unsigned)d1;
      start[id] = i;
    }
  }
}

__global__ __launch_bounds__(256) void k_us_counts(const int32_t *__restrict__ start, const int32_t *__restrict__ total, int n, int64_t *__restrict__ counts,
                                                   int32_t *__restrict__ n_unique) {
  const int U = *total;
  for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < U; u += gridDim.x * blockDim.x) counts[u] = (int64_t)((u + 1 < U ? start[u + 1] : n) - start[u]);
  if (blockIdx.x == 0 && threadIdx.x == 0) *n_unique = U;
}

// one thread per point: the row [cyl(3), x, y, extra features, (cyl, x, y) - their voxel mean (5), cyl - voxel centre (3)] of
// prepare_input_feature, then x * scale + shift of PPmodel's leading BatchNorm1d (eval), zero padded to ld columns
__global__ __launch_bounds__(256) void k_dyn_features(const float *__restrict__ points, int n, int stride, int n_extra, const float *__restrict__ cyl,
                                                      const int64_t *__restrict__ vcoors, const int64_t *__restrict__ inverse, const float *__restrict__ mean5,
                                                      float vs0, float vs1, float vs2, float lo0, float lo1, float lo2, const float *__restrict__ scale,
                                                      const float *__restrict__ shift, float *__restrict__ out, int ld) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float *p = points + (size_t)i * stride;
    const float *cy = cyl + 5 * (size_t)i;
    const float *mu = mean5 + 5 * (size_t)inverse[i];
    const int64_t *vc = vcoors + 4 * (size_t)i;
    float *o = out + (size_t)i * ld;
    const float f5[5] = {cy[0], cy[1], cy[2], cy[3], cy[4]};
    int c = 0;
    auto put = [&](float v) {
      if (scale) v = __fmul_rn(v, scale[c]);
      if (shift) v = __fadd_rn(v, shift[c]);
      o[c++] = v;
    };
    for (int j = 0; j < 5; ++j) put(f5[j]);
    for (int j = 0; j < n_extra; ++j) put(p[4 + j]);
    for (int j = 0; j < 5; ++j) put(__fsub_rn(f5[j], mu[j]));
    // get_voxel_centers (core/utils/common_utils.py:74-90) on the three stored cell columns: centre_j = (col[3 - j] + 0.5) * vs_j + lo_j
    put(__fsub_rn(cy[0], __fadd_rn(__fmul_rn(__fadd_rn((float)vc[3], 0.5f), vs0), lo0)));
    put(__fsub_rn(cy[1], __fadd_rn(__fmul_rn(__fadd_rn((float)vc[2], 0.5f), vs1), lo1)));
    put(__fsub_rn(cy[2], __fadd_rn(__fmul_rn(__fadd_rn((float)vc[1], 0.5f), vs2), lo2)));
    for (; c < ld; ++c) o[c] = 0.f;
  }
}

// one thread per point of the merged sample: softmax of its row in each of the k variants (f32, exp(x - max) / sum as torch.softmax), summed in
// variant order, divided by k (torch.mean over the stacked variants), first index of the maximum (torch.argmax)
constexpr int TTA_MAX_CLASSES = 64, TTA_MAX_VARIANTS = 16;
struct TtaRows { int first[TTA_MAX_VARIANTS]; };
__global__ __launch_bounds__(256) void k_tta_merge(const float *__restrict__ logits, int ld, int C, int n, TtaRows rows, int k, float *__restrict__ probs,
                                                   int64_t *__restrict__ labels) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    float acc[TTA_MAX_CLASSES];
#pragma unroll 1
    for (int t = 0; t < k; ++t) {
      const float *x = logits + (size_t)(rows.first[t] + p) * ld;
      float mx = x[0];
      for (int c = 1; c < C; ++c) mx = fmaxf(mx, x[c]);
      float e[TTA_MAX_CLASSES], sum = 0.f;
      for (int c = 0; c < C; ++c) { e[c] = expf(__fsub_rn(x[c], mx)); sum = __fadd_rn(sum, e[c]); }
      for (int c = 0; c < C; ++c) {
        const float q = __fdiv_rn(e[c], sum);
        acc[c] = t == 0 ? q : __fadd_rn(acc[c], q);
      }
    }
    int best = 0;
    for (int c = 0; c < C; ++c) {
      acc[c] = __fdiv_rn(acc[c], (float)k);
      if (acc[c] > acc[best]) best = c;
      if (probs) probs[(size_t)p * C + c] = acc[c];
    }
    labels[p] = best;
  }
}

inline size_t dr_align(size_t v) { return (v + 255) & ~(size_t)255; }
inline bool dr_vec_ok(const void *p, int ld) { return p == nullptr || ((((uintptr_t)p) & 15) == 0 && (ld & 3) == 0); }

}  // namespace

extern "C" int ls3d_act_affine(const float *x, int x_ld, int n, const int32_t *n_dev, int c, int pre_act, int post_act, float slope, const float *scale,
                               const float *shift, const float *add, int add_ld, const float *mul, int mul_ld, float *y, int y_ld, ls3d_stream_t stream_) {
  if (n < 0 || c < 1 || pre_act < 0 || pre_act > 3 || post_act < 0 || post_act > 3) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  if (!x || !y || x_ld < c || y_ld < c || (add && add_ld < c) || (mul && mul_ld < c)) return LS3D_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const bool vec = (c & 3) == 0 && dr_vec_ok(x, x_ld) && dr_vec_ok(y, y_ld) && dr_vec_ok(add, add_ld) && dr_vec_ok(mul, mul_ld) && dr_vec_ok(scale, 0) &&
                   dr_vec_ok(shift, 0);
  if (vec)
    hipLaunchKernelGGL(k_act_affine<true>, ls3d_grid((long long)n * (c / 4)), dim3(256), 0, stream, x, x_ld, n, n_dev, c, pre_act, post_act, slope, scale, shift, add,
                       add_ld, mul, mul_ld, y, y_ld);
  else
    hipLaunchKernelGGL(k_act_affine<false>, ls3d_grid((long long)n * c), dim3(256), 0, stream, x, x_ld, n, n_dev, c, pre_act, post_act, slope, scale, shift, add,
                       add_ld, mul, mul_ld, y, y_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_cyl_voxelize(const float *points, int n, int stride, const ls3d_grid_t *grid_host, int reverse, int collapse_last, int batch, float *cyl,
                                 int64_t *vcoors, uint32_t *keys, ls3d_stream_t stream_) {
  if (n < 0 || !grid_host || stride < 4 || batch < 1) return LS3D_ERR_ARG;
  const ls3d_grid_t &g = *grid_host;
  if (g.grid[0] < 1 || g.grid[1] < 1 || g.grid[2] < 1) return LS3D_ERR_ARG;
  if ((double)batch * g.grid[0] * g.grid[1] * g.grid[2] >= 4294967296.0) return LS3D_ERR_UNSUPPORTED;  // 32-bit row keys
  if (n == 0) return LS3D_OK;
  if (!points || !cyl || !vcoors || !keys) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_cyl_voxelize, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream_, points, n, stride, g.lo[0], g.lo[1], g.lo[2], g.vs[0], g.vs[1], g.vs[2],
                     g.grid[0], g.grid[1], g.grid[2], reverse ? 1 : 0, collapse_last ? 1 : 0, batch, cyl, vcoors, keys);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" size_t ls3d_unique_sorted_workspace_bytes(int n) {
  const size_t m = (size_t)(n > 0 ? n : 1);
  return 3 * dr_align(m * 4) + dr_align(ls3d_scan_tmp_ints((long long)m) * 4) + 256;
}

extern "C" int ls3d_unique_sorted(const uint32_t *keys_sorted, const int32_t *perm, int n, const int32_t dims_host[3], void *workspace,
                                  size_t workspace_bytes, int64_t *inverse, int64_t *unique_rows, int64_t *counts, int32_t *n_unique_dev,
                                  ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || !n_unique_dev || !dims_host || dims_host[0] < 1 || dims_host[1] < 1 || dims_host[2] < 1) return LS3D_ERR_ARG;
  if (n == 0) return hipMemsetAsync(n_unique_dev, 0, 4, stream) == hipSuccess ? LS3D_OK : LS3D_ERR_LAUNCH;
  if (!keys_sorted || !perm || !workspace || !inverse || !unique_rows || !counts) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_unique_sorted_workspace_bytes(n)) return LS3D_ERR_WORKSPACE;
  char *w = (char *)workspace;
  int32_t *flag = (int32_t *)w; w += dr_align((size_t)n * 4);
  int32_t *rank = (int32_t *)w; w += dr_align((size_t)n * 4);
  int32_t *start = (int32_t *)w; w += dr_align((size_t)n * 4);
  int32_t *tmp = (int32_t *)w; w += dr_align(ls3d_scan_tmp_ints(n) * 4);
  int32_t *total = (int32_t *)w;
  const dim3 gp = ls3d_grid(n), blk(256);
  hipLaunchKernelGGL(k_us_flags, gp, blk, 0, stream, keys_sorted, n, flag);
  const int rc = ls3d_exclusive_scan_i32(flag, rank, n, tmp, total, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_us_assign, gp, blk, 0, stream, keys_sorted, perm, (const int32_t *)flag, (const int32_t *)rank, n, dims_host[0], dims_host[1], dims_host[2],
                     inverse, unique_rows, start);
  hipLaunchKernelGGL(k_us_counts, gp, blk, 0, stream, (const int32_t *)start, (const int32_t *)total, n, counts, n_unique_dev);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_dyn_point_features(const float *points, int n, int stride, const float *cyl, const int64_t *vcoors, const int64_t *inverse,
                                       const float *mean5, const ls3d_grid_t *grid_host, const float *scale, const float *shift, float *out, int out_ld,
                                       ls3d_stream_t stream_) {
  if (n < 0 || stride < 4 || !grid_host || out_ld < stride + 9) return LS3D_ERR_ARG;  // 5 + (stride - 4) + 5 + 3 columns
  if (n == 0) return LS3D_OK;
  if (!points || !cyl || !vcoors || !inverse || !mean5 || !out) return LS3D_ERR_ARG;
  const ls3d_grid_t &g = *grid_host;
  hipLaunchKernelGGL(k_dyn_features, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream_, points, n, stride, stride - 4, cyl, vcoors, inverse, mean5, g.vs[0], g.vs[1],
                     g.vs[2], g.lo[0], g.lo[1], g.lo[2], scale, shift, out, out_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_tta_merge(const float *logits, int ld, int num_class, int n, const int32_t *variant_first_row_host, int k, float *probs_out,
                              int64_t *labels_out, ls3d_stream_t stream_) {
  if (n < 0 || k < 1 || k > TTA_MAX_VARIANTS || num_class < 1 || ld < num_class || !variant_first_row_host) return LS3D_ERR_ARG;
  if (num_class > TTA_MAX_CLASSES) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) return LS3D_OK;
  if (!logits || !labels_out) return LS3D_ERR_ARG;
  TtaRows rows;
  for (int t = 0; t < TTA_MAX_VARIANTS; ++t) rows.first[t] = t < k ? variant_first_row_host[t] : 0;
  hipLaunchKernelGGL(k_tta_merge, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream_, logits, ld, num_class, n, rows, k, probs_out, labels_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
