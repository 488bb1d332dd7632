// tokenattn.hip — the SF-Phase decoder's point -> class-token attention under autograd (reference:
// det3d/models/point_heads/context_module.py:222-257, MultiheadAttention of 10^5 - 10^6 point queries over the frame's L = 2 x classes
// semantic-embedding tokens, H heads of hd = 24 channels).
//
//   forward   att = softmax(scale q_h K_h), out_h = att V_h                      per point and head
//   backward  d att = d out_h V_h^T, d s = scale att (d att - <d att, att>), d q_h = d s K_h^T      per point and head;
//             d K_h = q_h^T d s and d V_h = d out_h^T att reduce over the points: the kernel writes d s and att as [n, H L] rows and
//             the caller takes them with the tall-skinny weight-gradient kernel it already has (ls3d_spconv_wgrad on identity pairs).
//
// torch runs this as batched GEMMs of [n, 24] x [24, L] per head plus six elementwise / reduction passes over [n, H, L] tensors that it
// also keeps from the forward (132 MB per frame and layer on a Waymo frame): 1.2 ms per frame and layer, 14 ms of a training step.  Here
// one thread owns one (point, head): its 24 query channels, L scores and 24 outputs live in registers, K_h and V_h (2 x 24 x L floats,
// the same for every thread of the workgroup: blockIdx.y is the head) are read through the scalar cache as SGPR operands of the FMAs, and
// nothing of the forward is kept - the backward recomputes the probabilities from q.  Exact f32, exp via expf.
#include "common.h"

constexpr int TA_HD = 24;  // channels per head (d_model 96 / 4 heads)

template <int L>
__device__ __forceinline__ void ta_probabilities(const float (&q)[TA_HD], const float *__restrict__ K, float scale, float (&p)[L]) {
#pragma unroll
  for (int l = 0; l < L; ++l) p[l] = 0.0f;
#pragma unroll
  for (int d = 0; d < TA_HD; ++d)
#pragma unroll
    for (int l = 0; l < L; ++l) p[l] = fmaf(q[d], K[d * L + l], p[l]);
  float m = -3.0e38f;
#pragma unroll
  for (int l = 0; l < L; ++l) { p[l] *= scale; m = fmaxf(m, p[l]); }
  float den = 0.0f;
#pragma unroll
  for (int l = 0; l < L; ++l) { p[l] = expf(p[l] - m); den += p[l]; }
  const float inv = 1.0f / den;
#pragma unroll
  for (int l = 0; l < L; ++l) p[l] *= inv;
}

__device__ __forceinline__ void ta_load_row(const float *__restrict__ src, float (&dst)[TA_HD]) {
#pragma unroll
  for (int j = 0; j < TA_HD / 4; ++j) {
    const float4 t = ((const float4 *)src)[j];
    dst[4 * j] = t.x; dst[4 * j + 1] = t.y; dst[4 * j + 2] = t.z; dst[4 * j + 3] = t.w;
  }
}

__device__ __forceinline__ void ta_store_row(float *__restrict__ dst, const float (&src)[TA_HD]) {
#pragma unroll
  for (int j = 0; j < TA_HD / 4; ++j) ((float4 *)dst)[j] = make_float4(src[4 * j], src[4 * j + 1], src[4 * j + 2], src[4 * j + 3]);
}

// grid (point blocks, heads); q / out: [n, H * 24]; k, v: [H, 24, L]
template <int L>
__global__ __launch_bounds__(256) void k_token_attn_fwd(const float *__restrict__ q, int n, int heads, const float *__restrict__ k, const float *__restrict__ v,
                                                        float scale, float *__restrict__ out) {
  const int h = blockIdx.y;
  const float *__restrict__ K = k + (size_t)h * TA_HD * L, *__restrict__ V = v + (size_t)h * TA_HD * L;
  const int ld = heads * TA_HD;
  for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
    const int pt = p0 + threadIdx.x;
    if (pt >= n) continue;
    float qr[TA_HD], pr[L], o[TA_HD];
    ta_load_row(q + (size_t)pt * ld + h * TA_HD, qr);
    ta_probabilities<L>(qr, K, scale, pr);
#pragma unroll
    for (int d = 0; d < TA_HD; ++d) {
      float s = 0.0f;
#pragma unroll
      for (int l = 0; l < L; ++l) s = fmaf(pr[l], V[d * L + l], s);
      o[d] = s;
    }
    ta_store_row(out + (size_t)pt * ld + h * TA_HD, o);
  }
}

// dq: [n, H * 24]; ds, att: [n, H * L] (row p, columns h L .. h L + L of the thread's head).  The L values of a thread go through a
// per-wave LDS tile so that the stores run along the rows (a thread writing its own 4 L bytes would touch 64 lines per store instruction).
template <int L>
__global__ __launch_bounds__(256) void k_token_attn_bwd(const float *__restrict__ q, const float *__restrict__ dout, int n, int heads,
                                                        const float *__restrict__ k, const float *__restrict__ v, float scale, float *__restrict__ dq,
                                                        float *__restrict__ ds_out, float *__restrict__ att_out) {
  static_assert(L % 2 == 0, "rows of L floats are stored as float2");
  __shared__ float s_tile[4][64][L + 1];
  const int h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *__restrict__ K = k + (size_t)h * TA_HD * L, *__restrict__ V = v + (size_t)h * TA_HD * L;
  const int ld = heads * TA_HD, ldl = heads * L;
  for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
    const int w0 = p0 + wave * 64, pt = w0 + lane;
    const bool live = pt < n;
    const size_t row = (size_t)(live ? pt : n - 1);
    float qr[TA_HD], g[TA_HD], pr[L], da[L];
    ta_load_row(q + row * ld + h * TA_HD, qr);
    ta_load_row(dout + row * ld + h * TA_HD, g);
    ta_probabilities<L>(qr, K, scale, pr);
#pragma unroll
    for (int l = 0; l < L; ++l) da[l] = 0.0f;
#pragma unroll
    for (int d = 0; d < TA_HD; ++d)
#pragma unroll
      for (int l = 0; l < L; ++l) da[l] = fmaf(g[d], V[d * L + l], da[l]);
    float dot = 0.0f;
#pragma unroll
    for (int l = 0; l < L; ++l) dot = fmaf(da[l], pr[l], dot);
#pragma unroll
    for (int l = 0; l < L; ++l) da[l] = pr[l] * (da[l] - dot) * scale;  // d s
    {
      float o[TA_HD];
#pragma unroll
      for (int d = 0; d < TA_HD; ++d) {
        float s = 0.0f;
#pragma unroll
        for (int l = 0; l < L; ++l) s = fmaf(da[l], K[d * L + l], s);
        o[d] = s;
      }
      if (live) ta_store_row(dq + (size_t)pt * ld + h * TA_HD, o);
    }
    // the wave's 64 x L tiles of att, then d s: through LDS, stored as float2 along the rows (every thread of the workgroup runs the same trips)
    const int nrows = min(64, n - w0);
#pragma unroll
    for (int which = 0; which < 2; ++which) {
#pragma unroll
      for (int l = 0; l < L; ++l) s_tile[wave][lane][l] = which ? da[l] : pr[l];
      __syncthreads();
      float *__restrict__ dst = which ? ds_out : att_out;
      for (int i = lane; i < nrows * (L / 2); i += 64) {
        const int r = i / (L / 2), c2 = i % (L / 2);
        *(float2 *)(dst + (size_t)(w0 + r) * ldl + h * L + 2 * c2) = make_float2(s_tile[wave][r][2 * c2], s_tile[wave][r][2 * c2 + 1]);
      }
      __syncthreads();
    }
  }
}

static inline bool ta_shape_ok(int heads, int hd, int L) { return hd == TA_HD && heads >= 1 && heads <= 64 && (L == 34 || L == 38 || L == 40 || L == 46); }

#define TA_DISPATCH(L_, CALL)                    \
  switch (L_) {                                  \
    case 34: { constexpr int LL = 34; CALL; break; } \
    case 38: { constexpr int LL = 38; CALL; break; } \
    case 40: { constexpr int LL = 40; CALL; break; } \
    default: { constexpr int LL = 46; CALL; break; } \
  }

extern "C" int ls3d_token_attention_forward(const float *q, int n, int heads, int hd, const float *k, const float *v, int L, float scale, float *out,
                                            ls3d_stream_t stream_) {
  if (n < 0 || heads < 1 || hd < 1 || L < 1) return LS3D_ERR_ARG;
  if (!ta_shape_ok(heads, hd, L)) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) return LS3D_OK;
  if (!q || !k || !v || !out || ((uintptr_t)q & 15) || ((uintptr_t)out & 15)) return LS3D_ERR_ARG;
  const dim3 grid((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), (unsigned)heads);
  TA_DISPATCH(L, hipLaunchKernelGGL((k_token_attn_fwd<LL>), grid, dim3(256), 0, (hipStream_t)stream_, q, n, heads, k, v, scale, out))
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_token_attention_backward(const float *q, const float *dout, int n, int heads, int hd, const float *k, const float *v, int L, float scale,
                                             float *dq, float *ds, float *att, ls3d_stream_t stream_) {
  if (n < 0 || heads < 1 || hd < 1 || L < 1) return LS3D_ERR_ARG;
  if (!ta_shape_ok(heads, hd, L)) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) return LS3D_OK;
  if (!q || !dout || !k || !v || !dq || !ds || !att || ((uintptr_t)q & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 15) || ((uintptr_t)ds & 7) ||
      ((uintptr_t)att & 7))
    return LS3D_ERR_ARG;
  const dim3 grid((unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024), (unsigned)heads);
  TA_DISPATCH(L, hipLaunchKernelGGL((k_token_attn_bwd<LL>), grid, dim3(256), 0, (hipStream_t)stream_, q, dout, n, heads, k, v, scale, dq, ds, att))
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
