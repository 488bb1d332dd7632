// tokenattn.hip — the SF-Phase decoder's point -> class-token attention under autograd (reference:
// det3d/models/point_heads/context_module.py:222-257, MultiheadAttention of 10^5 - 10^6 point queries over the frame's L = 2 x classes
// semantic-embedding tokens, H heads of hd = 24 channels).
//
//   forward   att = softmax(scale q_h K_h), out_h = att V_h                      per point and head
//   backward  d att = d out_h V_h^T, d s = scale att (d att - <d att, att>), d q_h = d s K_h^T      per point and head;
//             d K_h = q_h^T d s and d V_h = d out_h^T att reduce over the points: on the matrix pipe inside the same kernel (below).

//
// torch runs this as batched GEMMs of [n, 24] x [24, L] per head plus six elementwise / reduction passes over [n, H, L] tensors that it
// also keeps from the forward (132 MB per frame and layer on a Waymo frame): 1.2 ms per frame and layer, 14 ms of a training step.  Here
// one thread owns one (point, head): its 24 query channels, L scores and 24 outputs live in registers, K_h and V_h (2 x 24 x L floats,
// the same for every thread of the workgroup: blockIdx.y is the head) are read through the scalar cache as SGPR operands of the FMAs, and
// nothing of the forward is kept - the backward recomputes the probabilities from q.  Exact f32, exp via expf.
#include "common.h"

constexpr int TA_HD = 24;  // channels per head (d_model 96 / 4 heads)

// K_h / V_h are the same in every trip of a kernel's point loop, and hipcc hoists all 2 x 24 x L scalar loads out of it - 2208 SGPR values that it then
// parks in VGPR lanes and scratch and fetches back with one v_readlane per FMA.  Offsetting the base pointers by a zero that comes out of an empty asm statement per trip
// (an opaque zero added to them) keeps the loads where they are used: s_load_dwordx16 from the scalar cache straight into the FMAs' SGPR operands.
#ifdef HIPSIM
#define TA_OPAQUE_KV(K0_, V0_) const float *__restrict__ K = K0_, *__restrict__ V = V0_;
#else
#define TA_OPAQUE_KV(K0_, V0_)                \
  int zero_ = 0;                              \
  asm volatile("" : "+s"(zero_));             \
  const float *__restrict__ K = (K0_) + zero_, *__restrict__ V = (V0_) + zero_;
#endif

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
  const float *__restrict__ K0 = k + (size_t)h * TA_HD * L, *__restrict__ V0 = v + (size_t)h * TA_HD * L;
  const int ld = heads * TA_HD;
  for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
    TA_OPAQUE_KV(K0, V0)
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

// Backward.  dq per (point, head) as above; the token-side gradients d K_h[d][l] = sum_p q[p][d] ds[p][l] and d V_h[d][l] = sum_p dout[p][d] att[p][l]
// reduce over the points: the wave's 64 points' rows go through LDS (they are held point-per-lane, the MFMA wants them point-per-k-step) and
// every PAIR of points is one v_mfma_f32_32x32x2_f32 step per 32 tokens: A[d][kk] = q[p_kk][d] (rows >= 24 zero), B[kk][l] = ds[p_kk][l]
// (tokens >= L zero) - 2 x 2 accumulator blocks of 32 x 32 per wave that live across the workgroup's whole share of the points.  At the end
// the four waves' blocks are added in wave order and written as partial[block][head][2][24][L]; k_token_attn_reduce adds the blocks in order:
// deterministic, no atomics, and neither ds nor att ever reaches memory (2 x 132 MB per frame and layer on a Waymo frame).
typedef float ta_f32x16 __attribute__((ext_vector_type(16)));
constexpr int TA_QLD = TA_HD + 1;  // LDS row stride of the staged q / dout rows

template <int L>
struct TaSmem {
  float tile[4][64 + 1][L + 1];  // a wave's probabilities / d s rows (+ a spare row: lanes of masked columns read past the last one)
  float rows[4][64 + 2][TA_QLD]; // a wave's q / dout rows
};

template <int L>
constexpr int ta_lds_bytes() { return (int)(sizeof(TaSmem<L>) > 4 * 2 * 32 * 64 * sizeof(float) ? sizeof(TaSmem<L>) : 4 * 2 * 32 * 64 * sizeof(float)); }

template <int L>
__device__ __forceinline__ void ta_accumulate(const float (*tile)[L + 1], const float (*rows)[TA_QLD], int lane, ta_f32x16 (&acc)[2]) {
  const int i = lane & 31, kk = lane >> 5;
#pragma unroll 4
  for (int j = 0; j < 32; ++j) {
    const int p = 2 * j + kk;
    const float a = i < TA_HD ? rows[p][i] : 0.0f;
    const float b0 = tile[p][i], b1 = (32 + i) < L ? tile[p][32 + i] : 0.0f;
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc[1], 0, 0, 0);
  }
}

template <int L>
__global__ __launch_bounds__(256, 2) void k_token_attn_bwd(const float *__restrict__ q, const float *__restrict__ dout, int n, int heads,
                                                        const float *__restrict__ k, const float *__restrict__ v, float scale, float *__restrict__ dq,
                                                        float *__restrict__ partial) {
  static_assert(L > 32 && L <= 64, "two 32-token blocks");
  HIP_DYNAMIC_SHARED(char, smem_raw)
  TaSmem<L> &sm = *(TaSmem<L> *)smem_raw;
  const int h = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float *__restrict__ K0 = k + (size_t)h * TA_HD * L, *__restrict__ V0 = v + (size_t)h * TA_HD * L;
  const int ld = heads * TA_HD;
  ta_f32x16 aK[2], aV[2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) { aK[b][r] = 0.0f; aV[b][r] = 0.0f; }
  for (int p0 = blockIdx.x * 256; p0 < n; p0 += gridDim.x * 256) {
    TA_OPAQUE_KV(K0, V0)
    const int w0 = p0 + wave * 64, pt = w0 + lane;
    const bool live = pt < n;
    const size_t row = (size_t)(live ? pt : n - 1);
    // (register budget: two waves per SIMD beside the 64 accumulator registers - dout and the probabilities are re-read from the thread's own
    // row of the LDS tiles once they are staged, so at most q and one L-vector are live at a time)
    float qr[TA_HD], da[L];
    ta_load_row(q + row * ld + h * TA_HD, qr);
    {
      float g[TA_HD], pr[L];
      ta_load_row(dout + row * ld + h * TA_HD, g);
      ta_probabilities<L>(qr, K, scale, pr);
      // d V += dout^T att: the wave's rows through its LDS tiles (points past the end contribute zero rows).  Every thread of the workgroup runs
      // the same trips, so the block barriers are uniform.
#pragma unroll
      for (int l = 0; l < L; ++l) sm.tile[wave][lane][l] = live ? pr[l] : 0.0f;
#pragma unroll
      for (int d = 0; d < TA_HD; ++d) sm.rows[wave][lane][d] = live ? g[d] : 0.0f;  // a dead lane re-read row n - 1: 0 x Inf / NaN of that row must not reach d V
    }
    __syncthreads();
    ta_accumulate<L>(sm.tile[wave], sm.rows[wave], lane, aV);
#pragma unroll
    for (int l = 0; l < L; ++l) da[l] = 0.0f;
#pragma unroll
    for (int d = 0; d < TA_HD; ++d) {
      const float gd = sm.rows[wave][lane][d];
#pragma unroll
      for (int l = 0; l < L; ++l) da[l] = fmaf(gd, V[d * L + l], da[l]);
    }
    {
      float dot = 0.0f;
#pragma unroll
      for (int l = 0; l < L; ++l) dot = fmaf(da[l], sm.tile[wave][lane][l], dot);
#pragma unroll
      for (int l = 0; l < L; ++l) da[l] = sm.tile[wave][lane][l] * (da[l] - dot) * scale;  // d s (a dead point's probabilities are staged as zeros: d s = 0)
    }
    __syncthreads();  // every wave is done with the tiles of d V
    // d K += q^T d s
#pragma unroll
    for (int l = 0; l < L; ++l) sm.tile[wave][lane][l] = live ? da[l] : 0.0f;
#pragma unroll
    for (int d = 0; d < TA_HD; ++d) sm.rows[wave][lane][d] = live ? qr[d] : 0.0f;
    __syncthreads();
    ta_accumulate<L>(sm.tile[wave], sm.rows[wave], lane, aK);
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
    __syncthreads();
  }
  // the four waves' accumulator blocks, added in wave order: dump [wave][matrix][d 32][l 64] over the tiles (dead by now; ta_lds_bytes covers it)
  float *dump = (float *)smem_raw;
  {
    const int i = lane & 31, half = lane >> 5;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int d = (r & 3) + 8 * (r >> 2) + 4 * half;  // fragment layout of the 32 x 32 MFMA: row d, column i of block b
          dump[((wave * 2 + m) * 32 + d) * 64 + b * 32 + i] = m ? aV[b][r] : aK[b][r];
        }
  }
  __syncthreads();
  float *__restrict__ dst = partial + ((size_t)blockIdx.x * heads + h) * 2 * TA_HD * L;  // [dK | dV][24][L]
  for (int e = threadIdx.x; e < 2 * TA_HD * L; e += 256) {
    const int m = e / (TA_HD * L), d = (e / L) % TA_HD, l = e % L;
    float s = dump[((0 * 2 + m) * 32 + d) * 64 + l];
#pragma unroll
    for (int w = 1; w < 4; ++w) s += dump[((w * 2 + m) * 32 + d) * 64 + l];
    dst[e] = s;
  }
}

// dk, dv [heads][24][L] = sum over the row blocks' partials [block][head][2][24][L], in block order
__global__ __launch_bounds__(256) void k_token_attn_reduce(const float *__restrict__ partial, int nblocks, int heads, int L, float *__restrict__ dk,
                                                           float *__restrict__ dv) {
  const int per = TA_HD * L, total = heads * 2 * per;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    float s = 0.0f;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * total + e];
    const int h = e / (2 * per), m = (e / per) & 1, r = e % per;
    (m ? dv : dk)[(size_t)h * per + r] = s;
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

static inline int ta_bwd_blocks(int n) { const int b = (n + 255) / 256; return b < 1 ? 1 : (b < 256 ? b : 256); }

extern "C" size_t ls3d_token_attention_workspace_bytes(int n, int heads, int L) {
  if (n < 0 || heads < 1 || L < 1) return 0;
  return (size_t)ta_bwd_blocks(n) * heads * 2 * TA_HD * L * sizeof(float) + 256;
}

extern "C" int ls3d_token_attention_backward(const float *q, const float *dout, int n, int heads, int hd, const float *k, const float *v, int L, float scale,
                                             float *dq, float *dk, float *dv, void *workspace, size_t workspace_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || heads < 1 || hd < 1 || L < 1) return LS3D_ERR_ARG;
  if (!ta_shape_ok(heads, hd, L)) return LS3D_ERR_UNSUPPORTED;
  if (!dk || !dv) return LS3D_ERR_ARG;
  if (n == 0) {
    const size_t bytes = (size_t)heads * TA_HD * L * sizeof(float);
    return hipMemsetAsync(dk, 0, bytes, stream) == hipSuccess && hipMemsetAsync(dv, 0, bytes, stream) == hipSuccess ? LS3D_OK : LS3D_ERR_LAUNCH;
  }
  if (!q || !dout || !k || !v || !dq || !workspace || ((uintptr_t)q & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dq & 15) || ((uintptr_t)workspace & 15))
    return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_token_attention_workspace_bytes(n, heads, L)) return LS3D_ERR_WORKSPACE;
  const int nb = ta_bwd_blocks(n);
  static bool attr_set[LS3D_MAX_DEVICES] = {};
  const int slot = ls3d_device_slot();
  if (!attr_set[slot]) {  // the opt-in to > 64 KB of dynamic LDS is per device
    bool ok = true;
    TA_DISPATCH(46, ok = ok && hipFuncSetAttribute((const void *)k_token_attn_bwd<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes<LL>()) == hipSuccess)
    TA_DISPATCH(40, ok = ok && hipFuncSetAttribute((const void *)k_token_attn_bwd<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes<LL>()) == hipSuccess)
    TA_DISPATCH(38, ok = ok && hipFuncSetAttribute((const void *)k_token_attn_bwd<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes<LL>()) == hipSuccess)
    TA_DISPATCH(34, ok = ok && hipFuncSetAttribute((const void *)k_token_attn_bwd<LL>, hipFuncAttributeMaxDynamicSharedMemorySize, ta_lds_bytes<LL>()) == hipSuccess)
    if (!ok) return LS3D_ERR_LAUNCH;
    attr_set[slot] = true;
  }
  TA_DISPATCH(L, hipLaunchKernelGGL((k_token_attn_bwd<LL>), dim3((unsigned)nb, (unsigned)heads), dim3(256), ta_lds_bytes<LL>(), stream, q, dout, n, heads, k, v,
                                    scale, dq, (float *)workspace))
  hipLaunchKernelGGL(k_token_attn_reduce, dim3((unsigned)((heads * 2 * TA_HD * L + 255) / 256)), dim3(256), 0, stream, (const float *)workspace, nb, heads, L, dk, dv);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
