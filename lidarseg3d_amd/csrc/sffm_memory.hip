// sffm_memory.hip — the class-embedding ("memory") side of the SF-Phase decoder for all layers in ONE launch.
//
// Reference: det3d/models/point_heads/context_module.py:147-171 (TransformerDecoder.forward), :211-250
// (TransformerDecoderLayer.forward_post: self-attention of the memory tokens + residual + norm1, then the cross attention's k_proj /
// v_proj - Conv1d(k = 1) over the tokens, :320-338).  The memory of a frame is L = 2 * num_class tokens of 96 floats (34 x 96 on
// nuScenes) and never sees the points, so ls3d_sffm_decoder takes the k / v of every layer as an input.  Layer by layer that side is
// five launches per layer on a 34-row matrix (qkv projection, attention core, out-projection + residual + LayerNorm, k, v) plus the copies
// that put k / v into [layer][B][E][L] - ~40 launches of 7 - 30 us each, serial, in front of the decoder (0.35 ms of the 9 ms MSeg3D frame).
// Here: one workgroup per frame keeps the tokens in LDS through all layers and writes k / v in the decoder's layout.  The contractions
// are plain f32 fma chains in ascending k (2 MFLOP per layer and frame: nothing for the matrix pipe to win); a thread owns an output
// column and 8 rows at a time, so a weight is read once per 8 rows (coalesced along the columns, from L2) and a token value is an LDS
// broadcast.
#include "common.h"

constexpr int SM_E = 96, SM_H = 4, SM_HD = 24, SM_LMAX = 64, SM_MAX_LAYERS = 8;
constexpr int SM_MS = SM_E + 1;       // row stride of the token tiles (floats)
constexpr int SM_QS = 3 * SM_E + 1;   // row stride of the q | k | v tile

struct SmLayer {
  const float *wqkv_t, *bqkv, *wo_t, *bo, *n1g, *n1b, *wk_t, *bk, *wv_t, *bv;
  float n1eps;
};
struct SmParams {
  int num_layers;
  SmLayer layer[SM_MAX_LAYERS];
};

// out[r][c] (+)= bias[c] + sum_e A[r][e] * Wt[e][c] for the 8 rows r0 .. r0 + 7 of one column c
__device__ __forceinline__ void sm_column8(const float *A, int lda, int r0, const float *__restrict__ Wt, int ldw, int c, float (&acc)[8]) {
  for (int e = 0; e < SM_E; ++e) {
    const float w = Wt[(size_t)e * ldw + c];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fmaf(A[(r0 + i) * lda + e], w, acc[i]);
  }
}

__global__ __launch_bounds__(256) void k_sffm_memory(const float *__restrict__ mem, int batch, int L, SmParams prm, float *__restrict__ kv,
                                                    float *__restrict__ mem_out) {
  HIP_DYNAMIC_SHARED(float, smem)
  float *M = smem;                       // [SM_LMAX][SM_MS] tokens (rows >= L stay zero)
  float *Q = M + SM_LMAX * SM_MS;        // [SM_LMAX][SM_QS] q | k | v of the self-attention
  float *A = Q + SM_LMAX * SM_QS;        // [SM_LMAX][SM_MS] attention output
  const int tid = threadIdx.x, b = blockIdx.x;
  const int Lp = (L + 7) & ~7;           // rows in blocks of 8
  for (int i = tid; i < SM_LMAX * SM_MS; i += 256) {
    const int r = i / SM_MS, c = i - r * SM_MS;
    M[i] = (r < L && c < SM_E) ? mem[((size_t)b * L + r) * SM_E + c] : 0.0f;
    A[i] = 0.0f;
  }
  __syncthreads();
  for (int l = 0; l < prm.num_layers; ++l) {
    const SmLayer &P = prm.layer[l];
    // ---- q | k | v of the self-attention: 288 columns
    for (int c = tid; c < 3 * SM_E; c += 256) {
      const float bc = P.bqkv[c];
      for (int r0 = 0; r0 < Lp; r0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = bc;
        sm_column8(M, SM_MS, r0, P.wqkv_t, 3 * SM_E, c, acc);
#pragma unroll
        for (int i = 0; i < 8; ++i) Q[(r0 + i) * SM_QS + c] = acc[i];
      }
    }
    __syncthreads();
    // ---- softmax(q k^T / sqrt(hd)) v: one thread per (token, head), the arithmetic of k_mha_core (vfe.hip)
    if (tid < L * SM_H) {
      const int r = tid / SM_H, h = tid - r * SM_H;
      const float scale = 1.0f / sqrtf((float)SM_HD);
      const float *qp = Q + r * SM_QS + h * SM_HD;
      float q[SM_HD], o[SM_HD];
#pragma unroll
      for (int d = 0; d < SM_HD; ++d) { q[d] = qp[d] * scale; o[d] = 0.0f; }
      float m = -3.0e38f;
      for (int j = 0; j < L; ++j) {
        const float *kp = Q + j * SM_QS + SM_E + h * SM_HD;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < SM_HD; ++d) s = fmaf(q[d], kp[d], s);
        m = fmaxf(m, s);
      }
      float den = 0.0f;
      for (int j = 0; j < L; ++j) {
        const float *kp = Q + j * SM_QS + SM_E + h * SM_HD;
        const float *vp = kp + SM_E;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < SM_HD; ++d) s = fmaf(q[d], kp[d], s);
        const float p = expf(s - m);
        den += p;
#pragma unroll
        for (int d = 0; d < SM_HD; ++d) o[d] = fmaf(p, vp[d], o[d]);
      }
      const float inv = 1.0f / den;
#pragma unroll
      for (int d = 0; d < SM_HD; ++d) A[r * SM_MS + h * SM_HD + d] = o[d] * inv;
    }
    __syncthreads();
    // ---- out-projection + residual, in place: element (row, column) of M is read and written by one thread only
    if (tid < 2 * SM_E) {
      const int c = tid % SM_E, part = tid / SM_E;
      const float bc = P.bo[c];
      for (int r0 = part * 8; r0 < Lp; r0 += 16) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = bc;
        sm_column8(A, SM_MS, r0, P.wo_t, SM_E, c, acc);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r0 + i < L) M[(r0 + i) * SM_MS + c] += acc[i];
      }
    }
    __syncthreads();
    // ---- norm1: four lanes per token, two-pass statistics
    {
      const int r = tid >> 2, part = tid & 3;
      float *row = M + r * SM_MS + part * (SM_E / 4);
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < SM_E / 4; ++d) s += row[d];
      s += __shfl_xor(s, 1);
      s += __shfl_xor(s, 2);
      const float mean = s / (float)SM_E;
      float v = 0.0f;
#pragma unroll
      for (int d = 0; d < SM_E / 4; ++d) { const float x = row[d] - mean; v = fmaf(x, x, v); }
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      const float rstd = 1.0f / sqrtf(v / (float)SM_E + P.n1eps);
      if (r < L) {
#pragma unroll
        for (int d = 0; d < SM_E / 4; ++d) {
          const int c = part * (SM_E / 4) + d;
          row[d] = (row[d] - mean) * rstd * P.n1g[c] + P.n1b[c];
        }
      }
    }
    __syncthreads();
    // ---- k_proj / v_proj of the cross attention -> kv[2 l + {0, 1}][b][c][token]
    if (tid < 2 * SM_E) {
      const int c = tid % SM_E, which = tid / SM_E;
      const float *Wt = which ? P.wv_t : P.wk_t;
      const float bc = (which ? P.bv : P.bk)[c];
      float *dst = kv + (((size_t)(2 * l + which) * batch + b) * SM_E + c) * L;
      for (int r0 = 0; r0 < Lp; r0 += 8) {
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = bc;
        sm_column8(M, SM_MS, r0, Wt, SM_E, c, acc);
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (r0 + i < L) dst[r0 + i] = acc[i];
      }
    }
    // (the next layer's first phase only reads M and writes Q; its barrier orders it against this one's reads)
  }
  if (mem_out) {
    __syncthreads();
    for (int i = tid; i < L * SM_E; i += 256) mem_out[(size_t)b * L * SM_E + i] = M[(i / SM_E) * SM_MS + i % SM_E];
  }
}

extern "C" int ls3d_sffm_memory(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers,
                                float *kv, float *mem_out, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!mem || !kv || batch < 0 || L < 1 || num_layers < 0 || (num_layers > 0 && !layers)) return LS3D_ERR_ARG;
  if (embed != SM_E || heads != SM_H || L > SM_LMAX || num_layers > SM_MAX_LAYERS) return LS3D_ERR_UNSUPPORTED;  // the caller composes it layer by layer
  if (batch == 0 || num_layers == 0) return LS3D_OK;
  SmParams prm;
  prm.num_layers = num_layers;
  for (int l = 0; l < num_layers; ++l) {
    const ls3d_sffm_memory_layer_t &s = layers[l];
    if (!s.wqkv_t || !s.bqkv || !s.wo_t || !s.bo || !s.n1_gamma || !s.n1_beta || !s.wk_t || !s.bk || !s.wv_t || !s.bv) return LS3D_ERR_ARG;
    prm.layer[l] = SmLayer{s.wqkv_t, s.bqkv, s.wo_t, s.bo, s.n1_gamma, s.n1_beta, s.wk_t, s.bk, s.wv_t, s.bv, s.n1_eps};
  }
  const int lds = (2 * SM_LMAX * SM_MS + SM_LMAX * SM_QS) * (int)sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_sffm_memory, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_sffm_memory, dim3((unsigned)batch), dim3(256), lds, stream, mem, batch, L, prm, kv, mem_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
