// sffm_memory.hip — the class-embedding ("memory") side of the SF-Phase decoder for all layers in ONE launch.
//
// Reference: det3d/models/point_heads/context_module.py:147-171 (TransformerDecoder.forward), :211-250
// (TransformerDecoderLayer.forward_post: self-attention of the memory tokens + residual + norm1, then the cross attention's k_proj /
// v_proj - Conv1d(k = 1) over the tokens, :320-338).  The memory of a frame is L = 2 * num_class tokens of 96 floats (34 x 96 on
// nuScenes) and never sees the points, so ls3d_sffm_decoder takes the k / v of every layer as an input.  Layer by layer that side is
// five launches per layer on a 34-row matrix (qkv projection, attention core, out-projection + residual + LayerNorm, k, v) plus the copies
// that put k / v into [layer][B][E][L] - ~40 launches of 7 - 30 us each, serial, in front of the decoder (0.35 ms of the 9 ms MSeg3D frame).
// Here: one workgroup of 16 waves per frame keeps the tokens in LDS through all layers and writes k / v in the decoder's layout.  Round 3's first
// version (256 threads, a thread per output column walking 96 dependent weight loads per 8 rows) was latency-bound: 0.66 ms against 0.37 ms for
// the ~40 launches.  Now every contraction is a [64 x 96] x [96 x N] product on v_mfma_f32_32x32x2_f32 (exact f32): a wave owns 32 x 32 output
// tiles, loads the 48 B-operand values of a tile (two weight rows x 32 columns per step: 2 x 128 contiguous bytes) up front - 48 independent
// loads in flight instead of one - and reads its A operand from the token tile in LDS (row stride 97 floats: conflict-free); the
// self-attention runs on 16 threads per token (4 heads x 4 key groups, softmax statistics and outputs merged by shuffles).
#include "common.h"

constexpr int SM_E = 96, SM_H = 4, SM_HD = 24, SM_LMAX = 64, SM_MAX_LAYERS = 8;
constexpr int SM_MS = SM_E + 1;       // row stride of the token tiles (floats)
constexpr int SM_QS = 3 * SM_E + 1;   // row stride of the q | k | v tile
constexpr int SM_THREADS = 1024, SM_WAVES = SM_THREADS / 64;
typedef float sm_f32x16 __attribute__((ext_vector_type(16)));

struct SmLayer {
  const float *wqkv_t, *bqkv, *wo_t, *bo, *n1g, *n1b, *wk_t, *bk, *wv_t, *bv;
  float n1eps;
};
struct SmParams {
  int num_layers;
  SmLayer layer[SM_MAX_LAYERS];
};

// C tile (rb, cb) = A[rb * 32 .. + 32][0 .. 96) (LDS, row stride lda) x W[0 .. 96)[32 columns], W(e, c) = wtile(cb)[e * ldw + c].  Tile
// t = rb + nrb * cb goes to wave t % 16.  `epi(rb, cb, acc)` gets the tile in the MFMA's C layout: acc[r] = C[(r & 3) + 8 (r >> 2) + 4 (lane >> 5)][lane & 31].
template <typename WTile, typename Epi>
__device__ __forceinline__ void sm_gemm(const float *A, int lda, int nrb, int ncb, WTile wtile, int ldw, Epi epi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31, kk = lane >> 5;
  for (int t = wave; t < nrb * ncb; t += SM_WAVES) {
    const int rb = t % nrb, cb = t / nrb;
    const float *w = wtile(cb) + (size_t)kk * ldw + col;
    float bv[SM_E / 2];
#pragma unroll
    for (int s2 = 0; s2 < SM_E / 2; ++s2) bv[s2] = w[(size_t)(2 * s2) * ldw];  // all 48 loads of the tile in flight at once
    const float *a = A + (rb * 32 + col) * lda + kk;
    sm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < SM_E / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s2], bv[s2], acc, 0, 0, 0);
    epi(rb, cb, acc);
  }
}

__global__ __launch_bounds__(SM_THREADS) void k_sffm_memory(const float *__restrict__ mem, int batch, int L, SmParams prm, float *__restrict__ kv,
                                                           float *__restrict__ mem_out) {
  HIP_DYNAMIC_SHARED(float, smem)
  float *M = smem;                       // [SM_LMAX][SM_MS] tokens (rows >= L stay zero)
  float *Q = M + SM_LMAX * SM_MS;        // [SM_LMAX][SM_QS] q | k | v of the self-attention
  float *A = Q + SM_LMAX * SM_QS;        // [SM_LMAX][SM_MS] attention output
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.x;
  const int col = lane & 31, kk = lane >> 5;
  const int nrb = L > 32 ? 2 : 1;        // 32-row blocks that hold tokens
  for (int i = tid; i < SM_LMAX * SM_MS; i += SM_THREADS) {
    const int r = i / SM_MS, c = i - r * SM_MS;
    M[i] = (r < L && c < SM_E) ? mem[((size_t)b * L + r) * SM_E + c] : 0.0f;
    A[i] = 0.0f;
  }
  __syncthreads();
  for (int l = 0; l < prm.num_layers; ++l) {
    const SmLayer &P = prm.layer[l];
    // ---- q | k | v of the self-attention: 288 columns = 9 column blocks (rows >= L hold the bias: never read)
    sm_gemm(M, SM_MS, nrb, 9, [&](int cb) { return P.wqkv_t + cb * 32; }, 3 * SM_E, [&](int rb, int cb, const sm_f32x16 &acc) {
      const int c = cb * 32 + col;
      const float bc = P.bqkv[c];
#pragma unroll
      for (int r = 0; r < 16; ++r) Q[(rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk) * SM_QS + c] = acc[r] + bc;
    });
    __syncthreads();
    // ---- softmax(q k^T / sqrt(hd)) v: 16 threads per token = 4 heads x 4 key groups (keys j = g, g + 4, ...); the groups' maxima, sums and
    //      outputs are merged over the 4 adjacent lanes by shuffles
    {
      const int r = tid >> 4, h = (tid >> 2) & 3, g = tid & 3;
      const bool on = r < L;
      const float scale = 1.0f / sqrtf((float)SM_HD);
      const float *qp = Q + (on ? r : 0) * SM_QS + h * SM_HD;
      float q[SM_HD], o[SM_HD], sc[SM_LMAX / 4];
#pragma unroll
      for (int d = 0; d < SM_HD; ++d) { q[d] = qp[d] * scale; o[d] = 0.0f; }
      float m = -3.0e38f;
#pragma unroll
      for (int u = 0; u < SM_LMAX / 4; ++u) {
        const int j = g + 4 * u;
        float s2 = -3.0e38f;
        if (j < L) {
          const float *kp = Q + j * SM_QS + SM_E + h * SM_HD;
          s2 = 0.0f;
#pragma unroll
          for (int d = 0; d < SM_HD; ++d) s2 = fmaf(q[d], kp[d], s2);
        }
        sc[u] = s2;
        m = fmaxf(m, s2);
      }
      m = fmaxf(m, __shfl_xor(m, 1));
      m = fmaxf(m, __shfl_xor(m, 2));
      float den = 0.0f;
#pragma unroll
      for (int u = 0; u < SM_LMAX / 4; ++u) {
        const int j = g + 4 * u;
        if (j < L) {
          const float *vp = Q + j * SM_QS + 2 * SM_E + h * SM_HD;
          const float p = expf(sc[u] - m);
          den += p;
#pragma unroll
          for (int d = 0; d < SM_HD; ++d) o[d] = fmaf(p, vp[d], o[d]);
        }
      }
      den += __shfl_xor(den, 1);
      den += __shfl_xor(den, 2);
#pragma unroll
      for (int d = 0; d < SM_HD; ++d) {
        o[d] += __shfl_xor(o[d], 1);
        o[d] += __shfl_xor(o[d], 2);
      }
      if (on && g == 0) {
        const float inv = 1.0f / den;
#pragma unroll
        for (int d = 0; d < SM_HD; ++d) A[r * SM_MS + h * SM_HD + d] = o[d] * inv;
      }
    }
    __syncthreads();
    // ---- out-projection + residual, in place: element (row, column) of M is read and written by one lane only
    sm_gemm(A, SM_MS, nrb, 3, [&](int cb) { return P.wo_t + cb * 32; }, SM_E, [&](int rb, int cb, const sm_f32x16 &acc) {
      const int c = cb * 32 + col;
      const float bc = P.bo[c];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row < L) M[row * SM_MS + c] += acc[r] + bc;
      }
    });
    __syncthreads();
    // ---- norm1: four lanes per token, two-pass statistics
    if (tid < 4 * SM_LMAX) {
      const int r = tid >> 2, part = tid & 3;
      float *row = M + r * SM_MS + part * (SM_E / 4);
      float s2 = 0.0f;
#pragma unroll
      for (int d = 0; d < SM_E / 4; ++d) s2 += row[d];
      s2 += __shfl_xor(s2, 1);
      s2 += __shfl_xor(s2, 2);
      const float mean = s2 / (float)SM_E;
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
    // ---- k_proj / v_proj of the cross attention -> kv[2 l + {0, 1}][b][c][token]: column blocks 0 - 2 = k, 3 - 5 = v
    sm_gemm(M, SM_MS, nrb, 6, [&](int cb) { return (cb < 3 ? P.wk_t : P.wv_t) + (cb % 3) * 32; }, SM_E, [&](int rb, int cb, const sm_f32x16 &acc) {
      const int which = cb / 3, c = (cb % 3) * 32 + col;
      const float bc = (which ? P.bv : P.bk)[c];
      float *dst = kv + (((size_t)(2 * l + which) * batch + b) * SM_E + c) * L;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (row < L) dst[row] = acc[r] + bc;
      }
    });
    // (the next layer's first phase only reads M and writes Q; its barrier orders it against this one's reads)
  }
  if (mem_out) {
    __syncthreads();
    for (int i = tid; i < L * SM_E; i += SM_THREADS) mem_out[(size_t)b * L * SM_E + i] = M[(i / SM_E) * SM_MS + i % SM_E];
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
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_sffm_memory, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_sffm_memory, dim3((unsigned)batch), dim3(SM_THREADS), lds, stream, mem, batch, L, prm, kv, mem_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
