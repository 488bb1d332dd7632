// sffm_memory.hip — the class-embedding ("memory") side of the SF-Phase decoder for all layers in ONE launch.
//
// Reference: det3d/models/point_heads/context_module.py:147-171 (TransformerDecoder.forward), :211-250
// (TransformerDecoderLayer.forward_post: self-attention of the memory tokens + residual + norm1, then the cross attention's k_proj /
// v_proj - Conv1d(k = 1) over the tokens, :320-338).  The memory of a frame is L = 2 * num_class tokens of 96 floats (34 x 96 on
// nuScenes) and never sees the points, so ls3d_sffm_decoder takes the k / v of every layer as an input.  Layer by layer that side is
// five launches per layer on a 34-row matrix (qkv projection, attention core, out-projection + residual + LayerNorm, k, v) plus the copies
// that put k / v into [layer][B][E][L] - ~40 launches of 7 - 30 us each, serial, in front of the decoder (0.35 ms of the 9 ms MSeg3D frame).
// Here: one workgroup of 16 waves per frame keeps the tokens in LDS through all layers and writes k / v in the decoder's layout (round 6: four
// phases per layer instead of five, and no second MFMA row block for a 2-token tail: 0.21 -> see profiles/round6_experiments.md).  Round 3's first
// version (256 threads, a thread per output column walking 96 dependent weight loads per 8 rows) was latency-bound: 0.66 ms against 0.37 ms for
// the ~40 launches.  Now every contraction is a [64 x 96] x [96 x N] product on v_mfma_f32_32x32x2_f32 (exact f32): a wave owns 32 x 32 output
// tiles, loads the 48 B-operand values of a tile (two weight rows x 32 columns per step: 2 x 128 contiguous bytes) up front - 48 independent
// loads in flight instead of one - and reads its A operand from the token tile in LDS (row stride 97 floats: conflict-free); the
// self-attention runs on 16 threads per token (4 heads x 4 key groups, softmax statistics and outputs merged by shuffles).
#include "common.h"

constexpr int SM_E = 96, SM_H = 4, SM_HD = 24, SM_LMAX = 64, SM_MAX_LAYERS = 8;
constexpr int SM_MS = SM_E + 1;       // row stride of the token tiles (floats)
constexpr int SM_QS = SM_E + 1;       // row stride of the q tile of the self-attention
constexpr int SM_KS = 2 * SM_E + 4;   // row stride of its k | v tile: a multiple of 4 floats, so that a head's 24 values are six 16-byte reads
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

// One GEMM phase over the token tile A (LDS, row stride lda, K = 96): `ntile` column blocks of 32 outputs, tile t = W_t[0 .. 96)[32 columns] with
// W_t(e, c) = wtile(t)[e * ldw(t) + c]; put(t, row, col, value + bias(t, col)) receives every output of the rows < L.
//   * rows 0 .. 31 (and 32 .. 63 when the tail is long) on v_mfma_f32_32x32x2_f32, one 32 x 32 tile per wave.  ALL 48 B-operand values of a tile (two
//     weight rows x 32 columns per step) and the tile's bias are loaded before the first MFMA - a scheduling fence keeps hipcc from interleaving them
//     with the MFMAs, which it does with 5 - 9 loads in flight (round 6 trace, tools/trace_memory.py: a 15-tile phase took 44 k cycles, 48 dependent
//     MFMAs are 3 k) - the A operand comes from LDS;
//   * a SHORT tail (L - 32 <= SM_TAIL rows: nuScenes' 34 = 2 x 17 tokens leave 2) from the SAME registers: lane (col, kk) holds W[2 s + kk][col] for
//     s = 0 .. 47, i.e. the even (kk = 0) or odd (kk = 1) half of its column's weights, so a tail row's output is two 48-term dot products and one
//     cross-half shuffle - no second MFMA row block (a full tile and a second round of the 16 waves for 2 live rows), no further loads.
constexpr int SM_TAIL = 8;
template <typename WTile, typename Ldw, typename Bias, typename Put>
__device__ __forceinline__ void sm_phase(const float *A, int lda, int L, int ntile, WTile wtile, Ldw ldw, Bias bias, Put put) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int col = lane & 31, kk = lane >> 5;
#ifndef HIPSIM
  asm volatile("" : "+v"(col), "+v"(kk));  // per-call values for the compiler: hoisted out of the layer loop, the 16 row predicates and LDS addresses of
#endif                                       // every phase would live (and spill) across the whole kernel
  const bool short_tail = L > 32 && L - 32 <= SM_TAIL;
  const int nrb = (L > 32 && !short_tail) ? 2 : 1;
  for (int t = wave; t < nrb * ntile; t += SM_WAVES) {
    const int rb = t % nrb, cb = t / nrb, ld = ldw(cb);
    const float *w = wtile(cb) + (size_t)kk * ld + col;
    float bv[SM_E / 2];
#pragma unroll
    for (int s2 = 0; s2 < SM_E / 2; ++s2) bv[s2] = w[(size_t)(2 * s2) * ld];
    const float bc = bias(cb, col);
    LS3D_SCHED_FENCE();
    const float *a = A + (rb * 32 + col) * lda + kk;
    sm_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s2 = 0; s2 < SM_E / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * s2], bv[s2], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {  // the MFMA's C layout: acc[r] = C[(r & 3) + 8 (r >> 2) + 4 (lane >> 5)][lane & 31]
      const int row = rb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (row < L) put(cb, row, col, acc[r] + bc);
    }
    if (short_tail) {
      for (int row = 32; row < L; ++row) {
        const float *ar = A + row * lda + kk;
        float part = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < SM_E / 2; ++s2) part = fmaf(ar[2 * s2], bv[s2], part);
        part += __shfl_xor(part, 32);
        if (kk == ((row - 32) & 1)) put(cb, row, col, part + bc);  // the two halves hold the same sum: they take the rows alternately
      }
    }
  }
}

// TR: tracing build (ls3d_sffm_memory_trace): thread 0 of frame 0 records the shader clock at every phase boundary
template <bool TR>
__global__ __launch_bounds__(SM_THREADS) void k_sffm_memory(const float *__restrict__ mem, int batch, int L, SmParams prm, float *__restrict__ kv,
                                                           float *__restrict__ mem_out, unsigned long long *__restrict__ trace) {
  [[maybe_unused]] int tr_n = 0;
#define SM_MARK()                                                                   \
  if constexpr (TR) {                                                               \
    if (threadIdx.x == 0 && blockIdx.x == 0) trace[tr_n] = ls3d_cycles();           \
    ++tr_n;                                                                         \
  }
  HIP_DYNAMIC_SHARED(float, smem)
  float *M = smem;                       // [SM_LMAX][SM_MS] tokens (rows >= L stay zero)
  float *Q = M + SM_LMAX * SM_MS;        // [SM_LMAX][SM_QS] q of the self-attention
  float *A = Q + SM_LMAX * SM_QS;        // [SM_LMAX][SM_MS] attention output
  float *KV = A + SM_LMAX * SM_MS;       // [SM_LMAX][SM_KS] k | v of the self-attention (16-byte aligned rows: the attention reads them with ds_read_b128 -
                                         // a key's 24 values per head were 24 ds_read_b32 on the odd-stride q | k | v tile, 432 LDS instructions per thread and layer)
  const int tid = threadIdx.x, b = blockIdx.x;
  const int NL = prm.num_layers;
  for (int i = tid; i < SM_LMAX * SM_MS; i += SM_THREADS) {
    const int r = i / SM_MS, c = i - r * SM_MS;
    M[i] = (r < L && c < SM_E) ? mem[((size_t)b * L + r) * SM_E + c] : 0.0f;
    A[i] = 0.0f;
  }
  __syncthreads();
  SM_MARK()  // 0: tokens staged
  // Everything that reads the tokens as they stand between two layers in ONE phase: the cross attention's k / v projections of layer l - 1
  // (6 column blocks -> kv[2 (l - 1) + {0, 1}][b][c][token], the decoder's layout) and the q | k | v projection of layer l's self-attention (9
  // column blocks -> Q): 15 tiles for the 16 waves, one round, one barrier - they were two phases (and, with 34 tokens, 18 + 12 tiles).
  auto token_phase = [&](int l) {
    const int nq = l < NL ? 9 : 0, nk = l >= 1 ? 6 : 0;
    const SmLayer &Pq = prm.layer[l < NL ? l : 0], &Pk = prm.layer[l >= 1 ? l - 1 : 0];
    sm_phase(M, SM_MS, L, nq + nk,
             [&](int t) { return t < nq ? Pq.wqkv_t + t * 32 : ((t - nq) < 3 ? Pk.wk_t : Pk.wv_t) + ((t - nq) % 3) * 32; },
             [&](int t) { return t < nq ? 3 * SM_E : SM_E; },
             [&](int t, int col) { return t < nq ? Pq.bqkv[t * 32 + col] : ((t - nq) < 3 ? Pk.bk : Pk.bv)[((t - nq) % 3) * 32 + col]; },
             [&](int t, int row, int col, float y) {
               if (t < nq) {
                 const int c = t * 32 + col;
                 if (c < SM_E) Q[row * SM_QS + c] = y;
                 else KV[row * SM_KS + c - SM_E] = y;
               } else {
                 const int which = (t - nq) / 3, c = ((t - nq) % 3) * 32 + col;
                 kv[(((size_t)(2 * (l - 1) + which) * batch + b) * SM_E + c) * L + row] = y;
               }
             });
  };
  token_phase(0);
  for (int l = 0; l < NL; ++l) {
    const SmLayer &P = prm.layer[l];
    __syncthreads();
    SM_MARK()  // 1 + 4 l: q | k | v (+ the previous layer's k / v projections) done
    // ---- softmax(q k^T / sqrt(hd)) v: 16 threads per token = 4 heads x 4 key groups (keys j = g, g + 4, ...); the groups' maxima, sums and
    //      outputs are merged over the 4 adjacent lanes by shuffles
    if ((tid >> 6) * 4 < L) {  // a wave holds 4 tokens: the waves beyond the last token (7 of 16 at 34 tokens) skip the phase
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
          const float4 *kp = (const float4 *)(KV + j * SM_KS + h * SM_HD);
          s2 = 0.0f;
#pragma unroll
          for (int d4 = 0; d4 < SM_HD / 4; ++d4) {
            const float4 kq = kp[d4];
            s2 = fmaf(q[4 * d4], kq.x, s2); s2 = fmaf(q[4 * d4 + 1], kq.y, s2); s2 = fmaf(q[4 * d4 + 2], kq.z, s2); s2 = fmaf(q[4 * d4 + 3], kq.w, s2);
          }
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
          const float4 *vp = (const float4 *)(KV + j * SM_KS + SM_E + h * SM_HD);
          const float p = expf(sc[u] - m);
          den += p;
#pragma unroll
          for (int d4 = 0; d4 < SM_HD / 4; ++d4) {
            const float4 vq = vp[d4];
            o[4 * d4] = fmaf(p, vq.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vq.y, o[4 * d4 + 1]);
            o[4 * d4 + 2] = fmaf(p, vq.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vq.w, o[4 * d4 + 3]);
          }
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
    SM_MARK()  // 2 + 4 l: self-attention done
    // ---- out-projection + residual, in place: element (row, column) of M is read and written by one thread only
    sm_phase(A, SM_MS, L, 3, [&](int t) { return P.wo_t + t * 32; }, [&](int) { return SM_E; }, [&](int t, int col) { return P.bo[t * 32 + col]; },
             [&](int t, int row, int col, float y) { M[row * SM_MS + t * 32 + col] += y; });
    __syncthreads();
    SM_MARK()  // 3 + 4 l: out-projection done
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
    SM_MARK()  // 4 + 4 l: norm1 done
    token_phase(l + 1);  // k / v of this layer for the decoder + q | k | v of the next layer's self-attention
  }
  __syncthreads();
  SM_MARK()  // 1 + 4 NL: the last k / v projections done
#undef SM_MARK
  if (mem_out) {
    __syncthreads();
    for (int i = tid; i < L * SM_E; i += SM_THREADS) mem_out[(size_t)b * L * SM_E + i] = M[(i / SM_E) * SM_MS + i % SM_E];
  }
}

static int sm_launch(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers,
                     float *kv, float *mem_out, unsigned long long *trace, ls3d_stream_t stream_) {
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
  const int lds = SM_LMAX * (2 * SM_MS + SM_QS + SM_KS) * (int)sizeof(float);
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_sffm_memory<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_memory<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  if (trace) hipLaunchKernelGGL(k_sffm_memory<true>, dim3((unsigned)batch), dim3(SM_THREADS), lds, stream, mem, batch, L, prm, kv, mem_out, trace);
  else hipLaunchKernelGGL(k_sffm_memory<false>, dim3((unsigned)batch), dim3(SM_THREADS), lds, stream, mem, batch, L, prm, kv, mem_out, trace);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_sffm_memory(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers,
                                float *kv, float *mem_out, ls3d_stream_t stream) {
  return sm_launch(mem, batch, L, embed, heads, num_layers, layers, kv, mem_out, nullptr, stream);
}

extern "C" int ls3d_sffm_memory_trace(const float *mem, int batch, int L, int embed, int heads, int num_layers, const ls3d_sffm_memory_layer_t *layers,
                                      float *kv, unsigned long long *trace, ls3d_stream_t stream) {
  if (!trace) return LS3D_ERR_ARG;
  return sm_launch(mem, batch, L, embed, heads, num_layers, layers, kv, nullptr, trace, stream);
}
