// sffm.hip — the point side of the SF-Phase decoder (SemanticFeatureFusionModule) as ONE kernel.
//
// Reference: det3d/models/point_heads/context_module.py:56-117 (SFFM.forward), :211-250 (TransformerDecoderLayer.forward_post),
// :320-376 (SparsePointCorssAttention).  Per point: input projection, then num_layers x { q-projection, attention of the point
// over the L = 2*num_class class embeddings of its frame (4 heads), out-projection + residual + LayerNorm, FFN(ReLU) + residual
// + LayerNorm }, then the decoder's final LayerNorm.  The class-embedding side (their self-attention, k / v projections) does
// not depend on the points: the caller evaluates it for every layer first (34 rows per frame) and passes k, v per layer.
//
// The layer-by-layer version (round 1) was ~60 launches per frame, each reading and writing the [N, 96] point matrix
// (2.0 ms GEMMs + 0.6 ms attention for 120k points).  Here a wave keeps its 32 points in LDS from the projected input to the
// final LayerNorm: a point is read once (d_in floats) and written once (96 floats); only the weights stream (L2 -> LDS,
// 221 KB per layer and 128-point tile).
//   * GEMMs: v_mfma_f32_32x32x2_f32 (exact f32; gemm_products = 6: the exact 3-plane bf16 split, sf_gemm_planes), all of shape [32 x K] x [K x 96], K in {d_in, 96}: three accumulators per wave,
//     A fragments from the wave's LDS tile (row stride 100 floats), B = 32 x 96 weight chunks in the packed layout of
//     ls3d_gather_gemm_pack(nt = 3), double buffered in LDS, staged by the 4 waves together.  The FFN (96 -> 192 -> 96) runs as two
//     96-wide halves accumulated into the same three accumulators, so the hidden tile is 96 wide too.
//   * attention: one lane per (point, head), q read from / the result written over the point's own slice of the scratch tile;
//     the frame's K and V of the layer are staged transposed in LDS ([head][l][24]); a tile that straddles two frames reads
//     them from L2 instead.
//   * LayerNorm: two lanes per row, statistics by one shuffle.
// LDS: 4 x 2 x 32 x 100 floats + 2 x 12 KB weight chunks + 26 KB K/V = 152 KB -> one workgroup per CU.
#include "common.h"
#include "gemm_common.h"

typedef float sf_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 sf_bf16x8 __attribute__((ext_vector_type(8)));

#ifdef HIPSIM
#define SF_WAVE_SYNC() hipsim::wave_barrier()
#else
#define SF_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

constexpr int SF_E = 96, SF_H = 4, SF_HD = 24, SF_LMAX = 36;
constexpr int SF_XS = SF_E + 4;              // row stride of the point tiles
constexpr int SF_WAVE_FLOATS = 2 * 32 * SF_XS;
constexpr int SF_BCHUNK = 32 * SF_E;         // floats in one staged weight chunk (32 k x 96 columns)
constexpr int SF_KV = 2 * SF_E * SF_LMAX;    // K and V of one frame and layer, transposed
#define SF_MAX_LAYERS 8

struct SfLayer {
  const float *wq, *bq, *wo, *bo, *w1a, *w1b, *b1, *w2a, *w2b, *b2, *n2g, *n2b, *n3g, *n3b;
  float n2eps, n3eps;
  const uint4 *pwq, *pwo, *pw1a, *pw1b, *pw2a, *pw2b;  // the same matrices as three bf16 planes (ls3d_tile_conv_pack, kvol = 1), GP = 6
};
struct SfParams {
  const float *win, *bin, *ng, *nb;
  float neps;
  int num_layers, d_in;
  int ablate;  // measurement only (gemm_products bits 8..): 1 no attention, 2 no FFN, 4 no LayerNorm, 8 GEMMs without MFMAs, 16 GEMMs without weight staging
  const uint4 *pwin;
  SfLayer layer[SF_MAX_LAYERS];
};

// acc[0..2] (+)= A(32 x K, LDS, stride lda) x W (packed nt = 3: [K][32][3]); K % 32 == 0.  All 4 waves call this together.
__device__ __forceinline__ void sf_gemm(const float *A, int lda, int K, const float *__restrict__ Wp, float *Bs, sf_f32x16 (&acc)[3], bool zero) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int col = lane & 31, kk = lane >> 5;
  const int nkc = K / 32;
  float4 r0, r1, r2;
  {
    const float4 *src = (const float4 *)Wp;
    r0 = src[tid]; r1 = src[tid + 256]; r2 = src[tid + 512];
  }
  __syncthreads();  // previous users of Bs are done
  ((float4 *)Bs)[tid] = r0; ((float4 *)Bs)[tid + 256] = r1; ((float4 *)Bs)[tid + 512] = r2;
  __syncthreads();
  if (zero) {
#pragma unroll
    for (int n = 0; n < 3; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  }
  int buf = 0;
  for (int c = 0; c < nkc; ++c) {
    if (c + 1 < nkc) {
      const float4 *src = (const float4 *)(Wp + (size_t)(c + 1) * SF_BCHUNK);
      r0 = src[tid]; r1 = src[tid + 256]; r2 = src[tid + 512];
    }
    {
      const float4 *ap = (const float4 *)(A + col * lda + c * 32 + kk * 16);
      const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
      const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
      const float *bs = Bs + buf * SF_BCHUNK + (kk * 16 * 32 + col) * 3;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float b0 = bs[u * 96], b1 = bs[u * 96 + 1], b2 = bs[u * 96 + 2];
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b0, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b1, acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b2, acc[2], 0, 0, 0);
      }
    }
    if (c + 1 < nkc) {
      float4 *dst = (float4 *)(Bs + (buf ^ 1) * SF_BCHUNK);
      dst[tid] = r0; dst[tid + 256] = r1; dst[tid + 512] = r2;
      __syncthreads();
      buf ^= 1;
    }
  }
}

// The same product on the exact 3-plane bf16 split of both operands (the arithmetic of k_tile_conv and of the reader's GEMMs):
// Wq = the matrix packed by ls3d_tile_conv_pack(kvol = 1, cout = 96): per 16-channel chunk [column block 4][plane 3][kk 2][col 32] x 8 bf16 =
// 12 KB, of which the first three column blocks (9 KB) are staged; the A fragment (8 contiguous floats per lane and chunk) is split in
// registers (round-to-nearest planes); six v_mfma_f32_32x32x16_bf16 per column block and chunk, head x head in its own accumulator:
// 18 MFMAs of 32 cycles per 16 channels instead of 24 of 64.  Chunks of 16 channels so that two weight buffers fit where the f32 path keeps
// its two 32-channel chunks (LDS stays at 152 KB).
constexpr int SF_PCHUNK = 576;   // uint4 staged per chunk (3 column blocks x 3 planes x 64)
constexpr int SF_PSTRIDE = 768;  // uint4 per chunk in the packed matrix (4 column blocks: 96 columns are padded to 128)
// one staged chunk in flight: 9 KB = 576 x 16 bytes over 256 threads (the third unit only for the first 64; an unconditional load from a
// clamped address otherwise, so that hipcc can count on it)
struct SfPre { uint4 a, b, c; };
__device__ __forceinline__ SfPre sf_fetch(const uint4 *__restrict__ chunk) {
  const int tid = threadIdx.x;
  SfPre p;
  p.a = chunk[tid]; p.b = chunk[tid + 256]; p.c = chunk[tid < SF_PCHUNK - 512 ? tid + 512 : tid];
  return p;
}
__device__ __forceinline__ void sf_stash(uint4 *dst, const SfPre &p) {
  const int tid = threadIdx.x;
  dst[tid] = p.a; dst[tid + 256] = p.b;
  if (tid < SF_PCHUNK - 512) dst[tid + 512] = p.c;
}
// accumulator fragment (register r of lane (col, kk)) -> tile row:  row = (r & 3) + 8 * (r >> 2) + 4 * kk; column = 32 n + col
#define SF_FOR_ACC(n, r, row) \
  _Pragma("unroll") for (int n = 0; n < 3; ++n) _Pragma("unroll") for (int r = 0, row = 4 * kk; r < 16; ++r, row = (r & 3) + 8 * (r >> 2) + 4 * kk)

// in-place LayerNorm of the 32 x 96 tile X (two lanes per row, 48 elements each)
__device__ __forceinline__ void sf_layernorm(float *X, const float *g, const float *b, float eps) {
  const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
  float4 *p = (float4 *)(X + row * SF_XS + half * 48);
  float4 v[12];
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 12; ++q) { v[q] = p[q]; s += (v[q].x + v[q].y) + (v[q].z + v[q].w); }
  s += __shfl_xor(s, 32);
  const float mean = s / (float)SF_E;
  float q2 = 0.0f;
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    v[q].x -= mean; v[q].y -= mean; v[q].z -= mean; v[q].w -= mean;
    q2 += (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
  }
  q2 += __shfl_xor(q2, 32);
  const float rstd = 1.0f / sqrtf(q2 / (float)SF_E + eps);
  const float4 *gp = (const float4 *)(g + half * 48), *bp = (const float4 *)(b + half * 48);
#pragma unroll
  for (int q = 0; q < 12; ++q) {
    const float4 gg = gp[q], bb = bp[q];
    float4 o;
    o.x = v[q].x * rstd * gg.x + bb.x; o.y = v[q].y * rstd * gg.y + bb.y;
    o.z = v[q].z * rstd * gg.z + bb.z; o.w = v[q].w * rstd * gg.w + bb.w;
    p[q] = o;
  }
}

// attention of the wave's 32 points over the L class embeddings, f32 on the vector pipe: one lane per (point, head), two passes
// of 16 points.  q in T[:, 0:96] (overwritten by the result).  kvs: the staged [2][H][L][HD] copy of frame `fs`, or null.
__device__ __forceinline__ void sf_attention_valu(float *T, const float *kvs, int fs, const float *kg, const float *vg, int L, const int *s_frame,
                                                  int wave) {
  const int lane = threadIdx.x & 63;
  const float scale = 1.0f / sqrtf((float)SF_HD);
  for (int ps = 0; ps < 2; ++ps) {
    const int row = ps * 16 + (lane >> 2), h = lane & 3;
    const int f = s_frame[wave * 32 + row];
    float *qp = T + row * SF_XS + h * SF_HD;
    float q[SF_HD], o[SF_HD];
#pragma unroll
    for (int d = 0; d < SF_HD; ++d) { q[d] = qp[d]; o[d] = 0.0f; }
    if (f >= 0) {
      float m = -3.0e38f, den = 0.0f;
      if (kvs && f == fs) {
        const float4 *kb = (const float4 *)(kvs + h * L * SF_HD), *vb = (const float4 *)(kvs + SF_E * L + h * L * SF_HD);
        for (int l = 0; l < L; ++l) {
          float s = 0.0f;
#pragma unroll
          for (int d4 = 0; d4 < SF_HD / 4; ++d4) {
            const float4 kv = kb[l * (SF_HD / 4) + d4];
            s = fmaf(q[4 * d4], kv.x, s); s = fmaf(q[4 * d4 + 1], kv.y, s); s = fmaf(q[4 * d4 + 2], kv.z, s); s = fmaf(q[4 * d4 + 3], kv.w, s);
          }
          m = fmaxf(m, s * scale);
        }
        for (int l = 0; l < L; ++l) {
          float s = 0.0f;
#pragma unroll
          for (int d4 = 0; d4 < SF_HD / 4; ++d4) {
            const float4 kv = kb[l * (SF_HD / 4) + d4];
            s = fmaf(q[4 * d4], kv.x, s); s = fmaf(q[4 * d4 + 1], kv.y, s); s = fmaf(q[4 * d4 + 2], kv.z, s); s = fmaf(q[4 * d4 + 3], kv.w, s);
          }
          const float pr = expf(s * scale - m);
          den += pr;
#pragma unroll
          for (int d4 = 0; d4 < SF_HD / 4; ++d4) {
            const float4 vv = vb[l * (SF_HD / 4) + d4];
            o[4 * d4] = fmaf(pr, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(pr, vv.y, o[4 * d4 + 1]);
            o[4 * d4 + 2] = fmaf(pr, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(pr, vv.w, o[4 * d4 + 3]);
          }
        }
      } else {  // another frame than the staged one: straight from L2 ([b][h][d][l])
        const float *kb = kg + ((size_t)f * SF_H + h) * SF_HD * L, *vb = vg + ((size_t)f * SF_H + h) * SF_HD * L;
        for (int l = 0; l < L; ++l) {
          float s = 0.0f;
#pragma unroll
          for (int d = 0; d < SF_HD; ++d) s = fmaf(q[d], kb[d * L + l], s);
          m = fmaxf(m, s * scale);
        }
        for (int l = 0; l < L; ++l) {
          float s = 0.0f;
#pragma unroll
          for (int d = 0; d < SF_HD; ++d) s = fmaf(q[d], kb[d * L + l], s);
          const float pr = expf(s * scale - m);
          den += pr;
#pragma unroll
          for (int d = 0; d < SF_HD; ++d) o[d] = fmaf(pr, vb[d * L + l], o[d]);
        }
      }
      const float inv = 1.0f / den;
#pragma unroll
      for (int d = 0; d < SF_HD; ++d) o[d] *= inv;
    }
#pragma unroll
    for (int d = 0; d < SF_HD; ++d) qp[d] = o[d];
  }
}

// attention of the wave's 32 points on the matrix pipe, everything in registers.  Per head the scores are computed TRANSPOSED,
// S^T[token][point] = K_h[token][:] . q[point][:]: the keys are the MFMA's rows, the points its columns, so lane (point, kk) ends
// up with 16 of the first 32 tokens' scores of ITS point (the other 16 sit in lane ^ 32): the softmax is a reduction inside a lane
// plus one shuffle.  The second product O^T[d][point] = sum_token V_h[token][d] P[token][point] then takes P straight from those
// registers as its B operand (the MFMA's K index is a free permutation: step s, half kk <-> the token register s of half kk
// holds), with V_h^T as the A operand from LDS.  Tokens 32..L-1 (two of the 34 on nuScenes) go through the vector pipe.
// MODE 0: v_mfma_f32_32x32x2_f32 (exact f32 products: the same arithmetic as the vector-pipe version up to summation order);
// MODE 1: v_mfma_f32_32x32x16_bf16, operands rounded to bf16; MODE 2: v_mfma_f32_32x32x16_fp8_fp8, operands rounded to OCP e4m3
// (the probabilities scaled by 256 so that small ones do not underflow) - f32 accumulation and f32 softmax in both (BASELINE configs[4]).
// Requires all 32 points in frame fs (staged K / V) and L <= SF_LMAX.
__device__ __forceinline__ long sf_pack_fp8(const float (&v)[8]) {
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
  return (long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template <int MODE>
__device__ __forceinline__ void sf_attention_mfma(float *T, const float *kvs, int L, const unsigned *kvmax) {
  constexpr bool BF16 = MODE != 0;  // the two reduced-precision forms share the K = 16 operand walk
  constexpr float F8_TOP = 240.0f;  // e4m3 operands are scaled so that their largest magnitude lands here (max normal 448)
  const int lane = threadIdx.x & 63, col = lane & 31, kk = lane >> 5;
  const float scale = 1.0f / sqrtf((float)SF_HD);
  const int nx = L > 32 ? L - 32 : 0;  // tokens handled on the vector pipe (<= 4)
  for (int h = 0; h < SF_H; ++h) {
    const float *Kh = kvs + h * L * SF_HD, *Vh = kvs + SF_E * L + h * L * SF_HD;
    const float *qrow = T + col * SF_XS + h * SF_HD;
    sf_f32x16 sc;
#pragma unroll
    for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
    const int tok = col < L ? col : 0;  // this lane's key row of the A operand (rows >= L are masked below)
    float ksc = 1.0f, qsc = 1.0f, vsc = 1.0f, unscale = 1.0f;
    if constexpr (MODE == 2) {  // e4m3: per-head scale for K and V, per-point scale for q (both lanes of a point agree by one shuffle)
      const float km = __uint_as_float(kvmax[h]), vm = __uint_as_float(kvmax[4 + h]);
      float qm = 0.0f;
#pragma unroll
      for (int d = 0; d < SF_HD; ++d) qm = fmaxf(qm, fabsf(qrow[d]));
      ksc = km > 0.0f ? F8_TOP / km : 1.0f;
      vsc = vm > 0.0f ? F8_TOP / vm : 1.0f;
      qsc = qm > 0.0f ? F8_TOP / qm : 1.0f;
      unscale = 1.0f / (ksc * qsc);
    }
    if constexpr (!BF16) {
      const float4 *ka = (const float4 *)(Kh + tok * SF_HD + kk * 12), *qb = (const float4 *)(qrow + kk * 12);
      const float4 k0 = ka[0], k1 = ka[1], k2 = ka[2], q0 = qb[0], q1 = qb[1], q2 = qb[2];
      const float kv_[12] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y, k1.z, k1.w, k2.x, k2.y, k2.z, k2.w};
      const float qv_[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
      for (int u = 0; u < 12; ++u) sc = __builtin_amdgcn_mfma_f32_32x32x2f32(kv_[u], qv_[u], sc, 0, 0, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {  // dims 16 t + 8 kk + j, zero beyond 24
        float af[8], bf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 16 * t + 8 * kk + j;
          af[j] = d < SF_HD ? Kh[tok * SF_HD + d] * ksc : 0.0f;
          bf[j] = d < SF_HD ? qrow[d] * qsc : 0.0f;
        }
        if constexpr (MODE == 1) {
          sf_bf16x8 a, b;
#pragma unroll
          for (int j = 0; j < 8; ++j) { a[j] = (__bf16)af[j]; b[j] = (__bf16)bf[j]; }
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, sc, 0, 0, 0);
        } else {
          sc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(sf_pack_fp8(af), sf_pack_fp8(bf), sc, 0, 0, 0);
        }
      }
    }
    // extra tokens 32 + kk + 2 j on the vector pipe
    float sx[2] = {-3.0e38f, -3.0e38f};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int te = 32 + kk + 2 * j;
      if (te < L) {
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < SF_HD; ++d) s = fmaf(qrow[d], Kh[te * SF_HD + d], s);
        sx[j] = s * scale;
      }
    }
    float m = fmaxf(sx[0], sx[1]);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int token = (r & 3) + 8 * (r >> 2) + 4 * kk;
      sc[r] = token < L ? sc[r] * (scale * unscale) : -3.0e38f;
      m = fmaxf(m, sc[r]);
    }
    m = fmaxf(m, __shfl_xor(m, 32));
    float den = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int token = (r & 3) + 8 * (r >> 2) + 4 * kk;
      sc[r] = token < L ? expf(sc[r] - m) : 0.0f;
      den += sc[r];
    }
    float px[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      px[j] = (32 + kk + 2 * j) < L ? expf(sx[j] - m) : 0.0f;
      den += px[j];
    }
    den += __shfl_xor(den, 32);
    // O^T[d][point]: A = V_h^T (row d, this lane's half of the step's tokens), B = the probabilities this lane holds
    sf_f32x16 oc;
#pragma unroll
    for (int r = 0; r < 16; ++r) oc[r] = 0.0f;
    const int dv = col < SF_HD ? col : 0;  // rows >= 24 of the output are discarded
    if constexpr (!BF16) {
#pragma unroll
      for (int s2 = 0; s2 < 16; ++s2) {
        const int token = (s2 & 3) + 8 * (s2 >> 2) + 4 * kk;
        const float va = token < L ? Vh[token * SF_HD + dv] : 0.0f;
        oc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, sc[s2], oc, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float af[8], bf[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int s2 = 8 * t + j, token = (s2 & 3) + 8 * (s2 >> 2) + 4 * kk;
          af[j] = token < L ? Vh[token * SF_HD + dv] * vsc : 0.0f;
          bf[j] = MODE == 2 ? sc[s2] * 256.0f : sc[s2];
        }
        if constexpr (MODE == 1) {
          sf_bf16x8 a, b;
#pragma unroll
          for (int j = 0; j < 8; ++j) { a[j] = (__bf16)af[j]; b[j] = (__bf16)bf[j]; }
          oc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, oc, 0, 0, 0);
        } else {
          oc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(sf_pack_fp8(af), sf_pack_fp8(bf), oc, 0, 0, 0);
        }
      }
      if constexpr (MODE == 2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) oc[r] *= (1.0f / 256.0f) / vsc;
      }
    }
    // the extra tokens' share: this lane holds p of tokens 32 + kk + 2 j, its partner those of 32 + (1 - kk) + 2 j
    float pall[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float other = __shfl_xor(px[j], 32);
      pall[2 * j + kk] = px[j];
      pall[2 * j + (1 - kk)] = other;
    }
    const float inv = 1.0f / den;
    float *orow = T + col * SF_XS + h * SF_HD;
#pragma unroll
    for (int r = 0; r < 12; ++r) {  // registers 12..15 are output rows d >= 24
      const int d = (r & 3) + 8 * (r >> 2) + 4 * kk;
      float o = oc[r];
      for (int e = 0; e < nx; ++e) o = fmaf(pall[e], Vh[(32 + e) * SF_HD + d], o);
      orow[d] = o * inv;
    }
  }
}

// one workgroup = 128 consecutive points (4 waves x 32); GEMMs on exact-f32 MFMA (gemm_products = 0)
__global__ __launch_bounds__(256, 1) void k_sffm_decoder(const float *__restrict__ x, int x_ld, int n, const float *__restrict__ points,
                                                         int pt_stride, const float *__restrict__ kv, int L, int batch, SfParams prm,
                                                         float *__restrict__ out, int out_ld, int att_mode) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  float *X = smem + wave * SF_WAVE_FLOATS;       // [32][SF_XS] the point tile
  float *T = X + 32 * SF_XS;                     // [32][SF_XS] scratch: input / q / attention output / FFN hidden half
  float *Bs = smem + 4 * SF_WAVE_FLOATS;         // [2][SF_BCHUNK]
  float *KVs = Bs + 2 * SF_BCHUNK;               // [2][H][L][HD]
  int *s_frame = (int *)(KVs + SF_KV);           // [128] frame of each point (-1 beyond n)
  unsigned *s_kvmax = (unsigned *)(s_frame + 128);  // [8] max |K_h|, max |V_h| of the staged frame and layer (bit patterns)
  const size_t kv_layer = (size_t)2 * batch * SF_E * L;  // floats per layer: k[batch][E][L] then v[batch][E][L]
  for (int p0 = blockIdx.x * 128; p0 < n; p0 += gridDim.x * 128) {
    __syncthreads();
    if (tid < 128) {  // rows of no frame - beyond n, or padding rows of a point-count bucket (batch index >= batch, graph.FrameGraph) - count as the wave's frame
      const int fr = (p0 + tid < n) ? (int)points[(size_t)(p0 + tid) * pt_stride] : -1;
      s_frame[tid] = (fr >= 0 && fr < batch) ? fr : -1;
    }
    // ---- the wave's 32 input rows -> T[:, 0:d_in]
    for (int i = lane; i < 32 * (prm.d_in / 4); i += 64) {
      const int row = i / (prm.d_in / 4), c4 = i - row * (prm.d_in / 4);
      const int p = p0 + wave * 32 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < n) v = *(const float4 *)(x + (size_t)p * x_ld + c4 * 4);
      *(float4 *)(T + row * SF_XS + c4 * 4) = v;
    }
    __syncthreads();
    const int fs = s_frame[0] >= 0 ? s_frame[0] : 0;  // the frame whose K / V are staged (the tile's first point; a tile of padding rows only: any frame)
    sf_f32x16 acc[3];
#define SF_GEMM(A_, K_, Wf_, Wp_, Wnext_, acc_, zero_) sf_gemm(A_, SF_XS, K_, Wf_, Bs, acc_, zero_)
    // ---- input projection -> X
    SF_GEMM(T, prm.d_in, prm.win, prm.pwin, (prm.num_layers ? prm.layer[0].pwq : nullptr), acc, true);
    SF_FOR_ACC(nn, r, row) X[row * SF_XS + nn * 32 + col] = acc[nn][r] + prm.bin[nn * 32 + col];
    SF_WAVE_SYNC();
    for (int l = 0; l < prm.num_layers; ++l) {
      const SfLayer &Ly = prm.layer[l];
      const float *kg = kv + (size_t)l * kv_layer, *vg = kg + (size_t)batch * SF_E * L;
      // ---- q projection -> T
      SF_GEMM(X, SF_E, Ly.wq, Ly.pwq, Ly.pwo, acc, true);
      SF_FOR_ACC(nn, r, row) T[row * SF_XS + nn * 32 + col] = acc[nn][r] + Ly.bq[nn * 32 + col];
      // ---- K / V of the layer and the tile's first frame, transposed: source [h][d][l] -> LDS [h][l][d]
      __syncthreads();  // every wave is past the previous layer's attention (KVs) and has written its q
      const bool staged = L <= SF_LMAX;  // more class embeddings than the LDS window holds: K / V are read from L2
      if (tid < 8) s_kvmax[tid] = 0u;
      if (staged) {
        if (att_mode == 3) __syncthreads();
        const float *kb0 = kg + (size_t)fs * SF_E * L, *vb0 = vg + (size_t)fs * SF_E * L;
        for (int i = tid; i < SF_E * L; i += 256) {
          const int ll = i % L, hd = i / L, h = hd / SF_HD, d = hd - h * SF_HD;
          const int o = (h * L + ll) * SF_HD + d;
          const float kx = kb0[i], vx = vb0[i];
          KVs[o] = kx;
          KVs[SF_E * L + o] = vx;
          if (att_mode == 3) {  // per-head magnitudes for the e4m3 operand scaling (bit patterns of non-negative floats order like uints)
            atomicMax(&s_kvmax[h], __float_as_uint(fabsf(kx)));
            atomicMax(&s_kvmax[4 + h], __float_as_uint(fabsf(vx)));
          }
        }
      }
      __syncthreads();
      {
        // matrix-pipe attention when the wave's 32 points all belong to the staged frame (the rule: frames are contiguous runs of
        // tens of thousands of points) - rows beyond n count as that frame: their q is the bias row, their output is never stored
        const int fr = s_frame[wave * 32 + col];
        const bool mixed = __any(fr >= 0 && fr != fs);
        if (!mixed && att_mode != 2 && staged) {
          if (att_mode == 1) sf_attention_mfma<1>(T, KVs, L, s_kvmax);
          else if (att_mode == 3) sf_attention_mfma<2>(T, KVs, L, s_kvmax);
          else sf_attention_mfma<0>(T, KVs, L, s_kvmax);
        } else {
          sf_attention_valu(T, staged ? KVs : nullptr, fs, kg, vg, L, s_frame, wave);
        }
      }
      SF_WAVE_SYNC();
      // ---- out projection + residual -> X, LayerNorm (norm2)
      SF_GEMM(T, SF_E, Ly.wo, Ly.pwo, Ly.pw1a, acc, true);
      SF_FOR_ACC(nn, r, row) X[row * SF_XS + nn * 32 + col] += acc[nn][r] + Ly.bo[nn * 32 + col];
      SF_WAVE_SYNC();
      sf_layernorm(X, Ly.n2g, Ly.n2b, Ly.n2eps);
      SF_WAVE_SYNC();
      // ---- FFN in two 96-wide halves of the hidden layer: T = relu(X W1[:, half] + b1[half]); acc += T W2[half, :]
      sf_f32x16 acf[3];
      SF_GEMM(X, SF_E, Ly.w1a, Ly.pw1a, Ly.pw2a, acc, true);
      SF_FOR_ACC(nn, r, row) T[row * SF_XS + nn * 32 + col] = fmaxf(acc[nn][r] + Ly.b1[nn * 32 + col], 0.0f);
      SF_WAVE_SYNC();
      SF_GEMM(T, SF_E, Ly.w2a, Ly.pw2a, Ly.pw1b, acf, true);
      SF_GEMM(X, SF_E, Ly.w1b, Ly.pw1b, Ly.pw2b, acc, true);
      SF_WAVE_SYNC();  // the first half's A fragments have been read by this wave's MFMAs (same wave: program order) - keep T writes behind
      SF_FOR_ACC(nn, r, row) T[row * SF_XS + nn * 32 + col] = fmaxf(acc[nn][r] + Ly.b1[SF_E + nn * 32 + col], 0.0f);
      SF_WAVE_SYNC();
      SF_GEMM(T, SF_E, Ly.w2b, Ly.pw2b, (l + 1 < prm.num_layers ? prm.layer[l + 1].pwq : prm.pwin), acf, false);
      SF_FOR_ACC(nn, r, row) X[row * SF_XS + nn * 32 + col] += acf[nn][r] + Ly.b2[nn * 32 + col];
      SF_WAVE_SYNC();
      sf_layernorm(X, Ly.n3g, Ly.n3b, Ly.n3eps);
      SF_WAVE_SYNC();
    }
    if (prm.ng) {
      sf_layernorm(X, prm.ng, prm.nb, prm.neps);
      SF_WAVE_SYNC();
    }
    for (int i = lane; i < 32 * (SF_E / 4); i += 64) {
      const int row = i / (SF_E / 4), c4 = i - row * (SF_E / 4);
      const int p = p0 + wave * 32 + row;
      if (p < n) *(float4 *)(out + (size_t)p * out_ld + c4 * 4) = *(const float4 *)(X + row * SF_XS + c4 * 4);
    }
#undef SF_GEMM
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The register-resident form (gemm_products = 6): the decoder TRANSPOSED, activations never leave the registers.
//
// k_sffm_decoder above keeps a wave's 32 points as [32][96] tiles in LDS: every GEMM reads its A fragments from LDS and writes its
// result back through 48 ds_write_b32 per lane, 152 KB of LDS allow one workgroup per CU (one wave per SIMD: every LDS round trip, every
// barrier is exposed) - with the GEMMs on the 3-plane bf16 split the kernel ran no faster than in exact f32: its matrix work is a third
// of its time (round-4 ablations: attention 0.45, FFN 0.62, LayerNorms 0.08 of 1.51 ms per 120k-point frame).
// Here the products are computed transposed, Y^T[channel][point] = W^T X^T:
//   * the weights are the MFMA's A operand (rows = output channels), the activations its B operand (columns = the wave's 32 points).  The
//     C layout of v_mfma_f32_32x32x16_bf16 gives lane (point, kk) the channels 32 n + 8 i + 4 kk + j (n: 32-channel block, register
//     r = 4 i + j) of ITS point - and the B operand of the next product wants, from lane (point, kk), eight K-values per step.  The K index of
//     an MFMA is a free permutation as long as both operands agree: step (n, u) takes registers 8 u .. 8 u + 7 of block n, i.e. channels
//     32 n + 8 (2 u + q / 4) + 4 kk + q % 4, and the weights are packed in that channel order (ops.SffmModel permutes the rows of the plain
//     matrix before ls3d_tile_conv_pack).  A GEMM's output registers ARE the next GEMM's input registers;
//   * bias, residual, ReLU are register arithmetic; LayerNorm is a reduction over a lane's 48 registers plus one shuffle with lane ^ 32;
//   * attention: S^T = K_h q^T with the head's 24 q-values of a point taken straight from the registers (12 per lane: groups 3 h .. 3 h + 2 of
//     eight channels), all L <= 64 class embeddings as two 32-token row blocks on the matrix pipe, softmax over a lane's score registers + one
//     shuffle, O^T = V_h^T P with the probabilities as B operand; the output registers are the out-projection's input registers;
//   * LDS holds only what the four waves share: the weight stream (two 9 KB chunks), K / V of the frame and layer, the layer's bias / LayerNorm
//     vectors: 73 KB -> two workgroups per CU (<= 256 VGPRs), two waves per SIMD to hide each other's waits.
// A workgroup's 128 points belong to ONE frame (tiles are cut per frame from pt_off), so K / V are staged once for all four waves and no wave
// straddles a frame.
constexpr int RT_LMAX = 64;
#ifndef RT_ABLATE_HOOKS
#define RT_ABLATE_HOOKS 0  // 1: a measurement build for tools/bench_decoder.py (gemm_products bits 8.. leave parts of the kernel out)
#endif
#ifndef RT_KSTAGE
#define RT_KSTAGE 24  // channels of K a thread has in flight at the top of a layer (24 = all)
#endif
#ifndef RT_WGS_PER_CU
#define RT_WGS_PER_CU 1
#endif
constexpr int RT_KV = 2 * SF_E * RT_LMAX;   // floats: K [head][token][24] | V [channel][token]
constexpr int RT_VEC = 10 * SF_E;           // floats per layer: bq bo b1[192] b2 n2g n2b n3g n3b (+ 96 spare)
constexpr int RT_LDS_BYTES = 2 * SF_PCHUNK * 16 + RT_KV * 4 + 2 * RT_VEC * 4 + 3 * SF_E * 4;

// four consecutive channels 32 n + 8 i + 4 kk .. + 3 of a per-channel vector in LDS (one broadcast ds_read_b128 per lane half)
#define RT_VEC4(vec_, n_, i_) (*(const float4 *)((vec_) + 32 * (n_) + 8 * (i_) + 4 * kk))
#define RT_FOR(n, i) _Pragma("unroll") for (int n = 0; n < 3; ++n) _Pragma("unroll") for (int i = 0; i < 4; ++i)

// out[m] (+)= W^T[32 m ..][K] in[K][point]:  in = NB blocks of 32 channels in the C layout, weights staged chunk by chunk (16 channels) as in
// sf_gemm_planes of round 4's first attempt: chunk c + 2 in flight while chunk c multiplies, `pre` = chunk 0 of this matrix on entry and of
// `next` on exit.  Six plane products per f32 product, head x head alone in `out`, the five small ones in a second accumulator.
template <int NB>
__device__ __forceinline__ void rt_gemm(const sf_f32x16 (&in)[3], const uint4 *__restrict__ Wq, const uint4 *__restrict__ next, uint4 *Bq,
                                        sf_f32x16 (&out)[3], bool zero, SfPre &pre, int ablate = 0) {
  constexpr int NKC = 2 * NB;
  const int lane = threadIdx.x & 63;
  if (!next) next = Wq;
  const bool stage = !(ablate & 16), mul = !(ablate & 8);
  SfPre p1 = pre;
  if (stage) {
    __syncthreads();  // previous users of the weight buffers are done
    sf_stash(Bq, pre);
    p1 = sf_fetch(NKC > 1 ? Wq + SF_PSTRIDE : next);
    __syncthreads();
  }
  sf_f32x16 acs[3];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acs[m][r] = 0.0f;
      if (zero) out[m][r] = 0.0f;
    }
#pragma unroll
  for (int c = 0; c < NKC; ++c) {
    SfPre p2 = p1;
    if (c + 2 <= NKC && stage) p2 = sf_fetch(c + 2 < NKC ? Wq + (size_t)(c + 2) * SF_PSTRIDE : next);
    if (mul) {
      constexpr int dummy = 0; (void)dummy;
      const int n = c >> 1, u = c & 1;
      uint4 xh, xm, xl;
      ls3d_split_pair3_rne(in[n][8 * u + 0], in[n][8 * u + 1], xh.x, xm.x, xl.x);
      ls3d_split_pair3_rne(in[n][8 * u + 2], in[n][8 * u + 3], xh.y, xm.y, xl.y);
      ls3d_split_pair3_rne(in[n][8 * u + 4], in[n][8 * u + 5], xh.z, xm.z, xl.z);
      ls3d_split_pair3_rne(in[n][8 * u + 6], in[n][8 * u + 7], xh.w, xm.w, xl.w);
      const sf_bf16x8 Xh = __builtin_bit_cast(sf_bf16x8, xh), Xm = __builtin_bit_cast(sf_bf16x8, xm), Xl = __builtin_bit_cast(sf_bf16x8, xl);
      const uint4 *bs = Bq + (c & 1) * SF_PCHUNK + lane;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const sf_bf16x8 Wh = __builtin_bit_cast(sf_bf16x8, bs[(m * 3 + 0) * 64]);
        const sf_bf16x8 Wm = __builtin_bit_cast(sf_bf16x8, bs[(m * 3 + 1) * 64]);
        const sf_bf16x8 Wl = __builtin_bit_cast(sf_bf16x8, bs[(m * 3 + 2) * 64]);
        out[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wh, Xh, out[m], 0, 0, 0);
        acs[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wl, Xh, acs[m], 0, 0, 0);
        acs[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wh, Xl, acs[m], 0, 0, 0);
        acs[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wm, Xm, acs[m], 0, 0, 0);
        acs[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wm, Xh, acs[m], 0, 0, 0);
        acs[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wh, Xm, acs[m], 0, 0, 0);
      }
    }
    if (c + 1 < NKC && stage) {
      sf_stash(Bq + ((c + 1) & 1) * SF_PCHUNK, p1);
      __syncthreads();
    }
    p1 = p2;
  }
  pre = p1;  // chunk 0 of `next`
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[m][r] += acs[m][r];
}

// x = LayerNorm(x) over the 96 channels of a point: 48 registers here, 48 in lane ^ 32
__device__ __forceinline__ void rt_layernorm(sf_f32x16 (&x)[3], const float *g, const float *b, float eps, int kk) {
  float s = 0.0f;
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += x[n][r];
  s += __shfl_xor(s, 32);
  const float mean = s / (float)SF_E;
  float q2 = 0.0f;
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) { x[n][r] -= mean; q2 = fmaf(x[n][r], x[n][r], q2); }
  q2 += __shfl_xor(q2, 32);
  const float rstd = 1.0f / sqrtf(q2 / (float)SF_E + eps);
  RT_FOR(n, i) {
    const float4 gg = RT_VEC4(g, n, i), bb = RT_VEC4(b, n, i);
    x[n][4 * i + 0] = x[n][4 * i + 0] * rstd * gg.x + bb.x;
    x[n][4 * i + 1] = x[n][4 * i + 1] * rstd * gg.y + bb.y;
    x[n][4 * i + 2] = x[n][4 * i + 2] * rstd * gg.z + bb.z;
    x[n][4 * i + 3] = x[n][4 * i + 3] * rstd * gg.w + bb.w;
  }
}

// attention of the wave's 32 points over the frame's L <= 64 class embeddings, one head: q = t's registers of groups 3 h .. 3 h + 2 (overwritten
// by the head's output).  kvs in LDS: K as [head][token][24] (a token's 4 consecutive channels are one ds_read_b128), V as [channel][token] (the layout of `kv`).  Exact f32 products (v_mfma_f32_32x32x2_f32), f32 softmax.
// exp of a non-positive argument for the softmax: v_exp_f32(x log2 e).  The product loses |x| 2^-24 absolutely, i.e. a probability e^x is
// off by |x| e^x 6e-8 relative to the largest one: < 2.2e-8 everywhere (expf: 13 more instructions per value, 768 values per point).
#ifdef HIPSIM
#define RT_EXP(x_) expf(x_)
#else
#define RT_EXP(x_) __expf(x_)
#endif
// NS2: the number of second-block steps / registers that hold tokens below L, known at compile time (register r and P V step r of the second block
// cover tokens 32 + (r & 3) + 8 (r >> 2) and that + 4, ascending in r) - 0: L <= 32, 2: L = 33 .. 34 (nuScenes: 2 x 17 classes), -1: decided at
// run time with wave-uniform branches.  With it the four heads of a layer are ONE basic block, which hipcc schedules across (the next head's
// K reads and MFMAs under this head's softmax).
template <int H, int NS2>
__device__ __forceinline__ void rt_attention_head(sf_f32x16 (&t)[3], const float *kvs, int L) {
  const int lane = threadIdx.x & 63, col = lane & 31, kk = lane >> 5;
  const float scale = 1.0f / sqrtf((float)SF_HD);
  const bool two = NS2 < 0 ? L > 32 : NS2 > 0;
#define RT_IN2(r_) (NS2 < 0 ? (32 + ((r_) & 3) + 8 * ((r_) >> 2) < L) : ((r_) < NS2))
  const float *Kh = kvs + H * L * SF_HD, *Vh = kvs + SF_E * L + H * SF_HD * L;  // K: [head][token][24]; V: [channel][token] (the layout of `kv`)
  const int tok0 = col < L ? col : 0, tok1 = (32 + col) < L ? 32 + col : 0;  // this lane's key rows of the two A operands (rows >= L are masked below)
  sf_f32x16 s0, s1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { s0[r] = 0.0f; s1[r] = 0.0f; }
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    constexpr int dummy = 0; (void)dummy;
    const int G = 3 * H + g, n = G >> 2, i = G & 3;
    const float4 k0 = *(const float4 *)(Kh + tok0 * SF_HD + 8 * g + 4 * kk);
    const float kv0[4] = {k0.x, k0.y, k0.z, k0.w};
    if (two) {  // the two key blocks' products alternate: two independent accumulator chains
      const float4 k1 = *(const float4 *)(Kh + tok1 * SF_HD + 8 * g + 4 * kk);
      const float kv1[4] = {k1.x, k1.y, k1.z, k1.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kv0[j], t[n][4 * i + j], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kv1[j], t[n][4 * i + j], s1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) s0 = __builtin_amdgcn_mfma_f32_32x32x2f32(kv0[j], t[n][4 * i + j], s0, 0, 0, 0);
    }
  }
  float m = -3.0e38f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int token = (r & 3) + 8 * (r >> 2) + 4 * kk;
    s0[r] = token < L ? s0[r] * scale : -3.0e38f;
    m = fmaxf(m, s0[r]);
    if (two && RT_IN2(r)) {
      s1[r] = token + 32 < L ? s1[r] * scale : -3.0e38f;
      m = fmaxf(m, s1[r]);
    }
  }
  m = fmaxf(m, __shfl_xor(m, 32));
  float den = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int token = (r & 3) + 8 * (r >> 2) + 4 * kk;
    s0[r] = token < L ? RT_EXP(s0[r] - m) : 0.0f;
    den += s0[r];
    if (two && RT_IN2(r)) {  // (L = 34 evaluates 2 of the second block's 16 registers)
      s1[r] = token + 32 < L ? RT_EXP(s1[r] - m) : 0.0f;
      den += s1[r];
    }
  }
  den += __shfl_xor(den, 32);
  // O^T[d][point] = sum_token V_h[token][d] P[token][point]: A = V_h^T (row d: this lane's half of the step's two tokens), B = the probabilities.
  // Two accumulator chains (steps 0 - 7 | steps 8 - 15 and the second block), added at the end.
  sf_f32x16 oc, od;
#pragma unroll
  for (int r = 0; r < 16; ++r) { oc[r] = 0.0f; od[r] = 0.0f; }
  const int dv = col < SF_HD ? col : 0;  // rows >= 24 of the output are discarded
#pragma unroll
  for (int s2 = 0; s2 < 8; ++s2) {
    const int ta = (s2 & 3) + 8 * (s2 >> 2) + 4 * kk, tb = ta + 16;
    const float va = ta < L ? Vh[dv * L + ta] : 0.0f, vb = tb < L ? Vh[dv * L + tb] : 0.0f;
    oc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s0[s2], oc, 0, 0, 0);
    od = __builtin_amdgcn_mfma_f32_32x32x2f32(vb, s0[s2 + 8], od, 0, 0, 0);
  }
  if (two) {
#pragma unroll
    for (int s2 = 0; s2 < 16; ++s2) {
      // steps whose tokens are all >= L add zeros (L = 34: 14 of the 16)
      if (RT_IN2(s2)) {
        const int token = 32 + (s2 & 3) + 8 * (s2 >> 2) + 4 * kk;
        const float va = token < L ? Vh[dv * L + token] : 0.0f;
        if (s2 & 1) od = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s1[s2], od, 0, 0, 0);
        else oc = __builtin_amdgcn_mfma_f32_32x32x2f32(va, s1[s2], oc, 0, 0, 0);
      }
    }
  }
#undef RT_IN2
#pragma unroll
  for (int r = 0; r < 12; ++r) oc[r] += od[r];
  const float inv = 1.0f / den;
#pragma unroll
  for (int r = 0; r < 12; ++r) {  // output row d = (r & 3) + 8 (r >> 2) + 4 kk of the head = channel 24 h + d: group 3 h + (r >> 2), j = r & 3
    constexpr int dummy = 0; (void)dummy;
    const int G = 3 * H + (r >> 2), n = G >> 2, i = G & 3;
    t[n][4 * i + (r & 3)] = oc[r] * inv;
  }
}

template <int NB, int NS2>
__global__ __launch_bounds__(256, RT_WGS_PER_CU) void k_sffm_decoder_rt(const float *__restrict__ x, int x_ld, int n, const int32_t *__restrict__ pt_off,
                                                            const float *__restrict__ kv, int L, int batch, SfParams prm,
                                                            float *__restrict__ out, int out_ld) {
  HIP_DYNAMIC_SHARED(float, smem)
  uint4 *Bq = (uint4 *)smem;                                   // [2][SF_PCHUNK] weight chunks
  float *KVs = smem + 2 * SF_PCHUNK * 4;                       // K [head][token][24] | V [channel][token]
  float *VEC = KVs + RT_KV;                                    // [2][RT_VEC] the layer's per-channel vectors (double buffered over the layers)
  float *GV = VEC + 2 * RT_VEC;                                // bin | ng | nb
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  const int ab = RT_ABLATE_HOOKS ? prm.ablate : 0;  // 0 at compile time in the product build: the branches below disappear
  // ---- this workgroup's tile: 128 consecutive points of ONE frame
  int f = 0, tile = (int)blockIdx.x, lo = 0, hi = 0;
  for (;; ++f) {
    if (f >= batch) return;  // beyond the tiles of the batch (the grid covers the worst case: ceil(n / 128) + batch)
    lo = pt_off[f]; hi = pt_off[f + 1];
    const int nt = (hi - lo + 127) >> 7;
    if (tile < nt) break;
    tile -= nt;
  }
  const int p = lo + tile * 128 + wave * 32 + col;  // this lane's point (both halves kk of a column hold the same point)
  const bool live = p < hi;
  const size_t kv_layer = (size_t)2 * batch * SF_E * L;
  for (int i = tid; i < SF_E; i += 256) {
    GV[i] = prm.bin[i];
    GV[SF_E + i] = prm.ng ? prm.ng[i] : 1.0f;
    GV[2 * SF_E + i] = prm.nb ? prm.nb[i] : 0.0f;
  }
  SfPre pre = sf_fetch(prm.pwin);
  // ---- the point's input row -> registers in the C layout (channels 32 n + 8 i + 4 kk + j)
  sf_f32x16 xin[3], xs[3], t[3], acc[3];
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) xin[nb][r] = 0.0f;
  {
    const float *row = x + (size_t)(live ? p : lo) * x_ld;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = *(const float4 *)(row + 32 * nb + 8 * i + 4 * kk);
        xin[nb][4 * i + 0] = v.x; xin[nb][4 * i + 1] = v.y; xin[nb][4 * i + 2] = v.z; xin[nb][4 * i + 3] = v.w;
      }
  }
  // ---- input projection -> xs
  rt_gemm<NB>(xin, prm.pwin, prm.num_layers ? prm.layer[0].pwq : nullptr, Bq, xs, true, pre, ab);
  RT_FOR(nn, i) {
    const float4 b4 = RT_VEC4(GV, nn, i);
    xs[nn][4 * i + 0] += b4.x; xs[nn][4 * i + 1] += b4.y; xs[nn][4 * i + 2] += b4.z; xs[nn][4 * i + 3] += b4.w;
  }
  for (int l = 0; l < prm.num_layers; ++l) {
    const SfLayer &Ly = prm.layer[l];
    float *V = VEC + (l & 1) * RT_VEC;
    // ---- what the four waves share in this layer: the per-channel vectors and K / V of the frame (as they lie in `kv`: [channel][token]).
    //      No barrier of its own: every wave is past the previous layer's attention and LayerNorms of layer l - 2 (the GEMMs in
    //      between synchronise the workgroup), and the q-projection's first barrier orders these writes against their readers.
    for (int i = tid; i < SF_E; i += 256) {
      V[i] = Ly.bq[i]; V[SF_E + i] = Ly.bo[i]; V[2 * SF_E + i] = Ly.b1[i]; V[3 * SF_E + i] = Ly.b1[SF_E + i]; V[4 * SF_E + i] = Ly.b2[i];
      V[5 * SF_E + i] = Ly.n2g[i]; V[6 * SF_E + i] = Ly.n2b[i]; V[7 * SF_E + i] = Ly.n3g[i]; V[8 * SF_E + i] = Ly.n3b[i];
    }
    {
      // K: [channel][token] -> [head][token][24], one token per lane and 24 channels per thread: 24 loads in flight, no division by L.
      // V: a straight copy in 16-byte units (96 L floats, 96 * 4 bytes a multiple of 16).  Each matrix's loads are issued before its first LDS
      // write: two memory latencies per layer.  (The loop this replaces transposed both with two runtime divisions and a dependent
      // load -> ds_write per element, 25 trips: a latency per trip.)
      const float *kg = kv + (size_t)l * kv_layer + (size_t)f * SF_E * L, *vg = kg + (size_t)batch * SF_E * L;
      {
        int tok = tid & 63, c0 = tid >> 6;
#ifndef HIPSIM
        asm volatile("" : "+v"(tok), "+v"(c0));  // not loop-invariant for the compiler: 48 hoisted addresses would live (and spill) across the whole layer
#endif
        const int tk = tok < L ? tok : 0;
#pragma unroll
        for (int jb = 0; jb < 24; jb += RT_KSTAGE) {
          float q[RT_KSTAGE];
#pragma unroll
          for (int j = 0; j < RT_KSTAGE; ++j) q[j] = kg[(c0 + 4 * (jb + j)) * L + tk];
          if (tok < L) {
#pragma unroll
            for (int j = 0; j < RT_KSTAGE; ++j) {
              const int c = c0 + 4 * (jb + j), h = c / SF_HD, d = c - h * SF_HD;
              KVs[(h * L + tok) * SF_HD + d] = q[j];
            }
          }
          LS3D_SCHED_FENCE();  // keeps the next batch's loads behind this batch's stores (all 24 in flight at once spill)
        }
      }
      {
        const int units = (SF_E / 4) * L;  // <= 1536
        const float4 *src = (const float4 *)vg;
        float4 *dst = (float4 *)(KVs + SF_E * L);
        float4 q[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int i = tid + 256 * j;
          q[j] = src[i < units ? i : 0];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          const int i = tid + 256 * j;
          if (i < units) dst[i] = q[j];
        }
      }
    }
    // ---- q projection -> t
    rt_gemm<3>(xs, Ly.pwq, Ly.pwo, Bq, t, true, pre, ab);
    RT_FOR(nn, i) {
      const float4 b4 = RT_VEC4(V, nn, i);
      t[nn][4 * i + 0] += b4.x; t[nn][4 * i + 1] += b4.y; t[nn][4 * i + 2] += b4.z; t[nn][4 * i + 3] += b4.w;
    }
    if (!(ab & 1)) {
      rt_attention_head<0, NS2>(t, KVs, L);
      rt_attention_head<1, NS2>(t, KVs, L);
      rt_attention_head<2, NS2>(t, KVs, L);
      rt_attention_head<3, NS2>(t, KVs, L);
    }
    // ---- out projection + residual, LayerNorm (norm2)
    rt_gemm<3>(t, Ly.pwo, Ly.pw1a, Bq, acc, true, pre, ab);
    RT_FOR(nn, i) {
      const float4 b4 = RT_VEC4(V + SF_E, nn, i);
      xs[nn][4 * i + 0] += acc[nn][4 * i + 0] + b4.x; xs[nn][4 * i + 1] += acc[nn][4 * i + 1] + b4.y;
      xs[nn][4 * i + 2] += acc[nn][4 * i + 2] + b4.z; xs[nn][4 * i + 3] += acc[nn][4 * i + 3] + b4.w;
    }
    if (!(ab & 4)) rt_layernorm(xs, V + 5 * SF_E, V + 6 * SF_E, Ly.n2eps, kk);
    if (!(ab & 2)) {
    // ---- FFN in two 96-wide halves of the hidden layer: t = relu(W1[half]^T x + b1[half]); acc += W2[half]^T t
    rt_gemm<3>(xs, Ly.pw1a, Ly.pw2a, Bq, t, true, pre, ab);
    RT_FOR(nn, i) {
      const float4 b4 = RT_VEC4(V + 2 * SF_E, nn, i);
      t[nn][4 * i + 0] = fmaxf(t[nn][4 * i + 0] + b4.x, 0.0f); t[nn][4 * i + 1] = fmaxf(t[nn][4 * i + 1] + b4.y, 0.0f);
      t[nn][4 * i + 2] = fmaxf(t[nn][4 * i + 2] + b4.z, 0.0f); t[nn][4 * i + 3] = fmaxf(t[nn][4 * i + 3] + b4.w, 0.0f);
    }
    rt_gemm<3>(t, Ly.pw2a, Ly.pw1b, Bq, acc, true, pre, ab);
    rt_gemm<3>(xs, Ly.pw1b, Ly.pw2b, Bq, t, true, pre, ab);
    RT_FOR(nn, i) {
      const float4 b4 = RT_VEC4(V + 3 * SF_E, nn, i);
      t[nn][4 * i + 0] = fmaxf(t[nn][4 * i + 0] + b4.x, 0.0f); t[nn][4 * i + 1] = fmaxf(t[nn][4 * i + 1] + b4.y, 0.0f);
      t[nn][4 * i + 2] = fmaxf(t[nn][4 * i + 2] + b4.z, 0.0f); t[nn][4 * i + 3] = fmaxf(t[nn][4 * i + 3] + b4.w, 0.0f);
    }
    rt_gemm<3>(t, Ly.pw2b, (l + 1 < prm.num_layers ? prm.layer[l + 1].pwq : nullptr), Bq, acc, false, pre, ab);
    RT_FOR(nn, i) {
      const float4 b4 = RT_VEC4(V + 4 * SF_E, nn, i);
      xs[nn][4 * i + 0] += acc[nn][4 * i + 0] + b4.x; xs[nn][4 * i + 1] += acc[nn][4 * i + 1] + b4.y;
      xs[nn][4 * i + 2] += acc[nn][4 * i + 2] + b4.z; xs[nn][4 * i + 3] += acc[nn][4 * i + 3] + b4.w;
    }
    }
    if (!(ab & 4)) rt_layernorm(xs, V + 7 * SF_E, V + 8 * SF_E, Ly.n3eps, kk);
  }
  if (prm.ng) rt_layernorm(xs, GV + SF_E, GV + 2 * SF_E, prm.neps, kk);
  if (live) {
    float *row = out + (size_t)p * out_ld;
    RT_FOR(nn, i) *(float4 *)(row + 32 * nn + 8 * i + 4 * kk) = make_float4(xs[nn][4 * i + 0], xs[nn][4 * i + 1], xs[nn][4 * i + 2], xs[nn][4 * i + 3]);
  }
}

extern "C" int ls3d_sffm_decoder(const float *x, int x_ld, int n, const float *points, int pt_stride, const float *kv, int L, int batch,
                                 const ls3d_sffm_t *m, float *out, int out_ld, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!x || !points || !kv || !m || !out || n < 0 || batch < 1 || pt_stride < 1) return LS3D_ERR_ARG;
  if (m->d_model != SF_E || m->heads != SF_H || m->ffn != 2 * SF_E || L < 1 || L > 64 || m->num_layers < 0 || m->num_layers > SF_MAX_LAYERS ||
      m->d_in < 32 || m->d_in > SF_E || (m->d_in % 32))
    return LS3D_ERR_UNSUPPORTED;  // the caller composes the layer from ls3d_gather_gemm / ls3d_cross_attn / ls3d_layernorm
  if ((x_ld % 4) || x_ld < m->d_in || (out_ld % 4) || out_ld < SF_E || ((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return LS3D_ERR_ARG;
  if (!m->w_in || !m->b_in || (m->num_layers > 0 && !m->layers)) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  SfParams prm;
  prm.win = m->w_in; prm.bin = m->b_in; prm.ng = m->norm_gamma; prm.nb = m->norm_beta; prm.neps = m->norm_eps;
  prm.num_layers = m->num_layers; prm.d_in = m->d_in;
  const bool planes = (m->gemm_products & 255) == 6;
  if ((m->gemm_products & 255) != 0 && !planes) return LS3D_ERR_ARG;
  prm.ablate = m->gemm_products >> 8;
  prm.pwin = (const uint4 *)m->w_in_planes;
  if (planes && (!m->w_in_planes || !m->pt_off || (m->d_in % 32) || ((uintptr_t)kv & 15))) return LS3D_ERR_ARG;
  if (planes && (m->attention != 0 || L > RT_LMAX)) return LS3D_ERR_UNSUPPORTED;  // the reduced-precision attentions live in the LDS-tile kernel
  for (int l = 0; l < m->num_layers; ++l) {
    const ls3d_sffm_layer_t &s = m->layers[l];
    if (!s.wq || !s.bq || !s.wo || !s.bo || !s.w1a || !s.w1b || !s.b1 || !s.w2a || !s.w2b || !s.b2 || !s.n2_gamma || !s.n2_beta || !s.n3_gamma || !s.n3_beta)
      return LS3D_ERR_ARG;
    if (planes && (!s.wq_planes || !s.wo_planes || !s.w1a_planes || !s.w1b_planes || !s.w2a_planes || !s.w2b_planes)) return LS3D_ERR_ARG;
    prm.layer[l] = SfLayer{s.wq, s.bq, s.wo, s.bo, s.w1a, s.w1b, s.b1, s.w2a, s.w2b, s.b2, s.n2_gamma, s.n2_beta, s.n3_gamma, s.n3_beta, s.n2_eps, s.n3_eps,
                           (const uint4 *)s.wq_planes, (const uint4 *)s.wo_planes, (const uint4 *)s.w1a_planes, (const uint4 *)s.w1b_planes,
                           (const uint4 *)s.w2a_planes, (const uint4 *)s.w2b_planes};
  }
  const int lds = (4 * SF_WAVE_FLOATS + 2 * SF_BCHUNK + SF_KV) * (int)sizeof(float) + (128 + 8) * (int)sizeof(int);
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_sffm_decoder, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder_rt<1, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder_rt<2, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder_rt<3, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder_rt<2, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder_rt<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, RT_LDS_BYTES) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  long long blocks = ((long long)n + 127) / 128;
  if (blocks > 65536) blocks = 65536;
  const int att = (m->attention >= 0 && m->attention <= 3) ? m->attention : 0;
  if (planes) {  // the register-resident form: tiles cut per frame
    const unsigned grid = (unsigned)(((long long)n + 127) / 128 + batch);
    const int32_t *po = (const int32_t *)m->pt_off;
#define RT_LAUNCH(NB_, NS2_) hipLaunchKernelGGL((k_sffm_decoder_rt<NB_, NS2_>), dim3(grid), dim3(256), RT_LDS_BYTES, stream, x, x_ld, n, po, kv, L, batch, prm, out, out_ld)
    // the shipped heads (d_in = 64) with up to 32 or with 33 - 34 class embeddings (nuScenes: 2 x 17) have the attention's token bounds compiled in
    if (m->d_in == 64 && L <= 32) RT_LAUNCH(2, 0);
    else if (m->d_in == 64 && L <= 34) RT_LAUNCH(2, 2);
    else if (m->d_in == 32) RT_LAUNCH(1, -1); else if (m->d_in == 64) RT_LAUNCH(2, -1); else RT_LAUNCH(3, -1);
#undef RT_LAUNCH
  } else {
    hipLaunchKernelGGL(k_sffm_decoder, dim3((unsigned)blocks), dim3(256), lds, stream, x, x_ld, n, points, pt_stride, kv, L, batch, prm, out, out_ld, att);
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
