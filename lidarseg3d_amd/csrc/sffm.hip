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
// `pre` comes in holding chunk 0 of THIS matrix (fetched while the previous GEMM / attention / LayerNorm ran: the weights do not depend on the
// data) and goes out holding chunk 0 of `next`; inside, chunk c + 2 is fetched while chunk c multiplies: a chunk's MFMAs (18 x 32 cycles) are
// shorter than an L2 round trip, one chunk ahead left every barrier waiting for memory (measured: slower than the f32 form).
__device__ __forceinline__ void sf_gemm_planes(const float *A, int lda, int K, const uint4 *__restrict__ Wq, const uint4 *__restrict__ next, float *Bs,
                                               sf_f32x16 (&acc)[3], bool zero, SfPre &pre) {
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, kk = lane >> 5;
  const int nkc = K / 16;
  uint4 *Bq = (uint4 *)Bs;
  if (!next) next = Wq;
  __syncthreads();  // previous users of Bs are done
  sf_stash(Bq, pre);
  SfPre p1 = sf_fetch(nkc > 1 ? Wq + SF_PSTRIDE : next);
  __syncthreads();
  sf_f32x16 acs[3];
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acs[n][r] = 0.0f;
      if (zero) acc[n][r] = 0.0f;
    }
  int buf = 0;
  for (int c = 0; c < nkc; ++c) {
    SfPre p2 = p1;
    if (c + 2 <= nkc) p2 = sf_fetch(c + 2 < nkc ? Wq + (size_t)(c + 2) * SF_PSTRIDE : next);
    {
      const float4 *ap = (const float4 *)(A + col * lda + c * 16 + kk * 8);
      const float4 a0 = ap[0], a1 = ap[1];
      uint4 ah, am, al;
      ls3d_split_pair3_rne(a0.x, a0.y, ah.x, am.x, al.x);
      ls3d_split_pair3_rne(a0.z, a0.w, ah.y, am.y, al.y);
      ls3d_split_pair3_rne(a1.x, a1.y, ah.z, am.z, al.z);
      ls3d_split_pair3_rne(a1.z, a1.w, ah.w, am.w, al.w);
      const sf_bf16x8 Ah = __builtin_bit_cast(sf_bf16x8, ah), Am = __builtin_bit_cast(sf_bf16x8, am), Al = __builtin_bit_cast(sf_bf16x8, al);
      const uint4 *bs = Bq + buf * SF_PCHUNK + lane;
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        const sf_bf16x8 Bh = __builtin_bit_cast(sf_bf16x8, bs[(n * 3 + 0) * 64]);
        const sf_bf16x8 Bm = __builtin_bit_cast(sf_bf16x8, bs[(n * 3 + 1) * 64]);
        const sf_bf16x8 Bl = __builtin_bit_cast(sf_bf16x8, bs[(n * 3 + 2) * 64]);
        acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[n], 0, 0, 0);
        acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acs[n], 0, 0, 0);
        acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acs[n], 0, 0, 0);
        acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acs[n], 0, 0, 0);
        acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acs[n], 0, 0, 0);
        acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acs[n], 0, 0, 0);
      }
    }
    if (c + 1 < nkc) {
      sf_stash(Bq + (buf ^ 1) * SF_PCHUNK, p1);
      __syncthreads();
      buf ^= 1;
    }
    p1 = p2;
  }
  pre = p1;  // chunk 0 of `next`
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] += acs[n][r];
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

// one workgroup = 128 consecutive points (4 waves x 32).  GP = 0: the GEMMs on exact-f32 MFMA; GP = 6: on the 3-plane bf16 split (six products)
template <int GP>
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
    if (tid < 128) s_frame[tid] = (p0 + tid < n) ? (int)points[(size_t)(p0 + tid) * pt_stride] : -1;
    // ---- the wave's 32 input rows -> T[:, 0:d_in]
    for (int i = lane; i < 32 * (prm.d_in / 4); i += 64) {
      const int row = i / (prm.d_in / 4), c4 = i - row * (prm.d_in / 4);
      const int p = p0 + wave * 32 + row;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < n) v = *(const float4 *)(x + (size_t)p * x_ld + c4 * 4);
      *(float4 *)(T + row * SF_XS + c4 * 4) = v;
    }
    __syncthreads();
    const int fs = s_frame[0];  // the frame whose K / V are staged (the tile's first point)
    sf_f32x16 acc[3];
    SfPre pre;
    if constexpr (GP == 6) pre = sf_fetch(prm.pwin);  // chunk 0 of the first matrix: in flight while the tile's rows land
#define SF_GEMM(A_, K_, Wf_, Wp_, Wnext_, acc_, zero_)                                        \
  do {                                                                                        \
    if constexpr (GP == 6) sf_gemm_planes(A_, SF_XS, K_, Wp_, Wnext_, Bs, acc_, zero_, pre);   \
    else sf_gemm(A_, SF_XS, K_, Wf_, Bs, acc_, zero_);                                        \
  } while (0)
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
  const bool planes = m->gemm_products == 6;
  if (m->gemm_products != 0 && !planes) return LS3D_ERR_ARG;
  prm.pwin = (const uint4 *)m->w_in_planes;
  if (planes && (!m->w_in_planes || (m->d_in % 16))) return LS3D_ERR_ARG;
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
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_sffm_decoder<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess ||
        hipFuncSetAttribute((const void *)k_sffm_decoder<6>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  long long blocks = ((long long)n + 127) / 128;
  if (blocks > 65536) blocks = 65536;
  const int att = (m->attention >= 0 && m->attention <= 3) ? m->attention : 0;
  if (planes)
    hipLaunchKernelGGL(k_sffm_decoder<6>, dim3((unsigned)blocks), dim3(256), lds, stream, x, x_ld, n, points, pt_stride, kv, L, batch, prm, out, out_ld, att);
  else
    hipLaunchKernelGGL(k_sffm_decoder<0>, dim3((unsigned)blocks), dim3(256), lds, stream, x, x_ld, n, points, pt_stride, kv, L, batch, prm, out, out_ld, att);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
