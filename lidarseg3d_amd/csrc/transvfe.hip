// transvfe.hip — TransformerVoxelFeatureExtractor as ONE kernel.
//
// Reference: det3d/models/readers/voxel_encoder.py:202-270 (TransformerVoxelFeatureExtractor.forward) with
// TransformerEncoderLayerPreNorm (:149-163, the residual is taken from the NORMED tensor) iterated by hand.  Per voxel:
// its P point slots are tokens [point feats | descriptor]; Conv1d(k=1) embedding -> num_layers x {LayerNorm, 4-head
// self-attention over the P tokens (no padding mask), out-proj + residual, LayerNorm, FF(ReLU) + residual} -> max over
// the tokens -> Linear + ReLU compression.
//
// The layer-by-layer version (ls3d_vfe_tokens -> ls3d_gather_gemm x (1 + 4 per layer) -> ls3d_mha_core -> ls3d_group_max
// -> ls3d_gather_gemm) streams the [V*P, 64..192] token matrices through HBM a dozen times (1.35 ms of a 13.6 ms frame on
// MI355X).  Here a wave owns G = 32 / P voxels = one 32-row MFMA tile of tokens and keeps it in LDS from the raw points
// to the 16-float voxel feature; only the weights (staged per workgroup in LDS, shared by the 4 waves) and the points are
// read, and V x 16 floats are written.
//   * GEMMs: v_mfma_f32_32x32x2_f32 (exact f32), A fragments from the wave's LDS tile (row stride K + 4 floats:
//     conflict-free ds_read_b128, the MFMA K index is permuted so that a lane reads 16 contiguous floats), B chunks of
//     32 x 64 weights in the packed layout of ls3d_gather_gemm_pack(nt = 2), double buffered in LDS;
//   * attention: one lane per (voxel, head, query token), the result overwrites the query's own slice of the QKV tile;
//   * LayerNorm: two lanes per row, statistics by one shuffle.
// LDS: 4 waves x (32 x 68 + 32 x 196) floats + 2 x 8 KB weight chunks = 151 KB -> one workgroup per CU; the kernel is
// bound by the f32 matrix pipe (1568 MFMAs per 32-token tile for 3 layers; measured 1.08 ms for 65.9k voxels = 45 % of
// the matrix-pipe bound, vs 1.75 ms for the layer-by-layer version).
#include "common.h"
#include "gemm_common.h"
#include "vfe_descriptor.h"

typedef float tv_f32x16 __attribute__((ext_vector_type(16)));

#define TV_MAX_LAYERS 4
struct TvLayer {
  const float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2, *n1g, *n1b, *n2g, *n2b;
  float n1eps, n2eps;
};
struct TvParams {
  const float *we, *be;          // embedding, packed nt=2, K = KT
  const float *wc, *bc;          // compression: PLAIN nn.Linear weight [ncomp][E] and bias, or null
  TvLayer layer[TV_MAX_LAYERS];
  int num_layers, ncomp;
};

#ifdef HIPSIM
#define TV_WAVE_SYNC() hipsim::wave_barrier()
#else
#define TV_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
#endif

constexpr int TV_E = 64, TV_FF = 128, TV_HD = 16, TV_H = 4, TV_KT = 32;
constexpr int TV_XS = TV_E + 4;          // row stride of the activation tile X
constexpr int TV_TS = 3 * TV_E + 4;      // row stride of the scratch tile T (QKV / FF hidden / tokens)
constexpr int TV_WAVE_FLOATS = 32 * TV_XS + 32 * TV_TS;
constexpr int TV_BCHUNK = 32 * 64;       // floats in one staged weight chunk

// out(32 x N) = A(32 x K, LDS, stride lda) x W (packed nt=2: [slab][K][32][2]), handed to `epi(slab, acc0, acc1)` per
// 64-column slab.  All 4 waves of the workgroup call this together (the weight chunks are staged by the workgroup).
// DIRECT: no LDS staging and no workgroup barrier - every wave reads its B fragments (float2 per lane, 512 contiguous bytes per
// MFMA pair) straight from the L2-resident packed weights, one chunk ahead in registers; the 4 waves of a workgroup then never wait
// for each other.  Experimental (ls3d_set_transvfe_direct), same arithmetic in the same order.
// MODE 0: f32 MFMA, weights staged through LDS.  MODE 1: f32 MFMA, weights straight from L2 (experimental).  MODE 6 / 8: the exact
// 3-plane bf16 split of both operands - Wp then points at the plane-packed weights of ls3d_transvfe_pack_planes
// (per chunk [column block 2][K step 2][plane 3][kk 2][col 32] x 8 bf16 = 12 KB), the A fragment (the same 16 contiguous floats per
// lane) is split in registers, 6 / 8 v_mfma_f32_32x32x16_bf16 per (column block, K step) with head x head in its own accumulator:
// 24 / 32 MFMAs of 32 cycles per chunk instead of 32 of 64.
constexpr int TV_PCHUNK = 768;  // uint4 per plane-packed chunk
// one 12 KB plane-packed weight chunk in flight (3 x 16 bytes per thread).  MODE 6 / 8: a chunk's MFMAs (24 / 32 x 32 cycles) are shorter than
// an L2 round trip, so chunk c + 2 is fetched while chunk c multiplies, and the first chunk of the NEXT matrix while this GEMM's last chunk
// and the attention / LayerNorm behind it run (`pre`: in = chunk 0 of this matrix, out = chunk 0 of `next`; the weights do not depend on the data).
struct TvPre { uint4 a, b, c; };
__device__ __forceinline__ TvPre tv_fetch(const uint4 *__restrict__ chunk) {
  const int tid = threadIdx.x;
  TvPre p;
  p.a = chunk[tid]; p.b = chunk[tid + 256]; p.c = chunk[tid + 512];
  return p;
}
// K, N are template parameters: in the plane modes the chunk loop is fully unrolled - a kernel with one wave per SIMD has nobody to hide
// a branch behind, and every per-chunk condition (first / last chunk of a slab, anything left to fetch) splits the MFMAs, the weight
// prefetch and the LDS stash into basic blocks that hipcc does not schedule across.
// bias: the GEMM's per-column bias [N]; a lane's two values per 64-column slab are loaded HERE, in front of the chunk loop, and handed to the epilogue:
// loaded inside the epilogue (round 5) they were two global loads per slab with ~250 cycles of cover - 22 exposed L2 round trips per tile in a kernel
// that runs one wave per SIMD (round 6, `tools/asm_waits.py`-style reading of the disassembly: `G2 v62 s_waitcnt vmcnt(0)` behind every slab's last MFMA).
template <int MODE, int K, int N, typename Epi>
__device__ __forceinline__ void tv_gemm(const float *A, int lda, const float *__restrict__ Wp, float *Bs, const float *__restrict__ bias, Epi epi, TvPre &pre,
                                        const float *__restrict__ next) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int col = lane & 31, kk = lane >> 5;
  constexpr int nslab = N / 64, nkc = K / 32, nchunks = nslab * nkc;
  float bz0[nslab], bz1[nslab];
#pragma unroll
  for (int sb = 0; sb < nslab; ++sb) { bz0[sb] = bias[sb * 64 + col]; bz1[sb] = bias[sb * 64 + 32 + col]; }
  if constexpr (MODE >= 6) {
    const uint4 *Wq = (const uint4 *)Wp, *Wn = next ? (const uint4 *)next : (const uint4 *)Wp;
    uint4 *Bq = (uint4 *)Bs;
    __syncthreads();  // previous users of Bs are done
    Bq[tid] = pre.a; Bq[tid + 256] = pre.b; Bq[tid + 512] = pre.c;
    TvPre p1 = tv_fetch(nchunks > 1 ? Wq + TV_PCHUNK : Wn);
    __syncthreads();
    tv_f32x16 acc0, acc1, acs0, acs1;
    int buf = 0;
#pragma unroll
    for (int c = 0; c < nchunks; ++c) {
      const int slab = c / nkc, kc = c - slab * nkc;
      if (kc == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; acs0[r] = 0.0f; acs1[r] = 0.0f; }
      }
      TvPre p2 = p1;
      if (c + 2 <= nchunks) p2 = tv_fetch(c + 2 < nchunks ? Wq + (size_t)(c + 2) * TV_PCHUNK : Wn);
      {
        const float4 *ap = (const float4 *)(A + col * lda + kc * 32 + kk * 16);
        const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
        uint4 ah[2], am[2], al[2];  // [K step]: floats [8 s, 8 s + 8) of the lane's 16
        ls3d_split_pair3_rne(a0.x, a0.y, ah[0].x, am[0].x, al[0].x);
        ls3d_split_pair3_rne(a0.z, a0.w, ah[0].y, am[0].y, al[0].y);
        ls3d_split_pair3_rne(a1.x, a1.y, ah[0].z, am[0].z, al[0].z);
        ls3d_split_pair3_rne(a1.z, a1.w, ah[0].w, am[0].w, al[0].w);
        ls3d_split_pair3_rne(a2.x, a2.y, ah[1].x, am[1].x, al[1].x);
        ls3d_split_pair3_rne(a2.z, a2.w, ah[1].y, am[1].y, al[1].y);
        ls3d_split_pair3_rne(a3.x, a3.y, ah[1].z, am[1].z, al[1].z);
        ls3d_split_pair3_rne(a3.z, a3.w, ah[1].w, am[1].w, al[1].w);
        const uint4 *bs = Bq + buf * TV_PCHUNK + lane;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah[st]), Am = __builtin_bit_cast(bf16x8, am[st]), Al = __builtin_bit_cast(bf16x8, al[st]);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, bs[((nb * 2 + st) * 3 + 0) * 64]);
            const bf16x8 Bm = __builtin_bit_cast(bf16x8, bs[((nb * 2 + st) * 3 + 1) * 64]);
            const bf16x8 Bl = __builtin_bit_cast(bf16x8, bs[((nb * 2 + st) * 3 + 2) * 64]);
            tv_f32x16 &hh = nb ? acc1 : acc0, &sm = nb ? acs1 : acs0;
            hh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, hh, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, sm, 0, 0, 0);
            if constexpr (MODE >= 8) {
              sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bm, sm, 0, 0, 0);
              sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bl, sm, 0, 0, 0);
            }
            sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, sm, 0, 0, 0);
            sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, sm, 0, 0, 0);
          }
        }
      }
      if (kc == nkc - 1) {
        tv_f32x16 o0, o1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] = acc0[r] + acs0[r]; o1[r] = acc1[r] + acs1[r]; }
        epi(slab, o0, o1, bz0[slab], bz1[slab]);
      }
      if (c + 1 < nchunks) {
        uint4 *dst = Bq + (buf ^ 1) * TV_PCHUNK;
        dst[tid] = p1.a; dst[tid + 256] = p1.b; dst[tid + 512] = p1.c;
        __syncthreads();
        buf ^= 1;
      }
      p1 = p2;
    }
    pre = p1;  // chunk 0 of `next`
    return;
  }
  if constexpr (MODE == 1) {
    const float2 *wl = (const float2 *)Wp + kk * 16 * 32 + col;  // chunk c starts c * 1024 float2 further: chunks are contiguous
    float2 bc[16], bn[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bc[u] = wl[u * 32];
    tv_f32x16 acc0, acc1;
    for (int c = 0; c < nchunks; ++c) {
      const int slab = c / nkc, kc = c - slab * nkc;
      if (kc == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
      }
      if (c + 1 < nchunks) {
#pragma unroll
        for (int u = 0; u < 16; ++u) bn[u] = wl[(size_t)(c + 1) * 1024 + u * 32];
      }
      const float4 *ap = (const float4 *)(A + col * lda + kc * 32 + kk * 16);
      const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
      const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bc[u].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bc[u].y, acc1, 0, 0, 0);
      }
      if (kc == nkc - 1) epi(slab, acc0, acc1, bz0[slab], bz1[slab]);
      if (c + 1 < nchunks) {
#pragma unroll
        for (int u = 0; u < 16; ++u) bc[u] = bn[u];
      }
    }
    return;
  }
  // chunk c = (slab, kc): 32 x 32 x 2 floats at Wp + (slab * K + kc * 32) * 64
  float4 r0, r1;
  {
    const float4 *src = (const float4 *)Wp;
    r0 = src[tid]; r1 = src[tid + 256];
  }
  __syncthreads();  // previous users of Bs are done
  ((float4 *)Bs)[tid] = r0; ((float4 *)Bs)[tid + 256] = r1;
  __syncthreads();
  tv_f32x16 acc0, acc1;
  int buf = 0;
  for (int c = 0; c < nchunks; ++c) {
    const int slab = c / nkc, kc = c - slab * nkc;
    if (kc == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    }
    if (c + 1 < nchunks) {
      const int s2 = (c + 1) / nkc, k2 = (c + 1) - s2 * nkc;
      const float4 *src = (const float4 *)(Wp + ((size_t)s2 * K + k2 * 32) * 64);
      r0 = src[tid]; r1 = src[tid + 256];
    }
    {
      const float4 *ap = (const float4 *)(A + col * lda + kc * 32 + kk * 16);
      const float4 a0 = ap[0], a1 = ap[1], a2 = ap[2], a3 = ap[3];
      const float av[16] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w, a2.x, a2.y, a2.z, a2.w, a3.x, a3.y, a3.z, a3.w};
      const float2 *bs = (const float2 *)(Bs + buf * TV_BCHUNK) + kk * 16 * 32 + col;
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float2 b = bs[u * 32];
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], b.y, acc1, 0, 0, 0);
      }
    }
    if (kc == nkc - 1) epi(slab, acc0, acc1, bz0[slab], bz1[slab]);
    if (c + 1 < nchunks) {
      float4 *dst = (float4 *)(Bs + (buf ^ 1) * TV_BCHUNK);
      dst[tid] = r0; dst[tid + 256] = r1;
      __syncthreads();
      buf ^= 1;
    }
  }
}

// accumulator fragment (register r of lane (col, kk)) -> tile row / column:  row = (r & 3) + 8 * (r >> 2) + 4 * kk
#define TV_FOR_ACC(r, row) _Pragma("unroll") for (int r = 0, row = 4 * kk; r < 16; ++r, row = (r & 3) + 8 * (r >> 2) + 4 * kk)

// in-place LayerNorm of the 32 x 64 tile X (two lanes per row)
// a LayerNorm's scale / shift for this lane's half row (32 channels): fetched BEFORE the GEMM whose output it normalises and held in registers across it -
// fetched inside tv_layernorm the 16 loads had ~130 VALU instructions of cover for an L2 round trip (seven LayerNorms per tile, one wave per SIMD)
struct TvLn { float4 g[8], b[8]; };
__device__ __forceinline__ TvLn tv_ln_fetch(const float *__restrict__ g, const float *__restrict__ b) {
  const int half = (threadIdx.x & 63) >> 5;
  const float4 *gp = (const float4 *)(g + half * 32), *bp = (const float4 *)(b + half * 32);
  TvLn w;
#pragma unroll
  for (int q = 0; q < 8; ++q) { w.g[q] = gp[q]; w.b[q] = bp[q]; }
  return w;
}

__device__ __forceinline__ void tv_layernorm(float *X, const TvLn &w, float eps) {
  const int lane = threadIdx.x & 63, row = lane & 31, half = lane >> 5;
  float4 *p = (float4 *)(X + row * TV_XS + half * 32);
  float4 v[8];
  float s = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) { v[q] = p[q]; s += (v[q].x + v[q].y) + (v[q].z + v[q].w); }
  s += __shfl_xor(s, 32);
  const float mean = s / (float)TV_E;
  float q2 = 0.0f;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    v[q].x -= mean; v[q].y -= mean; v[q].z -= mean; v[q].w -= mean;
    q2 += (v[q].x * v[q].x + v[q].y * v[q].y) + (v[q].z * v[q].z + v[q].w * v[q].w);
  }
  q2 += __shfl_xor(q2, 32);
  const float rstd = 1.0f / sqrtf(q2 / (float)TV_E + eps);
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float4 gg = w.g[q], bb = w.b[q];
    float4 o;
    o.x = v[q].x * rstd * gg.x + bb.x; o.y = v[q].y * rstd * gg.y + bb.y;
    o.z = v[q].z * rstd * gg.z + bb.z; o.w = v[q].w * rstd * gg.w + bb.w;
    p[q] = o;
  }
}

// self-attention of the tokens of a voxel (<= 32 rows): one lane per (voxel g, head h, query t); q | k | v at columns 0 | E | 2 E of T, the output
// replaces q.  RC = rows per voxel at compile time (0: run time, RR): the scores are computed once and kept, the loops are straight code.
template <int RC>
__device__ __forceinline__ void tv_attention(float *T, int G, int kpts, float wpad, int lane, int RR = RC) {
  constexpr int RM = RC > 0 ? RC : 32;
  const int R = RC > 0 ? RC : RR;
  const float scale = 1.0f / sqrtf((float)TV_HD);
  const int items = G * TV_H * R;
  for (int ps = 0; ps * 64 < items; ++ps) {
    const int it = lane + 64 * ps;
    const bool on = it < items;
    const int g = on ? it / (TV_H * R) : 0, rem = on ? it - g * TV_H * R : 0;
    const int h = rem / R, t = rem - h * R;
    float *qp = T + (g * R + t) * TV_TS + h * TV_HD;
    float q[TV_HD], o[TV_HD];
#pragma unroll
    for (int d = 0; d < TV_HD; ++d) { q[d] = qp[d] * scale; o[d] = 0.0f; }
    if constexpr (RC > 0) {
      float sc[RM];
      float m = -3.0e38f;
#pragma unroll
      for (int j = 0; j < RM; ++j) {
        const float *kp = T + (g * R + j) * TV_TS + TV_E + h * TV_HD;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) s = fmaf(q[d], kp[d], s);
        sc[j] = s;
        m = fmaxf(m, s);
      }
      float den = 0.0f;
#pragma unroll
      for (int j = 0; j < RM; ++j) {
        const float *vp = T + (g * R + j) * TV_TS + 2 * TV_E + h * TV_HD;
        float pr = expf(sc[j] - m);
        if (j >= kpts) pr *= wpad;  // the padding row stands for P - kpts identical keys
        den += pr;
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) o[d] = fmaf(pr, vp[d], o[d]);
      }
      const float inv = 1.0f / den;
      if (on) {
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) qp[d] = o[d] * inv;
      }
    } else {
      float m = -3.0e38f;
      for (int j = 0; j < R; ++j) {
        const float *kp = T + (g * R + j) * TV_TS + TV_E + h * TV_HD;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) s = fmaf(q[d], kp[d], s);
        m = fmaxf(m, s);
      }
      float den = 0.0f;
      for (int j = 0; j < R; ++j) {
        const float *kp = T + (g * R + j) * TV_TS + TV_E + h * TV_HD;
        const float *vp = kp + TV_E;
        float s = 0.0f;
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) s = fmaf(q[d], kp[d], s);
        float pr = expf(s - m);
        if (j >= kpts) pr *= wpad;  // the padding row stands for P - kpts identical keys
        den += pr;
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) o[d] = fmaf(pr, vp[d], o[d]);
      }
      const float inv = 1.0f / den;
      if (on) {
#pragma unroll
        for (int d = 0; d < TV_HD; ++d) qp[d] = o[d] * inv;
      }
    }
  }
}

// Token deduplication (cls != NULL).  The reference runs the transformer over all P point slots of a voxel, zero padding included
// and unmasked (voxel_encoder.py:154-161), so the padding slots of a voxel are IDENTICAL tokens (zero point + the voxel's descriptor)
// and stay identical through every layer (the layers are permutation-equivariant).  A voxel with k < P points therefore has k + 1
// distinct rows: its k points and ONE padding row that stands for the P - k copies - in the softmax of the attention its key counts
// P - k times, everywhere else (GEMMs, LayerNorm, residuals, max over the slots) a copy changes nothing.  Voxels are grouped by
// k (ls3d_transvfe sorts them: cls = [perm[n] | off[P + 2]], off[k] = first position of class k in perm) and a 32-row wave tile takes
// floor(32 / (k + 1)) voxels of one class instead of floor(32 / P): on LiDAR data (1.4 points per voxel after the 5-point cap) 2.1x
// fewer tiles for the same result up to summation order inside the softmax.  A voxel whose slots beyond num_points are not all zero is
// put in class P (every slot a row of its own), so the deduplication never assumes what it has not checked.
template <int MODE>
__global__ __launch_bounds__(256, 1) void k_transvfe(const float *__restrict__ voxels, const int32_t *__restrict__ num, int n,
                                                     const int32_t *n_dev, int P, int C, TvParams prm, float *__restrict__ out, int out_ld,
                                                     const int32_t *__restrict__ cls) {
  HIP_DYNAMIC_SHARED(float, smem)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  float *X = smem + wave * TV_WAVE_FLOATS;     // [32][TV_XS]
  float *T = X + 32 * TV_XS;                   // [32][TV_TS]
  float *Bs = smem + 4 * TV_WAVE_FLOATS;       // [2][TV_BCHUNK] floats, or [2][TV_PCHUNK] uint4 in the plane modes
  const int N = ls3d_count(n, n_dev);
  const int32_t *perm = cls, *coff = cls ? cls + n : nullptr;
  int ntiles = 0;
  if (cls) {
    for (int c = 0; c <= P; ++c) {
      const int pb = 4 * (32 / (c < P ? c + 1 : P));
      ntiles += (coff[c + 1] - coff[c] + pb - 1) / pb;
    }
  } else {
    ntiles = (N + 4 * (32 / P) - 1) / (4 * (32 / P));
  }
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    // this tile: voxels perm[base + wave * G + g] (or base + wave * G + g), g < G, of class `kpts` points; R rows per voxel
    int kpts = P, R = P, base = tile * 4 * (32 / P), lim = N;
    if (cls) {
      int t = tile;
      for (int c = 0; c <= P; ++c) {
        const int rc = c < P ? c + 1 : P, pb = 4 * (32 / rc), tc = (coff[c + 1] - coff[c] + pb - 1) / pb;
        if (t < tc) { kpts = c; R = rc; base = coff[c] + t * pb; lim = coff[c + 1]; break; }
        t -= tc;
      }
    }
    const int G = 32 / R;                      // voxels per wave tile
    const int vw = base + wave * G;            // position of this wave's first voxel
    const float wpad = (float)(P - kpts);      // weight of the padding row's key in the softmax (kpts < P)
#define TV_VOXEL(g_) (((g_) < G && vw + (g_) < lim) ? (perm ? perm[vw + (g_)] : vw + (g_)) : -1)
    // ---- tokens -> T[:, 0:KT]: row g * R + p = [point p of voxel g (zeros for the padding row) | descriptor of voxel g | 0...]
    {
      const int row = lane & 31;
      const int g = row / R, p = row - g * R, v = TV_VOXEL(g);
      float *t = T + row * TV_TS;
      if (lane < 32 && P == 5 && C == 5) {
        // the shipped readers' voxel shape, compile-time: 25 loads in flight, no runtime-indexed arrays; rows of absent voxels read voxel 0 and
        // are zeroed by the select (no branch around the loads)
        float xv[25], desc[13];
        vfe_descriptor_fixed<5, 5>(voxels + (size_t)(v >= 0 ? v : 0) * 25, v >= 0 ? num[v] : 1, xv, desc);
        const bool pad = p >= kpts, live = v >= 0;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
          float pv = xv[c];  // point p of the voxel: a select chain over the five slots (p is a lane value)
#pragma unroll
          for (int q = 1; q < 5; ++q) pv = p == q ? xv[q * 5 + c] : pv;
          t[c] = (live && !pad) ? pv : 0.0f;
        }
#pragma unroll
        for (int c = 0; c < 13; ++c) t[5 + c] = live ? desc[c] : 0.0f;
#pragma unroll
        for (int c = 18; c < TV_KT; ++c) t[c] = 0.0f;
      } else if (lane < 32) {
        if (v >= 0) {
          const float *vox = voxels + (size_t)v * P * C;
          float desc[LS3D_MAX_FEAT + 8];
          vfe_descriptor(vox, P, C, num[v], desc);
          const bool pad = p >= kpts;
          for (int c = 0; c < C; ++c) t[c] = pad ? 0.0f : vox[p * C + c];
          for (int c = 0; c < C + 8; ++c) t[C + c] = desc[c];
          for (int c = 2 * C + 8; c < TV_KT; ++c) t[c] = 0.0f;
        } else {
          for (int c = 0; c < TV_KT; ++c) t[c] = 0.0f;
        }
      }
    }
    TV_WAVE_SYNC();
    TvPre pre;  // the weight chunk in flight across the GEMMs of the tile (plane modes)
    if constexpr (MODE >= 6) pre = tv_fetch((const uint4 *)prm.we);
    TvLn ln;    // scale / shift of the LayerNorm behind the next GEMM
    if (prm.num_layers > 0) ln = tv_ln_fetch(prm.layer[0].n1g, prm.layer[0].n1b);
    // ---- embedding (+ norm1 of layer 0)
    tv_gemm<MODE, TV_KT, TV_E>(T, TV_TS, prm.we, Bs, prm.be, [&](int slab, const tv_f32x16 &a0, const tv_f32x16 &a1, float b0, float b1) {
      TV_FOR_ACC(r, row) {
        X[row * TV_XS + col] = a0[r] + b0;
        X[row * TV_XS + 32 + col] = a1[r] + b1;
      }
    }, pre, (prm.num_layers > 0 ? prm.layer[0].wqkv : nullptr));
    TV_WAVE_SYNC();
    if (prm.num_layers > 0) tv_layernorm(X, ln, prm.layer[0].n1eps);
    TV_WAVE_SYNC();
    for (int l = 0; l < prm.num_layers; ++l) {
      const TvLayer &L = prm.layer[l];
      // ---- QKV -> T[:, 0:192]
      tv_gemm<MODE, TV_E, 3 * TV_E>(X, TV_XS, L.wqkv, Bs, L.bqkv, [&](int slab, const tv_f32x16 &a0, const tv_f32x16 &a1, float b0, float b1) {
        TV_FOR_ACC(r, row) {
          T[row * TV_TS + slab * 64 + col] = a0[r] + b0;
          T[row * TV_TS + slab * 64 + 32 + col] = a1[r] + b1;
        }
      }, pre, L.wo);
      TV_WAVE_SYNC();
      // ---- attention inside each voxel: one lane per (voxel g, head h, query token t); output over the query's slice.  The rows per voxel R
      //      (1 .. P, wave-uniform) select a compile-time instance: scores once, all key / value reads of a query issued together
      switch (R) {
        case 1: tv_attention<1>(T, G, kpts, wpad, lane); break;
        case 2: tv_attention<2>(T, G, kpts, wpad, lane); break;
        case 3: tv_attention<3>(T, G, kpts, wpad, lane); break;
        case 4: tv_attention<4>(T, G, kpts, wpad, lane); break;
        case 5: tv_attention<5>(T, G, kpts, wpad, lane); break;
        default: tv_attention<0>(T, G, kpts, wpad, lane, R); break;
      }
      TV_WAVE_SYNC();
      // ---- out-proj + residual (from the normed X) -> X, then norm2
      ln = tv_ln_fetch(L.n2g, L.n2b);
      tv_gemm<MODE, TV_E, TV_E>(T, TV_TS, L.wo, Bs, L.bo, [&](int slab, const tv_f32x16 &a0, const tv_f32x16 &a1, float b0, float b1) {
        TV_FOR_ACC(r, row) {
          X[row * TV_XS + col] = a0[r] + b0 + X[row * TV_XS + col];
          X[row * TV_XS + 32 + col] = a1[r] + b1 + X[row * TV_XS + 32 + col];
        }
      }, pre, L.w1);
      TV_WAVE_SYNC();
      tv_layernorm(X, ln, L.n2eps);
      TV_WAVE_SYNC();
      // ---- FF1 + ReLU -> T[:, 0:128]
      tv_gemm<MODE, TV_E, TV_FF>(X, TV_XS, L.w1, Bs, L.b1, [&](int slab, const tv_f32x16 &a0, const tv_f32x16 &a1, float b0, float b1) {
        TV_FOR_ACC(r, row) {
          T[row * TV_TS + slab * 64 + col] = fmaxf(a0[r] + b0, 0.0f);
          T[row * TV_TS + slab * 64 + 32 + col] = fmaxf(a1[r] + b1, 0.0f);
        }
      }, pre, L.w2);
      TV_WAVE_SYNC();
      // ---- FF2 + residual -> X, then norm1 of the next layer
      if (l + 1 < prm.num_layers) ln = tv_ln_fetch(prm.layer[l + 1].n1g, prm.layer[l + 1].n1b);
      tv_gemm<MODE, TV_FF, TV_E>(T, TV_TS, L.w2, Bs, L.b2, [&](int slab, const tv_f32x16 &a0, const tv_f32x16 &a1, float b0, float b1) {
        TV_FOR_ACC(r, row) {
          X[row * TV_XS + col] = a0[r] + b0 + X[row * TV_XS + col];
          X[row * TV_XS + 32 + col] = a1[r] + b1 + X[row * TV_XS + 32 + col];
        }
      }, pre, (l + 1 < prm.num_layers ? prm.layer[l + 1].wqkv : nullptr));
      TV_WAVE_SYNC();
      if (l + 1 < prm.num_layers) {
        tv_layernorm(X, ln, prm.layer[l + 1].n1eps);
        TV_WAVE_SYNC();
      }
    }
    // ---- max over the P token slots (padding slots take part, as in the reference) -> T[g][0:64]
    for (int g = 0; g < G; ++g) {
      float m = X[(g * R) * TV_XS + lane];
      for (int p = 1; p < R; ++p) m = fmaxf(m, X[(g * R + p) * TV_XS + lane]);
      T[g * TV_TS + lane] = m;
    }
    TV_WAVE_SYNC();
    if (prm.wc) {  // Linear + ReLU compression: one lane per (voxel, output)
      for (int it = lane; it < G * prm.ncomp; it += 64) {
        const int g = it / prm.ncomp, o = it - g * prm.ncomp;
        const float *w = prm.wc + (size_t)o * TV_E, *x = T + g * TV_TS;
        float s = 0.0f;
        for (int c = 0; c < TV_E; ++c) s = fmaf(x[c], w[c], s);
        s = fmaxf(s + prm.bc[o], 0.0f);
        const int v = TV_VOXEL(g);
        if (v >= 0) out[(size_t)v * out_ld + o] = s;
      }
    } else {
      for (int g = 0; g < G; ++g) {
        const int v = TV_VOXEL(g);
        if (v >= 0) out[(size_t)v * out_ld + lane] = T[g * TV_TS + lane];
      }
    }
    TV_WAVE_SYNC();
#undef TV_VOXEL
  }
}

// class of a voxel for the token deduplication: its point count k (clamped to [0, P]) when the slots beyond it are all zero, else P
__global__ __launch_bounds__(256) void k_tv_class_keys(const float *__restrict__ voxels, const int32_t *__restrict__ num, int n, const int32_t *n_dev,
                                                       int P, int C, uint32_t *__restrict__ keys) {
  const int N = ls3d_count(n, n_dev);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    int k = num[v];
    k = k < 0 ? 0 : (k > P ? P : k);
    const float *vox = voxels + (size_t)v * P * C;
    bool zero = true;
    for (int i = k * C; i < P * C; ++i) zero = zero && (vox[i] == 0.0f);
    keys[v] = (uint32_t)(zero ? k : P);
  }
}


// packed nt = 2 weights ([slab][K][32][2] floats: W[k][slab * 64 + nb * 32 + col] at ((slab * K + k) * 32 + col) * 2 + nb) -> plane chunks
__global__ __launch_bounds__(256) void k_tv_pack_planes(const float *__restrict__ w, int K, int N, uint4 *__restrict__ out) {
  const int nkc = K / 32;
  const long long total = (long long)(N / 64) * nkc * TV_PCHUNK;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    long long r = t;
    const int lane = (int)(r % 64); r /= 64;
    const int pl = (int)(r % 3); r /= 3;
    const int st = (int)(r % 2); r /= 2;
    const int nb = (int)(r % 2); r /= 2;
    const int kc = (int)(r % nkc); r /= nkc;
    const int slab = (int)r;
    const int col = lane & 31, kk = lane >> 5;
    unsigned wd[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      unsigned hm[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = kc * 32 + kk * 16 + st * 8 + pr * 2 + e;
        const float v = w[(((size_t)slab * K + k) * 32 + col) * 2 + nb];
        const unsigned hb = ls3d_bf16_rne(v) << 16;
        const float r1 = v - __uint_as_float(hb);
        const unsigned mb = ls3d_bf16_rne(r1) << 16;
        hm[e] = pl == 0 ? (hb >> 16) : pl == 1 ? (mb >> 16) : ls3d_bf16_rne(r1 - __uint_as_float(mb));
      }
      wd[pr] = hm[0] | (hm[1] << 16);
    }
    uint4 o; o.x = wd[0]; o.y = wd[1]; o.z = wd[2]; o.w = wd[3];
    out[t] = o;
  }
}

extern "C" size_t ls3d_transvfe_planes_bytes(int K, int N) { return (K % 32 || N % 64 || K < 32 || N < 64) ? 0 : (size_t)(N / 64) * (K / 32) * TV_PCHUNK * 16; }

extern "C" int ls3d_transvfe_pack_planes(const float *w_packed_nt2, int K, int N, void *out, ls3d_stream_t stream) {
  if (!w_packed_nt2 || !out || K < 32 || N < 64 || (K % 32) || (N % 64)) return LS3D_ERR_ARG;
  const long long total = (long long)(N / 64) * (K / 32) * TV_PCHUNK;
  hipLaunchKernelGGL(k_tv_pack_planes, ls3d_grid(total), dim3(256), 0, (hipStream_t)stream, w_packed_nt2, K, N, (uint4 *)out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t ls3d_radix_sort_workspace_bytes(int n);
extern "C" int ls3d_frame_offsets(const void *table, int is_float, int stride, int col, int n, const int32_t *n_dev, int batch, int32_t *off,
                                  ls3d_stream_t stream);

static inline size_t tv_align(size_t v) { return (v + 255) & ~(size_t)255; }
// workspace of the token deduplication: class keys, sorted keys, [perm | class offsets], sort buffers
extern "C" size_t ls3d_transvfe_workspace_bytes(int n, int P) {
  if (n <= 0 || P < 1 || P > 32) return 0;
  return 2 * tv_align((size_t)n * 4) + tv_align(((size_t)n + P + 2) * 4) + ls3d_radix_sort_workspace_bytes(n);
}

template <int MODE>
static int tv_launch(hipStream_t stream, long long blocks, int lds, const float *voxels, const int32_t *num_points, int n, const int32_t *n_dev, int P, int C,
                     const TvParams &prm, float *out, int out_ld, const int32_t *cls) {
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_transvfe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_transvfe<MODE>, dim3((unsigned)blocks), dim3(256), lds, stream, voxels, num_points, n, n_dev, P, C, prm, out, out_ld, cls);
  return LS3D_OK;
}

extern "C" int ls3d_transvfe(const float *voxels, const int32_t *num_points, int n, const int32_t *n_dev, int P, int C,
                             const ls3d_transvfe_t *m, float *out, int out_ld, void *workspace, size_t workspace_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!voxels || !num_points || !m || !out || n < 0 || P < 1 || C < 3) return LS3D_ERR_ARG;
  if (!m->w_embed || !m->b_embed || (m->num_layers > 0 && !m->layers)) return LS3D_ERR_ARG;
  if (m->planes != 0 && m->planes != 6 && m->planes != 8) return LS3D_ERR_ARG;
  // the fused kernel is specialised for the reference's configuration (num_embed 64, 4 heads, FF 128, <= 32-wide tokens)
  if (m->embed != TV_E || m->heads != TV_H || m->ffn != TV_FF || m->token_ld != TV_KT || 2 * C + 8 > TV_KT || C > LS3D_MAX_FEAT || P > 32 ||
      m->num_layers < 0 || m->num_layers > TV_MAX_LAYERS || (m->w_compress && (m->num_compressed < 1 || m->num_compressed > 64)))
    return LS3D_ERR_UNSUPPORTED;
  if (out_ld < (m->w_compress ? m->num_compressed : TV_E)) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  TvParams prm;
  prm.we = m->w_embed; prm.be = m->b_embed; prm.wc = m->w_compress; prm.bc = m->b_compress;
  prm.num_layers = m->num_layers; prm.ncomp = m->num_compressed;
  for (int l = 0; l < m->num_layers; ++l) {
    const ls3d_transvfe_layer_t &s = m->layers[l];
    if (!s.wqkv || !s.bqkv || !s.wo || !s.bo || !s.w1 || !s.b1 || !s.w2 || !s.b2 || !s.n1_gamma || !s.n1_beta || !s.n2_gamma || !s.n2_beta)
      return LS3D_ERR_ARG;
    prm.layer[l] = TvLayer{s.wqkv, s.bqkv, s.wo, s.bo, s.w1, s.b1, s.w2, s.b2, s.n1_gamma, s.n1_beta, s.n2_gamma, s.n2_beta, s.n1_eps, s.n2_eps};
  }
  if (m->w_compress && !m->b_compress) return LS3D_ERR_ARG;
  const int lds_f32 = (4 * TV_WAVE_FLOATS + 2 * TV_BCHUNK) * (int)sizeof(float);
  const int lds_planes = 4 * TV_WAVE_FLOATS * (int)sizeof(float) + 2 * TV_PCHUNK * 16;
  const int per_block = 4 * (32 / P);
  long long blocks = ((long long)n + per_block - 1) / per_block;
  // token deduplication (see k_transvfe) when the caller provides the workspace and does not switch it off (flags bit 1): class keys ->
  // one stable radix pass -> positions of the classes; 4 small launches in front of the reader
  const int32_t *cls = nullptr;
  if (workspace && !(m->flags & 2) && P > 1) {
    if (workspace_bytes < ls3d_transvfe_workspace_bytes(n, P) || ((uintptr_t)workspace & 15)) return LS3D_ERR_WORKSPACE;
    char *ws = (char *)workspace;
    uint32_t *keys = (uint32_t *)ws, *skeys = (uint32_t *)(ws + tv_align((size_t)n * 4));
    int32_t *pc = (int32_t *)(ws + 2 * tv_align((size_t)n * 4));
    void *sort_ws = ws + 2 * tv_align((size_t)n * 4) + tv_align(((size_t)n + P + 2) * 4);
    int bits = 1;
    while ((1 << bits) <= P) ++bits;
    hipLaunchKernelGGL(k_tv_class_keys, ls3d_grid(n), dim3(256), 0, stream, voxels, num_points, n, n_dev, P, C, keys);
    int rc2 = ls3d_radix_sort_pairs(keys, nullptr, n, n_dev, bits, skeys, pc, sort_ws, ls3d_radix_sort_workspace_bytes(n), stream);
    if (rc2 != LS3D_OK) return rc2;
    rc2 = ls3d_frame_offsets(skeys, 0, 1, 0, n, n_dev, P + 1, pc + n, stream_);  // off[c] = first position with class >= c, off[P + 1] = count
    if (rc2 != LS3D_OK) return rc2;
    cls = pc;
    blocks += P + 1;  // every class may end in a partial tile
  }
  if (blocks > 65536) blocks = 65536;
  int rc;
  if (m->planes == 6) rc = tv_launch<6>(stream, blocks, lds_planes, voxels, num_points, n, n_dev, P, C, prm, out, out_ld, cls);
  else if (m->planes == 8) rc = tv_launch<8>(stream, blocks, lds_planes, voxels, num_points, n, n_dev, P, C, prm, out, out_ld, cls);
  else if (m->flags & 1) rc = tv_launch<1>(stream, blocks, lds_f32, voxels, num_points, n, n_dev, P, C, prm, out, out_ld, cls);
  else rc = tv_launch<0>(stream, blocks, lds_f32, voxels, num_points, n, n_dev, P, C, prm, out, out_ld, cls);
  if (rc != LS3D_OK) return rc;
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
