// spconv_bwd.hip — weight gradient of the sparse convolutions (spconv v1.x's indice_conv_backward,
// call sites det3d/models/backbones/scn_unet.py through autograd).
//
//   forward   out[o]      = sum_k W[k]^T in[tbl[o][k]]
//   dgrad     grad_in[i]  = sum_k W[k] grad_out[tblT[i][k]]      -> the forward gather-GEMM on the transposed table with
//                                                                   transposed (SubM: also mirrored) weights, no new kernel
//   wgrad     grad_W[k]   = sum_{o : tbl[o][k] >= 0} in[tbl[o][k]] (x) grad_out[o]          (this file)
//
// wgrad on CDNA4: for one kernel offset k the sum over output rows is a GEMM with the ROWS as the reduction dimension, so
// two rows feed one v_mfma_f32_32x32x2_f32: lane (i, half) supplies in[tbl[o_half][k]][ci0 + i] as the A operand and
// grad_out[o_half][co0 + i] as the B operand (32 consecutive floats of a row per half wave: coalesced 128-byte reads, no
// LDS staging).  A wave owns a 32 x (32 COB) block of grad_W[k] in COB accumulators and walks its share of the rows in
// mask-sorted order, skipping row pairs without a neighbour at k.  The 4 waves of a workgroup are CIW x RG: CIW = min(4, cin/32)
// waves side by side along cin over the SAME row pairs (they share the grad_out rows through the L1: grad_out is read kvol times
// per layer instead of kvol x cin/32 times - round 2 measured the layout with one 32-channel block per workgroup at 1.9 ms per
// 128 -> 128 layer on 241k rows, bound by those re-reads) and RG = 4 / CIW groups of interleaved row pairs, summed through LDS
// in a fixed order; row chunks are summed by a second kernel: deterministic, no atomics.
// Exact f32 (fmaf chain per element).
#include "common.h"
#include "gemm_common.h"

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_UNROLL = 4;  // row pairs in flight per wave

// Pair lists per kernel offset, compacted: pin[k][j] / pout[k][j] = input row / output row of the j-th pair of offset k in processing
// order (j < cnt[k]).  Round 2 handed the kernels the transposed table and let them skip groups without any pair; but a group of 16
// consecutive rows almost always has SOME row with the neighbour (pair density 0.46 / 0.65 / 0.72 on levels 2 - 4 of the 120k frame), so
// 30 - 55 % of the rows a workgroup staged and multiplied were zero rows.  Three launches (count per block of 1024 positions, scan of the
// block counts per offset, emit); the order of the pairs - and with it every summation order - is fixed by the table and the row order.
constexpr int PL_BLK = 1024;
__device__ __forceinline__ int pl_entry(const int32_t *tbl, const int32_t *order, int N, int kvol, int k, int p, int *o_out) {
  if (p >= N) return -1;
  const int o = order ? order[p] : p;
  *o_out = o;
  return tbl[(size_t)o * kvol + k];
}
__global__ __launch_bounds__(256) void k_pairs_count(const int32_t *tbl, const int32_t *order, int n, const int32_t *n_dev, int kvol, int nb,
                                                    int32_t *bcount) {
  __shared__ int s_c[256];
  const int tid = threadIdx.x, k = blockIdx.y, b = blockIdx.x, N = ls3d_count(n, n_dev);
  int c = 0, o;
  for (int q = 0; q < PL_BLK / 256; ++q) c += pl_entry(tbl, order, N, kvol, k, b * PL_BLK + tid * (PL_BLK / 256) + q, &o) >= 0;
  s_c[tid] = c;
  __syncthreads();
  for (int d = 128; d >= 1; d >>= 1) {
    if (tid < d) s_c[tid] += s_c[tid + d];
    __syncthreads();
  }
  if (tid == 0) bcount[(size_t)k * nb + b] = s_c[0];
}
__global__ __launch_bounds__(256) void k_pairs_scan(int32_t *bcount, int nb, int32_t *cnt) {  // one workgroup per offset: exclusive scan in place
  __shared__ int s_a[256];
  __shared__ int s_carry;
  const int tid = threadIdx.x, k = blockIdx.x;
  int32_t *a = bcount + (size_t)k * nb;
  if (tid == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 256) {
    const int v = base + tid < nb ? a[base + tid] : 0;
    s_a[tid] = v;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
      const int t = tid >= d ? s_a[tid - d] : 0;
      __syncthreads();
      s_a[tid] += t;
      __syncthreads();
    }
    const int incl = s_a[tid], carry = s_carry;
    if (base + tid < nb) a[base + tid] = carry + incl - v;
    __syncthreads();
    if (tid == 255) s_carry = carry + incl;
    __syncthreads();
  }
  if (tid == 0) cnt[k] = s_carry;
}
__global__ __launch_bounds__(256) void k_pairs_emit(const int32_t *tbl, const int32_t *order, int n, const int32_t *n_dev, int kvol, int nb,
                                                   const int32_t *boff, int32_t *pin, int32_t *pout) {
  __shared__ int s_s[256];
  const int tid = threadIdx.x, k = blockIdx.y, b = blockIdx.x, N = ls3d_count(n, n_dev);
  int idx[PL_BLK / 256], o[PL_BLK / 256], c = 0;
#pragma unroll
  for (int q = 0; q < PL_BLK / 256; ++q) {  // the thread's consecutive positions
    o[q] = 0;
    idx[q] = pl_entry(tbl, order, N, kvol, k, b * PL_BLK + tid * (PL_BLK / 256) + q, &o[q]);
    c += idx[q] >= 0;
  }
  s_s[tid] = c;
  __syncthreads();
  for (int d = 1; d < 256; d <<= 1) {
    const int t = tid >= d ? s_s[tid - d] : 0;
    __syncthreads();
    s_s[tid] += t;
    __syncthreads();
  }
  int pos = boff[(size_t)k * nb + b] + s_s[tid] - c;
#pragma unroll
  for (int q = 0; q < PL_BLK / 256; ++q)
    if (idx[q] >= 0) {
      pin[(size_t)k * n + pos] = idx[q];
      pout[(size_t)k * n + pos] = o[q];
      ++pos;
    }
}

template <int COB>
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float *__restrict__ in, int in_ld, const float *__restrict__ gout, int go_ld,
                                                      const int32_t *__restrict__ tbl_t, const int32_t *__restrict__ o_all, int kvol, int cin,
                                                      int cout, int n_rows, const int32_t *__restrict__ pair_cnt, int nchunks,
                                                      float *__restrict__ partial) {
  __shared__ float red[32 * 32 * COB * 2];  // the tiles handed over in one round: one per cin block of the workgroup with RG > 1
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int N = pair_cnt[blockIdx.y];  // pairs of this offset (compacted lists: every entry below N is a pair)
  const int32_t *o_t = o_all + (size_t)blockIdx.y * n_rows;
  const int ci_blocks = (cin + 31) / 32;
  const int CIW = ci_blocks >= 4 ? 4 : ci_blocks >= 2 ? 2 : 1, RG = 4 / CIW;
  const int ci_groups = (ci_blocks + CIW - 1) / CIW;
  const int cw = wave % CIW, rg = wave / CIW;
  const int k = blockIdx.y;
  const int chunk = blockIdx.x / ci_groups, cb = (blockIdx.x % ci_groups) * CIW + cw;
  const int ci = cb * 32 + i;
  const int npairs = (N + 1) / 2;
  const int per_chunk = (npairs + nchunks - 1) / nchunks;
  const int p0 = chunk * per_chunk, p1 = min(npairs, p0 + per_chunk);
  const int32_t *tk = tbl_t + (size_t)k * n_rows;
  wg_f32x16 acc[COB];
#pragma unroll
  for (int n = 0; n < COB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  // Software pipeline over groups of WG_UNROLL row pairs, written so that hipcc never waits for a load before the step that consumes it
  // (the same rules as in k_spconv_wgrad_lds below): every load is unconditional - rows past the end re-read entry 0 of the list (a valid
  // pair), channels past cin / cout a clamped column whose products are never stored - validity is a function of the position alone and is
  // applied when a set is consumed (only the last group of a list has invalid rows); indices run three groups ahead of the MFMAs in two
  // register sets, operands two groups ahead in two more, the index loads of a step in front of its operand loads (vmcnt counts in
  // order: waiting for the indices leaves the operands in flight); the MFMAs are unconditional, so the accumulators never move.
  const int cic = ci < cin ? ci : cin - 1;
  int colc[COB];
#pragma unroll
  for (int n = 0; n < COB; ++n) colc[n] = n * 32 + i < cout ? n * 32 + i : cout - 1;
  int oP[WG_UNROLL], iP[WG_UNROLL], oQ[WG_UNROLL], iQ[WG_UNROLL];
  float a0[WG_UNROLL], b0[WG_UNROLL][COB], a1[WG_UNROLL], b1[WG_UNROLL][COB];
#define WG_LOAD_IDX(IX, OX, p_)                                                               \
  _Pragma("unroll") for (int u = 0; u < WG_UNROLL; ++u) {                                     \
    const int r_ = 2 * ((p_) + u) + half;                                                     \
    const bool ok_ = (p_) + u < p1 && r_ < N;                                                 \
    OX[u] = o_t[ok_ ? r_ : 0];                                                                \
    IX[u] = tk[ok_ ? r_ : 0];                                                                 \
  }
#define WG_LOAD_OPS(A, B, IX, OX)                                                             \
  _Pragma("unroll") for (int u = 0; u < WG_UNROLL; ++u) {                                     \
    A[u] = in[(size_t)IX[u] * in_ld + cic];                                                   \
    _Pragma("unroll") for (int n = 0; n < COB; ++n) B[u][n] = gout[(size_t)OX[u] * go_ld + colc[n]]; \
  }
  // one group at position p_: take the set's operands (zero the rows past the end), fetch the indices of the group three ahead into the free
  // index set, refill the operand set through the other index set, multiply
#define WG_STEP(A, B, IXU, OXU, IXL, OXL, p_)                                                 \
  {                                                                                           \
    float a[WG_UNROLL], bb[WG_UNROLL][COB];                                                   \
    const bool full_ = (p_) + WG_UNROLL <= p1 && 2 * ((p_) + WG_UNROLL) <= N;                 \
    _Pragma("unroll") for (int u = 0; u < WG_UNROLL; ++u) {                                   \
      const bool ok_ = full_ || ((p_) + u < p1 && 2 * ((p_) + u) + half < N);                 \
      a[u] = ok_ ? A[u] : 0.0f;                                                               \
      _Pragma("unroll") for (int n = 0; n < COB; ++n) bb[u][n] = ok_ ? B[u][n] : 0.0f;        \
    }                                                                                         \
    LS3D_SCHED_FENCE();                                                                       \
    WG_LOAD_IDX(IXL, OXL, (p_) + 3 * pstep)                                                   \
    LS3D_SCHED_FENCE();                                                                       \
    WG_LOAD_OPS(A, B, IXU, OXU)                                                               \
    LS3D_SCHED_FENCE();                                                                       \
    _Pragma("unroll") for (int u = 0; u < WG_UNROLL; ++u)                                     \
      _Pragma("unroll") for (int n = 0; n < COB; ++n)                                         \
          acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bb[u][n], acc[n], 0, 0, 0);     \
  }
  const int pstep = RG * WG_UNROLL;
  int p = p0 + rg * WG_UNROLL;
  if (p0 < p1) {  // (entry 0 of an empty list is not a pair: nothing is read then)
    WG_LOAD_IDX(iP, oP, p)
    WG_LOAD_OPS(a0, b0, iP, oP)
    WG_LOAD_IDX(iQ, oQ, p + pstep)
    LS3D_SCHED_FENCE();
    WG_LOAD_IDX(iP, oP, p + 2 * pstep)
    LS3D_SCHED_FENCE();
    WG_LOAD_OPS(a1, b1, iQ, oQ)
    LS3D_SCHED_FENCE();
    for (; p < p1; p += 2 * pstep) {
      WG_STEP(a0, b0, iP, oP, iQ, oQ, p)
      WG_STEP(a1, b1, iQ, oQ, iP, oP, p + pstep)
    }
  }
#undef WG_STEP
#undef WG_LOAD_IDX
#undef WG_LOAD_OPS
  // ---- the row groups 1..RG-1 of a cin block hand their tiles to row group 0 one after the other (fixed order; at most two
  //      tiles in flight: the LDS footprint, not the registers, limits the resident workgroups), which then writes the chunk's
  //      partial [32][cout] block
  for (int r2 = 1; r2 < RG; ++r2) {
    float *slot = red + (CIW == 2 ? cw * (32 * 32 * COB) : 0);
    if (rg == r2) {
#pragma unroll
      for (int n = 0; n < COB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) slot[(n * 16 + r) * 64 + lane] = acc[n][r];
    }
    __syncthreads();
    if (rg == 0) {
#pragma unroll
      for (int n = 0; n < COB; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] += slot[(n * 16 + r) * 64 + lane];
    }
    __syncthreads();
  }
  if (rg == 0 && cb < ci_blocks) {
    float *dst = partial + (((size_t)chunk * kvol + k) * cin) * cout;
#pragma unroll
    for (int n = 0; n < COB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = n * 32 + i;  // fragment layout of the 32x32 MFMA
        if (row < cin && col < cout) dst[(size_t)row * cout + col] = acc[n][r];
      }
  }
}

// The same sums on the exact 3-plane bf16 split: 16 rows feed one v_mfma_f32_32x32x16_bf16 per plane product, and the
// ROWS are the reduction dimension of that MFMA, so both operands have to be transposed: a lane of the A operand holds 8 consecutive rows
// of ONE input channel, a lane of the B operand 8 consecutive rows of one output channel.  A workgroup owns offset k, a tile of <= 128 input
// channels and all <= 128 output channels for a chunk of the rows.  Per group of 16 rows: thread (c, rg) loads rows [8 rg, 8 rg + 8) of
// channel c of the gathered input rows and of grad_out (lanes run along the channels: every load instruction reads whole 256-byte row
// segments), splits its 8 + 8 values into three round-to-nearest bf16 planes in registers and writes each plane with ONE 16-byte LDS store
// at [plane][row half rg][channel] - which is exactly the MFMA operand layout, so every wave fetches its fragments with ds_read_b128.  The
// (cin / 32) x (cout / 32) output blocks are dealt to the 4 waves (<= 4 each); LDS is double buffered (one barrier per group) and the next
// group's rows are in flight while this group's MFMAs run.  head x head goes into `acc`, the small products into `acs` (tileconv.hip).
constexpr int WGL_ROWS = 16;
template <int NP>
__device__ __forceinline__ void wgl_multiply(const uint4 (*sa)[2][128], const uint4 (*sb)[2][128], f32x16 *acc, f32x16 *acs, int wave, int i, int half,
                                             int nblk, int NB) {
  int nb_prev = -1;
  bf16x8 Bh, Bm, Bl;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int blk = wave + 4 * t;
    if (blk < nblk) {
      const int cb = blk / NB, nb = blk % NB;
      if (nb != nb_prev) {
        Bh = __builtin_bit_cast(bf16x8, sb[0][half][nb * 32 + i]);
        Bm = __builtin_bit_cast(bf16x8, sb[1][half][nb * 32 + i]);
        Bl = __builtin_bit_cast(bf16x8, sb[2][half][nb * 32 + i]);
        nb_prev = nb;
      }
      const bf16x8 Ah = __builtin_bit_cast(bf16x8, sa[0][half][cb * 32 + i]);
      const bf16x8 Am = __builtin_bit_cast(bf16x8, sa[1][half][cb * 32 + i]);
      const bf16x8 Al = __builtin_bit_cast(bf16x8, sa[2][half][cb * 32 + i]);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[t], 0, 0, 0);
      acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acs[t], 0, 0, 0);
      acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acs[t], 0, 0, 0);
      if constexpr (NP >= 8) {
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bm, acs[t], 0, 0, 0);
        acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bl, acs[t], 0, 0, 0);
      }
      acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bm, acs[t], 0, 0, 0);
      acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, acs[t], 0, 0, 0);
      acs[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, acs[t], 0, 0, 0);
    }
  }
}

template <int NP>
__global__ __launch_bounds__(256, 2) void k_spconv_wgrad_lds(const float *__restrict__ in, int in_ld, const float *__restrict__ gout, int go_ld,
                                                             const int32_t *__restrict__ tbl_t, const int32_t *__restrict__ o_all, int kvol, int cin,
                                                             int cout, int n_rows, const int32_t *__restrict__ pair_cnt, int nchunks,
                                                             float *__restrict__ partial) {
  __shared__ uint4 sA[2][3][2][128];  // [buffer][plane][row half][channel] x 8 bf16 (rows 8 h .. 8 h + 8 of the group): lanes run along the
                                      // channels in the staging stores and in the fragment reads alike -> consecutive 16-byte words, no bank conflicts
  __shared__ uint4 sB[2][3][2][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, half = lane >> 5;
  const int c = tid & 127, rg = __builtin_amdgcn_readfirstlane(tid >> 7);
  const int N = pair_cnt[blockIdx.y];  // pairs of this offset (compacted lists)
  const int32_t *o_t = o_all + (size_t)blockIdx.y * n_rows;
  const int ci_tiles = (cin + 127) / 128;
  const int k = blockIdx.y;
  const int chunk = blockIdx.x / ci_tiles, ct = blockIdx.x % ci_tiles;
  const int ci0 = ct * 128;
  const int CB = (min(128, cin - ci0) + 31) / 32, NB = (cout + 31) / 32, nblk = CB * NB;
  const int ngroups = (N + WGL_ROWS - 1) / WGL_ROWS;
  const int per_chunk = (ngroups + nchunks - 1) / nchunks;
  const int g0 = chunk * per_chunk, g1 = min(ngroups, g0 + per_chunk);
  const int32_t *tk = tbl_t + (size_t)k * n_rows;
  const bool a_live = c < CB * 32, b_live = c < NB * 32, a_in = ci0 + c < cin, b_in = c < cout;
  const int ca = a_in ? ci0 + c : cin - 1, cbc = b_in ? c : cout - 1;  // clamped columns: every load address is valid
  f32x16 acc[4], acs[4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = acs[t][r] = 0.0f;
  // Row pipeline.  The 16 row indices of a group come with ONE unconditional vector load each (lane l holds row l & 15; rows past the end
  // re-read entry 0, a valid pair) and are handed out with v_readlane; every row load is unconditional too.  Nothing in the loop
  // consumes a loaded value before the step that needs it: validity is a wave-uniform bit mask computed from the group number alone, rows
  // past the end are zeroed at split time (only the last group of a list has any), channels past cin / cout are loaded from a clamped
  // column and never stored.  (Selecting on the loaded values right behind the loads made hipcc wait for all 16 of them in place -
  // vmcnt(0) per group, the whole memory latency exposed: 22 % MFMA busy.)  Indices run three groups ahead of the MFMAs in two register
  // sets, rows two groups ahead in two more; the index load of a step is issued BEFORE its row loads, so that waiting for it (in the next
  // step) leaves those row loads in flight (vmcnt counts in order).
  float ra0[8], rb0[8], ra1[8], rb1[8];
  unsigned m0 = 0, m1 = 0, mP = 0, mQ = 0;  // valid rows (bit r = row r of the group) of the two row sets and the two index sets
  int tkP = 0, opP = 0, tkQ = 0, opQ = 0;
#define WGL_IDX(TK, OP, MK, g_)                                                                  \
  {                                                                                              \
    const int r16_ = (g_) * WGL_ROWS + (lane & 15);                                              \
    const bool v16_ = (g_) < g1 && r16_ < N;                                                     \
    TK = tk[v16_ ? r16_ : 0];                                                                    \
    OP = o_t[v16_ ? r16_ : 0];                                                                   \
    MK = (unsigned)__ballot(v16_) & 0xFFFFu;                                                     \
  }
#define WGL_ROWS_LOAD(RA, RB, TK, OP)                                                            \
  {                                                                                              \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                              \
      const int idx_ = __builtin_amdgcn_readlane(TK, rg * 8 + j);                                \
      const int o_ = __builtin_amdgcn_readlane(OP, rg * 8 + j);                                  \
      RA[j] = in[(size_t)idx_ * in_ld + ca];                                                     \
      RB[j] = gout[(size_t)o_ * go_ld + cbc];                                                    \
    }                                                                                            \
  }
  // one group: split this set's rows into LDS, fetch the indices of group gi_ into the free index set, refill the row set with the rows
  // of the group whose indices are pending in the other one, multiply
#define WGL_STEP(RA, RB, MS, TKU, OPU, MKU, TKL, OPL, MKL, gi_)                                  \
  {                                                                                              \
    const unsigned ms_ = MS;                                                                     \
    if (ms_) {                                                                                   \
      if (ms_ != 0xFFFFu) {                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                            \
          if (!((ms_ >> (rg * 8 + j)) & 1u)) RA[j] = RB[j] = 0.0f;                               \
      }                                                                                          \
      uint4 h, m, l;                                                                             \
      if (a_live) {                                                                              \
        ls3d_split_pair3_rne(RA[0], RA[1], h.x, m.x, l.x);                                       \
        ls3d_split_pair3_rne(RA[2], RA[3], h.y, m.y, l.y);                                       \
        ls3d_split_pair3_rne(RA[4], RA[5], h.z, m.z, l.z);                                       \
        ls3d_split_pair3_rne(RA[6], RA[7], h.w, m.w, l.w);                                       \
        sA[buf][0][rg][c] = h; sA[buf][1][rg][c] = m; sA[buf][2][rg][c] = l;                     \
      }                                                                                          \
      if (b_live) {                                                                              \
        ls3d_split_pair3_rne(RB[0], RB[1], h.x, m.x, l.x);                                       \
        ls3d_split_pair3_rne(RB[2], RB[3], h.y, m.y, l.y);                                       \
        ls3d_split_pair3_rne(RB[4], RB[5], h.z, m.z, l.z);                                       \
        ls3d_split_pair3_rne(RB[6], RB[7], h.w, m.w, l.w);                                       \
        sB[buf][0][rg][c] = h; sB[buf][1][rg][c] = m; sB[buf][2][rg][c] = l;                     \
      }                                                                                          \
    }                                                                                            \
    WGL_IDX(TKL, OPL, MKL, gi_)                                                                  \
    LS3D_SCHED_FENCE(); /* index loads stay in front of the row loads */                         \
    WGL_ROWS_LOAD(RA, RB, TKU, OPU)                                                              \
    LS3D_SCHED_FENCE();                                                                          \
    MS = MKU;                                                                                    \
    if (ms_) {                                                                                   \
      __syncthreads();                                                                           \
      wgl_multiply<NP>(sA[buf], sB[buf], acc, acs, wave, i, half, nblk, NB);                     \
      buf ^= 1;                                                                                  \
    }                                                                                            \
  }
  int buf = 0;
  if (g0 < g1) {  // (an empty chunk has nothing to read: entry 0 of an empty list is not a pair)
    // (the prologue leaves the loads in the order a loop step does - indices of g0 + 2, then the rows of g0 + 1 - so that the wait counts
    // hipcc derives for the loop head are the steady-state ones on both ways in)
    WGL_IDX(tkP, opP, mP, g0)
    WGL_ROWS_LOAD(ra0, rb0, tkP, opP)
    m0 = mP;
    WGL_IDX(tkQ, opQ, mQ, g0 + 1)
    LS3D_SCHED_FENCE();
    WGL_IDX(tkP, opP, mP, g0 + 2)
    LS3D_SCHED_FENCE();
    WGL_ROWS_LOAD(ra1, rb1, tkQ, opQ)
    LS3D_SCHED_FENCE();
    m1 = mQ;
    for (int g = g0; g < g1; g += 2) {
      WGL_STEP(ra0, rb0, m0, tkP, opP, mP, tkQ, opQ, mQ, g + 3)
      WGL_STEP(ra1, rb1, m1, tkQ, opQ, mQ, tkP, opP, mP, g + 4)
    }
  }
#undef WGL_STEP
#undef WGL_ROWS_LOAD
#undef WGL_IDX
  float *dst = partial + (((size_t)chunk * kvol + k) * cin) * cout;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int blk = wave + 4 * t;
    if (blk < nblk) {
      const int cb = blk / NB, nb = blk % NB;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = ci0 + cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = nb * 32 + i;  // fragment layout of the 32x32 MFMA
        if (row < cin && col < cout) dst[(size_t)row * cout + col] = acc[t][r] + acs[t][r];
      }
    }
  }
}

// sum of the row chunks' partial gradients [chunk][kvol * cin][cw] -> columns [col0, col0 + cw) of gw[kvol * cin][cout]
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *partial, int nchunks, long long elems, int cw, int cout, int col0, float *gw) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < elems; t += (long long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * elems + t];
    gw[(t / cw) * cout + col0 + (t % cw)] = s;
  }
}
// the same for many chunks of a small gradient (a Linear layer: <= 1024 chunks of <= 256 x 128 sums): a workgroup owns 32 elements, its 8
// thread groups each add every 8th chunk in order, then the 8 sums are added in order - a fixed summation tree with 8 x the threads
__global__ __launch_bounds__(256) void k_wgrad_reduce_split(const float *partial, int nchunks, long long elems, int cw, int cout, int col0, float *gw) {
  __shared__ float red[8][32];
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const long long t = (long long)blockIdx.x * 32 + lane;
  float s = 0.0f;
  if (t < elems) {
#pragma unroll 8
    for (int c = grp; c < nchunks; c += 8) s += partial[(size_t)c * elems + t];
  }
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && t < elems) {
    float v = red[0][lane];
#pragma unroll
    for (int g = 1; g < 8; ++g) v += red[g][lane];
    gw[(t / cw) * cout + col0 + (t % cw)] = v;
  }
}

static inline int wg_ci_groups(int cin) {
  const int ci_blocks = (cin + 31) / 32, ciw = ci_blocks >= 4 ? 4 : ci_blocks >= 2 ? 2 : 1;
  return (ci_blocks + ciw - 1) / ciw;
}

static inline int wg_chunks(int n_rows, int kvol, int cin) {
  // enough workgroups for 256 CUs x ~8, at least ~256 row pairs per chunk; the chunks' partial sums (kvol x cin x <= 128 floats each,
  // written once and read once by the reduction) stay under ~256 MB.  Few offsets (a Linear layer's weight gradient is one offset
  // over 10^5..10^6 rows) need many chunks: a fixed cap of 96 left 160 CUs idle there.
  const long long per = (long long)kvol * wg_ci_groups(cin);
  long long want = (2048 + per - 1) / per;
  long long cap = ((long long)n_rows / 2 + 255) / 256;
  if (want > cap) want = cap;
  long long cap_bytes = (256ll << 20) / ((long long)kvol * cin * 128 * 4);
  if (cap_bytes > 1024) cap_bytes = 1024;
  if (want > cap_bytes) want = cap_bytes;
  if (want < 1) want = 1;
  return (int)want;
}

static inline size_t wg_align(size_t v) { return (v + 255) & ~(size_t)255; }
// pair lists of one (table, row order): [pin kvol x n][pout kvol x n][block counts kvol x nb][pair counts kvol]
extern "C" size_t ls3d_spconv_pairs_bytes(int kvol, int n_rows) {
  const size_t nr = n_rows > 0 ? n_rows : 1, nb = (nr + PL_BLK - 1) / PL_BLK;
  return 2 * wg_align((size_t)kvol * nr * 4) + wg_align((size_t)kvol * nb * 4) + wg_align((size_t)kvol * 4);
}
extern "C" size_t ls3d_spconv_wgrad_workspace_bytes(int kvol, int cin, int cout, int n_rows) {
  if (cout > 128) cout = 128;  // wider layers run in slabs of 128 output columns through the same partial-sum buffer
  return wg_align((size_t)wg_chunks(n_rows, kvol, cin) * kvol * cin * cout * sizeof(float)) + ls3d_spconv_pairs_bytes(kvol, n_rows) + 256;
}

extern "C" int ls3d_spconv_pairs(const int32_t *tbl, const int32_t *row_order, int n_rows, const int32_t *n_rows_dev, int kvol, void *pairs,
                                 size_t pairs_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows == 0 && kvol >= 1) return LS3D_OK;
  if (!tbl || !pairs || n_rows < 0 || kvol < 1) return LS3D_ERR_ARG;
  if (pairs_bytes < ls3d_spconv_pairs_bytes(kvol, n_rows) || ((uintptr_t)pairs & 15)) return LS3D_ERR_WORKSPACE;
  const int nb = (n_rows + PL_BLK - 1) / PL_BLK;
  char *wsp = (char *)pairs;
  int32_t *pin = (int32_t *)wsp; wsp += wg_align((size_t)kvol * n_rows * 4);
  int32_t *pout = (int32_t *)wsp; wsp += wg_align((size_t)kvol * n_rows * 4);
  int32_t *bcount = (int32_t *)wsp; wsp += wg_align((size_t)kvol * nb * 4);
  int32_t *pair_cnt = (int32_t *)wsp;
  hipLaunchKernelGGL(k_pairs_count, dim3(nb, kvol), dim3(256), 0, stream, tbl, row_order, n_rows, n_rows_dev, kvol, nb, bcount);
  hipLaunchKernelGGL(k_pairs_scan, dim3(kvol), dim3(256), 0, stream, bcount, nb, pair_cnt);
  hipLaunchKernelGGL(k_pairs_emit, dim3(nb, kvol), dim3(256), 0, stream, tbl, row_order, n_rows, n_rows_dev, kvol, nb, (const int32_t *)bcount, pin, pout);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

static int wg_on_pairs(const float *in, int in_ld, const float *grad_out, int go_ld, const void *pairs, int kvol, int cin, int cout, int n_rows,
                       int products, void *workspace, float *grad_w, hipStream_t stream);

extern "C" int ls3d_spconv_wgrad(const float *in, int in_ld, const float *grad_out, int go_ld, const int32_t *tbl, const int32_t *row_order,
                                 int kvol, int cin, int cout, int n_rows, const int32_t *n_rows_dev, int products, void *workspace,
                                 size_t workspace_bytes, float *grad_w, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows == 0 && grad_w && kvol >= 1 && cin >= 1 && cout >= 1) {  // empty tensors have no storage: the gradient is zero
    hipMemsetAsync(grad_w, 0, (size_t)kvol * cin * cout * sizeof(float), stream);
    return LS3D_OK;
  }
  if (!in || !grad_out || !tbl || !grad_w || !workspace || kvol < 1 || cin < 1 || cout < 1 || n_rows < 0) return LS3D_ERR_ARG;
  if (in_ld < cin || go_ld < cout) return LS3D_ERR_ARG;
  if ((products & 63) != 0 && (products & 63) != 6 && (products & 63) != 8) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_spconv_wgrad_workspace_bytes(kvol, cin, cout, n_rows)) return LS3D_ERR_WORKSPACE;
  // the pair lists go behind the partial sums of the same workspace
  const int cw_max = cout < 128 ? cout : 128;
  void *pairs = (char *)workspace + wg_align((size_t)wg_chunks(n_rows, kvol, cin) * kvol * cin * cw_max * sizeof(float));
  const int rc = ls3d_spconv_pairs(tbl, row_order, n_rows, n_rows_dev, kvol, pairs, ls3d_spconv_pairs_bytes(kvol, n_rows), stream_);
  if (rc != LS3D_OK) return rc;
  return wg_on_pairs(in, in_ld, grad_out, go_ld, pairs, kvol, cin, cout, n_rows, products, workspace, grad_w, stream);
}

// the same on pair lists built once (ls3d_spconv_pairs) for every layer that shares the table: the SubM layers of a UNet level
// Identity pair lists (a Linear layer's weight gradient = one kernel offset, row i of x paired with row i of grad_out) for a row CAPACITY:
// built once per capacity, the number of valid rows set per use - the lists of every row count up to the capacity are prefixes of it.
__global__ __launch_bounds__(256) void k_pairs_identity(int n, int32_t *pin, int32_t *pout) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) pin[i] = pout[i] = i;
}
__global__ void k_pairs_set_count(int32_t *pair_cnt, int kvol, int count) {
  if ((int)threadIdx.x < kvol) pair_cnt[threadIdx.x] = count;
}
extern "C" int ls3d_spconv_identity_pairs(int n_rows_cap, int n_rows, int build, void *pairs, size_t pairs_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!pairs || n_rows_cap < 1 || n_rows < 0 || n_rows > n_rows_cap || ((uintptr_t)pairs & 15)) return LS3D_ERR_ARG;
  if (pairs_bytes < ls3d_spconv_pairs_bytes(1, n_rows_cap)) return LS3D_ERR_WORKSPACE;
  const int nb = (n_rows_cap + PL_BLK - 1) / PL_BLK;
  char *wsp = (char *)pairs;
  int32_t *pin = (int32_t *)wsp; wsp += wg_align((size_t)n_rows_cap * 4);
  int32_t *pout = (int32_t *)wsp; wsp += wg_align((size_t)n_rows_cap * 4);
  wsp += wg_align((size_t)nb * 4);
  int32_t *pair_cnt = (int32_t *)wsp;
  if (build) hipLaunchKernelGGL(k_pairs_identity, ls3d_grid(n_rows_cap), dim3(256), 0, stream, n_rows_cap, pin, pout);
  hipLaunchKernelGGL(k_pairs_set_count, dim3(1), dim3(64), 0, stream, pair_cnt, 1, n_rows);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_spconv_wgrad_on_pairs(const float *in, int in_ld, const float *grad_out, int go_ld, const void *pairs, int kvol, int cin, int cout,
                                          int n_rows, int products, void *workspace, size_t workspace_bytes, float *grad_w, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows == 0 && grad_w && kvol >= 1 && cin >= 1 && cout >= 1) {
    hipMemsetAsync(grad_w, 0, (size_t)kvol * cin * cout * sizeof(float), stream);
    return LS3D_OK;
  }
  if (!in || !grad_out || !pairs || !grad_w || !workspace || kvol < 1 || cin < 1 || cout < 1 || n_rows < 0) return LS3D_ERR_ARG;
  if (in_ld < cin || go_ld < cout) return LS3D_ERR_ARG;
  if ((products & 63) != 0 && (products & 63) != 6 && (products & 63) != 8) return LS3D_ERR_ARG;
  const int cw_max = cout < 128 ? cout : 128;
  if (workspace_bytes < wg_align((size_t)wg_chunks(n_rows, kvol, cin) * kvol * cin * cw_max * sizeof(float))) return LS3D_ERR_WORKSPACE;
  return wg_on_pairs(in, in_ld, grad_out, go_ld, pairs, kvol, cin, cout, n_rows, products, workspace, grad_w, stream);
}

// 32 x 32 output blocks per staged row group from which the plane kernel wins (products bit 6 forces it: A/B).  tools/probe_wgrad_sparse.py on
// the tables of a 2 x 180k Waymo batch: 64 -> 64 (4 blocks, 5.8 M pairs) exact f32 1.04 ms, planes 0.90 ms; 32 -> 32 (1 block) 0.10 vs 0.22 ms;
// 128 -> 128 (16 blocks) 1.88 vs 0.96 ms.  (Round 3 set the limit to 8 from the two outer points.)
constexpr int wg_plane_min_blocks = 4;
static int wg_on_pairs(const float *in, int in_ld, const float *grad_out, int go_ld, const void *pairs, int kvol, int cin, int cout, int n_rows,
                       int products, void *workspace, float *grad_w, hipStream_t stream) {
  const bool force_planes = (products & 64) != 0;
  products &= 63;
  const int nchunks = wg_chunks(n_rows, kvol, cin);
  float *partial = (float *)workspace;
  const int cout_all = cout;
  const int nb = (n_rows + PL_BLK - 1) / PL_BLK;
  const char *wsp = (const char *)pairs;
  const int32_t *tbl_t = (const int32_t *)wsp; wsp += wg_align((size_t)kvol * n_rows * 4);   // pin[k][j]
  const int32_t *o_t = (const int32_t *)wsp; wsp += wg_align((size_t)kvol * n_rows * 4);     // pout[k][j]
  wsp += wg_align((size_t)kvol * nb * 4);
  const int32_t *pair_cnt = (const int32_t *)wsp;
  const dim3 grid((unsigned)(nchunks * wg_ci_groups(cin)), (unsigned)kvol);
  const dim3 grid_lds((unsigned)(nchunks * ((cin + 127) / 128)), (unsigned)kvol);  // k_spconv_wgrad_lds: tiles of 128 input channels
  const int products_all = products;
  const float *grad_out_all = grad_out;
  // layers wider than 128 output columns (SCALING_RATIO > 2 of the reference's UNet): one pass per slab of 128 columns of grad_out over
  // the same transposed table; every pass reduces its chunks' partial sums into its columns of grad_w
  for (int col0 = 0; col0 < cout_all; col0 += 128) {
  cout = cout_all - col0 < 128 ? cout_all - col0 : 128;
  grad_out = grad_out_all + col0;
  products = products_all;
  const long long elems = (long long)kvol * cin * cout;
  const int cob = (cout + 31) / 32;
#define LS3D_WG(COB_) hipLaunchKernelGGL((k_spconv_wgrad<COB_>), grid, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, (const int32_t *)tbl_t, \
                                          (const int32_t *)o_t, kvol, cin, cout, n_rows, (const int32_t *)pair_cnt, nchunks, partial)
  // the plane kernel pays where a workgroup has many 32 x 32 output blocks per staged row group (measured on 241k rows: 128 -> 128 0.67 ms
  // against 1.84 ms; 32 -> 32 1.75 ms against 0.20 ms: one block leaves three waves idle behind the same staging cost) - narrower
  // layers keep the exact-f32 kernel, which is f32-grade by construction
  const int nblk = ((cin < 128 ? cin : 128) + 31) / 32 * cob;
  if (products != 0 && nblk < wg_plane_min_blocks && !force_planes) products = 0;
  if (products == 0) {
    if (cob == 1) LS3D_WG(1);
    else if (cob == 2) LS3D_WG(2);
    else LS3D_WG(4);
  } else if (products == 6) {
    hipLaunchKernelGGL((k_spconv_wgrad_lds<6>), grid_lds, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, (const int32_t *)tbl_t, (const int32_t *)o_t, kvol,
                       cin, cout, n_rows, (const int32_t *)pair_cnt, nchunks, partial);
  } else {
    hipLaunchKernelGGL((k_spconv_wgrad_lds<8>), grid_lds, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, (const int32_t *)tbl_t, (const int32_t *)o_t, kvol,
                       cin, cout, n_rows, (const int32_t *)pair_cnt, nchunks, partial);
  }
#undef LS3D_WG
  if (nchunks > 128)
    hipLaunchKernelGGL(k_wgrad_reduce_split, dim3((unsigned)((elems + 31) / 32)), dim3(256), 0, stream, (const float *)partial, nchunks, elems, cout,
                       cout_all, col0, grad_w);
  else
    hipLaunchKernelGGL(k_wgrad_reduce, ls3d_grid(elems), dim3(256), 0, stream, (const float *)partial, nchunks, elems, cout, cout_all, col0, grad_w);
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
