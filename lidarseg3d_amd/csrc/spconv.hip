// spconv.hip — the output-stationary gather-GEMM:  out[r] = epilogue( sum_k W[k]^T in[tbl[r][k]] ).
//
// One kernel for SubMConv3d / SparseConv3d / SparseInverseConv3d (tbl = output-major rulebook, rulebook.hip)
// and for every dense Linear layer on the path (tbl == NULL, kvol == 1).  Replaces spconv v1.x's
// per-offset gather -> mm -> scatter-add (call sites det3d/models/backbones/scn_unet.py:15-24).
//
// Mapping to CDNA4:
//   * workgroup = 4 waves = a tile of 128 output rows; wave w owns rows [32w, 32w+32) and ALL output
//     columns of its slab (NT accumulators of 32x32, v_mfma_f32_32x32x2_f32: exact f32, bit-equal to an fmaf
//     chain, so results only differ from the CPU oracle by summation order);
//   * A operand (gathered input rows) goes global -> VGPR directly, no LDS: the MFMA K index is a free
//     permutation, so lane (row, half) loads KC/2 CONTIGUOUS floats of its row (float4 loads, each 128 B row
//     chunk is consumed whole by two lanes) and step s pairs element s of both halves;
//   * B operand (weights of the current kernel offset, KC x slab chunk) is staged once per workgroup in LDS
//     (row-major, lane -> consecutive columns: conflict-free ds_read_b32) and shared by the 4 waves;
//   * kernel offsets with no active neighbour in the whole tile are skipped (block-uniform), waves whose 32
//     rows have none skip their MFMAs;
//   * epilogue fuses eval-BatchNorm (scale/shift), residual add, ReLU and the UNet decoder's
//     channel-reduction add, and writes 128 B row segments.
// Roofline: levels with C<=64 are HBM/L2-bound in the pair model (8-16 flop/B), C=128 sits at the f32-MFMA
// ridge; algorithmic bytes per layer = P*(Cin+Cout)*4 + P*8 + K*Cin*Cout*4.
#include "common.h"

#include "gemm_common.h"

// workgroup -> (tile, slab) mapping: ls3d_gather_gemm's per-call `flags` bits 0-1 (see k_gather_gemm); 0 = slabs of a tile share an
// XCD, tiles interleaved over the XCDs.  (Round 1's LDS-DMA pipelined variant k_gg_pipe - 10-20 % slower on every layer,
// profiles/round1_experiments.md - and its process-global switch are gone since round 3.)


// Template parameters
//   KC     K-chunk (input channels per LDS weight chunk): 32, or 16 when cin % 32 != 0
//   NT     32-column blocks per WAVE (accumulators per wave)
//   WC     waves along the columns; the 4 waves form a (4/WC) x WC grid, so a workgroup covers
//          TR = 32*(4/WC) rows x SLAB = 32*NT*WC columns.  WC > 1 trades tile height for width: more workgroups for
//          small levels WITHOUT splitting the columns across workgroups (which would gather every input row once
//          per column slab from L2/MALL) — the WC waves of a row group read the same A rows through the CU's L1.
//   SPARSE a rulebook table is present (sparse convolution); false = dense Linear.  A template parameter so that the
//          two show up as separate kernels in rocprof traces.
template <int KC, int NT, int WC, bool SPARSE>
__global__ __launch_bounds__(256, (NT == 1 && WC == 1) ? 5 : (NT == 2 && WC == 1) ? 4 : 2) void k_gather_gemm(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl,
                                                    const int32_t *__restrict__ order, int kvol, const float *__restrict__ w, int cin,
                                                    int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e,
                                                    float *__restrict__ out, int out_ld, int xcd_map) {
  constexpr int WR = 4 / WC;                           // waves along the rows
  constexpr int TR = 32 * WR;                          // rows per workgroup tile
  constexpr int SPL = KC / 2;                          // floats of a row chunk held per lane
  constexpr int WSLAB = NT * 32;                       // columns per wave
  constexpr int SLAB = WSLAB * WC;                     // columns per workgroup
  constexpr int PV = KC * WSLAB / 4;                   // float4s in one wave-slab weight piece
  constexpr int BV = PV * WC;                          // float4s in the workgroup's weight chunk
  constexpr int BPT = (BV + 255) / 256;                // float4s staged per thread
  struct alignas(NT == 3 ? 4 : 4 * NT) BVec { float v[NT]; };
  __shared__ __attribute__((aligned(16))) float Bs[2][KC * SLAB];  // double-buffered weight chunk, [piece][k][32][NT]
  __shared__ unsigned long long s_kmask;               // kernel offsets with an active neighbour in this tile
  __shared__ int s_rows[TR];                           // output row handled by each tile slot (-1 = none)
  __shared__ float s_stat[2 * 64];                     // per-row mean / rstd of the LayerNorm epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int col = lane & 31, kk = lane >> 5;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ntiles = (N + TR - 1) / TR;
  const int nwslab = w_ld / WSLAB;                     // packed: [kvol][wslab][cin][32][NT]
  const int nslab = w_ld / SLAB;                       // column slabs: one workgroup per (tile, slab)
  // Workgroup -> (tile, slab), 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch; speed only, never
  // correctness): the nslab column slabs of one tile get consecutive b/8, i.e. they run on the SAME XCD at about the
  // same time, so the rows the first slab gathers are L2 hits for the others.  xcd_map bit 0: each XCD takes a
  // contiguous range of tiles instead of every 8th; bit 1: slab-major order (all tiles of slab 0 first - the
  // pre-remap behaviour, kept for A/B measurements).
  const int tiles_per_xcd = (ntiles + 7) / 8;
  for (int b = blockIdx.x; b < tiles_per_xcd * 8 * nslab; b += gridDim.x) {
    int tile, slab;
    if (xcd_map & 2) {
      tile = b % (tiles_per_xcd * 8); slab = b / (tiles_per_xcd * 8);
    } else {
      const int xcd = b & 7, j = b >> 3;
      slab = j % nslab;
      tile = (xcd_map & 1) ? xcd * tiles_per_xcd + j / nslab : (j / nslab) * 8 + xcd;
    }
    if (tile >= ntiles) continue;
    const int n0 = slab * SLAB;
    const float *wbase = w + (size_t)slab * WC * cin * WSLAB;
    // ---- tile slots -> output rows.  With `order` (rows sorted by their neighbour bitmask, rulebook.hip) the 32
    //      rows of a wave share most of their empty kernel offsets, so the skips below remove most zero work.
    if (tid == 0) s_kmask = 0ull;
    if (tid < TR) {
      const int r = tile * TR + tid;
      s_rows[tid] = r < N ? (order ? order[r] : r) : -1;
    }
    __syncthreads();
    const int row = s_rows[wr * 32 + col];
    // ---- which kernel offsets does this tile / this wave need at all?
    unsigned long long wmask = 0ull;
    if (SPARSE) {
      for (int k = 0; k < kvol; ++k) {
        const int idx = (row >= 0) ? tbl[(size_t)row * kvol + k] : -1;
        if (__any(idx >= 0)) wmask |= 1ull << k;
      }
    } else {
      wmask = __any(row >= 0) ? 1ull : 0ull;
    }
    if (lane == 0 && wmask) atomicOr(&s_kmask, wmask);
    __syncthreads();
    unsigned long long rem = s_kmask;
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    if (rem) {
      // ---- software pipeline over the chunks (k, c0): while chunk i feeds the MFMAs, the gathered A rows and
      //      the weight chunk of i+1 are already in flight (registers / the other LDS buffer).
      //      Every load in the loop is unconditional (k_gather_gemm_bf16x3 below has the why): the neighbour index comes two offsets
      //      ahead from a clamped address, an absent neighbour's row reads valid bytes of the weights and is zeroed when the registers
      //      move up, the weight loads are clamped.
#define LS3D_POP_OFFSET(k)                       \
  {                                              \
    k = rem ? __ffsll((long long)rem) - 1 : -1;  \
    rem &= rem - 1;                              \
  }
      int k_cur, k_nxt, k_nn;
      LS3D_POP_OFFSET(k_cur) LS3D_POP_OFFSET(k_nxt) LS3D_POP_OFFSET(k_nn)
#define LS3D_LOAD_IDX(k) ((row >= 0) ? (SPARSE ? tbl[(size_t)row * kvol + (k)] : row) : -1)
#define LS3D_LOAD_IDX_RAW(k) (SPARSE ? tbl[(size_t)(row >= 0 ? row : 0) * kvol + ((k) >= 0 ? (k) : 0)] : row)
#define LS3D_LOAD_A(dst, valid, idx, c0_)                                                       \
  do {                                                                                          \
    valid = (idx) >= 0;                                                                         \
    const float4 *p_ = valid ? (const float4 *)(in + (size_t)(idx)*in_ld + (c0_) + kk * SPL) : (const float4 *)wbase; \
    _Pragma("unroll") for (int q = 0; q < SPL / 4; ++q) dst[q] = p_[q];                         \
  } while (0)
// weight staging registers are named scalars (not an array): an array indexed inside the pipelined loop is not
// promoted to registers by hipcc and ends up in scratch.  Float4 i_ of the chunk = piece (i_/PV), offset (i_%PV).
#define LS3D_B_ONE(j, reg, OP, UNCOND)                                                          \
  if constexpr (BPT > (j)) {                                                                    \
    const int i_ = tid + (j)*256;                                                               \
    if (UNCOND || BV % 256 == 0 || i_ < BV) { OP(reg, i_); }                                    \
  }
#define LS3D_B_LD(reg, i_)                                                                      \
  {                                                                                             \
    const int ic_ = (BV % 256 == 0 || (i_) < BV) ? (i_) : BV - 1; /* unconditional, clamped */  \
    reg = *(const float4 *)(wk_ + (size_t)(ic_ / PV) * cin * WSLAB + (size_t)(ic_ % PV) * 4);   \
  }
#define LS3D_B_ST(reg, i_) *(float4 *)(dst_ + (i_)*4) = reg
#define LS3D_LOAD_B(k, c0_)                                                                     \
  do {                                                                                          \
    const float *wk_ = wbase + ((size_t)(k)*cin * nwslab + (c0_)) * WSLAB;                      \
    LS3D_B_ONE(0, breg0, LS3D_B_LD, true) LS3D_B_ONE(1, breg1, LS3D_B_LD, true)                 \
    LS3D_B_ONE(2, breg2, LS3D_B_LD, true) LS3D_B_ONE(3, breg3, LS3D_B_LD, true)                 \
  } while (0)
#define LS3D_STORE_B(dst)                                                                       \
  do {                                                                                          \
    float *dst_ = (dst);                                                                        \
    LS3D_B_ONE(0, breg0, LS3D_B_ST, false) LS3D_B_ONE(1, breg1, LS3D_B_ST, false)               \
    LS3D_B_ONE(2, breg2, LS3D_B_ST, false) LS3D_B_ONE(3, breg3, LS3D_B_ST, false)               \
  } while (0)
      int idx_cur = LS3D_LOAD_IDX(k_cur);
      int idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
      int idx_nn = LS3D_LOAD_IDX_RAW(k_nn);
      float4 a_cur[SPL / 4], a_nxt[SPL / 4];
      bool va_cur = false, va_nxt = false;  // the row loaded into a_cur / a_nxt exists
      float4 breg0, breg1, breg2, breg3;
      static_assert(BPT <= 4, "weight chunk too large for the staging registers");
      int c0 = 0, buf = 0;
      LS3D_LOAD_A(a_cur, va_cur, idx_cur, 0);
      LS3D_LOAD_B(k_cur, 0);
      LS3D_STORE_B(Bs[0]);
#pragma unroll
      for (int q = 0; q < SPL / 4; ++q) a_cur[q] = va_cur ? a_cur[q] : make_float4(0.f, 0.f, 0.f, 0.f);  // (nothing in flight at the loop head)
      __syncthreads();
      for (;;) {
        int nk = k_cur, nc0 = c0 + KC, nidx = idx_cur;
        bool has_next = true;
        if (nc0 >= cin) {
          nc0 = 0; nk = k_nxt; nidx = idx_nxt;
          has_next = nk >= 0;
        }
        if (has_next) {
          LS3D_LOAD_A(a_nxt, va_nxt, nidx, nc0);
          LS3D_LOAD_B(nk, nc0);
        }
        if ((wmask >> k_cur) & 1ull) {
          // packed weight layout: the NT values a lane needs for one k-step are adjacent -> one ds_read of NT dwords;
          // the read for step s+1 is issued before the MFMAs of step s (register double buffer).
          const float *bs = Bs[buf] + wc * (KC * WSLAB) + (kk * SPL * 32 + col) * NT;
          BVec bb0, bb1;
          bb0 = *(const BVec *)bs;
#pragma unroll
          for (int q = 0; q < SPL / 4; ++q) {
#define LS3D_MFMA_STEP(u, aval, cur, nxt)                                                        \
  if (4 * q + (u) + 1 < SPL) nxt = *(const BVec *)(bs + (4 * q + (u) + 1) * 32 * NT);            \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                  \
      acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32((aval), cur.v[n], acc[n], 0, 0, 0);
            LS3D_MFMA_STEP(0, a_cur[q].x, bb0, bb1)
            LS3D_MFMA_STEP(1, a_cur[q].y, bb1, bb0)
            LS3D_MFMA_STEP(2, a_cur[q].z, bb0, bb1)
            LS3D_MFMA_STEP(3, a_cur[q].w, bb1, bb0)
#undef LS3D_MFMA_STEP
          }
        }
        if (!has_next) break;
        LS3D_STORE_B(Bs[buf ^ 1]);
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int q = 0; q < SPL / 4; ++q) a_cur[q] = va_nxt ? a_nxt[q] : make_float4(0.f, 0.f, 0.f, 0.f);  // (behind the wait for this chunk's loads)
        if (nk != k_cur) {
          k_cur = nk; idx_cur = idx_nxt;
          k_nxt = k_nn; idx_nxt = (k_nn >= 0 && row >= 0) ? idx_nn : -1;
          LS3D_POP_OFFSET(k_nn)
          idx_nn = LS3D_LOAD_IDX_RAW(k_nn);
        }
        c0 = nc0;
      }
    }
#undef LS3D_POP_OFFSET
#undef LS3D_LOAD_IDX_RAW
#undef LS3D_LOAD_IDX
#undef LS3D_LOAD_A
#undef LS3D_LOAD_B
#undef LS3D_B_ONE
#undef LS3D_B_LD
#undef LS3D_B_ST
#undef LS3D_STORE_B
    constexpr int RPP = (2 * KC < TR) ? 2 * KC : TR;  // tile rows that fit in the weight buffer per pass
    gg_epilogue<NT, WC, TR, RPP>(acc, &Bs[0][0], s_rows, s_stat, wr, wc, kk, col, n0, cout, e, out, out_ld);
  }
}

// ------------------------------------------------------------------------------------------------------------
// Split-bf16 ("bf16x3") variant: every f32 operand is split into a bf16 head and a bf16 tail (a = a_hi + a_lo,
// |a - a_hi - a_lo| <= 2^-16 |a|) and a*b is evaluated as a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation — 3 bf16 MFMAs (3 x 32 cycles per K=16) instead of 8 f32 MFMAs
// (8 x 64 cycles), i.e. 5.3x less matrix-pipe time, at ~1e-5 relative error per layer (the dropped a_lo*b_lo term
// and the tail's rounding).  Measured end to end on SDSeg3D: 1.4e-5 of the logit range (tests/…), i.e. well inside
// the 1e-3 budget for logits of magnitude <= 50.  Weights are split once at pack time; gathered rows are split in
// registers right after the load (truncated head, round-to-nearest tail: 4 VALU ops per element).
// Same tiling / pipeline / epilogue as k_gather_gemm (KC = 32, WC = 1).

// PL = number of bf16 planes per operand: 2 -> a*b ~ a0*b0 + a0*b1 + a1*b0 ("bf16x3", ~2^-16 relative per product),
//      3 -> a = a0 + a1 + a2 (24 mantissa bits: an exact split of an f32), a*b ~ the 6 products of weight <= 2^-16
//           ("bf16x6": dropped terms 2^-24 relative, i.e. the size of an f32 rounding error) = f32-grade results at
//           6 x 32 MFMA cycles per K=16 instead of 8 x 64.
template <int NT, bool SPARSE, int PL>
__global__ __launch_bounds__(256, (NT == 1 ? 4 : NT == 2 ? 3 : 2)) void k_gather_gemm_bf16x3(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl,
                                                           const int32_t *__restrict__ order, int kvol, const float *__restrict__ w,
                                                           int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e,
                                                           float *__restrict__ out, int out_ld, int xcd_map) {
  constexpr int KC = 32, TR = 128, SLAB = NT * 32;
  constexpr int BV = NT * 2 * PL * 64;   // 16-byte units in one weight chunk: [n][t][plane][kk][col] x (8 bf16)
  constexpr int BPT = (BV + 255) / 256;  // units staged per thread
  constexpr int CHF = BV * 4;            // floats per chunk
  __shared__ __attribute__((aligned(16))) float Bs[2][CHF];
  __shared__ unsigned long long s_kmask;
  __shared__ int s_rows[TR];
  __shared__ float s_stat[2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ntiles = (N + TR - 1) / TR;
  const int nslab = w_ld / SLAB;                       // packed: [kvol][slab][cin/32][chunk]
  // Workgroup -> (tile, slab), 1-D grid.  Workgroup b runs on XCD b % 8 (observed dispatch; speed only, never
  // correctness): the nslab column slabs of one tile get consecutive b/8, i.e. they run on the SAME XCD at about the
  // same time, so the rows the first slab gathers are L2 hits for the others.  xcd_map bit 0: each XCD takes a
  // contiguous range of tiles instead of every 8th; bit 1: slab-major order (all tiles of slab 0 first - the
  // pre-remap behaviour, kept for A/B measurements).
  const int tiles_per_xcd = (ntiles + 7) / 8;
  for (int b = blockIdx.x; b < tiles_per_xcd * 8 * nslab; b += gridDim.x) {
    int tile, slab;
    if (xcd_map & 2) {
      tile = b % (tiles_per_xcd * 8); slab = b / (tiles_per_xcd * 8);
    } else {
      const int xcd = b & 7, j = b >> 3;
      slab = j % nslab;
      tile = (xcd_map & 1) ? xcd * tiles_per_xcd + j / nslab : (j / nslab) * 8 + xcd;
    }
    if (tile >= ntiles) continue;
    const int n0 = slab * SLAB;
    const float *wbase = w + (size_t)slab * (cin / KC) * CHF;  // packed: [kvol][slab][cin/32][chunk]
    if (tid == 0) s_kmask = 0ull;
    if (tid < TR) {
      const int r = tile * TR + tid;
      s_rows[tid] = r < N ? (order ? order[r] : r) : -1;
    }
    __syncthreads();
    const int row = s_rows[wave * 32 + col];
    unsigned long long wmask = 0ull;
    if (SPARSE) {
      for (int k = 0; k < kvol; ++k) {
        const int idx = (row >= 0) ? tbl[(size_t)row * kvol + k] : -1;
        if (__any(idx >= 0)) wmask |= 1ull << k;
      }
    } else {
      wmask = __any(row >= 0) ? 1ull : 0ull;
    }
    if (lane == 0 && wmask) atomicOr(&s_kmask, wmask);
    __syncthreads();
    unsigned long long rem = s_kmask;
    f32x16 acc[NT];
    f32x16 acs[PL == 3 ? NT : 1];  // PL == 3: the five small products (head x head alone in `acc`: the bf16 MFMA's accumulate is biased, tileconv.hip)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
#pragma unroll
    for (int n = 0; n < (PL == 3 ? NT : 1); ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acs[n][r] = 0.0f;
    if (rem) {
      // active offsets are taken off `rem` lowest first; the neighbour index of an offset is loaded TWO offsets before its rows are
      // (k_nn / idx_nn, an unconditional load from a clamped address whose validity is applied when it moves up): loaded one offset
      // ahead and under its validity condition it was waited for in place
#define LS3D_LOAD_IDX_RAW(k) (SPARSE ? tbl[(size_t)(row >= 0 ? row : 0) * kvol + ((k) >= 0 ? (k) : 0)] : row)
#define LS3D_POP_OFFSET(k)                       \
  {                                              \
    k = rem ? __ffsll((long long)rem) - 1 : -1;  \
    rem &= rem - 1;                              \
  }
      int k_cur, k_nxt, k_nn;
      LS3D_POP_OFFSET(k_cur) LS3D_POP_OFFSET(k_nxt) LS3D_POP_OFFSET(k_nn)
#define LS3D_LOAD_IDX(k) ((row >= 0) ? (SPARSE ? tbl[(size_t)row * kvol + (k)] : row) : -1)
// the 16 floats of a row are loaded UNCONDITIONALLY: an absent neighbour reads 64 valid bytes of the packed weights instead and is zeroed when
// the registers move up (`valid`: at the end of a chunk, behind the wait that the weights need anyway).  Loads under a per-lane condition are loads hipcc cannot count on: it then waited for them in
// place - `s_waitcnt vmcnt(1)` right behind the four row loads, in front of the weight loads - and the gather latency of every chunk
// was exposed.
#define LS3D_LOAD_A(dst, valid, idx, c0_)                                                       \
  do {                                                                                          \
    valid = (idx) >= 0;                                                                         \
    const float4 *p_ = valid ? (const float4 *)(in + (size_t)(idx)*in_ld + (c0_) + kk * 16) : (const float4 *)wbase; \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) dst[q] = p_[q];                               \
  } while (0)
#define LS3D_B_ONE(j, reg, OP, UNCOND)                                                                \
  if constexpr (BPT > (j)) {                                                                    \
    const int i_ = tid + (j)*256;                                                               \
    if (UNCOND || BV % 256 == 0 || i_ < BV) { OP(reg, i_); }                                    \
  }
#define LS3D_B_LD(reg, i_) reg = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || (i_) < BV) ? (i_) : BV - 1) * 4) /* unconditional, clamped */
#define LS3D_B_ST(reg, i_) *(float4 *)(dst_ + (i_)*4) = reg
#define LS3D_LOAD_B(k, c0_)                                                                     \
  do {                                                                                          \
    const float *wk_ = wbase + ((size_t)(k) * nslab * (cin / KC) + (c0_) / KC) * CHF;           \
    LS3D_B_ONE(0, breg0, LS3D_B_LD, true) LS3D_B_ONE(1, breg1, LS3D_B_LD, true)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_LD, true) LS3D_B_ONE(3, breg3, LS3D_B_LD, true)                             \
    LS3D_B_ONE(4, breg4, LS3D_B_LD, true) LS3D_B_ONE(5, breg5, LS3D_B_LD, true)                             \
  } while (0)
#define LS3D_STORE_B(dst)                                                                       \
  do {                                                                                          \
    float *dst_ = (dst);                                                                        \
    LS3D_B_ONE(0, breg0, LS3D_B_ST, false) LS3D_B_ONE(1, breg1, LS3D_B_ST, false)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_ST, false) LS3D_B_ONE(3, breg3, LS3D_B_ST, false)                             \
    LS3D_B_ONE(4, breg4, LS3D_B_ST, false) LS3D_B_ONE(5, breg5, LS3D_B_ST, false)                             \
  } while (0)
      int idx_cur = LS3D_LOAD_IDX(k_cur);
      int idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
      int idx_nn = LS3D_LOAD_IDX_RAW(k_nn);
      float4 a_cur[4], a_nxt[4];
      bool va_cur = false, va_nxt = false;  // the row loaded into a_cur / a_nxt exists
      float4 breg0, breg1, breg2, breg3, breg4, breg5;
      static_assert(BPT <= 6, "weight chunk too large for the staging registers");
      int c0 = 0, buf = 0;
      LS3D_LOAD_A(a_cur, va_cur, idx_cur, 0);
      LS3D_LOAD_B(k_cur, 0);
      LS3D_STORE_B(Bs[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) a_cur[q] = va_cur ? a_cur[q] : make_float4(0.f, 0.f, 0.f, 0.f);  // (nothing in flight at the loop head)
      __syncthreads();
      for (;;) {
        int nk = k_cur, nc0 = c0 + KC, nidx = idx_cur;
        bool has_next = true;
        if (nc0 >= cin) {
          nc0 = 0; nk = k_nxt; nidx = idx_nxt;
          has_next = nk >= 0;
        }
        if (has_next) {
          LS3D_LOAD_A(a_nxt, va_nxt, nidx, nc0);
          LS3D_LOAD_B(nk, nc0);
        }
        if ((wmask >> k_cur) & 1ull) {
          // planes of this lane's 16 floats: k-step 0 uses floats 0-7, k-step 1 floats 8-15
          const uint4 *bs = (const uint4 *)Bs[buf] + kk * 32 + col;  // unit index (((n*2+t)*PL+plane)*2+kk)*32+col
          if constexpr (PL == 2) {
            uint4 ah0, al0, ah1, al1;
            ls3d_split8(a_cur[0], a_cur[1], ah0, al0);
            ls3d_split8(a_cur[2], a_cur[3], ah1, al1);
            const bf16x8 vah0 = __builtin_bit_cast(bf16x8, ah0), val0 = __builtin_bit_cast(bf16x8, al0);
            const bf16x8 vah1 = __builtin_bit_cast(bf16x8, ah1), val1 = __builtin_bit_cast(bf16x8, al1);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const bf16x8 bh0 = __builtin_bit_cast(bf16x8, bs[((n * 2 + 0) * 2 + 0) * 64]);
              const bf16x8 bl0 = __builtin_bit_cast(bf16x8, bs[((n * 2 + 0) * 2 + 1) * 64]);
              const bf16x8 bh1 = __builtin_bit_cast(bf16x8, bs[((n * 2 + 1) * 2 + 0) * 64]);
              const bf16x8 bl1 = __builtin_bit_cast(bf16x8, bs[((n * 2 + 1) * 2 + 1) * 64]);
              // small terms first
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val0, bh0, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah0, bl0, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val1, bh1, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah1, bl1, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah0, bh0, acc[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah1, bh1, acc[n], 0, 0, 0);
            }
          } else {
            uint4 a0[3], a1[3];  // [plane]: head / middle / tail, round-to-nearest planes (the products left out are zero-mean)
            ls3d_split_pair3_rne(a_cur[0].x, a_cur[0].y, a0[0].x, a0[1].x, a0[2].x);
            ls3d_split_pair3_rne(a_cur[0].z, a_cur[0].w, a0[0].y, a0[1].y, a0[2].y);
            ls3d_split_pair3_rne(a_cur[1].x, a_cur[1].y, a0[0].z, a0[1].z, a0[2].z);
            ls3d_split_pair3_rne(a_cur[1].z, a_cur[1].w, a0[0].w, a0[1].w, a0[2].w);
            ls3d_split_pair3_rne(a_cur[2].x, a_cur[2].y, a1[0].x, a1[1].x, a1[2].x);
            ls3d_split_pair3_rne(a_cur[2].z, a_cur[2].w, a1[0].y, a1[1].y, a1[2].y);
            ls3d_split_pair3_rne(a_cur[3].x, a_cur[3].y, a1[0].z, a1[1].z, a1[2].z);
            ls3d_split_pair3_rne(a_cur[3].z, a_cur[3].w, a1[0].w, a1[1].w, a1[2].w);
#pragma unroll
            for (int n = 0; n < NT; ++n) {
#pragma unroll
              for (int t = 0; t < 2; ++t) {
                const bf16x8 ah = __builtin_bit_cast(bf16x8, t ? a1[0] : a0[0]), am = __builtin_bit_cast(bf16x8, t ? a1[1] : a0[1]);
                const bf16x8 al = __builtin_bit_cast(bf16x8, t ? a1[2] : a0[2]);
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 0) * 64]);
                const bf16x8 bm = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 1) * 64]);
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 2) * 64]);
                // the six products of weight >= 2^-16, smallest first
                acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acs[n], 0, 0, 0);
                acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acs[n], 0, 0, 0);
                acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acs[n], 0, 0, 0);
                acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acs[n], 0, 0, 0);
                acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acs[n], 0, 0, 0);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
              }
            }
          }
        }
        if (!has_next) break;
        LS3D_STORE_B(Bs[buf ^ 1]);
        __syncthreads();
        buf ^= 1;
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = va_nxt ? a_nxt[q] : make_float4(0.f, 0.f, 0.f, 0.f);  // (behind the wait for this chunk's loads)
        if (nk != k_cur) {
          k_cur = nk; idx_cur = idx_nxt;
          k_nxt = k_nn; idx_nxt = (k_nn >= 0 && row >= 0) ? idx_nn : -1;
          LS3D_POP_OFFSET(k_nn)
          idx_nn = LS3D_LOAD_IDX_RAW(k_nn);
        }
        c0 = nc0;
      }
    }
#undef LS3D_POP_OFFSET
#undef LS3D_LOAD_IDX_RAW
#undef LS3D_LOAD_IDX
#undef LS3D_LOAD_A
#undef LS3D_LOAD_B
#undef LS3D_B_ONE
#undef LS3D_B_LD
#undef LS3D_B_ST
#undef LS3D_STORE_B
    if constexpr (PL == 3) {
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] += acs[n][r];
    }
    gg_epilogue<NT, 1, TR, 64>(acc, &Bs[0][0], s_rows, s_stat, wave, 0, kk, col, n0, cout, e, out, out_ld);
  }
}

// ------------------------------------------------------------------------------------------------------------
// The same 6-product operator for the sparse layers (strided / inverse convolutions of the 3-plane modes), pipelined TWO stages deep.
// k_gather_gemm_bf16x3 issues the gathered rows and the weight chunk of stage i + 1 at the top of stage i and needs both at its end:
// a stage's MFMAs (12 - 24 x 32 cycles) are a fraction of an L2 gather round trip, so every stage of the seven layers waited
// for memory (tools/asm_waits.py: vmcnt(0) in front of every step barrier; conv2.0 of the 120k frame: 27 stages x 2.2 us per tile).
// Here the gathered rows of stage i + 2 are issued while stage i multiplies (three row sets in registers); the weight chunk of stage i + 1 -
// a contiguous L2-resident read - is issued in FRONT of them and goes to the other LDS buffer behind stage i.  The neighbour indices of the tile (27 x 128 ints) are read once into LDS - the prologue needs them
// all for the offset masks anyway - so a stage's row addresses cost one LDS read instead of an index load two offsets ahead.
// Same arithmetic, same summation order as k_gather_gemm_bf16x3<NT, true, 3>: bit-identical results.
template <int NT>
__global__ __launch_bounds__(256, (NT == 1 ? 4 : NT == 2 ? 3 : 2)) void k_gather_gemm_x6(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl,
                                                           const int32_t *__restrict__ order, int kvol, const float *__restrict__ w,
                                                           int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e,
                                                           float *__restrict__ out, int out_ld, int xcd_map) {
  constexpr int KC = 32, TR = 128, SLAB = NT * 32, PL = 3;
  constexpr int BV = NT * 2 * PL * 64;   // 16-byte units in one weight chunk: [n][t][plane][kk][col] x (8 bf16)
  constexpr int BPT = (BV + 255) / 256;  // units staged per thread
  constexpr int CHF = BV * 4;            // floats per chunk
  __shared__ __attribute__((aligned(16))) float Bs[2][CHF];
  __shared__ int s_idx[27][TR];          // neighbour row of (kernel offset, tile slot), -1 = none (kvol <= 27: 3 x 3 x 3 kernels)
  __shared__ unsigned long long s_kmask;
  __shared__ int s_rows[TR];
  __shared__ float s_stat[2 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ntiles = (N + TR - 1) / TR;
  const int nslab = w_ld / SLAB;
  const int nchunk = cin / KC;
  const int tiles_per_xcd = (ntiles + 7) / 8;
  for (int b = blockIdx.x; b < tiles_per_xcd * 8 * nslab; b += gridDim.x) {
    int tile, slab;
    if (xcd_map & 2) {
      tile = b % (tiles_per_xcd * 8); slab = b / (tiles_per_xcd * 8);
    } else {
      const int xcd = b & 7, j = b >> 3;
      slab = j % nslab;
      tile = (xcd_map & 1) ? xcd * tiles_per_xcd + j / nslab : (j / nslab) * 8 + xcd;
    }
    if (tile >= ntiles) continue;
    const int n0 = slab * SLAB;
    const float *wbase = w + (size_t)slab * nchunk * CHF;  // packed: [kvol][slab][cin/32][chunk]
    __syncthreads();  // the previous tile's epilogue is done with s_rows / Bs
    if (tid == 0) s_kmask = 0ull;
    if (tid < TR) {
      const int r = tile * TR + tid;
      s_rows[tid] = r < N ? (order ? order[r] : r) : -1;
    }
    __syncthreads();
    {  // the tile's table rows -> LDS: two threads per slot, every other offset each
      const int slot = tid & (TR - 1), part = tid >> 7;
      const int row = s_rows[slot];
      const int32_t *trow = tbl + (size_t)(row >= 0 ? row : 0) * kvol;
      for (int k = part; k < kvol; k += 2) s_idx[k][slot] = row >= 0 ? trow[k] : -1;
    }
    __syncthreads();
    unsigned long long wmask = 0ull;
    for (int k = 0; k < kvol; ++k)
      if (__any(s_idx[k][wave * 32 + col] >= 0)) wmask |= 1ull << k;
    if (lane == 0 && wmask) atomicOr(&s_kmask, wmask);
    __syncthreads();
    unsigned long long rem = s_kmask;
    f32x16 acc[NT], acs[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[n][r] = 0.0f; acs[n][r] = 0.0f; }
    if (rem) {
      // stage = (kernel offset k, 32-channel chunk c0); k = -1: past the end.  The sequence is the same for the four waves (the weights are shared).
      int k0, c00 = 0, k1, c01, k2, c02;
#define GX_POP(k_) { k_ = rem ? __ffsll((long long)rem) - 1 : -1; rem &= rem - 1; }
#define GX_NEXT(kn_, cn_, k_, c_) { kn_ = k_; cn_ = c_ + KC; if (cn_ >= cin) { cn_ = 0; GX_POP(kn_) } }
      // rows of stage (k_, c_): unconditional loads; an absent neighbour (or a stage past the end) reads valid bytes of the weights and is
      // zeroed when the registers are split
#define GX_LOAD_A(dst_, ok_, k_, c_)                                                              \
  {                                                                                               \
    const int idx_ = (k_) >= 0 ? s_idx[(k_)][wave * 32 + col] : -1;                               \
    ok_ = idx_ >= 0;                                                                              \
    const float4 *p_ = ok_ ? (const float4 *)(in + (size_t)idx_ * in_ld + (c_) + kk * 16) : (const float4 *)wbase; \
    dst_##_0 = p_[0]; dst_##_1 = p_[1]; dst_##_2 = p_[2]; dst_##_3 = p_[3];                       \
  }
// (weight staging registers are named members, not an array: an array inside the pipelined loop ends up in scratch)
#define GX_B_ONE(j_, OP_) if constexpr (BPT > (j_)) { const int i_ = tid + (j_) * 256; OP_ }
#define GX_LOAD_B(dst_, k_, c_)                                                                   \
  {                                                                                               \
    const float *wk_ = wbase + ((size_t)((k_) >= 0 ? (k_) : 0) * nslab * nchunk + (c_) / KC) * CHF; \
    GX_B_ONE(0, dst_##_0 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
    GX_B_ONE(1, dst_##_1 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
    GX_B_ONE(2, dst_##_2 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
    GX_B_ONE(3, dst_##_3 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
    GX_B_ONE(4, dst_##_4 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
    GX_B_ONE(5, dst_##_5 = *(const float4 *)(wk_ + (size_t)((BV % 256 == 0 || i_ < BV) ? i_ : BV - 1) * 4);) \
  }
#define GX_STORE_B(buf_, src_)                                                                    \
  {                                                                                               \
    GX_B_ONE(0, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_0;)          \
    GX_B_ONE(1, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_1;)          \
    GX_B_ONE(2, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_2;)          \
    GX_B_ONE(3, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_3;)          \
    GX_B_ONE(4, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_4;)          \
    GX_B_ONE(5, if (BV % 256 == 0 || i_ < BV) *(float4 *)(Bs[buf_] + i_ * 4) = src_##_5;)          \
  }
      GX_POP(k0)
      GX_NEXT(k1, c01, k0, c00)
      float4 a0_0, a0_1, a0_2, a0_3, a1_0, a1_1, a1_2, a1_3, a2_0, a2_1, a2_2, a2_3;  // three row sets (named: arrays / structs land in scratch here)
      bool ok0, ok1, ok2;
      float4 b1_0, b1_1, b1_2, b1_3, b1_4, b1_5;  // the next stage's weight chunk (named registers)
      static_assert(BPT <= 6, "weight chunk too large for the staging registers");
      GX_LOAD_A(a0, ok0, k0, c00)
      GX_LOAD_B(b1, k0, c00)
      GX_STORE_B(0, b1)
      GX_LOAD_A(a1, ok1, k1, c01)
      __syncthreads();
      int buf = 0;
      for (;;) {
        GX_NEXT(k2, c02, k1, c01)
        if (k1 < 0) { k2 = -1; c02 = 0; }
        // the weight chunk of stage i + 1 first (contiguous, L2-resident: one stage of flight is enough and it is waited for at the end of this
        // stage), the gathered rows of stage i + 2 behind it (they stay in flight across that wait: vmcnt counts in order)
        GX_LOAD_B(b1, k1, c01)
        GX_LOAD_A(a2, ok2, k2, c02)
        if ((wmask >> k0) & 1ull) {
          const uint4 *bs = (const uint4 *)Bs[buf] + kk * 32 + col;  // unit index (((n*2+t)*PL+plane)*2+kk)*32+col
          uint4 x0[3], x1[3];  // [plane]: head / middle / tail, round-to-nearest planes
          const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 r0 = ok0 ? a0_0 : z, r1 = ok0 ? a0_1 : z, r2 = ok0 ? a0_2 : z, r3 = ok0 ? a0_3 : z;
          ls3d_split_pair3_rne(r0.x, r0.y, x0[0].x, x0[1].x, x0[2].x);
          ls3d_split_pair3_rne(r0.z, r0.w, x0[0].y, x0[1].y, x0[2].y);
          ls3d_split_pair3_rne(r1.x, r1.y, x0[0].z, x0[1].z, x0[2].z);
          ls3d_split_pair3_rne(r1.z, r1.w, x0[0].w, x0[1].w, x0[2].w);
          ls3d_split_pair3_rne(r2.x, r2.y, x1[0].x, x1[1].x, x1[2].x);
          ls3d_split_pair3_rne(r2.z, r2.w, x1[0].y, x1[1].y, x1[2].y);
          ls3d_split_pair3_rne(r3.x, r3.y, x1[0].z, x1[1].z, x1[2].z);
          ls3d_split_pair3_rne(r3.z, r3.w, x1[0].w, x1[1].w, x1[2].w);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const bf16x8 ah = __builtin_bit_cast(bf16x8, t ? x1[0] : x0[0]), am = __builtin_bit_cast(bf16x8, t ? x1[1] : x0[1]);
              const bf16x8 al = __builtin_bit_cast(bf16x8, t ? x1[2] : x0[2]);
              const bf16x8 bh = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 0) * 64]);
              const bf16x8 bm = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 1) * 64]);
              const bf16x8 bl = __builtin_bit_cast(bf16x8, bs[((n * 2 + t) * 3 + 2) * 64]);
              // the six products of weight >= 2^-16, smallest first (the order of k_gather_gemm_bf16x3)
              acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acs[n], 0, 0, 0);
              acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acs[n], 0, 0, 0);
              acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acs[n], 0, 0, 0);
              acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acs[n], 0, 0, 0);
              acs[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acs[n], 0, 0, 0);
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[n], 0, 0, 0);
            }
          }
        }
        if (k1 < 0) break;
        GX_STORE_B(buf ^ 1, b1)
        __syncthreads();
        buf ^= 1;
        a0_0 = a1_0; a0_1 = a1_1; a0_2 = a1_2; a0_3 = a1_3; ok0 = ok1;
        a1_0 = a2_0; a1_1 = a2_1; a1_2 = a2_2; a1_3 = a2_3; ok1 = ok2;
        k0 = k1; c00 = c01; k1 = k2; c01 = c02;
      }
#undef GX_POP
#undef GX_NEXT
#undef GX_LOAD_A
#undef GX_LOAD_B
#undef GX_STORE_B
#undef GX_B_ONE
    }
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] += acs[n][r];
    gg_epilogue<NT, 1, TR, 64>(acc, &Bs[0][0], s_rows, s_stat, wave, 0, kk, col, n0, cout, e, out, out_ld);
  }
}

// plain [kvol][cin_src][cout] -> split-bf16 packed [kvol][slab][cin_pad/32][n][t][hi/lo][kk][col] x 8 bf16
__global__ __launch_bounds__(256) void k_gg_pack_bf16x3(const float *src, int kvol, int cin_src, int cin_pad, int cout, int nt, int pl, uint4 *dst) {
  const int slab = nt * 32, nslab = ((cout + 31) / 32) / nt, nchunk = cin_pad / 32;
  const long long total = (long long)kvol * nslab * nchunk * nt * 2 * pl * 64;  // 16-byte units
  for (long long t_ = (long long)blockIdx.x * blockDim.x + threadIdx.x; t_ < total; t_ += (long long)gridDim.x * blockDim.x) {
    long long r = t_;
    const int col = (int)(r % 32); r /= 32;
    const int kk = (int)(r % 2); r /= 2;
    const int h = (int)(r % pl); r /= pl;
    const int t = (int)(r % 2); r /= 2;
    const int n = (int)(r % nt); r /= nt;
    const int ch = (int)(r % nchunk); r /= nchunk;
    const int sl = (int)(r % nslab); r /= nslab;
    const int k = (int)r;
    const int oc = sl * slab + n * 32 + col;
    unsigned wds[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      unsigned half[2];
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int c = ch * 32 + kk * 16 + t * 8 + pr * 2 + e2;
        const float v = (c < cin_src && oc < cout) ? src[((size_t)k * cin_src + c) * cout + oc] : 0.0f;
        if (pl == 2) {  // head rounded to nearest, tail = the rest
          const unsigned hb = ls3d_bf16_rne(v);
          half[e2] = h == 0 ? hb : ls3d_bf16_rne(v - __uint_as_float(hb << 16));
        } else {        // exact 3-way split (the same one the kernel applies to the gathered rows)
          const unsigned hb = __float_as_uint(v) & 0xFFFF0000u;
          const float r1 = v - __uint_as_float(hb);
          const unsigned mb = __float_as_uint(r1) & 0xFFFF0000u;
          half[e2] = h == 0 ? (hb >> 16) : h == 1 ? (mb >> 16) : ls3d_bf16_rne(r1 - __uint_as_float(mb));
        }
      }
      wds[pr] = half[0] | (half[1] << 16);
    }
    uint4 o;
    o.x = wds[0]; o.y = wds[1]; o.z = wds[2]; o.w = wds[3];
    dst[t_] = o;
  }
}

template <int NT, int PL>
static void launch_gg3(dim3 grid, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, const int32_t *order, int kvol,
                       const float *w, int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld,
                       int g_xcd_map) {
  if (tbl)
    hipLaunchKernelGGL((k_gather_gemm_bf16x3<NT, true, PL>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout, n_rows,
                       n_rows_dev, e, out, out_ld, g_xcd_map);
  else
    hipLaunchKernelGGL((k_gather_gemm_bf16x3<NT, false, PL>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout, n_rows,
                       n_rows_dev, e, out, out_ld, g_xcd_map);
}

template <int KC, int NT, int WC>
static void launch_gg(dim3 grid, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, const int32_t *order, int kvol,
                      const float *w, int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld,
                      int g_xcd_map) {
  if (tbl)
    hipLaunchKernelGGL((k_gather_gemm<KC, NT, WC, true>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout,
                       n_rows, n_rows_dev, e, out, out_ld, g_xcd_map);
  else
    hipLaunchKernelGGL((k_gather_gemm<KC, NT, WC, false>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout,
                       n_rows, n_rows_dev, e, out, out_ld, g_xcd_map);
}

// column-block decomposition shared by the kernel dispatch and the weight packer
static inline int gg_nt(int cout) {
  const int nt_total = (cout + 31) / 32;
  for (int c = 4; c >= 1; --c)
    if (nt_total % c == 0) return c;
  return 1;
}

// plain [kvol][cin_src][cout] -> packed [kvol][slab][cin_pad][32][NT] (zero padded), see ls3d.h
__global__ __launch_bounds__(256) void k_gg_pack(const float *src, int kvol, int cin_src, int cin_pad, int cout, int nt, float *dst) {
  const int slab = nt * 32, nslab = ((cout + 31) / 32) / nt;
  const long long total = (long long)kvol * nslab * cin_pad * slab;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    long long r = t;
    const int n = (int)(r % nt); r /= nt;
    const int col = (int)(r % 32); r /= 32;
    const int c = (int)(r % cin_pad); r /= cin_pad;
    const int sl = (int)(r % nslab); r /= nslab;
    const int k = (int)r;
    const int oc = sl * slab + n * 32 + col;
    dst[t] = (c < cin_src && oc < cout) ? src[((size_t)k * cin_src + c) * cout + oc] : 0.0f;
  }
}

extern "C" size_t ls3d_gather_gemm_packed_floats(int kvol, int cin_pad, int cout) {
  // sized for the largest layout (LS3D_PRECISION_BF16X6: three bf16 planes = 6 bytes per weight)
  return (size_t)kvol * cin_pad * ((cout + 31) / 32 * 32) * 3 / 2;
}

extern "C" int ls3d_gather_gemm_default_nt(int cout) { return gg_nt(cout); }

static inline bool gg_nt_ok(int cout, int nt) { return nt >= 1 && nt <= 4 && (((cout + 31) / 32) % nt) == 0; }

extern "C" int ls3d_gather_gemm_pack(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, int nt, int precision,
                                     float *w_packed, ls3d_stream_t stream) {
  if (!w_plain || !w_packed || kvol < 1 || cin_src < 1 || cin_pad < cin_src || (cin_pad % 16) || cout < 1) return LS3D_ERR_ARG;
  if (nt == 0) nt = gg_nt(cout);
  if (!gg_nt_ok(cout, nt)) return LS3D_ERR_ARG;
  const long long total = (long long)kvol * cin_pad * ((cout + 31) / 32 * 32);  // weights incl. padding
  if (precision == LS3D_PRECISION_BF16X3 || precision == LS3D_PRECISION_BF16X6) {
    if (cin_pad % 32) return LS3D_ERR_ARG;
    const int pl = precision == LS3D_PRECISION_BF16X3 ? 2 : 3;
    hipLaunchKernelGGL(k_gg_pack_bf16x3, ls3d_grid(total * pl / 8), dim3(256), 0, (hipStream_t)stream, w_plain, kvol, cin_src, cin_pad, cout, nt, pl,
                       (uint4 *)w_packed);
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  if (precision != LS3D_PRECISION_F32) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_gg_pack, ls3d_grid(total), dim3(256), 0, (hipStream_t)stream, w_plain, kvol, cin_src, cin_pad, cout, nt, w_packed);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_gather_gemm(const float *in, int in_ld, const int32_t *tbl, const int32_t *row_order, int kvol, const float *w, int nt,
                                int wc, int precision, int cin, int cout, int n_rows, const int32_t *n_rows_dev,
                                const ls3d_epilogue_t *epi, float *out, int out_ld, int flags, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  const int xcd_map = flags & 3;
  if (!in || !w || !out || n_rows < 0 || kvol < 1 || cin < 16 || cout < 1) return LS3D_ERR_ARG;
  if ((cin % 16) || (in_ld % 4) || in_ld < cin || out_ld < cout) return LS3D_ERR_ARG;
  if ((!tbl && kvol != 1) || kvol > 64) return LS3D_ERR_ARG;
  if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return LS3D_ERR_ARG;
  if (n_rows == 0) return LS3D_OK;
  EpiDev e = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.0f};
  if (epi) {
    e.scale = epi->scale; e.shift = epi->shift; e.res_pre = epi->res_pre; e.pair = epi->pair;
    e.res_pre_ld = epi->res_pre_ld; e.pair_ld = epi->pair_ld; e.relu = epi->relu;
    e.ln_gamma = epi->ln_gamma; e.ln_beta = epi->ln_beta; e.ln_eps = epi->ln_eps;
    if ((e.ln_gamma != nullptr) != (e.ln_beta != nullptr) || (e.ln_gamma && e.pair)) return LS3D_ERR_ARG;
  }
  const int w_ld = (cout + 31) / 32 * 32;
  const int nt_total = w_ld / 32;
  if (nt == 0) nt = gg_nt(cout);
  if (wc == 0) wc = 1;
  if (!gg_nt_ok(cout, nt) || (wc != 1 && wc != 2 && wc != 4) || (nt_total % (nt * wc)) || nt * wc > 4) return LS3D_ERR_ARG;
  const int slabs = nt_total / (nt * wc);
  if (e.ln_gamma && slabs != 1) return LS3D_ERR_ARG;  // the LayerNorm epilogue needs the whole row in one workgroup
  const int tr = 32 * (4 / wc);
  const int ntiles = (n_rows + tr - 1) / tr;
  const long long nwg = (long long)((ntiles + 7) / 8) * 8 * slabs;  // one workgroup per (tile, slab), see the kernels
  dim3 grid((unsigned)(nwg < (1 << 20) ? nwg : (1 << 20)));
  if (precision == LS3D_PRECISION_BF16X3 || precision == LS3D_PRECISION_BF16X6) {
    if ((cin % 32) || wc != 1) return LS3D_ERR_ARG;
#define LS3D_GG3(NT_, PL_) launch_gg3<NT_, PL_>(grid, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld, xcd_map)
    if (precision == LS3D_PRECISION_BF16X3) {
      switch (nt) {
        case 1: LS3D_GG3(1, 2); break;
        case 2: LS3D_GG3(2, 2); break;
        case 3: LS3D_GG3(3, 2); break;
        default: LS3D_GG3(4, 2);
      }
    } else if (tbl && kvol <= 27 && nt != 3 && !(flags & 4)) {
      // sparse layers of the 3-plane modes: the two-stages-deep pipeline (flags bit 2: the one-stage kernel, for A/B runs; bit-identical)
#define LS3D_GX6(NT_) hipLaunchKernelGGL((k_gather_gemm_x6<NT_>), grid, dim3(256), 0, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld, xcd_map)
      switch (nt) {
        case 1: LS3D_GX6(1); break;
        case 2: LS3D_GX6(2); break;
        default: LS3D_GX6(4);
      }
#undef LS3D_GX6
    } else {
      switch (nt) {
        case 1: LS3D_GG3(1, 3); break;
        case 2: LS3D_GG3(2, 3); break;
        case 3: LS3D_GG3(3, 3); break;
        default: LS3D_GG3(4, 3);
      }
    }
#undef LS3D_GG3
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  if (precision != LS3D_PRECISION_F32) return LS3D_ERR_ARG;
  const bool k32 = (cin % 32) == 0;
#define LS3D_GG(KC, NT, WC) launch_gg<KC, NT, WC>(grid, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld, xcd_map)
#define LS3D_GG_K(KC)                                              \
  switch (nt * 10 + wc) {                                          \
    case 11: LS3D_GG(KC, 1, 1); break;                             \
    case 12: LS3D_GG(KC, 1, 2); break;                             \
    case 14: LS3D_GG(KC, 1, 4); break;                             \
    case 21: LS3D_GG(KC, 2, 1); break;                             \
    case 22: LS3D_GG(KC, 2, 2); break;                             \
    case 31: LS3D_GG(KC, 3, 1); break;                             \
    case 41: LS3D_GG(KC, 4, 1); break;                             \
    default: return LS3D_ERR_ARG;                                  \
  }
  if (k32) { LS3D_GG_K(32) } else { LS3D_GG_K(16) }
#undef LS3D_GG_K
#undef LS3D_GG
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
