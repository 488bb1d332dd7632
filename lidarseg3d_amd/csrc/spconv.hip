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

static int g_xcd_map = 0;  // workgroup -> (tile, slab) mapping flags (see k_gather_gemm); 0 = slabs of a tile share an XCD, tiles
                           // interleaved over the XCDs.  LS3D_XCD_MAP=<flags> selects the alternatives for A/B measurements
extern "C" void ls3d_set_xcd_map(int on) { g_xcd_map = on & 3; }


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
      int k_cur = __ffsll((long long)rem) - 1;
      rem &= rem - 1;
      int k_nxt = rem ? __ffsll((long long)rem) - 1 : -1;
#define LS3D_LOAD_IDX(k) ((row >= 0) ? (SPARSE ? tbl[(size_t)row * kvol + (k)] : row) : -1)
#define LS3D_LOAD_A(dst, idx, c0_)                                                              \
  do {                                                                                          \
    if ((idx) >= 0) {                                                                           \
      const float4 *p_ = (const float4 *)(in + (size_t)(idx)*in_ld + (c0_) + kk * SPL);         \
      _Pragma("unroll") for (int q = 0; q < SPL / 4; ++q) dst[q] = p_[q];                       \
    } else {                                                                                    \
      _Pragma("unroll") for (int q = 0; q < SPL / 4; ++q) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f); \
    }                                                                                           \
  } while (0)
// weight staging registers are named scalars (not an array): an array indexed inside the pipelined loop is not
// promoted to registers by hipcc and ends up in scratch.  Float4 i_ of the chunk = piece (i_/PV), offset (i_%PV).
#define LS3D_B_ONE(j, reg, OP)                                                                  \
  if constexpr (BPT > (j)) {                                                                    \
    const int i_ = tid + (j)*256;                                                               \
    if (BV % 256 == 0 || i_ < BV) { OP(reg, i_); }                                              \
  }
#define LS3D_B_LD(reg, i_) reg = *(const float4 *)(wk_ + (size_t)((i_) / PV) * cin * WSLAB + (size_t)((i_) % PV) * 4)
#define LS3D_B_ST(reg, i_) *(float4 *)(dst_ + (i_)*4) = reg
#define LS3D_LOAD_B(k, c0_)                                                                     \
  do {                                                                                          \
    const float *wk_ = wbase + ((size_t)(k)*cin * nwslab + (c0_)) * WSLAB;                      \
    LS3D_B_ONE(0, breg0, LS3D_B_LD) LS3D_B_ONE(1, breg1, LS3D_B_LD)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_LD) LS3D_B_ONE(3, breg3, LS3D_B_LD)                             \
  } while (0)
#define LS3D_STORE_B(dst)                                                                       \
  do {                                                                                          \
    float *dst_ = (dst);                                                                        \
    LS3D_B_ONE(0, breg0, LS3D_B_ST) LS3D_B_ONE(1, breg1, LS3D_B_ST)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_ST) LS3D_B_ONE(3, breg3, LS3D_B_ST)                             \
  } while (0)
      int idx_cur = LS3D_LOAD_IDX(k_cur);
      int idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
      float4 a_cur[SPL / 4], a_nxt[SPL / 4];
      float4 breg0, breg1, breg2, breg3;
      static_assert(BPT <= 4, "weight chunk too large for the staging registers");
      int c0 = 0, buf = 0;
      LS3D_LOAD_A(a_cur, idx_cur, 0);
      LS3D_LOAD_B(k_cur, 0);
      LS3D_STORE_B(Bs[0]);
      __syncthreads();
      for (;;) {
        int nk = k_cur, nc0 = c0 + KC, nidx = idx_cur;
        bool has_next = true;
        if (nc0 >= cin) {
          nc0 = 0; nk = k_nxt; nidx = idx_nxt;
          has_next = nk >= 0;
        }
        if (has_next) {
          LS3D_LOAD_A(a_nxt, nidx, nc0);
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
        for (int q = 0; q < SPL / 4; ++q) a_cur[q] = a_nxt[q];
        if (nk != k_cur) {
          k_cur = nk; idx_cur = idx_nxt;
          rem &= rem - 1;
          k_nxt = rem ? __ffsll((long long)rem) - 1 : -1;
          idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
        }
        c0 = nc0;
      }
    }
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
__global__ __launch_bounds__(256, 2) void k_gather_gemm_bf16x3(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl,
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
      int k_cur = __ffsll((long long)rem) - 1;
      rem &= rem - 1;
      int k_nxt = rem ? __ffsll((long long)rem) - 1 : -1;
#define LS3D_LOAD_IDX(k) ((row >= 0) ? (SPARSE ? tbl[(size_t)row * kvol + (k)] : row) : -1)
#define LS3D_LOAD_A(dst, idx, c0_)                                                              \
  do {                                                                                          \
    if ((idx) >= 0) {                                                                           \
      const float4 *p_ = (const float4 *)(in + (size_t)(idx)*in_ld + (c0_) + kk * 16);          \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) dst[q] = p_[q];                             \
    } else {                                                                                    \
      _Pragma("unroll") for (int q = 0; q < 4; ++q) dst[q] = make_float4(0.f, 0.f, 0.f, 0.f);   \
    }                                                                                           \
  } while (0)
#define LS3D_B_ONE(j, reg, OP)                                                                  \
  if constexpr (BPT > (j)) {                                                                    \
    const int i_ = tid + (j)*256;                                                               \
    if (BV % 256 == 0 || i_ < BV) { OP(reg, i_); }                                              \
  }
#define LS3D_B_LD(reg, i_) reg = *(const float4 *)(wk_ + (size_t)(i_)*4)
#define LS3D_B_ST(reg, i_) *(float4 *)(dst_ + (i_)*4) = reg
#define LS3D_LOAD_B(k, c0_)                                                                     \
  do {                                                                                          \
    const float *wk_ = wbase + ((size_t)(k) * nslab * (cin / KC) + (c0_) / KC) * CHF;           \
    LS3D_B_ONE(0, breg0, LS3D_B_LD) LS3D_B_ONE(1, breg1, LS3D_B_LD)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_LD) LS3D_B_ONE(3, breg3, LS3D_B_LD)                             \
    LS3D_B_ONE(4, breg4, LS3D_B_LD) LS3D_B_ONE(5, breg5, LS3D_B_LD)                             \
  } while (0)
#define LS3D_STORE_B(dst)                                                                       \
  do {                                                                                          \
    float *dst_ = (dst);                                                                        \
    LS3D_B_ONE(0, breg0, LS3D_B_ST) LS3D_B_ONE(1, breg1, LS3D_B_ST)                             \
    LS3D_B_ONE(2, breg2, LS3D_B_ST) LS3D_B_ONE(3, breg3, LS3D_B_ST)                             \
    LS3D_B_ONE(4, breg4, LS3D_B_ST) LS3D_B_ONE(5, breg5, LS3D_B_ST)                             \
  } while (0)
      int idx_cur = LS3D_LOAD_IDX(k_cur);
      int idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
      float4 a_cur[4], a_nxt[4];
      float4 breg0, breg1, breg2, breg3, breg4, breg5;
      static_assert(BPT <= 6, "weight chunk too large for the staging registers");
      int c0 = 0, buf = 0;
      LS3D_LOAD_A(a_cur, idx_cur, 0);
      LS3D_LOAD_B(k_cur, 0);
      LS3D_STORE_B(Bs[0]);
      __syncthreads();
      for (;;) {
        int nk = k_cur, nc0 = c0 + KC, nidx = idx_cur;
        bool has_next = true;
        if (nc0 >= cin) {
          nc0 = 0; nk = k_nxt; nidx = idx_nxt;
          has_next = nk >= 0;
        }
        if (has_next) {
          LS3D_LOAD_A(a_nxt, nidx, nc0);
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
        for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
        if (nk != k_cur) {
          k_cur = nk; idx_cur = idx_nxt;
          rem &= rem - 1;
          k_nxt = rem ? __ffsll((long long)rem) - 1 : -1;
          idx_nxt = k_nxt >= 0 ? LS3D_LOAD_IDX(k_nxt) : -1;
        }
        c0 = nc0;
      }
    }
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

// ------------------------------------------------------------------------------------------------------------
// Pipelined sparse gather-GEMM (k_gg_pipe): the same tile / fragment / packed-weight conventions as the two kernels
// above, but the operands of step i + S - 1 are already on their way while step i runs on the matrix pipe:
//   * a step = (active kernel offset k, 32-channel chunk c0): 128 gathered rows x 128 B (A) + the KC x SLAB weight chunk (B);
//   * both go global -> LDS by LDS-DMA (ls3d_glds16: no staging VGPRs, asynchronous), into a ring of S stages; the gather
//     is 8 lanes per row (one 128-byte line per row and step), its 16-byte pieces XOR-swizzled on the SOURCE side with
//     (row >> 1) & 7 so that the ds_read_b128 of the MFMA fragments (lane = row, 4 consecutive pieces) is conflict-free;
//   * per step and wave: s_waitcnt vmcnt(#DMAs issued after this step's) -> raw s_barrier -> issue the DMAs of step
//     i + S - 1 into the stage step i - 1 just released -> LDS reads + MFMAs of step i.  A __syncthreads() in this loop
//     would drain the DMA queue (vmcnt(0)) and reduce the ring to depth 1;
//   * the neighbour indices of the tile's 128 rows are staged once in LDS (s_idx[k][row]) so that no compiler-issued
//     global load (and its conservative vmcnt(0)) sits in the pipelined loop; absent neighbours fetch row 0 and are
//     zeroed in registers;
//   * workgroup = 4 * WCOL waves: wave (wr, wc) owns rows [32 wr, 32 wr + 32) and the NTW 32-column blocks of column
//     group wc; with WCOL = 2 the 8 waves (2 per SIMD) share one copy of the gathered rows in LDS, i.e. a 128-column
//     layer gathers its input once instead of once per column slab.
// Why: rocprofv3 counters of the register-prefetch kernels (profiles/round1_pmc_sq.md): L2 hit rate of the gathers 50 %,
// waves parked on s_waitcnt/barrier 42 % (f32) / 59 % (split-bf16) of their cycles, MFMA busy 51 % / 18 % — one chunk
// of prefetch (0.3-0.45 us of MFMA work) does not cover an L2 miss.
template <int NTW, int WCOL, bool BF16, int S>
__global__ __launch_bounds__(256 * WCOL, 1) void k_gg_pipe(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl,
                                                           const int32_t *__restrict__ order, int kvol, const float *__restrict__ w,
                                                           int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e,
                                                           float *__restrict__ out, int out_ld, int xcd_map, int *tile_counter) {
  constexpr int KC = 32, TR = 128, NW = 4 * WCOL, NTH = 64 * NW;
  constexpr int WSLAB = NTW * 32, SLAB = WSLAB * WCOL;
  constexpr int A_BYTES = TR * KC * 4;                 // 16 KB of gathered rows per step
  constexpr int B_BYTES = KC * SLAB * 4;               // weight chunk per step (f32, or bf16 heads + tails: same bytes)
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_DMA = (A_BYTES / 1024) / NW;         // LDS-DMA instructions per wave and step (1 KB each)
  constexpr int B_DMA = (B_BYTES / 1024) / NW;
  constexpr int LPS = A_DMA + B_DMA;
  constexpr int PIECE_BLKS = KC * WSLAB * 4 / 1024;    // 1 KB blocks in one column group's weight piece
  static_assert(S >= 2 && S <= 5 && (S - 2) * LPS <= 63, "ring depth");
  static_assert(S * STAGE >= TR * SLAB * 4, "the epilogue transposes the whole tile through the ring");
  struct alignas(NTW == 3 ? 4 : 4 * NTW) BVec { float v[NTW]; };
  // ALL of this kernel's LDS is the dynamic allocation, so that the ring starts at LDS offset 0: behind static __shared__
  // variables it started at byte 1544 and every ds_read_b128 / 16-byte DMA was 8-byte misaligned (SQ_LDS_UNALIGNED_STALL on
  // 80 % of the LDS cycles, 2x slower than the register-prefetch kernels).
  HIP_DYNAMIC_SHARED(char, smem)                       // [S][A | B], s_rows[TR], s_stat[2 TR], s_kmask, s_idx[kvol][TR]
  int *s_rows = (int *)(smem + S * STAGE);
  float *s_stat = (float *)(s_rows + TR);
  unsigned long long *s_kmask_p = (unsigned long long *)(s_stat + 2 * TR);
  int *s_idx = (int *)(s_kmask_p + 2);
#define s_kmask (*s_kmask_p)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave & 3, wc = wave >> 2;
  const int col = lane & 31, kk = lane >> 5;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ntiles = (N + TR - 1) / TR;
  const int nwslab = w_ld / WSLAB;                     // packed: [kvol][wslab][cin][32][NTW] / [kvol][wslab][cin/32][chunk]
  const int nslab = w_ld / SLAB;
  const int nchunk = cin / KC;
  const int tiles_per_xcd = (ntiles + 7) / 8;          // work item -> (tile, slab): see k_gather_gemm
  // Persistent workgroups take work items from a global counter: with ~1 resident workgroup per CU a static grid of 680
  // tiles on 256 CUs runs 3 waves of workgroups for 2.66 of work (measured: SIMDs without a wave 24 % of the time).
  // Items are handed out in order, i.e. (mask-sorted rows) densest tiles first.
  for (;;) {
    __syncthreads();                                   // the previous item's epilogue has read s_rows / the ring
    if (tid == 0) *(int *)s_kmask_p = atomicAdd(tile_counter, 1);
    __syncthreads();
    const int b = *(volatile int *)s_kmask_p;
    if (b >= tiles_per_xcd * 8 * nslab) break;
    __syncthreads();                                   // everyone has read the ticket before s_kmask is reused
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
    const float *wbase = w + (size_t)slab * WCOL * cin * WSLAB;
    // ---- tile prologue: output rows, their neighbour indices (LDS), per-lane / per-wave / per-tile offset masks
    if (tid == 0) s_kmask = 0ull;
    if (tid < TR) {
      const int r = tile * TR + tid;
      s_rows[tid] = r < N ? (order ? order[r] : r) : -1;
    }
    __syncthreads();
    for (int i = tid; i < TR * kvol; i += NTH) {
      const int r = i / kvol, k = i - r * kvol;
      const int row = s_rows[r];
      s_idx[k * TR + r] = row >= 0 ? tbl[(size_t)row * kvol + k] : -1;
    }
    __syncthreads();
    unsigned vbits = 0u;                               // offsets at which THIS lane's row has a neighbour
    unsigned long long wmask = 0ull;                   // ... at which any row of this wave has one
    for (int k = 0; k < kvol; ++k) {
      const int idx = s_idx[k * TR + wr * 32 + col];
      if (idx >= 0) vbits |= 1u << k;
      if (__any(idx >= 0)) wmask |= 1ull << k;
    }
    if (lane == 0 && wmask && wc == 0) atomicOr(&s_kmask, wmask);
    __syncthreads();
    f32x16 acc[NTW];
#pragma unroll
    for (int n = 0; n < NTW; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    unsigned long long remI = s_kmask, remC = remI;
    const int nsteps = __popcll(remI) * nchunk;
    int kI = remI ? __ffsll((long long)remI) - 1 : 0, cI = 0;   // issue cursor
    int kC = kI, cC = 0;                                         // compute cursor
// ---- the DMAs of the step under the issue cursor, into ring stage st_
#define LS3D_PIPE_ISSUE(st_)                                                                          \
  do {                                                                                                \
    char *stg_ = smem + (st_)*STAGE;                                                                  \
    _Pragma("unroll") for (int j_ = 0; j_ < A_DMA; ++j_) {                                            \
      const int blk_ = wave * A_DMA + j_;              /* 1 KB = 8 tile rows x 128 B */               \
      const int r_ = blk_ * 8 + (lane >> 3);                                                          \
      const int idx_ = s_idx[kI * TR + r_];                                                           \
      const int pc_ = (lane & 7) ^ ((r_ >> 1) & 7);                                                   \
      ls3d_glds16(in + (size_t)(idx_ >= 0 ? idx_ : 0) * in_ld + cI + pc_ * 4, stg_ + blk_ * 1024);    \
    }                                                                                                 \
    const float *wk_ = wbase + ((size_t)kI * cin * nwslab + cI) * WSLAB;                              \
    _Pragma("unroll") for (int q_ = 0; q_ < B_DMA; ++q_) {                                            \
      const int blk_ = wave * B_DMA + q_;                                                             \
      const int p_ = blk_ / PIECE_BLKS, o_ = blk_ % PIECE_BLKS;                                       \
      ls3d_glds16(wk_ + (size_t)p_ * cin * WSLAB + o_ * 256 + lane * 4, stg_ + A_BYTES + blk_ * 1024); \
    }                                                                                                 \
    cI += KC;                                                                                         \
    if (cI >= cin) {                                                                                  \
      cI = 0; remI &= remI - 1;                                                                       \
      kI = remI ? __ffsll((long long)remI) - 1 : 0;                                                   \
    }                                                                                                 \
  } while (0)
#pragma unroll
    for (int s0 = 0; s0 < S - 2; ++s0)                // steps 0 .. S-3; iteration i of the loop below issues step i + S - 1
      if (s0 < nsteps) LS3D_PIPE_ISSUE(s0);
    // ---- MFMA fragments of one step, register resident: the 16 floats of this lane's row (channels [16 kk, 16 kk + 16) of
    //      the chunk) and the weight values of the wave's NTW column blocks.  They are read from LDS one step AHEAD, while
    //      the MFMAs of the current step run: the 8 waves of a workgroup leave the barrier together, so without this every
    //      step opens with a 96 KB LDS read burst during which all four matrix pipes idle.
    struct Frag {
      float4 a0, a1, a2, a3;
      uint4 bw[BF16 ? NTW : 1][4];                     // split-bf16: [n][t*2+h] heads / tails of k-step t
      BVec bf[BF16 ? 1 : 16];                          // f32: one NTW-vector per k-step
    };
#define LS3D_PIPE_LOAD(fr_, st_)                                                                      \
  do {                                                                                                \
    const char *stg_ = smem + (st_)*STAGE;                                                            \
    const int r_ = wr * 32 + col, sw_ = (r_ >> 1) & 7;                                                \
    const float4 *ap_ = (const float4 *)(stg_ + r_ * 128);                                            \
    fr_.a0 = ap_[(kk * 4 + 0) ^ sw_]; fr_.a1 = ap_[(kk * 4 + 1) ^ sw_];                               \
    fr_.a2 = ap_[(kk * 4 + 2) ^ sw_]; fr_.a3 = ap_[(kk * 4 + 3) ^ sw_];                               \
    if constexpr (BF16) {                                                                             \
      const uint4 *bs_ = (const uint4 *)(stg_ + A_BYTES + wc * (KC * WSLAB * 4)) + kk * 32 + col;     \
      _Pragma("unroll") for (int n_ = 0; n_ < NTW; ++n_)                                              \
          _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) fr_.bw[n_][u_] = bs_[(n_ * 4 + u_) * 64];  \
    } else {                                                                                          \
      const float *bs_ = (const float *)(stg_ + A_BYTES) + wc * (KC * WSLAB) + (kk * 16 * 32 + col) * NTW; \
      _Pragma("unroll") for (int u_ = 0; u_ < 16; ++u_) fr_.bf[u_] = *(const BVec *)(bs_ + u_ * 32 * NTW); \
    }                                                                                                 \
  } while (0)
    Frag cur, nxt;
    int kN = kC, cN = 0;                               // cursor of step i + 1 (the compute cursor kC, cC follows it)
    unsigned long long remN = remC;
#define LS3D_PIPE_ADVANCE(k_, c_, rem_)                                                               \
  do {                                                                                                \
    c_ += KC;                                                                                         \
    if (c_ >= cin) {                                                                                  \
      c_ = 0; rem_ &= rem_ - 1;                                                                       \
      k_ = rem_ ? __ffsll((long long)rem_) - 1 : 0;                                                   \
    }                                                                                                 \
  } while (0)
    int stN = 0, stI = S - 2;                          // ring stage of step i + 1 / of step i + S - 1
    // iteration i: MFMAs of step i on `cur` while the fragments of step i + 1 travel LDS -> `nxt` and the DMAs of step
    // i + S - 1 are issued.  Iteration -1 only loads the fragments of step 0 (one code path for every fragment load keeps
    // hipcc's lgkmcnt bookkeeping at the loop head trivial: `cur` is always the product of the copy at the loop tail).
    for (int i = -1; i < nsteps; ++i) {
      // step i + 1 must have landed; steps issued after it so far: min(S - 3, nsteps - 2 - i)
      if (i + 1 < nsteps) {
        if (S >= 5 && nsteps - 2 - i >= 2) LS3D_WAIT_VMCNT(2 * LPS);
        else if (S >= 4 && nsteps - 2 - i >= 1) LS3D_WAIT_VMCNT(1 * LPS);
        else LS3D_WAIT_VMCNT(0);
      }
      LS3D_RAW_BARRIER();                              // step i+1 landed for every wave; the stage of step i-1 is free
      if (i + S - 1 < nsteps) LS3D_PIPE_ISSUE(stI);
      // unconditional (a wave that skips step i + 1, or the last iteration, reads bytes it never uses): a conditional load
      // turns `nxt` into a phi and hipcc then copies every fragment register right behind its ds_read, i.e. waits for it
      LS3D_PIPE_LOAD(nxt, stN);
      LS3D_SCHED_FENCE();                              // the reads stay in flight under the MFMAs; `cur = nxt` (and its wait) after them
      const bool act = i >= 0 && ((wmask >> kC) & 1ull);
      if (act) {
        const bool keep = (vbits >> kC) & 1u;
        float4 z0 = cur.a0, z1 = cur.a1, z2 = cur.a2, z3 = cur.a3;
        if (!keep) z0 = z1 = z2 = z3 = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (BF16) {
          uint4 ah0, al0, ah1, al1;
          ls3d_split8(z0, z1, ah0, al0);
          ls3d_split8(z2, z3, ah1, al1);
          const bf16x8 vah0 = __builtin_bit_cast(bf16x8, ah0), val0 = __builtin_bit_cast(bf16x8, al0);
          const bf16x8 vah1 = __builtin_bit_cast(bf16x8, ah1), val1 = __builtin_bit_cast(bf16x8, al1);
#pragma unroll
          for (int n = 0; n < NTW; ++n) {
            const bf16x8 bh0 = __builtin_bit_cast(bf16x8, cur.bw[n][0]), bl0 = __builtin_bit_cast(bf16x8, cur.bw[n][1]);
            const bf16x8 bh1 = __builtin_bit_cast(bf16x8, cur.bw[n][2]), bl1 = __builtin_bit_cast(bf16x8, cur.bw[n][3]);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val0, bh0, acc[n], 0, 0, 0);  // small terms first
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah0, bl0, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(val1, bh1, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah1, bl1, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah0, bh0, acc[n], 0, 0, 0);
            acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vah1, bh1, acc[n], 0, 0, 0);
          }
        } else {
          const float av[16] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w, z2.x, z2.y, z2.z, z2.w, z3.x, z3.y, z3.z, z3.w};
#pragma unroll
          for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int n = 0; n < NTW; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], cur.bf[u].v[n], acc[n], 0, 0, 0);
        }
      }
      LS3D_SCHED_FENCE();
      cur = nxt;
      kC = kN; cC = cN; remC = remN;
      if (i + 1 < nsteps) LS3D_PIPE_ADVANCE(kN, cN, remN);
      stN = stN + 1 == S ? 0 : stN + 1;
      stI = stI + 1 == S ? 0 : stI + 1;
    }
#undef LS3D_PIPE_LOAD
#undef LS3D_PIPE_ADVANCE
#undef LS3D_PIPE_ISSUE
    gg_epilogue<NTW, WCOL, TR, TR, NTH>(acc, (float *)smem, s_rows, s_stat, wr, wc, kk, col, n0, cout, e, out, out_ld);
  }
}

#undef s_kmask

static int g_pipe = 0;  // 1: k_gg_pipe where it applies.  Off by default: on MI355X it is 10-20 % slower than the register-prefetch
                        // kernels on every layer of the 120k-point frame (profiles/round1_experiments.md)
extern "C" void ls3d_set_gather_pipeline(int on) { g_pipe = on ? 1 : 0; }

// work-item counters of the persistent k_gg_pipe launches: a small device-resident ring, one slot per launch, zeroed on the
// launch's own stream right before the kernel
static int *g_pipe_counters = nullptr;
static unsigned g_pipe_launches = 0;
static int g_num_cus = 0;
constexpr int LS3D_PIPE_COUNTER_SLOTS = 1024;

template <int NTW, int WCOL, bool BF16, int S>
static int launch_pipe(long long work_items, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, const int32_t *order, int kvol,
                       const float *w, int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld) {
  constexpr int STAGE = 128 * 32 * 4 + 32 * NTW * 32 * WCOL * 4;
  constexpr int FIXED = 128 * 4 + 2 * 128 * 4 + 16;  // s_rows, s_stat, s_kmask
  const size_t lds = (size_t)S * STAGE + FIXED + (size_t)kvol * 128 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_gg_pipe<NTW, WCOL, BF16, S>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(S * STAGE + FIXED + 32 * 128 * 4)) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  if (!g_pipe_counters) {
    if (hipMalloc((void **)&g_pipe_counters, LS3D_PIPE_COUNTER_SLOTS * sizeof(int)) != hipSuccess) return LS3D_ERR_LAUNCH;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&g_num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || g_num_cus < 1)
      g_num_cus = 256;
  }
  int *counter = g_pipe_counters + (g_pipe_launches++ % LS3D_PIPE_COUNTER_SLOTS);
  if (hipMemsetAsync(counter, 0, sizeof(int), stream) != hipSuccess) return LS3D_ERR_LAUNCH;
  const int per_cu = (160 * 1024) / (int)(lds + 512) > 0 ? (160 * 1024) / (int)(lds + 512) : 1;  // resident workgroups per CU (LDS-limited)
  long long nblk = (long long)g_num_cus * per_cu;
  if (nblk > work_items) nblk = work_items;
  hipLaunchKernelGGL((k_gg_pipe<NTW, WCOL, BF16, S>), dim3((unsigned)nblk), dim3(256 * WCOL), lds, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld,
                     cout, n_rows, n_rows_dev, e, out, out_ld, g_xcd_map, counter);
  return LS3D_OK;
}

template <int NT, int PL>
static void launch_gg3(dim3 grid, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, const int32_t *order, int kvol,
                       const float *w, int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld) {
  if (tbl)
    hipLaunchKernelGGL((k_gather_gemm_bf16x3<NT, true, PL>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout, n_rows,
                       n_rows_dev, e, out, out_ld, g_xcd_map);
  else
    hipLaunchKernelGGL((k_gather_gemm_bf16x3<NT, false, PL>), grid, dim3(256), 0, stream, in, in_ld, tbl, order, kvol, w, cin, w_ld, cout, n_rows,
                       n_rows_dev, e, out, out_ld, g_xcd_map);
}

template <int KC, int NT, int WC>
static void launch_gg(dim3 grid, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, const int32_t *order, int kvol,
                      const float *w, int cin, int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld) {
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
                                const ls3d_epilogue_t *epi, float *out, int out_ld, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
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
  // ---- sparse convolutions with cin % 32 == 0: the LDS-DMA pipelined kernel (geometries (nt, wc) = (1,1), (1,2), (2,2);
  //      wc there = column groups of an 8-wave workgroup over a 128-row tile)
  if (g_pipe && tbl && (cin % 32) == 0 && kvol <= 32 && !e.ln_gamma && (precision == LS3D_PRECISION_F32 || precision == LS3D_PRECISION_BF16X3) &&
      ((nt == 1 && (wc == 1 || wc == 2)) || (nt == 2 && wc == 2))) {
    const int ntiles_p = (n_rows + 127) / 128;
    const long long nwg_p = (long long)((ntiles_p + 7) / 8) * 8 * slabs;
    const bool bf = precision == LS3D_PRECISION_BF16X3;
    int rc;
#define LS3D_PIPE(NTW, WCOL, S) (bf ? launch_pipe<NTW, WCOL, true, S>(nwg_p, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld) \
                                    : launch_pipe<NTW, WCOL, false, S>(nwg_p, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld))
    if (nt == 1 && wc == 1) rc = LS3D_PIPE(1, 1, 3);
    else if (nt == 1) rc = LS3D_PIPE(1, 2, 4);
    else rc = LS3D_PIPE(2, 2, 4);
#undef LS3D_PIPE
    if (rc != LS3D_OK) return rc;
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  const int tr = 32 * (4 / wc);
  const int ntiles = (n_rows + tr - 1) / tr;
  const long long nwg = (long long)((ntiles + 7) / 8) * 8 * slabs;  // one workgroup per (tile, slab), see the kernels
  dim3 grid((unsigned)(nwg < (1 << 20) ? nwg : (1 << 20)));
  if (precision == LS3D_PRECISION_BF16X3 || precision == LS3D_PRECISION_BF16X6) {
    if ((cin % 32) || wc != 1) return LS3D_ERR_ARG;
#define LS3D_GG3(NT_, PL_) launch_gg3<NT_, PL_>(grid, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld)
    if (precision == LS3D_PRECISION_BF16X3) {
      switch (nt) {
        case 1: LS3D_GG3(1, 2); break;
        case 2: LS3D_GG3(2, 2); break;
        case 3: LS3D_GG3(3, 2); break;
        default: LS3D_GG3(4, 2);
      }
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
#define LS3D_GG(KC, NT, WC) launch_gg<KC, NT, WC>(grid, stream, in, in_ld, tbl, row_order, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld)
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
