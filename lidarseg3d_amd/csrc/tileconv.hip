// tileconv.hip — tile-halo sparse convolution:  out[r] = epilogue( sum_k W[k]^T in[tbl[r][k]] )  on spatially compact tiles.
//
// Same operator as ls3d_gather_gemm (spconv v1.x SubMConv3d / SparseConv3d / SparseInverseConv3d semantics, SURVEY.md §2.3,
// call sites det3d/models/backbones/scn_unet.py:15-24,34-69), different data movement.  The gather-GEMM fetches every
// (output, offset) pair's input row from L2 — 17 fetches per row and column slab on the 128-channel levels of the
// 120k-point frame, at a 50 % L2 hit rate (profiles/round1_pmc_sq.md).  Here
//   * output rows are ordered along a space-filling key (ls3d_tile_keys: Morton order of 4x4 (y,x) columns) and cut into
//     tiles of 128 rows, so a tile is a compact patch of the LiDAR surface;
//   * ls3d_tile_build writes, per tile, the UNIQUE input rows it touches (its halo, ~2.4x128 rows instead of 17.6x128
//     gathers) and a local table tloc[k][slot] -> halo position; rows of a tile are sorted by neighbour mask so that the
//     four waves still skip most empty offsets;
//   * the kernel stages one 16-channel chunk of the halo in LDS ONCE per tile — split there into three bf16 planes
//     (exact 8+8+8-bit split of the f32 mantissa) — and all kernel offsets run out of LDS: A fragments by local index,
//     weight chunks (pre-split at pack time) double-buffered through LDS, v_mfma_f32_32x32x16_bf16 with f32 accumulation;
//   * products: 8 of the 9 plane products (everything but tail x tail, 2^-32 relative) = exact-f32-grade arithmetic at
//     half the matrix time of v_mfma_f32_32x32x2_f32, or the 6 products of weight >= 2^-16 (the BF16X6 mode of ls3d_gather_gemm);
//   * a halo larger than the LDS window (TC_HCAP rows) is processed in several passes of the same loop, absent rows read a
//     zero row: any table works, only speed depends on locality.
// Summation order of one output row: input-channel chunks outer, kernel offsets inner — fixed by the plan, so results are
// bit-reproducible run to run.
#include "gemm_common.h"

constexpr int TC_TR = 128;     // output rows per tile (4 waves x 32)
constexpr int TC_HCAP = 448;   // halo rows resident in LDS per pass
constexpr int TC_KMAX = 32;    // kernel offsets per table (3x3x3 = 27)
constexpr int TC_META = 8;     // ints per tile: halo slots, tile offset mask, 4 wave offset masks, live rows, producer tiles (-1: more than TC_DEPCAP)
constexpr int TC_DEPCAP = 32;  // producer tiles listed per tile (ls3d_tile_conv_chain)
constexpr int TC_DEPHASH = 128;

// ---------------------------------------------------------------------------------------------------------------
// spatial sort keys
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t tc_spread(uint32_t v) {  // 16 bits -> every other bit
  v &= 0xFFFFu;
  v = (v | (v << 8)) & 0x00FF00FFu;
  v = (v | (v << 4)) & 0x0F0F0F0Fu;
  v = (v | (v << 2)) & 0x33333333u;
  v = (v | (v << 1)) & 0x55555555u;
  return v;
}

__global__ __launch_bounds__(256) void k_tile_keys(const int32_t *__restrict__ coords, int n, const int32_t *n_dev, int shift, int mbits,
                                                   uint32_t *__restrict__ keys) {
  const int N = ls3d_count(n, n_dev);
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    uint32_t key = 0x7FFFFFFFu;  // rows beyond the device count sort last
    if (i < N) {
      const int b = coords[4 * i], y = coords[4 * i + 2], x = coords[4 * i + 3];
      key = ((uint32_t)b << mbits) | (tc_spread((uint32_t)(y >> shift)) << 1) | tc_spread((uint32_t)(x >> shift));
    }
    keys[i] = key;
  }
}

static int tc_key_layout(const int32_t shape_zyx[3], int batch, int &shift, int &mbits) {
  int bbits = 0;
  while ((1 << bbits) < batch) ++bbits;
  const int ext = shape_zyx[1] > shape_zyx[2] ? shape_zyx[1] : shape_zyx[2];
  int cbits;
  shift = 2;  // 4x4 (y,x) columns; coarser only if the key would not fit 31 bits
  for (;; ++shift) {
    cbits = 0;
    while ((1 << cbits) < ((ext + (1 << shift) - 1) >> shift)) ++cbits;
    if (2 * cbits + bbits <= 31) break;
  }
  mbits = 2 * cbits;
  return cbits > 16 ? -1 : mbits + bbits;  // number of key bits
}

extern "C" int ls3d_tile_keys(const int32_t *coords, int n, const int32_t *n_dev, const int32_t shape_zyx[3], int batch, uint32_t *keys,
                              ls3d_stream_t stream) {
  if (n == 0 && shape_zyx && batch >= 1) return LS3D_OK;  // nothing to do: empty tensors have no storage
  if (!coords || !keys || !shape_zyx || n < 0 || batch < 1) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  int shift, mbits;
  if (tc_key_layout(shape_zyx, batch, shift, mbits) < 0) return LS3D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(k_tile_keys, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, coords, n, n_dev, shift, mbits, keys);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// tile plan
// ---------------------------------------------------------------------------------------------------------------
// Coloured halo layout (ls3d_tile_plan flag bit 1, 3x3x3 SubM tables).  The convolution's A fragments are gathered from the LDS halo with
// ds_read_b128: four groups of 16 lanes ({0-3,12-15,20-27}, {4-11,16-19,28-31} of each half-wave), one LDS cycle per group when the 16 halo
// rows sit in 16 different 16-byte columns of the 256-byte bank row, i.e. (with the kernel's half swizzle) when their LDS slots differ mod 16.
// Neighbours of spatially ordered rows are NOT consecutive halo rows (9.1 cycles per read measured instead of 4).  With a LINEAR colour
// f(z, y, x) = (a z + b y + c x) mod 16 the neighbour at offset d of a row of colour q has colour q + f(d): if the 16 rows of a lane group
// have 16 different colours, so have their neighbours at every offset.  So: a halo row of colour q gets a slot = q (mod 16), the tile's 128 rows
// are sorted by colour and dealt round-robin to the 8 lane groups (a colour with <= 8 rows never meets itself), an ABSENT neighbour reads the
// zero row of the colour it would have had (16 zero rows), and (a, b, c) is the best of 16 candidates per tile (fewest rows beyond 8 per
// colour; the candidates: greedy choice over the tiles of synthetic nuScenes / Waymo frames, levels 2 - 4).  Slots are holes where a colour
// has fewer rows than the fullest one; a hole stages a duplicate of the tile's first halo row.  Placement only: any assignment is correct.
constexpr unsigned long long TC_COL_Z = 0x2124281128828188ull, TC_COL_Y = 0x4C834294712C1741ull, TC_COL_X = 0x3B187349CE51AC1Cull;  // 16 x 4 bits each
__device__ __forceinline__ int tc_color(int cand, int z, int y, int x) {
  const int a = (int)((TC_COL_Z >> (4 * cand)) & 15ull), b = (int)((TC_COL_Y >> (4 * cand)) & 15ull), c = (int)((TC_COL_X >> (4 * cand)) & 15ull);
  return (a * z + b * y + c * x) & 15;
}

struct TilePlan {
  int ntiles, kvol, hs;
  int32_t *trow;    // [T][128]      output row of each tile slot (-1 = none), slots sorted by neighbour mask (densest first) or dealt by colour
  int32_t *tmeta;   // [T][TC_META]
  int32_t *thalo;   // [T][hs]       input rows of the tile's halo slots: the unique rows ascending, or (coloured) by colour with duplicates in the holes
  uint16_t *tloc;   // [T][kvol][128] position of tbl[row][k] in the tile's halo; >= 0xFFF0 = no neighbour (low 4 bits: the zero row to read)
  int32_t *torder;  // [T + 1]       dispatch order of the tiles: most expensive first (k_tile_order); torder[T] = number of LIVE tiles (tiles
                    //               with at least one row below the device row count)
  int32_t *rowtile; // [T * 128]     tile that holds output row r (inverse of the spatial order, / 128)
  int32_t *tdep;    // [T][TC_DEPCAP] the tiles whose OUTPUT rows the tile's halo reads (its producers in a chained launch; meaningful when
                    //               the table's input sites are its output sites: SubM); their number in tmeta[7]
};

static inline size_t tc_align(size_t v) { return (v + 255) & ~(size_t)255; }

static TilePlan tc_plan(void *buf, int n_rows, int kvol) {
  TilePlan p;
  p.ntiles = (n_rows + TC_TR - 1) / TC_TR;
  p.kvol = kvol;
  p.hs = kvol * TC_TR;
  char *b = (char *)buf;
  p.trow = (int32_t *)b; b += tc_align((size_t)p.ntiles * TC_TR * 4);
  p.tmeta = (int32_t *)b; b += tc_align((size_t)p.ntiles * TC_META * 4);
  p.thalo = (int32_t *)b; b += tc_align((size_t)p.ntiles * p.hs * 4);
  p.tloc = (uint16_t *)b; b += tc_align((size_t)p.ntiles * kvol * TC_TR * 2);
  p.torder = (int32_t *)b; b += tc_align((size_t)(p.ntiles + 1) * 4);
  p.rowtile = (int32_t *)b; b += tc_align((size_t)p.ntiles * TC_TR * 4);
  p.tdep = (int32_t *)b;
  return p;
}

extern "C" size_t ls3d_tile_plan_bytes(int n_rows, int kvol) {
  if (n_rows < 0 || kvol < 1) return 0;
  const size_t t = (size_t)(n_rows + TC_TR - 1) / TC_TR;
  return tc_align(t * TC_TR * 4) + tc_align(t * TC_META * 4) + tc_align(t * kvol * TC_TR * 4) + tc_align(t * kvol * TC_TR * 2) + tc_align((t + 1) * 4) +
         tc_align(t * TC_TR * 4) + tc_align(t * TC_DEPCAP * 4);
}

// rowtile[r] = tile of output row r: position of r in the spatial order / 128 (the order is a permutation of all n rows, spare rows last)
__global__ __launch_bounds__(256) void k_tile_rowtile(const int32_t *__restrict__ sorder, int n, const int32_t *n_dev, int32_t *__restrict__ rowtile) {
  const int N = ls3d_count(n, n_dev);  // the order beyond the device count is unspecified (ls3d_radix_sort_pairs)
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
    const int r = sorder[i];
    if ((unsigned)r < (unsigned)n) rowtile[r] = i / TC_TR;
  }
}

// one workgroup per tile.  The tile's distinct input rows: the <= kvol*128 table entries go through an LDS hash set (insertion
// order is arbitrary, the SET is not), the occupied slots are compacted and only that list (a few hundred rows) is sorted
// (bitonic), so the halo list, and with it every local index, is the same on every run.
// COLOR: the build with the coloured layout compiled in (it needs 250 registers: two workgroups per CU instead of three - 88 instead of 70 us per
// call - so the plain build stays a kernel of its own)
template <bool COLOR>
__global__ __launch_bounds__(256) void k_tile_build(const int32_t *__restrict__ tbl, int n, const int32_t *n_dev, int kvol,
                                                    const int32_t *__restrict__ sorder, const int32_t *__restrict__ coords, TilePlan p) {
  constexpr int NC = TC_KMAX * TC_TR;  // 4096 candidate slots
  constexpr int HS = 2 * NC;           // hash slots
  __shared__ int s_row[TC_TR], s_srow[TC_TR], s_rank[TC_TR];
  __shared__ unsigned s_mask[TC_TR], s_smask[TC_TR];
  __shared__ int s_hash[HS];
  __shared__ int s_uniq[NC];
  __shared__ int s_scan[2][256];
  __shared__ int s_col[TC_TR];
  // the coloured layout's scratch lives in arrays that are dead while it is needed (no more than the one array the plain build can afford beside its three workgroups per CU): the candidates' colour histograms in s_uniq (empty until the compaction), the rest in the second scan row
  int *const s_ch = s_uniq, *const s_exc = &s_scan[1][0], *const s_cw = &s_scan[1][32], *const s_run = &s_scan[1][96], *const s_cnt2 = &s_scan[1][112];
  const int tid = threadIdx.x;
  const int N = ls3d_count(n, n_dev);
  const bool colored = COLOR && coords != nullptr && kvol == 27;  // the coloured layout (see tc_color)
  for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
    __syncthreads();
    if (tid < TC_TR) {
      const int r = tile * TC_TR + tid;
      s_row[tid] = r < N ? sorder[r] : -1;
      s_mask[tid] = 0u;
    }
    for (int i = tid; i < HS; i += 256) s_hash[i] = -1;
    s_ch[tid] = 0;
    __syncthreads();
    // neighbour masks + hash-set insertion of every neighbour: two threads per slot.  The thread's <= 16 table entries are fetched with
    // unconditional loads first (clamped addresses, masked afterwards: one memory latency instead of one per entry - under a condition
    // hipcc waits for each load before the next) and stay in registers for the local-index pass at the end (no second read of the table).
    constexpr int KPT = TC_KMAX / 2;
    const int slot = tid & (TC_TR - 1), part = tid >> 7;
    int vals[KPT];
    {
      const int row = s_row[slot];
      const int32_t *trow = tbl + (size_t)(row >= 0 ? row : 0) * kvol;
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int k = part + 2 * i;
        vals[i] = trow[k < kvol ? k : kvol - 1];
      }
      unsigned m = 0u;
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int k = part + 2 * i;
        if (row < 0 || k >= kvol) vals[i] = -1;
        const int v = vals[i];
        if (v >= 0) {
          m |= 1u << k;
          unsigned h = ((unsigned)v * 2654435761u) >> 19;  // 13 bits
          for (;;) {
            const int prev = atomicCAS(&s_hash[h], -1, v);
            if (prev == -1 || prev == v) break;
            h = (h + 1) & (HS - 1);
          }
        }
      }
      if (m) atomicOr(&s_mask[slot], m);
    }
    int best = 0;  // the tile's colour candidate
    if (colored) {  // colour histograms of the tile's rows under the 16 candidates (two threads per row, 8 candidates each)
      const int row = s_row[slot];
      int z = 0, y = 0, x = 0;
      if (row >= 0) {
        z = coords[4 * (size_t)row + 1]; y = coords[4 * (size_t)row + 2]; x = coords[4 * (size_t)row + 3];
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) atomicAdd(&s_ch[(part * 8 + c8) * 16 + tc_color(part * 8 + c8, z, y, x)], 1);
      }
      __syncthreads();
      if (tid < 16) {
        int e = 0;
        for (int q = 0; q < 16; ++q) e += s_ch[tid * 16 + q] > 8 ? s_ch[tid * 16 + q] - 8 : 0;  // rows that must share a lane group with their colour
        s_exc[tid] = e;
      }
      __syncthreads();
      int be = s_exc[0];
      for (int q = 1; q < 16; ++q)
        if (s_exc[q] < be) { be = s_exc[q]; best = q; }
      if (part == 0) s_col[slot] = row >= 0 ? tc_color(best, z, y, x) : 16;
    }
    __syncthreads();
    if (tid < TC_TR) {
      const unsigned mine = s_mask[tid];
      const bool live = s_row[tid] >= 0;
      int rank = 0;
      if (!colored) {  // slots sorted by mask, densest first (ties keep the spatial order); empty slots go last
        for (int j = 0; j < TC_TR; ++j) {
          const unsigned mj = s_mask[j];
          const bool lj = s_row[j] >= 0;
          rank += (lj && !live) || (lj == live && (mj > mine || (mj == mine && j < tid)));
        }
      } else {  // sorted by colour (ties: spatial order; empty slots last), then dealt round-robin to the 8 ds_read_b128 lane groups
        const int key = s_col[tid] * TC_TR + tid;
        int r = 0;
        for (int j = 0; j < TC_TR; ++j) r += (s_col[j] * TC_TR + j) < key;
        const int g = r & 7, j = r >> 3;
        rank = (g >> 1) * 32 + ((g & 1) ? (j < 8 ? j + 4 : j < 12 ? j + 8 : j + 16) : (j < 4 ? j : j < 8 ? j + 8 : j + 12));
      }
      s_srow[rank] = s_row[tid];
      s_smask[rank] = mine;
      s_rank[tid] = rank;
    }
    // compact the occupied hash slots -> s_uniq (unsorted), H
    constexpr int PER = HS / 256;
    int cnt = 0;
    for (int u = 0; u < PER; ++u) cnt += s_hash[tid * PER + u] >= 0;
    s_scan[0][tid] = cnt;
    __syncthreads();
    int src = 0;
    for (int d = 1; d < 256; d <<= 1) {  // inclusive scan of the per-thread counts
      s_scan[src ^ 1][tid] = s_scan[src][tid] + (tid >= d ? s_scan[src][tid - d] : 0);
      src ^= 1;
      __syncthreads();
    }
    const int H = s_scan[src][255];
    int pos = s_scan[src][tid] - cnt;
    for (int u = 0; u < PER; ++u) {
      const int v = s_hash[tid * PER + u];
      if (v >= 0) s_uniq[pos++] = v;
    }
    int np2 = 2;
    while (np2 < H) np2 <<= 1;
    __syncthreads();
    for (int i = H + tid; i < np2; i += 256) s_uniq[i] = 0x7FFFFFFF;
    __syncthreads();
    for (int kk = 2; kk <= np2; kk <<= 1)  // bitonic sort of the unique list, ascending
      for (int j = kk >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < np2 / 2; t += 256) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));  // lower index of the pair
          const int q = i | j;
          const bool up = (i & kk) == 0;
          const int a = s_uniq[i], b = s_uniq[q];
          if ((a > b) == up) { s_uniq[i] = b; s_uniq[q] = a; }
        }
        __syncthreads();
      }
    // Coloured tiles (one LDS pass): the halo entry of colour q that is the r-th of its colour (in ascending row order: ranks from wave
    // ballots, no atomics on the placement) gets slot q + 16 r; entries beyond the window (a colour with more than TC_HCAP / 16 rows) take
    // the free slots in order.  s_hash is dead since the compaction: s_slot = slot of entry i, s_inv = entry at slot s (-1: a hole).
    const bool tcol = colored && H <= TC_HCAP;
    int *const s_slot = s_hash, *const s_inv = s_hash + TC_HCAP;
    int nslot = H;
    if (tcol) {
      const int lane = tid & 63, wave = tid >> 6;
      for (int i = tid; i < TC_HCAP; i += 256) s_inv[i] = -1;
      if (tid < 16) s_run[tid] = 0;
      if (tid < 2) s_cnt2[tid] = 0;
      __syncthreads();
      for (int base = 0; base < H; base += 256) {
        const int i = base + tid;
        int c = 16;
        if (i < H) {
          const int v = s_uniq[i];
          c = v < n ? tc_color(best, coords[4 * (size_t)v + 1], coords[4 * (size_t)v + 2], coords[4 * (size_t)v + 3]) : 0;
        }
        int myrank = 0;
        for (int q = 0; q < 16; ++q) {
          const unsigned long long m = __ballot(c == q);
          if (c == q) myrank = __popcll(m & ((1ull << lane) - 1ull));
          if (lane == 0) s_cw[wave * 16 + q] = __popcll(m);
        }
        __syncthreads();
        if (i < H) {
          int r = s_run[c] + myrank;
          for (int w = 0; w < wave; ++w) r += s_cw[w * 16 + c];
          const int sl = c + 16 * r;
          if (sl < TC_HCAP) {
            s_slot[i] = sl;
            s_inv[sl] = i;
            atomicMax(&s_cnt2[0], sl + 1);
          } else {
            s_slot[i] = -1;
            atomicAdd(&s_cnt2[1], 1);
          }
        }
        __syncthreads();
        if (tid < 16) s_run[tid] += s_cw[tid] + s_cw[16 + tid] + s_cw[32 + tid] + s_cw[48 + tid];
        __syncthreads();
      }
      if (tid == 0 && s_cnt2[1] > 0) {
        int sl = 0;
        for (int i = 0; i < H; ++i)
          if (s_slot[i] < 0) {
            while (s_inv[sl] >= 0) ++sl;
            s_slot[i] = sl;
            s_inv[sl] = i;
            if (sl + 1 > s_cnt2[0]) s_cnt2[0] = sl + 1;
          }
      }
      __syncthreads();
      nslot = s_cnt2[0];
      for (int i = tid; i < nslot; i += 256) p.thalo[(size_t)tile * p.hs + i] = s_uniq[s_inv[i] >= 0 ? s_inv[i] : 0];
    } else {
      for (int i = tid; i < H; i += 256) p.thalo[(size_t)tile * p.hs + i] = s_uniq[i];
    }
    {  // local indices of the thread's own table entries (still in registers), written at the slot's position in the mask / colour order
      const int s2 = s_rank[slot];
#pragma unroll
      for (int i = 0; i < KPT; ++i) {
        const int k = part + 2 * i;
        if (k < kvol) {
          const int v = vals[i];
          // no neighbour: 0xFFF0 | the colour it would have had (the kernel reads the zero row of that colour; 0xFFFF where there are no colours)
          unsigned li = tcol ? (0xFFF0u | (unsigned)((s_col[slot] + tc_color(best, k / 9 - 1, (k / 3) % 3 - 1, k % 3 - 1)) & 15)) : 0xFFFFu;
          if (v >= 0) {
            int lo = 0, hi = H;  // lower bound of v in s_uniq (it is present)
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (s_uniq[mid] < v) lo = mid + 1; else hi = mid;
            }
            li = (unsigned)(tcol ? s_slot[lo] : lo);
          }
          p.tloc[((size_t)tile * kvol + k) * TC_TR + s2] = (uint16_t)li;
        }
      }
    }
    // Producer tiles (ls3d_tile_conv_chain: layer l + 1 of a chained launch starts a tile as soon as the tiles that own its halo rows have
    // finished layer l): the distinct rowtile[] of the halo rows through a small LDS hash set (s_hash is dead since the compaction); the
    // list's order is arbitrary, the set is not.  More than TC_DEPCAP producers: tmeta[7] = -1 ("wait for the whole layer").
    __syncthreads();
    if (tid < TC_DEPHASH) s_hash[tid] = -1;
    if (tid < 2) s_scan[0][tid] = 0;
    __syncthreads();
    {
      const int nrt = N;  // halo rows are valid input rows; only rows below the device count are output rows of this plan
      for (int i = tid; i < H; i += 256) {
        const int v = s_uniq[i];
        if (v >= nrt) { atomicAdd(&s_scan[0][0], TC_DEPHASH); continue; }  // not an output row of this plan (a table that is not SubM)
        const int t = p.rowtile[v];
        unsigned h = ((unsigned)t * 2654435761u) >> 25;  // 7 bits
        int probe = 0;
        for (; probe < TC_DEPHASH; ++probe) {
          const int prev = atomicCAS(&s_hash[h], -1, t);
          if (prev == -1) { atomicAdd(&s_scan[0][0], 1); break; }
          if (prev == t) break;
          h = (h + 1) & (TC_DEPHASH - 1);
        }
        if (probe == TC_DEPHASH) atomicAdd(&s_scan[0][0], TC_DEPHASH);
      }
    }
    __syncthreads();
    const int ndep = s_scan[0][0];
    if (tid < TC_DEPHASH && ndep <= TC_DEPCAP) {
      const int t = s_hash[tid];
      if (t >= 0) p.tdep[(size_t)tile * TC_DEPCAP + atomicAdd(&s_scan[0][1], 1)] = t;
    }
    if (tid < TC_TR) p.trow[(size_t)tile * TC_TR + tid] = s_srow[tid];
    if (tid < 8) {
      int v = 0;
      if (tid == 0) v = nslot;
      else if (tid == 1) { unsigned m = 0u; for (int s2 = 0; s2 < TC_TR; ++s2) m |= s_smask[s2]; v = (int)m; }
      else if (tid < 6) { unsigned m = 0u; for (int s2 = 0; s2 < 32; ++s2) m |= s_smask[(tid - 2) * 32 + s2]; v = (int)m; }
      else if (tid == 6) { int c2 = 0; for (int s2 = 0; s2 < TC_TR; ++s2) c2 += s_srow[s2] >= 0; v = c2; }
      else if (tid == 7) v = ndep <= TC_DEPCAP ? ndep : -1;
      p.tmeta[(size_t)tile * TC_META + tid] = v;
    }
  }
}

// Dispatch order of the tiles of a plan: most expensive first (cost = LDS passes x active kernel offsets, what the tile's MFMA
// time is proportional to), so that the last workgroups of a launch are the cheap ones; ties keep the spatial order.  One
// workgroup, stable counting sort on 256 cost levels: the rank of a tile among the equal keys of its round of 1024 comes from
// wave ballots over the key bits (as in k_rs_scatter) - no atomics on the placement, so the order (and with it the choice of the
// tiles that ls3d_tile_conv splits over the input channels) is the same on every run.
__global__ __launch_bounds__(1024) void k_tile_order(TilePlan p, int lpt) {
  __shared__ int s_base[256];
  __shared__ int s_cnt[16][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, T = p.ntiles;
  auto key_of = [&](int t) -> unsigned {
    if (!lpt) return 0u;
    const int32_t *m = p.tmeta + (size_t)t * TC_META;
    const int nseg = (m[0] + TC_HCAP - 1) / TC_HCAP;
    int cost = m[6] ? nseg * __popc((unsigned)m[1]) : 0;
    return 255u - (unsigned)(cost < 255 ? cost : 255);
  };
  __shared__ int s_live;
  if (tid < 256) s_base[tid] = 0;
  if (tid == 0) s_live = 0;
  __syncthreads();
  for (int t = tid; t < T; t += 1024) {
    atomicAdd(&s_base[key_of(t)], 1);  // counts are order-independent
    if (p.tmeta[(size_t)t * TC_META + 6]) atomicAdd(&s_live, 1);
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int d = 0; d < 256; ++d) { const int c = s_base[d]; s_base[d] = run; run += c; }
    p.torder[T] = s_live;
  }
  __syncthreads();
  for (int r0 = 0; r0 < T; r0 += 1024) {
    const int t = r0 + tid;
    const bool live = t < T;
    for (int w = 0; w < 16; ++w) if (tid < 256) s_cnt[w][tid] = 0;
    __syncthreads();
    const unsigned d = live ? key_of(t) : 0u;
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long m = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? m : ~m;
    }
    const int rank = __popcll(peers & ((1ull << lane) - 1ull));
    if (live && rank == 0) s_cnt[wave][d] = __popcll(peers);
    __syncthreads();
    if (live) {
      int off = s_base[d] + rank;
      for (int w = 0; w < wave; ++w) off += s_cnt[w][d];
      p.torder[off] = t;
    }
    __syncthreads();
    if (tid < 256) {
      int c = 0;
      for (int w = 0; w < 16; ++w) c += s_cnt[w][tid];
      s_base[tid] += c;
    }
    __syncthreads();
  }
}

extern "C" int ls3d_tile_build(const int32_t *tbl, int n_rows, const int32_t *n_rows_dev, int kvol, const int32_t *spatial_order, void *plan,
                               size_t plan_bytes, int flags, ls3d_stream_t stream) {
  if (n_rows == 0 && kvol >= 1 && kvol <= TC_KMAX) return LS3D_OK;
  if (!tbl || !spatial_order || !plan || n_rows < 0 || kvol < 1) return LS3D_ERR_ARG;
  if (kvol > TC_KMAX) return LS3D_ERR_UNSUPPORTED;
  if (plan_bytes < ls3d_tile_plan_bytes(n_rows, kvol)) return LS3D_ERR_WORKSPACE;
  if (((uintptr_t)plan & 15)) return LS3D_ERR_ARG;
  if (n_rows == 0) return LS3D_OK;
  TilePlan p = tc_plan(plan, n_rows, kvol);
  hipLaunchKernelGGL(k_tile_rowtile, ls3d_grid(n_rows), dim3(256), 0, (hipStream_t)stream, spatial_order, n_rows, n_rows_dev, p.rowtile);
  hipLaunchKernelGGL(k_tile_build<false>, dim3((unsigned)(p.ntiles < 8192 ? p.ntiles : 8192)), dim3(256), 0, (hipStream_t)stream, tbl, n_rows, n_rows_dev,
                     kvol, spatial_order, (const int32_t *)nullptr, p);
  hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, (flags & 1) ? 0 : 1);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t ls3d_radix_sort_workspace_bytes(int n);

extern "C" size_t ls3d_tile_plan_workspace_bytes(int n_rows) {
  return ((size_t)n_rows * 4 + 255) / 256 * 256 * 2 + ls3d_radix_sort_workspace_bytes(n_rows);  // keys, order, sort buffers
}

// keys -> stable radix sort -> plan, back to back on `stream`: what a caller does for every table of a frame
extern "C" int ls3d_tile_plan(const int32_t *tbl, const int32_t *coords, int n_rows, const int32_t *n_rows_dev, int kvol,
                              const int32_t shape_zyx[3], int batch, void *workspace, size_t workspace_bytes, void *plan, size_t plan_bytes,
                              int flags, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows == 0 && shape_zyx && kvol >= 1 && kvol <= TC_KMAX && batch >= 1) return LS3D_OK;
  if (!tbl || !coords || !shape_zyx || !workspace || !plan || n_rows < 0 || kvol < 1 || batch < 1) return LS3D_ERR_ARG;
  if (kvol > TC_KMAX) return LS3D_ERR_UNSUPPORTED;
  if (workspace_bytes < ls3d_tile_plan_workspace_bytes(n_rows) || plan_bytes < ls3d_tile_plan_bytes(n_rows, kvol)) return LS3D_ERR_WORKSPACE;
  if (((uintptr_t)plan & 15) || ((uintptr_t)workspace & 15)) return LS3D_ERR_ARG;
  if (n_rows == 0) return LS3D_OK;
  int shift, mbits;
  const int bits = tc_key_layout(shape_zyx, batch, shift, mbits);
  if (bits < 0) return LS3D_ERR_UNSUPPORTED;
  const size_t seg = ((size_t)n_rows * 4 + 255) / 256 * 256;
  uint32_t *keys = (uint32_t *)workspace;
  int32_t *order = (int32_t *)((char *)workspace + seg);
  void *sort_ws = (char *)workspace + 2 * seg;
  hipLaunchKernelGGL(k_tile_keys, ls3d_grid(n_rows), dim3(256), 0, stream, coords, n_rows, n_rows_dev, shift, mbits, keys);
  int rc = ls3d_radix_sort_pairs(keys, nullptr, n_rows, n_rows_dev, bits, nullptr, order, sort_ws, workspace_bytes - 2 * seg, stream);
  if (rc != LS3D_OK) return rc;
  TilePlan p = tc_plan(plan, n_rows, kvol);
  hipLaunchKernelGGL(k_tile_rowtile, ls3d_grid(n_rows), dim3(256), 0, stream, (const int32_t *)order, n_rows, n_rows_dev, p.rowtile);
  if ((flags & 2) && kvol == 27)
    hipLaunchKernelGGL(k_tile_build<true>, dim3((unsigned)(p.ntiles < 8192 ? p.ntiles : 8192)), dim3(256), 0, stream, tbl, n_rows, n_rows_dev, kvol,
                       (const int32_t *)order, coords, p);
  else
    hipLaunchKernelGGL(k_tile_build<false>, dim3((unsigned)(p.ntiles < 8192 ? p.ntiles : 8192)), dim3(256), 0, stream, tbl, n_rows, n_rows_dev, kvol,
                       (const int32_t *)order, (const int32_t *)nullptr, p);
  hipLaunchKernelGGL(k_tile_order, dim3(1), dim3(1024), 0, stream, p, (flags & 1) ? 0 : 1);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// weights: plain [kvol][cin_src][cout] f32 -> [kvol][cin_pad/16][nt][plane 3][kk 2][col 32] x (8 bf16), the exact 3-way split
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tile_pack(const float *__restrict__ src, int kvol, int cin_src, int cin_pad, int cout, int nt,
                                                   int trunc_split, int col0, int cout_all, int planes, uint4 *__restrict__ dst) {
  const int nchunk = cin_pad / 16;
  const long long total = (long long)kvol * nchunk * nt * planes * 64;
  for (long long t_ = (long long)blockIdx.x * blockDim.x + threadIdx.x; t_ < total; t_ += (long long)gridDim.x * blockDim.x) {
    long long r = t_;
    const int col = (int)(r % 32); r /= 32;
    const int kk = (int)(r % 2); r /= 2;
    const int pl = (int)(r % planes); r /= planes;
    const int n = (int)(r % nt); r /= nt;
    const int ch = (int)(r % nchunk); r /= nchunk;
    const int k = (int)r;
    const int oc = n * 32 + col;
    unsigned wds[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      unsigned half[2];
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        const int c = ch * 16 + kk * 8 + pr * 2 + e2;
        const float v = (c < cin_src && oc < cout) ? src[((size_t)k * cin_src + c) * cout_all + col0 + oc] : 0.0f;
        // round-to-nearest planes (v = h + m + l exactly, as for the activations): the remainders carry no sign bias, so the
        // products a mode leaves out (bf16x6: a_m w_l + a_l w_m, weight 2^-24) are zero-mean
        const unsigned hb = trunc_split ? (__float_as_uint(v) & 0xFFFF0000u) : (ls3d_bf16_rne(v) << 16);
        const float r1 = v - __uint_as_float(hb);
        const unsigned mb = trunc_split ? (__float_as_uint(r1) & 0xFFFF0000u) : (ls3d_bf16_rne(r1) << 16);
        half[e2] = pl == 0 ? (hb >> 16) : pl == 1 ? (mb >> 16) : ls3d_bf16_rne(r1 - __uint_as_float(mb));
      }
      wds[pr] = half[0] | (half[1] << 16);
    }
    uint4 o;
    o.x = wds[0]; o.y = wds[1]; o.z = wds[2]; o.w = wds[3];
    dst[t_] = o;
  }
}

// layers of more than 128 output columns (SCALING_RATIO > 2 of the reference's UNet) are slabs of <= 128 columns: packed slab after
// slab, convolved launch after launch on the same plan
// planes = 3: the exact split (products 6 / 8); planes = 1: the head plane only (products 1: plain bf16 operands, BASELINE configs[4]) - a third
// of the bytes, so that a 12 KB step of the kernel's weight stream carries three times the kernel offsets
static inline size_t tc_slab_packed_bytes(int kvol, int cin_pad, int cw, int planes = 3) {
  return (size_t)kvol * cin_pad * (cw <= 32 ? 32 : cw <= 64 ? 64 : 128) * 2 * planes;
}
static size_t tc_packed_bytes(int kvol, int cin_pad, int cout, int planes) {
  size_t b = 0;
  for (int col0 = 0; col0 < cout; col0 += 128) b += tc_slab_packed_bytes(kvol, cin_pad, cout - col0 < 128 ? cout - col0 : 128, planes);
  return b;
}
extern "C" size_t ls3d_tile_conv_packed_bytes(int kvol, int cin_pad, int cout) { return tc_packed_bytes(kvol, cin_pad, cout, 3); }
extern "C" size_t ls3d_tile_conv_packed_bytes_bf16(int kvol, int cin_pad, int cout) { return tc_packed_bytes(kvol, cin_pad, cout, 1); }

static int tc_pack(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, void *w_packed, int planes, ls3d_stream_t stream) {
  if (!w_plain || !w_packed || kvol < 1 || cin_src < 1 || cin_pad < cin_src || (cin_pad % 16) || cout < 1) return LS3D_ERR_ARG;
  char *dst = (char *)w_packed;
  for (int col0 = 0; col0 < cout; col0 += 128) {
    const int cw = cout - col0 < 128 ? cout - col0 : 128;
    const int nt = cw <= 32 ? 1 : cw <= 64 ? 2 : 4;  // column blocks of the kernel variant that will run (zero padded)
    const long long total = (long long)kvol * (cin_pad / 16) * nt * planes * 64;
    hipLaunchKernelGGL(k_tile_pack, ls3d_grid(total), dim3(256), 0, (hipStream_t)stream, w_plain, kvol, cin_src, cin_pad, cw, nt, 0, col0, cout, planes,
                       (uint4 *)dst);
    dst += tc_slab_packed_bytes(kvol, cin_pad, cw, planes);
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
extern "C" int ls3d_tile_conv_pack(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, void *w_packed, ls3d_stream_t stream) {
  return tc_pack(w_plain, kvol, cin_src, cin_pad, cout, w_packed, 3, stream);
}
extern "C" int ls3d_tile_conv_pack_bf16(const float *w_plain, int kvol, int cin_src, int cin_pad, int cout, void *w_packed, ls3d_stream_t stream) {
  return tc_pack(w_plain, kvol, cin_src, cin_pad, cout, w_packed, 1, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// the convolution
// ---------------------------------------------------------------------------------------------------------------
constexpr int TC_PLANE_BYTES = (TC_HCAP + 16) * 32;            // one bf16 plane of the halo chunk: [row][16 channels], + 16 zero rows (one per colour)
constexpr int TC_HALO_BYTES = ((3 * TC_PLANE_BYTES + 255) / 256) * 256;
constexpr int TC_WBUF_UNITS = 768;                             // 16-byte units of one step's weight pieces (12 KB)
constexpr int TC_LOC_BYTES = TC_KMAX * TC_TR * 2;
constexpr int TC_HID_BYTES = TC_HCAP * 4;
constexpr int TC_LDS_BYTES = TC_HALO_BYTES + 2 * TC_WBUF_UNITS * 16 + TC_LOC_BYTES + TC_HID_BYTES + TC_TR * 4 + TC_TR * 4;
constexpr int TC_THREADS = 256;

// Epilogue of a tile (no LayerNorm, float4-aligned operands): all 128 rows at once.  The accumulators are transposed through LDS
// (the halo planes and the weight buffers are dead by now: 128 x NT*32 floats <= 64 KB fit in front of s_loc) with ONE barrier pair,
// then every thread owns one float4 column group (its scale / shift are loaded once, before the barrier) and 128 / RPI rows, whose
// residual / pair operands are fetched in batches of branch-free loads: one memory latency per batch.  gg_epilogue (two passes of 64
// rows, a conditional load chain per row group) took 18 of a 128 -> 128 tile's 204 us (tools/trace_tile.py).  Same arithmetic per
// element, in the same order: results are bit-identical to gg_epilogue's.
// CH (chained launch): the output rows are read by other XCDs later in the SAME launch - they are written through to memory with coherent
// (sc1) 16-byte stores (common.h: ls3d_store4_agent).  The residual / pair operands - rows another XCD wrote earlier in this launch - are
// read with ordinary cached loads: see the note on coherence at k_tile_conv.
template <int NT, bool CH = false>
__device__ __forceinline__ void tc_epilogue(f32x16 (&acc)[NT], float *stage, const int *s_rows, int wave, int kk, int col, int cout, const EpiDev &e,
                                            float *__restrict__ out, int out_ld) {
  constexpr int SLAB = NT * 32, C4 = SLAB / 4, RPI = TC_THREADS / C4, ITER = TC_TR / RPI, BATCH = ITER < 8 ? ITER : 8;
  static_assert(TC_TR * SLAB * 4 <= TC_HALO_BYTES + 2 * TC_WBUF_UNITS * 16, "the transposed tile fits in front of s_loc");
  const int tid = threadIdx.x, c4 = tid % C4, lr0 = tid / C4, oc = c4 * 4;
  const bool oncol = oc < cout;
  const int occ = oncol ? oc : 0;  // clamped column of the branch-free loads
  __syncthreads();                 // the MFMA loop's readers are done with the halo planes and the weight buffers
  {
    float *dst = stage + (wave * 32 + 4 * kk) * SLAB + col;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * SLAB + n * 32] = acc[n][r];
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e.scale) sc = *(const float4 *)(e.scale + occ);
  if (e.shift) sh = *(const float4 *)(e.shift + occ);
  __syncthreads();
  // rows of a partial tile that do not exist (s_rows < 0) still issue the branch-free operand loads.  In a chained launch they must read a
  // row THIS tile's own dependencies cover - its first slot, live in every live tile (k_tile_build puts the empty slots last) - not row 0
  // of the buffer: that row belongs to some other tile, which may still be writing it, and an ordinary cached load would leave a stale
  // line of it in this XCD's L2 for a later halo / pair load of the same launch to hit.
  const int dead_row = CH ? s_rows[0] : 0;
#pragma unroll
  for (int b0 = 0; b0 < ITER; b0 += BATCH) {
    int orow[BATCH];
    float4 q[BATCH], p0[BATCH], p1[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) orow[j] = s_rows[lr0 + (b0 + j) * RPI];
    {
    if (e.res_pre) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) q[j] = *(const float4 *)(e.res_pre + (size_t)(orow[j] >= 0 ? orow[j] : dead_row) * e.res_pre_ld + occ);
    }
    if (e.pair) {
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const float *pp = e.pair + (size_t)(orow[j] >= 0 ? orow[j] : dead_row) * e.pair_ld + 2 * occ;
        p0[j] = *(const float4 *)pp;
        p1[j] = *(const float4 *)(pp + 4);
      }
    }
    }
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      float4 v = *(const float4 *)(stage + (lr0 + (b0 + j) * RPI) * SLAB + oc);
      if (e.scale) { v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
      if (e.shift) { v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w; }
      if (e.res_pre) { v.x += q[j].x; v.y += q[j].y; v.z += q[j].z; v.w += q[j].w; }
      if (e.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
      if (e.pair) { v.x += p0[j].x + p0[j].y; v.y += p0[j].z + p0[j].w; v.z += p1[j].x + p1[j].y; v.w += p1[j].z + p1[j].w; }
      if constexpr (CH) {
        if (oncol && orow[j] >= 0) ls3d_store4_agent(ls3d_cohbuf_make(out), ((unsigned)orow[j] * (unsigned)out_ld + (unsigned)oc) * 4u, v);
      } else {
        if (oncol && orow[j] >= 0) *(float4 *)(out + (size_t)orow[j] * out_ld + oc) = v;
      }
    }
  }
}

// ---- chained launch (ls3d_tile_conv_chain): up to TC_CHAIN_MAX layers on ONE plan in one persistent launch.
// Work units (layer, tile [, half]) are handed out in layer-major order by a ticket counter; a unit of layer l > 0 first waits until the
// tiles that own its halo rows (the plan's producer lists) have finished layer l - 1.  A unit only ever waits for units with smaller
// tickets, every ticket is held by a running workgroup, so the smallest outstanding ticket never waits: no deadlock whatever the number
// of resident workgroups.  What it buys: the tail of a launch - 677 tiles on 512 workgroup slots leave the second round 32 % full - is
// filled with the next layer's tiles.
constexpr int TC_CHAIN_MAX = 8;
template <typename T>
__device__ __forceinline__ T *tc_uniform(T *p) {  // a pointer every lane holds the same value of -> scalar registers
  const unsigned long long v = (unsigned long long)(uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (T *)(uintptr_t)(((unsigned long long)hi << 32) | lo);
}
struct TcLayer {
  const float *in;
  const uint4 *wpk;
  float *out;
  EpiDev e;
  int in_ld, cin, cout, out_ld, n_split, pad_;
};
// state (ints): [0] ticket, [1] error (1: a wait ran into the watchdog), [2] total units, [3 .. 3 + TC_CHAIN_MAX] first ticket of each
// layer (+ the total again), [16 .. 16 + TC_CHAIN_MAX) finished tiles per layer, [32 .. 32 + T) layers finished per tile;
// the layer table follows at byte TC_CH_LAYERS_OFF(T)
constexpr int TC_CH_USTART = 3, TC_CH_LDONE = 16, TC_CH_DONE = 32;
constexpr int TC_CH_SPIN_MAX = 1 << 22;  // ~1 s of polling: a wait that long is a bug (producers take their tickets before their consumers and never
//                                         wait for later tickets) - the unit flags state[1] and TRAPS: the process dies loudly instead of publishing rows
//                                         computed from a halo that was not there yet
static inline size_t tc_chain_layers_off(int ntiles) { return tc_align((size_t)(TC_CH_DONE + ntiles) * 4); }
struct TcChain {
  const TcLayer *layers;
  int *state;
  int n_layers;
};
struct TcChainSetup {
  TcLayer l[TC_CHAIN_MAX];
};

// one workgroup in front of the chained launch: zeroes the ticket / completion counters, decides every layer's split over the input
// channels from the plan's live tile count (the rule of ls3d_tile_conv) and writes the layer table to device memory
__global__ __launch_bounds__(256) void k_tile_chain_setup(TilePlan p, TcChainSetup su, int n_layers, int split_small, TcLayer *dst, int *state) {
  const int tid = threadIdx.x;
  const int tlive = p.torder[p.ntiles];
  for (int i = tid; i < p.ntiles; i += 256) state[TC_CH_DONE + i] = 0;
  if (tid < TC_CH_DONE) state[tid] = 0;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
#pragma unroll
    for (int l = 0; l < TC_CHAIN_MAX; ++l) {
      if (l < n_layers) {
        TcLayer L = su.l[l];
        L.n_split = (L.n_split && tlive <= split_small) ? tlive : 0;  // host: n_split = 1 when the layer may split (cin >= 64, workspace + counters given)
        dst[l] = L;
        state[TC_CH_USTART + l] = run;
        run += tlive + L.n_split;
      }
    }
    for (int l = n_layers; l <= TC_CHAIN_MAX; ++l) state[TC_CH_USTART + l] = run;
    state[2] = run;
  }
}

// Workgroup = 4 waves over one tile: wave w owns rows [32 w, 32 w + 32) and all NT 32-column blocks (cout <= 32: NT = 1,
// <= 64: NT = 2, <= 128: NT = 4).  Two workgroups fit a CU (LDS 79 KB).  A step = G = 4 / NT consecutive active kernel offsets of
// one 16-channel chunk, so that every step moves the same 12 KB of weights and feeds 32 MFMAs per wave between two barriers.
// The loop around the MFMAs is kept to a few dozen instructions per step — a wave issues one instruction every ~4 cycles, so
// every 8 instructions of bookkeeping cost as much as one MFMA: weights go global -> LDS by LDS-DMA from a scalar base (3 blocks
// of 1 KB per wave and step, issued one step ahead into the other buffer, no staging registers, no address VALU), the local
// indices of the A fragments are read one step ahead, every per-wave decision is a scalar branch.
// NP = plane products per f32 product (6 or 8).
// TR = tracing build of the same kernel (flags bit 5 of ls3d_tile_conv): every wave of every unit writes one TC_TRACE_WORDS-word record -
// when and where it ran and how its cycles split into prologue / halo staging / waits at the step barriers / epilogue (the rest is
// the MFMA loop itself) - for tools/trace_tile.py.  The product instantiations (TR = false) carry none of it.
// LP = 1 (NT >= 2, NP == 6): the offset loop software-pipelined one offset ahead with the step barrier in the MIDDLE of an offset's
// MFMAs (see the loop); LP = 0: fragments read at the start of the step that uses them (NT = 1, NP = 1 / 8, flags bit 0).
constexpr int TC_TRACE_WORDS = 16;
// CH = chained launch (ls3d_tile_conv_chain): persistent workgroups take (layer, tile [, half]) units from a ticket counter, the layer's operands
// come from the device-side layer table, a unit of layer l > 0 waits for the producer tiles of its halo at layer l - 1.  Tiles are taken in the
// plan's spatial order (neighbours finish close together).  The arithmetic of a unit is the same code: results are bit-identical to
// layer-by-layer launches.
// Coherence inside the launch (the eight XCDs' L2s do not snoop each other): a tile's output rows are WRITTEN THROUGH (sc1 stores), every
// wave waits for its stores' acknowledgements, then the tile's completion counter is set with an sc1 store; a consumer polls that counter
// with sc1 loads and then reads the rows with ORDINARY loads.  That is safe because no L2 can hold a stale copy of such a line: nobody reads
// an output row of the launch before its producer has finished it (the wait above), a 128-byte line belongs to ONE row (the host takes only
// layers of >= 32 output channels with 128-byte-aligned rows into a chain), hence to one writer, and lines cached before the launch were
// dropped at its start like at any kernel boundary.  The first reader on an XCD misses and fetches the written-through data; later readers on
// that XCD hit.  (First build of the round: sc1 LOADS for the halo - 10 % slower than layer-by-layer launches: the halo loads of the next chunk
// are issued one step ahead of a `s_waitcnt vmcnt(0)`, which an L2 hit meets and a trip to memory does not.)
template <int NT, int NP, bool TR = false, int LP = 0, bool CH = false>
__global__ __launch_bounds__(TC_THREADS, 2) void k_tile_conv(const float *__restrict__ in_a, int in_ld_a, TilePlan p, const uint4 *__restrict__ wpk_a,
                                                             int cin_a, int cout_a, EpiDev e_a, float *__restrict__ out_a, int out_ld_a, int ablate, int swz,
                                                             int split_small, int split_tail, int split_forced, float *partial, int *counters,
                                                             unsigned *trace, TcChain ch) {
  static_assert(!(CH && TR), "no tracing build of the chained launch");
  constexpr int PLN = NP == 1 ? 1 : 3;     // weight planes in the packed layout (NP == 1: the head plane only, ls3d_tile_conv_pack_bf16)
  constexpr int PU = NT * 64 * PLN;        // 16-byte units of one (offset, chunk) weight piece
  constexpr int PB = NT * PLN;             // ... in 1 KB LDS-DMA blocks
  constexpr int G = 12 / PB;               // offsets per step
  constexpr int HPT = (TC_HCAP * 4 + TC_THREADS - 1) / TC_THREADS;  // float4 items of the halo chunk per thread
  static_assert(G * PU == TC_WBUF_UNITS && G * PB == 12, "a step moves 12 KB of weights");
  static_assert(64 * NT * 32 * 4 <= TC_HALO_BYTES, "the epilogue transposes 64 rows at a time through the halo buffer");
  HIP_DYNAMIC_SHARED(char, smem)
  uint4 *Bs = (uint4 *)(smem + TC_HALO_BYTES);                                       // [2][TC_WBUF_UNITS]
  uint16_t *s_loc = (uint16_t *)(smem + TC_HALO_BYTES + 2 * TC_WBUF_UNITS * 16);     // [kvol][128]
  int *s_hid = (int *)((char *)s_loc + TC_LOC_BYTES);                                // [TC_HCAP] halo rows of this pass
  int *s_rows = s_hid + TC_HCAP;                                                     // [128]
  float *s_stat = (float *)(s_rows + TC_TR);                                         // [128]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                         // scalar: per-wave decisions are s_cbranch
  const int col = lane & 31, kk = lane >> 5;
  const int kvol = p.kvol;
  // this wave's share of a step's weight DMA: blocks [3 wave, 3 wave + 3) of the 12; with three planes all inside one offset's piece
  // (dma_g, from block dma_j on), with one plane each block in its own (offset, block) - TC_DMA_GROUP
  const int dma_g = (3 * wave) / PB, dma_j = (3 * wave) % PB;
  const unsigned voff0 = (unsigned)lane * 16u, voff1 = voff0 + 1024u, voff2 = voff0 + 2048u;
  const uint16_t *loc_w = s_loc + wave * 32 + col;
  uint4 *const dma_lds0 = Bs + dma_g * PU + dma_j * 64, *const dma_lds1 = dma_lds0 + TC_WBUF_UNITS;
  // One workgroup per work unit, dispatched in the plan's most-expensive-first order (p.torder).  The LAST n_split tiles of that
  // order - the cheapest ones, the tail of the launch - are two units each, over half of the 16-channel chunks: the unit that
  // finishes second adds the other's partial sums (through `partial`, [split tile][part][NT * 16][256] floats) and runs the
  // epilogue.  Half-size units at the end of the dispatch order keep the last round of a launch from running full-size tiles alone
  // on their CUs (677 tiles on 512 workgroup slots), and give launches of fewer tiles than slots two workgroups per CU.
  // The split is decided HERE, from the plan's LIVE tile count (tiles with rows below the device row count), not from the table's
  // capacity: a plan built on spare rows (capacity mode) and the plan of the exact table run the same units on the same tiles.
  // The launch covers the worst case; workgroups beyond the units of the live tiles leave at once.
  const int tlive = p.torder[p.ntiles];
  [[maybe_unused]] int ch_next = -1;  // thread 0: the next ticket, drawn early
  for (;;) {  // CH: one work unit per ticket until the tickets run out; otherwise one pass (the unit of this workgroup)
    const float *__restrict__ in = in_a;
    const uint4 *__restrict__ wpk = wpk_a;
    float *__restrict__ out = out_a;
    EpiDev e = e_a;
    int in_ld = in_ld_a, cin = cin_a, cout = cout_a, out_ld = out_ld_a;
    int n_split, blk;
    [[maybe_unused]] int ch_layer = 0;
    if constexpr (CH) {
      __syncthreads();  // the previous unit of this workgroup is done with LDS
      // (the ticket of this unit was drawn while the previous unit ran its epilogue - thread 0 holds it: the atomic's round trip to memory is
      // hidden; a unit still only waits for smaller tickets, and the holder of a drawn-but-not-started ticket runs a smaller one)
      if (tid == 0) *(int *)s_stat = ch_next >= 0 ? ch_next : atomicAdd(ch.state, 1);
      ch_next = -1;
      __syncthreads();
      const int u = __builtin_amdgcn_readfirstlane(*(const int *)s_stat);
      if (u >= ch.state[2]) return;
      for (int l = 1; l < ch.n_layers; ++l)
        if (u >= ch.state[TC_CH_USTART + l]) ch_layer = l;
      blk = u - ch.state[TC_CH_USTART + ch_layer];
      // the layer's operands, made wave-uniform explicitly (scalar registers: the weight DMA takes a scalar base, the coherent accesses a
      // scalar buffer descriptor)
      // (what the MFMA loop needs now; the epilogue's operands are read from the table when the epilogue starts - fewer live registers)
      const TcLayer *L = ch.layers + ch_layer;
      in = tc_uniform(L->in); wpk = tc_uniform(L->wpk);
      in_ld = __builtin_amdgcn_readfirstlane(L->in_ld); cin = __builtin_amdgcn_readfirstlane(L->cin);
      n_split = __builtin_amdgcn_readfirstlane(L->n_split);
    } else {
      // split_tail < 0: the tiles beyond the last full round of TC_SPLIT_MAX workgroup slots (the tail that would run one per CU)
      n_split = split_forced >= 0 ? split_forced : (tlive <= split_small ? tlive : (split_tail < 0 ? tlive % 512 : split_tail));  // scalar
      blk = (int)blockIdx.x;
    }
    if (n_split > tlive) n_split = tlive;
    const int nfull = tlive - n_split;
    if (blk >= nfull + 2 * n_split) return;
    const int nchunk = cin / 16;
    int tile, part = 0, ksplit = 1, sidx = 0;
    if (blk < nfull) {
      tile = (CH && !(ablate & 1024)) ? blk : p.torder[blk];
    } else {
      const int v = blk - nfull;
      sidx = v >> 1;
      tile = CH ? nfull + sidx : p.torder[nfull + sidx];
      part = v & 1;
      ksplit = 2;
    }
    const int *meta = p.tmeta + (size_t)tile * TC_META;
    const int H = meta[0];
    const unsigned kmask = (unsigned)meta[1];
    const unsigned wmask = (unsigned)__builtin_amdgcn_readfirstlane(meta[2 + wave]);
    if (meta[6] == 0) {
      if constexpr (CH) continue; else return;
    }
    [[maybe_unused]] ls3d_cohbuf in_buf;
    if constexpr (CH) in_buf = ls3d_cohbuf_make(in);
    if constexpr (CH) {
      if (ch_layer > 0 && !(ablate & 256)) {
        // the tiles that own this tile's halo rows must have finished the previous layer (its output rows are this layer's halo, residual
        // and pair operands; everything older follows by induction: a tile is its own producer).  One lane per producer polls its
        // completion counter with coherent loads; more than TC_DEPCAP producers: wait for the whole previous layer.
        const int nd = meta[7];
        if (nd >= 0) {
          if (tid < nd) {
            const int *flag = ch.state + TC_CH_DONE + p.tdep[(size_t)tile * TC_DEPCAP + tid];
            for (int spins = 0; ls3d_load_agent_i32(flag) < ch_layer; ++spins) {
              ls3d_sleep();
              if (spins > TC_CH_SPIN_MAX) { ls3d_store_agent_i32(ch.state + 1, 1); LS3D_TRAP(); break; }
            }
          }
        } else if (tid == 0) {
          const int *flag = ch.state + TC_CH_LDONE + ch_layer - 1;
          for (int spins = 0; ls3d_load_agent_i32(flag) < tlive; ++spins) {
            ls3d_sleep();
            if (spins > TC_CH_SPIN_MAX) { ls3d_store_agent_i32(ch.state + 1, 1); LS3D_TRAP(); break; }
          }
        }
        __syncthreads();
      }
    }
    unsigned long long tr_w0 = 0, tr_t0 = 0, tr_mark = 0;
    unsigned tr_pro = 0, tr_stage = 0, tr_bar = 0, tr_steps = 0;
    if constexpr (TR) { tr_w0 = ls3d_walltime(); tr_t0 = ls3d_cycles(); }
#define TC_TRACE_WRITE(done_)                                                                        \
  if constexpr (TR) {                                                                                \
    const unsigned long long t1_ = ls3d_cycles(), w1_ = ls3d_walltime();                             \
    if (lane == 0) {                                                                                 \
      unsigned *rec_ = trace + ((size_t)blockIdx.x * 4 + wave) * TC_TRACE_WORDS;                     \
      rec_[0] = (unsigned)tr_w0; rec_[1] = (unsigned)(tr_w0 >> 32);                                  \
      rec_[2] = (unsigned)w1_; rec_[3] = (unsigned)(w1_ >> 32);                                      \
      rec_[4] = ls3d_hw_id(); rec_[5] = ls3d_xcc_id(); rec_[6] = (unsigned)tile;                     \
      rec_[7] = (unsigned)part | ((unsigned)ksplit << 8) | ((unsigned)(done_) << 16);                \
      rec_[8] = (unsigned)H; rec_[9] = (unsigned)__popc(kmask) | ((unsigned)__popc(wmask) << 8);     \
      rec_[10] = (unsigned)(t1_ - tr_t0); rec_[11] = tr_pro; rec_[12] = tr_stage; rec_[13] = tr_bar; \
      rec_[14] = (unsigned)(t1_ - tr_mark); rec_[15] = tr_steps;                                     \
    }                                                                                                \
  }
    if (tid < TC_TR) s_rows[tid] = p.trow[(size_t)tile * TC_TR + tid];
    {
      const uint4 *src = (const uint4 *)(p.tloc + (size_t)tile * kvol * TC_TR);
      for (int i = tid; i < kvol * (TC_TR / 8); i += TC_THREADS) ((uint4 *)s_loc)[i] = src[i];
    }
    for (int i = tid; i < 3 * 128; i += TC_THREADS) ((unsigned *)(smem + (i >> 7) * TC_PLANE_BYTES + TC_HCAP * 32))[i & 127] = 0u;  // the 16 zero rows of each plane
    f32x16 acc[NT], acs[NT];  // head x head products / everything else (see the MFMA block)
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = acs[n][r] = 0.0f;
    const int nseg = (H + TC_HCAP - 1) / TC_HCAP;
    if constexpr (TR) tr_pro = (unsigned)(ls3d_cycles() - tr_t0);
    for (int seg = 0; seg < (kmask ? nseg : 0); ++seg) {
      const int seg_lo = seg * TC_HCAP;
      const int nh = (H - seg_lo) < TC_HCAP ? (H - seg_lo) : TC_HCAP;
      __syncthreads();  // the previous pass (or the previous tile's epilogue) is done with s_hid and the halo buffer
      for (int i = tid; i < nh; i += TC_THREADS) s_hid[i] = p.thalo[(size_t)tile * p.hs + seg_lo + i];
      __syncthreads();
      // The halo chunk goes global f32 -> registers -> three bf16 planes in LDS, software-pipelined over the chunks: the loads of
      // chunk c + 1 are issued when chunk c's MFMA steps start and land while they run (the first step's wait for its weight DMA
      // also waits for them: they have had that step's MFMAs to arrive), so only the split + LDS writes of a chunk stay exposed.
      // Branch-free loads (rows past the end re-read the last one): under a per-load condition hipcc waits for each load before it
      // issues the next, i.e. HPT memory latencies instead of one.
      float4 hv[HPT];
#define TC_LOAD_HALO(c_)                                                                  \
  _Pragma("unroll") for (int j = 0; j < HPT; ++j) {                                       \
    const int i = tid + j * TC_THREADS, hrow = i >> 2, q = i & 3;                         \
    const int hr = hrow < nh ? hrow : nh - 1;                                             \
    if constexpr (CH)  /* scalar base + 32-bit offsets: fewer address registers (the chained build is at the register limit) */ \
      hv[j] = ls3d_load4_buf(in_buf, ((unsigned)s_hid[hr] * (unsigned)in_ld + (unsigned)((c_) * 16 + q * 4)) * 4u); \
    else                                                                                  \
      hv[j] = *(const float4 *)(in + (size_t)s_hid[hr] * in_ld + (c_) * 16 + q * 4);      \
  }
      const int c_lo = nchunk * part / ksplit, c_hi = nchunk * (part + 1) / ksplit;
      if (!(ablate & 16)) { TC_LOAD_HALO(c_lo) }
      for (int c = c_lo; c < c_hi; ++c) {
        if constexpr (TR) tr_mark = ls3d_cycles();
        if (c > c_lo) __syncthreads();  // the previous chunk's MFMAs have read their fragments
        unsigned rem = kmask;
        int ks[G], kn[G];
#define TC_NEXT_GROUP(dst_)                                                       \
  _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                              \
    dst_[g_] = rem ? __ffs((int)rem) - 1 : -1;                                    \
    rem &= rem - 1;                                                               \
  }
        // 32-bit scalar arithmetic per step: piece(k, c) = wchunk + k * kstride bytes (a layer's packed weights are < 4 GB)
        const char *wchunk = (const char *)(wpk + (size_t)c * PU + dma_j * 64);
        const unsigned kstride = (unsigned)nchunk * PU * 16u;
#define TC_DMA_GROUP(kg_, buf_)                                                   \
  if constexpr (PLN == 3) {                                                       \
    if (kg_[dma_g] >= 0 && !(ablate & 8))                                         \
      ls3d_glds16x3(wchunk + (unsigned)kg_[dma_g] * kstride, voff0, voff1, voff2, (buf_) ? dma_lds1 : dma_lds0); \
  } else {                                                                        \
    _Pragma("unroll") for (int q_ = 0; q_ < 3; ++q_) {                            \
      const int blk_ = 3 * wave + q_, g1_ = blk_ / PB, j1_ = blk_ % PB;           \
      if (kg_[g1_] >= 0 && !(ablate & 8))                                         \
        ls3d_glds16((const char *)(wpk + (size_t)c * PU + j1_ * 64) + (unsigned)kg_[g1_] * kstride + voff0, \
                    Bs + ((buf_) ? TC_WBUF_UNITS : 0) + g1_ * PU + j1_ * 64);     \
    }                                                                             \
  }
        // LP = 1: the DMA walks the active offsets with its own iterator, G per step, two steps ahead of the MFMAs
        unsigned rem_d = kmask;
#define TC_DMA_STEP(buf_)                                                          \
  {                                                                               \
    int kd_ = -1;                                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < G; ++g_) {                            \
      const int k_ = rem_d ? __ffs((int)rem_d) - 1 : -1;                          \
      rem_d &= rem_d - 1;                                                         \
      if (g_ == dma_g) kd_ = k_;                                                  \
    }                                                                             \
    if (kd_ >= 0 && !(ablate & 8))                                                \
      ls3d_glds16x3(wchunk + (unsigned)kd_ * kstride, voff0, voff1, voff2, (buf_) ? dma_lds1 : dma_lds0); \
  }
        if constexpr (LP == 0) {
          TC_NEXT_GROUP(ks)
          TC_DMA_GROUP(ks, 0)
        } else {
          TC_DMA_STEP(0)
        }
        if (!(ablate & 16)) {
#pragma unroll
          for (int j = 0; j < HPT; ++j) {
            const int i = tid + j * TC_THREADS, hrow = i >> 2, q = i & 3;
            if (hrow < nh) {
              uint2 h, m, l;
              ls3d_split_pair3_rne(hv[j].x, hv[j].y, h.x, m.x, l.x);
              ls3d_split_pair3_rne(hv[j].z, hv[j].w, h.y, m.y, l.y);
              // LDS bank swizzle: a fragment read takes one 16-byte half of 16 gathered rows per lane group; with the halves of
              // rows r and r + 8 swapped the 16 lanes use all 16 slots of the 256-byte bank row instead of 8 (tools/sim_lds_conflicts.py)
              char *dst = smem + hrow * 32 + ((q * 8) ^ (((hrow >> 3) & swz) << 4));
              *(uint2 *)(dst) = h;
              if constexpr (NP > 1) {  // NP == 1 (plain bf16 arithmetic, BASELINE configs[4]): the head plane is all there is
                *(uint2 *)(dst + TC_PLANE_BYTES) = m;
                *(uint2 *)(dst + 2 * TC_PLANE_BYTES) = l;
              }
            }
          }
        }
        LS3D_WAIT_VMCNT(0);
        __syncthreads();
        if constexpr (TR) tr_stage += (unsigned)(ls3d_cycles() - tr_mark);
        if (c + 1 < c_hi && !(ablate & 16)) { TC_LOAD_HALO(c + 1) }
        int buf = 0;
        if constexpr (LP == 1) {
          // Software-pipelined offset loop.  tools/trace_tile.py: a workgroup alone on its CU (the tail of a 677-tile launch) keeps the
          // matrix pipe 50 % busy, two sharing a CU 70 %: with the fragments read at the top of the step that uses them every step starts
          // with an empty pipe (barrier, bookkeeping, DMA issue, 11 LDS reads before the first MFMA), and what a wave issues between
          // two MFMA groups is not hidden by anything when it is alone on its SIMD.  Here every non-MFMA instruction sits in the shadow
          // of a group of NT MFMAs (NT x 32 cycles of matrix pipe):
          //   * while offset j's MFMAs run, offset j + 1's fragments are read - the halo planes are static for the whole chunk, and the
          //     weights of the next STEP become readable in the middle of the step's last offset: the barrier sits between the b0
          //     products and the b1 / b2 products, after this wave's last read of the current weight buffer (b2) and after its share
          //     of the next buffer's DMA (issued a whole step earlier) has landed; behind it the DMA of the step after next goes into
          //     the buffer just released;
          //   * the l and m planes of the next offset replace the current ones as soon as their last product is issued, only the head
          //     plane is double-buffered (registers: 256 with the accumulators' 128);
          //   * no per-wave skipping of offsets (94-99 % of the (wave, offset) blocks are active where this loop runs; an absent
          //     neighbour reads the zero row), no conditional code inside an offset: the offsets of a step are unrolled, the last offset
          //     of a chunk is peeled (nothing to prefetch, no barrier).
          // Per accumulator the products are issued in the same order as in the LP = 0 loop: results are bit-identical.
          static_assert(NP == 6 && NT >= 2, "pipelined loop: 6-product kernels with >= 2 column blocks");
          constexpr int PL = TC_PLANE_BYTES / 16;
          TC_DMA_STEP(1)
          unsigned rem_c = kmask;
#define TC_POP(dst_)                                  \
  {                                                   \
    dst_ = rem_c ? __ffs((int)rem_c) - 1 : -1;        \
    rem_c &= rem_c - 1;                               \
  }
#define TC_HALO_PTR(raw_, dst_)                                                                        \
  {                                                                                                    \
    const int rw_ = (raw_), li_ = rw_ - seg_lo;                                                        \
    const int lz_ = ((unsigned)li_ < (unsigned)TC_HCAP) ? li_ : TC_HCAP + (rw_ & 15); /* absent / other pass -> a zero row (its colour: tc_color) */ \
    dst_ = (const uint4 *)smem + lz_ * 2 + (kk ^ ((lz_ >> 3) & swz));                                  \
  }
#define TC_MFMA(dst_, a_, b_)                                                                         \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                      \
      dst_[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, __builtin_bit_cast(bf16x8, b_[n]), dst_[n], 0, 0, 0);
          int k1, k2;  // the next offset and the one after it (-1: none)
          {
            int k0;
            TC_POP(k0) TC_POP(k1) TC_POP(k2)
            (void)k0;
          }
          bf16x8 ah, am, al;
          uint4 b0[NT], b1[NT], b2[NT];
          const uint4 *hpn;  // this lane's row at the NEXT offset: address of its head-plane fragment
          int loc2;          // ... at the offset after next: raw local index
          {
            const uint4 *hp, *bs = Bs + lane;
            TC_HALO_PTR((int)loc_w[(__ffs((int)kmask) - 1) * TC_TR], hp)
            ah = __builtin_bit_cast(bf16x8, hp[0]);
            am = __builtin_bit_cast(bf16x8, hp[PL]);
            al = __builtin_bit_cast(bf16x8, hp[2 * PL]);
#pragma unroll
            for (int n = 0; n < NT; ++n) { b0[n] = bs[(n * 3 + 0) * 64]; b1[n] = bs[(n * 3 + 1) * 64]; }
            TC_HALO_PTR((int)loc_w[(k1 >= 0 ? k1 : 0) * TC_TR], hpn)
            loc2 = loc_w[(k2 >= 0 ? k2 : 0) * TC_TR];
          }
          // one offset.  JG_: its position in the step (compile time); BAR_: it is the last offset of a step that has a successor (barrier +
          // DMA); PRE_: an offset follows (prefetch its fragments; JGN_ / buffer bufn_: where its weights are)
#define TC_OFFSET(JG_, BAR_, PRE_, JGN_, bufn_)                                                                      \
  {                                                                                                                  \
    const uint4 *bs_ = Bs + buf * TC_WBUF_UNITS + (JG_) * PU + lane;                                                 \
    const uint4 *bsn_ = Bs + (bufn_) * TC_WBUF_UNITS + (JGN_) * PU + lane;                                           \
    _Pragma("unroll") for (int n = 0; n < NT; ++n) b2[n] = bs_[(n * 3 + 2) * 64];                                    \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acs, al, b0)                                                                                             \
    if (PRE_) al = __builtin_bit_cast(bf16x8, hpn[2 * PL]);                                                          \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acs, am, b0)                                                                                             \
    int k3_;                                                                                                         \
    TC_POP(k3_)                                                                                                      \
    const int loc3_ = (PRE_) ? (int)loc_w[(k3_ >= 0 ? k3_ : 0) * TC_TR] : 0;                                         \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acc, ah, b0)                                                                                             \
    LS3D_SCHED_FENCE();                                                                                              \
    if constexpr (TR) ++tr_steps;                                                                                    \
    if (BAR_) {                                                                                                      \
      if constexpr (TR) tr_mark = ls3d_cycles();                                                                     \
      LS3D_WAIT_VMCNT(0);                                                                                            \
      __syncthreads(); /* every wave: done reading this step's weights, its share of the next step's has landed */   \
      if constexpr (TR) tr_bar += (unsigned)(ls3d_cycles() - tr_mark);                                               \
    }                                                                                                                \
    bf16x8 ahn_ = ah;                                                                                                \
    if (PRE_) {                                                                                                      \
      ahn_ = __builtin_bit_cast(bf16x8, hpn[0]);                                                                     \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) b0[n] = bsn_[(n * 3 + 0) * 64];                                 \
    }                                                                                                                \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acs, am, b1)                                                                                             \
    if (PRE_) am = __builtin_bit_cast(bf16x8, hpn[PL]);                                                              \
    if (BAR_) TC_DMA_STEP(buf)                                                                                       \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acs, ah, b1)                                                                                             \
    if (PRE_) {                                                                                                      \
      _Pragma("unroll") for (int n = 0; n < NT; ++n) b1[n] = bsn_[(n * 3 + 1) * 64];                                 \
      TC_HALO_PTR(loc2, hpn)                                                                                         \
    }                                                                                                                \
    LS3D_SCHED_FENCE();                                                                                              \
    TC_MFMA(acs, ah, b2)                                                                                             \
    LS3D_SCHED_FENCE();                                                                                              \
    ah = ahn_;                                                                                                       \
    loc2 = loc3_;                                                                                                    \
    k1 = k2;                                                                                                         \
    k2 = k3_;                                                                                                        \
  }
          if constexpr (G == 1) {
            while (k1 >= 0) {
              TC_OFFSET(0, true, true, 0, buf ^ 1)
              buf ^= 1;
            }
            TC_OFFSET(0, false, false, 0, buf)
          } else {
            static_assert(G == 2, "two column blocks: two offsets per step");
            while (k2 >= 0) {  // a whole step with a successor
              TC_OFFSET(0, false, true, 1, buf)
              TC_OFFSET(1, true, true, 0, buf ^ 1)
              buf ^= 1;
            }
            int jg_last = 0;   // the last step: one or two offsets (no early exits above: the accumulators stay where they are)
            if (k1 >= 0) {
              TC_OFFSET(0, false, true, 1, buf)
              jg_last = 1;
            }
            TC_OFFSET(jg_last, false, false, 0, buf)
          }
#undef TC_OFFSET
#undef TC_MFMA
#undef TC_POP
#undef TC_HALO_PTR
        } else {
        int lc[G], ln[G];  // raw local indices of this lane's row: current step / next step
#pragma unroll
        for (int g = 0; g < G; ++g) lc[g] = loc_w[(ks[g] >= 0 ? ks[g] : 0) * TC_TR];
        for (;;) {
          TC_NEXT_GROUP(kn)
          TC_DMA_GROUP(kn, buf ^ 1)
#pragma unroll
          for (int g = 0; g < G; ++g) ln[g] = loc_w[(kn[g] >= 0 ? kn[g] : 0) * TC_TR];  // branch-free: used one step later
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int k = ks[g];
            if (k >= 0 && ((wmask >> k) & 1u) && !(ablate & 4)) {
              const int li = lc[g] - seg_lo;
              const int lz = ((unsigned)li < (unsigned)TC_HCAP) ? li : TC_HCAP + (lc[g] & 15);  // absent / other pass -> a zero row
              const uint4 *hp = (const uint4 *)smem + lz * 2 + (kk ^ ((lz >> 3) & swz));
              const uint4 *bs = Bs + buf * TC_WBUF_UNITS + g * PU + lane;
              // weight fragments plane by plane (the MFMAs are grouped by the weight plane they need): plane 2 is read after plane
              // 0's MFMAs are issued and takes over its registers: 32 instead of 48 VGPRs of fragments
              uint4 b0[NT], b1[NT];
#pragma unroll
              for (int n = 0; n < NT; ++n) b0[n] = bs[(n * PLN + 0) * 64];
              const bf16x8 ah = __builtin_bit_cast(bf16x8, hp[0]);
              if constexpr (NP == 1) {  // bf16 x bf16 -> f32: one product per f32 product, operands rounded to bf16
#pragma unroll
                for (int n = 0; n < NT; ++n)
                  acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, b0[n]), acc[n], 0, 0, 0);
                continue;
              }
#pragma unroll
              for (int n = 0; n < NT; ++n) b1[n] = bs[(n * 3 + 1) * 64];
              const bf16x8 am = __builtin_bit_cast(bf16x8, hp[TC_PLANE_BYTES / 16]);
              const bf16x8 al = __builtin_bit_cast(bf16x8, hp[2 * (TC_PLANE_BYTES / 16)]);
              // product-major: consecutive MFMAs go to different accumulators
#define TC_MFMA(dst_, a_, b_)                                                                           \
  _Pragma("unroll") for (int n = 0; n < NT; ++n)                                                        \
      dst_[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_, __builtin_bit_cast(bf16x8, b_[n]), dst_[n], 0, 0, 0);
              // head x head (99.6 % of the value) goes to `acc`, the seven (five) small products to `acs`: the bf16 MFMA's
              // accumulate is not round-to-nearest (measured: ~1e-3 ulp of the accumulator lost towards zero per MFMA), and
              // eight MFMAs per step into one accumulator made the end-to-end error 3.3x the exact-f32 path's.  With the small
              // terms kept apart the large accumulator sees one MFMA per step and the bias of the small one is 2^-8 of its own.
              TC_MFMA(acs, al, b0) TC_MFMA(acs, am, b0) TC_MFMA(acc, ah, b0)
              LS3D_SCHED_FENCE();
              uint4 b2[NT];
#pragma unroll
              for (int n = 0; n < NT; ++n) b2[n] = bs[(n * 3 + 2) * 64];
              if constexpr (NP >= 8) { TC_MFMA(acs, al, b1) }
              TC_MFMA(acs, am, b1) TC_MFMA(acs, ah, b1)
              if constexpr (NP >= 8) { TC_MFMA(acs, am, b2) }
              TC_MFMA(acs, ah, b2)
#undef TC_MFMA
            }
          }
          if constexpr (TR) ++tr_steps;
          if (kn[0] < 0) break;
          if constexpr (TR) tr_mark = ls3d_cycles();
          LS3D_WAIT_VMCNT(0);
          __syncthreads();
          if constexpr (TR) tr_bar += (unsigned)(ls3d_cycles() - tr_mark);
          buf ^= 1;
#pragma unroll
          for (int g = 0; g < G; ++g) { ks[g] = kn[g]; lc[g] = ln[g]; }
        }
        }
#undef TC_NEXT_GROUP
#undef TC_DMA_GROUP
#undef TC_DMA_STEP
#undef TC_LOAD_HALO
      }
    }
    if constexpr (TR) tr_mark = ls3d_cycles();
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] += acs[n][r];
    if (ksplit == 2) {
      float *mine = partial + ((size_t)sidx * 2 + part) * (NT * 16 * TC_THREADS) + tid;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) ls3d_store_agent(mine + (n * 16 + r) * TC_THREADS, acc[n][r]);
      LS3D_WAIT_VMCNT(0);  // the stores are at the coherence point ...
      __syncthreads();     // ... for every thread of the unit, before the counter says so
      // arrival counter of the split tile: never reset - the first unit of a launch finds it even, the second odd (the caller
      // provides the counters zeroed once; every completed launch leaves them even)
      if (tid == 0) s_hid[0] = atomicAdd(counters + sidx, 1) & 1;
      __syncthreads();
      if (s_hid[0] == 0) {  // first of the two: the other unit finishes the tile
        TC_TRACE_WRITE(0)
        if constexpr (CH) continue; else return;
      }
      const float *other = partial + ((size_t)sidx * 2 + (part ^ 1)) * (NT * 16 * TC_THREADS) + tid;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] += ls3d_load_agent(other + (n * 16 + r) * TC_THREADS);  // a + b == b + a: order-independent
    }
    if constexpr (CH) {  // (the host takes only float4-aligned layers without LayerNorm into a chain)
      if (tid == 0) ch_next = atomicAdd(ch.state, 1);
      {
        const TcLayer *L = ch.layers + ch_layer;
        out = tc_uniform(L->out);
        e.scale = tc_uniform(L->e.scale); e.shift = tc_uniform(L->e.shift); e.res_pre = tc_uniform(L->e.res_pre); e.pair = tc_uniform(L->e.pair);
        e.ln_gamma = e.ln_beta = nullptr; e.ln_eps = 0.0f;
        e.res_pre_ld = __builtin_amdgcn_readfirstlane(L->e.res_pre_ld); e.pair_ld = __builtin_amdgcn_readfirstlane(L->e.pair_ld);
        e.relu = __builtin_amdgcn_readfirstlane(L->e.relu);
        cout = __builtin_amdgcn_readfirstlane(L->cout); out_ld = __builtin_amdgcn_readfirstlane(L->out_ld);
      }
      if (ablate & 512) tc_epilogue<NT, false>(acc, (float *)smem, s_rows, wave, kk, col, cout, e, out, out_ld);
      else tc_epilogue<NT, true>(acc, (float *)smem, s_rows, wave, kk, col, cout, e, out, out_ld);
      // the tile's output rows are written through; when every wave's stores have been acknowledged the tile counts as finished
      if (!(ablate & 512)) LS3D_WAIT_VMCNT(0);
      __syncthreads();
      if (tid == 0) {
        ls3d_store_agent_i32(ch.state + TC_CH_DONE + tile, ch_layer + 1);
        atomicAdd(ch.state + TC_CH_LDONE + ch_layer, 1);
      }
    } else {
    if (!e.ln_gamma && !(cout & 3) && !(out_ld & 3) && !(e.res_pre && (e.res_pre_ld & 3)) && !(e.pair && (e.pair_ld & 3)) && !(ablate & 2))
      tc_epilogue<NT>(acc, (float *)smem, s_rows, wave, kk, col, cout, e, out, out_ld);
    else  // LayerNorm epilogue, unaligned leading dimensions, or flags bit 1 (A/B): the general one
      gg_epilogue<NT, 1, TC_TR, 64, TC_THREADS>(acc, (float *)smem, s_rows, s_stat, wave, 0, kk, col, 0, cout, e, out, out_ld);
    }
    TC_TRACE_WRITE(1)
#undef TC_TRACE_WRITE
    if constexpr (!CH) return;
  }
}

template <int NT, int NP, bool TR = false, int LP = 0>
static int tc_launch(hipStream_t stream, const float *in, int in_ld, const TilePlan &p, const uint4 *wpk, int cin, int cout, const EpiDev &e, float *out,
                     int out_ld, int ablate, int swz, int max_units, int split_small, int split_tail, int split_forced, float *partial, int *counters,
                     unsigned *trace) {
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};  // the attribute is per device (multi-GPU servers, multi-device tests)
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_tile_conv<NT, NP, TR, LP>, hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS_BYTES) != hipSuccess) return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_tile_conv<NT, NP, TR, LP>), dim3((unsigned)max_units), dim3(TC_THREADS), TC_LDS_BYTES, stream, in, in_ld, p, wpk, cin, cout,
                     e, out, out_ld, ablate, swz, split_small, split_tail, split_forced, partial, counters, trace, TcChain{nullptr, nullptr, 0});
  return LS3D_OK;
}

template <int NT, int NP, int LP>
static int tc_launch_chain(hipStream_t stream, const TilePlan &p, int swz, int ablate, int grid, float *partial, int *counters, const TcChain &ch) {
  static bool attr_set_on[LS3D_MAX_DEVICES] = {};
  bool &attr_set = attr_set_on[ls3d_device_slot()];
  if (!attr_set) {
    if (hipFuncSetAttribute((const void *)k_tile_conv<NT, NP, false, LP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TC_LDS_BYTES) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    attr_set = true;
  }
  const EpiDev e0 = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.0f};
  hipLaunchKernelGGL((k_tile_conv<NT, NP, false, LP, true>), dim3((unsigned)grid), dim3(TC_THREADS), TC_LDS_BYTES, stream, (const float *)nullptr, 0, p,
                     (const uint4 *)nullptr, 16, 32, e0, (float *)nullptr, 0, ablate, swz, 0, 0, -1, partial, counters, (unsigned *)nullptr, ch);
  return LS3D_OK;
}

// the split over the input channels: two partial accumulator sets per split tile in the per-call workspace, one arrival counter per
// split tile in a caller-owned, zeroed-once array of TC_SPLIT_MAX ints
constexpr int TC_SPLIT_MAX = 512;       // = the chip's workgroup slots for this kernel (2 per CU x 256 CUs)
static inline size_t tc_partial_bytes(int n_split, int nt) { return tc_align((size_t)n_split * 2 * nt * 16 * TC_THREADS * sizeof(float)); }

extern "C" size_t ls3d_tile_conv_workspace_bytes(int n_rows, int cout) {
  if (n_rows <= 0 || cout < 1) return 0;
  if (cout > 128) cout = 128;  // slabs of 128 columns, one after the other through the same workspace
  const int nt = cout <= 32 ? 1 : cout <= 64 ? 2 : 4;
  const int t = (n_rows + TC_TR - 1) / TC_TR, ns = t < TC_SPLIT_MAX ? t : TC_SPLIT_MAX;
  return tc_partial_bytes(ns, nt);
}

extern "C" size_t ls3d_tile_conv_counter_bytes(void) { return (size_t)TC_SPLIT_MAX * sizeof(int32_t); }

// trace records of one launch (flags bit 5): [units][4 waves][TC_TRACE_WORDS] uint32 behind the partial sums in the workspace
static inline size_t tc_trace_bytes(int ntiles) {
  return (size_t)(ntiles + (ntiles < TC_SPLIT_MAX ? ntiles : TC_SPLIT_MAX)) * 4 * TC_TRACE_WORDS * sizeof(unsigned);
}
extern "C" size_t ls3d_tile_conv_trace_bytes(int n_rows) { return n_rows <= 0 ? 0 : tc_trace_bytes((n_rows + TC_TR - 1) / TC_TR); }

static int tc_conv_slab(const float *in, int in_ld, const void *plan, int n_rows, int kvol, const void *w_packed, int cin, int cout, int products,
                        const ls3d_epilogue_t *epi, float *out, int out_ld, void *workspace, size_t workspace_bytes, int32_t *counters, int flags,
                        ls3d_stream_t stream_);

extern "C" int ls3d_tile_conv(const float *in, int in_ld, const void *plan, int n_rows, int kvol, const void *w_packed, int cin, int cout, int products,
                              const ls3d_epilogue_t *epi, float *out, int out_ld, void *workspace, size_t workspace_bytes, int32_t *counters, int flags,
                              ls3d_stream_t stream_) {
  if (cout <= 128)
    return tc_conv_slab(in, in_ld, plan, n_rows, kvol, w_packed, cin, cout, products, epi, out, out_ld, workspace, workspace_bytes, counters, flags, stream_);
  // more than 128 output columns: one launch per slab of 128 on the same plan, workspace and counters (the launches are ordered on the
  // stream); the epilogue operands move with the columns.  A row LayerNorm needs the whole row in one workgroup: not available here.
  if (!w_packed || !out || cin < 16 || (cin % 16) || kvol < 1 || out_ld < cout) return LS3D_ERR_ARG;
  if (epi && epi->ln_gamma) return LS3D_ERR_UNSUPPORTED;
  const char *wp = (const char *)w_packed;
  for (int col0 = 0; col0 < cout; col0 += 128) {
    const int cw = cout - col0 < 128 ? cout - col0 : 128;
    ls3d_epilogue_t e2;
    if (epi) {
      e2 = *epi;
      if (e2.scale) e2.scale += col0;
      if (e2.shift) e2.shift += col0;
      if (e2.res_pre) e2.res_pre += col0;
      if (e2.pair) e2.pair += 2 * col0;
    }
    const int rc = tc_conv_slab(in, in_ld, plan, n_rows, kvol, wp, cin, cw, products, epi ? &e2 : nullptr, out + col0, out_ld, workspace, workspace_bytes,
                                counters, flags, stream_);
    if (rc != LS3D_OK) return rc;
    wp += tc_slab_packed_bytes(kvol, cin, cw, products == 1 ? 1 : 3);
  }
  return LS3D_OK;
}

static int tc_conv_slab(const float *in, int in_ld, const void *plan, int n_rows, int kvol, const void *w_packed, int cin, int cout, int products,
                        const ls3d_epilogue_t *epi, float *out, int out_ld, void *workspace, size_t workspace_bytes, int32_t *counters, int flags,
                        ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n_rows == 0 && w_packed && kvol >= 1 && kvol <= TC_KMAX && cin >= 16 && !(cin % 16) && cout >= 1 && cout <= 128 && (products == 1 || products == 6 || products == 8))
    return LS3D_OK;
  if (!in || !plan || !w_packed || !out || n_rows < 0 || kvol < 1 || cin < 16 || cout < 1) return LS3D_ERR_ARG;
  if ((cin % 16) || (in_ld % 4) || in_ld < cin || out_ld < cout) return LS3D_ERR_ARG;
  if (((uintptr_t)in & 15) || ((uintptr_t)w_packed & 15) || ((uintptr_t)plan & 15) || ((uintptr_t)workspace & 15)) return LS3D_ERR_ARG;
  if (kvol > TC_KMAX || cout > 128) return LS3D_ERR_UNSUPPORTED;
  if (products != 1 && products != 6 && products != 8) return LS3D_ERR_ARG;
  if (n_rows == 0) return LS3D_OK;
  EpiDev e = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0.0f};
  if (epi) {
    e.scale = epi->scale; e.shift = epi->shift; e.res_pre = epi->res_pre; e.pair = epi->pair;
    e.res_pre_ld = epi->res_pre_ld; e.pair_ld = epi->pair_ld; e.relu = epi->relu;
    e.ln_gamma = epi->ln_gamma; e.ln_beta = epi->ln_beta; e.ln_eps = epi->ln_eps;
    if ((e.ln_gamma != nullptr) != (e.ln_beta != nullptr) || (e.ln_gamma && e.pair)) return LS3D_ERR_ARG;
  }
  const TilePlan p = tc_plan(const_cast<void *>(plan), n_rows, kvol);
  const int nt = cout <= 32 ? 1 : cout <= 64 ? 2 : 4;  // column blocks of the kernel variant (the weights are packed for it)
  const int ablate = flags & 30, split_mode = (flags >> 6) & 3, forced = (flags >> 8) & 0xFFF, tail = (flags >> 20) & 0x3FF;
  // Split over the input channels (two work units per tile, each over half of the 16-channel chunks) when the caller provides the
  // workspace and the counters and the layer has >= 4 chunks: every tile of a launch whose LIVE tiles do not fill the chip's
  // TC_SPLIT_MAX workgroup slots (level 4 of the 120k frame: 160 -> 125 us per layer).  Larger launches: the `tail` tiles at the
  // end of the dispatch order when flags ask for it (measured on the 677- and 1071-tile levels: no gain, so off by default).  The
  // kernel applies the rule to the plan's live tile count; the host only sizes the grid for the worst case.
  int split_small = 0, split_tail = 0, split_forced = -1, ns_grid = 0;
  if (workspace && counters && cin >= 64 && split_mode != 1) {
    split_small = TC_SPLIT_MAX;                          // every tile when the live tiles do not fill the workgroup slots
    split_tail = tail ? tail - 1 : 0;                    // otherwise this many at the end of the dispatch order
    if (forced) split_forced = forced - 1;               // exactly this many, whatever the tile count
    if (split_mode == 2) split_forced = TC_SPLIT_MAX;
    if (split_mode == 3) split_tail = -1;                // the remainder of the live tiles modulo the workgroup slots
    if (split_tail > TC_SPLIT_MAX) split_tail = TC_SPLIT_MAX;
    if (split_forced > TC_SPLIT_MAX) split_forced = TC_SPLIT_MAX;
    ns_grid = p.ntiles < TC_SPLIT_MAX ? p.ntiles : TC_SPLIT_MAX;
    if (workspace_bytes < tc_partial_bytes(ns_grid, nt)) return LS3D_ERR_WORKSPACE;
  }
  float *partial = (float *)workspace;
  const int swz = (flags >> 30) & 1 ? 0 : 1;
  int rc;
  unsigned *trace = nullptr;
  const bool pipelined = products == 6 && nt >= 2 && !(flags & 1);  // the software-pipelined offset loop (flags bit 0: the plain one)
#define TC_ARGS stream, in, in_ld, p, (const uint4 *)w_packed, cin, cout, e, out, out_ld, ablate, swz, p.ntiles + ns_grid, split_small, split_tail, split_forced, partial, (int *)counters, trace
  if (flags & 32) {  // tracing build (6-product kernels only): the records follow the partial sums of the worst-case split in the workspace
    const size_t off = ls3d_tile_conv_workspace_bytes(n_rows, cout);
    if (products != 6) return LS3D_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < off + tc_trace_bytes(p.ntiles)) return LS3D_ERR_WORKSPACE;
    trace = (unsigned *)((char *)workspace + off);
    if (hipMemsetAsync(trace, 0, tc_trace_bytes(p.ntiles), stream) != hipSuccess) return LS3D_ERR_LAUNCH;
    rc = nt == 1 ? tc_launch<1, 6, true>(TC_ARGS)
       : nt == 2 ? (pipelined ? tc_launch<2, 6, true, 1>(TC_ARGS) : tc_launch<2, 6, true>(TC_ARGS))
                 : (pipelined ? tc_launch<4, 6, true, 1>(TC_ARGS) : tc_launch<4, 6, true>(TC_ARGS));
    if (rc != LS3D_OK) return rc;
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  rc = nt == 1 ? (products == 8 ? tc_launch<1, 8>(TC_ARGS) : products == 6 ? tc_launch<1, 6>(TC_ARGS) : tc_launch<1, 1>(TC_ARGS))
     : nt == 2 ? (products == 8 ? tc_launch<2, 8>(TC_ARGS) : products == 6 ? (pipelined ? tc_launch<2, 6, false, 1>(TC_ARGS) : tc_launch<2, 6>(TC_ARGS)) : tc_launch<2, 1>(TC_ARGS))
               : (products == 8 ? tc_launch<4, 8>(TC_ARGS) : products == 6 ? (pipelined ? tc_launch<4, 6, false, 1>(TC_ARGS) : tc_launch<4, 6>(TC_ARGS)) : tc_launch<4, 1>(TC_ARGS));
#undef TC_ARGS
  if (rc != LS3D_OK) return rc;
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// chained launch
// ---------------------------------------------------------------------------------------------------------------
extern "C" size_t ls3d_tile_chain_state_bytes(int n_rows) {
  if (n_rows < 0) return 0;
  const int t = (n_rows + TC_TR - 1) / TC_TR;
  return tc_chain_layers_off(t) + tc_align(sizeof(TcLayer) * TC_CHAIN_MAX);
}

static int tc_workgroup_slots() {  // persistent workgroups of a chained launch: two per CU (LDS 77 KB each)
  static int slots_on[LS3D_MAX_DEVICES] = {};
  int &slots = slots_on[ls3d_device_slot()];
  if (slots == 0) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slots = 2 * cus;
  }
  return slots;
}

extern "C" int ls3d_tile_conv_chain(const void *plan, int n_rows, int kvol, const ls3d_tile_chain_layer_t *layers, int n_layers, int products,
                                    void *state, size_t state_bytes, void *workspace, size_t workspace_bytes, int32_t *counters, int flags,
                                    ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!layers || n_layers < 1 || kvol < 1) return LS3D_ERR_ARG;
  if (n_layers > TC_CHAIN_MAX || kvol > TC_KMAX || products != 6) return LS3D_ERR_UNSUPPORTED;
  if (n_rows == 0) return LS3D_OK;
  if (!plan || !state || n_rows < 0 || ((uintptr_t)plan & 15) || ((uintptr_t)state & 15) || ((uintptr_t)workspace & 15)) return LS3D_ERR_ARG;
  if (state_bytes < ls3d_tile_chain_state_bytes(n_rows)) return LS3D_ERR_WORKSPACE;
  const TilePlan p = tc_plan(const_cast<void *>(plan), n_rows, kvol);
  TcChainSetup su;
  int nt = 0;
  bool any_split = false;
  size_t need_ws = 0;
  for (int l = 0; l < n_layers; ++l) {
    const ls3d_tile_chain_layer_t &a = layers[l];
    if (!a.in || !a.w_packed || !a.out || a.cin < 16 || a.cout < 1) return LS3D_ERR_ARG;
    if ((a.cin % 16) || (a.in_ld % 4) || a.in_ld < a.cin || a.out_ld < a.cout) return LS3D_ERR_ARG;
    if (((uintptr_t)a.in & 15) || ((uintptr_t)a.w_packed & 15) || ((uintptr_t)a.out & 15)) return LS3D_ERR_ARG;
    if (a.cout > 128) return LS3D_ERR_UNSUPPORTED;
    // one 128-byte line = one row = one writer (the coherence argument at k_tile_conv): rows of at least 32 floats, 128-byte aligned
    if (a.cout < 32 || (a.out_ld % 32) || ((uintptr_t)a.out & 127)) return LS3D_ERR_UNSUPPORTED;
    const int nt_l = a.cout <= 32 ? 1 : a.cout <= 64 ? 2 : 4;
    if (nt && nt_l != nt) return LS3D_ERR_UNSUPPORTED;  // one kernel variant (column blocks) per chain
    nt = nt_l;
    // the chain runs the single-pass float4 epilogue on coherent 16-byte accesses with 32-bit byte offsets
    const ls3d_epilogue_t &ep = a.epi;
    if (ep.ln_gamma || ep.ln_beta) return LS3D_ERR_UNSUPPORTED;
    if ((a.cout & 3) || (a.out_ld & 3) || (ep.res_pre && ((ep.res_pre_ld & 3) || ((uintptr_t)ep.res_pre & 15))) ||
        (ep.pair && ((ep.pair_ld & 3) || ((uintptr_t)ep.pair & 15))))
      return LS3D_ERR_UNSUPPORTED;
    const unsigned long long lim = 0xFFFFFFFFull;
    if ((unsigned long long)n_rows * a.in_ld * 4 > lim || (unsigned long long)n_rows * a.out_ld * 4 > lim ||
        (ep.res_pre && (unsigned long long)n_rows * ep.res_pre_ld * 4 > lim) || (ep.pair && (unsigned long long)n_rows * ep.pair_ld * 4 > lim))
      return LS3D_ERR_UNSUPPORTED;
    TcLayer &L = su.l[l];
    L.in = a.in; L.wpk = (const uint4 *)a.w_packed; L.out = a.out;
    L.e = EpiDev{ep.scale, ep.shift, ep.res_pre, ep.pair, nullptr, nullptr, ep.res_pre_ld, ep.pair_ld, ep.relu, 0.0f};
    L.in_ld = a.in_ld; L.cin = a.cin; L.cout = a.cout; L.out_ld = a.out_ld; L.pad_ = 0;
    // the split over the input channels, layer by layer as ls3d_tile_conv decides it (flags bits 6-7 == 1: never)
    L.n_split = (workspace && counters && a.cin >= 64 && ((flags >> 6) & 3) != 1) ? 1 : 0;
    if (L.n_split) {
      any_split = true;
      const int ns = p.ntiles < TC_SPLIT_MAX ? p.ntiles : TC_SPLIT_MAX;
      if (tc_partial_bytes(ns, nt_l) > need_ws) need_ws = tc_partial_bytes(ns, nt_l);
    }
  }
  for (int l = n_layers; l < TC_CHAIN_MAX; ++l) su.l[l] = su.l[0];
  if (any_split && workspace_bytes < need_ws) return LS3D_ERR_WORKSPACE;
  int *st = (int *)state;
  TcLayer *dst = (TcLayer *)((char *)state + tc_chain_layers_off(p.ntiles));
  hipLaunchKernelGGL(k_tile_chain_setup, dim3(1), dim3(256), 0, stream, p, su, n_layers, TC_SPLIT_MAX, dst, st);
  const long long worst = (long long)n_layers * p.ntiles * 2;
  const int slots = tc_workgroup_slots();
  const int grid = (int)(worst < slots ? worst : slots);
  const int swz = (flags >> 30) & 1 ? 0 : 1;
  const TcChain ch = {dst, st, n_layers};
  // timing experiments, flags bits 1-3 (results are WRONG with bits 1 / 2): bit 1 no producer waits, bit 2 cached stores, bit 3 cost order;
  // bit 4 (16): no halo loads, bits 8 in ablate: no weight DMA (the kernel's own ablation bits)
  const int ablate = ((flags & 14) << 7) | (flags & 16) | ((flags & 32) ? 8 : 0);  // (round 5 also timed the loop without its DMA wait - bit 0, a branch inside the step: removed after the experiment)
  int rc = nt == 1 ? tc_launch_chain<1, 6, 0>(stream, p, swz, ablate, grid, (float *)workspace, (int *)counters, ch)
         : nt == 2 ? tc_launch_chain<2, 6, 1>(stream, p, swz, ablate, grid, (float *)workspace, (int *)counters, ch)
                   : tc_launch_chain<4, 6, 1>(stream, p, swz, ablate, grid, (float *)workspace, (int *)counters, ch);
  if (rc != LS3D_OK) return rc;
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
