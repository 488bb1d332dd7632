// sort.hip — stable LSD radix sort of (uint32 key, int32 value) pairs, 8 bits per pass, for the row orders of the sparse
// convolutions: spatial keys of the tile plans (ls3d_tile_keys, <= 20 bits: 2-3 passes) and neighbour-mask keys of the
// gather-GEMM (ls3d_rulebook_sort_keys, 31 bits: 4 passes).  Replaces torch.argsort on the frame's geometry stream (round 1:
// ~20 rocPRIM launches per sort + index conversions).
//
// Round 3 layout (round 2: 2048 elements per block = 32 blocks for a 65k-key level, an 8-round scatter and a one-workgroup scan
// between them, ~190 us per pass; now ~15 us): TWO launches per pass over blocks of EPB = 512 .. elements (at most 512 blocks):
//   k_rs_hist    per-block digit histograms (LDS atomics: counts are order-independent) -> hist[block][digit];
//   k_rs_scatter every block first derives its own 256 output bases from the histogram array - thread d adds up digit d of the
//                blocks before it and the totals of the digits below d (the whole array is <= 512 KB and L2 resident; this replaces
//                the separate scan launch) - then places its elements: 256 per round, the rank of an element among the equal digits
//                of its round from wave ballots over the 8 digit bits plus per-wave counters in LDS.
// Deterministic and stable: equal keys keep their input order, so the plans built on the orders are reproducible.
// Device-side element count (n_dev): only the first min(*n_dev, n) elements are sorted, into the first positions of the outputs;
// blocks beyond the count leave at once, so a sort over a table with spare rows (capacity mode) costs what its live rows cost.
#include "common.h"

constexpr int RS_MAX_BLOCKS = 512;

static inline int rs_epb(int n, int max_blocks = RS_MAX_BLOCKS) {  // elements per block: 512, or more so that a sort never has more than max_blocks blocks
  int epb = 512;
  while ((long long)epb * max_blocks < n) epb *= 2;
  return epb;
}

// Batched form (ls3d_seg_loss: one segment per class): blockIdx.y = segment, `seg` elements / `hseg` histogram ints between segments.
__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t *__restrict__ keys, int n, const int32_t *n_dev, int epb, int shift,
                                                 int32_t *__restrict__ hist, long long seg, int hseg) {
  __shared__ int s_h[256];
  const int tid = threadIdx.x, blk = blockIdx.x;
  keys += (size_t)blockIdx.y * seg;
  hist += (size_t)blockIdx.y * hseg;
  const int N = ls3d_count(n, n_dev);
  const int lo = blk * epb;
  if (lo >= N) return;
  s_h[tid] = 0;
  __syncthreads();
  const int hi = lo + epb < N ? lo + epb : N;
  for (int i = lo + tid; i < hi; i += 256) atomicAdd(&s_h[(keys[i] >> shift) & 255u], 1);
  __syncthreads();
  hist[blk * 256 + tid] = s_h[tid];  // [block][digit]: coalesced here and in the scatter
}

__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int n, const int32_t *n_dev,
                                                    int epb, int shift, const int32_t *__restrict__ hist, uint32_t *__restrict__ keys_out,
                                                    int32_t *__restrict__ vals_out, long long seg, int hseg) {
  __shared__ int s_base[256];
  __shared__ int s_tot[256];
  __shared__ int s_cnt[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  keys += (size_t)blockIdx.y * seg;
  if (vals) vals += (size_t)blockIdx.y * seg;
  keys_out += (size_t)blockIdx.y * seg;
  vals_out += (size_t)blockIdx.y * seg;
  hist += (size_t)blockIdx.y * hseg;
  const int N = ls3d_count(n, n_dev);
  const int lo = blk * epb;
  if (lo >= N) return;
  const int nb = (N + epb - 1) / epb;  // live blocks
  {
    // digit d: elements with that digit in the blocks before this one, and in all blocks.  The loop is latency bound (one L2 round trip per batch:
    // 278 blocks for a 142k-key level were 35 batches of 8 = most of the launch's 16 us): RS_DEEP loads in flight, unconditional (clamped row, masked value)
    constexpr int RS_DEEP = 32;
    int before = 0, total = 0;
    for (int b = 0; b < nb; b += RS_DEEP) {
      int v[RS_DEEP];
#pragma unroll
      for (int j = 0; j < RS_DEEP; ++j) v[j] = hist[(b + j < nb ? b + j : nb - 1) * 256 + tid];
#pragma unroll
      for (int j = 0; j < RS_DEEP; ++j) {
        const int x = b + j < nb ? v[j] : 0;
        total += x;
        if (b + j < blk) before += x;
      }
    }
    s_tot[tid] = total;
    s_base[tid] = before;
    __syncthreads();
    int below = 0;  // exclusive prefix of the digit totals: 256 threads x <= 255 LDS reads, once per block
    for (int j = 0; j < tid; ++j) below += s_tot[j];
    s_base[tid] += below;
  }
  const int hi = lo + epb < N ? lo + epb : N;
  for (int r0 = lo; r0 < hi; r0 += 256) {
    const int i = r0 + tid;
    const bool live = i < hi;
    s_cnt[0][tid] = s_cnt[1][tid] = s_cnt[2][tid] = s_cnt[3][tid] = 0;
    __syncthreads();  // also orders the s_base updates of the previous round / the prologue
    const uint32_t key = live ? keys[i] : 0u;
    const int val = live ? (vals ? vals[i] : i) : 0;
    const unsigned d = (key >> shift) & 255u;
    unsigned long long peers = __ballot(live);  // lanes of this wave with the same digit
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
      keys_out[off] = key;
      vals_out[off] = val;
    }
    __syncthreads();
    s_base[tid] += s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
  }
}

extern "C" size_t ls3d_radix_sort_workspace_bytes(int n) {
  return ((size_t)n * 8 + 255) / 256 * 256 * 2 + (size_t)RS_MAX_BLOCKS * 256 * 4;  // key / value ping-pong buffers + histograms
}

// Sort the first min(*n_dev, n) pairs (all n when n_dev == NULL) by the low `bits` bits of the key (ascending, stable).  vals_in ==
// NULL means the identity (the result is the sorting permutation).  keys_out / vals_out receive the result in their first positions
// (keys_out may be NULL); entries beyond the count are unspecified.  In-place is not supported.
int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                          void *workspace, size_t workspace_bytes, hipStream_t stream) {
  if (n == 0 && bits >= 1 && bits <= 32) return LS3D_OK;
  if (!keys_in || !vals_out || !workspace || n < 0 || bits < 1 || bits > 32) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_radix_sort_workspace_bytes(n)) return LS3D_ERR_WORKSPACE;
  if (n == 0) return LS3D_OK;
  const int epb = rs_epb(n), nb = (n + epb - 1) / epb;
  const size_t half = ((size_t)n * 8 + 255) / 256 * 256;
  char *ws = (char *)workspace;
  uint32_t *kbuf[2] = {(uint32_t *)ws, (uint32_t *)(ws + half)};
  int32_t *vbuf[2] = {(int32_t *)(ws + (size_t)n * 4), (int32_t *)(ws + half + (size_t)n * 4)};
  int32_t *hist = (int32_t *)(ws + 2 * half);
  const int passes = (bits + 7) / 8;
  const uint32_t *ksrc = keys_in;
  const int32_t *vsrc = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    uint32_t *kdst = last && keys_out ? keys_out : kbuf[p & 1];
    int32_t *vdst = last ? vals_out : vbuf[p & 1];
    hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(256), 0, stream, ksrc, n, n_dev, epb, 8 * p, hist, 0ll, 0);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(256), 0, stream, ksrc, vsrc, n, n_dev, epb, 8 * p, (const int32_t *)hist, kdst, vdst, 0ll, 0);
    ksrc = kdst;
    vsrc = vdst;
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// `batch` independent sorts of n pairs each (segment b at element offset b * n in every array), back to back in the same launches:
// keys -> keys_out / vals_out (the sorting permutation of each segment), `tmp_keys` / `tmp_vals` = ping-pong buffers of batch * n
// elements, `hist` = batch * rs_batched_hist_ints(n) ints.  Blocks of >= 4096 elements: every block of the scatter reads its segment's
// whole histogram array, and with hundreds of blocks per segment times dozens of segments that read is what the pass would cost.
constexpr int RS_BATCH_MAX_BLOCKS = 128;
size_t ls3d_rs_batched_hist_ints(int n) { return (size_t)RS_BATCH_MAX_BLOCKS * 256; }
int ls3d_radix_sort_batched(const uint32_t *keys_in, int n, int batch, int bits, uint32_t *keys_out, int32_t *vals_out, uint32_t *tmp_keys,
                            int32_t *tmp_vals, int32_t *hist, hipStream_t stream) {
  if (n <= 0 || batch <= 0) return LS3D_OK;
  int epb = rs_epb(n, RS_BATCH_MAX_BLOCKS);
  if (epb < 4096 && n > 4096) epb = 4096;
  const int nb = (n + epb - 1) / epb, hseg = RS_BATCH_MAX_BLOCKS * 256;
  const int passes = (bits + 7) / 8;
  const uint32_t *ksrc = keys_in;
  const int32_t *vsrc = nullptr;
  for (int p = 0; p < passes; ++p) {
    // ping-pong so that the LAST pass lands in the outputs
    const bool to_out = ((passes - 1 - p) & 1) == 0;
    uint32_t *kdst = to_out ? keys_out : tmp_keys;
    int32_t *vdst = to_out ? vals_out : tmp_vals;
    hipLaunchKernelGGL(k_rs_hist, dim3(nb, batch), dim3(256), 0, stream, ksrc, n, (const int32_t *)nullptr, epb, 8 * p, hist, (long long)n, hseg);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nb, batch), dim3(256), 0, stream, ksrc, vsrc, n, (const int32_t *)nullptr, epb, 8 * p, (const int32_t *)hist, kdst,
                       vdst, (long long)n, hseg);
    ksrc = kdst;
    vsrc = vdst;
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_radix_sort(const uint32_t *keys, const int32_t *vals, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                               void *workspace, size_t workspace_bytes, ls3d_stream_t stream) {
  return ls3d_radix_sort_pairs(keys, vals, n, n_dev, bits, keys_out, vals_out, workspace, workspace_bytes, (hipStream_t)stream);
}

// diagnostics (tools/probe_graph_timeline.py): one lane writes the 100 MHz wall clock when the stream reaches this point - a node of a captured frame
// like any other kernel, so the replay of a hipGraph can be timed stream by stream (rocprofv3 serialises the kernels of the side streams)
__global__ void k_stamp(unsigned long long *dst) { *dst = ls3d_walltime(); }
extern "C" int ls3d_stamp(unsigned long long *dst, ls3d_stream_t stream) {
  if (!dst) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_stamp, dim3(1), dim3(1), 0, (hipStream_t)stream, dst);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
