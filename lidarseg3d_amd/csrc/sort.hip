// sort.hip — stable LSD radix sort of (uint32 key, int32 value) pairs, 8 bits per pass, for the row orders of the sparse
// convolutions: spatial keys of the tile plans (ls3d_tile_keys, <= 20 bits: 2-3 passes) and neighbour-mask keys of the
// gather-GEMM (ls3d_rulebook_sort_keys, 31 bits: 4 passes).  Replaces torch.argsort on the frame's geometry stream (round 1:
// ~20 rocPRIM launches per sort + index conversions).
//
// Per pass three launches: block histograms (LDS atomics: counts are order-independent) -> one-workgroup scan of the
// [block][digit] counters in (digit, block) order -> stable scatter (each block walks its chunk in order, 256 elements per round; inside a round the
// rank of an element among equal digits comes from wave ballots over the 8 digit bits plus per-wave counters in LDS).
// Deterministic and stable: equal keys keep their input order, so the plans built on the orders are reproducible.
#include "common.h"

constexpr int RS_CHUNK = 2048;  // elements per block and pass

__global__ __launch_bounds__(256) void k_rs_hist(const uint32_t *__restrict__ keys, int n, int shift, int nb, int32_t *__restrict__ hist) {
  __shared__ int s_h[256];
  const int tid = threadIdx.x, blk = blockIdx.x;
  s_h[tid] = 0;
  __syncthreads();
  const int lo = blk * RS_CHUNK, hi = lo + RS_CHUNK < n ? lo + RS_CHUNK : n;
  for (int i = lo + tid; i < hi; i += 256) atomicAdd(&s_h[(keys[i] >> shift) & 255u], 1);
  __syncthreads();
  hist[blk * 256 + tid] = s_h[tid];  // [block][digit]: coalesced here, in the scan and in the scatter
}

// exclusive scan of the counters in (digit, block) order over the [block][digit] array: one workgroup of 1024 threads = 4 block
// groups x 256 digits; group g owns blocks [g*q, (g+1)*q), every access is a coalesced 1 KB row and the loads are issued 8 deep
// (the loop is latency bound: ~1 us per dependent row with one load in flight)
__global__ __launch_bounds__(1024) void k_rs_scan(int32_t *hist, int nb) {
  __shared__ int s_part[4][256];
  __shared__ int s_tot[256];
  const int d = threadIdx.x & 255, g = threadIdx.x >> 8;
  const int q = (nb + 3) / 4, b0 = g * q, b1 = b0 + q < nb ? b0 + q : nb;
  int sum = 0;
  for (int b = b0; b < b1; b += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b + j < b1 ? hist[(b + j) * 256 + d] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) sum += v[j];
  }
  s_part[g][d] = sum;
  __syncthreads();
  if (g == 0) s_tot[d] = s_part[0][d] + s_part[1][d] + s_part[2][d] + s_part[3][d];
  __syncthreads();
  int base = 0;
  for (int j = 0; j < d; ++j) base += s_tot[j];
  for (int j = 0; j < g; ++j) base += s_part[j][d];
  for (int b = b0; b < b1; b += 8) {
    int v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b + j < b1 ? hist[(b + j) * 256 + d] : 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (b + j < b1) hist[(b + j) * 256 + d] = base;
      base += v[j];
    }
  }
}

__global__ __launch_bounds__(256) void k_rs_scatter(const uint32_t *__restrict__ keys, const int32_t *__restrict__ vals, int n, int shift, int nb,
                                                    const int32_t *__restrict__ hist, uint32_t *__restrict__ keys_out, int32_t *__restrict__ vals_out) {
  __shared__ int s_base[256];
  __shared__ int s_cnt[4][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, blk = blockIdx.x;
  s_base[tid] = hist[blk * 256 + tid];
  const int lo = blk * RS_CHUNK;
  for (int r = 0; r < RS_CHUNK / 256; ++r) {
    const int i = lo + r * 256 + tid;
    const bool live = i < n;
    s_cnt[0][tid] = s_cnt[1][tid] = s_cnt[2][tid] = s_cnt[3][tid] = 0;
    __syncthreads();
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
  const size_t nb = (size_t)(n + RS_CHUNK - 1) / RS_CHUNK;
  return ((size_t)n * 8 + 255) / 256 * 256 * 2 + (nb * 256 * 4 + 255) / 256 * 256;  // key / value ping-pong buffers + histograms
}

// Sort n pairs by the low `bits` bits of the key (ascending, stable).  vals_in == NULL means the identity (the result is the
// sorting permutation).  keys_out / vals_out receive the result (keys_out may be NULL).  In-place is not supported.
int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, int bits, uint32_t *keys_out, int32_t *vals_out, void *workspace,
                          size_t workspace_bytes, hipStream_t stream) {
  if (n == 0 && bits >= 1 && bits <= 32) return LS3D_OK;
  if (!keys_in || !vals_out || !workspace || n < 0 || bits < 1 || bits > 32) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_radix_sort_workspace_bytes(n)) return LS3D_ERR_WORKSPACE;
  if (n == 0) return LS3D_OK;
  const int nb = (n + RS_CHUNK - 1) / RS_CHUNK;
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
    hipLaunchKernelGGL(k_rs_hist, dim3(nb), dim3(256), 0, stream, ksrc, n, 8 * p, nb, hist);
    hipLaunchKernelGGL(k_rs_scan, dim3(1), dim3(1024), 0, stream, hist, nb);
    hipLaunchKernelGGL(k_rs_scatter, dim3(nb), dim3(256), 0, stream, ksrc, vsrc, n, 8 * p, nb, (const int32_t *)hist, kdst, vdst);
    ksrc = kdst;
    vsrc = vdst;
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_radix_sort(const uint32_t *keys, const int32_t *vals, int n, int bits, uint32_t *keys_out, int32_t *vals_out, void *workspace,
                               size_t workspace_bytes, ls3d_stream_t stream) {
  return ls3d_radix_sort_pairs(keys, vals, n, bits, keys_out, vals_out, workspace, workspace_bytes, (hipStream_t)stream);
}
