// voxelize.hip — hard / dynamic voxelization and dynamic scatter for gfx950.
//
// The reference algorithms are serial (numba loop det3d/ops/point_cloud/point_cloud_ops.py:7-55; CUDA
// `determin_voxel_num<<<1,1>>>` det3d/ops/voxel/src/voxelization_cuda.cu:150-180 after an O(N^2) duplicate
// search :106-147).  Here the same result (first-appearance voxel ids, first max_points points per voxel,
// max_voxels overflow rule) comes out of a parallel formulation:
//   1. coordinate -> 64-bit key -> open-addressing hash slot; per slot atomicMin of the point index
//      (= the point that "opens" the voxel in the serial order);
//   2. voxel id = rank of that opening point among all opening points = exclusive prefix sum of flags;
//   3. slot r of voxel v = r-th smallest point index of the voxel: max_points-1 atomicMin rounds, each
//      restricted to indices above the previous round's winner;
//   4. gather.
// All integer work; HBM traffic N*stride*4 (read points) + V*(max_points*C+4)*4 (write) + hash table,
// bound by L2 atomics latency rather than HBM (tables are L2/MALL resident).
#include "common.h"

// ------------------------------------------------------------------------------------------------ scan
__global__ __launch_bounds__(256) void k_scan_local(const int32_t *in, int32_t *out, int n, int32_t *sums) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long base = (long long)blockIdx.x * 1024 + tid * 4;
  int v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (base + j < n) ? in[base + j] : 0;
  const int t = v[0] + v[1] + v[2] + v[3];
  int x = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  if (lane == 63) wsum[wave] = x;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += wsum[w];
  int run = woff + x - t;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (base + j < n) out[base + j] = run;
    run += v[j];
  }
  if (tid == 255) sums[blockIdx.x] = run;
}

__global__ __launch_bounds__(256) void k_scan_sums(int32_t *sums, int nb, int32_t *total_out) {
  __shared__ int wsum[4];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < nb; base += 256) {
    const int i = base + tid;
    const int v = i < nb ? sums[i] : 0;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      int y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wsum[wave] = x;
    __syncthreads();
    int woff = carry;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    if (i < nb) sums[i] = woff + x - v;
    __syncthreads();
    if (tid == 255) carry = woff + x;
    __syncthreads();
  }
  if (tid == 0 && total_out) *total_out = carry;
}

__global__ __launch_bounds__(256) void k_scan_add(int32_t *out, int n, const int32_t *sums) {
  const int add = sums[blockIdx.x];
  const long long base = (long long)blockIdx.x * 1024 + threadIdx.x * 4;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (base + j < n) out[base + j] += add;
}

int ls3d_exclusive_scan_i32(const int32_t *in, int32_t *out, int n, int32_t *tmp, int32_t *total_out,
                            hipStream_t stream) {
  const int nb = (n + 1023) / 1024;
  if (nb == 0) {
    if (total_out) hipMemsetAsync(total_out, 0, sizeof(int32_t), stream);
    return LS3D_OK;
  }
  hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, stream, in, out, n, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, tmp, nb, total_out);
  hipLaunchKernelGGL(k_scan_add, dim3(nb), dim3(256), 0, stream, out, n, (const int32_t *)tmp);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------ coordinates
struct VoxGeom {
  float vs[3], lo[3];
  int grid[3];
  int stride, xyz_col, batch_col, feat_col, n_feat;
};

// (z,y,x) of a point or false.  f32 subtract, f32 divide (correctly rounded), floor — the reference's
// expression `floor((p - lo) / vs)`; no reciprocal, no FMA.
__device__ __forceinline__ bool vox_coord(const float *p, const VoxGeom &g, int &cx, int &cy, int &cz) {
  float c0 = floorf(__fdiv_rn(__fsub_rn(p[0], g.lo[0]), g.vs[0]));
  float c1 = floorf(__fdiv_rn(__fsub_rn(p[1], g.lo[1]), g.vs[1]));
  float c2 = floorf(__fdiv_rn(__fsub_rn(p[2], g.lo[2]), g.vs[2]));
  bool ok = (c0 >= 0.0f) && (c0 < (float)g.grid[0]) && (c1 >= 0.0f) && (c1 < (float)g.grid[1]) && (c2 >= 0.0f) &&
            (c2 < (float)g.grid[2]);
  cx = (int)c0; cy = (int)c1; cz = (int)c2;
  return ok;
}

__global__ __launch_bounds__(256) void k_vox_dynamic(const float *points, int n, VoxGeom g, int32_t *coors) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float *p = points + (size_t)i * g.stride + g.xyz_col;
    int cx, cy, cz;
    bool ok = vox_coord(p, g, cx, cy, cz);
    coors[3 * i + 0] = ok ? cz : -1;
    coors[3 * i + 1] = ok ? cy : -1;
    coors[3 * i + 2] = ok ? cx : -1;
  }
}

// step 1: hash insert + opening point per slot
__global__ __launch_bounds__(256) void k_vox_insert(const float *points, int n, VoxGeom g, uint64_t *keys, uint32_t mask,
                                                   int32_t *first_pt, int32_t *slot_of_pt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float *row = points + (size_t)i * g.stride;
    int cx, cy, cz;
    int slot = -1;
    if (vox_coord(row + g.xyz_col, g, cx, cy, cz)) {
      const int b = g.batch_col >= 0 ? (int)row[g.batch_col] : 0;
      const uint64_t key = ls3d_key(b, cz, cy, cx, g.grid[2], g.grid[1], g.grid[0]);
      slot = ls3d_hash_claim(keys, mask, key);
      atomicMin(&first_pt[slot], i);
    }
    slot_of_pt[i] = slot;
  }
}

// step 2a: flag[i] = point i opens its voxel
__global__ __launch_bounds__(256) void k_vox_flag(int n, const int32_t *slot_of_pt, const int32_t *first_pt, int32_t *flag) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = slot_of_pt[i];
    flag[i] = (s >= 0 && first_pt[s] == i) ? 1 : 0;
  }
}

// step 2b: voxel ids, coordinates, overflow cut-off
__global__ __launch_bounds__(256) void k_vox_assign(int n, const int32_t *flag, const int32_t *vid_excl, const int32_t *slot_of_pt,
                                                   const uint64_t *keys, VoxGeom g, int max_voxels, int max_points,
                                                   int32_t *slot_vid, int32_t *sel, int32_t *coors, int coors_cols,
                                                   int32_t *cutoff, const int32_t *total, int32_t *num_voxels) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!flag[i]) continue;
    const int vid = vid_excl[i];
    if (vid < max_voxels) {
      const int s = slot_of_pt[i];
      slot_vid[s] = vid;
      sel[(size_t)vid * max_points] = i;
      uint64_t k = keys[s];
      const int x = (int)(k % (uint64_t)g.grid[0]); k /= (uint64_t)g.grid[0];
      const int y = (int)(k % (uint64_t)g.grid[1]); k /= (uint64_t)g.grid[1];
      const int z = (int)(k % (uint64_t)g.grid[2]); k /= (uint64_t)g.grid[2];
      int32_t *c = coors + (size_t)vid * coors_cols;
      if (coors_cols == 4) { c[0] = (int)k; c[1] = z; c[2] = y; c[3] = x; }
      else { c[0] = z; c[1] = y; c[2] = x; }
    } else if (vid == max_voxels) {
      *cutoff = i;  // the point at which the serial `break` variant stops scanning
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const int t = *total;
    *num_voxels = t < max_voxels ? t : max_voxels;
  }
}

// step 3: round r picks, per voxel, the smallest point index above round r-1's pick
__global__ __launch_bounds__(256) void k_vox_round(int r, int n, const int32_t *slot_of_pt, const int32_t *slot_vid, int32_t *sel,
                                                  int max_points, const int32_t *cutoff, int use_cutoff) {
  const int cut = use_cutoff ? *cutoff : LS3D_INF_I32;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = slot_of_pt[i];
    if (s < 0 || i >= cut) continue;
    const int v = slot_vid[s];
    if (v < 0) continue;
    int32_t *row = sel + (size_t)v * max_points;
    if (i > row[r - 1]) atomicMin(&row[r], i);
  }
}

// step 4: gather point rows into voxels[V, max_points, n_feat]; zero padding; num_points
__global__ __launch_bounds__(256) void k_vox_gather(const float *points, int n, VoxGeom g, const int32_t *sel, int max_points,
                                                   int cap, const int32_t *num_voxels, float *voxels, int32_t *num_points) {
  const int V = ls3d_count(cap, num_voxels);
  const long long work = (long long)V * max_points;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / max_points), r = (int)(t % max_points);
    const int i = sel[t];
    float *dst = voxels + (size_t)t * g.n_feat;
    if (i < n) {
      const float *src = points + (size_t)i * g.stride + g.feat_col;
      for (int c = 0; c < g.n_feat; ++c) dst[c] = src[c];
    } else {
      for (int c = 0; c < g.n_feat; ++c) dst[c] = 0.0f;
    }
    if (r == 0) {
      int cnt = 0;
      for (int q = 0; q < max_points; ++q) cnt += (sel[(size_t)v * max_points + q] < n) ? 1 : 0;
      num_points[v] = cnt;
    }
  }
}

static VoxGeom make_geom(const ls3d_points_layout_t *lay, const ls3d_grid_t *grid) {
  VoxGeom g;
  for (int a = 0; a < 3; ++a) { g.vs[a] = grid->vs[a]; g.lo[a] = grid->lo[a]; g.grid[a] = grid->grid[a]; }
  g.stride = lay->stride; g.xyz_col = lay->xyz_col; g.batch_col = lay->batch_col;
  g.feat_col = lay->feat_col; g.n_feat = lay->n_feat;
  return g;
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
static inline uint32_t hash_cap_for(long long n) {
  uint32_t c = 1024;
  while ((long long)c < 2 * n) c <<= 1;
  return c;
}

extern "C" const char *ls3d_version(void) { return "ls3d 0.1 gfx950"; }

extern "C" int ls3d_voxelize_dynamic(const float *points, int n, const ls3d_points_layout_t *lay,
                                     const ls3d_grid_t *grid, int32_t *coors, ls3d_stream_t stream) {
  if (!points || !lay || !grid || !coors || n < 0) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_vox_dynamic, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, points, n, make_geom(lay, grid), coors);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

struct VoxWs {
  uint64_t *keys; int32_t *first_pt, *slot_vid, *slot_of_pt, *flag, *vid, *scan_tmp, *sel, *cutoff, *total;
  uint32_t cap; size_t bytes;
};
static VoxWs vox_ws_layout(char *base, int n, int max_points, int max_voxels) {
  VoxWs w;
  w.cap = hash_cap_for(n);
  const long long vmax = n < max_voxels ? n : max_voxels;
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return base + o; };
  w.keys = (uint64_t *)take((size_t)w.cap * 8);
  w.first_pt = (int32_t *)take((size_t)w.cap * 4);
  w.slot_vid = (int32_t *)take((size_t)w.cap * 4);
  w.slot_of_pt = (int32_t *)take((size_t)n * 4);
  w.flag = (int32_t *)take((size_t)n * 4);
  w.vid = (int32_t *)take((size_t)n * 4);
  w.scan_tmp = (int32_t *)take(ls3d_scan_tmp_ints(n) * 4);
  w.sel = (int32_t *)take((size_t)(vmax > 0 ? vmax : 1) * max_points * 4);
  w.cutoff = (int32_t *)take(4);
  w.total = (int32_t *)take(4);
  w.bytes = off;
  return w;
}

extern "C" size_t ls3d_voxelize_hard_workspace_bytes(int n, int max_points, int max_voxels) {
  return vox_ws_layout(nullptr, n > 0 ? n : 1, max_points, max_voxels).bytes;
}

extern "C" int ls3d_voxelize_hard(const float *points, int n, const ls3d_points_layout_t *lay,
                                  const ls3d_grid_t *grid, int max_points, int max_voxels, int overflow_mode,
                                  void *workspace, size_t workspace_bytes, float *voxels, int32_t *coors,
                                  int coors_cols, int32_t *num_points, int32_t *num_voxels_dev, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!points || !lay || !grid || !workspace || !voxels || !coors || !num_points || !num_voxels_dev) return LS3D_ERR_ARG;
  if (n < 0 || max_points < 1 || max_voxels < 1 || (coors_cols != 3 && coors_cols != 4)) return LS3D_ERR_ARG;
  if (lay->batch_col >= 0 && coors_cols != 4) return LS3D_ERR_ARG;
  if (n == 0) { hipMemsetAsync(num_voxels_dev, 0, 4, stream); return LS3D_OK; }
  VoxWs w = vox_ws_layout((char *)workspace, n, max_points, max_voxels);
  if (workspace_bytes < w.bytes) return LS3D_ERR_WORKSPACE;
  const VoxGeom g = make_geom(lay, grid);
  const long long vmax = n < max_voxels ? n : max_voxels;
  hipMemsetAsync(w.keys, 0xFF, (size_t)w.cap * 8, stream);
  hipMemsetAsync(w.first_pt, 0x7F, (size_t)w.cap * 4, stream);
  hipMemsetAsync(w.slot_vid, 0xFF, (size_t)w.cap * 4, stream);
  hipMemsetAsync(w.sel, 0x7F, (size_t)vmax * max_points * 4, stream);
  hipMemsetAsync(w.cutoff, 0x7F, 4, stream);
  const dim3 gp = ls3d_grid(n), blk(256);
  hipLaunchKernelGGL(k_vox_insert, gp, blk, 0, stream, points, n, g, w.keys, w.cap - 1, w.first_pt, w.slot_of_pt);
  hipLaunchKernelGGL(k_vox_flag, gp, blk, 0, stream, n, (const int32_t *)w.slot_of_pt, (const int32_t *)w.first_pt, w.flag);
  int rc = ls3d_exclusive_scan_i32(w.flag, w.vid, n, w.scan_tmp, w.total, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_vox_assign, gp, blk, 0, stream, n, (const int32_t *)w.flag, (const int32_t *)w.vid,
                     (const int32_t *)w.slot_of_pt, (const uint64_t *)w.keys, g, max_voxels, max_points, w.slot_vid, w.sel,
                     coors, coors_cols, w.cutoff, (const int32_t *)w.total, num_voxels_dev);
  for (int r = 1; r < max_points; ++r)
    hipLaunchKernelGGL(k_vox_round, gp, blk, 0, stream, r, n, (const int32_t *)w.slot_of_pt, (const int32_t *)w.slot_vid, w.sel,
                       max_points, (const int32_t *)w.cutoff, overflow_mode == 1 ? 1 : 0);
  hipLaunchKernelGGL(k_vox_gather, ls3d_grid(vmax * max_points), blk, 0, stream, points, n, g, (const int32_t *)w.sel,
                     max_points, (int)vmax, (const int32_t *)num_voxels_dev, voxels, num_points);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}


// ------------------------------------------------------------------------------------------------ ordered segmented mean
// Points-in-voxel pooling without float atomics: rows are sorted by segment id with the stable radix sort (csrc/sort.hip), so the
// rows of a segment sit together in their input order (= the slot order of the reference's padded [V, M, C] tensor,
// scatter_points.py:85-98, scatter_points_cpu.cpp:8-60); one thread per (segment, channel) adds them sequentially in f32.  The
// result is the slot-order sum divided by the count: bit-reproducible and bit-equal to that serial evaluation.
int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t ls3d_radix_sort_workspace_bytes(int n);

struct SegWs { uint32_t *keys, *skeys; int32_t *perm, *start, *end; void *sort_ws; size_t sort_bytes, bytes; };
static SegWs seg_ws_layout(char *base, int n, int n_seg) {
  SegWs w;
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return base + o; };
  w.keys = (uint32_t *)take((size_t)n * 4);
  w.skeys = (uint32_t *)take((size_t)n * 4);
  w.perm = (int32_t *)take((size_t)n * 4);
  w.start = (int32_t *)take((size_t)(n_seg + 1) * 4);
  w.end = (int32_t *)take((size_t)(n_seg + 1) * 4);
  w.sort_bytes = ls3d_radix_sort_workspace_bytes(n);
  w.sort_ws = take(w.sort_bytes);
  w.bytes = off;
  return w;
}

__global__ __launch_bounds__(256) void k_seg_bounds(const uint32_t *skeys, int n, int n_seg, int32_t *start, int32_t *end) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t k = skeys[i];
    if (k >= (uint32_t)n_seg) continue;
    if (i == 0 || skeys[i - 1] != k) start[k] = i;
    if (i == n - 1 || skeys[i + 1] != k) end[k] = i + 1;
  }
}

__global__ __launch_bounds__(256) void k_seg_mean_ordered(const float *src, int C, const int32_t *perm, const int32_t *start, const int32_t *end,
                                                         int n_seg, const int32_t *n_seg_dev, float *out, int32_t *counts, int divide) {
  const int S = ls3d_count(n_seg, n_seg_dev);
  const long long work = (long long)S * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int sg = (int)(t / C), c = (int)(t % C);
    const int a = start[sg], b = end[sg];
    float acc = 0.0f;
    for (int p = a; p < b; ++p) acc = __fadd_rn(acc, src[(size_t)perm[p] * C + c]);
    out[t] = b > a ? (divide ? __fdiv_rn(acc, (float)(b - a)) : acc) : 0.0f;  // divide == 0: the ordered SUM (ls3d_dynamic_scatter mode 2)
    if (c == 0 && counts) counts[sg] = b - a;
  }
}

// ids[n] (uint32, >= n_seg = not in any segment) -> out[n_seg, C] = ordered mean; counts (optional) [n_seg]
static int seg_mean_ordered(const float *src, int n, int C, int n_seg, const int32_t *n_seg_dev, SegWs &w, float *out, int32_t *counts, hipStream_t stream,
                            int divide = 1) {
  int bits = 1;
  while ((1u << bits) <= (unsigned)n_seg && bits < 31) ++bits;
  int rc = ls3d_radix_sort_pairs(w.keys, nullptr, n, nullptr, bits, w.skeys, w.perm, w.sort_ws, w.sort_bytes, stream);
  if (rc != LS3D_OK) return rc;
  hipMemsetAsync(w.start, 0, (size_t)(n_seg + 1) * 4, stream);
  hipMemsetAsync(w.end, 0, (size_t)(n_seg + 1) * 4, stream);
  hipLaunchKernelGGL(k_seg_bounds, ls3d_grid(n), dim3(256), 0, stream, (const uint32_t *)w.skeys, n, n_seg, w.start, w.end);
  hipLaunchKernelGGL(k_seg_mean_ordered, ls3d_grid((long long)n_seg * C), dim3(256), 0, stream, src, C, (const int32_t *)w.perm,
                     (const int32_t *)w.start, (const int32_t *)w.end, n_seg, n_seg_dev, out, counts, divide);
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------ dynamic scatter
// DynamicScatter: all points of a voxel are reduced (no cap).  Mean: the ordered segmented mean above (slot-order f32 sum, no
// atomics, bit-reproducible); max uses an order-preserving integer atomicMax (exact, order-independent).
__device__ __forceinline__ int f2ord(float f) { int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7FFFFFFF; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7FFFFFFF); }

__global__ __launch_bounds__(256) void k_ds_insert(const int32_t *coors, int n, int cols, int Z, int Y, int X, uint64_t *keys,
                                                  uint32_t mask, int32_t *first_pt, int32_t *slot_of_pt) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int32_t *c = coors + (size_t)i * cols;
    int slot = -1;
    const int o = cols - 3;
    if (c[o] >= 0 && c[o + 1] >= 0 && c[o + 2] >= 0) {
      const int b = cols == 4 ? c[0] : 0;
      slot = ls3d_hash_claim(keys, mask, ls3d_key(b, c[o], c[o + 1], c[o + 2], Z, Y, X));
      atomicMin(&first_pt[slot], i);
    }
    slot_of_pt[i] = slot;
  }
}

__global__ __launch_bounds__(256) void k_ds_assign(int n, const int32_t *flag, const int32_t *vid_excl, const int32_t *slot_of_pt,
                                                  const int32_t *coors, int cols, int32_t *slot_vid, int32_t *voxel_coors,
                                                  const int32_t *total, int32_t *num_voxels) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!flag[i]) continue;
    const int vid = vid_excl[i];
    slot_vid[slot_of_pt[i]] = vid;
    for (int c = 0; c < cols; ++c) voxel_coors[(size_t)vid * cols + c] = coors[(size_t)i * cols + c];
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *num_voxels = *total;
}

__global__ __launch_bounds__(256) void k_ds_reduce(const float *feats, int n, int C, const int32_t *slot_of_pt, const int32_t *slot_vid,
                                                  int mode, float *out, int32_t *counts, int32_t *point2voxel, uint32_t *keys) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), c = (int)(t % C);
    const int s = slot_of_pt[i];
    const int v = s >= 0 ? slot_vid[s] : -1;
    if (c == 0 && point2voxel) point2voxel[i] = v;
    if (c == 0 && keys) keys[i] = v >= 0 ? (uint32_t)v : 0x7FFFFFFFu;
    if (v < 0 || mode == 0) continue;  // mode 0 (mean) only needs the segment keys here
    const float x = feats[t];
    atomicMax((int *)&out[(size_t)v * C + c], f2ord(x));
    if (c == 0) atomicAdd(&counts[v], 1);
  }
}

__global__ __launch_bounds__(256) void k_ds_maxcount(int cap, const int32_t *num_voxels, const int32_t *counts, int32_t *maxcnt) {
  const int V = ls3d_count(cap, num_voxels);
  int m = 0;
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) m = max(m, counts[v]);
  if (m > 0) atomicMax(maxcnt, m);
}

// mean: divide by the count.  max: the reference reduces the zero-padded [V, max_count, C] tensor
// (scatter_points.py:89-91), so voxels with fewer points than the fullest voxel also see the padding zeros.
__global__ __launch_bounds__(256) void k_ds_finish(float *out, int cap, const int32_t *num_voxels, int C, int mode, const int32_t *counts,
                                                  const int32_t *maxcnt) {
  const int V = ls3d_count(cap, num_voxels);
  const int mc = *maxcnt;
  const long long work = (long long)V * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / C);
    if (mode == 0) out[t] = __fdiv_rn(out[t], (float)counts[v]);
    else {
      float m = ord2f(((int *)out)[t]);
      out[t] = counts[v] < mc ? fmaxf(m, 0.0f) : m;
    }
  }
}

__global__ __launch_bounds__(256) void k_fill_i32(int32_t *p, long long n, int32_t v) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) p[t] = v;
}

struct DsWs { uint64_t *keys; int32_t *first_pt, *slot_vid, *slot_of_pt, *flag, *vid, *scan_tmp, *counts, *total, *maxcnt; uint32_t cap; size_t bytes; SegWs seg; };
static DsWs ds_ws_layout(char *base, int n) {
  DsWs w; w.cap = hash_cap_for(n);
  size_t off = 0;
  auto take = [&](size_t b) { size_t o = off; off += align256(b); return base + o; };
  w.keys = (uint64_t *)take((size_t)w.cap * 8);
  w.first_pt = (int32_t *)take((size_t)w.cap * 4);
  w.slot_vid = (int32_t *)take((size_t)w.cap * 4);
  w.slot_of_pt = (int32_t *)take((size_t)n * 4);
  w.flag = (int32_t *)take((size_t)n * 4);
  w.vid = (int32_t *)take((size_t)n * 4);
  w.scan_tmp = (int32_t *)take(ls3d_scan_tmp_ints(n) * 4);
  w.counts = (int32_t *)take((size_t)n * 4);
  w.total = (int32_t *)take(4);
  w.maxcnt = (int32_t *)take(4);
  w.seg = seg_ws_layout(base ? base + off : nullptr, n, n);
  off += w.seg.bytes;
  w.bytes = off;
  return w;
}
extern "C" size_t ls3d_dynamic_scatter_workspace_bytes(int n) { return ds_ws_layout(nullptr, n > 0 ? n : 1).bytes; }

extern "C" int ls3d_dynamic_scatter(const float *feats_in, int n, int n_feat, const int32_t *coors, int coors_cols,
                                    const int32_t shape_zyx[3], int mode, void *workspace, size_t workspace_bytes,
                                    float *feats_out, int32_t *voxel_coors, int32_t *point2voxel,
                                    int32_t *num_voxels_dev, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!feats_in || !coors || !shape_zyx || !workspace || !feats_out || !voxel_coors || !num_voxels_dev) return LS3D_ERR_ARG;
  if (n < 0 || n_feat < 1 || (coors_cols != 3 && coors_cols != 4) || (mode != 0 && mode != 1 && mode != 2)) return LS3D_ERR_ARG;
  if (n == 0) { hipMemsetAsync(num_voxels_dev, 0, 4, stream); return LS3D_OK; }
  const int divide = mode == 2 ? 0 : 1;  // mode 2: the ordered sum of a voxel's points (DynamicScatterWithDistance's weighted average)
  if (mode == 2) mode = 0;
  DsWs w = ds_ws_layout((char *)workspace, n);
  if (workspace_bytes < w.bytes) return LS3D_ERR_WORKSPACE;
  hipMemsetAsync(w.keys, 0xFF, (size_t)w.cap * 8, stream);
  hipMemsetAsync(w.first_pt, 0x7F, (size_t)w.cap * 4, stream);
  hipMemsetAsync(w.slot_vid, 0xFF, (size_t)w.cap * 4, stream);
  hipMemsetAsync(w.counts, 0, (size_t)n * 4, stream);
  hipMemsetAsync(w.maxcnt, 0, 4, stream);
  const dim3 gp = ls3d_grid(n), blk(256);
  if (mode == 0) hipMemsetAsync(feats_out, 0, (size_t)n * n_feat * 4, stream);
  else hipLaunchKernelGGL(k_fill_i32, ls3d_grid((long long)n * n_feat), blk, 0, stream, (int32_t *)feats_out, (long long)n * n_feat, (int32_t)0x80000000);
  hipLaunchKernelGGL(k_ds_insert, gp, blk, 0, stream, coors, n, coors_cols, shape_zyx[0], shape_zyx[1], shape_zyx[2], w.keys,
                     w.cap - 1, w.first_pt, w.slot_of_pt);
  hipLaunchKernelGGL(k_vox_flag, gp, blk, 0, stream, n, (const int32_t *)w.slot_of_pt, (const int32_t *)w.first_pt, w.flag);
  int rc = ls3d_exclusive_scan_i32(w.flag, w.vid, n, w.scan_tmp, w.total, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_ds_assign, gp, blk, 0, stream, n, (const int32_t *)w.flag, (const int32_t *)w.vid, (const int32_t *)w.slot_of_pt,
                     coors, coors_cols, w.slot_vid, voxel_coors, (const int32_t *)w.total, num_voxels_dev);
  hipLaunchKernelGGL(k_ds_reduce, ls3d_grid((long long)n * n_feat), blk, 0, stream, feats_in, n, n_feat, (const int32_t *)w.slot_of_pt,
                     (const int32_t *)w.slot_vid, mode, feats_out, w.counts, point2voxel, mode == 0 ? w.seg.keys : nullptr);
  if (mode == 0) {  // ordered mean: rows beyond *num_voxels_dev stay zero (the memset above)
    rc = seg_mean_ordered(feats_in, n, n_feat, n, num_voxels_dev, w.seg, feats_out, nullptr, stream, divide);
    if (rc != LS3D_OK) return rc;
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  hipLaunchKernelGGL(k_ds_maxcount, gp, blk, 0, stream, n, (const int32_t *)num_voxels_dev, (const int32_t *)w.counts, w.maxcnt);
  hipLaunchKernelGGL(k_ds_finish, ls3d_grid((long long)n * n_feat), blk, 0, stream, feats_out, n, (const int32_t *)num_voxels_dev, n_feat,
                     mode, (const int32_t *)w.counts, (const int32_t *)w.maxcnt);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------ dynamic scatter, backward
// The reference composes  voxels[V,M,C] = scatter(points)  ->  torch mean / max over M  (scatter_points.py:85-98); its backward is
// torch's reduce backward followed by map_voxel_to_point_kernel (scatter_points_cuda.cu:247-282).  Fused here: mean sends
// grad/count to every point of the voxel, max sends grad to the FIRST point (lowest index = lowest slot) that attains the reduced
// value; if only the zero padding attains it (all points negative, voxel not the fullest) nothing flows.
__global__ __launch_bounds__(256) void k_dsb_prepare(const float *feats, int n, int C, const int32_t *point2voxel, const float *feats_out,
                                                    int mode, int32_t *aux) {
  const long long work = (long long)n * (mode == 0 ? 1 : C);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    if (mode == 0) {
      const int v = point2voxel[t];
      if (v >= 0) atomicAdd(&aux[v], 1);
    } else {
      const int i = (int)(t / C), c = (int)(t % C);
      const int v = point2voxel[i];
      if (v >= 0 && feats[t] == feats_out[(size_t)v * C + c]) atomicMin(&aux[(size_t)v * C + c], i);
    }
  }
}

__global__ __launch_bounds__(256) void k_dsb_apply(const float *grad_voxels, int n, int C, const int32_t *point2voxel, int mode,
                                                  const int32_t *aux, float *grad_points) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), c = (int)(t % C);
    const int v = point2voxel[i];
    float g = 0.0f;
    if (v >= 0) {
      const float gv = grad_voxels[(size_t)v * C + c];
      if (mode == 0) g = __fdiv_rn(gv, (float)aux[v]);
      else if (aux[(size_t)v * C + c] == i) g = gv;
    }
    grad_points[t] = g;
  }
}

extern "C" size_t ls3d_dynamic_scatter_backward_workspace_bytes(int n, int n_feat) {
  return align256((size_t)(n > 0 ? n : 1) * (size_t)(n_feat > 0 ? n_feat : 1) * 4);
}

extern "C" int ls3d_dynamic_scatter_backward(const float *grad_voxels, const int32_t *point2voxel, int n, int n_feat, int mode,
                                             const float *feats_in, const float *feats_out, void *workspace, size_t workspace_bytes,
                                             float *grad_points, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!grad_voxels || !point2voxel || !workspace || !grad_points) return LS3D_ERR_ARG;
  if (n < 0 || n_feat < 1 || (mode != 0 && mode != 1)) return LS3D_ERR_ARG;
  if (mode == 1 && (!feats_in || !feats_out)) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  if (workspace_bytes < ls3d_dynamic_scatter_backward_workspace_bytes(n, n_feat)) return LS3D_ERR_WORKSPACE;
  int32_t *aux = (int32_t *)workspace;
  const dim3 blk(256);
  // at most n voxels: counts[n] (mean) or the winning point per (voxel, channel) [n*C] (max), "none" = INT_MAX
  if (mode == 0) hipMemsetAsync(aux, 0, (size_t)n * 4, stream);
  else hipMemsetAsync(aux, 0x7F, (size_t)n * n_feat * 4, stream);
  hipLaunchKernelGGL(k_dsb_prepare, ls3d_grid((long long)n * (mode == 0 ? 1 : n_feat)), blk, 0, stream, feats_in, n, n_feat, point2voxel,
                     feats_out, mode, aux);
  hipLaunchKernelGGL(k_dsb_apply, ls3d_grid((long long)n * n_feat), blk, 0, stream, grad_voxels, n, n_feat, point2voxel, mode,
                     (const int32_t *)aux, grad_points);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------ dynamic_point_to_voxel_{forward,backward}
// The two remaining functions of the reference's `voxel_layer` module with their own argument shapes (det3d/ops/voxel/src/voxelization.h:63-111,
// bound at voxelization.cpp:6-11; CUDA: scatter_points_cuda.cu:142-282): the padded voxels[V, M, C] tensor, point_to_voxelidx (a point's slot
// inside its voxel), coor_to_voxelidx (its voxel).  The reference sizes voxels from two device counts it copies to the host between its
// kernels (scatter_points_cuda.cu:218-221); a C ABI that never allocates splits at exactly that point: ls3d_dynamic_point_to_voxel_index
// leaves (voxel_num, max_points) on the device, the caller reads them, allocates and calls ls3d_dynamic_point_to_voxel_forward.
// Voxel ids are first-appearance order, slots the point order inside a voxel (what the O(N^2) search + the <<<1,1>>> kernel of the
// reference compute, scatter_points_cuda.cu:72-138); M = the fullest voxel's count as the CPU twin computes it (scatter_points_cpu.cpp:28-31;
// the CUDA kernel leaves M = 0 when no voxel holds two points, :131 - a zero-width tensor - not reproduced).
__global__ __launch_bounds__(256) void k_p2v_keys(int n, const int32_t *slot_of_pt, const int32_t *slot_vid, int32_t *coor_to_voxelidx, uint32_t *keys) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int s = slot_of_pt[i];
    const int v = s >= 0 ? slot_vid[s] : -1;
    coor_to_voxelidx[i] = v;
    keys[i] = v >= 0 ? (uint32_t)v : 0x7FFFFFFFu;
  }
}

__global__ __launch_bounds__(256) void k_p2v_slots(int n, const uint32_t *skeys, const int32_t *perm, const int32_t *start, const int32_t *end,
                                                  int32_t *point_to_voxelidx, int32_t *num_points_per_voxel, int32_t *counts) {
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const uint32_t v = skeys[p];
    if (v >= (uint32_t)n) continue;  // points outside the grid keep slot -1
    const int a = start[v];
    point_to_voxelidx[perm[p]] = p - a;  // the stable sort keeps a voxel's points in point order
    if (p == a) {
      num_points_per_voxel[v] = end[v] - a;
      atomicMax(&counts[1], end[v] - a);
    }
  }
}

__global__ __launch_bounds__(256) void k_p2v_scatter(const float *points, int n, int C, const int32_t *p2v, const int32_t *c2v, int M, float *voxels) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), k = (int)(t % C);
    const int num = p2v[i], v = c2v[i];
    if (num > -1 && v > -1 && num < M) voxels[((size_t)v * M + num) * C + k] = points[t];
  }
}

__global__ __launch_bounds__(256) void k_p2v_gather(float *grad_points, int n, int C, const float *grad_voxels, const int32_t *p2v, const int32_t *c2v, int M) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), k = (int)(t % C);
    const int num = p2v[i];
    if (num > -1 && num < M) grad_points[t] = grad_voxels[((size_t)c2v[i] * M + num) * C + k];  // map_voxel_to_point_kernel: other rows untouched
  }
}

extern "C" size_t ls3d_dynamic_point_to_voxel_workspace_bytes(int n) { return ds_ws_layout(nullptr, n > 0 ? n : 1).bytes; }

extern "C" int ls3d_dynamic_point_to_voxel_index(const int32_t *voxel_mapping, int n, int ndim, const int32_t shape_zyx[3], void *workspace,
                                                 size_t workspace_bytes, int32_t *point_to_voxelidx, int32_t *coor_to_voxelidx,
                                                 int32_t *num_points_per_voxel, int32_t *voxel_coors, int32_t *counts_dev, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!counts_dev || !shape_zyx || n < 0 || (ndim != 3 && ndim != 4)) return LS3D_ERR_ARG;
  hipMemsetAsync(counts_dev, 0, 8, stream);
  if (n == 0) return LS3D_OK;
  if (!voxel_mapping || !workspace || !point_to_voxelidx || !coor_to_voxelidx || !num_points_per_voxel || !voxel_coors) return LS3D_ERR_ARG;
  DsWs w = ds_ws_layout((char *)workspace, n);
  if (workspace_bytes < w.bytes) return LS3D_ERR_WORKSPACE;
  hipMemsetAsync(w.keys, 0xFF, (size_t)w.cap * 8, stream);
  hipMemsetAsync(w.first_pt, 0x7F, (size_t)w.cap * 4, stream);
  hipMemsetAsync(w.slot_vid, 0xFF, (size_t)w.cap * 4, stream);
  hipMemsetAsync(point_to_voxelidx, 0xFF, (size_t)n * 4, stream);
  hipMemsetAsync(num_points_per_voxel, 0, (size_t)n * 4, stream);
  const dim3 gp = ls3d_grid(n), blk(256);
  hipLaunchKernelGGL(k_ds_insert, gp, blk, 0, stream, voxel_mapping, n, ndim, shape_zyx[0], shape_zyx[1], shape_zyx[2], w.keys, w.cap - 1, w.first_pt,
                     w.slot_of_pt);
  hipLaunchKernelGGL(k_vox_flag, gp, blk, 0, stream, n, (const int32_t *)w.slot_of_pt, (const int32_t *)w.first_pt, w.flag);
  int rc = ls3d_exclusive_scan_i32(w.flag, w.vid, n, w.scan_tmp, w.total, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_ds_assign, gp, blk, 0, stream, n, (const int32_t *)w.flag, (const int32_t *)w.vid, (const int32_t *)w.slot_of_pt, voxel_mapping, ndim,
                     w.slot_vid, voxel_coors, (const int32_t *)w.total, counts_dev);
  hipLaunchKernelGGL(k_p2v_keys, gp, blk, 0, stream, n, (const int32_t *)w.slot_of_pt, (const int32_t *)w.slot_vid, coor_to_voxelidx, w.seg.keys);
  int bits = 1;
  while ((1u << bits) <= (unsigned)n && bits < 31) ++bits;
  rc = ls3d_radix_sort_pairs(w.seg.keys, nullptr, n, nullptr, bits, w.seg.skeys, w.seg.perm, w.seg.sort_ws, w.seg.sort_bytes, stream);
  if (rc != LS3D_OK) return rc;
  hipMemsetAsync(w.seg.start, 0, (size_t)(n + 1) * 4, stream);
  hipMemsetAsync(w.seg.end, 0, (size_t)(n + 1) * 4, stream);
  hipLaunchKernelGGL(k_seg_bounds, gp, blk, 0, stream, (const uint32_t *)w.seg.skeys, n, n, w.seg.start, w.seg.end);
  hipLaunchKernelGGL(k_p2v_slots, gp, blk, 0, stream, n, (const uint32_t *)w.seg.skeys, (const int32_t *)w.seg.perm, (const int32_t *)w.seg.start,
                     (const int32_t *)w.seg.end, point_to_voxelidx, num_points_per_voxel, counts_dev);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_dynamic_point_to_voxel_forward(const float *points, int n, int n_feat, const int32_t *point_to_voxelidx, const int32_t *coor_to_voxelidx,
                                                   int voxel_num, int max_points, float *voxels, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n < 0 || n_feat < 1 || voxel_num < 0 || max_points < 0) return LS3D_ERR_ARG;
  const size_t cells = (size_t)voxel_num * max_points * n_feat;
  if (cells == 0 || n == 0) return LS3D_OK;
  if (!points || !point_to_voxelidx || !coor_to_voxelidx || !voxels) return LS3D_ERR_ARG;
  hipMemsetAsync(voxels, 0, cells * 4, stream);
  hipLaunchKernelGGL(k_p2v_scatter, ls3d_grid((long long)n * n_feat), dim3(256), 0, stream, points, n, n_feat, point_to_voxelidx, coor_to_voxelidx, max_points,
                     voxels);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_dynamic_point_to_voxel_backward(float *grad_input_points, const float *grad_output_voxels, const int32_t *point_to_voxelidx,
                                                    const int32_t *coor_to_voxelidx, int n, int n_feat, int max_points, ls3d_stream_t stream_) {
  if (n < 0 || n_feat < 1 || max_points < 0) return LS3D_ERR_ARG;
  if (n == 0 || max_points == 0) return LS3D_OK;
  if (!grad_input_points || !grad_output_voxels || !point_to_voxelidx || !coor_to_voxelidx) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_p2v_gather, ls3d_grid((long long)n * n_feat), dim3(256), 0, (hipStream_t)stream_, grad_input_points, n, n_feat, grad_output_voxels,
                     point_to_voxelidx, coor_to_voxelidx, max_points);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------ segment reductions
// What the dynamic readers (det3d/models/readers/voxel_encoder.py:366-372,451-456,594-600,682-686) ask of torch_scatter
// (third-party, absent from the reference tree: scatter_mean / scatter_max over dim 0 with an int64 segment id per row, e.g.
// torch.unique's inverse).  out[n_seg,C]; segments without rows give 0 (and arg = n), as torch_scatter does.
// mean: the ordered segmented mean (rows added in input order, no atomics: bit-reproducible, unlike torch_scatter's CUDA path);
// max: order-preserving integer atomicMax, arg = the LOWEST row index attaining it (torch_scatter leaves ties to a race).
__global__ __launch_bounds__(256) void k_seg_keys(const int64_t *index, int n, int n_seg, uint32_t *keys, int32_t *bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int64_t sg = index[i];
    const bool ok = sg >= 0 && sg < n_seg;
    if (!ok) *bad = 1;
    keys[i] = ok ? (uint32_t)sg : 0x7FFFFFFFu;
  }
}

__global__ __launch_bounds__(256) void k_seg_accum(const float *src, const int64_t *index, int n, int C, int n_seg, int mode, float *out,
                                                  int32_t *counts, int32_t *bad) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), c = (int)(t % C);
    const int64_t sg = index[i];
    if (sg < 0 || sg >= n_seg) { *bad = 1; continue; }
    if (mode == 0) atomicAdd(&out[(size_t)sg * C + c], src[t]);
    else atomicMax((int *)&out[(size_t)sg * C + c], f2ord(src[t]));
    if (c == 0) atomicAdd(&counts[sg], 1);
  }
}

__global__ __launch_bounds__(256) void k_seg_finish(float *out, int n_seg, int C, int mode, const int32_t *counts) {
  const long long work = (long long)n_seg * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int cnt = counts[t / C];
    if (mode == 0) out[t] = cnt > 0 ? __fdiv_rn(out[t], (float)cnt) : 0.0f;
    else out[t] = cnt > 0 ? ord2f(((int *)out)[t]) : 0.0f;
  }
}

__global__ __launch_bounds__(256) void k_seg_arg(const float *src, const int64_t *index, int n, int C, int n_seg, const float *out,
                                                unsigned long long *arg) {
  const long long work = (long long)n * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / C), c = (int)(t % C);
    const int64_t sg = index[i];
    if (sg < 0 || sg >= n_seg) continue;
    if (src[t] == out[(size_t)sg * C + c]) atomicMin(&arg[(size_t)sg * C + c], (unsigned long long)i);
  }
}

__global__ __launch_bounds__(256) void k_fill_u64(unsigned long long *p, long long n, unsigned long long v) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) p[t] = v;
}

extern "C" size_t ls3d_segment_reduce_workspace_bytes(int n, int n_seg) {
  return align256((size_t)(n_seg > 0 ? n_seg : 1) * 4) + 256 + seg_ws_layout(nullptr, n > 0 ? n : 1, n_seg > 0 ? n_seg : 1).bytes;
}

extern "C" int ls3d_segment_reduce(const float *src, const int64_t *index, int n, int n_feat, int n_seg, int mode, void *workspace,
                                   size_t workspace_bytes, float *out, int64_t *arg_out, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out || !workspace || n < 0 || n_feat < 1 || n_seg < 0 || (mode != 0 && mode != 1)) return LS3D_ERR_ARG;
  if (n > 0 && (!src || !index)) return LS3D_ERR_ARG;
  if (n_seg == 0) return LS3D_OK;
  if (workspace_bytes < ls3d_segment_reduce_workspace_bytes(n, n_seg)) return LS3D_ERR_WORKSPACE;
  int32_t *counts = (int32_t *)workspace;
  int32_t *bad = (int32_t *)((char *)workspace + align256((size_t)n_seg * 4));
  if (mode == 0) {
    hipMemsetAsync(bad, 0, 4, stream);
    if (n == 0) { hipMemsetAsync(out, 0, (size_t)n_seg * n_feat * 4, stream); return LS3D_OK; }
    SegWs sw = seg_ws_layout((char *)workspace + align256((size_t)n_seg * 4) + 256, n, n_seg);
    hipLaunchKernelGGL(k_seg_keys, ls3d_grid(n), dim3(256), 0, stream, index, n, n_seg, sw.keys, bad);
    int rc = seg_mean_ordered(src, n, n_feat, n_seg, nullptr, sw, out, counts, stream);
    if (rc != LS3D_OK) return rc;
    LS3D_RETURN_IF_LAUNCH_FAILED();
    return LS3D_OK;
  }
  const dim3 blk(256);
  const long long cells = (long long)n_seg * n_feat;
  hipMemsetAsync(counts, 0, (size_t)n_seg * 4, stream);
  hipMemsetAsync(bad, 0, 4, stream);
  if (mode == 0) hipMemsetAsync(out, 0, (size_t)cells * 4, stream);
  else hipLaunchKernelGGL(k_fill_i32, ls3d_grid(cells), blk, 0, stream, (int32_t *)out, cells, (int32_t)0x80000000);
  if (n > 0)
    hipLaunchKernelGGL(k_seg_accum, ls3d_grid((long long)n * n_feat), blk, 0, stream, src, index, n, n_feat, n_seg, mode, out, counts, bad);
  hipLaunchKernelGGL(k_seg_finish, ls3d_grid(cells), blk, 0, stream, out, n_seg, n_feat, mode, (const int32_t *)counts);
  if (mode == 1 && arg_out) {
    hipLaunchKernelGGL(k_fill_u64, ls3d_grid(cells), blk, 0, stream, (unsigned long long *)arg_out, cells, (unsigned long long)n);
    if (n > 0)
      hipLaunchKernelGGL(k_seg_arg, ls3d_grid((long long)n * n_feat), blk, 0, stream, src, index, n, n_feat, n_seg, (const float *)out,
                         (unsigned long long *)arg_out);
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
