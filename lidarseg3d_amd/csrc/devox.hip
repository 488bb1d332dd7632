// devox.hip — devoxelization: exact 3-nearest-voxel search + inverse-distance interpolation.
//
// Reference: det3d/models/point_heads/point_utils.py:8-52 (three_interpolate_wrap) over
// det3d/ops/pointnet2_batch/src/interpolate_gpu.cu:16-59 (three_nn_kernel_fast: every thread streams ALL
// known points from global memory) and :84-104 (three_interpolate_kernel_fast).
// Here: the known points of a frame are streamed through LDS in tiles (float4 per point, one broadcast
// ds_read_b128 per candidate per wave), every lane keeps its own top-3 in registers; the fused variant
// (ls3d_devoxelize) then turns (idx, d2) into weights and writes the interpolated feature rows with
// coalesced 128 B stores — no idx / dist / weight round trip through HBM.
// Semantics kept bit-exact: f32 distance fma(dz,dz,fma(dy,dy,dx*dx)) (what nvcc emits for the reference
// expression), strict '<' insertion in ascending index order (lowest index wins ties).
#include "common.h"

#define NN_TILE 1024

struct Top3 {
  float d0, d1, d2;
  int i0, i1, i2;
};

__device__ __forceinline__ void top3_init(Top3 &t) {
  t.d0 = t.d1 = t.d2 = __int_as_float(0x7f800000);  // +inf == (float)1e40
  t.i0 = t.i1 = t.i2 = 0;
}

__device__ __forceinline__ void top3_push(Top3 &t, float d, int k) {
  if (d < t.d2) {
    if (d < t.d0) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = t.d0; t.i1 = t.i0; t.d0 = d; t.i0 = k; }
    else if (d < t.d1) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k; }
    else { t.d2 = d; t.i2 = k; }
  }
}

// scan known[0..m) (xyz at known + k*kstride) for the point (ux,uy,uz); all threads of the block take part
// in staging, `active` threads search.
__device__ __forceinline__ void nn_scan(const float *known, int kstride, int m, bool active, float ux, float uy, float uz, Top3 &t,
                                        float4 *tile) {
  for (int base = 0; base < m; base += NN_TILE) {
    const int cnt = min(NN_TILE, m - base);
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
      const float *p = known + (size_t)(base + j) * kstride;
      tile[j] = make_float4(p[0], p[1], p[2], 0.0f);
    }
    __syncthreads();
    if (active) {
      for (int j = 0; j < cnt; ++j) {
        const float4 q = tile[j];
        const float dx = ux - q.x, dy = uy - q.y, dz = uz - q.z;
        const float d = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
        top3_push(t, d, base + j);
      }
    }
  }
}

__global__ __launch_bounds__(256) void k_three_nn(int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx) {
  __shared__ float4 tile[NN_TILE];
  const int b = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = i < n;
  const float *u = unknown + ((size_t)b * n + (active ? i : 0)) * 3;
  Top3 t;
  top3_init(t);
  nn_scan(known + (size_t)b * m * 3, 3, m, active, u[0], u[1], u[2], t, tile);
  if (active) {
    float *d = dist2 + ((size_t)b * n + i) * 3;
    int32_t *q = idx + ((size_t)b * n + i) * 3;
    d[0] = t.d0; d[1] = t.d1; d[2] = t.d2;
    q[0] = t.i0; q[1] = t.i1; q[2] = t.i2;
  }
}

__global__ __launch_bounds__(256) void k_three_interp_cm(int c, int m, int n, const float *points, const int32_t *idx, const float *weight,
                                                        float *out) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *w = weight + ((size_t)b * n + i) * 3;
  const int32_t *q = idx + ((size_t)b * n + i) * 3;
  const float *f = points + ((size_t)b * c + ch) * m;
  out[((size_t)b * c + ch) * n + i] = fmaf(w[2], f[q[2]], fmaf(w[1], f[q[1]], w[0] * f[q[0]]));
}

__global__ __launch_bounds__(256) void k_three_interp_grad_cm(int c, int n, int m, const float *grad_out, const int32_t *idx,
                                                             const float *weight, float *grad_points) {
  const int b = blockIdx.z, ch = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float *w = weight + ((size_t)b * n + i) * 3;
  const int32_t *q = idx + ((size_t)b * n + i) * 3;
  const float g = grad_out[((size_t)b * c + ch) * n + i];
  float *gp = grad_points + ((size_t)b * c + ch) * m;
  atomicAdd(gp + q[0], g * w[0]);
  atomicAdd(gp + q[1], g * w[1]);
  atomicAdd(gp + q[2], g * w[2]);
}

// (b, (x+.5)*vx+x0, ...) — multiply then add, separately rounded (torch evaluates the two ops unfused)
__global__ __launch_bounds__(256) void k_voxel_centers(const int32_t *coords, int n, const int32_t *n_dev, float vx, float vy, float vz,
                                                      float x0, float y0, float z0, float *out) {
  const int N = ls3d_count(n, n_dev);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    const int32_t *c = coords + 4 * (size_t)v;
    float *o = out + 4 * (size_t)v;
    o[0] = (float)c[0];
    o[1] = __fadd_rn(__fmul_rn(__fadd_rn((float)c[3], 0.5f), vx), x0);
    o[2] = __fadd_rn(__fmul_rn(__fadd_rn((float)c[2], 0.5f), vy), y0);
    o[3] = __fadd_rn(__fmul_rn(__fadd_rn((float)c[1], 0.5f), vz), z0);
  }
}

// fused per-frame devoxelization; grid = (ceil(max_frame_points/256), batch)
__global__ __launch_bounds__(256) void k_devoxelize(const float *points, int pt_stride, const int32_t *pt_off, const float *centers,
                                                   const int32_t *vx_off, const float *feat, int feat_ld, int C, float *out, int out_ld,
                                                   int32_t *idx_out) {
  __shared__ float4 tile[NN_TILE];
  __shared__ int s_idx[256 * 3];
  __shared__ float s_w[256 * 3];
  const int f = blockIdx.y;
  const int p0 = pt_off[f], p1 = pt_off[f + 1];
  const int v0 = vx_off[f], m = vx_off[f + 1] - v0;
  const int first = p0 + blockIdx.x * 256;
  if (first >= p1) return;  // block-uniform
  const int i = first + threadIdx.x;
  const bool active = i < p1;
  const float *u = points + (size_t)(active ? i : p0) * pt_stride + 1;
  Top3 t;
  top3_init(t);
  nn_scan(centers + (size_t)v0 * 4 + 1, 4, m, active, u[0], u[1], u[2], t, tile);
  // w_j = (1/(sqrt(d2_j)+1e-8)) / sum_j(...)   (point_utils.py:30-32)
  const float r0 = __fdiv_rn(1.0f, sqrtf(t.d0) + 1e-8f), r1 = __fdiv_rn(1.0f, sqrtf(t.d1) + 1e-8f),
              r2 = __fdiv_rn(1.0f, sqrtf(t.d2) + 1e-8f);
  const float norm = (r0 + r1) + r2;
  s_idx[threadIdx.x * 3 + 0] = t.i0; s_idx[threadIdx.x * 3 + 1] = t.i1; s_idx[threadIdx.x * 3 + 2] = t.i2;
  s_w[threadIdx.x * 3 + 0] = __fdiv_rn(r0, norm); s_w[threadIdx.x * 3 + 1] = __fdiv_rn(r1, norm); s_w[threadIdx.x * 3 + 2] = __fdiv_rn(r2, norm);
  if (active && idx_out) {
    int32_t *q = idx_out + (size_t)i * 3;
    q[0] = t.i0; q[1] = t.i1; q[2] = t.i2;
  }
  __syncthreads();
  const int cnt = min(256, p1 - first);
  const int c4n = C >> 2;
  for (int e = threadIdx.x; e < cnt * c4n; e += 256) {
    const int p = e / c4n, c4 = e % c4n;
    if (m <= 0) {  // frame without voxels: nothing to interpolate from
      *(float4 *)(out + (size_t)(first + p) * out_ld + c4 * 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      continue;
    }
    const float w0 = s_w[p * 3], w1 = s_w[p * 3 + 1], w2 = s_w[p * 3 + 2];
    const float4 a = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3]) * feat_ld + c4 * 4);
    const float4 b = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3 + 1]) * feat_ld + c4 * 4);
    const float4 c = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3 + 2]) * feat_ld + c4 * 4);
    float4 o;
    o.x = fmaf(w2, c.x, fmaf(w1, b.x, w0 * a.x));
    o.y = fmaf(w2, c.y, fmaf(w1, b.y, w0 * a.y));
    o.z = fmaf(w2, c.z, fmaf(w1, b.z, w0 * a.z));
    o.w = fmaf(w2, c.w, fmaf(w1, b.w, w0 * a.w));
    *(float4 *)(out + (size_t)(first + p) * out_ld + c4 * 4) = o;
  }
}

// ------------------------------------------------------------------------------------------------
// Exact 3-NN through a coarse uniform grid over the voxel centres.
// Voxel centres live on the voxel lattice, so they are binned into coarse cells of CGX x CGY x CGZ fine
// voxels (counting sort: count -> exclusive scan -> fill).  A point visits the coarse cells around its own
// in Chebyshev shells r = 0,1,2,...; after shell r every unvisited centre is at least
//     lb(r) = min_axis( gap_axis + r * cell_size_axis )
// away (gap = distance from the point to the nearer face of its own coarse cell, 0 if it lies outside), so
// the search stops as soon as the 3rd best squared distance is strictly below lb^2 (with a 1e-5 safety
// margin for f32 rounding).  Candidates are compared with the same f32 expression as the brute-force kernel
// and ordered lexicographically by (distance, index), which yields exactly the brute-force result
// (strict '<' in ascending index order == smallest index among equal distances).  Points far outside the
// range simply walk more shells: always exact, no fallback path.
#define CG_RMAX 3   // shells a lane walks for its own point in k_devox_grid; beyond: k_devox_hard

struct CGeom {
  float vs[3], lo[3];
  int grid[3];  // fine cells x,y,z
  int cg[3];    // fine cells per coarse cell x,y,z
  int dim[3];   // coarse cells x,y,z
  int wpf;      // occupancy-bitmap words per frame
};

__device__ __forceinline__ void top3_push_lex(Top3 &t, float d, int k) {
  if (d < t.d2 || (d == t.d2 && k < t.i2)) {
    if (d < t.d0 || (d == t.d0 && k < t.i0)) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = t.d0; t.i1 = t.i0; t.d0 = d; t.i0 = k; }
    else if (d < t.d1 || (d == t.d1 && k < t.i1)) { t.d2 = t.d1; t.i2 = t.i1; t.d1 = d; t.i1 = k; }
    else { t.d2 = d; t.i2 = k; }
  }
}

__global__ __launch_bounds__(256) void k_cg_count(const int32_t *coords, int n, const int32_t *n_dev, CGeom g, int32_t *cell_of, int32_t *cnt,
                                                 uint32_t *occ) {
  const int N = ls3d_count(n, n_dev);
  const int ncf = g.dim[0] * g.dim[1] * g.dim[2];
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    const int32_t *c = coords + 4 * (size_t)v;
    const int cz = min(c[1] / g.cg[2], g.dim[2] - 1), cy = min(c[2] / g.cg[1], g.dim[1] - 1), cx = min(c[3] / g.cg[0], g.dim[0] - 1);
    const int cf = (cz * g.dim[1] + cy) * g.dim[0] + cx;
    cell_of[v] = c[0] * ncf + cf;
    atomicAdd(&cnt[c[0] * ncf + cf], 1);
    atomicOr(&occ[(size_t)c[0] * g.wpf + (cf >> 5)], 1u << (cf & 31));
  }
}

__global__ __launch_bounds__(256) void k_cg_fill(const float *centers, int n, const int32_t *n_dev, const int32_t *vx_off, const int32_t *cell_of,
                                                const int32_t *start, int32_t *cursor, float4 *sorted) {
  const int N = ls3d_count(n, n_dev);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    const int cell = cell_of[v];
    const int pos = start[cell] + atomicAdd(&cursor[cell], 1);
    const float *c = centers + 4 * (size_t)v;
    const int f = (int)c[0];
    sorted[pos] = make_float4(c[1], c[2], c[3], __int_as_float(v - vx_off[f]));
  }
}

// own coarse cell of a point (fine coordinate as in voxelization, clamped into the grid)
__device__ __forceinline__ void cg_point_cell(const CGeom &g, const float *pu, int *cc) {
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float fcoord = floorf(__fdiv_rn(__fsub_rn(pu[a], g.lo[a]), g.vs[a]));
    fcoord = fminf(fmaxf(fcoord, 0.0f), (float)(g.grid[a] - 1));
    cc[a] = min((int)fcoord / g.cg[a], g.dim[a] - 1);
  }
}

// The query points are processed in coarse-cell order (counting sort -> `perm`): the lanes of a wave then walk
// the same cells, so the search neither diverges nor scatters its candidate reads, whatever order the LiDAR
// points arrive in.
// Rows whose batch index lies outside [0, batch) belong to no frame (the padding rows of a point-count bucket, graph.BucketedFrameGraph):
// they are not counted, get no slot in `perm` and - pt_off being the offsets of the frames 0 .. batch - 1 - are never searched.
__global__ __launch_bounds__(256) void k_pt_count(const float *points, int pt_stride, int n, int batch, CGeom g, int32_t *cell_of, int32_t *cnt) {
  const int ncf = g.dim[0] * g.dim[1] * g.dim[2];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float *u = points + (size_t)i * pt_stride;
    const int f = (int)u[0];
    if (f < 0 || f >= batch) { cell_of[i] = -1; continue; }
    int cc[3];
    cg_point_cell(g, u + 1, cc);
    const int cell = f * ncf + (cc[2] * g.dim[1] + cc[1]) * g.dim[0] + cc[0];
    cell_of[i] = cell;
    atomicAdd(&cnt[cell], 1);
  }
}

__global__ __launch_bounds__(256) void k_pt_fill(int n, const int32_t *cell_of, const int32_t *start, int32_t *cursor, int32_t *perm) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int cell = cell_of[i];
    if (cell < 0) continue;
    perm[start[cell] + atomicAdd(&cursor[cell], 1)] = i;
  }
}

// grid = (ceil(max_frame_points/256), batch); dynamic LDS = the frame's coarse-cell occupancy bitmap, so that
// walking through empty space costs LDS bit tests only.
__global__ __launch_bounds__(256) void k_devox_grid(const float *points, int pt_stride, const int32_t *pt_off, const int32_t *perm, CGeom g,
                                                   const int32_t *start, const float4 *sorted, const uint32_t *occ, const int32_t *vx_off,
                                                   const float *feat, int feat_ld, int C, float *out, int out_ld, int32_t *idx_out, float *w_out,
                                                   int32_t *hard_list, int32_t *hard_count) {
  HIP_DYNAMIC_SHARED(uint32_t, s_occ)
  __shared__ int s_idx[256 * 3];
  __shared__ float s_w[256 * 3];
  __shared__ int s_pt[256];
  const int frame = blockIdx.y;
  const int p0 = pt_off[frame], p1 = pt_off[frame + 1];
  const int first = p0 + blockIdx.x * 256;
  if (first >= p1) return;  // block-uniform
  for (int w = threadIdx.x; w < g.wpf; w += 256) s_occ[w] = occ[(size_t)frame * g.wpf + w];
  __syncthreads();
  const bool active = first + (int)threadIdx.x < p1;
  const int i = active ? perm[first + threadIdx.x] : 0;  // points are visited in coarse-cell order
  s_pt[threadIdx.x] = i;
  const int ncf = g.dim[0] * g.dim[1] * g.dim[2];
  const int32_t *fstart = start + (size_t)frame * ncf;
  Top3 t;
  top3_init(t);
  float ux = 0.0f, uy = 0.0f, uz = 0.0f, gap[3] = {0.0f, 0.0f, 0.0f}, cs[3];
  int cc[3] = {0, 0, 0};
#pragma unroll
  for (int a = 0; a < 3; ++a) cs[a] = g.vs[a] * (float)g.cg[a];
  const float xhi = g.lo[0] + g.vs[0] * (float)g.grid[0], yhi = g.lo[1] + g.vs[1] * (float)g.grid[1], zhi = g.lo[2] + g.vs[2] * (float)g.grid[2];
  // one coarse cell against the point (px, py, pz): skipped when its box is strictly farther than `best` (with a rounding margin, so
  // equal-distance candidates with a lower index are never lost), else its centres are compared - 4 per trip, loads issued together
  auto scan_cell = [&](int cell, int cxi, int y, int z, float px, float py, float pz, float best, Top3 &tt) {
    const float bx0 = g.lo[0] + cs[0] * (float)cxi, bx1 = (cxi == g.dim[0] - 1) ? xhi : bx0 + cs[0];
    const float by0 = g.lo[1] + cs[1] * (float)y, by1 = (y == g.dim[1] - 1) ? yhi : by0 + cs[1];
    const float bz0 = g.lo[2] + cs[2] * (float)z, bz1 = (z == g.dim[2] - 1) ? zhi : bz0 + cs[2];
    const float ex = fmaxf(fmaxf(bx0 - px, px - bx1), 0.0f), ey = fmaxf(fmaxf(by0 - py, py - by1), 0.0f), ez = fmaxf(fmaxf(bz0 - pz, pz - bz1), 0.0f);
    if ((ex * ex + ey * ey + ez * ez) * 0.9999f > best) return;
    const int s0 = fstart[cell], s1 = fstart[cell + 1];
    for (int j = s0; j < s1; j += 4) {
      float4 q[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) q[e] = sorted[min(j + e, s1 - 1)];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j + e < s1) {
          const float dx = px - q[e].x, dy = py - q[e].y, dz = pz - q[e].z;
          top3_push_lex(tt, fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(q[e].w));
        }
      }
    }
  };
  // after shell r every unvisited centre is at least lb away; the search is over when the 3rd best is strictly below (or the grid is covered)
  auto shell_done = [&](int r, const int *c3, const float *gp, float d2) {
    const float lb = fminf(gp[0] + (float)r * cs[0], fminf(gp[1] + (float)r * cs[1], gp[2] + (float)r * cs[2]));
    if (d2 < lb * lb * 0.99999f) return true;
    return c3[0] - r <= 0 && c3[1] - r <= 0 && c3[2] - r <= 0 && c3[0] + r >= g.dim[0] - 1 && c3[1] + r >= g.dim[1] - 1 && c3[2] + r >= g.dim[2] - 1;
  };
  bool resolved = !active;
  if (active) {
    const float *u = points + (size_t)i * pt_stride;
    ux = u[1]; uy = u[2]; uz = u[3];
    const float pu[3] = {ux, uy, uz};
    cg_point_cell(g, pu, cc);
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float clo = g.lo[a] + cs[a] * (float)cc[a];
      // the last coarse cell of an axis may hold more fine cells: use its true upper face
      const float chi = (cc[a] == g.dim[a] - 1) ? g.lo[a] + g.vs[a] * (float)g.grid[a] : clo + cs[a];
      gap[a] = fmaxf(fminf(pu[a] - clo, chi - pu[a]), 0.0f);
    }
    // ---- shells 0 .. CG_RMAX, every lane for its own point: 98 % of the points of a LiDAR frame end in shells 0 - 1 (~5 occupied cells,
    //      ~60 centres).  (Round 3 shared shells 2 - 3 between the 64 lanes of the wave, one unresolved point at a time with a butterfly
    //      merge of the top-3 sets: bit-identical, but 356 us instead of 170 us per 120k-point frame on the device - removed.)
    for (int r = 0; r <= CG_RMAX; ++r) {
      const int z0 = max(cc[2] - r, 0), z1 = min(cc[2] + r, g.dim[2] - 1);
      const int y0 = max(cc[1] - r, 0), y1 = min(cc[1] + r, g.dim[1] - 1);
      const int x0 = max(cc[0] - r, 0), x1 = min(cc[0] + r, g.dim[0] - 1);
      for (int z = z0; z <= z1; ++z) {
        const bool zs = (z == cc[2] - r) || (z == cc[2] + r);
        for (int y = y0; y <= y1; ++y) {
          const bool face = zs || (y == cc[1] - r) || (y == cc[1] + r);
          const int rowbase = (z * g.dim[1] + y) * g.dim[0];
          // cells of this row that lie on shell r: the whole clipped x-range on a face row, else only the two ends
          const int b0 = rowbase + x0, b1 = rowbase + x1;
          for (int wi = b0 >> 5; wi <= (b1 >> 5); ++wi) {
            uint32_t m = s_occ[wi];
            if (!m) continue;
            const int wlo = wi << 5;
            if (b0 > wlo) m &= ~0u << (b0 - wlo);
            if (b1 < wlo + 31) m &= ~0u >> (wlo + 31 - b1);
            if (!face) {
              uint32_t keep = 0;
              const int e0 = rowbase + cc[0] - r, e1 = rowbase + cc[0] + r;
              if (cc[0] - r >= 0 && (e0 >> 5) == wi) keep |= 1u << (e0 & 31);
              if (cc[0] + r < g.dim[0] && (e1 >> 5) == wi) keep |= 1u << (e1 & 31);
              m &= keep;
            }
            while (m) {
              const int cell = wlo + __ffs((int)m) - 1;
              m &= m - 1;
              scan_cell(cell, cell - rowbase, y, z, ux, uy, uz, t.d2, t);
            }
          }
        }
      }
      if (shell_done(r, cc, gap, t.d2)) { resolved = true; break; }
    }
  }
  if (active && !resolved) {
    // beyond CG_RMAX (the ~2 % of a frame whose neighbours are metres away: isolated returns, points outside the voxel range): k_devox_hard,
    // a workgroup per point
    hard_list[atomicAdd(hard_count, 1)] = i;
    s_pt[threadIdx.x] = -1;  // no output from this kernel for the point
  }
  const float r0 = __fdiv_rn(1.0f, sqrtf(t.d0) + 1e-8f), r1 = __fdiv_rn(1.0f, sqrtf(t.d1) + 1e-8f), r2 = __fdiv_rn(1.0f, sqrtf(t.d2) + 1e-8f);
  const float norm = (r0 + r1) + r2;
  s_idx[threadIdx.x * 3 + 0] = t.i0; s_idx[threadIdx.x * 3 + 1] = t.i1; s_idx[threadIdx.x * 3 + 2] = t.i2;
  s_w[threadIdx.x * 3 + 0] = __fdiv_rn(r0, norm); s_w[threadIdx.x * 3 + 1] = __fdiv_rn(r1, norm); s_w[threadIdx.x * 3 + 2] = __fdiv_rn(r2, norm);
  if (active && idx_out && s_pt[threadIdx.x] >= 0) {
    int32_t *q = idx_out + (size_t)i * 3;
    q[0] = t.i0; q[1] = t.i1; q[2] = t.i2;
  }
  if (active && w_out && s_pt[threadIdx.x] >= 0) {
    float *q = w_out + (size_t)i * 3;
    q[0] = s_w[threadIdx.x * 3]; q[1] = s_w[threadIdx.x * 3 + 1]; q[2] = s_w[threadIdx.x * 3 + 2];
  }
  __syncthreads();
  const int cnt = min(256, p1 - first);
  const int c4n = feat ? (C >> 2) : 0;  // search only (feat == NULL): indices + weights, interpolation by ls3d_interpolate_rows
  const int v0 = vx_off[frame], m = vx_off[frame + 1] - v0;
  for (int e = threadIdx.x; e < cnt * c4n; e += 256) {
    const int p = e / c4n, c4 = e % c4n;
    if (s_pt[p] < 0) continue;
    if (m <= 0) {
      *(float4 *)(out + (size_t)s_pt[p] * out_ld + c4 * 4) = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      continue;
    }
    const float w0 = s_w[p * 3], w1 = s_w[p * 3 + 1], w2 = s_w[p * 3 + 2];
    const float4 a = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3]) * feat_ld + c4 * 4);
    const float4 b = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3 + 1]) * feat_ld + c4 * 4);
    const float4 c = *(const float4 *)(feat + (size_t)(v0 + s_idx[p * 3 + 2]) * feat_ld + c4 * 4);
    float4 o;
    o.x = fmaf(w2, c.x, fmaf(w1, b.x, w0 * a.x));
    o.y = fmaf(w2, c.y, fmaf(w1, b.y, w0 * a.y));
    o.z = fmaf(w2, c.z, fmaf(w1, b.z, w0 * a.z));
    o.w = fmaf(w2, c.w, fmaf(w1, b.w, w0 * a.w));
    *(float4 *)(out + (size_t)s_pt[p] * out_ld + c4 * 4) = o;
  }
}

// block-wide merge of the threads' top-3 sets in lexicographic (distance, index) order: wave butterfly, then thread 0 over the 4 waves'
// sets; every thread returns with the workgroup's set.  Untouched slots are (+inf, 0): they never displace a real candidate, and the real
// candidates of different threads are distinct.
__device__ __forceinline__ void dh_block_top3(Top3 &t, float *s_d, int *s_i, float *s_fd, int *s_fi) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const float e0 = __shfl_xor(t.d0, off), e1 = __shfl_xor(t.d1, off), e2 = __shfl_xor(t.d2, off);
    const int j0 = __shfl_xor(t.i0, off), j1 = __shfl_xor(t.i1, off), j2 = __shfl_xor(t.i2, off);
    if (e0 < __int_as_float(0x7f800000)) top3_push_lex(t, e0, j0);
    if (e1 < __int_as_float(0x7f800000)) top3_push_lex(t, e1, j1);
    if (e2 < __int_as_float(0x7f800000)) top3_push_lex(t, e2, j2);
  }
  __syncthreads();  // earlier readers of s_* are done
  if (lane == 0) {
    s_d[wave * 3 + 0] = t.d0; s_d[wave * 3 + 1] = t.d1; s_d[wave * 3 + 2] = t.d2;
    s_i[wave * 3 + 0] = t.i0; s_i[wave * 3 + 1] = t.i1; s_i[wave * 3 + 2] = t.i2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    Top3 b;
    top3_init(b);
    for (int q = 0; q < 12; ++q)
      if (s_d[q] < __int_as_float(0x7f800000)) top3_push_lex(b, s_d[q], s_i[q]);
    s_fd[0] = b.d0; s_fd[1] = b.d1; s_fd[2] = b.d2;
    s_fi[0] = b.i0; s_fi[1] = b.i1; s_fi[2] = b.i2;
  }
  __syncthreads();
  t.d0 = s_fd[0]; t.d1 = s_fd[1]; t.d2 = s_fd[2];
  t.i0 = s_fi[0]; t.i1 = s_fi[1]; t.i2 = s_fi[2];
}

// One workgroup per deferred point (the ~2 % of a frame whose neighbours are metres away: isolated returns, points outside the voxel
// range): exact top-3 in lexicographic (distance, index) order, then the interpolated row.  Scanning every voxel centre of the frame
// for each of them cost more than the other 98 % of the points (2.4k points x 65.9k centres = 2.4 GB of L2 reads per frame), so the
// workgroup prunes with the coarse grid the shell search uses: (1) an upper bound on the 3rd-nearest distance from 1024 centres taken
// evenly from the cell-sorted list, (2) the threads walk the occupancy bitmap, and only cells whose box is not farther than the bound
// (nor than the thread's own 3rd best; the margin of the shell search) have their centres compared - with the same f32 expression and
// the same frame-local indices as everywhere else, so the result is the brute-force one.  Frames with few centres are scanned whole.
#define DH_SCAN_ALL 2048
__global__ __launch_bounds__(256) void k_devox_hard(const float *points, int pt_stride, const int32_t *hard_list, const int32_t *hard_count,
                                                   const float *centers, const int32_t *vx_off, CGeom g, const int32_t *start,
                                                   const float4 *sorted, const uint32_t *occ, const float *feat, int feat_ld, int C, float *out,
                                                   int out_ld, int32_t *idx_out, float *w_out) {
  __shared__ float s_d[4 * 3];
  __shared__ int s_i[4 * 3];
  __shared__ float s_fd[3];
  __shared__ int s_fi[3];
  __shared__ float s_w3[3];
  const int count = *hard_count;
  const int ncf = g.dim[0] * g.dim[1] * g.dim[2];
  for (int h = blockIdx.x; h < count; h += gridDim.x) {
    const int i = hard_list[h];
    const float *u = points + (size_t)i * pt_stride;
    const int frame = (int)u[0];
    const float ux = u[1], uy = u[2], uz = u[3];
    const int v0 = vx_off[frame], m = vx_off[frame + 1] - v0;
    Top3 t;
    top3_init(t);
    if (m < DH_SCAN_ALL) {
      // 4 centres per trip, their 16-byte loads issued together from clamped addresses; the result does not depend on the order of the pushes
      const float4 *cen = (const float4 *)centers + v0;
      for (int k = threadIdx.x; k < m; k += 1024) {
        float4 q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = cen[min(k + e * 256, m - 1)];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (k + e * 256 < m) {
            const float dx = ux - q[e].y, dy = uy - q[e].z, dz = uz - q[e].w;
            top3_push_lex(t, fmaf(dz, dz, fmaf(dy, dy, dx * dx)), k + e * 256);
          }
        }
      }
    } else {
      const int32_t *fstart = start + (size_t)frame * ncf;
      const int s_lo = fstart[0];
      // (1) any three centres bound the 3rd-nearest distance from above: 4 per thread, evenly spread over the cell-sorted list
      {
        float4 q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) q[e] = sorted[s_lo + (int)(((long long)(threadIdx.x * 4 + e) * m) >> 10)];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dx = ux - q[e].x, dy = uy - q[e].y, dz = uz - q[e].z;
          top3_push_lex(t, fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(q[e].w));
        }
      }
      dh_block_top3(t, s_d, s_i, s_fd, s_fi);
      const float bound = t.d2;
      top3_init(t);  // the sampled centres are met again in their cells
      // (2) occupied cells whose box can hold a centre at most `bound` away
      float cs[3];
#pragma unroll
      for (int a = 0; a < 3; ++a) cs[a] = g.vs[a] * (float)g.cg[a];
      const uint32_t *focc = occ + (size_t)frame * g.wpf;
      const int plane = g.dim[0] * g.dim[1];
      const float xhi = g.lo[0] + g.vs[0] * (float)g.grid[0], yhi = g.lo[1] + g.vs[1] * (float)g.grid[1], zhi = g.lo[2] + g.vs[2] * (float)g.grid[2];
      for (int w = threadIdx.x; w < g.wpf; w += 256) {
        uint32_t bits = focc[w];
        if (!bits) continue;
        // the word's first cell (one pair of divisions per word, not per cell); its 32 cells run along x and usually stay in one row
        const int base = w << 5;
        int z = base / plane, rem = base - z * plane, y = rem / g.dim[0];
        const int x0 = rem - y * g.dim[0];
        const bool one_row = x0 + 31 < g.dim[0];
        float eyz = 0.0f;  // squared distance to the row's y / z slab
        if (one_row) {
          // (the last coarse cell of an axis may hold more fine cells: its true upper face)
          const float by0 = g.lo[1] + cs[1] * (float)y, by1 = (y == g.dim[1] - 1) ? yhi : by0 + cs[1];
          const float bz0 = g.lo[2] + cs[2] * (float)z, bz1 = (z == g.dim[2] - 1) ? zhi : bz0 + cs[2];
          const float ey = fmaxf(fmaxf(by0 - uy, uy - by1), 0.0f), ez = fmaxf(fmaxf(bz0 - uz, uz - bz1), 0.0f);
          eyz = ey * ey + ez * ez;
          // the whole word at once: the box of its 32 cells
          const float wx0 = g.lo[0] + cs[0] * (float)x0, wx1 = (x0 + 31 == g.dim[0] - 1) ? xhi : g.lo[0] + cs[0] * (float)(x0 + 32);
          const float ewx = fmaxf(fmaxf(wx0 - ux, ux - wx1), 0.0f);
          if ((ewx * ewx + eyz) * 0.9999f > fminf(bound, t.d2)) continue;
        }
        while (bits) {
          const int bit = __ffs((int)bits) - 1;
          bits &= bits - 1;
          int x = x0 + bit, yy = y, zz = z;
          float e2 = eyz;
          if (!one_row) {  // the word wraps into the next row(s)
            while (x >= g.dim[0]) {
              x -= g.dim[0];
              if (++yy == g.dim[1]) { yy = 0; ++zz; }
            }
            const float by0 = g.lo[1] + cs[1] * (float)yy, by1 = (yy == g.dim[1] - 1) ? yhi : by0 + cs[1];
            const float bz0 = g.lo[2] + cs[2] * (float)zz, bz1 = (zz == g.dim[2] - 1) ? zhi : bz0 + cs[2];
            const float ey = fmaxf(fmaxf(by0 - uy, uy - by1), 0.0f), ez = fmaxf(fmaxf(bz0 - uz, uz - bz1), 0.0f);
            e2 = ey * ey + ez * ez;
          }
          const float bx0 = g.lo[0] + cs[0] * (float)x, bx1 = (x == g.dim[0] - 1) ? xhi : bx0 + cs[0];
          const float ex = fmaxf(fmaxf(bx0 - ux, ux - bx1), 0.0f);
          // strictly farther than the bound (with the rounding margin of the shell search): equal-distance candidates are never lost
          if ((ex * ex + e2) * 0.9999f > fminf(bound, t.d2)) continue;
          const int cell = base + bit;
          const int c0 = fstart[cell], c1 = fstart[cell + 1];
          for (int j = c0; j < c1; j += 4) {
            float4 q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = sorted[min(j + e, c1 - 1)];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (j + e < c1) {
                const float dx = ux - q[e].x, dy = uy - q[e].y, dz = uz - q[e].z;
                top3_push_lex(t, fmaf(dz, dz, fmaf(dy, dy, dx * dx)), __float_as_int(q[e].w));
              }
            }
          }
        }
      }
    }
    dh_block_top3(t, s_d, s_i, s_fd, s_fi);
    if (threadIdx.x == 0) {
      const float r0 = __fdiv_rn(1.0f, sqrtf(t.d0) + 1e-8f), r1 = __fdiv_rn(1.0f, sqrtf(t.d1) + 1e-8f), r2 = __fdiv_rn(1.0f, sqrtf(t.d2) + 1e-8f);
      const float norm = (r0 + r1) + r2;
      s_w3[0] = __fdiv_rn(r0, norm); s_w3[1] = __fdiv_rn(r1, norm); s_w3[2] = __fdiv_rn(r2, norm);
      if (idx_out) { idx_out[(size_t)i * 3] = t.i0; idx_out[(size_t)i * 3 + 1] = t.i1; idx_out[(size_t)i * 3 + 2] = t.i2; }
      if (w_out) { w_out[(size_t)i * 3] = s_w3[0]; w_out[(size_t)i * 3 + 1] = s_w3[1]; w_out[(size_t)i * 3 + 2] = s_w3[2]; }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < (feat ? C : 0); c += 256) {
      float o = 0.0f;
      if (m > 0) {
        const float a = feat[(size_t)(v0 + t.i0) * feat_ld + c], b2 = feat[(size_t)(v0 + t.i1) * feat_ld + c],
                    c2 = feat[(size_t)(v0 + t.i2) * feat_ld + c];
        o = fmaf(s_w3[2], c2, fmaf(s_w3[1], b2, s_w3[0] * a));
      }
      out[(size_t)i * out_ld + c] = o;
    }
  }
}

static inline size_t dv_align(size_t v) { return (v + 255) & ~(size_t)255; }
// coarse cells of 8x8x4 fine voxels, doubled until one frame's occupancy bitmap fits in 48 KB of LDS
static void cg_setup(const int32_t grid_xyz[3], CGeom &g) {
  int cg[3] = {8, 8, 4};
  for (;;) {
    long long cells = 1;
    for (int a = 0; a < 3; ++a) { g.cg[a] = cg[a]; g.dim[a] = grid_xyz[a] / cg[a]; if (g.dim[a] < 1) g.dim[a] = 1; cells *= g.dim[a]; }
    g.wpf = (int)((cells + 31) / 32);
    if ((long long)g.wpf * 4 <= 48 * 1024) break;
    for (int a = 0; a < 3; ++a) cg[a] *= 2;
  }
}

static size_t dv_points_ws(int n_points, long long ncell);
extern "C" size_t ls3d_devoxelize_grid_workspace_bytes(int n_points, int n_voxels, int batch, const int32_t grid_xyz[3]) {
  CGeom g;
  cg_setup(grid_xyz, g);
  const long long ncell = (long long)batch * g.dim[0] * g.dim[1] * g.dim[2];
  return dv_points_ws(n_points, ncell) + dv_align((size_t)(n_voxels > 0 ? n_voxels : 1) * 4) + 3 * dv_align((size_t)(ncell + 1) * 4) + dv_align(ls3d_scan_tmp_ints(ncell + 1) * 4) +
         dv_align((size_t)(n_voxels > 0 ? n_voxels : 1) * 16) + dv_align((size_t)batch * g.wpf * 4);
}

// extra workspace for the coarse-cell ordering of the query points
static size_t dv_points_ws(int n_points, long long ncell) { return 3 * dv_align((size_t)(n_points > 0 ? n_points : 1) * 4) + 3 * dv_align((size_t)(ncell + 1) * 4) + 256; }

extern "C" int ls3d_devoxelize_grid(const float *points, int pt_stride, int n_points, const int32_t *pt_off, int max_frame_points,
                                    const int32_t *coords, const float *centers, int n_voxels, const int32_t *n_voxels_dev,
                                    const int32_t *vx_off, int batch, const float vs[3], const float lo[3], const int32_t grid_xyz[3],
                                    const float *feat, int feat_ld, int c, float *out, int out_ld, int32_t *idx_out, float *w_out,
                                    void *workspace, size_t workspace_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!points || !pt_off || !coords || !centers || !vx_off || !vs || !lo || !grid_xyz || !workspace || batch < 1 || pt_stride < 4)
    return LS3D_ERR_ARG;
  if (feat ? (!out || (c % 4) || (feat_ld % 4) || (out_ld % 4) || feat_ld < c || out_ld < c) : (!idx_out || !w_out)) return LS3D_ERR_ARG;
  if (n_voxels < 1) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_devoxelize_grid_workspace_bytes(n_points, n_voxels, batch, grid_xyz)) return LS3D_ERR_WORKSPACE;
  if (n_points == 0 || max_frame_points == 0) return LS3D_OK;
  CGeom g;
  for (int a = 0; a < 3; ++a) { g.vs[a] = vs[a]; g.lo[a] = lo[a]; g.grid[a] = grid_xyz[a]; }
  cg_setup(grid_xyz, g);
  const long long ncell = (long long)batch * g.dim[0] * g.dim[1] * g.dim[2];
  if (ncell + 1 > 0x7FFFFFFFLL) return LS3D_ERR_UNSUPPORTED;
  // everything that starts at zero lies in front, one block = one memset: cell counts / cursors of the centres and of the query points, the occupancy
  // bitmaps, the deferred-point counter (six fills per frame before)
  char *base = (char *)workspace;
  char *const zero_begin = base;
  int32_t *cnt = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  int32_t *cursor = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  int32_t *pcnt = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  int32_t *pcursor = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  uint32_t *occ = (uint32_t *)base; base += dv_align((size_t)batch * g.wpf * 4);
  int32_t *hard_count = (int32_t *)base; base += 256;
  char *const zero_end = base;
  int32_t *cell_of = (int32_t *)base; base += dv_align((size_t)n_voxels * 4);
  int32_t *start = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  int32_t *scan_tmp = (int32_t *)base; base += dv_align(ls3d_scan_tmp_ints(ncell + 1) * 4);
  float4 *sorted = (float4 *)base; base += dv_align((size_t)n_voxels * 16);
  int32_t *pcell = (int32_t *)base; base += dv_align((size_t)n_points * 4);
  int32_t *perm = (int32_t *)base; base += dv_align((size_t)n_points * 4);
  int32_t *pstart = (int32_t *)base; base += dv_align((size_t)(ncell + 1) * 4);
  int32_t *hard_list = (int32_t *)base;
  hipMemsetAsync(zero_begin, 0, (size_t)(zero_end - zero_begin), stream);
  hipLaunchKernelGGL(k_pt_count, ls3d_grid(n_points), dim3(256), 0, stream, points, pt_stride, n_points, batch, g, pcell, pcnt);
  int rcp = ls3d_exclusive_scan_i32(pcnt, pstart, (int)(ncell + 1), scan_tmp, nullptr, stream);
  if (rcp != LS3D_OK) return rcp;
  hipLaunchKernelGGL(k_pt_fill, ls3d_grid(n_points), dim3(256), 0, stream, n_points, (const int32_t *)pcell, (const int32_t *)pstart, pcursor, perm);
  hipLaunchKernelGGL(k_cg_count, ls3d_grid(n_voxels), dim3(256), 0, stream, coords, n_voxels, n_voxels_dev, g, cell_of, cnt, occ);
  int rc = ls3d_exclusive_scan_i32(cnt, start, (int)(ncell + 1), scan_tmp, nullptr, stream);
  if (rc != LS3D_OK) return rc;
  hipLaunchKernelGGL(k_cg_fill, ls3d_grid(n_voxels), dim3(256), 0, stream, centers, n_voxels, n_voxels_dev, vx_off, (const int32_t *)cell_of,
                     (const int32_t *)start, cursor, sorted);
  hipLaunchKernelGGL(k_devox_grid, dim3((max_frame_points + 255) / 256, batch), dim3(256), (size_t)g.wpf * 4, stream, points, pt_stride, pt_off,
                     (const int32_t *)perm, g, (const int32_t *)start, (const float4 *)sorted, (const uint32_t *)occ, vx_off, feat, feat_ld, c, out, out_ld, idx_out, w_out,
                     hard_list, hard_count);
  hipLaunchKernelGGL(k_devox_hard, dim3(n_points < 2048 ? (n_points > 0 ? n_points : 1) : 2048), dim3(256), 0, stream, points, pt_stride,
                     (const int32_t *)hard_list, (const int32_t *)hard_count, centers, vx_off, g, (const int32_t *)start, (const float4 *)sorted,
                     (const uint32_t *)occ, feat, feat_ld, c, out, out_ld, idx_out, w_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// out[p] = sum_j w[p][j] * feat[vx_off[frame(p)] + idx[p][j]]: the interpolation half of ls3d_devoxelize_grid when the search ran
// earlier (feat == NULL there); same arithmetic as the fused kernels
__global__ __launch_bounds__(256) void k_interp_rows(const float *feat, int feat_ld, int C, const int32_t *idx, const float *w, const float *points,
                                                    int pt_stride, const int32_t *vx_off, int n, float *out, int out_ld) {
  const int c4n = C >> 2;
  const long long work = (long long)n * c4n;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int p = (int)(t / c4n), c4 = (int)(t % c4n);
    const int f = (int)points[(size_t)p * pt_stride];
    const int v0 = vx_off[f], m = vx_off[f + 1] - v0;
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (m > 0) {
      const float w0 = w[(size_t)p * 3], w1 = w[(size_t)p * 3 + 1], w2 = w[(size_t)p * 3 + 2];
      const float4 a = *(const float4 *)(feat + (size_t)(v0 + idx[(size_t)p * 3]) * feat_ld + c4 * 4);
      const float4 b = *(const float4 *)(feat + (size_t)(v0 + idx[(size_t)p * 3 + 1]) * feat_ld + c4 * 4);
      const float4 c = *(const float4 *)(feat + (size_t)(v0 + idx[(size_t)p * 3 + 2]) * feat_ld + c4 * 4);
      o.x = fmaf(w2, c.x, fmaf(w1, b.x, w0 * a.x));
      o.y = fmaf(w2, c.y, fmaf(w1, b.y, w0 * a.y));
      o.z = fmaf(w2, c.z, fmaf(w1, b.z, w0 * a.z));
      o.w = fmaf(w2, c.w, fmaf(w1, b.w, w0 * a.w));
    }
    *(float4 *)(out + (size_t)p * out_ld + c4 * 4) = o;
  }
}

// Backward of ls3d_interpolate_rows with respect to the voxel features (interpolate_gpu.cu:127-149 scatters with atomicAdd: the order of the
// float additions, and with it the last bits of the gradient, changes from run to run).  Here: the 3 n (point, neighbour) entries are sorted
// by their voxel row (stable radix sort: entries of a voxel stay in entry order), and every (voxel, 4-channel group) sums its entries in that
// order: bit-reproducible.  grad_feat[v] = sum over entries e = (p, j) with vx_off[frame(p)] + idx[p][j] == v of weight[p][j] * grad_out[p].
__global__ __launch_bounds__(256) void k_interp_bwd_keys(const int32_t *idx, const float *points, int pt_stride, const int32_t *vx_off, int n, int V,
                                                         uint32_t *keys) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < 3 * n; e += gridDim.x * blockDim.x) {
    const int p = e / 3;
    const int f = (int)points[(size_t)p * pt_stride];
    const int v0 = vx_off[f], m = vx_off[f + 1] - v0;
    const int i = idx[e];
    keys[e] = (m > 0 && i >= 0 && i < m) ? (uint32_t)(v0 + i) : (uint32_t)V;  // frames without voxels: the entry sorts behind every row
  }
}
__global__ __launch_bounds__(256) void k_interp_bwd_starts(const uint32_t *skeys, int n3, int V, int32_t *start) {
  // start[v] = first sorted position with key >= v (v = 0 .. V): every thread looks at one boundary of the sorted key list
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= n3; i += gridDim.x * blockDim.x) {
    const int lo = i == 0 ? -1 : (int)min(skeys[i - 1], (uint32_t)V), hi = i == n3 ? V : (int)min(skeys[i], (uint32_t)V);
    for (int v = lo + 1; v <= hi; ++v) start[v] = i;
  }
}
__global__ __launch_bounds__(256) void k_interp_bwd_sum(const float *gout, int go_ld, int C, const float *w, const int32_t *order, const int32_t *start,
                                                        int V, float *gfeat, int gf_ld) {
  const int c4n = C >> 2;
  const long long work = (long long)V * c4n;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / c4n), c4 = (int)(t % c4n);
    float4 o = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int i = start[v]; i < start[v + 1]; ++i) {
      const int e = order[i];
      const float we = w[e];
      const float4 g = *(const float4 *)(gout + (size_t)(e / 3) * go_ld + c4 * 4);
      o.x = fmaf(we, g.x, o.x); o.y = fmaf(we, g.y, o.y); o.z = fmaf(we, g.z, o.z); o.w = fmaf(we, g.w, o.w);
    }
    *(float4 *)(gfeat + (size_t)v * gf_ld + c4 * 4) = o;
  }
}
int ls3d_radix_sort_pairs(const uint32_t *keys_in, const int32_t *vals_in, int n, const int32_t *n_dev, int bits, uint32_t *keys_out, int32_t *vals_out,
                          void *workspace, size_t workspace_bytes, hipStream_t stream);
extern "C" size_t ls3d_radix_sort_workspace_bytes(int n);
static inline size_t ib_align(size_t v) { return (v + 255) & ~(size_t)255; }
extern "C" size_t ls3d_interpolate_rows_backward_workspace_bytes(int n_points, int n_voxels) {
  if (n_points < 0 || n_voxels < 0) return 0;
  const size_t n3 = (size_t)3 * n_points;
  return 3 * ib_align(n3 * 4) + ib_align(((size_t)n_voxels + 2) * 4) + ls3d_radix_sort_workspace_bytes((int)n3) + 256;
}
extern "C" int ls3d_interpolate_rows_backward(const float *grad_out, int go_ld, int c, const int32_t *idx, const float *weight, const float *points,
                                              int pt_stride, const int32_t *vx_off, int n_points, int n_voxels, void *workspace,
                                              size_t workspace_bytes, float *grad_feat, int gf_ld, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!grad_feat || n_points < 0 || n_voxels < 0 || (c % 4) || (go_ld % 4) || (gf_ld % 4) || go_ld < c || gf_ld < c) return LS3D_ERR_ARG;
  if (n_voxels == 0) return LS3D_OK;
  if (!workspace || ((uintptr_t)workspace & 15)) return LS3D_ERR_ARG;
  if (n_points > 0 && (!grad_out || !idx || !weight || !points || !vx_off || pt_stride < 1)) return LS3D_ERR_ARG;
  if ((long long)3 * n_points > 0x7FFFFFFFll) return LS3D_ERR_UNSUPPORTED;
  if (workspace_bytes < ls3d_interpolate_rows_backward_workspace_bytes(n_points, n_voxels)) return LS3D_ERR_WORKSPACE;
  const int n3 = 3 * n_points;
  char *wsp = (char *)workspace;
  uint32_t *keys = (uint32_t *)wsp; wsp += ib_align((size_t)n3 * 4);
  uint32_t *skeys = (uint32_t *)wsp; wsp += ib_align((size_t)n3 * 4);
  int32_t *order = (int32_t *)wsp; wsp += ib_align((size_t)n3 * 4);
  int32_t *start = (int32_t *)wsp; wsp += ib_align(((size_t)n_voxels + 2) * 4);
  int bits = 1;
  while ((1ll << bits) <= (long long)n_voxels) ++bits;  // keys 0 .. V
  if (n3 > 0) {  // (no points: every segment is empty, the gradient is zero)
    hipLaunchKernelGGL(k_interp_bwd_keys, ls3d_grid(n3), dim3(256), 0, stream, idx, points, pt_stride, vx_off, n_points, n_voxels, keys);
    const int rc = ls3d_radix_sort_pairs(keys, nullptr, n3, nullptr, bits, skeys, order, wsp, workspace_bytes - (size_t)(wsp - (char *)workspace), stream);
    if (rc != LS3D_OK) return rc;
  }
  hipLaunchKernelGGL(k_interp_bwd_starts, ls3d_grid((long long)n3 + 1), dim3(256), 0, stream, (const uint32_t *)skeys, n3, n_voxels, start);
  hipLaunchKernelGGL(k_interp_bwd_sum, ls3d_grid((long long)n_voxels * (c / 4)), dim3(256), 0, stream, grad_out, go_ld, c, weight, (const int32_t *)order,
                     (const int32_t *)start, n_voxels, grad_feat, gf_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_interpolate_rows(const float *feat, int feat_ld, int c, const int32_t *idx, const float *weight, const float *points,
                                     int pt_stride, const int32_t *vx_off, int n_points, float *out, int out_ld, ls3d_stream_t stream) {
  if (!feat || !idx || !weight || !points || !vx_off || !out || n_points < 0 || pt_stride < 1) return LS3D_ERR_ARG;
  if ((c % 4) || (feat_ld % 4) || (out_ld % 4) || feat_ld < c || out_ld < c) return LS3D_ERR_ARG;
  if (n_points == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_interp_rows, ls3d_grid((long long)n_points * (c / 4)), dim3(256), 0, (hipStream_t)stream, feat, feat_ld, c, idx, weight, points,
                     pt_stride, vx_off, n_points, out, out_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// off[b] = first row whose batch index (column `col` of a frame-sorted table) is >= b, b = 0..batch  (binary search)
__global__ void k_frame_offsets(const void *table, int is_float, int stride, int col, int n, const int32_t *n_dev, int batch, int32_t *off) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b > batch) return;
  int lo = 0, hi = ls3d_count(n, n_dev);
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    const int v = is_float ? (int)((const float *)table)[(size_t)mid * stride + col] : ((const int32_t *)table)[(size_t)mid * stride + col];
    if (v < b) lo = mid + 1; else hi = mid;
  }
  off[b] = lo;
}

extern "C" int ls3d_frame_offsets(const void *table, int is_float, int stride, int col, int n, const int32_t *n_dev, int batch, int32_t *off,
                                  ls3d_stream_t stream) {
  if (!table || !off || n < 0 || batch < 1 || stride < 1 || col < 0 || col >= stride) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_frame_offsets, dim3((batch + 1 + 63) / 64), dim3(64), 0, (hipStream_t)stream, table, is_float, stride, col, n, n_dev, batch, off);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_voxel_centers(const int32_t *coords, int n, const int32_t *n_dev, const float vs[3], const float lo[3], float *out,
                                  ls3d_stream_t stream) {
  if (!coords || !vs || !lo || !out || n < 0) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_voxel_centers, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, coords, n, n_dev, vs[0], vs[1], vs[2], lo[0], lo[1],
                     lo[2], out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_three_nn(int b, int n, int m, const float *unknown, const float *known, float *dist2, int32_t *idx,
                             ls3d_stream_t stream) {
  if (!unknown || !known || !dist2 || !idx || b < 0 || n < 0 || m < 0) return LS3D_ERR_ARG;
  if (b == 0 || n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_three_nn, dim3((n + 255) / 256, b), dim3(256), 0, (hipStream_t)stream, n, m, unknown, known, dist2, idx);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_three_interpolate(int b, int c, int m, int n, const float *points, const int32_t *idx, const float *weight, float *out,
                                      ls3d_stream_t stream) {
  if (!points || !idx || !weight || !out || b < 0 || c < 0 || n < 0 || m < 1) return LS3D_ERR_ARG;
  if (b == 0 || c == 0 || n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_three_interp_cm, dim3((n + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream, c, m, n, points, idx, weight, out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_three_interpolate_grad(int b, int c, int n, int m, const float *grad_out, const int32_t *idx, const float *weight,
                                           float *grad_points, ls3d_stream_t stream) {
  if (!grad_out || !idx || !weight || !grad_points || b < 0 || c < 0 || n < 0 || m < 1) return LS3D_ERR_ARG;
  if (b == 0 || c == 0 || n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_three_interp_grad_cm, dim3((n + 255) / 256, c, b), dim3(256), 0, (hipStream_t)stream, c, n, m, grad_out, idx, weight,
                     grad_points);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_devoxelize(const float *points, int pt_stride, int n_points, const int32_t *pt_off, const float *centers,
                               const int32_t *vx_off, int batch, int max_frame_points, const float *feat, int feat_ld, int c, float *out,
                               int out_ld, int32_t *idx_out, ls3d_stream_t stream) {
  if (!points || !pt_off || !centers || !vx_off || !feat || !out || batch < 1 || pt_stride < 4) return LS3D_ERR_ARG;
  if ((c % 4) || (feat_ld % 4) || (out_ld % 4) || feat_ld < c || out_ld < c) return LS3D_ERR_ARG;
  if (n_points == 0 || max_frame_points == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_devoxelize, dim3((max_frame_points + 255) / 256, batch), dim3(256), 0, (hipStream_t)stream, points, pt_stride, pt_off,
                     centers, vx_off, feat, feat_ld, c, out, out_ld, idx_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
