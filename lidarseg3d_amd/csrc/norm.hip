// norm.hip — row LayerNorm over [n, c] activations, forward and backward, for the training step (c <= 256, c % 4 == 0).
//
// The reader's and the SF-Phase decoder's LayerNorms (det3d/models/readers/voxel_encoder.py:149-163 TransformerEncoderLayerPreNorm,
// context_module.py:319-376 decoder layers: nn.LayerNorm over 64 / 96 channels of 10^5 - 10^6 token rows) run under torch autograd in
// training; torch's kernels take 0.36 ms forward and 0.59 ms backward (two launches + a column reduction) for 360 000 x 96 floats,
// which is 0.4 GB of traffic = 60 us at HBM speed.  Here: 32 lanes per row (one float4 per lane and 128 columns), statistics by shuffles
// (two-pass: mean, then the centred second moment - no cancellation), the backward's column sums d gamma / d beta accumulated per lane
// over a grid-stride walk of the rows, reduced per block in LDS and over the blocks by a second launch in a fixed order: deterministic.
#include "common.h"

constexpr int LN_MAXV = 2;  // float4 per lane: c <= 256

template <int NV>
__global__ __launch_bounds__(256) void k_ln_fwd(const float *__restrict__ x, int n, int c, const float *__restrict__ gamma, const float *__restrict__ beta,
                                                float eps, float *__restrict__ y, float *__restrict__ stats) {
  const int tid = threadIdx.x, sub = tid & 31, grp = tid >> 5;
  const float inv_c = 1.0f / (float)c;
  float4 g[NV], b[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = sub * 4 + v * 128;
    g[v] = col < c ? *(const float4 *)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[v] = col < c ? *(const float4 *)(beta + col) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row0 = blockIdx.x * 8; row0 < n; row0 += gridDim.x * 8) {  // the same trip count for the two row groups of a wave (shuffles)
    const int row = row0 + grp;
    const bool live = row < n;
    float4 a[NV];
    float s = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = sub * 4 + v * 128;
      a[v] = (live && col < c) ? *(const float4 *)(x + (size_t)row * c + col) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (a[v].x + a[v].y) + (a[v].z + a[v].w);
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    const float mean = s * inv_c;
    float q = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = sub * 4 + v * 128;
      if (col < c) {
        const float dx = a[v].x - mean, dy = a[v].y - mean, dz = a[v].z - mean, dw = a[v].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) q += __shfl_xor(q, d);
    const float rstd = 1.0f / sqrtf(q * inv_c + eps);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = sub * 4 + v * 128;
      if (live && col < c) {
        float4 o;
        o.x = (a[v].x - mean) * rstd * g[v].x + b[v].x;
        o.y = (a[v].y - mean) * rstd * g[v].y + b[v].y;
        o.z = (a[v].z - mean) * rstd * g[v].z + b[v].z;
        o.w = (a[v].w - mean) * rstd * g[v].w + b[v].w;
        *(float4 *)(y + (size_t)row * c + col) = o;
      }
    }
    if (live && sub == 0 && stats) {
      stats[2 * (size_t)row] = mean;
      stats[2 * (size_t)row + 1] = rstd;
    }
  }
}

// d x per row; per block the column sums of dy * xhat and dy over the block's rows -> partial[block][2][c]
template <int NV>
__global__ __launch_bounds__(256) void k_ln_bwd(const float *__restrict__ x, const float *__restrict__ dy, const float *__restrict__ gamma,
                                                const float *__restrict__ stats, int n, int c, float *__restrict__ dx, float *__restrict__ partial) {
  __shared__ float s_red[8][2][NV * 128];
  const int tid = threadIdx.x, sub = tid & 31, grp = tid >> 5;
  const float inv_c = 1.0f / (float)c;
  float4 g[NV], ag[NV], ab[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = sub * 4 + v * 128;
    g[v] = col < c ? *(const float4 *)(gamma + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    ag[v] = make_float4(0.f, 0.f, 0.f, 0.f);
    ab[v] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int row0 = blockIdx.x * 8; row0 < n; row0 += gridDim.x * 8) {
    const int row = row0 + grp;
    const bool live = row < n;
    const float mean = live ? stats[2 * (size_t)row] : 0.0f, rstd = live ? stats[2 * (size_t)row + 1] : 0.0f;
    float4 xh[NV], d[NV];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = sub * 4 + v * 128;
      if (live && col < c) {
        const float4 a = *(const float4 *)(x + (size_t)row * c + col);
        d[v] = *(const float4 *)(dy + (size_t)row * c + col);
        xh[v] = make_float4((a.x - mean) * rstd, (a.y - mean) * rstd, (a.z - mean) * rstd, (a.w - mean) * rstd);
      } else {
        d[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        xh[v] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      ag[v].x += d[v].x * xh[v].x; ag[v].y += d[v].y * xh[v].y; ag[v].z += d[v].z * xh[v].z; ag[v].w += d[v].w * xh[v].w;
      ab[v].x += d[v].x; ab[v].y += d[v].y; ab[v].z += d[v].z; ab[v].w += d[v].w;
      d[v].x *= g[v].x; d[v].y *= g[v].y; d[v].z *= g[v].z; d[v].w *= g[v].w;  // d xhat
      s1 += (d[v].x + d[v].y) + (d[v].z + d[v].w);
      s2 += (d[v].x * xh[v].x + d[v].y * xh[v].y) + (d[v].z * xh[v].z + d[v].w * xh[v].w);
    }
#pragma unroll
    for (int k = 16; k >= 1; k >>= 1) {
      s1 += __shfl_xor(s1, k);
      s2 += __shfl_xor(s2, k);
    }
    const float c1 = s1 * inv_c, c2 = s2 * inv_c;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int col = sub * 4 + v * 128;
      if (live && col < c) {
        float4 o;
        o.x = rstd * (d[v].x - c1 - xh[v].x * c2);
        o.y = rstd * (d[v].y - c1 - xh[v].y * c2);
        o.z = rstd * (d[v].z - c1 - xh[v].z * c2);
        o.w = rstd * (d[v].w - c1 - xh[v].w * c2);
        *(float4 *)(dx + (size_t)row * c + col) = o;
      }
    }
  }
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    *(float4 *)&s_red[grp][0][sub * 4 + v * 128] = ag[v];
    *(float4 *)&s_red[grp][1][sub * 4 + v * 128] = ab[v];
  }
  __syncthreads();
  for (int i = tid; i < 2 * c; i += 256) {  // the 8 row groups of the block, in order
    const int which = i / c, col = i % c;
    float s = 0.0f;
#pragma unroll
    for (int gq = 0; gq < 8; ++gq) s += s_red[gq][which][col];
    partial[((size_t)blockIdx.x * 2 + which) * c + col] = s;
  }
}

// column sums of the row blocks' partials [block][2][c] -> dgamma, dbeta.  A workgroup owns 32 columns (lanes along the columns: 128-byte
// reads); its 8 row-block groups each add every 8th block in order, then the 8 sums are added in order: a fixed summation tree.  (One thread
// per column walking all 1024 blocks took 230 us per call: 1024 dependent steps on 6 wavefronts.)
__global__ __launch_bounds__(256) void k_ln_bwd_reduce(const float *__restrict__ partial, int nblocks, int c, float *__restrict__ dgamma, float *__restrict__ dbeta) {
  __shared__ float red[8][32];
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + lane;  // column of the [2 c] wide partial rows: dgamma columns, then dbeta columns
  float s = 0.0f;
  if (i < 2 * c) {
#pragma unroll 8
    for (int b = grp; b < nblocks; b += 8) s += partial[(size_t)b * 2 * c + i];
  }
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && i < 2 * c) {
    float t = red[0][lane];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][lane];
    (i >= c ? dbeta : dgamma)[i >= c ? i - c : i] = t;
  }
}

static inline int ln_blocks(int n) {
  const int want = (n + 7) / 8;
  return want < 1024 ? (want < 1 ? 1 : want) : 1024;
}

extern "C" size_t ls3d_layer_norm_workspace_bytes(int n, int c) { return (size_t)ln_blocks(n) * 2 * (c > 0 ? c : 1) * sizeof(float) + 256; }

extern "C" int ls3d_layer_norm_forward(const float *x, int n, int c, const float *gamma, const float *beta, float eps, float *y, float *stats,
                                       ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (n == 0 && c >= 4) return LS3D_OK;
  if (!x || !gamma || !beta || !y || n < 0 || c < 4 || (c & 3)) return LS3D_ERR_ARG;
  if (c > 128 * LN_MAXV) return LS3D_ERR_UNSUPPORTED;
  if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return LS3D_ERR_ARG;
  if (c <= 128)
    hipLaunchKernelGGL((k_ln_fwd<1>), dim3(ln_blocks(n) * 2), dim3(256), 0, stream, x, n, c, gamma, beta, eps, y, stats);
  else
    hipLaunchKernelGGL((k_ln_fwd<2>), dim3(ln_blocks(n) * 2), dim3(256), 0, stream, x, n, c, gamma, beta, eps, y, stats);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_layer_norm_backward(const float *x, const float *dy, const float *gamma, const float *stats, int n, int c, float *dx, float *dgamma,
                                        float *dbeta, void *workspace, size_t workspace_bytes, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!gamma || !dgamma || !dbeta || n < 0 || c < 4 || (c & 3)) return LS3D_ERR_ARG;
  if (c > 128 * LN_MAXV) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) {
    if (hipMemsetAsync(dgamma, 0, c * sizeof(float), stream) != hipSuccess || hipMemsetAsync(dbeta, 0, c * sizeof(float), stream) != hipSuccess)
      return LS3D_ERR_LAUNCH;
    return LS3D_OK;
  }
  if (!x || !dy || !stats || !dx || !workspace) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_layer_norm_workspace_bytes(n, c) || ((uintptr_t)workspace & 15)) return LS3D_ERR_WORKSPACE;
  if (((uintptr_t)x & 15) || ((uintptr_t)dy & 15) || ((uintptr_t)dx & 15) || ((uintptr_t)gamma & 15)) return LS3D_ERR_ARG;
  const int nb = ln_blocks(n);
  float *partial = (float *)workspace;
  if (c <= 128)
    hipLaunchKernelGGL((k_ln_bwd<1>), dim3(nb), dim3(256), 0, stream, x, dy, gamma, stats, n, c, dx, partial);
  else
    hipLaunchKernelGGL((k_ln_bwd<2>), dim3(nb), dim3(256), 0, stream, x, dy, gamma, stats, n, c, dx, partial);
  hipLaunchKernelGGL(k_ln_bwd_reduce, dim3((2 * c + 31) / 32), dim3(256), 0, stream, (const float *)partial, nb, c, dgamma, dbeta);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

// ------------------------------------------------------------------------------------------------------------------------------------
// BatchNorm1d over [n, c] rows in TRAINING mode (batch statistics), forward and backward, with the ReLU and the residual add that follow it
// in the UNet fused in (det3d/models/backbones/scn_unet.py:11-69: SubMConv3d -> BatchNorm1d(eps 1e-3, momentum 0.01) -> ReLU, and the
// SparseBasicBlock's relu(bn2(conv2(.)) + identity)); the point heads' Linear -> BatchNorm1d -> ReLU chains use the same entry points.
// torch runs this as collect_statistics + transform (+ add + relu) forward and backward_reduce + backward_elemt (+ relu / add backward):
// 13 ms of BatchNorm kernels and ~6 ms of elementwise kernels per 2 x 180k-point Waymo step.  Here:
//   statistics: one pass over x.  A workgroup owns BN_ROWS consecutive rows: column sums -> the block's mean, then the centred second moment
//     around it from a second walk over the same rows (they are in L2: one HBM read), and the partials (mean_b, M2_b) of the row blocks are
//     merged per column with the pairwise update of Chan et al. in block order: deterministic, no E[x^2] - mean^2 cancellation;
//   apply:      y = [relu]((x - mean) * rstd * gamma + beta [+ res]);
//   backward:   g = dy * [y > 0]; column sums of g and g * xhat per row block -> fixed-order reduction; dx = gamma rstd (g - sum g / N - xhat
//     sum(g xhat) / N), dres = g.  The two sums are exposed between the launches so that a data-parallel step can all-reduce them
//     (count-weighted SyncBN, lidarseg3d_amd/syncbn.py).
constexpr int BN_ROWS = 512;  // rows per workgroup of the reductions

// thread (rg, cg): column group cg (4 channels), rows rg, rg + RP, ... of the block; c4 = c / 4 <= 64 column groups, RP = 256 / c4 row lanes
__global__ __launch_bounds__(256) void k_bn_stats_part(const float *__restrict__ x, int ld, int n, int c, float *__restrict__ part) {
  __shared__ float4 s_red[256];
  __shared__ float4 s_mean[64];
  const int c4 = c >> 2, RP = 256 / c4, tid = threadIdx.x;
  const int cg = tid % c4, rg = tid / c4;
  const int r0 = blockIdx.x * BN_ROWS, r1 = min(n, r0 + BN_ROWS);
  const bool on = rg < RP;
  // (four rows per trip, their loads issued together from clamped addresses: a lane walking its rows with one load in flight is bound by
  // the memory latency, not the bandwidth - round 3's BatchNorm attempt lost to torch for that reason)
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on)
    for (int r = r0 + rg; r < r1; r += 4 * RP) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(x + (size_t)min(r + u * RP, r1 - 1) * ld + cg * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * RP < r1) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
  s_red[tid] = s;
  __syncthreads();
  if (tid < c4) {
    float4 t = s_red[tid];
    for (int g = 1; g < RP; ++g) { const float4 u = s_red[g * c4 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    const float inv = 1.0f / (float)(r1 - r0);
    s_mean[tid] = make_float4(t.x * inv, t.y * inv, t.z * inv, t.w * inv);
  }
  __syncthreads();
  const float4 mu = s_mean[cg];
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
  if (on)
    for (int r = r0 + rg; r < r1; r += 4 * RP) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *(const float4 *)(x + (size_t)min(r + u * RP, r1 - 1) * ld + cg * 4);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * RP < r1) {
          const float dx = v[u].x - mu.x, dy = v[u].y - mu.y, dz = v[u].z - mu.z, dw = v[u].w - mu.w;
          q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
        }
    }
  s_red[tid] = q;
  __syncthreads();
  if (tid < c4) {
    float4 t = s_red[tid];
    for (int g = 1; g < RP; ++g) { const float4 u = s_red[g * c4 + tid]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *(float4 *)(part + ((size_t)blockIdx.x * 2 + 0) * c + tid * 4) = mu;
    *(float4 *)(part + ((size_t)blockIdx.x * 2 + 1) * c + tid * 4) = t;
  }
}

// (mean_b, M2_b, n_b) of the row blocks -> mean[c], M2[c] of all n rows: Chan's pairwise update in a FIXED-SHAPE tree - 32 chains per column
// (chain l merges blocks l, l + 32, ... in order), then five pairwise levels through LDS.  Deterministic (the shape depends on nblocks only),
// merged in double.  Round 4's version walked all row blocks of a column with ONE thread: up to 703 dependent double divisions = 120 us per
// call, 5x its own k_bn_stats_part, 5 ms of a Waymo training step.  8 columns per workgroup.
__device__ __forceinline__ void bn_chan_merge(double &mean, double &m2, double &cnt, double mb, double qb, double nb) {
  const double tot = cnt + nb;
  if (tot > 0.0) {
    const double d = mb - mean;
    mean += d * (nb / tot);
    m2 += qb + d * d * (cnt * nb / tot);
    cnt = tot;
  }
}

__global__ __launch_bounds__(256) void k_bn_stats_merge(const float *__restrict__ part, int nblocks, int n, int c, float *__restrict__ out) {
  __shared__ double s_mean[256], s_m2[256], s_cnt[256];
  const int tid = threadIdx.x, lane = tid & 31, col = blockIdx.x * 8 + (tid >> 5);
  double mean = 0.0, m2 = 0.0, cnt = 0.0;
  if (col < c)
    for (int b = lane; b < nblocks; b += 32)
      bn_chan_merge(mean, m2, cnt, (double)part[((size_t)b * 2 + 0) * c + col], (double)part[((size_t)b * 2 + 1) * c + col],
                    (double)min(BN_ROWS, n - b * BN_ROWS));
  s_mean[tid] = mean; s_m2[tid] = m2; s_cnt[tid] = cnt;
  __syncthreads();
  for (int d = 16; d >= 1; d >>= 1) {
    if (lane < d) {
      bn_chan_merge(mean, m2, cnt, s_mean[tid + d], s_m2[tid + d], s_cnt[tid + d]);
      s_mean[tid] = mean; s_m2[tid] = m2; s_cnt[tid] = cnt;
    }
    __syncthreads();
  }
  if (lane == 0 && col < c) {
    out[col] = (float)mean;
    out[c + col] = (float)m2;
  }
  if (blockIdx.x == 0 && tid == 0) out[2 * c] = (float)n;  // the (mean, M2, n) triple a data-parallel step gathers over the ranks
}

// The ranks' (mean, M2, n) triples -> the batch statistics of all rows, rstd, the total row count and nn.BatchNorm's running statistics, in
// ONE launch (round 4 composed this from ~14 small torch kernels per layer - and two more for rstd, six for the running statistics: ~700 of
// the ~1480 elementwise launches of a Waymo training step).  Chan's update over the ranks in rank order, in double: the same on every rank.
__global__ __launch_bounds__(256) void k_bn_finalize(const float *__restrict__ parts, int world, int c, int local_n, float eps, float momentum,
                                                     float *__restrict__ running_mean, float *__restrict__ running_var, long long *num_batches,
                                                     float *__restrict__ mean_out, float *__restrict__ var_out, float *__restrict__ rstd_out, float *count_out) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= c) return;
  const int stride = 2 * c + 1;
  double mean = 0.0, m2 = 0.0, cnt = 0.0;
  for (int r = 0; r < world; ++r) {
    const float *p = parts + (size_t)r * stride;
    const double nb = (world == 1 && local_n >= 0) ? (double)local_n : (double)p[2 * c];
    bn_chan_merge(mean, m2, cnt, (double)p[col], (double)p[c + col], nb);
  }
  const double tot = cnt > 1.0 ? cnt : 1.0;  // every rank empty: statistics 0 / eps, no NaN
  const double var = m2 / tot;
  mean_out[col] = (float)mean;
  var_out[col] = (float)var;
  rstd_out[col] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean) {
    running_mean[col] = (float)((1.0 - (double)momentum) * (double)running_mean[col] + (double)momentum * mean);
    running_var[col] = (float)((1.0 - (double)momentum) * (double)running_var[col] + (double)momentum * var * (tot / (tot - 1.0 > 1.0 ? tot - 1.0 : 1.0)));
  }
  if (col == 0) {
    count_out[0] = (float)tot;
    if (num_batches) num_batches[0] += 1;
  }
}

__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ x, int ld, int n, int c, const float *__restrict__ mean, const float *__restrict__ rstd,
                                                  const float *__restrict__ gamma, const float *__restrict__ beta, const float *__restrict__ res, int res_ld, int relu,
                                                  float *__restrict__ y, int y_ld) {
  const int c4 = c >> 2;
  const long long work = (long long)n * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(t / c4), cg = (int)(t % c4);
    const float4 v = *(const float4 *)(x + (size_t)r * ld + cg * 4), mu = *(const float4 *)(mean + cg * 4), rs = *(const float4 *)(rstd + cg * 4);
    const float4 g = *(const float4 *)(gamma + cg * 4), b = *(const float4 *)(beta + cg * 4);
    float4 o;
    o.x = (v.x - mu.x) * rs.x * g.x + b.x; o.y = (v.y - mu.y) * rs.y * g.y + b.y;
    o.z = (v.z - mu.z) * rs.z * g.z + b.z; o.w = (v.w - mu.w) * rs.w * g.w + b.w;
    if (res) {
      const float4 q = *(const float4 *)(res + (size_t)r * res_ld + cg * 4);
      o.x += q.x; o.y += q.y; o.z += q.z; o.w += q.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    *(float4 *)(y + (size_t)r * y_ld + cg * 4) = o;
  }
}

// per row block: column sums of g = dy * [y > 0] and of g * xhat -> part[block][2][c]
__global__ __launch_bounds__(256) void k_bn_bwd_part(const float *__restrict__ x, int ld, const float *__restrict__ dy, const float *__restrict__ y, int n, int c,
                                                     const float *__restrict__ mean, const float *__restrict__ rstd, float *__restrict__ part) {
  __shared__ float4 s_red[2][256];
  const int c4 = c >> 2, RP = 256 / c4, tid = threadIdx.x;
  const int cg = tid % c4, rg = tid / c4;
  const int r0 = blockIdx.x * BN_ROWS, r1 = min(n, r0 + BN_ROWS);
  float4 sg = make_float4(0.f, 0.f, 0.f, 0.f), sx = make_float4(0.f, 0.f, 0.f, 0.f);
  if (rg < RP) {
    const float4 mu = *(const float4 *)(mean + cg * 4), rs = *(const float4 *)(rstd + cg * 4);
    for (int r = r0 + rg; r < r1; r += 2 * RP) {
      float4 v[2], gg[2], oo[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const size_t rr = (size_t)min(r + u * RP, r1 - 1);
        v[u] = *(const float4 *)(x + rr * ld + cg * 4);
        gg[u] = *(const float4 *)(dy + rr * c + cg * 4);
        oo[u] = y ? *(const float4 *)(y + rr * c + cg * 4) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (r + u * RP < r1) {
          float4 g = gg[u];
          g.x = oo[u].x > 0.f ? g.x : 0.f; g.y = oo[u].y > 0.f ? g.y : 0.f; g.z = oo[u].z > 0.f ? g.z : 0.f; g.w = oo[u].w > 0.f ? g.w : 0.f;
          sg.x += g.x; sg.y += g.y; sg.z += g.z; sg.w += g.w;
          sx.x = fmaf(g.x, (v[u].x - mu.x) * rs.x, sx.x); sx.y = fmaf(g.y, (v[u].y - mu.y) * rs.y, sx.y);
          sx.z = fmaf(g.z, (v[u].z - mu.z) * rs.z, sx.z); sx.w = fmaf(g.w, (v[u].w - mu.w) * rs.w, sx.w);
        }
    }
  }
  s_red[0][tid] = sg;
  s_red[1][tid] = sx;
  __syncthreads();
  if (tid < 2 * c4) {
    const int which = tid / c4, k = tid % c4;
    float4 t = s_red[which][k];
    for (int g = 1; g < RP; ++g) { const float4 u = s_red[which][g * c4 + k]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
    *(float4 *)(part + ((size_t)blockIdx.x * 2 + which) * c + k * 4) = t;
  }
}

// sums[2 c] = (sum g, sum g xhat) over all rows, ALREADY divided by nothing: dx = gamma rstd (g - sums[0] / N - xhat sums[1] / N); dres = g
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float *__restrict__ x, int ld, const float *__restrict__ dy, const float *__restrict__ y, int n, int c,
                                                      const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ gamma,
                                                      const float *__restrict__ sums, float inv_count_host, const float *count_dev, float *__restrict__ dx,
                                                      float *__restrict__ dres) {
  const int c4 = c >> 2;
  const float inv_count = count_dev ? 1.0f / fmaxf(count_dev[0], 1.0f) : inv_count_host;  // the total row count of all ranks stays on the device
  const long long work = (long long)n * c4;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int r = (int)(t / c4), cg = (int)(t % c4);
    const float4 v = *(const float4 *)(x + (size_t)r * ld + cg * 4), mu = *(const float4 *)(mean + cg * 4), rs = *(const float4 *)(rstd + cg * 4);
    const float4 gm = *(const float4 *)(gamma + cg * 4), s1 = *(const float4 *)(sums + cg * 4), s2 = *(const float4 *)(sums + c + cg * 4);
    float4 g = *(const float4 *)(dy + (size_t)r * c + cg * 4);
    if (y) {
      const float4 o = *(const float4 *)(y + (size_t)r * c + cg * 4);
      g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
    }
    if (dres) *(float4 *)(dres + (size_t)r * c + cg * 4) = g;
    float4 o;
    o.x = gm.x * rs.x * (g.x - s1.x * inv_count - (v.x - mu.x) * rs.x * (s2.x * inv_count));
    o.y = gm.y * rs.y * (g.y - s1.y * inv_count - (v.y - mu.y) * rs.y * (s2.y * inv_count));
    o.z = gm.z * rs.z * (g.z - s1.z * inv_count - (v.z - mu.z) * rs.z * (s2.z * inv_count));
    o.w = gm.w * rs.w * (g.w - s1.w * inv_count - (v.w - mu.w) * rs.w * (s2.w * inv_count));
    *(float4 *)(dx + (size_t)r * c + cg * 4) = o;
  }
}

// per row block: column sums of x -> part[block][c] (a Linear layer's bias gradient: the sum of grad_out over 10^5 - 10^6 point rows, which
// torch's reduce serves at 117 us per call on [360 000, 64] and 0.9 ms on [360 000, 23]; same blocks, same fixed tree as the BatchNorm sums:
// deterministic).  VEC = 4: float4 loads (c % 4 == 0, aligned rows); VEC = 1: any c <= 256 (the class logits' 17 / 20 / 23 columns).
template <int VEC>
__global__ __launch_bounds__(256) void k_colsum_part(const float *__restrict__ x, int ld, int n, int c, float *__restrict__ part) {
  __shared__ float s_red[256][VEC];
  const int cv = c / VEC, RP = 256 / cv, tid = threadIdx.x;
  const int cg = tid % cv, rg = tid / cv;
  const int r0 = blockIdx.x * BN_ROWS, r1 = min(n, r0 + BN_ROWS);
  float s[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) s[e] = 0.f;
  if (rg < RP) {
    for (int r = r0 + rg; r < r1; r += 4 * RP) {
      float v[4][VEC];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float *src = x + (size_t)min(r + u * RP, r1 - 1) * ld + cg * VEC;
        if constexpr (VEC == 4) {
          const float4 t = *(const float4 *)src;
          v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w;
        } else {
          v[u][0] = *src;
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (r + u * RP < r1) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) s[e] += v[u][e];
        }
    }
  }
#pragma unroll
  for (int e = 0; e < VEC; ++e) s_red[tid][e] = s[e];
  __syncthreads();
  if (tid < cv) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      float t = s_red[tid][e];
      for (int g = 1; g < RP; ++g) t += s_red[g * cv + tid][e];
      part[(size_t)blockIdx.x * c + tid * VEC + e] = t;
    }
  }
}

// out[c] = the row blocks' partials [block][c] added: a workgroup owns 32 columns, its 8 groups each add every 8th block in order, then the 8 sums in order
__global__ __launch_bounds__(256) void k_colsum_reduce(const float *__restrict__ partial, int nblocks, int c, float *__restrict__ out) {
  __shared__ float red[8][32];
  const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + lane;
  float s = 0.0f;
  if (i < c) {
#pragma unroll 8
    for (int b = grp; b < nblocks; b += 8) s += partial[(size_t)b * c + i];
  }
  red[grp][lane] = s;
  __syncthreads();
  if (grp == 0 && i < c) {
    float t = red[0][lane];
#pragma unroll
    for (int g = 1; g < 8; ++g) t += red[g][lane];
    out[i] = t;
  }
}

static inline int bn_blocks(int n) { return n > 0 ? (n + BN_ROWS - 1) / BN_ROWS : 1; }
static inline bool bn_shape_ok(int c) { return c >= 4 && !(c & 3) && c <= 256 && (256 % (c >> 2)) == 0; }

extern "C" size_t ls3d_batch_norm_workspace_bytes(int n, int c) { return (size_t)bn_blocks(n) * 2 * (c > 0 ? c : 1) * sizeof(float) + 256; }

extern "C" int ls3d_batch_norm_stats(const float *x, int ld, int n, int c, void *workspace, size_t workspace_bytes, float *mean_m2, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!mean_m2 || n < 0 || c < 4 || (c & 3)) return LS3D_ERR_ARG;
  if (!bn_shape_ok(c)) return LS3D_ERR_UNSUPPORTED;
  if (n == 0) return hipMemsetAsync(mean_m2, 0, (2 * c + 1) * sizeof(float), stream) == hipSuccess ? LS3D_OK : LS3D_ERR_LAUNCH;
  if (!x || !workspace || ld < c || (ld & 3) || ((uintptr_t)x & 15) || ((uintptr_t)workspace & 15)) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_batch_norm_workspace_bytes(n, c)) return LS3D_ERR_WORKSPACE;
  const int nb = bn_blocks(n);
  hipLaunchKernelGGL(k_bn_stats_part, dim3(nb), dim3(256), 0, stream, x, ld, n, c, (float *)workspace);
  hipLaunchKernelGGL(k_bn_stats_merge, dim3((c + 7) / 8), dim3(256), 0, stream, (const float *)workspace, nb, n, c, mean_m2);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_batch_norm_finalize(const float *parts, int world, int c, int local_n, float eps, float momentum, float *running_mean,
                                        float *running_var, int64_t *num_batches_tracked, float *mean, float *var, float *rstd, float *count_out,
                                        ls3d_stream_t stream_) {
  if (!parts || !mean || !var || !rstd || !count_out || world < 1 || c < 1) return LS3D_ERR_ARG;
  if ((running_mean != nullptr) != (running_var != nullptr)) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_bn_finalize, dim3((c + 255) / 256), dim3(256), 0, (hipStream_t)stream_, parts, world, c, local_n, eps, momentum, running_mean, running_var,
                     (long long *)num_batches_tracked, mean, var, rstd, count_out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" size_t ls3d_column_sums_workspace_bytes(int n, int c) { return (size_t)bn_blocks(n) * (c > 0 ? c : 1) * sizeof(float) + 256; }

// out[c] = sum over the n rows of x[n, c] (row stride ld): row blocks, then a fixed tree over the blocks' partials
extern "C" int ls3d_column_sums(const float *x, int ld, int n, int c, void *workspace, size_t workspace_bytes, float *out, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!out || n < 0 || c < 1 || c > 256) return LS3D_ERR_ARG;
  if (n == 0) return hipMemsetAsync(out, 0, c * sizeof(float), stream) == hipSuccess ? LS3D_OK : LS3D_ERR_LAUNCH;
  if (!x || !workspace || ld < c) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_column_sums_workspace_bytes(n, c)) return LS3D_ERR_WORKSPACE;
  const int nb = bn_blocks(n);
  if (!(c & 3) && !(ld & 3) && !((uintptr_t)x & 15))
    hipLaunchKernelGGL((k_colsum_part<4>), dim3(nb), dim3(256), 0, stream, x, ld, n, c, (float *)workspace);
  else
    hipLaunchKernelGGL((k_colsum_part<1>), dim3(nb), dim3(256), 0, stream, x, ld, n, c, (float *)workspace);
  hipLaunchKernelGGL(k_colsum_reduce, dim3((c + 31) / 32), dim3(256), 0, stream, (const float *)workspace, nb, c, out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_batch_norm_apply(const float *x, int ld, int n, int c, const float *mean, const float *rstd, const float *gamma, const float *beta,
                                     const float *res, int res_ld, int relu, float *y, int y_ld, ls3d_stream_t stream_) {
  if (n == 0 && bn_shape_ok(c)) return LS3D_OK;
  if (!x || !mean || !rstd || !gamma || !beta || !y || n < 0 || !bn_shape_ok(c) || ld < c || (ld & 3) || y_ld < c || (y_ld & 3) || (res && (res_ld < c || (res_ld & 3))))
    return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_bn_apply, ls3d_grid((long long)n * (c >> 2)), dim3(256), 0, (hipStream_t)stream_, x, ld, n, c, mean, rstd, gamma, beta, res, res_ld, relu, y, y_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_batch_norm_backward_sums(const float *x, int ld, const float *dy, const float *y_or_null, int n, int c, const float *mean, const float *rstd,
                                             void *workspace, size_t workspace_bytes, float *sums, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!sums || n < 0 || !bn_shape_ok(c)) return LS3D_ERR_ARG;
  if (n == 0) return hipMemsetAsync(sums, 0, 2 * c * sizeof(float), stream) == hipSuccess ? LS3D_OK : LS3D_ERR_LAUNCH;
  if (!x || !dy || !mean || !rstd || !workspace || ld < c || (ld & 3)) return LS3D_ERR_ARG;
  if (workspace_bytes < ls3d_batch_norm_workspace_bytes(n, c)) return LS3D_ERR_WORKSPACE;
  const int nb = bn_blocks(n);
  hipLaunchKernelGGL(k_bn_bwd_part, dim3(nb), dim3(256), 0, stream, x, ld, dy, y_or_null, n, c, mean, rstd, (float *)workspace);
  hipLaunchKernelGGL(k_ln_bwd_reduce, dim3((2 * c + 31) / 32), dim3(256), 0, stream, (const float *)workspace, nb, c, sums, sums + c);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_batch_norm_backward_apply(const float *x, int ld, const float *dy, const float *y_or_null, int n, int c, const float *mean, const float *rstd,
                                              const float *gamma, const float *sums, float inv_count, const float *count_dev, float *dx, float *dres,
                                              ls3d_stream_t stream_) {
  if (n == 0 && bn_shape_ok(c)) return LS3D_OK;
  if (!x || !dy || !mean || !rstd || !gamma || !sums || !dx || n < 0 || !bn_shape_ok(c) || ld < c || (ld & 3)) return LS3D_ERR_ARG;
  hipLaunchKernelGGL(k_bn_bwd_apply, ls3d_grid((long long)n * (c >> 2)), dim3(256), 0, (hipStream_t)stream_, x, ld, dy, y_or_null, n, c, mean, rstd, gamma, sums,
                     inv_count, count_dev, dx, dres);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
