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
