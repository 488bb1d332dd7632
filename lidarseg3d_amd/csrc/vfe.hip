// vfe.hip — voxel feature extractors (readers) and the small row-wise kernels around the GEMMs.
//
// Reference: det3d/models/readers/voxel_encoder.py — MeanVoxelFeatureExtractor (:51-58),
// ImprovedMeanVoxelFeatureExtractor (:74-124), TransformerVoxelFeatureExtractor (:202-270) with
// TransformerEncoderLayerPreNorm (:149-163).  The dense projections of the transformer run through
// ls3d_gather_gemm (MFMA); what lives here is the per-voxel descriptor, the 5-token attention core, the
// token max-pool and LayerNorm.  All of it is HBM-bound streaming (a few hundred bytes per voxel).
#include "common.h"

#include "vfe_descriptor.h"

__global__ __launch_bounds__(256) void k_vfe_mean(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C,
                                                 float *out, int out_ld) {
  const int N = ls3d_count(n, n_dev);
  const long long work = (long long)N * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(t / C), c = (int)(t % C);
    const float *vox = voxels + (size_t)v * P * C;
    float s = 0.0f;
    for (int p = 0; p < P; ++p) s += vox[p * C + c];
    out[(size_t)v * out_ld + c] = __fdiv_rn(s, (float)num[v]);
  }
}

__global__ __launch_bounds__(256) void k_vfe_improved(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C,
                                                     float *out, int out_ld) {
  const int N = ls3d_count(n, n_dev);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    float desc[LS3D_MAX_FEAT + 8];
    vfe_descriptor(voxels + (size_t)v * P * C, P, C, num[v], desc);
    float *o = out + (size_t)v * out_ld;
    for (int c = 0; c < C + 8; ++c) o[c] = desc[c];
    for (int c = C + 8; c < out_ld; ++c) o[c] = 0.0f;
  }
}

// tokens[v*P + p] = [point p of voxel v (C), descriptor (C+8), 0...]   (voxel_encoder.py:246-252)
__global__ __launch_bounds__(256) void k_vfe_tokens(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C,
                                                   float *tokens, int ld) {
  const int N = ls3d_count(n, n_dev);
  for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < N; v += gridDim.x * blockDim.x) {
    const float *vox = voxels + (size_t)v * P * C;
    float desc[LS3D_MAX_FEAT + 8];
    vfe_descriptor(vox, P, C, num[v], desc);
    for (int p = 0; p < P; ++p) {
      float *o = tokens + ((size_t)v * P + p) * ld;
      for (int c = 0; c < C; ++c) o[c] = vox[p * C + c];
      for (int c = 0; c < C + 8; ++c) o[C + c] = desc[c];
      for (int c = 2 * C + 8; c < ld; ++c) o[c] = 0.0f;
    }
  }
}

// softmax(q k^T / sqrt(hd)) v inside groups of `seq` rows; one thread per (row, head).
template <int HD>
__global__ __launch_bounds__(256) void k_mha_core(const float *qkv, int groups, const int32_t *groups_dev, int seq, int E, int H, float *out) {
  const int G = ls3d_count(groups, groups_dev);
  const long long work = (long long)G * seq * H;
  const float scale = 1.0f / sqrtf((float)HD);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const int h = (int)(t % H);
    const long long row = t / H;
    const long long g0 = (row / seq) * seq;
    const float *qp = qkv + (size_t)row * 3 * E + h * HD;
    float q[HD], acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = qp[d] * scale; acc[d] = 0.0f; }
    float m = -3.0e38f;
    for (int j = 0; j < seq; ++j) {
      const float *kp = qkv + (size_t)(g0 + j) * 3 * E + E + h * HD;
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(q[d], kp[d], s);
      m = fmaxf(m, s);
    }
    float den = 0.0f;
    for (int j = 0; j < seq; ++j) {
      const float *kp = qkv + (size_t)(g0 + j) * 3 * E + E + h * HD;
      const float *vp = kp + E;
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(q[d], kp[d], s);
      const float p = expf(s - m);
      den += p;
#pragma unroll
      for (int d = 0; d < HD; ++d) acc[d] = fmaf(p, vp[d], acc[d]);
    }
    float *op = out + (size_t)row * E + h * HD;
    const float inv = 1.0f / den;
#pragma unroll
    for (int d = 0; d < HD; ++d) op[d] = acc[d] * inv;
  }
}

__global__ __launch_bounds__(256) void k_group_max(const float *in, int groups, const int32_t *groups_dev, int seq, int C, float *out) {
  const int G = ls3d_count(groups, groups_dev);
  const long long work = (long long)G * C;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < work; t += (long long)gridDim.x * blockDim.x) {
    const long long g = t / C;
    const int c = (int)(t % C);
    float m = in[(size_t)(g * seq) * C + c];
    for (int j = 1; j < seq; ++j) m = fmaxf(m, in[(size_t)(g * seq + j) * C + c]);
    out[t] = m;
  }
}

// one wave per row, up to 4 elements per lane (C <= 256)
__global__ __launch_bounds__(256) void k_layernorm(const float *x, const float *res, const float *gamma, const float *beta, float eps,
                                                  int rows, const int32_t *rows_dev, int C, float *y) {
  const int R = ls3d_count(rows, rows_dev);
  const int lane = threadIdx.x & 63;
  const int wpb = blockDim.x >> 6;
  for (long long row = (long long)blockIdx.x * wpb + (threadIdx.x >> 6); row < R; row += (long long)gridDim.x * wpb) {
    float v[4];
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      float t = 0.0f;
      if (c < C) {
        t = x[(size_t)row * C + c];
        if (res) t += res[(size_t)row * C + c];
      }
      v[j] = t;
      s += t;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      const float d0 = (c < C) ? v[j] - mean : 0.0f;
      q += d0 * d0;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) q += __shfl_xor(q, d);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = lane + 64 * j;
      if (c < C) y[(size_t)row * C + c] = (v[j] - mean) * rstd * gamma[c] + beta[c];
    }
  }
}

extern "C" int ls3d_vfe_mean(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C, float *out, int out_ld,
                             ls3d_stream_t stream) {
  if (!voxels || !num || !out || n < 0 || P < 1 || C < 1 || out_ld < C) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_vfe_mean, ls3d_grid((long long)n * C), dim3(256), 0, (hipStream_t)stream, voxels, num, n, n_dev, P, C, out, out_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_vfe_improved_mean(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C, float *out,
                                      int out_ld, ls3d_stream_t stream) {
  if (!voxels || !num || !out || n < 0 || P < 1 || C < 3 || C > LS3D_MAX_FEAT || out_ld < C + 8) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_vfe_improved, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, voxels, num, n, n_dev, P, C, out, out_ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_vfe_tokens(const float *voxels, const int32_t *num, int n, const int32_t *n_dev, int P, int C, float *tokens, int ld,
                               ls3d_stream_t stream) {
  if (!voxels || !num || !tokens || n < 0 || P < 1 || C < 3 || C > LS3D_MAX_FEAT || ld < 2 * C + 8) return LS3D_ERR_ARG;
  if (n == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_vfe_tokens, ls3d_grid(n), dim3(256), 0, (hipStream_t)stream, voxels, num, n, n_dev, P, C, tokens, ld);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_mha_core(const float *qkv, int groups, const int32_t *groups_dev, int seq, int E, int H, float *out,
                             ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!qkv || !out || groups < 0 || seq < 1 || H < 1 || E % H) return LS3D_ERR_ARG;
  if (groups == 0) return LS3D_OK;
  const dim3 grid = ls3d_grid((long long)groups * seq * H);
  switch (E / H) {
    case 8: hipLaunchKernelGGL((k_mha_core<8>), grid, dim3(256), 0, stream, qkv, groups, groups_dev, seq, E, H, out); break;
    case 16: hipLaunchKernelGGL((k_mha_core<16>), grid, dim3(256), 0, stream, qkv, groups, groups_dev, seq, E, H, out); break;
    case 24: hipLaunchKernelGGL((k_mha_core<24>), grid, dim3(256), 0, stream, qkv, groups, groups_dev, seq, E, H, out); break;
    case 32: hipLaunchKernelGGL((k_mha_core<32>), grid, dim3(256), 0, stream, qkv, groups, groups_dev, seq, E, H, out); break;
    default: return LS3D_ERR_UNSUPPORTED;
  }
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_group_max(const float *in, int groups, const int32_t *groups_dev, int seq, int C, float *out, ls3d_stream_t stream) {
  if (!in || !out || groups < 0 || seq < 1 || C < 1) return LS3D_ERR_ARG;
  if (groups == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_group_max, ls3d_grid((long long)groups * C), dim3(256), 0, (hipStream_t)stream, in, groups, groups_dev, seq, C, out);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}

extern "C" int ls3d_layernorm(const float *x, const float *res, const float *gamma, const float *beta, float eps, int rows,
                              const int32_t *rows_dev, int C, float *y, ls3d_stream_t stream) {
  if (!x || !gamma || !beta || !y || rows < 0 || C < 1 || C > 256) return LS3D_ERR_ARG;
  if (rows == 0) return LS3D_OK;
  hipLaunchKernelGGL(k_layernorm, ls3d_grid((long long)rows * 64), dim3(256), 0, (hipStream_t)stream, x, res, gamma, beta, eps, rows,
                     rows_dev, C, y);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
