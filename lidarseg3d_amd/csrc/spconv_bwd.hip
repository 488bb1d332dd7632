// spconv_bwd.hip — weight gradient of the sparse convolutions (spconv v1.x's indice_conv_backward,
// call sites det3d/models/backbones/scn_unet.py through autograd).
//
//   forward   out[o]      = sum_k W[k]^T in[tbl[o][k]]
//   dgrad     grad_in[i]  = sum_k W[k] grad_out[tblT[i][k]]      -> the forward gather-GEMM on the transposed table with
//                                                                   transposed (SubM: also mirrored) weights, no new kernel
//   wgrad     grad_W[k]   = sum_{o : tbl[o][k] >= 0} in[tbl[o][k]] (x) grad_out[o]          (this file)
//
// wgrad on CDNA4: for one kernel offset k the sum over output rows is a GEMM with the ROWS as the reduction dimension, so
// two rows feed one v_mfma_f32_32x32x2_f32: lane (i, half) supplies in[tbl[o_half][k]][ci0 + i] as the A operand and
// grad_out[o_half][co0 + i] as the B operand (32 consecutive floats of a row per half wave: coalesced 128-byte reads, no
// LDS staging).  A wave owns a 32 x (32 COB) block of grad_W[k] in COB accumulators and walks its share of the rows in
// mask-sorted order, skipping row pairs without a neighbour at k; the 4 waves of a workgroup take interleaved row pairs
// and are summed through LDS in a fixed order; row chunks are summed by a second kernel: deterministic, no atomics.
// Exact f32 (fmaf chain per element); bound by the f32 matrix pipe like the forward (same flops).
#include "common.h"

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_UNROLL = 4;  // row pairs in flight per wave

template <int COB>
__global__ __launch_bounds__(256) void k_spconv_wgrad(const float *__restrict__ in, int in_ld, const float *__restrict__ gout, int go_ld,
                                                      const int32_t *__restrict__ tbl, const int32_t *__restrict__ order, int kvol, int cin,
                                                      int cout, int n_rows, const int32_t *n_rows_dev, int nchunks, float *__restrict__ partial) {
  __shared__ float red[3][32 * 32 * COB];  // partial tiles of waves 1..3
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = lane & 31, half = lane >> 5;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ci_blocks = (cin + 31) / 32;
  const int k = blockIdx.y;
  const int chunk = blockIdx.x / ci_blocks, cb = blockIdx.x % ci_blocks;
  const int ci = cb * 32 + i;
  const int npairs = (N + 1) / 2;
  const int per_chunk = (npairs + nchunks - 1) / nchunks;
  const int p0 = chunk * per_chunk, p1 = min(npairs, p0 + per_chunk);
  wg_f32x16 acc[COB];
#pragma unroll
  for (int n = 0; n < COB; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
  for (int p = p0 + wave * WG_UNROLL; p < p1; p += 4 * WG_UNROLL) {
    int o[WG_UNROLL], idx[WG_UNROLL];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      const int r = 2 * (p + u) + half;
      o[u] = (p + u < p1 && r < N) ? (order ? order[r] : r) : -1;
      idx[u] = o[u] >= 0 ? tbl[(size_t)o[u] * kvol + k] : -1;
    }
    float a[WG_UNROLL], b[WG_UNROLL][COB];
    bool any[WG_UNROLL];
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      any[u] = __any(idx[u] >= 0);
      const bool on = idx[u] >= 0;
      a[u] = (on && ci < cin) ? in[(size_t)idx[u] * in_ld + ci] : 0.0f;
#pragma unroll
      for (int n = 0; n < COB; ++n) b[u][n] = (on && n * 32 + i < cout) ? gout[(size_t)o[u] * go_ld + n * 32 + i] : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < WG_UNROLL; ++u) {
      if (any[u]) {
#pragma unroll
        for (int n = 0; n < COB; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u][n], acc[n], 0, 0, 0);
      }
    }
  }
  // ---- waves 1..3 hand their tiles to wave 0 (fixed order), which writes the chunk's partial [32][cout] block
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < COB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[wave - 1][(n * 16 + r) * 64 + lane] = acc[n][r];
  }
  __syncthreads();
  if (wave == 0) {
    float *dst = partial + (((size_t)chunk * kvol + k) * cin) * cout;
#pragma unroll
    for (int n = 0; n < COB; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[n][r];
        v += red[0][(n * 16 + r) * 64 + lane];
        v += red[1][(n * 16 + r) * 64 + lane];
        v += red[2][(n * 16 + r) * 64 + lane];
        const int row = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, col = n * 32 + i;  // fragment layout of the 32x32 MFMA
        if (row < cin && col < cout) dst[(size_t)row * cout + col] = v;
      }
  }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *partial, int nchunks, long long elems, float *gw) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < elems; t += (long long)gridDim.x * blockDim.x) {
    float s = 0.0f;
    for (int c = 0; c < nchunks; ++c) s += partial[(size_t)c * elems + t];
    gw[t] = s;
  }
}

static inline int wg_chunks(int n_rows, int kvol, int cin) {
  // enough workgroups for 256 CUs x ~8, at least ~256 row pairs per chunk
  const int ci_blocks = (cin + 31) / 32;
  long long want = (2048 + (long long)kvol * ci_blocks - 1) / ((long long)kvol * ci_blocks);
  long long cap = ((long long)n_rows / 2 + 255) / 256;
  if (want > cap) want = cap;
  if (want < 1) want = 1;
  if (want > 64) want = 64;
  return (int)want;
}

extern "C" size_t ls3d_spconv_wgrad_workspace_bytes(int kvol, int cin, int cout, int n_rows) {
  return (size_t)wg_chunks(n_rows, kvol, cin) * kvol * cin * cout * sizeof(float) + 256;
}

extern "C" int ls3d_spconv_wgrad(const float *in, int in_ld, const float *grad_out, int go_ld, const int32_t *tbl, const int32_t *row_order,
                                 int kvol, int cin, int cout, int n_rows, const int32_t *n_rows_dev, void *workspace, size_t workspace_bytes,
                                 float *grad_w, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !grad_out || !tbl || !grad_w || !workspace || kvol < 1 || cin < 1 || cout < 1 || n_rows < 0) return LS3D_ERR_ARG;
  if (in_ld < cin || go_ld < cout) return LS3D_ERR_ARG;
  if (cout > 128) return LS3D_ERR_UNSUPPORTED;
  if (workspace_bytes < ls3d_spconv_wgrad_workspace_bytes(kvol, cin, cout, n_rows)) return LS3D_ERR_WORKSPACE;
  const long long elems = (long long)kvol * cin * cout;
  if (n_rows == 0) {
    hipMemsetAsync(grad_w, 0, (size_t)elems * sizeof(float), stream);
    return LS3D_OK;
  }
  const int nchunks = wg_chunks(n_rows, kvol, cin);
  const int ci_blocks = (cin + 31) / 32;
  float *partial = (float *)workspace;
  const dim3 grid((unsigned)(nchunks * ci_blocks), (unsigned)kvol);
  const int cob = (cout + 31) / 32;
  if (cob == 1)
    hipLaunchKernelGGL((k_spconv_wgrad<1>), grid, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, tbl, row_order, kvol, cin, cout, n_rows,
                       n_rows_dev, nchunks, partial);
  else if (cob == 2)
    hipLaunchKernelGGL((k_spconv_wgrad<2>), grid, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, tbl, row_order, kvol, cin, cout, n_rows,
                       n_rows_dev, nchunks, partial);
  else
    hipLaunchKernelGGL((k_spconv_wgrad<4>), grid, dim3(256), 0, stream, in, in_ld, grad_out, go_ld, tbl, row_order, kvol, cin, cout, n_rows,
                       n_rows_dev, nchunks, partial);
  hipLaunchKernelGGL(k_wgrad_reduce, ls3d_grid(elems), dim3(256), 0, stream, (const float *)partial, nchunks, elems, grad_w);
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
