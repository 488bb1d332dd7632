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

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct EpiDev {
  const float *scale, *shift, *res_pre, *pair;
  int res_pre_ld, pair_ld, relu;
};

template <int KC, int NT>
__global__ __launch_bounds__(256) void k_gather_gemm(const float *__restrict__ in, int in_ld, const int32_t *__restrict__ tbl, int kvol,
                                                    const float *__restrict__ w, int cin, int w_ld, int cout, int n_rows,
                                                    const int32_t *n_rows_dev, EpiDev e, float *__restrict__ out, int out_ld) {
  constexpr int SPL = KC / 2;    // floats of a row chunk held per lane
  constexpr int SLAB = NT * 32;  // output columns handled by this workgroup
  __shared__ float Bs[KC * SLAB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 31, kk = lane >> 5;
  const int n0 = blockIdx.y * SLAB;
  const int N = ls3d_count(n_rows, n_rows_dev);
  const int ntiles = (N + 127) / 128;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row = tile * 128 + wave * 32 + col;
    f32x16 acc[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][r] = 0.0f;
    for (int k = 0; k < kvol; ++k) {
      int idx = -1;
      if (row < N) idx = tbl ? tbl[(size_t)row * kvol + k] : row;
      if (!__syncthreads_or(idx >= 0)) continue;  // nobody in the tile has this neighbour
      const bool wave_any = __any(idx >= 0);
      const float *wk = w + (size_t)k * cin * w_ld + n0;
      for (int c0 = 0; c0 < cin; c0 += KC) {
        float a[SPL];
        if (idx >= 0) {
          const float4 *p = (const float4 *)(in + (size_t)idx * in_ld + c0 + kk * SPL);
#pragma unroll
          for (int q = 0; q < SPL / 4; ++q) {
            const float4 v = p[q];
            a[4 * q + 0] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < SPL; ++q) a[q] = 0.0f;
        }
        __syncthreads();  // every wave is done reading the previous chunk of Bs
        for (int i = tid; i < KC * (SLAB / 4); i += 256) {
          const int r = i / (SLAB / 4), c4 = i % (SLAB / 4);
          *(float4 *)(Bs + r * SLAB + c4 * 4) = *(const float4 *)(wk + (size_t)(c0 + r) * w_ld + c4 * 4);
        }
        __syncthreads();
        if (wave_any) {
#pragma unroll
          for (int s = 0; s < SPL; ++s) {
#pragma unroll
            for (int n = 0; n < NT; ++n) {
              const float b = Bs[(kk * SPL + s) * SLAB + n * 32 + col];
              acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], b, acc[n], 0, 0, 0);
            }
          }
        }
      }
    }
    // epilogue: acc register r of lane (col, kk) is output row (r&3) + 8*(r>>2) + 4*kk, column col
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int oc = n0 + n * 32 + col;
      if (oc >= cout) continue;
      const float sc = e.scale ? e.scale[oc] : 1.0f;
      const float sh = e.shift ? e.shift[oc] : 0.0f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = tile * 128 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (orow >= N) continue;
        float v = fmaf(acc[n][r], sc, sh);
        if (e.res_pre) v += e.res_pre[(size_t)orow * e.res_pre_ld + oc];
        if (e.relu) v = fmaxf(v, 0.0f);
        if (e.pair) {
          const float *pp = e.pair + (size_t)orow * e.pair_ld + 2 * oc;
          v += pp[0] + pp[1];
        }
        out[(size_t)orow * out_ld + oc] = v;
      }
    }
  }
}

template <int KC, int NT>
static void launch_gg(dim3 grid, hipStream_t stream, const float *in, int in_ld, const int32_t *tbl, int kvol, const float *w, int cin,
                      int w_ld, int cout, int n_rows, const int32_t *n_rows_dev, EpiDev e, float *out, int out_ld) {
  hipLaunchKernelGGL((k_gather_gemm<KC, NT>), grid, dim3(256), 0, stream, in, in_ld, tbl, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e,
                     out, out_ld);
}

extern "C" int ls3d_gather_gemm(const float *in, int in_ld, const int32_t *tbl, int kvol, const float *w, int cin, int cout, int n_rows,
                                const int32_t *n_rows_dev, const ls3d_epilogue_t *epi, float *out, int out_ld, ls3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!in || !w || !out || n_rows < 0 || kvol < 1 || cin < 16 || cout < 1) return LS3D_ERR_ARG;
  if ((cin % 16) || (in_ld % 4) || in_ld < cin || out_ld < cout) return LS3D_ERR_ARG;
  if (!tbl && kvol != 1) return LS3D_ERR_ARG;
  if (((uintptr_t)in & 15) || ((uintptr_t)w & 15)) return LS3D_ERR_ARG;
  if (n_rows == 0) return LS3D_OK;
  EpiDev e = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
  if (epi) {
    e.scale = epi->scale; e.shift = epi->shift; e.res_pre = epi->res_pre; e.pair = epi->pair;
    e.res_pre_ld = epi->res_pre_ld; e.pair_ld = epi->pair_ld; e.relu = epi->relu;
  }
  const int w_ld = (cout + 31) / 32 * 32;
  const int nt_total = w_ld / 32;
  int nt = 1;
  for (int c = 4; c >= 1; --c)
    if (nt_total % c == 0) { nt = c; break; }
  const int slabs = nt_total / nt;
  const int ntiles = (n_rows + 127) / 128;
  dim3 grid((unsigned)(ntiles < 2048 ? ntiles : 2048), (unsigned)slabs);
  const bool k32 = (cin % 32) == 0;
#define LS3D_GG(KC, NT) launch_gg<KC, NT>(grid, stream, in, in_ld, tbl, kvol, w, cin, w_ld, cout, n_rows, n_rows_dev, e, out, out_ld)
  if (k32) {
    switch (nt) { case 1: LS3D_GG(32, 1); break; case 2: LS3D_GG(32, 2); break; case 3: LS3D_GG(32, 3); break; default: LS3D_GG(32, 4); }
  } else {
    switch (nt) { case 1: LS3D_GG(16, 1); break; case 2: LS3D_GG(16, 2); break; case 3: LS3D_GG(16, 3); break; default: LS3D_GG(16, 4); }
  }
#undef LS3D_GG
  LS3D_RETURN_IF_LAUNCH_FAILED();
  return LS3D_OK;
}
