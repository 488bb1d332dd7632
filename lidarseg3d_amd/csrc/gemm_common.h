// gemm_common.h — pieces shared by the gather-GEMM kernels (spconv.hip) and the tile-halo convolution (tileconv.hip):
// MFMA fragment types, the fused epilogue, the exact f32 -> bf16-plane splits.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct EpiDev {
  const float *scale, *shift, *res_pre, *pair, *ln_gamma, *ln_beta;
  int res_pre_ld, pair_ld, relu;
  float ln_eps;
};

// ---- epilogue through LDS (shared by the f32 and split-bf16 kernels): the accumulators (fragment layout: register r of
//      lane (col,kk) = output row (r&3) + 8*(r>>2) + 4*kk, column col) are transposed into row-major tiles in the weight
//      buffer, then all NTH threads apply scale/shift, residual, ReLU, pair-sum, optionally a row LayerNorm (whole row in
//      this workgroup's slab), and store whole rows with float4.
//      CS = false: acc[n] = column block n of the wave's 32 rows (wave = row block wr, column slab wc of WC);
//      CS = true (column-stationary waves, k_tile_conv_cs; WC = 1): acc[i] = row block wr * NT + i, column block wc of the NT.
template <int NT, int WC, int TR, int RPP, int NTH = 256, bool CS = false>
__device__ __forceinline__ void gg_epilogue(f32x16 (&acc)[NT], float *stage, const int *s_rows, float *s_stat, int wr, int wc, int kk,
                                            int col, int n0, int cout, const EpiDev &e, float *__restrict__ out, int out_ld) {
  constexpr int WSLAB = NT * 32, SLAB = WSLAB * WC;
  const int tid = threadIdx.x;
  const bool vec = ((cout & 3) == 0) && ((out_ld & 3) == 0) && (!e.res_pre || (e.res_pre_ld & 3) == 0) && (!e.pair || (e.pair_ld & 3) == 0);
#pragma unroll
  for (int pass = 0; pass < TR / RPP; ++pass) {
    __syncthreads();  // previous readers of the buffer (MFMA loop or previous pass) are done
    if constexpr (CS) {
      static_assert(WC == 1, "column-stationary waves: one slab");
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        const int r0 = (wr * NT + i) * 32;
        if (r0 / RPP == pass) {
          float *dst = stage + (r0 % RPP + 4 * kk) * SLAB + wc * 32 + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * SLAB] = acc[i][r];
        }
      }
    } else if ((wr * 32) / RPP == pass) {
      float *dst = stage + ((wr * 32) % RPP + 4 * kk) * SLAB + wc * WSLAB + col;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2)) * SLAB + n * 32] = acc[n][r];
    }
    __syncthreads();
    if (e.ln_gamma && vec) {
      // LayerNorm epilogue, vectorised: a row's SLAB/4 float4s sit in LW consecutive lanes (LW = next power of two), so
      // mean and variance are two shuffle reductions; nothing goes back through LDS.
      constexpr int LW = (SLAB / 4 <= 8) ? 8 : (SLAB / 4 <= 16) ? 16 : 32;
      for (int i = tid; i < RPP * LW; i += NTH) {
        const int lr = i / LW, c4 = i % LW;
        const int orow = s_rows[pass * RPP + lr], oc = n0 + c4 * 4;
        const bool on = (c4 < SLAB / 4) && (oc < cout);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (on && orow >= 0) {
          v = *(const float4 *)(stage + lr * SLAB + c4 * 4);
          if (e.scale) {
            const float4 sc = *(const float4 *)(e.scale + oc);
            v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
          }
          if (e.shift) {
            const float4 sh = *(const float4 *)(e.shift + oc);
            v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w;
          }
          if (e.res_pre) {
            const float4 q = *(const float4 *)(e.res_pre + (size_t)orow * e.res_pre_ld + oc);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          if (e.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        }
        float sum = (v.x + v.y) + (v.z + v.w);
#pragma unroll
        for (int d = LW / 2; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        const float mean = sum / (float)cout;
        const float dx = on ? v.x - mean : 0.f, dy = on ? v.y - mean : 0.f, dz = on ? v.z - mean : 0.f, dw = on ? v.w - mean : 0.f;
        float q2 = (dx * dx + dy * dy) + (dz * dz + dw * dw);
#pragma unroll
        for (int d = LW / 2; d >= 1; d >>= 1) q2 += __shfl_xor(q2, d);
        const float rstd = 1.0f / sqrtf(q2 / (float)cout + e.ln_eps);
        if (on && orow >= 0) {
          const float4 g = *(const float4 *)(e.ln_gamma + oc), bt = *(const float4 *)(e.ln_beta + oc);
          float4 o;
          o.x = dx * rstd * g.x + bt.x; o.y = dy * rstd * g.y + bt.y; o.z = dz * rstd * g.z + bt.z; o.w = dw * rstd * g.w + bt.w;
          *(float4 *)(out + (size_t)orow * out_ld + oc) = o;
        }
      }
    } else if (e.ln_gamma) {
      // scalar fallback (unaligned leading dimensions): element-wise part in place, per-row statistics, normalise
      for (int i = tid; i < RPP * SLAB; i += NTH) {
        const int lr = i / SLAB, c = i % SLAB;
        const int orow = s_rows[pass * RPP + lr], oc = n0 + c;
        if (orow >= 0 && oc < cout) {
          float v = stage[i];
          if (e.scale) v *= e.scale[oc];
          if (e.shift) v += e.shift[oc];
          if (e.res_pre) v += e.res_pre[(size_t)orow * e.res_pre_ld + oc];
          if (e.relu) v = fmaxf(v, 0.0f);
          stage[i] = v;
        }
      }
      __syncthreads();
      for (int lr = tid; lr < RPP; lr += NTH) {
        float s = 0.0f;
        for (int c = 0; c < cout; ++c) s += stage[lr * SLAB + c];
        const float mean = s / (float)cout;
        float q = 0.0f;
        for (int c = 0; c < cout; ++c) { const float d = stage[lr * SLAB + c] - mean; q += d * d; }
        s_stat[2 * lr] = mean;
        s_stat[2 * lr + 1] = 1.0f / sqrtf(q / (float)cout + e.ln_eps);
      }
      __syncthreads();
      for (int i = tid; i < RPP * SLAB; i += NTH) {
        const int lr = i / SLAB, c = i % SLAB;
        const int orow = s_rows[pass * RPP + lr], oc = n0 + c;
        if (orow >= 0 && oc < cout) out[(size_t)orow * out_ld + oc] = (stage[i] - s_stat[2 * lr]) * s_stat[2 * lr + 1] * e.ln_gamma[oc] + e.ln_beta[oc];
      }
    } else if (vec) {
      for (int i = tid; i < RPP * (SLAB / 4); i += NTH) {
        const int lr = i / (SLAB / 4), c4 = i % (SLAB / 4);
        const int orow = s_rows[pass * RPP + lr], oc = n0 + c4 * 4;
        if (orow >= 0 && oc < cout) {
          float4 v = *(const float4 *)(stage + lr * SLAB + c4 * 4);
          if (e.scale) {
            const float4 sc = *(const float4 *)(e.scale + oc);
            v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w;
          }
          if (e.shift) {
            const float4 sh = *(const float4 *)(e.shift + oc);
            v.x += sh.x; v.y += sh.y; v.z += sh.z; v.w += sh.w;
          }
          if (e.res_pre) {
            const float4 q = *(const float4 *)(e.res_pre + (size_t)orow * e.res_pre_ld + oc);
            v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
          }
          if (e.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
          if (e.pair) {
            const float4 p0 = *(const float4 *)(e.pair + (size_t)orow * e.pair_ld + 2 * oc);
            const float4 p1 = *(const float4 *)(e.pair + (size_t)orow * e.pair_ld + 2 * oc + 4);
            v.x += p0.x + p0.y; v.y += p0.z + p0.w; v.z += p1.x + p1.y; v.w += p1.z + p1.w;
          }
          *(float4 *)(out + (size_t)orow * out_ld + oc) = v;
        }
      }
    } else {
      for (int i = tid; i < RPP * SLAB; i += NTH) {
        const int lr = i / SLAB, c = i % SLAB;
        const int orow = s_rows[pass * RPP + lr], oc = n0 + c;
        if (orow >= 0 && oc < cout) {
          float v = stage[lr * SLAB + c];
          if (e.scale) v *= e.scale[oc];
          if (e.shift) v += e.shift[oc];
          if (e.res_pre) v += e.res_pre[(size_t)orow * e.res_pre_ld + oc];
          if (e.relu) v = fmaxf(v, 0.0f);
          if (e.pair) v += e.pair[(size_t)orow * e.pair_ld + 2 * oc] + e.pair[(size_t)orow * e.pair_ld + 2 * oc + 1];
          out[(size_t)orow * out_ld + oc] = v;
        }
      }
    }
  }
  __syncthreads();  // buffer and s_rows are reused by the next tile
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned ls3d_bf16_rne(float x) {  // bits of bf16(x), round to nearest even
  const unsigned u = __float_as_uint(x);
  return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
__device__ __forceinline__ void ls3d_split_pair(float a, float b, unsigned &hi, unsigned &lo) {
  const unsigned ha = __float_as_uint(a) & 0xFFFF0000u, hb = __float_as_uint(b) & 0xFFFF0000u;
  hi = (ha >> 16) | hb;
  lo = ls3d_bf16_rne(a - __uint_as_float(ha)) | (ls3d_bf16_rne(b - __uint_as_float(hb)) << 16);
}
__device__ __forceinline__ void ls3d_split8(const float4 &f0, const float4 &f1, uint4 &hi, uint4 &lo) {
  ls3d_split_pair(f0.x, f0.y, hi.x, lo.x);
  ls3d_split_pair(f0.z, f0.w, hi.y, lo.y);
  ls3d_split_pair(f1.x, f1.y, hi.z, lo.z);
  ls3d_split_pair(f1.z, f1.w, hi.w, lo.w);
}

// exact 3-way split of an f32 into bf16 planes: a = h + m + l with h, m truncated and l rounded (8 + 8 + 8 mantissa bits)
__device__ __forceinline__ void ls3d_split_pair3(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const unsigned ha = __float_as_uint(a) & 0xFFFF0000u, hb = __float_as_uint(b) & 0xFFFF0000u;
  const float ra = a - __uint_as_float(ha), rb = b - __uint_as_float(hb);
  const unsigned ma = __float_as_uint(ra) & 0xFFFF0000u, mb = __float_as_uint(rb) & 0xFFFF0000u;
  h = (ha >> 16) | hb;
  m = (ma >> 16) | mb;
  l = ls3d_bf16_rne(ra - __uint_as_float(ma)) | (ls3d_bf16_rne(rb - __uint_as_float(mb)) << 16);
}
// the same exact split with round-to-nearest planes (a = h + m + l still holds exactly: every remainder has <= 16, then <= 8
// significant bits), written with __bf16 conversions: two-element converts compile to v_cvt_pk_bf16_f32 on gfx950, ~1/3 of the
// VALU work of the bit-twiddling version.  Used where the split runs once per staged element (tileconv.hip).
typedef __bf16 ls3d_bf16x2 __attribute__((ext_vector_type(2)));
typedef float ls3d_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void ls3d_split_pair3_rne(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const ls3d_f32x2 v = {a, b};
  const ls3d_bf16x2 hh = __builtin_convertvector(v, ls3d_bf16x2);
  const ls3d_f32x2 r1 = v - __builtin_convertvector(hh, ls3d_f32x2);
  const ls3d_bf16x2 mm = __builtin_convertvector(r1, ls3d_bf16x2);
  const ls3d_f32x2 r2 = r1 - __builtin_convertvector(mm, ls3d_f32x2);
  const ls3d_bf16x2 ll = __builtin_convertvector(r2, ls3d_bf16x2);
  h = __builtin_bit_cast(unsigned, hh);
  m = __builtin_bit_cast(unsigned, mm);
  l = __builtin_bit_cast(unsigned, ll);
}

__device__ __forceinline__ void ls3d_split8x3(const float4 &f0, const float4 &f1, uint4 &h, uint4 &m, uint4 &l) {
  ls3d_split_pair3(f0.x, f0.y, h.x, m.x, l.x);
  ls3d_split_pair3(f0.z, f0.w, h.y, m.y, l.y);
  ls3d_split_pair3(f1.x, f1.y, h.z, m.z, l.z);
  ls3d_split_pair3(f1.z, f1.w, h.w, m.w, l.w);
}
