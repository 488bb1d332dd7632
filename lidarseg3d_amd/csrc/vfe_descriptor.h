// vfe_descriptor.h — the per-voxel descriptor shared by the voxel feature extractors (vfe.hip, transvfe.hip).
#pragma once
#include "common.h"

#define LS3D_MAX_FEAT 16

// [mean_xyz(3), max_xyz(3), min_xyz(3), mean_other(C-3), density, std]   (voxel_encoder.py:82-121)
// Zero-padding slots are recognised as the reference does: row sum == 0 (:87).
__device__ __forceinline__ void vfe_descriptor(const float *vox, int P, int C, int num, float *desc) {
  const float cnt = (float)num;
  float mean[LS3D_MAX_FEAT];
  for (int c = 0; c < C; ++c) {
    float s = 0.0f;
    for (int p = 0; p < P; ++p) s += vox[p * C + c];
    mean[c] = __fdiv_rn(s, cnt);
  }
  float mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, mn[3] = {3.0e38f, 3.0e38f, 3.0e38f};
  float nmask = 0.0f, dev = 0.0f;
  for (int p = 0; p < P; ++p) {
    float rs = 0.0f;
    for (int c = 0; c < C; ++c) rs += vox[p * C + c];
    const float m = (rs != 0.0f) ? 1.0f : 0.0f;
    const float big = (1.0f - m) * 1e5f;
    float sq = 0.0f;
    for (int a = 0; a < 3; ++a) {
      const float x = vox[p * C + a];
      mx[a] = fmaxf(mx[a], x - big);
      mn[a] = fminf(mn[a], x + big);
      const float d = (x - mean[a]) * m;
      sq += d * d;
    }
    nmask += m;
    dev += sqrtf(sq);
  }
  desc[0] = mean[0]; desc[1] = mean[1]; desc[2] = mean[2];
  desc[3] = mx[0]; desc[4] = mx[1]; desc[5] = mx[2];
  desc[6] = mn[0]; desc[7] = mn[1]; desc[8] = mn[2];
  for (int c = 3; c < C; ++c) desc[6 + c] = mean[c];
  desc[C + 6] = __fdiv_rn(nmask, (float)P);
  desc[C + 7] = __fdiv_rn(dev, cnt);
}


// The same descriptor with the voxel shape known at compile time (P point slots of C features: 5 x 5 in the shipped nuScenes / Waymo / KITTI readers),
// every loop unrolled: the voxel's P * C floats are P * C independent loads in flight instead of a dependent load per trip of a runtime loop, and
// mean[] / desc[] are registers instead of runtime-indexed arrays.  Same operations in the same order as vfe_descriptor: bit-identical.
template <int P, int C>
__device__ __forceinline__ void vfe_descriptor_fixed(const float *__restrict__ vox, int num, float (&x)[P * C], float (&desc)[C + 8]) {
  static_assert(C >= 3 && C <= LS3D_MAX_FEAT, "3 <= C <= LS3D_MAX_FEAT");
#pragma unroll
  for (int i = 0; i < P * C; ++i) x[i] = vox[i];
  const float cnt = (float)num;
  float mean[C];
#pragma unroll
  for (int c = 0; c < C; ++c) {
    float s = 0.0f;
#pragma unroll
    for (int p = 0; p < P; ++p) s += x[p * C + c];
    mean[c] = __fdiv_rn(s, cnt);
  }
  float mx[3] = {-3.0e38f, -3.0e38f, -3.0e38f}, mn[3] = {3.0e38f, 3.0e38f, 3.0e38f};
  float nmask = 0.0f, dev = 0.0f;
#pragma unroll
  for (int p = 0; p < P; ++p) {
    float rs = 0.0f;
#pragma unroll
    for (int c = 0; c < C; ++c) rs += x[p * C + c];
    const float m = (rs != 0.0f) ? 1.0f : 0.0f;
    const float big = (1.0f - m) * 1e5f;
    float sq = 0.0f;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float v = x[p * C + a];
      mx[a] = fmaxf(mx[a], v - big);
      mn[a] = fminf(mn[a], v + big);
      const float d = (v - mean[a]) * m;
      sq += d * d;
    }
    nmask += m;
    dev += sqrtf(sq);
  }
  desc[0] = mean[0]; desc[1] = mean[1]; desc[2] = mean[2];
  desc[3] = mx[0]; desc[4] = mx[1]; desc[5] = mx[2];
  desc[6] = mn[0]; desc[7] = mn[1]; desc[8] = mn[2];
#pragma unroll
  for (int c = 3; c < C; ++c) desc[6 + c] = mean[c];
  desc[C + 6] = __fdiv_rn(nmask, (float)P);
  desc[C + 7] = __fdiv_rn(dev, cnt);
}
