// Probe (test infrastructure): does global_load_lds_dwordx4 (LDS-DMA) reach LDS offsets beyond 64 KB on gfx950, and does the
// M0 save/restore recipe work under hipcc?  Prints PASS/FAIL per destination offset.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

__device__ __forceinline__ void glds16(const void *gsrc, void *lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__global__ void k_probe(const float *src, float *out, int lds_off_bytes) {
  HIP_DYNAMIC_SHARED(char, smem)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // every lane fetches a scattered 16-byte piece: piece index (lane * 7 + wave) % 256 of src
  const float *g = src + ((lane * 7 + wave * 3) % 256) * 4;
  glds16(g, smem + lds_off_bytes + wave * 1024);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const float4 v = *(const float4 *)(smem + lds_off_bytes + wave * 1024 + lane * 16);
  *(float4 *)(out + (size_t)threadIdx.x * 4) = v;
}

int main() {
  std::vector<float> h(1024);
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  float *src, *out;
  hipMalloc(&src, 4096); hipMalloc(&out, 256 * 16);
  hipMemcpy(src, h.data(), 4096, hipMemcpyHostToDevice);
  const int lds_total = 160 * 1024;
  hipError_t ae = hipFuncSetAttribute((const void *)k_probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_total);
  printf("setattr %d\n", (int)ae);
  int offs[] = {0, 4096, 61440, 65536, 66560, 98304, 131072, 150 * 1024};
  for (int o : offs) {
    hipMemset(out, 0xFF, 256 * 16);
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), lds_total, 0, src, out, o);
    hipError_t e = hipDeviceSynchronize();
    std::vector<float> r(1024);
    hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int t = 0; t < 256; ++t) {
      const int lane = t & 63, wave = t >> 6, piece = (lane * 7 + wave * 3) % 256;
      for (int j = 0; j < 4; ++j) bad += r[t * 4 + j] != (float)(piece * 4 + j);
    }
    printf("lds offset %6d: %s (err %d, mismatches %d)\n", o, bad ? "FAIL" : "PASS", (int)e, bad);
  }
  return 0;
}
