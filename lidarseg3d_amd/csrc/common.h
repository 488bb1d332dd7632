// common.h — shared device helpers for libls3d (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ls3d.h"

#define LS3D_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define LS3D_INF_I32 0x7F7F7F7F  // memset(0x7F) sentinel: larger than any row index

#define LS3D_RETURN_IF_LAUNCH_FAILED()            \
  do {                                            \
    if (hipGetLastError() != hipSuccess) return LS3D_ERR_LAUNCH; \
  } while (0)

// hipFuncSetAttribute (the opt-in to > 64 KB of dynamic LDS) is a per-DEVICE property of a kernel: call sites remember it per device
constexpr int LS3D_MAX_DEVICES = 64;
static inline int ls3d_device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
  return d % LS3D_MAX_DEVICES;
}

// 1-D launch geometry for grid-stride kernels: enough blocks to cover `work` items, capped so that a
// launch never exceeds ~8 blocks per CU (256 CUs); the kernels loop over the remainder.
static inline dim3 ls3d_grid(long long work, int block = 256, int max_blocks = 2048) {
  long long b = (work + block - 1) / block;
  if (b < 1) b = 1;
  if (b > max_blocks) b = max_blocks;
  return dim3((unsigned)b);
}

__device__ __forceinline__ int ls3d_count(int n, const int32_t *n_dev) {
  if (n_dev) {
    int d = *n_dev;
    return d < n ? d : n;
  }
  return n;
}

// ---- LDS-DMA (global_load_lds_dwordx4): every lane copies 16 bytes from its own global address straight into LDS at
//      lds_wave_base + lane * 16 (the destination is wave-uniform base + lane offset; no VGPRs, asynchronous, counted in
//      vmcnt).  hipcc does not count inline-asm memory operations, so the callers wait with LS3D_WAIT_VMCNT themselves and
//      synchronise waves with the raw s_barrier (a __syncthreads() would drain the whole DMA queue).  M0 (the DMA's LDS
//      base) is written and restored inside one asm statement.  Under tests/hipsim these degrade to memcpy / no-op.
#ifdef HIPSIM
__device__ __forceinline__ void ls3d_glds16(const void *gsrc, void *lds_wave_base) {
  memcpy((char *)lds_wave_base + hipsim::lane() * 16, gsrc, 16);
}
__device__ __forceinline__ void ls3d_glds16x3(const void *gbase, unsigned voff0, unsigned voff1, unsigned voff2, void *lds_wave_base) {
  memcpy((char *)lds_wave_base + hipsim::lane() * 16, (const char *)gbase + voff0, 16);
  memcpy((char *)lds_wave_base + 1024 + hipsim::lane() * 16, (const char *)gbase + voff1, 16);
  memcpy((char *)lds_wave_base + 2048 + hipsim::lane() * 16, (const char *)gbase + voff2, 16);
}
__device__ __forceinline__ float ls3d_load_agent(const float *p) { return *p; }
__device__ __forceinline__ void ls3d_store_agent(float *p, float v) { *p = v; }
__device__ __forceinline__ int ls3d_load_agent_i32(const int *p) { return *(const volatile int *)p; }
__device__ __forceinline__ void ls3d_store_agent_i32(int *p, int v) { *(volatile int *)p = v; }
struct ls3d_cohbuf { const char *base; };
__device__ __forceinline__ ls3d_cohbuf ls3d_cohbuf_make(const void *base) { return ls3d_cohbuf{(const char *)base}; }
__device__ __forceinline__ float4 ls3d_load4_agent(const ls3d_cohbuf &b, unsigned byte_off) { return *(const float4 *)(b.base + byte_off); }
__device__ __forceinline__ void ls3d_store4_agent(const ls3d_cohbuf &b, unsigned byte_off, const float4 &v) { *(float4 *)(b.base + byte_off) = v; }
__device__ __forceinline__ float4 ls3d_load4_buf(const ls3d_cohbuf &b, unsigned byte_off) { return *(const float4 *)(b.base + byte_off); }
__device__ __forceinline__ void ls3d_sleep() {}
#define LS3D_WAIT_VMCNT(n) ((void)0)
#define LS3D_WAIT_LGKMCNT0() ((void)0)
#define LS3D_SCHED_FENCE() ((void)0)
#define LS3D_RAW_BARRIER() __syncthreads()
#define LS3D_TRAP() abort()
#else
// the wave enters the trap handler: the queue reports an exception and the process aborts.  (As inline assembly, not __builtin_trap(): a noreturn call
// changes the control flow graph of the kernel around it.)
#define LS3D_TRAP() asm volatile("s_trap 2" ::: "memory")
__device__ __forceinline__ void ls3d_glds16(const void *gsrc, void *lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// three consecutive 1 KB blocks: global gbase (wave-uniform, SGPR pair) + voff_i (per lane, = lane * 16 + i * 1024 + a uniform offset)
// -> LDS lds_wave_base + i * 1024 + lane * 16.  Scalar base + 32-bit lane offsets: no 64-bit VALU address arithmetic per block.
__device__ __forceinline__ void ls3d_glds16x3(const void *gbase, unsigned voff0, unsigned voff1, unsigned voff2, void *lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_wave_base);
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
      "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %4\n\ts_mov_b32 m0, %0"
      : "=&s"(keep) : "v"(voff0), "v"(voff1), "v"(voff2), "s"(gbase), "s"(dst) : "memory", "scc");
}
// agent-scope relaxed atomics on plain floats: they go through to the point where every XCD's L2 agrees (the L2s of the eight XCDs
// are not coherent with each other inside a kernel for ordinary cached accesses), without the L2 write-back / invalidate that a
// release / acquire fence pair would cost every other workgroup of the XCD
__device__ __forceinline__ float ls3d_load_agent(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ls3d_store_agent(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int ls3d_load_agent_i32(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ls3d_store_agent_i32(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// 16-byte accesses with the same cache policy as those atomics (`sc1`: the access is performed where all XCDs agree, a load never returns
// another XCD's stale line, a store is written through): raw buffer instructions over [base, base + 4 GB) with the sc1 bit in their
// cache-policy operand - hipcc tracks them in vmcnt like any load.  Used by the chained tile convolution (tileconv.hip), whose layers
// read each other's output rows across XCDs inside ONE launch.
struct ls3d_cohbuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ ls3d_cohbuf ls3d_cohbuf_make(const void *base) {
  return ls3d_cohbuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xFFFFFFFF, 0x00020000)};
}
typedef int ls3d_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ls3d_load4_agent(const ls3d_cohbuf &b, unsigned byte_off) {
  const ls3d_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)byte_off, 0, 16 /* sc1 */);
  return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void ls3d_store4_agent(const ls3d_cohbuf &b, unsigned byte_off, const float4 &v) {
  const ls3d_i32x4 w = {__float_as_int(v.x), __float_as_int(v.y), __float_as_int(v.z), __float_as_int(v.w)};
  __builtin_amdgcn_raw_buffer_store_b128(w, b.r, (int)byte_off, 0, 16 /* sc1 */);
}
// the same addressing (scalar base + 32-bit lane offset: half the address registers of a 64-bit global load) with the ordinary cache policy
__device__ __forceinline__ float4 ls3d_load4_buf(const ls3d_cohbuf &b, unsigned byte_off) {
  const ls3d_i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)byte_off, 0, 0);
  return make_float4(__int_as_float(v.x), __int_as_float(v.y), __int_as_float(v.z), __int_as_float(v.w));
}
__device__ __forceinline__ void ls3d_sleep() { __builtin_amdgcn_s_sleep(16); }
#define LS3D_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define LS3D_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")  /* every LDS read issued so far has returned */
#define LS3D_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)  /* nothing is scheduled across: e.g. keeps a batch of ds_reads together */
#define LS3D_RAW_BARRIER()                          \
  do {                                              \
    asm volatile("" ::: "memory");                  \
    __builtin_amdgcn_s_barrier();                   \
    asm volatile("" ::: "memory");                  \
  } while (0)
#endif

// ---- in-kernel tracing (ls3d_tile_conv's trace flag): shader-cycle counter, the 100 MHz wall clock shared by all CUs, and where a
//      wave runs (HW_ID: wave / SIMD / CU / SH / SE fields, XCC_ID: the XCD)
#ifdef HIPSIM
__device__ __forceinline__ unsigned long long ls3d_cycles() { return 0; }
__device__ __forceinline__ unsigned long long ls3d_walltime() { return 0; }
__device__ __forceinline__ unsigned ls3d_hw_id() { return 0; }
__device__ __forceinline__ unsigned ls3d_xcc_id() { return 0; }
#else
__device__ __forceinline__ unsigned long long ls3d_cycles() { return __builtin_readcyclecounter(); }          // s_memtime
__device__ __forceinline__ unsigned long long ls3d_walltime() { return __builtin_amdgcn_s_memrealtime(); }    // 100 MHz
__device__ __forceinline__ unsigned ls3d_hw_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v));
  return v;
}
__device__ __forceinline__ unsigned ls3d_xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v;
}
#endif

__device__ __forceinline__ uint64_t ls3d_mix(uint64_t k) {  // 64-bit finaliser (splitmix)
  k ^= k >> 30; k *= 0xbf58476d1ce4e5b9ull;
  k ^= k >> 27; k *= 0x94d049bb133111ebull;
  k ^= k >> 31;
  return k;
}

// find-or-insert; returns the slot.  keys[] initialised to LS3D_EMPTY_KEY (memset 0xFF).
__device__ __forceinline__ int ls3d_hash_claim(uint64_t *keys, uint32_t mask, uint64_t key) {
  uint32_t s = (uint32_t)ls3d_mix(key) & mask;
  for (;;) {
    unsigned long long prev = atomicCAS((unsigned long long *)&keys[s], (unsigned long long)LS3D_EMPTY_KEY,
                                        (unsigned long long)key);
    if (prev == LS3D_EMPTY_KEY || prev == key) return (int)s;
    s = (s + 1) & mask;
  }
}

// lookup; returns the slot or -1
__device__ __forceinline__ int ls3d_hash_find(const uint64_t *keys, uint32_t mask, uint64_t key) {
  uint32_t s = (uint32_t)ls3d_mix(key) & mask;
  for (;;) {
    uint64_t k = keys[s];
    if (k == key) return (int)s;
    if (k == LS3D_EMPTY_KEY) return -1;
    s = (s + 1) & mask;
  }
}

__device__ __forceinline__ uint64_t ls3d_key(int b, int z, int y, int x, int Z, int Y, int X) {
  return (((uint64_t)b * (uint64_t)Z + (uint64_t)z) * (uint64_t)Y + (uint64_t)y) * (uint64_t)X + (uint64_t)x;
}

// exclusive scan of int32 (shared by voxelize / rulebook): three launches, any n.
// tmp must hold roundup(n,1024)/1024 + 1 ints.  total_out (device) receives the grand total (may be NULL).
int ls3d_exclusive_scan_i32(const int32_t *in, int32_t *out, int n, int32_t *tmp, int32_t *total_out,
                            hipStream_t stream);
static inline size_t ls3d_scan_tmp_ints(long long n) { return (size_t)((n + 1023) / 1024 + 1); }
