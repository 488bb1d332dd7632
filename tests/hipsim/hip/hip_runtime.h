// tests/hipsim/hip/hip_runtime.h — a tiny HOST emulation of the HIP device model, TEST INFRASTRUCTURE ONLY.
//
// Purpose: there is no GPU in the development container, and GPU minutes are rationed.  This header lets
// the *unmodified* kernel sources under lidarseg3d_amd/csrc/ be compiled with the host clang++ and executed
// on the CPU so that indexing / protocol bugs are caught before a GPU run.  It is NOT a backend: the package
// never builds or loads it, the product library is built by hipcc for gfx950 only, and nothing here is a
// fallback path (tests that use it are the `-m "not gpu"` logic tests; parity proper runs on the MI355X).
//
// Model: blocks run one after another; the threads of a block are ucontext fibers on one OS thread.
//   __syncthreads()            -> block barrier (all live fibers must arrive)
//   wave collectives (64 wide) -> wave barrier + exchange buffer (__shfl*, __ballot, MFMA, ...)
// Atomics are trivially atomic (single OS thread).  Races between threads of different waves are NOT
// detected; divergent collectives deadlock and are reported.
#pragma once
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <functional>
#include <vector>

#define HIPSIM 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;
static const int warpSize = 64;

typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
inline hipError_t hipGetLastError() { return 0; }
inline const char *hipGetErrorString(hipError_t) { return "hipsim"; }
inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount };
inline hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? 0 : 1; }
inline hipError_t hipGetDevice(int *d) { *d = 0; return 0; }
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 3; return 0; }
inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return 0; }

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
inline float2 make_float2(float a, float b) { return float2{a, b}; }
inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
inline int2 make_int2(int a, int b) { return int2{a, b}; }

// ---------------------------------------------------------------------------------------- scheduler
namespace hipsim {
enum Wait { RUN = 0, BLOCK_BAR = 1, WAVE_BAR = 2, DONE = 3 };
struct Fiber {
  void *sp = nullptr;
  char *stack = nullptr;
  int wait = DONE;
  uint3_ tid;
};
extern std::vector<Fiber> fibers;
extern int cur;
extern uint64_t xbuf[64][64];     // per-wave exchange buffer: [slot][lane]; wave id indexes via cur/64
extern uint64_t (*wave_x)[64][64];
extern int bar_acc;               // accumulator for __syncthreads_or/count
void yield_wait(int kind);
void wave_barrier();
inline int lane() { return cur & 63; }
inline int wave() { return cur >> 6; }
uint64_t *wslot(int slot);        // exchange row for the current wave
void run_grid(dim3 grid, dim3 block, const std::function<void()> &body);
}  // namespace hipsim

inline void __syncthreads() { hipsim::yield_wait(hipsim::BLOCK_BAR); }
int __syncthreads_or(int pred);
int __syncthreads_count(int pred);
inline void __threadfence() {}
inline void __threadfence_block() {}

namespace hipsim { extern char dyn_smem[160 * 1024]; }
#define HIP_DYNAMIC_SHARED(type, var) type *var = (type *)hipsim::dyn_smem;
template <typename... KArgs, typename... Args>
inline void hipLaunchKernelGGL(void (*k)(KArgs...), dim3 grid, dim3 block, size_t shmem, hipStream_t, Args... args) {
  if (shmem > sizeof(hipsim::dyn_smem)) { fprintf(stderr, "hipsim: dynamic LDS too large\n"); abort(); }
  hipsim::run_grid(grid, block, [=]() { k(args...); });
}

// ---------------------------------------------------------------------------------------- wave collectives
template <typename T>
inline T hipsim_xchg(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "xchg");
  uint64_t *row = hipsim::wslot(0);
  uint64_t raw = 0;
  memcpy(&raw, &v, sizeof(T));
  row[hipsim::lane()] = raw;
  hipsim::wave_barrier();
  uint64_t got = row[src_lane & 63];
  hipsim::wave_barrier();
  T out;
  memcpy(&out, &got, sizeof(T));
  return out;
}
template <typename T> inline T __shfl(T v, int src, int width = 64) {
  int l = hipsim::lane();
  return hipsim_xchg(v, (l & ~(width - 1)) | (src & (width - 1)));
}
template <typename T> inline T __shfl_xor(T v, int mask, int width = 64) {
  int l = hipsim::lane();
  return hipsim_xchg(v, (l & ~(width - 1)) | ((l ^ mask) & (width - 1)));
}
template <typename T> inline T __shfl_down(T v, unsigned d, int width = 64) {
  int l = hipsim::lane();
  int s = (l & (width - 1)) + (int)d;
  return hipsim_xchg(v, s < width ? (l & ~(width - 1)) | s : l);
}
template <typename T> inline T __shfl_up(T v, unsigned d, int width = 64) {
  int l = hipsim::lane();
  int s = (l & (width - 1)) - (int)d;
  return hipsim_xchg(v, s >= 0 ? (l & ~(width - 1)) | s : l);
}
unsigned long long __ballot(int pred);
inline int __any(int p) { return __ballot(p) != 0ull; }
inline int __all(int p);
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }
inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
inline float __fdividef(float a, float b) { return a / b; }
inline float __fdiv_rn(float a, float b) { return a / b; }
inline float __fsub_rn(float a, float b) { return a - b; }
inline float __fmul_rn(float a, float b) { return a * b; }
inline float __fadd_rn(float a, float b) { return a + b; }
inline float __fsqrt_rn(float a) { return sqrtf(a); }
inline float __expf(float a) { return expf(a); }
inline float rsqrtf(float a) { return 1.0f / sqrtf(a); }
inline int __builtin_amdgcn_readfirstlane_sim(int v) { return __shfl(v, 0); }
#define __builtin_amdgcn_readfirstlane __builtin_amdgcn_readfirstlane_sim
inline int __builtin_amdgcn_readlane_sim(int v, int l) { return __shfl(v, l); }
#define __builtin_amdgcn_readlane __builtin_amdgcn_readlane_sim

using std::max;
using std::min;

// atomics (single OS thread => plain RMW)
template <typename T> inline T atomicAdd(T *p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> inline T atomicMin(T *p, T v) { T o = *p; *p = o < v ? o : v; return o; }
template <typename T> inline T atomicMax(T *p, T v) { T o = *p; *p = o > v ? o : v; return o; }
template <typename T> inline T atomicOr(T *p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> inline T atomicAnd(T *p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> inline T atomicExch(T *p, T v) { T o = *p; *p = v; return o; }
template <typename T> inline T atomicCAS(T *p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// MFMA 32x32x2 f32 (semantics per /opt/skills/guides/cdna_hip_programming.md §3):
//   A: lane l holds A[i=l&31][k=l>>5];  B: lane l holds B[k=l>>5][j=l&31]
//   C/D (16 regs): col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
//   result == k-ordered fmaf chain on top of C.
typedef float hipsim_f32x16 __attribute__((ext_vector_type(16)));
typedef float hipsim_f32x4 __attribute__((ext_vector_type(4)));
hipsim_f32x16 hipsim_mfma_32x32x2f32(float a, float b, hipsim_f32x16 c, int, int, int);
#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipsim_mfma_32x32x2f32
typedef __bf16 hipsim_bf16x8 __attribute__((ext_vector_type(8)));
hipsim_f32x16 hipsim_mfma_32x32x16_bf16(hipsim_bf16x8 a, hipsim_bf16x8 b, hipsim_f32x16 c, int, int, int);
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipsim_mfma_32x32x16_bf16
// fp8 (OCP e4m3fn, the gfx950 format): 8 values per lane packed in an int64, same A / B / C layout as the bf16 form
hipsim_f32x16 hipsim_mfma_32x32x16_fp8(long a, long b, hipsim_f32x16 c, int, int, int);
#define __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8 hipsim_mfma_32x32x16_fp8
int hipsim_cvt_pk_fp8_f32(float a, float b, int old, bool word_sel);  // two e4m3 bytes into the low / high half of `old`
#define __builtin_amdgcn_cvt_pk_fp8_f32 hipsim_cvt_pk_fp8_f32
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
