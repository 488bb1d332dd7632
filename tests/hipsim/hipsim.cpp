// tests/hipsim/hipsim.cpp — fiber scheduler behind tests/hipsim/hip/hip_runtime.h (TEST INFRASTRUCTURE ONLY).
#include <hip/hip_runtime.h>

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

// minimal x86-64 SysV context switch (callee-saved registers + stack pointer); ~50x cheaper than
// swapcontext(), which makes a sigprocmask system call per switch.
extern "C" void hipsim_switch(void **save_sp, void *load_sp);
asm(R"(
.text
.globl hipsim_switch
.type hipsim_switch,@function
hipsim_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size hipsim_switch,.-hipsim_switch
)");

namespace hipsim {
char dyn_smem[160 * 1024];
std::vector<Fiber> fibers;
void *sched_sp = nullptr;
int cur = 0;
static uint64_t xstore[16][4][64];  // [wave][slot][lane]
int bar_acc = 0;
static const std::function<void()> *body_fn = nullptr;
static const size_t STACK = 256 * 1024;

uint64_t *wslot(int slot) { return xstore[wave()][slot]; }

static void trampoline() {
  (*body_fn)();
  fibers[cur].wait = DONE;
  hipsim_switch(&fibers[cur].sp, sched_sp);
  abort();  // a finished fiber is never resumed
}

void yield_wait(int kind) {
  fibers[cur].wait = kind;
  hipsim_switch(&fibers[cur].sp, sched_sp);
}

void wave_barrier() { yield_wait(WAVE_BAR); }

static void resume(int i) {
  cur = i;
  threadIdx = fibers[i].tid;
  fibers[i].wait = RUN;
  hipsim_switch(&sched_sp, fibers[i].sp);
}

static void run_block(int nthreads) {
  if ((int)fibers.size() < nthreads) fibers.resize(nthreads);
  for (int i = 0; i < nthreads; ++i) {
    Fiber &f = fibers[i];
    if (!f.stack) f.stack = (char *)malloc(STACK);
    uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
    void **sp = (void **)top;
    *--sp = nullptr;               // fake return address of trampoline
    *--sp = (void *)trampoline;    // `ret` of the first switch jumps here
    for (int r = 0; r < 6; ++r) *--sp = nullptr;  // rbp rbx r12 r13 r14 r15
    f.sp = (void *)sp;
    f.tid.x = i % blockDim.x;
    f.tid.y = (i / blockDim.x) % blockDim.y;
    f.tid.z = i / (blockDim.x * blockDim.y);
    f.wait = RUN;
  }
  // first pass: start every fiber
  for (int i = 0; i < nthreads; ++i) resume(i);
  int nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool progress = false, all_done = true;
    // wave barriers
    for (int w = 0; w < nwaves; ++w) {
      int lo = w * 64, hi = std::min(nthreads, lo + 64);
      int waiting = 0, live = 0;
      for (int i = lo; i < hi; ++i) {
        if (fibers[i].wait != DONE) ++live;
        if (fibers[i].wait == WAVE_BAR) ++waiting;
      }
      if (live && waiting == live) {
        for (int i = lo; i < hi; ++i)
          if (fibers[i].wait == WAVE_BAR) resume(i);
        progress = true;
      }
    }
    int live = 0, atbar = 0;
    for (int i = 0; i < nthreads; ++i) {
      if (fibers[i].wait != DONE) { ++live; all_done = false; }
      if (fibers[i].wait == BLOCK_BAR) ++atbar;
    }
    if (all_done) break;
    if (live && atbar == live) {
      for (int i = 0; i < nthreads; ++i)
        if (fibers[i].wait == BLOCK_BAR) resume(i);
      progress = true;
    }
    if (!progress) {
      fprintf(stderr, "hipsim: DEADLOCK in block (%u,%u,%u): divergent barrier/collective\n", blockIdx.x, blockIdx.y, blockIdx.z);
      for (int i = 0; i < nthreads; ++i) fprintf(stderr, "%d", fibers[i].wait);
      fprintf(stderr, "\n");
      abort();
    }
  }
}

void run_grid(dim3 grid, dim3 block, const std::function<void()> &body) {
  body_fn = &body;
  gridDim = grid;
  blockDim = block;
  int nthreads = block.x * block.y * block.z;
  if (nthreads > 1024) { fprintf(stderr, "hipsim: block too large\n"); abort(); }
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        blockIdx.x = x; blockIdx.y = y; blockIdx.z = z;
        run_block(nthreads);
      }
  body_fn = nullptr;
}
}  // namespace hipsim

int __syncthreads_or(int pred) {
  static int acc;
  if (pred) acc = 1;
  __syncthreads();
  int r = acc;
  __syncthreads();
  if (hipsim::cur == 0) acc = 0;
  __syncthreads();
  return r;
}

int __syncthreads_count(int pred) {
  static int acc;
  if (pred) acc += 1;
  __syncthreads();
  int r = acc;
  __syncthreads();
  if (hipsim::cur == 0) acc = 0;
  __syncthreads();
  return r;
}

unsigned long long __ballot(int pred) {
  uint64_t *row = hipsim::wslot(1);
  row[hipsim::lane()] = pred ? 1 : 0;
  // lanes that already exited must read as 0: clear is done by the first arriving lane of each op,
  // so instead gather only over lanes that wrote in this round using a round tag
  static uint64_t tag[16][64];
  static uint64_t round[16];
  int w = hipsim::wave();
  uint64_t my = round[w] + 1;
  tag[w][hipsim::lane()] = my;
  hipsim::wave_barrier();
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (tag[w][l] == my && row[l]) m |= 1ull << l;
  hipsim::wave_barrier();
  round[w] = my;
  return m;
}

int __all(int p) {
  uint64_t *row = hipsim::wslot(2);
  static uint64_t tag[16][64];
  static uint64_t round[16];
  int w = hipsim::wave();
  uint64_t my = round[w] + 1;
  tag[w][hipsim::lane()] = my;
  row[hipsim::lane()] = p ? 1 : 0;
  hipsim::wave_barrier();
  int ok = 1;
  for (int l = 0; l < 64; ++l)
    if (tag[w][l] == my && !row[l]) ok = 0;
  hipsim::wave_barrier();
  round[w] = my;
  return ok;
}

// The emulated MFMAs: every lane stages its operand fragments and its 16 accumulator values in per-wave arrays, ONE wave barrier, then the FIRST
// lane released from it computes the whole 32 x 32 x K product (rows x K x columns with the columns innermost: the compiler vectorises them;
// per output the products are added in ascending k, one fmaf each, as before) and the others read their 16 results.  The arrays are double
// buffered by a per-wave generation count, advanced by that first lane: a lane that runs ahead into the next MFMA writes the other buffer
// while slower lanes still read this one, and nobody reaches the MFMA after that - which reuses this buffer - before every lane has passed the
// next barrier, i.e. has read its results.  (Per-lane products and a second barrier per instruction were ~2/3 of the CPU suite's time.)
static unsigned mfma_gen[16];
static float mfma_A[2][16][32][16], mfma_B[2][16][16][32], mfma_D[2][16][32][32];

template <int K>
static inline hipsim_f32x16 mfma_finish(int p, int w, int l, unsigned gen, const hipsim_f32x16 &c) {
  const int col = l & 31, half4 = 4 * (l >> 5);
  for (int r = 0; r < 16; ++r) mfma_D[p][w][(r & 3) + 8 * (r >> 2) + half4][col] = c[r];
  hipsim::wave_barrier();
  if (mfma_gen[w] == gen) {  // the first lane out of the barrier: the whole product
    for (int i = 0; i < 32; ++i) {
      float *__restrict__ drow = mfma_D[p][w][i];
      for (int k = 0; k < K; ++k) {
        const float a = mfma_A[p][w][i][k];
        const float *__restrict__ brow = mfma_B[p][w][k];
        for (int j = 0; j < 32; ++j) drow[j] = fmaf(a, brow[j], drow[j]);
      }
    }
    mfma_gen[w] = gen + 1;
  }
  hipsim_f32x16 d;
  for (int r = 0; r < 16; ++r) d[r] = mfma_D[p][w][(r & 3) + 8 * (r >> 2) + half4][col];
  return d;
}

hipsim_f32x16 hipsim_mfma_32x32x2f32(float a, float b, hipsim_f32x16 c, int, int, int) {
  int w = hipsim::wave(), l = hipsim::lane();
  const unsigned gen = mfma_gen[w];
  const int p = gen & 1;
  mfma_A[p][w][l & 31][l >> 5] = a;
  mfma_B[p][w][l >> 5][l & 31] = b;
  return mfma_finish<2>(p, w, l, gen, c);
}

// 32x32x16 bf16: lane l holds A[i=l&31][k=8*(l>>5)+j] and B[k=8*(l>>5)+j][col=l&31], j = 0..7; C/D as the f32 form.
// (Any consistent k assignment gives the same product; kernels must not depend on it beyond A/B pairing.)
hipsim_f32x16 hipsim_mfma_32x32x16_bf16(hipsim_bf16x8 a, hipsim_bf16x8 b, hipsim_f32x16 c, int, int, int) {
  int w = hipsim::wave(), l = hipsim::lane();
  const unsigned gen = mfma_gen[w];
  const int p = gen & 1;
  uint16_t ra[8], rb[8];
  memcpy(ra, &a, 16);
  memcpy(rb, &b, 16);
  for (int j = 0; j < 8; ++j) {
    uint32_t ua = (uint32_t)ra[j] << 16, ub = (uint32_t)rb[j] << 16;
    float fa, fb;
    memcpy(&fa, &ua, 4);
    memcpy(&fb, &ub, 4);
    mfma_A[p][w][l & 31][8 * (l >> 5) + j] = fa;
    mfma_B[p][w][8 * (l >> 5) + j][l & 31] = fb;
  }
  return mfma_finish<16>(p, w, l, gen, c);
}

// ---- fp8 e4m3fn (OCP): 1 sign, 4 exponent (bias 7), 3 mantissa bits; max 448, no inf, 0x7F / 0xFF = NaN; round to nearest even,
//      saturating (the hardware convert saturates by default)
static uint8_t hipsim_f32_to_e4m3(float f) {
  if (f != f) return 0x7F;
  const uint8_t sign = std::signbit(f) ? 0x80 : 0;
  float a = fabsf(f);
  if (a >= 448.0f) return sign | 0x7E;
  if (a < 0.0009765625f) return sign;  // below half the smallest subnormal (2^-9 / 2): zero
  int e;
  frexpf(a, &e);  // a = m * 2^e, m in [0.5, 1)
  int E = e - 1;  // a = 1.x * 2^E
  if (E < -6) E = -6;  // subnormal range: fixed exponent
  const float q = ldexpf(a, 3 - E);  // in units of the last mantissa bit
  float r = nearbyintf(q);  // RNE (default rounding mode)
  int mant = (int)r;
  if (E == -6 && mant < 8) return sign | (uint8_t)mant;  // subnormal (or 0)
  if (mant == 16) { mant = 8; ++E; }
  if (E > 8 || (E == 8 && mant > 14)) return sign | 0x7E;
  return sign | (uint8_t)(((E + 7) << 3) | (mant - 8));
}
static float hipsim_e4m3_to_f32(uint8_t v) {
  const int sign = v >> 7, ex = (v >> 3) & 15, m = v & 7;
  float r;
  if (ex == 15 && m == 7) r = NAN;
  else if (ex == 0) r = ldexpf((float)m, -9);
  else r = ldexpf(1.0f + m / 8.0f, ex - 7);
  return sign ? -r : r;
}
int hipsim_cvt_pk_fp8_f32(float a, float b, int old, bool word_sel) {
  const unsigned pk = (unsigned)hipsim_f32_to_e4m3(a) | ((unsigned)hipsim_f32_to_e4m3(b) << 8);
  const unsigned o = (unsigned)old;
  return (int)(word_sel ? ((o & 0x0000FFFFu) | (pk << 16)) : ((o & 0xFFFF0000u) | pk));
}
hipsim_f32x16 hipsim_mfma_32x32x16_fp8(long a, long b, hipsim_f32x16 c, int, int, int) {
  int w = hipsim::wave(), l = hipsim::lane();
  const unsigned gen = mfma_gen[w];
  const int p = gen & 1;
  uint8_t ra[8], rb[8];
  memcpy(ra, &a, 8);
  memcpy(rb, &b, 8);
  for (int j = 0; j < 8; ++j) {
    mfma_A[p][w][l & 31][8 * (l >> 5) + j] = hipsim_e4m3_to_f32(ra[j]);
    mfma_B[p][w][8 * (l >> 5) + j][l & 31] = hipsim_e4m3_to_f32(rb[j]);
  }
  return mfma_finish<16>(p, w, l, gen, c);
}
