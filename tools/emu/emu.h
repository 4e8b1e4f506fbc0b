// TEST INFRASTRUCTURE ONLY -- never part of the product path.
//
// A tiny CPU emulator for the restricted HIP dialect used by 3dunetcnn_amd/csrc/*.hip.
// It exists so the kernels' index arithmetic (LDS halo tiles, MFMA fragment maps,
// epilogues) can be exercised by `pytest -m "not gpu"` in a container without a GPU.
// The product library (libmi355unet3d.so) is built by hipcc for gfx950 and never
// includes this file; only tools/emu/build_emu.sh defines MI355_EMU.
//
// Model: one OS thread runs one workgroup at a time; every work-item is a ucontext
// fiber; __syncthreads() and the wave-level collectives (shuffles, MFMA) are
// counting barriers that yield to the round-robin scheduler. MFMA is emulated as the
// k-ordered fmaf chain the hardware implements (MI355X guide: "bit-for-bit a k-ordered
// f32 fmaf chain"), so emulated results match the GPU's bitwise for the f32 MFMA path.
#pragma once
#include <ucontext.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>
#include <atomic>
#include <algorithm>

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned x_=1, unsigned y_=1, unsigned z_=1):x(x_),y(y_),z(z_){} };
typedef emu_dim3 dim3;
typedef void* hipStream_t;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(...)
#define __shared__ static thread_local

struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x,y,z,w}; }
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x,y,z,w}; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float2 make_float2(float x, float y) { return float2{x,y}; }
typedef float f32x16 __attribute__((vector_size(64)));
typedef float f32x4  __attribute__((vector_size(16)));

namespace emu {

struct Fiber {
  ucontext_t ctx;
  char* stack = nullptr;
  bool done = false;
};

struct BlockState {
  std::vector<Fiber> fibers;
  ucontext_t sched;
  int cur = -1;
  unsigned nthreads = 0;
  // block barrier
  unsigned bar_count = 0; unsigned bar_gen = 0;
  // wave barriers
  unsigned wbar_count[32]; unsigned wbar_gen[32];
  // exchange scratch
  double xchg[2048];
  float  mfma_a[2048]; float mfma_b[2048];
  uint4  mfma_a4[1024]; uint4 mfma_b4[1024];
  std::function<void()> body;
  char* dyn_lds = nullptr;
  // LDS-DMA model (glds16 / WAIT_VMCNT_LGKM0 below): per wave the log of DMA instructions (their wave-uniform LDS base) and of the
  // instruction count at each counted wait, both written by the wave's lane 0; per lane the requests not yet landed
  struct DmaReq { unsigned tag; float* dst; float v[4]; };
  std::vector<std::vector<float*>> dma_log;          // [wave] -> LDS base of instruction 1, 2, ...
  std::vector<std::vector<unsigned>> dma_wait_log;   // [wave] -> instructions issued when lane 0 reached its k-th wait
  std::vector<std::vector<DmaReq>> dma_pending;      // [thread], in issue order
  std::vector<unsigned> dma_cursor, dma_waits;       // [thread]: next log entry to match / waits executed
};

extern thread_local BlockState* g_bs;
extern thread_local emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void set_tid(int t) {
  g_threadIdx.x = t % g_blockDim.x;
  g_threadIdx.y = (t / g_blockDim.x) % g_blockDim.y;
  g_threadIdx.z = t / (g_blockDim.x * g_blockDim.y);
}

inline void yield() {
  BlockState* bs = g_bs;
  int me = bs->cur;
  swapcontext(&bs->fibers[me].ctx, &bs->sched);
}

inline void block_barrier() {
  BlockState* bs = g_bs;
  unsigned gen = bs->bar_gen;
  if (++bs->bar_count == bs->nthreads) { bs->bar_count = 0; bs->bar_gen++; return; }
  while (bs->bar_gen == gen) yield();
}

inline int flat_tid() {
  return g_threadIdx.x + g_blockDim.x * (g_threadIdx.y + g_blockDim.y * g_threadIdx.z);
}

inline void wave_barrier() {
  BlockState* bs = g_bs;
  int w = flat_tid() / 64;
  unsigned wsize = std::min(64u, bs->nthreads - w * 64);
  unsigned gen = bs->wbar_gen[w];
  if (++bs->wbar_count[w] == wsize) { bs->wbar_count[w] = 0; bs->wbar_gen[w]++; return; }
  while (bs->wbar_gen[w] == gen) yield();
}

void fiber_entry();

void run_block(BlockState& bs);

template <class F>
void launch(emu_dim3 grid, emu_dim3 block, size_t lds_bytes, F body) {
  // MI355_EMU_NOEXEC=1: a launch returns at once (tools/host_enqueue.py times the host side of a step -- Python, ctypes, the C dispatch --
  // at the real shapes without a GPU; outputs are garbage, nothing may check them)
  static const bool noexec = [] { const char* v = getenv("MI355_EMU_NOEXEC"); return v && v[0] == '1'; }();
  if (noexec) return;
  size_t nblocks = (size_t)grid.x * grid.y * grid.z;
  unsigned nthr = std::min<size_t>(std::max(1u, std::thread::hardware_concurrency()), nblocks);
  const char* env = getenv("MI355_EMU_THREADS");
  if (env) nthr = std::max(1, std::min<int>(atoi(env), (int)nblocks));
  std::atomic<size_t> next{0};
  auto worker = [&]() {
    BlockState bs;
    bs.nthreads = block.x * block.y * block.z;
    bs.fibers.resize(bs.nthreads);
    for (auto& f : bs.fibers) f.stack = (char*)malloc(128 * 1024);
    bs.dyn_lds = (char*)aligned_alloc(64, std::max<size_t>(64, (lds_bytes + 63) / 64 * 64));
    bs.body = body;
    g_bs = &bs;
    g_blockDim = block; g_gridDim = grid;
    for (;;) {
      size_t b = next.fetch_add(1);
      if (b >= nblocks) break;
      g_blockIdx.x = b % grid.x; g_blockIdx.y = (b / grid.x) % grid.y; g_blockIdx.z = b / ((size_t)grid.x * grid.y);
      run_block(bs);
    }
    for (auto& f : bs.fibers) free(f.stack);
    free(bs.dyn_lds);
    g_bs = nullptr;
  };
  if (nthr <= 1) { worker(); return; }
  std::vector<std::thread> ts;
  for (unsigned i = 0; i < nthr; ++i) ts.emplace_back(worker);
  for (auto& t : ts) t.join();
}

}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx  (emu::g_blockIdx)
#define blockDim  (emu::g_blockDim)
#define gridDim   (emu::g_gridDim)

static inline void __syncthreads() { emu::block_barrier(); }

#define DYN_LDS(name) float* name = reinterpret_cast<float*>(emu::g_bs->dyn_lds)

template <class T>
static inline T emu_shfl(T v, int src_lane) {
  emu::BlockState* bs = emu::g_bs;
  int t = emu::flat_tid();
  int wbase = (t / 64) * 64;
  static_assert(sizeof(T) <= 8, "shfl type");
  memcpy(&bs->xchg[t], &v, sizeof(T));
  emu::wave_barrier();
  T r; memcpy(&r, &bs->xchg[wbase + (src_lane & 63)], sizeof(T));
  emu::wave_barrier();
  return r;
}
template <class T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; return emu_shfl(v, (emu::flat_tid() % 64) ^ mask); }
template <class T> static inline T __shfl_down(T v, int d, int width = 64) { (void)width; int l = emu::flat_tid() % 64; return emu_shfl(v, l + d < 64 ? l + d : l); }
template <class T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return emu_shfl(v, src); }

// D = A(32x2) * B(2x32) + C ; lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// lane holds C[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] for r in [0,16).
static inline f32x16 emu_mfma_32x32x2(float a, float b, f32x16 c) {
  emu::BlockState* bs = emu::g_bs;
  int t = emu::flat_tid();
  int wbase = (t / 64) * 64, l = t % 64;
  bs->mfma_a[t] = a; bs->mfma_b[t] = b;
  emu::wave_barrier();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    acc = fmaf(bs->mfma_a[wbase + row], bs->mfma_b[wbase + col], acc);            // k = 0
    acc = fmaf(bs->mfma_a[wbase + 32 + row], bs->mfma_b[wbase + 32 + col], acc);  // k = 1
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}
#define MFMA_32x32x2(a, b, c) emu_mfma_32x32x2(a, b, c)

// v_pk_fma_f32: two independent fmaf per lane
struct pkf2 { float x, y; };
static inline pkf2 make_pkf2(float x, float y) { return pkf2{x, y}; }
static inline pkf2 pk_fma(pkf2 a, pkf2 b, pkf2 c) { return pkf2{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)}; }

// bf16 helpers (round to nearest even, as v_cvt_pk_bf16_f32)
static inline unsigned emu_f2bf(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7f800000u) == 0x7f800000u && (u & 0x007fffffu)) return (u >> 16) | 0x40u;   // NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return u >> 16;
}
static inline unsigned pack_bf16x2(float lo, float hi) { return emu_f2bf(lo) | (emu_f2bf(hi) << 16); }
// D = A(32x16) * B(16x32) + C, bf16 inputs: lane l supplies A[i=l&31][k=8*(l>>5)+e], B[k=8*(l>>5)+e][j=l&31], e in [0,8).
// Products of bf16 are exact in f32; the 16-term sum is formed in double and rounded once with C (the hardware's internal
// order is unspecified; this is test infrastructure for index logic, not a bitwise model of the bf16 MFMA).
static inline f32x16 emu_mfma_32x32x16_bf16(uint4 a, uint4 b, f32x16 c) {
  emu::BlockState* bs = emu::g_bs;
  int t = emu::flat_tid();
  int wbase = (t / 64) * 64, l = t % 64;
  bs->mfma_a4[t] = a; bs->mfma_b4[t] = b;
  emu::wave_barrier();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = (double)c[r];
    for (int h = 0; h < 2; ++h) {
      const unsigned* pa = &bs->mfma_a4[wbase + 32 * h + row].x;
      const unsigned* pb = &bs->mfma_b4[wbase + 32 * h + col].x;
      for (int e = 0; e < 4; ++e) {
        acc += (double)__uint_as_float(pa[e] << 16) * (double)__uint_as_float(pb[e] << 16);
        acc += (double)__uint_as_float(pa[e] & 0xffff0000u) * (double)__uint_as_float(pb[e] & 0xffff0000u);
      }
    }
    c[r] = (float)acc;
  }
  emu::wave_barrier();
  return c;
}
#define MFMA_32x32x16_BF16(a, b, c) emu_mfma_32x32x16_bf16(a, b, c)
// fp16 operands (v_cvt_pk_f16_f32: round to nearest even; v_mfma_f32_32x32x16_f16): same maps, same "exact products, one rounding" model
static inline unsigned emu_f2h(float f) {           // IEEE binary16, round to nearest even, subnormals and overflow to inf
  const unsigned u = __float_as_uint(f), sign = (u >> 16) & 0x8000u;
  const unsigned e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
  if (e == 0xffu) return sign | 0x7c00u | (m ? 0x200u : 0u);
  const int ex = (int)e - 127 + 15;
  if (ex >= 31) return sign | 0x7c00u;
  if (ex <= 0) {
    if (ex < -10) return sign;
    const unsigned mm = m | 0x800000u;                     // 24-bit significand
    const int shift = 14 - ex;                             // 14 .. 24
    unsigned h = mm >> shift;
    const unsigned rem = mm & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (h & 1u))) ++h;
    return sign | h;
  }
  unsigned h = ((unsigned)ex << 10) | (m >> 13);
  const unsigned rem = m & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;   // may carry into the exponent (and to inf): still the right value
  return sign | h;
}
static inline float emu_h2f(unsigned h) {
  const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  if (e == 0) return (sign ? -1.f : 1.f) * ldexpf((float)m, -24);
  if (e == 31) return __uint_as_float(sign | 0x7f800000u | (m << 13));
  return __uint_as_float(sign | ((e + 112u) << 23) | (m << 13));
}
static inline unsigned pack_f16x2(float lo, float hi) { return emu_f2h(lo) | (emu_f2h(hi) << 16); }
static inline f32x16 emu_mfma_32x32x16_f16(uint4 a, uint4 b, f32x16 c) {
  emu::BlockState* bs = emu::g_bs;
  int t = emu::flat_tid();
  int wbase = (t / 64) * 64, l = t % 64;
  bs->mfma_a4[t] = a; bs->mfma_b4[t] = b;
  emu::wave_barrier();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    double acc = (double)c[r];
    for (int h = 0; h < 2; ++h) {
      const unsigned* pa = &bs->mfma_a4[wbase + 32 * h + row].x;
      const unsigned* pb = &bs->mfma_b4[wbase + 32 * h + col].x;
      for (int e = 0; e < 4; ++e) {
        acc += (double)emu_h2f(pa[e] & 0xffffu) * (double)emu_h2f(pb[e] & 0xffffu);
        acc += (double)emu_h2f(pa[e] >> 16) * (double)emu_h2f(pb[e] >> 16);
      }
    }
    c[r] = (float)acc;
  }
  emu::wave_barrier();
  return c;
}
#define MFMA_32x32x16_F16(a, b, c) emu_mfma_32x32x16_f16(a, b, c)

static inline float atomicAdd(float* p, float v) {
  std::atomic_ref<float> r(*p); float old = r.load();
  while (!r.compare_exchange_weak(old, old + v)) {}
  return old;
}
static inline double atomicAdd(double* p, double v) {
  std::atomic_ref<double> r(*p); double old = r.load();
  while (!r.compare_exchange_weak(old, old + v)) {}
  return old;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

#define LAUNCH(kernel, grid, block, lds, stream, ...) \
  emu::launch((grid), (block), (lds), [=]() { kernel(__VA_ARGS__); })
#define LAUNCH_CHECK() 0
#define SET_MAX_DYN_LDS(kernel, bytes) do {} while (0)
#define SCHED_BARRIER() do {} while (0)
#define MIN_WAVES_PER_SIMD(n)
#define ONE_WAVE_PER_SIMD
#define PIN_IN_AGPR(v) ((void)0)
#define PIN_IN_VGPR(v) ((void)0)
#define PIN_IN_SGPR(v) ((void)0)
typedef unsigned long long LaneMask;
static inline LaneMask emu_lane_mask(bool c) { LaneMask m = 0; for (int l = 0; l < 64; ++l) m |= (LaneMask)(emu_shfl((int)c, l) & 1) << l; return m; }
#define LANE_MASK(cond) emu_lane_mask(cond)
#define LANE_IN_MASK(m) ((((m) >> (emu::flat_tid() % 64)) & 1ull) != 0)
// LDS-DMA (global_load_lds_dwordx4 + counted s_waitcnt vmcnt(n)). Two models, MI355_EMU_DMA=late (default) | early:
//   late : a request lands as LATE as the protocol allows -- only when a counted wait of its wave retires it (the wave's VM counter
//          retires in order: vmcnt(n) leaves the n most recent INSTRUCTIONS of the wave outstanding, whichever lanes took part in them).
//          A fragment read before the wait + barrier that publish its plane sees stale data: the RAW half of the protocol, including the
//          immediates of the waits, is checked here and not only on the GPU.
//   early: a request lands at once: a buffer restaged while another fiber still reads it shows up as a wrong result (the WAR half).
// Wave-level instruction counting with per-lane fibers: lane 0 of a wave takes part in every DMA instruction and every wait of the
// kernels that use this (it runs first between two barriers); it logs each instruction's wave-uniform LDS base and, at each wait, how
// many instructions the wave has issued. Another lane finds the instruction it is executing by that base (forward from its last match).
static inline bool emu_dma_early() { static const bool e = [] { const char* v = getenv("MI355_EMU_DMA"); return v && !strcmp(v, "early"); }(); return e; }
static inline void glds16(const void* src, float* lds_wave_base) {
  const int t = emu::flat_tid(), l = t % 64, w = t / 64;
  if (emu_dma_early()) { memcpy(lds_wave_base + 4 * l, src, 16); return; }
  emu::BlockState* bs = emu::g_bs;
  auto& log = bs->dma_log[w];
  unsigned tag;
  if (l == 0) { log.push_back(lds_wave_base); tag = (unsigned)log.size(); bs->dma_cursor[t] = tag; }
  else {
    unsigned c = bs->dma_cursor[t];
    for (int spins = 0;; ++spins) {                          // (the last fiber to reach a barrier runs on before the wave's lane 0: let that one catch up)
      while (c < log.size() && log[c] != lds_wave_base) ++c;
      if (c < log.size()) break;
      if (spins > 1000000) { fprintf(stderr, "emu: LDS-DMA by lane %d of wave %d has no matching instruction of lane 0\n", l, w); abort(); }
      emu::yield();
    }
    tag = c + 1; bs->dma_cursor[t] = tag;
  }
  emu::BlockState::DmaReq r; r.tag = tag; r.dst = lds_wave_base + 4 * l; memcpy(r.v, src, 16);
  bs->dma_pending[t].push_back(r);
}
static inline void glds16_uniform_base(const void* ubase, unsigned lane_off, float* lds_wave_base) { glds16(reinterpret_cast<const char*>(ubase) + lane_off, lds_wave_base); }
static inline void emu_wait_vmcnt(unsigned n) {
  if (emu_dma_early()) return;
  emu::BlockState* bs = emu::g_bs;
  const int t = emu::flat_tid(), l = t % 64, w = t / 64;
  auto& wl = bs->dma_wait_log[w];
  const unsigned k = bs->dma_waits[t]++;
  if (l == 0) wl.push_back((unsigned)bs->dma_log[w].size());
  for (int spins = 0; k >= wl.size(); ++spins) {
    if (spins > 1000000) { fprintf(stderr, "emu: counted wait %u of lane %d, wave %d: lane 0 never executed it\n", k, l, w); abort(); }
    emu::yield();
  }
  const unsigned issued = wl[k], retire = issued > n ? issued - n : 0;      // instructions 1 .. retire have landed
  auto& q = bs->dma_pending[t];
  size_t i = 0;
  for (; i < q.size() && q[i].tag <= retire; ++i) memcpy(q[i].dst, q[i].v, 16);
  q.erase(q.begin(), q.begin() + i);
}
#define WAIT_VMCNT_LGKM0(n) emu_wait_vmcnt(n)
#define RAW_BARRIER() __syncthreads()
#define COMPILER_FENCE() do {} while (0)
#define SET_PRIO(n) do {} while (0)
// v_permlane32_swap_b32: lane 32 + k of the first register <-> lane k of the second
static inline void permlane32_swap(float& upper_from, float& lower_from) {
  const int l = emu::flat_tid() % 64;
  const float got = l < 32 ? emu_shfl(lower_from, l + 32) : emu_shfl(lower_from, l - 32);      // what the partner lane holds in `lower_from`
  const float got2 = l < 32 ? emu_shfl(upper_from, l + 32) : emu_shfl(upper_from, l - 32);     // ... and in `upper_from`
  if (l < 32) lower_from = got2;        // lower lane: its `lower_from` <- partner's (lane l + 32) `upper_from`
  else upper_from = got;                // upper lane: its `upper_from` <- partner's (lane l - 32) `lower_from`
}
typedef uint4 u32x4_t;
#define SLEEP_64CLK(n) do {} while (0)
