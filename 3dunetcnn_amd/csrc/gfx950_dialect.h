// Dialect header for the kernels in this directory.
// Product build: hipcc --offload-arch=gfx950 (CDNA4 only; no other backend is supported).
// MI355_EMU is defined ONLY by tools/emu/build_emu.sh (CPU test infrastructure, see tools/emu/emu.h).
#pragma once

#ifdef MI355_EMU
#include "emu.h"
#include <cstring>
#include <cstdio>
#else
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x2_f32: exact f32 in / f32 accumulate, 64 cycles per SIMD (157 TFLOP/s chip peak).
// lane l supplies A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31];
// lane holds D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] for r in [0,16).
#define MFMA_32x32x2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define DYN_LDS(name) extern __shared__ __attribute__((aligned(16))) float name[]
// register budget: the allocator must fit at least n waves per SIMD (512 unified registers / n, accumulators included)
#define MIN_WAVES_PER_SIMD(n) __attribute__((amdgpu_waves_per_eu(n)))
// exactly one wave per SIMD: the wave owns a lane's whole 512-entry register file (256 architectural VGPRs + 256 AGPRs)
#define ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
// keep a value in the accumulation half of the register file from here on (an MFMA reads its A / B operands from AGPRs directly; without
// the pin hipcc parks such values there as spill space and copies them back with v_accvgpr_read before every use)
#define PIN_IN_AGPR(v) asm volatile("" : "+a"(v))
// ... and in the architectural half: accumulators of a kernel whose AGPRs are full of pinned operands (left to itself hipcc may put the
// MFMA results there too and then spills the operands)
#define PIN_IN_VGPR(v) asm volatile("" : "+v"(v))
// a per-lane condition as a 64-bit wave mask in scalar registers, and back: `LANE_IN_MASK(m)` is this lane's bit of a UNIFORM mask, used
// directly as the select / branch condition (no vector instruction tests it)
typedef unsigned long long LaneMask;
#define LANE_MASK(cond) __builtin_amdgcn_ballot_w64(cond)
#define LANE_IN_MASK(m) __builtin_amdgcn_inverse_ballot_w64(m)
// a wave-uniform value (pointer, offset) kept in scalar registers from here on: what is added to it afterwards stays a lane offset
#define PIN_IN_SGPR(v) asm volatile("" : "+s"(v))
// LDS-DMA, 16 bytes per lane: LDS[lds_wave_base + 16 * lane] = *(16 bytes at this lane's global address); no register is written, the
// instruction counts on the VM counter like a load (global_load_lds_dwordx4, LDS base in M0, wave-uniform). Issued from inline asm ON
// PURPOSE: hipcc cannot tell which LDS bytes a DMA writes once the addresses are computed (one dynamic array, run-time offsets) and puts
// `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY address -- every request would be waited for at once. Hidden from the
// compiler, the requests are ordered by the counted waits the kernel writes itself (WAIT_VMCNT_LGKM0) -- which therefore must also
// cover the compiler's own view: no ordinary global load may be outstanding across such a DMA's lifetime that hipcc waits for by count.
__device__ __forceinline__ unsigned lds_byte_address(const float* p) {
  return (unsigned)reinterpret_cast<unsigned long long>((const __attribute__((address_space(3))) float*)p);
}
__device__ __forceinline__ void glds16(const void* src, float* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte_address(lds_wave_base));
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(m0v) : "memory");
}
// ... from (uniform base in scalar registers) + (32-bit lane byte offset)
__device__ __forceinline__ void glds16_uniform_base(const void* ubase, unsigned lane_off, float* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte_address(lds_wave_base));
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_off), "s"(ubase), "s"(m0v) : "memory");
}
// s_waitcnt vmcnt(n) lgkmcnt(0): at most n of this wave's VMEM instructions (loads, LDS-DMA) outstanding -- they retire in order -- and
// its LDS instructions done; s_barrier without the fence of __syncthreads() (which waits vmcnt(0) while a DMA is in flight)
#define WAIT_VMCNT_LGKM0(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (((n) >> 4) << 14) | 0x0070)
#define RAW_BARRIER() __builtin_amdgcn_s_barrier()
// The waitcnt and barrier builtins carry no memory semantics at IR level, and the DMA they order is hidden in inline asm: this
// compiler-only fence (no instruction) keeps hipcc from moving an LDS access of the next phase above the wait + barrier that publish
// its data, or one of this phase below them
#define COMPILER_FENCE() asm volatile("" ::: "memory")
// v_permlane32_swap_b32: lanes 32..63 of `upper_from` trade values with lanes 0..31 of `lower_from` (lane 32 + k of the first <-> lane k
// of the second): two registers of a lane pair (l, l + 32) become a 2 x 2 transpose without touching the LDS
__device__ __forceinline__ void permlane32_swap(float& upper_from, float& lower_from) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(upper_from), __float_as_uint(lower_from), false, false);
  upper_from = __uint_as_float(r[0]); lower_from = __uint_as_float(r[1]);
}
// s_setprio: issue priority of this wave among the waves of its SIMD (0 lowest .. 3)
#define SET_PRIO(n) __builtin_amdgcn_s_setprio(n)
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
// v_mfma_f32_32x32x16_bf16: lane l supplies A[i=l&31][k=8*(l>>5)+0..7] and B[k=8*(l>>5)+0..7][j=l&31] as 8 packed bf16
// (16 bytes, carried here as uint4); same C/D map as the f32 form. 32 cycles per SIMD = 16x the f32 MFMA rate.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define MFMA_32x32x16_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)
// two floats -> two bf16 (round to nearest even) packed in one dword (first argument in the low half): v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// v_mfma_f32_32x32x16_f16: the same operand / result maps with IEEE half operands (what torch's CUDA autocast runs convolutions in)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
#define MFMA_32x32x16_F16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
// two floats -> two fp16 (round to nearest even, as tensor.half()) packed in one dword: v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned pack_f16x2(float lo, float hi) {
  f32x2_t v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
}
// two fp32 lanes per VGPR pair: v_pk_fma_f32 (the vector ALU's full fp32 rate needs the packed form)
typedef f32x2_t pkf2;
__device__ __forceinline__ pkf2 make_pkf2(float x, float y) { pkf2 v = {x, y}; return v; }
__device__ __forceinline__ pkf2 pk_fma(pkf2 a, pkf2 b, pkf2 c) { return __builtin_elementwise_fma(a, b, c); }
#define LAUNCH(kernel, grid, block, lds, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (lds), (hipStream_t)(stream), __VA_ARGS__)
#define LAUNCH_CHECK() (hipGetLastError() == hipSuccess ? 0 : -3)
// pins the instruction order across this point (keeps software-prefetch loads ahead of the MFMA block they overlap with)
#define SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// s_sleep: n * 64 clocks (n <= 127)
#define SLEEP_64CLK(n) __builtin_amdgcn_s_sleep(n)
// kernels that need more than the default 64 KiB dynamic LDS window (gfx950 has 160 KiB per CU)
#define SET_MAX_DYN_LDS(kernel, bytes) \
  do { if ((bytes) > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); } while (0)
#endif

// a value the program knows to be the same in all 64 lanes (derived from the wave index): moved to a scalar register, so that a branch on
// it is a scalar branch and a select on it disappears
#ifdef MI355_EMU
#define WAVE_UNIFORM(x) (x)
#else
#define WAVE_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)
#endif

#define MI355_OK 0
#define MI355_EINVAL (-1)
#define MI355_EUNSUPPORTED (-2)
#define MI355_ELAUNCH (-3)
#define MI355_EWORKSPACE (-4)

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// f(integral_constant<int, I>) for I = B .. N - 1, unrolled at compile time (C++17: no templated lambdas)
#include <type_traits>
template <int I, int N, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>()); static_for<I + 1, N>(f); }
}

// 16-bit operand type of the matrix pipe: bf16 (the split-operand modes and MI355_PREC_BF16) or fp16 (MI355_PREC_F16)
template <bool F16> __device__ __forceinline__ unsigned pack_lp2(float lo, float hi) { return F16 ? pack_f16x2(lo, hi) : pack_bf16x2(lo, hi); }
template <bool F16> __device__ __forceinline__ f32x16 mfma_lp(const uint4& a, const uint4& b, f32x16 c) {
  if constexpr (F16) return MFMA_32x32x16_F16(a, b, c);
  else return MFMA_32x32x16_BF16(a, b, c);
}
// value of the bf16 stored in the low / high half of a packed dword
__device__ __forceinline__ float bf16lo_to_f32(unsigned p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf16hi_to_f32(unsigned p) { return __uint_as_float(p & 0xffff0000u); }
// value of the IEEE fp16 stored in the low / high half of a packed dword (v_cvt_f32_f16)
#ifdef MI355_EMU
static inline float f16lo_to_f32(unsigned p) { return emu_h2f(p & 0xffffu); }
static inline float f16hi_to_f32(unsigned p) { return emu_h2f(p >> 16); }
#else
__device__ __forceinline__ float f16lo_to_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p & 0xffffu)); }
__device__ __forceinline__ float f16hi_to_f32(unsigned p) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(p >> 16)); }
#endif
