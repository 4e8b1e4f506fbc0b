// Developer micro-benchmark (GPU): does a wave's own vector-ALU work overlap with its own MFMAs on gfx950?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_overlap.hip -o /tmp/mfma_overlap && /tmp/mfma_overlap
// One workgroup per CU (256 of them), W waves per SIMD (W = 1, 2, 4), each wave runs ITER iterations of
//   [ 1 x v_mfma_f32_32x32x16_bf16 (32 cycles of matrix pipe) + V independent v_fma_f32 ]
// with the MFMAs on NACC rotating accumulators (no back-to-back dependence) and the FMAs on their own registers.
// If MFMA and VALU of the SAME wave overlap, time(V) stays flat until 4 * V cycles exceed the 32-cycle MFMA; if they serialize,
// time grows as 32 + 4 * V per iteration from V = 0 on. With W >= 2 the other wave fills the pipe either way.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V, int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float s) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 A, B;
  for (int e = 0; e < 8; ++e) { A[e] = (__bf16)(float)(threadIdx.x & 7); B[e] = (__bf16)(float)(e + 1); }
  float f[V > 0 ? V : 1];
  for (int v = 0; v < (V > 0 ? V : 1); ++v) f[v] = (float)threadIdx.x + v;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int a = 0; a < NACC; ++a) {
      acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[a], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < V; ++v) f[v] = __builtin_fmaf(f[v], s, 1.0f);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float t = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
  for (int v = 0; v < (V > 0 ? V : 1); ++v) t += f[v];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int V>
static void run(int wavesPerSimd, float* d_out) {
  const int iters = 2000, NACC = 4;
  const int threads = 64 * 4 * wavesPerSimd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<V, NACC>), dim3(256), dim3(threads), 0, 0, d_out, iters, 1.0001f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V, NACC>), dim3(256), dim3(threads), 0, 0, d_out, iters, 1.0001f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma_per_simd = (double)iters * NACC * wavesPerSimd;
  printf("waves/SIMD %d  VALU per MFMA %2d: %7.3f ms  = %6.1f ns per MFMA slot per SIMD (32-cycle MFMA at 2.1-2.4 GHz = 13-15 ns)\n", wavesPerSimd, V, ms,
         ms * 1e6 / mfma_per_simd);
}

int main() {
  float* d; hipMalloc(&d, 256 * 1024 * sizeof(float));
  for (int w : {1, 2, 4}) {
    run<0>(w, d); run<2>(w, d); run<4>(w, d); run<8>(w, d); run<12>(w, d); run<16>(w, d); run<24>(w, d);
  }
  return 0;
}
