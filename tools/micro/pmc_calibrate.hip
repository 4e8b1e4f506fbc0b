// Developer micro-benchmark (GPU): what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access patterns of this library's kernels?
//   hipcc --offload-arch=gfx950 -O3 tools/micro/pmc_calibrate.hip -o tools/micro/pmc_calibrate
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/pmc_calibrate     (and again with WRITE_SIZE)
// Every kernel moves exactly 1 GiB (256 Mi floats) once, far beyond the 256 MB Infinity Cache and the L2:
//   read_f4     each lane loads 16 contiguous bytes, a wave 1 KiB contiguous            (staging loads of every conv kernel)
//   read_f1     each lane loads 4 bytes, 32 lanes = 128 contiguous bytes, the two half-waves 2 KiB apart   (residual / normalised-tensor
//               reads of the 16-bit plane-ring epilogues)
//   write_f4    each lane stores 16 contiguous bytes                                     (epilogue of conv3d_wino2d_w8, streaming kernels)
//   write_f1    each lane stores 4 bytes, 32 lanes = 128 contiguous bytes, half-waves 2 KiB apart   (epilogue of the 16-bit conv kernels)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_f4(const float4* x, float* sink, size_t n4) {
  float acc = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = x[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void read_f1(const float* x, float* sink, size_t n) {
  float acc = 0.f;
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  // a wave covers 1024 floats per step: rows of 32 floats; half-wave h takes rows h, h + 16 apart pattern -> (row = 2 * k + half)
  for (size_t blk = wave; blk * 1024 < n; blk += nwaves)
    for (int k = 0; k < 16; ++k) acc += x[blk * 1024 + (size_t)(k + 16 * half) * 32 + li];
  if (acc == 12345.678f) sink[0] = acc;
}
__global__ void write_f4(float4* y, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = make_float4(1.f, 2.f, 3.f, 4.f);
}
__global__ void write_f1(float* y, size_t n) {
  const int lane = threadIdx.x & 63, half = lane >> 5, li = lane & 31;
  const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t blk = wave; blk * 1024 < n; blk += nwaves)
    for (int k = 0; k < 16; ++k) y[blk * 1024 + (size_t)(k + 16 * half) * 32 + li] = (float)k;
}
int main() {
  const size_t n = (size_t)256 << 20;                       // floats = 1 GiB
  float *x, *y, *sink;
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&sink, 64);
  hipMemset(x, 0, n * 4); hipMemset(y, 0, n * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(read_f4, dim3(4096), dim3(256), 0, 0, (const float4*)x, sink, n / 4);
    hipLaunchKernelGGL(read_f1, dim3(4096), dim3(256), 0, 0, x, sink, n);
    hipLaunchKernelGGL(write_f4, dim3(4096), dim3(256), 0, 0, (float4*)y, n / 4);
    hipLaunchKernelGGL(write_f1, dim3(4096), dim3(256), 0, 0, y, n);
  }
  hipDeviceSynchronize();
  printf("each kernel moved exactly 1 GiB (1048576 KiB)\n");
  return 0;
}
