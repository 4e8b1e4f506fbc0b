// Micro-benchmark for round 5 (tools/NEXT.md candidate 1): what does ds_read_b64_tr_b16 hand each lane?
// The LDS is filled with its own 16-bit element index (element i holds the value i), one wave issues ONE transpose read with a chosen
// per-lane byte address, and the 4 elements every lane received are printed. Three address patterns:
//   A  every lane the same address (0): does the instruction add a lane offset of its own?
//   B  lane l -> byte 8 * l: consecutive 8-byte runs (lane l's own 4 elements 4 l .. 4 l + 3 in a plain ds_read_b64)
//   C  lane l -> byte 32 * (l & 15) + 8 * (l >> 4): 16 rows of 16 elements, lane group g = l >> 4 takes columns 4 g .. 4 g + 3
//      (the [voxel][channel] tile a weight-gradient producer would write as loaded: row = voxel, 16 channels of 2 bytes)
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/micro/tr_read_probe.hip -o tools/micro/tr_read_probe && tools/micro/tr_read_probe
#include <hip/hip_runtime.h>
#include <cstdio>

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

__global__ void probe(int pattern, unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int byte = 0;
  if (pattern == 1) byte = 8 * l;
  if (pattern == 2) byte = 32 * (l & 15) + 8 * (l >> 4);
  // the builtin takes an LDS (address space 3) pointer to 4 x 16-bit; the raw bits come back whatever the nominal element type
  typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 llvm_bf16x4_t;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  auto* p = (__attribute__((address_space(3))) llvm_bf16x4_t*)((char*)lds + byte);
#pragma clang diagnostic pop
  const llvm_bf16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(p);
  const short4v v = __builtin_bit_cast(short4v, r);
  for (int j = 0; j < 4; ++j) out[4 * l + j] = (unsigned short)v[j];
}

int main() {
  unsigned short* d; unsigned short h[256];
  (void)hipMalloc(&d, sizeof(h));
  const char* names[3] = {"A: all lanes address 0", "B: lane l -> byte 8 l", "C: lane l -> byte 32 (l & 15) + 8 (l >> 4)"};
  for (int pat = 0; pat < 3; ++pat) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, pat, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %s\n", names[pat]);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "");
  }
  (void)hipFree(d);
  return 0;
}
