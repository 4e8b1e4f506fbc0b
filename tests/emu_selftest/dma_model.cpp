// TEST INFRASTRUCTURE: self-test of the emulator's LDS-DMA model (tools/emu/emu.h: glds16 / WAIT_VMCNT_LGKM0 / RAW_BARRIER), built and run by
// tests/test_emu_dma_model.py. One 128-thread block (two waves). Checks, under the default "late" model:
//   1. a request is NOT visible in LDS before a counted wait retires it (a kernel that reads too early must see stale data here);
//   2. vmcnt(n) leaves exactly the wave's n most recent INSTRUCTIONS outstanding -- also for a lane that took no part in some of them;
//   3. vmcnt(0) retires everything; under MI355_EMU_DMA=early a request lands at once.
#define MI355_EMU
#include "emu.h"
#include <cstdio>

static int fails = 0;
#define CHECK(c) do { if (!(c)) { ++fails; fprintf(stderr, "FAIL line %d tid %d: %s\n", __LINE__, emu::flat_tid(), #c); } } while (0)

static float src[4][128 * 4];      // 4 source images of 128 x 16 bytes

static void kernel(bool early) {
  DYN_LDS(lds);                    // 4 destination images [img][128 slots x 4 floats]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) lds[(i * 128 + tid) * 4 + e] = -1.f;
  __syncthreads();
  auto dst = [&](int img) { return lds + (img * 128 + wave * 64) * 4; };            // wave-uniform base, + 4 * lane by the DMA
  auto got = [&](int img) { return lds[(img * 128 + tid) * 4] == src[img][tid * 4]; };
  // instruction 1: every lane; instruction 2: lanes < 51 only; instruction 3: every lane
  glds16(&src[0][tid * 4], dst(0));
  if (lane < 51) glds16(&src[1][tid * 4], dst(1));
  glds16(&src[2][tid * 4], dst(2));
  CHECK(early ? got(0) : !got(0));                                                 // 1.
  WAIT_VMCNT_LGKM0(2);                                                             // instruction 1 retired, 2 and 3 outstanding
  CHECK(got(0));
  if (!early) { CHECK(!got(1)); CHECK(!got(2)); }
  WAIT_VMCNT_LGKM0(1);                                                             // 2 retired -- a lane >= 51 has nothing to land, and its
  if (lane < 51) CHECK(got(1)); else CHECK(lds[(1 * 128 + tid) * 4] == -1.f);      //    instruction 3 must still be outstanding (2.)
  if (!early) CHECK(!got(2));
  glds16(&src[3][tid * 4], dst(3));                                                // instruction 4
  WAIT_VMCNT_LGKM0(1);                                                             // 3 retired, 4 outstanding
  CHECK(got(2));
  if (!early) CHECK(!got(3));
  RAW_BARRIER();
  WAIT_VMCNT_LGKM0(0);                                                             // 3.
  CHECK(got(3));
}

int main() {
  for (int i = 0; i < 4; ++i) for (int k = 0; k < 512; ++k) src[i][k] = (float)(1000 * (i + 1) + k);
  const char* m = getenv("MI355_EMU_DMA");
  const bool early = m && !strcmp(m, "early");
  emu::launch(emu_dim3{1, 1, 1}, emu_dim3{128, 1, 1}, 4 * 128 * 16, [=]() { kernel(early); });
  printf("%s model: %d failures\n", early ? "early" : "late", fails);
  return fails ? 1 : 0;
}
