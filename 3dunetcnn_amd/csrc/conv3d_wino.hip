// 3x3x3 stride-1 conv3d forward / dgrad in exact-type fp32 with FEWER multiplications: Winograd F(2x2, 3x3) in the (y, x) plane,
// direct along z (round 3; the default route of the eligible layers since profiles/r3_winograd_landing.txt).
//
// Same op as conv3d_fwd.hip (reference: unet3d/models/pytorch/classification/resnet.py:12-22 called from myronenko.py:17-21; the
// GroupNorm-apply + ReLU prologue and the bias / residual / Dropout3d-scale epilogue are fused the same way). Arithmetic per output
// voxel and (ci, co): 16 transform points per 2x2 outputs x 3 z-taps = 12 multiplications instead of 27. Numerics: the transform
// matrices have entries 0, +-1, +-1/2; measured on the CPU (tools/winograd_probe.py) and on the GPU (tests/test_wino_gpu.py,
// tests/test_launch_audit.py) the fp32 result is as close to an fp64 convolution as the direct fp32 kernel's (3e-7 .. 9e-7 of max |y|).
//
//   input transform   V = B^T d B    (4x4 input window d of one channel, B^T rows: d0-d2, d1+d2, d2-d1, d1-d3)
//   filter transform  U = G g G^T    (3x3 (dy,dx) slice g of one (co, ci, dz), done once per optimizer step by the pack kernel)
//   point-wise        M[p] = sum_{ci, dz} V[p][plane z + dz - 1][ci] * U[p][dz][ci][co]          <- the MFMA work, p = 0..15
//   output transform  Y = A^T M A    (2x2 outputs, A^T rows: m0+m1+m2, m1-m2-m3)
//
// Output tile of a workgroup = 2 z-planes x 8 (y) x 16 (x) voxels = per plane 4 x 8 = 32 Winograd tiles of 2x2 = one dimension of a
// 32x32 MFMA tile; the other = 32 output channels; K = 8 input channels per chunk (a lane supplies 4 of them as one 16-byte operand
// per MFMA quadruple, lane half = the other 4). A workgroup walks the 4 input planes its 2 output planes see; every plane chunk is staged
// once (haloed 10 x 18 voxels) and used by the 1-2 (output plane, dz) pairs that see it.
//
// History of the forms (all measured on MI355X; the losers were deleted): round 3 (profiles/r3_wino_forms.txt) a 4-wave workgroup with 8
// accumulator tiles per wave (layer set 33.3 ms), a z-marching 8-wave workgroup with three output planes in registers (36.9 ms: one
// workgroup per CU in lock step), the 8-wave tile kernel conv3d_wino2d_w8 with the transformed planes in LDS (28.7 ms; rounds 3-4);
// round 5 (profiles/r5_wino_r8_experiment.txt, r5_wino_d8.txt): conv3d_wino2d_r8, the same with register-generated fragments (a quarter
// fewer vector and half the LDS instructions per MFMA, 3 % SLOWER: the kernel is not bound by its instruction count), and
// conv3d_wino2d_d8 below, r8 with every global load of the main loop replaced by an LDS-DMA requested two phases ahead.
#include "gfx950_dialect.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"
#include "pack_values.h"
#include "wino_common.h"

#ifndef WINO_ZBRICK
#define WINO_ZBRICK 8             // z tiles per brick of the workgroup order (A/B: -DWINO_ZBRICK=1 = x fastest)
#endif
#ifndef WINO_ABL
// developer ablations of conv3d_wino2d_d8 (tools/build_variant.sh ... -DWINO_ABL=mask; results wrong by construction, timing only):
// 1 no input requests, 2 no weight requests, 4 weight fragments from constants instead of the slab, 32 the loop never waits for a request,
// 64 / 128 every input / weight lane fetches the same 16 bytes
#define WINO_ABL 0
#endif

// =====================================================================================================================================
// conv3d_wino2d_d8 (round 5): 512 threads = 8 waves on a 2 (z) x 8 x 16 voxel x 32 channel tile, wave w = (point row i = w >> 1, j half
// jh = w & 1) owns two of the 16 transform points for both output planes (4 accumulator tiles = 64 registers; 128 per wave, four waves
// per SIMD, two workgroups per CU). What bounded its predecessor conv3d_wino2d_w8 (transformed planes in LDS, register-staged loads one
// phase ahead) was not the instruction count -- a quarter fewer vector and half the LDS instructions per MFMA changed nothing
// (profiles/r5_wino_r8_experiment.txt) -- but the global loads on a one-phase lead: a phase is 8-16 MFMAs per wave, the in-order VMEM
// counter ties the L2-hit weight request to the HBM-latency input request issued before it, and more lead needs registers the kernel
// does not have. So here NO global load of the main loop touches a register (profiles/r5_wino_d8.txt):
//  * input planes AND weights travel by LDS-DMA (global_load_lds_dwordx4, 16 bytes per lane, destination = wave base + 16 x lane):
//    a ring of 4 staged plane chunks (6.5 KB each; a request goes out four phases before the plane's fragments are generated) and 3
//    weight slabs W[dz] (16 points x 8 input x 32 output channels = 16 KB each); the waits are counted -- `s_waitcnt vmcnt(N)`, N = the
//    DMA instructions this wave issued in the current phase, everything older has landed: the counter retires in order, so every
//    wave issues the SAME number of requests per phase, whatever its lanes fetch -- and the barriers raw s_barrier (__syncthreads()
//    would drain the queue). The requests sit BETWEEN the MFMA pairs: issuing one occupies the wave for 60-190 cycles;
//  * the staged layout [quad][row][even columns | odd columns] is filled in lane order, as the DMA requires (slot = 51 x wave + lane, the
//    slot picks the voxel), and is conflict-free for the reads below: the 16 lanes of every ds_read_b128 lane group hit 16 different
//    16-byte bank groups (tile (ty, tx), column class c -> group tx + 40 ty + const);
//  * A fragments generated in registers straight from the staged plane: a wave's two points share the point row i (window rows ra,
//    rb: R = d[ra] + beta d[rb]) and three adjacent window columns X0 X1 X2 (q0 = R[X0] - R[X2], q1 = R[X1] +- R[X2]): 6
//    ds_read_b128 + 20 fp32 instructions per phase and lane, one phase ahead of the MFMAs that use them. No transformed planes in
//    LDS -- which is what makes room for the weight slabs. beta, the rows and the column order are wave constants:
//        i = 0: d0 - d2   i = 1: d1 + d2   i = 2: d2 - d1   i = 3: d1 - d3                  (ra, rb, beta) = (0,2,-) (1,2,+) (2,1,-) (1,3,-)
//        jh = 0 (j = 0, 1): X = columns (0, 1, 2): q0 = X0 - X2 = V[i][0], q1 = X1 + X2 = V[i][1]
//        jh = 1 (j = 3, 2): X = columns (3, 2, 1): q0 = X0 - X2 = -V[i][3], q1 = X1 - X2 = V[i][2]
//    (the accumulator of q0 holds -M[i][3] in the second half; the in-wave part of the output transform takes the sign for free:
//    b0 = a1, b1 = a0 - a1 instead of b0 = a0 + a1, b1 = a1); the main loop exists once per j half, so the column offsets are immediates;
//  * weight fragments read from the slab (2 ds_read_b128 per use, consecutive lanes = consecutive 16 bytes): those of a phase's first
//    use at the bottom of the phase before, those of a second use behind the first use's MFMAs;
//  * the norm prologue act(scale x + shift), which a DMA cannot apply on the way: the lane that requested a slot rewrites it in place
//    once its own counted wait says the slot has landed, at the bottom of the phase before the plane is first read;
//  * MFMA operands swapped (A = weights, B = input): a lane holds 4 CONSECUTIVE output channels of one tile per accumulator quad, so
//    the output-transform exchange is written with 8 ds_write_b128 per plane (rows padded to 36 floats: conflict-free) and read
//    voxel-major with ds_read_b128; residual, normalised tensor (dgrad) and output move as 16-byte accesses; fused norm statistics as
//    plain sums about a common shift, the 8 waves merged by Chan's formula once (`wino_fuse_records`);
//  * workgroups are dealt so that an XCD gets a contiguous range of spatial tiles of one channel tile, walked in bricks of 8 z tiles,
//    then x, then y (z neighbours share two of their four input planes, x neighbours the halo, in that XCD's L2).
__device__ const float wino_zero16[4] = {0.f, 0.f, 0.f, 0.f};
template <int INMODE, int FUSE>
__global__ __launch_bounds__(512) MIN_WAVES_PER_SIMD(4) void conv3d_wino2d_d8(WinoArgs a) {
  constexpr int TZ = 2, TY = 8, TX = 16, HY = TY + 2;
  constexpr int KC = 8;
  // staged plane chunk: [channel quad][row][even columns | odd columns] in 16-byte groups. (Measured and not kept: [row][position][quad],
  // the two quads of a voxel in adjacent request lanes = one cache line per lane pair, 2-way bank conflicts of the fragment reads:
  // layer set -0.7 %, profiles/r5_wino_d8.txt.)
  constexpr int RS = 20;                                   // 16-byte groups per staged row: even columns 0..8 | pad | odd columns 10..18 | pad
  constexpr int QS = HY * RS + 4;                          // groups per channel quad (+ 4: the two quads of a voxel land on different banks)
  constexpr int XSF = 2 * QS * 4;                          // floats of one staged plane chunk (408 slots of 16 bytes = 8 waves x 51 lanes)
  constexpr int WSF = 16 * 2 * 32 * 4;                     // floats of one weight slab
  constexpr int PS = 36, PW = 2 * 32 * PS;                 // exchange: [wave][b][tile][32 channels + 4 pad]
  static_assert(XSF == 8 * 51 * 4, "one DMA instruction per wave fills a staged plane chunk");
  DYN_LDS(lds);
  float* xs = lds;                                         // ring of 4 staged plane chunks
  float* ws = lds + 4 * XSF;                               // 3 weight slabs
  float* P = lds;                                          // epilogue: output-transform exchange (reuses everything)
  float* prm = lds + 4 * XSF + 3 * WSF;                    // norm prologue of this sample: scale | shift | slope, CinP each
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  // workgroup -> (channel tile, spatial tile): an XCD (consecutive workgroup ids go round-robin to the 8 XCDs) gets a contiguous range of spatial tiles of one channel tile,
  // walked in bricks of WINO_ZBRICK z tiles, then x, y)
  int b = blockIdx.x, cot;
  {
    const int nct = a.coTiles, S = gridDim.x / nct, ng = nct < 8 && 8 % nct == 0 ? 8 / nct : 0;
    if (ng > 0 && S % ng == 0) {
      const int x = b & 7;
      cot = x % nct;
      b = (x / nct) * (S / ng) + (b >> 3);
    } else {
      cot = b % nct; b /= nct;
    }
  }
  int tz;
  if (a.tilesZ % WINO_ZBRICK == 0) {
    const int zi = b % WINO_ZBRICK; b /= WINO_ZBRICK;
    const int txi = b % a.tilesX; b /= a.tilesX;
    const int tyi = b % a.tilesY; b /= a.tilesY;
    const int zbk = b % (a.tilesZ / WINO_ZBRICK); b /= (a.tilesZ / WINO_ZBRICK);
    tz = zbk * WINO_ZBRICK + zi;
    b = (b * a.tilesY + tyi) * a.tilesX + txi;
  } else {
    const int txi = b % a.tilesX, r1 = b / a.tilesX;
    const int tyi = r1 % a.tilesY, r2 = r1 / a.tilesY;
    tz = r2 % a.tilesZ;
    b = ((r2 / a.tilesZ) * a.tilesY + tyi) * a.tilesX + txi;
  }
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = tz * TZ;
  const int n = b;
  const int co_base = cot * 32;

  f32x16 acc[TZ][2];                                       // [output plane][q]
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[oz][q][r] = 0.f;

  // ---- staging by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, LDS destination = wave base + 16 x lane) ----
  // Input: slot s = 51 wave + lane (lane < 51) of a staged plane chunk [quad kq][row][20 positions]; the slot decides which (voxel, quad)
  // the lane fetches. EVERY slot is written by every request: a pad, a voxel outside the image, the missing second quad of a last
  // chunk (Cin % 8 == 4) and every slot of a plane outside the volume fetch 16 zero bytes (`wino_zero16`) -- the number of DMA
  // instructions a wave issues per phase is a constant, which is what the counted waits below rely on.
  const int slot = wave * 51 + (lane < 50 ? lane : 50);     // lanes 51..63 of a wave request nothing
  const int skq = slot / QS, sr = slot % QS, srow = sr / RS, spos = sr % RS;
  const int scol = spos < 9 ? 2 * spos : 2 * (spos - 10) + 1;
  const bool sunit = lane < 51 && srow < HY && spos != 9 && spos != 19;
  const int siy = ty0 - 1 + srow, six = tx0 - 1 + scol;
  const bool sin = sunit && siy >= 0 && siy < a.H && six >= 0 && six < a.W;
  const size_t xplane = (size_t)a.H * a.W * a.xld;
  const float* xn = a.x + (size_t)n * a.D * xplane;        // sample n
  const unsigned xoff = sin ? (unsigned)((siy * a.W + six) * a.xld + 4 * skq) * 4u : 0u;      // bytes inside the plane
  const LaneMask m_in = LANE_MASK(sin), m_in0 = LANE_MASK(sin && skq == 0);
  const float4* up4 = reinterpret_cast<const float4*>(a.up);
  const int CQ = a.CinP / 4;
  const size_t bstep = (size_t)CQ * a.CoutP;               // float4s between consecutive (point, dz) slabs of the weight pack
  const unsigned boff = (unsigned)(half * a.CoutP + co_base + li) * 16u;    // bytes: this lane's (channel quad half, output channel)
  // plane pz_ of the chunk at channel c0_ -> ring buffer pz_
  auto dma_in = [&](int c0_, int pz_) {
    const int iz = tz0 - 1 + pz_;
    const int izc = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1);
    const LaneMask m = iz >= 0 && iz < a.D ? (c0_ + 8 <= a.Cin ? m_in : m_in0) : (LaneMask)0;      // uniform
    const char* real = reinterpret_cast<const char*>(xn + (size_t)izc * xplane + c0_) + xoff;
#if WINO_ABL & 64
    const char* src = reinterpret_cast<const char*>(wino_zero16); (void)real; (void)m;      // timing only: every lane fetches the same 16 bytes
#else
    const char* src = LANE_IN_MASK(m) ? real : reinterpret_cast<const char*>(wino_zero16);
#endif
#if !(WINO_ABL & 1)
    if (lane < 51) glds16(src, xs + pz_ * XSF + wave * (51 * 4));
#else
    (void)src;
#endif
  };
  // Norm prologue (INMODE = MI355_IN_AFFINE_ACT): a DMA cannot apply act(scale x + shift) on the way, so the lane that requested a slot
  // rewrites it in place once its request has landed (its own counted wait at the end of the previous phase: no barrier needed for its
  // own slot), at the bottom of the phase before the one whose fragment generation reads the plane -- behind the wave's last MFMA,
  // while the matrix pipe drains and the wave's requests of this phase are still on their way; the barrier at the end of the phase
  // publishes the result. Slots that
  // fetched zeros (pads, halo outside the image) are left alone: the padding of the ACTIVATED tensor is zero.
  auto activate = [&](int c0_, int pz_) {
    if (INMODE != MI355_IN_AFFINE_ACT) return;
    const int iz = tz0 - 1 + pz_;
    const LaneMask m = iz >= 0 && iz < a.D ? (c0_ + 8 <= a.Cin ? m_in : m_in0) : (LaneMask)0;      // as in dma_in
    if (LANE_IN_MASK(m)) {                                 // (m has no bit for lanes 51..63)
      const int c = c0_ + 4 * skq;
      float* p = xs + pz_ * XSF + slot * 4;
      float4 v = *reinterpret_cast<const float4*>(p);
      const float4 sc = *reinterpret_cast<const float4*>(prm + c), sh = *reinterpret_cast<const float4*>(prm + a.CinP + c);
      const float4 sl = *reinterpret_cast<const float4*>(prm + 2 * a.CinP + c);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
      *reinterpret_cast<float4*>(p) = v;
    }
  };
  // weights W[dz] of the chunk at c0_ -> slab dz: [point 16][quad half 2][output channel 32] x 16 bytes, unit u = 512 j + tid
  auto dma_w = [&](int c0_, int dz) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const float4* src = up4 + (size_t)(c0_ / 4) * a.CoutP + (size_t)((8 * j + wave) * 3 + dz) * bstep;      // uniform
#if WINO_ABL & 128
      glds16_uniform_base(wino_zero16, 0u, ws + dz * WSF + j * 2048 + wave * 256); (void)src;      // timing only
#elif !(WINO_ABL & 2)
      glds16_uniform_base(src, boff, ws + dz * WSF + j * 2048 + wave * 256);
#else
      (void)src;
#endif
    }
  };
  // A-fragment generation (as conv3d_wino2d_r8): wave constants (point row i, j half jh), the lane's two row addresses
  const int pi = wave >> 1, jh = wave & 1;
  const int ra = pi == 0 ? 0 : (pi == 2 ? 2 : 1), rb = pi == 2 ? 1 : (pi == 3 ? 3 : 2);
  const float beta = pi == 1 ? 1.f : -1.f;
  const int tty = li >> 3, ttx = li & 7;
  const unsigned ga = (unsigned)((half * QS + (2 * tty + ra) * RS + ttx) * 4);      // floats
  const unsigned gb = (unsigned)((half * QS + (2 * tty + rb) * RS + ttx) * 4);
  struct AF { pkf2 v[2][2]; };                             // [q][channel pair]: 4 channels = two register pairs
  // weight fragments from the slab: points p0 = 4 i + 3 jh (q = 0), p1 = 4 i + 1 + jh (q = 1); `bfa` = the lower of the two + this lane
  const unsigned bfa = (unsigned)((4 * pi + 2 * jh) * 256 + lane * 4);      // floats
  // MFMA pair e of a use: the wave's two points x input channels (e, e + 4 by lane half) -- independent accumulators, issued back to back
#define D8_MF0(f, b, ac) do { ac[0] = MFMA_32x32x2(b[0].x, f.v[0][0].x, ac[0]); ac[1] = MFMA_32x32x2(b[1].x, f.v[1][0].x, ac[1]); } while (0)
#define D8_MF1(f, b, ac) do { ac[0] = MFMA_32x32x2(b[0].y, f.v[0][0].y, ac[0]); ac[1] = MFMA_32x32x2(b[1].y, f.v[1][0].y, ac[1]); } while (0)
#define D8_MF2(f, b, ac) do { ac[0] = MFMA_32x32x2(b[0].z, f.v[0][1].x, ac[0]); ac[1] = MFMA_32x32x2(b[1].z, f.v[1][1].x, ac[1]); } while (0)
#define D8_MF3(f, b, ac) do { ac[0] = MFMA_32x32x2(b[0].w, f.v[0][1].y, ac[0]); ac[1] = MFMA_32x32x2(b[1].w, f.v[1][1].y, ac[1]); } while (0)
#ifdef MI355_EMU
#define D8_SGB(mask, n)
#else
#define D8_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif
  // End of a phase that issued N DMA instructions per wave: everything issued in EARLIER phases has landed (the counter retires in
  // order), this wave's LDS accesses are done, barrier. A raw s_barrier: __syncthreads() would drain the DMA queue (vmcnt(0)).
#if WINO_ABL & 32
#define D8_PHASE_END(N) do { COMPILER_FENCE(); WAIT_VMCNT_LGKM0((N) ? 63 : 0); RAW_BARRIER(); COMPILER_FENCE(); } while (0)      // timing only: never waits for a DMA inside the loop
#else
#define D8_PHASE_END(N) do { COMPILER_FENCE(); WAIT_VMCNT_LGKM0(N); RAW_BARRIER(); COMPILER_FENCE(); } while (0)
#endif
  // The main loop, once per j half (a scalar branch around it): JH is a compile-time constant inside.
  auto run = [&](auto jhc) {
    constexpr int JH = decltype(jhc)::value;
    auto b_lds = [&](float4 (&bu)[2], int dz) {
#if WINO_ABL & 4
      bu[0] = make_float4((float)dz, 0.5f, 0.25f, 1.f); bu[1] = make_float4(0.25f, (float)dz, 0.5f, 2.f);
#else
      const float* sp = ws + dz * WSF + bfa;
      bu[JH ? 1 : 0] = *reinterpret_cast<const float4*>(sp);
      bu[JH ? 0 : 1] = *reinterpret_cast<const float4*>(sp + 256);
#endif
    };
    // Fragments of the next plane, interleaved with the first two MFMA pairs of the phase (one scheduling region): the four window reads of
    // X0 / X2 go out first, their row sums and q0 (12 instructions) behind the first pair, then the two reads of X1 (at most 16
    // registers of window values in flight), its row sum and q1 (8) behind the second pair.
#define D8_GEN_WITH_PAIRS_01(fnext, xsb, fcur, b, ac) do {                                                                         \
      constexpr int c0_ = JH ? 3 : 0, c1_ = JH ? 2 : 1, c2_ = JH ? 1 : 2;                                                            \
      constexpr int o0 = ((c0_ >> 1) + 10 * (c0_ & 1)) * 4, o1 = ((c1_ >> 1) + 10 * (c1_ & 1)) * 4, o2 = ((c2_ >> 1) + 10 * (c2_ & 1)) * 4;  \
      const float* xb_ = (xsb);                                                                                                      \
      const float4 a0 = *reinterpret_cast<const float4*>(xb_ + ga + o0), b0 = *reinterpret_cast<const float4*>(xb_ + gb + o0);       \
      const float4 a2 = *reinterpret_cast<const float4*>(xb_ + ga + o2), b2 = *reinterpret_cast<const float4*>(xb_ + gb + o2);       \
      D8_MF0(fcur, b, ac);                                                                                                           \
      const float s00 = fmaf(b0.x, beta, a0.x), s01 = fmaf(b0.y, beta, a0.y), s02 = fmaf(b0.z, beta, a0.z), s03 = fmaf(b0.w, beta, a0.w);  \
      const float s20 = fmaf(b2.x, beta, a2.x), s21 = fmaf(b2.y, beta, a2.y), s22 = fmaf(b2.z, beta, a2.z), s23 = fmaf(b2.w, beta, a2.w);  \
      fnext.v[0][0] = make_pkf2(s00 - s20, s01 - s21); fnext.v[0][1] = make_pkf2(s02 - s22, s03 - s23);                              \
      const float4 a1 = *reinterpret_cast<const float4*>(xb_ + ga + o1), b1 = *reinterpret_cast<const float4*>(xb_ + gb + o1);       \
      D8_MF1(fcur, b, ac);                                                                                                           \
      const float s10 = fmaf(b1.x, beta, a1.x), s11 = fmaf(b1.y, beta, a1.y), s12 = fmaf(b1.z, beta, a1.z), s13 = fmaf(b1.w, beta, a1.w);  \
      fnext.v[1][0] = JH ? make_pkf2(s10 - s20, s11 - s21) : make_pkf2(s10 + s20, s11 + s21);                                        \
      fnext.v[1][1] = JH ? make_pkf2(s12 - s22, s13 - s23) : make_pkf2(s12 + s22, s13 + s23);                                        \
      /* the fragments are used by the NEXT phase only: without the pins hipcc sinks the arithmetic behind the barrier */            \
      PIN_IN_VGPR(fnext.v[0][0]); PIN_IN_VGPR(fnext.v[0][1]); PIN_IN_VGPR(fnext.v[1][0]); PIN_IN_VGPR(fnext.v[1][1]);                \
      D8_SGB(0x100, 4); D8_SGB(0x008, 2); D8_SGB(0x002, 12); D8_SGB(0x100, 2); D8_SGB(0x008, 2); D8_SGB(0x002, 8);                   \
      SCHED_BARRIER();                                                                                                               \
    } while (0)
    // One channel chunk = 4 phases (input planes pz = 0..3 of the tile). Phase pz:
    //   MFMAs with the fragments f(pz) generated in phase pz - 1 | fragments of plane pz + 1 | the DMA requests BETWEEN the MFMA pairs
    //   (issuing one occupies the wave for 60-190 cycles: in front of the MFMAs, as the first version had them, that was 16 % of the
    //   kernel -- profiles/r5_wino_d8.txt; behind a pair it overlaps with the 128 cycles the matrix pipe spends on that pair) |
    //   [plane pz + 2 activated in place] | the weight fragments of the next phase's first use read at the bottom (`bu`: in registers when
    //   the phase opens), those of a second use (`bv`) behind the first use's MFMAs.
    // Requests: input plane pz of the NEXT chunk (into the buffer whose plane was consumed in phase pz - 1); its W0 in phase 0, W1
    // in phase 2, W2 in phase 3 (the slab's fragments were last read one phase earlier) -- everything a phase reads from LDS was
    // requested at least two phases earlier and waited for at the end of the phase before.
    auto chunk = [&](int c0, float4 (&bu)[2], AF& fa, AF& fb) {
      const bool more = c0 + KC < a.CinP;                  // another chunk follows (workgroup-uniform)
      const int cn = more ? c0 + KC : c0;                  // after the last chunk the requests repeat it (never read)
      float4 bv[2];
      // phase 0: plane 0 x W0 -> output plane 0 | fragments of plane 1 | requests: next plane 0, next W0 | activate plane 2
      D8_GEN_WITH_PAIRS_01(fb, xs + XSF, fa, bu, acc[0]);
      dma_in(cn, 0);
      SCHED_BARRIER();
      D8_MF2(fa, bu, acc[0]);
      SCHED_BARRIER();
      dma_w(cn, 0);
      SCHED_BARRIER();
      D8_MF3(fa, bu, acc[0]);
      SCHED_BARRIER();
      activate(c0, 2);
      D8_PHASE_END(3);                                     // (bu = W0 again for the first use of phase 1)
      // phase 1: plane 1 x W0 -> output plane 1, x W1 -> output plane 0 | fragments of plane 2 | requests: next plane 1 | activate plane 3
      D8_GEN_WITH_PAIRS_01(fa, xs + 2 * XSF, fb, bu, acc[1]);
      b_lds(bv, 1);
      D8_MF2(fb, bu, acc[1]);
      D8_MF3(fb, bu, acc[1]);
      D8_SGB(0x100, 2); D8_SGB(0x008, 4);
      SCHED_BARRIER();
      dma_in(cn, 1);
      SCHED_BARRIER();
      D8_MF0(fb, bv, acc[0]); D8_MF1(fb, bv, acc[0]); D8_MF2(fb, bv, acc[0]); D8_MF3(fb, bv, acc[0]);
      SCHED_BARRIER();
      activate(c0, 3);
      bu[0] = bv[0]; bu[1] = bv[1];                        // W1: first use of phase 2
      D8_PHASE_END(1);
      // phase 2: plane 2 x W1 -> output plane 1, x W2 -> output plane 0 | fragments of plane 3 | requests: next plane 2, next W1 |
      // activate the next chunk's plane 0
      D8_GEN_WITH_PAIRS_01(fb, xs + 3 * XSF, fa, bu, acc[1]);
      b_lds(bv, 2);
      D8_MF2(fa, bu, acc[1]);
      D8_MF3(fa, bu, acc[1]);
      D8_SGB(0x100, 2); D8_SGB(0x008, 4);
      SCHED_BARRIER();
      dma_in(cn, 2);
      SCHED_BARRIER();
      D8_MF0(fa, bv, acc[0]); D8_MF1(fa, bv, acc[0]);
      SCHED_BARRIER();
      dma_w(cn, 1);
      SCHED_BARRIER();
      D8_MF2(fa, bv, acc[0]); D8_MF3(fa, bv, acc[0]);
      SCHED_BARRIER();
      activate(cn, 0);
      bu[0] = bv[0]; bu[1] = bv[1];                        // W2: first use of phase 3
      D8_PHASE_END(3);
      // phase 3: plane 3 x W2 -> output plane 1 | fragments of the next chunk's plane 0 | requests: its plane 3, its W2 | activate its
      // plane 1 | read its W0
      D8_GEN_WITH_PAIRS_01(fa, xs, fb, bu, acc[1]);
      dma_in(cn, 3);
      SCHED_BARRIER();
      D8_MF2(fb, bu, acc[1]);
      SCHED_BARRIER();
      dma_w(cn, 2);
      SCHED_BARRIER();
      D8_MF3(fb, bu, acc[1]);
      SCHED_BARRIER();
      activate(cn, 1);
      b_lds(bu, 0);
      D8_PHASE_END(3);
    };
    float4 bA[2];
    AF fA, fB;
    b_lds(bA, 0);
    {
      AF& f = fA;                                          // fragments of plane 0 (no MFMA to ride behind yet)
      constexpr int c0_ = JH ? 3 : 0, c1_ = JH ? 2 : 1, c2_ = JH ? 1 : 2;
      constexpr int o0 = ((c0_ >> 1) + 10 * (c0_ & 1)) * 4, o1 = ((c1_ >> 1) + 10 * (c1_ & 1)) * 4, o2 = ((c2_ >> 1) + 10 * (c2_ & 1)) * 4;
      const float4 a0 = *reinterpret_cast<const float4*>(xs + ga + o0), b0 = *reinterpret_cast<const float4*>(xs + gb + o0);
      const float4 a2 = *reinterpret_cast<const float4*>(xs + ga + o2), b2 = *reinterpret_cast<const float4*>(xs + gb + o2);
      const float4 a1 = *reinterpret_cast<const float4*>(xs + ga + o1), b1 = *reinterpret_cast<const float4*>(xs + gb + o1);
      const float s00 = fmaf(b0.x, beta, a0.x), s01 = fmaf(b0.y, beta, a0.y), s02 = fmaf(b0.z, beta, a0.z), s03 = fmaf(b0.w, beta, a0.w);
      const float s20 = fmaf(b2.x, beta, a2.x), s21 = fmaf(b2.y, beta, a2.y), s22 = fmaf(b2.z, beta, a2.z), s23 = fmaf(b2.w, beta, a2.w);
      const float s10 = fmaf(b1.x, beta, a1.x), s11 = fmaf(b1.y, beta, a1.y), s12 = fmaf(b1.z, beta, a1.z), s13 = fmaf(b1.w, beta, a1.w);
      f.v[0][0] = make_pkf2(s00 - s20, s01 - s21); f.v[0][1] = make_pkf2(s02 - s22, s03 - s23);
      f.v[1][0] = JH ? make_pkf2(s10 - s20, s11 - s21) : make_pkf2(s10 + s20, s11 + s21);
      f.v[1][1] = JH ? make_pkf2(s12 - s22, s13 - s23) : make_pkf2(s12 + s22, s13 + s23);
    }
    D8_PHASE_END(0);                                       // plane 0 and W0 have been read: phase 0 requests into their buffer / slab
    for (int c0 = 0; c0 < a.CinP; c0 += KC) chunk(c0, bA, fA, fB);
  };
#undef D8_GEN_WITH_PAIRS_01

  // prologue: the four planes and the three weight slabs of the first chunk requested together and waited for [; parameters to LDS,
  // planes 0 and 1 activated]
  dma_in(0, 0); dma_in(0, 1); dma_in(0, 2); dma_in(0, 3);
  dma_w(0, 0); dma_w(0, 1); dma_w(0, 2);
  if (INMODE == MI355_IN_AFFINE_ACT) {
    for (int c = tid; c < a.CinP; c += 512) {
      const bool in = c < a.Cin;
      prm[c] = in ? a.in_scale[(size_t)n * a.Cin + c] : 0.f;
      prm[a.CinP + c] = in ? a.in_shift[(size_t)n * a.Cin + c] : 0.f;
      prm[2 * a.CinP + c] = in ? (a.in_slope ? a.in_slope[c] : a.slope) : 0.f;
    }
    D8_PHASE_END(0);                                       // the parameters are in LDS, every request of this wave has landed
    activate(0, 0); activate(0, 1);                        // (planes 2 and 3 in phases 0 and 1, where the loop activates them in every chunk)
  }
  D8_PHASE_END(0);
  if (jh) run(std::integral_constant<int, 1>()); else run(std::integral_constant<int, 0>());
  D8_PHASE_END(0);                                         // the requests of the last phases (never read) have landed: the exchange reuses the LDS

  // ---- output transform Y = A^T M A, bias / residual / dropout scale, store ----
  // In-wave over the wave's two j (scalar branch on the j half), written to the exchange P[wave][b][tile][co] as 16-byte runs of the 4
  // consecutive channels an accumulator quad holds; across the waves over i on the way out, voxel-major.
  const int coq = tid & 7, ea = (wave >> 1) & 1;           // voxel rows: y = (tid >> 7) + 4 s -> a = y & 1 is wave-uniform
  const int co4 = co_base + 4 * coq;
  float bs[4], cs[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool cv = co4 + e < a.Cout;
    bs[e] = cv && a.bias ? a.bias[co4 + e] : 0.f;
    cs[e] = cv && a.out_chscale ? a.out_chscale[(size_t)n * a.Cout + co4 + e] : 1.f;
  }
  const bool q_in = co4 < a.Cout, q_full = co4 + 4 <= a.Cout;      // any / all four channels of the quad exist
  float K0[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  float gsc[4], gsh[4], gmean[4], grstd[4];
  int cnt = 0;
  if constexpr (FUSE == 2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int coc = co4 + e < a.Cout ? co4 + e : a.Cout - 1;
      const int grp = coc / (a.Cout / a.g.ggroups);
      gsc[e] = a.g.gscale[(size_t)n * a.Cout + coc]; gsh[e] = a.g.gshift[(size_t)n * a.Cout + coc];
      gmean[e] = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd[e] = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
    }
  }
  float* pw = P + wave * PW + li * PS + 4 * half;          // this lane's partials: + b * 32 * PS + 8 * g for accumulator quad g
  auto ld4 = [&](const float* base, size_t off, float (&v)[4]) {          // 4 channels of a voxel; scalar where 16-byte access is not legal
    if (a.vec4) {
      const float4 t = *reinterpret_cast<const float4*>(base + off);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = co4 + e < a.Cout ? base[off + e] : 0.f;
    }
  };
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz) {
    if (oz > 0) __syncthreads();                           // the previous plane's exchange has been read (the main loop ends on a barrier)
    if (jh) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16& m0 = acc[oz][0]; const f32x16& m1 = acc[oz][1];
        *reinterpret_cast<float4*>(pw + 8 * g) = make_float4(m1[4 * g], m1[4 * g + 1], m1[4 * g + 2], m1[4 * g + 3]);
        *reinterpret_cast<float4*>(pw + 32 * PS + 8 * g) =
            make_float4(m0[4 * g] - m1[4 * g], m0[4 * g + 1] - m1[4 * g + 1], m0[4 * g + 2] - m1[4 * g + 2], m0[4 * g + 3] - m1[4 * g + 3]);
      }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x16& m0 = acc[oz][0]; const f32x16& m1 = acc[oz][1];
        *reinterpret_cast<float4*>(pw + 8 * g) =
            make_float4(m0[4 * g] + m1[4 * g], m0[4 * g + 1] + m1[4 * g + 1], m0[4 * g + 2] + m1[4 * g + 2], m0[4 * g + 3] + m1[4 * g + 3]);
        *reinterpret_cast<float4*>(pw + 32 * PS + 8 * g) = make_float4(m1[4 * g], m1[4 * g + 1], m1[4 * g + 2], m1[4 * g + 3]);
      }
    }
    // reads that do not depend on the exchange go out before the barrier: the normalised tensor (FUSE 2) and the residual
    const int z = tz0 + oz, zc = z < a.D ? z : a.D - 1;
    float gxv[2][4], rsv[2][4];
    size_t vox[2];
    bool vin[2];
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      const int v = (tid >> 3) + 64 * sI;
      const int yy = ty0 + (v >> 4), xx = tx0 + (v & 15);
      vin[sI] = q_in && z < a.D && yy < a.H && xx < a.W;
      const int yc = yy < a.H ? yy : a.H - 1, xc = xx < a.W ? xx : a.W - 1;
      vox[sI] = (((size_t)n * a.D + zc) * a.H + yc) * a.W + xc;
      const int cq = q_in ? co4 : 0;                       // a quad beyond Cout reads (and drops) the first one
      if constexpr (FUSE == 2) ld4(a.g.gx, vox[sI] * a.g.gxld + cq, gxv[sI]);
      if (a.res) ld4(a.res, vox[sI] * a.resld + cq, rsv[sI]);
      else { rsv[sI][0] = rsv[sI][1] = rsv[sI][2] = rsv[sI][3] = 0.f; }
    }
    __syncthreads();
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      const int v = (tid >> 3) + 64 * sI;
      const int tile = ((v >> 5) << 3) + ((v & 15) >> 1), eb = v & 1;      // (y >> 1) * 8 + (x >> 1); b = x & 1
      const float* pz = P + (eb * 32 + tile) * PS + 4 * coq;      // wave w at + w * PW floats
      auto rd = [&](int w) { return *reinterpret_cast<const float4*>(pz + w * PW); };
      // across the waves: row i of the point grid = waves 2 i, 2 i + 1; A^T rows over i: (1, 1, 1, 0) and (0, 1, -1, -1)
      float4 o;
      if (ea == 0) {
        const float4 q0 = rd(0), q1 = rd(1), q2 = rd(2), q3 = rd(3), q4 = rd(4), q5 = rd(5);
        o.x = (q0.x + q1.x) + (q2.x + q3.x) + (q4.x + q5.x); o.y = (q0.y + q1.y) + (q2.y + q3.y) + (q4.y + q5.y);
        o.z = (q0.z + q1.z) + (q2.z + q3.z) + (q4.z + q5.z); o.w = (q0.w + q1.w) + (q2.w + q3.w) + (q4.w + q5.w);
      } else {
        const float4 q2 = rd(2), q3 = rd(3), q4 = rd(4), q5 = rd(5), q6 = rd(6), q7 = rd(7);
        o.x = (q2.x + q3.x) - (q4.x + q5.x) - (q6.x + q7.x); o.y = (q2.y + q3.y) - (q4.y + q5.y) - (q6.y + q7.y);
        o.z = (q2.z + q3.z) - (q4.z + q5.z) - (q6.z + q7.z); o.w = (q2.w + q3.w) - (q4.w + q5.w) - (q6.w + q7.w);
      }
      if (!vin[sI]) continue;
      float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (ov[e] + bs[e] + rsv[sI][e]) * cs[e];
      float* yp = a.y + vox[sI] * a.yld + co4;
      if (a.vec4 && q_full) *reinterpret_cast<float4*>(yp) = make_float4(ov[0], ov[1], ov[2], ov[3]);
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (co4 + e < a.Cout) yp[e] = ov[e];
      }
      if constexpr (FUSE == 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (cnt == 0) K0[e] = ov[e];
          const float t = ov[e] - K0[e];
          s0[e] += t; s1[e] += t * t;
        }
        ++cnt;
      } else if constexpr (FUSE == 2) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float xv = gxv[sI][e];
          const float u = xv * gsc[e] + gsh[e];
          const float du = u > 0.f ? ov[e] : ov[e] * a.g.gslope;
          s0[e] += du; s1[e] += du * ((xv - gmean[e]) * grstd[e]);
        }
      }
    }
  }
  if constexpr (FUSE != 0) wino_fuse_records<FUSE>(a, P, tid, lane, wave, coq, co_base, n, tz0, ty0, tx0, cnt, K0, s0, s1);
}
#undef D8_SGB
#undef D8_PHASE_END
#undef D8_MF0
#undef D8_MF1
#undef D8_MF2
#undef D8_MF3

// ---- filter transform: U[(i,j)][dz][ci][co] = sum_{dy,dx} G[i][dy] G[j][dx] w[...] (pack_values.h: pack_wino_item) ----
__global__ void wino_pack_weight_kernel(const float* w, float* up, int cout, int cin, int coutP, int cinP, int mode) {
  const size_t items = (size_t)3 * (cinP / 4) * coutP * 4;          // one per (dz, ci, co): 16 points each
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < items; r += (size_t)gridDim.x * blockDim.x)
    pack_wino_item(w, up, r, cout, cin, coutP, cinP, mode);
}

extern "C" size_t mi355_wino_weight_elems(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t coutP = (cout + 31) / 32 * 32, cinP = (cin + 7) / 8 * 8;
  return (size_t)48 * cinP * coutP;
}

extern "C" int mi355_wino_pack_weight(const float* w, float* up, int32_t cout, int32_t cin, int32_t mode, void* stream) {
  if (!w || !up || cout <= 0 || cin <= 0 || mode < 0 || mode > 1) return MI355_EINVAL;
  const int coutP = (cout + 31) / 32 * 32, cinP = (cin + 7) / 8 * 8;     // logical packed dims: cout = "out", cin = "in" of THIS conv
  const size_t items = (size_t)3 * cinP * coutP;
  int grid = (int)((items + 255) / 256); if (grid > 4096) grid = 4096;
  LAUNCH(wino_pack_weight_kernel, dim3(grid), dim3(256), 0, stream, w, up, cout, cin, coutP, cinP, mode);
  return LAUNCH_CHECK();
}

extern "C" int mi355_conv3d_wino_supported(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  return wino_check(x, y, d) == MI355_OK;
}

extern "C" int mi355_conv3d_wino_fwd(const mi355_act* x, const float* up, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!up || ((uintptr_t)up & 15)) return MI355_EINVAL;
  { const int rc = wino_check(x, y, d); if (rc) return rc; }
  WinoArgs a;
  memset(&a.g, 0, sizeof(a.g));
  a.g.mom = d->moments_out;
  if (d->gn_bwd) {
    const mi355_gn_bwd_fuse* f = d->gn_bwd;
    if (!f->gx || !f->scale || !f->shift || !f->mean_rstd || !f->partials_out || f->groups <= 0 || y->c % f->groups || f->gx_ld < y->c) return MI355_EINVAL;
    a.g.gnb = f->partials_out; a.g.gx = (const float*)f->gx; a.g.gxld = f->gx_ld; a.g.gscale = f->scale; a.g.gshift = f->shift; a.g.gmr = f->mean_rstd;
    a.g.ggroups = f->groups; a.g.gslope = f->act_slope;
  }
  a.x = (const float*)x->p; a.xld = x->ld; a.up = up; a.y = (float*)y->p; a.yld = y->ld;
  a.res = (const float*)d->residual; a.resld = d->residual_ld;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.out_chscale = d->out_chscale; a.bias = d->bias;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.CinP = (x->c + 7) / 8 * 8;
  a.Cout = y->c; a.CoutP = (y->c + 31) / 32 * 32;
  a.tilesZ = ceil_div(a.D, 2); a.tilesY = ceil_div(a.H, 8); a.tilesX = ceil_div(a.W, 16); a.coTiles = a.CoutP / 32;
  a.vec4 = a.Cout % 4 == 0 && a.yld % 4 == 0 && !((uintptr_t)a.y & 15) && (!a.res || (a.resld % 4 == 0 && !((uintptr_t)a.res & 15))) &&
           (!a.g.gnb || (a.g.gxld % 4 == 0 && !((uintptr_t)a.g.gx & 15)));
  const long long blocks = (long long)a.N * a.tilesZ * a.tilesY * a.tilesX * a.coTiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
  const dim3 grid((unsigned)blocks), blk(512);
  // 4 staged plane chunks + 3 weight slabs (75 264 bytes; the padded exchange of the epilogue, 73 728, lives inside) + norm prologue: two
  // workgroups per CU
  const int lds_bytes = (4 * 1632 + 3 * 4096 + 3 * a.CinP) * (int)sizeof(float);
#define WINO_LAUNCH(IM, FU)                                                                          \
  do { SET_MAX_DYN_LDS((conv3d_wino2d_d8<IM, FU>), lds_bytes);                                         \
       LAUNCH((conv3d_wino2d_d8<IM, FU>), grid, blk, lds_bytes, stream, a); } while (0)
  if (a.g.mom) {
    if (d->in_mode == MI355_IN_PLAIN) WINO_LAUNCH(MI355_IN_PLAIN, 1); else WINO_LAUNCH(MI355_IN_AFFINE_ACT, 1);
  } else if (a.g.gnb) {
    WINO_LAUNCH(MI355_IN_PLAIN, 2);
  } else if (d->in_mode == MI355_IN_PLAIN) WINO_LAUNCH(MI355_IN_PLAIN, 0);
  else WINO_LAUNCH(MI355_IN_AFFINE_ACT, 0);
#undef WINO_LAUNCH
  return LAUNCH_CHECK();
}

// epilogue records per sample: one per 2 x 8 x 16 tile
extern "C" int32_t mi355_conv3d_wino_stats_blocks(const mi355_act* y) {
  if (!y) return 0;
  const long long b = (long long)ceil_div(y->d, 2) * ceil_div(y->h, 8) * ceil_div(y->w, 16);
  return b > 0 && b <= 0x7fffffffLL ? (int32_t)b : 0;
}
