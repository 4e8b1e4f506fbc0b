// 3x3x3 stride-1 conv3d forward / dgrad in exact-type fp32 with Winograd F(2x2x2, 3x3x3): the transform of conv3d_wino.hip applied along z
// as well (round 6). 64 transform points per 2x2x2 outputs = 8 multiplications per output voxel and (ci, co) instead of 12 (F(2x2, 3x3) x
// direct z) or 27 (direct): the matrix-pipe floor of the eligible layers drops by a third against conv3d_wino2d_d8.
//
// Same op as conv3d_fwd.hip / conv3d_wino.hip (reference: unet3d/models/pytorch/classification/resnet.py:12-22 called from
// myronenko.py:17-21; GroupNorm-apply + ReLU prologue, bias / residual / Dropout3d-scale epilogue and the fused norm statistics as
// there). Numerics: every transform matrix has entries 0, +-1, +-1/2; tools/winograd_probe.py (CPU, fp32 against fp64) puts the 3-D form
// in the error class of the direct fp32 convolution (2.8e-7 .. 7.3e-7 of max |y| against 4.2e-7 .. 4.6e-7), tests/test_wino3d_*.py and
// the per-launch audit hold the kernel to the same bounds as the 2-D form.
//
//   input transform   V = B^T d B (x) B^T along z   (4x4x4 window d of one channel; B^T rows: d0-d2, d1+d2, d2-d1, d1-d3)
//   filter transform  U = G g G^T (x) G along z     (once per optimizer step by the pack kernel below)
//   point-wise        M[p] = sum_ci V[p][ci] * U[p][ci][co]                              <- the MFMA work, p = (zi, i, j) = 0..63
//   output transform  Y = A^T M A (x) A^T along z   (2x2x2 outputs; A^T rows: m0+m1+m2, m1-m2-m3)
//
// Workgroup = 1024 threads = 16 waves on a 2 (z) x 8 x 16 voxel x 32 output-channel tile: the 32 plane tiles of 2x2 are one dimension of
// a 32x32 MFMA tile, the 32 output channels the other; wave w = (z point zi = w >> 2, row point i = w & 3) owns the four column points
// j = 0..3 (4 accumulator tiles = 64 registers; 128 registers per wave, four waves per SIMD, ONE workgroup per CU: 64 accumulator tiles
// are half the register file of a CU whichever way they are dealt). K = 4 input channels per phase (one 16-byte channel quad of every
// staged voxel; a lane half supplies two of them): 8 MFMAs per wave and phase, one barrier per phase.
//  * Every global load of the main loop is an LDS-DMA (global_load_lds_dwordx4), as in conv3d_wino2d_d8, with counted waits and raw
//    barriers: per phase a wave issues ONE input request (16 waves x 51 lanes = the 816 slots of the four haloed 10 x 18 planes of a
//    channel quad) and TWO weight requests (the 4 points x 4 input x 32 output channels the wave itself consumes: the weights never
//    cross waves). Input: ring of 4 staged chunks, requested four phases before its MFMAs (landed after two, activated in place by the
//    requesting lane in the third, fragments generated in the fourth); weights: ring of 3 slabs of 32 KB, requested three phases ahead.
//  * A fragments generated in registers from the RAW staged planes: a wave's points share the two planes (za, zb) and two window rows
//    (ra, rb) its (zi, i) combine; lane half h reads window column 2P + h of column pair P with four conflict-free ds_read_b128 (all four
//    channels of the quad), forms R = (d[za][ra] + b d[za][rb]) + bz (d[zb][ra] + b d[zb][rb]) on them, and two v_permlane32_swap hand each
//    half the two channels it feeds the MFMAs with of BOTH columns (a 2 x 2 transpose of the lane pair, no LDS): 8 reads + 36 vector
//    instructions per phase; then V[j] = R0 - R2, R1 + R2, R2 - R1, R1 - R3.
//  * MFMA operands as in the 2-D kernel (A = weights, B = input): a lane holds 4 consecutive output channels of a tile per accumulator quad.
//    Weight fragments: the pack stores, per point and channel quad, [lane half][output channel][2 channels], so that a fragment is one
//    conflict-free ds_read_b64.
//  * Epilogue: the column part of the output transform in registers (4 j -> 2 b), ONE exchange of all 16 waves' partials through LDS
//    (144 KB: the rings are dead by then), the row and z parts (9 signed terms) on the way out, voxel-major, 16-byte stores; residual,
//    Dropout3d scale, bias and the fused statistics (one record per 2 x 8 x 16 tile, the geometry of the 2-D kernel: same folding code).
#include "gfx950_dialect.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"
#include "pack_values.h"
#include "wino_common.h"

#ifndef WINO3_ZBRICK
#define WINO3_ZBRICK 8            // z tiles per brick of the workgroup order
#endif
#ifndef WINO3_ABL
// developer ablations (tools/build_variant.sh ... -DWINO3_ABL=mask; results wrong by construction, timing only):
// 1 no input requests, 2 no weight requests, 4 no fragment generation (the loop reuses the first fragments), 8 no MFMAs, 16 no epilogue
// (one store per lane keeps the accumulators alive), 32 no barrier at the end of a phase, 64 no weight-fragment reads in the loop
#define WINO3_ABL 0
#endif

__device__ const float wino3_zero16[4] = {0.f, 0.f, 0.f, 0.f};

// glds16 (gfx950_dialect.h) for lanes 0..50 of a wave only, without a branch: `if (lane < 51) glds16(...)` becomes s_and_saveexec +
// s_cbranch_execz, which cuts the phase into basic blocks the scheduler cannot interleave across; here the lane mask is put into EXEC
// around the one instruction inside the asm statement.
#ifdef MI355_EMU
static inline void glds16_lanes51(const void* src, float* lds_wave_base) { if (emu::flat_tid() % 64 < 51) glds16(src, lds_wave_base); }
#define W3_SGB(mask, n)
#else
__device__ __forceinline__ void glds16_lanes51(const void* src, float* lds_wave_base) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_byte_address(lds_wave_base));
  unsigned long long saved;
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %3\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b64 exec, %0"
               : "=&s"(saved) : "v"(src), "s"(m0v), "s"(0x0007ffffffffffffull) : "memory");
}
#define W3_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#endif

template <int INMODE, int FUSE>
__global__ __launch_bounds__(1024) void conv3d_wino3d(WinoArgs a) {
  constexpr int TZ = 2, TY = 8, TX = 16, HY = TY + 2;
  constexpr int RS = 20;                                   // 16-byte slots per staged row: even columns 0..8 | pad | odd columns 10..18 | pad
  constexpr int PSL = HY * RS + 4;                         // slots per staged plane (204 = 4 waves x 51 lanes)
  constexpr int XSF = 4 * PSL * 4;                         // floats of one staged chunk: 4 planes of one channel quad (816 slots)
  constexpr int WSF = 64 * 128;                            // floats of one weight slab: 64 points x [half 2][co 32][2]
  constexpr int NXB = 4, NWB = 3;                          // ring depths
  constexpr int PS = 36, PW = 2 * 32 * PS;                 // exchange: [wave][b][tile][32 channels + 4 pad]
  static_assert(16 * 51 == 4 * PSL, "one DMA instruction per wave fills a staged chunk");
  DYN_LDS(lds);
  float* xs = lds;                                         // ring of NXB staged chunks
  float* ws = lds + NXB * XSF;                             // ring of NWB weight slabs
  float* P = lds;                                          // epilogue: output-transform exchange (reuses everything)
  float* prm = lds + NXB * XSF + NWB * WSF;                // norm prologue of this sample: scale | shift | slope, Cin each
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  // workgroup -> (channel tile, spatial tile), as conv3d_wino2d_d8: an XCD gets a contiguous range of spatial tiles of one channel tile,
  // walked in bricks of WINO3_ZBRICK z tiles, then x, y
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
  if (a.tilesZ % WINO3_ZBRICK == 0) {
    const int zi_ = b % WINO3_ZBRICK; b /= WINO3_ZBRICK;
    const int txi = b % a.tilesX; b /= a.tilesX;
    const int tyi = b % a.tilesY; b /= a.tilesY;
    const int zbk = b % (a.tilesZ / WINO3_ZBRICK); b /= (a.tilesZ / WINO3_ZBRICK);
    tz = zbk * WINO3_ZBRICK + zi_;
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
  const int NQ = a.Cin / 4;                                // channel quads = phases of the main loop (Cin % 4 == 0: wino_check)

  f32x16 acc[4];                                           // [column point j]
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // ---- staging by LDS-DMA ----
  // Input: slot s = 51 wave + lane (lane < 51) of a staged chunk [plane 4][row 10][position 20 (+ 4 pad slots per plane)]: waves 4 pz ..
  // 4 pz + 3 fill plane pz. EVERY slot is written by every request: pads, voxels outside the image and every slot of a plane outside the
  // volume fetch 16 zero bytes -- a wave issues the same number of DMA instructions in every phase, which the counted waits rely on.
  const int spz = wave >> 2;                               // the plane this wave's requests fill (uniform)
  const int sr = (wave & 3) * 51 + (lane < 50 ? lane : 50);
  const int srow = sr / RS, spos = sr % RS;
  const int scol = spos < 9 ? 2 * spos : 2 * (spos - 10) + 1;
  const bool sunit = lane < 51 && srow < HY && spos != 9 && spos != 19;
  const int siy = ty0 - 1 + srow, six = tx0 - 1 + scol;
  const int siz = tz0 - 1 + spz;
  const bool sin = sunit && siy >= 0 && siy < a.H && six >= 0 && six < a.W && siz >= 0 && siz < a.D;
  const LaneMask m_in = LANE_MASK(sin);
  const size_t xplane = (size_t)a.H * a.W * a.xld;
  const float* xn = a.x + ((size_t)n * a.D + (siz >= 0 && siz < a.D ? siz : 0)) * xplane;      // sample n, this wave's plane
  const unsigned xoff = sin ? (unsigned)((siy * a.W + six) * a.xld) * 4u : 0u;               // bytes inside the plane
  // channel quad q_ -> ring buffer bx. The ring runs NXB quads ahead: the requests past the last quad of the tile keep the instruction
  // count of a phase constant and fetch the 16 zero bytes in every lane (one cache line: such requests cost nothing, profiles/r5_wino_d8.txt)
  auto dma_in = [&](int q_, int bx) {
    const char* real = reinterpret_cast<const char*>(xn + 4 * q_) + xoff;
    const LaneMask m = q_ < NQ ? m_in : (LaneMask)0;       // uniform
    const char* src = LANE_IN_MASK(m) ? real : reinterpret_cast<const char*>(wino3_zero16);
#if !(WINO3_ABL & 1)
    glds16_lanes51(src, xs + bx * XSF + wave * (51 * 4));
#else
    (void)src;
#endif
  };
  // Norm prologue (INMODE = MI355_IN_AFFINE_ACT): the lane that requested a slot rewrites it in place once its own counted wait says it
  // has landed (no barrier needed for its own slot), one phase before the fragment generation reads the chunk; slots that fetched zeros
  // are left alone (the padding of the ACTIVATED tensor is zero).
  auto activate = [&](int q_, int bx) {
    if (INMODE != MI355_IN_AFFINE_ACT) return;
    if (LANE_IN_MASK(m_in)) {                              // (no bit for lanes 51..63)
      const int c = 4 * q_;
      float* p = xs + bx * XSF + (wave * 51 + lane) * 4;
      float4 v = *reinterpret_cast<const float4*>(p);
      const float4 sc = *reinterpret_cast<const float4*>(prm + c), sh = *reinterpret_cast<const float4*>(prm + a.Cin + c);
      const float4 sl = *reinterpret_cast<const float4*>(prm + 2 * a.Cin + c);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
      *reinterpret_cast<float4*>(p) = v;
    }
  };
  // Weights of channel quad q_ -> slab bw: the wave's own four points p = 4 wave + j, two per request (lane half = point of the pair);
  // pack layout [point][quad][channel tile][half 2][co 32][2] floats = 512 bytes per (point, quad, channel tile)
  const float4* up4 = reinterpret_cast<const float4*>(a.up);
  const size_t pstep = (size_t)NQ * a.coTiles * 32;        // float4s between consecutive points of the pack
  const unsigned woff = (unsigned)(half * pstep + li) * 16u;      // bytes: (point of the pair, 16-byte unit of the point's 512 bytes)
  auto dma_w = [&](int q_, int bw) {
    const bool live = q_ < NQ;                             // uniform; past the last quad: zeros, as the input
#pragma unroll
    for (int j2 = 0; j2 < 2; ++j2) {
      const float4* src = up4 + ((size_t)(4 * wave + 2 * j2) * NQ + q_) * (a.coTiles * 32) + (size_t)cot * 32;      // uniform
#if !(WINO3_ABL & 2)
      glds16_uniform_base(live ? reinterpret_cast<const void*>(src) : reinterpret_cast<const void*>(wino3_zero16), woff & (live ? ~0u : 0u),
                          ws + bw * WSF + (4 * wave + 2 * j2) * 128);
#else
      (void)src; (void)live;
#endif
    }
  };

  // ---- fragment generation: wave constants (planes za zb, rows ra rb and their signs), the lane's four window addresses ----
  const int zi = wave >> 2, pi = wave & 3;
  const int za = zi == 0 ? 0 : (zi == 2 ? 2 : 1), zb = zi == 2 ? 1 : (zi == 3 ? 3 : 2);
  const int ra = pi == 0 ? 0 : (pi == 2 ? 2 : 1), rb = pi == 2 ? 1 : (pi == 3 ? 3 : 2);
  const float betaz = zi == 1 ? 1.f : -1.f, beta = pi == 1 ? 1.f : -1.f;
  const int tty = li >> 3, ttx = li & 7;
  // window column 2 P + h of column pair P: even columns (h = 0) at position ttx + P, odd ones (h = 1) at 10 + ttx + P
  const unsigned g_aa = (unsigned)((za * PSL + (2 * tty + ra) * RS + ttx + 10 * half) * 4);      // floats
  const unsigned g_ab = (unsigned)((za * PSL + (2 * tty + rb) * RS + ttx + 10 * half) * 4);
  const unsigned g_ba = (unsigned)((zb * PSL + (2 * tty + ra) * RS + ttx + 10 * half) * 4);
  const unsigned g_bb = (unsigned)((zb * PSL + (2 * tty + rb) * RS + ttx + 10 * half) * 4);
  struct AF { float v[4][2]; };                            // [j][channel of the lane's pair]
  struct Win { float4 aa, ab, ba, bb; };                   // the four window values (plane, row) of one column, four channels each
  auto win_read = [&](const float* xb, int Pc, Win& t) {
    t.aa = *reinterpret_cast<const float4*>(xb + g_aa + 4 * Pc); t.ab = *reinterpret_cast<const float4*>(xb + g_ab + 4 * Pc);
    t.ba = *reinterpret_cast<const float4*>(xb + g_ba + 4 * Pc); t.bb = *reinterpret_cast<const float4*>(xb + g_bb + 4 * Pc);
  };
  // R = (d[za][ra] + b d[za][rb]) + bz (d[zb][ra] + b d[zb][rb]) of this lane's column (12 fma), then the 2 x 2 transpose of the lane pair:
  // lane (t, 0) holds column 2 Pc, lane (t, 1) column 2 Pc + 1, four channels each -> both hold both columns, channels 2 h, 2 h + 1
  auto win_combine = [&](const Win& t, float (&Ra)[2], float (&Rb)[2]) {
    float r0 = fmaf(t.ab.x, beta, t.aa.x), r1 = fmaf(t.ab.y, beta, t.aa.y), r2 = fmaf(t.ab.z, beta, t.aa.z), r3 = fmaf(t.ab.w, beta, t.aa.w);
    const float t0 = fmaf(t.bb.x, beta, t.ba.x), t1 = fmaf(t.bb.y, beta, t.ba.y), t2 = fmaf(t.bb.z, beta, t.ba.z), t3 = fmaf(t.bb.w, beta, t.ba.w);
    r0 = fmaf(t0, betaz, r0); r1 = fmaf(t1, betaz, r1); r2 = fmaf(t2, betaz, r2); r3 = fmaf(t3, betaz, r3);
    permlane32_swap(r0, r2);
    permlane32_swap(r1, r3);
    Ra[0] = r0; Ra[1] = r1; Rb[0] = r2; Rb[1] = r3;
  };
  auto gen_finish = [&](const float (&R)[4][2], AF& f) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      f.v[0][e] = R[0][e] - R[2][e];
      f.v[1][e] = R[1][e] + R[2][e];
      f.v[2][e] = R[2][e] - R[1][e];
      f.v[3][e] = R[1][e] - R[3][e];
    }
  };
  auto gen = [&](const float* xb, AF& f) {                 // (prologue: nothing to interleave with)
    float R[4][2];
    Win t;
    win_read(xb, 0, t); win_combine(t, R[0], R[1]);
    win_read(xb, 1, t); win_combine(t, R[2], R[3]);
    gen_finish(R, f);
  };
  // weight fragments of the wave's four points from slab bw: one ds_read_b64 each (lanes of a half read consecutive 8 bytes)
  struct WF { float2 v[4]; };
  const unsigned wfa = (unsigned)(4 * wave * 128 + half * 64 + li * 2);      // floats
  auto w_lds = [&](int bw, WF& w) {
    const float* sp = ws + bw * WSF + wfa;
#pragma unroll
    for (int j = 0; j < 4; ++j) w.v[j] = *reinterpret_cast<const float2*>(sp + j * 128);
  };
#if WINO3_ABL & 8
#define W3_MF(j, e) do { } while (0)
#else
#define W3_MF(j, e) acc[j] = MFMA_32x32x2((e) ? wcur.v[j].y : wcur.v[j].x, fcur.v[j][e], acc[j])
#endif
  // End of a phase that issued N DMA instructions per wave: everything issued in EARLIER phases has landed (in-order counter), this wave's
  // LDS accesses are done, barrier (raw: __syncthreads() would drain the DMA queue); the compiler may not move LDS accesses across it
#if WINO3_ABL & 32
#define W3_PHASE_END(N) do { COMPILER_FENCE(); WAIT_VMCNT_LGKM0(N); if ((N) == 0) RAW_BARRIER(); COMPILER_FENCE(); } while (0)
#else
#define W3_PHASE_END(N) do { COMPILER_FENCE(); WAIT_VMCNT_LGKM0(N); RAW_BARRIER(); COMPILER_FENCE(); } while (0)
#endif

  // ---- prologue: the first four input chunks and three weight slabs requested together and waited for ----
  dma_in(0, 0); dma_in(1, 1); dma_in(2, 2); dma_in(3, 3);
  dma_w(0, 0); dma_w(1, 1); dma_w(2, 2);
  if (INMODE == MI355_IN_AFFINE_ACT) {
    for (int c = tid; c < a.Cin; c += 1024) {
      prm[c] = a.in_scale[(size_t)n * a.Cin + c];
      prm[a.Cin + c] = a.in_shift[(size_t)n * a.Cin + c];
      prm[2 * a.Cin + c] = a.in_slope ? a.in_slope[c] : a.slope;
    }
    W3_PHASE_END(0);                                       // the parameters are in LDS, every request of this wave has landed
    activate(0, 0); if (NQ > 1) activate(1, 1);                // (chunks 2, 3 in phases 0, 1, where the loop activates chunk p + 2)
  }
  W3_PHASE_END(0);
  AF fA, fB;
  WF wA, wB;
  gen(xs, fA);
  w_lds(0, wA);
  W3_PHASE_END(0);                                         // chunk 0 and slab 0 have been read: phase 0 requests into them

  // ---- main loop. Phase p: MFMAs of quad p (fragments + weights in registers) | fragments of quad p + 1 from ring[(p + 1) % 4] |
  // quad p + 2 activated in place | requests: input quad p + 4 -> ring[p % 4] (read in phase p - 1), weights of quad p + 3 -> slab[p % 3]
  // (read at the bottom of phase p - 1) | weight fragments of quad p + 1 read at the bottom. Everything a phase reads from LDS was
  // requested at least two phases earlier and waited for at the end of the phase before. Two phases per trip: the fragment sets swap.
  int bx = 0, bw = 0;                                      // p % 4, p % 3
  // The order inside a phase (scheduling regions, pinned): the window reads of column pair 0 go out first, two MFMAs behind them cover
  // the LDS latency; every DMA request sits behind an MFMA (issuing one occupies the wave for 60-190 cycles, the matrix pipe spends
  // 64 on an MFMA and has three other waves of the SIMD to take from); the 36 fragment instructions ride between the remaining MFMAs.
  auto phase = [&](int p, const AF& fcur, const WF& wcur, AF& fnext, WF& wnext) {
    const int bx1 = (bx + 1) & 3, bx2 = (bx + 2) & 3, bw1 = bw == 2 ? 0 : bw + 1;
    const float* xb = xs + bx1 * XSF;
    float R[4][2];
    Win t0, t1;
#if !(WINO3_ABL & 4)
    win_read(xb, 0, t0);
#endif
    SCHED_BARRIER();
    W3_MF(0, 0); W3_MF(1, 0);
    SCHED_BARRIER();
    dma_in(p + 4, bx);
    SCHED_BARRIER();
#if !(WINO3_ABL & 4)
    win_read(xb, 1, t1);
    win_combine(t0, R[0], R[1]);
#endif
    W3_MF(2, 0); W3_MF(3, 0);
    W3_SGB(0x100, 4); W3_SGB(0x008, 1); W3_SGB(0x002, 7); W3_SGB(0x008, 1); W3_SGB(0x002, 7);
    SCHED_BARRIER();
    dma_w(p + 3, bw);
    SCHED_BARRIER();
#if !(WINO3_ABL & 4)
    win_combine(t1, R[2], R[3]);
#endif
    W3_MF(0, 1); W3_MF(1, 1);
    W3_SGB(0x008, 1); W3_SGB(0x002, 7); W3_SGB(0x008, 1); W3_SGB(0x002, 7);
    SCHED_BARRIER();
#if !(WINO3_ABL & 4)
    gen_finish(R, fnext);
#endif
#if !(WINO3_ABL & 64)
    w_lds(bw1, wnext);
#endif
    W3_MF(2, 1); W3_MF(3, 1);
    W3_SGB(0x008, 1); W3_SGB(0x100, 4); W3_SGB(0x002, 4); W3_SGB(0x008, 1); W3_SGB(0x002, 4);
    SCHED_BARRIER();
    if (p + 2 < NQ) activate(p + 2, bx2);                  // (uniform)
    // the fragments are used by the NEXT phase only: without the pins hipcc sinks the arithmetic behind the barrier
#pragma unroll
    for (int j = 0; j < 4; ++j) { PIN_IN_VGPR(fnext.v[j][0]); PIN_IN_VGPR(fnext.v[j][1]); }
    SCHED_BARRIER();
    W3_PHASE_END(3);
    bx = bx1; bw = bw1;
  };
  for (int p = 0; p < NQ; p += 2) {
    phase(p, fA, wA, fB, wB);
    if (p + 1 < NQ) phase(p + 1, fB, wB, fA, wA);
  }
  W3_PHASE_END(0);                                         // the requests of the last phases (never read) have landed: the exchange reuses the LDS

#if WINO3_ABL & 16
  { float t = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[j][r];
    a.y[(size_t)blockIdx.x * 1024 % 4096 + tid] = t; return; }
#endif
  // ---- output transform Y = A^T M A (x) A^T, bias / residual / dropout scale, store ----
  // Columns (j -> b) in registers, written to the exchange P[wave][b][tile][co] as 16-byte runs of the 4 consecutive channels an accumulator
  // quad holds; rows (i -> a) and planes (zi -> oz) across the waves on the way out, voxel-major.
  {
    float* pw = P + wave * PW + li * PS + 4 * half;        // + b * 32 * PS + 8 * g for accumulator quad g
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float4 b0, b1;
      b0.x = (acc[0][4 * g] + acc[1][4 * g]) + acc[2][4 * g];             b1.x = (acc[1][4 * g] - acc[2][4 * g]) - acc[3][4 * g];
      b0.y = (acc[0][4 * g + 1] + acc[1][4 * g + 1]) + acc[2][4 * g + 1]; b1.y = (acc[1][4 * g + 1] - acc[2][4 * g + 1]) - acc[3][4 * g + 1];
      b0.z = (acc[0][4 * g + 2] + acc[1][4 * g + 2]) + acc[2][4 * g + 2]; b1.z = (acc[1][4 * g + 2] - acc[2][4 * g + 2]) - acc[3][4 * g + 2];
      b0.w = (acc[0][4 * g + 3] + acc[1][4 * g + 3]) + acc[2][4 * g + 3]; b1.w = (acc[1][4 * g + 3] - acc[2][4 * g + 3]) - acc[3][4 * g + 3];
      *reinterpret_cast<float4*>(pw + 8 * g) = b0;
      *reinterpret_cast<float4*>(pw + 32 * PS + 8 * g) = b1;
    }
  }
  const int coq = tid & 7, ea = (wave >> 1) & 1;           // voxel v = tid >> 3 of a plane: y = v >> 4 = wave >> 1 -> a = y & 1 is wave-uniform
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
  auto ld4 = [&](const float* base, size_t off, float (&v)[4]) {          // 4 channels of a voxel; scalar where 16-byte access is not legal
    if (a.vec4) {
      const float4 t = *reinterpret_cast<const float4*>(base + off);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = co4 + e < a.Cout ? base[off + e] : 0.f;
    }
  };
  // reads that do not depend on the exchange go out before the barrier: the normalised tensor (FUSE 2) and the residual
  const int v = tid >> 3;
  const int yy = ty0 + (v >> 4), xx = tx0 + (v & 15);
  const int yc = yy < a.H ? yy : a.H - 1, xc = xx < a.W ? xx : a.W - 1;
  float gxv[2][4], rsv[2][4];
  size_t vox[2];
  bool vin[2];
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz) {
    const int z = tz0 + oz, zc = z < a.D ? z : a.D - 1;
    vin[oz] = q_in && z < a.D && yy < a.H && xx < a.W;
    vox[oz] = (((size_t)n * a.D + zc) * a.H + yc) * a.W + xc;
    const int cq = q_in ? co4 : 0;                         // a quad beyond Cout reads (and drops) the first one
    if constexpr (FUSE == 2) ld4(a.g.gx, vox[oz] * a.g.gxld + cq, gxv[oz]);
    if (a.res) ld4(a.res, vox[oz] * a.resld + cq, rsv[oz]);
    else { rsv[oz][0] = rsv[oz][1] = rsv[oz][2] = rsv[oz][3] = 0.f; }
  }
  __syncthreads();
  const int tile = ((v >> 5) << 3) + ((v & 15) >> 1), eb = v & 1;      // (y >> 1) * 8 + (x >> 1); b = x & 1
  const float* pz = P + (eb * 32 + tile) * PS + 4 * coq;  // wave (zi, i) at + (4 zi + i) * PW floats
  // rows over i for this thread's a: A^T rows (1, 1, 1, 0) and (0, 1, -1, -1)
  auto rowsum = [&](int zi_) {
    const float* q = pz + 4 * zi_ * PW;
    float4 o;
    if (ea == 0) {
      const float4 q0 = *reinterpret_cast<const float4*>(q), q1 = *reinterpret_cast<const float4*>(q + PW), q2 = *reinterpret_cast<const float4*>(q + 2 * PW);
      o.x = (q0.x + q1.x) + q2.x; o.y = (q0.y + q1.y) + q2.y; o.z = (q0.z + q1.z) + q2.z; o.w = (q0.w + q1.w) + q2.w;
    } else {
      const float4 q1 = *reinterpret_cast<const float4*>(q + PW), q2 = *reinterpret_cast<const float4*>(q + 2 * PW), q3 = *reinterpret_cast<const float4*>(q + 3 * PW);
      o.x = (q1.x - q2.x) - q3.x; o.y = (q1.y - q2.y) - q3.y; o.z = (q1.z - q2.z) - q3.z; o.w = (q1.w - q2.w) - q3.w;
    }
    return o;
  };
  const float4 z0 = rowsum(0), z1 = rowsum(1), z2 = rowsum(2), z3 = rowsum(3);
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz) {
    if (!vin[oz]) continue;
    float ov[4];
    if (oz == 0) { ov[0] = (z0.x + z1.x) + z2.x; ov[1] = (z0.y + z1.y) + z2.y; ov[2] = (z0.z + z1.z) + z2.z; ov[3] = (z0.w + z1.w) + z2.w; }
    else { ov[0] = (z1.x - z2.x) - z3.x; ov[1] = (z1.y - z2.y) - z3.y; ov[2] = (z1.z - z2.z) - z3.z; ov[3] = (z1.w - z2.w) - z3.w; }
#pragma unroll
    for (int e = 0; e < 4; ++e) ov[e] = (ov[e] + bs[e] + rsv[oz][e]) * cs[e];
    float* yp = a.y + vox[oz] * a.yld + co4;
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
        const float xv = gxv[oz][e];
        const float u = xv * gsc[e] + gsh[e];
        const float du = u > 0.f ? ov[e] : ov[e] * a.g.gslope;
        s0[e] += du; s1[e] += du * ((xv - gmean[e]) * grstd[e]);
      }
    }
  }
  if constexpr (FUSE != 0) wino_fuse_records<FUSE, 16>(a, P, tid, lane, wave, coq, co_base, n, tz0, ty0, tx0, cnt, K0, s0, s1);
}
#undef W3_PHASE_END
#undef W3_MF
#undef W3_SGB

// ---- filter transform (pack_values.h: pack_wino3_item) ----
__global__ void wino3_pack_weight_kernel(const float* w, float* up, int cout, int cin, int coutP, int cinP, int mode) {
  const size_t items = (size_t)cinP * coutP;               // one per (ci, co): 64 points each
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < items; r += (size_t)gridDim.x * blockDim.x)
    pack_wino3_item(w, up, r, cout, cin, coutP, cinP, mode);
}

// cout / cin: the PACKED roles (out / in channels of THIS conv). cin is padded to the channel quad, cout to the 32-channel tile.
extern "C" size_t mi355_wino3d_weight_elems(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t coutP = (cout + 31) / 32 * 32, cinP = (cin + 3) / 4 * 4;
  return (size_t)64 * cinP * coutP;
}

extern "C" int mi355_wino3d_pack_weight(const float* w, float* up, int32_t cout, int32_t cin, int32_t mode, void* stream) {
  if (!w || !up || cout <= 0 || cin <= 0 || mode < 0 || mode > 1) return MI355_EINVAL;
  const int coutP = (cout + 31) / 32 * 32, cinP = (cin + 3) / 4 * 4;
  const size_t items = (size_t)cinP * coutP;
  int grid = (int)((items + 255) / 256); if (grid > 4096) grid = 4096;
  LAUNCH(wino3_pack_weight_kernel, dim3(grid), dim3(256), 0, stream, w, up, cout, cin, coutP, cinP, mode);
  return LAUNCH_CHECK();
}

// the calls mi355_conv3d_wino3d_fwd accepts: those of mi355_conv3d_wino_fwd whose norm-prologue parameters fit beside the rings
static int wino3_check(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  const int rc = wino_check(x, y, d);
  if (rc) return rc;
  if (x->c > 1024) return MI355_EUNSUPPORTED;
  return MI355_OK;
}

extern "C" int mi355_conv3d_wino3d_supported(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  return wino3_check(x, y, d) == MI355_OK;
}

// x, y, desc as mi355_conv3d_wino_fwd; up: mi355_wino3d_pack_weight of the weights. Statistics records: one per 2 x 8 x 16 tile
// (mi355_conv3d_wino_stats_blocks, the geometry of the 2-D kernel).
extern "C" int mi355_conv3d_wino3d_fwd(const mi355_act* x, const float* up, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!up || ((uintptr_t)up & 15)) return MI355_EINVAL;
  { const int rc = wino3_check(x, y, d); if (rc) return rc; }
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
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.CinP = x->c;
  a.Cout = y->c; a.CoutP = (y->c + 31) / 32 * 32;
  a.tilesZ = ceil_div(a.D, 2); a.tilesY = ceil_div(a.H, 8); a.tilesX = ceil_div(a.W, 16); a.coTiles = a.CoutP / 32;
  a.vec4 = a.Cout % 4 == 0 && a.yld % 4 == 0 && !((uintptr_t)a.y & 15) && (!a.res || (a.resld % 4 == 0 && !((uintptr_t)a.res & 15))) &&
           (!a.g.gnb || (a.g.gxld % 4 == 0 && !((uintptr_t)a.g.gx & 15)));
  const long long blocks = (long long)a.N * a.tilesZ * a.tilesY * a.tilesX * a.coTiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
  const dim3 grid((unsigned)blocks), blk(1024);
  // 4 staged chunks (52 224 bytes) + 3 weight slabs (98 304) + norm prologue; the exchange of the epilogue (147 456) lives inside: one
  // workgroup per CU
  const int lds_bytes = (4 * 3264 + 3 * 8192 + 3 * a.Cin) * (int)sizeof(float);
#define WINO3_LAUNCH(IM, FU)                                                                         \
  do { SET_MAX_DYN_LDS((conv3d_wino3d<IM, FU>), lds_bytes);                                            \
       LAUNCH((conv3d_wino3d<IM, FU>), grid, blk, lds_bytes, stream, a); } while (0)
  if (a.g.mom) {
    if (d->in_mode == MI355_IN_PLAIN) WINO3_LAUNCH(MI355_IN_PLAIN, 1); else WINO3_LAUNCH(MI355_IN_AFFINE_ACT, 1);
  } else if (a.g.gnb) {
    WINO3_LAUNCH(MI355_IN_PLAIN, 2);
  } else if (d->in_mode == MI355_IN_PLAIN) WINO3_LAUNCH(MI355_IN_PLAIN, 0);
  else WINO3_LAUNCH(MI355_IN_AFFINE_ACT, 0);
#undef WINO3_LAUNCH
  return LAUNCH_CHECK();
}
