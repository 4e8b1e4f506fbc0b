// 3x3x3 stride-1 conv3d forward / dgrad in exact-type fp32 with FEWER multiplications: Winograd F(2x2, 3x3) in the (y, x) plane,
// direct along z. PREPARED ON THE CPU EMULATOR, NOT YET MEASURED ON AN MI355X (tools/NEXT.md "The one algorithmic lever left").
//
// Same op as conv3d_fwd.hip (reference: unet3d/models/pytorch/classification/resnet.py:12-22 called from myronenko.py:17-21; the
// GroupNorm-apply + ReLU prologue and the bias / residual / Dropout3d-scale epilogue are fused the same way). Arithmetic per output
// voxel and (ci, co): 16 transform points per 2x2 outputs x 3 z-taps = 12 multiplications instead of 27. Numerics: the transform
// matrices have entries 0, +-1, +-1/2; measured on the CPU (tools/winograd_probe.py) the fp32 result is as close to an fp64
// convolution as the direct fp32 kernel's (max error 3e-7 .. 9e-7 of max |y|).
//
//   input transform   V = B^T d B    (4x4 input window d of one channel, B^T rows: d0-d2, d1+d2, d2-d1, d1-d3)
//   filter transform  U = G g G^T    (3x3 (dy,dx) slice g of one (co, ci, dz), done once per optimizer step by the pack kernel)
//   point-wise        M[p] = sum_{ci, dz} V[p][plane z + dz - 1][ci] * U[p][dz][ci][co]          <- the MFMA work, p = 0..15
//   output transform  Y = A^T M A    (2x2 outputs, A^T rows: m0+m1+m2, m1-m2-m3)
//
// Workgroup = 256 threads = 4 waves; output tile = 2 z-planes x 8 (y) x 16 (x) voxels = per plane 4 x 8 = 32 Winograd tiles of 2x2 = the
// M dimension of one 32x32 MFMA tile; N = 32 output channels; K = 8 input channels per LDS chunk (one float4 per k-half, exactly the
// operand scheme of conv3d_mfma: lane l supplies A[tile = l & 31][k = l >> 5], four MFMAs per float4). A workgroup walks the 4 input
// planes its 2 output planes see; every input plane is staged (haloed 10 x 18 voxels, normalised + activated on the way in), transformed
// ONCE into the 16 points (32 additions per (tile, channel)) and used by the 1-2 (output plane, dz) pairs that see it. Wave w owns the
// points (i = w, j = 0..3) of both output planes: 8 accumulator tiles = 128 registers. The output transform contracts j inside the wave
// and i across the waves through LDS.
#include "hipcompat.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"

struct WinoArgs {
  const float* x; int xld;
  const float* up;                       // transformed weights [(p * 3 + dz)][ciP / 4][coP][4]
  float* y; int yld;
  const float* res; int resld;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* out_chscale; const float* bias;
  int N, D, H, W, Cin, CinP, Cout, CoutP;
  int tilesZ, tilesY, tilesX, coTiles;
  int zsplits, zper;                     // conv3d_wino2d_zring: z ranges [zs * zper, min(D, (zs + 1) * zper)) per workgroup
  int vec4;                              // conv3d_wino2d_w8: output / residual / normalised tensor take 16-byte accesses per channel quad
  GnFuseArgs g;                          // norm statistics fused into the epilogue (gn_fuse.h), as in conv3d_mfma
};

// FUSE: 0 plain epilogue, 1 + moment records of the output, 2 + norm-backward sums (dgrad): as conv3d_mfma
// PIPE: software-pipelined main loop (one barrier per input plane: the MFMAs of plane k share the instruction stream with the transform
// of plane k + 1, the global loads of plane k + 2 are in flight) instead of stage / barrier / transform / barrier / MFMA / barrier.
// BMODE 1 (with PIPE): the phase's weight fragments are requested first, the transform of the next plane runs under their latency, then
// the MFMAs -- instead of MFMAs (weights requested at their use) followed by the transform.
// BMODE 2 (with PIPE): the weight fragments of a phase's FIRST (output plane, dz) use are requested during the previous phase, those of its
// second use at its start (they arrive under the first use's MFMAs): the ISA of BMODE 0 shows every phase opening with a wait for its weights.
template <int INMODE, int FUSE = 0, bool PIPE = false, int BMODE = 0>
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD(2) void conv3d_wino2d(WinoArgs a) {      // 128 accumulator registers + <= 128 others
  constexpr int TZ = 2, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HV = HY * HX;      // output tile; haloed input plane
  constexpr int KC = 8;                  // input channels per chunk
  constexpr int XS = 12;                 // floats per staged voxel (8 + 4 pad: the transform's strided reads stay conflict-free)
  constexpr int NT = 32;                 // Winograd tiles per plane (4 x 8 of 2x2 outputs)
  constexpr int XSF = HV * XS + 16, VSF = 16 * NT * KC;          // floats per staged plane (16-byte aligned) / per transformed plane
  constexpr int LDSF = PIPE ? 2 * (XSF + VSF) : 8192;           // PIPE: two of each (49 KB); else xs | vs in 32 KB; later zs [4][2][32][32]
  static_assert(LDSF >= 8192, "the output transform exchanges 4 x 2 x 32 x 32 floats through LDS");
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* xs = lds;
  float* vs = lds + (PIPE ? 2 : 1) * XSF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int cot = b % a.coTiles; b /= a.coTiles;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
  const int n = b;
  const int co_base = cot * 32;

  f32x16 acc[TZ][4];
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[oz][j][r] = 0.f;

  const float4* up4 = reinterpret_cast<const float4*>(a.up);
  const int CQ = a.CinP / 4;
  // staging: a thread owns (halo voxel, channel quad) units; 2 quads per voxel
  // transform: thread (tile t = tid >> 3, channel c = tid & 7)
  const int tt = tid >> 3, tc = tid & 7;
  const int tty = tt >> 3, ttx = tt & 7;

  if constexpr (PIPE) {
    // phase k = (channel chunk, input plane pz): planes of a chunk are unrolled (pz, hence the (output plane, dz) pairs and the buffer
    // parity, are compile-time); the phase after (c0, 3) is (c0 + KC, 0)
    auto plane_loads = [&](int c0_, int pz_, float4 (&ld)[2], bool (&ok)[2]) {
      const int iz = tz0 - 1 + pz_;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int u = tid + 256 * k;
        const bool act = u < HV * 2;
        if (!act) u = 0;
        const int hv = u >> 1, q = u & 1;
        const int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
        const int c = c0_ + 4 * q;
        ok[k] = act && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin;
        const int izc = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1), iyc = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1), ixc = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
        ld[k] = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + izc) * a.H + iyc) * a.W + ixc) * a.xld + (c < a.Cin ? c : 0));
      }
    };
    auto plane_store = [&](float* xsb, int c0_, const float4 (&ld)[2], const bool (&ok)[2]) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int u = tid + 256 * k;
        if (u >= HV * 2) continue;
        const int hv = u >> 1, q = u & 1, c = c0_ + 4 * q;
        float4 v = ld[k];
        if (INMODE == MI355_IN_AFFINE_ACT && ok[k]) {
          const float4 sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c);
          const float4 sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c);
          float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
          if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope + c);
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
          v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
        }
        if (!ok[k]) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xsb + hv * XS + 4 * q) = v;
      }
    };
    auto transform = [&](const float* xsb, float* vsb) {
      float t[4][4];
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {          // one window column at a time: 4 values live instead of 16
        const float* col = xsb + ((2 * tty) * HX + 2 * ttx + s2) * XS + tc;
        const float d0 = col[0], d1 = col[HX * XS], d2 = col[2 * HX * XS], d3 = col[3 * HX * XS];
        t[0][s2] = d0 - d2; t[1][s2] = d1 + d2; t[2][s2] = d2 - d1; t[3][s2] = d1 - d3;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        vsb[((4 * i + 0) * NT + tt) * KC + tc] = t[i][0] - t[i][2];
        vsb[((4 * i + 1) * NT + tt) * KC + tc] = t[i][1] + t[i][2];
        vsb[((4 * i + 2) * NT + tt) * KC + tc] = t[i][2] - t[i][1];
        vsb[((4 * i + 3) * NT + tt) * KC + tc] = t[i][1] - t[i][3];
      }
    };
    auto b_loads = [&](float4 (&bfr)[TZ][4], int c0_, auto pzc) {
      constexpr int PZ = decltype(pzc)::value;
#pragma unroll
      for (int oz = 0; oz < TZ; ++oz) {
        const int dz = PZ - oz;
        if (dz < 0 || dz > 2) continue;          // compile-time after unrolling
#pragma unroll
        for (int j = 0; j < 4; ++j)
          bfr[oz][j] = up4[((size_t)((4 * wave + j) * 3 + dz) * CQ + c0_ / 4 + half) * a.CoutP + co_base + li];
      }
    };
    auto mfma_run = [&](const float* vsb, const float4 (&bfr)[TZ][4], auto pzc) {
      constexpr int PZ = decltype(pzc)::value;
#pragma unroll
      for (int oz = 0; oz < TZ; ++oz) {
        const int dz = PZ - oz;
        if (dz < 0 || dz > 2) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 af = *reinterpret_cast<const float4*>(vsb + ((4 * wave + j) * NT + li) * KC + 4 * half);
          acc[oz][j] = MFMA_32x32x2(af.x, bfr[oz][j].x, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.y, bfr[oz][j].y, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.z, bfr[oz][j].z, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.w, bfr[oz][j].w, acc[oz][j]);
        }
      }
    };
    // one phase: MFMAs of the plane in `vcur`, transform of the next plane `xnext` -> `vnext` (if any), in the order BEARLY selects
    auto phase = [&](const float* vcur, int c0_, auto pzc, bool has_next, const float* xnext, float* vnext) {
      float4 bfr[TZ][4];
      b_loads(bfr, c0_, pzc);
      if constexpr (BMODE == 1) {
        if (has_next) transform(xnext, vnext);
        SCHED_BARRIER();
        mfma_run(vcur, bfr, pzc);
      } else {
        mfma_run(vcur, bfr, pzc);
        if (has_next) transform(xnext, vnext);
      }
    };
    // BMODE 2. First use of phase pz: output plane 0 (dz = pz) for pz = 0..2, output plane 1 (dz = 2) for pz = 3; second use (pz = 1, 2):
    // output plane 1 with dz = pz - 1.
    auto b_use = [&](float4 (&bu)[4], int c0_, int dz) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bu[j] = up4[((size_t)((4 * wave + j) * 3 + dz) * CQ + c0_ / 4 + half) * a.CoutP + co_base + li];
    };
    auto mfma_use = [&](const float* vsb, const float4 (&bu)[4], f32x16 (&ac)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 af = *reinterpret_cast<const float4*>(vsb + ((4 * wave + j) * NT + li) * KC + 4 * half);
        ac[j] = MFMA_32x32x2(af.x, bu[j].x, ac[j]);
        ac[j] = MFMA_32x32x2(af.y, bu[j].y, ac[j]);
        ac[j] = MFMA_32x32x2(af.z, bu[j].z, ac[j]);
        ac[j] = MFMA_32x32x2(af.w, bu[j].w, ac[j]);
      }
    };
    // Two fragment sets, each requested one USE ahead. Single-use phases (pz = 0, 3): first use from bx, the next phase's first use is
    // requested into by. Two-use phases (pz = 1, 2): first use from bx, second use requested into by at the start, the next phase's first
    // use into bx once the first use's MFMAs have consumed it. (c0n, pzn): the next phase; pzn < 0: none.
    auto phase_pf = [&](const float* vcur, int c0_, auto pzc, bool has_next, const float* xnext, float* vnext,
                        float4 (&bx)[4], float4 (&by)[4], int c0n, int pzn) {
      constexpr int PZ = decltype(pzc)::value;
      const int dzn = pzn == 3 ? 2 : pzn;
      if constexpr (PZ == 1 || PZ == 2) {
        b_use(by, c0_, PZ - 1);
        SCHED_BARRIER();
        mfma_use(vcur, bx, acc[0]);
        if (has_next) transform(xnext, vnext);
        if (pzn >= 0) b_use(bx, c0n, dzn);
        SCHED_BARRIER();
        mfma_use(vcur, by, acc[1]);
      } else {
        if (pzn >= 0) b_use(by, c0n, dzn);
        SCHED_BARRIER();
        mfma_use(vcur, bx, acc[PZ == 3 ? 1 : 0]);
        if (has_next) transform(xnext, vnext);
      }
    };
    float4 ld[2];
    bool ok[2];
    // prologue: plane (0, 0) staged and transformed, plane (0, 1) staged
    plane_loads(0, 0, ld, ok);
    plane_store(xs, 0, ld, ok);
    plane_loads(0, 1, ld, ok);
    __syncthreads();
    transform(xs, vs);
    plane_store(xs + XSF, 0, ld, ok);
    __syncthreads();
    float4 bA[4], bB[4];
    if constexpr (BMODE == 2) b_use(bA, 0, 0);                   // first use of phase (0, 0)
    for (int c0 = 0; c0 < a.CinP; c0 += KC) {
      const bool more = c0 + KC < a.CinP;                 // another chunk follows (workgroup-uniform)
      if constexpr (BMODE == 2) {
        plane_loads(c0, 2, ld, ok);
        phase_pf(vs, c0, std::integral_constant<int, 0>(), true, xs + XSF, vs + VSF, bA, bB, c0, 1);      // next first use -> bB
        plane_store(xs, c0, ld, ok);
        __syncthreads();
        plane_loads(c0, 3, ld, ok);
        phase_pf(vs + VSF, c0, std::integral_constant<int, 1>(), true, xs, vs, bB, bA, c0, 2);                // second use in bA, next first -> bB
        plane_store(xs + XSF, c0, ld, ok);
        __syncthreads();
        if (more) plane_loads(c0 + KC, 0, ld, ok);
        phase_pf(vs, c0, std::integral_constant<int, 2>(), true, xs + XSF, vs + VSF, bB, bA, c0, 3);      // second use in bA, next first -> bB
        if (more) plane_store(xs, c0 + KC, ld, ok);
        __syncthreads();
        if (more) plane_loads(c0 + KC, 1, ld, ok);
        phase_pf(vs + VSF, c0, std::integral_constant<int, 3>(), more, xs, vs, bB, bA, c0 + KC, more ? 0 : -1);      // next chunk's first use -> bA
        if (more) plane_store(xs + XSF, c0 + KC, ld, ok);
        __syncthreads();
        continue;
      }
      // phase (c0, 0): MFMA plane 0 | transform plane 1 | loads of plane 2
      plane_loads(c0, 2, ld, ok);
      phase(vs, c0, std::integral_constant<int, 0>(), true, xs + XSF, vs + VSF);
      plane_store(xs, c0, ld, ok);
      __syncthreads();
      // phase (c0, 1): MFMA plane 1 | transform plane 2 | loads of plane 3
      plane_loads(c0, 3, ld, ok);
      phase(vs + VSF, c0, std::integral_constant<int, 1>(), true, xs, vs);
      plane_store(xs + XSF, c0, ld, ok);
      __syncthreads();
      // phase (c0, 2): MFMA plane 2 | transform plane 3 | loads of the next chunk's plane 0
      if (more) plane_loads(c0 + KC, 0, ld, ok);
      phase(vs, c0, std::integral_constant<int, 2>(), true, xs + XSF, vs + VSF);
      if (more) plane_store(xs, c0 + KC, ld, ok);
      __syncthreads();
      // phase (c0, 3): MFMA plane 3 | transform of the next chunk's plane 0 | loads of its plane 1
      if (more) plane_loads(c0 + KC, 1, ld, ok);
      phase(vs + VSF, c0, std::integral_constant<int, 3>(), more, xs, vs);
      if (more) plane_store(xs + XSF, c0 + KC, ld, ok);
      __syncthreads();
    }
  } else {
  for (int c0 = 0; c0 < a.CinP; c0 += KC) {
    for (int pz = 0; pz < TZ + 2; ++pz) {
      const int iz = tz0 - 1 + pz;
      __syncthreads();                   // the previous plane's MFMAs are done with vs, its transform with xs
      // ---- stage the haloed input plane iz, channels [c0, c0 + 8) ----
      for (int u = tid; u < HV * 2; u += 256) {
        const int hv = u >> 1, q = u & 1;
        const int hy = hv / HX, hx = hv % HX;
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const int c = c0 + 4 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && c < a.Cin) {
          v = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + iz) * a.H + iy) * a.W + ix) * a.xld + c);
          if (INMODE == MI355_IN_AFFINE_ACT) {
            const float4 sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c);
            const float4 sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c);
            float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
            if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope + c);
            v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
            v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
          }
        }
        *reinterpret_cast<float4*>(xs + hv * XS + 4 * q) = v;
      }
      __syncthreads();
      // ---- input transform: V = B^T d B of the 4x4 window of tile (tty, ttx), channel tc ----
      {
        float d[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int s = 0; s < 4; ++s) d[r][s] = xs[((2 * tty + r) * HX + 2 * ttx + s) * XS + tc];
        float t[4][4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {          // rows: B^T d
          t[0][s] = d[0][s] - d[2][s]; t[1][s] = d[1][s] + d[2][s]; t[2][s] = d[2][s] - d[1][s]; t[3][s] = d[1][s] - d[3][s];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {          // columns: (B^T d) B
          const float v0 = t[i][0] - t[i][2], v1 = t[i][1] + t[i][2], v2 = t[i][2] - t[i][1], v3 = t[i][1] - t[i][3];
          vs[((4 * i + 0) * NT + tt) * KC + tc] = v0;
          vs[((4 * i + 1) * NT + tt) * KC + tc] = v1;
          vs[((4 * i + 2) * NT + tt) * KC + tc] = v2;
          vs[((4 * i + 3) * NT + tt) * KC + tc] = v3;
        }
      }
      __syncthreads();
      // ---- point-wise products: wave w owns points (w, 0..3); this input plane serves output plane oz with z-tap dz = pz - oz ----
#pragma unroll
      for (int oz = 0; oz < TZ; ++oz) {
        const int dz = pz - oz;
        if (dz < 0 || dz > 2) continue;        // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int p = 4 * wave + j;
          const float4 af = *reinterpret_cast<const float4*>(vs + (p * NT + li) * KC + 4 * half);
          const float4 bf = up4[((size_t)(p * 3 + dz) * CQ + c0 / 4 + half) * a.CoutP + co_base + li];
          acc[oz][j] = MFMA_32x32x2(af.x, bf.x, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.y, bf.y, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.z, bf.z, acc[oz][j]);
          acc[oz][j] = MFMA_32x32x2(af.w, bf.w, acc[oz][j]);
        }
      }
    }
  }
  }

  // ---- output transform Y = A^T M A, bias / residual / dropout scale, store ----
  // inside the wave: Z[b] = sum_j A^T[b][j] M[w][j];  across the waves (LDS): Y[a][b] = sum_i A^T[a][i] Z_i[b];  wave w' then owns (a, b) = (w' >> 1, w' & 1)
  float* zs = lds;                        // [i = wave][b][tile 32][co 32]
  const int oa = wave >> 1, ob = wave & 1;
  const int co = co_base + li;
  const bool cov = co < a.Cout;
  float bs = 0.f, cs = 1.f;
  if (cov && a.bias) bs = a.bias[co];
  if (cov && a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + co];
  // FUSE 1: one-pass moments about K0 = the lane's first stored value; FUSE 2: sum du, sum du * xhat (gn_fuse.h)
  float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
  int cnt = 0;
  if constexpr (FUSE == 2) {
    const int coc = cov ? co : a.Cout - 1;
    const int grp = coc / (a.Cout / a.g.ggroups);
    gsc = a.g.gscale[(size_t)n * a.Cout + coc]; gsh = a.g.gshift[(size_t)n * a.Cout + coc];
    gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
  }
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz) {
    __syncthreads();                      // every wave is done with vs / the previous plane's zs
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;      // tile index of accumulator register r
      const float m0 = acc[oz][0][r], m1 = acc[oz][1][r], m2 = acc[oz][2][r], m3 = acc[oz][3][r];
      zs[((wave * 2 + 0) * 32 + row) * 32 + li] = m0 + m1 + m2;
      zs[((wave * 2 + 1) * 32 + row) * 32 + li] = m1 - m2 - m3;
    }
    __syncthreads();
    const int z = tz0 + oz;
    // FUSE 2: the 16 reads of the normalised tensor go out first, from clamped (always valid) addresses (conv3d_fwd.hip: inside the
    // loop every one of them is a dependent round trip behind the stores -- the first GPU measurement showed exactly that)
    float gxv[16];
    if constexpr (FUSE == 2) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        int yy = ty0 + 2 * (row >> 3) + oa, xx = tx0 + 2 * (row & 7) + ob, zc = z;
        zc = zc < a.D ? zc : a.D - 1; yy = yy < a.H ? yy : a.H - 1; xx = xx < a.W ? xx : a.W - 1;
        gxv[r] = a.g.gx[((((size_t)n * a.D + zc) * a.H + yy) * a.W + xx) * a.g.gxld + (cov ? co : a.Cout - 1)];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float z0 = zs[((0 * 2 + ob) * 32 + row) * 32 + li], z1 = zs[((1 * 2 + ob) * 32 + row) * 32 + li];
      const float z2 = zs[((2 * 2 + ob) * 32 + row) * 32 + li], z3 = zs[((3 * 2 + ob) * 32 + row) * 32 + li];
      float v = (oa == 0 ? z0 + z1 + z2 : z1 - z2 - z3) + bs;
      const int yy = ty0 + 2 * (row >> 3) + oa, xx = tx0 + 2 * (row & 7) + ob;
      if (!cov || z >= a.D || yy >= a.H || xx >= a.W) continue;
      const size_t vox = (((size_t)n * a.D + z) * a.H + yy) * a.W + xx;
      if (a.res) v += a.res[vox * a.resld + co];
      v *= cs;
      a.y[vox * a.yld + co] = v;
      if constexpr (FUSE == 1) {
        if (cnt == 0) K0 = v;
        const float t = v - K0;
        s0 += t; s1 += t * t;
        ++cnt;
      } else if constexpr (FUSE == 2) {
        const float xv = gxv[r];
        const float u = xv * gsc + gsh;
        const float du = u > 0.f ? v : v * a.g.gslope;
        s0 += du; s1 += du * ((xv - gmean) * grstd);
      }
    }
  }
  if constexpr (FUSE != 0) {
    // wave w' holds the (a, b) = (w' >> 1, w' & 1) outputs of every tile: the four waves are the "WM" waves of gn_fuse_reduce_store
    constexpr int K = FUSE == 1 ? 3 : 2;
    float vals[1][K];
    if constexpr (FUSE == 1) {
      const float c = (float)cnt;
      const float m2 = cnt > 0 ? s1 - s0 * s0 / c : 0.f;
      vals[0][0] = c; vals[0][1] = s0 + c * K0; vals[0][2] = m2 > 0.f ? m2 : 0.f;
    } else {
      vals[0][0] = s0; vals[0][1] = s1;
    }
    const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
    const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
    float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * K;
    gn_fuse_reduce_store<K, 1, 4, 1>(vals, lds, wave, 0, half, li, tid, dst, co_base, a.Cout);
  }
}

// =====================================================================================================================================
// The same operation as a z-MARCHING workgroup (round 3). conv3d_wino2d above stages and transforms 4 input planes for every 2 output
// planes, so a transformed (plane, channel chunk) feeds 1.5 MFMA groups on average and the SQ counters show the price
// (profiles/r3_sq_counters_wino.txt: 8.2 vector-ALU instructions per MFMA against 2.8 in the direct kernel; MFMA pipe busy 45 %).
// Here a 512-thread workgroup owns an 8 (y) x 16 (x) voxel column x 32 output channels over a whole z range and walks the input planes
// once: every (plane, 8-channel chunk) is staged and transformed ONCE and multiplied into the three output planes that see it
// (dz = 0, 1, 2), whose accumulators stay in registers: 16 points x 3 output planes = 48 tiles over 8 waves = 6 per wave (96 registers),
// 24 MFMAs per wave and phase from 2 A fragments (ds_read_b128) and 6 weight fragments. When input plane t has been multiplied, output
// plane t - 1 is complete: its output transform (in-wave over j, across the waves over i through 64 KB of LDS), bias / residual /
// Dropout3d scale / store and the fused norm statistics run, and its accumulator slot is zeroed for output plane t + 2. The plane loop
// is unrolled by three, so the slot of every accumulator is a compile-time index.
// Software pipeline, one barrier per phase (= input plane x channel chunk): MFMAs of phase k | the weight fragments of phase k + 1 are
// requested once the MFMAs of phase k have been issued | transform of phase k + 1 | global loads of phase k + 2 in flight, written to
// the other staging buffer after the MFMAs.
// Fused statistics: one record per (sample, workgroup, channel) accumulated over the whole z range (instead of one per 2 x 8 x 16 tile).
template <int INMODE, int FUSE>
__global__ __launch_bounds__(512) void conv3d_wino2d_zring(WinoArgs a) {
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HV = HY * HX;
  constexpr int KC = 8, XS = 12, NT = 32;
  constexpr int XSF = HV * XS + 16, VSF = 16 * NT * KC, PF = 8 * 2 * 32 * 32;
  DYN_LDS(lds);
  float* xs = lds;                                         // 2 staged chunks [halo voxel][8 + 4 pad]
  float* vs = lds + 2 * XSF;                               // 2 transformed chunks [point][tile][channel]
  float* P = vs + 2 * VSF;                                 // output-transform exchange [wave][b][tile][co]
  float* prm = P + PF;                                     // norm prologue of this sample: scale | shift | slope, CinP each
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int cot = b % a.coTiles; b /= a.coTiles;
  const int txi = b % a.tilesX; b /= a.tilesX;
  const int tyi = b % a.tilesY; b /= a.tilesY;
  const int zs = b % a.zsplits; b /= a.zsplits;
  const int n = b;
  const int tx0 = txi * TX, ty0 = tyi * TY, co_base = cot * 32;
  const int zb = zs * a.zper, ze = zb + a.zper < a.D ? zb + a.zper : a.D, L = ze - zb;
  const int NC = a.CinP / KC, K = (L + 2) * NC;            // phases: input planes zb - 1 .. ze, NC chunks each

  if (INMODE == MI355_IN_AFFINE_ACT) {
    for (int c = tid; c < a.CinP; c += 512) {
      const bool in = c < a.Cin;
      prm[c] = in ? a.in_scale[(size_t)n * a.Cin + c] : 0.f;
      prm[a.CinP + c] = in ? a.in_shift[(size_t)n * a.Cin + c] : 0.f;
      prm[2 * a.CinP + c] = in ? (a.in_slope ? a.in_slope[c] : a.slope) : 0.f;
    }
  }

  f32x16 acc[3][2];                                        // [output plane slot][point q: p = 2 * wave + q]
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][q][r] = 0.f;

  // staging unit of this thread: (halo voxel sv, channel quad sq) of the 180 x 2 units of a chunk; fixed for the whole kernel
  const bool sunit = tid < HV * 2;
  const int sv = sunit ? tid >> 1 : 0, sq = tid & 1;
  const int siy = ty0 - 1 + sv / HX, six = tx0 - 1 + sv % HX;
  const bool sin = sunit && siy >= 0 && siy < a.H && six >= 0 && six < a.W;
  const size_t xplane = (size_t)a.H * a.W * a.xld;
  const float* xsrc = a.x + (size_t)n * a.D * xplane + ((size_t)(siy < 0 ? 0 : (siy < a.H ? siy : a.H - 1)) * a.W + (six < 0 ? 0 : (six < a.W ? six : a.W - 1))) * a.xld;
  // transform unit: (channel tc, tile tt) and the row half th (wave-uniform): point rows i = 2 th, 2 th + 1
  const int tc = tid & 7, tt = (tid >> 3) & 31, th = wave >> 2;
  const int tty = tt >> 3, ttx = tt & 7;
  const float4* up4 = reinterpret_cast<const float4*>(a.up);
  const int CQ = a.CinP / 4;

  float4 ld;
  bool lok;
  // a phase is (input plane index t, chunk index ci); `adv` steps it forward by one (no division in the loop)
  auto adv = [&](int& t, int& ci) { if (++ci >= NC) { ci = 0; ++t; } };
  auto loads = [&](int t, int ci) {
    const int ip = zb - 1 + t, c0 = ci * KC;
    const int c = c0 + 4 * sq;
    lok = sin && ip >= 0 && ip < a.D && c < a.Cin;
    const int ipc = ip < 0 ? 0 : (ip < a.D ? ip : a.D - 1);
    ld = *reinterpret_cast<const float4*>(xsrc + (size_t)ipc * xplane + (c < a.Cin ? c : 0));
  };
  auto commit = [&](int ci, int par) {
    if (!sunit) return;
    float4 v = ld;
    if (INMODE == MI355_IN_AFFINE_ACT) {
      const int c = ci * KC + 4 * sq;
      const float4 sc = *reinterpret_cast<const float4*>(prm + c), sh = *reinterpret_cast<const float4*>(prm + a.CinP + c);
      const float4 sl = *reinterpret_cast<const float4*>(prm + 2 * a.CinP + c);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
    }
    if (!lok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(xs + par * XSF + sv * XS + 4 * sq) = v;
  };
  auto transform = [&](int par) {
    const float* col = xs + par * XSF + ((2 * tty) * HX + 2 * ttx) * XS + tc;
    float* vd = vs + par * VSF + tt * KC + tc;             // point 4 i + j at ((4 i + j) * NT + tt) * KC + tc
    float t0[4], t1[4];
    if (th == 0) {                                         // rows i = 0: d0 - d2, i = 1: d1 + d2
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const float d0 = col[s2 * XS], d1 = col[(HX + s2) * XS], d2 = col[(2 * HX + s2) * XS];
        t0[s2] = d0 - d2; t1[s2] = d1 + d2;
      }
      vd[(0 * NT) * KC] = t0[0] - t0[2]; vd[(1 * NT) * KC] = t0[1] + t0[2]; vd[(2 * NT) * KC] = t0[2] - t0[1]; vd[(3 * NT) * KC] = t0[1] - t0[3];
      vd[(4 * NT) * KC] = t1[0] - t1[2]; vd[(5 * NT) * KC] = t1[1] + t1[2]; vd[(6 * NT) * KC] = t1[2] - t1[1]; vd[(7 * NT) * KC] = t1[1] - t1[3];
    } else {                                               // rows i = 2: d2 - d1, i = 3: d1 - d3
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const float d1 = col[(HX + s2) * XS], d2 = col[(2 * HX + s2) * XS], d3 = col[(3 * HX + s2) * XS];
        t0[s2] = d2 - d1; t1[s2] = d1 - d3;
      }
      vd[(8 * NT) * KC] = t0[0] - t0[2]; vd[(9 * NT) * KC] = t0[1] + t0[2]; vd[(10 * NT) * KC] = t0[2] - t0[1]; vd[(11 * NT) * KC] = t0[1] - t0[3];
      vd[(12 * NT) * KC] = t1[0] - t1[2]; vd[(13 * NT) * KC] = t1[1] + t1[2]; vd[(14 * NT) * KC] = t1[2] - t1[1]; vd[(15 * NT) * KC] = t1[1] - t1[3];
    }
  };
  float4 bfr[2][3];                                        // weight fragments of the next MFMA phase: [point q][dz]
  auto b_loads = [&](int ci) {
    const float4* bp = up4 + (size_t)(2 * ci + half) * a.CoutP + co_base + li;
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int dz = 0; dz < 3; ++dz) bfr[q][dz] = bp[(size_t)(((2 * wave + q) * 3 + dz) * CQ) * a.CoutP];
  };

  // ---- epilogue state: fused statistics accumulate over the whole z range ----
  const int ea = (wave & 3) >> 1, eb = wave & 1, eth = wave >> 2;    // this wave's (a, b) output of every 2x2 tile, and its tile half
  const int co = co_base + li;
  const bool cov = co < a.Cout;
  float bs = 0.f, cs = 1.f;
  if (cov && a.bias) bs = a.bias[co];
  if (cov && a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + co];
  float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
  int cnt = 0;
  if constexpr (FUSE == 2) {
    const int coc = cov ? co : a.Cout - 1;
    const int grp = coc / (a.Cout / a.g.ggroups);
    gsc = a.g.gscale[(size_t)n * a.Cout + coc]; gsh = a.g.gshift[(size_t)n * a.Cout + coc];
    gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
  }
  // output plane z from accumulator slot SE (its last input plane has been multiplied and the phase barrier passed)
  auto epilogue = [&](int z, auto sec) {
    constexpr int SE = decltype(sec)::value;
    // in-wave part over this wave's two j: A^T rows (1, 1, 1, 0) and (0, 1, -1, -1)
    const bool jh = wave & 1;                              // wave-uniform: j = 0, 1 or j = 2, 3
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;   // tile
      const float m0 = acc[SE][0][r], m1 = acc[SE][1][r];
      P[((wave * 2 + 0) * 32 + row) * 32 + li] = jh ? m0 : m0 + m1;
      P[((wave * 2 + 1) * 32 + row) * 32 + li] = jh ? -m0 - m1 : m1;
    }
    // reads that do not depend on the exchange go out before the barrier: the normalised tensor (FUSE 2) and the residual
    float gxv[8], rsv[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int tile = eth * 16 + half * 8 + r;
      int yy = ty0 + 2 * (tile >> 3) + ea, xx = tx0 + 2 * (tile & 7) + eb;
      yy = yy < a.H ? yy : a.H - 1; xx = xx < a.W ? xx : a.W - 1;
      const size_t vox = (((size_t)n * a.D + z) * a.H + yy) * a.W + xx;
      if constexpr (FUSE == 2) gxv[r] = a.g.gx[vox * a.g.gxld + (cov ? co : a.Cout - 1)];
      rsv[r] = a.res ? a.res[vox * a.resld + (cov ? co : a.Cout - 1)] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int tile = eth * 16 + half * 8 + r;
      const float* pz = P + (eb * 32 + tile) * 32 + li;     // wave w, output column b at ((w * 2 + b) * 32 + tile) * 32 + li
      // across the waves: row i of the point grid = waves 2 i, 2 i + 1; A^T rows over i: (1, 1, 1, 0) and (0, 1, -1, -1)
      float v;
      if (ea == 0) v = (pz[0 * 2048] + pz[1 * 2048]) + (pz[2 * 2048] + pz[3 * 2048]) + (pz[4 * 2048] + pz[5 * 2048]);
      else v = (pz[2 * 2048] + pz[3 * 2048]) - (pz[4 * 2048] + pz[5 * 2048]) - (pz[6 * 2048] + pz[7 * 2048]);
      v += bs;
      const int yy = ty0 + 2 * (tile >> 3) + ea, xx = tx0 + 2 * (tile & 7) + eb;
      if (!cov || yy >= a.H || xx >= a.W) continue;
      const size_t vox = (((size_t)n * a.D + z) * a.H + yy) * a.W + xx;
      v += rsv[r];
      v *= cs;
      a.y[vox * a.yld + co] = v;
      if constexpr (FUSE == 1) {
        if (cnt == 0) K0 = v;
        const float t = v - K0;
        s0 += t; s1 += t * t;
        ++cnt;
      } else if constexpr (FUSE == 2) {
        const float xv = gxv[r];
        const float u = xv * gsc + gsh;
        const float du = u > 0.f ? v : v * a.g.gslope;
        s0 += du; s1 += du * ((xv - gmean) * grstd);
      }
    }
  };

  // ---- pipeline prologue: phase 0 staged and transformed, phase 1 staged, the weight fragments of phase 0 requested ----
  __syncthreads();                                         // prm
  int t1 = 0, c1 = 0;                                      // phase k + 1 and k + 2 of the loop below, kept one / two steps ahead
  loads(0, 0);
  commit(0, 0);
  adv(t1, c1);
  if (K > 1) loads(t1, c1);
  __syncthreads();
  transform(0);
  if (K > 1) commit(c1, 1);
  b_loads(0);
  __syncthreads();
  int t2 = t1, c2 = c1;
  adv(t2, c2);

  // one phase; SL = slot of the output plane with the index of this input plane (t % 3)
  auto phase = [&](int k, auto slc) {
    constexpr int SL = decltype(slc)::value, S0 = (SL + 1) % 3, S1 = SL, S2 = (SL + 2) % 3;   // dz = 0 -> plane t + 1, 1 -> t, 2 -> t - 1
    const int par = k & 1;
    if (k + 2 < K) loads(t2, c2);
    SCHED_BARRIER();
    const float* vb = vs + par * VSF + ((2 * wave) * NT + li) * KC + 4 * half;
    const float4 af0 = *reinterpret_cast<const float4*>(vb), af1 = *reinterpret_cast<const float4*>(vb + NT * KC);
#define WZ_MFMA(e)                                                            \
    acc[S0][0] = MFMA_32x32x2(af0.e, bfr[0][0].e, acc[S0][0]);                \
    acc[S1][0] = MFMA_32x32x2(af0.e, bfr[0][1].e, acc[S1][0]);                \
    acc[S2][0] = MFMA_32x32x2(af0.e, bfr[0][2].e, acc[S2][0]);                \
    acc[S0][1] = MFMA_32x32x2(af1.e, bfr[1][0].e, acc[S0][1]);                \
    acc[S1][1] = MFMA_32x32x2(af1.e, bfr[1][1].e, acc[S1][1]);                \
    acc[S2][1] = MFMA_32x32x2(af1.e, bfr[1][2].e, acc[S2][1]);
    WZ_MFMA(x) WZ_MFMA(y) WZ_MFMA(z) WZ_MFMA(w)
#undef WZ_MFMA
    SCHED_BARRIER();
    if (k + 1 < K) {
      b_loads(c1);                                         // into the registers the MFMAs above have read: in flight under the transform
      transform(par ^ 1);
    }
    if (k + 2 < K) commit(c2, par);
    __syncthreads();
    t1 = t2; c1 = c2;
    adv(t2, c2);
  };
  // input plane index t = 0 .. L + 1 (plane zb - 1 + t); after it, output plane index t - 1 (plane zb + t - 2) is complete
  auto plane = [&](int t, auto slc) {
    constexpr int SL = decltype(slc)::value, SE = (SL + 2) % 3;
    for (int ci = 0; ci < NC; ++ci) phase(t * NC + ci, slc);
    if (t >= 2) epilogue(zb + t - 2, std::integral_constant<int, SE>());
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[SE][q][r] = 0.f;
  };
  for (int t = 0; t < L + 2; t += 3) {
    plane(t, std::integral_constant<int, 0>());
    if (t + 1 >= L + 2) break;
    plane(t + 1, std::integral_constant<int, 1>());
    if (t + 2 >= L + 2) break;
    plane(t + 2, std::integral_constant<int, 2>());
  }

  if constexpr (FUSE != 0) {
    constexpr int KK = FUSE == 1 ? 3 : 2;
    float vals[1][KK];
    if constexpr (FUSE == 1) {
      const float c = (float)cnt;
      const float m2 = cnt > 0 ? s1 - s0 * s0 / c : 0.f;
      vals[0][0] = c; vals[0][1] = s0 + c * K0; vals[0][2] = m2 > 0.f ? m2 : 0.f;
    } else {
      vals[0][0] = s0; vals[0][1] = s1;
    }
    const size_t rec = (size_t)n * ((size_t)a.zsplits * a.tilesY * a.tilesX) + ((size_t)zs * a.tilesY + tyi) * a.tilesX + txi;
    float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * KK;
    gn_fuse_reduce_store<KK, 1, 8, 1>(vals, P, wave, 0, half, li, tid, dst, co_base, a.Cout);
  }
}

// =====================================================================================================================================
// The tile kernel again with EIGHT waves per workgroup (round 3). The SQ counters of conv3d_wino2d (profiles/r3_sq_counters_wino.txt)
// show a latency-bound kernel, not a busy one: matrix pipe 51 %, vector ALU 22 % of the SIMD cycles, the waves spend half of their
// resident time in s_waitcnt -- at 256 registers per wave only two waves share a SIMD, and both run the same barrier-separated phases.
// Here the same 2 x 8 x 16 voxel x 32 channel tile (same grid, same statistics records) is computed by 512 threads: wave w owns the two
// points p = 2 w, 2 w + 1 of both output planes = 4 accumulator tiles = 64 registers, the kernel fits 128 registers, and FOUR waves
// (two workgroups) share a SIMD. Per phase a wave reads its two A fragments once (both uses of a two-use phase share them), a thread
// stages one float4 and transforms half a (tile, channel) window (two point rows, as conv3d_wino2d_zring), the norm prologue comes from
// LDS. Weight fragments: two rotating sets, each requested one use ahead (the BMODE 2 order of conv3d_wino2d).
// Output transform: in-wave over the wave's two j, across the waves through a 64 KB exchange (the main loop's LDS, reused).
#ifndef WINO_ABL
#define WINO_ABL 0            // developer ablations of conv3d_wino2d_w8 (tools/build_variant.sh ... -DWINO_ABL=mask): 1 no input loads, 2 no weight loads, 4 no transform, 8 no stores
#endif
template <int INMODE, int FUSE>
__global__ __launch_bounds__(512) MIN_WAVES_PER_SIMD(4) void conv3d_wino2d_w8(WinoArgs a) {
  constexpr int TZ = 2, TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HV = HY * HX;
  constexpr int KC = 8, XS = 12, NT = 32;
  constexpr int XSF = HV * XS + 16, VSF = 16 * NT * KC, PF = 8 * 2 * 32 * 32;
  static_assert(2 * (XSF + VSF) <= PF, "the main loop's buffers live inside the exchange area");
  DYN_LDS(lds);
  float* xs = lds;                                         // 2 staged planes [halo voxel][8 + 4 pad]
  float* vs = lds + 2 * XSF;                               // 2 transformed planes [point][tile][channel]
  float* P = lds;                                          // epilogue: output-transform exchange [wave][b][tile][co]
  float* prm = lds + PF;                                   // norm prologue of this sample: scale | shift | slope, CinP each
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  // Workgroup -> (channel tile, spatial tile). The hardware deals consecutive workgroups round-robin to the 8 XCDs (each with its own
  // L2), so the plain order b = spatial * coTiles + cot gives XCD x the channel tiles cot = x mod coTiles (their weights stay in that
  // L2: good) but every (8 / coTiles)-th spatial tile (the halo and the z overlap of neighbouring tiles are fetched once per XCD).
  // Where the numbers divide, XCD x = (cot, group g) takes a CONTIGUOUS range of spatial tiles instead.
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
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
  const int n = b;
  const int co_base = cot * 32;

  if (INMODE == MI355_IN_AFFINE_ACT) {
    for (int c = tid; c < a.CinP; c += 512) {
      const bool in = c < a.Cin;
      prm[c] = in ? a.in_scale[(size_t)n * a.Cin + c] : 0.f;
      prm[a.CinP + c] = in ? a.in_shift[(size_t)n * a.Cin + c] : 0.f;
      prm[2 * a.CinP + c] = in ? (a.in_slope ? a.in_slope[c] : a.slope) : 0.f;
    }
  }

  f32x16 acc[TZ][2];                                       // [output plane][point q: p = 2 * wave + q]
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[oz][q][r] = 0.f;

  // staging unit of this thread: (halo voxel sv, channel quad sq) of the 180 x 2 units of a plane chunk; fixed for the whole kernel.
  // Addresses are a workgroup-uniform base (scalar registers) plus one 32-bit lane offset inside the plane.
  const bool sunit = tid < HV * 2;
  const int sv = sunit ? tid >> 1 : 0, sq = tid & 1;
  const int siy = ty0 - 1 + sv / HX, six = tx0 - 1 + sv % HX;
  const bool sin = sunit && siy >= 0 && siy < a.H && six >= 0 && six < a.W;
  const size_t xplane = (size_t)a.H * a.W * a.xld;
  const float* xn = a.x + (size_t)n * a.D * xplane;        // sample n
  const unsigned xoff = (unsigned)(((siy < 0 ? 0 : (siy < a.H ? siy : a.H - 1)) * a.W + (six < 0 ? 0 : (six < a.W ? six : a.W - 1))) * a.xld + 4 * sq);
  const unsigned soff = (unsigned)(sv * XS + 4 * sq);      // staged position
  const bool sq0 = sq == 0;
  // transform unit: (channel tc, tile tt) and the row half th (wave-uniform): point rows i = 2 th, 2 th + 1
  const int tc = tid & 7, tt = (tid >> 3) & 31, th = wave >> 2;
  const int tty = tt >> 3, ttx = tt & 7;
  const float4* up4 = reinterpret_cast<const float4*>(a.up);
  const int CQ = a.CinP / 4;

  float4 ld = make_float4(0.f, 0.f, 0.f, 0.f);
  bool lok = false;
  auto loads = [&](int c0_, int pz_) {
    if (!sunit) return;
    const int iz = tz0 - 1 + pz_;
    const bool call = c0_ + 8 <= a.Cin;                    // both quads of the chunk exist (Cin is a multiple of 4)
    lok = sin && iz >= 0 && iz < a.D && (call || sq0);
    const int izc = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1);
    const float* base = xn + (size_t)izc * xplane + c0_;  // uniform
#if WINO_ABL & 1
    ld = make_float4((float)izc, (float)c0_, 1.f, 2.f); (void)base;
#else
    ld = *reinterpret_cast<const float4*>(base + (call ? xoff : xoff - 4u * sq));      // a missing quad re-reads the first one (zeroed by lok)
#endif
  };
  auto commit = [&](float* xsb, int c0_) {
    if (!sunit) return;
    float4 v = ld;
    if (INMODE == MI355_IN_AFFINE_ACT) {
      const int c = c0_ + 4 * sq;
      const float4 sc = *reinterpret_cast<const float4*>(prm + c), sh = *reinterpret_cast<const float4*>(prm + a.CinP + c);
      const float4 sl = *reinterpret_cast<const float4*>(prm + 2 * a.CinP + c);
      v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
      v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
    }
    if (!lok) v = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(xsb + soff) = v;
  };
  // Input transform of this thread's (tile, channel), two of the four point rows. `th` (waves 0-3: 0, waves 4-7: 1) selects them WITHOUT
  // a branch, so that a phase is one basic block and its transform can be spread between the MFMAs (scheduler directives in `chunk`):
  //   th = 0 (window rows d0 d1 d2): u = d0 - d2 -> point row 0, o = d1 + d2 -> point row 1
  //   th = 1 (window rows d1 d2 d3): u = d1 - d3 -> point row 3, o = d2 - d1 -> point row 2        (r0 r1 r2 = the rows read, base contains th)
  //   u = r0 - r2,  o = r1 + sg * (th ? r0 : r2),  sg = th ? -1 : +1       (x * (+-1) + y is exact)
  const float sg = th ? -1.f : 1.f;
  const int toff = ((2 * tty + th) * HX + 2 * ttx) * XS + tc;                 // first window value read in a staged plane
  const int uoff = ((th ? 12 : 0) * NT + tt) * KC + tc, ooff = ((th ? 8 : 4) * NT + tt) * KC + tc;      // point (4 i + j) at ((4 i + j) * NT + tt) * KC + tc
  struct TIn { float d[3][4]; };
  auto transform_reads = [&](TIn& t, const float* xsb) {
#if WINO_ABL & 4
    return;
#endif
    const float* col = xsb + toff;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) t.d[r][s2] = col[(r * HX + s2) * XS];
  };
  auto transform_math = [&](const TIn& t, float* vsb) {
#if WINO_ABL & 4
    return;
#endif
    float u[4], o[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      u[s2] = t.d[0][s2] - t.d[2][s2];
      o[s2] = fmaf(th ? t.d[0][s2] : t.d[2][s2], sg, t.d[1][s2]);
    }
    float* vu = vsb + uoff;
    float* vo = vsb + ooff;
    vu[(0 * NT) * KC] = u[0] - u[2]; vu[(1 * NT) * KC] = u[1] + u[2]; vu[(2 * NT) * KC] = u[2] - u[1]; vu[(3 * NT) * KC] = u[1] - u[3];
    vo[(0 * NT) * KC] = o[0] - o[2]; vo[(1 * NT) * KC] = o[1] + o[2]; vo[(2 * NT) * KC] = o[2] - o[1]; vo[(3 * NT) * KC] = o[1] - o[3];
  };
  auto transform = [&](const float* xsb, float* vsb) { TIn t; transform_reads(t, xsb); transform_math(t, vsb); };
  // weight fragments of one use: the wave's two points, z-tap dz, channels [c0_, c0_ + 8): uniform base + one 32-bit lane offset
  const unsigned boff = (unsigned)(half * a.CoutP + co_base + li);          // in float4s
  const size_t bstep = (size_t)CQ * a.CoutP;               // float4s between consecutive (point, dz) slabs
  auto b_use = [&](float4 (&bu)[2], int c0_, int dz) {
    const float4* q0 = up4 + (size_t)(c0_ / 4) * a.CoutP + (size_t)((2 * wave) * 3 + dz) * bstep;      // uniform
#if WINO_ABL & 2
    bu[0] = make_float4((float)c0_, (float)dz, 0.5f, 0.25f); bu[1] = make_float4((float)dz, (float)c0_, 0.25f, 0.5f); (void)q0;
#else
    bu[0] = q0[boff];
    bu[1] = (q0 + 3 * bstep)[boff];
#endif
  };
  auto mfma_use = [&](const float4 (&af)[2], const float4 (&bu)[2], f32x16 (&ac)[2]) {
    ac[0] = MFMA_32x32x2(af[0].x, bu[0].x, ac[0]);
    ac[1] = MFMA_32x32x2(af[1].x, bu[1].x, ac[1]);
    ac[0] = MFMA_32x32x2(af[0].y, bu[0].y, ac[0]);
    ac[1] = MFMA_32x32x2(af[1].y, bu[1].y, ac[1]);
    ac[0] = MFMA_32x32x2(af[0].z, bu[0].z, ac[0]);
    ac[1] = MFMA_32x32x2(af[1].z, bu[1].z, ac[1]);
    ac[0] = MFMA_32x32x2(af[0].w, bu[0].w, ac[0]);
    ac[1] = MFMA_32x32x2(af[1].w, bu[1].w, ac[1]);
  };
  // One channel chunk = 4 phases (input planes pz = 0..3 of the tile). The weights W[dz] of a chunk are loaded ONCE: input plane pz
  // multiplies W[pz] into output plane 0 and W[pz - 1] into output plane 1, so W[dz] serves two consecutive phases from the same
  // registers (the ablation run profiles/r3_wino_w8_ablation.txt prices the weight stream from L2 at 12-20 % of the kernel when every
  // use re-requests it). Two fragment sets rotate: on entry S0 holds W0 and S1 is free; phase 0 requests W1 into S1; phase 1 runs
  // (plane 1, W0), then requests W2 into S0, then (plane 0, W1); phase 2 runs (plane 1, W1), requests the NEXT chunk's W0 into S1, then
  // (plane 0, W2); phase 3 runs (plane 1, W2). On exit S1 holds the next W0 (moved to S0 by the caller). Every request has at
  // least one phase of lead, and a phase-opening request precedes the phase's input-plane loads in program order (vmcnt retires in
  // order: a weight wait must never include a younger HBM load).
  auto a_frags = [&](float4 (&af)[2], const float* vcur) {
    const float* vb = vcur + ((2 * wave) * NT + li) * KC + 4 * half;
    af[0] = *reinterpret_cast<const float4*>(vb);
    af[1] = *reinterpret_cast<const float4*>(vb + NT * KC);
  };
  // Scheduler directives of a phase body (one basic block: A fragments, transform reads, MFMAs, weight request, transform arithmetic
  // and writes): the transform's LDS reads go out behind the first MFMAs, its arithmetic and writes ride in the shadow of the others
  // (a 32x32x2 fp32 MFMA holds the pipe for 64 cycles), the weight request of a two-use phase is issued once the first use's MFMAs
  // (which read the registers it overwrites) are out.
#ifdef MI355_EMU
#define W8_PATTERN_ONE_USE()
#define W8_PATTERN_TWO_USE()
#else
#define W8_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#define W8_PATTERN_ONE_USE() do {                                                            \
    W8_SGB(0x100, 2);                                                                        \
    W8_SGB(0x008, 1); W8_SGB(0x100, 3); W8_SGB(0x008, 1); W8_SGB(0x100, 3);                  \
    W8_SGB(0x008, 1); W8_SGB(0x002, 4); W8_SGB(0x008, 1); W8_SGB(0x002, 4);                  \
    W8_SGB(0x008, 1); W8_SGB(0x002, 4);                                                      \
    W8_SGB(0x008, 1); W8_SGB(0x002, 3); W8_SGB(0x200, 2);                                    \
    W8_SGB(0x008, 1); W8_SGB(0x002, 3); W8_SGB(0x200, 2);                                    \
    W8_SGB(0x008, 1); W8_SGB(0x002, 3); W8_SGB(0x200, 2); } while (0)
#define W8_PATTERN_TWO_USE() do {                                                            \
    W8_SGB(0x100, 2);                                                                        \
    W8_SGB(0x008, 1); W8_SGB(0x100, 2); W8_SGB(0x008, 1); W8_SGB(0x100, 2);                  \
    W8_SGB(0x008, 1); W8_SGB(0x100, 2);                                                      \
    W8_SGB(0x008, 1); W8_SGB(0x002, 3); W8_SGB(0x008, 1); W8_SGB(0x002, 3);                  \
    W8_SGB(0x008, 1); W8_SGB(0x002, 3); W8_SGB(0x008, 1); W8_SGB(0x002, 3);                  \
    W8_SGB(0x008, 1); W8_SGB(0x020, 2);                                                      \
    W8_SGB(0x008, 1); W8_SGB(0x002, 2); W8_SGB(0x200, 1);                                    \
    W8_SGB(0x008, 1); W8_SGB(0x002, 2); W8_SGB(0x200, 1);                                    \
    W8_SGB(0x008, 1); W8_SGB(0x002, 2); W8_SGB(0x200, 1);                                    \
    W8_SGB(0x008, 1); W8_SGB(0x002, 2); W8_SGB(0x200, 1);                                    \
    W8_SGB(0x008, 4); } while (0)
#endif
  auto chunk = [&](int c0, float4 (&S0)[2], float4 (&S1)[2]) {
    const bool more = c0 + KC < a.CinP;                    // another chunk follows (workgroup-uniform)
    const int cn = more ? c0 + KC : c0;                    // the weight request of phase 2 is unconditional (no branch inside a phase body)
    float4 af[2];
    TIn t;
    // phase 0: plane 0 x W0 -> output plane 0 | transform plane 1 | loads of plane 2
    b_use(S1, c0, 1);
    loads(c0, 2);
    SCHED_BARRIER();
    a_frags(af, vs);
    transform_reads(t, xs + XSF);
    mfma_use(af, S0, acc[0]);
    transform_math(t, vs + VSF);
    W8_PATTERN_ONE_USE();
    SCHED_BARRIER();
    commit(xs, c0);
    __syncthreads();
    // phase 1: plane 1 x W0 -> output plane 1, x W1 -> output plane 0 | transform plane 2 | loads of plane 3
    loads(c0, 3);
    SCHED_BARRIER();
    a_frags(af, vs + VSF);
    transform_reads(t, xs);
    mfma_use(af, S0, acc[1]);
    b_use(S0, c0, 2);
    transform_math(t, vs);
    mfma_use(af, S1, acc[0]);
    W8_PATTERN_TWO_USE();
    SCHED_BARRIER();
    commit(xs + XSF, c0);
    __syncthreads();
    // phase 2: plane 2 x W1 -> output plane 1, x W2 -> output plane 0 | transform plane 3 | loads of the next chunk's plane 0
    if (more) loads(c0 + KC, 0);
    SCHED_BARRIER();
    a_frags(af, vs);
    transform_reads(t, xs + XSF);
    mfma_use(af, S1, acc[1]);
    b_use(S1, cn, 0);
    transform_math(t, vs + VSF);
    mfma_use(af, S0, acc[0]);
    W8_PATTERN_TWO_USE();
    SCHED_BARRIER();
    if (more) commit(xs, c0 + KC);
    __syncthreads();
    // phase 3: plane 3 x W2 -> output plane 1 | transform of the next chunk's plane 0 (after the last chunk: of a stale buffer, into a
    // buffer nobody reads) | loads of its plane 1
    if (more) loads(c0 + KC, 1);
    SCHED_BARRIER();
    a_frags(af, vs + VSF);
    transform_reads(t, xs);
    mfma_use(af, S0, acc[1]);
    transform_math(t, vs);
    W8_PATTERN_ONE_USE();
    SCHED_BARRIER();
    if (more) commit(xs + XSF, c0 + KC);
    __syncthreads();
  };

  // prologue: planes 0 and 1 of the first chunk requested together, W0 requested; plane 0 staged and transformed, plane 1 staged
  __syncthreads();                                         // prm
  float4 bA[2], bB[2];
  b_use(bA, 0, 0);
  loads(0, 0);
  const float4 ld0 = ld;
  const bool lok0 = lok;
  loads(0, 1);
  { const float4 ld1 = ld; const bool lok1 = lok; ld = ld0; lok = lok0; commit(xs, 0); ld = ld1; lok = lok1; }
  commit(xs + XSF, 0);
  __syncthreads();
  transform(xs, vs);
  __syncthreads();
  for (int c0 = 0; c0 < a.CinP; c0 += KC) {
    chunk(c0, bA, bB);
    bA[0] = bB[0]; bA[1] = bB[1];                          // the next chunk's W0 (8 register moves per 4 phases keep ONE loop body)
  }

  // ---- output transform Y = A^T M A, bias / residual / dropout scale, store ----
  // wave w = (i = w >> 1, j half = w & 1). In-wave over its two j: A^T rows (1, 1, 1, 0) and (0, 1, -1, -1), written to the exchange
  // P[wave][b][tile][co]; across the waves over i on the way out. The read side is VOXEL-major: a thread owns 4 consecutive channels
  // (quad coq = tid & 7) of two output voxels per plane, so the exchange is read with ds_read_b128, residual / normalised tensor come
  // in and the result goes out as 16-byte accesses (8 lanes = the 32 channels of a voxel, a wave = 8 consecutive voxels of a row; the
  // ablation run priced the first, channel-per-lane epilogue -- 48 ds_read_b32 and 8 dword stores per lane and plane -- at 10 % of
  // the layer set, profiles/r3_wino_w8_ablation.txt). `a.vec4` == 0 (an output, residual or normalised tensor that is not 16-byte
  // aligned per voxel, or a channel count that is not a multiple of 4) takes the same path with scalar accesses.
  const bool jh = wave & 1;
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
  float* pw = P + ((wave * 2) * 32 + 4 * half) * 32 + li;  // this lane's partial rows: + (b * 32 + (r & 3) + 8 * (r >> 2)) * 32
  auto ld4 = [&](const float* base, size_t off, float (&v)[4]) {          // 4 channels of a voxel; scalar where 16-byte access is not legal
    if (a.vec4) {
      const float4 t = *reinterpret_cast<const float4*>(base + off);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = co4 + e < a.Cout ? base[off + e] : 0.f;
    }
  };
#if WINO_ABL & 16
  {
    float sacc = 0.f;
#pragma unroll
    for (int oz = 0; oz < TZ; ++oz)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[oz][q][r];
    if (sacc == 12345.678f) a.y[tid] = sacc + bs[0] + cs[0];
    return;
  }
#endif
#pragma unroll
  for (int oz = 0; oz < TZ; ++oz) {
    if (oz > 0) __syncthreads();                           // the previous plane's exchange has been read (the main loop ends on a barrier)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rowoff = (r & 3) + 8 * (r >> 2);           // tile index of accumulator register r, less 4 * half
      const float m0 = acc[oz][0][r], m1 = acc[oz][1][r], sm = m0 + m1;
      pw[rowoff * 32] = jh ? m0 : sm;
      pw[(32 + rowoff) * 32] = jh ? -sm : m1;
    }
    // reads that do not depend on the exchange go out before the barrier: the normalised tensor (FUSE 2) and the residual (requesting
    // both planes' at the top of the epilogue was measured: 27 spilled registers, the norm-backward form 16 % slower)
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
      const float4* pz = reinterpret_cast<const float4*>(P + (eb * 32 + tile) * 32 + 4 * coq);      // wave w at + w * 2048 floats
      // across the waves: row i of the point grid = waves 2 i, 2 i + 1; A^T rows over i: (1, 1, 1, 0) and (0, 1, -1, -1)
      float4 o;
      if (ea == 0) {
        const float4 p0 = pz[0 * 512], p1 = pz[1 * 512], p2 = pz[2 * 512], p3 = pz[3 * 512], p4 = pz[4 * 512], p5 = pz[5 * 512];
        o.x = (p0.x + p1.x) + (p2.x + p3.x) + (p4.x + p5.x); o.y = (p0.y + p1.y) + (p2.y + p3.y) + (p4.y + p5.y);
        o.z = (p0.z + p1.z) + (p2.z + p3.z) + (p4.z + p5.z); o.w = (p0.w + p1.w) + (p2.w + p3.w) + (p4.w + p5.w);
      } else {
        const float4 p2 = pz[2 * 512], p3 = pz[3 * 512], p4 = pz[4 * 512], p5 = pz[5 * 512], p6 = pz[6 * 512], p7 = pz[7 * 512];
        o.x = (p2.x + p3.x) - (p4.x + p5.x) - (p6.x + p7.x); o.y = (p2.y + p3.y) - (p4.y + p5.y) - (p6.y + p7.y);
        o.z = (p2.z + p3.z) - (p4.z + p5.z) - (p6.z + p7.z); o.w = (p2.w + p3.w) - (p4.w + p5.w) - (p6.w + p7.w);
      }
      if (!vin[sI]) continue;
      float ov[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) ov[e] = (ov[e] + bs[e] + rsv[sI][e]) * cs[e];
      float* yp = a.y + vox[sI] * a.yld + co4;
#if WINO_ABL & 8
      if (ov[0] == 12345.678f)
#endif
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
  if constexpr (FUSE != 0) {
    // Per-lane partials of 4 channels -> one record per (tile, channel). Lanes coq + 8 m (m = 0..7) of a wave hold the same channels:
    // three xor-shuffle steps of PLAIN sums (fixed order: the lane with the lower m first), then the eight waves through LDS in wave
    // order (Chan's merge, as everywhere). Moments: a lane's sums are about its own first value K0; before the shuffles they are moved
    // to the wave's common shift Kc = K0 of lane m = 0 (sum (v - Kc) = s0 + c d, sum (v - Kc)^2 = s1 + d (2 s0 + c d), d = K0 - Kc: no
    // division, no E[x^2] - E[x]^2 of raw values), and M2 = s1 - s0^2 / c is formed once per wave and channel.
    constexpr int KK = FUSE == 1 ? 3 : 2;
    float vals[4][KK];
    float cw = (float)cnt;                                  // FUSE 1: stored voxels of this lane (the same for its four channels)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (FUSE == 1) {
        const float Kc = __shfl(K0[e], coq);
        const float d = K0[e] - Kc;
        vals[e][0] = Kc;
        vals[e][2] = s1[e] + d * (2.f * s0[e] + cw * d);
        vals[e][1] = s0[e] + cw * d;
      } else {
        vals[e][0] = s0[e]; vals[e][1] = s1[e];
      }
    }
#pragma unroll
    for (int step = 8; step < 64; step <<= 1) {
      const bool upper = lane & step;
      if constexpr (FUSE == 1) { const float oc = __shfl_xor(cw, step); cw = upper ? oc + cw : cw + oc; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = (FUSE == 1 ? 1 : 0); k < KK; ++k) {
          const float o = __shfl_xor(vals[e][k], step);
          vals[e][k] = upper ? o + vals[e][k] : vals[e][k] + o;
        }
    }
    // the eight waves through LDS: moments as (count, sum about Kc, sum of squares about Kc, Kc) per wave, moved to wave 0's shift by
    // the same identity and added in wave order; M2 = s1 - s0^2 / c once per channel
    constexpr int KW = FUSE == 1 ? 4 : 2;
    __syncthreads();                                       // every wave is done with the exchange
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* pr = P + ((wave * 32) + 4 * coq + e) * KW;
        if constexpr (FUSE == 1) { pr[0] = cw; pr[1] = vals[e][1]; pr[2] = vals[e][2]; pr[3] = vals[e][0]; }
        else { pr[0] = vals[e][0]; pr[1] = vals[e][1]; }
      }
    }
    __syncthreads();
    if (tid < 32) {
      float r[KK];
      if constexpr (FUSE == 1) {
        const float K = P[tid * KW + 3];
        float c = P[tid * KW], t0 = P[tid * KW + 1], t1 = P[tid * KW + 2];
#pragma unroll
        for (int w = 1; w < 8; ++w) {
          const float* pr = P + (w * 32 + tid) * KW;
          const float cwv = pr[0], d = pr[3] - K;
          t1 += pr[2] + d * (2.f * pr[1] + cwv * d);
          t0 += pr[1] + cwv * d;
          c += cwv;
        }
        const float m2 = c > 0.f ? t1 - t0 * t0 / c : 0.f;
        r[0] = c; r[1] = t0 + c * K; r[2] = m2 > 0.f ? m2 : 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < KK; ++k) r[k] = P[tid * KW + k];
#pragma unroll
        for (int w = 1; w < 8; ++w)
#pragma unroll
          for (int k = 0; k < KK; ++k) r[k] += P[(w * 32 + tid) * KW + k];
      }
      const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
      const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
      float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * KK;
      const int co = co_base + tid;
      if (co < a.Cout) {
#pragma unroll
        for (int k = 0; k < KK; ++k) dst[(size_t)co * KK + k] = r[k];
      }
    }
  }
}
#undef W8_SGB
#undef W8_PATTERN_ONE_USE
#undef W8_PATTERN_TWO_USE

// z-range plan of conv3d_wino2d_zring for an output [n, d, h, w, c]: ~256 workgroups (one per CU), whole z ranges of >= 8 planes
struct WinoZPlan { int tilesY, tilesX, coTiles, zsplits, zper, use; };      // use: 0 tile, 1 zring, 2 w8
#ifndef WINO_DEFAULT_FORM
#define WINO_DEFAULT_FORM 2      // conv3d_wino2d_w8: measured 30.5 ms over the layer set against 33.3 (tile) and 36.9 (zring), profiles/r3_wino_forms.txt
#endif
static WinoZPlan plan_wino_zring(int n, int d, int h, int w, int cout) {
  WinoZPlan p;
  p.tilesY = ceil_div(h, 8); p.tilesX = ceil_div(w, 16); p.coTiles = ceil_div(cout, 32);
  const long long spatial = (long long)n * p.tilesY * p.tilesX * p.coTiles;
  int zsplits = (int)((256 + spatial - 1) / spatial);
  const char* ze = getenv("MI355_WINO_ZSPLITS");            // tests: force the number of z ranges (1 = whole columns)
  if (ze && atoi(ze) > 0) zsplits = atoi(ze);
  if (zsplits < 1) zsplits = 1;
  if (zsplits > d) zsplits = d;
  p.zper = ceil_div(d, zsplits);
  p.zsplits = ceil_div(d, p.zper);
  // MI355_WINO_FORM: tile (conv3d_wino2d, 4 waves) | w8 (conv3d_wino2d_w8, the same tile with 8 waves) | zring (z-marching) |
  // auto (z-marching where a z range has >= 8 planes). Measured (profiles/r3_wino_forms.txt): zring loses to the tile form on the
  // step (36.9 vs 33.3 ms over the layer set), so it is opt-in.
  const char* fe = getenv("MI355_WINO_FORM");
  if (fe && fe[0] == 'z') p.use = 1;
  else if (fe && fe[0] == 'a') p.use = p.zper >= 8;
  else if (fe && fe[0] == 't') p.use = 0;
  else if (fe && fe[0] == 'w') p.use = 2;
  else p.use = WINO_DEFAULT_FORM;
  return p;
}

// ---- filter transform: U[(i,j)][dz][ci][co] = sum_{dy,dx} G[i][dy] G[j][dx] w[...], G rows: g0, (g0+g1+g2)/2, (g0-g1+g2)/2, g2 ----
// mode 0: forward, w OIDHW [cout][cin][3][3][3]. mode 1: dgrad of Conv3d: roles swapped ("out" = ci, "in" = co), all three taps flipped.
__global__ void wino_pack_weight_kernel(const float* w, float* up, int cout, int cin, int coutP, int cinP, int mode) {
  const size_t total = (size_t)48 * (cinP / 4) * coutP * 4;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e = idx & 3;
    size_t r = idx >> 2;
    const int o = r % coutP; r /= coutP;
    const int iq = r % (cinP / 4); r /= (cinP / 4);
    const int pd = (int)r, p = pd / 3, dz = pd % 3;
    const int pi = p >> 2, pj = p & 3;
    const int i = iq * 4 + e;
    float v = 0.f;
    if (o < cout && i < cin) {
      float g[3][3];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          g[dy][dx] = mode == 0 ? w[((size_t)o * cin + i) * 27 + (dz * 3 + dy) * 3 + dx]
                                : w[((size_t)i * cout + o) * 27 + ((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx)];      // w[co = i][ci = o], flipped
      float t[3];                           // row pi of G applied along dy
      for (int dx = 0; dx < 3; ++dx)
        t[dx] = pi == 0 ? g[0][dx] : pi == 1 ? 0.5f * (g[0][dx] + g[1][dx] + g[2][dx]) : pi == 2 ? 0.5f * (g[0][dx] - g[1][dx] + g[2][dx]) : g[2][dx];
      v = pj == 0 ? t[0] : pj == 1 ? 0.5f * (t[0] + t[1] + t[2]) : pj == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2];
    }
    up[idx] = v;
  }
}

extern "C" size_t mi355_wino_weight_elems(int32_t cout, int32_t cin) {
  if (cout <= 0 || cin <= 0) return 0;
  const size_t coutP = (cout + 31) / 32 * 32, cinP = (cin + 7) / 8 * 8;
  return (size_t)48 * cinP * coutP;
}

extern "C" int mi355_wino_pack_weight(const float* w, float* up, int32_t cout, int32_t cin, int32_t mode, void* stream) {
  if (!w || !up || cout <= 0 || cin <= 0 || mode < 0 || mode > 1) return MI355_EINVAL;
  const int coutP = (cout + 31) / 32 * 32, cinP = (cin + 7) / 8 * 8;     // logical packed dims: cout = "out", cin = "in" of THIS conv
  const size_t total = (size_t)48 * cinP * coutP;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  LAUNCH(wino_pack_weight_kernel, dim3(grid), dim3(256), 0, stream, w, up, cout, cin, coutP, cinP, mode);
  return LAUNCH_CHECK();
}

// x, y: NDHWC activations of the same extent; up: mi355_wino_pack_weight of the [y->c][x->c] (mode 0) weights; desc: kd 3, stride 1, pad 1,
// plain / norm-prologue input, plain un-windowed output; bias, residual, out_chscale as in mi355_conv3d_fwd.
extern "C" int mi355_conv3d_wino_fwd(const mi355_act* x, const float* up, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!x || !y || !up || !d || !x->p || !y->p) return MI355_EINVAL;
  if (d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (d->off_z || d->off_y || d->off_x || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return MI355_EUNSUPPORTED;
  if (x->d != y->d || x->h != y->h || x->w != y->w || x->n != y->n) return MI355_EINVAL;
  if (x->c % 4 || x->ld % 4 || x->ld < x->c || y->ld < y->c || ((uintptr_t)x->p & 15) || ((uintptr_t)up & 15)) return MI355_EINVAL;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift || !(d->act_slope >= 0.f && d->act_slope <= 1.f))) return MI355_EINVAL;
  if (d->residual && d->residual_ld < y->c) return MI355_EINVAL;
  WinoArgs a;
  memset(&a.g, 0, sizeof(a.g));
  if (d->moments_out && d->gn_bwd) return MI355_EUNSUPPORTED;
  a.g.mom = d->moments_out;
  if (d->gn_bwd) {
    const mi355_gn_bwd_fuse* f = d->gn_bwd;
    if (d->in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
    if (!f->gx || !f->scale || !f->shift || !f->mean_rstd || !f->partials_out || f->groups <= 0 || y->c % f->groups || f->gx_ld < y->c) return MI355_EINVAL;
    a.g.gnb = f->partials_out; a.g.gx = f->gx; a.g.gxld = f->gx_ld; a.g.gscale = f->scale; a.g.gshift = f->shift; a.g.gmr = f->mean_rstd;
    a.g.ggroups = f->groups; a.g.gslope = f->act_slope;
  }
  a.x = (const float*)x->p; a.xld = x->ld; a.up = up; a.y = (float*)y->p; a.yld = y->ld;
  a.res = d->residual; a.resld = d->residual_ld;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.out_chscale = d->out_chscale; a.bias = d->bias;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.CinP = (x->c + 7) / 8 * 8;
  a.Cout = y->c; a.CoutP = (y->c + 31) / 32 * 32;
  a.tilesZ = ceil_div(a.D, 2); a.tilesY = ceil_div(a.H, 8); a.tilesX = ceil_div(a.W, 16); a.coTiles = a.CoutP / 32;
  a.vec4 = a.Cout % 4 == 0 && a.yld % 4 == 0 && !((uintptr_t)a.y & 15) && (!a.res || (a.resld % 4 == 0 && !((uintptr_t)a.res & 15))) &&
           (!a.g.gnb || (a.g.gxld % 4 == 0 && !((uintptr_t)a.g.gx & 15)));
  const WinoZPlan zp = plan_wino_zring(a.N, a.D, a.H, a.W, a.Cout);
  if (zp.use == 1) {
    a.zsplits = zp.zsplits; a.zper = zp.zper;
    const long long zblocks = (long long)a.N * zp.zsplits * a.tilesY * a.tilesX * a.coTiles;
    if (zblocks <= 0 || zblocks > 0x7fffffffLL) return MI355_EINVAL;
    const int lds_bytes = (2 * (180 * 12 + 16) + 2 * 16 * 32 * 8 + 8 * 2 * 32 * 32 + 3 * a.CinP) * (int)sizeof(float);
    const dim3 zgrid((unsigned)zblocks), zblk(512);
#define WINO_ZLAUNCH(IM, FU)                                                                         \
    do { SET_MAX_DYN_LDS((conv3d_wino2d_zring<IM, FU>), lds_bytes);                                    \
         LAUNCH((conv3d_wino2d_zring<IM, FU>), zgrid, zblk, lds_bytes, stream, a); } while (0)
    if (a.g.mom) {
      if (d->in_mode == MI355_IN_PLAIN) WINO_ZLAUNCH(MI355_IN_PLAIN, 1); else WINO_ZLAUNCH(MI355_IN_AFFINE_ACT, 1);
    } else if (a.g.gnb) {
      WINO_ZLAUNCH(MI355_IN_PLAIN, 2);
    } else if (d->in_mode == MI355_IN_PLAIN) WINO_ZLAUNCH(MI355_IN_PLAIN, 0);
    else WINO_ZLAUNCH(MI355_IN_AFFINE_ACT, 0);
#undef WINO_ZLAUNCH
    return LAUNCH_CHECK();
  }
  a.zsplits = 1; a.zper = a.D;
  const long long blocks = (long long)a.N * a.tilesZ * a.tilesY * a.tilesX * a.coTiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
  if (zp.use == 2) {
    const int lds_bytes = (8 * 2 * 32 * 32 + 3 * a.CinP) * (int)sizeof(float);
    const dim3 wgrid((unsigned)blocks), wblk(512);
#define WINO_WLAUNCH(IM, FU)                                                                         \
    do { SET_MAX_DYN_LDS((conv3d_wino2d_w8<IM, FU>), lds_bytes);                                       \
         LAUNCH((conv3d_wino2d_w8<IM, FU>), wgrid, wblk, lds_bytes, stream, a); } while (0)
    if (a.g.mom) {
      if (d->in_mode == MI355_IN_PLAIN) WINO_WLAUNCH(MI355_IN_PLAIN, 1); else WINO_WLAUNCH(MI355_IN_AFFINE_ACT, 1);
    } else if (a.g.gnb) {
      WINO_WLAUNCH(MI355_IN_PLAIN, 2);
    } else if (d->in_mode == MI355_IN_PLAIN) WINO_WLAUNCH(MI355_IN_PLAIN, 0);
    else WINO_WLAUNCH(MI355_IN_AFFINE_ACT, 0);
#undef WINO_WLAUNCH
    return LAUNCH_CHECK();
  }
  const dim3 grid((unsigned)blocks), blk(256);
  const char* pe = getenv("MI355_WINO_PIPE");                  // A/B switches, read per call (tests flip them); default: pipelined
  const bool pipe = !(pe && pe[0] == '0');
  const char* be_ = getenv("MI355_WINO_BMODE");                // weights: 0 at their use | 1 first, transform under their latency | 2 a phase ahead
  const int bmode = be_ ? (be_[0] == '1' ? 1 : be_[0] == '0' ? 0 : 2) : 2;      // default: 2 (never measured: the ISA of 0 stalls on them)
#define WINO_LAUNCH(IM, FU)                                                                         \
  do { if (pipe && bmode == 2) LAUNCH((conv3d_wino2d<IM, FU, true, 2>), grid, blk, 0, stream, a); \
       else if (pipe && bmode == 1) LAUNCH((conv3d_wino2d<IM, FU, true, 1>), grid, blk, 0, stream, a); \
       else if (pipe) LAUNCH((conv3d_wino2d<IM, FU, true, 0>), grid, blk, 0, stream, a);           \
       else LAUNCH((conv3d_wino2d<IM, FU, false>), grid, blk, 0, stream, a); } while (0)
  if (a.g.mom) {
    if (d->in_mode == MI355_IN_PLAIN) WINO_LAUNCH(MI355_IN_PLAIN, 1); else WINO_LAUNCH(MI355_IN_AFFINE_ACT, 1);
  } else if (a.g.gnb) {
    WINO_LAUNCH(MI355_IN_PLAIN, 2);
  } else if (d->in_mode == MI355_IN_PLAIN) WINO_LAUNCH(MI355_IN_PLAIN, 0);
  else WINO_LAUNCH(MI355_IN_AFFINE_ACT, 0);
#undef WINO_LAUNCH
  return LAUNCH_CHECK();
}

// epilogue records per sample: the 2 x 8 x 16 tiles of conv3d_wino2d, or the (z range, 8 x 16 column) workgroups of conv3d_wino2d_zring
extern "C" int32_t mi355_conv3d_wino_stats_blocks(const mi355_act* y) {
  if (!y) return 0;
  const WinoZPlan zp = plan_wino_zring(y->n, y->d, y->h, y->w, y->c);
  if (zp.use == 1) return (int32_t)((long long)zp.zsplits * zp.tilesY * zp.tilesX);
  const long long b = (long long)ceil_div(y->d, 2) * ceil_div(y->h, 8) * ceil_div(y->w, 16);
  return b > 0 && b <= 0x7fffffffLL ? (int32_t)b : 0;
}
