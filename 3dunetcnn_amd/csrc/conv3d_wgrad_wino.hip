// 3x3x3 stride-1 conv3d WEIGHT GRADIENT with fewer multiplications: Winograd F(3x3, 2x2) in the (y, x) plane, direct along z, as a
// z-marching plane ring (round 3). Replaces autograd's weight gradient of the reference's conv3x3x3 layers
// (unet3d/models/pytorch/classification/resnet.py:12-17, called from myronenko.py:17-21; the GroupNorm-apply + ReLU prologue on x is
// recomputed from the saved per-(n, c) scale / shift exactly as in conv3d_wgrad.hip).
//
//   dw[co][ci][dz][a][b] = sum_{n, z, tiles} sum_{i,j<2} dy[z][2ty+i][2tx+j][co] * in(x)[z+dz-1][2ty+i+a-1][2tx+j+b-1][ci]
//                        = G^T [ sum_{n, z, tiles} (A h A^T) (.) (B^T d B) ] G            per (co, ci, dz)
// h: 2x2 dy tile, d: the 4x4 input window of the same tile, B^T rows d0-d2, d1+d2, d2-d1, d1-d3, A = [[1,0],[1,1],[1,-1],[0,-1]],
// G^T = [[1,1/2,1/2,0],[0,1/2,-1/2,0],[0,1/2,1/2,1]]: 16 multiplications per tile and (ci, co, dz) instead of 36 (2.25x fewer MFMAs).
//
// Why a new kernel (the first Winograd weight gradient, conv3d_wino.hip: conv3d_wino2d_wgrad, measured 2x SLOWER than the direct ring
// kernel, profiles/r3_winograd_landing.txt): there a workgroup owned ONE dz, so every transformed value (both operands are transformed
// per tile: 4x the raw data through LDS) fed 2 MFMAs per wave and the input plane was staged and transformed three times. Here a workgroup
// owns all three dz of a (32 co x 32 ci) pair and marches a 4 (y) x 8 (x) voxel column (8 Winograd tiles = K of 4 MFMAs) along z:
//   * every plane of x and dy is staged and transformed ONCE into a ring of 3 transformed planes each (V = B^T d B, Dv = A h A^T,
//     [16 points][8 tiles][32 channels]: the MFMA operand is a conflict-free ds_read_b32 of 32 consecutive channels);
//   * step p multiplies (Dv[p-1], V[p]) -> dz 2, (Dv[p], V[p]) -> dz 1, (Dv[p], V[p-1]) -> dz 0: 48 accumulator tiles (16 points x 3 dz)
//     over 8 waves = 6 per wave (96 accumulator registers), 24 MFMAs per wave and step from 32 operand fragments read once;
//   * software pipeline, ONE barrier per plane: the global loads of plane p+2 are in flight under the MFMAs of plane p, plane p+1 is
//     transformed from the staging buffer in the same phase, plane p+2 is written to the other staging buffer after the MFMAs.
// LDS: 2 x 3 x 16 KB rings + 2 x (6x10 + 4x8) voxels x 128 B staging = 119 KB -> one 512-thread workgroup per CU, 2 waves per SIMD.
// Partial 27-tap tiles go to the slab workspace of conv3d_wgrad.hip ([pair][slab][tap][32 co][32 ci]) and are reduced by its
// deterministic second pass.
#include "gfx950_dialect.h"
#include <type_traits>
#include "../../include/mi355_unet3d.h"

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream);

struct WWRArgs {
  const float* x; int xld;
  const float* dy; int dyld;
  float* ws;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  int N, D, H, W, Cin, Cout;
  int tilesY, tilesX, ncols;             // columns: index = (n * tilesY + ty) * tilesX + tx
  int splits, ciTiles, coTiles;
};

template <int INMODE>
__global__ __launch_bounds__(512) void conv3d_wgrad_wino_ring(WWRArgs a) {
  constexpr int TY = 4, TX = 8, HY = TY + 2, HX = TX + 2, HV = HY * HX, PV = TY * TX;      // plane tile, haloed input plane tile
  constexpr int NT = 8;                                     // Winograd tiles per plane tile (2 x 4 of 2x2 outputs) = K of 4 MFMAs
  constexpr int RS = 16 * NT * 32;                          // floats per transformed plane: [point][tile][channel]
  constexpr int XS = HV * 32, DS = PV * 32;                 // staged planes [voxel][channel]
  DYN_LDS(lds);
  float* Vr = lds;                                          // ring of 3 transformed input planes
  float* Dr = lds + 3 * RS;                                 // ring of 3 transformed dy planes
  float* xst = lds + 6 * RS;                                // 2 staged input planes
  float* dst_ = xst + 2 * XS;                               // 2 staged dy planes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  const int split = blockIdx.x, pair = blockIdx.y;
  const int cot = pair / a.ciTiles, cit = pair % a.ciTiles;
  const int ci0 = cit * 32, co0 = cot * 32;

  f32x16 acc[3][2];                                         // [dz][point q of this wave: p = 2 * wave + q]
#pragma unroll
  for (int dz = 0; dz < 3; ++dz)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dz][q][r] = 0.f;

  // staging units: (voxel, channel quad). x: 60 x 8 = 480 units -> thread tid < 480; dy: 32 x 8 = 256 units -> thread tid < 256
  const int sq = tid & 7, sv = tid >> 3;
  const int cx = ci0 + 4 * sq, cdy = co0 + 4 * sq;
  const bool xvalid = cx < a.Cin, dyvalid = cdy < a.Cout, xunit = tid < HV * 8, dunit = tid < PV * 8;
  // transform units: thread = (channel tc, tile tt, row half th): rows i = 2 th, 2 th + 1 of the 4 x 4 point grid
  const int tc = tid & 31, tt = (tid >> 5) & 7, th = WAVE_UNIFORM(tid >> 8);
  const int tty = tt >> 2, ttx = tt & 3;
  const size_t xplane = (size_t)a.H * a.W * a.xld, dyplane = (size_t)a.H * a.W * a.dyld;

  for (int col = split; col < a.ncols; col += a.splits) {
    int b = col;
    const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
    const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
    const int n = b;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
    if (INMODE == MI355_IN_AFFINE_ACT && xvalid) {
      sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + cx);
      sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + cx);
      if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope + cx);
    }
    // in-plane geometry of this thread's staging units (fixed for the column): clamped always-valid addresses, masks applied at commit
    const int hvc = xunit ? sv : HV - 1;
    const int iy = ty0 - 1 + hvc / HX, ix = tx0 - 1 + hvc % HX;
    const bool xin = xvalid && xunit && iy >= 0 && ix >= 0 && iy < a.H && ix < a.W;
    const int cy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1), cxx = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
    const float* xsrc = a.x + (size_t)n * a.D * xplane + ((size_t)cy * a.W + cxx) * a.xld + (xvalid ? cx : 0);
    const int dvc = dunit ? sv : PV - 1;
    const int oy = ty0 + dvc / TX, ox = tx0 + dvc % TX;
    const bool din = dyvalid && dunit && oy < a.H && ox < a.W;
    const float* dsrc = a.dy + (size_t)n * a.D * dyplane + ((size_t)(oy < a.H ? oy : a.H - 1) * a.W + (ox < a.W ? ox : a.W - 1)) * a.dyld +
                        (dyvalid ? cdy : 0);

    float4 px, pd;
    auto load_plane = [&](int p) {                          // 0 <= p < D
      px = *reinterpret_cast<const float4*>(xsrc + (size_t)p * xplane);
      pd = *reinterpret_cast<const float4*>(dsrc + (size_t)p * dyplane);
    };
    // PAR = p & 1 (staging buffer), SLOT = p % 3 (ring slot): compile-time in the unrolled plane loop below, so every LDS address of a
    // step is a per-lane base fixed for the column + an immediate (the first version computed p % 3 at run time: 39 address
    // instructions and 20 selects per step, SQ counters profiles/r3_sq_counters_wino.txt)
    auto commit_plane = [&](auto parc) {
      constexpr int PAR = decltype(parc)::value;
      if (xunit) {
        float4 v = px;
        if (INMODE == MI355_IN_AFFINE_ACT) {
          v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
          v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
        }
        if (!xin) v = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xst + PAR * XS + sv * 32 + 4 * sq) = v;
      }
      if (dunit) *reinterpret_cast<float4*>(dst_ + PAR * DS + sv * 32 + 4 * sq) = din ? pd : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    // Transform of a plane from staging buffer PAR into ring slot SLOT: this thread's two point rows of V and of Dv. `th` (waves 0-3: 0,
    // waves 4-7: 1) selects the rows, and it does so WITHOUT a branch -- the three window rows th, th + 1, th + 2 are read through a base
    // address that contains th, the one sign that differs is a scalar, and the two result rows go to scalar-selected ring rows -- so that
    // a plane step is ONE basic block and the scheduler directives below can spread the transform between the MFMAs:
    //   th = 0 (window rows d0 d1 d2):  u = d0 - d2 -> point row 0,   o = d1 + d2 -> point row 1
    //   th = 1 (window rows d1 d2 d3):  u = d1 - d3 -> point row 3,   o = d2 - d1 -> point row 2          (r0 r1 r2 = the rows read)
    //   u = r0 - r2,  o = r1 + sg * (th ? r0 : r2),  sg = th ? -1 : +1       (x * (+-1) + y is exact: the same values as the branchy form)
    // Dv = A h A^T, rows (h0., h0. + h1., h0. - h1., -h1.), columns (r0, r0 + r1, r0 - r1, -r1):
    //   X = (h00 + sg h10, h01 + sg h11) -> point row th ? 2 : 1,   Y = th ? (-h10, -h11) : (h00, h01) -> point row th ? 3 : 0
    struct TIn { float d[3][4]; float h00, h01, h10, h11; };
    const float sg = th ? -1.f : 1.f;
    const int urow = th ? 3 : 0, orow = th ? 2 : 1;            // V: point rows of u and o; Dv: rows of Y and X
    auto transform_reads = [&](TIn& t, auto parc) {
      constexpr int PAR = decltype(parc)::value;
      const float* xb = xst + PAR * XS + ((2 * tty + th) * HX + 2 * ttx) * 32 + tc;
      const float* db = dst_ + PAR * DS + ((2 * tty) * TX + 2 * ttx) * 32 + tc;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 4; ++s) t.d[r][s] = xb[(r * HX + s) * 32];
      t.h00 = db[0]; t.h01 = db[32]; t.h10 = db[TX * 32]; t.h11 = db[(TX + 1) * 32];
    };
    auto transform_math = [&](const TIn& t, auto slotc) {
      constexpr int SLOT = decltype(slotc)::value;
      float* vu = Vr + SLOT * RS + (urow * 4 * NT + tt) * 32 + tc;               // point 4 i + j at ((4 i + j) * NT + tt) * 32 + tc
      float* vo = Vr + SLOT * RS + (orow * 4 * NT + tt) * 32 + tc;
      float* dY = Dr + SLOT * RS + (urow * 4 * NT + tt) * 32 + tc;
      float* dX = Dr + SLOT * RS + (orow * 4 * NT + tt) * 32 + tc;
      float u[4], o[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        u[s] = t.d[0][s] - t.d[2][s];
        o[s] = fmaf(th ? t.d[0][s] : t.d[2][s], sg, t.d[1][s]);
      }
      vu[(0 * NT) * 32] = u[0] - u[2]; vu[(1 * NT) * 32] = u[1] + u[2]; vu[(2 * NT) * 32] = u[2] - u[1]; vu[(3 * NT) * 32] = u[1] - u[3];
      vo[(0 * NT) * 32] = o[0] - o[2]; vo[(1 * NT) * 32] = o[1] + o[2]; vo[(2 * NT) * 32] = o[2] - o[1]; vo[(3 * NT) * 32] = o[1] - o[3];
      const float xa = fmaf(t.h10, sg, t.h00), xb_ = fmaf(t.h11, sg, t.h01);
      const float ya = th ? -t.h10 : t.h00, yb = th ? -t.h11 : t.h01;
      dX[(0 * NT) * 32] = xa; dX[(1 * NT) * 32] = xa + xb_; dX[(2 * NT) * 32] = xa - xb_; dX[(3 * NT) * 32] = -xb_;
      dY[(0 * NT) * 32] = ya; dY[(1 * NT) * 32] = ya + yb; dY[(2 * NT) * 32] = ya - yb; dY[(3 * NT) * 32] = -yb;
    };
    auto transform = [&](auto parc, auto slotc) {             // the whole transform (prologue of a column)
      TIn t;
      transform_reads(t, parc);
      transform_math(t, slotc);
    };
    // The matrix work of plane p and the transform of plane p + 1 as ONE basic block, interleaved by the scheduler directives at its end:
    // with one 512-thread workgroup per CU all eight waves run the same phase, so a transform that FOLLOWS the MFMAs (the first form of
    // this kernel: 24 MFMAs, then 16 LDS reads, a wait, 24 adds, 16 LDS writes) leaves the matrix pipe idle for its whole duration. Here
    // the transform's reads go out under the first MFMAs and its arithmetic and writes ride in the shadow of the rest (a 32x32x2 fp32
    // MFMA occupies the pipe for 64 cycles; the wave issues a few other instructions behind each for free). The last plane of a column
    // transforms a stale staging buffer into a ring slot nobody reads (no branch: a branch would split the block).
    auto body = [&](auto p6c) {
      constexpr int P6 = decltype(p6c)::value, SP = P6 % 3, SM = (P6 + 2) % 3;       // ring slots of planes p and p - 1
      TIn t;
      // operand fragments of this wave's two points: tile (K) index 2 ks + half, channel li -- read once, used by three MFMAs each
      float fvp[NT / 2][2], fvm[NT / 2][2], fdp[NT / 2][2], fdm[NT / 2][2];
      const float* fb = lds + ((2 * wave) * NT + half) * 32 + li;
      auto frags = [&](int ks) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          fdm[ks][q] = fb[(3 + SM) * RS + q * NT * 32 + ks * 64]; fvp[ks][q] = fb[SP * RS + q * NT * 32 + ks * 64];
          fdp[ks][q] = fb[(3 + SP) * RS + q * NT * 32 + ks * 64]; fvm[ks][q] = fb[SM * RS + q * NT * 32 + ks * 64];
        }
      };
      frags(0);
      transform_reads(t, std::integral_constant<int, (P6 + 1) & 1>());
      frags(1); frags(2); frags(3);
#pragma unroll
      for (int ks = 0; ks < NT / 2; ++ks)
#pragma unroll
        for (int q = 0; q < 2; ++q) {                        // six independent accumulators in flight
          acc[2][q] = MFMA_32x32x2(fdm[ks][q], fvp[ks][q], acc[2][q]);     // dy plane p - 1, input plane p     : dz = 2
          acc[1][q] = MFMA_32x32x2(fdp[ks][q], fvp[ks][q], acc[1][q]);     // dy plane p,     input plane p     : dz = 1
          acc[0][q] = MFMA_32x32x2(fdp[ks][q], fvm[ks][q], acc[0][q]);     // dy plane p,     input plane p - 1 : dz = 0
        }
      transform_math(t, std::integral_constant<int, (P6 + 1) % 3>());
#ifndef MI355_EMU
      // Issue order (LDS reads in source order: k-group 0, the transform's 16, k-groups 1..3): 8 reads, then per MFMA i of the 24:
      // reads 4 (i < 6: the transform's inputs and k-group 1), 2 (i = 6..9, 12..15: k-groups 2, 3, one group ahead of their use);
      // from i = 6 two vector-ALU instructions, from i = 8 one LDS write
      __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        else if ((i >= 6 && i < 10) || (i >= 12 && i < 16)) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        if (i >= 6) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        if (i >= 8) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
      }
#endif
    };
    // one plane step; P6 = p % 6 fixes the staging parity and the ring slots at compile time
    auto step = [&](int p, auto p6c) {
      constexpr int P6 = decltype(p6c)::value;
      const bool has2 = p + 2 < a.D;                         // workgroup-uniform
      if (has2) load_plane(p + 2);
      SCHED_BARRIER();                                       // the loads stay above the MFMAs they overlap with
      body(p6c);
      SCHED_BARRIER();
      if (has2) commit_plane(std::integral_constant<int, P6 & 1>());        // plane p + 2 has the parity of p
      __syncthreads();
    };

    // ---- prologue of the column: plane 0 staged and transformed, plane 1 staged, ring slot of plane -1 zeroed ----
    load_plane(0);
    commit_plane(std::integral_constant<int, 0>());
    if (a.D > 1) load_plane(1);
    {
      float4* z4 = reinterpret_cast<float4*>(Vr + 2 * RS);  // slot (-1) % 3 == 2 of both rings
      float4* y4 = reinterpret_cast<float4*>(Dr + 2 * RS);
      for (int i = tid; i < RS / 4; i += 512) { z4[i] = make_float4(0.f, 0.f, 0.f, 0.f); y4[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    }
    __syncthreads();
    transform(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
    if (a.D > 1) commit_plane(std::integral_constant<int, 1>());
    __syncthreads();

    for (int p = 0; p < a.D; p += 6) {
      step(p, std::integral_constant<int, 0>());
      if (p + 1 >= a.D) break;
      step(p + 1, std::integral_constant<int, 1>());
      if (p + 2 >= a.D) break;
      step(p + 2, std::integral_constant<int, 2>());
      if (p + 3 >= a.D) break;
      step(p + 3, std::integral_constant<int, 3>());
      if (p + 4 >= a.D) break;
      step(p + 4, std::integral_constant<int, 4>());
      if (p + 5 >= a.D) break;
      step(p + 5, std::integral_constant<int, 5>());
    }
  }

  // ---- output transform per dz: taps[a][b] = sum_{i,j} G^T[a][i] G^T[b][j] M[4 i + j], through LDS; slab write ----
  float* Ms = lds;                                           // [16 points][32 co][32 ci] = 64 KB (the rings are done)
  const float GT[3][4] = {{1.f, 0.5f, 0.5f, 0.f}, {0.f, 0.5f, -0.5f, 0.f}, {0.f, 0.5f, 0.5f, 1.f}};
#pragma unroll
  for (int dz = 0; dz < 3; ++dz) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;   // co
        Ms[((2 * wave + q) * 32 + row) * 32 + li] = acc[dz][q][r];
      }
    __syncthreads();
    float* slab = a.ws + (((size_t)pair * a.splits + split) * 27 + dz * 9) * 1024;
    for (int idx = tid; idx < 9 * 1024; idx += 512) {
      const int tap = idx >> 10, e = idx & 1023;
      const int ta = tap / 3, tb = tap % 3;
      float o = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) o += GT[ta][i] * GT[tb][j] * Ms[(4 * i + j) * 1024 + e];
      slab[(size_t)tap * 1024 + e] = o;
    }
  }
}

constexpr int WWR_LDS_BYTES = (6 * 16 * 8 * 32 + 2 * 60 * 32 + 2 * 32 * 32) * 4;      // 121856

struct WWRPlan { int tilesY, tilesX, ncols, splits, ciTiles, coTiles; size_t ws_bytes; int ok; };
static WWRPlan plan_wwr(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WWRPlan p; memset(&p, 0, sizeof(p));
  if (!x || !dy || !d || d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return p;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return p;
  if (x->d != dy->d || x->h != dy->h || x->w != dy->w || x->n != dy->n) return p;
  if (x->dtype != MI355_ACT_F32 || dy->dtype != MI355_ACT_F32) return p;      // the fp32 path
  if (x->c % 4 || x->ld % 4 || dy->c % 4 || dy->ld % 4 || ((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & 15)) return p;
  p.tilesY = ceil_div(dy->h, 4); p.tilesX = ceil_div(dy->w, 8);
  const long long nc = (long long)dy->n * p.tilesY * p.tilesX;
  if (nc <= 0 || nc > 0x7fffffffLL) return p;
  p.ncols = (int)nc;
  p.ciTiles = ceil_div(x->c, 32); p.coTiles = ceil_div(dy->c, 32);
  const int pairs = p.ciTiles * p.coTiles;
  int splits = ceil_div(256, pairs);                        // one 512-thread workgroup per CU: ~256 workgroups, each several columns
  if (splits > p.ncols) splits = p.ncols;
  const int per = ceil_div(p.ncols, splits);                // whole columns per workgroup, no tail wave
  p.splits = ceil_div(p.ncols, per);
  p.ws_bytes = (size_t)pairs * p.splits * 27 * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}

extern "C" size_t mi355_conv3d_wgrad_wino_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  const WWRPlan p = plan_wwr(x, dy, d);
  return p.ok ? p.ws_bytes : 0;
}

// dw: OIDHW [dy->c][x->c][3][3][3]; same contract as mi355_conv3d_wgrad for kd 3 / stride 1 / pad 1 (norm prologue on x honoured).
// Returns MI355_EUNSUPPORTED for shapes it does not take (channel counts that are not multiples of 4): the caller falls back to
// mi355_conv3d_wgrad.
extern "C" int mi355_conv3d_wgrad_wino(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (!x || !dy || !dw || !d || !ws || !x->p || !dy->p) return MI355_EINVAL;
  const WWRPlan p = plan_wwr(x, dy, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < p.ws_bytes) return MI355_EWORKSPACE;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift || !(d->act_slope >= 0.f && d->act_slope <= 1.f))) return MI355_EINVAL;
  WWRArgs a;
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = dy->c;
  a.tilesY = p.tilesY; a.tilesX = p.tilesX; a.ncols = p.ncols; a.splits = p.splits; a.ciTiles = p.ciTiles; a.coTiles = p.coTiles;
  const dim3 grid(p.splits, p.ciTiles * p.coTiles);
  if (d->in_mode == MI355_IN_PLAIN) {
    SET_MAX_DYN_LDS(conv3d_wgrad_wino_ring<MI355_IN_PLAIN>, WWR_LDS_BYTES);
    LAUNCH((conv3d_wgrad_wino_ring<MI355_IN_PLAIN>), grid, dim3(512), WWR_LDS_BYTES, stream, a);
  } else {
    SET_MAX_DYN_LDS(conv3d_wgrad_wino_ring<MI355_IN_AFFINE_ACT>, WWR_LDS_BYTES);
    LAUNCH((conv3d_wgrad_wino_ring<MI355_IN_AFFINE_ACT>), grid, dim3(512), WWR_LDS_BYTES, stream, a);
  }
  const int rc = LAUNCH_CHECK(); if (rc) return rc;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, 27, p.splits, p.ciTiles, stream);
}
