// conv3d weight gradient on gfx950 fp32 MFMA.
//
// Replaces autograd's conv3d weight-gradient for the reference's Conv3d layers
// (unet3d/models/pytorch/classification/resnet.py:12-22):
//   dw[co][ci][t] = sum_{n,q} dy[n,q,co] * in(x)[n, stride*q + t - pad, ci]
// where in() is the same fused GroupNorm-apply + (Leaky)ReLU input transform as the forward
// (myronenko.py:18-19), recomputed from x and the saved per-(n,c) scale/shift instead of storing the
// activated tensor.
//
// GEMM view per tap: D[co][ci] (32x32 MFMA tile) += A[co][voxel] * B[voxel][ci], K = voxels (2 per MFMA). Both operands sit in LDS
// in their natural voxel-major layout, so every MFMA operand is a conflict-free ds_read_b32 (32 consecutive channels per
// half-wave), and the 27 tap accumulators are distributed over the 4 waves of a workgroup (7/7/7/6). Partial tiles go to a
// workspace slab; a second kernel reduces the slabs in a fixed order (deterministic, no atomics) and writes OIDHW.
//   conv3d_wgrad_ring : 3x3x3 stride 1 (the dominant kernel of the training step): 4x8 columns marched along z through an LDS ring
//                       of input planes, global loads overlapped with the MFMA loop (below).
//   conv3d_wgrad_mfma : 3x3x3 stride 2 and 1x1x1 (incl. the depth-to-space transposed-conv form): split-K over voxel tiles staged
//                       whole in LDS, register-prefetch software pipeline.
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

struct WgradArgs {
  const float* x; int xld;
  const float* dy; int dyld;
  float* ws;
  const float* in_scale; const float* in_shift; float slope;
  int N, Di, Hi, Wi, Cin;
  int Do, Ho, Wo, Cout;
  int pad;
  int tilesZ, tilesY, tilesX, ntiles;  // output-voxel tiles (per whole batch: ntiles = N*tilesZ*tilesY*tilesX)
  int splits, ciTiles, coTiles;
  const float* in_slope;   // per-input-channel negative slope (NULL: scalar slope)
  int dymode;              // MI355_OUT_D2S: dy is the fine tensor of a ConvTranspose3d(k2,s2); logical Cout = 8*fC (1x1x1 only)
  int cD, cH, cW, fC;
};

// TA: storage type of x and dy (act_io.h); the struct carries them as float pointers whatever the type
template <int KD, int STRIDE, int TZ, int TY, int TX, int INMODE, typename TA = float>
__global__ __launch_bounds__(256) void conv3d_wgrad_mfma(WgradArgs a) {
  const TA* const ax = reinterpret_cast<const TA*>(a.x);
  const TA* const ady = reinterpret_cast<const TA*>(a.dy);
  constexpr int T = KD * KD * KD;
  constexpr int TV = TZ * TY * TX;
  constexpr int HZ = (TZ - 1) * STRIDE + KD, HY = (TY - 1) * STRIDE + KD, HX = (TX - 1) * STRIDE + KD;
  constexpr int HV = HZ * HY * HX;
  constexpr int WV = (T == 1) ? 4 : 1;          // waves split voxels (1x1x1) or taps (3x3x3)
  constexpr int NTW = (T == 1) ? 1 : (T + 3) / 4;  // taps per wave
  static_assert(TX % 2 == 0 && TV % (2 * WV) == 0, "voxel pairs");
  DYN_LDS(lds);
  float* lds_dy = lds;              // [TV][32]
  float* lds_x = lds + TV * 32;     // [HV][32]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int split = blockIdx.x, cit = blockIdx.y, cot = blockIdx.z;
  const int ci0 = cit * 32, co0 = cot * 32;

  int toff[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    int tap = (T == 1) ? 0 : wave + 4 * ti;
    if (tap >= T) tap = T - 1;
    const int dz = tap / (KD * KD), dyy = (tap / KD) % KD, dx = tap % KD;
    toff[ti] = ((dz * HY + dyy) * HX + dx) * 32;
  }
  f32x16 acc[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

  const int per = (a.ntiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < a.ntiles ? t_begin + per : a.ntiles;
  const int sq = tid & 7, sv0 = tid >> 3;   // staging: 8 quads per 32-channel row
  constexpr int UPD = (TV + 31) / 32;       // dy staging units (one voxel x 4 channels) per thread and tile
  constexpr int UPX = (HV + 31) / 32;       // x  staging units per thread and tile
  const int cdy = co0 + 4 * sq, cx = ci0 + 4 * sq;
  const bool dyvalid = cdy < a.Cout, xvalid = cx < a.Cin;   // Cout % 4 == 0 and Cin % 4 == 0 are required
  const bool d2s = KD == 1 && a.dymode == MI355_OUT_D2S;
  const int d2s_p = (d2s && dyvalid) ? cdy / a.fC : 0, d2s_k = cdy - d2s_p * a.fC;

  // Software pipeline over the tiles of this workgroup: the global loads of tile t+1 are issued (into registers, from
  // clamped always-valid addresses, all in flight together) BEFORE the MFMA loop of tile t and written to LDS after it,
  // so HBM/L2 latency overlaps the matrix work instead of alternating with it.
  float4 pdy[UPD], px[UPX];
  auto decode = [&](int tile, int& n, int& tz0, int& ty0, int& tx0) {
    int b = tile;
    tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
    ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
    tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
    n = b;
  };
  auto prefetch = [&](int tile) {
    int n, tz0, ty0, tx0;
    decode(tile, n, tz0, ty0, tx0);
#pragma unroll
    for (int k = 0; k < UPD; ++k) {
      int v = sv0 + k * 32; if (v >= TV) v = TV - 1;
      if (d2s) {
        int ox = tx0 + v; if (ox >= a.Wo) ox = a.Wo - 1;
        const int xx = ox % a.cW, yy = (ox / a.cW) % a.cH, zz = ox / (a.cW * a.cH);
        const size_t fv = (((size_t)n * (2 * a.cD) + 2 * zz + (d2s_p >> 2)) * (2 * a.cH) + 2 * yy + ((d2s_p >> 1) & 1)) * (2 * a.cW) + 2 * xx + (d2s_p & 1);
        pdy[k] = ld4(ady + fv * a.dyld + (dyvalid ? d2s_k : 0));
      } else {
        int oz = tz0 + v / (TY * TX), oy = ty0 + (v / TX) % TY, ox = tx0 + v % TX;
        oz = oz < a.Do ? oz : a.Do - 1; oy = oy < a.Ho ? oy : a.Ho - 1; ox = ox < a.Wo ? ox : a.Wo - 1;
        pdy[k] = ld4(ady + ((((size_t)n * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.dyld + (dyvalid ? cdy : 0));
      }
    }
#pragma unroll
    for (int k = 0; k < UPX; ++k) {
      int hv = sv0 + k * 32; if (hv >= HV) hv = HV - 1;
      const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
      int iz = tz0 * STRIDE - a.pad + hz, iy = ty0 * STRIDE - a.pad + hy, ix = tx0 * STRIDE - a.pad + hx;
      iz = iz < 0 ? 0 : (iz < a.Di ? iz : a.Di - 1);
      iy = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1);
      ix = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
      px[k] = ld4(ax + ((((size_t)n * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * a.xld + (xvalid ? cx : 0));
    }
  };
  // mask / transform the prefetched registers of `tile` and write them to LDS
  auto commit = [&](int tile) {
    int n, tz0, ty0, tx0;
    decode(tile, n, tz0, ty0, tx0);
#pragma unroll
    for (int k = 0; k < UPD; ++k) {
      const int v = sv0 + k * 32;
      if (v >= TV) continue;
      bool ok;
      if (d2s) ok = dyvalid && tx0 + v < a.Wo;
      else {
        const int oz = tz0 + v / (TY * TX), oy = ty0 + (v / TX) % TY, ox = tx0 + v % TX;
        ok = dyvalid && oz < a.Do && oy < a.Ho && ox < a.Wo;
      }
      *reinterpret_cast<float4*>(lds_dy + v * 32 + 4 * sq) = ok ? pdy[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
    if (INMODE == MI355_IN_AFFINE_ACT && xvalid) {
      sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + cx);
      sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + cx);
      if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope + cx);
    }
#pragma unroll
    for (int k = 0; k < UPX; ++k) {
      const int hv = sv0 + k * 32;
      if (hv >= HV) continue;
      const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
      const int iz = tz0 * STRIDE - a.pad + hz, iy = ty0 * STRIDE - a.pad + hy, ix = tx0 * STRIDE - a.pad + hx;
      const bool ok = xvalid && iz >= 0 && iy >= 0 && ix >= 0 && iz < a.Di && iy < a.Hi && ix < a.Wi;
      float4 v = px[k];
      if (INMODE == MI355_IN_AFFINE_ACT) {
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        // act(u) = max(u, slope * u) for 0 <= slope <= 1 (ReLU 0, LeakyReLU 0.01, identity 1): 2 packed ops per pair, no compare/select
        v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
      }
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(lds_x + hv * 32 + 4 * sq) = v;
    }
  };

  if (t_begin < t_end) prefetch(t_begin);
  for (int tile = t_begin; tile < t_end; ++tile) {
    __syncthreads();                 // every wave is done reading the previous tile
    commit(tile);
    __syncthreads();
    if (tile + 1 < t_end) prefetch(tile + 1);
    SCHED_BARRIER();                 // keep the prefetch loads above the MFMA loop
    // ---- K loop over voxel pairs ----
    constexpr int KSTEPS = TV / 2 / WV;
    const int ks0 = (T == 1) ? wave * KSTEPS : 0;
#pragma unroll 4
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int v = 2 * (ks0 + ks) + half;
      const int tz = v / (TY * TX), ty = (v / TX) % TY, tx = v % TX;
      const float av = lds_dy[v * 32 + li];
      const int xb = (((tz * STRIDE) * HY + ty * STRIDE) * HX + tx * STRIDE) * 32 + li;
#pragma unroll
      for (int ti = 0; ti < NTW; ++ti) {
        const float bv = lds_x[xb + toff[ti]];
        acc[ti] = MFMA_32x32x2(av, bv, acc[ti]);
      }
    }
  }

  // ---- write the partial tiles: ws[pair][slab][tap][32 co][32 ci] ----
  const int SL = a.splits * WV;
  const size_t pair = (size_t)cot * a.ciTiles + cit;
  const int slab = split * WV + ((T == 1) ? wave : 0);
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    const int tap = (T == 1) ? 0 : wave + 4 * ti;
    if (tap >= T) continue;
    float* dst = a.ws + (((pair * SL + slab) * T + tap) * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;   // co within tile
      dst[row * 32 + li] = acc[ti][r];
    }
  }
}

// ---- 3x3x3 stride-1 wgrad, plane-ring form (the dominant kernel of the training step) ----
// A workgroup owns a TY(y) x 8(x) column of output voxels for one (32 co x 32 ci) pair and marches along z. LDS holds a ring of
// 4 haloed input planes ((TY+2) x 10 voxels x 32 ci) and 2 dy planes (TY*8 voxels x 32 co): while the 4 waves run the TY*4
// voxel-pair k-steps x 7 taps of output plane z out of ring slots z-1, z, z+1, the global loads of input plane z+2 and dy plane
// z+1 are in flight (issued before the MFMA loop); they are normalised / activated and written to the free ring slot after it --
// one barrier per plane, and the HBM/L2 latency never sits between two MFMA loops. Compared with the tile form above (stage
// 4x4x8 tile -> barrier -> MFMAs -> barrier) the halo overhead drops from 2.8 to 1.9 (TY = 4) / 1.56 (TY = 8) input voxels per
// output voxel. Operand layout and MFMA mapping are the same: conflict-free ds_read_b32 of 32 consecutive channels,
// A = dy[voxel][co], B = in(x)[voxel + tap][ci], K = voxel pairs.
// Round-2 A/B on MI355X of three prepared variants: all lost and are gone.
//   * TY = 8 (8x8 columns, 67.6 KB of LDS, half the barriers per output voxel): +1.9 % in the isolated micro-benchmark
//     (32->32 @128^3: 1.843 -> 1.809 ms) but -8 % INSIDE the training step (rocprofv3 trace of bench.py: the 15 launches of the
//     128^3 .. 32^3 layers 23.2 -> 25.1 ms per step) -- the step is what counts; the TY parameter stays in the template.
//   * v_mfma_f32_16x16x4_f32 tiles, 27 per wave (tap-balanced): -3 % (twice the LDS operand reads per flop).
//   * a slab reduction with 16 loads in flight per thread: -1 .. -6 %.
template <int INMODE, int TY>
__global__ __launch_bounds__(256) void conv3d_wgrad_ring(WgradArgs a) {
  constexpr int TX = 8, HY = TY + 2, HX = 10, PV = TY * TX, HPV = HY * HX;
  constexpr int XSLOT = HPV * 32, DSLOT = PV * 32;
  constexpr int NXU = (HPV * 8 + 255) / 256, NDU = (PV * 8) / 256;     // staging units (voxel, channel quad) per thread and plane
  static_assert((PV * 8) % 256 == 0, "dy plane staging");
  constexpr int NTW = 7;
  DYN_LDS(lds);
  float* lds_x = lds;                    // ring of 4 haloed input planes [HPV][32]
  float* lds_dy = lds + 4 * XSLOT;       // 2 dy planes [PV][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int split = blockIdx.x, cit = blockIdx.y, cot = blockIdx.z;
  const int ci0 = cit * 32, co0 = cot * 32;
  int tin[NTW], tdz[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    int tap = wave + 4 * ti;
    if (tap >= 27) tap = 26;
    tdz[ti] = tap / 9;
    tin[ti] = (((tap / 3) % 3) * HX + tap % 3) * 32 + li;
  }
  f32x16 acc[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

  const int sq = tid & 7, sv0 = tid >> 3;          // staging unit u: voxel sv0 + 32 * u of the plane, channel quad sq
  const int cdy = co0 + 4 * sq, cx = ci0 + 4 * sq;
  const bool dyvalid = cdy < a.Cout, xvalid = cx < a.Cin;
  // Chunks (column x z range) split, split + splits, ... of the a.ntiles chunks belong to this workgroup: neighbouring
  // columns are walked by neighbouring workgroups at the same time (their halos meet in L2) and all of them accumulate into
  // the same 27 tap tiles, so the number of partial slabs is the number of workgroups, not of columns.
  for (int chunk = split; chunk < a.ntiles; chunk += a.splits) {
    int b = chunk;
    const int zc = b % a.tilesZ; b /= a.tilesZ;
    const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
    const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
    const int n = b;
    const int zb = zc * a.pad, ze = zb + a.pad < a.Do ? zb + a.pad : a.Do;      // a.pad = planes per chunk (ring plan)
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
    if (INMODE == MI355_IN_AFFINE_ACT && xvalid) {
      sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + cx);
      sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + cx);
      if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope + cx);
    }
    // in-plane geometry of this thread's staging units (fixed for the whole column)
    bool xin[NXU], xuse[NXU]; size_t xo[NXU]; int xl[NXU];
#pragma unroll
    for (int u = 0; u < NXU; ++u) {
      const int hv = sv0 + 32 * u;
      xuse[u] = hv < HPV;
      const int hvc = xuse[u] ? hv : HPV - 1;
      const int hy = hvc / HX, hx = hvc % HX;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      xin[u] = xvalid && xuse[u] && iy >= 0 && ix >= 0 && iy < a.Hi && ix < a.Wi;
      const int cy = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), cxx = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
      xo[u] = ((size_t)cy * a.Wi + cxx) * a.xld + (xvalid ? cx : 0);
      xl[u] = hvc * 32 + 4 * sq;
    }
    bool din[NDU]; size_t dyo[NDU]; int dl[NDU];
#pragma unroll
    for (int u = 0; u < NDU; ++u) {
      const int v = sv0 + 32 * u;
      const int oy = ty0 + v / TX, ox = tx0 + v % TX;
      din[u] = dyvalid && oy < a.Ho && ox < a.Wo;
      dyo[u] = ((size_t)(oy < a.Ho ? oy : a.Ho - 1) * a.Wo + (ox < a.Wo ? ox : a.Wo - 1)) * a.dyld + (dyvalid ? cdy : 0);
      dl[u] = v * 32 + 4 * sq;
    }
    const float* xn = a.x + (size_t)n * a.Di * a.Hi * a.Wi * a.xld;
    const float* dyn = a.dy + (size_t)n * a.Do * a.Ho * a.Wo * a.dyld;
    const size_t xplane = (size_t)a.Hi * a.Wi * a.xld, dyplane = (size_t)a.Ho * a.Wo * a.dyld;

    float4 px[NXU], pdy[NDU];
    auto load_x = [&](int iz) {                    // input plane iz (may lie outside the volume: clamped address, masked at commit)
      const int cz = iz < 0 ? 0 : (iz < a.Di ? iz : a.Di - 1);
#pragma unroll
      for (int u = 0; u < NXU; ++u) px[u] = *reinterpret_cast<const float4*>(xn + cz * xplane + xo[u]);
    };
    auto load_dy = [&](int oz) {
      const int cz = oz < a.Do ? oz : a.Do - 1;
#pragma unroll
      for (int u = 0; u < NDU; ++u) pdy[u] = *reinterpret_cast<const float4*>(dyn + cz * dyplane + dyo[u]);
    };
    auto prologue = [&](float4 v) {
      if (INMODE == MI355_IN_AFFINE_ACT) {
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        // act(u) = max(u, slope * u) for 0 <= slope <= 1 (ReLU 0, LeakyReLU 0.01, identity 1): 2 packed ops per pair, no compare/select
        v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
      }
      return v;
    };
    auto commit_x = [&](int iz) {
      const bool zin = iz >= 0 && iz < a.Di;
      float* slot = lds_x + ((iz + 1) & 3) * XSLOT;
#pragma unroll
      for (int u = 0; u < NXU; ++u)
        if (xuse[u]) *reinterpret_cast<float4*>(slot + xl[u]) = (zin && xin[u]) ? prologue(px[u]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto commit_dy = [&](int oz) {
#pragma unroll
      for (int u = 0; u < NDU; ++u)
        *reinterpret_cast<float4*>(lds_dy + (oz & 1) * DSLOT + dl[u]) = (din[u] && oz < a.Do) ? pdy[u] : make_float4(0.f, 0.f, 0.f, 0.f);
    };

    // fill: input planes zb-1, zb, zb+1 and dy plane zb (the previous chunk ended with a barrier: every wave is done with the ring)
    for (int iz = zb - 1; iz <= zb + 1; ++iz) { load_x(iz); commit_x(iz); }
    load_dy(zb); commit_dy(zb);
    __syncthreads();

    for (int z = zb; z < ze; ++z) {
      const bool more = z + 1 < ze;
      if (more) { load_x(z + 2); load_dy(z + 1); }
      SCHED_BARRIER();                 // the loads stay above the MFMA loop they overlap with
      // Operand addresses = one per-lane base per tap (ring slot, tap offset, channel li, and the lane's voxel parity `half`:
      // voxel v = 2 ks + half sits in the same row as 2 ks, so half contributes a constant 32 floats) + a COMPILE-TIME offset per
      // k-step: with the k loop fully unrolled every ds_read carries its offset as an immediate and the loop has no address
      // arithmetic at all. (SQ counters, profiles/r2_sq_counters_conv_kernels.txt: MFMA-busy % + 4 x VALU instructions % ~ 94 % of the
      // SIMD cycles in every conv kernel -- a VALU instruction costs its 4 issue cycles of matrix time, so they are counted.)
      const float* xp[NTW];
#pragma unroll
      for (int ti = 0; ti < NTW; ++ti) xp[ti] = lds_x + ((z + tdz[ti]) & 3) * XSLOT + tin[ti] + half * 32;
      const float* dys = lds_dy + (z & 1) * DSLOT + li + half * 32;
#pragma unroll
      for (int ks = 0; ks < PV / 2; ++ks) {
        const int v2 = 2 * ks;
        const float av = dys[v2 * 32];
        const int xb = ((v2 / TX) * HX + v2 % TX) * 32;
#pragma unroll
        for (int ti = 0; ti < NTW; ++ti) acc[ti] = MFMA_32x32x2(av, xp[ti][xb], acc[ti]);
      }
      SCHED_BARRIER();
      if (more) { commit_x(z + 2); commit_dy(z + 1); }
      __syncthreads();
    }
  }

  // ---- partial tiles: ws[pair][slab = split][tap][32 co][32 ci] ----
  const size_t pair = (size_t)cot * a.ciTiles + cit;
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    const int tap = wave + 4 * ti;
    if (tap >= 27) continue;
    float* dst = a.ws + (((pair * a.splits + split) * 27 + tap) * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      dst[row * 32 + li] = acc[ti][r];
    }
  }
}

// Deterministic slab reduction. One workgroup per (pair, tap) 32x32 tile: its 256 threads split the SL slabs 4 ways (float4
// = 4 ci per thread, 64 float4 per slab), every thread keeps 8 independent running sums so 8 loads are in flight, and the 4
// slab-groups are combined through LDS in a fixed order. (A thread that walks all slabs with one dependent load at a time is
// latency-bound: 5 % of the training step at 512 slabs.)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, int parts) {
  __shared__ float4 part[4][256];
  // parts = 4: four workgroups share one tile (64 float4 each) -- launches with few (pair, tap) tiles are otherwise latency-bound
  const int tile = blockIdx.x / parts, sub0 = (blockIdx.x % parts) * (4 / parts), sub1 = sub0 + 4 / parts;
  const int tap = tile % T;
  const int pair = tile / T;
  const int cot = pair / ciTiles, cit = pair % ciTiles;
  const int tid = threadIdx.x;
  const int e4 = tid & 63 /* which float4 of ... */, grp = tid >> 6;
  // 1024 floats per slab tile = 256 float4: thread (grp, e4) handles float4 indices e4, e4+64, e4+128, e4+192 of slabs k = grp, grp+4, ...
  const float4* base = reinterpret_cast<const float4*>(ws + (((size_t)pair * SL) * T + tap) * 1024);
  const size_t slab_stride = (size_t)T * 256;       // in float4
  for (int sub = sub0; sub < sub1; ++sub) {
    const int f4 = e4 + 64 * sub;
    float4 acc[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = grp;
    for (; k + 28 < SL; k += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float4 v = base[(size_t)(k + 4 * u) * slab_stride + f4];
        acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
      }
    }
    for (int u = 0; k < SL; k += 4, ++u) {
      const float4 v = base[(size_t)k * slab_stride + f4];
      acc[u & 7].x += v.x; acc[u & 7].y += v.y; acc[u & 7].z += v.z; acc[u & 7].w += v.w;
    }
    float4 s = acc[0];
#pragma unroll
    for (int u = 1; u < 8; ++u) { s.x += acc[u].x; s.y += acc[u].y; s.z += acc[u].z; s.w += acc[u].w; }
    part[grp][f4] = s;
  }
  __syncthreads();
  // final: thread t combines the 4 groups for float4 index t (row = co in tile, 8 float4 per row of 32 ci)
  if ((tid >> 6) >= sub0 && (tid >> 6) < sub1) {
    const int f4 = tid;
    float4 s = part[0][f4];
#pragma unroll
    for (int g = 1; g < 4; ++g) { s.x += part[g][f4].x; s.y += part[g][f4].y; s.z += part[g][f4].z; s.w += part[g][f4].w; }
    const int co = cot * 32 + f4 / 8, ci = cit * 32 + (f4 % 8) * 4;
    if (co < Cout) {
      const float v[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ci + e < Cin) dw[((size_t)co * Cin + ci + e) * T + tap] = v[e];
    }
  }
}

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream) {
  const int coTiles = ceil_div(Cout, 32);
  const long long tiles = (long long)coTiles * ciTiles * T;
  const int parts = tiles < 256 ? 4 : 1;
  LAUNCH(wgrad_reduce_kernel, dim3((unsigned)(tiles * parts)), dim3(256), 0, stream, ws, dw, Cout, Cin, T, SL, ciTiles, parts);
  return LAUNCH_CHECK();
}

size_t mi355_conv3d_wgrad_bf16_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
int mi355_conv3d_wgrad_bf16_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                                 void* ws, size_t ws_bytes, void* stream);
int mi355_conv3d_c4_ok(const mi355_act* x, const mi355_conv_desc* d);
size_t mi355_conv3d_c4_wgrad_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
int mi355_conv3d_c4_wgrad_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                               void* ws, size_t ws_bytes, void* stream);
// conv3d_s2.hip: the z-marching 32 -> 32 channel stride-2 weight gradient
int mi355_conv3d_s2c32_wgrad_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
size_t mi355_conv3d_s2c32_wgrad_workspace(const mi355_act* dy);
int mi355_conv3d_s2c32_wgrad_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream);
// conv3d_wgrad_lp.hip: the z-marching 3x3x3 weight gradient of 16-bit tensors (LDS transpose reads)
int mi355_conv3d_wgrad_lp_tr_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
size_t mi355_conv3d_wgrad_lp_tr_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
int mi355_conv3d_wgrad_lp_tr_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream);
int mi355_conv3d_wgrad_k1_lp_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
size_t mi355_conv3d_wgrad_k1_lp_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d);
int mi355_conv3d_wgrad_k1_lp_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream);
static int wgrad_uses_bf16(const mi355_conv_desc* d) {
  return d->precision != MI355_PREC_F32 && d->kd == 3 && d->stride == 1 && d->pad == 1 && d->out_mode == MI355_OUT_PLAIN &&
         (d->in_mode == MI355_IN_PLAIN || d->in_mode == MI355_IN_AFFINE_ACT);
}

struct WgradPlan { int tz, ty, tx, ntiles, tilesZ, tilesY, tilesX, splits, ciTiles, coTiles, wv, coutL; size_t ws_bytes; int ok; };

static WgradPlan plan_wgrad(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WgradPlan p; memset(&p, 0, sizeof(p));
  if (!x || !dy || !d) return p;
  if ((d->kd != 1 && d->kd != 3) || (d->stride != 1 && d->stride != 2)) return p;
  if (d->kd == 1 && d->stride != 1) return p;
  int Do = dy->d, Ho = dy->h, Wo = dy->w;
  int coutL = dy->c;   // logical output channels
  if (d->out_mode == MI355_OUT_D2S) {
    if (d->kd != 1 || dy->d != 2 * x->d || dy->h != 2 * x->h || dy->w != 2 * x->w) return p;
    Do = x->d; Ho = x->h; Wo = x->w; coutL = 8 * dy->c;
  }
  if (d->kd == 1) { p.tz = 1; p.ty = 1; p.tx = 256; long long v = (long long)Do * Ho * Wo; if (v > 0x7fffffffLL) return p; Do = 1; Ho = 1; Wo = (int)v; p.wv = 4; }
  else if (d->stride == 1) return p;   // 3x3x3 stride 1: plane-ring kernel (plan_wgrad_ring)
  else { p.tz = 2; p.ty = 2; p.tx = 8; p.wv = 1; }
  p.tilesZ = ceil_div(Do, p.tz); p.tilesY = ceil_div(Ho, p.ty); p.tilesX = ceil_div(Wo, p.tx);
  const long long nt = (long long)dy->n * p.tilesZ * p.tilesY * p.tilesX;
  p.coutL = coutL;
  if (nt <= 0 || nt > 0x7fffffffLL) return p;
  p.ntiles = (int)nt;
  p.ciTiles = ceil_div(x->c, 32); p.coTiles = ceil_div(coutL, 32);
  const int pairs = p.ciTiles * p.coTiles;
  int splits = ceil_div(1024, pairs);
  const int max_splits = p.ntiles >= 8 ? p.ntiles / 8 : 1;   // >= 8 tiles per workgroup amortise the slab write
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  // make every split non-empty
  const int per = ceil_div(p.ntiles, splits);
  splits = ceil_div(p.ntiles, per);
  p.splits = splits;
  const int T = d->kd * d->kd * d->kd;
  p.ws_bytes = (size_t)pairs * splits * p.wv * T * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}

// plane-ring plan (3x3x3 stride 1): columns of 4x8 output voxels, cut into z chunks of >= 16 planes only when a (co, ci) pair
// would otherwise have fewer than 2 chunks per workgroup; 512 workgroups in total (2 per CU = what the register file holds:
// one round, no tail), i.e. 512 / pairs workgroups and partial slabs per pair.
struct RingPlan { int ty, tilesY, tilesX, zchunks, planes, chunks, splits, ciTiles, coTiles; size_t ws_bytes; int ok; };
static RingPlan plan_wgrad_ring(const mi355_act* x, const mi355_act* dy) {
  RingPlan p; memset(&p, 0, sizeof(p));
  if (x->d != dy->d || x->h != dy->h || x->w != dy->w) return p;
  p.ty = 4;                              // 4x8 columns (8x8 lost inside the step, see conv3d_wgrad_ring)
  p.tilesY = ceil_div(dy->h, p.ty); p.tilesX = ceil_div(dy->w, 8);
  p.ciTiles = ceil_div(x->c, 32); p.coTiles = ceil_div(dy->c, 32);
  const long long cols = (long long)dy->n * p.tilesY * p.tilesX;
  const long long pairs = (long long)p.ciTiles * p.coTiles;
  if (cols <= 0 || cols > 0x3fffffffLL || pairs > 0xffff) return p;
  long long S = 512 / pairs; if (S < 1) S = 1;
  int zc = 1;
  while (cols * zc < 2 * S && dy->d / (2 * zc) >= 16) zc *= 2;
  p.planes = ceil_div(dy->d, zc);
  p.zchunks = ceil_div(dy->d, p.planes);
  const long long chunks = cols * p.zchunks;
  if (chunks > 0x7fffffffLL) return p;
  if (S > chunks) S = chunks;
  p.chunks = (int)chunks; p.splits = (int)S;
  p.ws_bytes = (size_t)pairs * p.splits * 27 * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}
static int wgrad_uses_ring(const mi355_conv_desc* d) { return d->kd == 3 && d->stride == 1 && d->pad == 1 && d->out_mode == MI355_OUT_PLAIN; }

extern "C" size_t mi355_conv3d_wgrad_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  if (mi355_conv3d_c4_ok(x, d)) return mi355_conv3d_c4_wgrad_workspace(x, dy, d);
  if (mi355_conv3d_wgrad_lp_tr_ok(x, dy, d)) return mi355_conv3d_wgrad_lp_tr_workspace(x, dy, d);
  if (mi355_conv3d_wgrad_k1_lp_ok(x, dy, d)) return mi355_conv3d_wgrad_k1_lp_workspace(x, dy, d);
  if (d && wgrad_uses_bf16(d)) return mi355_conv3d_wgrad_bf16_workspace(x, dy, d);
  if (x && dy && d && wgrad_uses_ring(d)) { RingPlan r = plan_wgrad_ring(x, dy); return r.ok ? r.ws_bytes : 0; }
  if (mi355_conv3d_s2c32_wgrad_ok(x, dy, d)) return mi355_conv3d_s2c32_wgrad_workspace(dy);
  WgradPlan p = plan_wgrad(x, dy, d);
  return p.ok ? p.ws_bytes : 0;
}

template <int KD, int STRIDE, int TZ, int TY, int TX, typename TA>
static int launch_wgrad(WgradArgs& a, int in_mode, void* stream) {
  constexpr int HZ = (TZ - 1) * STRIDE + KD, HY = (TY - 1) * STRIDE + KD, HX = (TX - 1) * STRIDE + KD;
  constexpr size_t lds = (size_t)(TZ * TY * TX + HZ * HY * HX) * 32 * sizeof(float);
  static_assert(lds <= 64 * 1024, "LDS tile must fit the default 64 KiB dynamic window");
  dim3 grid(a.splits, a.ciTiles, a.coTiles);
  if (in_mode == MI355_IN_PLAIN)
    LAUNCH((conv3d_wgrad_mfma<KD, STRIDE, TZ, TY, TX, MI355_IN_PLAIN, TA>), grid, dim3(256), lds, stream, a);
  else
    LAUNCH((conv3d_wgrad_mfma<KD, STRIDE, TZ, TY, TX, MI355_IN_AFFINE_ACT, TA>), grid, dim3(256), lds, stream, a);
  return LAUNCH_CHECK();
}

extern "C" int mi355_conv3d_wgrad(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                                  void* ws, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw || !d || !ws || !x->p || !dy->p) return MI355_EINVAL;
  if (x->c % 4 || x->ld % 4 || dy->c % 4 || dy->ld % 4 || x->n != dy->n || !act_dtype_ok(x) || !act_dtype_ok(dy)) return MI355_EINVAL;
  if (((uintptr_t)x->p & act_align_mask(x->dtype)) || ((uintptr_t)dy->p & act_align_mask(dy->dtype))) return MI355_EINVAL;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift)) return MI355_EINVAL;
  if (d->in_mode == MI355_IN_AFFINE_ACT && !(d->act_slope >= 0.f && d->act_slope <= 1.f)) return MI355_EINVAL;
  if (d->precision < MI355_PREC_F32 || d->precision > MI355_PREC_F16) return MI355_EINVAL;
  if (mi355_conv3d_c4_ok(x, d)) return mi355_conv3d_c4_wgrad_impl(x, dy, dw, d, ws, ws_bytes, stream);
  if (mi355_conv3d_wgrad_lp_tr_ok(x, dy, d)) return mi355_conv3d_wgrad_lp_tr_impl(x, dy, dw, d, ws, ws_bytes, stream);
  if (mi355_conv3d_wgrad_k1_lp_ok(x, dy, d)) return mi355_conv3d_wgrad_k1_lp_impl(x, dy, dw, d, ws, ws_bytes, stream);
  if (wgrad_uses_bf16(d)) return mi355_conv3d_wgrad_bf16_impl(x, dy, dw, d, ws, ws_bytes, stream);
  if (x->dtype != dy->dtype) return MI355_EUNSUPPORTED;       // (the first-layer kernel above takes fp32 x with either dy)
  if (mi355_conv3d_s2c32_wgrad_ok(x, dy, d)) return mi355_conv3d_s2c32_wgrad_impl(x, dy, dw, d, ws, ws_bytes, stream);
  if (wgrad_uses_ring(d)) {
    if (x->dtype != MI355_ACT_F32) return MI355_EUNSUPPORTED;   // exact-fp32 3x3x3 arithmetic on 16-bit tensors: no kernel (and no caller)
    RingPlan r = plan_wgrad_ring(x, dy);
    if (!r.ok) return MI355_EUNSUPPORTED;
    if (ws_bytes < r.ws_bytes) return MI355_EWORKSPACE;
    WgradArgs a; memset(&a, 0, sizeof(a));
    a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
    a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
    a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Cin = x->c;
    a.Do = dy->d; a.Ho = dy->h; a.Wo = dy->w; a.Cout = dy->c;
    a.tilesZ = r.zchunks; a.pad = r.planes;          // ring kernel: z chunks per column / planes per chunk
    a.tilesY = r.tilesY; a.tilesX = r.tilesX; a.splits = r.splits; a.ntiles = r.chunks; a.ciTiles = r.ciTiles; a.coTiles = r.coTiles;
    dim3 grid(r.splits, r.ciTiles, r.coTiles);
#define MI355_LAUNCH_RING(IM, TYV)                                                                              \
    do {                                                                                                        \
      constexpr size_t ldsb = (size_t)(4 * ((TYV) + 2) * 10 * 32 + 2 * (TYV) * 8 * 32) * sizeof(float);         \
      SET_MAX_DYN_LDS((conv3d_wgrad_ring<IM, TYV>), ldsb);                                                      \
      LAUNCH((conv3d_wgrad_ring<IM, TYV>), grid, dim3(256), ldsb, stream, a);                                   \
    } while (0)
    if (d->in_mode == MI355_IN_PLAIN) MI355_LAUNCH_RING(MI355_IN_PLAIN, 4); else MI355_LAUNCH_RING(MI355_IN_AFFINE_ACT, 4);
#undef MI355_LAUNCH_RING
    int rc = LAUNCH_CHECK(); if (rc) return rc;
    return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, 27, r.splits, r.ciTiles, stream);
  }
  WgradPlan p = plan_wgrad(x, dy, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < p.ws_bytes) return MI355_EWORKSPACE;
  WgradArgs a;
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Cin = x->c;
  a.Do = dy->d; a.Ho = dy->h; a.Wo = dy->w; a.Cout = dy->c; a.pad = d->pad;
  a.tilesZ = p.tilesZ; a.tilesY = p.tilesY; a.tilesX = p.tilesX; a.ntiles = p.ntiles;
  a.splits = p.splits; a.ciTiles = p.ciTiles; a.coTiles = p.coTiles;
  a.in_slope = d->in_slope; a.dymode = d->out_mode; a.cD = x->d; a.cH = x->h; a.cW = x->w; a.fC = dy->c;
  a.Cout = p.coutL;
  int rc;
  if (d->kd == 1) {
    const long long vi = (long long)x->d * x->h * x->w;
    const long long vo = d->out_mode == MI355_OUT_D2S ? vi : (long long)dy->d * dy->h * dy->w;
    if (vi != vo) return MI355_EINVAL;
    a.Di = a.Hi = 1; a.Wi = (int)vi; a.Do = a.Ho = 1; a.Wo = (int)vo; a.pad = 0;
    ACT_TYPED(x->dtype, TA, rc = (launch_wgrad<1, 1, 1, 1, 256, TA>(a, d->in_mode, stream)));
  } else if (d->stride == 2) {
    ACT_TYPED(x->dtype, TA, rc = (launch_wgrad<3, 2, 2, 2, 8, TA>(a, d->in_mode, stream)));
  } else {
    return MI355_EUNSUPPORTED;       // 3x3x3 stride 1 with pad != 1 (pad 1 runs on conv3d_wgrad_ring above)
  }
  if (rc) return rc;
  const int T = d->kd * d->kd * d->kd;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, T, p.splits * p.wv, p.ciTiles, stream);
}
