// 3x3x3 stride-1 conv3d forward / dgrad on gfx950 bf16 MFMA (v_mfma_f32_32x32x16_bf16) with SPLIT fp32 operands.
//
// Same op and same fusions as conv3d_fwd.hip (reference: unet3d/models/pytorch/classification/resnet.py:12-22 called from
// myronenko.py:17-21; GroupNorm-apply + ReLU prologue, residual / Dropout3d / concat-slice epilogue), but the MACs run on the
// bf16 matrix pipe, which is 16x the rate of the f32 MFMA (MI355X_MICROARCH.md: 2.5 PFLOP/s vs 157 TFLOP/s):
//
//   NS = 2  (MI355_PREC_BF16X3): every fp32 operand x is split as x = hi + lo, hi = bf16(x), lo = bf16(x - hi), and
//           a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi (3 MFMAs, each product exact, fp32 accumulate). The dropped a_lo*b_lo
//           term and the residual of the split are <= ~2^-16 relative: fp32 in / fp32 out at ~1e-5 relative accuracy.
//   NS = 3  (MI355_PREC_BF16X6): three planes (hi, mid, lo = 24 mantissa bits) and the 6 products of order <= 2:
//           fp32-class accuracy (~2^-23) at 6 MFMAs per k-step.
//   NS = 1  (MI355_PREC_BF16):   operands rounded to bf16, one product -- autocast-style mixed precision.
//
// The split costs nothing per MAC: activations are split once while the haloed tile is staged into LDS (each staged element
// is then read by 27 taps x Cout), weights are split once per optimizer step by the pack kernel.
//
// Implicit GEMM, no im2col: M = 32 output voxels per MFMA tile (2 x-rows of 16), N = 32 output channels, K = 16 input
// channels of one tap per MFMA. LDS holds the haloed input tile [halo voxel][plane][KC channels] (bf16), 16*VSQ bytes per
// voxel with VSQ odd; the lane -> voxel map of an M tile sends the two ds_read_b128 lane groups ({0-3,12-15,20-27} and
// {4-11,16-19,28-31}) to the two x-rows of the tile, so every 16-lane group reads 16 distinct 16-byte slots: conflict-free.
// Weights are pre-packed [tap][ci/8][plane][co][8] (bf16) so a lane fetches its B fragment with one 16-byte global load
// (L2/L1 resident), software-prefetched one k-step ahead.
#include "conv3d_lp.h"
#include "pack_values.h"
#include "act_io.h"
#ifndef LP_GNB_BATCH
#define LP_GNB_BATCH 4
#endif

// FUSE: 0 plain epilogue, 1 + moment records of the output, 2 + norm-backward sums (dgrad): as conv3d_fwd.hip
// F16: MI355_PREC_F16 -- the single operand plane is IEEE fp16 instead of bf16 (same tile, same loop, v_mfma_f32_32x32x16_f16)
// TA: storage type of x, y, the residual and the normalised tensor of the norm-backward sums (act_io.h). bf16 storage (NS == 1 bf16
// operands only): a plain input goes from global memory to LDS without a conversion, outputs are rounded once on store, statistics are
// taken over the values as stored.
template <int TZ, int TY, int J, int NS, int WM, int WN, int MT, int NT, int INMODE, int FUSE = 0, bool F16 = false, typename TA = float>
// LDS holds 3 workgroups of the largest tile: the register allocator must fit 3 waves per SIMD too (several variants sat one or two
// registers above), except the 4-tile waves with a norm prologue or the norm-backward epilogue, which would spill.
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD((MT * NT >= 8 || (MT * NT >= 4 && (INMODE == MI355_IN_AFFINE_ACT || (FUSE == 2 && !(NT == 2 && WN == 2))))) ? 2 : 3)
void conv3d_k3_bf16(ConvBArgs a) {
  const TA* const ax = reinterpret_cast<const TA*>(a.x);
  TA* const ay = reinterpret_cast<TA*>(a.y);
  const TA* const ares = reinterpret_cast<const TA*>(a.res);
  const TA* const agx = reinterpret_cast<const TA*>(a.g.gx);
  constexpr bool RAW16 = lp_storage_is_operand<TA, F16>::value && INMODE == MI355_IN_PLAIN && NS == 1;      // staged values ARE the stored ones
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(TZ * TY / 2 == WM * MT, "M tiles (2 x-rows of 16 voxels) must equal WM*MT");
  constexpr int TX = 16;
  constexpr int HZ = TZ + 2, HY = TY + 2, HX = TX + 2;
  constexpr int HV = HZ * HY * HX;
  constexpr int KC = 16 * J;                 // input channels per LDS chunk
  constexpr int OCT = KC / 8;                // channel octets per chunk
  constexpr int VSQ = NS * OCT + 1;          // voxel stride in 16-byte units (odd)
  constexpr int P = Products<NS>::P;
  DYN_LDS(lds_f);
  uint4* lds = reinterpret_cast<uint4*>(lds_f);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, li = lane & 31;
  const int wm = wave / WN, wn = wave % WN;

  // XCD-aware tile order: consecutive workgroup ids land on different XCDs (id % 8); give each XCD a contiguous range of
  // spatial tiles so that neighbouring tiles (which share halo voxels) hit the same L2.
  int b = blockIdx.x;
  const int cot = b % a.coTiles; b /= a.coTiles;
  {
    const int nsp = a.spatialTiles, per = nsp / 8;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
  const int n = b;
  const int co_base = cot * (32 * WN * NT) + wn * (32 * NT);

  int lrow, ltx;
  mtile_lane(li, lrow, ltx);
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = wm * MT + mt;
    const int mz = m / (TY / 2), my = (m % (TY / 2)) * 2 + lrow;
    abase[mt] = ((mz * HY + my) * HX + ltx) * VSQ + half;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const int CQ8 = a.CinP / 8;
  const int so = tid % OCT, sv0 = tid / OCT;

  const unsigned lane_b = (unsigned)(half * NS * a.CoutP + co_base + li) * 16u;      // this lane's byte offset inside a weight slab
  const size_t tap_slab = (size_t)CQ8 * NS * a.CoutP;                                // uint4 per tap
  auto load_b = [&](uint4 (&bf)[J][NS][NT], const uint4* slab, int jn_) {           // slab: tap and chunk applied, wave-uniform
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int p = 0; p < NS; ++p)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          bf[j][p][nt] = make_uint4(0u, 0u, 0u, 0u);
          if (j < jn_)
            bf[j][p][nt] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(slab + ((size_t)(2 * j) * NS + p) * a.CoutP + nt * 32) + lane_b);
        }
  };

  for (int c0 = 0; c0 < a.CinP; c0 += KC) {
    const int jn = (a.CinP - c0) / 16 < J ? (a.CinP - c0) / 16 : J;   // k-steps of this chunk that exist; wave-uniform
    const uint4* wpc = a.wp + (size_t)(c0 / 8) * NS * a.CoutP;          // this chunk's channel octets inside every tap slab
    uint4 b0[J][NS][NT], b1[J][NS][NT], b2[J][NS][NT];
    // ---- stage the haloed input tile for channels [c0, c0+KC): normalise/activate, split into bf16 planes ----
    __syncthreads();
    {
      const int c = c0 + 8 * so;
      const bool v0ok = c < a.Cin, v1ok = c + 4 < a.Cin;
      float sc[8], sh[8], sl[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; sl[e] = a.slope; }
      if (INMODE == MI355_IN_AFFINE_ACT) {
#pragma unroll
        for (int hq = 0; hq < 2; ++hq) {
          if (hq ? v1ok : v0ok) {
            const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c + 4 * hq);
            const float4 h4 = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c + 4 * hq);
            sc[4 * hq] = s4.x; sc[4 * hq + 1] = s4.y; sc[4 * hq + 2] = s4.z; sc[4 * hq + 3] = s4.w;
            sh[4 * hq] = h4.x; sh[4 * hq + 1] = h4.y; sh[4 * hq + 2] = h4.z; sh[4 * hq + 3] = h4.w;
            if (a.in_slope) {
              const float4 l4 = *reinterpret_cast<const float4*>(a.in_slope + c + 4 * hq);
              sl[4 * hq] = l4.x; sl[4 * hq + 1] = l4.y; sl[4 * hq + 2] = l4.z; sl[4 * hq + 3] = l4.w;
            }
          }
        }
      }
      // Batches of UB staging units, two phases each: (1) the global loads of the batch are issued from clamped, always-valid
      // addresses -- no branch around a load, so they are in flight together instead of one dependent round trip per unit --
      // (2) mask / normalise / split / write to LDS. One batch of all 11 units held 88 registers and cost the norm-prologue
      // variants a wave of occupancy (2 instead of 3 workgroups per CU: 0.88 vs 0.63 ms on the 32-channel 128^3 layer).
      constexpr int UP = (HV * OCT + 255) / 256;       // staging units (one halo voxel x 8 channels) per thread
      constexpr int UB = UP <= 6 ? UP : (UP + 1) / 2;
      const int c0q = v0ok ? c : 0, c1q = v1ok ? c + 4 : c0q;
#pragma unroll
      for (int k0 = 0; k0 < UP; k0 += UB) {
        float4 ld0[UB], ld1[UB];
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) {
          if (k0 + kk >= UP) continue;
          int hv = sv0 + (k0 + kk) * (256 / OCT);
          if (hv >= HV) hv = HV - 1;
          const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
          int iz = tz0 - a.pad + hz, iy = ty0 - a.pad + hy, ix = tx0 - a.pad + hx;
          iz = iz < 0 ? 0 : (iz < a.Di ? iz : a.Di - 1);
          iy = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1);
          ix = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
          const TA* src = ax + ((((size_t)n * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * a.xld;
          if constexpr (RAW16) {
            const uint2 r0 = *reinterpret_cast<const uint2*>(src + c0q), r1 = *reinterpret_cast<const uint2*>(src + c1q);
            ld0[kk] = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y));      // 8 packed bf16
          } else {
            ld0[kk] = ld4(src + c0q);
            ld1[kk] = ld4(src + c1q);
          }
        }
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) {
          if (k0 + kk >= UP) continue;
          const int hv = sv0 + (k0 + kk) * (256 / OCT);
          if (hv >= HV) continue;
          const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
          const int iz = tz0 - a.pad + hz, iy = ty0 - a.pad + hy, ix = tx0 - a.pad + hx;
          const bool inb = iz >= 0 && iy >= 0 && ix >= 0 && iz < a.Di && iy < a.Hi && ix < a.Wi;
          if constexpr (RAW16) {
            const bool k0v = inb && v0ok, k1v = inb && v1ok;
            lds[hv * VSQ + so] = make_uint4(k0v ? __float_as_uint(ld0[kk].x) : 0u, k0v ? __float_as_uint(ld0[kk].y) : 0u,
                                            k1v ? __float_as_uint(ld0[kk].z) : 0u, k1v ? __float_as_uint(ld0[kk].w) : 0u);
            continue;
          }
          float v[8] = {ld0[kk].x, ld0[kk].y, ld0[kk].z, ld0[kk].w, ld1[kk].x, ld1[kk].y, ld1[kk].z, ld1[kk].w};
          if (INMODE == MI355_IN_AFFINE_ACT) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const float u = v[e] * sc[e] + sh[e];
              v[e] = fmaxf(u, u * sl[e]);          // act(u) for 0 <= slope <= 1
            }
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { if (!(inb && v0ok)) v[e] = 0.f; if (!(inb && v1ok)) v[4 + e] = 0.f; }
          uint4 pl[NS];
          split8<NS, F16>(v, pl);
#pragma unroll
          for (int p = 0; p < NS; ++p) lds[hv * VSQ + p * OCT + so] = pl[p];
        }
      }
    }
    load_b(b0, wpc, jn);              // taps 0 and 1: requested once the staging registers are free, in flight across the barrier
    load_b(b1, wpc + tap_slab, jn);
    __syncthreads();

    // ---- 27 taps x J k-steps. B fragments (weights, L1/L2 resident) are requested TWO TAPS ahead of the MFMAs that use them, in
    // three register sets that rotate through an unroll-by-three body (27 = 9 x 3): a bf16 k-step is only 32 * MT * NT matrix
    // cycles, so one step of lead (the first version) left every step waiting for its weights. Addresses = wave-uniform slab
    // pointer + one 32-bit lane offset; no "current = next" moves, no tap / k-step counters in vector registers.
    auto run_tap = [&](const uint4 (&bf)[J][NS][NT], int tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;      // wave-uniform: scalar ALU
      const int toff = ((dz * HY + dy) * HX + dx) * VSQ;
#pragma unroll
      for (int j = 0; j < J; ++j) {
        if (j >= jn) continue;
        uint4 af[MT][NS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < NS; ++p) af[mt][p] = lds[abase[mt] + toff + 2 * j + p * OCT];
#pragma unroll
        for (int q = 0; q < P; ++q)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = mfma_lp<F16>(af[mt][Products<NS>::pa[q]], bf[j][Products<NS>::pb[q]][nt], acc[mt][nt]);
      }
    };
#pragma unroll 1
    for (int tap = 0; tap < 27; tap += 3) {
      load_b(b2, wpc + tap_slab * (tap + 2), jn);
      SCHED_BARRIER();      // the requests stay above the MFMAs (the scheduler otherwise sinks them to their use)
      run_tap(b0, tap);
      if (tap + 3 < 27) load_b(b0, wpc + tap_slab * (tap + 3), jn);      // wave-uniform branches
      SCHED_BARRIER();
      run_tap(b1, tap + 1);
      if (tap + 4 < 27) load_b(b1, wpc + tap_slab * (tap + 4), jn);
      SCHED_BARRIER();
      run_tap(b2, tap + 2);
    }
  }

  // ---- epilogue, interior tiles (all of a 128^3 layer's but its ragged edge): accumulator register r of an M tile is x position r of
  // x-row ((0b0110 >> (r >> 2)) & 1) ^ half (mtile_lane, inverted), so a lane needs two row pointers per tile and wave-uniform
  // offsets r * ld -- the general path below recomputes the lane map, three bound checks and two 64-bit voxel indices per value,
  // which cost more vector-ALU time than the whole MFMA phase of a bf16 tile. ----
  const bool interior = tz0 + TZ <= a.Do && ty0 + TY <= a.Ho && tx0 + TX <= a.Wo && a.offz == 0 && a.offy == 0 && a.offx == 0 &&
                        a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo;                      // workgroup-uniform
  if (interior) {
    constexpr int K = FUSE == 1 ? 3 : 2;
    float vals[NT][K];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = co_base + nt * 32 + li;
      const bool cov = co < a.Cout;
      const int coc = cov ? co : a.Cout - 1;
      float bs = 0.f, cs = 1.f;
      if (a.bias) bs = a.bias[coc];
      if (a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + coc];
      float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
      if constexpr (FUSE == 2) {
        const int grp = coc / (a.Cout / a.g.ggroups);
        gsc = a.g.gscale[(size_t)n * a.Cout + coc]; gsh = a.g.gshift[(size_t)n * a.Cout + coc];
        gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        if constexpr (MT * NT >= 8) SCHED_BARRIER();      // one tile's reads at a time: hoisted above the previous tile they spill (265 registers)
        const int m = wm * MT + mt;
        const int mz = m / (TY / 2), my0 = (m % (TY / 2)) * 2;
        const size_t vrow = (((size_t)n * a.Do + tz0 + mz) * a.Ho + ty0 + my0) * a.Wo + tx0;      // x-row 0 of the tile, x = 0
        const size_t vA = vrow + (size_t)half * a.Wo, vB = vrow + (size_t)(half ^ 1) * a.Wo;      // this lane's two x-rows
        TA* yA = ay + vA * a.yld + coc;
        TA* yB = ay + vB * a.yld + coc;
        // the reads of the normalised tensor in batches of GB values: all 16 of a tile at once beside the 128 accumulators of an 8-tile wave
        // spill (LP_GNB_BATCH; 265 registers with 16)
        constexpr int GB = (FUSE == 2 && MT * NT >= 8) ? LP_GNB_BATCH : 16;
        const TA* gA = FUSE == 2 ? agx + vA * a.g.gxld + coc : nullptr;
        const TA* gB = FUSE == 2 ? agx + vB * a.g.gxld + coc : nullptr;
        const TA* rA = a.res ? ares + vA * a.resld + coc : nullptr;
        const TA* rB = a.res ? ares + vB * a.resld + coc : nullptr;
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += GB) {
          float gxv[GB];
          if constexpr (GB < 16) SCHED_BARRIER();
          if constexpr (FUSE == 2) {
#pragma unroll
            for (int q = 0; q < GB; ++q) gxv[q] = ld1((((0x6 >> ((r0 + q) >> 2)) & 1) ? gB : gA) + (size_t)(r0 + q) * a.g.gxld);
          }
#pragma unroll
          for (int q = 0; q < GB; ++q) {
            const int r = r0 + q;
            const bool rowb = (0x6 >> (r >> 2)) & 1;
            float v = acc[mt][nt][r] + bs;
            if (a.res) v += ld1((rowb ? rB : rA) + (size_t)r * a.resld);
            v *= cs;
            if (cov) st1((rowb ? yB : yA) + (size_t)r * a.yld, v);
            if constexpr (FUSE != 0) v = as_stored(ay, v);
            if constexpr (FUSE == 1) {
              if (mt == 0 && r == 0) K0 = v;
              const float t = v - K0;
              s0 += t; s1 += t * t;
            } else if constexpr (FUSE == 2) {
              const float xv = gxv[q];
              const float u = xv * gsc + gsh;
              const float du = u > 0.f ? v : v * a.g.gslope;
              s0 += du; s1 += du * ((xv - gmean) * grstd);
            }
          }
        }
      }
      if constexpr (FUSE == 1) {
        const float c = cov ? (float)(MT * 16) : 0.f;
        const float m2 = s1 - s0 * s0 / (float)(MT * 16);
        vals[nt][0] = c; vals[nt][1] = cov ? s0 + c * K0 : 0.f; vals[nt][2] = (cov && m2 > 0.f) ? m2 : 0.f;
      } else if constexpr (FUSE == 2) {
        vals[nt][0] = cov ? s0 : 0.f; vals[nt][1] = cov ? s1 : 0.f;
      }
    }
    if constexpr (FUSE != 0) {
      const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
      const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
      float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * K;
      gn_fuse_reduce_store<K, NT, WM, WN>(vals, lds_f, wm, wn, half, li, tid, dst, cot * (32 * WN * NT), a.Cout);
    }
    return;
  }

  // the 128-channel workgroups (WN = NT = 2) with the norm-backward sums: launched on whole tiles only (launch_b) -- with the general path
  // below in the same kernel the allocator spills 265 registers on the 8-tile waves (33 without it)
  if constexpr (NT == 2 && WN == 2 && FUSE == 2) __builtin_trap();
  // ---- epilogue: bias, residual, dropout scale, windowed store (channel-contiguous across lanes) ----
  if constexpr (FUSE == 0) {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int m = wm * MT + mt;
      const int mz = m / (TY / 2), my0 = (m % (TY / 2)) * 2;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;     // MFMA row = A-operand lane index
        int rr, rtx;
        mtile_lane(row, rr, rtx);
        const int oz = tz0 + mz, oy = ty0 + my0 + rr, ox = tx0 + rtx;
        if (oz >= a.Do || oy >= a.Ho || ox >= a.Wo) continue;
        const int sz = oz + a.offz, sy = oy + a.offy, sx = ox + a.offx;
        if (sz < 0 || sy < 0 || sx < 0 || sz >= a.yD || sy >= a.yH || sx >= a.yW) continue;
        const size_t ovox = (((size_t)n * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
        const size_t svox = (((size_t)n * a.yD + sz) * a.yH + sy) * a.yW + sx;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int co = co_base + nt * 32 + li;
          if (co >= a.Cout) continue;
          float v = acc[mt][nt][r];
          if (a.bias) v += a.bias[co];
          if (a.res) v += ld1(ares + ovox * a.resld + co);
          if (a.out_chscale) v *= a.out_chscale[(size_t)n * a.Cout + co];
          st1(ay + svox * a.yld + co, v);
        }
      }
    }
  } else {
    // the same epilogue + norm statistics of what it stores (gn_fuse.h; host side: un-windowed plain outputs only), one N tile at a time
    constexpr int K = FUSE == 1 ? 3 : 2;
    float vals[NT][K];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int co = co_base + nt * 32 + li;
      const bool cov = co < a.Cout;
      const int coc = cov ? co : a.Cout - 1;
      float bs = 0.f, cs = 1.f;
      if (a.bias) bs = a.bias[coc];
      if (a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + coc];
      float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
      int cnt = 0;
      if constexpr (FUSE == 2) {
        const int grp = coc / (a.Cout / a.g.ggroups);
        gsc = a.g.gscale[(size_t)n * a.Cout + coc]; gsh = a.g.gshift[(size_t)n * a.Cout + coc];
        gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        const int m = wm * MT + mt;
        const int mz = m / (TY / 2), my0 = (m % (TY / 2)) * 2;
        // FUSE 2: the 16 reads of the normalised tensor of this tile first, from clamped addresses (see conv3d_fwd.hip)
        float gxv[16];
        if constexpr (FUSE == 2) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
            int rr, rtx;
            mtile_lane(row, rr, rtx);
            int oz = tz0 + mz, oy = ty0 + my0 + rr, ox = tx0 + rtx;
            oz = oz < a.Do ? oz : a.Do - 1; oy = oy < a.Ho ? oy : a.Ho - 1; ox = ox < a.Wo ? ox : a.Wo - 1;
            gxv[r] = ld1(agx + ((((size_t)n * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.g.gxld + coc);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
          int rr, rtx;
          mtile_lane(row, rr, rtx);
          const int oz = tz0 + mz, oy = ty0 + my0 + rr, ox = tx0 + rtx;
          if (oz >= a.Do || oy >= a.Ho || ox >= a.Wo || !cov) continue;
          const size_t ovox = (((size_t)n * a.Do + oz) * a.Ho + oy) * a.Wo + ox;
          float v = acc[mt][nt][r] + bs;
          if (a.res) v += ld1(ares + ovox * a.resld + co);
          v *= cs;
          st1(ay + ovox * a.yld + co, v);
          v = as_stored(ay, v);
          if constexpr (FUSE == 1) {
            if (cnt == 0) K0 = v;
            const float t = v - K0;
            s0 += t; s1 += t * t;
          } else {
            const float xv = gxv[r];
            const float u = xv * gsc + gsh;
            const float du = u > 0.f ? v : v * a.g.gslope;
            s0 += du; s1 += du * ((xv - gmean) * grstd);
          }
          ++cnt;
        }
      }
      if constexpr (FUSE == 1) {
        const float c = (float)cnt;
        const float m2 = cnt > 0 ? s1 - s0 * s0 / c : 0.f;
        vals[nt][0] = c; vals[nt][1] = s0 + c * K0; vals[nt][2] = m2 > 0.f ? m2 : 0.f;
      } else {
        vals[nt][0] = s0; vals[nt][1] = s1;
      }
    }
    const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
    const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
    float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * K;
    gn_fuse_reduce_store<K, NT, WM, WN>(vals, lds_f, wm, wn, half, li, tid, dst, cot * (32 * WN * NT), a.Cout);
  }
}

// channels, 32 output channels), 2: conv3d_k3_lp_zring2 (17..64 input channels, output channels in tiles of 32; ks = channel slices).
// MI355_BF16_FORM: auto (default: zring2 on eligible shapes with enough workgroups to fill the chip), zring (zring2 on any eligible
// shape: tests), zring1 (the round-3 kernel where it applies: the A/B switch), tile (never).
// Measured: profiles/r3_bf16_zring.txt (zring1), profiles/r4_bf16_zring2.txt (zring2).
struct LpZPlan { int tilesY, tilesX, zsplits, zper, use, ks, coTiles; };
static LpZPlan plan_lp_zring(int n, int cin, int cout, int d, int h, int w, int precision, const mi355_conv_desc* desc, int act_dtype = MI355_ACT_F32) {
  LpZPlan p; memset(&p, 0, sizeof(p));
  if (!act_matches_precision(act_dtype, precision)) return p;      // 16-bit storage goes with operands of its own type
  const char* fe = getenv("MI355_BF16_FORM");
  const bool v1 = fe && !strncmp(fe, "zring1", 6);
  const char form = fe && fe[0] ? fe[0] : 'a';
  if (form != 'z' && form != 'a') return p;
  if (precision != MI355_PREC_BF16 && precision != MI355_PREC_F16) return p;
  if (cin % 4 || h % 8 || w % 16 || d < 1) return p;
  if (desc->pad != 1 || desc->off_z || desc->off_y || desc->off_x || desc->out_d != d || desc->out_h != h || desc->out_w != w) return p;
  p.tilesY = h / 8; p.tilesX = w / 16;
  const long long cols = (long long)n * p.tilesY * p.tilesX;
  const char* ze = getenv("MI355_BF16_ZSPLITS");            // tests: force the number of z ranges
  // Norm-backward sums in the epilogue (desc->gn_bwd): conv3d_k3_lp_zring2 has no such form. On 33..64 input channels the normalised
  // tensor's 16 values per lane do not fit beside 216 weight and 96 accumulator registers (140 spills); the 32-channel instantiation
  // compiled clean and ran correctly on the CPU emulator but faulted on the MI355X (HSA memory aperture violation on the loads of the
  // normalised tensor; bisected with variant builds, profiles/r4_bf16_zring2.txt) and was removed. Such calls take the round-3 kernel
  // where it applies (17..32 -> 32 channels); elsewhere the statistics query answers 0 (mi355_conv3d_bf16_stats_blocks), the sums take
  // their own pass and the conv itself runs on zring2 with the plain epilogue.
  const bool v1ok = cin > 16 && cin <= 32 && cout == 32;
  if (v1 || (desc->gn_bwd && v1ok)) {
    if (!v1ok) return p;                                     // 17..32 input channels (two k-steps per tap), 32 output channels
    if (form == 'a' && cols < 64) return p;                  // too few columns to fill the chip with whole-CU workgroups
    int zsplits = (int)((256 + cols - 1) / cols);
    if (ze && atoi(ze) > 0) zsplits = atoi(ze);
    if (zsplits > d) zsplits = d;
    if (zsplits < 1) zsplits = 1;
    p.zper = ceil_div(d, zsplits);
    p.zsplits = ceil_div(d, p.zper);
    p.use = 1; p.ks = 1; p.coTiles = 1;
    return p;
  }
  if (cin <= 16 || cin > 64 || cout % 32 || d < 4) return p;
  p.ks = cin > 32 ? 2 : 1;
  if (p.ks == 2 && desc->in_slope) return p;                 // the channel-split form takes the scalar activation slope only
  p.coTiles = cout / 32;
  const long long wg = cols * p.coTiles;
  int zsplits = (int)((256 + wg - 1) / wg);                  // one workgroup per CU: at least one wave of workgroups
  if (ze && atoi(ze) > 0) zsplits = atoi(ze);
  if (zsplits > d / 4) zsplits = d / 4;                      // every range has >= 4 planes (the head / tail steps of the kernel)
  if (zsplits < 1) zsplits = 1;
  p.zper = ceil_div(d, zsplits);
  p.zsplits = ceil_div(d, p.zper);
  if (d - (p.zsplits - 1) * p.zper < 4) {                    // a short last range: fold it into even ranges of >= 4 planes
    p.zsplits = d / 4 < p.zsplits ? d / 4 : p.zsplits - 1;
    if (p.zsplits < 1) p.zsplits = 1;
    p.zper = ceil_div(d, p.zsplits);
    p.zsplits = ceil_div(d, p.zper);
    if (d - (p.zsplits - 1) * p.zper < 4) { p.zsplits = 1; p.zper = d; }
  }
  if (form == 'a' && wg * p.zsplits < 96) return p;          // too few whole-CU workgroups to fill the chip
  p.use = 2;
  return p;
}

// ---- weight packing: OIDHW fp32 -> [tap][ciP/8][plane][coP][8] bf16 planes ------------------------------------------
// mode 0: forward pack of a Conv3d weight; mode 1: dgrad pack (taps flipped, roles of ci/co swapped); cout/cin are the
// PACKED roles as in mi355_pack_conv_weight.
__global__ void pack_weight_bf16_kernel(const float* w, unsigned short* wp, int cout, int cin, int T, int coutP, int cinP, int mode, int NS, int f16) {
  const size_t total = (size_t)T * (cinP / 8) * coutP * 8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x)
    pack_lp_item(w, wp, idx, cout, cin, T, coutP, cinP, mode, NS, f16);      // pack_values.h (shared with mi355_pack_weights_batch)
}

static int nsplit_of(int precision) {
  switch (precision) {
    case MI355_PREC_BF16X3: return 2;
    case MI355_PREC_BF16X6: return 3;
    case MI355_PREC_BF16: return 1;
    case MI355_PREC_F16: return 1;
    default: return 0;
  }
}

extern "C" size_t mi355_packed_weight_bytes_bf16(int32_t cout, int32_t cin, int32_t kd, int32_t precision) {
  const int ns = nsplit_of(precision);
  if (!ns) return 0;
  const int coutP = (cout + 31) / 32 * 32, cinP = (cin + 15) / 16 * 16;
  return (size_t)kd * kd * kd * cinP * coutP * ns * 2;
}

extern "C" int mi355_pack_conv_weight_bf16(const float* w, void* wp, int32_t cout, int32_t cin, int32_t kd, int32_t mode,
                                           int32_t precision, void* stream) {
  const int ns = nsplit_of(precision);
  if (!w || !wp || cout <= 0 || cin <= 0 || kd != 3 || mode < 0 || mode > 1 || !ns) return MI355_EINVAL;
  const int coutP = (cout + 31) / 32 * 32, cinP = (cin + 15) / 16 * 16;
  const size_t total = (size_t)27 * cinP * coutP;
  int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
  LAUNCH(pack_weight_bf16_kernel, dim3(grid), dim3(256), 0, stream, w, (unsigned short*)wp, cout, cin, 27, coutP, cinP, mode, ns,
         precision == MI355_PREC_F16 ? 1 : 0);
  return LAUNCH_CHECK();
}

// ---- dispatch -------------------------------------------------------------------------------------------------------
template <int TZ, int TY, int J, int NS, int WM, int WN, int MT, int NT, bool F16, typename TA = float>
static int launch_b(ConvBArgs& a, int in_mode, void* stream) {
  constexpr int HV = (TZ + 2) * (TY + 2) * 18;
  constexpr int VSQ = NS * 2 * J + 1;
  constexpr size_t lds = (size_t)HV * VSQ * 16;
  static_assert(lds <= 64 * 1024, "LDS tile must fit the default 64 KiB dynamic window");
  a.tilesZ = ceil_div(a.Do, TZ); a.tilesY = ceil_div(a.Ho, TY); a.tilesX = ceil_div(a.Wo, 16);
  a.coTiles = ceil_div(a.Cout, 32 * WN * NT);
  a.spatialTiles = a.N * a.tilesZ * a.tilesY * a.tilesX;
  const long long blocks = (long long)a.spatialTiles * a.coTiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
  if (a.g.mom && a.g.gnb) return MI355_EUNSUPPORTED;
  if (a.g.mom) {
    if (in_mode == MI355_IN_PLAIN)
      LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 1, F16, TA>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    else
      LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_AFFINE_ACT, 1, F16, TA>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  } else if (a.g.gnb) {
    if (in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
    if constexpr (NT == 2 && WN == 2)      // the 128-channel workgroups carry the interior epilogue only (lp_tile_cfg routes whole-tile calls here)
      if (a.Do % TZ || a.Ho % TY || a.Wo % 16 || a.offz || a.offy || a.offx || a.yD != a.Do || a.yH != a.Ho || a.yW != a.Wo) return MI355_EUNSUPPORTED;
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 2, F16, TA>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  } else if (in_mode == MI355_IN_PLAIN)
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 0, F16, TA>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  else
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_AFFINE_ACT, 0, F16, TA>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  return LAUNCH_CHECK();
}

// Tile configuration of a call. Spatial tile: 4 x 4 x 16 voxels on big volumes (>= 128 K voxels: 4 waves along M), 2 x 4 x 16 on small ones
// so the grid still fills the chip. Output channels per workgroup: 32, 64, or -- round 6, one-plane operands (NS = 1), Cout % 128 == 0 --
// 128 with twice the accumulator tiles per wave (4 x 2 on the big tile, 2 x 2 on the small one): a weight fragment fetched through the
// L1 then feeds four / two MFMAs instead of two / one. Measured at batch 4 (profiles/r6_lp_tile_wide.txt): 128 -> 128 @32^3 0.149 -> 0.120 ms,
// 256 -> 256 @16^3 0.108 -> 0.091 ms. With the norm-backward epilogue (gnb): the wide forms on whole tiles only (`whole`, lp_whole:
// their kernels carry the interior epilogue alone; with the general one beside 128 accumulators hipcc spills 265
// registers and the call ran slower, 0.136 -> 0.185 ms). The small wide form is not routed there: interior-only it fits three waves per
// SIMD (168 registers, 219 with the general epilogue) and still measured 0.111 against 0.107 ms on 256 -> 256 @16^3 (the 64-channel form).
// MI355_BF16_WIDE: 0 = never, big / small = that wide form on every eligible call whatever its size (tests), otherwise by size.
struct LpTileCfg { bool big; int nw; };      // nw: output channels per workgroup (32, 64, 128)
static int lp_whole(int d, int h, int w) { return (h % 4 == 0 && w % 16 == 0) ? (d % 4 == 0 ? 3 : (d % 2 == 0 ? 2 : 0)) : 0; }      // bit 0: whole 4 x 4 x 16 tiles, bit 1: whole 2 x 4 x 16
static LpTileCfg lp_tile_cfg(int ns, long long vox, int cout, bool gnb, int whole) {
  LpTileCfg c;
  c.big = ns < 3 && vox >= 256LL * 512;      // the 3-plane tile of the big configuration would exceed the 64 KiB LDS window
  c.nw = cout > 32 ? 64 : 32;
  const char* e = getenv("MI355_BF16_WIDE");
  if (ns == 1 && cout % 128 == 0 && !(e && e[0] == '0')) {
    const bool fb = e && e[0] == 'b', fs = e && e[0] == 's';
    if (gnb) {
      if ((whole & 1) && (fb || (!fs && c.big))) { c.big = true; c.nw = 128; }
    } else if (fb) { c.big = true; c.nw = 128; }
    else if (fs) { c.big = false; c.nw = 128; }
    else if (c.big || vox * (cout / 128) >= 128LL * 256) c.nw = 128;      // small tile: only with a workgroup per CU left
  }
  return c;
}

template <int NS, bool F16 = false, typename TA = float>
static int dispatch_ns(ConvBArgs& a, int in_mode, long long vox, void* stream) {
  constexpr int J = NS == 1 ? 2 : 1;      // (one 16-channel k-step per chunk for NS = 1 too: more workgroups per CU, measured 10 % slower)
  const LpTileCfg c = lp_tile_cfg(NS, vox, a.Cout, a.g.gnb != nullptr, lp_whole(a.Do, a.Ho, a.Wo));
  if constexpr (NS < 3) {
    if (c.big) {
      if constexpr (NS == 1)
        if (c.nw == 128) return launch_b<4, 4, J, NS, 2, 2, 4, 2, F16, TA>(a, in_mode, stream);
      if (c.nw == 64) return launch_b<4, 4, J, NS, 4, 1, 2, 2, F16, TA>(a, in_mode, stream);
      return launch_b<4, 4, J, NS, 4, 1, 2, 1, F16, TA>(a, in_mode, stream);
    }
  }
  if constexpr (NS == 1)
    if (c.nw == 128) return launch_b<2, 4, J, NS, 2, 2, 2, 2, F16, TA>(a, in_mode, stream);
  if (c.nw == 64) return launch_b<2, 4, J, NS, 2, 2, 2, 1, F16, TA>(a, in_mode, stream);
  return launch_b<2, 4, J, NS, 4, 1, 1, 1, F16, TA>(a, in_mode, stream);
}

// spatial tiles (= epilogue records per sample) of the configuration dispatch_ns picks; 0: this call cannot fuse statistics
int32_t mi355_conv3d_bf16_stats_blocks(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  const int ns = nsplit_of(d->precision);
  if (!ns || d->kd != 3 || d->stride != 1 || d->out_mode != MI355_OUT_PLAIN) return 0;
  if (d->off_z || d->off_y || d->off_x || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return 0;
  const LpZPlan zp = plan_lp_zring(x->n, x->c, y->c, y->d, y->h, y->w, d->precision, d, x->dtype);
  // plane-ring kernels: one record per (z range, column); zring2 leaves one per wave and half-wave of the column's workgroup (x 8)
  if (zp.use && x->d == y->d && x->h == y->h && x->w == y->w) {
    if (zp.use == 2 && d->gn_bwd) return 0;                  // zring2 has no norm-backward form (plan_lp_zring): unfused sums
    return (int32_t)((long long)zp.zsplits * zp.tilesY * zp.tilesX * (zp.use == 2 ? 8 : 1));
  }
  const long long vox = (long long)y->d * y->h * y->w * x->n;
  const bool big = lp_tile_cfg(ns, vox, y->c, d->gn_bwd != nullptr, lp_whole(y->d, y->h, y->w)).big;
  const int tz = big ? 4 : 2, ty = 4;
  const long long b = (long long)ceil_div(y->d, tz) * ceil_div(y->h, ty) * ceil_div(y->w, 16);
  return b > 0 && b <= 0x7fffffffLL ? (int32_t)b : 0;
}

// Trace name (as rocprofv3 prints it) of the kernel mi355_conv3d_fwd_bf16_impl launches for this call: bench.py keys its per-instantiation
// roofline rows and the PMC traffic look-up on it (mi355_conv3d_fwd_config). Mirrors the dispatch below.
int mi355_conv3d_bf16_kernel_name(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d, char* out, size_t n) {
  const int ns = nsplit_of(d->precision);
  if (!ns || !out || n < 8) return MI355_EINVAL;
  const int fuse = d->moments_out ? 1 : (d->gn_bwd ? 2 : 0);
  const char* f16 = d->precision == MI355_PREC_F16 ? "true" : "false";
  const LpZPlan zp = plan_lp_zring(x->n, x->c, y->c, d->out_d, d->out_h, d->out_w, d->precision, d, x->dtype);
  if (zp.use && y->d == d->out_d && y->h == d->out_h && y->w == d->out_w && x->d == d->out_d && x->h == d->out_h && x->w == d->out_w) {
    const char* st = x->dtype == MI355_ACT_BF16 ? "unsigned short" : (x->dtype == MI355_ACT_F16 ? "f16_t" : "float");
    if (zp.use == 1) snprintf(out, n, "conv3d_k3_lp_zring<2, %d, %d, %s, %s>", d->in_mode, fuse, f16, st);
    else snprintf(out, n, "conv3d_k3_lp_zring2<2, %d, %d, %d, %s, %s>", zp.ks, d->in_mode, fuse, f16, st);
    return 0;
  }
  const long long vox = (long long)d->out_d * d->out_h * d->out_w * x->n;
  const int J = ns == 1 ? 2 : 1;
  const LpTileCfg c = lp_tile_cfg(ns, vox, y->c, d->gn_bwd != nullptr, lp_whole(d->out_d, d->out_h, d->out_w));
  const char* tile = c.big ? (c.nw == 128 ? "4, 4, %d, %d, 2, 2, 4, 2" : c.nw == 64 ? "4, 4, %d, %d, 4, 1, 2, 2" : "4, 4, %d, %d, 4, 1, 2, 1")
                           : (c.nw == 128 ? "2, 4, %d, %d, 2, 2, 2, 2" : c.nw == 64 ? "2, 4, %d, %d, 2, 2, 2, 1" : "2, 4, %d, %d, 4, 1, 1, 1");
  char t[64];
  snprintf(t, sizeof(t), tile, J, ns);
  snprintf(out, n, "conv3d_k3_bf16<%s, %d, %d, %s, %s>", t, d->in_mode, fuse, f16, x->dtype == MI355_ACT_BF16 ? "unsigned short" : (x->dtype == MI355_ACT_F16 ? "f16_t" : "float"));
  return 0;
}

// called by mi355_conv3d_fwd (conv3d_fwd.hip) when desc->precision selects a bf16 path and the problem qualifies
int mi355_conv3d_fwd_bf16_impl(const mi355_act* x, const void* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  const int ns = nsplit_of(d->precision);
  if (!ns || d->kd != 3 || d->stride != 1) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (x->dtype != y->dtype) return MI355_EUNSUPPORTED;
  const bool lp = act_is_lp16(x->dtype);
  if (!act_matches_precision(x->dtype, d->precision)) return MI355_EUNSUPPORTED;      // 16-bit storage goes with operands of its own type
  if (lp && (((uintptr_t)y->p & 1) || ((uintptr_t)x->p & 7))) return MI355_EINVAL;
  ConvBArgs a;
  memset(&a.g, 0, sizeof(a.g));
  if (d->moments_out || d->gn_bwd) {
    if (!mi355_conv3d_bf16_stats_blocks(x, y, d)) return MI355_EUNSUPPORTED;
    a.g.mom = d->moments_out;
    if (d->gn_bwd) {
      const mi355_gn_bwd_fuse* f = d->gn_bwd;
      if (!f->gx || !f->scale || !f->shift || !f->mean_rstd || !f->partials_out || f->groups <= 0 || y->c % f->groups || f->gx_ld < y->c) return MI355_EINVAL;
      a.g.gnb = f->partials_out; a.g.gx = (const float*)f->gx; a.g.gxld = f->gx_ld; a.g.gscale = f->scale; a.g.gshift = f->shift; a.g.gmr = f->mean_rstd;
      a.g.ggroups = f->groups; a.g.gslope = f->act_slope;
    }
  }
  a.x = (const float*)x->p; a.xld = x->ld; a.wp = (const uint4*)wp; a.y = (float*)y->p; a.yld = y->ld;
  a.res = (const float*)d->residual; a.resld = d->residual_ld;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.out_chscale = d->out_chscale; a.bias = d->bias;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Cin = x->c; a.CinP = (x->c + 15) / 16 * 16;
  a.Do = d->out_d; a.Ho = d->out_h; a.Wo = d->out_w; a.Cout = y->c; a.CoutP = (y->c + 31) / 32 * 32;
  a.yD = y->d; a.yH = y->h; a.yW = y->w; a.offz = d->off_z; a.offy = d->off_y; a.offx = d->off_x;
  a.pad = d->pad;
  if (a.res && a.resld < a.Cout) return MI355_EINVAL;
  const long long vox = (long long)a.Do * a.Ho * a.Wo * a.N;
  const LpZPlan zp = plan_lp_zring(a.N, a.Cin, a.Cout, a.Do, a.Ho, a.Wo, d->precision, d, x->dtype);
  if (zp.use && a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo && a.Di == a.Do && a.Hi == a.Ho && a.Wi == a.Wo) {
    if (a.g.mom && a.g.gnb) return MI355_EUNSUPPORTED;
    if (a.g.gnb && d->in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
    a.tilesY = zp.tilesY; a.tilesX = zp.tilesX; a.zsplits = zp.zsplits; a.zper = zp.zper;
    const long long blocks = (long long)a.N * zp.zsplits * zp.tilesY * zp.tilesX;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
    const int fuse = a.g.mom ? 1 : (a.g.gnb ? 2 : 0);
    const bool f16 = d->precision == MI355_PREC_F16;
    if (zp.use == 2) {
      a.coTiles = zp.coTiles;
      const long long wgs = blocks * zp.coTiles;
      if (wgs > 0x7fffffffLL) return MI355_EINVAL;
      return mi355_lp_zring2_launch(a, zp.ks, d->in_mode, fuse, f16, lp, wgs, stream);
    }
    return mi355_lp_zring_launch(a, d->in_mode, fuse, f16, lp, blocks, stream);
  }
  if (x->dtype == MI355_ACT_BF16) return dispatch_ns<1, false, bf16_t>(a, d->in_mode, vox, stream);
  if (x->dtype == MI355_ACT_F16) return dispatch_ns<1, true, f16_t>(a, d->in_mode, vox, stream);
  if (d->precision == MI355_PREC_F16) return dispatch_ns<1, true>(a, d->in_mode, vox, stream);
  if (ns == 1) return dispatch_ns<1>(a, d->in_mode, vox, stream);
  if (ns == 2) return dispatch_ns<2>(a, d->in_mode, vox, stream);
  return dispatch_ns<3>(a, d->in_mode, vox, stream);
}
