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
#include "hipcompat.h"
#include <type_traits>
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"

struct ConvBArgs {
  const float* x; int xld;
  const uint4* wp;
  float* y; int yld;
  const float* res; int resld;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* out_chscale; const float* bias;
  int N, Di, Hi, Wi, Cin, CinP;       // CinP = roundup(Cin, 16)
  int Do, Ho, Wo, Cout, CoutP;
  int yD, yH, yW, offz, offy, offx;
  int pad;
  int tilesZ, tilesY, tilesX, coTiles, spatialTiles;
  int zsplits, zper;         // conv3d_k3_lp_zring: z ranges [zs * zper, min(Do, (zs + 1) * zper)) per workgroup
  GnFuseArgs g;            // norm statistics fused into the epilogue (gn_fuse.h)
};

// lane (0..31) of an M tile -> (x-row 0/1, x position 0..15): row = which ds_read_b128 lane group the lane belongs to
__device__ __forceinline__ void mtile_lane(int li, int& row, int& tx) {
  const int q = li >> 2;                    // quad index 0..7: quads {0,3,5,6} are group 0, {1,2,4,7} group 1
  const int g1 = (0x96 >> q) & 1;           // 0b10010110
  row = g1;
  const int rank = g1 ? ((q == 1) ? 0 : (q == 2) ? 1 : (q == 4) ? 2 : 3) : ((q == 0) ? 0 : (q == 3) ? 1 : (q == 5) ? 2 : 3);
  tx = rank * 4 + (li & 3);
}

template <int NS> struct Products;
template <> struct Products<1> { static constexpr int P = 1; static constexpr int pa[1] = {0}; static constexpr int pb[1] = {0}; };
template <> struct Products<2> { static constexpr int P = 3; static constexpr int pa[3] = {1, 0, 0}; static constexpr int pb[3] = {0, 1, 0}; };
// smallest terms first
template <> struct Products<3> { static constexpr int P = 6; static constexpr int pa[6] = {2, 1, 0, 1, 0, 0}; static constexpr int pb[6] = {0, 1, 2, 0, 1, 0}; };

// split 8 floats into NS bf16 planes, each plane one uint4 (8 packed bf16)
template <int NS, bool F16 = false>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[NS]) {
  static_assert(!F16 || NS == 1, "fp16 operands are not split");
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e];
#pragma unroll
  for (int p = 0; p < NS; ++p) {
    unsigned w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      w[e] = pack_lp2<F16>(r[2 * e], r[2 * e + 1]);
      if (p + 1 < NS) { r[2 * e] -= bf16lo_to_f32(w[e]); r[2 * e + 1] -= bf16hi_to_f32(w[e]); }
    }
    out[p] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

// FUSE: 0 plain epilogue, 1 + moment records of the output, 2 + norm-backward sums (dgrad): as conv3d_fwd.hip
// F16: MI355_PREC_F16 -- the single operand plane is IEEE fp16 instead of bf16 (same tile, same loop, v_mfma_f32_32x32x16_f16)
template <int TZ, int TY, int J, int NS, int WM, int WN, int MT, int NT, int INMODE, int FUSE = 0, bool F16 = false>
// LDS holds 3 workgroups of the largest tile: the register allocator must fit 3 waves per SIMD too (several variants sat one or two
// registers above), except the 4-tile waves with a norm prologue or the norm-backward epilogue, which would spill.
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD((MT * NT >= 4 && (INMODE == MI355_IN_AFFINE_ACT || FUSE == 2)) ? 2 : 3)
void conv3d_k3_bf16(ConvBArgs a) {
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
          const float* src = a.x + ((((size_t)n * a.Di + iz) * a.Hi + iy) * a.Wi + ix) * a.xld;
          ld0[kk] = *reinterpret_cast<const float4*>(src + c0q);
          ld1[kk] = *reinterpret_cast<const float4*>(src + c1q);
        }
#pragma unroll
        for (int kk = 0; kk < UB; ++kk) {
          if (k0 + kk >= UP) continue;
          const int hv = sv0 + (k0 + kk) * (256 / OCT);
          if (hv >= HV) continue;
          const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
          const int iz = tz0 - a.pad + hz, iy = ty0 - a.pad + hy, ix = tx0 - a.pad + hx;
          const bool inb = iz >= 0 && iy >= 0 && ix >= 0 && iz < a.Di && iy < a.Hi && ix < a.Wi;
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
        const int m = wm * MT + mt;
        const int mz = m / (TY / 2), my0 = (m % (TY / 2)) * 2;
        const size_t vrow = (((size_t)n * a.Do + tz0 + mz) * a.Ho + ty0 + my0) * a.Wo + tx0;      // x-row 0 of the tile, x = 0
        const size_t vA = vrow + (size_t)half * a.Wo, vB = vrow + (size_t)(half ^ 1) * a.Wo;      // this lane's two x-rows
        float* yA = a.y + vA * a.yld + coc;
        float* yB = a.y + vB * a.yld + coc;
        float gxv[16];
        if constexpr (FUSE == 2) {
          const float* gA = a.g.gx + vA * a.g.gxld + coc;
          const float* gB = a.g.gx + vB * a.g.gxld + coc;
#pragma unroll
          for (int r = 0; r < 16; ++r) gxv[r] = (((0x6 >> (r >> 2)) & 1) ? gB : gA)[(size_t)r * a.g.gxld];
        }
        const float* rA = a.res ? a.res + vA * a.resld + coc : nullptr;
        const float* rB = a.res ? a.res + vB * a.resld + coc : nullptr;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool rowb = (0x6 >> (r >> 2)) & 1;
          float v = acc[mt][nt][r] + bs;
          if (a.res) v += (rowb ? rB : rA)[(size_t)r * a.resld];
          v *= cs;
          if (cov) (rowb ? yB : yA)[(size_t)r * a.yld] = v;
          if constexpr (FUSE == 1) {
            if (mt == 0 && r == 0) K0 = v;
            const float t = v - K0;
            s0 += t; s1 += t * t;
          } else if constexpr (FUSE == 2) {
            const float xv = gxv[r];
            const float u = xv * gsc + gsh;
            const float du = u > 0.f ? v : v * a.g.gslope;
            s0 += du; s1 += du * ((xv - gmean) * grstd);
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
          if (a.res) v += a.res[ovox * a.resld + co];
          if (a.out_chscale) v *= a.out_chscale[(size_t)n * a.Cout + co];
          a.y[svox * a.yld + co] = v;
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
            gxv[r] = a.g.gx[((((size_t)n * a.Do + oz) * a.Ho + oy) * a.Wo + ox) * a.g.gxld + coc];
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
          if (a.res) v += a.res[ovox * a.resld + co];
          v *= cs;
          a.y[ovox * a.yld + co] = v;
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

// =====================================================================================================================================
// Plane-ring (z-marching), WEIGHTS-STATIONARY form for the 16-bit single-product modes (MI355_PREC_BF16 / MI355_PREC_F16) with <= 32
// input and exactly 32 output channels -- the 32 -> 32 layers of the 128^3 level are the largest group of launches of BASELINE configs[2].
// Written at the end of round 3; default for the shapes it takes since its first measurement (plan_lp_zring below).
// conv3d_k3_bf16 above runs at 20 % matrix-pipe utilisation on these layers: a 4 x 4 x 16 voxel tile is ONE staging round (load ->
// convert -> LDS -> barrier) followed by 1.4 us of MFMAs and an epilogue, and its weight fragments stream from L1 (one 1 KB fragment per
// 32-cycle MFMA and wave would be twice the L1 bandwidth of a CU at full matrix rate). Here a 256-thread workgroup owns an 8 (y) x 16 (x)
// voxel column and marches it along z:
//   * one wave per SIMD owns a lane's whole register file: the 27 x J weight fragments of the layer (216 registers at 32 input
//     channels) are loaded ONCE per workgroup and pinned in AGPRs, from where the MFMA reads its B operand directly;
//   * LDS holds a ring of 4 haloed input planes (10 x 18 voxels x 32 channels as 16-bit, 14.4 KB each): output plane z reads planes
//     z - 1, z, z + 1 (27 x J conflict-free ds_read_b128 per wave) while plane z + 2 is converted and written; every input voxel is
//     fetched once per column (halo 1.4x, no z re-reads: 2.5x in the tile form);
//   * one barrier per output plane; the global loads of plane z + 3 are issued a whole step ahead (two register sets), the residual /
//     normalised-tensor reads of the epilogue at the start of the step that consumes them;
//   * the epilogue of plane z - 1 (two accumulator sets), the conversion of plane z + 2 and the MFMAs of plane z are ONE basic block,
//     interleaved by scheduler directives -- with one wave per SIMD nothing else hides a latency. Fused statistics accumulate in
//     registers over the whole z range: one record per (z range, column).
// Wave w owns the M tile of rows 2 w, 2 w + 1 (32 voxels, the conflict-free lane -> voxel map of mtile_lane). Interior columns only
// (H % 8 == 0, W % 16 == 0, plain un-windowed output): the dispatcher keeps the tile kernel for everything else.
template <int J, int INMODE, int FUSE, bool F16>
__global__ __launch_bounds__(256) ONE_WAVE_PER_SIMD void conv3d_k3_lp_zring(ConvBArgs a) {
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HVP = HY * HX;      // haloed plane: 180 voxels
  constexpr int OCT = 2 * J;                 // channel octets of the (padded) input: 2 (16 channels) or 4 (32)
  constexpr int VSQ = OCT + 1;               // voxel stride in 16-byte units (odd)
  constexpr int PLANE = HVP * VSQ;           // uint4 per ring slot
  constexpr int UNITS = HVP * OCT;           // staging units (halo voxel, octet) per plane
  constexpr int UP = (UNITS + 255) / 256;    // per thread
  constexpr int NM = 27 * J;                 // MFMAs per output plane and wave
  static_assert(256 % OCT == 0, "a thread stages one fixed channel octet");
  static_assert(9 * UP + 16 <= NM, "the pieces of a step (conversion, unit stores, 16 output values) must fit its MFMAs");
  DYN_LDS(lds_f);
  uint4* lds = reinterpret_cast<uint4*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int zs = b % a.zsplits; b /= a.zsplits;
  const int n = b;
  const int zb = zs * a.zper, ze = zb + a.zper < a.Do ? zb + a.zper : a.Do;

  // ---- weights: all 27 x J fragments of this lane, once, pinned in the accumulation registers ----
  u32x4_t bw[NM];                              // (a native vector type: the register-class pin does not take HIP's uint4 struct)
  {
    const int CQ8 = a.CinP / 8;
    const uint4* wl = a.wp + (size_t)half * a.CoutP + li;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      bw[i] = __builtin_bit_cast(u32x4_t, wl[(size_t)((i / J) * CQ8 + 2 * (i % J)) * a.CoutP]);
      PIN_IN_AGPR(bw[i]);
    }
  }

  // ---- staging units of this thread: (halo voxel, octet so); geometry fixed for the column ----
  const int so = tid % OCT;
  const int c = 8 * so;
  const bool v0ok = c < a.Cin, v1ok = c + 4 < a.Cin;
  unsigned uoff[UP];                         // float offset inside an input plane (clamped, always valid)
  bool uin[UP];                              // inside the volume in y and x
  int ulds[UP];                              // uint4 offset inside a ring slot
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    const int u = tid + 256 * k;
    const int hv = u < UNITS ? u / OCT : HVP - 1;           // threads beyond the unit count repeat the last voxel's unit (identical store)
    const int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
    uin[k] = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
    const int iyc = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), ixc = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
    uoff[k] = (unsigned)((iyc * a.Wi + ixc) * a.xld);
    ulds[k] = hv * VSQ + so;
  }
  const int c0q = v0ok ? c : 0, c1q = v1ok ? c + 4 : c0q;
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
  const size_t xplane = (size_t)a.Hi * a.Wi * a.xld;
  const float* xn = a.x + (size_t)n * a.Di * xplane;
  // plane p (may lie outside the volume: clamped address, zeroed at the conversion) -> register set
  auto loads = [&](float4 (&ld)[UP][2], int p) {
    const int pc = p < 0 ? 0 : (p < a.Di ? p : a.Di - 1);
    const float* base = xn + (size_t)pc * xplane;             // workgroup-uniform
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      ld[k][0] = *reinterpret_cast<const float4*>(base + uoff[k] + c0q);
      ld[k][1] = *reinterpret_cast<const float4*>(base + uoff[k] + c1q);
    }
  };
  // conversion of one staged element (norm + activation, mask) / of a unit's eight elements into its ring slot. The step below
  // spreads these over the MFMAs of a plane one element at a time; the prologue runs a whole plane at once (`commit`).
  auto conv_elem = [&](const float4 (&ld)[UP][2], int k, int e, bool pin) -> float {
    const float4 q = ld[k][e >> 2];
    float v = (e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w;
    if (INMODE == MI355_IN_AFFINE_ACT) {
      const float u = v * sc[e] + sh[e];
      v = fmaxf(u, u * sl[e]);
    }
    return (pin && uin[k] && (e < 4 ? v0ok : v1ok)) ? v : 0.f;
  };
  auto store_unit = [&](const float (&v)[8], int k, auto slotc) {
    constexpr int SLOT = decltype(slotc)::value;
    uint4 pl[1];
    split8<1, F16>(v, pl);
    lds[SLOT * PLANE + ulds[k]] = pl[0];
  };
  auto commit = [&](const float4 (&ld)[UP][2], int p, auto slotc) {
    const bool pin = p >= 0 && p < a.Di;                      // workgroup-uniform
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = conv_elem(ld, k, e, pin);
      store_unit(v, k, slotc);
    }
  };

  // ---- this lane's A-operand position and its epilogue rows ----
  int lrow, ltx;
  mtile_lane(li, lrow, ltx);
  const int abase = ((2 * wave + lrow) * HX + ltx) * VSQ + half;
  const int co = li;                                          // Cout == 32 (dispatcher): every lane owns a real channel
  float bs = 0.f, cs = 1.f;
  if (a.bias) bs = a.bias[co];
  if (a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + co];
  float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
  bool first = true;
  if constexpr (FUSE == 2) {
    const int grp = co / (a.Cout / a.g.ggroups);
    gsc = a.g.gscale[(size_t)n * a.Cout + co]; gsh = a.g.gshift[(size_t)n * a.Cout + co];
    gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
  }
  // Output plane z: accumulator register r is x position r of x-row ((0b0110 >> (r >> 2)) & 1) ^ half (mtile_lane inverted). Every
  // address of the epilogue is a wave-uniform base (plane z, this wave's row pair, x position r: scalar registers) plus ONE 32-bit
  // lane offset per x-row (the 16 + 16 + 16 per-lane 64-bit pointers of the first version cost 60 vector registers).
  const size_t vrow0 = (((size_t)n * a.Do) * a.Ho + ty0 + 2 * wave) * a.Wo + tx0;      // plane 0, x-row 0 of this wave's tile, x = 0
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const unsigned yoA = (unsigned)(half * a.Wo * a.yld + co), yoB = (unsigned)((half ^ 1) * a.Wo * a.yld + co);
  const unsigned roA = (unsigned)(half * a.Wo * a.resld + co), roB = (unsigned)((half ^ 1) * a.Wo * a.resld + co);
  const unsigned goA = (unsigned)(half * a.Wo * a.g.gxld + co), goB = (unsigned)((half ^ 1) * a.Wo * a.g.gxld + co);
  // the reads that do not depend on the MFMAs (residual; the normalised tensor of the norm-backward form) are requested at the start
  // of the step that runs the epilogue: 16 + 16 dword loads in flight under that step's MFMAs
  struct Side { float rs[16], gx[16]; };
  auto side_loads = [&](Side& sd, int z) {
    const size_t v0 = vrow0 + (size_t)z * oplane;             // wave-uniform
    if constexpr (FUSE == 2) {
      const float* gb = a.g.gx + v0 * a.g.gxld;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.gx[r] = (gb + (size_t)r * a.g.gxld)[((0x6 >> (r >> 2)) & 1) ? goB : goA];
    }
    if (a.res) {                                             // workgroup-uniform
      const float* rb = a.res + v0 * a.resld;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = (rb + (size_t)r * a.resld)[((0x6 >> (r >> 2)) & 1) ? roB : roA];
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = 0.f;
    }
  };
  auto epilogue_value = [&](const f32x16& acc, const Side& sd, float* yb, int r) {      // yb: wave-uniform base of the plane
    const float v = (acc[r] + bs + sd.rs[r]) * cs;
    (yb + (size_t)r * a.yld)[((0x6 >> (r >> 2)) & 1) ? yoB : yoA] = v;
    if constexpr (FUSE == 1) {
      if (first && r == 0) K0 = v;
      const float t = v - K0;
      s0 += t; s1 += t * t;
    } else if constexpr (FUSE == 2) {
      const float xv = sd.gx[r];
      const float u = xv * gsc + gsh;
      const float du = u > 0.f ? v : v * a.g.gslope;
      s0 += du; s1 += du * ((xv - gmean) * grstd);
    }
  };
  auto epilogue = [&](const f32x16& acc, const Side& sd, int z) {
    float* yb = a.y + (vrow0 + (size_t)z * oplane) * a.yld;
#pragma unroll
    for (int r = 0; r < 16; ++r) epilogue_value(acc, sd, yb, r);
    first = false;
  };

  // ---- prologue: planes zb - 1, zb, zb + 1 into ring slots 0, 1, 2 (plane q of this range lives in slot (q - zb + 1) & 3); plane
  //      zb + 2 requested ----
  float4 ldA[UP][2], ldB[UP][2];
  loads(ldA, zb - 1);
  loads(ldB, zb);
  commit(ldA, zb - 1, std::integral_constant<int, 0>());
  loads(ldA, zb + 1);
  commit(ldB, zb, std::integral_constant<int, 1>());
  loads(ldB, zb + 2);
  commit(ldA, zb + 1, std::integral_constant<int, 2>());
  __syncthreads();

  f32x16 accE, accO;                                         // output planes at even / odd distance from zb
#pragma unroll
  for (int r = 0; r < 16; ++r) { accE[r] = 0.f; accO[r] = 0.f; }
  // One output plane. R = (z - zb) & 3: its input planes z - 1, z, z + 1 sit in slots R, R + 1, R + 2 (mod 4), plane z + 2 (in the
  // register set `cur`) is written to slot R + 3, plane z + 3 is requested into `nxt`; `acc` takes plane z while the epilogue of plane
  // z - 1 (in `prev`) rides along.
  auto step = [&](int z, auto rc, auto hpc, float4 (&cur)[UP][2], float4 (&nxt)[UP][2], f32x16& acc, const f32x16& prev) {
    constexpr int R = decltype(rc)::value;
    constexpr bool HASPREV = decltype(hpc)::value;           // all but the first plane of the range
    loads(nxt, z + 3);
    Side sd;
    if constexpr (HASPREV) side_loads(sd, z - 1);
    const bool pin = z + 2 >= 0 && z + 2 < a.Di;             // plane z + 2 (in `cur`) exists; workgroup-uniform
    float* yb = a.y + (vrow0 + (size_t)(z - 1) * oplane) * a.yld;
    SCHED_BARRIER();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A fragments: a ring of 4, requested three MFMAs ahead. After MFMA i one PIECE of the other work of the step is issued, pinned in
    // place (one wave per SIMD: whatever is not between two MFMAs idles the matrix pipe): pieces 0..23 convert one staged element of
    // plane z + 2 each, 24..26 pack and write its three units, 27..42 are the 16 output values of plane z - 1.
    auto afrag = [&](int i) {
      const int tap = i / J, j = i % J;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      return lds[abase + ((R + dz) & 3) * PLANE + (dy * HX + dx) * VSQ + 2 * j];
    };
    uint4 af[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) af[i] = afrag(i);
    float cv[UP][8];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (i + 3 < NM) af[(i + 3) & 3] = afrag(i + 3);
      acc = mfma_lp<F16>(af[i & 3], __builtin_bit_cast(uint4, bw[i]), acc);
      if (i < 8 * UP) cv[i / 8][i % 8] = conv_elem(cur, i / 8, i % 8, pin);
      else if (i < 8 * UP + UP) store_unit(cv[i - 8 * UP], i - 8 * UP, std::integral_constant<int, (R + 3) & 3>());
      else if (HASPREV && i < 9 * UP + 16) epilogue_value(prev, sd, yb, i - 9 * UP);
      SCHED_BARRIER();
    }
    if constexpr (HASPREV) first = false;
    __syncthreads();
  };
  // first plane of the range (slot phase R = 0, no previous plane), then the rest with the phase cycling 1, 2, 3, 0
  step(zb, std::integral_constant<int, 0>(), std::false_type(), ldB, ldA, accE, accO);
  for (int z = zb + 1; z < ze; z += 4) {
    step(z, std::integral_constant<int, 1>(), std::true_type(), ldA, ldB, accO, accE);
    if (z + 1 >= ze) break;
    step(z + 1, std::integral_constant<int, 2>(), std::true_type(), ldB, ldA, accE, accO);
    if (z + 2 >= ze) break;
    step(z + 2, std::integral_constant<int, 3>(), std::true_type(), ldA, ldB, accO, accE);
    if (z + 3 >= ze) break;
    step(z + 3, std::integral_constant<int, 0>(), std::true_type(), ldB, ldA, accE, accO);
  }
  {                                                          // the last plane's epilogue (nothing left to hide it under)
    Side sd;
    side_loads(sd, ze - 1);
    if ((ze - 1 - zb) & 1) epilogue(accO, sd, ze - 1); else epilogue(accE, sd, ze - 1);
  }

  if constexpr (FUSE != 0) {
    constexpr int K = FUSE == 1 ? 3 : 2;
    float vals[1][K];
    const int cnt = (ze - zb) * 16;
    if constexpr (FUSE == 1) {
      const float cf = (float)cnt;
      const float m2 = s1 - s0 * s0 / cf;
      vals[0][0] = cf; vals[0][1] = s0 + cf * K0; vals[0][2] = m2 > 0.f ? m2 : 0.f;
    } else {
      vals[0][0] = s0; vals[0][1] = s1;
    }
    const size_t rec = (size_t)n * ((size_t)a.zsplits * a.tilesY * a.tilesX) + ((size_t)zs * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
    float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * K;
    gn_fuse_reduce_store<K, 1, 4, 1>(vals, lds_f, wave, 0, half, li, tid, dst, 0, a.Cout);
  }
}

// z-range plan of conv3d_k3_lp_zring; use = 0: the call does not qualify (tile kernel). MI355_BF16_FORM: auto (default: eligible shapes
// with enough columns to fill the chip), zring (any eligible shape: tests), tile (never: the A/B switch). Measured at the end of round 3
// (profiles/r3_bf16_zring.txt): 32 -> 32 @128^3 0.53 -> 0.39 ms (plain), 0.54 -> 0.35 (norm + moments), bf16 step 33.7 -> 28.4 ms.
struct LpZPlan { int tilesY, tilesX, zsplits, zper, use; };
static LpZPlan plan_lp_zring(int n, int cin, int cout, int d, int h, int w, int precision, const mi355_conv_desc* desc) {
  LpZPlan p; memset(&p, 0, sizeof(p));
  const char* fe = getenv("MI355_BF16_FORM");
  const char form = fe && fe[0] ? fe[0] : 'a';
  if (form != 'z' && form != 'a') return p;
  if (precision != MI355_PREC_BF16 && precision != MI355_PREC_F16) return p;
  if (cin <= 16 || cin > 32 || cout != 32 || cin % 4 || h % 8 || w % 16 || d < 1) return p;      // 17..32 input channels (two k-steps per tap), 32 output channels (no channel mask in the epilogue)
  if (desc->pad != 1 || desc->off_z || desc->off_y || desc->off_x || desc->out_d != d || desc->out_h != h || desc->out_w != w) return p;
  p.tilesY = h / 8; p.tilesX = w / 16;
  const long long cols = (long long)n * p.tilesY * p.tilesX;
  if (form == 'a' && cols < 64) return p;                  // too few columns to fill the chip with whole-CU workgroups
  int zsplits = (int)((256 + cols - 1) / cols);
  const char* ze = getenv("MI355_BF16_ZSPLITS");            // tests: force the number of z ranges
  if (ze && atoi(ze) > 0) zsplits = atoi(ze);
  if (zsplits > d) zsplits = d;
  if (zsplits < 1) zsplits = 1;
  p.zper = ceil_div(d, zsplits);
  p.zsplits = ceil_div(d, p.zper);
  p.use = 1;
  return p;
}

// ---- weight packing: OIDHW fp32 -> [tap][ciP/8][plane][coP][8] bf16 planes ------------------------------------------
// mode 0: forward pack of a Conv3d weight; mode 1: dgrad pack (taps flipped, roles of ci/co swapped); cout/cin are the
// PACKED roles as in mi355_pack_conv_weight.
__global__ void pack_weight_bf16_kernel(const float* w, unsigned short* wp, int cout, int cin, int T, int coutP, int cinP, int mode, int NS, int f16) {
  const size_t total = (size_t)T * (cinP / 8) * coutP * 8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int e = idx & 7;
    size_t r = idx >> 3;
    const int o = r % coutP; r /= coutP;
    const int i8 = r % (cinP / 8); r /= (cinP / 8);
    const int t = (int)r;
    const int i = i8 * 8 + e;
    float v = 0.f;
    if (o < cout && i < cin) {
      const int tf = T - 1 - t;
      if (mode == 0) v = w[((size_t)o * cin + i) * T + t];
      else v = w[((size_t)i * cout + o) * T + tf];
    }
    for (int p = 0; p < NS; ++p) {
      const unsigned pk = f16 ? pack_f16x2(v, 0.f) : pack_bf16x2(v, 0.f);
      wp[((((size_t)t * (cinP / 8) + i8) * NS + p) * coutP + o) * 8 + e] = (unsigned short)(pk & 0xffffu);
      v -= bf16lo_to_f32(pk);
    }
  }
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
template <int TZ, int TY, int J, int NS, int WM, int WN, int MT, int NT, bool F16>
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
      LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 1, F16>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
    else
      LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_AFFINE_ACT, 1, F16>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  } else if (a.g.gnb) {
    if (in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 2, F16>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  } else if (in_mode == MI355_IN_PLAIN)
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_PLAIN, 0, F16>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  else
    LAUNCH((conv3d_k3_bf16<TZ, TY, J, NS, WM, WN, MT, NT, MI355_IN_AFFINE_ACT, 0, F16>), dim3((unsigned)blocks), dim3(256), lds, stream, a);
  return LAUNCH_CHECK();
}

template <int NS, bool F16 = false>
static int dispatch_ns(ConvBArgs& a, int in_mode, long long vox, void* stream) {
  // big volumes: 4x4x16 tiles (256 voxels), 4 waves along M; small: 2x4x16 tiles (128 voxels) so the grid still fills the chip
  constexpr int J = NS == 1 ? 2 : 1;      // (one 16-channel k-step per chunk for NS = 1 too: more workgroups per CU, measured 10 % slower)
  if constexpr (NS < 3) {     // the 3-plane tile of the big configuration would exceed the 64 KiB LDS window
    if (vox >= 256LL * 512) {
      if (a.Cout > 32) return launch_b<4, 4, J, NS, 4, 1, 2, 2, F16>(a, in_mode, stream);
      return launch_b<4, 4, J, NS, 4, 1, 2, 1, F16>(a, in_mode, stream);
    }
  }
  if (a.Cout > 32) return launch_b<2, 4, J, NS, 2, 2, 2, 1, F16>(a, in_mode, stream);
  return launch_b<2, 4, J, NS, 4, 1, 1, 1, F16>(a, in_mode, stream);
}

// spatial tiles (= epilogue records per sample) of the configuration dispatch_ns picks; 0: this call cannot fuse statistics
int32_t mi355_conv3d_bf16_stats_blocks(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  const int ns = nsplit_of(d->precision);
  if (!ns || d->kd != 3 || d->stride != 1 || d->out_mode != MI355_OUT_PLAIN) return 0;
  if (d->off_z || d->off_y || d->off_x || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return 0;
  const LpZPlan zp = plan_lp_zring(x->n, x->c, y->c, y->d, y->h, y->w, d->precision, d);
  if (zp.use && x->d == y->d && x->h == y->h && x->w == y->w) return (int32_t)((long long)zp.zsplits * zp.tilesY * zp.tilesX);
  const long long vox = (long long)y->d * y->h * y->w * x->n;
  const bool big = ns < 3 && vox >= 256LL * 512;
  const int tz = big ? 4 : 2, ty = 4;
  const long long b = (long long)ceil_div(y->d, tz) * ceil_div(y->h, ty) * ceil_div(y->w, 16);
  return b > 0 && b <= 0x7fffffffLL ? (int32_t)b : 0;
}

// called by mi355_conv3d_fwd (conv3d_fwd.hip) when desc->precision selects a bf16 path and the problem qualifies
int mi355_conv3d_fwd_bf16_impl(const mi355_act* x, const void* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  const int ns = nsplit_of(d->precision);
  if (!ns || d->kd != 3 || d->stride != 1) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  ConvBArgs a;
  memset(&a.g, 0, sizeof(a.g));
  if (d->moments_out || d->gn_bwd) {
    if (!mi355_conv3d_bf16_stats_blocks(x, y, d)) return MI355_EUNSUPPORTED;
    a.g.mom = d->moments_out;
    if (d->gn_bwd) {
      const mi355_gn_bwd_fuse* f = d->gn_bwd;
      if (!f->gx || !f->scale || !f->shift || !f->mean_rstd || !f->partials_out || f->groups <= 0 || y->c % f->groups || f->gx_ld < y->c) return MI355_EINVAL;
      a.g.gnb = f->partials_out; a.g.gx = f->gx; a.g.gxld = f->gx_ld; a.g.gscale = f->scale; a.g.gshift = f->shift; a.g.gmr = f->mean_rstd;
      a.g.ggroups = f->groups; a.g.gslope = f->act_slope;
    }
  }
  a.x = (const float*)x->p; a.xld = x->ld; a.wp = (const uint4*)wp; a.y = (float*)y->p; a.yld = y->ld;
  a.res = d->residual; a.resld = d->residual_ld;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.out_chscale = d->out_chscale; a.bias = d->bias;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Cin = x->c; a.CinP = (x->c + 15) / 16 * 16;
  a.Do = d->out_d; a.Ho = d->out_h; a.Wo = d->out_w; a.Cout = y->c; a.CoutP = (y->c + 31) / 32 * 32;
  a.yD = y->d; a.yH = y->h; a.yW = y->w; a.offz = d->off_z; a.offy = d->off_y; a.offx = d->off_x;
  a.pad = d->pad;
  if (a.res && a.resld < a.Cout) return MI355_EINVAL;
  const long long vox = (long long)a.Do * a.Ho * a.Wo * a.N;
  const LpZPlan zp = plan_lp_zring(a.N, a.Cin, a.Cout, a.Do, a.Ho, a.Wo, d->precision, d);
  if (zp.use && a.yD == a.Do && a.yH == a.Ho && a.yW == a.Wo && a.Di == a.Do && a.Hi == a.Ho && a.Wi == a.Wo) {
    if (a.g.mom && a.g.gnb) return MI355_EUNSUPPORTED;
    if (a.g.gnb && d->in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
    a.tilesY = zp.tilesY; a.tilesX = zp.tilesX; a.zsplits = zp.zsplits; a.zper = zp.zper;
    const long long blocks = (long long)a.N * zp.zsplits * zp.tilesY * zp.tilesX;
    if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
    const int fuse = a.g.mom ? 1 : (a.g.gnb ? 2 : 0);
    const bool f16 = d->precision == MI355_PREC_F16, norm = d->in_mode == MI355_IN_AFFINE_ACT;
    const int lds_bytes = 4 * 180 * 5 * 16;                   // ring of 4 planes, 180 voxels, 5 x 16 bytes per voxel (32 channels + pad)
    const dim3 grid((unsigned)blocks), blk(256);
#define LPZ_LAUNCH(JJ, IM, FU, HF)                                                                          \
    do { SET_MAX_DYN_LDS((conv3d_k3_lp_zring<JJ, IM, FU, HF>), lds_bytes);                                  \
         LAUNCH((conv3d_k3_lp_zring<JJ, IM, FU, HF>), grid, blk, lds_bytes, stream, a); } while (0)
#define LPZ_FUSE(JJ, HF)                                                                                    \
    do { if (fuse == 1) { if (norm) LPZ_LAUNCH(JJ, MI355_IN_AFFINE_ACT, 1, HF); else LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 1, HF); } \
         else if (fuse == 2) LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 2, HF);                                         \
         else if (norm) LPZ_LAUNCH(JJ, MI355_IN_AFFINE_ACT, 0, HF); else LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 0, HF); } while (0)
    if (f16) LPZ_FUSE(2, true); else LPZ_FUSE(2, false);
#undef LPZ_FUSE
#undef LPZ_LAUNCH
    return LAUNCH_CHECK();
  }
  if (d->precision == MI355_PREC_F16) return dispatch_ns<1, true>(a, d->in_mode, vox, stream);
  if (ns == 1) return dispatch_ns<1>(a, d->in_mode, vox, stream);
  if (ns == 2) return dispatch_ns<2>(a, d->in_mode, vox, stream);
  return dispatch_ns<3>(a, d->in_mode, vox, stream);
}
