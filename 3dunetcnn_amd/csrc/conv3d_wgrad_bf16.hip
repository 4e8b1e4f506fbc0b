// 3x3x3 stride-1 conv3d weight gradient on gfx950 bf16 MFMA (v_mfma_f32_32x32x16_bf16) with split fp32 operands.
//
//   dw[co][ci][tap] = sum_{n,v} dy[n,v,co] * in(x)[n, v + tap - 1, ci]        (autograd of F.conv3d wrt the weight;
//   reference layers: unet3d/models/pytorch/classification/resnet.py:12-22, input transform myronenko.py:18-19)
//
// Same precision modes as conv3d_bf16.hip (NS bf16 planes per operand, Products<NS> MFMA products, fp32 accumulate).
// GEMM view per tap: D[co][ci] (32x32) += A[co][voxel] * B[voxel][ci], K = 16 consecutive x-voxels per MFMA. Both operands
// need 8 consecutive VOXELS per lane, so the tiles are transposed to channel-major while they are staged (each thread loads
// 8 voxels x 4 channels as float4s, applies the fused norm + activation, splits into bf16 planes and writes one 16-byte
// run per channel). LDS (16-byte units, odd channel strides -> conflict-free ds_read_b128):
//   dy tile  [plane][32*MT co][4 y-rows][2 x-octets]
//   x  tile  [plane][32 ci][4 z-planes (ring)][6 y-rows][3 x-octets]   (halo: x0-1 .. x0+16)
// A workgroup = 9 consumer waves = the 9 (dz,dy) tap rows + 3 producer waves. A consumer owns the 3 dx taps of its row
// (3*MT accumulators) and forms the dx-shifted B fragments from two aligned octets with funnel shifts (v_alignbit) -- no
// unaligned LDS access, no copies. The workgroup walks a contiguous range of output tiles (1 x 4 x 16 voxels) with z fastest
// and keeps a ring of four input z-planes in LDS, so each step stages ONE new plane (x halo re-read factor 1.7 instead of 5)
// while the consumers compute on the other three. Partial sums go to the same
// workspace slabs as the f32 kernel and are reduced by the same deterministic second pass.
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream);

struct WgradBArgs {
  const float* x; int xld;
  const float* dy; int dyld;
  float* ws;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  int N, D, H, W, Cin, Cout;
  int tilesY, tilesX, ntiles;      // tile index = ((n*tilesY + ty)*tilesX + tx)*D + z   (z fastest)
  int splits, ciTiles, coTiles32;
};

template <int NS> struct WProducts;
template <> struct WProducts<1> { static constexpr int P = 1; static constexpr int pa[1] = {0}; static constexpr int pb[1] = {0}; };
template <> struct WProducts<2> { static constexpr int P = 3; static constexpr int pa[3] = {1, 0, 0}; static constexpr int pb[3] = {0, 1, 0}; };
template <> struct WProducts<3> { static constexpr int P = 6; static constexpr int pa[6] = {2, 1, 0, 1, 0, 0}; static constexpr int pb[6] = {0, 1, 2, 0, 1, 0}; };

// column c of an 8-voxel x 4-channel register block -> NS planes of 8 packed bf16
template <int NS, bool F16 = false>
__device__ __forceinline__ void split_col(const float (&v)[8][4], int c, uint4 (&out)[NS]) {
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = v[e][c];
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

// 8 bf16 starting `dx` elements into the 16-element run {b0, b1}
template <int DX>
__device__ __forceinline__ uint4 shift_run(const uint4& b0, const uint4& b1) {
  if (DX == 0) return b0;
  if (DX == 1) return make_uint4((b0.x >> 16) | (b0.y << 16), (b0.y >> 16) | (b0.z << 16), (b0.z >> 16) | (b0.w << 16), (b0.w >> 16) | (b1.x << 16));
  return make_uint4(b0.y, b0.z, b0.w, b1.x);
}

// Steady-state producer of the single-product 32-output-channel form (UNMEASURED, round-5 candidate; tools/NEXT.md). Inside a column of tiles
// (fixed sample, y and x origin; only z moves) a producer thread stages the SAME unit of every tile: everything but z is a per-column
// constant -- 8-voxel base offset, validity bits, LDS slot, the unit's norm parameters. The generic staging code of the kernel below spends
// ~650 instructions per unit re-deriving them every step (unit decomposition, eight clamps, eight 64-bit addresses, eight predicates).
template <int INMODE, bool F16, typename TA>
struct WgradLpLean {
  static constexpr int TY = 4, HY = 6, XO = 3, CSA = 9, CSB = 73, NXU = HY * XO * 8, NDYU = 4 * 2 * 8;
  const WgradBArgs& a;
  uint4* ldsA; uint4* ldsB;
  int ci0, co0, ltid;
  struct Col {
    const TA* base;                 // tensor + sample + channel of this thread's unit (input or dy)
    unsigned rowoff; int ix0, ld;   // element offset of voxel e inside a z plane = rowoff + clamp(ix0 + e) * ld
    unsigned plane;                 // elements per z plane of that tensor
    unsigned ok;                    // bit e: voxel e exists (row, column and channel inside the tensor)
    int lds;                        // uint4 index of the unit's channel-0 write, without the ring-slot (input) / buffer (dy) term
    float4 sc, sh;                  // fused norm of the unit's 4 channels (input units)
    int ch;
    bool isx;
  };
  __device__ __forceinline__ void col_setup(Col& c, int tile) const {
    const int col = tile / a.D;
    const int tx0 = (col % a.tilesX) * 16, ty0 = ((col / a.tilesX) % a.tilesY) * TY, n = col / (a.tilesX * a.tilesY);
    const int u = ltid < NXU + NDYU ? ltid : NXU + NDYU - 1;        // (threads past the last unit shadow it; they never write)
    c.isx = u < NXU;
    int ch, iy, ix0, ld; bool chok;
    c.ok = 0;
    if (c.isx) {
      const int ro = u >> 3, oct = ro % XO, hy = ro / XO, q = u & 7;
      ch = ci0 + 4 * q; chok = ch < a.Cin; iy = ty0 - 1 + hy; ix0 = tx0 - 1 + 8 * oct; ld = a.xld;
      c.lds = (4 * q) * CSB + hy * XO + oct;
#pragma unroll
      for (int e = 0; e < 8; ++e) c.ok |= (unsigned)(chok && iy >= 0 && iy < a.H && ix0 + e >= 0 && ix0 + e < a.W && 8 * oct + e < 18) << e;
    } else {
      const int v = u - NXU, qq = v % 8, ro = v / 8, row = ro >> 1, oct = ro & 1;
      ch = co0 + 4 * qq; chok = ch < a.Cout; iy = ty0 + row; ix0 = tx0 + 8 * oct; ld = a.dyld;
      c.lds = (4 * qq) * CSA + row * 2 + oct;
#pragma unroll
      for (int e = 0; e < 8; ++e) c.ok |= (unsigned)(chok && iy < a.H && ix0 + e < a.W) << e;
    }
    if (!chok) ch = 0;
    c.ch = ch;
    const int iyc = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1);
    c.plane = (unsigned)a.H * (unsigned)a.W * (unsigned)ld;
    c.base = reinterpret_cast<const TA*>(c.isx ? a.x : a.dy) + (size_t)n * a.D * c.plane + ch;
    c.rowoff = (unsigned)iyc * (unsigned)a.W * (unsigned)ld; c.ix0 = ix0; c.ld = ld;
    c.sc = make_float4(1.f, 1.f, 1.f, 1.f); c.sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (INMODE == MI355_IN_AFFINE_ACT && c.isx && chok) {
      c.sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + ch);
      c.sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + ch);
    }
  }
  // the unit of the tile at depth z: input units read plane z + 1 (the one new plane of that tile), dy units plane z
  __device__ __forceinline__ void issue(float4 (&t8)[8], const Col& c, int z) const {
    int iz = c.isx ? z + 1 : z;
    iz = iz < a.D ? iz : a.D - 1;
    const TA* p = c.base + (size_t)iz * c.plane;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ix = c.ix0 + e;
      const int ixc = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      t8[e] = ld4(p + (c.rowoff + (unsigned)ixc * (unsigned)c.ld));
    }
  }
  __device__ __forceinline__ void commit(const float4 (&t8)[8], const Col& c, int z, int buf) const {
    if (ltid >= NXU + NDYU) return;
    const unsigned ok = (c.isx && z + 1 >= a.D) ? 0u : c.ok;          // the plane past the volume is the zero halo
    float4 sl = make_float4(a.slope, a.slope, a.slope, a.slope);
    if (INMODE == MI355_IN_AFFINE_ACT && a.in_slope && c.isx) sl = *reinterpret_cast<const float4*>(a.in_slope + c.ch);      // (rare: DynUNet concat)
    unsigned w[4][4];                                                  // [channel][voxel pair]
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      float4 t[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = 2 * pr + h;
        t[h] = t8[e];
        if (INMODE == MI355_IN_AFFINE_ACT) {
          if (c.isx) {                                                 // (dy units are not normalised / activated)
            t[h].x = t[h].x * c.sc.x + c.sh.x; t[h].y = t[h].y * c.sc.y + c.sh.y; t[h].z = t[h].z * c.sc.z + c.sh.z; t[h].w = t[h].w * c.sc.w + c.sh.w;
            t[h].x = fmaxf(t[h].x, t[h].x * sl.x); t[h].y = fmaxf(t[h].y, t[h].y * sl.y);
            t[h].z = fmaxf(t[h].z, t[h].z * sl.z); t[h].w = fmaxf(t[h].w, t[h].w * sl.w);
          }
        }
        if (!((ok >> e) & 1u)) { t[h].x = 0.f; t[h].y = 0.f; t[h].z = 0.f; t[h].w = 0.f; }
      }
      w[0][pr] = pack_lp2<F16>(t[0].x, t[1].x); w[1][pr] = pack_lp2<F16>(t[0].y, t[1].y);
      w[2][pr] = pack_lp2<F16>(t[0].z, t[1].z); w[3][pr] = pack_lp2<F16>(t[0].w, t[1].w);
    }
    uint4* dst = c.isx ? ldsB + c.lds + ((z + 1) & 3) * (HY * XO) : ldsA + c.lds + buf * (32 * CSA);
    const int cs = c.isx ? CSB : CSA;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) dst[cc * cs] = make_uint4(w[cc][0], w[cc][1], w[cc][2], w[cc][3]);
  }
};

// Producer / consumer workgroup: waves 0..8 are the 9 (dz,dy) tap rows and only read LDS + issue MFMAs; the NLW producer
// waves only stage: while the consumers work on tile t they load tile t+1 (its dy tile and the one new input plane) from
// global memory -- every load of the tile issued before the first use, from clamped always-valid addresses (a branch around a
// load would serialise the round trips) --, transform/split/transpose it and write it to the other dy buffer / the free slot
// of a 4-plane ring. One barrier per tile; global-memory latency is never on the consumers' path.
// TA: storage type of x and dy (act_io.h; bf16 storage halves the bytes this kernel is bound by, profiles/r4_ab_experiments.txt section 6)
template <int NS, int MT, int NLW, int INMODE, bool F16 = false, typename TA = float>      // F16: MI355_PREC_F16, the single plane is fp16
// Register budget of the single-product, 32-output-channel form: 13 waves per workgroup over 4 SIMDs = 4 + 3 + 3 + 3; at the 79 - 88 registers
// hipcc takes when left alone a SIMD holds 5 waves, i.e. ONE workgroup per CU -- producers and consumers of one workgroup in lock step through
// a barrier per 64-voxel step, nothing to fill the waits. 72 registers (7 waves per SIMD) let TWO workgroups (26 waves) share the CU.
// UNMEASURED (round-5 candidate): A/B with tools/build_variant.sh <tag> conv3d_wgrad_bf16.hip -DWGRAD_LP_WAVES=1 (= hipcc's own allocation).
#ifndef WGRAD_LP_WAVES
#define WGRAD_LP_WAVES 7
#endif
__global__ __launch_bounds__(576 + 64 * NLW) MIN_WAVES_PER_SIMD((NS == 1 && MT == 1) ? WGRAD_LP_WAVES : 1) void conv3d_wgrad_k3_bf16(WgradBArgs a) {
  constexpr int TY = 4, ROWS = 4, HY = 6, XO = 3, RING = 4;
  constexpr int COT = 32 * MT;
  constexpr int CSA = ROWS * 2 + 1;            // 9
  constexpr int CSB = RING * HY * XO + 1;      // 73
  constexpr int P = WProducts<NS>::P;
  constexpr int NCONS = 576, NLOAD = 64 * NLW;
  constexpr int NXU = HY * XO * 8;             // staging units (8 voxels x 4 channels) of one input plane: 144
  constexpr int NDYU = ROWS * 2 * 8 * MT;      // of one dy tile: 64 * MT
  constexpr int UPT = (NXU + NDYU + NLOAD - 1) / NLOAD;   // units per producer thread and tile
  DYN_LDS(lds_f);
  uint4* ldsA = reinterpret_cast<uint4*>(lds_f);       // [2 buffers][NS][COT][CSA]
  uint4* ldsB = ldsA + 2 * NS * COT * CSA;             // [NS][32][CSB]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 9;
  const int split = blockIdx.x, cit = blockIdx.y, cot = blockIdx.z;
  const int ci0 = cit * 32, co0 = cot * COT;
  const int per = (a.ntiles + a.splits - 1) / a.splits;
  const int t_begin = split * per;
  const int t_end = t_begin + per < a.ntiles ? t_begin + per : a.ntiles;

  if (loader) {
    // ================================ producers ================================
    const int ltid = tid - NCONS;
    // unit u of a tile: u < nx -> input-plane unit (plane index u / NXU relative to z_lo), else dy unit u - nx
    auto stage = [&](int tile, int buf, int z_lo, int nplanes, bool with_dy) {
      const int z = tile % a.D, col = tile / a.D;
      const int tx0 = (col % a.tilesX) * 16, ty0 = ((col / a.tilesX) % a.tilesY) * TY, n = col / (a.tilesX * a.tilesY);
      const int nx = nplanes * NXU, total = nx + (with_dy ? NDYU : 0);
      for (int u0 = ltid; u0 < total; u0 += NLOAD * UPT) {
        float4 t8[UPT][8];
        // ---- issue every load of this round ----
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
          const int u = u0 + k * NLOAD;
          const bool isx = u < nx;
          int c, iz, iy, ix0, ld; const TA* base;
          if (isx) {
            const int ro = u >> 3, oct = ro % XO, hy = (ro / XO) % HY;
            c = ci0 + 4 * (u & 7); if (c >= a.Cin) c = 0;
            iz = z_lo + ro / (XO * HY); iy = ty0 - 1 + hy; ix0 = tx0 - 1 + 8 * oct; ld = a.xld; base = reinterpret_cast<const TA*>(a.x);
          } else {
            const int v = u - nx, qq = v % (8 * MT), ro = v / (8 * MT);
            c = co0 + 4 * qq; if (c >= a.Cout) c = 0;
            iz = z; iy = ty0 + (ro >> 1); ix0 = tx0 + 8 * (ro & 1); ld = a.dyld; base = reinterpret_cast<const TA*>(a.dy);
          }
          const int izc = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1), iyc = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1);
          const TA* rowp = base + (((size_t)n * a.D + izc) * a.H + iyc) * (size_t)a.W * ld + c;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ix = ix0 + e;
            const int ixc = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
            t8[k][e] = ld4(rowp + (size_t)ixc * ld);
          }
        }
        // ---- transform, split, transpose, write ----
#pragma unroll
        for (int k = 0; k < UPT; ++k) {
          const int u = u0 + k * NLOAD;
          if (u >= total) continue;
          float v[8][4];
          if (u < nx) {
            const int ro = u >> 3, oct = ro % XO, hy = (ro / XO) % HY, q = u & 7;
            const int c = ci0 + 4 * q;
            const int iz = z_lo + ro / (XO * HY), iy = ty0 - 1 + hy, ix0 = tx0 - 1 + 8 * oct;
            const bool rowok = c < a.Cin && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H;
            float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f}, sl[4] = {a.slope, a.slope, a.slope, a.slope};
            if (INMODE == MI355_IN_AFFINE_ACT && c < a.Cin) {
              const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c);
              const float4 h4 = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c);
              sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
              sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
              if (a.in_slope) { const float4 l4 = *reinterpret_cast<const float4*>(a.in_slope + c); sl[0] = l4.x; sl[1] = l4.y; sl[2] = l4.z; sl[3] = l4.w; }
            }
            const int slot = iz & 3;                          // iz >= -1: (-1 & 3) == 3
            if constexpr (NS == 1) {
              // single-plane operands: two voxels become one packed dword per channel as soon as both are transformed -- no 8 x 4 float
              // block between the loads and the four LDS writes (the kernel's register count decides how many workgroups share a CU)
              unsigned wq[4][4];
#pragma unroll
              for (int pr = 0; pr < 4; ++pr) {
                float4 tt[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                  const int e = 2 * pr + h2, ix = ix0 + e;
                  float4 t = t8[k][e];
                  if (INMODE == MI355_IN_AFFINE_ACT) {
                    t.x = t.x * sc[0] + sh[0]; t.y = t.y * sc[1] + sh[1]; t.z = t.z * sc[2] + sh[2]; t.w = t.w * sc[3] + sh[3];
                    t.x = fmaxf(t.x, t.x * sl[0]); t.y = fmaxf(t.y, t.y * sl[1]); t.z = fmaxf(t.z, t.z * sl[2]); t.w = fmaxf(t.w, t.w * sl[3]);
                  }
                  const bool ok = rowok && ix >= 0 && ix < a.W && 8 * oct + e < 18;
                  tt[h2] = ok ? t : make_float4(0.f, 0.f, 0.f, 0.f);
                }
                wq[0][pr] = pack_lp2<F16>(tt[0].x, tt[1].x); wq[1][pr] = pack_lp2<F16>(tt[0].y, tt[1].y);
                wq[2][pr] = pack_lp2<F16>(tt[0].z, tt[1].z); wq[3][pr] = pack_lp2<F16>(tt[0].w, tt[1].w);
              }
#pragma unroll
              for (int cc = 0; cc < 4; ++cc)
                ldsB[(4 * q + cc) * CSB + (slot * HY + hy) * XO + oct] = make_uint4(wq[cc][0], wq[cc][1], wq[cc][2], wq[cc][3]);
              continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int ix = ix0 + e;
              float4 t = t8[k][e];
              if (INMODE == MI355_IN_AFFINE_ACT) {
                t.x = t.x * sc[0] + sh[0]; t.y = t.y * sc[1] + sh[1]; t.z = t.z * sc[2] + sh[2]; t.w = t.w * sc[3] + sh[3];
                t.x = fmaxf(t.x, t.x * sl[0]); t.y = fmaxf(t.y, t.y * sl[1]); t.z = fmaxf(t.z, t.z * sl[2]); t.w = fmaxf(t.w, t.w * sl[3]);   // 0 <= slope <= 1
              }
              const bool ok = rowok && ix >= 0 && ix < a.W && 8 * oct + e < 18;
              v[e][0] = ok ? t.x : 0.f; v[e][1] = ok ? t.y : 0.f; v[e][2] = ok ? t.z : 0.f; v[e][3] = ok ? t.w : 0.f;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              uint4 pl[NS];
              split_col<NS, F16>(v, cc, pl);
#pragma unroll
              for (int p = 0; p < NS; ++p) ldsB[(p * 32 + 4 * q + cc) * CSB + (slot * HY + hy) * XO + oct] = pl[p];
            }
          } else {
            const int w = u - nx, qq = w % (8 * MT), ro = w / (8 * MT), row = ro >> 1, oct = ro & 1;
            const int co = co0 + 4 * qq, y = ty0 + row, x0 = tx0 + 8 * oct;
            if constexpr (NS == 1) {
              unsigned wq[4][4];
#pragma unroll
              for (int pr = 0; pr < 4; ++pr) {
                const bool ok0 = co < a.Cout && y < a.H && x0 + 2 * pr < a.W, ok1 = co < a.Cout && y < a.H && x0 + 2 * pr + 1 < a.W;
                const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 t0 = ok0 ? t8[k][2 * pr] : z4, t1 = ok1 ? t8[k][2 * pr + 1] : z4;
                wq[0][pr] = pack_lp2<F16>(t0.x, t1.x); wq[1][pr] = pack_lp2<F16>(t0.y, t1.y);
                wq[2][pr] = pack_lp2<F16>(t0.z, t1.z); wq[3][pr] = pack_lp2<F16>(t0.w, t1.w);
              }
#pragma unroll
              for (int cc = 0; cc < 4; ++cc)
                ldsA[(buf * COT + 4 * qq + cc) * CSA + row * 2 + oct] = make_uint4(wq[cc][0], wq[cc][1], wq[cc][2], wq[cc][3]);
              continue;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const bool ok = co < a.Cout && y < a.H && x0 + e < a.W;
              v[e][0] = ok ? t8[k][e].x : 0.f; v[e][1] = ok ? t8[k][e].y : 0.f; v[e][2] = ok ? t8[k][e].z : 0.f; v[e][3] = ok ? t8[k][e].w : 0.f;
            }
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
              uint4 pl[NS];
              split_col<NS, F16>(v, cc, pl);
#pragma unroll
              for (int p = 0; p < NS; ++p) ldsA[((buf * NS + p) * COT + 4 * qq + cc) * CSA + row * 2 + oct] = pl[p];
            }
          }
        }
      }
    };
    if (t_begin < t_end) stage(t_begin, 0, t_begin % a.D - 1, 3, true);       // prologue: everything tile t_begin needs
    __syncthreads();
#ifndef WGRAD_LP_NO_LEAN      // A/B switch (tools/build_variant.sh ... -DWGRAD_LP_NO_LEAN): the generic staging code in the steady state too
    if constexpr (NS == 1 && MT == 1) {
      using Lean = WgradLpLean<INMODE, F16, TA>;
      const Lean lean{a, ldsA, ldsB, ci0, co0, ltid};
      typename Lean::Col c;
      if (t_begin < t_end) lean.col_setup(c, t_begin);
      for (int tile = t_begin; tile < t_end; ++tile) {
        const int z = tile % a.D, buf = (tile - t_begin) & 1;
        const bool has_next = tile + 1 < t_end;
        const bool next_same_col = has_next && z + 1 < a.D;
        if (next_same_col) {
          float4 t8[8];
          lean.issue(t8, c, z + 1);
          lean.commit(t8, c, z + 1, buf ^ 1);
        }
        __syncthreads();
        if (has_next && !next_same_col) {
          stage(tile + 1, buf ^ 1, -1, 3, true);
          __syncthreads();
          lean.col_setup(c, tile + 1);
        }
      }
      return;
    }
#endif
    for (int tile = t_begin; tile < t_end; ++tile) {
      const int z = tile % a.D, buf = (tile - t_begin) & 1;
      const bool has_next = tile + 1 < t_end;
      const bool next_same_col = has_next && z + 1 < a.D;   // z fastest: the next tile continues this column iff z+1 < D
      if (next_same_col) stage(tile + 1, buf ^ 1, z + 2, 1, true);   // the one new plane -> the slot not read now
      __syncthreads();
      if (has_next && !next_same_col) {                      // new column: refill the ring while the consumers wait
        stage(tile + 1, buf ^ 1, -1, 3, true);
        __syncthreads();
      }
    }
    return;
  }

  // ================================ consumers ================================
  const int half = lane >> 5, li = lane & 31;
  const int wdz = wave / 3, wdy = wave % 3;
  f32x16 acc[3][MT];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[dx][mt][r] = 0.f;
  __syncthreads();                                           // prologue
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int z = tile % a.D, buf = (tile - t_begin) & 1;
    const bool has_next = tile + 1 < t_end;
    const bool next_same_col = has_next && z + 1 < a.D;
    {
      // 4 k-steps (y-rows of 16 voxels); this wave: tap row (wdz, wdy), dx = 0..2
      const int bslot = (z - 1 + wdz) & 3;
#pragma unroll
      for (int r = 0; r < ROWS; ++r) {
        uint4 af[MT][NS], b0[NS], b1[NS];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int p = 0; p < NS; ++p) af[mt][p] = ldsA[((buf * NS + p) * COT + mt * 32 + li) * CSA + r * 2 + half];
        const int hrow = bslot * HY + r + wdy;
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          b0[p] = ldsB[(p * 32 + li) * CSB + hrow * XO + half];
          b1[p] = ldsB[(p * 32 + li) * CSB + hrow * XO + half + 1];
        }
        uint4 bf[3][NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) { bf[0][p] = shift_run<0>(b0[p], b1[p]); bf[1][p] = shift_run<1>(b0[p], b1[p]); bf[2][p] = shift_run<2>(b0[p], b1[p]); }
        // products outermost: consecutive MFMAs go to different accumulators (no back-to-back dependent issue)
#pragma unroll
        for (int qq = 0; qq < P; ++qq)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
              acc[dx][mt] = mfma_lp<F16>(af[mt][WProducts<NS>::pa[qq]], bf[dx][WProducts<NS>::pb[qq]], acc[dx][mt]);
      }
    }
    __syncthreads();
    if (has_next && !next_same_col) __syncthreads();
  }

  // ---- write the partial tiles: ws[pair][slab][tap][32 co][32 ci] (same layout as the f32 kernel) ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int cot32 = cot * MT + mt;
    if (cot32 >= a.coTiles32) continue;
    const size_t pair = (size_t)cot32 * a.ciTiles + cit;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int tap = (wdz * 3 + wdy) * 3 + dx;
      float* dst = a.ws + (((pair * a.splits + split) * 27 + tap) * 1024);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        dst[row * 32 + li] = acc[dx][mt][r];
      }
    }
  }
}

static int nsplit_of_w(int precision) {
  switch (precision) {
    case MI355_PREC_BF16X3: return 2;
    case MI355_PREC_BF16X6: return 3;
    case MI355_PREC_BF16: return 1;
    case MI355_PREC_F16: return 1;
    default: return 0;
  }
}

struct WBPlan { int tilesY, tilesX, ntiles, splits, ciTiles, coTiles32, coTilesWG, mt; size_t ws_bytes; int ok; };

static WBPlan plan_wb(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WBPlan p; memset(&p, 0, sizeof(p));
  if (!x || !dy || !d || !nsplit_of_w(d->precision)) return p;
  if (d->kd != 3 || d->stride != 1 || d->pad != 1) return p;
  if (x->dtype != dy->dtype || !act_matches_precision(x->dtype, d->precision)) return p;      // 16-bit storage goes with operands of its own type
  if (x->d != dy->d || x->h != dy->h || x->w != dy->w) return p;
  p.tilesY = ceil_div(dy->h, 4); p.tilesX = ceil_div(dy->w, 16);
  const long long nt = (long long)dy->n * p.tilesY * p.tilesX * dy->d;
  if (nt <= 0 || nt > 0x7fffffffLL) return p;
  p.ntiles = (int)nt;
  p.ciTiles = ceil_div(x->c, 32); p.coTiles32 = ceil_div(dy->c, 32);
  p.mt = (dy->c > 32 && nsplit_of_w(d->precision) < 3) ? 2 : 1;     // 3 planes x 64 co would not fit the 160 KiB LDS
  { const char* e = getenv("MI355_WGRAD_LP_MT"); if (e && atoi(e) == 1) p.mt = 1; }      // A/B switch (tools/r5_prep_ab.sh): 32-channel workgroups everywhere
  p.coTilesWG = ceil_div(p.coTiles32, p.mt);
  const int wgs = p.ciTiles * p.coTilesWG;
  int splits = ceil_div(512, wgs);
  const int max_splits = p.ntiles >= 8 ? p.ntiles / 8 : 1;      // >= 8 tiles per workgroup: amortise the ring fill and the slab write
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int per = ceil_div(p.ntiles, splits);
  p.splits = ceil_div(p.ntiles, per);
  p.ws_bytes = (size_t)p.ciTiles * p.coTiles32 * p.splits * 27 * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}

size_t mi355_conv3d_wgrad_bf16_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WBPlan p = plan_wb(x, dy, d);
  return p.ok ? p.ws_bytes : 0;
}

template <int NS, int MT, bool F16 = false, typename TA = float>
static int launch_wb(WgradBArgs& a, const WBPlan& p, int in_mode, void* stream) {
  constexpr int NLW = MT == 1 ? 4 : 3;     // producer waves: 4 cover a tile's 208 staging units in one round; MT = 2 is VGPR-limited to 12 waves
  constexpr size_t lds = ((size_t)2 * NS * 32 * MT * 9 + (size_t)NS * 32 * 73) * 16;
  static_assert(lds <= 160 * 1024, "LDS");
  dim3 grid(p.splits, p.ciTiles, p.coTilesWG);
  if (in_mode == MI355_IN_PLAIN) {
    SET_MAX_DYN_LDS((conv3d_wgrad_k3_bf16<NS, MT, NLW, MI355_IN_PLAIN, F16, TA>), lds);
    LAUNCH((conv3d_wgrad_k3_bf16<NS, MT, NLW, MI355_IN_PLAIN, F16, TA>), grid, dim3(576 + 64 * NLW), lds, stream, a);
  } else {
    SET_MAX_DYN_LDS((conv3d_wgrad_k3_bf16<NS, MT, NLW, MI355_IN_AFFINE_ACT, F16, TA>), lds);
    LAUNCH((conv3d_wgrad_k3_bf16<NS, MT, NLW, MI355_IN_AFFINE_ACT, F16, TA>), grid, dim3(576 + 64 * NLW), lds, stream, a);
  }
  return LAUNCH_CHECK();
}

// called by mi355_conv3d_wgrad (conv3d_wgrad.hip) when desc->precision selects a bf16 path and the problem qualifies
int mi355_conv3d_wgrad_bf16_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                                 void* ws, size_t ws_bytes, void* stream) {
  WBPlan p = plan_wb(x, dy, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < p.ws_bytes) return MI355_EWORKSPACE;
  WgradBArgs a;
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = dy->c;
  a.tilesY = p.tilesY; a.tilesX = p.tilesX; a.ntiles = p.ntiles;
  a.splits = p.splits; a.ciTiles = p.ciTiles; a.coTiles32 = p.coTiles32;
  const int ns = nsplit_of_w(d->precision);
  int rc;
  if (x->dtype == MI355_ACT_BF16) rc = p.mt == 2 ? launch_wb<1, 2, false, bf16_t>(a, p, d->in_mode, stream) : launch_wb<1, 1, false, bf16_t>(a, p, d->in_mode, stream);
  else if (x->dtype == MI355_ACT_F16) rc = p.mt == 2 ? launch_wb<1, 2, true, f16_t>(a, p, d->in_mode, stream) : launch_wb<1, 1, true, f16_t>(a, p, d->in_mode, stream);
  else if (d->precision == MI355_PREC_F16) rc = p.mt == 2 ? launch_wb<1, 2, true>(a, p, d->in_mode, stream) : launch_wb<1, 1, true>(a, p, d->in_mode, stream);
  else if (p.mt == 2) rc = ns == 1 ? launch_wb<1, 2>(a, p, d->in_mode, stream) : launch_wb<2, 2>(a, p, d->in_mode, stream);
  else rc = ns == 1 ? launch_wb<1, 1>(a, p, d->in_mode, stream) : ns == 2 ? launch_wb<2, 1>(a, p, d->in_mode, stream) : launch_wb<3, 1>(a, p, d->in_mode, stream);
  if (rc) return rc;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, 27, p.splits, p.ciTiles, stream);
}
