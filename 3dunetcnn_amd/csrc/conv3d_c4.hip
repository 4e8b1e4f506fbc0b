// First-layer kernels: 3x3x3 stride-1 Conv3d with FOUR input channels (the 4 MRI modalities of BASELINE.json's
// "128^3 x 4ch" layer: UNet3D encoder block 0 conv1, unet3d/models/pytorch/classification/myronenko.py:17-21 via resnet.py:12-17;
// DynUNet input_block.conv1), forward and weight gradient, exact fp32 MFMA.
//
// The generic kernels pad the input channels to 8 (forward) / 32 (wgrad) per tap, i.e. run 2x / 8x the useful MFMAs on the
// layer whose algorithmic intensity is lowest (48 flop/B). Here the GEMM K (forward) / N (wgrad) index is the fused (tap, ci)
// pair: 27*4 = 108 values, so
//   forward: 54 v_mfma_f32_32x32x2_f32 per 32-voxel x 32-channel tile instead of 108;
//   wgrad:   4 MFMAs per voxel pair (one (tap,ci) tile of 32 per wave) instead of 27.
// Both stage the haloed 4-channel input tile in LDS as one float4 per voxel (normalised + activated on the way in); an MFMA
// operand is ONE float per lane, so the per-lane (tap, ci) gather is a plain ds_read_b32 whose tap offset is an instruction
// immediate (forward) or a per-lane constant (wgrad) -- no address arithmetic in the inner loops.
// The forward reads the UNPACKED OIDHW weight (3456 floats per 32 output channels, staged to LDS in [(tap,ci)][co] order).
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"
#include "act_io.h"

struct C4Args {
  const float* x; int xld;
  const float* w;                  // forward: OIDHW [Cout][4][27]
  float* y; int yld;
  const float* res; int resld;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* out_chscale; const float* bias;
  const float* dy; int dyld; float* ws;      // wgrad
  int N, D, H, W, Cout;
  int yD, yH, yW, offz, offy, offx;
  int tilesZ, tilesY, tilesX, coTiles, ntiles, splits;
  float* mom;                      // forward: fused norm statistics of the output (gn_fuse.h), or NULL
};

__device__ __forceinline__ float4 c4_prologue(float4 v, const float4& sc, const float4& sh, const float4& sl) {
  v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
  v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);   // 0 <= slope <= 1
  return v;
}

// stage the haloed tile (HZ x HY x HX voxels x 4 channels) of sample n at origin (z0-1, y0-1, x0-1) into lds_x[hv] (float4)
template <int HZ, int HY, int HX, int INMODE, int NTHREADS>
__device__ __forceinline__ void c4_stage_x(const C4Args& a, float4* lds_x, int n, int z0, int y0, int x0, int tid) {
  constexpr int HV = HZ * HY * HX;
  constexpr int UP = (HV + NTHREADS - 1) / NTHREADS;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(a.slope, a.slope, a.slope, a.slope);
  if (INMODE == MI355_IN_AFFINE_ACT) {
    sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * 4);
    sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * 4);
    if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope);
  }
  float4 ld[UP];
#pragma unroll
  for (int k = 0; k < UP; ++k) {          // all loads first, from clamped always-valid addresses
    int hv = tid + k * NTHREADS; if (hv >= HV) hv = HV - 1;
    int iz = z0 - 1 + hv / (HY * HX), iy = y0 - 1 + (hv / HX) % HY, ix = x0 - 1 + hv % HX;
    iz = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1);
    iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1);
    ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
    ld[k] = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + iz) * a.H + iy) * a.W + ix) * a.xld);
  }
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    const int hv = tid + k * NTHREADS;
    if (hv >= HV) continue;
    const int iz = z0 - 1 + hv / (HY * HX), iy = y0 - 1 + (hv / HX) % HY, ix = x0 - 1 + hv % HX;
    const bool ok = iz >= 0 && iy >= 0 && ix >= 0 && iz < a.D && iy < a.H && ix < a.W;
    float4 v = ld[k];
    if (INMODE == MI355_IN_AFFINE_ACT) v = c4_prologue(v, sc, sh, sl);
    lds_x[hv] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

// Epilogue of a 4x8x8 x 32-channel tile held as 2 M tiles per wave (wave = z plane, M tile mt = y rows 4mt .. 4mt+3, accumulator row
// r -> y = 4 mt + (r >> 2), x = (r & 3) + 4 half). INTERIOR tiles with a plain destination take the fast path: one base pointer and
// two strides, so a store costs two integer operations instead of the ~40 of the general index arithmetic -- in these kernels
// (54 or 7*P MFMAs per tile) the 32 stores per lane are a visible share of the tile, unlike in the big convolutions.
// Returns false when the tile needs the general path (ragged edge, window shift, destination of another extent).
// YT: storage type of y and of the residual (act_io.h). Statistics are taken over the values as stored.
template <int MT, typename YT = float>
__device__ __forceinline__ bool c4_store_fast(const C4Args& a, const f32x16 (&acc)[MT], int n, int tz0, int ty0, int tx0, int wave, int half, int co,
                                              bool fuse, float& mK, float& ms0, float& ms1) {
  if (tz0 + 4 > a.D || ty0 + 8 > a.H || tx0 + 8 > a.W || a.offz || a.offy || a.offx || a.yD != a.D || a.yH != a.H || a.yW != a.W) return false;
  if (co >= a.Cout) return true;
  const size_t vox0 = (((size_t)n * a.D + tz0 + wave) * a.H + ty0) * a.W + tx0 + 4 * half;
  YT* yp = reinterpret_cast<YT*>(a.y) + vox0 * a.yld + co;
  const YT* rp = a.res ? reinterpret_cast<const YT*>(a.res) + vox0 * a.resld + co : nullptr;
  const size_t yrow = (size_t)a.W * a.yld, rrow = (size_t)a.W * a.resld;
  const float bs = a.bias ? a.bias[co] : 0.f, cs = a.out_chscale ? a.out_chscale[(size_t)n * a.Cout + co] : 1.f;
  bool first = true;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int yy = 4 * mt + (r >> 2), xx = r & 3;
      float v = acc[mt][r] + bs;
      if (rp) v += ld1(rp + yy * rrow + (size_t)xx * a.resld);
      v *= cs;
      st1(yp + yy * yrow + (size_t)xx * a.yld, v);
      v = as_stored(yp, v);
      if (fuse) {
        if (first) { mK = v; first = false; }
        const float t = v - mK;
        ms0 += t; ms1 += t * t;
      }
    }
  return true;
}

// ---- forward: 4x8x8 output voxels x 32 output channels per workgroup; 4 waves x 2 M tiles ----
template <int INMODE, bool FUSE>
__global__ __launch_bounds__(256) void conv3d_c4_fwd(C4Args a) {
  constexpr int TZ = 4, TY = 8, TX = 8, HZ = 6, HY = 10, HX = 10, HV = HZ * HY * HX, MT = 2;
  __shared__ float4 lds_x[HV];            // 9600 B
  __shared__ float lds_w[108 * 33];       // [(tap*4+ci)][co], row stride 33: the k-fast staging writes are conflict-free
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int cot = b % a.coTiles; b /= a.coTiles;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
  const int n = b;
  const int co0 = cot * 32;
  for (int i = tid; i < 108 * 32; i += 256) {
    const int co = i / 108, k = i % 108;           // consecutive threads walk k (contiguous in OIDHW: [co][ci][tap])
    const int ci = k / 27, tap = k % 27;
    lds_w[(tap * 4 + ci) * 33 + co] = (co0 + co < a.Cout) ? a.w[((size_t)(co0 + co) * 4 + ci) * 27 + tap] : 0.f;
  }
  c4_stage_x<HZ, HY, HX, INMODE, 256>(a, lds_x, n, tz0, ty0, tx0, tid);
  __syncthreads();
  const float* xs = reinterpret_cast<const float*>(lds_x);
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tv = (wave * MT + mt) * 32 + li;
    abase[mt] = ((tv / (TY * TX)) * HY * HX + ((tv / TX) % TY) * HX + tv % TX) * 4 + half;     // + channel (k parity) via half
  }
  const int bbase = half * 33 + li;
  f32x16 acc[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
  // k = 2s + half = tap*4 + ci  ->  tap = s >> 1 (same for both halves), ci = 2*(s & 1) + half: every offset below is an immediate
#pragma unroll
  for (int s = 0; s < 54; ++s) {
    const int tap = s >> 1;
    const int toff = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 4 + 2 * (s & 1);
    const float bv = lds_w[bbase + s * 66];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) acc[mt] = MFMA_32x32x2(xs[abase[mt] + toff], bv, acc[mt]);
  }
  const int co = co0 + li;
  unsigned vmask = 0;                      // bit mt*16 + r: that accumulator row is a voxel of the output
  float mK = 0.f, ms0 = 0.f, ms1 = 0.f;    // FUSE: one-pass moments about K = the lane's first stored value
  const bool fast = c4_store_fast<MT>(a, acc, n, tz0, ty0, tx0, wave, half, co, FUSE, mK, ms0, ms1);
  if (fast) vmask = 0xffffffffu;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    if (fast) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const int tv = (wave * MT + mt) * 32 + row;
      const int oz = tz0 + tv / (TY * TX), oy = ty0 + (tv / TX) % TY, ox = tx0 + tv % TX;
      if (oz >= a.D || oy >= a.H || ox >= a.W) continue;
      const int sz = oz + a.offz, sy = oy + a.offy, sx = ox + a.offx;
      if (sz < 0 || sy < 0 || sx < 0 || sz >= a.yD || sy >= a.yH || sx >= a.yW) continue;
      const bool first = FUSE && vmask == 0;
      if (FUSE) vmask |= 1u << (mt * 16 + r);
      if (co >= a.Cout) continue;
      const size_t ovox = (((size_t)n * a.D + oz) * a.H + oy) * a.W + ox;
      const size_t svox = (((size_t)n * a.yD + sz) * a.yH + sy) * a.yW + sx;
      float v = acc[mt][r];
      if (a.bias) v += a.bias[co];
      if (a.res) v += a.res[ovox * a.resld + co];
      if (a.out_chscale) v *= a.out_chscale[(size_t)n * a.Cout + co];
      a.y[svox * a.yld + co] = v;
      if constexpr (FUSE) {
        if (first) mK = v;
        const float t = v - mK;
        ms0 += t; ms1 += t * t;
      }
    }
  }
  if constexpr (FUSE) {
    // the 32-channel x 128^3 output of this layer is the largest tensor a norm ever reads: its statistics leave with the tile
    float vals[1][3];
    const float cnt = (float)__builtin_popcount(vmask);
    const float m2 = cnt > 0.f ? ms1 - ms0 * ms0 / cnt : 0.f;
    vals[0][0] = cnt; vals[0][1] = ms0 + cnt * mK; vals[0][2] = m2 > 0.f ? m2 : 0.f;
    const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
    const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
    gn_fuse_reduce_store<3, 1, 4, 1>(vals, reinterpret_cast<float*>(lds_x), wave, 0, half, li, tid, a.mom + rec * a.Cout * 3, co0, a.Cout);
  }
}

// ---- forward on the bf16 matrix pipe with SPLIT fp32 operands (the opt-in precision modes, include/mi355_unet3d.h MI355_PREC_*) ----
// Same GEMM (K = the fused (tap, ci) index, 108 values padded to 7 k-steps of 16), but on v_mfma_f32_32x32x16_bf16: 7 * P MFMAs of
// 32 cycles per 32-voxel x 32-channel tile (P = 1 / 3 / 6 products) instead of 54 fp32 MFMAs of 64 cycles -- in exact fp32 this layer
// is matrix-bound at ~1.7 TB/s algorithmic (22 % of the HBM roofline; <= 41 % is the fp32 ceiling); on the bf16
// pipe it is the OUTPUT write that bounds it (32 channels x 128^3 fp32). A lane's A operand of a k-step is 8 consecutive k = the 4
// channels of two consecutive taps = two ds_read_b64 of the haloed tile, kept in LDS as NS bf16 planes of 4 channels per voxel; the
// B operand (weights, split once per workgroup while they are staged) is one ds_read_b128. A workgroup walks several z tiles of its
// (y, x) column so that the weight staging is amortised; the next tile's input loads are in flight during the MFMA loop.
template <int NS> struct C4Prod;
template <> struct C4Prod<1> { static constexpr int P = 1; static constexpr int pa[1] = {0}; static constexpr int pb[1] = {0}; };
template <> struct C4Prod<2> { static constexpr int P = 3; static constexpr int pa[3] = {1, 0, 0}; static constexpr int pb[3] = {0, 1, 0}; };
template <> struct C4Prod<3> { static constexpr int P = 6; static constexpr int pa[6] = {2, 1, 0, 1, 0, 0}; static constexpr int pb[6] = {0, 1, 2, 0, 1, 0}; };   // smallest terms first

template <int NS, int INMODE, bool FUSE, bool F16 = false, typename YT = float>      // F16: MI355_PREC_F16, the single plane is fp16
__global__ __launch_bounds__(256) void conv3d_c4_fwd_bf16(C4Args a) {
  YT* const ay = reinterpret_cast<YT*>(a.y);
  const YT* const ares = reinterpret_cast<const YT*>(a.res);
  constexpr int TZ = 4, TY = 8, TX = 8, HZ = 6, HY = 10, HX = 10, HV = HZ * HY * HX, MT = 2, KS = 7;
  constexpr int P = C4Prod<NS>::P;
  __shared__ uint2 lds_x[HV * NS];                 // [halo voxel][plane]: 4 bf16 (the 4 input channels)
  __shared__ uint4 lds_w[KS * 2 * NS * 32];        // [k-step][half][plane][co]: 8 bf16 = k 8*half .. 8*half+7 of that step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int cot = b % a.coTiles; b /= a.coTiles;
  const int zc = b % a.splits; b /= a.splits;                  // a.splits = z chunks per column, a.ntiles = z tiles per chunk
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int n = b;
  const int co0 = cot * 32;
  // ---- weights: OIDHW fp32 -> split bf16 planes in MFMA B-fragment order ----
  {
    unsigned short* w16 = reinterpret_cast<unsigned short*>(lds_w);
    for (int i = tid; i < KS * 16 * 32; i += 256) {
      const int co = i / (KS * 16), k = i % (KS * 16);
      const int tap = k >> 2, ci = k & 3;
      float v = (tap < 27 && co0 + co < a.Cout) ? a.w[((size_t)(co0 + co) * 4 + ci) * 27 + tap] : 0.f;
      const int ks = k >> 4, hh = (k >> 3) & 1, e = k & 7;
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const unsigned pk = pack_lp2<F16>(v, 0.f);
        w16[((((ks * 2 + hh) * NS + p) * 32 + co) << 3) + e] = (unsigned short)(pk & 0xffffu);
        v -= bf16lo_to_f32(pk);
      }
    }
  }
  // per-lane LDS offsets (in voxels) of the two taps this half supplies in every k-step; tap 27 is padding (zero weights)
  int o0[KS], o1[KS];
#pragma unroll
  for (int s2 = 0; s2 < KS; ++s2) {
    const int t0 = 4 * s2 + 2 * half, t1 = t0 + 1 < 27 ? t0 + 1 : 26;
    o0[s2] = ((t0 / 9) * HY + (t0 / 3) % 3) * HX + t0 % 3;
    o1[s2] = ((t1 / 9) * HY + (t1 / 3) % 3) * HX + t1 % 3;
  }
  int hv0[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int tv = (wave * MT + mt) * 32 + li;
    hv0[mt] = ((tv / (TY * TX)) * HY + (tv / TX) % TY) * HX + tv % TX;
  }
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(a.slope, a.slope, a.slope, a.slope);
  if (INMODE == MI355_IN_AFFINE_ACT) {
    sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * 4);
    sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * 4);
    if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope);
  }
  constexpr int UP = (HV + 255) / 256;
  float4 ld[UP];
  auto load_tile = [&](int tz0) {
#pragma unroll
    for (int k = 0; k < UP; ++k) {            // clamped, always-valid addresses: all loads in flight together
      int hv = tid + k * 256; if (hv >= HV) hv = HV - 1;
      int iz = tz0 - 1 + hv / (HY * HX), iy = ty0 - 1 + (hv / HX) % HY, ix = tx0 - 1 + hv % HX;
      iz = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1);
      iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1);
      ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      ld[k] = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + iz) * a.H + iy) * a.W + ix) * a.xld);
    }
  };
  auto commit_tile = [&](int tz0) {
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      const int hv = tid + k * 256;
      if (hv >= HV) continue;
      const int iz = tz0 - 1 + hv / (HY * HX), iy = ty0 - 1 + (hv / HX) % HY, ix = tx0 - 1 + hv % HX;
      const bool ok = iz >= 0 && iy >= 0 && ix >= 0 && iz < a.D && iy < a.H && ix < a.W;
      float4 v = ld[k];
      if (INMODE == MI355_IN_AFFINE_ACT) v = c4_prologue(v, sc, sh, sl);
      if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int p = 0; p < NS; ++p) {
        const unsigned lo = pack_lp2<F16>(v.x, v.y), hi = pack_lp2<F16>(v.z, v.w);
        lds_x[hv * NS + p] = make_uint2(lo, hi);
        if (p + 1 < NS) { v.x -= bf16lo_to_f32(lo); v.y -= bf16hi_to_f32(lo); v.z -= bf16lo_to_f32(hi); v.w -= bf16hi_to_f32(hi); }
      }
    }
  };
  const int tz_begin = zc * a.ntiles, tz_end = tz_begin + a.ntiles < a.tilesZ ? tz_begin + a.ntiles : a.tilesZ;
  if (tz_begin < tz_end) load_tile(tz_begin * TZ);
  const int co = co0 + li;
  for (int tzi = tz_begin; tzi < tz_end; ++tzi) {
    const int tz0 = tzi * TZ;
    __syncthreads();                          // the previous tile's MFMA loop / statistics scratch is done with lds_x
    commit_tile(tz0);
    __syncthreads();                          // (first iteration: also publishes lds_w)
    if (tzi + 1 < tz_end) load_tile(tz0 + TZ);
    SCHED_BARRIER();
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
    for (int s2 = 0; s2 < KS; ++s2) {
      uint4 bf[NS];
#pragma unroll
      for (int p = 0; p < NS; ++p) bf[p] = lds_w[((s2 * 2 + half) * NS + p) * 32 + li];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        uint4 af[NS];
#pragma unroll
        for (int p = 0; p < NS; ++p) {
          const uint2 lo = lds_x[(hv0[mt] + o0[s2]) * NS + p], hi = lds_x[(hv0[mt] + o1[s2]) * NS + p];
          af[p] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int q = 0; q < P; ++q) acc[mt] = mfma_lp<F16>(af[C4Prod<NS>::pa[q]], bf[C4Prod<NS>::pb[q]], acc[mt]);
      }
    }
    // ---- epilogue (as conv3d_c4_fwd) ----
    unsigned vmask = 0;
    float mK = 0.f, ms0 = 0.f, ms1 = 0.f;
    const bool fast = c4_store_fast<MT, YT>(a, acc, n, tz0, ty0, tx0, wave, half, co, FUSE, mK, ms0, ms1);
    if (fast) vmask = 0xffffffffu;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (fast) break;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        const int tv = (wave * MT + mt) * 32 + row;
        const int oz = tz0 + tv / (TY * TX), oy = ty0 + (tv / TX) % TY, ox = tx0 + tv % TX;
        if (oz >= a.D || oy >= a.H || ox >= a.W) continue;
        const int sz = oz + a.offz, sy = oy + a.offy, sx = ox + a.offx;
        if (sz < 0 || sy < 0 || sx < 0 || sz >= a.yD || sy >= a.yH || sx >= a.yW) continue;
        const bool first = FUSE && vmask == 0;
        if (FUSE) vmask |= 1u << (mt * 16 + r);
        if (co >= a.Cout) continue;
        const size_t ovox = (((size_t)n * a.D + oz) * a.H + oy) * a.W + ox;
        const size_t svox = (((size_t)n * a.yD + sz) * a.yH + sy) * a.yW + sx;
        float v = acc[mt][r];
        if (a.bias) v += a.bias[co];
        if (a.res) v += ld1(ares + ovox * a.resld + co);
        if (a.out_chscale) v *= a.out_chscale[(size_t)n * a.Cout + co];
        st1(ay + svox * a.yld + co, v);
        v = as_stored(ay, v);
        if constexpr (FUSE) {
          if (first) mK = v;
          const float t = v - mK;
          ms0 += t; ms1 += t * t;
        }
      }
    }
    if constexpr (FUSE) {
      float vals[1][3];
      const float cnt = (float)__builtin_popcount(vmask);
      const float m2 = cnt > 0.f ? ms1 - ms0 * ms0 / cnt : 0.f;
      vals[0][0] = cnt; vals[0][1] = ms0 + cnt * mK; vals[0][2] = m2 > 0.f ? m2 : 0.f;
      const int tile = (tzi * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
      const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
      gn_fuse_reduce_store<3, 1, 4, 1>(vals, reinterpret_cast<float*>(lds_x), wave, 0, half, li, tid, a.mom + rec * a.Cout * 3, co0, a.Cout);
    }
  }
}

// ---- wgrad: one 32-output-channel tile per workgroup column; wave w owns (tap, ci) columns 32w .. 32w+31 ----
template <int INMODE, typename TD = float>      // TD: storage type of dy (x, the network input, is fp32)
__global__ __launch_bounds__(256) void conv3d_c4_wgrad(C4Args a) {
  const TD* const ady = reinterpret_cast<const TD*>(a.dy);
  constexpr int TZ = 4, TY = 4, TX = 8, TV = TZ * TY * TX, HZ = 6, HY = 6, HX = 10, HV = HZ * HY * HX;
  __shared__ float4 lds_x[HV];            // 5760 B
  __shared__ float lds_dy[TV * 32];       // 16384 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, li = lane & 31;
  const int split = blockIdx.x, cot = blockIdx.y, co0 = cot * 32;
  const int jcol = wave * 32 + li;                      // (tap, ci) column of this lane
  const int tap = jcol >> 2 < 27 ? jcol >> 2 : 26, ci = jcol & 3;
  const int boff = (((tap / 9) * HY + (tap / 3) % 3) * HX + tap % 3) * 4 + ci;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int per = (a.ntiles + a.splits - 1) / a.splits;
  const int t_begin = split * per, t_end = t_begin + per < a.ntiles ? t_begin + per : a.ntiles;
  const int sq = tid & 7, sv0 = tid >> 3;
  const int cdy = co0 + 4 * sq;
  const bool dyvalid = cdy < a.Cout;
  const float* xs = reinterpret_cast<const float*>(lds_x);
  for (int tile = t_begin; tile < t_end; ++tile) {
    int b = tile;
    const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
    const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
    const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
    const int n = b;
    __syncthreads();
    {
      float4 ld[TV / 32];
#pragma unroll
      for (int k = 0; k < TV / 32; ++k) {
        const int v = sv0 + k * 32;
        int oz = tz0 + v / (TY * TX), oy = ty0 + (v / TX) % TY, ox = tx0 + v % TX;
        oz = oz < a.D ? oz : a.D - 1; oy = oy < a.H ? oy : a.H - 1; ox = ox < a.W ? ox : a.W - 1;
        ld[k] = ld4(ady + ((((size_t)n * a.D + oz) * a.H + oy) * a.W + ox) * a.dyld + (dyvalid ? cdy : 0));
      }
#pragma unroll
      for (int k = 0; k < TV / 32; ++k) {
        const int v = sv0 + k * 32;
        const int oz = tz0 + v / (TY * TX), oy = ty0 + (v / TX) % TY, ox = tx0 + v % TX;
        const bool ok = dyvalid && oz < a.D && oy < a.H && ox < a.W;
        *reinterpret_cast<float4*>(lds_dy + v * 32 + 4 * sq) = ok ? ld[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    c4_stage_x<HZ, HY, HX, INMODE, 256>(a, lds_x, n, tz0, ty0, tx0, tid);
    __syncthreads();
#pragma unroll 8
    for (int ks = 0; ks < TV / 2; ++ks) {
      const int v = 2 * ks + half;
      const int xb = ((v / (TY * TX)) * HY * HX + ((v / TX) % TY) * HX + v % TX) * 4;
      acc = MFMA_32x32x2(lds_dy[v * 32 + li], xs[xb + boff], acc);
    }
  }
  // partial tile -> ws[cot][split][32 co][128 (tap,ci) columns] (coalesced: the 32 lanes of a half write 128 contiguous bytes)
  float* dst = a.ws + ((size_t)cot * a.splits + split) * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    dst[row * 128 + jcol] = acc[r];
  }
}

// deterministic split reduction: dw[co][ci][tap] = sum over splits. One workgroup per 32 consecutive (co, column) elements:
// 8 split-groups x 32 elements, 4 independent partial sums per thread, groups combined through LDS in a fixed order.
__global__ __launch_bounds__(256) void conv3d_c4_wgrad_reduce(const float* ws, float* dw, int Cout, int splits) {
  __shared__ float part[8][32];
  const int cot = blockIdx.x >> 7;
  const int e = (blockIdx.x & 127) * 32 + (threadIdx.x & 31);
  const int grp = threadIdx.x >> 5;
  const float* src = ws + (size_t)cot * splits * 4096 + e;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int k = grp;
  for (; k + 24 < splits; k += 32) {
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] += src[(size_t)(k + 8 * u) * 4096];
  }
  for (int u = 0; k < splits; k += 8, ++u) acc[u & 3] += src[(size_t)k * 4096];
  part[grp][threadIdx.x & 31] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = part[0][threadIdx.x];
#pragma unroll
    for (int g = 1; g < 8; ++g) s += part[g][threadIdx.x];
    const int col = e & 127, co = cot * 32 + (e >> 7);
    if (col < 108 && co < Cout) dw[((size_t)co * 4 + (col & 3)) * 27 + (col >> 2)] = s;
  }
}

// the same reduction for the slabs of conv3d_c4_bwd (conv3d_c4_bwd.hip): one channel tile, `splits` slabs
void conv3d_c4_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int splits, void* stream) {
  LAUNCH(conv3d_c4_wgrad_reduce, dim3(128), dim3(256), 0, stream, ws, dw, Cout, splits);
}

// ---- dgrad of the first layer: 3x3x3 stride-1 conv with <= 4 OUTPUT channels on the vector ALU ----
// The gradient wrt the activated first-layer input (needed for the gamma/beta gradients of the network's first norm) is a
// conv Cout -> 4. On the 32-wide MFMA N tile 7/8 of the matrix work would be padding (measured: 2.0 ms at 128^3, batch 2 --
// as long as a 32 -> 32 layer). Here one thread owns one output voxel and its <= 4 output channels: the haloed input tile
// (6x10x10 voxels x CQ*4 = 16 channels = 38.4 KB, XOR-swizzled by (x + y) to spread the ds_read_b128 of x-neighbours and adjacent
// y rows over the banks without padding) is staged once per 16-channel chunk; per (tap, 4-channel group) a thread issues one
// ds_read_b128 and 8 packed FMAs whose weight operands are wave-uniform (scalar loads straight from the packed
// [tap][ci/4][32][4] buffer, only the first 4 of the 32 padded output columns are touched). fp32 FMA chains: exact products.
struct NarrowArgs {
  const float* x; int xld;
  const float* wp;                 // fp32 pack [27][CinQ][32][4]
  float* y; int yld;
  int N, D, H, W, Cq, CinQ, Cout;   // Cq = input channels / 4 (real quads), CinQ = quads of the pack (cinP / 4)
  int tilesZ, tilesY, tilesX, spatialTiles;
};

template <int CQ, typename XT = float, typename YT = float>      // storage types of the input (a gradient tensor) and of the <= 4-channel output
__global__ __launch_bounds__(256) void conv3d_c4_dgrad(NarrowArgs a) {
  const XT* const ax = reinterpret_cast<const XT*>(a.x);
  constexpr int TZ = 4, TY = 8, TX = 8, HZ = 6, HY = 10, HX = 10, HV = HZ * HY * HX;
  DYN_LDS(lds_f);                           // float4 [HV][CQ], quad q of haloed voxel (hz,hy,hx) at slot q ^ ((hx + hy) & (CQ - 1))
  float4* lds = reinterpret_cast<float4*>(lds_f);
  const int tid = threadIdx.x, lz = tid >> 6, ly = (tid >> 3) & 7, lx = tid & 7;
  int b = blockIdx.x;
  {                                         // XCD-aware order: each XCD (workgroup id % 8) walks a contiguous range of tiles
    const int per = a.spatialTiles / 8;
    if (b < per * 8) b = (b & 7) * per + (b >> 3);
  }
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int tz0 = (b % a.tilesZ) * TZ; b /= a.tilesZ;
  const int n = b;
  pkf2 acc[4][2];                           // per output channel: partial sums over input channels 4q + {0,1} and 4q + {2,3}
#pragma unroll
  for (int o = 0; o < 4; ++o) acc[o][0] = acc[o][1] = make_pkf2(0.f, 0.f);
  constexpr int UNITS = HV * CQ, UP = (UNITS + 255) / 256;
  for (int c0 = 0; c0 < a.Cq; c0 += CQ) {
    const int nq = a.Cq - c0 < CQ ? a.Cq - c0 : CQ;
    __syncthreads();
    {
      float4 ld[UP];
#pragma unroll
      for (int k = 0; k < UP; ++k) {        // all loads first, from clamped always-valid addresses
        int u = tid + k * 256; if (u >= UNITS) u = UNITS - 1;
        const int hv = u / CQ; int q = u % CQ; if (q >= nq) q = nq - 1;
        int iz = tz0 - 1 + hv / (HY * HX), iy = ty0 - 1 + (hv / HX) % HY, ix = tx0 - 1 + hv % HX;
        iz = iz < 0 ? 0 : (iz < a.D ? iz : a.D - 1);
        iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1);
        ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
        ld[k] = ld4(ax + ((((size_t)n * a.D + iz) * a.H + iy) * a.W + ix) * a.xld + 4 * (c0 + q));
      }
#pragma unroll
      for (int k = 0; k < UP; ++k) {
        const int u = tid + k * 256;
        if (u >= UNITS) continue;
        const int hv = u / CQ, q = u % CQ;
        const int hz = hv / (HY * HX), hy = (hv / HX) % HY, hx = hv % HX;
        const int iz = tz0 - 1 + hz, iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        const bool ok = q < nq && iz >= 0 && iy >= 0 && ix >= 0 && iz < a.D && iy < a.H && ix < a.W;
        lds[hv * CQ + (q ^ ((hx + hy) & (CQ - 1)))] = ok ? ld[k] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    __syncthreads();
    for (int tap = 0; tap < 27; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int hy = ly + dy, hx = lx + dx;
      const int hv = ((lz + dz) * HY + hy) * HX + hx, sw = (hx + hy) & (CQ - 1);
      const float* wt = a.wp + (size_t)tap * a.CinQ * 128;              // [CinQ][32 output columns][4]: columns 0..3 used
      float4 v[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) v[q] = lds[hv * CQ + (q ^ sw)];      // all reads in flight before the first FMA
#pragma unroll
      for (int q = 0; q < CQ; ++q) {
        // branch-free over the chunk: quads past the input's last one hold zeros in LDS; their weight address is clamped
        // into the pack (0 * w contributes nothing)
        const int qq = c0 + q < a.CinQ ? c0 + q : a.CinQ - 1;
        const float* wq = wt + qq * 128;
        const pkf2 vlo = make_pkf2(v[q].x, v[q].y), vhi = make_pkf2(v[q].z, v[q].w);
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          acc[o][0] = pk_fma(vlo, make_pkf2(wq[o * 4 + 0], wq[o * 4 + 1]), acc[o][0]);
          acc[o][1] = pk_fma(vhi, make_pkf2(wq[o * 4 + 2], wq[o * 4 + 3]), acc[o][1]);
        }
      }
    }
  }
  const int oz = tz0 + lz, oy = ty0 + ly, ox = tx0 + lx;
  if (oz < a.D && oy < a.H && ox < a.W) {
    YT* dst = reinterpret_cast<YT*>(a.y) + ((((size_t)n * a.D + oz) * a.H + oy) * a.W + ox) * a.yld;
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < a.Cout) st1(dst + o, (acc[o][0].x + acc[o][0].y) + (acc[o][1].x + acc[o][1].y));
  }
}

static void c4_fill(C4Args& a, const mi355_act* x, const mi355_conv_desc* d) {
  memset(&a, 0, sizeof(a));
  a.x = (const float*)x->p; a.xld = x->ld;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w;
}

int mi355_conv3d_c4_ok(const mi355_act* x, const mi355_conv_desc* d) {
  return x && d && x->c == 4 && d->kd == 3 && d->stride == 1 && d->pad == 1 && d->out_mode == MI355_OUT_PLAIN &&
         (d->in_mode == MI355_IN_PLAIN || d->in_mode == MI355_IN_AFFINE_ACT);
}

// wp = the UNPACKED OIDHW weight
int mi355_conv3d_c4_fwd_impl(const mi355_act* x, const float* w, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!mi355_conv3d_c4_ok(x, d) || d->out_d != x->d || d->out_h != x->h || d->out_w != x->w) return MI355_EUNSUPPORTED;
  // the 4-channel input is the network's input volume: fp32. A 16-bit output is written by the 16-bit-precision kernels only.
  if (x->dtype != MI355_ACT_F32 || !act_dtype_ok(y) || (y->dtype != MI355_ACT_F32 && d->precision == MI355_PREC_F32)) return MI355_EUNSUPPORTED;
  C4Args a; c4_fill(a, x, d);
  a.w = w; a.y = (float*)y->p; a.yld = y->ld; a.res = (const float*)d->residual; a.resld = d->residual_ld;
  a.out_chscale = d->out_chscale; a.bias = d->bias; a.Cout = y->c;
  a.yD = y->d; a.yH = y->h; a.yW = y->w; a.offz = d->off_z; a.offy = d->off_y; a.offx = d->off_x;
  if (a.res && a.resld < a.Cout) return MI355_EINVAL;
  if (d->gn_bwd) return MI355_EUNSUPPORTED;
  if (d->moments_out && (d->off_z || d->off_y || d->off_x || y->d != x->d || y->h != x->h || y->w != x->w)) return MI355_EUNSUPPORTED;
  a.mom = d->moments_out;
  a.tilesZ = ceil_div(a.D, 4); a.tilesY = ceil_div(a.H, 8); a.tilesX = ceil_div(a.W, 8); a.coTiles = ceil_div(a.Cout, 32);
  const long long blocks = (long long)a.N * a.tilesZ * a.tilesY * a.tilesX * a.coTiles;
  if (blocks <= 0 || blocks > 0x7fffffffLL) return MI355_EINVAL;
  if (d->precision != MI355_PREC_F32) {
    // bf16 matrix pipe with split operands: a workgroup walks a.ntiles z tiles of its (y, x) column; enough z chunks for >= ~2048 workgroups
    const long long cols = (long long)a.N * a.tilesY * a.tilesX * a.coTiles;
    int zs = (int)((2048 + cols - 1) / cols); if (zs < 1) zs = 1; if (zs > a.tilesZ) zs = a.tilesZ;
    a.ntiles = ceil_div(a.tilesZ, zs); a.splits = ceil_div(a.tilesZ, a.ntiles);
    const long long wg = cols * a.splits;
    if (wg > 0x7fffffffLL) return MI355_EINVAL;
    const int ns = d->precision == MI355_PREC_BF16X3 ? 2 : (d->precision == MI355_PREC_BF16X6 ? 3 : 1);
    if (!act_matches_precision(y->dtype, d->precision)) return MI355_EUNSUPPORTED;      // 16-bit storage goes with operands of its own type
#define MI355_C4B(NSV, IM, FU) \
    do { if (y->dtype == MI355_ACT_F16) LAUNCH((conv3d_c4_fwd_bf16<1, IM, FU, true, f16_t>), dim3((unsigned)wg), dim3(256), 0, stream, a); \
         else if (d->precision == MI355_PREC_F16) LAUNCH((conv3d_c4_fwd_bf16<NSV == 1 ? 1 : NSV, IM, FU, NSV == 1>), dim3((unsigned)wg), dim3(256), 0, stream, a); \
         else if (y->dtype == MI355_ACT_BF16) LAUNCH((conv3d_c4_fwd_bf16<1, IM, FU, false, bf16_t>), dim3((unsigned)wg), dim3(256), 0, stream, a); \
         else LAUNCH((conv3d_c4_fwd_bf16<NSV, IM, FU>), dim3((unsigned)wg), dim3(256), 0, stream, a); } while (0)
#define MI355_C4B_NS(NSV)                                                                                  \
    do {                                                                                                   \
      if (a.mom) { if (d->in_mode == MI355_IN_PLAIN) MI355_C4B(NSV, MI355_IN_PLAIN, true); else MI355_C4B(NSV, MI355_IN_AFFINE_ACT, true); }   \
      else { if (d->in_mode == MI355_IN_PLAIN) MI355_C4B(NSV, MI355_IN_PLAIN, false); else MI355_C4B(NSV, MI355_IN_AFFINE_ACT, false); }       \
    } while (0)
    if (ns == 1) MI355_C4B_NS(1); else if (ns == 2) MI355_C4B_NS(2); else MI355_C4B_NS(3);
#undef MI355_C4B_NS
#undef MI355_C4B
    return LAUNCH_CHECK();
  }
  if (a.mom) {
    if (d->in_mode == MI355_IN_PLAIN) LAUNCH((conv3d_c4_fwd<MI355_IN_PLAIN, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else LAUNCH((conv3d_c4_fwd<MI355_IN_AFFINE_ACT, true>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
  } else {
    if (d->in_mode == MI355_IN_PLAIN) LAUNCH((conv3d_c4_fwd<MI355_IN_PLAIN, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
    else LAUNCH((conv3d_c4_fwd<MI355_IN_AFFINE_ACT, false>), dim3((unsigned)blocks), dim3(256), 0, stream, a);
  }
  return LAUNCH_CHECK();
}

static int c4_wgrad_plan(const mi355_act* x, const mi355_act* dy, C4Args& a) {
  a.tilesZ = ceil_div(x->d, 4); a.tilesY = ceil_div(x->h, 4); a.tilesX = ceil_div(x->w, 8); a.coTiles = ceil_div(dy->c, 32);
  const long long nt = (long long)x->n * a.tilesZ * a.tilesY * a.tilesX;
  if (nt <= 0 || nt > 0x7fffffffLL) return 0;
  a.ntiles = (int)nt;
  int splits = ceil_div(1024, a.coTiles);
  const int max_splits = a.ntiles >= 8 ? a.ntiles / 8 : 1;
  if (splits > max_splits) splits = max_splits;
  const int per = ceil_div(a.ntiles, splits);
  a.splits = ceil_div(a.ntiles, per);
  return 1;
}

size_t mi355_conv3d_c4_wgrad_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  C4Args a; memset(&a, 0, sizeof(a));
  if (!mi355_conv3d_c4_ok(x, d) || !c4_wgrad_plan(x, dy, a)) return 0;
  return (size_t)a.coTiles * a.splits * 4096 * sizeof(float);
}

int mi355_conv3d_c4_wgrad_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                               void* ws, size_t ws_bytes, void* stream) {
  if (!mi355_conv3d_c4_ok(x, d) || x->d != dy->d || x->h != dy->h || x->w != dy->w) return MI355_EUNSUPPORTED;
  if (x->dtype != MI355_ACT_F32 || !act_dtype_ok(dy)) return MI355_EUNSUPPORTED;
  C4Args a; c4_fill(a, x, d);
  if (!c4_wgrad_plan(x, dy, a)) return MI355_EINVAL;
  if (ws_bytes < (size_t)a.coTiles * a.splits * 4096 * sizeof(float)) return MI355_EWORKSPACE;
  a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws; a.Cout = dy->c;
  dim3 grid(a.splits, a.coTiles);
  ACT_TYPED(dy->dtype, TD,
            if (d->in_mode == MI355_IN_PLAIN) LAUNCH((conv3d_c4_wgrad<MI355_IN_PLAIN, TD>), grid, dim3(256), 0, stream, a);
            else LAUNCH((conv3d_c4_wgrad<MI355_IN_AFFINE_ACT, TD>), grid, dim3(256), 0, stream, a));
  int rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(conv3d_c4_wgrad_reduce, dim3(a.coTiles * 128), dim3(256), 0, stream, (const float*)ws, dw, a.Cout, a.splits);
  return LAUNCH_CHECK();
}

// 3x3x3 stride-1 pad-1 conv with <= 4 output channels from the fp32 pack (any mode of mi355_pack_conv_weight), no epilogue fusion
int mi355_conv3d_narrow_ok(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  return x && y && d && y->c >= 1 && y->c <= 4 && x->c % 4 == 0 && d->kd == 3 && d->stride == 1 && d->pad == 1 &&
         d->in_mode == MI355_IN_PLAIN && d->out_mode == MI355_OUT_PLAIN && !d->bias && !d->residual && !d->out_chscale &&
         d->off_z == 0 && d->off_y == 0 && d->off_x == 0 && d->out_d == x->d && d->out_h == x->h && d->out_w == x->w &&
         y->d == x->d && y->h == x->h && y->w == x->w;
}

int mi355_conv3d_narrow_impl(const mi355_act* x, const float* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!mi355_conv3d_narrow_ok(x, y, d)) return MI355_EUNSUPPORTED;
  NarrowArgs a; memset(&a, 0, sizeof(a));
  a.x = (const float*)x->p; a.xld = x->ld; a.wp = wp; a.y = (float*)y->p; a.yld = y->ld;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cq = x->c / 4; a.CinQ = (x->c + 7) / 8 * 2; a.Cout = y->c;
  a.tilesZ = ceil_div(a.D, 4); a.tilesY = ceil_div(a.H, 8); a.tilesX = ceil_div(a.W, 8);
  const long long sp = (long long)a.N * a.tilesZ * a.tilesY * a.tilesX;
  if (sp <= 0 || sp > 0x7fffffffLL) return MI355_EINVAL;
  a.spatialTiles = (int)sp;
  // 16 input channels per LDS chunk: 38.4 KB and 103 VGPRs -> four workgroups per CU (measured on the 128^3 batch-2 layer: 0.53 ms;
  // 32-channel chunks, two workgroups per CU: 0.67 ms; 8-channel chunks: 0.96 ms)
  constexpr size_t lds = (size_t)6 * 10 * 10 * 4 * 16;
  if (!act_dtype_ok(x) || !act_dtype_ok(y)) return MI355_EINVAL;
  ACT_TYPED(x->dtype, XT, ACT_TYPED(y->dtype, YT, LAUNCH((conv3d_c4_dgrad<4, XT, YT>), dim3((unsigned)sp), dim3(256), lds, stream, a)));
  return LAUNCH_CHECK();
}
