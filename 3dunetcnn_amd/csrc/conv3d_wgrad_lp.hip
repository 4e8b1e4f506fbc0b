// 3x3x3 stride-1 conv3d weight gradient for 16-bit tensors (bf16 / fp16 storage with operands of the same type), gfx950:
// a z-marching workgroup whose staging writes the tiles AS LOADED (channels-last, one 16-byte run of 8 channels per request) and whose
// matrix operands come out of LDS already K-major through the transpose read ds_read_b64_tr_b16.
//
//   dw[co][ci][tap] = sum_{n,v} dy[n,v,co] * in(x)[n, v + tap - 1, ci]        (autograd of F.conv3d wrt the weight;
//   reference layers: unet3d/models/pytorch/classification/resnet.py:12-22, input transform myronenko.py:18-19)
//
// What it replaces: conv3d_wgrad_k3_bf16 (conv3d_wgrad_bf16.hip) on 16-bit tensors. That kernel's producer waves transpose every 8-voxel x
// 4-channel unit through registers (~200 vector instructions per unit) and one 64-voxel tile goes through a barrier per step: 0.17 of the bf16
// matrix peak on the 32 -> 32 layers of the 128^3 level (1.1 ms per launch at batch 4, HBM floor 0.29 ms).
//
// Workgroup = 6 matrix waves = (dz = 0..2) x (row half h = 0 / 1) + 2 staging waves, on one (32 ci, 32 co) pair and one column of 8 x 16
// output voxels, marching z:
//   * ring of 4 input planes [10 rows][18 voxels][32 ci] + 2 dy planes [8 rows][16 voxels][32 co] in LDS, 64 bytes per voxel; the staging
//     waves request the planes TWO steps ahead (two register sets), apply the fused norm + activation in channels-last form (8 fma + 8
//     mul/max + 4 packs per 16 bytes) and write them a step ahead: one barrier per plane. Eight waves = two per SIMD: SIMDs 0 and 1 carry two
//     matrix waves, SIMDs 2 and 3 one matrix wave and one staging wave (a wave's own vector instructions add to its MFMA time, another
//     wave's do not: profiles/r4_mfma_valu_overlap.txt);
//   * wave (dz, h) owns the 9 taps (dz, *, *) on output rows 4 h .. 4 h + 3: 144 accumulator registers. Per output row it reads ONE new input
//     row (three transpose reads: 12 consecutive voxels per lane, the dx = 1 / 2 fragments are funnel shifts of them) and one dy row (two
//     reads) for 9 MFMAs: 0.28 KB of LDS per v_mfma_f32_32x32x16 (the register-transposing kernel: 1 KB);
//   * the transpose read: inside each group of 16 lanes the 16 addressed 8-byte rows form a 16 x 4 matrix of 16-bit elements and lane l
//     receives column l & 3 of rows (l >> 2) + 4 j (profiles/r5_tr_read_probe.txt). Lane r of a group addresses voxel (r >> 2), channel
//     block 4 (r & 3): lane l then holds channel (l & 15) of 4 consecutive voxels -- the operand layout of the MFMA (lane = row / column,
//     8 consecutive k per lane and k-group) with two reads;
//   * the two row halves are added through LDS at the end; partial tiles go to the slab workspace of conv3d_wgrad.hip and its
//     deterministic reduction ([pair][slab = column chunk][tap][32 co][32 ci]).
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream);

// ds_read_b64_tr_b16 and its CPU twin for the emulated build (tools/emu: fibers, one per lane; the 16 row addresses travel by shuffle)
#ifdef MI355_EMU
static inline uint2 lds_read_tr16_b64(const unsigned char* p) {
  const int l = emu::flat_tid() % 64, gb = l & ~15, a = (l & 15) >> 2, c = l & 3;
  unsigned v[4];
  for (int j = 0; j < 4; ++j) {
    const unsigned char* row = emu_shfl(p, gb + a + 4 * j);
    unsigned short e; memcpy(&e, row + 2 * c, 2); v[j] = e;
  }
  return make_uint2(v[0] | (v[1] << 16), v[2] | (v[3] << 16));
}
#else
__device__ __forceinline__ uint2 lds_read_tr16_b64(const unsigned char* p) {
  typedef __attribute__((__vector_size__(4 * sizeof(short)))) short v4s;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
  auto* q = (__attribute__((address_space(3))) v4s*)p;
#pragma clang diagnostic pop
  return __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(q));
}
#endif

struct WgradTArgs {
  const void* x; int xld;
  const void* dy; int dyld;
  float* ws;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  int N, D, H, W, Cin, Cout;
  int tilesY, tilesX, zchunks, zper, ncol;      // column chunk = ((n * tilesY + ty) * tilesX + tx) * zchunks + zc
  int colsPer, nslab;                           // a workgroup walks colsPer consecutive column chunks into the same accumulators: nslab = ceil(ncol / colsPer) slabs
  int ciTiles, coTiles;
};

template <int INMODE, typename TA>
__global__ __launch_bounds__(512) void conv3d_wgrad_lp_tr(WgradTArgs a) {
  constexpr bool F16 = std::is_same<TA, f16_t>::value;
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2;
  constexpr int XPL = HY * HX * 64, DYPL = TY * TX * 64;          // bytes per staged plane
  constexpr int NP = 128;                                         // staging threads (waves 6, 7)
  constexpr int NXU = HY * HX * 4, NDYU = TY * TX * 4;            // 16-byte units per plane: 720, 512
  constexpr int UX = (NXU + NP - 1) / NP, UD = NDYU / NP;         // per staging thread: 6 input units, 4 dy units
  DYN_LDS(lds_f);
  unsigned char* const xs = reinterpret_cast<unsigned char*>(lds_f);
  unsigned char* const dys = xs + 4 * XPL;
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6);
  // workgroup -> (column chunk c, pair): the pairs of one column chunk sit on the same XCD (blockIdx % 8) next to each other in launch
  // order, so that the tiles they share come out of that XCD's L2
  const int P = a.ciTiles * a.coTiles;
  const int b = blockIdx.x, grp = b / (8 * P), rem = b % (8 * P);
  const int pair = rem >> 3, cg = grp * 8 + (rem & 7);             // cg: the workgroup's group of a.colsPer consecutive column chunks = its slab
  if (cg >= a.nslab) return;
  const int cit = pair % a.ciTiles, cot = pair / a.ciTiles;
  const int ci0 = cit * 32, co0 = cot * 32;
  const int c_begin = cg * a.colsPer, c_end = c_begin + a.colsPer < a.ncol ? c_begin + a.colsPer : a.ncol;
#define WT_DECODE_COLUMN(c)                                                              \
  int cc_ = (c);                                                                         \
  const int zc = cc_ % a.zchunks; cc_ /= a.zchunks;                                      \
  const int tx0 = (cc_ % a.tilesX) * TX; cc_ /= a.tilesX;                                \
  const int ty0 = (cc_ % a.tilesY) * TY;                                                 \
  const int n = cc_ / a.tilesY;                                                          \
  const int z0 = zc * a.zper, z1 = z0 + a.zper < a.D ? z0 + a.zper : a.D

  if (wave >= 6) {
    // ================================ the two staging waves ================================
    // (waves 6, 7 share SIMDs 2, 3 with one matrix wave each; SIMDs 0, 1 carry two matrix waves: the staging arithmetic fills the lighter pair)
    const int pt = tid - 384, q8 = pt & 3;                         // the thread's channel octet (all of its units: 128 % 4 == 0)
    for (int c = c_begin; c < c_end; ++c) {
    WT_DECODE_COLUMN(c);
    unsigned xgoff[UX], dgoff[UD], okx = 0, okd = 0;
#pragma unroll
    for (int k = 0; k < UX; ++k) {
      int u = pt + NP * k; const bool live = u < NXU; if (!live) u = NXU - 1;
      const int hv = u >> 2, hy = hv / HX, hx = hv % HX;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const bool in = iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      const int cy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1), cx = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      xgoff[k] = (unsigned)((cy * a.W + cx) * a.xld + ci0 + 8 * q8);
      okx |= (unsigned)(in && live) << k;
    }
#pragma unroll
    for (int k = 0; k < UD; ++k) {
      const int dv = (pt + NP * k) >> 2, oy = ty0 + (dv >> 4), ox = tx0 + (dv & 15);
      const bool din = oy < a.H && ox < a.W;
      dgoff[k] = (unsigned)(((din ? oy : 0) * a.W + (din ? ox : 0)) * a.dyld + co0 + 8 * q8);
      okd |= (unsigned)din << k;
    }
    const unsigned loff = (unsigned)(pt * 16);                     // unit u = pt + 128 k sits at byte 16 u of its plane: 64 (u >> 2) + 16 (u & 3)
    const size_t xplane = (size_t)a.H * a.W * a.xld, dyplane = (size_t)a.H * a.W * a.dyld;
    const TA* const xn = reinterpret_cast<const TA*>(a.x) + (size_t)n * a.D * xplane;
    const TA* const dyn = reinterpret_cast<const TA*>(a.dy) + (size_t)n * a.D * dyplane;
    float sc[8], sh[8], sl[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; sl[e] = a.slope; }
    if (INMODE == MI355_IN_AFFINE_ACT) {
      const float* ps = a.in_scale + (size_t)n * a.Cin + ci0 + 8 * q8;
      const float* ph = a.in_shift + (size_t)n * a.Cin + ci0 + 8 * q8;
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = ps[e]; sh[e] = ph[e]; }
      if (a.in_slope) {                                            // (rare: per-channel slopes of a concatenated input)
#pragma unroll
        for (int e = 0; e < 8; ++e) sl[e] = a.in_slope[ci0 + 8 * q8 + e];
      }
    }
    auto ld16 = [](const TA* p) { return *reinterpret_cast<const uint4*>(p); };
    auto load_x = [&](uint4 (&r)[UX], int p) {                     // input plane p (clamped: the out-of-volume planes are zeroed at the commit)
      const TA* pl = xn + (size_t)(p < 0 ? 0 : (p < a.D ? p : a.D - 1)) * xplane;
#pragma unroll
      for (int k = 0; k < UX; ++k) r[k] = ld16(pl + xgoff[k]);
    };
    auto load_dy = [&](uint4 (&r)[UD], int p) {
      const TA* pl = dyn + (size_t)(p < a.D ? p : a.D - 1) * dyplane;
#pragma unroll
      for (int k = 0; k < UD; ++k) r[k] = ld16(pl + dgoff[k]);
    };
    auto commit_x = [&](const uint4 (&r)[UX], int p) {
      const bool zok = p >= 0 && p < a.D;
      unsigned char* dstp = xs + (p & 3) * XPL + loff;             // (-1 & 3 == 3)
#pragma unroll
      for (int k = 0; k < UX; ++k) {
        if (pt + NP * k >= NXU) continue;
        unsigned w[4] = {r[k].x, r[k].y, r[k].z, r[k].w};
        if (INMODE == MI355_IN_AFFINE_ACT) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float lo = lp_lo<TA>(w[e]) * sc[2 * e] + sh[2 * e], hi = lp_hi<TA>(w[e]) * sc[2 * e + 1] + sh[2 * e + 1];
            lo = fmaxf(lo, lo * sl[2 * e]); hi = fmaxf(hi, hi * sl[2 * e + 1]);
            w[e] = lp_pack2<TA>(lo, hi);
          }
        }
        const bool ok = ((okx >> k) & 1u) && zok;
        *reinterpret_cast<uint4*>(dstp + k * (NP * 16)) = ok ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    auto commit_dy = [&](const uint4 (&r)[UD], int p) {
      unsigned char* dstp = dys + (p & 1) * DYPL + loff;
#pragma unroll
      for (int k = 0; k < UD; ++k) {
        const bool ok = ((okd >> k) & 1u) && p < a.D;
        *reinterpret_cast<uint4*>(dstp + k * (NP * 16)) = ok ? r[k] : make_uint4(0u, 0u, 0u, 0u);
      }
    };
    // prologue: input planes z0 - 1, z0, z0 + 1 and dy plane z0
    {
      uint4 p0[UX], p1[UX], p2[UX], pd[UD];
      load_x(p0, z0 - 1); load_x(p1, z0); load_x(p2, z0 + 1); load_dy(pd, z0);
      commit_x(p0, z0 - 1); commit_x(p1, z0); commit_x(p2, z0 + 1); commit_dy(pd, z0);
    }
    __syncthreads();
    // steady state, four steps deep (four register sets: the step is shorter than the memory latency under load -- with two sets the kernel
    // ran at two steps per round trip whatever the matrix waves did): step s writes the planes requested four steps ago (input plane s + 2,
    // dy plane s + 1) into the ring slot / dy buffer last read in step s - 1, which that step's barrier closed, and requests those of step s + 4
    uint4 X0[UX], X1[UX], X2[UX], X3[UX], D0[UD], D1[UD], D2[UD], D3[UD];
    load_x(X0, z0 + 2); load_dy(D0, z0 + 1); load_x(X1, z0 + 3); load_dy(D1, z0 + 2);
    load_x(X2, z0 + 4); load_dy(D2, z0 + 3); load_x(X3, z0 + 5); load_dy(D3, z0 + 4);
#define WT_STAGE_STEP(X, Dd, s) do { commit_x(X, (s) + 2); commit_dy(Dd, (s) + 1); load_x(X, (s) + 6); load_dy(Dd, (s) + 5); __syncthreads(); } while (0)
    for (int z = z0; z < z1; z += 4) {
      WT_STAGE_STEP(X0, D0, z);
      if (z + 1 >= z1) break;
      WT_STAGE_STEP(X1, D1, z + 1);
      if (z + 2 >= z1) break;
      WT_STAGE_STEP(X2, D2, z + 2);
      if (z + 3 >= z1) break;
      WT_STAGE_STEP(X3, D3, z + 3);
    }
#undef WT_STAGE_STEP
    }
    __syncthreads();                                               // (the matrix waves' exchange barrier)
    return;
  }

  // ================================ the six matrix waves ================================
  // wave = (dz, h); lane = 16 g + r: rows / columns 16 (g & 1) + (l & 15) of the operand, k-group g >> 1
  const int dz = wave % 3, h = wave / 3;
  const int g = lane >> 4, r16 = lane & 15, half = lane >> 5, li = lane & 31;
  const unsigned lb = (unsigned)((8 * (g >> 1) + (r16 >> 2)) * 64 + 32 * (g & 1) + 8 * (r16 & 3));
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  struct Row { uint4 f0, f2, f1; };                                // voxels 8 kg + 0..7 / 2..9 / 1..8 of an input row: the dx = 0 / 2 / 1 operands
  for (int c = c_begin; c < c_end; ++c) {
    WT_DECODE_COLUMN(c);
    (void)tx0; (void)ty0; (void)n;
    Row B[6]; uint4 A[4];
    const unsigned char* bp; const unsigned char* ap;
    auto set_plane = [&](int z) {
      bp = xs + ((z - 1 + dz) & 3) * XPL + lb + (4 * h) * (HX * 64);
      ap = dys + (z & 1) * DYPL + lb + (4 * h) * (TX * 64);
    };
    // input row rr of the wave's six: four transpose reads (voxels 0..3, 4..7, 2..5, 6..9 of the lane's run); dx = 1 is a funnel shift, once per row
    auto rdB = [&](int rr) {
      const unsigned char* q = bp + rr * (HX * 64);
#ifndef WT_ABL
#define WT_ABL 0
#endif
#if WT_ABL == 1        // timing only: two reads per row
      const uint2 v0 = lds_read_tr16_b64(q), v1 = lds_read_tr16_b64(q + 256);
      B[rr].f0 = make_uint4(v0.x, v0.y, v1.x, v1.y); B[rr].f2 = make_uint4(v0.y, v0.x, v1.y, v1.x);
#elif WT_ABL == 2      // three reads per row, dx = 2 from register copies
      const uint2 v0 = lds_read_tr16_b64(q), v1 = lds_read_tr16_b64(q + 256), v2 = lds_read_tr16_b64(q + 512);
      B[rr].f0 = make_uint4(v0.x, v0.y, v1.x, v1.y); B[rr].f2 = make_uint4(v0.y, v1.x, v1.y, v2.x);
#else
      const uint2 v0 = lds_read_tr16_b64(q), v1 = lds_read_tr16_b64(q + 256), w0 = lds_read_tr16_b64(q + 128), w1 = lds_read_tr16_b64(q + 384);
      B[rr].f0 = make_uint4(v0.x, v0.y, v1.x, v1.y); B[rr].f2 = make_uint4(w0.x, w0.y, w1.x, w1.y);
#endif
    };
    auto mk1 = [&](int rr) {
      const uint4 f0 = B[rr].f0;
      B[rr].f1 = make_uint4((f0.x >> 16) | (f0.y << 16), (f0.y >> 16) | (f0.z << 16), (f0.z >> 16) | (f0.w << 16), (f0.w >> 16) | (B[rr].f2.w << 16));
    };
    auto rdA = [&](int r) {
      const uint2 v0 = lds_read_tr16_b64(ap + r * (TX * 64)), v1 = lds_read_tr16_b64(ap + r * (TX * 64) + 256);
      A[r] = make_uint4(v0.x, v0.y, v1.x, v1.y);
    };
    auto mm = [&](int r) {
#pragma unroll
      for (int dy = 0; dy < (WT_ABL == 3 ? 1 : 3); ++dy) {
        acc[dy * 3 + 0] = mfma_lp<F16>(A[r], B[r + dy].f0, acc[dy * 3 + 0]);
        acc[dy * 3 + 1] = mfma_lp<F16>(A[r], B[r + dy].f1, acc[dy * 3 + 1]);
        acc[dy * 3 + 2] = mfma_lp<F16>(A[r], B[r + dy].f2, acc[dy * 3 + 2]);
      }
    };
    __syncthreads();                                               // the column's prologue
    set_plane(z0);
    rdB(0); rdB(1); rdB(2); rdA(0);
    for (int z = z0; z < z1; ++z) {
      // rows 0..2 with the reads one row ahead; the step's barrier sits BEFORE the last row's MFMAs (their operands are in registers), so that
      // the first reads of the next step are in flight behind them
      rdB(3); rdA(1);
      SCHED_BARRIER();
      mk1(0); mk1(1); mk1(2); mm(0);
      SCHED_BARRIER();
      rdB(4); rdA(2);
      SCHED_BARRIER();
      mk1(3); mm(1);
      SCHED_BARRIER();
      rdB(5); rdA(3);
      SCHED_BARRIER();
      mk1(4); mm(2);
      SCHED_BARRIER();
      __syncthreads();                                             // every read of this step has landed: the staging waves may overwrite its oldest plane
      mk1(5);
      if (z + 1 < z1) { set_plane(z + 1); rdB(0); rdB(1); rdB(2); rdA(0); }
      SCHED_BARRIER();
      mm(3);
      SCHED_BARRIER();
    }
  }

  // ---- the two row halves through LDS, then the partial tiles: ws[pair][slab = column chunk][tap][32 co][32 ci] ----
  float* ex = lds_f;                                               // [dz][tap 9][r 16][lane 64]
  if (h == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) ex[((dz * 9 + t) * 16 + r) * 64 + lane] = acc[t][r];
  }
  __syncthreads();
  if (h == 0) {
    const size_t pairi = (size_t)cot * a.ciTiles + cit;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float* dst = a.ws + (((pairi * a.nslab + cg) * 27 + dz * 9 + t) * 1024);
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[t][r] + ex[((dz * 9 + t) * 16 + r) * 64 + lane];
    }
  }
}

struct WTPlan { int tilesY, tilesX, zchunks, zper, ncol, colsPer, nslab, ciTiles, coTiles, ok; size_t ws_bytes; };

// the calls this kernel takes: 16-bit tensors with operands of their own type (MI355_PREC_BF16 on bf16 tensors, MI355_PREC_F16 on fp16
// tensors), 3x3x3 stride 1 pad 1, plain or normalised + activated input, plain output, channel counts in multiples of 32, 16-byte
// aligned voxels, at least one full 8 x 16 tile per plane. MI355_WGRAD_LP_TR=0 (read once): never -- the A/B switch.
static WTPlan plan_wt(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WTPlan p; memset(&p, 0, sizeof(p));
  static const bool off = [] { const char* v = getenv("MI355_WGRAD_LP_TR"); return v && v[0] == '0'; }();
  if (off || !x || !dy || !d) return p;
  if (d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return p;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return p;
  if (d->precision != MI355_PREC_BF16 && d->precision != MI355_PREC_F16) return p;
  if (!act_is_lp16(x->dtype) || x->dtype != dy->dtype || !act_matches_precision(x->dtype, d->precision)) return p;
  if (x->n != dy->n || x->d != dy->d || x->h != dy->h || x->w != dy->w) return p;
  if (x->c % 32 || dy->c % 32 || x->ld % 8 || dy->ld % 8 || ((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & 15)) return p;
  if (x->h < 8 || x->w < 16) return p;
  if ((long long)x->h * x->w * x->ld > 0x7fffffffLL || (long long)dy->h * dy->w * dy->ld > 0x7fffffffLL) return p;
  p.tilesY = ceil_div(x->h, 8); p.tilesX = ceil_div(x->w, 16);
  p.ciTiles = x->c / 32; p.coTiles = dy->c / 32;
  const long long cols = (long long)x->n * p.tilesY * p.tilesX, pairs = (long long)p.ciTiles * p.coTiles;
  if (cols <= 0 || cols * pairs > 0x3fffffffLL) return p;
  int zch = (int)((256 + cols * pairs - 1) / (cols * pairs));      // one 6-wave workgroup per CU: at least one round of the chip
  const int maxch = x->d >= 8 ? x->d / 8 : 1;                      // >= 8 planes per chunk (a chunk stages 3 planes before its first MFMA)
  if (zch > maxch) zch = maxch;
  if (zch < 1) zch = 1;
  p.zper = ceil_div(x->d, zch);
  p.zchunks = ceil_div(x->d, p.zper);
  if (cols * p.zchunks * pairs > 0x3fffffffLL) return p;
  p.ncol = (int)(cols * p.zchunks);
  p.colsPer = 1;      // (2 for more than 1024 workgroups was measured: 512 -> 256 @16^3 x4 0.259 -> 0.338 ms -- short columns want the parallelism)
  { const char* e = getenv("MI355_WGRAD_LP_COLS"); if (e && atoi(e) > 0) p.colsPer = atoi(e); }
  p.nslab = ceil_div(p.ncol, p.colsPer);
  p.ws_bytes = (size_t)pairs * p.nslab * 27 * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}

int mi355_conv3d_wgrad_lp_tr_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) { return plan_wt(x, dy, d).ok; }

size_t mi355_conv3d_wgrad_lp_tr_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  const WTPlan p = plan_wt(x, dy, d);
  return p.ok ? p.ws_bytes : 0;
}

template <int INMODE, typename TA>
static int launch_wt(const WgradTArgs& a, unsigned grid, void* stream) {
  constexpr int lds = 3 * 9 * 16 * 64 * 4;                         // the final exchange (110 592 bytes) > ring + dy planes (62 464)
  static_assert(lds >= 4 * 10 * 18 * 64 + 2 * 8 * 16 * 64 + 256, "LDS");
  SET_MAX_DYN_LDS((conv3d_wgrad_lp_tr<INMODE, TA>), lds);
  LAUNCH((conv3d_wgrad_lp_tr<INMODE, TA>), dim3(grid), dim3(512), lds, stream, a);
  return LAUNCH_CHECK();
}

int mi355_conv3d_wgrad_lp_tr_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream) {
  const WTPlan p = plan_wt(x, dy, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < p.ws_bytes) return MI355_EWORKSPACE;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift)) return MI355_EINVAL;
  WgradTArgs a; memset(&a, 0, sizeof(a));
  a.x = x->p; a.xld = x->ld; a.dy = dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w; a.Cin = x->c; a.Cout = dy->c;
  a.tilesY = p.tilesY; a.tilesX = p.tilesX; a.zchunks = p.zchunks; a.zper = p.zper; a.ncol = p.ncol;
  a.colsPer = p.colsPer; a.nslab = p.nslab;
  a.ciTiles = p.ciTiles; a.coTiles = p.coTiles;
  const unsigned grid = (unsigned)(ceil_div(p.nslab, 8) * 8 * p.ciTiles * p.coTiles);
  int rc;
  if (x->dtype == MI355_ACT_BF16) rc = d->in_mode == MI355_IN_PLAIN ? launch_wt<MI355_IN_PLAIN, bf16_t>(a, grid, stream) : launch_wt<MI355_IN_AFFINE_ACT, bf16_t>(a, grid, stream);
  else rc = d->in_mode == MI355_IN_PLAIN ? launch_wt<MI355_IN_PLAIN, f16_t>(a, grid, stream) : launch_wt<MI355_IN_AFFINE_ACT, f16_t>(a, grid, stream);
  if (rc) return rc;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, 27, p.nslab, p.ciTiles, stream);
}

// ---- 1x1x1 weight gradient as a set of streams: dw[co][ci] = sum_v dy[v][co] * x[v][ci]  (reference: the shortcut / projection
// convolutions, unet3d/models/pytorch/classification/resnet.py:20-22, decoder.py:99-106) ----
// HBM-bound (one MFMA per 2-6 KB of input), so the kernel is a set of independent streams: every wave walks its own contiguous voxel range in
// chunks of CV voxels through a wave-private LDS ring filled by LDS-DMA (1 KB requests, R chunks in flight per wave, counted vmcnt waits, no
// barrier in the loop) and keeps all CIT x COT output tiles in registers.
//   16-bit tensors: [16 voxels][32 channels] tiles read with the transpose read above, v_mfma_f32_32x32x16. Products of two 16-bit values are
//     exact in fp32 and the input is plain: this IS the exact-fp32 weight gradient of the stored tensors in every precision mode.
//   fp32 tensors: [CV voxels][32 channels] tiles, lane (channel, voxel parity) reads one float per k-step (the two voxels of a k-step sit
//     32 banks apart), v_mfma_f32_32x32x2_f32: the arithmetic of conv3d_wgrad_mfma<1, 1>.
// What it replaces: conv3d_wgrad_mfma<1, 1> (one workgroup per (ci, co) tile pair and 256-voxel tile through a barrier per step: 1.4 TB/s on
// 16-bit tensors, 2.8 TB/s on fp32 ones). The four waves of a workgroup add their tiles through LDS (two rounds); one slab per workgroup for
// the reduce of conv3d_wgrad.hip.
__device__ const float wlp_zero16[4] = {0.f, 0.f, 0.f, 0.f};

struct WgradK1Args {
  const void* x; int xld;
  const void* dy; int dyld;
  float* ws;
  long long V, nchunks;          // voxels, CV-voxel chunks
  int chunksPer, nslab;          // chunks per wave; workgroups = slabs
};

template <int CIT, int COT, int R, int CV, typename TA>
__global__ __launch_bounds__(256) void conv3d_wgrad_k1_stream(WgradK1Args a) {
  constexpr bool F32 = std::is_same<TA, float>::value, F16 = std::is_same<TA, f16_t>::value;
  constexpr int EPL = 16 / (int)sizeof(TA), LPV = 32 / EPL, VPI = 64 / LPV;      // elements per lane and request, lanes per voxel, voxels per request
  constexpr int SUB = CV / VPI;                                                    // requests per tile
  constexpr int TILEB = CV * 32 * (int)sizeof(TA);
  constexpr int TI = CIT + COT, CH = TI * TILEB, T = CIT * COT, NREQ = TI * SUB;
  static_assert(CV % VPI == 0 && (R - 1) * NREQ <= 63 && (F32 || CV == 16), "chunk");
  DYN_LDS(lds_f);
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6);
  float* const ring = lds_f + wave * (R * CH / 4);
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long c_begin = gw * a.chunksPer;
  const long long c_end = c_begin + a.chunksPer < a.nchunks ? c_begin + a.chunksPer : a.nchunks;
  const TA* const xg = reinterpret_cast<const TA*>(a.x);
  const TA* const dyg = reinterpret_cast<const TA*>(a.dy);
  const int lv = lane / LPV, lq = lane % LPV;                     // the lane's part of a request: voxel, 16-byte run
  // chunk -> ring slot: x tiles then dy tiles; voxels past the tensor: x re-reads the last voxel (finite), dy fetches zeros
  auto request = [&](long long chunk, int slot) {
#pragma unroll
    for (int sidx = 0; sidx < SUB; ++sidx) {
      long long v = chunk * CV + sidx * VPI + lv;
      const bool ok = v < a.V;
      if (!ok) v = a.V - 1;
      const TA* xp = xg + (size_t)v * a.xld + EPL * lq;
      const TA* dp = dyg + (size_t)v * a.dyld + EPL * lq;
#pragma unroll
      for (int t = 0; t < CIT; ++t) glds16(xp + 32 * t, ring + (slot * CH + t * TILEB + sidx * 1024) / 4);
#pragma unroll
      for (int t = 0; t < COT; ++t)
        glds16(ok ? reinterpret_cast<const void*>(dp + 32 * t) : reinterpret_cast<const void*>(wlp_zero16),
               ring + (slot * CH + (CIT + t) * TILEB + sidx * 1024) / 4);
    }
  };
  const int g = lane >> 4, r16 = lane & 15, half = lane >> 5, li = lane & 31;
  const unsigned lb = F32 ? (unsigned)(half * 128 + 4 * li) : (unsigned)((8 * (g >> 1) + (r16 >> 2)) * 64 + 32 * (g & 1) + 8 * (r16 & 3));
  f32x16 acc[COT][CIT];
#pragma unroll
  for (int i = 0; i < COT; ++i)
#pragma unroll
    for (int j = 0; j < CIT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
  for (int k = 0; k < R - 1; ++k) request(c_begin + k, k);        // (chunks past the end are requested too: the counts below stay fixed)
  int slot = 0;
  for (long long c = c_begin; c < c_end; ++c) {
    request(c + R - 1, slot == 0 ? R - 1 : slot - 1);             // the slot read in the previous step
    COMPILER_FENCE();
    WAIT_VMCNT_LGKM0((R - 1) * NREQ);                             // chunk c has landed
    COMPILER_FENCE();
    const unsigned char* base = reinterpret_cast<const unsigned char*>(ring) + slot * CH + lb;
    if constexpr (F32) {
#pragma unroll
      for (int ks = 0; ks < CV / 2; ++ks) {
        float A[COT], B[CIT];
#pragma unroll
        for (int t = 0; t < CIT; ++t) B[t] = *reinterpret_cast<const float*>(base + t * TILEB + ks * 256);
#pragma unroll
        for (int t = 0; t < COT; ++t) A[t] = *reinterpret_cast<const float*>(base + (CIT + t) * TILEB + ks * 256);
#pragma unroll
        for (int i = 0; i < COT; ++i)
#pragma unroll
          for (int j = 0; j < CIT; ++j) acc[i][j] = MFMA_32x32x2(A[i], B[j], acc[i][j]);
      }
    } else {
      uint4 A[COT], B[CIT];
#pragma unroll
      for (int t = 0; t < CIT; ++t) {
        const uint2 v0 = lds_read_tr16_b64(base + t * TILEB), v1 = lds_read_tr16_b64(base + t * TILEB + 256);
        B[t] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
#pragma unroll
      for (int t = 0; t < COT; ++t) {
        const uint2 v0 = lds_read_tr16_b64(base + (CIT + t) * TILEB), v1 = lds_read_tr16_b64(base + (CIT + t) * TILEB + 256);
        A[t] = make_uint4(v0.x, v0.y, v1.x, v1.y);
      }
#pragma unroll
      for (int i = 0; i < COT; ++i)
#pragma unroll
        for (int j = 0; j < CIT; ++j) acc[i][j] = mfma_lp<F16>(A[i], B[j], acc[i][j]);
    }
    slot = slot + 1 == R ? 0 : slot + 1;
  }
  COMPILER_FENCE();
  WAIT_VMCNT_LGKM0(0);                                            // the look-ahead requests must have landed before the ring is reused
  COMPILER_FENCE();
  __syncthreads();

  // ---- waves 2, 3 -> LDS -> waves 0, 1; wave 1 -> LDS -> wave 0 -> slab [pair = co tile * CIT + ci tile][slab][32 co][32 ci] ----
  float* ex = lds_f;                                              // [sender 2][tile T][r 16][lane 64]
  auto put = [&](int sd) {
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
      for (int j = 0; j < CIT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[((sd * T + i * CIT + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  };
  auto get = [&](int sd) {
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
      for (int j = 0; j < CIT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += ex[((sd * T + i * CIT + j) * 16 + r) * 64 + lane];
  };
  if (wave >= 2) put(wave - 2);
  __syncthreads();
  if (wave < 2) get(wave);
  __syncthreads();
  if (wave == 1) put(0);
  __syncthreads();
  if (wave == 0) {
    get(0);
#pragma unroll
    for (int i = 0; i < COT; ++i)
#pragma unroll
      for (int j = 0; j < CIT; ++j) {
        float* dst = a.ws + (((size_t)(i * CIT + j) * a.nslab + blockIdx.x) * 1024);
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[i][j][r];
      }
  }
}

struct WK1Plan { int cit, cot, cv, chunksPer, nslab, ok; long long V, nchunks; size_t ws_bytes; };

// the calls it takes: 1x1x1 stride 1, plain input and output, x and dy of one storage type (fp32, or 16-bit in any precision mode: see
// above), the (ci, co) tile counts of UNet3D's shortcut / projection convolutions (1 x 2, 2 x 1, 2 x 4, 4 x 2 tiles of 32), 16-byte aligned
// voxels. MI355_WGRAD_K1_STREAM=0 (read once): never -- the A/B switch.
static WK1Plan plan_wk1(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  WK1Plan p; memset(&p, 0, sizeof(p));
  static const bool off = [] { const char* v = getenv("MI355_WGRAD_K1_STREAM"); return v && v[0] == '0'; }();
  if (off || !x || !dy || !d) return p;
  if (d->kd != 1 || d->stride != 1 || d->pad != 0 || d->in_mode != MI355_IN_PLAIN || d->out_mode != MI355_OUT_PLAIN) return p;
  if (!act_dtype_ok(x) || x->dtype != dy->dtype) return p;
  const int epl = act_is_lp16(x->dtype) ? 8 : 4;
  if (x->n != dy->n || x->d != dy->d || x->h != dy->h || x->w != dy->w) return p;
  if (x->c % 32 || dy->c % 32 || x->ld % epl || dy->ld % epl || ((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & 15)) return p;
  p.cit = x->c / 32; p.cot = dy->c / 32;
  if (!((p.cit == 1 && p.cot == 2) || (p.cit == 2 && p.cot == 1) || (p.cit == 2 && p.cot == 4) || (p.cit == 4 && p.cot == 2))) return p;
  p.cv = (epl == 4 && p.cit + p.cot > 3) ? 8 : 16;                // fp32 tensors, six tiles: 8-voxel chunks (6 KB, like the others)
  p.V = (long long)x->n * x->d * x->h * x->w;
  if (p.V < 16) return p;
  p.nchunks = (p.V + p.cv - 1) / p.cv;
  long long per = (p.nchunks + 2047) / 2048;                      // ~512 workgroups of four waves: two per CU
  if (per < 8) per = 8;                                           // (small tensors: fewer workgroups rather than shorter streams)
  if (per > 0x7fffffffLL) return p;
  p.chunksPer = (int)per;
  const long long wgs = (p.nchunks + 4 * per - 1) / (4 * per);
  if (wgs > 0x7fffffffLL) return p;
  p.nslab = (int)wgs;
  p.ws_bytes = (size_t)p.cit * p.cot * p.nslab * 1024 * sizeof(float);
  p.ok = 1;
  return p;
}

int mi355_conv3d_wgrad_k1_lp_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) { return plan_wk1(x, dy, d).ok; }

size_t mi355_conv3d_wgrad_k1_lp_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  const WK1Plan p = plan_wk1(x, dy, d);
  return p.ok ? p.ws_bytes : 0;
}

template <int CIT, int COT, typename TA>
static int launch_wk1(const WgradK1Args& a, void* stream) {
  constexpr bool F32 = std::is_same<TA, float>::value;
  constexpr int TI = CIT + COT;
  constexpr int CV = (F32 && TI > 3) ? 8 : 16;
  constexpr int chunk = TI * CV * 32 * (int)sizeof(TA);           // 3 or 6 KB
  constexpr int R = 18432 / chunk;                                // 18 KB of ring per wave, 72 KB per workgroup: two workgroups per CU
  constexpr int ring = 4 * R * chunk, ex = 2 * CIT * COT * 4096;
  constexpr int lds = ring > ex ? ring : ex;
  SET_MAX_DYN_LDS((conv3d_wgrad_k1_stream<CIT, COT, R, CV, TA>), lds);
  LAUNCH((conv3d_wgrad_k1_stream<CIT, COT, R, CV, TA>), dim3((unsigned)a.nslab), dim3(256), lds, stream, a);
  return LAUNCH_CHECK();
}

template <typename TA>
static int launch_wk1_tiles(const WK1Plan& p, const WgradK1Args& a, void* stream) {
  if (p.cit == 1) return launch_wk1<1, 2, TA>(a, stream);
  if (p.cit == 4) return launch_wk1<4, 2, TA>(a, stream);
  return p.cot == 1 ? launch_wk1<2, 1, TA>(a, stream) : launch_wk1<2, 4, TA>(a, stream);
}

int mi355_conv3d_wgrad_k1_lp_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream) {
  const WK1Plan p = plan_wk1(x, dy, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < p.ws_bytes) return MI355_EWORKSPACE;
  WgradK1Args a; memset(&a, 0, sizeof(a));
  a.x = x->p; a.xld = x->ld; a.dy = dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.V = p.V; a.nchunks = p.nchunks; a.chunksPer = p.chunksPer; a.nslab = p.nslab;
  const int rc = x->dtype == MI355_ACT_BF16 ? launch_wk1_tiles<bf16_t>(p, a, stream)
               : x->dtype == MI355_ACT_F16 ? launch_wk1_tiles<f16_t>(p, a, stream) : launch_wk1_tiles<float>(p, a, stream);
  if (rc) return rc;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, dy->c, x->c, 1, p.nslab, p.cit, stream);
}
