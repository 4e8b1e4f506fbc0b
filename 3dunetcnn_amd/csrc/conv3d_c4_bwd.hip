// First-layer backward in ONE pass over dy (round 6): the weight gradient of the 4 -> 32 channel 3x3x3 conv (the "128^3 x 4ch" layer of
// BASELINE.json's north star: UNet3D encoder block 0 conv1, unet3d/models/pytorch/classification/myronenko.py:17-21 via resnet.py:12-17)
// AND the norm-backward sums of the GroupNorm in front of it (myronenko.py:9-15: dgamma = sum du * xhat, dbeta = sum du, du = dA * relu'(u),
// dA = the data gradient of the conv), without ever writing dA: the network input needs no gradient, so dA exists only to be reduced to
// those 2 x 4 numbers per sample.
//
// Before (conv3d_c4.hip): conv3d_c4_wgrad read dy once (0.37 ms at 128^3 x 2), conv3d_c4_dgrad read it again through 6 x 10 x 10 halo tiles
// (PMC: 1 476 MiB fetched for 537 MB of dy) and wrote dA (0.48-0.58 ms), gn_act_bwd read dA and x again. Here a 512-thread workgroup
// marches an 8 x 16 voxel column along z with a ring of four haloed dy planes (10 x 18 voxels x 32 channels, every plane of the column
// fetched once: 1.4x dy in total, the (y, x) halo; three are read, the fourth is being filled: one barrier per plane) and the activated
// x planes beside them in LDS. Per plane, everything on the matrix pipe:
//   * data gradient with v_mfma_f32_4x4x1_16b_f32 (16 blocks of 4 voxels x 4 input channels, K = one dy channel per instruction: exact
//     fp32 FMA chains at the fp32 matrix rate WITHOUT the 7/8 padding a 32-wide N tile would carry for 4 output channels): wave (voxel
//     half, channel quarter) owns 64 voxels x 8 dy channels; per tap and channel quad one ds_read_b128 of the voxel's quad (A) and one of
//     the weights of the lane's input channel (B: the 3 456 dgrad weights live in LDS, read as a 16-address broadcast) feed 4 MFMAs. The
//     first version ran this part on the vector ALU with wave-uniform weights from scalar loads: 1.22 ms -- every tap waited for its
//     scalar loads behind the LDS counter (profiles/r6_first_layer.txt);
//   * the four channel quarters meet in LDS; du = dA * act'(u) and the two sums per input channel stay in registers across the march;
//   * weight gradient beside it: wave (column group, K half) owns 32 of the 128 (tap, ci) columns (108 used) and half of the plane's 128
//     voxels: 32 v_mfma_f32_32x32x2_f32 per plane, one between the taps of the data gradient;
// and the accumulators leave once per workgroup: two weight-gradient slabs into the workspace of conv3d_c4_wgrad_reduce, one
// (sum du, sum du xhat) record per workgroup and input channel in the format of gn_fuse.h (mi355_gn_bwd_params finalises them).
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4 x 4 outer products. Lane l = 4 b + r supplies A[b][row r] and B[b][column r]; it holds
// D[b][rows 0..3][column r] in its 4 result registers.
#ifdef MI355_EMU
static inline f32x4 emu_mfma_4x4x1(float a, float b, f32x4 c) {
  emu::BlockState* bs = emu::g_bs;
  const int t = emu::flat_tid(), wbase = (t / 64) * 64, l = t % 64;
  bs->mfma_a[t] = a;
  emu::wave_barrier();
  for (int i = 0; i < 4; ++i) c[i] = fmaf(bs->mfma_a[wbase + (l & ~3) + i], b, c[i]);
  emu::wave_barrier();
  return c;
}
#define MFMA_4x4x1(a, b, c) emu_mfma_4x4x1(a, b, c)
#else
#define MFMA_4x4x1(a, b, c) __builtin_amdgcn_mfma_f32_4x4x1f32((a), (b), (c), 0, 0, 0)
#endif

struct C4BArgs {
  const float* x; int xld;               // network input, 4 channels (fp32)
  const float* dy; int dyld;             // gradient wrt the conv output, 32 channels (storage type TD of the kernel: fp32, bf16 or fp16)
  const float* wp;                       // dgrad pack of the conv weight (mi355_pack_conv_weight mode 1: [27][8][32][4], columns 0..3 used)
  float* ws;                             // weight-gradient slabs [2 x workgroup][32 co][128 (tap, ci) columns]
  float* part;                           // norm-backward records [n][B][4][2]
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* mean_rstd; int groups;
  int N, D, H, W, tilesY, tilesX, zchunks, zper;
};

void conv3d_c4_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int splits, void* stream);      // conv3d_c4.hip

// TD: storage type of dy (act_io.h). The arithmetic is exact fp32 on the stored values in every precision mode, as in the kernels this one
// replaces (the 4-channel first layer's weight gradient and data gradient never ran on the 16-bit pipe).
template <int INMODE, typename TD = float>
__global__ __launch_bounds__(512) MIN_WAVES_PER_SIMD(2) void conv3d_c4_bwd(C4BArgs a) {
  const TD* const ady = reinterpret_cast<const TD*>(a.dy);
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HV = HY * HX;
  constexpr int DYP = HV * 32, XP = HV * 4;              // floats per staged dy / x plane
  constexpr int DYU = (HV * 8 + 511) / 512;              // 16-byte units of a dy plane per thread
  DYN_LDS(lds);
  float* dyr = lds;                                      // ring of 4 dy planes: voxel hv at hv * 32, channel quad q at slot q ^ s(hx)
  float* xr = lds + 4 * DYP;                             // ring of 4 activated x planes (float4 per voxel, zero outside the image)
  float* wl = xr + 4 * XP;                               // dgrad weights [tap][quad][input channel 4][4 dy channels of the quad]
  float* cmb = wl + 27 * 8 * 16;                         // partial data gradients of channel quarters 1..3: [plane parity][voxel half][3][64 lanes][4]
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int zc = b % a.zchunks; b /= a.zchunks;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int n = b;
  const int z_begin = zc * a.zper, z_end = z_begin + a.zper < a.D ? z_begin + a.zper : a.D;
  // swizzle of the channel quads of a staged dy voxel by its column: 16 x-neighbours reading one quad hit 16 different bank groups
  auto swz = [](int hx) { return (hx ^ (hx >> 3)) & 7; };

  // ---- staging ----
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f), sl = make_float4(a.slope, a.slope, a.slope, a.slope);
  if (INMODE == MI355_IN_AFFINE_ACT) {
    sc = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * 4);
    sh = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * 4);
    if (a.in_slope) sl = *reinterpret_cast<const float4*>(a.in_slope);
  }
  for (int i = tid; i < 27 * 8 * 16; i += 512) wl[i] = a.wp[(size_t)(i >> 4) * 128 + (i & 15)];      // columns 0..3 of every (tap, quad)
  float4 dld[DYU], xld_;
  auto load_plane = [&](int z) {                         // global -> registers (clamped always-valid addresses; validity decided at the commit)
    const int zcl = z < 0 ? 0 : (z < a.D ? z : a.D - 1);
#pragma unroll
    for (int k = 0; k < DYU; ++k) {
      int u = tid + k * 512; if (u >= HV * 8) u = HV * 8 - 1;
      const int hv = u >> 3, q = u & 7;
      int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
      iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1); ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      dld[k] = ld4(ady + ((((size_t)n * a.D + zcl) * a.H + iy) * a.W + ix) * a.dyld + 4 * q);
    }
    {
      const int hv = tid < HV ? tid : HV - 1;
      int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
      iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1); ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      xld_ = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + zcl) * a.H + iy) * a.W + ix) * a.xld);
    }
  };
  auto commit_plane = [&](int z) {                       // registers -> ring slot z mod 4 (z >= -1)
    const int slot = (z + 4) & 3;
    const bool zin = z >= 0 && z < a.D;
#pragma unroll
    for (int k = 0; k < DYU; ++k) {
      const int u = tid + k * 512;
      if (u >= HV * 8) continue;
      const int hv = u >> 3, q = u & 7, hy = hv / HX, hx = hv % HX;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const bool ok = zin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      *reinterpret_cast<float4*>(dyr + slot * DYP + hv * 32 + ((q ^ swz(hx)) << 2)) = ok ? dld[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (tid < HV) {
      const int hy = tid / HX, hx = tid % HX;
      const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
      const bool ok = zin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      float4 v = xld_;
      if (INMODE == MI355_IN_AFFINE_ACT) {
        v.x = v.x * sc.x + sh.x; v.y = v.y * sc.y + sh.y; v.z = v.z * sc.z + sh.z; v.w = v.w * sc.w + sh.w;
        v.x = fmaxf(v.x, v.x * sl.x); v.y = fmaxf(v.y, v.y * sl.y); v.z = fmaxf(v.z, v.z * sl.z); v.w = fmaxf(v.w, v.w * sl.w);
      }
      *reinterpret_cast<float4*>(xr + slot * XP + tid * 4) = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // ---- roles ----
  // data gradient: wave = (voxel half vh: rows 4 vh .. 4 vh + 3, channel quarter cq: dy quads 2 cq, 2 cq + 1). As the A operand lane l is
  // voxel (row l >> 4, column l & 15) of the half; as the B operand and in the result it is input channel cj = l & 3, and its 4 result
  // registers are the voxels 4 (l >> 2) + 0..3 of the half (one row: 16 divides by 4)
  const int vh = wave & 1, cq = wave >> 1, ly = lane >> 4, lx = lane & 15, cj = lane & 3;
  const int vy = 4 * vh + ly;
  const int ry = 4 * vh + (lane >> 4), rx0 = (lane >> 2 & 3) * 4;      // the result voxels: row ry, columns rx0 .. rx0 + 3
  const bool rin = ty0 + ry < a.H;
  // weight gradient: wave = (column group cg: (tap, ci) columns 32 cg .. 32 cg + 31, K half kh: voxel pairs 32 kh .. 32 kh + 31 of the plane);
  // A = dy (32 channels x voxel pairs), B = activated x at voxel + tap
  const int cg = wave & 3, kh = wave >> 2;
  const int jcol = cg * 32 + li;
  const int tapc = jcol >> 2 < 27 ? jcol >> 2 : 26, cic = jcol & 3;
  const int dzc = tapc / 9, boff = (((tapc / 3) % 3) * HX + tapc % 3) * 4 + cic;
  f32x16 wacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
  float sdu = 0.f, sdx = 0.f;                            // sum du, sum du * xhat of input channel cj over this lane's voxels (cq == 0 waves)
  const float scj = cj == 0 ? sc.x : cj == 1 ? sc.y : cj == 2 ? sc.z : sc.w, shj = cj == 0 ? sh.x : cj == 1 ? sh.y : cj == 2 ? sh.z : sh.w;
  const float slj = cj == 0 ? sl.x : cj == 1 ? sl.y : cj == 2 ? sl.z : sl.w;
  const int grp = cj / (4 / a.groups);
  const float gmean = a.mean_rstd[((size_t)n * a.groups + grp) * 2], grstd = a.mean_rstd[((size_t)n * a.groups + grp) * 2 + 1];

  // ---- prologue: planes z_begin - 1, z_begin, z_begin + 1 staged; plane z_begin + 2 in registers ----
  // Four ring slots, ONE barrier per plane: plane z + 2 (loaded during plane z - 1) is committed at the top of plane z into the slot plane
  // z - 2 left at the previous barrier, while nobody reads it; the barrier at the end of plane z publishes it for plane z + 1.
  load_plane(z_begin - 1); commit_plane(z_begin - 1);
  load_plane(z_begin); commit_plane(z_begin);
  load_plane(z_begin + 1); commit_plane(z_begin + 1);
  load_plane(z_begin + 2);
  __syncthreads();

  for (int z = z_begin; z < z_end; ++z) {
    commit_plane(z + 2);
    load_plane(z + 3);                                   // in flight during the plane's arithmetic
    // raw input of this lane's result voxels, channel cj (the activation mask and xhat of the sums)
    float xc[4] = {0.f, 0.f, 0.f, 0.f};
    if (cq == 0 && rin) {
      const float* xp = a.x + ((((size_t)n * a.D + z) * a.H + ty0 + ry) * a.W + tx0 + rx0) * a.xld + cj;
#pragma unroll
      for (int i = 0; i < 4; ++i) if (tx0 + rx0 + i < a.W) xc[i] = xp[(size_t)i * a.xld];
    }
    const int s0 = (z + 3) & 3, s1 = z & 3, s2 = (z + 1) & 3;      // ring slots of planes z - 1, z, z + 1
    float* cmz = cmb + (z & 1) * (2 * 3 * 64 * 4);       // (double-buffered: the next plane's partials may be written before a slow reader is done)
    const float* dyc = dyr + s1 * DYP;
    const float* xb = xr + (dzc == 0 ? s0 : (dzc == 1 ? s1 : s2)) * XP + boff;      // this lane's B operand plane + tap offset
    // Software pipeline, written out (left to itself hipcc reads the two operands of a channel quad, waits for the LDS and issues the four
    // dependent MFMAs, quad after quad: 31 k cycles per plane against 7.6 k of matrix time): the operands of tap t + 1 -- two dy quads, two
    // weight quads, the two scalars of the weight gradient's k-step -- are requested before the MFMAs of tap t issue; two accumulators
    // (one per quad) halve the dependent chains. All 27 taps unrolled: every tap offset is an immediate.
    f32x4 dacc0 = {0.f, 0.f, 0.f, 0.f}, dacc1 = {0.f, 0.f, 0.f, 0.f};
    struct TapOps { float4 d0, d1, w0, w1; float av, bv; };
    const float* pls[3] = {dyr + s0 * DYP, dyr + s1 * DYP, dyr + s2 * DYP};
    const int hbase = vy * HX + lx;
    // the weight gradient's k-step t of this wave's K half: voxel pair 32 kh + t -> row 4 kh + (t >> 3), column (2 t & 15) + half
    const float* dyA = dyc + ((4 * kh + 1) * HX + half + 1) * 32 + (li & 3);
    const float* xbB = xb + ((4 * kh) * HX + half) * 4;
    auto tap_read = [&](int tap, TapOps& o) {            // tap: compile-time after unrolling
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      const int hx = lx + dx, sw = swz(hx);
      const float* pv = pls[dz] + (hbase + dy * HX + dx) * 32;
      o.d0 = *reinterpret_cast<const float4*>(pv + (((2 * cq) ^ sw) << 2));
      o.d1 = *reinterpret_cast<const float4*>(pv + (((2 * cq + 1) ^ sw) << 2));
      const float* wt = wl + (tap * 8 + 2 * cq) * 16 + cj * 4;
      o.w0 = *reinterpret_cast<const float4*>(wt);
      o.w1 = *reinterpret_cast<const float4*>(wt + 16);
    };
    auto wg_read = [&](int t, TapOps& o) {               // t = 0..31
      const int wyo = t >> 3, wxo = (2 * t) & 15;
      const int wx1 = wxo + half + 1;
      o.av = dyA[(wyo * HX + wxo) * 32 + (((li >> 2) ^ swz(wx1)) << 2)];
      o.bv = xbB[(wyo * HX + wxo) * 4];
    };
    auto tap_mfma = [&](const TapOps& o) {
      dacc0 = MFMA_4x4x1(o.d0.x, o.w0.x, dacc0); dacc1 = MFMA_4x4x1(o.d1.x, o.w1.x, dacc1);
      dacc0 = MFMA_4x4x1(o.d0.y, o.w0.y, dacc0); dacc1 = MFMA_4x4x1(o.d1.y, o.w1.y, dacc1);
      wacc = MFMA_32x32x2(o.av, o.bv, wacc);
      dacc0 = MFMA_4x4x1(o.d0.z, o.w0.z, dacc0); dacc1 = MFMA_4x4x1(o.d1.z, o.w1.z, dacc1);
      dacc0 = MFMA_4x4x1(o.d0.w, o.w0.w, dacc0); dacc1 = MFMA_4x4x1(o.d1.w, o.w1.w, dacc1);
    };
    TapOps oa, ob;
    tap_read(0, oa); wg_read(0, oa);
    static_for<0, 27>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      TapOps& cur = (t & 1) ? ob : oa;
      TapOps& nxt = (t & 1) ? oa : ob;
      if constexpr (t + 1 < 27) tap_read(t + 1, nxt);
      wg_read(t + 1, nxt);                               // (t + 1 = 27: the first of the five k-steps after the taps)
      SCHED_BARRIER();
      tap_mfma(cur);
      SCHED_BARRIER();
    });
    {                                                    // k-steps 27..31 of the weight gradient (step 27 was read into ob by tap 26)
      TapOps& c27 = ob;
      wacc = MFMA_32x32x2(c27.av, c27.bv, wacc);
      static_for<28, 32>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        TapOps o; wg_read(t, o);
        wacc = MFMA_32x32x2(o.av, o.bv, wacc);
      });
    }
    f32x4 dacc = dacc0 + dacc1;
    // the four channel quarters meet: quarters 1..3 through LDS
    if (cq != 0) *reinterpret_cast<float4*>(cmz + (((vh * 3 + cq - 1) * 64) + lane) * 4) = make_float4(dacc[0], dacc[1], dacc[2], dacc[3]);
    __syncthreads();                                     // the plane's only barrier: partials and plane z + 2 published, plane z - 1 released
    if (cq == 0 && rin) {
      float g[4] = {dacc[0], dacc[1], dacc[2], dacc[3]};
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float4 up = *reinterpret_cast<const float4*>(cmz + (((vh * 3 + k) * 64) + lane) * 4);
        g[0] += up.x; g[1] += up.y; g[2] += up.z; g[3] += up.w;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (tx0 + rx0 + i >= a.W) continue;
        float du = g[i];
        if (INMODE == MI355_IN_AFFINE_ACT) { const float u = xc[i] * scj + shj; du = u > 0.f ? g[i] : g[i] * slj; }
        sdu += du; sdx += du * ((xc[i] - gmean) * grstd);
      }
    }
  }

  // ---- outputs ----
  // weight-gradient slab of this wave set (K half kh): [32 co][128 columns] (coalesced: the 32 lanes of a half write 128 contiguous bytes)
  float* dst = a.ws + ((size_t)blockIdx.x * 2 + kh) * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    dst[row * 128 + jcol] = wacc[r];
  }
  // norm-backward record of this workgroup: lanes of equal input channel (l & 3) of the cq == 0 waves, fixed order (xor shuffles over the
  // 16 lanes of a channel, then the two voxel halves through LDS)
  float vals[2] = {sdu, sdx};
#pragma unroll
  for (int step = 4; step < 64; step <<= 1) {
    const bool upper = lane & step;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float o = __shfl_xor(vals[k], step);
      vals[k] = upper ? o + vals[k] : vals[k] + o;
    }
  }
  __syncthreads();
  if (cq == 0 && lane < 4) { cmb[(vh * 4 + lane) * 2] = vals[0]; cmb[(vh * 4 + lane) * 2 + 1] = vals[1]; }
  __syncthreads();
  if (tid < 8) {
    const int c = tid >> 1, k = tid & 1;                 // k = 0: sum du, 1: sum du * xhat
    const float r = cmb[c * 2 + k] + cmb[(4 + c) * 2 + k];
    const size_t B = (size_t)a.tilesY * a.tilesX * a.zchunks;
    a.part[(((size_t)n * B + (blockIdx.x % B)) * 4 + c) * 2 + k] = r;
  }
}

static int c4b_plan(const mi355_act* x, C4BArgs& a) {
  a.tilesY = ceil_div(x->h, 8); a.tilesX = ceil_div(x->w, 16);
  const long long cols = (long long)x->n * a.tilesY * a.tilesX;
  if (cols <= 0 || cols > 0x7fffffffLL) return 0;
  // enough z chunks for ~512 workgroups (one per CU, two rounds), at least 8 planes each (a chunk re-reads two halo planes)
  int zch = (int)((512 + cols - 1) / cols);
  const int maxch = x->d >= 8 ? x->d / 8 : 1;
  if (zch > maxch) zch = maxch;
  if (zch < 1) zch = 1;
  a.zper = ceil_div(x->d, zch);
  a.zchunks = ceil_div(x->d, a.zper);
  return cols * a.zchunks <= 0x7fffffffLL;
}

static int c4b_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  if (!x || !dy || !d || !x->p || !dy->p) return MI355_EINVAL;
  if (x->c != 4 || dy->c != 32 || d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (x->dtype != MI355_ACT_F32 || !act_dtype_ok(dy)) return MI355_EUNSUPPORTED;      // (any precision mode: this layer's backward is exact fp32 in all of them)
  if (x->n != dy->n || x->d != dy->d || x->h != dy->h || x->w != dy->w) return MI355_EINVAL;
  if (x->ld % 4 || dy->ld % 4 || x->ld < 4 || dy->ld < 32 || ((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & act_align_mask(dy->dtype))) return MI355_EINVAL;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift || !(d->act_slope >= 0.f && d->act_slope <= 1.f))) return MI355_EINVAL;
  return MI355_OK;
}

extern "C" int mi355_conv3d_c4_bwd_supported(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  return c4b_ok(x, dy, d) == MI355_OK;
}

extern "C" int32_t mi355_conv3d_c4_bwd_blocks(const mi355_act* x) {
  C4BArgs a; memset(&a, 0, sizeof(a));
  if (!x || !c4b_plan(x, a)) return 0;
  return a.tilesY * a.tilesX * a.zchunks;
}

extern "C" size_t mi355_conv3d_c4_bwd_workspace(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  C4BArgs a; memset(&a, 0, sizeof(a));
  if (c4b_ok(x, dy, d) || !c4b_plan(x, a)) return 0;
  return (size_t)x->n * a.tilesY * a.tilesX * a.zchunks * 2 * 4096 * sizeof(float);
}

// dw: OIDHW [32][4][3][3][3]; wp_dgrad: mi355_pack_conv_weight(mode 1) of the same weight (packed roles out 4, in 32); desc: the conv's
// descriptor (in_mode / in_scale / in_shift / act_slope = the norm prologue of the forward); mean_rstd, groups: the statistics of that
// norm (mi355_gn_stats); partials_out: [n][mi355_conv3d_c4_bwd_blocks(x)][4][2] records for mi355_gn_bwd_params.
extern "C" int mi355_conv3d_c4_bwd(const mi355_act* x, const mi355_act* dy, const float* wp_dgrad, float* dw, const mi355_conv_desc* d,
                                   const float* mean_rstd, int32_t groups, float* partials_out, void* ws, size_t ws_bytes, void* stream) {
  { const int rc = c4b_ok(x, dy, d); if (rc) return rc; }
  if (!wp_dgrad || !dw || !mean_rstd || !partials_out || !ws || groups <= 0 || 4 % groups) return MI355_EINVAL;
  C4BArgs a; memset(&a, 0, sizeof(a));
  if (!c4b_plan(x, a)) return MI355_EINVAL;
  const long long wgs = (long long)x->n * a.tilesY * a.tilesX * a.zchunks;
  if (ws_bytes < (size_t)wgs * 2 * 4096 * sizeof(float)) return MI355_EWORKSPACE;
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.wp = wp_dgrad; a.ws = (float*)ws; a.part = partials_out;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope; a.mean_rstd = mean_rstd; a.groups = groups;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w;
  const int lds_bytes = (4 * 180 * 32 + 4 * 180 * 4 + 27 * 8 * 16 + 2 * 2 * 3 * 64 * 4) * (int)sizeof(float);      // 129 792: one 8-wave workgroup per CU
  ACT_TYPED(dy->dtype, TD,
            if (d->in_mode == MI355_IN_PLAIN) {
              SET_MAX_DYN_LDS((conv3d_c4_bwd<MI355_IN_PLAIN, TD>), lds_bytes);
              LAUNCH((conv3d_c4_bwd<MI355_IN_PLAIN, TD>), dim3((unsigned)wgs), dim3(512), lds_bytes, stream, a);
            } else {
              SET_MAX_DYN_LDS((conv3d_c4_bwd<MI355_IN_AFFINE_ACT, TD>), lds_bytes);
              LAUNCH((conv3d_c4_bwd<MI355_IN_AFFINE_ACT, TD>), dim3((unsigned)wgs), dim3(512), lds_bytes, stream, a);
            });
  const int rc = LAUNCH_CHECK(); if (rc) return rc;
  conv3d_c4_wgrad_reduce_launch((const float*)ws, dw, 32, (int)wgs * 2, stream);
  return LAUNCH_CHECK();
}
