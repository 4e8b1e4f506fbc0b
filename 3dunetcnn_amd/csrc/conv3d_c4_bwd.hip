// First-layer backward in ONE pass over dy (round 6): the weight gradient of the 4 -> 32 channel 3x3x3 conv (the "128^3 x 4ch" layer of
// BASELINE.json's north star: UNet3D encoder block 0 conv1, unet3d/models/pytorch/classification/myronenko.py:17-21 via resnet.py:12-17)
// AND the norm-backward sums of the GroupNorm in front of it (myronenko.py:9-15: dgamma = sum du * xhat, dbeta = sum du, du = dA * relu'(u),
// dA = the data gradient of the conv), without ever writing dA: the network input needs no gradient, so dA exists only to be reduced to
// those 2 x 4 numbers per sample.
//
// Before (conv3d_c4.hip): conv3d_c4_wgrad read dy once (0.38 ms at 128^3 x 2), conv3d_c4_dgrad read it again through 6 x 10 x 10 halo tiles
// (PMC: 1 476 MiB fetched for 537 MB of dy) and wrote dA (0.50 ms), gn_act_bwd read dA and x again. Here a workgroup marches an 8 x 16 voxel
// column along z with a ring of three haloed dy planes (10 x 18 voxels x 32 channels, every plane of the column fetched once: 1.4x dy in
// total, the (y, x) halo) and three activated x planes in LDS. Per plane:
//   * data gradient on the vector ALU (exact fp32 FMA chains; on the 32-wide MFMA N tile 7/8 of the matrix work would be padding): wave
//     (voxel half, channel half) owns 64 voxels x 16 dy channels, 27 taps x 4 ds_read_b128 x 16 FMAs with wave-uniform (scalar) weights
//     from the dgrad pack; the two channel halves meet in LDS, du and the two sums stay in registers across the march;
//   * weight gradient on the matrix pipe beside it: wave w owns the (tap, ci) columns 32 w .. 32 w + 31 (108 used), K = the 128 voxels of
//     the plane: 64 v_mfma_f32_32x32x2_f32, interleaved with the taps of the data gradient (two or three per tap) -- the two halves of
//     the backward run on different pipes of the same SIMD from the same staged bytes;
// and the accumulators leave once per workgroup: the weight-gradient slab into the workspace of conv3d_c4_wgrad_reduce, one
// (sum du, sum du xhat) record per workgroup and input channel in the format of gn_fuse.h (mi355_gn_bwd_params finalises them).
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"

struct C4BArgs {
  const float* x; int xld;               // network input, 4 channels (fp32)
  const float* dy; int dyld;             // gradient wrt the conv output, 32 channels
  const float* wp;                       // dgrad pack of the conv weight (mi355_pack_conv_weight mode 1: [27][8][32][4], columns 0..3 used)
  float* ws;                             // weight-gradient slabs [workgroup][32 co][128 (tap, ci) columns]
  float* part;                           // norm-backward records [n][B][4][2]
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* mean_rstd; int groups;
  int N, D, H, W, tilesY, tilesX, zchunks, zper;
};

void conv3d_c4_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int splits, void* stream);      // conv3d_c4.hip

template <int INMODE>
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD(2) void conv3d_c4_bwd(C4BArgs a) {
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HV = HY * HX;
  constexpr int DYP = HV * 32, XP = HV * 4;              // floats per staged dy / x plane
  constexpr int DYU = (HV * 8 + 255) / 256;              // 16-byte units of a dy plane per thread
  DYN_LDS(lds);
  float* dyr = lds;                                      // ring of 3 dy planes: voxel hv at hv * 32, channel quad q at slot q ^ s(hx)
  float* xr = lds + 3 * DYP;                             // ring of 3 activated x planes (float4 per voxel, zero outside the image)
  float* cmb = xr + 3 * XP;                              // the upper channel half's partial data gradient: [128 voxels][4]
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
  float4 dld[DYU], xld_;
  auto load_plane = [&](int z) {                         // global -> registers (clamped always-valid addresses; validity decided at the commit)
    const int zcl = z < 0 ? 0 : (z < a.D ? z : a.D - 1);
#pragma unroll
    for (int k = 0; k < DYU; ++k) {
      int u = tid + k * 256; if (u >= HV * 8) u = HV * 8 - 1;
      const int hv = u >> 3, q = u & 7;
      int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
      iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1); ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      dld[k] = *reinterpret_cast<const float4*>(a.dy + ((((size_t)n * a.D + zcl) * a.H + iy) * a.W + ix) * a.dyld + 4 * q);
    }
    {
      const int hv = tid < HV ? tid : HV - 1;
      int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
      iy = iy < 0 ? 0 : (iy < a.H ? iy : a.H - 1); ix = ix < 0 ? 0 : (ix < a.W ? ix : a.W - 1);
      xld_ = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + zcl) * a.H + iy) * a.W + ix) * a.xld);
    }
  };
  auto commit_plane = [&](int z) {                       // registers -> ring slot z mod 3 (z >= -1)
    const int slot = (z + 3) % 3;
    const bool zin = z >= 0 && z < a.D;
#pragma unroll
    for (int k = 0; k < DYU; ++k) {
      const int u = tid + k * 256;
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
  // data gradient: wave = (voxel half vh: rows 4 vh .. 4 vh + 3, channel half ch: dy quads 4 ch .. 4 ch + 3), lane = (row ly, column lx)
  const int vh = wave >> 1, ch = wave & 1, ly = lane >> 4, lx = lane & 15;
  const int vy = 4 * vh + ly;
  const int vox = vy * TX + lx;                          // voxel of the plane tile this lane owns
  const bool vin = ty0 + vy < a.H && tx0 + lx < a.W;
  // weight gradient: wave w owns (tap, ci) columns 32 w .. 32 w + 31; A = dy (32 channels x voxel pairs), B = activated x at voxel + tap
  const int jcol = wave * 32 + li;
  const int tapc = jcol >> 2 < 27 ? jcol >> 2 : 26, cic = jcol & 3;
  const int dzc = tapc / 9, boff = (((tapc / 3) % 3) * HX + tapc % 3) * 4 + cic;
  f32x16 wacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
  float sdu[4] = {0.f, 0.f, 0.f, 0.f}, sdx[4] = {0.f, 0.f, 0.f, 0.f};      // sum du, sum du * xhat of this lane's voxels (ch == 0 waves)
  float gmean[4], grstd[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int grp = c / (4 / a.groups);
    gmean[c] = a.mean_rstd[((size_t)n * a.groups + grp) * 2]; grstd[c] = a.mean_rstd[((size_t)n * a.groups + grp) * 2 + 1];
  }

  // ---- prologue: planes z_begin - 1, z_begin, z_begin + 1 ----
  load_plane(z_begin - 1); commit_plane(z_begin - 1);
  load_plane(z_begin); commit_plane(z_begin);
  load_plane(z_begin + 1); commit_plane(z_begin + 1);
  __syncthreads();

  for (int z = z_begin; z < z_end; ++z) {
    load_plane(z + 2);                                   // in flight during the plane's arithmetic
    // raw input of this lane's voxel (the activation mask and xhat of the sums)
    float4 xc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ch == 0 && vin) xc = *reinterpret_cast<const float4*>(a.x + ((((size_t)n * a.D + z) * a.H + ty0 + vy) * a.W + tx0 + lx) * a.xld);
    const int s0 = (z + 2) % 3, s1 = z % 3, s2 = (z + 1) % 3;      // ring slots of planes z - 1, z, z + 1
    const float* dyc = dyr + s1 * DYP;
    const float* xb = xr + (dzc == 0 ? s0 : (dzc == 1 ? s1 : s2)) * XP + boff;      // this lane's B operand plane + tap offset
    float acc[4][2];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o][0] = acc[o][1] = 0.f;
    // (a rolled loop: unrolled, hipcc hoists the 1 728 scalar weight loads and the window reads of all taps -- 256 registers and spills)
#pragma unroll 1
    for (int tap = 0; tap < 27; ++tap) {
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;      // wave-uniform (scalar) arithmetic
      const float* pl = dyr + (dz == 0 ? s0 : (dz == 1 ? s1 : s2)) * DYP;
      const int hx = lx + dx;
      const int hv = (vy + dy) * HX + hx, sw = swz(hx);
      float4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(pl + hv * 32 + (((4 * ch + q) ^ sw) << 2));
      // the weight gradient's MFMAs of this plane ride between the taps: two per tap (k-steps 2 tap, 2 tap + 1), the last ten after the loop
      float av[2], bv[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int v2 = 2 * (2 * tap + e) + half;         // voxel of the plane tile: row v2 >> 4, column v2 & 15
        const int wy = v2 >> 4, wx = v2 & 15;
        av[e] = dyc[((wy + 1) * HX + wx + 1) * 32 + (((li >> 2) ^ swz(wx + 1)) << 2) + (li & 3)];
        bv[e] = xb[(wy * HX + wx) * 4];
      }
      wacc = MFMA_32x32x2(av[0], bv[0], wacc);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* wq = a.wp + ((size_t)tap * 8 + 4 * ch + q) * 128;      // wave-uniform: scalar loads
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          acc[o][0] = fmaf(v[q].x, wq[o * 4 + 0], acc[o][0]); acc[o][1] = fmaf(v[q].y, wq[o * 4 + 1], acc[o][1]);
          acc[o][0] = fmaf(v[q].z, wq[o * 4 + 2], acc[o][0]); acc[o][1] = fmaf(v[q].w, wq[o * 4 + 3], acc[o][1]);
        }
        if (q == 1) wacc = MFMA_32x32x2(av[1], bv[1], wacc);
      }
    }
#pragma unroll
    for (int ks = 54; ks < 64; ++ks) {
      const int v2 = 2 * ks + half;
      const int wy = v2 >> 4, wx = v2 & 15;
      wacc = MFMA_32x32x2(dyc[((wy + 1) * HX + wx + 1) * 32 + (((li >> 2) ^ swz(wx + 1)) << 2) + (li & 3)], xb[(wy * HX + wx) * 4], wacc);
    }
    // the two channel halves meet: the upper one through LDS
    float dA[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) dA[o] = acc[o][0] + acc[o][1];
    if (ch == 1) *reinterpret_cast<float4*>(cmb + vox * 4) = make_float4(dA[0], dA[1], dA[2], dA[3]);
    __syncthreads();                                     // also: every wave is done with plane z - 1 and the x ring
    if (ch == 0 && vin) {
      const float4 up = *reinterpret_cast<const float4*>(cmb + vox * 4);
      const float xv[4] = {xc.x, xc.y, xc.z, xc.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w}, slv[4] = {sl.x, sl.y, sl.z, sl.w};
      const float upv[4] = {up.x, up.y, up.z, up.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float g = dA[c] + upv[c];
        float du = g;
        if (INMODE == MI355_IN_AFFINE_ACT) { const float u = xv[c] * scv[c] + shv[c]; du = u > 0.f ? g : g * slv[c]; }
        sdu[c] += du; sdx[c] += du * ((xv[c] - gmean[c]) * grstd[c]);
      }
    }
    commit_plane(z + 2);                                 // into the slot of plane z - 1
    __syncthreads();
  }

  // ---- outputs ----
  // weight-gradient slab of this workgroup: [32 co][128 columns] (coalesced: the 32 lanes of a half write 128 contiguous bytes)
  float* dst = a.ws + (size_t)blockIdx.x * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    dst[row * 128 + jcol] = wacc[r];
  }
  // norm-backward record of this workgroup: the 128 lanes of the ch == 0 waves, fixed order (xor shuffles, then the two waves through LDS)
  float vals[8] = {sdu[0], sdu[1], sdu[2], sdu[3], sdx[0], sdx[1], sdx[2], sdx[3]};
#pragma unroll
  for (int step = 1; step < 64; step <<= 1) {
    const bool upper = lane & step;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float o = __shfl_xor(vals[k], step);
      vals[k] = upper ? o + vals[k] : vals[k] + o;
    }
  }
  if (ch == 0 && lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) cmb[vh * 8 + k] = vals[k];
  }
  __syncthreads();
  if (tid < 8) {
    const float r = cmb[tid] + cmb[8 + tid];
    const int c = tid & 3, k = tid >> 2;                 // k = 0: sum du, 1: sum du * xhat
    const size_t B = (size_t)a.tilesY * a.tilesX * a.zchunks;
    a.part[(((size_t)n * B + (blockIdx.x % B)) * 4 + c) * 2 + k] = r;
  }
}

static int c4b_plan(const mi355_act* x, C4BArgs& a) {
  a.tilesY = ceil_div(x->h, 8); a.tilesX = ceil_div(x->w, 16);
  const long long cols = (long long)x->n * a.tilesY * a.tilesX;
  if (cols <= 0 || cols > 0x7fffffffLL) return 0;
  // enough z chunks for ~1024 workgroups (two per CU, two rounds), at least 8 planes each (a chunk re-reads two halo planes)
  int zch = (int)((1024 + cols - 1) / cols);
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
  if (x->dtype != MI355_ACT_F32 || dy->dtype != MI355_ACT_F32 || d->precision != MI355_PREC_F32) return MI355_EUNSUPPORTED;
  if (x->n != dy->n || x->d != dy->d || x->h != dy->h || x->w != dy->w) return MI355_EINVAL;
  if (x->ld % 4 || dy->ld % 4 || x->ld < 4 || dy->ld < 32 || ((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & 15)) return MI355_EINVAL;
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
  return (size_t)x->n * a.tilesY * a.tilesX * a.zchunks * 4096 * sizeof(float);
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
  if (ws_bytes < (size_t)wgs * 4096 * sizeof(float)) return MI355_EWORKSPACE;
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.wp = wp_dgrad; a.ws = (float*)ws; a.part = partials_out;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope; a.mean_rstd = mean_rstd; a.groups = groups;
  a.N = x->n; a.D = x->d; a.H = x->h; a.W = x->w;
  const int lds_bytes = (3 * 180 * 32 + 3 * 180 * 4 + 512) * (int)sizeof(float);      // 79 808: two workgroups per CU
  if (d->in_mode == MI355_IN_PLAIN) {
    SET_MAX_DYN_LDS((conv3d_c4_bwd<MI355_IN_PLAIN>), lds_bytes);
    LAUNCH((conv3d_c4_bwd<MI355_IN_PLAIN>), dim3((unsigned)wgs), dim3(256), lds_bytes, stream, a);
  } else {
    SET_MAX_DYN_LDS((conv3d_c4_bwd<MI355_IN_AFFINE_ACT>), lds_bytes);
    LAUNCH((conv3d_c4_bwd<MI355_IN_AFFINE_ACT>), dim3((unsigned)wgs), dim3(256), lds_bytes, stream, a);
  }
  const int rc = LAUNCH_CHECK(); if (rc) return rc;
  conv3d_c4_wgrad_reduce_launch((const float*)ws, dw, 32, (int)wgs, stream);
  return LAUNCH_CHECK();
}
