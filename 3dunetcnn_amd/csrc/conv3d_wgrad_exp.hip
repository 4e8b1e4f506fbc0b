// EXPERIMENTAL variants of the plane-ring 3x3x3 stride-1 weight-gradient kernel (conv3d_wgrad.hip: conv3d_wgrad_ring), reachable
// only through mi355_conv3d_wgrad_ring_exp -- mi355_conv3d_wgrad (the product path) never dispatches here. Prepared for an A/B on
// hardware; kept apart so that the validated kernel's code is untouched.
//
// What they try: the shipped kernel distributes the 27 tap tiles of a (32 co x 32 ci) pair over its
// 4 waves as 7/7/7/6, and the 28th v_mfma_f32_32x32x2_f32 of every k-step is discarded (3.6 % of the matrix work); it also pays
// one barrier + ring commit per plane of 4x8 voxels.
//   M16 : the same GEMM on v_mfma_f32_16x16x4_f32 (same 64 flop/clk/SIMD): 2 (co halves) x 54 (tap, ci half) = 108 tiles of 16x16,
//         exactly 27 per wave. Operands are ds_read_b32 of 16 consecutive channels for 4 consecutive voxels; a channel row is stored
//         with its two 16-channel halves swapped when bit 1 of the voxel's (haloed) x coordinate is set, so the 4 voxels x 16
//         channels of one read cover the 64 banks for every tap offset.
//   TY8 : columns of 8x8 instead of 4x8 voxels: half the barriers per output voxel, halo 1.56 instead of 1.88 input voxels per
//         output voxel, 67.6 KB of LDS (two workgroups per CU, which is what the register file allows anyway).
// Output format, workspace layout and the deterministic slab reduction are those of the shipped kernel.
#include "hipcompat.h"
#include "../../include/mi355_unet3d.h"

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream);

struct RingXArgs {
  const float* x; int xld;
  const float* dy; int dyld;
  float* ws;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  int N, Di, Hi, Wi, Cin;
  int Do, Ho, Wo, Cout;
  int zchunks, planes, tilesY, tilesX, chunks, splits, ciTiles, coTiles;
};

template <int INMODE, int TY, bool M16>
__global__ __launch_bounds__(256) void conv3d_wgrad_ring_x(RingXArgs a) {
  constexpr int TX = 8, HY = TY + 2, HX = 10, PV = TY * TX, HPV = HY * HX;
  constexpr int XSLOT = HPV * 32, DSLOT = PV * 32;
  constexpr int NXU = (HPV * 8 + 255) / 256, NDU = (PV * 8) / 256;     // staging units (voxel, channel quad) per thread and plane
  static_assert((PV * 8) % 256 == 0, "dy plane staging");
  DYN_LDS(lds);
  float* lds_x = lds;                    // ring of 4 haloed input planes [HPV][32]
  float* lds_dy = lds + 4 * XSLOT;       // 2 dy planes [PV][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x, cit = blockIdx.y, cot = blockIdx.z;
  const int ci0 = cit * 32, co0 = cot * 32;

  // ---- MFMA tile ownership ----
  // M32: wave w owns taps w, w+4, ... (7 accumulators of 32x32, the 7th of wave 3 is a discarded duplicate), as the shipped kernel.
  // M16: wave w owns n-tiles 13w .. 13w+12 (n = 2*tap + ci half) for both co halves, plus the single tile (n = 52 + w/2, m = w & 1).
  constexpr int NT32 = 7, NN16 = 13;
  const int half = lane >> 5, li = lane & 31;          // M32 lane roles
  const int j16 = lane & 15, kq = lane >> 4;           // M16 lane roles: column / row index, k (voxel) index
  f32x16 acc32[M16 ? 1 : NT32];
  f32x4 acc16[M16 ? NN16 : 1][2], accx;
  int tdz[M16 ? NN16 + 1 : NT32], tin[M16 ? NN16 + 1 : NT32];   // per owned tile: dz of its tap, in-plane LDS offset (floats) incl. the channel index
  if constexpr (M16) {
#pragma unroll
    for (int i = 0; i <= NN16; ++i) {
      const int n = i < NN16 ? NN16 * wave + i : 52 + (wave >> 1);
      const int tap = n >> 1, h = n & 1;
      const int dyy = (tap / 3) % 3, dx = tap % 3;
      // channel position of this lane inside a voxel row: halves swapped where bit 1 of the haloed x coordinate is set; the lane's
      // x within its 4-voxel group is kq, so that bit is ((kq + dx) >> 1) & 1 for every k-step (4 * (ks & 1) only moves bit 2)
      const int sb = ((kq + dx) >> 1) & 1;
      tdz[i] = tap / 9;
      tin[i] = (dyy * HX + dx) * 32 + ((h * 16 + j16) ^ (sb << 4));
    }
#pragma unroll
    for (int i = 0; i < NN16; ++i)
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[i][m][r] = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) accx[r] = 0.f;
  } else {
#pragma unroll
    for (int ti = 0; ti < NT32; ++ti) {
      int tap = wave + 4 * ti;
      if (tap >= 27) tap = 26;
      tdz[ti] = tap / 9;
      tin[ti] = (((tap / 3) % 3) * HX + tap % 3) * 32 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc32[ti][r] = 0.f;
    }
  }

  const int sq = tid & 7, sv0 = tid >> 3;          // staging unit u: voxel sv0 + 32 * u of the plane, channel quad sq
  const int cdy = co0 + 4 * sq, cx = ci0 + 4 * sq;
  const bool dyvalid = cdy < a.Cout, xvalid = cx < a.Cin;
  for (int chunk = split; chunk < a.chunks; chunk += a.splits) {
    int b = chunk;
    const int zc = b % a.zchunks; b /= a.zchunks;
    const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
    const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
    const int n = b;
    const int zb = zc * a.planes, ze = zb + a.planes < a.Do ? zb + a.planes : a.Do;
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
      xl[u] = hvc * 32 + 4 * (M16 ? (sq ^ (((hx >> 1) & 1) << 2)) : sq);
    }
    bool din[NDU]; size_t dyo[NDU]; int dl[NDU];
#pragma unroll
    for (int u = 0; u < NDU; ++u) {
      const int v = sv0 + 32 * u;
      const int oy = ty0 + v / TX, ox = tx0 + v % TX;
      din[u] = dyvalid && oy < a.Ho && ox < a.Wo;
      dyo[u] = ((size_t)(oy < a.Ho ? oy : a.Ho - 1) * a.Wo + (ox < a.Wo ? ox : a.Wo - 1)) * a.dyld + (dyvalid ? cdy : 0);
      dl[u] = v * 32 + 4 * (M16 ? (sq ^ ((((v % TX) >> 1) & 1) << 2)) : sq);
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
        v.x = v.x > 0.f ? v.x : v.x * sl.x; v.y = v.y > 0.f ? v.y : v.y * sl.y;
        v.z = v.z > 0.f ? v.z : v.z * sl.z; v.w = v.w > 0.f ? v.w : v.w * sl.w;
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

    __syncthreads();                               // the previous chunk's last plane may still be read by other waves
    for (int iz = zb - 1; iz <= zb + 1; ++iz) { load_x(iz); commit_x(iz); }
    load_dy(zb); commit_dy(zb);
    __syncthreads();

    for (int z = zb; z < ze; ++z) {
      const bool more = z + 1 < ze;
      if (more) { load_x(z + 2); load_dy(z + 1); }
      SCHED_BARRIER();                 // the loads stay above the MFMA loop they overlap with
      if constexpr (M16) {
        int soff[NN16 + 1];
#pragma unroll
        for (int i = 0; i <= NN16; ++i) soff[i] = ((z + tdz[i]) & 3) * XSLOT + tin[i];
        // A operand: dy[voxel][co]: lane's co position inside the row, halves swapped where bit 1 of x = 4 * (ks & 1) + kq is set
        const float* dys = lds_dy + (z & 1) * DSLOT;
        const int ca0 = (0 * 16 + j16) ^ (((kq >> 1) & 1) << 4), ca1 = (1 * 16 + j16) ^ (((kq >> 1) & 1) << 4);
        const int xm = wave & 1;                   // co half of the single extra tile
#pragma unroll 1                   // 27 MFMAs per iteration; unrolling by 2 costs 14+ VGPRs and the second wave per SIMD (144 VGPR + 108 AGPR fit two)
        for (int ks = 0; ks < PV / 4; ++ks) {
          const int v = 4 * ks + kq;
          const float a0 = dys[v * 32 + ca0], a1 = dys[v * 32 + ca1];
          const int xb = ((v / TX) * HX + v % TX) * 32;
#pragma unroll
          for (int i = 0; i < NN16; ++i) {
            const float bv = lds_x[soff[i] + xb];
            acc16[i][0] = MFMA_16x16x4(a0, bv, acc16[i][0]);
            acc16[i][1] = MFMA_16x16x4(a1, bv, acc16[i][1]);
          }
          accx = MFMA_16x16x4(xm ? a1 : a0, lds_x[soff[NN16] + xb], accx);
        }
      } else {
        int soff[NT32];
#pragma unroll
        for (int ti = 0; ti < NT32; ++ti) soff[ti] = ((z + tdz[ti]) & 3) * XSLOT + tin[ti];
        const float* dys = lds_dy + (z & 1) * DSLOT + li;
#pragma unroll 4
        for (int ks = 0; ks < PV / 2; ++ks) {
          const int v = 2 * ks + half;
          const float av = dys[v * 32];
          const int xb = ((v / TX) * HX + v % TX) * 32;
#pragma unroll
          for (int ti = 0; ti < NT32; ++ti) acc32[ti] = MFMA_32x32x2(av, lds_x[soff[ti] + xb], acc32[ti]);
        }
      }
      SCHED_BARRIER();
      if (more) { commit_x(z + 2); commit_dy(z + 1); }
      __syncthreads();
    }
  }

  // ---- partial tiles: ws[pair][slab = split][tap][32 co][32 ci] ----
  const size_t pair = (size_t)cot * a.ciTiles + cit;
  float* slab = a.ws + (pair * a.splits + split) * 27 * 1024;
  if constexpr (M16) {
    // lane holds D[i = 4 * (lane >> 4) + r][j = lane & 15] of its 16x16 tile: co = 16 m + i, ci = 16 h + j
#pragma unroll
    for (int i = 0; i <= NN16; ++i) {
      const int n = i < NN16 ? NN16 * wave + i : 52 + (wave >> 1);
      const int tap = n >> 1, h = n & 1;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (i == NN16 && m != (wave & 1)) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = i < NN16 ? acc16[i < NN16 ? i : 0][m][r] : accx[r];
          slab[tap * 1024 + (16 * m + 4 * kq + r) * 32 + 16 * h + j16] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int ti = 0; ti < NT32; ++ti) {
      const int tap = wave + 4 * ti;
      if (tap >= 27) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
        slab[tap * 1024 + row * 32 + li] = acc32[ti][r];
      }
    }
  }
}

// ---- slab reduction with more bytes in flight (variant bit 2) ----
// The shipped reduction gives a (pair, tap) tile to 1 or 4 workgroups, 8 loads in flight per thread: with one pair (the 32 -> 32
// layers: 512 slabs, 56.6 MB) that is 108 workgroups x 256 threads x 8 x 16 B = 3.5 MB in flight, a third of what the HBM latency
// needs (measured 39 us per launch on average, 1.4 ms per step, for ~12 us of traffic). Here every tile is always split over 4
// workgroups (one 64-float4 column group each) and a thread keeps 16 running sums = 16 loads in flight; same fixed summation tree
// per element on every run (deterministic), different from the shipped kernel's in the last bits.
__global__ __launch_bounds__(256) void wgrad_reduce_x_kernel(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles) {
  __shared__ float4 part[4][64];
  const int tile = blockIdx.x >> 2, sub = blockIdx.x & 3;
  const int tap = tile % T, pair = tile / T;
  const int cot = pair / ciTiles, cit = pair % ciTiles;
  const int tid = threadIdx.x, e4 = tid & 63, grp = tid >> 6;
  const int f4 = e4 + 64 * sub;                                  // float4 index inside the 32x32 tile: row co = f4 / 8, 4 ci
  const float4* base = reinterpret_cast<const float4*>(ws + (((size_t)pair * SL) * T + tap) * 1024) + f4;
  const size_t slab_stride = (size_t)T * 256;                   // in float4
  float4 acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
  int k = grp;
  for (; k + 60 < SL; k += 64) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const float4 v = base[(size_t)(k + 4 * u) * slab_stride];
      acc[u].x += v.x; acc[u].y += v.y; acc[u].z += v.z; acc[u].w += v.w;
    }
  }
  for (int u = 0; k < SL; k += 4, ++u) {
    const float4 v = base[(size_t)k * slab_stride];
    acc[u & 15].x += v.x; acc[u & 15].y += v.y; acc[u & 15].z += v.z; acc[u & 15].w += v.w;
  }
#pragma unroll
  for (int w = 8; w >= 1; w >>= 1)
#pragma unroll
    for (int u = 0; u < w; ++u) { acc[u].x += acc[u + w].x; acc[u].y += acc[u + w].y; acc[u].z += acc[u + w].z; acc[u].w += acc[u + w].w; }
  part[grp][e4] = acc[0];
  __syncthreads();
  if (grp == 0) {
    float4 s4 = part[0][e4];
#pragma unroll
    for (int g = 1; g < 4; ++g) { s4.x += part[g][e4].x; s4.y += part[g][e4].y; s4.z += part[g][e4].z; s4.w += part[g][e4].w; }
    const int co = cot * 32 + f4 / 8, ci = cit * 32 + (f4 % 8) * 4;
    if (co < Cout) {
      const float v[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (ci + e < Cin) dw[((size_t)co * Cin + ci + e) * T + tap] = v[e];
    }
  }
}

static int wgrad_reduce_x_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream) {
  const long long tiles = (long long)ceil_div(Cout, 32) * ciTiles * T;
  if (tiles <= 0 || tiles * 4 > 0x7fffffffLL) return MI355_EINVAL;
  LAUNCH(wgrad_reduce_x_kernel, dim3((unsigned)(tiles * 4)), dim3(256), 0, stream, ws, dw, Cout, Cin, T, SL, ciTiles);
  return LAUNCH_CHECK();
}

// plan of the shipped ring kernel (conv3d_wgrad.hip: plan_wgrad_ring) with the column height as a parameter
struct RingXPlan { int tilesY, tilesX, zchunks, planes, chunks, splits, ciTiles, coTiles; size_t ws_bytes; int ok; };
static RingXPlan plan_ring_x(const mi355_act* x, const mi355_act* dy, int ty) {
  RingXPlan p; memset(&p, 0, sizeof(p));
  if (!x || !dy || x->d != dy->d || x->h != dy->h || x->w != dy->w || x->n != dy->n) return p;
  p.tilesY = ceil_div(dy->h, ty); p.tilesX = ceil_div(dy->w, 8);
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

// variant: bit 0 = M16 (16x16x4 MFMA tiles, 27 per wave), bit 1 = TY8 (8x8-voxel columns), bit 2 = the 16-in-flight slab reduction
extern "C" size_t mi355_conv3d_wgrad_ring_exp_workspace(const mi355_act* x, const mi355_act* dy, int32_t variant) {
  RingXPlan p = plan_ring_x(x, dy, (variant & 2) ? 8 : 4);
  return p.ok ? p.ws_bytes : 0;
}

template <int TY, bool M16>
static int launch_ring_x(RingXArgs& a, int in_mode, void* stream) {
  constexpr size_t lds = (size_t)(4 * (TY + 2) * 10 * 32 + 2 * TY * 8 * 32) * sizeof(float);
  dim3 grid(a.splits, a.ciTiles, a.coTiles);
  if (in_mode == MI355_IN_PLAIN) {
    SET_MAX_DYN_LDS((conv3d_wgrad_ring_x<MI355_IN_PLAIN, TY, M16>), lds);
    LAUNCH((conv3d_wgrad_ring_x<MI355_IN_PLAIN, TY, M16>), grid, dim3(256), lds, stream, a);
  } else {
    SET_MAX_DYN_LDS((conv3d_wgrad_ring_x<MI355_IN_AFFINE_ACT, TY, M16>), lds);
    LAUNCH((conv3d_wgrad_ring_x<MI355_IN_AFFINE_ACT, TY, M16>), grid, dim3(256), lds, stream, a);
  }
  return LAUNCH_CHECK();
}

extern "C" int mi355_conv3d_wgrad_ring_exp(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d,
                                           void* ws, size_t ws_bytes, int32_t variant, void* stream) {
  if (!x || !dy || !dw || !d || !ws || !x->p || !dy->p) return MI355_EINVAL;
  if (x->c % 4 || x->ld % 4 || dy->c % 4 || dy->ld % 4 || x->n != dy->n) return MI355_EINVAL;
  if (((uintptr_t)x->p & 15) || ((uintptr_t)dy->p & 15)) return MI355_EINVAL;
  if (d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift)) return MI355_EINVAL;
  if (variant < 0 || variant > 7) return MI355_EINVAL;
  RingXPlan r = plan_ring_x(x, dy, (variant & 2) ? 8 : 4);
  if (!r.ok) return MI355_EUNSUPPORTED;
  if (ws_bytes < r.ws_bytes) return MI355_EWORKSPACE;
  RingXArgs a; memset(&a, 0, sizeof(a));
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.in_scale = d->in_scale; a.in_shift = d->in_shift; a.slope = d->act_slope; a.in_slope = d->in_slope;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Cin = x->c;
  a.Do = dy->d; a.Ho = dy->h; a.Wo = dy->w; a.Cout = dy->c;
  a.zchunks = r.zchunks; a.planes = r.planes; a.tilesY = r.tilesY; a.tilesX = r.tilesX; a.chunks = r.chunks; a.splits = r.splits;
  a.ciTiles = r.ciTiles; a.coTiles = r.coTiles;
  int rc;
  switch (variant & 3) {
    case 0: rc = launch_ring_x<4, false>(a, d->in_mode, stream); break;
    case 1: rc = launch_ring_x<4, true>(a, d->in_mode, stream); break;
    case 2: rc = launch_ring_x<8, false>(a, d->in_mode, stream); break;
    default: rc = launch_ring_x<8, true>(a, d->in_mode, stream); break;
  }
  if (rc) return rc;
  if (variant & 4) return wgrad_reduce_x_launch((const float*)ws, dw, a.Cout, a.Cin, 27, r.splits, r.ciTiles, stream);
  return mi355_wgrad_reduce_launch((const float*)ws, dw, a.Cout, a.Cin, 27, r.splits, r.ciTiles, stream);
}
