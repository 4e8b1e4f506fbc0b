// 3x3x3 stride-2 pad-1 conv3d forward for 32 -> 32 channels (round 6): the first down-sampling convolution of the default UNet3D
// (unet3d/models/pytorch/classification/myronenko.py:93-101, `downsampling_convolutions[0]`: 128^3 -> 64^3 at the headline size), exact fp32.
//
// Why a kernel of its own: the generic conv3d_mfma<3, 2, ...> stages a haloed 9 x 9 x 17 voxel tile 8 channels at a time, i.e. takes 32
// bytes of every voxel's 128-byte line per pass, four passes per tile -- and 64 resident workgroups x 176 KB of such tiles do not fit an
// XCD's 4 MB L2: PMC 2.4 GB fetched for 0.6 GB algorithmic, the launch runs at HBM speed (0.43 ms at 128^3 x 2 against 0.18 ms of matrix
// time). 16-channel chunks in that kernel need 110 KB of LDS (one workgroup per CU) and measured slower.
// Here a 256-thread workgroup marches a 4 x 8 output-voxel column along z: a ring of three haloed input planes (9 x 17 voxels x ALL 32
// channels: every 128-byte line is fetched whole and once per column, 1.2x with the (y, x) halo; two new planes per output plane, loaded into
// registers during the previous plane's MFMAs) and the four waves split K: wave w owns input channels 8 w .. 8 w + 7 of all 27 taps with
// its 27 weight quads per lane resident in registers for the whole march (108 VGPRs: this is why 32 channels and not more); per tap one
// ds_read_b128 of the voxel's channel quad feeds 4 MFMAs (read one tap ahead, by hand: hipcc does not pipeline LDS reads into MFMAs); the
// four partial 32 x 32 accumulators meet in LDS once per plane, where the 256 threads add them, store 16 bytes each (coalesced 128-byte
// voxel rows) and keep the output's moments for the next norm (one record per workgroup, gn_fuse.h format).
// TA: storage type of x and y (act_io.h): the 16-bit modes run this layer in exact fp32 on the stored values, like conv3d_mfma.
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

struct S2Args {
  const float* x; int xld;
  const float* wp;                       // forward pack [27][8][32][4] (mi355_pack_conv_weight mode 0, cin = cout = 32)
  float* y; int yld;
  float* mom;                            // moment records [n][B][32][3] or NULL
  const float* res; int resld;           // (data gradient only) NULL or a tensor of y's shape and storage type added in the epilogue: the skip gradient
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int tilesY, tilesX, zchunks, zper;
};

template <typename TA, bool FUSE>
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD(2) void conv3d_s2c32_fwd(S2Args a) {
  constexpr int TY = 4, TX = 8, HY = 2 * TY + 1, HX = 2 * TX + 1, HV = HY * HX;      // 9 x 17 = 153 input voxels per plane
  constexpr int PL = HV * 32;                            // floats per staged plane
  constexpr int UNITS = 2 * HV * 8, UP = (UNITS + 255) / 256;      // 16-byte units of the two planes a step loads, per thread
  const TA* const ax = reinterpret_cast<const TA*>(a.x);
  TA* const ay = reinterpret_cast<TA*>(a.y);
  DYN_LDS(lds);
  float* xs = lds;                                       // ring of 3 planes: input plane p in slot (p + 3) % 3
  float* ex = lds + 3 * PL;                              // exchange [wave][voxel 32][co 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int zc = b % a.zchunks; b /= a.zchunks;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int n = b;
  const int z_begin = zc * a.zper, z_end = z_begin + a.zper < a.Do ? z_begin + a.zper : a.Do;
  const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;        // input origin of the haloed plane tile

  // the wave's weights: 27 quads per lane (quad 2 w + half of the input channels, output channel li), resident for the whole march
  float4 W[27];
  {
    const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
#pragma unroll
    for (int t = 0; t < 27; ++t) W[t] = wp4[(size_t)(t * 8 + 2 * wave + half) * 32 + li];
  }

  // Staging units of a step: two input planes x 153 voxels x 8 channel quads = 2 448 units of 16 bytes, unit u = tid + 256 k. Everything a
  // unit needs per step is 32 bits wide and computed ONCE: its element offset inside a plane (clamped to the image: the load is always
  // legal), its LDS offset inside a slot, one validity bit. (The first form kept ten 64-bit addresses per thread, spilled them, and every
  // scratch reload put an `s_waitcnt vmcnt(0)` between the global loads: 13 us per plane.)
  float4 st[UP];
  unsigned goff[UP], loff[UP];
  unsigned okmask = 0, ppmask = 0;                       // bit k: unit k lies inside the image in (y, x); unit k belongs to the pair's second plane
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    int u = tid + k * 256; const bool live = u < UNITS; if (!live) u = UNITS - 1;
    const int pp = u / (HV * 8), r = u % (HV * 8), hv = r >> 3, q = r & 7;
    const int iy = iy0 + hv / HX, ix = ix0 + hv % HX;
    const bool in = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
    const int cy = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), cx = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
    goff[k] = (unsigned)((cy * a.Wi + cx) * a.xld + 4 * q);
    loff[k] = live ? (unsigned)(hv * 32 + 4 * q) : 0xffffffffu;
    okmask |= (unsigned)(in && live) << k;
    ppmask |= (unsigned)pp << k;
  }
  const size_t xplane = (size_t)a.Hi * a.Wi * a.xld;
  const TA* const xn = ax + (size_t)n * a.Di * xplane;
  auto load_planes = [&](int p0) {                       // input planes p0, p0 + 1 -> registers
    const int z0 = p0 < 0 ? 0 : (p0 < a.Di ? p0 : a.Di - 1), z1 = p0 + 1 < 0 ? 0 : (p0 + 1 < a.Di ? p0 + 1 : a.Di - 1);
    const TA* pA = xn + (size_t)z0 * xplane;             // wave-uniform plane bases
    const TA* pB = xn + (size_t)z1 * xplane;
#pragma unroll
    for (int k = 0; k < UP; ++k) st[k] = ld4(((ppmask >> k) & 1u ? pB : pA) + goff[k]);
  };
  auto commit_planes = [&](int p0, bool first_only) {
    const bool zokA = p0 >= 0 && p0 < a.Di, zokB = p0 + 1 >= 0 && p0 + 1 < a.Di;
    float* sA = xs + ((p0 + 3) % 3) * PL;
    float* sB = xs + ((p0 + 4) % 3) * PL;
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      const bool second = (ppmask >> k) & 1u;
      if (loff[k] == 0xffffffffu || (first_only && second)) continue;
      const bool ok = ((okmask >> k) & 1u) && (second ? zokB : zokA);
      *reinterpret_cast<float4*>((second ? sB : sA) + loff[k]) = ok ? st[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  // A operand: lane li = output voxel (row li >> 3, column li & 7) of the plane tile, lane half = which of the wave's two channel quads
  const int abase = ((2 * (li >> 3)) * HX + 2 * (li & 7)) * 32 + (2 * wave + half) * 4;
  // final stage: thread = (voxel tid >> 3, output-channel quad tid & 7)
  const int fv = tid >> 3, fq = tid & 7;
  const int oy = ty0 + (fv >> 3), ox = tx0 + (fv & 7);
  const bool fin = oy < a.Ho && ox < a.Wo;
  float K0[4] = {0.f, 0.f, 0.f, 0.f}, s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
  float cnt = 0.f;

  // ---- prologue: planes 2 z_begin - 1, 2 z_begin, 2 z_begin + 1 (loads come in pairs: the second pair's second plane is dropped --
  // the loop's first step loads the pair (2 z_begin + 2, 2 z_begin + 3) like every other step: one code path) ----
  load_planes(2 * z_begin - 1); commit_planes(2 * z_begin - 1, false);
  load_planes(2 * z_begin + 1); commit_planes(2 * z_begin + 1, true);
  __syncthreads();

  for (int zo = z_begin; zo < z_end; ++zo) {
    // planes 2 zo - 1, 2 zo, 2 zo + 1 are staged. The next output plane needs 2 zo + 2, 2 zo + 3: in flight during the MFMAs
    load_planes(2 * zo + 2);
    const float* pl[3] = {xs + ((2 * zo - 1 + 3) % 3) * PL, xs + ((2 * zo + 3) % 3) * PL, xs + ((2 * zo + 1 + 3) % 3) * PL};
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 cur = *reinterpret_cast<const float4*>(pl[0] + abase), nxt = cur;
    static_for<0, 27>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      if constexpr (t + 1 < 27) {
        constexpr int t1 = t + 1, dz = t1 / 9, dy = (t1 / 3) % 3, dx = t1 % 3;
        nxt = *reinterpret_cast<const float4*>(pl[dz] + abase + (dy * HX + dx) * 32);
      }
      SCHED_BARRIER();
      acc = MFMA_32x32x2(cur.x, W[t].x, acc); acc = MFMA_32x32x2(cur.y, W[t].y, acc);
      acc = MFMA_32x32x2(cur.z, W[t].z, acc); acc = MFMA_32x32x2(cur.w, W[t].w, acc);
      SCHED_BARRIER();
      cur = nxt;
    });
    // partial tile -> exchange [wave][voxel row][co]: accumulator register r is voxel (r & 3) + 8 (r >> 2) + 4 half, lane li = co
#pragma unroll
    for (int r = 0; r < 16; ++r) ex[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
    __syncthreads();                                     // every wave is done with the ring; the partials are visible
    commit_planes(2 * zo + 2, false);                    // into the slots of planes 2 zo - 1 and 2 zo
    {
      const float* e0 = ex + fv * 32 + 4 * fq;
      const float4 p0 = *reinterpret_cast<const float4*>(e0), p1 = *reinterpret_cast<const float4*>(e0 + 1024);
      const float4 p2 = *reinterpret_cast<const float4*>(e0 + 2048), p3 = *reinterpret_cast<const float4*>(e0 + 3072);
      float ov[4] = {(p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w)};
      if (fin) {
        TA* yp = ay + ((((size_t)n * a.Do + zo) * a.Ho + oy) * a.Wo + ox) * a.yld + 4 * fq;
        st4(yp, make_float4(ov[0], ov[1], ov[2], ov[3]));
        if constexpr (FUSE) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = as_stored(yp, ov[e]);
            if (cnt == 0.f) K0[e] = v;
            const float d = v - K0[e];
            s0[e] += d; s1[e] += d * d;
          }
          cnt += 1.f;
        }
      }
    }
    __syncthreads();                                     // the ring is complete for the next plane; the exchange is free
  }

  if constexpr (FUSE) {
    // one record per workgroup and channel: lanes fq + 8 m of a wave hold the same channels -> xor-shuffle steps 8, 16, 32 (the lower lane
    // first), then the four waves through LDS in wave order; partial (c, K, s0, s1) pairs merge by moving the second to the first's shift
    auto merge = [](float& ca, float& ka, float& a0, float& a1, float cb, float kb, float b0, float b1) {
      if (ca == 0.f) { ca = cb; ka = kb; a0 = b0; a1 = b1; return; }
      const float d = kb - ka;
      a1 += b1 + d * (2.f * b0 + cb * d);
      a0 += b0 + cb * d;
      ca += cb;
    };
#pragma unroll
    for (int step = 8; step < 64; step <<= 1) {
      const bool upper = lane & step;
      const float oc = __shfl_xor(cnt, step);
      float nc = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float ok_ = __shfl_xor(K0[e], step), o0 = __shfl_xor(s0[e], step), o1 = __shfl_xor(s1[e], step);
        float ca, ka, a0, a1;
        if (upper) { ca = oc; ka = ok_; a0 = o0; a1 = o1; merge(ca, ka, a0, a1, cnt, K0[e], s0[e], s1[e]); }
        else { ca = cnt; ka = K0[e]; a0 = s0[e]; a1 = s1[e]; merge(ca, ka, a0, a1, oc, ok_, o0, o1); }
        K0[e] = ka; s0[e] = a0; s1[e] = a1; nc = ca;
      }
      cnt = nc;
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* pr = ex + ((wave * 32) + 4 * fq + e) * 4;
        pr[0] = cnt; pr[1] = K0[e]; pr[2] = s0[e]; pr[3] = s1[e];
      }
    }
    __syncthreads();
    if (tid < 32) {
      float c = ex[tid * 4], k = ex[tid * 4 + 1], t0 = ex[tid * 4 + 2], t1 = ex[tid * 4 + 3];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float* pr = ex + (w * 32 + tid) * 4;
        merge(c, k, t0, t1, pr[0], pr[1], pr[2], pr[3]);
      }
      const float m2 = c > 0.f ? t1 - t0 * t0 / c : 0.f;
      const size_t B = (size_t)a.tilesY * a.tilesX * a.zchunks;
      float* dst = a.mom + (((size_t)n * B + (blockIdx.x % B)) * 32 + tid) * 3;
      dst[0] = c; dst[1] = t0 + c * k; dst[2] = m2 > 0.f ? m2 : 0.f;
    }
  }
}

// ---- weight gradient of the same layer: dw[co][ci][tap] = sum over output voxels q of dy[q][co] * x[2 q + tap - 1][ci] ----
// The same march and the same input ring; the plane's 32 output voxels are K (16 MFMA k-steps), the 27 taps are dealt to the four waves
// (7 / 7 / 7 / 6: tap = (wave & 3) + 4 ti; the other wave bit halves K: 512 threads, so that a thread stages 5 units instead of 10 -- with
// 112 accumulator registers the 10-unit form spilled) whose 32 (co) x 32 (ci) accumulators stay in registers for the whole march -- no exchange; A = dy
// (lane = co, k = voxel of the pair), B = x at the tap's offset (lane = ci): plain ds_read_b32, read one k-step ahead. The generic
// conv3d_wgrad_mfma<3, 2, 2, 2, 8> stages a haloed 5 x 5 x 17 tile per 32 output voxels (1.7x the input) and 0.51 ms.
// Partial tiles go to the slab workspace of conv3d_wgrad.hip ([slab][tap][32 co][32 ci]) and its deterministic reduction.
struct S2WArgs {
  const float* x; int xld;
  const float* dy; int dyld;
  float* ws;
  int N, Di, Hi, Wi, Do, Ho, Wo;
  int tilesY, tilesX, zchunks, zper;
};

template <typename TA>
__global__ __launch_bounds__(512) MIN_WAVES_PER_SIMD(2) void conv3d_s2c32_wgrad(S2WArgs a) {
  constexpr int TY = 4, TX = 8, HY = 2 * TY + 1, HX = 2 * TX + 1, HV = HY * HX;
  constexpr int PL = HV * 32;
  constexpr int UNITS = 2 * HV * 8, UP = (UNITS + 511) / 512;      // 512 threads: 5 staging units each (256 threads: 10, and spills)
  constexpr int NTW = 7;                                 // taps per wave (the last wave's seventh is a duplicate that is never written)
  const TA* const ax = reinterpret_cast<const TA*>(a.x);
  const TA* const ady = reinterpret_cast<const TA*>(a.dy);
  DYN_LDS(lds);
  float* xs = lds;                                       // ring of 3 input planes, as the forward
  float* dys = lds + 3 * PL;                             // two dy planes [parity of the output plane][voxel 32][co 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int zc = b % a.zchunks; b /= a.zchunks;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int n = b;
  const int z_begin = zc * a.zper, z_end = z_begin + a.zper < a.Do ? z_begin + a.zper : a.Do;
  const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;

  float4 st[UP], dst4;
  unsigned goff[UP];                                     // (the LDS offset of a unit is recomputed at the commit: 112 accumulator registers live here)
  unsigned okmask = 0, ppmask = 0;
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    int u = tid + k * 512; const bool live = u < UNITS; if (!live) u = UNITS - 1;
    const int pp = u / (HV * 8), r = u % (HV * 8), hv = r >> 3, q = r & 7;
    const int iy = iy0 + hv / HX, ix = ix0 + hv % HX;
    const bool in = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
    const int cy = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), cx = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
    goff[k] = (unsigned)((cy * a.Wi + cx) * a.xld + 4 * q);
    okmask |= (unsigned)(in && live) << k;
    ppmask |= (unsigned)pp << k;
  }
  const size_t xplane = (size_t)a.Hi * a.Wi * a.xld;
  const TA* const xn = ax + (size_t)n * a.Di * xplane;
  // dy plane: thread (of the first 256) = (voxel tid >> 3, channel quad tid & 7)
  const int dv = (tid & 255) >> 3, dq = tid & 7;
  const int doy = ty0 + (dv >> 3), dox = tx0 + (dv & 7);
  const bool din = doy < a.Ho && dox < a.Wo;
  const unsigned dgoff = (unsigned)(((din ? doy : 0) * a.Wo + (din ? dox : 0)) * a.dyld + 4 * dq);
  const size_t dyplane = (size_t)a.Ho * a.Wo * a.dyld;
  const TA* const dyn = ady + (size_t)n * a.Do * dyplane;
  auto load_planes = [&](int p0, int zo) {               // input planes p0, p0 + 1 and the dy plane zo -> registers
    const int z0 = p0 < 0 ? 0 : (p0 < a.Di ? p0 : a.Di - 1), z1 = p0 + 1 < 0 ? 0 : (p0 + 1 < a.Di ? p0 + 1 : a.Di - 1);
    const TA* pA = xn + (size_t)z0 * xplane;
    const TA* pB = xn + (size_t)z1 * xplane;
#pragma unroll
    for (int k = 0; k < UP; ++k) st[k] = ld4(((ppmask >> k) & 1u ? pB : pA) + goff[k]);
    if (tid < 256) dst4 = ld4(dyn + (size_t)(zo < a.Do ? zo : a.Do - 1) * dyplane + dgoff);
  };
  auto commit_planes = [&](int p0, bool first_only) {
    const bool zokA = p0 >= 0 && p0 < a.Di, zokB = p0 + 1 >= 0 && p0 + 1 < a.Di;
    float* sA = xs + ((p0 + 3) % 3) * PL;
    float* sB = xs + ((p0 + 4) % 3) * PL;
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      const int u = tid + k * 512;
      const bool second = (ppmask >> k) & 1u;
      if (u >= UNITS || (first_only && second)) continue;
      const int r = u % (HV * 8);
      const bool ok = ((okmask >> k) & 1u) && (second ? zokB : zokA);
      *reinterpret_cast<float4*>((second ? sB : sA) + (r >> 3) * 32 + 4 * (r & 7)) = ok ? st[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto commit_dy = [&](int zo) {
    if (tid < 256) *reinterpret_cast<float4*>(dys + (zo & 1) * 1024 + dv * 32 + 4 * dq) = (din && zo < a.Do) ? dst4 : make_float4(0.f, 0.f, 0.f, 0.f);
  };

  // wave = (tap group tg = w & 3: taps tg + 4 ti, K half kh = w >> 2: k-steps 8 kh .. 8 kh + 7 of the plane's 16); the taps' offsets inside a
  // staged plane; B operand lane = ci, A operand lane = co
  const int tg = wave & 3, kh = wave >> 2;
  int tdz[NTW], toff[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    int tap = tg + 4 * ti; if (tap > 26) tap = 26;
    tdz[ti] = tap / 9;
    toff[ti] = (((tap / 3) % 3) * HX + tap % 3) * 32 + li;
  }
  f32x16 acc[NTW];
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[ti][r] = 0.f;

  // ---- prologue: input planes 2 zb - 1, 2 zb, 2 zb + 1 and dy plane zb ----
  load_planes(2 * z_begin - 1, z_begin); commit_planes(2 * z_begin - 1, false); commit_dy(z_begin);
  load_planes(2 * z_begin + 1, z_begin); commit_planes(2 * z_begin + 1, true);
  __syncthreads();

  for (int zo = z_begin; zo < z_end; ++zo) {
    load_planes(2 * zo + 2, zo + 1);                     // the next step's planes: in flight during the MFMAs
    const float* pl[3] = {xs + ((2 * zo - 1 + 3) % 3) * PL, xs + ((2 * zo + 3) % 3) * PL, xs + ((2 * zo + 1 + 3) % 3) * PL};
    const float* bp[NTW];
#pragma unroll
    for (int ti = 0; ti < NTW; ++ti) bp[ti] = pl[tdz[ti]] + toff[ti];
    const float* ap = dys + (zo & 1) * 1024 + li;
    // k-step ks: voxels 2 ks + half of the plane tile -> input voxel (2 (v >> 3), 2 (v & 7))
    struct Ops { float a, b[NTW]; };
    auto rd = [&](int ks, Ops& o) {
      const int v = 2 * ks + half;
      o.a = ap[v * 32];
      const int xo = ((2 * (v >> 3)) * HX + 2 * (v & 7)) * 32;
#pragma unroll
      for (int ti = 0; ti < NTW; ++ti) o.b[ti] = bp[ti][xo];
    };
    Ops oa, ob;
    rd(8 * kh, oa);
    static_for<0, 8>([&](auto kc) {
      constexpr int ks = decltype(kc)::value;
      Ops& cur = (ks & 1) ? ob : oa;
      Ops& nxt = (ks & 1) ? oa : ob;
      if constexpr (ks + 1 < 8) rd(8 * kh + ks + 1, nxt);
      SCHED_BARRIER();
#pragma unroll
      for (int ti = 0; ti < NTW; ++ti) acc[ti] = MFMA_32x32x2(cur.a, cur.b[ti], acc[ti]);
      SCHED_BARRIER();
    });
    __syncthreads();                                     // every wave is done with the ring and this dy plane
    commit_planes(2 * zo + 2, false);
    commit_dy(zo + 1);
    __syncthreads();
  }

  // ---- partial tiles: ws[slab = 2 workgroup + K half][tap][32 co][32 ci] ----
#pragma unroll
  for (int ti = 0; ti < NTW; ++ti) {
    const int tap = tg + 4 * ti;
    if (tap > 26) continue;
    float* dst = a.ws + ((((size_t)blockIdx.x * 2 + kh) * 27 + tap) * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[ti][r];
  }
}

// ---- data gradient of the same layer: dx[i][ci] = sum over (o, k) with 2 o + k - 1 = i of dy[o][co] * w[co][ci][k]  -- the call
// mi355_conv3d_fwd(dy, dgrad pack, dx, in_mode = MI355_IN_ZERO_INSERT) ----
// Per dimension an even position takes one tap (k = 1, o = i / 2), an odd one two (k = 2 at o = (i - 1) / 2, k = 0 at o = (i + 1) / 2): the 8
// parity classes of a 2 x 2 x 2 output block carry 1, 2, 2, 4, 2, 4, 4, 8 of the 27 taps. The generic template runs this as a stride-1
// convolution over the zero-inserted dy with parity-class tiles, re-staging the 110 KB of weights for every 256 output voxels (0.57 ms).
// Here: the forward kernel's march, mirrored. A workgroup walks a column of 4 x 8 dy voxels along z with a ring of three dy planes (5 x 9
// voxels with the +1 halo); per dy plane it produces the two dx planes 2 m, 2 m + 1 as 8 parity classes of 32 voxels x 32 input channels. The
// four waves split K (wave = two quads of the dy channels), each keeps the 27 weight quads of its lane in registers for the whole march (the
// dgrad pack [27][8][32][4] = flipped taps, roles swapped: packed tap t per dimension is k = 2 - t), class after class: 1-8 taps x 4
// MFMAs, partial tile to a double-buffered LDS exchange, one barrier, 256 threads add the four partials and store 16 bytes each.
// pair i of the data gradient's 27 (parity class, tap) pairs, classes in order 0..7 (class c = (pz, py, px) bits, 1 / 2 / 2 / 4 / 2 / 4 / 4 / 8 taps):
// per dimension parity 0 -> packed tap 1 at dy offset 0; parity 1 -> packed tap 0 at offset 0 and packed tap 2 at offset + 1
struct S2DgradTap {
  int c, t, oz, off; bool first, last;
  static constexpr S2DgradTap at(int i) {
    int c = 0, start = 0;
    for (; c < 8; ++c) {
      const int nt = (1 + (c >> 2)) * (1 + ((c >> 1) & 1)) * (1 + (c & 1));
      if (i < start + nt) break;
      start += nt;
    }
    const int pz = c >> 2, py = (c >> 1) & 1, px = c & 1, ny = 1 + py, nx = 1 + px, nt = (1 + pz) * ny * nx, k = i - start;
    const int iz = k / (ny * nx), iy = (k / nx) % ny, ix = k % nx;
    const int tz = pz ? 2 * iz : 1, ty = py ? 2 * iy : 1, tx = px ? 2 * ix : 1;
    return S2DgradTap{c, (tz * 3 + ty) * 3 + tx, pz ? iz : 0, ((py ? iy : 0) * 9 + (px ? ix : 0)) * 32, k == 0, k == nt - 1};
  }
};

template <typename TA>
__global__ __launch_bounds__(256) MIN_WAVES_PER_SIMD(2) void conv3d_s2c32_dgrad(S2Args a) {
  // (roles in S2Args: x = dy [N][Do][Ho][Wo][32 co], y = dx [N][Di][Hi][Wi][32 ci], wp = the dgrad pack)
  constexpr int TY = 4, TX = 8, HY = TY + 1, HX = TX + 1, HV = HY * HX;      // 5 x 9 = 45 dy voxels per plane
  constexpr int PL = HV * 32;
  constexpr int UNITS = HV * 8, UP = (UNITS + 255) / 256;                    // 360 units of 16 bytes per plane: 2 per thread
  const TA* const ady = reinterpret_cast<const TA*>(a.x);
  TA* const adx = reinterpret_cast<TA*>(a.y);
  DYN_LDS(lds);
  float* ds = lds;                                       // ring of 3 dy planes: plane p in slot p % 3
  float* ex = lds + 3 * PL;                              // exchange [buffer 2][wave][voxel 32][ci 32]
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int zc = b % a.zchunks; b /= a.zchunks;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int n = b;
  const int z_begin = zc * a.zper, z_end = z_begin + a.zper < a.Do ? z_begin + a.zper : a.Do;

  float4 W[27];
  {
    const float4* wp4 = reinterpret_cast<const float4*>(a.wp);
#pragma unroll
    for (int t = 0; t < 27; ++t) W[t] = wp4[(size_t)(t * 8 + 2 * wave + half) * 32 + li];
  }
  float4 st[UP];
  unsigned goff[UP], loff[UP], okmask = 0;
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    int u = tid + k * 256; const bool live = u < UNITS; if (!live) u = UNITS - 1;
    const int hv = u >> 3, q = u & 7;
    const int oy = ty0 + hv / HX, ox = tx0 + hv % HX;
    const bool in = oy < a.Ho && ox < a.Wo;
    goff[k] = (unsigned)(((in ? oy : 0) * a.Wo + (in ? ox : 0)) * a.xld + 4 * q);
    loff[k] = live ? (unsigned)(hv * 32 + 4 * q) : 0xffffffffu;
    okmask |= (unsigned)(in && live) << k;
  }
  const size_t dyplane = (size_t)a.Ho * a.Wo * a.xld;
  const TA* const dyn = ady + (size_t)n * a.Do * dyplane;
  auto load_plane = [&](int p) {
    const TA* pl = dyn + (size_t)(p < a.Do ? p : a.Do - 1) * dyplane;
#pragma unroll
    for (int k = 0; k < UP; ++k) st[k] = ld4(pl + goff[k]);
  };
  auto commit_plane = [&](int p) {                       // the plane past the tensor is the zero halo of the last odd dx plane
    float* sp = ds + (p % 3) * PL;
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      if (loff[k] == 0xffffffffu) continue;
      const bool ok = ((okmask >> k) & 1u) && p < a.Do;
      *reinterpret_cast<float4*>(sp + loff[k]) = ok ? st[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  // A operand: lane li = dy voxel (row li >> 3, column li & 7) of the plane tile, lane half = which of the wave's two channel quads
  const int abase = ((li >> 3) * HX + (li & 7)) * 32 + (2 * wave + half) * 4;
  // final stage: thread = (voxel tid >> 3, input-channel quad tid & 7)
  const int fv = tid >> 3, fq = tid & 7;
  const int fy = 2 * (ty0 + (fv >> 3)), fx = 2 * (tx0 + (fv & 7));

  load_plane(z_begin); commit_plane(z_begin);
  load_plane(z_begin + 1); commit_plane(z_begin + 1);
  __syncthreads();

  for (int m = z_begin; m < z_end; ++m) {
    load_plane(m + 2);                                   // in flight during the step; its slot held plane m - 1
    const float* pl[2] = {ds + (m % 3) * PL, ds + ((m + 1) % 3) * PL};
    // the 27 (class, tap) pairs as one sequence: the operand of pair i + 1 is read before the MFMAs of pair i (hipcc does not pipeline LDS reads
    // into MFMAs by itself); a class's last pair is followed by its exchange
    f32x16 acc;
    float4 cur = *reinterpret_cast<const float4*>(pl[S2DgradTap::at(0).oz] + abase + S2DgradTap::at(0).off), nxt = cur;
    static_for<0, 27>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr S2DgradTap tp = S2DgradTap::at(i);
      if constexpr (tp.first) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      }
      if constexpr (i + 1 < 27) {
        constexpr S2DgradTap tn = S2DgradTap::at(i + 1);
        nxt = *reinterpret_cast<const float4*>(pl[tn.oz] + abase + tn.off);
      }
      SCHED_BARRIER();
      acc = MFMA_32x32x2(cur.x, W[tp.t].x, acc); acc = MFMA_32x32x2(cur.y, W[tp.t].y, acc);
      acc = MFMA_32x32x2(cur.z, W[tp.t].z, acc); acc = MFMA_32x32x2(cur.w, W[tp.t].w, acc);
      SCHED_BARRIER();
      cur = nxt;
      if constexpr (tp.last) {
        constexpr int c = tp.c, pz = c >> 2, py = (c >> 1) & 1, px = c & 1;
        float* eb = ex + (c & 1) * 4096;
#pragma unroll
        for (int r = 0; r < 16; ++r) eb[(wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 32 + li] = acc[r];
        if constexpr (c == 7) commit_plane(m + 2);       // published by the same barrier as the last class's partials
        __syncthreads();                                 // partials visible; the other exchange buffer (class c - 1) has been read by everyone
        const float* e0 = eb + fv * 32 + 4 * fq;
        const float4 p0 = *reinterpret_cast<const float4*>(e0), p1 = *reinterpret_cast<const float4*>(e0 + 1024);
        const float4 p2 = *reinterpret_cast<const float4*>(e0 + 2048), p3 = *reinterpret_cast<const float4*>(e0 + 3072);
        const int z = 2 * m + pz, y = fy + py, x = fx + px;
        if (z < a.Di && y < a.Hi && x < a.Wi) {
          const size_t vox = (((size_t)n * a.Di + z) * a.Hi + y) * a.Wi + x;
          float4 o = make_float4((p0.x + p1.x) + (p2.x + p3.x), (p0.y + p1.y) + (p2.y + p3.y), (p0.z + p1.z) + (p2.z + p3.z), (p0.w + p1.w) + (p2.w + p3.w));
          if (a.res) {
            const float4 rv = ld4(reinterpret_cast<const TA*>(a.res) + vox * a.resld + 4 * fq);
            o.x += rv.x; o.y += rv.y; o.z += rv.z; o.w += rv.w;
          }
          st4(adx + vox * a.yld + 4 * fq, o);
        }
      }
    });
  }
}

static int s2_plan(const mi355_act* y, S2Args& a, int target = 256) {
  a.tilesY = ceil_div(y->h, 4); a.tilesX = ceil_div(y->w, 8);
  const long long cols1 = (long long)a.tilesY * a.tilesX, cols = cols1 * y->n;
  if (cols <= 0 || cols > 0x7fffffffLL) return 0;
  // z chunks PER SAMPLE (~256 workgroups per sample in the forward: two per CU and one round at batch 2; 1024 in all measured 0.332 -> below):
  // the moments records of a sample -- and with them the bits of its normalised output -- must not depend on how many samples ride along
  // (tests/test_fullsize_gpu.py: a sample alone equals the same sample in a batch, bit for bit)
  int zch = (int)((target + cols1 - 1) / cols1);
  const int maxch = y->d >= 8 ? y->d / 8 : 1;            // >= 8 output planes per chunk (a chunk stages 3 planes before its first MFMA)
  if (zch > maxch) zch = maxch;
  if (zch < 1) zch = 1;
  a.zper = ceil_div(y->d, zch);
  a.zchunks = ceil_div(y->d, a.zper);
  return cols * a.zchunks <= 0x7fffffffLL;
}

// the calls this kernel takes: 3x3x3 stride 2 pad 1, 32 -> 32 channels, plain input, plain un-windowed output, no bias / residual / channel
// scale, exact-fp32 arithmetic (every precision mode runs the stride-2 convolutions so), x and y of one storage type; optional moments.
// MI355_S2_KERNEL=0 (read once): never -- the A/B switch.
int mi355_conv3d_s2c32_ok(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  static const bool off = [] { const char* v = getenv("MI355_S2_KERNEL"); return v && v[0] == '0'; }();
  if (off || !x || !y || !d) return 0;
  if (d->kd != 3 || d->stride != 2 || d->pad != 1 || x->c != 32 || y->c != 32 || d->in_mode != MI355_IN_PLAIN || d->out_mode != MI355_OUT_PLAIN) return 0;
  if (d->bias || d->residual || d->out_chscale || d->gn_bwd || d->off_z || d->off_y || d->off_x) return 0;
  if (d->wformat != MI355_W_PACKED || x->dtype != y->dtype || !act_dtype_ok(x)) return 0;
  if (y->d != (x->d - 1) / 2 + 1 || y->h != (x->h - 1) / 2 + 1 || y->w != (x->w - 1) / 2 + 1 || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return 0;
  if (x->ld % 4 || y->ld % 4 || ((uintptr_t)x->p & act_align_mask(x->dtype)) || ((uintptr_t)y->p & act_align_mask(y->dtype))) return 0;
  return 1;
}

int32_t mi355_conv3d_s2c32_stats_blocks(const mi355_act* y) {
  S2Args a; memset(&a, 0, sizeof(a));
  if (!y || !s2_plan(y, a)) return 0;
  return a.tilesY * a.tilesX * a.zchunks;
}

int mi355_conv3d_s2c32_fwd_impl(const mi355_act* x, const float* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!mi355_conv3d_s2c32_ok(x, y, d)) return MI355_EUNSUPPORTED;
  S2Args a; memset(&a, 0, sizeof(a));
  if (!s2_plan(y, a)) return MI355_EINVAL;
  a.x = (const float*)x->p; a.xld = x->ld; a.wp = wp; a.y = (float*)y->p; a.yld = y->ld; a.mom = d->moments_out;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Do = y->d; a.Ho = y->h; a.Wo = y->w;
  const long long wgs = (long long)x->n * a.tilesY * a.tilesX * a.zchunks;
  const int lds_bytes = (3 * 153 * 32 + 4 * 32 * 32) * (int)sizeof(float);      // 75 136: two workgroups per CU
  ACT_TYPED(x->dtype, TA,
            if (a.mom) { SET_MAX_DYN_LDS((conv3d_s2c32_fwd<TA, true>), lds_bytes);
                         LAUNCH((conv3d_s2c32_fwd<TA, true>), dim3((unsigned)wgs), dim3(256), lds_bytes, stream, a); }
            else { SET_MAX_DYN_LDS((conv3d_s2c32_fwd<TA, false>), lds_bytes);
                   LAUNCH((conv3d_s2c32_fwd<TA, false>), dim3((unsigned)wgs), dim3(256), lds_bytes, stream, a); });
  return LAUNCH_CHECK();
}

// the data gradient of the same layer (mi355_conv3d_fwd with MI355_IN_ZERO_INSERT: x = dy, y = dx, the dgrad pack; optional residual = the
// skip gradient; no other epilogue, no window)
int mi355_conv3d_s2c32_dgrad_ok(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  static const bool off = [] { const char* v = getenv("MI355_S2_KERNEL"); return v && v[0] == '0'; }();
  if (off || !x || !y || !d) return 0;
  if (d->kd != 3 || d->stride != 1 || d->pad != 1 || x->c != 32 || y->c != 32 || d->in_mode != MI355_IN_ZERO_INSERT || d->out_mode != MI355_OUT_PLAIN) return 0;
  if (d->bias || d->out_chscale || d->gn_bwd || d->moments_out || d->off_z || d->off_y || d->off_x) return 0;
  if (d->residual && (d->residual_ld % 4 || ((uintptr_t)d->residual & act_align_mask(y->dtype)))) return 0;
  if (d->wformat != MI355_W_PACKED || x->dtype != y->dtype || !act_dtype_ok(x) || x->n != y->n) return 0;
  if (x->d != (y->d - 1) / 2 + 1 || x->h != (y->h - 1) / 2 + 1 || x->w != (y->w - 1) / 2 + 1 || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return 0;
  if (x->ld % 4 || y->ld % 4 || ((uintptr_t)x->p & act_align_mask(x->dtype)) || ((uintptr_t)y->p & act_align_mask(y->dtype))) return 0;
  return 1;
}

int mi355_conv3d_s2c32_dgrad_impl(const mi355_act* x, const float* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  if (!mi355_conv3d_s2c32_dgrad_ok(x, y, d)) return MI355_EUNSUPPORTED;
  S2Args a; memset(&a, 0, sizeof(a));
  if (!s2_plan(x, a)) return MI355_EINVAL;                 // columns of dy voxels
  a.x = (const float*)x->p; a.xld = x->ld; a.wp = wp; a.y = (float*)y->p; a.yld = y->ld;
  a.res = (const float*)d->residual; a.resld = d->residual_ld;
  a.N = x->n; a.Do = x->d; a.Ho = x->h; a.Wo = x->w; a.Di = y->d; a.Hi = y->h; a.Wi = y->w;
  const long long wgs = (long long)x->n * a.tilesY * a.tilesX * a.zchunks;
  const int lds_bytes = (3 * 45 * 32 + 2 * 4 * 32 * 32) * (int)sizeof(float);      // 50 048: two (three) workgroups per CU
  ACT_TYPED(x->dtype, TA, SET_MAX_DYN_LDS((conv3d_s2c32_dgrad<TA>), lds_bytes);
            LAUNCH((conv3d_s2c32_dgrad<TA>), dim3((unsigned)wgs), dim3(256), lds_bytes, stream, a));
  return LAUNCH_CHECK();
}

int mi355_wgrad_reduce_launch(const float* ws, float* dw, int Cout, int Cin, int T, int SL, int ciTiles, void* stream);      // conv3d_wgrad.hip

// the weight gradient of the same calls (x: input of the conv, dy: gradient of its output; plain input): MI355_S2_KERNEL=0 -> never
int mi355_conv3d_s2c32_wgrad_ok(const mi355_act* x, const mi355_act* dy, const mi355_conv_desc* d) {
  static const bool off = [] { const char* v = getenv("MI355_S2_KERNEL"); return v && v[0] == '0'; }();
  if (off || !x || !dy || !d) return 0;
  if (d->kd != 3 || d->stride != 2 || d->pad != 1 || x->c != 32 || dy->c != 32 || d->in_mode != MI355_IN_PLAIN || d->out_mode != MI355_OUT_PLAIN) return 0;
  if (x->dtype != dy->dtype || !act_dtype_ok(x) || x->n != dy->n) return 0;
  if (dy->d != (x->d - 1) / 2 + 1 || dy->h != (x->h - 1) / 2 + 1 || dy->w != (x->w - 1) / 2 + 1) return 0;
  if (x->ld % 4 || dy->ld % 4 || ((uintptr_t)x->p & act_align_mask(x->dtype)) || ((uintptr_t)dy->p & act_align_mask(dy->dtype))) return 0;
  return 1;
}

size_t mi355_conv3d_s2c32_wgrad_workspace(const mi355_act* dy) {
  S2Args a; memset(&a, 0, sizeof(a));
  if (!dy || !s2_plan(dy, a, 128)) return 0;              // ~128 workgroups per sample: one 8-wave workgroup per CU at batch 2
  return (size_t)dy->n * a.tilesY * a.tilesX * a.zchunks * 2 * 27 * 1024 * sizeof(float);
}

int mi355_conv3d_s2c32_wgrad_impl(const mi355_act* x, const mi355_act* dy, float* dw, const mi355_conv_desc* d, void* ws, size_t ws_bytes,
                                  void* stream) {
  if (!mi355_conv3d_s2c32_wgrad_ok(x, dy, d)) return MI355_EUNSUPPORTED;
  S2Args pl; memset(&pl, 0, sizeof(pl));
  if (!s2_plan(dy, pl, 128)) return MI355_EINVAL;
  const long long wgs = (long long)dy->n * pl.tilesY * pl.tilesX * pl.zchunks;
  if (ws_bytes < (size_t)wgs * 2 * 27 * 1024 * sizeof(float)) return MI355_EWORKSPACE;
  S2WArgs a; memset(&a, 0, sizeof(a));
  a.x = (const float*)x->p; a.xld = x->ld; a.dy = (const float*)dy->p; a.dyld = dy->ld; a.ws = (float*)ws;
  a.N = x->n; a.Di = x->d; a.Hi = x->h; a.Wi = x->w; a.Do = dy->d; a.Ho = dy->h; a.Wo = dy->w;
  a.tilesY = pl.tilesY; a.tilesX = pl.tilesX; a.zchunks = pl.zchunks; a.zper = pl.zper;
  const int lds_bytes = (3 * 153 * 32 + 2 * 32 * 32) * (int)sizeof(float);      // 66 944 bytes; 8 waves of up to 256 registers: one workgroup per CU
  ACT_TYPED(x->dtype, TA, SET_MAX_DYN_LDS((conv3d_s2c32_wgrad<TA>), lds_bytes);
            LAUNCH((conv3d_s2c32_wgrad<TA>), dim3((unsigned)wgs), dim3(512), lds_bytes, stream, a));
  const int rc = LAUNCH_CHECK(); if (rc) return rc;
  return mi355_wgrad_reduce_launch((const float*)ws, dw, 32, 32, 27, (int)wgs * 2, 1, stream);
}
