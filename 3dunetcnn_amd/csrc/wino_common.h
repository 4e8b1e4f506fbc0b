// Shared by the two Winograd forward / dgrad kernels of the 3x3x3 stride-1 convolution: conv3d_wino.hip (F(2x2, 3x3) in the plane x direct z)
// and conv3d_wino3d.hip (F(2x2x2, 3x3x3)). Argument block, the statistics tail of the fused epilogues, the eligibility check.
#pragma once
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "gn_fuse.h"

struct WinoArgs {
  const float* x; int xld;
  const float* up;                       // transformed weights [(p * 3 + dz)][ciP / 4][coP][4]
  float* y; int yld;
  const float* res; int resld;
  const float* in_scale; const float* in_shift; float slope; const float* in_slope;
  const float* out_chscale; const float* bias;
  int N, D, H, W, Cin, CinP, Cout, CoutP;
  int tilesZ, tilesY, tilesX, coTiles;
  int vec4;                              // output / residual / normalised tensor take 16-byte accesses per channel quad
  GnFuseArgs g;                          // norm statistics fused into the epilogue (gn_fuse.h), as in conv3d_mfma
};


// The statistics tail of a fused epilogue (FUSE 1: moments of the stored output, FUSE 2: norm-backward sums): per-lane partials of 4 channels
// -> one record per (tile, channel).
template <int FUSE, int NW = 8>
__device__ __forceinline__ void wino_fuse_records(const WinoArgs& a, float* P, int tid, int lane, int wave, int coq, int co_base, int n, int tz0, int ty0,
                                                  int tx0, int cnt, float (&K0)[4], float (&s0)[4], float (&s1)[4]) {
  constexpr int TZ = 2, TY = 8, TX = 16;

    // Per-lane partials of 4 channels -> one record per (tile, channel). Lanes coq + 8 m (m = 0..7) of a wave hold the same channels:
    // three xor-shuffle steps of PLAIN sums (fixed order: the lane with the lower m first), then the eight waves through LDS in wave
    // order (Chan's merge, as everywhere). Moments: a lane's sums are about its own first value K0; before the shuffles they are moved
    // to the wave's common shift Kc = K0 of lane m = 0 (sum (v - Kc) = s0 + c d, sum (v - Kc)^2 = s1 + d (2 s0 + c d), d = K0 - Kc: no
    // division, no E[x^2] - E[x]^2 of raw values), and M2 = s1 - s0^2 / c is formed once per wave and channel.
    constexpr int KK = FUSE == 1 ? 3 : 2;
    float vals[4][KK];
    float cw = (float)cnt;                                  // FUSE 1: stored voxels of this lane (the same for its four channels)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if constexpr (FUSE == 1) {
        const float Kc = __shfl(K0[e], coq);
        const float d = K0[e] - Kc;
        vals[e][0] = Kc;
        vals[e][2] = s1[e] + d * (2.f * s0[e] + cw * d);
        vals[e][1] = s0[e] + cw * d;
      } else {
        vals[e][0] = s0[e]; vals[e][1] = s1[e];
      }
    }
#pragma unroll
    for (int step = 8; step < 64; step <<= 1) {
      const bool upper = lane & step;
      if constexpr (FUSE == 1) { const float oc = __shfl_xor(cw, step); cw = upper ? oc + cw : cw + oc; }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = (FUSE == 1 ? 1 : 0); k < KK; ++k) {
          const float o = __shfl_xor(vals[e][k], step);
          vals[e][k] = upper ? o + vals[e][k] : vals[e][k] + o;
        }
    }
    // the NW waves of the workgroup (8, or 16 in conv3d_wino3d) through LDS: moments as (count, sum about Kc, sum of squares about Kc, Kc) per wave, moved to wave 0's shift by
    // the same identity and added in wave order; M2 = s1 - s0^2 / c once per channel
    constexpr int KW = FUSE == 1 ? 4 : 2;
    __syncthreads();                                       // every wave is done with the exchange
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float* pr = P + ((wave * 32) + 4 * coq + e) * KW;
        if constexpr (FUSE == 1) { pr[0] = cw; pr[1] = vals[e][1]; pr[2] = vals[e][2]; pr[3] = vals[e][0]; }
        else { pr[0] = vals[e][0]; pr[1] = vals[e][1]; }
      }
    }
    __syncthreads();
    if (tid < 32) {
      float r[KK];
      if constexpr (FUSE == 1) {
        const float K = P[tid * KW + 3];
        float c = P[tid * KW], t0 = P[tid * KW + 1], t1 = P[tid * KW + 2];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float* pr = P + (w * 32 + tid) * KW;
          const float cwv = pr[0], d = pr[3] - K;
          t1 += pr[2] + d * (2.f * pr[1] + cwv * d);
          t0 += pr[1] + cwv * d;
          c += cwv;
        }
        const float m2 = c > 0.f ? t1 - t0 * t0 / c : 0.f;
        r[0] = c; r[1] = t0 + c * K; r[2] = m2 > 0.f ? m2 : 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < KK; ++k) r[k] = P[tid * KW + k];
#pragma unroll
        for (int w = 1; w < NW; ++w)
#pragma unroll
          for (int k = 0; k < KK; ++k) r[k] += P[(w * 32 + tid) * KW + k];
      }
      const int tile = ((tz0 / TZ) * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
      const size_t rec = (size_t)n * ((size_t)a.tilesZ * a.tilesY * a.tilesX) + tile;
      float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * KK;
      const int co = co_base + tid;
      if (co < a.Cout) {
#pragma unroll
        for (int k = 0; k < KK; ++k) dst[(size_t)co * KK + k] = r[k];
      }
    }
  }


// x, y: NDHWC activations of the same extent; up: mi355_wino_pack_weight of the [y->c][x->c] (mode 0) weights; desc: kd 3, stride 1, pad 1,
// plain / norm-prologue input, plain un-windowed output; bias, residual, out_chscale as in mi355_conv3d_fwd.
// 0 = this call is one mi355_conv3d_wino_fwd accepts, else the status it would return (shape / mode / alignment): the caller routes a
// call that is not eligible to mi355_conv3d_fwd instead of failing (ops.Backend.conv_fwd).
static inline int wino_check(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  if (!x || !y || !d || !x->p || !y->p) return MI355_EINVAL;
  if (x->dtype != MI355_ACT_F32 || y->dtype != MI355_ACT_F32) return MI355_EUNSUPPORTED;      // the Winograd kernels are the fp32 path
  if (d->kd != 3 || d->stride != 1 || d->pad != 1 || d->out_mode != MI355_OUT_PLAIN) return MI355_EUNSUPPORTED;
  if (d->in_mode != MI355_IN_PLAIN && d->in_mode != MI355_IN_AFFINE_ACT) return MI355_EUNSUPPORTED;
  if (d->off_z || d->off_y || d->off_x || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return MI355_EUNSUPPORTED;
  if (x->d != y->d || x->h != y->h || x->w != y->w || x->n != y->n) return MI355_EINVAL;
  if (x->c % 4 || x->ld % 4 || x->ld < x->c || y->ld < y->c || ((uintptr_t)x->p & 15)) return MI355_EINVAL;
  if (d->in_mode == MI355_IN_AFFINE_ACT && (!d->in_scale || !d->in_shift || !(d->act_slope >= 0.f && d->act_slope <= 1.f))) return MI355_EINVAL;
  if (d->residual && d->residual_ld < y->c) return MI355_EINVAL;
  if (d->moments_out && d->gn_bwd) return MI355_EUNSUPPORTED;
  if (d->gn_bwd && d->in_mode != MI355_IN_PLAIN) return MI355_EUNSUPPORTED;
  return MI355_OK;
}

