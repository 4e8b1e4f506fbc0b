// 1x1x1 convolution (forward and data gradient) of bf16 tensors as independent wave streams, gfx950.
//
//   y[v][co] = sum_ci x[v][ci] * w[ci][co] (+ residual[v][co])        (reference: the shortcut / projection convolutions,
//   unet3d/models/pytorch/classification/resnet.py:20-22, decoder.py:99-106; their data gradients are the same call with the mode-1 pack)
//
// What it replaces: conv3d_mfma<1, 1, 1, 1, 256, 32, ...> on bf16 tensors. The 16-bit modes run these convolutions in exact fp32 arithmetic
// (fp32 weights), and the template does that on the fp32 matrix pipe behind an LDS tile: 2.2-2.8 TB/s on the two launches that carry the
// bytes (64 -> 32 channels @128^3 and its data gradient), slower per byte than the same launches on fp32 tensors.
//
// Here the matrix orientation is turned round so that NOTHING goes through LDS: M = output channels (A = the weights, resident in registers),
// N = 32 voxels, K = input channels. The B operand of v_mfma_f32_32x32x16_bf16 is "lane (voxel j, k-group) supplies 8 consecutive k": for a
// channels-last bf16 tensor that is one 16-byte global load per lane, straight into the operand register. The accumulator holds, per lane, 16
// output channels of ONE voxel in four runs of four; the lane pair of a voxel trades runs (v_permlane32_swap) and each lane stores two 16-byte
// runs of the voxel's row after the residual add.
// Exactness: the fp32 weights are split into three bf16 pieces in the prologue (w = w0 + w1 + w2 exactly: 3 x 8 significand bits), x is
// bf16 already, products of two bf16 values are exact in fp32: three MFMAs per k-step give the fp32 result of the template, summation order
// aside -- the kernel is HBM-bound (12-24 MFMAs per 6 KB), the extra matrix work is free.
// Every wave walks its own contiguous voxel range in chunks of 32 voxels with the loads four chunks ahead (registers).
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

struct K1SArgs {
  const bf16_t* x; int xld;
  const float* wp; int coutP;          // fp32 pack [1][cinP / 4][coutP][4] (mi355_pack_conv_weight mode 0 / 1: roles already swapped for the dgrad)
  bf16_t* y; int yld;
  const bf16_t* res; int resld;        // NULL or a tensor of y's shape
  long long V, nchunks;                // voxels, 32-voxel chunks
  int chunksPer;                       // chunks per wave
};

__device__ __forceinline__ uint4 ldg16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }

template <int CIK, int COT>            // input channels / 16, output channels / 32
__global__ __launch_bounds__(256) void conv3d_k1_stream_bf16(K1SArgs a) {
  constexpr int D = 4;                 // chunks in flight per wave
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  // ---- the weights: lane (co = li, k-group half) holds w[ci = 16 s + 8 half + e][co], e = 0..7, as three bf16 pieces ----
  uint4 A[COT][CIK][3];
#pragma unroll
  for (int t = 0; t < COT; ++t)
#pragma unroll
    for (int s = 0; s < CIK; ++s) {
      const float4* wq = reinterpret_cast<const float4*>(a.wp) + ((size_t)(4 * s + 2 * half) * a.coutP + 32 * t + li);
      const float4 w0 = wq[0], w1 = wq[a.coutP];
      float r[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        unsigned pk[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pk[e] = pack_bf16x2(r[2 * e], r[2 * e + 1]);
          r[2 * e] -= bf16lo_to_f32(pk[e]); r[2 * e + 1] -= bf16hi_to_f32(pk[e]);
        }
        A[t][s][p] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
      }
    }
  const long long gw = (long long)blockIdx.x * 4 + wave;
  const long long c_begin = gw * a.chunksPer;
  const long long c_end = c_begin + a.chunksPer < a.nchunks ? c_begin + a.chunksPer : a.nchunks;
  // B operand: lane (voxel li of the chunk, k-group half): channels 16 s + 8 half + 0..7
  auto load_chunk = [&](uint4 (&X)[CIK], long long c) {
    long long v = c * 32 + li;
    if (v >= a.V) v = a.V - 1;                                    // (past the tensor: a legal address, the stores are masked)
    const bf16_t* p = a.x + (size_t)v * a.xld + 8 * half;
#pragma unroll
    for (int s = 0; s < CIK; ++s) X[s] = ldg16(p + 16 * s);
  };
  uint4 X[D][CIK];
#pragma unroll
  for (int u = 0; u < D; ++u) load_chunk(X[u], c_begin + u < a.nchunks ? c_begin + u : a.nchunks - 1);
  for (long long c0 = c_begin; c0 < c_end; c0 += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const long long c = c0 + u;
      if (c >= c_end) break;
      const long long v = c * 32 + li;
      const bool live = v < a.V;
      f32x16 acc[COT];
      uint4 rs[COT][2];
#pragma unroll
      for (int t = 0; t < COT; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        if (a.res) {
          const bf16_t* rp = a.res + (size_t)(live ? v : a.V - 1) * a.resld + 32 * t + 8 * half;
#pragma unroll
          for (int q = 0; q < 2; ++q) rs[t][q] = ldg16(rp + 16 * q);
        }
      }
#pragma unroll
      for (int s = 0; s < CIK; ++s)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int t = 0; t < COT; ++t) acc[t] = MFMA_32x32x16_BF16(A[t][s][p], X[u][s], acc[t]);
      {
        const long long cn = c + D;                               // the chunk that takes this register set next
        load_chunk(X[u], cn < a.nchunks ? cn : a.nchunks - 1);
      }
      // accumulator register r = output channel (r & 3) + 8 (r >> 2) + 4 half of tile t, this lane's voxel: runs of 4 channels. The lane pair
      // (voxel, half 0 / 1) trades runs (v_permlane32_swap: the upper half's run g = 2 q <-> the lower half's run 2 q + 1) so that each lane
      // holds 8 consecutive channels 16 q + 8 half + 0..7: 16-byte stores (the first form stored 8-byte runs: 2.5 TB/s on the write-heavy
      // 32 -> 64 launch)
#pragma unroll
      for (int t = 0; t < COT; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float up = acc[t][8 * q + e], lo = acc[t][8 * q + 4 + e];      // (vector elements do not bind to references)
            permlane32_swap(up, lo);
            acc[t][8 * q + e] = up; acc[t][8 * q + 4 + e] = lo;
          }
      if (live) {
        bf16_t* yp = a.y + (size_t)v * a.yld + 8 * half;
#pragma unroll
        for (int t = 0; t < COT; ++t)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = acc[t][8 * q + e];
            if (a.res) {
              const unsigned rw[4] = {rs[t][q].x, rs[t][q].y, rs[t][q].z, rs[t][q].w};
#pragma unroll
              for (int e = 0; e < 4; ++e) { o[2 * e] += bf16lo_to_f32(rw[e]); o[2 * e + 1] += bf16hi_to_f32(rw[e]); }
            }
            *reinterpret_cast<uint4*>(yp + 32 * t + 16 * q) =
                make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
          }
      }
    }
  }
}

struct K1SPlan { int cik, cot, chunksPer, grid, ok; long long V, nchunks; };

// the calls it takes: bf16 tensors, 1x1x1 stride 1 pad 0, plain input and output, optional residual, no bias / channel scale / statistics /
// window, packed fp32 weights, (cin, cout) in {(32, 64), (64, 32), (64, 64)} -- the launches of UNet3D that carry the bytes; 16-byte aligned
// voxels. MI355_K1_STREAM=0 (read once): never -- the A/B switch.
static K1SPlan plan_k1s(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) {
  K1SPlan p; memset(&p, 0, sizeof(p));
  static const bool off = [] { const char* v = getenv("MI355_K1_STREAM"); return v && v[0] == '0'; }();
  if (off || !x || !y || !d) return p;
  if (d->kd != 1 || d->stride != 1 || d->pad != 0 || d->in_mode != MI355_IN_PLAIN || d->out_mode != MI355_OUT_PLAIN) return p;
  if (d->bias || d->out_chscale || d->gn_bwd || d->moments_out || d->off_z || d->off_y || d->off_x || d->wformat != MI355_W_PACKED) return p;
  if (x->dtype != MI355_ACT_BF16 || y->dtype != MI355_ACT_BF16) return p;
  if (x->n != y->n || x->d != y->d || x->h != y->h || x->w != y->w || d->out_d != y->d || d->out_h != y->h || d->out_w != y->w) return p;
  if (!((x->c == 32 && y->c == 64) || (x->c == 64 && y->c == 32) || (x->c == 64 && y->c == 64))) return p;
  if (x->ld % 8 || ((uintptr_t)x->p & 15) || y->ld % 8 || ((uintptr_t)y->p & 15)) return p;
  if (d->residual && (d->residual_ld % 8 || ((uintptr_t)d->residual & 15))) return p;
  p.cik = x->c / 16; p.cot = y->c / 32;
  p.V = (long long)x->n * x->d * x->h * x->w;
  if (p.V < 32) return p;
  p.nchunks = (p.V + 31) / 32;
  long long per = (p.nchunks + 4095) / 4096;                      // ~1024 workgroups of four waves: up to four per CU
  if (per < 8) per = 8;
  if (per > 0x7fffffffLL) return p;
  p.chunksPer = (int)per;
  const long long wgs = (p.nchunks + 4 * per - 1) / (4 * per);
  if (wgs > 0x7fffffffLL) return p;
  p.grid = (int)wgs;
  p.ok = 1;
  return p;
}

int mi355_conv3d_k1_stream_ok(const mi355_act* x, const mi355_act* y, const mi355_conv_desc* d) { return plan_k1s(x, y, d).ok; }

int mi355_conv3d_k1_stream_impl(const mi355_act* x, const float* wp, const mi355_act* y, const mi355_conv_desc* d, void* stream) {
  const K1SPlan p = plan_k1s(x, y, d);
  if (!p.ok) return MI355_EUNSUPPORTED;
  K1SArgs a; memset(&a, 0, sizeof(a));
  a.x = (const bf16_t*)x->p; a.xld = x->ld; a.wp = wp; a.coutP = (y->c + 31) / 32 * 32;
  a.y = (bf16_t*)y->p; a.yld = y->ld; a.res = (const bf16_t*)d->residual; a.resld = d->residual_ld;
  a.V = p.V; a.nchunks = p.nchunks; a.chunksPer = p.chunksPer;
  if (p.cik == 2) LAUNCH((conv3d_k1_stream_bf16<2, 2>), dim3((unsigned)p.grid), dim3(256), 0, stream, a);
  else if (p.cot == 1) LAUNCH((conv3d_k1_stream_bf16<4, 1>), dim3((unsigned)p.grid), dim3(256), 0, stream, a);
  else LAUNCH((conv3d_k1_stream_bf16<4, 2>), dim3((unsigned)p.grid), dim3(256), 0, stream, a);
  return LAUNCH_CHECK();
}
