// Shared declarations of the 16-bit-operand 3x3x3 convolution kernels: the tile forms (conv3d_bf16.hip) and the plane-ring forms
// (conv3d_bf16_zring.hip). Reference op: unet3d/models/pytorch/classification/resnet.py:12-22 (conv3x3x3).
#pragma once
#include "gfx950_dialect.h"
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


// plane-ring launches (conv3d_bf16_zring.hip); the planner and the dispatcher live in conv3d_bf16.hip
int mi355_lp_zring_launch(ConvBArgs& a, int in_mode, int fuse, bool f16, bool lps, long long blocks, void* stream);      // lps: bf16 storage
int mi355_lp_zring2_launch(ConvBArgs& a, int ks, int in_mode, int fuse, bool f16, bool lps, long long wgs, void* stream);
