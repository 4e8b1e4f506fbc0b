// Element-wise definitions of the packed weight layouts, shared by the single-weight pack kernels (conv3d_fwd.hip, conv3d_wino.hip) and
// the table-driven batch kernel (mi355_pack_weights_batch): a training step repacks ~70 weights after every optimizer step, and 70
// launches of 5-12 us each are pure launch latency (0.6 ms per step) where one launch does the same work.
#pragma once

// fp32 MFMA layout [tap][ciP / 4][coP][4] (conv3d_fwd.hip). cout / cin are the PACKED roles; modes as mi355_pack_conv_weight.
__device__ __forceinline__ float pack_f32_value(const float* w, size_t idx, int cout, int cin, int T, int coutP, int cinP, int mode) {
  const int e = idx & 3;
  size_t r = idx >> 2;
  const int o = r % coutP; r /= coutP;
  const int iq = r % (cinP / 4); r /= (cinP / 4);
  const int t = (int)r;
  const int i = iq * 4 + e;
  if (o >= cout || i >= cin) return 0.f;
  const int tf = T - 1 - t;
  if (mode == 0) return w[((size_t)o * cin + i) * T + t];
  if (mode == 1) return w[((size_t)i * cout + o) * T + tf];   // w[co=i][ci=o], here cout/cin are the packed roles
  if (mode == 2) return w[((size_t)i * cout + o) * T + tf];   // w[ci=i][co=o][flip]
  return w[((size_t)o * cin + i) * T + t];                    // mode 3: w[ci=o][co=i][t]
}

// Winograd-domain layout [(p * 3 + dz)][ciP / 4][coP][4], U = G g G^T of the (dy, dx) slice (conv3d_wino.hip), G rows: g0,
// (g0+g1+g2)/2, (g0-g1+g2)/2, g2. mode 0: forward, w OIDHW [cout][cin][3][3][3]; mode 1: dgrad of Conv3d (roles swapped, taps flipped).
// One WORK ITEM = one (dz, ci, co): its 9 weights are read once and all 16 points written (a thread per output element, the first
// form, read 9 weights per element: 16x the loads). Item index r runs over [dz][ciP / 4][coP][4], so that consecutive threads write
// consecutive addresses inside each of the 16 point slabs.
__device__ __forceinline__ void pack_wino_item(const float* w, float* up, size_t r, int cout, int cin, int coutP, int cinP, int mode) {
  const int e = r & 3;
  size_t q = r >> 2;
  const int o = q % coutP; q /= coutP;
  const int iq = q % (cinP / 4); q /= (cinP / 4);
  const int dz = (int)q;
  const int i = iq * 4 + e;
  const size_t slab = (size_t)3 * (cinP / 4) * coutP * 4;          // floats between consecutive points
  float* dst = up + ((size_t)dz * (cinP / 4) + iq) * coutP * 4 + (size_t)o * 4 + e;
  float u[4][4];
  if (o < cout && i < cin) {
    float g[3][3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
        g[dy][dx] = mode == 0 ? w[((size_t)o * cin + i) * 27 + (dz * 3 + dy) * 3 + dx]
                              : w[((size_t)i * cout + o) * 27 + ((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx)];      // w[co = i][ci = o], flipped
    float t[4][3];                        // G applied along dy
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      t[0][dx] = g[0][dx]; t[1][dx] = 0.5f * (g[0][dx] + g[1][dx] + g[2][dx]); t[2][dx] = 0.5f * (g[0][dx] - g[1][dx] + g[2][dx]); t[3][dx] = g[2][dx];
    }
#pragma unroll
    for (int pi = 0; pi < 4; ++pi) {      // ... and along dx: the same expressions as the per-element form, bit for bit
      u[pi][0] = t[pi][0]; u[pi][1] = 0.5f * (t[pi][0] + t[pi][1] + t[pi][2]); u[pi][2] = 0.5f * (t[pi][0] - t[pi][1] + t[pi][2]); u[pi][3] = t[pi][2];
    }
  } else {
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
      for (int pj = 0; pj < 4; ++pj) u[pi][pj] = 0.f;
  }
#pragma unroll
  for (int pi = 0; pi < 4; ++pi)
#pragma unroll
    for (int pj = 0; pj < 4; ++pj) dst[(size_t)(pi * 4 + pj) * slab] = u[pi][pj];
}

// Winograd F(2x2x2, 3x3x3) layout (conv3d_wino3d.hip): U[(zi, i, j)][ci][co] = sum_{dz,dy,dx} G[zi][dz] G[i][dy] G[j][dx] w[...], G as above.
// Pack layout [point 64][ci / 4][co tile][lane half 2][co 32][2]: element (h, co, e) of a (point, quad, channel tile) is channel 4 q + 2 h + e.
// mode 0: forward, w OIDHW [cout][cin][3][3][3]; mode 1: dgrad of Conv3d (roles swapped, taps flipped). One WORK ITEM = one (ci, co): its 27
// weights are read once and all 64 points written; item index r runs over [quad][channel tile][half][co][e], the order inside a point's slab.
__device__ __forceinline__ void pack_wino3_item(const float* w, float* up, size_t r, int cout, int cin, int coutP, int cinP, int mode) {
  const int e = r & 1;
  size_t q = r >> 1;
  const int col = q & 31; q >>= 5;
  const int h = q & 1; q >>= 1;
  const int ct = q % (coutP / 32); q /= (coutP / 32);
  const int cq = (int)q;
  const int o = ct * 32 + col, i = cq * 4 + 2 * h + e;
  const size_t slab = (size_t)cinP * coutP;                // floats between consecutive points
  float* dst = up + r;
  float u[4][4][4];
  if (o < cout && i < cin) {
    float u2[3][4][4];                                     // the plane transform of every dz slice: the expressions of pack_wino_item
#pragma unroll
    for (int dz = 0; dz < 3; ++dz) {
      float g[3][3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
          g[dy][dx] = mode == 0 ? w[((size_t)o * cin + i) * 27 + (dz * 3 + dy) * 3 + dx]
                                : w[((size_t)i * cout + o) * 27 + ((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx)];      // w[co = i][ci = o], flipped
      float t[4][3];
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        t[0][dx] = g[0][dx]; t[1][dx] = 0.5f * (g[0][dx] + g[1][dx] + g[2][dx]); t[2][dx] = 0.5f * (g[0][dx] - g[1][dx] + g[2][dx]); t[3][dx] = g[2][dx];
      }
#pragma unroll
      for (int pi = 0; pi < 4; ++pi) {
        u2[dz][pi][0] = t[pi][0]; u2[dz][pi][1] = 0.5f * (t[pi][0] + t[pi][1] + t[pi][2]); u2[dz][pi][2] = 0.5f * (t[pi][0] - t[pi][1] + t[pi][2]);
        u2[dz][pi][3] = t[pi][2];
      }
    }
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
      for (int pj = 0; pj < 4; ++pj) {
        u[0][pi][pj] = u2[0][pi][pj];
        u[1][pi][pj] = 0.5f * (u2[0][pi][pj] + u2[1][pi][pj] + u2[2][pi][pj]);
        u[2][pi][pj] = 0.5f * (u2[0][pi][pj] - u2[1][pi][pj] + u2[2][pi][pj]);
        u[3][pi][pj] = u2[2][pi][pj];
      }
  } else {
#pragma unroll
    for (int pz = 0; pz < 4; ++pz)
#pragma unroll
      for (int pi = 0; pi < 4; ++pi)
#pragma unroll
        for (int pj = 0; pj < 4; ++pj) u[pz][pi][pj] = 0.f;
  }
#pragma unroll
  for (int pz = 0; pz < 4; ++pz)
#pragma unroll
    for (int pi = 0; pi < 4; ++pi)
#pragma unroll
      for (int pj = 0; pj < 4; ++pj) dst[(size_t)((pz * 4 + pi) * 4 + pj) * slab] = u[pz][pi][pj];
}


// 16-bit operand layouts of the bf16 matrix paths [tap][ciP / 8][plane][coP][8] (conv3d_bf16.hip), ciP = roundup(cin, 16): NS planes per
// element (hi, residual of hi, ...) of bf16, or one plane of IEEE fp16. One WORK ITEM = one (tap, ci, co) element (its NS planes).
__device__ __forceinline__ void pack_lp_item(const float* w, unsigned short* wp, size_t idx, int cout, int cin, int T, int coutP, int cinP,
                                             int mode, int NS, int f16) {
  const int e = idx & 7;
  size_t r = idx >> 3;
  const int o = r % coutP; r /= coutP;
  const int i8 = r % (cinP / 8); r /= (cinP / 8);
  const int t = (int)r;
  const int i = i8 * 8 + e;
  float v = 0.f;
  if (o < cout && i < cin) {
    const int tf = T - 1 - t;
    if (mode == 0) v = w[((size_t)o * cin + i) * T + t];
    else v = w[((size_t)i * cout + o) * T + tf];
  }
  for (int p = 0; p < NS; ++p) {
    const unsigned pk = f16 ? pack_f16x2(v, 0.f) : pack_bf16x2(v, 0.f);
    wp[((((size_t)t * (cinP / 8) + i8) * NS + p) * coutP + o) * 8 + e] = (unsigned short)(pk & 0xffffu);
    v -= bf16lo_to_f32(pk);
  }
}
