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
__device__ __forceinline__ float pack_wino_value(const float* w, size_t idx, int cout, int cin, int coutP, int cinP, int mode) {
  const int e = idx & 3;
  size_t r = idx >> 2;
  const int o = r % coutP; r /= coutP;
  const int iq = r % (cinP / 4); r /= (cinP / 4);
  const int pd = (int)r, p = pd / 3, dz = pd % 3;
  const int pi = p >> 2, pj = p & 3;
  const int i = iq * 4 + e;
  if (o >= cout || i >= cin) return 0.f;
  float g[3][3];
  for (int dy = 0; dy < 3; ++dy)
    for (int dx = 0; dx < 3; ++dx)
      g[dy][dx] = mode == 0 ? w[((size_t)o * cin + i) * 27 + (dz * 3 + dy) * 3 + dx]
                            : w[((size_t)i * cout + o) * 27 + ((2 - dz) * 3 + (2 - dy)) * 3 + (2 - dx)];      // w[co = i][ci = o], flipped
  float t[3];                           // row pi of G applied along dy
  for (int dx = 0; dx < 3; ++dx)
    t[dx] = pi == 0 ? g[0][dx] : pi == 1 ? 0.5f * (g[0][dx] + g[1][dx] + g[2][dx]) : pi == 2 ? 0.5f * (g[0][dx] - g[1][dx] + g[2][dx]) : g[2][dx];
  return pj == 0 ? t[0] : pj == 1 ? 0.5f * (t[0] + t[1] + t[2]) : pj == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2];
}
