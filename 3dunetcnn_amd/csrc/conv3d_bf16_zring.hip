// Plane-ring (z-marching, weights-stationary) forms of the 3x3x3 stride-1 conv3d forward / dgrad on 16-bit operands (MI355_PREC_BF16 /
// MI355_PREC_F16; reference op: unet3d/models/pytorch/classification/resnet.py:12-22 called from myronenko.py:17-21, same fusions as
// conv3d_bf16.hip). Own translation unit: the kernels are long unrolled instruction streams (minutes of compile time).
#include "conv3d_lp.h"
#include "act_io.h"

// =====================================================================================================================================
// Plane-ring (z-marching), WEIGHTS-STATIONARY form for the 16-bit single-product modes (MI355_PREC_BF16 / MI355_PREC_F16) with <= 32
// input and exactly 32 output channels -- the 32 -> 32 layers of the 128^3 level are the largest group of launches of BASELINE configs[2].
// Written at the end of round 3; default for the shapes it takes since its first measurement (plan_lp_zring below).
// conv3d_k3_bf16 above runs at 20 % matrix-pipe utilisation on these layers: a 4 x 4 x 16 voxel tile is ONE staging round (load ->
// convert -> LDS -> barrier) followed by 1.4 us of MFMAs and an epilogue, and its weight fragments stream from L1 (one 1 KB fragment per
// 32-cycle MFMA and wave would be twice the L1 bandwidth of a CU at full matrix rate). Here a 256-thread workgroup owns an 8 (y) x 16 (x)
// voxel column and marches it along z:
//   * one wave per SIMD owns a lane's whole register file: the 27 x J weight fragments of the layer (216 registers at 32 input
//     channels) are loaded ONCE per workgroup and pinned in AGPRs, from where the MFMA reads its B operand directly;
//   * LDS holds a ring of 4 haloed input planes (10 x 18 voxels x 32 channels as 16-bit, 14.4 KB each): output plane z reads planes
//     z - 1, z, z + 1 (27 x J conflict-free ds_read_b128 per wave) while plane z + 2 is converted and written; every input voxel is
//     fetched once per column (halo 1.4x, no z re-reads: 2.5x in the tile form);
//   * one barrier per output plane; the global loads of plane z + 3 are issued a whole step ahead (two register sets), the residual /
//     normalised-tensor reads of the epilogue at the start of the step that consumes them;
//   * the epilogue of plane z - 1 (two accumulator sets), the conversion of plane z + 2 and the MFMAs of plane z are ONE basic block,
//     interleaved by scheduler directives -- with one wave per SIMD nothing else hides a latency. Fused statistics accumulate in
//     registers over the whole z range: one record per (z range, column).
// Wave w owns the M tile of rows 2 w, 2 w + 1 (32 voxels, the conflict-free lane -> voxel map of mtile_lane). Interior columns only
// (H % 8 == 0, W % 16 == 0, plain un-windowed output): the dispatcher keeps the tile kernel for everything else.
// TA: storage type of x, y, the residual and the normalised tensor of the norm-backward sums (act_io.h; bf16 storage with bf16 operands only).
template <int J, int INMODE, int FUSE, bool F16, typename TA = float>
__global__ __launch_bounds__(256) ONE_WAVE_PER_SIMD void conv3d_k3_lp_zring(ConvBArgs a) {
  constexpr bool LPS = is_lp16<TA>::value;      // 16-bit storage: a staging unit is one 16-byte run of 8 channels
  static_assert(!LPS || lp_storage_is_operand<TA, F16>::value, "16-bit storage goes with operands of its own type");
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HVP = HY * HX;      // haloed plane: 180 voxels
  constexpr int OCT = 2 * J;                 // channel octets of the (padded) input: 2 (16 channels) or 4 (32)
  constexpr int VSQ = OCT + 1;               // voxel stride in 16-byte units (odd)
  constexpr int PLANE = HVP * VSQ;           // uint4 per ring slot
  constexpr int UNITS = HVP * OCT;           // staging units (halo voxel, octet) per plane
  constexpr int UP = (UNITS + 255) / 256;    // per thread
  constexpr int NM = 27 * J;                 // MFMAs per output plane and wave
  static_assert(256 % OCT == 0, "a thread stages one fixed channel octet");
  static_assert(9 * UP + 16 <= NM, "the pieces of a step (conversion, unit stores, 16 output values) must fit its MFMAs");
  DYN_LDS(lds_f);
  uint4* lds = reinterpret_cast<uint4*>(lds_f);
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  int b = blockIdx.x;
  const int tx0 = (b % a.tilesX) * TX; b /= a.tilesX;
  const int ty0 = (b % a.tilesY) * TY; b /= a.tilesY;
  const int zs = b % a.zsplits; b /= a.zsplits;
  const int n = b;
  const int zb = zs * a.zper, ze = zb + a.zper < a.Do ? zb + a.zper : a.Do;

  // ---- weights: all 27 x J fragments of this lane, once, pinned in the accumulation registers ----
  u32x4_t bw[NM];                              // (a native vector type: the register-class pin does not take HIP's uint4 struct)
  {
    const int CQ8 = a.CinP / 8;
    const uint4* wl = a.wp + (size_t)half * a.CoutP + li;
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      bw[i] = __builtin_bit_cast(u32x4_t, wl[(size_t)((i / J) * CQ8 + 2 * (i % J)) * a.CoutP]);
      PIN_IN_AGPR(bw[i]);
    }
  }

  // ---- staging units of this thread: (halo voxel, octet so); geometry fixed for the column ----
  const int so = tid % OCT;
  const int c = 8 * so;
  const bool v0ok = c < a.Cin, v1ok = c + 4 < a.Cin;
  unsigned uoff[UP];                         // float offset inside an input plane (clamped, always valid)
  bool uin[UP];                              // inside the volume in y and x
  int ulds[UP];                              // uint4 offset inside a ring slot
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    const int u = tid + 256 * k;
    const int hv = u < UNITS ? u / OCT : HVP - 1;           // threads beyond the unit count repeat the last voxel's unit (identical store)
    const int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
    uin[k] = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
    const int iyc = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), ixc = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
    uoff[k] = (unsigned)((iyc * a.Wi + ixc) * a.xld);
    ulds[k] = hv * VSQ + so;
  }
  const int c0q = v0ok ? c : 0, c1q = v1ok ? c + 4 : c0q;
  float sc[8], sh[8], sl[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; sl[e] = a.slope; }
  if (INMODE == MI355_IN_AFFINE_ACT) {
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      if (hq ? v1ok : v0ok) {
        const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c + 4 * hq);
        const float4 h4 = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c + 4 * hq);
        sc[4 * hq] = s4.x; sc[4 * hq + 1] = s4.y; sc[4 * hq + 2] = s4.z; sc[4 * hq + 3] = s4.w;
        sh[4 * hq] = h4.x; sh[4 * hq + 1] = h4.y; sh[4 * hq + 2] = h4.z; sh[4 * hq + 3] = h4.w;
        if (a.in_slope) {
          const float4 l4 = *reinterpret_cast<const float4*>(a.in_slope + c + 4 * hq);
          sl[4 * hq] = l4.x; sl[4 * hq + 1] = l4.y; sl[4 * hq + 2] = l4.z; sl[4 * hq + 3] = l4.w;
        }
      }
    }
  }
  const size_t xplane = (size_t)a.Hi * a.Wi * a.xld;
  const TA* xn = reinterpret_cast<const TA*>(a.x) + (size_t)n * a.Di * xplane;
  // plane p (may lie outside the volume: clamped address, zeroed at the conversion) -> register set
  auto loads = [&](float4 (&ld)[UP][2], int p) {
    const int pc = p < 0 ? 0 : (p < a.Di ? p : a.Di - 1);
    const TA* base = xn + (size_t)pc * xplane;                // workgroup-uniform
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      if constexpr (LPS) {                                   // 8 packed bf16 in set element 0 (element 1 stays unused)
        const uint2 r0 = *reinterpret_cast<const uint2*>(base + uoff[k] + c0q), r1 = *reinterpret_cast<const uint2*>(base + uoff[k] + c1q);
        ld[k][0] = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y));
      } else {
        ld[k][0] = *reinterpret_cast<const float4*>(base + uoff[k] + c0q);
        ld[k][1] = *reinterpret_cast<const float4*>(base + uoff[k] + c1q);
      }
    }
  };
  // conversion of one staged element (norm + activation, mask) / of a unit's eight elements into its ring slot. The step below
  // spreads these over the MFMAs of a plane one element at a time; the prologue runs a whole plane at once (`commit`).
  auto conv_elem = [&](const float4 (&ld)[UP][2], int k, int e, bool pin) -> float {
    float v;
    if constexpr (LPS) {
      const float4 q = ld[k][0];
      const unsigned w = __float_as_uint((e >> 1) == 0 ? q.x : (e >> 1) == 1 ? q.y : (e >> 1) == 2 ? q.z : q.w);
      v = (e & 1) ? lp_hi<TA>(w) : lp_lo<TA>(w);
    } else {
      const float4 q = ld[k][e >> 2];
      v = (e & 3) == 0 ? q.x : (e & 3) == 1 ? q.y : (e & 3) == 2 ? q.z : q.w;
    }
    if (INMODE == MI355_IN_AFFINE_ACT) {
      const float u = v * sc[e] + sh[e];
      v = fmaxf(u, u * sl[e]);
    }
    return (pin && uin[k] && (e < 4 ? v0ok : v1ok)) ? v : 0.f;
  };
  auto store_unit = [&](const float (&v)[8], int k, auto slotc) {
    constexpr int SLOT = decltype(slotc)::value;
    uint4 pl[1];
    split8<1, F16>(v, pl);
    lds[SLOT * PLANE + ulds[k]] = pl[0];
  };
  auto commit = [&](const float4 (&ld)[UP][2], int p, auto slotc) {
    const bool pin = p >= 0 && p < a.Di;                      // workgroup-uniform
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = conv_elem(ld, k, e, pin);
      store_unit(v, k, slotc);
    }
  };

  // ---- this lane's A-operand position and its epilogue rows ----
  int lrow, ltx;
  mtile_lane(li, lrow, ltx);
  const int abase = ((2 * wave + lrow) * HX + ltx) * VSQ + half;
  const int co = li;                                          // Cout == 32 (dispatcher): every lane owns a real channel
  float bs = 0.f, cs = 1.f;
  if (a.bias) bs = a.bias[co];
  if (a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + co];
  float K0 = 0.f, s0 = 0.f, s1 = 0.f, gsc = 1.f, gsh = 0.f, gmean = 0.f, grstd = 1.f;
  bool first = true;
  if constexpr (FUSE == 2) {
    const int grp = co / (a.Cout / a.g.ggroups);
    gsc = a.g.gscale[(size_t)n * a.Cout + co]; gsh = a.g.gshift[(size_t)n * a.Cout + co];
    gmean = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2]; grstd = a.g.gmr[((size_t)n * a.g.ggroups + grp) * 2 + 1];
  }
  // Output plane z: accumulator register r is x position r of x-row ((0b0110 >> (r >> 2)) & 1) ^ half (mtile_lane inverted). Every
  // address of the epilogue is a wave-uniform base (plane z, this wave's row pair, x position r: scalar registers) plus ONE 32-bit
  // lane offset per x-row (the 16 + 16 + 16 per-lane 64-bit pointers of the first version cost 60 vector registers).
  const size_t vrow0 = (((size_t)n * a.Do) * a.Ho + ty0 + 2 * wave) * a.Wo + tx0;      // plane 0, x-row 0 of this wave's tile, x = 0
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const unsigned yoA = (unsigned)(half * a.Wo * a.yld + co), yoB = (unsigned)((half ^ 1) * a.Wo * a.yld + co);
  const unsigned roA = (unsigned)(half * a.Wo * a.resld + co), roB = (unsigned)((half ^ 1) * a.Wo * a.resld + co);
  const unsigned goA = (unsigned)(half * a.Wo * a.g.gxld + co), goB = (unsigned)((half ^ 1) * a.Wo * a.g.gxld + co);
  // the reads that do not depend on the MFMAs (residual; the normalised tensor of the norm-backward form) are requested at the start
  // of the step that runs the epilogue: 16 + 16 dword loads in flight under that step's MFMAs
  struct Side { float rs[16], gx[16]; };
  auto side_loads = [&](Side& sd, int z) {
    const size_t v0 = vrow0 + (size_t)z * oplane;             // wave-uniform
    if constexpr (FUSE == 2) {
      const TA* gb = reinterpret_cast<const TA*>(a.g.gx) + v0 * a.g.gxld;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.gx[r] = ld1(gb + (size_t)r * a.g.gxld + (((0x6 >> (r >> 2)) & 1) ? goB : goA));
    }
    if (a.res) {                                             // workgroup-uniform
      const TA* rb = reinterpret_cast<const TA*>(a.res) + v0 * a.resld;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = ld1(rb + (size_t)r * a.resld + (((0x6 >> (r >> 2)) & 1) ? roB : roA));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = 0.f;
    }
  };
  auto epilogue_value = [&](const f32x16& acc, const Side& sd, TA* yb, int r) {      // yb: wave-uniform base of the plane
    float v = (acc[r] + bs + sd.rs[r]) * cs;
    st1(yb + (size_t)r * a.yld + (((0x6 >> (r >> 2)) & 1) ? yoB : yoA), v);
    if constexpr (FUSE != 0) v = as_stored(yb, v);
    if constexpr (FUSE == 1) {
      if (first && r == 0) K0 = v;
      const float t = v - K0;
      s0 += t; s1 += t * t;
    } else if constexpr (FUSE == 2) {
      const float xv = sd.gx[r];
      const float u = xv * gsc + gsh;
      const float du = u > 0.f ? v : v * a.g.gslope;
      s0 += du; s1 += du * ((xv - gmean) * grstd);
    }
  };
  auto epilogue = [&](const f32x16& acc, const Side& sd, int z) {
    TA* yb = reinterpret_cast<TA*>(a.y) + (vrow0 + (size_t)z * oplane) * a.yld;
#pragma unroll
    for (int r = 0; r < 16; ++r) epilogue_value(acc, sd, yb, r);
    first = false;
  };

  // ---- prologue: planes zb - 1, zb, zb + 1 into ring slots 0, 1, 2 (plane q of this range lives in slot (q - zb + 1) & 3); plane
  //      zb + 2 requested ----
  float4 ldA[UP][2], ldB[UP][2];
  loads(ldA, zb - 1);
  loads(ldB, zb);
  commit(ldA, zb - 1, std::integral_constant<int, 0>());
  loads(ldA, zb + 1);
  commit(ldB, zb, std::integral_constant<int, 1>());
  loads(ldB, zb + 2);
  commit(ldA, zb + 1, std::integral_constant<int, 2>());
  __syncthreads();

  f32x16 accE, accO;                                         // output planes at even / odd distance from zb
#pragma unroll
  for (int r = 0; r < 16; ++r) { accE[r] = 0.f; accO[r] = 0.f; }
  // One output plane. R = (z - zb) & 3: its input planes z - 1, z, z + 1 sit in slots R, R + 1, R + 2 (mod 4), plane z + 2 (in the
  // register set `cur`) is written to slot R + 3, plane z + 3 is requested into `nxt`; `acc` takes plane z while the epilogue of plane
  // z - 1 (in `prev`) rides along.
  auto step = [&](int z, auto rc, auto hpc, float4 (&cur)[UP][2], float4 (&nxt)[UP][2], f32x16& acc, const f32x16& prev) {
    constexpr int R = decltype(rc)::value;
    constexpr bool HASPREV = decltype(hpc)::value;           // all but the first plane of the range
    loads(nxt, z + 3);
    Side sd;
    if constexpr (HASPREV) side_loads(sd, z - 1);
    const bool pin = z + 2 >= 0 && z + 2 < a.Di;             // plane z + 2 (in `cur`) exists; workgroup-uniform
    TA* yb = reinterpret_cast<TA*>(a.y) + (vrow0 + (size_t)(z - 1) * oplane) * a.yld;
    SCHED_BARRIER();
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // A fragments: a ring of 4, requested three MFMAs ahead. After MFMA i one PIECE of the other work of the step is issued, pinned in
    // place (one wave per SIMD: whatever is not between two MFMAs idles the matrix pipe): pieces 0..23 convert one staged element of
    // plane z + 2 each, 24..26 pack and write its three units, 27..42 are the 16 output values of plane z - 1.
    auto afrag = [&](int i) {
      const int tap = i / J, j = i % J;
      const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
      return lds[abase + ((R + dz) & 3) * PLANE + (dy * HX + dx) * VSQ + 2 * j];
    };
    uint4 af[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) af[i] = afrag(i);
    float cv[UP][8];
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (i + 3 < NM) af[(i + 3) & 3] = afrag(i + 3);
      acc = mfma_lp<F16>(af[i & 3], __builtin_bit_cast(uint4, bw[i]), acc);
      if (i < 8 * UP) cv[i / 8][i % 8] = conv_elem(cur, i / 8, i % 8, pin);
      else if (i < 8 * UP + UP) store_unit(cv[i - 8 * UP], i - 8 * UP, std::integral_constant<int, (R + 3) & 3>());
      else if (HASPREV && i < 9 * UP + 16) epilogue_value(prev, sd, yb, i - 9 * UP);
      SCHED_BARRIER();
    }
    if constexpr (HASPREV) first = false;
    __syncthreads();
  };
  // first plane of the range (slot phase R = 0, no previous plane), then the rest with the phase cycling 1, 2, 3, 0
  step(zb, std::integral_constant<int, 0>(), std::false_type(), ldB, ldA, accE, accO);
  for (int z = zb + 1; z < ze; z += 4) {
    step(z, std::integral_constant<int, 1>(), std::true_type(), ldA, ldB, accO, accE);
    if (z + 1 >= ze) break;
    step(z + 1, std::integral_constant<int, 2>(), std::true_type(), ldB, ldA, accE, accO);
    if (z + 2 >= ze) break;
    step(z + 2, std::integral_constant<int, 3>(), std::true_type(), ldA, ldB, accO, accE);
    if (z + 3 >= ze) break;
    step(z + 3, std::integral_constant<int, 0>(), std::true_type(), ldB, ldA, accE, accO);
  }
  {                                                          // the last plane's epilogue (nothing left to hide it under)
    Side sd;
    side_loads(sd, ze - 1);
    if ((ze - 1 - zb) & 1) epilogue(accO, sd, ze - 1); else epilogue(accE, sd, ze - 1);
  }

  if constexpr (FUSE != 0) {
    constexpr int K = FUSE == 1 ? 3 : 2;
    float vals[1][K];
    const int cnt = (ze - zb) * 16;
    if constexpr (FUSE == 1) {
      const float cf = (float)cnt;
      const float m2 = s1 - s0 * s0 / cf;
      vals[0][0] = cf; vals[0][1] = s0 + cf * K0; vals[0][2] = m2 > 0.f ? m2 : 0.f;
    } else {
      vals[0][0] = s0; vals[0][1] = s1;
    }
    const size_t rec = (size_t)n * ((size_t)a.zsplits * a.tilesY * a.tilesX) + ((size_t)zs * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX;
    float* dst = (FUSE == 1 ? a.g.mom : a.g.gnb) + rec * a.Cout * K;
    gn_fuse_reduce_store<K, 1, 4, 1>(vals, lds_f, wave, 0, half, li, tid, dst, 0, a.Cout);
  }
}

// =====================================================================================================================================
// Plane ring, second form (round 4): INPUT-PLANE-MAJOR with three output planes in flight, and the input channels SPLIT OVER THE WAVES.
//
// conv3d_k3_lp_zring above walks OUTPUT planes: plane z reads the three input planes z - 1, z, z + 1 from the LDS ring, one 1 KB A
// fragment per MFMA and wave -- at one 32-output-channel tile per workgroup that is 128 B/clk per CU = the whole LDS bandwidth at full
// matrix rate (tools/NEXT.md), and its SQ counters (profiles/r4_start_sq_counters_bf16_zring.txt) show the matrix pipe 45 % busy. Here a
// step consumes ONE input plane p: every A fragment (tile, dy, dx, k-step) is read once and multiplies the three z-taps into the three
// output planes that see plane p -- p + 1 (dz = 0), p (dz = 1), p - 1 (dz = 2) -- so the LDS delivers one fragment per THREE MFMAs,
// consecutive MFMAs never share an accumulator, and the ring shrinks to two slots (the plane being read, the plane being written).
// Plane p - 1 is complete after step p: its accumulators go to a 4 KB-per-tile LDS result buffer (double-buffered by step parity) and
// its epilogue rides under the MFMAs of step p + 1, reading the values back four at a time (no register copy of the finished tile).
//
// Layers wider than 32 input channels do not fit the weights-stationary scheme with every wave holding the whole filter (27 taps x
// Cin / 16 fragments of 4 registers: 432 registers at 64 channels). KS = 2 splits the channels over wave PAIRS instead: wave (ks, mg)
// holds the 27 x 2 fragments of channels [32 ks, 32 ks + 32) (216 AGPRs, as the 32-channel form) and accumulates the two M tiles of
// its pair over that half of K; both partial tiles of a finished plane go to the result buffer (same lane map on both sides: no shuffle)
// and each partner finishes ONE tile: its own partial plus the partner's. Per wave and step 108 MFMAs (3456 matrix cycles) against 36
// A-fragment reads, 6 staged units and one 16-value epilogue. Measured (profiles/r4_bf16_zring2.txt, r4_mfma_valu_overlap.txt): 38-47 %
// MFMA busy -- with one wave per SIMD the wave's own vector-ALU / LDS / memory instructions add to its MFMA time instead of hiding under it.
//   J  : k-steps (16 channels) per tap of a wave's channel slice (2)        KS : channel slices = M tiles per wave (1 | 2)
// Output channels: one 32-channel tile per workgroup (blockIdx carries the tile), Cout a multiple of 32.
// TA: storage type of x, y and the residual (act_io.h). bf16 storage (bf16 operands only): a staging unit is ONE 16-byte run of 8 channels
// (a plain input goes to the ring as loaded, no conversion), outputs are rounded once on store, moments are taken over the values as stored.
template <int J, int KS, int INMODE, int FUSE, bool F16, typename TA = float>
__global__ __launch_bounds__(256) ONE_WAVE_PER_SIMD void conv3d_k3_lp_zring2(ConvBArgs a) {
  constexpr bool LPS = is_lp16<TA>::value;      // 16-bit storage
  static_assert(!LPS || lp_storage_is_operand<TA, F16>::value, "16-bit storage goes with operands of its own type");
  constexpr int TY = 8, TX = 16, HY = TY + 2, HX = TX + 2, HVP = HY * HX;      // haloed plane: 180 voxels
  constexpr int MT = KS;                     // M tiles (two x-rows of 16 voxels) per wave
  constexpr int CT = KS * 16 * J;            // channels of a staged plane (the padded input)
  constexpr int OCT = CT / 8;                // channel octets per voxel
  constexpr int VSQ = OCT + 1;               // voxel stride in 16-byte units (odd: the 16 x positions of a lane group hit 16 distinct slots)
  constexpr int PLANE = HVP * VSQ;           // uint4 per ring slot
  constexpr int UNITS = HVP * OCT;           // staging units (halo voxel, octet) per plane
  constexpr int UP = (UNITS + 255) / 256;    // per thread
  constexpr int NW = 27 * J;                 // weight fragments of a wave
  constexpr int XCH = 4 * MT * 64 * 4;       // uint4 per result buffer: 4 waves x MT finished / partial tiles x 64 lanes x 16 floats
  constexpr int NP = 4 * UP + 1 + 20;        // pieces of a step: 4 pair conversions per unit, the next plane's loads, 4 result reads + 16 output values
  static_assert(256 % OCT == 0, "a thread stages one fixed channel octet");
  DYN_LDS(lds_f);
  uint4* lds = reinterpret_cast<uint4*>(lds_f);
  uint4* xch = lds + 2 * PLANE;                               // finished planes: [buffer][wave][local tile][q][lane]
  float* prm = reinterpret_cast<float*>(xch + 2 * XCH);       // norm prologue of this sample (KS > 1): scale | shift | slope, CT each
  const int tid = threadIdx.x, lane = tid & 63, wave = WAVE_UNIFORM(tid >> 6), half = lane >> 5, li = lane & 31;
  // Fused statistics: a lane's running sums live in LDS, not in registers: two loop-carried floats through the twelve unrolled step
  // bodies tipped hipcc's allocation of this kernel over (accumulators moved to the AGPRs, which are full of weights: 40-400 spilled
  // registers, reloaded in front of MFMAs; measured with tools/lpz_one.sh). A step sums its 16 values in two step-local registers and
  // adds them to the lane's own LDS slot once (a plain read-modify-write: nobody else touches the slot). The first version issued two
  // ds_add_f32 per VALUE: 8192 LDS atomics per plane and workgroup were slower than the plane's MFMAs (64 -> 64 @64^3: 0.46 ms).
  float* stat = prm + 3 * CT + 3 * tid;                       // [thread]: sum 0 | sum 1 | shift K0
  const int ks = wave % KS, mg = wave / KS;                   // channel slice, M group
  // workgroup -> (column, z range, sample, channel tile). Consecutive workgroup ids go to different XCDs: where the numbers divide, XCD x
  // takes a contiguous range of (column, z range) items -- neighbouring columns share halo voxels -- and all channel tiles of an item
  // (they read the same input) run on the same XCD.
  int cot, item;
  {
    const int nct = a.coTiles, G = gridDim.x, items = G / nct;
    int b = blockIdx.x;
    if (items % 8 == 0) {
      const int x = b & 7, r = b >> 3;                        // r-th workgroup of XCD x
      cot = r % nct;
      item = x * (items / 8) + r / nct;
    } else {
      cot = b % nct; item = b / nct;
    }
  }
  const int tx0 = (item % a.tilesX) * TX; item /= a.tilesX;
  const int ty0 = (item % a.tilesY) * TY; item /= a.tilesY;
  const int zs = item % a.zsplits; item /= a.zsplits;
  const int n = item;
  const int zb = zs * a.zper, ze = zb + a.zper < a.Do ? zb + a.zper : a.Do;
  const int nq = ze - zb + 2;                                 // steps: input planes zb - 1 .. ze
  const int co = cot * 32 + li;                               // Cout is a multiple of 32: every lane owns a real channel

  // ---- weights: the 27 x J fragments of this wave's channel slice, once, pinned in the accumulation registers ----
  u32x4_t bw[NW];
  {
    const int CQ8 = a.CinP / 8;                               // octets of the PACKED weights (cin rounded up to 16)
    const uint4* wl = a.wp + (size_t)half * a.CoutP + co;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const int kq = ks * J + (i % J);                        // k-step inside the padded input
      const bool kin = 16 * kq < a.CinP;                      // wave-uniform; a k-step beyond the packed channels re-reads k-step 0 and is masked
      const unsigned km = kin ? 0xffffffffu : 0u;             // (no branch around a load: all fragments are in flight together)
      u32x4_t w = __builtin_bit_cast(u32x4_t, wl[(size_t)((i / J) * CQ8 + (kin ? 2 * kq : 0)) * a.CoutP]);
      w.x &= km; w.y &= km; w.z &= km; w.w &= km;
      bw[i] = w;
      PIN_IN_AGPR(bw[i]);
    }
  }

  // ---- staging units of this thread: (halo voxel, octet so); geometry fixed for the column ----
  const int so = tid % OCT;
  const int c = 8 * so;
  const bool v0ok = c < a.Cin, v1ok = c + 4 < a.Cin;
  unsigned uoff[UP];
  bool uin[UP];
  int ulds[UP];
#pragma unroll
  for (int k = 0; k < UP; ++k) {
    const int u = tid + 256 * k;
    const int hv = u < UNITS ? u / OCT : HVP - 1;             // threads beyond the unit count repeat the last voxel's unit (identical store)
    const int iy = ty0 - 1 + hv / HX, ix = tx0 - 1 + hv % HX;
    uin[k] = iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
    const int iyc = iy < 0 ? 0 : (iy < a.Hi ? iy : a.Hi - 1), ixc = ix < 0 ? 0 : (ix < a.Wi ? ix : a.Wi - 1);
    uoff[k] = (unsigned)((iyc * a.Wi + ixc) * a.xld);
    ulds[k] = hv * VSQ + so;
  }
  const int c0q = v0ok ? c : 0, c1q = v1ok ? c + 4 : c0q;
  float sc[8], sh[8], sl[8];                                  // KS == 1: the norm prologue of this thread's octet in registers
#pragma unroll
  for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; sl[e] = a.slope; }
  if (INMODE == MI355_IN_AFFINE_ACT) {
    if constexpr (KS == 1) {
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        if (hq ? v1ok : v0ok) {
          const float4 s4 = *reinterpret_cast<const float4*>(a.in_scale + (size_t)n * a.Cin + c + 4 * hq);
          const float4 h4 = *reinterpret_cast<const float4*>(a.in_shift + (size_t)n * a.Cin + c + 4 * hq);
          sc[4 * hq] = s4.x; sc[4 * hq + 1] = s4.y; sc[4 * hq + 2] = s4.z; sc[4 * hq + 3] = s4.w;
          sh[4 * hq] = h4.x; sh[4 * hq + 1] = h4.y; sh[4 * hq + 2] = h4.z; sh[4 * hq + 3] = h4.w;
          if (a.in_slope) {
            const float4 l4 = *reinterpret_cast<const float4*>(a.in_slope + c + 4 * hq);
            sl[4 * hq] = l4.x; sl[4 * hq + 1] = l4.y; sl[4 * hq + 2] = l4.z; sl[4 * hq + 3] = l4.w;
          }
        }
      }
    } else {
      for (int cc = tid; cc < CT; cc += 256) {                // through LDS: 24 registers per thread are not to spare at KS = 2
        const bool in = cc < a.Cin;
        prm[cc] = in ? a.in_scale[(size_t)n * a.Cin + cc] : 1.f;
        prm[CT + cc] = in ? a.in_shift[(size_t)n * a.Cin + cc] : 0.f;
      }
    }
  }
  const size_t xplane = (size_t)a.Hi * a.Wi * a.xld;
  const TA* xn = reinterpret_cast<const TA*>(a.x) + (size_t)n * a.Di * xplane;
  // Input planes in flight. One wave per SIMD and one workgroup per CU: nothing but the distance between a request and its first use
  // hides the HBM latency (~2 us under load, a step is 2-4 us). KS = 2 has registers for ONE set: requested mid-step, consumed in the
  // first half of the next step (half a step of lead). KS = 1 has 100 registers to spare: a second set `ldn` holds the plane after
  // that; mid-step it moves into `ld` (24 v_mov) and the request after next goes out -- a whole step of lead.
  constexpr bool TWOSETS = KS == 1;
  float4 ld[UP][2], ldn[TWOSETS ? UP : 1][2];
  auto loads_into = [&](float4 (&dst)[UP][2], int p) {        // plane p (may lie outside the volume: clamped address, zeroed at the conversion)
    const int pc = p < 0 ? 0 : (p < a.Di ? p : a.Di - 1);
    const TA* base = xn + (size_t)pc * xplane;                // workgroup-uniform
#pragma unroll
    for (int k = 0; k < UP; ++k) {
      if constexpr (LPS) {                                   // 8 packed bf16 in set element 0 (element 1 stays unused)
        const uint2 r0 = *reinterpret_cast<const uint2*>(base + uoff[k] + c0q), r1 = *reinterpret_cast<const uint2*>(base + uoff[k] + c1q);
        dst[k][0] = make_float4(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r1.x), __uint_as_float(r1.y));
      } else {
        dst[k][0] = *reinterpret_cast<const float4*>(base + uoff[k] + c0q);
        dst[k][1] = *reinterpret_cast<const float4*>(base + uoff[k] + c1q);
      }
    }
  };
  auto loads = [&](int p) { loads_into(ld, p); };
  auto loads_mid_step = [&](int p) {                          // after plane p + 1's conversions: `ld` is free
    if constexpr (TWOSETS) {
#pragma unroll
      for (int k = 0; k < UP; ++k) { ld[k][0] = ldn[k][0]; if constexpr (!LPS) ld[k][1] = ldn[k][1]; }      // plane p + 2 (requested a step ago)
      loads_into(ldn, p + 3);
    } else {
      loads_into(ld, p + 2);
    }
  };
  // two staged elements (2 e2, 2 e2 + 1) of unit k: norm + activation, mask, one packed dword of the operand type. KS = 1: the norm
  // parameters of the thread's octet sit in registers; KS = 2 (no registers to spare): scale / shift of the pair come from LDS, requested
  // one piece ahead (`pq`: two alternating sets -- a read issued in the piece that uses it would expose the LDS latency to the next
  // MFMA, one wave per SIMD), the slope is the descriptor's scalar (a per-channel slope vector keeps the tile kernel: plan_lp_zring).
  float2 pq[2][2];                                            // [set][scale | shift]
  auto param_read = [&](int e2, int set) {
    if constexpr (KS > 1 && INMODE == MI355_IN_AFFINE_ACT) {
      pq[set][0] = *reinterpret_cast<const float2*>(prm + c + 2 * e2);
      pq[set][1] = *reinterpret_cast<const float2*>(prm + CT + c + 2 * e2);
    }
  };
  auto conv_pair = [&](int k, int e2, bool pin, int set) -> unsigned {
    float v0, v1;
    if constexpr (LPS) {
      const float4 q = ld[k][0];
      const unsigned w = __float_as_uint(e2 == 0 ? q.x : e2 == 1 ? q.y : e2 == 2 ? q.z : q.w);      // channels 2 e2, 2 e2 + 1 as stored
      if constexpr (INMODE == MI355_IN_PLAIN) {              // the operand IS the stored value
        return (pin && uin[k] && (e2 < 2 ? v0ok : v1ok)) ? w : 0u;
      }
      v0 = lp_lo<TA>(w); v1 = lp_hi<TA>(w);
    } else {
      const float4 q = ld[k][e2 >> 1];
      v0 = (e2 & 1) ? q.z : q.x; v1 = (e2 & 1) ? q.w : q.y;
    }
    if (INMODE == MI355_IN_AFFINE_ACT) {
      float s0, s1, h0, h1, l0, l1;
      if constexpr (KS == 1) {
        s0 = sc[2 * e2]; s1 = sc[2 * e2 + 1]; h0 = sh[2 * e2]; h1 = sh[2 * e2 + 1]; l0 = sl[2 * e2]; l1 = sl[2 * e2 + 1];
      } else {
        s0 = pq[set][0].x; s1 = pq[set][0].y; h0 = pq[set][1].x; h1 = pq[set][1].y; l0 = l1 = a.slope;
      }
      const float u0 = v0 * s0 + h0, u1 = v1 * s1 + h1;
      v0 = fmaxf(u0, u0 * l0); v1 = fmaxf(u1, u1 * l1);
    }
    const unsigned m = (pin && uin[k] && (e2 < 2 ? v0ok : v1ok)) ? 0xffffffffu : 0u;      // (a mask, not a select: no branch around the arithmetic)
    return pack_lp2<F16>(v0, v1) & m;
  };

  // ---- this wave's M tiles: local tile 0 is the one it finishes (global tile g0), local tile 1 (KS = 2) its partner's ----
  int lrow, ltx;
  mtile_lane(li, lrow, ltx);
  const int g0 = mg * MT + ks;
  int abase[MT];
#pragma unroll
  for (int l = 0; l < MT; ++l) {
    const int g = l == 0 ? g0 : mg * MT + (1 - ks);
    abase[l] = ((2 * g + lrow) * HX + ltx) * VSQ + half + 2 * J * ks;
  }
  float bs = 0.f, cs = 1.f;
  if (a.bias) bs = a.bias[co];
  if (a.out_chscale) cs = a.out_chscale[(size_t)n * a.Cout + co];
  static_assert(FUSE == 0 || FUSE == 1, "no norm-backward form (see plan_lp_zring)");
  if constexpr (FUSE != 0) { stat[0] = 0.f; stat[1] = 0.f; stat[2] = 0.f; }
  // epilogue addresses: wave-uniform base (plane z, this wave's row pair, x position r) + ONE 32-bit lane offset per x-row
  const size_t vrow0 = (((size_t)n * a.Do) * a.Ho + ty0 + 2 * g0) * a.Wo + tx0;
  const size_t oplane = (size_t)a.Ho * a.Wo;
  const unsigned yoA = (unsigned)(half * a.Wo * a.yld + co), yoB = (unsigned)((half ^ 1) * a.Wo * a.yld + co);
  const unsigned roA = (unsigned)(half * a.Wo * a.resld + co), roB = (unsigned)((half ^ 1) * a.Wo * a.resld + co);
  struct Side { float rs[16]; };
  auto side_loads = [&](Side& sd, int z) {
    const size_t v0 = vrow0 + (size_t)z * oplane;             // wave-uniform
    if (a.res) {                                             // workgroup-uniform
      const TA* rb = reinterpret_cast<const TA*>(a.res) + v0 * a.resld;
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = ld1(rb + (size_t)r * a.resld + (((0x6 >> (r >> 2)) & 1) ? roB : roA));
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) sd.rs[r] = 0.f;
    }
  };
  auto epilogue_value = [&](float accv, const Side& sd, TA* yb, int r, float K0, float& ps0, float& ps1) {      // yb: wave-uniform base of the plane
    float v = (accv + bs + sd.rs[r]) * cs;
    st1(yb + (size_t)r * a.yld + (((0x6 >> (r >> 2)) & 1) ? yoB : yoA), v);
    if constexpr (FUSE == 1) {
      v = as_stored(yb, v);
      const float t = v - K0;                                // K0: this lane's shift (set once when the first plane completes)
      ps0 += t; ps1 += t * t;
    }
  };

  // ---- prologue: input plane zb - 1 into slot 0, plane zb requested ----
  loads(zb - 1);
  if (INMODE == MI355_IN_AFFINE_ACT && KS > 1) __syncthreads();      // the norm parameters are in LDS
  {
    const bool pin = zb - 1 >= 0;
    unsigned* l32 = reinterpret_cast<unsigned*>(lds);
#pragma unroll
    for (int k = 0; k < UP; ++k)
#pragma unroll
      for (int e2 = 0; e2 < 4; ++e2) { param_read(e2, 0); l32[4 * ulds[k] + e2] = conv_pair(k, e2, pin, 0); }
  }
  loads(zb);
  if constexpr (TWOSETS) loads_into(ldn, zb + 1);
  __syncthreads();

  f32x16 acc[3][MT];                                         // output planes (o - zb) mod 3
  // One step = input plane p = zb - 1 + q from slot q & 1. R = q % 3: the plane adds into Z = acc[R] (output p + 1, first contribution),
  // Y = acc[R + 2] (output p) and X = acc[R + 1] (output p - 1, complete afterwards). MASK: which of the three outputs lie in the z range
  // (bit dz). HASPREV: output plane p - 2, finished by the previous step, waits in the result buffer of that step's parity: a wave's own
  // tile (local tile 0) and, for KS = 2, its partner's partial of the same tile -- the epilogue reads them back four values at a time.
  auto step = [&](int q, auto rc, auto mc, auto hpc) {
    constexpr int R = decltype(rc)::value, MASK = decltype(mc)::value;
    constexpr bool HASPREV = decltype(hpc)::value;
    constexpr int NZ = ((MASK >> 0) & 1) + ((MASK >> 1) & 1) + ((MASK >> 2) & 1);
    f32x16 (&Z)[MT] = acc[R];
    f32x16 (&Y)[MT] = acc[(R + 2) % 3];
    f32x16 (&X)[MT] = acc[(R + 1) % 3];
    const int p = zb - 1 + q;
    const uint4* cur = lds + (q & 1) * PLANE;
    uint4* nxt = lds + ((q + 1) & 1) * PLANE;
    Side sd;
    TA* yb = reinterpret_cast<TA*>(a.y) + (vrow0 + (size_t)(p - 2) * oplane) * a.yld;
    float K0 = 0.f, ps0 = 0.f, ps1 = 0.f;                 // the plane's partial sums (step-local)
    if constexpr (HASPREV) {
      side_loads(sd, p - 2);
      if constexpr (FUSE == 1) K0 = stat[2];
    }
    const uint4* rown = xch + ((q + 1) & 1) * XCH + ((wave * MT) * 4) * 64 + lane;                    // [q] at + 64 q
    const uint4* rpar = xch + ((q + 1) & 1) * XCH + (((wave ^ (KS - 1)) * MT + (MT - 1)) * 4) * 64 + lane;   // KS = 2: the partner's local tile 1
    const bool pin = p + 1 >= 0 && p + 1 < a.Di;             // plane p + 1 (in `ld`) exists; workgroup-uniform
    if constexpr (MASK & 1) {
#pragma unroll
      for (int l = 0; l < MT; ++l) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Z[l][r] = 0.f;
        PIN_IN_VGPR(Z[l]);                                   // the AGPRs belong to the weights
      }
    }
    SCHED_BARRIER();
    // pieces of the step's other work, one after every MFMA (one wave per SIMD: whatever is not between two MFMAs idles the matrix
    // pipe): 0 .. 5 UP - 1 convert plane p + 1 (four packed pairs, then the unit's LDS write), 5 UP requests plane p + 2, then the 16
    // output values of plane p - 2
    // one after every MFMA (one wave per SIMD: whatever is not between two MFMAs idles the matrix pipe). Order: 4 UP pair conversions
    // of plane p + 1 (each written to the ring as one dword), the request of plane p + 2, then plane p - 2's epilogue in groups of four
    // values, each group's result read two groups ahead: R0 R1 V0-3 R2 V4-7 R3 V8-11 V12-15.
    unsigned* nxt32 = reinterpret_cast<unsigned*>(nxt);
    param_read(0, 0);
    f32x4 ev[4];
    auto result_read = [&](int g) {
      const uint4 u = rown[g * 64];
      ev[g] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
      if constexpr (KS > 1) {
        const uint4 v = rpar[g * 64];
        ev[g] += f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
    };
    auto piece = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      if constexpr (I < 4 * UP) {
        constexpr int k = I / 4, e = I % 4;
        if constexpr (I + 1 < 4 * UP) param_read((I + 1) % 4, (I + 1) & 1);      // the next piece's parameters
        nxt32[4 * ulds[k] + e] = conv_pair(k, e, pin, I & 1);
      } else if constexpr (I == 4 * UP) {
        loads_mid_step(p);
      } else if constexpr (I < NP && HASPREV) {
        constexpr int E = I - 4 * UP - 1;                   // 0 .. 19
        //            R0  R1  V0 V1 V2 V3  R2  V4 V5 V6 V7  R3  V8 V9 V10 V11 V12 V13 V14 V15
        constexpr int tab[20] = {-1, -2, 0, 1, 2, 3, -3, 4, 5, 6, 7, -4, 8, 9, 10, 11, 12, 13, 14, 15};
        if constexpr (tab[E] < 0) result_read(-tab[E] - 1);
        else epilogue_value(ev[tab[E] / 4][tab[E] % 4], sd, yb, tab[E], K0, ps0, ps1);
      }
    };
    auto afrag = [&](auto ic) {                              // A fragment i = (dydx * J + j) * MT + l
      constexpr int I = decltype(ic)::value;
      constexpr int l = I % MT, j = (I / MT) % J, t = I / (MT * J), dy = t / 3, dx = t % 3;
      return cur[abase[l] + (dy * HX + dx) * VSQ + 2 * j];
    };
    constexpr int NA = 9 * J * MT;
    uint4 af[4];
    af[0] = afrag(std::integral_constant<int, 0>());
    af[1] = afrag(std::integral_constant<int, 1>());
    auto body = [&](auto ic) {
      constexpr int I = decltype(ic)::value;
      constexpr int l = I % MT, j = (I / MT) % J, t = I / (MT * J);
      if constexpr (I + 2 < NA) af[(I + 2) & 3] = afrag(std::integral_constant<int, I + 2>());
      if constexpr (MASK & 1) {
        Z[l] = mfma_lp<F16>(af[I & 3], __builtin_bit_cast(uint4, bw[(0 * 9 + t) * J + j]), Z[l]);
        piece(std::integral_constant<int, I * NZ + 0>());
        SCHED_BARRIER();
      }
      if constexpr (MASK & 2) {
        Y[l] = mfma_lp<F16>(af[I & 3], __builtin_bit_cast(uint4, bw[(1 * 9 + t) * J + j]), Y[l]);
        piece(std::integral_constant<int, I * NZ + ((MASK & 1) ? 1 : 0)>());
        SCHED_BARRIER();
      }
      if constexpr (MASK & 4) {
        X[l] = mfma_lp<F16>(af[I & 3], __builtin_bit_cast(uint4, bw[(2 * 9 + t) * J + j]), X[l]);
        piece(std::integral_constant<int, I * NZ + NZ - 1>());
        SCHED_BARRIER();
      }
    };
    static_for<0, NA>(body);
    static_for<NA * NZ, NP>(piece);                          // pieces the MFMAs of a short step (first / last planes of the range) did not cover
    // moments: sums about a per-lane shift close to the data (no E[x^2] - E[x]^2 cancellation); any value near the lane's outputs
    // serves -- the first accumulator value of the first finished plane, straight from the register (no residual, for KS = 2 this wave's
    // partial only). Set in the one step that completes the first plane: no run-time flag, no branch in the epilogue pieces.
    if constexpr (FUSE != 0 && HASPREV) { stat[0] += ps0; stat[1] += ps1; }
    if constexpr (FUSE == 1 && MASK == 7 && !HASPREV) stat[2] = as_stored(reinterpret_cast<TA*>(a.y), (X[0][0] + bs) * cs);
    if constexpr (MASK & 4) {                                // output plane p - 1 is complete: its accumulators -> this step's result buffer
      uint4* px = xch + (q & 1) * XCH + ((wave * MT) * 4) * 64 + lane;
#pragma unroll
      for (int l = 0; l < MT; ++l)
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4)
          px[(l * 4 + t4) * 64] = make_uint4(__float_as_uint(X[l][4 * t4]), __float_as_uint(X[l][4 * t4 + 1]), __float_as_uint(X[l][4 * t4 + 2]),
                                             __float_as_uint(X[l][4 * t4 + 3]));
    }
    __syncthreads();
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using M1 = std::integral_constant<int, 1>; using M3 = std::integral_constant<int, 3>; using M7 = std::integral_constant<int, 7>;
  using M6 = std::integral_constant<int, 6>; using M4 = std::integral_constant<int, 4>;
  // head (the range has >= 4 planes: plan_lp_zring2): q = 0 feeds output zb only, q = 1 outputs zb + 1 and zb, q = 2 is the first full step
  step(0, I0(), M1(), std::false_type());
  step(1, I1(), M3(), std::false_type());
  step(2, I2(), M7(), std::false_type());
  int q = 3;
  while (q <= nq - 3) {
    step(q, I0(), M7(), std::true_type()); if (++q > nq - 3) break;
    step(q, I1(), M7(), std::true_type()); if (++q > nq - 3) break;
    step(q, I2(), M7(), std::true_type()); ++q;
  }
  // tail: q = nq - 2 (outputs ze - 1 and ze - 2), q = nq - 1 (output ze - 1)
  switch (q % 3) {
    case 0: step(q, I0(), M6(), std::true_type()); step(q + 1, I1(), M4(), std::true_type()); break;
    case 1: step(q, I1(), M6(), std::true_type()); step(q + 1, I2(), M4(), std::true_type()); break;
    default: step(q, I2(), M6(), std::true_type()); step(q + 1, I0(), M4(), std::true_type()); break;
  }
  {                                                          // the last plane's epilogue (nothing left to hide it under)
    Side sd;
    side_loads(sd, ze - 1);
    const uint4* rown = xch + ((nq - 1) & 1) * XCH + ((wave * MT) * 4) * 64 + lane;
    const uint4* rpar = xch + ((nq - 1) & 1) * XCH + (((wave ^ (KS - 1)) * MT + (MT - 1)) * 4) * 64 + lane;
    TA* yb = reinterpret_cast<TA*>(a.y) + (vrow0 + (size_t)(ze - 1) * oplane) * a.yld;
    const float K0 = FUSE == 1 ? stat[2] : 0.f;
    float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const uint4 u = rown[g * 64];
      f32x4 e4 = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w)};
      if constexpr (KS > 1) {
        const uint4 v = rpar[g * 64];
        e4 += f32x4{__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) epilogue_value(e4[r], sd, yb, 4 * g + r, K0, ps0, ps1);
    }
    if constexpr (FUSE != 0) { stat[0] += ps0; stat[1] += ps1; }
  }

  if constexpr (FUSE == 1) {
    // One moments record per (z range, column, wave, half-wave) and channel: every lane stores the partial statistics of its 16 (z range)
    // values itself -- no merge through LDS, no barrier, no branch after the main loop (the merge tree of gn_fuse_reduce_store behind this
    // kernel's register allocation cost hundreds of spilled registers); mi355_gn_records_reduce folds the records (ops.Backend._fold_records).
    const size_t col = (size_t)n * ((size_t)a.zsplits * a.tilesY * a.tilesX) * 8 + (((size_t)zs * a.tilesY + ty0 / TY) * a.tilesX + tx0 / TX) * 8;
    float* dst = a.g.mom + ((col + wave * 2 + half) * a.Cout + co) * 3;
    const float s0 = stat[0], s1 = stat[1];
    const float cf = (float)((ze - zb) * 16);
    const float m2 = s1 - s0 * s0 / cf;
    dst[0] = cf; dst[1] = s0 + cf * stat[2]; dst[2] = m2 > 0.f ? m2 : 0.f;
  }
}

#ifndef LPZ_NO_LAUNCH      // (developer builds of single instantiations: tools/lpz_one.sh)
// ---- launches (called by mi355_conv3d_fwd_bf16_impl, conv3d_bf16.hip) ----
int mi355_lp_zring_launch(ConvBArgs& a, int in_mode, int fuse, bool f16, bool lps, long long blocks, void* stream) {
  const bool norm = in_mode == MI355_IN_AFFINE_ACT;
  const int lds_bytes = 4 * 180 * 5 * 16;                   // ring of 4 planes, 180 voxels, 5 x 16 bytes per voxel (32 channels + pad)
  const dim3 grid((unsigned)blocks), blk(256);
#define LPZ_LAUNCH_T(JJ, IM, FU, HF, TT)                                                                    \
  do { SET_MAX_DYN_LDS((conv3d_k3_lp_zring<JJ, IM, FU, HF, TT>), lds_bytes);                              \
       LAUNCH((conv3d_k3_lp_zring<JJ, IM, FU, HF, TT>), grid, blk, lds_bytes, stream, a); } while (0)
#define LPZ_LAUNCH(JJ, IM, FU, HF)                                                                          \
  do { if (lps) { if constexpr (HF) LPZ_LAUNCH_T(JJ, IM, FU, true, f16_t); else LPZ_LAUNCH_T(JJ, IM, FU, false, bf16_t); break; } \
       LPZ_LAUNCH_T(JJ, IM, FU, HF, float); } while (0)
#define LPZ_FUSE(JJ, HF)                                                                                    \
  do { if (fuse == 1) { if (norm) LPZ_LAUNCH(JJ, MI355_IN_AFFINE_ACT, 1, HF); else LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 1, HF); } \
       else if (fuse == 2) LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 2, HF);                                         \
       else if (norm) LPZ_LAUNCH(JJ, MI355_IN_AFFINE_ACT, 0, HF); else LPZ_LAUNCH(JJ, MI355_IN_PLAIN, 0, HF); } while (0)
  if (f16) LPZ_FUSE(2, true); else LPZ_FUSE(2, false);
#undef LPZ_FUSE
#undef LPZ_LAUNCH
#undef LPZ_LAUNCH_T
  return LAUNCH_CHECK();
}

int mi355_lp_zring2_launch(ConvBArgs& a, int ks, int in_mode, int fuse, bool f16, bool lps, long long wgs, void* stream) {
  const bool norm = in_mode == MI355_IN_AFFINE_ACT;
  const dim3 grid2((unsigned)wgs), blk2(256);
  // ring of 2 planes (180 voxels x (channels / 8 + 1) x 16 bytes) + two result buffers (4 waves x KS tiles x 4 KB) + KS = 2: the norm prologue
  const int lds1 = 2 * 180 * 5 * 16 + 2 * 4 * 1 * 64 * 4 * 16 + (3 * 32 + 3 * 256) * 4;      // ... + norm prologue slots + per-thread statistics
  const int lds2 = 2 * 180 * 9 * 16 + 2 * 4 * 2 * 64 * 4 * 16 + (3 * 64 + 3 * 256) * 4;
#define LPZ2_LAUNCH_T(KSV, IM, FU, HF, TT)                                                                    \
  do { const int lb = (KSV) == 1 ? lds1 : lds2;                                                            \
       SET_MAX_DYN_LDS((conv3d_k3_lp_zring2<2, KSV, IM, FU, HF, TT>), lb);                                 \
       LAUNCH((conv3d_k3_lp_zring2<2, KSV, IM, FU, HF, TT>), grid2, blk2, lb, stream, a); } while (0)
#define LPZ2_LAUNCH(KSV, IM, FU, HF)                                                                          \
  do { if (lps) { if constexpr (HF) LPZ2_LAUNCH_T(KSV, IM, FU, true, f16_t); else LPZ2_LAUNCH_T(KSV, IM, FU, false, bf16_t); break; } \
       LPZ2_LAUNCH_T(KSV, IM, FU, HF, float); } while (0)
#define LPZ2_FUSE(KSV, HF)                                                                                    \
  do { if (fuse == 1) { if (norm) LPZ2_LAUNCH(KSV, MI355_IN_AFFINE_ACT, 1, HF); else LPZ2_LAUNCH(KSV, MI355_IN_PLAIN, 1, HF); } \
       else if (fuse == 2) return MI355_EUNSUPPORTED;                                                      \
       else if (norm) LPZ2_LAUNCH(KSV, MI355_IN_AFFINE_ACT, 0, HF); else LPZ2_LAUNCH(KSV, MI355_IN_PLAIN, 0, HF); } while (0)
  // (no norm-backward form: plan_lp_zring routes those calls to conv3d_k3_lp_zring where it applies and answers the statistics query
  // with 0 otherwise -- the sums then take their own pass and the conv runs here with the plain epilogue)
  if (ks == 1) { if (f16) LPZ2_FUSE(1, true); else LPZ2_FUSE(1, false); }
  else { if (f16) LPZ2_FUSE(2, true); else LPZ2_FUSE(2, false); }
#undef LPZ2_FUSE
#undef LPZ2_LAUNCH
#undef LPZ2_LAUNCH_T
  return LAUNCH_CHECK();
}
#endif
