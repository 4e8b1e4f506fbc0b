// GroupNorm / InstanceNorm statistics fused into the epilogue of the conv that PRODUCES the normalised tensor (forward) or
// its gradient (backward): the reference's block is norm -> act -> conv (unet3d/models/pytorch/classification/myronenko.py:17-21),
// so every conv output is the next block's norm input, and every dgrad output is the gradient wrt an activated norm output.
// While the values still sit in the MFMA accumulators the workgroup reduces them over its spatial tile and writes one small
// partial record per (sample, tile, channel); the finalisation kernels in norm.hip combine the records in double. The
// standalone statistics passes over the tensor (1 read forward, 2 reads backward) disappear.
//
// Record formats (also the C ABI: mi355_conv_desc.moments_out / mi355_gn_bwd_fuse.partials_out):
//   moments  [n][B][C][3] = (count, sum, M2)  M2 = sum (v - sum/count)^2 over the tile's valid voxels (two passes over the
//            registers; tiles merge by Chan's formula, so no E[x^2] - E[x]^2 cancellation anywhere)
//   gn-bwd   [n][B][C][2] = (sum du, sum du * xhat)  plain sums
// Lane layout assumed (v_mfma_f32_32x32x* C/D map): lane (half, li) holds, per N tile nt, the rows
// (r & 3) + 8 * (r >> 2) + 4 * half of channel column li; the two half-waves therefore hold disjoint rows of the same channel.
// Everything is combined in a fixed order: bitwise reproducible run to run.
#pragma once

struct GnFuseArgs {
  float* mom;                 // moments_out or NULL
  float* gnb;                 // partials_out or NULL
  const float* gx; int gxld;  // gn-bwd: raw values of the normalised tensor
  const float* gscale; const float* gshift; const float* gmr; int ggroups; float gslope;
};

__device__ __forceinline__ void gn_mom_merge(float& ca, float& sa, float& qa, float cb, float sb, float qb) {
  // (ca, sa, qa) <- (ca, sa, qa) (+) (cb, sb, qb); either side may be empty
  const float ct = ca + cb;
  float q = qa + qb;
  if (ca > 0.f && cb > 0.f) {
    const float d = sb / cb - sa / ca;
    q += d * d * (ca * cb / ct);
  }
  ca = ct; sa += sb; qa = q;
}

// vals[NT][K]: this lane's K partial values per N tile (K = 3 moments, K = 2 plain sums), already merged over the lane's own rows.
// Merges the two half-waves (shuffle), then the WM waves that share an N range (through `lds`, >= 4 * NT * 32 * K floats, which the
// caller no longer needs: a barrier is issued first), and writes dst[co * K + k] for the channels this workgroup owns.
template <int K, int NT, int WM, int WN>
__device__ __forceinline__ void gn_fuse_reduce_store(float (&vals)[NT][K], float* lds, int wm, int wn, int half, int li, int tid,
                                                     float* dst, int co_wg_base, int Cout) {
  __syncthreads();                                   // every wave is done with the conv's LDS tile
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    float o[K];
#pragma unroll
    for (int k = 0; k < K; ++k) o[k] = __shfl_xor(vals[nt][k], 32);
    if (K == 3) {
      // fixed order: half 0's rows first
      float c = half ? o[0] : vals[nt][0], s = half ? o[1] : vals[nt][1], q = half ? o[2] : vals[nt][2];
      gn_mom_merge(c, s, q, half ? vals[nt][0] : o[0], half ? vals[nt][1] : o[1], half ? vals[nt][2] : o[2]);
      vals[nt][0] = c; vals[nt][1] = s; vals[nt][2] = q;
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) vals[nt][k] = half ? o[k] + vals[nt][k] : vals[nt][k] + o[k];
    }
    if (half == 0) {
      float* p = lds + (((wm * WN + wn) * NT + nt) * 32 + li) * K;
#pragma unroll
      for (int k = 0; k < K; ++k) p[k] = vals[nt][k];
    }
  }
  __syncthreads();
  if (tid < WN * NT * 32) {
    const int wn_ = tid / (NT * 32), nt_ = (tid / 32) % NT, l = tid & 31;
    float r[K];
    const float* p0 = lds + (((0 * WN + wn_) * NT + nt_) * 32 + l) * K;
#pragma unroll
    for (int k = 0; k < K; ++k) r[k] = p0[k];
#pragma unroll
    for (int w = 1; w < WM; ++w) {
      const float* p = lds + (((w * WN + wn_) * NT + nt_) * 32 + l) * K;
      if (K == 3) gn_mom_merge(r[0], r[1], r[2], p[0], p[1], p[2]);
      else {
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] += p[k];
      }
    }
    const int co = co_wg_base + wn_ * (32 * NT) + nt_ * 32 + l;
    if (co < Cout) {
#pragma unroll
      for (int k = 0; k < K; ++k) dst[(size_t)co * K + k] = r[k];
    }
  }
}
