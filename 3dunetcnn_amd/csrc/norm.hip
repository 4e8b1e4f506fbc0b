// GroupNorm / InstanceNorm statistics and the backward of act(GroupNorm(x)) -- HBM-bound streaming kernels.
//
// Replaces torch.nn.GroupNorm(G, C, eps=1e-5, affine) as built by
// unet3d/models/pytorch/classification/myronenko.py:23-31 (G = 8, or G = C when C < 8 or C % 8 != 0), and
// InstanceNorm3d(affine) of MONAI DynUNet (G == C). The normalise+affine+ReLU *apply* is not a kernel here: it is
// folded into a per-(n,c) scale/shift pair that the consuming conv applies while staging its LDS tile
// (conv3d_fwd.hip / conv3d_wgrad.hip), so the normalised tensor never exists in HBM.
//
// Algorithmic traffic: stats = 1 read of x; backward = reads of (x, dA) twice + 1 write of dx. When the tensor (forward) or its
// gradient (backward) leaves a conv, that conv's epilogue emits the per-tile partial records instead (gn_fuse.h) and the first
// read disappears: 0 extra bytes for the forward statistics, 1 pass fewer in the backward.
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

#define GN_MAX_BLOCKS_PER_SAMPLE 256

// per-(n, block, c) partial sums (sum du, sum du*xhat), du = dA * act'(scale*x+shift), xhat = (x-mean)*rstd: the first pass of the
// backward when the dgrad conv that produced dA could not emit them from its epilogue (gn_fuse.h)
template <typename T, int VW>
__global__ void gn_bwd_partial_kernel(const T* x, int xld, const T* dA, int dald, long long V, int C, int Q, int R,
                                      int G, float slope, const float* mean_rstd, const float* scale, const float* shift,
                                      float* ws) {
  DYN_LDS(lds);  // [R][Q][2 VW]
  const int tid = threadIdx.x;
  const int q = tid % Q, r = tid / Q;
  const int n = blockIdx.y, blk = blockIdx.x, B = gridDim.x;
  const long long per = (V + B - 1) / B;
  const long long vb = (long long)blk * per;
  const long long ve = vb + per < V ? vb + per : V;
  float s0[VW], s1[VW], mean[VW], rstd[VW], sc[VW], sh[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    const int c = VW * q + e;
    const int g = c / (C / G);
    s0[e] = 0.f; s1[e] = 0.f;
    mean[e] = mean_rstd[((size_t)n * G + g) * 2];
    rstd[e] = mean_rstd[((size_t)n * G + g) * 2 + 1];
    sc[e] = scale[(size_t)n * C + c];
    sh[e] = shift[(size_t)n * C + c];
  }
  const T* xn = x + (size_t)n * V * xld + VW * q;
  const T* dn = dA + (size_t)n * V * dald + VW * q;
  for (long long v = vb + r; v < ve; v += R) {
    float xe[VW], de[VW];
    ldv<VW>(xn + (size_t)v * xld, xe);
    ldv<VW>(dn + (size_t)v * dald, de);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const float u = xe[e] * sc[e] + sh[e];
      const float du = u > 0.f ? de[e] : de[e] * slope;
      const float xh = (xe[e] - mean[e]) * rstd[e];
      s0[e] += du; s1[e] += du * xh;
    }
  }
  float* my = lds + (size_t)tid * (2 * VW);
#pragma unroll
  for (int e = 0; e < VW; ++e) { my[e] = s0[e]; my[VW + e] = s1[e]; }
  __syncthreads();
  if (r == 0) {
    float t[2 * VW];
#pragma unroll
    for (int e = 0; e < 2 * VW; ++e) t[e] = 0.f;
    for (int rr = 0; rr < R; ++rr) {
      const float* o = lds + (size_t)(rr * Q + q) * (2 * VW);
#pragma unroll
      for (int e = 0; e < 2 * VW; ++e) t[e] += o[e];
    }
    float* dst = ws + (((size_t)n * B + blk) * C + VW * q) * 2;
#pragma unroll
    for (int e = 0; e < VW; ++e) { dst[2 * e] = t[e]; dst[2 * e + 1] = t[VW + e]; }
  }
}

__device__ __forceinline__ double block_sum_double(double v, double* red) {
  const int tid = threadIdx.x;
  red[tid] = v;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// one block per (n, g): per-channel sums -> coefficients (k0, c0, k2, mean) with dx = k0*du + c0 + k2*(x - mean), and the
// per-(n,c) sums for dgamma/dbeta.
__global__ void gn_bwd_finalize_kernel(const float* ws, int B, int C, int G, long long V, const float* gamma,
                                       const float* mean_rstd, float* coef, float* nc_sums) {
  __shared__ double red[256];
  __shared__ double p1[256], p2[256];
  const int g = blockIdx.x, n = blockIdx.y;
  const int cpg = C / G;
  const int tid = threadIdx.x;
  double m1 = 0.0, m2 = 0.0;
  // per-channel sums first: the 256 threads cover (channel, slice of the block range); slices are combined in a fixed order
  for (int c0 = 0; c0 < cpg; c0 += 256) {
    const int nch = cpg - c0 < 256 ? cpg - c0 : 256;
    const int S = 256 / nch;
    const int i = tid % nch, sl = tid / nch;
    double a1 = 0.0, a2 = 0.0;
    if (sl < S) {
      const int c = g * cpg + c0 + i;
      for (int blk = sl; blk < B; blk += S) {
        const float* p = ws + (((size_t)n * B + blk) * C + c) * 2;
        a1 += (double)p[0]; a2 += (double)p[1];
      }
    }
    p1[tid] = a1; p2[tid] = a2;
    __syncthreads();
    if (tid < nch) {
      const int c = g * cpg + c0 + tid;
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < S; ++k) { s1 += p1[k * nch + tid]; s2 += p2[k * nch + tid]; }
      nc_sums[((size_t)n * C + c) * 2] = (float)s1;
      nc_sums[((size_t)n * C + c) * 2 + 1] = (float)s2;
      const double ga = gamma ? (double)gamma[c] : 1.0;
      m1 += ga * s1; m2 += ga * s2;
    }
    __syncthreads();
  }
  m1 = block_sum_double(m1, red);
  m2 = block_sum_double(m2, red);
  const double M = (double)V * cpg;
  m1 /= M; m2 /= M;
  const double mean = (double)mean_rstd[((size_t)n * G + g) * 2], rstd = (double)mean_rstd[((size_t)n * G + g) * 2 + 1];
  for (int i = threadIdx.x; i < cpg; i += blockDim.x) {
    const int c = g * cpg + i;
    const double ga = gamma ? (double)gamma[c] : 1.0;
    float* k = coef + ((size_t)n * C + c) * 4;
    k[0] = (float)(rstd * ga);
    k[1] = (float)(-rstd * m1);
    k[2] = (float)(-rstd * rstd * m2);
    k[3] = (float)mean;
  }
}

__global__ void gn_bwd_param_kernel(const float* nc_sums, int N, int C, float* dgamma, float* dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n) { s1 += (double)nc_sums[((size_t)n * C + c) * 2]; s2 += (double)nc_sums[((size_t)n * C + c) * 2 + 1]; }
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
}

template <typename T, int VW>
__global__ void gn_bwd_apply_kernel(const T* x, int xld, const T* dA, int dald, T* dx, int dxld,
                                    const T* addend, int addld, long long V, int C, int N, float slope,
                                    const float* scale, const float* shift, const float* coef) {
  const int Q = C / VW;
  const long long total = (long long)N * V * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q);
    const long long nv = idx / Q;
    const int n = (int)(nv / V);
    const int c = VW * q;
    float xe[VW], de[VW], o[VW], se[VW], he[VW];
    ldv<VW>(x + (size_t)nv * xld + c, xe);
    ldv<VW>(dA + (size_t)nv * dald + c, de);
    ldv<VW>(scale + (size_t)n * C + c, se);
    ldv<VW>(shift + (size_t)n * C + c, he);
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const float* k = coef + ((size_t)n * C + c + e) * 4;
      const float u = xe[e] * se[e] + he[e];
      const float du = u > 0.f ? de[e] : de[e] * slope;
      o[e] = k[0] * du + k[1] + k[2] * (xe[e] - k[3]);
    }
    if (addend) {
      float av[VW];
      ldv<VW>(addend + (size_t)nv * addld + c, av);
#pragma unroll
      for (int e = 0; e < VW; ++e) o[e] += av[e];
    }
    stv<VW>(dx + (size_t)nv * dxld + c, o);
  }
}

// The same pass with the (sample, channel-run) constants in registers: grid = (blocks, samples); a thread keeps ONE channel run q = tid % Q
// and walks voxels, so scale / shift / the four coefficients per channel are loaded once per thread instead of once per 16-byte run (the
// grid-stride form above reads 96-192 bytes of parameters beside every 32 bytes of payload through the vector L1). Two voxels per iteration.
// Needs 256 % Q == 0 (every channel count of the networks here); otherwise the form above. Launched for 16-bit tensors (VW = 8).
template <typename T, int VW>
__global__ __launch_bounds__(256) void gn_bwd_apply_rows_kernel(const T* x, int xld, const T* dA, int dald, T* dx, int dxld,
                                                              const T* addend, int addld, long long V, int C, float slope,
                                                              const float* scale, const float* shift, const float* coef) {
  const int Q = C / VW, VPB = 256 / Q;
  const int n = blockIdx.y, q = threadIdx.x % Q, c = VW * q;
  float se[VW], he[VW], k0[VW], k1[VW], k2[VW], k3[VW];
  ldv<VW>(scale + (size_t)n * C + c, se);
  ldv<VW>(shift + (size_t)n * C + c, he);
#pragma unroll
  for (int e = 0; e < VW; ++e) {
    const float4 k = *reinterpret_cast<const float4*>(coef + ((size_t)n * C + c + e) * 4);
    k0[e] = k.x; k1[e] = k.y; k2[e] = k.z; k3[e] = k.w;
  }
  const long long stride = (long long)gridDim.x * VPB;
  const size_t nbase = (size_t)n * V;
  auto one = [&](const float (&xe)[VW], const float (&de)[VW], float (&o)[VW]) {
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      const float u = xe[e] * se[e] + he[e];
      const float du = u > 0.f ? de[e] : de[e] * slope;
      o[e] = k0[e] * du + k1[e] + k2[e] * (xe[e] - k3[e]);
    }
  };
  long long v = (long long)blockIdx.x * VPB + threadIdx.x / Q;
  for (; v + stride < V; v += 2 * stride) {
    const size_t r0 = nbase + v, r1 = r0 + stride;
    float xa[VW], da[VW], xb[VW], db[VW], oa[VW], ob[VW];
    ldv<VW>(x + r0 * xld + c, xa); ldv<VW>(dA + r0 * dald + c, da);
    ldv<VW>(x + r1 * xld + c, xb); ldv<VW>(dA + r1 * dald + c, db);
    one(xa, da, oa); one(xb, db, ob);
    if (addend) {
      float av[VW], bv[VW];
      ldv<VW>(addend + r0 * addld + c, av); ldv<VW>(addend + r1 * addld + c, bv);
#pragma unroll
      for (int e = 0; e < VW; ++e) { oa[e] += av[e]; ob[e] += bv[e]; }
    }
    stv<VW>(dx + r0 * dxld + c, oa); stv<VW>(dx + r1 * dxld + c, ob);
  }
  if (v < V) {
    const size_t r0 = nbase + v;
    float xa[VW], da[VW], oa[VW];
    ldv<VW>(x + r0 * xld + c, xa); ldv<VW>(dA + r0 * dald + c, da);
    one(xa, da, oa);
    if (addend) {
      float av[VW];
      ldv<VW>(addend + r0 * addld + c, av);
#pragma unroll
      for (int e = 0; e < VW; ++e) oa[e] += av[e];
    }
    stv<VW>(dx + r0 * dxld + c, oa);
  }
}

// ---- statistics from partial moments (gn_fuse.h record format: (count, sum, M2) per (sample, block, channel)) ----
// Standalone producer of the records: one streaming read of x. Inside a block the sums are taken about K_c = the block's first
// voxel of that channel (a sample of the data, so they do not cancel when |mean| >> std) and converted to (count, sum, M2) once.
template <typename T, int VW>
__global__ void gn_moments_kernel(const T* x, int xld, long long V, int C, int Q, int R, float* out) {
  DYN_LDS(lds);  // [R][Q][2 VW]
  const int tid = threadIdx.x;
  const int q = tid % Q, r = tid / Q;
  const int n = blockIdx.y, blk = blockIdx.x, B = gridDim.x;
  const long long per = (V + B - 1) / B;
  const long long vb = (long long)blk * per;
  const long long ve = vb + per < V ? vb + per : V;
  const T* xn = x + (size_t)n * V * xld + VW * q;
  float s0[VW], s1[VW], kc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) { s0[e] = 0.f; s1[e] = 0.f; kc[e] = 0.f; }
  if (vb < ve) ldv<VW>(xn + (size_t)vb * xld, kc);
  for (long long v = vb + r; v < ve; v += R) {
    float xe[VW];
    ldv<VW>(xn + (size_t)v * xld, xe);
#pragma unroll
    for (int e = 0; e < VW; ++e) { const float t = xe[e] - kc[e]; s0[e] += t; s1[e] += t * t; }
  }
  float* my = lds + (size_t)tid * (2 * VW);
#pragma unroll
  for (int e = 0; e < VW; ++e) { my[e] = s0[e]; my[VW + e] = s1[e]; }
  __syncthreads();
  if (r == 0) {
    float t[2 * VW];
#pragma unroll
    for (int e = 0; e < 2 * VW; ++e) t[e] = 0.f;
    for (int rr = 0; rr < R; ++rr) {
      const float* o = lds + (size_t)(rr * Q + q) * (2 * VW);
#pragma unroll
      for (int e = 0; e < 2 * VW; ++e) t[e] += o[e];
    }
    const float cnt = (float)(ve > vb ? ve - vb : 0);
    float* dst = out + (((size_t)n * B + blk) * C + VW * q) * 3;
#pragma unroll
    for (int e = 0; e < VW; ++e) {
      float m2 = cnt > 0.f ? t[VW + e] - t[e] * t[e] / cnt : 0.f;
      dst[3 * e] = cnt; dst[3 * e + 1] = t[e] + cnt * kc[e]; dst[3 * e + 2] = m2 > 0.f ? m2 : 0.f;
    }
  }
}

// Pre-reduction of epilogue records: a 128^3 layer leaves 8192 records per (sample, channel), and the finalisation kernels run one
// workgroup per (sample, group) -- 16 workgroups walking 8192 records each were taking 25-35 us per norm (0.9 + 0.6 ms per training
// step). This kernel folds [n][B][C][K] into [n][B2][C][K] with B2 * n workgroups (same record format, each output record = the
// merge of a contiguous run of input records in a fixed order, in double), after which the finalisation reads B2 <= 64 records.
template <int K>
__global__ void gn_records_reduce_kernel(const float* in, int B, int C, int B2, float* out) {
  __shared__ double part[256][K];
  const int b2 = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int per = (B + B2 - 1) / B2;
  const int b_begin = b2 * per, b_end = b_begin + per < B ? b_begin + per : B;
  for (int c0 = 0; c0 < C; c0 += 256) {
    const int nch = C - c0 < 256 ? C - c0 : 256;
    const int S = 256 / nch, i = tid % nch, sl = tid / nch;
    const float* base = in + ((size_t)n * B * C + c0 + i) * K;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    double Kref = 0.0;
    if (K == 3 && b_begin < b_end) {
      const float* r0 = base + (size_t)b_begin * C * K;
      Kref = r0[0] > 0.f ? (double)r0[1] / (double)r0[0] : 0.0;
    }
    if (sl < S) {
      for (int blk = b_begin + sl; blk < b_end; blk += S) {
        const float* r = base + (size_t)blk * C * K;
        if (K == 3) {
          const double bc = (double)r[0];
          if (bc > 0.0) { const double bs = (double)r[1], d = bs / bc - Kref; acc[0] += bc; acc[1] += bs; acc[2] += (double)r[2] + bc * d * d; }
        } else {
#pragma unroll
          for (int k = 0; k < K; ++k) acc[k] += (double)r[k];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) part[tid][k] = acc[k];
    __syncthreads();
    if (tid < nch) {
      double t[K];
#pragma unroll
      for (int k = 0; k < K; ++k) t[k] = 0.0;
      for (int s2 = 0; s2 < S; ++s2)
#pragma unroll
        for (int k = 0; k < K; ++k) t[k] += part[s2 * nch + tid][k];
      float* dst = out + (((size_t)n * B2 + b2) * C + c0 + tid) * K;
      if (K == 3) {
        const double mc = t[0] > 0.0 ? t[1] / t[0] : 0.0;
        double m2 = t[2] - t[0] * (mc - Kref) * (mc - Kref);
        if (m2 < 0.0) m2 = 0.0;
        dst[0] = (float)t[0]; dst[1] = (float)t[1]; dst[2] = (float)m2;
      } else {
#pragma unroll
        for (int k = 0; k < K; ++k) dst[k] = (float)t[k];
      }
    }
  }
}

struct MomSrc { const float* p; int B, C, c0; };   // channels [c0, c0 + C) of the normalised tensor: records p[n][B][C][3]

// one block per (n, g). Per channel: the block records are merged in double about the reference K = the mean of the channel's first
// record (total count, total sum, M2 about K -> centred M2); the group statistics are then combined from the channels by the
// parallel-variance formula, all in a fixed order (deterministic).
__global__ void gn_moments_finalize_kernel(MomSrc sa, MomSrc sb, int C, int G, float eps, const float* gamma, const float* beta,
                                           float* mean_rstd, float* scale, float* shift) {
  __shared__ double red[256];
  __shared__ double p0[256], p1[256], p2[256];
  const int g = blockIdx.x, n = blockIdx.y;
  const int cpg = C / G;
  const int tid = threadIdx.x;
  double wsum = 0.0, cntsum = 0.0;            // this thread's channels: sum of cnt_c * mean_c, sum of cnt_c
  double mc_local[4], m2_local[4], cn_local[4];   // cpg <= 1024 -> at most 4 channel rounds of 256
  int rounds = 0;
  for (int c0 = 0; c0 < cpg; c0 += 256, ++rounds) {
    const int nch = cpg - c0 < 256 ? cpg - c0 : 256;
    const int S = 256 / nch;                       // slices of the record range per channel
    const int i = tid % nch, sl = tid / nch;
    double cn = 0.0, sm = 0.0, qq = 0.0;
    if (sl < S) {
      const int c = g * cpg + c0 + i;
      const MomSrc src = (sb.p != nullptr && c >= sb.c0) ? sb : sa;
      const float* base = src.p + ((size_t)n * src.B * src.C + (c - src.c0)) * 3;
      const size_t stride = (size_t)src.C * 3;
      const double K = base[0] > 0.f ? (double)base[1] / (double)base[0] : 0.0;
      for (int blk = sl; blk < src.B; blk += S) {
        const float* r = base + (size_t)blk * stride;
        const double bc = (double)r[0];
        if (bc > 0.0) {
          const double bs = (double)r[1], d = bs / bc - K;
          cn += bc; sm += bs; qq += (double)r[2] + bc * d * d;
        }
      }
    }
    p0[tid] = cn; p1[tid] = sm; p2[tid] = qq;
    __syncthreads();
    double mc = 0.0, m2 = 0.0, cc = 0.0;
    if (tid < nch) {
      double a0 = 0.0, a1 = 0.0, a2 = 0.0;
      for (int k = 0; k < S; ++k) { a0 += p0[k * nch + tid]; a1 += p1[k * nch + tid]; a2 += p2[k * nch + tid]; }
      const int c = g * cpg + c0 + tid;
      const MomSrc src = (sb.p != nullptr && c >= sb.c0) ? sb : sa;
      const float* base = src.p + ((size_t)n * src.B * src.C + (c - src.c0)) * 3;
      const double K = base[0] > 0.f ? (double)base[1] / (double)base[0] : 0.0;
      cc = a0;
      mc = a0 > 0.0 ? a1 / a0 : 0.0;
      m2 = a2 - a0 * (mc - K) * (mc - K);
      if (m2 < 0.0) m2 = 0.0;
      wsum += a0 * mc; cntsum += a0;
    }
    mc_local[rounds] = mc; m2_local[rounds] = m2; cn_local[rounds] = cc;
    __syncthreads();
  }
  const double M = block_sum_double(cntsum, red);
  const double mean = M > 0.0 ? block_sum_double(wsum, red) / M : 0.0;
  double m2sum = 0.0;
  rounds = 0;
  for (int c0 = 0; c0 < cpg; c0 += 256, ++rounds) {
    const int nch = cpg - c0 < 256 ? cpg - c0 : 256;
    if (tid < nch) { const double d = mc_local[rounds] - mean; m2sum += m2_local[rounds] + cn_local[rounds] * d * d; }
  }
  double var = M > 0.0 ? block_sum_double(m2sum, red) / M : 0.0;
  if (var < 0.0) var = 0.0;
  const double rstd = 1.0 / sqrt(var + (double)eps);
  if (tid == 0) {
    mean_rstd[((size_t)n * G + g) * 2] = (float)mean;
    mean_rstd[((size_t)n * G + g) * 2 + 1] = (float)rstd;
  }
  for (int i = tid; i < cpg; i += blockDim.x) {
    const int c = g * cpg + i;
    const double ga = gamma ? (double)gamma[c] : 1.0, be = beta ? (double)beta[c] : 0.0;
    scale[(size_t)n * C + c] = (float)(ga * rstd);
    shift[(size_t)n * C + c] = (float)(be - mean * ga * rstd);
  }
}

static int gn_blocks_per_sample(long long V) {
  long long b = V / 16;   // >= 16 voxels per block: the deep 8^3 / 16^3 levels still get hundreds of blocks (V/2048 left them latency-bound)   // >= 16 voxels per block; small deep levels still get hundreds of blocks (they were latency-bound at V/2048)
  if (b < 1) b = 1;
  if (b > GN_MAX_BLOCKS_PER_SAMPLE) b = GN_MAX_BLOCKS_PER_SAMPLE;
  return (int)b;
}

// workspace layout (floats): forward: moment records [N][B][C][3]; backward: partials [N][B][C][2] | coef [N][C][4] | nc_sums [N][C][2]
// (sized for the larger of the two)
extern "C" size_t mi355_gn_workspace(const mi355_act* x) {
  if (!x) return 0;
  const long long V = (long long)x->d * x->h * x->w;
  const int B = gn_blocks_per_sample(V);
  return ((size_t)x->n * B * x->c * 3 + (size_t)x->n * x->c * 6) * sizeof(float);
}

static int gn_check(const mi355_act* x, int groups) {
  if (!x || !x->p || x->c % 4 || x->ld % 4 || x->ld < x->c || groups <= 0 || x->c % groups || !act_dtype_ok(x)) return MI355_EINVAL;
  if (x->c / 4 > 256 || x->c / groups > 1024) return MI355_EUNSUPPORTED;
  if ((uintptr_t)x->p & act_align_mask(x->dtype)) return MI355_EINVAL;
  return 0;
}

extern "C" int32_t mi355_gn_moments_blocks(const mi355_act* x) {
  if (!x) return 0;
  return gn_blocks_per_sample((long long)x->d * x->h * x->w);
}

extern "C" int mi355_gn_moments(const mi355_act* x, float* out, void* stream) {
  int rc = gn_check(x, 1);
  if (rc) return rc;
  if (!out) return MI355_EINVAL;
  const long long V = (long long)x->d * x->h * x->w;
  const int B = gn_blocks_per_sample(V), C = x->c;
  if (act_vw8(x)) {
    const int Q = C / 8, R = 256 / Q > 0 ? 256 / Q : 1;
    ACT_TYPED_LP16(x->dtype, T, LAUNCH((gn_moments_kernel<T, 8>), dim3(B, x->n), dim3(Q * R), (size_t)Q * R * 16 * sizeof(float), stream, (const T*)x->p, x->ld, V, C, Q, R, out));
  } else {
    const int Q = C / 4, R = 256 / Q > 0 ? 256 / Q : 1;
    ACT_TYPED(x->dtype, T, LAUNCH((gn_moments_kernel<T, 4>), dim3(B, x->n), dim3(Q * R), (size_t)Q * R * 8 * sizeof(float), stream, (const T*)x->p, x->ld, V, C, Q, R, out));
  }
  return LAUNCH_CHECK();
}

extern "C" int mi355_gn_records_reduce(const float* in, int32_t n, int32_t blocks, int32_t c, int32_t k, float* out, int32_t out_blocks,
                                       void* stream) {
  if (!in || !out || n <= 0 || blocks <= 0 || c <= 0 || out_blocks <= 0 || out_blocks > blocks || (k != 2 && k != 3)) return MI355_EINVAL;
  if (k == 3) LAUNCH((gn_records_reduce_kernel<3>), dim3(out_blocks, n), dim3(256), 0, stream, in, blocks, c, out_blocks, out);
  else LAUNCH((gn_records_reduce_kernel<2>), dim3(out_blocks, n), dim3(256), 0, stream, in, blocks, c, out_blocks, out);
  return LAUNCH_CHECK();
}

extern "C" int mi355_gn_finalize(const float* part_a, int32_t blocks_a, int32_t c_a, const float* part_b, int32_t blocks_b, int32_t c_b,
                                 int32_t n, int32_t groups, float eps, const float* gamma, const float* beta,
                                 float* mean_rstd, float* scale, float* shift, void* stream) {
  if (!part_a || blocks_a <= 0 || c_a <= 0 || n <= 0 || groups <= 0 || !mean_rstd || !scale || !shift) return MI355_EINVAL;
  if (part_b && (blocks_b <= 0 || c_b <= 0)) return MI355_EINVAL;
  const int C = c_a + (part_b ? c_b : 0);
  if (C % groups) return MI355_EINVAL;
  if (C / groups > 1024) return MI355_EUNSUPPORTED;
  MomSrc sa = {part_a, blocks_a, c_a, 0}, sb = {part_b, part_b ? blocks_b : 0, part_b ? c_b : 0, c_a};
  LAUNCH(gn_moments_finalize_kernel, dim3(groups, n), dim3(256), 0, stream, sa, sb, C, groups, eps, gamma, beta, mean_rstd, scale, shift);
  return LAUNCH_CHECK();
}

// standalone statistics = the standalone record producer + the same finalisation the fused epilogues feed
extern "C" int mi355_gn_stats(const mi355_act* x, int32_t groups, float eps, const float* gamma, const float* beta,
                              float* mean_rstd, float* scale, float* shift, void* ws, size_t ws_bytes, void* stream) {
  int rc = gn_check(x, groups);
  if (rc) return rc;
  if (!mean_rstd || !scale || !shift || !ws) return MI355_EINVAL;
  if (ws_bytes < mi355_gn_workspace(x)) return MI355_EWORKSPACE;
  rc = mi355_gn_moments(x, (float*)ws, stream);
  if (rc) return rc;
  return mi355_gn_finalize((const float*)ws, mi355_gn_moments_blocks(x), x->c, nullptr, 0, 0, x->n, groups, eps, gamma, beta,
                           mean_rstd, scale, shift, stream);
}

static int gn_act_bwd_impl(const mi355_act* x, const mi355_act* dA, const mi355_act* dx, const void* addend, int32_t addend_ld,
                           int32_t groups, float act_slope, const float* gamma, const float* mean_rstd,
                           const float* scale, const float* shift, float* dgamma, float* dbeta,
                           const float* partials, int32_t blocks, void* ws, size_t ws_bytes, void* stream) {
  int rc = gn_check(x, groups);
  if (rc) return rc;
  if (!dA || !dx || !dA->p || !dx->p || !mean_rstd || !scale || !shift || !ws) return MI355_EINVAL;
  const uintptr_t amask = act_align_mask(x->dtype);
  if (dA->c != x->c || dx->c != x->c || dA->ld % 4 || dx->ld % 4 || ((uintptr_t)dA->p & amask) || ((uintptr_t)dx->p & amask)) return MI355_EINVAL;
  if (dA->dtype != x->dtype || dx->dtype != x->dtype) return MI355_EUNSUPPORTED;      // a tensor's gradient has the tensor's storage type
  if (addend && (addend_ld % 4 || ((uintptr_t)addend & amask))) return MI355_EINVAL;
  if (ws_bytes < mi355_gn_workspace(x)) return MI355_EWORKSPACE;
  if (partials && blocks <= 0) return MI355_EINVAL;
  const long long V = (long long)x->d * x->h * x->w;
  const int C = x->c, N = x->n, Q = C / 4, R = 256 / Q > 0 ? 256 / Q : 1;
  int B = gn_blocks_per_sample(V);
  float* part = (float*)ws;
  float* coef = part + (size_t)N * B * C * 2;
  float* ncs = coef + (size_t)N * C * 4;
  if (partials) {
    B = blocks;                                   // first pass already done by the dgrad conv's epilogue (gn_fuse.h)
  } else {
    if (act_vw8(x) && act_vw8(dA)) {
      const int Q8 = C / 8, R8 = 256 / Q8 > 0 ? 256 / Q8 : 1;
      ACT_TYPED_LP16(x->dtype, T, LAUNCH((gn_bwd_partial_kernel<T, 8>), dim3(B, N), dim3(Q8 * R8), (size_t)Q8 * R8 * 16 * sizeof(float), stream,
             (const T*)x->p, x->ld, (const T*)dA->p, dA->ld, V, C, Q8, R8, groups, act_slope, mean_rstd, scale, shift, part));
    } else {
      ACT_TYPED(x->dtype, T, LAUNCH((gn_bwd_partial_kernel<T, 4>), dim3(B, N), dim3(Q * R), (size_t)Q * R * 8 * sizeof(float), stream,
                                    (const T*)x->p, x->ld, (const T*)dA->p, dA->ld, V, C, Q, R, groups, act_slope, mean_rstd, scale, shift, part));
    }
    rc = LAUNCH_CHECK(); if (rc) return rc;
    partials = part;
  }
  LAUNCH(gn_bwd_finalize_kernel, dim3(groups, N), dim3(256), 0, stream, partials, B, C, groups, V, gamma, mean_rstd, coef, ncs);
  rc = LAUNCH_CHECK(); if (rc) return rc;
  if (dgamma || dbeta) {
    LAUNCH(gn_bwd_param_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, (const float*)ncs, N, C, dgamma, dbeta);
    rc = LAUNCH_CHECK(); if (rc) return rc;
  }
  const long long total = (long long)N * V * Q;
  long long grid = (total + 255) / 256; if (grid > 8192) grid = 8192;
  const bool vw8 = act_vw8(x) && act_vw8(dA) && act_vw8(dx) && (!addend || (addend_ld % 8 == 0 && !((uintptr_t)addend & 15)));
  // the rows form (constants in registers, two voxels per iteration) where a 256-thread block holds whole voxels of Q channel runs;
  // MI355_GN_APPLY_ROWS=0 (read once): the grid-stride form everywhere -- the A/B switch
  static const bool rows_off = [] { const char* v = getenv("MI355_GN_APPLY_ROWS"); return v && v[0] == '0'; }();
  const int Qa = C / 8;
  // 16-byte runs of 16-bit tensors only: there the constants are 192 bytes beside 32 bytes of payload and the rows form runs 32 ch @128^3 x 4 in
  // 0.133 instead of 0.248 ms; on fp32 tensors both forms sit at the memory system's 5 TB/s (tools/bench_gn_bwd.py) and the step is a wash
  if (!rows_off && vw8 && Qa <= 256 && 256 % Qa == 0 && N <= 65535) {
    const int VPB = 256 / Qa;
    long long gx = (V + (long long)VPB * 8 - 1) / ((long long)VPB * 8);      // >= 8 voxels per thread where the tensor allows
    const long long cap = (8192 + N - 1) / N;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    ACT_TYPED_LP16(x->dtype, T, LAUNCH((gn_bwd_apply_rows_kernel<T, 8>), dim3((unsigned)gx, N), dim3(256), 0, stream, (const T*)x->p, x->ld, (const T*)dA->p,
           dA->ld, (T*)dx->p, dx->ld, (const T*)addend, addend_ld, V, C, act_slope, scale, shift, (const float*)coef));
    return LAUNCH_CHECK();
  }
  if (vw8) {
    const long long total8 = (long long)N * V * (C / 8);
    long long grid8 = (total8 + 255) / 256; if (grid8 > 8192) grid8 = 8192;
    ACT_TYPED_LP16(x->dtype, T, LAUNCH((gn_bwd_apply_kernel<T, 8>), dim3((unsigned)grid8), dim3(256), 0, stream, (const T*)x->p, x->ld, (const T*)dA->p, dA->ld,
           (T*)dx->p, dx->ld, (const T*)addend, addend_ld, V, C, N, act_slope, scale, shift, (const float*)coef));
  } else {
    ACT_TYPED(x->dtype, T, LAUNCH((gn_bwd_apply_kernel<T, 4>), dim3((unsigned)grid), dim3(256), 0, stream, (const T*)x->p, x->ld, (const T*)dA->p, dA->ld,
                                  (T*)dx->p, dx->ld, (const T*)addend, addend_ld, V, C, N, act_slope, scale, shift, (const float*)coef));
  }
  return LAUNCH_CHECK();
}

// The parameter gradients alone, from partial records (gn_fuse.h format) a producer left: no data-gradient pass. For the network's FIRST
// norm (its input needs no gradient): conv3d_c4_bwd.hip leaves the records, nothing reads or writes a tensor here.
extern "C" int mi355_gn_bwd_params(const mi355_act* x, int32_t groups, const float* gamma, const float* mean_rstd, float* dgamma, float* dbeta,
                                   const float* partials, int32_t blocks, void* ws, size_t ws_bytes, void* stream) {
  int rc = gn_check(x, groups);
  if (rc) return rc;
  if (!partials || blocks <= 0 || !mean_rstd || !ws || (!dgamma && !dbeta)) return MI355_EINVAL;
  if (ws_bytes < mi355_gn_workspace(x)) return MI355_EWORKSPACE;
  const long long V = (long long)x->d * x->h * x->w;
  const int C = x->c, N = x->n;
  const int B = gn_blocks_per_sample(V);
  float* part = (float*)ws;
  float* coef = part + (size_t)N * B * C * 2;
  float* ncs = coef + (size_t)N * C * 4;
  LAUNCH(gn_bwd_finalize_kernel, dim3(groups, N), dim3(256), 0, stream, partials, (int)blocks, C, groups, V, gamma, mean_rstd, coef, ncs);
  rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(gn_bwd_param_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, stream, (const float*)ncs, N, C, dgamma, dbeta);
  return LAUNCH_CHECK();
}

extern "C" int mi355_gn_act_bwd(const mi355_act* x, const mi355_act* dA, const mi355_act* dx, const void* addend, int32_t addend_ld,
                                int32_t groups, float act_slope, const float* gamma, const float* mean_rstd,
                                const float* scale, const float* shift, float* dgamma, float* dbeta,
                                void* ws, size_t ws_bytes, void* stream) {
  return gn_act_bwd_impl(x, dA, dx, addend, addend_ld, groups, act_slope, gamma, mean_rstd, scale, shift, dgamma, dbeta, nullptr, 0,
                         ws, ws_bytes, stream);
}

extern "C" int mi355_gn_act_bwd_fused(const mi355_act* x, const mi355_act* dA, const mi355_act* dx, const void* addend, int32_t addend_ld,
                                      int32_t groups, float act_slope, const float* gamma, const float* mean_rstd,
                                      const float* scale, const float* shift, float* dgamma, float* dbeta,
                                      const float* partials, int32_t blocks, void* ws, size_t ws_bytes, void* stream) {
  if (!partials) return MI355_EINVAL;
  return gn_act_bwd_impl(x, dA, dx, addend, addend_ld, groups, act_slope, gamma, mean_rstd, scale, shift, dgamma, dbeta, partials, blocks,
                         ws, ws_bytes, stream);
}
