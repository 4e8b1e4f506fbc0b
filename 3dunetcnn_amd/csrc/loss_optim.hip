// Fused sigmoid-Dice loss (forward + gradient) and fused Adam -- HBM-bound streaming kernels.
//
// Dice: monai.losses.DiceLoss(include_background=True, sigmoid=True) as selected by
// examples/brats2020/brats2020_config.json:112-116 through unet3d/scripts/script_utils.py:61-77 and evaluated at
// unet3d/train/training_utils.py:111. MONAI is un-vendored; formula per SURVEY.md Appendix C:
//   p = sigmoid(z); per (n,c): I = sum p*y, D = sum p + sum y (sum p^2 + sum y^2 if squared_pred);
//   f = 1 - (2I + smooth_nr)/(D + smooth_dr); loss = mean f   (batch=True sums I, D over n first).
// Adam: torch.optim.Adam defaults (script_utils.py:80-81), single flat parameter buffer.
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"

#define DICE_MAX_BLOCKS 128

__device__ __forceinline__ float sigmoidf_(float z) { return 1.f / (1.f + expf(-z)); }

__global__ void dice_partial_kernel(const float* logits, const void* target, int target_u8, long long V, int sigmoid, int squared,
                                    float* ws) {
  __shared__ float red[3][256];
  const int nc = blockIdx.y, blk = blockIdx.x, B = gridDim.x;
  const long long per = (V + B - 1) / B;
  const long long vb = (long long)blk * per, ve = vb + per < V ? vb + per : V;
  const float* z = logits + (size_t)nc * V;
  const unsigned char* t8 = (const unsigned char*)target + (size_t)nc * V;
  const float* tf = (const float*)target + (size_t)nc * V;
  float sI = 0.f, sP = 0.f, sY = 0.f;
  for (long long v = vb + threadIdx.x; v < ve; v += blockDim.x) {
    const float p = sigmoid ? sigmoidf_(z[v]) : z[v];
    const float y = target_u8 ? (float)t8[v] : tf[v];
    sI += p * y;
    sP += squared ? p * p : p;
    sY += squared ? y * y : y;
  }
  red[0][threadIdx.x] = sI; red[1][threadIdx.x] = sP; red[2][threadIdx.x] = sY;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      red[0][threadIdx.x] += red[0][threadIdx.x + s];
      red[1][threadIdx.x] += red[1][threadIdx.x + s];
      red[2][threadIdx.x] += red[2][threadIdx.x + s];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* d = ws + ((size_t)nc * B + blk) * 3;
    d[0] = red[0][0]; d[1] = red[1][0]; d[2] = red[2][0];
  }
}

// single block: sums -> loss and per-(n,c) gradient coefficients (a, b): dL/dp = -a*y + b*(1 or 2p).
// variant 0: DiceLoss, mean over (n, c) [or c with batch] of 1 - (2I + snr)/(P + G + sdr).
// variant 1: monai GeneralizedDiceLoss (w_type "square"): per sample n [or once with batch] w_c = 1/G_c^2, an infinite weight (empty
//            class) is replaced by the largest finite weight of that sample; f = 1 - (2 sum_c w_c I_c + snr)/(sum_c w_c (G_c + P_c) + sdr);
//            loss = mean over samples.
// c0 = 1 when include_background=False: channel 0 is left out of the loss (zero gradient).
// jaccard (variant 0): denominator 2 (D - I) instead of D. class_w: one factor per counted class (or NULL). reduction 0 mean / 1 sum /
// 2 none: with "none" loss[] receives one value per term (n-major, counted classes only; one row with batch) and the coefficients are those
// of sum (the caller applies the upstream gradient of every term).
__global__ void dice_finalize_kernel(const float* ws, int B, int N, int C, int batch, float snr, float sdr, float grad_scale,
                                     int variant, int c0, float* stats, float* coef, float* loss,
                                     int jaccard, const float* class_w, int reduction) {
  __shared__ double sums[3 * 1024];
  __shared__ double fsum[256];
  const int NC = N * C;
  for (int i = threadIdx.x; i < NC; i += blockDim.x) {
    double a = 0.0, b = 0.0, c = 0.0;
    for (int k = 0; k < B; ++k) { const float* p = ws + ((size_t)i * B + k) * 3; a += (double)p[0]; b += (double)p[1]; c += (double)p[2]; }
    sums[3 * i] = a; sums[3 * i + 1] = b; sums[3 * i + 2] = c;
    stats[3 * i] = (float)a; stats[3 * i + 1] = (float)b; stats[3 * i + 2] = (float)c;
  }
  __syncthreads();
  double f = 0.0;
  const int Ce = C - c0;                                  // channels that count
  if (variant == 0) {
    const int K = reduction != 0 ? 1 : (batch ? Ce : N * Ce);      // number of terms in the mean
    for (int i = threadIdx.x; i < NC; i += blockDim.x) {
      const int c = i % C;
      if (c < c0) { coef[2 * i] = 0.f; coef[2 * i + 1] = 0.f; continue; }
      double I, D;
      if (batch) {
        I = 0.0; D = 0.0;
        for (int n = 0; n < N; ++n) { I += sums[3 * (n * C + c)]; D += sums[3 * (n * C + c) + 1] + sums[3 * (n * C + c) + 2]; }
      } else { I = sums[3 * i]; D = sums[3 * i + 1] + sums[3 * i + 2]; }
      const double w = class_w ? (double)class_w[c - c0] : 1.0;
      const double num = 2.0 * I + (double)snr;
      const double den = (jaccard ? 2.0 * (D - I) : D) + (double)sdr;
      const double s = (double)grad_scale * w / K;
      // f = 1 - num / den:  df/dI = -(2 den - num dden/dI) / den^2,  df/dD = num dden/dD / den^2;  dL/dp = -a y + b (1 | 2p)
      coef[2 * i] = (float)(s * (jaccard ? 2.0 * (den + num) : 2.0 * den) / (den * den));
      coef[2 * i + 1] = (float)(s * (jaccard ? 2.0 : 1.0) * num / (den * den));
      const double term = w * (1.0 - num / den);
      if (!batch || i < C) {
        if (reduction == 2) loss[batch ? c - c0 : (i / C) * Ce + c - c0] = (float)term;
        else f += term / K;
      }
    }
  } else {
    const int K = batch ? 1 : N;
    for (int i = threadIdx.x; i < NC; i += blockDim.x) {
      const int c = i % C, n = i / C;
      if (c < c0) { coef[2 * i] = 0.f; coef[2 * i + 1] = 0.f; continue; }
      // class sums of this sample (or of the whole batch) and the weights
      double wmax = 0.0, numer = (double)snr, denom = (double)sdr, wme = 0.0;
      for (int pass = 0; pass < 2; ++pass) {
        for (int cc = c0; cc < C; ++cc) {
          double I = 0.0, P = 0.0, G = 0.0;
          const int n0 = batch ? 0 : n, n1 = batch ? N : n + 1;
          for (int nn = n0; nn < n1; ++nn) { I += sums[3 * (nn * C + cc)]; P += sums[3 * (nn * C + cc) + 1]; G += sums[3 * (nn * C + cc) + 2]; }
          const bool empty = G == 0.0;
          if (pass == 0) { if (!empty) { const double w = 1.0 / (G * G); wmax = w > wmax ? w : wmax; } continue; }
          const double w = empty ? wmax : 1.0 / (G * G);
          numer += 2.0 * w * I; denom += w * (G + P);
          if (cc == c) wme = w;
        }
      }
      coef[2 * i] = (float)((double)grad_scale * 2.0 * wme / (K * denom));
      coef[2 * i + 1] = (float)((double)grad_scale * numer * wme / (K * denom * denom));
      if (c == c0 && (!batch || n == 0)) f += (1.0 - numer / denom) / K;
    }
  }
  fsum[threadIdx.x] = f;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) { if ((int)threadIdx.x < s) fsum[threadIdx.x] += fsum[threadIdx.x + s]; __syncthreads(); }
  if (threadIdx.x == 0 && !(variant == 0 && reduction == 2)) loss[0] = (float)fsum[0];
}

__global__ void dice_grad_kernel(const float* logits, const void* target, int target_u8, long long V, int NC, int sigmoid, int squared,
                                 const float* coef, float* dlogits) {
  const long long total = (long long)NC * V;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int nc = (int)(idx / V);
    const float zz = logits[idx];
    const float p = sigmoid ? sigmoidf_(zz) : zz;
    const float y = target_u8 ? (float)((const unsigned char*)target)[idx] : ((const float*)target)[idx];
    float g = -coef[2 * nc] * y + coef[2 * nc + 1] * (squared ? 2.f * p : 1.f);
    if (sigmoid) g *= p * (1.f - p);
    dlogits[idx] = g;
  }
}


// ---- extended Dice (monai DiceLoss options beyond the shipped configuration: softmax, to_onehot_y, jaccard, weight, reduction) ----
#define DICE_MAX_C 16
__device__ __forceinline__ float dice_y(const void* target, int kind, long long V, int C, int n, int c, long long v) {
  if (kind == MI355_DICE_TARGET_LABELS) return ((const int*)target)[(size_t)n * V + v] == c ? 1.f : 0.f;
  const size_t i = ((size_t)n * C + c) * V + v;
  return kind == MI355_DICE_TARGET_U8 ? (float)((const unsigned char*)target)[i] : ((const float*)target)[i];
}
// p[c] = act(z[c]) for the C channels of one voxel
__device__ __forceinline__ void dice_probs(const float* z, long long V, int C, int act, long long v, float (&p)[DICE_MAX_C]) {
  if (act == MI355_DICE_ACT_SOFTMAX) {
    float m = -3.4e38f;
    for (int c = 0; c < C; ++c) { p[c] = z[(size_t)c * V + v]; m = p[c] > m ? p[c] : m; }
    float sum = 0.f;
    for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - m); sum += p[c]; }
    const float inv = 1.f / sum;
    for (int c = 0; c < C; ++c) p[c] *= inv;
  } else {
    for (int c = 0; c < C; ++c) { const float zz = z[(size_t)c * V + v]; p[c] = act == MI355_DICE_ACT_SIGMOID ? sigmoidf_(zz) : zz; }
  }
}

// grid (B, N): one block = one voxel range of one sample, all channels; partial layout = dice_partial_kernel's [n*C + c][B][3]
__global__ void dice_ex_partial_kernel(const float* logits, const void* target, int kind, long long V, int C, int act, int squared, float* ws) {
  __shared__ float red[256];
  const int n = blockIdx.y, blk = blockIdx.x, B = gridDim.x;
  const long long per = (V + B - 1) / B;
  const long long vb = (long long)blk * per, ve = vb + per < V ? vb + per : V;
  const float* z = logits + (size_t)n * C * V;
  float sI[DICE_MAX_C], sP[DICE_MAX_C], sY[DICE_MAX_C];
  for (int c = 0; c < C; ++c) { sI[c] = 0.f; sP[c] = 0.f; sY[c] = 0.f; }
  for (long long v = vb + threadIdx.x; v < ve; v += blockDim.x) {
    float p[DICE_MAX_C];
    dice_probs(z, V, C, act, v, p);
    for (int c = 0; c < C; ++c) {
      const float y = dice_y(target, kind, V, C, n, c, v);
      sI[c] += p[c] * y;
      sP[c] += squared ? p[c] * p[c] : p[c];
      sY[c] += squared ? y * y : y;
    }
  }
  for (int c = 0; c < C; ++c)
    for (int k = 0; k < 3; ++k) {
      __syncthreads();
      red[threadIdx.x] = k == 0 ? sI[c] : k == 1 ? sP[c] : sY[c];
      __syncthreads();
      for (int s = blockDim.x / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
      }
      if (threadIdx.x == 0) ws[(((size_t)n * C + c) * B + blk) * 3 + k] = red[0];
    }
}

// dlogits = sum over terms of upstream(term) * d(term)/dlogits. upstream: NULL (1 for every term), one value (mean / sum) or one per
// term ("none": n-major over the counted classes; one row with batch).
__global__ void dice_ex_grad_kernel(const float* logits, const void* target, int kind, long long V, int N, int C, int act, int squared,
                                    const float* coef, const float* upstream, int n_up, int batch, int c0, float* dlogits) {
  const long long total = (long long)N * V;
  const int Ce = C - c0;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(idx / V);
    const long long v = idx - (long long)n * V;
    const float* z = logits + (size_t)n * C * V;
    float p[DICE_MAX_C], g[DICE_MAX_C];
    dice_probs(z, V, C, act, v, p);
    float dot = 0.f;
    for (int c = 0; c < C; ++c) {
      float up = 1.f;
      if (upstream) up = n_up == 1 ? upstream[0] : (c >= c0 ? upstream[batch ? c - c0 : n * Ce + c - c0] : 0.f);
      const float y = dice_y(target, kind, V, C, n, c, v);
      g[c] = up * (-coef[2 * (n * C + c)] * y + coef[2 * (n * C + c) + 1] * (squared ? 2.f * p[c] : 1.f));
      dot += g[c] * p[c];
    }
    for (int c = 0; c < C; ++c) {
      float d = g[c];
      if (act == MI355_DICE_ACT_SIGMOID) d *= p[c] * (1.f - p[c]);
      else if (act == MI355_DICE_ACT_SOFTMAX) d = p[c] * (g[c] - dot);
      dlogits[((size_t)n * C + c) * V + v] = d;
    }
  }
}

static int dice_blocks(long long V) { long long b = V / 4096; if (b < 1) b = 1; if (b > DICE_MAX_BLOCKS) b = DICE_MAX_BLOCKS; return (int)b; }


// ---- cross-entropy (softmax-CE with probability targets / BCE-with-logits), value + gradient in one pass ----
#define CE_BLOCKS 1024
#define CE_MAX_C 16
template <typename TT>
__global__ void ce_kernel(const float* z, const TT* y, int N, int C, long long V, int mode, float gscale, float* dz, int accumulate,
                          float* part) {
  __shared__ float red[256];
  float local = 0.f;
  const long long NV = (long long)N * V;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < NV; i += (long long)gridDim.x * blockDim.x) {
    const long long n = i / V, v = i - n * V;
    const size_t base = (size_t)n * C * V + v;
    float zc[CE_MAX_C], yc[CE_MAX_C];
    for (int c = 0; c < C; ++c) { zc[c] = z[base + (size_t)c * V]; yc[c] = (float)y[base + (size_t)c * V]; }
    if (mode == MI355_CE_SOFTMAX) {
      float mx = zc[0];
      for (int c = 1; c < C; ++c) mx = zc[c] > mx ? zc[c] : mx;
      float se = 0.f, sy = 0.f;
      for (int c = 0; c < C; ++c) { se += expf(zc[c] - mx); sy += yc[c]; }
      const float lse = mx + logf(se);
      for (int c = 0; c < C; ++c) {
        local += yc[c] * (lse - zc[c]);
        if (dz) {
          const float g = (expf(zc[c] - lse) * sy - yc[c]) * gscale;
          dz[base + (size_t)c * V] = accumulate ? dz[base + (size_t)c * V] + g : g;
        }
      }
    } else {
      for (int c = 0; c < C; ++c) {
        const float a = fabsf(zc[c]);
        local += (zc[c] > 0.f ? zc[c] : 0.f) - zc[c] * yc[c] + log1pf(expf(-a));
        if (dz) {
          const float p = 1.f / (1.f + expf(-zc[c]));
          const float g = (p - yc[c]) * gscale;
          dz[base + (size_t)c * V] = accumulate ? dz[base + (size_t)c * V] + g : g;
        }
      }
    }
  }
  red[threadIdx.x] = local;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ void ce_finalize_kernel(const float* part, int B, double inv_count, float weight, float* loss, int accumulate) {
  __shared__ double red[256];
  double s = 0.0;
  for (int b = threadIdx.x; b < B; b += blockDim.x) s += (double)part[b];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float v = (float)(red[0] * inv_count) * weight;
    loss[0] = accumulate ? loss[0] + v : v;
  }
}

extern "C" size_t mi355_ce_workspace(int64_t voxels) { (void)voxels; return CE_BLOCKS * sizeof(float); }

extern "C" int mi355_ce_fwd_bwd(const float* logits, const void* target, int32_t target_is_u8, int32_t n, int32_t c, int64_t voxels,
                                int32_t mode, float weight, float* loss, int32_t accumulate_loss, float* dlogits, int32_t accumulate_grad,
                                float grad_scale, void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !target || !loss || !ws || n < 1 || c < 1 || c > CE_MAX_C || voxels <= 0) return MI355_EINVAL;
  if (mode != MI355_CE_SOFTMAX && mode != MI355_CE_BCE) return MI355_EINVAL;
  if (ws_bytes < mi355_ce_workspace(voxels)) return MI355_EWORKSPACE;
  const long long NV = (long long)n * voxels;
  const double count = mode == MI355_CE_SOFTMAX ? (double)NV : (double)NV * c;
  long long g = (NV + 255) / 256; if (g > CE_BLOCKS) g = CE_BLOCKS;
  const float gs = (float)((double)weight * (double)grad_scale / count);
  if (target_is_u8)
    LAUNCH((ce_kernel<unsigned char>), dim3((unsigned)g), dim3(256), 0, stream, logits, (const unsigned char*)target, n, c, (long long)voxels, mode,
           gs, dlogits, accumulate_grad, (float*)ws);
  else
    LAUNCH((ce_kernel<float>), dim3((unsigned)g), dim3(256), 0, stream, logits, (const float*)target, n, c, (long long)voxels, mode, gs, dlogits,
           accumulate_grad, (float*)ws);
  int rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(ce_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)ws, (int)g, 1.0 / count, weight, loss, accumulate_loss);
  return LAUNCH_CHECK();
}

// ws (floats): partials [NC][B][3] | stats [NC][3] | coef [NC][2]
extern "C" size_t mi355_dice_workspace(int32_t n, int32_t c, int64_t voxels) {
  const size_t NC = (size_t)n * c;
  return (NC * dice_blocks(voxels) * 3 + NC * 5) * sizeof(float);
}

extern "C" int mi355_dice_fwd_bwd(const float* logits, const void* target, int32_t target_is_u8, int32_t n, int32_t c, int64_t voxels,
                                  int32_t sigmoid, int32_t batch, int32_t squared_pred, int32_t variant, int32_t include_background,
                                  float smooth_nr, float smooth_dr, float* loss, float* dlogits, float grad_scale, void* ws,
                                  size_t ws_bytes, void* stream) {
  if (!logits || !target || !loss || !ws || n <= 0 || c <= 0 || voxels <= 0) return MI355_EINVAL;
  if (variant < MI355_DICE_PLAIN || variant > MI355_DICE_GENERALIZED || (variant == MI355_DICE_GENERALIZED && squared_pred)) return MI355_EINVAL;
  if (!include_background && c < 2) return MI355_EINVAL;
  if ((size_t)n * c > 1024) return MI355_EUNSUPPORTED;
  if (ws_bytes < mi355_dice_workspace(n, c, voxels)) return MI355_EWORKSPACE;
  const int NC = n * c, B = dice_blocks(voxels);
  float* part = (float*)ws; float* stats = part + (size_t)NC * B * 3; float* coef = stats + (size_t)NC * 3;
  LAUNCH(dice_partial_kernel, dim3(B, NC), dim3(256), 0, stream, logits, target, target_is_u8, (long long)voxels, sigmoid, squared_pred, part);
  int rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(dice_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)part, B, n, c, batch, smooth_nr, smooth_dr, grad_scale, variant,
         include_background ? 0 : 1, stats, coef, loss, 0, (const float*)nullptr, 0);
  rc = LAUNCH_CHECK(); if (rc) return rc;
  if (dlogits) {
    const long long total = (long long)NC * voxels;
    long long grid = (total + 255) / 256; if (grid > 16384) grid = 16384;
    LAUNCH(dice_grad_kernel, dim3((unsigned)grid), dim3(256), 0, stream, logits, target, target_is_u8, (long long)voxels, NC, sigmoid, squared_pred,
           (const float*)coef, dlogits);
    rc = LAUNCH_CHECK();
  }
  return rc;
}

static int dice_ex_check(const mi355_dice_opts* o, int32_t n, int32_t c, int64_t voxels) {
  if (!o || n <= 0 || c <= 0 || voxels <= 0) return MI355_EINVAL;
  if (o->activation < MI355_DICE_ACT_NONE || o->activation > MI355_DICE_ACT_SOFTMAX) return MI355_EINVAL;
  if (o->target_kind < MI355_DICE_TARGET_F32 || o->target_kind > MI355_DICE_TARGET_LABELS) return MI355_EINVAL;
  if (o->reduction < MI355_DICE_REDUCE_MEAN || o->reduction > MI355_DICE_REDUCE_NONE) return MI355_EINVAL;
  if (!o->include_background && c < 2) return MI355_EINVAL;
  if (c > DICE_MAX_C || (size_t)n * c > 1024) return MI355_EUNSUPPORTED;
  return MI355_OK;
}

extern "C" int mi355_dice_ex_forward(const mi355_dice_opts* o, const float* logits, const void* target, int32_t n, int32_t c, int64_t voxels,
                                     float* loss, void* ws, size_t ws_bytes, void* stream) {
  int rc = dice_ex_check(o, n, c, voxels); if (rc) return rc;
  if (!logits || !target || !loss || !ws) return MI355_EINVAL;
  if (ws_bytes < mi355_dice_workspace(n, c, voxels)) return MI355_EWORKSPACE;
  const int NC = n * c, B = dice_blocks(voxels);
  float* part = (float*)ws; float* stats = part + (size_t)NC * B * 3; float* coef = stats + (size_t)NC * 3;
  LAUNCH(dice_ex_partial_kernel, dim3(B, n), dim3(256), 0, stream, logits, target, o->target_kind, (long long)voxels, c, o->activation,
         o->squared_pred, part);
  rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(dice_finalize_kernel, dim3(1), dim3(256), 0, stream, (const float*)part, B, n, c, o->batch, o->smooth_nr, o->smooth_dr, 1.0f,
         MI355_DICE_PLAIN, o->include_background ? 0 : 1, stats, coef, loss, o->jaccard, o->class_weight, o->reduction);
  return LAUNCH_CHECK();
}

extern "C" int mi355_dice_ex_backward(const mi355_dice_opts* o, const float* logits, const void* target, int32_t n, int32_t c, int64_t voxels,
                                      const float* upstream, int32_t n_upstream, float* dlogits, const void* ws, void* stream) {
  int rc = dice_ex_check(o, n, c, voxels); if (rc) return rc;
  if (!logits || !target || !dlogits || !ws) return MI355_EINVAL;
  const int ce = c - (o->include_background ? 0 : 1);
  const int terms = o->reduction == MI355_DICE_REDUCE_NONE ? (o->batch ? ce : n * ce) : 1;
  if (upstream && n_upstream != terms) return MI355_EINVAL;
  const int NC = n * c, B = dice_blocks(voxels);
  const float* coef = (const float*)ws + (size_t)NC * B * 3 + (size_t)NC * 3;
  const long long total = (long long)n * voxels;
  long long grid = (total + 255) / 256; if (grid > 16384) grid = 16384;
  LAUNCH(dice_ex_grad_kernel, dim3((unsigned)grid), dim3(256), 0, stream, logits, target, o->target_kind, (long long)voxels, n, c, o->activation,
         o->squared_pred, coef, upstream, n_upstream, o->batch, o->include_background ? 0 : 1, dlogits);
  return LAUNCH_CHECK();
}

__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long long count, float step_size, float b1, float omb1, float b2,
                            float omb2, float eps, float wd, float bc2_sqrt, float gscale) {
  const long long n4 = count / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
    float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float gg = ge[e] * gscale;
      if (wd != 0.f) gg += wd * pe[e];
      me[e] = b1 * me[e] + omb1 * gg;
      ve[e] = b2 * ve[e] + omb2 * gg * gg;
      const float denom = sqrtf(ve[e]) / bc2_sqrt + eps;
      pe[e] -= step_size * (me[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
  // tail
  const long long tail0 = n4 * 4;
  const long long t = tail0 + (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < count) {
    float gg = g[t] * gscale;
    if (wd != 0.f) gg += wd * p[t];
    const float mm = b1 * m[t] + omb1 * gg;
    const float vv = b2 * v[t] + omb2 * gg * gg;
    m[t] = mm; v[t] = vv;
    p[t] -= step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

extern "C" int mi355_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t count,
                               double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step, float grad_scale,
                               void* stream) {
  if (!param || !grad || !exp_avg || !exp_avg_sq || count <= 0 || step < 1) return MI355_EINVAL;
  if (((uintptr_t)param & 15) || ((uintptr_t)grad & 15) || ((uintptr_t)exp_avg & 15) || ((uintptr_t)exp_avg_sq & 15)) return MI355_EINVAL;
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  long long grid = (count / 4 + 255) / 256; if (grid > 8192) grid = 8192; if (grid < 1) grid = 1;
  LAUNCH(adam_kernel, dim3((unsigned)grid), dim3(256), 0, stream, param, grad, exp_avg, exp_avg_sq, (long long)count, (float)(lr / bc1), (float)beta1,
         (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)sqrt(bc2), grad_scale);
  return LAUNCH_CHECK();
}

extern "C" const char* mi355_version(void) {
#ifdef MI355_EMU
  return "mi355_unet3d cpu-emulator (tests only) 0.1";
#else
  return "mi355_unet3d gfx950 0.1";
#endif
}
