// The steps either side of the network, on the device -- HBM-bound streaming kernels.
//
//  post: unet3d/predict/volumetric.py:151-156 applies sigmoid / softmax(dim=1) to the logits; the label map is then decoded by
//        unet3d/utils/one_hot.py:44-118 (convert_one_hot_to_label_map: hierarchy decode :101-118, or any/sum threshold mask +
//        argmax :68-92). One pass over the logits produces the probabilities and/or the int16 label map.
//  pre:  unet3d/utils/one_hot.py:7-37 (compile_one_hot_encoding: round, isclose against label values or label groups -> uint8
//        channels; used by transforms/one_hot.py:7-16) and MONAI NormalizeIntensityD(channel_wise=True, nonzero=False)
//        (datasets/segmentation.py:77-86): per-channel (x - mean) / std, population std, std == 0 -> 1.
#include "gfx950_dialect.h"
#include "../../include/mi355_unet3d.h"

#define PP_MAX_C 16

static inline unsigned pp_grid(long long total) {
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

__global__ void postprocess_kernel(const float* logits, int C, long long V, int act, float thr, const short* labels, int hierarchy,
                                   int sum_then_thr, float* probs, short* label_map) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long long)gridDim.x * blockDim.x) {
    float p[PP_MAX_C];
    float mx = -3.0e38f;
    for (int c = 0; c < C; ++c) { p[c] = logits[(size_t)c * V + v]; mx = p[c] > mx ? p[c] : mx; }
    if (act == 1) {
      for (int c = 0; c < C; ++c) p[c] = 1.f / (1.f + expf(-p[c]));
    } else if (act == 2) {
      float s = 0.f;
      for (int c = 0; c < C; ++c) { p[c] = expf(p[c] - mx); s += p[c]; }
      for (int c = 0; c < C; ++c) p[c] = p[c] / s;
    }
    if (probs) for (int c = 0; c < C; ++c) probs[(size_t)c * V + v] = p[c];
    if (label_map) {
      short lab = 0;
      if (hierarchy) {
        bool roi = true;                         // each label is contained in the region of the previous one (one_hot.py:101-118)
        for (int c = 0; c < C; ++c) { roi = roi && (p[c] > thr); if (roi) lab = labels[c]; }
      } else {
        bool mask; float sum = 0.f; bool any = false; int arg = 0; float best = p[0];
        for (int c = 0; c < C; ++c) { sum += p[c]; any = any || (p[c] > thr); if (p[c] > best) { best = p[c]; arg = c; } }
        mask = sum_then_thr ? (sum > thr) : any;   // mask_encoding (one_hot.py:77-81); argmax = first maximum (assign_labels :84-92)
        if (mask) lab = labels[arg];
      }
      label_map[v] = lab;
    }
  }
}

extern "C" int mi355_postprocess(const float* logits, int32_t c, int64_t voxels, int32_t activation, float threshold,
                                 const int16_t* labels, int32_t hierarchy, int32_t sum_then_threshold, float* probs,
                                 int16_t* label_map, void* stream) {
  if (!logits || c < 1 || c > PP_MAX_C || voxels <= 0 || activation < 0 || activation > 2) return MI355_EINVAL;
  if (label_map && !labels) return MI355_EINVAL;
  if (!probs && !label_map) return MI355_EINVAL;
  LAUNCH(postprocess_kernel, dim3(pp_grid(voxels)), dim3(256), 0, stream, logits, c, (long long)voxels, activation, threshold,
         (const short*)labels, hierarchy, sum_then_threshold, probs, (short*)label_map);
  return LAUNCH_CHECK();
}

// one-hot: channel g is 1 where round(label_map) isclose (atol 1e-8, rtol 1e-5) to any value of group g
__global__ void one_hot_kernel(const float* lm, long long V, const float* vals, const int* offs, int C, unsigned char* out) {
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += (long long)gridDim.x * blockDim.x) {
    const float d = rintf(lm[v]);                // torch.round(decimals=0): half to even
    for (int g = 0; g < C; ++g) {
      unsigned char hit = 0;
      for (int k = offs[g]; k < offs[g + 1]; ++k) {
        const float b = vals[k];
        if (fabsf(d - b) <= 1e-8f + 1e-5f * fabsf(b)) hit = 1;
      }
      out[(size_t)g * V + v] = hit;
    }
  }
}

extern "C" int mi355_one_hot(const float* label_map, int64_t voxels, const float* label_values, const int32_t* group_offsets,
                             int32_t c, uint8_t* out, void* stream) {
  if (!label_map || !label_values || !group_offsets || !out || c < 1 || voxels <= 0) return MI355_EINVAL;
  LAUNCH(one_hot_kernel, dim3(pp_grid(voxels)), dim3(256), 0, stream, label_map, (long long)voxels, label_values, (const int*)group_offsets, c,
         (unsigned char*)out);
  return LAUNCH_CHECK();
}

// z-score: per-channel shifted sums (K = first voxel) -> mean, 1/std -> apply
#define ZS_BLOCKS 256
__global__ void zscore_partial_kernel(const float* x, long long V, float* ws) {
  __shared__ float r0[256], r1[256];
  const int c = blockIdx.y, blk = blockIdx.x;
  const float* xc = x + (size_t)c * V;
  const float K = xc[0];
  float s0 = 0.f, s1 = 0.f;
  for (long long v = (long long)blk * blockDim.x + threadIdx.x; v < V; v += (long long)gridDim.x * blockDim.x) {
    const float t = xc[v] - K; s0 += t; s1 += t * t;
  }
  r0[threadIdx.x] = s0; r1[threadIdx.x] = s1;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) { r0[threadIdx.x] += r0[threadIdx.x + s]; r1[threadIdx.x] += r1[threadIdx.x + s]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { ws[((size_t)c * gridDim.x + blk) * 2] = r0[0]; ws[((size_t)c * gridDim.x + blk) * 2 + 1] = r1[0]; }
}
__global__ void zscore_finalize_kernel(const float* x, long long V, int B, float* ws, float* stats) {
  const int c = blockIdx.x;
  if (threadIdx.x != 0) return;
  double s0 = 0.0, s1 = 0.0;
  for (int b = 0; b < B; ++b) { s0 += (double)ws[((size_t)c * B + b) * 2]; s1 += (double)ws[((size_t)c * B + b) * 2 + 1]; }
  const double K = (double)x[(size_t)c * V];
  const double mean = K + s0 / (double)V;
  double var = (s1 - s0 * s0 / (double)V) / (double)V;       // population variance (torch.std(unbiased=False))
  if (var < 0.0) var = 0.0;
  double sd = sqrt(var);
  if (sd == 0.0) sd = 1.0;                                    // MONAI NormalizeIntensity: divisor 0 -> 1
  stats[2 * c] = (float)mean; stats[2 * c + 1] = (float)(1.0 / sd);
}
__global__ void zscore_apply_kernel(const float* x, float* y, long long V, int C, const float* stats) {
  const long long total = (long long)C * V;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i / V);
    y[i] = (x[i] - stats[2 * c]) * stats[2 * c + 1];
  }
}

extern "C" size_t mi355_zscore_workspace(int32_t c) { return ((size_t)c * ZS_BLOCKS * 2 + (size_t)c * 2) * sizeof(float); }

extern "C" int mi355_zscore(const float* x, float* y, int32_t c, int64_t voxels, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !y || !ws || c < 1 || voxels <= 0) return MI355_EINVAL;
  if (ws_bytes < mi355_zscore_workspace(c)) return MI355_EWORKSPACE;
  float* part = (float*)ws; float* stats = part + (size_t)c * ZS_BLOCKS * 2;
  LAUNCH(zscore_partial_kernel, dim3(ZS_BLOCKS, c), dim3(256), 0, stream, x, (long long)voxels, part);
  int rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(zscore_finalize_kernel, dim3(c), dim3(64), 0, stream, x, (long long)voxels, ZS_BLOCKS, part, stats);
  rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(zscore_apply_kernel, dim3(pp_grid((long long)c * voxels)), dim3(256), 0, stream, x, y, (long long)voxels, c, (const float*)stats);
  return LAUNCH_CHECK();
}

// ---- affine resampling (ResizeD / ResampleToMatch) ----
struct ResampleM { float m[12]; };

__global__ void resample_affine_kernel(const float* src, float* dst, int C, int sd, int sh, int sw, int dd, int dh, int dw, ResampleM M,
                                       int mode, int padding) {
  const long long DV = (long long)dd * dh * dw, SV = (long long)sd * sh * sw;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < DV; v += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(v % dw), y = (int)((v / dw) % dh), z = (int)(v / ((long long)dw * dh));
    float cz = M.m[0] * z + M.m[1] * y + M.m[2] * x + M.m[3];
    float cy = M.m[4] * z + M.m[5] * y + M.m[6] * x + M.m[7];
    float cx = M.m[8] * z + M.m[9] * y + M.m[10] * x + M.m[11];
    if (mode != MI355_RESAMPLE_TRILINEAR) {
      const float rz = mode == MI355_RESAMPLE_NEAREST ? rintf(cz) : floorf(cz);
      const float ry = mode == MI355_RESAMPLE_NEAREST ? rintf(cy) : floorf(cy);
      const float rx = mode == MI355_RESAMPLE_NEAREST ? rintf(cx) : floorf(cx);
      int iz = (int)rz, iy = (int)ry, ix = (int)rx;
      const bool inside = iz >= 0 && iy >= 0 && ix >= 0 && iz < sd && iy < sh && ix < sw;
      iz = iz < 0 ? 0 : (iz < sd ? iz : sd - 1); iy = iy < 0 ? 0 : (iy < sh ? iy : sh - 1); ix = ix < 0 ? 0 : (ix < sw ? ix : sw - 1);
      const size_t o = ((size_t)iz * sh + iy) * sw + ix;
      for (int c = 0; c < C; ++c) dst[(size_t)c * DV + v] = (padding == 1 && !inside) ? 0.f : src[(size_t)c * SV + o];
      continue;
    }
    if (padding == 0) {
      cz = cz < 0.f ? 0.f : (cz > (float)(sd - 1) ? (float)(sd - 1) : cz);
      cy = cy < 0.f ? 0.f : (cy > (float)(sh - 1) ? (float)(sh - 1) : cy);
      cx = cx < 0.f ? 0.f : (cx > (float)(sw - 1) ? (float)(sw - 1) : cx);
    }
    const float fz = floorf(cz), fy = floorf(cy), fx = floorf(cx);
    const float lz = cz - fz, ly = cy - fy, lx = cx - fx;
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    float wgt[8]; size_t off[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int a = k >> 2, b = (k >> 1) & 1, e = k & 1;
      int iz = z0 + a, iy = y0 + b, ix = x0 + e;
      float w = (a ? lz : 1.f - lz) * (b ? ly : 1.f - ly) * (e ? lx : 1.f - lx);
      const bool inside = iz >= 0 && iy >= 0 && ix >= 0 && iz < sd && iy < sh && ix < sw;
      if (!inside) {
        if (padding == 1) w = 0.f;               // zeros outside; with border padding the clamp above leaves only weight-0 corners outside
        iz = iz < 0 ? 0 : (iz < sd ? iz : sd - 1); iy = iy < 0 ? 0 : (iy < sh ? iy : sh - 1); ix = ix < 0 ? 0 : (ix < sw ? ix : sw - 1);
      }
      wgt[k] = w; off[k] = ((size_t)iz * sh + iy) * sw + ix;
    }
    for (int c = 0; c < C; ++c) {
      const float* sc = src + (size_t)c * SV;
      // x pairs first, then y, then z
      const float v00 = wgt[0] * sc[off[0]] + wgt[1] * sc[off[1]], v01 = wgt[2] * sc[off[2]] + wgt[3] * sc[off[3]];
      const float v10 = wgt[4] * sc[off[4]] + wgt[5] * sc[off[5]], v11 = wgt[6] * sc[off[6]] + wgt[7] * sc[off[7]];
      dst[(size_t)c * DV + v] = (v00 + v01) + (v10 + v11);
    }
  }
}

extern "C" int mi355_resample_affine(const float* src, float* dst, int32_t c, int32_t sd, int32_t sh, int32_t sw, int32_t dd, int32_t dh,
                                     int32_t dw, const float* m, int32_t mode, int32_t padding, void* stream) {
  if (!src || !dst || !m || c < 1 || sd < 1 || sh < 1 || sw < 1 || dd < 1 || dh < 1 || dw < 1) return MI355_EINVAL;
  if (mode < MI355_RESAMPLE_TRILINEAR || mode > MI355_RESAMPLE_NEAREST_FLOOR || padding < 0 || padding > 1) return MI355_EINVAL;
  ResampleM M;
  for (int i = 0; i < 12; ++i) M.m[i] = m[i];
  LAUNCH(resample_affine_kernel, dim3(pp_grid((long long)dd * dh * dw)), dim3(256), 0, stream, src, dst, c, sd, sh, sw, dd, dh, dw, M, mode, padding);
  return LAUNCH_CHECK();
}
