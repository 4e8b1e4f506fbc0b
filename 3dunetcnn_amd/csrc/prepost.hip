// The steps either side of the network, on the device -- HBM-bound streaming kernels.
//
//  post: unet3d/predict/volumetric.py:151-156 applies sigmoid / softmax(dim=1) to the logits; the label map is then decoded by
//        unet3d/utils/one_hot.py:44-118 (convert_one_hot_to_label_map: hierarchy decode :101-118, or any/sum threshold mask +
//        argmax :68-92). One pass over the logits produces the probabilities and/or the int16 label map.
//  pre:  unet3d/utils/one_hot.py:7-37 (compile_one_hot_encoding: round, isclose against label values or label groups -> uint8
//        channels; used by transforms/one_hot.py:7-16) and MONAI NormalizeIntensityD(channel_wise=True, nonzero=False)
//        (datasets/segmentation.py:77-86): per-channel (x - mean) / std, population std, std == 0 -> 1.
#include "hipcompat.h"
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
