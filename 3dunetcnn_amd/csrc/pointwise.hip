// HBM-bound streaming kernels around the conv stack: trilinear x2 upsample (+pad/crop, + concat slice write),
// layout changes at the module boundary, residual-gradient add, Dropout3d scale, and the few-class 1x1x1 projection.
//
// Reference call sites: unet3d/models/pytorch/classification/decoder.py:105-106 (F.interpolate trilinear x2,
// align_corners=False), segmentation/unet.py:34-42 (F.pad + torch.cat), autoencoder/variational.py:59-60 and
// segmentation/unet.py:50 (final 1x1x1 conv), classification/myronenko.py:70-79 (Dropout3d).
#include "gfx950_dialect.h"
#include <cstdlib>
#include "../../include/mi355_unet3d.h"
#include "act_io.h"

__device__ __forceinline__ void tri_src(int u, int n, int& i0, int& i1, float& l0, float& l1) {
  // PyTorch area_pixel_compute_source_index(scale=0.5, align_corners=False): src = max(0.5*(u+0.5)-0.5, 0)
  float s = 0.5f * ((float)u + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  i1 = i0 + 1 < n ? i0 + 1 : n - 1;
  l1 = s - (float)i0;
  l0 = 1.f - l1;
}

template <typename T, int VW>
__global__ void upsample2x_fwd_kernel(const T* lo, int lold, int N, int Dl, int Hl, int Wl, int C,
                                      T* cat, int catld, int Dc, int Hc, int Wc, int offz, int offy, int offx) {
  const int Q = C / VW;
  const long long total = (long long)N * Dc * Hc * Wc * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q); long long r = idx / Q;
    const int x = (int)(r % Wc); r /= Wc;
    const int y = (int)(r % Hc); r /= Hc;
    const int z = (int)(r % Dc); const int n = (int)(r / Dc);
    const int uz = z - offz, uy = y - offy, ux = x - offx;
    float o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = 0.f;
    if (uz >= 0 && uy >= 0 && ux >= 0 && uz < 2 * Dl && uy < 2 * Hl && ux < 2 * Wl) {
      int z0, z1, y0, y1, x0, x1; float lz0, lz1, ly0, ly1, lx0, lx1;
      tri_src(uz, Dl, z0, z1, lz0, lz1); tri_src(uy, Hl, y0, y1, ly0, ly1); tri_src(ux, Wl, x0, x1, lx0, lx1);
      const int zs[2] = {z0, z1}, ys[2] = {y0, y1}, xs[2] = {x0, x1};
      const float wz[2] = {lz0, lz1}, wy[2] = {ly0, ly1}, wx[2] = {lx0, lx1};
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float w = wz[a] * wy[b] * wx[c];
            float v[VW];
            ldv<VW>(lo + ((((size_t)n * Dl + zs[a]) * Hl + ys[b]) * Wl + xs[c]) * lold + VW * q, v);
#pragma unroll
            for (int e = 0; e < VW; ++e) o[e] += w * v[e];
          }
    }
    stv<VW>(cat + ((((size_t)n * Dc + z) * Hc + y) * Wc + x) * catld + VW * q, o);
  }
}

__device__ __forceinline__ float tri_weight(int u, int n, int d) {
  if (u < 0 || u >= 2 * n) return 0.f;
  int i0, i1; float l0, l1;
  tri_src(u, n, i0, i1, l0, l1);
  return (i0 == d ? l0 : 0.f) + (i1 == d ? l1 : 0.f);
}

template <typename T, int VW>
__global__ void upsample2x_bwd_kernel(const T* dcat, int catld, int N, int Dc, int Hc, int Wc, int C,
                                      T* dlo, int lold, int Dl, int Hl, int Wl, int offz, int offy, int offx) {
  const int Q = C / VW;
  const long long total = (long long)N * Dl * Hl * Wl * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q); long long r = idx / Q;
    const int x = (int)(r % Wl); r /= Wl;
    const int y = (int)(r % Hl); r /= Hl;
    const int z = (int)(r % Dl); const int n = (int)(r / Dl);
    float o[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) o[e] = 0.f;
    // the 4 x 4 x 4 fine voxels that can reach this coarse voxel: the 12 axis weights once (a tap outside the fine tensor or the crop window
    // weighs 0 and reads a clamped address), then 64 unconditional loads in the fixed (z, y, x) order -- a zero weight adds +0, so the sum is
    // the one the branching form produced. (That form evaluated 84 weights per coarse voxel and branched around every load.)
    float wz[4], wy[4], wx[4]; int pz[4], py[4], px[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int uz = 2 * z + a - 1, cz = uz + offz; const bool okz = cz >= 0 && cz < Dc;
      wz[a] = okz ? tri_weight(uz, Dl, z) : 0.f; pz[a] = cz < 0 ? 0 : (cz < Dc ? cz : Dc - 1);
      const int uy = 2 * y + a - 1, cy = uy + offy; const bool oky = cy >= 0 && cy < Hc;
      wy[a] = oky ? tri_weight(uy, Hl, y) : 0.f; py[a] = cy < 0 ? 0 : (cy < Hc ? cy : Hc - 1);
      const int ux = 2 * x + a - 1, cx = ux + offx; const bool okx = cx >= 0 && cx < Wc;
      wx[a] = okx ? tri_weight(ux, Wl, x) : 0.f; px[a] = cx < 0 ? 0 : (cx < Wc ? cx : Wc - 1);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (wz[a] == 0.f) continue;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        if (wy[b] == 0.f) continue;
        const float wzy = wz[a] * wy[b];
        const T* row = dcat + (((size_t)n * Dc + pz[a]) * Hc + py[b]) * (size_t)Wc * catld + VW * q;
        float v[4][VW];
#pragma unroll
        for (int c = 0; c < 4; ++c) ldv<VW>(row + (size_t)px[c] * catld, v[c]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          if (wx[c] == 0.f) continue;
          const float w = wzy * wx[c];
#pragma unroll
          for (int e = 0; e < VW; ++e) o[e] += w * v[c][e];
        }
      }
    }
    stv<VW>(dlo + ((((size_t)n * Dl + z) * Hl + y) * Wl + x) * lold + VW * q, o);
  }
}

// ---- the forward pass with the index arithmetic out of the element loop ----
// The grid-stride forms above spend four 64-bit divisions per 16-byte run (~300 vector instructions per run: they ran at 2 TB/s, bound by
// the vector ALU). Here a block walks ROWS (n, z, y): the row's decomposition and its z / y interpolation weights are computed once per
// row, a thread keeps one channel run q = tid % Q and strides x, so the element loop holds the x weights, the loads and the FMAs. Same
// taps, same weights, same accumulation order as the forms above (which remain for channel counts with 256 % Q != 0).
template <typename T, int VW>
__global__ __launch_bounds__(256) void upsample2x_fwd_rows_kernel(const T* lo, int lold, int N, int Dl, int Hl, int Wl, int C,
                                                                  T* cat, int catld, int Dc, int Hc, int Wc, int offz, int offy, int offx) {
  const int Q = C / VW, XPB = 256 / Q;
  const int q = threadIdx.x % Q, xl = threadIdx.x / Q;
  const int rows = N * Dc * Hc;
  for (int row = blockIdx.x; row < rows; row += gridDim.x) {
    const int y = row % Hc, t = row / Hc, z = t % Dc, n = t / Dc;
    const int uz = z - offz, uy = y - offy;
    const bool rowin = uz >= 0 && uy >= 0 && uz < 2 * Dl && uy < 2 * Hl;
    int z0 = 0, z1 = 0, y0 = 0, y1 = 0; float lz0 = 0.f, lz1 = 0.f, ly0 = 0.f, ly1 = 0.f;
    if (rowin) { tri_src(uz, Dl, z0, z1, lz0, lz1); tri_src(uy, Hl, y0, y1, ly0, ly1); }
    const size_t srow = (size_t)Wl * lold;
    const T* r00 = lo + (((size_t)n * Dl + z0) * Hl + y0) * srow + VW * q;
    const T* r01 = lo + (((size_t)n * Dl + z0) * Hl + y1) * srow + VW * q;
    const T* r10 = lo + (((size_t)n * Dl + z1) * Hl + y0) * srow + VW * q;
    const T* r11 = lo + (((size_t)n * Dl + z1) * Hl + y1) * srow + VW * q;
    const float w00 = lz0 * ly0, w01 = lz0 * ly1, w10 = lz1 * ly0, w11 = lz1 * ly1;
    T* orow = cat + (((size_t)n * Dc + z) * Hc + y) * (size_t)Wc * catld + VW * q;
    for (int x = xl; x < Wc; x += XPB) {
      const int ux = x - offx;
      float o[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) o[e] = 0.f;
      if (rowin && ux >= 0 && ux < 2 * Wl) {
        int x0, x1; float lx0, lx1;
        tri_src(ux, Wl, x0, x1, lx0, lx1);
        const size_t o0 = (size_t)x0 * lold, o1 = (size_t)x1 * lold;
        float v[8][VW];
        ldv<VW>(r00 + o0, v[0]); ldv<VW>(r00 + o1, v[1]); ldv<VW>(r01 + o0, v[2]); ldv<VW>(r01 + o1, v[3]);
        ldv<VW>(r10 + o0, v[4]); ldv<VW>(r10 + o1, v[5]); ldv<VW>(r11 + o0, v[6]); ldv<VW>(r11 + o1, v[7]);
        const float w[8] = {w00 * lx0, w00 * lx1, w01 * lx0, w01 * lx1, w10 * lx0, w10 * lx1, w11 * lx0, w11 * lx1};
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
          for (int e = 0; e < VW; ++e) o[e] += w[k] * v[k][e];
      }
      stv<VW>(orow + (size_t)x * catld, o);
    }
  }
}

// (The backward pass was measured in the same form and did not move -- 64 loads per coarse voxel bound it, not the index arithmetic; it keeps
// the grid-stride kernel above with its weights hoisted: 0.435 -> 0.381 ms over the three launches of the fp32 step, 0.499 -> 0.382 in bf16.)

// rows form: a 256-thread block holds whole voxels of Q channel runs and the row count fits an int; MI355_UPSAMPLE_ROWS=0 (read once): never
static bool upsample_rows_ok(int Q, long long rows) {
  static const bool off = [] { const char* v = getenv("MI355_UPSAMPLE_ROWS"); return v && v[0] == '0'; }();
  return !off && Q >= 1 && Q <= 256 && 256 % Q == 0 && rows > 0 && rows <= 0x7fffffffLL;
}
static unsigned rows_grid(long long rows) { return (unsigned)(rows < 16384 ? rows : 16384); }

static inline unsigned grid_for(long long total) {
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  if (g < 1) g = 1;
  return (unsigned)g;
}

static int act_ok(const mi355_act* t) { return act_view_ok(t); }

extern "C" int mi355_upsample2x_fwd(const mi355_act* lo, const mi355_act* cat, int32_t offz, int32_t offy, int32_t offx, void* stream) {
  if (!act_ok(lo) || !act_ok(cat) || lo->c != cat->c || lo->n != cat->n) return MI355_EINVAL;
  if (lo->dtype != cat->dtype) return MI355_EUNSUPPORTED;
  const long long total = (long long)cat->n * cat->d * cat->h * cat->w * (cat->c / 4);
  {
    const bool v8 = act_vw8(lo) && act_vw8(cat);
    const long long rows = (long long)cat->n * cat->d * cat->h;
    if (upsample_rows_ok(cat->c / (v8 ? 8 : 4), rows)) {
      if (v8)
        ACT_TYPED_LP16(lo->dtype, T, LAUNCH((upsample2x_fwd_rows_kernel<T, 8>), dim3(rows_grid(rows)), dim3(256), 0, stream, (const T*)lo->p, lo->ld, lo->n, lo->d,
               lo->h, lo->w, lo->c, (T*)cat->p, cat->ld, cat->d, cat->h, cat->w, offz, offy, offx));
      else
        ACT_TYPED(lo->dtype, T, LAUNCH((upsample2x_fwd_rows_kernel<T, 4>), dim3(rows_grid(rows)), dim3(256), 0, stream, (const T*)lo->p, lo->ld, lo->n, lo->d,
                                       lo->h, lo->w, lo->c, (T*)cat->p, cat->ld, cat->d, cat->h, cat->w, offz, offy, offx));
      return LAUNCH_CHECK();
    }
  }
  if (act_vw8(lo) && act_vw8(cat))
    ACT_TYPED_LP16(lo->dtype, T, LAUNCH((upsample2x_fwd_kernel<T, 8>), dim3(grid_for(total / 2)), dim3(256), 0, stream, (const T*)lo->p, lo->ld, lo->n, lo->d, lo->h,
           lo->w, lo->c, (T*)cat->p, cat->ld, cat->d, cat->h, cat->w, offz, offy, offx));
  else
    ACT_TYPED(lo->dtype, T, LAUNCH((upsample2x_fwd_kernel<T, 4>), dim3(grid_for(total)), dim3(256), 0, stream, (const T*)lo->p, lo->ld, lo->n, lo->d, lo->h,
                                   lo->w, lo->c, (T*)cat->p, cat->ld, cat->d, cat->h, cat->w, offz, offy, offx));
  return LAUNCH_CHECK();
}

extern "C" int mi355_upsample2x_bwd(const mi355_act* dcat, const mi355_act* dlo, int32_t offz, int32_t offy, int32_t offx, void* stream) {
  if (!act_ok(dlo) || !act_ok(dcat) || dlo->c != dcat->c || dlo->n != dcat->n) return MI355_EINVAL;
  if (dlo->dtype != dcat->dtype) return MI355_EUNSUPPORTED;
  const long long total = (long long)dlo->n * dlo->d * dlo->h * dlo->w * (dlo->c / 4);
  if (act_vw8(dlo) && act_vw8(dcat))
    ACT_TYPED_LP16(dlo->dtype, T, LAUNCH((upsample2x_bwd_kernel<T, 8>), dim3(grid_for(total / 2)), dim3(256), 0, stream, (const T*)dcat->p, dcat->ld, dcat->n, dcat->d,
           dcat->h, dcat->w, dcat->c, (T*)dlo->p, dlo->ld, dlo->d, dlo->h, dlo->w, offz, offy, offx));
  else
    ACT_TYPED(dlo->dtype, T, LAUNCH((upsample2x_bwd_kernel<T, 4>), dim3(grid_for(total)), dim3(256), 0, stream, (const T*)dcat->p, dcat->ld, dcat->n, dcat->d,
                                    dcat->h, dcat->w, dcat->c, (T*)dlo->p, dlo->ld, dlo->d, dlo->h, dlo->w, offz, offy, offx));
  return LAUNCH_CHECK();
}

// ---- layout ------------------------------------------------------------------------------------
template <typename T>
__global__ void ncdhw_to_ndhwc_kernel(const float* src, T* dst, int dld, int N, int C, long long V) {
  const long long total = (long long)N * V * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    // read-coalesced over v for small C: idx = (n*C + c)*V + v
    const long long v = idx % V; const long long nc = idx / V;
    const int c = (int)(nc % C); const long long n = nc / C;
    st1(dst + ((size_t)n * V + v) * dld + c, src[idx]);
  }
}
template <typename T>
__global__ void ndhwc_to_ncdhw_kernel(const T* src, int sld, float* dst, int N, int C, long long V) {
  const long long total = (long long)N * V * C;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long v = idx % V; const long long nc = idx / V;
    const int c = (int)(nc % C); const long long n = nc / C;
    dst[idx] = ld1(src + ((size_t)n * V + v) * sld + c);
  }
}

extern "C" int mi355_ncdhw_to_ndhwc(const float* src, const mi355_act* dst, void* stream) {
  if (!src || !dst || !dst->p || dst->ld < dst->c || !act_dtype_ok(dst)) return MI355_EINVAL;
  const long long V = (long long)dst->d * dst->h * dst->w;
  ACT_TYPED(dst->dtype, T, LAUNCH(ncdhw_to_ndhwc_kernel<T>, dim3(grid_for((long long)dst->n * V * dst->c)), dim3(256), 0, stream, src, (T*)dst->p, dst->ld,
                                  dst->n, dst->c, V));
  return LAUNCH_CHECK();
}
extern "C" int mi355_ndhwc_to_ncdhw(const mi355_act* src, float* dst, void* stream) {
  if (!dst || !src || !src->p || src->ld < src->c || !act_dtype_ok(src)) return MI355_EINVAL;
  const long long V = (long long)src->d * src->h * src->w;
  ACT_TYPED(src->dtype, T, LAUNCH(ndhwc_to_ncdhw_kernel<T>, dim3(grid_for((long long)src->n * V * src->c)), dim3(256), 0, stream, (const T*)src->p, src->ld,
                                  dst, src->n, src->c, V));
  return LAUNCH_CHECK();
}

// ---- add / channel scale ---------------------------------------------------------------------------
template <typename T>
__global__ void add_kernel(const T* a, int ald, const T* b, int bld, T* y, int yld, long long NV, int C) {
  const int Q = C / 4;
  const long long total = NV * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q); const long long v = idx / Q;
    const float4 av = ld4(a + (size_t)v * ald + 4 * q);
    const float4 bv = ld4(b + (size_t)v * bld + 4 * q);
    st4(y + (size_t)v * yld + 4 * q, make_float4(av.x + bv.x, av.y + bv.y, av.z + bv.z, av.w + bv.w));
  }
}
template <typename T, int VW>
__global__ void chscale_kernel(const T* x, int xld, const float* s, T* y, int yld, long long V, int N, int C) {
  const int Q = C / VW;
  const long long total = (long long)N * V * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q); const long long nv = idx / Q; const int n = (int)(nv / V);
    float xv[VW], sv[VW];
    ldv<VW>(x + (size_t)nv * xld + VW * q, xv);
    ldv<VW>(s + (size_t)n * C + VW * q, sv);
#pragma unroll
    for (int e = 0; e < VW; ++e) xv[e] *= sv[e];
    stv<VW>(y + (size_t)nv * yld + VW * q, xv);
  }
}
// change of storage type (fp32 <-> bf16, or a copy): the bridge to a kernel that has no bf16 form, and the 16-bit copy of the network input
template <typename TS, typename TD>
__global__ void cast_kernel(const TS* x, int xld, TD* y, int yld, long long NV, int C) {
  const int Q = C / 4;
  const long long total = NV * Q;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int q = (int)(idx % Q); const long long v = idx / Q;
    st4(y + (size_t)v * yld + 4 * q, ld4(x + (size_t)v * xld + 4 * q));
  }
}

static int same_shape(const mi355_act* a, const mi355_act* b) {
  return a->n == b->n && a->d == b->d && a->h == b->h && a->w == b->w && a->c == b->c;
}

extern "C" int mi355_add(const mi355_act* a, const mi355_act* b, const mi355_act* y, void* stream) {
  if (!act_ok(a) || !act_ok(b) || !act_ok(y) || !same_shape(a, b) || !same_shape(a, y)) return MI355_EINVAL;
  if (a->dtype != b->dtype || a->dtype != y->dtype) return MI355_EUNSUPPORTED;
  const long long NV = (long long)a->n * a->d * a->h * a->w;
  ACT_TYPED(a->dtype, T, LAUNCH(add_kernel<T>, dim3(grid_for(NV * (a->c / 4))), dim3(256), 0, stream, (const T*)a->p, a->ld, (const T*)b->p, b->ld, (T*)y->p,
                                y->ld, NV, a->c));
  return LAUNCH_CHECK();
}
extern "C" int mi355_cast(const mi355_act* x, const mi355_act* y, void* stream) {
  if (!act_ok(x) || !act_ok(y) || !same_shape(x, y)) return MI355_EINVAL;
  const long long NV = (long long)x->n * x->d * x->h * x->w;
  const dim3 grid(grid_for(NV * (x->c / 4)));
  ACT_TYPED(x->dtype, TS, ACT_TYPED(y->dtype, TD, LAUNCH((cast_kernel<TS, TD>), grid, dim3(256), 0, stream, (const TS*)x->p, x->ld, (TD*)y->p, y->ld, NV, x->c)));
  return LAUNCH_CHECK();
}
extern "C" int mi355_chscale(const mi355_act* x, const float* chscale, const mi355_act* y, void* stream) {
  if (!act_ok(x) || !act_ok(y) || !same_shape(x, y) || !chscale) return MI355_EINVAL;
  if (x->dtype != y->dtype) return MI355_EUNSUPPORTED;
  const long long V = (long long)x->d * x->h * x->w;
  if (act_vw8(x) && act_vw8(y))
    ACT_TYPED_LP16(x->dtype, T, LAUNCH((chscale_kernel<T, 8>), dim3(grid_for((long long)x->n * V * (x->c / 8))), dim3(256), 0, stream, (const T*)x->p, x->ld, chscale,
           (T*)y->p, y->ld, V, x->n, x->c));
  else
    ACT_TYPED(x->dtype, T, LAUNCH((chscale_kernel<T, 4>), dim3(grid_for((long long)x->n * V * (x->c / 4))), dim3(256), 0, stream, (const T*)x->p, x->ld, chscale,
                                  (T*)y->p, y->ld, V, x->n, x->c));
  return LAUNCH_CHECK();
}

// ---- few-class 1x1x1 projection: NDHWC in, NCDHW logits out ---------------------------------------------
#define PROJ_MAX_COUT 8
#define PROJ_MAX_CIN 96
#define PROJ_TV 128

// optional fused prologue act(scale[n,c]*x + shift[n,c]) applied while staging the x tile (same as the conv kernels)
__device__ __forceinline__ float4 proj_prologue(float4 v, const float* sc, const float* sh, float slope, int n, int Cin, int c) {
  if (!sc) return v;
  const float4 s4 = *reinterpret_cast<const float4*>(sc + (size_t)n * Cin + c);
  const float4 h4 = *reinterpret_cast<const float4*>(sh + (size_t)n * Cin + c);
  v.x = v.x * s4.x + h4.x; v.y = v.y * s4.y + h4.y; v.z = v.z * s4.z + h4.z; v.w = v.w * s4.w + h4.w;
  v.x = fmaxf(v.x, v.x * slope); v.y = fmaxf(v.y, v.y * slope); v.z = fmaxf(v.z, v.z * slope); v.w = fmaxf(v.w, v.w * slope);   // 0 <= slope <= 1
  return v;
}

template <typename T>
__global__ void proj_fwd_kernel(const T* x, int xld, const float* sc, const float* sh, float slope, const float* w, const float* bias,
                                float* out, int N, long long V, int Cin, int Cout) {
  DYN_LDS(lds);                    // x tile [PROJ_TV][Cin+1] | w [Cout][Cin]
  const int XS = Cin + 1;
  float* lx = lds; float* lw = lds + PROJ_TV * XS;
  const int tid = threadIdx.x, Q = Cin / 4;
  for (int i = tid; i < Cout * Cin; i += 256) lw[i] = w[i];
  const long long tilesPerN = (V + PROJ_TV - 1) / PROJ_TV;
  const long long ntiles = (long long)N * tilesPerN;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = (int)(tile / tilesPerN); const long long v0 = (tile % tilesPerN) * PROJ_TV;
    __syncthreads();
    for (int i = tid; i < PROJ_TV * Q; i += 256) {
      const int v = i / Q, q = i % Q;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v0 + v < V) val = proj_prologue(ld4(x + ((size_t)n * V + v0 + v) * xld + 4 * q), sc, sh, slope, n, Cin, 4 * q);
      float* d = lx + v * XS + 4 * q; d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    __syncthreads();
    {
      const int v = tid & (PROJ_TV - 1), h = tid >> 7;   // 256 threads: 128 voxels x 2 class-parities
      if (v0 + v < V) {
        float acc[PROJ_MAX_COUT / 2];
#pragma unroll
        for (int k = 0; k < PROJ_MAX_COUT / 2; ++k) { const int co = h + 2 * k; acc[k] = (bias && co < Cout) ? bias[co] : 0.f; }
        for (int ci = 0; ci < Cin; ++ci) {
          const float xv = lx[v * XS + ci];
#pragma unroll
          for (int k = 0; k < PROJ_MAX_COUT / 2; ++k) { const int co = h + 2 * k; if (co < Cout) acc[k] += xv * lw[co * Cin + ci]; }
        }
#pragma unroll
        for (int k = 0; k < PROJ_MAX_COUT / 2; ++k) { const int co = h + 2 * k; if (co < Cout) out[((size_t)n * Cout + co) * V + v0 + v] = acc[k]; }
      }
    }
  }
}

// dx tile + per-block partial dw/dbias
template <typename T>
__global__ void proj_bwd_kernel(const T* x, int xld, const float* sc, const float* sh, float slope, const float* w, const float* dz,
                                T* dx, int dxld, float* ws, int N, long long V, int Cin, int Cout) {
  DYN_LDS(lds);                    // x tile [TV][Cin+1] | dz tile [Cout][TV] | w [Cout][Cin]
  const int XS = Cin + 1;
  float* lx = lds; float* lz = lx + PROJ_TV * XS; float* lw = lz + PROJ_MAX_COUT * PROJ_TV;
  const int tid = threadIdx.x, Q = Cin / 4;
  for (int i = tid; i < Cout * Cin; i += 256) lw[i] = w[i];
  const int npairs = Cout * Cin + Cout;     // dw entries then dbias entries
  float part[(PROJ_MAX_COUT * PROJ_MAX_CIN + PROJ_MAX_COUT + 255) / 256];
#pragma unroll
  for (int k = 0; k < (int)(sizeof(part) / sizeof(float)); ++k) part[k] = 0.f;
  const long long tilesPerN = (V + PROJ_TV - 1) / PROJ_TV;
  const long long ntiles = (long long)N * tilesPerN;
  for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = (int)(tile / tilesPerN); const long long v0 = (tile % tilesPerN) * PROJ_TV;
    __syncthreads();
    for (int i = tid; i < PROJ_TV * Q; i += 256) {
      const int v = i / Q, q = i % Q;
      float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
      if (v0 + v < V) val = proj_prologue(ld4(x + ((size_t)n * V + v0 + v) * xld + 4 * q), sc, sh, slope, n, Cin, 4 * q);
      float* d = lx + v * XS + 4 * q; d[0] = val.x; d[1] = val.y; d[2] = val.z; d[3] = val.w;
    }
    for (int i = tid; i < Cout * PROJ_TV; i += 256) {
      const int co = i / PROJ_TV, v = i % PROJ_TV;
      lz[co * PROJ_TV + v] = (v0 + v < V) ? dz[((size_t)n * Cout + co) * V + v0 + v] : 0.f;
    }
    __syncthreads();
    // partial dw / dbias: entry e -> (co, ci) or bias co
#pragma unroll
    for (int k = 0; k < (int)(sizeof(part) / sizeof(float)); ++k) {
      const int e = tid + 256 * k;
      if (e < npairs) {
        float s = 0.f;
        if (e < Cout * Cin) {
          const int co = e / Cin, ci = e % Cin;
          for (int v = 0; v < PROJ_TV; ++v) s += lz[co * PROJ_TV + v] * lx[v * XS + ci];
        } else {
          const int co = e - Cout * Cin;
          for (int v = 0; v < PROJ_TV; ++v) s += lz[co * PROJ_TV + v];
        }
        part[k] += s;
      }
    }
    // dx for this tile: thread (v = i / Q, quad) computes 4 channels
    if (dx) {
      for (int i = tid; i < PROJ_TV * Q; i += 256) {
        const int v = i / Q, q = i % Q;
        if (v0 + v >= V) continue;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int co = 0; co < Cout; ++co) {
          const float g = lz[co * PROJ_TV + v];
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] += g * lw[co * Cin + 4 * q + e];
        }
        st4(dx + ((size_t)n * V + v0 + v) * dxld + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
  }
#pragma unroll
  for (int k = 0; k < (int)(sizeof(part) / sizeof(float)); ++k) {
    const int e = tid + 256 * k;
    if (e < npairs) ws[(size_t)blockIdx.x * npairs + e] = part[k];
  }
}

// 8 output elements per workgroup x 32 lanes striding the per-block partials; lanes combined through LDS in a fixed order
__global__ __launch_bounds__(256) void proj_reduce_kernel(const float* ws, int nblocks, int npairs, int ndw, float* dw, float* dbias) {
  __shared__ double part[8][32];
  const int l = threadIdx.x & 31, k = threadIdx.x >> 5;
  const int e = blockIdx.x * 8 + k;
  double s = 0.0;
  if (e < npairs)
    for (int b = l; b < nblocks; b += 32) s += (double)ws[(size_t)b * npairs + e];
  part[k][l] = s;
  __syncthreads();
  if (l == 0 && e < npairs) {
    double t = 0.0;
    for (int i = 0; i < 32; ++i) t += part[k][i];
    if (e < ndw) dw[e] = (float)t;
    else if (dbias) dbias[e - ndw] = (float)t;
  }
}

static int proj_blocks(const mi355_act* x) {
  const long long V = (long long)x->d * x->h * x->w;
  long long t = (long long)x->n * ((V + PROJ_TV - 1) / PROJ_TV);
  if (t > 1024) t = 1024;
  if (t < 1) t = 1;
  return (int)t;
}

extern "C" size_t mi355_proj_workspace(const mi355_act* x, int32_t cout) {
  if (!x) return 0;
  return (size_t)proj_blocks(x) * ((size_t)cout * x->c + cout) * sizeof(float);
}

extern "C" int mi355_proj_fwd(const mi355_act* x, const float* in_scale, const float* in_shift, float act_slope, const float* w,
                              const float* bias, float* logits, int32_t cout, void* stream) {
  if (!act_ok(x) || !w || !logits || cout < 1 || cout > PROJ_MAX_COUT || x->c > PROJ_MAX_CIN) return MI355_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MI355_EINVAL;
  if (in_scale && !(act_slope >= 0.f && act_slope <= 1.f)) return MI355_EINVAL;       // act(u) = max(u, slope * u)
  const long long V = (long long)x->d * x->h * x->w;
  const size_t lds = ((size_t)PROJ_TV * (x->c + 1) + (size_t)cout * x->c) * sizeof(float);
  if (lds > 64 * 1024) return MI355_EUNSUPPORTED;
  int grid = proj_blocks(x) * 8; long long t = (long long)x->n * ((V + PROJ_TV - 1) / PROJ_TV); if (grid > t) grid = (int)t;
  ACT_TYPED(x->dtype, T, LAUNCH(proj_fwd_kernel<T>, dim3(grid), dim3(256), lds, stream, (const T*)x->p, x->ld, in_scale, in_shift, act_slope, w, bias, logits,
                                x->n, V, x->c, cout));
  return LAUNCH_CHECK();
}

extern "C" int mi355_proj_bwd(const mi355_act* x, const float* in_scale, const float* in_shift, float act_slope, const float* w,
                              const float* dlogits, const mi355_act* dx, float* dw, float* dbias, int32_t cout,
                              void* ws, size_t ws_bytes, void* stream) {
  if (!act_ok(x) || !w || !dlogits || !dw || !ws || cout < 1 || cout > PROJ_MAX_COUT || x->c > PROJ_MAX_CIN) return MI355_EINVAL;
  if ((in_scale == nullptr) != (in_shift == nullptr)) return MI355_EINVAL;
  if (in_scale && !(act_slope >= 0.f && act_slope <= 1.f)) return MI355_EINVAL;
  if (dx && (!act_ok(dx) || !same_shape(x, dx))) return MI355_EINVAL;
  if (dx && dx->dtype != x->dtype) return MI355_EUNSUPPORTED;
  if (ws_bytes < mi355_proj_workspace(x, cout)) return MI355_EWORKSPACE;
  const long long V = (long long)x->d * x->h * x->w;
  const size_t lds = ((size_t)PROJ_TV * (x->c + 1) + (size_t)PROJ_MAX_COUT * PROJ_TV + (size_t)cout * x->c) * sizeof(float);
  if (lds > 64 * 1024) return MI355_EUNSUPPORTED;
  const int nb = proj_blocks(x);
  const int npairs = cout * x->c + cout;
  ACT_TYPED(x->dtype, T, LAUNCH(proj_bwd_kernel<T>, dim3(nb), dim3(256), lds, stream, (const T*)x->p, x->ld, in_scale, in_shift, act_slope, w, dlogits,
                                dx ? (T*)dx->p : (T*)nullptr, dx ? dx->ld : 0, (float*)ws, x->n, V, x->c, cout));
  int rc = LAUNCH_CHECK(); if (rc) return rc;
  LAUNCH(proj_reduce_kernel, dim3(ceil_div(npairs, 8)), dim3(256), 0, stream, (const float*)ws, nb, npairs, cout * x->c, dw, dbias);
  return LAUNCH_CHECK();
}

// ---- sliding-window accumulation (MONAI SlidingWindowInferer; script_utils.py:290-293, training_utils.py:106-107) ----
__global__ void sw_accumulate_kernel(const float* pred, const float* w, float* out, float* cnt, int C, int rd, int rh, int rw,
                                     int D, int H, int W, int z0, int y0, int x0) {
  const long long rv = (long long)rd * rh * rw;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < rv; idx += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(idx % rw), y = (int)((idx / rw) % rh), z = (int)(idx / ((long long)rw * rh));
    const int oz = z0 + z, oy = y0 + y, ox = x0 + x;
    if (oz < 0 || oy < 0 || ox < 0 || oz >= D || oy >= H || ox >= W) continue;
    const float wv = w[idx];
    const size_t o = ((size_t)oz * H + oy) * W + ox;
    cnt[o] += wv;
    for (int c = 0; c < C; ++c) out[(size_t)c * D * H * W + o] += wv * pred[(size_t)c * rv + idx];
  }
}
__global__ void sw_normalize_kernel(float* out, const float* cnt, int C, long long V) {
  const long long total = (long long)C * V;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x)
    out[idx] = out[idx] / cnt[idx % V];
}

// ---- batched form: one gather launch builds the whole window batch, one accumulate launch folds all its predictions ----
// starts: device int32 [nw][4] = (sample, z0, y0, x0) of every window of the batch.
__global__ void sw_gather_kernel(const float* vol, int C, int D, int H, int W, const int* starts, int nw, int rd, int rh, int rw, float* win) {
  const long long rv = (long long)rd * rh * rw, per = rv * C, total = per * nw;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int wi = (int)(idx / per); long long r = idx - (long long)wi * per;
    const int c = (int)(r / rv); r -= (long long)c * rv;
    const int x = (int)(r % rw), y = (int)((r / rw) % rh), z = (int)(r / ((long long)rw * rh));
    const int* st = starts + 4 * wi;
    win[idx] = vol[((((size_t)st[0] * C + c) * D + st[1] + z) * H + st[2] + y) * W + st[3] + x];
  }
}
// One thread per OUTPUT voxel walks the batch's windows in order: overlapping windows of one batch are folded race-free and in a fixed
// order (w = 0 .. nw-1), so the result is bitwise the one-launch-per-window form's. Only the BOUNDING BOX of the batch's windows (and the
// samples they belong to) is walked -- every thread derives it from the nw window records -- so a batch costs O(box voxels x nw), not
// O(volume x nw): on a 512^3 volume with 128^3 windows two windows per batch touch 1/32 of the voxels.
__global__ void sw_accumulate_batch_kernel(const float* pred, const float* w, float* out, float* cnt, int N, int C, int rd, int rh, int rw,
                                           int D, int H, int W, const int* starts, int nw) {
  int n0 = N - 1, n1 = 0, z0 = D, z1 = 0, y0 = H, y1 = 0, x0 = W, x1 = 0;
  for (int wi = 0; wi < nw; ++wi) {
    const int* st = starts + 4 * wi;
    n0 = st[0] < n0 ? st[0] : n0; n1 = st[0] > n1 ? st[0] : n1;
    z0 = st[1] < z0 ? st[1] : z0; z1 = st[1] + rd > z1 ? st[1] + rd : z1;
    y0 = st[2] < y0 ? st[2] : y0; y1 = st[2] + rh > y1 ? st[2] + rh : y1;
    x0 = st[3] < x0 ? st[3] : x0; x1 = st[3] + rw > x1 ? st[3] + rw : x1;
  }
  n0 = n0 < 0 ? 0 : n0; n1 = n1 > N - 1 ? N - 1 : n1; z0 = z0 < 0 ? 0 : z0; y0 = y0 < 0 ? 0 : y0; x0 = x0 < 0 ? 0 : x0;
  z1 = z1 > D ? D : z1; y1 = y1 > H ? H : y1; x1 = x1 > W ? W : x1;
  if (n1 < n0 || z1 <= z0 || y1 <= y0 || x1 <= x0) return;
  const int bd = z1 - z0, bh = y1 - y0, bw = x1 - x0;
  const long long V = (long long)D * H * W, bv = (long long)bd * bh * bw, total = bv * (n1 - n0 + 1), rv = (long long)rd * rh * rw;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const int n = n0 + (int)(idx / bv); const long long b = idx % bv;
    const int x = x0 + (int)(b % bw), y = y0 + (int)((b / bw) % bh), z = z0 + (int)(b / ((long long)bw * bh));
    const long long v = ((long long)z * H + y) * W + x;
    for (int wi = 0; wi < nw; ++wi) {
      const int* st = starts + 4 * wi;
      const int lz = z - st[1], ly = y - st[2], lx = x - st[3];
      if (st[0] != n || lz < 0 || ly < 0 || lx < 0 || lz >= rd || ly >= rh || lx >= rw) continue;
      const long long l = ((long long)lz * rh + ly) * rw + lx;
      const float wv = w[l];
      cnt[(long long)n * V + v] += wv;
      for (int c = 0; c < C; ++c) out[((size_t)n * C + c) * V + v] += wv * pred[((size_t)wi * C + c) * rv + l];
    }
  }
}

extern "C" int mi355_sw_gather(const float* volume, int32_t n, int32_t c, int32_t D, int32_t H, int32_t W, const int32_t* starts, int32_t nw,
                               int32_t rd, int32_t rh, int32_t rw, float* windows, void* stream) {
  if (!volume || !starts || !windows || n <= 0 || c <= 0 || nw <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || rd > D || rh > H || rw > W) return MI355_EINVAL;
  LAUNCH(sw_gather_kernel, dim3(grid_for((long long)nw * c * rd * rh * rw)), dim3(256), 0, stream, volume, c, D, H, W, starts, nw, rd, rh, rw, windows);
  return LAUNCH_CHECK();
}
extern "C" int mi355_sw_accumulate_batch(const float* pred, const float* importance, float* out, float* count, int32_t n, int32_t c,
                                         int32_t rd, int32_t rh, int32_t rw, int32_t D, int32_t H, int32_t W, const int32_t* starts, int32_t nw,
                                         void* stream) {
  if (!pred || !importance || !out || !count || !starts || n <= 0 || c <= 0 || nw <= 0 || rd <= 0 || rh <= 0 || rw <= 0) return MI355_EINVAL;
  // grid for the largest bounding box the batch can have (nw windows side by side, capped at the volume); threads beyond the box exit
  long long box = (long long)nw * rd * rh * rw;
  if (box > (long long)n * D * H * W) box = (long long)n * D * H * W;
  LAUNCH(sw_accumulate_batch_kernel, dim3(grid_for(box)), dim3(256), 0, stream, pred, importance, out, count, n, c, rd, rh, rw,
         D, H, W, starts, nw);
  return LAUNCH_CHECK();
}

extern "C" int mi355_sw_accumulate(const float* pred, const float* importance, float* out, float* count, int32_t c,
                                   int32_t rd, int32_t rh, int32_t rw, int32_t D, int32_t H, int32_t W,
                                   int32_t z0, int32_t y0, int32_t x0, void* stream) {
  if (!pred || !importance || !out || !count || c <= 0 || rd <= 0 || rh <= 0 || rw <= 0 || D <= 0 || H <= 0 || W <= 0) return MI355_EINVAL;
  LAUNCH(sw_accumulate_kernel, dim3(grid_for((long long)rd * rh * rw)), dim3(256), 0, stream, pred, importance, out, count, c, rd, rh, rw,
         D, H, W, z0, y0, x0);
  return LAUNCH_CHECK();
}
extern "C" int mi355_sw_normalize(float* out, const float* count, int32_t c, int64_t voxels, void* stream) {
  if (!out || !count || c <= 0 || voxels <= 0) return MI355_EINVAL;
  LAUNCH(sw_normalize_kernel, dim3(grid_for((long long)c * voxels)), dim3(256), 0, stream, out, count, c, (long long)voxels);
  return LAUNCH_CHECK();
}
