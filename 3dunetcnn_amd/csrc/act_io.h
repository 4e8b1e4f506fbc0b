// Typed access to activation tensors (mi355_act.dtype, include/mi355_unet3d.h): fp32, bf16 or IEEE fp16 STORAGE. Arithmetic is fp32 in
// every kernel; a 16-bit tensor is converted on load (bf16: a shift, fp16: v_cvt_f32_f16) and rounded to nearest-even on store
// (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32). The kernels are
// templates over the element type of each tensor they touch and address it in elements, so a typed pointer does the byte arithmetic.
// Reference: unet3d/models/pytorch/segmentation/unet.py:53-58 (AutocastUNet: conv outputs are 16-bit tensors under torch autocast).
#pragma once
#include "gfx950_dialect.h"
#include <type_traits>
#include "../../include/mi355_unet3d.h"

typedef unsigned short bf16_t;      // storage only: never an arithmetic type
struct f16_t { unsigned short v; }; // IEEE fp16 storage (the reference's amp tensors, train/train.py:33-37): a type of its own for overloading
// the two 16-bit storage types through one set of names: value of the low / high half of a packed dword, two floats packed
template <typename T> struct is_lp16 { static constexpr bool value = false; };
template <> struct is_lp16<bf16_t> { static constexpr bool value = true; };
template <> struct is_lp16<f16_t> { static constexpr bool value = true; };
template <typename T> __device__ __forceinline__ float lp_lo(unsigned p) { if constexpr (std::is_same<T, f16_t>::value) return f16lo_to_f32(p); else return bf16lo_to_f32(p); }
template <typename T> __device__ __forceinline__ float lp_hi(unsigned p) { if constexpr (std::is_same<T, f16_t>::value) return f16hi_to_f32(p); else return bf16hi_to_f32(p); }
template <typename T> __device__ __forceinline__ unsigned lp_pack2(float lo, float hi) { if constexpr (std::is_same<T, f16_t>::value) return pack_f16x2(lo, hi); else return pack_bf16x2(lo, hi); }
// 16-bit storage whose element type IS the operand type of the matrix path (F16: v_mfma_f32_32x32x16_f16): a plain input goes from
// global memory to LDS as loaded
template <typename T, bool F16> struct lp_storage_is_operand { static constexpr bool value = F16 ? std::is_same<T, f16_t>::value : std::is_same<T, bf16_t>::value; };

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16lo_to_f32(v.x), bf16hi_to_f32(v.x), bf16lo_to_f32(v.y), bf16hi_to_f32(v.y));
}
__device__ __forceinline__ float4 ld4(const f16_t* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(f16lo_to_f32(v.x), f16hi_to_f32(v.x), f16lo_to_f32(v.y), f16hi_to_f32(v.y));
}
__device__ __forceinline__ void st4(float* p, const float4& v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(f16_t* p, const float4& v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_f16x2(v.x, v.y), pack_f16x2(v.z, v.w)); }
__device__ __forceinline__ void st4(bf16_t* p, const float4& v) { *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w)); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ float ld1(const f16_t* p) { return f16lo_to_f32(p->v); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(f16_t* p, float v) { p->v = (unsigned short)(pack_f16x2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ void st1(bf16_t* p, float v) { *p = (bf16_t)(pack_bf16x2(v, 0.f) & 0xffffu); }
// the value a tensor of this type returns for `v` once stored: what statistics of "the tensor as stored" are taken over
__device__ __forceinline__ float as_stored(const float*, float v) { return v; }
__device__ __forceinline__ float as_stored(const bf16_t*, float v) { return bf16lo_to_f32(pack_bf16x2(v, 0.f)); }
__device__ __forceinline__ float as_stored(const f16_t*, float v) { return f16lo_to_f32(pack_f16x2(v, 0.f)); }
// 8 consecutive bf16 (16 bytes) / 8 consecutive floats as 8 floats
__device__ __forceinline__ void ld8(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&o)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  o[0] = bf16lo_to_f32(v.x); o[1] = bf16hi_to_f32(v.x); o[2] = bf16lo_to_f32(v.y); o[3] = bf16hi_to_f32(v.y);
  o[4] = bf16lo_to_f32(v.z); o[5] = bf16hi_to_f32(v.z); o[6] = bf16lo_to_f32(v.w); o[7] = bf16hi_to_f32(v.w);
}
__device__ __forceinline__ void ld8(const f16_t* p, float (&o)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  o[0] = f16lo_to_f32(v.x); o[1] = f16hi_to_f32(v.x); o[2] = f16lo_to_f32(v.y); o[3] = f16hi_to_f32(v.y);
  o[4] = f16lo_to_f32(v.z); o[5] = f16hi_to_f32(v.z); o[6] = f16lo_to_f32(v.w); o[7] = f16hi_to_f32(v.w);
}

// VW consecutive channels (VW = 4: the float4 / 8-byte granularity of every view; VW = 8: one 16-byte access per lane for bf16 tensors whose
// channel count, leading dimension and base address allow it -- an 8-byte access per lane leaves the streaming kernels at ~70 % of what
// 16 bytes per lane reach)
template <int VW, typename T> __device__ __forceinline__ void ldv(const T* p, float (&o)[VW]) {
  if constexpr (VW == 8) ld8(p, o);
  else { const float4 v = ld4(p); o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
}
template <int VW> __device__ __forceinline__ void stv(float* p, const float (&v)[VW]) {
  st4(p, make_float4(v[0], v[1], v[2], v[3]));
  if constexpr (VW == 8) st4(p + 4, make_float4(v[4], v[5], v[6], v[7]));
}
template <int VW> __device__ __forceinline__ void stv(bf16_t* p, const float (&v)[VW]) {
  if constexpr (VW == 8)
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  else st4(p, make_float4(v[0], v[1], v[2], v[3]));
}
template <int VW> __device__ __forceinline__ void stv(f16_t* p, const float (&v)[VW]) {
  if constexpr (VW == 8)
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_f16x2(v[0], v[1]), pack_f16x2(v[2], v[3]), pack_f16x2(v[4], v[5]), pack_f16x2(v[6], v[7]));
  else st4(p, make_float4(v[0], v[1], v[2], v[3]));
}
static inline bool act_is_lp16(int dtype) { return dtype == MI355_ACT_BF16 || dtype == MI355_ACT_F16; }
// can every one of these 16-bit views be walked 8 channels at a time?
static inline bool act_vw8(const mi355_act* t) {
  return act_is_lp16(t->dtype) && t->c % 8 == 0 && t->ld % 8 == 0 && !((uintptr_t)t->p & 15);
}

static inline size_t act_elem_bytes(int dtype) { return act_is_lp16(dtype) ? 2 : 4; }
static inline bool act_dtype_ok(const mi355_act* t) { return t->dtype == MI355_ACT_F32 || act_is_lp16(t->dtype); }
// base-address alignment of a view walked 4 elements at a time (float4 / 8 bytes of 16-bit elements)
static inline uintptr_t act_align_mask(int dtype) { return act_is_lp16(dtype) ? 7 : 15; }
// the 16-bit operand precision a 16-bit storage type goes with (bf16 tensors with bf16 operands, fp16 tensors with fp16 operands)
static inline bool act_matches_precision(int dtype, int precision) {
  return dtype == MI355_ACT_F32 || (dtype == MI355_ACT_BF16 && precision == MI355_PREC_BF16) || (dtype == MI355_ACT_F16 && precision == MI355_PREC_F16);
}
// a view the streaming kernels accept: 4-element (float4 / 8-byte) granularity
static inline int act_view_ok(const mi355_act* t) {
  return t && t->p && act_dtype_ok(t) && t->c > 0 && t->c % 4 == 0 && t->ld % 4 == 0 && t->ld >= t->c &&
         !((uintptr_t)t->p & act_align_mask(t->dtype));
}

// Host-side dispatch: runs the statement list with T = the element type of `dtype`
#define ACT_TYPED(dtype, T, ...)                                              \
  do {                                                                        \
    if ((dtype) == MI355_ACT_BF16) { typedef bf16_t T; __VA_ARGS__; }         \
    else if ((dtype) == MI355_ACT_F16) { typedef f16_t T; __VA_ARGS__; }      \
    else { typedef float T; __VA_ARGS__; }                                    \
  } while (0)
// ... of a 16-bit `dtype` only (the 8-channel-per-lane forms)
#define ACT_TYPED_LP16(dtype, T, ...)                                         \
  do {                                                                        \
    if ((dtype) == MI355_ACT_F16) { typedef f16_t T; __VA_ARGS__; }           \
    else { typedef bf16_t T; __VA_ARGS__; }                                   \
  } while (0)
