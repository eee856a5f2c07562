// seedx-mi355x — common device/host helpers for the gfx950 (CDNA4) kernels.
// Everything here is written for wave64 / MFMA / LDS-DMA on MI355X only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/seedx_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define SX_WAVE 64
#define SX_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- host-side error plumbing -------------------------------------------------------------
void sx_set_error(const char* fmt, ...);
#define SX_FAIL(...)            \
  do {                          \
    sx_set_error(__VA_ARGS__);  \
    return SX_ERR_INVALID;      \
  } while (0)
#define SX_CHECK(cond, ...) \
  do {                      \
    if (!(cond)) SX_FAIL(__VA_ARGS__); \
  } while (0)
#define SX_HIP_LAUNCH_CHECK()                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      sx_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
      return SX_ERR_HIP;                                                             \
    }                                                                                \
  } while (0)

// ---- 16-bit float element traits ----------------------------------------------------------
struct BF16 {
  typedef __bf16 elem;
  typedef bf16x8_t vec8;
  typedef bf16x4_t vec4;
  static __device__ __forceinline__ float to_f32(unsigned short u) { return __uint_as_float(((unsigned)u) << 16); }
  static __device__ __forceinline__ unsigned short from_f32(float f) {
    __bf16 h = (__bf16)f;  // hardware RNE convert on gfx950 (v_cvt_pk_bf16_f32)
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
  }
  static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  // acc + lo + hi of a packed pair in ONE VALU op (v_dot2c_f32_bf16 against {1, 1})
  static __device__ __forceinline__ float pair_sum(unsigned w, float acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __bf16 e2_t __attribute__((ext_vector_type(2)));
    const e2_t ones = {(__bf16)1.0f, (__bf16)1.0f};
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(e2_t, w), ones, acc, false);
#else
    return acc;
#endif
  }
};
struct F16 {
  typedef _Float16 elem;
  typedef f16x8_t vec8;
  typedef f16x4_t vec4;
  static __device__ __forceinline__ float to_f32(unsigned short u) {
    _Float16 h;
    __builtin_memcpy(&h, &u, 2);
    return (float)h;
  }
  static __device__ __forceinline__ unsigned short from_f32(float f) {
    _Float16 h = (_Float16)f;  // RNE, saturates to inf like torch .half()
    unsigned short u;
    __builtin_memcpy(&u, &h, 2);
    return u;
  }
  static __device__ __forceinline__ f32x4_t mfma16(vec8 a, vec8 b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ f32x16_t mfma32(vec8 a, vec8 b, f32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ float pair_sum(unsigned w, float acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef _Float16 e2_t __attribute__((ext_vector_type(2)));
    const e2_t ones = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(e2_t, w), ones, acc, false);
#else
    return acc;
#endif
  }
};

// max(a, b, c) in one VALU op. fmaxf() costs an extra canonicalising v_max per operand under IEEE semantics.
__device__ __forceinline__ float max3f(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
#else
  return a;
#endif
}

// pack two floats -> two 16-bit values in one dword (vector convert → v_cvt_pk_bf16_f32 / v_cvt_f16_f32 pairs)
template <typename TT>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef typename TT::elem e2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {lo, hi};
  const e2_t r = __builtin_convertvector(v, e2_t);
  unsigned u;
  __builtin_memcpy(&u, &r, 4);
  return u;
}

// ---- wave / block reductions --------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// activations (fp32 math, exact-erf GELU as torch nn.GELU())
// exact-erf GELU (torch nn.GELU()). erf by Abramowitz–Stegun 7.1.26 (|abs err| <= 1.5e-7, far below the 16-bit operand
// rounding): 1 v_rcp + 1 v_exp + ~10 FMAs per element instead of the ~30-instruction libm erff in the GEGLU epilogues
__device__ __forceinline__ float gelu_erf(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float pl = fmaf(1.061405429f, t, -1.453152027f);
  pl = fmaf(pl, t, 1.421413741f);
  pl = fmaf(pl, t, -0.284496736f);
  pl = fmaf(pl, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);
  const float erf_abs = fmaf(-pl * t, e, 1.0f);          // erf(|x|/sqrt2)
  const float h = 0.5f * x;
  return fmaf(copysignf(erf_abs, x), h, h);               // 0.5 x (1 + erf(x/sqrt2))
}
// two GELUs at once on <2 x float>: the polynomial / product part becomes v_pk_fma_f32 / v_pk_mul_f32 (2 lanes-worth per
// issue); only the reciprocal and the exponential stay scalar. Same formula and constants as gelu_erf().
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
  const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
  const f32x2_t z = ax * 0.70710678118654752440f;
  const f32x2_t d = z * 0.3275911f + 1.0f;
  const f32x2_t t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_t pl = t * 1.061405429f + (-1.453152027f);
  pl = pl * t + 1.421413741f;
  pl = pl * t + (-0.284496736f);
  pl = pl * t + 0.254829592f;
  const f32x2_t a = z * z * (-1.4426950408889634f);
  const f32x2_t e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
  const f32x2_t erf_abs = 1.0f - pl * t * e;
  const f32x2_t h = x * 0.5f;
  const f32x2_t sg = {copysignf(erf_abs[0], x[0]), copysignf(erf_abs[1], x[1])};
  return sg * h + h;
}
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float apply_act(float x, int act) {
  if (act == SX_ACT_GELU) return gelu_erf(x);
  if (act == SX_ACT_SILU) return silu_f(x);
  return x;
}

// bijective XCD-aware remap of a 1-D grid (block b is observed on XCD b % 8): give each XCD a
// contiguous chunk of the logical tile order so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, idx = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}
