// common.h — shared device/host helpers for libmgld_hip (gfx950 only; no CUDA/dual-platform paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mgld_hip.h"

typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MGLD_WAVE 64

void mgld_set_error(const char* what, hipError_t e);
int mgld_check_launch(const char* what);

#define MGLD_REQUIRE(cond, msg)                         \
  do {                                                  \
    if (!(cond)) {                                      \
      mgld_set_error(msg, hipErrorInvalidValue);        \
      return MGLD_E_ARG;                                \
    }                                                   \
  } while (0)

// x * sigmoid(x) with the hardware reciprocal (1 ulp) instead of an IEEE division (v_div_scale / v_div_fmas / v_div_fixup: ~10 VALU
// instructions per element in kernels that otherwise do 3-4)
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// erf(x) by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7: below fp32 round-off of the surrounding arithmetic, three orders below the
// fp16 rounding of the stored result): 1 rcp + 1 exp + 7 FMA instead of libm's branchy erff (~3x the VALU work, and the GEGLU
// epilogue of the feed-forward GEMMs is VALU-bound: 1777 VALU instructions per wave for 80 MFMAs before this change).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float r = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

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
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// the residual stream as a pair of fp16 planes (MgldIGemm.Rlo / Clo, mgld_*_lo): x = hi + 2^-11 lo, hi = fp16(x), lo = fp16((x - hi) 2^11)
#define MGLD_LO_SCALE 4.8828125e-4f
#define MGLD_LO_INV 2048.f
__device__ __forceinline__ f16 lo_plane(const float x, const f16 hi) { return (f16)((x - (float)hi) * MGLD_LO_INV); }

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
