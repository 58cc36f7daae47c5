// bl_common.h — device-side helpers shared by the gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/bridgelang_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // 16x16 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte vector
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;  // 8-byte vector

#define BL_WAVE 64
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// ---- bf16 <-> fp32 ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
// Plain cast lowers to v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving) on gfx950.
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }
__device__ __forceinline__ float rbf(float f) { return bf2f(f2bf(f)); }  // round through bf16
// ONE v_cvt_pk_bf16_f32 for the pair (the two-scalar-casts form cost two converts + shift + or; same rounding)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float bflo(uint32_t w) { return __builtin_bit_cast(float, w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __builtin_bit_cast(float, w & 0xffff0000u); }

// ---- wave reductions (xor butterfly over 64 lanes) ------------------------------------------------------------
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

// ---- activation functions, written so the CPU oracle can restate them op for op --------------------------------
// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7, below fp32 resolution of values near 1): one reciprocal, one
// exp and five FMAs instead of ocml's branchy erff — the exact-GELU epilogue of the ViT fc1 GEMMs was costing as much as
// their whole K = 1024 MFMA loop. Same evaluation structure as torch's `0.5 * x * (1 + erf(x / sqrt(2)))`.
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __frcp_rn(fmaf(0.3275911f, ax, 1.0f));
  const float poly = t * fmaf(t, fmaf(t, fmaf(t, fmaf(t, 1.061405429f, -1.453152027f), 1.421413741f), -0.284496736f), 0.254829592f);
  const float y = 1.0f - poly * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }
// x·sigmoid(x) with the hardware exp2 / reciprocal (≈ 3 fp32 ulp): the SwiGLU epilogue evaluates it 1.6e9 times per step
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// Backward of act = bf16(silu(g)) * u for one (gate, up) pair given d = dL/d act — ONE definition for
// bl_swiglu_backward_bf16 and the BL_EPI_SWIGLU_BWD GEMM epilogue (bit-identical paths).
__device__ __forceinline__ void swiglu_bwd_pair(float g, float u, float d, float& dg, float& du) {
  const float sg = 1.0f / (1.0f + expf(-g));
  dg = d * u * (sg * (1.0f + g * (1.0f - sg)));
  du = d * rbf(g * sg);
}
// d/dx of the exact-erf GELU (bl_gelu_backward_bf16 and the BL_EPI_GELU_BWD epilogue)
__device__ __forceinline__ float gelu_erf_grad(float x) {
  return 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f)) + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}

// integer hash shared by the synthetic-weight generator (glue.hip) and the dropout masks (train.hip); oracle/synth.py::_mix32
__device__ __forceinline__ uint32_t bl_mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// ---- launch helpers -------------------------------------------------------------------------------------------
#define BL_CHECK_LAUNCH()                                   \
  do {                                                      \
    hipError_t e__ = hipGetLastError();                     \
    if (e__ != hipSuccess) return BL_E_LAUNCH;              \
  } while (0)

static inline bool bl_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }
