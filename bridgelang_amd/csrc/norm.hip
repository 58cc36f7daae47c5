// norm.hip — LayerNorm (timm ViT blocks) and RMSNorm (HF Llama) for bf16 rows, fp32 statistics.
// HBM-bound: one wave per row, 16-byte vector loads, the row is held in registers between the statistics pass and
// the normalise pass (each element is read from HBM exactly once), wave-shuffle butterflies for the reductions.
#include "bl_common.h"

namespace bl_norm_impl {

// NCH = 16-byte chunks (8 bf16) per lane; a wave covers up to 64*NCH*8 columns.
template <int NCH, bool RMS>
__global__ __launch_bounds__(256) void norm_rows_kernel(const uint16_t* __restrict__ x, long ldx,
                                                        const uint16_t* __restrict__ w,
                                                        const uint16_t* __restrict__ b, uint16_t* __restrict__ y,
                                                        long ldy, int rows, int dim, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;   // whole wave exits together
  const int nchunk = dim >> 3;
  const uint16_t* xr = x + (long)row * ldx;
  float v[NCH][8];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = c * 64 + lane;
    u32x4_t q = {0u, 0u, 0u, 0u};
    if (ch < nchunk) q = *(const u32x4_t*)(xr + ch * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[c][2 * i] = bflo(q[i]); v[c][2 * i + 1] = bfhi(q[i]); }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += RMS ? v[c][i] * v[c][i] : v[c][i];
  }
  s = wave_sum(s);
  const float inv_n = 1.0f / (float)dim;
  float mean = 0.f, rstd;
  if constexpr (RMS) {
    rstd = 1.0f / sqrtf(s * inv_n + eps);
  } else {
    mean = s * inv_n;
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const bool in = (c * 64 + lane) < nchunk;
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = v[c][i] - mean; ss += in ? d * d : 0.f; }
    }
    ss = wave_sum(ss);
    rstd = 1.0f / sqrtf(ss * inv_n + eps);
  }
  uint16_t* yr = y + (long)row * ldy;
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = c * 64 + lane;
    if (ch >= nchunk) continue;
    const u32x4_t wq = *(const u32x4_t*)(w + ch * 8);
    float o[8];
    if constexpr (RMS) {
      // HF LlamaRMSNorm: weight * (x * rsqrt(var + eps)).to(bf16)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[2 * i] = bflo(wq[i]) * rbf(v[c][2 * i] * rstd);
        o[2 * i + 1] = bfhi(wq[i]) * rbf(v[c][2 * i + 1] * rstd);
      }
    } else {
      const u32x4_t bq = *(const u32x4_t*)(b + ch * 8);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[2 * i] = (v[c][2 * i] - mean) * rstd * bflo(wq[i]) + bflo(bq[i]);
        o[2 * i + 1] = (v[c][2 * i + 1] - mean) * rstd * bfhi(wq[i]) + bfhi(bq[i]);
      }
    }
    u32x4_t oq;
#pragma unroll
    for (int i = 0; i < 4; ++i) oq[i] = pack2bf(o[2 * i], o[2 * i + 1]);
    *(u32x4_t*)(yr + ch * 8) = oq;
  }
}

template <bool RMS>
int launch_norm(const uint16_t* x, long ldx, const uint16_t* w, const uint16_t* b, uint16_t* y, long ldy, int rows,
                int dim, float eps, hipStream_t s) {
  if (!x || !w || !y || (!RMS && !b)) return BL_E_ARG;
  if (rows <= 0 || dim <= 0 || (dim % 8) || dim > 64 * 8 * 10) return BL_E_SHAPE;
  if ((ldx % 8) || (ldy % 8) || !bl_aligned16(x) || !bl_aligned16(y) || !bl_aligned16(w) || (b && !bl_aligned16(b)))
    return BL_E_ALIGN;
  const int nch = (dim / 8 + 63) / 64;
  const dim3 grid((rows + 3) / 4), block(256);
#define BL_NORM_CASE(N)                                                                                     \
  case N: hipLaunchKernelGGL((norm_rows_kernel<N, RMS>), grid, block, 0, s, x, ldx, w, b, y, ldy, rows, dim, eps); break;
  switch (nch) {
    BL_NORM_CASE(1) BL_NORM_CASE(2) BL_NORM_CASE(3) BL_NORM_CASE(4) BL_NORM_CASE(5)
    BL_NORM_CASE(6) BL_NORM_CASE(7) BL_NORM_CASE(8) BL_NORM_CASE(9) BL_NORM_CASE(10)
    default: return BL_E_SHAPE;
  }
#undef BL_NORM_CASE
  BL_CHECK_LAUNCH();
  return BL_OK;
}

}  // namespace bl_norm_impl
using namespace bl_norm_impl;

extern "C" int bl_layernorm_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* b, bl_bf16* y,
                                 int64_t ldy, int32_t rows, int32_t dim, float eps, void* stream) {
  return launch_norm<false>(x, ldx, w, b, y, ldy, rows, dim, eps, (hipStream_t)stream);
}

extern "C" int bl_rmsnorm_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, bl_bf16* y, int64_t ldy,
                               int32_t rows, int32_t dim, float eps, void* stream) {
  return launch_norm<true>(x, ldx, w, nullptr, y, ldy, rows, dim, eps, (hipStream_t)stream);
}
