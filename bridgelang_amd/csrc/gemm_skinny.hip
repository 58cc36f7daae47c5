// gemm_skinny.hip — C[M<=16, N] = epi(x[M,K] · W[N,K]^T): the decode-step and last-row lm_head GEMMs.
//
// HBM-bound (every weight byte is read exactly once per call, 13.2 GB per 7B decode step), so the design is a
// weight-streaming one, not a tile GEMM:
//   * workgroup = 8 waves; the waves split K (wave w owns k ∈ [w·K/8, (w+1)·K/8)); the workgroup walks 16-row weight
//     tiles n-tile = blockIdx.x, + gridDim.x, …
//   * the activation slice x[0:16, k-slice] lives in REGISTERS as MFMA "B" fragments for the whole kernel (loaded once
//     per wave), so the only stream is the weights: straight HBM→VGPR 16-byte non-temporal loads (no LDS round trip —
//     guide §5 "GEMV / M ≤ 16" row), 8 loads in flight per wave, consumed by v_mfma_f32_16x16x32_bf16.
//   * the 8 partial 16×16 fp32 tiles are summed through LDS (one barrier per n-tile, double-buffered), and a rotating
//     wave applies the same fused epilogues as the tiled GEMM.
#include "gemm_common.h"

namespace bl_gemm_skinny_impl {
using namespace blgemm;

constexpr int NW = 8;   // waves per workgroup = K split factor

template <int KS, int EPI>   // KS = MFMA k-steps (32 wide) per wave: K == NW * KS * 32
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p, int n_tiles) {
  __shared__ f32x4_t red[2][NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const long kbase = (long)wave * (KS * 32) + lg * 8;

  // activation fragments ("B" operand): lane holds x[m = l15][kbase + 32*s .. +7]
  bf16x8_t xf[KS];
  {
    const uint16_t* xp = p.A + (long)l15 * p.lda + kbase;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4_t t = {0u, 0u, 0u, 0u};
      if (l15 < p.M) t = *(const u32x4_t*)(xp + s * 32);
      xf[s] = __builtin_bit_cast(bf16x8_t, t);
    }
  }

  int it = 0;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, ++it) {
    const u32x4_t* wp = (const u32x4_t*)(p.W + (long)(tile * 16 + l15) * p.ldw + kbase);   // 16-byte units
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < KS; s0 += 8) {
      u32x4_t w[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s0 + j < KS) w[j] = __builtin_nontemporal_load(wp + (s0 + j) * 4);   // +32 elements = 4 × 16 B
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s0 + j < KS)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[j]), xf[s0 + j], acc, 0, 0, 0);
    }
    red[it & 1][wave][lane] = acc;
    __syncthreads();
    if (wave == (it % NW)) {
      f32x4_t sum = red[it & 1][0][lane];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) sum += red[it & 1][w2][lane];
      epilogue_store4<EPI>(p, l15, tile * 16 + lg * 4, sum);
    }
  }
}

template <int KS, int EPI>
int launch_ks(const GemmArgs& a, hipStream_t s) {
  const int n_tiles = a.N / 16;
  const int per_cu = (KS <= 24) ? 2 : 1;          // x fragments cost 4·KS VGPRs per lane
  const int grid = n_tiles < 256 * per_cu ? n_tiles : 256 * per_cu;
  hipLaunchKernelGGL((gemm_skinny_kernel<KS, EPI>), dim3(grid), dim3(NW * 64), 0, s, a, n_tiles);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

template <int EPI>
int launch_skinny(const GemmArgs& a, hipStream_t s) {
  switch (a.K) {
    case 4096: return launch_ks<16, EPI>(a, s);    // Llama-2-7B hidden
    case 11008: return launch_ks<43, EPI>(a, s);   // Llama-2-7B MLP
    case 5120: return launch_ks<20, EPI>(a, s);    // Llama-2-13B hidden
    case 13824: return launch_ks<54, EPI>(a, s);   // Llama-2-13B MLP
    case 512: return launch_ks<2, EPI>(a, s);      // reduced-width test / oracle configs
    case 1024: return launch_ks<4, EPI>(a, s);
    case 1536: return launch_ks<6, EPI>(a, s);
    default: return BL_E_SHAPE;                    // caller falls back to bl_gemm_bf16
  }
}

}  // namespace bl_gemm_skinny_impl
using namespace bl_gemm_skinny_impl;

extern "C" int bl_gemm_skinny_bf16(const bl_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = fill_gemm_args(d, a);
  if (rc != BL_OK) return rc;
  if (d->M > 16) return BL_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case BL_EPI_NONE: return launch_skinny<BL_EPI_NONE>(a, s);
    case BL_EPI_RES: return launch_skinny<BL_EPI_RES>(a, s);
    case BL_EPI_SWIGLU: return launch_skinny<BL_EPI_SWIGLU>(a, s);
    case BL_EPI_F32: return launch_skinny<BL_EPI_F32>(a, s);
    case BL_EPI_F32_BF16R: return launch_skinny<BL_EPI_F32_BF16R>(a, s);
    default: return BL_E_ARG;
  }
}
