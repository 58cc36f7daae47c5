// gemm_skinny.hip — C[M<=16, N] = epi(x[M,K] · W[N,K]^T): the decode-step and last-row lm_head GEMMs.
//
// HBM-bound (every weight byte is read exactly once per call, 13.2 GB per 7B decode step), so the design is a
// weight-streaming one, not a tile GEMM:
//   * weights are stored fragment-major (bl_pack_weight_bf16): the 64 lanes × 16 B of one MFMA operand are one
//     contiguous KiB, a wave's K-slice of a 16-row tile is one contiguous run, a workgroup's tile one contiguous
//     16·K·2-byte region → whole-line requests and DRAM-page-friendly streaming (measured +20-25 % over reading the
//     same fragments out of a row-major matrix: 4.27 → 5.37 TB/s on the 180 MB gate/up matrix).
//   * workgroup = 8 waves; the waves split K (wave w owns k-steps [w·KS, (w+1)·KS)); the workgroup walks 16-row weight
//     tiles n-tile = blockIdx.x, + gridDim.x, …
//   * the activation slice x[0:16, k-slice] lives in REGISTERS as MFMA "B" fragments for the whole kernel, so the only
//     stream is the weights: straight HBM→VGPR 16-byte non-temporal loads (no LDS round trip — guide §5 "GEMV / M ≤ 16"
//     row) in two register buffers of GS loads; `sched_barrier` pins "issue group g+1, then the MFMAs of group g" (left
//     alone, hipcc collapses the batch to 2 loads in flight), and the prefetch runs across tile boundaries.
//   * the 8 partial 16×16 fp32 tiles are summed through LDS behind a RAW s_barrier (a __syncthreads() would emit
//     vmcnt(0) and drain the prefetch), double-buffered, and a rotating wave applies the fused epilogue.
#include "gemm_common.h"

namespace bl_gemm_skinny_impl {
using namespace blgemm;

constexpr int NW = 8;   // waves per workgroup = K split factor

// KS = MFMA k-steps (32 wide) per wave: K == NW * KS * 32.  GS = k-steps per load group; G = ceil(KS/GS) must be even
// so the two register buffers alternate with compile-time indices.
template <int KS, int GS, int EPI, bool NORM>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmArgs p, int n_tiles) {
  constexpr int G = (KS + GS - 1) / GS;
  static_assert(G % 2 == 0, "need an even number of load groups");
  __shared__ f32x4_t red[2][NW][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const long kbase = (long)wave * (KS * 32) + lg * 8;

  // activation fragments ("B" operand): lane holds x[m = l15][kbase + 32*s .. +7]; rows >= M read row M-1 and are
  // zeroed with a select (no per-load branches)
  bf16x8_t xf[KS];
  {
    const bool live = l15 < p.M;
    const uint16_t* xp = p.A + (long)(live ? l15 : p.M - 1) * p.lda + kbase;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      u32x4_t t = *(const u32x4_t*)(xp + s * 32);
      if (!live) t = (u32x4_t){0u, 0u, 0u, 0u};
      xf[s] = __builtin_bit_cast(bf16x8_t, t);
    }
  }

  // packed weights, 16-byte units: block (tile, k-step) = 64 units; this lane's unit = lane
  u32x4_t wbuf[2][GS];
  const u32x4_t* wbase = (const u32x4_t*)p.W + (long)wave * KS * 64 + lane;
  const long tile_stride16 = (long)(p.K / 32) * 64;

#define BL_LOAD_GROUP(BUF, TILE, GIDX)                                                                 \
  do {                                                                                                 \
    const u32x4_t* wp__ = wbase + (long)(TILE) * tile_stride16 + (GIDX) * GS * 64;                     \
    _Pragma("unroll") for (int j = 0; j < GS; ++j)                                                     \
      if ((GIDX) * GS + j < KS) wbuf[BUF][j] = __builtin_nontemporal_load(wp__ + j * 64);              \
  } while (0)

  int it = 0;
  int tile = blockIdx.x;
  if (tile < n_tiles) BL_LOAD_GROUP(0, tile, 0);   // first weight group is in flight while the norm prologue runs

  if constexpr (NORM) {
    // fused HF LlamaRMSNorm on the activation rows: sum of squares over the wave's K-slice, across the 4 lane groups
    // that share a row, then across the 8 waves through LDS; the fragments are rewritten in registers with the same
    // two roundings as bl_rmsnorm_bf16.
    __shared__ float nrm[NW][16];
    u32x4_t gw[KS];                       // norm weights for this lane's k positions: issued before the reduction
    {
      const uint16_t* wn = p.norm_w + kbase;
#pragma unroll
      for (int s = 0; s < KS; ++s) gw[s] = *(const u32x4_t*)(wn + s * 32);
    }
    float ss = 0.f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4_t t = __builtin_bit_cast(u32x4_t, xf[s]);
#pragma unroll
      for (int i = 0; i < 4; ++i) ss += bflo(t[i]) * bflo(t[i]) + bfhi(t[i]) * bfhi(t[i]);
    }
    ss += __shfl_xor(ss, 16, 64);
    ss += __shfl_xor(ss, 32, 64);
    if (lg == 0) nrm[wave][l15] = ss;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw barrier: keep the prefetched weight loads in flight
    __builtin_amdgcn_s_barrier();
    float tot = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < NW; ++w2) tot += nrm[w2][l15];
    const float rstd = 1.0f / sqrtf(tot * (1.0f / (float)p.K) + p.norm_eps);
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4_t t = __builtin_bit_cast(u32x4_t, xf[s]);
      const u32x4_t g = gw[s];
      u32x4_t o;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        o[i] = pack2bf(bflo(g[i]) * rbf(bflo(t[i]) * rstd), bfhi(g[i]) * rbf(bfhi(t[i]) * rstd));
      xf[s] = __builtin_bit_cast(bf16x8_t, o);
    }
  }

  for (; tile < n_tiles; tile += gridDim.x, ++it) {
    const int next_tile = tile + gridDim.x;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < G; ++g) {
      // keep the NEXT group's loads in flight while this group's MFMAs run (also across the tile boundary)
      if (g + 1 < G) BL_LOAD_GROUP((g + 1) & 1, tile, g + 1);
      else if (next_tile < n_tiles) BL_LOAD_GROUP(0, next_tile, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < GS; ++j)
        if (g * GS + j < KS)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wbuf[g & 1][j]), xf[g * GS + j],
                                                        acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    red[it & 1][wave][lane] = acc;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave == (it % NW)) {
      f32x4_t sum = red[it & 1][0][lane];
#pragma unroll
      for (int w2 = 1; w2 < NW; ++w2) sum += red[it & 1][w2][lane];
      epilogue_store4<EPI>(p, l15, tile * 16 + lg * 4, sum);
    }
  }
#undef BL_LOAD_GROUP
}

template <int KS, int GS, int EPI>
int launch_ks(const GemmArgs& a, hipStream_t s) {
  const int n_tiles = a.N / 16;
  // One 8-wave workgroup per CU: every variant needs > 128 VGPRs (x fragments 4·KS, two weight buffers, the fused-norm
  // weights: 148–254), so two workgroups never co-reside; a grid of 384 (768 tiles ÷ 2) ran as 1.5 rounds of 256.
  // Balanced tiles per workgroup on 256 slots: 768 tiles → 256 × 3; 1376 → 230 × 6 (not 256 × 5 + 96 stragglers).
  const int tpw = (n_tiles + 255) / 256;
  const int grid = (n_tiles + tpw - 1) / tpw;
  if (a.norm_w) hipLaunchKernelGGL((gemm_skinny_kernel<KS, GS, EPI, true>), dim3(grid), dim3(NW * 64), 0, s, a, n_tiles);
  else hipLaunchKernelGGL((gemm_skinny_kernel<KS, GS, EPI, false>), dim3(grid), dim3(NW * 64), 0, s, a, n_tiles);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

template <int EPI>
int launch_skinny(const GemmArgs& a, hipStream_t s) {
  switch (a.K) {
    case 4096: return launch_ks<16, 8, EPI>(a, s);    // Llama-2-7B hidden
    case 11008: return launch_ks<43, 8, EPI>(a, s);   // Llama-2-7B MLP
    case 5120: return launch_ks<20, 5, EPI>(a, s);    // Llama-2-13B hidden
    case 13824: return launch_ks<54, 9, EPI>(a, s);   // Llama-2-13B MLP
    case 512: return launch_ks<2, 1, EPI>(a, s);      // reduced-width test / oracle configs
    case 1024: return launch_ks<4, 2, EPI>(a, s);
    case 1536: return launch_ks<6, 3, EPI>(a, s);
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
