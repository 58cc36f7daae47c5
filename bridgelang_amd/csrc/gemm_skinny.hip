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
//   * the 8 partial 16×16 fp32 tiles are summed (balanced binary tree) through LDS behind a RAW s_barrier (a __syncthreads() would emit
//     vmcnt(0) and drain the prefetch), double-buffered, and a rotating wave applies the fused epilogue.
#include "gemm_common.h"

namespace bl_gemm_skinny_impl {
using namespace blgemm;

constexpr int NW = 8;   // waves per workgroup = K split factor

// lane (l15, lg) of wave `wave` holds x[row0 + l15][kbase + 32*s .. +7] for s < KS (kbase = wave*KS*32 + lg*8)
template <int KS>
__device__ __forceinline__ void load_x_fragments(const uint16_t* A, long lda, int M, int l15, long kbase,
                                                 bf16x8_t (&xf)[KS]) {
  const bool live = l15 < M;
  const uint16_t* xp = A + (long)(live ? l15 : M - 1) * lda + kbase;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    u32x4_t t = *(const u32x4_t*)(xp + s * 32);
    if (!live) t = (u32x4_t){0u, 0u, 0u, 0u};
    xf[s] = __builtin_bit_cast(bf16x8_t, t);
  }
}

// Fused HF LlamaRMSNorm on 16 activation rows held as fragments by the 8 waves of a workgroup: sum of squares over the
// lane's K positions (explicit FMAs, fixed order), across the 4 lane groups that share a row, then across the 8 waves
// through LDS in wave order; the fragments are rewritten in registers with HF's two roundings
// (weight * bf16(x * rstd)). ONE definition shared by the weight-streaming GEMM and bl_rmsnorm_skinny_bf16, so the
// merged-decode path (norm kernel + bl_gemm_skinny_rows_bf16) reproduces the fused path bit for bit.
template <int KS>
__device__ __forceinline__ void rmsnorm_fragments(const uint16_t* norm_w, float eps, int K, long kbase, int wave, int l15,
                                                  int lg, float (&nrm)[8][16], bf16x8_t (&xf)[KS]) {
  u32x4_t gw[KS];                       // norm weights for this lane's k positions: issued before the reduction
  {
    const uint16_t* wn = norm_w + kbase;
#pragma unroll
    for (int s = 0; s < KS; ++s) gw[s] = *(const u32x4_t*)(wn + s * 32);
  }
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const u32x4_t t = __builtin_bit_cast(u32x4_t, xf[s]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ss = __builtin_fmaf(bflo(t[i]), bflo(t[i]), ss);
      ss = __builtin_fmaf(bfhi(t[i]), bfhi(t[i]), ss);
    }
  }
  ss += __shfl_xor(ss, 16, 64);
  ss += __shfl_xor(ss, 32, 64);
  if (lg == 0) nrm[wave][l15] = ss;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // raw barrier: keep prefetched weight loads in flight
  __builtin_amdgcn_s_barrier();
  float tot = 0.f;
#pragma unroll
  for (int w2 = 0; w2 < 8; ++w2) tot += nrm[w2][l15];
  const float rstd = 1.0f / sqrtf(__builtin_fmaf(tot, 1.0f / (float)K, eps));
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
  load_x_fragments<KS>(p.A, p.lda, p.M, l15, kbase, xf);

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
    __shared__ float nrm[NW][16];
    rmsnorm_fragments<KS>(p.norm_w, p.norm_eps, p.K, kbase, wave, l15, lg, nrm, xf);
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
      // balanced binary tree over the 8 slice partials: an aligned group of slices has a value of its own, so the
      // many-rows form of this arithmetic (bl_gemm_skinny_rows_bf16) may split K across workgroups exactly
      const f32x4_t (&r)[NW][64] = red[it & 1];
      const f32x4_t sum = ((r[0][lane] + r[1][lane]) + (r[2][lane] + r[3][lane])) +
                          ((r[4][lane] + r[5][lane]) + (r[6][lane] + r[7][lane]));
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


// RMSNorm of rows in the fused-norm arithmetic of gemm_skinny_kernel: one workgroup per 16 rows, fragments in, fragments
// out (y may alias x).
template <int KS>
__global__ __launch_bounds__(NW * 64) void rmsnorm_skinny_kernel(const uint16_t* x, long ldx, const uint16_t* w,
                                                                 uint16_t* y, long ldy, int rows, int K, float eps) {
  __shared__ float nrm[NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const long kbase = (long)wave * (KS * 32) + lg * 8;
  const int row0 = blockIdx.x * 16, m = min(16, rows - row0);
  bf16x8_t xf[KS];
  load_x_fragments<KS>(x + (long)row0 * ldx, ldx, m, l15, kbase, xf);
  rmsnorm_fragments<KS>(w, eps, K, kbase, wave, l15, lg, nrm, xf);
  if (l15 < m) {
    uint16_t* yp = y + (long)(row0 + l15) * ldy + kbase;
#pragma unroll
    for (int s = 0; s < KS; ++s) *(u32x4_t*)(yp + s * 32) = __builtin_bit_cast(u32x4_t, xf[s]);
  }
}

template <int KS>
int launch_norm_skinny(const uint16_t* x, long ldx, const uint16_t* w, uint16_t* y, long ldy, int rows, int K, float eps,
                       hipStream_t s) {
  hipLaunchKernelGGL((rmsnorm_skinny_kernel<KS>), dim3((rows + 15) / 16), dim3(NW * 64), 0, s, x, ldx, w, y, ldy, rows, K, eps);
  BL_CHECK_LAUNCH();
  return BL_OK;
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

extern "C" int bl_rmsnorm_skinny_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, bl_bf16* y, int64_t ldy,
                                      int32_t rows, int32_t dim, float eps, void* stream) {
  if (!x || !w || !y) return BL_E_ARG;
  if (rows <= 0) return BL_E_SHAPE;
  if ((ldx % 8) || (ldy % 8) || ldx < dim || ldy < dim || !bl_aligned16(x) || !bl_aligned16(y) || !bl_aligned16(w))
    return BL_E_ALIGN;
  hipStream_t s = (hipStream_t)stream;
  switch (dim) {   // the K values gemm_skinny_kernel is instantiated for
    case 4096: return launch_norm_skinny<16>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 11008: return launch_norm_skinny<43>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 5120: return launch_norm_skinny<20>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 13824: return launch_norm_skinny<54>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 512: return launch_norm_skinny<2>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 1024: return launch_norm_skinny<4>(x, ldx, w, y, ldy, rows, dim, eps, s);
    case 1536: return launch_norm_skinny<6>(x, ldx, w, y, ldy, rows, dim, eps, s);
    default: return BL_E_SHAPE;
  }
}
