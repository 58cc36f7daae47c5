// gemm_bf16.hip — bf16 NT GEMM on gfx950 MFMA with fused epilogues:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Design (MI355X-first):
//   * both operands are K-contiguous (activations [M,K], nn.Linear weights [N,K]) so tiles are staged HBM→LDS with
//     LDS-DMA (`buffer_load_dwordx4 … lds`, 1 KiB per wave-instruction = 8 rows × 128 B of a BK=64 tile). The buffer
//     descriptor's bounds check zero-fills rows past M / N, so ragged M (e.g. 16·261) needs no host padding.
//   * LDS-DMA writes lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address
//     (16-byte chunk c of row r is fetched from chunk c ^ (r & 7)) and undone on the ds_read_b128 side.
//   * MFMA is v_mfma_f32_16x16x32_bf16 computed TRANSPOSED: the weight tile is the "A" operand and the activation
//     tile the "B" operand, so a lane ends up holding 4 consecutive output columns n of one row m → 8-byte stores
//     and vector loads of bias / LayerScale / residual in the epilogue.
//   * 1-D grid remapped so that each XCD (private L2) receives a contiguous run of tiles, walked in groups of
//     GROUP_M row-tiles so concurrently resident tiles share weight panels.
#include "gemm_common.h"

namespace bl_gemm_bf16_impl {
using namespace blgemm;


constexpr int BK = 64;          // bf16 elements per K-step = 128 B per tile row
constexpr int ROW_BYTES = 128;
constexpr int GROUP_M = 8;

// BM × BN output tile (BM activation rows, BN weight rows), WM × WN waves.
template <int BM, int BN, int WM, int WN, int EPI>
__global__ __launch_bounds__(WM* WN * 64) void gemm_nt_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // __amdgpu_buffer_rsrc_t does not exist in the host pass; an ill-formed host body
                                     // silently drops the kernel's host handle (undefined symbol at dlopen)
  constexpr int NWAVE = WM * WN;
  constexpr int TM = BM / WM, TN = BN / WN;      // per-wave tile
  constexpr int MT = TM / 16, NT = TN / 16;      // MFMA tiles per wave
  constexpr int PIECES_A = BM / 8, PIECES_W = BN / 8;   // 1-KiB LDS-DMA pieces per K-step
  constexpr int PA = PIECES_A / NWAVE, PW = PIECES_W / NWAVE;
  constexpr int A_BYTES = BM * ROW_BYTES, W_BYTES = BN * ROW_BYTES, BUF_BYTES = A_BYTES + W_BYTES;
  static_assert(PIECES_A % NWAVE == 0 && PIECES_W % NWAVE == 0, "pieces must divide over waves");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 × BUF_BYTES

  // ---- tile coordinates: XCD-contiguous + grouped ordering (speed only; any mapping is correct) ----
  const int nwg = p.tiles_m * p.tiles_n;
  int lin;
  {
    const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  int tm, tn;
  {
    const int width = GROUP_M * p.tiles_n, grp = lin / width, first = grp * GROUP_M;
    const int gsz = min(p.tiles_m - first, GROUP_M), rem = lin - grp * width;
    tm = first + rem % gsz;
    tn = rem / gsz;
  }
  const int m0 = tm * BM, n0 = tn * BN;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  // ---- buffer descriptors (bounds check zero-fills rows ≥ M / ≥ N) ----
  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.ldw * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  // per-lane source offsets of this wave's pieces (k-independent part); lane → (row = lane>>3, chunk = lane&7)
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;   // swizzled source chunk
  unsigned voffA[PA], voffW[PW];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int r = (j * NWAVE + wave) * 8 + prow;
    voffA[j] = (unsigned)(((long)(m0 + r) * p.lda) * 2 + pchunk * 16);
  }
#pragma unroll
  for (int j = 0; j < PW; ++j) {
    const int r = (j * NWAVE + wave) * 8 + prow;
    voffW[j] = (unsigned)(((long)(n0 + r) * p.ldw) * 2 + pchunk * 16);
  }

  // (a macro, not a lambda: a lambda inside a __global__ template makes hipcc's HOST pass drop the kernel handle)
#define BL_STAGE(BUF, KT)                                                                                             \
  do {                                                                                                                \
    const int koff__ = (KT) * (BK * 2);                                                                               \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                          \
    _Pragma("unroll") for (int j = 0; j < PA; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                          \
        rsA, LDS_PTR(base__ + (j * NWAVE + wave) * 1024), 16, voffA[j], koff__, 0, 0);                                \
    _Pragma("unroll") for (int j = 0; j < PW; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                          \
        rsW, LDS_PTR(base__ + A_BYTES + (j * NWAVE + wave) * 1024), 16, voffW[j], koff__, 0, 0);                      \
  } while (0)

  // ---- fragment read offsets ----
  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);                 // swizzled chunk of k-step 0 (k-step 1: c0 ^ 4)
  const int offA = (wm * TM + l15) * ROW_BYTES;   // activation tile ("B" operand of the MFMA)
  const int offW = A_BYTES + (wn * TN + l15) * ROW_BYTES;

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  BL_STAGE(0, 0);
  __syncthreads();   // includes vmcnt(0): tile 0 landed

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) BL_STAGE(cur ^ 1, kt + 1);
    const char* base = smem + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cb = (c0 ^ (ks * 4)) << 4;
      bf16x8_t wf[NT], af[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *(const bf16x8_t*)(base + offW + i * 16 * ROW_BYTES + cb);
#pragma unroll
      for (int j = 0; j < MT; ++j) af[j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();   // next tile landed (vmcnt(0)) and everyone is done reading `cur`
  }

  // ---- epilogue: lane holds D[n = 4*lg + r][m = l15] of each 16×16 tile ----
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      const int n = n0 + wn * TN + i * 16 + lg * 4;
      const int m = m0 + wm * TM + j * 16 + l15;
      epilogue_store4<EPI>(p, m, n, acc[i][j]);
    }
#endif
}

#undef BL_STAGE

template <int EPI>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
  constexpr int BM = 128, BN = 128, WM = 2, WN = 2;
  GemmArgs p = a;
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int lds = 2 * (BM + BN) * ROW_BYTES;
  static bool attr_set = false;   // idempotent; a benign race only repeats the same call
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_kernel<BM, BN, WM, WN, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
      return BL_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, EPI>), dim3(p.tiles_m * p.tiles_n), dim3(WM * WN * 64), lds, s, p);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

}  // namespace bl_gemm_bf16_impl
using namespace bl_gemm_bf16_impl;

extern "C" int bl_gemm_bf16(const bl_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = fill_gemm_args(d, a);
  if (rc != BL_OK) return rc;
  const int epi = d->epilogue;
  hipStream_t s = (hipStream_t)stream;
  switch (epi) {
    case BL_EPI_NONE: return launch_gemm<BL_EPI_NONE>(a, s);
    case BL_EPI_BIAS: return launch_gemm<BL_EPI_BIAS>(a, s);
    case BL_EPI_BIAS_GELU: return launch_gemm<BL_EPI_BIAS_GELU>(a, s);
    case BL_EPI_BIAS_RES: return launch_gemm<BL_EPI_BIAS_RES>(a, s);
    case BL_EPI_RES: return launch_gemm<BL_EPI_RES>(a, s);
    case BL_EPI_SWIGLU: return launch_gemm<BL_EPI_SWIGLU>(a, s);
    case BL_EPI_F32: return launch_gemm<BL_EPI_F32>(a, s);
    case BL_EPI_F32_BF16R: return launch_gemm<BL_EPI_F32_BF16R>(a, s);
    default: return BL_E_ARG;
  }
}
