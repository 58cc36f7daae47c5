// gemm_fp8.hip — FP8 (OCP e4m3) NT GEMM at the block-scaled MFMA rate:  C[M,N] = epi(sa[m]·sw[n]·Σ_k A8[m,k]·W8[n,k])
//
// BASELINE configs[4] ("fp8 MFMA GEMMs") has no counterpart in the reference; this is the bf16 256×256 staggered kernel of
// gemm_bf16.hip moved to `v_mfma_scale_f32_16x16x128_f8f6f4` with unit block scales (E8M0 0x7F), which runs e4m3 operands
// at twice the bf16 rate per clock (MI355X_MICROARCH.md, matrix cores). Everything that moves BYTES is unchanged: a K-tile
// is still 128 B per row (now 128 fp8 values), staged by the same LDS-DMA pieces into the same swizzled image, and the
// weights use the same fragment-major packing (bl_pack_weight_bf16 on the matrix viewed as bf16 pairs). One MFMA consumes
// the two 16-byte chunks a lane used to feed to two bf16 MFMAs: lane (r, g) supplies k ∈ {16g … 16g+15} ∪ {64+16g …} of
// its row for BOTH operands — a permutation of k inside the 128-wide step, which a dot product does not see (checked with
// exact integer data, tools/micro/fp8_mfma_check.hip). Per-token activation scales sa and per-output-channel weight
// scales sw (fp32) are applied to the fp32 accumulators before the usual fused epilogue. Quantisation of the activations:
// bl_quantize_rows_fp8 below (amax / 448 per row, hardware RNE conversion).
#include "gemm_common.h"

namespace bl_gemm_fp8_impl {
using namespace blgemm;

typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
constexpr int BK = 64;          // K-tile in 2-byte units = 128 fp8 values = 128 B per row
constexpr int ROW_BYTES = 128;
constexpr int GROUP_M = 8;
constexpr int ONE_E8M0 = 0x7F7F7F7F;   // block scale 2^0 in every byte

__device__ __forceinline__ void lin_to_tile(const GemmArgs& p, int lin, int& tm, int& tn) {
  const int width = GROUP_M * p.tiles_n, grp = lin / width, first = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first, GROUP_M), rem = lin - grp * width;
  tm = first + rem % gsz;
  tn = rem / gsz;
}

__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& tm, int& tn) {   // XCD-contiguous, grouped (gemm_bf16.hip)
  const int nwg = gridDim.x;
  const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  lin_to_tile(p, lin, tm, tn);
}

#define BL_GLDS(RS, LDSP, VOFF, SOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, LDS_PTR(LDSP), 16, VOFF, SOFF, 0, 0)

// p.K, p.lda are in 2-byte units (the host entry point halves the fp8 counts), so every address below is the bf16 kernel's.
template <int EPI>
__global__ __launch_bounds__(512) void gemm256s_fp8_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256;
  constexpr int STAGE = 65536, W_OFF = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int tm, tn;
  const int nk_all = p.K / BK;
  int kt_begin = 0, nk = nk_all;
  if (p.splitk > 1) {
    lin_to_tile(p, p.tail_base + blockIdx.x / p.splitk, tm, tn);
    const int slice = blockIdx.x % p.splitk;
    kt_begin = (int)(((long)slice * nk_all) / p.splitk) & ~1;
    nk = slice + 1 == p.splitk ? nk_all : ((int)(((long)(slice + 1) * nk_all) / p.splitk) & ~1);
  } else {
    tile_coords(p, tm, tn);
  }
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);

  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffX[2][2], voffY[2][2];
  int ldsX[2][2], ldsY[2];
#pragma unroll
  for (int mh = 0; mh < 2; ++mh)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int pi = 2 * wave + j, row0 = (pi >> 3) * 128 + mh * 64 + (pi & 7) * 8;
      voffX[mh][j] = (unsigned)(((long)(m0 + row0 + prow) * p.lda) * 2 + pchunk * 16);
      ldsX[mh][j] = row0 * ROW_BYTES;
    }
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
    const int nt = (wave >> 1) * 4 + 2 * nh + (wave & 1);
    voffY[nh][0] = (unsigned)((long)(n0 / 16 + nt) * kt32 * 1024 + lane * 16);
    voffY[nh][1] = voffY[nh][0] + 1024;
    ldsY[nh] = W_OFF + nt * 2048;
  }
#define ISSUE_X(MH, TILE)                                                                   \
  do {                                                                                      \
    const int t__ = (TILE);                                                                 \
    const __amdgpu_buffer_rsrc_t rs__ = t__ < nk ? rsA : rsA0;                              \
    char* b__ = smem + (t__ & 1) * STAGE;                                                   \
    BL_GLDS(rs__, b__ + ldsX[MH][0], voffX[MH][0], t__ * 128);                              \
    BL_GLDS(rs__, b__ + ldsX[MH][1], voffX[MH][1], t__ * 128);                              \
  } while (0)
#define ISSUE_Y(NH, TILE)                                                                   \
  do {                                                                                      \
    const int t__ = (TILE);                                                                 \
    const __amdgpu_buffer_rsrc_t rs__ = t__ < nk ? rsW : rsW0;                              \
    char* b__ = smem + (t__ & 1) * STAGE;                                                   \
    BL_GLDS(rs__, b__ + ldsY[NH], voffY[NH][0], t__ * 2048);                                \
    BL_GLDS(rs__, b__ + ldsY[NH] + 1024, voffY[NH][1], t__ * 2048);                         \
  } while (0)
  const int cb0 = (lg ^ (lane & 7)) << 4;
  const int offX = (wm * 128 + l15) * ROW_BYTES;
  const int offY = W_OFF + wn * 8192 + lane * 16;
#define READ_X(DST, MH, SB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
    const i32x4_t lo__ = *(const i32x4_t*)((SB) + offX + (MH) * 8192 + i * 2048 + cb0);                 \
    const i32x4_t hi__ = *(const i32x4_t*)((SB) + offX + (MH) * 8192 + i * 2048 + (cb0 ^ 64));          \
    DST[i] = __builtin_shufflevector(lo__, hi__, 0, 1, 2, 3, 4, 5, 6, 7);                               \
  }
#define READ_Y(DST, NH, SB)                                                                             \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
    const i32x4_t lo__ = *(const i32x4_t*)((SB) + offY + (2 * (NH) + j) * 2048);                        \
    const i32x4_t hi__ = *(const i32x4_t*)((SB) + offY + (2 * (NH) + j) * 2048 + 1024);                 \
    DST[j] = __builtin_shufflevector(lo__, hi__, 0, 1, 2, 3, 4, 5, 6, 7);                               \
  }
#define MMA(XR, YR, MH, NH)                                                                             \
  do {                                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
        acc[2 * (NH) + j][4 * (MH) + i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(             \
            YR[j], XR[i], acc[2 * (NH) + j][4 * (MH) + i], 0, 0, 0, ONE_E8M0, 0, ONE_E8M0);             \
    __builtin_amdgcn_s_setprio(0);                                                                      \
  } while (0)
#define BAR()                                   \
  do {                                          \
    __builtin_amdgcn_s_barrier();               \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define WAIT_VM10_LGKM() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory")
#define WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  // H(p+6) for phase q of K-tile T:  q=0 → X1(T+1)   q=1 → Y0(T+2)   q=2 → X0(T+2)   q=3 → Y1(T+2)
#define ISSUE_READ(I, R) do { R; __builtin_amdgcn_sched_barrier(0); I; } while (0)   // reads first (gemm_bf16.hip)
#define ISSUE_Q0(T) ISSUE_X(1, (T) + 1)
#define ISSUE_Q1(T) ISSUE_Y(0, (T) + 2)
#define ISSUE_Q2(T) ISSUE_X(0, (T) + 2)
#define ISSUE_Q3(T) ISSUE_Y(1, (T) + 2)
  // each group runs its own loop AND its own copy of the epilogue (a common epilogue after an if/else would force the
  // 128 accumulator registers of both loops into one assignment)
#define BL_EPILOGUE()                                                                                     \
  do {                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    if (p.splitk > 1) {                                                                                   \
      float* dst = p.slab + ((long)blockIdx.x * 32 * 512 + tid) * 4;                                      \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) *(f32x4_t*)(dst + (long)(i * 8 + j) * 512 * 4) = acc[i][j]; \
    } else {                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                     \
        const int n__ = n0 + wn * 64 + i * 16 + lg * 4;                                                   \
        const f32x4_t sw__ = n__ < p.N ? *(const f32x4_t*)(p.qw + n__) : (f32x4_t){0.f, 0.f, 0.f, 0.f};  \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                   \
          const int m__ = m0 + wm * 128 + j * 16 + l15;                                                   \
          const float sa__ = m__ < p.M ? p.qa[m__] : 0.f;                                                 \
          acc[i][j] = acc[i][j] * sw__ * sa__;          /* dequantise in place, then the whole-tile epilogue */ \
        }                                                                                                 \
      }                                                                                                   \
      const int ncol__[4] = {n0 + wn * 64 + lg * 4, n0 + wn * 64 + 16 + lg * 4, n0 + wn * 64 + 32 + lg * 4,       \
                             n0 + wn * 64 + 48 + lg * 4};                                                 \
      epilogue_tile<EPI, 4, 8>(p, m0 + wm * 128 + l15, ncol__, m0 + 256, n0 + 256, acc);                  \
    }                                                                                                     \
  } while (0)

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  i32x8_t X[4], Ya[2], Yb[2];
  char* const S0 = smem;
  char* const S1 = smem + STAGE;
  const int t0 = kt_begin;

  // ---- prologue: H(-1) = Y0(t0), H(0) = X0(t0), H(1) = Y1(t0), H(2) = X1(t0), H(3) = Y0(t0+1), H(4) = X0(t0+1),
  //      H(5) = Y1(t0+1); the first two must have landed before the first reads ----
  ISSUE_Y(0, t0); ISSUE_X(0, t0); ISSUE_Y(1, t0); ISSUE_X(1, t0); ISSUE_Y(0, t0 + 1); ISSUE_X(0, t0 + 1); ISSUE_Y(1, t0 + 1);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  BAR();
  READ_Y(Ya, 0, S0);

  if (wm == 0) {
    // ============================== group A: MMA first, then issue + read the NEXT phase ==============================
    READ_X(X, 0, S0);
    WAIT_LGKM();
    BAR();
    for (int t = t0; t < nk; t += 2) {
      // K-tile t (stage 0): Y0 in Ya, Y1 → Yb
      MMA(X, Ya, 0, 0); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q0(t), READ_Y(Yb, 1, S0)); WAIT_LGKM(); BAR();
      MMA(X, Yb, 0, 1); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q1(t), READ_X(X, 1, S0));  WAIT_LGKM(); BAR();
      MMA(X, Yb, 1, 1); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q2(t), READ_Y(Yb, 0, S1)); WAIT_LGKM(); BAR();
      MMA(X, Ya, 1, 0); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q3(t), READ_X(X, 0, S1));  WAIT_LGKM(); BAR();
      // K-tile t+1 (stage 1): Y0 in Yb, Y1 → Ya
      MMA(X, Yb, 0, 0); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q0(t + 1), READ_Y(Ya, 1, S1)); WAIT_LGKM(); BAR();
      MMA(X, Ya, 0, 1); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q1(t + 1), READ_X(X, 1, S1));  WAIT_LGKM(); BAR();
      MMA(X, Ya, 1, 1); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q2(t + 1), READ_Y(Ya, 0, S0)); WAIT_LGKM(); BAR();
      MMA(X, Yb, 1, 0); WAIT_VM8(); BAR();   ISSUE_READ(ISSUE_Q3(t + 1), READ_X(X, 0, S0));  WAIT_LGKM(); BAR();
    }
    BL_EPILOGUE();
    return;
  }
  // ============================== group B: issue + read THIS phase, then MMA ==============================
  WAIT_LGKM();
  BAR();
  for (int t = t0; t < nk; t += 2) {
    ISSUE_READ(ISSUE_Q0(t), READ_X(X, 0, S0));  WAIT_VM10_LGKM(); BAR();   MMA(X, Ya, 0, 0); BAR();
    ISSUE_READ(ISSUE_Q1(t), READ_Y(Yb, 1, S0)); WAIT_VM10_LGKM(); BAR();   MMA(X, Yb, 0, 1); BAR();
    ISSUE_READ(ISSUE_Q2(t), READ_X(X, 1, S0));  WAIT_VM10_LGKM(); BAR();   MMA(X, Yb, 1, 1); BAR();
    ISSUE_READ(ISSUE_Q3(t), READ_Y(Yb, 0, S1)); WAIT_VM10_LGKM(); BAR();   MMA(X, Ya, 1, 0); BAR();
    ISSUE_READ(ISSUE_Q0(t + 1), READ_X(X, 0, S1));  WAIT_VM10_LGKM(); BAR();   MMA(X, Yb, 0, 0); BAR();
    ISSUE_READ(ISSUE_Q1(t + 1), READ_Y(Ya, 1, S1)); WAIT_VM10_LGKM(); BAR();   MMA(X, Ya, 0, 1); BAR();
    ISSUE_READ(ISSUE_Q2(t + 1), READ_X(X, 1, S1));  WAIT_VM10_LGKM(); BAR();   MMA(X, Ya, 1, 1); BAR();
    ISSUE_READ(ISSUE_Q3(t + 1), READ_Y(Ya, 0, S0)); WAIT_VM10_LGKM(); BAR();   MMA(X, Yb, 1, 0); BAR();
  }
  BL_EPILOGUE();
#undef BL_EPILOGUE
#undef ISSUE_X
#undef ISSUE_Y
#undef READ_X
#undef READ_Y
#undef MMA
#undef BAR
#undef WAIT_VM8
#undef WAIT_VM10_LGKM
#undef WAIT_LGKM
#undef ISSUE_READ
#undef ISSUE_Q0
#undef ISSUE_Q1
#undef ISSUE_Q2
#undef ISSUE_Q3
#endif
}

// Row-resident form of the kernel below for cols <= 512·NCH: the lane's NCH 16-byte chunks of the row are requested at once
// and stay in registers between the amax pass and the conversion — the row crosses the memory system once, with NCH loads in
// flight per lane instead of one (training-step quantisations 19 → 13 ms per step). Same arithmetic, same results.
template <int NCH>
__global__ __launch_bounds__(256) void quantize_rows_fp8_resident_kernel(const uint16_t* x, long ldx, int rows, int cols, uint8_t* q,
                                                                         long ldq, float* scales) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint16_t* xr = x + (long)row * ldx;
  u32x4_t t[NCH];
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lane * 8 + k * 512;
    t[k] = (u32x4_t){0u, 0u, 0u, 0u};
    if (c < cols) t[k] = *(const u32x4_t*)(xr + c);
  }
  float amax = 0.f;
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bflo(t[k][i])), fabsf(bfhi(t[k][i]))));
  amax = wave_max(amax);
  const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
  if (lane == 0) scales[row] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
  uint8_t* qr = q + (long)row * ldq;
#pragma unroll
  for (int k = 0; k < NCH; ++k)       // keep the row PACKED between the passes (hipcc would carry the 8 unpacked floats per chunk)
    asm volatile("" : "+v"(t[k][0]), "+v"(t[k][1]), "+v"(t[k][2]), "+v"(t[k][3]));
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int c = lane * 8 + k * 512;
    if (c < cols) {
      u32x2_t o;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(t[k][2 * i]) * inv, bfhi(t[k][2 * i]) * inv, w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(t[k][2 * i + 1]) * inv, bfhi(t[k][2 * i + 1]) * inv, w, true);
        o[i] = (uint32_t)w;
      }
      *(u32x2_t*)(qr + c) = o;
    }
  }
}

// x bf16 [rows, cols] → q fp8 e4m3 [rows, cols] + scale[row] = amax / 448 (1 for an all-zero row): one wave per row.
__global__ __launch_bounds__(256) void quantize_rows_fp8_kernel(const uint16_t* x, long ldx, int rows, int cols, uint8_t* q,
                                                                long ldq, float* scales) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const uint16_t* xr = x + (long)row * ldx;
  float amax = 0.f;
  for (int c = lane * 8; c < cols; c += 512) {
    const u32x4_t t = *(const u32x4_t*)(xr + c);
#pragma unroll
    for (int i = 0; i < 4; ++i) amax = fmaxf(amax, fmaxf(fabsf(bflo(t[i])), fabsf(bfhi(t[i]))));
  }
  amax = wave_max(amax);
  const float inv = amax > 0.f ? __fdiv_rn(448.0f, amax) : 1.0f;
  if (lane == 0) scales[row] = amax > 0.f ? __fdiv_rn(amax, 448.0f) : 1.0f;
  uint8_t* qr = q + (long)row * ldq;
  for (int c = lane * 8; c < cols; c += 512) {
    const u32x4_t t = *(const u32x4_t*)(xr + c);
    u32x2_t o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(t[2 * i]) * inv, bfhi(t[2 * i]) * inv, w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(bflo(t[2 * i + 1]) * inv, bfhi(t[2 * i + 1]) * inv, w, true);
      o[i] = (uint32_t)w;
    }
    *(u32x2_t*)(qr + c) = o;
  }
}


// ---- weight-gradient operands on the e4m3 path (TrainStep(fp8_wgrad=True)) -------------------------------------------
// dW[n, k] = Σ_t dy[t, n]·x[t, k] contracts over TOKENS: both operands must be token-contiguous with one scale per
// channel. Rounds 2-3 made them in five passes (transpose, quantise, transpose, quantise, pack: 16 B moved per pair of
// elements, 37 ms per 7B step — more than the e4m3 GEMMs saved). Now two: a column |max| pass (read 2 B) and ONE pass
// that reads the bf16 [T, C] matrix where it lies, scales by 448 / amax[c], converts, transposes through LDS and writes
// the e4m3 codes token-contiguous (1 B) — row-major [C, Tq] for the GEMM's activation side, or straight in the
// fragment-major packing of its weight side (one wave = one 1-KiB block of 16 channels x 64 tokens).
__global__ __launch_bounds__(256) void colamax_bf16_kernel(const uint16_t* x, long ldx, int rows, int cols, int rows_per_wg,
                                                           unsigned* amax_bits) {
  __shared__ float sh[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 512 + lane * 8;
  const int r0 = blockIdx.y * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = 0.f;
  if (c < cols) {
    int r = r0 + wave;
    for (; r + 12 < r1; r += 16) {                               // four rows of this wave in flight
      u32x4_t t[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) t[u] = *(const u32x4_t*)(x + (long)(r + 4 * u) * ldx + c);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          m[2 * i] = fmaxf(m[2 * i], fabsf(bflo(t[u][i])));
          m[2 * i + 1] = fmaxf(m[2 * i + 1], fabsf(bfhi(t[u][i])));
        }
    }
    for (; r < r1; r += 4) {
      const u32x4_t t = *(const u32x4_t*)(x + (long)r * ldx + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m[2 * i] = fmaxf(m[2 * i], fabsf(bflo(t[i])));
        m[2 * i + 1] = fmaxf(m[2 * i + 1], fabsf(bfhi(t[i])));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) sh[wave][lane * 8 + i] = m[i];
  __syncthreads();
  for (int j = threadIdx.x; j < 512; j += 256) {
    const float v = fmaxf(fmaxf(sh[0][j], sh[1][j]), fmaxf(sh[2][j], sh[3][j]));
    const int cc = blockIdx.x * 512 + j;
    if (cc < cols && v > 0.f) atomicMax(amax_bits + cc, __float_as_uint(v));      // non-negative floats order like their bits
  }
}

// x bf16 [rows = tokens, cols = channels] → e4m3 codes, token-contiguous per channel, tokens zero-padded to ldq.
// PACKED: q = [cols/16][ldq/64][64 lanes][16 B], lane (ch % 16) + 16 * ((tok % 64) / 16) — the layout bl_gemm_fp8 reads its
// weight operand in; else q = [cols][ldq] row-major (its activation operand). One workgroup = 64 channels x 256 tokens.
template <bool PACKED>
__global__ __launch_bounds__(256) void transpose_quantize_fp8_kernel(const uint16_t* x, long ldx, int rows, int cols,
                                                                     const float* amax, uint8_t* q, long ldq, float* scales) {
  constexpr int ROWB = 132;                                      // 64 channels x 2 B + 4: the 16-token stride lands 16 banks apart
  __shared__ __attribute__((aligned(16))) char tile[64 * ROWB];
  __shared__ float inv_s[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.x * 64;
  if (tid < 64) {
    const int c = c0 + tid;
    const float a = c < cols ? amax[c] : 0.f;
    inv_s[tid] = a > 0.f ? __fdiv_rn(448.0f, a) : 1.0f;
    if (blockIdx.y == 0 && c < cols) scales[c] = a > 0.f ? __fdiv_rn(a, 448.0f) : 1.0f;
  }
  const int ch = lane & 15, g = lane >> 4;                       // this lane's channel (within the wave's 16) and 16-token group
  const int cw = wave * 16 + ch;                                 // channel within the tile
  for (int tt = 0; tt < 4; ++tt) {
    const int t0 = (blockIdx.y * 4 + tt) * 64;
    if (t0 >= (int)ldq) break;
    __syncthreads();                                             // previous tile consumed (and inv_s visible)
    // stage 64 tokens x 64 channels: 512 chunks of 16 B, two per thread
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int p = tid + u * 256, r = p >> 3, k = p & 7;
      u32x4_t v = {0u, 0u, 0u, 0u};
      if (t0 + r < rows && c0 + k * 8 < cols) v = *(const u32x4_t*)(x + (long)(t0 + r) * ldx + c0 + k * 8);
      uint32_t* d = (uint32_t*)(tile + r * ROWB + k * 16);
      d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    __syncthreads();
    const float inv = inv_s[cw];
    u32x4_t o;
#pragma unroll
    for (int w4 = 0; w4 < 4; ++w4) {
      float f[4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
        f[i] = bf2f(*(const uint16_t*)(tile + (g * 16 + w4 * 4 + i) * ROWB + cw * 2)) * inv;
      int w = 0;
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
      w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
      o[w4] = (uint32_t)w;
    }
    if (c0 + cw < cols) {
      if (PACKED) *(u32x4_t*)(q + (((long)((c0 >> 4) + wave) * (ldq >> 6) + (t0 >> 6)) * 64 + lane) * 16) = o;
      else *(u32x4_t*)(q + (long)(c0 + cw) * ldq + t0 + g * 16) = o;
    }
  }
}

template <int EPI>
int launch_fp8(const GemmArgs& a, hipStream_t s) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_fp8_kernel<EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * 65536) != hipSuccess)
      return BL_E_LAUNCH;
    done = true;
  }
  GemmArgs p = a;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  hipLaunchKernelGGL((gemm256s_fp8_kernel<EPI>), dim3(p.tiles_m * p.tiles_n), dim3(512), 2 * 65536, s, p);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

}  // namespace bl_gemm_fp8_impl
using namespace bl_gemm_fp8_impl;

extern "C" int bl_gemm_fp8(const bl_gemm_desc* d, const float* scale_a, const float* scale_w, void* stream) {
  if (!d || !scale_a || !scale_w) return BL_E_ARG;
  if ((d->K % 128) || (d->lda % 16) || d->ldw != d->K || d->a_norm_weight || d->workspace) return BL_E_SHAPE;
  if ((((uintptr_t)scale_w) & 15) || (((uintptr_t)scale_a) & 3)) return BL_E_ALIGN;
  bl_gemm_desc h = *d;                       // the kernel addresses in 2-byte units
  h.K = d->K / 2; h.lda = d->lda / 2; h.ldw = h.K;
  GemmArgs a;
  const int rc = fill_gemm_args(&h, a);
  if (rc != BL_OK) return rc;
  a.qa = scale_a; a.qw = scale_w;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case BL_EPI_NONE: return launch_fp8<BL_EPI_NONE>(a, s);
    case BL_EPI_BIAS: return launch_fp8<BL_EPI_BIAS>(a, s);
    case BL_EPI_BIAS_GELU: return launch_fp8<BL_EPI_BIAS_GELU>(a, s);
    case BL_EPI_BIAS_RES: return launch_fp8<BL_EPI_BIAS_RES>(a, s);
    case BL_EPI_RES: return launch_fp8<BL_EPI_RES>(a, s);
    case BL_EPI_SWIGLU: return launch_fp8<BL_EPI_SWIGLU>(a, s);
    case BL_EPI_F32: return launch_fp8<BL_EPI_F32>(a, s);
    case BL_EPI_F32_BF16R: return launch_fp8<BL_EPI_F32_BF16R>(a, s);
    default: return BL_E_ARG;
  }
}

extern "C" int bl_quantize_rows_fp8(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, uint8_t* q, int64_t ldq,
                                    float* scales, void* stream) {
  if (!x || !q || !scales) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || ldx < cols || ldq < cols) return BL_E_SHAPE;
  if ((ldx % 8) || (ldq % 8) || !bl_aligned16(x) || (((uintptr_t)q) & 7)) return BL_E_ALIGN;
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t s = (hipStream_t)stream;
#define BL_QR(NCH) hipLaunchKernelGGL((quantize_rows_fp8_resident_kernel<NCH>), grid, block, 0, s, x, (long)ldx, rows, cols, q, (long)ldq, scales)
  if (cols <= 512 * 8) BL_QR(8);
  else if (cols <= 512 * 11) BL_QR(11);
  else if (cols <= 512 * 24) BL_QR(24);
  else if (cols <= 512 * 27) BL_QR(27);
  else if (cols <= 512 * 43) BL_QR(43);
  else if (cols <= 512 * 54) BL_QR(54);
  else hipLaunchKernelGGL(quantize_rows_fp8_kernel, grid, block, 0, s, x, (long)ldx, rows, cols, q, (long)ldq, scales);
#undef BL_QR
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_colamax_bf16(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, float* amax, void* stream) {
  if (!x || !amax) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || ldx < cols) return BL_E_SHAPE;
  if ((ldx % 8) || !bl_aligned16(x) || (((uintptr_t)amax) & 3)) return BL_E_ALIGN;
  const int rpw = 256;
  hipLaunchKernelGGL(colamax_bf16_kernel, dim3((cols + 511) / 512, (rows + rpw - 1) / rpw), dim3(256), 0, (hipStream_t)stream, x,
                     (long)ldx, rows, cols, rpw, (unsigned*)amax);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_transpose_quantize_fp8(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, const float* amax, uint8_t* q,
                                         int64_t ldq, int32_t packed, float* scales, void* stream) {
  if (!x || !amax || !q || !scales) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 16) || ldx < cols || ldq < rows || (ldq % 64)) return BL_E_SHAPE;
  if ((ldx % 8) || !bl_aligned16(x) || !bl_aligned16(q)) return BL_E_ALIGN;
  const dim3 grid((cols + 63) / 64, (unsigned)((ldq + 255) / 256)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (packed) hipLaunchKernelGGL((transpose_quantize_fp8_kernel<true>), grid, block, 0, s, x, (long)ldx, rows, cols, amax, q, (long)ldq, scales);
  else hipLaunchKernelGGL((transpose_quantize_fp8_kernel<false>), grid, block, 0, s, x, (long)ldx, rows, cols, amax, q, (long)ldq, scales);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
