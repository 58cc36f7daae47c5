// gemm_bf16.hip — bf16 NT GEMM on gfx950 MFMA with fused epilogues:  C[M,N] = epi(A[M,K] · W[N,K]^T)
//
// Design (MI355X-first):
//   * activations A[M,K] are row-major; weights are fragment-major (bl_pack_weight_bf16: one contiguous KiB per
//     16-row × 32-k MFMA operand). Tiles are staged HBM→LDS with LDS-DMA (`buffer_load_dwordx4 … lds`, 1 KiB per
//     wave-instruction). The buffer descriptor's bounds check zero-fills rows past M / weight tiles past N, so ragged M
//     (e.g. 16·261) needs no host padding.
//   * LDS-DMA writes lane-linear. Weight blocks therefore land already in ds_read_b128 order (address = block + 16·lane:
//     conflict-free, no swizzle). Activation rows (128 B per row per K-step) get the bank-conflict swizzle on the
//     per-lane SOURCE address (chunk c of row r is fetched from chunk c ^ (r & 7)) and undone on the read side.
//   * MFMA is v_mfma_f32_16x16x32_bf16 computed TRANSPOSED: the weight tile is the "A" operand and the activation
//     tile the "B" operand, so a lane ends up holding 4 consecutive output columns n of one row m → 8-byte stores
//     and vector loads of bias / LayerScale / residual in the epilogue.
//   * 1-D grid remapped so that each XCD (private L2) receives a contiguous run of tiles, walked in groups of
//     GROUP_M row-tiles so concurrently resident tiles share weight panels.
//
// Two kernels:
//   gemm128_kernel — 128×128×64 tile, 4 waves (64×64 each), 2 workgroups/CU, one vmcnt(0)+barrier per K-step. Small or
//                    skinny problems, and the ragged tail of big ones.
//   gemm256_kernel — 256×256×64 tile, 8 waves (128(m)×64(n) each), 128 KiB LDS = 2 stages, 1 workgroup/CU. Each K-tile
//                    is staged as four 16-KiB HALF-TILES (X0/X1 = the two 64-row halves of every wave's activation
//                    rows, Y0/Y1 = the two 32-row halves of every wave's weight rows) and consumed in four PHASES of 16
//                    MFMAs (output quadrants (0,0) (0,1) (1,1) (1,0)). Per phase: issue one half-tile of LDS-DMA for a
//                    later K-tile, prefetch the NEXT phase's fragments LDS→VGPR, run this phase's MFMAs, counted
//                    `s_waitcnt vmcnt(4)` (two half-tiles stay in flight across the barrier — never 0 in the loop),
//                    ONE raw s_barrier. Every half-tile is issued ≥ 3 phases before its first read.
#include "gemm_common.h"
#include <stdlib.h>

namespace bl_gemm_bf16_impl {
using namespace blgemm;

constexpr int BK = 64;          // bf16 elements per K-step = 128 B per activation-tile row
constexpr int ROW_BYTES = 128;
constexpr int GROUP_M = 4;   // round 3 same-box A/B: 4 beats 8 by 0.75 % end to end (gate||up -1.7 %), 2 / 3 tie, 9 … 18 lose

// linear tile index → (row tile, column tile): groups of GROUP_M row-tiles walked column by column
__device__ __forceinline__ void lin_to_tile(const GemmArgs& p, int lin, int& tm, int& tn) {
  const int width = GROUP_M * p.tiles_n, grp = lin / width, first = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first, GROUP_M), rem = lin - grp * width;
  tm = first + rem % gsz;
  tn = rem / gsz;
}

__device__ __forceinline__ void tile_coords(const GemmArgs& p, int& tm, int& tn) {
  // XCD-contiguous ordering over the LAUNCHED blocks (speed only; any bijective mapping is correct). The grid may
  // cover only the first gridDim.x tiles of the tiles_m x tiles_n grid (whole rounds; the tail goes to gemm128).
  const int nwg = gridDim.x;
  const int orig = blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  const int width = GROUP_M * p.tiles_n, grp = lin / width, first = grp * GROUP_M;
  const int gsz = min(p.tiles_m - first, GROUP_M), rem = lin - grp * width;
  tm = first + rem % gsz;
  tn = rem / gsz;
}

// the same mapping for a VIRTUAL block index vb of nwg (persistent kernels: a workgroup walks vb = blockIdx.x + r·gridDim.x;
// with gridDim.x a multiple of 8 every tile of a workgroup belongs to its own XCD's contiguous run)
__device__ __forceinline__ void tile_coords_v(const GemmArgs& p, int vb, int nwg, int& tm, int& tn) {
  const int xcd = vb & 7, q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (vb >> 3);
  lin_to_tile(p, lin, tm, tn);
}

#define BL_GLDS(RS, LDSP, VOFF, SOFF) __builtin_amdgcn_raw_ptr_buffer_load_lds(RS, LDS_PTR(LDSP), 16, VOFF, SOFF, 0, 0)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;   // result of ds_read_b64_tr_b16

// ======================================================================================================================
// 128 × 128 tile
// ======================================================================================================================
template <int EPI>
__global__ __launch_bounds__(256) void gemm128_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // __amdgpu_buffer_rsrc_t does not exist in the host pass; an ill-formed host body
                                      // silently drops the kernel's host handle (undefined symbol at dlopen)
  constexpr int BM = 128, BN = 128, WN = 2, NWAVE = 4;
  constexpr int TM = 64, TN = 64, MT = 4, NT = 4;
  constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = A_BYTES + BN * ROW_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 × BUF_BYTES

  int tm, tn, m0, n0;
  if (p.tail_base >= 0) {   // tail mode: 4 small tiles per leftover 256x256 tile of the big kernel's grid
    lin_to_tile(p, p.tail_base + (blockIdx.x >> 2), tm, tn);
    m0 = tm * 256 + ((blockIdx.x >> 1) & 1) * 128;
    n0 = tn * 256 + (blockIdx.x & 1) * 128;
  } else {
    tile_coords(p, tm, tn);
    m0 = tm * BM, n0 = tn * BN;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  // activation pieces: 16 × (8 rows × 128 B); wave takes pieces j*4 + wave. lane → (row = lane>>3, chunk = lane&7)
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[4], voffW[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) voffA[j] = (unsigned)(((long)(m0 + (j * NWAVE + wave) * 8 + prow) * p.lda) * 2 + pchunk * 16);
  // weight blocks: 8 n-tiles × 2 k-steps; wave takes n-tiles 2*wave, 2*wave+1 (both k-steps: 2 KiB contiguous each)
#pragma unroll
  for (int j = 0; j < 4; ++j)
    voffW[j] = (unsigned)(((long)(n0 / 16 + 2 * wave + (j >> 1)) * kt32 + (j & 1)) * 1024 + lane * 16);

#define BL_STAGE(BUF, KT)                                                                                   \
  do {                                                                                                      \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) BL_GLDS(rsA, base__ + (j * NWAVE + wave) * 1024, voffA[j], (KT) * 128); \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                           \
        BL_GLDS(rsW, base__ + A_BYTES + ((2 * wave + (j >> 1)) * 2 + (j & 1)) * 1024, voffW[j], (KT) * 2048); \
  } while (0)

  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);
  const int offA = (wm * TM + l15) * ROW_BYTES;
  const int offW = A_BYTES + (wn * 4) * 2048 + lane * 16;   // block (wn*4 + i)*2 + ks

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // few-tile problems (tall-skinny: LoRA rank projections, small ViT shapes): grid.y slices K; every slice writes an fp32
  // partial [M, N] to the slab, gemm128_splitk_reduce_kernel sums them in slice order and applies the epilogue
  const int nk_all = p.K / BK;
  int kt0 = 0, nk = nk_all;
  if (p.splitk > 1 && p.tail_base < 0) {
    kt0 = (int)(((long)blockIdx.y * nk_all) / p.splitk);
    nk = (int)(((long)(blockIdx.y + 1) * nk_all) / p.splitk);
  }
  BL_STAGE(kt0 & 1, kt0);
  __syncthreads();
  for (int kt = kt0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) BL_STAGE(cur ^ 1, kt + 1);
    const char* base = smem + cur * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cb = (c0 ^ (ks * 4)) << 4;
      bf16x8_t wf[NT], af[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *(const bf16x8_t*)(base + offW + i * 2048 + ks * 1024);
#pragma unroll
      for (int j = 0; j < MT; ++j) af[j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
#undef BL_STAGE
  if (p.splitk > 1 && p.tail_base < 0) {
    float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * TM + j * 16 + l15, n = n0 + wn * TN + i * 16 + lg * 4;
        if (m < p.M && n < p.N) *(f32x4_t*)(slab + (long)m * p.N + n) = acc[i][j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
      epilogue_store4<EPI>(p, m0 + wm * TM + j * 16 + l15, n0 + wn * TN + i * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// tail kernel: BM × BN sub-tiles (128×128, 128×64 or 64×64) of the big kernel's leftover 256×256 tiles, 4 waves (2 × 2),
// NST-stage LDS-DMA ring with NST-1 K-tiles in flight behind a counted vmcnt and ONE barrier per K-step. The leftover
// round has at most one workgroup per CU, so nothing else hides the HBM/L2 latency — the two-stage gemm128 loop spent
// ≈ 3/4 of every K-step waiting there. K order per output = the big kernel's (bit-identical results).
// Stand-alone mode (p.tail_base < 0): the same loop as the main kernel of problems whose 160 × 128 tiles fit ONE round of
// 256 CUs — the narrow ViT layers at 16 images (attn.proj / mlp.fc2 / patch embed: M = 4176 or 4096, N = 1024 or 1152:
// 216 / 234 tiles; 160 divides the ragged M with 1.5–3.4 % waste where 128-row tiles need 264 / 288 > 256 workgroups) —
// where gemm128's one-stage prefetch left a lone workgroup per CU waiting on every K-step (fc2: 65 µs → see DESIGN).
// ======================================================================================================================
template <int EPI, int BM, int BN, int NST, int WM = 2>
__global__ __launch_bounds__(WM * 128) void gemm_tail_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NWAVE = WM * 2, TM = BM / WM, TN = BN / 2, MT = TM / 16, NT = TN / 16;
  constexpr int APW = BM / 8 / NWAVE, WPW = BN / 8 / NWAVE, LPS = APW + WPW;   // LDS-DMA pieces per wave and stage
  constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = A_BYTES + BN * ROW_BYTES;
  constexpr int SUB_N = 256 / BN, SUBS = (256 / BM) * SUB_N;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST × BUF_BYTES

  // workgroups go to XCDs round-robin: give each XCD (private L2) a contiguous run of sub-tiles, i.e. WHOLE leftover tiles —
  // with the plain order the SUBS pieces of a tile land on SUBS different L2s and every one of them fetches the same rows
  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  int m0, n0;
  if (p.tail_base >= 0) {
    int tm, tn;
    lin_to_tile(p, p.tail_base + lin / SUBS, tm, tn);
    const int sub = lin % SUBS;
    m0 = tm * 256 + (sub / SUB_N) * BM, n0 = tn * 256 + (sub % SUB_N) * BN;
  } else {
    // stand-alone mode: the whole problem as BM × BN tiles (p.tiles_n of them per row panel), one tile per workgroup and —
    // the launcher's condition — at most one workgroup per CU. Column tiles of a row panel are neighbours in `lin`, so an
    // XCD's contiguous run re-uses its activation panels from L2.
    m0 = (lin / p.tiles_n) * BM, n0 = (lin % p.tiles_n) * BN;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  static_assert(APW >= 1 && WPW >= 1, "tile too small for the wave count");
  unsigned voffA[APW], voffW[WPW];
#pragma unroll
  for (int j = 0; j < APW; ++j) voffA[j] = (unsigned)(((long)(m0 + (j * NWAVE + wave) * 8 + prow) * p.lda) * 2 + pchunk * 16);
#pragma unroll
  for (int j = 0; j < WPW; ++j) {   // weight block b = j*NWAVE + wave: n-tile b >> 1, k-step b & 1
    const int blk = j * NWAVE + wave;
    voffW[j] = (unsigned)(((long)(n0 / 16 + (blk >> 1)) * kt32 + (blk & 1)) * 1024 + lane * 16);
  }

#define BL_STAGE(BUF, KT)                                                                                   \
  do {                                                                                                      \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                \
    _Pragma("unroll") for (int j = 0; j < APW; ++j) BL_GLDS(rsA, base__ + (j * NWAVE + wave) * 1024, voffA[j], (KT) * 128); \
    _Pragma("unroll") for (int j = 0; j < WPW; ++j)                                                         \
        BL_GLDS(rsW, base__ + A_BYTES + (j * NWAVE + wave) * 1024, voffW[j], (KT) * 2048);                  \
  } while (0)

  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);
  const int offA = (wm * TM + l15) * ROW_BYTES;
  const int offW = A_BYTES + (wn * NT) * 2048 + lane * 16;   // block (wn*NT + i)*2 + ks

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // Branch-free ring: every iteration issues one stage — past the end of K it re-fetches the last K-tile into a slot nobody
  // reads again — so exactly NST-1 stages are in flight at every wait and the loop body has no control flow (with branches
  // between the MFMAs the compiler parks the accumulators in VGPRs and copies them to AGPRs around every MFMA).
  const int nk = p.K / BK;
#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2) BL_STAGE(s2, min(s2, nk - 1));
  static_assert((NST - 2) * LPS <= 63, "counted vmcnt range");
  for (int kt = 0; kt < nk; ++kt) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");   // K-tile kt landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();      // K-tile kt landed for every wave; every wave is done reading K-tile kt-1
    const char* base = smem + (kt % NST) * BUF_BYTES;
    // one wave per SIMD: nothing else covers this wave's issue slots — the next ring slot's LDS-DMA pieces are issued in
    // the shadow of the MFMAs (sched_group_barrier order below)
    char* nbase = smem + ((kt + NST - 1) % NST) * BUF_BYTES;     // the slot of K-tile kt-1
    const int nkt = min(kt + NST - 1, nk - 1);
#pragma unroll
    for (int pc = 0; pc < APW; ++pc) BL_GLDS(rsA, nbase + (pc * NWAVE + wave) * 1024, voffA[pc], nkt * 128);
#pragma unroll
    for (int jw = 0; jw < WPW; ++jw) BL_GLDS(rsW, nbase + A_BYTES + (jw * NWAVE + wave) * 1024, voffW[jw], nkt * 2048);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cb = (c0 ^ (ks * 4)) << 4;
      bf16x8_t wf[NT], af[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *(const bf16x8_t*)(base + offW + i * 2048 + ks * 1024);
#pragma unroll
      for (int j = 0; j < MT; ++j) af[j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    constexpr int NMF = 2 * NT * MT, EVERY = NMF / LPS;          // one piece every EVERY MFMAs
    static_assert(EVERY >= 1, "more DMA pieces than MFMAs");
    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NT + MT), 0);   // all DS reads first
#pragma unroll
    for (int pc = 0; pc < LPS; ++pc) {
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);             // VMEM read (one LDS-DMA piece)
      __builtin_amdgcn_sched_group_barrier(0x008, EVERY, 0);         // MFMA
    }
    __builtin_amdgcn_sched_group_barrier(0x008, NMF - EVERY * LPS, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus stages must not outlive the workgroup's LDS allocation
#undef BL_STAGE
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
      epilogue_store4<EPI>(p, m0 + wm * TM + j * 16 + l15, n0 + wn * TN + i * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// ring kernel with TWO waves per SIMD (round 4): the tail kernel's stand-alone mode (one round of BM × BN = 160 × 128 tiles,
// NST-stage LDS-DMA ring, counted vmcnt, one barrier per K-step) run by 8 waves — 2 along m × 4 along n, 80 × 32 outputs
// each. With one wave per SIMD the wave that issues a ring slot's LDS-DMA pieces (≈ 60 issue cycles per 1-KiB piece) is the
// only one that could issue MFMAs there, so a K-step took pieces + fragment reads + MFMAs back to back (ISA: 9 pieces, 18
// ds_read_b128, lgkmcnt(0), 40 MFMAs: ≈ 0.82 µs for 640 cycles of matrix work). Here every wave carries half the MFMAs and
// half the pieces of its SIMD, the pieces are spread through the wave's MFMAs (sched_barrier-pinned groups), and the
// partner wave's MFMAs issue while this wave sits in a piece's issue. The 20 activation pieces do not divide by 8 waves:
// waves 4–7 issue a third, out-of-range piece (zero-fill, no memory request) into a dump KiB behind the ring, which keeps
// the loop branch-free and the counted vmcnt the same for every wave. K order per output = every other tile kernel's.
// ======================================================================================================================
template <int EPI, int BM, int BN, int NST>
__global__ __launch_bounds__(512) void gemm_ring8_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int NWAVE = 8, TM = BM / 2, TN = BN / 4, MT = TM / 16, NT = TN / 16;      // waves: 2 along m x 4 along n
  constexpr int APIECES = BM / 8, WPIECES = BN / 8;
  constexpr int APW = (APIECES + NWAVE - 1) / NWAVE, WPW = WPIECES / NWAVE, LPS = APW + WPW;
  static_assert(TM % 16 == 0 && TN % 16 == 0 && WPIECES % NWAVE == 0, "tile shape");
  constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = A_BYTES + BN * ROW_BYTES, DUMP = NST * BUF_BYTES;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST × BUF_BYTES + 1 KiB dump

  const int nwg = gridDim.x, xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  int m0, n0;
  if (p.tail_base >= 0) {   // tail mode (as gemm_tail_kernel): SUBS sub-tiles per leftover 256 × 256 tile of the big kernel's grid
    constexpr int SUB_N = 256 / BN, SUBS = (256 / BM) * SUB_N;
    int tm, tn;
    lin_to_tile(p, p.tail_base + lin / SUBS, tm, tn);
    const int sub = lin % SUBS;
    m0 = tm * 256 + (sub / SUB_N) * BM, n0 = tn * 256 + (sub % SUB_N) * BN;
  } else {
    m0 = (lin / p.tiles_n) * BM, n0 = (lin % p.tiles_n) * BN;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[APW], voffW[WPW];
  int ldsA[APW];                       // slot-relative LDS byte offset of this wave's j-th activation piece (wave-uniform)
#pragma unroll
  for (int j = 0; j < APW; ++j) {
    const int pi = j * NWAVE + wave;
    const bool real = pi < APIECES;
    voffA[j] = real ? (unsigned)(((long)(m0 + pi * 8 + prow) * p.lda) * 2 + pchunk * 16) : 0xfffffff0u;
    ldsA[j] = real ? pi * 1024 : -1;
  }
#pragma unroll
  for (int j = 0; j < WPW; ++j) {   // weight block b = j*NWAVE + wave: n-tile b >> 1, k-step b & 1
    const int blk = j * NWAVE + wave;
    voffW[j] = (unsigned)(((long)(n0 / 16 + (blk >> 1)) * kt32 + (blk & 1)) * 1024 + lane * 16);
  }
  // piece `pc` (0 … LPS-1) of this wave for K-tile KT into ring slot BUF
#define BL_PIECE(BUF, KT, PC)                                                                                     \
  do {                                                                                                            \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                      \
    if ((PC) < APW) {                                                                                             \
      const int ja__ = (PC) < APW ? (PC) : 0;                                                                     \
      BL_GLDS(rsA, ldsA[ja__] >= 0 ? base__ + ldsA[ja__] : smem + DUMP, voffA[ja__], (KT) * 128);                 \
    } else {                                                                                                      \
      const int jw__ = (PC) >= APW ? (PC) - APW : 0;                                                              \
      BL_GLDS(rsW, base__ + A_BYTES + (jw__ * NWAVE + wave) * 1024, voffW[jw__], (KT) * 2048);                    \
    }                                                                                                             \
  } while (0)

  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);
  const int offA = (wm * TM + l15) * ROW_BYTES;
  const int offW = A_BYTES + (wn * NT) * 2048 + lane * 16;   // block (wn*NT + i)*2 + ks

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  // grid.y > 1 slices K (few-tile, long-K problems — LoRA's rank projections t = x·Aᵀ / dt = dy·(sB): 37 row tiles, N = 64 … 192,
  // K up to 22016 — which gemm128's one-stage prefetch ran at one global-load latency per K-step, ≈ 19.5 µs per call): every
  // slice writes its fp32 partial [M, N] to the slab, gemm128_splitk_reduce_kernel sums them in slice order + epilogue
  const int nk_all = p.K / BK;
  int ktb = 0, nk = nk_all;
  const bool sliced = p.splitk > 1 && p.tail_base < 0;
  if (sliced) {
    ktb = (int)(((long)blockIdx.y * nk_all) / p.splitk);
    nk = (int)(((long)(blockIdx.y + 1) * nk_all) / p.splitk) - ktb;
  }
#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2) {
    const int kt0 = ktb + min(s2, nk - 1);
#pragma unroll
    for (int pc = 0; pc < LPS; ++pc) BL_PIECE(s2, kt0, pc);
  }
  static_assert((NST - 2) * LPS <= 63 && NST >= 4, "counted vmcnt range / ring depth");
  constexpr int NMF = 2 * NT * MT, EVERY = NMF / LPS;          // one piece every EVERY MFMAs
  constexpr int NRD = 2 * (NT + MT), RPG = (NRD + LPS - 1) / LPS;   // fragment reads per K-tile, per group
  static_assert(EVERY >= 1, "more DMA pieces than MFMAs");
  // Two fragment register sets: the reads of K-tile kt+1 are issued among the MFMAs of K-tile kt (after the barrier that
  // says kt+1 has landed), so no wave ever waits on LDS at the head of a K-step — with one set all eight waves read their
  // 14 fragments right behind the barrier (≈ 450 LDS cycles per CU with no MFMA in flight: 0.58 µs per K-step measured).
  // The loop is unrolled by two K-steps for the set swap; an odd last K-step follows the loop.
  bf16x8_t wf[2][2][NT], af[2][2][MT];
#define BL_READ(SET, BASE, RI)                                                                                   \
  do {                                                                                                           \
    const int ks__ = (RI) / (NT + MT), e__ = (RI) % (NT + MT);                                                   \
    if (e__ < NT) wf[SET][ks__][e__ < NT ? e__ : 0] = *(const bf16x8_t*)((BASE) + offW + e__ * 2048 + ks__ * 1024);   \
    else af[SET][ks__][e__ >= NT ? e__ - NT : 0] =                                                               \
        *(const bf16x8_t*)((BASE) + offA + (e__ - NT) * 16 * ROW_BYTES + ((c0 ^ (ks__ * 4)) << 4));              \
  } while (0)
#define BL_KSTEP(CUR, NXT, KT)                                                                                   \
  do {                                                                                                           \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 3) * LPS) : "memory");   /* K-tile KT+1 landed (my pieces) */ \
    __builtin_amdgcn_s_barrier();   /* … for every wave; every wave has its fragments of K-tile KT-1 and older */ \
    const char* nb__ = smem + (((KT) + 1) % NST) * BUF_BYTES;                                                    \
    const int nslot__ = ((KT) + NST - 1) % NST;                      /* the slot of K-tile KT-1 */               \
    const int nkt__ = ktb + min((KT) + NST - 1, nk - 1);                                                         \
    _Pragma("unroll") for (int g = 0; g < LPS; ++g) {                                                            \
      BL_PIECE(nslot__, nkt__, g);                                                                               \
      _Pragma("unroll") for (int ri = g * RPG; ri < ((g + 1) * RPG < NRD ? (g + 1) * RPG : NRD); ++ri)           \
          BL_READ(NXT, nb__, ri);                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
      _Pragma("unroll") for (int idx = g * EVERY; idx < ((g + 1 < LPS) ? (g + 1) * EVERY : NMF); ++idx) {        \
        const int ks = idx / (NT * MT), i = (idx / MT) % NT, j = idx % MT;                                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[CUR][ks][i], af[CUR][ks][j], acc[i][j], 0, 0, 0); \
      }                                                                                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                         \
    }                                                                                                            \
  } while (0)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * LPS) : "memory");   // K-tile 0 landed (this wave's pieces)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ri = 0; ri < NRD; ++ri) BL_READ(0, smem, ri);
  for (int kt = 0; kt + 1 < nk; kt += 2) {
    BL_KSTEP(0, 1, kt);
    BL_KSTEP(1, 0, kt + 1);
  }
  if (nk & 1) BL_KSTEP(0, 1, nk - 1);    // odd K-tile count (LoRA's K + rank columns): one more step from set 0, outside the loop
#undef BL_KSTEP
#undef BL_READ
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the surplus stages must not outlive the workgroup's LDS allocation
#undef BL_PIECE
  if (sliced) {
    float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int m = m0 + wm * TM + j * 16 + l15, n = n0 + wn * TN + i * 16 + lg * 4;
        if (m < p.M && n < p.N) *(f32x4_t*)(slab + (long)m * p.N + n) = acc[i][j];
      }
    return;
  }
  int ncol[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) ncol[i] = n0 + wn * TN + i * 16 + lg * 4;
  epilogue_tile<EPI, NT, MT>(p, m0 + wm * TM + l15, ncol, m0 + BM, n0 + BN, acc);
#endif
}

// out(m, n..n+3) = epilogue(Σ_slices slab[slice][m][n..n+3]) for the split-K form of the 128 kernel
template <int EPI>
__global__ __launch_bounds__(256) void gemm128_splitk_reduce_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int n4 = p.N >> 2;
  const long total = (long)p.M * n4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    f32x4_t sum = *(const f32x4_t*)(p.slab + (long)m * p.N + n);
    for (int s2 = 1; s2 < p.splitk; ++s2) sum += *(const f32x4_t*)(p.slab + ((long)s2 * p.M + m) * p.N + n);
    epilogue_store4<EPI>(p, m, n, sum);
  }
#endif
}

// out(m, n..n+3) = epilogue(tree of the S = 2 / 4 / 8 slab partials) for gemm_mid_kernel<SK>: the upper levels of the
// skinny order's balanced binary tree (each partial is already the tree value of an aligned group of 8/S slices)
template <int EPI>
__global__ __launch_bounds__(256) void gemm_rows_tree_reduce_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int n4 = p.N >> 2, S = p.splitk;
  const long total = (long)p.M * n4, stride = (long)p.M * p.N;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    const float* q = p.slab + (long)m * p.N + n;
    f32x4_t v[8];
    for (int s2 = 0; s2 < S; ++s2) v[s2] = *(const f32x4_t*)(q + s2 * stride);
    for (int w = 1; w < S; w *= 2)
      for (int s2 = 0; s2 < S; s2 += 2 * w) v[s2] = v[s2] + v[s2 + w];
    epilogue_store4<EPI>(p, m, n, v[0]);
  }
#endif
}

// ======================================================================================================================
// "mid" kernel: ALL M ≤ 64·MB rows × (16·NB) columns per workgroup (batch-1 prefill S = 288, one image's ViT tokens, the
// projector at one image): every weight byte leaves HBM exactly once per GEMM — these shapes are bound by the weight
// stream, not by MFMA — while the activation K-slabs (M × 128 B per K-step) are re-read from L2 by every workgroup.
// 4 waves, wave w owns rows [16·MB·w, 16·MB·(w+1)) × all NB column blocks; 3-stage LDS ring (≤ 144 KiB), two K-tiles of
// LDS-DMA in flight behind a counted vmcnt. NB = 4 (64-column slabs) for wide layers; NB = 1 (16-column slabs) when 64
// would leave most CUs without a workgroup (N = 4096 → 256 workgroups instead of 64). The K order of every output is the
// tile kernels' (K-tile by K-tile, two 32-steps each), so results are bit-identical to theirs — a sequence's result does
// not depend on the batch size it was run in. grid.y may slice K when the caller provides a workspace (opt-in).
// ======================================================================================================================
//
// SK > 0 ("skinny order", bl_gemm_skinny_rows_bf16): the fp32 summation order of gemm_skinny_kernel instead of the tile
// kernels' — K is cut into 8 slices of p.fold_ks MFMA k-steps; each slice is accumulated from zero in k order; the slice
// partials are combined as a balanced binary tree ((p0+p1)+(p2+p3))+((p4+p5)+(p6+p7)). A workgroup walks SK consecutive
// slices (SK = 8: all of K; SK = 2 with grid.y = 4 for the narrow layers, whose 64 column slabs alone would leave 3/4 of
// the CUs idle) and folds finished slices into a binary-counter stack of held partial sums (log2 SK accumulator sets),
// so an aligned group of slices yields exactly its subtree's value; grid.y > 1 writes that value as an fp32 partial and
// gemm_rows_tree_reduce_kernel finishes the tree. Every row's result is bit-identical to what the weight-streaming
// kernel gives that row in a batch of <= 16, whatever the grid (the merged decode iteration of StaggeredDecodePipeline
// stacks 6 such batches and streams the weights once).
template <int EPI, int MB, int NB, int SK = 0>
__global__ __launch_bounds__(256, 2) void gemm_mid_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 64 * MB, BN = 16 * NB, NWAVE = 4, NST = 3;
  constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = A_BYTES + NB * 2048;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST × BUF_BYTES
  const int n0 = blockIdx.x * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  // activation pieces: 8·MB × (8 rows × 128 B); wave takes pieces j*4 + wave. lane → (row = lane>>3, chunk = lane&7)
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[2 * MB];
#pragma unroll
  for (int j = 0; j < 2 * MB; ++j) voffA[j] = (unsigned)(((long)((j * NWAVE + wave) * 8 + prow) * p.lda) * 2 + pchunk * 16);
  // weight blocks (1 KiB each): NB n-tiles × 2 k-steps. NB = 4: wave w stages n-tile w (two pieces); NB = 1: waves 0 / 1
  // stage k-step 0 / 1 of the single n-tile, waves 2 / 3 none (their counted waits are one piece shorter)
  constexpr int WP = NB == 4 ? 2 : 1;
  const bool has_w = NB == 4 || wave < 2;
  unsigned voffW[WP];
  int ldsW[WP];
#pragma unroll
  for (int q = 0; q < WP; ++q) {
    const int nt = NB == 4 ? wave : 0, ks = NB == 4 ? q : (wave & 1);
    voffW[q] = (unsigned)(((long)(n0 / 16 + nt) * kt32 + ks) * 1024 + lane * 16);
    ldsW[q] = A_BYTES + (nt * 2 + ks) * 1024;
  }

#define BL_STAGE(BUF, KT)                                                                                   \
  do {                                                                                                      \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                \
    _Pragma("unroll") for (int j = 0; j < 2 * MB; ++j) BL_GLDS(rsA, base__ + (j * NWAVE + wave) * 1024, voffA[j], (KT) * 128); \
    if (has_w) {                                                                                            \
      _Pragma("unroll") for (int q = 0; q < WP; ++q) BL_GLDS(rsW, base__ + ldsW[q], voffW[q], (KT) * 2048); \
    }                                                                                                       \
  } while (0)
#define BL_WAIT_TILES(T)                                                                                    \
  do {                                                                                                      \
    if (has_w) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((T) * (2 * MB + WP)) : "memory");                   \
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((T) * (2 * MB)) : "memory");                              \
  } while (0)

  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);
  const int offA = (wave * 16 * MB + l15) * ROW_BYTES;
  const int offW = A_BYTES + lane * 16;                 // block i*2 + ks

  f32x4_t acc[NB][MB];
  constexpr int LV = SK >= 8 ? 3 : SK >= 4 ? 2 : SK >= 2 ? 1 : 0;   // levels of held partial sums
  f32x4_t hold[LV ? LV : 1][SK ? NB : 1][SK ? MB : 1];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int fold_next = p.fold_ks, sl = 0;        // SK: k-step count at which the current slice ends; local slice index

  const int nk_all = p.K / BK;
  int kt0 = 0, nk = nk_all;
  if constexpr (SK > 0) {                   // slices [blockIdx.y·SK, +SK): SK even → the range starts on a K-tile
    kt0 = (int)blockIdx.y * SK * p.fold_ks / 2;
    nk = kt0 + SK * p.fold_ks / 2;
  } else if (p.splitk > 1) {
    kt0 = (int)(((long)blockIdx.y * nk_all) / p.splitk);
    nk = (int)(((long)(blockIdx.y + 1) * nk_all) / p.splitk);
  }
  BL_STAGE(0, kt0);
  if (kt0 + 1 < nk) BL_STAGE(1, kt0 + 1);
  for (int kt = kt0; kt < nk; ++kt) {
    const int it = kt - kt0;
    if (kt + 2 < nk) {                 // the ring slot read in the previous iteration (all waves are past its barrier)
      BL_STAGE((it + 2) % NST, kt + 2);
      BL_WAIT_TILES(2);
    } else if (kt + 1 < nk) {
      BL_WAIT_TILES(1);
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();      // every wave's pieces of K-tile kt have landed
    const char* base = smem + (it % NST) * BUF_BYTES;
    if constexpr (SK == 0) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int cb = (c0 ^ (ks * 4)) << 4;
        bf16x8_t wf[NB], af[MB];
#pragma unroll
        for (int i = 0; i < NB; ++i) wf[i] = *(const bf16x8_t*)(base + offW + (i * 2 + ks) * 1024);
#pragma unroll
        for (int j = 0; j < MB; ++j) af[j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < MB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
      }
    } else {
      // both k-steps' fragments are read before the first MFMA: the slice-boundary branch between the k-steps would
      // otherwise keep the second k-step's LDS reads from overlapping the first one's MFMAs
      bf16x8_t wf[2][NB], af[2][MB];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int cb = (c0 ^ (ks * 4)) << 4;
#pragma unroll
        for (int i = 0; i < NB; ++i) wf[ks][i] = *(const bf16x8_t*)(base + offW + (i * 2 + ks) * 1024);
#pragma unroll
        for (int j = 0; j < MB; ++j) af[ks][j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
          for (int j = 0; j < MB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][i], af[ks][j], acc[i][j], 0, 0, 0);
        if (it * 2 + ks + 1 == fold_next) {       // slice boundary (may fall between the two k-steps of a K-tile)
          // binary counter over the local slice index sl (wave-uniform → scalar branches): a finished slice is parked at
          // level 0 or merged upwards while the levels below are occupied; after the group's last slice acc holds the
          // tree value of the whole group
          bool parked = false;
          // opaque to the optimiser: without this hipcc speculates the level-0 add (hold + acc) on EVERY k-step, a VALU
          // read of the accumulators that drains the MFMA pipeline each time (2x slower)
#pragma unroll
          for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < MB; ++j) asm volatile("" : "+v"(acc[i][j]));
#define BL_FOLD_LEVEL(L)                                                                   \
  if constexpr (LV > L) {                                                                   \
    if (!parked) {                                                                          \
      if (!((sl >> L) & 1)) {                                                               \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                      \
          _Pragma("unroll") for (int j = 0; j < MB; ++j) hold[L][i][j] = acc[i][j];         \
        parked = true;                                                                      \
      } else {                                                                              \
        _Pragma("unroll") for (int i = 0; i < NB; ++i)                                      \
          _Pragma("unroll") for (int j = 0; j < MB; ++j) acc[i][j] = hold[L][i][j] + acc[i][j]; \
      }                                                                                     \
    }                                                                                       \
  }
          BL_FOLD_LEVEL(0) BL_FOLD_LEVEL(1) BL_FOLD_LEVEL(2)
#undef BL_FOLD_LEVEL
          if (parked) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
#pragma unroll
              for (int j = 0; j < MB; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
          }
          fold_next += p.fold_ks;
          ++sl;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // all reads of this slot done before it is re-issued next iteration
  }
#undef BL_STAGE
#undef BL_WAIT_TILES
  if constexpr (SK > 0) {
    float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const int m = wave * 16 * MB + j * 16 + l15, n = n0 + i * 16 + lg * 4;
        if (gridDim.y == 1) epilogue_store4<EPI>(p, m, n, acc[i][j]);
        else if (m < p.M && n < p.N) *(f32x4_t*)(slab + (long)m * p.N + n) = acc[i][j];
      }
    return;
  }
  if (p.splitk > 1) {
    float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
      for (int j = 0; j < MB; ++j) {
        const int m = wave * 16 * MB + j * 16 + l15, n = n0 + i * 16 + lg * 4;
        if (m < p.M && n < p.N) *(f32x4_t*)(slab + (long)m * p.N + n) = acc[i][j];
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int j = 0; j < MB; ++j)
      epilogue_store4<EPI>(p, wave * 16 * MB + j * 16 + l15, n0 + i * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// "rows-stream" kernel: the skinny order at M ≤ 96 rows (the merged decode iteration) as a WEIGHT-STREAMING kernel for the
// wide layers. Every wave owns one 16-column weight tile and walks the workgroup's K range; weight fragments go
// HBM → VGPR with non-temporal 16-byte loads straight out of the fragment-major packing (never through LDS) into a ring
// of NWB register buffers of GS k-steps: NWB − 1 groups (12 KiB per wave) are in flight while one is consumed. The ≤ 96
// activation rows of a GS-k-step chunk are staged ONCE per workgroup by LDS-DMA (the mid kernel's swizzled 128-byte rows,
// a ring of NWB chunks) and read by every wave as 6 B-fragments per weight fragment: 0.75 – 1.0 staged bytes per weight
// byte instead of the mid kernel's 3 — what a CU can pull through its vector-memory path is capped (≈ 64 KiB of requests
// outstanding: ≈ 57 GB/s per CU out of HBM, tools/micro/hbm_stream.hip), so staged activation bytes cost weight bytes.
// One raw barrier per chunk; counted vmcnt: every wave issues exactly OPS = XP + GS vector-memory operations per chunk
// on every path (surplus chunks re-read the last one), so vmcnt(2·OPS) always leaves exactly the two younger chunks in
// flight and hipcc's own counting for the weight registers comes out the same. The refill of the slot freed by the
// previous chunk is issued a quarter at a time BEHIND each k-step's MFMAs: a load that stalls at issue because the CU's
// request queue is full then stalls under MFMAs that are already in the pipe (−5 % against issuing it in one block).
// Slices, the binary-counter fold and the grid.y split are gemm_mid_kernel<SK>'s: results are bit-identical to it and to
// gemm_skinny_kernel. fold_ks must be a multiple of GS (K a multiple of 1024: slices end on chunk boundaries, the
// slice-boundary test runs once per chunk) — the launcher sends everything else to the mid kernel.
// ======================================================================================================================
template <int EPI, int NWV, int SK>
__global__ __launch_bounds__(NWV * 64) void gemm_rows_stream_kernel(GemmArgs p, int n_tiles) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int MR = 6, GS = 4, NWB = 4;
  constexpr int KT_BYTES = MR * 16 * ROW_BYTES, XBUF = 2 * KT_BYTES;   // one K-tile (2 k-steps) of all rows; one chunk
  constexpr int XP = 24 / NWV, OPS = XP + GS;
  static_assert(24 % NWV == 0 && XP <= GS, "the chunk's 24 LDS-DMA pieces are dealt evenly to the waves, one per k-step");
  constexpr int LV = SK >= 8 ? 3 : SK >= 4 ? 2 : SK >= 2 ? 1 : 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // NWB × XBUF
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int tile_raw = (int)blockIdx.x * NWV + wave;
  const bool live = tile_raw < n_tiles;
  const int tile = live ? tile_raw : n_tiles - 1;        // a surplus wave streams the last tile again and stores nothing
  const int kt32 = p.K >> 5;
  const int ks0 = (int)blockIdx.y * SK * p.fold_ks, nks = SK * p.fold_ks;   // this workgroup's k-steps [ks0, ks0 + nks)
  const int NC = (nks + GS - 1) / GS;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  // activation pieces (8 rows × 128 B): piece q = j·NWV + wave → K-tile q / 12 of the chunk, row block q % 12
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[XP];
  int ldsA[XP];
#pragma unroll
  for (int j = 0; j < XP; ++j) {
    const int q = j * NWV + wave, kt = q / 12, rb = q - kt * 12;
    voffA[j] = (unsigned)(((long)(rb * 8 + prow) * p.lda) * 2 + kt * 128 + pchunk * 16);
    ldsA[j] = kt * KT_BYTES + rb * 1024;
  }
  const u32x4_t* wbase = (const u32x4_t*)p.W + ((long)tile * kt32 + ks0) * 64 + lane;
  u32x4_t wbuf[NWB][GS];

  // part J (of GS) of chunk C into ring slot SLOT: LDS-DMA piece J (if the wave has one) and weight fragment J. Chunk
  // and k-step indices are clamped, so every call issues the same operations whatever C is.
#define BL_RS_ISSUE_PART(SLOT, C, J)                                                                          \
  do {                                                                                                        \
    const int cc__ = min((C), NC - 1);                                                                        \
    if ((J) < XP) BL_GLDS(rsA, smem + (SLOT) * XBUF + ldsA[(J) < XP ? (J) : 0], voffA[(J) < XP ? (J) : 0], (ks0 + cc__ * GS) * 64); \
    wbuf[SLOT][J] = __builtin_nontemporal_load(wbase + (long)min(cc__ * GS + (J), nks - 1) * 64);             \
  } while (0)
#define BL_RS_READX(DST, XB, S)                                                                               \
  do {                                                                                                        \
    const char* xs__ = (XB) + ((S) >> 1) * KT_BYTES + offA + ((c0 ^ (((S) & 1) * 4)) << 4);                    \
    _Pragma("unroll") for (int j = 0; j < MR; ++j) DST[j] = *(const bf16x8_t*)(xs__ + j * 16 * ROW_BYTES);    \
  } while (0)
#define BL_RS_FOLD_LEVEL(L)                                                                \
  if constexpr (LV > L) {                                                                   \
    if (!parked) {                                                                          \
      if (!((sl >> L) & 1)) {                                                               \
        _Pragma("unroll") for (int j = 0; j < MR; ++j) hold[L][j] = acc[j];                 \
        parked = true;                                                                      \
      } else {                                                                              \
        _Pragma("unroll") for (int j = 0; j < MR; ++j) acc[j] = hold[L][j] + acc[j];        \
      }                                                                                     \
    }                                                                                       \
  }
  // one chunk: counted wait, barrier (every wave's pieces of chunk C have landed; every wave is done with chunk C − 1, whose
  // ring slot the refill below overwrites), then GS × (next k-step's fragments LDS → VGPR, 6 MFMAs, a quarter of the refill)
#define BL_RS_CHUNK(SLOT, C)                                                                                  \
  {                                                                                                           \
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * OPS) : "memory");                                            \
    __builtin_amdgcn_s_barrier();                                                                             \
    const char* xb = smem + (SLOT) * XBUF;                                                                    \
    bf16x8_t xf[2][MR];                                                                                       \
    BL_RS_READX(xf[0], xb, 0);                                                                                \
    _Pragma("unroll") for (int s2 = 0; s2 < GS; ++s2) {                                                       \
      if (s2 + 1 < GS) BL_RS_READX(xf[(s2 + 1) & 1], xb, s2 + 1);                                             \
      __builtin_amdgcn_sched_barrier(0);   /* next k-step's fragments are in flight under this one's MFMAs */ \
      const int kidx = (C) * GS + s2;                                                                         \
      if (kidx < nks) {                                                                                       \
        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wbuf[SLOT][s2]);                                     \
        _Pragma("unroll") for (int j = 0; j < MR; ++j)                                                        \
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s2 & 1][j], acc[j], 0, 0, 0);               \
        if (SK > 1 && s2 == GS - 1 && kidx + 1 == fold_next) {   /* slices end on chunk boundaries */        \
          bool parked = false;                                                                                \
          /* opaque to the optimiser, as in gemm_mid_kernel: no speculated VALU read of the accumulators */    \
          _Pragma("unroll") for (int j = 0; j < MR; ++j) asm volatile("" : "+v"(acc[j]));                     \
          BL_RS_FOLD_LEVEL(0) BL_RS_FOLD_LEVEL(1) BL_RS_FOLD_LEVEL(2)                                         \
          if (parked) {                                                                                       \
            _Pragma("unroll") for (int j = 0; j < MR; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};            \
          }                                                                                                   \
          fold_next += p.fold_ks;                                                                             \
          ++sl;                                                                                               \
        }                                                                                                     \
      }                                                                                                       \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
      BL_RS_ISSUE_PART(((SLOT) + NWB - 1) % NWB, (C) + NWB - 1, s2);                                          \
      __builtin_amdgcn_sched_barrier(0);                                                                      \
    }                                                                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                        \
  }

  const int c0 = lg ^ (lane & 7);
  const int offA = l15 * ROW_BYTES;
  f32x4_t acc[MR], hold[LV ? LV : 1][MR];
#pragma unroll
  for (int j = 0; j < MR; ++j) acc[j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  int fold_next = p.fold_ks, sl = 0;

  // prologue: chunks 0 … NWB-2 in flight, chunk by chunk (the counted waits rely on whole chunks being issued in order)
#pragma unroll
  for (int c = 0; c < NWB - 1; ++c) {
#pragma unroll
    for (int j = 0; j < GS; ++j) BL_RS_ISSUE_PART(c, c, j);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int cb = 0; cb < NC; cb += NWB) {   // whole rounds of NWB chunks; chunks past NC only skip their MFMAs
    BL_RS_CHUNK(0, cb)
    BL_RS_CHUNK(1, cb + 1)
    BL_RS_CHUNK(2, cb + 2)
    BL_RS_CHUNK(3, cb + 3)
  }
#undef BL_RS_CHUNK
#undef BL_RS_FOLD_LEVEL
#undef BL_RS_READX
#undef BL_RS_ISSUE_PART
  if (!live) return;
  float* slab = p.slab + (long)blockIdx.y * p.M * p.N;
#pragma unroll
  for (int j = 0; j < MR; ++j) {
    const int m = j * 16 + l15, n = tile * 16 + lg * 4;
    if (gridDim.y == 1) epilogue_store4<EPI>(p, m, n, acc[j]);
    else if (m < p.M) *(f32x4_t*)(slab + (long)m * p.N + n) = acc[j];
  }
#endif
}

// ======================================================================================================================
// "mid2" kernel: 160 rows × 32 columns per workgroup, 4 waves as 2 (m) × 2 (n), 3-stage ring — the narrow layers
// (N = 4096 o / down, ViT proj / fc2) at 128 < M ≤ 320. The mid kernel's 16-column slabs re-stage ALL rows of A per
// workgroup (42 KB per K-step for 0.66 MFLOP), and the CU's global→LDS path (one 1-KiB piece per ≈ 20 cycles) is what
// bounds it; half the rows and twice the columns stage 24 KB for 0.66 MFLOP with as many workgroups. Same K order per
// output as every other kernel (bit-identical results).
// ======================================================================================================================
template <int EPI, int NT, int NST = 3>   // NT = 16-column tiles per wave: 32·NT columns per workgroup; NST ring stages
__global__ __launch_bounds__(256) void gemm_mid2_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 160, BN = 32 * NT, MT = 5, LPS = 5 + NT;   // LPS: LDS-DMA pieces per wave and stage (5 A + NT W)
  constexpr int A_BYTES = BM * ROW_BYTES, BUF_BYTES = A_BYTES + NT * 4096;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // NST × BUF_BYTES
  const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int kt32 = p.K >> 5;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);

  // activation pieces: 20 × (8 rows × 128 B); wave takes pieces j*4 + wave. weight blocks: (n-tile, k-step) = (wave >> 1, wave & 1)
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) voffA[j] = (unsigned)(((long)(m0 + (j * 4 + wave) * 8 + prow) * p.lda) * 2 + pchunk * 16);
  unsigned voffW[NT];                            // weight block b = j*4 + wave: n-tile b >> 1, k-step b & 1, LDS slot b
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int blk = j * 4 + wave;
    voffW[j] = (unsigned)(((long)(n0 / 16 + (blk >> 1)) * kt32 + (blk & 1)) * 1024 + lane * 16);
  }

#define BL_STAGE(BUF, KT)                                                                                   \
  do {                                                                                                      \
    char* base__ = smem + (BUF) * BUF_BYTES;                                                                \
    _Pragma("unroll") for (int j = 0; j < MT; ++j) BL_GLDS(rsA, base__ + (j * 4 + wave) * 1024, voffA[j], (KT) * 128); \
    _Pragma("unroll") for (int j = 0; j < NT; ++j) BL_GLDS(rsW, base__ + A_BYTES + (j * 4 + wave) * 1024, voffW[j], (KT) * 2048); \
  } while (0)

  const int l15 = lane & 15, lg = lane >> 4;
  const int c0 = lg ^ (lane & 7);
  const int offA = (wm * 80 + l15) * ROW_BYTES;
  const int offW = A_BYTES + (wn * NT) * 2048 + lane * 16;   // block (wn*NT + i)*2 + ks

  f32x4_t acc[NT][MT];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  static_assert(NST == 2 || NST == 3, "ring depth");
  BL_STAGE(0, 0);
  if (NST == 3 && 1 < nk) BL_STAGE(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // issue K-tile kt+NST-1 into the ring slot read in the previous iteration (all waves are past its barrier), then wait
    // for K-tile kt with the newer ones still in flight
    if (kt + NST - 1 < nk) BL_STAGE((kt + NST - 1) % NST, kt + NST - 1);
    const int later = min(NST - 1, nk - 1 - kt);
    if (later == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LPS) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // every wave's pieces of K-tile kt have landed
    const char* base = smem + (kt % NST) * BUF_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int cb = (c0 ^ (ks * 4)) << 4;
      bf16x8_t wf[NT], af[MT];
#pragma unroll
      for (int i = 0; i < NT; ++i) wf[i] = *(const bf16x8_t*)(base + offW + i * 2048 + ks * 1024);
#pragma unroll
      for (int j = 0; j < MT; ++j) af[j] = *(const bf16x8_t*)(base + offA + j * 16 * ROW_BYTES + cb);
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // all reads of this slot done before it is re-issued next iteration
  }
#undef BL_STAGE
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j)
      epilogue_store4<EPI>(p, m0 + wm * 80 + j * 16 + l15, n0 + (wn * NT + i) * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// 288 × 256 tile — one 288-token sequence (S = 1 + 256 + 31) per row tile.
// At the bench shape M = 16 · 288 = 4608 the 256-row tiles come out as 18 row tiles: Llama o_proj / down_proj get 288 tiles
// = one round + a 32-tile tail on sub-tiles at a third of the efficiency (+ 45 % time for 12.5 % of the work), qkv gets
// 864 = 3.375 rounds (a quarter-filled fourth round). With 288 rows per tile the same GEMMs are exactly 256 tiles
// (o / down) and 768 = 3 rounds (qkv): no tail, no partial round.
// 8 waves as 2 (m) × 4 (n): a wave owns 144 rows × 64 columns = 9 × 4 accumulator tiles (144 VGPRs); per K-tile six
// phases (row third t ∈ 0..2 × column half h ∈ 0..1, 12 MFMAs each, order (0,0) (0,1) (1,1) (1,0) (2,0) (2,1)) on two
// fragment register sets per operand, so a phase's MFMAs run over the next phase's ds_reads. LDS: 2 stages × 72 KiB
// (288 rows × 128 B + 4 KiB landing area for the dummy pieces + 32 KiB weights). A wave issues 9 LDS-DMA pieces per
// K-tile (5 activation — 36 real ones over 8 waves, the 4 surplus ones hit a zero-record descriptor: no traffic — and
// 4 weight) for K-tile t+1 during phases 6 (of t-1), 1, 2, 3 (of t); ONE `vmcnt(0)` + barrier per K-tile, at the end of
// phase 5, two phases after the last issue: it publishes K-tile t+1 (first read in phase 6) and retires every read of
// K-tile t-1's stage before phase 6 re-fills it. No other barrier: the two waves of a SIMD drift apart inside a
// K-tile and overlap one's LDS reads / DMA issue with the other's MFMAs.
// K order per output = every other kernel's (K-tile by K-tile, two 32-steps each): bit-identical results.
// ======================================================================================================================
template <int EPI>
__global__ __launch_bounds__(512) void gemm288_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 288, BN = 256;
  constexpr int A_BYTES = BM * ROW_BYTES, DUMMY = A_BYTES, W_OFF = A_BYTES + 4096, STAGE = W_OFF + 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 × STAGE
  int tm, tn;
  tile_coords(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int kt32 = p.K >> 5, nk = p.K / BK;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, w_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);

  // activation pieces (8 rows × 128 B): wave takes pieces wave, wave + 8, …, wave + 32; pieces ≥ 36 do not exist
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[5];
  int ldsA[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int pi = wave + 8 * j;
    voffA[j] = (unsigned)(((long)(m0 + pi * 8 + prow) * p.lda) * 2 + pchunk * 16);
    ldsA[j] = pi < 36 ? pi * 1024 : DUMMY + (pi - 36) * 1024;
  }
  const bool a4_real = wave < 4;                       // piece wave + 32 exists only for waves 0..3
  // weight pieces: n-tiles 2·wave and 2·wave + 1, both k-steps (2 KiB contiguous per n-tile and K-tile)
  unsigned voffW[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) voffW[q] = (unsigned)((long)(n0 / 16 + 2 * wave + q) * kt32 * 1024 + lane * 16);
  const int ldsW = W_OFF + 2 * wave * 2048;

#define ISSUE_A(J, TILE)                                                                                  \
  do {                                                                                                    \
    const int t__ = (TILE);                                                                               \
    /* no branch here: control flow inside the K loop makes hipcc shuttle the accumulators between VGPRs and AGPRs   \
       around every MFMA (measured 1.8x slower); the surplus fifth piece of waves 4..7 goes to a zero-record descriptor */ \
    const __amdgpu_buffer_rsrc_t rs__ = (t__ < nk && ((J) < 4 || a4_real)) ? rsA : rsA0;                  \
    BL_GLDS(rs__, smem + (t__ & 1) * STAGE + ldsA[J], voffA[J], t__ * 128);                               \
  } while (0)
#define ISSUE_W(Q, KS, TILE)                                                                              \
  do {                                                                                                    \
    const int t__ = (TILE);                                                                               \
    const __amdgpu_buffer_rsrc_t rs__ = t__ < nk ? rsW : rsW0;                                            \
    BL_GLDS(rs__, smem + (t__ & 1) * STAGE + ldsW + (Q) * 2048 + (KS) * 1024, voffW[Q] + (KS) * 1024, t__ * 2048); \
  } while (0)

  const int cb0 = (lg ^ (lane & 7)) << 4;                       // swizzled chunk of k-step 0; k-step 1 = cb0 ^ 64
  const int offX = (wm * 144 + l15) * ROW_BYTES;                // + (3·t + i)·2048
  const int offY = W_OFF + wn * 8192 + lane * 16;               // + (2·h + j)·2048 + ks·1024
#define READ_X(DST, T3, SB)                                                                               \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                         \
    DST[i * 2] = *(const bf16x8_t*)((SB) + offX + (3 * (T3) + i) * 2048 + cb0);                           \
    DST[i * 2 + 1] = *(const bf16x8_t*)((SB) + offX + (3 * (T3) + i) * 2048 + (cb0 ^ 64));                \
  }
#define READ_Y(DST, NH, SB)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
    DST[j * 2] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048);                                 \
    DST[j * 2 + 1] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048 + 1024);                      \
  }
#define MMA(XR, YR, T3, NH)                                                                               \
  do {                                                                                                    \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                     \
          acc[2 * (NH) + j][3 * (T3) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
              YR[j * 2 + ks], XR[i * 2 + ks], acc[2 * (NH) + j][3 * (T3) + i], 0, 0, 0);                  \
  } while (0)
#define PHASE_SYNC()                                                                                      \
  do {                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                           \
    __builtin_amdgcn_s_barrier();                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
  // one K-tile: XC holds row third 0 and YC column half 0 of stage SC (K-tile T) on entry; YO takes half 1 and both stay
  // for the whole K-tile, the thirds alternate between XC and XO; phase 6 loads third 0 / half 0 of K-tile T+1 (stage SN)
  // into XO / YC, so the next K-tile runs with the X sets swapped
  // (measured and rejected on this loop: s_setprio(1) around the MFMAs −8 %; sched_group_barrier placement of reads / pieces
  // −5 % and 8 spilled registers; a wave-uniform branch around the surplus piece −77 % — control flow inside the K loop makes
  // hipcc shuttle the accumulators through AGPRs around every MFMA. The compiler's own interleave below is the fastest.)
#define KTILE(XC, XO, YC, YO, SC, SN, T)                                                                  \
  do {                                                                                                    \
    ISSUE_A(2, (T) + 1); ISSUE_A(3, (T) + 1); READ_Y(YO, 1, SC);                 MMA(XC, YC, 0, 0);       \
    ISSUE_A(4, (T) + 1); ISSUE_W(0, 0, (T) + 1); READ_X(XO, 1, SC);              MMA(XC, YO, 0, 1);       \
    ISSUE_W(0, 1, (T) + 1); ISSUE_W(1, 0, (T) + 1); ISSUE_W(1, 1, (T) + 1);      MMA(XO, YO, 1, 1);       \
    READ_X(XC, 2, SC);                                                           MMA(XO, YC, 1, 0);       \
                                                                                 MMA(XC, YC, 2, 0);       \
    PHASE_SYNC();                                                                                         \
    ISSUE_A(0, (T) + 2); ISSUE_A(1, (T) + 2); READ_X(XO, 0, SN); READ_Y(YC, 0, SN); MMA(XC, YO, 2, 1);    \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)

  f32x4_t acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t Xa[6], Xb[6], Ya[4], Yb[4];
  char* const S0 = smem;
  char* const S1 = smem + STAGE;

  // prologue: all of K-tile 0, then the first two activation pieces of K-tile 1 (what phase 6 of the previous tile issues)
  ISSUE_A(0, 0); ISSUE_A(1, 0); ISSUE_A(2, 0); ISSUE_A(3, 0); ISSUE_A(4, 0);
  ISSUE_W(0, 0, 0); ISSUE_W(0, 1, 0); ISSUE_W(1, 0, 0); ISSUE_W(1, 1, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  ISSUE_A(0, 1); ISSUE_A(1, 1);
  READ_X(Xa, 0, S0);
  READ_Y(Ya, 0, S0);
  for (int t = 0; t < nk; t += 2) {        // nk is even (launcher: K % 128 == 0)
    KTILE(Xa, Xb, Ya, Yb, S0, S1, t);
    KTILE(Xb, Xa, Ya, Yb, S1, S0, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the out-of-range issues past the last K-tile
#undef ISSUE_A
#undef ISSUE_W
#undef READ_X
#undef READ_Y
#undef MMA
#undef PHASE_SYNC
#undef KTILE
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j)
      epilogue_store4<EPI>(p, m0 + wm * 144 + j * 16 + l15, n0 + wn * 64 + i * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// 288 × 256 tile, staggered wave groups (the structure of gemm256s_kernel on the 288-row tile): the two 4-wave groups that
// share each SIMD alternate roles segment by segment — one runs a 12-MFMA block while the other issues its LDS-DMA pieces
// and reads the fragments of its next block — with one raw s_barrier per segment. A single X register set suffices (a
// group's fragment reads happen while its MFMAs are not running): 9 × 4 accumulator tiles (144 VGPRs) + 6 + 4 + 4 fragments.
// K-tile T+1's nine pieces per wave are issued over the six slots p6(T-1), p1 … p5(T) (2, 1, 2, 1, 2, 1); slot p6(T) issues
// the first two pieces of T+2 and waits `vmcnt(2)`: everything older — all of T+1 — has landed, two barriers before its
// first read (slot p1 of T+1), and the DMA stream never drains. Stage T is re-filled from slot p6(T) on; its last reads
// are in slot p5(T) of either group, one barrier earlier.
// ======================================================================================================================
template <int EPI>
__global__ __launch_bounds__(512) void gemm288s_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 288, BN = 256;
  constexpr int A_BYTES = BM * ROW_BYTES, DUMMY = A_BYTES, W_OFF = A_BYTES + 4096, STAGE = W_OFF + 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 × STAGE
  int tm, tn;
  tile_coords(p, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int kt32 = p.K >> 5, nk = p.K / BK;

  const unsigned a_bytes = (unsigned)min((long)p.M * p.lda * 2, 0xffffffffL);
  const unsigned w_bytes = (unsigned)min((long)p.N * p.K * 2, 0xffffffffL);
  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffA[5];
  int ldsA[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int pi = wave + 8 * j;
    voffA[j] = (unsigned)(((long)(m0 + pi * 8 + prow) * p.lda) * 2 + pchunk * 16);
    ldsA[j] = pi < 36 ? pi * 1024 : DUMMY + (pi - 36) * 1024;
  }
  const unsigned a4_bytes = wave < 4 ? a_bytes : 0u;      // piece wave + 32 exists for waves 0..3 only (zero-size descriptor)
  unsigned voffW[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) voffW[q] = (unsigned)((long)(n0 / 16 + 2 * wave + q) * kt32 * 1024 + lane * 16);
  const int ldsW = W_OFF + 2 * wave * 2048;
#define BL_RS(PTR, BYTES) __builtin_amdgcn_make_buffer_rsrc((void*)(PTR), 0, (BYTES), 0x00020000)
#define ISSUE_A(J, TILE)                                                                                  \
  do {                                                                                                    \
    const int t__ = (TILE);                                                                               \
    const __amdgpu_buffer_rsrc_t rs__ = BL_RS(p.A, t__ < nk ? ((J) < 4 ? a_bytes : a4_bytes) : 0u);       \
    BL_GLDS(rs__, smem + (t__ & 1) * STAGE + ldsA[J], voffA[J], t__ * 128);                               \
  } while (0)
#define ISSUE_W(Q, KS, TILE)                                                                              \
  do {                                                                                                    \
    const int t__ = (TILE);                                                                               \
    const __amdgpu_buffer_rsrc_t rs__ = BL_RS(p.W, t__ < nk ? w_bytes : 0u);                              \
    BL_GLDS(rs__, smem + (t__ & 1) * STAGE + ldsW + (Q) * 2048 + (KS) * 1024, voffW[Q] + (KS) * 1024, t__ * 2048); \
  } while (0)
  const int cb0 = (lg ^ (lane & 7)) << 4;
  const int offX = (wm * 144 + l15) * ROW_BYTES;
  const int offY = W_OFF + wn * 8192 + lane * 16;
#define READ_X(T3, SB)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                         \
    X[i * 2] = *(const bf16x8_t*)((SB) + offX + (3 * (T3) + i) * 2048 + cb0);                             \
    X[i * 2 + 1] = *(const bf16x8_t*)((SB) + offX + (3 * (T3) + i) * 2048 + (cb0 ^ 64));                  \
  }
#define READ_Y(DST, NH, SB)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                         \
    DST[j * 2] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048);                                 \
    DST[j * 2 + 1] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048 + 1024);                      \
  }
#define MMA(YR, T3, NH)                                                                                   \
  do {                                                                                                    \
    __builtin_amdgcn_s_setprio(1);                                                                        \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                       \
        _Pragma("unroll") for (int i = 0; i < 3; ++i)                                                     \
          acc[2 * (NH) + j][3 * (T3) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                      \
              YR[j * 2 + ks], X[i * 2 + ks], acc[2 * (NH) + j][3 * (T3) + i], 0, 0, 0);                   \
    __builtin_amdgcn_s_setprio(0);                                                                        \
  } while (0)
#define BAR()                                                                                             \
  do {                                                                                                    \
    __builtin_amdgcn_s_barrier();                                                                         \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  } while (0)
#define WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define WAIT_VM2_LGKM() asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory")
  // the six slots of K-tile T (stage SC; SN = the other stage): fragment reads first, then the LDS-DMA issue
#define SLOT1(SC, T) do { READ_X(0, SC); READ_Y(Ya, 0, SC); __builtin_amdgcn_sched_barrier(0); ISSUE_A(2, (T) + 1); WAIT_LGKM(); } while (0)
#define SLOT2(SC, T) do { READ_Y(Yb, 1, SC); __builtin_amdgcn_sched_barrier(0); ISSUE_A(3, (T) + 1); ISSUE_A(4, (T) + 1); WAIT_LGKM(); } while (0)
#define SLOT3(SC, T) do { READ_X(1, SC); __builtin_amdgcn_sched_barrier(0); ISSUE_W(0, 0, (T) + 1); WAIT_LGKM(); } while (0)
#define SLOT4(SC, T) do { ISSUE_W(0, 1, (T) + 1); ISSUE_W(1, 0, (T) + 1); } while (0)
#define SLOT5(SC, T) do { READ_X(2, SC); __builtin_amdgcn_sched_barrier(0); ISSUE_W(1, 1, (T) + 1); WAIT_LGKM(); } while (0)
#define SLOT6(SC, T) do { ISSUE_A(0, (T) + 2); ISSUE_A(1, (T) + 2); WAIT_VM2_LGKM(); } while (0)
#define BL_EPILOGUE()                                                                                     \
  do {                                                                                                    \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                      \
    const int ncol__[4] = {n0 + wn * 64 + lg * 4, n0 + wn * 64 + 16 + lg * 4, n0 + wn * 64 + 32 + lg * 4,      \
                           n0 + wn * 64 + 48 + lg * 4};                                                   \
    epilogue_tile<EPI, 4, 9>(p, m0 + wm * 144 + l15, ncol__, m0 + 288, n0 + 256, acc);                    \
  } while (0)

  f32x4_t acc[4][9];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t X[6], Ya[4], Yb[4];
  char* const S0 = smem;
  char* const S1 = smem + STAGE;

  // prologue: all nine pieces of K-tile 0, then what slot p6 of "K-tile -1" issues (the first two pieces of K-tile 1)
  ISSUE_A(0, 0); ISSUE_A(1, 0); ISSUE_A(2, 0); ISSUE_A(3, 0); ISSUE_A(4, 0);
  ISSUE_W(0, 0, 0); ISSUE_W(0, 1, 0); ISSUE_W(1, 0, 0); ISSUE_W(1, 1, 0);
  ISSUE_A(0, 1); ISSUE_A(1, 1);
  asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  BAR();
  if (wm == 0) {
    // ======================= group A: MFMA block first, then the slot of the NEXT block =======================
    SLOT1(S0, 0);
    BAR();
    for (int t = 0; t < nk; t += 2) {        // nk is even (launcher: K % 128 == 0)
      MMA(Ya, 0, 0); BAR();   SLOT2(S0, t); BAR();
      MMA(Yb, 0, 1); BAR();   SLOT3(S0, t); BAR();
      MMA(Yb, 1, 1); BAR();   SLOT4(S0, t); BAR();
      MMA(Ya, 1, 0); BAR();   SLOT5(S0, t); BAR();
      MMA(Ya, 2, 0); BAR();   SLOT6(S0, t); BAR();
      MMA(Yb, 2, 1); BAR();   SLOT1(S1, t + 1); BAR();
      MMA(Ya, 0, 0); BAR();   SLOT2(S1, t + 1); BAR();
      MMA(Yb, 0, 1); BAR();   SLOT3(S1, t + 1); BAR();
      MMA(Yb, 1, 1); BAR();   SLOT4(S1, t + 1); BAR();
      MMA(Ya, 1, 0); BAR();   SLOT5(S1, t + 1); BAR();
      MMA(Ya, 2, 0); BAR();   SLOT6(S1, t + 1); BAR();
      MMA(Yb, 2, 1); BAR();   SLOT1(S0, t + 2); BAR();
    }
    BL_EPILOGUE();
    return;
  }
  // ========================= group B: the slot of THIS block first, then its MFMA block =========================
  BAR();
  for (int t = 0; t < nk; t += 2) {
    SLOT1(S0, t); BAR();       MMA(Ya, 0, 0); BAR();
    SLOT2(S0, t); BAR();       MMA(Yb, 0, 1); BAR();
    SLOT3(S0, t); BAR();       MMA(Yb, 1, 1); BAR();
    SLOT4(S0, t); BAR();       MMA(Ya, 1, 0); BAR();
    SLOT5(S0, t); BAR();       MMA(Ya, 2, 0); BAR();
    SLOT6(S0, t); BAR();       MMA(Yb, 2, 1); BAR();
    SLOT1(S1, t + 1); BAR();   MMA(Ya, 0, 0); BAR();
    SLOT2(S1, t + 1); BAR();   MMA(Yb, 0, 1); BAR();
    SLOT3(S1, t + 1); BAR();   MMA(Yb, 1, 1); BAR();
    SLOT4(S1, t + 1); BAR();   MMA(Ya, 1, 0); BAR();
    SLOT5(S1, t + 1); BAR();   MMA(Ya, 2, 0); BAR();
    SLOT6(S1, t + 1); BAR();   MMA(Yb, 2, 1); BAR();
  }
  BL_EPILOGUE();
#undef BL_RS
#undef ISSUE_A
#undef ISSUE_W
#undef READ_X
#undef READ_Y
#undef MMA
#undef BAR
#undef WAIT_LGKM
#undef WAIT_VM2_LGKM
#undef SLOT1
#undef SLOT2
#undef SLOT3
#undef SLOT4
#undef SLOT5
#undef SLOT6
#undef BL_EPILOGUE
#endif
}

// ======================================================================================================================
// 256 × 256 tile, half-tile LDS-DMA ring, 4 phases per K-tile
// ======================================================================================================================
template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256;
  constexpr int STAGE = 65536, W_OFF = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages × (32 KiB activations + 32 KiB weights)

  int tm, tn;
  const int nk_all = p.K / BK;
  int kt_begin = 0, nk = nk_all;        // this workgroup's K-tile range [kt_begin, nk)
  if (p.splitk > 1) {                   // split-K tail: blockIdx = leftover tile * splitk + slice; even slice starts
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
  // zero-record descriptors: every access is out of range → LDS gets zeros, no memory traffic (K-tiles past the end)
  const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, 0, 0x00020000);

  // ---- LDS-DMA source offsets. Half-tile X<mh>: rows {wm'*128 + mh*64 + 0..63}, wm' = 0,1 → 16 pieces of 8 rows;
  //      this wave takes pieces 2*wave, 2*wave+1. Half-tile Y<nh>: for wn' = 0..3 the weight n-tiles wn'*4 + 2*nh + {0,1},
  //      both k-steps → 16 KiB; this wave takes (wn' = wave>>1, n-tile 2*nh + (wave&1)), k-steps 0 and 1. ----
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

  // ---- fragment reads ----
  const int cb0 = (lg ^ (lane & 7)) << 4;                       // swizzled chunk of k-step 0; k-step 1 = cb0 ^ 64
  const int offX = (wm * 128 + l15) * ROW_BYTES;                // + mh*8192 + i*2048
  const int offY = W_OFF + wn * 8192 + lane * 16;               // + (2*nh + j)*2048 + ks*1024
#define READ_X(DST, MH, SB)                                                                             \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                       \
    DST[i * 2] = *(const bf16x8_t*)((SB) + offX + (MH) * 8192 + i * 2048 + cb0);                        \
    DST[i * 2 + 1] = *(const bf16x8_t*)((SB) + offX + (MH) * 8192 + i * 2048 + (cb0 ^ 64));            \
  }
#define READ_Y(DST, NH, SB)                                                                             \
  _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
    DST[j * 2] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048);                               \
    DST[j * 2 + 1] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048 + 1024);                    \
  }
#define MMA(XR, YR, MH, NH)                                                                             \
  do {                                                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
          acc[2 * (NH) + j][4 * (MH) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                    \
              YR[j * 2 + ks], XR[i * 2 + ks], acc[2 * (NH) + j][4 * (MH) + i], 0, 0, 0);                \
    __builtin_amdgcn_s_setprio(0);                                                                      \
  } while (0)
  // lgkmcnt(0): this phase's prefetch reads have returned (so a later LDS-DMA into the same half-tile cannot overtake
  // them); vmcnt(4): all but the two youngest half-tiles have landed; the barrier publishes them to the other waves.
#define PHASE_END(WAITVM)                                                         \
  do {                                                                            \
    if (WAITVM) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");       \
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       \
    __builtin_amdgcn_s_barrier();                                                 \
    __builtin_amdgcn_sched_barrier(0);                                            \
  } while (0)

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t Xa[8], Xb[8], Ya[4], Yb[4];
  char* const S0 = smem;
  char* const S1 = smem + STAGE;

  // ---- prologue: X0 Y0 Y1 X1 of tile 0 and X0 of tile 1; the first three must have landed ----
  ISSUE_X(0, kt_begin); ISSUE_Y(0, kt_begin); ISSUE_Y(1, kt_begin); ISSUE_X(1, kt_begin); ISSUE_X(0, kt_begin + 1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  READ_X(Xa, 0, S0);
  READ_Y(Ya, 0, S0);

  for (int t = kt_begin; t < nk; t += 2) {
    // ---- K-tile t (stage 0; kt_begin is even) ----
    ISSUE_Y(0, t + 1); READ_Y(Yb, 1, S0);                    MMA(Xa, Ya, 0, 0); PHASE_END(1);
    ISSUE_Y(1, t + 1); READ_X(Xb, 1, S0);                    MMA(Xa, Yb, 0, 1); PHASE_END(0);
    ISSUE_X(1, t + 1); READ_Y(Ya, 0, S0);                    MMA(Xb, Yb, 1, 1); PHASE_END(1);
    ISSUE_X(0, t + 2); READ_X(Xa, 0, S1); READ_Y(Yb, 0, S1); MMA(Xb, Ya, 1, 0); PHASE_END(1);
    // ---- K-tile t+1 (stage 1); the two weight register sets have swapped roles ----
    ISSUE_Y(0, t + 2); READ_Y(Ya, 1, S1);                    MMA(Xa, Yb, 0, 0); PHASE_END(1);
    ISSUE_Y(1, t + 2); READ_X(Xb, 1, S1);                    MMA(Xa, Ya, 0, 1); PHASE_END(0);
    ISSUE_X(1, t + 2); READ_Y(Yb, 0, S1);                    MMA(Xb, Ya, 1, 1); PHASE_END(1);
    ISSUE_X(0, t + 3); READ_X(Xa, 0, S0); READ_Y(Ya, 0, S0); MMA(Xb, Yb, 1, 0); PHASE_END(1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // out-of-range tail loads must not outlive the wave's LDS
#undef ISSUE_X
#undef ISSUE_Y
#undef READ_X
#undef READ_Y
#undef MMA
#undef PHASE_END

  if (p.splitk > 1) {   // fp32 partial tile → slab[blockIdx][i*8+j][thread] (16 B per lane, fully coalesced)
    float* dst = p.slab + ((long)blockIdx.x * 32 * 512 + tid) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) *(f32x4_t*)(dst + (long)(i * 8 + j) * 512 * 4) = acc[i][j];
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      epilogue_store4<EPI>(p, m0 + wm * 128 + j * 16 + l15, n0 + wn * 64 + i * 16 + lg * 4, acc[i][j]);
#endif
}

// ======================================================================================================================
// 256 × 256 tile, STAGGERED wave groups: waves 0-3 (group A, wm = 0) and waves 4-7 (group B, wm = 1) share each SIMD
// pairwise; running them half a phase apart lets one group's 16 MFMAs cover the other group's LDS-DMA issue and
// fragment reads, so the matrix pipe no longer idles while both waves of a SIMD load in lockstep
// (square 8k: 1328 → 1430 TFLOP/s; Llama layer GEMMs 1719 → 1594 µs).
//
//   segment 2p   : A: MMA(p)                      B: ISSUE H(p+6), READ(p)
//   segment 2p+1 : A: ISSUE H(p+6), READ(p+1)     B: MMA(p)                       (one raw s_barrier after every segment)
//
// p = global phase (4 per K-tile, quadrants (0,0)(0,1)(1,1)(1,0)); H(p) = the half-tile first read for phase p:
// X0(t), Y1(t), X1(t), Y0(t+1). Half-tiles are issued 6 phases ahead in need order; H(p+1) must be complete before
// segment 2p+1, so at the end of segment 2p group A (issued up to H(p+5)) waits vmcnt(8) and group B (up to H(p+6))
// vmcnt(10). A buffer is re-issued ≥ 2 barriers after its last read. Fragments: one activation set (X), two weight sets.
// (Tried and rejected: prefetching the next phase's fragments inside the group's own MMA segment — hipcc serialises the
// reads ahead of the MFMAs, the MMA segment grows, 1430 → 1286 TFLOP/s.)
// ======================================================================================================================
template <int EPI, bool TN = false>
__global__ __launch_bounds__(512) void gemm256s_kernel(GemmArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int BM = 256, BN = 256;
  constexpr int STAGE = 65536, W_OFF = 32768;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int tm, tn;
  const int nk_all = TN ? (p.K + BK - 1) / BK : p.K / BK;
  int kt_begin = 0, nk = nk_all;
  // Persistent form (p.ptiles > 0, round 4): the launch has one workgroup per CU and workgroup b walks tiles b, b + grid,
  // b + 2·grid … < ptiles of the XCD-contiguous order. The K streams of consecutive tiles are FLATTENED: the loop below
  // issues its LDS-DMA half-tiles 1.5 K-tiles ahead, and what used to be zero-size loads past the end of K (K-tiles nk,
  // nk + 1) are now the first two K-tiles of the workgroup's next tile — exactly the seven half-tiles of the prologue, in
  // its order, into the stages the next tile expects (nk is even) — so the next tile starts with its operands in LDS and
  // its first fragments in registers, behind this tile's epilogue instead of behind a workgroup launch and a cold prologue.
  const int ptiles = p.splitk > 1 ? 0 : p.ptiles;
  int vb = blockIdx.x;                     // virtual block index of the current tile
  if (p.splitk > 1) {
    lin_to_tile(p, p.tail_base + blockIdx.x / p.splitk, tm, tn);
    const int slice = blockIdx.x % p.splitk;
    kt_begin = (int)(((long)slice * nk_all) / p.splitk) & ~1;
    nk = slice + 1 == p.splitk ? nk_all : ((int)(((long)(slice + 1) * nk_all) / p.splitk) & ~1);
  } else if (ptiles > 0) {
    tile_coords_v(p, vb, ptiles, tm, tn);
  } else {
    tile_coords(p, tm, tn);
  }
  int m0 = tm * BM, n0 = tn * BN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, lg = lane >> 4;
  const int kt32 = p.K >> 5;

  // TN (C = Aᵀ·B, the weight gradient dW = dyᵀ·x read UNtransposed): A is [K tokens, M] and W is [K tokens, N], both
  // row-major; a K-tile is 64 token rows of each. The descriptors end with the last token row's last column (the exact
  // extent of a column-slice view), so K-tiles past the end and the ragged last one read zeros.
  // The per-lane offsets below are TILE-RELATIVE; the tile's origin goes into the buffer descriptor's base (uniform), and
  // its size shrinks by the same bytes, so the bounds check (zero-fill past M / N / K) sees what it saw with absolute
  // offsets. A tile's (base, size) pair is rebuilt per tile from (m0, n0): TILE_A / TILE_W.
  const long a_total = TN ? ((long)(p.K - 1) * p.lda + p.M) * 2 : (long)p.M * p.lda * 2;
  const long w_total = TN ? ((long)(p.K - 1) * p.ldw + p.N) * 2 : (long)p.N * p.K * 2;
  const int kstepX = TN ? (int)(64 * p.lda * 2) : 128, kstepY = TN ? (int)(64 * p.ldw * 2) : 2048;
#define TILE_A_OFF(M0) (TN ? (long)(M0) * 2 : (long)(M0) * p.lda * 2)
#define TILE_W_OFF(N0) (TN ? (long)(N0) * 2 : (long)((N0) / 16) * (p.K >> 5) * 1024)
  const char* Acur = (const char*)p.A + TILE_A_OFF(m0);
  const char* Wcur = (const char*)p.W + TILE_W_OFF(n0);
  unsigned a_bytes = (unsigned)min(a_total - TILE_A_OFF(m0), 0xffffffffL);
  unsigned w_bytes = (unsigned)min(w_total - TILE_W_OFF(n0), 0xffffffffL);
  // the workgroup's NEXT tile (size 0 = none: loads past the end of K write zeros, as they always did)
  const char* Anext = Acur; const char* Wnext = Wcur;
  unsigned a_bytes_next = 0u, w_bytes_next = 0u;
  int m0_next = 0, n0_next = 0;
#define BL_NEXT_TILE()                                                                      \
  do {                                                                                      \
    a_bytes_next = 0u; w_bytes_next = 0u;                                                   \
    if (ptiles > 0 && vb + (int)gridDim.x < ptiles) {                                       \
      int tm__, tn__;                                                                       \
      tile_coords_v(p, vb + (int)gridDim.x, ptiles, tm__, tn__);                            \
      m0_next = tm__ * BM; n0_next = tn__ * BN;                                             \
      Anext = (const char*)p.A + TILE_A_OFF(m0_next);                                       \
      Wnext = (const char*)p.W + TILE_W_OFF(n0_next);                                       \
      a_bytes_next = (unsigned)min(a_total - TILE_A_OFF(m0_next), 0xffffffffL);             \
      w_bytes_next = (unsigned)min(w_total - TILE_W_OFF(n0_next), 0xffffffffL);             \
    }                                                                                       \
  } while (0)
  BL_NEXT_TILE();

  const int prow = lane >> 3, pchunk = (lane & 7) ^ prow;
  unsigned voffX[2][2], voffY[2][2];
  int ldsX[2][2], ldsY[2];
  if constexpr (!TN) {
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pi = 2 * wave + j, row0 = (pi >> 3) * 128 + mh * 64 + (pi & 7) * 8;
        voffX[mh][j] = (unsigned)(((long)(row0 + prow) * p.lda) * 2 + pchunk * 16);
        ldsX[mh][j] = row0 * ROW_BYTES;
      }
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int nt = (wave >> 1) * 4 + 2 * nh + (wave & 1);
      voffY[nh][0] = (unsigned)((long)nt * kt32 * 1024 + lane * 16);
      voffY[nh][1] = voffY[nh][0] + 1024;
      ldsY[nh] = W_OFF + nt * 2048;
    }
  } else {
    // LDS keeps the NT kernel's regions but token-major inside them: X block (wm', mh) = [64 tokens][64 columns = 128 B],
    // Y half nh = [64 tokens][128 columns = 256 B]. The fragments come from transposing reads (ds_read_b64_tr_b16:
    // a 32-lane half touches 8 token rows × 32 B), so 32-byte chunk PAIRS are swizzled against the token index:
    // X pair' = pair ^ ((t >> 1) & 3), Y pair' = pair ^ (t & 7) — the 8 rows then cover all 64 banks once.
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int pi = 2 * wave + j, t = (pi & 7) * 8 + prow, pos = lane & 7;
        const int c = ((((pos >> 1) ^ ((t >> 1) & 3)) << 1) | (pos & 1));
        voffX[mh][j] = (unsigned)((long)t * p.lda * 2 + (long)((pi >> 3) * 128 + mh * 64) * 2 + c * 16);
        ldsX[mh][j] = ((pi >> 3) * 128 + mh * 64 + (pi & 7) * 8) * ROW_BYTES;
      }
    // Y half nh = columns nh*128 .. +127 of the tile: [64 tokens][256 B], whole 128-byte lines per piece (4 token rows × 256 B;
    // a 64-byte-segment layout made every line travel twice: −30 %). Wave wn computes columns nh*128 + wn*32 + j*16 + (0..15).
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int t = (2 * wave + j) * 4 + (lane >> 4), pos = lane & 15;
        const int c = ((((pos >> 1) ^ (t & 7)) << 1) | (pos & 1));
        voffY[nh][j] = (unsigned)((long)t * p.ldw * 2 + (long)(nh * 128) * 2 + c * 16);
      }
      ldsY[nh] = W_OFF + nh * 16384 + wave * 2048;
    }
  }
// past the end of K the descriptor's size is 0 (the load then writes zeros): only that one dword of it varies
#define BL_RS(PTR, BYTES) __builtin_amdgcn_make_buffer_rsrc((void*)(PTR), 0, (BYTES), 0x00020000)
#define ISSUE_X(MH, TILE)                                                                   \
  do {                                                                                      \
    const int t__ = (TILE);                                                                 \
    const bool nx__ = t__ >= nk;                 /* K-tile t - nk of the next tile */        \
    const int tt__ = nx__ ? t__ - nk : t__;                                                 \
    const __amdgpu_buffer_rsrc_t rs__ = BL_RS(nx__ ? Anext : Acur, nx__ ? a_bytes_next : a_bytes); \
    char* b__ = smem + (t__ & 1) * STAGE;                                                   \
    BL_GLDS(rs__, b__ + ldsX[MH][0], voffX[MH][0], tt__ * kstepX);                             \
    BL_GLDS(rs__, b__ + ldsX[MH][1], voffX[MH][1], tt__ * kstepX);                             \
  } while (0)
#define ISSUE_Y(NH, TILE)                                                                   \
  do {                                                                                      \
    const int t__ = (TILE);                                                                 \
    const bool nx__ = t__ >= nk;                                                            \
    const int tt__ = nx__ ? t__ - nk : t__;                                                 \
    const __amdgpu_buffer_rsrc_t rs__ = BL_RS(nx__ ? Wnext : Wcur, nx__ ? w_bytes_next : w_bytes); \
    char* b__ = smem + (t__ & 1) * STAGE;                                                   \
    BL_GLDS(rs__, b__ + ldsY[NH], voffY[NH][0], tt__ * kstepY);                               \
    BL_GLDS(rs__, b__ + ldsY[NH] + 1024, voffY[NH][1], tt__ * kstepY);                        \
  } while (0)
  const int cb0 = (lg ^ (lane & 7)) << 4;
  const int offX = (wm * 128 + l15) * ROW_BYTES;
  const int offY = W_OFF + wn * 8192 + lane * 16;
  // TN: lane 4q+u of a 16-lane group addresses token row 4·lg + q, 8-byte unit u of the 16-column block and receives
  // column l15 of the four rows; the second read is 16 tokens further → slots 0-3 / 4-7 of the MFMA operand hold tokens
  // {4lg..4lg+3} and {16+4lg..}, the same for both operands.
  const int trX = (wm * 128) * ROW_BYTES + (4 * lg + (l15 >> 2)) * 128 + (l15 & 3) * 8, swX = ((lg & 1) << 1) | (l15 >> 3);
  const int trY = W_OFF + (4 * lg + (l15 >> 2)) * 256 + (l15 & 3) * 8, swY = ((lg & 1) << 2) | (l15 >> 2);
  // The transposing reads are inline asm: through the builtin, hipcc orders every one of them behind ALL outstanding LDS-DMA
  // (`s_waitcnt vmcnt(0)` before each read group — the intrinsic carries no memory operand to disambiguate), which
  // serialises the staging pipeline (−30 %). The asm results land asynchronously, so they are only touched after the
  // explicit lgkmcnt wait + barrier that precede every MMA, where BL_PIN marks them as produced.
  const unsigned ldsbase = (unsigned)(size_t)LDS_PTR(smem);
#define BL_TR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "n"(OFF))
#define BL_PIN(R) asm volatile("" : "+v"(R))
#define READ_X(DST, MH, SB)                                                                             \
  do {                                                                                                  \
    if constexpr (!TN) {                                                                                \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
        DST[i * 2] = *(const bf16x8_t*)((SB) + offX + (MH) * 8192 + i * 2048 + cb0);                    \
        DST[i * 2 + 1] = *(const bf16x8_t*)((SB) + offX + (MH) * 8192 + i * 2048 + (cb0 ^ 64));        \
      }                                                                                                 \
    } else {                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                   \
        const unsigned a__ = ldsbase + (unsigned)((SB) - smem) + trX + ((i ^ swX) << 5);                \
        BL_TR(DST##2[i * 4 + 0], a__, (MH) * 8192);                                                     \
        BL_TR(DST##2[i * 4 + 1], a__, (MH) * 8192 + 2048);                                              \
        BL_TR(DST##2[i * 4 + 2], a__, (MH) * 8192 + 4096);                                              \
        BL_TR(DST##2[i * 4 + 3], a__, (MH) * 8192 + 4096 + 2048);                                       \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
#define READ_Y(DST, NH, SB)                                                                             \
  do {                                                                                                  \
    if constexpr (!TN) {                                                                                \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
        DST[j * 2] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048);                           \
        DST[j * 2 + 1] = *(const bf16x8_t*)((SB) + offY + (2 * (NH) + j) * 2048 + 1024);                \
      }                                                                                                 \
    } else {                                                                                            \
      _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                   \
        const unsigned a__ = ldsbase + (unsigned)((SB) - smem) + trY + (((wn * 2 + j) ^ swY) << 5);     \
        BL_TR(DST##2[j * 4 + 0], a__, (NH) * 16384);                                                    \
        BL_TR(DST##2[j * 4 + 1], a__, (NH) * 16384 + 4096);                                             \
        BL_TR(DST##2[j * 4 + 2], a__, (NH) * 16384 + 8192);                                             \
        BL_TR(DST##2[j * 4 + 3], a__, (NH) * 16384 + 8192 + 4096);                                      \
      }                                                                                                 \
    }                                                                                                   \
  } while (0)
// TN fragments: DST2[f * 4 + ks * 2 + half] (8 bytes each); operand ks of fragment f = {[f*4 + 2ks], [f*4 + 2ks + 1]}
#define BL_FRAG(R2, F, KS)                                                                              \
  __builtin_bit_cast(bf16x8_t, ((u32x4_t){R2[(F) * 4 + 2 * (KS)][0], R2[(F) * 4 + 2 * (KS)][1],         \
                                          R2[(F) * 4 + 2 * (KS) + 1][0], R2[(F) * 4 + 2 * (KS) + 1][1]}))
#define MMA(XR, YR, MH, NH)                                                                             \
  do {                                                                                                  \
    if constexpr (TN) {                                                                                 \
      _Pragma("unroll") for (int q = 0; q < 16; ++q) BL_PIN(XR##2[q]);                                  \
      _Pragma("unroll") for (int q = 0; q < 8; ++q) BL_PIN(YR##2[q]);                                   \
    }                                                                                                   \
    __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                    \
      _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                 \
          if constexpr (TN)                                                                             \
            acc[2 * (NH) + j][4 * (MH) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
                BL_FRAG(YR##2, j, ks), BL_FRAG(XR##2, i, ks), acc[2 * (NH) + j][4 * (MH) + i], 0, 0, 0); \
          else                                                                                          \
            acc[2 * (NH) + j][4 * (MH) + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                  \
                YR[j * 2 + ks], XR[i * 2 + ks], acc[2 * (NH) + j][4 * (MH) + i], 0, 0, 0);              \
        }                                                                                               \
    __builtin_amdgcn_s_setprio(0);                                                                      \
  } while (0)
// (the timing-only variants of these four macros — no barrier / no wait / no issue / no reads — are generated into a
//  patched copy of this file by tools/micro/gemm_loop_experiments.py; nothing here is conditional)
#define BAR()                                   \
  do {                                          \
    __builtin_amdgcn_s_barrier();               \
    __builtin_amdgcn_sched_barrier(0);          \
  } while (0)
#define WAIT_VM8() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
#define WAIT_VM10_LGKM() asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory")
#define WAIT_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
  // H(p+6) for phase q of K-tile T:  q=0 → X1(T+1)   q=1 → Y0(T+2)   q=2 → X0(T+2)   q=3 → Y1(T+2)
// the fragment reads go first, the LDS-DMA issue second: the ≈ 80 cycles each piece takes to issue cover the reads' latency
// instead of preceding it (+1.5–2.3 % on every shape, same box A/B). Tried and rejected: moving one of the two pieces between
// the two k-steps of the wave's next MFMA segment (to thin out the burst of 8 pieces per segment): −3 %, the piece's issue
// time then stalls the MFMA stream itself.
#define ISSUE_READ(I, R) do { R; __builtin_amdgcn_sched_barrier(0); I; } while (0)
#define ISSUE_Q0(T) ISSUE_X(1, (T) + 1)
#define ISSUE_Q1(T) ISSUE_Y(0, (T) + 2)
#define ISSUE_Q2(T) ISSUE_X(0, (T) + 2)
#define ISSUE_Q3(T) ISSUE_Y(1, (T) + 2)
  // each group runs its own loop AND its own copy of the epilogue (a common epilogue after an if/else would force the
  // 128 accumulator registers of both loops into one assignment)
#define BL_EPILOGUE()                                                                                     \
  do {                                                                                                    \
    if (a_bytes_next == 0u && w_bytes_next == 0u)   /* last tile: zero-fill loads must not outlive the LDS */ \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                    \
    if (p.splitk > 1) {                                                                                   \
      long off__ = ((long)blockIdx.x * 32 * 512 + tid) * 4;                                               \
      asm volatile("" : "+v"(off__));   /* keeps the 32 slab addresses out of the tile loop's preheader (spills) */ \
      float* dst = p.slab + off__;                                                                        \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) *(f32x4_t*)(dst + (long)(i * 8 + j) * 512 * 4) = acc[i][j]; \
    } else {                                                                                              \
      int ncol__[4];                                                                                      \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                       \
        ncol__[i] = TN ? n0 + (i >> 1) * 128 + wn * 32 + (i & 1) * 16 + lg * 4 : n0 + wn * 64 + i * 16 + lg * 4; \
      epilogue_tile<EPI, 4, 8>(p, m0 + wm * 128 + l15, ncol__, m0 + 256, n0 + 256, acc);                  \
    }                                                                                                     \
  } while (0)

  // after a tile's epilogue: the next tile becomes the current one (its first two K-tiles are in LDS or in flight, its
  // first fragments in registers), the accumulators restart from zero
#define BL_ADVANCE()                                                                                      \
  do {                                                                                                    \
    vb += (int)gridDim.x;                                                                                 \
    m0 = m0_next; n0 = n0_next; Acur = Anext; Wcur = Wnext; a_bytes = a_bytes_next; w_bytes = w_bytes_next; \
    BL_NEXT_TILE();                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};            \
  } while (0)

  f32x4_t acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  bf16x8_t X[8], Ya[4], Yb[4];            // NT fragments (unused and eliminated in the TN form)
  u32x2_t X2[16], Ya2[8], Yb2[8];         // TN fragments, one 8-byte transposing read each
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
    for (;;) {
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
    const bool more = a_bytes_next != 0u;
    BL_EPILOGUE();
    if (!more) break;
    BL_ADVANCE();
    }
    return;
  }
  // ============================== group B: issue + read THIS phase, then MMA ==============================
  WAIT_LGKM();
  BAR();
  for (;;) {
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
  const bool more = a_bytes_next != 0u;
  BL_EPILOGUE();
  if (!more) break;
  BL_ADVANCE();
  }
#undef BL_EPILOGUE
#undef BL_ADVANCE
#undef BL_NEXT_TILE
#undef TILE_A_OFF
#undef TILE_W_OFF
#undef ISSUE_X
#undef ISSUE_Y
#undef READ_X
#undef READ_Y
#undef BL_TR
#undef BL_PIN
#undef BL_FRAG
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

// Sum the split-K slabs of the leftover tiles and apply the fused epilogue. Same thread → (m, n) map as the 256x256
// kernels; grid = leftover tiles × 32: one block per accumulator vector (i, j) of a tile, so the slab reads are spread
// over ≥ 1024 workgroups instead of 128.
template <int EPI, bool TN = false>
__global__ __launch_bounds__(512) void gemm_splitk_reduce_kernel(GemmArgs p) {
  int tm, tn;
  const int tile_local = blockIdx.x >> 5, idx = blockIdx.x & 31, i = idx >> 3, j = idx & 7;
  lin_to_tile(p, p.tail_base + tile_local, tm, tn);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l15 = lane & 15, lg = lane >> 4;
  f32x4_t sum = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < p.splitk; ++s)
    sum += *(const f32x4_t*)(p.slab + (((long)(tile_local * p.splitk + s) * 32 + idx) * 512 + tid) * 4);
  epilogue_store4<EPI>(p, tm * 256 + wm * 128 + j * 16 + l15,
                       TN ? tn * 256 + (i >> 1) * 128 + wn * 32 + (i & 1) * 16 + lg * 4 : tn * 256 + wn * 64 + i * 16 + lg * 4, sum);
}

template <int MB, int NB>
constexpr int mid_lds_bytes() { return 3 * (64 * MB * ROW_BYTES + NB * 2048); }
template <int EPI, int MB, int NB, int SK = 0>
bool mid_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mid_kernel<EPI, MB, NB, SK>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, mid_lds_bytes<MB, NB>()) == hipSuccess;
}

constexpr int RS_LDS_BYTES = 4 * 2 * 6 * 16 * ROW_BYTES;   // gemm_rows_stream_kernel: NWB chunks of 96 rows × 2 K-tiles
template <int EPI, int NWV, int SK>
int launch_rs(const GemmArgs& p, hipStream_t s, int n_tiles) {
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_rows_stream_kernel<EPI, NWV, SK>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES) != hipSuccess) return BL_E_LAUNCH;
    done = true;
  }
  hipLaunchKernelGGL((gemm_rows_stream_kernel<EPI, NWV, SK>), dim3((n_tiles + NWV - 1) / NWV, 8 / SK), dim3(NWV * 64),
                     RS_LDS_BYTES, s, p, n_tiles);
  return BL_OK;
}

// bl_gemm_skinny_rows_bf16: M <= 128 rows in the skinny kernel's summation order — gemm_rows_stream_kernel for the wide
// layers at M <= 96 (qkv, gate/up, lm_head), gemm_mid_kernel<SK> otherwise
template <int EPI>
int launch_rows_sk(const GemmArgs& a, hipStream_t s) {
  static bool done = false;
  if (!done) {
    if (!mid_attr<EPI, 2, 4, 8>() || !mid_attr<EPI, 2, 4, 2>()) return BL_E_LAUNCH;
    done = true;
  }
  GemmArgs p = a;
  p.fold_ks = p.K / 256;                       // 8 slices of K/8 columns = K/256 MFMA k-steps each
  // Rows-stream form: slices that end on its 4-k-step chunks (K a multiple of 1024), more than 256 weight tiles (the narrow
  // layers' two launches — K split + reduce — are launch-bound and the mid kernel's are shorter: o 16.7 vs 19.5 µs, down
  // 28.3 vs 33.5). 7B at 96 rows, same box: gate/up 53.5 → 40.2 µs (6 waves × 230 workgroups), qkv 38.1 → 32.7 (8 waves ×
  // 96 column groups × 2 K-halves + the tree's last level in the reduce kernel). BL_ROWS_STREAM=0 switches it off (A/B).
  const int n_tiles = p.N / 16;
  const char* e_on = getenv("BL_ROWS_STREAM");
  if (p.M <= 96 && p.fold_ks % 4 == 0 && n_tiles > 256 && (e_on ? atoi(e_on) : 1)) {
    const bool can_split = p.slab && p.slab_bytes >= 2L * p.M * p.N * 4;
    int rc;
    if ((n_tiles + 7) / 8 >= 200) {                       // one round of 8-wave workgroups fills the chip (lm_head)
      p.splitk = 1;
      rc = launch_rs<EPI, 8, 8>(p, s, n_tiles);
    } else if ((n_tiles + 5) / 6 >= 200 || !can_split) {  // 6-wave workgroups do (gate/up: 230)
      p.splitk = 1;
      rc = launch_rs<EPI, 6, 8>(p, s, n_tiles);
    } else {                                              // two K-halves of 4 slices each + the last tree level
      p.splitk = 2;
      rc = launch_rs<BL_EPI_NONE, 8, 4>(p, s, n_tiles);
      const long work = (long)p.M * (p.N / 4);
      if (rc == BL_OK)
        hipLaunchKernelGGL((gemm_rows_tree_reduce_kernel<EPI>), dim3((int)min((work + 255) / 256, 2048L)), dim3(256), 0, s, p);
    }
    if (rc != BL_OK) return rc;
    BL_CHECK_LAUNCH();
    return BL_OK;
  }
  const int slabs64 = (p.N + 63) / 64;
  // 64-column slabs, every weight byte once, all rows of A staged once per workgroup. Where that leaves most CUs without
  // a workgroup (N = 4096: 64 slabs) and the caller gave a workspace, grid.y = 4 workgroups take two K-slices each and
  // the tree is finished by the reduce kernel — the split is exact (see gemm_mid_kernel) and the same for every row.
  const bool split = slabs64 * 2 <= 256 && p.slab && p.slab_bytes >= 4L * p.M * p.N * 4;
  p.splitk = split ? 4 : 1;
  if (split) {
    hipLaunchKernelGGL((gemm_mid_kernel<EPI, 2, 4, 2>), dim3(slabs64, 4), dim3(256), (mid_lds_bytes<2, 4>()), s, p);
    const long work = (long)p.M * (p.N / 4);
    hipLaunchKernelGGL((gemm_rows_tree_reduce_kernel<EPI>), dim3((int)min((work + 255) / 256, 2048L)), dim3(256), 0, s, p);
  } else {
    hipLaunchKernelGGL((gemm_mid_kernel<EPI, 2, 4, 8>), dim3(slabs64), dim3(256), (mid_lds_bytes<2, 4>()), s, p);
  }
  BL_CHECK_LAUNCH();
  return BL_OK;
}

template <int EPI>
int set_lds_attr() {
  static bool done = false;   // idempotent; a benign race only repeats the same calls
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 65536) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm288_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 73728) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm288s_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 73728) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128_kernel<EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * ROW_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tail_kernel<EPI, 128, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * ROW_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tail_kernel<EPI, 128, 64, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 192 * ROW_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tail_kernel<EPI, 64, 64, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * ROW_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tail_kernel<EPI, 160, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 288 * ROW_BYTES) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring8_kernel<EPI, 160, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 288 * ROW_BYTES + 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring8_kernel<EPI, 128, 128, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 256 * ROW_BYTES + 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring8_kernel<EPI, 128, 64, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 192 * ROW_BYTES + 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_ring8_kernel<EPI, 64, 64, 4>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 128 * ROW_BYTES + 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mid2_kernel<EPI, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * (160 * ROW_BYTES + 4096)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mid2_kernel<EPI, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            3 * (160 * ROW_BYTES + 4 * 4096)) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_mid2_kernel<EPI, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            2 * (160 * ROW_BYTES + 2 * 4096)) != hipSuccess ||
        !mid_attr<EPI, 2, 4>() || !mid_attr<EPI, 4, 4>() || !mid_attr<EPI, 5, 4>() || !mid_attr<EPI, 2, 1>() ||
        !mid_attr<EPI, 4, 1>() || !mid_attr<EPI, 5, 1>())
      return BL_E_LAUNCH;
    done = true;
  }
  return BL_OK;
}

template <int EPI>
int launch_gemm(const GemmArgs& a, hipStream_t s) {
  constexpr int CUS = 256;              // MI355X: the 256x256 kernel runs one workgroup per CU
  constexpr int LDS128 = 2 * 256 * ROW_BYTES, LDS256 = 2 * 65536;
  if (set_lds_attr<EPI>() != BL_OK) return BL_E_LAUNCH;
  GemmArgs p = a;
  static const char* force = getenv("BL_GEMM_TILE");   // "128" / "256": benchmarking aid
  const int bm = (p.M + 255) / 256, bn = (p.N + 255) / 256, big_tiles = bm * bn;
  // one (partial) round of big tiles beats 1.5+ rounds of the 128 kernel once about half of the CUs get a tile (ViT qkv at
  // B = 16: 204 / 224 tiles, 45 → 39 µs; round 3, the narrow ViT layers at the training batch of 32 images — 132 / 160 tiles
  // for M = 8352 / 8192, N = 1024 / 1152 — where the 128 kernel needs 528 / 576 > 512 workgroup slots: 104 → 82 µs at
  // K = 4096, 35 → 30 µs at K = 1024)
  static const int big_min = getenv("BL_GEMM_BIG_MIN") ? atoi(getenv("BL_GEMM_BIG_MIN")) : 128;   // A/B aid
  bool big = big_tiles >= big_min && p.K >= 512;
  if (force) big = force[0] == '2';
  static const bool no_mid = getenv("BL_GEMM_NO_MID") != nullptr;      // A/B aid
  // M <= 320: the weight-streaming mid kernels; up to 640 rows (B = 2 prefill) the 160-row mid2 kernel still beats the
  // tile kernels (38.2 -> 36.4 ms per batch), beyond that it loses (B = 4: 42.7 vs 48.2 ms)
  static const bool no_mid2 = getenv("BL_GEMM_NO_MID2") != nullptr;      // A/B aid
  const bool mid2_only = p.M > 320;
  if (p.M <= 640 && p.M > 32 && p.K >= 512 && !force && !no_mid && !(mid2_only && (no_mid2 || p.slab || (p.N % 32)))) {
    // every weight byte once: one workgroup per column slab, all rows; 64-column slabs when that already gives ≥ 160
    // workgroups, else 16-column slabs. grid.y slices K only with a workspace (opt-in).
    const int slabs64 = (p.N + 63) / 64, nkm = p.K / BK;
    bool wide = slabs64 >= 160;   // measured: 64-column slabs for N = 4096 without K slicing (64 workgroups) cost +2.2 ms at B = 1
    int S = 1;
    static const bool no_split_mid = getenv("BL_GEMM_NO_SPLITK") != nullptr;
    if (!wide && p.slab && !no_split_mid && slabs64 * 2 <= CUS) {
      // Every workgroup re-reads ALL M rows of A from L2, so the L2 traffic is (N / slab width) · M · K · 2 B: with a
      // workspace, narrow layers (N = 4096: 64 slabs of 64 columns) keep the 64-column slabs — a quarter of the activation
      // traffic of 16-column slabs — and fill the chip by slicing K instead (same slicing for every row: slot-invariant)
      int S2 = min(8, CUS / slabs64);
      while (S2 > 1 && (nkm / S2 < 8 || p.slab_bytes < (long)S2 * p.M * p.N * 4)) --S2;
      if (S2 > 1) { wide = true; S = S2; }
    }
    if (S == 1 && p.M > 128 && (p.N % 32) == 0 && !no_mid2) {
      // 160-row workgroups (gemm_mid2_kernel): narrow layers 32 columns (3-stage ring, 2 workgroups per CU); wide layers
      // 64 columns on a 2-stage ring so that two workgroups share a CU and one's LDS-DMA issue runs under the other's
      // MFMAs (qkv 59 -> 44 us, gate/up 112 -> 83 us at M = 288); the widest (lm_head) 128 columns.
      const int mb = (p.M + 159) / 160;
      if (!wide)
        hipLaunchKernelGGL((gemm_mid2_kernel<EPI, 1>), dim3(p.N / 32, mb), dim3(256), 3 * (160 * ROW_BYTES + 4096), s, p);
      else if (slabs64 >= 400)
        hipLaunchKernelGGL((gemm_mid2_kernel<EPI, 4>), dim3((p.N + 127) / 128, mb), dim3(256), 3 * (160 * ROW_BYTES + 4 * 4096), s, p);
      else
        hipLaunchKernelGGL((gemm_mid2_kernel<EPI, 2, 2>), dim3((p.N + 63) / 64, mb), dim3(256), 2 * (160 * ROW_BYTES + 2 * 4096), s, p);
      BL_CHECK_LAUNCH();
      return BL_OK;
    }
    const int slabs = wide ? slabs64 : (p.N + 15) / 16;
    if (S == 1 && !wide && p.slab && !no_split_mid && slabs < CUS) {
      S = min(8, (CUS + CUS / 2 + slabs - 1) / slabs);
      while (S > 1 && (nkm / S < 8 || p.slab_bytes < (long)S * p.M * p.N * 4)) --S;
    }
    p.splitk = S;
    const dim3 grid(slabs, S), block(256);
#define BL_MID(MBV)                                                                                            \
  do {                                                                                                         \
    if (wide) hipLaunchKernelGGL((gemm_mid_kernel<EPI, MBV, 4>), grid, block, (mid_lds_bytes<MBV, 4>()), s, p);  \
    else hipLaunchKernelGGL((gemm_mid_kernel<EPI, MBV, 1>), grid, block, (mid_lds_bytes<MBV, 1>()), s, p);       \
  } while (0)
    if (p.M <= 128) BL_MID(2);
    else if (p.M <= 256) BL_MID(4);
    else BL_MID(5);
#undef BL_MID
    if (S > 1) {
      const long work = (long)p.M * (p.N / 4);
      hipLaunchKernelGGL((gemm128_splitk_reduce_kernel<EPI>), dim3((int)min((work + 255) / 256, 2048L)), dim3(256), 0, s, p);
    }
    BL_CHECK_LAUNCH();
    return BL_OK;
  }
  static const bool no_ring160 = getenv("BL_GEMM_NO_RING160") != nullptr;      // A/B aid
  if (!big && !force && !no_ring160 && p.K >= 512) {
    // one round of 160 × 128 tiles on the ring-buffered kernel when that covers the problem with ≥ 3/4 of the CUs busy
    const int t160 = ((p.M + 159) / 160) * ((p.N + 127) / 128);
    if (t160 <= CUS && t160 >= (3 * CUS) / 4) {
      p.tiles_m = (p.M + 159) / 160;
      p.tiles_n = (p.N + 127) / 128;
      p.tail_base = -1;
      static const bool ring_w4 = getenv("BL_GEMM_RING160_W4") != nullptr;      // A/B aid: the one-wave-per-SIMD form
      if (ring_w4) hipLaunchKernelGGL((gemm_tail_kernel<EPI, 160, 128, 4>), dim3(t160), dim3(256), 4 * 288 * ROW_BYTES, s, p);
      else hipLaunchKernelGGL((gemm_ring8_kernel<EPI, 160, 128, 4>), dim3(t160), dim3(512), 4 * 288 * ROW_BYTES + 1024, s, p);
      BL_CHECK_LAUNCH();
      return BL_OK;
    }
  }
  if (!big) {
    p.tiles_m = (p.M + 127) / 128;
    p.tiles_n = (p.N + 127) / 128;
    const int tiles = p.tiles_m * p.tiles_n, nk128 = p.K / BK;
    // few tiles, long K (tall-skinny): slice K over grid.y when the caller gave a workspace (opt-in, as for the 256
    // kernel: sliced sums are not batch-slot invariant)
    int S128 = 1;
    static const bool ring_split = getenv("BL_GEMM_SPLITK_128") == nullptr;   // A/B aid: set → gemm128's split-K form
    if (p.slab && tiles <= CUS / 2 && !force && !getenv("BL_GEMM_NO_SPLITK")) {
      // the ring kernel holds one workgroup per CU (129 KiB of LDS), gemm128 two
      S128 = min(8, ((ring_split ? 1 : 2) * CUS) / tiles);
      while (S128 > 1 && (nk128 / S128 < 8 || p.slab_bytes < (long)S128 * p.M * p.N * 4)) --S128;
    }
    if (S128 > 1 && ring_split) {
      p.splitk = S128;
      hipLaunchKernelGGL((gemm_ring8_kernel<EPI, 128, 128, 4>), dim3(tiles, S128), dim3(512), 4 * 256 * ROW_BYTES + 1024, s, p);
      const long work = (long)p.M * (p.N / 4);
      hipLaunchKernelGGL((gemm128_splitk_reduce_kernel<EPI>), dim3((int)min((work + 255) / 256, 2048L)), dim3(256), 0, s, p);
    } else if (S128 > 1) {
      p.splitk = S128;
      hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(tiles, S128), dim3(256), LDS128, s, p);
      const long work = (long)p.M * (p.N / 4);
      hipLaunchKernelGGL((gemm128_splitk_reduce_kernel<EPI>), dim3((int)min((work + 255) / 256, 2048L)), dim3(256), 0, s, p);
    } else {
      hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(tiles), dim3(256), LDS128, s, p);
    }
    BL_CHECK_LAUNCH();
    return BL_OK;
  }
  // 288-row tiles (one 288-token sequence per row tile) where they remove the leftover round: estimated cost in units of
  // one round of 256 × 256 tiles — 256-row tiling: full rounds + 0.45 for a ≤ 64-tile tail on sub-tiles, 1 for a larger
  // partial round; 288-row tiling: rounds × 1.12 (12.5 % more MFMA work and 6 % more staging per tile).
  static const bool no_288 = getenv("BL_GEMM_NO_288") != nullptr;      // A/B aid
  if (!force && !no_288 && (p.K % 128) == 0) {
    const int t256 = big_tiles, r256 = t256 % CUS;
    const float cost256 = (float)(t256 / CUS) + (r256 == 0 ? 0.f : (r256 <= 64 && t256 > CUS) ? 0.45f : 1.0f);
    const int bm288 = (p.M + 287) / 288, t288 = bm288 * bn;
    const float cost288 = 1.12f * (float)((t288 + CUS - 1) / CUS);
    if (cost288 < 0.97f * cost256) {
      p.tiles_m = bm288;
      p.tiles_n = bn;
      static const bool lockstep288 = getenv("BL_GEMM_288_LOCKSTEP") != nullptr;      // A/B aid
      if (lockstep288) hipLaunchKernelGGL((gemm288_kernel<EPI>), dim3(t288), dim3(512), 2 * 73728, s, p);
      else hipLaunchKernelGGL((gemm288s_kernel<EPI>), dim3(t288), dim3(512), 2 * 73728, s, p);
      BL_CHECK_LAUNCH();
      return BL_OK;
    }
  }
  // Whole rounds of 256 tiles on the pipelined kernel; a partial last round would leave most CUs idle for a full tile
  // time, so its tiles are cut into 128x128 quarters and run by the small kernel (2 workgroups per CU) instead.
  p.tiles_m = bm;
  p.tiles_n = bn;
  // The last round of 256-tile launches is usually partial (e.g. 288 tiles = 1.125 rounds). Measured cost of the
  // leftover `tail` tiles in units of one full round T(K) (tools/bench_gemm.py, profiles/): plain partial round 1.0;
  // 128x128 quarters on the small kernel ≈ 0.65 when they fit one small round (tail ≤ 64 … 128), > 1 beyond; split-K
  // over S = 256/tail slices ≈ 1/S + 45 µs of fp32 slab traffic, i.e. ≈ 0.28 at K = 11008 but ≈ 0.7 at K = 4096.
  int main_tiles = big_tiles, tail = big_tiles % CUS;
  const int nk = p.K / BK;
  int S = tail ? CUS / tail : 1;
  if (S > 16) S = 16;
  while (S > 1 && nk / S < 4) --S;
  static const bool no_split = getenv("BL_GEMM_NO_SPLITK") != nullptr;
  static const bool lockstep = getenv("BL_GEMM_LOCKSTEP") != nullptr;   // A/B aid: the non-staggered 256 kernel
  // more than one round of tiles: the persistent form (one workgroup per CU walks its tiles, the next tile's first K-tiles
  // land behind this tile's epilogue); needs an even number of K-tiles (stage parity carries over) and 32-bit extents
  static const bool no_persist = getenv("BL_GEMM_NO_PERSIST") != nullptr;      // A/B aid
  const bool persist_ok = !no_persist && !lockstep && (nk % 2) == 0 && (long)p.M * p.lda * 2 < (1L << 32) &&
                          (long)p.N * p.K * 2 < (1L << 32);
#define BL_LAUNCH256(GRID)                                                                              \
  do {                                                                                                  \
    if (lockstep) hipLaunchKernelGGL((gemm256_kernel<EPI>), dim3(GRID), dim3(512), LDS256, s, p);       \
    else if (persist_ok && p.splitk <= 1 && (GRID) > CUS) {                                             \
      GemmArgs pp = p;                                                                                  \
      pp.ptiles = (GRID);                                                                               \
      hipLaunchKernelGGL((gemm256s_kernel<EPI>), dim3(CUS), dim3(512), LDS256, s, pp);                  \
    } else hipLaunchKernelGGL((gemm256s_kernel<EPI>), dim3(GRID), dim3(512), LDS256, s, p);             \
  } while (0)
  const bool can_split = tail && S >= 2 && p.K >= 8192 && p.slab &&
                         p.slab_bytes >= (long)tail * S * 256 * 256 * 4 && !no_split && !force;
  if (can_split) {
    main_tiles = big_tiles - tail;
    if (main_tiles) BL_LAUNCH256(main_tiles);
    p.tail_base = main_tiles;
    p.splitk = S;
    BL_LAUNCH256(tail * S);
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<EPI>), dim3(tail * 32), dim3(512), 0, s, p);
  } else {
    // (65 … 128 leftover tiles — Llama qkv at B = 16: 96 — as 256 × 128 half tiles instead of a quarter-filled fourth round
    // was measured at −0.6 % end to end: the half tiles stage 3/4 of a full tile's bytes for half its FLOPs)
    if (tail != 0 && tail <= 64 && main_tiles > tail && !force) main_tiles = big_tiles - tail; else tail = 0;
    BL_LAUNCH256(main_tiles);
    if (tail) {
      // leftover 256x256 tiles, cut so that the sub-tiles cover (up to) every CU once
      p.tail_base = main_tiles;
      static const bool old_tail = getenv("BL_GEMM_OLD_TAIL") != nullptr;   // A/B aid
      static const bool tail_w4 = getenv("BL_GEMM_TAIL_W4") != nullptr;     // A/B aid: the one-wave-per-SIMD tail kernels
      if (old_tail) hipLaunchKernelGGL((gemm128_kernel<EPI>), dim3(tail * 4), dim3(256), LDS128, s, p);
      else if (tail_w4 && tail <= 16)
        hipLaunchKernelGGL((gemm_tail_kernel<EPI, 64, 64, 4>), dim3(tail * 16), dim3(256), 4 * 128 * ROW_BYTES, s, p);
      else if (tail_w4 && tail <= 32)
        hipLaunchKernelGGL((gemm_tail_kernel<EPI, 128, 64, 4>), dim3(tail * 8), dim3(256), 4 * 192 * ROW_BYTES, s, p);
      else if (tail_w4)
        hipLaunchKernelGGL((gemm_tail_kernel<EPI, 128, 128, 4>), dim3(tail * 4), dim3(256), 4 * 256 * ROW_BYTES, s, p);
      // two waves per SIMD (gemm_ring8_kernel in tail mode): the sub-tiles' K-steps are LDS-DMA issue + fragment reads + MFMAs
      // of ONE wave per SIMD back to back otherwise
      else if (tail <= 16)
        hipLaunchKernelGGL((gemm_ring8_kernel<EPI, 64, 64, 4>), dim3(tail * 16), dim3(512), 4 * 128 * ROW_BYTES + 1024, s, p);
      else if (tail <= 32)
        hipLaunchKernelGGL((gemm_ring8_kernel<EPI, 128, 64, 4>), dim3(tail * 8), dim3(512), 4 * 192 * ROW_BYTES + 1024, s, p);
      else
        hipLaunchKernelGGL((gemm_ring8_kernel<EPI, 128, 128, 4>), dim3(tail * 4), dim3(512), 4 * 256 * ROW_BYTES + 1024, s, p);
    }
  }
#undef BL_LAUNCH256
  BL_CHECK_LAUNCH();
  return BL_OK;
}

// bl_gemm_tn_bf16: C[M, N] (fp32) = Aᵀ·B over K token rows, A = [K, M] and B = [K, N] row-major — the weight gradient
// dW = dyᵀ·x straight from the row-major gradient and activation buffers (no transposed copies). Whole rounds of 256 × 256
// tiles on the staggered kernel's TN form; a partial last round is split along K when the caller gave a workspace.
int launch_gemm_tn(const GemmArgs& a, hipStream_t s) {
  constexpr int CUS = 256, LDS256 = 2 * 65536;
  static bool done = false;
  if (!done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256s_kernel<BL_EPI_F32, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, LDS256) != hipSuccess)
      return BL_E_LAUNCH;
    done = true;
  }
  GemmArgs p = a;
  p.tiles_m = (p.M + 255) / 256;
  p.tiles_n = (p.N + 255) / 256;
  const int tiles = p.tiles_m * p.tiles_n, nk = (p.K + BK - 1) / BK;
  // fewer tiles than half the CUs (ViT blocks: 16 … 85 tiles): every tile is K-split so the launch fills the chip
  int tail = tiles > CUS ? tiles % CUS : (2 * tiles <= CUS ? tiles : 0);
  int S = tail ? CUS / tail : 1;
  if (tiles <= CUS && S > 8) S = 8;
  if (S > 16) S = 16;
  while (S > 1 && nk / S < 4) --S;
  static const bool no_split = getenv("BL_GEMM_NO_SPLITK") != nullptr;
  const bool can_split = tail && S >= 2 && (nk >= 128 || tiles <= CUS) && p.slab &&
                         p.slab_bytes >= (long)tail * S * 256 * 256 * 4 && !no_split;
  static const bool no_persist = getenv("BL_GEMM_NO_PERSIST") != nullptr;      // A/B aid
  const bool persist_ok = !no_persist && (nk % 2) == 0 && ((long)(p.K - 1) * p.lda + p.M) * 2 < (1L << 32) &&
                          ((long)(p.K - 1) * p.ldw + p.N) * 2 < (1L << 32);
  auto launch_main = [&](int grid) {
    if (persist_ok && grid > CUS) {
      GemmArgs pp = p;
      pp.ptiles = grid;
      hipLaunchKernelGGL((gemm256s_kernel<BL_EPI_F32, true>), dim3(CUS), dim3(512), LDS256, s, pp);
    } else {
      hipLaunchKernelGGL((gemm256s_kernel<BL_EPI_F32, true>), dim3(grid), dim3(512), LDS256, s, p);
    }
  };
  if (can_split) {
    if (tiles > tail) launch_main(tiles - tail);
    p.tail_base = tiles - tail;
    p.splitk = S;
    hipLaunchKernelGGL((gemm256s_kernel<BL_EPI_F32, true>), dim3(tail * S), dim3(512), LDS256, s, p);
    hipLaunchKernelGGL((gemm_splitk_reduce_kernel<BL_EPI_F32, true>), dim3(tail * 32), dim3(512), 0, s, p);
  } else {
    launch_main(tiles);
  }
  BL_CHECK_LAUNCH();
  return BL_OK;
}

}  // namespace bl_gemm_bf16_impl
using namespace bl_gemm_bf16_impl;

extern "C" int bl_gemm_bf16(const bl_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = fill_gemm_args(d, a);
  if (rc != BL_OK) return rc;
  if (d->a_norm_weight) return BL_E_ARG;   // the fused A-operand RMSNorm exists only in the skinny kernel
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case BL_EPI_NONE: return launch_gemm<BL_EPI_NONE>(a, s);
    case BL_EPI_BIAS: return launch_gemm<BL_EPI_BIAS>(a, s);
    case BL_EPI_BIAS_GELU: return launch_gemm<BL_EPI_BIAS_GELU>(a, s);
    case BL_EPI_BIAS_RES: return launch_gemm<BL_EPI_BIAS_RES>(a, s);
    case BL_EPI_RES: return launch_gemm<BL_EPI_RES>(a, s);
    case BL_EPI_SWIGLU: return launch_gemm<BL_EPI_SWIGLU>(a, s);
    case BL_EPI_F32: return launch_gemm<BL_EPI_F32>(a, s);
    case BL_EPI_F32_BF16R: return launch_gemm<BL_EPI_F32_BF16R>(a, s);
    case BL_EPI_SWIGLU_KEEP: return launch_gemm<BL_EPI_SWIGLU_KEEP>(a, s);
    case BL_EPI_BIAS_GELU_KEEP: return launch_gemm<BL_EPI_BIAS_GELU_KEEP>(a, s);
    case BL_EPI_SWIGLU_BWD: return launch_gemm<BL_EPI_SWIGLU_BWD>(a, s);
    case BL_EPI_GELU_BWD: return launch_gemm<BL_EPI_GELU_BWD>(a, s);
    default: return BL_E_ARG;
  }
}

extern "C" int bl_gemm_skinny_rows_bf16(const bl_gemm_desc* d, void* stream) {
  GemmArgs a;
  const int rc = fill_gemm_args(d, a);
  if (rc != BL_OK) return rc;
  if (d->M > 128 || (d->K % 256) || d->a_norm_weight) return BL_E_SHAPE;   // the norm is its own launch: bl_rmsnorm_skinny_bf16
  if (d->out_group) return BL_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  switch (d->epilogue) {
    case BL_EPI_NONE: return launch_rows_sk<BL_EPI_NONE>(a, s);
    case BL_EPI_RES: return launch_rows_sk<BL_EPI_RES>(a, s);
    case BL_EPI_SWIGLU: return launch_rows_sk<BL_EPI_SWIGLU>(a, s);
    case BL_EPI_F32: return launch_rows_sk<BL_EPI_F32>(a, s);
    case BL_EPI_F32_BF16R: return launch_rows_sk<BL_EPI_F32_BF16R>(a, s);
    default: return BL_E_ARG;
  }
}

extern "C" int bl_gemm_tn_bf16(const bl_gemm_desc* d, void* stream) {
  if (!d || !d->A || !d->W || !d->C) return BL_E_ARG;
  if (d->epilogue != BL_EPI_F32 || d->out_group || d->a_norm_weight) return BL_E_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->M % 8) || (d->N % 8)) return BL_E_SHAPE;
  if ((d->lda % 8) || (d->ldw % 8) || (d->ldc % 4) || d->lda < d->M || d->ldw < d->N || d->ldc < d->N) return BL_E_ALIGN;
  if (!bl_aligned16(d->A) || !bl_aligned16(d->W) || (((uintptr_t)d->C) & 15)) return BL_E_ALIGN;
  // byte offsets inside the kernel are 32-bit (buffer addressing)
  if ((long)d->K * d->lda * 2 >= (1L << 31) || (long)d->K * d->ldw * 2 >= (1L << 31)) return BL_E_SHAPE;
  GemmArgs a = {};
  a.A = d->A; a.W = d->W; a.C = d->C;
  a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.tail_base = -1;
  a.slab = (float*)d->workspace; a.slab_bytes = d->workspace_bytes; a.splitk = 1;
  return launch_gemm_tn(a, (hipStream_t)stream);
}
