// gemm_common.h — argument block and fused epilogues shared by the tiled GEMM (gemm_bf16.hip) and the skinny
// weight-streaming GEMM (gemm_skinny.hip). Both kernels compute the MFMA tile TRANSPOSED, so a lane owns 4 consecutive
// output columns n..n+3 of one row m; the epilogue below is written for that layout.
#pragma once
#include "bl_common.h"

namespace blgemm {

struct GemmArgs {
  const uint16_t* A; const uint16_t* W; void* C;
  const uint16_t* bias; const uint16_t* scale; const uint16_t* res;
  long lda, ldw, ldc, ldres;
  int M, N, K;
  int res_row_mod, out_group, out_stride, out_offset;
  int tiles_m, tiles_n;
  const uint16_t* norm_w; float norm_eps;   // fused RMSNorm on A (skinny kernel)
  float* slab; long slab_bytes; int splitk;   // split-K tail of the 256x256 kernel (fp32 partial tiles)
  int tail_base;   // >= 0: this launch covers big (256x256) tiles tail_base.. of the tiles_m x tiles_n big-tile grid
  const float* qa; const float* qw;   // fp8 GEMM: per-row activation / per-column weight dequantisation scales
  int fold_ks;   // gemm_mid_kernel<SK>: MFMA k-steps per K-slice (skinny summation order); 0 = off
  uint16_t* C2; long ldc2;   // second output of the *_KEEP training epilogues
  int ptiles;   // gemm256s persistent form: > 0 = the launch walks this many tiles with one workgroup per CU (0 = one tile per workgroup)
};

__device__ __forceinline__ int out_row_of(const GemmArgs& p, int m) {
  if (p.out_group == 0) return m;
  const int g = m / p.out_group, r = m - g * p.out_group + p.out_offset;
  if (r < 0 || r >= p.out_stride) return -1;
  return g * p.out_stride + r;
}

// Epilogue for one lane's 4 consecutive columns n..n+3 of row m.
template <int EPI>
__device__ __forceinline__ void epilogue_store4(const GemmArgs& p, int m, int n, f32x4_t acc) {
  if (m >= p.M || n >= p.N) return;
  const int orow = out_row_of(p, m);
  if (orow < 0) return;
  if constexpr (EPI == BL_EPI_F32 || EPI == BL_EPI_F32_BF16R) {
    float* c = (float*)p.C + (long)orow * p.ldc + n;
    if constexpr (EPI == BL_EPI_F32_BF16R) acc = (f32x4_t){rbf(acc[0]), rbf(acc[1]), rbf(acc[2]), rbf(acc[3])};
    *(f32x4_t*)c = acc;
    return;
  } else if constexpr (EPI == BL_EPI_SWIGLU_KEEP) {
    // training forward of gate/up: the pre-activations stay (autograd's saved tensor), the activation goes to C2
    const float g0 = rbf(acc[0]), u0 = rbf(acc[1]), g1 = rbf(acc[2]), u1 = rbf(acc[3]);
    u32x2_t o; o[0] = pack2bf(g0, u0); o[1] = pack2bf(g1, u1);
    *(u32x2_t*)((uint16_t*)p.C + (long)orow * p.ldc + n) = o;
    *(uint32_t*)(p.C2 + (long)orow * p.ldc2 + (n >> 1)) = pack2bf(rbf(silu_f(g0)) * u0, rbf(silu_f(g1)) * u1);
    return;
  } else if constexpr (EPI == BL_EPI_SWIGLU_BWD) {
    // acc = dL/d act[m, n..n+3] (input gradient of down_proj); res = the saved gate/up pairs of those 4 columns
    float v[8], o[8];
    const u32x4_t q = *(const u32x4_t*)(p.res + (long)m * p.ldres + 2 * n);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[2 * i] = bflo(q[i]); v[2 * i + 1] = bfhi(q[i]); }
#pragma unroll
    for (int e = 0; e < 4; ++e) swiglu_bwd_pair(v[2 * e], v[2 * e + 1], rbf(acc[e]), o[2 * e], o[2 * e + 1]);
    u32x4_t w;
#pragma unroll
    for (int i = 0; i < 4; ++i) w[i] = pack2bf(o[2 * i], o[2 * i + 1]);
    *(u32x4_t*)((uint16_t*)p.C + (long)orow * p.ldc + 2 * n) = w;
    return;
  } else if constexpr (EPI == BL_EPI_GELU_BWD) {
    const u32x2_t t = *(const u32x2_t*)(p.res + (long)m * p.ldres + n);
    u32x2_t o;
    o[0] = pack2bf(rbf(acc[0]) * gelu_erf_grad(bflo(t[0])), rbf(acc[1]) * gelu_erf_grad(bfhi(t[0])));
    o[1] = pack2bf(rbf(acc[2]) * gelu_erf_grad(bflo(t[1])), rbf(acc[3]) * gelu_erf_grad(bfhi(t[1])));
    *(u32x2_t*)((uint16_t*)p.C + (long)orow * p.ldc + n) = o;
    return;
  } else if constexpr (EPI == BL_EPI_SWIGLU) {
    // rows 2j / 2j+1 of W are gate_j / up_j → regs (0,1) and (2,3) are (gate, up) pairs
    const float g0 = rbf(acc[0]), u0 = rbf(acc[1]), g1 = rbf(acc[2]), u1 = rbf(acc[3]);
    const float s0 = rbf(silu_f(g0)), s1 = rbf(silu_f(g1));
    uint16_t* c = (uint16_t*)p.C + (long)orow * p.ldc + (n >> 1);
    *(uint32_t*)c = pack2bf(s0 * u0, s1 * u1);
    return;
  } else {
    float v[4] = {acc[0], acc[1], acc[2], acc[3]};
    if constexpr (EPI == BL_EPI_BIAS || EPI == BL_EPI_BIAS_GELU || EPI == BL_EPI_BIAS_RES || EPI == BL_EPI_BIAS_GELU_KEEP) {
      const u32x2_t b = *(const u32x2_t*)(p.bias + n);
      v[0] += bflo(b[0]); v[1] += bfhi(b[0]); v[2] += bflo(b[1]); v[3] += bfhi(b[1]);
    }
    if constexpr (EPI == BL_EPI_BIAS_GELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gelu_erf(rbf(v[i]));
    }
    if constexpr (EPI == BL_EPI_BIAS_GELU_KEEP) {     // C keeps t = bf16(acc + bias), C2 gets gelu(t)
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = rbf(v[i]);
      u32x2_t a2; a2[0] = pack2bf(gelu_erf(v[0]), gelu_erf(v[1])); a2[1] = pack2bf(gelu_erf(v[2]), gelu_erf(v[3]));
      *(u32x2_t*)(p.C2 + (long)orow * p.ldc2 + n) = a2;
    }
    if constexpr (EPI == BL_EPI_BIAS_RES || EPI == BL_EPI_RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = rbf(v[i]);
      if (EPI == BL_EPI_BIAS_RES && p.scale != nullptr) {
        const u32x2_t s = *(const u32x2_t*)(p.scale + n);
        v[0] = rbf(v[0] * bflo(s[0])); v[1] = rbf(v[1] * bfhi(s[0]));
        v[2] = rbf(v[2] * bflo(s[1])); v[3] = rbf(v[3] * bfhi(s[1]));
      }
      const int rrow = p.res_row_mod ? (m % p.res_row_mod) : m;
      const u32x2_t r = *(const u32x2_t*)(p.res + (long)rrow * p.ldres + n);
      v[0] += bflo(r[0]); v[1] += bfhi(r[0]); v[2] += bflo(r[1]); v[3] += bfhi(r[1]);
    }
    uint16_t* c = (uint16_t*)p.C + (long)orow * p.ldc + n;
    u32x2_t o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
    *(u32x2_t*)c = o;
  }
}

// Whole-tile epilogue of the tile kernels (round 4). A lane owns NI column groups (4 consecutive columns from ncol[i]) × NJ
// rows (mrow + 16 j) of its workgroup's tile, which ends before row m_end / column n_end (wave-uniform). Interior tiles of
// plain problems — no output-row remap, no residual-row wrap, the whole tile inside M × N, byte offsets below 4 GiB — take a
// straight-line path: bias / LayerScale for all column groups loaded once up front, a row's residual loads issued together,
// 32-bit offsets against the uniform base pointers, no guards. Called per (i, j), epilogue_store4 spent ≈ 25 instructions
// around each 8-byte store (bounds guards, exec save / restore, the row-remap branch, two 64-bit multiplies) and exposed one
// bias / residual load latency per call behind its guards: + 4–5 µs per one-round GEMM for a bias add, + 13 µs with GELU
// (tools/bench_gemm_shapes.py). The arithmetic per element is the same statements in the same order as epilogue_store4's.
// Every epilogue family has the straight-line form (the training *_KEEP / *_BWD ones since the same round: fused SwiGLU / GELU
// forward and backward became a net win with it). Ragged edge tiles and remapped / wrapped rows go through epilogue_store4 as
// before. Outputs leave as 16-byte stores (store8 / store4 below), which requires ncol[i + 1] == ncol[i] + 16 inside each pair
// (8-byte outputs) or group of four (SwiGLU's 4-byte outputs) of column groups — true of every caller.
template <int EPI, int NI, int NJ, bool WIDE = true>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& p, int mrow, const int (&ncol)[NI], int m_end, int n_end,
                                              f32x4_t (&acc)[NI][NJ]) {
  constexpr bool kHasBias = EPI == BL_EPI_BIAS || EPI == BL_EPI_BIAS_GELU || EPI == BL_EPI_BIAS_RES || EPI == BL_EPI_BIAS_GELU_KEEP;
  constexpr bool kHasRes = EPI == BL_EPI_BIAS_RES || EPI == BL_EPI_RES;
  constexpr bool kReadsRes = kHasRes || EPI == BL_EPI_SWIGLU_BWD || EPI == BL_EPI_GELU_BWD;
  constexpr bool kHasC2 = EPI == BL_EPI_SWIGLU_KEEP || EPI == BL_EPI_BIAS_GELU_KEEP;
  constexpr bool kF32 = EPI == BL_EPI_F32 || EPI == BL_EPI_F32_BF16R;      // fp32 output (weight gradients, logits)
  // 16-byte stores need ncol[i + 1] == ncol[i] + 16 within each pair (8-byte outputs) / group of four (4-byte outputs)
  constexpr bool kWide2 = WIDE && (NI % 2) == 0, kWide4 = WIDE && (NI % 4) == 0;
  const bool plain = p.out_group == 0 && p.res_row_mod == 0 && m_end <= p.M && n_end <= p.N &&
                     (long)p.M * p.ldc * (kF32 ? 4 : 2) < (1L << 32) && (!kReadsRes || (long)p.M * p.ldres * 2 < (1L << 32)) &&
                     (!kHasC2 || (long)p.M * p.ldc2 * 2 < (1L << 32));
  if (plain) {
    u32x2_t bq[NI], sq[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      bq[i] = (u32x2_t){0u, 0u};
      sq[i] = (u32x2_t){0x3f803f80u, 0x3f803f80u};                 // 1.0: rbf(v · 1) == v for a bf16-valued v
      if constexpr (kHasBias) bq[i] = *(const u32x2_t*)(p.bias + ncol[i]);
      if constexpr (EPI == BL_EPI_BIAS_RES) {
        if (p.scale != nullptr) sq[i] = *(const u32x2_t*)(p.scale + ncol[i]);
      }
    }
    char* const cbase = (char*)p.C;
    char* const c2base = (char*)p.C2;
    const char* const rbase = (const char*)p.res;
    const int lg_ = (int)((threadIdx.x & 63u) >> 4);
    const uint32_t ldc = (uint32_t)p.ldc, ldres = (uint32_t)p.ldres, ldc2 = (uint32_t)p.ldc2;
    // 16-byte stores of 8-byte outputs: column groups i (even) and i + 1 lie 16 columns apart and a lane row lg owns columns
    // 4 lg … 4 lg + 3 of each. v_permlane16_swap trades the odd lane rows of group i against the even lane rows of group
    // i + 1, after which lane rows 0 / 2 hold columns 0-7 / 8-15 of group i and lane rows 1 / 3 those of group i + 1: one
    // dwordx4 per lane and pair instead of two dwordx2 — half the store instructions, 64 contiguous bytes per output row
    // and instruction instead of 32 (the one-round GEMMs' store tail).
    const int wide_dc = -4 * lg_ + (lg_ & 1) * 16 + (lg_ >> 1) * 8;      // this lane's first column after the swap, relative
    auto store8 = [&](char* base, uint32_t ld, uint32_t m, int i, u32x2_t o, u32x2_t& held) {
      if constexpr (kWide2) {
        if ((i & 1) == 0) {
          held = o;
        } else {
          const auto r0 = __builtin_amdgcn_permlane16_swap(held[0], o[0], false, false);
          const auto r1 = __builtin_amdgcn_permlane16_swap(held[1], o[1], false, false);
          const u32x4_t w = {r0[0], r1[0], r0[1], r1[1]};
          *(u32x4_t*)(base + (size_t)((m * ld + (uint32_t)(ncol[i - 1] + wide_dc)) * 2u)) = w;
        }
      } else {
        *(u32x2_t*)(base + (size_t)((m * ld + (uint32_t)ncol[i]) * 2u)) = o;
      }
    };
    // 16-byte stores of 4-byte outputs (SwiGLU activations, column n >> 1): a lane row lg owns ONE dword (activation columns
    // 2 lg, 2 lg + 1) of each of four column groups, which lie 8 activation columns apart: a 4 x 4 transpose over (group, lane
    // row) — permlane16_swap on the pairs (0, 1), (2, 3), then permlane32_swap on (0, 2), (1, 3) — leaves lane row r with the
    // four dwords of group r: one dwordx4 (8 columns) per lane instead of four dword stores
    const int wide_dc_sw = 6 * lg_;                                      // lane row r stores activation columns 8 r … 8 r + 7
    auto store4 = [&](char* base, uint32_t ld, uint32_t m, int i, uint32_t o, uint32_t (&sw)[4]) {
      if constexpr (kWide4) {
        sw[i & 3] = o;
        if ((i & 3) == 3) {
          const auto p01 = __builtin_amdgcn_permlane16_swap(sw[0], sw[1], false, false);
          const auto p23 = __builtin_amdgcn_permlane16_swap(sw[2], sw[3], false, false);
          const auto q02 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
          const auto q13 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
          const u32x4_t w = {q02[0], q13[0], q02[1], q13[1]};
          *(u32x4_t*)(base + (size_t)((m * ld + (uint32_t)((ncol[i - 3] >> 1) + wide_dc_sw)) * 2u)) = w;
        }
      } else {
        *(uint32_t*)(base + (size_t)((m * ld + ((uint32_t)ncol[i] >> 1)) * 2u)) = o;
      }
    };
    u32x2_t held = {0u, 0u}, held2 = {0u, 0u};
    uint32_t sw[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint32_t m = (uint32_t)(mrow + j * 16);
      u32x2_t rq[NI];
      u32x4_t rq4[NI];
      if constexpr (kHasRes || EPI == BL_EPI_GELU_BWD) {
#pragma unroll
        for (int i = 0; i < NI; ++i) rq[i] = *(const u32x2_t*)(rbase + (size_t)((m * ldres + (uint32_t)ncol[i]) * 2u));
      }
      if constexpr (EPI == BL_EPI_SWIGLU_BWD) {
#pragma unroll
        for (int i = 0; i < NI; ++i) rq4[i] = *(const u32x4_t*)(rbase + (size_t)((m * ldres + 2u * (uint32_t)ncol[i]) * 2u));
      }
#pragma unroll
      for (int i = 0; i < NI; ++i) {
        const f32x4_t a = acc[i][j];
        if constexpr (kF32) {
          f32x4_t o = a;
          if constexpr (EPI == BL_EPI_F32_BF16R) o = (f32x4_t){rbf(a[0]), rbf(a[1]), rbf(a[2]), rbf(a[3])};
          *(f32x4_t*)(cbase + (size_t)((m * ldc + (uint32_t)ncol[i]) * 4u)) = o;
        } else if constexpr (EPI == BL_EPI_SWIGLU) {
          const float g0 = rbf(a[0]), u0 = rbf(a[1]), g1 = rbf(a[2]), u1 = rbf(a[3]);
          const float s0 = rbf(silu_f(g0)), s1 = rbf(silu_f(g1));
          store4(cbase, ldc, m, i, pack2bf(s0 * u0, s1 * u1), sw);
        } else if constexpr (EPI == BL_EPI_SWIGLU_KEEP) {
          const float g0 = rbf(a[0]), u0 = rbf(a[1]), g1 = rbf(a[2]), u1 = rbf(a[3]);
          u32x2_t o; o[0] = pack2bf(g0, u0); o[1] = pack2bf(g1, u1);
          store8(cbase, ldc, m, i, o, held);
          store4(c2base, ldc2, m, i, pack2bf(rbf(silu_f(g0)) * u0, rbf(silu_f(g1)) * u1), sw);
        } else if constexpr (EPI == BL_EPI_SWIGLU_BWD) {
          float v[8], o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { v[2 * e] = bflo(rq4[i][e]); v[2 * e + 1] = bfhi(rq4[i][e]); }
#pragma unroll
          for (int e = 0; e < 4; ++e) swiglu_bwd_pair(v[2 * e], v[2 * e + 1], rbf(a[e]), o[2 * e], o[2 * e + 1]);
          u32x4_t w;
#pragma unroll
          for (int e = 0; e < 4; ++e) w[e] = pack2bf(o[2 * e], o[2 * e + 1]);
          *(u32x4_t*)(cbase + (size_t)((m * ldc + 2u * (uint32_t)ncol[i]) * 2u)) = w;
        } else if constexpr (EPI == BL_EPI_GELU_BWD) {
          u32x2_t o;
          o[0] = pack2bf(rbf(a[0]) * gelu_erf_grad(bflo(rq[i][0])), rbf(a[1]) * gelu_erf_grad(bfhi(rq[i][0])));
          o[1] = pack2bf(rbf(a[2]) * gelu_erf_grad(bflo(rq[i][1])), rbf(a[3]) * gelu_erf_grad(bfhi(rq[i][1])));
          store8(cbase, ldc, m, i, o, held);
        } else {
          float v[4] = {a[0], a[1], a[2], a[3]};
          if constexpr (kHasBias) {
            v[0] += bflo(bq[i][0]); v[1] += bfhi(bq[i][0]); v[2] += bflo(bq[i][1]); v[3] += bfhi(bq[i][1]);
          }
          if constexpr (EPI == BL_EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_erf(rbf(v[e]));
          }
          if constexpr (EPI == BL_EPI_BIAS_GELU_KEEP) {     // C keeps t = bf16(acc + bias), C2 gets gelu(t)
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
            u32x2_t a2; a2[0] = pack2bf(gelu_erf(v[0]), gelu_erf(v[1])); a2[1] = pack2bf(gelu_erf(v[2]), gelu_erf(v[3]));
            store8(c2base, ldc2, m, i, a2, held2);
          }
          if constexpr (kHasRes) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = rbf(v[e]);
            if constexpr (EPI == BL_EPI_BIAS_RES) {
              v[0] = rbf(v[0] * bflo(sq[i][0])); v[1] = rbf(v[1] * bfhi(sq[i][0]));
              v[2] = rbf(v[2] * bflo(sq[i][1])); v[3] = rbf(v[3] * bfhi(sq[i][1]));
            }
            v[0] += bflo(rq[i][0]); v[1] += bfhi(rq[i][0]); v[2] += bflo(rq[i][1]); v[3] += bfhi(rq[i][1]);
          }
          u32x2_t o; o[0] = pack2bf(v[0], v[1]); o[1] = pack2bf(v[2], v[3]);
          store8(cbase, ldc, m, i, o, held);
        }
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j) epilogue_store4<EPI>(p, mrow + j * 16, ncol[i], acc[i][j]);
}

// host side: validate a descriptor and copy it into the device argument block
inline int fill_gemm_args(const bl_gemm_desc* d, GemmArgs& a) {
  if (!d || !d->A || !d->W || !d->C) return BL_E_ARG;
  if (d->M <= 0 || d->N <= 0 || d->K <= 0 || (d->K % 64) != 0 || (d->N % 16) != 0) return BL_E_SHAPE;
  if ((d->lda % 8) || d->lda < d->K || d->ldw != d->K) return BL_E_ALIGN;   // W is packed: no leading dimension
  if (!bl_aligned16(d->A) || !bl_aligned16(d->W) || (((uintptr_t)d->C) & 15)) return BL_E_ALIGN;
  const int epi = d->epilogue;
  if (epi == BL_EPI_SWIGLU) { if ((d->ldc % 2) || (d->N % 32)) return BL_E_ALIGN; }
  else if (epi == BL_EPI_SWIGLU_BWD) { if (d->ldc % 8) return BL_E_ALIGN; }
  else if (d->ldc % 4) return BL_E_ALIGN;
  if ((epi == BL_EPI_BIAS || epi == BL_EPI_BIAS_GELU || epi == BL_EPI_BIAS_RES || epi == BL_EPI_BIAS_GELU_KEEP) && !d->bias) return BL_E_ARG;
  if ((epi == BL_EPI_BIAS_RES || epi == BL_EPI_RES || epi == BL_EPI_GELU_BWD) && (!d->res || (d->ldres % 4))) return BL_E_ARG;
  if (epi == BL_EPI_SWIGLU_BWD && (!d->res || (d->ldres % 8) || !bl_aligned16(d->res) || d->res_row_mod || d->out_group)) return BL_E_ARG;
  if (epi == BL_EPI_GELU_BWD && (d->res_row_mod || d->out_group)) return BL_E_ARG;
  if (epi == BL_EPI_SWIGLU_KEEP || epi == BL_EPI_BIAS_GELU_KEEP) {
    if (!d->C2 || (((uintptr_t)d->C2) & 15)) return BL_E_ARG;
    if (epi == BL_EPI_SWIGLU_KEEP ? ((d->ldc2 % 2) || (d->N % 32)) : (d->ldc2 % 4) != 0) return BL_E_ALIGN;
  }
  if (d->out_group < 0 || (d->out_group > 0 && d->out_stride <= 0)) return BL_E_SHAPE;
  a.A = d->A; a.W = d->W; a.C = d->C; a.bias = d->bias; a.scale = d->scale; a.res = d->res;
  a.lda = d->lda; a.ldw = d->ldw; a.ldc = d->ldc; a.ldres = d->ldres;
  a.M = d->M; a.N = d->N; a.K = d->K;
  a.res_row_mod = d->res_row_mod; a.out_group = d->out_group; a.out_stride = d->out_stride; a.out_offset = d->out_offset;
  a.tiles_m = a.tiles_n = 0;
  a.tail_base = -1;
  a.slab = (float*)d->workspace; a.slab_bytes = d->workspace_bytes; a.splitk = 1;
  a.norm_w = d->a_norm_weight; a.norm_eps = d->a_norm_eps;
  a.qa = a.qw = nullptr;
  a.fold_ks = 0;
  a.C2 = (uint16_t*)d->C2; a.ldc2 = d->ldc2;
  a.ptiles = 0;
  return BL_OK;
}

}  // namespace blgemm
