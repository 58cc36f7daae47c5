// attention_bwd.hip — backward of the whole-sequence attention (training step, SURVEY §8 a12: what
// `loss.backward()` runs under HF LlamaAttention / timm Attention, reference prismatic/training/strategies/base_strategy.py:300).
//
// The OpenVLA training sequences are short (256 patches + ≤ 64 prompt/action tokens; ViT 256/257), so — like the
// forward whole-sequence kernel — one workgroup (8 waves) owns one (batch, head) and keeps two of the four operands
// resident in LDS (2 × s_pad × 256 B ≤ 160 KiB), with the per-tile operands in registers:
//
//   dq kernel : K, V in LDS.  per 16-row query tile (one wave):  S^T = K·Q^T, dP^T = V·dO^T (both "swapped": the lane
//               owns one query column, so lse / delta are per-lane scalars), dS^T = P∘(dP − δ) stays in registers as
//               the B operand of dQ^T = K^T·dS^T (K^T through ds_read_b64_tr_b16).  Also emits δ = rowsum(dO∘O).
//   dkv kernel: Q, dO in LDS. per 16-row key tile (one wave):    S = Q·K^T, dP = dO·V^T (lane owns one key column),
//               P and dS feed dV^T = dO^T·P and dK^T = Q^T·dS as B operands (Q^T / dO^T through transposing reads).
//
// P is recomputed from the forward's base-2 log-sum-exp (bl_attention_lse_bf16): P = exp2(s·scale·log2e − lse).
// fp32 accumulation; P and dS are rounded to bf16 for the MFMAs (as autograd does under bf16 autocast).
// Attention backward is ≈ 1 % of the step's FLOPs (the GEMMs are 6·7e9·tokens), so the kernels favour simplicity:
// one LDS layout per staged operand (row reads conflict-free, transposing reads 2-way conflicted).
#include "bl_common.h"
#include <math.h>
#include <stdlib.h>

namespace bl_attention_bwd_impl {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct BwdArgs {
  const uint16_t *q, *k, *v, *o, *dout;
  uint16_t *dq, *dk, *dv;
  const uint8_t* mask;
  const float* lse;
  float* delta;
  long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, mask_bs;
  int B, H, Sq, Skv, stat_rs;
  float scale, scale_log2e;
};

// stage `rows` rows (HD bf16 each, row stride rs) into LDS at chunk' = chunk ^ (row & MASK); rows ≥ n_valid and pad
// chunks are zero
template <int HD, int ROWB>
__device__ __forceinline__ void stage_two(char* a_lds, char* b_lds, const uint16_t* a, long a_rs, const uint16_t* b, long b_rs,
                                          int s_pad, int n_valid, int tid) {
  constexpr int CH = ROWB / 16, KCH = HD / 8, MASK = CH - 1, UNR = 6;
  const int npieces = s_pad * CH;
  for (int base = tid; base < npieces; base += 512 * UNR) {
    u32x4_t aq[UNR], bq[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int piece = base + u * 512, row = piece / CH, ch = piece - row * CH;
      aq[u] = (u32x4_t){0u, 0u, 0u, 0u};
      bq[u] = (u32x4_t){0u, 0u, 0u, 0u};
      if (piece < npieces && row < n_valid && ch < KCH) {
        aq[u] = *(const u32x4_t*)(a + (long)row * a_rs + ch * 8);
        bq[u] = *(const u32x4_t*)(b + (long)row * b_rs + ch * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int piece = base + u * 512, row = piece / CH, ch = piece - row * CH;
      if (piece < npieces) {
        *(u32x4_t*)(a_lds + row * ROWB + ((ch ^ (row & MASK)) << 4)) = aq[u];
        *(u32x4_t*)(b_lds + row * ROWB + ((ch ^ (row & MASK)) << 4)) = bq[u];
      }
    }
  }
}

// transposed 16(d) × 32(row) fragment of a row-major LDS operand: MFMA "A" operand T[d = 16*dt + l15][rows row0 ..],
// element j of lane group lg ↔ row row0 + 16*(j>>2) + 4*lg + (j&3)
template <int ROWB>
__device__ __forceinline__ bf16x8_t tr_frag(const char* lds, int row0, int dt, int l15, int lg) {
  constexpr int MASK = ROWB / 16 - 1;
  const int row = row0 + 4 * lg + (l15 >> 2);                 // (row + 16) & MASK == row & MASK
  const int ch = dt * 2 + ((l15 & 3) >> 1);
  const char* vp = lds + row * ROWB + ((ch ^ (row & MASK)) << 4) + (l15 & 1) * 8;
  const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)vp);
  const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 16 * ROWB));
  const u32x2_t w0 = __builtin_bit_cast(u32x2_t, v0), w1 = __builtin_bit_cast(u32x2_t, v1);
  const u32x4_t vv = {w0[0], w0[1], w1[0], w1[1]};
  return __builtin_bit_cast(bf16x8_t, vv);
}

template <int ROWB>
__device__ __forceinline__ bf16x8_t row_frag(const char* lds, int row, int ch) {
  constexpr int MASK = ROWB / 16 - 1;
  return *(const bf16x8_t*)(lds + row * ROWB + ((ch ^ (row & MASK)) << 4));
}

__device__ __forceinline__ bf16x8_t pack8(const float* e) {
  u32x4_t t;
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = pack2bf(e[2 * j], e[2 * j + 1]);
  return __builtin_bit_cast(bf16x8_t, t);
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dq_kernel(BwdArgs p, int s_pad) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HDP = (HD + 31) / 32 * 32, KS = HDP / 32, KCH = HD / 8;
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  constexpr int DT = (HD + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + s_pad * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int off = p.Skv - p.Sq;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;
  stage_two<HD, ROWB>(k_lds, v_lds, p.k + (long)b * p.k_bs + (long)h * p.k_hs, p.k_rs,
                      p.v + (long)b * p.v_bs + (long)h * p.v_hs, p.v_rs, s_pad, p.Skv, tid);
  __syncthreads();
  // key visibility (key < Skv and not hidden by the key-padding mask) of this lane's keys kt·16 + lg·4 + {0..3}, one bit each
  // (bit kt·4 + rr of an 80-bit set): read ONCE per workgroup — the score loop used to load one mask byte per score from
  // global memory and wait for it (8 exposed L2 round trips per 32-key step, the dominant cost of this kernel in training)
  uint32_t mb0 = 0, mb1 = 0, mb2 = 0;
#pragma unroll
  for (int kt = 0; kt < 20; ++kt) {
    uint32_t w = 0;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int key = kt * 16 + lg * 4 + rr;
      if (key < p.Skv && (mrow == nullptr || mrow[key] != 0)) w |= 1u << rr;
    }
    if (kt < 8) mb0 |= w << (kt * 4); else if (kt < 16) mb1 |= w << ((kt - 8) * 4); else mb2 |= w << ((kt - 16) * 4);
  }

  const int nqt = (p.Sq + 15) >> 4;
  for (int r = 0; r * 8 < nqt; ++r) {
    const int idx = r * 8 + ((r & 1) ? 7 - wave : wave);
    if (idx >= nqt) continue;
    const int qt = nqt - 1 - idx, q0 = qt * 16, qrow = q0 + l15;
    int kv_hi = p.Skv;
    if (CAUSAL) kv_hi = min(p.Skv, q0 + 16 + off);

    bf16x8_t qf[KS], dof[KS];
    float dsum = 0.f;
    {
      const uint16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
      const long oo = (long)b * p.o_bs + (long)h * p.o_hs + (long)qrow * p.o_rs;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = lg + 4 * ks;
        u32x4_t t = {0u, 0u, 0u, 0u}, g = {0u, 0u, 0u, 0u}, ov = {0u, 0u, 0u, 0u};
        if (qrow < p.Sq && ch < KCH) {
          t = *(const u32x4_t*)(qp + ch * 8);
          g = *(const u32x4_t*)(p.dout + oo + ch * 8);
          ov = *(const u32x4_t*)(p.o + oo + ch * 8);
        }
        qf[ks] = __builtin_bit_cast(bf16x8_t, t);
        dof[ks] = __builtin_bit_cast(bf16x8_t, g);
#pragma unroll
        for (int i = 0; i < 4; ++i) dsum += bflo(g[i]) * bflo(ov[i]) + bfhi(g[i]) * bfhi(ov[i]);
      }
    }
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    const long stat = ((long)b * p.H + h) * p.stat_rs + qrow;
    if (lg == 0 && qrow < p.Sq) p.delta[stat] = dsum;
    const float lse_q = qrow < p.Sq ? p.lse[stat] : INFINITY;

    f32x4_t acc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 1   // one scheduled body: unrolled, hipcc reuses ONE fragment buffer from the second step on (read, wait, MFMA, read, …)
    for (int s2 = 0; s2 * 32 < kv_hi; ++s2) {
      float e[8];
      const uint32_t vis8 = ((s2 < 4 ? mb0 : (s2 < 8 ? mb1 : mb2)) >> ((s2 & 3) * 8)) & 0xffu;   // bits half·4 + rr of this step
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * s2 + half;
        f32x4_t as = {0.f, 0.f, 0.f, 0.f}, ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(k_lds, kt * 16 + l15, lg + 4 * ks), qf[ks], as, 0, 0, 0);
          ap = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(v_lds, kt * 16 + l15, lg + 4 * ks), dof[ks], ap, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int key = kt * 16 + lg * 4 + rr;
          bool vis = ((vis8 >> (half * 4 + rr)) & 1u) != 0;
          if (CAUSAL) vis = vis && (key <= qrow + off);
          const float pr = vis ? __builtin_amdgcn_exp2f(as[rr] * p.scale_log2e - lse_q) : 0.f;
          e[half * 4 + rr] = pr * (ap[rr] - dsum);
        }
      }
      const bf16x8_t dsf = pack8(e);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(k_lds, 32 * s2, dt, l15, lg), dsf, acc[dt], 0, 0, 0);
    }
    if (qrow < p.Sq) {
      uint16_t* op = p.dq + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < HD) {
          u32x2_t w;
          w[0] = pack2bf(acc[dt][0] * p.scale, acc[dt][1] * p.scale);
          w[1] = pack2bf(acc[dt][2] * p.scale, acc[dt][3] * p.scale);
          *(u32x2_t*)(op + d) = w;
        }
      }
    }
  }
#endif
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dkv_kernel(BwdArgs p, int s_pad) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HDP = (HD + 31) / 32 * 32, KS = HDP / 32, KCH = HD / 8;
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  constexpr int DT = (HD + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_lds = smem;
  char* g_lds = smem + s_pad * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int off = p.Skv - p.Sq;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;
  stage_two<HD, ROWB>(q_lds, g_lds, p.q + (long)b * p.q_bs + (long)h * p.q_hs, p.q_rs,
                      p.dout + (long)b * p.o_bs + (long)h * p.o_hs, p.o_rs, s_pad, p.Sq, tid);
  __syncthreads();
  const float* lse = p.lse + ((long)b * p.H + h) * p.stat_rs;
  const float* delta = p.delta + ((long)b * p.H + h) * p.stat_rs;

  const int nkt = (p.Skv + 15) >> 4;
  for (int r = 0; r * 8 < nkt; ++r) {
    const int kt = r * 8 + ((r & 1) ? 7 - wave : wave);   // causal: low key tiles are the heavy ones
    if (kt >= nkt) continue;
    const int k0 = kt * 16, key = k0 + l15;
    bool key_ok = key < p.Skv;
    if (mrow) key_ok = key_ok && (key < p.Skv ? mrow[key] != 0 : false);

    bf16x8_t kf[KS], vf[KS];
    {
      const uint16_t* kp = p.k + (long)b * p.k_bs + (long)h * p.k_hs + (long)key * p.k_rs;
      const uint16_t* vp = p.v + (long)b * p.v_bs + (long)h * p.v_hs + (long)key * p.v_rs;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = lg + 4 * ks;
        u32x4_t t = {0u, 0u, 0u, 0u}, u = {0u, 0u, 0u, 0u};
        if (key < p.Skv && ch < KCH) {
          t = *(const u32x4_t*)(kp + ch * 8);
          u = *(const u32x4_t*)(vp + ch * 8);
        }
        kf[ks] = __builtin_bit_cast(bf16x8_t, t);
        vf[ks] = __builtin_bit_cast(bf16x8_t, u);
      }
    }
    f32x4_t dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) dk[i] = dv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int q_lo = 0;
    if (CAUSAL) q_lo = max(0, k0 - off);                  // first query row that can see this key tile
    // lse / delta of the query rows of step s2 are fetched one step ahead (their global-load latency used to sit in front
    // of every step's softmax recomputation)
    f32x4_t ln[2], dn[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      ln[half] = *(const f32x4_t*)(lse + (2 * (q_lo >> 5) + half) * 16 + lg * 4);
      dn[half] = *(const f32x4_t*)(delta + (2 * (q_lo >> 5) + half) * 16 + lg * 4);
    }
#pragma unroll 1
    for (int s2 = q_lo >> 5; s2 * 32 < p.Sq; ++s2) {
      float pe[8], de[8];
      f32x4_t lc[2] = {ln[0], ln[1]}, dc[2] = {dn[0], dn[1]};
      if ((s2 + 1) * 32 < p.Sq) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          ln[half] = *(const f32x4_t*)(lse + (2 * (s2 + 1) + half) * 16 + lg * 4);
          dn[half] = *(const f32x4_t*)(delta + (2 * (s2 + 1) + half) * 16 + lg * 4);
        }
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * s2 + half;
        f32x4_t as = {0.f, 0.f, 0.f, 0.f}, ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(q_lds, qt * 16 + l15, lg + 4 * ks), kf[ks], as, 0, 0, 0);
          ap = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(g_lds, qt * 16 + l15, lg + 4 * ks), vf[ks], ap, 0, 0, 0);
        }
        const f32x4_t l4 = lc[half], d4 = dc[half];                        // stat rows are padded to 32
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int q = qt * 16 + lg * 4 + rr;
          bool vis = key_ok && q < p.Sq;
          if (CAUSAL) vis = vis && (key <= q + off);
          const float pr = vis ? __builtin_amdgcn_exp2f(as[rr] * p.scale_log2e - l4[rr]) : 0.f;
          pe[half * 4 + rr] = pr;
          de[half * 4 + rr] = vis ? pr * (ap[rr] - d4[rr]) : 0.f;
        }
      }
      const bf16x8_t pf = pack8(pe), dsf = pack8(de);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(g_lds, 32 * s2, dt, l15, lg), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(q_lds, 32 * s2, dt, l15, lg), dsf, dk[dt], 0, 0, 0);
      }
    }
    if (key < p.Skv) {
      uint16_t* kp = p.dk + (long)b * p.k_bs + (long)h * p.k_hs + (long)key * p.k_rs;
      uint16_t* vp = p.dv + (long)b * p.v_bs + (long)h * p.v_hs + (long)key * p.v_rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < HD) {
          u32x2_t w;
          w[0] = pack2bf(dk[dt][0] * p.scale, dk[dt][1] * p.scale);
          w[1] = pack2bf(dk[dt][2] * p.scale, dk[dt][3] * p.scale);
          *(u32x2_t*)(kp + d) = w;
          w[0] = pack2bf(dv[dt][0], dv[dt][1]);
          w[1] = pack2bf(dv[dt][2], dv[dt][3]);
          *(u32x2_t*)(vp + d) = w;
        }
      }
    }
  }
#endif
}

template <int HD, bool CAUSAL>
int launch_bwd(const BwdArgs& a, hipStream_t s) {
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_kernel<HD, CAUSAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_kernel<HD, CAUSAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return BL_E_LAUNCH;
    attr_set = true;
  }
  const int kv_pad = (a.Skv + 31) / 32 * 32, q_pad = (a.Sq + 31) / 32 * 32;
  hipLaunchKernelGGL((attn_bwd_dq_kernel<HD, CAUSAL>), dim3(a.B * a.H), dim3(512), 2 * kv_pad * ROWB, s, a, kv_pad);
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<HD, CAUSAL>), dim3(a.B * a.H), dim3(512), 2 * q_pad * ROWB, s, a, q_pad);
  return BL_OK;
}


// ---- long sequences (Sq or Skv > 320: prompts beyond 55 tokens; the collator pads up to model_max_length = 2048,
// prismatic/util/data_utils.py:101-142, configuration_prismatic.py:84) ----
// Same arithmetic, tiled the other way round: a workgroup owns 8 tiles of the REGISTER-side operand (grid.y blocks of 128
// query rows for dq, 128 key rows for dk/dv; one 16-row tile per wave, accumulators live in registers for the whole kernel)
// and the LDS-side operand streams through in chunks of KC = 256 rows. P is recomputed from the stored log-sum-exp, so
// chunks simply accumulate — no running max, no rescaling; the summation order over keys (dq) / queries (dk, dv) is the
// whole-sequence kernels' order, so both forms give the same bits (tests/test_train_ops_gpu.py runs them against each other).
constexpr int KC = 256;

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dq_chunk_kernel(BwdArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HDP = (HD + 31) / 32 * 32, KS = HDP / 32, KCH = HD / 8;
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  constexpr int DT = (HD + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + KC * ROWB;
  uint8_t* m_lds = (uint8_t*)(smem + 2 * KC * ROWB);          // key visibility of the staged chunk, one byte per key
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int off = p.Skv - p.Sq;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;
  const uint16_t* kbase = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const uint16_t* vbase = p.v + (long)b * p.v_bs + (long)h * p.v_hs;

  const int q0 = (blockIdx.y * 8 + wave) * 16, qrow = q0 + l15;          // tiles past the end: every guard below is false
  int kv_hi = q0 < p.Sq ? p.Skv : 0, block_hi = p.Skv;
  if (CAUSAL) {
    kv_hi = q0 < p.Sq ? min(p.Skv, q0 + 16 + off) : 0;
    block_hi = min(p.Skv, (int)blockIdx.y * 128 + 128 + off);
  }
  bf16x8_t qf[KS], dof[KS];
  float dsum = 0.f;
  {
    const uint16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
    const long oo = (long)b * p.o_bs + (long)h * p.o_hs + (long)qrow * p.o_rs;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = lg + 4 * ks;
      u32x4_t t = {0u, 0u, 0u, 0u}, g = {0u, 0u, 0u, 0u}, ov = {0u, 0u, 0u, 0u};
      if (qrow < p.Sq && ch < KCH) {
        t = *(const u32x4_t*)(qp + ch * 8);
        g = *(const u32x4_t*)(p.dout + oo + ch * 8);
        ov = *(const u32x4_t*)(p.o + oo + ch * 8);
      }
      qf[ks] = __builtin_bit_cast(bf16x8_t, t);
      dof[ks] = __builtin_bit_cast(bf16x8_t, g);
#pragma unroll
      for (int i = 0; i < 4; ++i) dsum += bflo(g[i]) * bflo(ov[i]) + bfhi(g[i]) * bfhi(ov[i]);
    }
  }
  dsum += __shfl_xor(dsum, 16, 64);
  dsum += __shfl_xor(dsum, 32, 64);
  const long stat = ((long)b * p.H + h) * p.stat_rs + qrow;
  if (lg == 0 && qrow < p.Sq) p.delta[stat] = dsum;
  const float lse_q = qrow < p.Sq ? p.lse[stat] : INFINITY;

  f32x4_t acc[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  for (int key0 = 0; key0 < block_hi; key0 += KC) {
    __syncthreads();                                              // the previous chunk has been consumed by every wave
    stage_two<HD, ROWB>(k_lds, v_lds, kbase + (long)key0 * p.k_rs, p.k_rs, vbase + (long)key0 * p.v_rs, p.v_rs, KC,
                        min(KC, p.Skv - key0), tid);
    if (tid < KC) m_lds[tid] = (key0 + tid < p.Skv && (mrow == nullptr || mrow[key0 + tid] != 0)) ? 1 : 0;
    __syncthreads();
#pragma unroll 1
    for (int s2 = 0; s2 < KC / 32 && key0 + s2 * 32 < kv_hi; ++s2) {
      float e[8];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int kt = 2 * s2 + half;
        f32x4_t as = {0.f, 0.f, 0.f, 0.f}, ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(k_lds, kt * 16 + l15, lg + 4 * ks), qf[ks], as, 0, 0, 0);
          ap = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(v_lds, kt * 16 + l15, lg + 4 * ks), dof[ks], ap, 0, 0, 0);
        }
        const uint32_t vis4 = *(const uint32_t*)(m_lds + kt * 16 + lg * 4);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int key = key0 + kt * 16 + lg * 4 + rr;
          bool vis = ((vis4 >> (8 * rr)) & 0xffu) != 0;
          if (CAUSAL) vis = vis && (key <= qrow + off);
          const float pr = vis ? __builtin_amdgcn_exp2f(as[rr] * p.scale_log2e - lse_q) : 0.f;
          e[half * 4 + rr] = pr * (ap[rr] - dsum);
        }
      }
      const bf16x8_t dsf = pack8(e);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(k_lds, 32 * s2, dt, l15, lg), dsf, acc[dt], 0, 0, 0);
    }
  }
  if (qrow < p.Sq) {
    uint16_t* op = p.dq + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < HD) {
        u32x2_t w;
        w[0] = pack2bf(acc[dt][0] * p.scale, acc[dt][1] * p.scale);
        w[1] = pack2bf(acc[dt][2] * p.scale, acc[dt][3] * p.scale);
        *(u32x2_t*)(op + d) = w;
      }
    }
  }
#endif
}

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(512) void attn_bwd_dkv_chunk_kernel(BwdArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int HDP = (HD + 31) / 32 * 32, KS = HDP / 32, KCH = HD / 8;
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  constexpr int DT = (HD + 15) / 16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* q_lds = smem;
  char* g_lds = smem + KC * ROWB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int off = p.Skv - p.Sq;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;
  const uint16_t* qbase = p.q + (long)b * p.q_bs + (long)h * p.q_hs;
  const uint16_t* gbase = p.dout + (long)b * p.o_bs + (long)h * p.o_hs;
  const float* lse = p.lse + ((long)b * p.H + h) * p.stat_rs;
  const float* delta = p.delta + ((long)b * p.H + h) * p.stat_rs;

  const int k0 = (blockIdx.y * 8 + wave) * 16, key = k0 + l15;
  bool key_ok = key < p.Skv;
  if (mrow) key_ok = key_ok && (key < p.Skv ? mrow[key] != 0 : false);
  bf16x8_t kf[KS], vf[KS];
  {
    const uint16_t* kp = p.k + (long)b * p.k_bs + (long)h * p.k_hs + (long)key * p.k_rs;
    const uint16_t* vp = p.v + (long)b * p.v_bs + (long)h * p.v_hs + (long)key * p.v_rs;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = lg + 4 * ks;
      u32x4_t t = {0u, 0u, 0u, 0u}, u = {0u, 0u, 0u, 0u};
      if (key < p.Skv && ch < KCH) {
        t = *(const u32x4_t*)(kp + ch * 8);
        u = *(const u32x4_t*)(vp + ch * 8);
      }
      kf[ks] = __builtin_bit_cast(bf16x8_t, t);
      vf[ks] = __builtin_bit_cast(bf16x8_t, u);
    }
  }
  f32x4_t dk[DT], dv[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) dk[i] = dv[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  // first 32-row query step this wave / this workgroup needs (causal: rows below the key tile see nothing of it)
  int s2_wave = 0, s2_block = 0;
  if (CAUSAL) {
    s2_wave = max(0, k0 - off) >> 5;
    s2_block = max(0, (int)blockIdx.y * 128 - off) >> 5;
  }
  if (k0 >= p.Skv) s2_wave = (p.Sq + 31) >> 5;                       // tile past the end: no steps
  for (int qc0 = (s2_block * 32) / KC * KC; qc0 < p.Sq; qc0 += KC) {
    __syncthreads();
    stage_two<HD, ROWB>(q_lds, g_lds, qbase + (long)qc0 * p.q_rs, p.q_rs, gbase + (long)qc0 * p.o_rs, p.o_rs, KC,
                        min(KC, p.Sq - qc0), tid);
    __syncthreads();
#pragma unroll 1
    for (int s2 = max(s2_wave, qc0 >> 5); s2 * 32 < min(p.Sq, qc0 + KC); ++s2) {
      float pe[8], de[8];
      const int lrow = s2 * 32 - qc0;                                // row of this step inside the staged chunk
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int qt = 2 * s2 + half;
        const f32x4_t l4 = *(const f32x4_t*)(lse + qt * 16 + lg * 4);     // stat rows are padded to 32
        const f32x4_t d4 = *(const f32x4_t*)(delta + qt * 16 + lg * 4);
        f32x4_t as = {0.f, 0.f, 0.f, 0.f}, ap = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          as = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(q_lds, lrow + half * 16 + l15, lg + 4 * ks), kf[ks], as, 0, 0, 0);
          ap = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag<ROWB>(g_lds, lrow + half * 16 + l15, lg + 4 * ks), vf[ks], ap, 0, 0, 0);
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int q = qt * 16 + lg * 4 + rr;
          bool vis = key_ok && q < p.Sq;
          if (CAUSAL) vis = vis && (key <= q + off);
          const float pr = vis ? __builtin_amdgcn_exp2f(as[rr] * p.scale_log2e - l4[rr]) : 0.f;
          pe[half * 4 + rr] = pr;
          de[half * 4 + rr] = vis ? pr * (ap[rr] - d4[rr]) : 0.f;
        }
      }
      const bf16x8_t pf = pack8(pe), dsf = pack8(de);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(g_lds, lrow, dt, l15, lg), pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag<ROWB>(q_lds, lrow, dt, l15, lg), dsf, dk[dt], 0, 0, 0);
      }
    }
  }
  if (key < p.Skv) {
    uint16_t* kp = p.dk + (long)b * p.k_bs + (long)h * p.k_hs + (long)key * p.k_rs;
    uint16_t* vp = p.dv + (long)b * p.v_bs + (long)h * p.v_hs + (long)key * p.v_rs;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < HD) {
        u32x2_t w;
        w[0] = pack2bf(dk[dt][0] * p.scale, dk[dt][1] * p.scale);
        w[1] = pack2bf(dk[dt][2] * p.scale, dk[dt][3] * p.scale);
        *(u32x2_t*)(kp + d) = w;
        w[0] = pack2bf(dv[dt][0], dv[dt][1]);
        w[1] = pack2bf(dv[dt][2], dv[dt][3]);
        *(u32x2_t*)(vp + d) = w;
      }
    }
  }
#endif
}

template <int HD, bool CAUSAL>
int launch_bwd_chunked(const BwdArgs& a, hipStream_t s) {
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  constexpr int LDS = 2 * KC * ROWB + KC;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dq_chunk_kernel<HD, CAUSAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_dkv_chunk_kernel<HD, CAUSAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return BL_E_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((attn_bwd_dq_chunk_kernel<HD, CAUSAL>), dim3(a.B * a.H, (a.Sq + 127) / 128), dim3(512), LDS, s, a);
  hipLaunchKernelGGL((attn_bwd_dkv_chunk_kernel<HD, CAUSAL>), dim3(a.B * a.H, (a.Skv + 127) / 128), dim3(512), LDS, s, a);
  return BL_OK;
}

}  // namespace bl_attention_bwd_impl
using namespace bl_attention_bwd_impl;

extern "C" int bl_attention_backward_bf16(const bl_attn_desc* d, const bl_bf16* dout, const float* lse, float* delta,
                                          bl_bf16* dq, bl_bf16* dk, bl_bf16* dv, void* stream) {
  if (!d || !d->q || !d->k || !d->v || !d->o || !dout || !lse || !delta || !dq || !dk || !dv) return BL_E_ARG;
  if (d->B <= 0 || d->H <= 0 || d->Sq <= 0 || d->Skv <= 0) return BL_E_SHAPE;
  if (d->causal && d->Skv < d->Sq) return BL_E_SHAPE;
  if ((long)d->B * d->H > 0x7fffffffL || (d->Sq + 127) / 128 > 65535 || (d->Skv + 127) / 128 > 65535) return BL_E_SHAPE;
  const int64_t st[] = {d->q_bs, d->q_hs, d->q_rs, d->k_bs, d->k_hs, d->k_rs, d->v_bs, d->v_hs, d->v_rs, d->o_bs, d->o_hs, d->o_rs};
  for (int64_t s : st) if (s % 8) return BL_E_ALIGN;
  const void* ptrs[] = {d->q, d->k, d->v, d->o, dout, dq, dk, dv, lse, delta};
  for (const void* x : ptrs) if (!bl_aligned16(x)) return BL_E_ALIGN;
  BwdArgs a;
  a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o; a.dout = dout; a.dq = dq; a.dk = dk; a.dv = dv;
  a.mask = d->key_mask; a.lse = lse; a.delta = delta;
  a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_rs = d->k_rs;
  a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_rs = d->v_rs; a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_rs = d->o_rs;
  a.mask_bs = d->mask_bs;
  a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Skv = d->Skv; a.stat_rs = (d->Sq + 31) / 32 * 32;
  a.scale = d->scale; a.scale_log2e = d->scale * 1.44269504088896340736f;
  hipStream_t s = (hipStream_t)stream;
  int r = BL_E_SHAPE;
  const bool chunked_only = getenv("BL_ATTN_BWD_CHUNKED") != nullptr;    // test aid: both forms on one shape (bit-identical)
  // short sequences (every OpenVLA batch whose prompts stay under 56 tokens): one workgroup per (batch, head) with the
  // LDS-side operand of the whole sequence resident; longer ones stream it in 256-row chunks
  const bool whole = d->Sq <= 320 && d->Skv <= 320 && !chunked_only;
#define BL_BWD_CASE(HD)                                                                                             \
  case HD:                                                                                                          \
    if (whole) r = d->causal ? launch_bwd<HD, true>(a, s) : launch_bwd<HD, false>(a, s);                            \
    else r = d->causal ? launch_bwd_chunked<HD, true>(a, s) : launch_bwd_chunked<HD, false>(a, s);                  \
    break;
  switch (d->head_dim) { BL_BWD_CASE(64) BL_BWD_CASE(72) BL_BWD_CASE(128) default: return BL_E_SHAPE; }
#undef BL_BWD_CASE
  if (r != BL_OK) return r;
  BL_CHECK_LAUNCH();
  return BL_OK;
}
