// attention.hip — flash-style attention for the three head sizes on the OpenVLA path (64 DINOv2, 72 SigLIP,
// 128 Llama-2) plus the single-query decode kernel.
//
// Prefill / ViT kernel (attn_fwd_kernel):
//   * one workgroup = 4 waves = 64 query rows of one (batch, head); key/value chunks of 64 keys stream through LDS;
//     online softmax (running max / sum) so the sequence length is unbounded.
//   * QK^T is computed SWAPPED (S^T = K·Q^T: K rows are the MFMA "A" operand, Q the "B" operand), so a lane owns one
//     query column and 4 keys per 16-key tile: the softmax row reductions are 15 in-lane ops + two cross-lane
//     shuffles (xor 16, 32), and the fp32 scores are already laid out as the "B" operand of the PV product
//     O^T = V^T·P^T — no LDS round trip for P (guide §3, "accumulator tile as the next MFMA's operand").
//   * V is staged row-major exactly like K; the PV "A" operand (V^T) comes from gfx950's transposing LDS read
//     ds_read_b64_tr_b16 (4 keys x 16 d per 16-lane group), two per MFMA — no transpose pass, no 2-byte LDS writes.
//   * head_dim 72 is zero-padded in LDS to K=96 for QK^T and to 80 output rows for PV; no padded bytes touch HBM.
//   * numerics: fp32 scores, fp32 exp/sum, P rounded to bf16 for the PV MFMA, O = bf16(acc / sum) — restated op for
//     op by oracle/restate.py::attention.
#include "bl_common.h"
#include <math.h>
#include <stdlib.h>

namespace bl_attention_impl {
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

struct AttnArgs {
  const uint16_t* q; const uint16_t* k; const uint16_t* v; uint16_t* o; const uint8_t* mask;
  long q_bs, q_hs, q_rs, k_bs, k_hs, k_rs, v_bs, v_hs, v_rs, o_bs, o_hs, o_rs, mask_bs;
  int B, H, Sq, Skv;
  float scale_log2e;
  float* lse;      // optional [B*H, lse_rs] base-2 log-sum-exp of the scaled scores (training forward); +inf for empty rows
  int lse_rs;
  const int* rope_pos;   // decode kernel, optional [B]: per-sequence position (right-padded prompts) = cache row, rotation, length - 1
  // fused RoPE + KV-cache write (whole-sequence kernel, head_dim 128): q / k are rotated while they are loaded, the rotated
  // k and v rows are also written to the caches [B, H, cache_len, 128] at positions pos0 + key
  const uint16_t* cos_tab; const uint16_t* sin_tab; uint16_t* k_cache; uint16_t* v_cache; int cache_len, pos0;
};

// HF apply_rotary_pos_emb on one 8-element chunk pair (x1 = dims c..c+7, x2 = dims 64+c..): the arithmetic (three bf16
// roundings per element) of rope_kvcache_kernel in glue.hip, bit for bit.
__device__ __forceinline__ void rope_pair(const u32x4_t x1, const u32x4_t x2, const u32x4_t cq, const u32x4_t sq, u32x4_t& o1,
                                          u32x4_t& o2) {
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    float r1[2], r2[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const float a = e ? bfhi(x1[w]) : bflo(x1[w]);
      const float c2 = e ? bfhi(x2[w]) : bflo(x2[w]);
      const float co = e ? bfhi(cq[w]) : bflo(cq[w]);
      const float si = e ? bfhi(sq[w]) : bflo(sq[w]);
      r1[e] = rbf(a * co) + rbf(-c2 * si);
      r2[e] = rbf(c2 * co) + rbf(a * si);
    }
    o1[w] = pack2bf(r1[0], r1[1]);
    o2[w] = pack2bf(r2[0], r2[1]);
  }
}

constexpr int KV_CHUNK = 64;
constexpr int VROW = KV_CHUNK * 2 + 16;   // bytes per V^T row (64 keys + 16 B pad → conflict-free ds_read_b64)

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnArgs p) {
  constexpr int HDP = (HD + 31) / 32 * 32;     // QK^T reduction length (zero padded)
  constexpr int KS = HDP / 32;                 // MFMA k-steps for QK^T
  constexpr int KCH = HD / 8;                  // 16-byte chunks per K/V row in HBM
  constexpr int KROW = HDP * 2 + 16;           // LDS bytes per K row (padded against bank conflicts)
  constexpr int DT = (HD + 15) / 16;           // 16-row output tiles of O^T
  static_assert(HD % 8 == 0, "head_dim must be a multiple of 8");

  __shared__ __attribute__((aligned(16))) char k_lds[KV_CHUNK * KROW];
  __shared__ __attribute__((aligned(16))) char v_lds[KV_CHUNK * KROW];   // row-major like K; read transposed (tr_b16)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * 64;
  const int qrow = q0 + wave * 16 + l15;
  const int off = p.Skv - p.Sq;                // causal: key j visible to query i iff j <= i + off

  // zero the LDS padding once (never overwritten by the staging loops)
  for (int i = tid; i < (int)sizeof(k_lds) / 16; i += 256) ((u32x4_t*)k_lds)[i] = (u32x4_t){0u, 0u, 0u, 0u};
  for (int i = tid; i < (int)sizeof(v_lds) / 16; i += 256) ((u32x4_t*)v_lds)[i] = (u32x4_t){0u, 0u, 0u, 0u};

  // Q fragments ("B" operand): lane holds Q[qrow][8*(lg+4ks) .. +7]
  bf16x8_t qf[KS];
  {
    const uint16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int ch = lg + 4 * ks;
      u32x4_t t = {0u, 0u, 0u, 0u};
      if (qrow < p.Sq && ch < KCH) t = *(const u32x4_t*)(qp + ch * 8);
      qf[ks] = __builtin_bit_cast(bf16x8_t, t);
    }
  }

  f32x4_t o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i) o[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
  float m_run = -INFINITY, l_part = 0.f;

  int kv_end = p.Skv;
  if (CAUSAL) kv_end = min(p.Skv, q0 + 64 + off);
  const int nchunk = (kv_end + KV_CHUNK - 1) / KV_CHUNK;
  const uint16_t* kbase = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const uint16_t* vbase = p.v + (long)b * p.v_bs + (long)h * p.v_hs;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;

  __syncthreads();
  for (int c = 0; c < nchunk; ++c) {
    const int key0 = c * KV_CHUNK;
    // ---- stage K chunk (row-major) and V chunk (transposed) ----
    for (int piece = tid; piece < KV_CHUNK * KCH; piece += 256) {
      const int key = piece / KCH, ch = piece - key * KCH;
      u32x4_t t = {0u, 0u, 0u, 0u};
      if (key0 + key < p.Skv) t = *(const u32x4_t*)(kbase + (long)(key0 + key) * p.k_rs + ch * 8);
      *(u32x4_t*)(k_lds + key * KROW + ch * 16) = t;
    }
    for (int piece = tid; piece < KV_CHUNK * KCH; piece += 256) {
      const int key = piece / KCH, ch = piece - key * KCH;
      u32x4_t t = {0u, 0u, 0u, 0u};
      if (key0 + key < p.Skv) t = *(const u32x4_t*)(vbase + (long)(key0 + key) * p.v_rs + ch * 8);
      *(u32x4_t*)(v_lds + key * KROW + ch * 16) = t;
    }
    __syncthreads();

    // ---- S^T = K · Q^T : lane owns query column l15 and keys 16*kt + 4*lg + r ----
    float s[4][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8_t kf = *(const bf16x8_t*)(k_lds + (kt * 16 + l15) * KROW + (lg + 4 * ks) * 16);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + kt * 16 + lg * 4 + r;
        bool vis = key < p.Skv;
        if (CAUSAL) vis = vis && (key <= qrow + off);
        if (mrow) vis = vis && (key < p.Skv ? mrow[key] != 0 : false);
        s[kt][r] = vis ? acc[r] * p.scale_log2e : -INFINITY;
      }
    }
    // ---- online softmax (base-2 domain) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);   // m_run = -inf → 0
    m_run = m_new;
    float psum = 0.f;
    bf16x8_t pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        e[j] = __builtin_amdgcn_exp2f(s[2 * s2 + (j >> 2)][j & 3] - m_use);
        psum += e[j];
      }
      u32x4_t t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = pack2bf(e[2 * j], e[2 * j + 1]);
      pf[s2] = __builtin_bit_cast(bf16x8_t, t);
    }
    l_part = l_part * alpha + psum;
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] *= alpha;

    // ---- O^T += V^T · P^T : element j of lane group lg ↔ key 32*s2 + 16*(j>>2) + 4*lg + (j&3) on both operands ----
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        // hardware-transposing read: within each 16-lane group, lane 4q+p supplies the address of key row q, d
        // columns 4p..4p+3 of a 4-key x 16-d block and receives column (lane & 15) of the 4 rows → the V^T fragment.
        const char* vp = v_lds + (32 * s2 + 4 * lg + (l15 >> 2)) * KROW + (dt * 16 + (l15 & 3) * 4) * 2;
        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)vp);
        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 16 * KROW));
        const u32x2_t w0 = __builtin_bit_cast(u32x2_t, v0), w1 = __builtin_bit_cast(u32x2_t, v1);
        const u32x4_t vv = {w0[0], w0[1], w1[0], w1[1]};
        o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf[s2], o[dt], 0, 0, 0);
      }
    }
    __syncthreads();   // everyone done with this chunk before it is overwritten
  }

  // ---- finalize: lane holds O^T[d = 16*dt + 4*lg + r][q = l15] ----
  float l = l_part;
  l += __shfl_xor(l, 16, 64);
  l += __shfl_xor(l, 32, 64);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  if (p.lse && lg == 0 && qrow < p.Sq)
    p.lse[((long)b * p.H + h) * p.lse_rs + qrow] = l > 0.f ? m_run + __builtin_amdgcn_logf(l) : INFINITY;
  if (qrow < p.Sq) {
    uint16_t* op = p.o + (long)b * p.o_bs + (long)h * p.o_hs + (long)qrow * p.o_rs;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int d = dt * 16 + lg * 4;
      if (d < HD) {
        u32x2_t w;
        w[0] = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
        w[1] = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
        *(u32x2_t*)(op + d) = w;
      }
    }
  }
}

// ---- whole-sequence kernel: one workgroup (8 waves) per (batch, head), Skv ≤ 320 ----
// The OpenVLA sequences are short (256 patches + prompt ≈ 288; ViT 256/261), so K and V of a head fit in LDS at once
// (≤ 160 KiB): they are staged ONCE (XOR-swizzled 16-byte chunks, no padding), then every wave walks its 16-row query
// tiles with no further barriers: full score row in registers (≤ 20 key tiles × 4 fp32), exact two-pass softmax (no
// online rescaling — the oracle's algorithm), P → bf16 → PV with transposing LDS reads. Query tiles are dealt to the 8
// waves in snake order from the heaviest (causal) tile down, so wave loads differ by < 10 %.
template <int HD, bool CAUSAL, bool ROPE = false>
__global__ __launch_bounds__(512) void attn_seq_kernel(AttnArgs p, int s_pad) {
  static_assert(!ROPE || HD == 128, "fused RoPE is built for head_dim 128");
  constexpr int HDP = (HD + 31) / 32 * 32;
  constexpr int KS = HDP / 32;
  constexpr int KCH = HD / 8;                       // 16-byte chunks per row in HBM
  constexpr int ROWB = (HD <= 64) ? 128 : 256;      // LDS bytes per row (power of two so the XOR swizzle stays in-row)
  constexpr int MASK = ROWB / 16 - 1;
  // V is read with ds_read_b64_tr_b16: a 32-lane half touches 8 keys x 32 B, so V swizzles 32-byte chunk PAIRS
  // (pair' = pair ^ ((key >> VSH) & VPM)) — with K's per-chunk swizzle adjacent keys would share a slot pair (2-way).
  constexpr int VSH = (ROWB == 256) ? 0 : 1, VPM = (ROWB == 256) ? 7 : 3;
  constexpr int DT = (HD + 15) / 16;
  constexpr int NKT = 20;                           // key tiles held in registers → Skv ≤ 320
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* k_lds = smem;
  char* v_lds = smem + s_pad * ROWB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x / p.H, h = blockIdx.x - b * p.H;
  const int off = p.Skv - p.Sq;
  const uint16_t* kbase = p.k + (long)b * p.k_bs + (long)h * p.k_hs;
  const uint16_t* vbase = p.v + (long)b * p.v_bs + (long)h * p.v_hs;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;

  // ---- stage K and V once: chunk c of key row r lives at r*ROWB + ((c ^ (r & MASK)) << 4). One pass over all
  //      s_pad × CH LDS chunks (pad rows / pad chunks get zeros); loads are issued 6 deep per tensor before the
  //      LDS writes so a thread pays the global-load latency once per batch, not once per chunk. ----
  if constexpr (ROPE) {
    // one piece = the chunk pair (ch, ch + 8) of one key row: rotate k, copy v, fill LDS and (first workgroup of the head)
    // the KV cache rows pos0 + key — what bl_rope_kvcache_bf16 did in a pass of its own over the fused qkv buffer
    constexpr int UNR = 3;
    const int npieces = s_pad * 8;
    const long cache_base = ((long)b * p.H + h) * p.cache_len;
    for (int base = tid; base < npieces; base += 512 * UNR) {
      u32x4_t k1[UNR], k2[UNR], v1[UNR], v2[UNR], cq[UNR], sq[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int piece = base + u * 512, key = piece >> 3, ch = piece & 7;
        k1[u] = k2[u] = v1[u] = v2[u] = cq[u] = sq[u] = (u32x4_t){0u, 0u, 0u, 0u};
        if (piece < npieces && key < p.Skv) {
          const uint16_t* kr = kbase + (long)key * p.k_rs + ch * 8;
          const uint16_t* vr = vbase + (long)key * p.v_rs + ch * 8;
          k1[u] = *(const u32x4_t*)kr; k2[u] = *(const u32x4_t*)(kr + 64);
          v1[u] = *(const u32x4_t*)vr; v2[u] = *(const u32x4_t*)(vr + 64);
          cq[u] = *(const u32x4_t*)(p.cos_tab + (long)(p.pos0 + key) * 64 + ch * 8);
          sq[u] = *(const u32x4_t*)(p.sin_tab + (long)(p.pos0 + key) * 64 + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int piece = base + u * 512, key = piece >> 3, ch = piece & 7;
        if (piece < npieces) {
          u32x4_t o1, o2;
          rope_pair(k1[u], k2[u], cq[u], sq[u], o1, o2);
          *(u32x4_t*)(k_lds + key * ROWB + ((ch ^ (key & MASK)) << 4)) = o1;
          *(u32x4_t*)(k_lds + key * ROWB + (((ch + 8) ^ (key & MASK)) << 4)) = o2;
          const int pr1 = ((ch >> 1) ^ ((key >> VSH) & VPM)), pr2 = (((ch + 8) >> 1) ^ ((key >> VSH) & VPM));
          *(u32x4_t*)(v_lds + key * ROWB + (((pr1 << 1) | (ch & 1)) << 4)) = v1[u];
          *(u32x4_t*)(v_lds + key * ROWB + (((pr2 << 1) | (ch & 1)) << 4)) = v2[u];
          if (blockIdx.y == 0 && key < p.Skv) {
            uint16_t* kc = p.k_cache + (cache_base + p.pos0 + key) * 128 + ch * 8;
            uint16_t* vc = p.v_cache + (cache_base + p.pos0 + key) * 128 + ch * 8;
            *(u32x4_t*)kc = o1; *(u32x4_t*)(kc + 64) = o2;
            *(u32x4_t*)vc = v1[u]; *(u32x4_t*)(vc + 64) = v2[u];
          }
        }
      }
    }
  } else {
    constexpr int CH = ROWB / 16, UNR = 6;
    const int npieces = s_pad * CH;
    for (int base = tid; base < npieces; base += 512 * UNR) {
      u32x4_t kq[UNR], vq[UNR];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int piece = base + u * 512, key = piece / CH, ch = piece - key * CH;
        kq[u] = (u32x4_t){0u, 0u, 0u, 0u};
        vq[u] = (u32x4_t){0u, 0u, 0u, 0u};
        if (piece < npieces && key < p.Skv && ch < KCH) {
          kq[u] = *(const u32x4_t*)(kbase + (long)key * p.k_rs + ch * 8);
          vq[u] = *(const u32x4_t*)(vbase + (long)key * p.v_rs + ch * 8);
        }
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int piece = base + u * 512, key = piece / CH, ch = piece - key * CH;
        if (piece < npieces) {
          *(u32x4_t*)(k_lds + key * ROWB + ((ch ^ (key & MASK)) << 4)) = kq[u];
          *(u32x4_t*)(v_lds + key * ROWB + (((((ch >> 1) ^ ((key >> VSH) & VPM)) << 1) | (ch & 1)) << 4)) = vq[u];
        }
      }
    }
  }
  __syncthreads();
  // key visibility (key < Skv and not hidden by the key-padding mask) of this lane's keys kt·16 + lg·4 + {0..3} of every
  // key tile, one bit each: read from global memory ONCE per workgroup instead of inside every query tile's score loop
  uint32_t mbits[(NKT * 4 + 31) / 32] = {};            // bit kt·4 + rr
  {
    const bool words = mrow != nullptr && ((((uintptr_t)mrow) & 3) == 0);
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      const int k0 = kt * 16 + lg * 4;
      uint32_t w = 0;                                   // 4 bits: visibility of keys k0 .. k0 + 3
      if (mrow == nullptr) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) w |= (k0 + rr < p.Skv ? 1u : 0u) << rr;
      } else if (words && k0 + 3 < p.Skv) {
        const uint32_t raw = *(const uint32_t*)(mrow + k0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) w |= (((raw >> (8 * rr)) & 0xffu) != 0 ? 1u : 0u) << rr;
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
          if (k0 + rr < p.Skv && mrow[k0 + rr] != 0) w |= 1u << rr;
      }
      mbits[(kt * 4) >> 5] |= w << ((kt * 4) & 31);
    }
  }

  // query tiles, heaviest (causal) first: with few (batch, head) pairs (batch-1 serving: 32 workgroups on 256 CUs) gridDim.y
  // workgroups share a head — workgroup y takes tiles y, y + gridDim.y, … and stages K / V itself (L2-resident)
  const int nqt = (p.Sq + 15) >> 4;
  const int nsp = gridDim.y, sp = blockIdx.y;
  const int nloc = (nqt - sp + nsp - 1) / nsp;
  for (int r = 0; r * 8 < nloc; ++r) {
    const int local = r * 8 + ((r & 1) ? 7 - wave : wave);
    if (local >= nloc) continue;                    // wave-uniform
    const int idx = local * nsp + sp;
    const int qt = nqt - 1 - idx, q0 = qt * 16, qrow = q0 + l15;
    int kv_hi = p.Skv;
    if (CAUSAL) kv_hi = min(p.Skv, q0 + 16 + off);

    bf16x8_t qf[KS];
    {
      const uint16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qrow * p.q_rs;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int ch = lg + 4 * ks;
        u32x4_t t = {0u, 0u, 0u, 0u};
        if (qrow < p.Sq && ch < KCH) t = *(const u32x4_t*)(qp + ch * 8);
        qf[ks] = __builtin_bit_cast(bf16x8_t, t);
      }
      if constexpr (ROPE) {   // the lane holds chunks lg, lg+4 | lg+8, lg+12: both halves of two rotation pairs
        const int pos = p.pos0 + min(qrow, p.Sq - 1);
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const int c = lg + 4 * pr;
          const u32x4_t cq = *(const u32x4_t*)(p.cos_tab + (long)pos * 64 + c * 8);
          const u32x4_t sq = *(const u32x4_t*)(p.sin_tab + (long)pos * 64 + c * 8);
          u32x4_t o1, o2;
          rope_pair(__builtin_bit_cast(u32x4_t, qf[pr]), __builtin_bit_cast(u32x4_t, qf[pr + 2]), cq, sq, o1, o2);
          qf[pr] = __builtin_bit_cast(bf16x8_t, o1);
          qf[pr + 2] = __builtin_bit_cast(bf16x8_t, o2);
        }
      }
    }
    // ---- S^T = K · Q^T for every needed key tile, two key tiles (32 keys) per step. The K fragments of step g + 1 are
    //      read from LDS BEFORE the MFMAs of step g are issued (two register buffers), so a fragment's LDS latency passes
    //      under the previous step's matrix work instead of in front of every MFMA (the single-buffer form the compiler
    //      chose at 249 VGPRs waited lgkmcnt(0) before each of the 144 MFMAs of a query tile: 6 x the MFMA time). ----
    float sc[NKT][4];
    float mx = -INFINITY;
    bf16x8_t kb[2][2 * KS];
    auto load_k = [&](int g, bf16x8_t* dst) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          dst[t * KS + ks] = *(const bf16x8_t*)(k_lds + ((2 * g + t) * 16 + l15) * ROWB + (((lg + 4 * ks) ^ (l15 & MASK)) << 4));
    };
    load_k(0, kb[0]);
#pragma unroll
    for (int g = 0; g < NKT / 2; ++g) {
      if (g * 32 < kv_hi) {
        if (g + 1 < NKT / 2 && (g + 1) * 32 < kv_hi) load_k(g + 1, kb[(g + 1) & 1]);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int kt = 2 * g + t;
          f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb[g & 1][t * KS + ks], qf[ks], acc, 0, 0, 0);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int key = kt * 16 + lg * 4 + rr;
            bool vis = ((mbits[(kt * 4 + rr) >> 5] >> ((kt * 4 + rr) & 31)) & 1u) != 0;   // in range and not masked
            if (CAUSAL) vis = vis && (key <= qrow + off);
            sc[kt][rr] = vis ? acc[rr] * p.scale_log2e : -INFINITY;
            mx = fmaxf(mx, sc[kt][rr]);
          }
        }
      } else {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sc[2 * g + t][rr] = -INFINITY;
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_use = (mx == -INFINITY) ? 0.f : mx;
    // ---- P = exp2(S - m); O^T += V^T · P^T, 32 keys (two key tiles) per MFMA k-step; V fragments double-buffered like K ----
    f32x4_t o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) o[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float l = 0.f;
    s16x4_t vb[2][2 * DT];
    auto load_v = [&](int s2, s16x4_t* dst) {
      const int key = 32 * s2 + 4 * lg + (l15 >> 2);            // +16 for the second read: same (key & MASK)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int pr = dt ^ ((key >> VSH) & VPM);                // swizzled 32-byte pair (same for key + 16)
        const char* vp = v_lds + key * ROWB + (((pr << 1) | ((l15 & 3) >> 1)) << 4) + (l15 & 1) * 8;
        dst[2 * dt] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)vp);
        dst[2 * dt + 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vp + 16 * ROWB));
      }
    };
    load_v(0, vb[0]);
#pragma unroll
    for (int s2 = 0; s2 < NKT / 2; ++s2) {
      if (s2 * 32 < kv_hi) {
        if (s2 + 1 < NKT / 2 && (s2 + 1) * 32 < kv_hi) load_v(s2 + 1, vb[(s2 + 1) & 1]);
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          e[j] = __builtin_amdgcn_exp2f(sc[2 * s2 + (j >> 2)][j & 3] - m_use);
          l += e[j];
        }
        u32x4_t t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = pack2bf(e[2 * j], e[2 * j + 1]);
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, t);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const u32x2_t w0 = __builtin_bit_cast(u32x2_t, vb[s2 & 1][2 * dt]), w1 = __builtin_bit_cast(u32x2_t, vb[s2 & 1][2 * dt + 1]);
          const u32x4_t vv = {w0[0], w0[1], w1[0], w1[1]};
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, vv), pf, o[dt], 0, 0, 0);
        }
      }
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (p.lse && lg == 0 && qrow < p.Sq)
      p.lse[((long)b * p.H + h) * p.lse_rs + qrow] = l > 0.f ? m_use + __builtin_amdgcn_logf(l) : INFINITY;
    if (qrow < p.Sq) {
      uint16_t* op = p.o + (long)b * p.o_bs + (long)h * p.o_hs + (long)qrow * p.o_rs;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int d = dt * 16 + lg * 4;
        if (d < HD) {
          u32x2_t w;
          w[0] = pack2bf(o[dt][0] * inv, o[dt][1] * inv);
          w[1] = pack2bf(o[dt][2] * inv, o[dt][3] * inv);
          *(u32x2_t*)(op + d) = w;
        }
      }
    }
  }
}

// ---- decode: one workgroup (4 waves) per (batch, head), single query row, head_dim 128 ----
// HBM-bound (reads the head's K and V rows once: 2·Skv·256 B). Lane = (key slot ks = lane>>4, 16-byte chunk dc =
// lane&15), so one wave-instruction reads 4 whole 256-byte rows; the 4 waves interleave 16-key groups, and every
// wave keeps 4 independent row loads in flight. Scores go through LDS once; the softmax statistics are recomputed by
// every wave (Skv ≤ 2048 floats), the PV partials of the 4 waves are summed through LDS.
// ROPE = true additionally rotates q and the new token's k at position `pos`, appends k', v to cache row `pos`
// and treats that row as the last key (HF apply_rotary_pos_emb + DynamicCache.update, fused).
// Several decode iterations in one launch (StaggeredDecodePipeline): group g = B consecutive rows of q / o with their own
// KV caches and position. By-value kernel argument, indexed with the (uniform) group id.
struct DecodeGroups {
  uint16_t* k[8]; uint16_t* v[8]; int pos[8];
};

template <bool ROPE, bool GROUPED = false>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs p, const uint16_t* kn, const uint16_t* vn,
                                                          const uint16_t* cos_tab, const uint16_t* sin_tab, int pos,
                                                          DecodeGroups gr) {
  constexpr int MAXKV = 2048;   // head_dim is fixed at 128 (16 lanes × 16 B per row)
  __shared__ float sc[MAXKV];
  __shared__ float part[4][128];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bh = blockIdx.x;
  if constexpr (GROUPED) {
    const int per = p.B * p.H, g = bh / per;
    bh -= g * per;
    p.k = gr.k[g]; p.v = gr.v[g];
    pos = gr.pos[g];
    p.Skv = pos + 1;
    const long roff = (long)g * p.B * p.q_bs;
    p.q += roff; kn += roff; vn += roff;
    p.o += (long)g * p.B * p.o_bs;
    if (p.mask) p.mask += (long)g * p.B * p.mask_bs;
  }
  const int b = bh / p.H, h = bh - b * p.H;
  const int ks = lane >> 4, dc = lane & 15;
  const long qoff = (long)b * p.q_bs + (long)h * p.q_hs;
  const uint8_t* mrow = p.mask ? p.mask + (long)b * p.mask_bs : nullptr;
  if (ROPE && p.rope_pos) {   // right-padded batch: this sequence's own position = its cache row, rotation angle and length
    pos = p.rope_pos[b];
    p.Skv = pos + 1;
  }
  const int n = p.Skv, n_cache = ROPE ? n - 1 : n;
  uint16_t* kc = const_cast<uint16_t*>(p.k) + (long)b * p.k_bs + (long)h * p.k_hs + dc * 8;
  uint16_t* vc = const_cast<uint16_t*>(p.v) + (long)b * p.v_bs + (long)h * p.v_hs + dc * 8;

  // ---- groups of 64 keys per workgroup iteration; wave w takes keys g*64 + w*16 + j*4 + ks, j = 0..3 ----
#define BL_LOAD_ROWS(DST, BASE, RS, G0)                                                            \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
    const int key = (G0) + wave * 16 + j * 4 + ks;                                                  \
    DST[j] = (u32x4_t){0u, 0u, 0u, 0u};                                                             \
    if (key < n_cache) DST[j] = *(const u32x4_t*)(BASE + (long)key * (RS));                         \
  }
#define BL_SCORES(KQ, G0)                                                                           \
  _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
    const int key = (G0) + wave * 16 + j * 4 + ks;                                                  \
    float d = 0.f;                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) d += qv[2 * i] * bflo(KQ[j][i]) + qv[2 * i + 1] * bfhi(KQ[j][i]); \
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64); \
    if (dc == 0 && key < n_cache) sc[key] = (mrow && mrow[key] == 0) ? -INFINITY : d * p.scale_log2e; \
  }
#define BL_PV(VQ, G0)                                                                               \
  do {                                                                                              \
    float pk[4];                                                                                    \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                 \
      const int key = (G0) + wave * 16 + j * 4 + ks;                                                \
      pk[j] = key < n_cache ? rbf(__builtin_amdgcn_exp2f(sc[key] - m_use)) : 0.f;                   \
    }                                                                                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                   \
      _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                               \
        acc[2 * i] += pk[j] * bflo(VQ[j][i]); acc[2 * i + 1] += pk[j] * bfhi(VQ[j][i]);             \
      }                                                                                             \
  } while (0)
  // Caches of up to 320 rows (every OpenVLA decode step: 288 prompt + patch positions + ≤ 7 new tokens): ALL of the wave's
  // K and V rows are requested before anything is computed — one memory round trip per launch instead of one per 64-key
  // group and pass (ten at Skv = 294: 13.7 → ≈ 6 µs at batch 1). Same arithmetic in the same order as the streaming form.
  constexpr int NG = 5;
  const bool resident = n_cache <= NG * 64;
  u32x4_t kr[NG][4], vr[NG][4];
  // V is requested only after the scores (into registers next to the ones K is leaving), not together with K: with two
  // workgroups per CU and ≈ 64 KiB of requests outstanding per CU, one workgroup's 75 KB of V — needed last — otherwise
  // queue ahead of the other's K, needed first. Merged decode iteration (3072 workgroups): 101.9 → 90.1 µs per layer
  // (4.5 → 5.1 TB/s of K/V), one-batch decode −0.7 %; same arithmetic (tools/bench_attn_decode_grouped.py).
  constexpr bool V_LATE = true;
  if (resident) {        // requested first: the q / new-token loads and the rotation below run under their latency
#pragma unroll
    for (int g = 0; g < NG; ++g) BL_LOAD_ROWS(kr[g], kc, p.k_rs, g * 64);
    if constexpr (!V_LATE) {
#pragma unroll
      for (int g = 0; g < NG; ++g) BL_LOAD_ROWS(vr[g], vc, p.v_rs, g * 64);
    }
  }

  float qv[8];
  float knv[8], vnv[8];   // ROPE: rotated new key / new value chunk (bf16 values)
  if constexpr (ROPE) {
    // half-split rotation: chunk dc pairs with chunk dc ^ 8; out = bf16(bf16(own*cos) + bf16(±partner*sin))
    const u32x4_t cq = *(const u32x4_t*)(cos_tab + (long)pos * 64 + (dc & 7) * 8);
    const u32x4_t sq = *(const u32x4_t*)(sin_tab + (long)pos * 64 + (dc & 7) * 8);
    const float sgn = dc < 8 ? -1.f : 1.f;
    const u32x4_t q_own = *(const u32x4_t*)(p.q + qoff + dc * 8), q_par = *(const u32x4_t*)(p.q + qoff + (dc ^ 8) * 8);
    const u32x4_t k_own = *(const u32x4_t*)(kn + qoff + dc * 8), k_par = *(const u32x4_t*)(kn + qoff + (dc ^ 8) * 8);
    const u32x4_t v_own = *(const u32x4_t*)(vn + qoff + dc * 8);
    u32x4_t kw;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float c0 = bflo(cq[i]), c1 = bfhi(cq[i]), s0 = bflo(sq[i]) * sgn, s1 = bfhi(sq[i]) * sgn;
      qv[2 * i] = rbf(rbf(bflo(q_own[i]) * c0) + rbf(bflo(q_par[i]) * s0));
      qv[2 * i + 1] = rbf(rbf(bfhi(q_own[i]) * c1) + rbf(bfhi(q_par[i]) * s1));
      knv[2 * i] = rbf(rbf(bflo(k_own[i]) * c0) + rbf(bflo(k_par[i]) * s0));
      knv[2 * i + 1] = rbf(rbf(bfhi(k_own[i]) * c1) + rbf(bfhi(k_par[i]) * s1));
      vnv[2 * i] = bflo(v_own[i]); vnv[2 * i + 1] = bfhi(v_own[i]);
      kw[i] = pack2bf(knv[2 * i], knv[2 * i + 1]);
    }
    if (wave == 0 && ks == 0) {   // append the new token to the cache (row `pos`)
      *(u32x4_t*)(kc + (long)pos * p.k_rs) = kw;
      *(u32x4_t*)(vc + (long)pos * p.v_rs) = v_own;
    }
  } else {
    const u32x4_t t = *(const u32x4_t*)(p.q + qoff + dc * 8);
#pragma unroll
    for (int i = 0; i < 4; ++i) { qv[2 * i] = bflo(t[i]); qv[2 * i + 1] = bfhi(t[i]); }
  }

  if (resident) {
#pragma unroll
    for (int g = 0; g < NG; ++g) BL_SCORES(kr[g], g * 64);
    if constexpr (V_LATE) {
#pragma unroll
      for (int g = 0; g < NG; ++g) BL_LOAD_ROWS(vr[g], vc, p.v_rs, g * 64);
    }
  } else {
    for (int g0 = 0; g0 < n_cache; g0 += 64) {
      u32x4_t kq[4];
      BL_LOAD_ROWS(kq, kc, p.k_rs, g0);
      BL_SCORES(kq, g0);
    }
  }
  if constexpr (ROPE) {   // score of the new key, from registers
    float d = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) d += qv[i] * knv[i];
    d += __shfl_xor(d, 1, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 8, 64);
    if (wave == 0 && lane == 0) sc[pos] = (mrow && mrow[pos] == 0) ? -INFINITY : d * p.scale_log2e;
  }
  __syncthreads();
  // ---- softmax statistics (every wave, redundantly; identical results) ----
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sc[i]);
  mx = wave_max(mx);
  const float m_use = (mx == -INFINITY) ? 0.f : mx;
  float l = 0.f;
  for (int i = lane; i < n; i += 64) l += __builtin_amdgcn_exp2f(sc[i] - m_use);
  l = wave_sum(l);

  // ---- PV: same key assignment; P rounded to bf16 as in the prefill kernel ----
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (resident) {
#pragma unroll
    for (int g = 0; g < NG; ++g) BL_PV(vr[g], g * 64);
  } else {
    for (int g0 = 0; g0 < n_cache; g0 += 64) {
      u32x4_t vq[4];
      BL_LOAD_ROWS(vq, vc, p.v_rs, g0);
      BL_PV(vq, g0);
    }
  }
#undef BL_LOAD_ROWS
#undef BL_SCORES
#undef BL_PV
  if constexpr (ROPE) {
    if (wave == 0 && ks == 0) {
      const float pn = rbf(__builtin_amdgcn_exp2f(sc[pos] - m_use));
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += pn * vnv[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] += __shfl_xor(acc[i], 16, 64); acc[i] += __shfl_xor(acc[i], 32, 64); }
  if (ks == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) part[wave][dc * 8 + i] = acc[i];
  }
  __syncthreads();
  if (threadIdx.x < 64) {       // wave 0: lane handles 2 output dims
    const int d0 = lane * 2;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    const float o0 = (part[0][d0] + part[1][d0] + part[2][d0] + part[3][d0]) * inv;
    const float o1 = (part[0][d0 + 1] + part[1][d0 + 1] + part[2][d0 + 1] + part[3][d0 + 1]) * inv;
    *(uint32_t*)(p.o + (long)b * p.o_bs + (long)h * p.o_hs + d0) = pack2bf(o0, o1);
  }
}

int fill_args(const bl_attn_desc* d, AttnArgs& a) {
  if (!d || !d->q || !d->k || !d->v || !d->o) return BL_E_ARG;
  if (d->B <= 0 || d->H <= 0 || d->Sq <= 0 || d->Skv <= 0) return BL_E_SHAPE;
  const int64_t st[] = {d->q_bs, d->q_hs, d->q_rs, d->k_bs, d->k_hs, d->k_rs, d->v_bs, d->v_hs, d->v_rs};
  for (int64_t s : st) if (s % 8) return BL_E_ALIGN;
  if ((d->o_bs % 4) || (d->o_hs % 4) || (d->o_rs % 4)) return BL_E_ALIGN;
  if (!bl_aligned16(d->q) || !bl_aligned16(d->k) || !bl_aligned16(d->v) || (((uintptr_t)d->o) & 7)) return BL_E_ALIGN;
  a.q = d->q; a.k = d->k; a.v = d->v; a.o = d->o; a.mask = d->key_mask;
  a.q_bs = d->q_bs; a.q_hs = d->q_hs; a.q_rs = d->q_rs; a.k_bs = d->k_bs; a.k_hs = d->k_hs; a.k_rs = d->k_rs;
  a.v_bs = d->v_bs; a.v_hs = d->v_hs; a.v_rs = d->v_rs; a.o_bs = d->o_bs; a.o_hs = d->o_hs; a.o_rs = d->o_rs;
  a.mask_bs = d->mask_bs;
  a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Skv = d->Skv;
  a.scale_log2e = d->scale * 1.44269504088896340736f;
  a.lse = nullptr; a.lse_rs = 0;
  a.rope_pos = nullptr;
  a.cos_tab = a.sin_tab = nullptr; a.k_cache = a.v_cache = nullptr; a.cache_len = a.pos0 = 0;
  return BL_OK;
}

}  // namespace bl_attention_impl
using namespace bl_attention_impl;

template <int HD, bool CAUSAL>
int launch_seq(const AttnArgs& a, const bl_attn_desc* d, hipStream_t s) {
  constexpr int ROWB = (HD <= 64) ? 128 : 256;
  const int s_pad = (d->Skv + 31) / 32 * 32;
  const int lds = 2 * s_pad * ROWB;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_seq_kernel<HD, CAUSAL>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return BL_E_LAUNCH;
    attr_set = true;
  }
  const int heads = d->B * d->H;
  const int nsp = heads >= 128 ? 1 : (heads >= 64 ? 2 : 4);
  hipLaunchKernelGGL((attn_seq_kernel<HD, CAUSAL>), dim3(heads, nsp), dim3(512), lds, s, a, s_pad);
  return BL_OK;
}

static int attention_launch(const bl_attn_desc* d, float* lse, void* stream) {
  AttnArgs a;
  const int rc = fill_args(d, a);
  if (rc != BL_OK) return rc;
  if (d->causal && d->Skv < d->Sq) return BL_E_SHAPE;
  a.lse = lse; a.lse_rs = (d->Sq + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  static const bool chunked_only = getenv("BL_ATTN_CHUNKED") != nullptr;   // A/B aid
  // short sequences (the whole OpenVLA path): K and V of a head fit in LDS → whole-sequence kernel
  if (d->Skv <= 320 && d->Sq <= 320 && !chunked_only) {
    int r = BL_E_SHAPE;
#define BL_SEQ_CASE(HD) case HD: r = d->causal ? launch_seq<HD, true>(a, d, s) : launch_seq<HD, false>(a, d, s); break;
    switch (d->head_dim) { BL_SEQ_CASE(64) BL_SEQ_CASE(72) BL_SEQ_CASE(128) default: return BL_E_SHAPE; }
#undef BL_SEQ_CASE
    if (r != BL_OK) return r;
    BL_CHECK_LAUNCH();
    return BL_OK;
  }
  const dim3 grid((d->Sq + 63) / 64, d->H, d->B), block(256);
#define BL_ATTN_CASE(HD)                                                                       \
  case HD:                                                                                     \
    if (d->causal) hipLaunchKernelGGL((attn_fwd_kernel<HD, true>), grid, block, 0, s, a);      \
    else hipLaunchKernelGGL((attn_fwd_kernel<HD, false>), grid, block, 0, s, a);               \
    break;
  switch (d->head_dim) {
    BL_ATTN_CASE(64) BL_ATTN_CASE(72) BL_ATTN_CASE(128)
    default: return BL_E_SHAPE;
  }
#undef BL_ATTN_CASE
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_attention_bf16(const bl_attn_desc* d, void* stream) { return attention_launch(d, nullptr, stream); }

extern "C" int bl_attention_rope_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab, int32_t pos0,
                                      bl_bf16* k_cache, bl_bf16* v_cache, int32_t cache_len, void* stream) {
  AttnArgs a;
  const int rc = fill_args(d, a);
  if (rc != BL_OK) return rc;
  if (!cos_tab || !sin_tab || !k_cache || !v_cache) return BL_E_ARG;
  if (d->head_dim != 128 || !d->causal || d->Skv != d->Sq || d->Sq > 320 || pos0 < 0 || pos0 + d->Skv > cache_len) return BL_E_SHAPE;
  if (!bl_aligned16(cos_tab) || !bl_aligned16(sin_tab) || !bl_aligned16(k_cache) || !bl_aligned16(v_cache)) return BL_E_ALIGN;
  a.cos_tab = cos_tab; a.sin_tab = sin_tab; a.k_cache = k_cache; a.v_cache = v_cache; a.cache_len = cache_len; a.pos0 = pos0;
  const int s_pad = (d->Skv + 31) / 32 * 32, lds = 2 * s_pad * 256;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_seq_kernel<128, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return BL_E_LAUNCH;
    attr_set = true;
  }
  const int heads = d->B * d->H;
  const int nsp = heads >= 128 ? 1 : (heads >= 64 ? 2 : 4);
  hipLaunchKernelGGL((attn_seq_kernel<128, true, true>), dim3(heads, nsp), dim3(512), lds, (hipStream_t)stream, a, s_pad);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_attention_lse_bf16(const bl_attn_desc* d, float* lse, void* stream) {
  if (!lse) return BL_E_ARG;
  return attention_launch(d, lse, stream);
}

extern "C" int bl_attention_decode_bf16(const bl_attn_desc* d, void* stream) {
  AttnArgs a;
  const int rc = fill_args(d, a);
  if (rc != BL_OK) return rc;
  if (d->head_dim != 128 || d->Sq != 1 || d->Skv > 2048) return BL_E_SHAPE;
  if (((uintptr_t)d->o) & 15) return BL_E_ALIGN;
  hipLaunchKernelGGL((attn_decode_kernel<false>), dim3(d->B * d->H), dim3(256), 0, (hipStream_t)stream, a,
                     (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)nullptr, (const uint16_t*)nullptr, 0,
                     DecodeGroups{});
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_attention_decode_rope_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab,
                                             int32_t pos, void* stream) {
  AttnArgs a;
  const int rc = fill_args(d, a);
  if (rc != BL_OK) return rc;
  if (!cos_tab || !sin_tab) return BL_E_ARG;
  if (d->head_dim != 128 || d->Sq != 1 || d->Skv > 2048 || pos < 0 || d->Skv != pos + 1) return BL_E_SHAPE;
  if ((((uintptr_t)d->o) & 15) || !bl_aligned16(cos_tab) || !bl_aligned16(sin_tab)) return BL_E_ALIGN;
  const long D = (long)d->H * d->head_dim;   // q / k_new / v_new are the three thirds of the fused qkv row
  hipLaunchKernelGGL((attn_decode_kernel<true>), dim3(d->B * d->H), dim3(256), 0, (hipStream_t)stream, a, d->q + D,
                     d->q + 2 * D, cos_tab, sin_tab, pos, DecodeGroups{});
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_attention_decode_rope_pos_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab,
                                                 int32_t pos, const int32_t* rope_pos, void* stream) {
  AttnArgs a;
  const int rc = fill_args(d, a);
  if (rc != BL_OK) return rc;
  if (!cos_tab || !sin_tab || !rope_pos) return BL_E_ARG;
  if (d->head_dim != 128 || d->Sq != 1 || d->Skv > 2048 || pos < 0 || d->Skv != pos + 1) return BL_E_SHAPE;
  if ((((uintptr_t)d->o) & 15) || !bl_aligned16(cos_tab) || !bl_aligned16(sin_tab)) return BL_E_ALIGN;
  a.rope_pos = rope_pos;
  const long D = (long)d->H * d->head_dim;
  hipLaunchKernelGGL((attn_decode_kernel<true>), dim3(d->B * d->H), dim3(256), 0, (hipStream_t)stream, a, d->q + D,
                     d->q + 2 * D, cos_tab, sin_tab, pos, DecodeGroups{});
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_attention_decode_rope_grouped_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab,
                                                     int32_t n_groups, bl_bf16* const* k_caches, bl_bf16* const* v_caches,
                                                     const int32_t* pos, void* stream) {
  if (!d || !d->q || !d->o || !cos_tab || !sin_tab || !k_caches || !v_caches || !pos) return BL_E_ARG;
  if (n_groups < 1 || n_groups > 8) return BL_E_SHAPE;
  bl_attn_desc d0 = *d;                       // validate strides / alignment with group 0's caches in place of k / v
  d0.k = k_caches[0]; d0.v = v_caches[0]; d0.Skv = pos[0] + 1;
  AttnArgs a;
  const int rc = fill_args(&d0, a);
  if (rc != BL_OK) return rc;
  DecodeGroups gr{};
  for (int g = 0; g < n_groups; ++g) {
    if (!k_caches[g] || !v_caches[g]) return BL_E_ARG;
    if (pos[g] < 0 || pos[g] + 1 > 2048) return BL_E_SHAPE;
    if (!bl_aligned16(k_caches[g]) || !bl_aligned16(v_caches[g])) return BL_E_ALIGN;
    gr.k[g] = k_caches[g]; gr.v[g] = v_caches[g]; gr.pos[g] = pos[g];
  }
  if (d->head_dim != 128 || d->Sq != 1) return BL_E_SHAPE;
  if ((((uintptr_t)d->o) & 15) || !bl_aligned16(cos_tab) || !bl_aligned16(sin_tab)) return BL_E_ALIGN;
  const long D = (long)d->H * d->head_dim;
  hipLaunchKernelGGL((attn_decode_kernel<true, true>), dim3(n_groups * d->B * d->H), dim3(256), 0, (hipStream_t)stream, a,
                     d->q + D, d->q + 2 * D, cos_tab, sin_tab, 0, gr);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
