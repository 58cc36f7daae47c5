// train.hip — backward / optimizer kernels of the VLA training step (prismatic/training/strategies/base_strategy.py:284-366,
// fsdp.py:190-246): everything around the GEMMs. All HBM-bound: 16-byte vector accesses, one wave per row for the row-wise
// ops, fp32 math on bf16 tensors, deterministic reductions (per-block partials + a second pass; no float atomics except
// the sparse embedding scatter).
//
// Gradient conventions: activations and their gradients are bf16; parameter gradients, AdamW moments and master weights
// are fp32 (the reference trains fp32 masters under bf16 autocast with fp32 gradient reduction: train.py:156-157,
// fsdp.py:140-146). Roundings of the forward (`bf16(...)`) are treated as identity in the backward, exactly as autograd
// treats a dtype cast.
#include "bl_common.h"
#include <math.h>
#include <algorithm>
#include <stdlib.h>

namespace bl_train_impl {

__device__ __forceinline__ void unpack8(const u32x4_t q, float* v) {
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[2 * i] = bflo(q[i]); v[2 * i + 1] = bfhi(q[i]); }
}
__device__ __forceinline__ u32x4_t pack8(const float* v) {
  u32x4_t q;
#pragma unroll
  for (int i = 0; i < 4; ++i) q[i] = pack2bf(v[2 * i], v[2 * i + 1]);
  return q;
}

// ---- cross-entropy backward: dlogits = (softmax(logits) - onehot(target)) / n_valid, 0 on ignored rows ----
__global__ __launch_bounds__(256) void ce_backward_kernel(const float* logits, long ld, int n, const int64_t* targets,
                                                          long ignore_index, const float* mean_and_count,
                                                          uint16_t* dlogits, long ldd) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const long tgt = targets[row];
  uint16_t* dr = dlogits + (long)row * ldd;
  if (tgt == ignore_index) {
    for (int i = threadIdx.x * 8; i < n; i += 2048) *(u32x4_t*)(dr + i) = (u32x4_t){0u, 0u, 0u, 0u};
    return;
  }
  const float* lr = logits + (long)row * ld;
  float mx = -INFINITY;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {
    const f32x4_t q = *(const f32x4_t*)(lr + i);
    mx = fmaxf(fmaxf(mx, fmaxf(q[0], q[1])), fmaxf(q[2], q[3]));
  }
  mx = wave_max(mx);
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x * 4; i < n; i += 1024) {
    const f32x4_t q = *(const f32x4_t*)(lr + i);
    s += expf(q[0] - mx) + expf(q[1] - mx) + expf(q[2] - mx) + expf(q[3] - mx);
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / ((red[0] + red[1] + red[2] + red[3]) * mean_and_count[1]);   // 1 / (sum_exp * n_valid)
  const float inv_cnt = 1.0f / mean_and_count[1];
  for (int i = threadIdx.x * 8; i < n; i += 2048) {
    const f32x4_t a = *(const f32x4_t*)(lr + i), b = *(const f32x4_t*)(lr + i + 4);
    float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = expf(v[e] - mx) * inv - ((long)(i + e) == tgt ? inv_cnt : 0.f);
    *(u32x4_t*)(dr + i) = pack8(v);
  }
}

// ---- RMSNorm / LayerNorm backward (+ residual-stream gradient add), one wave per row ----
//   RMS: g = w ⊙ dy;  x̂ = x·rstd;        dx = rstd·(g − x̂·mean(g ⊙ x̂)) [+ dres];          dw = Σ_rows dy ⊙ bf16(x̂)
//   LN : g = w ⊙ dy;  x̂ = (x − μ)·rstd;  dx = rstd·(g − mean(g) − x̂·mean(g ⊙ x̂)) [+ dres]; dw = Σ dy ⊙ x̂;  db = Σ dy
// Per-block partials of dw (and db, stored behind the dw partials) are summed by reduce_partials_kernel.
// RPW rows per wave are in flight together (their loads are issued back to back and the wave-wide reductions of the rows
// interleave): one row per wave left a single 2·dim-byte request outstanding and ran at 0.7 TB/s (LayerNorm, dim ≈ 1 K) /
// 1.7 TB/s (RMSNorm, dim 4 K). x and dy stay PACKED in registers (4 VGPRs per 8 values) and are unpacked in each pass.
template <int NCH> constexpr int norm_bwd_rpw() { return NCH <= 3 ? 4 : NCH <= 4 ? 2 : 1; }

// DRL (one row per wave, dim > 2 K: the decoder's RMSNorm, which always adds the residual-stream gradient): the row of
// dres travels HBM → LDS by LDS-DMA (NCH wave-instructions of 1 KiB, lane-linear = the register layout) while the three
// reduction passes run, and is read with ds_read_b128 in the output pass. Loaded inside that pass, chunk by chunk behind
// the arithmetic that consumes it, it cost 42 of the kernel's 111 µs at 7B (69 µs without dres); as a fourth register set
// it would halve the occupancy (248 VGPRs already). Same arithmetic: results are bit-identical.
// NCH <= 2 (DINOv2's LayerNorm, dim 1024): three workgroups per CU — 32 x 261 rows in blocks of 16 are 522 blocks, which
// two per CU (181 VGPRs left alone) run as one round of 512 and a second one of 10.
template <int NCH, bool LN, bool DRL = false>
__global__ __launch_bounds__(256, NCH <= 2 ? 3 : 1) void norm_bwd_kernel(const uint16_t* x, long ldx, const uint16_t* w,
                                                       const uint16_t* dy, long lddy, const uint16_t* dres,
                                                       long lddres, uint16_t* dx, long lddx, float* dw_partial,
                                                       int rows, int dim, float eps, int rows_per_block) {
#if defined(__HIP_DEVICE_COMPILE__)   // __amdgpu_buffer_rsrc_t does not exist in the host pass
  constexpr int RPW = norm_bwd_rpw<NCH>();
  static_assert(!DRL || RPW == 1, "the LDS path of dres is written for one row per wave");
  __shared__ __attribute__((aligned(16))) char drl[DRL ? 4 * NCH * 1024 : 16];   // [wave][NCH KiB]
  // the packed registers are re-read through an empty asm before each pass: otherwise hipcc keeps every unpacked float of
  // every row alive across the passes (256 VGPRs + 250 AGPRs, one wave per SIMD)
#define BL_FRESH(Q) asm volatile("" : "+v"(Q))
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nchunk = dim >> 3;
  const float inv_dim = 1.0f / (float)dim;
  float dwacc[NCH][8], dbacc[LN ? NCH : 1][8];
#pragma unroll
  for (int c = 0; c < NCH; ++c)
#pragma unroll
    for (int i = 0; i < 8; ++i) { dwacc[c][i] = 0.f; if (LN) dbacc[c][i] = 0.f; }
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  // the norm weight, once per block, in LDS: read from global inside the two passes that use it (one 16-byte load per
  // chunk, each behind the arithmetic of the chunk before — the kernel has no registers to batch them) every row paid
  // 2 · NCH dependent L2 round trips
  __shared__ __attribute__((aligned(16))) u32x4_t wl[NCH * 64];
  for (int ch = threadIdx.x; ch < NCH * 64; ch += 256) wl[ch] = ch < nchunk ? *(const u32x4_t*)(w + ch * 8) : (u32x4_t){0u, 0u, 0u, 0u};
  __syncthreads();
  __amdgpu_buffer_rsrc_t rsr;
  char* mydrl = drl + (DRL ? wave * NCH * 1024 : 0);
  if constexpr (DRL) rsr = __builtin_amdgcn_make_buffer_rsrc((void*)dres, 0, (unsigned)((long)r1 * lddres * 2), 0x00020000);
  for (int rr = r0 + wave * RPW; rr < r1; rr += 4 * RPW) {
    u32x4_t qx[RPW][NCH], qd[RPW][NCH];
    if constexpr (DRL) {      // dim == NCH · 512 (launcher): every chunk is whole. The previous row's ds_reads are complete.
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsr, LDS_PTR(mydrl + c * 1024), 16, lane * 16,
                                                 (unsigned)((long)rr * lddres * 2) + c * 1024, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u)
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = c * 64 + lane, row = rr + u;
        qx[u][c] = qd[u][c] = (u32x4_t){0u, 0u, 0u, 0u};      // rows past the block / columns past dim contribute zeros
        if (ch < nchunk && row < r1) {
          qx[u][c] = *(const u32x4_t*)(x + (long)row * ldx + ch * 8);
          qd[u][c] = *(const u32x4_t*)(dy + (long)row * lddy + ch * 8);
        }
      }
    float mu[RPW], rstd[RPW], dot[RPW], gsum[RPW];
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      mu[u] = 0.f;
      if (LN) {
        float sx = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          float v[8];
          BL_FRESH(qx[u][c]);
          unpack8(qx[u][c], v);
#pragma unroll
          for (int i = 0; i < 8; ++i) sx += v[i];
        }
        mu[u] = wave_sum(sx) * inv_dim;
      }
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      float ss = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const bool in = c * 64 + lane < nchunk;
        float v[8];
        BL_FRESH(qx[u][c]);
        unpack8(qx[u][c], v);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xc = (LN && in) ? v[i] - mu[u] : (LN ? 0.f : v[i]);
          ss += xc * xc;
        }
      }
      rstd[u] = 1.0f / sqrtf(wave_sum(ss) * inv_dim + eps);
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      float d = 0.f, g1 = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = c * 64 + lane;
        float wv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, xv[8], dv[8];
        if (ch < nchunk) unpack8(wl[ch], wv);
        BL_FRESH(qx[u][c]); BL_FRESH(qd[u][c]);
        unpack8(qx[u][c], xv); unpack8(qd[u][c], dv);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float g = wv[i] * dv[i];
          const float xc = (LN && ch < nchunk) ? xv[i] - mu[u] : (LN ? 0.f : xv[i]);
          d += g * xc * rstd[u];
          if (LN) g1 += g;
        }
      }
      dot[u] = wave_sum(d) * inv_dim;
      gsum[u] = LN ? wave_sum(g1) * inv_dim : 0.f;
    }
    if constexpr (DRL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the dres row is in LDS (issued three passes ago)
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
      const int row = rr + u;
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        const int ch = c * 64 + lane;
        if (ch >= nchunk || row >= r1) continue;
        float o[8], dr[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, wv[8], xv[8], dv[8];
        unpack8(wl[ch], wv);
        BL_FRESH(qx[u][c]); BL_FRESH(qd[u][c]);
        unpack8(qx[u][c], xv); unpack8(qd[u][c], dv);
        if constexpr (DRL) unpack8(*(const u32x4_t*)(mydrl + c * 1024 + lane * 16), dr);
        else if (dres) unpack8(*(const u32x4_t*)(dres + (long)row * lddres + ch * 8), dr);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xh = (LN ? xv[i] - mu[u] : xv[i]) * rstd[u];
          o[i] = rstd[u] * (wv[i] * dv[i] - gsum[u] - xh * dot[u]) + dr[i];
          dwacc[c][i] += dv[i] * (LN ? xh : rbf(xh));
          if (LN) dbacc[c][i] += dv[i];
        }
        *(u32x4_t*)(dx + (long)row * lddx + ch * 8) = pack8(o);
      }
    }
    if constexpr (DRL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // dres row consumed before the next one lands on it
  }
  // per-block partials: the 4 waves of the block own disjoint rows → sum them through LDS, lane-major
  __shared__ float sh[4][64 * 8];
#pragma unroll
  for (int pass = 0; pass < (LN ? 2 : 1); ++pass) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int i = 0; i < 8; ++i) sh[wave][lane * 8 + i] = (LN && pass) ? dbacc[c][i] : dwacc[c][i];
      __syncthreads();
      if (wave == 0) {
        const int ch = c * 64 + lane;
        if (ch < nchunk) {
          float* dst = dw_partial + ((long)pass * gridDim.x + blockIdx.x) * dim + ch * 8;
#pragma unroll
          for (int i = 0; i < 8; ++i)
            dst[i] = sh[0][lane * 8 + i] + sh[1][lane * 8 + i] + sh[2][lane * 8 + i] + sh[3][lane * 8 + i];
        }
      }
      __syncthreads();
    }
  }
#undef BL_FRESH
#endif
}

// out[j] = Σ_b partial[b][j]   (fixed order → deterministic). Block = 64 columns × 4 row groups.
// grid.y = 1: a second set of partials (LayerNorm's db, stored behind the dw partials) goes to out2 in the same launch.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* partial, int nblocks, int dim, float* out,
                                                              float* out2 = nullptr) {
  if (blockIdx.y) { partial += (long)nblocks * dim; out = out2; }
  __shared__ float sh[4][64];
  const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + c;
  // eight independent partial sums per thread (fixed assignment → still deterministic): the single dependent chain of
  // the first version (130+ serial loads per thread on 16–64 blocks) cost more than the norm backward pass it finished
  float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (j < dim) {
    int b = rg;
    for (; b + 28 < nblocks; b += 32) {
#pragma unroll
      for (int u = 0; u < 8; ++u) s8[u] += partial[(long)(b + 4 * u) * dim + j];
    }
    for (int u = 0; b < nblocks; b += 4, ++u) s8[u] += partial[(long)b * dim + j];
  }
  const float s = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
  sh[rg][c] = s;
  __syncthreads();
  if (rg == 0 && j < dim) out[j] = (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]);
}

// column sums of a bf16 matrix [rows, cols] → per-block partials (bias gradients). Block = 64 column chunks (512
// columns, one contiguous KiB per row for a wave) × 4 row groups; every thread keeps four independent row loads in flight
// (the first version walked its rows one dependent load at a time: 0.8 TB/s); the row groups are summed through LDS.
__global__ __launch_bounds__(256) void colsum_partial_kernel(const uint16_t* a, long lda, int rows, int cols,
                                                             int rows_per_block, float* partial) {
  __shared__ float sh[4][64 * 8];
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col8 = (blockIdx.y * 64 + lane) * 8;
  const bool live = col8 < cols;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int r = r0 + rg; r < r1; r += 16) {
    u32x4_t q[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      q[u] = (u32x4_t){0u, 0u, 0u, 0u};
      if (live && r + 4 * u < r1) q[u] = *(const u32x4_t*)(a + (long)(r + 4 * u) * lda + col8);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float v[8];
      unpack8(q[u], v);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) sh[rg][lane * 8 + i] = acc[i];
  __syncthreads();
  if (rg == 0 && live) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
      partial[(long)blockIdx.x * cols + col8 + i] = (sh[0][lane * 8 + i] + sh[1][lane * 8 + i]) + (sh[2][lane * 8 + i] + sh[3][lane * 8 + i]);
  }
}

// ---- SwiGLU on an interleaved gate/up buffer gu [M, 2I] (col 2j = gate_j, 2j+1 = up_j) ----
__global__ void swiglu_fwd_kernel(const uint16_t* gu, long ldg, uint16_t* act, long lda, long rows, int inter) {
  const long total = rows * (inter >> 2);
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / (inter >> 2);
    const int j = (int)(t - r * (inter >> 2)) * 4;
    float v[8];
    unpack8(*(const u32x4_t*)(gu + r * ldg + 2 * j), v);
    u32x2_t o;
    o[0] = pack2bf(rbf(silu_f(v[0])) * v[1], rbf(silu_f(v[2])) * v[3]);
    o[1] = pack2bf(rbf(silu_f(v[4])) * v[5], rbf(silu_f(v[6])) * v[7]);
    *(u32x2_t*)(act + r * lda + j) = o;
  }
}
__global__ void swiglu_bwd_kernel(const uint16_t* gu, long ldg, const uint16_t* dact, long ldd, uint16_t* dgu,
                                  long ldo, long rows, int inter) {
  const long total = rows * (inter >> 2);
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / (inter >> 2);
    const int j = (int)(t - r * (inter >> 2)) * 4;
    float v[8], o[8];
    unpack8(*(const u32x4_t*)(gu + r * ldg + 2 * j), v);
    const u32x2_t dq = *(const u32x2_t*)(dact + r * ldd + j);
    const float da[4] = {bflo(dq[0]), bfhi(dq[0]), bflo(dq[1]), bfhi(dq[1])};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      swiglu_bwd_pair(v[2 * e], v[2 * e + 1], da[e], o[2 * e], o[2 * e + 1]);   // (d gate, d up = dact · bf16(silu(gate)))
    }
    *(u32x4_t*)(dgu + r * ldo + 2 * j) = pack8(o);
  }
}

// ---- exact-erf GELU forward / backward (projector; training keeps the pre-activation) ----
__global__ void gelu_fwd_kernel(const uint16_t* x, long ldx, uint16_t* y, long ldy, long rows, int cols) {
  const int cpr = cols >> 3;
  const long n8 = rows * cpr;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long r = t / cpr;
    const int c = (int)(t - r * cpr) * 8;
    float v[8];
    unpack8(*(const u32x4_t*)(x + r * ldx + c), v);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = gelu_erf(v[i]);
    *(u32x4_t*)(y + r * ldy + c) = pack8(v);
  }
}
__global__ void gelu_bwd_kernel(const uint16_t* x, long ldx, const uint16_t* dy, long lddy, uint16_t* dx, long lddx, long rows,
                                int cols) {
  const int cpr = cols >> 3;
  const long n8 = rows * cpr;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n8; t += (long)gridDim.x * blockDim.x) {
    const long r = t / cpr;
    const int c = (int)(t - r * cpr) * 8;
    float v[8], d[8];
    unpack8(*(const u32x4_t*)(x + r * ldx + c), v);
    unpack8(*(const u32x4_t*)(dy + r * lddy + c), d);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      d[i] *= gelu_erf_grad(v[i]);
    *(u32x4_t*)(dx + r * lddx + c) = pack8(d);
  }
}

// ---- RoPE backward on the q and k thirds of a fused dqkv buffer [B*S, 3*H*hd] (transpose of the rotation) ----
__global__ void rope_bwd_kernel(uint16_t* dqkv, long ld, int B, int S, int H, int hd, const uint16_t* cos_tab,
                                const uint16_t* sin_tab, int pos0) {
  const int half = hd >> 1, cpr = half >> 3;
  const long total = (long)B * S * H * cpr * 2;
  const long D = (long)H * hd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int part = (int)(i & 1);
    const long ii = i >> 1;
    const int ch = (int)(ii % cpr), h = (int)((ii / cpr) % H);
    const long tok = ii / ((long)cpr * H);
    const int pos = pos0 + (int)(tok % S);
    float c[8], s[8], d1[8], d2[8], o1[8], o2[8];
    unpack8(*(const u32x4_t*)(cos_tab + (long)pos * half + ch * 8), c);
    unpack8(*(const u32x4_t*)(sin_tab + (long)pos * half + ch * 8), s);
    uint16_t* p = dqkv + tok * ld + part * D + (long)h * hd + ch * 8;
    unpack8(*(const u32x4_t*)p, d1);
    unpack8(*(const u32x4_t*)(p + half), d2);
#pragma unroll
    for (int e = 0; e < 8; ++e) { o1[e] = d1[e] * c[e] + d2[e] * s[e]; o2[e] = d2[e] * c[e] - d1[e] * s[e]; }
    *(u32x4_t*)p = pack8(o1);
    *(u32x4_t*)(p + half) = pack8(o2);
  }
}

// ---- LayerScale residual (timm LayerScale patched to `scale_factor`, modeling_prismatic.py:52-59) ----
// forward:  y = bf16(bf16(u ⊙ ls) + res)           (u = the branch output, kept for the backward)
// backward: du = dy ⊙ ls;  dls partial[block][j] = Σ_rows dy ⊙ u
__global__ void scale_residual_kernel(const uint16_t* u, long ldu, const uint16_t* ls, const uint16_t* res, long ldr,
                                      uint16_t* y, long ldy, long rows, int cols) {
  const int cpr = cols >> 3;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cpr;
    const int ch = (int)(i - r * cpr);
    float a[8], s[8], b[8], o[8];
    unpack8(*(const u32x4_t*)(u + r * ldu + ch * 8), a);
    unpack8(*(const u32x4_t*)(ls + ch * 8), s);
    unpack8(*(const u32x4_t*)(res + r * ldr + ch * 8), b);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = rbf(a[e] * s[e]) + b[e];
    *(u32x4_t*)(y + r * ldy + ch * 8) = pack8(o);
  }
}
// One thread per 8 columns, blockDim.x = cols / 8 threads per row block (≤ 256), four rows in flight per thread (their loads
// are issued before any is consumed). Round 4: 64-row blocks of a 256-thread workgroup whose upper half returned at once
// (cols = 1024 → 128 live threads) were 131 two-wave workgroups on 256 CUs, one dependent row after the other: 1.3 TB/s.
__global__ __launch_bounds__(256) void layerscale_bwd_kernel(const uint16_t* dy, long lddy, const uint16_t* u, long ldu,
                                                             const uint16_t* ls, uint16_t* du, long lddu, int rows,
                                                             int cols, int rows_per_block, float* partial) {
  const int col8 = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
  if (col8 >= cols) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s[8];
  unpack8(*(const u32x4_t*)(ls + col8), s);
  const int r0 = blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
  for (int r = r0; r < r1; r += 4) {
    u32x4_t gq[4], aq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int rr = min(r + k, r1 - 1);
      gq[k] = *(const u32x4_t*)(dy + (long)rr * lddy + col8);
      aq[k] = *(const u32x4_t*)(u + (long)rr * ldu + col8);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (r + k >= r1) break;
      float g[8], a[8], o[8];
      unpack8(gq[k], g);
      unpack8(aq[k], a);
#pragma unroll
      for (int e = 0; e < 8; ++e) { acc[e] += g[e] * a[e]; o[e] = g[e] * s[e]; }
      *(u32x4_t*)(du + (long)(r + k) * lddu + col8) = pack8(o);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) partial[(long)blockIdx.x * cols + col8 + e] = acc[e];
}

// ---- LoRA dropout (PEFT: result = base(x) + lora_B(lora_A(dropout(x))) · scaling; vla-scripts/finetune.py:101,177) ----
// Counter-based mask, recomputable in the backward pass: element (row, col) of the adapted linear `salt` is KEPT at step
// *seed iff the top 24 bits of mix32(idx ^ mix32(mix32(*seed) + salt)) reach p · 2^24 (oracle/synth.py::dropout_keep restates
// it). The seed lives on the device so that a captured plan draws a fresh mask on every replay.
__device__ __forceinline__ bool drop_keep(uint32_t key, uint32_t idx, uint32_t thr) { return (bl_mix32(idx ^ key) >> 8) >= thr; }

// out = bf16(x / (1 - p)) where kept, 0 elsewhere (nn.Dropout in training mode on a bf16 tensor)
__global__ void dropout_kernel(const uint16_t* x, long ldx, int rows, int cols, uint32_t thr, float inv_keep, const uint32_t* seed,
                               uint32_t salt, uint16_t* out, long ldo) {
  const uint32_t key = bl_mix32(bl_mix32(*seed) + salt);
  const int cpr = cols >> 3;
  const long total = (long)rows * cpr;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / cpr;
    const int c = (int)(t - r * cpr) * 8;
    float v[8];
    unpack8(*(const u32x4_t*)(x + r * ldx + c), v);
    const uint32_t idx0 = (uint32_t)(r * cols + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = drop_keep(key, idx0 + i, thr) ? v[i] * inv_keep : 0.f;
    *(u32x4_t*)(out + r * ldo + c) = pack8(v);
  }
}
// The single extended input-gradient GEMM gives dx = dy·W + u with u = dt·A (un-masked); dropout's backward wants dy·W + m̃ ⊙ u:
// dx ← bf16(dx + bf16((1 / (1 - p) - 1) · u)) where kept, bf16(dx - u) where dropped
__global__ void dropout_grad_fix_kernel(const uint16_t* u, long ldu, int rows, int cols, uint32_t thr, float inv_keep,
                                        const uint32_t* seed, uint32_t salt, uint16_t* dx, long lddx) {
  const uint32_t key = bl_mix32(bl_mix32(*seed) + salt);
  const int cpr = cols >> 3;
  const long total = (long)rows * cpr;
  const float kf = inv_keep - 1.0f;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const long r = t / cpr;
    const int c = (int)(t - r * cpr) * 8;
    float a[8], d[8];
    unpack8(*(const u32x4_t*)(u + r * ldu + c), a);
    unpack8(*(const u32x4_t*)(dx + r * lddx + c), d);
    const uint32_t idx0 = (uint32_t)(r * cols + c);
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] += drop_keep(key, idx0 + i, thr) ? rbf(a[i] * kf) : -a[i];
    *(u32x4_t*)(dx + r * lddx + c) = pack8(d);
  }
}

// ---- LoRA helpers (vla-scripts/finetune.py:174-189) ----
// out = bf16(s · x) on a small [T, R] tensor (the scaling alpha / r applied once to t and to dt)
__global__ void scale_bf16_kernel(const uint16_t* x, float s, uint16_t* out, long n8) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    unpack8(*(const u32x4_t*)(x + i * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] *= s;
    *(u32x4_t*)(out + i * 8) = pack8(v);
  }
}
// fused adapters (q‖k‖v, interleaved gate/up): zero dB[n][c] unless column block c / rp belongs to row n's member
__global__ void lora_block_mask_kernel(float* g, int n_rows, int R, int rp, int members, int interleave) {
  const long total = (long)n_rows * R;
  const int rows_per_member = n_rows / members;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / R), c = (int)(i - (long)n * R);
    const int mem = interleave ? (n % members) : (n / rows_per_member);
    if (c / rp != mem) g[i] = 0.f;
  }
}

// ---- y += a·x on flat fp32 buffers: gradient accumulation over micro-batches (finetune.py:256-262, 307-310) ----
__global__ void axpy_f32_kernel(float* y, const float* x, float a, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4_t xv = __builtin_nontemporal_load((const f32x4_t*)x + i);
    ((f32x4_t*)y)[i] += xv * a;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += a * x[i];
}

// ---- wire format of the gradient reduce-scatter (fsdp.py:139-147 reduce_dtype = bf16): fp32 ↔ bf16 casts ----
__global__ void cast_f32_bf16_kernel(const float* src, uint16_t* dst, long n) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const f32x4_t a = *(const f32x4_t*)(src + i * 8), b = *(const f32x4_t*)(src + i * 8 + 4);
    const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    *(u32x4_t*)(dst + i * 8) = pack8(v);
  }
  for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = f2bf(src[i]);
}
__global__ void cast_bf16_f32_kernel(const uint16_t* src, float* dst, long n) {
  const long n8 = n >> 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    float v[8];
    unpack8(*(const u32x4_t*)(src + i * 8), v);
    *(f32x4_t*)(dst + i * 8) = (f32x4_t){v[0], v[1], v[2], v[3]};
    *(f32x4_t*)(dst + i * 8 + 4) = (f32x4_t){v[4], v[5], v[6], v[7]};
  }
  for (long i = (n8 << 3) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = bf2f(src[i]);
}

// ---- row gather / scatter: out row r ↔ src row (r / group) * stride + offset + r % group (the projector's 256 patch rows
//      inside the [B, S, D] embedding buffer) ----
__global__ void map_rows_kernel(const uint16_t* src, long lds_, uint16_t* dst, long ldd, long rows, int cols, int group,
                                int stride, int offset, int scatter) {
  const int cpr = cols >> 3;
  const long total = rows * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cpr;
    const int ch = (int)(i - r * cpr);
    const long m = (r / group) * stride + offset + r % group;
    const long rs = scatter ? r : m, rd = scatter ? m : r;
    *(u32x4_t*)(dst + rd * ldd + ch * 8) = *(const u32x4_t*)(src + rs * lds_ + ch * 8);
  }
}

// ---- transpose bf16 [rows, cols] → [cols, rows_pad] (rows_pad ≥ rows, pad zero-filled): feeds the wgrad GEMMs ----
__global__ __launch_bounds__(256) void transpose_pad_kernel(const uint16_t* in, long ldi, int rows, int cols,
                                                            uint16_t* out, long ldo, int rows_pad) {
  __shared__ uint16_t tile[64][66];
  const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < rows && c0 + c < cols) ? in[(long)(r0 + r) * ldi + c0 + c] : (uint16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (c0 + c < cols && r0 + r < rows_pad) out[(long)(c0 + c) * ldo + r0 + r] = tile[r][c];
  }
}

// Fast path (cols % 64 == 0): 256 × 64 tile through LDS, 16-byte global loads, transposing LDS reads
// (ds_read_b64_tr_b16: a 16-lane group reads a 4-row × 16-column block and each lane receives its column's 4 values),
// 16-byte stores. LDS rows are 128 B; 32-byte chunk PAIRS are swizzled pair' = pair ^ g(row) so the 8 rows one
// half-wave touches per transposing read fall on disjoint banks. PACKED writes the fragment-major layout of
// bl_pack_weight_bf16 for the [cols, rows_pad] matrix directly (the "B" operand of the wgrad GEMM) — one pass instead of
// transpose + pack.
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int tr_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }

template <bool PACKED>
__device__ __forceinline__ void transpose_fast_body(char* tile, int bx, int by, const uint16_t* in, long ldi, int rows, int cols,
                                                    uint16_t* out, long ldo, int rows_pad, long kt_total, long kb_off, float scale) {
#if defined(__HIP_DEVICE_COMPILE__)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  // column tiles vary fastest over the grid: workgroups running together read neighbouring 128-byte segments of the
  // same rows (whole DRAM bursts / pages) instead of isolated segments 256 rows apart
  const int r0 = by * 256, c0 = bx * 64;
  u32x4_t v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int piece = tid + 256 * u, row = piece >> 3, ch = piece & 7;
    v[u] = (u32x4_t){0u, 0u, 0u, 0u};
    if (r0 + row < rows) v[u] = *(const u32x4_t*)(in + (long)(r0 + row) * ldi + c0 + ch * 8);
  }
  if (scale != 1.0f) {                       // bf16(s · x) — the arithmetic of bl_scale_bf16 (LoRA: s · B before it is packed)
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float f[8];
      unpack8(v[u], f);
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= scale;
      v[u] = pack8(f);
    }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int piece = tid + 256 * u, row = piece >> 3, ch = piece & 7;
    *(u32x4_t*)(tile + row * 128 + (((((ch >> 1) ^ tr_swz(row)) << 1) | (ch & 1)) << 4)) = v[u];
  }
  __syncthreads();
  const int col = c0 + 16 * wave + l15;
#pragma unroll
  for (int tb = 0; tb < 8; ++tb) {
    if (r0 + tb * 32 >= rows_pad) break;
    u32x2_t w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = tb * 32 + 8 * lg + 4 * i + (l15 >> 2);
      const char* ap = tile + row * 128 + ((((wave ^ tr_swz(row)) << 1) | ((l15 & 3) >> 1)) << 4) + (l15 & 1) * 8;
      w[i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)ap));
    }
    const u32x4_t o = {w[0][0], w[0][1], w[1][0], w[1][1]};      // 8 consecutive rows t = r0 + 32 tb + 8 lg + j of column `col`
    if (PACKED) {
      const long unit = ((long)(c0 / 16 + wave) * kt_total + kb_off + (r0 / 32 + tb)) * 64 + lane;
      *(u32x4_t*)(out + unit * 8) = o;
    } else {
      *(u32x4_t*)(out + (long)col * ldo + r0 + tb * 32 + 8 * lg) = o;
    }
  }
#endif
}

template <bool PACKED>
__global__ __launch_bounds__(256) void transpose_fast_kernel(const uint16_t* in, long ldi, int rows, int cols,
                                                             uint16_t* out, long ldo, int rows_pad, long kt_total, long kb_off) {
  __shared__ __attribute__((aligned(16))) char tile[256 * 128];
  transpose_fast_body<PACKED>(tile, blockIdx.x, blockIdx.y, in, ldi, rows, cols, out, ldo, rows_pad, kt_total, kb_off, 1.0f);
}

// ---- batched small ops (round 4): ONE launch for a table of independent copy / (scaled) pack / (scaled) transpose-pack ops.
// The LoRA step re-packed its 330 adapters with five launches each and copied 660 updated tensors back one launch at a time
// (≈ 2 300 launches of 3-5 µs behind ≈ 6 µs of launch gap each: 10 % of the step); the full fine-tune's re-pack plan is ≈ 800
// launches. Each table entry owns a contiguous range of workgroups; a workgroup finds its entry by binary search over the
// block prefix sums and runs the same device code as the stand-alone kernels (identical results).
struct BatchOp {
  int32_t kind, nblocks, rows, cols, rows_pad, bx_count;   // kind: 0 copy bytes (n = bytes), 1 pack [rows = n, cols = k], 2 transpose-pack
  int64_t ld, kt_total, kb_off, n;
  const void* src;
  void* dst;
  float scale;
  int32_t pad;
};

__global__ __launch_bounds__(256) void batched_ops_kernel(const BatchOp* ops, const int32_t* block_start, int n_ops) {
  __shared__ __attribute__((aligned(16))) char tile[256 * 128];
  int lo = 0, hi = n_ops - 1;                                  // last entry whose first block is <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (block_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const BatchOp op = ops[lo];
  const long lb = (long)blockIdx.x - block_start[lo], nb = op.nblocks;
  if (op.kind == 0) {                                          // bl_copy_bytes
    const long n16 = op.n >> 4;
    const u32x4_t* s4 = (const u32x4_t*)op.src;
    u32x4_t* d4 = (u32x4_t*)op.dst;
    const bool vec = (((uintptr_t)op.src | (uintptr_t)op.dst) & 15) == 0;
    if (vec) {
      for (long i = lb * 256 + threadIdx.x; i < n16; i += nb * 256) d4[i] = s4[i];
      for (long i = (n16 << 4) + lb * 256 + threadIdx.x; i < op.n; i += nb * 256) ((uint8_t*)op.dst)[i] = ((const uint8_t*)op.src)[i];
    } else {
      for (long i = lb * 256 + threadIdx.x; i < op.n; i += nb * 256) ((uint8_t*)op.dst)[i] = ((const uint8_t*)op.src)[i];
    }
  } else if (op.kind == 1) {                                   // bl_pack_weight(_into)_bf16, optionally bf16(s · x) first
    const uint16_t* src = (const uint16_t*)op.src;
    uint16_t* dst = (uint16_t*)op.dst;
    const long n = op.rows, k = op.cols, ks_n = k >> 5, total = (n >> 4) * ks_n * 64;
    for (long i = lb * 256 + threadIdx.x; i < total; i += nb * 256) {
      const int lane = (int)(i & 63);
      const long blk = i >> 6, ks = blk % ks_n, nt = blk / ks_n;
      const long row = nt * 16 + (lane & 15), col = ks * 32 + (lane >> 4) * 8;
      u32x4_t v = *(const u32x4_t*)(src + row * op.ld + col);
      if (op.scale != 1.0f) {
        float f[8];
        unpack8(v, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] *= op.scale;
        v = pack8(f);
      }
      *(u32x4_t*)(dst + ((nt * op.kt_total + op.kb_off + ks) * 64 + lane) * 8) = v;
    }
  } else {                                                     // bl_transpose_pack(_into)_bf16
    transpose_fast_body<true>(tile, (int)(lb % op.bx_count), (int)(lb / op.bx_count), (const uint16_t*)op.src, op.ld, op.rows, op.cols,
                              (uint16_t*)op.dst, 0L, op.rows_pad, op.kt_total, op.kb_off, op.scale);
  }
}

// ---- small-output TN GEMM:  C = Pᵀ·Q  with P [T, R] (R ∈ {64, 128, 192}: a LoRA rank block) and Q [T, N] (N % 64 == 0),
//      reduction over the rows — the adapter gradients dA = dtᵀ·x and dB = (tsᵀ·dy)ᵀ read dy / x exactly once, untransposed.
// HBM-bound on Q. Workgroup = 4 waves = one 64-column slab of Q (× an optional slice of T: partials to a workspace,
// summed by reduce_partials_kernel — deterministic, no atomics). Per 32-row step both slabs are staged row-major in LDS
// (double buffered) and turned into MFMA operands by transposing reads; wave w owns 16 of the 64 columns and all R rows.
// TRANS = false: C[R][N] (lane owns 4 consecutive n);  TRANS = true: C[N][R] (lane owns 4 consecutive r).
__device__ __forceinline__ int tr_swz8(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }

template <int R, bool TRANS>
__global__ __launch_bounds__(256) void gemm_tn_small_kernel(const uint16_t* P, long ldp, const uint16_t* Q, long ldq, int T,
                                                            int N, float* C, long ldc, int t_per_split, long split_stride, float alpha) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int RB = R / 16, PROW = R * 2, PCH = R / 8;          // P tile: 32 rows × PROW bytes, PCH 16-byte chunks per row
  constexpr int PLD = (32 * PCH + 255) / 256;                    // 16-byte P loads per thread per step
  __shared__ __attribute__((aligned(16))) char q_lds[2][32 * 128];
  __shared__ __attribute__((aligned(16))) char p_lds[2][32 * PROW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * 64;
  const int t_begin = blockIdx.y * t_per_split, t_end = min(T, t_begin + t_per_split);
  float* Cs = C + (long)blockIdx.y * split_stride;
  f32x4_t acc[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

  auto pswz = [](int row) { return (PROW == 256) ? tr_swz8(row) : tr_swz(row); };
  // Register ring of D steps: the rows of step s + D are requested while step s is multiplied, so D − 1 steps (8 KiB each at
  // R = 64) are in flight per workgroup — with one step ahead the kernel moved 0.8 TB/s (one load latency per 32 rows).
  constexpr int D = 6;
  u32x4_t qreg[D], preg[D][PLD];
  auto load = [&](int slot, int t0) {
    const int row = tid >> 3, ch = tid & 7;
    qreg[slot] = (u32x4_t){0u, 0u, 0u, 0u};
    if (t0 + row < t_end) qreg[slot] = *(const u32x4_t*)(Q + (long)(t0 + row) * ldq + n0 + ch * 8);
#pragma unroll
    for (int u = 0; u < PLD; ++u) {
      const int piece = tid + 256 * u, prow = piece / PCH, pch = piece - prow * PCH;
      preg[slot][u] = (u32x4_t){0u, 0u, 0u, 0u};
      if (piece < 32 * PCH && t0 + prow < t_end) preg[slot][u] = *(const u32x4_t*)(P + (long)(t0 + prow) * ldp + pch * 8);
    }
  };
  auto stage = [&](int slot, int buf) {
    const int row = tid >> 3, ch = tid & 7;
    *(u32x4_t*)(q_lds[buf] + row * 128 + (((((ch >> 1) ^ tr_swz(row)) << 1) | (ch & 1)) << 4)) = qreg[slot];
#pragma unroll
    for (int u = 0; u < PLD; ++u) {
      const int piece = tid + 256 * u, prow = piece / PCH, pch = piece - prow * PCH;
      if (piece < 32 * PCH)
        *(u32x4_t*)(p_lds[buf] + prow * PROW + (((((pch >> 1) ^ pswz(prow)) << 1) | (pch & 1)) << 4)) = preg[slot][u];
    }
  };
  // transposing fragment: 8 consecutive rows (8·lg + j) of column block `cb` (16 columns) of a staged tile
  auto frag = [&](const char* tile, int rowbytes, int cb, bool p_tile) -> bf16x8_t {
    u32x2_t w[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * lg + 4 * i + (l15 >> 2);
      const int sw = p_tile ? pswz(row) : tr_swz(row);
      const char* ap = tile + row * rowbytes + ((((cb ^ sw) << 1) | ((l15 & 3) >> 1)) << 4) + (l15 & 1) * 8;
      w[i] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)ap));
    }
    const u32x4_t o = {w[0][0], w[0][1], w[1][0], w[1][1]};
    return __builtin_bit_cast(bf16x8_t, o);
  };

  const int nsteps = t_begin < t_end ? (t_end - t_begin + 31) / 32 : 0;
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, t_begin + 32 * d);         // rows past t_end come back as zeros
  if (nsteps) stage(0, 0);
  __syncthreads();
  int buf = 0;
  for (int base = 0; base < nsteps; base += D) {
#pragma unroll
    for (int u = 0; u < D; ++u) {
      const int st = base + u;
      if (st < nsteps) {                                           // uniform
        load(u, t_begin + 32 * (st + D));                          // slot u (step st, staged one step ago) ← step st + D
        const bf16x8_t qf = frag(q_lds[buf], 128, wave, false);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          const bf16x8_t pf = frag(p_lds[buf], PROW, rb, true);
          acc[rb] = TRANS ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, qf, acc[rb], 0, 0, 0)
                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf, pf, acc[rb], 0, 0, 0);
        }
        if (st + 1 < nsteps) stage((u + 1) % D, buf ^ 1);
        __syncthreads();
        buf ^= 1;
      }
    }
  }
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    if (TRANS) {   // acc: rows r = 16 rb + 4 lg + reg, column n = n0 + 16 wave + l15  →  C[n][r..r+3]
      *(f32x4_t*)(Cs + (long)(n0 + 16 * wave + l15) * ldc + rb * 16 + lg * 4) = acc[rb] * alpha;
    } else {       // acc: rows n = n0 + 16 wave + 4 lg + reg, column r = 16 rb + l15   →  C[r][n..n+3]
      *(f32x4_t*)(Cs + (long)(rb * 16 + l15) * ldc + n0 + 16 * wave + lg * 4) = acc[rb] * alpha;
    }
  }
#endif
}

// ---- optimizer ----
// sum of squares of an fp32 tensor → per-block partials (grad-norm)
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long n, float* partial) {
  __shared__ float red[4];
  float s = 0.f;
  if ((((uintptr_t)g) & 15) == 0) {        // 16-byte loads, four independent accumulation chains per thread
    const long n4 = n >> 2;
    const f32x4_t* g4 = (const f32x4_t*)g;
    f32x4_t a = {0.f, 0.f, 0.f, 0.f};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
      const f32x4_t t = __builtin_nontemporal_load(g4 + i);
      a += t * t;
    }
    s = (a[0] + a[1]) + (a[2] + a[3]);
    for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// total norm + clip coefficient (torch.nn.utils.clip_grad_norm_: coef = clamp(max_norm / (norm + 1e-6), max = 1))
__global__ __launch_bounds__(256) void clip_coef_kernel(const float* partial, int n, float max_norm, float* out_norm_coef) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  const float norm = (float)sqrt(red[0]);
  out_norm_coef[0] = norm;
  out_norm_coef[1] = fminf(1.0f, max_norm / (norm + 1e-6f));
}
// torch.optim.AdamW single-tensor step, fp32 master + moments, optional bf16 copy of the new weights
__global__ void adamw_kernel(float* p, float* m, float* v, const float* g, const float* norm_coef, long n, float lr,
                             float beta1, float beta2, float eps, float wd, float bias_c1, float bias_c2_sqrt,
                             uint16_t* p_bf16) {
  const float coef = norm_coef ? norm_coef[1] : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bias_c2_sqrt + eps;
    pi -= (lr / bias_c1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

// ---- embedding backward: dW[ids[b,j]] += dx[b, row(j)] for the text positions of the multimodal splice ----
__global__ void embed_bwd_kernel(const int64_t* ids, int B, int L, const uint16_t* dx, int dim, int n_patches, float* dw) {
  const int cpr = dim >> 3;
  const long total = (long)B * L * cpr;
  const int S = L + n_patches;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const long t = i / cpr;
    const int j = (int)(t % L), b = (int)(t / L);
    const long id = ids[(long)b * L + j];
    const int row = (j == 0) ? 0 : j + n_patches;
    float v[8];
    unpack8(*(const u32x4_t*)(dx + ((long)b * S + row) * dim + ch * 8), v);
#pragma unroll
    for (int e = 0; e < 8; ++e) atomicAdd(dw + id * dim + ch * 8 + e, v[e]);
  }
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}

}  // namespace bl_train_impl
using namespace bl_train_impl;

extern "C" int bl_cross_entropy_backward_f32(const float* logits, int64_t ld, int32_t rows, int32_t n,
                                             const int64_t* targets, int64_t ignore_index, const float* mean_and_count,
                                             bl_bf16* dlogits, int64_t ldd, void* stream) {
  if (!logits || !targets || !mean_and_count || !dlogits) return BL_E_ARG;
  if (rows <= 0 || n <= 0 || (n % 8) || (ld % 4) || (ldd % 8)) return BL_E_SHAPE;
  if (!bl_aligned16(logits) || !bl_aligned16(dlogits)) return BL_E_ALIGN;
  hipLaunchKernelGGL(ce_backward_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, n, targets,
                     (long)ignore_index, mean_and_count, dlogits, (long)ldd);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

template <bool LN>
static int norm_backward(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* dy, int64_t lddy, const bl_bf16* dres,
                         int64_t lddres, bl_bf16* dx, int64_t lddx, float* dw, float* db, float* partial_ws,
                         int64_t partial_ws_floats, int32_t rows, int32_t dim, float eps, void* stream) {
  if (!x || !w || !dy || !dx || !dw || !partial_ws || (LN && !db)) return BL_E_ARG;
  if (rows <= 0 || dim <= 0 || (dim % 8) || dim > 64 * 8 * 10) return BL_E_SHAPE;
  if ((ldx % 8) || (lddy % 8) || (lddx % 8) || (dres && (lddres % 8))) return BL_E_ALIGN;
  // rows per block: as few as the partial workspace allows (>= 16: four rows per wave; fewer rows per block cost more in dw partial traffic than they gain in occupancy), so the grid covers the chip
  int rpb = 16;
  while ((int64_t)((rows + rpb - 1) / rpb) * dim * (LN ? 2 : 1) > partial_ws_floats) {
    rpb *= 2;
    if (rpb > 4096) return BL_E_SHAPE;
  }
  const int nch = (dim / 8 + 63) / 64;
  // one row per wave (nch >= 5): two workgroups per CU = 512 resident blocks. 9472 rows in blocks of 16 are 592 blocks = one
  // full round and a second one of 80; blocks of 20 rows (five per wave) are 474 = one round (92 -> 80 us at 7B).
  if (nch >= 5 && (rows + rpb - 1) / rpb > 512) rpb = std::max(rpb, ((rows + 511) / 512 + 3) / 4 * 4);
  const int nblk = (rows + rpb - 1) / rpb;
  hipStream_t s = (hipStream_t)stream;
#define BL_CASE(N) case N: hipLaunchKernelGGL((norm_bwd_kernel<N, LN>), dim3(nblk), dim3(256), 0, s, x, (long)ldx, w, dy, \
    (long)lddy, dres, (long)lddres, dx, (long)lddx, partial_ws, rows, dim, eps, rpb); break;
#define BL_CASE_DRL(N) case N: hipLaunchKernelGGL((norm_bwd_kernel<N, LN, true>), dim3(nblk), dim3(256), 0, s, x, (long)ldx, w, dy, \
    (long)lddy, dres, (long)lddres, dx, (long)lddx, partial_ws, rows, dim, eps, rpb); break;
  // dres through LDS: one row per wave (nch >= 5), whole 512-column chunks, 32-bit buffer offsets. BL_NORM_BWD_DRL=0: A/B.
  const char* e_drl = getenv("BL_NORM_BWD_DRL");      // read per call: the tests switch it inside one process
  const int drl_on = e_drl ? atoi(e_drl) : 1;
  const bool drl = drl_on && !LN && dres && nch >= 5 && dim == nch * 512 && (long)rows * lddres * 2 < (1L << 32) && bl_aligned16(dres);
  if (drl) {
    switch (nch) { BL_CASE_DRL(5) BL_CASE_DRL(6) BL_CASE_DRL(7) BL_CASE_DRL(8) BL_CASE_DRL(9) BL_CASE_DRL(10) default: return BL_E_SHAPE; }
  } else {
    switch (nch) { BL_CASE(1) BL_CASE(2) BL_CASE(3) BL_CASE(4) BL_CASE(5) BL_CASE(6) BL_CASE(7) BL_CASE(8) BL_CASE(9) BL_CASE(10)
      default: return BL_E_SHAPE; }
  }
#undef BL_CASE
#undef BL_CASE_DRL
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((dim + 63) / 64, LN ? 2 : 1), dim3(256), 0, s, partial_ws, nblk, dim, dw, db);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_rmsnorm_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* dy, int64_t lddy,
                                        const bl_bf16* dres, int64_t lddres, bl_bf16* dx, int64_t lddx, float* dw,
                                        float* partial_ws, int64_t partial_ws_floats, int32_t rows, int32_t dim,
                                        float eps, void* stream) {
  return norm_backward<false>(x, ldx, w, dy, lddy, dres, lddres, dx, lddx, dw, nullptr, partial_ws, partial_ws_floats, rows,
                              dim, eps, stream);
}

extern "C" int bl_layernorm_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* dy, int64_t lddy,
                                          const bl_bf16* dres, int64_t lddres, bl_bf16* dx, int64_t lddx, float* dw,
                                          float* db, float* partial_ws, int64_t partial_ws_floats, int32_t rows,
                                          int32_t dim, float eps, void* stream) {
  return norm_backward<true>(x, ldx, w, dy, lddy, dres, lddres, dx, lddx, dw, db, partial_ws, partial_ws_floats, rows, dim,
                             eps, stream);
}

extern "C" int bl_scale_residual_bf16(const bl_bf16* u, int64_t ldu, const bl_bf16* scale, const bl_bf16* res, int64_t ldres,
                                      bl_bf16* y, int64_t ldy, int64_t rows, int32_t cols, void* stream) {
  if (!u || !scale || !res || !y) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (ldu % 8) || (ldres % 8) || (ldy % 8)) return BL_E_SHAPE;
  hipLaunchKernelGGL(scale_residual_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, u,
                     (long)ldu, scale, res, (long)ldres, y, (long)ldy, (long)rows, cols);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_layerscale_backward_bf16(const bl_bf16* dy, int64_t lddy, const bl_bf16* u, int64_t ldu, const bl_bf16* scale,
                                           bl_bf16* du, int64_t lddu, float* dscale, float* partial_ws,
                                           int64_t partial_ws_floats, int32_t rows, int32_t cols, void* stream) {
  if (!dy || !u || !scale || !du || !dscale || !partial_ws) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (lddy % 8) || (ldu % 8) || (lddu % 8)) return BL_E_SHAPE;
  int rpb = 16;                                  // as many row blocks as the partial workspace holds (16 rows: 522 blocks at 8352)
  while ((int64_t)((rows + rpb - 1) / rpb) * cols > partial_ws_floats) {
    rpb *= 2;
    if (rpb > 4096) return BL_E_SHAPE;
  }
  const int nblk = (rows + rpb - 1) / rpb;
  const int tpb = std::min(256, (cols / 8 + 63) / 64 * 64);       // threads per block: the row's 8-column groups, whole waves
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(layerscale_bwd_kernel, dim3(nblk, (cols / 8 + tpb - 1) / tpb), dim3(tpb), 0, s, dy, (long)lddy, u, (long)ldu,
                     scale, du, (long)lddu, rows, cols, rpb, partial_ws);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, partial_ws, nblk, cols, dscale, (float*)nullptr);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

// plain device memset / copy as replayable ops of the step plans (gradient buffers that are accumulated into, small
// gradient slots that are filled from a shared scratch). Kernels, not hipMemsetAsync / hipMemcpyAsync: inside a captured
// HIP graph the runtime's memset / copy nodes were observed to run out of order with the neighbouring kernel nodes
// (7B full-train step replayed as a graph: garbage vision-tower gradients; eager and kernel-only graphs are exact).
__global__ void zero_bytes_kernel(uint8_t* dst, long bytes) {
  const long n16 = bytes >> 4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
    ((u32x4_t*)dst)[i] = (u32x4_t){0u, 0u, 0u, 0u};
  for (long i = (n16 << 4) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (long)gridDim.x * blockDim.x) dst[i] = 0;
}
__global__ void copy_bytes_kernel(uint8_t* dst, const uint8_t* src, long bytes, int vec) {
  if (vec) {
    const long n16 = bytes >> 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
      ((u32x4_t*)dst)[i] = ((const u32x4_t*)src)[i];
    for (long i = (n16 << 4) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
  } else {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < bytes; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
  }
}
extern "C" int bl_memset_zero(void* dst, int64_t bytes, void* stream) {
  if (!dst || bytes <= 0) return BL_E_ARG;
  if (((uintptr_t)dst) & 15) return BL_E_ALIGN;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3(grid_for((bytes + 15) / 16, 256)), dim3(256), 0, (hipStream_t)stream, (uint8_t*)dst,
                     (long)bytes);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_copy_bytes(void* dst, const void* src, int64_t bytes, void* stream) {
  if (!dst || !src || bytes <= 0) return BL_E_ARG;
  const int vec = ((((uintptr_t)dst) | ((uintptr_t)src)) & 15) == 0;
  hipLaunchKernelGGL(copy_bytes_kernel, dim3(grid_for(vec ? (bytes + 15) / 16 : bytes, 256)), dim3(256), 0, (hipStream_t)stream,
                     (uint8_t*)dst, (const uint8_t*)src, (long)bytes, vec);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_scale_bf16(const bl_bf16* x, float s, bl_bf16* out, int64_t n, void* stream) {
  if (!x || !out) return BL_E_ARG;
  if (n <= 0 || (n % 8)) return BL_E_SHAPE;
  if (!bl_aligned16(x) || !bl_aligned16(out)) return BL_E_ALIGN;
  hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n / 8, 256)), dim3(256), 0, (hipStream_t)stream, x, s, out, (long)(n / 8));
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_lora_block_mask_f32(float* g, int32_t n_rows, int32_t R, int32_t rp, int32_t members, int32_t interleave,
                                      void* stream) {
  if (!g) return BL_E_ARG;
  if (n_rows <= 0 || R <= 0 || rp <= 0 || members <= 0 || R != rp * members || (n_rows % members)) return BL_E_SHAPE;
  hipLaunchKernelGGL(lora_block_mask_kernel, dim3(grid_for((long)n_rows * R, 256)), dim3(256), 0, (hipStream_t)stream, g,
                     n_rows, R, rp, members, interleave);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream) {
  if (!y || !x) return BL_E_ARG;
  if (n <= 0) return BL_E_SHAPE;
  if (!bl_aligned16(y) || !bl_aligned16(x)) return BL_E_ALIGN;
  hipLaunchKernelGGL(axpy_f32_kernel, dim3(grid_for((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, y, x, a, (long)n);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_cast_f32_bf16(const float* src, bl_bf16* dst, int64_t n, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (n <= 0) return BL_E_SHAPE;
  if (!bl_aligned16(src) || !bl_aligned16(dst)) return BL_E_ALIGN;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_cast_bf16_f32(const bl_bf16* src, float* dst, int64_t n, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (n <= 0) return BL_E_SHAPE;
  if (!bl_aligned16(src) || !bl_aligned16(dst)) return BL_E_ALIGN;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for((n + 7) / 8, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, (long)n);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_colsum_bf16(const bl_bf16* a, int64_t lda, int32_t rows, int32_t cols, float* out, float* partial_ws,
                              int64_t partial_ws_floats, void* stream) {
  if (!a || !out || !partial_ws) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (lda % 8)) return BL_E_SHAPE;
  // rows per block: as few as the workspace allows (>= 64: 16 rows per row group) so the grid covers the chip
  int rpb = 64;
  while ((int64_t)((rows + rpb - 1) / rpb) * cols > partial_ws_floats) {
    rpb *= 2;
    if (rpb > (1 << 20)) return BL_E_SHAPE;
  }
  const int nblk = (rows + rpb - 1) / rpb;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, (cols / 8 + 63) / 64), dim3(256), 0, s, a, (long)lda, rows, cols,
                     rpb, partial_ws);
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((cols + 63) / 64), dim3(256), 0, s, partial_ws, nblk, cols, out, (float*)nullptr);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_swiglu_bf16(const bl_bf16* gu, int64_t ldg, bl_bf16* act, int64_t lda, int64_t rows, int32_t inter,
                              void* stream) {
  if (!gu || !act) return BL_E_ARG;
  if (rows <= 0 || inter <= 0 || (inter % 4) || (ldg % 8) || (lda % 4)) return BL_E_SHAPE;
  hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for(rows * (inter / 4), 256)), dim3(256), 0, (hipStream_t)stream, gu,
                     (long)ldg, act, (long)lda, (long)rows, inter);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_swiglu_backward_bf16(const bl_bf16* gu, int64_t ldg, const bl_bf16* dact, int64_t ldd, bl_bf16* dgu,
                                       int64_t ldo, int64_t rows, int32_t inter, void* stream) {
  if (!gu || !dact || !dgu) return BL_E_ARG;
  if (rows <= 0 || inter <= 0 || (inter % 4) || (ldg % 8) || (ldd % 4) || (ldo % 8)) return BL_E_SHAPE;
  hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for(rows * (inter / 4), 256)), dim3(256), 0, (hipStream_t)stream, gu,
                     (long)ldg, dact, (long)ldd, dgu, (long)ldo, (long)rows, inter);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_gelu_bf16(const bl_bf16* x, int64_t ldx, bl_bf16* y, int64_t ldy, int64_t rows, int32_t cols, void* stream) {
  if (!x || !y) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (ldx % 8) || (ldy % 8) || ldx < cols || ldy < cols) return BL_E_SHAPE;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, y,
                     (long)ldy, (long)rows, cols);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_gelu_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* dy, int64_t lddy, bl_bf16* dx, int64_t lddx,
                                     int64_t rows, int32_t cols, void* stream) {
  if (!x || !dy || !dx) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (ldx % 8) || (lddy % 8) || (lddx % 8)) return BL_E_SHAPE;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, dy,
                     (long)lddy, dx, (long)lddx, (long)rows, cols);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_rope_backward_bf16(bl_bf16* dqkv, int64_t ld, int32_t B, int32_t S, int32_t H, int32_t hd, const bl_bf16* cos_tab,
                                     const bl_bf16* sin_tab, int32_t pos0, void* stream) {
  if (!dqkv || !cos_tab || !sin_tab) return BL_E_ARG;
  if (B <= 0 || S <= 0 || H <= 0 || hd <= 0 || (hd % 16) || pos0 < 0 || ld < 3L * H * hd || (ld % 8)) return BL_E_SHAPE;
  const long total = (long)B * S * H * (hd / 16) * 2;
  hipLaunchKernelGGL(rope_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, dqkv, (long)ld, B, S, H, hd,
                     cos_tab, sin_tab, pos0);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_map_rows_bf16(const bl_bf16* src, int64_t ld_src, bl_bf16* dst, int64_t ld_dst, int64_t rows, int32_t cols,
                                int32_t group, int32_t stride, int32_t offset, int32_t scatter, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || group <= 0 || stride < 0 || (ld_src % 8) || (ld_dst % 8)) return BL_E_SHAPE;
  if (!bl_aligned16(src) || !bl_aligned16(dst)) return BL_E_ALIGN;
  hipLaunchKernelGGL(map_rows_kernel, dim3(grid_for(rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld_src, dst, (long)ld_dst, (long)rows, cols, group, stride, offset, scatter);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_transpose_pad_bf16(const bl_bf16* in, int64_t ldi, int32_t rows, int32_t cols, bl_bf16* out,
                                     int64_t ldo, int32_t rows_pad, void* stream) {
  if (!in || !out) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || rows_pad < rows || ldo < rows_pad) return BL_E_SHAPE;
  if ((cols % 64) == 0 && (rows_pad % 32) == 0 && (ldi % 8) == 0 && (ldo % 8) == 0 && bl_aligned16(in) && bl_aligned16(out))
    hipLaunchKernelGGL((transpose_fast_kernel<false>), dim3(cols / 64, (rows_pad + 255) / 256), dim3(256), 0,
                       (hipStream_t)stream, in, (long)ldi, rows, cols, out, (long)ldo, rows_pad, 0L, 0L);
  else
    hipLaunchKernelGGL(transpose_pad_kernel, dim3((rows_pad + 63) / 64, (cols + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       in, (long)ldi, rows, cols, out, (long)ldo, rows_pad);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_transpose_pack_bf16(const bl_bf16* in, int64_t ldi, int32_t rows, int32_t cols, bl_bf16* out_packed,
                                      int32_t rows_pad, void* stream) {
  if (!in || !out_packed) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || rows_pad < rows || (cols % 64) || (rows_pad % 32) || (ldi % 8)) return BL_E_SHAPE;
  if (!bl_aligned16(in) || !bl_aligned16(out_packed)) return BL_E_ALIGN;
  hipLaunchKernelGGL((transpose_fast_kernel<true>), dim3(cols / 64, (rows_pad + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, in, (long)ldi, rows, cols, out_packed, 0L, rows_pad, (long)(rows_pad / 32), 0L);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_transpose_pack_into_bf16(const bl_bf16* in, int64_t ldi, int32_t rows, int32_t cols, bl_bf16* out_packed,
                                           int32_t rows_pad, int64_t kt_total, int64_t kb_offset, void* stream) {
  if (!in || !out_packed) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || rows_pad < rows || (cols % 64) || (rows_pad % 32) || (ldi % 8) || kb_offset < 0 ||
      kb_offset + rows_pad / 32 > kt_total)
    return BL_E_SHAPE;
  if (!bl_aligned16(in) || !bl_aligned16(out_packed)) return BL_E_ALIGN;
  hipLaunchKernelGGL((transpose_fast_kernel<true>), dim3(cols / 64, (rows_pad + 255) / 256), dim3(256), 0,
                     (hipStream_t)stream, in, (long)ldi, rows, cols, out_packed, 0L, rows_pad, (long)kt_total, (long)kb_offset);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_gemm_tn_small_bf16(const bl_bf16* P, int64_t ldp, const bl_bf16* Q, int64_t ldq, int32_t T, int32_t R,
                                     int32_t N, float* C, int64_t ldc, int32_t transpose_out, float alpha, float* partial_ws,
                                     int64_t partial_ws_floats, void* stream) {
  if (!P || !Q || !C) return BL_E_ARG;
  if (T <= 0 || (R != 64 && R != 128 && R != 192) || N <= 0 || (N % 64) || (ldp % 8) || (ldq % 8) || (ldc % 4)) return BL_E_SHAPE;
  if (!bl_aligned16(P) || !bl_aligned16(Q) || !bl_aligned16(C)) return BL_E_ALIGN;
  if (ldc != (transpose_out ? R : N)) return BL_E_SHAPE;          // dense output (the split partials mirror it)
  // split T until the grid has ≥ 256 workgroups, when the caller provides room for the partials
  // (≥ 1024 was measured too: no faster, and every split call pays a reduce launch)
  const int slabs = N / 64;
  int splits = 1;
  while (slabs * splits < 256 && (T + splits * 2 - 1) / (splits * 2) >= 256 && partial_ws &&
         (int64_t)splits * 2 * R * N <= partial_ws_floats)
    splits *= 2;
  const int t_per = ((T + splits - 1) / splits + 31) / 32 * 32;
  float* out = splits > 1 ? partial_ws : C;
  const long stride = (long)R * N;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(slabs, splits), block(256);
#define BL_TN(RR)                                                                                                        \
  case RR:                                                                                                               \
    if (transpose_out) hipLaunchKernelGGL((gemm_tn_small_kernel<RR, true>), grid, block, 0, s, P, (long)ldp, Q, (long)ldq, T, N, \
                                          out, (long)ldc, t_per, stride, alpha);                                         \
    else hipLaunchKernelGGL((gemm_tn_small_kernel<RR, false>), grid, block, 0, s, P, (long)ldp, Q, (long)ldq, T, N, out, \
                            (long)ldc, t_per, stride, alpha);                                                            \
    break;
  switch (R) { BL_TN(64) BL_TN(128) BL_TN(192) }
#undef BL_TN
  if (splits > 1)
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((int)((stride + 63) / 64)), dim3(256), 0, s, partial_ws, splits, (int)stride, C, (float*)nullptr);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_sumsq_partial_f32(const float* g, int64_t n, float* partial, int32_t nblocks, void* stream) {
  if (!g || !partial) return BL_E_ARG;
  if (n <= 0 || nblocks <= 0 || nblocks > 8192) return BL_E_SHAPE;
  hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)stream, g, (long)n, partial);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_clip_coef_f32(const float* partial, int32_t n, float max_norm, float* out_norm_coef, void* stream) {
  if (!partial || !out_norm_coef) return BL_E_ARG;
  if (n <= 0) return BL_E_SHAPE;
  hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n, max_norm, out_norm_coef);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_adamw_f32(float* p, float* m, float* v, const float* g, const float* norm_coef, int64_t n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int32_t step, bl_bf16* p_bf16,
                            void* stream) {
  if (!p || !m || !v || !g) return BL_E_ARG;
  if (n <= 0 || step <= 0) return BL_E_SHAPE;
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, p, m, v, g, norm_coef,
                     (long)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, p_bf16);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_embed_backward_bf16(const int64_t* ids, int32_t B, int32_t L, const bl_bf16* dx, int32_t dim,
                                      int32_t n_patches, float* dw, void* stream) {
  if (!ids || !dx || !dw) return BL_E_ARG;
  if (B <= 0 || L <= 0 || dim <= 0 || (dim % 8) || n_patches < 0) return BL_E_SHAPE;
  hipLaunchKernelGGL(embed_bwd_kernel, dim3(grid_for((long)B * L * (dim / 8), 256)), dim3(256), 0, (hipStream_t)stream,
                     ids, B, L, dx, dim, n_patches, dw);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_batched_ops(const void* ops_table, const int32_t* block_start, int32_t n_ops, int32_t total_blocks, void* stream) {
  if (!ops_table || !block_start) return BL_E_ARG;
  if (n_ops <= 0 || total_blocks <= 0) return BL_E_SHAPE;
  hipLaunchKernelGGL(batched_ops_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, (const BatchOp*)ops_table, block_start, n_ops);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

static int dropout_args_ok(const void* a, const void* b, const void* seed, int32_t rows, int32_t cols, float p, int64_t lda, int64_t ldb) {
  if (!a || !b || !seed) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || (cols % 8) || (lda % 8) || (ldb % 8) || lda < cols || ldb < cols || (long)rows * cols > 0xffffffffL) return BL_E_SHAPE;
  if (!(p >= 0.f && p < 1.f)) return BL_E_ARG;
  if (!bl_aligned16(a) || !bl_aligned16(b)) return BL_E_ALIGN;
  return BL_OK;
}

extern "C" int bl_dropout_bf16(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, float p, const uint32_t* seed, uint32_t salt,
                               bl_bf16* out, int64_t ldo, void* stream) {
  const int rc = dropout_args_ok(x, out, seed, rows, cols, p, ldx, ldo);
  if (rc != BL_OK) return rc;
  const uint32_t thr = (uint32_t)lrintf(p * 16777216.0f);
  hipLaunchKernelGGL(dropout_kernel, dim3(grid_for((long)rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, (long)ldx, rows,
                     cols, thr, 1.0f / (1.0f - p), seed, salt, out, (long)ldo);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_dropout_grad_fix_bf16(const bl_bf16* u, int64_t ldu, int32_t rows, int32_t cols, float p, const uint32_t* seed,
                                        uint32_t salt, bl_bf16* dx, int64_t lddx, void* stream) {
  const int rc = dropout_args_ok(u, dx, seed, rows, cols, p, ldu, lddx);
  if (rc != BL_OK) return rc;
  const uint32_t thr = (uint32_t)lrintf(p * 16777216.0f);
  hipLaunchKernelGGL(dropout_grad_fix_kernel, dim3(grid_for((long)rows * (cols / 8), 256)), dim3(256), 0, (hipStream_t)stream, u, (long)ldu,
                     rows, cols, thr, 1.0f / (1.0f - p), seed, salt, dx, (long)lddx);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
