// glue.hip — small HBM-bound kernels around the GEMMs: synthetic tensor fill, RoPE + KV-cache append, embedding
// gather for the multimodal splice, fp32 argmax, 14x14 im2col, prefix-token broadcast. All use 16-byte vector
// accesses along the contiguous dimension.
#include "bl_common.h"

namespace bl_glue_impl {

// ---- synthetic tensors: integer hash → Irwin-Hall(4) → two exactly-rounded fp32 ops. oracle/synth.py restates this
// bit for bit in numpy, so the CPU oracle and the GPU hold identical weights without any host↔device copy. ----
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint16_t synth_value(uint32_t seed, uint32_t idx, float mean, float scale) {
  const uint32_t h1 = mix32(idx ^ mix32(seed));
  const uint32_t h2 = mix32(h1 + 0x9e3779b9u);
  const int s = (int)((h1 & 0xffffu) + (h1 >> 16) + (h2 & 0xffffu) + (h2 >> 16)) - 131070;
  const float v = __fadd_rn(mean, __fmul_rn((float)s, scale));   // no FMA contraction: matches numpy
  return f2bf(v);
}

__global__ void fill_synth_kernel(uint16_t* dst, long rows, long cols, long ld, uint32_t seed, float mean,
                                  float scale) {
  const long total = rows * cols;   // logical elements only: pad columns [cols, ld) are left untouched
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / cols, c = i - r * cols;
    dst[r * ld + c] = synth_value(seed, (uint32_t)i, mean, scale);
  }
}

// ---- RoPE + KV-cache append. One thread per (token, head, 8 rotation pairs). ----
// HF apply_rotary_pos_emb in bf16: q' = bf16(bf16(q*cos) + bf16(rotate_half(q)*sin)), rotate_half = cat(-x2, x1).
__global__ void rope_kvcache_kernel(uint16_t* qkv, int B, int S, int H, int hd, const uint16_t* cos_tab,
                                    const uint16_t* sin_tab, int pos0, uint16_t* k_cache, uint16_t* v_cache,
                                    int cache_len, long ld) {
  const int half = hd >> 1, cpr = half >> 3;          // 16-byte chunks per half head
  const long total = (long)B * S * H * cpr;
  const long D = (long)H * hd;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const int h = (int)((i / cpr) % H);
    const long tok = i / ((long)cpr * H);
    const int s = (int)(tok % S), b = (int)(tok / S);
    const int pos = pos0 + s;
    const u32x4_t cq = *(const u32x4_t*)(cos_tab + (long)pos * half + ch * 8);
    const u32x4_t sq = *(const u32x4_t*)(sin_tab + (long)pos * half + ch * 8);
    uint16_t* row = qkv + tok * ld;
    const long cache_off = (((long)b * H + h) * cache_len + pos) * hd;
#pragma unroll
    for (int part = 0; part < 2; ++part) {             // 0 = q (in place), 1 = k (→ cache)
      uint16_t* src = row + part * D + (long)h * hd;
      const u32x4_t x1 = *(const u32x4_t*)(src + ch * 8);
      const u32x4_t x2 = *(const u32x4_t*)(src + half + ch * 8);
      u32x4_t o1, o2;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        float r1[2], r2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float a = e ? bfhi(x1[w]) : bflo(x1[w]);
          const float c2 = e ? bfhi(x2[w]) : bflo(x2[w]);
          const float co = e ? bfhi(cq[w]) : bflo(cq[w]);
          const float si = e ? bfhi(sq[w]) : bflo(sq[w]);
          r1[e] = rbf(a * co) + rbf(-c2 * si);          // first half:  x1*cos + (-x2)*sin
          r2[e] = rbf(c2 * co) + rbf(a * si);           // second half: x2*cos + x1*sin
        }
        o1[w] = pack2bf(r1[0], r1[1]);
        o2[w] = pack2bf(r2[0], r2[1]);
      }
      uint16_t* dst = (part == 0 || !k_cache) ? src : (k_cache + cache_off);   // no cache: rotate k in place
      *(u32x4_t*)(dst + ch * 8) = o1;
      *(u32x4_t*)(dst + half + ch * 8) = o2;
    }
    // v: plain copy into the cache (two chunks per thread cover the head together with the other threads)
    if (!v_cache) continue;
    const uint16_t* vsrc = row + 2 * D + (long)h * hd;
    uint16_t* vdst = v_cache + cache_off;
    *(u32x4_t*)(vdst + ch * 8) = *(const u32x4_t*)(vsrc + ch * 8);
    *(u32x4_t*)(vdst + half + ch * 8) = *(const u32x4_t*)(vsrc + half + ch * 8);
  }
}

// ---- embedding gather for the multimodal splice ----
__global__ void embed_splice_kernel(const int64_t* ids, int B, int L, const uint16_t* table, int dim, int n_patches,
                                    uint16_t* dst) {
  const int cpr = dim >> 3;
  const long total = (long)B * L * cpr;
  const int S = L + n_patches;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const long t = i / cpr;
    const int j = (int)(t % L), b = (int)(t / L);
    const long id = ids[(long)b * L + j];
    const int orow = (j == 0) ? 0 : j + n_patches;
    *(u32x4_t*)(dst + ((long)b * S + orow) * dim + ch * 8) = *(const u32x4_t*)(table + id * dim + ch * 8);
  }
}

// ---- argmax over fp32 rows; ties → lowest index (torch.argmax) ----
__global__ __launch_bounds__(256) void argmax_kernel(const float* logits, long ld, int n, int64_t* out) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const float* row = logits + (long)blockIdx.x * ld;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = threadIdx.x * 4; i < n; i += 256 * 4) {
    const f32x4_t q = *(const f32x4_t*)(row + i);     // n % 4 == 0 checked on the host
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (q[e] > best || (q[e] == best && i + e < bi)) { best = q[e]; bi = i + e; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sv[wave] = best; si[wave] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w)
      if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
    out[blockIdx.x] = (int64_t)bi;
  }
}

// ---- 14×14 patch im2col (stride 14) for one 3-channel tower; K padded to ld ≥ 588 with zeros ----
// One thread per (patch, channel, patch-row i): 14 contiguous pixels = 28 B in, 28 B out.
__global__ void im2col_patch14_kernel(const uint16_t* pv, int B, int chan0, uint16_t* out, long ld) {
  const long total = (long)B * 256 * 42;   // 3 channels × 14 rows
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(t % 42), c = ci / 14, i = ci % 14;
    const long pb = t / 42;
    const int p = (int)(pb % 256), b = (int)(pb / 256);
    const int py = p >> 4, px = p & 15;
    const uint16_t* src = pv + (((long)b * 6 + chan0 + c) * 224 + py * 14 + i) * 224 + px * 14;
    uint16_t* dst = out + ((long)b * 256 + p) * ld + c * 196 + i * 14;
#pragma unroll
    for (int j = 0; j < 14; j += 2) *(uint32_t*)(dst + j) = *(const uint32_t*)(src + j);   // 4-byte aligned: all offsets even
    if (ci == 0)
      for (long z = 588; z < ld; z += 2) *(uint32_t*)(out + ((long)b * 256 + p) * ld + z) = 0u;
  }
}

__global__ void write_prefix_kernel(const uint16_t* prefix, int n_prefix, int dim, uint16_t* x, int B, int T) {
  const int cpr = dim >> 3;
  const long total = (long)B * n_prefix * cpr;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ch = (int)(i % cpr);
    const long t = i / cpr;
    const int j = (int)(t % n_prefix), b = (int)(t / n_prefix);
    *(u32x4_t*)(x + ((long)b * T + j) * dim + ch * 8) = *(const u32x4_t*)(prefix + (long)j * dim + ch * 8);
  }
}

// ---- shifted cross-entropy over fp32 logits (HF CausalLM loss, ignore_index = -100) ----
// One workgroup per row: loss[row] = logsumexp(logits[row]) - logits[row][target]; rows whose target is ignored get 0.
__global__ __launch_bounds__(256) void cross_entropy_rows_kernel(const float* logits, long ld, int n,
                                                                 const int64_t* targets, long ignore_index,
                                                                 float* row_loss) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const long tgt = targets[row];
  if (tgt == ignore_index) {          // uniform per workgroup
    if (threadIdx.x == 0) row_loss[row] = 0.f;
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
  if (threadIdx.x == 0) row_loss[row] = logf(red[0] + red[1] + red[2] + red[3]) + mx - lr[tgt];
}

// mean of row_loss over rows whose target is not ignored (single workgroup, fixed summation order → deterministic)
__global__ __launch_bounds__(256) void masked_mean_kernel(const float* row_loss, const int64_t* targets, long ignore_index,
                                                          int rows, float* out) {
  __shared__ float rs[4];
  __shared__ int rc[4];
  float s = 0.f;
  int c = 0;
  for (int i = threadIdx.x; i < rows; i += 256)
    if (targets[i] != ignore_index) { s += row_loss[i]; ++c; }
  s = wave_sum(s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
  if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int cnt = rc[0] + rc[1] + rc[2] + rc[3];
    out[0] = cnt > 0 ? (rs[0] + rs[1] + rs[2] + rs[3]) / (float)cnt : 0.f;
    out[1] = (float)cnt;
  }
}

// ---- weight packing: row-major [N, K] → MFMA-fragment-major [N/16][K/32][64 lanes][8] ----
// One 16-byte chunk per thread: packed chunk (nt, ks, lane) = W[16*nt + (lane & 15)][32*ks + 8*(lane >> 4) .. +7], i.e.
// exactly what lane `lane` feeds v_mfma_f32_16x16x32_bf16 for weight tile nt, k-step ks. Every (nt, ks) block is one
// contiguous KiB, so decode streams weights with whole-line requests and LDS-DMA lands them already in read order.
// kt_total / kb_off: the destination's k-blocks per n-tile and the first k-block written (a [n, k] matrix packed into
// columns [32·kb_off, 32·kb_off + k) of a wider packed matrix: the adapter columns of a K-concatenated LoRA weight)
__global__ void pack_weight_kernel(const uint16_t* src, long ld, long n, long k, uint16_t* dst, long kt_total, long kb_off) {
  const long ks_n = k >> 5;
  const long total = (n >> 4) * ks_n * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long blk = i >> 6, ks = blk % ks_n, nt = blk / ks_n;
    const long row = nt * 16 + (lane & 15), col = ks * 32 + (lane >> 4) * 8;
    *(u32x4_t*)(dst + ((nt * kt_total + kb_off + ks) * 64 + lane) * 8) = *(const u32x4_t*)(src + row * ld + col);
  }
}

// ---- frames already at the model resolution: uint8 HWC → the two normalised, channel-stacked bf16 images ----
// out[b][c][y][x] = bf16(((u8 / 255) − mean0[c]) / std0[c]) for c < 3 and with (mean1, std1) for c ≥ 3: to_tensor + normalize of
// PrismaticImageProcessor.apply_transform (processing_prismatic.py:128-145) in the same fp32 operation order, then the
// `.to(torch.bfloat16)` of the call sites. One thread = 8 consecutive pixels of a row (24 input bytes, 6 × 16-byte stores).
__global__ void preprocess_u8_kernel(const uint8_t* frames, int B, int hw, uint16_t* out, const float* mean_std) {
  const long total = (long)B * (hw >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / (hw >> 3);
    const int p0 = (int)(i - b * (hw >> 3)) * 8;
    const uint8_t* src = frames + (b * hw + p0) * 3;
    uint8_t px[24];
#pragma unroll
    for (int j = 0; j < 6; ++j) *(uint32_t*)(px + 4 * j) = *(const uint32_t*)(src + 4 * j);
#pragma unroll
    for (int c6 = 0; c6 < 6; ++c6) {
      const int c = c6 % 3;
      const float mean = mean_std[c6], sd = mean_std[6 + c6];
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)px[e * 3 + c], 255.0f), mean), sd);
      u32x4_t o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
      *(u32x4_t*)(out + ((b * 6 + c6) * hw + p0)) = o;
    }
  }
}

// One pass of Pillow's 8-bit separable resampling (libImaging/Resample.c ImagingResampleHorizontal_8bpc /
// ImagingResampleVertical_8bpc — what `TVF.resize(PIL image, BICUBIC)` runs, processing_prismatic.py:133): per output
// sample ss = 2^21 + Σ_k src[first + k] · coef[k] in int32 (22-bit fixed-point coefficients from the host), result
// clip8(ss >> 22). The intermediate image between the two passes is uint8, as in Pillow. One thread = one output pixel
// (3 channels). HORIZONTAL: src [B, lines, in_len, 3] → dst [B, lines, out_len, 3]; else src [B, in_len, lines, 3] →
// dst [B, out_len, lines, 3].
template <bool HORIZONTAL>
__global__ void resample_pass_kernel(const uint8_t* src, uint8_t* dst, int B, int lines, int in_len, int out_len,
                                     const int* bounds, const int* kk, int ksize) {
  const long total = (long)B * lines * out_len;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int o, line;
    long b;
    if (HORIZONTAL) { o = (int)(i % out_len); line = (int)((i / out_len) % lines); b = i / ((long)out_len * lines); }
    else { line = (int)(i % lines); o = (int)((i / lines) % out_len); b = i / ((long)out_len * lines); }
    const int first = bounds[2 * o], n = bounds[2 * o + 1];
    const int* k = kk + (long)o * ksize;
    int s0 = 1 << 21, s1 = 1 << 21, s2 = 1 << 21;
    const long step = HORIZONTAL ? 3 : (long)lines * 3;
    const uint8_t* p = HORIZONTAL ? src + ((b * lines + line) * in_len + first) * 3 : src + ((b * in_len + first) * lines + line) * 3;
    for (int x = 0; x < n; ++x, p += step) {
      const int c = k[x];
      s0 += p[0] * c; s1 += p[1] * c; s2 += p[2] * c;
    }
    uint8_t* q = HORIZONTAL ? dst + ((b * lines + line) * out_len + o) * 3 : dst + ((b * out_len + o) * lines + line) * 3;
    q[0] = (uint8_t)min(max(s0 >> 22, 0), 255);
    q[1] = (uint8_t)min(max(s1 >> 22, 0), 255);
    q[2] = (uint8_t)min(max(s2 >> 22, 0), 255);
  }
}

// tf.image.crop_and_resize for uint8 frames, as the robot evaluation loops apply it (`get_vla_action(center_crop=True)`,
// experiments/robot/openvla_utils.py:81-155: convert_image_dtype(uint8 → float32) → crop_and_resize(bilinear, one centred
// box) → clip [0, 1] → convert_image_dtype(float32 → uint8, saturate) ), one thread per output pixel (3 channels):
//   ys = y_base + i · y_step, xs likewise (fp32, the two constants come from the host so both sides use the same bits);
//   0 outside [0, H-1] × [0, W-1]; else, on p = u8 · (1/255), TF's own form (crop_and_resize_op.cc): top = tl + (tr − tl)·x_lerp,
//   bottom = bl + (br − bl)·x_lerp, v = top + (bottom − top)·y_lerp with lerp = in − floor(in), neighbours floor / ceil
//   out = (uint8) clamp(clamp(v, 0, 1) · 255.5, 0, 255)       (TF scales by 255.5 and truncates when it saturates).
// Every fp32 operation is a separate, individually rounded instruction (no FMA contraction), in the order of the host
// restatement vla/eval_preprocess.py::crop_and_resize_bilinear — results are bit-identical to it.
__global__ void crop_resize_bilinear_kernel(const uint8_t* src, uint8_t* dst, int B, int H, int W, int oh, int ow,
                                            float y_base, float y_step, float x_base, float x_step) {
  // Contraction OFF for this body, and plain operators only: HIP's __fmul_rn / __fadd_rn are header functions compiled under
  // the default -ffp-contract=fast, so after inlining hipcc still fuses them into v_fma (seen in the ISA and as 1-LSB
  // differences against the host restatement); operators written here carry this pragma's setting.
#pragma clang fp contract(off)
  const long total = (long)B * oh * ow;
  for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
    const int j = (int)(t % ow), i = (int)((t / ow) % oh);
    const long b = t / ((long)ow * oh);
    const float iy = (float)i * y_step, jx = (float)j * x_step;
    const float ys = y_base + iy, xs = x_base + jx;
    uint8_t* q = dst + t * 3;
    if (!(ys >= 0.0f && ys <= (float)(H - 1) && xs >= 0.0f && xs <= (float)(W - 1))) { q[0] = q[1] = q[2] = 0; continue; }
    const float fy = floorf(ys), fx = floorf(xs);
    const float wy = ys - fy, wx = xs - fx;                        // TF: y_lerp = in_y - top_y_index
    const int y0 = min(max((int)fy, 0), H - 1), y1 = min(max((int)ceilf(ys), 0), H - 1);
    const int x0 = min(max((int)fx, 0), W - 1), x1 = min(max((int)ceilf(xs), 0), W - 1);
    const uint8_t* r0 = src + (b * H + y0) * (long)W * 3;
    const uint8_t* r1 = src + (b * H + y1) * (long)W * 3;
    const float k = 1.0f / 255.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float p00 = (float)r0[x0 * 3 + c] * k, p01 = (float)r0[x1 * 3 + c] * k;
      const float p10 = (float)r1[x0 * 3 + c] * k, p11 = (float)r1[x1 * 3 + c] * k;
      const float dt = p01 - p00, db = p11 - p10;                  // TF's form: top = tl + (tr - tl)·x_lerp, …
      const float et = dt * wx, eb = db * wx;
      const float top = p00 + et, bot = p10 + eb;
      const float dv = bot - top;
      const float ev = dv * wy;
      float v = top + ev;
      v = fminf(fmaxf(v, 0.0f), 1.0f);
      const float s = v * 255.5f;
      q[c] = (uint8_t)fminf(fmaxf(s, 0.0f), 255.0f);
    }
  }
}

inline int grid_for(long total, int block) {
  long g = (total + block - 1) / block;
  return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));   // cap at 256 CUs × 8 and grid-stride the rest
}

}  // namespace bl_glue_impl
using namespace bl_glue_impl;

extern "C" int bl_abi_version(void) { return 2; }   // 2: leading dimensions on rope / gelu, alpha on the TN GEMM
extern "C" const char* bl_build_arch(void) { return "gfx950"; }

extern "C" int bl_fill_synth_bf16_2d(bl_bf16* dst, int64_t rows, int64_t cols, int64_t ld, uint32_t seed, float mean,
                                     float scale, void* stream) {
  if (!dst) return BL_E_ARG;
  if (rows <= 0 || cols <= 0 || ld < cols || rows * cols > 0xffffffffLL) return BL_E_SHAPE;
  hipLaunchKernelGGL(fill_synth_kernel, dim3(grid_for(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, dst,
                     (long)rows, (long)cols, (long)ld, seed, mean, scale);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
extern "C" int bl_pack_weight_bf16(const bl_bf16* src, int64_t ld, int64_t n, int64_t k, bl_bf16* dst, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (n <= 0 || k <= 0 || (n % 16) || (k % 32) || ld < k) return BL_E_SHAPE;
  if ((ld % 8) || !bl_aligned16(src) || !bl_aligned16(dst)) return BL_E_ALIGN;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(n * k / 8, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld, (long)n, (long)k, dst, (long)(k / 32), 0L);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_pack_weight_into_bf16(const bl_bf16* src, int64_t ld, int64_t n, int64_t k, bl_bf16* dst, int64_t kt_total,
                                        int64_t kb_offset, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (n <= 0 || k <= 0 || (n % 16) || (k % 32) || ld < k || (ld % 8) || kb_offset < 0 || kb_offset + k / 32 > kt_total)
    return BL_E_SHAPE;
  if (!bl_aligned16(src) || !bl_aligned16(dst)) return BL_E_ALIGN;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid_for(n * k / 8, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld, (long)n, (long)k, dst, (long)kt_total, (long)kb_offset);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_fill_synth_bf16(bl_bf16* dst, int64_t n, uint32_t seed, float mean, float scale, void* stream) {
  return bl_fill_synth_bf16_2d(dst, 1, n, n, seed, mean, scale, stream);
}

extern "C" int bl_rope_kvcache_bf16(bl_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t hd, const bl_bf16* cos_tab,
                                    const bl_bf16* sin_tab, int32_t pos0, bl_bf16* k_cache, bl_bf16* v_cache,
                                    int32_t cache_len, void* stream) {
  if (!qkv || !cos_tab || !sin_tab || !k_cache || !v_cache) return BL_E_ARG;
  if (B <= 0 || S <= 0 || H <= 0 || hd <= 0 || (hd % 16) || pos0 < 0 || pos0 + S > cache_len) return BL_E_SHAPE;
  if (!bl_aligned16(qkv) || !bl_aligned16(cos_tab) || !bl_aligned16(sin_tab) || !bl_aligned16(k_cache) ||
      !bl_aligned16(v_cache))
    return BL_E_ALIGN;
  const long total = (long)B * S * H * (hd / 16);
  hipLaunchKernelGGL(rope_kvcache_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, qkv, B, S, H,
                     hd, cos_tab, sin_tab, pos0, k_cache, v_cache, cache_len, 3L * H * hd);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_rope_bf16(bl_bf16* qkv, int64_t ld, int32_t B, int32_t S, int32_t H, int32_t hd, const bl_bf16* cos_tab,
                            const bl_bf16* sin_tab, int32_t pos0, void* stream) {
  if (!qkv || !cos_tab || !sin_tab) return BL_E_ARG;
  if (B <= 0 || S <= 0 || H <= 0 || hd <= 0 || (hd % 16) || pos0 < 0 || ld < 3L * H * hd || (ld % 8)) return BL_E_SHAPE;
  if (!bl_aligned16(qkv) || !bl_aligned16(cos_tab) || !bl_aligned16(sin_tab)) return BL_E_ALIGN;
  const long total = (long)B * S * H * (hd / 16);
  hipLaunchKernelGGL(rope_kvcache_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, qkv, B, S, H,
                     hd, cos_tab, sin_tab, pos0, (uint16_t*)nullptr, (uint16_t*)nullptr, 0, (long)ld);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_embed_splice_bf16(const int64_t* ids, int32_t B, int32_t L, const bl_bf16* table, int32_t dim,
                                    int32_t n_patches, bl_bf16* dst, void* stream) {
  if (!ids || !table || !dst) return BL_E_ARG;
  if (B <= 0 || L <= 0 || dim <= 0 || (dim % 8) || n_patches < 0) return BL_E_SHAPE;
  if (!bl_aligned16(table) || !bl_aligned16(dst)) return BL_E_ALIGN;
  const long total = (long)B * L * (dim / 8);
  hipLaunchKernelGGL(embed_splice_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, ids, B, L,
                     table, dim, n_patches, dst);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_argmax_f32(const float* logits, int64_t ld, int32_t rows, int32_t n, int64_t* out, void* stream) {
  if (!logits || !out) return BL_E_ARG;
  if (rows <= 0 || n <= 0 || (n % 4) || (ld % 4)) return BL_E_SHAPE;
  if (!bl_aligned16(logits)) return BL_E_ALIGN;
  hipLaunchKernelGGL(argmax_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, n, out);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_preprocess_u8_bf16(const uint8_t* frames, int32_t B, int32_t height, int32_t width, const float* mean_std,
                                     bl_bf16* out, void* stream) {
  if (!frames || !mean_std || !out) return BL_E_ARG;
  const long hw = (long)height * width;
  if (B <= 0 || height <= 0 || width <= 0 || (hw % 8)) return BL_E_SHAPE;
  if ((((uintptr_t)frames) & 3) || !bl_aligned16(out)) return BL_E_ALIGN;
  hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for((long)B * (hw / 8), 256)), dim3(256), 0, (hipStream_t)stream, frames,
                     B, (int)hw, out, mean_std);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_resample_pass_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t lines, int32_t in_len, int32_t out_len,
                                  int32_t horizontal, const int32_t* bounds, const int32_t* coefs, int32_t ksize, void* stream) {
  if (!src || !dst || !bounds || !coefs) return BL_E_ARG;
  if (B <= 0 || lines <= 0 || in_len <= 0 || out_len <= 0 || ksize <= 0) return BL_E_SHAPE;
  const long total = (long)B * lines * out_len;
  if (horizontal)
    hipLaunchKernelGGL(resample_pass_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                       lines, in_len, out_len, bounds, coefs, ksize);
  else
    hipLaunchKernelGGL(resample_pass_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, B,
                       lines, in_len, out_len, bounds, coefs, ksize);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_crop_resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t H, int32_t W, int32_t out_h,
                                          int32_t out_w, float y_base, float y_step, float x_base, float x_step, void* stream) {
  if (!src || !dst) return BL_E_ARG;
  if (B <= 0 || H <= 0 || W <= 0 || out_h <= 0 || out_w <= 0) return BL_E_SHAPE;
  hipLaunchKernelGGL(crop_resize_bilinear_kernel, dim3(grid_for((long)B * out_h * out_w, 256)), dim3(256), 0, (hipStream_t)stream,
                     src, dst, B, H, W, out_h, out_w, y_base, y_step, x_base, x_step);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_im2col_patch14_bf16(const bl_bf16* pixel_values, int32_t B, int32_t chan0, bl_bf16* out, int64_t ld,
                                      void* stream) {
  if (!pixel_values || !out) return BL_E_ARG;
  if (B <= 0 || (chan0 != 0 && chan0 != 3) || ld < 588 || (ld % 8)) return BL_E_SHAPE;
  if ((((uintptr_t)pixel_values) & 3) || !bl_aligned16(out)) return BL_E_ALIGN;
  const long total = (long)B * 256 * 42;
  hipLaunchKernelGGL(im2col_patch14_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     pixel_values, B, chan0, out, (long)ld);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_write_prefix_tokens_bf16(const bl_bf16* prefix, int32_t n_prefix, int32_t dim, bl_bf16* x, int32_t B,
                                           int32_t T, void* stream) {
  if (!prefix || !x) return BL_E_ARG;
  if (n_prefix <= 0 || dim <= 0 || (dim % 8) || B <= 0 || T < n_prefix) return BL_E_SHAPE;
  if (!bl_aligned16(prefix) || !bl_aligned16(x)) return BL_E_ALIGN;
  const long total = (long)B * n_prefix * (dim / 8);
  hipLaunchKernelGGL(write_prefix_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, prefix,
                     n_prefix, dim, x, B, T);
  BL_CHECK_LAUNCH();
  return BL_OK;
}

extern "C" int bl_cross_entropy_f32(const float* logits, int64_t ld, int32_t rows, int32_t n, const int64_t* targets,
                                    int64_t ignore_index, float* row_loss, float* mean_and_count, void* stream) {
  if (!logits || !targets || !row_loss || !mean_and_count) return BL_E_ARG;
  if (rows <= 0 || n <= 0 || (n % 4) || (ld % 4)) return BL_E_SHAPE;
  if (!bl_aligned16(logits)) return BL_E_ALIGN;
  hipLaunchKernelGGL(cross_entropy_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, logits, (long)ld, n,
                     targets, (long)ignore_index, row_loss);
  hipLaunchKernelGGL(masked_mean_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_loss, targets,
                     (long)ignore_index, rows, mean_and_count);
  BL_CHECK_LAUNCH();
  return BL_OK;
}
