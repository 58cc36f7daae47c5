/*
 * bridgelang_hip.h — C ABI of libbridgelang_hip.so (gfx950 / MI355X).
 *
 * The reference (CliffKai/BridgeLang, a thin OpenVLA fork) has no C ABI of its own: its hot path is Python that
 * calls timm / transformers / flash-attn / torch. The drop-in seam is therefore the Python classes
 * in bridgelang_amd/extern/hf/modeling_prismatic.py; *behind* that seam every arithmetic operation goes through the
 * entry points declared here. Each entry point cites the reference call site whose arithmetic it replaces.
 *
 * Conventions
 *   - plain pointers (device memory) and integer sizes only; no torch / C++ types cross the boundary
 *   - every function enqueues work on `stream` (a hipStream_t passed as void*) and returns immediately
 *   - no allocation, no host synchronisation, no global mutable state (thread-compatible, graph-capturable)
 *   - return value: 0 = ok, negative = error (BL_E_*); nothing is thrown across the ABI
 *   - bf16 tensors are raw uint16 bit patterns (`bl_bf16`); row-major; strides in ELEMENTS
 */
#ifndef BRIDGELANG_HIP_H
#define BRIDGELANG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t bl_bf16;

#define BL_OK 0
#define BL_E_SHAPE (-1)   /* unsupported / inconsistent shape        */
#define BL_E_ALIGN (-2)   /* pointer or stride alignment not met     */
#define BL_E_LAUNCH (-3)  /* hipLaunchKernel reported an error       */
#define BL_E_ARG (-4)     /* null pointer / bad enum                 */

/* ---- library info ------------------------------------------------------------------------------------------- */
int bl_abi_version(void);          /* bumps when a signature changes */
const char* bl_build_arch(void);   /* "gfx950" */

/* ---- deterministic synthetic tensors ------------------------------------------------------------------------ */
/* Fills dst[i] = bf16(mean + scale * irwin_hall4(seed, i)) — bit-identical to oracle/synth.py (integer hash +
 * two exactly-rounded fp32 ops). Stands in for `from_pretrained` weights (no checkpoint exists offline): the
 * reference initialises with normal(0, initializer_range), prismatic/extern/hf/modeling_prismatic.py:185-205. */
int bl_fill_synth_bf16(bl_bf16* dst, int64_t n, uint32_t seed, float mean, float scale, void* stream);
/* Same generator for a [rows, cols] logical matrix stored with leading dimension ld >= cols: element (r, c) gets the
 * value of logical index r*cols + c, so padded / interleaved device layouts hold the same numbers as the unpadded
 * tensor. Columns [cols, ld) are NOT written (allocate zeroed). */
int bl_fill_synth_bf16_2d(bl_bf16* dst, int64_t rows, int64_t cols, int64_t ld, uint32_t seed, float mean,
                          float scale, void* stream);

/* ---- weight layout ------------------------------------------------------------------------------------------------- */
/* Every GEMM weight lives in HBM in MFMA-fragment-major order: packed[nt][ks][lane][8] (nt = n/16, ks = k/32) holds
 * W[16*nt + (lane & 15)][32*ks + 8*(lane >> 4) .. +7] — the 16 bytes lane `lane` feeds v_mfma_f32_16x16x32_bf16. Each
 * (nt, ks) block is one contiguous KiB. bl_pack_weight_bf16 converts an nn.Linear-layout matrix src[n, k] (leading
 * dimension ld) once at load time; n % 16 == 0, k % 32 == 0 (pad with zeros first). */
int bl_pack_weight_bf16(const bl_bf16* src, int64_t ld, int64_t n, int64_t k, bl_bf16* dst, void* stream);

/* ---- GEMM: C[M,N] = epilogue(A[M,K] * W[N,K]^T) ------------------------------------------------------------- */
/* Replaces every nn.Linear on the path: timm Attention.qkv/proj + Mlp.fc1/fc2 (created at
 * modeling_prismatic.py:78-101), PrismaticProjector fc1/fc2/fc3 (modeling_prismatic.py:146-158), and the HF Llama
 * q/k/v/o/gate/up/down projections + lm_head (built at modeling_prismatic.py:248-250). fp32 accumulate on MFMA;
 * rounding points follow the bf16 reference (output of each Linear rounded to bf16 before the next elementwise op). */
enum bl_epilogue {
  BL_EPI_NONE = 0,        /* C = bf16(acc)                                                              */
  BL_EPI_BIAS = 1,        /* C = bf16(acc + bias[n])                                                    */
  BL_EPI_BIAS_GELU = 2,   /* t = bf16(acc + bias[n]);  C = bf16(gelu_erf(t))                            */
  BL_EPI_BIAS_RES = 3,    /* t = bf16(acc + bias[n]);  [t = bf16(t*scale[n])];  C = bf16(res + t)       */
  BL_EPI_RES = 4,         /* t = bf16(acc);            C = bf16(res + t)                                */
  BL_EPI_SWIGLU = 5,      /* W rows interleaved (2j = gate_j, 2j+1 = up_j); C[m, j] = silu(g)*u, N/2 wide */
  BL_EPI_F32 = 6,         /* C = acc  (fp32 output)                                                     */
  BL_EPI_F32_BF16R = 7,   /* C = (float)bf16(acc): HF `lm_head(h).float()` — logits are bf16 values upcast     */
  /* Training-step forms (bl_gemm_bf16 only; what autograd keeps / computes around the same nn.Linear under
   * `loss.backward()`, prismatic/training/strategies/base_strategy.py:296-302). The *_KEEP forms write the pre-activation
   * to C (autograd's saved tensor) AND the activation to C2; the *_BWD forms apply the activation's backward to the
   * input-gradient GEMM's result, reading the saved pre-activation through `res`. Bit-identical to the plain epilogue
   * followed by bl_swiglu_bf16 / bl_gelu_bf16 / bl_swiglu_backward_bf16 / bl_gelu_backward_bf16. */
  BL_EPI_SWIGLU_KEEP = 8,     /* C[m, n] = bf16(acc) (gate/up interleaved, N wide); C2[m, j] = silu(g)*u (N/2 wide)  */
  BL_EPI_BIAS_GELU_KEEP = 9,  /* C = t = bf16(acc + bias[n]);  C2 = bf16(gelu_erf(t))                                */
  BL_EPI_SWIGLU_BWD = 10,     /* d = bf16(acc) = dL/d act[m, n];  res = saved gate/up [m, 2n..]; C[m, 2n..2n+1] = (d gate, d up) — C is 2N wide */
  BL_EPI_GELU_BWD = 11        /* d = bf16(acc) = dL/d gelu(t)[m, n];  res = saved t;  C = bf16(d * gelu'(t))          */
};

typedef struct bl_gemm_desc {
  const bl_bf16* A;  int64_t lda;      /* [M, K] activations                                                  */
  const bl_bf16* W;  int64_t ldw;      /* [N, K] weights, fragment-major (bl_pack_weight_bf16); ldw must equal K */
  void* C;           int64_t ldc;      /* [M, N] (or [M, N/2] for SWIGLU); bf16, or fp32 for BL_EPI_F32*      */
  int32_t M, N, K;                     /* K % 64 == 0 (pad in HBM), N % 16 == 0                               */
  int32_t epilogue;                    /* enum bl_epilogue                                                    */
  const bl_bf16* bias;                 /* [N] or NULL                                                         */
  const bl_bf16* scale;                /* [N] LayerScale (modeling_prismatic.py:52-59) or NULL                */
  const bl_bf16* res; int64_t ldres;   /* residual / additive table; row = res_row_mod ? m % res_row_mod : m  */
  int32_t res_row_mod;
  /* output-row remap: out_row = (m / out_group) * out_stride + (m % out_group) + out_offset; rows whose
   * (m % out_group) + out_offset falls outside [0, out_stride) are dropped. out_group == 0 → identity.
   * Used to write patch-embeddings behind the cls/register tokens, to strip prefix tokens at the ViT tap
   * (modeling_prismatic.py:85-87,121-123) and to write projector output straight into LLM embedding rows 1..256
   * (modeling_prismatic.py:383-385). */
  int32_t out_group, out_stride, out_offset;
  /* Optional fused HF LlamaRMSNorm on the A operand (bl_gemm_skinny_bf16 only): A rows are normalised in registers as
   * t = bf16(a * rsqrt(mean(a^2) + eps)), a' = bf16(a_norm_weight[k] * t) before the product. NULL = off. */
  const bl_bf16* a_norm_weight;
  float a_norm_eps;
  /* Optional scratch for bl_gemm_bf16's split-K tail (fp32 partial tiles of the last, partially filled round of
   * 256x256 tiles); 64 MiB always suffices. NULL / too small → the tail runs as 128x128 tiles instead. Contents are
   * undefined after the call; calls that share a workspace must be ordered on one stream. */
  void* workspace;
  int64_t workspace_bytes;
  /* second output of the BL_EPI_*_KEEP epilogues (bf16); NULL otherwise */
  bl_bf16* C2;
  int64_t ldc2;
} bl_gemm_desc;

int bl_gemm_bf16(const bl_gemm_desc* d, void* stream);

/* Skinny GEMM for M <= 16 rows (decode steps and last-row lm_head): weight-streaming, HBM-bound.
 * Supports BL_EPI_NONE / BL_EPI_RES / BL_EPI_SWIGLU / BL_EPI_F32 / BL_EPI_F32_BF16R. Replaces the same nn.Linear modules on the
 * cached-decode branch (modeling_prismatic.py:325-341). */
int bl_gemm_skinny_bf16(const bl_gemm_desc* d, void* stream);

/* The skinny kernel's arithmetic for 16 < M <= 128 stacked rows: the same 8-way K partition (slice w = K/8 contiguous
 * columns), the same per-slice k order on the MFMA, the same slice-order fp32 combine and fused epilogue, so every row's
 * result is bit-identical to bl_gemm_skinny_bf16 on that row alone — whatever batch or row position it is computed in.
 * Used by the merged decode iteration of StaggeredDecodePipeline (the decode steps of n_new-1 batches stacked into one
 * pass over the weights; same nn.Linear modules on the cached-decode branch, modeling_prismatic.py:325-341).
 * K % 256 == 0; epilogues as bl_gemm_skinny_bf16; no out_map, no fused a_norm (see bl_rmsnorm_skinny_bf16). */
int bl_gemm_skinny_rows_bf16(const bl_gemm_desc* d, void* stream);

/* Weight gradient of an nn.Linear straight from row-major buffers: C[M, N] (fp32, BL_EPI_F32 only) = A^T * W with
 * A = dy [K token rows, M] (lda) and W = x [K token rows, N] (ldw), both row-major bf16 — what autograd computes for
 * `weight.grad` in loss.backward() (base_strategy.py:301; torch.nn.functional.linear backward). K is arbitrary (token
 * rows past the end read as zeros); M, N multiples of 8. workspace (optional): fp32 scratch for a K-split last round. */
int bl_gemm_tn_bf16(const bl_gemm_desc* d, void* stream);

/* HF LlamaRMSNorm (transformers modeling_llama.py LlamaRMSNorm.forward; called per decoder layer from the cached-decode
 * branch, modeling_prismatic.py:325-341) in exactly the arithmetic of bl_gemm_skinny_bf16's fused a_norm (same
 * sum-of-squares order, same two roundings): y = bf16(w * bf16(x * rsqrt(mean(x^2) + eps))). dim in {512, 1024, 1536,
 * 4096, 5120, 11008, 13824}. norm kernel + bl_gemm_skinny_rows_bf16 == bl_gemm_skinny_bf16 with a_norm, bit for bit. */
int bl_rmsnorm_skinny_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, bl_bf16* y, int64_t ldy, int32_t rows,
                           int32_t dim, float eps, void* stream);

/* ---- FP8 (OCP e4m3) GEMM family — BASELINE configs[4] "fp8 MFMA GEMMs"; the reference has no fp8 path ------------- */
/* C = epilogue(scale_a[m] * scale_w[n] * (A8 @ W8^T)) on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales, 2x the
 * bf16 MFMA rate). d->A: e4m3 codes [M, K] (lda in bytes, % 16), d->W: e4m3 weight in the fragment-major packing of
 * bl_pack_weight_bf16 applied to the [N, K/2] matrix of byte pairs, K % 128 == 0; scale_a fp32 [M] (per token),
 * scale_w fp32 [N] (per output channel, 16-byte aligned). Epilogues, bias / residual / output as bl_gemm_bf16; no
 * workspace, no fused A-norm. */
int bl_gemm_fp8(const bl_gemm_desc* d, const float* scale_a, const float* scale_w, void* stream);
/* x bf16 [rows, cols] (cols % 8 == 0) -> q e4m3 [rows, cols] (ldq bytes) + scales[row] = amax(row) / 448 (1 for a zero row);
 * q = RNE_e4m3(x * (448 / amax)). */
int bl_quantize_rows_fp8(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, uint8_t* q, int64_t ldq, float* scales,
                         void* stream);
/* Weight-gradient operands of the e4m3 path (dW = dyT . x contracts over tokens: one scale per CHANNEL, token-contiguous
 * codes). bl_colamax_bf16: amax[c] = max_t |x[t, c]| into a PRE-ZEROED fp32 vector (atomic max; several calls may
 * accumulate into one vector). bl_transpose_quantize_fp8: x bf16 [rows = tokens, cols = channels] -> e4m3 codes
 * q[c][t] = rne(x[t, c] * 448 / amax[c]), tokens zero-padded to ldq (a multiple of 64), scales[c] = amax[c] / 448 (1 for an
 * all-zero channel); packed = 0: q row-major [cols, ldq] (bl_gemm_fp8's activation operand), packed = 1: q in the
 * fragment-major packing of its weight operand, [cols/16][ldq/64][64][16 B]. No reference counterpart. */
int bl_colamax_bf16(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, float* amax, void* stream);
int bl_transpose_quantize_fp8(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, const float* amax, uint8_t* q,
                              int64_t ldq, int32_t packed, float* scales, void* stream);

/* ---- normalisation ------------------------------------------------------------------------------------------ */
/* timm Block.norm1/norm2: LayerNorm(eps, affine), fp32 statistics, bf16 out. y may alias x. */
int bl_layernorm_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* b, bl_bf16* y, int64_t ldy,
                      int32_t rows, int32_t dim, float eps, void* stream);
/* HF LlamaRMSNorm: t = bf16(x * rsqrt(mean(x^2) + eps)); y = bf16(w * t). */
int bl_rmsnorm_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, bl_bf16* y, int64_t ldy, int32_t rows,
                    int32_t dim, float eps, void* stream);

/* ---- attention ---------------------------------------------------------------------------------------------- */
/* Flash-style softmax(q k^T * scale [+ causal] [+ key mask]) v, fp32 scores/softmax, bf16 P for the PV product.
 * Replaces timm Attention (SDPA, non-causal, head_dim 64 / 72) and HF LlamaAttention prefill (causal, head_dim 128;
 * the reference callers hard-wire flash_attention_2 / sdpa: deploy.py:67, run_openvla_demo.py:25).
 * Element strides: *_bs batch, *_hs head, *_rs row (token). head_dim contiguous. key_mask: [B, Skv] uint8 or NULL. */
typedef struct bl_attn_desc {
  const bl_bf16* q; int64_t q_bs, q_hs, q_rs;
  const bl_bf16* k; int64_t k_bs, k_hs, k_rs;
  const bl_bf16* v; int64_t v_bs, v_hs, v_rs;
  bl_bf16* o;       int64_t o_bs, o_hs, o_rs;
  const uint8_t* key_mask; int64_t mask_bs;
  int32_t B, H, Sq, Skv, head_dim;
  int32_t causal;      /* 1: key j visible to query i iff j <= i + (Skv - Sq) */
  float scale;         /* head_dim^-0.5 */
} bl_attn_desc;
int bl_attention_bf16(const bl_attn_desc* d, void* stream);
/* Prefill attention with the rotary embedding and the KV-cache write fused in (HF apply_rotary_pos_emb +
 * DynamicCache.update + attention of LlamaAttention.forward in one pass): d->q / d->k / d->v are the UN-rotated thirds of the
 * fused qkv rows; q and k are rotated at positions pos0 + row while they are loaded (same three bf16 roundings as
 * bl_rope_kvcache_bf16), the rotated k and v rows are written to k_cache / v_cache [B, H, cache_len, 128]. head_dim 128,
 * causal, Skv == Sq <= 320. The qkv buffer is left untouched. */
int bl_attention_rope_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab, int32_t pos0, bl_bf16* k_cache,
                           bl_bf16* v_cache, int32_t cache_len, void* stream);
/* Training forward: as bl_attention_bf16, and also writes the base-2 log-sum-exp of the scaled scores of every query
 * row to lse[(b*H + h) * pad32(Sq) + i] (fp32; +inf for a row with no visible key) for bl_attention_backward_bf16. */
int bl_attention_lse_bf16(const bl_attn_desc* d, float* lse, void* stream);
/* Backward of bl_attention_bf16 (what autograd runs under HF LlamaAttention / timm Attention in the reference's
 * `loss.backward()`, prismatic/training/strategies/base_strategy.py:300). d describes the forward call (o = its
 * output); dout uses o's strides; dq / dk / dv use q's / k's / v's strides (so they can alias the three thirds of a
 * fused dqkv row). lse from bl_attention_lse_bf16; delta is scratch of the same size, [B*H*pad32(Sq)] fp32.
 * Any Sq, Skv (causal: Skv >= Sq): up to 320 positions one workgroup per (batch, head) holds the whole sequence in LDS;
 * longer sequences (the collator pads to model_max_length = 2048, prismatic/util/data_utils.py:101-142) stream 256-row
 * chunks against 128-row register blocks — same summation order, same bits. All strides multiples of 8. */
int bl_attention_backward_bf16(const bl_attn_desc* d, const bl_bf16* dout, const float* lse, float* delta, bl_bf16* dq,
                               bl_bf16* dk, bl_bf16* dv, void* stream);
/* Single-query decode attention over a KV cache; kv_len = number of valid keys (same for the whole batch). */
int bl_attention_decode_bf16(const bl_attn_desc* d, void* stream);

/* Training forward: rotate q and k of the fused qkv rows IN PLACE (same arithmetic as bl_rope_kvcache_bf16, no cache). */
int bl_rope_bf16(bl_bf16* qkv, int64_t ld, int32_t B, int32_t S, int32_t H, int32_t head_dim, const bl_bf16* cos_tab,
                 const bl_bf16* sin_tab, int32_t pos0, void* stream);

/* Decode attention with the rotary embedding and the KV-cache append of the NEW token fused in: q / k_new / v_new are
 * the three thirds of the step's fused qkv row (strides d->q_*; k_new = q + H*hd, v_new = q + 2*H*hd elements). q and
 * k_new are rotated at position `pos` (HF apply_rotary_pos_emb, bf16 roundings as bl_rope_kvcache_bf16), k_new', v_new
 * are written to cache row `pos` of d->k / d->v, and attention runs over keys 0..pos (d->Skv must equal pos + 1). */
int bl_attention_decode_rope_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab, int32_t pos,
                                  void* stream);
/* bl_attention_decode_rope_bf16 for a batch of RIGHT-PADDED prompts (HF generation with an attention mask,
 * modeling_prismatic.py:387-390 + transformers' position_ids = cumsum(mask) - 1): sequence b's new token sits at ITS
 * position rope_pos[b] (device int32 [B]) = the number of real tokens before it — that is its rotation angle, the cache
 * row its k / v are written to (over the pad rows' entries) and its key count - 1; `pos` = max_b rope_pos[b] bounds the
 * cache. The cache of every sequence is then laid out exactly as in its own un-padded run: identical results, bit for
 * bit. */
int bl_attention_decode_rope_pos_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab, int32_t pos,
                                      const int32_t* rope_pos, void* stream);

/* n_groups (<= 8) decode iterations of DIFFERENT batches in one launch (continuous batching, pipeline.py): rows
 * g*B .. (g+1)*B-1 of the fused qkv / output buffers belong to group g, which has its own KV caches
 * k_caches[g] / v_caches[g] ([B, H, cache_len, 128], strides from d->k_* / d->v_*) and position pos[g]. The three arrays
 * are HOST arrays read at call time. d->k / d->v / d->Skv are ignored. Per group identical to
 * bl_attention_decode_rope_bf16. */
int bl_attention_decode_rope_grouped_bf16(const bl_attn_desc* d, const bl_bf16* cos_tab, const bl_bf16* sin_tab, int32_t n_groups,
                                          bl_bf16* const* k_caches, bl_bf16* const* v_caches, const int32_t* pos, void* stream);

/* ---- Llama glue -------------------------------------------------------------------------------------------- */
/* Half-split RoPE (HF apply_rotary_pos_emb) on the q and k thirds of a fused qkv buffer [B*S, 3*H*hd], bf16 cos/sin
 * tables [max_pos, hd/2] (built on the host exactly as HF does: fp32 trig → bf16). q is rotated in place; rotated k
 * and v are appended to the KV cache [B, H, cache_len, hd] at positions pos0 .. pos0+S-1. */
int bl_rope_kvcache_bf16(bl_bf16* qkv, int32_t B, int32_t S, int32_t H, int32_t hd, const bl_bf16* cos_tab,
                         const bl_bf16* sin_tab, int32_t pos0, bl_bf16* k_cache, bl_bf16* v_cache,
                         int32_t cache_len, void* stream);
/* Token-embedding gather for the multimodal splice (modeling_prismatic.py:380-385): writes
 * dst[b, 0] = table[ids[b,0]] and dst[b, 1+n_patches+j] = table[ids[b,1+j]]; rows 1..n_patches are left for the
 * projector epilogue. ids int64 [B, L]; dst [B, L+n_patches, dim]. With n_patches = 0 it is a plain gather. */
int bl_embed_splice_bf16(const int64_t* ids, int32_t B, int32_t L, const bl_bf16* table, int32_t dim,
                         int32_t n_patches, bl_bf16* dst, void* stream);
/* Row-wise argmax of fp32 logits [rows, n] (first maximal index, as torch.argmax); out int64 [rows]. */
int bl_argmax_f32(const float* logits, int64_t ld, int32_t rows, int32_t n, int64_t* out, void* stream);

/* Shifted causal-LM cross-entropy (HF LlamaForCausalLM loss; labels prepared by the caller as `targets[row]` =
 * label of the NEXT position, -100 = ignore; base_strategy.py:287-297 consumes `output.loss`). row_loss[rows] receives
 * the per-row loss (0 where ignored); mean_and_count[0] = mean over valid rows, [1] = number of valid rows. */
int bl_cross_entropy_f32(const float* logits, int64_t ld, int32_t rows, int32_t n, const int64_t* targets,
                         int64_t ignore_index, float* row_loss, float* mean_and_count, void* stream);

/* ---- training step: backward + optimizer (base_strategy.py:284-366, fsdp.py:190-246) --------------------------------- */
/* Activation gradients are bf16, parameter gradients / AdamW state / master weights fp32 (train.py:156-157,
 * fsdp.py:140-146). Reductions are deterministic (per-block partials in caller-provided workspaces + a second pass). */
/* dlogits = (softmax(logits) - onehot(target)) / n_valid (0 on ignored rows); mean_and_count from bl_cross_entropy_f32. */
int bl_cross_entropy_backward_f32(const float* logits, int64_t ld, int32_t rows, int32_t n, const int64_t* targets,
                                  int64_t ignore_index, const float* mean_and_count, bl_bf16* dlogits, int64_t ldd,
                                  void* stream);
/* LlamaRMSNorm backward: dx = rstd*(w*dy - xhat*mean(w*dy*xhat)) [+ dres: the residual stream's own gradient];
 * dw[j] = sum_rows dy*bf16(xhat). partial_ws >= ceil(rows/64)*dim floats. */
int bl_rmsnorm_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* dy, int64_t lddy,
                             const bl_bf16* dres, int64_t lddres, bl_bf16* dx, int64_t lddx, float* dw, float* partial_ws,
                             int64_t partial_ws_floats, int32_t rows, int32_t dim, float eps, void* stream);
/* timm nn.LayerNorm backward (vision towers, stages vla-full-train / sandwich): as above with the mean removed;
 * db = column sums of dy. partial_ws needs >= 2 * ceil(rows / rpb) * dim floats for some rpb in {8, 16, ...}. */
int bl_layernorm_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* w, const bl_bf16* dy, int64_t lddy,
                               const bl_bf16* dres, int64_t lddres, bl_bf16* dx, int64_t lddx, float* dw, float* db,
                               float* partial_ws, int64_t partial_ws_floats, int32_t rows, int32_t dim, float eps,
                               void* stream);
/* out[c] = sum_r a[r][c] (bias gradients). partial_ws >= ceil(rows/rpb)*cols floats for some rpb = 16, 32, 64, ...: the smallest
 * that fits is used (more room = more workgroups). */
int bl_colsum_bf16(const bl_bf16* a, int64_t lda, int32_t rows, int32_t cols, float* out, float* partial_ws,
                   int64_t partial_ws_floats, void* stream);
/* SwiGLU on an interleaved [rows, 2*inter] gate/up buffer (the training forward keeps it for the backward). */
int bl_swiglu_bf16(const bl_bf16* gu, int64_t ldg, bl_bf16* act, int64_t lda, int64_t rows, int32_t inter, void* stream);
int bl_swiglu_backward_bf16(const bl_bf16* gu, int64_t ldg, const bl_bf16* dact, int64_t ldd, bl_bf16* dgu, int64_t ldo,
                            int64_t rows, int32_t inter, void* stream);
/* exact-erf GELU and its derivative on [rows, cols] matrices with leading dimensions (projector: nn_utils.py:42-48;
 * timm Mlp). */
int bl_gelu_bf16(const bl_bf16* x, int64_t ldx, bl_bf16* y, int64_t ldy, int64_t rows, int32_t cols, void* stream);
int bl_gelu_backward_bf16(const bl_bf16* x, int64_t ldx, const bl_bf16* dy, int64_t lddy, bl_bf16* dx, int64_t lddx, int64_t rows,
                          int32_t cols, void* stream);
/* Transposed rotation on the q and k thirds of dqkv [B*S, 3*H*hd], in place. */
int bl_rope_backward_bf16(bl_bf16* dqkv, int64_t ld, int32_t B, int32_t S, int32_t H, int32_t hd, const bl_bf16* cos_tab,
                          const bl_bf16* sin_tab, int32_t pos0, void* stream);
/* out[c][r] = in[r][c], rows padded with zeros to rows_pad (the reduction dim of a wgrad GEMM must be a multiple of 64). */
int bl_transpose_pad_bf16(const bl_bf16* in, int64_t ldi, int32_t rows, int32_t cols, bl_bf16* out, int64_t ldo,
                          int32_t rows_pad, void* stream);
/* Same transpose written straight into the fragment-major packed layout of bl_pack_weight_bf16 for the [cols, rows_pad]
 * matrix (cols % 64 == 0, rows_pad % 32 == 0): the packed operand of a wgrad GEMM in one pass. */
int bl_transpose_pack_bf16(const bl_bf16* in, int64_t ld_in, int32_t rows, int32_t cols, bl_bf16* out_packed,
                           int32_t rows_pad, void* stream);
/* The two packers with a destination window: the source becomes k-blocks [kb_offset, kb_offset + K/32) of every n-tile
 * of a packed matrix that has kt_total k-blocks per n-tile — the adapter columns of a K-concatenated LoRA weight
 * [W | s·B] / [Wᵀ | Aᵀ] (training/step.py), refreshed after every optimizer step without touching the frozen part. */
int bl_pack_weight_into_bf16(const bl_bf16* w, int64_t ldw, int64_t N, int64_t K, bl_bf16* out, int64_t kt_total,
                             int64_t kb_offset, void* stream);
int bl_transpose_pack_into_bf16(const bl_bf16* in, int64_t ld_in, int32_t rows, int32_t cols, bl_bf16* out_packed,
                                int32_t rows_pad, int64_t kt_total, int64_t kb_offset, void* stream);
/* ONE launch for a table of independent small ops — copies, (scaled) packs, (scaled) transposing packs: the re-pack plan of a
 * training step (every LoRA adapter's five refresh launches, the optimizer's write-backs). ops_table: device array of
 *   struct { int32 kind, nblocks, rows, cols, rows_pad, bx_count; int64 ld, kt_total, kb_off, n; const void* src; void* dst;
 *            float scale; int32 pad; }        (kind 0: copy n bytes; 1: pack rows x cols into k-blocks [kb_off, ...) of a packed
 *   matrix with kt_total k-blocks per 16-row tile; 2: transposing pack of [rows, cols], rows padded to rows_pad; scale != 1:
 *   bf16(scale * x) first), block_start: device int32 [n_ops] prefix sums of nblocks. Same device code, same results, as
 * bl_copy_bytes / bl_pack_weight_into_bf16 / bl_transpose_pack_into_bf16 (+ bl_scale_bf16). */
int bl_batched_ops(const void* ops_table, const int32_t* block_start, int32_t n_ops, int32_t total_blocks, void* stream);
/* LoRA dropout (PEFT: base(x) + lora_B(lora_A(dropout(x))) * scaling; vla-scripts/finetune.py:101,177 `lora_dropout`).
 * bl_dropout_bf16: out = bf16(x / (1 - p)) where kept, 0 elsewhere; element (row, col) is kept iff the top 24 bits of
 * mix32((row * cols + col) ^ mix32(mix32(*seed) + salt)) >= p * 2^24 (seed on the DEVICE: a replayed plan draws a new mask when the
 * host bumps it; salt = the adapted linear). bl_dropout_grad_fix_bf16: turns the fused input gradient dx = dy.W + u (u = dt.A,
 * un-masked) into dy.W + mask/(1-p) * u with the same mask: dx += (1/(1-p) - 1) * u where kept, dx -= u where dropped. */
int bl_dropout_bf16(const bl_bf16* x, int64_t ldx, int32_t rows, int32_t cols, float p, const uint32_t* seed, uint32_t salt,
                    bl_bf16* out, int64_t ldo, void* stream);
int bl_dropout_grad_fix_bf16(const bl_bf16* u, int64_t ldu, int32_t rows, int32_t cols, float p, const uint32_t* seed, uint32_t salt,
                             bl_bf16* dx, int64_t lddx, void* stream);
/* Global gradient norm (fsdp.py:268-270): per-tensor partial sums of squares, then norm and clip coefficient
 * out_norm_coef = {||g||, min(1, max_norm / (||g|| + 1e-6))} (torch.nn.utils.clip_grad_norm_). */
int bl_sumsq_partial_f32(const float* g, int64_t n, float* partial, int32_t nblocks, void* stream);
int bl_clip_coef_f32(const float* partial, int32_t n, float max_norm, float* out_norm_coef, void* stream);
/* torch.optim.AdamW step `step` (1-based) on fp32 master weights; g is scaled by norm_coef[1] when given; p_bf16
 * (optional) receives the bf16 copy of the updated weights. */
int bl_adamw_f32(float* p, float* m, float* v, const float* g, const float* norm_coef, int64_t n, float lr, float beta1,
                 float beta2, float eps, float weight_decay, int32_t step, bl_bf16* p_bf16, void* stream);
/* timm LayerScale (patched to `scale_factor`, modeling_prismatic.py:52-59) around a residual branch, training form:
 * y = bf16(bf16(u * scale) + res) with the branch output u kept; backward du = dy * scale, dscale = sum_rows dy * u
 * (partial_ws >= ceil(rows / 64) * cols floats). */
int bl_scale_residual_bf16(const bl_bf16* u, int64_t ldu, const bl_bf16* scale, const bl_bf16* res, int64_t ldres, bl_bf16* y,
                           int64_t ldy, int64_t rows, int32_t cols, void* stream);
int bl_layerscale_backward_bf16(const bl_bf16* dy, int64_t lddy, const bl_bf16* u, int64_t ldu, const bl_bf16* scale,
                                bl_bf16* du, int64_t lddu, float* dscale, float* partial_ws, int64_t partial_ws_floats,
                                int32_t rows, int32_t cols, void* stream);
/* Small-output TN GEMM for the LoRA adapter gradients (dA = dt^T x, dB = (ts^T dy)^T; finetune.py:174-189 → PEFT's
 * autograd): C = P^T Q with P [T, R] (R = 64, 128 or 192), Q [T, N] (N % 64 == 0), both row-major bf16, reduction over
 * the T rows, fp32 output scaled by alpha — C [R, N] (transpose_out = 0) or C [N, R] (transpose_out = 1), dense (ldc = N or R). Reads Q
 * once, untransposed. partial_ws (optional, fp32) lets small-N calls split T across workgroups deterministically. */
int bl_gemm_tn_small_bf16(const bl_bf16* P, int64_t ldp, const bl_bf16* Q, int64_t ldq, int32_t T, int32_t R, int32_t N,
                          float* C, int64_t ldc, int32_t transpose_out, float alpha, float* partial_ws, int64_t partial_ws_floats,
                          void* stream);
/* LoRA (vla-scripts/finetune.py:174-189): out = bf16(s * x) for the small rank-space tensors; gradient mask that keeps
 * the off-block entries of a fused adapter's B [n_rows, members * rp] at zero (row n belongs to member n / (n_rows /
 * members), or n % members when the members' rows are interleaved). */
int bl_scale_bf16(const bl_bf16* x, float s, bl_bf16* out, int64_t n, void* stream);
int bl_lora_block_mask_f32(float* g, int32_t n_rows, int32_t R, int32_t rp, int32_t members, int32_t interleave, void* stream);
/* y += a * x on flat fp32 buffers: gradient accumulation over micro-batches with the loss normalised by the number of
 * accumulation steps (vla-scripts/finetune.py:256-262, 307-310). */
int bl_axpy_f32(float* y, const float* x, float a, int64_t n, void* stream);
/* fp32 <-> bf16 casts of flat buffers: the bf16 wire format of the gradient reduce-scatter (fsdp.py:139-147,
 * `reduce_in_full_precision=False`); round-to-nearest-even. 16-byte aligned pointers. */
int bl_cast_f32_bf16(const float* src, bl_bf16* dst, int64_t n, void* stream);
int bl_cast_bf16_f32(const bl_bf16* src, float* dst, int64_t n, void* stream);
/* Device memset / device-to-device copy on `stream` (replayable steps of the training plans). */
int bl_memset_zero(void* dst, int64_t bytes, void* stream);
int bl_copy_bytes(void* dst, const void* src, int64_t bytes, void* stream);
/* Row gather (scatter = 0: dst[r] = src[map(r)]) or scatter (dst[map(r)] = src[r]) with
 * map(r) = (r / group) * stride + offset + r % group — the 256 projected patch rows inside the [B, S, D] embedding
 * buffer (modeling_prismatic.py:343-351) for the projector's backward. */
int bl_map_rows_bf16(const bl_bf16* src, int64_t ld_src, bl_bf16* dst, int64_t ld_dst, int64_t rows, int32_t cols,
                     int32_t group, int32_t stride, int32_t offset, int32_t scatter, void* stream);
/* dW_embed[ids[b,j]] += dx[b, row(j)] over the text positions of the multimodal splice (fp32 atomics; dw pre-zeroed). */
int bl_embed_backward_bf16(const int64_t* ids, int32_t B, int32_t L, const bl_bf16* dx, int32_t dim, int32_t n_patches,
                           float* dw, void* stream);

/* ---- vision glue ------------------------------------------------------------------------------------------- */
/* Frames already at the model resolution: uint8 [B, H, W, 3] → pixel_values [B, 6, H, W] bf16 = to_tensor + the two
 * normalisations of PrismaticImageProcessor.apply_transform (processing_prismatic.py:128-145; DINOv2 ImageNet mean/std,
 * SigLIP 0.5/0.5) in its fp32 operation order, channel-stacked, rounded to bf16 as the call sites do
 * (`.to(torch.bfloat16)`). mean_std: 12 device floats = mean[6] | std[6]. H*W % 8 == 0. (Resize first: bl_resample_pass_u8.) */
int bl_preprocess_u8_bf16(const uint8_t* frames, int32_t B, int32_t height, int32_t width, const float* mean_std, bl_bf16* out,
                          void* stream);
/* One pass of the PIL resize behind `TVF.resize(img, size, BICUBIC)` (processing_prismatic.py:133; Pillow
 * libImaging/Resample.c, 8 bits per channel): out = clip8((2^21 + Σ_k src[first + k]·coef[k]) >> 22) with the host's
 * 22-bit fixed-point coefficient table coefs [out_len, ksize] and bounds [out_len, 2] = (first, count).
 * horizontal != 0: src [B, lines, in_len, 3] → dst [B, lines, out_len, 3]; else src [B, in_len, lines, 3] →
 * dst [B, out_len, lines, 3]. uint8 RGB, packed. Bit-exact against Pillow (tests/test_ops_gpu.py). */
int bl_resample_pass_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t lines, int32_t in_len, int32_t out_len,
                        int32_t horizontal, const int32_t* bounds, const int32_t* coefs, int32_t ksize, void* stream);
/* The eval-time centre crop of the robot loops, `get_vla_action(center_crop=True)` (experiments/robot/openvla_utils.py:
 * 81-155: tf.image.convert_image_dtype → tf.image.crop_and_resize(bilinear, one box) → clip → convert back, saturating):
 * uint8 frames src [B, H, W, 3] → dst [B, out_h, out_w, 3]. Output pixel (i, j) samples ys = y_base + i·y_step,
 * xs = x_base + j·x_step (fp32 constants computed by the host: y1·(H−1) and (y2−y1)·(H−1)/(out_h−1) for the normalised
 * box), bilinear on u8/255, 0 outside the image, then uint8(clamp(clamp(v,0,1)·255.5, 0, 255)). Bit-identical to the host
 * restatement bridgelang_amd/vla/eval_preprocess.py (TensorFlow itself is absent: unpinned against TF). */
int bl_crop_resize_bilinear_u8(const uint8_t* src, uint8_t* dst, int32_t B, int32_t H, int32_t W, int32_t out_h, int32_t out_w,
                               float y_base, float y_step, float x_base, float x_step, void* stream);
/* pixel_values [B, 6, 224, 224] bf16 (processing_prismatic.py:128-145 layout) → 14x14 patch rows for one tower:
 * out[b*256 + py*16 + px, c*196 + i*14 + j] = pixel_values[b, chan0 + c, py*14 + i, px*14 + j]; columns 588..ld-1
 * are zeroed (K padded to a multiple of 64 for the patch-embed GEMM; timm PatchEmbed conv flattening order). */
int bl_im2col_patch14_bf16(const bl_bf16* pixel_values, int32_t B, int32_t chan0, bl_bf16* out, int64_t ld,
                           void* stream);
/* Broadcast `n_prefix` learned prefix tokens (cls + registers) into rows [b*T, b*T + n_prefix) of x [B*T, dim]. */
int bl_write_prefix_tokens_bf16(const bl_bf16* prefix, int32_t n_prefix, int32_t dim, bl_bf16* x, int32_t B,
                                int32_t T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BRIDGELANG_HIP_H */
