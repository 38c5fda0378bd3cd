/*
 * include/llama_gguf_hip.h — C ABI of the MI355X (gfx950) GPU-resident decode engine for llama-gguf.
 *
 * This is the drop-in boundary.  The reference (Lexmata/llama-gguf v0.14.0, Rust; paths below are
 * relative to /root/reference) talks to every GPU engine through
 *
 *     pub trait GpuInference: Send {                       // src/backend/mod.rs:283-296
 *         fn forward(&mut self, token_id: u32) -> BackendResult<Vec<f32>>;
 *         fn prefill_token(&mut self, token_id: u32) -> BackendResult<()>;
 *         fn reset(&mut self);
 *         fn position(&self) -> usize;
 *     }
 *     pub fn from_model(model: LlamaModel, max_seq_len: usize) -> BackendResult<Self>
 *                                                          // src/backend/cuda/gpu_only.rs:426
 *
 * and adapts it to `trait Model` with GpuModelWrapper (src/backend/mod.rs:302-364).  A Rust
 * `impl GpuInference for HipGpuInference` binds the lgh_* entry points below 1:1 (binding shown in
 * INTEGRATION.md); each declaration cites the reference item it replaces.
 *
 * Conventions: plain pointers and sizes, opaque handle, `int` status (0 = ok).  Status codes map
 * 1:1 onto the reference's BackendError variants (src/backend/error.rs:3-37).  The library keeps no
 * thread-local state and binds its device on every call, so a context may be used from any one
 * thread at a time (the reference serialises access with a Mutex, src/backend/mod.rs:303,328-332).
 * All tensor payloads are handed over in NATIVE GGUF order (a weight [in, out] is `out` rows of
 * in/block_size blocks, src/backend/cpu/ops.rs:1120-1122); the library owns any device re-layout.
 */
#ifndef LLAMA_GGUF_HIP_H
#define LLAMA_GGUF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* BackendError variants, src/backend/error.rs:3-37 */
typedef enum lgh_status {
  LGH_OK = 0,
  LGH_NOT_AVAILABLE = 1,         /* BackendError::NotAvailable         (no HIP device) */
  LGH_SHAPE_MISMATCH = 2,        /* BackendError::ShapeMismatch */
  LGH_DTYPE_MISMATCH = 3,        /* BackendError::DTypeMismatch */
  LGH_UNSUPPORTED_DTYPE = 4,     /* BackendError::UnsupportedDType */
  LGH_UNSUPPORTED = 5,           /* BackendError::Unsupported */
  LGH_INVALID_ARGUMENT = 6,      /* BackendError::InvalidArgument      (incl. pos >= max_seq_len) */
  LGH_TENSOR_ERROR = 7,          /* BackendError::Tensor(_) */
  LGH_INITIALIZATION_FAILED = 8, /* BackendError::InitializationFailed */
  LGH_ALLOCATION_FAILED = 9,     /* BackendError::AllocationFailed */
  LGH_OPERATION_FAILED = 10      /* BackendError::OperationFailed      (launch / sync / copy) */
} lgh_status;

/* ggml type ids, src/gguf/constants.rs:92-126 */
typedef enum lgh_ggml_type {
  LGH_TYPE_F32 = 0, LGH_TYPE_F16 = 1, LGH_TYPE_Q4_0 = 2, LGH_TYPE_Q4_1 = 3, LGH_TYPE_Q5_0 = 6,
  LGH_TYPE_Q5_1 = 7, LGH_TYPE_Q8_0 = 8, LGH_TYPE_Q8_1 = 9, LGH_TYPE_Q2_K = 10, LGH_TYPE_Q3_K = 11,
  LGH_TYPE_Q4_K = 12, LGH_TYPE_Q5_K = 13, LGH_TYPE_Q6_K = 14, LGH_TYPE_Q8_K = 15, LGH_TYPE_BF16 = 30
} lgh_ggml_type;

/* What GpuOnlyInference::from_model reads out of ModelConfig and the per-layer Attention scalars
 * (src/backend/cuda/gpu_only.rs:488-502, 527-533, 555-572; src/model/layers.rs:262-298). */
typedef struct lgh_model_desc {
  uint32_t struct_size;              /* sizeof(lgh_model_desc), for ABI evolution */
  uint32_t hidden_size;
  uint32_t intermediate_size;
  uint32_t num_layers;               /* layers of the WHOLE model */
  uint32_t num_heads;
  uint32_t num_kv_heads;
  uint32_t head_dim;                 /* key_length == value_length */
  uint32_t vocab_size;
  uint32_t max_seq_len;              /* KV capacity (the engine's gpu_seq_len, src/engine.rs:554-564) */
  uint32_t num_experts;              /* 0 = dense FFN */
  uint32_t num_experts_per_token;
  uint32_t expert_intermediate_size;
  uint32_t use_neox_rope;            /* RopeType::NeoX vs Normal, src/model/loader.rs:145-162 */
  float norm_eps;
  float rope_freq_base;
  float rope_freq_scale;
  int32_t device_id;                 /* HIP device ordinal */
  /* pipeline stage (src/distributed/pipeline.rs:50-96 partitioning): this context owns layers
   * [layer_begin, layer_end); 0,0 means all.  The first stage owns the embedding, the last one the
   * final norm + output projection. */
  uint32_t layer_begin;
  uint32_t layer_end;
  uint32_t flags;                    /* LGH_FLAG_* */
  uint32_t kv_cache_type;            /* LGH_KV_*.  What the reference's `--kv-cache-type` can select is KVCacheType::{F32, TurboQuantMSE,
                                        TurboQuantProd} (src/config.rs:808-817): LGH_KV_F32, LGH_KV_TQ2 / TQ3, LGH_KV_TQ2_QJL / TQ3_QJL here.
                                        LGH_KV_INT8 / FP8_* are the formats of QuantizedKVCache (src/model/kv_quantized.rs:
                                        11-20), which the reference exports (model/mod.rs:31, lib.rs:77) but which NO forward path and no
                                        CLI flag of it reaches; they are offered for hosts that use that cache type directly.
                                        0 = f32, or int8 when LGH_FLAG_KV_INT8 is set (the field was added after the flag) */
} lgh_model_desc;

enum {
  LGH_KV_F32 = 0,
  LGH_KV_INT8 = 1,                   /* int8 rows + one f32 scale per (kv head, position) */
  LGH_KV_FP8_E4M3 = 2,               /* one byte per element, no scales: the reference's quantize_fp8_e4m3 (mantissa truncated) */
  LGH_KV_FP8_E5M2 = 3,
  /* KVCacheType::TurboQuantMSE { bits } (src/model/mod.rs:182-213): what `--kv-cache-type turboquant2 | tq2` / `turboquant3 | tq3`
   * selects (src/config.rs:808-817) — randomized Hadamard rotation + Lloyd-Max scalar codes, src/model/turboquant/, cache and
   * attention src/model/kv_turboquant.rs.  head_dim 64 or 128. */
  LGH_KV_TQ2 = 4,
  LGH_KV_TQ3 = 5,
  /* KVCacheType::TurboQuantProd { bits }: `turboquant2-qjl | tq2-qjl` / `turboquant3-qjl | tq3-qjl` — the same codes plus, per K row,
   * the sign bits of a Gaussian projection of the quantization residual and the residual's norm (QJL, src/model/turboquant/qjl.rs);
   * scores = codes' dot product + the QJL correction (quant.rs:133-168).  The projection matrices are an input:
   * lgh_set_kv_qjl_matrices. */
  LGH_KV_TQ2_QJL = 6,
  LGH_KV_TQ3_QJL = 7,
};

enum {
  LGH_FLAG_NO_GRAPH = 1u << 0,       /* launch kernels eagerly instead of replaying a hipGraph */
  LGH_FLAG_EXACT_PREFILL = 1u << 2,  /* lgh_prefill_batch feeds the tokens one by one (f32 throughout) instead of the batched f16 GEMM path */
  LGH_FLAG_KV_INT8 = 1u << 4,        /* KV cache as the reference's QuantizedKVCache with KVCacheFormat::Int8 (src/model/kv_quantized.rs): int8
                                        rows + one f32 scale per (kv head, position), a quarter of the f32 cache.  (Same as kv_cache_type =
                                        LGH_KV_INT8; no CLI flag of the reference reaches this format, see kv_cache_type.) */
  /* bits 1, 3, 5, 6, 7 and 24..31 selected the decode structures that round 2 built and measured SLOWER than the default graph of
   * one launch per op (chained FFN with grid barriers, persistent token kernel, flag-ordered launches on two streams, flow
   * launch, split attention merged by the last split / by the output projection: profiles/r02_decode_experiments.md).  They
   * were removed in round 3; lgh_create answers LGH_UNSUPPORTED when one of them is set. */
  LGH_FLAG_REMOVED_MASK = (1u << 1) | (1u << 3) | (1u << 5) | (1u << 6) | (1u << 7) | (0xFFu << 24),
  LGH_FLAG_ATTN_SPLITS_SHIFT = 8,    /* bits 8..15: KV splits per kv-head in decode attention (0 = auto) */
  LGH_FLAG_ATTN_DIRECT_SHIFT = 16    /* bits 16..23: contexts of up to 64 * n rows use the single-launch decode attention (one workgroup
                                        per kv head, no split + combine pair); 0 = the tuned default, 255 = never */
};

typedef struct lgh_ctx lgh_ctx;

/* kernel classes reported by lgh_get_stats (profiling mode) */
enum {
  LGH_K_EMBED = 0, LGH_K_QKV = 1, LGH_K_ATTN = 2, LGH_K_ATTN_COMBINE = 3, LGH_K_WO = 4, LGH_K_GATEUP = 5,
  LGH_K_DOWN = 6, LGH_K_ROUTER = 7, LGH_K_OUTPUT = 8, LGH_K_ARGMAX = 9, LGH_K_MISC = 10,
  LGH_K_TOKEN = 11,   /* (unused since round 3: was the persistent token kernel) */
  LGH_K_COUNT = 16
};

/* kernel SYMBOLS reported by lgh_get_stats (profiling mode): what `rocprofv3 --kernel-trace --stats` groups by.
 * LGH_SYM_MV_* are the instantiations of lgh::mv_kernel<MASK, MAXT> (the VALU dequant mat-vec, csrc/matvec.hip; MASK bits:
 * Q4_K 1, Q5_K 2, Q6_K 4, Q8_0 8, Q4_0 16; MAXT = the instantiation's thread cap).  LGH_SYM_MVQ_* are the instantiations of
 * lgh::mvq_kernel<MASK> (the int8 matrix-core mat-vec, csrc/matvec_mfma.hip; ITS mask bits: Q4_K 1, Q6_K 2, Q5_K 4, Q8_0 8,
 * Q4_0 16). */
enum {
  LGH_SYM_MV_Q4K = 0,      /* lgh::mv_kernel<1u, 1024>  : every matrix of the launch Q4_K */
  LGH_SYM_MV_Q80 = 1,      /* lgh::mv_kernel<8u, 1024>  : Q8_0 */
  LGH_SYM_MV_Q40 = 2,      /* lgh::mv_kernel<16u, 1024> : Q4_0 */
  LGH_SYM_MV_Q5K = 3,      /* lgh::mv_kernel<2u, 768>   : Q5_K */
  LGH_SYM_MV_Q6K = 4,      /* lgh::mv_kernel<4u, 512>   : Q6_K */
  LGH_SYM_MV_Q4K_Q6K = 5,  /* lgh::mv_kernel<5u, 512>   : Q4_K + Q6_K in one launch (fused QKV of Q4_K_M) */
  LGH_SYM_MV_Q5K_Q6K = 6,  /* lgh::mv_kernel<6u, 512>   : Q5_K + Q6_K */
  LGH_SYM_MV_ALL = 7,      /* lgh::mv_kernel<31u, 512>  : any other mix */
  LGH_SYM_F32_MATVEC = 8,  /* lgh::f32_matvec_kernel */
  LGH_SYM_ATTN = 9,        /* lgh::attn_partial_kernel<D, G> */
  LGH_SYM_ATTN_COMBINE = 10,
  LGH_SYM_EMBED = 11,
  LGH_SYM_ARGMAX = 12,     /* argmax_stage1 + argmax_stage2 (two launches, timed together) */
  LGH_SYM_ROUTER = 13,
  LGH_SYM_OTHER = 14,
  LGH_SYM_MVQ_Q4K = 15,    /* lgh::mvq_kernel<1u> : int8 matrix-core mat-vec, every matrix of the launch Q4_K */
  LGH_SYM_MVQ_Q6K = 16,    /* lgh::mvq_kernel<2u> : ... every matrix Q6_K */
  LGH_SYM_MVQ_MIXED = 17,  /* lgh::mvq_kernel<3u> : ... Q4_K and Q6_K matrices in one launch (fused QKV of Q4_K_M) */
  LGH_SYM_MVQ_Q5K = 18,    /* lgh::mvq_kernel<4u> / <6u> : Q5_K (and Q5_K + Q6_K) */
  LGH_SYM_MVQ_Q80_Q40 = 19, /* lgh::mvq_kernel<8u> / <16u> : Q8_0 / Q4_0 */
  LGH_SYM_PTOK = 20,       /* (unused since round 3) */
  LGH_SYM_COUNT = 24
};

typedef struct lgh_stats {
  uint64_t weight_bytes;             /* quantized + f32 weight bytes resident in HBM */
  uint64_t kv_bytes;                 /* KV cache bytes */
  uint64_t scratch_bytes;
  uint64_t tokens_processed;
  uint64_t graph_nodes;              /* kernel nodes in one decode-step graph */
  /* profiling mode only: per-class launches, device time (hipEvent) and algorithmic HBM bytes */
  uint64_t k_launches[LGH_K_COUNT];
  double k_time_us[LGH_K_COUNT];
  uint64_t k_alg_bytes[LGH_K_COUNT];
  /* the same, grouped by kernel symbol (LGH_SYM_*) */
  uint64_t sym_launches[LGH_SYM_COUNT];
  double sym_time_us[LGH_SYM_COUNT];
  uint64_t sym_alg_bytes[LGH_SYM_COUNT];
  /* algorithmic bytes of one decode step at the current position (SURVEY.md §8d formula) */
  uint64_t step_alg_bytes;
  /* profiling mode: mean elapsed time of an EMPTY hipEvent bracket on the launch stream, measured once per
   * profiled token — the fixed cost every k_time_us / sym_time_us sample carries on top of the kernel itself */
  double event_bracket_us;
  uint64_t event_bracket_samples;
  uint64_t overlapped_edges;         /* always 0 (reserved: belonged to the flag-ordered launches removed in round 3) */
} lgh_stats;

/* ---- lifecycle: replaces GpuOnlyInference::from_model (src/backend/cuda/gpu_only.rs:426-726) ---- */

/* Number of HIP devices visible (0 when none / no driver).  Never fails. */
int lgh_device_count(void);

/* CudaDevice::new + scratch/KV allocation (gpu_only.rs:438-440, 527-598). */
int lgh_create(const lgh_model_desc* desc, lgh_ctx** out);

/* upload_model_weights (src/backend/cuda/dequant_weights.rs:244-505): one call per tensor, GGUF name
 * (`token_embd.weight`, `blk.{i}.attn_q.weight`, ... `blk.{i}.ffn_gate_exps.weight` or the loader's
 * per-expert `blk.{i}.ffn_gate.{e}.weight`, `output_norm.weight`, `output.weight`), GGML dims (dim 0
 * fastest), host bytes in native GGUF order.  Tensors of layers outside this stage are ignored. */
int lgh_upload_tensor(lgh_ctx* ctx, const char* gguf_name, uint32_t ggml_type, const uint64_t ne[4],
                      const void* host_bytes, size_t nbytes);

/* Validates that every tensor the stage needs is present, builds RoPE tables and the hipGraphs. */
int lgh_finalize(lgh_ctx* ctx);

/* impl Drop */
void lgh_destroy(lgh_ctx* ctx);

/* ---- GpuInference (src/backend/mod.rs:283-296) ---- */

/* GpuInference::forward (gpu_only.rs:728-771): logits_out = vocab_size f32, caller-owned. */
int lgh_forward(lgh_ctx* ctx, uint32_t token_id, float* logits_out);
/* GpuInference::prefill_token (gpu_only.rs:792-806) */
int lgh_prefill_token(lgh_ctx* ctx, uint32_t token_id);
/* GpuOnlyInference::forward_batch minus the last token (gpu_only.rs:776-790: n prefill_tokens).  Single-stage models
 * (dense or MoE) whose 2-D weights are all in matrix-core tile layouts (Q4_K, Q5_K, Q6_K, Q8_0, Q4_0 with k % 256 == 0)
 * take the batched path: blocks of up to 128 tokens, one f16-MFMA GEMM per weight (MoE: per expert, over
 * the tokens routed to it), causal attention over the block; the KV cache it leaves equals the token-by-token one within
 * 1e-2 relative (f16 operands, f32 accumulation).  Everything else (pipeline stages, other formats), and any
 * context created with LGH_FLAG_EXACT_PREFILL, runs n exact prefill_tokens. */
int lgh_prefill_batch(lgh_ctx* ctx, const uint32_t* tokens, size_t n);
/* 1 when lgh_prefill_batch will take the batched GEMM path for this (finalized) context, else 0 */
int lgh_prefill_is_batched(lgh_ctx* ctx);
/* GpuInference::reset (gpu_only.rs:808-843) — O(1): only the position is rewound (model/mod.rs:110-117) */
void lgh_reset(lgh_ctx* ctx);
/* GpuInference::position (gpu_only.rs:845-847) */
size_t lgh_position(const lgh_ctx* ctx);
/* KVCache::truncate (src/model/mod.rs:130-134): forget the rows from new_len on (no-op when new_len >= position). */
int lgh_kv_truncate(lgh_ctx* ctx, size_t new_len);
/* KVCache::shift_left (src/model/mod.rs:142-172) on the DEVICE cache, with the position following as ChatEngine does
 * (src/engine.rs:1394-1411: the reference shifts its host-side cache only, so a GPU engine keeps the untrimmed rows).
 * Rows [amount, position) move to the front; amount == 0 or >= position clears the cache, as in the reference.  The rows
 * keep the RoPE rotation of their old positions (the reference does not re-rotate either). */
int lgh_kv_shift_left(lgh_ctx* ctx, size_t amount);

/* forward + the bench's arg-max (src/main.rs:1812-1822; ties -> LAST maximal index) on device:
 * only 4 bytes cross PCIe. */
int lgh_forward_argmax(lgh_ctx* ctx, uint32_t token_id, uint32_t* next_token);
/* n_steps of greedy decode with the token fed back ON DEVICE (the loop of src/main.rs:1812-1822);
 * tokens_out[i] = arg-max after step i.  One host sync at the end. */
int lgh_decode_greedy(lgh_ctx* ctx, uint32_t first_token, size_t n_steps, uint32_t* tokens_out);

/* TurboQuant KV cache: the sign vectors of the rotations, [owned layer][kv head][k engine, v engine][head_dim] values of +1 / -1 =
 * HadamardRotation::signs() (src/model/turboquant/rotation.rs:126-129) of TurboQuantKVCache's engines_k / engines_v
 * (src/model/kv_turboquant.rs:44-71).  Call between lgh_create and lgh_finalize; without it the library uses a deterministic
 * stand-in (valid rotations, but not the reference's StdRng stream). */
int lgh_set_kv_rotation_signs(lgh_ctx* ctx, const float* signs, size_t n);
/* one row through the TurboQuant compressor (TurboQuantEngine::compress without QJL, src/model/turboquant/quant.rs:71-103):
 * x[dim] (dim 64 or 128), bits 2 or 3, signs[dim] -> codes[dim / 4 or dim / 8 * 3].  Bit-exact with the reference's arithmetic. */
int lgh_op_tq_compress(int device, int bits, const float* x, size_t dim, const float* signs, uint8_t* codes);
/* TurboQuantProd: the QJL projection matrices of the K engines, [owned layer][kv head][head_dim][head_dim] f32, row i = the i-th
 * projection vector in the order QjlProjector draws it (src/model/turboquant/qjl.rs:44-52: StdRng::seed_from_u64(seed), one
 * StandardNormal sample per (i, j), j fastest; the K engine's seed is 4 * (layer * kv_heads + head) + 1, kv_turboquant.rs:55-58).
 * Call between lgh_create and lgh_finalize on a context with kv_cache_type LGH_KV_TQ2_QJL / TQ3_QJL; without it the library uses
 * a deterministic stand-in (i.i.d. N(0, 1), but not the reference's RNG stream).  (The V engines' projectors are not needed: the
 * reference never reads the QJL bits of V rows, kv_turboquant.rs:154-170.) */
int lgh_set_kv_qjl_matrices(lgh_ctx* ctx, const float* matrices, size_t n);
/* lgh_op_tq_compress with use_qjl (quant.rs:71-103): + qjl_matrix[dim][dim] -> qjl_bits[dim / 64] (bit i % 64 of word i / 64 =
 * (S r)_i >= 0, r the residual) and *residual_norm.  Bit-exact with the reference's arithmetic for the given matrix. */
int lgh_op_tq_compress_qjl(int device, int bits, const float* x, size_t dim, const float* signs, const float* qjl_matrix, uint8_t* codes,
                           uint64_t* qjl_bits, float* residual_norm);

/* ---- multi-sequence decode: the device side of BatchedEngine (src/engine_batched.rs:23-194, 200-330, 355-400) ----
 * The reference keeps one InferenceContext (KV cache + position) per ActiveSequence (engine_batched.rs:84-100, 332-353) and every
 * iteration of its loop calls model.forward for each active sequence in turn (236-290 -> step_sequence 355-400): B sequences read
 * the weights B times.  Here a finalized single-stage context owns `max_batch` <= 16 SLOTS (a slot = one sequence's f32 KV caches
 * + position), and one step feeds one token to each listed slot while reading every weight tile ONCE.  Every sequence's
 * logits are bit-identical to what lgh_forward returns for the same token history (same arithmetic, same summation orders).
 * Dense and MoE models whose matrices are in the matrix-core tile layouts (Q4_K / Q5_K / Q6_K / Q8_0 / Q4_0, k % 256 == 0);
 * anything else answers LGH_UNSUPPORTED.  The slots' caches have the context's kv_cache_type: f32, or the TurboQuant formats
 * (LGH_KV_TQ2 / TQ3 / TQ2_QJL / TQ3_QJL: per slot the packed code rows (+ QJL rows); the reference's BatchedEngine itself only
 * ever creates f32 caches, engine_batched.rs:355-357); int8 / FP8 caches are not batched.  The single-sequence entry points keep
 * working on the context's own cache.  (A serving context is better created with 8 attention splits — flags bits 8..15 — than with
 * the single-sequence default: the split count also sizes the multi-sequence attention launch, profiles/r03e_batched_decode.md.) */
int lgh_batch_create(lgh_ctx* ctx, uint32_t max_batch);                 /* BatchedEngineConfig::max_batch_size (engine_batched.rs:23-41) */
int lgh_batch_reset(lgh_ctx* ctx, uint32_t slot);                       /* create_active_sequence: model.create_context (332-353); O(1) */
size_t lgh_batch_position(lgh_ctx* ctx, uint32_t slot);                 /* ActiveSequence.ctx.position */
/* the slot's prompt (step_sequence's first call: position == 0 -> the whole prompt, 373-379), on the batched prompt path */
int lgh_batch_prefill(lgh_ctx* ctx, uint32_t slot, const uint32_t* tokens, size_t n);
/* One iteration of the loop (236-290): tokens[i] goes to slot slots[i] (distinct slots, 1 <= n_seq <= max_batch).
 * logits_out: n_seq x vocab f32 (row i = sequence i) or NULL; next_tokens: the greedy choice per sequence (last maximal index)
 * or NULL.  InvalidArgument when a slot is full (position >= max_seq_len), nothing is changed then. */
int lgh_forward_multi(lgh_ctx* ctx, const uint32_t* slots, const uint32_t* tokens, uint32_t n_seq, float* logits_out, uint32_t* next_tokens);
/* n_steps greedy iterations with every sequence's token fed back on the device; tokens_out[step * n_seq + i] (or NULL).  One host
 * synchronisation at the end (the bench's timed region for --batch). */
int lgh_decode_greedy_multi(lgh_ctx* ctx, const uint32_t* slots, const uint32_t* first_tokens, uint32_t n_seq, size_t n_steps, uint32_t* tokens_out);

const char* lgh_last_error(const lgh_ctx* ctx);
int lgh_get_stats(lgh_ctx* ctx, lgh_stats* out);
/* on: run eagerly with hipEvent pairs around every launch and accumulate lgh_stats.k_* */
int lgh_set_profiling(lgh_ctx* ctx, int on);
/* Use an external HIP stream for all work; NULL = the context's own (a non-blocking stream).  NB the handle of the legacy
 * default stream IS NULL: a caller whose other work runs on the default stream (torch without an explicit stream) must create
 * a stream, run that work on it and pass it here — otherwise nothing orders that work against the context's kernels. */
int lgh_set_stream(lgh_ctx* ctx, void* hip_stream);
void* lgh_get_stream(lgh_ctx* ctx);
int lgh_synchronize(lgh_ctx* ctx);

/* debug tap: copy the hidden state (f32[hidden_size]) of the last processed token to host */
int lgh_read_hidden(lgh_ctx* ctx, float* out);

/* ---- pipeline stages over xGMI (replaces ShardServer::forward, src/distributed/shard.rs:377-445) ----
 * The hop between stages is one f32[hidden_size] vector; the host side moves it with RCCL
 * send/recv (or a peer copy) between the device buffers returned here. */
int lgh_stage_hidden_buffer(lgh_ctx* ctx, void** device_ptr);     /* in/out residual stream buffer */
/* Run this stage's layers for one token at the context's position.  First stage: embeds token_id;
 * other stages: consume the hidden buffer.  Last stage with want_logits: final norm + output
 * projection (+ device arg-max into *next_token if non-NULL).  Asynchronous on the stream unless
 * logits_out / next_token are requested. */
int lgh_stage_forward(lgh_ctx* ctx, uint32_t token_id, int want_logits, float* logits_out, uint32_t* next_token);
/* A block of up to 128 prompt tokens through this stage's layers on the batched path (lgh_prefill_batch's, for layer
 * ranges).  The first stage reads `tokens`; any other stage reads the block of hidden vectors [n][hidden_size] f32 at
 * lgh_stage_hidden_block_buffer, where a stage that is not the last leaves its output block for the next hop.
 * LGH_UNSUPPORTED when the context has no batched path (feed the tokens through lgh_stage_forward then). */
int lgh_stage_prefill_batch(lgh_ctx* ctx, const uint32_t* tokens, size_t n);
int lgh_stage_hidden_block_buffer(lgh_ctx* ctx, void** device_ptr);
/* Token feedback WITHOUT the host (the reference's coordinator carries every token through host memory,
 * src/distributed/pipeline.rs:50-96): *token_in = the device word the first stage embeds from, *argmax_out = the device word
 * the last stage's arg-max lands in (int32 each).  The host side moves argmax_out -> token_in with an RCCL send/recv or a
 * peer copy on the stages' streams. */
int lgh_stage_io_buffers(lgh_ctx* ctx, void** token_in, void** argmax_out);
/* Stages that share a DEVICE and a stream (lgh_set_stream) can hop inside the producing stage's graph: from now on every token
 * of `ctx` ends with hidden -> *hidden_dst (the next stage's lgh_stage_hidden_buffer; not the last stage) and, in mode 2,
 * arg-max -> *token_dst (the first stage's token_in; last stage only).  NULL switches a hop off.  Both addresses must be memory
 * of ctx's device.  Measured on one MI355X: a hop as a runtime copy between two graph launches costs 12-18 us, as a graph node
 * 3 us (profiles/r03d_pipeline_boundary.md). */
int lgh_stage_set_forward_targets(lgh_ctx* ctx, void* hidden_dst, void* token_dst);
/* lgh_stage_forward without any host value: the token is already in *token_in (first stage) / the hidden vector in the
 * stage's hidden buffer; nothing is copied back and nothing is waited for.  mode 0 = this stage's layers only, 1 = + final
 * norm and output projection (last stage), 2 = + device arg-max into *argmax_out and the token log (last stage). */
int lgh_stage_step(lgh_ctx* ctx, int mode);
/* Last stage: the arg-max tokens logged at positions [pos0, pos0 + n) (one device-to-host copy, synchronises). */
int lgh_stage_read_tokens(lgh_ctx* ctx, size_t pos0, size_t n, uint32_t* out);
/* Last stage: the logits of the last lgh_stage_step(ctx, 1 or 2) (vocab_size f32; one device-to-host copy, synchronises). */
int lgh_stage_read_logits(lgh_ctx* ctx, float* logits_out);

/* ---- the layer pipeline inside the library: ONE process, n_stages stage contexts on the given devices (device_ids NULL:
 * all on desc->device_id), contiguous near-equal layer ranges (earlier stages take the remainder).  Replaces
 * PipelineExecutor::forward (src/distributed/pipeline.rs:50-96) + ShardServer::forward (src/distributed/shard.rs:377-445)
 * for a host that links this library: per token and stage boundary one f32[hidden_size] peer copy over xGMI on the producing
 * stage's stream and an event the consuming stage waits for; in lgh_pipeline_decode_greedy the arg-max token goes from the
 * last stage's device word into the first stage's the same way — no host value crosses a stage boundary per token.  The
 * entry points mirror the single-context ones (GpuInference, src/backend/mod.rs:283-296), so GpuModelWrapper drives a
 * pipeline handle unchanged. ---- */
typedef struct lgh_pipeline lgh_pipeline;
int lgh_pipeline_create(const lgh_model_desc* desc, const int* device_ids, int n_stages, lgh_pipeline** out);
int lgh_pipeline_upload_tensor(lgh_pipeline* p, const char* gguf_name, uint32_t ggml_type, const uint64_t ne[4],
                               const void* host_bytes, size_t nbytes);
int lgh_pipeline_finalize(lgh_pipeline* p);
void lgh_pipeline_destroy(lgh_pipeline* p);
int lgh_pipeline_forward(lgh_pipeline* p, uint32_t token_id, float* logits_out);      /* GpuInference::forward */
int lgh_pipeline_prefill_token(lgh_pipeline* p, uint32_t token_id);                   /* GpuInference::prefill_token */
int lgh_pipeline_decode_greedy(lgh_pipeline* p, uint32_t first_token, size_t n_steps, uint32_t* tokens_out);
void lgh_pipeline_reset(lgh_pipeline* p);                                             /* GpuInference::reset */
size_t lgh_pipeline_position(const lgh_pipeline* p);                                  /* GpuInference::position */
int lgh_pipeline_kv_truncate(lgh_pipeline* p, size_t new_len);                         /* lgh_kv_truncate on every stage */
int lgh_pipeline_stages(const lgh_pipeline* p);
const char* lgh_pipeline_last_error(const lgh_pipeline* p);

/* ---- per-op surface: the `Backend` trait ops on the path (src/backend/mod.rs:29-265), host tensors
 * in / host tensors out, for parity tests of each kernel against the CPU backend. ---- */
int lgh_op_dequantize(int device, uint32_t ggml_type, const void* src, size_t n_elems, float* dst);
/* Backend::vec_mat_q / vec_mat: out[j] = sum_i x[i] * W[i,j], W = n rows of k/bs blocks */
int lgh_op_vec_mat(int device, uint32_t ggml_type, const void* w, const float* x, float* out, size_t k, size_t n);
/* out[m][n] = x[m][k] . W^T through the batched-prefill GEMM (f16 matrix cores, f32 accumulation; m <= 128, n % 16 == 0,
 * k % 256 == 0, type in {Q4_K, Q5_K, Q6_K, Q8_0, Q4_0}); no reference counterpart */
int lgh_op_mat_mat(int device, uint32_t type, const void* w, const float* x, float* out, size_t k, size_t n, size_t m);
int lgh_op_rms_norm(int device, const float* x, const float* w, float eps, float* out, size_t n);
/* Backend::rope with seq_len 1: q [n_heads, d], k [n_kv_heads, d] rotated in place */
int lgh_op_rope(int device, float* q, float* k, size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t pos,
                float freq_base, float freq_scale, int use_neox);
/* Backend::attention_cached: caches [n_kv_heads, max_seq_len, d] */
int lgh_op_attention_cached(int device, const float* q, const float* k_cache, const float* v_cache, float* out,
                            size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_seq_len, float scale,
                            size_t kv_len, int n_splits);
/* simd::silu_mul_inplace: out = silu(gate) * up */
int lgh_op_silu_mul(int device, const float* gate, const float* up, float* out, size_t n);
/* The rest of the per-op `Backend` trait (src/backend/mod.rs:29-265; CPU: src/backend/cpu/ops.rs), host tensors in and out,
 * for `select_gpu_backend` (src/engine.rs:738-812).  add / mul / scale (ops.rs:24-300) and matmul (row-major [m,k] @ [k,n],
 * ops.rs:429-528) are bit-exact with the CPU backend; silu / gelu / softmax (ops.rs:303-385, softmax along the last
 * dimension of [rows, last_dim]) differ by the device exp / tanh only; matvec ([m,k] @ [k], ops.rs:531-570) and matvec_q
 * (ops.rs:922-950) read the same bytes as vec_mat / vec_mat_q with n = m; attention is the causal GQA attention of
 * ops.rs:1353-1472 (q, out [heads, seq, d]; k, v [kv_heads, kv_len, d]). */
int lgh_op_add(int device, const float* a, const float* b, float* out, size_t n);
int lgh_op_mul(int device, const float* a, const float* b, float* out, size_t n);
int lgh_op_scale(int device, const float* a, float scalar, float* out, size_t n);
/* One row of n values through a KV cache format and back: quantize_int8 / quantize_fp8_e4m3 / quantize_fp8_e5m2 and their
 * dequantizers (src/model/kv_quantized.rs:385-565) as the attention launch applies them.  kv_cache_type LGH_KV_INT8 /
 * LGH_KV_FP8_E4M3 / LGH_KV_FP8_E5M2; bytes_out[n]; *scale_out = the int8 row scale (1 for the FP8 formats; may be NULL). */
int lgh_op_kv_roundtrip(int device, uint32_t kv_cache_type, const float* row, size_t n, uint8_t* bytes_out, float* scale_out, float* back_out);
int lgh_op_silu(int device, const float* x, float* out, size_t n);
int lgh_op_gelu(int device, const float* x, float* out, size_t n);
int lgh_op_softmax(int device, const float* x, float* out, size_t rows, size_t last_dim);
int lgh_op_matmul(int device, const float* a, const float* b, float* out, size_t m, size_t k, size_t n);
int lgh_op_matvec(int device, const float* a, const float* x, float* out, size_t m, size_t k);
int lgh_op_matvec_q(int device, uint32_t ggml_type, const void* a, const float* x, float* out, size_t m, size_t k);
int lgh_op_attention(int device, const float* q, const float* k, const float* v, float* out, size_t n_heads, size_t n_kv_heads,
                     size_t seq_len, size_t kv_len, size_t head_dim, float scale);
/* Device-resident weights by tensor name for the per-op surface: `CudaBackend::load_model_weights` and the `b.name()`
 * lookups in its vec_mat / vec_mat_q (src/backend/cuda/mod.rs:121-146, 436-470, 511-575).  A weight is uploaded once
 * (native GGUF bytes; the library re-lays it out as lgh_upload_tensor does) and later calls name it. */
typedef struct lgh_backend lgh_backend;
int lgh_backend_create(int device, lgh_backend** out);
void lgh_backend_destroy(lgh_backend* be);
int lgh_backend_load_weight(lgh_backend* be, const char* name, uint32_t ggml_type, const void* w, size_t k, size_t n);
int lgh_backend_has_weight(const lgh_backend* be, const char* name);
int lgh_backend_vec_mat_q(lgh_backend* be, const char* name, const float* x, float* out, size_t k, size_t n);
const char* lgh_backend_last_error(const lgh_backend* be);
/* fused decode kernels, for kernel-level parity: out = resid + W.(rms_norm(x)*norm_w) etc. */
int lgh_op_norm_vec_mat(int device, uint32_t ggml_type, const void* w, const float* x, const float* norm_w, float eps,
                        float* out, size_t k, size_t n);
int lgh_op_swiglu_vec_mat(int device, uint32_t ggml_type, const void* w_gate, const void* w_up, const float* x,
                          const float* norm_w, float eps, float* out, size_t k, size_t n);

/* Micro-benchmark of the fused quantized mat-vec kernel on device-resident synthetic data:
 * `iters` back-to-back launches timed with hipEvents on the launch stream.  mode 0 = plain matvec,
 * 1 = rmsnorm prologue, 2 = gate/up SwiGLU pair.  `copies` device copies of the weights are cycled so the
 * 256 MiB Infinity Cache does not serve re-reads.  avg_us = mean device time per launch. */
int lgh_bench_vec_mat(int device, uint32_t ggml_type, const void* w, const void* w2, size_t k, size_t n, int mode,
                      int iters, int copies, double* avg_us);

/* ---- GGUF -> HBM direct loader: replaces GgufReader + ModelLoader::parse_config + from_model for this engine
 * (src/gguf/reader.rs:49-104, src/model/loader.rs:62-170, src/backend/cuda/dequant_weights.rs:244-505).  The file is mapped
 * and every tensor the engine knows is uploaded straight from the mapping in native GGUF layout. ---- */
typedef struct lgh_gguf_info {
  uint32_t version;                  /* GGUF version (1..3) */
  uint32_t alignment;                /* general.alignment (default 32) */
  uint64_t n_tensors, n_kv;
  uint64_t data_offset, file_bytes;
  char architecture[64];             /* general.architecture */
  lgh_model_desc desc;               /* what ModelLoader::parse_config reads from the `{arch}.*` keys; max_seq_len = context_length */
} lgh_gguf_info;
/* GgufReader::new + read (src/gguf/reader.rs:24-104), header only, no GPU needed.  Accepts every well-formed GGUF v1-v3 as
 * the reference's reader does; when the `{arch}.*` keys of a model this engine runs are absent, desc.struct_size is 0 (and
 * `err` says which key is missing) while version / counts / alignment / data_offset are still filled.  Failures carry the
 * reference's GgufError texts (src/gguf/error.rs:2-19): "Invalid magic number: ...", "Unsupported GGUF version: N"
 * (status LGH_UNSUPPORTED), "Unexpected end of file". */
int lgh_gguf_inspect(const char* path, lgh_gguf_info* out, char* err, size_t errlen);
/* One metadata value by key: GgufData::get_string / get_u32 / get_u64 / get_f32 / get_bool (src/gguf/types.rs:71-104).
 * type = GGUF value type (0 u8, 1 i8, 2 u16, 3 i16, 4 u32, 5 i32, 6 f32, 7 bool, 8 string, 9 array, 10 u64, 11 i64, 12 f64);
 * integers (sign-extended) and bools in `u`, floats in `f`, strings in `s` (truncated to 255 bytes), arrays: only `arr_len`. */
typedef struct lgh_gguf_value {
  uint32_t type;
  uint32_t reserved;
  uint64_t u;
  double f;
  uint64_t arr_len;
  char s[256];
} lgh_gguf_value;
int lgh_gguf_get(const char* path, const char* key, lgh_gguf_value* out, char* err, size_t errlen);
/* create + upload + finalize from a GGUF file.  max_seq_len 0 = the file's context_length; layer_begin/end 0,0 = all. */
int lgh_load_gguf(const char* path, uint32_t max_seq_len, int device, uint32_t flags, uint32_t layer_begin, uint32_t layer_end,
                  lgh_ctx** out, char* err, size_t errlen);

/* Streaming-read probe: `iters` passes over a `bytes`-long device buffer with 16-byte non-temporal loads from every CU
 * (the access pattern of the weight stream); gbps = bytes * iters / device time.  The practical HBM ceiling the decode
 * roofline is also quoted against (bench.py: hbm_roofline.measured_read_peak_GBps). */
int lgh_bench_hbm_read(int device, size_t bytes, int iters, double* gbps);

#ifdef __cplusplus
}
#endif
#endif
