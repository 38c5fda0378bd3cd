/*
 * oracle/oracle.h — C ABI of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This library is a CPU restatement of the arithmetic of the
 * reference's CPU backend (Lexmata/llama-gguf v0.14.0) for the single-stream decode hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load
 * it, and only as the checker / timed CPU baseline.  The product (llama-gguf_amd/) never
 * links, loads or calls it.
 *
 * Pinning status: every function is checked against the known-answer tests the reference's
 * own test-suite holds for this path (tests/test_oracle_kat.py lists each one with its
 * reference file:line).  The reference holds NO golden vector for any K-quant block, for a
 * fused quantized dot or for a whole forward pass, and the reference (Rust) cannot be built
 * in this image; for those, parity is pinned only by this restatement plus the
 * self-consistency identity dot_qX(blocks, x) == sum(dequantize_qX(blocks) * x).
 * TurboQuant (turboquant.cpp): pinned by the unit tests of codebook.rs / rotation.rs / quant.rs / qjl.rs / kv_turboquant.rs /
 * simd.rs that do not depend on the reference's RNG stream; the rotations' sign vectors and the QJL projection matrices are INPUTS
 * (the reference draws them from rand's StdRng / rand_distr's StandardNormal, which are not restated), so the tests that depend on
 * that stream are restated as properties over inputs drawn in the test — parity for a model run with the reference's OWN seeds is
 * therefore pinned only up to "same signs / matrices in, same codes and scores out".
 *
 * Each function cites the reference file:line (relative to /root/reference) it follows.
 */
#ifndef LLAMA_GGUF_ORACLE_H
#define LLAMA_GGUF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ggml type ids (src/gguf/constants.rs:92-126) */
enum {
  ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 6, ORC_Q5_1 = 7,
  ORC_Q8_0 = 8, ORC_Q8_1 = 9, ORC_Q2_K = 10, ORC_Q3_K = 11, ORC_Q4_K = 12, ORC_Q5_K = 13,
  ORC_Q6_K = 14, ORC_Q8_K = 15, ORC_BF16 = 30
};

/* dot_f32 ISA variant (src/backend/cpu/simd.rs:80-100 dispatches at run time) */
enum { ORC_ISA_AUTO = 0, ORC_ISA_SCALAR = 1, ORC_ISA_AVX2 = 2, ORC_ISA_AVX512 = 3 };

void orc_set_isa(int isa);      /* AUTO = what the reference would pick on this host */
int orc_get_isa(void);          /* resolved ISA (never AUTO) */
void orc_set_threads(int n);    /* worker threads for per-output-column parallel loops (rayon stand-in) */
int orc_get_threads(void);

size_t orc_block_size(int type);  /* elements per block, 0 if unknown */
size_t orc_block_bytes(int type); /* bytes per block, 0 if unknown */

uint16_t orc_f32_to_f16(float f);
float orc_f16_to_f32(uint16_t h);

/* n = number of elements, must be a multiple of the block size. Return 0 on success. */
int orc_quantize(int type, const float* in, size_t n, void* out);
int orc_dequantize(int type, const void* in, size_t n, float* out);

/* fused quantized dot (simd.rs:931-1166); returns NaN for types without a fused dot */
float orc_dot_q(int type, const void* blocks, const float* x, size_t k);
int orc_has_fused_dot(int type);
float orc_dot_f32(const float* a, const float* b, size_t n);

/* out[j] = sum_i x[i] * W[i,j];  W = n rows of k/bs blocks (ops.rs:1008-1039,1123-1199) */
int orc_vec_mat_q(int type, const void* w, const float* x, float* out, size_t k, size_t n);
/* f32 weights, strictly sequential sum (ops.rs:959-1002) */
void orc_vec_mat_f32(const float* w, const float* x, float* out, size_t k, size_t n);

void orc_rms_norm(const float* x, const float* w, float eps, float* out, size_t n);
void orc_rope(float* q, float* k, size_t n_heads, size_t n_kv_heads, size_t seq_len,
              size_t head_dim, size_t pos, float freq_base, float freq_scale, int use_neox);
void orc_attention_cached(const float* q, const float* k_cache, const float* v_cache, float* out,
                          size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_seq_len,
                          float scale, size_t kv_len);
void orc_softmax_inplace(float* x, size_t n);
/* per-op Backend surface (src/backend/mod.rs:29-265; CPU: ops.rs) */
void orc_add(const float* a, const float* b, float* out, size_t n);
void orc_mul(const float* a, const float* b, float* out, size_t n);
void orc_scale(const float* a, float s, float* out, size_t n);
void orc_gelu(const float* x, float* out, size_t n);
void orc_softmax_rows(const float* x, float* out, size_t rows, size_t last_dim);
void orc_matmul(const float* a, const float* b, float* c, size_t m, size_t k, size_t n);
void orc_matvec(const float* a, const float* x, float* out, size_t m, size_t k);
void orc_attention(const float* q, const float* k, const float* v, float* out, size_t n_heads, size_t n_kv_heads, size_t seq_len,
                   size_t kv_len, size_t head_dim, float scale);
void orc_silu(const float* x, float* out, size_t n);
void orc_silu_mul_inplace(float* gate, const float* up, size_t n);
float orc_max_f32(const float* x, size_t n);
float orc_sum_f32(const float* a, size_t n);   /* simd.rs:449-487 */
void orc_axpy_f32(float alpha, const float* x, float* y, size_t n);

/* bench arg-max (main.rs:1815-1821): last maximal index */
uint32_t orc_argmax_last(const float* logits, size_t n);
/* Sampler greedy branch (sampling/mod.rs:224-242): softmax, then last maximal index */
uint32_t orc_greedy_sample(const float* logits, size_t n);

/* MoE router (moe.rs:128-198): w is [n_experts][hidden] f32. */
void orc_moe_route(const float* h, const float* w, size_t hidden, size_t n_experts, size_t top_k,
                   int normalize, uint32_t* idx_out, float* weight_out);

/* ---------------- whole-model forward (llama.rs:275-362) ---------------- */

typedef struct orc_config {
  uint32_t hidden_size, intermediate_size, num_layers, num_heads, num_kv_heads, head_dim;
  uint32_t vocab_size, max_seq_len;
  uint32_t num_experts, num_experts_per_token, expert_intermediate_size;
  uint32_t use_neox_rope;
  float norm_eps, rope_freq_base, rope_freq_scale;
} orc_config;

typedef struct orc_model orc_model;

orc_model* orc_model_create(const orc_config* cfg);
void orc_model_destroy(orc_model* m);
/* GGUF tensor names (blk.{i}.attn_q.weight ...); ne = GGML dims (dim 0 fastest).
 * borrow != 0: the oracle keeps the pointer (caller keeps the bytes alive); else it copies. */
int orc_model_add_tensor(orc_model* m, const char* name, int type, const uint64_t ne[4],
                         const void* data, size_t nbytes, int borrow);
int orc_model_finalize(orc_model* m);
const char* orc_model_last_error(const orc_model* m);
/* Model::forward(tokens, ctx): layer-major over the tokens, logits of the last token.
 * faithful_embedding != 0 re-dequantizes the whole embedding table on every call like
 * llama.rs:288 does (timing fidelity only; the values are identical). */
int orc_model_forward(orc_model* m, const uint32_t* tokens, size_t n_tokens, float* logits,
                      int faithful_embedding);
void orc_model_reset(orc_model* m);
size_t orc_model_position(const orc_model* m);
/* KVCache::truncate / shift_left (model/mod.rs:130-172); the position follows the cache (engine.rs:1407-1408) */
void orc_model_kv_truncate(orc_model* m, size_t new_len);
void orc_model_kv_shift_left(orc_model* m, size_t amount);
/* the reference's int8 KV format (src/model/kv_quantized.rs:385-410, KVCacheFormat::Int8): one scale per head row */
void orc_kv_quantize_int8(const float* x, size_t n, int8_t* q, float* scale);
void orc_kv_dequantize_int8(const int8_t* q, float scale, size_t n, float* out);
/* on: K/V rows pass through that format on their way into the cache (QuantizedKVCache::write_kv / read_k_range) */
void orc_model_set_kv_int8(orc_model* m, int on);
/* the reference's FP8 KV formats (kv_quantized.rs:413-565, KVCacheFormat::Fp8E4M3 / Fp8E5M2): fmt 1 = E4M3, 2 = E5M2 */
uint8_t orc_kv_quantize_fp8(int fmt, float value);
float orc_kv_dequantize_fp8(int fmt, uint8_t bits);
/* fmt 1 / 2: K/V rows pass through that format on their way into the cache; 0: off */
void orc_model_set_kv_fp8(orc_model* m, int fmt);

/* ---- TurboQuant KV cache (oracle/turboquant.cpp): what `--kv-cache-type tq2 | tq3` selects in the reference
 * (src/config.rs:808-817; src/model/turboquant/{rotation,codebook,quant}.rs, src/model/kv_turboquant.rs).  The rotations' sign
 * vectors are inputs (HadamardRotation::signs()); the QJL variants are not restated. */
size_t orc_tq_padded_dim(size_t dim);
int orc_tq_codebook(size_t dim, int bits, float* centroids, float* boundaries);
uint8_t orc_tq_quantize(const float* boundaries, int bits, float val);
size_t orc_tq_packed_bytes(int bits, size_t count);
void orc_tq_quantize_vector(size_t dim, int bits, const float* data, size_t count, uint8_t* out);
void orc_tq_dequantize_vector(size_t dim, int bits, const uint8_t* packed, size_t count, float* out);
float orc_tq_dot_with_packed(size_t dim, int bits, const float* query, const uint8_t* packed, size_t count);
void orc_tq_rotate(const float* x, size_t dim, const float* signs, float* out);
void orc_tq_rotate_inverse(const float* x, size_t dim, const float* signs, float* out);
void orc_tq_compress(const float* x, size_t dim, int bits, const float* signs, uint8_t* packed);
void orc_tq_attention_head(const float* query, const uint8_t* k_codes, const uint8_t* v_codes, size_t kv_len, size_t dim, int bits,
                           const float* signs_k, const float* signs_v, float scale, float* out);
int orc_model_set_kv_turboquant(orc_model* m, int bits, const float* signs, size_t n_signs);
/* TurboQuantProd (`tq2-qjl | tq3-qjl`, src/model/turboquant/qjl.rs): the projector's Gaussian matrix S[dim][dim] is an input */
float orc_tq_dot_with_sign_bits(const float* values, const uint64_t* bits, size_t count);
void orc_tq_qjl_project(const float* S, size_t dim, const float* q, float* out);
void orc_tq_qjl_compress(const float* S, size_t dim, const float* x, uint64_t* bits, float* norm);
float orc_tq_qjl_inner_product_fast(size_t dim, const float* projected_query, const uint64_t* bits, float key_norm);
size_t orc_tq_bytes_per_entry(size_t dim, int bits, int use_qjl);
void orc_tq_compress_qjl(const float* x, size_t dim, int bits, const float* signs, const float* S, uint8_t* packed, uint64_t* qjl_bits,
                         float* residual_norm);
void orc_tq_attention_head_qjl(const float* query, const uint8_t* k_codes, const uint64_t* k_qjl, const float* k_norm, const uint8_t* v_codes,
                               size_t kv_len, size_t dim, int bits, const float* signs_k, const float* signs_v, const float* S_k, float scale,
                               float* out);
/* ... on a model: qjl = [layer][kv head][padded_dim][padded_dim], the K engines' matrices (NULL / 0: TurboQuantMSE) */
int orc_model_set_kv_turboquant_qjl(orc_model* m, const float* qjl, size_t n_qjl);
/* debug taps: hidden state after the last forward's final layer (pre-norm) */
int orc_model_last_hidden(const orc_model* m, float* out);

#ifdef __cplusplus
}
#endif
#endif
