/*
 * include/llama_gguf_synth.h — synthetic GGUF tensor payload generator (host only, C ABI).
 *
 * There are no model files in this environment and none may be fetched, so benchmarks and parity
 * tests run on random-init weights.  This generator writes VALID quantized blocks directly (it is
 * not a quantizer): a counter-based RNG keyed by the GGUF tensor name produces the packed integer
 * fields, and the f16 scales are chosen so that the dequantized weights are ~zero-mean with
 * std ~= 1/sqrt(in_features) (activations stay O(1) through 80 layers).  Byte layouts are the
 * reference's #[repr(C)] blocks (src/tensor/quant/blocks.rs:8-168) in native GGUF order: a weight
 * [in_features, out_features] is out_features consecutive rows of in_features/block_size blocks
 * (src/backend/cpu/ops.rs:1120-1122).
 *
 * The same bytes are handed to the HIP engine (lgh_upload_tensor) and to the CPU oracle, so both
 * sides compute on identical weights without any file being shipped.
 */
#ifndef LLAMA_GGUF_SYNTH_H
#define LLAMA_GGUF_SYNTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bytes needed for n_elems elements of ggml type `type` (0 if the type is unknown or n_elems is
 * not a multiple of the block size). */
size_t lgs_tensor_nbytes(uint32_t ggml_type, uint64_t n_elems);

/* Fill `out` (nbytes == lgs_tensor_nbytes) with the payload of tensor `name`.
 *   in_features : row length (GGML dim 0); sets the weight scale 1/sqrt(in_features)
 *   kind        : 0 = linear weight, 1 = norm weight (1 + 0.01*u, F32 only), 2 = bias (0.01*u, F32 only)
 *   seed        : global seed, mixed with fnv1a(name)
 *   threads     : worker threads (<=0: all cores)
 * Supported types: F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K.
 * Returns 0 on success, 1 on bad arguments. */
int lgs_fill_tensor(const char* name, uint32_t ggml_type, uint64_t n_elems, uint64_t in_features, int kind,
                    uint64_t seed, void* out, size_t nbytes, int threads);

#ifdef __cplusplus
}
#endif
#endif
