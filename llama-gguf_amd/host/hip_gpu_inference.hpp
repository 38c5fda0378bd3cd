// hip_gpu_inference.hpp — C++ host-side mirror of the reference's GPU-engine interface over the C ABI.
//
// The reference (Rust) drives GPU engines through `trait GpuInference` (src/backend/mod.rs:283-296), builds
// them with `from_model(model, max_seq_len)` (src/backend/cuda/gpu_only.rs:426) and adapts them to `Model`
// with `GpuModelWrapper` (src/backend/mod.rs:302-364).  No Rust toolchain exists in this image, so the Rust
// binding ships as source; this header is the same surface in C++17 — same names, argument
// meaning and error behaviour — for host programs that link libllama_gguf_hip.so directly.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/llama_gguf_hip.h"

namespace llama_gguf {

// BackendError (src/backend/error.rs:3-37): `variant()` names the Rust variant the status maps to.
class BackendError : public std::runtime_error {
 public:
  BackendError(int status, const std::string& msg) : std::runtime_error(variant_name(status) + ": " + msg), status_(status) {}
  int status() const { return status_; }
  std::string variant() const { return variant_name(status_); }
  static std::string variant_name(int s) {
    static const char* names[] = {"Ok", "NotAvailable", "ShapeMismatch", "DTypeMismatch", "UnsupportedDType", "Unsupported",
                                  "InvalidArgument", "Tensor", "InitializationFailed", "AllocationFailed", "OperationFailed"};
    return s >= 0 && s <= 10 ? names[s] : "Status" + std::to_string(s);
  }
 private:
  int status_;
};

// One tensor as LlamaModel::into_parts hands it over (src/model/llama.rs:138-160): GGUF name, GGML dims
// (dim 0 fastest), ggml type, host bytes in native GGUF order.
struct HostTensor {
  std::string name;
  uint32_t ggml_type;
  uint64_t ne[4];
  const void* data;
  size_t nbytes;
};

// impl GpuInference for HipGpuInference
class HipGpuInference {
 public:
  // pub fn from_model(model: LlamaModel, max_seq_len: usize) -> BackendResult<Self>
  // kv_rotation_signs / kv_qjl_matrices: what a TurboQuant KV cache (desc.kv_cache_type LGH_KV_TQ*; `--kv-cache-type tq2 | tq3 |
  // tq2-qjl | tq3-qjl`) draws from its seeded RNG in the reference — HadamardRotation::signs() of every (layer, kv head, K / V) engine
  // and the K engines' QjlProjector matrices (src/model/kv_turboquant.rs:44-71) — as data; empty = the library's stand-ins
  static HipGpuInference from_model(lgh_model_desc desc, const std::vector<HostTensor>& tensors, size_t max_seq_len,
                                    const std::vector<float>& kv_rotation_signs = {}, const std::vector<float>& kv_qjl_matrices = {}) {
    desc.struct_size = sizeof(lgh_model_desc);
    desc.max_seq_len = (uint32_t)max_seq_len;
    lgh_ctx* h = nullptr;
    int rc = lgh_create(&desc, &h);
    if (rc) throw BackendError(rc, "lgh_create");
    HipGpuInference g(h, desc.vocab_size);
    if (!kv_rotation_signs.empty()) g.check(lgh_set_kv_rotation_signs(h, kv_rotation_signs.data(), kv_rotation_signs.size()));
    if (!kv_qjl_matrices.empty()) g.check(lgh_set_kv_qjl_matrices(h, kv_qjl_matrices.data(), kv_qjl_matrices.size()));
    for (const HostTensor& t : tensors) g.check(lgh_upload_tensor(h, t.name.c_str(), t.ggml_type, t.ne, t.data, t.nbytes));
    g.check(lgh_finalize(h));
    return g;
  }
  HipGpuInference(HipGpuInference&& o) noexcept : h_(o.h_), vocab_(o.vocab_) { o.h_ = nullptr; }
  HipGpuInference& operator=(HipGpuInference&& o) noexcept {
    if (this != &o) { lgh_destroy(h_); h_ = o.h_; vocab_ = o.vocab_; o.h_ = nullptr; }
    return *this;
  }
  HipGpuInference(const HipGpuInference&) = delete;
  HipGpuInference& operator=(const HipGpuInference&) = delete;
  ~HipGpuInference() { lgh_destroy(h_); }  // impl Drop

  std::vector<float> forward(uint32_t token_id) {     // fn forward(&mut self, token_id: u32) -> BackendResult<Vec<f32>>
    std::vector<float> logits(vocab_);
    check(lgh_forward(h_, token_id, logits.data()));
    return logits;
  }
  void prefill_token(uint32_t token_id) { check(lgh_prefill_token(h_, token_id)); }
  // GpuOnlyInference::forward_batch minus the final forward (gpu_only.rs:776-790); batched f16-GEMM path when the model allows
  void forward_batch(const std::vector<uint32_t>& tokens) { check(lgh_prefill_batch(h_, tokens.data(), tokens.size())); }
  bool prefill_is_batched() { return lgh_prefill_is_batched(h_) != 0; }
  void kv_truncate(size_t new_len) { check(lgh_kv_truncate(h_, new_len)); }        // KVCache::truncate (model/mod.rs:130-134)
  void kv_shift_left(size_t amount) { check(lgh_kv_shift_left(h_, amount)); }      // KVCache::shift_left (model/mod.rs:142-172)
  void reset() { lgh_reset(h_); }                      // infallible, O(1)
  size_t position() const { return lgh_position(h_); }
  uint32_t forward_argmax(uint32_t token_id) {
    uint32_t next = 0;
    check(lgh_forward_argmax(h_, token_id, &next));
    return next;
  }
  lgh_ctx* handle() { return h_; }

  // ---- the device side of BatchedEngine (src/engine_batched.rs:23-194, 200-330, 355-400): a slot = one ActiveSequence's
  // InferenceContext; one step reads the weights once for all listed slots, every sequence gets the single-sequence logits bit for bit
  void batch_create(uint32_t max_batch) { check(lgh_batch_create(h_, max_batch)); }     // BatchedEngineConfig::max_batch_size
  void batch_reset(uint32_t slot) { check(lgh_batch_reset(h_, slot)); }                 // create_active_sequence: model.create_context()
  size_t batch_position(uint32_t slot) { return lgh_batch_position(h_, slot); }
  void batch_prefill(uint32_t slot, const std::vector<uint32_t>& tokens) { check(lgh_batch_prefill(h_, slot, tokens.data(), tokens.size())); }
  // step (engine_batched.rs:236-290): logits[i] for slots[i] fed tokens[i]; next_tokens (optional) = the bench's arg-max rule per sequence
  std::vector<std::vector<float>> forward_multi(const std::vector<uint32_t>& slots, const std::vector<uint32_t>& tokens,
                                                std::vector<uint32_t>* next_tokens = nullptr) {
    if (slots.size() != tokens.size()) throw std::invalid_argument("slots and tokens differ in length");
    std::vector<float> flat(slots.size() * (size_t)vocab_);
    if (next_tokens) next_tokens->assign(slots.size(), 0);
    check(lgh_forward_multi(h_, slots.data(), tokens.data(), (uint32_t)slots.size(), flat.data(), next_tokens ? next_tokens->data() : nullptr));
    std::vector<std::vector<float>> out(slots.size());
    for (size_t i = 0; i < slots.size(); i++) out[i].assign(flat.begin() + i * vocab_, flat.begin() + (i + 1) * vocab_);
    return out;
  }
  // n_steps greedy steps with the tokens fed back on the device: out[step][i]
  std::vector<std::vector<uint32_t>> decode_greedy_multi(const std::vector<uint32_t>& slots, const std::vector<uint32_t>& first_tokens, size_t n_steps) {
    if (slots.size() != first_tokens.size()) throw std::invalid_argument("slots and tokens differ in length");
    std::vector<uint32_t> flat(n_steps * slots.size());
    check(lgh_decode_greedy_multi(h_, slots.data(), first_tokens.data(), (uint32_t)slots.size(), n_steps, flat.data()));
    std::vector<std::vector<uint32_t>> out(n_steps);
    for (size_t s = 0; s < n_steps; s++) out[s].assign(flat.begin() + s * slots.size(), flat.begin() + (s + 1) * slots.size());
    return out;
  }

 private:
  HipGpuInference(lgh_ctx* h, uint32_t vocab) : h_(h), vocab_(vocab) {}
  void check(int rc) { if (rc) throw BackendError(rc, lgh_last_error(h_)); }
  lgh_ctx* h_;
  uint32_t vocab_;
};

struct InferenceContext { size_t position = 0; size_t kv_seq_len = 0; };

// GpuModelWrapper<T: GpuInference>::forward (src/backend/mod.rs:323-355)
class GpuModelWrapper {
 public:
  explicit GpuModelWrapper(HipGpuInference gpu) : gpu_(std::move(gpu)) {}
  std::vector<float> forward(const std::vector<uint32_t>& tokens, InferenceContext& ctx) {
    if (ctx.position == 0 && gpu_.position() > 0) gpu_.reset();
    if (tokens.empty()) throw std::invalid_argument("No tokens to process");
    for (size_t i = 0; i + 1 < tokens.size(); i++) gpu_.prefill_token(tokens[i]);
    std::vector<float> logits = gpu_.forward(tokens.back());
    ctx.position += tokens.size();
    ctx.kv_seq_len = ctx.position;
    return logits;
  }
  HipGpuInference& gpu() { return gpu_; }
 private:
  HipGpuInference gpu_;
};

}  // namespace llama_gguf
