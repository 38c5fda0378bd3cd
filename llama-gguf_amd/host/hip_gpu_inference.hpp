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
  static HipGpuInference from_model(lgh_model_desc desc, const std::vector<HostTensor>& tensors, size_t max_seq_len) {
    desc.struct_size = sizeof(lgh_model_desc);
    desc.max_seq_len = (uint32_t)max_seq_len;
    lgh_ctx* h = nullptr;
    int rc = lgh_create(&desc, &h);
    if (rc) throw BackendError(rc, "lgh_create");
    HipGpuInference g(h, desc.vocab_size);
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
