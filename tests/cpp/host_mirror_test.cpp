// Drives the engine through the C++ host mirror (llama-gguf_amd/host/hip_gpu_inference.hpp) exactly as the
// reference's GpuModelWrapper would: from_model -> forward(tokens, ctx) -> logits.  Prints the logits of a
// 5-token prompt and 3 greedy continuations as hex floats; tests/test_gpu_model.py compares them to the oracle.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/llama_gguf_synth.h"
#include "../../llama-gguf_amd/host/hip_gpu_inference.hpp"

using namespace llama_gguf;

int main() {
  // the "test-dense" config of llama-gguf_amd/synth.py with a plain Q4_K mix
  const uint32_t H = 512, FFN = 1024, L = 2, NH = 8, NKV = 2, D = 64, V = 1024;
  const uint64_t seed = 0x9E3779B97F4A7C15ull;
  std::vector<std::vector<uint8_t>> store;
  std::vector<HostTensor> tensors;
  auto add = [&](const std::string& name, uint32_t type, uint64_t n0, uint64_t n1, int kind) {
    const uint64_t n = n0 * (n1 ? n1 : 1);
    store.emplace_back(lgs_tensor_nbytes(type, n));
    if (lgs_fill_tensor(name.c_str(), type, n, n0, kind, seed, store.back().data(), store.back().size(), 2)) std::exit(3);
    tensors.push_back(HostTensor{name, type, {n0, n1, 0, 0}, store.back().data(), store.back().size()});
  };
  const uint32_t Q4K = LGH_TYPE_Q4_K, F32 = LGH_TYPE_F32;
  add("token_embd.weight", Q4K, H, V, 0);
  for (uint32_t i = 0; i < L; i++) {
    const std::string p = "blk." + std::to_string(i) + ".";
    add(p + "attn_norm.weight", F32, H, 0, 1);
    add(p + "attn_q.weight", Q4K, H, NH * D, 0);
    add(p + "attn_k.weight", Q4K, H, NKV * D, 0);
    add(p + "attn_v.weight", Q4K, H, NKV * D, 0);
    add(p + "attn_output.weight", Q4K, NH * D, H, 0);
    add(p + "ffn_norm.weight", F32, H, 0, 1);
    add(p + "ffn_gate.weight", Q4K, H, FFN, 0);
    add(p + "ffn_up.weight", Q4K, H, FFN, 0);
    add(p + "ffn_down.weight", Q4K, FFN, H, 0);
  }
  add("output_norm.weight", F32, H, 0, 1);
  add("output.weight", Q4K, H, V, 0);
  lgh_model_desc d{};
  d.hidden_size = H; d.intermediate_size = FFN; d.num_layers = L; d.num_heads = NH; d.num_kv_heads = NKV; d.head_dim = D;
  d.vocab_size = V; d.norm_eps = 1e-5f; d.rope_freq_base = 10000.0f; d.rope_freq_scale = 1.0f; d.device_id = 0;
  try {
    GpuModelWrapper model(HipGpuInference::from_model(d, tensors, 32));
    InferenceContext ctx;
    std::vector<uint32_t> toks = {3, 17, 255, 9, 700};
    for (int step = 0; step < 4; step++) {
      std::vector<float> logits = model.forward(toks, ctx);
      uint32_t best = 0;
      for (uint32_t i = 1; i < V; i++) if (!(logits[i] < logits[best])) best = i;   // last maximum (main.rs:1815-1821)
      std::printf("step %d pos %zu argmax %u :", step, ctx.position, best);
      for (uint32_t i = 0; i < V; i++) std::printf(" %a", logits[i]);
      std::printf("\n");
      toks = {best};
    }
    try { model.gpu().forward(V); std::puts("ERR no exception"); return 4; }
    catch (const BackendError& e) { std::printf("error-variant %s\n", e.variant().c_str()); }
    // the device side of BatchedEngine through the mirror: two slots with different histories; slot 1 replays the single-sequence
    // run above token by token (slot 0 joins late, through the batched prompt path), so its logits equal the first line printed
    // above (to rounding: below 64 rows the single-sequence engine uses its one-launch attention, another summation tree)
    HipGpuInference& g = model.gpu();
    g.batch_create(2);
    for (uint32_t t : {3u, 17u, 255u, 9u}) g.forward_multi({1}, {t});
    g.batch_prefill(0, {5, 6});
    std::vector<uint32_t> nxt;
    std::vector<std::vector<float>> lm = g.forward_multi({1, 0}, {700, 7}, &nxt);
    std::printf("multi pos %zu %zu next %u %u :", g.batch_position(1), g.batch_position(0), nxt[0], nxt[1]);
    for (uint32_t i = 0; i < V; i++) std::printf(" %a", lm[0][i]);
    std::printf("\n");
    std::vector<std::vector<uint32_t>> dev = g.decode_greedy_multi({1, 0}, {nxt[0], nxt[1]}, 3);
    std::printf("multi-greedy %u %u %u\n", dev[0][0], dev[1][0], dev[2][0]);
  } catch (const BackendError& e) {
    std::fprintf(stderr, "BackendError %s\n", e.what());
    return 2;
  }
  return 0;
}
