// pipeline.hip — the layer pipeline INSIDE the library: one process, N stage contexts on N devices (or on one), the hidden
// vector hopped device to device over xGMI and the greedy token fed back the same way.  Host code only (no kernels here).
//
// Replaces, for a Rust host that links this library, the reference's pipeline executor: `PipelineExecutor::forward`
// (src/distributed/pipeline.rs:50-96) walks the shards in order and carries the activation — and the sampled token —
// through host memory and gRPC; `ShardServer::forward` (src/distributed/shard.rs:377-445) runs a shard's layers.  Here a
// stage is an `lgh_ctx` that owns layers [begin, end) (lgh_model_desc::layer_begin / layer_end: the embedding on the first,
// final norm + output projection on the last, src/distributed/model.rs:22-27), and per token and stage boundary ONE
// `hipMemcpyPeerAsync` of f32[hidden] on the producing stage's stream + an event the consuming stage's stream waits for.
// The host only enqueues: no host value crosses a stage boundary per token, and `lgh_pipeline_decode_greedy` synchronises
// once, at the end.  Stages that share a device share ONE stream and hop inside the producing stage's graph (no events, no runtime copies).
// (The multi-process form of the same pipeline — one rank per GPU, RCCL send/recv — is
// llama-gguf_amd/pipeline.py; both drive the same stage entry points.)
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/llama_gguf_hip.h"

struct lgh_pipeline {
  std::vector<lgh_ctx*> stage;
  std::vector<int> device;
  std::vector<hipEvent_t> handed;       // stage i's output has been copied into stage i + 1's hidden buffer
  std::vector<hipEvent_t> consumed;     // stage i + 1 is done with what stage i handed it (its own token has left its hidden buffer)
  std::vector<bool> consumed_set;
  hipEvent_t token_back = nullptr;      // the last stage's arg-max has been copied into the first stage's token word
  std::vector<void*> hidden;            // each stage's in / out residual-stream buffer
  void* token_in = nullptr;             // first stage
  void* argmax_out = nullptr;           // last stage
  uint32_t hidden_size = 0, vocab = 0, max_seq = 0;
  size_t pos = 0;
  std::string err;
};

namespace {

int pfail(lgh_pipeline* p, int status, const std::string& msg) {
  if (p) p->err = msg;
  return status;
}

int stage_fail(lgh_pipeline* p, size_t s, int status) {
  return pfail(p, status, "stage " + std::to_string(s) + ": " + lgh_last_error(p->stage[s]));
}

#define PIPE_HIP(p, expr)                                                                             \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess) return pfail((p), LGH_OPERATION_FAILED, std::string(#expr) + ": " + hipGetErrorString(e__)); \
  } while (0)

// contiguous, near-equal layer ranges; earlier stages take the remainder (pipeline.py: split_layers)
void split_layers(uint32_t n_layers, int n_stages, int s, uint32_t* lo, uint32_t* hi) {
  const uint32_t base = n_layers / (uint32_t)n_stages, rem = n_layers % (uint32_t)n_stages;
  *lo = (uint32_t)s * base + ((uint32_t)s < rem ? (uint32_t)s : rem);
  *hi = *lo + base + ((uint32_t)s < rem ? 1u : 0u);
}

hipStream_t stream_of(lgh_ctx* c) { return (hipStream_t)lgh_get_stream(c); }

int copy_words(lgh_pipeline* p, void* dst, int dst_dev, const void* src, int src_dev, uint32_t n, hipStream_t st) {
  PIPE_HIP(p, hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, (size_t)n * 4, st));
  return LGH_OK;
}

// stage s has produced its output on its stream: hand it to stage s + 1.  A stage's hidden buffer is its residual stream for
// the whole token, so the copy of token t + 1 into it waits until stage s + 1 has finished token t (without token feedback — a
// prompt — nothing else holds an early stage back).
// `fed_back`: the token being enqueued was produced by the previous token's LAST stage (greedy decode): the first stage cannot
// start it before every stage has finished the previous one, so the back-pressure wait is implied and skipped — a wait across
// two streams costs tens of microseconds of GPU idle time (measured: 2 stages on one GPU 1.90 vs 1.63 ms per token with three
// cross-stream waits per token; r03 DESIGN.md §6).  Stages that share a stream (same device) need no events at all.
int hand_over(lgh_pipeline* p, size_t s, bool fed_back) {
  PIPE_HIP(p, hipSetDevice(p->device[s]));
  hipStream_t st = stream_of(p->stage[s]), nx = stream_of(p->stage[s + 1]);
  if (p->consumed_set[s] && !fed_back && st != nx) PIPE_HIP(p, hipStreamWaitEvent(st, p->consumed[s], 0));
  if (st == nx) return LGH_OK;   // same device: the hop is a node of stage s's graph (lgh_stage_set_forward_targets)
  if (int rc = copy_words(p, p->hidden[s + 1], p->device[s + 1], p->hidden[s], p->device[s], p->hidden_size, st)) return rc;
  PIPE_HIP(p, hipEventRecord(p->handed[s], st));
  PIPE_HIP(p, hipSetDevice(p->device[s + 1]));
  PIPE_HIP(p, hipStreamWaitEvent(nx, p->handed[s], 0));
  return LGH_OK;
}

// One token through all stages, nothing synchronised.  mode_last: what the last stage does (0 layers only, 1 + logits, 2 + arg-max).
int run_token(lgh_pipeline* p, int mode_last, bool fed_back = false) {
  const size_t n = p->stage.size();
  for (size_t s = 0; s < n; s++) {
    int rc = lgh_stage_step(p->stage[s], s + 1 == n ? mode_last : 0);
    if (rc) return stage_fail(p, s, rc);
    if (s + 1 < n && (rc = hand_over(p, s, fed_back))) return rc;
    if (s > 0 && stream_of(p->stage[s]) != stream_of(p->stage[s - 1])) {
      // this stage's token is out of its hidden buffer (copied on, or — the last stage — fully processed)
      PIPE_HIP(p, hipSetDevice(p->device[s]));
      PIPE_HIP(p, hipEventRecord(p->consumed[s - 1], stream_of(p->stage[s])));
      p->consumed_set[s - 1] = true;
    }
  }
  p->pos += 1;
  return LGH_OK;
}

int set_token(lgh_pipeline* p, uint32_t token) {
  if (token >= p->vocab) return pfail(p, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");
  PIPE_HIP(p, hipSetDevice(p->device[0]));
  PIPE_HIP(p, hipMemsetD32Async((hipDeviceptr_t)p->token_in, (int)token, 1, stream_of(p->stage[0])));
  return LGH_OK;
}

int sync_all(lgh_pipeline* p) {
  for (size_t s = 0; s < p->stage.size(); s++) {
    const int rc = lgh_synchronize(p->stage[s]);
    if (rc) return stage_fail(p, s, rc);
  }
  return LGH_OK;
}

}  // namespace

extern "C" {

int lgh_pipeline_create(const lgh_model_desc* desc, const int* device_ids, int n_stages, lgh_pipeline** out) {
  if (!desc || !out || n_stages < 1 || desc->struct_size != sizeof(lgh_model_desc)) return LGH_INVALID_ARGUMENT;   // (lgh_create re-checks per stage)
  *out = nullptr;
  if ((uint32_t)n_stages > desc->num_layers) return LGH_INVALID_ARGUMENT;
  lgh_pipeline* p = new lgh_pipeline();
  p->hidden_size = desc->hidden_size;
  p->vocab = desc->vocab_size;
  p->max_seq = desc->max_seq_len;
  for (int s = 0; s < n_stages; s++) {
    lgh_model_desc d = *desc;
    d.device_id = device_ids ? device_ids[s] : desc->device_id;
    split_layers(desc->num_layers, n_stages, s, &d.layer_begin, &d.layer_end);
    lgh_ctx* c = nullptr;
    const int rc = lgh_create(&d, &c);
    if (rc) {
      for (lgh_ctx* q : p->stage) lgh_destroy(q);
      delete p;
      return rc;
    }
    p->stage.push_back(c);
    p->device.push_back(d.device_id);
  }
  *out = p;
  return LGH_OK;
}

int lgh_pipeline_upload_tensor(lgh_pipeline* p, const char* gguf_name, uint32_t ggml_type, const uint64_t ne[4], const void* host_bytes,
                               size_t nbytes) {
  if (!p) return LGH_INVALID_ARGUMENT;
  for (size_t s = 0; s < p->stage.size(); s++) {   // every stage sees every tensor and keeps what it owns
    const int rc = lgh_upload_tensor(p->stage[s], gguf_name, ggml_type, ne, host_bytes, nbytes);
    if (rc) return stage_fail(p, s, rc);
  }
  return LGH_OK;
}

int lgh_pipeline_finalize(lgh_pipeline* p) {
  if (!p) return LGH_INVALID_ARGUMENT;
  const size_t n = p->stage.size();
  p->hidden.assign(n, nullptr);
  for (size_t s = 0; s < n; s++) {
    int rc = lgh_finalize(p->stage[s]);
    if (rc) return stage_fail(p, s, rc);
    if ((rc = lgh_stage_hidden_buffer(p->stage[s], &p->hidden[s]))) return stage_fail(p, s, rc);
  }
  // stages on one device run on ONE stream (in order, no events between them); the first such stage's stream is the device's
  for (size_t s = 1; s < n; s++)
    for (size_t t = 0; t < s; t++)
      if (p->device[t] == p->device[s]) {
        const int rs = lgh_set_stream(p->stage[s], lgh_get_stream(p->stage[t]));
        if (rs) return stage_fail(p, s, rs);
        break;
      }
  int rc = lgh_stage_io_buffers(p->stage[0], &p->token_in, nullptr);
  if (rc) return stage_fail(p, 0, rc);
  if ((rc = lgh_stage_io_buffers(p->stage[n - 1], nullptr, &p->argmax_out))) return stage_fail(p, n - 1, rc);
  // ... and hop inside the producing stage's graph
  for (size_t s = 0; s < n && n > 1; s++) {
    void* hd = s + 1 < n && p->device[s] == p->device[s + 1] ? p->hidden[s + 1] : nullptr;
    void* td = s + 1 == n && p->device[s] == p->device[0] ? p->token_in : nullptr;
    if ((hd || td) && (rc = lgh_stage_set_forward_targets(p->stage[s], hd, td))) return stage_fail(p, s, rc);
  }
  // direct peer copies between neighbouring stages (and last -> first for the token); "already enabled" is fine
  for (size_t s = 0; s < n; s++) {
    const int a = p->device[s], b = p->device[(s + 1) % n];
    if (a == b) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
      PIPE_HIP(p, hipSetDevice(a));
      const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return pfail(p, LGH_INITIALIZATION_FAILED, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
      (void)hipGetLastError();
    }
  }
  p->handed.assign(n > 1 ? n - 1 : 0, nullptr);
  p->consumed.assign(n > 1 ? n - 1 : 0, nullptr);
  p->consumed_set.assign(n > 1 ? n - 1 : 0, false);
  for (size_t s = 0; s + 1 < n; s++) {
    PIPE_HIP(p, hipSetDevice(p->device[s]));
    PIPE_HIP(p, hipEventCreateWithFlags(&p->handed[s], hipEventDisableTiming));
    PIPE_HIP(p, hipSetDevice(p->device[s + 1]));
    PIPE_HIP(p, hipEventCreateWithFlags(&p->consumed[s], hipEventDisableTiming));
  }
  PIPE_HIP(p, hipSetDevice(p->device[n - 1]));
  PIPE_HIP(p, hipEventCreateWithFlags(&p->token_back, hipEventDisableTiming));
  p->pos = 0;
  return LGH_OK;
}

void lgh_pipeline_destroy(lgh_pipeline* p) {
  if (!p) return;
  for (size_t s = 0; s < p->stage.size(); s++) (void)lgh_synchronize(p->stage[s]);
  for (hipEvent_t e : p->handed)
    if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->consumed)
    if (e) (void)hipEventDestroy(e);
  if (p->token_back) (void)hipEventDestroy(p->token_back);
  for (lgh_ctx* c : p->stage) lgh_destroy(c);
  delete p;
}

int lgh_pipeline_forward(lgh_pipeline* p, uint32_t token, float* logits_out) {
  if (!p || p->hidden.empty()) return LGH_INVALID_ARGUMENT;
  if (!logits_out) return pfail(p, LGH_INVALID_ARGUMENT, "logits_out is NULL");
  if (p->pos >= p->max_seq) return pfail(p, LGH_INVALID_ARGUMENT, "position " + std::to_string(p->pos) + " >= max_seq_len " + std::to_string(p->max_seq));
  int rc = set_token(p, token);
  if (rc || (rc = run_token(p, 1))) return rc;
  const size_t last = p->stage.size() - 1;
  // the logits live in the last stage: lgh_stage_forward's copy path without running anything is not exposed, so read them
  // through the stage's own entry point for logits (a device-to-host copy on its stream + one synchronisation)
  if ((rc = lgh_stage_read_logits(p->stage[last], logits_out))) return stage_fail(p, last, rc);
  return LGH_OK;
}

int lgh_pipeline_prefill_token(lgh_pipeline* p, uint32_t token) {
  if (!p || p->hidden.empty()) return LGH_INVALID_ARGUMENT;
  if (p->pos >= p->max_seq) return pfail(p, LGH_INVALID_ARGUMENT, "position " + std::to_string(p->pos) + " >= max_seq_len " + std::to_string(p->max_seq));
  int rc = set_token(p, token);
  if (rc) return rc;
  return run_token(p, 0);
}

int lgh_pipeline_decode_greedy(lgh_pipeline* p, uint32_t first_token, size_t n_steps, uint32_t* tokens_out) {
  if (!p || p->hidden.empty()) return LGH_INVALID_ARGUMENT;
  if (n_steps && !tokens_out) return pfail(p, LGH_INVALID_ARGUMENT, "tokens_out is NULL");
  if (p->pos + n_steps > p->max_seq) return pfail(p, LGH_INVALID_ARGUMENT, "decode would exceed max_seq_len");
  const size_t n = p->stage.size(), pos0 = p->pos;
  int rc = set_token(p, first_token);
  if (rc) return rc;
  const auto t_host0 = std::chrono::steady_clock::now();
  for (size_t i = 0; i < n_steps; i++) {
    if ((rc = run_token(p, 2, i > 0))) return rc;
    if (n > 1) {   // the arg-max word -> the first stage's token word, device to device, behind the last stage's kernels
      PIPE_HIP(p, hipSetDevice(p->device[n - 1]));
      hipStream_t st = stream_of(p->stage[n - 1]), s0 = stream_of(p->stage[0]);
      if (st != s0) {   // (same stream: the last stage's graph ends with the copy)
        if ((rc = copy_words(p, p->token_in, p->device[0], p->argmax_out, p->device[n - 1], 1, st))) return rc;
        PIPE_HIP(p, hipEventRecord(p->token_back, st));
        PIPE_HIP(p, hipSetDevice(p->device[0]));
        PIPE_HIP(p, hipStreamWaitEvent(s0, p->token_back, 0));
      }
    }
  }
  if (std::getenv("LGH_DEBUG_PIPE") && n_steps)   // how long the HOST took to enqueue the steps (nothing synchronised yet)
    std::fprintf(stderr, "[lgh] pipeline: %zu steps x %zu stages enqueued in %.1f us per step\n", n_steps, n,
                 std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_host0).count() / (double)n_steps);
  if ((rc = lgh_stage_read_tokens(p->stage[n - 1], pos0, n_steps, tokens_out))) return stage_fail(p, n - 1, rc);
  return sync_all(p);
}

void lgh_pipeline_reset(lgh_pipeline* p) {
  if (!p) return;
  for (lgh_ctx* c : p->stage) lgh_reset(c);
  p->pos = 0;
}

// KVCache::truncate on every stage (src/model/mod.rs: the positions from new_len on are forgotten)
int lgh_pipeline_kv_truncate(lgh_pipeline* p, size_t new_len) {
  if (!p) return LGH_INVALID_ARGUMENT;
  if (new_len > p->pos) return pfail(p, LGH_INVALID_ARGUMENT, "new_len exceeds the current position");
  for (size_t s = 0; s < p->stage.size(); s++) {
    const int rc = lgh_kv_truncate(p->stage[s], new_len);
    if (rc) return stage_fail(p, s, rc);
  }
  p->pos = new_len;
  return LGH_OK;
}

size_t lgh_pipeline_position(const lgh_pipeline* p) { return p ? p->pos : 0; }

int lgh_pipeline_stages(const lgh_pipeline* p) { return p ? (int)p->stage.size() : 0; }

const char* lgh_pipeline_last_error(const lgh_pipeline* p) { return p ? p->err.c_str() : "null pipeline"; }

}  // extern "C"
