// engine.h — the GPU-resident decode engine behind the lgh_* C ABI (internal).
#pragma once

#include <string>
#include <vector>

#include "common.h"

namespace lgh {

struct LayerW {
  bool owned = false;  // layer belongs to this pipeline stage
  DevWeight wq, wk, wv, wo, gate, up, down;           // dense
  DevWeight gate_exps, up_exps, down_exps;            // MoE stacks [in, out, n_expert]
  float* attn_norm = nullptr;
  float* ffn_norm = nullptr;
  float* router = nullptr;                            // [n_experts][hidden] f32
  float* bq = nullptr; float* bk = nullptr; float* bv = nullptr; float* bo = nullptr;
  float* kcache = nullptr;
  float* vcache = nullptr;
  // LGH_FLAG_KV_INT8: int8 rows [kv_head][max_seq][head_dim] + one f32 scale per (kv_head, position) instead of the f32 caches
  int8_t *k8 = nullptr, *v8 = nullptr;
  float *kscale = nullptr, *vscale = nullptr;
  uint32_t* kx = nullptr;   // TurboQuantProd: QJL rows of the K cache, [kv head][max_seq][head_dim / 32 + 1] words (sign bits, residual norm)
  bool moe() const { return router != nullptr; }
};

enum TokenMode { MODE_PREFILL = 0, MODE_FORWARD = 1, MODE_GREEDY = 2, MODE_COUNT = 3 };

// The XQ image (xq.h) of an f32 activation buffer, for int8-MFMA consumers.  `fresh` = the image matches the buffer's
// current contents; `tag` = the RMSNorm weights it was multiplied with (nullptr: none).
struct XqBuf { const float* f32 = nullptr; uint8_t* xq = nullptr; float* ssq = nullptr; uint32_t k = 0; bool fresh = false; const float* tag = nullptr; };

// scratch of the batched prompt path (prefill.hip), allocated at the first lgh_prefill_batch that uses it
struct PfScratch {
  bool ready = false;
  uint8_t *xh_h = nullptr, *xh_attn = nullptr, *xh_act = nullptr;   // XH activations: [128][hidden], [128][QD], [128][ffn]
  float *hidden = nullptr, *q = nullptr, *part = nullptr, *ssq = nullptr;
  size_t part_bytes = 0;
  int* tokens = nullptr;
  uint32_t* tok_pinned = nullptr;      // pinned host staging of the prompt's token ids, one slot per position
  hipEvent_t tok_copied = nullptr;     // recorded behind the last copy out of it
  uint32_t tok_lo = 0, tok_hi = 0;     // slots [tok_lo, tok_hi) may still be read by a copy enqueued since the last wait for tok_copied
  // MoE layers: routing of the block's tokens, tokens grouped by expert, one expert's gathered input, per-slot expert outputs
  int *moe_sel = nullptr, *moe_cnt = nullptr, *moe_base = nullptr, *moe_list = nullptr, *moe_rowmap = nullptr, *moe_tokmap = nullptr;
  float* moe_w = nullptr;
  uint8_t *xh_gather = nullptr, *xh_act_e = nullptr;   // per expert: gathered inputs [E][xh_bytes(H)], activations [E][xh_bytes(EI)]
};

struct ProfRec { int cls; int sym; uint64_t bytes; hipEvent_t a, b; };

// multi-sequence decode (engine_batch.hip): `max_batch` slots, each one sequence's KV caches and position, and the vectors of
// up to max_batch sequences side by side ([sequence][...], sequence s of a step at s times the vector's length)
struct BatchScratch {
  bool ready = false;
  uint32_t max_batch = 0, ffn = 0;
  uint64_t cache_stride = 0;                          // floats between two slots' caches of one layer
  float *hidden = nullptr, *xnorm = nullptr, *q = nullptr, *attn_out = nullptr, *act = nullptr, *act2 = nullptr, *logits = nullptr,
        *part_ml = nullptr, *part_acc = nullptr, *amax_v = nullptr, *moe_w = nullptr, *mv_part = nullptr;
  uint64_t mv_part_floats = 0;
  int *amax_i = nullptr, *moe_sel = nullptr;
  int *d_tokens = nullptr, *d_pos = nullptr, *d_slot = nullptr, *d_log = nullptr;   // the step's control words (device), the greedy token log
  int* h_ctl = nullptr;                               // pinned staging of the control words
  std::vector<float*> kcache, vcache;                 // per layer: [slot][kv_head][max_seq][head_dim]
  // TurboQuant caches (kv_cache_type LGH_KV_TQ*): per layer the slots' code rows [slot][kv_head][max_seq][row bytes] (K, V) and, with
  // QJL, the K rows' sign bits + norms [slot][kv_head][max_seq][head_dim / 32 + 1]; kv_tmp = the step's rotated K / V rows, f32,
  // [sequence][K row | V row]
  std::vector<uint8_t*> kq, vq;
  std::vector<uint32_t*> kx;
  float* kv_tmp = nullptr;
  uint64_t code_stride = 0, x_stride = 0;             // bytes / words between two slots
  // MoE layers, experts read once per step (top-k <= 2): the (sequence, slot) pairs' activations [pair][ffn] (+ XQ views) and expert
  // outputs [pair][hidden]; per expert the number of pairs that chose it and their list
  float *moe_act = nullptr, *moe_tmp = nullptr;
  int *moe_cnt = nullptr, *moe_idx = nullptr;
  std::vector<size_t> pos;                            // per slot: tokens in its cache
  hipGraphExec_t graph[kMaxBatch + 1][2] = {};        // [n_seq][0 logits only, 1 + arg-max fed back]
};

}  // namespace lgh

struct lgh_ctx {
  lgh_model_desc d{};
  int device = 0;
  uint32_t l0 = 0, l1 = 0;
  bool first = true, last = true;
  hipStream_t own_stream = nullptr, stream = nullptr;
  std::vector<lgh::LayerW> layers;
  // embedding table in native GGUF layout (row-dequantized per token) and the output projection
  uint8_t* embd_raw = nullptr;
  int embd_type = -1;
  size_t embd_bytes = 0;
  std::vector<uint8_t> embd_host;  // kept only until finalize, for a tied output projection
  lgh::DevWeight output;
  float* output_norm = nullptr;
  // scratch (device)
  void *fwd_hidden = nullptr, *fwd_token = nullptr;   // in-graph hops to a same-device stage (lgh_stage_set_forward_targets)
  float *hidden = nullptr, *xnorm = nullptr, *q = nullptr, *kv_tmp = nullptr, *attn_out = nullptr, *act = nullptr,
        *act2 = nullptr, *logits = nullptr, *part_ml = nullptr, *part_acc = nullptr, *rope_cs = nullptr, *moe_w = nullptr,
        *amax_v = nullptr;
  int *moe_sel = nullptr, *state = nullptr, *amax_i = nullptr, *tok_log = nullptr;
  uint32_t n_splits = 8;
  // host mirrors
  size_t pos = 0;
  bool finalized = false;
  bool profiling = false;
  std::string err;
  hipGraphExec_t graph[lgh::MODE_COUNT][2] = {};   // [mode][attention variant: 0 split + combine, 1 single launch (short context)]
  bool attn_direct = false;                        // variant of the token being enqueued (chosen by the host-side position)
  uint32_t direct_attn_max_kv = 0;                 // contexts up to this many rows take the single-launch attention
  uint64_t graph_nodes = 0;
  // accounting
  lgh_stats stats{};
  std::vector<lgh::ProfRec> prof;
  std::vector<void*> allocs;  // everything hipMalloc'ed by this context
  std::vector<lgh::XqBuf> xqs;
  lgh::PfScratch pf;
  lgh::BatchScratch batch;
  float* tq_signs = nullptr;                   // TurboQuant KV cache: [owned layer][kv head][k, v][head_dim] rotation signs (device)
  std::vector<float> tq_signs_host;
  float* tq_qjl = nullptr;                     // TurboQuantProd: [owned layer][kv head][head_dim][head_dim] QJL projection matrices (device)
  std::vector<float> tq_qjl_host;
  float* kv_shift_tmp = nullptr;               // scratch of lgh_kv_shift_left (one cache tensor), allocated at first use
};

// ---- helpers shared by engine.hip and ops_api.hip ----
#include <cstring>

struct LayoutInfo {
  int dev_type;
  int nplanes;
  uint32_t bpb[4];      // bytes per block, per plane
  uint32_t belems;      // elements per block
};

struct SegSpec {
  int npass = 1;
  // also leave `out` as an XQ image for the next (int8-MFMA) consumer: 0 no, 1 plain, 2 multiplied by xq_next_nw (+ sum of squares)
  int xq_next = 0;
  const float* xq_next_nw = nullptr;
  const lgh::DevWeight* W[4] = {nullptr, nullptr, nullptr, nullptr};
  const float* x[4] = {nullptr, nullptr, nullptr, nullptr};
  const int* sel[4] = {nullptr, nullptr, nullptr, nullptr};
  int epi = lgh::EPI_STORE;
  float* out = nullptr;
  float* out2 = nullptr;
  const float* resid = nullptr;
  const float* bias = nullptr;
  const float* moe_w = nullptr;
};

// the vectors the FFN half of a layer works on (one sequence's)
struct FfnView { float* hidden; float* act; float* act2; float* xnorm; int* moe_sel; float* moe_w; };

static inline bool kv_is_tq(uint32_t t) { return t == LGH_KV_TQ2 || t == LGH_KV_TQ3 || t == LGH_KV_TQ2_QJL || t == LGH_KV_TQ3_QJL; }
static inline bool kv_is_qjl(uint32_t t) { return t == LGH_KV_TQ2_QJL || t == LGH_KV_TQ3_QJL; }
static inline int kv_tq_bits(uint32_t t) { return t == LGH_KV_TQ2 || t == LGH_KV_TQ2_QJL ? 2 : 3; }
int fail(lgh_ctx* c, int status, const std::string& msg);
int build_mv_group(lgh_ctx* c, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k, bool mfma, lgh::MvLaunch& L, uint32_t& wg,
                   uint32_t& threads, uint64_t& alg, uint32_t tile_cap);
int ffn_forward(lgh_ctx* c, lgh::LayerW& Lw, const FfnView& v, const float* next_nw, bool next_mfma);
int engine_shape_check(const lgh_model_desc& d, std::string& why);   // LGH_OK, or the status lgh_create returns and why
int dev_alloc(lgh_ctx* c, void** p, size_t bytes);
LayoutInfo layout_for(int src_type);
bool fused_type(int t);
int upload_matrix(lgh_ctx* c, lgh::DevWeight& W, int src_type, uint32_t k, uint32_t n, uint32_t n_stack, int slot,
                  const void* host, size_t nbytes);
int upload_f32(lgh_ctx* c, float** dst, int src_type, uint64_t n, const void* host, size_t nbytes);
int launch_mv(lgh_ctx* c, int cls, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k);
lgh::XqBuf* xq_get(lgh_ctx* c, const float* f32, uint32_t k);   // finds or registers (allocating) the image of a buffer
void xq_stale(lgh_ctx* c, const float* f32);
int linear_any(lgh_ctx* c, int cls, const lgh::DevWeight& W, const float* x, float* out, const float* norm_w,
               const float* resid, const float* bias, int xq_next = 0, const float* xq_next_nw = nullptr);
int drain_prof(lgh_ctx* c);

// ------------------------------------------------------------------------------------------------
// launch recording (profiling mode: hipEvent pair per launch, on the launch stream)
// ------------------------------------------------------------------------------------------------
template <class F>
inline int run_k(lgh_ctx* c, int cls, int sym, uint64_t alg_bytes, F&& f) {
  lgh::ProfRec rec{cls, sym, alg_bytes, nullptr, nullptr};
  if (c->profiling) {
    if (hipEventCreate(&rec.a) != hipSuccess || hipEventCreate(&rec.b) != hipSuccess) return fail(c, LGH_OPERATION_FAILED, "hipEventCreate");
    (void)hipEventRecord(rec.a, c->stream);
  }
  hipError_t e = f();
  if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("kernel launch (class ") + std::to_string(cls) + "): " + hipGetErrorString(e));
  if (c->profiling) {
    (void)hipEventRecord(rec.b, c->stream);
    c->prof.push_back(rec);
  }
  return LGH_OK;
}

