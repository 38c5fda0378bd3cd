// engine.h — the GPU-resident decode engine behind the lgh_* C ABI (internal).
#pragma once

#include <string>
#include <vector>

#include "common.h"
#include "ptok.h"

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
  bool moe() const { return router != nullptr; }
};

enum TokenMode { MODE_PREFILL = 0, MODE_FORWARD = 1, MODE_GREEDY = 2, MODE_COUNT = 3 };

// The XQ image (xq.h) of an f32 activation buffer, for int8-MFMA consumers.  `fresh` = the image matches the buffer's
// current contents; `tag` = the RMSNorm weights it was multiplied with (nullptr: none).
struct XqBuf { const float* f32 = nullptr; uint8_t* xq = nullptr; float* ssq = nullptr; uint32_t k = 0; bool fresh = false; const float* tag = nullptr; };

// one chained FFN launch (engine.hip: launch_ffn_chain): host copy of the descriptors + their device image
struct ChainSlot { MvChainHost host; uint8_t* dev = nullptr; bool prepared = false, uploaded = false; };

// scratch of the batched prompt path (prefill.hip), allocated at the first lgh_prefill_batch that uses it
struct PfScratch {
  bool ready = false;
  uint8_t *xh_h = nullptr, *xh_attn = nullptr, *xh_act = nullptr;   // XH activations: [128][hidden], [128][QD], [128][ffn]
  float *hidden = nullptr, *q = nullptr, *part = nullptr, *ssq = nullptr;
  size_t part_bytes = 0;
  int* tokens = nullptr;
  uint32_t* tok_pinned = nullptr;      // pinned host staging of the prompt's token ids, one slot per position
  hipEvent_t tok_copied = nullptr;     // recorded behind the last copy out of it
  // MoE layers: routing of the block's tokens, tokens grouped by expert, one expert's gathered input, per-slot expert outputs
  int *moe_sel = nullptr, *moe_cnt = nullptr, *moe_base = nullptr, *moe_list = nullptr, *moe_rowmap = nullptr, *moe_tokmap = nullptr;
  float* moe_w = nullptr;
  uint8_t *xh_gather = nullptr, *xh_act_e = nullptr;   // per expert: gathered inputs [E][xh_bytes(H)], activations [E][xh_bytes(EI)]
};

struct ProfRec { int cls; int sym; uint64_t bytes; hipEvent_t a, b; };

// The persistent token kernel's program for one graph mode (decode_persistent.hip): built once at finalize.
struct PtProg {
  bool built = false, usable = false;
  std::vector<PtHostOp> ops;
  uint8_t* dev = nullptr;        // device image: PtOp[n], MvLaunch[n_mv], PtAttn[n_attn]
  unsigned* sync = nullptr;      // epoch, error flag, hand-off counters (never reset: they only grow)
  PtProgram P{};
  uint32_t mask = 0;
  size_t lds = 0;
  uint64_t weight_bytes = 0;     // algorithmic bytes of the program's matrices + vectors (KV rows are added per position)
  const float* first_nw = nullptr;   // norm weights the program's first op expects its input XQ image multiplied with
  std::string why;               // why the program cannot be used (diagnostics)
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
  hipGraphExec_t graph[lgh::MODE_COUNT][3] = {};   // [mode][attention variant: 0 split + combine, 1 single launch (short context), 2 split + merge in wo]
  bool attn_direct = false;                        // variant of the token being enqueued (chosen by the host-side position)
  bool attn_merge = false;                         // ... 8 splits, merged by the output projection's waves (no combine kernel)
  uint32_t merge_attn_max_kv = 0, merge_splits = 8;
  uint32_t direct_attn_max_kv = 0;                 // contexts up to this many rows take the single-launch attention
  uint64_t graph_nodes = 0;
  // accounting
  lgh_stats stats{};
  std::vector<lgh::ProfRec> prof;
  std::vector<void*> allocs;  // everything hipMalloc'ed by this context
  std::vector<lgh::XqBuf> xqs;
  std::vector<lgh::ChainSlot> chains;          // [graph mode][layer]
  unsigned* attn_arrive = nullptr;             // arrival counters of the split attention that merges itself (16 words per kv head)
  bool attn_fuse = false;
  std::vector<lgh::ChainSlot> flows;           // [graph mode][layer]: flow launches (wo | gate-up | down in one launch, hand-off counters)
  bool flow_mode = false;
  std::vector<lgh::ChainSlot*> chain_pending;  // descriptor uploads deferred past a stream capture
  unsigned* chain_sync = nullptr;              // grid-barrier words of the chained launches
  // flag-ordered graphs (handoff.h): two consecutive mat-vec launches of a layer run side by side on two streams, ordered by
  // hand-off counters instead of a kernel boundary, so that the second one's weight tiles are in flight while the first finishes
  bool flag_mode = false;                      // the context's graphs are captured that way
  bool flagging = false;                       // ... and one is being captured right now
  unsigned* flag_sync = nullptr;               // error word + counters
  std::vector<uint32_t> flag_edges;            // per owned layer: overlapped edges (1: wo->gate-up, 2: gate-up->down, 4: down->next QKV)
  std::vector<uint32_t> flag_cnt;              // per owned layer: first counter of wo's / gate-up's / down's output records (3 per layer)
  hipStream_t stream2 = nullptr;               // the other stream of a flag-ordered capture
  std::vector<hipEvent_t> flag_events;         // fork / join markers of the capture
  size_t flag_ev_next = 0;
  hipStream_t flag_origin = nullptr;           // the stream the token being enqueued started on
  uint32_t flag_qkv_wait = lgh::kFlagNone;     // counters the next fused QKV launch waits for (the previous layer's down projection)
  lgh::PfScratch pf;
  lgh::PtProg pt[lgh::MODE_COUNT];             // persistent token kernel, per graph mode
  float* pt_part = nullptr;                    // attention split partials of the persistent kernel
  uint32_t pt_s_max = 0, pt_rows_per_split = 64;
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
  // the input vector is the attention output, merged from split partials inside the kernel (MvLaunch::attn_*)
  const float* attn_ml = nullptr;
  const float* attn_acc = nullptr;
  uint32_t attn_splits = 0;
  // flag-ordered launch (MvLaunch::flag_*): first counters of the input's / the output's XQ records
  uint32_t flag_wait_first = lgh::kFlagNone, flag_sig_first = lgh::kFlagNone;
};

int fail(lgh_ctx* c, int status, const std::string& msg);
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

