// engine_batch.hip — multi-sequence decode on one GPU: the device side of the reference's BatchedEngine
// (src/engine_batched.rs:23-194 config / request types, 200-330 the loop, 355-400 `step_sequence`).
//
// The reference keeps one InferenceContext (KV cache + position) per ActiveSequence and, every iteration, runs
// `model.forward` for each active sequence in turn: B sequences read the weights B times.  Here a context owns `max_batch`
// SLOTS — a slot = one sequence's KV caches + position — and lgh_forward_multi feeds one token to each listed slot in ONE pass
// over the weights: every quantized mat-vec launch reads its tiles once and multiplies them with all n_seq input vectors
// (matvec_batch.hip), attention / its merge / embedding / arg-max run with the sequence as the grid's second dimension (over f32 or
// TurboQuant slots); MoE layers route every sequence to its own experts: from 3 sequences on (moe_group_min) the step's (sequence,
// slot) pairs are grouped by expert and each selected expert is read once — all experts of a layer in one gate | up and one down
// launch, the expert as the grid's third dimension — below that the FFN runs sequence by sequence.  Every sequence's logits are
// bit-identical to what the single-sequence engine computes for the same tokens (tests/test_gpu_batch.py): same kernels'
// arithmetic, same summation orders.
//
// A step is a constant hipGraph per n_seq: tokens, positions and slot numbers live in device words (d_tokens / d_pos /
// d_slot), the arg-max feeds the next token back on the device and `batch_advance` moves the positions, so
// lgh_decode_greedy_multi replays the graph with no host work in between.
#include "engine.h"
#include "xq.h"

#include <algorithm>
#include <cstdlib>
#include <cmath>

using namespace lgh;

#define HIP_TRYB(c, status, expr)                                                                 \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) return fail((c), (status), std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

namespace lgh {

__global__ void batch_advance_kernel(int* pos, int n) {
  if ((int)threadIdx.x < n) pos[threadIdx.x] += 1;
}

}  // namespace lgh

namespace {

uint32_t xq_stride_of(uint32_t k) { return (uint32_t)xq_bytes(k); }
uint32_t ssq_stride_of(uint32_t k) { return k / 16 + 64; }

// sequence-0 view of a batch vector is registered in the context's XQ registry under its f32 address (engine.hip: xq_get), the
// other sequences' views follow at the strides above: build_mv_group finds the images it needs there
int register_views(lgh_ctx* c, float* f32, uint8_t* xq, float* ssq, uint32_t k, uint32_t n) {
  for (uint32_t s = 0; s < n; s++) {
    XqBuf q;
    q.f32 = f32 + (size_t)s * k;
    q.k = k;
    q.xq = xq + (size_t)s * xq_stride_of(k);
    q.ssq = ssq + (size_t)s * ssq_stride_of(k);
    c->xqs.push_back(q);
  }
  return LGH_OK;
}

XqBuf* view_of(lgh_ctx* c, const float* f32) {
  for (auto& q : c->xqs)
    if (q.f32 == f32) return &q;
  return nullptr;
}

// after a batched launch produced / consumed the images of sequence 0's view, the other sequences' views are in the same state
void spread_state(lgh_ctx* c, const float* f32, uint32_t k, uint32_t n) {
  XqBuf* q0 = view_of(c, f32);
  if (!q0) return;
  for (uint32_t s = 1; s < n; s++)
    if (XqBuf* q = view_of(c, f32 + (size_t)s * k)) { q->fresh = q0->fresh; q->tag = q0->tag; }
}

// one batched quantized mat-vec launch: `specs` name sequence 0's vectors
struct MvIndirect { const int* cnt = nullptr; const int* idx = nullptr; uint32_t div = 0, nz = 0, stride = 0; };

// MoE layers: from how many sequences on a step reads every selected expert ONCE (one router launch, a grouping launch, gate | up
// and down over all experts with the expert as the grid's third dimension + their epilogue launches, one combine: 7 per layer
// whatever the step's size) instead of running the FFN sequence by sequence (3 launches per sequence).  Measured on Mixtral-8x7B
// Q5_K_M, tokens/s grouped vs sequence by sequence: 2 sequences 363 vs 392, 3: 447 vs 421, 4: 520 vs 436, 8: 764 vs 468,
// 16: 1 030 vs 480 (profiles/r03e_batched_decode.md).
uint32_t moe_group_min() {
  static const uint32_t v = [] { const char* e = std::getenv("LGH_MOE_GROUP_MIN"); return e ? (uint32_t)std::max(2, std::atoi(e)) : 3u; }();
  return v;
}

int launch_mvb(lgh_ctx* c, int cls, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k, uint32_t n_seq, const uint32_t* out_stride,
               const uint32_t* resid_stride, const uint32_t* xq_out_k, const MvIndirect ind = MvIndirect{}) {
  BatchScratch& Bs = c->batch;
  MvLaunch L;
  uint32_t wg = 0, threads = 0;
  uint64_t alg = 0;
  // partial sums of all sequences in LDS: n_seq * npass * T * 16 * R floats <= ~120 KB
  uint32_t cap = 32;
  {
    int npass_max = 1;
    for (int s = 0; s < nseg; s++) npass_max = std::max(npass_max, specs[s].npass);
    const uint32_t budget_floats = 120u * 1024u / 4u / n_seq;
    cap = std::max(1u, budget_floats / ((uint32_t)npass_max * 8u * 16u));
  }
  int rc = build_mv_group(c, specs, nseg, norm_w, k, true, L, wg, threads, alg, cap);
  if (rc) return rc;
  if (mvqb_lds_bytes(n_seq, L.red_floats) > 160 * 1024) return fail(c, LGH_UNSUPPORTED, "multi-sequence mat-vec: partial sums do not fit LDS");
  MvBatch B{};
  B.n_seq = n_seq;
  B.pos = Bs.d_pos;
  B.slot = Bs.d_slot;
  B.cache_stride = Bs.cache_stride;
  B.part = Bs.mv_part;
  B.part_floats = Bs.mv_part_floats;
  B.ind_cnt = ind.cnt;
  B.ind_idx = ind.idx;
  B.ind_div = ind.div;
  B.ind_nz = ind.nz;
  B.ind_stride = ind.stride;
  {   // the input vector's images lie at the strides its views were registered with
    const XqBuf* q0 = view_of(c, specs[0].x[0]);
    const uint32_t kreg = q0 ? q0->k : k;
    B.xq_stride = xq_stride_of(kreg);
    B.ssq_stride = ssq_stride_of(kreg);
  }
  for (int s = 0; s < nseg; s++) {
    B.out_stride[s] = out_stride[s];
    B.resid_stride[s] = resid_stride[s];
    B.xq_out_stride[s] = xq_out_k[s] ? xq_stride_of(xq_out_k[s]) : 0;
    B.ssq_out_stride[s] = xq_out_k[s] ? ssq_stride_of(xq_out_k[s]) : 0;
  }
  return run_k(c, cls, LGH_SYM_OTHER, alg, [&] { return mvqb_launch(L, B, wg, threads, c->stream); });
}

bool batch_weights_ok(const lgh_ctx* c, std::string& why) {
  const lgh_model_desc& d = c->d;
  if (d.hidden_size % 256 || (d.num_heads * d.head_dim) % 256) { why = "hidden size and heads x head_dim must be multiples of 256"; return false; }
  if (d.use_neox_rope) { why = "NeoX RoPE is not fused into the QKV launch"; return false; }
  if (!attn_shape_has_fast_kernel(d.head_dim, d.num_heads / d.num_kv_heads)) { why = "attention shape outside the split kernels"; return false; }
  if ((d.flags & LGH_FLAG_KV_INT8) && !kv_is_tq(d.kv_cache_type)) { why = "the int8 / FP8 KV caches are not batched (f32 and TurboQuant are)"; return false; }
  if (!(c->first && c->last)) { why = "pipeline stages are not batched"; return false; }
  for (uint32_t i = c->l0; i < c->l1; i++) {
    const LayerW& L = c->layers[i];
    for (const DevWeight* W : {&L.wq, &L.wk, &L.wv, &L.wo})
      if (!mfma_type(W->type) || W->n % 16) { why = "attention weights of layer " + std::to_string(i) + " are not in a matrix-core tile layout"; return false; }
    if (L.bq || L.bk || L.bv) { /* biases are per row: fine */ }
    if (!L.moe()) {
      if (d.intermediate_size % 256) { why = "intermediate size must be a multiple of 256"; return false; }
      for (const DevWeight* W : {&L.gate, &L.up, &L.down})
        if (!mfma_type(W->type) || W->n % 16) { why = "FFN weights of layer " + std::to_string(i) + " are not in a matrix-core tile layout"; return false; }
      if (L.gate.type != L.up.type) { why = "gate and up differ in format"; return false; }
    }
  }
  if (!mfma_type(c->output.type)) { why = "the output projection is not in a matrix-core tile layout"; return false; }
  return true;
}

// everything one step of n_seq sequences needs, in stream order (eager or under capture); tokens / positions / slots are read
// from the device words
int enqueue_multi(lgh_ctx* c, uint32_t n_seq, bool greedy) {
  BatchScratch& Bs = c->batch;
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size, QD = d.num_heads * d.head_dim;
  int rc;
  auto stale_all = [&] { for (auto& q : c->xqs) q.fresh = false; };
  stale_all();
  // ---- embedding rows + the first layer's XQ image (embed_kernel's arithmetic)
  {
    LayerW& L0 = c->layers[c->l0];
    XqBuf* qh = view_of(c, Bs.hidden);
    if ((rc = run_k(c, LGH_K_EMBED, LGH_SYM_EMBED, (uint64_t)n_seq * H * blk_bytes(c->embd_type) / blk_elems(c->embd_type), [&] {
           return embed_multi_launch(c->embd_type, c->embd_raw, Bs.d_tokens, Bs.hidden, H, n_seq, qh->xq, L0.attn_norm, qh->ssq, xq_stride_of(H),
                                     ssq_stride_of(H), c->stream);
         })))
      return rc;
    for (uint32_t s = 0; s < n_seq; s++)
      if (XqBuf* q = view_of(c, Bs.hidden + (size_t)s * H)) { q->fresh = true; q->tag = L0.attn_norm; }
  }
  const float scale = 1.0f / std::sqrt((float)d.head_dim);
  for (uint32_t li = c->l0; li < c->l1; li++) {
    LayerW& Lw = c->layers[li];
    const float* next_nw = li + 1 < c->l1 ? c->layers[li + 1].attn_norm : c->output_norm;
    // ---- Q, K, V (+ RoPE at every sequence's own position, K / V rows into its own cache slot)
    {
      const bool tq = kv_is_tq(d.kv_cache_type);
      const uint32_t KD = d.num_kv_heads * d.head_dim;
      SegSpec sp[3];
      sp[0].W[0] = &Lw.wq; sp[0].x[0] = Bs.hidden; sp[0].epi = EPI_ROPE_Q; sp[0].out = Bs.q; sp[0].bias = Lw.bq;
      // (TurboQuant: the rotated K row and the V row stay f32 in a staging vector per sequence; the attention launch compresses them)
      sp[1].W[0] = &Lw.wk; sp[1].x[0] = Bs.hidden; sp[1].epi = tq ? EPI_ROPE_Q : EPI_ROPE_K; sp[1].out = tq ? Bs.kv_tmp : Bs.kcache[li]; sp[1].bias = Lw.bk;
      sp[2].W[0] = &Lw.wv; sp[2].x[0] = Bs.hidden; sp[2].epi = tq ? EPI_STORE : EPI_V_CACHE; sp[2].out = tq ? Bs.kv_tmp + KD : Bs.vcache[li]; sp[2].bias = Lw.bv;
      const uint32_t os[3] = {QD, tq ? 2 * KD : 0, tq ? 2 * KD : 0}, rs[3] = {0, 0, 0}, xk[3] = {0, 0, 0};
      // (the three matrices may come in formats without a common instantiation: one launch each then, as launch_mv does)
      bool q4 = false, q5 = false, other = false, uniform = true;
      for (int s = 0; s < 3; s++) {
        const int t = sp[s].W[0]->type;
        q4 |= t == kDevQ4K_T16; q5 |= t == kDevQ5K_T16; other |= t == kDevQ80_T16 || t == kDevQ40_T16;
        uniform &= t == sp[0].W[0]->type;
      }
      if (!uniform && ((q4 && q5) || other)) {
        for (int s = 0; s < 3; s++)
          if ((rc = launch_mvb(c, LGH_K_QKV, sp + s, 1, Lw.attn_norm, H, n_seq, os + s, rs + s, xk + s))) return rc;
      } else if ((rc = launch_mvb(c, LGH_K_QKV, sp, 3, Lw.attn_norm, H, n_seq, os, rs, xk))) {
        return rc;
      }
    }
    // ---- attention_cached per sequence (ops.rs:1479-1537): split + merge, the sequence as the grid's second dimension
    if (kv_is_tq(d.kv_cache_type)) {
      // TurboQuantKVCache (kv_turboquant.rs) per sequence: write_kv + attention over the slot's codes; the merge inverts the V rotation
      const int bits = kv_tq_bits(d.kv_cache_type);
      const bool qjl = kv_is_qjl(d.kv_cache_type);
      const uint32_t KD = d.num_kv_heads * d.head_dim;
      const float* signs = c->tq_signs + (size_t)(li - c->l0) * d.num_kv_heads * 2 * d.head_dim;
      const float* qjl_s = qjl ? c->tq_qjl + (size_t)(li - c->l0) * d.num_kv_heads * d.head_dim * d.head_dim : nullptr;
      if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, 0, [&] {
             return attn_tq_multi_launch(bits, Bs.q, Bs.kq[li], Bs.vq[li], Bs.kv_tmp, Bs.kv_tmp + KD, signs, d.num_heads, d.num_kv_heads, d.head_dim,
                                         d.max_seq_len, scale, Bs.d_pos, Bs.d_slot, Bs.code_stride, Bs.x_stride, 2 * KD, n_seq, c->n_splits, Bs.part_ml,
                                         Bs.part_acc, c->stream, qjl_s, qjl ? Bs.kx[li] : nullptr);
           })))
        return rc;
      XqBuf* qa = view_of(c, Bs.attn_out);
      if ((rc = run_k(c, LGH_K_ATTN_COMBINE, LGH_SYM_ATTN_COMBINE, 0, [&] {
             return attn_tq_combine_launch(bits, Bs.part_ml, Bs.part_acc, signs, d.num_heads, d.num_kv_heads, d.head_dim, c->n_splits, Bs.attn_out, qa->xq,
                                           c->stream, n_seq, (uint32_t)xq_stride_of(QD));
           })))
        return rc;
      for (uint32_t s = 0; s < n_seq; s++)
        if (XqBuf* q = view_of(c, Bs.attn_out + (size_t)s * QD)) { q->fresh = true; q->tag = nullptr; }
    } else {
    if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, 0, [&] {
           return attn_multi_launch(Bs.q, Bs.kcache[li], Bs.vcache[li], d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len, scale, Bs.d_pos, Bs.d_slot,
                                    Bs.cache_stride, n_seq, c->n_splits, Bs.part_ml, Bs.part_acc, c->stream);
         })))
      return rc;
    {
      XqBuf* qa = view_of(c, Bs.attn_out);
      if ((rc = run_k(c, LGH_K_ATTN_COMBINE, LGH_SYM_ATTN_COMBINE, 0, [&] {
             return attn_combine_multi_launch(Bs.part_ml, Bs.part_acc, d.num_heads, d.num_kv_heads, d.head_dim, c->n_splits, n_seq, Bs.attn_out, qa->xq,
                                              c->stream);
           })))
        return rc;
      for (uint32_t s = 0; s < n_seq; s++)
        if (XqBuf* q = view_of(c, Bs.attn_out + (size_t)s * QD)) { q->fresh = true; q->tag = nullptr; }
    }
    }
    // ---- h = x + wo(attn)
    {
      SegSpec sp;
      sp.W[0] = &Lw.wo; sp.x[0] = Bs.attn_out; sp.epi = EPI_RESID; sp.out = Bs.hidden; sp.resid = Bs.hidden; sp.bias = Lw.bo;
      sp.xq_next = 2; sp.xq_next_nw = Lw.ffn_norm;
      const uint32_t os[1] = {H}, rs[1] = {H}, xk[1] = {H};
      if ((rc = launch_mvb(c, LGH_K_WO, &sp, 1, nullptr, Lw.wo.k, n_seq, os, rs, xk))) return rc;
      spread_state(c, Bs.hidden, H, n_seq);
    }
    // ---- FFN
    if (!Lw.moe()) {
      {
        SegSpec sp;
        sp.npass = 2;
        sp.W[0] = &Lw.gate; sp.W[1] = &Lw.up;
        sp.x[0] = sp.x[1] = Bs.hidden;
        sp.epi = EPI_SWIGLU;
        sp.out = Bs.act;
        sp.xq_next = 1;
        const uint32_t os[1] = {Bs.ffn}, rs[1] = {0}, xk[1] = {Bs.ffn};
        if ((rc = launch_mvb(c, LGH_K_GATEUP, &sp, 1, Lw.ffn_norm, H, n_seq, os, rs, xk))) return rc;
        spread_state(c, Bs.act, Bs.ffn, n_seq);
      }
      {
        SegSpec sp;
        sp.W[0] = &Lw.down; sp.x[0] = Bs.act; sp.epi = EPI_RESID; sp.out = Bs.hidden; sp.resid = Bs.hidden;
        sp.xq_next = 2; sp.xq_next_nw = next_nw;
        const uint32_t os[1] = {H}, rs[1] = {H}, xk[1] = {H};
        if ((rc = launch_mvb(c, LGH_K_DOWN, &sp, 1, nullptr, Lw.down.k, n_seq, os, rs, xk))) return rc;
        spread_state(c, Bs.hidden, H, n_seq);
      }
    } else if (Bs.moe_act && n_seq >= moe_group_min()) {
      // ---- MoE, every selected expert's matrices read ONCE for the step (MoeLayer::forward per sequence, moe.rs:321-413, regrouped):
      // router -> the (sequence, slot) pairs grouped by expert -> a gate|up launch, every expert over its pairs (SwiGLU, activation
      // + XQ image per pair), and a down launch (output per pair) -> h += sum_p w_p * down_p in selection order
      const uint32_t topk = d.num_experts_per_token, ne = d.num_experts, EF = Lw.gate_exps.n;
      // (the router for all sequences in one launch: selections and weights packed [sequence][top_k])
      if ((rc = run_k(c, LGH_K_ROUTER, LGH_SYM_ROUTER, (uint64_t)ne * H * 4, [&] {
             return moe_router_launch(Bs.hidden, Lw.ffn_norm, d.norm_eps, Lw.router, H, ne, topk, Bs.moe_sel, Bs.moe_w, c->stream, n_seq);
           })))
        return rc;
      if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] { return moe_group_launch(Bs.moe_sel, n_seq, topk, ne, Bs.moe_cnt, Bs.moe_idx, kMaxBatch, c->stream); })))
        return rc;
      // all experts of the layer in one launch each (grid z = expert: its count, its entry list, its slice of the stacked matrices)
      const MvIndirect ind_gu{Bs.moe_cnt, Bs.moe_idx, topk, ne, (uint32_t)kMaxBatch}, ind_dn{Bs.moe_cnt, Bs.moe_idx, 1, ne, (uint32_t)kMaxBatch};
      {
        SegSpec sp;
        sp.npass = 2;
        sp.W[0] = &Lw.gate_exps; sp.W[1] = &Lw.up_exps;
        sp.x[0] = sp.x[1] = Bs.hidden;
        sp.epi = EPI_SWIGLU;
        sp.out = Bs.moe_act;
        sp.xq_next = 1;
        const uint32_t os[1] = {EF}, rs[1] = {0}, xk[1] = {EF};
        if ((rc = launch_mvb(c, LGH_K_GATEUP, &sp, 1, Lw.ffn_norm, H, n_seq, os, rs, xk, ind_gu))) return rc;
      }
      for (uint32_t v = 0; v < n_seq * topk; v++)
        if (XqBuf* q = view_of(c, Bs.moe_act + (size_t)v * EF)) { q->fresh = true; q->tag = nullptr; }
      {
        SegSpec sp;
        sp.W[0] = &Lw.down_exps; sp.x[0] = Bs.moe_act; sp.epi = EPI_STORE; sp.out = Bs.moe_tmp;
        const uint32_t os[1] = {H}, rs[1] = {0}, xk[1] = {0};
        if ((rc = launch_mvb(c, LGH_K_DOWN, &sp, 1, nullptr, Lw.down_exps.k, n_seq, os, rs, xk, ind_dn))) return rc;
      }
      {
        XqBuf* qh = view_of(c, Bs.hidden);
        if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] {
               return moe_combine_launch(Bs.moe_tmp, Bs.moe_w, topk, Bs.hidden, H, n_seq, next_nw, qh->xq, xq_stride_of(H), qh->ssq, ssq_stride_of(H), c->stream);
             })))
          return rc;
        for (uint32_t s = 0; s < n_seq; s++)
          if (XqBuf* q = view_of(c, Bs.hidden + (size_t)s * H)) { q->fresh = true; q->tag = next_nw; }
      }
    } else {
      for (uint32_t s = 0; s < n_seq; s++) {   // every sequence routes to its own experts: the single-sequence launches on its vectors
        const FfnView v{Bs.hidden + (size_t)s * H, Bs.act + (size_t)s * Bs.ffn, Bs.act2 + (size_t)s * Bs.ffn, Bs.xnorm + (size_t)s * H,
                        Bs.moe_sel + s * 8, Bs.moe_w + s * 8};
        if ((rc = ffn_forward(c, Lw, v, next_nw, true))) return rc;
      }
    }
  }
  // ---- compute_logits: final RMSNorm fused into the output projection
  {
    SegSpec sp;
    sp.W[0] = &c->output; sp.x[0] = Bs.hidden; sp.epi = EPI_STORE; sp.out = Bs.logits;
    const uint32_t os[1] = {d.vocab_size}, rs[1] = {0}, xk[1] = {0};
    if ((rc = launch_mvb(c, LGH_K_OUTPUT, &sp, 1, c->output_norm, H, n_seq, os, rs, xk))) return rc;
  }
  if (greedy) {
    // arg-max per sequence (last maximal index), fed back as the next step's token; the positions move on
    if ((rc = run_k(c, LGH_K_ARGMAX, LGH_SYM_ARGMAX, (uint64_t)n_seq * d.vocab_size * 4, [&] {
           return argmax_multi_launch(Bs.logits, d.vocab_size, n_seq, Bs.amax_v, Bs.amax_i, Bs.d_tokens, c->stream);
         })))
      return rc;
  }
  if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] {
         hipLaunchKernelGGL(batch_advance_kernel, dim3(1), dim3(64), 0, c->stream, Bs.d_pos, (int)n_seq);
         return hipGetLastError();
       })))
    return rc;
  stale_all();
  return LGH_OK;
}

int stage_control(lgh_ctx* c, const uint32_t* slots, const uint32_t* tokens, uint32_t n_seq) {
  BatchScratch& Bs = c->batch;
  const lgh_model_desc& d = c->d;
  if (!Bs.ready) return fail(c, LGH_INVALID_ARGUMENT, "lgh_batch_create has not been called");
  if (n_seq == 0 || n_seq > Bs.max_batch) return fail(c, LGH_INVALID_ARGUMENT, "n_seq must be 1 .. max_batch");
  bool seen[kMaxBatch] = {};
  for (uint32_t i = 0; i < n_seq; i++) {
    if (slots[i] >= Bs.max_batch || seen[slots[i]]) return fail(c, LGH_INVALID_ARGUMENT, "slot numbers must be distinct and below max_batch");
    seen[slots[i]] = true;
    if (Bs.pos[slots[i]] >= d.max_seq_len)
      return fail(c, LGH_INVALID_ARGUMENT, "slot " + std::to_string(slots[i]) + ": position " + std::to_string(Bs.pos[slots[i]]) + " >= max_seq_len");
    if (tokens && tokens[i] >= d.vocab_size) return fail(c, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");
  }
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));   // the pinned words below are free again
  for (uint32_t i = 0; i < n_seq; i++) {
    Bs.h_ctl[i] = tokens ? (int)tokens[i] : 0;
    Bs.h_ctl[kMaxBatch + i] = (int)Bs.pos[slots[i]];
    Bs.h_ctl[2 * kMaxBatch + i] = (int)slots[i];
  }
  if (tokens) HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(Bs.d_tokens, Bs.h_ctl, (size_t)n_seq * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(Bs.d_pos, Bs.h_ctl + kMaxBatch, (size_t)n_seq * 4, hipMemcpyHostToDevice, c->stream));
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(Bs.d_slot, Bs.h_ctl + 2 * kMaxBatch, (size_t)n_seq * 4, hipMemcpyHostToDevice, c->stream));
  return LGH_OK;
}

int run_multi(lgh_ctx* c, uint32_t n_seq, bool greedy) {
  BatchScratch& Bs = c->batch;
  if (c->profiling || (c->d.flags & LGH_FLAG_NO_GRAPH)) {
    int rc = enqueue_multi(c, n_seq, greedy);
    if (rc) return rc;
    return c->profiling ? drain_prof(c) : LGH_OK;
  }
  hipGraphExec_t& ge = Bs.graph[n_seq][greedy ? 1 : 0];
  if (!ge) {
    hipGraph_t g = nullptr;
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    int rc = enqueue_multi(c, n_seq, greedy);
    hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
    if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  }
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipGraphLaunch(ge, c->stream));
  return LGH_OK;
}

int check_batch(lgh_ctx* c) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (!c->finalized) return fail(c, LGH_INVALID_ARGUMENT, "context not finalized");
  if (hipSetDevice(c->device) != hipSuccess) return fail(c, LGH_NOT_AVAILABLE, "hipSetDevice failed");
  return LGH_OK;
}

}  // namespace

extern "C" {

int lgh_batch_create(lgh_ctx* c, uint32_t max_batch) {
  int rc = check_batch(c);
  if (rc) return rc;
  BatchScratch& Bs = c->batch;
  if (Bs.ready) return Bs.max_batch == max_batch ? LGH_OK : fail(c, LGH_INVALID_ARGUMENT, "lgh_batch_create was already called with another max_batch");
  if (max_batch == 0 || max_batch > (uint32_t)kMaxBatch) return fail(c, LGH_INVALID_ARGUMENT, "max_batch must be 1 .. 16");
  std::string why;
  if (!batch_weights_ok(c, why)) return fail(c, LGH_UNSUPPORTED, "multi-sequence decode: " + why);
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size, QD = d.num_heads * d.head_dim, G = d.num_heads / d.num_kv_heads;
  const uint32_t EI = d.expert_intermediate_size ? d.expert_intermediate_size : d.intermediate_size;
  const uint32_t ffn = std::max(d.intermediate_size, EI);
  Bs.max_batch = max_batch;
  Bs.ffn = ffn;
  Bs.cache_stride = (uint64_t)d.num_kv_heads * d.max_seq_len * d.head_dim;
  const size_t B = max_batch;
  // partial sums between the two launches of a multi-sequence mat-vec (matvec_batch.hip, mvqb2): per sequence either one value per
  // launch row (any matrix) or 8 slices x the rows of a matrix with fewer than 1600 row tiles
  const uint64_t moe_part = d.num_experts ? (uint64_t)d.num_experts * std::max<uint64_t>((uint64_t)2 * EI, (uint64_t)8 * H) : 0;   // every expert's region
  Bs.mv_part_floats = B * std::max<uint64_t>({moe_part, (uint64_t)8 * 1600 * 16, (uint64_t)d.vocab_size + 16, (uint64_t)2 * ffn, (uint64_t)QD + 2 * d.num_kv_heads * d.head_dim, (uint64_t)H});
  uint8_t *xq_h = nullptr, *xq_a = nullptr, *xq_f = nullptr, *xq_f2 = nullptr;
  float *ssq_h = nullptr, *ssq_a = nullptr, *ssq_f = nullptr, *ssq_f2 = nullptr;
  struct { void** p; size_t n; } bufs[] = {
      {(void**)&Bs.hidden, B * H * 4},   {(void**)&Bs.xnorm, B * H * 4},   {(void**)&Bs.q, B * QD * 4},   {(void**)&Bs.attn_out, B * QD * 4},
      {(void**)&Bs.act, B * ffn * 4},    {(void**)&Bs.act2, B * ffn * 4},  {(void**)&Bs.logits, B * d.vocab_size * 4},
      {(void**)&Bs.part_ml, B * d.num_kv_heads * c->n_splits * G * 2 * 4}, {(void**)&Bs.part_acc, B * d.num_kv_heads * c->n_splits * G * d.head_dim * 4},
      {(void**)&Bs.mv_part, Bs.mv_part_floats * 4},
      {(void**)&Bs.amax_v, B * 64 * 4},  {(void**)&Bs.amax_i, B * 64 * 4}, {(void**)&Bs.moe_sel, B * 8 * 4}, {(void**)&Bs.moe_w, B * 8 * 4},
      {(void**)&Bs.d_tokens, kMaxBatch * 4}, {(void**)&Bs.d_pos, kMaxBatch * 4}, {(void**)&Bs.d_slot, kMaxBatch * 4},
      {(void**)&Bs.d_log, (size_t)d.max_seq_len * kMaxBatch * 4},
      {(void**)&xq_h, B * xq_stride_of(H)},   {(void**)&ssq_h, B * ssq_stride_of(H) * 4},
      {(void**)&xq_a, B * xq_stride_of(QD)},  {(void**)&ssq_a, B * ssq_stride_of(QD) * 4},
      {(void**)&xq_f, B * xq_stride_of(ffn)}, {(void**)&ssq_f, B * ssq_stride_of(ffn) * 4},
      {(void**)&xq_f2, B * xq_stride_of(ffn)}, {(void**)&ssq_f2, B * ssq_stride_of(ffn) * 4},
  };
  for (auto& b : bufs) {
    if ((rc = dev_alloc(c, b.p, b.n))) return rc;
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(*b.p, 0, b.n, c->stream));
    c->stats.scratch_bytes += b.n;
  }
  Bs.kcache.assign(d.num_layers, nullptr);
  Bs.vcache.assign(d.num_layers, nullptr);
  Bs.kq.assign(d.num_layers, nullptr);
  Bs.vq.assign(d.num_layers, nullptr);
  Bs.kx.assign(d.num_layers, nullptr);
  if (kv_is_tq(d.kv_cache_type)) {   // TurboQuantKVCache::new per slot (kv_turboquant.rs:36-86): packed codes (+ QJL rows of K)
    const size_t rows = (size_t)d.num_kv_heads * d.max_seq_len;
    Bs.code_stride = rows * tq_row_bytes_host(kv_tq_bits(d.kv_cache_type), d.head_dim);
    Bs.x_stride = rows * (d.head_dim / 32 + 1);
    if ((rc = dev_alloc(c, (void**)&Bs.kv_tmp, B * 2 * d.num_kv_heads * d.head_dim * 4))) return rc;
    for (uint32_t i = c->l0; i < c->l1; i++) {
      const size_t n = B * Bs.code_stride;
      if ((rc = dev_alloc(c, (void**)&Bs.kq[i], n)) || (rc = dev_alloc(c, (void**)&Bs.vq[i], n))) return rc;
      HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(Bs.kq[i], 0, n, c->stream));
      HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(Bs.vq[i], 0, n, c->stream));
      c->stats.kv_bytes += 2 * n;
      if (kv_is_qjl(d.kv_cache_type)) {
        const size_t nx = B * Bs.x_stride * 4;
        if ((rc = dev_alloc(c, (void**)&Bs.kx[i], nx))) return rc;
        HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(Bs.kx[i], 0, nx, c->stream));
        c->stats.kv_bytes += nx;
      }
    }
  }
  for (uint32_t i = c->l0; i < c->l1 && !kv_is_tq(d.kv_cache_type); i++) {   // per layer K / V of every slot: [slot][kv_head][max_seq][head_dim] f32
    const size_t n = B * Bs.cache_stride * 4;
    if ((rc = dev_alloc(c, (void**)&Bs.kcache[i], n)) || (rc = dev_alloc(c, (void**)&Bs.vcache[i], n))) return rc;
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(Bs.kcache[i], 0, n, c->stream));
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(Bs.vcache[i], 0, n, c->stream));
    c->stats.kv_bytes += 2 * n;
  }
  {   // MoE layers: the expert-grouped step's buffers ((sequence, slot) pairs)
    bool any_moe = false;
    for (uint32_t i = c->l0; i < c->l1; i++) any_moe = any_moe || c->layers[i].moe();
    // (the multi-sequence kernels run 8-wave workgroups: both expert shapes must plan to T x G = 8, as Mixtral's do)
    MvPlan pg{}, pd{};
    const bool plans = mvq_plan(H, EI, 2, &pg, EI) == hipSuccess && mvq_plan(EI, H, 1, &pd, H) == hipSuccess && pg.threads == 512 && pd.threads == 512;
    if (any_moe && plans && d.num_experts_per_token <= 8 && d.num_experts <= 64 && EI % 256 == 0) {
      const size_t np = B * d.num_experts_per_token;
      uint8_t* xq_m = nullptr;
      float* ssq_m = nullptr;
      struct { void** p; size_t n; } mb[] = {
          {(void**)&Bs.moe_act, np * EI * 4}, {(void**)&Bs.moe_tmp, np * H * 4}, {(void**)&Bs.moe_cnt, 64 * 4}, {(void**)&Bs.moe_idx, 64 * kMaxBatch * 4},
          {(void**)&xq_m, np * xq_stride_of(EI)}, {(void**)&ssq_m, np * ssq_stride_of(EI) * 4}};
      for (auto& b : mb) {
        if ((rc = dev_alloc(c, b.p, b.n))) return rc;
        HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemsetAsync(*b.p, 0, b.n, c->stream));
        c->stats.scratch_bytes += b.n;
      }
      register_views(c, Bs.moe_act, xq_m, ssq_m, EI, (uint32_t)np);
    }
  }
  HIP_TRYB(c, LGH_ALLOCATION_FAILED, hipHostMalloc((void**)&Bs.h_ctl, 3 * kMaxBatch * 4, hipHostMallocDefault));
  register_views(c, Bs.hidden, xq_h, ssq_h, H, max_batch);
  register_views(c, Bs.attn_out, xq_a, ssq_a, QD, max_batch);
  register_views(c, Bs.act, xq_f, ssq_f, ffn, max_batch);
  register_views(c, Bs.act2, xq_f2, ssq_f2, ffn, max_batch);
  Bs.pos.assign(max_batch, 0);
  Bs.ready = true;
  // one step through every kernel of the path, eagerly, before any of them is first launched inside a capture (engine.hip:
  // warm_kernels explains the ROCm trap); it writes row 0 of slot 0's caches, which that slot's first real token overwrites
  {
    const uint32_t slot0 = 0, tok0 = 0;
    if ((rc = stage_control(c, &slot0, &tok0, 1))) return rc;
    if ((rc = enqueue_multi(c, 1, true))) return rc;
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
    if (max_batch >= 2) {   // ... and the kernels only a step of several sequences launches (rows 0 of slots 0 and 1)
      const uint32_t slots2[2] = {0, 1}, toks2[2] = {0, 0};
      if ((rc = stage_control(c, slots2, toks2, 2))) return rc;
      if ((rc = enqueue_multi(c, 2, true))) return rc;
      HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
    }
  }
  return LGH_OK;
}

int lgh_batch_reset(lgh_ctx* c, uint32_t slot) {   // create_active_sequence: a fresh context for the slot (O(1): position rewind)
  int rc = check_batch(c);
  if (rc) return rc;
  if (!c->batch.ready || slot >= c->batch.max_batch) return fail(c, LGH_INVALID_ARGUMENT, "no such slot");
  c->batch.pos[slot] = 0;
  return LGH_OK;
}

size_t lgh_batch_position(lgh_ctx* c, uint32_t slot) {
  if (!c || !c->batch.ready || slot >= c->batch.max_batch) return 0;
  return c->batch.pos[slot];
}

int lgh_forward_multi(lgh_ctx* c, const uint32_t* slots, const uint32_t* tokens, uint32_t n_seq, float* logits_out, uint32_t* next_tokens) {
  int rc = check_batch(c);
  if (rc) return rc;
  if (!slots || !tokens) return fail(c, LGH_INVALID_ARGUMENT, "slots / tokens is NULL");
  if ((rc = stage_control(c, slots, tokens, n_seq))) return rc;
  if ((rc = run_multi(c, n_seq, next_tokens != nullptr))) return rc;
  BatchScratch& Bs = c->batch;
  if (logits_out)
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(logits_out, Bs.logits, (size_t)n_seq * c->d.vocab_size * 4, hipMemcpyDeviceToHost, c->stream));
  if (next_tokens) HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(next_tokens, Bs.d_tokens, (size_t)n_seq * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  for (uint32_t i = 0; i < n_seq; i++) Bs.pos[slots[i]] += 1;
  c->stats.tokens_processed += n_seq;
  return LGH_OK;
}

int lgh_decode_greedy_multi(lgh_ctx* c, const uint32_t* slots, const uint32_t* first_tokens, uint32_t n_seq, size_t n_steps, uint32_t* tokens_out) {
  int rc = check_batch(c);
  if (rc) return rc;
  if (!slots || !first_tokens) return fail(c, LGH_INVALID_ARGUMENT, "slots / first_tokens is NULL");
  BatchScratch& Bs = c->batch;
  if (Bs.ready)
    for (uint32_t i = 0; i < n_seq && i < (uint32_t)kMaxBatch; i++)
      if (slots[i] < Bs.max_batch && Bs.pos[slots[i]] + n_steps > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "the steps would run past max_seq_len");
  if (n_steps > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "too many steps");
  if ((rc = stage_control(c, slots, first_tokens, n_seq))) return rc;
  for (size_t st = 0; st < n_steps; st++) {
    if ((rc = run_multi(c, n_seq, true))) return rc;
    if (tokens_out)
      HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpyAsync(Bs.d_log + st * kMaxBatch, Bs.d_tokens, (size_t)n_seq * 4, hipMemcpyDeviceToDevice, c->stream));
  }
  HIP_TRYB(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  if (tokens_out && n_steps) {
    std::vector<int> log(n_steps * kMaxBatch);
    HIP_TRYB(c, LGH_OPERATION_FAILED, hipMemcpy(log.data(), Bs.d_log, log.size() * 4, hipMemcpyDeviceToHost));
    for (size_t st = 0; st < n_steps; st++)
      for (uint32_t i = 0; i < n_seq; i++) tokens_out[st * n_seq + i] = (uint32_t)log[st * kMaxBatch + i];
  }
  for (uint32_t i = 0; i < n_seq; i++) Bs.pos[slots[i]] += n_steps;
  c->stats.tokens_processed += n_steps * n_seq;
  return LGH_OK;
}

}  // extern "C"
