// oracle/model.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference's decode graph for Llama-family models (dense SwiGLU FFN or
// Mixtral-style MoE), in the reference's op order:
//
//   LlamaModel::forward / compute_logits         src/model/llama.rs:247-362
//   TransformerLayer::forward (serial residual)  src/model/layers.rs:1082-1245
//   Attention::forward (full-attention branch)   src/model/layers.rs:409-704
//   FeedForward::forward                         src/model/layers.rs:908-929
//   MoeLayer / MoeExpert::forward                src/model/moe.rs:227-268, 321-413
//   Linear::forward (+bias)                      src/model/layers.rs:56-77
//   KVCache layout [kv_heads, max_seq, head_dim] src/model/mod.rs:64-108
//   tensor names                                 src/model/loader.rs:592-854, 1140-1313
#include "oracle.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

void orc_parallel_for(size_t total, size_t grain, const std::function<void(size_t, size_t)>& fn);

namespace {

struct TensorRef {
  int type = -1;
  uint64_t ne[4] = {0, 0, 0, 0};
  const uint8_t* data = nullptr;
  size_t nbytes = 0;
  std::vector<uint8_t> owned;
  bool present() const { return data != nullptr; }
  size_t numel() const { return (size_t)(ne[0] * (ne[1] ? ne[1] : 1) * (ne[2] ? ne[2] : 1) * (ne[3] ? ne[3] : 1)); }
};

struct LinearRef {  // GGUF: weight [in_features, out_features]
  const TensorRef* w = nullptr;
  const TensorRef* bias = nullptr;
  size_t in_f = 0, out_f = 0;
};

struct ExpertRef {
  const uint8_t *gate = nullptr, *up = nullptr, *down = nullptr;
  int gate_type = 0, up_type = 0, down_type = 0;
};

struct LayerRef {
  const TensorRef *attn_norm = nullptr, *ffn_norm = nullptr;
  LinearRef wq, wk, wv, wo, gate, up, down;
  const TensorRef* router = nullptr;  // ffn_gate_inp.weight [hidden, n_experts] f32
  std::vector<ExpertRef> experts;
};

}  // namespace

struct orc_model {
  orc_config cfg;
  std::map<std::string, TensorRef> tensors;
  std::vector<LayerRef> layers;
  const TensorRef* token_embd = nullptr;
  const TensorRef* output_norm = nullptr;
  LinearRef output;
  std::vector<float> embd_f32;  // dequantized table (kept unless faithful mode re-does it per call)
  std::vector<std::vector<float>> k_cache, v_cache;  // per layer [kv_heads][max_seq][head_dim]
  bool kv_int8 = false;                               // rows go through the reference's int8 KV format on their way into the cache
  int kv_fp8 = 0;                                     // ... or through one of its FP8 formats (1: E4M3, 2: E5M2)
  // TurboQuantKVCache (src/model/kv_turboquant.rs): Attention::forward_turboquant (layers.rs:711-873) instead of the f32 cache
  int kv_tq_bits = 0;                                 // 2 / 3: KVCacheType::TurboQuantMSE { bits } (model/mod.rs:182-213)
  std::vector<float> tq_signs;                        // [layer][kv head][k, v][padded_dim]: HadamardRotation::signs() of every engine
  std::vector<std::vector<uint8_t>> tq_k, tq_v;       // per layer [kv head][max_seq][packed bytes]
  // KVCacheType::TurboQuantProd { bits }: + the K engines' QJL projectors (qjl.rs) and, per cached K row, sign bits + residual norm
  std::vector<float> tq_qjl;                          // [layer][kv head][padded_dim][padded_dim]; empty = TurboQuantMSE
  std::vector<std::vector<uint64_t>> tq_kbits;        // per layer [kv head][max_seq][(padded_dim + 63) / 64]
  std::vector<std::vector<float>> tq_knorm;           // per layer [kv head][max_seq]
  std::vector<float> last_hidden;
  size_t position = 0;
  bool finalized = false;
  std::string err;
};

namespace {

int fail(orc_model* m, const std::string& s) {
  m->err = s;
  return 1;
}

// Linear::forward (layers.rs:56-77): quantized -> vec_mat_q, else vec_mat (F32 only); then += bias
int linear_forward(orc_model* m, const LinearRef& l, const float* x, float* out) {
  if (!l.w) return fail(m, "missing linear weight");
  if (l.w->type == ORC_F32) {
    orc_vec_mat_f32((const float*)l.w->data, x, out, l.in_f, l.out_f);
  } else if (l.w->type == ORC_F16 || l.w->type == ORC_BF16) {
    // quirk Q3: the CPU path rejects non-F32 dense weights (ops.rs:960-962)
    return fail(m, "F16/BF16 2-D weights are rejected by the reference CPU path (ops.rs:960)");
  } else if (orc_vec_mat_q(l.w->type, l.w->data, x, out, l.in_f, l.out_f)) {
    return fail(m, "vec_mat_q failed");
  }
  if (l.bias) {
    const float* b = (const float*)l.bias->data;
    for (size_t i = 0; i < l.out_f; i++) out[i] += b[i];
  }
  return 0;
}

int raw_vec_mat(orc_model* m, int type, const uint8_t* w, const float* x, float* out, size_t k, size_t n) {
  if (type == ORC_F32) {
    orc_vec_mat_f32((const float*)w, x, out, k, n);
    return 0;
  }
  if (orc_vec_mat_q(type, w, x, out, k, n)) return fail(m, "expert vec_mat_q failed");
  return 0;
}

const TensorRef* find(orc_model* m, const std::string& name) {
  auto it = m->tensors.find(name);
  return it == m->tensors.end() ? nullptr : &it->second;
}

int bind_linear(orc_model* m, LinearRef& l, const std::string& base, bool required) {
  l.w = find(m, base + ".weight");
  if (!l.w) return required ? fail(m, "missing tensor " + base + ".weight") : 0;
  l.bias = find(m, base + ".bias");
  l.in_f = (size_t)l.w->ne[0];
  l.out_f = (size_t)l.w->ne[1];
  return 0;
}

// Attention::forward, full-attention / standard-RoPE branch (layers.rs:409-704)
int attention_forward(orc_model* m, size_t li, const float* x, size_t pos, float* out) {
  const orc_config& c = m->cfg;
  const LayerRef& L = m->layers[li];
  size_t nh = c.num_heads, nkv = c.num_kv_heads, d = c.head_dim;
  std::vector<float> q(nh * d), k(nkv * d), v(nkv * d);
  if (linear_forward(m, L.wq, x, q.data())) return 1;  // 438
  if (linear_forward(m, L.wk, x, k.data())) return 1;  // 439
  if (linear_forward(m, L.wv, x, v.data())) return 1;  // 440
  orc_rope(q.data(), k.data(), nh, nkv, 1, d, pos, c.rope_freq_base, c.rope_freq_scale, (int)c.use_neox_rope);  // 565-575
  size_t ms = c.max_seq_len;
  if (m->kv_tq_bits) {
    // forward_turboquant (layers.rs:843-852): tq_cache.write_kv(layer, k, v) — every kv head's row compressed by that head's
    // K / V engine (kv_turboquant.rs:88-122) — then backend.attention_turboquant = attention_layer (173-201): query head h reads
    // kv head h / (heads per kv head)
    const int bits = m->kv_tq_bits;
    const size_t pd = orc_tq_padded_dim(d), pb = orc_tq_packed_bytes(bits, pd);
    uint8_t* kq = m->tq_k[li].data();
    uint8_t* vq = m->tq_v[li].data();
    auto signs = [&](size_t h, int kv) { return m->tq_signs.data() + ((li * nkv + h) * 2 + (size_t)kv) * pd; };
    const bool qjl = !m->tq_qjl.empty();
    const size_t nw = (pd + 63) / 64;
    auto S = [&](size_t h) { return m->tq_qjl.data() + (li * nkv + h) * pd * pd; };
    for (size_t h = 0; h < nkv; h++) {
      if (qjl)
        orc_tq_compress_qjl(k.data() + h * d, d, bits, signs(h, 0), S(h), kq + (h * ms + pos) * pb, m->tq_kbits[li].data() + (h * ms + pos) * nw,
                            m->tq_knorm[li].data() + h * ms + pos);
      else
        orc_tq_compress(k.data() + h * d, d, bits, signs(h, 0), kq + (h * ms + pos) * pb);
      orc_tq_compress(v.data() + h * d, d, bits, signs(h, 1), vq + (h * ms + pos) * pb);   // (V's own QJL bits are never read)
    }
    std::vector<float> attn_tq(nh * d);
    const float scale_tq = 1.0f / std::sqrt((float)d);  // layers.rs:374
    const size_t per = nh / nkv;
    for (size_t h = 0; h < nh; h++) {
      const size_t kvh = h / per;
      if (qjl)
        orc_tq_attention_head_qjl(q.data() + h * d, kq + kvh * ms * pb, m->tq_kbits[li].data() + kvh * ms * nw, m->tq_knorm[li].data() + kvh * ms,
                                  vq + kvh * ms * pb, pos + 1, d, bits, signs(kvh, 0), signs(kvh, 1), S(kvh), scale_tq, attn_tq.data() + h * d);
      else
        orc_tq_attention_head(q.data() + h * d, kq + kvh * ms * pb, vq + kvh * ms * pb, pos + 1, d, bits, signs(kvh, 0), signs(kvh, 1), scale_tq,
                              attn_tq.data() + h * d);
    }
    return linear_forward(m, L.wo, attn_tq.data(), out);
  }
  float* kc = m->k_cache[li].data();
  float* vc = m->v_cache[li].data();
  if (m->kv_int8) {
    // QuantizedKVCache::write_kv + read_*_range with KVCacheFormat::Int8 (src/model/kv_quantized.rs:143-300): every head's row
    // of every position is stored as int8 with one scale; what attention later reads back is scale * q
    std::vector<int8_t> q8(d);
    for (size_t h = 0; h < nkv; h++)
      for (float* row : {k.data() + h * d, v.data() + h * d}) {
        float sc = 1.0f;
        orc_kv_quantize_int8(row, d, q8.data(), &sc);
        orc_kv_dequantize_int8(q8.data(), sc, d, row);
      }
  }
  if (m->kv_fp8) {
    // KVCacheFormat::Fp8E4M3 / Fp8E5M2 (kv_quantized.rs:190-205, 256-268): one byte per element, no scales
    for (float* vec : {k.data(), v.data()})
      for (size_t i = 0; i < nkv * d; i++) vec[i] = orc_kv_dequantize_fp8(m->kv_fp8, orc_kv_quantize_fp8(m->kv_fp8, vec[i]));
  }
  for (size_t h = 0; h < nkv; h++) {  // 577-600
    std::memcpy(kc + h * ms * d + pos * d, k.data() + h * d, d * 4);
    std::memcpy(vc + h * ms * d + pos * d, v.data() + h * d, d * 4);
  }
  std::vector<float> attn(nh * d);
  float scale = 1.0f / std::sqrt((float)d);  // layers.rs:374
  orc_attention_cached(q.data(), kc, vc, attn.data(), nh, nkv, d, ms, scale, pos + 1);  // 675-682
  return linear_forward(m, L.wo, attn.data(), out);  // 700-701
}

// FeedForward::forward (layers.rs:908-929)
int dense_ffn_forward(orc_model* m, const LayerRef& L, const float* x, float* out) {
  size_t ffn = L.gate.out_f;
  std::vector<float> g(ffn), u(ffn);
  if (linear_forward(m, L.gate, x, g.data())) return 1;
  if (linear_forward(m, L.up, x, u.data())) return 1;
  orc_silu_mul_inplace(g.data(), u.data(), ffn);
  return linear_forward(m, L.down, g.data(), out);
}

// MoeLayer::forward (moe.rs:321-413) with MoeExpert::forward (227-268); router normalize=false (loader.rs:1155-1159)
int moe_forward(orc_model* m, const LayerRef& L, const float* x, float* out) {
  const orc_config& c = m->cfg;
  size_t hidden = c.hidden_size, ne = c.num_experts, topk = c.num_experts_per_token, ei = c.expert_intermediate_size;
  std::vector<uint32_t> idx(topk);
  std::vector<float> wts(topk);
  orc_moe_route(x, (const float*)L.router->data, hidden, ne, topk, 0, idx.data(), wts.data());
  for (size_t i = 0; i < hidden; i++) out[i] = 0.0f;
  std::vector<float> g(ei), gs(ei), u(ei), inter(ei), eo(hidden);
  for (size_t s = 0; s < topk; s++) {
    const ExpertRef& E = L.experts[idx[s]];
    if (raw_vec_mat(m, E.gate_type, E.gate, x, g.data(), hidden, ei)) return 1;
    orc_silu(g.data(), gs.data(), ei);
    if (raw_vec_mat(m, E.up_type, E.up, x, u.data(), hidden, ei)) return 1;
    for (size_t i = 0; i < ei; i++) inter[i] = gs[i] * u[i];
    if (raw_vec_mat(m, E.down_type, E.down, inter.data(), eo.data(), ei, hidden)) return 1;
    for (size_t i = 0; i < hidden; i++) out[i] += wts[s] * eo[i];  // moe.rs:363-368
  }
  return 0;
}

// TransformerLayer::forward, serial-residual branch (layers.rs:1082-1245)
int layer_forward(orc_model* m, size_t li, std::vector<float>& hidden, size_t pos) {
  const orc_config& c = m->cfg;
  const LayerRef& L = m->layers[li];
  size_t hs = c.hidden_size;
  std::vector<float> norm(hs), h(hs), ffn_norm(hs), ffn_out(hs);
  orc_rms_norm(hidden.data(), (const float*)L.attn_norm->data, c.norm_eps, norm.data(), hs);  // 1095-1096
  if (attention_forward(m, li, norm.data(), pos, h.data())) return 1;
  for (size_t i = 0; i < hs; i++) h[i] += hidden[i];  // 1201-1208
  orc_rms_norm(h.data(), (const float*)L.ffn_norm->data, c.norm_eps, ffn_norm.data(), hs);  // 1210-1211
  int rc = L.router ? moe_forward(m, L, ffn_norm.data(), ffn_out.data())
                    : dense_ffn_forward(m, L, ffn_norm.data(), ffn_out.data());
  if (rc) return 1;
  for (size_t i = 0; i < hs; i++) ffn_out[i] += h[i];  // 1235-1241
  hidden.swap(ffn_out);
  return 0;
}

int dequant_embeddings(orc_model* m, std::vector<float>& dst) {  // llama.rs:181-193
  const TensorRef* t = m->token_embd;
  dst.resize(t->numel());
  if (t->type == ORC_F32) {
    std::memcpy(dst.data(), t->data, dst.size() * 4);
    return 0;
  }
  size_t bs = orc_block_size(t->type), bb = orc_block_bytes(t->type);
  if (!bs) return fail(m, "unsupported embedding dtype");
  size_t nblk = dst.size() / bs;
  int bad = 0;
  orc_parallel_for(nblk, 4096, [&](size_t b0, size_t b1) {
    if (orc_dequantize(t->type, t->data + b0 * bb, (b1 - b0) * bs, dst.data() + b0 * bs)) bad = 1;
  });
  return bad ? fail(m, "embedding dequantize failed") : 0;
}

}  // namespace

extern "C" {

orc_model* orc_model_create(const orc_config* cfg) {
  orc_model* m = new orc_model();
  m->cfg = *cfg;
  return m;
}

void orc_model_destroy(orc_model* m) { delete m; }

const char* orc_model_last_error(const orc_model* m) { return m->err.c_str(); }

int orc_model_add_tensor(orc_model* m, const char* name, int type, const uint64_t ne[4], const void* data,
                         size_t nbytes, int borrow) {
  if (m->finalized) return fail(m, "add_tensor after finalize");
  size_t bs = orc_block_size(type), bb = orc_block_bytes(type);
  if (!bs) return fail(m, std::string("unsupported dtype for ") + name);
  TensorRef t;
  t.type = type;
  for (int i = 0; i < 4; i++) t.ne[i] = ne[i];
  size_t numel = t.numel();
  if (ne[0] % bs || numel / bs * bb != nbytes) return fail(m, std::string("byte size mismatch for ") + name);
  t.nbytes = nbytes;
  TensorRef& slot = m->tensors[name];
  slot = std::move(t);
  if (borrow) {
    slot.data = (const uint8_t*)data;
  } else {
    slot.owned.assign((const uint8_t*)data, (const uint8_t*)data + nbytes);
    slot.data = slot.owned.data();
  }
  return 0;
}

int orc_model_finalize(orc_model* m) {
  const orc_config& c = m->cfg;
  m->token_embd = find(m, "token_embd.weight");
  if (!m->token_embd) return fail(m, "missing token_embd.weight");
  m->output_norm = find(m, "output_norm.weight");
  if (!m->output_norm) return fail(m, "missing output_norm.weight");
  // tied output: reuse token_embd.weight when output.weight is absent (loader.rs:348-355)
  m->output.w = find(m, "output.weight");
  if (!m->output.w) m->output.w = m->token_embd;
  m->output.bias = nullptr;
  m->output.in_f = (size_t)m->output.w->ne[0];
  m->output.out_f = (size_t)m->output.w->ne[1];
  m->layers.assign(c.num_layers, LayerRef());
  for (uint32_t i = 0; i < c.num_layers; i++) {
    LayerRef& L = m->layers[i];
    std::string p = "blk." + std::to_string(i) + ".";
    L.attn_norm = find(m, p + "attn_norm.weight");
    L.ffn_norm = find(m, p + "ffn_norm.weight");
    if (!L.attn_norm || !L.ffn_norm) return fail(m, "missing norm weights in " + p);
    if (bind_linear(m, L.wq, p + "attn_q", true) || bind_linear(m, L.wk, p + "attn_k", true) ||
        bind_linear(m, L.wv, p + "attn_v", true) || bind_linear(m, L.wo, p + "attn_output", true))
      return 1;
    L.router = find(m, p + "ffn_gate_inp.weight");
    if (L.router) {
      if (L.router->type != ORC_F32) return fail(m, "router weight must be F32 (moe.rs:132-135)");
      const TensorRef* ge = find(m, p + "ffn_gate_exps.weight");
      const TensorRef* ue = find(m, p + "ffn_up_exps.weight");
      const TensorRef* de = find(m, p + "ffn_down_exps.weight");
      if (!ge || !ue || !de) return fail(m, "missing expert stacks in " + p);
      // 3-D [in, out, n_expert], expert outermost: expert e = e-th contiguous slice (loader.rs:1256-1303)
      L.experts.resize(c.num_experts);
      size_t gs = ge->nbytes / c.num_experts, us = ue->nbytes / c.num_experts, ds = de->nbytes / c.num_experts;
      for (uint32_t e = 0; e < c.num_experts; e++) {
        L.experts[e] = ExpertRef{ge->data + e * gs, ue->data + e * us, de->data + e * ds, ge->type, ue->type, de->type};
      }
    } else {
      if (bind_linear(m, L.gate, p + "ffn_gate", true) || bind_linear(m, L.up, p + "ffn_up", true) ||
          bind_linear(m, L.down, p + "ffn_down", true))
        return 1;
    }
  }
  size_t kv = (size_t)c.num_kv_heads * c.max_seq_len * c.head_dim;
  m->k_cache.assign(c.num_layers, std::vector<float>());
  m->v_cache.assign(c.num_layers, std::vector<float>());
  for (uint32_t i = 0; i < c.num_layers; i++) {
    m->k_cache[i].assign(kv, 0.0f);
    m->v_cache[i].assign(kv, 0.0f);
  }
  m->last_hidden.assign(c.hidden_size, 0.0f);
  m->finalized = true;
  return 0;
}

void orc_model_reset(orc_model* m) { m->position = 0; }  // KVCache::reset (model/mod.rs:110-117): O(1)
size_t orc_model_position(const orc_model* m) { return m->position; }

// KVCacheType::TurboQuantMSE { bits } for this model: `signs` = [layers][kv heads][2][padded head_dim] (+1 / -1), the sign vectors
// the reference would draw per (layer, head, k / v) engine (kv_turboquant.rs:44-71); bits 0 switches back to the f32 cache
int orc_model_set_kv_turboquant(orc_model* m, int bits, const float* signs, size_t n_signs) {
  if (bits == 0) { m->kv_tq_bits = 0; return 0; }
  const orc_config& c = m->cfg;
  const size_t pd = orc_tq_padded_dim(c.head_dim);
  if ((bits != 2 && bits != 3) || !signs || n_signs != (size_t)c.num_layers * c.num_kv_heads * 2 * pd) return 1;
  for (size_t i = 0; i < n_signs; i++)
    if (signs[i] != 1.0f && signs[i] != -1.0f) return 1;
  m->kv_tq_bits = bits;
  m->kv_int8 = false;
  m->kv_fp8 = 0;
  m->tq_signs.assign(signs, signs + n_signs);
  const size_t pb = orc_tq_packed_bytes(bits, pd);
  m->tq_k.assign(c.num_layers, std::vector<uint8_t>((size_t)c.num_kv_heads * c.max_seq_len * pb));
  m->tq_v = m->tq_k;
  m->tq_qjl.clear();
  return 0;
}

// KVCacheType::TurboQuantProd { bits }: after orc_model_set_kv_turboquant(bits, signs), the K engines' QJL projection matrices
// [layers][kv heads][padded_dim][padded_dim] (what QjlProjector draws from seed 4 base + 1, kv_turboquant.rs:55-58); NULL / 0 = none
int orc_model_set_kv_turboquant_qjl(orc_model* m, const float* qjl, size_t n_qjl) {
  if (!qjl || n_qjl == 0) { m->tq_qjl.clear(); return 0; }
  const orc_config& c = m->cfg;
  const size_t pd = orc_tq_padded_dim(c.head_dim);
  if (!m->kv_tq_bits || n_qjl != (size_t)c.num_layers * c.num_kv_heads * pd * pd) return 1;
  m->tq_qjl.assign(qjl, qjl + n_qjl);
  m->tq_kbits.assign(c.num_layers, std::vector<uint64_t>((size_t)c.num_kv_heads * c.max_seq_len * ((pd + 63) / 64)));
  m->tq_knorm.assign(c.num_layers, std::vector<float>((size_t)c.num_kv_heads * c.max_seq_len));
  return 0;
}

void orc_model_set_kv_int8(orc_model* m, int on) { m->kv_int8 = on != 0; if (on) m->kv_fp8 = 0; }

// quantize_int8 / dequantize_int8 (src/model/kv_quantized.rs:385-410): symmetric, scale = max|x| / 127 (1 when the row is
// all ~zero), q = round(x / scale) (f32::round: half away from zero) clamped to [-128, 127]
void orc_kv_quantize_int8(const float* x, size_t n, int8_t* q, float* scale) {
  float max_abs = 0.0f;
  for (size_t i = 0; i < n; i++) max_abs = std::fmax(max_abs, std::fabs(x[i]));
  const float sc = max_abs > 1e-10f ? max_abs / 127.0f : 1.0f;
  for (size_t i = 0; i < n; i++) {
    float r = std::round(x[i] / sc);
    r = r < -128.0f ? -128.0f : (r > 127.0f ? 127.0f : r);
    q[i] = (int8_t)r;
  }
  *scale = sc;
}

void orc_kv_dequantize_int8(const int8_t* q, float scale, size_t n, float* out) {
  for (size_t i = 0; i < n; i++) out[i] = (float)q[i] * scale;
}

void orc_model_set_kv_fp8(orc_model* m, int fmt) { m->kv_fp8 = (fmt == 1 || fmt == 2) ? fmt : 0; m->kv_int8 = false; }

// quantize_fp8_e4m3 / quantize_fp8_e5m2 (src/model/kv_quantized.rs:413-449, 492-528) as written there: the f32 mantissa is
// TRUNCATED (no rounding), magnitudes past the largest exponent saturate to 0x7E / 0x7C, and an E4M3 value whose exponent
// field is 15 and whose three kept mantissa bits are all ones (|x| in [480, 512)) comes out as 0x7F — the NaN pattern.
// fmt: 1 = E4M3 (bias 7, 3 mantissa bits), 2 = E5M2 (bias 15, 2 mantissa bits).
uint8_t orc_kv_quantize_fp8(int fmt, float value) {
  const bool e4 = fmt == 1;
  if (std::isnan(value)) return 0xFF;
  if (std::isinf(value)) return e4 ? (value > 0.0f ? 0x7F : 0xFF) : (value > 0.0f ? 0x7C : 0xFC);
  if (value == 0.0f) return 0x00;
  uint32_t bits;
  std::memcpy(&bits, &value, 4);
  const uint8_t sign = (uint8_t)((bits >> 31) & 1u);
  const int exponent = (int)((bits >> 23) & 0xFFu) - 127;
  uint32_t mantissa = bits & 0x7FFFFFu;
  if (exponent != -127) mantissa |= 0x800000u;
  if (e4) {
    const int e = exponent + 7;
    if (e > 15) return (uint8_t)((sign << 7) | 0x7E);
    if (e > -3 && e <= 0) {
      const uint32_t shift_bits = (uint32_t)(3 + e);
      const uint32_t mask = 0x7u >> (uint32_t)(-e);
      return (uint8_t)((sign << 7) | (uint8_t)((mantissa >> (24 - shift_bits)) & mask));
    }
    if (e <= -3) return (uint8_t)(sign << 7);
    return (uint8_t)((sign << 7) | ((uint8_t)e << 3) | (uint8_t)((mantissa >> 20) & 0x7u));
  }
  const int e = exponent + 15;
  if (e > 31) return (uint8_t)((sign << 7) | 0x7C);
  if (e >= -1 && e <= 0) {
    const uint32_t shift_bits = (uint32_t)(2 + e);
    const uint32_t mask = 0x3u >> (uint32_t)(-e);
    return (uint8_t)((sign << 7) | (uint8_t)((mantissa >> (24 - shift_bits)) & mask));
  }
  if (e < -1) return (uint8_t)(sign << 7);
  return (uint8_t)((sign << 7) | ((uint8_t)e << 2) | (uint8_t)((mantissa >> 21) & 0x3u));
}

// dequantize_fp8_e4m3 / dequantize_fp8_e5m2 (kv_quantized.rs:452-489, 531-565), case by case as written there
float orc_kv_dequantize_fp8(int fmt, uint8_t bits) {
  const bool e4 = fmt == 1;
  auto from_bits = [](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
  if ((bits & 0x7F) == 0) return 0.0f;
  const uint32_t sign = (bits >> 7) & 1u;
  if (e4) {
    if ((bits & 0x7F) == 0x7F) return std::nanf("");
    const uint32_t e = (bits >> 3) & 0xFu, mt = bits & 0x7u;
    const uint32_t float_exp = (uint32_t)((int)e - 7 + 127);
    if (e > 0) return from_bits(sign << 31 | float_exp << 23 | mt << 20);
    if (mt >= 4) return from_bits(sign << 31 | float_exp << 23 | (mt & 3u) << 21);
    if (mt > 1) return from_bits(sign << 31 | (float_exp - 1) << 23 | (mt & 1u) << 22);
    return from_bits(sign << 31 | (float_exp - 2) << 23);   // mt == 1 (mt == 0 is the zero above)
  }
  if ((bits & 0x7F) == 0x7C) return sign ? -INFINITY : INFINITY;
  if ((bits & 0x7F) >= 0x7D) return std::nanf("");
  const uint32_t e = (bits >> 2) & 0x1Fu, mt = bits & 0x3u;
  const uint32_t float_exp = (uint32_t)((int)e - 15 + 127);
  if (e > 0) return from_bits(sign << 31 | float_exp << 23 | mt << 21);
  if (mt >= 2) return from_bits(sign << 31 | float_exp << 23 | (mt & 1u) << 22);
  return from_bits(sign << 31 | (float_exp - 1) << 23);     // mt == 1
}

void orc_model_kv_truncate(orc_model* m, size_t new_len) {  // KVCache::truncate (model/mod.rs:130-134)
  if (new_len < m->position) m->position = new_len;
}

void orc_model_kv_shift_left(orc_model* m, size_t amount) {  // KVCache::shift_left (model/mod.rs:142-172) + engine.rs:1407-1408
  // NB: a shift by 0 CLEARS the cache, exactly as a shift by >= seq_len does (model/mod.rs:143-146)
  if (amount == 0 || amount >= m->position) { m->position = 0; return; }
  const orc_config& c = m->cfg;
  size_t new_len = m->position - amount, row_stride = (size_t)c.max_seq_len * c.head_dim, n = new_len * c.head_dim;
  for (size_t li = 0; li < m->k_cache.size(); li++)
    for (auto* cache : {&m->k_cache[li], &m->v_cache[li]})
      for (size_t h = 0; h < c.num_kv_heads; h++)
        std::memmove(cache->data() + h * row_stride, cache->data() + h * row_stride + amount * c.head_dim, n * sizeof(float));
  if (m->kv_tq_bits) {   // TurboQuantKVCache::shift_left (kv_turboquant.rs:245-266): the first `amount` entries of every head are drained
    const size_t pb = orc_tq_packed_bytes(m->kv_tq_bits, orc_tq_padded_dim(c.head_dim)), rs = (size_t)c.max_seq_len * pb;
    for (size_t li = 0; li < m->tq_k.size(); li++)
      for (auto* cache : {&m->tq_k[li], &m->tq_v[li]})
        for (size_t h = 0; h < c.num_kv_heads; h++) std::memmove(cache->data() + h * rs, cache->data() + h * rs + amount * pb, new_len * pb);
    if (!m->tq_qjl.empty()) {
      const size_t nw = (orc_tq_padded_dim(c.head_dim) + 63) / 64;
      for (size_t li = 0; li < m->tq_kbits.size(); li++)
        for (size_t h = 0; h < c.num_kv_heads; h++) {
          uint64_t* b = m->tq_kbits[li].data() + h * c.max_seq_len * nw;
          std::memmove(b, b + amount * nw, new_len * nw * 8);
          float* nn = m->tq_knorm[li].data() + h * c.max_seq_len;
          std::memmove(nn, nn + amount, new_len * 4);
        }
    }
  }
  m->position = new_len;   // the rows keep the RoPE rotation of their OLD positions (the reference does not re-rotate)
}

int orc_model_forward(orc_model* m, const uint32_t* tokens, size_t n_tokens, float* logits, int faithful_embedding) {
  if (!m->finalized) return fail(m, "forward before finalize");
  const orc_config& c = m->cfg;
  if (n_tokens == 0) return fail(m, "No tokens to process");
  if (m->position + n_tokens > c.max_seq_len) return fail(m, "context length exceeded");  // llama.rs:280-286
  size_t hs = c.hidden_size;
  std::vector<float> scratch;
  const std::vector<float>* table = &m->embd_f32;
  if (faithful_embedding) {  // llama.rs:288: whole table, every call
    if (dequant_embeddings(m, scratch)) return 1;
    table = &scratch;
  } else if (m->embd_f32.empty()) {
    if (dequant_embeddings(m, m->embd_f32)) return 1;
  }
  std::vector<std::vector<float>> hiddens(n_tokens);
  for (size_t t = 0; t < n_tokens; t++) {  // llama.rs:293-307
    if (tokens[t] >= c.vocab_size) return fail(m, "token id exceeds vocab size");
    const float* src = table->data() + (size_t)tokens[t] * hs;
    hiddens[t].assign(src, src + hs);
  }
  for (uint32_t li = 0; li < c.num_layers; li++)  // llama.rs:327-345: layer-major over the tokens
    for (size_t t = 0; t < n_tokens; t++)
      if (layer_forward(m, li, hiddens[t], m->position + t)) return 1;
  m->position += n_tokens;
  m->last_hidden = hiddens.back();
  // compute_logits (llama.rs:247-266)
  std::vector<float> normed(hs);
  orc_rms_norm(hiddens.back().data(), (const float*)m->output_norm->data, c.norm_eps, normed.data(), hs);
  return linear_forward(m, m->output, normed.data(), logits);
}

int orc_model_last_hidden(const orc_model* m, float* out) {
  std::memcpy(out, m->last_hidden.data(), m->last_hidden.size() * 4);
  return 0;
}

}  // extern "C"
