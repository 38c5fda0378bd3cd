// gguf_loader.cpp — GGUF -> HBM direct loader.
//
// Replaces, for this engine, the reference's GgufReader + ModelLoader + from_model chain
// (src/gguf/reader.rs:49-104 header / metadata / tensor-info parsing, alignment 84-96; src/model/loader.rs:62-170
// ModelConfig from `{arch}.*` keys; src/backend/cuda/dequant_weights.rs:244-505 upload): the file is mapped, the
// header parsed in place, and every tensor the engine knows is handed to lgh_upload_tensor straight from the mapping
// in native GGUF layout — no host Vec<u8> copy (loader.rs:1356-1359) and no host transposition
// (dequant_weights.rs:143-153).  Host-only code: no HIP here, it drives the C ABI of the same library.
//
// The file is untrusted input: every read is bounds-checked against the mapping, counts and lengths are capped.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/llama_gguf_hip.h"

namespace {

constexpr uint32_t kMagic = 0x46554747u;   // "GGUF" little-endian (src/gguf/constants.rs:4)
constexpr uint64_t kMaxCount = 1u << 24, kMaxString = 1u << 26;

enum VT : uint32_t { U8 = 0, I8, U16, I16, U32, I32, F32, BOOL, STR, ARR, U64, I64, F64 };   // constants.rs:17-31

struct Value { uint32_t type = 0; uint64_t u = 0; double f = 0.0; std::string s; uint64_t arr_len = 0; };
struct TInfo { std::string name; uint32_t n_dims = 0; uint64_t ne[4] = {1, 1, 1, 1}; uint32_t type = 0; uint64_t offset = 0; };

struct Cursor {
  const uint8_t* p;
  uint64_t n, pos = 0;
  bool ok = true;
  bool need(uint64_t k) { if (!ok || k > n - pos) { ok = false; return false; } return true; }
  template <class T> T rd() { T v{}; if (need(sizeof(T))) { std::memcpy(&v, p + pos, sizeof(T)); pos += sizeof(T); } return v; }
  uint64_t len(uint32_t version) { return version == 1 ? rd<uint32_t>() : rd<uint64_t>(); }
  std::string str(uint32_t version) {
    const uint64_t l = len(version);
    if (l > kMaxString || !need(l)) { ok = false; return {}; }
    std::string s(reinterpret_cast<const char*>(p + pos), (size_t)l);
    pos += l;
    return s;
  }
};

uint64_t scalar_size(uint32_t t) {
  switch (t) { case U8: case I8: case BOOL: return 1; case U16: case I16: return 2; case U32: case I32: case F32: return 4;
               case U64: case I64: case F64: return 8; default: return 0; }
}

bool read_value(Cursor& c, uint32_t version, Value& v, int depth = 0) {
  v.type = c.rd<uint32_t>();
  switch (v.type) {
    case U8: v.u = c.rd<uint8_t>(); break;
    case I8: v.u = (uint64_t)(int64_t)c.rd<int8_t>(); break;
    case U16: v.u = c.rd<uint16_t>(); break;
    case I16: v.u = (uint64_t)(int64_t)c.rd<int16_t>(); break;
    case U32: v.u = c.rd<uint32_t>(); break;
    case I32: v.u = (uint64_t)(int64_t)c.rd<int32_t>(); break;
    case F32: v.f = c.rd<float>(); break;
    case BOOL: v.u = c.rd<uint8_t>() != 0; break;
    case STR: v.s = c.str(version); break;
    case U64: v.u = c.rd<uint64_t>(); break;
    case I64: v.u = (uint64_t)c.rd<int64_t>(); break;
    case F64: v.f = c.rd<double>(); break;
    case ARR: {   // skipped element by element (only its length is kept: tokenizer.ggml.tokens -> vocab size)
      if (depth > 2) { c.ok = false; return false; }
      const uint32_t et = c.rd<uint32_t>();
      const uint64_t cnt = c.len(version);
      if (cnt > (1ull << 32)) { c.ok = false; return false; }
      v.arr_len = cnt;
      if (const uint64_t ss = scalar_size(et)) {
        if (!c.need(cnt * ss)) return false;
        c.pos += cnt * ss;
      } else if (et == STR) {
        for (uint64_t i = 0; i < cnt && c.ok; i++) {
          const uint64_t l = c.len(version);
          if (l > kMaxString || !c.need(l)) { c.ok = false; return false; }
          c.pos += l;
        }
      } else if (et == ARR) {   // arrays of scalar arrays
        for (uint64_t i = 0; i < cnt && c.ok; i++) {
          const uint32_t et2 = c.rd<uint32_t>();
          const uint64_t cnt2 = c.len(version), s2 = scalar_size(et2);
          if (!s2 || cnt2 > (1ull << 32) || !c.need(cnt2 * s2)) { c.ok = false; return false; }
          c.pos += cnt2 * s2;
        }
      } else { c.ok = false; return false; }
      break;
    }
    default: c.ok = false; return false;
  }
  return c.ok;
}

struct Parsed {
  uint32_t version = 0, alignment = 32;
  uint64_t data_offset = 0;
  std::map<std::string, Value> kv;
  std::vector<TInfo> tensors;
};

struct Mapping {
  int fd = -1; const uint8_t* p = nullptr; uint64_t n = 0;
  ~Mapping() { if (p) munmap((void*)p, n); if (fd >= 0) close(fd); }
};

void set_err(char* err, size_t errlen, const std::string& m) {
  if (err && errlen) { std::snprintf(err, errlen, "%s", m.c_str()); }
}

int open_map(const char* path, Mapping& m, std::string& why) {
  m.fd = open(path, O_RDONLY);
  if (m.fd < 0) { why = std::string("cannot open ") + path; return LGH_INVALID_ARGUMENT; }
  struct stat st;
  if (fstat(m.fd, &st) != 0 || st.st_size < 1) { why = "Unexpected end of file"; return LGH_INVALID_ARGUMENT; }   // GgufError::UnexpectedEof
  m.n = (uint64_t)st.st_size;
  void* p = mmap(nullptr, m.n, PROT_READ, MAP_PRIVATE, m.fd, 0);
  if (p == MAP_FAILED) { why = "mmap failed"; return LGH_ALLOCATION_FAILED; }
  m.p = (const uint8_t*)p;
  return LGH_OK;
}

int parse(const Mapping& m, Parsed& g, std::string& why) {
  // error vocabulary of the reference's reader (src/gguf/error.rs:2-19; raised at reader.rs:24-47 for the header)
  Cursor c{m.p, m.n};
  const uint32_t magic = c.rd<uint32_t>();
  if (!c.ok) { why = "Unexpected end of file"; return LGH_INVALID_ARGUMENT; }
  if (magic != kMagic) {
    char b[80];
    std::snprintf(b, sizeof(b), "Invalid magic number: expected 0x46554747, got 0x%08X", magic);
    why = b;
    return LGH_INVALID_ARGUMENT;
  }
  g.version = c.rd<uint32_t>();
  if (!c.ok) { why = "Unexpected end of file"; return LGH_INVALID_ARGUMENT; }
  if (g.version < 1 || g.version > 3) { why = "Unsupported GGUF version: " + std::to_string(g.version); return LGH_UNSUPPORTED; }
  const uint64_t n_tensors = c.len(g.version), n_kv = c.len(g.version);   // reader.rs:52-63: u32 counts in v1, u64 after
  if (!c.ok) { why = "Unexpected end of file"; return LGH_INVALID_ARGUMENT; }
  if (n_tensors > kMaxCount || n_kv > kMaxCount) { why = "Invalid data: implausible tensor / metadata counts"; return LGH_INVALID_ARGUMENT; }
  for (uint64_t i = 0; i < n_kv; i++) {
    std::string key = c.str(g.version);
    Value v;
    if (!c.ok || !read_value(c, g.version, v)) { why = "truncated or malformed metadata at entry " + std::to_string(i); return LGH_INVALID_ARGUMENT; }
    g.kv[key] = v;
  }
  g.tensors.resize((size_t)n_tensors);
  for (auto& t : g.tensors) {
    t.name = c.str(g.version);
    t.n_dims = c.rd<uint32_t>();
    if (!c.ok || t.n_dims > 4) { why = "malformed tensor info"; return LGH_INVALID_ARGUMENT; }
    for (uint32_t d = 0; d < t.n_dims; d++) t.ne[d] = g.version == 1 ? c.rd<uint32_t>() : c.rd<uint64_t>();
    t.type = c.rd<uint32_t>();
    t.offset = c.rd<uint64_t>();
    if (!c.ok) { why = "truncated tensor info"; return LGH_INVALID_ARGUMENT; }
  }
  auto a = g.kv.find("general.alignment");   // reader.rs:84-92
  if (a != g.kv.end() && (a->second.type == U32 || a->second.type == U64) && a->second.u >= 1 && a->second.u <= (1u << 20)) g.alignment = (uint32_t)a->second.u;
  g.data_offset = (c.pos + g.alignment - 1) / g.alignment * g.alignment;
  // (a file without tensors may end before its aligned data offset: reader.rs:84-96 only computes the offset)
  if (g.data_offset > m.n && !g.tensors.empty()) { why = "Invalid data: data section beyond the end of the file"; return LGH_INVALID_ARGUMENT; }
  return LGH_OK;
}

uint32_t blk_elems_of(uint32_t t) {
  switch (t) { case LGH_TYPE_F32: case LGH_TYPE_F16: case LGH_TYPE_BF16: return 1; case LGH_TYPE_Q4_0: case LGH_TYPE_Q4_1: case LGH_TYPE_Q5_0:
               case LGH_TYPE_Q5_1: case LGH_TYPE_Q8_0: case LGH_TYPE_Q8_1: return 32; case LGH_TYPE_Q2_K: case LGH_TYPE_Q3_K: case LGH_TYPE_Q4_K:
               case LGH_TYPE_Q5_K: case LGH_TYPE_Q6_K: case LGH_TYPE_Q8_K: return 256; default: return 0; }
}
uint32_t blk_bytes_of(uint32_t t) {   // src/tensor/quant/blocks.rs:293-305
  switch (t) { case LGH_TYPE_F32: return 4; case LGH_TYPE_F16: case LGH_TYPE_BF16: return 2; case LGH_TYPE_Q4_0: return 18; case LGH_TYPE_Q4_1: return 20;
               case LGH_TYPE_Q5_0: return 22; case LGH_TYPE_Q5_1: return 24; case LGH_TYPE_Q8_0: return 34; case LGH_TYPE_Q8_1: return 36;
               case LGH_TYPE_Q2_K: return 84; case LGH_TYPE_Q3_K: return 110; case LGH_TYPE_Q4_K: return 144; case LGH_TYPE_Q5_K: return 176;
               case LGH_TYPE_Q6_K: return 210; case LGH_TYPE_Q8_K: return 292; default: return 0; }
}

bool get_u(const Parsed& g, const std::string& k, uint64_t& out) {
  auto it = g.kv.find(k);
  if (it == g.kv.end()) return false;
  const Value& v = it->second;
  if (v.type == F32 || v.type == F64 || v.type == STR || v.type == ARR) return false;
  out = v.u;
  return true;
}
bool get_f(const Parsed& g, const std::string& k, double& out) {
  auto it = g.kv.find(k);
  if (it == g.kv.end() || (it->second.type != F32 && it->second.type != F64)) return false;
  out = it->second.f;
  return true;
}
const TInfo* find_tensor(const Parsed& g, const std::string& n) {
  for (auto& t : g.tensors) if (t.name == n) return &t;
  return nullptr;
}

// ModelConfig from the `{arch}.*` keys, as ModelLoader::parse_config does (loader.rs:62-170)
int describe(const Parsed& g, lgh_gguf_info& info, std::string& why) {
  std::memset(&info, 0, sizeof(info));
  info.version = g.version;
  info.n_tensors = g.tensors.size();
  info.n_kv = g.kv.size();
  info.alignment = g.alignment;
  info.data_offset = g.data_offset;
  auto ai = g.kv.find("general.architecture");
  if (ai == g.kv.end() || ai->second.type != STR) { why = "general.architecture missing"; return LGH_INVALID_ARGUMENT; }   // (header fields above stay filled)
  const std::string arch = ai->second.s;
  std::snprintf(info.architecture, sizeof(info.architecture), "%s", arch.c_str());
  lgh_model_desc& d = info.desc;
  d.struct_size = sizeof(lgh_model_desc);
  uint64_t u = 0;
  double f = 0.0;
  auto need = [&](const char* key, uint32_t& dst) {
    if (!get_u(g, arch + "." + key, u) || u == 0 || u > 0xFFFFFFFFull) { why = arch + "." + key + " missing"; return false; }
    dst = (uint32_t)u;
    return true;
  };
  if (!need("embedding_length", d.hidden_size) || !need("block_count", d.num_layers) || !need("attention.head_count", d.num_heads)) return LGH_INVALID_ARGUMENT;
  d.num_kv_heads = get_u(g, arch + ".attention.head_count_kv", u) && u ? (uint32_t)u : d.num_heads;
  d.head_dim = get_u(g, arch + ".attention.key_length", u) && u ? (uint32_t)u : d.hidden_size / d.num_heads;
  d.intermediate_size = get_u(g, arch + ".feed_forward_length", u) && u ? (uint32_t)u : d.hidden_size * 4 * 2 / 3;
  d.norm_eps = get_f(g, arch + ".attention.layer_norm_rms_epsilon", f) || get_f(g, arch + ".attention.layer_norm_epsilon", f) ? (float)f : 1e-5f;
  d.rope_freq_base = get_f(g, arch + ".rope.freq_base", f) ? (float)f : 10000.0f;
  d.rope_freq_scale = get_f(g, arch + ".rope.scale_linear", f) && f != 0.0 ? (float)f : 1.0f;
  d.num_experts = get_u(g, arch + ".expert_count", u) ? (uint32_t)u : 0;
  d.num_experts_per_token = get_u(g, arch + ".expert_used_count", u) ? (uint32_t)u : 0;
  d.expert_intermediate_size = get_u(g, arch + ".expert_feed_forward_length", u) ? (uint32_t)u : 0;
  d.max_seq_len = get_u(g, arch + ".context_length", u) && u ? (uint32_t)(u > 0xFFFFFFFFull ? 0xFFFFFFFFu : u) : 2048;
  // vocabulary: `{arch}.vocab_size`, tokenizer.ggml.vocab_size, the token list, or the embedding's shape (loader.rs:78-97)
  if (get_u(g, arch + ".vocab_size", u) && u) d.vocab_size = (uint32_t)u;
  else if (get_u(g, "tokenizer.ggml.vocab_size", u) && u) d.vocab_size = (uint32_t)u;
  else {
    auto ti = g.kv.find("tokenizer.ggml.tokens");
    const TInfo* emb = find_tensor(g, "token_embd.weight");
    if (ti != g.kv.end() && ti->second.type == ARR && ti->second.arr_len) d.vocab_size = (uint32_t)ti->second.arr_len;
    else if (emb && emb->n_dims == 2) d.vocab_size = (uint32_t)emb->ne[1];
    else d.vocab_size = 32000;
  }
  if (d.num_experts && !d.expert_intermediate_size) {   // from the expert stack's shape, as GpuOnlyInference does (gpu_only.rs:488-492)
    if (const TInfo* e = find_tensor(g, "blk.0.ffn_gate_exps.weight")) d.expert_intermediate_size = (uint32_t)e->ne[1];
    else d.expert_intermediate_size = d.intermediate_size;
  }
  // RopeType: NeoX pairing for the Qwen / Phi / NeoX families (loader.rs:145-162), interleaved pairs otherwise
  static const char* kNeox[] = {"qwen2", "qwen2moe", "qwen3", "qwen3moe", "gptneox", "falcon", "phi2", "phi3", "stablelm", "gptj"};
  for (const char* a : kNeox) if (arch == a) d.use_neox_rope = 1;
  return LGH_OK;
}

// the engine runs pre-norm RMSNorm + softmax attention + gated FFN / MoE; everything else is another path of the reference
bool arch_supported(const std::string& a) {
  static const char* ok[] = {"llama", "mistral", "mixtral", "qwen2", "qwen2moe"};
  for (const char* s : ok) if (a == s) return true;
  return false;
}

bool known_tensor(const std::string& n) {
  if (n == "token_embd.weight" || n == "output_norm.weight" || n == "output.weight") return true;
  if (n.compare(0, 4, "blk.") != 0) return false;
  const size_t dot = n.find('.', 4);
  if (dot == std::string::npos) return false;
  const std::string sub = n.substr(dot + 1);
  static const char* subs[] = {"attn_norm.weight", "ffn_norm.weight", "attn_q.weight", "attn_k.weight", "attn_v.weight", "attn_output.weight",
                               "attn_q.bias", "attn_k.bias", "attn_v.bias", "attn_output.bias", "ffn_gate.weight", "ffn_up.weight", "ffn_down.weight",
                               "ffn_gate_inp.weight", "ffn_gate_exps.weight", "ffn_up_exps.weight", "ffn_down_exps.weight"};
  for (const char* s : subs) if (sub == s) return true;
  return false;
}

}  // namespace

int engine_shape_check(const lgh_model_desc& d, std::string& why);   // engine.hip: what the kernels are built for

extern "C" {

int lgh_gguf_inspect(const char* path, lgh_gguf_info* out, char* err, size_t errlen) {
  if (!path || !out) return LGH_INVALID_ARGUMENT;
  Mapping m;
  Parsed g;
  std::string why;
  int rc = open_map(path, m, why);
  if (!rc) rc = parse(m, g, why);
  if (rc) { set_err(err, errlen, why); return rc; }
  // A well-formed GGUF that is not a model of this engine's families (the reference's reader accepts any metadata,
  // tests/gguf_reader_test.rs:4-44) still inspects fine: header fields filled, desc.struct_size == 0 says "no model config"
  // (ModelLoader::parse_config would be the one to complain, loader.rs:62-77 — here lgh_load_gguf).
  std::string model_why;
  if (describe(g, *out, model_why) != LGH_OK) {
    out->desc = lgh_model_desc{};
    set_err(err, errlen, model_why);
  }
  out->file_bytes = m.n;
  return LGH_OK;
}

/* GgufData::get_string / get_u32 / get_u64 / get_f32 / get_bool (src/gguf/types.rs:71-104): one metadata value by key. */
int lgh_gguf_get(const char* path, const char* key, lgh_gguf_value* out, char* err, size_t errlen) {
  if (!path || !key || !out) return LGH_INVALID_ARGUMENT;
  Mapping m;
  Parsed g;
  std::string why;
  int rc = open_map(path, m, why);
  if (!rc) rc = parse(m, g, why);
  if (rc) { set_err(err, errlen, why); return rc; }
  auto it = g.kv.find(key);
  if (it == g.kv.end()) { set_err(err, errlen, std::string("no metadata key ") + key); return LGH_INVALID_ARGUMENT; }
  std::memset(out, 0, sizeof(*out));
  const Value& v = it->second;
  out->type = v.type;
  out->u = v.u;
  out->f = v.f;
  out->arr_len = v.arr_len;
  std::snprintf(out->s, sizeof(out->s), "%s", v.s.c_str());
  return LGH_OK;
}

int lgh_load_gguf(const char* path, uint32_t max_seq_len, int device, uint32_t flags, uint32_t layer_begin, uint32_t layer_end,
                  lgh_ctx** out, char* err, size_t errlen) {
  if (!path || !out) return LGH_INVALID_ARGUMENT;
  *out = nullptr;
  Mapping m;
  Parsed g;
  lgh_gguf_info info;
  std::string why;
  int rc = open_map(path, m, why);
  if (!rc) rc = parse(m, g, why);
  if (!rc) rc = describe(g, info, why);   // ModelLoader::parse_config's required keys (loader.rs:62-77)
  if (!rc && !arch_supported(info.architecture)) { rc = LGH_UNSUPPORTED; why = std::string("architecture '") + info.architecture + "' is not on this engine's path"; }
  if (rc) { set_err(err, errlen, why); return rc; }
  lgh_model_desc d = info.desc;
  if (max_seq_len) d.max_seq_len = max_seq_len;
  d.device_id = device;
  d.flags = flags;
  d.layer_begin = layer_begin;
  d.layer_end = layer_end;
  lgh_ctx* c = nullptr;
  if ((rc = engine_shape_check(d, why))) { set_err(err, errlen, why); return rc; }   // names the shape the kernels are not built for
  if ((rc = lgh_create(&d, &c))) { set_err(err, errlen, "lgh_create failed (is a HIP device visible?)"); return rc; }
  for (const TInfo& t : g.tensors) {
    if (!known_tensor(t.name)) continue;   // rope_freqs, tokenizer tables, architectures' extras: not on this path
    const uint32_t be = blk_elems_of(t.type), bb = blk_bytes_of(t.type);
    uint64_t n_elems = 1;
    for (uint32_t i = 0; i < t.n_dims; i++) {
      if (t.ne[i] == 0 || n_elems > (1ull << 40) / t.ne[i]) { n_elems = 0; break; }
      n_elems *= t.ne[i];
    }
    if (!be || !n_elems || t.ne[0] % be) { rc = LGH_UNSUPPORTED_DTYPE; why = t.name + ": unsupported type or shape"; break; }
    const uint64_t nbytes = n_elems / be * bb, off = g.data_offset + t.offset;
    if (t.offset % g.alignment || off > m.n || nbytes > m.n - off) { rc = LGH_INVALID_ARGUMENT; why = t.name + ": data outside the file"; break; }
    if ((rc = lgh_upload_tensor(c, t.name.c_str(), t.type, t.ne, m.p + off, (size_t)nbytes))) { why = t.name + ": " + lgh_last_error(c); break; }
  }
  if (!rc && (rc = lgh_finalize(c))) why = lgh_last_error(c);
  if (rc) {
    set_err(err, errlen, why);
    lgh_destroy(c);
    return rc;
  }
  *out = c;
  return LGH_OK;
}

}  // extern "C"
