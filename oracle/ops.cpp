// oracle/ops.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference CPU backend's tensor ops on the decode path.
//
//   dot_f32 / axpy / max / softmax / sum_of_squares / rms_norm / silu_mul   src/backend/cpu/simd.rs
//   rms_norm, vec_mat, vec_mat_q, rope, attention_cached                   src/backend/cpu/ops.rs
//   MoE router                                                            src/model/moe.rs:128-198
//   greedy selection                                   src/sampling/mod.rs:224-242, src/main.rs:1815-1821
//
// The reference picks AVX-512 / AVX2 / scalar variants at run time (simd.rs:80-100); their lane
// structure changes the f32 summation order, so "the reference CPU result" depends on the host.
// The variants are restated here lane by lane in portable C++ (fmaf == one vfmadd lane), selected by
// orc_set_isa(); AUTO resolves to what the reference would choose on this host.
//
// Built with -ffp-contract=off: only the explicit fmaf() calls fuse, exactly where the reference
// uses _mm*_fmadd_ps.
#include "oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <limits>
#include <mutex>
#include <thread>
#include <vector>

// ------------------------------------------------------------------ ISA selection
static int g_isa = ORC_ISA_AUTO;

static int resolve_isa() {
  if (g_isa != ORC_ISA_AUTO) return g_isa;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("avx512f")) return ORC_ISA_AVX512;
  if (__builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma")) return ORC_ISA_AVX2;
#endif
  return ORC_ISA_SCALAR;
}

// ------------------------------------------------------------------ thread pool (rayon stand-in)
// The reference parallelises with rayon one task per output element (ops.rs:1136-1180) or per head
// (ops.rs:1508); no task ever combines partial results of another, so scheduling cannot change a value.
namespace {
class Pool {
 public:
  explicit Pool(int n) : n_(n) {
    for (int i = 1; i < n_; i++) workers_.emplace_back([this, i] { loop(i); });
  }
  ~Pool() {
    {
      std::lock_guard<std::mutex> lk(mu_);
      stop_ = true;
      gen_++;
    }
    cv_.notify_all();
    for (auto& t : workers_) t.join();
  }
  int size() const { return n_; }
  // calls fn(begin, end) over [0, total) in chunks
  void run(size_t total, size_t grain, const std::function<void(size_t, size_t)>& fn) {
    if (n_ <= 1 || total <= grain) {
      fn(0, total);
      return;
    }
    {
      std::lock_guard<std::mutex> lk(mu_);
      fn_ = &fn;
      total_ = total;
      grain_ = grain;
      next_.store(0);
      pending_ = n_ - 1;
      gen_++;
    }
    cv_.notify_all();
    work();
    std::unique_lock<std::mutex> lk(mu_);
    done_cv_.wait(lk, [this] { return pending_ == 0; });
    fn_ = nullptr;
  }

 private:
  void work() {
    for (;;) {
      size_t b = next_.fetch_add(grain_);
      if (b >= total_) break;
      size_t e = std::min(total_, b + grain_);
      (*fn_)(b, e);
    }
  }
  void loop(int) {
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu_);
        cv_.wait(lk, [&] { return gen_ != seen; });
        seen = gen_;
        if (stop_) return;
      }
      work();
      {
        std::lock_guard<std::mutex> lk(mu_);
        if (--pending_ == 0) done_cv_.notify_one();
      }
    }
  }
  int n_;
  std::vector<std::thread> workers_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  const std::function<void(size_t, size_t)>* fn_ = nullptr;
  size_t total_ = 0, grain_ = 1;
  std::atomic<size_t> next_{0};
  int pending_ = 0;
  uint64_t gen_ = 0;
  bool stop_ = false;
};

Pool* g_pool = nullptr;
int g_threads = 1;
std::mutex g_pool_mu;

Pool* pool() {
  std::lock_guard<std::mutex> lk(g_pool_mu);
  if (!g_pool || g_pool->size() != g_threads) {
    delete g_pool;
    g_pool = new Pool(g_threads);
  }
  return g_pool;
}
}  // namespace

void orc_parallel_for(size_t total, size_t grain, const std::function<void(size_t, size_t)>& fn) {
  pool()->run(total, grain, fn);
}

// ------------------------------------------------------------------ simd.rs restated lane by lane
namespace {

// hsum_avx2 (simd.rs:199-212): (v[i+4]+v[i]) -> pairs -> final
inline float hsum8(const float v[8]) {
  float s[4];
  for (int i = 0; i < 4; i++) s[i] = v[i + 4] + v[i];
  float p0 = s[0] + s[1];
  float p2 = s[2] + s[3];
  return p0 + p2;
}

// _mm512_reduce_add_ps as stdarch lowers it: 16 -> 8 -> 4 -> 2 -> 1 by halves
inline float hsum16(const float v[16]) {
  float a[8], b[4];
  for (int i = 0; i < 8; i++) a[i] = v[i] + v[i + 8];
  for (int i = 0; i < 4; i++) b[i] = a[i] + a[i + 4];
  float c0 = b[0] + b[2], c1 = b[1] + b[3];
  return c0 + c1;
}

float dot_scalar(const float* a, const float* b, size_t n) {  // simd.rs:103-105
  float s = 0.0f;
  for (size_t i = 0; i < n; i++) s += a[i] * b[i];
  return s;
}

template <int L>
float dot_lanes(const float* a, const float* b, size_t n) {  // simd.rs:108-165
  float acc[L];
  for (int l = 0; l < L; l++) acc[l] = 0.0f;
  size_t chunks = n / L;
  for (size_t c = 0; c < chunks; c++)
    for (int l = 0; l < L; l++) acc[l] = std::fmaf(a[c * L + l], b[c * L + l], acc[l]);
  float r = (L == 8) ? hsum8(acc) : hsum16(acc);
  for (size_t i = chunks * L; i < n; i++) r += a[i] * b[i];
  return r;
}

}  // namespace

extern "C" {

void orc_set_isa(int isa) { g_isa = isa; }
int orc_get_isa(void) { return resolve_isa(); }
void orc_set_threads(int n) {
  if (n < 1) n = (int)std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  std::lock_guard<std::mutex> lk(g_pool_mu);
  g_threads = n;
}
int orc_get_threads(void) { return g_threads; }

float orc_dot_f32(const float* a, const float* b, size_t n) {  // simd.rs:80-100
  switch (resolve_isa()) {
    case ORC_ISA_AVX512: return dot_lanes<16>(a, b, n);
    case ORC_ISA_AVX2: return dot_lanes<8>(a, b, n);
    default: return dot_scalar(a, b, n);
  }
}

void orc_axpy_f32(float alpha, const float* x, float* y, size_t n) {  // simd.rs:356-427
  int isa = resolve_isa();
  size_t L = isa == ORC_ISA_AVX512 ? 16 : (isa == ORC_ISA_AVX2 ? 8 : 0);
  size_t done = 0;
  if (L) {
    done = (n / L) * L;
    for (size_t i = 0; i < done; i++) y[i] = std::fmaf(alpha, x[i], y[i]);
  }
  for (size_t i = done; i < n; i++) y[i] += alpha * x[i];
}

float orc_sum_f32(const float* a, size_t n) {  // simd.rs:449-487: AVX2 whenever the host has it (8 lanes + hsum_avx2, scalar tail), else sequential
  if (resolve_isa() == ORC_ISA_SCALAR) {
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) s += a[i];
    return s;
  }
  float lanes[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const size_t chunks = n / 8;
  for (size_t c = 0; c < chunks; c++)
    for (int l = 0; l < 8; l++) lanes[l] += a[c * 8 + l];
  float r = hsum8(lanes);
  for (size_t i = chunks * 8; i < n; i++) r += a[i];
  return r;
}

float orc_max_f32(const float* x, size_t n) {  // simd.rs:511-563 (order-independent for non-NaN)
  float m = -std::numeric_limits<float>::infinity();
  for (size_t i = 0; i < n; i++) m = x[i] > m ? x[i] : m;
  return m;
}

void orc_softmax_inplace(float* x, size_t n) {  // simd.rs:679-750
  if (n == 0) return;
  float mx = orc_max_f32(x, n);
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) {
    x[i] = std::exp(x[i] - mx);  // f32::exp -> libm expf
    sum += x[i];
  }
  float inv = 1.0f / sum;
  for (size_t i = 0; i < n; i++) x[i] *= inv;
}

void orc_silu(const float* x, float* out, size_t n) {  // ops.rs:303-325
  for (size_t i = 0; i < n; i++) out[i] = x[i] / (1.0f + std::exp(-x[i]));
}

void orc_silu_mul_inplace(float* gate, const float* up, size_t n) {  // simd.rs:598-649
  // every variant computes (x / (1 + exp(-x))) * up[i] with the same two roundings
  for (size_t i = 0; i < n; i++) {
    float x = gate[i];
    gate[i] = x / (1.0f + std::exp(-x)) * up[i];
  }
}

static float sum_of_squares(const float* x, size_t n) {  // simd.rs:785-823
  int isa = resolve_isa();
  if (isa == ORC_ISA_AVX2 || isa == ORC_ISA_AVX512) {  // has_avx2() is true on AVX-512 hosts too
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t chunks = n / 8;
    for (size_t c = 0; c < chunks; c++)
      for (int l = 0; l < 8; l++) acc[l] = std::fmaf(x[c * 8 + l], x[c * 8 + l], acc[l]);
    float r = hsum8(acc);
    for (size_t i = chunks * 8; i < n; i++) r += x[i] * x[i];
    return r;
  }
  float s = 0.0f;
  for (size_t i = 0; i < n; i++) s += x[i] * x[i];
  return s;
}

void orc_rms_norm(const float* x, const float* w, float eps, float* out, size_t n) {  // simd.rs:847-878, ops.rs:392-422
  float ss = sum_of_squares(x, n);
  float rms = std::sqrt(ss / (float)n + eps);
  float inv_rms = 1.0f / rms;
  for (size_t i = 0; i < n; i++) out[i] = x[i] * inv_rms * w[i];
}

static void rope_tensor(float* data, size_t num_heads, size_t seq_len, size_t head_dim, size_t pos,
                        float freq_base, float freq_scale, int use_neox) {  // ops.rs:1285-1337
  size_t half = head_dim / 2;
  for (size_t head = 0; head < num_heads; head++) {
    for (size_t s = 0; s < seq_len; s++) {
      float position = (float)(pos + s) / freq_scale;
      size_t ho = head * seq_len * head_dim + s * head_dim;
      for (size_t i = 0; i < half; i++) {
        float freq = 1.0f / std::pow(freq_base, (float)(2 * i) / (float)head_dim);  // f32::powf -> libm powf
        float theta = position * freq;
        float c = std::cos(theta), sn = std::sin(theta);
        size_t i0 = use_neox ? ho + i : ho + 2 * i;
        size_t i1 = use_neox ? ho + i + half : ho + 2 * i + 1;
        float x0 = data[i0], x1 = data[i1];
        data[i0] = x0 * c - x1 * sn;
        data[i1] = x0 * sn + x1 * c;
      }
    }
  }
}

void orc_rope(float* q, float* k, size_t n_heads, size_t n_kv_heads, size_t seq_len, size_t head_dim,
              size_t pos, float freq_base, float freq_scale, int use_neox) {  // ops.rs:1216-1273
  rope_tensor(q, n_heads, seq_len, head_dim, pos, freq_base, freq_scale, use_neox);
  rope_tensor(k, n_kv_heads, seq_len, head_dim, pos, freq_base, freq_scale, use_neox);
}

void orc_attention_cached(const float* q, const float* k_cache, const float* v_cache, float* out,
                          size_t n_heads, size_t n_kv_heads, size_t head_dim, size_t max_seq_len,
                          float scale, size_t kv_len) {  // ops.rs:1479-1537
  size_t per_kv = n_heads / n_kv_heads;
  size_t head_stride = max_seq_len * head_dim;
  orc_parallel_for(n_heads, 1, [&](size_t h0, size_t h1) {
    std::vector<float> scores(kv_len);
    for (size_t head = h0; head < h1; head++) {
      size_t kv_head = head / per_kv;
      const float* qv = q + head * head_dim;
      const float* kb = k_cache + kv_head * head_stride;
      const float* vb = v_cache + kv_head * head_stride;
      for (size_t p = 0; p < kv_len; p++) scores[p] = orc_dot_f32(qv, kb + p * head_dim, head_dim) * scale;
      orc_softmax_inplace(scores.data(), kv_len);
      float* o = out + head * head_dim;
      for (size_t d = 0; d < head_dim; d++) o[d] = 0.0f;
      for (size_t p = 0; p < kv_len; p++)
        if (scores[p] > 1e-8f) orc_axpy_f32(scores[p], vb + p * head_dim, o, head_dim);
    }
  });
}

// ---- the rest of the per-op `Backend` surface (src/backend/mod.rs:29-265), CPU implementations in ops.rs ----
void orc_add(const float* a, const float* b, float* out, size_t n) {  // ops.rs:24-116: out = a + b, one rounding per element
  for (size_t i = 0; i < n; i++) out[i] = a[i] + b[i];
}
void orc_mul(const float* a, const float* b, float* out, size_t n) {  // ops.rs:119-208
  for (size_t i = 0; i < n; i++) out[i] = a[i] * b[i];
}
void orc_scale(const float* a, float s, float* out, size_t n) {  // ops.rs:211-300
  for (size_t i = 0; i < n; i++) out[i] = a[i] * s;
}
void orc_gelu(const float* x, float* out, size_t n) {  // ops.rs:328-347: tanh approximation, f32 throughout
  const float kSqrt2OverPi = 0.7978846f;
  for (size_t i = 0; i < n; i++) {
    float v = x[i];
    float inner = kSqrt2OverPi * (v + 0.044715f * v * v * v);
    out[i] = 0.5f * v * (1.0f + std::tanh(inner));
  }
}
void orc_softmax_rows(const float* x, float* out, size_t rows, size_t last_dim) {  // ops.rs:350-385: along the last dimension
  for (size_t r = 0; r < rows; r++) {
    for (size_t i = 0; i < last_dim; i++) out[r * last_dim + i] = x[r * last_dim + i];
    orc_softmax_inplace(out + r * last_dim, last_dim);
  }
}
void orc_matmul(const float* a, const float* b, float* c, size_t m, size_t k, size_t n) {
  // ops.rs:429-528: row-major [m,k] @ [k,n]; both variants (simple / tiled in k-chunks of 32 that continue the same
  // running sum) add the products of one output element in ascending k, one f32 rounding per step
  orc_parallel_for(m, 1, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++)
      for (size_t j = 0; j < n; j++) {
        float sum = 0.0f;
        for (size_t kk = 0; kk < k; kk++) sum += a[i * k + kk] * b[kk * n + j];
        c[i * n + j] = sum;
      }
  });
}
void orc_matvec(const float* a, const float* x, float* out, size_t m, size_t k) {  // ops.rs:531-570: dot_f32 per row
  orc_parallel_for(m, 8, [&](size_t i0, size_t i1) {
    for (size_t i = i0; i < i1; i++) out[i] = orc_dot_f32(a + i * k, x, k);
  });
}
void orc_attention(const float* q, const float* k, const float* v, float* out, size_t n_heads, size_t n_kv_heads, size_t seq_len,
                   size_t kv_len, size_t head_dim, float scale) {  // ops.rs:1353-1472: causal, GQA, scalar loops
  size_t per_kv = n_heads / n_kv_heads;
  std::vector<float> scores(kv_len);
  for (size_t head = 0; head < n_heads; head++) {
    size_t kvh = head / per_kv;
    for (size_t s = 0; s < seq_len; s++) {
      const float* qv = q + (head * seq_len + s) * head_dim;
      size_t q_abs = (kv_len >= seq_len ? kv_len - seq_len : 0) + s;   // saturating_sub (ops.rs:1411)
      for (size_t p = 0; p < kv_len; p++) {
        if (p > q_abs) { scores[p] = -std::numeric_limits<float>::infinity(); continue; }
        const float* kv = k + (kvh * kv_len + p) * head_dim;
        float dot = 0.0f;
        for (size_t d = 0; d < head_dim; d++) dot += qv[d] * kv[d];
        scores[p] = dot * scale;
      }
      float mx = -std::numeric_limits<float>::infinity();
      for (size_t p = 0; p < kv_len; p++) mx = std::fmax(mx, scores[p]);   // fold(NEG_INFINITY, f32::max)
      float sum = 0.0f;
      for (size_t p = 0; p < kv_len; p++) { scores[p] = std::exp(scores[p] - mx); sum += scores[p]; }
      float inv = 1.0f / sum;
      for (size_t p = 0; p < kv_len; p++) scores[p] *= inv;
      float* o = out + (head * seq_len + s) * head_dim;
      for (size_t d = 0; d < head_dim; d++) o[d] = 0.0f;
      for (size_t p = 0; p < kv_len; p++)
        if (scores[p] > 0.0f) {
          const float* vv = v + (kvh * kv_len + p) * head_dim;
          for (size_t d = 0; d < head_dim; d++) o[d] += scores[p] * vv[d];
        }
    }
  }
}

void orc_vec_mat_f32(const float* w, const float* x, float* out, size_t k, size_t n) {  // ops.rs:959-1002
  orc_parallel_for(n, 16, [&](size_t j0, size_t j1) {
    for (size_t j = j0; j < j1; j++) {
      float sum = 0.0f;
      for (size_t i = 0; i < k; i++) sum += x[i] * w[i + j * k];
      out[j] = sum;
    }
  });
}

int orc_vec_mat_q(int type, const void* w, const float* x, float* out, size_t k, size_t n) {  // ops.rs:1123-1199
  size_t bs = orc_block_size(type), bb = orc_block_bytes(type);
  if (!bs || k % bs) return 1;
  const uint8_t* raw = (const uint8_t*)w;
  size_t row_bytes = k / bs * bb;
  if (orc_has_fused_dot(type)) {
    orc_parallel_for(n, 8, [&](size_t j0, size_t j1) {
      for (size_t j = j0; j < j1; j++) out[j] = orc_dot_q(type, raw + j * row_bytes, x, k);
    });
    return 0;
  }
  // no fused dot: dequantize the whole matrix, then dot_f32(x, column)   (ops.rs:1182-1199)
  std::vector<float> wf(k * n);
  if (orc_dequantize(type, w, k * n, wf.data())) return 1;
  orc_parallel_for(n, 8, [&](size_t j0, size_t j1) {
    for (size_t j = j0; j < j1; j++) out[j] = orc_dot_f32(x, wf.data() + j * k, k);
  });
  return 0;
}

uint32_t orc_argmax_last(const float* v, size_t n) {  // main.rs:1815-1821: Iterator::max_by keeps the LAST max
  if (n == 0) return 0;
  size_t best = 0;
  for (size_t i = 1; i < n; i++)
    if (!(v[i] < v[best])) best = i;  // partial_cmp: Less keeps the old one, Equal/Greater take the new
  return (uint32_t)best;
}

uint32_t orc_greedy_sample(const float* logits, size_t n) {  // sampling/mod.rs:224-242 (temperature 0, no penalties)
  std::vector<float> p(logits, logits + n);
  float mx = -std::numeric_limits<float>::infinity();
  for (size_t i = 0; i < n; i++) mx = p[i] > mx ? p[i] : mx;
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) {
    p[i] = std::exp(p[i] - mx);
    sum += p[i];
  }
  for (size_t i = 0; i < n; i++) p[i] /= sum;
  return orc_argmax_last(p.data(), n);
}

void orc_moe_route(const float* h, const float* w, size_t hidden, size_t n_experts, size_t top_k, int normalize,
                   uint32_t* idx_out, float* weight_out) {  // moe.rs:128-198
  std::vector<float> logits(n_experts);
  for (size_t e = 0; e < n_experts; e++) logits[e] = orc_dot_f32(h, w + e * hidden, hidden);
  if (normalize) {
    float mx = -std::numeric_limits<float>::infinity();
    for (float l : logits) mx = l > mx ? l : mx;
    for (float& l : logits) l -= mx;
  }
  std::vector<std::pair<size_t, float>> idx(n_experts);
  for (size_t e = 0; e < n_experts; e++) idx[e] = {e, logits[e]};
  // slice::sort_by is stable: descending by logit, ties keep the lower expert index first
  std::stable_sort(idx.begin(), idx.end(),
                   [](const std::pair<size_t, float>& a, const std::pair<size_t, float>& b) { return b.second < a.second; });
  float mx = -std::numeric_limits<float>::infinity();
  for (size_t i = 0; i < top_k; i++) mx = idx[i].second > mx ? idx[i].second : mx;
  float exp_sum = 0.0f;
  for (size_t i = 0; i < top_k; i++) exp_sum += std::exp(idx[i].second - mx);
  for (size_t i = 0; i < top_k; i++) {
    idx_out[i] = (uint32_t)idx[i].first;
    weight_out[i] = std::exp(idx[i].second - mx) / exp_sum;
  }
}

}  // extern "C"
