// ops_api.hip — the per-op C entry points (host tensors in / host tensors out) that mirror the
// reference's `Backend` trait ops on the decode path (src/backend/mod.rs:29-265), plus the kernel
// micro-benchmark.  They run the SAME kernels the engine runs, on a throw-away context, so that each
// kernel can be checked against the CPU backend in isolation.
#include "engine.h"
#include "prefill.h"

#include <cmath>
#include <vector>

using namespace lgh;

#include <algorithm>

#ifdef LGH_STAMPS
namespace lgh { hipError_t pf_read_stamps(unsigned long long* host, size_t n); }
extern "C" int lgh_debug_pf_stamps(unsigned long long* out, size_t n) { return lgh::pf_read_stamps(out, n) == hipSuccess ? 0 : 10; }
namespace lgh { hipError_t mv_read_stamps(unsigned long long* host, size_t n); hipError_t mvq_read_stamps(unsigned long long* host, size_t n); hipError_t mvq_read_wave_stamps(unsigned long long* host, size_t n); hipError_t mvq_spans(unsigned long long* host, int reset); }
#include <cstdio>
#include "timeline.h"
namespace lgh { hipError_t tl_read_mvq(void*); hipError_t tl_read_attn(void*); hipError_t tl_read_deq(void*); hipError_t tl_read_misc(void*); }
// the per-node timeline buffers of the four translation units on the decode path (timeline.h), `which` = 0..3;
// out: sizeof(TlBuf) bytes.  Returns the buffer size when out == nullptr.
extern "C" long long lgh_debug_timeline(int which, void* out) {
  if (!out) return (long long)sizeof(lgh::TlBuf);
  hipError_t e = which == 0 ? lgh::tl_read_mvq(out) : which == 1 ? lgh::tl_read_attn(out) : which == 2 ? lgh::tl_read_deq(out) : lgh::tl_read_misc(out);
  return e == hipSuccess ? 0 : -1;
}
#endif

namespace {

struct Tmp {  // bare context: stream, device state words, tracked allocations
  lgh_ctx* c = nullptr;
  int rc = LGH_OK;
  explicit Tmp(int device) {
    int n = lgh_device_count();
    if (n <= 0 || device < 0 || device >= n) { rc = LGH_NOT_AVAILABLE; return; }
    c = new lgh_ctx();
    c->device = device;
    c->d.norm_eps = 1e-5f;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
      rc = LGH_INITIALIZATION_FAILED;
      delete c;
      c = nullptr;
      return;
    }
    c->stream = c->own_stream;
    if ((rc = dev_alloc(c, (void**)&c->state, ST_WORDS * 4))) return;
    (void)hipMemsetAsync(c->state, 0, ST_WORDS * 4, c->stream);
  }
  ~Tmp() {
    if (!c) return;
    (void)hipStreamSynchronize(c->stream);
    for (void* p : c->allocs) (void)hipFree(p);
    (void)hipStreamDestroy(c->own_stream);
    delete c;
  }
  float* up(const float* host, size_t n) {  // device copy of a host f32 vector (nullptr on failure)
    float* d = nullptr;
    if (dev_alloc(c, (void**)&d, n * 4)) return nullptr;
    if (host && hipMemcpyAsync(d, host, n * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return nullptr;
    return d;
  }
  int down(float* host, const float* dev, size_t n) {
    if (hipMemcpyAsync(host, dev, n * 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
    return hipStreamSynchronize(c->stream) == hipSuccess ? LGH_OK : LGH_OPERATION_FAILED;
  }
};

}  // namespace

extern "C" {

int lgh_op_dequantize(int device, uint32_t type, const void* src, size_t n, float* dst) {
  Tmp t(device);
  if (t.rc) return t.rc;
  const uint32_t be = blk_elems((int)type);
  if (!be || n % be) return LGH_UNSUPPORTED_DTYPE;
  const size_t nbytes = n / be * blk_bytes((int)type);
  uint8_t* raw = nullptr;
  float* out = nullptr;
  if (dev_alloc(t.c, (void**)&raw, nbytes) || dev_alloc(t.c, (void**)&out, n * 4)) return LGH_ALLOCATION_FAILED;
  if (hipMemcpyAsync(raw, src, nbytes, hipMemcpyHostToDevice, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  if (dequant_launch((int)type, raw, out, n, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(dst, out, n);
}

static int vec_mat_impl(int device, uint32_t type, const void* w, const void* w2, const float* x, const float* norm_w,
                        float eps, const float* resid, float* out, size_t k, size_t n) {
  Tmp t(device);
  if (t.rc) return t.rc;
  t.c->d.norm_eps = eps;
  const uint32_t be = blk_elems((int)type);
  if (!be || k % be) return LGH_SHAPE_MISMATCH;
  const size_t nbytes = n * (k / be) * blk_bytes((int)type);
  DevWeight W, W2;
  int rc;
  if ((rc = upload_matrix(t.c, W, (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w, nbytes))) return rc;
  if (w2 && (rc = upload_matrix(t.c, W2, (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w2, nbytes))) return rc;
  float* dx = t.up(x, k);
  float* dnw = norm_w ? t.up(norm_w, k) : nullptr;
  float* dres = resid ? t.up(resid, n) : nullptr;
  float* dout = t.up(nullptr, n);
  if (!dx || !dout || (norm_w && !dnw) || (resid && !dres)) return LGH_ALLOCATION_FAILED;
  if (w2) {
    if (!fused_type(W.type)) return LGH_UNSUPPORTED;
    SegSpec sp;
    sp.npass = 2;
    sp.W[0] = &W; sp.W[1] = &W2;
    sp.x[0] = sp.x[1] = dx;
    sp.epi = EPI_SWIGLU;
    sp.out = dout;
    if ((rc = launch_mv(t.c, LGH_K_GATEUP, &sp, 1, dnw, (uint32_t)k))) return rc;
  } else if ((rc = linear_any(t.c, LGH_K_MISC, W, dx, dout, dnw, dres, nullptr))) {
    return rc;
  }
  return t.down(out, dout, n);
}

int lgh_op_vec_mat(int device, uint32_t type, const void* w, const float* x, float* out, size_t k, size_t n) {
  return vec_mat_impl(device, type, w, nullptr, x, nullptr, 1e-5f, nullptr, out, k, n);
}

int lgh_op_norm_vec_mat(int device, uint32_t type, const void* w, const float* x, const float* norm_w, float eps, float* out,
                        size_t k, size_t n) {
  return vec_mat_impl(device, type, w, nullptr, x, norm_w, eps, nullptr, out, k, n);
}

int lgh_op_swiglu_vec_mat(int device, uint32_t type, const void* w_gate, const void* w_up, const float* x,
                          const float* norm_w, float eps, float* out, size_t k, size_t n) {
  return vec_mat_impl(device, type, w_gate, w_up, x, norm_w, eps, nullptr, out, k, n);
}

// out[m][n] = x[m][k] . W^T on the batched-prefill GEMM path (prefill.hip): f16 operands, f32 accumulation
int lgh_op_mat_mat(int device, uint32_t type, const void* w, const float* x, float* out, size_t k, size_t n, size_t m) {
  Tmp t(device);
  if (t.rc) return t.rc;
  const uint32_t be = blk_elems((int)type);
  if (!be || k % be || m == 0 || m > (size_t)kPfTokens) return LGH_SHAPE_MISMATCH;
  const size_t nbytes = n * (k / be) * blk_bytes((int)type);
  DevWeight W;
  int rc;
  if ((rc = upload_matrix(t.c, W, (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w, nbytes))) return rc;
  if (!pf_supported_type(W.type) || n % 16) return LGH_UNSUPPORTED;
  const uint32_t nr[1] = {(uint32_t)n};
  const size_t pb = pf_part_bytes(nr, 1, (uint32_t)k);
  float* dx = t.up(x, m * k);
  float* dout = t.up(nullptr, (size_t)kPfTokens * n);
  uint8_t* xh = nullptr;
  float *part = nullptr, *ssq = nullptr;
  if (n > 2048u * kPfSsqChunks) return LGH_UNSUPPORTED;
  if (!dx || !dout || dev_alloc(t.c, (void**)&xh, xh_bytes((uint32_t)k)) || dev_alloc(t.c, (void**)&part, pb) ||
      dev_alloc(t.c, (void**)&ssq, (size_t)kPfTokens * kPfSsqChunks * 4))
    return LGH_ALLOCATION_FAILED;
  hipStream_t st = t.c->stream;
  if (hipMemsetAsync(dout, 0, (size_t)kPfTokens * n * 4, st) != hipSuccess || hipMemsetAsync(xh, 0, xh_bytes((uint32_t)k), st) != hipSuccess)
    return LGH_OPERATION_FAILED;
  const DevWeight* Ws[1] = {&W};
  uint32_t S = 0, nc = 0;
  if (pf_to_xh_launch(dx, (uint32_t)k, xh, (uint32_t)m, st) != hipSuccess) return LGH_OPERATION_FAILED;
  if (pf_gemm_launch(Ws, 1, xh, part, pb, (uint32_t)m, &S, &nc, st) != hipSuccess) return LGH_OPERATION_FAILED;
  if (pf_row_epi_launch(part, S, nc, 0, nullptr, dout, (uint32_t)n, nullptr, nullptr, ssq, (uint32_t)m, st) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, m * n);
}

int lgh_op_rms_norm(int device, const float* x, const float* w, float eps, float* out, size_t n) {
  Tmp t(device);
  if (t.rc) return t.rc;
  float *dx = t.up(x, n), *dw = t.up(w, n), *dout = t.up(nullptr, n);
  if (!dx || !dw || !dout) return LGH_ALLOCATION_FAILED;
  if (rms_norm_launch(dx, dw, eps, dout, (uint32_t)n, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, n);
}

int lgh_op_rope(int device, float* q, float* k, size_t n_heads, size_t n_kv, size_t d, size_t pos, float freq_base,
                float freq_scale, int neox) {
  Tmp t(device);
  if (t.rc) return t.rc;
  const size_t half = d / 2;
  std::vector<float> cs(half * 2);  // table row for `pos`, reference arithmetic (ops.rs:1300-1313)
  const float position = (float)pos / freq_scale;
  for (size_t i = 0; i < half; i++) {
    const float freq = 1.0f / std::pow(freq_base, (float)(2 * i) / (float)d);
    const float theta = position * freq;
    cs[2 * i] = std::cos(theta);
    cs[2 * i + 1] = std::sin(theta);
  }
  float *dq = t.up(q, n_heads * d), *dk = t.up(k, n_kv * d), *dcs = t.up(cs.data(), cs.size());
  if (!dq || !dk || !dcs) return LGH_ALLOCATION_FAILED;
  // state[ST_POS] stays 0: the one-row table is indexed at position 0
  if (rope_launch(dq, dk, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)d, t.c->state + ST_POS, dcs, neox, t.c->stream) != hipSuccess)
    return LGH_OPERATION_FAILED;
  int rc = t.down(q, dq, n_heads * d);
  if (rc) return rc;
  return t.down(k, dk, n_kv * d);
}

int lgh_op_attention_cached(int device, const float* q, const float* kc, const float* vc, float* out, size_t n_heads,
                            size_t n_kv, size_t d, size_t max_seq, float scale, size_t kv_len, int n_splits) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (kv_len == 0 || kv_len > max_seq || n_kv == 0 || n_heads % n_kv) return LGH_INVALID_ARGUMENT;
  if (n_splits <= 0) n_splits = 8;
  const size_t g = n_heads / n_kv, cache = n_kv * max_seq * d;
  float *dq = t.up(q, n_heads * d), *dk = t.up(kc, cache), *dv = t.up(vc, cache), *dout = t.up(nullptr, n_heads * d);
  float *pml = t.up(nullptr, n_kv * n_splits * g * 2), *pacc = t.up(nullptr, n_kv * n_splits * g * d);
  if (!dq || !dk || !dv || !dout || !pml || !pacc) return LGH_ALLOCATION_FAILED;
  if (!((d == 64 || d == 128) && (g == 1 || g == 2 || g == 4 || g == 8))) {   // shapes outside the engine's kernels
    if (attn_generic_launch(dq, dk, dv, dout, (uint32_t)n_heads, (uint32_t)n_kv, 1, (uint32_t)kv_len, (uint32_t)max_seq, (uint32_t)d, scale,
                            t.c->stream) != hipSuccess)
      return LGH_UNSUPPORTED;
    return t.down(out, dout, n_heads * d);
  }
  if (attn_launch(dq, dk, dv, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)d, (uint32_t)max_seq, scale, nullptr, (int)kv_len,
                  (uint32_t)n_splits, pml, pacc, t.c->stream) != hipSuccess)
    return LGH_UNSUPPORTED;
  if (attn_combine_launch(pml, pacc, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)d, (uint32_t)n_splits, dout, nullptr, t.c->stream) != hipSuccess)
    return LGH_OPERATION_FAILED;
  return t.down(out, dout, n_heads * d);
}

int lgh_op_silu_mul(int device, const float* gate, const float* up, float* out, size_t n) {
  Tmp t(device);
  if (t.rc) return t.rc;
  float *dg = t.up(gate, n), *du = t.up(up, n), *dout = t.up(nullptr, n);
  if (!dg || !du || !dout) return LGH_ALLOCATION_FAILED;
  if (silu_mul_launch(dg, du, dout, (uint32_t)n, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, n);
}

// ---- Backend::add / mul / scale / silu / gelu / softmax / matmul / matvec / matvec_q / attention (backend/mod.rs:29-265) ----
static int ewise_impl(int device, int op, const float* a, const float* b, float s, float* out, size_t n) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (n == 0) return LGH_OK;
  if (!a || !out || ((op == 0 || op == 1) && !b)) return LGH_INVALID_ARGUMENT;
  float *da = t.up(a, n), *db = b ? t.up(b, n) : nullptr, *dout = t.up(nullptr, n);
  if (!da || !dout || (b && !db)) return LGH_ALLOCATION_FAILED;
  if (ewise_launch(op, da, db, s, dout, n, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, n);
}
int lgh_op_add(int device, const float* a, const float* b, float* out, size_t n) { return ewise_impl(device, 0, a, b, 0.0f, out, n); }
int lgh_op_mul(int device, const float* a, const float* b, float* out, size_t n) { return ewise_impl(device, 1, a, b, 0.0f, out, n); }
int lgh_op_scale(int device, const float* a, float scalar, float* out, size_t n) { return ewise_impl(device, 2, a, nullptr, scalar, out, n); }
int lgh_op_silu(int device, const float* x, float* out, size_t n) { return ewise_impl(device, 3, x, nullptr, 0.0f, out, n); }
int lgh_op_gelu(int device, const float* x, float* out, size_t n) { return ewise_impl(device, 4, x, nullptr, 0.0f, out, n); }

int lgh_op_softmax(int device, const float* x, float* out, size_t rows, size_t last_dim) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (rows * last_dim == 0) return LGH_OK;
  if (!x || !out || rows > 0x7FFFFFFFu || last_dim > 0xFFFFFFFFu) return LGH_INVALID_ARGUMENT;
  float *dx = t.up(x, rows * last_dim), *dout = t.up(nullptr, rows * last_dim);
  if (!dx || !dout) return LGH_ALLOCATION_FAILED;
  if (softmax_rows_launch(dx, dout, (uint32_t)rows, (uint32_t)last_dim, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, rows * last_dim);
}

int lgh_op_kv_roundtrip(int device, uint32_t kv_cache_type, const float* row, size_t n, uint8_t* bytes_out, float* scale_out, float* back_out) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (kv_cache_type < LGH_KV_INT8 || kv_cache_type > LGH_KV_FP8_E5M2) return LGH_UNSUPPORTED;
  if (n == 0) return LGH_OK;
  if (!row || !bytes_out || !back_out || n > 0x7FFFFFFFu) return LGH_INVALID_ARGUMENT;
  float *dx = t.up(row, n), *dback = t.up(nullptr, n), *dsc = t.up(nullptr, 1);
  uint8_t* db = reinterpret_cast<uint8_t*>(t.up(nullptr, (n + 3) / 4));
  if (!dx || !dback || !dsc || !db) return LGH_ALLOCATION_FAILED;
  if (kv_roundtrip_launch((int)kv_cache_type, dx, (uint32_t)n, db, dsc, dback, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  std::vector<float> tmp((n + 3) / 4);
  int rc = t.down(tmp.data(), reinterpret_cast<float*>(db), (n + 3) / 4);
  if (rc) return rc;
  std::memcpy(bytes_out, tmp.data(), n);
  float sc = 1.0f;
  if (kv_cache_type == LGH_KV_INT8 && (rc = t.down(&sc, dsc, 1))) return rc;
  if (scale_out) *scale_out = sc;
  return t.down(back_out, dback, n);
}

int lgh_op_tq_compress(int device, int bits, const float* x, size_t dim, const float* signs, uint8_t* codes) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!x || !signs || !codes) return LGH_INVALID_ARGUMENT;
  if ((bits != 2 && bits != 3) || (dim != 64 && dim != 128)) return LGH_UNSUPPORTED;
  for (size_t i = 0; i < dim; i++)
    if (signs[i] != 1.0f && signs[i] != -1.0f) return LGH_INVALID_ARGUMENT;
  float *dx = t.up(x, dim), *ds = t.up(signs, dim);
  const size_t rb = tq_row_bytes_host(bits, (uint32_t)dim);
  uint8_t* dc = reinterpret_cast<uint8_t*>(t.up(nullptr, (rb + 3) / 4));
  if (!dx || !ds || !dc) return LGH_ALLOCATION_FAILED;
  if (tq_compress_launch(bits, dx, (uint32_t)dim, ds, dc, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  std::vector<float> tmp((rb + 3) / 4);
  int rc = t.down(tmp.data(), reinterpret_cast<float*>(dc), (rb + 3) / 4);
  if (rc) return rc;
  std::memcpy(codes, tmp.data(), rb);
  return LGH_OK;
}

int lgh_op_tq_compress_qjl(int device, int bits, const float* x, size_t dim, const float* signs, const float* qjl_matrix, uint8_t* codes,
                           uint64_t* qjl_bits, float* residual_norm) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!x || !signs || !codes || !qjl_matrix || !qjl_bits || !residual_norm) return LGH_INVALID_ARGUMENT;
  if ((bits != 2 && bits != 3) || (dim != 64 && dim != 128)) return LGH_UNSUPPORTED;
  for (size_t i = 0; i < dim; i++)
    if (signs[i] != 1.0f && signs[i] != -1.0f) return LGH_INVALID_ARGUMENT;
  float *dx = t.up(x, dim), *ds = t.up(signs, dim), *dS = t.up(qjl_matrix, dim * dim);
  const size_t rb = tq_row_bytes_host(bits, (uint32_t)dim), xw = dim / 32 + 1;
  uint8_t* dc = reinterpret_cast<uint8_t*>(t.up(nullptr, (rb + 3) / 4));
  uint32_t* dq = reinterpret_cast<uint32_t*>(t.up(nullptr, xw));
  if (!dx || !ds || !dS || !dc || !dq) return LGH_ALLOCATION_FAILED;
  if (tq_compress_launch(bits, dx, (uint32_t)dim, ds, dc, t.c->stream, dS, dq) != hipSuccess) return LGH_OPERATION_FAILED;
  std::vector<float> tmp((rb + 3) / 4), tq(xw);
  int rc = t.down(tmp.data(), reinterpret_cast<float*>(dc), (rb + 3) / 4);
  if (rc) return rc;
  if ((rc = t.down(tq.data(), reinterpret_cast<float*>(dq), xw))) return rc;
  std::memcpy(codes, tmp.data(), rb);
  uint32_t w[8] = {};
  std::memcpy(w, tq.data(), xw * 4);
  for (size_t i = 0; i < dim / 64; i++) qjl_bits[i] = (uint64_t)w[2 * i] | (uint64_t)w[2 * i + 1] << 32;
  std::memcpy(residual_norm, &w[dim / 32], 4);
  return LGH_OK;
}

int lgh_op_matmul(int device, const float* a, const float* b, float* out, size_t m, size_t k, size_t n) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!a || !b || !out || m == 0 || n == 0 || m > 65535) return LGH_INVALID_ARGUMENT;
  float *da = t.up(a, m * k), *db = t.up(b, k * n), *dout = t.up(nullptr, m * n);
  if (!da || !db || !dout) return LGH_ALLOCATION_FAILED;
  if (matmul_f32_launch(da, db, dout, (uint32_t)m, (uint32_t)k, (uint32_t)n, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  return t.down(out, dout, m * n);
}

int lgh_op_matvec(int device, const float* a, const float* x, float* out, size_t m, size_t k) {
  // [m,k] row-major times [k]: the memory layout of vec_mat's GGUF-order weight with n = m (ops.rs:531-570 vs 959-1002)
  return vec_mat_impl(device, LGH_TYPE_F32, a, nullptr, x, nullptr, 1e-5f, nullptr, out, k, m);
}

int lgh_op_matvec_q(int device, uint32_t type, const void* a, const float* x, float* out, size_t m, size_t k) {
  // quantized rows of k elements, m of them: the same bytes vec_mat_q reads (ops.rs:922-950 vs 1008-1039)
  return vec_mat_impl(device, type, a, nullptr, x, nullptr, 1e-5f, nullptr, out, k, m);
}

int lgh_op_attention(int device, const float* q, const float* k, const float* v, float* out, size_t n_heads, size_t n_kv, size_t seq_len,
                     size_t kv_len, size_t d, float scale) {
  // q / out [heads, seq, d], k / v [kv_heads, kv_len, d]; query s sits at position kv_len - seq_len + s and sees the
  // rows up to itself (ops.rs:1353-1472).  One split-attention pass per query position over the SAME kernels the engine
  // runs: k / v are a cache with max_seq = kv_len.
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!q || !k || !v || !out || n_kv == 0 || n_heads % n_kv || seq_len == 0 || kv_len == 0) return LGH_INVALID_ARGUMENT;
  const size_t g = n_heads / n_kv, cache = n_kv * kv_len * d;
  const int n_splits = 8;
  float *dq = t.up(q, n_heads * seq_len * d), *dk = t.up(k, cache), *dv = t.up(v, cache), *dout = t.up(nullptr, n_heads * seq_len * d);
  float *qs = t.up(nullptr, n_heads * d), *os = t.up(nullptr, n_heads * d);
  float *pml = t.up(nullptr, n_kv * n_splits * g * 2), *pacc = t.up(nullptr, n_kv * n_splits * g * d);
  if (!dq || !dk || !dv || !dout || !qs || !os || !pml || !pacc) return LGH_ALLOCATION_FAILED;
  hipStream_t st = t.c->stream;
  const bool fast = (d == 64 || d == 128) && (g == 1 || g == 2 || g == 4 || g == 8);
  if (!fast) {   // shapes outside the engine's kernels: the generic kernel
    if (attn_generic_launch(dq, dk, dv, dout, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)seq_len, (uint32_t)kv_len, (uint32_t)kv_len, (uint32_t)d, scale, st) != hipSuccess)
      return LGH_UNSUPPORTED;
    return t.down(out, dout, n_heads * seq_len * d);
  }
  for (size_t s = 0; s < seq_len; s++) {
    const size_t q_abs = (kv_len >= seq_len ? kv_len - seq_len : 0) + s;   // saturating_sub (ops.rs:1411)
    const size_t visible = q_abs + 1 < kv_len ? q_abs + 1 : kv_len;
    if (hipMemcpy2DAsync(qs, d * 4, dq + s * d, seq_len * d * 4, d * 4, n_heads, hipMemcpyDeviceToDevice, st) != hipSuccess) return LGH_OPERATION_FAILED;
    if (attn_launch(qs, dk, dv, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)d, (uint32_t)kv_len, scale, nullptr, (int)visible,
                    (uint32_t)n_splits, pml, pacc, st) != hipSuccess)
      return LGH_UNSUPPORTED;
    if (attn_combine_launch(pml, pacc, (uint32_t)n_heads, (uint32_t)n_kv, (uint32_t)d, (uint32_t)n_splits, os, nullptr, st) != hipSuccess)
      return LGH_OPERATION_FAILED;
    if (hipMemcpy2DAsync(dout + s * d, seq_len * d * 4, os, d * 4, d * 4, n_heads, hipMemcpyDeviceToDevice, st) != hipSuccess) return LGH_OPERATION_FAILED;
  }
  return t.down(out, dout, n_heads * seq_len * d);
}

// ---- device-resident weights by tensor name: CudaBackend::load_model_weights + the `b.name()` lookups of its vec_mat /
// vec_mat_q (src/backend/cuda/mod.rs:121-146, 436-470, 511-575; store: cuda/dequant_weights.rs) ----
}  // extern "C"

#include <map>
#include <memory>
#include <mutex>
#include <string>

struct lgh_backend {
  std::unique_ptr<Tmp> t;
  std::map<std::string, DevWeight> weights;
  float *x = nullptr, *out = nullptr;   // staging, grown on demand
  size_t x_cap = 0, out_cap = 0;
  uint64_t hits = 0;
  std::mutex mu;   // `Backend: Send + Sync` (backend/mod.rs:29): calls on one handle may come from several threads
};

extern "C" {

int lgh_backend_create(int device, lgh_backend** out) {
  if (!out) return LGH_INVALID_ARGUMENT;
  *out = nullptr;
  auto be = std::make_unique<lgh_backend>();
  be->t = std::make_unique<Tmp>(device);
  if (be->t->rc) return be->t->rc;
  *out = be.release();
  return LGH_OK;
}

void lgh_backend_destroy(lgh_backend* be) { delete be; }

int lgh_backend_load_weight(lgh_backend* be, const char* name, uint32_t type, const void* w, size_t k, size_t n) {
  if (!be || !name || !w) return LGH_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(be->mu);
  lgh_ctx* c = be->t->c;
  if (hipSetDevice(c->device) != hipSuccess) return LGH_NOT_AVAILABLE;
  const uint32_t bs = blk_elems((int)type);
  if (!bs || k % bs) return LGH_SHAPE_MISMATCH;
  if (be->weights.count(name)) return fail(c, LGH_INVALID_ARGUMENT, std::string("weight already loaded: ") + name);
  DevWeight W;
  int rc = upload_matrix(c, W, (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w, n * (k / bs) * blk_bytes((int)type));
  if (rc) return rc;
  be->weights.emplace(name, W);
  return LGH_OK;
}

int lgh_backend_has_weight(const lgh_backend* be, const char* name) {
  if (!be || !name) return 0;
  std::lock_guard<std::mutex> lock(const_cast<lgh_backend*>(be)->mu);
  return be->weights.count(name) ? 1 : 0;
}

int lgh_backend_vec_mat_q(lgh_backend* be, const char* name, const float* x, float* out, size_t k, size_t n) {
  if (!be || !name || !x || !out) return LGH_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(be->mu);
  lgh_ctx* c = be->t->c;
  if (hipSetDevice(c->device) != hipSuccess) return LGH_NOT_AVAILABLE;
  auto it = be->weights.find(name);
  if (it == be->weights.end()) return fail(c, LGH_INVALID_ARGUMENT, std::string("no device-resident weight named ") + name);
  const DevWeight& W = it->second;
  if (W.k != k || W.n != n) return fail(c, LGH_SHAPE_MISMATCH, std::string(name) + ": vec_mat_q dimension mismatch");   // cuda/mod.rs:528-534
  if (k > be->x_cap) { if (dev_alloc(c, (void**)&be->x, k * 4)) return LGH_ALLOCATION_FAILED; be->x_cap = k; }
  if (n > be->out_cap) { if (dev_alloc(c, (void**)&be->out, n * 4)) return LGH_ALLOCATION_FAILED; be->out_cap = n; }
  if (hipMemcpyAsync(be->x, x, k * 4, hipMemcpyHostToDevice, c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  xq_stale(c, be->x);
  int rc = linear_any(c, LGH_K_MISC, W, be->x, be->out, nullptr, nullptr, nullptr);
  if (rc) return rc;
  be->hits++;
  return be->t->down(out, be->out, n);
}

const char* lgh_backend_last_error(const lgh_backend* be) { return be ? be->t->c->err.c_str() : "null backend"; }

int lgh_bench_vec_mat(int device, uint32_t type, const void* w, const void* w2, size_t k, size_t n, int mode, int iters,
                      int copies, double* avg_us) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!avg_us || iters <= 0) return LGH_INVALID_ARGUMENT;
  const bool two_streams = (mode & 16) != 0;
  mode &= 15;
  const uint32_t be = blk_elems((int)type);
  if (!be || k % be) return LGH_SHAPE_MISMATCH;
  const size_t nbytes = n * (k / be) * blk_bytes((int)type);
  if (copies < 1) copies = 1;
  if (copies > 64) copies = 64;
  // `copies` distinct device copies are cycled so that consecutive launches do not re-read weights that the
  // 256 MiB Infinity Cache still holds (in the engine a token streams GBs between two uses of a matrix)
  std::vector<DevWeight> Ws(copies), W2s(copies);
  int rc;
  for (int i = 0; i < copies; i++) {
    if ((rc = upload_matrix(t.c, Ws[i], (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w, nbytes))) return rc;
    if (mode == 2 && (rc = upload_matrix(t.c, W2s[i], (int)type, (uint32_t)k, (uint32_t)n, 1, -1, w2 ? w2 : w, nbytes))) return rc;
  }
  // mode bit 4 (experiment): consecutive launches alternate between two streams with nothing ordering them — how much of a
  // launch's fixed cost hides behind its neighbour when the hardware may overlap them
  hipStream_t s2 = nullptr;
  if (two_streams && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return LGH_OPERATION_FAILED;
  hipStream_t s1 = t.c->stream;
  int flip = 0;
  int cur = 0;
  std::vector<float> hx(k), hw(k, 1.0f);
  for (size_t i = 0; i < k; i++) hx[i] = 0.001f * (float)((i * 2654435761u) % 2001) - 1.0f;
  float *dx = t.up(hx.data(), k), *dnw = t.up(hw.data(), k), *dout = t.up(nullptr, n);
  if (!dx || !dnw || !dout) return LGH_ALLOCATION_FAILED;
  auto once = [&]() -> int {
    DevWeight& W = Ws[cur];
    DevWeight& W2 = W2s[cur];
    cur = (cur + 1) % copies;
    if (two_streams) { t.c->stream = (flip ^= 1) ? s2 : s1; for (auto& q : t.c->xqs) q.fresh = true; }
    if (mode == 2) {
      SegSpec sp;
      sp.npass = 2;
      sp.W[0] = &W; sp.W[1] = &W2;
      sp.x[0] = sp.x[1] = dx;
      sp.epi = EPI_SWIGLU;
      sp.out = dout;
      return launch_mv(t.c, LGH_K_GATEUP, &sp, 1, dnw, (uint32_t)k);
    }
    return linear_any(t.c, LGH_K_MISC, W, dx, dout, mode == 1 ? dnw : nullptr, nullptr, nullptr);
  };
  for (int i = 0; i < 3; i++)
    if ((rc = once())) return rc;
  hipEvent_t a, b;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return LGH_OPERATION_FAILED;
#ifdef LGH_STAMPS
  (void)lgh::mvq_spans(nullptr, 1);
#endif
  if (two_streams) { (void)hipStreamSynchronize(s1); (void)hipStreamSynchronize(s2); t.c->stream = s1; }
  (void)hipEventRecord(a, s1);
  if (two_streams) { (void)hipStreamWaitEvent(s2, a, 0); }
  for (int i = 0; i < iters; i++)
    if ((rc = once())) break;
  if (two_streams) {   // b on s1 after both streams have drained
    hipEvent_t j;
    (void)hipEventCreateWithFlags(&j, hipEventDisableTiming);
    (void)hipEventRecord(j, s2);
    (void)hipStreamWaitEvent(s1, j, 0);
    t.c->stream = s1;
  }
  (void)hipEventRecord(b, s1);
  hipError_t e = hipEventSynchronize(b);
  if (s2) { (void)hipStreamSynchronize(s2); (void)hipStreamDestroy(s2); }
  float ms = 0.0f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  if (rc) return rc;
  if (e != hipSuccess) return LGH_OPERATION_FAILED;
  *avg_us = (double)ms * 1000.0 / iters;
#ifdef LGH_STAMPS
  {  // phase profile of the LAST launch: per stamp, min / median / max over workgroups, relative to the first start
    std::vector<unsigned long long> st(8192 * 8);
    if ((mfma_type(Ws[0].type) ? lgh::mvq_read_stamps(st.data(), st.size()) : lgh::mv_read_stamps(st.data(), st.size())) == hipSuccess) {
      MvPlan plan;
      if (mfma_type(Ws[0].type)) (void)mvq_plan((uint32_t)k, (uint32_t)n, mode == 2 ? 2 : 1, &plan, (uint32_t)n);
      else (void)mv_plan(Ws[0].type, (uint32_t)k, (uint32_t)n, mode == 2 ? 2 : 1, &plan, (uint32_t)n, 0);
      size_t nwg = std::min<size_t>(plan.n_wg, 8192);
      unsigned long long t0 = ~0ull;
      for (size_t w = 0; w < nwg; w++) t0 = std::min(t0, st[w * 8]);
      std::fprintf(stderr, "  stamps (us since first workgroup start; %zu workgroups x %u threads, rows/wg %u):\n", nwg, plan.threads, plan.rows_per_wg);
      const char* names[8] = {"start", "x published", "first item done", "stream done", "after barrier", "end", "loads issued", "x staged"};
      for (int i = 0; i < 8; i++) {
        std::vector<double> v;
        for (size_t w = 0; w < nwg; w++) v.push_back((double)(st[w * 8 + i] - t0) / 100.0);
        std::sort(v.begin(), v.end());
        std::fprintf(stderr, "    %-16s min %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", names[i], v[0], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
      }
      if (mfma_type(Ws[0].type)) {
        unsigned long long sp[128];
        if (lgh::mvq_spans(sp, 0) == hipSuccess) {   // consecutive launches: busy span and idle gap between them
          std::vector<double> busy, idle;
          for (int i = 0; i < 64; i++) {
            const int j = (i + 63) & 63;
            if (sp[2 * i] == ~0ull || sp[2 * j] == ~0ull || sp[2 * i] < sp[2 * j]) continue;
            busy.push_back((double)(sp[2 * i + 1] - sp[2 * i]) / 100.0);
            idle.push_back((double)((long long)sp[2 * i] - (long long)sp[2 * j + 1]) / 100.0);
          }
          std::sort(busy.begin(), busy.end());
          std::sort(idle.begin(), idle.end());
          if (!busy.empty())
            std::fprintf(stderr, "    launch spans (%zu): first start -> last end p50 %.2f us; previous last end -> first start p50 %.2f us (min %.2f)\n",
                         busy.size(), busy[busy.size() / 2], idle[idle.size() / 2], idle[0]);
        }
        std::vector<unsigned long long> ws(2048 * 16 * 8);
        if (lgh::mvq_read_wave_stamps(ws.data(), ws.size()) == hipSuccess) {
          const size_t nw = plan.threads / 64, ng = std::min<size_t>(nwg, 2048);
          const char* wn[8] = {"start", "scalars", "pre-x", "issued", "x here", "staged", "streamed", "end"};
          for (int i = 0; i < 8; i++) {
            std::fprintf(stderr, "    per-wave %-8s p50:", wn[i]);
            for (size_t w = 0; w < nw; w++) {
              std::vector<double> v;
              for (size_t g = 0; g < ng; g++) v.push_back((double)(ws[(g * 16 + w) * 8 + i] - t0) / 100.0);
              std::sort(v.begin(), v.end());
              std::fprintf(stderr, " %5.2f", v[v.size() / 2]);
            }
            std::fprintf(stderr, "\n");
          }
        }
      }
    }
  }
#endif
  return LGH_OK;
}

int lgh_bench_hbm_read(int device, size_t bytes, int iters, double* gbps) {
  Tmp t(device);
  if (t.rc) return t.rc;
  if (!gbps || iters <= 0 || bytes < (1u << 20)) return LGH_INVALID_ARGUMENT;
  bytes &= ~(size_t)4095;
  uint8_t* buf = nullptr;
  float* sink = nullptr;
  if (dev_alloc(t.c, (void**)&buf, bytes) || dev_alloc(t.c, (void**)&sink, 4096 * 4)) return LGH_ALLOCATION_FAILED;
  if (hipMemsetAsync(buf, 1, bytes, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;
  double best = 0.0;
  for (int nt = 0; nt < 2; nt++) {   // both cache policies; the better one is the ceiling
    if (hbm_read_launch(buf, bytes, sink, nt, t.c->stream) != hipSuccess) return LGH_OPERATION_FAILED;   // warm-up
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return LGH_OPERATION_FAILED;
    (void)hipEventRecord(a, t.c->stream);
    hipError_t e = hipSuccess;
    for (int i = 0; i < iters && e == hipSuccess; i++) e = hbm_read_launch(buf, bytes, sink, nt, t.c->stream);
    (void)hipEventRecord(b, t.c->stream);
    if (e == hipSuccess) e = hipEventSynchronize(b);
    float ms = 0.0f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (e != hipSuccess) return LGH_OPERATION_FAILED;
    best = std::max(best, (double)bytes * iters / ((double)ms * 1e-3) / 1e9);
  }
  *gbps = best;
  return LGH_OK;
}

}  // extern "C"
