// engine.hip — GPU-resident decode engine: weight store, KV cache, per-token launch sequence,
// hipGraph replay.  Mirrors what the reference's GpuOnlyInference does around its kernels
// (src/backend/cuda/gpu_only.rs:426-1024) with an MI355X-first structure:
//
//   reference (per token)                                   here
//   ------------------------------------------------------- ---------------------------------------------
//   H2D embedding row from a host f32 table (849-858)        row dequantized on device from the quantized table
//   ~20 driver calls per layer, no graph (860-1024)          5-6 launches per layer, one hipGraph replay per token
//   norm, QKV x3, dtod, rope, kv write (865-875,1056-1281)   ONE launch: RMSNorm prologue + QKV + RoPE + cache write
//   add + dtod after wo / down (969-978, 1012-1021)          residual add is the mat-vec epilogue, in place
//   gate, up, silu(+alloc+dtod), mul (1605-1651)             ONE launch with a SwiGLU epilogue
//   MoE: D2H router logits, host top-k, 3 expert matrices    router + top-k on device; experts resident in HBM and
//   re-uploaded per expert per layer per token (1765-2011)   selected by a device-side index
//   D2H full logits every token (764-767)                    kept for lgh_forward; lgh_decode_greedy feeds the
//                                                            arg-max back on device
//   reset() zero-fills every cache over PCIe (808-843)       O(1): position rewind
#include "engine.h"
#include "prefill.h"
#include "xq.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace lgh;

namespace lgh {
hipError_t kv_store_launch(const float* k, const float* v, float* kcache, float* vcache, uint32_t n_kv, uint32_t d,
                           uint32_t max_seq, const int* pos, hipStream_t st);
}

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
int fail(lgh_ctx* c, int status, const std::string& msg) {
  if (c) c->err = msg;
  return status;
}

#define HIP_TRY(c, status, expr)                                                                      \
  do {                                                                                                \
    hipError_t e__ = (expr);                                                                          \
    if (e__ != hipSuccess)                                                                            \
      return fail((c), (status), std::string(#expr) + ": " + hipGetErrorString(e__));                 \
  } while (0)

int dev_alloc(lgh_ctx* c, void** p, size_t bytes) {
  if (bytes == 0) bytes = 4;
  hipError_t e = hipMalloc(p, bytes);
  if (e != hipSuccess) return fail(c, LGH_ALLOCATION_FAILED, std::string("hipMalloc(") + std::to_string(bytes) + "): " + hipGetErrorString(e));
  c->allocs.push_back(*p);
  return LGH_OK;
}

static void dev_free_tracked(lgh_ctx* c, void* p) {
  for (auto& a : c->allocs)
    if (a == p) { a = c->allocs.back(); c->allocs.pop_back(); break; }
  (void)hipFree(p);
}

// Contexts up to this many rows run the single-launch decode attention (attention.hip, DIRECT) unless the flags say
// otherwise.  Measured on Llama-3-8B Q4_K_M: 640 vs 618 tokens/s at kv <= 64, equal at kv 69..128, 581 vs 614 at kv
// 137..272 — one workgroup per kv head fetches that head's whole K/V (1 KB per row) through ONE CU's memory path.
constexpr uint32_t kDirectAttnDefaultKv = 64;
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ------------------------------------------------------------------------------------------------
// device layouts
// ------------------------------------------------------------------------------------------------
LayoutInfo layout_for(int src_type) {
  switch (src_type) {
    case LGH_TYPE_Q4_K: return {LGH_TYPE_Q4_K, 1, {144, 0, 0, 0}, 256};
    case LGH_TYPE_Q5_K: return {LGH_TYPE_Q5_K, 1, {176, 0, 0, 0}, 256};
    case LGH_TYPE_Q6_K: return {LGH_TYPE_Q6_K, 4, {128, 64, 16, 2}, 256};
    case LGH_TYPE_Q8_0: return {LGH_TYPE_Q8_0, 2, {32, 2, 0, 0}, 32};
    case LGH_TYPE_Q4_0: return {LGH_TYPE_Q4_0, 2, {16, 2, 0, 0}, 32};
    default: return {LGH_TYPE_F32, 1, {4, 0, 0, 0}, 1};
  }
}

bool fused_type(int t) {
  return mfma_type(t) || t == LGH_TYPE_Q4_K || t == LGH_TYPE_Q5_K || t == LGH_TYPE_Q6_K || t == LGH_TYPE_Q8_0 || t == LGH_TYPE_Q4_0;
}

// Upload one matrix (or expert `slot` of a stack, or the whole stack when slot < 0) given in native
// GGUF order.  The first call for a DevWeight allocates it.
int upload_matrix(lgh_ctx* c, DevWeight& W, int src_type, uint32_t k, uint32_t n, uint32_t n_stack, int slot,
                         const void* host, size_t nbytes) {
  const uint32_t sbe = blk_elems(src_type), sbb = blk_bytes(src_type);
  if (!sbe) return fail(c, LGH_UNSUPPORTED_DTYPE, "unsupported ggml type " + std::to_string(src_type));
  if (k % sbe) return fail(c, LGH_SHAPE_MISMATCH, "in_features not a multiple of the block size");
  LayoutInfo li = layout_for(src_type);
  const uint64_t per_expert_src = (uint64_t)n * (k / sbe) * sbb;
  const uint32_t n_in_payload = slot < 0 ? n_stack : 1;
  if (nbytes != per_expert_src * n_in_payload) return fail(c, LGH_SHAPE_MISMATCH, "tensor byte size does not match its shape");
  uint64_t blocks_per_expert = (uint64_t)n * (k / li.belems);
  // int8-MFMA tile layouts (16 rows x 256 elements, rows padded to 16) for the five fused formats when k allows it
  const bool t16 = k % 256 == 0 && (src_type == LGH_TYPE_Q4_K || src_type == LGH_TYPE_Q6_K || src_type == LGH_TYPE_Q5_K ||
                                    src_type == LGH_TYPE_Q8_0 || src_type == LGH_TYPE_Q4_0);
  if (t16) {
    li.nplanes = 1;
    li.belems = 256;
    li.bpb[1] = li.bpb[2] = li.bpb[3] = 0;
    switch (src_type) {   // bytes per row-block = tile bytes / 16
      case LGH_TYPE_Q4_K: li.dev_type = kDevQ4K_T16; li.bpb[0] = 144; break;
      case LGH_TYPE_Q6_K: li.dev_type = kDevQ6K_T16; li.bpb[0] = 212; break;
      case LGH_TYPE_Q5_K: li.dev_type = kDevQ5K_T16; li.bpb[0] = 176; break;
      case LGH_TYPE_Q8_0: li.dev_type = kDevQ80_T16; li.bpb[0] = 272; break;
      default: li.dev_type = kDevQ40_T16; li.bpb[0] = 144; break;
    }
    blocks_per_expert = (uint64_t)((n + 15) / 16) * 16 * (k / 256);
  }
  if (!W.present()) {
    uint64_t off = 0;
    uint64_t plane_off[4] = {0, 0, 0, 0};
    for (int p = 0; p < li.nplanes; p++) {
      plane_off[p] = off;
      W.stack_stride[p] = blocks_per_expert * li.bpb[p];
      off = align_up(off + W.stack_stride[p] * n_stack, 256);
    }
    void* base = nullptr;
    int rc = dev_alloc(c, &base, off + 256);
    if (rc) return rc;
    W.base = (uint8_t*)base;
    for (int p = 0; p < li.nplanes; p++) W.plane[p] = W.base + plane_off[p];
    W.type = li.dev_type;
    W.src_type = src_type;
    W.k = k; W.n = n; W.n_stack = n_stack;
    W.bytes = 0;
    for (int p = 0; p < li.nplanes; p++) W.bytes += W.stack_stride[p];  // per expert
    if (t16) W.bytes = per_expert_src;                                  // algorithmic bytes exclude the row padding
    c->stats.weight_bytes += (li.dev_type == LGH_TYPE_F32 ? (uint64_t)n * k * 4 : per_expert_src) * n_stack;
    W.filled.assign(n_stack, false);
  } else if (W.src_type != src_type || W.k != k || W.n != n || W.n_stack != n_stack) {
    return fail(c, LGH_SHAPE_MISMATCH, "expert tensors of one stack differ in type or shape");
  }
  const uint32_t e0 = slot < 0 ? 0 : (uint32_t)slot;
  for (uint32_t i = 0; i < n_in_payload; i++) W.filled[e0 + i] = true;   // lgh_finalize refuses a stack with an empty slot
  if (li.dev_type == src_type && li.nplanes == 1 && !t16) {  // native layout: straight copy
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy((void*)(W.plane[0] + (uint64_t)e0 * W.stack_stride[0]), host, nbytes, hipMemcpyHostToDevice));
    return LGH_OK;
  }
  void* raw = nullptr;
  HIP_TRY(c, LGH_ALLOCATION_FAILED, hipMalloc(&raw, nbytes));
  hipError_t e = hipMemcpy(raw, host, nbytes, hipMemcpyHostToDevice);
  for (uint32_t i = 0; i < n_in_payload && e == hipSuccess; i++) {
    const uint8_t* src = (const uint8_t*)raw + (uint64_t)i * per_expert_src;
    if (li.dev_type == LGH_TYPE_F32) {
      e = dequant_launch(src_type, src, (float*)(W.plane[0] + (uint64_t)(e0 + i) * W.stack_stride[0]), (uint64_t)n * k, c->stream);
    } else if (t16) {
      e = repack_t16_launch(li.dev_type, src, W.base + (uint64_t)(e0 + i) * W.stack_stride[0], n, k / 256, c->stream);
    } else {
      uint64_t po[4];
      for (int p = 0; p < 4; p++) po[p] = (uint64_t)(W.plane[p] - W.base) + (uint64_t)(e0 + i) * W.stack_stride[p];
      e = repack_launch(src_type, src, W.base, po, blocks_per_expert, c->stream);
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(raw);
  if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("weight re-layout: ") + hipGetErrorString(e));
  return LGH_OK;
}

// 1-D tensors (norm weights, biases) and the f32 router matrix: always f32 on device
int upload_f32(lgh_ctx* c, float** dst, int src_type, uint64_t n, const void* host, size_t nbytes) {
  const uint32_t sbe = blk_elems(src_type), sbb = blk_bytes(src_type);
  if (!sbe || n % sbe || nbytes != n / sbe * sbb) return fail(c, LGH_SHAPE_MISMATCH, "vector byte size does not match its shape");
  if (!*dst) {
    int rc = dev_alloc(c, (void**)dst, n * 4);
    if (rc) return rc;
    c->stats.weight_bytes += n * 4;
  }
  if (src_type == LGH_TYPE_F32) {
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy(*dst, host, nbytes, hipMemcpyHostToDevice));
    return LGH_OK;
  }
  void* raw = nullptr;
  HIP_TRY(c, LGH_ALLOCATION_FAILED, hipMalloc(&raw, nbytes));
  hipError_t e = hipMemcpy(raw, host, nbytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = dequant_launch(src_type, (const uint8_t*)raw, *dst, n, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(raw);
  if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("vector upload: ") + hipGetErrorString(e));
  return LGH_OK;
}

int drain_prof(lgh_ctx* c) {
  if (c->prof.empty()) return LGH_OK;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  for (auto& r : c->prof) {
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess && r.cls < 0) {
      const double n = (double)c->stats.event_bracket_samples;
      c->stats.event_bracket_us = (c->stats.event_bracket_us * n + (double)ms * 1000.0) / (n + 1.0);
      c->stats.event_bracket_samples += 1;
    } else if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      c->stats.k_time_us[r.cls] += (double)ms * 1000.0;
      c->stats.k_launches[r.cls] += 1;
      c->stats.k_alg_bytes[r.cls] += r.bytes;
      c->stats.sym_time_us[r.sym] += (double)ms * 1000.0;
      c->stats.sym_launches[r.sym] += 1;
      c->stats.sym_alg_bytes[r.sym] += r.bytes;
    }
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  c->prof.clear();
  return LGH_OK;
}

// ------------------------------------------------------------------------------------------------
// fused mat-vec launch assembly
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// XQ images of activation buffers (xq.h)
// ------------------------------------------------------------------------------------------------
XqBuf* xq_get(lgh_ctx* c, const float* f32, uint32_t k) {
  for (auto& q : c->xqs)
    if (q.f32 == f32 && q.k >= k) return &q;
  XqBuf q;
  q.f32 = f32;
  q.k = k;
  if (dev_alloc(c, (void**)&q.xq, xq_bytes(k)) || dev_alloc(c, (void**)&q.ssq, (size_t)(k / 16 + 64) * 4)) return nullptr;
  c->xqs.push_back(q);
  return &c->xqs.back();
}
static XqBuf* xq_find(lgh_ctx* c, const float* f32) {
  for (auto& q : c->xqs)
    if (q.f32 == f32) return &q;
  return nullptr;
}
void xq_stale(lgh_ctx* c, const float* f32) {
  if (XqBuf* q = xq_find(c, f32)) q->fresh = false;
}

static uint32_t g_launch_seq = 0;   // diagnostic builds: consecutive launches get consecutive span slots
// Assembles the launch descriptor of one group of segments (and keeps the XQ bookkeeping: images this launch consumes are
// converted here if their producer did not leave them; images it produces are marked fresh).
// `tile_cap` (multi-sequence launches, engine_batch.hip): at most that many 16-row tiles per workgroup — the partial sums of
// every sequence of the step must fit LDS.  It changes which workgroup computes a row, never the row's arithmetic.
int build_mv_group(lgh_ctx* c, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k, bool mfma, MvLaunch& L,
                   uint32_t& wg, uint32_t& threads, uint64_t& alg, uint32_t tile_cap) {
  std::memset(&L, 0, sizeof(L));
  L.nseg = nseg;
  L.k = k;
  L.do_norm = norm_w != nullptr;
  L.eps = c->d.norm_eps;
  L.norm_w = norm_w;
  L.pos = c->state + ST_POS;
  L.rope_cs = c->rope_cs;
  L.dbg_slot = g_launch_seq++ & 63u;
  wg = 0; threads = 0; alg = 0;
  uint32_t launch_rows = 0;
  uint32_t wave_cap = 16;
  for (int s = 0; s < nseg; s++) {
    launch_rows += specs[s].W[0]->n;
    if (!mfma) wave_cap = std::min(wave_cap, mv_wave_cap(specs[s].W[0]->type));
  }
  if (!mfma && nseg > 1) {  // mixed-type launches run in the 512-thread instantiations
    for (int s = 1; s < nseg; s++)
      if (specs[s].W[0]->type != specs[0].W[0]->type) wave_cap = std::min(wave_cap, 8u);
  }
  // A fused launch over formats with different bytes per tile (the "_M" mixes: Q and K in Q4_K, V in Q6_K) is as long as its
  // heaviest workgroup: the segments in the heavier format get fewer tiles per workgroup, as long as the launch still fits one
  // workgroup per CU.  (Llama-3-8B QKV: 2 / 2 / 1 tiles -> 224 workgroups whose heaviest streams 2 x 2304 B per block instead
  // of 192 whose heaviest streams 2 x 3392.)
  uint32_t force_tiles[3] = {0, 0, 0};
  if (mfma && nseg > 1) {
    uint32_t R[3], Gs[3], tiles[3], w[3];
    bool ok = true, mixed = false;
    for (int s = 0; s < nseg && ok; s++) {
      const DevWeight& W0 = *specs[s].W[0];
      MvPlan p;
      ok = mvq_plan(W0.k, W0.n, specs[s].npass, &p, launch_rows) == hipSuccess;
      R[s] = p.rows_per_wg / 16; Gs[s] = p.G; tiles[s] = (W0.n + 15) / 16;
      w[s] = mvq_tile_bytes(W0.type) * (uint32_t)specs[s].npass;
      ok = ok && w[s] != 0;
      mixed = mixed || w[s] != w[0];
    }
    for (int it = 0; ok && mixed && it < 32; it++) {
      int h = 0;
      for (int s = 1; s < nseg; s++)
        if ((uint64_t)R[s] * w[s] > (uint64_t)R[h] * w[h]) h = s;
      if (R[h] < 2 * Gs[h]) break;                                    // (a workgroup keeps at least one tile per row group)
      uint32_t wgs = 0;
      for (int s = 0; s < nseg; s++) { const uint32_t r = s == h ? R[h] - Gs[h] : R[s]; wgs += (tiles[s] + r - 1) / r; }
      if (wgs > (uint32_t)kNumCU) break;
      R[h] -= Gs[h];
    }
    if (ok && mixed)
      for (int s = 0; s < nseg; s++) force_tiles[s] = R[s];
  }
  if (mfma && tile_cap) {
    for (int s = 0; s < nseg; s++) {
      const DevWeight& W0 = *specs[s].W[0];
      MvPlan p;
      if (mvq_plan(W0.k, W0.n, specs[s].npass, &p, launch_rows, force_tiles[s]) != hipSuccess) continue;
      if (p.rows_per_wg / 16 > tile_cap) force_tiles[s] = std::max(p.G, tile_cap / p.G * p.G);
    }
  }
  for (int s = 0; s < nseg; s++) {
    const SegSpec& sp = specs[s];
    const DevWeight& W0 = *sp.W[0];
    MvPlan plan;
    hipError_t pe = mfma ? mvq_plan(W0.k, W0.n, sp.npass, &plan, launch_rows, force_tiles[s])
                         : mv_plan(W0.type, W0.k, W0.n, sp.npass, &plan, launch_rows, wave_cap);
    if (pe != hipSuccess)
      return fail(c, LGH_UNSUPPORTED, "no fused mat-vec plan for type " + std::to_string(W0.type) + " k=" + std::to_string(W0.k));
    MvSeg& S = L.seg[s];
    S.type = W0.type;
    S.epi = sp.epi;
    S.n_rows = W0.n;
    S.nblk = mfma ? W0.k / 256 : W0.k / layout_for(W0.src_type).belems;
    S.units = plan.units; S.T = plan.T; S.G = plan.G;
    S.rows_per_wg = plan.rows_per_wg;
    S.wg_begin = wg;
    S.npass = sp.npass;
    for (int p = 0; p < sp.npass; p++) {
      const DevWeight& W = *sp.W[p];
      if (W.type != W0.type || W.k != W0.k || W.n != W0.n) return fail(c, LGH_SHAPE_MISMATCH, "passes of one segment differ in type/shape");
      for (int i = 0; i < 4; i++) { S.pass[p].plane[i] = W.plane[i]; S.pass[p].sel_stride[i] = W.stack_stride[i]; }
      S.pass[p].x = sp.x[p];
      S.pass[p].sel = sp.sel[p];
      if (mfma) {   // the input vector as XQ records: left by its producer, or converted here
        XqBuf* q = xq_get(c, sp.x[p], k);
        if (!q) return fail(c, LGH_ALLOCATION_FAILED, "XQ image allocation failed");
        if (!q->fresh || q->tag != norm_w) {
          int rq = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, (uint64_t)k * 4, [&] {
            return xq_quantize_launch(sp.x[p], norm_w, q->xq, norm_w ? q->ssq : nullptr, k, c->stream);
          });
          if (rq) return rq;
          q->fresh = true;
          q->tag = norm_w;
        }
        S.pass[p].xq = q->xq;
        if (norm_w) { L.ssq_part = q->ssq; L.n_ssq_part = k / 16; }
      }
      alg += W.bytes;
    }
    S.out = sp.out; S.out2 = sp.out2; S.resid = sp.resid; S.bias = sp.bias; S.moe_w = sp.moe_w;
    {  // XQ image of the output for the next consumer, where this epilogue can write one
      XqBuf* qo = sp.out ? xq_find(c, sp.out) : nullptr;
      const bool can = sp.xq_next && qo && W0.n % 16 == 0 && W0.n <= qo->k && plan.rows_per_wg % 16 == 0 &&   // thread t <-> row t, chunk-aligned
                      
                       (sp.epi == EPI_STORE || sp.epi == EPI_RESID || sp.epi == EPI_SWIGLU || sp.epi == EPI_MOE_DOWN);
      if (can) {
        S.xq_out = qo->xq;
        S.xq_nw = sp.xq_next == 2 ? sp.xq_next_nw : nullptr;
        S.xq_ssq = sp.xq_next == 2 ? qo->ssq : nullptr;
        qo->fresh = true;
        qo->tag = S.xq_nw;
      } else if (sp.xq_next && qo && sp.epi == EPI_MOE_SWIGLU && W0.n % 16 == 0 && W0.n <= qo->k && plan.rows_per_wg % 16 == 0) {
        XqBuf* q2 = sp.out2 ? xq_find(c, sp.out2) : nullptr;
        S.xq_out = qo->xq;
        qo->fresh = true; qo->tag = nullptr;
        if (q2 && sp.npass > 2) { S.xq_out2 = q2->xq; q2->fresh = true; q2->tag = nullptr; }
      } else if (qo && sp.epi != EPI_ROPE_K && sp.epi != EPI_V_CACHE) {
        qo->fresh = false;
      }
      if (sp.out2 && !S.xq_out2) xq_stale(c, sp.out2);
    }
    S.head_dim = c->d.head_dim;
    S.max_seq = c->d.max_seq_len;
    wg += plan.n_wg;
    if (plan.threads > threads) threads = plan.threads;
    if (plan.red_floats > L.red_floats) L.red_floats = plan.red_floats;
  }
  alg += (uint64_t)k * 4 * (norm_w ? 2 : 1);
  return LGH_OK;
}

static int mvq_symbol(const MvLaunch& L) {
  bool has[8] = {false, false, false, false, false, false, false, false};
  for (int s = 0; s < L.nseg; s++) {
    const int t = L.seg[s].type;
    has[t == kDevQ4K_T16 ? 0 : t == kDevQ6K_T16 ? 1 : t == kDevQ5K_T16 ? 2 : t == kDevQ80_T16 ? 3 : 4] = true;
  }
  return has[2] ? LGH_SYM_MVQ_Q5K : (has[3] || has[4]) ? LGH_SYM_MVQ_Q80_Q40 : (has[0] && has[1]) ? LGH_SYM_MVQ_MIXED
         : has[1] ? LGH_SYM_MVQ_Q6K : LGH_SYM_MVQ_Q4K;
}

static int launch_mv_group(lgh_ctx* c, int cls, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k, bool mfma) {
  MvLaunch L;
  uint32_t wg, threads;
  uint64_t alg;
  int rc = build_mv_group(c, specs, nseg, norm_w, k, mfma, L, wg, threads, alg, 0);
  if (rc) return rc;
  if (mfma) return run_k(c, cls, mvq_symbol(L), alg, [&] { return mvq_launch(L, wg, threads, c->stream); });
  return run_k(c, cls, mv_symbol(L), alg, [&] { return mv_launch(L, wg, threads, c->stream); });
}

// Segments are independent (disjoint outputs), so a launch whose matrices live in different kernel families
// (Q4_K on the matrix cores, the rest on the VALU kernel) is issued as one launch per family.
int launch_mv(lgh_ctx* c, int cls, const SegSpec* specs, int nseg, const float* norm_w, uint32_t k) {
  SegSpec a[3], b[3];
  int na = 0, nb = 0;
  for (int s = 0; s < nseg; s++) {
    if (mfma_type(specs[s].W[0]->type)) a[na++] = specs[s];
    else b[nb++] = specs[s];
  }
  int rc = LGH_OK;
  if (na > 1) {   // formats that have a common instantiation: Q4_K+Q6_K, Q5_K+Q6_K (the "_M" mixes); otherwise one launch each
    bool q4 = false, q5 = false, other = false;
    for (int s = 0; s < na; s++) {
      const int t = a[s].W[0]->type;
      q4 |= t == kDevQ4K_T16; q5 |= t == kDevQ5K_T16; other |= t == kDevQ80_T16 || t == kDevQ40_T16;
    }
    bool uniform = true;
    for (int s = 1; s < na; s++) uniform &= a[s].W[0]->type == a[0].W[0]->type;
    if (!uniform && ((q4 && q5) || other)) {
      for (int s = 0; s < na; s++)
        if ((rc = launch_mv_group(c, cls, a + s, 1, norm_w, k, true))) return rc;
      na = 0;
    }
  }
  if (na && (rc = launch_mv_group(c, cls, a, na, norm_w, k, true))) return rc;
  if (nb && (rc = launch_mv_group(c, cls, b, nb, norm_w, k, false))) return rc;
  return rc;
}

// one Linear with optional norm prologue / residual epilogue, any device type
int linear_any(lgh_ctx* c, int cls, const DevWeight& W, const float* x, float* out, const float* norm_w,
                      const float* resid, const float* bias, int xq_next, const float* xq_next_nw) {
  if (fused_type(W.type)) {
    SegSpec sp;
    sp.W[0] = &W; sp.x[0] = x;
    sp.epi = resid ? EPI_RESID : EPI_STORE;
    sp.out = out; sp.resid = resid; sp.bias = bias;
    sp.xq_next = xq_next; sp.xq_next_nw = xq_next_nw;
    return launch_mv(c, cls, &sp, 1, norm_w, W.k);
  }
  xq_stale(c, out);
  if (bias) return fail(c, LGH_UNSUPPORTED, "bias on a non-quantized linear layer is not supported");
  return run_k(c, cls, LGH_SYM_F32_MATVEC, (uint64_t)W.n * W.k * 4, [&] {
    return f32_matvec_launch((const float*)W.plane[0], x, out, W.k, W.n, norm_w, c->d.norm_eps, resid, c->stream);
  });
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// the FFN half of a layer on one sequence's vectors: FeedForward::forward (layers.rs:908-929) or MoeLayer::forward
// (moe.rs:321-413), residual included.  The single-sequence path passes the context's own buffers; the multi-sequence path
// (engine_batch.hip) runs MoE layers through here sequence by sequence — every sequence selects its own experts.
// ------------------------------------------------------------------------------------------------
int ffn_forward(lgh_ctx* c, LayerW& Lw, const FfnView& v, const float* next_nw, bool next_mfma) {
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size;
  int rc;
  // ---- FFN
  if (!Lw.moe()) {
    if (fused_type(Lw.gate.type) && Lw.gate.type == Lw.up.type) {  // FeedForward::forward (layers.rs:908-929)
      SegSpec sp;
      sp.npass = 2;
      sp.W[0] = &Lw.gate; sp.W[1] = &Lw.up;
      sp.x[0] = sp.x[1] = v.hidden;
      sp.epi = EPI_SWIGLU;
      sp.out = v.act;
      sp.xq_next = mfma_type(Lw.down.type) ? 1 : 0;
      if ((rc = launch_mv(c, LGH_K_GATEUP, &sp, 1, Lw.ffn_norm, H))) return rc;
    } else {
      if ((rc = linear_any(c, LGH_K_GATEUP, Lw.gate, v.hidden, v.act, Lw.ffn_norm, nullptr, nullptr))) return rc;
      if ((rc = linear_any(c, LGH_K_GATEUP, Lw.up, v.hidden, v.act2, Lw.ffn_norm, nullptr, nullptr))) return rc;
      if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] { return silu_mul_launch(v.act, v.act2, v.act, Lw.gate.n, c->stream); }))) return rc;
      xq_stale(c, v.act);
    }
    return linear_any(c, LGH_K_DOWN, Lw.down, v.act, v.hidden, nullptr, v.hidden, nullptr, next_mfma ? 2 : 0, next_nw);
  }
  // ---- MoE (moe.rs:321-413): router + top-k on device, experts selected by device-side index
  const uint32_t topk = d.num_experts_per_token;
  if ((rc = run_k(c, LGH_K_ROUTER, LGH_SYM_ROUTER, (uint64_t)d.num_experts * H * 4, [&] {
         return moe_router_launch(v.hidden, Lw.ffn_norm, d.norm_eps, Lw.router, H, d.num_experts, topk, v.moe_sel, v.moe_w, c->stream);
       })))
    return rc;
  if (!fused_type(Lw.gate_exps.type) || Lw.gate_exps.type != Lw.up_exps.type || !fused_type(Lw.down_exps.type) || topk > 8)
    return fail(c, LGH_UNSUPPORTED, "MoE needs fused-format experts and top-k <= 8");
  // The selected experts run two at a time (a launch carries up to four passes: gate and up of two experts).  Every group
  // reads the SAME normalised h, so the running sum lives in a scratch vector until the last group writes the residual
  // stream: tmp = 0 + w0 e0 + w1 e1; tmp = tmp + w2 e2 + w3 e3; ...; h = (tmp + ...) + h — moe.rs:363-368's order exactly:
  // one sum over the weighted expert outputs in selection order, then the residual.
  for (uint32_t g0 = 0; g0 < topk; g0 += 2) {
    const uint32_t ng = std::min(2u, topk - g0);
    const bool first_g = g0 == 0, last_g = g0 + ng >= topk;
    {
      SegSpec sp;
      sp.npass = (int)(2 * ng);
      for (uint32_t s = 0; s < ng; s++) {
        sp.W[2 * s] = &Lw.gate_exps; sp.W[2 * s + 1] = &Lw.up_exps;
        sp.x[2 * s] = sp.x[2 * s + 1] = v.hidden;
        sp.sel[2 * s] = sp.sel[2 * s + 1] = v.moe_sel + g0 + s;
      }
      sp.epi = EPI_MOE_SWIGLU;
      sp.out = v.act; sp.out2 = v.act2;
      sp.xq_next = mfma_type(Lw.down_exps.type) ? 1 : 0;
      if ((rc = launch_mv(c, LGH_K_GATEUP, &sp, 1, Lw.ffn_norm, H))) return rc;
    }
    {
      SegSpec sp;
      sp.npass = (int)ng;
      for (uint32_t s = 0; s < ng; s++) {
        sp.W[s] = &Lw.down_exps;
        sp.x[s] = s == 0 ? v.act : v.act2;
        sp.sel[s] = v.moe_sel + g0 + s;
      }
      sp.epi = EPI_MOE_DOWN;
      sp.out = last_g ? v.hidden : v.xnorm;
      sp.out2 = first_g ? nullptr : v.xnorm;     // (EPI_MOE_DOWN: the running sum of the earlier groups)
      sp.resid = last_g ? v.hidden : nullptr;
      sp.moe_w = v.moe_w + g0;
      sp.xq_next = last_g && next_mfma ? 2 : 0; sp.xq_next_nw = next_nw;
      if ((rc = launch_mv(c, LGH_K_DOWN, &sp, 1, nullptr, Lw.down_exps.k))) return rc;
    }
  }
  return LGH_OK;
}

// ------------------------------------------------------------------------------------------------
// one transformer layer (TransformerLayer::forward serial-residual branch, layers.rs:1187-1244)
// ------------------------------------------------------------------------------------------------
// `next_nw` / `next_mfma`: the norm weights and kernel family of whatever consumes this layer's output (the next layer's
// QKV, or the output projection)
static int layer_forward(lgh_ctx* c, uint32_t li, const float* next_nw, bool next_mfma, int mode) {
  LayerW& Lw = c->layers[li];
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size;
  int rc;
  // ---- attention: norm -> q,k,v -> rope -> cache write (layers.rs:438-600)
  const bool fused_qkv = fused_type(Lw.wq.type) && fused_type(Lw.wk.type) && fused_type(Lw.wv.type) && !d.use_neox_rope;
  const bool kv8 = (d.flags & LGH_FLAG_KV_INT8) != 0;
  float* const k_new = c->kv_tmp;                                           // int8 cache: the current token's rotated K row ...
  float* const v_new = c->kv_tmp + (size_t)d.num_kv_heads * d.head_dim;     // ... and V row, f32, quantized by the attention launch
  if (fused_qkv) {
    SegSpec sp[3];
    sp[0].W[0] = &Lw.wq; sp[0].x[0] = c->hidden; sp[0].epi = EPI_ROPE_Q; sp[0].out = c->q; sp[0].bias = Lw.bq;
    sp[1].W[0] = &Lw.wk; sp[1].x[0] = c->hidden; sp[1].epi = kv8 ? EPI_ROPE_Q : EPI_ROPE_K; sp[1].out = kv8 ? k_new : Lw.kcache; sp[1].bias = Lw.bk;
    sp[2].W[0] = &Lw.wv; sp[2].x[0] = c->hidden; sp[2].epi = kv8 ? EPI_STORE : EPI_V_CACHE; sp[2].out = kv8 ? v_new : Lw.vcache; sp[2].bias = Lw.bv;
    if ((rc = launch_mv(c, LGH_K_QKV, sp, 3, Lw.attn_norm, H))) return rc;
  } else {
    float* kt = c->kv_tmp;
    float* vt = c->kv_tmp + (size_t)d.num_kv_heads * d.head_dim;
    if ((rc = linear_any(c, LGH_K_QKV, Lw.wq, c->hidden, c->q, Lw.attn_norm, nullptr, Lw.bq))) return rc;
    if ((rc = linear_any(c, LGH_K_QKV, Lw.wk, c->hidden, kt, Lw.attn_norm, nullptr, Lw.bk))) return rc;
    if ((rc = linear_any(c, LGH_K_QKV, Lw.wv, c->hidden, vt, Lw.attn_norm, nullptr, Lw.bv))) return rc;
    xq_stale(c, c->q);
    if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] {
           return rope_launch(c->q, kt, d.num_heads, d.num_kv_heads, d.head_dim, c->state + ST_POS, c->rope_cs, (int)d.use_neox_rope, c->stream);
         })))
      return rc;
    if (!kv8 && (rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] {
           return kv_store_launch(kt, vt, Lw.kcache, Lw.vcache, d.num_kv_heads, d.head_dim, d.max_seq_len, c->state + ST_POS, c->stream);
         })))
      return rc;
  }
  // ---- attention_cached (ops.rs:1479-1537)
  const float scale = 1.0f / std::sqrt((float)d.head_dim);  // layers.rs:374
  const uint64_t kv_bytes = (uint64_t)2 * d.num_kv_heads * (c->pos + 1) *
                            (d.kv_cache_type == LGH_KV_INT8 ? d.head_dim + 4 : kv8 ? d.head_dim : d.head_dim * 4);
  if (kv_is_tq(d.kv_cache_type)) {
    // TurboQuantKVCache (kv_turboquant.rs): write_kv + attention_layer over the codes; the merge also inverts the V rotation
    const int bits = kv_tq_bits(d.kv_cache_type);
    const float* signs = c->tq_signs + (size_t)(li - c->l0) * d.num_kv_heads * 2 * d.head_dim;
    const bool qjl = kv_is_qjl(d.kv_cache_type);
    const float* qjl_s = qjl ? c->tq_qjl + (size_t)(li - c->l0) * d.num_kv_heads * d.head_dim * d.head_dim : nullptr;
    const uint64_t tq_bytes = (uint64_t)d.num_kv_heads * (c->pos + 1) * (2 * tq_row_bytes_host(bits, d.head_dim) + (qjl ? d.head_dim / 8 + 4 : 0)) +
                              (qjl ? (uint64_t)d.num_kv_heads * d.head_dim * d.head_dim * 4 : 0);
    if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, tq_bytes, [&] {
           return attn_tq_launch(bits, c->q, (uint8_t*)Lw.k8, (uint8_t*)Lw.v8, k_new, v_new, signs, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len,
                                 scale, c->state + ST_POS, c->n_splits, c->part_ml, c->part_acc, c->stream, qjl_s, qjl ? Lw.kx : nullptr);
         })))
      return rc;
    XqBuf* qa = mfma_type(Lw.wo.type) ? xq_get(c, c->attn_out, d.num_heads * d.head_dim) : nullptr;
    if ((rc = run_k(c, LGH_K_ATTN_COMBINE, LGH_SYM_ATTN_COMBINE, 0, [&] {
           return attn_tq_combine_launch(bits, c->part_ml, c->part_acc, signs, d.num_heads, d.num_kv_heads, d.head_dim, c->n_splits, c->attn_out,
                                         qa ? qa->xq : nullptr, c->stream);
         })))
      return rc;
    if (qa) { qa->fresh = true; qa->tag = nullptr; }
    else xq_stale(c, c->attn_out);
  } else if (kv8) {
    // int8 rows + scales (kv_quantized.rs); the launch also quantizes and stores the current token's rows
    if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, kv_bytes, [&] {
           return attn_q8_launch((int)d.kv_cache_type, c->q, Lw.k8, Lw.v8, Lw.kscale, Lw.vscale, k_new, v_new, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len,
                                 scale, c->state + ST_POS, c->n_splits, c->part_ml, c->part_acc, c->stream);
         })))
      return rc;
    XqBuf* qa = mfma_type(Lw.wo.type) ? xq_get(c, c->attn_out, d.num_heads * d.head_dim) : nullptr;
    if ((rc = run_k(c, LGH_K_ATTN_COMBINE, LGH_SYM_ATTN_COMBINE, 0, [&] {
           return attn_combine_launch(c->part_ml, c->part_acc, d.num_heads, d.num_kv_heads, d.head_dim, c->n_splits, c->attn_out, qa ? qa->xq : nullptr,
                                      c->stream);
         })))
      return rc;
    if (qa) { qa->fresh = true; qa->tag = nullptr; }
    else xq_stale(c, c->attn_out);
  } else if (!attn_shape_has_fast_kernel(d.head_dim, d.num_heads / d.num_kv_heads)) {
    if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, kv_bytes, [&] {
           return attn_decode_any_launch(c->q, Lw.kcache, Lw.vcache, c->attn_out, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len, scale,
                                         c->state + ST_POS, c->stream);
         })))
      return rc;
    xq_stale(c, c->attn_out);
  } else if (c->attn_direct) {
    XqBuf* qa = mfma_type(Lw.wo.type) ? xq_get(c, c->attn_out, d.num_heads * d.head_dim) : nullptr;
    if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, kv_bytes, [&] {
           return attn_direct_launch(c->q, Lw.kcache, Lw.vcache, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len, scale,
                                     c->state + ST_POS, c->attn_out, qa ? qa->xq : nullptr, c->stream);
         })))
      return rc;
    if (qa) { qa->fresh = true; qa->tag = nullptr; }
    else xq_stale(c, c->attn_out);
  } else {
  if ((rc = run_k(c, LGH_K_ATTN, LGH_SYM_ATTN, kv_bytes, [&] {
           return attn_launch(c->q, Lw.kcache, Lw.vcache, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len, scale,
                              c->state + ST_POS, 0, c->n_splits, c->part_ml, c->part_acc, c->stream);
         })))
      return rc;
    {
      XqBuf* qa = mfma_type(Lw.wo.type) ? xq_get(c, c->attn_out, d.num_heads * d.head_dim) : nullptr;   // wo's input as XQ, straight from the merge
      if ((rc = run_k(c, LGH_K_ATTN_COMBINE, LGH_SYM_ATTN_COMBINE, 0, [&] {
             return attn_combine_launch(c->part_ml, c->part_acc, d.num_heads, d.num_kv_heads, d.head_dim, c->n_splits, c->attn_out,
                                        qa ? qa->xq : nullptr, c->stream);
           })))
        return rc;
      if (qa) { qa->fresh = true; qa->tag = nullptr; }
      else xq_stale(c, c->attn_out);
    }
  }
  // ---- h = x + wo(attn)   (layers.rs:700-701, 1201-1208)
  const bool ffn_mfma = Lw.moe() ? mfma_type(Lw.gate_exps.type) : mfma_type(Lw.gate.type);
  if ((rc = linear_any(c, LGH_K_WO, Lw.wo, c->attn_out, c->hidden, nullptr, c->hidden, Lw.bo, ffn_mfma ? 2 : 0, Lw.ffn_norm))) return rc;
  if (c->profiling) {  // an EMPTY event bracket in mid-stream: what the measurement itself adds to every sample (at the
    // head of a token, on an idle stream, the same bracket reads differently from run to run)
    if ((rc = run_k(c, -1, -1, 0, [&] { return hipSuccess; }))) return rc;
  }
  return ffn_forward(c, Lw, FfnView{c->hidden, c->act, c->act2, c->xnorm, c->moe_sel, c->moe_w}, next_nw, next_mfma);
}

// Everything one token needs, in stream order.  Used eagerly and under graph capture.
static int enqueue_token(lgh_ctx* c, int mode) {
  const lgh_model_desc& d = c->d;
  int rc;
  for (auto& q : c->xqs) q.fresh = false;   // the residual stream is (re)written in f32 now (embedding / previous stage)
  if (c->first) {
    // the embedding row, and — when the first layer's QKV runs on the matrix cores — its XQ image with that layer's norm weights
    XqBuf* qh = nullptr;
    const float* nw0 = nullptr;
    if (c->l0 < c->l1 && mfma_type(c->layers[c->l0].wq.type) && d.hidden_size % 256 == 0) {
      qh = xq_get(c, c->hidden, d.hidden_size);
      nw0 = c->layers[c->l0].attn_norm;
    }
    if ((rc = run_k(c, LGH_K_EMBED, LGH_SYM_EMBED, (uint64_t)d.hidden_size * blk_bytes(c->embd_type) / blk_elems(c->embd_type), [&] {
           return embed_launch(c->embd_type, c->embd_raw, c->state + ST_TOKEN, c->hidden, d.hidden_size, c->state,
                               qh ? qh->xq : nullptr, nw0, qh ? qh->ssq : nullptr, c->stream);
         })))
      return rc;
    if (qh) { qh->fresh = true; qh->tag = nw0; }
  } else {
    if ((rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 0, [&] { return advance_launch(c->state, c->stream); }))) return rc;
  }
  for (uint32_t li = c->l0; li < c->l1; li++) {
    // who consumes this layer's output: the next layer's QKV (attn_norm), the output projection (output_norm), or — at a
    // pipeline-stage boundary and at the end of a prefill step — nobody on this device
    const float* next_nw = nullptr;
    bool next_mfma = false;
    if (li + 1 < c->l1) {
      next_nw = c->layers[li + 1].attn_norm;
      next_mfma = mfma_type(c->layers[li + 1].wq.type);
    } else if (c->last && mode != MODE_PREFILL) {
      next_nw = c->output_norm;
      next_mfma = mfma_type(c->output.type);
    }
    if ((rc = layer_forward(c, li, next_nw, next_mfma, mode))) return rc;
  }
  if (c->last && mode != MODE_PREFILL) {
    // compute_logits (llama.rs:247-266): final RMSNorm fused into the output projection
    if ((rc = linear_any(c, LGH_K_OUTPUT, c->output, c->hidden, c->logits, c->output_norm, nullptr, nullptr))) return rc;
    if (mode == MODE_GREEDY) {
      if ((rc = run_k(c, LGH_K_ARGMAX, LGH_SYM_ARGMAX, (uint64_t)d.vocab_size * 4, [&] {
             return argmax_launch(c->logits, d.vocab_size, c->amax_v, c->amax_i, c->state, c->tok_log, c->stream);
           })))
        return rc;
    }
  }
  // in-graph hops to a stage on the same device (lgh_stage_set_forward_targets)
  if (c->fwd_hidden && !c->last &&
      (rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, (uint64_t)d.hidden_size * 8, [&] { return copy_words_launch(c->fwd_hidden, c->hidden, d.hidden_size, c->stream); })))
    return rc;
  if (c->fwd_token && c->last && mode == MODE_GREEDY &&
      (rc = run_k(c, LGH_K_MISC, LGH_SYM_OTHER, 8, [&] { return copy_words_launch(c->fwd_token, c->state + ST_ARGMAX, 1, c->stream); })))
    return rc;
  return LGH_OK;
}

// One token through the context's kernels, launched eagerly, before any of them is first launched inside a stream
// capture.  Measured on ROCm 7.0 / MI355X: a kernel whose FIRST launch in the process happens during a capture is not
// replayed with the graph — a pipeline stage behind the first (advance + stand-alone XQ kernels, which only such stages
// use) then decoded from a stale position in a fresh process and correctly in every later context of the same process
// (tools/diag_stage_first_capture.py).  Token 0 at position 0 with whatever is in the buffers: it writes K/V
// row 0 of every layer, which the sequence's real first token overwrites before anything reads it.
static int warm_kernels(lgh_ctx* c) {
  if (c->d.flags & LGH_FLAG_NO_GRAPH) return LGH_OK;
  int rc = LGH_OK;
  const bool keep_direct = c->attn_direct;
  for (int v = 0; v < 2 && !rc; v++) {
    if (v == 1 && c->direct_attn_max_kv == 0) continue;
    c->attn_direct = v == 1;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(c->state, 0, ST_WORDS * 4, c->stream));
    for (int mode : {c->last ? MODE_GREEDY : MODE_PREFILL, MODE_PREFILL})   // the last stage: with and without the output head
      if (!rc) rc = enqueue_token(c, mode);
  }
  c->attn_direct = keep_direct;
  for (auto& q : c->xqs) q.fresh = false;
  if (rc) return rc;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(c->state, 0, ST_WORDS * 4, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

static int ensure_graph(lgh_ctx* c, int mode) {
  const int var = c->attn_direct ? 1 : 0;
  if (c->graph[mode][var]) return LGH_OK;
  hipGraph_t g = nullptr;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
  int rc = enqueue_token(c, mode);
  hipError_t e = hipStreamEndCapture(c->stream, &g);
  if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
  size_t n_nodes = 0;
  (void)hipGraphGetNodes(g, nullptr, &n_nodes);
  if (mode == MODE_GREEDY || c->graph_nodes == 0) c->graph_nodes = n_nodes;
  e = hipGraphInstantiate(&c->graph[mode][var], g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return fail(c, LGH_OPERATION_FAILED, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  return LGH_OK;
}

// run one token in `mode` (token id already in the device state)
static int step(lgh_ctx* c, int mode) {
  if (c->pos >= c->d.max_seq_len)  // the reference has no such check: OOB write past the KV capacity
    return fail(c, LGH_INVALID_ARGUMENT, "position " + std::to_string(c->pos) + " >= max_seq_len " + std::to_string(c->d.max_seq_len));
  int rc;
  c->attn_direct = c->pos + 1 <= c->direct_attn_max_kv;   // the token at position pos attends to pos + 1 rows
  if (c->profiling || (c->d.flags & LGH_FLAG_NO_GRAPH)) {
    if ((rc = enqueue_token(c, mode))) return rc;
    if (c->profiling && (rc = drain_prof(c))) return rc;
  } else {
    if ((rc = ensure_graph(c, mode))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipGraphLaunch(c->graph[mode][c->attn_direct ? 1 : 0], c->stream));
  }
  c->pos += 1;
  c->stats.tokens_processed += 1;
  return LGH_OK;
}

static void drop_graphs(lgh_ctx* c) {
  for (int m = 0; m < MODE_COUNT; m++)
    for (int v = 0; v < 2; v++)
      if (c->graph[m][v]) { (void)hipGraphExecDestroy(c->graph[m][v]); c->graph[m][v] = nullptr; }
  for (auto& gg : c->batch.graph)
    for (auto& ge : gg)
      if (ge) { (void)hipGraphExecDestroy(ge); ge = nullptr; }
}

static int bind(const lgh_ctx* c) { return hipSetDevice(c->device) == hipSuccess ? LGH_OK : LGH_NOT_AVAILABLE; }

static int set_token(lgh_ctx* c, uint32_t token) {
  if (c->first && token >= c->d.vocab_size) return fail(c, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");  // llama.rs:296-302
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_TOKEN), (int)token, 1, c->stream));
  return LGH_OK;
}

// ------------------------------------------------------------------------------------------------
// batched prompt processing (prefill.hip)
// ------------------------------------------------------------------------------------------------
static bool pf_eligible(const lgh_ctx* c) {
  const lgh_model_desc& d = c->d;
  if (d.flags & (LGH_FLAG_EXACT_PREFILL | LGH_FLAG_KV_INT8)) return false;   // (the batched path writes f32 K/V rows)
  const uint32_t QD = d.num_heads * d.head_dim, KD = d.num_kv_heads * d.head_dim, g = d.num_heads / d.num_kv_heads;
  if (d.hidden_size % 256 || d.hidden_size > 2048u * kPfSsqChunks || QD % 256 || KD % 16) return false;
  if ((d.head_dim != 64 && d.head_dim != 128) || (g != 1 && g != 2 && g != 4 && g != 8)) return false;
  for (uint32_t i = c->l0; i < c->l1; i++) {
    const LayerW& L = c->layers[i];
    for (const DevWeight* W : {&L.wq, &L.wk, &L.wv, &L.wo})
      if (!pf_supported_type(W->type) || W->n % 16) return false;
    if (L.moe()) {   // experts: tokens are grouped by expert (prefill.hip); up to 8 selected, at most 64 experts
      if (d.num_experts > (uint32_t)kPfMaxExperts || d.num_experts_per_token == 0 || d.num_experts_per_token > (uint32_t)kPfMaxTopK ||
          (uint32_t)kPfTokens * d.num_experts_per_token + 15 * d.num_experts > (uint32_t)kPfMoeRows)
        return false;
      for (const DevWeight* W : {&L.gate_exps, &L.up_exps, &L.down_exps})
        if (!pf_supported_type(W->type) || W->n % 16 || W->k % 256) return false;
    } else {
      if (d.intermediate_size % 256) return false;
      for (const DevWeight* W : {&L.gate, &L.up, &L.down})
        if (!pf_supported_type(W->type) || W->n % 16) return false;
    }
  }
  return true;
}

static int pf_ensure(lgh_ctx* c) {
  PfScratch& P = c->pf;
  if (P.ready) return LGH_OK;
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size, QD = d.num_heads * d.head_dim, KD = d.num_kv_heads * d.head_dim;
  bool any_moe = false, any_dense = false;
  for (uint32_t i = c->l0; i < c->l1; i++) (c->layers[i].moe() ? any_moe : any_dense) = true;
  const uint32_t F = any_dense ? d.intermediate_size : 0;
  const uint32_t EI = any_moe ? (d.expert_intermediate_size ? d.expert_intermediate_size : d.intermediate_size) : 0;
  const uint32_t qkv[3] = {QD, KD, KD}, one[1] = {H};
  size_t pb = pf_part_bytes(qkv, 3, H);
  pb = std::max(pb, pf_part_bytes(one, 1, QD));
  if (F) { const uint32_t gu[2] = {F, F}; pb = std::max({pb, pf_part_bytes(gu, 2, H), pf_part_bytes(one, 1, F)}); }
  if (EI) { const uint32_t gu[2] = {EI, EI}; pb = std::max({pb, pf_part_bytes(gu, 2, H, kPfMoeRows), pf_part_bytes(one, 1, EI, kPfMoeRows)}); }
  const uint32_t topk = d.num_experts_per_token ? d.num_experts_per_token : 1;
  int rc;
  struct { void** p; size_t n; } bufs[] = {
      {(void**)&P.xh_h, xh_bytes(H)},           {(void**)&P.xh_attn, xh_bytes(QD)},
      {(void**)&P.xh_act, xh_bytes(std::max(F, EI))}, {(void**)&P.hidden, (size_t)kPfTokens * H * 4},
      {(void**)&P.q, (size_t)kPfTokens * QD * 4},
      {(void**)&P.part, pb},                    {(void**)&P.tokens, (size_t)kPfTokens * 4},
      {(void**)&P.ssq, (size_t)kPfTokens * kPfSsqChunks * 4},
      {(void**)&P.moe_sel, any_moe ? (size_t)kPfTokens * topk * 4 : 0},  {(void**)&P.moe_w, any_moe ? (size_t)kPfTokens * topk * 4 : 0},
      {(void**)&P.moe_cnt, any_moe ? (size_t)kPfMaxExperts * 4 : 0},    {(void**)&P.moe_list, any_moe ? (size_t)kPfMaxExperts * kPfTokens * 4 : 0},
      {(void**)&P.moe_base, any_moe ? (size_t)kPfMaxExperts * 4 : 0},   {(void**)&P.moe_rowmap, any_moe ? (size_t)kPfMoeRows * 4 : 0},
      {(void**)&P.moe_tokmap, any_moe ? (size_t)kPfTokens * kPfMaxTopK * 4 : 0},
      {(void**)&P.xh_gather, any_moe ? xh_bytes(H) * d.num_experts : 0}, {(void**)&P.xh_act_e, any_moe ? xh_bytes(EI) * d.num_experts : 0},
  };
  for (auto& b : bufs) {
    if (!b.n) continue;
    if ((rc = dev_alloc(c, b.p, b.n))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(*b.p, 0, b.n, c->stream));
    c->stats.scratch_bytes += b.n;
  }
  // the zero-fills are done before anybody else (another stream, a peer's copy into the stage block) touches the buffers
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  P.part_bytes = pb;
  P.ready = true;
  return LGH_OK;
}

// m <= 128 prompt tokens at positions pos .. pos+m-1: fills every owned layer's K/V rows.  The first stage starts from the
// tokens' embedding rows, any other stage from the block of hidden vectors its predecessor left in pf.hidden; a stage that
// is not the last leaves its output block there (the last one stops after its final layer's K/V rows: nothing else of a
// prefill survives).
static int prefill_block(lgh_ctx* c, const uint32_t* tokens, uint32_t m) {
  int rc = pf_ensure(c);
  if (rc) return rc;
  PfScratch& P = c->pf;
  const lgh_model_desc& d = c->d;
  const uint32_t H = d.hidden_size, QD = d.num_heads * d.head_dim, KD = d.num_kv_heads * d.head_dim, F = d.intermediate_size;
  const uint32_t pos0 = (uint32_t)c->pos;
  hipStream_t st = c->stream;
  auto K = [&](hipError_t e, const char* what) -> int {
    return e == hipSuccess ? LGH_OK : fail(c, LGH_OPERATION_FAILED, std::string("batched prefill, ") + what + ": " + hipGetErrorString(e));
  };
  if (c->first) {
    // The caller's `tokens` may be freed as soon as this returns (lgh_stage_prefill_batch does not synchronise), so the ids
    // go through a context-owned PINNED buffer, one slot per position; a slot is only rewritten after the copy that last
    // read it has completed (reset + a second prompt before the first one has run).
    if (!P.tok_pinned) {
      HIP_TRY(c, LGH_ALLOCATION_FAILED, hipHostMalloc((void**)&P.tok_pinned, (size_t)d.max_seq_len * 4, hipHostMallocDefault));
      HIP_TRY(c, LGH_OPERATION_FAILED, hipEventCreateWithFlags(&P.tok_copied, hipEventDisableTiming));
    } else if (pos0 < P.tok_hi && pos0 + m > P.tok_lo) {
      // only when a slot about to be rewritten may still be read: a reset / shift / truncate followed by a new prompt.  The blocks
      // of ONE prompt use ascending slots and never wait here (lgh_stage_prefill_batch stays asynchronous).
      HIP_TRY(c, LGH_OPERATION_FAILED, hipEventSynchronize(P.tok_copied));
      P.tok_lo = P.tok_hi = 0;
    }
    if (P.tok_hi == P.tok_lo) { P.tok_lo = pos0; P.tok_hi = pos0 + m; }
    else { P.tok_lo = std::min(P.tok_lo, pos0); P.tok_hi = std::max(P.tok_hi, pos0 + m); }
    std::memcpy(P.tok_pinned + pos0, tokens, (size_t)m * 4);
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(P.tokens, P.tok_pinned + pos0, (size_t)m * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(c, LGH_OPERATION_FAILED, hipEventRecord(P.tok_copied, st));
    if ((rc = K(embed_batch_launch(c->embd_type, c->embd_raw, P.tokens, P.hidden, H, m, st), "embedding"))) return rc;
  }
  if ((rc = K(pf_row_epi_launch(nullptr, 0, 0, 0, nullptr, P.hidden, H, c->layers[c->l0].attn_norm, P.xh_h, P.ssq, m, st), "attn_norm"))) return rc;
  const float scale = 1.0f / std::sqrt((float)d.head_dim);  // layers.rs:374
  for (uint32_t li = c->l0; li < c->l1; li++) {
    LayerW& L = c->layers[li];
    uint32_t S = 0, nc = 0;
    const DevWeight* qkv[3] = {&L.wq, &L.wk, &L.wv};
    if ((rc = K(pf_gemm_launch(qkv, 3, P.xh_h, P.part, P.part_bytes, m, &S, &nc, st), "qkv GEMM"))) return rc;
    if ((rc = K(pf_qkv_epi_launch(P.part, S, nc, QD, KD, d.head_dim, L.bq, L.bk, L.bv, c->rope_cs, pos0, d.max_seq_len, P.q, L.kcache, L.vcache, P.ssq, H,
                                  d.norm_eps, (int)d.use_neox_rope, m, st),
                "qkv epilogue")))
      return rc;
    if (li + 1 == c->l1 && c->last) break;   // the model's last layer: its K/V rows are written, its output would be discarded
    if ((rc = K(attn_prefill_launch(P.q, L.kcache, L.vcache, d.num_heads, d.num_kv_heads, d.head_dim, d.max_seq_len, scale, pos0, m, P.xh_attn, st),
                "attention")))
      return rc;
    const DevWeight* wo[1] = {&L.wo};
    if ((rc = K(pf_gemm_launch(wo, 1, P.xh_attn, P.part, P.part_bytes, m, &S, &nc, st), "wo GEMM"))) return rc;
    if ((rc = K(pf_row_epi_launch(P.part, S, nc, 0, L.bo, P.hidden, H, L.ffn_norm, P.xh_h, P.ssq, m, st), "wo epilogue"))) return rc;
    const float* next_nw = li + 1 < c->l1 ? c->layers[li + 1].attn_norm : nullptr;   // nullptr: the block goes to the next stage as f32
    uint8_t* next_xh = next_nw ? P.xh_h : nullptr;
    if (!L.moe()) {
      const DevWeight* gu[2] = {&L.gate, &L.up};
      if ((rc = K(pf_gemm_launch(gu, 2, P.xh_h, P.part, P.part_bytes, m, &S, &nc, st), "gate/up GEMM"))) return rc;
      if ((rc = K(pf_swiglu_launch(P.part, S, F, P.xh_act, P.ssq, H, d.norm_eps, m, st), "SwiGLU"))) return rc;
      const DevWeight* dn[1] = {&L.down};
      if ((rc = K(pf_gemm_launch(dn, 1, P.xh_act, P.part, P.part_bytes, m, &S, &nc, st), "down GEMM"))) return rc;
      if ((rc = K(pf_row_epi_launch(P.part, S, nc, 0, nullptr, P.hidden, H, next_nw, next_xh, P.ssq, m, st), "down epilogue"))) return rc;
      continue;
    }
    // ---- MoE (moe.rs:321-413): route every token of the block (f32, the decode router), group the (token, slot) pairs by
    // expert, and run each expert once over its rows: gather -> gate|up GEMM -> SwiGLU -> down GEMM -> rows back to tokens
    const uint32_t topk = d.num_experts_per_token, EI = L.gate_exps.n;
    if ((rc = K(moe_router_launch(P.hidden, L.ffn_norm, d.norm_eps, L.router, H, d.num_experts, topk, P.moe_sel, P.moe_w, st, m), "router"))) return rc;
    if ((rc = K(pf_moe_group_launch(P.moe_sel, m, topk, d.num_experts, P.moe_cnt, P.moe_base, P.moe_list, P.moe_rowmap, P.moe_tokmap, st), "expert grouping")))
      return rc;
    if ((rc = K(pf_moe_gather_launch(P.xh_h, H, P.moe_list, P.moe_cnt, P.xh_gather, d.num_experts, st), "expert gather"))) return rc;
    for (uint32_t e = 0; e < d.num_experts; e++) {   // every expert's gate|up over its rows, partial sums side by side in one row space
      const DevWeight* gu[2] = {&L.gate_exps, &L.up_exps};
      if ((rc = K(pf_gemm_launch(gu, 2, P.xh_gather + (size_t)e * xh_bytes(H), P.part, P.part_bytes, kPfTokens, &S, &nc, st, e, P.moe_cnt + e, kPfMoeRows,
                                 P.moe_base + e),
                  "expert gate/up GEMM")))
        return rc;
    }
    if ((rc = K(pf_moe_swiglu_launch(P.part, S, EI, P.xh_act_e, P.moe_rowmap, P.moe_list, P.ssq, H, d.norm_eps, st), "expert SwiGLU"))) return rc;
    for (uint32_t e = 0; e < d.num_experts; e++) {
      const DevWeight* dn[1] = {&L.down_exps};
      if ((rc = K(pf_gemm_launch(dn, 1, P.xh_act_e + (size_t)e * xh_bytes(EI), P.part, P.part_bytes, kPfTokens, &S, &nc, st, e, P.moe_cnt + e, kPfMoeRows,
                                 P.moe_base + e),
                  "expert down GEMM")))
        return rc;
    }
    // h += sum over the selected experts, in selection order, of routing weight * expert output (moe.rs:363-368), then the
    // next layer's input
    if ((rc = K(pf_moe_combine_launch(P.part, S, P.moe_tokmap, P.moe_w, topk, P.hidden, H, next_nw, next_xh, P.ssq, m, st), "MoE combine"))) return rc;
  }
  c->pos += m;
  c->stats.tokens_processed += m;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_NEXT), (int)c->pos, 1, st));
  return LGH_OK;
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// What the engine's kernels are built for, checked BEFORE anything is allocated or uploaded (a Qwen2-7B, 28 / 4 = 7 query
// heads per kv head, used to fail only in lgh_finalize's warm-up with a generic launch error).
int engine_shape_check(const lgh_model_desc& d, std::string& why) {
  if (!d.hidden_size || !d.num_layers || !d.num_heads || !d.num_kv_heads || !d.head_dim || !d.vocab_size || !d.max_seq_len) {
    why = "a model dimension is zero";
    return LGH_INVALID_ARGUMENT;
  }
  if (d.num_heads % d.num_kv_heads || d.head_dim % 2 || d.hidden_size % 32) {
    why = "num_heads must be a multiple of num_kv_heads, head_dim even, hidden_size a multiple of 32";
    return LGH_INVALID_ARGUMENT;
  }
  const uint32_t g = d.num_heads / d.num_kv_heads;
  // head_dim 64 / 128 with 1, 2, 4, 8 query heads per kv head take the split attention kernels; every other shape takes the
  // one-workgroup-per-head kernel (attention.hip: attn_decode_any_kernel), whose scores live in LDS
  if (!attn_shape_has_fast_kernel(d.head_dim, g) && (size_t)d.max_seq_len * 4 > 150 * 1024) {
    why = "head_dim " + std::to_string(d.head_dim) + " with " + std::to_string(g) + " query heads per kv head runs the generic attention "
          "kernel, which holds max_seq_len scores in LDS: max_seq_len must be <= 38400 for it (got " + std::to_string(d.max_seq_len) + ")";
    return LGH_UNSUPPORTED;
  }
  if ((d.flags & LGH_FLAG_KV_INT8) && !attn_shape_has_fast_kernel(d.head_dim, g)) {
    why = "the int8 KV cache runs on the split attention kernels (head_dim 64 / 128; 1, 2, 4 or 8 query heads per kv head) of the default decode path";
    return LGH_UNSUPPORTED;
  }
  if (kv_is_tq(d.kv_cache_type) && (d.head_dim != 64 && d.head_dim != 128)) {
    why = "the TurboQuant KV cache rotates rows of 64 or 128 values (head_dim a power of two)";
    return LGH_UNSUPPORTED;
  }
  if (d.num_experts && (d.num_experts_per_token == 0 || d.num_experts_per_token > 8 || d.num_experts_per_token > d.num_experts || d.num_experts > 64)) {
    why = "MoE layers route top-1 .. top-8 over at most 64 experts; this model routes top-" +
          std::to_string(d.num_experts_per_token) + " over " + std::to_string(d.num_experts);
    return LGH_UNSUPPORTED;
  }
  return LGH_OK;
}

extern "C" {

int lgh_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int lgh_create(const lgh_model_desc* desc, lgh_ctx** out) {
  // (a descriptor from before kv_cache_type was appended is still taken: its cache type comes from the flags alone)
  if (!desc || !out || (desc->struct_size != sizeof(lgh_model_desc) && desc->struct_size != offsetof(lgh_model_desc, kv_cache_type)))
    return LGH_INVALID_ARGUMENT;
  *out = nullptr;
  lgh_model_desc d{};
  std::memcpy(&d, desc, desc->struct_size);
  d.struct_size = sizeof(lgh_model_desc);
  if (d.kv_cache_type > LGH_KV_TQ3_QJL) return LGH_INVALID_ARGUMENT;
  if (d.flags & LGH_FLAG_REMOVED_MASK) return LGH_UNSUPPORTED;   // the decode structures removed in round 3 (llama_gguf_hip.h)
  if (d.kv_cache_type == LGH_KV_F32 && (d.flags & LGH_FLAG_KV_INT8)) d.kv_cache_type = LGH_KV_INT8;
  // every byte-per-element cache shares the int8 cache's structure (staged f32 rows, the attention launch quantizes and stores
  // the current token's rows): the flag marks all of them from here on, kv_cache_type tells them apart
  if (d.kv_cache_type != LGH_KV_F32) d.flags |= LGH_FLAG_KV_INT8;
  std::string why;
  if (int rc = engine_shape_check(d, why)) return rc;
  int ndev = lgh_device_count();
  if (ndev <= 0 || d.device_id < 0 || d.device_id >= ndev) return LGH_NOT_AVAILABLE;
  lgh_ctx* c = new lgh_ctx();
  c->d = d;
  c->device = d.device_id;
  c->l0 = d.layer_begin;
  c->l1 = d.layer_end == 0 ? d.num_layers : d.layer_end;
  if (c->l0 >= c->l1 || c->l1 > d.num_layers) { delete c; return LGH_INVALID_ARGUMENT; }
  c->first = c->l0 == 0;
  c->last = c->l1 == d.num_layers;
  c->layers.resize(d.num_layers);
  for (uint32_t i = c->l0; i < c->l1; i++) c->layers[i].owned = true;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return LGH_INITIALIZATION_FAILED;
  }
  c->stream = c->own_stream;
  uint32_t splits = (d.flags >> LGH_FLAG_ATTN_SPLITS_SHIFT) & 0xFFu;
  if (!splits) {  // one attention workgroup per CU (measured: 605 vs 596 tokens/s at kv 272 and 470 vs 383 at kv 4000 against 128 workgroups)
    splits = 256 / d.num_kv_heads;
    if (splits < 1) splits = 1;
    if (splits > 32) splits = 32;
  }
  if (splits > 32) splits = 32;   // the split merge keeps one partial per split in registers
  c->n_splits = splits;
  const uint32_t dsel = (d.flags >> LGH_FLAG_ATTN_DIRECT_SHIFT) & 0xFFu;
  c->direct_attn_max_kv = dsel == 255 ? 0 : dsel ? dsel * 64 : kDirectAttnDefaultKv;
  if (d.flags & LGH_FLAG_KV_INT8) c->direct_attn_max_kv = 0;   // the byte caches have one attention structure: splits + combine
  *out = c;
  return LGH_OK;
}

// The sign vectors of the TurboQuant rotations (HadamardRotation::signs(), src/model/turboquant/rotation.rs:126-129), before
// lgh_finalize: [owned layer][kv head][k engine, v engine][head_dim] values of +1 / -1 — what the reference draws per engine from
// its seeds (kv_turboquant.rs:44-71: base = layer * kv_heads + head; rotation seeds 4 base and 4 base + 2).
int lgh_set_kv_rotation_signs(lgh_ctx* c, const float* signs, size_t n) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (c->finalized) return fail(c, LGH_INVALID_ARGUMENT, "the rotation signs must be given before lgh_finalize");
  if (!kv_is_tq(c->d.kv_cache_type)) return fail(c, LGH_INVALID_ARGUMENT, "this context has no TurboQuant KV cache");
  const size_t want = (size_t)(c->l1 - c->l0) * c->d.num_kv_heads * 2 * c->d.head_dim;
  if (!signs || n != want) return fail(c, LGH_INVALID_ARGUMENT, "expected " + std::to_string(want) + " sign values");
  for (size_t i = 0; i < n; i++)
    if (signs[i] != 1.0f && signs[i] != -1.0f) return fail(c, LGH_INVALID_ARGUMENT, "sign values must be +1 or -1");
  c->tq_signs_host.assign(signs, signs + n);
  return LGH_OK;
}

// The QJL projection matrices of the K engines (TurboQuantProd; QjlProjector, src/model/turboquant/qjl.rs:21-62), before lgh_finalize:
// [owned layer][kv head][head_dim][head_dim] in the order the reference draws them (row i, then column j).
int lgh_set_kv_qjl_matrices(lgh_ctx* c, const float* m, size_t n) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (c->finalized) return fail(c, LGH_INVALID_ARGUMENT, "the QJL matrices must be given before lgh_finalize");
  if (!kv_is_qjl(c->d.kv_cache_type)) return fail(c, LGH_INVALID_ARGUMENT, "this context has no TurboQuantProd (QJL) KV cache");
  const size_t want = (size_t)(c->l1 - c->l0) * c->d.num_kv_heads * c->d.head_dim * c->d.head_dim;
  if (!m || n != want) return fail(c, LGH_INVALID_ARGUMENT, "expected " + std::to_string(want) + " matrix elements");
  for (size_t i = 0; i < n; i++)
    if (!std::isfinite(m[i])) return fail(c, LGH_INVALID_ARGUMENT, "QJL matrix elements must be finite");
  c->tq_qjl_host.assign(m, m + n);
  return LGH_OK;
}

static bool parse_layer_name(const char* name, uint32_t* layer, std::string* sub) {
  if (std::strncmp(name, "blk.", 4) != 0) return false;
  char* end = nullptr;
  unsigned long l = std::strtoul(name + 4, &end, 10);
  if (end == name + 4 || *end != '.') return false;
  *layer = (uint32_t)l;
  *sub = end + 1;
  return true;
}

int lgh_upload_tensor(lgh_ctx* c, const char* name, uint32_t type, const uint64_t ne[4], const void* host, size_t nbytes) {
  if (!c || !name || !ne || !host) return LGH_INVALID_ARGUMENT;
  if (bind(c)) return fail(c, LGH_NOT_AVAILABLE, "hipSetDevice failed");
  if (c->finalized) return fail(c, LGH_INVALID_ARGUMENT, "upload after finalize");
  const lgh_model_desc& d = c->d;
  const std::string nm(name);
  const uint64_t n0 = ne[0], n1 = ne[1] ? ne[1] : 1, n2 = ne[2] ? ne[2] : 1;
  if (nm == "token_embd.weight") {
    if (n0 != d.hidden_size || n1 != d.vocab_size) return fail(c, LGH_SHAPE_MISMATCH, "token_embd.weight shape");
    const uint32_t be = blk_elems((int)type);
    if (!be || n0 % be || nbytes != n0 / be * blk_bytes((int)type) * n1) return fail(c, LGH_SHAPE_MISMATCH, "token_embd.weight bytes");
    if (c->first) {
      int rc = dev_alloc(c, (void**)&c->embd_raw, nbytes);
      if (rc) return rc;
      HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy(c->embd_raw, host, nbytes, hipMemcpyHostToDevice));
      c->stats.weight_bytes += nbytes;
    }
    c->embd_type = (int)type;
    c->embd_bytes = nbytes;
    if (c->last && !c->output.present()) c->embd_host.assign((const uint8_t*)host, (const uint8_t*)host + nbytes);  // tied output?
    return LGH_OK;
  }
  if (nm == "output_norm.weight") {
    if (!c->last) return LGH_OK;
    if (n0 != d.hidden_size) return fail(c, LGH_SHAPE_MISMATCH, "output_norm.weight shape");
    return upload_f32(c, &c->output_norm, (int)type, n0, host, nbytes);
  }
  if (nm == "output.weight") {
    if (!c->last) return LGH_OK;
    if (n0 != d.hidden_size || n1 != d.vocab_size) return fail(c, LGH_SHAPE_MISMATCH, "output.weight shape");
    c->embd_host.clear();
    c->embd_host.shrink_to_fit();
    return upload_matrix(c, c->output, (int)type, (uint32_t)n0, (uint32_t)n1, 1, -1, host, nbytes);
  }
  uint32_t li = 0;
  std::string sub;
  if (!parse_layer_name(name, &li, &sub)) return fail(c, LGH_INVALID_ARGUMENT, "unknown tensor name " + nm);
  if (li >= d.num_layers) return fail(c, LGH_INVALID_ARGUMENT, "layer index out of range: " + nm);
  LayerW& L = c->layers[li];
  if (!L.owned) return LGH_OK;  // another pipeline stage owns it
  const uint32_t H = d.hidden_size, QD = d.num_heads * d.head_dim, KD = d.num_kv_heads * d.head_dim;
  auto mat = [&](DevWeight& W, uint64_t k, uint64_t n) -> int {
    if (n0 != k || n1 != n) return fail(c, LGH_SHAPE_MISMATCH, nm + ": expected [" + std::to_string(k) + "," + std::to_string(n) + "]");
    return upload_matrix(c, W, (int)type, (uint32_t)k, (uint32_t)n, 1, -1, host, nbytes);
  };
  auto vec = [&](float** dst, uint64_t n) -> int {
    if (n0 * n1 != n) return fail(c, LGH_SHAPE_MISMATCH, nm + ": expected " + std::to_string(n) + " elements");
    return upload_f32(c, dst, (int)type, n, host, nbytes);
  };
  const uint32_t EI = d.expert_intermediate_size ? d.expert_intermediate_size : d.intermediate_size;
  if (sub == "attn_norm.weight") return vec(&L.attn_norm, H);
  if (sub == "ffn_norm.weight") return vec(&L.ffn_norm, H);
  if (sub == "attn_q.weight") return mat(L.wq, H, QD);
  if (sub == "attn_k.weight") return mat(L.wk, H, KD);
  if (sub == "attn_v.weight") return mat(L.wv, H, KD);
  if (sub == "attn_output.weight") return mat(L.wo, QD, H);
  if (sub == "attn_q.bias") return vec(&L.bq, QD);
  if (sub == "attn_k.bias") return vec(&L.bk, KD);
  if (sub == "attn_v.bias") return vec(&L.bv, KD);
  if (sub == "attn_output.bias") return vec(&L.bo, H);
  if (sub == "ffn_gate.weight") return mat(L.gate, H, d.intermediate_size);
  if (sub == "ffn_up.weight") return mat(L.up, H, d.intermediate_size);
  if (sub == "ffn_down.weight") return mat(L.down, d.intermediate_size, H);
  if (sub == "ffn_gate_inp.weight") {  // router [hidden, n_experts], must be F32 (moe.rs:132-135)
    if (type != LGH_TYPE_F32) return fail(c, LGH_DTYPE_MISMATCH, "router weight must be F32");
    return vec(&L.router, (uint64_t)H * d.num_experts);
  }
  // expert stacks: 3-D [in, out, n_expert], expert outermost (loader.rs:1256-1303)
  auto stack = [&](DevWeight& W, uint64_t k, uint64_t n) -> int {
    if (n0 != k || n1 != n || n2 != d.num_experts) return fail(c, LGH_SHAPE_MISMATCH, nm + ": bad expert stack shape");
    return upload_matrix(c, W, (int)type, (uint32_t)k, (uint32_t)n, d.num_experts, -1, host, nbytes);
  };
  if (sub == "ffn_gate_exps.weight") return stack(L.gate_exps, H, EI);
  if (sub == "ffn_up_exps.weight") return stack(L.up_exps, H, EI);
  if (sub == "ffn_down_exps.weight") return stack(L.down_exps, EI, H);
  // per-expert tensors as the loader renames them: ffn_gate.{e}.weight (loader.rs:1171-1173)
  auto one_expert = [&](const char* prefix, DevWeight& W, uint64_t k, uint64_t n) -> int {
    size_t pl = std::strlen(prefix);
    if (sub.compare(0, pl, prefix) != 0) return -1;
    char* end = nullptr;
    unsigned long e = std::strtoul(sub.c_str() + pl, &end, 10);
    if (end == sub.c_str() + pl || std::strcmp(end, ".weight") != 0) return -1;
    if (e >= d.num_experts) return fail(c, LGH_INVALID_ARGUMENT, nm + ": expert index out of range");
    if (n0 != k || n1 != n) return fail(c, LGH_SHAPE_MISMATCH, nm + ": bad expert shape");
    return upload_matrix(c, W, (int)type, (uint32_t)k, (uint32_t)n, d.num_experts, (int)e, host, nbytes);
  };
  int r;
  if ((r = one_expert("ffn_gate.", L.gate_exps, H, EI)) >= 0) return r;
  if ((r = one_expert("ffn_up.", L.up_exps, H, EI)) >= 0) return r;
  if ((r = one_expert("ffn_down.", L.down_exps, EI, H)) >= 0) return r;
  return fail(c, LGH_INVALID_ARGUMENT, "unknown tensor name " + nm);
}

int lgh_finalize(lgh_ctx* c) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (bind(c)) return fail(c, LGH_NOT_AVAILABLE, "hipSetDevice failed");
  if (c->finalized) return LGH_OK;
  const lgh_model_desc& d = c->d;
  int rc;
  if (c->first && !c->embd_raw) return fail(c, LGH_INITIALIZATION_FAILED, "missing token_embd.weight");
  if (c->last) {
    if (!c->output_norm) return fail(c, LGH_INITIALIZATION_FAILED, "missing output_norm.weight");
    if (!c->output.present()) {  // tied output projection (loader.rs:348-355)
      if (c->embd_host.empty()) return fail(c, LGH_INITIALIZATION_FAILED, "missing output.weight and token_embd.weight");
      if ((rc = upload_matrix(c, c->output, c->embd_type, d.hidden_size, d.vocab_size, 1, -1, c->embd_host.data(), c->embd_host.size()))) return rc;
    }
    c->embd_host.clear();
    c->embd_host.shrink_to_fit();
  }
  const size_t kv_elems = (size_t)d.num_kv_heads * d.max_seq_len * d.head_dim;
  for (uint32_t i = c->l0; i < c->l1; i++) {
    LayerW& L = c->layers[i];
    const std::string p = "blk." + std::to_string(i) + ".";
    if (!L.attn_norm || !L.ffn_norm) return fail(c, LGH_INITIALIZATION_FAILED, "missing norm weights of " + p);
    if (!L.wq.present() || !L.wk.present() || !L.wv.present() || !L.wo.present()) return fail(c, LGH_INITIALIZATION_FAILED, "missing attention weights of " + p);
    if (L.moe()) {
      if (!L.gate_exps.present() || !L.up_exps.present() || !L.down_exps.present()) return fail(c, LGH_INITIALIZATION_FAILED, "missing expert stacks of " + p);
      // experts uploaded one at a time (blk.N.ffn_{gate,up,down}.E.weight): every slot of every stack must have arrived,
      // otherwise the layer would decode from whatever the allocation held
      const std::pair<const char*, const DevWeight*> stacks[3] = {{"ffn_gate", &L.gate_exps}, {"ffn_up", &L.up_exps}, {"ffn_down", &L.down_exps}};
      for (const auto& sk : stacks) {
        std::string missing;
        for (uint32_t e = 0; e < sk.second->n_stack; e++)
          if (e >= sk.second->filled.size() || !sk.second->filled[e]) missing += (missing.empty() ? "" : ", ") + std::to_string(e);
        if (!missing.empty())
          return fail(c, LGH_INITIALIZATION_FAILED, "missing expert tensors " + p + sk.first + ".{" + missing + "}.weight");
      }
    } else if (!L.gate.present() || !L.up.present() || !L.down.present()) {
      return fail(c, LGH_INITIALIZATION_FAILED, "missing FFN weights of " + p);
    }
    if (d.flags & LGH_FLAG_KV_INT8) {
      // QuantizedKVCache::new with KVCacheFormat::Int8 (kv_quantized.rs:57-102): int8 rows + a scale per (kv head, position)
      const size_t n_rows = (size_t)d.num_kv_heads * d.max_seq_len;
      if (kv_is_tq(d.kv_cache_type)) {   // TurboQuantKVCache::new (kv_turboquant.rs:36-86): packed codes, no scales, no norms
        const size_t bytes = n_rows * tq_row_bytes_host(kv_tq_bits(d.kv_cache_type), d.head_dim);
        for (int8_t** p8 : {&L.k8, &L.v8}) {
          if ((rc = dev_alloc(c, (void**)p8, bytes))) return rc;
          HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(*p8, 0, bytes, c->stream));
        }
        c->stats.kv_bytes += 2 * bytes;
        if (kv_is_qjl(d.kv_cache_type)) {   // + per K row: sign bits of the projected residual and its norm (quant.rs:176-186)
          const size_t xb = n_rows * (d.head_dim / 32 + 1) * 4;
          if ((rc = dev_alloc(c, (void**)&L.kx, xb))) return rc;
          HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(L.kx, 0, xb, c->stream));
          c->stats.kv_bytes += xb;
        }
        continue;
      }
      for (int8_t** p8 : {&L.k8, &L.v8}) {
        if ((rc = dev_alloc(c, (void**)p8, kv_elems))) return rc;
        HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(*p8, 0, kv_elems, c->stream));
      }
      const bool scales = d.kv_cache_type == LGH_KV_INT8;   // (the FP8 formats have none, kv_quantized.rs:31-35)
      for (float** ps : {&L.kscale, &L.vscale}) {
        if (!scales) break;
        if ((rc = dev_alloc(c, (void**)ps, n_rows * 4))) return rc;
        HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(*ps, 0, n_rows * 4, c->stream));
      }
      c->stats.kv_bytes += 2 * (kv_elems + (scales ? n_rows * 4 : 0));
      continue;
    }
    // per-layer K/V [kv_heads, max_seq, head_dim] f32 (gpu_only.rs:555-572)
    if ((rc = dev_alloc(c, (void**)&L.kcache, kv_elems * 4))) return rc;
    if ((rc = dev_alloc(c, (void**)&L.vcache, kv_elems * 4))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(L.kcache, 0, kv_elems * 4, c->stream));
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(L.vcache, 0, kv_elems * 4, c->stream));
    c->stats.kv_bytes += 2 * kv_elems * 4;
  }
  const uint32_t EI = d.expert_intermediate_size ? d.expert_intermediate_size : d.intermediate_size;
  const size_t ffn = std::max<size_t>(d.intermediate_size, EI);
  const size_t G = d.num_heads / d.num_kv_heads;
  struct { void** p; size_t n; } bufs[] = {
      {(void**)&c->hidden, (size_t)d.hidden_size * 4},
      {(void**)&c->xnorm, (size_t)d.hidden_size * 4},
      {(void**)&c->q, (size_t)d.num_heads * d.head_dim * 4},
      {(void**)&c->kv_tmp, (size_t)2 * d.num_kv_heads * d.head_dim * 4},
      {(void**)&c->attn_out, (size_t)d.num_heads * d.head_dim * 4},
      {(void**)&c->act, ffn * 4},
      {(void**)&c->act2, ffn * 4},
      {(void**)&c->logits, (size_t)d.vocab_size * 4},
      {(void**)&c->part_ml, (size_t)d.num_kv_heads * c->n_splits * G * 2 * 4},
      {(void**)&c->part_acc, (size_t)d.num_kv_heads * c->n_splits * G * d.head_dim * 4},
      {(void**)&c->rope_cs, (size_t)d.max_seq_len * d.head_dim * 4},
      {(void**)&c->moe_w, 8 * 4},
      {(void**)&c->moe_sel, 8 * 4},
      {(void**)&c->state, ST_WORDS * 4},
      {(void**)&c->amax_v, 64 * 4},
      {(void**)&c->amax_i, 64 * 4},
      {(void**)&c->tok_log, (size_t)d.max_seq_len * 4},
  };
  for (auto& b : bufs) {
    if ((rc = dev_alloc(c, b.p, b.n))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetAsync(*b.p, 0, b.n, c->stream));
    c->stats.scratch_bytes += b.n;
  }
  if (kv_is_tq(d.kv_cache_type)) {
    // the rotations' sign vectors, [owned layer][kv head][k, v][head_dim]: given through lgh_set_kv_rotation_signs (the Rust host
    // passes every engine's HadamardRotation::signs()), otherwise a deterministic stand-in — NOT the reference's StdRng stream
    const size_t n = (size_t)(c->l1 - c->l0) * d.num_kv_heads * 2 * d.head_dim;
    if (c->tq_signs_host.empty()) {
      c->tq_signs_host.resize(n);
      for (size_t i = 0; i < n; i++) {
        const uint64_t engine = (uint64_t)c->l0 * d.num_kv_heads * 2 + i / d.head_dim;   // (layer * kv_heads + head) * 2 + {k, v}
        uint64_t z = (engine * 0x9E3779B97F4A7C15ull) ^ ((i % d.head_dim) * 0xBF58476D1CE4E5B9ull);
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        c->tq_signs_host[i] = (z & 1) ? 1.0f : -1.0f;
      }
    }
    if (c->tq_signs_host.size() != n) return fail(c, LGH_INVALID_ARGUMENT, "the rotation sign vector has the wrong length for this context");
    if ((rc = dev_alloc(c, (void**)&c->tq_signs, n * 4))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy(c->tq_signs, c->tq_signs_host.data(), n * 4, hipMemcpyHostToDevice));
    if (kv_is_qjl(d.kv_cache_type)) {
      // the K engines' projection matrices, [owned layer][kv head][head_dim][head_dim]: given through lgh_set_kv_qjl_matrices, otherwise
      // a deterministic stand-in (Box-Muller over a counter hash: i.i.d. N(0, 1), NOT the reference's StdRng / ziggurat stream)
      const size_t nm = (size_t)(c->l1 - c->l0) * d.num_kv_heads * d.head_dim * d.head_dim;
      if (c->tq_qjl_host.empty()) {
        c->tq_qjl_host.resize(nm);
        auto mix = [](uint64_t z) { z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31; return z; };
        const uint64_t base = (uint64_t)c->l0 * d.num_kv_heads * d.head_dim * d.head_dim;
        for (size_t i = 0; i < nm; i += 2) {
          const uint64_t a = mix((base + i) * 0x9E3779B97F4A7C15ull + 0x51ED270B7F4A7C15ull), b = mix(a + 0xD1B54A32D192ED03ull);
          const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740993.0), u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
          const double r = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586 * u2;
          c->tq_qjl_host[i] = (float)(r * std::cos(th));
          if (i + 1 < nm) c->tq_qjl_host[i + 1] = (float)(r * std::sin(th));
        }
      }
      if (c->tq_qjl_host.size() != nm) return fail(c, LGH_INVALID_ARGUMENT, "the QJL matrices have the wrong size for this context");
      if ((rc = dev_alloc(c, (void**)&c->tq_qjl, nm * 4))) return rc;
      HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy(c->tq_qjl, c->tq_qjl_host.data(), nm * 4, hipMemcpyHostToDevice));
    }
  }
  // XQ images of the vectors that feed quantized mat-vecs (allocated here, never during a graph capture)
  if (!xq_get(c, c->hidden, d.hidden_size) || !xq_get(c, c->attn_out, d.num_heads * d.head_dim) || !xq_get(c, c->act, (uint32_t)ffn) ||
      !xq_get(c, c->act2, (uint32_t)ffn))
    return fail(c, LGH_ALLOCATION_FAILED, "XQ image allocation failed");
  {
    // RoPE table in the reference's own arithmetic (ops.rs:1303-1313): libm powf / cosf / sinf on the host
    const uint32_t half = d.head_dim / 2;
    std::vector<float> cs((size_t)d.max_seq_len * half * 2);
    for (uint32_t p = 0; p < d.max_seq_len; p++) {
      const float position = (float)p / d.rope_freq_scale;
      for (uint32_t i = 0; i < half; i++) {
        const float freq = 1.0f / std::pow(d.rope_freq_base, (float)(2 * i) / (float)d.head_dim);
        const float theta = position * freq;
        cs[((size_t)p * half + i) * 2] = std::cos(theta);
        cs[((size_t)p * half + i) * 2 + 1] = std::sin(theta);
      }
    }
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy(c->rope_cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  c->finalized = true;
  c->pos = 0;
  return warm_kernels(c);
}

void lgh_destroy(lgh_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  for (int m = 0; m < MODE_COUNT; m++)
    for (int v = 0; v < 2; v++)
      if (c->graph[m][v]) (void)hipGraphExecDestroy(c->graph[m][v]);
  for (auto& r : c->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (void* p : c->allocs) (void)hipFree(p);
  for (auto& gg : c->batch.graph)
    for (auto& ge : gg)
      if (ge) (void)hipGraphExecDestroy(ge);
  if (c->batch.h_ctl) (void)hipHostFree(c->batch.h_ctl);
  if (c->pf.tok_pinned) (void)hipHostFree(c->pf.tok_pinned);
  if (c->pf.tok_copied) (void)hipEventDestroy(c->pf.tok_copied);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

static int check_ready(lgh_ctx* c) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (!c->finalized) return fail(c, LGH_INVALID_ARGUMENT, "context not finalized");
  if (bind(c)) return fail(c, LGH_NOT_AVAILABLE, "hipSetDevice failed");
  return LGH_OK;
}

int lgh_forward(lgh_ctx* c, uint32_t token, float* logits_out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!logits_out) return fail(c, LGH_INVALID_ARGUMENT, "logits_out is NULL");
  if (!c->first || !c->last) return fail(c, LGH_INVALID_ARGUMENT, "lgh_forward needs a single-stage context; use lgh_stage_forward");
  if ((rc = set_token(c, token))) return rc;
  if ((rc = step(c, MODE_FORWARD))) return rc;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(logits_out, c->logits, (size_t)c->d.vocab_size * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

int lgh_prefill_token(lgh_ctx* c, uint32_t token) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!c->first || !c->last) return fail(c, LGH_INVALID_ARGUMENT, "lgh_prefill_token needs a single-stage context");
  if ((rc = set_token(c, token))) return rc;
  return step(c, MODE_PREFILL);
}

int lgh_prefill_is_batched(lgh_ctx* c) { return c && c->finalized && pf_eligible(c) ? 1 : 0; }

int lgh_stage_hidden_block_buffer(lgh_ctx* c, void** p) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!p) return fail(c, LGH_INVALID_ARGUMENT, "p is NULL");
  if (!pf_eligible(c)) return fail(c, LGH_UNSUPPORTED, "this context has no batched prompt path");
  if ((rc = pf_ensure(c))) return rc;
  *p = c->pf.hidden;
  return LGH_OK;
}

int lgh_stage_prefill_batch(lgh_ctx* c, const uint32_t* tokens, size_t n) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!pf_eligible(c)) return fail(c, LGH_UNSUPPORTED, "this context has no batched prompt path");
  if (n == 0 || n > (size_t)kPfTokens) return fail(c, LGH_INVALID_ARGUMENT, "a stage block holds 1..128 tokens");
  if (c->first && !tokens) return fail(c, LGH_INVALID_ARGUMENT, "the first stage needs the token ids");
  if (c->pos + n > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "prompt block exceeds max_seq_len");
  if (c->first)
    for (size_t i = 0; i < n; i++)
      if (tokens[i] >= c->d.vocab_size) return fail(c, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");
  return prefill_block(c, tokens, (uint32_t)n);
}

int lgh_prefill_batch(lgh_ctx* c, const uint32_t* tokens, size_t n) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (n && !tokens) return fail(c, LGH_INVALID_ARGUMENT, "tokens is NULL");
  if (n >= 2 && c->first && c->last && pf_eligible(c)) {
    if (c->pos + n > c->d.max_seq_len)
      return fail(c, LGH_INVALID_ARGUMENT, "prompt of " + std::to_string(n) + " tokens at position " + std::to_string(c->pos) + " exceeds max_seq_len");
    for (size_t i = 0; i < n; i++)
      if (tokens[i] >= c->d.vocab_size) return fail(c, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");
    for (size_t i = 0; i < n; i += kPfTokens)
      if ((rc = prefill_block(c, tokens + i, (uint32_t)std::min<size_t>(kPfTokens, n - i)))) return rc;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
    return LGH_OK;
  }
  for (size_t i = 0; i < n; i++)
    if ((rc = lgh_prefill_token(c, tokens[i]))) return rc;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

// The prompt of one slot of the multi-sequence engine (engine_batch.hip): the batched prompt path with the slot's caches and
// position in the context's place (same kernels, so the slot's K / V rows are the ones lgh_prefill_batch would write).
int lgh_batch_prefill(lgh_ctx* c, uint32_t slot, const uint32_t* tokens, size_t n) {
  int rc = check_ready(c);
  if (rc) return rc;
  BatchScratch& Bs = c->batch;
  if (!Bs.ready || slot >= Bs.max_batch) return fail(c, LGH_INVALID_ARGUMENT, "no such slot (lgh_batch_create first)");
  if (n && !tokens) return fail(c, LGH_INVALID_ARGUMENT, "tokens is NULL");
  if (Bs.pos[slot] + n > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "prompt exceeds max_seq_len");
  for (size_t i = 0; i < n; i++)
    if (tokens[i] >= c->d.vocab_size) return fail(c, LGH_INVALID_ARGUMENT, "token id exceeds vocab size");
  if (n == 0) return LGH_OK;
  if (!pf_eligible(c)) {   // no batched prompt path for this model: token by token through the multi-sequence step
    for (size_t i = 0; i < n; i++)
      if ((rc = lgh_forward_multi(c, &slot, tokens + i, 1, nullptr, nullptr))) return rc;
    return LGH_OK;
  }
  const size_t keep = c->pos;
  std::vector<std::pair<float*, float*>> saved;
  for (uint32_t i = c->l0; i < c->l1; i++) {
    saved.emplace_back(c->layers[i].kcache, c->layers[i].vcache);
    c->layers[i].kcache = Bs.kcache[i] + (size_t)slot * Bs.cache_stride;
    c->layers[i].vcache = Bs.vcache[i] + (size_t)slot * Bs.cache_stride;
  }
  c->pos = Bs.pos[slot];
  for (size_t i = 0; i < n && !rc; i += kPfTokens) rc = prefill_block(c, tokens + i, (uint32_t)std::min<size_t>(kPfTokens, n - i));
  if (!rc) Bs.pos[slot] = c->pos;
  c->pos = keep;
  for (uint32_t i = c->l0; i < c->l1; i++) { c->layers[i].kcache = saved[i - c->l0].first; c->layers[i].vcache = saved[i - c->l0].second; }
  if (hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_NEXT), (int)keep, 1, c->stream) != hipSuccess && !rc) rc = fail(c, LGH_OPERATION_FAILED, "state reset");
  if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = fail(c, LGH_OPERATION_FAILED, "hipStreamSynchronize");
  return rc;
}

void lgh_reset(lgh_ctx* c) {
  if (!c || !c->finalized) return;
  if (bind(c)) return;
  (void)hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_NEXT), 0, 1, c->stream);
  c->pos = 0;
}

size_t lgh_position(const lgh_ctx* c) { return c ? c->pos : 0; }

int lgh_kv_truncate(lgh_ctx* c, size_t new_len) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (new_len < c->pos) {   // KVCache::truncate (model/mod.rs:130-134): only ever shortens
    c->pos = new_len;
    HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_NEXT), (int)c->pos, 1, c->stream));
  }
  return LGH_OK;
}

int lgh_kv_shift_left(lgh_ctx* c, size_t amount) {
  int rc = check_ready(c);
  if (rc) return rc;
  const lgh_model_desc& d = c->d;
  if (amount == 0 || amount >= c->pos) {   // model/mod.rs:143-146 — a shift by 0 clears the cache too
    c->pos = 0;
  } else {
    // rows [amount, pos) of every kv head move to [0, pos - amount) (model/mod.rs:148-169).  The ranges overlap, so each
    // tensor goes through a scratch buffer: two strided device-to-device copies instead of the host's memmove.
    const size_t new_len = c->pos - amount, row = (size_t)d.head_dim * 4;
    if (!c->kv_shift_tmp && (rc = dev_alloc(c, (void**)&c->kv_shift_tmp, (size_t)d.num_kv_heads * d.max_seq_len * row))) return rc;
    // (base, bytes per position): the f32 caches, or the int8 rows and their scales (QuantizedKVCache::shift_left, kv_quantized.rs:330-372)
    auto shift = [&](void* base, size_t rb) -> int {
      uint8_t* b = (uint8_t*)base;
      HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy2DAsync(c->kv_shift_tmp, new_len * rb, b + amount * rb, (size_t)d.max_seq_len * rb, new_len * rb,
                                                        d.num_kv_heads, hipMemcpyDeviceToDevice, c->stream));
      HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpy2DAsync(b, (size_t)d.max_seq_len * rb, c->kv_shift_tmp, new_len * rb, new_len * rb, d.num_kv_heads,
                                                        hipMemcpyDeviceToDevice, c->stream));
      return LGH_OK;
    };
    for (uint32_t li = c->l0; li < c->l1; li++) {
      LayerW& L = c->layers[li];
      if (kv_is_tq(d.kv_cache_type)) {   // TurboQuantKVCache::shift_left (kv_turboquant.rs:245-266): code rows
        const size_t rb = tq_row_bytes_host(kv_tq_bits(d.kv_cache_type), d.head_dim);
        if ((rc = shift(L.k8, rb)) || (rc = shift(L.v8, rb))) return rc;
        if (kv_is_qjl(d.kv_cache_type) && (rc = shift(L.kx, (size_t)(d.head_dim / 32 + 1) * 4))) return rc;
      } else if (d.flags & LGH_FLAG_KV_INT8) {
        if ((rc = shift(L.k8, d.head_dim)) || (rc = shift(L.v8, d.head_dim))) return rc;
        if (d.kv_cache_type == LGH_KV_INT8 && ((rc = shift(L.kscale, 4)) || (rc = shift(L.vscale, 4)))) return rc;
      } else if ((rc = shift(L.kcache, row)) || (rc = shift(L.vcache, row))) {
        return rc;
      }
    }
    c->pos = new_len;
  }
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemsetD32Async((hipDeviceptr_t)(c->state + ST_NEXT), (int)c->pos, 1, c->stream));
  return LGH_OK;
}

int lgh_forward_argmax(lgh_ctx* c, uint32_t token, uint32_t* next_token) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!next_token) return fail(c, LGH_INVALID_ARGUMENT, "next_token is NULL");
  if (!c->first || !c->last) return fail(c, LGH_INVALID_ARGUMENT, "needs a single-stage context");
  if ((rc = set_token(c, token))) return rc;
  if ((rc = step(c, MODE_GREEDY))) return rc;
  int tok = 0;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(&tok, c->state + ST_ARGMAX, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  *next_token = (uint32_t)tok;
  return LGH_OK;
}

int lgh_decode_greedy(lgh_ctx* c, uint32_t first_token, size_t n_steps, uint32_t* tokens_out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (n_steps && !tokens_out) return fail(c, LGH_INVALID_ARGUMENT, "tokens_out is NULL");
  if (!c->first || !c->last) return fail(c, LGH_INVALID_ARGUMENT, "needs a single-stage context");
  if (c->pos + n_steps > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "decode would exceed max_seq_len");
  if ((rc = set_token(c, first_token))) return rc;
  // argmax_stage2 writes each token into state[TOKEN] (the feedback) and into tok_log[position]
  const size_t pos0 = c->pos;
  for (size_t i = 0; i < n_steps; i++)
    if ((rc = step(c, MODE_GREEDY))) return rc;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(tokens_out, c->tok_log + pos0, n_steps * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

const char* lgh_last_error(const lgh_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lgh_get_stats(lgh_ctx* c, lgh_stats* out) {
  if (!c || !out) return LGH_INVALID_ARGUMENT;
  c->stats.graph_nodes = c->graph_nodes;
  // algorithmic bytes of one decode step at the current position
  uint64_t b = 0;
  const lgh_model_desc& d = c->d;
  if (c->finalized) {
    if (c->first) b += (uint64_t)d.hidden_size * blk_bytes(c->embd_type) / blk_elems(c->embd_type);
    for (uint32_t i = c->l0; i < c->l1; i++) {
      const LayerW& L = c->layers[i];
      b += L.wq.bytes + L.wk.bytes + L.wv.bytes + L.wo.bytes;
      if (L.moe()) b += (uint64_t)d.num_experts_per_token * (L.gate_exps.bytes + L.up_exps.bytes + L.down_exps.bytes) + (uint64_t)d.num_experts * d.hidden_size * 4;
      else b += L.gate.bytes + L.up.bytes + L.down.bytes;
      b += (uint64_t)2 * d.hidden_size * 4;                                       // norm weights
      const uint64_t kv_row = kv_is_tq(d.kv_cache_type) ? tq_row_bytes_host(kv_tq_bits(d.kv_cache_type), d.head_dim)
                              : d.kv_cache_type == LGH_KV_INT8 ? d.head_dim + 4                                // int8 row + its scale
                              : d.kv_cache_type != LGH_KV_F32 ? d.head_dim : (uint64_t)d.head_dim * 4;
      b += (uint64_t)2 * d.num_kv_heads * (c->pos + 1) * kv_row;                  // KV read (kv_len = pos+1)
      b += (uint64_t)2 * d.num_kv_heads * kv_row;                                 // KV write
    }
    if (c->last) b += c->output.bytes + (uint64_t)d.hidden_size * 4 + (uint64_t)d.vocab_size * 4;
  }
  c->stats.step_alg_bytes = b;
  c->stats.overlapped_edges = 0;
  *out = c->stats;
  return LGH_OK;
}

int lgh_set_profiling(lgh_ctx* c, int on) {
  if (!c) return LGH_INVALID_ARGUMENT;
  c->profiling = on != 0;
  if (on) {
    std::memset(c->stats.k_launches, 0, sizeof(c->stats.k_launches));
    std::memset(c->stats.k_time_us, 0, sizeof(c->stats.k_time_us));
    std::memset(c->stats.k_alg_bytes, 0, sizeof(c->stats.k_alg_bytes));
    std::memset(c->stats.sym_launches, 0, sizeof(c->stats.sym_launches));
    std::memset(c->stats.sym_time_us, 0, sizeof(c->stats.sym_time_us));
    std::memset(c->stats.sym_alg_bytes, 0, sizeof(c->stats.sym_alg_bytes));
    c->stats.event_bracket_us = 0.0;
    c->stats.event_bracket_samples = 0;
  }
  return LGH_OK;
}

int lgh_set_stream(lgh_ctx* c, void* s) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (bind(c)) return LGH_NOT_AVAILABLE;
  (void)hipStreamSynchronize(c->stream);
  hipStream_t ns = s ? (hipStream_t)s : c->own_stream;
  if (ns != c->stream) drop_graphs(c);
  c->stream = ns;
  return LGH_OK;
}

int lgh_stage_set_forward_targets(lgh_ctx* c, void* hidden_dst, void* token_dst) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (hidden_dst && c->last) return fail(c, LGH_INVALID_ARGUMENT, "the last stage hands no hidden vector on");
  if (token_dst && !c->last) return fail(c, LGH_INVALID_ARGUMENT, "only the last stage produces the arg-max token");
  if (hidden_dst == c->hidden) return fail(c, LGH_INVALID_ARGUMENT, "hidden_dst is this stage's own buffer");
  (void)hipStreamSynchronize(c->stream);
  if (hidden_dst != c->fwd_hidden || token_dst != c->fwd_token) drop_graphs(c);
  c->fwd_hidden = hidden_dst;
  c->fwd_token = token_dst;
  // first launch outside any capture (see warm_kernels); what it copies is overwritten before anything reads it
  if (hidden_dst) HIP_TRY(c, LGH_OPERATION_FAILED, copy_words_launch(hidden_dst, c->hidden, c->d.hidden_size, c->stream));
  if (token_dst) HIP_TRY(c, LGH_OPERATION_FAILED, copy_words_launch(token_dst, c->state + ST_ARGMAX, 1, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

void* lgh_get_stream(lgh_ctx* c) { return c ? (void*)c->stream : nullptr; }

int lgh_synchronize(lgh_ctx* c) {
  if (!c) return LGH_INVALID_ARGUMENT;
  if (bind(c)) return LGH_NOT_AVAILABLE;
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

int lgh_read_hidden(lgh_ctx* c, float* out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!out) return fail(c, LGH_INVALID_ARGUMENT, "out is NULL");
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(out, c->hidden, (size_t)c->d.hidden_size * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

int lgh_stage_hidden_buffer(lgh_ctx* c, void** p) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!p) return LGH_INVALID_ARGUMENT;
  *p = c->hidden;
  return LGH_OK;
}

int lgh_stage_forward(lgh_ctx* c, uint32_t token, int want_logits, float* logits_out, uint32_t* next_token) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->first && (rc = set_token(c, token))) return rc;
  int mode = MODE_PREFILL;
  if (c->last && want_logits) mode = next_token ? MODE_GREEDY : MODE_FORWARD;
  if ((rc = step(c, mode))) return rc;
  if (c->last && want_logits) {
    int tok = 0;
    if (logits_out) HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(logits_out, c->logits, (size_t)c->d.vocab_size * 4, hipMemcpyDeviceToHost, c->stream));
    if (next_token) HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(&tok, c->state + ST_ARGMAX, 4, hipMemcpyDeviceToHost, c->stream));
    if (logits_out || next_token) HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
    if (next_token) *next_token = (uint32_t)tok;
  }
  return LGH_OK;
}

// ---- device-side token feedback for multi-process pipelines: no host value crosses a stage boundary per token ----
int lgh_stage_io_buffers(lgh_ctx* c, void** token_in, void** argmax_out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (token_in) *token_in = c->state + ST_TOKEN;
  if (argmax_out) *argmax_out = c->state + ST_ARGMAX;
  return LGH_OK;
}

int lgh_stage_step(lgh_ctx* c, int mode) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (mode < 0 || mode >= MODE_COUNT) return fail(c, LGH_INVALID_ARGUMENT, "mode must be 0 (layers only), 1 (logits) or 2 (arg-max)");
  if (!c->last) mode = MODE_PREFILL;
  return step(c, mode);
}

int lgh_stage_read_logits(lgh_ctx* c, float* logits_out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!logits_out) return fail(c, LGH_INVALID_ARGUMENT, "logits_out is NULL");
  if (!c->last) return fail(c, LGH_INVALID_ARGUMENT, "only the last stage holds logits");
  HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(logits_out, c->logits, (size_t)c->d.vocab_size * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

int lgh_stage_read_tokens(lgh_ctx* c, size_t pos0, size_t n, uint32_t* out) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!out && n) return fail(c, LGH_INVALID_ARGUMENT, "out is NULL");
  if (!c->last) return fail(c, LGH_INVALID_ARGUMENT, "only the last stage logs the arg-max tokens");
  if (pos0 + n > c->d.max_seq_len) return fail(c, LGH_INVALID_ARGUMENT, "token log range exceeds max_seq_len");
  if (n) HIP_TRY(c, LGH_OPERATION_FAILED, hipMemcpyAsync(out, c->tok_log + pos0, n * 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, LGH_OPERATION_FAILED, hipStreamSynchronize(c->stream));
  return LGH_OK;
}

}  // extern "C"
