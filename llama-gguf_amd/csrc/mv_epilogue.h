// mv_epilogue.h — per-row epilogues shared by the VALU and the int8-MFMA mat-vec kernels (internal).
#pragma once

#include "device_utils.h"
#include "xq.h"

namespace lgh {

__device__ __forceinline__ float silu_f(float g) { return g / (1.0f + expf(-g)); }

// Operands of the epilogue that do not depend on the mat-vec — the residual element, or RoPE's cos/sin at the current
// position — can be fetched at kernel START and carried in two registers: loaded after the final barrier they are a
// dependent chain of global loads (position -> table index -> cos/sin) of ~1.5 us on the critical path.
struct MvEpiPre { float a, b; bool valid; };

// The residual element: issued first thing (one load, older than everything else in the wave's queue).  `epi` and
// `resid` come from the caller's first batch of scalar loads — fetched here they would be two more dependent round trips.
__device__ __forceinline__ void mv_epilogue_prefetch_resid(int epi, const float* resid, const float* xq_nw, uint32_t n_rows, uint32_t wg,
                                                           uint32_t rows_per_wg, MvEpiPre& pre, uint32_t tid = threadIdx.x) {
  if (resid && (epi == EPI_RESID || epi == EPI_MOE_DOWN)) {
    const uint32_t t = tid, row = wg * rows_per_wg + t;
    if (t < rows_per_wg && row < n_rows) {
      pre.a = resid[row];
      pre.b = xq_nw ? xq_nw[row] : 1.0f;   // the next consumer's norm weight, for the XQ image of the output
      pre.valid = true;
    }
  }
}
// RoPE's cos/sin at the current position (`pos` already in a scalar register).  Like the residual it must be issued
// BEFORE the input vector and the weight tiles: a conditional load issued after them makes the compiler's wait for the
// input vector conservative (it then also waits for the first tile to arrive from HBM).
__device__ __forceinline__ void mv_epilogue_prefetch_rope(int epi, uint32_t pos, const float* rope_cs, uint32_t head_dim,
                                                          uint32_t n_rows, uint32_t wg, uint32_t rows_per_wg, MvEpiPre& pre,
                                                          uint32_t tid = threadIdx.x) {
  if (epi == EPI_ROPE_Q || epi == EPI_ROPE_K) {
    const uint32_t rl = 2 * tid, row = wg * rows_per_wg + rl;
    if (rl < rows_per_wg && row < n_rows) {
      const uint32_t half = head_dim / 2, i = (row % head_dim) / 2;
      pre.a = rope_cs[((size_t)pos * half + i) * 2];
      pre.b = rope_cs[((size_t)pos * half + i) * 2 + 1];
      pre.valid = true;
    }
  }
}

// The vectors an epilogue writes and reads: the segment's own (single-sequence launches) or one sequence's slice of them
// (matvec_batch.hip: every sequence has its own output, residual, XQ image, position and KV cache slot).
struct MvEpiView { float* out; const float* resid; uint8_t* xq_out; float* xq_ssq; const int* pos; };

// Per-row epilogue, one thread per row (or per row pair for RoPE)
// `nslots` = partial sums per (pass, row) in `red`, laid out red[(p * nslots + slot) * rows_per_wg + row]
__device__ __forceinline__ void mv_epilogue_view(const MvLaunch& L, const MvSeg& S, const MvEpiView V, uint32_t wg, const float* red,
                                                 const float* ssq, uint32_t nslots, const MvEpiPre pre = MvEpiPre{0.0f, 0.0f, false},
                                                 uint32_t tid = threadIdx.x) {
  const uint32_t t = tid;
  const uint32_t rbase = wg * S.rows_per_wg;
  float inv = 1.0f;
  if (L.do_norm) {  // simd.rs:853-855: rms = sqrt(ss/n + eps); inv = 1/rms
    float tot = 0.0f;
    for (uint32_t w = 0; w < (blockDim.x >> 6); w++) tot += ssq[w];
    inv = 1.0f / __builtin_sqrtf(tot / (float)L.k + L.eps);
  }
  auto rowval = [&](int p, uint32_t rl) {
    float v = 0.0f;
    for (uint32_t ks = 0; ks < nslots; ks++) v += red[(size_t)(p * nslots + ks) * S.rows_per_wg + rl];
    return v * inv;
  };
  if (S.epi == EPI_ROPE_Q || S.epi == EPI_ROPE_K) {
    uint32_t rl = 2 * t, row = rbase + rl;
    if (rl >= S.rows_per_wg || row >= S.n_rows) return;
    float x0 = rowval(0, rl), x1 = rowval(0, rl + 1);
    if (S.bias) { x0 += S.bias[row]; x1 += S.bias[row + 1]; }
    const uint32_t pos = (uint32_t)*V.pos, d = S.head_dim, half = d / 2;
    const uint32_t head = row / d, i = (row % d) / 2;
    const float c = pre.valid ? pre.a : L.rope_cs[((size_t)pos * half + i) * 2];
    const float s = pre.valid ? pre.b : L.rope_cs[((size_t)pos * half + i) * 2 + 1];
    float y0 = x0 * c - x1 * s, y1 = x0 * s + x1 * c;  // ops.rs:1326-1331
    if (S.epi == EPI_ROPE_Q) {
      V.out[row] = y0;
      V.out[row + 1] = y1;
    } else {
      float* dst = V.out + ((size_t)head * S.max_seq + pos) * d + (row % d);
      dst[0] = y0;
      dst[1] = y1;
    }
    return;
  }
  if (t >= S.rows_per_wg) return;
  const uint32_t row = rbase + t;
  if (row >= S.n_rows) return;
  float v0 = rowval(0, t);
  if (S.bias) v0 += S.bias[row];
  float outv = 0.0f;        // the value written to out[row] by the epilogues that can also leave an XQ image
  bool has_out = false;
  switch (S.epi) {
    case EPI_STORE: outv = v0; V.out[row] = outv; has_out = true; break;
    case EPI_RESID: outv = v0 + (pre.valid ? pre.a : V.resid[row]); V.out[row] = outv; has_out = true; break;
    case EPI_SWIGLU: {
      float up = rowval(1, t);
      outv = silu_f(v0) * up;
      V.out[row] = outv;
      has_out = true;
      break;
    }
    case EPI_V_CACHE: {
      const uint32_t pos = (uint32_t)*V.pos, d = S.head_dim;
      V.out[((size_t)(row / d) * S.max_seq + pos) * d + (row % d)] = v0;
      break;
    }
    case EPI_MOE_SWIGLU: {
      for (int e = 0; 2 * e + 1 < S.npass; e++) {
        float g = rowval(2 * e, t), up = rowval(2 * e + 1, t);
        float* o = e == 0 ? V.out : S.out2;
        const float a = silu_f(g) * up;
        o[row] = a;
        uint8_t* xo = e == 0 ? V.xq_out : S.xq_out2;   // each expert's activation feeds its own down projection
        if (xo) xq_store_chunk(xo, row >> 4, a, nullptr, 0.0f, tid);
      }
      break;
    }
    case EPI_MOE_DOWN: {
      // moe.rs:363-368: ONE zero-initialised sum, += weight * expert_out over all selected experts in selection order, then the
      // residual.  The experts run two per launch: a launch continues the running sum the earlier ones left (S.out2; none: it
      // starts from zero) and only the last one adds the residual (resid; none: the output is the running sum).
      float acc = S.out2 ? S.out2[row] : 0.0f;
      for (int p = 0; p < S.npass; p++) acc += S.moe_w[p] * rowval(p, t);
      outv = V.resid ? acc + (pre.valid ? pre.a : V.resid[row]) : acc;
      V.out[row] = outv;
      has_out = true;
      break;
    }
    default: break;
  }
  // ---- XQ image of the output for an int8-MFMA consumer (xq.h).  n_rows is a multiple of 16 here (host-checked), so the
  // 16 threads of a chunk are all live; with a norm in front of the consumer the record holds out * norm_weight and each
  // chunk leaves its sum of out^2.
  uint8_t* xq = V.xq_out;
  if (xq && has_out) {
    const float* nw = S.xq_nw;
    const float w = !nw ? 1.0f : (pre.valid && (S.epi == EPI_RESID || S.epi == EPI_MOE_DOWN)) ? pre.b : nw[row];
    xq_store_chunk(xq, row >> 4, outv * w, V.xq_ssq, outv, tid);
  }
}

__device__ __forceinline__ void mv_epilogue(const MvLaunch& L, const MvSeg& S, uint32_t wg, const float* red, const float* ssq,
                                            uint32_t nslots, const MvEpiPre pre = MvEpiPre{0.0f, 0.0f, false}, uint32_t tid = threadIdx.x) {
  mv_epilogue_view(L, S, MvEpiView{S.out, S.resid, S.xq_out, S.xq_ssq, L.pos}, wg, red, ssq, nslots, pre, tid);
}

}  // namespace lgh
