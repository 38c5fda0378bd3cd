// matvec_batch.hip — the int8 matrix-core mat-vec for SEVERAL sequences at once (multi-sequence decode, engine.hip:
// lgh_forward_multi; the reference's BatchedEngine steps its active sequences one after the other, src/engine_batched.rs:
// 236-290, 355-400 — every sequence with its own InferenceContext / KV cache).  One launch reads every weight tile ONCE and
// multiplies it with the XQ records of all n_seq input vectors.
//
// Arithmetic.  A sequence's result is bit-identical to the single-sequence kernel (matvec_mfma.hip) on the same input: the
// same k-slices (T, blocks per slice), per slice the blocks in ascending order through the same fused operations
// (mvq_core.h: mvq_mac_tile(unpack, load) == mvq_consume_tile), the same two shuffles and the same sum over the slices in the
// epilogue.  Only WHO computes differs: how many tiles a workgroup owns (the partial sums of n_seq sequences must fit LDS).
//
// Structure.  One workgroup of 8 waves = T k-slices x G row groups as in the single-sequence kernel; a wave walks its
// (pass, tile) pairs two at a time and, for each pair of pairs, its blocks in order — a "step" = up to two weight tiles of
// one block.  The next step's tiles are requested at the start of a step (registers, non-temporal loads); a step unpacks its
// tiles once (B operands + scales, mvq_unpack_tile) and then loops over the sequences: the sequence's XQ operands of that
// block come straight from memory (L2: they were written by the previous launch) into registers, one sequence ahead, and are
// used for both tiles.  No LDS on the way in; LDS holds the partial sums [sequence][pass][slice][row] for the epilogue.
// Epilogues: store, +residual, SwiGLU pair, RoPE (every sequence at its own position) and the K / V rows into the sequence's
// own cache slot; optional XQ image of the output for the next launch.  MoE experts are not batched (every sequence selects
// its own experts): engine.hip runs those layers' FFN sequence by sequence through the single-sequence kernel.
#include <algorithm>

#include "device_utils.h"
#include "xq.h"
#include "mvq_core.h"
#include "mv_epilogue.h"

namespace lgh {

constexpr int kBWaves = 8;

template <uint32_t MASK, int NB>
__global__ void __launch_bounds__(kBWaves * 64) mvqb_kernel(uint32_t wbpack, uint32_t geom, uint32_t geom2, uint32_t red_floats, const MvLaunch L,
                                                            const MvBatch B) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const uint32_t bid = blockIdx.x;
  const int sg = (int)(bid >= (wbpack & 0xFFFFu)) + (int)(bid >= (wbpack >> 16));
  const MvSeg& S = L.seg[sg];
  constexpr bool kSingle = (MASK & (MASK - 1)) == 0;
  const int fmt = kSingle ? __builtin_ctz(MASK) : fmt_of_dev_type(S.type);
  auto is = [&](int f) { return ((MASK >> f) & 1u) != 0 && (kSingle || fmt == f); };
  const uint32_t tb = is(F_Q4K) ? fmt_tile_bytes(F_Q4K) : is(F_Q6K) ? fmt_tile_bytes(F_Q6K) : is(F_Q5K) ? fmt_tile_bytes(F_Q5K)
                      : is(F_Q80) ? fmt_tile_bytes(F_Q80) : fmt_tile_bytes(F_Q40);
  const uint32_t S_T = geom & 0xFFu, S_G = (geom >> 8) & 0xFFu, nbw = (geom >> 16) & 0x3FFFu;
  const bool nrm = (geom >> 31) != 0;
  const uint32_t S_nblk = geom2 & 0xFFFFu;
  const uint32_t Rg = L.nseg > 1 ? S.rows_per_wg >> (4 + __builtin_ctz(S_G)) : geom2 >> 16;
  const uint32_t S_rpw = 16u * Rg * S_G;
  const uint32_t wg = bid - S.wg_begin;
  const uint32_t n_seq = B.n_seq;
  const int npass = S.npass;   // 1 or 2 (gate | up); both passes share the input vector

  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t ks = wave, rg = 0;
  while (ks >= S_T) { ks -= S_T; rg++; }
  const bool active = rg < S_G;
  const uint32_t blk0 = ks * nbw;
  const uint32_t nblk_w = active && blk0 < S_nblk ? min(nbw, S_nblk - blk0) : 0;
  const uint32_t ntiles = (S.n_rows + 15) >> 4;
  const uint32_t tile0 = (wg * S_G + rg) * Rg;
  const uint32_t ntile_w = active && tile0 < ntiles ? min(Rg, ntiles - tile0) : 0;
  float* red = reinterpret_cast<float*>(smem8);           // [n_seq][red_floats]
  float* ssq = red + (size_t)n_seq * red_floats;          // [n_seq][8]
  const uint32_t n = lane & 15, mq = lane >> 4;

  const uint64_t woff = ((uint64_t)tile0 * S_nblk + blk0) * tb;
  const uint8_t* pb0 = S.pass[0].plane[0] + woff;
  const uint8_t* pb1 = npass > 1 ? S.pass[1].plane[0] + woff : pb0;
  const uint8_t* xg = S.pass[0].xq + (size_t)blk0 * kXqRecord;   // sequence 0's records of this wave's k-slice

  // a slice beyond the last block, or a row group beyond the last tile, still owns partial-sum slots: they must read as zero
  const uint32_t npairs = (uint32_t)npass * ntile_w;
  if (nblk_w == 0 && ntile_w > 0) {
    for (uint32_t s = 0; s < n_seq; s++)
      for (uint32_t pr = 0; pr < npairs; pr++) {
        const uint32_t p = pr / ntile_w, tl = pr - p * ntile_w;
        if (mq == 0) red[(size_t)s * red_floats + (size_t)(p * S_T + ks) * S_rpw + (rg * Rg + tl) * 16 + n] = 0.0f;
      }
  }
  const uint32_t ngroups = (npairs + 1) / 2;
  const uint32_t nsteps = nblk_w ? ngroups * nblk_w : 0;

  // ---- the weight stream: step (g, b) = tiles (pair 2g, block b) and (pair 2g + 1, block b); a missing second pair re-reads
  // the first one's tile (an L1 / L2 hit: the issue pattern stays the same for every step, which keeps the waits exact)
  auto tile_ptr = [&](uint32_t pr, uint32_t b) -> const uint8_t* {
    const uint32_t p = pr / ntile_w, tl = pr - p * ntile_w;
    return (p == 0 ? pb0 : pb1) + ((size_t)tl * S_nblk + b) * tb;
  };
  auto issue_step = [&](uint32_t g, uint32_t b, RawT16 (&dst)[2]) {
    const uint32_t p0 = 2 * g, p1 = 2 * g + 1 < npairs ? 2 * g + 1 : 2 * g;
    mvq_issue_tile<MASK>(fmt, tile_ptr(p0, b), lane, dst[0]);
    mvq_issue_tile<MASK>(fmt, tile_ptr(p1, b), lane, dst[1]);
  };
  float acc[NB][2];
  // The sequences' XQ operands come from L2 (~0.7 us away): a ring of kXd operand sets in registers; sequence s + kXd of the step
  // is requested when sequence s has been used, and the first kXd sequences of the NEXT step at the end of a step — i.e. before
  // that step requests its weight prefetch (a wave's loads return in order: a load issued behind the prefetch waits for HBM).
  constexpr int kXd = NB >= 8 && !(NB == 16 && !kSingle) ? 3 : 2;   // (the mixed-format instantiations have no registers to spare at 16 sequences)
  XqOps xr[kXd];
  auto x_issue = [&](uint32_t b, int s, XqOps& o) { xq_load_ops(xg + (size_t)s * B.xq_stride + (size_t)b * kXqRecord, lane, o); };
  auto x_prime = [&](uint32_t b) {
#pragma unroll
    for (int s = 0; s < kXd; s++)
      if ((uint32_t)s < n_seq) x_issue(b, s, xr[s]);
  };
  auto process = [&](uint32_t g, uint32_t b, const RawT16 (&w)[2], bool more, uint32_t b_next) {
    const bool two = 2 * g + 1 < npairs;
    if (b == 0) {
#pragma unroll
      for (int s = 0; s < NB; s++) acc[s][0] = acc[s][1] = 0.0f;
    }
    TileOps t0, t1;
    mvq_unpack_tile<MASK>(fmt, w[0], lane, t0);
    mvq_unpack_tile<MASK>(fmt, w[1], lane, t1);
#pragma unroll
    for (int s = 0; s < NB; s++) {
      if ((uint32_t)s < n_seq) {
#ifdef LGH_MVQB_NOMAC   // experiment: memory side only
        acc[s][0] += xr[s % kXd].xs[0] + __builtin_bit_cast(float, xr[s % kXd].a[3][3]) + t0.dd;
        acc[s][1] += xr[s % kXd].sx[1] + t1.dd;
#else
        mvq_mac_tile<MASK>(fmt, t0, xr[s % kXd], acc[s][0]);
        if (two) mvq_mac_tile<MASK>(fmt, t1, xr[s % kXd], acc[s][1]);
#endif
#ifndef LGH_MVQB_NOX    // experiment: arithmetic side only
        if ((uint32_t)(s + kXd) < n_seq) x_issue(b, s + kXd, xr[s % kXd]);
#endif
      }
    }
#ifndef LGH_MVQB_NOX
    if (more) x_prime(b_next);   // the next step's first sequences, ahead of its weight prefetch
#endif
    if (b + 1 == nblk_w) {   // last block of the group's pairs: the four lane groups -> one partial sum per row and sequence
#pragma unroll
      for (int s = 0; s < NB; s++) {
        if ((uint32_t)s < n_seq) {
#pragma unroll
          for (int j = 0; j < 2; j++) {
            if (j == 0 || two) {
              const uint32_t pr = 2 * g + j, p = pr / ntile_w, tl = pr - p * ntile_w;
              float t = acc[s][j] + __shfl_xor(acc[s][j], 16, 64);
              t += __shfl_xor(t, 32, 64);
              if (mq == 0) red[(size_t)s * red_floats + (size_t)(p * S_T + ks) * S_rpw + (rg * Rg + tl) * 16 + n] = t;
            }
          }
        }
      }
    }
  };

  if (nsteps) {
    RawT16 wa[2], wb[2];
    uint32_t g = 0, b = 0;                                  // the step being processed
    auto next = [&](uint32_t& gg, uint32_t& bb) { if (++bb == nblk_w) { bb = 0; ++gg; } };
    x_prime(0);
    issue_step(0, 0, wa);
    for (uint32_t st = 0; st < nsteps; st += 2) {
      uint32_t g1 = g, b1 = b;
      next(g1, b1);
      const bool has1 = st + 1 < nsteps;
      if (has1) issue_step(g1, b1, wb); else issue_step(g, b, wb);
      process(g, b, wa, has1, b1);
      uint32_t g2 = g1, b2 = b1;
      next(g2, b2);
      const bool has2 = st + 2 < nsteps;
      if (has2) issue_step(g2, b2, wa); else issue_step(g, b, wa);
      if (has1) process(g1, b1, wb, has2, b2);
      g = g2; b = b2;
    }
  }

  // ---- RMSNorm: per sequence, the producer's partial sums of x^2 -> ssq[s][0] (wave 0; exactly the single-sequence kernel's order)
  if (nrm && wave == 0) {
    const uint32_t L_n_ssq = L.n_ssq_part;
    for (uint32_t s = 0; s < n_seq; s++) {
      const float* part = L.ssq_part + (size_t)s * B.ssq_stride;
      float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (lane + 64 * j < L_n_ssq) ssp[j] = part[lane + 64 * j];
      float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
      for (uint32_t i = 256 + lane; i < L_n_ssq; i += 64) ss += part[i];
      ss = wave_sum_to_lane63(ss);
      if (lane == 63) ssq[s * 8] = ss;
    }
  } else if (nrm && lane == 0) {
    for (uint32_t s = 0; s < n_seq; s++) ssq[s * 8 + wave] = 0.0f;
  }
  __syncthreads();

  // ---- epilogue, sequence by sequence: the single-sequence epilogue on that sequence's vectors
  for (uint32_t s = 0; s < n_seq; s++) {
    const bool cache = S.epi == EPI_ROPE_K || S.epi == EPI_V_CACHE;
    MvEpiView V;
    V.out = cache ? S.out + (size_t)B.slot[s] * B.cache_stride : S.out + (size_t)s * B.out_stride[sg];
    V.resid = S.resid ? S.resid + (size_t)s * B.resid_stride[sg] : nullptr;
    V.xq_out = S.xq_out ? S.xq_out + (size_t)s * B.xq_out_stride[sg] : nullptr;
    V.xq_ssq = S.xq_ssq ? S.xq_ssq + (size_t)s * B.ssq_out_stride[sg] : nullptr;
    V.pos = B.pos + s;
    mv_epilogue_view(L, S, V, wg, red + (size_t)s * red_floats, ssq + s * 8, S_T);
  }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
size_t mvqb_lds_bytes(uint32_t n_seq, uint32_t red_floats) { return ((size_t)n_seq * red_floats + (size_t)n_seq * 8) * 4 + 64; }

template <uint32_t MASK, int NB>
static hipError_t mvqb_go(const MvLaunch& L, const MvBatch& B, const MvGeom& g, uint32_t threads, size_t lds, hipStream_t st) {
  static bool attr_set[64] = {};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mvqb_kernel<MASK, NB>), 160 * 1024, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL((mvqb_kernel<MASK, NB>), dim3(g.n_wg), dim3(threads), lds, st, g.wbpack, g.geom, g.geom2, L.red_floats, L, B);
  return hipGetLastError();
}

// `L` built like a single-sequence launch (engine.hip: build_mv_group, with the batch cap on tiles per workgroup); sequences'
// vectors at the strides in `B`
hipError_t mvqb_launch(const MvLaunch& L, const MvBatch& B, uint32_t n_wg, uint32_t threads, hipStream_t st) {
  if (B.n_seq == 0 || B.n_seq > (uint32_t)kMaxBatch || threads != kBWaves * 64) return hipErrorInvalidValue;
  MvGeom g;
  size_t lds1 = 0;
  const uint32_t mask = mvq_pack(L, n_wg, threads, &g, &lds1);
  if (!mask) return hipErrorInvalidValue;
  for (int i = 0; i < L.nseg; i++)
    if (L.seg[i].npass > 2 || L.seg[i].pass[0].sel || (L.seg[i].npass == 2 && L.seg[i].pass[1].xq != L.seg[i].pass[0].xq)) return hipErrorInvalidValue;
  const size_t lds = mvqb_lds_bytes(B.n_seq, L.red_floats);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
#define LGH_MVQB_NB(M, NBV) return mvqb_go<M, NBV>(L, B, g, threads, lds, st)
#define LGH_MVQB_CASE(M)                                         \
  case M:                                                        \
    if (B.n_seq <= 4) LGH_MVQB_NB(M, 4);                         \
    else if (B.n_seq <= 8) LGH_MVQB_NB(M, 8);                    \
    else LGH_MVQB_NB(M, 16)
  switch (mask) {
    LGH_MVQB_CASE(1u << F_Q4K); LGH_MVQB_CASE(1u << F_Q6K); LGH_MVQB_CASE(1u << F_Q5K); LGH_MVQB_CASE(1u << F_Q80); LGH_MVQB_CASE(1u << F_Q40);
    LGH_MVQB_CASE((1u << F_Q4K) | (1u << F_Q6K)); LGH_MVQB_CASE((1u << F_Q5K) | (1u << F_Q6K));
    default: return hipErrorInvalidValue;
  }
#undef LGH_MVQB_CASE
#undef LGH_MVQB_NB
}

}  // namespace lgh
