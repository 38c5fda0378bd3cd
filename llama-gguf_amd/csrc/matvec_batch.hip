// matvec_batch.hip — the int8 matrix-core mat-vec for SEVERAL sequences at once (multi-sequence decode, engine_batch.hip:
// lgh_forward_multi; the reference's BatchedEngine steps its active sequences one after the other, src/engine_batched.rs:
// 236-290, 355-400 — every sequence with its own InferenceContext / KV cache).  One launch reads every weight tile ONCE and
// multiplies it with the XQ records of all n_seq input vectors.
//
// Arithmetic.  A sequence's result is bit-identical to the single-sequence kernel (matvec_mfma.hip) on the same input: the
// same k-slices (T, blocks per slice), per slice the blocks in ascending order through the same fused operations
// (mvq_core.h: mvq_mac_tile(unpack, load) == mvq_consume_tile), the same two shuffles, the same sum over the slices and the same
// epilogue (mv_epilogue.h) per sequence.  Only WHO computes differs.
//
// Two structures.  mvqb2 (below, the default from 2 sequences on): a workgroup = 8 (row tile, pass) units x one range of blocks,
// the sequences' records staged ONCE per workgroup in LDS and shared by its 8 waves, sequences two at a time (the pair's 8 MFMAs,
// then the vector work while the next pair's operands load), the epilogue as a second small launch — what bounded the first
// structure and what each change bought: profiles/r03e_batched_decode.md.  mvqb (first): the single-sequence decomposition (a
// workgroup = the 8 k-slices of a few row tiles) with a loop over the sequences inside it and their records from L2 into a
// register ring; kept for one sequence and for launches whose segments do not share their input vector.
// Epilogues: store, +residual, SwiGLU pair, RoPE (every sequence at its own position) and the K / V rows into the sequence's
// own cache slot; optional XQ image of the output for the next launch.  MoE layers: an expert launch works on an INDIRECT
// entry list (MvBatch::ind_*: the (sequence, top-k slot) pairs that chose the expert, their number read from device memory), so
// that an expert's matrices are read once per step; below 7 sequences engine_batch.hip runs the FFN sequence by sequence instead.
#include <algorithm>
#include <cstdlib>

#include "device_utils.h"
#include "xq.h"
#include "mvq_core.h"
#include "mv_epilogue.h"

namespace lgh {

constexpr int kBWaves = 8;

// ---- the tail of a multi-sequence launch: per sequence the RMSNorm's sum of squares (exactly the single-sequence kernel's
// order), then the single-sequence epilogue on that sequence's vectors.  red: [n_seq][red_floats], ssq: [n_seq][8] (LDS)
__device__ __forceinline__ void mvqb_tail(const MvLaunch& L, const MvSeg& S, const MvBatch& B, int sg, uint32_t wg, bool nrm, uint32_t n_seq,
                                          uint32_t red_floats, float* red, float* ssq, uint32_t nslots, const int* ind_idx = nullptr) {
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (nrm) {   // wave w: sequences w, w + waves, ... — per sequence what wave 0 of the single-sequence kernel does; slots 1..7 stay zero
    const uint32_t L_n_ssq = L.n_ssq_part, nw = blockDim.x >> 6;
    for (uint32_t s = wave; s < n_seq; s += nw) {
      const float* part = L.ssq_part + (size_t)(ind_idx ? (uint32_t)ind_idx[s] / B.ind_div : s) * B.ssq_stride;
      float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (lane + 64 * j < L_n_ssq) ssp[j] = part[lane + 64 * j];
      float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
      for (uint32_t i = 256 + lane; i < L_n_ssq; i += 64) ss += part[i];
      ss = wave_sum_to_lane63(ss);
      if (lane == 63) ssq[s * 8] = ss;
      if (lane >= 1 && lane < 8) ssq[s * 8 + lane] = 0.0f;
    }
  }
  __syncthreads();
  // the sequences side by side: a sequence's epilogue needs one thread per row (pair) — groups of `tpr` threads take sequences
  // s, s + n_par, ...; each runs the single-sequence epilogue
  const uint32_t tpr = (S.rows_per_wg + 15u) & ~15u;   // (16-lane rows: what the XQ image's cross-lane steps work on)
  const uint32_t n_par = max(1u, blockDim.x / tpr);
  const uint32_t grp = tid / tpr, tl = tid - grp * tpr;
  if (grp >= n_par) return;
  for (uint32_t s = grp; s < n_seq; s += n_par) {
    const bool cache = S.epi == EPI_ROPE_K || S.epi == EPI_V_CACHE;
    const uint32_t so = ind_idx ? (uint32_t)ind_idx[s] : s;   // whose output vectors (an indirect entry's (sequence, slot) pair)
    MvEpiView V;
    V.out = cache ? S.out + (size_t)B.slot[s] * B.cache_stride : S.out + (size_t)so * B.out_stride[sg];
    V.resid = S.resid ? S.resid + (size_t)so * B.resid_stride[sg] : nullptr;
    V.xq_out = S.xq_out ? S.xq_out + (size_t)so * B.xq_out_stride[sg] : nullptr;
    V.xq_ssq = S.xq_ssq ? S.xq_ssq + (size_t)so * B.ssq_out_stride[sg] : nullptr;
    V.pos = B.pos + s;
    mv_epilogue_view(L, S, V, wg, red + (size_t)s * red_floats, ssq + s * 8, nslots, MvEpiPre{0.0f, 0.0f, false}, tl);
  }
}

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
        mvq_mac_tile<MASK>(fmt, t0, xr[s % kXd], acc[s][0]);
        if (two) mvq_mac_tile<MASK>(fmt, t1, xr[s % kXd], acc[s][1]);
        if ((uint32_t)(s + kXd) < n_seq) x_issue(b, s + kXd, xr[s % kXd]);
      }
    }
    if (more) x_prime(b_next);   // the next step's first sequences, ahead of its weight prefetch
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

  mvqb_tail(L, S, B, sg, wg, nrm, n_seq, red_floats, red, ssq, S_T);
}

// ================================================================================================
// Second structure (mvqb2): the sequences' XQ records through LDS, shared by the eight waves of a workgroup.
//
// Why.  With the k-slices of a row tile spread over the waves of ONE workgroup (above, as in the single-sequence kernel) every wave
// walks its own blocks: an XQ record fetched for a (sequence, block) serves one or two tiles, and a step of 16 sequences moves
// ~2 GB of records per layer from L2 into registers — measured: the memory side alone (arithmetic compiled out) 7.8 ms of the
// 9.9 ms step at 16 sequences, the arithmetic alone 7.6 (profiles/r03e_batched_decode.md).  Here a workgroup is 8 "units" (a unit =
// one row tile of one pass) x ONE range of blocks: all eight waves need the same records, which are staged ONCE per workgroup in
// LDS (cb blocks x n_seq records, <= 40 KB) and read from there — 8x less L2 traffic, no operand ring in registers, half the
// vector registers, two to three workgroups per CU.
//
// Arithmetic.  Unchanged: a unit's blocks in ascending order through mvq_mac_tile; at the end of every k-slice of the
// single-sequence kernel (blocks [ks * nbw, (ks + 1) * nbw)) the four lane groups are folded with the same two shuffles, giving
// the slice's partial sum p_ks.  The single-sequence epilogue adds the slices up as ((0 + p_0) + p_1) + ...:
//   SEQ  = true   a workgroup walks ALL slices of its units and keeps that running sum in registers (same additions, same order);
//                 one value per (sequence, row) goes to the partial buffer (nslots = 1: the epilogue computes 0 + sum = sum);
//   SEQ  = false  (matrices with too few units to fill the chip that way: wo, down, QKV) blockIdx.y = the slice; p_ks goes to
//                 the partial buffer (nslots = T) and the epilogue adds them in slice order.
// Either way a second, small launch (mvqb2_epilogue_kernel) loads the partial sums into LDS in the layout the single-sequence
// epilogue reads and runs exactly that epilogue per sequence — every sequence's result is bit-identical to matvec_mfma.hip's.
// ================================================================================================
struct Mvqb2Geom {
  uint32_t T, nbw, nblk;      // the single-sequence kernel's k-slices
  uint32_t n_units;           // sum over segments of npass * row tiles
  uint32_t ub1, ub2;          // first unit of segment 1 / 2 (n_units when absent)
  uint32_t cb;                // blocks staged per chunk
  uint32_t nslots;            // partial sums per (sequence, row) in the buffer: 1 (SEQ) or T
};

#ifndef LGH_XSTAGE_KB
#define LGH_XSTAGE_KB 40   /* (experiment builds vary it: the cost of a chunk boundary, profiles/r03e_batched_decode.md) */
#endif
constexpr uint32_t kXStageBytes = LGH_XSTAGE_KB * 1024;
// (the two-format instantiations at 16 sequences have no registers to spare: smaller chunks, fewer staging registers)
__host__ __device__ constexpr uint32_t mvqb2_stage_bytes(bool single_format, int nb) { return !single_format && nb == 16 ? 24u * 1024u : kXStageBytes; }
constexpr uint32_t kXZeroBytes = 2 * 1280 + 256;   // the zero block; also what an odd last sequence's partner may read past the records
constexpr uint32_t kXSelBytes = 64;                // behind it: whose input vector entry s reads (s itself, or an indirect entry's sequence)

template <uint32_t MASK, int NB, bool SEQ>
__global__ void __launch_bounds__(kBWaves * 64) mvqb2_kernel(const Mvqb2Geom Gm, const MvLaunch L, const MvBatch B, float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t n_seq = B.n_seq;
  const uint32_t ez = blockIdx.z;                                 // expert (indirect launches over a layer's experts; else 0)
  const int* ind_idx = nullptr;
  if (B.ind_cnt) {   // indirect entries (MoE expert launch): how many (sequence, slot) pairs chose this expert — possibly none
    uint32_t cw;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cw) : "s"(B.ind_cnt + ez) : "memory");
    n_seq = min(cw, n_seq);
    if (n_seq == 0) return;
    ind_idx = B.ind_idx + (size_t)ez * B.ind_stride;
    part += (size_t)ez * B.part_z_floats;
  }
  const uint32_t u = blockIdx.x * kBWaves + wave;                 // this wave's unit
  const bool active = u < Gm.n_units;
  const int sg = (int)(u >= Gm.ub1) + (int)(u >= Gm.ub2);
  const MvSeg& S = L.seg[active ? sg : 0];
  constexpr bool kSingle = (MASK & (MASK - 1)) == 0;
  const int fmt = kSingle ? __builtin_ctz(MASK) : fmt_of_dev_type(S.type);
  auto is = [&](int f) { return ((MASK >> f) & 1u) != 0 && (kSingle || fmt == f); };
  const uint32_t tb = is(F_Q4K) ? fmt_tile_bytes(F_Q4K) : is(F_Q6K) ? fmt_tile_bytes(F_Q6K) : is(F_Q5K) ? fmt_tile_bytes(F_Q5K)
                      : is(F_Q80) ? fmt_tile_bytes(F_Q80) : fmt_tile_bytes(F_Q40);
  const uint32_t ntiles = (S.n_rows + 15) >> 4;
  const uint32_t ul = u - (sg == 0 ? 0u : sg == 1 ? Gm.ub1 : Gm.ub2);
  const uint32_t pass = active ? ul / ntiles : 0, tl = active ? ul - pass * ntiles : 0;
  const uint8_t* wbase = (pass == 0 ? S.pass[0].plane[0] + (uint64_t)ez * S.pass[0].sel_stride[0] : S.pass[1].plane[0] + (uint64_t)ez * S.pass[1].sel_stride[0]) +
                         (size_t)tl * Gm.nblk * tb;
  const uint8_t* xg = L.seg[0].pass[0].xq;                        // sequence 0's records (all segments and passes share the input)
  const uint32_t n = lane & 15, mq = lane >> 4;

  // this workgroup's slices
  const uint32_t ks0 = SEQ ? 0u : blockIdx.y, ks1 = SEQ ? Gm.T : blockIdx.y + 1;
  float acc[NB], tot[NB];
#pragma unroll
  for (int s = 0; s < NB; s++) acc[s] = tot[s] = 0.0f;

  // LDS: [2][cb * n_seq records][a block of zeros].  A lane outside its own k-chunk must feed zeros to the MFMA's A operand: instead of
  // a masked load + a register clear per sequence, such a lane reads the zero block (one unmasked ds_read_b128 per operand).
  const uint32_t zero_off = 2 * Gm.cb * n_seq * (uint32_t)kXqRecord;   // (behind the two halves of the record buffer)
  for (uint32_t i = tid; i < kXZeroBytes / 16; i += kBWaves * 64) *reinterpret_cast<u32x4*>(smem8 + zero_off + i * 16) = u32x4{0u, 0u, 0u, 0u};
  uint32_t* xsel = reinterpret_cast<uint32_t*>(smem8 + zero_off + kXZeroBytes);
  if (tid < n_seq) xsel[tid] = ind_idx ? (uint32_t)ind_idx[tid] / B.ind_div : tid;
  __syncthreads();
  const bool a_valid = (n >> 2) == mq;
  const uint32_t a_off = (mq >> 1) * 128 + (n & 3) * 32 + (mq & 1) * 16;
  const uint32_t a_step = a_valid ? (uint32_t)kXqRecord : 0u;   // per sequence
  struct XqA { i32x4 a[4]; };
  struct XqS { f32x4 xs, sx; };
  auto load_a = [&](uint32_t pa, XqA& o) {
#pragma unroll
    for (int pp = 0; pp < 4; pp++) o.a[pp] = *reinterpret_cast<const i32x4*>(smem8 + pa + pp * 256);
  };
  auto load_s = [&](uint32_t px, XqS& o) {
    o.xs = *reinterpret_cast<const f32x4*>(smem8 + px);
    o.sx = *reinterpret_cast<const f32x4*>(smem8 + px + (kXqSx16 - kXqXs16));
  };

  RawT16 nxt;
  {
    const uint32_t bfirst = ks0 * Gm.nbw;
    if (active && bfirst < Gm.nblk) mvq_issue_tile<MASK>(fmt, wbase + (size_t)bfirst * tb, lane, nxt);
  }
  // ---- chunks of <= cb blocks, never across a slice boundary.  The records of chunk i + 1 are fetched into registers while chunk i
  // is computed, and stored into the other half of a two-deep LDS buffer at the next barrier: one barrier per chunk, no load exposed.
  constexpr uint32_t kPieces = (mvqb2_stage_bytes(kSingle, NB) / 16 + kBWaves * 64 - 1) / (kBWaves * 64);   // 16-byte pieces per thread and chunk
  const uint32_t buf_bytes = Gm.cb * n_seq * (uint32_t)kXqRecord;
  u32x4 stg[kPieces];
  auto chunk_len = [&](uint32_t ks, uint32_t b0) { return min(Gm.cb, min(ks * Gm.nbw + Gm.nbw, Gm.nblk) - b0); };
  auto fetch = [&](uint32_t b0, uint32_t cn) {
    const uint32_t pieces = cn * n_seq * (kXqRecord / 16);
#pragma unroll
    for (uint32_t j = 0; j < kPieces; j++) {
      const uint32_t i = tid + j * kBWaves * 64;
      if (i < pieces) {
        const uint32_t rec = i / (kXqRecord / 16), off = i - rec * (kXqRecord / 16);
        const uint32_t bi = rec / n_seq, s = rec - bi * n_seq;
        const uint32_t sx = xsel[s];                                          // whose input vector
        stg[j] = *reinterpret_cast<const u32x4*>(xg + (size_t)sx * B.xq_stride + (size_t)(b0 + bi) * kXqRecord + off * 16);
      }
    }
  };
  auto stash = [&](uint32_t cn, uint32_t buf) {
    const uint32_t pieces = cn * n_seq * (kXqRecord / 16);
#pragma unroll
    for (uint32_t j = 0; j < kPieces; j++) {
      const uint32_t i = tid + j * kBWaves * 64;
      if (i < pieces) *reinterpret_cast<u32x4*>(smem8 + buf * buf_bytes + (size_t)i * 16) = stg[j];
    }
  };
  // the first non-empty chunk of this workgroup's slices
  uint32_t ks = ks0, b0 = ks0 * Gm.nbw;
  while (ks < ks1 && b0 >= min(ks * Gm.nbw + Gm.nbw, Gm.nblk)) { ks++; b0 = ks * Gm.nbw; }
  uint32_t buf = 0;
  if (ks < ks1) fetch(b0, chunk_len(ks, b0));
  while (ks < ks1) {
    const uint32_t sb1 = min(ks * Gm.nbw + Gm.nbw, Gm.nblk);
    const uint32_t cn = chunk_len(ks, b0);
    stash(cn, buf);
    __syncthreads();
    // the chunk after this one
    uint32_t nks = ks, nb0 = b0 + cn;
    if (nb0 >= sb1) {
      nks = ks + 1; nb0 = nks * Gm.nbw;
      while (nks < ks1 && nb0 >= min(nks * Gm.nbw + Gm.nbw, Gm.nblk)) { nks++; nb0 = nks * Gm.nbw; }
    }
    if (nks < ks1) fetch(nb0, chunk_len(nks, nb0));
    if (active) {
      for (uint32_t bi = 0; bi < cn; bi++) {
        const uint32_t b = b0 + bi;
        const RawT16 cur = nxt;
        {   // the unit's next tile (its next block — of this slice, or of the next one this workgroup walks)
          uint32_t bn = b + 1;
          if (bn == sb1) bn = nks < ks1 ? nb0 : Gm.nblk;
          if (bn < Gm.nblk) mvq_issue_tile<MASK>(fmt, wbase + (size_t)bn * tb, lane, nxt);
        }
        TileOps t;
        mvq_unpack_tile<MASK>(fmt, cur, lane, t);
        const uint32_t xb = buf * buf_bytes + bi * n_seq * (uint32_t)kXqRecord;
        const uint32_t pa0 = a_valid ? xb + a_off : zero_off, px0 = xb + (uint32_t)kXqXs16 + mq * 16;
        // The sequences two at a time: the pair's eight matrix-core products first, then — while the vector ALU turns them into the
        // pair's sums — the NEXT pair's A operands are re-read into the registers the products have just released, and its x sums /
        // scales into the other half of a two-deep buffer.  An odd last sequence is paired with whatever follows it in LDS (the next
        // record or the zero block): computed, never stored.
        XqA xa[2];
        XqS xs2[2][2];
        load_a(pa0, xa[0]);
        load_a(pa0 + a_step, xa[1]);
        load_s(px0, xs2[0][0]);
        load_s(px0 + (uint32_t)kXqRecord, xs2[0][1]);
#ifdef LGH_MVQB2_NOMATH   /* experiment: the weight stream, the unpacking and the record staging only (one sequence's arithmetic) */
        const uint32_t n_seq_math = min(n_seq, 1u);
#else
        const uint32_t n_seq_math = n_seq;
#endif
#pragma unroll
        for (int g = 0; g < NB / 2; g++) {
          if ((uint32_t)(2 * g) < n_seq_math) {
            MacD m0, m1;
            mvq_mac_mfma(t, xa[0].a, m0);
            mvq_mac_mfma(t, xa[1].a, m1);
            if ((uint32_t)(2 * g + 2) < n_seq) {
              load_a(pa0 + (2 * g + 2) * a_step, xa[0]);
              load_a(pa0 + (2 * g + 3) * a_step, xa[1]);
              load_s(px0 + (2 * g + 2) * (uint32_t)kXqRecord, xs2[(g + 1) & 1][0]);
              load_s(px0 + (2 * g + 3) * (uint32_t)kXqRecord, xs2[(g + 1) & 1][1]);
            }
            mvq_mac_finish<MASK>(fmt, t, m0, xs2[g & 1][0].xs, xs2[g & 1][0].sx, acc[2 * g]);
            mvq_mac_finish<MASK>(fmt, t, m1, xs2[g & 1][1].xs, xs2[g & 1][1].sx, acc[2 * g + 1]);
          }
        }
      }
      // ---- end of the k-slice: the four lane groups -> the slice's partial sum (lane group 0), as in the single-sequence kernel
      if (b0 + cn >= sb1) {
#pragma unroll
        for (int s = 0; s < NB; s++) {
          if ((uint32_t)s < n_seq) {
            float t = acc[s] + __shfl_xor(acc[s], 16, 64);
            t += __shfl_xor(t, 32, 64);
            if (SEQ) tot[s] += t;
            else tot[s] = t;
            acc[s] = 0.0f;
          }
        }
      }
    }
    ks = nks; b0 = nb0; buf ^= 1u;
  }
  if (active && mq == 0) {
    const uint32_t slot = SEQ ? 0u : blockIdx.y;
    const size_t rows16 = (size_t)Gm.n_units * 16;
#pragma unroll
    for (int s = 0; s < NB; s++)
      if ((uint32_t)s < n_seq) part[((size_t)slot * B.n_seq + s) * rows16 + (size_t)u * 16 + n] = tot[s];
  }
}

// grid / block = the single-sequence geometry of `L` (segments' wg_begin, rows_per_wg).  dynamic LDS: mvqb_lds_bytes(n_seq, L.red_floats)
__global__ void __launch_bounds__(kBWaves * 64) mvqb2_epilogue_kernel(uint32_t wbpack, uint32_t geom, uint32_t geom2, uint32_t red_floats,
                                                                      const Mvqb2Geom Gm, const MvLaunch L, const MvBatch B,
                                                                      const float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const uint32_t bid = blockIdx.x;
  const int sg = (int)(bid >= (wbpack & 0xFFFFu)) + (int)(bid >= (wbpack >> 16));
  const MvSeg& S = L.seg[sg];
  const uint32_t S_G = (geom >> 8) & 0xFFu;
  const bool nrm = (geom >> 31) != 0;
  const uint32_t Rg = L.nseg > 1 ? S.rows_per_wg >> (4 + __builtin_ctz(S_G)) : geom2 >> 16;
  const uint32_t S_rpw = 16u * Rg * S_G;
  const uint32_t wg = bid - S.wg_begin;
  uint32_t n_seq = B.n_seq;
  const int* ind_idx = nullptr;
  if (B.ind_cnt) {
    uint32_t cw;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cw) : "s"(B.ind_cnt + blockIdx.z) : "memory");
    n_seq = min(cw, n_seq);
    if (n_seq == 0) return;
    ind_idx = B.ind_idx + (size_t)blockIdx.z * B.ind_stride;
    part += (size_t)blockIdx.z * B.part_z_floats;
  }
  const uint32_t nslots = Gm.nslots;
  const uint32_t npass = (uint32_t)S.npass;
  float* red = reinterpret_cast<float*>(smem8);           // [n_seq][red_floats]: red[(p * nslots + slot) * S_rpw + rl]
  float* ssq = red + (size_t)n_seq * red_floats;          // [n_seq][8]
  const uint32_t ntiles = (S.n_rows + 15) >> 4;
  const uint32_t ub = sg == 0 ? 0u : sg == 1 ? Gm.ub1 : Gm.ub2;
  const size_t rows16 = (size_t)Gm.n_units * 16;
  const uint32_t per_seq = npass * nslots * S_rpw;
  for (uint32_t i = threadIdx.x; i < n_seq * per_seq; i += blockDim.x) {
    const uint32_t s = i / per_seq, j = i - s * per_seq;
    const uint32_t ps = j / S_rpw, rl = j - ps * S_rpw;     // ps = p * nslots + slot
    const uint32_t p = ps / nslots, slot = ps - p * nslots;
    const uint32_t row = wg * S_rpw + rl;
    float v = 0.0f;
    if (row < ntiles * 16) v = part[((size_t)slot * B.n_seq + s) * rows16 + (size_t)(ub + p * ntiles) * 16 + row];
    red[(size_t)s * red_floats + j] = v;
  }
  mvqb_tail(L, S, B, sg, wg, nrm, n_seq, red_floats, red, ssq, nslots, ind_idx);
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

template <uint32_t MASK, int NB>
static hipError_t mvqb2_go(const MvLaunch& L, const MvBatch& B, const MvGeom& g, const Mvqb2Geom& Gm, bool seq, uint32_t groups, size_t lds_x,
                           size_t lds_e, uint32_t threads, hipStream_t st) {
  static bool attr_set[3][64] = {};
  const uint32_t nz = B.ind_cnt && B.ind_nz > 1 ? B.ind_nz : 1;   // a layer's experts as the grid's third dimension
  if (seq) {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mvqb2_kernel<MASK, NB, true>), 160 * 1024, attr_set[0]); e != hipSuccess) return e;
    hipLaunchKernelGGL((mvqb2_kernel<MASK, NB, true>), dim3(groups, 1, nz), dim3(kBWaves * 64), lds_x, st, Gm, L, B, B.part);
  } else {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mvqb2_kernel<MASK, NB, false>), 160 * 1024, attr_set[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL((mvqb2_kernel<MASK, NB, false>), dim3(groups, Gm.T, nz), dim3(kBWaves * 64), lds_x, st, Gm, L, B, B.part);
  }
  if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mvqb2_epilogue_kernel), 160 * 1024, attr_set[2]); e != hipSuccess) return e;
  hipLaunchKernelGGL(mvqb2_epilogue_kernel, dim3(g.n_wg, 1, nz), dim3(threads), lds_e, st, g.wbpack, g.geom, g.geom2, L.red_floats, Gm, L, B,
                     (const float*)B.part);
  return hipGetLastError();
}

// 0: the records-through-LDS structure (mvqb2) from LGH_MVQB2_MIN sequences on (default: measured crossover), 1: never, 2: always
static uint32_t mvqb2_min_seq() {
  static const uint32_t v = [] {
    const char* e = std::getenv("LGH_MVQB2_MIN");
    return e ? (uint32_t)std::atoi(e) : 2u;
  }();
  return v;
}

// `L` built like a single-sequence launch (engine.hip: build_mv_group, with the batch cap on tiles per workgroup); sequences'
// vectors at the strides in `B`
hipError_t mvqb_launch(const MvLaunch& L, const MvBatch& Bin, uint32_t n_wg, uint32_t threads, hipStream_t st) {
  MvBatch B = Bin;
  if (B.n_seq == 0 || B.n_seq > (uint32_t)kMaxBatch || threads != kBWaves * 64) return hipErrorInvalidValue;
  MvGeom g;
  size_t lds1 = 0;
  const uint32_t mask = mvq_pack(L, n_wg, threads, &g, &lds1);
  if (!mask) return hipErrorInvalidValue;
  for (int i = 0; i < L.nseg; i++)
    if (L.seg[i].npass > 2 || L.seg[i].pass[0].sel || (L.seg[i].npass == 2 && L.seg[i].pass[1].xq != L.seg[i].pass[0].xq)) return hipErrorInvalidValue;
  const size_t lds = mvqb_lds_bytes(B.n_seq, L.red_floats);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (B.ind_cnt && (B.part == nullptr || B.n_seq < mvqb2_min_seq() || !B.ind_idx || B.ind_div == 0)) return hipErrorInvalidValue;   // indirect entries: mvqb2 only
  bool shared_x = B.part != nullptr;
  for (int i = 0; i < L.nseg; i++) shared_x = shared_x && L.seg[i].pass[0].xq == L.seg[0].pass[0].xq;
  if (shared_x && B.n_seq >= mvqb2_min_seq()) {
    const MvSeg& S0 = L.seg[0];
    Mvqb2Geom Gm{};
    Gm.T = S0.T; Gm.nbw = S0.units; Gm.nblk = S0.nblk;
    uint32_t ub[4] = {0, 0, 0, 0};
    for (int i = 0; i < L.nseg; i++) ub[i + 1] = ub[i] + (uint32_t)L.seg[i].npass * ((L.seg[i].n_rows + 15) / 16);
    Gm.n_units = ub[L.nseg];
    Gm.ub1 = L.nseg > 1 ? ub[1] : Gm.n_units;
    Gm.ub2 = L.nseg > 2 ? ub[2] : Gm.n_units;
    const uint32_t groups = (Gm.n_units + kBWaves - 1) / kBWaves;
    static const uint32_t seq_min_groups = [] { const char* e = std::getenv("LGH_MVQB2_SEQ_GROUPS"); return e ? (uint32_t)std::atoi(e) : 160u; }();
    const bool seq = groups >= seq_min_groups || Gm.T == 1;  // enough units to fill the chip with whole rows per workgroup
    Gm.nslots = seq ? 1 : Gm.T;
    const int nb_inst = B.n_seq <= 4 ? 4 : B.n_seq <= 8 ? 8 : 16;
    Gm.cb = std::max(1u, std::min(Gm.nbw, mvqb2_stage_bytes((mask & (mask - 1)) == 0, nb_inst) / (B.n_seq * (uint32_t)kXqRecord)));
    const size_t lds_x = (size_t)2 * Gm.cb * B.n_seq * kXqRecord + kXZeroBytes + kXSelBytes;
    B.part_z_floats = (uint64_t)Gm.nslots * B.n_seq * Gm.n_units * 16;
    if (B.part_z_floats * (B.ind_cnt && B.ind_nz > 1 ? B.ind_nz : 1) <= B.part_floats) {
#define LGH_MVQB2_CASE(M)                                                                         \
  case M:                                                                                         \
    if (B.n_seq <= 4) return mvqb2_go<M, 4>(L, B, g, Gm, seq, groups, lds_x, lds, threads, st);  \
    if (B.n_seq <= 8) return mvqb2_go<M, 8>(L, B, g, Gm, seq, groups, lds_x, lds, threads, st);  \
    return mvqb2_go<M, 16>(L, B, g, Gm, seq, groups, lds_x, lds, threads, st)
      switch (mask) {
        LGH_MVQB2_CASE(1u << F_Q4K); LGH_MVQB2_CASE(1u << F_Q6K); LGH_MVQB2_CASE(1u << F_Q5K); LGH_MVQB2_CASE(1u << F_Q80); LGH_MVQB2_CASE(1u << F_Q40);
        LGH_MVQB2_CASE((1u << F_Q4K) | (1u << F_Q6K)); LGH_MVQB2_CASE((1u << F_Q5K) | (1u << F_Q6K));
        default: break;
      }
#undef LGH_MVQB2_CASE
    }
  }
  if (B.ind_cnt) return hipErrorInvalidValue;   // (indirect entries exist in the mvqb2 structure only)
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
