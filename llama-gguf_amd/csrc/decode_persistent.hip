// decode_persistent.hip — the persistent token kernel: ONE launch runs every quantized mat-vec and the attention of a
// decode step (QKV -> attention -> wo -> gate/up -> down per layer, then the output projection) on 256 resident
// workgroups, one per CU.
//
// What it replaces: the reference launches ~20 driver calls per layer (src/backend/cuda/gpu_only.rs:860-1024); round 1 of
// this engine replayed one hipGraph of 6 kernels per layer, and measured that every graph node costs >= 4.2 us of fixed
// time (dispatch gap + wave launch + the first dependent round trips to HBM + drain) next to bytes / 5.9 TB/s of
// streaming — HBM idled half the time.  A grid barrier between ops costs as much as a kernel boundary (measured: 553 vs
// 604 tokens/s for the barrier version, matvec_mfma.hip: mvq_chain_kernel), so there is none here.  Instead:
//
//   * DATA-FLOW HAND-OFFS.  An op's epilogue writes its output (f32 + XQ records, xq.h) with write-through (sc1) stores,
//     drains them, and one lane adds the rows it produced to the counters of the output's signal groups (one counter
//     per 256-element XQ record, or per attention head).  A consumer WAVE polls only the counters of the k-slice it
//     needs (sc1 loads), then pulls that slice into LDS with sc1 LDS-DMA.  Nobody waits for the whole grid.  Counters
//     only ever grow: the target of token t is (epoch + 1) x rows-per-group, the epoch lives in device memory and is
//     advanced by the last workgroup to finish, so a replayed hipGraph needs no host-side change and no reset.
//   * THE WEIGHT STREAM RUNS AHEAD OF THE DEPENDENCY.  Weights do not depend on activations.  Every wave walks ONE flat
//     list of weight tiles across all ops of the token and keeps kPtDepth tiles in flight in registers; when it reaches the
//     end of an op it is already fetching the next op's tiles, so while it waits for the input vector of the next op
//     (one hand-off latency) the first 72+ KB per CU of that op are landing.
//   * ATTENTION IN THE SAME LAUNCH.  n_kv x splits workgroups run the split-KV online-softmax attention as soon as their kv
//     head's q / k / v rows are signalled; the split partials are merged by the CONSUMER (each wo wave merges the heads of its
//     own k-slice and converts them to XQ in LDS), so there is no combine kernel and no second hand-off.
//
// The arithmetic is the launch-per-op kernels': tiles go through mvq_core.h, epilogues through mv_epilogue.h (COH = true),
// the same k-slice / row-group geometry, so a mat-vec op gives the same bits as mvq_kernel.  Attention differs only in the
// number of splits (a different, equally valid merge order of the online softmax).
//
// Safety: every spin is bounded (kPtSpinLimit polls with s_sleep); a timeout raises sync[16] and the kernel runs to its end.
// All 256 workgroups are resident (the launch is kNumCU workgroups of 512 threads whose registers and LDS allow one per
// CU), which the hand-offs need for progress; the host refuses the path on a device that reports fewer CUs.
#include <algorithm>
#include <type_traits>

#include "device_utils.h"
#include "mv_epilogue.h"
#include "mvq_core.h"
#include "ptok.h"
#include "xq.h"

namespace lgh {

#ifndef PT_DEPTH
#define PT_DEPTH 3
#endif
constexpr int kPtDepth = PT_DEPTH;   // weight tiles a wave keeps in flight (registers)
constexpr float kPtNegBig = -1e30f;

// Diagnostic build only (-DLGH_STAMPS, `make stamps`): every wave of ONE workgroup records s_memrealtime (100 MHz) at the
// phase boundaries of every op into a buffer nothing else reads (tools/pt_phases.py prints the profile).
#ifdef LGH_STAMPS
constexpr int kPtStampOps = 1024, kPtStampWg = 1;
__device__ unsigned long long g_pt_stamps[kPtStampOps * kPtWaves * 8];
hipError_t ptok_read_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pt_stamps), n * sizeof(unsigned long long));
}
#define PT_STAMP(i)                                                                                           \
  do {                                                                                                        \
    if (blockIdx.x == kPtStampWg && (threadIdx.x & 63) == 0 && op < (uint32_t)kPtStampOps)                     \
      g_pt_stamps[(op * kPtWaves + ((threadIdx.x >> 6) & 7)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime();         \
  } while (0)
#else
#define PT_STAMP(i)
#endif

__device__ __forceinline__ unsigned pt_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float pt_ldf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pt_stf(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Lanes l < n each wait until counter (first + l) has reached `target` (wrap-safe); the whole wave leaves together.
__device__ __forceinline__ void pt_wait(unsigned* sync, uint32_t first, uint32_t n, unsigned target, uint32_t lane) {
#ifdef PT_DEBUG_KNOBS
  if (pt_ld(sync + 48) & 1u) return;
#endif
  const unsigned* c = sync + kPtSyncHeader + (size_t)(first + (lane < n ? lane : 0)) * kPtCntStride;
  unsigned spins = 0;
  for (;;) {
    const bool ok = lane >= n || (int)(pt_ld(c) - target) >= 0;
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if ((spins & 255u) == 0 && (spins > kPtSpinLimit || pt_ld(sync + 16) != 0)) {   // timed out (here, or somewhere else already)
      if (lane == 0) __hip_atomic_store(sync + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      break;
    }
  }
}

__device__ __forceinline__ void pt_signal(unsigned* sync, uint32_t counter, unsigned add) {
  __hip_atomic_fetch_add(sync + kPtSyncHeader + (size_t)counter * kPtCntStride, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// splits in use at this position (same value in the attention op and in its consumer)
__device__ __forceinline__ uint32_t pt_splits(const PtAttn& A, uint32_t kv_len) {
  uint32_t s = (kv_len + A.rows_per_split - 1) / A.rows_per_split;
  return s < 1 ? 1 : (s > A.s_max ? A.s_max : s);
}

// ------------------------------------------------------------------------------------------------
// a wave's share of one mat-vec op (all scalars)
// ------------------------------------------------------------------------------------------------
struct PtGeo {
  const uint8_t *pb0, *pb1, *pb2, *pb3;   // per pass: first tile of this wave
  uint32_t nblk_w, ntile_w, npass, nblk;  // blocks per tile row of this wave, its tiles, passes, blocks per matrix row
  uint32_t fmt, tb;
  uint32_t nitems;
  // consume side
  uint32_t seg, wg, ks, rg, T, Rg, rpw, blk0;
  bool in_op;                             // this workgroup has rows in the op
};

__device__ __forceinline__ void pt_geo(const PtOp& O, const MvLaunch& L, uint32_t bid, uint32_t wave, PtGeo& g) {
  g.nitems = 0; g.nblk_w = 0; g.ntile_w = 0; g.npass = 0;
  g.in_op = bid < O.n_wg;
  if (!g.in_op) return;
  const uint32_t s = (uint32_t)(bid >= (O.wbpack & 0xFFFFu)) + (uint32_t)(bid >= (O.wbpack >> 16));
  const MvSeg& S = L.seg[s];
  const uint32_t T = O.geom & 0xFFu, G = (O.geom >> 8) & 0xFFu, nbw = (O.geom >> 16) & 0x3FFFu;
  const uint32_t nblk = O.geom2 & 0xFFFFu, Rg = O.geom2 >> 16;
  g.seg = s; g.T = T; g.Rg = Rg; g.nblk = nblk;
  g.rpw = 16u * Rg * G;
  g.wg = bid - S.wg_begin;
  uint32_t ks = wave, rg = 0;
  while (ks >= T) { ks -= T; rg++; }
  g.ks = ks; g.rg = rg;
  const bool active = rg < G;
  g.blk0 = ks * nbw;
  g.nblk_w = active && g.blk0 < nblk ? min(nbw, nblk - g.blk0) : 0;
  const uint32_t ntiles = (S.n_rows + 15) >> 4;
  const uint32_t tile0 = (g.wg * G + rg) * Rg;
  g.ntile_w = active && tile0 < ntiles ? min(Rg, ntiles - tile0) : 0;
  g.fmt = (uint32_t)fmt_of_dev_type(S.type);
  g.tb = fmt_tile_bytes((int)g.fmt);
  g.npass = (uint32_t)S.npass;
  const uint64_t woff = ((uint64_t)tile0 * nblk + g.blk0) * g.tb;
  g.pb0 = S.pass[0].plane[0] + woff;
  g.pb1 = g.npass > 1 ? S.pass[1].plane[0] + woff : nullptr;
  g.pb2 = g.npass > 2 ? S.pass[2].plane[0] + woff : nullptr;
  g.pb3 = g.npass > 3 ? S.pass[3].plane[0] + woff : nullptr;
  g.nitems = g.npass * g.ntile_w * g.nblk_w;
}

// ------------------------------------------------------------------------------------------------
// attention op: workgroup (kv head, split slot); NW = 8 waves, rows dealt round-robin over splits x waves
// ------------------------------------------------------------------------------------------------
#ifndef PT_ATTN_INLINE
#define PT_ATTN_INLINE __forceinline__
#endif
template <int D, int G>
__device__ PT_ATTN_INLINE void pt_attention(const PtAttn& A, uint32_t bid, unsigned epoch, unsigned* sync, uint8_t* smem8, uint32_t tid, uint32_t op) {
  constexpr int NW = kPtWaves;
  constexpr int LPR = D / 4;       // lanes per row
  constexpr int RPW = 64 / LPR;    // rows per wave-instruction
  const uint32_t n_kv = A.n_kv;
  const uint32_t kvh = bid % n_kv, sp = bid / n_kv;
  if (sp >= A.s_max) return;       // not an attention workgroup
  const uint32_t lane = tid & 63, wave = tid >> 6;
  const uint32_t pos = __builtin_amdgcn_readfirstlane((uint32_t)*A.pos), kv_len = pos + 1;   // (written before this launch)
  const uint32_t n_splits = pt_splits(A, kv_len);
  if (sp >= n_splits) {            // an unused split slot still arrives, so that every token adds s_max to the counter
    if (tid == 0) pt_signal(sync, A.out_cnt + kvh, 1u);
    return;
  }
  float (*s_ml)[G][2] = reinterpret_cast<float (*)[G][2]>(smem8 + A.lds_off);   // (a region of its own: the mat-vec ops' x / partial-sum regions stay live)
  float (*s_acc)[G][D] = reinterpret_cast<float (*)[G][D]>(smem8 + A.lds_off + sizeof(float) * NW * G * 2);

  const uint32_t sub = lane / LPR, li = lane % LPR;
  const float* kbase = A.kc + (size_t)kvh * A.max_seq * D + li * 4;
  const float* vbase = A.vc + (size_t)kvh * A.max_seq * D + li * 4;
  const uint32_t stride = n_splits * NW * RPW;
  // rows [0, pos) were written by earlier launches: requested NOW, before this token's q exists (plain loads); the row of
  // the current token comes from the QKV op of this launch and is read coherently after the wait, by split 0 / wave 0
#ifndef PT_AHEAD
#define PT_AHEAD 4
#endif
  constexpr int kAhead = PT_AHEAD;
  f32x4 kk[kAhead], vv[kAhead];
  const uint32_t base0 = (sp * NW + wave) * RPW;
  auto row_of = [&](uint32_t base) { const uint32_t p = base + sub; return p < pos ? p : (pos ? pos - 1 : 0); };
  auto request = [&](uint32_t base) {
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const uint32_t r = row_of(base + j * stride);
      kk[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kbase + (size_t)r * D));
      vv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vbase + (size_t)r * D));
    }
  };
  if (base0 < pos) request(base0);

  // ---- wait for this kv head's q heads, k row and v row (wave 0 polls, the workgroup's barrier releases the rest)
#ifdef PT_DEBUG_KNOBS
  const bool nowait = (pt_ld(sync + 48) & 1u) != 0;
#else
  constexpr bool nowait = false;
#endif
  if (wave == 0 && !nowait) {
    const unsigned target = (epoch + 1u) * (unsigned)D;
    const uint32_t c = lane < (uint32_t)G ? A.in_cnt + kvh * G + lane : lane == (uint32_t)G ? A.in_cnt + A.n_heads + kvh : A.in_cnt + A.n_heads + n_kv + kvh;
    const unsigned* cp = sync + kPtSyncHeader + (size_t)c * kPtCntStride;
    unsigned spins = 0;
    for (;;) {
      const bool ok = lane >= (uint32_t)G + 2 || (int)(pt_ld(cp) - target) >= 0;
      if (__all(ok)) break;
      __builtin_amdgcn_s_sleep(2);
      ++spins;
      if ((spins & 255u) == 0 && (spins > kPtSpinLimit || pt_ld(sync + 16) != 0)) {
        if (lane == 0) __hip_atomic_store(sync + 16, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  PT_STAMP(1);
  __syncthreads();
  PT_STAMP(2);

  f32x4 qv[G];
#pragma unroll
  for (int g = 0; g < G; g++) {
    const float* qp = A.q + ((size_t)kvh * G + g) * D + li * 4;
    qv[g].x = pt_ldf(qp); qv[g].y = pt_ldf(qp + 1); qv[g].z = pt_ldf(qp + 2); qv[g].w = pt_ldf(qp + 3);
  }
  float m[G], l[G];
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; g++) { m[g] = kPtNegBig; l[g] = 0.0f; acc[g] = (f32x4)(0.0f); }
  const float scale = A.scale;
  auto step = [&](bool valid, f32x4 k4, f32x4 v4) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      float s = qv[g].x * k4.x;
      s = __builtin_fmaf(qv[g].y, k4.y, s);
      s = __builtin_fmaf(qv[g].z, k4.z, s);
      s = __builtin_fmaf(qv[g].w, k4.w, s);
      s += dpp_f<0xB1>(s);
      s += dpp_f<0x4E>(s);
      s += dpp_f<0x141>(s);
      s += dpp_f<0x140>(s);
      if (LPR == 32) s += __shfl_xor(s, 16, 64);
      s *= scale;
      const float mn = valid ? fmaxf(m[g], s) : m[g];
      const float a = __expf(m[g] - mn);
      const float pe = valid ? __expf(s - mn) : 0.0f;
      l[g] = __builtin_fmaf(l[g], a, pe);
      acc[g] = acc[g] * a + v4 * pe;
      m[g] = mn;
    }
  };
  for (uint32_t base = base0; base < pos; base += kAhead * stride) {
    if (base != base0) request(base);
#pragma unroll
    for (int j = 0; j < kAhead; j++)
      if (base + j * stride < pos) step(base + j * stride + sub < pos, kk[j], vv[j]);   // wave-uniform condition
  }
  if (sp == 0 && wave == 0) {   // the current token's row
    const float* kp = kbase + (size_t)pos * D;
    const float* vp = vbase + (size_t)pos * D;
    f32x4 k4, v4;
    k4.x = pt_ldf(kp); k4.y = pt_ldf(kp + 1); k4.z = pt_ldf(kp + 2); k4.w = pt_ldf(kp + 3);
    v4.x = pt_ldf(vp); v4.y = pt_ldf(vp + 1); v4.z = pt_ldf(vp + 2); v4.w = pt_ldf(vp + 3);
    step(sub == 0, k4, v4);
  }

  PT_STAMP(3);
  // merge the RPW row slots of the wave (lanes with equal li hold the same output dims)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      const float om = __shfl_xor(m[g], off, 64), ol = __shfl_xor(l[g], off, 64);
      f32x4 oa;
      oa.x = __shfl_xor(acc[g].x, off, 64); oa.y = __shfl_xor(acc[g].y, off, 64);
      oa.z = __shfl_xor(acc[g].z, off, 64); oa.w = __shfl_xor(acc[g].w, off, 64);
      const float mn = fmaxf(m[g], om);
      const float a = expf(m[g] - mn), b = expf(om - mn);
      l[g] = l[g] * a + ol * b;
      acc[g] = acc[g] * a + oa * b;
      m[g] = mn;
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      if (li == 0) { s_ml[wave][g][0] = m[g]; s_ml[wave][g][1] = l[g]; }
      *reinterpret_cast<f32x4*>(&s_acc[wave][g][li * 4]) = acc[g];
    }
  }
  __syncthreads();
  PT_STAMP(4);
  // merge the waves and publish the split's partial: ml[G][2], acc[G][D]
  float* part = A.part + ((size_t)kvh * A.s_max + sp) * (size_t)(G * (D + 2));
  for (uint32_t e = tid; e < (uint32_t)(G * D); e += NW * 64) {
    const uint32_t g = e / D, dim = e % D;
    float mn = s_ml[0][g][0];
#pragma unroll
    for (int w = 1; w < NW; w++) mn = fmaxf(mn, s_ml[w][g][0]);
    float lsum = 0.0f, a = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float f = expf(s_ml[w][g][0] - mn);
      lsum += s_ml[w][g][1] * f;
      a += s_acc[w][g][dim] * f;
    }
    pt_stf(part + 2 * G + g * D + dim, a);
    if (dim == 0) { pt_stf(part + g * 2, mn); pt_stf(part + g * 2 + 1, lsum); }
  }
  PT_STAMP(5);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this thread's write-through stores have left
  PT_STAMP(6);
  __syncthreads();
  if (tid == 0) pt_signal(sync, A.out_cnt + kvh, 1u);
}

// The consumer side of attention: a wo wave merges the split partials of the heads in ITS k-slice (elements
// [blk0 * 256, (blk0 + nblk_w) * 256) of the attention output) and writes them as XQ records into its LDS region.
// Two memory round trips in all: the splits' (m, l) of the slice's heads, then every accumulator the slice needs, four
// splits at a time, all requested before the first one is used.
template <int D, int G>
__device__ __forceinline__ void pt_gather_attn(const PtAttn& A, uint32_t blk0, uint32_t nblk_w, unsigned epoch, unsigned* sync,
                                               uint8_t* xrec, uint32_t lane, uint32_t kv_len) {
  const uint32_t n_splits = pt_splits(A, kv_len);
  const uint32_t e0 = blk0 * 256, e1 = e0 + nblk_w * 256;
  const uint32_t kv0 = e0 / (D * G), kv1 = (e1 - 1) / (D * G);
  pt_wait(sync, A.out_cnt + kv0, kv1 - kv0 + 1, (epoch + 1u) * A.s_max, lane);
  constexpr uint32_t pstride = (uint32_t)(G * (D + 2));
  const uint32_t nit = nblk_w * 4;                      // 64 consecutive elements per step: one head (64 divides D)
  for (uint32_t it0 = 0; it0 < nit; it0 += 4) {         // four steps (256 elements) at a time
    // the (m, l) of step j's head, split s = lane & 31 (lanes >= 32 mirror the lower half; s_max <= 32)
    float fw[4], linv[4];
    float ms[4], ls[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint32_t h = (e0 + (it0 + j) * 64) / D, kvh = h / G, g = h % G, s = lane & 31;
      const float* p = A.part + ((size_t)kvh * A.s_max + (s < n_splits ? s : 0)) * pstride + g * 2;
      ms[j] = pt_ldf(p);
      ls[j] = pt_ldf(p + 1);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const bool live = (lane & 31) < n_splits;
      const float m = live ? ms[j] : kPtNegBig;
      const float mn = wave_max(m);
      fw[j] = live ? expf(m - mn) : 0.0f;
      const float lsum = wave_sum(lane < 32 ? ls[j] * fw[j] : 0.0f);
      linv[j] = 1.0f / lsum;                            // simd.rs:718-720: multiply by 1/sum
    }
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (uint32_t s0 = 0; s0 < n_splits; s0 += 4) {
      float v[4][4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const uint32_t e = e0 + (it0 + j) * 64 + lane, h = e / D, dim = e % D, kvh = h / G, g = h % G;
        const float* p0 = A.part + (size_t)kvh * A.s_max * pstride + 2 * G + g * D + dim;
#pragma unroll
        for (int t = 0; t < 4; t++) v[j][t] = pt_ldf(p0 + (size_t)(s0 + t < n_splits ? s0 + t : s0) * pstride);
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
          const float fs = __shfl(fw[j], (int)(s0 + t), 64);
          a[j] = __builtin_fmaf(v[j][t], s0 + t < n_splits ? fs : 0.0f, a[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) xq_store_chunk<false>(xrec, (it0 + j) * 4 + (lane >> 4), a[j] * linv[j], nullptr, 0.0f, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// the kernel
// ------------------------------------------------------------------------------------------------
template <uint32_t MASK, int D, int G>
__global__ void __launch_bounds__(kPtWaves * 64) ptok_kernel(const PtProgram P) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const uint32_t bid = blockIdx.x, tid0 = threadIdx.x, lane0 = tid0 & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid0 >> 6);
  unsigned* sync = P.sync;
  const unsigned epoch = pt_ld(sync);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem8;

  // ---- the issue side: a flat stream of this wave's tiles over all mat-vec ops, kPtDepth of them in flight
  RawT16 buf[kPtDepth];
  uint32_t meta[kPtDepth];         // b | tl << 8 | p << 16 | fmt << 20 | last block of its tile << 24
  uint32_t iop = P.first_mv;
  PtGeo gi;
  gi.nitems = 0;
  uint32_t ip = 0, itl = 0, ib = 0, ileft = 0;
  auto issue_find = [&]() {        // moves to the next op in which this wave has tiles (iop already points at a candidate)
    while (iop != kPtNone) {
      const PtOp& O = P.ops[iop];
      pt_geo(O, P.mv[O.mv], bid, wave, gi);
      if (gi.nitems) { ip = itl = ib = 0; ileft = gi.nitems; return; }
      iop = O.next_mv;
    }
    ileft = 0;
  };
  auto issue_one = [&](RawT16& r, uint32_t& mt, uint32_t lane) {
    if (ileft == 0) { mt = kPtNone; return; }
    const uint8_t* base = ip == 0 ? gi.pb0 : ip == 1 ? gi.pb1 : ip == 2 ? gi.pb2 : gi.pb3;
    mvq_issue_tile<MASK>((int)gi.fmt, base + ((size_t)itl * gi.nblk + ib) * gi.tb, lane, r);
    mt = ib | itl << 8 | ip << 16 | gi.fmt << 20 | (uint32_t)(ib + 1 == gi.nblk_w) << 24;
    if (++ib == gi.nblk_w) { ib = 0; if (++itl == gi.ntile_w) { itl = 0; ++ip; } }
    if (--ileft == 0) { iop = P.ops[iop].next_mv; issue_find(); }
  };
  issue_find();
#pragma unroll
  for (int j = 0; j < kPtDepth; j++) issue_one(buf[j], meta[j], lane0);

  // ---- the ops, in program order
  for (uint32_t op = 0; op < P.nops; op++) {
    const PtOp& O = P.ops[op];
    // A per-op opaque copy of the thread id: without it every lane-dependent constant of the attention, gather and epilogue
    // code (dozens of registers) is hoisted in front of this loop and lives across it, next to the ring — the kernel then
    // needs more than 256 VGPRs and spills.
    uint32_t tid = tid0;
    asm volatile("" : "+v"(tid));
#ifdef PT_EXP_LANE0
    const uint32_t lane = lane0;
#else
    const uint32_t lane = tid & 63;
#endif
    const uint32_t n = lane & 15, c = lane >> 4;
    PT_STAMP(0);
    if (O.kind == PT_ATTN) {
#ifndef PT_EXP_NOATTN
      pt_attention<D, G>(P.attn[O.attn], bid, epoch, sync, smem8, tid, op);
#endif
      PT_STAMP(7);
      continue;
    }
    const MvLaunch& L = P.mv[O.mv];
    PtGeo gc;
    pt_geo(O, L, bid, wave, gc);
    if (!gc.in_op) continue;
    const MvSeg& S = L.seg[gc.seg];
    const uint32_t nbw = (O.geom >> 16) & 0x3FFFu;
    const bool nrm = (O.geom >> 31) != 0;
    uint8_t* xrec = smem8 + wave * nbw * kXqRecord;
    const uint32_t xrec_lds = lds_base + wave * nbw * kXqRecord;
    float* red = reinterpret_cast<float*>(smem8 + O.lds_red_off);
    float* ssq = red + O.red_floats;

    // epilogue operands that do not depend on the mat-vec
    MvEpiPre epi_pre = {0.0f, 0.0f, false};
    mv_epilogue_prefetch_resid<true>(S.epi, S.resid, S.xq_nw, S.n_rows, gc.wg, gc.rpw, epi_pre, tid);
    if (S.epi == EPI_ROPE_Q || S.epi == EPI_ROPE_K) {
      const uint32_t pos_now = __builtin_amdgcn_readfirstlane((uint32_t)*L.pos);
      mv_epilogue_prefetch_rope(S.epi, pos_now, L.rope_cs, S.head_dim, S.n_rows, gc.wg, gc.rpw, epi_pre, tid);
    }

    // ---- the input vector of this wave's k-slice -> its LDS region
    float ss_w = 0.0f;
    if (gc.nitems) {
      if (O.in_kind == PT_IN_ATTN) {
#ifndef PT_EXP_NOGATHER
        pt_gather_attn<D, G>(P.attn[O.attn], gc.blk0, gc.nblk_w, epoch, sync, xrec, lane, __builtin_amdgcn_readfirstlane((uint32_t)*L.pos) + 1);
#endif
      } else {
        if (O.in_kind == PT_IN_XQ) pt_wait(sync, O.in_cnt + gc.blk0, gc.nblk_w, (epoch + 1u) * 256u, lane);
        PT_STAMP(1);
        const uint8_t* xg = S.pass[0].xq + (size_t)gc.blk0 * kXqRecord;
        for (uint32_t b = 0; b < gc.nblk_w; b++) {     // 1024 B + 256 B per record by LDS-DMA (agent-scope loads)
          const uint8_t* src = xg + b * kXqRecord;
          const uint32_t dst = __builtin_amdgcn_readfirstlane(xrec_lds + b * kXqRecord);
          uint32_t keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off sc1\n\ts_add_u32 m0, m0, 0x400\n\t"
                       "s_nop 0\n\tglobal_load_lds_dword %2, off sc1\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(src + lane * 16), "v"(src + 1024 + lane * 4), "s"(dst) : "memory");
        }
        // RMSNorm: the producer's sums of x^2 per 16-element chunk, this wave's k-slice only (row-group 0 contributes)
        if (nrm && gc.rg == 0) {
          const uint32_t nchunk = gc.nblk_w * 16;
          for (uint32_t i = lane; i < nchunk; i += 64) ss_w += pt_ldf(L.ssq_part + gc.blk0 * 16 + i);
        }
      }
    } else if (gc.nblk_w == 0 && gc.ntile_w > 0) {
      // a k-slice beyond the last block (T does not divide the block count): its partial-sum slots must read as zero
      for (uint32_t p = 0; p < gc.npass; p++)
        for (uint32_t tl = 0; tl < gc.ntile_w; tl++)
          if (c == 0) red[(size_t)(p * gc.T + gc.ks) * gc.rpw + (gc.rg * gc.Rg + tl) * 16 + n] = 0.0f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the records (and every prefetched tile behind them) have landed
    if (O.in_kind == PT_IN_ATTN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    PT_STAMP(2);

    // ---- this wave's tiles of the op: consume one, put the next one of the flat stream in flight
    float acc = 0.0f;
    auto consume = [&](uint32_t mt, const RawT16& r) {
      const uint32_t b = mt & 0xFFu, tl = (mt >> 8) & 0xFFu, p = (mt >> 16) & 0xFu;
      mvq_consume_tile<MASK>((int)((mt >> 20) & 0xFu), r, xrec + b * kXqRecord, lane, acc);
      if ((mt >> 24) & 1u) {      // last block of a (pass, tile): the four lane groups -> one partial sum per row
        float t = acc + __shfl_xor(acc, 16, 64);
        t += __shfl_xor(t, 32, 64);
        if (c == 0) red[(size_t)(p * gc.T + gc.ks) * gc.rpw + (gc.rg * gc.Rg + tl) * 16 + n] = t;
        acc = 0.0f;
      }
    };
    const uint32_t nit = gc.nitems;
    for (uint32_t i = 0; i + kPtDepth <= nit; i += kPtDepth) {
#pragma unroll
      for (int j = 0; j < kPtDepth; j++) { consume(meta[j], buf[j]); issue_one(buf[j], meta[j], lane); }
    }
    const uint32_t rem = nit % kPtDepth;
#pragma unroll
    for (int j = 0; j < kPtDepth - 1; j++)
      if ((uint32_t)j < rem) { consume(meta[j], buf[j]); issue_one(buf[j], meta[j], lane); }

    PT_STAMP(3);
    if (nrm) {
      ss_w = wave_sum_to_lane63(ss_w);
      if (lane == 63) ssq[wave] = ss_w;
    }
    __syncthreads();
    PT_STAMP(4);
#ifndef PT_EXP_NOEPI
    mv_epilogue<true>(L, S, gc.wg, red, ssq, gc.T, epi_pre, tid);
#endif
    PT_STAMP(5);
    // ---- signal: the rows this workgroup produced, per signal group of the segment's output.  Only the waves that ran the
    // epilogue have stores to drain; the others are already on their way to the next op (the partial-sum buffers alternate
    // between ops, so nothing they write next is still being read here).
    const uint32_t oc = O.out_cnt[gc.seg];
    const uint32_t epi_threads = (S.epi == EPI_ROPE_Q || S.epi == EPI_ROPE_K) ? gc.rpw / 2 : gc.rpw;
    const bool many = epi_threads > 64;
    if (oc != kPtNone && (many || wave == 0)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through stores have left
    PT_STAMP(6);
    if (oc != kPtNone && many) __syncthreads();
    if (oc != kPtNone && wave == 0) {
      const uint32_t sh = O.out_shift[gc.seg];
      const uint32_t r0 = gc.wg * gc.rpw, r1 = min(r0 + gc.rpw, S.n_rows);
      if (r0 < r1) {
        const uint32_t g0 = r0 >> sh, g1 = (r1 - 1) >> sh;
        if (lane <= g1 - g0) {
          const uint32_t lo = max(r0, (g0 + lane) << sh), hi = min(r1, (g0 + lane + 1) << sh);
          pt_signal(sync, oc + g0 + lane, hi - lo);
        }
      }
    }
    PT_STAMP(7);
    // the ring starts every op at slot 0: rotate the consumed slots to the back (their loads have landed)
#pragma unroll
    for (int r = 0; r < kPtDepth - 1; r++) {
      if ((uint32_t)r < rem) {
        const RawT16 t0 = buf[0];
        const uint32_t m0 = meta[0];
#pragma unroll
        for (int j = 0; j + 1 < kPtDepth; j++) { buf[j] = buf[j + 1]; meta[j] = meta[j + 1]; }
        buf[kPtDepth - 1] = t0;
        meta[kPtDepth - 1] = m0;
      }
    }
  }

  // ---- the last workgroup to finish opens the next token's epoch (every workgroup has read this one's by then)
  __syncthreads();
  if (tid0 == 0) {
    const unsigned done = __hip_atomic_fetch_add(sync + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1 == (epoch + 1u) * gridDim.x) __hip_atomic_store(sync, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
size_t pt_part_floats(uint32_t n_kv, uint32_t s_max, uint32_t group, uint32_t head_dim) {
  return (size_t)n_kv * s_max * group * (head_dim + 2);
}

size_t ptok_layout_lds(PtHostOp* ops, size_t n, uint32_t head_dim, uint32_t group) {
  size_t x_max = 0, red_max = 0;
  for (size_t i = 0; i < n; i++)
    if (ops[i].op.kind == PT_MV) {
      x_max = std::max(x_max, (size_t)ops[i].op.lds_red_off);                       // = waves x blocks per k-slice x record bytes
      red_max = std::max(red_max, (size_t)(ops[i].op.red_floats + 16) * 4);
    }
  x_max = (x_max + 255) / 256 * 256;
  red_max = (red_max + 255) / 256 * 256;
  uint32_t parity = 0;
  for (size_t i = 0; i < n; i++) {
    if (ops[i].op.kind == PT_MV) { ops[i].op.lds_red_off = (uint32_t)(x_max + parity * red_max); parity ^= 1u; }
    else ops[i].attn.lds_off = (uint32_t)(x_max + 2 * red_max);
  }
  return x_max + 2 * red_max + (size_t)kPtWaves * group * (head_dim + 2) * 4 + 256;
}

uint32_t ptok_mask(const PtHostOp* ops, size_t n) {
  uint32_t m = 0;
  for (size_t i = 0; i < n; i++) {
    if (ops[i].op.kind != PT_MV) continue;
    for (int s = 0; s < ops[i].mv.nseg; s++) {
      const int f = fmt_of_dev_type(ops[i].mv.seg[s].type);
      if (f < 0) return 0;
      m |= 1u << f;
    }
  }
  // instantiated: the Q4_K / Q6_K family (Q4_K_M), the Q5_K / Q6_K family (Q5_K_M), Q8_0, Q4_0
  const uint32_t q46 = (1u << F_Q4K) | (1u << F_Q6K), q56 = (1u << F_Q5K) | (1u << F_Q6K);
  if (m && (m & ~q46) == 0) return q46;
  if (m && (m & ~q56) == 0) return q56;
  if (m == (1u << F_Q80)) return m;
  return 0;   // (Q4_0-only models stay on the launch-per-op path: that instantiation keeps its tile ring on the stack)
}

bool ptok_supported(uint32_t mask, uint32_t head_dim, uint32_t group) {
  const uint32_t q46 = (1u << F_Q4K) | (1u << F_Q6K), q56 = (1u << F_Q5K) | (1u << F_Q6K);
  if (mask != q46 && mask != q56 && mask != (1u << F_Q80)) return false;
  return (head_dim == 128 && (group == 4 || group == 8)) || (head_dim == 64 && (group == 2 || group == 4 || group == 8));
}

template <uint32_t MASK, int D, int G>
static hipError_t ptok_go(const PtProgram& P, size_t lds, hipStream_t st) {
  static bool attr_set[64] = {};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&ptok_kernel<MASK, D, G>), 160 * 1024, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL((ptok_kernel<MASK, D, G>), dim3(kNumCU), dim3(kPtWaves * 64), lds, st, P);
  return hipGetLastError();
}

template <uint32_t MASK>
static hipError_t ptok_dg(const PtProgram& P, uint32_t d, uint32_t g, size_t lds, hipStream_t st) {
  if (d == 128 && g == 4) return ptok_go<MASK, 128, 4>(P, lds, st);
  if (d == 128 && g == 8) return ptok_go<MASK, 128, 8>(P, lds, st);
  if (d == 64 && g == 2) return ptok_go<MASK, 64, 2>(P, lds, st);
  if (d == 64 && g == 4) return ptok_go<MASK, 64, 4>(P, lds, st);
  if (d == 64 && g == 8) return ptok_go<MASK, 64, 8>(P, lds, st);
  return hipErrorInvalidValue;
}

hipError_t ptok_launch(const PtProgram& P, uint32_t mask, uint32_t head_dim, uint32_t group, size_t lds, hipStream_t st) {
  constexpr uint32_t q46 = (1u << F_Q4K) | (1u << F_Q6K), q56 = (1u << F_Q5K) | (1u << F_Q6K);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  switch (mask) {
    case q46: return ptok_dg<q46>(P, head_dim, group, lds, st);
    case q56: return ptok_dg<q56>(P, head_dim, group, lds, st);
    case 1u << F_Q80: return ptok_dg<(1u << F_Q80)>(P, head_dim, group, lds, st);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace lgh
