// matvec.hip — dequant-fused quantized mat-vec for single-token decode on gfx950.
//
// Replaces the reference's `vec_mat_q*` CUDA kernels (src/backend/cuda/kernels.rs:443-729: one thread
// per output column walking K serially over blocks at stride n*144 — uncoalesced) and their call
// sites `linear_gpu` / `dense_ffn_gpu_forward` / `moe_gpu_forward`
// (src/backend/cuda/gpu_only.rs:209-329, 1605-1651, 1765-2011).  Arithmetic follows the CPU backend's
// fused dots (src/backend/cpu/simd.rs:931-1146): x stays f32 (no activation quantization); per
// 32-group `sum(q*x)` and `sum(x)` are accumulated and the f16 scales applied once per group.
//
// Design for CDNA4 (HBM-bound, 3.5 flop/B):
//   * ONE workgroup per CU (grid ~ 256); its waves are T k-slices x G row-groups.  x is loaded and (with the
//     RMSNorm prologue) scaled by the norm weight ONCE per workgroup and staged through LDS; every wave then
//     keeps its k-slice of x (and the per-16 sums used by the min/offset terms) in registers for the whole kernel;
//   * weights are the only streamed operand: each lane issues 16-byte non-temporal loads straight to VGPRs,
//     software-pipelined in double-buffered batches of B rows with static load counts (no LDS round trip);
//   * partial sums of the k-slices meet in LDS once, at the end of the workgroup;
//   * the mat-vec is linear in x, so the RMSNorm factor 1/rms is applied per output row in the epilogue; residual
//     add / SwiGLU / RoPE + KV-cache write / MoE expert mixing are epilogues too (mv_epilogue.h), so a dense layer
//     is 6 launches instead of the reference's ~20.
// Q4_K launches go to the MFMA kernel (matvec_mfma.hip); this kernel serves Q5_K, Q6_K, Q8_0, Q4_0 and mixed launches.
#include "device_utils.h"
#include "mv_epilogue.h"

namespace lgh {

// Diagnostic build only (-DLGH_STAMPS): wave 0 of every workgroup records s_memrealtime (100 MHz) at phase
// boundaries into a buffer nothing else reads; lgh_bench_vec_mat prints the phase profile.  Never in the product .so.
#ifdef LGH_STAMPS
__device__ unsigned long long g_stamps[8192 * 8];
#define LGH_STAMP(i)                                                                             \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
hipError_t mv_read_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#else
#define LGH_STAMP(i)
#endif

enum : uint32_t { M_Q4K = 1, M_Q5K = 2, M_Q6K = 4, M_Q80 = 8, M_Q40 = 16, M_ALL = 31 };

// x runs of 16 consecutive floats into registers.  With the RMSNorm prologue the registers hold x*w (the norm
// weight) and `ss` accumulates sum(x^2) of the raw values: the mat-vec is linear in x, so the 1/rms factor
// is applied once per output row in the epilogue instead of per element before the dot — the reduction over
// all of x then overlaps the weight stream instead of preceding it.
// ALL loads of all runs are issued before any value is used (a branch between loads would make hipcc wait
// for each one in turn: measured 4 us of serialized L2 round trips).
template <int NRUN>
__device__ __forceinline__ void load_runs(float (*dst)[16], const float* const* xp, const float* const* wp, float& ss,
                                          bool do_norm) {
  f32x4 xv[NRUN][4], wv[NRUN][4];
#pragma unroll
  for (int r = 0; r < NRUN; r++)
#pragma unroll
    for (int i = 0; i < 4; i++) xv[r][i] = *reinterpret_cast<const f32x4*>(xp[r] + 4 * i);
  if (do_norm) {
#pragma unroll
    for (int r = 0; r < NRUN; r++)
#pragma unroll
      for (int i = 0; i < 4; i++) wv[r][i] = *reinterpret_cast<const f32x4*>(wp[r] + 4 * i);
#pragma unroll
    for (int r = 0; r < NRUN; r++)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        f32x4 v = xv[r][i];
        ss = __builtin_fmaf(v.x, v.x, ss);
        ss = __builtin_fmaf(v.y, v.y, ss);
        ss = __builtin_fmaf(v.z, v.z, ss);
        ss = __builtin_fmaf(v.w, v.w, ss);
        xv[r][i] = v * wv[r][i];
      }
  }
#pragma unroll
  for (int r = 0; r < NRUN; r++)
#pragma unroll
    for (int i = 0; i < 4; i++) {
      dst[r][4 * i + 0] = xv[r][i].x; dst[r][4 * i + 1] = xv[r][i].y;
      dst[r][4 * i + 2] = xv[r][i].z; dst[r][4 * i + 3] = xv[r][i].w;
    }
}

__device__ __forceinline__ float sum16(const float* v) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += v[i];
  return s;
}

// ------------------------------------------------------------------------------------------ Q4_K
// unit u of a row: block u>>3, nibble pair p=(u>>1)&3, half h=u&1.  The lane's 16 qs bytes
// qs[32p+16h .. +16) hold elements 64p+16h+j (low nibbles, sub-block 2p) and 64p+32+16h+j (high
// nibbles, sub-block 2p+1)   — layout of dequant.rs:232-255.
struct FmtQ4K {
  static constexpr int B = 3;
  static constexpr uint32_t UNIT = 32;
  struct X { float v[2][16]; float slo, shi; uint32_t sh; uint32_t upper; };  // v[0] low-nibble run, v[1] high
  struct Raw { u32x4 hd, qs; };

  static __device__ __forceinline__ void load_x(X& X_, const float* x, const float* nw, float& ss, bool nrm, uint32_t u) {
    uint32_t p = (u >> 1) & 3;
    uint32_t e0 = (u >> 3) * 256 + p * 64 + (u & 1) * 16;
    const float* xp[2] = {x + e0, x + e0 + 32};
    const float* wp[2] = {nw + e0, nw + e0 + 32};
    load_runs<2>(X_.v, xp, wp, ss, nrm);
    X_.slo = sum16(X_.v[0]);
    X_.shi = sum16(X_.v[1]);
    X_.sh = (p & 1) * 16;
    X_.upper = p >> 1;
  }
  static __device__ __forceinline__ void load(Raw& r, const uint8_t* const* pl, uint32_t row, uint32_t nblk, uint32_t u) {
    const uint8_t* b = pl[0] + ((size_t)row * nblk + (u >> 3)) * 144;
    r.hd = ldg_nt128(b);
    r.qs = ldg_nt128(b + 16 + (u & 7) * 16);
  }
  // 6-bit (scale,min) of sub-blocks 2p and 2p+1 (dequant.rs:210-223), two at a time in 16-bit lanes
  static __device__ __forceinline__ void scales(const u32x4& hd, const X& X_, float& sc_lo, float& sc_hi, float& m_lo,
                                                float& m_hi) {
    uint32_t a = (hd.y >> X_.sh) & 0xFFFFu, b = (hd.z >> X_.sh) & 0xFFFFu, c = (hd.w >> X_.sh) & 0xFFFFu;
    uint32_t sc_l = a & 0x3F3Fu, m_l = b & 0x3F3Fu;
    uint32_t sc_u = (c & 0x0F0Fu) | ((a >> 2) & 0x3030u);
    uint32_t m_u = ((c >> 4) & 0x0F0Fu) | ((b >> 2) & 0x3030u);
    uint32_t sc2 = X_.upper ? sc_u : sc_l, m2 = X_.upper ? m_u : m_l;
    sc_lo = ub0(sc2); sc_hi = ub1(sc2);
    m_lo = ub0(m2); m_hi = ub1(m2);
  }
  static __device__ __forceinline__ float dot(const Raw& r, const X& X_) {
    // one short FMA chain per dword (8 independent chains): a dependent v_fmac issues only every ~8 cycles
    float pl[2], ph[2];  // two chains per nibble half: enough ILP to cover the dependent-FMA latency, few registers
#pragma unroll
    for (int w = 0; w < 4; w++) {
      uint32_t v = r.qs[w];
      if (w < 2) {
        pl[w] = fma4z(opaque(v & 0x0F0F0F0Fu), X_.v[0] + 4 * w);
        ph[w] = fma4z(opaque((v >> 4) & 0x0F0F0F0Fu), X_.v[1] + 4 * w);
      } else {
        pl[w - 2] = fma4(opaque(v & 0x0F0F0F0Fu), X_.v[0] + 4 * w, pl[w - 2]);
        ph[w - 2] = fma4(opaque((v >> 4) & 0x0F0F0F0Fu), X_.v[1] + 4 * w, ph[w - 2]);
      }
    }
    const float alo = pl[0] + pl[1], ahi = ph[0] + ph[1];
    float sc_lo, sc_hi, m_lo, m_hi;
    scales(r.hd, X_, sc_lo, sc_hi, m_lo, m_hi);
    float d = h2f(r.hd.x & 0xFFFFu), dmin = h2f(r.hd.x >> 16);
    return d * __builtin_fmaf(sc_lo, alo, sc_hi * ahi) - dmin * __builtin_fmaf(m_lo, X_.slo, m_hi * X_.shi);
  }
};

// ------------------------------------------------------------------------------------------ Q5_K
// as Q4_K plus qh[32]: bit 2p of qh[16h+j] is the 5th bit of the low-nibble element, bit 2p+1 of the
// high-nibble element (dequant.rs:287-315).
struct FmtQ5K {
  static constexpr int B = 3;
  static constexpr uint32_t UNIT = 32;
  struct X { float v[2][16]; float slo, shi; uint32_t sh; uint32_t upper; uint32_t qsh; };
  struct Raw { u32x4 hd, qh, qs; };

  static __device__ __forceinline__ void load_x(X& X_, const float* x, const float* nw, float& ss, bool nrm, uint32_t u) {
    uint32_t p = (u >> 1) & 3;
    uint32_t e0 = (u >> 3) * 256 + p * 64 + (u & 1) * 16;
    const float* xp[2] = {x + e0, x + e0 + 32};
    const float* wp[2] = {nw + e0, nw + e0 + 32};
    load_runs<2>(X_.v, xp, wp, ss, nrm);
    X_.slo = sum16(X_.v[0]);
    X_.shi = sum16(X_.v[1]);
    X_.sh = (p & 1) * 16;
    X_.upper = p >> 1;
    X_.qsh = 2 * p;
  }
  static __device__ __forceinline__ void load(Raw& r, const uint8_t* const* pl, uint32_t row, uint32_t nblk, uint32_t u) {
    const uint8_t* b = pl[0] + ((size_t)row * nblk + (u >> 3)) * 176;
    r.hd = ldg_nt128(b);
    r.qh = ldg_nt128(b + 16 + (u & 1) * 16);
    r.qs = ldg_nt128(b + 48 + (u & 7) * 16);
  }
  static __device__ __forceinline__ float dot(const Raw& r, const X& X_) {
    float pl[2], ph[2];
#pragma unroll
    for (int w = 0; w < 4; w++) {
      uint32_t v = r.qs[w], t = r.qh[w] >> X_.qsh;
      uint32_t qlo = (v & 0x0F0F0F0Fu) | ((t << 4) & 0x10101010u);
      uint32_t qhi = ((v >> 4) & 0x0F0F0F0Fu) | ((t << 3) & 0x10101010u);
      if (w < 2) {
        pl[w] = fma4z(opaque(qlo), X_.v[0] + 4 * w);
        ph[w] = fma4z(opaque(qhi), X_.v[1] + 4 * w);
      } else {
        pl[w - 2] = fma4(opaque(qlo), X_.v[0] + 4 * w, pl[w - 2]);
        ph[w - 2] = fma4(opaque(qhi), X_.v[1] + 4 * w, ph[w - 2]);
      }
    }
    const float alo = pl[0] + pl[1], ahi = ph[0] + ph[1];
    float sc_lo, sc_hi, m_lo, m_hi;
    FmtQ4K::X sx;  // reuse the Q4_K scale unpack (same 12-byte packing)
    sx.sh = X_.sh; sx.upper = X_.upper;
    FmtQ4K::scales(r.hd, sx, sc_lo, sc_hi, m_lo, m_hi);
    float d = h2f(r.hd.x & 0xFFFFu), dmin = h2f(r.hd.x >> 16);
    return d * __builtin_fmaf(sc_lo, alo, sc_hi * ahi) - dmin * __builtin_fmaf(m_lo, X_.slo, m_hi * X_.shi);
  }
};

// ------------------------------------------------------------------------------------------ Q6_K
// planes: ql[128], qh[64], scales[16] (i8), d per block.  unit u: block u>>2, half n=(u>>1)&1,
// column group c=u&1 (l = 16c..16c+15).  Quarter t of the half (elements 128n+32t+l) takes its low
// nibble from ql[64n+32(t&1)+l] (high nibble for t>=2) and bits 2t..2t+1 of qh[32n+l]; scale index
// 8n+c+2t  (dequant.rs:321-356).  q-32 is folded in as  sum(q*x) - 32*sum(x).
struct FmtQ6K {
  static constexpr int B = 2;
  static constexpr uint32_t UNIT = 64;
  struct X { float q[4][16]; float s[4]; uint32_t shc; };
  struct Raw { u32x4 qa, qb, qh; u32x2 sc; uint32_t d; };

  static __device__ __forceinline__ void load_x(X& X_, const float* x, const float* nw, float& ss, bool nrm, uint32_t u) {
    uint32_t e0 = (u >> 2) * 256 + ((u >> 1) & 1) * 128 + (u & 1) * 16;
    const float* xp[4] = {x + e0, x + e0 + 32, x + e0 + 64, x + e0 + 96};
    const float* wp[4] = {nw + e0, nw + e0 + 32, nw + e0 + 64, nw + e0 + 96};
    load_runs<4>(X_.q, xp, wp, ss, nrm);
#pragma unroll
    for (int t = 0; t < 4; t++) X_.s[t] = sum16(X_.q[t]);
    X_.shc = (u & 1) * 8;
  }
  static __device__ __forceinline__ void load(Raw& r, const uint8_t* const* pl, uint32_t row, uint32_t nblk, uint32_t u) {
    size_t blk = (size_t)row * nblk + (u >> 2);
    uint32_t n = (u >> 1) & 1, c = u & 1;
    const uint8_t* ql = pl[0] + blk * 128 + n * 64 + c * 16;
    r.qa = ldg_nt128(ql);
    r.qb = ldg_nt128(ql + 32);
    r.qh = ldg_nt128(pl[1] + blk * 64 + n * 32 + c * 16);
    r.sc = ldg_nt64(pl[2] + blk * 16 + n * 8);
    r.d = ldg_nt16(pl[3] + blk * 2);
  }
  static __device__ __forceinline__ float dot(const Raw& r, const X& X_) {
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;  // four independent chains (one per quarter)
#pragma unroll
    for (int w = 0; w < 4; w++) {
      uint32_t a = r.qa[w], b = r.qb[w], h = r.qh[w];
      uint32_t v0 = (a & 0x0F0F0F0Fu) | ((h << 4) & 0x30303030u);
      uint32_t v1 = (b & 0x0F0F0F0Fu) | ((h << 2) & 0x30303030u);
      uint32_t v2 = ((a >> 4) & 0x0F0F0F0Fu) | (h & 0x30303030u);
      uint32_t v3 = ((b >> 4) & 0x0F0F0F0Fu) | ((h >> 2) & 0x30303030u);
      a0 = fma4(opaque(v0), X_.q[0] + 4 * w, a0);
      a1 = fma4(opaque(v1), X_.q[1] + 4 * w, a1);
      a2 = fma4(opaque(v2), X_.q[2] + 4 * w, a2);
      a3 = fma4(opaque(v3), X_.q[3] + 4 * w, a3);
    }
    uint64_t sv = (((uint64_t)r.sc.y << 32) | r.sc.x) >> X_.shc;
    float s0 = (float)(int)(int8_t)(sv), s1 = (float)(int)(int8_t)(sv >> 16);
    float s2 = (float)(int)(int8_t)(sv >> 32), s3 = (float)(int)(int8_t)(sv >> 48);
    float t = s0 * __builtin_fmaf(-32.0f, X_.s[0], a0);
    t = __builtin_fmaf(s1, __builtin_fmaf(-32.0f, X_.s[1], a1), t);
    t = __builtin_fmaf(s2, __builtin_fmaf(-32.0f, X_.s[2], a2), t);
    t = __builtin_fmaf(s3, __builtin_fmaf(-32.0f, X_.s[3], a3), t);
    return h2f(r.d) * t;
  }
};

// ------------------------------------------------------------------------------------------ Q8_0
// planes: qs[32] (i8), d per block; unit = one block.  q is read as q+128 (sign bit flipped) so the
// unsigned byte->f32 convert applies:  sum(q*x) = sum((q^0x80)*x) - 128*sum(x).
struct FmtQ80 {
  static constexpr int B = 3;
  static constexpr uint32_t UNIT = 32;
  struct X { float v[2][16]; float s; };
  struct Raw { u32x4 q0, q1; uint32_t d; };

  static __device__ __forceinline__ void load_x(X& X_, const float* x, const float* nw, float& ss, bool nrm, uint32_t u) {
    const float* xp[2] = {x + u * 32, x + u * 32 + 16};
    const float* wp[2] = {nw + u * 32, nw + u * 32 + 16};
    load_runs<2>(X_.v, xp, wp, ss, nrm);
    X_.s = sum16(X_.v[0]) + sum16(X_.v[1]);
  }
  static __device__ __forceinline__ void load(Raw& r, const uint8_t* const* pl, uint32_t row, uint32_t nblk, uint32_t u) {
    size_t blk = (size_t)row * nblk + u;
    r.q0 = ldg_nt128(pl[0] + blk * 32);
    r.q1 = ldg_nt128(pl[0] + blk * 32 + 16);
    r.d = ldg_nt16(pl[1] + blk * 2);
  }
  static __device__ __forceinline__ float dot(const Raw& r, const X& X_) {
    float pa[2], pb[2];
#pragma unroll
    for (int w = 0; w < 2; w++) {
      pa[w] = fma4z(opaque(r.q0[w] ^ 0x80808080u), X_.v[0] + 4 * w);
      pb[w] = fma4z(opaque(r.q1[w] ^ 0x80808080u), X_.v[1] + 4 * w);
    }
#pragma unroll
    for (int w = 2; w < 4; w++) {
      pa[w - 2] = fma4(opaque(r.q0[w] ^ 0x80808080u), X_.v[0] + 4 * w, pa[w - 2]);
      pb[w - 2] = fma4(opaque(r.q1[w] ^ 0x80808080u), X_.v[1] + 4 * w, pb[w - 2]);
    }
    const float a = (pa[0] + pa[1]) + (pb[0] + pb[1]);
    return h2f(r.d) * __builtin_fmaf(-128.0f, X_.s, a);
  }
};

// ------------------------------------------------------------------------------------------ Q4_0
// planes: qs[16], d per block; unit = one block: low nibble j -> element j, high nibble -> 16+j
// (dequant.rs:16-30);  (q-8) folded in as sum(q*x) - 8*sum(x).
struct FmtQ40 {
  static constexpr int B = 2;
  static constexpr uint32_t UNIT = 32;
  struct X { float v[2][16]; float s; };
  struct Raw { u32x4 qs; uint32_t d; };

  static __device__ __forceinline__ void load_x(X& X_, const float* x, const float* nw, float& ss, bool nrm, uint32_t u) {
    const float* xp[2] = {x + u * 32, x + u * 32 + 16};
    const float* wp[2] = {nw + u * 32, nw + u * 32 + 16};
    load_runs<2>(X_.v, xp, wp, ss, nrm);
    X_.s = sum16(X_.v[0]) + sum16(X_.v[1]);
  }
  static __device__ __forceinline__ void load(Raw& r, const uint8_t* const* pl, uint32_t row, uint32_t nblk, uint32_t u) {
    size_t blk = (size_t)row * nblk + u;
    r.qs = ldg_nt128(pl[0] + blk * 16);
    r.d = ldg_nt16(pl[1] + blk * 2);
  }
  static __device__ __forceinline__ float dot(const Raw& r, const X& X_) {
    float pl[2], ph[2];  // two chains per nibble half: enough ILP to cover the dependent-FMA latency, few registers
#pragma unroll
    for (int w = 0; w < 4; w++) {
      uint32_t v = r.qs[w];
      if (w < 2) {
        pl[w] = fma4z(opaque(v & 0x0F0F0F0Fu), X_.v[0] + 4 * w);
        ph[w] = fma4z(opaque((v >> 4) & 0x0F0F0F0Fu), X_.v[1] + 4 * w);
      } else {
        pl[w - 2] = fma4(opaque(v & 0x0F0F0F0Fu), X_.v[0] + 4 * w, pl[w - 2]);
        ph[w - 2] = fma4(opaque((v >> 4) & 0x0F0F0F0Fu), X_.v[1] + 4 * w, ph[w - 2]);
      }
    }
    const float alo = pl[0] + pl[1], ahi = ph[0] + ph[1];
    return h2f(r.d) * __builtin_fmaf(-8.0f, X_.s, alo + ahi);
  }
};

// ------------------------------------------------------------------------------------------ body

// block-wide variant for the f32 kernel
__device__ __forceinline__ float block_inv_rms(const float* x, uint32_t k, float eps, float* wsum) {
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  float ss = 0.0f;
  for (uint32_t i = tid * 4; i < k; i += nthr * 4) {
    f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
    ss = __builtin_fmaf(v.x, v.x, ss);
    ss = __builtin_fmaf(v.y, v.y, ss);
    ss = __builtin_fmaf(v.z, v.z, ss);
    ss = __builtin_fmaf(v.w, v.w, ss);
  }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.0f;
  const uint32_t nw = nthr >> 6;
  for (uint32_t w = 0; w < nw; w++) tot += wsum[w];
  float rms = __builtin_sqrtf(tot / (float)k + eps);
  return 1.0f / rms;
}

constexpr int kXPT = 8;  // float4 x-loads a thread keeps in flight while staging x into LDS

// Workgroup prologue + the streaming loop of one wave.
//
// Prologue (whole workgroup, ONE workgroup per CU): x is read from L2 exactly once per workgroup — each thread
// fetches a few float4s, applies the norm weight, accumulates its share of sum(x^2) and parks the values in LDS,
// from where every wave then pulls the K-slice its lanes own into registers.  (Letting each wave fetch its own
// slice from L2 cost as many bytes as the weights themselves: 4096 waves x 16 KB on the gate/up launch.)
// The first weight loads are issued BEFORE x is waited for, and the barrier that publishes x in LDS does not
// drain them (explicit lgkmcnt-only wait + s_barrier).
//
// Stream: work items = (pass, batch of F::B rows), flattened so the pipeline runs across passes (gate -> up).
// Two register buffers alternate: the loads of item i+1 are in flight while item i is being reduced.  Every
// load of an item is unconditional (out-of-range rows are clamped onto the last valid one) so that hipcc counts
// the outstanding loads statically and emits partial `s_waitcnt vmcnt(N)` instead of draining the next batch.
template <class F>
__device__ __forceinline__ void mv_rows(const MvLaunch& L, const MvSeg& S, uint32_t wg, float* xs, float* red, float* ssq) {
  const uint32_t tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63;
  // wave-uniform bookkeeping lives in SGPRs (readfirstlane): no vector divides, no exec-masked branches
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool active = wave < S.T * S.G;
  const uint32_t ks = active ? wave % S.T : 0, rg = active ? wave / S.T : 0;
  const uint32_t u = ks * 64 + lane;
  const bool uvalid = u < S.units;
  const uint32_t uc = uvalid ? u : S.units - 1;
  const uint32_t rpg = S.rows_per_wg / S.G;
  const uint32_t row0 = wg * S.rows_per_wg + rg * rpg;
  const bool nrm = L.do_norm != 0;
  const bool has_rows = active && row0 < S.n_rows;
  const uint32_t rows_here = has_rows ? min(rpg, S.n_rows - row0) : 0;
  const uint32_t nb = (rows_here + F::B - 1) / F::B;
  const uint32_t rbase = has_rows ? row0 : 0;

  auto issue = [&](uint32_t p, uint32_t j, typename F::Raw* buf) {
    const MvPass& P = S.pass[p];
    const uint64_t e = P.sel ? (uint64_t)(uint32_t)(*P.sel) : 0;
    const uint8_t* pl[4];
#pragma unroll
    for (int i = 0; i < 4; i++) pl[i] = P.plane[i] + e * P.sel_stride[i];
#pragma unroll
    for (int b = 0; b < F::B; b++) F::load(buf[b], pl, rbase + min(j * F::B + b, rows_here - 1), S.nblk, uc);
  };
  auto consume = [&](uint32_t p, uint32_t j, const typename F::Raw* buf, const typename F::X& X_) {
    float* rp = red + (size_t)(p * S.T + ks) * S.rows_per_wg + rg * rpg;
#pragma unroll
    for (int b = 0; b < F::B; b++) {
      const uint32_t rl = j * F::B + b;
      if (rl < rows_here) {  // clamped duplicate rows are loaded (static load count) but not reduced
        float part = F::dot(buf[b], X_);
        part = wave_sum_to_lane63(uvalid ? part : 0.0f);
        if (lane == 63) rp[rl] = part;
      }
      __builtin_amdgcn_sched_barrier(0);  // keep each row's work together: hoisting all converts up front spills
    }
  };

  typename F::Raw A[F::B], Bq[F::B];
  LGH_STAMP(0);
#ifdef LGH_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + 6] = __builtin_amdgcn_s_memtime();
#endif
  uint32_t ip = 0, ij = 0;  // (pass, batch) of the next item to issue
  auto advance = [&]() { if (++ij == nb) { ij = 0; ++ip; } };
  uint32_t ap = 0, aj = 0, bp = 0, bj = 0;

  // ---- prologue: x -> (x*w, sum x^2) -> LDS, with the first weight batch already in flight
  const float* xg = S.pass[0].x;
  const uint32_t k4 = L.k >> 2;
  float ss = 0.0f;
  for (uint32_t base = 0; base < k4; base += kXPT * nthr) {
    f32x4 xv[kXPT], wv[kXPT];
#pragma unroll
    for (int j = 0; j < kXPT; j++) {
      const uint32_t i = base + tid + j * nthr;
      if (i < k4) xv[j] = reinterpret_cast<const f32x4*>(xg)[i];
    }
    if (nrm) {
#pragma unroll
      for (int j = 0; j < kXPT; j++) {
        const uint32_t i = base + tid + j * nthr;
        if (i < k4) wv[j] = reinterpret_cast<const f32x4*>(L.norm_w)[i];
      }
    }
    if (base == 0 && has_rows) { issue(0, 0, A); advance(); }
#pragma unroll
    for (int j = 0; j < kXPT; j++) {
      const uint32_t i = base + tid + j * nthr;
      if (i < k4) {
        f32x4 v = xv[j];
        if (nrm) {
          ss = __builtin_fmaf(v.x, v.x, ss);
          ss = __builtin_fmaf(v.y, v.y, ss);
          ss = __builtin_fmaf(v.z, v.z, ss);
          ss = __builtin_fmaf(v.w, v.w, ss);
          v = v * wv[j];
        }
        reinterpret_cast<f32x4*>(xs)[i] = v;
      }
    }
  }
  if (nrm) {  // this wave's share of sum(x^2); the epilogue adds the shares of all waves
    ss = wave_sum_to_lane63(ss);
    if (lane == 63) ssq[wave] = ss;
  }
  // publish x: wait for the LDS writes only — the weight loads stay in flight across the barrier
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  LGH_STAMP(1);
  if (!has_rows) return;

  typename F::X X_;
  const float* xcur = xg;
  {
    float dummy = 0.0f;
    F::load_x(X_, xs, nullptr, dummy, false, uc);
  }
  auto fix_x = [&](uint32_t p) {  // a later pass with a different input vector (MoE down): straight from L2
    const float* xp = S.pass[p].x;
    if (xp != xcur) { xcur = xp; float dummy = 0.0f; F::load_x(X_, xcur, nullptr, dummy, false, uc); }
  };
  uint32_t remaining = (uint32_t)S.npass * nb;
  while (remaining > 2) {   // steady state: every issue / consume is unconditional
    bp = ip; bj = ij; issue(ip, ij, Bq); advance();
    fix_x(ap);
    consume(ap, aj, A, X_);
    ap = ip; aj = ij; issue(ip, ij, A); advance();
    fix_x(bp);
    consume(bp, bj, Bq, X_);
    remaining -= 2;
  }
  if (remaining == 2) {
    bp = ip; bj = ij; issue(ip, ij, Bq);
    fix_x(ap);
    consume(ap, aj, A, X_);
    fix_x(bp);
    consume(bp, bj, Bq, X_);
  } else {
    fix_x(ap);
    consume(ap, aj, A, X_);
  }
  LGH_STAMP(3);
}

// MAXT: launch bound = the widest workgroup of the instantiation (one workgroup per CU): 16 waves at <=128
// VGPRs for Q4_K / Q8_0 / Q4_0, 12 waves at <=168 for Q5_K, 8 waves at <=256 for Q6_K and mixed-type launches.
// Dynamic LDS: x[K] | red[npass*T*rows_per_wg] | ssq[16].
template <uint32_t MASK, int MAXT>
__global__ void __launch_bounds__(MAXT) mv_kernel(const MvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;
  float* red = smem + L.k;
  float* ssq = red + L.red_floats;
  int s = 0;
  const uint32_t bid = blockIdx.x;
  if (L.nseg > 1 && bid >= L.seg[1].wg_begin) s = 1;
  if (L.nseg > 2 && bid >= L.seg[2].wg_begin) s = 2;
  const MvSeg& S = L.seg[s];
  const uint32_t wg = bid - S.wg_begin;
  switch (S.type) {
    case LGH_TYPE_Q4_K: if constexpr (MASK & M_Q4K) mv_rows<FmtQ4K>(L, S, wg, xs, red, ssq); break;
    case LGH_TYPE_Q5_K: if constexpr (MASK & M_Q5K) mv_rows<FmtQ5K>(L, S, wg, xs, red, ssq); break;
    case LGH_TYPE_Q6_K: if constexpr (MASK & M_Q6K) mv_rows<FmtQ6K>(L, S, wg, xs, red, ssq); break;
    case LGH_TYPE_Q8_0: if constexpr (MASK & M_Q80) mv_rows<FmtQ80>(L, S, wg, xs, red, ssq); break;
    case LGH_TYPE_Q4_0: if constexpr (MASK & M_Q40) mv_rows<FmtQ40>(L, S, wg, xs, red, ssq); break;
    default: break;
  }
  __syncthreads();
  LGH_STAMP(4);
  mv_epilogue(L, S, wg, red, ssq, S.T);
  LGH_STAMP(5);
#ifdef LGH_STAMPS
  if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + 7] = __builtin_amdgcn_s_memtime();
#endif
}

static uint32_t type_mask(int t) {
  switch (t) {
    case LGH_TYPE_Q4_K: return M_Q4K; case LGH_TYPE_Q5_K: return M_Q5K; case LGH_TYPE_Q6_K: return M_Q6K;
    case LGH_TYPE_Q8_0: return M_Q80; case LGH_TYPE_Q4_0: return M_Q40; default: return 0;
  }
}

static uint32_t unit_elems(int t) { return t == LGH_TYPE_Q6_K ? 64u : 32u; }

// waves of the single workgroup a CU runs, by register budget (16 at <=128 VGPRs, 12 at <=168, 8 at <=256)
uint32_t mv_wave_cap(int t) { return t == LGH_TYPE_Q6_K ? 8u : (t == LGH_TYPE_Q5_K ? 12u : (type_mask(t) ? 16u : 0u)); }

// Geometry for one weight shape: ONE workgroup per CU (256 of them), T k-slices x G row-groups = wave_cap
// waves, each workgroup owning a contiguous chunk of launch_rows/256 output rows.  `launch_rows` = output
// rows of the WHOLE launch (all segments); `wave_cap` = the launch's wave budget (min over its segments).
hipError_t mv_plan(int dev_type, uint32_t k, uint32_t n_rows, int npass, MvPlan* plan, uint32_t launch_rows,
                   uint32_t wave_cap) {
  if (!type_mask(dev_type) || k == 0 || n_rows == 0 || npass < 1 || npass > 4) return hipErrorInvalidValue;
  uint32_t ue = unit_elems(dev_type);
  if (k % ue) return hipErrorInvalidValue;
  if (wave_cap == 0 || wave_cap > mv_wave_cap(dev_type)) wave_cap = mv_wave_cap(dev_type);
  uint32_t units = k / ue;
  uint32_t T = (units + 63) / 64;
  if (T > wave_cap) return hipErrorInvalidValue;  // K beyond every supported model
  uint32_t G = wave_cap / T;
  if (launch_rows < n_rows) launch_rows = n_rows;
  uint32_t rpg = (launch_rows + kNumCU * G - 1) / (kNumCU * G);  // rows per wave
  if (rpg < 1) rpg = 1;
  if ((G * rpg) & 1) rpg += 1;  // RoPE epilogues rotate row pairs (2i, 2i+1) inside one workgroup
  plan->units = units;
  plan->T = T;
  plan->G = G;
  plan->rows_per_wg = G * rpg;
  plan->n_wg = (n_rows + plan->rows_per_wg - 1) / plan->rows_per_wg;
  plan->threads = T * G * 64;
  plan->red_floats = (uint32_t)npass * T * plan->rows_per_wg;
  return hipSuccess;
}

int mv_symbol(const MvLaunch& L) {
  uint32_t mask = 0;
  for (int i = 0; i < L.nseg; i++) mask |= type_mask(L.seg[i].type);
  switch (mask) {
    case M_Q4K: return LGH_SYM_MV_Q4K; case M_Q80: return LGH_SYM_MV_Q80; case M_Q40: return LGH_SYM_MV_Q40;
    case M_Q5K: return LGH_SYM_MV_Q5K; case M_Q6K: return LGH_SYM_MV_Q6K;
    case M_Q4K | M_Q6K: return LGH_SYM_MV_Q4K_Q6K; case M_Q5K | M_Q6K: return LGH_SYM_MV_Q5K_Q6K;
    default: return LGH_SYM_MV_ALL;
  }
}

template <uint32_t MASK, int MAXT>
static hipError_t mv_go(const MvLaunch& L, uint32_t n_wg, uint32_t threads, size_t lds, hipStream_t st) {
  static bool attr_set[64] = {};  // > 64 KB of dynamic LDS needs the opt-in once per kernel and device
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mv_kernel<MASK, MAXT>), 160 * 1024, attr_set); e != hipSuccess) return e;
  if (threads > (uint32_t)MAXT) return hipErrorInvalidValue;
  hipLaunchKernelGGL((mv_kernel<MASK, MAXT>), dim3(n_wg), dim3(threads), lds, st, L);
  return hipGetLastError();
}

hipError_t mv_launch(const MvLaunch& L, uint32_t n_wg, uint32_t threads, hipStream_t st) {
  uint32_t mask = 0;
  for (int i = 0; i < L.nseg; i++) mask |= type_mask(L.seg[i].type);
  if (!mask || n_wg == 0 || threads == 0) return hipErrorInvalidValue;
  const size_t lds = ((size_t)L.k + L.red_floats + 16) * sizeof(float);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (mask == M_Q4K) return mv_go<M_Q4K, 1024>(L, n_wg, threads, lds, st);
  if (mask == M_Q80) return mv_go<M_Q80, 1024>(L, n_wg, threads, lds, st);
  if (mask == M_Q40) return mv_go<M_Q40, 1024>(L, n_wg, threads, lds, st);
  if (mask == M_Q5K) return mv_go<M_Q5K, 768>(L, n_wg, threads, lds, st);
  if (mask == M_Q6K) return mv_go<M_Q6K, 512>(L, n_wg, threads, lds, st);
  if (mask == (M_Q4K | M_Q6K)) return mv_go<M_Q4K | M_Q6K, 512>(L, n_wg, threads, lds, st);
  if (mask == (M_Q5K | M_Q6K)) return mv_go<M_Q5K | M_Q6K, 512>(L, n_wg, threads, lds, st);
  return mv_go<M_ALL, 512>(L, n_wg, threads, lds, st);
}

// ------------------------------------------------------------------------------------------ F32
// f32 weights (MoE router, and every GGUF type the engine dequantizes at upload): one wave per row,
// float4 loads, x from L2.  The reference's `vec_mat` sums strictly sequentially (ops.rs:993-999);
// here lanes stride K and meet in a wave reduction.
__global__ void __launch_bounds__(256) f32_matvec_kernel(const float* __restrict__ w, const float* __restrict__ x,
                                                         float* out, uint32_t k, uint32_t n, const float* norm_w, float eps,
                                                         const float* resid) {
  __shared__ float wsum[16];
  float inv = 1.0f;
  if (norm_w) inv = block_inv_rms(x, k, eps, wsum);
  const uint32_t lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const float* wr = w + (size_t)row * k;
  float acc = 0.0f;
  for (uint32_t i = lane * 4; i < k; i += 256) {
    f32x4 wv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wr + i));
    f32x4 xv = *reinterpret_cast<const f32x4*>(x + i);
    if (norm_w) xv = (xv * inv) * *reinterpret_cast<const f32x4*>(norm_w + i);
    acc = __builtin_fmaf(wv.x, xv.x, acc);
    acc = __builtin_fmaf(wv.y, xv.y, acc);
    acc = __builtin_fmaf(wv.z, xv.z, acc);
    acc = __builtin_fmaf(wv.w, xv.w, acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) out[row] = resid ? acc + resid[row] : acc;
}

hipError_t f32_matvec_launch(const float* w, const float* x, float* out, uint32_t k, uint32_t n, const float* norm_w,
                             float eps, const float* resid, hipStream_t st) {
  if (k % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(f32_matvec_kernel, dim3((n + 3) / 4), dim3(256), 0, st, w, x, out, k, n, norm_w, eps, resid);
  return hipGetLastError();
}

}  // namespace lgh
