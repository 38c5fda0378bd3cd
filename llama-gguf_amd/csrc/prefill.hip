// prefill.hip — batched prompt processing: up to 128 prompt tokens per pass as one GEMM per weight on the f16 matrix
// cores (v_mfma_f32_16x16x32_f16), causal attention over the block, everything else row-wise.
//
// The reference has no counterpart: its prefill is `prefill_token` once per prompt token (src/backend/cuda/gpu_only.rs:
// 776-806; CPU: src/model/llama.rs:327-345), i.e. the whole model is streamed from memory once per token.  Here the
// quantized weights are read ONCE per 128 tokens.  This is the one place where f16 rounding enters the engine: the
// activations and the dequantized weights are rounded to f16 (11 significant bits each) in front of the matrix cores,
// products are accumulated in f32.  The KV cache it leaves must equal the one left by the same tokens fed one by one
// within the stated tolerance (tests/test_gpu_prefill.py); the exact token-by-token path stays available
// (LGH_FLAG_EXACT_PREFILL).
//
// Data flow per layer (all launches on the context's stream, nothing is captured in a graph):
//   XH(h * attn_norm) --GEMM wq|wk|wv--> partial sums --pf_qkv_epi (x 1/rms, bias, RoPE)--> q [M][QD] f32, K/V cache rows
//   q, cache --attention (attention.hip, PF variant; causal: token t sees pos0 + t + 1 rows)--> XH(attn)
//   XH(attn) --GEMM wo--> partials --pf_resid (+bias, +residual)--> h, XH(h * ffn_norm), sums of squares per token
//   XH --GEMM gate|up--> partials --pf_swiglu (x 1/rms)--> XH(act) --GEMM down--> partials --pf_resid--> h, XH(h * attn_norm')
// MoE layers (moe.rs:321-413): the f32 decode router runs once per token of the block, the (token, slot) pairs are grouped
// by expert into one row space (prefill.h), and every expert runs ONCE over its rows: one gather launch, a gate|up GEMM per
// expert, one SwiGLU launch, a down GEMM per expert, one combine launch (routing weight * expert output in selection order,
// then the residual).
// The RMSNorm's 1/rms is a per-token scalar and the GEMM is linear: XH holds h * norm_weight and the kernel that adds
// up the GEMM's partial sums multiplies by 1/rms (from the sums of squares the producer of h left) — as in the decode path.
//
// "XH": an activation matrix [128 tokens][K] in f16, laid out for the GEMM's B operand: one 64 KB slab per 256
// elements of K (copied verbatim into LDS by LDS-DMA), inside a slab token t owns 512 B = 32 chunks of 8 elements, chunk
// q stored at position q ^ (t & 15) (so that the 16 tokens of an MFMA operand read 16 different LDS banks groups), inside
// a chunk the elements are in the order 0,2,1,3,4,6,5,7 (the order in which two masks pull nibbles out of a tile16 word).
//
// GEMM: a workgroup = 4 waves = 256 weight rows x one k-range x all 128 tokens; a wave = 4 row tiles (64 rows): weights
// go HBM -> registers -> f16 (never through LDS: the dequantized lane layout IS the A operand), an activation fragment
// read from LDS feeds 4 MFMAs (LDS at 50 % of its bandwidth), 128 f32 accumulators per lane.  K is split over
// gridDim.y workgroups to fill the chip; the f32 partial sums [split][token][column] are added up by the row-wise
// kernel that consumes them (together with bias, RoPE, residual, RMSNorm, SwiGLU).
// Weights are dequantized as w * 2^8 (kPfScale) to keep small block scales out of the f16 subnormals; |w| < 256 is
// assumed (f16 overflow otherwise) — GGUF weights are O(1).
#include "device_utils.h"
#include "mv_epilogue.h"
#include "prefill.h"

namespace lgh {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

// formats as in matvec_mfma.hip (same tile16 layouts)
enum : int { PF_Q4K = 0, PF_Q6K = 1, PF_Q5K = 2, PF_Q80 = 3, PF_Q40 = 4 };
__host__ __device__ constexpr uint32_t pf_tile_bytes(int f) {
  return f == PF_Q4K ? 2304u : f == PF_Q6K ? 3392u : f == PF_Q5K ? 2816u : f == PF_Q80 ? 4352u : 2304u;
}
static int pf_fmt_of(int dev_type) {
  return dev_type == kDevQ4K_T16 ? PF_Q4K : dev_type == kDevQ6K_T16 ? PF_Q6K : dev_type == kDevQ5K_T16 ? PF_Q5K
         : dev_type == kDevQ80_T16 ? PF_Q80 : dev_type == kDevQ40_T16 ? PF_Q40 : -1;
}

// waves per workgroup: 4 (one per SIMD, 4 row tiles each).  8 waves of 2 tiles (-DPF_WAVES=8: two per SIMD, twice the LDS
// reads per MFMA) measured the same 4.68 ms per 128-token Llama-3-8B prompt: the loop is not latency-bound.
#ifndef PF_EXP
#define PF_EXP 0
#endif
#ifndef PF_WAVES
#define PF_WAVES 4
#endif
#ifndef PF_SHAPE32
#define PF_SHAPE32 0   /* experiment build (-DPF_SHAPE32=1): v_mfma_f32_32x32x16_f16 (pf_body32); product: v_mfma_f32_16x16x32_f16 (pf_body) */
#endif
constexpr int kPfWaves = PF_WAVES;
constexpr int kPfRT = 16 / kPfWaves;     // row tiles per wave (a workgroup covers 256 rows)
constexpr int kPfMT = kPfTokens / 16;    // token tiles of a full block
constexpr float kPfScale = 256.0f;

#ifdef LGH_STAMPS
// diagnostic build: per-workgroup phase stamps (100 MHz s_memrealtime) of the GEMM launches whose output is LGH_PF_STAMP_COLS wide
#ifndef LGH_PF_STAMP_COLS
#define LGH_PF_STAMP_COLS 28672
#endif
__device__ unsigned long long g_pf_stamps[1024 * 16];
#define LGH_PF_STAMP(i)                                                                                         \
  do {                                                                                                          \
    if (G.ncols == LGH_PF_STAMP_COLS && threadIdx.x == 0 && (i) < 16) {                                         \
      const uint32_t wgid = blockIdx.y * gridDim.x + blockIdx.x;                                                \
      if (wgid < 1024) g_pf_stamps[wgid * 16 + (i)] = __builtin_amdgcn_s_memrealtime();                         \
    }                                                                                                           \
  } while (0)
hipError_t pf_read_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_pf_stamps), std::min(n, (size_t)1024 * 16) * 8);
}
#else
#define LGH_PF_STAMP(i)
#endif

struct PfRaw { u32x4 hd; u32x4 q[4]; };

template <int F>
__device__ __forceinline__ void pf_load(PfRaw& r, const uint8_t* tile, uint32_t lane, uint32_t n) {
  if (F == PF_Q6K) {
    r.hd = ldg_nt128(tile + 3072 + n * 16);
#pragma unroll
    for (int i = 0; i < 3; i++) r.q[i] = ldg_nt128(tile + i * 1024 + lane * 16);
    r.q[3].x = ldg_nt32(tile + 3328 + (n >> 1) * 4);
  } else if (F == PF_Q80) {
    r.hd = ldg_nt128(tile + 4096 + n * 16);
#pragma unroll
    for (int i = 0; i < 4; i++) r.q[i] = ldg_nt128(tile + i * 1024 + lane * 16);
  } else {
    r.hd = ldg_nt128(tile + 2048 + n * 16);
    r.q[0] = ldg_nt128(tile + lane * 16);
    r.q[1] = ldg_nt128(tile + 1024 + lane * 16);
    if (F == PF_Q5K) {
      const u32x2 h = ldg_nt64(tile + 2304 + lane * 8);
      r.q[2].x = h.x;
      r.q[2].y = h.y;
    }
  }
}

__device__ __forceinline__ h16x2 pf_bcast(float v) {
  const _Float16 h = (_Float16)v;
  h16x2 r = {h, h};
  return r;
}

// scale S and offset O (both times kPfScale, as f16 pairs) of MFMA step pp of the lane's row n and k-chunk c:
// weight = S * u + O with u the unsigned stored quant
template <int F>
__device__ __forceinline__ void pf_scale(const PfRaw& r, int pp, uint32_t n, uint32_t c, h16x2& S, h16x2& O) {
  if (F == PF_Q6K) {   // y = d * sc * (q' - 32), one int8 scale per 16 elements (dequant.rs:321-356)
    const uint32_t hdw = pp == 0 ? r.hd.x : pp == 1 ? r.hd.y : pp == 2 ? r.hd.z : r.hd.w;
    const float scf = (float)(int)__builtin_amdgcn_sbfe((int)hdw, c * 8, 8);
    const uint32_t dh = (n & 1) ? r.q[3].x >> 16 : r.q[3].x & 0xFFFFu;
    S = pf_bcast(h2f(dh) * scf * kPfScale);
    O = S * pf_bcast(-32.0f);
  } else if (F == PF_Q80 || F == PF_Q40) {   // y = d * q (Q8_0, stored q + 128 after the sign flip) / d * (q - 8) (Q4_0)
    const uint32_t hdw = pp == 0 ? r.hd.x : pp == 1 ? r.hd.y : pp == 2 ? r.hd.z : r.hd.w;
    const float dd = h2f((c >> 1) ? hdw >> 16 : hdw & 0xFFFFu);
    S = pf_bcast(dd * kPfScale);
    O = S * pf_bcast(F == PF_Q80 ? -128.0f : -8.0f);
  } else {   // Q4_K / Q5_K: y = d * sc * q - dmin * m, 6-bit (sc, m) per 32 elements (dequant.rs:210-255)
    const uint32_t s8 = (c >> 1) * 8;
    const uint32_t a = (r.hd.y >> s8) & 0x00FF00FFu, bq = (r.hd.z >> s8) & 0x00FF00FFu, cq = (r.hd.w >> s8) & 0x00FF00FFu;
    uint32_t sc, mn;
    if (pp < 2) {
      const uint32_t sc01 = a & 0x003F003Fu, mn01 = bq & 0x003F003Fu;
      sc = pp == 0 ? sc01 & 0xFFu : sc01 >> 16;
      mn = pp == 0 ? mn01 & 0xFFu : mn01 >> 16;
    } else {
      const uint32_t sc23 = (cq & 0x000F000Fu) | ((a >> 2) & 0x00300030u);
      const uint32_t mn23 = ((cq >> 4) & 0x000F000Fu) | ((bq >> 2) & 0x00300030u);
      sc = pp == 2 ? sc23 & 0xFFu : sc23 >> 16;
      mn = pp == 2 ? mn23 & 0xFFu : mn23 >> 16;
    }
    const float dd = h2f(r.hd.x & 0xFFFFu), dmin = h2f(r.hd.x >> 16);
    S = pf_bcast(dd * (float)sc * kPfScale);
    O = pf_bcast(-(dmin * (float)mn) * kPfScale);
  }
}

// four words of two f16 (0x6400 | u == 1024 + u exactly) -> S * u + O.  Stage by stage over the four words (not word by
// word): a packed-f16 op that consumes the previous instruction's result costs an extra s_nop on gfx950.
__device__ __forceinline__ h16x8 pf_fin4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3, h16x2 S, h16x2 O) {
  const h16x2 k = {(_Float16)1024.0f, (_Float16)1024.0f};
  const h16x2 u0 = __builtin_bit_cast(h16x2, b0) - k, u1 = __builtin_bit_cast(h16x2, b1) - k;
  const h16x2 u2 = __builtin_bit_cast(h16x2, b2) - k, u3 = __builtin_bit_cast(h16x2, b3) - k;
  const h16x2 w0 = __builtin_elementwise_fma(u0, S, O), w1 = __builtin_elementwise_fma(u1, S, O);
  const h16x2 w2 = __builtin_elementwise_fma(u2, S, O), w3 = __builtin_elementwise_fma(u3, S, O);
  h16x8 r = {w0.x, w0.y, w1.x, w1.y, w2.x, w2.y, w3.x, w3.y};
  return r;
}

// eight nibbles (word byte t = w[t] | w[t+4] << 4) -> f16 in the order w0 w2 w1 w3 w4 w6 w5 w7
__device__ __forceinline__ h16x8 pf_frag_nib(uint32_t N, h16x2 S, h16x2 O) {
  const uint32_t m = opaque(0x000F000Fu), e = 0x64006400u;   // mask in a VGPR: (x & m) | e is then one v_and_or_b32
  return pf_fin4((N & m) | e, ((N >> 8) & m) | e, ((N >> 4) & m) | e, ((N >> 12) & m) | e, S, O);
}
// eight bytes (B0 = w0..w3, B1 = w4..w7) -> f16 in the same order
__device__ __forceinline__ h16x8 pf_frag_bytes(uint32_t B0, uint32_t B1, h16x2 S, h16x2 O) {
  const uint32_t e = 0x64646464u;
  return pf_fin4(__builtin_amdgcn_perm(e, B0, 0x04020400u), __builtin_amdgcn_perm(e, B0, 0x04030401u),
                 __builtin_amdgcn_perm(e, B1, 0x04020400u), __builtin_amdgcn_perm(e, B1, 0x04030401u), S, O);
}

// the A operand of MFMA (pp, h): elements 64pp + 16c + 8h .. +7 of row n, dequantized
template <int F>
__device__ __forceinline__ h16x8 pf_frag(const PfRaw& r, int pp, int h, h16x2 S, h16x2 O) {
  if (F == PF_Q80) {
    const u32x4 b = r.q[pp];
    const uint32_t B0 = (h ? b.z : b.x) ^ 0x80808080u, B1 = (h ? b.w : b.y) ^ 0x80808080u;
    return pf_frag_bytes(B0, B1, S, O);
  }
  const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
  const uint32_t N = h ? N1 : N0;
  if (F == PF_Q4K || F == PF_Q40) return pf_frag_nib(N, S, O);
  uint32_t B0 = N & 0x0F0F0F0Fu, B1 = (N >> 4) & 0x0F0F0F0Fu;
  if (F == PF_Q6K) {   // H byte t = f[t] | f[t+4] << 2 | f[t+8] << 4 | f[t+12] << 6 (the two high bits of the chunk's 16 weights)
    const uint32_t H = pp == 0 ? r.q[2].x : pp == 1 ? r.q[2].y : pp == 2 ? r.q[2].z : r.q[2].w;
    B0 |= ((H >> (4 * h)) & 0x03030303u) << 4;
    B1 |= ((H >> (4 * h + 2)) & 0x03030303u) << 4;
  } else {             // Q5_K: fifth bits, dword of the step pair, low nibbles = even step, high nibbles = odd step
    const uint32_t H = ((pp >> 1) ? r.q[2].y : r.q[2].x) >> (4 * (pp & 1));
    B0 |= ((H >> (2 * h)) & 0x01010101u) << 4;
    B1 |= ((H >> (2 * h + 1)) & 0x01010101u) << 4;
  }
  return pf_frag_bytes(B0, B1, S, O);
}

// MT = token tiles computed: 8 for a full block, 4 / 2 when at most 64 / 32 rows are real (short prompts, and the rows routed
// to one MoE expert — about 32 of 128 with 2-of-8 routing): the MFMAs, the LDS reads and the activation DMA shrink with it.
template <int F, int MT>
__device__ __forceinline__ void pf_body(const PfGemm& G, const PfSeg& sg, uint32_t rg, uint32_t ks, uint32_t m_tiles, uint8_t* smem) {
  constexpr uint32_t kDma = (uint32_t)MT * 16 * 512 / kPfWaves;   // a wave's share of the MT * 16 rows of a slab that are copied
  const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n = lane & 15, c = lane >> 4;
  constexpr uint32_t tb = pf_tile_bytes(F);
  // this split's k-blocks
  const uint32_t per = (G.nblk + G.S - 1) / G.S;
  const uint32_t b0 = ks * per, b1 = b0 + per < G.nblk ? b0 + per : G.nblk;
  const uint32_t tile0 = (rg * kPfWaves + wave) * kPfRT;
  const uint8_t* wt[kPfRT];
#pragma unroll
  for (int r = 0; r < kPfRT; r++) {
    const uint32_t tl = tile0 + r < sg.ntiles ? tile0 + r : sg.ntiles - 1;   // clamped: loads are unconditional
    wt[r] = sg.w + (size_t)tl * G.nblk * tb;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

  f32x4 acc[kPfRT][MT];
#pragma unroll
  for (int r = 0; r < kPfRT; r++)
#pragma unroll
    for (int t = 0; t < MT; t++) acc[r][t] = (f32x4)(0.0f);

  // one 64 KB slab -> LDS buffer `buf`: this wave's quarter, 16 x 1 KB by LDS-DMA (no registers)
  auto x_dma = [&](uint32_t b, uint32_t buf) {
    const uint8_t* src = G.xh + (size_t)b * kPfSlabBytes + wave * kDma + lane * 16;
    const uint32_t dst = lds_base + buf * kPfSlabBytes + wave * kDma;
#pragma unroll
    for (int i = 0; i < (int)(kDma / 1024); i++) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + i * 1024), "s"(dst + i * 1024) : "memory");
    }
  };

  LGH_PF_STAMP(0);
  PfRaw nxt[kPfRT];
  if (b0 < b1) {
    x_dma(b0, 0);
#pragma unroll
    for (int r = 0; r < kPfRT; r++) pf_load<F>(nxt[r], wt[r] + (size_t)b0 * tb, lane, n);
  }
  // (the BUILTIN, not inline asm: hipcc's wait-count pass must see that nothing is outstanding past this point.  With an
  // asm wait it still counted the previous block's weight loads as pending and protected their first uses with
  // `s_waitcnt vmcnt(21..12)` — in the hardware's count those are the NEXT slab's 16 LDS-DMA loads, issued just before, so
  // every block began by waiting for part of the next slab.  They come from L2: 4.86 -> 4.80 ms per prompt pass.)
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
  __syncthreads();
  LGH_PF_STAMP(1);
  for (uint32_t b = b0; b < b1; b++) {
    const uint32_t cur = (b - b0) & 1;
    PfRaw w[kPfRT];
#pragma unroll
    for (int r = 0; r < kPfRT; r++) w[r] = nxt[r];
#if PF_EXP != 4 && PF_EXP != 6   /* experiment 4 / 6: no activation slabs after the first */
    if (b + 1 < b1) x_dma(b + 1, cur ^ 1);
#endif
#if PF_EXP != 5 && PF_EXP != 6   /* experiment 5 / 6: no weight loads after the first block's */
    {
      const uint32_t bn = b + 1 < b1 ? b + 1 : b;   // the last iteration re-requests its own block (unconditional loads)
#pragma unroll
      for (int r = 0; r < kPfRT; r++) pf_load<F>(nxt[r], wt[r] + (size_t)bn * tb, lane, n);
    }
#endif
    // The block's 32 units (MFMA step pp, half h, row tile r) as a software pipeline, one scheduling region per unit:
    // the unit's 8 MFMAs, the dequantization of the NEXT unit's A fragment, and — in the first unit of a (pp, h) phase —
    // the LDS reads of the next phase's activation fragments.  Measured on Llama-3-8B Q4_K_M (128 tokens, 4.9 ms): with
    // the LDS reads removed 4.89 ms, with the dequantization removed 4.19 ms; forcing an MFMA / VALU interleave with
    // sched_group_barrier changed nothing (4.94 vs 4.88 ms).  Phase stamps of the gate|up GEMM (tools/pf_phases.py, the PF_EXP
    // switches below; 224 workgroups, 256 MFMAs + ~720 VALU ops per wave and block): 4.3 us per block as written, 3.3 us with
    // MFMAs only (no dequantization, no LDS reads: 1.3 PFLOP/s chip-wide, what f16 MFMAs sustain here with every CU issuing
    // them; PF_EXP 4 / 5 / 6 — no activation slabs / no weight loads / neither after the first block — 4.35 / 4.19 / 4.12 ms per pass
    // against 4.63: the loads are not what bounds the loop), 3.7 us with the dequantization and LDS reads only — the two overlap to within 30 %, the wait + barrier at the end
    // of a block is 0.2 us.  The loop runs at 77 % of its MFMA-only rate.
    const uint8_t* xb = smem + cur * kPfSlabBytes + n * 512;
    h16x8 bf[2][MT];
    h16x2 S[kPfRT], O[kPfRT];
    auto read_bf = [&](int ph, h16x8* dst) {
      const uint32_t q = (uint32_t)((ph >> 1) * 8 + (ph & 1)) + c * 2;
#pragma unroll
      for (int t = 0; t < MT; t++) dst[t] = *reinterpret_cast<const h16x8*>(xb + t * 8192 + ((q ^ n) << 4));
    };
    read_bf(0, bf[0]);
    pf_scale<F>(w[0], 0, n, c, S[0], O[0]);
    h16x8 af[2];
    af[0] = pf_frag<F>(w[0], 0, 0, S[0], O[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8 * kPfRT; u++) {
      const int ph = u / kPfRT, r = u % kPfRT;
      if (u + 1 < 8 * kPfRT) {
        const int ph2 = (u + 1) / kPfRT, r2 = (u + 1) % kPfRT, pp2 = ph2 >> 1, h2 = ph2 & 1;
#if PF_EXP == 1 || PF_EXP == 3   /* experiment: no dequantization */
        af[(u + 1) & 1] = __builtin_bit_cast(h16x8, w[r2].q[pp2 & 1]);
#else
        if (h2 == 0) pf_scale<F>(w[r2], pp2, n, c, S[r2], O[r2]);
        af[(u + 1) & 1] = pf_frag<F>(w[r2], pp2, h2, S[r2], O[r2]);
#endif
      }
#if PF_EXP == 2   /* experiment: no MFMAs (the fragment is folded into one accumulator so that it stays live) */
      acc[r][0] += __builtin_bit_cast(f32x4, af[u & 1]) * __builtin_bit_cast(f32x4, bf[ph & 1][u & 7]);
#else
#pragma unroll
      for (int t = 0; t < MT; t++) acc[r][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u & 1], bf[ph & 1][t], acc[r][t], 0, 0, 0);
#endif
      const bool rd = r == 0 && ph + 1 < 8;
#if PF_EXP == 3   /* experiment: no LDS reads either */
      if (rd) { for (int t = 0; t < MT; t++) bf[(ph + 1) & 1][t] = bf[ph & 1][t] + af[u & 1]; }
#else
      if (rd) read_bf(ph + 1, bf[(ph + 1) & 1]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    LGH_PF_STAMP(2 + 2 * (b - b0 < 5 ? b - b0 : 5));
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    LGH_PF_STAMP(3 + 2 * (b - b0 < 5 ? b - b0 : 5));
  }
  // partial sums: lane holds, per (row tile, token tile), rows 4c .. 4c+3 of token n
  uint32_t row0 = 0;
  if (G.row_base) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(row0) : "s"(G.row_base) : "memory");
  float* part = G.part + ((size_t)ks * G.part_rows + row0) * G.ncols;
#pragma unroll
  for (int r = 0; r < kPfRT; r++) {
    if (tile0 + r >= sg.ntiles) continue;
    const uint32_t col = sg.col0 + (tile0 + r) * 16 + c * 4;
#pragma unroll
    for (int t = 0; t < MT; t++) {
      if ((uint32_t)t >= m_tiles) continue;
      const f32x4 v = acc[r][t] * (1.0f / kPfScale);
      *reinterpret_cast<f32x4*>(part + (size_t)(t * 16 + n) * G.ncols + col) = v;
    }
  }
#ifdef LGH_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  LGH_PF_STAMP(14);
}

// EXPERIMENT (not in the product build; `make variant NAME=pf32 VSRC=prefill DEFS=-DPF_SHAPE32=1`), measured and not faster: Llama-3-8B
// Q4_K_M 4.90 ms per 128-token pass against 4.82 on the same box, Mixtral 15.19 against 15.24; test_gpu_prefill.py passes on it; 290
// registers per wave instead of 424.
// The same GEMM on v_mfma_f32_32x32x16_f16: an MFMA holds the SIMD's vector issue port for 8 cycles whatever its shape
// (MI355X_MICROARCH.md, cycle constants), and this loop is bound by that port — 256 MFMAs x 8 + ~700 vector instructions x 4
// against 4096 cycles of matrix pipe per block; the 32x32x16 form does the same products with half the MFMA instructions.
// No other layout changes: a wave's four 16-row tiles become two 32-row tiles (tile pair 2v, 2v + 1); lane l = 32g + 16sub + n
// loads row n of tile 2v + sub and, of that row, the two 16-element chunks 2g and 2g + 1 of every 64 (the pieces lanes 16(2g + j) + n
// of the 16-row scheme hold — the tile16 layout addresses them directly); MFMA step (pp, j, h) contracts the 16 elements
// 64pp + 16(2g + j) + 8h .. + 7, g = 0, 1, whose activation chunks token l & 31 reads from the same XH slab.  Both chunks of a lane
// lie in one 32-element sub-block (Q4_K / Q5_K / Q8_0 / Q4_0: one scale pair per (row tile, pp); Q6_K: one per chunk).
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int F, int MT2>
__device__ __forceinline__ void pf_body32(const PfGemm& G, const PfSeg& sg, uint32_t rg, uint32_t ks, uint32_t m_tiles, uint8_t* smem) {
  static_assert(kPfRT % 2 == 0, "pairs of 16-row tiles");
  constexpr int kVT = kPfRT / 2;                                       // 32-row tiles per wave
  constexpr uint32_t kDma = (uint32_t)MT2 * 32 * 512 / kPfWaves;      // a wave's share of the MT2 * 32 rows of a slab that are copied
  const uint32_t lane = threadIdx.x & 63, wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n = lane & 15, sub = (lane >> 4) & 1, g = lane >> 5, tk = lane & 31;
  constexpr uint32_t tb = pf_tile_bytes(F);
  const uint32_t per = (G.nblk + G.S - 1) / G.S;
  const uint32_t b0 = ks * per, b1 = b0 + per < G.nblk ? b0 + per : G.nblk;
  const uint32_t tile0 = (rg * kPfWaves + wave) * kPfRT;
  const uint8_t* wt[kVT];
#pragma unroll
  for (int v = 0; v < kVT; v++) {
    const uint32_t tl = tile0 + 2 * v + sub < sg.ntiles ? tile0 + 2 * v + sub : sg.ntiles - 1;   // clamped: loads are unconditional
    wt[v] = sg.w + (size_t)tl * G.nblk * tb;
  }
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;

  f32x16 acc[kVT][MT2];
#pragma unroll
  for (int v = 0; v < kVT; v++)
#pragma unroll
    for (int t = 0; t < MT2; t++) acc[v][t] = (f32x16)(0.0f);

  auto x_dma = [&](uint32_t b, uint32_t buf) {
    const uint8_t* src = G.xh + (size_t)b * kPfSlabBytes + wave * kDma + lane * 16;
    const uint32_t dst = lds_base + buf * kPfSlabBytes + wave * kDma;
#pragma unroll
    for (int i = 0; i < (int)(kDma / 1024); i++) {
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + i * 1024), "s"(dst + i * 1024) : "memory");
    }
  };
  auto w_load = [&](PfRaw (&r)[kVT][2], uint32_t b) {
#pragma unroll
    for (int v = 0; v < kVT; v++)
#pragma unroll
      for (int j = 0; j < 2; j++) pf_load<F>(r[v][j], wt[v] + (size_t)b * tb, 16 * (2 * g + j) + n, n);
  };

  LGH_PF_STAMP(0);
  PfRaw nxt[kVT][2];
  if (b0 < b1) {
    x_dma(b0, 0);
    w_load(nxt, b0);
  }
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0) (the builtin: see pf_body)
  __syncthreads();
  LGH_PF_STAMP(1);
  for (uint32_t b = b0; b < b1; b++) {
    const uint32_t cur = (b - b0) & 1;
    PfRaw w[kVT][2];
#pragma unroll
    for (int v = 0; v < kVT; v++)
#pragma unroll
      for (int j = 0; j < 2; j++) w[v][j] = nxt[v][j];
    if (b + 1 < b1) x_dma(b + 1, cur ^ 1);
    w_load(nxt, b + 1 < b1 ? b + 1 : b);   // the last iteration re-requests its own block (unconditional loads)
    // 16 steps (pp, j, h) x kVT tiles = 16 kVT units, one scheduling region each: the unit's MT2 MFMAs, the dequantization of the
    // next unit's A fragment and — in a step's first unit — the LDS reads of the next step's activation fragments
    const uint8_t* xb = smem + cur * kPfSlabBytes + tk * 512;
    const uint32_t swz = tk & 15;
    h16x8 bf[2][MT2];
    h16x2 S[kVT], O[kVT];
    auto read_bf = [&](int st, h16x8* dst) {
      const uint32_t q = (uint32_t)(8 * (st >> 2) + 2 * ((st >> 1) & 1) + (st & 1)) + 4 * g;
#pragma unroll
      for (int t = 0; t < MT2; t++) dst[t] = *reinterpret_cast<const h16x8*>(xb + t * 16384 + ((q ^ swz) << 4));
    };
    read_bf(0, bf[0]);
    pf_scale<F>(w[0][0], 0, n, 2 * g, S[0], O[0]);
    h16x8 af[2];
    af[0] = pf_frag<F>(w[0][0], 0, 0, S[0], O[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 16 * kVT; u++) {
      const int st = u / kVT, v = u % kVT;
      if (u + 1 < 16 * kVT) {
        const int st2 = (u + 1) / kVT, v2 = (u + 1) % kVT, pp2 = st2 >> 2, j2 = (st2 >> 1) & 1, h2 = st2 & 1;
        if (h2 == 0 && (F == PF_Q6K || j2 == 0)) pf_scale<F>(w[v2][j2], pp2, n, 2 * g + j2, S[v2], O[v2]);
        af[(u + 1) & 1] = pf_frag<F>(w[v2][j2], pp2, h2, S[v2], O[v2]);
      }
#pragma unroll
      for (int t = 0; t < MT2; t++) acc[v][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u & 1], bf[st & 1][t], acc[v][t], 0, 0, 0);
      if (v == 0 && st + 1 < 16) read_bf(st + 1, bf[(st + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    LGH_PF_STAMP(2 + 2 * (b - b0 < 5 ? b - b0 : 5));
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __syncthreads();
    LGH_PF_STAMP(3 + 2 * (b - b0 < 5 ? b - b0 : 5));
  }
  // partial sums: lane holds, per (32-row tile, 32-token tile), rows 8i + 4g .. + 3 (i = 0..3) of token l & 31
  uint32_t row0 = 0;
  if (G.row_base) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(row0) : "s"(G.row_base) : "memory");
  float* part = G.part + ((size_t)ks * G.part_rows + row0) * G.ncols;
#pragma unroll
  for (int v = 0; v < kVT; v++) {
#pragma unroll
    for (int t = 0; t < MT2; t++) {
      const uint32_t tok = (uint32_t)t * 32 + tk;
      if (tok >= m_tiles * 16) continue;   // (exactly the rows the 16-token form writes)
#pragma unroll
      for (int i = 0; i < 4; i++) {
        if (tile0 + 2 * v + (i >> 1) >= sg.ntiles) continue;
        const uint32_t col = sg.col0 + (tile0 + 2 * v) * 16 + 8 * i + 4 * g;
        const f32x16 a = acc[v][t];
        const f32x4 o = {a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]};
        *reinterpret_cast<f32x4*>(part + (size_t)tok * G.ncols + col) = o * (1.0f / kPfScale);
      }
    }
  }
#ifdef LGH_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  LGH_PF_STAMP(14);
}

template <uint32_t MASK>
__global__ void __launch_bounds__(kPfWaves * 64) pf_gemm_kernel(const PfGemm G) {
  extern __shared__ __attribute__((aligned(16))) uint8_t pf_smem[];
  const uint32_t rgid = blockIdx.x;
  int si = 0;
  if (G.nseg > 1 && rgid >= G.seg[1].rg_begin) si = 1;
  if (G.nseg > 2 && rgid >= G.seg[2].rg_begin) si = 2;
  const PfSeg& sg = G.seg[si];
  const uint32_t rg = rgid - sg.rg_begin;
  auto is = [&](int f) { return (MASK & (1u << f)) && (MASK == (1u << f) || sg.fmt == f); };
  uint32_t m_tiles = G.m_tiles;
  if (G.m_count) {   // MoE: this expert's row count lives on the device (no host round trip per layer)
    uint32_t cnt;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(cnt) : "s"(G.m_count) : "memory");
    if (cnt == 0) return;
    m_tiles = (cnt + 15) / 16;
  }
#if PF_SHAPE32 && PF_EXP == 0
#define LGH_PF_RUN(F)                                                                    \
  do {                                                                                   \
    if (m_tiles <= 2) pf_body32<F, 1>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);          \
    else if (m_tiles <= 4) pf_body32<F, 2>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);     \
    else pf_body32<F, kPfMT / 2>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);               \
  } while (0)
#else
#define LGH_PF_RUN(F)                                                                    \
  do {                                                                                   \
    if (m_tiles <= 2) pf_body<F, 2>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);            \
    else if (m_tiles <= 4) pf_body<F, 4>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);       \
    else pf_body<F, kPfMT>(G, sg, rg, blockIdx.y, m_tiles, pf_smem);                     \
  } while (0)
#endif
  if (is(PF_Q4K)) LGH_PF_RUN(PF_Q4K);
  else if (is(PF_Q6K)) LGH_PF_RUN(PF_Q6K);
  else if (is(PF_Q5K)) LGH_PF_RUN(PF_Q5K);
  else if (is(PF_Q80)) LGH_PF_RUN(PF_Q80);
  else if (is(PF_Q40)) LGH_PF_RUN(PF_Q40);
#undef LGH_PF_RUN
}

template <uint32_t MASK>
static hipError_t pf_gemm_go(const PfGemm& G, uint32_t n_rg, hipStream_t st) {
  static bool attr_set[64] = {};
  constexpr size_t lds = 2 * kPfSlabBytes;
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&pf_gemm_kernel<MASK>), (int)lds, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL((pf_gemm_kernel<MASK>), dim3(n_rg, G.S), dim3(kPfWaves * 64), lds, st, G);
  return hipGetLastError();
}

bool pf_supported_type(int dev_type) { return pf_fmt_of(dev_type) >= 0; }

// k-splits of a GEMM over matrices with n_rows[i] rows and k columns: as many as keep the launch within one workgroup per CU
static void pf_plan(const uint32_t* n_rows, int nw, uint32_t k, uint32_t* rg_out, uint32_t* S_out) {
  const uint32_t nblk = k / 256;
  uint32_t rg = 0;
  for (int i = 0; i < nw; i++) rg += (n_rows[i] / 16 + kPfWaves * kPfRT - 1) / (kPfWaves * kPfRT);
  uint32_t S = rg >= (uint32_t)kNumCU ? 1 : kNumCU / rg;
  if (S > nblk) S = nblk;
  const uint32_t per = (nblk + S - 1) / S;
  *S_out = (nblk + per - 1) / per;
  *rg_out = rg;
}

// bytes of partial sums such a GEMM writes
size_t pf_part_bytes(const uint32_t* n_rows, int nw, uint32_t k, uint32_t part_rows) {
  uint32_t rg, S, cols = 0;
  pf_plan(n_rows, nw, k, &rg, &S);
  for (int i = 0; i < nw; i++) cols += n_rows[i];
  return (size_t)S * part_rows * cols * 4;
}

// Plans and launches one GEMM: up to 3 weight matrices that share the input XH (k elements per token), outputs side by
// side in the partial-sum buffer.  Returns the split count through *S_out (the consumer adds that many partials).
hipError_t pf_gemm_launch(const DevWeight* const* W, int nw, const uint8_t* xh, float* part, size_t part_bytes, uint32_t m_tokens,
                          uint32_t* S_out, uint32_t* ncols_out, hipStream_t st, uint32_t expert, const int* m_count, uint32_t part_rows,
                          const int* row_base) {
  if (nw < 1 || nw > 3 || m_tokens == 0 || m_tokens > (uint32_t)kPfTokens) return hipErrorInvalidValue;
  PfGemm G{};
  const uint32_t k = W[0]->k;
  if (k % 256) return hipErrorInvalidValue;
  G.nseg = nw;
  G.nblk = k / 256;
  uint32_t rg = 0, col = 0, mask = 0, n_rows[3] = {0, 0, 0};
  for (int i = 0; i < nw; i++) {
    const int f = pf_fmt_of(W[i]->type);
    if (f < 0 || W[i]->k != k || W[i]->n % 16 || expert >= W[i]->n_stack) return hipErrorInvalidValue;
    G.seg[i].w = W[i]->plane[0] + (uint64_t)expert * W[i]->stack_stride[0];   // expert stacks: [expert][tiles]
    G.seg[i].ntiles = W[i]->n / 16;
    G.seg[i].col0 = col;
    G.seg[i].rg_begin = rg;
    G.seg[i].fmt = f;
    mask |= 1u << f;
    rg += (G.seg[i].ntiles + kPfWaves * kPfRT - 1) / (kPfWaves * kPfRT);
    col += W[i]->n;
    n_rows[i] = W[i]->n;
  }
  uint32_t rg2;
  pf_plan(n_rows, nw, k, &rg2, &G.S);
  G.ncols = col;
  G.xh = xh;
  G.part = part;
  G.m_tiles = (m_tokens + 15) / 16;
  G.m_count = m_count;
  G.part_rows = part_rows;
  G.row_base = row_base;
  if (part_rows < (uint32_t)kPfTokens || (size_t)G.S * part_rows * col * 4 > part_bytes) return hipErrorInvalidValue;
  *S_out = G.S;
  *ncols_out = col;
  switch (mask) {
    case 1u << PF_Q4K: return pf_gemm_go<1u << PF_Q4K>(G, rg, st);
    case 1u << PF_Q6K: return pf_gemm_go<1u << PF_Q6K>(G, rg, st);
    case (1u << PF_Q4K) | (1u << PF_Q6K): return pf_gemm_go<(1u << PF_Q4K) | (1u << PF_Q6K)>(G, rg, st);
    case 1u << PF_Q5K: return pf_gemm_go<1u << PF_Q5K>(G, rg, st);
    case (1u << PF_Q5K) | (1u << PF_Q6K): return pf_gemm_go<(1u << PF_Q5K) | (1u << PF_Q6K)>(G, rg, st);
    case 1u << PF_Q80: return pf_gemm_go<1u << PF_Q80>(G, rg, st);
    case 1u << PF_Q40: return pf_gemm_go<1u << PF_Q40>(G, rg, st);
    default: return pf_gemm_go<31u>(G, rg, st);   // any other mix of the five formats
  }
}

// ------------------------------------------------------------------------------------------------
// XH writers and the row-wise kernels between the GEMMs
// ------------------------------------------------------------------------------------------------
// chunk `ch` (8 consecutive elements starting at 8*ch) of token t
__device__ __forceinline__ void xh_store_chunk(uint8_t* xh, uint32_t t, uint32_t ch, const float v[8]) {
  const uint32_t b = ch >> 5, q = ch & 31;
  h16x8 o = {(_Float16)v[0], (_Float16)v[2], (_Float16)v[1], (_Float16)v[3], (_Float16)v[4], (_Float16)v[6], (_Float16)v[5], (_Float16)v[7]};
  *reinterpret_cast<h16x8*>(xh + (size_t)b * kPfSlabBytes + t * 512 + ((q ^ (t & 15)) << 4)) = o;
}

// 1 / rms of token t from the per-chunk sums of squares its producer left (simd.rs:847-878); 1 when there is no norm
__device__ __forceinline__ float pf_inv_rms(const float* ssq, uint32_t n_ssq, uint32_t t, uint32_t H, float eps) {
  if (!ssq) return 1.0f;
  float tot = 0.0f;
  for (uint32_t j = 0; j < n_ssq; j++) tot += ssq[t * kPfSsqChunks + j];
  return 1.0f / __builtin_sqrtf(tot / (float)H + eps);
}

// Row epilogue, (2048-column chunks) x (tokens) workgroups:
//   v[i] = sum_s part[s][t][col0 + i] (+ bias[i]) + hidden[t][i]  (the residual, layers.rs:1201-1208, 1235-1241);
//   hidden[t][i] = v[i];  XH[t][i] = f16(v[i] * nw[i]);  ssq[t][chunk] = sum of v^2 over the chunk.
// The RMSNorm's 1/rms is a per-token scalar and the GEMM is linear, so it is applied by the kernel that consumes the
// GEMM's partial sums (pf_inv_rms) — no second pass over the row.  S == 0: v = hidden[t][i] as it is (embedding rows).
__global__ void __launch_bounds__(256) pf_resid_kernel(const float* __restrict__ part, uint32_t S, uint32_t ncols, uint32_t col0,
                                                       const float* __restrict__ bias, float* __restrict__ hidden, uint32_t H,
                                                       const float* __restrict__ nw, uint8_t* __restrict__ xh, float* __restrict__ ssq,
                                                       const float* __restrict__ moe_w) {
  __shared__ float s_ss[4];
  const uint32_t t = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x, i = ch * 8;
  float ss = 0.0f;
  if (i < H) {
    f32x4 v0 = *reinterpret_cast<const f32x4*>(hidden + (size_t)t * H + i), v1 = *reinterpret_cast<const f32x4*>(hidden + (size_t)t * H + i + 4);
    if (S) {
      f32x4 a0 = (f32x4)(0.0f), a1 = (f32x4)(0.0f);
      for (uint32_t s = 0; s < S; s++) {
        const float* row = part + ((size_t)s * kPfTokens + t) * ncols + col0 + i;
        if (moe_w) {   // MoeLayer::forward (moe.rs:363-368): zero-initialised, += routing weight * expert output in selection order
          const float w = moe_w[t * S + s];
          a0 += *reinterpret_cast<const f32x4*>(row) * w;
          a1 += *reinterpret_cast<const f32x4*>(row + 4) * w;
        } else {
          a0 += *reinterpret_cast<const f32x4*>(row);
          a1 += *reinterpret_cast<const f32x4*>(row + 4);
        }
      }
      if (bias) { a0 += *reinterpret_cast<const f32x4*>(bias + i); a1 += *reinterpret_cast<const f32x4*>(bias + i + 4); }
      v0 += a0;
      v1 += a1;
      *reinterpret_cast<f32x4*>(hidden + (size_t)t * H + i) = v0;
      *reinterpret_cast<f32x4*>(hidden + (size_t)t * H + i + 4) = v1;
    }
    const f32x4 q = v0 * v0 + v1 * v1;
    ss = (q.x + q.y) + (q.z + q.w);
    if (xh) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(nw + i), w1 = *reinterpret_cast<const f32x4*>(nw + i + 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; j++) { v[j] = v0[j] * w0[j]; v[4 + j] = v1[j] * w1[j]; }
      xh_store_chunk(xh, t, ch, v);
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) s_ss[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0 && ssq) ssq[t * kPfSsqChunks + blockIdx.x] = (s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]);
}

// chunks of per-token sums of squares a row of H elements is reduced to (pf_inv_rms adds them up)
uint32_t pf_ssq_chunks(uint32_t H) { return (H + 2047) / 2048; }

hipError_t pf_row_epi_launch(const float* part, uint32_t S, uint32_t ncols, uint32_t col0, const float* bias, float* hidden, uint32_t H,
                             const float* nw, uint8_t* xh, float* ssq, uint32_t m_tokens, hipStream_t st, const float* moe_w) {
  if (H % 8 || (xh && (!nw || !ssq)) || (S && (ncols % 4 || col0 % 4)) || pf_ssq_chunks(H) > (uint32_t)kPfSsqChunks) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_resid_kernel, dim3(pf_ssq_chunks(H), m_tokens), dim3(256), 0, st, part, S, ncols, col0, bias, hidden, H, nw, xh, ssq, moe_w);
  return hipGetLastError();
}

// q, k, v of token t: partial sums (+bias), RoPE on the (2i, 2i+1) — NeoX: (i, i + d/2) — pairs of q and k at position
// pos0 + t (ops.rs:1285-1337), q -> qbuf[t][QD], k / v -> cache rows pos0 + t (layers.rs:577-600)
__global__ void __launch_bounds__(256) pf_qkv_epi_kernel(const float* __restrict__ part, uint32_t S, uint32_t ncols, uint32_t QD, uint32_t KD,
                                                         uint32_t head_dim, const float* __restrict__ bq, const float* __restrict__ bk,
                                                         const float* __restrict__ bv, const float* __restrict__ rope_cs, uint32_t pos0,
                                                         uint32_t max_seq, float* __restrict__ qbuf, float* __restrict__ kcache,
                                                         float* __restrict__ vcache, const float* __restrict__ ssq, uint32_t n_ssq,
                                                         uint32_t H, float eps, int neox) {
  const uint32_t t = blockIdx.y, pos = pos0 + t, half = head_dim / 2;
  const float inv = pf_inv_rms(ssq, n_ssq, t, H, eps);
  const uint32_t npairs = (QD + 2 * KD) / 2, nq = QD / 2, nk = KD / 2;
  for (uint32_t p = blockIdx.x * 256 + threadIdx.x; p < npairs; p += gridDim.x * 256) {
    // the two columns of pair p: (2i, 2i+1) of a head for RopeType::Normal, (i, i + d/2) for NeoX (ops.rs:1316-1331); V: plain pairs
    const bool isq = p < nq, isk = !isq && p < nq + nk;
    const uint32_t pr = isq ? p : isk ? p - nq : p - nq - nk;           // pair index inside its region
    const uint32_t base = isq ? 0 : isk ? QD : QD + KD;                 // first column of the region
    const bool rot = isq || isk;
    const uint32_t i = rot ? (neox ? pr % half : (2 * pr % head_dim) / 2) : 0;
    const uint32_t r0 = rot && neox ? (pr / half) * head_dim + i : 2 * pr, r1 = rot && neox ? r0 + half : r0 + 1;   // rows of the matrix
    float x0 = 0.0f, x1 = 0.0f;
    for (uint32_t s = 0; s < S; s++) {
      const float* row = part + ((size_t)s * kPfTokens + t) * ncols + base;
      x0 += row[r0];
      x1 += row[r1];
    }
    x0 *= inv;
    x1 *= inv;
    const float* b = isq ? bq : isk ? bk : bv;
    if (b) { x0 += b[r0]; x1 += b[r1]; }
    float y0 = x0, y1 = x1;
    if (rot) {
      const float cs = rope_cs[((size_t)pos * half + i) * 2], sn = rope_cs[((size_t)pos * half + i) * 2 + 1];
      y0 = x0 * cs - x1 * sn;
      y1 = x0 * sn + x1 * cs;
    }
    if (isq) {
      qbuf[(size_t)t * QD + r0] = y0;
      qbuf[(size_t)t * QD + r1] = y1;
    } else {
      float* cache = isk ? kcache : vcache;
      cache[((size_t)(r0 / head_dim) * max_seq + pos) * head_dim + (r0 % head_dim)] = y0;
      cache[((size_t)(r1 / head_dim) * max_seq + pos) * head_dim + (r1 % head_dim)] = y1;
    }
  }
}

hipError_t pf_qkv_epi_launch(const float* part, uint32_t S, uint32_t ncols, uint32_t QD, uint32_t KD, uint32_t head_dim, const float* bq,
                             const float* bk, const float* bv, const float* rope_cs, uint32_t pos0, uint32_t max_seq, float* qbuf,
                             float* kcache, float* vcache, const float* ssq, uint32_t H, float eps, int neox, uint32_t m_tokens, hipStream_t st) {
  if (head_dim % 2 || ncols != QD + 2 * KD) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_qkv_epi_kernel, dim3(((QD + 2 * KD) / 2 + 255) / 256, m_tokens), dim3(256), 0, st, part, S, ncols, QD, KD, head_dim, bq, bk, bv, rope_cs, pos0, max_seq,
                     qbuf, kcache, vcache, ssq, pf_ssq_chunks(H), H, eps, neox);
  return hipGetLastError();
}

// act = silu(gate) * up (simd.rs:598-649) from the partial sums (gate in columns [0, F), up in [F, 2F)) -> XH[t][F]
__global__ void __launch_bounds__(256) pf_swiglu_kernel(const float* __restrict__ part, uint32_t S, uint32_t F, uint8_t* __restrict__ xh,
                                                        const float* __restrict__ ssq, uint32_t n_ssq, uint32_t H, float eps,
                                                        const int* __restrict__ row_tok, const int* __restrict__ m_count) {
  const uint32_t t = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= F / 8 || (m_count && t >= (uint32_t)*m_count)) return;
  // MoE: row t of this expert's batch is token row_tok[t] & 0xFF (its 1/rms), dense: row t is token t
  const float inv = pf_inv_rms(ssq, n_ssq, row_tok ? (uint32_t)row_tok[t] & 0xFFu : t, H, eps);
  float g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < S; s++) {
    const float* row = part + ((size_t)s * kPfTokens + t) * (2 * (size_t)F);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(row + ch * 8), g1 = *reinterpret_cast<const f32x4*>(row + ch * 8 + 4);
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(row + F + ch * 8), u1 = *reinterpret_cast<const f32x4*>(row + F + ch * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) { g[j] += g0[j]; g[4 + j] += g1[j]; u[j] += u0[j]; u[4 + j] += u1[j]; }
  }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = silu_f(g[j] * inv) * (u[j] * inv);
  xh_store_chunk(xh, t, ch, v);
}

hipError_t pf_swiglu_launch(const float* part, uint32_t S, uint32_t F, uint8_t* xh, const float* ssq, uint32_t H, float eps, uint32_t m_tokens,
                            hipStream_t st, const int* row_tok, const int* m_count) {
  if (F % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_swiglu_kernel, dim3((F / 8 + 255) / 256, m_tokens), dim3(256), 0, st, part, S, F, xh, ssq, pf_ssq_chunks(H), H, eps, row_tok, m_count);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// MoE layers: tokens grouped by expert so that every expert's matrices are read once per block of tokens
// ------------------------------------------------------------------------------------------------
// Routing table -> per-expert row lists in ONE row space (prefill.h).  One wave per expert (round-robin): 64 routing entries per
// step, positions by ballot + popcount, entry order (token-major) kept.  (A thread per expert walking the table entry by entry
// took 30 us: 256 dependent LDS reads.)  Pass 0 counts, thread 0 lays out the bases (padded to 16 rows), pass 1 writes.
__global__ void __launch_bounds__(256) pf_moe_group_kernel(const int* __restrict__ sel, uint32_t m_tokens, uint32_t top_k, uint32_t n_experts,
                                                           int* __restrict__ counts, int* __restrict__ bases, int* __restrict__ lists,
                                                           int* __restrict__ rowmap, int* __restrict__ tokmap) {
  __shared__ int s_sel[kPfTokens * kPfMaxTopK];
  __shared__ int s_cnt[kPfMaxExperts], s_base[kPfMaxExperts];
  const uint32_t n = m_tokens * top_k;
  for (uint32_t i = threadIdx.x; i < n; i += 256) s_sel[i] = sel[i];
  for (uint32_t r = threadIdx.x; r < (uint32_t)kPfMoeRows; r += 256) rowmap[r] = -1;
  __syncthreads();
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int pass = 0; pass < 2; pass++) {
    for (uint32_t e = wave; e < n_experts; e += 4) {
      uint32_t pos0 = 0;
      const uint32_t base = pass ? (uint32_t)s_base[e] : 0u;
      for (uint32_t j0 = 0; j0 < n; j0 += 64) {
        const uint32_t i = j0 + lane;
        const bool hit = i < n && (uint32_t)s_sel[i] == e;
        const unsigned long long mask = __ballot(hit);
        const uint32_t pos = pos0 + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (pass && hit && pos < (uint32_t)kPfTokens) {
          lists[e * kPfTokens + pos] = (int)((i / top_k) | (i % top_k) << 8);
          rowmap[base + pos] = (int)(e | pos << 8);
          tokmap[i] = (int)(base + pos);
        }
        pos0 += (uint32_t)__popcll(mask);
      }
      if (!pass && lane == 0) s_cnt[e] = (int)(pos0 < (uint32_t)kPfTokens ? pos0 : (uint32_t)kPfTokens);
    }
    __syncthreads();
    if (!pass && threadIdx.x == 0) {
      int b = 0;
      for (uint32_t e = 0; e < n_experts; e++) {
        s_base[e] = b;
        counts[e] = s_cnt[e];
        bases[e] = b;
        b += (s_cnt[e] + 15) & ~15;
      }
    }
    __syncthreads();
  }
}

hipError_t pf_moe_group_launch(const int* sel, uint32_t m_tokens, uint32_t top_k, uint32_t n_experts, int* counts, int* bases, int* lists,
                               int* rowmap, int* tokmap, hipStream_t st) {
  // rows: m * top_k real ones + up to 15 of padding per expert
  if (n_experts > (uint32_t)kPfMaxExperts || top_k == 0 || top_k > (uint32_t)kPfMaxTopK || m_tokens > (uint32_t)kPfTokens ||
      m_tokens * top_k + 15 * n_experts > (uint32_t)kPfMoeRows)
    return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_moe_group_kernel, dim3(1), dim3(256), 0, st, sel, m_tokens, top_k, n_experts, counts, bases, lists, rowmap, tokmap);
  return hipGetLastError();
}

// row i of the expert's XH batch = the XH row of token list[i] (the chunk swizzle depends on the row index: re-swizzled)
// (all experts in one launch: blockIdx.z = expert, its list / count / output at stride kPfTokens / 1 / out_stride)
__global__ void __launch_bounds__(256) pf_moe_gather_kernel(const uint8_t* __restrict__ xh, const int* __restrict__ list, const int* __restrict__ count,
                                                            uint8_t* __restrict__ xh_out, size_t out_stride) {
  const uint32_t slab = blockIdx.x, i = blockIdx.y * 8 + (threadIdx.x >> 5), q = threadIdx.x & 31;
  list += blockIdx.z * kPfTokens;
  count += blockIdx.z;
  xh_out += blockIdx.z * out_stride;
  if (i >= (uint32_t)*count) return;
  const uint32_t ts = (uint32_t)list[i] & 0xFFu;
  const u32x4 v = *reinterpret_cast<const u32x4*>(xh + (size_t)slab * kPfSlabBytes + ts * 512 + ((q ^ (ts & 15)) << 4));
  *reinterpret_cast<u32x4*>(xh_out + (size_t)slab * kPfSlabBytes + i * 512 + ((q ^ (i & 15)) << 4)) = v;
}

hipError_t pf_moe_gather_launch(const uint8_t* xh, uint32_t K, const int* lists, const int* counts, uint8_t* xh_out, uint32_t n_experts,
                                hipStream_t st) {
  if (K % 256 || n_experts == 0 || n_experts > (uint32_t)kPfMaxExperts) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_moe_gather_kernel, dim3(K / 256, kPfTokens / 8, n_experts), dim3(256), 0, st, xh, lists, counts, xh_out, xh_bytes(K));
  return hipGetLastError();
}

// act = silu(gate) * up (simd.rs:598-649) for the rows of ALL experts: row r of the shared row space belongs to expert
// rowmap[r] & 0xFF, is its row rowmap[r] >> 8, and carries the 1/rms of token lists[e][i] & 0xFF
__global__ void __launch_bounds__(256) pf_moe_swiglu_kernel(const float* __restrict__ part, uint32_t S, uint32_t F, uint8_t* __restrict__ xh_out,
                                                            size_t out_stride, const int* __restrict__ rowmap, const int* __restrict__ lists,
                                                            const float* __restrict__ ssq, uint32_t n_ssq, uint32_t H, float eps) {
  const uint32_t r = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x;
  const int rm = rowmap[r];
  if (rm < 0 || ch >= F / 8) return;
  const uint32_t e = (uint32_t)rm & 0xFFu, i = (uint32_t)rm >> 8;
  const float inv = pf_inv_rms(ssq, n_ssq, (uint32_t)lists[e * kPfTokens + i] & 0xFFu, H, eps);
  float g[8] = {0, 0, 0, 0, 0, 0, 0, 0}, u[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (uint32_t s = 0; s < S; s++) {
    const float* row = part + ((size_t)s * kPfMoeRows + r) * (2 * (size_t)F);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(row + ch * 8), g1 = *reinterpret_cast<const f32x4*>(row + ch * 8 + 4);
    const f32x4 u0 = *reinterpret_cast<const f32x4*>(row + F + ch * 8), u1 = *reinterpret_cast<const f32x4*>(row + F + ch * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; j++) { g[j] += g0[j]; g[4 + j] += g1[j]; u[j] += u0[j]; u[4 + j] += u1[j]; }
  }
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = silu_f(g[j] * inv) * (u[j] * inv);
  xh_store_chunk(xh_out + e * out_stride, i, ch, v);
}

hipError_t pf_moe_swiglu_launch(const float* part, uint32_t S, uint32_t F, uint8_t* xh_out, const int* rowmap, const int* lists,
                                const float* ssq, uint32_t H, float eps, hipStream_t st) {
  if (F % 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_moe_swiglu_kernel, dim3((F / 8 + 255) / 256, kPfMoeRows), dim3(256), 0, st, part, S, F, xh_out, xh_bytes(F), rowmap, lists, ssq,
                     pf_ssq_chunks(H), H, eps);
  return hipGetLastError();
}

// MoeLayer::forward's tail (moe.rs:363-368) + the layer's residual for every token of the block: out = 0; out += w[s] * expert
// output of slot s, in selection order; h += out; then the next layer's XH and sums of squares (as pf_resid_kernel).
__global__ void __launch_bounds__(256) pf_moe_combine_kernel(const float* __restrict__ part, uint32_t S, const int* __restrict__ tokmap,
                                                             const float* __restrict__ moe_w, uint32_t top_k, float* __restrict__ hidden, uint32_t H,
                                                             const float* __restrict__ nw, uint8_t* __restrict__ xh, float* __restrict__ ssq) {
  __shared__ float s_ss[4];
  const uint32_t t = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x, i = ch * 8;
  float ss = 0.0f;
  if (i < H) {
    f32x4 a0 = (f32x4)(0.0f), a1 = (f32x4)(0.0f);
    for (uint32_t s = 0; s < top_k; s++) {
      const uint32_t row = (uint32_t)tokmap[t * top_k + s];
      f32x4 y0 = (f32x4)(0.0f), y1 = (f32x4)(0.0f);
      for (uint32_t sp = 0; sp < S; sp++) {
        const float* p = part + ((size_t)sp * kPfMoeRows + row) * H + i;
        y0 += *reinterpret_cast<const f32x4*>(p);
        y1 += *reinterpret_cast<const f32x4*>(p + 4);
      }
      const float w = moe_w[t * top_k + s];
      a0 += y0 * w;
      a1 += y1 * w;
    }
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(hidden + (size_t)t * H + i) + a0;
    const f32x4 v1 = *reinterpret_cast<const f32x4*>(hidden + (size_t)t * H + i + 4) + a1;
    *reinterpret_cast<f32x4*>(hidden + (size_t)t * H + i) = v0;
    *reinterpret_cast<f32x4*>(hidden + (size_t)t * H + i + 4) = v1;
    const f32x4 q = v0 * v0 + v1 * v1;
    ss = (q.x + q.y) + (q.z + q.w);
    if (xh) {
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(nw + i), w1 = *reinterpret_cast<const f32x4*>(nw + i + 4);
      float v[8];
#pragma unroll
      for (int j = 0; j < 4; j++) { v[j] = v0[j] * w0[j]; v[4 + j] = v1[j] * w1[j]; }
      xh_store_chunk(xh, t, ch, v);
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) s_ss[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0 && ssq) ssq[t * kPfSsqChunks + blockIdx.x] = (s_ss[0] + s_ss[1]) + (s_ss[2] + s_ss[3]);
}

hipError_t pf_moe_combine_launch(const float* part, uint32_t S, const int* tokmap, const float* moe_w, uint32_t top_k, float* hidden, uint32_t H,
                                 const float* nw, uint8_t* xh, float* ssq, uint32_t m_tokens, hipStream_t st) {
  if (H % 8 || (xh && (!nw || !ssq)) || pf_ssq_chunks(H) > (uint32_t)kPfSsqChunks) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_moe_combine_kernel, dim3(pf_ssq_chunks(H), m_tokens), dim3(256), 0, st, part, S, tokmap, moe_w, top_k, hidden, H, nw, xh, ssq);
  return hipGetLastError();
}

// plain f32 [M][K] -> XH (no scaling): the attention output in front of wo
__global__ void __launch_bounds__(256) pf_to_xh_kernel(const float* __restrict__ x, uint32_t K, uint8_t* __restrict__ xh) {
  const uint32_t t = blockIdx.y, ch = blockIdx.x * 256 + threadIdx.x;
  if (ch >= K / 8) return;
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = x[(size_t)t * K + ch * 8 + j];
  xh_store_chunk(xh, t, ch, v);
}

hipError_t pf_to_xh_launch(const float* x, uint32_t K, uint8_t* xh, uint32_t m_tokens, hipStream_t st) {
  if (K % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(pf_to_xh_kernel, dim3((K / 8 + 255) / 256, m_tokens), dim3(256), 0, st, x, K, xh);
  return hipGetLastError();
}

}  // namespace lgh
