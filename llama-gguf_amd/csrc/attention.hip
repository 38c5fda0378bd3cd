// attention.hip — single-token GQA attention over the HBM-resident f32 KV cache.
//
// Replaces `flash_attention_cached` (src/backend/cuda/kernels.rs:1395-1458: one workgroup per query
// head, a SERIAL loop over kv positions with two block-wide reductions per position) and follows the
// CPU backend's `attention_cached` (src/backend/cpu/ops.rs:1479-1537): score = <q,k>*scale, softmax over
// kv_len, out = sum_p w_p * V[p].  KV layout is the reference's [n_kv_heads, max_seq, head_dim] f32
// (src/model/mod.rs:64-108).
//
// Decode attention is a latency problem at short context and an HBM stream at long context, so:
//   * work is split over (kv head) x (n_splits) workgroups, each with 4 waves; positions are dealt
//     round-robin so every workgroup has work from the first cached token on;
//   * the G = n_heads/n_kv_heads query heads of a kv group are processed together: one 16-byte K/V load
//     per lane serves all G heads (GQA reuse), K/V rows go straight to VGPRs (no LDS staging);
//   * a row of head_dim floats is spread over head_dim/4 lanes, so one wave-instruction covers 2 rows
//     (d=128) or 4 rows (d=64), fully coalesced 512-byte / 256-byte rows;
//   * online softmax per lane group; partial (m, l, acc) states merge in registers, then LDS, then in
//     a tiny combine kernel across splits.
// The reference CPU path skips softmax weights <= 1e-8 (ops.rs:1529); like the reference's own GPU
// kernel this one does not (each skipped term is < 1e-8 of the output scale).
#include <cstdlib>

#include "device_utils.h"
#include "xq.h"
#include "prefill.h"
#include "timeline.h"

LGH_TL_DEFINE(attn)

namespace lgh {

constexpr float kNegBig = -1e30f;

// NW waves per workgroup (positions are dealt round-robin over splits x waves): 4 is faster at short context (608 vs 600
// tokens/s at kv 272), 8 at long (543 vs 524 at kv 4000); the launcher picks by the cache capacity.
// PF (prefill.hip): blockIdx.y = token t of a block of prompt tokens; it sees kv_len_fixed + t cache rows (causal), its
// query is q + t * n_heads * D, there is one split, and the normalised output goes, as f16, into the XH matrix at
// `part_acc` (prefill.h: the wo GEMM's input).
// DIRECT: one workgroup of NW = 16 waves per kv head and no second kernel: the workgroup merges its waves and writes the
// normalised output (f32 at part_acc, XQ records at part_ml when not null) itself.  It saves one launch floor per layer
// but pulls a kv head's whole K/V through one CU, so it only pays below ~100 rows (engine.hip: kDirectAttnDefaultKv).
// MULTI (multi-sequence decode, engine_batch.hip): blockIdx.y = sequence s of the step; its query is q + s * n_heads * D, its
// position pos_ptr[s], its K / V the cache slot slot[s] (slot_stride floats apart), its partials the s-th block of
// part_ml / part_acc.  Per sequence the arithmetic is the single-sequence kernel's.
template <int D, int G, int NW, bool PF = false, bool DIRECT = false, bool MULTI = false>
__global__ void __launch_bounds__(NW * 64) attn_partial_kernel(const float* __restrict__ q, const float* __restrict__ kc,
                                                           const float* __restrict__ vc, uint32_t max_seq, float scale,
                                                           const int* pos_ptr, int kv_len_fixed, uint32_t n_splits,
                                                           float* __restrict__ part_ml, float* __restrict__ part_acc,
                                                           const int* __restrict__ slot = nullptr, uint64_t slot_stride = 0) {
  constexpr int LPR = D / 4;       // lanes per row
  constexpr int RPW = 64 / LPR;    // rows per wave-instruction
  __shared__ float s_ml[NW][G][2];
  __shared__ float s_acc[NW][G][D];
  LGH_TL_BEGIN(attn, lgh::TL_ATTN, n_splits);

  const uint32_t kvh = blockIdx.x / n_splits, sp = blockIdx.x % n_splits;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sub = lane / LPR, li = lane % LPR;
  // the position word by SCALAR load: the loop bounds depend on it, and a vector load here would be a full memory round
  // trip before the first K/V row can be requested
  uint32_t kv_len = (uint32_t)kv_len_fixed;
  if (PF) {
    kv_len += blockIdx.y;
    q += (size_t)blockIdx.y * gridDim.x * G * D;   // gridDim.x = kv heads (one split)
  } else if (MULTI) {
    const uint32_t sq = blockIdx.y;
    uint32_t pw, sl;
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pw), "=&s"(sl) : "s"(pos_ptr + sq), "s"(slot + sq) : "memory");
    kv_len = pw + 1;
    q += (size_t)sq * (gridDim.x / n_splits) * G * D;
    kc += (size_t)sl * slot_stride;
    vc += (size_t)sl * slot_stride;
    part_ml += (size_t)sq * gridDim.x * G * 2;
    part_acc += (size_t)sq * gridDim.x * G * D;
  } else if (pos_ptr) {
    uint32_t pw;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pw) : "s"(pos_ptr) : "memory");
    kv_len = pw + 1;
  }

  f32x4 qv[G];
#pragma unroll
  for (int g = 0; g < G; g++) qv[g] = *reinterpret_cast<const f32x4*>(q + ((size_t)kvh * G + g) * D + li * 4);

  float m[G], l[G];
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; g++) { m[g] = kNegBig; l[g] = 0.0f; acc[g] = (f32x4)(0.0f); }

  const float* kbase = kc + (size_t)kvh * max_seq * D + li * 4;
  const float* vbase = vc + (size_t)kvh * max_seq * D + li * 4;
  const uint32_t stride = n_splits * NW * RPW;
  auto row_of = [&](uint32_t base) { const uint32_t p = base + sub; return p < kv_len ? p : kv_len - 1; };   // clamped: loads are unconditional
  auto step = [&](uint32_t base, f32x4 k4, f32x4 v4) {
    const bool valid = base + sub < kv_len;
#pragma unroll
    for (int g = 0; g < G; g++) {
      float s = qv[g].x * k4.x;
      s = __builtin_fmaf(qv[g].y, k4.y, s);
      s = __builtin_fmaf(qv[g].z, k4.z, s);
      s = __builtin_fmaf(qv[g].w, k4.w, s);
      // sum over the LPR lanes of the row, in every lane: DPP inside a 16-lane row (no LDS crossbar), one cross-row
      // exchange when a row spans 32 lanes
      s += dpp_f<0xB1>(s);
      s += dpp_f<0x4E>(s);
      s += dpp_f<0x141>(s);
      s += dpp_f<0x140>(s);
      if (LPR == 32) s += __shfl_xor(s, 16, 64);
      s *= scale;
      const float mn = valid ? fmaxf(m[g], s) : m[g];
      const float a = __expf(m[g] - mn);            // v_exp_f32: ~2 ulp, far inside the attention tolerance
      const float pe = valid ? __expf(s - mn) : 0.0f;
      l[g] = __builtin_fmaf(l[g], a, pe);
      acc[g] = acc[g] * a + v4 * pe;
      m[g] = mn;
    }
  };
  // Batches of kAhead iterations: all their K/V rows are requested before the first one is used (clamped when past the
  // end, so the loads are unconditional).  At decode lengths a workgroup has 1-3 iterations and everything is in flight
  // at once; at 4K context it has ~30, and with a single iteration ahead the kernel ran at 1 TB/s (latency-bound).
  constexpr int kAhead = 4;   // (8 measured slower: 604 / 518 vs 614 / 528 tokens/s at kv 272 / 4000)
  for (uint32_t base = (sp * NW + wave) * RPW; base < kv_len; base += kAhead * stride) {
    f32x4 kk[kAhead], vv[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const uint32_t r = row_of(base + j * stride);
      kk[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(kbase + (size_t)r * D));
      vv[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(vbase + (size_t)r * D));
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++)
      if (base + j * stride < kv_len) step(base + j * stride, kk[j], vv[j]);   // wave-uniform
  }

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
  // merge the NW waves; thread t handles output elements t, t + 64 NW, ...
  const size_t pbase = ((size_t)kvh * n_splits + sp) * G;
  for (uint32_t e = threadIdx.x; e < (uint32_t)(G * D); e += NW * 64) {
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
    if (DIRECT) {
      const float o = a * (1.0f / lsum);   // simd.rs:718-720: multiply by 1/sum
      part_acc[(pbase + g) * D + dim] = o;
      if (part_ml) xq_store_chunk(reinterpret_cast<uint8_t*>(part_ml), (uint32_t)((pbase + g) * D + dim) >> 4, o);   // wo's input as XQ records
    } else if (PF) {
      const _Float16 o = (_Float16)(a * (1.0f / lsum));   // simd.rs:718-720: multiply by 1/sum
      *reinterpret_cast<_Float16*>(reinterpret_cast<uint8_t*>(part_acc) + xh_offset(blockIdx.y, (uint32_t)(pbase + g) * D + dim)) = o;
    } else {
      part_acc[(pbase + g) * D + dim] = a;
      if (dim == 0) { part_ml[(pbase + g) * 2] = mn; part_ml[(pbase + g) * 2 + 1] = lsum; }
    }
  }
  LGH_TL_END();
}

// ------------------------------------------------------------------------------------------------
// int8 KV cache (LGH_FLAG_KV_INT8): the reference's QuantizedKVCache with KVCacheFormat::Int8
// (src/model/kv_quantized.rs:143-300, 385-410).  Per layer K and V are int8 [kv_head][max_seq][D] plus one f32 scale per
// (kv_head, position): scale = max|x| / 127 (1 for an all-zero row), q = round(x / scale) clamped, and attention reads
// back scale * q.  The QKV launch leaves the current token's rows as f32 in a staging vector; every attention workgroup of
// a kv head quantizes that row itself (D values, the same bits everywhere), split 0 also stores it into the cache, and the
// row takes part in the softmax through its dequantized values — exactly what a later token will read.
// Same split / online-softmax structure as attn_partial_kernel; a row is D bytes (one dword per lane).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 unpack_i8x4(uint32_t w, float scale) {
  f32x4 r;
  r.x = (float)(int)(int8_t)(w & 0xFFu) * scale;
  r.y = (float)(int)(int8_t)((w >> 8) & 0xFFu) * scale;
  r.z = (float)(int)(int8_t)((w >> 16) & 0xFFu) * scale;
  r.w = (float)(int)(int8_t)(w >> 24) * scale;
  return r;
}

// quantizes the f32 row held 4 elements per lane by the LPR lanes of a row group; returns the packed int8 word and the scale
template <int LPR>
__device__ __forceinline__ uint32_t quantize_row_i8(f32x4 x, float& scale) {
  float amax = fmaxf(fmaxf(fabsf(x.x), fabsf(x.y)), fmaxf(fabsf(x.z), fabsf(x.w)));
  amax = fmaxf(amax, dpp_f<0xB1>(amax));
  amax = fmaxf(amax, dpp_f<0x4E>(amax));
  amax = fmaxf(amax, dpp_f<0x141>(amax));
  amax = fmaxf(amax, dpp_f<0x140>(amax));
  if (LPR == 32) amax = fmaxf(amax, __shfl_xor(amax, 16, 64));
  scale = amax > 1e-10f ? amax / 127.0f : 1.0f;
  auto qz = [&](float v) -> uint32_t {
    float r = roundf(v / scale);                           // f32::round: halves away from zero
    r = r < -128.0f ? -128.0f : (r > 127.0f ? 127.0f : r);
    return (uint32_t)(int)r & 0xFFu;
  };
  return qz(x.x) | qz(x.y) << 8 | qz(x.z) << 16 | qz(x.w) << 24;
}

// ---- the reference's FP8 KV formats (kv_quantized.rs:413-565; FMT 2 = E4M3, 3 = E5M2): one byte per element, no scales ----
// dequantize_fp8_*: case by case it is the standard decode of the format (E4M3 without infinities, 0x7F / 0xFF = NaN), except
// that both zeros read back as +0.  E5M2 is the upper byte of an IEEE half.
template <int FMT>
__device__ __forceinline__ float fp8_decode(uint32_t b) {
  const uint32_t mag = b & 0x7Fu;
  if (mag == 0) return 0.0f;
  if (FMT == 3) return (float)__builtin_bit_cast(_Float16, (uint16_t)(b << 8));
  if (mag == 0x7Fu) return __uint_as_float(0x7FC00000u);
  const uint32_t e = mag >> 3, mt = mag & 7u;
  const float v = e ? __uint_as_float((e + 120u) << 23 | mt << 20) : (float)mt * 0x1p-9f;
  return __uint_as_float(__float_as_uint(v) | (b & 0x80u) << 24);
}
// quantize_fp8_e4m3 / _e5m2 as written in the reference: the mantissa is truncated, magnitudes past the largest exponent
// saturate to 0x7E / 0x7C, and an E4M3 magnitude in [480, 512) becomes the NaN pattern 0x7F
template <int FMT>
__device__ __forceinline__ uint32_t fp8_encode(float value) {
  constexpr bool e4 = FMT == 2;
  const uint32_t bits = __float_as_uint(value);
  const uint32_t sign = (bits >> 31) << 7;
  if (value != value) return 0xFFu;
  if ((bits & 0x7FFFFFFFu) == 0x7F800000u) return e4 ? (sign ? 0xFFu : 0x7Fu) : (sign ? 0xFCu : 0x7Cu);
  if (value == 0.0f) return 0x00u;
  const int exponent = (int)((bits >> 23) & 0xFFu) - 127;
  uint32_t mantissa = bits & 0x7FFFFFu;
  if (exponent != -127) mantissa |= 0x800000u;
  if (e4) {
    const int e = exponent + 7;
    if (e > 15) return sign | 0x7Eu;
    if (e > -3 && e <= 0) return sign | ((mantissa >> (24 - (uint32_t)(3 + e))) & (0x7u >> (uint32_t)(-e)));
    if (e <= -3) return sign;
    return sign | (uint32_t)e << 3 | ((mantissa >> 20) & 0x7u);
  }
  const int e = exponent + 15;
  if (e > 31) return sign | 0x7Cu;
  if (e >= -1 && e <= 0) return sign | ((mantissa >> (24 - (uint32_t)(2 + e))) & (0x3u >> (uint32_t)(-e)));
  if (e < -1) return sign;
  return sign | (uint32_t)e << 2 | ((mantissa >> 21) & 0x3u);
}
// four cached values of one dword, any byte format (FMT 1 = int8 with the row's scale)
template <int FMT>
__device__ __forceinline__ f32x4 unpack_kv4(uint32_t w, float scale) {
  if (FMT == 1) return unpack_i8x4(w, scale);
  f32x4 r;
  r.x = fp8_decode<FMT>(w & 0xFFu); r.y = fp8_decode<FMT>((w >> 8) & 0xFFu);
  r.z = fp8_decode<FMT>((w >> 16) & 0xFFu); r.w = fp8_decode<FMT>(w >> 24);
  return r;
}

template <int D, int G, int NW, int FMT>
__global__ void __launch_bounds__(NW * 64) attn_partial_q8_kernel(const float* __restrict__ q, int8_t* __restrict__ k8, int8_t* __restrict__ v8,
                                                                  float* __restrict__ kscale, float* __restrict__ vscale,
                                                                  const float* __restrict__ k_new, const float* __restrict__ v_new,
                                                                  uint32_t max_seq, float scale, const int* pos_ptr, uint32_t n_splits,
                                                                  float* __restrict__ part_ml, float* __restrict__ part_acc) {
  constexpr int LPR = D / 4, RPW = 64 / LPR;
  __shared__ float s_ml[NW][G][2];
  __shared__ float s_acc[NW][G][D];
  const uint32_t kvh = blockIdx.x / n_splits, sp = blockIdx.x % n_splits;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t sub = lane / LPR, li = lane % LPR;
  uint32_t pw;
  asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pw) : "s"(pos_ptr) : "memory");
  const uint32_t pos = pw;   // rows [0, pos) come from the cache, row `pos` from the staging vectors
  f32x4 qv[G];
#pragma unroll
  for (int g = 0; g < G; g++) qv[g] = *reinterpret_cast<const f32x4*>(q + ((size_t)kvh * G + g) * D + li * 4);
  float m[G], l[G];
  f32x4 acc[G];
#pragma unroll
  for (int g = 0; g < G; g++) { m[g] = kNegBig; l[g] = 0.0f; acc[g] = (f32x4)(0.0f); }
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
  const size_t hrow = (size_t)kvh * max_seq;
  const uint32_t stride = n_splits * NW * RPW;
  constexpr int kAhead = 4;
  for (uint32_t base = (sp * NW + wave) * RPW; base < pos; base += kAhead * stride) {
    uint32_t kk[kAhead], vv[kAhead];
    float ks[kAhead], vs[kAhead];
#pragma unroll
    for (int j = 0; j < kAhead; j++) {
      const uint32_t p = base + j * stride + sub, r = p < pos ? p : pos - 1;
      kk[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(k8 + (hrow + r) * D + li * 4));
      vv[j] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(v8 + (hrow + r) * D + li * 4));
      ks[j] = FMT == 1 ? kscale[hrow + r] : 1.0f;
      vs[j] = FMT == 1 ? vscale[hrow + r] : 1.0f;
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++)
      if (base + j * stride < pos) step(base + j * stride + sub < pos, unpack_kv4<FMT>(kk[j], ks[j]), unpack_kv4<FMT>(vv[j], vs[j]));
  }
  if (wave == 0) {   // the current token's row: quantized here (every split the same bits), stored by split 0
    const f32x4 kx = *reinterpret_cast<const f32x4*>(k_new + (size_t)kvh * D + li * 4);
    const f32x4 vx = *reinterpret_cast<const f32x4*>(v_new + (size_t)kvh * D + li * 4);
    float ksc = 1.0f, vsc = 1.0f;
    uint32_t kq, vq;
    if (FMT == 1) {
      kq = quantize_row_i8<LPR>(kx, ksc);
      vq = quantize_row_i8<LPR>(vx, vsc);
    } else {
      kq = fp8_encode<FMT>(kx.x) | fp8_encode<FMT>(kx.y) << 8 | fp8_encode<FMT>(kx.z) << 16 | fp8_encode<FMT>(kx.w) << 24;
      vq = fp8_encode<FMT>(vx.x) | fp8_encode<FMT>(vx.y) << 8 | fp8_encode<FMT>(vx.z) << 16 | fp8_encode<FMT>(vx.w) << 24;
    }
    if (sp == 0) {
      if (sub == 0) {
        *reinterpret_cast<uint32_t*>(k8 + (hrow + pos) * D + li * 4) = kq;
        *reinterpret_cast<uint32_t*>(v8 + (hrow + pos) * D + li * 4) = vq;
        if (FMT == 1 && li == 0) { kscale[hrow + pos] = ksc; vscale[hrow + pos] = vsc; }
      }
      step(sub == 0, unpack_kv4<FMT>(kq, ksc), unpack_kv4<FMT>(vq, vsc));
    }
  }
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
  const size_t pbase = ((size_t)kvh * n_splits + sp) * G;
  for (uint32_t e = threadIdx.x; e < (uint32_t)(G * D); e += NW * 64) {
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
    part_acc[(pbase + g) * D + dim] = a;
    if (dim == 0) { part_ml[(pbase + g) * 2] = mn; part_ml[(pbase + g) * 2 + 1] = lsum; }
  }
}

template <int D, int G>
static hipError_t attn_q8_go(int fmt, const float* q, int8_t* k8, int8_t* v8, float* ks, float* vs, const float* k_new, const float* v_new, uint32_t n_kv,
                             uint32_t max_seq, float scale, const int* pos, uint32_t n_splits, float* part_ml, float* part_acc, hipStream_t st) {
#define LGH_Q8_GO(FMT)                                                                                                                          \
  hipLaunchKernelGGL((attn_partial_q8_kernel<D, G, 4, FMT>), dim3(n_kv * n_splits), dim3(256), 0, st, q, k8, v8, ks, vs, k_new, v_new, max_seq, \
                     scale, pos, n_splits, part_ml, part_acc)
  if (fmt == 1) LGH_Q8_GO(1);
  else if (fmt == 2) LGH_Q8_GO(2);
  else if (fmt == 3) LGH_Q8_GO(3);
  else return hipErrorInvalidValue;
#undef LGH_Q8_GO
  return hipGetLastError();
}

// fmt: lgh_model_desc.kv_cache_type (1 = int8 + scales, 2 = FP8 E4M3, 3 = FP8 E5M2; the scale arrays are unused for 2 and 3)
hipError_t attn_q8_launch(int fmt, const float* q, int8_t* k8, int8_t* v8, float* kscale, float* vscale, const float* k_new, const float* v_new,
                          uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, uint32_t n_splits,
                          float* part_ml, float* part_acc, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || !pos) return hipErrorInvalidValue;
  const uint32_t g = n_heads / n_kv;
#define LGH_ATTN_CASE(DD, GG) \
  if (head_dim == DD && g == GG) return attn_q8_go<DD, GG>(fmt, q, k8, v8, kscale, vscale, k_new, v_new, n_kv, max_seq, scale, pos, n_splits, part_ml, part_acc, st);
  LGH_ATTN_CASE(128, 1) LGH_ATTN_CASE(128, 2) LGH_ATTN_CASE(128, 4) LGH_ATTN_CASE(128, 8)
  LGH_ATTN_CASE(64, 1) LGH_ATTN_CASE(64, 2) LGH_ATTN_CASE(64, 4) LGH_ATTN_CASE(64, 8)
#undef LGH_ATTN_CASE
  return hipErrorInvalidValue;
}

// one row of n values through a byte KV format and back (the quantizers of attn_partial_q8_kernel, stand-alone): bytes and,
// for int8, the row's scale; `back` = what attention would read
template <int FMT>
__global__ void __launch_bounds__(64) kv_roundtrip_kernel(const float* __restrict__ x, uint32_t n, uint8_t* __restrict__ bytes,
                                                          float* __restrict__ scale_out, float* __restrict__ back) {
  const uint32_t lane = threadIdx.x;
  float scale = 1.0f;
  if (FMT == 1) {   // quantize_int8: one scale per row
    float amax = 0.0f;
    for (uint32_t i = lane; i < n; i += 64) amax = fmaxf(amax, fabsf(x[i]));
    amax = wave_max(amax);
    scale = amax > 1e-10f ? amax / 127.0f : 1.0f;
    if (lane == 0) *scale_out = scale;
  }
  for (uint32_t i = lane; i < n; i += 64) {
    uint32_t b;
    if (FMT == 1) {
      float r = roundf(x[i] / scale);
      r = r < -128.0f ? -128.0f : (r > 127.0f ? 127.0f : r);
      b = (uint32_t)(int)r & 0xFFu;
      back[i] = (float)(int)(int8_t)b * scale;
    } else {
      b = fp8_encode<FMT>(x[i]);
      back[i] = fp8_decode<FMT>(b);
    }
    bytes[i] = (uint8_t)b;
  }
}

hipError_t kv_roundtrip_launch(int fmt, const float* x, uint32_t n, uint8_t* bytes, float* scale_out, float* back, hipStream_t st) {
  if (fmt == 1) hipLaunchKernelGGL(kv_roundtrip_kernel<1>, dim3(1), dim3(64), 0, st, x, n, bytes, scale_out, back);
  else if (fmt == 2) hipLaunchKernelGGL(kv_roundtrip_kernel<2>, dim3(1), dim3(64), 0, st, x, n, bytes, scale_out, back);
  else if (fmt == 3) hipLaunchKernelGGL(kv_roundtrip_kernel<3>, dim3(1), dim3(64), 0, st, x, n, bytes, scale_out, back);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// out[h][dim] = sum_s acc_s * e^{m_s - m*} / sum_s l_s * e^{m_s - m*}.  Lane s of the first wave owns split s
// (all loads of a phase are independent and in flight together: the kernel is two memory round trips long).
template <bool MULTI>
__global__ void __launch_bounds__(128) attn_combine_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_acc,
                                                           uint32_t g_per_kv, uint32_t head_dim, uint32_t n_splits,
                                                           float* __restrict__ out, uint8_t* __restrict__ xq_out) {
  __shared__ float s_f[64];
  __shared__ float s_linv;
  LGH_TL_BEGIN(attn, lgh::TL_COMBINE, n_splits);
  if (MULTI) {   // blockIdx.y = sequence: its partials, its output vector and XQ image
    const uint32_t sq = blockIdx.y, n_heads = gridDim.x;
    part_ml += (size_t)sq * n_heads * n_splits * 2;
    part_acc += (size_t)sq * n_heads * n_splits * head_dim;
    out += (size_t)sq * n_heads * head_dim;
    if (xq_out) xq_out += (size_t)sq * xq_bytes((size_t)n_heads * head_dim);
  }
  const uint32_t h = blockIdx.x, kvh = h / g_per_kv, g = h % g_per_kv;
  const size_t p0 = (size_t)kvh * n_splits * g_per_kv + g;   // split s lives at p0 + s * g_per_kv
  // every partial this thread will need is requested up front (one dim per thread: blockDim == head_dim <= 128), so the
  // kernel is ONE memory round trip long instead of two dependent ones (m/l first, accumulators after the barrier)
  const uint32_t dim = threadIdx.x;
  float pa[32];
#pragma unroll
  for (uint32_t sidx = 0; sidx < 32; sidx++)
    pa[sidx] = (sidx < n_splits && dim < head_dim) ? part_acc[(p0 + (size_t)sidx * g_per_kv) * head_dim + dim] : 0.0f;
  if (threadIdx.x < 64) {
    const uint32_t sidx = threadIdx.x;
    const bool ok = sidx < n_splits;
    const float m = ok ? part_ml[(p0 + (size_t)sidx * g_per_kv) * 2] : kNegBig;
    const float l = ok ? part_ml[(p0 + (size_t)sidx * g_per_kv) * 2 + 1] : 0.0f;
    const float mn = wave_max(m);
    const float f = expf(m - mn);
    const float lsum = wave_sum(l * f);
    s_f[sidx] = f;
    if (sidx == 0) s_linv = 1.0f / lsum;  // simd.rs:718-720: multiply by 1/sum
  }
  __syncthreads();
  if (dim < head_dim) {
    float a = 0.0f;
#pragma unroll
    for (uint32_t sidx = 0; sidx < 32; sidx++) a += pa[sidx] * s_f[sidx];   // s_f of absent splits is exp(-1e30 - m) = 0
    const float o = a * s_linv;
    out[(size_t)h * head_dim + dim] = o;
    if (xq_out) xq_store_chunk(xq_out, (h * head_dim + dim) >> 4, o);   // wo's input as XQ records (head_dim % 16 == 0)
  }
  LGH_TL_END();
}

template <int D, int G>
static hipError_t attn_go(const float* q, const float* kc, const float* vc, uint32_t n_kv, uint32_t max_seq, float scale,
                          const int* pos, int kv_len_fixed, uint32_t n_splits, float* part_ml, float* part_acc,
                          hipStream_t st) {
  const bool long_ctx = (pos ? max_seq : (uint32_t)kv_len_fixed) >= 2048;
  if (long_ctx)
    hipLaunchKernelGGL((attn_partial_kernel<D, G, 8>), dim3(n_kv * n_splits), dim3(512), 0, st, q, kc, vc, max_seq, scale, pos,
                       kv_len_fixed, n_splits, part_ml, part_acc);
  else
    hipLaunchKernelGGL((attn_partial_kernel<D, G, 4>), dim3(n_kv * n_splits), dim3(256), 0, st, q, kc, vc, max_seq, scale, pos,
                       kv_len_fixed, n_splits, part_ml, part_acc);
  return hipGetLastError();
}

hipError_t attn_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv,
                       uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, int kv_len_fixed,
                       uint32_t n_splits, float* part_ml, float* part_acc, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv) return hipErrorInvalidValue;
  const uint32_t g = n_heads / n_kv;
#define LGH_ATTN_CASE(DD, GG) \
  if (head_dim == DD && g == GG)  \
    return attn_go<DD, GG>(q, kcache, vcache, n_kv, max_seq, scale, pos, kv_len_fixed, n_splits, part_ml, part_acc, st);
  LGH_ATTN_CASE(128, 1) LGH_ATTN_CASE(128, 2) LGH_ATTN_CASE(128, 4) LGH_ATTN_CASE(128, 8)
  LGH_ATTN_CASE(64, 1) LGH_ATTN_CASE(64, 2) LGH_ATTN_CASE(64, 4) LGH_ATTN_CASE(64, 8)
#undef LGH_ATTN_CASE
  return hipErrorInvalidValue;
}

template <int D, int G>
static hipError_t attn_direct_go(const float* q, const float* kc, const float* vc, uint32_t n_kv, uint32_t max_seq, float scale, const int* pos,
                                 float* out, uint8_t* xq_out, hipStream_t st) {
  hipLaunchKernelGGL((attn_partial_kernel<D, G, 16, false, true>), dim3(n_kv), dim3(1024), 0, st, q, kc, vc, max_seq, scale, pos, 0, 1u,
                     reinterpret_cast<float*>(xq_out), out);
  return hipGetLastError();
}

// single-launch decode attention for short contexts (engine.hip picks it by the host-side position)
hipError_t attn_direct_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                              uint32_t max_seq, float scale, const int* pos, float* out, uint8_t* xq_out, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || !pos) return hipErrorInvalidValue;
  const uint32_t g = n_heads / n_kv;
#define LGH_ATTN_CASE(DD, GG) \
  if (head_dim == DD && g == GG) return attn_direct_go<DD, GG>(q, kcache, vcache, n_kv, max_seq, scale, pos, out, xq_out, st);
  LGH_ATTN_CASE(128, 1) LGH_ATTN_CASE(128, 2) LGH_ATTN_CASE(128, 4) LGH_ATTN_CASE(128, 8)
  LGH_ATTN_CASE(64, 1) LGH_ATTN_CASE(64, 2) LGH_ATTN_CASE(64, 4) LGH_ATTN_CASE(64, 8)
#undef LGH_ATTN_CASE
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// Causal attention of a block of prompt tokens on the matrix cores, in f32 (v_mfma_f32_16x16x4_f32: exact f32 products,
// f32 accumulation — no f16 rounding enters here).  A workgroup = one query head x 16 tokens; its 4 waves take the kv tiles
// (16 cached rows each) round-robin, each with its own online-softmax state, merged through LDS at the end.
// Per kv tile a wave computes S^T = K Q^T (M = kv row, N = token; the contraction runs over d, lane group c owning the
// D/4 contiguous dims [c D/4, (c+1) D/4) of both operands, so K and Q fragments are plain 16-byte loads) and then
// O^T += V^T P^T: the D layout of S^T gives lane (token n, group c) the four kv rows 4c..4c+3 of token n, which is exactly
// the B operand of four MFMAs whose contraction index c stands for kv row 4c+i — the probabilities never leave their
// registers.  The A operand of those is V^T with output row m standing for dims D/16 m .. D/16 m + D/16 - 1 (contiguous
// loads again).  The VALU kernel this replaces (still taken by other head sizes; LGH_PF_ATTN_VALU=1 forces it) spent 17 us per
// layer of a 128-token Llama-3-8B prompt, bound by vector issue (four workgroups of four waves per CU); this one 10.7 us
// (8 waves, the next tile's fragments in flight during a tile; 11.7 us with 4 waves), and the 512-token prompt 19.0 instead
// of 21.8 ms.
// ------------------------------------------------------------------------------------------------
template <int D, int NW>
__global__ void __launch_bounds__(NW * 64) attn_pf_mfma_kernel(const float* __restrict__ q, const float* __restrict__ kc, const float* __restrict__ vc,
                                                               uint32_t n_heads, uint32_t g_per_kv, uint32_t max_seq, float scale, uint32_t pos0,
                                                               uint32_t m_tokens, uint8_t* __restrict__ xh_out) {
  constexpr int DJ = D / 4;    // contraction steps of S^T (dims per lane group)
  constexpr int DQ = D / 16;   // output dims per M index
  constexpr int DP = D + 4;    // padded row of the merge buffer
  extern __shared__ __attribute__((aligned(16))) float pf_lds[];
  float (*s_o)[16][DP] = reinterpret_cast<float (*)[16][DP]>(pf_lds);                 // [NW][16][DP]
  float (*s_m)[16] = reinterpret_cast<float (*)[16]>(pf_lds + NW * 16 * DP);          // [NW][16]
  float (*s_l)[16] = s_m + NW;
  const uint32_t head = blockIdx.x, tt = blockIdx.y, kvh = head / g_per_kv;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n = lane & 15, c = lane >> 4;
  const uint32_t tok = tt * 16 + n, tokc = tok < m_tokens ? tok : m_tokens - 1;
  const uint32_t kv_last = pos0 + m_tokens - 1;                               // last cached row
  const uint32_t n_tiles = (pos0 + tt * 16 + 16 + 15) / 16;                   // tiles the last token of this tile can see
  float qf[DJ];
  {
    const float* qp = q + ((size_t)tokc * n_heads + head) * D + c * DJ;
#pragma unroll
    for (int j = 0; j < DJ; j += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(qp + j);
      qf[j] = v.x; qf[j + 1] = v.y; qf[j + 2] = v.z; qf[j + 3] = v.w;
    }
  }
  const float* kbase = kc + (size_t)kvh * max_seq * D;
  const float* vbase = vc + (size_t)kvh * max_seq * D;
  float m_run = kNegBig, l_run = 0.0f;
  f32x4 acc[DQ];
#pragma unroll
  for (int x = 0; x < DQ; x++) acc[x] = (f32x4)(0.0f);
  // a tile's K fragment (row kv0 + n, dims [c DJ, (c+1) DJ)) and V fragments (rows kv0 + 4c + i, dims [n DQ, (n+1) DQ)); rows
  // past the block are clamped here and masked below
  f32x4 kn[DJ / 4], vn[4][DQ / 4];
  auto load_tile = [&](uint32_t tile) {
    const uint32_t kv0 = tile * 16;
    const uint32_t kr = kv0 + n < kv_last ? kv0 + n : kv_last;
    const float* kp = kbase + (size_t)kr * D + c * DJ;
#pragma unroll
    for (int j = 0; j < DJ / 4; j++) kn[j] = *reinterpret_cast<const f32x4*>(kp + 4 * j);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint32_t r = kv0 + 4 * c + i, vr = r < kv_last ? r : kv_last;
      const float* vp = vbase + (size_t)vr * D + n * DQ;
#pragma unroll
      for (int x = 0; x < DQ / 4; x++) vn[i][x] = *reinterpret_cast<const f32x4*>(vp + 4 * x);
    }
  };
  if (wave < n_tiles) load_tile(wave);
  for (uint32_t tile = wave; tile < n_tiles; tile += NW) {
    const uint32_t kv0 = tile * 16;
    f32x4 kf[DJ / 4], vf[4][DQ / 4];
#pragma unroll
    for (int j = 0; j < DJ / 4; j++) kf[j] = kn[j];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int x = 0; x < DQ / 4; x++) vf[i][x] = vn[i][x];
    if (tile + NW < n_tiles) load_tile(tile + NW);   // the next tile's fragments are in flight while this one is computed
    f32x4 st = (f32x4)(0.0f);
#pragma unroll
    for (int j = 0; j < DJ; j++) st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[j / 4][j % 4], qf[j], st, 0, 0, 0);
    // st[i] = <q(token n), k(row kv0 + 4c + i)>; token n sees rows <= pos0 + tok
    float sv[4], mx = kNegBig;
    bool ok[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      ok[i] = kv0 + 4 * c + i <= pos0 + tok;
      sv[i] = ok[i] ? st[i] * scale : kNegBig;
      mx = fmaxf(mx, sv[i]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - mn);
    float pv[4], ps = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++) { pv[i] = ok[i] ? __expf(sv[i] - mn) : 0.0f; ps += pv[i]; }
    ps += __shfl_xor(ps, 16, 64);
    ps += __shfl_xor(ps, 32, 64);
    l_run = __builtin_fmaf(l_run, alpha, ps);
    m_run = mn;
#pragma unroll
    for (int x = 0; x < DQ; x++) acc[x] = acc[x] * alpha;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int x = 0; x < DQ; x++) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i][x / 4][x % 4], pv[i], acc[x], 0, 0, 0);
  }
  // acc[x][i] = O^T[dim DQ (4c + i) + x][token n] of this wave's tiles
  if (c == 0) { s_m[wave][n] = m_run; s_l[wave][n] = l_run; }
#pragma unroll
  for (int x = 0; x < DQ; x++) {
    s_o[wave][n][DQ * (4 * c + 0) + x] = acc[x].x;
    s_o[wave][n][DQ * (4 * c + 1) + x] = acc[x].y;
    s_o[wave][n][DQ * (4 * c + 2) + x] = acc[x].z;
    s_o[wave][n][DQ * (4 * c + 3) + x] = acc[x].w;
  }
  __syncthreads();
  for (uint32_t e = threadIdx.x; e < 16u * D; e += NW * 64) {
    const uint32_t t = e / D, dim = e % D;
    if (tt * 16 + t >= m_tokens) continue;
    float mn = s_m[0][t];
#pragma unroll
    for (int w = 1; w < NW; w++) mn = fmaxf(mn, s_m[w][t]);
    float lsum = 0.0f, a = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float f = expf(s_m[w][t] - mn);
      lsum += s_l[w][t] * f;
      a += s_o[w][t][dim] * f;
    }
    const _Float16 o = (_Float16)(a * (1.0f / lsum));   // simd.rs:718-720: multiply by 1/sum
    *reinterpret_cast<_Float16*>(xh_out + xh_offset(tt * 16 + t, head * D + dim)) = o;
  }
}

template <int D>
static hipError_t attn_pf_mfma_go(const float* q, const float* kc, const float* vc, uint32_t n_heads, uint32_t g, uint32_t max_seq, float scale,
                                  uint32_t pos0, uint32_t m_tokens, uint8_t* xh_out, hipStream_t st) {
  constexpr int NW = 8;   // (4 waves: 11.7 us per layer of a 128-token Llama-3-8B prompt, the last token tile's waves take two kv tiles each)
  constexpr size_t lds = (size_t)(NW * 16 * (D + 4) + 2 * NW * 16) * 4;
  static bool attr_set[64] = {};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&attn_pf_mfma_kernel<D, NW>), (int)lds, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL((attn_pf_mfma_kernel<D, NW>), dim3(n_heads, (m_tokens + 15) / 16), dim3(NW * 64), lds, st, q, kc, vc, n_heads, g, max_seq, scale,
                     pos0, m_tokens, xh_out);
  return hipGetLastError();
}

// 4 waves per (kv head, token) workgroup: 8 and 16 waves measured slower on the 128-token Llama-3-8B prompt (4.79 / 5.07 vs
// 4.74 ms per prompt pass) — the grid is already 1024 workgroups wide.
template <int D, int G>
static hipError_t attn_pf_go(const float* q, const float* kc, const float* vc, uint32_t n_kv, uint32_t max_seq, float scale, uint32_t pos0,
                             uint32_t m_tokens, uint8_t* xh_out, hipStream_t st) {
  hipLaunchKernelGGL((attn_partial_kernel<D, G, 4, true>), dim3(n_kv, m_tokens), dim3(256), 0, st, q, kc, vc, max_seq, scale,
                     (const int*)nullptr, (int)(pos0 + 1), 1u, (float*)nullptr, reinterpret_cast<float*>(xh_out));
  return hipGetLastError();
}

hipError_t attn_prefill_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv,
                               uint32_t head_dim, uint32_t max_seq, float scale, uint32_t pos0, uint32_t m_tokens, uint8_t* xh_out,
                               hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || m_tokens == 0) return hipErrorInvalidValue;
  const uint32_t g = n_heads / n_kv;
  static const bool valu = [] { const char* e = std::getenv("LGH_PF_ATTN_VALU"); return e && std::atoi(e) != 0; }();   // (A/B switch: the VALU kernel)
  if (!valu && (head_dim == 128 || head_dim == 64)) {
    return head_dim == 128 ? attn_pf_mfma_go<128>(q, kcache, vcache, n_heads, g, max_seq, scale, pos0, m_tokens, xh_out, st)
                           : attn_pf_mfma_go<64>(q, kcache, vcache, n_heads, g, max_seq, scale, pos0, m_tokens, xh_out, st);
  }
#define LGH_ATTN_CASE(DD, GG) \
  if (head_dim == DD && g == GG) return attn_pf_go<DD, GG>(q, kcache, vcache, n_kv, max_seq, scale, pos0, m_tokens, xh_out, st);
  LGH_ATTN_CASE(128, 1) LGH_ATTN_CASE(128, 2) LGH_ATTN_CASE(128, 4) LGH_ATTN_CASE(128, 8)
  LGH_ATTN_CASE(64, 1) LGH_ATTN_CASE(64, 2) LGH_ATTN_CASE(64, 4) LGH_ATTN_CASE(64, 8)
#undef LGH_ATTN_CASE
  return hipErrorInvalidValue;
}

// Backend::attention for any head_dim / group size (per-op surface only; the engine's shapes take the kernels above):
// one workgroup per (head, query position), scores in LDS, the arithmetic of ops.rs:1353-1472 — sequential dot per score,
// max, exp, sum, weights times 1/sum, then V accumulated in position order.
__global__ void __launch_bounds__(256) attn_generic_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                           float* __restrict__ out, uint32_t per_kv, uint32_t seq_len, uint32_t kv_len,
                                                           uint32_t kv_rows, uint32_t d, float scale) {
  extern __shared__ float sc[];
  __shared__ float s_red[4];
  const uint32_t head = blockIdx.x, s = blockIdx.y, kvh = head / per_kv;
  const uint32_t q_abs = (kv_len >= seq_len ? kv_len - seq_len : 0) + s;
  const uint32_t visible = q_abs + 1 < kv_len ? q_abs + 1 : kv_len;
  const float* qv = q + ((size_t)head * seq_len + s) * d;
  const float* kb = k + (size_t)kvh * kv_rows * d;   // kv_rows: rows per kv head in memory (a cache: max_seq_len)
  const float* vb = v + (size_t)kvh * kv_rows * d;
  float m = -INFINITY;
  for (uint32_t p = threadIdx.x; p < visible; p += 256) {
    float dot = 0.0f;
    for (uint32_t i = 0; i < d; i++) dot += qv[i] * kb[(size_t)p * d + i];
    sc[p] = dot * scale;
    m = fmaxf(m, sc[p]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (uint32_t p = threadIdx.x; p < visible; p += 256) {
    const float e = expf(sc[p] - m);
    sc[p] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  for (uint32_t i = threadIdx.x; i < d; i += 256) {
    float o = 0.0f;
    for (uint32_t p = 0; p < visible; p++) o += (sc[p] * inv) * vb[(size_t)p * d + i];
    out[((size_t)head * seq_len + s) * d + i] = o;
  }
}

hipError_t attn_generic_launch(const float* q, const float* k, const float* v, float* out, uint32_t n_heads, uint32_t n_kv, uint32_t seq_len,
                               uint32_t kv_len, uint32_t kv_rows, uint32_t d, float scale, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || seq_len == 0 || kv_len == 0 || kv_len > 16000 || kv_rows < kv_len || seq_len > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_generic_kernel, dim3(n_heads, seq_len), dim3(256), (size_t)kv_len * 4, st, q, k, v, out, n_heads / n_kv, seq_len,
                     kv_len, kv_rows, d, scale);
  return hipGetLastError();
}

// Decode attention for ANY head_dim / query-heads-per-kv-head inside the engine (the templated kernels above cover
// head_dim 64 / 128 with 1, 2, 4, 8 heads per kv head): one workgroup per query head, kv_len = *pos + 1 from the device,
// scores in LDS (max_seq floats), the arithmetic of ops.rs:1479-1537 (dot per score, max, exp, sum, weights x 1/sum, V
// accumulated in position order).  Slower than the split kernels, correct for every shape the reference's kernel takes
// (kernels.rs:1395-1458).
__global__ void __launch_bounds__(256) attn_decode_any_kernel(const float* __restrict__ q, const float* __restrict__ kc, const float* __restrict__ vc,
                                                              float* __restrict__ out, uint32_t per_kv, uint32_t max_seq, uint32_t d, float scale,
                                                              const int* __restrict__ pos_ptr) {
  extern __shared__ float sc[];
  __shared__ float s_red[4];
  const uint32_t head = blockIdx.x, kvh = head / per_kv, kv_len = (uint32_t)*pos_ptr + 1;
  const float* qv = q + (size_t)head * d;
  const float* kb = kc + (size_t)kvh * max_seq * d;
  const float* vb = vc + (size_t)kvh * max_seq * d;
  float m = -INFINITY;
  for (uint32_t p = threadIdx.x; p < kv_len; p += 256) {
    float dot = 0.0f;
    for (uint32_t i = 0; i < d; i++) dot += qv[i] * kb[(size_t)p * d + i];
    sc[p] = dot * scale;
    m = fmaxf(m, sc[p]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (uint32_t p = threadIdx.x; p < kv_len; p += 256) {
    const float e = expf(sc[p] - m);
    sc[p] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  for (uint32_t i = threadIdx.x; i < d; i += 256) {
    float o = 0.0f;
    for (uint32_t p = 0; p < kv_len; p++) o += (sc[p] * inv) * vb[(size_t)p * d + i];
    out[(size_t)head * d + i] = o;
  }
}

bool attn_shape_has_fast_kernel(uint32_t head_dim, uint32_t group) {
  return (head_dim == 64 || head_dim == 128) && (group == 1 || group == 2 || group == 4 || group == 8);
}

hipError_t attn_decode_any_launch(const float* q, const float* kcache, const float* vcache, float* out, uint32_t n_heads, uint32_t n_kv,
                                  uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || !pos || (size_t)max_seq * 4 > 150 * 1024) return hipErrorInvalidValue;
  static bool attr_set[64] = {};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&attn_decode_any_kernel), 152 * 1024, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL(attn_decode_any_kernel, dim3(n_heads), dim3(256), (size_t)max_seq * 4, st, q, kcache, vcache, out, n_heads / n_kv, max_seq,
                     head_dim, scale, pos);
  return hipGetLastError();
}

hipError_t attn_combine_launch(const float* part_ml, const float* part_acc, uint32_t n_heads, uint32_t n_kv,
                               uint32_t head_dim, uint32_t n_splits, float* out, uint8_t* xq_out, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || head_dim % 16 || head_dim > 128 || n_splits > 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_combine_kernel<false>, dim3(n_heads), dim3(head_dim > 64 ? 128 : 64), 0, st, part_ml, part_acc,
                     n_heads / n_kv, head_dim, n_splits, out, xq_out);
  return hipGetLastError();
}

// ---- multi-sequence decode (engine_batch.hip): the split attention and its merge for n_seq sequences in one launch each
template <int D, int G>
static hipError_t attn_multi_go(const float* q, const float* kc, const float* vc, uint32_t n_kv, uint32_t max_seq, float scale, const int* pos,
                                const int* slot, uint64_t slot_stride, uint32_t n_seq, uint32_t n_splits, float* part_ml, float* part_acc,
                                hipStream_t st) {
  if (max_seq >= 2048)
    hipLaunchKernelGGL((attn_partial_kernel<D, G, 8, false, false, true>), dim3(n_kv * n_splits, n_seq), dim3(512), 0, st, q, kc, vc, max_seq, scale,
                       pos, 0, n_splits, part_ml, part_acc, slot, slot_stride);
  else
    hipLaunchKernelGGL((attn_partial_kernel<D, G, 4, false, false, true>), dim3(n_kv * n_splits, n_seq), dim3(256), 0, st, q, kc, vc, max_seq, scale,
                       pos, 0, n_splits, part_ml, part_acc, slot, slot_stride);
  return hipGetLastError();
}

hipError_t attn_multi_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                             uint32_t max_seq, float scale, const int* pos, const int* slot, uint64_t slot_stride, uint32_t n_seq,
                             uint32_t n_splits, float* part_ml, float* part_acc, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || !pos || !slot || n_seq == 0 || n_splits == 0 || n_splits > 32) return hipErrorInvalidValue;
  const uint32_t g = n_heads / n_kv;
#define LGH_ATTN_CASE(DD, GG) \
  if (head_dim == DD && g == GG)  \
    return attn_multi_go<DD, GG>(q, kcache, vcache, n_kv, max_seq, scale, pos, slot, slot_stride, n_seq, n_splits, part_ml, part_acc, st);
  LGH_ATTN_CASE(128, 1) LGH_ATTN_CASE(128, 2) LGH_ATTN_CASE(128, 4) LGH_ATTN_CASE(128, 8)
  LGH_ATTN_CASE(64, 1) LGH_ATTN_CASE(64, 2) LGH_ATTN_CASE(64, 4) LGH_ATTN_CASE(64, 8)
#undef LGH_ATTN_CASE
  return hipErrorInvalidValue;
}

hipError_t attn_combine_multi_launch(const float* part_ml, const float* part_acc, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                                     uint32_t n_splits, uint32_t n_seq, float* out, uint8_t* xq_out, hipStream_t st) {
  if (n_kv == 0 || n_heads % n_kv || head_dim % 16 || head_dim > 128 || n_splits > 32 || n_seq == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(attn_combine_kernel<true>, dim3(n_heads, n_seq), dim3(head_dim > 64 ? 128 : 64), 0, st, part_ml, part_acc,
                     n_heads / n_kv, head_dim, n_splits, out, xq_out);
  return hipGetLastError();
}

}  // namespace lgh
