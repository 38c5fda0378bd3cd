// attention_tq.hip — decode attention over the reference's TurboQuant KV cache (KVCacheType::TurboQuantMSE { bits: 2 | 3 } and
// TurboQuantProd { bits: 2 | 3 }: what `--kv-cache-type turboquant2 | turboquant3 (tq2 | tq3)` and `turboquant2-qjl | turboquant3-qjl
// (tq2-qjl | tq3-qjl)` select, src/config.rs:808-817, src/model/mod.rs:182-213).
//
// Replaces Backend::attention_turboquant (src/backend/mod.rs:240-264; its CPU body TurboQuantKVCache::attention_layer,
// src/model/kv_turboquant.rs:127-201; the reference's CUDA twin `turboquant_attention_2bit`, src/backend/cuda/kernels.rs:
// 1573-1687) and TurboQuantKVCache::write_kv (kv_turboquant.rs:88-122), as called from Attention::forward_turboquant
// (src/model/layers.rs:843-852).
//
// Format (src/model/turboquant/): a head's K or V row x (head_dim f32) is stored as codes of  y = (1 / sqrt(d)) H D x  — D the
// engine's random sign vector (rotation.rs:58-76; the signs are an INPUT of the library, lgh_set_kv_rotation_signs), H the
// Walsh-Hadamard butterfly in the reference's order (rotation.rs:113-130) — every coordinate replaced by the index of its
// Lloyd-Max cell for N(0, 1 / d) (codebook.rs:79-92: the number of boundaries it is >=), 2 bits (4 per byte) or 3 bits (8 per
// 3 bytes, little-endian 24-bit groups) (codebook.rs:131-170).  CODES ARE BIT-EXACT with the reference's arithmetic: the
// butterfly performs the same additions in the same order (tests: lgh_op_tq_compress vs the oracle).
// Attention (kv_turboquant.rs:127-172): score_p = sum_i (H D q)_i c[K_p,i] times scale; softmax; out = sum_p w_p R^-1(c[V_p]).
// The reference adds a row's 128 products one after the other and inverts the rotation per position.  Here (second structure,
// round 3: the first one walked a row's coordinates sequentially in one lane — 464 tokens/s at kv 200, 276 at kv 4000 on the bench
// model, against 545 / 490 now and 617 / 550 with the f32 cache) a cached row is shared by D / 16 lanes, 16 coordinates each: the
// lane's 4 (2-bit) or 6 (3-bit) code bytes are decoded to centroids in registers, the partial dot products meet by DPP, one pass
// with an online softmax accumulates sum_p w_p c[V_p] in the ROTATED space, and the merge kernel inverts the rotation ONCE per head
// (R^-1 is linear; rotation.rs:80-96) — the same terms in another order, equal to f32 rounding; like the f32 attention it does not
// apply the reference's `weight < 1e-8` skip (each skipped term is < 1e-8 of the output scale).
//
// Launch structure = the f32 / int8 caches': (kv head x split) workgroups leave (m, l, acc) partials, a merge kernel per query
// head combines them — plus, here, the inverse rotation and the XQ image for the output projection.  The QKV launch leaves the
// current token's rotated-by-RoPE K row and V row as f32 in a staging vector; every workgroup of a kv head compresses them
// itself (identical bits everywhere), split 0 stores the codes at row `pos`, and the row takes part through its codes.
#include <cmath>

#include "device_utils.h"
#include "xq.h"

namespace lgh {

struct TqTables { float cen[8]; float bnd[7]; float norm, inv_scale, inv_d, qjl_coeff; };   // host-computed (the oracle's own arithmetic)

template <int BITS>
__device__ __forceinline__ uint32_t tq_quantize(const TqTables& T, float v) {   // codebook.rs:79-92 (boundaries ascending)
  uint32_t idx = 0;
#pragma unroll
  for (int i = 0; i < (1 << BITS) - 1; i++) idx += v >= T.bnd[i] ? 1u : 0u;
  return idx;
}
template <int BITS>
__device__ __forceinline__ float tq_centroid(const TqTables& T, uint32_t idx) {   // selects, no indexed register file
  if (BITS == 2) {
    const float lo = (idx & 1u) ? T.cen[1] : T.cen[0], hi = (idx & 1u) ? T.cen[3] : T.cen[2];
    return (idx & 2u) ? hi : lo;
  }
  const float a = (idx & 1u) ? T.cen[1] : T.cen[0], b = (idx & 1u) ? T.cen[3] : T.cen[2];
  const float c = (idx & 1u) ? T.cen[5] : T.cen[4], d = (idx & 1u) ? T.cen[7] : T.cen[6];
  const float ab = (idx & 2u) ? b : a, cd = (idx & 2u) ? d : c;
  return (idx & 4u) ? cd : ab;
}
template <int BITS>
__device__ __forceinline__ uint32_t tq_index(const uint8_t* row, uint32_t i) {   // codebook.rs:173-252
  if (BITS == 2) return (row[i >> 2] >> ((i & 3u) * 2)) & 3u;
  const uint8_t* t = row + (i >> 3) * 3;
  const uint32_t acc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16;
  return (acc >> ((i & 7u) * 3)) & 7u;
}
template <int BITS>
__host__ __device__ constexpr uint32_t tq_row_bytes(uint32_t d) { return BITS == 2 ? d / 4 : d / 8 * 3; }

// In-place Walsh-Hadamard butterflies over `rows` vectors of D floats in LDS (rotation.rs:113-130): stage `half` combines (i, i +
// half) exactly as the reference's loop does; all threads of the workgroup take part (barriers inside).
template <int D>
__device__ __forceinline__ void tq_fwht_rows(float* buf, uint32_t rows) {
  for (uint32_t half = 1; half < (uint32_t)D; half <<= 1) {
    __syncthreads();
    for (uint32_t j = threadIdx.x; j < rows * (D / 2); j += blockDim.x) {
      const uint32_t r = j / (D / 2), jj = j % (D / 2);
      const uint32_t i = (jj / half) * 2 * half + (jj % half);
      float* v = buf + r * D;
      const float a = v[i], b = v[i + half];
      v[i] = a + b;
      v[i + half] = a - b;
    }
  }
  __syncthreads();
}

// The same butterflies for ONE vector per wave, in registers: lane l holds elements l (and l + 64 when D = 128).  Stage `half`
// pairs (i, i + half), i without bit `half`: new[i] = a + b, new[i + half] = a - b with a the lower element — the partner comes
// through a cross-lane exchange, the operations and their operands are the reference's (rotation.rs:113-130): bit-identical to
// tq_fwht_rows, without its barrier per stage.
template <int D>
__device__ __forceinline__ void tq_fwht_wave(float& x0, float& x1, uint32_t lane) {
#pragma unroll
  for (uint32_t half = 1; half < 64 && half < (uint32_t)D; half <<= 1) {
    const bool upper = (lane & half) != 0;
    const float p0 = __shfl_xor(x0, (int)half, 64);
    x0 = upper ? p0 - x0 : x0 + p0;
    if (D == 128) {
      const float p1 = __shfl_xor(x1, (int)half, 64);
      x1 = upper ? p1 - x1 : x1 + p1;
    }
  }
  if (D == 128) {   // half = 64: the lane's own two elements
    const float a = x0, b = x1;
    x0 = a + b;
    x1 = a - b;
  }
}

// codes of the D floats at `y` (already rotated) -> row bytes, by the first D / 4 (2 bits) or D / 8 (3 bits) threads
template <int D, int BITS>
__device__ __forceinline__ void tq_pack_row(const TqTables& T, const float* y, uint8_t* out) {
  if (BITS == 2) {
    if (threadIdx.x < D / 4) {
      uint32_t byte = 0;
#pragma unroll
      for (int i = 0; i < 4; i++) byte |= tq_quantize<2>(T, y[threadIdx.x * 4 + i]) << (i * 2);
      out[threadIdx.x] = (uint8_t)byte;
    }
  } else if (threadIdx.x < D / 8) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) acc |= tq_quantize<3>(T, y[threadIdx.x * 8 + i]) << (i * 3);
    out[threadIdx.x * 3] = (uint8_t)(acc & 0xFF);
    out[threadIdx.x * 3 + 1] = (uint8_t)((acc >> 8) & 0xFF);
    out[threadIdx.x * 3 + 2] = (uint8_t)((acc >> 16) & 0xFF);
  }
}

// QJL side of a new K row and of `nq` rotated queries (all 256 threads; LDS in, LDS out):
//   res = y - dequantized codes of y (quant.rs:85-89);  newx = sign bits of S res, then |res| (QjlProjector::compress, qjl.rs:36-62);
//   pq[r] = S xq[r] (project_query, qjl.rs:100-114).  The residual's dot products are the reference's loop (dot += s_ij * x[j] in
//   order, unfused: the stored sign bits must be its bits); the projected queries only enter the scores and are summed in parallel.
template <int D, int BITS>
__device__ __forceinline__ void tq_qjl_rows(const TqTables& T, const float* __restrict__ S, uint32_t nq, const float* xq, const float* y,
                                            const uint8_t* codes, float* res, float* pq, uint32_t* newx) {
  const uint32_t tid = threadIdx.x, lane = tid & 63;
  if (tid < D) res[tid] = y[tid] - tq_centroid<BITS>(T, tq_index<BITS>(codes, tid));
  __syncthreads();
  // residual row, threads [0, D): dot += s_ij * x[j] in order, unfused — the stored sign bits are the reference's bit for bit
  if (tid < D) {
    const uint32_t i = tid;
    const float4* srow = reinterpret_cast<const float4*>(S + (size_t)i * D);
    float dot = 0.0f;
#pragma unroll 8
    for (uint32_t j4 = 0; j4 < (uint32_t)D / 4; j4++) {   // (8 row loads in flight: enough to cover L2, no register blow-up)
      const float4 s4 = srow[j4];
      dot += s4.x * res[4 * j4];
      dot += s4.y * res[4 * j4 + 1];
      dot += s4.z * res[4 * j4 + 2];
      dot += s4.w * res[4 * j4 + 3];
    }
    const unsigned long long bl = __ballot(dot >= 0.0f);                // bit (i % 64) of word i / 64 (qjl.rs:58-60)
    if (lane == 0) { newx[(i >> 6) * 2] = (uint32_t)bl; newx[(i >> 6) * 2 + 1] = (uint32_t)(bl >> 32); }
  }
  // projected queries (only the scores use them, to rounding): four lanes share a dot product, 32 terms each, summed by DPP
  for (uint32_t e = tid; e < nq * D * 4; e += 256) {
    const uint32_t part = e & 3u, ri = e >> 2, r = ri / D, i = ri % D;
    const float4* srow = reinterpret_cast<const float4*>(S + (size_t)i * D + part * (D / 4));
    const float* x = xq + r * D + part * (D / 4);
    float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
    for (uint32_t j4 = 0; j4 < (uint32_t)D / 16; j4 += 2) {
      const float4 a4 = srow[j4], b4 = srow[j4 + 1];
      d0 += a4.x * x[4 * j4] + a4.y * x[4 * j4 + 1] + a4.z * x[4 * j4 + 2] + a4.w * x[4 * j4 + 3];
      d1 += b4.x * x[4 * j4 + 4] + b4.y * x[4 * j4 + 5] + b4.z * x[4 * j4 + 6] + b4.w * x[4 * j4 + 7];
    }
    float dot = d0 + d1;
    dot += dpp_f<0xB1>(dot);
    dot += dpp_f<0x4E>(dot);
    if (part == 0) pq[r * D + i] = dot;
  }
  if (tid == 255) {                                                    // l2_norm (qjl.rs:180-182): sequential sum of squares
    float ss = 0.0f;
    for (uint32_t i = 0; i < (uint32_t)D; i++) ss += res[i] * res[i];
    newx[D / 32] = __float_as_uint(sqrtf(ss));
  }
  __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// split attention over the code caches.  grid = n_kv * n_splits, 256 threads.
// signs: this layer's [kv head][k, v][D]; k_new / v_new: the current token's f32 rows [kv head][D]
// dynamic LDS: (G + 2) * D floats (rotated queries, the current K row, the current V row) [QJL: + (G + 1) * D floats: the projected
//              queries S (H D q) and the residual of the current K row] + 2 * row bytes (its codes, 16-aligned) + 32 bytes (its QJL
//              row) + G * cap floats (scores of this split's positions) + 64 floats of reduction scratch
//
// QJL = KVCacheType::TurboQuantProd (`tq2-qjl | tq3-qjl`; src/model/turboquant/qjl.rs, quant.rs:71-168): a K row additionally keeps
// the sign bits of S r — r = rotated row - its dequantized codes, S the engine's d x d Gaussian projection (an INPUT of the library,
// lgh_set_kv_qjl_matrices: [kv head][D][D] for this layer) — and |r|; its score is the codes' dot product PLUS
// sqrt(pi / 2) / d * |r| * sum_i (S H D q)_i * sign_i.  Bits, norms and projected queries are computed with the reference's own
// sequential unfused sums (qjl.rs:44-57, 104-111, 180-182), so the stored rows are BIT-EXACT; the sign-weighted sum runs in index
// order (qjl.rs:151-178 — the reference's host-SIMD variants, backend/cpu/simd.rs:1361-1374, differ from it by rounding).
// kx: [kv head][max_seq][D / 32 + 1] words (bits, then the norm).  V rows: codes only — the reference computes QJL bits for V rows
// too but never reads them (kv_turboquant.rs:154-170).
// ------------------------------------------------------------------------------------------------
// MULTI (multi-sequence decode, engine_batch.hip): blockIdx.y = the sequence; its position is pos_ptr[y], its caches are slot
// slot_ptr[y] of kq / vq / kx (ms.code_stride bytes / ms.x_stride words apart), its query, staging rows and partials follow at the
// strides in `ms`.  Same arithmetic, same order: a sequence's partials are the single-sequence launch's bit for bit.
struct TqMulti { const int* slot; uint64_t code_stride, x_stride; uint32_t kv_stride; };   // kv_stride: floats between two sequences' (K row, V row) staging

template <int D, int G, int BITS, bool QJL, bool MULTI>
__global__ void __launch_bounds__(256) attn_tq_partial_kernel(const float* __restrict__ q, uint8_t* __restrict__ kq, uint8_t* __restrict__ vq,
                                                              const float* __restrict__ k_new, const float* __restrict__ v_new,
                                                              const float* __restrict__ signs, const TqTables T, uint32_t max_seq, float scale,
                                                              const int* pos_ptr, uint32_t n_splits, uint32_t cap,
                                                              float* __restrict__ part_ml, float* __restrict__ part_acc,
                                                              const float* __restrict__ qjl_s, uint32_t* __restrict__ kx, const TqMulti ms) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  constexpr uint32_t RB = tq_row_bytes<BITS>(D);
  constexpr uint32_t NROT = QJL ? 2 * G + 3 : G + 2;
  constexpr uint32_t XW = D / 32 + 1;                                // words of a QJL row
  constexpr int NW = 4;                                              // waves
  constexpr int LPR = D / 16;                                        // lanes per cached row: 16 coordinates each
  constexpr int RPW = 64 / LPR;                                      // rows per wave-instruction
  float* rot = reinterpret_cast<float*>(smem);                       // [G + 2][D] (+ QJL: pq[G][D], res[D])
  float* pq = rot + (G + 2) * D;
  float* res = pq + G * D;
  uint8_t* newk = smem + NROT * D * 4;                               // [RB] (padded to 64)
  uint8_t* newv = newk + 64;
  uint32_t* newx = reinterpret_cast<uint32_t*>(newv + 64);           // [XW] (padded to 32 bytes)
  float* s_ml = reinterpret_cast<float*>(newv + 64 + 32);            // [NW][G][2]
  float* s_acc = s_ml + NW * G * 2;                                  // [NW][G][D]
  (void)cap;
  const uint32_t kvh = blockIdx.x / n_splits, sp = blockIdx.x % n_splits;
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t pw;
  if (MULTI) {
    const uint32_t sq = blockIdx.y;
    uint32_t sl;
    asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(pw), "=&s"(sl) : "s"(pos_ptr + sq), "s"(ms.slot + sq) : "memory");
    q += (size_t)sq * (gridDim.x / n_splits) * G * D;
    k_new += (size_t)sq * ms.kv_stride;
    v_new += (size_t)sq * ms.kv_stride;
    kq += (size_t)sl * ms.code_stride;
    vq += (size_t)sl * ms.code_stride;
    if (QJL) kx += (size_t)sl * ms.x_stride;
    part_ml += (size_t)sq * gridDim.x * G * 2;
    part_acc += (size_t)sq * gridDim.x * G * D;
  } else {
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pw) : "s"(pos_ptr) : "memory");
  }
  const uint32_t pos = pw, kv_len = pw + 1;
  const float* sk = signs + (size_t)(kvh * 2) * D;
  const float* sv = sk + D;
  uint8_t* krow0 = kq + (size_t)kvh * max_seq * RB;
  uint8_t* vrow0 = vq + (size_t)kvh * max_seq * RB;
  uint32_t* xrow0 = QJL ? kx + (size_t)kvh * max_seq * XW : nullptr;
  const uint32_t sub = lane / LPR, li = lane % LPR;
  // 16 codes of a row for this lane: bits [0, 16 * BITS) of the returned word(s)
  auto load_codes = [&](const uint8_t* row) -> unsigned long long {
    if (BITS == 2) return *reinterpret_cast<const uint32_t*>(row + li * 4);
    const uint32_t off = li * 6, al = off & ~3u;                        // 6 bytes at a 2-byte aligned offset of a 48-byte row
    const unsigned long long w = (unsigned long long)*reinterpret_cast<const uint32_t*>(row + al) |
                                 (unsigned long long)*reinterpret_cast<const uint32_t*>(row + (al + 4 < RB ? al + 4 : al)) << 32;
    return w >> ((off & 3u) * 8);
  };
  const uint32_t stride = n_splits * NW * RPW;
  auto row_of = [&](uint32_t base) { const uint32_t p = base + sub; return p < kv_len ? p : kv_len - 1; };
  auto fetch_mem = [&](uint32_t r, unsigned long long& kc, unsigned long long& vc, uint32_t& sgn, float& rnorm) {
    kc = load_codes(krow0 + (size_t)r * RB);
    vc = load_codes(vrow0 + (size_t)r * RB);
    sgn = 0; rnorm = 0.0f;
    if (QJL) {
      const uint32_t* xr = xrow0 + (size_t)r * XW;
      sgn = (xr[li >> 1] >> ((li & 1u) * 16)) & 0xFFFFu;               // sign bits of coordinates 16 li .. 16 li + 15
      rnorm = __uint_as_float(xr[D / 32]);
    }
  };
  auto fetch_lds = [&](unsigned long long& kc, unsigned long long& vc, uint32_t& sgn, float& rnorm) {   // the row this launch writes
    kc = load_codes(newk);
    vc = load_codes(newv);
    sgn = 0; rnorm = 0.0f;
    if (QJL) { sgn = (newx[li >> 1] >> ((li & 1u) * 16)) & 0xFFFFu; rnorm = __uint_as_float(newx[D / 32]); }
  };
  // the first rows of this wave are requested NOW, ahead of the rotations (row `pos` holds stale bytes until this launch has
  // written it: read anyway — it lies inside the cache — and replaced from LDS below)
  constexpr int kAhead = 2;
  const uint32_t base0 = (sp * NW + wave) * RPW;
  unsigned long long kk0[kAhead], vv0[kAhead];
  uint32_t sg0[kAhead];
  float rn0[kAhead];
#pragma unroll
  for (int j = 0; j < kAhead; j++) fetch_mem(row_of(base0 + j * stride), kk0[j], vv0[j], sg0[j], rn0[j]);
  // ---- D x (sign flip), then H, then 1 / sqrt(d): the G query heads with the K engine's signs, the new K row, the new V row
  // (a wave per vector, butterflies in registers: one barrier for all G + 2 vectors instead of one per stage)
  for (uint32_t r = wave; r < (uint32_t)(G + 2); r += NW) {
    const float* src = r < (uint32_t)G ? q + ((size_t)kvh * G + r) * D : r == (uint32_t)G ? k_new + (size_t)kvh * D : v_new + (size_t)kvh * D;
    const float* sg = r <= (uint32_t)G ? sk : sv;
    float x0 = src[lane] * sg[lane], x1 = 0.0f;
    if (D == 128) x1 = src[lane + 64] * sg[lane + 64];
    tq_fwht_wave<D>(x0, x1, lane);
    rot[r * D + lane] = x0 * T.norm;
    if (D == 128) rot[r * D + lane + 64] = x1 * T.norm;
  }
  __syncthreads();
  tq_pack_row<D, BITS>(T, rot + G * D, newk);
  tq_pack_row<D, BITS>(T, rot + (G + 1) * D, newv);
  __syncthreads();
  if (sp == 0 && tid < RB) { krow0[(size_t)pos * RB + tid] = newk[tid]; vrow0[(size_t)pos * RB + tid] = newv[tid]; }
  if (QJL) {
    tq_qjl_rows<D, BITS>(T, qjl_s + (size_t)kvh * D * D, G, rot, rot + G * D, newk, res, pq, newx);
    if (sp == 0 && tid < XW) xrow0[(size_t)pos * XW + tid] = newx[tid];
  }
  // ---- one pass over this split's positions, online softmax (the structure of the f32 cache's attn_partial_kernel): LPR lanes share
  // a cached row, 16 coordinates each — their codes are 4 bytes (2 bits) or 6 bytes (3 bits) of the row, decoded to centroids in
  // registers; a lane's partial dot products are summed over the row's lanes by DPP.  (The reference sums a row's 128 products one
  // after the other, codebook.rs:228-248; this sum is the same products in another order.)  V: sum_p w_p c[V_p] in the rotated space.
  float m[G], l[G], acc[G][16];
#pragma unroll
  for (int g = 0; g < G; g++) {
    m[g] = -1e30f; l[g] = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; c++) acc[g][c] = 0.0f;
  }
  auto step = [&](uint32_t base, unsigned long long kc, unsigned long long vc, uint32_t sgn, float rnorm) {
    const bool valid = base + sub < kv_len;
    // (8 heads per kv head: the rotated / projected queries stay in LDS — hoisted into registers beside the 128 accumulators they spill)
    const float* rotv = rot;
    const float* pqv = pq;
    if (G >= 8) asm volatile("" : "+v"(rotv), "+v"(pqv));
    float s[G];
#pragma unroll
    for (int g = 0; g < G; g++) s[g] = 0.0f;
    float tq_[G];
#pragma unroll
    for (int g = 0; g < G; g++) tq_[g] = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const float cv = tq_centroid<BITS>(T, (uint32_t)(kc >> (c * BITS)) & ((1u << BITS) - 1u));
#pragma unroll
      for (int g = 0; g < G; g++) s[g] = __builtin_fmaf(rotv[g * D + li * 16 + c], cv, s[g]);
      if (QJL) {
        const bool plus = (sgn >> c) & 1u;
#pragma unroll
        for (int g = 0; g < G; g++) { const float v = pqv[g * D + li * 16 + c]; tq_[g] += plus ? v : -v; }
      }
    }
    const float cn = QJL ? T.qjl_coeff * rnorm : 0.0f;                  // coeff * key_norm (qjl.rs:120-131)
#pragma unroll
    for (int g = 0; g < G; g++) {
      float sg = QJL ? s[g] + cn * tq_[g] : s[g];                       // polar_score + correction, this lane's share
      sg += dpp_f<0xB1>(sg);
      sg += dpp_f<0x4E>(sg);
      if (LPR >= 8) sg += dpp_f<0x141>(sg);
      sg *= scale;
      const float mn = valid ? fmaxf(m[g], sg) : m[g];
      const float a = __expf(m[g] - mn);
      const float pe = valid ? __expf(sg - mn) : 0.0f;
      l[g] = __builtin_fmaf(l[g], a, pe);
      m[g] = mn;
      s[g] = pe;
      tq_[g] = a;
    }
#pragma unroll
    for (int c = 0; c < 16; c++) {
      const float cv = tq_centroid<BITS>(T, (uint32_t)(vc >> (c * BITS)) & ((1u << BITS) - 1u));
#pragma unroll
      for (int g = 0; g < G; g++) acc[g][c] = __builtin_fmaf(acc[g][c], tq_[g], cv * s[g]);
    }
  };
  for (uint32_t base = base0; base < kv_len; base += kAhead * stride) {
    unsigned long long kk[kAhead], vv[kAhead];
    uint32_t sg[kAhead];
    float rn[kAhead];
    if (base == base0) {   // requested at kernel start; the row written by this launch comes from LDS
#pragma unroll
      for (int j = 0; j < kAhead; j++) {
        kk[j] = kk0[j]; vv[j] = vv0[j]; sg[j] = sg0[j]; rn[j] = rn0[j];
        if (row_of(base + j * stride) == pos) fetch_lds(kk[j], vv[j], sg[j], rn[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < kAhead; j++) {
        const uint32_t r = row_of(base + j * stride);
        if (r == pos) fetch_lds(kk[j], vv[j], sg[j], rn[j]);
        else fetch_mem(r, kk[j], vv[j], sg[j], rn[j]);
      }
    }
#pragma unroll
    for (int j = 0; j < kAhead; j++)
      if (base + j * stride < kv_len) step(base + j * stride, kk[j], vv[j], sg[j], rn[j]);   // wave-uniform
  }
  // merge the RPW row slots of the wave (lanes with equal li hold the same coordinates)
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      const float om = __shfl_xor(m[g], off, 64), ol = __shfl_xor(l[g], off, 64);
      const float mn = fmaxf(m[g], om);
      const float a = expf(m[g] - mn), b = expf(om - mn);
      l[g] = l[g] * a + ol * b;
#pragma unroll
      for (int c = 0; c < 16; c++) acc[g][c] = acc[g][c] * a + __shfl_xor(acc[g][c], off, 64) * b;
      m[g] = mn;
    }
  }
  if (sub == 0) {
#pragma unroll
    for (int g = 0; g < G; g++) {
      if (li == 0) { s_ml[(wave * G + g) * 2] = m[g]; s_ml[(wave * G + g) * 2 + 1] = l[g]; }
#pragma unroll
      for (int c = 0; c < 16; c++) s_acc[((size_t)wave * G + g) * D + li * 16 + c] = acc[g][c];
    }
  }
  __syncthreads();
  // merge the waves; thread t handles output elements t, t + 256, ...
  const size_t pbase = ((size_t)kvh * n_splits + sp) * G;
  for (uint32_t e = tid; e < (uint32_t)(G * D); e += 256) {
    const uint32_t g = e / D, dim = e % D;
    float mn = s_ml[g * 2];
#pragma unroll
    for (int w = 1; w < NW; w++) mn = fmaxf(mn, s_ml[(w * G + g) * 2]);
    float lsum = 0.0f, a = 0.0f;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const float f = expf(s_ml[(w * G + g) * 2] - mn);
      lsum += s_ml[(w * G + g) * 2 + 1] * f;
      a += s_acc[((size_t)w * G + g) * D + dim] * f;
    }
    part_acc[(pbase + g) * D + dim] = a;
    if (dim == 0) { part_ml[(pbase + g) * 2] = mn; part_ml[(pbase + g) * 2 + 1] = lsum; }
  }
}

// merge of the splits (as attn_combine_kernel) + the inverse rotation (rotation.rs:80-96): out = signs * (1 / d) * H (sqrt(d) * o)
// MULTI: blockIdx.y = the sequence (partials of gridDim.x heads per sequence; outputs n_heads * D floats / xq_stride bytes apart)
template <int D, bool MULTI>
__global__ void __launch_bounds__(D) attn_tq_combine_kernel(const float* __restrict__ part_ml, const float* __restrict__ part_acc,
                                                            const float* __restrict__ signs, const TqTables T, uint32_t g_per_kv,
                                                            uint32_t n_splits, float* __restrict__ out, uint8_t* __restrict__ xq_out,
                                                            uint32_t xq_stride) {
  if (MULTI) {
    const size_t per_seq = (size_t)gridDim.x * n_splits;   // (head, split) partials of one sequence
    part_ml += (size_t)blockIdx.y * per_seq * 2;
    part_acc += (size_t)blockIdx.y * per_seq * D;
    out += (size_t)blockIdx.y * gridDim.x * D;
    if (xq_out) xq_out += (size_t)blockIdx.y * xq_stride;
  }
  __shared__ float s_f[64];
  __shared__ float s_linv;
  __shared__ float buf[D];
  const uint32_t h = blockIdx.x, kvh = h / g_per_kv, g = h % g_per_kv;
  const size_t p0 = (size_t)kvh * n_splits * g_per_kv + g;
  const uint32_t dim = threadIdx.x;
  float pa[32];
#pragma unroll
  for (uint32_t s = 0; s < 32; s++) pa[s] = s < n_splits ? part_acc[(p0 + (size_t)s * g_per_kv) * D + dim] : 0.0f;
  if (threadIdx.x < 64) {
    const uint32_t s = threadIdx.x;
    const bool ok = s < n_splits;
    const float m = ok ? part_ml[(p0 + (size_t)s * g_per_kv) * 2] : -1e30f;
    const float l = ok ? part_ml[(p0 + (size_t)s * g_per_kv) * 2 + 1] : 0.0f;
    const float mn = wave_max(m);
    const float f = expf(m - mn);
    const float lsum = wave_sum(l * f);
    s_f[s] = f;
    if (s == 0) s_linv = 1.0f / lsum;
  }
  __syncthreads();
  float a = 0.0f;
#pragma unroll
  for (uint32_t s = 0; s < 32; s++) a += pa[s] * s_f[s];
  buf[dim] = (a * s_linv) * T.inv_scale;                              // x * sqrt(d)
  tq_fwht_rows<D>(buf, 1);
  const float o = buf[dim] * T.inv_d * signs[(size_t)(kvh * 2 + 1) * D + dim];
  out[(size_t)h * D + dim] = o;
  if (xq_out) xq_store_chunk(xq_out, (h * D + dim) >> 4, o);
}

// one row through the compressor (the code path of attn_tq_partial_kernel's new-row handling, stand-alone): lgh_op_tq_compress
template <int D, int BITS>
__global__ void __launch_bounds__(256) tq_compress_kernel(const float* __restrict__ x, const float* __restrict__ signs, const TqTables T,
                                                          uint8_t* __restrict__ out, const float* __restrict__ qjl_s, uint32_t* __restrict__ qjl_out) {
  __shared__ float rot[D];
  __shared__ float res[D];
  __shared__ __attribute__((aligned(16))) uint8_t code[64];
  __shared__ uint32_t newx[8];
  for (uint32_t i = threadIdx.x; i < D; i += 256) rot[i] = x[i] * signs[i];
  tq_fwht_rows<D>(rot, 1);
  for (uint32_t i = threadIdx.x; i < D; i += 256) rot[i] *= T.norm;
  __syncthreads();
  tq_pack_row<D, BITS>(T, rot, code);
  __syncthreads();
  if (threadIdx.x < tq_row_bytes<BITS>(D)) out[threadIdx.x] = code[threadIdx.x];
  if (qjl_s) {   // TurboQuantEngine::compress with use_qjl (quant.rs:83-97): D / 32 words of sign bits, then the residual norm
    tq_qjl_rows<D, BITS>(T, qjl_s, 0, nullptr, rot, code, res, nullptr, newx);
    if (threadIdx.x < D / 32 + 1) qjl_out[threadIdx.x] = newx[threadIdx.x];
  }
}

// ------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------
static TqTables tq_tables(uint32_t d, int bits) {
  // Codebook::new (codebook.rs:55-77) and the rotation's factors (rotation.rs:72, 84, 91) in host f32 arithmetic
  static const float l2[4] = {-1.5102326f, -0.4528427f, 0.4528427f, 1.5102326f};
  static const float l3[8] = {-2.1521645f, -1.3441838f, -0.7561303f, -0.2453404f, 0.2453404f, 0.7561303f, 1.3441838f, 2.1521645f};
  static const float b2[3] = {-0.98153765f, 0.0f, 0.98153765f};
  static const float b3[7] = {-1.74817415f, -1.05015705f, -0.50073535f, 0.0f, 0.50073535f, 1.05015705f, 1.74817415f};
  TqTables T{};
  const float inv_sqrt_d = 1.0f / std::sqrt((float)d);
  for (int i = 0; i < (1 << bits); i++) T.cen[i] = (bits == 2 ? l2[i] : l3[i]) * inv_sqrt_d;
  for (int i = 0; i < (1 << bits) - 1; i++) T.bnd[i] = (bits == 2 ? b2[i] : b3[i]) * inv_sqrt_d;
  T.norm = 1.0f / std::sqrt((float)d);
  T.inv_scale = std::sqrt((float)d);
  T.inv_d = 1.0f / (float)d;
  T.qjl_coeff = std::sqrt(1.57079632679489661923f) / (float)d;   // FRAC_PI_2.sqrt() / dim (qjl.rs:70, 128)
  return T;
}

uint32_t tq_row_bytes_host(int bits, uint32_t d) { return bits == 2 ? d / 4 : d / 8 * 3; }
// positions one split can hold scores for: max_seq spread over the splits
uint32_t tq_split_cap(uint32_t max_seq, uint32_t n_splits) { return (max_seq + n_splits - 1) / n_splits; }

struct TqArgs {
  const float* q; uint8_t *kq, *vq; const float *k_new, *v_new, *signs; uint32_t n_kv, max_seq; float scale; const int* pos; uint32_t n_splits;
  float *part_ml, *part_acc; const float* qjl_s; uint32_t* kx; hipStream_t st;
  uint32_t n_seq; TqMulti ms;   // n_seq 0: the single-sequence launch
};

template <int D, int G, int BITS, bool QJL>
static hipError_t attn_tq_go(const TqArgs& a) {
  static bool attr_set[2][64] = {};
  const uint32_t cap = tq_split_cap(a.max_seq, a.n_splits);
  const size_t lds = (size_t)(QJL ? 2 * G + 3 : G + 2) * D * 4 + 128 + 32 + (size_t)4 * G * 2 * 4 + (size_t)4 * G * D * 4 + 64;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  if (a.n_seq) {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&attn_tq_partial_kernel<D, G, BITS, QJL, true>), 160 * 1024, attr_set[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_tq_partial_kernel<D, G, BITS, QJL, true>), dim3(a.n_kv * a.n_splits, a.n_seq), dim3(256), lds, a.st, a.q, a.kq, a.vq, a.k_new,
                       a.v_new, a.signs, tq_tables(D, BITS), a.max_seq, a.scale, a.pos, a.n_splits, cap, a.part_ml, a.part_acc, a.qjl_s, a.kx, a.ms);
  } else {
    if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&attn_tq_partial_kernel<D, G, BITS, QJL, false>), 160 * 1024, attr_set[0]); e != hipSuccess) return e;
    hipLaunchKernelGGL((attn_tq_partial_kernel<D, G, BITS, QJL, false>), dim3(a.n_kv * a.n_splits), dim3(256), lds, a.st, a.q, a.kq, a.vq, a.k_new, a.v_new,
                       a.signs, tq_tables(D, BITS), a.max_seq, a.scale, a.pos, a.n_splits, cap, a.part_ml, a.part_acc, a.qjl_s, a.kx, a.ms);
  }
  return hipGetLastError();
}

static hipError_t attn_tq_dispatch(int bits, uint32_t head_dim, uint32_t g, const TqArgs& a) {
#define LGH_TQ_CASE(DD, GG)                                                                              \
  if (head_dim == DD && g == GG) {                                                                       \
    if (a.qjl_s) return bits == 2 ? attn_tq_go<DD, GG, 2, true>(a) : attn_tq_go<DD, GG, 3, true>(a);     \
    return bits == 2 ? attn_tq_go<DD, GG, 2, false>(a) : attn_tq_go<DD, GG, 3, false>(a);                \
  }
  LGH_TQ_CASE(128, 1) LGH_TQ_CASE(128, 2) LGH_TQ_CASE(128, 4) LGH_TQ_CASE(128, 8)
  LGH_TQ_CASE(64, 1) LGH_TQ_CASE(64, 2) LGH_TQ_CASE(64, 4) LGH_TQ_CASE(64, 8)
#undef LGH_TQ_CASE
  return hipErrorInvalidValue;
}

// bits 2 / 3; signs: this layer's [n_kv][2][head_dim]; qjl_s (TurboQuantProd; NULL = TurboQuantMSE): this layer's [n_kv][head_dim][head_dim],
// kx: its QJL rows [n_kv][max_seq][head_dim / 32 + 1]
hipError_t attn_tq_launch(int bits, const float* q, uint8_t* kq, uint8_t* vq, const float* k_new, const float* v_new, const float* signs,
                          uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, uint32_t n_splits,
                          float* part_ml, float* part_acc, hipStream_t st, const float* qjl_s, uint32_t* kx) {
  if (n_kv == 0 || n_heads % n_kv || !pos || (bits != 2 && bits != 3) || n_splits == 0 || n_splits > 32 || (qjl_s != nullptr) != (kx != nullptr))
    return hipErrorInvalidValue;
  const TqArgs a{q, kq, vq, k_new, v_new, signs, n_kv, max_seq, scale, pos, n_splits, part_ml, part_acc, qjl_s, kx, st, 0, TqMulti{}};
  return attn_tq_dispatch(bits, head_dim, n_heads / n_kv, a);
}

// the same for n_seq sequences (grid y): positions pos[s], cache slots slot[s] (code_stride bytes / x_stride words between slots),
// queries n_heads * head_dim floats apart, the staging (K row, V row) pairs kv_stride floats apart
hipError_t attn_tq_multi_launch(int bits, const float* q, uint8_t* kq, uint8_t* vq, const float* k_new, const float* v_new, const float* signs,
                                uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, const int* slot,
                                uint64_t code_stride, uint64_t x_stride, uint32_t kv_stride, uint32_t n_seq, uint32_t n_splits, float* part_ml,
                                float* part_acc, hipStream_t st, const float* qjl_s, uint32_t* kx) {
  if (n_kv == 0 || n_heads % n_kv || !pos || !slot || n_seq == 0 || (bits != 2 && bits != 3) || n_splits == 0 || n_splits > 32 ||
      (qjl_s != nullptr) != (kx != nullptr))
    return hipErrorInvalidValue;
  const TqArgs a{q, kq, vq, k_new, v_new, signs, n_kv, max_seq, scale, pos, n_splits, part_ml, part_acc, qjl_s, kx, st, n_seq,
                 TqMulti{slot, code_stride, x_stride, kv_stride}};
  return attn_tq_dispatch(bits, head_dim, n_heads / n_kv, a);
}

// n_seq 0: single sequence; otherwise grid y = sequences, XQ images xq_stride bytes apart
hipError_t attn_tq_combine_launch(int bits, const float* part_ml, const float* part_acc, const float* signs, uint32_t n_heads, uint32_t n_kv,
                                  uint32_t head_dim, uint32_t n_splits, float* out, uint8_t* xq_out, hipStream_t st, uint32_t n_seq,
                                  uint32_t xq_stride) {
  if (n_kv == 0 || n_heads % n_kv || n_splits > 32 || (head_dim != 64 && head_dim != 128)) return hipErrorInvalidValue;
  const TqTables T = tq_tables(head_dim, bits);
  const uint32_t gk = n_heads / n_kv;
  if (n_seq) {
    if (head_dim == 128) hipLaunchKernelGGL((attn_tq_combine_kernel<128, true>), dim3(n_heads, n_seq), dim3(128), 0, st, part_ml, part_acc, signs, T, gk, n_splits, out, xq_out, xq_stride);
    else hipLaunchKernelGGL((attn_tq_combine_kernel<64, true>), dim3(n_heads, n_seq), dim3(64), 0, st, part_ml, part_acc, signs, T, gk, n_splits, out, xq_out, xq_stride);
  } else {
    if (head_dim == 128) hipLaunchKernelGGL((attn_tq_combine_kernel<128, false>), dim3(n_heads), dim3(128), 0, st, part_ml, part_acc, signs, T, gk, n_splits, out, xq_out, 0u);
    else hipLaunchKernelGGL((attn_tq_combine_kernel<64, false>), dim3(n_heads), dim3(64), 0, st, part_ml, part_acc, signs, T, gk, n_splits, out, xq_out, 0u);
  }
  return hipGetLastError();
}

// x[dim] -> codes[row bytes] with the given sign vector (dim 64 or 128); qjl_s[dim][dim] (optional) -> qjl_out[dim / 32 + 1] words
hipError_t tq_compress_launch(int bits, const float* x, uint32_t dim, const float* signs, uint8_t* out, hipStream_t st, const float* qjl_s,
                              uint32_t* qjl_out) {
  if ((bits != 2 && bits != 3) || (dim != 64 && dim != 128) || (qjl_s != nullptr) != (qjl_out != nullptr)) return hipErrorInvalidValue;
  const TqTables T = tq_tables(dim, bits);
  if (dim == 128 && bits == 2) hipLaunchKernelGGL((tq_compress_kernel<128, 2>), dim3(1), dim3(256), 0, st, x, signs, T, out, qjl_s, qjl_out);
  else if (dim == 128) hipLaunchKernelGGL((tq_compress_kernel<128, 3>), dim3(1), dim3(256), 0, st, x, signs, T, out, qjl_s, qjl_out);
  else if (bits == 2) hipLaunchKernelGGL((tq_compress_kernel<64, 2>), dim3(1), dim3(256), 0, st, x, signs, T, out, qjl_s, qjl_out);
  else hipLaunchKernelGGL((tq_compress_kernel<64, 3>), dim3(1), dim3(256), 0, st, x, signs, T, out, qjl_s, qjl_out);
  return hipGetLastError();
}

}  // namespace lgh
