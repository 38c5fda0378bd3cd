// mvq_core.h — the tile-level core of the int8 matrix-core mat-vec (internal, device side): how one 16-row x 256-element
// weight tile is requested from HBM and how it is multiplied with the XQ record of its 256 input elements.  Shared by
// the launch-per-op kernel (matvec_mfma.hip: mvq_kernel) and the persistent token kernel (decode_persistent.hip), so both
// produce the same bits for the same tile.  Formats, tile layouts and the arithmetic are described in matvec_mfma.hip.
#pragma once

#include "device_utils.h"
#include "xq.h"

namespace lgh {

typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileBytes = 2304;      // Q4_K tile16
constexpr int kTileBytesQ6 = 3392;    // Q6_K tile16
// formats of the matrix-core mat-vec; a kernel instantiation handles the formats in its MASK (bit = 1 << format)
enum : int { F_Q4K = 0, F_Q6K = 1, F_Q5K = 2, F_Q80 = 3, F_Q40 = 4, F_COUNT = 5 };
__host__ __device__ constexpr uint32_t fmt_tile_bytes(int f) {   // = 16 rows x the GGUF bytes of 256 elements (Q6_K: +32 pad)
  return f == F_Q4K ? 2304u : f == F_Q6K ? 3392u : f == F_Q5K ? 2816u : f == F_Q80 ? 4352u : 2304u;
}
__host__ __device__ constexpr int fmt_loads_per_tile(int f) { return f == F_Q4K ? 3 : f == F_Q6K ? 5 : f == F_Q5K ? 4 : f == F_Q80 ? 5 : 3; }
__host__ __device__ constexpr int fmt_of_dev_type(int t) {
  return t == kDevQ4K_T16 ? F_Q4K : t == kDevQ6K_T16 ? F_Q6K : t == kDevQ5K_T16 ? F_Q5K : t == kDevQ80_T16 ? F_Q80 : t == kDevQ40_T16 ? F_Q40 : -1;
}


struct RawT16 { u32x4 hd; u32x4 q[4]; };   // one tile in flight: header / scales + up to four 16-byte payload loads per lane

// `fmt` is only read when MASK holds more than one format (otherwise the format is a compile-time constant)
template <uint32_t MASK>
__device__ __forceinline__ bool mvq_is(int fmt, int f) {
  constexpr bool kSingle = (MASK & (MASK - 1)) == 0;
  return ((MASK >> f) & 1u) != 0 && (kSingle || fmt == f);
}

// fmt_loads_per_tile(format) non-temporal loads of this lane's share of the tile at `tile` (lane roles: row n = lane & 15)
template <uint32_t MASK>
__device__ __forceinline__ void mvq_issue_tile(int fmt, const uint8_t* tile, uint32_t lane, RawT16& r) {
  const uint32_t n = lane & 15;
  auto is = [&](int f) { return mvq_is<MASK>(fmt, f); };
    if (is(F_Q6K)) {
      r.hd = ldg_nt128(tile + 3072 + n * 16);
#pragma unroll
      for (int i = 0; i < 3; i++) r.q[i] = ldg_nt128(tile + i * 1024 + lane * 16);
      r.q[3].x = ldg_nt32(tile + 3328 + (n >> 1) * 4);
    } else if (is(F_Q80)) {
      r.hd = ldg_nt128(tile + 4096 + n * 16);
#pragma unroll
      for (int i = 0; i < 4; i++) r.q[i] = ldg_nt128(tile + i * 1024 + lane * 16);
    } else {   // Q4_K, Q5_K, Q4_0: header + two nibble loads (+ the fifth bits)
      r.hd = ldg_nt128(tile + 2048 + n * 16);
      r.q[0] = ldg_nt128(tile + lane * 16);
      r.q[1] = ldg_nt128(tile + 1024 + lane * 16);
      if (is(F_Q5K)) {
        const u32x2 h = ldg_nt64(tile + 2304 + lane * 8);
        r.q[2].x = h.x;
        r.q[2].y = h.y;
      }
    }
}

// acc += (weight rows n of the tile) . (the 256 elements of XQ record `rec`, in LDS), this lane's k-chunk c = lane >> 4 only:
// four MFMAs (one per 64 elements); lane group c then holds, for weight row n, the four limb sums of chunk 4pp + c,
// recombined to V = sum_k q_k * I_k (exact int, rounded once to f32).
template <uint32_t MASK>
__device__ __forceinline__ void mvq_consume_tile(int fmt, const RawT16& r, const uint8_t* rec, uint32_t lane, float& acc) {
  const uint32_t n = lane & 15, c = lane >> 4;
  const bool a_valid = (n >> 2) == c;
  const uint32_t a_off = (c >> 1) * 128 + (n & 3) * 32 + (c & 1) * 16;   // + 2p * 128
  auto is = [&](int f) { return mvq_is<MASK>(fmt, f); };
    i32x4 areg[4];
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      i32x4 t = {0, 0, 0, 0};
      if (a_valid) t = *reinterpret_cast<const i32x4*>(rec + pp * 256 + a_off);
      areg[pp] = t;
    }
    const f32x4 xs = *reinterpret_cast<const f32x4*>(rec + kXqXs16 + c * 16);   // sum of x over chunk 4pp + c
    const f32x4 sx = *reinterpret_cast<const f32x4*>(rec + kXqSx16 + c * 16);   // its scale * 2^-30
    if (is(F_Q6K)) {
      const uint32_t s8 = c * 8;
      const uint32_t hdw[4] = {r.hd.x, r.hd.y, r.hd.z, r.hd.w};   // int8 scales 4pp .. 4pp+3 of row n (dequant.rs:343-350)
      float s1 = 0.0f;
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
        const uint32_t H = pp == 0 ? r.q[2].x : pp == 1 ? r.q[2].y : pp == 2 ? r.q[2].z : r.q[2].w;
        i32x4 bw;   // q' = ql | qh << 4, 0..63; the reference's "- 32" is applied through the chunk's sum of x below
        bw.x = (int)((N0 & 0x0F0F0F0Fu) | ((H & 0x03030303u) << 4));
        bw.y = (int)(((N0 >> 4) & 0x0F0F0F0Fu) | (((H >> 2) & 0x03030303u) << 4));
        bw.z = (int)((N1 & 0x0F0F0F0Fu) | (((H >> 4) & 0x03030303u) << 4));
        bw.w = (int)(((N1 >> 4) & 0x0F0F0F0Fu) | (((H >> 6) & 0x03030303u) << 4));
        const i32x4 zero = {0, 0, 0, 0};
        const i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(areg[pp], bw, zero, 0, 0, 0);
        // |D0| <= 64*63*16, so the high half is exact in f32; the low half may round at 2^-24 of a term that is 2^-16 of the sum
        const float hi = (float)((d.x << 8) + d.y), lo = (float)((d.z << 8) + d.w);
        const float V = __builtin_fmaf(hi, 65536.0f, lo);
        const float scf = (float)(int)__builtin_amdgcn_sbfe((int)hdw[pp], s8, 8);
        s1 = __builtin_fmaf(scf, __builtin_fmaf(sx[pp], V, -32.0f * xs[pp]), s1);
      }
      const uint32_t dh = (n & 1) ? r.q[3].x >> 16 : r.q[3].x & 0xFFFFu;
      acc = __builtin_fmaf(h2f(dh), s1, acc);
    } else if (is(F_Q80) || is(F_Q40)) {
      // 32-element blocks with one f16 scale: chunk 4pp + c lies in block 2pp + (c >> 1) of the row's eight
      const uint32_t hdw[4] = {r.hd.x, r.hd.y, r.hd.z, r.hd.w};
      float s1 = 0.0f;
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        i32x4 bw;
        if (is(F_Q80)) {
          bw.x = (int)r.q[pp].x; bw.y = (int)r.q[pp].y; bw.z = (int)r.q[pp].z; bw.w = (int)r.q[pp].w;   // int8 quants as they are
        } else {
          const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
          bw.x = (int)(N0 & 0x0F0F0F0Fu);
          bw.y = (int)((N0 >> 4) & 0x0F0F0F0Fu);
          bw.z = (int)(N1 & 0x0F0F0F0Fu);
          bw.w = (int)((N1 >> 4) & 0x0F0F0F0Fu);
        }
        const i32x4 zero = {0, 0, 0, 0};
        const i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(areg[pp], bw, zero, 0, 0, 0);
        const float hi = (float)((d.x << 8) + d.y), lo = (float)((d.z << 8) + d.w);
        const float V = __builtin_fmaf(hi, 65536.0f, lo);
        const float dd = h2f((c >> 1) ? hdw[pp] >> 16 : hdw[pp] & 0xFFFFu);
        // Q4_0: y = d * (q - 8) (dequant.rs:16-30); the offset goes through the chunk's sum of x
        s1 = __builtin_fmaf(dd, is(F_Q40) ? __builtin_fmaf(sx[pp], V, -8.0f * xs[pp]) : sx[pp] * V, s1);
      }
      acc += s1;
    } else {
      // Q4_K / Q5_K: 6-bit scales / mins of sub-blocks h, h+2, h+4, h+6 of row n, h = c >> 1 (packing: dequant.rs:210-223);
      // chunk 4pp + c lies in sub-block 2pp + h
      const uint32_t s8 = (c >> 1) * 8;
      const uint32_t a = (r.hd.y >> s8) & 0x00FF00FFu, bq = (r.hd.z >> s8) & 0x00FF00FFu, cq = (r.hd.w >> s8) & 0x00FF00FFu;
      const uint32_t sc01 = a & 0x003F003Fu, mn01 = bq & 0x003F003Fu;
      const uint32_t sc23 = (cq & 0x000F000Fu) | ((a >> 2) & 0x00300030u);
      const uint32_t mn23 = ((cq >> 4) & 0x000F000Fu) | ((bq >> 2) & 0x00300030u);
      const float scf[4] = {ub0(sc01), ub2(sc01), ub0(sc23), ub2(sc23)};
      const float mnf[4] = {ub0(mn01), ub2(mn01), ub0(mn23), ub2(mn23)};
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
        i32x4 bw;
        bw.x = (int)(N0 & 0x0F0F0F0Fu);
        bw.y = (int)((N0 >> 4) & 0x0F0F0F0Fu);
        bw.z = (int)(N1 & 0x0F0F0F0Fu);
        bw.w = (int)((N1 >> 4) & 0x0F0F0F0Fu);
        if (is(F_Q5K)) {   // the fifth bit: dword of the step pair, low nibbles = even step, high nibbles = odd step
          const uint32_t H = ((pp >> 1) ? r.q[2].y : r.q[2].x) >> (4 * (pp & 1));
          bw.x |= (int)((H & 0x01010101u) << 4);
          bw.y |= (int)(((H >> 1) & 0x01010101u) << 4);
          bw.z |= (int)(((H >> 2) & 0x01010101u) << 4);
          bw.w |= (int)(((H >> 3) & 0x01010101u) << 4);
        }
        const i32x4 zero = {0, 0, 0, 0};
        const i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(areg[pp], bw, zero, 0, 0, 0);
        // both halves exact in f32: |D0| <= 64*31*16, |D1..3| <= 128*31*16
        const float hi = (float)((d.x << 8) + d.y), lo = (float)((d.z << 8) + d.w);
        const float V = __builtin_fmaf(hi, 65536.0f, lo);
        s1 = __builtin_fmaf(scf[pp] * sx[pp], V, s1);
        s2 = __builtin_fmaf(mnf[pp], xs[pp], s2);   // the reference's x_acc per sub-block (simd.rs:1002-1008), split per chunk
      }
      const float dd = h2f(r.hd.x & 0xFFFFu), dmin = h2f(r.hd.x >> 16);
      acc += dd * s1 - dmin * s2;
    }
}


// ------------------------------------------------------------------------------------------------
// The same arithmetic in two halves, for the multi-sequence mat-vec (matvec_batch.hip): a weight tile is unpacked ONCE
// (mvq_unpack_tile: the B operands of its four MFMAs + its scales) and then multiplied with the XQ operands of every
// sequence (mvq_mac_tile).  mvq_mac_tile(unpack(r), load(rec)) performs, per format, exactly the floating-point
// operations of mvq_consume_tile(r, rec) in the same order — that is what makes a batched step bit-identical to the
// single-sequence engine (tests/test_gpu_batch.py).
// ------------------------------------------------------------------------------------------------
struct XqOps { i32x4 a[4]; f32x4 xs, sx; };   // the A operands of a block's four MFMAs (zero outside the lane's own chunk), x sums, x scales

// this lane's operands out of one XQ record (LDS or global memory)
__device__ __forceinline__ void xq_load_ops(const uint8_t* rec, uint32_t lane, XqOps& o) {
  const uint32_t n = lane & 15, c = lane >> 4;
  const bool a_valid = (n >> 2) == c;
  const uint32_t a_off = (c >> 1) * 128 + (n & 3) * 32 + (c & 1) * 16;
#pragma unroll
  for (int pp = 0; pp < 4; pp++) {
    i32x4 t = {0, 0, 0, 0};
    if (a_valid) t = *reinterpret_cast<const i32x4*>(rec + pp * 256 + a_off);
    o.a[pp] = t;
  }
  o.xs = *reinterpret_cast<const f32x4*>(rec + kXqXs16 + c * 16);
  o.sx = *reinterpret_cast<const f32x4*>(rec + kXqSx16 + c * 16);
}

struct TileOps {
  i32x4 bw[4];      // B operand of step pp
  float f[4];       // per step: Q4_K/Q5_K sub-block scale, Q6_K int8 scale, Q8_0/Q4_0 block d
  float g[4];       // Q4_K/Q5_K: sub-block min
  float dd, dmin;   // Q4_K/Q5_K: d, dmin; Q6_K: d
};

template <uint32_t MASK>
__device__ __forceinline__ void mvq_unpack_tile(int fmt, const RawT16& r, uint32_t lane, TileOps& t) {
  const uint32_t n = lane & 15, c = lane >> 4;
  auto is = [&](int f) { return mvq_is<MASK>(fmt, f); };
  if (is(F_Q6K)) {
    const uint32_t s8 = c * 8;
    const uint32_t hdw[4] = {r.hd.x, r.hd.y, r.hd.z, r.hd.w};
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
      const uint32_t H = pp == 0 ? r.q[2].x : pp == 1 ? r.q[2].y : pp == 2 ? r.q[2].z : r.q[2].w;
      t.bw[pp].x = (int)((N0 & 0x0F0F0F0Fu) | ((H & 0x03030303u) << 4));
      t.bw[pp].y = (int)(((N0 >> 4) & 0x0F0F0F0Fu) | (((H >> 2) & 0x03030303u) << 4));
      t.bw[pp].z = (int)((N1 & 0x0F0F0F0Fu) | (((H >> 4) & 0x03030303u) << 4));
      t.bw[pp].w = (int)(((N1 >> 4) & 0x0F0F0F0Fu) | (((H >> 6) & 0x03030303u) << 4));
      t.f[pp] = (float)(int)__builtin_amdgcn_sbfe((int)hdw[pp], s8, 8);
      t.g[pp] = 0.0f;
    }
    const uint32_t dh = (n & 1) ? r.q[3].x >> 16 : r.q[3].x & 0xFFFFu;
    t.dd = h2f(dh);
    t.dmin = 0.0f;
  } else if (is(F_Q80) || is(F_Q40)) {
    const uint32_t hdw[4] = {r.hd.x, r.hd.y, r.hd.z, r.hd.w};
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      if (is(F_Q80)) {
        t.bw[pp].x = (int)r.q[pp].x; t.bw[pp].y = (int)r.q[pp].y; t.bw[pp].z = (int)r.q[pp].z; t.bw[pp].w = (int)r.q[pp].w;
      } else {
        const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
        t.bw[pp].x = (int)(N0 & 0x0F0F0F0Fu);
        t.bw[pp].y = (int)((N0 >> 4) & 0x0F0F0F0Fu);
        t.bw[pp].z = (int)(N1 & 0x0F0F0F0Fu);
        t.bw[pp].w = (int)((N1 >> 4) & 0x0F0F0F0Fu);
      }
      t.f[pp] = h2f((c >> 1) ? hdw[pp] >> 16 : hdw[pp] & 0xFFFFu);
      t.g[pp] = 0.0f;
    }
    t.dd = t.dmin = 0.0f;
  } else {
    const uint32_t s8 = (c >> 1) * 8;
    const uint32_t a = (r.hd.y >> s8) & 0x00FF00FFu, bq = (r.hd.z >> s8) & 0x00FF00FFu, cq = (r.hd.w >> s8) & 0x00FF00FFu;
    const uint32_t sc01 = a & 0x003F003Fu, mn01 = bq & 0x003F003Fu;
    const uint32_t sc23 = (cq & 0x000F000Fu) | ((a >> 2) & 0x00300030u);
    const uint32_t mn23 = ((cq >> 4) & 0x000F000Fu) | ((bq >> 2) & 0x00300030u);
    t.f[0] = ub0(sc01); t.f[1] = ub2(sc01); t.f[2] = ub0(sc23); t.f[3] = ub2(sc23);
    t.g[0] = ub0(mn01); t.g[1] = ub2(mn01); t.g[2] = ub0(mn23); t.g[3] = ub2(mn23);
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      const uint32_t N0 = (pp & 1) ? r.q[pp >> 1].z : r.q[pp >> 1].x, N1 = (pp & 1) ? r.q[pp >> 1].w : r.q[pp >> 1].y;
      t.bw[pp].x = (int)(N0 & 0x0F0F0F0Fu);
      t.bw[pp].y = (int)((N0 >> 4) & 0x0F0F0F0Fu);
      t.bw[pp].z = (int)(N1 & 0x0F0F0F0Fu);
      t.bw[pp].w = (int)((N1 >> 4) & 0x0F0F0F0Fu);
      if (is(F_Q5K)) {
        const uint32_t H = ((pp >> 1) ? r.q[2].y : r.q[2].x) >> (4 * (pp & 1));
        t.bw[pp].x |= (int)((H & 0x01010101u) << 4);
        t.bw[pp].y |= (int)(((H >> 1) & 0x01010101u) << 4);
        t.bw[pp].z |= (int)(((H >> 2) & 0x01010101u) << 4);
        t.bw[pp].w |= (int)(((H >> 3) & 0x01010101u) << 4);
      }
    }
    t.dd = h2f(r.hd.x & 0xFFFFu);
    t.dmin = h2f(r.hd.x >> 16);
  }
}

// mvq_mac_tile in its two phases — the four matrix-core products, then everything the vector ALU does with them — so that a caller
// can put other sequences' products between a sequence's two phases (matvec_batch.hip, mvqb2).  Same operations, same order.
struct MacD { i32x4 d[4]; };

__device__ __forceinline__ void mvq_mac_mfma(const TileOps& t, const i32x4 (&a)[4], MacD& m) {
  const i32x4 zero = {0, 0, 0, 0};
#pragma unroll
  for (int pp = 0; pp < 4; pp++) m.d[pp] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[pp], t.bw[pp], zero, 0, 0, 0);
}

template <uint32_t MASK>
__device__ __forceinline__ void mvq_mac_finish(int fmt, const TileOps& t, const MacD& m, const f32x4 xs, const f32x4 sx, float& acc);

template <uint32_t MASK>
__device__ __forceinline__ void mvq_mac_tile(int fmt, const TileOps& t, const XqOps& o, float& acc) {
  MacD m;
  mvq_mac_mfma(t, o.a, m);
  mvq_mac_finish<MASK>(fmt, t, m, o.xs, o.sx, acc);
}

template <uint32_t MASK>
__device__ __forceinline__ void mvq_mac_finish(int fmt, const TileOps& t, const MacD& m, const f32x4 xs, const f32x4 sx, float& acc) {
  auto is = [&](int f) { return mvq_is<MASK>(fmt, f); };
  struct { f32x4 xs, sx; } o{xs, sx};
  float V[4];
#pragma unroll
  for (int pp = 0; pp < 4; pp++) {
    const i32x4 d = m.d[pp];
    const float hi = (float)((d.x << 8) + d.y), lo = (float)((d.z << 8) + d.w);
    V[pp] = __builtin_fmaf(hi, 65536.0f, lo);
  }
  if (is(F_Q6K)) {
    float s1 = 0.0f;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) s1 = __builtin_fmaf(t.f[pp], __builtin_fmaf(o.sx[pp], V[pp], -32.0f * o.xs[pp]), s1);
    acc = __builtin_fmaf(t.dd, s1, acc);
  } else if (is(F_Q80) || is(F_Q40)) {
    float s1 = 0.0f;
#pragma unroll
    for (int pp = 0; pp < 4; pp++)
      s1 = __builtin_fmaf(t.f[pp], is(F_Q40) ? __builtin_fmaf(o.sx[pp], V[pp], -8.0f * o.xs[pp]) : o.sx[pp] * V[pp], s1);
    acc += s1;
  } else {
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
      s1 = __builtin_fmaf(t.f[pp] * o.sx[pp], V[pp], s1);
      s2 = __builtin_fmaf(t.g[pp], o.xs[pp], s2);
    }
    acc += t.dd * s1 - t.dmin * s2;
  }
}

}  // namespace lgh
