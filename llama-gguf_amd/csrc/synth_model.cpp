// synth_model.cpp — synthetic GGUF tensor payloads (host only; C ABI in include/llama_gguf_synth.h).
//
// Writes valid quantized blocks directly from a counter-based RNG (splitmix64 keyed by
// seed ^ fnv1a(tensor name), one stream per block), in the reference's block layouts
// (/root/reference/src/tensor/quant/blocks.rs:8-168).  Scales are chosen so the dequantized weights
// are ~zero-mean with std ~= 1/sqrt(in_features).  See SURVEY.md §8(d).
#include "../../include/llama_gguf_synth.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

namespace {

enum : uint32_t {
  T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8,
  T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14
};

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed, uint64_t counter) : s(seed ^ (counter * 0xD1B54A32D192ED03ull)) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }        // [0,1)
  float sym() { return 2.0f * uni() - 1.0f; }                                 // [-1,1)
  uint32_t below(uint32_t n) { return (uint32_t)(((next() >> 32) * (uint64_t)n) >> 32); }
  void bytes(uint8_t* dst, size_t n) {
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t v = next(); std::memcpy(dst + i, &v, 8); }
    if (i < n) { uint64_t v = next(); std::memcpy(dst + i, &v, n - i); }
  }
};

uint64_t fnv1a(const char* s) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (; *s; s++) { h ^= (uint8_t)*s; h *= 0x100000001b3ull; }
  return h;
}

uint16_t f2h(float f) {  // IEEE binary16, round-to-nearest-even
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, exp = (x >> 23) & 0xFFu, man = x & 0x7FFFFFu;
  if (exp == 255) return (uint16_t)(sign | 0x7C00u | (man ? 0x200u : 0));
  int e = (int)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;
    man |= 0x800000u;
    int shift = 14 - e;
    uint32_t hm = man >> shift, rem = man & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1))) hm++;
    return (uint16_t)(sign | hm);
  }
  uint32_t hm = man >> 13, rem = man & 0x1FFFu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | hm);
  if (rem > 0x1000u || (rem == 0x1000u && (hm & 1))) h++;
  return h;
}

float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1Fu, man = h & 0x3FFu, bits;
  if (exp == 0) {
    if (!man) bits = sign;
    else {
      int e = -1;
      do { man <<= 1; e++; } while (!(man & 0x400u));
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3FFu) << 13);
    }
  } else if (exp == 31) bits = sign | 0x7F800000u | (man << 13);
  else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

void st16(uint8_t* p, uint16_t v) { std::memcpy(p, &v, 2); }

size_t blk_size(uint32_t t) {
  switch (t) {
    case T_F32: case T_F16: return 1;
    case T_Q4_0: case T_Q4_1: case T_Q5_0: case T_Q5_1: case T_Q8_0: return 32;
    case T_Q2_K: case T_Q3_K: case T_Q4_K: case T_Q5_K: case T_Q6_K: return 256;
    default: return 0;
  }
}

size_t blk_bytes(uint32_t t) {
  switch (t) {
    case T_F32: return 4; case T_F16: return 2;
    case T_Q4_0: return 18; case T_Q4_1: return 20; case T_Q5_0: return 22; case T_Q5_1: return 24;
    case T_Q8_0: return 34; case T_Q2_K: return 84; case T_Q3_K: return 110; case T_Q4_K: return 144;
    case T_Q5_K: return 176; case T_Q6_K: return 210;
    default: return 0;
  }
}

// 6-bit (scale, min) packing of Q4_K / Q5_K: the exact inverse of the unpack every decoder on the path uses
// (dequant.rs:210-223, simd.rs:985-994).  NB the reference's own quantize_q4_k/q5_k pack the upper four
// scales differently (dequant.rs:790-796: low 2 bits on top of byte j, high 4 bits in byte j+8), which its
// unpack does not invert — a reference quirk that only affects its round-trip tests, not decoding.
void pack_k4(const uint8_t sc[8], const uint8_t mn[8], uint8_t* out12) {
  for (int j = 0; j < 4; j++) {
    out12[j] = (uint8_t)((sc[j] & 0x3F) | ((sc[j + 4] >> 4) << 6));
    out12[j + 4] = (uint8_t)((mn[j] & 0x3F) | ((mn[j + 4] >> 4) << 6));
    out12[j + 8] = (uint8_t)((sc[j + 4] & 0x0F) | ((mn[j + 4] & 0x0F) << 4));
  }
}

// One block of `type` at `b`; base = 1/sqrt(in_features).
void fill_block(uint32_t type, Rng& r, float base, uint8_t* b) {
  float jitter = 0.75f + 0.5f * r.uni();
  switch (type) {
    case T_Q4_0: {  // {d, qs[16]}: (q-8)*d
      st16(b, f2h(base * jitter / 4.64f));
      r.bytes(b + 2, 16);
      break;
    }
    case T_Q4_1: {  // {d, m, qs[16]}: q*d + m
      uint16_t d = f2h(base * jitter / 4.61f);
      st16(b, d);
      st16(b + 2, f2h(-7.5f * h2f(d)));
      r.bytes(b + 4, 16);
      break;
    }
    case T_Q5_0: {  // {d, qh[4], qs[16]}: (q-16)*d
      st16(b, f2h(base * jitter / 9.25f));
      r.bytes(b + 2, 20);
      break;
    }
    case T_Q5_1: {  // {d, m, qh[4], qs[16]}
      uint16_t d = f2h(base * jitter / 9.23f);
      st16(b, d);
      st16(b + 2, f2h(-15.5f * h2f(d)));
      r.bytes(b + 4, 20);
      break;
    }
    case T_Q8_0: {  // {d, qs[32] i8}
      st16(b, f2h(base * jitter / 73.6f));
      r.bytes(b + 2, 32);
      for (int i = 0; i < 32; i++)
        if (b[2 + i] == 0x80) b[2 + i] = 0;  // keep q in [-127, 127]
      break;
    }
    case T_Q2_K: {  // {scales[16], qs[64], d, dmin}: d*sc*q - dmin*mn (reference's sequential layout)
      for (int i = 0; i < 16; i++) {
        uint32_t sc = 1 + r.below(15);
        uint32_t mn = (uint32_t)std::lround(0.75 * sc);
        b[i] = (uint8_t)(sc | (mn << 4));
      }
      r.bytes(b + 16, 64);
      uint16_t d = f2h(base * jitter / 10.16f);
      st16(b + 80, d);
      st16(b + 82, f2h(2.0f * h2f(d)));
      break;
    }
    case T_Q3_K: {  // {hmask[32], qs[64], scales[12], d}: any 12 bytes decode to valid 6-bit scales
      r.bytes(b, 108);
      st16(b + 108, f2h(base * jitter / 43.4f));
      break;
    }
    case T_Q4_K: {  // {d, dmin, scales[12], qs[128]}
      uint8_t sc[8], mn[8];
      for (int j = 0; j < 8; j++) {
        sc[j] = (uint8_t)(8 + r.below(48));
        mn[j] = (uint8_t)std::lround(0.9375 * sc[j]);  // dmin*mn == d*sc*7.5 with dmin = 8d
      }
      uint16_t d = f2h(base * jitter / 158.6f);
      st16(b, d);
      st16(b + 2, f2h(8.0f * h2f(d)));
      pack_k4(sc, mn, b + 4);
      r.bytes(b + 16, 128);
      break;
    }
    case T_Q5_K: {  // {d, dmin, scales[12], qh[32], qs[128]}
      uint8_t sc[8], mn[8];
      for (int j = 0; j < 8; j++) {
        sc[j] = (uint8_t)(8 + r.below(48));
        mn[j] = (uint8_t)std::lround(0.96875 * sc[j]);  // dmin*mn == d*sc*15.5 with dmin = 16d
      }
      uint16_t d = f2h(base * jitter / 317.6f);
      st16(b, d);
      st16(b + 2, f2h(16.0f * h2f(d)));
      pack_k4(sc, mn, b + 4);
      r.bytes(b + 16, 160);
      break;
    }
    case T_Q6_K: {  // {ql[128], qh[64], scales[16] i8, d}
      r.bytes(b, 192);
      for (int i = 0; i < 16; i++) {
        int v = (int)r.below(63) - 32;  // [-32, 30]
        if (v >= 0) v += 1;             // [-32,-1] u [1,31]
        b[192 + i] = (uint8_t)(int8_t)v;
      }
      st16(b + 208, f2h(base * jitter / 338.0f));
      break;
    }
    default: break;
  }
}

}  // namespace

extern "C" {

size_t lgs_tensor_nbytes(uint32_t type, uint64_t n_elems) {
  size_t bs = blk_size(type), bb = blk_bytes(type);
  if (!bs || n_elems % bs) return 0;
  return (size_t)(n_elems / bs) * bb;
}

int lgs_fill_tensor(const char* name, uint32_t type, uint64_t n_elems, uint64_t in_features, int kind, uint64_t seed,
                    void* out, size_t nbytes, int threads) {
  size_t bs = blk_size(type), bb = blk_bytes(type);
  if (!name || !out || !bs || n_elems % bs || lgs_tensor_nbytes(type, n_elems) != nbytes || in_features == 0) return 1;
  if (kind != 0 && type != T_F32) return 1;
  uint64_t key = seed ^ fnv1a(name);
  float base = 1.0f / std::sqrt((float)in_features);
  uint8_t* dst = (uint8_t*)out;
  // work units: one quantized block, or 256 scalar elements
  size_t unit_elems = bs == 1 ? 256 : bs;
  size_t units = (size_t)((n_elems + unit_elems - 1) / unit_elems);
  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  nt = std::max(1, std::min<int>(nt, (int)std::max<size_t>(1, units / 1024)));
  auto work = [&](size_t u0, size_t u1) {
    for (size_t u = u0; u < u1; u++) {
      Rng r(key, u);
      if (bs > 1) {
        fill_block(type, r, base, dst + u * bb);
        continue;
      }
      size_t e0 = u * 256, e1 = std::min<size_t>((size_t)n_elems, e0 + 256);
      for (size_t e = e0; e < e1; e++) {
        float v;
        if (kind == 1) v = 1.0f + 0.01f * r.sym();
        else if (kind == 2) v = 0.01f * r.sym();
        else v = 1.7320508f * base * r.sym();
        if (type == T_F32) std::memcpy(dst + e * 4, &v, 4);
        else st16(dst + e * 2, f2h(v));
      }
    }
  };
  if (nt == 1) {
    work(0, units);
  } else {
    std::vector<std::thread> th;
    size_t per = (units + nt - 1) / nt;
    for (int t = 0; t < nt; t++) {
      size_t u0 = std::min(units, (size_t)t * per), u1 = std::min(units, u0 + per);
      if (u0 < u1) th.emplace_back(work, u0, u1);
    }
    for (auto& t : th) t.join();
  }
  return 0;
}

}  // extern "C"
