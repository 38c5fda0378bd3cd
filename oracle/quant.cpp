// oracle/quant.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h).
//
// CPU restatement of the reference's quantized block formats, its (de)quantizers and its fused
// quantized dot products.  Built with -ffp-contract=off: Rust never contracts a*b+c, so every
// multiply and add below rounds separately exactly as in the reference.
//
//   block layouts     src/tensor/quant/blocks.rs:8-168 (sizes asserted 293-305)
//   dequantize_q*     src/tensor/quant/dequant.rs:16-367
//   quantize_q*       src/tensor/quant/dequant.rs:373-1030   (round-trip KATs only)
//   dot_q*            src/backend/cpu/simd.rs:931-1166
//   f16 <-> f32       crate half 2.7.1 (Cargo.lock:907): IEEE binary16, round-to-nearest-even
#include "oracle.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace {

// ---- IEEE binary16 conversion (exact; what half::f16::{to_f32,from_f32} compute) ----
inline float h2f(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t exp = (h >> 10) & 0x1Fu;
  uint32_t man = h & 0x3FFu;
  uint32_t bits;
  if (exp == 0) {
    if (man == 0) {
      bits = sign;
    } else {  // subnormal: normalise
      int e = -1;
      do { man <<= 1; e++; } while ((man & 0x400u) == 0);
      man &= 0x3FFu;
      bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
    }
  } else if (exp == 31) {
    bits = sign | 0x7F800000u | (man << 13);
  } else {
    bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
  }
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

inline uint16_t f2h(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t exp = (x >> 23) & 0xFFu;
  uint32_t man = x & 0x7FFFFFu;
  if (exp == 255) {  // inf / nan
    return (uint16_t)(sign | 0x7C00u | (man ? (0x200u | (man >> 13)) : 0));
  }
  int e = (int)exp - 127 + 15;
  if (e >= 31) return (uint16_t)(sign | 0x7C00u);  // overflow -> inf
  if (e <= 0) {
    if (e < -10) return (uint16_t)sign;  // underflow -> 0
    man |= 0x800000u;
    int shift = 14 - e;  // 14..24
    uint32_t half_man = man >> shift;
    uint32_t rem = man & ((1u << shift) - 1);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_man & 1))) half_man++;
    return (uint16_t)(sign | half_man);
  }
  uint32_t half_man = man >> 13;
  uint32_t rem = man & 0x1FFFu;
  uint16_t h = (uint16_t)(sign | ((uint32_t)e << 10) | half_man);
  if (rem > 0x1000u || (rem == 0x1000u && (half_man & 1))) h++;  // may carry into exponent: correct
  return h;
}

inline uint16_t ld16(const uint8_t* p) { uint16_t v; std::memcpy(&v, p, 2); return v; }
inline void st16(uint8_t* p, uint16_t v) { std::memcpy(p, &v, 2); }
inline float ldf32(const uint8_t* p) { float v; std::memcpy(&v, p, 4); return v; }

// Rust `as u8` / `as i8` / `as i32` from f32 saturate; inputs here are already clamped.
inline float rround(float v) { return std::roundf(v); }  // f32::round: half away from zero
inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// f32::max / f32::min semantics for non-NaN inputs
inline float fmaxr(float a, float b) { return a > b ? a : b; }
inline float fminr(float a, float b) { return a < b ? a : b; }

// K-quant 6-bit (scale, min) unpack shared by Q4_K and Q5_K (dequant.rs:210-223, simd.rs:985-994)
inline void unpack_scales_k4(const uint8_t* sc12, uint8_t scales[8], uint8_t mins[8]) {
  for (int j = 0; j < 4; j++) {
    scales[j] = sc12[j] & 0x3F;
    mins[j] = sc12[j + 4] & 0x3F;
  }
  for (int j = 4; j < 8; j++) {
    scales[j] = (uint8_t)((sc12[j + 4] & 0x0F) | ((sc12[j - 4] >> 6) << 4));
    mins[j] = (uint8_t)(((sc12[j + 4] >> 4) & 0x0F) | ((sc12[j] >> 6) << 4));
  }
}

// ------------------------------------------------------------------ dequantize (one block)
// Byte offsets follow the #[repr(C)] structs of blocks.rs.

void deq_q4_0(const uint8_t* b, float* out) {  // dequant.rs:16-30   {d:f16, qs[16]}
  float d = h2f(ld16(b));
  const uint8_t* qs = b + 2;
  for (int i = 0; i < 16; i++) {
    int lo = (int)(qs[i] & 0x0F) - 8;
    int hi = (int)((qs[i] >> 4) & 0x0F) - 8;
    out[i] = (float)lo * d;
    out[i + 16] = (float)hi * d;
  }
}

void deq_q4_1(const uint8_t* b, float* out) {  // dequant.rs:36-48   {d, m, qs[16]}
  float d = h2f(ld16(b)), m = h2f(ld16(b + 2));
  const uint8_t* qs = b + 4;
  for (int i = 0; i < 16; i++) {
    float lo = (float)(qs[i] & 0x0F);
    float hi = (float)((qs[i] >> 4) & 0x0F);
    out[i] = lo * d + m;
    out[i + 16] = hi * d + m;
  }
}

void deq_q5_0(const uint8_t* b, float* out) {  // dequant.rs:54-76   {d, qh[4], qs[16]}
  float d = h2f(ld16(b));
  uint32_t qh;
  std::memcpy(&qh, b + 2, 4);
  const uint8_t* qs = b + 6;
  for (int i = 0; i < 16; i++) {
    int lo4 = qs[i] & 0x0F, hi4 = (qs[i] >> 4) & 0x0F;
    int lo5 = (qh >> i) & 1, hi5 = (qh >> (i + 16)) & 1;
    int lo = (lo4 | (lo5 << 4)) - 16;
    int hi = (hi4 | (hi5 << 4)) - 16;
    out[i] = (float)lo * d;
    out[i + 16] = (float)hi * d;
  }
}

void deq_q5_1(const uint8_t* b, float* out) {  // dequant.rs:81-101  {d, m, qh[4], qs[16]}
  float d = h2f(ld16(b)), m = h2f(ld16(b + 2));
  uint32_t qh;
  std::memcpy(&qh, b + 4, 4);
  const uint8_t* qs = b + 8;
  for (int i = 0; i < 16; i++) {
    uint32_t lo4 = qs[i] & 0x0F, hi4 = (qs[i] >> 4) & 0x0F;
    uint32_t lo5 = (qh >> i) & 1, hi5 = (qh >> (i + 16)) & 1;
    uint32_t lo = lo4 | (lo5 << 4), hi = hi4 | (hi5 << 4);
    out[i] = (float)lo * d + m;
    out[i + 16] = (float)hi * d + m;
  }
}

void deq_q8_0(const uint8_t* b, float* out) {  // dequant.rs:106-112 {d, qs[32] i8}
  float d = h2f(ld16(b));
  const int8_t* qs = (const int8_t*)(b + 2);
  for (int i = 0; i < 32; i++) out[i] = (float)qs[i] * d;
}

void deq_q8_1(const uint8_t* b, float* out) {  // dequant.rs:117-123 {d:f32, qs[32] i8}
  float d = ldf32(b);
  const int8_t* qs = (const int8_t*)(b + 4);
  for (int i = 0; i < 32; i++) out[i] = (float)qs[i] * d;
}

// Q2_K: the reference's own SEQUENTIAL layout (not upstream ggml's interleave)
void deq_q2_k(const uint8_t* b, float* out) {  // dequant.rs:129-156 {scales[16], qs[64], d, dmin}
  const uint8_t* scales = b;
  const uint8_t* qs = b + 16;
  float d = h2f(ld16(b + 80)), dmin = h2f(ld16(b + 82));
  for (int i = 0; i < 16; i++) {
    float scale = (float)(scales[i] & 0x0F);
    float mn = (float)((scales[i] >> 4) & 0x0F);
    float d_scale = d * scale, d_min = dmin * mn;
    for (int j = 0; j < 4; j++) {
      uint8_t byte = qs[i * 4 + j];
      for (int k = 0; k < 4; k++) {
        float q = (float)((byte >> (k * 2)) & 0x03);
        out[i * 16 + j * 4 + k] = d_scale * q - d_min;
      }
    }
  }
}

// Q3_K: the reference's own sequential layout (quirk Q1)
void deq_q3_k(const uint8_t* b, float* out) {  // dequant.rs:161-200 {hmask[32], qs[64], scales[12], d}
  const uint8_t* hmask = b;
  const uint8_t* qs = b + 32;
  const uint8_t* sc = b + 96;
  float d = h2f(ld16(b + 108));
  int8_t scales[16];
  for (int i = 0; i < 4; i++) {
    int b0 = sc[i * 3], b1 = sc[i * 3 + 1], b2 = sc[i * 3 + 2];
    scales[i * 4] = (int8_t)((int8_t)(b0 & 0x3F) - 32);
    scales[i * 4 + 1] = (int8_t)((int8_t)((b0 >> 6) | ((b1 & 0x0F) << 2)) - 32);
    scales[i * 4 + 2] = (int8_t)((int8_t)((b1 >> 4) | ((b2 & 0x03) << 4)) - 32);
    scales[i * 4 + 3] = (int8_t)((int8_t)(b2 >> 2) - 32);
  }
  for (int i = 0; i < 16; i++) {
    float scale = d * (float)scales[i];
    for (int j = 0; j < 16; j++) {
      int idx = i * 16 + j;
      int lo2 = (qs[idx / 4] >> ((idx % 4) * 2)) & 0x03;
      int hi1 = (hmask[idx / 8] >> (idx % 8)) & 0x01;
      int q = (lo2 | (hi1 << 2)) - 4;
      out[idx] = scale * (float)q;
    }
  }
}

void deq_q4_k(const uint8_t* b, float* out) {  // dequant.rs:205-259 {d, dmin, scales[12], qs[128]}
  float d = h2f(ld16(b)), dmin = h2f(ld16(b + 2));
  uint8_t scales[8], mins[8];
  unpack_scales_k4(b + 4, scales, mins);
  const uint8_t* qs = b + 16;
  int o = 0, qp = 0, is = 0;
  for (int g = 0; g < 4; g++) {
    float d1 = d * (float)scales[is], m1 = dmin * (float)mins[is];
    float d2 = d * (float)scales[is + 1], m2 = dmin * (float)mins[is + 1];
    for (int l = 0; l < 32; l++) out[o++] = d1 * (float)(qs[qp + l] & 0x0F) - m1;
    for (int l = 0; l < 32; l++) out[o++] = d2 * (float)((qs[qp + l] >> 4) & 0x0F) - m2;
    qp += 32;
    is += 2;
  }
}

void deq_q5_k(const uint8_t* b, float* out) {  // dequant.rs:265-316 {d, dmin, scales[12], qh[32], qs[128]}
  float d = h2f(ld16(b)), dmin = h2f(ld16(b + 2));
  uint8_t scales[8], mins[8];
  unpack_scales_k4(b + 4, scales, mins);
  const uint8_t* qh = b + 16;
  const uint8_t* qs = b + 48;
  int o = 0, qp = 0, is = 0;
  uint8_t u1 = 1, u2 = 2;
  for (int g = 0; g < 4; g++) {
    float d1 = d * (float)scales[is], m1 = dmin * (float)mins[is];
    float d2 = d * (float)scales[is + 1], m2 = dmin * (float)mins[is + 1];
    for (int l = 0; l < 32; l++) {
      float lo4 = (float)(qs[qp + l] & 0x0F);
      float hi5 = (qh[l] & u1) ? 16.0f : 0.0f;
      out[o++] = d1 * (lo4 + hi5) - m1;
    }
    for (int l = 0; l < 32; l++) {
      float hi4 = (float)((qs[qp + l] >> 4) & 0x0F);
      float hi5 = (qh[l] & u2) ? 16.0f : 0.0f;
      out[o++] = d2 * (hi4 + hi5) - m2;
    }
    qp += 32;
    is += 2;
    u1 = (uint8_t)(u1 << 2);
    u2 = (uint8_t)(u2 << 2);
  }
}

inline void q6k_quad(const uint8_t* ql, const uint8_t* qh, int l, int q[4]) {
  q[0] = (int)((ql[l] & 0x0F) | ((qh[l] & 0x03) << 4)) - 32;
  q[1] = (int)((ql[l + 32] & 0x0F) | (((qh[l] >> 2) & 0x03) << 4)) - 32;
  q[2] = (int)((ql[l] >> 4) | (((qh[l] >> 4) & 0x03) << 4)) - 32;
  q[3] = (int)((ql[l + 32] >> 4) | (((qh[l] >> 6) & 0x03) << 4)) - 32;
}

void deq_q6_k(const uint8_t* b, float* out) {  // dequant.rs:322-356 {ql[128], qh[64], scales[16] i8, d}
  const int8_t* sc = (const int8_t*)(b + 192);
  float d = h2f(ld16(b + 208));
  for (int n = 0; n < 2; n++) {
    const uint8_t* ql = b + n * 64;
    const uint8_t* qh = b + 128 + n * 32;
    for (int l = 0; l < 32; l++) {
      int is = l / 16, q[4];
      q6k_quad(ql, qh, l, q);
      out[n * 128 + l] = d * (float)sc[n * 8 + is] * (float)q[0];
      out[n * 128 + l + 32] = d * (float)sc[n * 8 + is + 2] * (float)q[1];
      out[n * 128 + l + 64] = d * (float)sc[n * 8 + is + 4] * (float)q[2];
      out[n * 128 + l + 96] = d * (float)sc[n * 8 + is + 6] * (float)q[3];
    }
  }
}

void deq_q8_k(const uint8_t* b, float* out) {  // dequant.rs:361-367 {d:f32, qs[256] i8, bsums[16] i16}
  float d = ldf32(b);
  const int8_t* qs = (const int8_t*)(b + 4);
  for (int i = 0; i < 256; i++) out[i] = (float)qs[i] * d;
}

// ------------------------------------------------------------------ quantize (one block)

void q_q4_0(const float* in, uint8_t* b) {  // dequant.rs:374-397
  float amax = 0.0f;
  for (int i = 0; i < 32; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = amax / 7.0f;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  st16(b, f2h(d));
  for (int i = 0; i < 16; i++) {
    int lo = clampi((int)rround(in[i] * id), -8, 7) + 8;
    int hi = clampi((int)rround(in[i + 16] * id), -8, 7) + 8;
    b[2 + i] = (uint8_t)(lo | (hi << 4));
  }
}

void q_q4_1(const float* in, uint8_t* b) {  // dequant.rs:400-427
  float mn = std::numeric_limits<float>::infinity(), mx = -mn;
  for (int i = 0; i < 32; i++) { mn = fminr(mn, in[i]); mx = fmaxr(mx, in[i]); }
  float d = (mx - mn) / 15.0f, m = mn;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  st16(b, f2h(d));
  st16(b + 2, f2h(m));
  for (int i = 0; i < 16; i++) {
    int lo = clampi((int)rround((in[i] - m) * id), 0, 15);
    int hi = clampi((int)rround((in[i + 16] - m) * id), 0, 15);
    b[4 + i] = (uint8_t)(lo | (hi << 4));
  }
}

void q_q8_0(const float* in, uint8_t* b) {  // dequant.rs:430-451
  float amax = 0.0f;
  for (int i = 0; i < 32; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = amax / 127.0f;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  st16(b, f2h(d));
  for (int i = 0; i < 32; i++) b[2 + i] = (uint8_t)(int8_t)clampf(rround(in[i] * id), -127.0f, 127.0f);
}

void q_q5_0(const float* in, uint8_t* b) {  // dequant.rs:457-489
  float amax = 0.0f;
  for (int i = 0; i < 32; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = amax / 15.0f;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  uint32_t qh = 0;
  st16(b, f2h(d));
  for (int i = 0; i < 16; i++) {
    int lo = clampi((int)rround(in[i] * id), -16, 15) + 16;
    int hi = clampi((int)rround(in[i + 16] * id), -16, 15) + 16;
    b[6 + i] = (uint8_t)((lo & 0x0F) | ((hi & 0x0F) << 4));
    qh |= (uint32_t)((lo >> 4) & 1) << i;
    qh |= (uint32_t)((hi >> 4) & 1) << (i + 16);
  }
  std::memcpy(b + 2, &qh, 4);
}

void q_q5_1(const float* in, uint8_t* b) {  // dequant.rs:495-531
  float mn = std::numeric_limits<float>::infinity(), mx = -mn;
  for (int i = 0; i < 32; i++) { mn = fminr(mn, in[i]); mx = fmaxr(mx, in[i]); }
  float d = (mx - mn) / 31.0f, m = mn;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  uint32_t qh = 0;
  st16(b, f2h(d));
  st16(b + 2, f2h(m));
  for (int i = 0; i < 16; i++) {
    int lo = clampi((int)rround((in[i] - m) * id), 0, 31);
    int hi = clampi((int)rround((in[i + 16] - m) * id), 0, 31);
    b[8 + i] = (uint8_t)((lo & 0x0F) | ((hi & 0x0F) << 4));
    qh |= (uint32_t)((lo >> 4) & 1) << i;
    qh |= (uint32_t)((hi >> 4) & 1) << (i + 16);
  }
  std::memcpy(b + 4, &qh, 4);
}

void q_q8_1(const float* in, uint8_t* b) {  // dequant.rs:536-551
  float amax = 0.0f;
  for (int i = 0; i < 32; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = amax / 127.0f;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  std::memcpy(b, &d, 4);
  for (int i = 0; i < 32; i++) b[4 + i] = (uint8_t)(int8_t)clampf(rround(in[i] * id), -127.0f, 127.0f);
}

void q_q2_k(const float* in, uint8_t* b) {  // dequant.rs:558-634
  float ranges[16], mins_neg[16];
  for (int i = 0; i < 16; i++) {
    float gmin = std::numeric_limits<float>::infinity(), gmax = -gmin;
    for (int j = 0; j < 16; j++) { gmin = fminr(gmin, in[i * 16 + j]); gmax = fmaxr(gmax, in[i * 16 + j]); }
    ranges[i] = fmaxr(gmax - gmin, 0.0f);
    mins_neg[i] = fmaxr(-gmin, 0.0f);
  }
  float max_range = 0.0f, max_neg_min = 0.0f;
  for (int i = 0; i < 16; i++) { max_range = fmaxr(max_range, ranges[i]); max_neg_min = fmaxr(max_neg_min, mins_neg[i]); }
  float d = max_range > 0.0f ? max_range / (3.0f * 15.0f) : 1.0f;
  float dmin = max_neg_min > 0.0f ? max_neg_min / 15.0f : 1.0f;
  std::memset(b, 0, 84);
  for (int i = 0; i < 16; i++) {
    uint8_t scale = d > 0.0f ? (uint8_t)clampf(rround(ranges[i] / (3.0f * d)), 0.0f, 15.0f) : 0;
    uint8_t floor1 = ranges[i] > 0.0f ? 1 : 0;
    if (scale < floor1) scale = floor1;
    uint8_t min_val = dmin > 0.0f ? (uint8_t)clampf(rround(mins_neg[i] / dmin), 0.0f, 15.0f) : 0;
    float d_scale = d * (float)scale, d_min = dmin * (float)min_val;
    float id_scale = d_scale > 0.0f ? 1.0f / d_scale : 0.0f;
    b[i] = (uint8_t)(scale | (min_val << 4));
    for (int j = 0; j < 4; j++) {
      uint8_t byte = 0;
      for (int k = 0; k < 4; k++) {
        float v = in[i * 16 + j * 4 + k];
        uint8_t q = (uint8_t)clampf(rround((v + d_min) * id_scale), 0.0f, 3.0f);
        byte |= (uint8_t)((q & 0x03) << (k * 2));
      }
      b[16 + i * 4 + j] = byte;
    }
  }
  st16(b + 80, f2h(d));
  st16(b + 82, f2h(dmin));
}

void q_q3_k(const float* in, uint8_t* b) {  // dequant.rs:641-703
  float amax = 0.0f;
  for (int i = 0; i < 256; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = fmaxr(1.0f, amax / 32.0f);
  int8_t scales_signed[16];
  std::memset(b, 0, 110);
  uint8_t* hmask = b;
  uint8_t* qs = b + 32;
  for (int i = 0; i < 16; i++) {
    float gmax = 0.0f;
    for (int j = 0; j < 16; j++) gmax = fmaxr(gmax, std::fabs(in[i * 16 + j]));
    float scale_f = gmax > 1e-10f ? gmax / 3.0f / d : 0.0f;
    uint8_t sc = (uint8_t)(((int)rround(clampf(scale_f, -32.0f, 31.0f)) + 32)) & 0x3F;
    int8_t sc_signed = (int8_t)((int8_t)sc - 32);
    scales_signed[i] = sc_signed;
    float scale = d * (float)sc_signed;
    float id_scale = scale != 0.0f ? 1.0f / scale : 0.0f;
    for (int j = 0; j < 16; j++) {
      int q = (int)clampf(rround(in[i * 16 + j] * id_scale), -4.0f, 3.0f) + 4;
      q = clampi(q, 0, 7);
      int idx = i * 16 + j;
      int sh = (idx % 4) * 2;
      qs[idx / 4] = (uint8_t)((qs[idx / 4] & ~(0x03u << sh)) | ((q & 0x03) << sh));
      int hs = idx % 8;
      hmask[idx / 8] = (uint8_t)((hmask[idx / 8] & ~(1u << hs)) | (((q >> 2) & 1) << hs));
    }
  }
  uint8_t* scales = b + 96;
  for (int i = 0; i < 4; i++) {
    uint8_t sc0 = (uint8_t)(((int)scales_signed[i * 4] + 32) & 0x3F);
    uint8_t sc1 = (uint8_t)(((int)scales_signed[i * 4 + 1] + 32) & 0x3F);
    uint8_t sc2 = (uint8_t)(((int)scales_signed[i * 4 + 2] + 32) & 0x3F);
    uint8_t sc3 = (uint8_t)(((int)scales_signed[i * 4 + 3] + 32) & 0x3F);
    scales[i * 3] = (uint8_t)((sc0 & 0x3F) | ((sc1 & 0x03) << 6));
    scales[i * 3 + 1] = (uint8_t)(((sc1 >> 2) & 0x0F) | ((sc2 & 0x0F) << 4));
    scales[i * 3 + 2] = (uint8_t)(((sc2 >> 4) & 0x03) | ((sc3 & 0x3F) << 2));
  }
  st16(b + 108, f2h(d));
}

// shared first pass of quantize_q4_k / quantize_q5_k (dequant.rs:710-746, 816-848)
void k45_prepass(const float* in, float qmax, float ranges[8], float gmins[8], float* d, float* dmin) {
  for (int is = 0; is < 8; is++) {
    float gmin = std::numeric_limits<float>::infinity(), gmax = -gmin;
    for (int l = 0; l < 32; l++) { gmin = fminr(gmin, in[is * 32 + l]); gmax = fmaxr(gmax, in[is * 32 + l]); }
    ranges[is] = fmaxr(gmax - gmin, 0.0f);
    gmins[is] = gmin;
  }
  float max_range = 0.0f, max_neg_min = 0.0f;
  for (int is = 0; is < 8; is++) {
    max_range = fmaxr(max_range, ranges[is]);
    max_neg_min = fmaxr(max_neg_min, fmaxr(-gmins[is], 0.0f));
  }
  *d = max_range > 0.0f ? max_range / (qmax * 63.0f) : 1.0f;
  *dmin = max_neg_min > 0.0f ? max_neg_min / 63.0f : 1.0f;
}

void k45_pack_scales(const uint8_t scales[8], const uint8_t mins[8], uint8_t* sb) {  // dequant.rs:790-796
  for (int j = 0; j < 4; j++) {
    sb[j] = (uint8_t)((scales[j] & 0x3F) | ((scales[j + 4] & 0x03) << 6));
    sb[j + 4] = (uint8_t)((mins[j] & 0x3F) | ((mins[j + 4] & 0x03) << 6));
    sb[j + 8] = (uint8_t)(((scales[j + 4] >> 2) & 0x0F) | (((mins[j + 4] >> 2) & 0x0F) << 4));
  }
}

void q_q4_k(const float* in, uint8_t* b) {  // dequant.rs:710-804
  float ranges[8], gmins[8], d, dmin;
  k45_prepass(in, 15.0f, ranges, gmins, &d, &dmin);
  uint8_t scales[8], mins[8];
  std::memset(b, 0, 144);
  uint8_t* qs = b + 16;
  for (int is = 0; is < 8; is++) {
    uint8_t scale = d > 0.0f ? (uint8_t)clampf(rround(ranges[is] / (15.0f * d)), 0.0f, 63.0f) : 0;
    uint8_t floor1 = ranges[is] > 0.0f ? 1 : 0;
    if (scale < floor1) scale = floor1;
    uint8_t min_val = dmin > 0.0f ? (uint8_t)clampf(rround(fmaxr(-gmins[is], 0.0f) / dmin), 0.0f, 63.0f) : 0;
    scales[is] = scale;
    mins[is] = min_val;
    float d_scale = d * (float)scale, m = dmin * (float)min_val;
    float id_scale = d_scale > 0.0f ? 1.0f / d_scale : 0.0f;
    int qp = (is / 2) * 32;
    bool high = (is % 2) == 1;
    for (int l = 0; l < 32; l++) {
      uint8_t q = (uint8_t)clampf(rround((in[is * 32 + l] + m) * id_scale), 0.0f, 15.0f);
      if (high) qs[qp + l] = (uint8_t)((qs[qp + l] & 0x0F) | ((q & 0x0F) << 4));
      else qs[qp + l] = (uint8_t)((qs[qp + l] & 0xF0) | (q & 0x0F));
    }
  }
  st16(b, f2h(d));
  st16(b + 2, f2h(dmin));
  k45_pack_scales(scales, mins, b + 4);
}

void q_q5_k(const float* in, uint8_t* b) {  // dequant.rs:811-911
  float ranges[8], gmins[8], d, dmin;
  k45_prepass(in, 31.0f, ranges, gmins, &d, &dmin);
  uint8_t scales[8], mins[8];
  std::memset(b, 0, 176);
  uint8_t* qh = b + 16;
  uint8_t* qs = b + 48;
  for (int is = 0; is < 8; is++) {
    uint8_t scale = d > 0.0f ? (uint8_t)clampf(rround(ranges[is] / (31.0f * d)), 0.0f, 63.0f) : 0;
    uint8_t floor1 = ranges[is] > 0.0f ? 1 : 0;
    if (scale < floor1) scale = floor1;
    uint8_t min_val = dmin > 0.0f ? (uint8_t)clampf(rround(fmaxr(-gmins[is], 0.0f) / dmin), 0.0f, 63.0f) : 0;
    scales[is] = scale;
    mins[is] = min_val;
    float d_scale = d * (float)scale, m = dmin * (float)min_val;
    float id_scale = d_scale > 0.0f ? 1.0f / d_scale : 0.0f;
    int qp = (is / 2) * 32;
    bool high = (is % 2) == 1;
    for (int l = 0; l < 32; l++) {
      uint8_t q = (uint8_t)clampf(rround((in[is * 32 + l] + m) * id_scale), 0.0f, 31.0f);
      uint8_t lo4 = q & 0x0F, hi5 = (q >> 4) & 1;
      if (high) qs[qp + l] = (uint8_t)((qs[qp + l] & 0x0F) | (lo4 << 4));
      else qs[qp + l] = (uint8_t)((qs[qp + l] & 0xF0) | lo4);
      if (hi5) qh[l] |= (uint8_t)(1u << is);
    }
  }
  st16(b, f2h(d));
  st16(b + 2, f2h(dmin));
  k45_pack_scales(scales, mins, b + 4);
}

void q_q6_k(const float* in, uint8_t* b) {  // dequant.rs:917-996
  float amax = 0.0f;
  for (int i = 0; i < 256; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = fmaxr(amax / 16.0f, 1e-10f);
  std::memset(b, 0, 210);
  int8_t* scales = (int8_t*)(b + 192);
  for (int n = 0; n < 2; n++) {
    uint8_t* ql = b + n * 64;
    uint8_t* qh = b + 128 + n * 32;
    int sc_base = n * 8, out_base = n * 128;
    for (int s = 0; s < 8; s++) {
      float gmax = 0.0f;
      for (int i = 0; i < 16; i++) gmax = fmaxr(gmax, std::fabs(in[out_base + s * 16 + i]));
      float scale_f = gmax > 1e-10f ? gmax / 31.0f / d : 0.0f;
      int8_t sc = (int8_t)(int)rround(clampf(scale_f, -128.0f, 127.0f));
      scales[sc_base + s] = (sc == 0 && gmax > 1e-10f) ? (int8_t)1 : sc;
    }
    for (int l = 0; l < 32; l++) {
      int is = l / 16;
      int q[4];
      for (int t = 0; t < 4; t++) {
        float scale = d * (float)scales[sc_base + is + 2 * t];
        float id = scale != 0.0f ? 1.0f / scale : 0.0f;
        int v = (int)clampf(rround(in[out_base + l + 32 * t] * id), -32.0f, 31.0f) + 32;
        q[t] = clampi(v, 0, 63);
      }
      ql[l] = (uint8_t)((q[0] & 0x0F) | ((q[2] & 0x0F) << 4));
      ql[l + 32] = (uint8_t)((q[1] & 0x0F) | ((q[3] & 0x0F) << 4));
      qh[l] = (uint8_t)((q[0] >> 4) | ((q[1] >> 4) << 2) | ((q[2] >> 4) << 4) | ((q[3] >> 4) << 6));
    }
  }
  st16(b + 208, f2h(d));
}

void q_q8_k(const float* in, uint8_t* b) {  // dequant.rs:1002-1030
  float amax = 0.0f;
  for (int i = 0; i < 256; i++) amax = fmaxr(amax, std::fabs(in[i]));
  float d = amax / 127.0f;
  float id = d != 0.0f ? 1.0f / d : 0.0f;
  std::memcpy(b, &d, 4);
  int8_t* qs = (int8_t*)(b + 4);
  for (int i = 0; i < 256; i++) qs[i] = (int8_t)clampf(rround(in[i] * id), -127.0f, 127.0f);
  for (int i = 0; i < 16; i++) {
    int sum = 0;
    for (int j = 0; j < 16; j++) sum += qs[i * 16 + j];
    int16_t s16 = (int16_t)clampi(sum, -32768, 32767);
    std::memcpy(b + 260 + i * 2, &s16, 2);
  }
}

// ------------------------------------------------------------------ fused dots (whole rows)

float dot_q4_0(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:931-951
  float sum = 0.0f;
  size_t off = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 18) {
    float d = h2f(ld16(w));
    const uint8_t* qs = w + 2;
    float acc_lo = 0.0f, acc_hi = 0.0f;
    for (int i = 0; i < 16; i++) {
      acc_lo += (float)((int)(qs[i] & 0x0F) - 8) * x[off + i];
      acc_hi += (float)((int)((qs[i] >> 4) & 0x0F) - 8) * x[off + i + 16];
    }
    sum += d * (acc_lo + acc_hi);
    off += 32;
  }
  return sum;
}

float dot_q8_0(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:954-971
  float sum = 0.0f;
  size_t off = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 34) {
    float d = h2f(ld16(w));
    const int8_t* qs = (const int8_t*)(w + 2);
    float acc = 0.0f;
    for (int i = 0; i < 32; i++) acc += (float)qs[i] * x[off + i];
    sum += d * acc;
    off += 32;
  }
  return sum;
}

float dot_q4_k(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:978-1032
  float sum = 0.0f;
  size_t xo = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 144) {
    float d = h2f(ld16(w)), dmin = h2f(ld16(w + 2));
    uint8_t scales[8], mins[8];
    unpack_scales_k4(w + 4, scales, mins);
    const uint8_t* qs = w + 16;
    int qp = 0, is = 0;
    for (int g = 0; g < 4; g++) {
      float d1 = d * (float)scales[is], m1 = dmin * (float)mins[is];
      float d2 = d * (float)scales[is + 1], m2 = dmin * (float)mins[is + 1];
      float q_acc1 = 0.0f, x_acc1 = 0.0f;
      for (int l = 0; l < 32; l++) {
        float q = (float)(qs[qp + l] & 0x0F);
        q_acc1 += q * x[xo + l];
        x_acc1 += x[xo + l];
      }
      sum += d1 * q_acc1 - m1 * x_acc1;
      xo += 32;
      float q_acc2 = 0.0f, x_acc2 = 0.0f;
      for (int l = 0; l < 32; l++) {
        float q = (float)((qs[qp + l] >> 4) & 0x0F);
        q_acc2 += q * x[xo + l];
        x_acc2 += x[xo + l];
      }
      sum += d2 * q_acc2 - m2 * x_acc2;
      xo += 32;
      qp += 32;
      is += 2;
    }
  }
  return sum;
}

float dot_q5_k(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:1035-1095
  float sum = 0.0f;
  size_t xo = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 176) {
    float d = h2f(ld16(w)), dmin = h2f(ld16(w + 2));
    uint8_t scales[8], mins[8];
    unpack_scales_k4(w + 4, scales, mins);
    const uint8_t* qh = w + 16;
    const uint8_t* qs = w + 48;
    int qp = 0, is = 0;
    uint8_t u1 = 1, u2 = 2;
    for (int g = 0; g < 4; g++) {
      float d1 = d * (float)scales[is], m1 = dmin * (float)mins[is];
      float d2 = d * (float)scales[is + 1], m2 = dmin * (float)mins[is + 1];
      float q_acc1 = 0.0f, x_acc1 = 0.0f;
      for (int l = 0; l < 32; l++) {
        float lo4 = (float)(qs[qp + l] & 0x0F);
        float hi5 = (qh[l] & u1) ? 16.0f : 0.0f;
        q_acc1 += (lo4 + hi5) * x[xo + l];
        x_acc1 += x[xo + l];
      }
      sum += d1 * q_acc1 - m1 * x_acc1;
      xo += 32;
      float q_acc2 = 0.0f, x_acc2 = 0.0f;
      for (int l = 0; l < 32; l++) {
        float hi4 = (float)((qs[qp + l] >> 4) & 0x0F);
        float hi5 = (qh[l] & u2) ? 16.0f : 0.0f;
        q_acc2 += (hi4 + hi5) * x[xo + l];
        x_acc2 += x[xo + l];
      }
      sum += d2 * q_acc2 - m2 * x_acc2;
      xo += 32;
      qp += 32;
      is += 2;
      u1 = (uint8_t)(u1 << 2);
      u2 = (uint8_t)(u2 << 2);
    }
  }
  return sum;
}

float dot_q6_k(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:1098-1146
  float sum = 0.0f;
  size_t xo = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 210) {
    const int8_t* sc = (const int8_t*)(w + 192);
    float d = h2f(ld16(w + 208));
    for (int n = 0; n < 2; n++) {
      const uint8_t* ql = w + n * 64;
      const uint8_t* qh = w + 128 + n * 32;
      for (int l = 0; l < 32; l++) {
        int is = l / 16, q[4];
        q6k_quad(ql, qh, l, q);
        size_t ob = xo + (size_t)n * 128;
        float s1 = (float)sc[n * 8 + is], s2 = (float)sc[n * 8 + is + 2];
        float s3 = (float)sc[n * 8 + is + 4], s4 = (float)sc[n * 8 + is + 6];
        sum += d * s1 * (float)q[0] * x[ob + l];
        sum += d * s2 * (float)q[1] * x[ob + l + 32];
        sum += d * s3 * (float)q[2] * x[ob + l + 64];
        sum += d * s4 * (float)q[3] * x[ob + l + 96];
      }
    }
    xo += 256;
  }
  return sum;
}

float dot_q8_k(const uint8_t* w, size_t nb, const float* x) {  // simd.rs:1149-1166
  float sum = 0.0f;
  size_t off = 0;
  for (size_t bi = 0; bi < nb; bi++, w += 292) {
    float d = ldf32(w);
    const int8_t* qs = (const int8_t*)(w + 4);
    float acc = 0.0f;
    for (int i = 0; i < 256; i++) acc += (float)qs[i] * x[off + i];
    sum += d * acc;
    off += 256;
  }
  return sum;
}

typedef void (*deq_fn)(const uint8_t*, float*);
typedef void (*q_fn)(const float*, uint8_t*);

deq_fn deq_for(int type) {
  switch (type) {
    case ORC_Q4_0: return deq_q4_0; case ORC_Q4_1: return deq_q4_1;
    case ORC_Q5_0: return deq_q5_0; case ORC_Q5_1: return deq_q5_1;
    case ORC_Q8_0: return deq_q8_0; case ORC_Q8_1: return deq_q8_1;
    case ORC_Q2_K: return deq_q2_k; case ORC_Q3_K: return deq_q3_k;
    case ORC_Q4_K: return deq_q4_k; case ORC_Q5_K: return deq_q5_k;
    case ORC_Q6_K: return deq_q6_k; case ORC_Q8_K: return deq_q8_k;
    default: return nullptr;
  }
}

q_fn q_for(int type) {
  switch (type) {
    case ORC_Q4_0: return q_q4_0; case ORC_Q4_1: return q_q4_1;
    case ORC_Q5_0: return q_q5_0; case ORC_Q5_1: return q_q5_1;
    case ORC_Q8_0: return q_q8_0; case ORC_Q8_1: return q_q8_1;
    case ORC_Q2_K: return q_q2_k; case ORC_Q3_K: return q_q3_k;
    case ORC_Q4_K: return q_q4_k; case ORC_Q5_K: return q_q5_k;
    case ORC_Q6_K: return q_q6_k; case ORC_Q8_K: return q_q8_k;
    default: return nullptr;
  }
}

}  // namespace

extern "C" {

size_t orc_block_size(int type) {  // tensor/dtype.rs:50-74
  switch (type) {
    case ORC_F32: case ORC_F16: case ORC_BF16: return 1;
    case ORC_Q4_0: case ORC_Q4_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_Q8_0: case ORC_Q8_1: return 32;
    case ORC_Q2_K: case ORC_Q3_K: case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: case ORC_Q8_K: return 256;
    default: return 0;
  }
}

size_t orc_block_bytes(int type) {  // tensor/dtype.rs:77-108, blocks.rs:293-305
  switch (type) {
    case ORC_F32: return 4; case ORC_F16: case ORC_BF16: return 2;
    case ORC_Q4_0: return 18; case ORC_Q4_1: return 20; case ORC_Q5_0: return 22; case ORC_Q5_1: return 24;
    case ORC_Q8_0: return 34; case ORC_Q8_1: return 36;
    case ORC_Q2_K: return 84; case ORC_Q3_K: return 110; case ORC_Q4_K: return 144;
    case ORC_Q5_K: return 176; case ORC_Q6_K: return 210; case ORC_Q8_K: return 292;
    default: return 0;
  }
}

uint16_t orc_f32_to_f16(float f) { return f2h(f); }
float orc_f16_to_f32(uint16_t h) { return h2f(h); }

int orc_quantize(int type, const float* in, size_t n, void* out) {
  q_fn f = q_for(type);
  size_t bs = orc_block_size(type), bb = orc_block_bytes(type);
  if (!f || n % bs) return 1;
  uint8_t* o = (uint8_t*)out;
  for (size_t i = 0; i < n / bs; i++) f(in + i * bs, o + i * bb);
  return 0;
}

int orc_dequantize(int type, const void* in, size_t n, float* out) {
  const uint8_t* p = (const uint8_t*)in;
  if (type == ORC_F32) { std::memcpy(out, in, n * 4); return 0; }
  if (type == ORC_F16) { for (size_t i = 0; i < n; i++) out[i] = h2f(ld16(p + 2 * i)); return 0; }
  if (type == ORC_BF16) {
    for (size_t i = 0; i < n; i++) { uint32_t b = (uint32_t)ld16(p + 2 * i) << 16; std::memcpy(out + i, &b, 4); }
    return 0;
  }
  deq_fn f = deq_for(type);
  size_t bs = orc_block_size(type), bb = orc_block_bytes(type);
  if (!f || n % bs) return 1;
  for (size_t i = 0; i < n / bs; i++) f(p + i * bb, out + i * bs);
  return 0;
}

int orc_has_fused_dot(int type) {  // ops.rs:1132-1181
  return type == ORC_Q4_0 || type == ORC_Q8_0 || type == ORC_Q4_K || type == ORC_Q5_K || type == ORC_Q6_K ||
         type == ORC_Q8_K;
}

float orc_dot_q(int type, const void* blocks, const float* x, size_t k) {
  const uint8_t* w = (const uint8_t*)blocks;
  switch (type) {
    case ORC_Q4_0: return dot_q4_0(w, k / 32, x);
    case ORC_Q8_0: return dot_q8_0(w, k / 32, x);
    case ORC_Q4_K: return dot_q4_k(w, k / 256, x);
    case ORC_Q5_K: return dot_q5_k(w, k / 256, x);
    case ORC_Q6_K: return dot_q6_k(w, k / 256, x);
    case ORC_Q8_K: return dot_q8_k(w, k / 256, x);
    default: return std::numeric_limits<float>::quiet_NaN();
  }
}

}  // extern "C"
