// oracle/turboquant.cpp — TEST INFRASTRUCTURE ONLY (see oracle.h): a CPU restatement of the reference's TurboQuant KV cache,
// the quantized cache its `--kv-cache-type turboquant2 | turboquant3 (tq2 | tq3)` selects (src/config.rs:808-817 ->
// KVCacheType::TurboQuantMSE { bits }, src/model/mod.rs:182-213 -> TurboQuantKVCache, src/engine.rs:861-866).
//
//   rotation   src/model/turboquant/rotation.rs:58-130   sign flip, in-place Walsh-Hadamard butterfly, 1/sqrt(d) — and its inverse
//   codebook   src/model/turboquant/codebook.rs:21-263   Lloyd-Max centroids / boundaries for N(0,1) scaled by 1/sqrt(d), scalar
//                                                        quantizer, 1- / 2- / 3-bit packing, dot_with_packed
//   engine     src/model/turboquant/quant.rs:46-175      compress = rotate + quantize_vector; attention_scores
//   cache      src/model/kv_turboquant.rs:88-201         write_kv, attention_head (softmax_inplace quant.rs:228-242, the
//                                                        weight < 1e-8 skip, one inverse rotation PER POSITION), attention_layer
//
//   qjl        src/model/turboquant/qjl.rs:21-178       TurboQuantProd (`tq2-qjl | tq3-qjl`): sign bits of S r for the residual r of
//                                                        the scalar quantizer + its norm; asymmetric inner-product estimator
//
// The random sign vector of a rotation is an INPUT here (HadamardRotation::signs(), rotation.rs:126-129, exists for exactly that:
// "useful for CUDA upload"): the reference draws it from rand's StdRng, which is not restated.  The same holds for the QJL
// projector's Gaussian matrix S (qjl.rs regenerates it row by row from StdRng + rand_distr::StandardNormal on every call): it is
// an INPUT, [dim][dim] f32 in the order the reference draws it (row i = the i-th projection, column j).  The reference sums the
// estimator's sign-weighted dot product with whatever SIMD width the host has (backend/cpu/simd.rs:1361-1374: AVX2 8 lanes, NEON,
// or the scalar loop of qjl.rs:151-178); this restatement is the scalar loop.
// Pinned by the reference's own unit tests of these files (tests/test_oracle_kat.py: codebook.rs:274-348, rotation.rs:137-229
// — the ones that do not depend on the RNG stream — kv_turboquant.rs:289-428, qjl.rs:230-250, simd.rs:1302-1350, quant.rs:340-352);
// everything else rests on the restatement.
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

// codebook.rs:20-49
const float kLloyd1[2] = {-0.7978845608f, 0.7978845608f};
const float kLloyd2[4] = {-1.5102326f, -0.4528427f, 0.4528427f, 1.5102326f};
const float kLloyd3[8] = {-2.1521645f, -1.3441838f, -0.7561303f, -0.2453404f, 0.2453404f, 0.7561303f, 1.3441838f, 2.1521645f};
const float kBound1[1] = {0.0f};
const float kBound2[3] = {-0.98153765f, 0.0f, 0.98153765f};
const float kBound3[7] = {-1.74817415f, -1.05015705f, -0.50073535f, 0.0f, 0.50073535f, 1.05015705f, 1.74817415f};

// fast_walsh_hadamard (rotation.rs:113-130)
void fwht(float* data, size_t n) {
  for (size_t half = 1; half < n; half *= 2)
    for (size_t bs = 0; bs < n; bs += half * 2)
      for (size_t i = 0; i < half; i++) {
        const float a = data[bs + i], b = data[bs + i + half];
        data[bs + i] = a + b;
        data[bs + i + half] = a - b;
      }
}

}  // namespace

extern "C" {

size_t orc_tq_padded_dim(size_t dim) {   // usize::next_power_of_two (rotation.rs:31-33)
  size_t p = 1;
  while (p < dim) p *= 2;
  return p;
}

// Codebook::new (codebook.rs:55-77): centroids[2^bits], boundaries[2^bits - 1], both scaled by 1 / sqrt(dim) (f32)
int orc_tq_codebook(size_t dim, int bits, float* centroids, float* boundaries) {
  if (bits < 1 || bits > 3) return 1;
  const float inv_sqrt_d = 1.0f / std::sqrt((float)dim);
  const float* c = bits == 1 ? kLloyd1 : bits == 2 ? kLloyd2 : kLloyd3;
  const float* b = bits == 1 ? kBound1 : bits == 2 ? kBound2 : kBound3;
  for (int i = 0; i < (1 << bits); i++) centroids[i] = c[i] * inv_sqrt_d;
  for (int i = 0; i < (1 << bits) - 1; i++) boundaries[i] = b[i] * inv_sqrt_d;
  return 0;
}

// Codebook::quantize (codebook.rs:79-92)
uint8_t orc_tq_quantize(const float* boundaries, int bits, float val) {
  uint8_t idx = 0;
  for (int i = 0; i < (1 << bits) - 1; i++) {
    if (val >= boundaries[i]) idx++;
    else break;
  }
  return idx;
}

size_t orc_tq_packed_bytes(int bits, size_t count) {   // codebook.rs:255-262
  return bits == 1 ? (count + 7) / 8 : bits == 2 ? (count + 3) / 4 : (count + 7) / 8 * 3;
}

// Codebook::quantize_vector (codebook.rs:131-170)
void orc_tq_quantize_vector(size_t dim, int bits, const float* data, size_t count, uint8_t* out) {
  float cen[8], bnd[7];
  orc_tq_codebook(dim, bits, cen, bnd);
  size_t o = 0;
  if (bits == 1) {
    for (size_t c0 = 0; c0 < count; c0 += 8) {
      uint8_t byte = 0;
      for (size_t i = 0; i < 8 && c0 + i < count; i++) byte |= (uint8_t)(orc_tq_quantize(bnd, bits, data[c0 + i]) << i);
      out[o++] = byte;
    }
  } else if (bits == 2) {
    for (size_t c0 = 0; c0 < count; c0 += 4) {
      uint8_t byte = 0;
      for (size_t i = 0; i < 4 && c0 + i < count; i++) byte |= (uint8_t)(orc_tq_quantize(bnd, bits, data[c0 + i]) << (i * 2));
      out[o++] = byte;
    }
  } else {
    for (size_t c0 = 0; c0 < count; c0 += 8) {   // every 8 indices = 3 bytes (24 bits)
      uint32_t acc = 0;
      for (size_t i = 0; i < 8 && c0 + i < count; i++) acc |= (uint32_t)orc_tq_quantize(bnd, bits, data[c0 + i]) << (i * 3);
      out[o++] = (uint8_t)(acc & 0xFF);
      out[o++] = (uint8_t)((acc >> 8) & 0xFF);
      out[o++] = (uint8_t)((acc >> 16) & 0xFF);
    }
  }
}

// the index of element i of a packed vector (the unpacking of dequantize_vector / dot_with_packed, codebook.rs:173-252)
static inline uint8_t tq_index(const uint8_t* packed, int bits, size_t i) {
  if (bits == 1) return (packed[i / 8] >> (i % 8)) & 1;
  if (bits == 2) return (packed[i / 4] >> ((i % 4) * 2)) & 3;
  const uint8_t* t = packed + (i / 8) * 3;
  const uint32_t acc = (uint32_t)t[0] | (uint32_t)t[1] << 8 | (uint32_t)t[2] << 16;
  return (uint8_t)((acc >> ((i % 8) * 3)) & 7);
}

// Codebook::dequantize_vector (codebook.rs:173-215)
void orc_tq_dequantize_vector(size_t dim, int bits, const uint8_t* packed, size_t count, float* out) {
  float cen[8], bnd[7];
  orc_tq_codebook(dim, bits, cen, bnd);
  for (size_t i = 0; i < count; i++) out[i] = cen[tq_index(packed, bits, i)];
}

// Codebook::dot_with_packed (codebook.rs:217-252): sum += query[i] * centroid[idx_i], strictly sequential f32
float orc_tq_dot_with_packed(size_t dim, int bits, const float* query, const uint8_t* packed, size_t count) {
  float cen[8], bnd[7];
  orc_tq_codebook(dim, bits, cen, bnd);
  float sum = 0.0f;
  for (size_t i = 0; i < count; i++) sum += query[i] * cen[tq_index(packed, bits, i)];
  return sum;
}

// HadamardRotation::rotate (rotation.rs:58-76): x[dim] -> out[padded_dim]
void orc_tq_rotate(const float* x, size_t dim, const float* signs, float* out) {
  const size_t pd = orc_tq_padded_dim(dim);
  for (size_t i = 0; i < dim; i++) out[i] = x[i] * signs[i];
  for (size_t i = dim; i < pd; i++) out[i] = 0.0f;
  fwht(out, pd);
  const float norm = 1.0f / std::sqrt((float)pd);
  for (size_t i = 0; i < pd; i++) out[i] *= norm;
}

// HadamardRotation::rotate_inverse (rotation.rs:80-96): x[padded_dim] -> out[dim]
void orc_tq_rotate_inverse(const float* x, size_t dim, const float* signs, float* out) {
  const size_t pd = orc_tq_padded_dim(dim);
  std::vector<float> buf(pd);
  const float scale = std::sqrt((float)pd);
  for (size_t i = 0; i < pd; i++) buf[i] = x[i] * scale;
  fwht(buf.data(), pd);
  const float inv_d = 1.0f / (float)pd;
  for (size_t i = 0; i < dim; i++) out[i] = buf[i] * inv_d * signs[i];
}

// TurboQuantEngine::compress without QJL (quant.rs:71-103): packed[packed_bytes(bits, padded_dim)]
void orc_tq_compress(const float* x, size_t dim, int bits, const float* signs, uint8_t* packed) {
  const size_t pd = orc_tq_padded_dim(dim);
  std::vector<float> rot(pd);
  orc_tq_rotate(x, dim, signs, rot.data());
  orc_tq_quantize_vector(pd, bits, rot.data(), pd, packed);
}

// TurboQuantKVCache::attention_head (kv_turboquant.rs:127-172) for one query head over kv_len compressed positions of its kv head.
// k_codes / v_codes: [kv_len][packed_bytes]; signs_k / signs_v: the rotations of that (layer, kv head).
void orc_tq_attention_head(const float* query, const uint8_t* k_codes, const uint8_t* v_codes, size_t kv_len, size_t dim, int bits,
                           const float* signs_k, const float* signs_v, float scale, float* out) {
  const size_t pd = orc_tq_padded_dim(dim), pb = orc_tq_packed_bytes(bits, pd);
  std::vector<float> rot_q(pd), scores(kv_len);
  orc_tq_rotate(query, dim, signs_k, rot_q.data());                                     // attention_scores (quant.rs:133-147)
  for (size_t p = 0; p < kv_len; p++) scores[p] = orc_tq_dot_with_packed(pd, bits, rot_q.data(), k_codes + p * pb, pd);
  for (size_t p = 0; p < kv_len; p++) scores[p] *= scale;
  if (kv_len) {                                                                          // softmax_inplace (quant.rs:228-242)
    float mx = -INFINITY;
    for (size_t p = 0; p < kv_len; p++) mx = std::fmax(mx, scores[p]);
    float sum = 0.0f;
    for (size_t p = 0; p < kv_len; p++) { scores[p] = std::exp(scores[p] - mx); sum += scores[p]; }
    const float inv = 1.0f / sum;
    for (size_t p = 0; p < kv_len; p++) scores[p] *= inv;
  }
  for (size_t i = 0; i < dim; i++) out[i] = 0.0f;
  std::vector<float> deq(pd), orig(dim);
  for (size_t p = 0; p < kv_len; p++) {
    const float w = scores[p];
    if (w < 1e-8f) continue;
    orc_tq_dequantize_vector(pd, bits, v_codes + p * pb, pd, deq.data());
    orc_tq_rotate_inverse(deq.data(), dim, signs_v, orig.data());
    for (size_t i = 0; i < dim; i++) out[i] += w * orig[i];
  }
}

// ---- QJL (TurboQuantProd) ----------------------------------------------------------------------------------------------------

// dot_with_sign_bits (qjl.rs:151-178): sum += values[i] * (bit i set ? +1 : -1), strictly sequential f32
float orc_tq_dot_with_sign_bits(const float* values, const uint64_t* bits, size_t count) {
  float sum = 0.0f;
  for (size_t i = 0; i < count; i++) sum += values[i] * (((bits[i / 64] >> (i % 64)) & 1) ? 1.0f : -1.0f);
  return sum;
}

// QjlProjector::project_query (qjl.rs:100-114) with the matrix given: out[i] = sum_j S[i][j] * q[j], sequential, unfused
void orc_tq_qjl_project(const float* S, size_t dim, const float* q, float* out) {
  for (size_t i = 0; i < dim; i++) {
    float dot = 0.0f;
    for (size_t j = 0; j < dim; j++) dot += S[i * dim + j] * q[j];
    out[i] = dot;
  }
}

// QjlProjector::compress (qjl.rs:36-62): bits[(dim + 63) / 64] (bit i = (S x)_i >= 0), *norm = l2_norm(x) (qjl.rs:180-182)
void orc_tq_qjl_compress(const float* S, size_t dim, const float* x, uint64_t* bits, float* norm) {
  float ss = 0.0f;
  for (size_t i = 0; i < dim; i++) ss += x[i] * x[i];
  *norm = std::sqrt(ss);
  for (size_t w = 0; w < (dim + 63) / 64; w++) bits[w] = 0;
  for (size_t i = 0; i < dim; i++) {
    float dot = 0.0f;
    for (size_t j = 0; j < dim; j++) dot += S[i * dim + j] * x[j];
    if (dot >= 0.0f) bits[i / 64] |= (uint64_t)1 << (i % 64);
  }
}

// QjlProjector::inner_product_fast (qjl.rs:120-131): sqrt(pi / 2) / dim * key_norm * sum
float orc_tq_qjl_inner_product_fast(size_t dim, const float* projected_query, const uint64_t* bits, float key_norm) {
  const float coeff = std::sqrt(1.57079632679489661923f) / (float)dim;   // std::f32::consts::FRAC_PI_2.sqrt() / dim
  return coeff * key_norm * orc_tq_dot_with_sign_bits(projected_query, bits, dim);
}

// TurboQuantEngine::bytes_per_entry (quant.rs:176-186)
size_t orc_tq_bytes_per_entry(size_t dim, int bits, int use_qjl) {
  const size_t pd = orc_tq_padded_dim(dim), cb = orc_tq_packed_bytes(bits, pd);
  return use_qjl ? cb + (pd + 63) / 64 * 8 + 4 : cb;
}

// TurboQuantEngine::compress with QJL (quant.rs:71-103): codes as orc_tq_compress, then the residual rotated - dequantized
// through the projector.  qjl_bits[(padded_dim + 63) / 64]
void orc_tq_compress_qjl(const float* x, size_t dim, int bits, const float* signs, const float* S, uint8_t* packed, uint64_t* qjl_bits,
                         float* residual_norm) {
  const size_t pd = orc_tq_padded_dim(dim);
  std::vector<float> rot(pd), deq(pd), res(pd);
  orc_tq_rotate(x, dim, signs, rot.data());
  orc_tq_quantize_vector(pd, bits, rot.data(), pd, packed);
  orc_tq_dequantize_vector(pd, bits, packed, pd, deq.data());
  for (size_t i = 0; i < pd; i++) res[i] = rot[i] - deq[i];
  orc_tq_qjl_compress(S, pd, res.data(), qjl_bits, residual_norm);
}

// attention_head for TurboQuantProd: the scores are attention_scores with the projector (quant.rs:133-168: polar + correction, the
// query projected once); the V side is unchanged — the reference stores QJL bits for V rows too but never reads them
// (kv_turboquant.rs:154-170 dequantizes V through the codebook only).  k_qjl: [kv_len][(pd + 63) / 64] words, k_norm: [kv_len]
void orc_tq_attention_head_qjl(const float* query, const uint8_t* k_codes, const uint64_t* k_qjl, const float* k_norm, const uint8_t* v_codes,
                               size_t kv_len, size_t dim, int bits, const float* signs_k, const float* signs_v, const float* S_k, float scale,
                               float* out) {
  const size_t pd = orc_tq_padded_dim(dim), pb = orc_tq_packed_bytes(bits, pd), nw = (pd + 63) / 64;
  std::vector<float> rot_q(pd), proj_q(pd), scores(kv_len);
  orc_tq_rotate(query, dim, signs_k, rot_q.data());
  orc_tq_qjl_project(S_k, pd, rot_q.data(), proj_q.data());
  for (size_t p = 0; p < kv_len; p++) {
    const float polar = orc_tq_dot_with_packed(pd, bits, rot_q.data(), k_codes + p * pb, pd);
    scores[p] = polar + orc_tq_qjl_inner_product_fast(pd, proj_q.data(), k_qjl + p * nw, k_norm[p]);
  }
  for (size_t p = 0; p < kv_len; p++) scores[p] *= scale;
  if (kv_len) {
    float mx = -INFINITY;
    for (size_t p = 0; p < kv_len; p++) mx = std::fmax(mx, scores[p]);
    float sum = 0.0f;
    for (size_t p = 0; p < kv_len; p++) { scores[p] = std::exp(scores[p] - mx); sum += scores[p]; }
    const float inv = 1.0f / sum;
    for (size_t p = 0; p < kv_len; p++) scores[p] *= inv;
  }
  for (size_t i = 0; i < dim; i++) out[i] = 0.0f;
  std::vector<float> deq(pd), orig(dim);
  for (size_t p = 0; p < kv_len; p++) {
    const float w = scores[p];
    if (w < 1e-8f) continue;
    orc_tq_dequantize_vector(pd, bits, v_codes + p * pb, pd, deq.data());
    orc_tq_rotate_inverse(deq.data(), dim, signs_v, orig.data());
    for (size_t i = 0; i < dim; i++) out[i] += w * orig[i];
  }
}

}  // extern "C"
