// xq.h — "XQ": the activation vector in the form the int8 matrix-core mat-vec consumes (internal, device side).
//
// The reference keeps activations in f32 and so does this engine (residual stream, attention, epilogues).  The MFMA
// mat-vec (matvec_mfma.hip) additionally needs every input vector as exact int8 limbs; converting it inside the
// consumer costs ~65 instructions per 256-element block in EVERY workgroup (256x redundant: 2 us per launch for the
// 14336-wide FFN vector) plus, with an RMSNorm prologue, a second 4 B/element stream for the norm weights.  So the
// kernel that PRODUCES a vector writes its XQ image next to the f32 values, once:
//
//   one 1280-byte record per 256 elements (the consumer copies its records into LDS verbatim with two LDS-DMA
//   instructions per record and never touches them with the vector ALU):
//     [   0,1024)  limbs: byte ((g*4 + i)*32 + e) = digit 3-i (i = 0 most significant) of element 32g + e, where the
//                  balanced base-256 digits d3..d0 of I = rint(x / s_j * 2^30) satisfy I = d3*2^24 + d2*2^16 + d1*2^8 + d0
//     [1024,1088)  xs16[16]  f32 sum of x over 16-element chunk j, stored at [j & 3][j >> 2]
//     [1088,1152)  sx16[16]  f32 s_j * 2^-30, s_j = the power of two above max|x| of chunk j, same order
//     [1152,1280)  zero
//   x = sx16[j] * I exactly for every element within 2^6 of its chunk's maximum (2^-31 of the chunk maximum otherwise).
//   With an RMSNorm in front of the consumer the record holds x * norm_weight (the mat-vec is linear: the 1/rms factor
//   is applied per output row in the consumer's epilogue) and the producer also leaves sum(x^2) per chunk (k/16 partials,
//   which the consumer adds up).
// The scale granularity is 16 elements = one k-chunk of a lane group of v_mfma_i32_16x16x64_i8 = the 16 output rows of
// one weight tile, so every producing workgroup owns whole chunks.
#pragma once

#include "device_utils.h"

namespace lgh {

constexpr uint32_t kXqRecord = 1280;   // bytes per 256 elements
constexpr uint32_t kXqXs16 = 1024, kXqSx16 = 1088;

__host__ __device__ inline size_t xq_bytes(size_t k) { return (k + 255) / 256 * kXqRecord; }

// Called by the 16 consecutive lanes (a DPP row: lane % 16 == element % 16) that hold the 16 elements of chunk
// `chunk` (global index, element / 16) of a vector; `v` is this lane's element (already multiplied by the norm weight
// when the consumer normalises).  Writes the chunk's 64 limb bytes, its sum and its scale.
// With `ssq_part` the chunk's sum of `raw`^2 (the un-normalised values) goes to ssq_part[chunk].
__device__ __forceinline__ void xq_store_chunk(uint8_t* xq, uint32_t chunk, float v, float* ssq_part = nullptr, float raw = 0.0f,
                                               uint32_t tid = threadIdx.x) {
  const uint32_t l16 = tid & 15;
  // max |v| and sum over the 16 lanes, in every lane (quad_perm x2, row_half_mirror, row_mirror)
  float amax = fabsf(v), sum = v;
  float sq = raw * raw;
  amax = fmaxf(amax, dpp_f<0xB1>(amax)); sum += dpp_f<0xB1>(sum); sq += dpp_f<0xB1>(sq);
  amax = fmaxf(amax, dpp_f<0x4E>(amax)); sum += dpp_f<0x4E>(sum); sq += dpp_f<0x4E>(sq);
  amax = fmaxf(amax, dpp_f<0x141>(amax)); sum += dpp_f<0x141>(sum); sq += dpp_f<0x141>(sq);
  amax = fmaxf(amax, dpp_f<0x140>(amax)); sum += dpp_f<0x140>(sum); sq += dpp_f<0x140>(sq);
  uint32_t e = (__float_as_uint(amax) >> 23) & 0xFFu;       // biased exponent: amax in [2^(e-127), 2^(e-126))
  e = e < 30u ? 30u : (e > 250u ? 250u : e);                 // vanishing / overflowing chunks: clamp (|x'| stays < 1)
  const int I = (int)__builtin_rintf(v * __uint_as_float((283u - e) << 23));   // x' * 2^30, |I| <= 2^30
  const uint32_t w = ((uint32_t)I + 0x00808080u) ^ 0x00808080u;               // bytes = balanced digits d0..d3
  // the quad's four words -> this lane's limb word: lane q of the quad builds digit q of elements 4Q..4Q+3
  const uint32_t w0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x00, 0xF, 0xF, true);   // quad_perm [0,0,0,0]
  const uint32_t w1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0x55, 0xF, 0xF, true);   // [1,1,1,1]
  const uint32_t w2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xAA, 0xF, 0xF, true);   // [2,2,2,2]
  const uint32_t w3 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w, 0xFF, 0xF, 0xF, true);   // [3,3,3,3]
  const uint32_t q = l16 & 3, Q = l16 >> 2;
  const uint32_t sel2 = q | (q + 4) << 8;                    // byte q of the low word, byte q of the high word
  const uint32_t lo = __builtin_amdgcn_perm(w1, w0, sel2), hi = __builtin_amdgcn_perm(w3, w2, sel2);
  const uint32_t word = __builtin_amdgcn_perm(hi, lo, 0x05040100u);
  const uint32_t blk = chunk >> 4, j = chunk & 15, g = j >> 1, i = 3 - q;
  uint8_t* rec = xq + (size_t)blk * kXqRecord;
  *reinterpret_cast<uint32_t*>(rec + (g * 4 + i) * 32 + (j & 1) * 16 + Q * 4) = word;
  if (l16 == 0) {
    const uint32_t slot = (j & 3) * 4 + (j >> 2);
    *reinterpret_cast<float*>(rec + kXqXs16 + slot * 4) = sum;
    *reinterpret_cast<float*>(rec + kXqSx16 + slot * 4) = __uint_as_float((e + 1u - 30u) << 23);   // s * 2^-30, s = 2^(e-126)
    if (ssq_part) ssq_part[chunk] = sq;
  }
}

}  // namespace lgh
