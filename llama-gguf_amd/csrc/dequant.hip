// dequant.hip — block dequantization, upload-time re-layout and the embedding-row lookup.
//
// Dequantization here is BIT-EXACT with the reference's dequantize_q* (src/tensor/quant/dequant.rs:16-367):
// the same expressions in the same order, compiled with -ffp-contract=off so no multiply-add is fused.
// It serves (a) the embedding lookup — the reference dequantizes the table on the host and uploads one
// row per token (src/backend/cuda/gpu_only.rs:505-518, 849-858); here the table stays quantized in HBM
// and the one row is dequantized on device — and (b) GGUF types without a fused mat-vec, which the
// reference also expands to f32 at upload (src/backend/cuda/dequant_weights.rs:211-231).
#include "device_utils.h"
#include "xq.h"
#include "prefill.h"
#include "timeline.h"

LGH_TL_DEFINE(deq)

namespace lgh {

__device__ __forceinline__ uint32_t ld8(const uint8_t* p) { return *p; }
__device__ __forceinline__ uint32_t ld16u(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
__device__ __forceinline__ float ldh(const uint8_t* p) { return h2f(ld16u(p)); }
__device__ __forceinline__ float ldf(const uint8_t* p) {
  uint32_t b = (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
  return __uint_as_float(b);
}

// (scale, min) j of the 12-byte K-quant packing (dequant.rs:210-223)
__device__ __forceinline__ void k4_scale_min(const uint8_t* s, int j, uint32_t& sc, uint32_t& mn) {
  if (j < 4) {
    sc = s[j] & 0x3F;
    mn = s[j + 4] & 0x3F;
  } else {
    sc = (s[j + 4] & 0x0F) | ((s[j - 4] >> 6) << 4);
    mn = ((s[j + 4] >> 4) & 0x0F) | ((s[j] >> 6) << 4);
  }
}

// element `i` of the block at `b`
__device__ float deq_elem(int type, const uint8_t* b, uint32_t i) {
  switch (type) {
    case LGH_TYPE_Q4_0: {  // dequant.rs:16-30
      float d = ldh(b);
      uint32_t byte = b[2 + (i & 15)];
      int q = (int)(i < 16 ? (byte & 0x0F) : (byte >> 4)) - 8;
      return (float)q * d;
    }
    case LGH_TYPE_Q4_1: {  // dequant.rs:36-48
      float d = ldh(b), m = ldh(b + 2);
      uint32_t byte = b[4 + (i & 15)];
      float q = (float)(i < 16 ? (byte & 0x0F) : (byte >> 4));
      return q * d + m;
    }
    case LGH_TYPE_Q5_0: {  // dequant.rs:54-76
      float d = ldh(b);
      uint32_t qh = ld16u(b + 2) | (ld16u(b + 4) << 16);
      uint32_t byte = b[6 + (i & 15)];
      int q4 = (int)(i < 16 ? (byte & 0x0F) : (byte >> 4));
      int q5 = (int)((qh >> i) & 1);
      return (float)((q4 | (q5 << 4)) - 16) * d;
    }
    case LGH_TYPE_Q5_1: {  // dequant.rs:81-101
      float d = ldh(b), m = ldh(b + 2);
      uint32_t qh = ld16u(b + 4) | (ld16u(b + 6) << 16);
      uint32_t byte = b[8 + (i & 15)];
      uint32_t q4 = i < 16 ? (byte & 0x0F) : (byte >> 4);
      uint32_t q5 = (qh >> i) & 1;
      return (float)(q4 | (q5 << 4)) * d + m;
    }
    case LGH_TYPE_Q8_0: return (float)(int)(int8_t)b[2 + i] * ldh(b);       // dequant.rs:106-112
    case LGH_TYPE_Q8_1: return (float)(int)(int8_t)b[4 + i] * ldf(b);       // dequant.rs:117-123
    case LGH_TYPE_Q2_K: {  // dequant.rs:129-156 (the reference's sequential layout, quirk Q1)
      uint32_t g = i >> 4, w = i & 15;
      uint32_t sm = b[g];
      float d = ldh(b + 80), dmin = ldh(b + 82);
      float d_scale = d * (float)(sm & 0x0F), d_min = dmin * (float)(sm >> 4);
      uint32_t byte = b[16 + g * 4 + (w >> 2)];
      float q = (float)((byte >> ((w & 3) * 2)) & 3);
      return d_scale * q - d_min;
    }
    case LGH_TYPE_Q3_K: {  // dequant.rs:161-200 (sequential layout, quirk Q1)
      uint32_t g = i >> 4;
      const uint8_t* sc = b + 96 + (g >> 2) * 3;
      int b0 = sc[0], b1 = sc[1], b2 = sc[2], s6;
      switch (g & 3) {
        case 0: s6 = b0 & 0x3F; break;
        case 1: s6 = (b0 >> 6) | ((b1 & 0x0F) << 2); break;
        case 2: s6 = (b1 >> 4) | ((b2 & 0x03) << 4); break;
        default: s6 = b2 >> 2; break;
      }
      float scale = ldh(b + 108) * (float)(int)(int8_t)((int8_t)s6 - 32);
      int lo2 = (b[32 + (i >> 2)] >> ((i & 3) * 2)) & 3;
      int hi1 = (b[i >> 3] >> (i & 7)) & 1;
      return scale * (float)((lo2 | (hi1 << 2)) - 4);
    }
    case LGH_TYPE_Q4_K: {  // dequant.rs:205-259
      uint32_t g = i >> 6, w = i & 63, l = w & 31, sc, mn;
      k4_scale_min(b + 4, (int)(2 * g + (w >> 5)), sc, mn);
      float d = ldh(b), dmin = ldh(b + 2);
      float d1 = d * (float)sc, m1 = dmin * (float)mn;
      uint32_t byte = b[16 + g * 32 + l];
      float q = (float)(w < 32 ? (byte & 0x0F) : (byte >> 4));
      return d1 * q - m1;
    }
    case LGH_TYPE_Q5_K: {  // dequant.rs:265-316
      uint32_t g = i >> 6, w = i & 63, l = w & 31, sc, mn;
      uint32_t is = 2 * g + (w >> 5);
      k4_scale_min(b + 4, (int)is, sc, mn);
      float d = ldh(b), dmin = ldh(b + 2);
      float d1 = d * (float)sc, m1 = dmin * (float)mn;
      uint32_t byte = b[48 + g * 32 + l];
      float q4 = (float)(w < 32 ? (byte & 0x0F) : (byte >> 4));
      float hi5 = ((b[16 + l] >> is) & 1) ? 16.0f : 0.0f;
      return d1 * (q4 + hi5) - m1;
    }
    case LGH_TYPE_Q6_K: {  // dequant.rs:322-356
      uint32_t n = i >> 7, r = i & 127, t = r >> 5, l = r & 31;
      const uint8_t* ql = b + n * 64;
      uint32_t qhb = b[128 + n * 32 + l];
      uint32_t lb = ql[l + 32 * (t & 1)];
      uint32_t lo = t < 2 ? (lb & 0x0F) : (lb >> 4);
      int q = (int)(lo | (((qhb >> (2 * t)) & 3) << 4)) - 32;
      float sc = (float)(int)(int8_t)b[192 + n * 8 + (l >> 4) + 2 * t];
      return ldh(b + 208) * sc * (float)q;
    }
    case LGH_TYPE_Q8_K: return (float)(int)(int8_t)b[4 + i] * ldf(b);       // dequant.rs:361-367
    default: return 0.0f;
  }
}

__device__ __forceinline__ float deq_any(int type, const uint8_t* base, uint64_t e) {
  if (type == LGH_TYPE_F32) return ldf(base + e * 4);
  if (type == LGH_TYPE_F16) return ldh(base + e * 2);
  if (type == LGH_TYPE_BF16) return __uint_as_float(ld16u(base + e * 2) << 16);
  uint32_t bs = blk_elems(type);
  return deq_elem(type, base + (e / bs) * blk_bytes(type), (uint32_t)(e % bs));
}

__global__ void __launch_bounds__(256) dequant_kernel(int type, const uint8_t* __restrict__ raw, float* __restrict__ dst,
                                                      uint64_t n) {
  uint64_t e = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (; e < n; e += stride) dst[e] = deq_any(type, raw, e);
}

hipError_t dequant_launch(int src_type, const uint8_t* raw, float* dst, uint64_t n_elems, hipStream_t st) {
  if (!blk_elems(src_type)) return hipErrorInvalidValue;
  uint64_t blocks = (n_elems + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(dequant_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, src_type, raw, dst, n_elems);
  return hipGetLastError();
}

// Embedding lookup (llama.rs:293-307; gpu_only.rs:849-858): dst = dequantize(table row *token).
// Thread 0 of block 0 also opens the token: state[POS] = state[NEXT]++ (every later kernel of this
// token reads state[POS]; stream order makes the store visible).
__global__ void __launch_bounds__(256) embed_kernel(int type, const uint8_t* __restrict__ table, const int* token,
                                                    float* __restrict__ dst, uint32_t hidden, int* state,
                                                    uint8_t* __restrict__ xq, const float* __restrict__ xq_nw, float* __restrict__ xq_ssq) {
  LGH_TL_BEGIN(deq, lgh::TL_EMBED, hidden);
  if (state && blockIdx.x == 0 && threadIdx.x == 0) {
    int p = state[ST_NEXT];
    state[ST_POS] = p;
    state[ST_NEXT] = p + 1;
  }
  const uint64_t row = (uint64_t)(uint32_t)*token;
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < hidden) {
    const float v = deq_any(type, table, row * hidden + i);
    dst[i] = v;
    // the first layer's QKV is an int8-MFMA consumer: leave its input as XQ records too (hidden % 256 == 0, host-checked)
    if (xq) xq_store_chunk(xq, i >> 4, xq_nw ? v * xq_nw[i] : v, xq_ssq, v);
  }
  LGH_TL_END();
}

hipError_t embed_launch(int src_type, const uint8_t* table, const int* token, float* dst, uint32_t hidden, int* state,
                        uint8_t* xq, const float* xq_nw, float* xq_ssq, hipStream_t st) {
  if (!blk_elems(src_type) || (xq && hidden % 256)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_kernel, dim3((hidden + 255) / 256), dim3(256), 0, st, src_type, table, token, dst, hidden,
                     state, xq, xq_nw, xq_ssq);
  return hipGetLastError();
}

// rows tokens[0..m) of the table -> dst[m][hidden] (batched prompt processing, prefill.hip)
// multi-sequence decode: blockIdx.y = sequence s; row tokens[s] -> dst + s * hidden, its XQ image and sum-of-squares partials
// at s * their strides (the arithmetic of embed_kernel; the positions are kept by the host, engine_batch.hip)
__global__ void __launch_bounds__(256) embed_multi_kernel(int type, const uint8_t* __restrict__ table, const int* __restrict__ tokens,
                                                          float* __restrict__ dst, uint32_t hidden, uint8_t* __restrict__ xq,
                                                          const float* __restrict__ xq_nw, float* __restrict__ xq_ssq, uint32_t xq_stride,
                                                          uint32_t ssq_stride) {
  const uint32_t sq = blockIdx.y;
  const uint64_t row = (uint64_t)(uint32_t)tokens[sq];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < hidden) {
    const float v = deq_any(type, table, row * hidden + i);
    dst[(size_t)sq * hidden + i] = v;
    if (xq) xq_store_chunk(xq + (size_t)sq * xq_stride, i >> 4, xq_nw ? v * xq_nw[i] : v, xq_ssq ? xq_ssq + (size_t)sq * ssq_stride : nullptr, v);
  }
}

hipError_t embed_multi_launch(int src_type, const uint8_t* table, const int* tokens, float* dst, uint32_t hidden, uint32_t n_seq, uint8_t* xq,
                              const float* xq_nw, float* xq_ssq, uint32_t xq_stride, uint32_t ssq_stride, hipStream_t st) {
  if (!blk_elems(src_type) || (xq && hidden % 256) || n_seq == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_multi_kernel, dim3((hidden + 255) / 256, n_seq), dim3(256), 0, st, src_type, table, tokens, dst, hidden, xq, xq_nw,
                     xq_ssq, xq_stride, ssq_stride);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) embed_batch_kernel(int type, const uint8_t* __restrict__ table, const int* __restrict__ tokens,
                                                          float* __restrict__ dst, uint32_t hidden) {
  const uint64_t row = (uint64_t)(uint32_t)tokens[blockIdx.y];
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < hidden) dst[(size_t)blockIdx.y * hidden + i] = deq_any(type, table, row * hidden + i);
}

hipError_t embed_batch_launch(int src_type, const uint8_t* table, const int* tokens, float* dst, uint32_t hidden, uint32_t m_tokens,
                              hipStream_t st) {
  if (!blk_elems(src_type) || m_tokens == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_batch_kernel, dim3((hidden + 255) / 256, m_tokens), dim3(256), 0, st, src_type, table, tokens, dst, hidden);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Upload-time re-layout of formats whose blocks are not 16-byte multiples into aligned planes
// (see DevWeight in common.h).  One thread per block, 2-byte moves (all block sizes are even).
// The reference re-lays its quantized weights too ([n][k/bs] -> [k/bs][n], on the host:
// src/backend/cuda/dequant_weights.rs:143-153); this one keeps rows contiguous and runs on device.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void copy2(uint8_t* dst, const uint8_t* src, uint32_t nbytes) {
  for (uint32_t i = 0; i < nbytes; i += 2) *reinterpret_cast<uint16_t*>(dst + i) = *reinterpret_cast<const uint16_t*>(src + i);
}

__global__ void __launch_bounds__(256) repack_kernel(int type, const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                     uint64_t o0, uint64_t o1, uint64_t o2, uint64_t o3, uint64_t n_blocks) {
  uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
  const uint64_t stride = (uint64_t)gridDim.x * 256;
  for (; b < n_blocks; b += stride) {
    if (type == LGH_TYPE_Q6_K) {
      const uint8_t* s = raw + b * 210;
      copy2(dst + o0 + b * 128, s, 128);
      copy2(dst + o1 + b * 64, s + 128, 64);
      copy2(dst + o2 + b * 16, s + 192, 16);
      copy2(dst + o3 + b * 2, s + 208, 2);
    } else if (type == LGH_TYPE_Q8_0) {
      const uint8_t* s = raw + b * 34;
      copy2(dst + o0 + b * 32, s + 2, 32);
      copy2(dst + o1 + b * 2, s, 2);
    } else if (type == LGH_TYPE_Q4_0) {
      const uint8_t* s = raw + b * 18;
      copy2(dst + o0 + b * 16, s + 2, 16);
      copy2(dst + o1 + b * 2, s, 2);
    }
  }
}

hipError_t repack_launch(int src_type, const uint8_t* raw, uint8_t* dst, const uint64_t plane_off[4], uint64_t n_blocks,
                         hipStream_t st) {
  if (src_type != LGH_TYPE_Q6_K && src_type != LGH_TYPE_Q8_0 && src_type != LGH_TYPE_Q4_0) return hipErrorInvalidValue;
  uint64_t blocks = (n_blocks + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (blocks == 0) return hipSuccess;
  hipLaunchKernelGGL(repack_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, src_type, raw, dst, plane_off[0], plane_off[1],
                     plane_off[2], plane_off[3], n_blocks);
  return hipGetLastError();
}

}  // namespace lgh
