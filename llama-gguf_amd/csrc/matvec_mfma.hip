// matvec_mfma.hip — quantized mat-vec on the int8 matrix cores (v_mfma_i32_16x16x64_i8): Q4_K, Q5_K, Q6_K, Q8_0, Q4_0.
//
// Why MFMA for a mat-VEC.  The VALU formulation (matvec.hip) needs ~3.6 vector instructions per weight
// (byte->f32 convert, FMA, nibble masks); measured on MI355X a wave64 v_cvt_f32_ubyte costs ~2.0 ns and a
// v_fma ~1.2 ns of SIMD time with 4 waves per SIMD (tools/probes/mfma_i8_probe.hip), which made Llama-3-8B
// Q4_K_M decode ISSUE-bound at ~0.76 ms/token — a launch took 21 us whether its 66 MB of weights came from HBM or
// from the Infinity Cache.  The matrix pipe is a separate issue port: one 16x16x64 int8 MFMA multiplies 1024
// weights in ~15 ns, so the multiply-accumulates move there and the VALU only unpacks quants and applies scales.
//
// Arithmetic (the reference keeps x in f32, src/backend/cpu/simd.rs:978-1032; there is no activation quantization
// to mirror): x arrives as XQ records (xq.h) — per 16-element chunk x = s * 2^-30 * I with I = d3*2^24 + d2*2^16 +
// d1*2^8 + d0, balanced int8 digits, exact for every element within 2^6 of the chunk maximum.  The A operand of an
// MFMA is block-diagonal over its four k-chunks (rows 4c..4c+3 = the four digits of chunk c), 16 weight rows x 64
// quants are the B operand, and D holds sum_k q_k * digit_i,k as exact int32 — lane group c gets chunk c.  The four
// digit sums are recombined once in f32 (V = D0*2^24 + D1*2^16 + D2*2^8 + D3) and scaled by d*sc*s*2^-30; min terms
// (Q4_K, Q5_K) and offsets (Q6_K -32, Q4_0 -8) go through the f32 sums of x per chunk, the reference's x_acc.
// Error vs the reference's sequential f32 sum is at f32 rounding level and independent of summation order.
//
// Device layout "tile16" (same bytes as GGUF, rows padded to 16): a tile = 16 rows x one 256-element block = 2304 B.
// The nibbles of MFMA step pp (elements 64pp .. 64pp+63) and k-chunk c (16 elements) of row n belong to lane 16c + n,
// eight bytes per step:  N0, N1 with byte t = w[t] | w[t+4] << 4  (w = elements 0-7 / 8-15 of the chunk), so B-operand
// dwords are N0 & 0x0F.., N0 >> 4 & 0x0F.., N1 & 0x0F.., N1 >> 4 & 0x0F.. and every lane loads only its own nibbles:
//   [0, 1024) lane-major 16 B: N0,N1 of steps 0,1    [1024, 2048) steps 2,3    [2048, 2304) the 16 native 16-byte headers
#include <algorithm>
#include <type_traits>

#include "device_utils.h"
#include "mv_epilogue.h"
#include "xq.h"
#include "mvq_core.h"
#include "timeline.h"

LGH_TL_DEFINE(mvq)

namespace lgh {

#ifdef LGH_STAMPS
__device__ unsigned long long g_stamps[8192 * 8];
hipError_t mvq_read_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#define LGH_STAMP(i)                                                                             \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// per-WAVE stamps (8 slots): how far apart the waves of a workgroup run and where a wave's prologue goes
__device__ unsigned long long g_wstamps[2048 * 16 * 8];
hipError_t mvq_read_wave_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps), n * sizeof(unsigned long long));
}
// first workgroup start / last workgroup end of each of the last 64 launches (slot = MvLaunch::dbg_slot)
__device__ unsigned long long g_span[64 * 2];
hipError_t mvq_spans(unsigned long long* host, int reset) {
  if (reset) {
    unsigned long long init[128];
    for (int i = 0; i < 64; i++) { init[2 * i] = ~0ull; init[2 * i + 1] = 0; }
    return hipMemcpyToSymbol(HIP_SYMBOL(g_span), init, sizeof(init));
  }
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_span), 128 * sizeof(unsigned long long));
}
#define LGH_SPAN(i)                                                                              \
  do {                                                                                           \
    if (threadIdx.x == 0) {                                                                      \
      if ((i) == 0) atomicMin(&g_span[L.dbg_slot * 2], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
      else atomicMax(&g_span[L.dbg_slot * 2 + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime()); \
    }                                                                                            \
  } while (0)
#define LGH_WSTAMP(i)                                                                            \
  do {                                                                                           \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048)                                            \
      g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define LGH_STAMP(i)
#define LGH_WSTAMP(i)
#define LGH_SPAN(i)
#endif

// ------------------------------------------------------------------------------------------------
// native [row][block] Q4_K  ->  tile16
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_q4k_t16_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                            uint32_t n_rows, uint32_t nblk, uint64_t total) {
  // one thread per (row, block, step pp): idx = ((row * nblk) + b) * 4 + pp.  Elements 64pp .. 64pp+63 of the block live in
  // qs[32pp .. 32pp+31]: low nibbles = the first 32, high nibbles = the second 32 (dequant.rs:232-255).
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint32_t pp = (uint32_t)(idx & 3);
    const uint64_t rb = idx >> 2;
    const uint32_t row = (uint32_t)(rb / nblk), b = (uint32_t)(rb % nblk);
    const uint32_t rt = row >> 4, n = row & 15;
    uint8_t* tile = dst + ((size_t)rt * nblk + b) * kTileBytes;
    const bool live = row < n_rows;
    const uint8_t* src = raw + ((size_t)row * nblk + b) * 144;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
      uint32_t w[16];
#pragma unroll
      for (uint32_t t = 0; t < 16; t++) {
        const uint32_t byte = live ? src[16 + 32 * pp + 16 * (c & 1) + t] : 0u;
        w[t] = c < 2 ? (byte & 15u) : (byte >> 4);
      }
      uint32_t N0 = 0, N1 = 0;
#pragma unroll
      for (uint32_t t = 0; t < 4; t++) {
        N0 |= (w[t] | w[t + 4] << 4) << (8 * t);
        N1 |= (w[t + 8] | w[t + 12] << 4) << (8 * t);
      }
      uint32_t* nib = reinterpret_cast<uint32_t*>(tile + (pp >> 1) * 1024 + (16 * c + n) * 16 + (pp & 1) * 8);
      nib[0] = N0;
      nib[1] = N1;
    }
    // the native 16-byte header {d, dmin, scales[12]}, one dword per pp-thread
    *reinterpret_cast<uint32_t*>(tile + 2048 + n * 16 + pp * 4) = live ? *reinterpret_cast<const uint32_t*>(src + pp * 4) : 0u;
  }
}

hipError_t repack_q4k_t16_launch(const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st) {
  const uint64_t total = (uint64_t)((n_rows + 15) / 16) * 16 * nblk * 4;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(repack_q4k_t16_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, raw, dst, n_rows, nblk, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// native [row][block] Q6_K  ->  tile16 (3392 B per 16 rows x 256 elements)
//
// The 6-bit weights q' = ql | qh << 4 (0..63; the reference subtracts 32, dequant.rs:338-341) of MFMA step pp
// (elements 64pp .. 64pp+63) and k-chunk c (16 elements) of row n belong to lane 16c + n.  Per lane and step:
//   N0, N1   byte t = w[t] | w[t+4] << 4          (w = low nibbles of elements 0-7 / 8-15 of the chunk)
//   H        byte t = f[t] | f[t+4] << 2 | f[t+8] << 4 | f[t+12] << 6   (f = the 2 high bits)
// so B-operand dword j (elements 4j..4j+3) = nibbles_j | ((H >> 2j) & 0x03030303) << 4 with plain 32-bit ops.
//   [0    , 1024)  lane-major 16 B: N0,N1 of steps 0 and 1        [1024, 2048)  N0,N1 of steps 2 and 3
//   [2048 , 3072)  lane-major 16 B: H of steps 0..3
//   [3072 , 3328)  row-major 16 B: the 16 int8 sub-block scales    [3328 , 3360)  the 16 rows' f16 d   (+32 B pad)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t q6_raw(const uint8_t* b, uint32_t i) {   // q' of element i of a native block
  const uint32_t hn = i >> 7, r = i & 127, t = r >> 5, l = r & 31;
  const uint32_t lb = b[hn * 64 + l + 32 * (t & 1)];
  const uint32_t lo = t < 2 ? (lb & 0x0Fu) : (lb >> 4);
  return lo | (((b[128 + hn * 32 + l] >> (2 * t)) & 3u) << 4);
}

__global__ void __launch_bounds__(256) repack_q6k_t16_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                            uint32_t n_rows, uint32_t nblk, uint64_t total) {
  // one thread per (row, block, step pp): idx = ((row * nblk) + b) * 4 + pp
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint32_t pp = (uint32_t)(idx & 3);
    const uint64_t rb = idx >> 2;
    const uint32_t row = (uint32_t)(rb / nblk), b = (uint32_t)(rb % nblk);
    const uint32_t rt = row >> 4, n = row & 15;
    uint8_t* tile = dst + ((size_t)rt * nblk + b) * kTileBytesQ6;
    const bool live = row < n_rows;
    const uint8_t* src = raw + ((size_t)row * nblk + b) * 210;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
      uint32_t w[16];
#pragma unroll
      for (uint32_t t = 0; t < 16; t++) w[t] = live ? q6_raw(src, 64 * pp + 16 * c + t) : 0u;
      uint32_t N0 = 0, N1 = 0, H = 0;
#pragma unroll
      for (uint32_t t = 0; t < 4; t++) {
        N0 |= ((w[t] & 15u) | (w[t + 4] & 15u) << 4) << (8 * t);
        N1 |= ((w[t + 8] & 15u) | (w[t + 12] & 15u) << 4) << (8 * t);
        H |= ((w[t] >> 4) | (w[t + 4] >> 4) << 2 | (w[t + 8] >> 4) << 4 | (w[t + 12] >> 4) << 6) << (8 * t);
      }
      const uint32_t lane = 16 * c + n;
      uint32_t* nib = reinterpret_cast<uint32_t*>(tile + (pp >> 1) * 1024 + lane * 16 + (pp & 1) * 8);
      nib[0] = N0;
      nib[1] = N1;
      *reinterpret_cast<uint32_t*>(tile + 2048 + lane * 16 + pp * 4) = H;
    }
    // scales 4pp .. 4pp+3 of the row; d once per (row, block)
    uint32_t sc = 0;
    if (live) sc = (uint32_t)src[192 + 4 * pp] | (uint32_t)src[193 + 4 * pp] << 8 | (uint32_t)src[194 + 4 * pp] << 16 | (uint32_t)src[195 + 4 * pp] << 24;
    *reinterpret_cast<uint32_t*>(tile + 3072 + n * 16 + pp * 4) = sc;
    if (pp == 0) *reinterpret_cast<uint16_t*>(tile + 3328 + n * 2) = live ? (uint16_t)(src[208] | src[209] << 8) : (uint16_t)0;
    if (pp == 1 && n == 0) {   // the pad, so that repacked buffers are fully defined
#pragma unroll
      for (int i = 0; i < 8; i++) *reinterpret_cast<uint32_t*>(tile + 3360 + 4 * i) = 0u;
    }
  }
}

hipError_t repack_q6k_t16_launch(const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st) {
  const uint64_t total = (uint64_t)((n_rows + 15) / 16) * 16 * nblk * 4;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(repack_q6k_t16_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, raw, dst, n_rows, nblk, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------

// ------------------------------------------------------------------------------------------------
// Q5_K, Q8_0, Q4_0 -> tile16.  Same lane mapping as above (step pp, k-chunk c, row n -> lane 16c + n).
//   Q5_K (2816 B): the Q4_K tile (nibble words [0,2048), native headers [2048,2304)) + [2304,2816) lane-major 8 B: the fifth
//         bits, one dword per step PAIR: byte t = f[t] | f[t+4]<<1 | f[t+8]<<2 | f[t+12]<<3 of the even step in the low
//         nibble, of the odd step in the high nibble
//   Q8_0 (4352 B): [pp][lane] 16 quants as they are (int8) [0,4096), then per row the 8 f16 d of its 32-element blocks
//   Q4_0 (2304 B): nibble words [0,2048) as Q4_K (unsigned 0..15; the reference's "- 8" goes through the sums of x), then
//         per row the 8 f16 d
// One thread per (row, 256-element block, step).
// ------------------------------------------------------------------------------------------------
template <int F>
__global__ void __launch_bounds__(256) repack_t16_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                        uint32_t n_rows, uint32_t nblk, uint64_t total) {
  constexpr uint32_t TB = fmt_tile_bytes(F);
  constexpr uint32_t SRC = F == F_Q5K ? 176u : F == F_Q80 ? 8u * 34u : 8u * 18u;   // native bytes of 256 elements
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint32_t pp = (uint32_t)(idx & 3);
    const uint64_t rb = idx >> 2;
    const uint32_t row = (uint32_t)(rb / nblk), b = (uint32_t)(rb % nblk);
    const uint32_t rt = row >> 4, n = row & 15;
    uint8_t* tile = dst + ((size_t)rt * nblk + b) * TB;
    const bool live = row < n_rows;
    const uint8_t* src = raw + ((size_t)row * nblk + b) * SRC;
    auto quant = [&](uint32_t e) -> uint32_t {   // unsigned quant (Q8_0: the byte) of element e of the 256
      if (!live) return 0u;
      if (F == F_Q5K) {   // dequant.rs:265-316
        const uint32_t g = e >> 6, w = e & 63, l = w & 31, is = 2 * g + (w >> 5);
        const uint32_t byte = src[48 + g * 32 + l];
        return (w < 32 ? (byte & 15u) : (byte >> 4)) | (((src[16 + l] >> is) & 1u) << 4);
      }
      const uint32_t sb = e >> 5, i = e & 31;   // 32-element block and offset
      if (F == F_Q80) return src[sb * 34 + 2 + i];                       // dequant.rs:106-112
      const uint32_t byte = src[sb * 18 + 2 + (i & 15)];               // Q4_0, dequant.rs:16-30
      return i < 16 ? (byte & 15u) : (byte >> 4);
    };
    uint32_t hpair[4] = {0, 0, 0, 0};
#pragma unroll
    for (uint32_t c = 0; c < 4; c++) {
      uint32_t w[16];
#pragma unroll
      for (uint32_t t = 0; t < 16; t++) w[t] = quant(64 * pp + 16 * c + t);
      const uint32_t lane = 16 * c + n;
      if (F == F_Q80) {
        uint32_t* q = reinterpret_cast<uint32_t*>(tile + pp * 1024 + lane * 16);
#pragma unroll
        for (uint32_t j = 0; j < 4; j++) q[j] = w[4 * j] | w[4 * j + 1] << 8 | w[4 * j + 2] << 16 | w[4 * j + 3] << 24;
      } else {
        uint32_t N0 = 0, N1 = 0, H = 0;
#pragma unroll
        for (uint32_t t = 0; t < 4; t++) {
          N0 |= ((w[t] & 15u) | (w[t + 4] & 15u) << 4) << (8 * t);
          N1 |= ((w[t + 8] & 15u) | (w[t + 12] & 15u) << 4) << (8 * t);
          H |= ((w[t] >> 4) | (w[t + 4] >> 4) << 1 | (w[t + 8] >> 4) << 2 | (w[t + 12] >> 4) << 3) << (8 * t);
        }
        uint32_t* nib = reinterpret_cast<uint32_t*>(tile + (pp >> 1) * 1024 + lane * 16 + (pp & 1) * 8);
        nib[0] = N0;
        nib[1] = N1;
        hpair[c] = H;
      }
    }
    if (F == F_Q5K) {
      // the fifth bits of this step go into the low (even step) or high (odd step) nibbles of the pair's dword: two
      // threads (pp, pp^1) write disjoint nibbles of the same dword, so each does a byte-wise read-modify-write of ITS
      // nibbles only -> use 4-bit-disjoint atomicOr on a zero-initialised destination
#pragma unroll
      for (uint32_t c = 0; c < 4; c++)
        atomicOr(reinterpret_cast<uint32_t*>(tile + 2304 + (16 * c + n) * 8 + (pp >> 1) * 4), hpair[c] << (4 * (pp & 1)));
      *reinterpret_cast<uint32_t*>(tile + 2048 + n * 16 + pp * 4) = live ? *reinterpret_cast<const uint32_t*>(src + pp * 4) : 0u;
    } else {
      // two f16 d per step thread: blocks 2pp, 2pp+1
      const uint32_t bs = F == F_Q80 ? 34u : 18u, hoff = F == F_Q80 ? 4096u : 2048u;
      uint32_t d2 = 0;
      if (live) d2 = (uint32_t)(src[(2 * pp) * bs] | src[(2 * pp) * bs + 1] << 8) | (uint32_t)(src[(2 * pp + 1) * bs] | src[(2 * pp + 1) * bs + 1] << 8) << 16;
      *reinterpret_cast<uint32_t*>(tile + hoff + n * 16 + pp * 4) = d2;
    }
  }
}

hipError_t repack_t16_launch(int dev_type, const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st) {
  const uint64_t total = (uint64_t)((n_rows + 15) / 16) * 16 * nblk * 4;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  const dim3 g((uint32_t)blocks), t(256);
  switch (fmt_of_dev_type(dev_type)) {
    case F_Q4K: return repack_q4k_t16_launch(raw, dst, n_rows, nblk, st);
    case F_Q6K: return repack_q6k_t16_launch(raw, dst, n_rows, nblk, st);
    case F_Q5K: {
      hipError_t e = hipMemsetAsync(dst, 0, (size_t)((n_rows + 15) / 16) * nblk * fmt_tile_bytes(F_Q5K), st);   // fifth bits are OR-ed in
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((repack_t16_kernel<F_Q5K>), g, t, 0, st, raw, dst, n_rows, nblk, total);
      break;
    }
    case F_Q80: hipLaunchKernelGGL((repack_t16_kernel<F_Q80>), g, t, 0, st, raw, dst, n_rows, nblk, total); break;
    case F_Q40: hipLaunchKernelGGL((repack_t16_kernel<F_Q40>), g, t, 0, st, raw, dst, n_rows, nblk, total); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// stand-alone f32 -> XQ conversion (xq.h): for vectors whose producer cannot write XQ itself (the token embedding, the
// per-op API, pipeline-stage inputs).  One thread per element; with `nw` the record holds x * nw and ssq_part[chunk]
// receives each 16-element chunk's sum of x^2 (the RMSNorm prologue of the consumer).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) xq_quantize_kernel(const float* __restrict__ x, const float* __restrict__ nw,
                                                          uint8_t* __restrict__ xq, float* __restrict__ ssq_part, uint32_t k) {
  const uint32_t e = blockIdx.x * 256 + threadIdx.x;   // k is a multiple of 256
  const float v = x[e];
  xq_store_chunk(xq, e >> 4, nw ? v * nw[e] : v, ssq_part, v);
}

hipError_t xq_quantize_launch(const float* x, const float* nw, uint8_t* xq, float* ssq_part, uint32_t k, hipStream_t st) {
  if (k == 0 || k % 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(xq_quantize_kernel, dim3(k / 256), dim3(256), 0, st, x, nw, xq, ssq_part, k);
  return hipGetLastError();
}

constexpr int kWaves = 8;   // waves per workgroup: 2 per SIMD, 256 VGPRs each
constexpr int kDepth = 4;   // weight tiles a wave keeps in flight (4 x 2304 B x 8 waves = 72 KiB per CU)

// Geometry: ONE workgroup per CU.  Its waves are T k-slices x G row-groups; wave (ks, rg) owns blocks
// [ks*nbw, (ks+1)*nbw) of k for the Rg tiles of its row-group, i.e. a private stream of npass x Rg x nbw items
// (item = one 16-row tile x one 256-element block = 2304 B).  Nothing is shared between waves until the final
// reduction, so there is NO barrier on the way in: a wave requests its slice of x, puts up to kDepth weight tiles in
// flight right behind it, converts x to limbs in its private LDS region while the tiles travel, and streams.  Few,
// fat waves on purpose: the dispatcher starts the waves of a grid over ~1 us (measured: the 16 waves of a 1024-thread
// workgroup began 0.25 / 0.33 / 0.67 / 1.05 us after launch, 4 at a time), which a barrier turns into idle time.
// A wave issues at most one instruction every ~4 cycles, so a latency-bound prologue costs 1.7 ns PER INSTRUCTION
// (measured: the first, general version spent 590 instructions = 1.0 us before its first vector load).  Hence:
//   * everything launch-uniform is computed on the host, packed into eight scalars and delivered by KERNARG PRELOAD
//     (SGPRs filled at wave launch, no load: -mllvm -amdgpu-kernarg-preload-count=8);
//   * per-segment fields come from `L` in one batch of scalar loads, per-pass fields only for passes that exist.
// geom  = T | G << 8 | nbw << 16 | do_norm << 31         geom2 = nblk | Rg << 16        wbpack = wg_begin[1] | wg_begin[2] << 16
// offA/B/C = byte strides of a workgroup / a row-group / a k-slice inside one pass's tile array
// MASK: the formats (1 << F_*) the instantiation handles; with more than one the segment's type decides at run time
template <uint32_t MASK>
__device__ __forceinline__ void mvq_body(const uint32_t bid, uint32_t wbpack, uint32_t geom, uint32_t geom2, uint32_t L_red_floats,
                                         uint32_t lds_red_off, const MvLaunch& L, uint8_t* smem8) {
  // the position word (RoPE epilogues): its scalar load goes out with the very first kernarg loads, no wait here
  const int* L_pos = L.pos;
  uint32_t pos_now;
  asm volatile("s_load_dword %0, %1, 0x0" : "=s"(pos_now) : "s"(L_pos) : "memory");
  const int s = (int)(bid >= (wbpack & 0xFFFFu)) + (int)(bid >= (wbpack >> 16));   // 0xFFFF = no such segment
  const MvSeg& S = L.seg[s];
  const uint32_t S_nrows = S.n_rows, S_wgb = S.wg_begin, S_head_dim = S.head_dim;
  const int S_npass = S.npass, S_epi = S.epi;
  const float* S_resid = S.resid;
  const float* S_xq_nw = S.xq_nw;
  const float* L_rope_cs = L.rope_cs;
  constexpr bool kSingle = (MASK & (MASK - 1)) == 0;
  constexpr int kDp = kDepth;   // weight tiles in flight per wave
  const int fmt = kSingle ? __builtin_ctz(MASK) : fmt_of_dev_type(S.type);
  auto is = [&](int f) { return ((MASK >> f) & 1u) != 0 && (kSingle || fmt == f); };   // compile-time false for absent formats
  const uint32_t tb = is(F_Q4K) ? fmt_tile_bytes(F_Q4K) : is(F_Q6K) ? fmt_tile_bytes(F_Q6K) : is(F_Q5K) ? fmt_tile_bytes(F_Q5K)
                      : is(F_Q80) ? fmt_tile_bytes(F_Q80) : fmt_tile_bytes(F_Q40);
  const uint32_t S_T = geom & 0xFFu, S_G = (geom >> 8) & 0xFFu, nbw = (geom >> 16) & 0x3FFFu;
  const bool nrm = (geom >> 31) != 0;
  const uint32_t S_nblk = geom2 & 0xFFFFu;
  // tiles per workgroup and row group: launch-uniform for a single matrix, per segment in a fused launch (S_G is a power of two)
  const uint32_t Rg = L.nseg > 1 ? S.rows_per_wg >> (4 + __builtin_ctz(S_G)) : geom2 >> 16;
  const uint32_t S_rpw = 16u * Rg * S_G;
  const uint32_t wg = bid - S_wgb;

  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  LGH_WSTAMP(0);
  LGH_SPAN(0);
  uint32_t ks = wave, rg = 0;                                     // wave = rg * T + ks (at most G - 1 subtractions)
  while (ks >= S_T) { ks -= S_T; rg++; }
  const bool active = rg < S_G;
  const uint32_t blk0 = ks * nbw;
  uint32_t nblk_w = active && blk0 < S_nblk ? min(nbw, S_nblk - blk0) : 0;   // (not const: see the phase loop)
  const uint32_t ntiles = (S_nrows + 15) >> 4;
  const uint32_t tile0 = (wg * S_G + rg) * Rg;
  uint32_t ntile_w = active && tile0 < ntiles ? min(Rg, ntiles - tile0) : 0;
  // LDS: per-wave private copies of the XQ records of its k-slice (1280 B per block, xq.h), then the partial-sum slots
  // and the total sum(x^2)
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem8;   // LDS byte address
  const uint32_t wreg = wave * nbw * kXqRecord;
  const uint8_t* xrec = smem8 + wreg;
  const uint32_t xrec_lds = lds_base + wreg;
  float* red = reinterpret_cast<float*>(smem8 + lds_red_off);
  float* ssq = red + L_red_floats;
  // lane roles inside an MFMA: B operand = weight row n, k-chunk c (16 elements); the A operand is block-diagonal over the
  // four k-chunks — rows 4c'..4c'+3 = the four limbs of chunk c' — because XQ scales every 16 elements; so lane group
  // mq = c of D holds the limb sums of chunk c of weight row n
  const uint32_t n = lane & 15, c = lane >> 4;
  const uint32_t mq = c;

  // per-pass base of this wave's tiles and input vector.  The MoE expert index is a SCALAR load — a vector load would
  // sit in the same in-order queue as the weight tiles.
  const uint64_t woff = ((uint64_t)tile0 * S_nblk + blk0) * tb;
  // (four named scalars each, not arrays: an array indexed by the pass number ends up in scratch memory, and a kernel
  // that touches scratch pays for it at every launch)
  const uint8_t *pb0 = nullptr, *pb1 = nullptr, *pb2 = nullptr, *pb3 = nullptr;
  const uint8_t *px0 = nullptr, *px1 = nullptr, *px2 = nullptr, *px3 = nullptr;
  auto pass_setup = [&](int p, const uint8_t*& pb, const uint8_t*& px) {
    if (p < S_npass) {
      const MvPass& P = S.pass[p];
      const uint8_t* plane = P.plane[0];
      const int* sel = P.sel;
      px = P.xq;
      if (sel) {
        uint32_t e32;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e32) : "s"(sel) : "memory");
        plane += (uint64_t)e32 * P.sel_stride[0];
      }
      pb = plane + woff;
    }
  };
  pass_setup(0, pb0, px0);
  pass_setup(1, pb1, px1);
  pass_setup(2, pb2, px2);
  pass_setup(3, pb3, px3);
  // RMSNorm: the producer of x left partial sums of x^2; wave 0 gathers up to 256 of them now (oldest loads of the wave,
  // first used after the last tile) and the rest, if any, at the end
  const float* L_ssq_part = L.ssq_part;
  const uint32_t L_n_ssq = L.n_ssq_part;
  float ssp[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  auto load_ssp = [&]() {
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (lane + 64 * j < L_n_ssq) ssp[j] = L_ssq_part[lane + 64 * j];
  };
  if (nrm && wave == 0) load_ssp();
  MvEpiPre epi_pre = {0.0f, 0.0f, false};
  mv_epilogue_prefetch_resid(S_epi, S_resid, S_xq_nw, S_nrows, wg, S_rpw, epi_pre);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pos_now));   // long since there: the kernarg batch above was waited for
  mv_epilogue_prefetch_rope(S_epi, pos_now, L_rope_cs, S_head_dim, S_nrows, wg, S_rpw, epi_pre);
  LGH_WSTAMP(1);
  struct Pos { uint32_t p, tl, b; };
  auto issue = [&](const Pos& q, RawT16& r) {   // fmt_loads_per_tile(format) loads (x_finish counts on it)
    const uint8_t* base = q.p == 0 ? pb0 : q.p == 1 ? pb1 : q.p == 2 ? pb2 : pb3;
    mvq_issue_tile<MASK>(fmt, base + ((size_t)q.tl * S_nblk + q.b) * tb, lane, r);
  };

  LGH_STAMP(0);
  // item order: pass-major, then tile, then block (innermost: one accumulator per (pass, tile))
  Pos nx = {0, 0, 0};
  auto advance = [&]() {
    if (++nx.b == nblk_w) { nx.b = 0; if (++nx.tl == ntile_w) { nx.tl = 0; ++nx.p; } }
  };
  float acc = 0.0f;

  auto finish_tile = [&](const Pos& q) {   // last block of a (pass, tile): the four lane groups -> one partial sum per row
    float t = acc + __shfl_xor(acc, 16, 64);
    t += __shfl_xor(t, 32, 64);
    if (mq == 0) red[(size_t)(q.p * S_T + ks) * S_rpw + (rg * Rg + q.tl) * 16 + n] = t;
    acc = 0.0f;
  };
  // one item = 16 weight rows x 256 elements.  Both formats: four MFMAs (one per 64 elements); lane group c then holds,
  // for weight row n, the four limb sums of chunk 4pp + c, recombined to V = sum_k q_k * I_k (exact int, rounded once to f32).
  auto consume = [&](const Pos& q, const RawT16& r) {
    mvq_consume_tile<MASK>(fmt, r, xrec + q.b * kXqRecord, lane, acc);
    if (q.b + 1 == nblk_w) finish_tile(q);
  };

  auto x_of = [&](int p) -> const uint8_t* { return p == 0 ? px0 : p == 1 ? px1 : p == 2 ? px2 : px3; };
  for (int p0 = 0; p0 < S_npass;) {   // phases: runs of passes that share one input vector
    int p1 = p0 + 1;
    while (p1 < S_npass && x_of(p1) == x_of(p0)) p1++;
    const uint32_t npp = (uint32_t)(p1 - p0);
    const uint8_t* xg = x_of(p0) + (size_t)blk0 * kXqRecord;   // this wave's k-slice of XQ records
    nx.p = (uint32_t)p0; nx.tl = 0; nx.b = 0;
    // Opaque to the optimizer (the values do not change): otherwise every shape's loop-invariant bookkeeping — ~300
    // scalar instructions, most of them for paths not taken — is hoisted in front of the first load, at 1.7 ns each.
    asm volatile("" : "+s"(nblk_w), "+s"(ntile_w));
    const uint32_t nitems = npp * ntile_w * nblk_w;

    // ---- x staging: the wave's XQ records go straight into its LDS region by LDS-DMA (1024 B + 256 B per record, no
    // registers, no vector ALU), requested FIRST; the caller's tile issues follow, loads return in order, so x_finish
    // waits for x alone (vmcnt = the weight loads behind it).  Inline asm: with the builtin hipcc drains vmcnt to 0 at
    // the next load it issues.
    auto x_request = [&]() {
      for (uint32_t b = 0; b < nblk_w; b++) {
        const uint8_t* src = xg + b * kXqRecord;
        const uint32_t dst = xrec_lds + b * kXqRecord;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_add_u32 m0, m0, 0x400\n\t"
                     "s_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src + lane * 16), "v"(src + 1024 + lane * 4), "s"(dst) : "memory");
      }
    };
    auto x_finish = [&](auto n_tiles) {
      constexpr int NT = decltype(n_tiles)::value;
      // n_tiles x fmt_loads_per_tile loads were issued behind the x requests
#define LGH_WAIT_BEHIND(LPT)                                                                      \
  do {                                                                                           \
    if constexpr (NT * (LPT) == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");             \
    else if constexpr (NT * (LPT) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");        \
    else if constexpr (NT * (LPT) == 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");        \
    else if constexpr (NT * (LPT) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");        \
    else if constexpr (NT * (LPT) == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");        \
    else if constexpr (NT * (LPT) == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");          \
    else if constexpr (NT * (LPT) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");          \
    else if constexpr (NT * (LPT) == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          \
    else if constexpr (NT * (LPT) == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");          \
    else if constexpr (NT * (LPT) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          \
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                         \
  } while (0)
      if (is(F_Q6K) || is(F_Q80)) LGH_WAIT_BEHIND(5);
      else if (is(F_Q5K)) LGH_WAIT_BEHIND(4);
      else LGH_WAIT_BEHIND(3);
#undef LGH_WAIT_BEHIND
      LGH_WSTAMP(4);
    };

    RawT16 buf[kDp];
    Pos pos[kDp];
    // `run` for a static number of tiles put in flight up front; every path ends with nothing in flight
    auto run = [&](auto n_first) {
      constexpr int NF = decltype(n_first)::value;
      LGH_WSTAMP(2);
      x_request();
#pragma unroll
      for (int j = 0; j < NF; j++) { pos[j] = nx; issue(nx, buf[j]); advance(); }
      x_finish(n_first);
      LGH_WSTAMP(5);
      if constexpr (NF < kDp) {   // that was everything
#pragma unroll
        for (int j = 0; j < NF; j++) consume(pos[j], buf[j]);
      } else {
        uint32_t rem = nitems;       // not yet consumed; kDp of them in flight
        while (rem >= 2 * kDp) {
#pragma unroll
          for (int j = 0; j < kDp; j++) {
            consume(pos[j], buf[j]);
            pos[j] = nx; issue(nx, buf[j]); advance();
          }
          rem -= kDp;
        }
        const uint32_t unissued = rem - kDp;   // 0 .. kDp-1
#pragma unroll
        for (int j = 0; j < kDp; j++) {
          consume(pos[j], buf[j]);
          if ((uint32_t)j < unissued) { pos[j] = nx; issue(nx, buf[j]); advance(); }
        }
#pragma unroll
        for (int j = 0; j < kDp - 1; j++)
          if ((uint32_t)j < unissued) consume(pos[j], buf[j]);
      }
    };
    if (nitems == 0) {
      if (nblk_w == 0 && ntile_w > 0) {
        // a k-slice beyond the last block (T does not divide the block count): its partial-sum slots must read as zero
        for (uint32_t p = (uint32_t)p0; p < (uint32_t)p1; p++)
          for (uint32_t tl = 0; tl < ntile_w; tl++)
            if (mq == 0) red[(size_t)(p * S_T + ks) * S_rpw + (rg * Rg + tl) * 16 + n] = 0.0f;
      }
    } else if (nitems == 1) {
      run(std::integral_constant<int, 1>{});
    } else if (nitems == 2) {
      run(std::integral_constant<int, 2>{});
    } else if (nitems == 3) {
      run(std::integral_constant<int, 3>{});
    } else {
      run(std::integral_constant<int, kDp>{});
    }
    p0 = p1;
    LGH_STAMP(3);
    LGH_WSTAMP(6);
  }
  if (nrm && wave == 0) {   // the producer's partial sums of x^2 -> ssq[0]; the other waves contribute nothing
    float ss = (ssp[0] + ssp[1]) + (ssp[2] + ssp[3]);
    for (uint32_t i = 256 + lane; i < L_n_ssq; i += 64) ss += L_ssq_part[i];
    ss = wave_sum_to_lane63(ss);
    if (lane == 63) ssq[0] = ss;
  } else if (nrm && lane == 0) {
    ssq[wave] = 0.0f;
  }
  __syncthreads();
  LGH_STAMP(4);
  mv_epilogue(L, S, wg, red, ssq, S_T, epi_pre);
  LGH_STAMP(5);
  LGH_WSTAMP(7);
  LGH_SPAN(1);
}

template <uint32_t MASK>
__global__ void __launch_bounds__(kWaves * 64) mvq_kernel(uint32_t wbpack, uint32_t geom, uint32_t geom2, uint32_t L_red_floats,
                                                          uint32_t lds_red_off, const MvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  LGH_TL_BEGIN(mvq, lgh::TL_MVQ, geom2);
  mvq_body<MASK>(blockIdx.x, wbpack, geom, geom2, L_red_floats, lds_red_off, L, smem8);
  LGH_TL_END();
}

// ------------------------------------------------------------------------------------------------
// host: geometry and launch
// ------------------------------------------------------------------------------------------------
hipError_t mvq_plan(uint32_t k, uint32_t n_rows, int npass, MvPlan* plan, uint32_t launch_rows, uint32_t force_tiles) {
  if (k == 0 || k % 256 || n_rows == 0 || npass < 1 || npass > 4) return hipErrorInvalidValue;
  const uint32_t nblk = k / 256, W = kWaves;
  uint32_t T = 1;
  for (uint32_t t = 1; t <= W && t <= nblk; t++)
    if (nblk % t == 0) T = t;                       // largest divisor of nblk that fits the workgroup
  if (T < W / 2 && nblk > W) T = W;                 // awkward block counts: uneven k-slices
  if (T > nblk) T = nblk;
  const uint32_t G = W / T >= 1 ? W / T : 1;
  const uint32_t nbw = (nblk + T - 1) / T;
  if (launch_rows < n_rows) launch_rows = n_rows;
  const uint32_t tiles_launch = (launch_rows + 15) / 16;
  // tiles per workgroup: one workgroup per CU (two or three per CU, each with half / a third of the rows, measured the same
  // to 1 %: 15.5 / 15.6 / 15.7 us for the Llama-3-8B gate-up launch — a launch is bound by its fixed costs, not by waves in flight)
  uint32_t R = force_tiles ? force_tiles : (tiles_launch + kNumCU - 1) / kNumCU;   // (force_tiles: the engine balances mixed-format launches)
  R = (R + G - 1) / G * G;
  const uint32_t threads = T * G * 64;
  const uint32_t rmax = threads / 16 / G * G;         // the epilogue gives every row (pair) a thread
  if (R > rmax) R = rmax;
  plan->units = nbw;
  plan->T = T;
  plan->G = G;
  plan->rows_per_wg = 16 * R;
  plan->n_wg = ((n_rows + 15) / 16 + R - 1) / R;
  plan->threads = threads;
  plan->red_floats = (uint32_t)npass * T * 16 * R;
  return hipSuccess;
}

static uint32_t mvq_red_offset(uint32_t nwaves, uint32_t nbw) { return nwaves * nbw * kXqRecord; }
size_t mvq_lds_bytes(uint32_t nwaves, uint32_t nbw, uint32_t red_floats) {
  return (size_t)mvq_red_offset(nwaves, nbw) + (size_t)red_floats * 4 + 64;
}

template <uint32_t MASK>
static hipError_t mvq_go(const MvLaunch& L, uint32_t n_wg, uint32_t threads, size_t lds, hipStream_t st, uint32_t wbpack, uint32_t geom,
                         uint32_t geom2, uint32_t red_off) {
  static bool attr_set[64] = {};
  if (hipError_t e = lds_opt_in(reinterpret_cast<const void*>(&mvq_kernel<MASK>), 160 * 1024, attr_set); e != hipSuccess) return e;
  hipLaunchKernelGGL((mvq_kernel<MASK>), dim3(n_wg), dim3(threads), lds, st, wbpack, geom, geom2, L.red_floats, red_off, L);
  return hipGetLastError();
}

// launch-uniform geometry of one op, packed for the kernel; returns the format mask (0 = not launchable)
uint32_t mvq_pack(const MvLaunch& L, uint32_t n_wg, uint32_t threads, MvGeom* g, size_t* lds_out) {
  const MvSeg& S0 = L.seg[0];
  const size_t lds = mvq_lds_bytes(threads / 64, S0.units, L.red_floats);
  if (lds > 160 * 1024 || threads == 0 || threads > kWaves * 64 || n_wg == 0 || L.nseg < 1 || L.nseg > 3) return 0;
  uint32_t mask = 0;
  for (int i = 0; i < L.nseg; i++) {   // the launch-uniform geometry travels as scalars
    const MvSeg& Si = L.seg[i];
    // (rows per workgroup may differ between the segments of one launch: a segment in a format with more bytes per tile
    // gets fewer tiles per workgroup; the kernel reads it per segment)
    if (Si.T != S0.T || Si.G != S0.G || Si.units != S0.units || Si.nblk != S0.nblk || Si.rows_per_wg % (16 * S0.G) ||
        (i > 0 && Si.wg_begin >= 0xFFFFu))
      return 0;
    const int f = fmt_of_dev_type(Si.type);
    if (f < 0) return 0;
    mask |= 1u << f;
  }
  const uint32_t Rg = S0.rows_per_wg / 16 / S0.G;
  if (S0.T > 255 || S0.G > 255 || S0.units > 0x3FFF || S0.nblk > 0xFFFF || Rg > 0xFFFF) return 0;
  g->wbpack = (L.nseg > 1 ? L.seg[1].wg_begin : 0xFFFFu) | (L.nseg > 2 ? L.seg[2].wg_begin : 0xFFFFu) << 16;
  g->geom = S0.T | S0.G << 8 | S0.units << 16 | (L.do_norm ? 1u << 31 : 0u);
  g->geom2 = S0.nblk | Rg << 16;
  g->red_floats = L.red_floats;
  g->lds_red_off = mvq_red_offset(threads / 64, S0.units);
  g->n_wg = n_wg;
  *lds_out = lds;
  return mask;
}

hipError_t mvq_launch(const MvLaunch& L, uint32_t n_wg, uint32_t threads, hipStream_t st) {
  MvGeom g;
  size_t lds = 0;
  const uint32_t mask = mvq_pack(L, n_wg, threads, &g, &lds);
  const uint32_t wbpack = g.wbpack, geom = g.geom, geom2 = g.geom2, red_off = g.lds_red_off;
#define LGH_MVQ_CASE(M) case M: return mvq_go<M>(L, n_wg, threads, lds, st, wbpack, geom, geom2, red_off)
  switch (mask) {   // single formats and the two mixes of the "_M" quantisations (fused QKV: V in Q6_K)
    LGH_MVQ_CASE(1u << F_Q4K); LGH_MVQ_CASE(1u << F_Q6K); LGH_MVQ_CASE(1u << F_Q5K); LGH_MVQ_CASE(1u << F_Q80); LGH_MVQ_CASE(1u << F_Q40);
    LGH_MVQ_CASE((1u << F_Q4K) | (1u << F_Q6K)); LGH_MVQ_CASE((1u << F_Q5K) | (1u << F_Q6K));
    default: return hipErrorInvalidValue;   // other mixes: the caller launches one format at a time
  }
#undef LGH_MVQ_CASE
}

uint32_t mvq_tile_bytes(int dev_type) {
  const int f = fmt_of_dev_type(dev_type);
  return f < 0 ? 0u : fmt_tile_bytes(f);
}

uint32_t mvq_format_mask(const MvLaunch& L) {
  uint32_t m = 0;
  for (int i = 0; i < L.nseg; i++) {
    const int f = fmt_of_dev_type(L.seg[i].type);
    if (f < 0) return 0;
    m |= 1u << f;
  }
  return m;
}

}  // namespace lgh
