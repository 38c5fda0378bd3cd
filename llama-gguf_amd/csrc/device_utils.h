// device_utils.h — small gfx950 device helpers (internal).
#pragma once

#include "common.h"

namespace lgh {

// streamed-once weights: non-temporal 16-byte loads straight to VGPRs (no LDS round trip for GEMV)
__device__ __forceinline__ u32x4 ldg_nt128(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ u32x2 ldg_nt64(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(p));
}
__device__ __forceinline__ uint32_t ldg_nt32(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(p));
}
__device__ __forceinline__ uint16_t ldg_nt16(const void* p) {
  return __builtin_nontemporal_load(reinterpret_cast<const uint16_t*>(p));
}

__device__ __forceinline__ float h2f(uint32_t bits16) {
  _Float16 h;
  uint16_t b = (uint16_t)bits16;
  __builtin_memcpy(&h, &b, 2);
  return (float)h;  // v_cvt_f32_f16, subnormals preserved
}

__device__ __forceinline__ float ub0(uint32_t v) { return (float)(v & 0xFFu); }
__device__ __forceinline__ float ub1(uint32_t v) { return (float)((v >> 8) & 0xFFu); }
__device__ __forceinline__ float ub2(uint32_t v) { return (float)((v >> 16) & 0xFFu); }
__device__ __forceinline__ float ub3(uint32_t v) { return (float)(v >> 24); }

// 4 FMAs of the 4 bytes of `v` against x[0..3]
__device__ __forceinline__ float fma4(uint32_t v, const float* x, float acc) {
  acc = __builtin_fmaf(ub0(v), x[0], acc);
  acc = __builtin_fmaf(ub1(v), x[1], acc);
  acc = __builtin_fmaf(ub2(v), x[2], acc);
  acc = __builtin_fmaf(ub3(v), x[3], acc);
  return acc;
}

// Makes `v` opaque to the optimizer at this point (no instruction is emitted).  Used after nibble masks so
// that `(m >> 8) & 0xFF` stays a single v_cvt_f32_ubyte1 instead of being re-folded into v_bfe_u32 + cvt.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// the 4 bytes of `v` dotted with x[0..3], as a fresh chain (first term is a plain multiply)
__device__ __forceinline__ float fma4z(uint32_t v, const float* x) {
  float acc = ub0(v) * x[0];
  acc = __builtin_fmaf(ub1(v), x[1], acc);
  acc = __builtin_fmaf(ub2(v), x[2], acc);
  acc = __builtin_fmaf(ub3(v), x[3], acc);
  return acc;
}

// full-wave (64 lanes) sum; every lane gets the result; fixed order -> deterministic
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// DPP lane permutation of a float (gfx9 DPP controls): no LDS crossbar, full-rate VALU
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}

// 64-lane sum by DPP; the total is valid in LANE 63 only.  Fixed order -> deterministic.
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_f<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);       // row_half_mirror
  v += dpp_f<0x140>(v);       // row_mirror  -> every lane of a 16-lane row holds the row sum
  v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace lgh
