// matvec_mfma.hip — Q4_K mat-vec on the int8 matrix cores (v_mfma_i32_16x16x64_i8), exact integer dots.
//
// Why MFMA for a mat-VEC.  The VALU formulation (matvec.hip) needs ~3.6 vector instructions per weight
// (byte->f32 convert, FMA, nibble masks); measured on MI355X a wave64 v_cvt_f32_ubyte costs ~2.0 ns and a
// v_fma ~1.2 ns of SIMD time with 4 waves per SIMD (tools/probes/mfma_i8_probe.hip), which makes Llama-3-8B
// Q4_K_M decode ISSUE-bound at ~0.76 ms/token — the same launch takes 21 us whether its 66 MB of weights come
// from HBM or from the Infinity Cache.  The matrix pipe is a separate issue port: one 16x16x64 int8 MFMA
// multiplies 1024 weights in ~15 ns, so the multiply-accumulates move there and the VALU only unpacks nibbles
// (one v_and per 4 weights) and applies scales.
//
// Arithmetic (the reference keeps x in f32, src/backend/cpu/simd.rs:978-1032; there is no activation
// quantization to mirror): every 256-element block of x is split EXACTLY into four signed 7-bit limbs with a
// power-of-two block scale,  x = s * (l1*2^-6 + l2*2^-13 + l3*2^-20 + l4*2^-27),  |residual| <= 2^-28 * s.
// The limbs are the A operand (rows 0-3 = group 2p, rows 4-7 = group 2p+1 of a block-diagonal 16x64 tile), 16
// weight rows x 64 nibbles are the B operand, and D holds sum_k q_k*l_i,k as exact int32.  Per 32-element
// sub-block the four limb sums are recombined in f32 (V = D0*2^24 + D1*2^16 + D2*2^8 + D3) and scaled by
// d*sc*s*2^-30; the min term uses f32 sums of x per sub-block, exactly as the reference's x_acc.
// Error vs the reference's sequential f32 sum is below f32 rounding noise and independent of summation order.
//
// Device layout "tile16" (same bytes as GGUF, rows padded to 16): a tile = 16 rows x one 256-element block =
// 2304 B:  [p=0..3][c'=0..1][row 0..15] 16-byte pieces qs_row[32p+16c' .. +16)  (2048 B), then the 16 native
// 16-byte headers {d, dmin, scales[12]}.  Lane (n = l&15, c = l>>4) of a wave loads the piece (p, c&1, n): low
// nibbles are sub-block 2p (c<2), high nibbles sub-block 2p+1 (c>=2), 16 consecutive elements each.
#include "device_utils.h"
#include "mv_epilogue.h"

namespace lgh {

typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef LGH_STAMPS
__device__ unsigned long long g_stamps[8192 * 8];
hipError_t mvq_read_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(unsigned long long));
}
#define LGH_STAMP(i)                                                                             \
  do {                                                                                           \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
// per-WAVE stamps (slot 0 = wave start, 1 = x staged, before the barrier): how far apart the waves of a workgroup run
__device__ unsigned long long g_wstamps[2048 * 16 * 2];
hipError_t mvq_read_wave_stamps(unsigned long long* host, size_t n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wstamps), n * sizeof(unsigned long long));
}
#define LGH_WSTAMP(i)                                                                            \
  do {                                                                                           \
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 2048)                                            \
      g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + (i)] = __builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define LGH_STAMP(i)
#define LGH_WSTAMP(i)
#endif

constexpr int kTileBytes = 2304;
constexpr int kStageBlocks = 1;  // x blocks a wave keeps in flight while staging

// ------------------------------------------------------------------------------------------------
// native [row][block] Q4_K  ->  tile16
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) repack_q4k_t16_kernel(const uint8_t* __restrict__ raw, uint8_t* __restrict__ dst,
                                                            uint32_t n_rows, uint32_t nblk, uint64_t total) {
  for (uint64_t idx = (uint64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (uint64_t)gridDim.x * 256) {
    const uint32_t row = (uint32_t)(idx / nblk), b = (uint32_t)(idx % nblk);
    const uint32_t rt = row >> 4, n = row & 15;
    uint8_t* tile = dst + ((size_t)rt * nblk + b) * kTileBytes;
    u32x4 hd = {0, 0, 0, 0}, piece[8];
#pragma unroll
    for (int i = 0; i < 8; i++) piece[i] = (u32x4){0, 0, 0, 0};
    if (row < n_rows) {
      const uint8_t* src = raw + ((size_t)row * nblk + b) * 144;
      hd = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
      for (int i = 0; i < 8; i++) piece[i] = *reinterpret_cast<const u32x4*>(src + 16 + 16 * i);  // i = 2p + c'
    }
    *reinterpret_cast<u32x4*>(tile + 2048 + n * 16) = hd;
#pragma unroll
    for (int i = 0; i < 8; i++) *reinterpret_cast<u32x4*>(tile + (i >> 1) * 512 + ((i & 1) * 16 + n) * 16) = piece[i];
  }
}

hipError_t repack_q4k_t16_launch(const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st) {
  const uint64_t total = (uint64_t)((n_rows + 15) / 16) * 16 * nblk;
  uint64_t blocks = (total + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(repack_q4k_t16_kernel, dim3((uint32_t)blocks), dim3(256), 0, st, raw, dst, n_rows, nblk, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// kernel
// ------------------------------------------------------------------------------------------------
struct RawT16 { u32x4 hd; u32x4 q[4]; };

__device__ __forceinline__ float wave_max_all(float v) {  // max over the 64 lanes, in every lane
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  float r = fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0)),
                  __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16)));
  r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32)));
  r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48)));
  return r;
}

// One 256-element block of x -> limbs / sub-block sums / scale in LDS.  Called by a whole wave: lane i owns
// elements 4i..4i+3 of the block.
__device__ __forceinline__ void stage_block(f32x4 v, uint32_t blk, uint32_t lane, int8_t* limbs, float* xsum, float* sxs) {
  const float amax = wave_max_all(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  uint32_t e = (__float_as_uint(amax) >> 23) & 0xFFu;       // biased exponent: amax in [2^(e-127), 2^(e-126))
  e = e < 30u ? 30u : (e > 250u ? 250u : e);                 // vanishing / overflowing blocks: clamp (|x'| stays < 1)
  // x' = x * 2^-(e-126) lies in (-1, 1); I = rint(x' * 2^30) is an int32 with |I| <= 2^30.  Balanced base-256 digits
  // of I (each in [-128, 127], top digit in [-64, 64]) are the four int8 limbs:  adding 0x00808080 biases the three low
  // bytes by +128 with the carries landing where they belong, and the XOR takes the bias back out byte-wise
  // (u - 128 as a signed byte is u ^ 0x80).  I = d3*2^24 + d2*2^16 + d1*2^8 + d0 exactly.
  const float sc30 = __uint_as_float((283u - e) << 23);      // 2^-(e-126) * 2^30
  uint32_t w[4];
  const float xin[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int I = (int)__builtin_rintf(xin[k] * sc30);
    w[k] = ((uint32_t)I + 0x00808080u) ^ 0x00808080u;
  }
  // 4x4 byte transpose: element-major words -> one word per limb holding this lane's 4 consecutive elements
  const uint32_t t0 = __builtin_amdgcn_perm(w[1], w[0], 0x05010400u), t1 = __builtin_amdgcn_perm(w[1], w[0], 0x07030602u);
  const uint32_t u0 = __builtin_amdgcn_perm(w[3], w[2], 0x05010400u), u1 = __builtin_amdgcn_perm(w[3], w[2], 0x07030602u);
  const uint32_t limb[4] = {__builtin_amdgcn_perm(u1, t1, 0x07060302u),    // d3: most significant, limb row 0
                            __builtin_amdgcn_perm(u1, t1, 0x05040100u),    // d2
                            __builtin_amdgcn_perm(u0, t0, 0x07060302u),    // d1
                            __builtin_amdgcn_perm(u0, t0, 0x05040100u)};   // d0
  const uint32_t g = lane >> 3, kin = (lane & 7) * 4;        // sub-block and offset inside it
#pragma unroll
  for (int i = 0; i < 4; i++) *reinterpret_cast<uint32_t*>(limbs + ((size_t)(blk * 8 + g) * 4 + i) * 32 + kin) = limb[i];
  float gs = (v.x + v.y) + (v.z + v.w);                      // f32 sum of the sub-block's x (reference: x_acc, simd.rs:1002-1008)
  gs += dpp_f<0xB1>(gs);
  gs += dpp_f<0x4E>(gs);
  gs += dpp_f<0x141>(gs);                                    // row_half_mirror: the 8 lanes of a sub-block
  if ((lane & 7) == 0) xsum[(blk * 2 + (g & 1)) * 4 + (g >> 1)] = gs;   // layout [blk][mq = g&1][p = g>>1]
  if (lane == 0) sxs[blk] = __uint_as_float((e + 1u - 30u) << 23);      // s * 2^-30,  s = 2^(e-126)
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) mvq_kernel(const MvLaunch L) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem8[];
  const uint32_t K = L.k, nblk_all = K >> 8;
  int8_t* limbs = reinterpret_cast<int8_t*>(smem8);
  const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem8;   // LDS byte address
  float* xsum = reinterpret_cast<float*>(smem8 + (size_t)K * 4);
  float* sxs = xsum + (K >> 5);
  float* red = sxs + ((nblk_all + 3) & ~3u);
  float* ssq = red + L.red_floats;

  int s = 0;
  const uint32_t bid = blockIdx.x;
  if (L.nseg > 1 && bid >= L.seg[1].wg_begin) s = 1;
  if (L.nseg > 2 && bid >= L.seg[2].wg_begin) s = 2;
  const MvSeg& S = L.seg[s];
  // every scalar the prologue needs, read up front so the s_loads go out together (each lazily loaded kernarg
  // field used to cost a separate ~120 ns round trip: 1.1 us before the first vector load)
  const uint32_t S_T = S.T, S_G = S.G, S_units = S.units, S_nblk = S.nblk, S_rpw = S.rows_per_wg, S_nrows = S.n_rows;
  const int S_npass = S.npass;
  const float* S_x0 = S.pass[0].x;
  const uint32_t wg = bid - S.wg_begin;

  const uint32_t tid = threadIdx.x, lane = tid & 63;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = blockDim.x >> 6;
  const bool nrm = L.do_norm != 0;
  for (uint32_t i = tid; i < L.red_floats; i += blockDim.x) red[i] = 0.0f;   // k-slices without blocks leave their slots untouched
  // ---- this wave's share of the workgroup: k-slice ks (contiguous blocks), row-group rg (a run of tiles)
  const bool active = wave < S_T * S_G;
  const uint32_t ks = active ? wave % S_T : 0, rg = active ? wave / S_T : 0;
  const uint32_t nbw = S_units;                                   // blocks per k-slice
  const uint32_t blk0 = ks * nbw;
  const uint32_t nblk_w = active && blk0 < S_nblk ? min(nbw, S_nblk - blk0) : 0;
  const uint32_t R = S_rpw >> 4, Rg = R / S_G;            // tiles per workgroup / per row-group
  const uint32_t ntiles = (S_nrows + 15) >> 4;
  const uint32_t tile0 = wg * R + rg * Rg;
  const uint32_t ntile_w = active && tile0 < ntiles ? min(Rg, ntiles - tile0) : 0;
  const bool has_work = nblk_w > 0 && ntile_w > 0;
  // lane roles inside an MFMA
  const uint32_t n = lane & 15, c = lane >> 4;                    // B: weight row n, k-chunk c; D: row n, limb quad mq = c
  const uint32_t sh = (c >> 1) * 4;                               // c>=2 lanes take the high nibbles
  const uint32_t lane_off_q = ((c & 1) * 16 + n) * 16, lane_off_hd = 2048 + n * 16;
  // A operand: rows m = lane&15: m<4 -> limb m of sub-block 2p (k-chunks 0,1); m in 4..7 -> limb m-4 of sub-block 2p+1
  const bool a_valid = n < 8 && (n >> 2) == (c >> 1);
  const uint32_t a_off = ((c >> 1) * 4 + (n & 3)) * 32 + (c & 1) * 16;   // + (blk*8 + 2p) * 128
  const uint32_t mq = c;                                          // D lanes with mq < 2 hold sub-block 2p + mq

  auto issue = [&](uint32_t p, uint32_t tl, uint32_t b, RawT16& r) {
    const MvPass& P = S.pass[p];
    // MoE expert index: a SCALAR load (its own counter) — a vector load here would make the address computation wait
    // on vmcnt(0), i.e. on every x DMA and weight tile still in flight
    uint32_t e32 = 0;
    if (P.sel) asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e32) : "s"(P.sel) : "memory");
    const uint64_t e = e32;
    const uint8_t* tile = P.plane[0] + e * P.sel_stride[0] + ((size_t)(tile0 + tl) * S_nblk + (blk0 + b)) * kTileBytes;
    r.hd = ldg_nt128(tile + lane_off_hd);
#pragma unroll
    for (int pp = 0; pp < 4; pp++) r.q[pp] = ldg_nt128(tile + pp * 512 + lane_off_q);
  };

  RawT16 A, Bq;
  LGH_STAMP(0);
  LGH_WSTAMP(0);
  // item order: pass-major, then tile, then block (innermost: one accumulator per (pass, tile))
  uint32_t ip = 0, itl = 0, ib = 0;
  auto advance = [&]() {
    if (++ib == nblk_w) { ib = 0; if (++itl == ntile_w) { itl = 0; ++ip; } }
  };

  for (int p0 = 0; p0 < S_npass;) {   // phases: runs of passes that share one input vector
    int p1 = p0 + 1;
    while (p1 < S_npass && S.pass[p1].x == S.pass[p0].x) p1++;
    const uint32_t npp = (uint32_t)(p1 - p0);
    const float* xg = p0 == 0 ? S_x0 : S.pass[p0].x;
    if (p0 > 0) __syncthreads();      // everyone is done reading the previous phase's limbs
    // ---- stage x: block cb is handled by wave (cb % nwaves); the first weight tile is already in flight
    ip = (uint32_t)p0; itl = 0; ib = 0;
    uint32_t ap = 0, atl = 0, ab = 0, bp = 0, btl = 0, bb = 0;
    float ss = 0.0f;
    // pass 1: x (an L2 / Infinity-Cache hit) is requested BEFORE the first weight tile — loads return in order, so
    // the staging waits only for x while the HBM-latency weight loads stay in flight behind it — scaled by the norm
    // weight and parked as f32 in the block's own 1-KiB LDS region
    if (!nrm) {
      // plain input: LDS-DMA, 1 KiB per wave-instruction straight into the block's region — no registers, so every
      // block of this wave is requested back to back instead of one L2 round trip at a time
      // (inline asm: hipcc drains vmcnt to 0 before the next load it issues while a DMA it knows about is in flight)
      for (uint32_t cb = wave; cb < nblk_all; cb += nwaves) {
        const float* src = xg + (size_t)cb * 256 + lane * 4;
        const uint32_t dst = lds_base + cb * 1024;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
      }
    } else {
    for (uint32_t cb0 = wave; cb0 < nblk_all; cb0 += nwaves * kStageBlocks) {
      f32x4 xv[kStageBlocks], wv[kStageBlocks];
#pragma unroll
      for (int j = 0; j < kStageBlocks; j++) {
        const uint32_t cb = cb0 + j * nwaves;
        if (cb < nblk_all) xv[j] = reinterpret_cast<const f32x4*>(xg)[cb * 64 + lane];
      }
      if (nrm) {
#pragma unroll
        for (int j = 0; j < kStageBlocks; j++) {
          const uint32_t cb = cb0 + j * nwaves;
          if (cb < nblk_all) wv[j] = reinterpret_cast<const f32x4*>(L.norm_w)[cb * 64 + lane];
        }
      }
      LGH_STAMP(6);
#pragma unroll
      for (int j = 0; j < kStageBlocks; j++) {
        const uint32_t cb = cb0 + j * nwaves;
        if (cb < nblk_all) {   // wave-uniform
          f32x4 v = xv[j];
          if (nrm) {
            ss = __builtin_fmaf(v.x, v.x, ss);
            ss = __builtin_fmaf(v.y, v.y, ss);
            ss = __builtin_fmaf(v.z, v.z, ss);
            ss = __builtin_fmaf(v.w, v.w, ss);
            v = v * wv[j];
          }
          *reinterpret_cast<f32x4*>(limbs + (size_t)cb * 1024 + lane * 16) = v;
        }
      }
    }
    }
    // the first weight tile goes out here: behind x in the load queue, ahead of the limb arithmetic and the barrier
    // (issuing it before pass 1 costs 20 live registers there and spills at the 128-VGPR budget of 16-wave workgroups)
    if (has_work) {
      ap = ip; atl = itl; ab = ib; issue(ip, itl, ib, A); advance();
      if (!nrm) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");   // the x DMAs precede the tile's 5 loads in the queue
    } else if (!nrm) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // pass 2: each wave turns ITS blocks into limbs in place (same wave wrote the floats; reads complete before the
    // limb stores are issued, and no other wave touches the region)
    for (uint32_t cb = wave; cb < nblk_all; cb += nwaves) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(limbs + (size_t)cb * 1024 + lane * 16);
      stage_block(v, cb, lane, limbs, xsum, sxs);   // LDS ops of one wave execute in order: the read above precedes the stores
    }
    if (nrm && p0 == 0) {
      ss = wave_sum_to_lane63(ss);
      if (lane == 63) ssq[wave] = ss;
    }
    LGH_STAMP(7);
    LGH_WSTAMP(1);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // publish x; weight loads stay in flight
    LGH_STAMP(1);
    if (!has_work) { p0 = p1; continue; }

    // ---- stream
    i32x4 areg[4];
    f32x4 xs4 = {0.0f, 0.0f, 0.0f, 0.0f};
    float sxb = 0.0f;
    uint32_t cur_blk = 0xFFFFFFFFu;
    float acc = 0.0f;
    auto consume = [&](uint32_t p, uint32_t tl, uint32_t b, const RawT16& r) {
      const uint32_t blk = blk0 + b;
      if (blk != cur_blk) {   // wave-uniform: (re)load this block's x limbs, sub-block sums and scale
        cur_blk = blk;
#pragma unroll
        for (int pp = 0; pp < 4; pp++) {
          i32x4 t = {0, 0, 0, 0};
          if (a_valid) t = *reinterpret_cast<const i32x4*>(limbs + (size_t)(blk * 8 + 2 * pp) * 128 + a_off);
          areg[pp] = t;
        }
        xs4 = *reinterpret_cast<const f32x4*>(xsum + (blk * 2 + (mq & 1)) * 4);
        sxb = sxs[blk];
      }
      // 6-bit scales / mins of sub-blocks mq, mq+2, mq+4, mq+6 of row n (packing: dequant.rs:210-223)
      const uint32_t s8 = (mq & 1) * 8;
      const uint32_t a = (r.hd.y >> s8) & 0x00FF00FFu, bq = (r.hd.z >> s8) & 0x00FF00FFu, cq = (r.hd.w >> s8) & 0x00FF00FFu;
      const uint32_t sc01 = a & 0x003F003Fu, mn01 = bq & 0x003F003Fu;
      const uint32_t sc23 = (cq & 0x000F000Fu) | ((a >> 2) & 0x00300030u);
      const uint32_t mn23 = ((cq >> 4) & 0x000F000Fu) | ((bq >> 2) & 0x00300030u);
      const float scf[4] = {ub0(sc01), ub2(sc01), ub0(sc23), ub2(sc23)};
      const float mnf[4] = {ub0(mn01), ub2(mn01), ub0(mn23), ub2(mn23)};
      float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
      for (int pp = 0; pp < 4; pp++) {
        i32x4 bw;
        bw.x = (int)((r.q[pp].x >> sh) & 0x0F0F0F0Fu);
        bw.y = (int)((r.q[pp].y >> sh) & 0x0F0F0F0Fu);
        bw.z = (int)((r.q[pp].z >> sh) & 0x0F0F0F0Fu);
        bw.w = (int)((r.q[pp].w >> sh) & 0x0F0F0F0Fu);
        const i32x4 zero = {0, 0, 0, 0};
        const i32x4 d = __builtin_amdgcn_mfma_i32_16x16x64_i8(areg[pp], bw, zero, 0, 0, 0);
        // lanes mq<2: d = limb sums of sub-block 2pp+mq for row n.  V = D0*2^24 + D1*2^16 + D2*2^8 + D3; both halves
        // are exact in f32 (|D0| <= 64*480, |D1..3| <= 128*480, so |(Da<<8)+Db| < 2^24)
        const float hi = (float)((d.x << 8) + d.y), lo = (float)((d.z << 8) + d.w);
        const float V = __builtin_fmaf(hi, 65536.0f, lo);
        s1 = __builtin_fmaf(scf[pp], V, s1);
        s2 = __builtin_fmaf(mnf[pp], xs4[pp], s2);
      }
      const float dd = h2f(r.hd.x & 0xFFFFu), dmin = h2f(r.hd.x >> 16);
      acc += (dd * sxb) * s1 - dmin * s2;
      if (b + 1 == nblk_w) {   // last block of this (pass, tile): hand the partial sums to the epilogue
        if (mq < 2) red[(size_t)(p * (2 * S_T) + ks * 2 + mq) * S_rpw + (rg * Rg + tl) * 16 + n] = acc;
        acc = 0.0f;
      }
    };

    uint32_t remaining = npp * ntile_w * nblk_w;
#ifdef LGH_STAMPS
    if (remaining > 2) {
      bp = ip; btl = itl; bb = ib; issue(ip, itl, ib, Bq); advance();
      consume(ap, atl, ab, A);
      LGH_STAMP(2);
      ap = ip; atl = itl; ab = ib; issue(ip, itl, ib, A); advance();
      consume(bp, btl, bb, Bq);
      remaining -= 2;
    }
#endif
    while (remaining > 2) {
      bp = ip; btl = itl; bb = ib; issue(ip, itl, ib, Bq); advance();
      consume(ap, atl, ab, A);
      ap = ip; atl = itl; ab = ib; issue(ip, itl, ib, A); advance();
      consume(bp, btl, bb, Bq);
      remaining -= 2;
    }
    if (remaining == 2) {
      bp = ip; btl = itl; bb = ib; issue(ip, itl, ib, Bq);
      consume(ap, atl, ab, A);
      consume(bp, btl, bb, Bq);
    } else {   // remaining == 1
      consume(ap, atl, ab, A);
    }
    p0 = p1;
    LGH_STAMP(3);
  }
  __syncthreads();
  LGH_STAMP(4);
  mv_epilogue(L, S, wg, red, ssq, 2 * S.T);
  LGH_STAMP(5);
}

// ------------------------------------------------------------------------------------------------
// host: geometry and launch
// ------------------------------------------------------------------------------------------------
hipError_t mvq_plan(uint32_t k, uint32_t n_rows, int npass, MvPlan* plan, uint32_t launch_rows) {
  if (k == 0 || k % 256 || n_rows == 0 || npass < 1 || npass > 4) return hipErrorInvalidValue;
  const uint32_t nblk = k / 256, W = 16;
  uint32_t T = 1;
  for (uint32_t t = 1; t <= W && t <= nblk; t++)
    if (nblk % t == 0) T = t;                       // largest divisor of nblk that fits the workgroup
  if (T < 8 && nblk > W) T = W;                     // awkward block counts: uneven k-slices
  if (T > nblk) T = nblk;
  const uint32_t G = W / T >= 1 ? W / T : 1;
  const uint32_t nbw = (nblk + T - 1) / T;
  if (launch_rows < n_rows) launch_rows = n_rows;
  const uint32_t tiles_launch = (launch_rows + 15) / 16;
  uint32_t R = (tiles_launch + kNumCU - 1) / kNumCU;  // tiles per workgroup: one workgroup per CU
  R = (R + G - 1) / G * G;
  plan->units = nbw;
  plan->T = T;
  plan->G = G;
  plan->rows_per_wg = 16 * R;
  plan->n_wg = ((n_rows + 15) / 16 + R - 1) / R;
  plan->threads = T * G * 64;
  plan->red_floats = (uint32_t)npass * 2 * T * 16 * R;
  return hipSuccess;
}

size_t mvq_lds_bytes(uint32_t k, uint32_t red_floats) {
  const uint32_t nblk = k / 256;
  return (size_t)k * 4 + (size_t)(k / 32) * 4 + (size_t)((nblk + 3) & ~3u) * 4 + (size_t)red_floats * 4 + 64;
}

hipError_t mvq_launch(const MvLaunch& L, uint32_t n_wg, uint32_t threads, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mvq_kernel<1024>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const size_t lds = mvq_lds_bytes(L.k, L.red_floats);
  if (lds > 160 * 1024 || threads == 0 || threads > 1024 || n_wg == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL((mvq_kernel<1024>), dim3(n_wg), dim3(threads), lds, st, L);
  return hipGetLastError();
}

}  // namespace lgh
