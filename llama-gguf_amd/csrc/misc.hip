// misc.hip — small decode-path kernels: standalone RMSNorm / RoPE / SiLU*mul (per-op surface and
// fallbacks for shapes the fused epilogues do not cover), device arg-max, MoE router.
#include "device_utils.h"
#include "xq.h"
#include "timeline.h"

LGH_TL_DEFINE(misc)

namespace lgh {

// ---------------------------------------------------------------------------------------------
// RMSNorm  (simd.rs:847-878; CUDA twin rms_norm_fused, kernels.rs:131-168): one workgroup.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) rms_norm_kernel(const float* __restrict__ x, const float* __restrict__ w, float eps,
                                                        float* __restrict__ out, uint32_t n) {
  __shared__ float wsum[16];
  const uint32_t tid = threadIdx.x, nthr = blockDim.x;
  float ss = 0.0f;
  for (uint32_t i = tid; i < n; i += nthr) ss = __builtin_fmaf(x[i], x[i], ss);
  ss = wave_sum(ss);
  if ((tid & 63) == 0) wsum[tid >> 6] = ss;
  __syncthreads();
  float tot = 0.0f;
  for (uint32_t i = 0; i < (nthr >> 6); i++) tot += wsum[i];
  const float inv = 1.0f / __builtin_sqrtf(tot / (float)n + eps);
  for (uint32_t i = tid; i < n; i += nthr) out[i] = x[i] * inv * w[i];
}

hipError_t rms_norm_launch(const float* x, const float* w, float eps, float* out, uint32_t n, hipStream_t st) {
  uint32_t thr = n >= 4096 ? 1024 : 256;
  hipLaunchKernelGGL(rms_norm_kernel, dim3(1), dim3(thr), 0, st, x, w, eps, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// RoPE at one position (ops.rs:1285-1337; CUDA twin rope_single_pos, kernels.rs:379-429).  cos/sin
// come from the host-built table (libm powf/cosf/sinf, the reference's own arithmetic), so given the
// same q/k the rotation is bit-exact.  One thread per pair.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rope_kernel(float* q, float* k, uint32_t n_heads, uint32_t n_kv, uint32_t d,
                                                   const int* pos_ptr, const float* __restrict__ cs, int neox) {
  const uint32_t half = d / 2;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;
  const uint32_t total = (n_heads + n_kv) * half;
  if (t >= total) return;
  const uint32_t head = t / half, i = t % half;
  float* base = head < n_heads ? q + (size_t)head * d : k + (size_t)(head - n_heads) * d;
  const uint32_t pos = (uint32_t)*pos_ptr;
  const float c = cs[((size_t)pos * half + i) * 2], s = cs[((size_t)pos * half + i) * 2 + 1];
  const uint32_t i0 = neox ? i : 2 * i, i1 = neox ? i + half : 2 * i + 1;
  const float x0 = base[i0], x1 = base[i1];
  base[i0] = x0 * c - x1 * s;
  base[i1] = x0 * s + x1 * c;
}

hipError_t rope_launch(float* q, float* k, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, const int* pos,
                       const float* rope_cs, int neox, hipStream_t st) {
  uint32_t total = (n_heads + n_kv) * (head_dim / 2);
  hipLaunchKernelGGL(rope_kernel, dim3((total + 255) / 256), dim3(256), 0, st, q, k, n_heads, n_kv, head_dim, pos, rope_cs,
                     neox);
  return hipGetLastError();
}

// simd.rs:598-649: gate[i] = gate[i] / (1 + exp(-gate[i])) * up[i]
__global__ void __launch_bounds__(256) silu_mul_kernel(const float* __restrict__ gate, const float* __restrict__ up,
                                                       float* __restrict__ out, uint32_t n) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    float g = gate[i];
    out[i] = g / (1.0f + expf(-g)) * up[i];
  }
}

hipError_t silu_mul_launch(const float* gate, const float* up, float* out, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL(silu_mul_kernel, dim3((n + 255) / 256), dim3(256), 0, st, gate, up, out, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// The element-wise / activation / softmax / f32 matmul ops of the reference's per-op `Backend` trait
// (src/backend/mod.rs:29-265; CPU: src/backend/cpu/ops.rs:24-528).  Not on the decode path: they complete the
// trait surface for `select_gpu_backend` (src/engine.rs:738-812).  add / mul / scale and the f32 matmul (ascending k, one
// rounding per step, no contraction) are bit-exact with the CPU backend.
// ---------------------------------------------------------------------------------------------
enum { EW_ADD = 0, EW_MUL = 1, EW_SCALE = 2, EW_SILU = 3, EW_GELU = 4 };

template <int OP>
__global__ void __launch_bounds__(256) ewise_kernel(const float* __restrict__ a, const float* __restrict__ b, float s,
                                                    float* __restrict__ out, uint64_t n) {
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
    const float x = a[i];
    float r;
    if (OP == EW_ADD) r = x + b[i];                                   // ops.rs:24-116
    else if (OP == EW_MUL) r = x * b[i];                              // ops.rs:119-208
    else if (OP == EW_SCALE) r = x * s;                               // ops.rs:211-300
    else if (OP == EW_SILU) r = x / (1.0f + expf(-x));                // ops.rs:303-325
    else {                                                            // ops.rs:328-347
      const float inner = 0.7978846f * (x + 0.044715f * x * x * x);
      r = 0.5f * x * (1.0f + tanhf(inner));
    }
    out[i] = r;
  }
}

hipError_t ewise_launch(int op, const float* a, const float* b, float s, float* out, uint64_t n, hipStream_t st) {
  uint64_t blocks = (n + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  if (blocks == 0) return hipSuccess;
  const dim3 g((uint32_t)blocks), t(256);
  switch (op) {
    case EW_ADD: hipLaunchKernelGGL(ewise_kernel<EW_ADD>, g, t, 0, st, a, b, s, out, n); break;
    case EW_MUL: hipLaunchKernelGGL(ewise_kernel<EW_MUL>, g, t, 0, st, a, b, s, out, n); break;
    case EW_SCALE: hipLaunchKernelGGL(ewise_kernel<EW_SCALE>, g, t, 0, st, a, b, s, out, n); break;
    case EW_SILU: hipLaunchKernelGGL(ewise_kernel<EW_SILU>, g, t, 0, st, a, b, s, out, n); break;
    case EW_GELU: hipLaunchKernelGGL(ewise_kernel<EW_GELU>, g, t, 0, st, a, b, s, out, n); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

// softmax along the last dimension, one workgroup per row (ops.rs:350-385: max, exp(x - max), sum, times 1/sum)
__global__ void __launch_bounds__(256) softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ out, uint32_t last_dim) {
  __shared__ float s_red[4];
  const float* row = x + (size_t)blockIdx.x * last_dim;
  float* orow = out + (size_t)blockIdx.x * last_dim;
  float m = -INFINITY;
  for (uint32_t i = threadIdx.x; i < last_dim; i += 256) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
  __syncthreads();
  float sum = 0.0f;
  for (uint32_t i = threadIdx.x; i < last_dim; i += 256) {
    const float e = expf(row[i] - m);
    orow[i] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sum;
  __syncthreads();
  const float inv = 1.0f / ((s_red[0] + s_red[1]) + (s_red[2] + s_red[3]));
  for (uint32_t i = threadIdx.x; i < last_dim; i += 256) orow[i] *= inv;
}

hipError_t softmax_rows_launch(const float* x, float* out, uint32_t rows, uint32_t last_dim, hipStream_t st) {
  if (rows == 0 || last_dim == 0) return hipSuccess;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(rows), dim3(256), 0, st, x, out, last_dim);
  return hipGetLastError();
}

// row-major [m,k] @ [k,n] (ops.rs:429-528): one thread per output element, products added in ascending k exactly as both
// CPU variants do (the tiled one continues the same running sum across its k-chunks) -> bit-identical results
__global__ void __launch_bounds__(256) matmul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ c,
                                                         uint32_t m, uint32_t k, uint32_t n) {
  const uint32_t j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (j >= n) return;
  float sum = 0.0f;
  for (uint32_t kk = 0; kk < k; kk++) sum += a[(size_t)i * k + kk] * b[(size_t)kk * n + j];
  c[(size_t)i * n + j] = sum;
}

hipError_t matmul_f32_launch(const float* a, const float* b, float* c, uint32_t m, uint32_t k, uint32_t n, hipStream_t st) {
  if (m == 0 || n == 0) return hipSuccess;
  if (m > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(matmul_f32_kernel, dim3((n + 255) / 256, m), dim3(256), 0, st, a, b, c, m, k, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Arg-max with the reference's tie rule: Iterator::max_by returns the LAST maximal element
// (src/main.rs:1815-1821).  Stage 1: 64 workgroups -> partials; stage 2: one workgroup -> state.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void amax_merge(float& bv, int& bi, float v, int i) {
  if (v > bv || (v == bv && i > bi)) { bv = v; bi = i; }
}

__device__ __forceinline__ void amax_block_reduce(float& bv, int& bi, float* sv, int* si) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float ov = __shfl_xor(bv, off, 64);
    int oi = __shfl_xor(bi, off, 64);
    amax_merge(bv, bi, ov, oi);
  }
  const uint32_t tid = threadIdx.x;
  if ((tid & 63) == 0) { sv[tid >> 6] = bv; si[tid >> 6] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (uint32_t w = 1; w < (blockDim.x >> 6); w++) amax_merge(bv, bi, sv[w], si[w]);
  }
}

constexpr int kArgmaxParts = 64;

__global__ void __launch_bounds__(256) argmax_stage1(const float* __restrict__ v, uint32_t n, float* pv, int* pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  LGH_TL_BEGIN(misc, lgh::TL_ARGMAX1, n);
  float bv = -INFINITY;
  int bi = -1;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) amax_merge(bv, bi, v[i], (int)i);
  amax_block_reduce(bv, bi, sv, si);
  if (threadIdx.x == 0) { pv[blockIdx.x] = bv; pi[blockIdx.x] = bi; }
  LGH_TL_END();
}

__global__ void __launch_bounds__(64) argmax_stage2(const float* pv, const int* pi, int nparts, int* state, int* out_token) {
  LGH_TL_BEGIN(misc, lgh::TL_ARGMAX2, (unsigned)nparts);
  float bv = -INFINITY;
  int bi = -1;
  for (int i = threadIdx.x; i < nparts; i += 64) amax_merge(bv, bi, pv[i], pi[i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float ov = __shfl_xor(bv, off, 64);
    int oi = __shfl_xor(bi, off, 64);
    amax_merge(bv, bi, ov, oi);
  }
  if (threadIdx.x == 0) {
    if (bi < 0) bi = 0;
    if (state) { state[ST_ARGMAX] = bi; state[ST_TOKEN] = bi; }  // feed the token back on device
    if (out_token) out_token[state ? state[ST_POS] : 0] = bi;      // token log, indexed by position
  }
  LGH_TL_END();
}

hipError_t argmax_launch(const float* logits, uint32_t n, float* part_val, int* part_idx, int* state, int* out_token,
                         hipStream_t st) {
  hipLaunchKernelGGL(argmax_stage1, dim3(kArgmaxParts), dim3(256), 0, st, logits, n, part_val, part_idx);
  hipLaunchKernelGGL(argmax_stage2, dim3(1), dim3(64), 0, st, part_val, part_idx, kArgmaxParts, state, out_token);
  return hipGetLastError();
}

// multi-sequence decode: blockIdx.y = sequence s; logits v + s * n -> next[s] (last maximal index, as argmax_stage1/2)
__global__ void __launch_bounds__(256) argmax_multi_stage1(const float* __restrict__ v, uint32_t n, float* pv, int* pi) {
  __shared__ float sv[4];
  __shared__ int si[4];
  const uint32_t sq = blockIdx.y;
  v += (size_t)sq * n;
  float bv = -INFINITY;
  int bi = -1;
  for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) amax_merge(bv, bi, v[i], (int)i);
  amax_block_reduce(bv, bi, sv, si);
  if (threadIdx.x == 0) { pv[sq * kArgmaxParts + blockIdx.x] = bv; pi[sq * kArgmaxParts + blockIdx.x] = bi; }
}

__global__ void __launch_bounds__(64) argmax_multi_stage2(const float* pv, const int* pi, int nparts, int* next) {
  const uint32_t sq = blockIdx.x;
  float bv = -INFINITY;
  int bi = -1;
  for (int i = threadIdx.x; i < nparts; i += 64) amax_merge(bv, bi, pv[sq * kArgmaxParts + i], pi[sq * kArgmaxParts + i]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    float ov = __shfl_xor(bv, off, 64);
    int oi = __shfl_xor(bi, off, 64);
    amax_merge(bv, bi, ov, oi);
  }
  if (threadIdx.x == 0) next[sq] = bi < 0 ? 0 : bi;
}

// part_val / part_idx: n_seq * 64 words each
hipError_t argmax_multi_launch(const float* logits, uint32_t n, uint32_t n_seq, float* part_val, int* part_idx, int* next, hipStream_t st) {
  if (n_seq == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(argmax_multi_stage1, dim3(kArgmaxParts, n_seq), dim3(256), 0, st, logits, n, part_val, part_idx);
  hipLaunchKernelGGL(argmax_multi_stage2, dim3(n_seq), dim3(64), 0, st, part_val, part_idx, kArgmaxParts, next);
  return hipGetLastError();
}

// K/V rows of the current token into the caches (layers.rs:577-600; CUDA twin update_kv_cache,
// kernels.rs:800-817) — only used when RoPE + cache write are not fused into the QKV launch.
__global__ void __launch_bounds__(256) kv_store_kernel(const float* __restrict__ k, const float* __restrict__ v, float* kcache,
                                                       float* vcache, uint32_t n_kv, uint32_t d, uint32_t max_seq,
                                                       const int* pos_ptr) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n_kv * d) return;
  const uint32_t h = i / d, j = i % d, pos = (uint32_t)*pos_ptr;
  const size_t dst = ((size_t)h * max_seq + pos) * d + j;
  kcache[dst] = k[i];
  vcache[dst] = v[i];
}

hipError_t kv_store_launch(const float* k, const float* v, float* kcache, float* vcache, uint32_t n_kv, uint32_t d,
                           uint32_t max_seq, const int* pos, hipStream_t st) {
  hipLaunchKernelGGL(kv_store_kernel, dim3((n_kv * d + 255) / 256), dim3(256), 0, st, k, v, kcache, vcache, n_kv, d, max_seq, pos);
  return hipGetLastError();
}

// pipeline stages without an embedding open the token here
__global__ void advance_kernel(int* state) {
  int p = state[ST_NEXT];
  state[ST_POS] = p;
  state[ST_NEXT] = p + 1;
}

hipError_t advance_launch(int* state, hipStream_t st) {
  hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(1), 0, st, state);
  return hipGetLastError();
}

// ---- MoE layers of a multi-sequence step (engine_batch.hip): the step's (sequence, top-k slot) pairs grouped by expert, so that an
// expert's matrices are read ONCE for all sequences that selected it (the reference's BatchedEngine runs MoeLayer::forward,
// moe.rs:321-413, per sequence).  sel: [n_seq][top_k] expert ids (the router's output, selection order); idx[e][j] = the j-th pair
// v = sequence * top_k + slot that chose expert e, sequences ascending; cnt[e] = how many.
__global__ void moe_group_kernel(const int* __restrict__ sel, uint32_t n_seq, uint32_t top_k, uint32_t n_experts, int* __restrict__ cnt,
                                 int* __restrict__ idx, uint32_t idx_stride) {
  const uint32_t e = threadIdx.x;
  if (e >= n_experts) return;
  int c = 0;
  for (uint32_t v = 0; v < n_seq * top_k; v++)   // sel: [n_seq][top_k], the router's multi-token layout
    if ((uint32_t)sel[v] == e) idx[e * idx_stride + c++] = (int)v;
  cnt[e] = c;
}

hipError_t moe_group_launch(const int* sel, uint32_t n_seq, uint32_t top_k, uint32_t n_experts, int* cnt, int* idx, uint32_t idx_stride, hipStream_t st) {
  if (n_experts > 64 || top_k == 0 || top_k > 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moe_group_kernel, dim3(1), dim3(64), 0, st, sel, n_seq, top_k, n_experts, cnt, idx, idx_stride);
  return hipGetLastError();
}

// h[s] = (sum_p w[s][p] * down_p) + h[s], the experts in selection order from a zero-initialised sum (moe.rs:363-368; the
// single-sequence EPI_MOE_DOWN epilogue, mv_epilogue.h, operation for operation), + the XQ image of the new h times the next
// consumer's norm weights and the per-chunk sums of squares.  tmp: [n_seq * top_k][H] expert outputs; grid (H / 256, n_seq).
__global__ void __launch_bounds__(256) moe_combine_kernel(const float* __restrict__ tmp, const float* __restrict__ moe_w, uint32_t top_k, float* __restrict__ hidden,
                                                          uint32_t H, const float* __restrict__ nw, uint8_t* __restrict__ xq, uint32_t xq_stride,
                                                          float* __restrict__ ssq, uint32_t ssq_stride) {
  const uint32_t s = blockIdx.y, row = blockIdx.x * 256 + threadIdx.x;
  if (row >= H) return;
  float acc = 0.0f;
  for (uint32_t p = 0; p < top_k; p++) acc += moe_w[s * top_k + p] * tmp[((size_t)s * top_k + p) * H + row];
  const float outv = acc + hidden[(size_t)s * H + row];
  hidden[(size_t)s * H + row] = outv;
  if (xq) xq_store_chunk(xq + (size_t)s * xq_stride, row >> 4, outv * (nw ? nw[row] : 1.0f), ssq ? ssq + (size_t)s * ssq_stride : nullptr, outv);
}

hipError_t moe_combine_launch(const float* tmp, const float* moe_w, uint32_t top_k, float* hidden, uint32_t H, uint32_t n_seq, const float* nw,
                              uint8_t* xq, uint32_t xq_stride, float* ssq, uint32_t ssq_stride, hipStream_t st) {
  if (H % 256 || n_seq == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moe_combine_kernel, dim3(H / 256, n_seq), dim3(256), 0, st, tmp, moe_w, top_k, hidden, H, nw, xq, xq_stride, ssq, ssq_stride);
  return hipGetLastError();
}

// same-device pipeline hop (hidden vector / token word) as a graph node of the producing stage
__global__ void copy_words_kernel(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, uint32_t n) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

hipError_t copy_words_launch(void* dst, const void* src, uint32_t n, hipStream_t st) {
  hipLaunchKernelGGL(copy_words_kernel, dim3((n + 255) / 256), dim3(256), 0, st, (uint32_t*)dst, (const uint32_t*)src, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// MoE router (moe.rs:128-198; CUDA path gpu_only.rs:1765-1831 does this on the HOST after a D2H):
// logits[e] = <rms_norm(h), W_r[e]>, stable descending sort, top-k, softmax over the k logits.
// One workgroup; wave e handles experts e, e+nwaves, ...
// ---------------------------------------------------------------------------------------------
constexpr int kMaxExperts = 64;

__global__ void __launch_bounds__(512) moe_router_kernel(const float* __restrict__ x, const float* __restrict__ norm_w,
                                                         float eps, const float* __restrict__ w, uint32_t hidden,
                                                         uint32_t n_experts, uint32_t top_k, int* sel, float* sel_w) {
  __shared__ float wsum[16];
  __shared__ float logits[kMaxExperts];
  const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  x += (size_t)blockIdx.x * hidden;   // one workgroup per token: a single one in decode, a block of prompt tokens in prefill.hip
  sel += blockIdx.x * top_k;
  sel_w += blockIdx.x * top_k;
  // Every load of a phase is requested before the first one is used (16-byte loads, 8 in flight per lane and array): the
  // kernel is a handful of memory round trips long.  (One element per iteration — load, fma, next — made this ONE
  // workgroup 31 us long, 28 % of a Mixtral-8x7B token.)
  const bool vec = (hidden & 3u) == 0;
  float inv = 1.0f;
  if (norm_w) {
    float ss = 0.0f;
    if (vec) {
      for (uint32_t i = tid * 4; i < hidden; i += blockDim.x * 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i);
        ss = __builtin_fmaf(v.x, v.x, __builtin_fmaf(v.y, v.y, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.w, v.w, ss))));
      }
    } else {
      for (uint32_t i = tid; i < hidden; i += blockDim.x) ss = __builtin_fmaf(x[i], x[i], ss);
    }
    ss = wave_sum(ss);
    if (lane == 0) wsum[wave] = ss;
    __syncthreads();
    float tot = 0.0f;
    for (uint32_t i = 0; i < nw; i++) tot += wsum[i];
    inv = 1.0f / __builtin_sqrtf(tot / (float)hidden + eps);
  }
  for (uint32_t e = wave; e < n_experts; e += nw) {
    const float* we = w + (size_t)e * hidden;
    float acc = 0.0f;
    if (vec) {
      constexpr int kB = 8;   // 16-byte loads in flight per lane and array
      for (uint32_t base = lane * 4; base < hidden; base += 64 * 4 * kB) {
        f32x4 xv[kB], wv[kB], nv[kB];
#pragma unroll
        for (int j = 0; j < kB; j++) {
          const uint32_t i = base + j * 256;
          const bool ok = i < hidden;
          const uint32_t ic = ok ? i : 0;   // clamped: the loads are unconditional
          xv[j] = *reinterpret_cast<const f32x4*>(x + ic);
          wv[j] = *reinterpret_cast<const f32x4*>(we + ic);
          nv[j] = norm_w ? *reinterpret_cast<const f32x4*>(norm_w + ic) : (f32x4)(1.0f);
          if (!ok) wv[j] = (f32x4)(0.0f);
        }
#pragma unroll
        for (int j = 0; j < kB; j++) {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const float xn = norm_w ? xv[j][q] * inv * nv[j][q] : xv[j][q];   // (x * inv) * w (simd.rs:891-892)
            acc = __builtin_fmaf(xn, wv[j][q], acc);
          }
        }
      }
    } else {
      for (uint32_t i = lane; i < hidden; i += 64) {
        float xv = norm_w ? x[i] * inv * norm_w[i] : x[i];
        acc = __builtin_fmaf(xv, we[i], acc);
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) logits[e] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    // stable descending selection: ties keep the lower expert index first (slice::sort_by is stable)
    uint64_t taken = 0;
    float top[8];
    float mx = -INFINITY;
    for (uint32_t s = 0; s < top_k; s++) {
      int best = -1;
      for (uint32_t e = 0; e < n_experts; e++) {
        if ((taken >> e) & 1) continue;
        if (best < 0 || logits[e] > logits[best]) best = (int)e;
      }
      taken |= 1ull << best;
      sel[s] = best;
      top[s] = logits[best];
      mx = fmaxf(mx, top[s]);
    }
    float exp_sum = 0.0f;
    for (uint32_t s = 0; s < top_k; s++) exp_sum += expf(top[s] - mx);
    for (uint32_t s = 0; s < top_k; s++) sel_w[s] = expf(top[s] - mx) / exp_sum;
  }
}

hipError_t moe_router_launch(const float* x, const float* norm_w, float eps, const float* w, uint32_t hidden,
                             uint32_t n_experts, uint32_t top_k, int* sel, float* sel_w, hipStream_t st, uint32_t n_tokens) {
  if (n_experts > (uint32_t)kMaxExperts || top_k > 8 || top_k > n_experts || n_tokens == 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(moe_router_kernel, dim3(n_tokens), dim3(512), 0, st, x, norm_w, eps, w, hidden, n_experts, top_k, sel, sel_w);
  return hipGetLastError();
}

// Streaming-read probe: every workgroup walks its share of the buffer with 16-byte loads, 16 in flight per lane; the
// xor-reduction only keeps the loads alive.  NT selects non-temporal loads (the weight stream's policy).
template <bool NT>
__global__ void __launch_bounds__(512) hbm_read_kernel(const u32x4* __restrict__ buf, size_t n16, float* __restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * 512;
  u32x4 acc = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 512 + threadIdx.x;
  for (; i + 15 * stride < n16; i += 16 * stride) {
    u32x4 v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = NT ? __builtin_nontemporal_load(buf + i + j * stride) : buf[i + j * stride];
#pragma unroll
    for (int j = 0; j < 16; j++) acc ^= v[j];
  }
  for (; i < n16; i += stride) acc ^= buf[i];
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[blockIdx.x & 4095] = 1.0f;   // never true for the 0x01 fill
}

hipError_t hbm_read_launch(const uint8_t* buf, size_t bytes, float* sink, int nt, hipStream_t st) {
  if (nt) hipLaunchKernelGGL((hbm_read_kernel<true>), dim3(kNumCU * 2), dim3(512), 0, st, reinterpret_cast<const u32x4*>(buf), bytes / 16, sink);
  else hipLaunchKernelGGL((hbm_read_kernel<false>), dim3(kNumCU * 2), dim3(512), 0, st, reinterpret_cast<const u32x4*>(buf), bytes / 16, sink);
  return hipGetLastError();
}

}  // namespace lgh
