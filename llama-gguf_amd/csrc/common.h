// common.h — shared declarations of the gfx950 decode engine (internal; the public ABI is
// include/llama_gguf_hip.h).  Everything here is written for CDNA4 only: 64-lane waves, 256 CUs,
// no portability layers.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <vector>

#include "../../include/llama_gguf_hip.h"

namespace lgh {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWave = 64;
constexpr int kNumCU = 256;

// elements / bytes per block of a GGUF type (src/tensor/dtype.rs:50-108); 0 = unknown
__host__ __device__ inline uint32_t blk_elems(int t) {
  switch (t) {
    case LGH_TYPE_F32: case LGH_TYPE_F16: case LGH_TYPE_BF16: return 1;
    case LGH_TYPE_Q4_0: case LGH_TYPE_Q4_1: case LGH_TYPE_Q5_0: case LGH_TYPE_Q5_1: case LGH_TYPE_Q8_0:
    case LGH_TYPE_Q8_1: return 32;
    case LGH_TYPE_Q2_K: case LGH_TYPE_Q3_K: case LGH_TYPE_Q4_K: case LGH_TYPE_Q5_K: case LGH_TYPE_Q6_K:
    case LGH_TYPE_Q8_K: return 256;
    default: return 0;
  }
}

__host__ __device__ inline uint32_t blk_bytes(int t) {
  switch (t) {
    case LGH_TYPE_F32: return 4; case LGH_TYPE_F16: case LGH_TYPE_BF16: return 2;
    case LGH_TYPE_Q4_0: return 18; case LGH_TYPE_Q4_1: return 20; case LGH_TYPE_Q5_0: return 22;
    case LGH_TYPE_Q5_1: return 24; case LGH_TYPE_Q8_0: return 34; case LGH_TYPE_Q8_1: return 36;
    case LGH_TYPE_Q2_K: return 84; case LGH_TYPE_Q3_K: return 110; case LGH_TYPE_Q4_K: return 144;
    case LGH_TYPE_Q5_K: return 176; case LGH_TYPE_Q6_K: return 210; case LGH_TYPE_Q8_K: return 292;
    default: return 0;
  }
}

constexpr int kDevQ4K_T16 = 1012;  // device-only type tag: Q4_K in the 16-row tile layout of matvec_mfma.hip
constexpr int kDevQ6K_T16 = 1014;  // Q6_K in its 16-row tile layout (3392 B per tile), same kernel
constexpr int kDevQ5K_T16 = 1013, kDevQ80_T16 = 1008, kDevQ40_T16 = 1002;   // Q5_K (2816 B), Q8_0 (4352 B), Q4_0 (2304 B) tiles
inline bool mfma_type(int t) { return t == kDevQ4K_T16 || t == kDevQ6K_T16 || t == kDevQ5K_T16 || t == kDevQ80_T16 || t == kDevQ40_T16; }

// ---------------------------------------------------------------------------------------------
// Device weight layouts (what lgh_upload_tensor leaves in HBM).  Byte counts equal the GGUF payload;
// formats whose blocks are not 16-byte multiples are split into per-matrix planes so that every
// lane load is an aligned 16-byte (or 8-byte) access.
//
//   Q4_K  16-row x 256-element tiles of 2304 B for the int8-MFMA kernel (matvec_mfma.hip)   plane0
//         (native 144-B blocks only when k is not a multiple of 256 — never for GGUF K-quants)
//   Q5_K  native 176-B blocks {d,dmin,scales[12],qh[32],qs[128]}                plane0
//   Q6_K  plane0 ql[128]/blk, plane1 qh[64]/blk, plane2 scales[16]/blk, plane3 d(f16)/blk
//   Q8_0  plane0 qs[32]/blk,  plane1 d(f16)/blk
//   Q4_0  plane0 qs[16]/blk,  plane1 d(f16)/blk
//   F32   plane0 row-major [n][k]
// Every other GGUF type is dequantized to F32 at upload (the reference does the same for types
// without a quantized kernel, src/backend/cuda/dequant_weights.rs:211-231).
// ---------------------------------------------------------------------------------------------
struct DevWeight {
  int type = -1;            // device type: LGH_TYPE_{Q4_K,Q5_K,Q6_K,Q8_0,Q4_0,F32}
  int src_type = -1;        // GGUF type it was uploaded as
  uint32_t k = 0, n = 0;    // in_features, out_features (per expert for stacks)
  uint32_t n_stack = 1;     // experts stacked along the outermost dim
  uint8_t* base = nullptr;  // one allocation
  const uint8_t* plane[4] = {nullptr, nullptr, nullptr, nullptr};
  uint64_t stack_stride[4] = {0, 0, 0, 0};  // bytes between consecutive experts, per plane
  size_t bytes = 0;         // payload bytes (== algorithmic bytes of one full read)
  std::vector<bool> filled; // per stacked expert: its payload has been uploaded
  bool present() const { return base != nullptr; }
};

// ---------------------------------------------------------------------------------------------
// Fused mat-vec launch descriptor (passed by value as the kernel argument).
//
// A launch covers up to 3 SEGMENTS (e.g. Q, K, V of one layer); a workgroup belongs to exactly one
// segment and owns `rows_per_wg` consecutive output rows of it.  A segment runs up to 4 PASSES over
// its rows, each with its own weight matrix and input vector (gate+up share x; the two selected MoE
// experts have different x), and the epilogue combines the per-pass results of a row.
// Inside a workgroup, wave w works on k-slice (w % T) of row-group (w / T): a lane keeps the slice of
// x it needs in registers for the whole kernel, so weights are the only streamed operand.
// ---------------------------------------------------------------------------------------------
enum MvEpilogue : int {
  EPI_STORE = 0,       // out[row] = v0 (+bias)
  EPI_RESID = 1,       // out[row] = v0 (+bias) + resid[row]              (layers.rs:1201-1208, 1235-1241)
  EPI_SWIGLU = 2,      // out[row] = silu(v0) * v1                        (simd.rs:598-649)
  EPI_ROPE_Q = 3,      // rotate pairs (2i,2i+1) at *pos, out[row]        (ops.rs:1285-1337)
  EPI_ROPE_K = 4,      // rotate, then write into K cache row *pos        (layers.rs:577-600)
  EPI_V_CACHE = 5,     // write into V cache row *pos
  EPI_MOE_SWIGLU = 6,  // out[row] = silu(v0)*v1 ; out2[row] = silu(v2)*v3 (two selected experts)
  EPI_MOE_DOWN = 7,    // out[row] = (out2[row] or 0) + w0*v0 + w1*v1 (+ resid[row])   (moe.rs:363-368, two experts per launch)
};

struct MvPass {
  const uint8_t* plane[4];
  const float* x;            // f32 input vector (VALU kernel)
  const uint8_t* xq;         // the same vector as XQ records (xq.h; int8-MFMA kernel)
  const int* sel;            // optional device int: expert index; plane[i] += *sel * sel_stride[i]
  uint64_t sel_stride[4];
};

struct MvSeg {
  int type;                  // device weight type
  int epi;
  uint32_t n_rows;
  uint32_t nblk;             // blocks per row (k / block_size)
  uint32_t units;            // lane-units per row
  uint32_t T, G;             // k-slices (waves per row), row-groups
  uint32_t rows_per_wg;
  uint32_t wg_begin;         // first workgroup of this segment
  int npass;
  MvPass pass[4];
  float* out;
  float* out2;
  const float* resid;
  const float* bias;
  const float* moe_w;        // device: routing weights of the selected experts
  uint32_t head_dim;         // RoPE / cache epilogues
  uint32_t max_seq;
  // optional: also leave the output vector as XQ records for an int8-MFMA consumer (store / residual / SwiGLU epilogues)
  uint8_t* xq_out;           // XQ image of `out` (n_rows must be a multiple of 16)
  uint8_t* xq_out2;          // ... and of `out2` (MoE SwiGLU: the second expert's activation)
  const float* xq_nw;        // the consumer's RMSNorm weights: records hold out * xq_nw, and ...
  float* xq_ssq;             // ... xq_ssq[row / 16] = sum of out^2 over the chunk
};

struct MvLaunch {
  int nseg;
  uint32_t k;
  int do_norm;               // RMSNorm prologue on x (all passes share pass[0].x)
  float eps;
  const float* norm_w;
  const int* pos;            // device: current position (RoPE / cache epilogues)
  const float* rope_cs;      // [max_seq][head_dim/2][2] cos,sin
  uint32_t red_floats;       // LDS floats for per-row partial sums (max over segments)
  const float* ssq_part;     // int8-MFMA kernel with do_norm: partial sums of x^2 left by the producer of x
  uint32_t n_ssq_part;
  uint32_t dbg_slot;         // diagnostic builds: launch sequence number mod 64 (span stamps)
  MvSeg seg[3];
};

// Multi-sequence launch (matvec_batch.hip): where sequence s keeps its vectors.  The vectors named in MvLaunch / MvSeg are
// sequence 0's; sequence s adds s times the stride below (XQ images in bytes, everything else in floats).  The K / V cache
// epilogues write into cache slot slot[s] at position pos[s].
constexpr int kMaxBatch = 16;
struct MvBatch {
  uint32_t n_seq;
  const int* pos;             // device [n_seq]
  const int* slot;            // device [n_seq]
  uint64_t cache_stride;      // floats between the caches of two slots
  uint32_t xq_stride;         // bytes between the input XQ images of consecutive sequences
  uint32_t ssq_stride;        // floats between their sum-of-squares partials
  uint32_t out_stride[3];     // per segment
  uint32_t resid_stride[3];
  uint32_t xq_out_stride[3];  // bytes
  uint32_t ssq_out_stride[3];
  float* part;                // mvqb2: partial sums [slot][sequence][unit * 16 + row] between the two launches of an op
  uint64_t part_floats;
  // Indirect entries (MoE layers of a multi-sequence step, one launch per expert): the launch works on *ind_cnt <= n_seq entries;
  // entry j stands for the (sequence, top-k slot) pair v = ind_idx[j]: its OUTPUT vectors are the v-th at the strides above, its
  // INPUT vector / sum-of-squares partials the (v / ind_div)-th.  ind_cnt == NULL: entry j = sequence j (everything else).
  const int* ind_cnt;
  const int* ind_idx;
  uint32_t ind_div;
  // ... and all experts of a layer in ONE launch: blockIdx.z = the expert — its count is ind_cnt[z], its entry list ind_idx + z *
  // ind_stride, its matrices the launch's planes + z * (the stack stride in MvPass::sel_stride), its partial sums a region of their own
  uint32_t ind_stride;
  uint32_t ind_nz;            // experts in the launch (grid z); 0 / 1: a single entry list
  uint64_t part_z_floats;     // (set by the launcher)
};

// launch-uniform geometry of one int8-MFMA launch as the kernel takes it (mvq_pack)
struct MvGeom { uint32_t wbpack, geom, geom2, red_floats, lds_red_off, n_wg; };

// ---------------------------------------------------------------------------------------------
// host-side launchers (defined in the .hip files)
// ---------------------------------------------------------------------------------------------
struct MvPlan {               // geometry chosen by the host for one weight shape
  uint32_t units, T, G, rows_per_wg, n_wg, threads, red_floats;
};

// matvec
hipError_t mv_plan(int dev_type, uint32_t k, uint32_t n_rows, int npass, MvPlan* plan, uint32_t launch_rows,
                   uint32_t wave_cap);
uint32_t mv_wave_cap(int dev_type);  // waves of the one workgroup per CU for this format (0 = not a fused format)
hipError_t mv_launch(const MvLaunch& L, uint32_t n_wg, uint32_t threads, hipStream_t st);
int mv_symbol(const MvLaunch& L);
// int8-MFMA path (matvec_mfma.hip): Q4_K in the tile16 layout
hipError_t mvq_plan(uint32_t k, uint32_t n_rows, int npass, MvPlan* plan, uint32_t launch_rows, uint32_t force_tiles = 0);
hipError_t mvq_launch(const MvLaunch& L, uint32_t n_wg, uint32_t threads, hipStream_t st);
hipError_t repack_q4k_t16_launch(const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st);
hipError_t hbm_read_launch(const uint8_t* buf, size_t bytes, float* sink, int nt, hipStream_t st);   // streaming-read probe
// launch-uniform geometry of one int8-MFMA op, packed as the kernel takes it; returns the format mask (0 = not launchable)
uint32_t mvq_pack(const MvLaunch& L, uint32_t n_wg, uint32_t threads, MvGeom* g, size_t* lds_out);
uint32_t mvq_format_mask(const MvLaunch& L);
uint32_t mvq_tile_bytes(int dev_type);   // bytes of one 16-row x 256-element tile in the device layout of this type (0: not a tile16 type)
hipError_t mvqb_launch(const MvLaunch& L, const MvBatch& B, uint32_t n_wg, uint32_t threads, hipStream_t st);   // matvec_batch.hip
size_t mvqb_lds_bytes(uint32_t n_seq, uint32_t red_floats);
hipError_t xq_quantize_launch(const float* x, const float* nw, uint8_t* xq, float* ssq_part, uint32_t k, hipStream_t st);
hipError_t repack_t16_launch(int dev_type, const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st);
hipError_t repack_q6k_t16_launch(const uint8_t* raw, uint8_t* dst, uint32_t n_rows, uint32_t nblk, hipStream_t st);  // LGH_SYM_MV_* of the instantiation mv_launch will pick
hipError_t f32_matvec_launch(const float* w, const float* x, float* out, uint32_t k, uint32_t n, const float* norm_w,
                             float eps, const float* resid, hipStream_t st);

// upload-time re-layout and dequantization (dst device buffers)
hipError_t repack_launch(int src_type, const uint8_t* raw, uint8_t* dst, const uint64_t plane_off[4], uint64_t n_blocks,
                         hipStream_t st);
hipError_t dequant_launch(int src_type, const uint8_t* raw, float* dst, uint64_t n_elems, hipStream_t st);
// dequantize row `*token` of a [vocab][hidden] table in its NATIVE GGUF layout into dst (embedding lookup)
hipError_t embed_launch(int src_type, const uint8_t* table, const int* token, float* dst, uint32_t hidden, int* state,
                        uint8_t* xq, const float* xq_nw, float* xq_ssq, hipStream_t st);

// misc kernels
hipError_t rms_norm_launch(const float* x, const float* w, float eps, float* out, uint32_t n, hipStream_t st);
hipError_t rope_launch(float* q, float* k, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, const int* pos,
                       const float* rope_cs, int neox, hipStream_t st);
// multi-sequence decode (engine_batch.hip): one launch each for the n_seq sequences of a step
hipError_t embed_multi_launch(int src_type, const uint8_t* table, const int* tokens, float* dst, uint32_t hidden, uint32_t n_seq, uint8_t* xq,
                              const float* xq_nw, float* xq_ssq, uint32_t xq_stride, uint32_t ssq_stride, hipStream_t st);
hipError_t argmax_multi_launch(const float* logits, uint32_t n, uint32_t n_seq, float* part_val, int* part_idx, int* next, hipStream_t st);
hipError_t attn_multi_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                             uint32_t max_seq, float scale, const int* pos, const int* slot, uint64_t slot_stride, uint32_t n_seq,
                             uint32_t n_splits, float* part_ml, float* part_acc, hipStream_t st);
hipError_t attn_combine_multi_launch(const float* part_ml, const float* part_acc, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                                     uint32_t n_splits, uint32_t n_seq, float* out, uint8_t* xq_out, hipStream_t st);
// the rest of the per-op Backend surface (misc.hip): op = 0 add, 1 mul, 2 scale, 3 silu, 4 gelu
hipError_t ewise_launch(int op, const float* a, const float* b, float s, float* out, uint64_t n, hipStream_t st);
hipError_t softmax_rows_launch(const float* x, float* out, uint32_t rows, uint32_t last_dim, hipStream_t st);
hipError_t matmul_f32_launch(const float* a, const float* b, float* c, uint32_t m, uint32_t k, uint32_t n, hipStream_t st);
hipError_t silu_mul_launch(const float* gate, const float* up, float* out, uint32_t n, hipStream_t st);
hipError_t argmax_launch(const float* logits, uint32_t n, float* part_val, int* part_idx, int* state, int* out_token,
                         hipStream_t st);
hipError_t advance_launch(int* state, hipStream_t st);
hipError_t copy_words_launch(void* dst, const void* src, uint32_t n_words, hipStream_t st);
hipError_t moe_group_launch(const int* sel, uint32_t n_seq, uint32_t top_k, uint32_t n_experts, int* cnt, int* idx, uint32_t idx_stride, hipStream_t st);
hipError_t moe_combine_launch(const float* tmp, const float* moe_w, uint32_t top_k, float* hidden, uint32_t H, uint32_t n_seq, const float* nw,
                              uint8_t* xq, uint32_t xq_stride, float* ssq, uint32_t ssq_stride, hipStream_t st);
hipError_t moe_router_launch(const float* x, const float* norm_w, float eps, const float* w, uint32_t hidden,
                             uint32_t n_experts, uint32_t top_k, int* sel, float* sel_w, hipStream_t st, uint32_t n_tokens = 1);

// attention
hipError_t attn_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv,
                       uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, int kv_len_fixed,
                       uint32_t n_splits, float* part_ml, float* part_acc, hipStream_t st);
hipError_t attn_direct_launch(const float* q, const float* kcache, const float* vcache, uint32_t n_heads, uint32_t n_kv, uint32_t head_dim,
                              uint32_t max_seq, float scale, const int* pos, float* out, uint8_t* xq_out, hipStream_t st);
bool attn_shape_has_fast_kernel(uint32_t head_dim, uint32_t group);
// decode attention over the int8 KV cache (kv_quantized.rs); k_new / v_new: the current token's f32 rows [n_kv][D]
// one row through a byte KV format (lgh_model_desc.kv_cache_type 1..3) and back
hipError_t kv_roundtrip_launch(int fmt, const float* x, uint32_t n, uint8_t* bytes, float* scale_out, float* back, hipStream_t st);
hipError_t attn_q8_launch(int fmt, const float* q, int8_t* k8, int8_t* v8, float* kscale, float* vscale, const float* k_new, const float* v_new,
                          uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, uint32_t n_splits,
                          float* part_ml, float* part_acc, hipStream_t st);
// decode attention over the TurboQuant code caches (attention_tq.hip); signs: the layer's [n_kv][2][head_dim]
hipError_t attn_tq_launch(int bits, const float* q, uint8_t* kq, uint8_t* vq, const float* k_new, const float* v_new, const float* signs,
                          uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, uint32_t n_splits,
                          float* part_ml, float* part_acc, hipStream_t st, const float* qjl_s = nullptr, uint32_t* kx = nullptr);
hipError_t attn_tq_combine_launch(int bits, const float* part_ml, const float* part_acc, const float* signs, uint32_t n_heads, uint32_t n_kv,
                                  uint32_t head_dim, uint32_t n_splits, float* out, uint8_t* xq_out, hipStream_t st, uint32_t n_seq = 0,
                                  uint32_t xq_stride = 0);
hipError_t attn_tq_multi_launch(int bits, const float* q, uint8_t* kq, uint8_t* vq, const float* k_new, const float* v_new, const float* signs,
                                uint32_t n_heads, uint32_t n_kv, uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, const int* slot,
                                uint64_t code_stride, uint64_t x_stride, uint32_t kv_stride, uint32_t n_seq, uint32_t n_splits, float* part_ml,
                                float* part_acc, hipStream_t st, const float* qjl_s = nullptr, uint32_t* kx = nullptr);
hipError_t tq_compress_launch(int bits, const float* x, uint32_t dim, const float* signs, uint8_t* out, hipStream_t st, const float* qjl_s = nullptr,
                              uint32_t* qjl_out = nullptr);
uint32_t tq_row_bytes_host(int bits, uint32_t d);
hipError_t attn_decode_any_launch(const float* q, const float* kcache, const float* vcache, float* out, uint32_t n_heads, uint32_t n_kv,
                                  uint32_t head_dim, uint32_t max_seq, float scale, const int* pos, hipStream_t st);
hipError_t attn_generic_launch(const float* q, const float* k, const float* v, float* out, uint32_t n_heads, uint32_t n_kv, uint32_t seq_len,
                               uint32_t kv_len, uint32_t kv_rows, uint32_t d, float scale, hipStream_t st);
hipError_t attn_combine_launch(const float* part_ml, const float* part_acc, uint32_t n_heads, uint32_t n_kv,
                               uint32_t head_dim, uint32_t n_splits, float* out, uint8_t* xq_out, hipStream_t st);

// The opt-in to more than 64 KB of dynamic LDS is per kernel AND per device: `done` is that kernel's per-device record
// (a process may drive several GPUs: two stage contexts, a test creating contexts on different devices).
inline hipError_t lds_opt_in(const void* fn, int bytes, bool (&done)[64]) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
  if (done[dev]) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[dev] = true;
  return e;
}

// device state block: [0] token, [1] current position, [2] next position, [3] last arg-max
enum { ST_TOKEN = 0, ST_POS = 1, ST_NEXT = 2, ST_ARGMAX = 3, ST_WORDS = 8 };

}  // namespace lgh
