// ptok.h — the persistent token kernel's program (internal): one launch runs every mat-vec and attention op of a decode
// step on resident workgroups; see decode_persistent.hip.
#pragma once

#include "common.h"

namespace lgh {

constexpr int kPtWaves = 8;            // waves per workgroup, one workgroup per CU
constexpr uint32_t kPtNone = 0xFFFFFFFFu;
constexpr uint32_t kPtCntStride = 16;  // words between counters (each on a 64-byte line of its own)
constexpr uint32_t kPtSyncHeader = 64; // sync words before the first counter: [0] epoch  [16] error  [32] finished workgroups
constexpr unsigned kPtSpinLimit = 1u << 19;   // polls of ~0.1-0.3 us each before a wait gives up (the error flag then drains every other wait)

enum : uint32_t { PT_MV = 0, PT_ATTN = 1 };
// how an op's input vector becomes ready
enum : uint32_t {
  PT_IN_READY = 0,   // written before the launch (the embedding's XQ image, a stage's converted input)
  PT_IN_XQ = 1,      // XQ records written earlier IN THIS LAUNCH by another op's epilogue: wait for the records' counters
  PT_IN_ATTN = 2,    // the attention op's split partials: wait for the kv heads' counters, merge them, convert to XQ in LDS
};

// One op of the token program (device memory).  For PT_MV the launch-uniform geometry is what mvq_kernel takes as
// scalars (matvec_mfma.hip) and `mv` indexes the MvLaunch array; for PT_ATTN `attn` indexes the PtAttn array.
struct PtOp {
  uint32_t kind;
  uint32_t n_wg;                 // workgroups with rows in this op (blockIdx.x < n_wg)
  uint32_t wbpack, geom, geom2, red_floats, lds_red_off;
  uint32_t in_kind;
  uint32_t in_cnt;               // first counter of the input's signal groups
  uint32_t out_cnt[3];           // per segment: first counter of its output's signal groups (kPtNone: nobody waits in this launch)
  uint32_t out_shift[3];         // log2(rows per signal group)
  uint32_t next_mv;              // index of the next PT_MV op (kPtNone: none)
  uint32_t mv;
  uint32_t attn;                 // PT_ATTN, and the PT_MV op that consumes its partials
  uint32_t pad[14];
};
static_assert(sizeof(PtOp) == 128, "PtOp is padded to two 64-byte lines");

struct PtAttn {
  const float* q;                // [n_heads][D], RoPE applied (written by the QKV op of this launch)
  const float* kc;               // [n_kv][max_seq][D]
  const float* vc;
  float* part;                   // split partials: per (kv head, split): ml[G][2], acc[G][D]
  const int* pos;
  float scale;
  uint32_t n_heads, n_kv, max_seq;
  uint32_t in_cnt;               // QKV's signal groups of D rows: q heads, then k heads, then v heads
  uint32_t out_cnt;              // one counter per kv head: every split slot arrives once per token (s_max arrivals)
  uint32_t s_max;                // split slots per kv head (workgroups n_kv * s_max take part)
  uint32_t rows_per_split;       // splits in use = clamp(ceil(kv_len / rows_per_split), 1, s_max)
  uint32_t lds_off;              // LDS bytes in front of the attention's scratch (per-wave partial states)
  uint32_t pad[1];
};

struct PtProgram {
  const PtOp* ops;
  const MvLaunch* mv;
  const PtAttn* attn;
  unsigned* sync;
  uint32_t nops;
  uint32_t first_mv;             // index of the first PT_MV op (kPtNone: none)
};

// host side (decode_persistent.hip)
struct PtHostOp { PtOp op; MvLaunch mv; PtAttn attn; uint32_t threads; uint64_t alg_bytes; };
// lays out LDS: x regions | two partial-sum buffers (alternating between ops) | attention scratch; patches the ops; returns the total
size_t ptok_layout_lds(PtHostOp* ops, size_t n, uint32_t head_dim, uint32_t group);
// format mask of the program's matrices (0: not runnable) and whether an instantiation for (mask, head_dim, group) exists
uint32_t ptok_mask(const PtHostOp* ops, size_t n);
bool ptok_supported(uint32_t mask, uint32_t head_dim, uint32_t group);
hipError_t ptok_launch(const PtProgram& P, uint32_t mask, uint32_t head_dim, uint32_t group, size_t lds, hipStream_t st);
size_t pt_part_floats(uint32_t n_kv, uint32_t s_max, uint32_t group, uint32_t head_dim);

}  // namespace lgh
