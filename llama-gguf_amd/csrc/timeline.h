// timeline.h — DIAGNOSTIC BUILD ONLY (-DLGH_STAMPS, `make stamps`): a per-node timeline of a decode step.
//
// Every kernel on the default decode path stamps s_memrealtime (100 MHz, chip-wide) when thread 0 of each workgroup starts
// and when it ends, into a slot chosen by the address of its own AQL dispatch packet (__builtin_amdgcn_dispatch_ptr: 64 B
// per packet in the queue's ring, so consecutive graph nodes get consecutive slots and nothing has to be passed in).
// The host (tools/token_timeline.py via lgh_debug_timeline) reads the buffers of all translation units back after a
// replay, orders the slots by packet address and prints, per node: the gap since the previous node's last workgroup ended,
// the ramp (first -> last workgroup start), the body and the tail skew (first -> last workgroup end).
// Plain stores to per-workgroup words — an atomic min/max on one word per node costs ~30 ns per workgroup and was the
// dominant term of the first version of this probe (tools/probes/node_floor_probe.hip, section C).
// In the product build every macro expands to nothing.
#pragma once

#ifdef LGH_STAMPS
#include <hip/hip_runtime.h>
namespace lgh {
constexpr unsigned kTlSlots = 1024, kTlWg = 320;
enum TlKind : unsigned { TL_EMBED = 1, TL_MVQ = 2, TL_ATTN = 3, TL_COMBINE = 4, TL_ARGMAX1 = 5, TL_ARGMAX2 = 6, TL_ADVANCE = 7, TL_XQ = 8, TL_MV = 9, TL_OTHER = 10 };
struct TlBuf {
  unsigned long long t[kTlSlots][kTlWg][2];
  unsigned long long clk[kTlSlots][kTlWg][2];   // s_memtime (shader clock) at the same two points: the workgroup's clock = dclk / dt
  unsigned xcc[kTlSlots][kTlWg];                // HW_REG_XCC_ID of the workgroup's CU
  unsigned long long packet[kTlSlots];
  unsigned kind[kTlSlots], grid[kTlSlots], aux[kTlSlots], pad[kTlSlots];
};
}  // namespace lgh
#define LGH_TL_DEFINE(NAME)                                                                       \
  namespace lgh {                                                                                 \
  static __device__ TlBuf g_tl_##NAME;                                                            \
  hipError_t tl_read_##NAME(void* host) { return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_tl_##NAME), sizeof(TlBuf)); } \
  }
#define LGH_TL_BEGIN(NAME, KIND, AUX)                                                             \
  const unsigned long long tl_dp_ = (unsigned long long)__builtin_amdgcn_dispatch_ptr();          \
  const unsigned tl_slot_ = (unsigned)(tl_dp_ >> 6) % lgh::kTlSlots;                              \
  const bool tl_on_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0 && blockIdx.x < lgh::kTlWg && blockIdx.y == 0;   /* wave-uniform: all 64 lanes of wave 0 store the same word, no divergent branch */            \
  lgh::TlBuf& tl_buf_ = lgh::g_tl_##NAME;                                                         \
  if (tl_on_) {                                                                                   \
    tl_buf_.t[tl_slot_][blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();                        \
    tl_buf_.clk[tl_slot_][blockIdx.x][0] = __builtin_amdgcn_s_memtime();                          \
    tl_buf_.xcc[tl_slot_][blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11)); /* HW_REG_XCC_ID, all bits */ \
    if (blockIdx.x == 0) { tl_buf_.packet[tl_slot_] = tl_dp_; tl_buf_.kind[tl_slot_] = (KIND); tl_buf_.grid[tl_slot_] = gridDim.x; tl_buf_.aux[tl_slot_] = (AUX); } \
  }
// the same with the slot given by the caller (a kernel that cannot spare the two user SGPRs of the dispatch pointer)
#define LGH_TL_BEGIN_SLOT(NAME, KIND, AUX, SLOT)                                                  \
  const unsigned tl_slot_ = (unsigned)(SLOT) % lgh::kTlSlots;                                     \
  const bool tl_on_ = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0 && blockIdx.x < lgh::kTlWg && blockIdx.y == 0;   /* wave-uniform: all 64 lanes of wave 0 store the same word, no divergent branch */            \
  lgh::TlBuf& tl_buf_ = lgh::g_tl_##NAME;                                                         \
  if (tl_on_) {                                                                                   \
    tl_buf_.t[tl_slot_][blockIdx.x][0] = __builtin_amdgcn_s_memrealtime();                        \
    tl_buf_.clk[tl_slot_][blockIdx.x][0] = __builtin_amdgcn_s_memtime();                          \
    tl_buf_.xcc[tl_slot_][blockIdx.x] = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  \
    if (blockIdx.x == 0) { tl_buf_.packet[tl_slot_] = 0; tl_buf_.kind[tl_slot_] = (KIND); tl_buf_.grid[tl_slot_] = gridDim.x; tl_buf_.aux[tl_slot_] = (AUX); } \
  }
#define LGH_TL_END()                                                                              \
  do {                                                                                            \
    if (tl_on_) {                                                                                 \
      tl_buf_.t[tl_slot_][blockIdx.x][1] = __builtin_amdgcn_s_memrealtime();                      \
      tl_buf_.clk[tl_slot_][blockIdx.x][1] = __builtin_amdgcn_s_memtime();                        \
    }                                                                                             \
  } while (0)
#else
#define LGH_TL_DEFINE(NAME)
#define LGH_TL_BEGIN(NAME, KIND, AUX)
#define LGH_TL_BEGIN_SLOT(NAME, KIND, AUX, SLOT)
#define LGH_TL_END()
#endif
