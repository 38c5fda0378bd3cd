// handoff.h — device-side hand-off counters between producers and consumers that are NOT ordered by a kernel boundary
// (internal).  Used by the persistent token kernel (decode_persistent.hip: ops of one launch) and by the flag-ordered
// mat-vec launches (matvec_mfma.hip: two consecutive launches of a token running side by side on two streams).
//
// Protocol (MI355X_MICROARCH.md, "Workgroup dispatch ... inter-workgroup visibility", row 1 of the hand-off table): the
// producer writes the handed-off bytes with write-through (sc1) stores, every storing wave waits for them
// (s_waitcnt vmcnt(0)), the workgroup's barrier, then ONE lane adds the rows it produced to the counter of each signal
// group (agent-scope atomic).  A consumer wave polls the counters of the groups it needs with sc1 loads and then reads the
// bytes with sc1 loads only.  Counters only ever grow: the target of the n-th token is n x rows-per-group, n kept in a
// device word, so a replayed hipGraph needs no reset and no host-side change.  Every spin is bounded: a timeout raises
// the error word (sync[16]) and the waiter goes on (the host reports it, check_chain in engine.hip).
#pragma once

#include "common.h"

namespace lgh {

constexpr uint32_t kHoCntStride = 16;    // words between counters (each on a 64-byte line of its own)
constexpr uint32_t kHoHeader = 64;       // sync words before the first counter: [0] epoch  [16] error  [32] finished workgroups
constexpr unsigned kHoSpinLimit = 1u << 19;

#ifdef __HIPCC__
__device__ __forceinline__ unsigned ho_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Lanes l < n each wait until counter (first + l) has reached `target` (wrap-safe); the whole wave leaves together.
__device__ __forceinline__ void ho_wait(unsigned* sync, uint32_t first, uint32_t n, unsigned target, uint32_t lane) {
  const unsigned* c = sync + kHoHeader + (size_t)(first + (lane < n ? lane : 0)) * kHoCntStride;
  unsigned spins = 0;
  for (;;) {
    const bool ok = lane >= n || (int)(ho_ld(c) - target) >= 0;
    if (__all(ok)) break;
    __builtin_amdgcn_s_sleep(2);
    ++spins;
    if ((spins & 255u) == 0 && (spins > kHoSpinLimit || ho_ld(sync + 16) != 0)) {   // timed out (here, or somewhere else already)
      if (lane == 0 && ho_ld(sync + 16) == 0) {   // the first to give up leaves what it was waiting for: [17] counter [18] target [19] seen
        __hip_atomic_store(sync + 17, first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 18, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 19, ho_ld(c), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      break;
    }
  }
}

__device__ __forceinline__ void ho_signal(unsigned* sync, uint32_t counter, unsigned add) {
  __hip_atomic_fetch_add(sync + kHoHeader + (size_t)counter * kHoCntStride, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Lanes of one wave signal the groups of `1 << shift` rows that rows [r0, r1) overlap, each with its share of the rows.
__device__ __forceinline__ void ho_signal_rows(unsigned* sync, uint32_t first, uint32_t shift, uint32_t r0, uint32_t r1, uint32_t lane) {
  if (r0 >= r1) return;
  const uint32_t g0 = r0 >> shift, g1 = (r1 - 1) >> shift;
  if (lane <= g1 - g0) {
    const uint32_t lo = max(r0, (g0 + lane) << shift), hi = min(r1, (g0 + lane + 1) << shift);
    ho_signal(sync, first + g0 + lane, hi - lo);
  }
}
#endif

}  // namespace lgh
