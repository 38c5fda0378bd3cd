// grid_barrier_probe.hip — what does a grid-wide barrier cost on MI355X when every CU holds one 512-thread workgroup?
// (Input for a persistent per-token kernel: it only pays if a barrier + hand-off is cheaper than a kernel boundary, whose
// floor was measured at ~4.7 us per graph node.)  Three variants, 256 workgroups, R rounds each:
//   flat    one monotonic counter, relaxed polling with sc1 loads
//   xcd     per-XCD counters (blockIdx % 8), leader of each XCD arrives at a top counter, generation word per XCD
//   payload xcd + every workgroup publishes 64 B with write-through stores before arriving and reads another
//           workgroup's 64 B after leaving (the cost that actually matters: data hand-off across XCDs)
// All spins are bounded (a stuck barrier sets an error flag and the kernel runs to completion).
// Build: hipcc --offload-arch=gfx950 -O3 -o grid_barrier_probe grid_barrier_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kWG = 256, kThreads = 512, kXcd = 8;
constexpr unsigned kSpinLimit = 4u << 20;

struct Sync {
  unsigned flat;            // monotonic
  unsigned pad0[31];
  unsigned top;             // monotonic, leaders only
  unsigned pad1[31];
  unsigned xcd_count[kXcd * 32];   // one cache line per XCD
  unsigned xcd_gen[kXcd * 32];
  unsigned error;
};

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ void barrier_flat(Sync* s, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(&s->flat, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned want = (round + 1) * kWG;
    unsigned spins = 0;
    while (ld_sc1(&s->flat) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { s->error = 1; break; }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ void barrier_xcd(Sync* s, unsigned round) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned x = blockIdx.x % kXcd, per = kWG / kXcd;
    const unsigned old = __hip_atomic_fetch_add(&s->xcd_count[x * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned spins = 0;
    if (old == (round + 1) * per - 1) {   // last of this XCD: go to the top counter, then release the XCD
      __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (ld_sc1(&s->top) < (round + 1) * kXcd) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) { s->error = 2; break; }
      }
      __hip_atomic_store(&s->xcd_gen[x * 32], round + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      while (ld_sc1(&s->xcd_gen[x * 32]) < round + 1) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinLimit) { s->error = 3; break; }
      }
    }
  }
  __syncthreads();
}

template <int MODE>
__global__ void __launch_bounds__(kThreads) probe(Sync* s, unsigned rounds, float* payload, float* sink) {
  float acc = 0.0f;
  for (unsigned r = 0; r < rounds; r++) {
    if (MODE == 2) {   // publish 64 B (16 lanes x 4 B) with write-through stores, drained before arriving
      if (threadIdx.x < 16) __hip_atomic_store(&payload[(blockIdx.x * 16 + threadIdx.x)], (float)(r + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (MODE == 0) barrier_flat(s, r);
    else barrier_xcd(s, r);
    if (MODE == 2) {   // read the 64 B of a workgroup on another XCD, past the non-coherent caches
      const unsigned other = (blockIdx.x + 3) % kWG;
      if (threadIdx.x < 16) acc += __hip_atomic_load(&payload[other * 16 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (threadIdx.x < 16) sink[blockIdx.x * 16 + threadIdx.x] = acc;
}

template <int MODE>
static void run(const char* name, Sync* s, float* payload, float* sink, unsigned rounds) {
  CHECK(hipMemset(s, 0, sizeof(Sync)));
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  hipLaunchKernelGGL(probe<MODE>, dim3(kWG), dim3(kThreads), 0, 0, s, 8u, payload, sink);   // warm-up
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(s, 0, sizeof(Sync)));
  CHECK(hipEventRecord(a, 0));
  hipLaunchKernelGGL(probe<MODE>, dim3(kWG), dim3(kThreads), 0, 0, s, rounds, payload, sink);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  Sync h;
  CHECK(hipMemcpy(&h, s, sizeof(Sync), hipMemcpyDeviceToHost));
  std::printf("%-8s %u rounds: %.2f us per barrier  (error flag %u)\n", name, rounds, ms * 1000.0 / rounds, h.error);
}

int main() {
  Sync* s;
  float *payload, *sink;
  CHECK(hipMalloc(&s, sizeof(Sync)));
  CHECK(hipMalloc(&payload, kWG * 16 * 4));
  CHECK(hipMalloc(&sink, kWG * 16 * 4));
  CHECK(hipMemset(payload, 0, kWG * 16 * 4));
  run<0>("flat", s, payload, sink, 2000);
  run<1>("xcd", s, payload, sink, 2000);
  run<2>("payload", s, payload, sink, 2000);
  return 0;
}
