// cu_balance_probe.hip — does spreading a launch's row tiles UNEVENLY over all 256 CUs beat an even share on fewer CUs?
// The engine gives every workgroup of a launch the same number of 16-row tiles, one workgroup per CU: Llama-3-8B gate/up has 896
// tile pairs = 224 workgroups x 4 (32 CUs idle); with 256 workgroups half of them would get 4 and half 3.  If a launch is bound by
// what ONE CU can pull (the longest workgroup is unchanged), that changes nothing; if it is bound by the memory system, the
// extra CUs help.  Same stream as xcd_skew_probe.hip (8 waves = 8 k-slices, 4 tiles of 2304 B in flight per wave, nt loads), cold
// copies, us per launch.
// Build: hipcc --offload-arch=gfx950 -O3 -o cu_balance_probe cu_balance_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                        \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

typedef unsigned long long ull;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// workgroup b owns R tiles if b < n_full, else R - 1; tiles are contiguous in workgroup order
__global__ void __launch_bounds__(512) k_stream(const unsigned char* __restrict__ w, unsigned nblk, unsigned tb, unsigned R, unsigned n_full,
                                                unsigned npass, ull plane, float* sink) {
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = blockIdx.x;
  const unsigned Rb = b < n_full ? R : R - 1;
  const unsigned tile0 = b < n_full ? b * R : n_full * R + (b - n_full) * (R - 1);
  const unsigned nbw = nblk / 8, blk0 = wave * nbw;
  const unsigned nitems = npass * Rb * nbw;
  if (nitems == 0) return;
  u32x4 acc = {0, 0, 0, 0};
  auto addr = [&](unsigned it) -> const unsigned char* {
    const unsigned p = it / (Rb * nbw), r = it % (Rb * nbw), j = r / nbw, bb = r % nbw;
    return w + (ull)p * plane + ((ull)(tile0 + j) * nblk + blk0 + bb) * tb + lane * 16;
  };
  constexpr int D = 4;
  u32x4 q0[D], q1[D], hd[D];
  auto issue = [&](int j, unsigned it) {
    const unsigned char* a = addr(it < nitems ? it : nitems - 1);
    hd[j] = __builtin_nontemporal_load((const u32x4*)(a + 2048 - lane * 16 + (lane & 15) * 16));
    q0[j] = __builtin_nontemporal_load((const u32x4*)a);
    q1[j] = __builtin_nontemporal_load((const u32x4*)(a + 1024));
  };
#pragma unroll
  for (int j = 0; j < D; j++) issue(j, j);
  for (unsigned it = 0; it < nitems; it += D) {
#pragma unroll
    for (int j = 0; j < D; j++) {
      acc ^= q0[j] ^ q1[j] ^ hd[j];
      issue(j, it + D + j);
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1.0f;
}

int main() {
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const size_t pool = (size_t)3 << 30;
  unsigned char* w;
  CHECK(hipMalloc(&w, pool));
  CHECK(hipMemset(w, 0x5A, pool));
  float* sink;
  CHECK(hipMalloc(&sink, 4096));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct Cfg { const char* name; unsigned tiles, nblk, tb, npass, n_wg, R, n_full; };
  const Cfg cfgs[] = {
      {"gate/up 896 tile pairs: 224 wg x 4", 896, 16, 2304, 2, 224, 4, 224},
      {"gate/up 896 tile pairs: 256 wg = 128 x 4 + 128 x 3", 896, 16, 2304, 2, 256, 4, 128},
      {"gate/up 896 tile pairs: 448 wg x 2 (two per CU)", 896, 16, 2304, 2, 448, 2, 448},
      {"QKV-sized 384 tiles: 192 wg x 2", 384, 16, 2304, 1, 192, 2, 192},
      {"QKV-sized 384 tiles: 256 wg = 128 x 2 + 128 x 1", 384, 16, 2304, 1, 256, 2, 128},
      {"QKV-sized 384 tiles: 384 wg x 1", 384, 16, 2304, 1, 384, 1, 384},
      {"down 256 tiles x 56 blocks: 256 wg x 1", 256, 56, 2304, 1, 256, 1, 256},
  };
  std::printf("| configuration | us per launch (eager, cold copies) | TB/s |\n|---|---|---|\n");
  for (const Cfg& c : cfgs) {
    const size_t plane = (size_t)c.tiles * c.nblk * c.tb, mat = plane * c.npass;
    const size_t stride = (mat + (2u << 20) + 4095) / 4096 * 4096 + 4096 * 37;
    const int copies = (int)std::min<size_t>(24, pool / stride);
    auto launch = [&](int i) {
      hipLaunchKernelGGL(k_stream, dim3(c.n_wg), dim3(512), 0, s, w + (size_t)(i % copies) * stride, c.nblk, c.tb, c.R, c.n_full, c.npass, (ull)plane, sink);
    };
    double best = 1e9, sum = 0;
    for (int rep = 0; rep < 3; rep++) {
      for (int i = 0; i < 5; i++) launch(i);
      CHECK(hipStreamSynchronize(s));
      const int iters = 48;
      CHECK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; i++) launch(i + 5);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double us = ms * 1000.0 / iters;
      best = std::min(best, us); sum += us;
    }
    std::printf("| %s | %.2f (best %.2f) | %.2f |\n", c.name, sum / 3, best, mat / (sum / 3) / 1e6);
  }
  return 0;
}
