// wave_share_probe.hip — would UNEVEN shares between the waves of a mat-vec workgroup shorten a launch?
// In the engine's kernel the 8 waves of a workgroup are the 8 k-slices of its row tiles and all stream the same number of tiles;
// the ledger (profiles/r03a_decode_ledger.md) has the second four waves starting later and streaming slower, and the barrier at
// the end of gate/up waiting ~2.3 us for them.  The probe streams the gate/up-sized matrix (224 workgroups x 8 waves, 4 tiles of
// 2304 B in flight per wave, nt loads) with shares (a, b): waves 0-3 take `a` items each, waves 4-7 `b` items each (a + b = 32 =
// 2 passes x 4 tiles x 2 blocks x 2), and reports launch time and the mean end time of each wave (in-kernel, us).
// Build: hipcc --offload-arch=gfx950 -O3 -o wave_share_probe wave_share_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                                        \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

typedef unsigned long long ull;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// a workgroup owns 32 x 8 = 256 items of tb bytes, contiguous; wave w takes items [off_w, off_w + n_w)
__global__ void __launch_bounds__(512) k_stream(const unsigned char* __restrict__ w, unsigned tb, unsigned na, unsigned nb, float* sink, ull* stamps) {
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = blockIdx.x;
  const ull t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned nitems = wave < 4 ? na : nb;
  const unsigned off = wave < 4 ? wave * na : 4 * na + (wave - 4) * nb;
  const unsigned char* base = w + ((ull)b * 256 + off) * tb + lane * 16;
  u32x4 acc = {0, 0, 0, 0};
  constexpr int D = 4;
  u32x4 q0[D], q1[D], hd[D];
  auto issue = [&](int j, unsigned it) {
    const unsigned char* a = base + (ull)(it < nitems ? it : nitems - 1) * tb;
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
  if (lane == 0) { stamps[(b * 8 + wave) * 2] = t0; stamps[(b * 8 + wave) * 2 + 1] = __builtin_amdgcn_s_memrealtime(); }
  __syncthreads();
}

int main() {
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const unsigned n_wg = 224, tb = 2304;
  const size_t mat = (size_t)n_wg * 256 * tb;
  const size_t stride = (mat + (2u << 20) + 4095) / 4096 * 4096 + 4096 * 37;
  const int copies = 24;
  unsigned char* w;
  CHECK(hipMalloc(&w, stride * copies));
  CHECK(hipMemset(w, 0x5A, stride * copies));
  float* sink;
  CHECK(hipMalloc(&sink, 4096));
  ull* st;
  CHECK(hipMalloc(&st, sizeof(ull) * n_wg * 16));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const unsigned shares[][2] = {{32, 32}, {34, 30}, {36, 28}, {38, 26}, {40, 24}, {30, 34}};
  std::printf("| items per wave (waves 0-3 / 4-7) | us per launch | mean end of waves 0..7 after the workgroup's first start (us) | last wave end, mean over workgroups |\n|---|---|---|---|\n");
  for (auto& sh : shares) {
    auto launch = [&](int i) { hipLaunchKernelGGL(k_stream, dim3(n_wg), dim3(512), 0, s, w + (size_t)(i % copies) * stride, tb, sh[0], sh[1], sink, st); };
    for (int i = 0; i < 5; i++) launch(i);
    CHECK(hipStreamSynchronize(s));
    const int iters = 48;
    CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; i++) launch(i + 5);
    CHECK(hipEventRecord(e1, s));
    CHECK(hipStreamSynchronize(s));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    double wend[8] = {0}, last = 0;
    const int reps = 6;
    for (int r = 0; r < reps; r++) {
      launch(r * 3 + 1);
      CHECK(hipStreamSynchronize(s));
      std::vector<ull> h(n_wg * 16);
      CHECK(hipMemcpy(h.data(), st, sizeof(ull) * n_wg * 16, hipMemcpyDeviceToHost));
      for (unsigned b = 0; b < n_wg; b++) {
        ull first = ~0ull, le = 0;
        for (int v = 0; v < 8; v++) first = std::min(first, h[(b * 8 + v) * 2]);
        for (int v = 0; v < 8; v++) { wend[v] += (double)(h[(b * 8 + v) * 2 + 1] - first) * 0.01; le = std::max(le, h[(b * 8 + v) * 2 + 1]); }
        last += (double)(le - first) * 0.01;
      }
    }
    std::printf("| %u / %u | %.2f |", sh[0], sh[1], ms * 1000.0 / iters);
    for (int v = 0; v < 8; v++) std::printf(" %.2f", wend[v] / (reps * n_wg));
    std::printf(" | %.2f |\n", last / (reps * n_wg));
  }
  return 0;
}
