// node_floor_probe.hip — what does ONE node of a captured hipGraph chain cost on MI355X, and which launch attribute
// changes it?  (the decode engine pays ~4.3 us of fixed cost per graph node where the guide's
// boundary row is 1.45 us between trivial kernels; nobody had measured an empty-node baseline on the engine's own launch
// configuration: kernarg preload, >= 64 KB dynamic LDS opt-in, non-blocking stream, stream capture.)
//
// Sections (all times are DEVICE time per node, from the slope between a short and a long chain, so the graph-launch
// overhead cancels):
//   A  chain of empty kernels: grid x block sweep; capture vs explicit graph API; blocking vs non-blocking stream
//   B  one attribute at a time on a 256 x 512 launch: dynamic LDS 0 / 64 / 128 / 160 KB, kernarg 16 B / 1 KB / 4 KB by value,
//      256 VGPRs, scratch use, 1 / 2 dependent memory round trips, alternating between 16 distinct kernels
//   C  timeline with in-kernel s_memrealtime stamps: [stream 32 MB nt] -> [small] -> [small] ... : the gap between the last
//      wave of a node and the first wave of the next, the ramp (first wave -> last wave started), the span
//   D  instruction-fetch cost: a kernel with a long straight-line body, same kernel back to back vs 16 copies in rotation
//      vs copies in rotation with a cache-sweeping stream kernel between them
// Build twice (the engine uses kernarg preload):
//   hipcc --offload-arch=gfx950 -O3 -o node_floor_probe node_floor_probe.hip
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=8 -o node_floor_probe_pl node_floor_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#define CHECK(x)                                                                                        \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

typedef unsigned long long ull;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- kernels
template <int ID>
__global__ void k_nop() {}

template <int ID>
__global__ void k_nop_arg(unsigned a, unsigned b, float* sink) {
  if (a == 0xFFFFFFFFu && b == 1u + ID) sink[threadIdx.x] = 1.0f;   // never
}

struct Big1K { unsigned v[256]; };
struct Big4K { unsigned v[1000]; };   // the kernarg segment is capped at 4 KB
__global__ void k_big1k(const Big1K b, float* sink) {
  if (b.v[255] == 0xFFFFFFFFu) sink[threadIdx.x] = (float)b.v[threadIdx.x & 255];
}
__global__ void k_big4k(const Big4K b, float* sink) {
  if (b.v[999] == 0xFFFFFFFFu) sink[threadIdx.x] = (float)b.v[threadIdx.x & 511];
}

__global__ void k_lds(unsigned a, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  if (a == 0xFFFFFFFFu) { smem[threadIdx.x] = 1; __syncthreads(); sink[threadIdx.x] = smem[threadIdx.x ^ 1]; }
}

__global__ void __launch_bounds__(512) k_vgpr256(unsigned a, float* sink) {
  asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if (a == 0xFFFFFFFFu) sink[threadIdx.x] = 1.0f;
}

__global__ void k_scratch(unsigned a, float* sink) {
  volatile float arr[64];
  for (int i = 0; i < 64; i++) arr[i] = (float)(i + a);
  if (a == 0xFFFFFFFFu) sink[threadIdx.x] = arr[(threadIdx.x + a) & 63];
}

// one dependent round trip: out = in + 1 (every thread one dword)
__global__ void k_rt1(const float* __restrict__ in, float* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[i] + 1.0f;
}
// two dependent round trips
__global__ void k_rt2(const int* __restrict__ idx, const float* __restrict__ in, float* __restrict__ out) {
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  out[i] = in[idx[i]] + 1.0f;
}
// barrier + LDS reduction + one round trip: the shape of attn_combine / argmax_stage2
__global__ void k_rt1_bar(const float* __restrict__ in, float* __restrict__ out) {
  __shared__ float red[8];
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  float v = in[i];
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.0f;
  for (unsigned w = 0; w < blockDim.x / 64; w++) t += red[w];
  out[i] = t;
}

// streaming read (nt), the weight stream of a mat-vec launch: grid 256 x 512, bytes per launch given
__global__ void __launch_bounds__(512) k_stream(const u32x4* __restrict__ w, unsigned long long n16_per_wg, float* sink, ull* t, unsigned slot) {
  const bool st = t != nullptr && threadIdx.x == 0;
  if (st) { const ull now = __builtin_amdgcn_s_memrealtime(); atomicMin(&t[4 * slot], now); atomicMax(&t[4 * slot + 1], now); }
  const u32x4* p = w + (size_t)blockIdx.x * n16_per_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (ull i = threadIdx.x; i < n16_per_wg; i += 512 * 4) {
    u32x4 a = __builtin_nontemporal_load(p + i);
    u32x4 b = i + 512 < n16_per_wg ? __builtin_nontemporal_load(p + i + 512) : a;
    u32x4 c = i + 1024 < n16_per_wg ? __builtin_nontemporal_load(p + i + 1024) : a;
    u32x4 d = i + 1536 < n16_per_wg ? __builtin_nontemporal_load(p + i + 1536) : a;
    acc ^= a ^ b ^ c ^ d;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1.0f;
  if (st) { const ull now = __builtin_amdgcn_s_memrealtime(); atomicMax(&t[4 * slot + 2], now); }
}

// small stamped kernel: [0] first wave start (min), [1] last wave start (max), [2] last end (max)
__global__ void k_small_stamp(const float* __restrict__ in, float* __restrict__ out, ull* t, unsigned slot, int round_trips) {
  const bool st = (threadIdx.x & 63) == 0;
  if (st) { const ull now = __builtin_amdgcn_s_memrealtime(); atomicMin(&t[4 * slot], now); atomicMax(&t[4 * slot + 1], now); }
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if (round_trips > 0) {
    float v = in[i];
    if (round_trips > 1) v += in[((unsigned)v) & 1023u];
    out[i] = v + 1.0f;
  }
  if (st) { const ull now = __builtin_amdgcn_s_memrealtime(); atomicMax(&t[4 * slot + 2], now); }
}

// long straight-line body (instruction fetch): ~NI VALU instructions executed once
template <int ID, int NI>
__global__ void k_code(float seed, float* sink) {
  float a = seed + (float)ID, b = seed * 0.5f;
#pragma unroll
  for (int i = 0; i < NI / 2; i++) {
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a) : "v"(b));
    asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(b) : "v"(a));
  }
  if (a == 12345.678f) sink[threadIdx.x] = b;
}

// ---------------------------------------------------------------- measurement
struct Ctx {
  hipStream_t st;
  hipEvent_t e0, e1;
};

// per-node device time of a chain: capture n launches, replay, slope between two lengths
using LaunchFn = std::function<void(hipStream_t, int)>;

static double replay_us(Ctx& c, const LaunchFn& f, int n, bool explicit_api = false) {
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  (void)explicit_api;
  CHECK(hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; i++) f(c.st, i);
  CHECK(hipStreamEndCapture(c.st, &g));
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; i++) CHECK(hipGraphLaunch(ge, c.st));
  CHECK(hipStreamSynchronize(c.st));
  std::vector<double> ts;
  for (int r = 0; r < 9; r++) {
    CHECK(hipEventRecord(c.e0, c.st));
    CHECK(hipGraphLaunch(ge, c.st));
    CHECK(hipEventRecord(c.e1, c.st));
    CHECK(hipStreamSynchronize(c.st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
    ts.push_back(ms * 1000.0);
  }
  std::sort(ts.begin(), ts.end());
  CHECK(hipGraphExecDestroy(ge));
  CHECK(hipGraphDestroy(g));
  return ts[ts.size() / 2];
}

static double node_us(Ctx& c, const LaunchFn& f, int n1 = 64, int n2 = 448) {
  const double a = replay_us(c, f, n1), b = replay_us(c, f, n2);
  return (b - a) / (double)(n2 - n1);
}

static double eager_us(Ctx& c, const LaunchFn& f, int n = 2000) {
  for (int i = 0; i < 50; i++) f(c.st, i);
  CHECK(hipStreamSynchronize(c.st));
  CHECK(hipEventRecord(c.e0, c.st));
  for (int i = 0; i < n; i++) f(c.st, i);
  CHECK(hipEventRecord(c.e1, c.st));
  CHECK(hipStreamSynchronize(c.st));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
  return ms * 1000.0 / n;
}

// explicit graph API: a linear chain of kernel nodes of k_nop_arg<0>
static double explicit_chain_us(Ctx& c, int n, dim3 grid, dim3 block, float* sink) {
  hipGraph_t g;
  CHECK(hipGraphCreate(&g, 0));
  unsigned a = 1, b = 2;
  void* args[3] = {&a, &b, &sink};
  hipGraphNode_t prev = nullptr;
  for (int i = 0; i < n; i++) {
    hipKernelNodeParams p;
    std::memset(&p, 0, sizeof(p));
    p.func = (void*)k_nop_arg<0>;
    p.gridDim = grid;
    p.blockDim = block;
    p.sharedMemBytes = 0;
    p.kernelParams = args;
    p.extra = nullptr;
    hipGraphNode_t nd;
    CHECK(hipGraphAddKernelNode(&nd, g, prev ? &prev : nullptr, prev ? 1 : 0, &p));
    prev = nd;
  }
  hipGraphExec_t ge;
  CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int i = 0; i < 3; i++) CHECK(hipGraphLaunch(ge, c.st));
  CHECK(hipStreamSynchronize(c.st));
  std::vector<double> ts;
  for (int r = 0; r < 9; r++) {
    CHECK(hipEventRecord(c.e0, c.st));
    CHECK(hipGraphLaunch(ge, c.st));
    CHECK(hipEventRecord(c.e1, c.st));
    CHECK(hipStreamSynchronize(c.st));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, c.e0, c.e1));
    ts.push_back(ms * 1000.0);
  }
  std::sort(ts.begin(), ts.end());
  CHECK(hipGraphExecDestroy(ge));
  CHECK(hipGraphDestroy(g));
  return ts[ts.size() / 2];
}

template <int I>
static void launch_nop_rot(hipStream_t s, int i, dim3 g, dim3 b, float* sink) {
  if constexpr (I < 16) {
    if ((i & 15) == I) { hipLaunchKernelGGL(k_nop_arg<I>, g, b, 0, s, 1u, 2u, sink); return; }
    launch_nop_rot<I + 1>(s, i, g, b, sink);
  }
}
template <int NI, int I>
static void launch_code_rot(hipStream_t s, int i, dim3 g, dim3 b, float* sink) {
  if constexpr (I < 16) {
    if ((i & 15) == I) { hipLaunchKernelGGL((k_code<I, NI>), g, b, 0, s, 1.0f, sink); return; }
    launch_code_rot<NI, I + 1>(s, i, g, b, sink);
  }
}

int main(int argc, char** argv) {
  const bool quick = argc > 1 && std::string(argv[1]) == "quick";
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  std::printf("# device %s  CUs %d  clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  Ctx c;
  CHECK(hipStreamCreateWithFlags(&c.st, hipStreamNonBlocking));
  CHECK(hipEventCreate(&c.e0));
  CHECK(hipEventCreate(&c.e1));
  Ctx cb;
  CHECK(hipStreamCreate(&cb.st));
  cb.e0 = c.e0; cb.e1 = c.e1;

  float *sink, *bufA, *bufB;
  int* idx;
  CHECK(hipMalloc(&sink, 1 << 20));
  CHECK(hipMalloc(&bufA, 1 << 22));
  CHECK(hipMalloc(&bufB, 1 << 22));
  CHECK(hipMalloc(&idx, 1 << 22));
  CHECK(hipMemset(bufA, 0, 1 << 22));
  CHECK(hipMemset(bufB, 0, 1 << 22));
  CHECK(hipMemset(idx, 0, 1 << 22));
  CHECK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  // ---------------- A
  std::printf("\n## A. chain of empty kernels, us per node (graph, slope 64 -> 448 nodes)\n");
  std::printf("| grid x block | capture, non-blocking stream | capture, blocking stream | no-arg kernel | eager (host-bound floor) |\n|---|---|---|---|---|\n");
  const int grids[] = {1, 32, 256, 1024, 2048};
  const int blocks[] = {64, 128, 256, 512, 1024};
  for (int g : grids)
    for (int b : blocks) {
      if (quick && !(g == 256 || g == 32)) continue;
      if ((g == 1 && b != 64) || (g == 2048 && b != 256) || (g == 1024 && b > 256)) continue;
      LaunchFn f = [=](hipStream_t s, int) { hipLaunchKernelGGL(k_nop_arg<0>, dim3(g), dim3(b), 0, s, 1u, 2u, sink); };
      LaunchFn f0 = [=](hipStream_t s, int) { hipLaunchKernelGGL(k_nop<0>, dim3(g), dim3(b), 0, s); };
      std::printf("| %d x %d | %.2f | %.2f | %.2f | %.2f |\n", g, b, node_us(c, f), node_us(cb, f), node_us(c, f0), eager_us(c, f));
      std::fflush(stdout);
    }
  {
    const double a = explicit_chain_us(c, 64, dim3(256), dim3(512), sink), b = explicit_chain_us(c, 448, dim3(256), dim3(512), sink);
    std::printf("explicit hipGraphAddKernelNode chain, 256 x 512: %.2f us per node (64 nodes %.1f us, 448 nodes %.1f us)\n", (b - a) / 384.0, a, b);
  }

  // ---------------- B
  std::printf("\n## B. one attribute at a time, 256 x 512 (and 32 x 128), us per node\n| variant | 256 x 512 | 32 x 128 |\n|---|---|---|\n");
  auto row = [&](const char* name, std::function<LaunchFn(dim3, dim3)> mk) {
    std::printf("| %s | %.2f | %.2f |\n", name, node_us(c, mk(dim3(256), dim3(512))), node_us(c, mk(dim3(32), dim3(128))));
    std::fflush(stdout);
  };
  row("empty, 3 scalar args", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_nop_arg<0>, g, b, 0, s, 1u, 2u, sink); }); });
  for (int kb : {0, 16, 64, 128, 160})
    row((std::string("dynamic LDS ") + std::to_string(kb) + " KB").c_str(), [&, kb](dim3 g, dim3 b) {
      return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_lds, g, b, (size_t)kb * 1024, s, 1u, sink); });
    });
  {
    static Big1K b1; static Big4K b4;
    std::memset(&b1, 0, sizeof(b1)); std::memset(&b4, 0, sizeof(b4));
    row("kernarg 1 KB by value", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_big1k, g, b, 0, s, b1, sink); }); });
    row("kernarg 4 KB by value", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_big4k, g, b, 0, s, b4, sink); }); });
  }
  row("256 VGPRs", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_vgpr256, g, b, 0, s, 1u, sink); }); });
  row("scratch (256 B per lane)", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(k_scratch, g, b, 0, s, 1u, sink); }); });
  row("16 distinct empty kernels in rotation", [&](dim3 g, dim3 b) { return LaunchFn([=](hipStream_t s, int i) { launch_nop_rot<0>(s, i, g, b, sink); }); });
  row("1 dependent round trip (out = in + 1)", [&](dim3 g, dim3 b) {
    return LaunchFn([=](hipStream_t s, int i) { hipLaunchKernelGGL(k_rt1, g, b, 0, s, (i & 1) ? bufB : bufA, (i & 1) ? bufA : bufB); });
  });
  row("1 round trip + wave reduction + barrier", [&](dim3 g, dim3 b) {
    return LaunchFn([=](hipStream_t s, int i) { hipLaunchKernelGGL(k_rt1_bar, g, b, 0, s, (i & 1) ? bufB : bufA, (i & 1) ? bufA : bufB); });
  });
  row("2 dependent round trips", [&](dim3 g, dim3 b) {
    return LaunchFn([=](hipStream_t s, int i) { hipLaunchKernelGGL(k_rt2, g, b, 0, s, idx, (i & 1) ? bufB : bufA, (i & 1) ? bufA : bufB); });
  });

  // ---------------- D
  std::printf("\n## D. instruction fetch: straight-line body of N VALU instructions, 256 x 512, us per node\n| body | same kernel back to back | 16 copies in rotation |\n|---|---|---|\n");
  {
    auto same = [&](auto fn) { return LaunchFn([=](hipStream_t s, int) { hipLaunchKernelGGL(fn, dim3(256), dim3(512), 0, s, 1.0f, sink); }); };
    std::printf("| 256 instr (2 KB) | %.2f | %.2f |\n", node_us(c, same(k_code<0, 256>)),
                node_us(c, LaunchFn([=](hipStream_t s, int i) { launch_code_rot<256, 0>(s, i, dim3(256), dim3(512), sink); })));
    std::printf("| 1024 instr (8 KB) | %.2f | %.2f |\n", node_us(c, same(k_code<0, 1024>)),
                node_us(c, LaunchFn([=](hipStream_t s, int i) { launch_code_rot<1024, 0>(s, i, dim3(256), dim3(512), sink); })));
    std::printf("| 4096 instr (32 KB) | %.2f | %.2f |\n", node_us(c, same(k_code<0, 4096>)),
                node_us(c, LaunchFn([=](hipStream_t s, int i) { launch_code_rot<4096, 0>(s, i, dim3(256), dim3(512), sink); })));
    std::fflush(stdout);
  }

  // ---------------- C
  std::printf("\n## C. timeline with in-kernel stamps (100 MHz clock, 10 ns): chain = [stream S MB] [small] [small] repeated\n");
  {
    const size_t wbytes = (size_t)1 << 30;   // 1 GiB of weights to cycle through (cold)
    u32x4* w;
    CHECK(hipMalloc(&w, wbytes));
    CHECK(hipMemset(w, 1, wbytes));
    ull* t;
    const int kSlots = 512;
    CHECK(hipMalloc(&t, kSlots * 4 * sizeof(ull)));
    std::vector<ull> init(kSlots * 4), h(kSlots * 4);
    for (int i = 0; i < kSlots; i++) { init[4 * i] = ~0ull; init[4 * i + 1] = 0; init[4 * i + 2] = 0; init[4 * i + 3] = 0; }
    for (int smb : {0, 8, 32, 64}) {
      for (int cfg = 0; cfg < 3; cfg++) {
        const dim3 sg = cfg == 0 ? dim3(32) : cfg == 1 ? dim3(256) : dim3(256);
        const dim3 sb = cfg == 0 ? dim3(128) : cfg == 1 ? dim3(256) : dim3(512);
        const int n_rep = 24;
        // slots: per repetition 3 nodes (stream, small rt=1, small rt=2)
        hipGraph_t g; hipGraphExec_t ge;
        CHECK(hipStreamBeginCapture(c.st, hipStreamCaptureModeThreadLocal));
        size_t off16 = 0;
        const size_t n16 = (size_t)smb * 1024 * 1024 / 16;
        for (int r = 0; r < n_rep; r++) {
          if (smb > 0) {
            if ((off16 + n16) * 16 > wbytes) off16 = 0;
            hipLaunchKernelGGL(k_stream, dim3(256), dim3(512), 0, c.st, w + off16, (ull)(n16 / 256), sink, t, (unsigned)(3 * r));
            off16 += n16;
          } else {
            hipLaunchKernelGGL(k_small_stamp, sg, sb, 0, c.st, bufA, bufB, t, (unsigned)(3 * r), 0);
          }
          hipLaunchKernelGGL(k_small_stamp, sg, sb, 0, c.st, bufA, bufB, t, (unsigned)(3 * r + 1), 1);
          hipLaunchKernelGGL(k_small_stamp, sg, sb, 0, c.st, bufB, bufA, t, (unsigned)(3 * r + 2), 2);
        }
        CHECK(hipStreamEndCapture(c.st, &g));
        CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        double acc[3][3] = {{0}};   // [node kind][gap before, ramp, span]
        int cnt = 0;
        for (int it = 0; it < 6; it++) {
          CHECK(hipMemcpy(t, init.data(), init.size() * sizeof(ull), hipMemcpyHostToDevice));
          CHECK(hipGraphLaunch(ge, c.st));
          CHECK(hipStreamSynchronize(c.st));
          if (it < 2) continue;
          CHECK(hipMemcpy(h.data(), t, h.size() * sizeof(ull), hipMemcpyDeviceToHost));
          for (int r = 1; r < n_rep; r++)
            for (int k = 0; k < 3; k++) {
              const int s = 3 * r + k;
              acc[k][0] += (double)(h[4 * s] - h[4 * (s - 1) + 2]) * 0.01;
              acc[k][1] += (double)(h[4 * s + 1] - h[4 * s]) * 0.01;
              acc[k][2] += (double)(h[4 * s + 2] - h[4 * s]) * 0.01;
            }
          cnt += n_rep - 1;
        }
        std::printf("stream %2d MB, small = %u x %u:", smb, sg.x, sb.x);
        const char* nm[3] = {smb ? "stream" : "small rt0", "small rt1", "small rt2"};
        for (int k = 0; k < 3; k++)
          std::printf("  [%s: gap %.2f ramp %.2f span %.2f]", nm[k], acc[k][0] / cnt, acc[k][1] / cnt, acc[k][2] / cnt);
        std::printf("\n");
        std::fflush(stdout);
        CHECK(hipGraphExecDestroy(ge));
        CHECK(hipGraphDestroy(g));
      }
    }
    CHECK(hipFree(w));
    CHECK(hipFree(t));
  }
  std::printf("\ndone\n");
  return 0;
}
