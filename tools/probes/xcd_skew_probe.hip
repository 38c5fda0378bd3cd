// xcd_skew_probe.hip — why do the workgroups of one mat-vec launch end up to 30 % apart, XCD by XCD?
// (tools/token_timeline.py, r03a: in the Llama-3-8B gate/up launch all 28 workgroups of one XCD end within 0.6 us of each
// other, but the XCDs end between 12.3 and 17.2 us; which XCD is slow changes from layer to layer, the shader clocks are
// equal to 3 % — so it follows the ADDRESSES, not the silicon.)
// The probe streams a matrix the way mvq_kernel does — one workgroup per CU, 8 waves, wave ks reads blocks [ks*nbw,(ks+1)*nbw)
// of every 16-row tile of its workgroup, 4 tiles of 2304 B in flight per wave, nt loads — and varies only WHICH tiles a
// workgroup gets:
//   map 0  identity: workgroup b owns tiles [b*R, (b+1)*R)           (what the engine does)
//   map 1  XCD-major: the workgroups of one XCD (b % 8) own one contiguous eighth of the matrix
//   map 2  tile-interleaved: workgroup b owns tiles b, b + n_wg, b + 2 n_wg, ...
//   map 3  identity with every workgroup's tile order rotated by b (workgroups of one XCD do not walk in lockstep)
// Per configuration: kernel time (cold weights: 24 copies cycled), and the spread of workgroup end times by physical XCD.
// Build: hipcc --offload-arch=gfx950 -O3 -o xcd_skew_probe xcd_skew_probe.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                        \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

typedef unsigned long long ull;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct Stamp { ull t0, t1; unsigned xcc, pad; };

// tiles_total 16-row tiles of nblk blocks of tb bytes; a workgroup owns R tiles (per pass); npass passes, pass p at +p*plane
__global__ void __launch_bounds__(512) k_mv_stream(const unsigned char* __restrict__ w, unsigned tiles_total, unsigned nblk, unsigned tb,
                                                   unsigned R, unsigned npass, ull plane, int map, float* sink, Stamp* st) {
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = blockIdx.x, n_wg = gridDim.x;
  if (wave == 0) { st[b].t0 = __builtin_amdgcn_s_memrealtime(); st[b].xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF; }
  const unsigned nbw = nblk / 8;                       // blocks per wave (T = 8 k-slices)
  const unsigned blk0 = wave * nbw;
  auto tile_of = [&](unsigned j) -> unsigned {         // j-th tile of this workgroup
    if (map == 1) { const unsigned per = n_wg / 8; return ((b % 8) * per + b / 8) * R + j; }
    if (map == 2) return b + j * n_wg;
    if (map == 3) return b * R + (j + b) % R;
    return b * R + j;
  };
  const unsigned nitems = npass * R * nbw;
  u32x4 acc = {0, 0, 0, 0};
  auto addr = [&](unsigned it) -> const unsigned char* {
    const unsigned p = it / (R * nbw), r = it % (R * nbw), j = r / nbw, bb = r % nbw;
    unsigned t = tile_of(j);
    if (t >= tiles_total) t = tiles_total - 1;
    return w + (ull)p * plane + ((ull)t * nblk + blk0 + bb) * tb + lane * 16;
  };
  // 4 items in flight, three 16-byte loads per lane and item (header + two nibble halves: 2304 B = 2 x 1024 + 256)
  constexpr int D = 4;
  u32x4 q0[D], q1[D], hd[D];
  // static pipeline (the compiler counts vmcnt exactly): every slot is refilled unconditionally, past the end with the last item again
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
  __syncthreads();
  if (wave == 0) st[b].t1 = __builtin_amdgcn_s_memrealtime();
}

// The same stream through an LDS ring filled by LDS-DMA ("weight stream by LDS-DMA nt into an LDS ring"): every
// wave keeps RING tiles of 2304 B in flight in its own 16 KB of LDS (8 waves x 7 tiles = 126 KB per CU against the 72 KB the
// register version holds), waits for the oldest with a counted vmcnt, reads it back with three ds_read_b128 / b32 per lane (what
// a consumer would do) and refills the slot.  NT: non-temporal DMA (aux nt).
template <int RING, bool NT>
__global__ void __launch_bounds__(512) k_mv_stream_lds(const unsigned char* __restrict__ w, unsigned tiles_total, unsigned nblk, unsigned tb,
                                                       unsigned R, unsigned npass, ull plane, float* sink, Stamp* st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const unsigned b = blockIdx.x;
  if (wave == 0) { st[b].t0 = __builtin_amdgcn_s_memrealtime(); st[b].xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF; }
  const unsigned nbw = nblk / 8, blk0 = wave * nbw;
  const unsigned nitems = npass * R * nbw;
  auto addr = [&](unsigned it) -> const unsigned char* {
    if (it >= nitems) it = nitems - 1;
    const unsigned p = it / (R * nbw), r = it % (R * nbw), j = r / nbw, bb = r % nbw;
    unsigned t = b * R + j;
    if (t >= tiles_total) t = tiles_total - 1;
    return w + (ull)p * plane + ((ull)t * nblk + blk0 + bb) * tb;
  };
  constexpr unsigned kSlot = 2304;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem + wave * RING * kSlot;
  auto dma = [&](unsigned slot, const unsigned char* src) {
    const unsigned dst = lds0 + slot * kSlot;
    unsigned keep;
    if (NT)
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, off nt\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dword %3, off nt\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + lane * 16), "v"(src + 1024 + lane * 16), "v"(src + 2048 + lane * 4), "s"(dst) : "memory");
    else
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, off\n\ts_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dword %3, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src + lane * 16), "v"(src + 1024 + lane * 16), "v"(src + 2048 + lane * 4), "s"(dst) : "memory");
  };
  u32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int j = 0; j < RING; j++) dma(j, addr(j));
  const unsigned char* my = smem + wave * RING * kSlot;
  for (unsigned it = 0; it < nitems; it += RING) {
#pragma unroll
    for (int j = 0; j < RING; j++) {
      // RING - 1 younger tiles (3 DMA instructions each) may stay in flight
      if constexpr (RING == 7) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
      else if constexpr (RING == 4) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const u32x4 a = *(const u32x4*)(my + j * kSlot + lane * 16);
      const u32x4 c = *(const u32x4*)(my + j * kSlot + 1024 + lane * 16);
      const unsigned h = *(const unsigned*)(my + j * kSlot + 2048 + lane * 4);
      acc ^= a ^ c;
      acc.x ^= h;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc) :: "memory");   // the slot has been read: it may be refilled
      dma(j, addr(it + RING + j));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = 1.0f;
  __syncthreads();
  if (wave == 0) st[b].t1 = __builtin_amdgcn_s_memrealtime();
}

// a latency-bound interlude: 32 small workgroups that idle for `us` microseconds (the attention nodes between two mat-vecs)
__global__ void k_idle(unsigned us, float* sink) {
  const ull t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (ull)us * 100) __builtin_amdgcn_s_sleep(8);
  if (us == 0xFFFFFFFFu) sink[threadIdx.x] = 1.0f;
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
  Stamp* st;
  CHECK(hipMalloc(&st, sizeof(Stamp) * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  struct Cfg { const char* name; unsigned tiles, nblk, tb, R, npass, n_wg; };
  const Cfg cfgs[] = {
      {"gate_up 14336x4096 Q4_K x2, 224 wg x 4 tiles", 896, 16, 2304, 4, 2, 224},
      {"gate_up, 256 wg x 4 tiles (last 32 clamp)", 896, 16, 2304, 4, 2, 256},
      {"down 4096x14336 Q4_K, 256 wg x 1 tile", 256, 56, 2304, 1, 1, 256},
      {"wo 4096x4096 Q4_K, 256 wg x 1 tile", 256, 16, 2304, 1, 1, 256},
      {"output 128256x4096 Q6_K-sized, 251 wg x 32 tiles", 8016, 16, 3392, 32, 1, 251},
  };
  for (const Cfg& c : cfgs) {
    const size_t plane = (size_t)c.tiles * c.nblk * c.tb;
    const size_t mat = plane * c.npass;
    const size_t stride = (mat + (2u << 20) + 4095) / 4096 * 4096 + 4096 * 37;   // copies at odd 4 KB offsets, like separate allocations
    const int copies = (int)std::min<size_t>(24, pool / stride);
    std::printf("\n## %s  (%.1f MB, %d copies)\n| map | us per launch | TB/s | ends by XCD: median of each (us) | first / median / last end | body 256-wg equal share would be |\n|---|---|---|---|---|---|\n",
                c.name, mat / 1e6, copies);
    for (int map = 0; map < 3; map += 2) {
      if (map == 1 && c.n_wg % 8) continue;
      if (map == 2 && c.tiles < c.n_wg * c.R) { /* interleaved needs full coverage */ }
      auto launch = [&](int i) {
        hipLaunchKernelGGL(k_mv_stream, dim3(c.n_wg), dim3(512), 0, s, w + (size_t)(i % copies) * stride, c.tiles, c.nblk, c.tb, c.R, c.npass,
                           (ull)plane, map, sink, st);
      };
      for (int i = 0; i < 5; i++) launch(i);
      CHECK(hipStreamSynchronize(s));
      const int iters = 48;
      CHECK(hipEventRecord(e0, s));
      for (int i = 0; i < iters; i++) launch(i + 5);
      CHECK(hipEventRecord(e1, s));
      CHECK(hipStreamSynchronize(s));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      // spread of single launches: copy 1 three times (is the slow XCD reproducible for one address?), then other copies
      double fst = 0, med = 0, lst = 0;
      const int reps = 6;
      std::vector<std::string> lines;
      for (int r = 0; r < reps; r++) {
        const int copy = r < 3 ? 1 : r * 3 + 1;
        launch(copy);
        CHECK(hipStreamSynchronize(s));
        std::vector<Stamp> h(c.n_wg);
        CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * c.n_wg, hipMemcpyDeviceToHost));
        ull base = ~0ull;
        for (auto& x : h) base = std::min(base, x.t0);
        std::vector<double> all;
        std::vector<double> per[8];
        for (auto& x : h) { const double e = (double)(x.t1 - base) * 0.01; all.push_back(e); per[x.xcc & 7].push_back(e); }
        std::sort(all.begin(), all.end());
        fst += all.front(); med += all[all.size() / 2]; lst += all.back();
        char buf[256]; int o = std::snprintf(buf, sizeof(buf), "    copy %2d:", copy % copies);
        for (int x = 0; x < 8; x++)
          if (!per[x].empty()) { std::sort(per[x].begin(), per[x].end()); o += std::snprintf(buf + o, sizeof(buf) - o, " %5.1f", per[x][per[x].size() / 2]); }
        o += std::snprintf(buf + o, sizeof(buf) - o, "   first %.2f last %.2f", all.front(), all.back());
        lines.push_back(buf);
      }
      const double us = ms * 1000.0 / iters;
      std::printf("| %d | %.2f | %.2f | see below | %.2f / %.2f / %.2f | %.2f |\n", map, us, mat / us / 1e6, fst / reps, med / reps, lst / reps, mat / 256.0 / 24.6e3);
      for (auto& l : lines) std::printf("%s\n", l.c_str());
      std::fflush(stdout);
    }
  }
  // ---- F: the same streams through LDS-DMA rings (7 or 4 tiles per wave; default policy or nt) against the register version
  {
    std::printf("\n## F. weight stream through an LDS-DMA ring vs registers (us per launch eager, cold copies; first / median / last workgroup end in-kernel)\n| shape | registers, 4 tiles/wave (nt) | LDS ring 4 (nt) | LDS ring 7 (default policy) | LDS ring 7 (nt) |\n|---|---|---|---|---|\n");
    hipFuncSetAttribute((const void*)k_mv_stream_lds<7, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_mv_stream_lds<7, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)k_mv_stream_lds<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (const Cfg& c : cfgs) {
      if (c.tb != 2304) continue;
      const size_t plane = (size_t)c.tiles * c.nblk * c.tb, mat = plane * c.npass;
      const size_t stride = (mat + (2u << 20) + 4095) / 4096 * 4096 + 4096 * 37;
      const int copies = (int)std::min<size_t>(24, pool / stride);
      std::printf("| %s |", c.name);
      for (int v = 0; v < 4; v++) {
        auto launch = [&](int i) {
          const unsigned char* base = w + (size_t)(i % copies) * stride;
          if (v == 0) hipLaunchKernelGGL(k_mv_stream, dim3(c.n_wg), dim3(512), 0, s, base, c.tiles, c.nblk, c.tb, c.R, c.npass, (ull)plane, 0, sink, st);
          else if (v == 1) hipLaunchKernelGGL((k_mv_stream_lds<4, true>), dim3(c.n_wg), dim3(512), 8 * 4 * 2304, s, base, c.tiles, c.nblk, c.tb, c.R, c.npass, (ull)plane, sink, st);
          else if (v == 2) hipLaunchKernelGGL((k_mv_stream_lds<7, false>), dim3(c.n_wg), dim3(512), 8 * 7 * 2304, s, base, c.tiles, c.nblk, c.tb, c.R, c.npass, (ull)plane, sink, st);
          else hipLaunchKernelGGL((k_mv_stream_lds<7, true>), dim3(c.n_wg), dim3(512), 8 * 7 * 2304, s, base, c.tiles, c.nblk, c.tb, c.R, c.npass, (ull)plane, sink, st);
        };
        for (int i = 0; i < 5; i++) launch(i);
        CHECK(hipStreamSynchronize(s));
        const int iters = 48;
        CHECK(hipEventRecord(e0, s));
        for (int i = 0; i < iters; i++) launch(i + 5);
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double fst = 0, med = 0, lst = 0;
        for (int r = 0; r < 4; r++) {
          launch(r * 5 + 2);
          CHECK(hipStreamSynchronize(s));
          std::vector<Stamp> h(c.n_wg);
          CHECK(hipMemcpy(h.data(), st, sizeof(Stamp) * c.n_wg, hipMemcpyDeviceToHost));
          ull base = ~0ull;
          for (auto& x : h) base = std::min(base, x.t0);
          std::vector<double> all;
          for (auto& x : h) all.push_back((double)(x.t1 - base) * 0.01);
          std::sort(all.begin(), all.end());
          fst += all.front() / 4; med += all[all.size() / 2] / 4; lst += all.back() / 4;
        }
        std::printf(" %.2f (%.2f / %.2f / %.2f) |", ms * 1000.0 / iters, fst, med, lst);
        std::fflush(stdout);
      }
      std::printf("\n");
    }
  }
  // ---- E: does the skew come from what runs BEFORE a streaming launch?  One graph: 12 gate_up-sized launches (map 0), separated by
  // nothing / an idle interlude of 4, 8 or 16 us / a short streaming launch (wo-sized).  Per launch: first and last workgroup end
  // and the median end of the fastest and the slowest XCD.
  {
    const Cfg c = cfgs[0];
    const size_t plane = (size_t)c.tiles * c.nblk * c.tb, mat = plane * c.npass;
    const size_t stride = (mat + (2u << 20) + 4095) / 4096 * 4096 + 4096 * 37;
    const int copies = (int)std::min<size_t>(24, pool / stride);
    Stamp* stn;
    const int kN = 12;
    CHECK(hipMalloc(&stn, sizeof(Stamp) * 256 * kN));
    std::printf("\n## E. gate_up-sized launches inside ONE graph, by what separates them (mean over launches 2..12 and 5 replays)\n| interlude | first end | median end | last end | fastest XCD median | slowest XCD median | graph time per (launch + interlude) us |\n|---|---|---|---|---|---|---|\n");
    for (int mode = 0; mode < 6; mode++) {
      hipGraph_t g; hipGraphExec_t ge;
      CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < kN; i++) {
        hipLaunchKernelGGL(k_mv_stream, dim3(c.n_wg), dim3(512), 0, s, w + (size_t)((i * 2 + 1) % copies) * stride, c.tiles, c.nblk, c.tb, c.R, c.npass,
                           (ull)plane, 0, sink, stn + 256 * i);
        if (mode >= 1 && mode <= 3) hipLaunchKernelGGL(k_idle, dim3(32), dim3(128), 0, s, mode == 1 ? 4u : mode == 2 ? 8u : 16u, sink);
        if (mode == 4) hipLaunchKernelGGL(k_mv_stream, dim3(256), dim3(512), 0, s, w + (size_t)((i * 2 + 2) % copies) * stride, 256u, 16u, 2304u, 1u, 1u, (ull)0, 0, sink, stn + 256 * kN - 256);
        if (mode == 5) { hipLaunchKernelGGL(k_idle, dim3(32), dim3(128), 0, s, 3u, sink); hipLaunchKernelGGL(k_idle, dim3(32), dim3(128), 0, s, 3u, sink); }
      }
      CHECK(hipStreamEndCapture(s, &g));
      CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      double a_f = 0, a_m = 0, a_l = 0, a_fx = 0, a_sx = 0, a_t = 0;
      int cnt = 0;
      for (int rep = 0; rep < 7; rep++) {
        CHECK(hipEventRecord(e0, s));
        CHECK(hipGraphLaunch(ge, s));
        CHECK(hipEventRecord(e1, s));
        CHECK(hipStreamSynchronize(s));
        if (rep < 2) continue;
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        a_t += ms * 1000.0 / kN;
        std::vector<Stamp> h(256 * kN);
        CHECK(hipMemcpy(h.data(), stn, sizeof(Stamp) * 256 * kN, hipMemcpyDeviceToHost));
        for (int i = 1; i < (mode == 4 ? kN - 1 : kN); i++) {
          ull base = ~0ull;
          for (unsigned b = 0; b < c.n_wg; b++) base = std::min(base, h[256 * i + b].t0);
          std::vector<double> all, per[8];
          for (unsigned b = 0; b < c.n_wg; b++) { const double e = (double)(h[256 * i + b].t1 - base) * 0.01; all.push_back(e); per[h[256 * i + b].xcc & 7].push_back(e); }
          std::sort(all.begin(), all.end());
          double fx = 1e9, sx = 0;
          for (int x = 0; x < 8; x++) if (!per[x].empty()) { std::sort(per[x].begin(), per[x].end()); const double m = per[x][per[x].size() / 2]; fx = std::min(fx, m); sx = std::max(sx, m); }
          a_f += all.front(); a_m += all[all.size() / 2]; a_l += all.back(); a_fx += fx; a_sx += sx; cnt++;
        }
      }
      const char* nm[6] = {"none (back to back)", "idle 4 us", "idle 8 us", "idle 16 us", "a wo-sized streaming launch", "two idle 3 us nodes"};
      std::printf("| %s | %.2f | %.2f | %.2f | %.2f | %.2f | %.2f |\n", nm[mode], a_f / cnt, a_m / cnt, a_l / cnt, a_fx / cnt, a_sx / cnt, a_t / 5);
      std::fflush(stdout);
      CHECK(hipGraphExecDestroy(ge));
      CHECK(hipGraphDestroy(g));
    }
  }
  std::printf("\ndone\n");
  return 0;
}
