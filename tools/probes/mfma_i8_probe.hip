// Probe (diagnostic, not part of the product): lane layout of v_mfma_i32_16x16x32_i8 / 32x32x32_i8 on gfx950,
// and sustained issue cost of the VALU instructions the dequant mat-vec is made of.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

__global__ void k16(const long* a, const long* b, i32x4* d) {
  i32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_i32_16x16x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d[threadIdx.x] = c;
}
__global__ void k32(const i32x4* a, const i32x4* b, i32x16* d) {
  i32x16 c = {};
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0);
  d[threadIdx.x] = c;
}

// VALU issue probe: N iterations of a block of 8 independent ops per iteration
template <int KIND>
__global__ void valu(float* out, int iters, unsigned long long* cyc) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  unsigned u0 = threadIdx.x * 2654435761u, u1 = u0 + 77, u2 = u0 * 3, u3 = u0 ^ 0x5555;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {  // fma, 8 independent chains
      a0 = __builtin_fmaf(a0, 1.0001f, 0.5f); a1 = __builtin_fmaf(a1, 1.0001f, 0.5f); a2 = __builtin_fmaf(a2, 1.0001f, 0.5f); a3 = __builtin_fmaf(a3, 1.0001f, 0.5f);
      a4 = __builtin_fmaf(a4, 1.0001f, 0.5f); a5 = __builtin_fmaf(a5, 1.0001f, 0.5f); a6 = __builtin_fmaf(a6, 1.0001f, 0.5f); a7 = __builtin_fmaf(a7, 1.0001f, 0.5f);
    } else if (KIND == 1) {  // cvt_f32_ubyte + fma (the dequant pair)
      asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a0) : "v"(u0)); asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a1) : "v"(u0));
      asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a2) : "v"(u1)); asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(a3) : "v"(u1));
      asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a4) : "v"(u2)); asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a5) : "v"(u2));
      asm volatile("v_cvt_f32_ubyte2 %0, %1" : "=v"(a6) : "v"(u3)); asm volatile("v_cvt_f32_ubyte3 %0, %1" : "=v"(a7) : "v"(u3));
    } else if (KIND == 2) {  // integer and
      asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(u0)); asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(u1));
      asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(u2)); asm volatile("v_and_b32 %0, 0x0f0f0f0f, %0" : "+v"(u3));
      asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u0)); asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u1));
      asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u2)); asm volatile("v_lshrrev_b32 %0, 4, %0" : "+v"(u3));
    } else {  // v_pk_fma_f32
      typedef float f2 __attribute__((ext_vector_type(2)));
      f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, m = {1.0001f, 1.0001f}, c = {0.5f, 0.5f};
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(m), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(m), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(m), "v"(c));
      asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(m), "v"(c)); asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(m), "v"(c));
      a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(u0 + u1 + u2 + u3);
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// MFMA issue probe
__global__ void mfma_rate(i32x4* out, int iters, unsigned long long* cyc) {
  long a = threadIdx.x * 0x0101010101010101L, b = 0x0203040506070809L;
  i32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
    c0 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_i32_16x16x32_i8(a, b, c3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main() {
  // ---- layout check 16x16x32
  std::vector<int8_t> A(16 * 32), B(32 * 16);
  srand(1);
  for (auto& v : A) v = (int8_t)(rand() % 255 - 127);
  for (auto& v : B) v = (int8_t)(rand() % 255 - 127);
  std::vector<long> ha(64), hb(64);
  for (int l = 0; l < 64; l++) {
    long va = 0, vb = 0;
    for (int j = 0; j < 8; j++) {
      int k = 8 * (l >> 4) + j;
      va |= (long)(uint8_t)A[(l & 15) * 32 + k] << (8 * j);   // A[m = l&15][k]
      vb |= (long)(uint8_t)B[k * 16 + (l & 15)] << (8 * j);   // B[k][n = l&15]
    }
    ha[l] = va; hb[l] = vb;
  }
  long *da, *db; i32x4* dd;
  CK(hipMalloc(&da, 64 * 8)); CK(hipMalloc(&db, 64 * 8)); CK(hipMalloc(&dd, 64 * 16));
  CK(hipMemcpy(da, ha.data(), 64 * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 64 * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, da, db, dd);
  std::vector<int> hd(64 * 4);
  CK(hipMemcpy(hd.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; l++) for (int i = 0; i < 4; i++) {
    int m = 4 * (l >> 4) + i, n = l & 15, ref = 0;
    for (int k = 0; k < 32; k++) ref += (int)A[m * 32 + k] * (int)B[k * 16 + n];
    if (ref != hd[l * 4 + i]) bad++;
  }
  printf("16x16x32_i8 layout: A[m=l&15][k=8(l>>4)+j], B[k][n=l&15], D[m=4(l>>4)+i][n=l&15]: %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
  // ---- layout check 32x32x32: assume A lane l: m = l&31, k = 16(l>>5)+j (16 bytes); D reg r: m = (r&3)+8(r>>2)+4(l>>5), n = l&31
  {
    std::vector<int8_t> A2(32 * 32), B2(32 * 32);
    for (auto& v : A2) v = (int8_t)(rand() % 255 - 127);
    for (auto& v : B2) v = (int8_t)(rand() % 255 - 127);
    std::vector<int8_t> pa(64 * 16), pb(64 * 16);
    for (int l = 0; l < 64; l++) for (int j = 0; j < 16; j++) {
      int k = 16 * (l >> 5) + j;
      pa[l * 16 + j] = A2[(l & 31) * 32 + k];
      pb[l * 16 + j] = B2[k * 32 + (l & 31)];
    }
    i32x4 *a4, *b4; i32x16* d16;
    CK(hipMalloc(&a4, 1024)); CK(hipMalloc(&b4, 1024)); CK(hipMalloc(&d16, 64 * 64));
    CK(hipMemcpy(a4, pa.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(b4, pb.data(), 1024, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, a4, b4, d16);
    std::vector<int> h16(64 * 16);
    CK(hipMemcpy(h16.data(), d16, 64 * 64, hipMemcpyDeviceToHost));
    int bad2 = 0;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) {
      int m = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), n = l & 31, ref = 0;
      for (int k = 0; k < 32; k++) ref += (int)A2[m * 32 + k] * (int)B2[k * 32 + n];
      if (ref != h16[l * 16 + r]) bad2++;
    }
    printf("32x32x32_i8 layout: A[m=l&31][k=16(l>>5)+j], D[m=(r&3)+8(r>>2)+4(l>>5)][n=l&31]: %s (%d mismatches)\n", bad2 ? "WRONG" : "OK", bad2);
  }
  // ---- issue-rate probes: 256 CUs x (waves per SIMD = 1, 2, 4)
  float* outf; unsigned long long* cyc; i32x4* outi;
  CK(hipMalloc(&outf, 1024 * 1024 * 4)); CK(hipMalloc(&cyc, 4096 * 8)); CK(hipMalloc(&outi, 1024 * 1024 * 16));
  const int iters = 20000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[4] = {"v_fma_f32 (8 chains)", "v_cvt_f32_ubyteN", "v_and/v_lshr", "v_pk_fma_f32"};
  for (int wps = 1; wps <= 4; wps *= 2) {
    int threads = 256 * wps;  // one workgroup per CU, wps waves per SIMD
    for (int kind = 0; kind < 4; kind++) {
      CK(hipEventRecord(e0, 0));
      if (kind == 0) hipLaunchKernelGGL(valu<0>, dim3(256), dim3(threads), 0, 0, outf, iters, cyc);
      if (kind == 1) hipLaunchKernelGGL(valu<1>, dim3(256), dim3(threads), 0, 0, outf, iters, cyc);
      if (kind == 2) hipLaunchKernelGGL(valu<2>, dim3(256), dim3(threads), 0, 0, outf, iters, cyc);
      if (kind == 3) hipLaunchKernelGGL(valu<3>, dim3(256), dim3(threads), 0, 0, outf, iters, cyc);
      CK(hipEventRecord(e1, 0));
      CK(hipDeviceSynchronize());
      float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
      unsigned long long c[256];
      CK(hipMemcpy(c, cyc, 256 * 8, hipMemcpyDeviceToHost));
      double avg = 0; for (int i = 0; i < 256; i++) avg += (double)c[i]; avg /= 256;
      // per SIMD: wps waves each issuing iters*8 instructions
      printf("%-22s waves/SIMD=%d: %.2f memtime-ticks, %.3f ns wall per wave-instruction per SIMD (kernel %.3f ms, %.0f ticks -> %.2f GHz)\n", names[kind], wps,
             avg / ((double)iters * 8 * wps), ms * 1e6 / ((double)iters * 8 * wps), ms, avg, avg / (ms * 1e6));
    }
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_rate, dim3(256), dim3(threads), 0, 0, outi, iters, cyc);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c[256];
    CK(hipMemcpy(c, cyc, 256 * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (int i = 0; i < 256; i++) avg += (double)c[i]; avg /= 256;
    printf("%-22s waves/SIMD=%d: %.2f memtime-ticks, %.3f ns wall per MFMA per SIMD\n", "mfma_i32_16x16x32_i8", wps, avg / ((double)iters * 4 * wps), ms * 1e6 / ((double)iters * 4 * wps));
  }
  return 0;
}
