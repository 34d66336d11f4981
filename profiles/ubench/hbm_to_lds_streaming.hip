// HBM -> LDS streaming rate per loader wave with every CU streaming (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TIME(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
// mode 0: dword + dwordx4 DMA per 1280-B tile; mode 1: dwordx4 DMA only, 1 KB chunks; mode 2: plain dwordx4 loads (8 in flight) + ds_write_b128
template <int MODE>
__global__ void k(const uint32_t *src, unsigned long long *out, int kb, int waves_used) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= waves_used) return;
  constexpr int RS = 20;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * RS * 1280;
  const uint32_t *g = src + ((size_t)blockIdx.x * 4 + wave) * (size_t)kb * 256;
  unsigned long long t0, t1;
  TIME(t0);
  if (MODE == 0) {
    const int tiles = kb * 1024 / 1280;
    for (int i = 0; i < tiles; ++i) {
      const uint32_t *p = g + (size_t)i * 320;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (i % RS) * 1280);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(p + lane) : "m0", "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst + 256), "v"(p + 64 + lane * 4) : "m0", "memory");
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  } else if (MODE == 1) {
    for (int i = 0; i < kb; ++i) {
      const uint32_t *p = g + (size_t)i * 256;
      const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (i % RS) * 1024);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(p + lane * 4) : "m0", "memory");
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
  } else {
    for (int i = 0; i + 8 <= kb; i += 8) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const uint4 *>(g + (size_t)(i + j) * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4 *>(lds + wave * RS * 320 + ((i + j) % RS) * 256 + lane * 4) = v[j];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TIME(t1);
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (lds[lane] == 0x12345) out[4000] = 1;
}
int main() {
  const int kb = 192;  // KiB per wave
  uint32_t *src; unsigned long long *d, h[1024];
  const size_t words = (size_t)256 * 4 * kb * 256;
  hipMalloc(&src, words * 4); hipMemset(src, 1, words * 4); hipMalloc(&d, 8192 * 8);
  uint32_t *junk; hipMalloc(&junk, 600u << 20);
  const char *names[] = {"DMA dword+dwordx4 per 1280-B tile", "DMA dwordx4 only (1 KiB chunks)", "plain dwordx4 loads x8 + ds_write_b128"};
  const size_t lds = 4 * 20 * 1280;
  hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int waves : {1, 2, 4})
    for (int mode = 0; mode < 3; ++mode) {
      hipMemset(junk, 0, 600u << 20);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), lds, 0, src, d, kb, waves);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), lds, 0, src, d, kb, waves);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(256), lds, 0, src, d, kb, waves);
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, 1024 * 8, hipMemcpyDeviceToHost);
      double avg = 0; int n = 0; for (int b = 0; b < 256; ++b) for (int w = 0; w < waves; ++w) { avg += (double)h[b * 4 + w]; ++n; } avg /= n;
      printf("256 blocks, %d waves/block, %-40s %7.1f cycles per KiB per wave  (kernel %.1f us, %.0f GB/s)\n", waves, names[mode], avg / kb, ms * 1e3,
             (double)256 * waves * kb * 1024 / (ms * 1e-3) / 1e9);
    }
  return 0;
}
