// loader-only streaming test: tile-shaped LDS-DMA (dword + dwordx4 per 1280-B tile) from HBM (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TIME(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
template <int AHEAD>
__global__ void k(const uint32_t *src, unsigned long long *out, int tiles, int waves_used) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= waves_used) return;
  constexpr int RS = AHEAD + 4;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * RS * 1280;
  const uint32_t *g = src + ((size_t)blockIdx.x * 4 + wave) * (size_t)tiles * 320;
  unsigned long long t0, t1;
  TIME(t0);
  for (int i = 0; i < tiles; ++i) {
    const uint32_t *p = g + (size_t)i * 320;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (i % RS) * 1280);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(p + lane) : "m0", "memory");
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst + 256), "v"(p + 64 + lane * 4) : "m0", "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * AHEAD) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  TIME(t1);
  if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
  if (lds[lane] == 0x12345) out[4000] = 1;
}
int main() {
  const int tiles = 150;
  uint32_t *src; unsigned long long *d, h[1024];
  const size_t words = (size_t)256 * 4 * tiles * 320;
  hipMalloc(&src, words * 4); hipMemset(src, 1, words * 4); hipMalloc(&d, 8192 * 8);
  // flush: touch another big buffer between runs
  uint32_t *junk; hipMalloc(&junk, 600u << 20);
  for (int blocks : {8, 256})
    for (int waves : {1, 2, 4})
      for (int ahead : {4, 8, 16, 28}) {
        hipMemset(junk, 0, 600u << 20);
        const size_t lds = (size_t)waves * (ahead + 4) * 1280;
        if (lds > 160 * 1024) continue;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (ahead == 4) { hipFuncSetAttribute((const void*)k<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), lds, 0, src, d, tiles, waves); }
        if (ahead == 8) { hipFuncSetAttribute((const void*)k<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), lds, 0, src, d, tiles, waves); }
        if (ahead == 16) { hipFuncSetAttribute((const void*)k<16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<16>, dim3(blocks), dim3(256), lds, 0, src, d, tiles, waves); }
        if (ahead == 28) { hipFuncSetAttribute((const void*)k<28>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); hipLaunchKernelGGL(k<28>, dim3(blocks), dim3(256), lds, 0, src, d, tiles, waves); }
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, d, 1024 * 8, hipMemcpyDeviceToHost);
        double avg = 0; int n = 0; for (int b = 0; b < blocks; ++b) for (int w = 0; w < waves; ++w) { avg += (double)h[b * 4 + w]; ++n; } avg /= n;
        printf("blocks %3d waves/block %d ahead %2d: %7.1f cycles per tile per wave  (kernel %.1f us, %.0f GB/s)\n", blocks, waves, ahead, avg / tiles,
               ms * 1e3, (double)blocks * waves * tiles * 1280 / (ms * 1e-3) / 1e9);
      }
  return 0;
}
