// Achievable HBM read rate on this card by access pattern (diagnostic only; r02).
//   A: classic grid-stride sweep (consecutive 1 KiB pieces go to consecutive workgroups)
//   B: every workgroup streams its own contiguous region (the tile-program pattern), W waves per
//      workgroup each with its own sub-region, D x 1 KiB in flight per wave
//   C: as B but through LDS-DMA (global_load_lds_dwordx4) into a ring
// hipcc -O2 --offload-arch=gfx950 read_patterns.hip -o read_patterns && ./read_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

typedef uint32_t u4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ntload(const uint4 *p) {
  const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v *>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}

template <int D>
__global__ void k_gridstride(const uint4 *src, size_t n16, float *sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (; i + (D - 1) * stride < n16; i += D * stride) {
    uint4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = ntload(src + i + d * stride);
#pragma unroll
    for (int d = 0; d < D; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345678u) *sink = 1.0f;
}

// region per wave: bytes_per_wave contiguous; D pieces of 1 KiB in flight
template <int D>
__global__ void k_private(const uint4 *src, size_t pieces_per_wave, float *sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  const uint4 *g = src + ((size_t)blockIdx.x * W + wave) * pieces_per_wave * 64 + lane;
  uint32_t acc = 0;
  for (size_t p = 0; p + D <= pieces_per_wave; p += D) {
    uint4 v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = ntload(g + (p + d) * 64);
#pragma unroll
    for (int d = 0; d < D; ++d) acc += v[d].x ^ v[d].y ^ v[d].z ^ v[d].w;
  }
  if (acc == 0x12345678u) *sink = 1.0f;
}

// D: the tile-program case itself: 2 loader waves per workgroup, 148 KiB each, the same 76 MB
// re-read by every launch (it fits the Infinity Cache), with and without the nt policy
template <int D, bool NT>
__global__ void k_short_dma(const uint4 *src, size_t pieces_per_wave, float *sink) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  const uint4 *g = src + ((size_t)blockIdx.x * W + wave) * pieces_per_wave * 64 + lane;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * (D + 1) * 1024;
  for (size_t p = 0; p < pieces_per_wave; ++p) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (uint32_t)(p % (D + 1)) * 1024);
    if (NT) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(dst), "v"(g + p * 64) : "m0", "memory");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(g + p * 64) : "m0", "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[lane] == 0x12345678u) *sink = 1.0f;
}

// E: D plus what the loader wave of the engine does around every DMA: a flag store to LDS after the
// counted wait (FLAGS & 1), a flag load + v_readfirstlane before the issue (FLAGS & 2)
template <int D, int FLAGS>
__global__ void k_short_dma_flags(const uint4 *src, size_t pieces_per_wave, float *sink) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  const uint4 *g = src + ((size_t)blockIdx.x * W + wave) * pieces_per_wave * 64 + lane;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * (D + 1) * 1024;
  volatile uint32_t *flag = lds + W * (D + 1) * 256 + wave * 4;
  uint32_t seen = 0;
  for (size_t p = 0; p < pieces_per_wave; ++p) {
    if (FLAGS & 2) seen += __builtin_amdgcn_readfirstlane(flag[1]);
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (uint32_t)(p % (D + 1)) * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(dst), "v"(g + p * 64) : "m0", "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    if (FLAGS & 1) flag[0] = (uint32_t)p;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[lane] == 0x12345678u + seen) *sink = 1.0f;
}

template <int D>
__global__ void k_private_dma(const uint4 *src, size_t pieces_per_wave, float *sink) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), W = blockDim.x >> 6;
  const uint4 *g = src + ((size_t)blockIdx.x * W + wave) * pieces_per_wave * 64 + lane;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + wave * (D + 1) * 1024;
  for (size_t p = 0; p < pieces_per_wave; ++p) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(base + (uint32_t)(p % (D + 1)) * 1024);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(dst), "v"(g + p * 64) : "m0", "memory");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lds[lane] == 0x12345678u) *sink = 1.0f;
}

int main() {
  const size_t bytes = (size_t)2 << 30;  // 2 GiB: far beyond the 256 MiB Infinity Cache
  uint4 *src; float *sink;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char *name, float ms) { printf("%-64s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, bytes / (ms * 1e-3) / 1e9); };
  float ms;
  char name[128];
#define RUN(NAME, ...)                                   \
  __VA_ARGS__; hipDeviceSynchronize();                   \
  hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1);   \
  hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); report(NAME, ms);
  const size_t n16 = bytes / 16;
  for (int blocks : {1024, 4096}) {
    snprintf(name, 128, "A grid-stride, %d blocks x 256, 4 x 16 B in flight per lane", blocks);
    RUN(name, hipLaunchKernelGGL(k_gridstride<4>, dim3(blocks), dim3(256), 0, 0, src, n16, sink));
    snprintf(name, 128, "A grid-stride, %d blocks x 256, 8 x 16 B in flight per lane", blocks);
    RUN(name, hipLaunchKernelGGL(k_gridstride<8>, dim3(blocks), dim3(256), 0, 0, src, n16, sink));
  }
  for (int W : {1, 2, 4, 8}) {
    const size_t ppw = bytes / 1024 / (256 * W);
    snprintf(name, 128, "B private regions, 256 blocks x %d waves, 8 KiB in flight per wave", W);
    RUN(name, hipLaunchKernelGGL(k_private<8>, dim3(256), dim3(64 * W), 0, 0, src, ppw, sink));
    snprintf(name, 128, "B private regions, 256 blocks x %d waves, 16 KiB in flight per wave", W);
    RUN(name, hipLaunchKernelGGL(k_private<16>, dim3(256), dim3(64 * W), 0, 0, src, ppw, sink));
  }
  for (int W : {1, 2, 4}) {
    const size_t ppw = bytes / 1024 / (256 * W);
    snprintf(name, 128, "C private regions by LDS-DMA, 256 blocks x %d waves, 8 KiB in flight", W);
    RUN(name, hipLaunchKernelGGL(k_private_dma<8>, dim3(256), dim3(64 * W), W * 9 * 1024, 0, src, ppw, sink));
    hipFuncSetAttribute((const void *)k_private_dma<24>, hipFuncAttributeMaxDynamicSharedMemorySize, W * 25 * 1024);
    snprintf(name, 128, "C private regions by LDS-DMA, 256 blocks x %d waves, 24 KiB in flight", W);
    RUN(name, hipLaunchKernelGGL(k_private_dma<24>, dim3(256), dim3(64 * W), W * 25 * 1024, 0, src, ppw, sink));
  }
  // D: 50 launches over the same 74 MiB
  {
    const size_t ppw = 148;
    const size_t small = (size_t)256 * 2 * ppw * 1024;
    for (int nt = 0; nt < 2; ++nt) {
      for (int r = 0; r < 3; ++r) { if (nt) hipLaunchKernelGGL((k_short_dma<8, true>), dim3(256), dim3(128), 2 * 9 * 1024, 0, src, ppw, sink); else hipLaunchKernelGGL((k_short_dma<8, false>), dim3(256), dim3(128), 2 * 9 * 1024, 0, src, ppw, sink); }
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 50; ++r) { if (nt) hipLaunchKernelGGL((k_short_dma<8, true>), dim3(256), dim3(128), 2 * 9 * 1024, 0, src, ppw, sink); else hipLaunchKernelGGL((k_short_dma<8, false>), dim3(256), dim3(128), 2 * 9 * 1024, 0, src, ppw, sink); }
      hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
      printf("D 256 blocks x 2 loader waves x 148 KiB, 8 in flight, %s, re-read by 50 launches: %.1f us per launch, %.0f GB/s\n", nt ? "nt" : "default policy",
             ms * 1e3 / 50, small / (ms * 1e-3 / 50) / 1e9);
    }
  }
  {
    const size_t ppw = 148;
    const size_t small = (size_t)256 * 2 * ppw * 1024;
    for (int fl = 1; fl < 4; ++fl) {
      auto go = [&]() {
        if (fl == 1) hipLaunchKernelGGL((k_short_dma_flags<8, 1>), dim3(256), dim3(128), 2 * 9 * 1024 + 64, 0, src, ppw, sink);
        if (fl == 2) hipLaunchKernelGGL((k_short_dma_flags<8, 2>), dim3(256), dim3(128), 2 * 9 * 1024 + 64, 0, src, ppw, sink);
        if (fl == 3) hipLaunchKernelGGL((k_short_dma_flags<8, 3>), dim3(256), dim3(128), 2 * 9 * 1024 + 64, 0, src, ppw, sink);
      };
      for (int r = 0; r < 3; ++r) go();
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 50; ++r) go();
      hipEventRecord(e1); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1);
      printf("E as D (nt) + %s%s per tile: %.1f us per launch, %.0f GB/s\n", (fl & 1) ? "[LDS flag store] " : "", (fl & 2) ? "[LDS flag load + readfirstlane]" : "",
             ms * 1e3 / 50, small / (ms * 1e-3 / 50) / 1e9);
    }
  }
  // B with more, smaller regions: 2048 blocks (8 per CU) x 1 wave
  {
    const size_t ppw = bytes / 1024 / 2048;
    RUN("B private regions, 2048 blocks x 1 wave, 8 KiB in flight per wave", hipLaunchKernelGGL(k_private<8>, dim3(2048), dim3(64), 0, 0, src, ppw, sink));
  }
  return 0;
}
