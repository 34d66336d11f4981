// LDS-DMA vs plain load+ds_write issue cost for one wave (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define TIME(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
__global__ void k(const uint32_t *src, unsigned long long *out, int iters, int mode) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x & 63;
  if (threadIdx.x >= 64) return;
  const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
  unsigned long long t0, t1;
  const uint32_t *g = src + (size_t)blockIdx.x * 65536;
  TIME(t0);
  if (mode == 0) {  // DMA16, 8 in flight, ring of 16 KB
    for (int i = 0; i < iters; ++i) {
      const uint32_t *p = g + (size_t)(i & 63) * 256 + lane * 4;
      uint32_t dst = base + (i & 15) * 1024;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(8)" ::"s"(dst), "v"(p) : "m0", "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (mode == 1) {  // DMA16 + DMA4 per iteration (a U=4 tile), 8 tiles in flight
    for (int i = 0; i < iters; ++i) {
      const uint32_t *p = g + (size_t)(i & 31) * 320;
      uint32_t dst = base + (i & 15) * 1280;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(p + lane) : "m0", "memory");
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_waitcnt vmcnt(16)" ::"s"(dst + 256), "v"(p + 64 + lane * 4) : "m0", "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else if (mode == 2) {  // plain global_load_dwordx4 -> ds_write_b128, 1 in flight (latency bound reference)
    for (int i = 0; i < iters; ++i) {
      const uint4 v = *reinterpret_cast<const uint4 *>(g + (size_t)(i & 63) * 256 + lane * 4);
      *reinterpret_cast<uint4 *>(lds + (i & 15) * 256 + lane * 4) = v;
    }
  } else {  // plain loads, 8 independent in flight per iteration then 8 ds_writes
    for (int i = 0; i < iters; i += 8) {
      uint4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const uint4 *>(g + (size_t)((i + j) & 63) * 256 + lane * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<uint4 *>(lds + ((i + j) & 15) * 256 + lane * 4) = v[j];
    }
  }
  TIME(t1);
  if (lane == 0) out[blockIdx.x] = t1 - t0;
  if (lds[lane] == 0x12345) out[1000] = 1;
}
int main() {
  uint32_t *src; unsigned long long *d, h[256];
  hipMalloc(&src, 256 * 65536 * 4); hipMemset(src, 1, 256 * 65536 * 4); hipMalloc(&d, 2048 * 8);
  const char *names[] = {"DMA16 x1 per iter, 8 in flight", "DMA4+DMA16 per iter (U=4 tile), 8 tiles in flight", "plain dwordx4 load + ds_write, serial", "plain loads 8 in flight + 8 ds_write_b128 per 8 iters"};
  for (int blocks : {1, 256})
    for (int mode = 0; mode < 4; ++mode) {
      for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 32768, 0, src, d, 2048, mode); hipDeviceSynchronize(); }
      hipMemcpy(h, d, 256 * 8, hipMemcpyDeviceToHost);
      double avg = 0; for (int i = 0; i < blocks; ++i) avg += (double)h[i]; avg /= blocks;
      printf("blocks %3d  %-55s %8.1f cycles per iteration\n", blocks, names[mode], avg / 2048.0);
    }
  return 0;
}
