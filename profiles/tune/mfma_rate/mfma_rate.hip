// Float32 MFMA issue rate on one MI355X: 256 workgroups x 16 waves, each wave N MFMAs on 1 or 4 independent accumulators.
// hipcc -O3 --offload-arch=gfx950 -o mfma_rate mfma_rate.hip && ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ __launch_bounds__(1024) void k16(float *out, int n, float a, float b) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0.f, 0.f, 0.f, 0.f};
  float x = a + threadIdx.x * 1e-6f, y = b;
  for (int it = 0; it < n; it += NACC) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[i], 0, 0, 0);
  }
  f4 s = acc[0];
  for (int i = 1; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

template <int NACC>
__global__ __launch_bounds__(1024) void k32(float *out, int n, float a, float b) {
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float x = a + threadIdx.x * 1e-6f, y = b;
  for (int it = 0; it < n; it += NACC) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static void run(const char *name, K kern, double flop_per_mfma, int waves) {
  float *out;
  hipMalloc(&out, 256 * 1024 * sizeof(float));
  const int n = 1 << 14;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, out, n, 1.0f, 0.5f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(256), dim3(64 * waves), 0, 0, out, n, 1.0f, 0.5f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double mfmas = 256.0 * waves * n;
  printf("%-28s waves/CU %2d  %8.3f ms  %7.1f TFLOP/s  %6.1f ns per MFMA per SIMD\n", name, waves, ms,
         mfmas * flop_per_mfma / ms / 1e9, ms * 1e6 / (mfmas / (256.0 * 4)));
  hipFree(out);
}

int main() {
  run("16x16x4 f32, 1 accumulator", k16<1>, 2048, 16);
  run("16x16x4 f32, 4 accumulators", k16<4>, 2048, 16);
  run("16x16x4 f32, 4 acc, 4 waves", k16<4>, 2048, 4);
  run("32x32x2 f32, 1 accumulator", k32<1>, 4096, 16);
  run("32x32x2 f32, 2 accumulators", k32<2>, 4096, 16);
  return 0;
}
