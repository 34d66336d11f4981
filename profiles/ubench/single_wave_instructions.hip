// single-wave instruction timing microbenchmarks (diagnostic only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define TIME(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")

__global__ void k(unsigned long long *out, int iters, int extra_wait) {
  __shared__ float2 lds[4096];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = make_float2(1.0f, __int_as_float((i * 37 + 11) & 4095));
  __syncthreads();
  if (wave != 0) { if (extra_wait) __syncthreads(); return; }
  unsigned long long t0, t1;
  float a = lane, b = 1.0f, c = 2.0f, d = 3.0f, e = 4.0f;
  int ia = lane, ib = 3;
  uint64_t m0 = 0xaaaaaaaaaaaaaaaaull, sv;
  int n = 0;
  // 0: independent v_add_f32 x16
  TIME(t0);
  for (int i = 0; i < iters; ++i)
    asm volatile(REP4("v_add_f32 %0, %0, %4\n\tv_add_f32 %1, %1, %4\n\tv_add_f32 %2, %2, %4\n\tv_add_f32 %3, %3, %4\n\t") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  // 1: dependent v_add_f32 x16
  TIME(t0);
  for (int i = 0; i < iters; ++i) asm volatile(REP16("v_add_f32 %0, %0, %1\n\t") : "+v"(a) : "v"(e));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  // 2: dependent (v_max_i32_dpp + s_nop1 ... ) 8 pairs: dpp, cndmask, nop
  TIME(t0);
  for (int i = 0; i < iters; ++i)
    asm volatile(REP4("v_max_i32_dpp %1, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32 %0, %0, %1, %2\n\ts_nop 1\n\t"
                      "v_max_i32_dpp %1, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_cndmask_b32 %0, %0, %1, %2\n\ts_nop 1\n\t")
                 : "+v"(ia), "+v"(ib) : "s"(m0));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;   // 8 dpp + 8 cndmask + 8 nop
  // 3: exec-masked dpp: (s_mov exec, s_nop 0, dpp) x8
  TIME(t0);
  for (int i = 0; i < iters; ++i)
    asm volatile("s_mov_b64 %1, exec\n\t"
                 REP4("s_mov_b64 exec, %2\n\ts_nop 0\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                      "s_mov_b64 exec, %2\n\ts_nop 0\n\tv_max_i32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t")
                 "s_mov_b64 exec, %1\n\t" : "+v"(ia), "=&s"(sv) : "s"(m0));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;   // 8 x (2 salu + dpp)
  // 4: LDS pointer chase ds_read_b64, dependent, x16
  {
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + lane * 8;
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    int v;
    TIME(t0);
    for (int i = 0; i < iters; ++i)
      asm volatile(REP16("ds_read_b32 %0, %1 offset:4\n\ts_waitcnt lgkmcnt(0)\n\tv_lshl_add_u32 %1, %0, 3, %2\n\t") : "=&v"(v), "+v"(addr) : "v"(base));
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
    ia += v;
  }
  // 5: ds_write_b64 then ds_read_b64 same address, wait, x16
  {
    uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds + lane * 8;
    float2 v = make_float2(a, b);
    TIME(t0);
    for (int i = 0; i < iters; ++i)
      asm volatile(REP16("ds_write_b64 %1, %0\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\t") : "+v"(v) : "v"(addr));
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
    a += v.x;
  }
  // 6: 16 SALU s_add
  {
    int s = 1;
    TIME(t0);
    for (int i = 0; i < iters; ++i) asm volatile(REP16("s_add_u32 %0, %0, 3\n\t") : "+s"(s));
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
    ia += s;
  }
  // 7: taken branches x16
  TIME(t0);
  for (int i = 0; i < iters; ++i) asm volatile(REP16("s_branch 1f\n\ts_nop 0\n1:\n\t"));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  // 8: v_readfirstlane -> s_cmp -> s_cbranch (not taken) x16
  {
    int s;
    TIME(t0);
    for (int i = 0; i < iters; ++i)
      asm volatile(REP16("v_readfirstlane_b32 %0, %1\n\ts_cmp_eq_u32 %0, 12345\n\ts_cbranch_scc1 2f\n\t") "2:\n\t" : "=&s"(s) : "v"(ia) : "scc");
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  }
  // 9: v_cmp -> sgpr -> s_and_saveexec -> s_mov exec restore x16
  {
    uint64_t s, s2;
    TIME(t0);
    for (int i = 0; i < iters; ++i)
      asm volatile(REP16("v_cmp_gt_i32 %0, 10, %2\n\ts_and_saveexec_b64 %1, %0\n\ts_mov_b64 exec, %1\n\t") : "=&s"(s), "=&s"(s2) : "v"(ia) : "scc");
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  }
  // 10: 4 random ds_read_b64 gathers + wait, x4 (conflicts like the sweep)
  {
    uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds;
    uint32_t a0 = base + ((lane * 613 + 7) & 2047) * 8, a1 = base + ((lane * 1291 + 3) & 2047) * 8, a2 = base + ((lane * 97 + 1) & 2047) * 8, a3 = base + ((lane * 1777 + 5) & 2047) * 8;
    float2 v0, v1, v2, v3;
    TIME(t0);
    for (int i = 0; i < iters; ++i)
      asm volatile(REP4("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)\n\t")
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
    TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
    a += v0.x + v1.x + v2.x + v3.x;
  }
  // 11: v_ldexp / v_frexp dependent x16
  TIME(t0);
  for (int i = 0; i < iters; ++i) asm volatile(REP4("v_ldexp_f32 %0, %0, %1\n\tv_frexp_mant_f32 %0, %0\n\tv_sub_u32 %1, %1, %1\n\tv_max3_i32 %1, %1, %1, %1\n\t") : "+v"(a), "+v"(ib));
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  // 12: empty loop overhead
  TIME(t0);
  for (int i = 0; i < iters; ++i) asm volatile("");
  TIME(t1); if (lane == 0) out[n] = t1 - t0; ++n;
  if (a == 12345.0f && ia == 77) out[63] = 1;
  if (extra_wait) __syncthreads();
}

int main() {
  unsigned long long *d, h[64];
  hipMalloc(&d, 64 * 8);
  const char *names[] = {"16 indep v_add", "16 dep v_add", "8x(dpp,cndmask,nop1) dep", "8x(s_mov exec,s_nop0,dpp) dep", "16x(ds_read_b32 chase + valu)",
                         "4x(4 gathers b64 + wait)", "16 dep ldexp/frexp/sub/max3", "empty loop"};
  for (int nt : {64, 1024}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(d, 0, 64 * 8);
      hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, d, 1000, 1);
      hipDeviceSynchronize();
    }
    hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
    printf("block of %d threads (other waves wait at a barrier)\n", nt);
    for (int i = 0; i < 13; ++i) printf("  %-40s %8.1f cycles per iteration\n", names[i], (double)h[i] / 1000.0);
  }
  return 0;
}
