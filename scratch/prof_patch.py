"""Builds a diagnostic copy of the library with s_memtime stamps in ring_sweep
(never shipped; output only to a debug buffer)."""
import re, subprocess, sys, os
src = open('nfst_amd/csrc/kernels.hip').read()
# debug buffer + exported reader
src = src.replace('namespace {\n\nconstexpr int kEZero', '__device__ unsigned long long g_dbg[8192];\nextern "C" int nfst_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long)*8192); }\nnamespace {\n#define STAMP(slot) do { if (dbg_on && t >= 20 && t < 52) { unsigned long long c_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory"); if (lane == 0) g_dbg[dbg_base + (t-20)*8 + (slot)] = c_; } } while(0)\n\nconstexpr int kEZero',1)
# instrument ring_sweep
src = src.replace('  const uint32_t *ring = rg.lds;\n  const bool has_extra = ex.any();','  const uint32_t *ring = rg.lds;\n  const bool has_extra = ex.any();\n  const bool dbg_on = (blockIdx.x == 7);\n  const int dbg_base = (int)(threadIdx.x >> 6) * 512;')
src = src.replace('    if (t < my_steps) {\n      const int na = (int)na_u;','    if (t < my_steps) {\n      STAMP(0);\n      const int na = (int)na_u;')
src = src.replace('      const int spw = 64 >> kl;\n      const int k = 1 << kl;\n      for (int base = w * spw; base < ns; base += kSweepWaves * spw) {','      STAMP(1);\n      const int spw = 64 >> kl;\n      const int k = 1 << kl;\n      for (int base = w * spw; base < ns; base += kSweepWaves * spw) {')
src = src.replace('        group_reduce(M, E, kl);\n        if (i < ns && r == 0) {\n          if (accum) {','        STAMP(2);\n        group_reduce(M, E, kl);\n        STAMP(3);\n        if (i < ns && r == 0) {\n          if (accum) {')
src = src.replace('      off = next_off;\n      arc_base += na;','      STAMP(4);\n      off = next_off;\n      arc_base += na;')
src = src.replace('      if (t + 1 < my_steps) ring_wait(rg, off + kLookahead);\n    }\n    lds_barrier();','      if (t + 1 < my_steps) ring_wait(rg, off + kLookahead);\n      STAMP(5);\n    }\n    lds_barrier();\n    STAMP(6);')
open('scratch/kernels_prof.hip','w').write(src)
subprocess.check_call(['/opt/rocm/bin/hipcc','-O3','--offload-arch=gfx950','-fPIC','-shared','-pthread','-std=c++17','-Iinclude','nfst_amd/csrc/pack.cpp','scratch/kernels_prof.hip','-o','scratch/libnfst_prof.so'])
print("built")
