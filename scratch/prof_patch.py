"""Diagnostic copy of the library with s_memtime stamps in ring_sweep (never shipped)."""
import subprocess
src = open('nfst_amd/csrc/kernels.hip').read()
src = src.replace('namespace {\n\nconstexpr int kEZero', '__device__ unsigned long long g_dbg[8192];\nextern "C" int nfst_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long)*8192); }\nnamespace {\n#define STAMP(slot) do { if (dbg_on && t >= 20 && t < 52) { unsigned long long c_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory"); if (lane == 0) g_dbg[dbg_base + (t-20)*8 + (slot)] = c_; } } while(0)\n\nconstexpr int kEZero',1)
def rep(a,b):
    global src
    assert a in src, a
    src = src.replace(a,b,1)
rep('  const uint32_t *ring = ring_lds;\n  const bool has_extra = ex.any();','  const uint32_t *ring = ring_lds;\n  const bool has_extra = ex.any();\n  const bool dbg_on = (blockIdx.x == 7);\n  const int dbg_base = (int)(threadIdx.x >> 6) * 512;')
rep('      ring_advance(rg, H0.off, lane);','      STAMP(0);\n      ring_advance(rg, H0.off, lane);')
rep('        // --- A: gathers of this tile','        STAMP(1);\n        // --- A: gathers of this tile')
rep('        // --- C: this tile\'s sum with one shared exponent','        STAMP(2);\n        // --- C: this tile\'s sum with one shared exponent')
rep('        // --- E: lanes with more than kUnroll arcs','        STAMP(3);\n        // --- E: lanes with more than kUnroll arcs')
rep('        // --- F: reduce over the state\'s lanes, normalise, store','        STAMP(4);\n        // --- F: reduce over the state\'s lanes, normalise, store')
rep('        cur = nxt;\n        have = hv;','        STAMP(5);\n        cur = nxt;\n        have = hv;')
rep('    if (W > 1) lds_barrier();\n  }\n  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring stays in flight','    if (W > 1) lds_barrier();\n    STAMP(6);\n  }\n  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring stays in flight')
open('scratch/kernels_prof.hip','w').write(src)
subprocess.check_call(['/opt/rocm/bin/hipcc','-O3','--offload-arch=gfx950','-fPIC','-shared','-pthread','-std=c++17','-Iinclude','nfst_amd/csrc/pack.cpp','scratch/kernels_prof.hip','-o','scratch/libnfst_prof.so'])
print("built")
