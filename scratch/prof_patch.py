"""Diagnostic copy of the library with s_memtime stamps in tile_sweep (never shipped).
usage: prof_patch.py POINT   (POINT in B C D N: where the second stamp goes; N = only the loop-top stamp)"""
import subprocess, sys
point = sys.argv[1]
src = open('nfst_amd/csrc/kernels.hip').read()
hdr = r'''__device__ unsigned long long g_dbg[8192];
extern "C" int nfst_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long)*8192); }
namespace {
#define TSTAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")

constexpr int kEZero'''
src = src.replace('namespace {\n\nconstexpr int kEZero', hdr, 1)
def rep(a, b):
    global src
    assert a in src, a
    src = src.replace(a, b, 1)
rep('  int landed = 0;  // wave-uniform copy of the decoder\'s counter',
    '  unsigned long long tA = 0, tX = 0, accX = 0, accI = 0, tPrev = 0, t_begin, polls = 0; TSTAMP(t_begin);\n  int landed = 0;  // wave-uniform copy of the decoder\'s counter')
rep('      landed = __builtin_amdgcn_readfirstlane(lds_flag_load(land));\n      if (landed < need) __builtin_amdgcn_s_sleep(1);\n    }\n    asm volatile("" ::: "memory");\n  };\n  const int last',
    '      landed = __builtin_amdgcn_readfirstlane(lds_flag_load(land));\n      ++polls;\n      if (landed < need) __builtin_amdgcn_s_sleep(1);\n    }\n    asm volatile("" ::: "memory");\n  };\n  const int last')
rep('    // --- operand gathers: the head of the dependency chain',
    '    TSTAMP(tA); if (T > 0) accI += tA - tPrev; tPrev = tA;')
pts = {
 'B': ('    // --- reduce over the state\'s lanes, normalise, store.  The tile\'s largest group', None),
 'C': ('    // tiles 0 .. T+1 are consumed: the words of tile T+1 were read above', None),
}
if point == 'B':
    rep("    const uint32_t w0 = cur.w0;\n    const int gl", "    asm volatile(\"\" :: \"v\"(M), \"v\"(E));\n    TSTAMP(tX); accX += tX - tA;\n    const uint32_t w0 = cur.w0;\n    const int gl")
elif point == 'G':  # right after the gathers have returned
    rep("    // --- this lane's partial sum with one shared exponent", "    asm volatile(\"\" :: \"v\"(vv[0]), \"v\"(vv[U-1]));\n    TSTAMP(tX); accX += tX - tA;")
elif point == 'D':
    rep("    if (publish) lds_flag_store(prog, T + 2);\n  };", "    TSTAMP(tX); accX += tX - tA;\n    if (publish) lds_flag_store(prog, T + 2);\n  };")
rep('    step(T + 1, db, da, true);\n  }\n', '    step(T + 1, db, da, true);\n  }\n  { unsigned long long t_end; TSTAMP(t_end); if (lane == 0 && blockIdx.x == 7) { unsigned long long *o = g_dbg + (threadIdx.x >> 6) * 16; o[0] = t_end - t_begin; o[1] = accI; o[2] = accX; o[3] = polls; o[4] = n_tiles; } }\n')
open('scratch/kernels_prof.hip', 'w').write(src)
subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-pthread', '-std=c++17', '-Wno-inline-asm',
                       '-Iinclude', 'nfst_amd/csrc/pack.cpp', 'scratch/kernels_prof.hip', '-o', 'scratch/libnfst_prof_%s.so' % point])
print("built", point)
