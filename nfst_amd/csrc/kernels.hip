// kernels.hip -- gfx950 (MI355X) kernels of the nFST lattice engine and their
// C-ABI launchers.  Design: DESIGN.md.  One workgroup owns one lattice; alpha and
// beta of all its states live in LDS as (mantissa, exponent) pairs -- an
// extended-exponent probability semiring: exact path sums like the reference's
// probability-domain beta sweep (/root/reference/src/modules/scorers.py:692-751)
// but without its float32 overflow and without exp/log on
// the level-to-level critical path.  Arc records stream once per sweep from HBM in
// level order; per-state sums are reduced by 2^k neighbouring lanes with wave64
// shuffles.  No MFMA: this is a sparse gather/reduce.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cmath>

#include "nfst_hip.h"

namespace {

constexpr int kEZero = -(1 << 28);      // exponent of an exact zero
constexpr float kNegInf = -__builtin_huge_valf();

struct ME {
  float m;
  int e;
};

// exp(x) = m * 2^e with m in [0.70, 1.42]; x = -inf (or below -9e7) gives zero.
__device__ __forceinline__ ME exp_split(float x) {
  ME r;
  // weights below e^-9e7 count as zero and scores above 9e7 are clamped: exponents then
  // stay far from the int32 range when they are added up along a path
  if (!(x > -9.0e7f)) { r.m = 0.0f; r.e = kEZero; return r; }
  x = fminf(x, 9.0e7f);
  float kf = rintf(x * 1.44269504088896341f);
  float t = fmaf(-kf, 0.693145751953125f, x);         // ln2 high part (exact product)
  t = fmaf(-kf, 1.42860682030941723e-6f, t);          // ln2 low part
  // exp(t), |t| <= 0.3466: degree-7 Taylor, relative error < 1e-8
  float p = 1.0f / 5040.0f;
  p = fmaf(p, t, 1.0f / 720.0f);
  p = fmaf(p, t, 1.0f / 120.0f);
  p = fmaf(p, t, 1.0f / 24.0f);
  p = fmaf(p, t, 1.0f / 6.0f);
  p = fmaf(p, t, 0.5f);
  p = fmaf(p, t, 1.0f);
  p = fmaf(p, t, 1.0f);
  r.m = p;
  r.e = (int)kf;
  return r;
}

// normalise a sum to mantissa in [0.5, 1).  A zero sum keeps mantissa 0 (frexp(0) = 0,
// exponent 0): its exponent stays near kEZero, which never wins a max against a real
// term, so no select is needed.
__device__ __forceinline__ float2 me_pack(float M, int E) {
  int ex;
  float mant = frexpf(M, &ex);
  return make_float2(mant, __int_as_float(max(E + ex, kEZero)));  // saturates at 2^(-2^28): no wrap-around
}

// natural log of an (m, e) pair in float64 / float32
__device__ __forceinline__ double me_log64(float2 v) {
  if (!(v.x > 0.0f)) return -__builtin_huge_val();
  return log((double)v.x) + (double)__float_as_int(v.y) * 0.693147180559945309417232;
}
__device__ __forceinline__ float me_log32(float2 v) {
  if (!(v.x > 0.0f)) return kNegInf;
  return (float)((double)logf(v.x) + (double)__float_as_int(v.y) * 0.693147180559945309417232);
}

struct Meta {
  int row_off, n_rows, arc_off, n_arcs, fwd_off, fwd_tiles, bwd_off, bwd_tiles, sink, n_reach, depth, n_dp,
      fwd_u, bwd_u, fwd_wide, bwd_wide, fwd_slot_off, bwd_slot_off;
};
__device__ __forceinline__ Meta load_meta(const int32_t *meta, int b) {
  const int32_t *m = meta + (size_t)b * NFST_META_WORDS;
  Meta r;
  r.row_off = m[NFST_META_ROW_OFF]; r.n_rows = m[NFST_META_N_ROWS];
  r.arc_off = m[NFST_META_ARC_OFF]; r.n_arcs = m[NFST_META_N_ARCS];
  r.fwd_off = m[NFST_META_FWD_OFF]; r.fwd_tiles = m[NFST_META_FWD_TILES];
  r.bwd_off = m[NFST_META_BWD_OFF]; r.bwd_tiles = m[NFST_META_BWD_TILES];
  r.sink = m[NFST_META_SINK]; r.n_reach = m[NFST_META_N_REACH]; r.depth = m[NFST_META_DEPTH];
  r.n_dp = m[NFST_META_N_DP];
  // program format code (1, 2, 4: slots per lane; 8: compact tiles), and bit 8: the program has
  // tiles with groups wider than 8 lanes
  r.fwd_u = m[NFST_META_FWD_U] & 0xff; r.bwd_u = m[NFST_META_BWD_U] & 0xff;
  r.fwd_wide = (m[NFST_META_FWD_U] >> 8) & 1; r.bwd_wide = (m[NFST_META_BWD_U] >> 8) & 1;
  r.fwd_slot_off = m[NFST_META_FWD_SLOT_OFF]; r.bwd_slot_off = m[NFST_META_BWD_SLOT_OFF];
  return r;
}

// Extra per-arc log weight (weighted tables and/or caller-supplied arc scores),
// addressed by canonical arc id.
struct Extra {
  const float *arc_w;
  const float *arc_scores;
  __device__ __forceinline__ bool any() const { return arc_w != nullptr || arc_scores != nullptr; }
  __device__ __forceinline__ float at(int a) const {
    float x = 0.0f;
    if (arc_w) x += arc_w[a];
    if (arc_scores) x += arc_scores[a];
    return x;
  }
};

// ---------------------------------------------------------------- tile programs
// A sweep is a "tile program" laid out by the host packer (pack.cpp, DESIGN.md
// section 3): a sequence of fixed-size tiles, each one wave-wide unit of work --
// 64 control words and 64*U arc records (U = 1, 2 or 4 slots per lane).  ONE wave
// runs one sweep: its LDS accesses are ordered, a tile only reads states that an
// earlier tile wrote, so a sweep needs no barrier at all, and the alpha and beta
// sweeps of a lattice run as two independent waves of the workgroup.
//
// control word: [0:16) 8 x state id (the byte offset of its value in the alpha / beta array)
//               [20:23) g: the state's lanes are the 2^g-aligned group of 2^g lanes
//               [23:26) largest g in this tile (same in every lane)
//               [26] the tile holds a continuation piece (same in every lane)
//               [30] continuation piece (its first record is the carry)  [31] leader lane
//               (stores the state's sum)
// record:       [0:16) 8 x operand state | [16:32) label (vocab = the null label: weight 0,
//               vocab + 1 = the unit label of a carry record: weight 1)
//
// The program does not depend on DP values, so two helper waves of the workgroup run far
// ahead of the sweep.  The LOADER copies tiles from HBM into a small staging ring in LDS
// with global_load_lds (LDS-DMA, no VGPR staging; kDmaAhead tiles in flight, counted
// s_waitcnt vmcnt) -- issuing an LDS-DMA costs the issuing wave 60-100 cycles, which is
// why this is a wave of its own.  The DECODER turns every record into what the sweep
// needs -- the LDS
// address of the operand and the (mantissa, exponent) weight of the arc, label weight x
// per-arc extra -- and writes the decoded tile into a ring of R slots in LDS.  The sweep
// wave reads only decoded tiles: nothing but the dependency chain is left on it.
//
// decoded tile, 64 * (1 + 3U) words:
//   [0, 64)            word 0 per lane: the control word + the LDS address of alpha / beta:
//                      [0:20) LDS byte address of the state's value, the rest as above
//   [64, 64 + 64U)     U operand LDS byte addresses per lane
//   then               (m, e) weights, slots (2k, 2k+1) of all lanes in block k (16 B per lane)
// tiles the loader keeps in flight (HBM -> LDS by LDS-DMA) and raw-tile staging slots per
// sweep: deep when a workgroup has a CU's LDS to itself, shallow when two share it
constexpr int kDmaAheadDeep = 8, kRawSlotsDeep = 12, kDmaAheadShared = 4, kRawSlotsShared = kDmaAheadShared + 1;
constexpr int kRawWords = 64 * (1 + 4);         // raw tile for U = 4: 1280 B
constexpr int kRawWordsX = kRawWords + 64 * 4;  // + the slots' canonical arc ids (kernels with per-arc extras)
constexpr int kSlotWords = 64 * (1 + 3 * 4);    // decoded tile for U = 4: 3328 B
// program format code (meta word, bits 0..7): 1, 2, 4 = slots per lane with 32-bit records and a
// separate control block; 8 = the compact tile: four slots per lane, 16 bytes per lane = control
// word + four 24-bit records (state 13 bits | label 11 bits)
__host__ __device__ constexpr int fmt_u(int F) { return F == 8 ? 4 : F; }
__host__ __device__ constexpr int fmt_words(int F) { return F == 8 ? 256 : 64 * (1 + F); }
constexpr int kMaxRing = 12, kMinRing = 3;       // ring slots per sweep (chosen at launch from the LDS budget)

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// Segmented all-reduce of (M, E) partial sums: a lane whose state owns 2^g lanes takes
// part in stages 0 .. g-1.  Stage partners: lane^1, lane^2 (quad permutes), 7-lane and
// 15-lane mirrors inside a row (DPP modifiers, no LDS traffic), then lane^16 and
// lane^32 (shuffles).  GMAX (the tile's largest g) bounds the stages executed.  All
// lanes of a state end with bitwise the same (M, E): max of exponents, one rescale,
// then the sum.
// Segmented all-reduce of (M, E) partial sums: a lane whose state owns 2^g lanes takes
// part in stages 0 .. g-1.  Stage partners: lane^1, lane^2 (quad permutes), 7-lane and
// 15-lane mirrors inside a row (DPP modifiers, no LDS traffic), then lane^16 and
// lane^32 (shuffles).  Stages 0..2 always run (one predicated select each, no
// branch); stages 3..5 only when the tile's largest g needs them.  All lanes of a state
// end with bitwise the same (M, E): max of exponents, one rescale, then the sum.
template <int STAGES>
__device__ __forceinline__ int seg_max(int Em, int g) {
  if (STAGES >= 1) { const int o = dpp_i<0xB1>(Em); Em = (g >= 1) ? max(Em, o) : Em; }
  if (STAGES >= 2) { const int o = dpp_i<0x4E>(Em); Em = (g >= 2) ? max(Em, o) : Em; }
  if (STAGES >= 3) { const int o = dpp_i<0x141>(Em); Em = (g >= 3) ? max(Em, o) : Em; }
  if (STAGES >= 4) { const int o = dpp_i<0x140>(Em); Em = (g >= 4) ? max(Em, o) : Em; }
  if (STAGES >= 5) { const int o = __shfl_xor(Em, 16); Em = (g >= 5) ? max(Em, o) : Em; }
  if (STAGES >= 6) { const int o = __shfl_xor(Em, 32); Em = (g >= 6) ? max(Em, o) : Em; }
  return Em;
}
template <int STAGES>
__device__ __forceinline__ float seg_sum(float M, int g) {
  if (STAGES >= 1) { const float o = dpp_f<0xB1>(M); M = (g >= 1) ? M + o : M; }
  if (STAGES >= 2) { const float o = dpp_f<0x4E>(M); M = (g >= 2) ? M + o : M; }
  if (STAGES >= 3) { const float o = dpp_f<0x141>(M); M = (g >= 3) ? M + o : M; }
  if (STAGES >= 4) { const float o = dpp_f<0x140>(M); M = (g >= 4) ? M + o : M; }
  if (STAGES >= 5) { const float o = __shfl_xor(M, 16); M = (g >= 5) ? M + o : M; }
  if (STAGES >= 6) { const float o = __shfl_xor(M, 32); M = (g >= 6) ? M + o : M; }
  return M;
}
// The same reduction for groups of up to 8 lanes with the per-lane select replaced by the
// execution mask: m[s] = lanes whose state owns more than 2^s lanes (wave masks, computed
// off the dependency chain); a DPP instruction executed under m[s] updates exactly the
// lanes that take part in stage s and leaves the others as they are, so a stage is ONE
// vector instruction.  The scalar moves in between also provide the two wait states a
// DPP read needs after a vector write.  Returns the group's exponent in E, the sum in M.
template <int STAGES>
__device__ __forceinline__ void seg_reduce_exec(float &M, int &E, uint64_t m0, uint64_t m1, uint64_t m2) {
  static_assert(STAGES == 2 || STAGES == 3, "");
  const int e0 = E;
  int d;
  uint64_t sv;
  if (STAGES == 2) {
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[e], %[e], %[e] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[e], %[e], %[e] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_sub_u32 %[d], %[e0], %[e]\n\t"
        "v_ldexp_f32 %[m], %[m], %[d]\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[m], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m] "+v"(M), [e] "+&v"(E), [d] "=&v"(d), [sv] "=&s"(sv)
        : [m0] "s"(m0), [m1] "s"(m1), [e0] "v"(e0));
  } else {
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[e], %[e], %[e] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[e], %[e], %[e] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "s_nop 0\n\t"
        "v_max_i32_dpp %[e], %[e], %[e] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]\n\t"
        "v_sub_u32 %[d], %[e0], %[e]\n\t"
        "v_ldexp_f32 %[m], %[m], %[d]\n\t"
        "s_mov_b64 exec, %[m0]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[m], %[m], %[m] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m1]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[m], %[m], %[m] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[m2]\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %[m], %[m], %[m] row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_mov_b64 exec, %[sv]"
        : [m] "+v"(M), [e] "+&v"(E), [d] "=&v"(d), [sv] "=&s"(sv)
        : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [e0] "v"(e0));
  }
}

template <int STAGES>
__device__ __forceinline__ void seg_reduce_n(float &M, int &E, int g) {
  const int Em = seg_max<STAGES>(E, g);
  M = seg_sum<STAGES>(ldexpf(M, E - Em), g);
  E = Em;
}

__device__ __forceinline__ int lds_flag_load(const int *p) {
  return __atomic_load_n(p, __ATOMIC_RELAXED);
}
__device__ __forceinline__ void lds_flag_store(int *p, int v) {
  __atomic_store_n(p, v, __ATOMIC_RELAXED);
}

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef __attribute__((address_space(3))) v4f lds_v4f;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) v2u lds_v2u;
typedef __attribute__((address_space(3))) v4u lds_v4u;

__device__ __forceinline__ uint32_t lds_addr(const void *p) {
  return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// ---- producer / consumer protocol -------------------------------------------------
// Two LDS words per sweep, both only grow:
//   land: tiles 0 .. land-1 are decoded and in the ring (written by the decoder)
//   prog: tiles 0 .. prog-1 are consumed, their slots are free (written by the sweep)
// LDS accesses of one wave execute in order and LDS is coherent within the CU, so
// "write slot -> store land" / "load land -> read slot" need no barrier.

// LDS-DMA (global_load_lds_*): lane i's `bytes` go to LDS address m0 + i*bytes.  Issued
// from inline asm on purpose: the compiler then keeps no record of a pending LDS-DMA and
// does not put s_waitcnt vmcnt(0) in front of every LDS access; the counted waits are
// placed by hand (vm_wait).  In-flight data never lives in registers, so no compiler-made
// register copy can touch it early.  Only full-wave 4- and 16-byte forms are used (the
// 12-byte and exec-masked forms do not lay lanes out at lane x size on gfx950).  `nt`: a tile
// program is read once per launch by one CU (measured: 1-2 % on the whole step).
__device__ __forceinline__ void lds_dma16(const void *gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(lds_dst), "v"(gsrc) : "m0", "memory");
}
__device__ __forceinline__ void lds_dma4(const void *gsrc, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off nt" ::"s"(lds_dst), "v"(gsrc) : "m0", "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {  // at most N vector-memory operations of this wave stay in flight
  static_assert(N >= 0 && N <= 63, "vmcnt is 6 bits");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// LDS-DMA instructions per tile
template <int F, bool EXTRA>
struct DmaOps {
  static constexpr int value = F == 8 ? (EXTRA ? 2 : 1) : (F == 2 ? 3 : 2) + (EXTRA ? (F == 2 ? 2 : 1) : 0);
};

// raw staging slot: [64 control words][64 U records]([64 U canonical arc ids]); compact tiles:
// [64 x (control word, 3 record words)]([256 canonical arc ids])
template <int F, bool EXTRA>
__device__ __forceinline__ void tile_issue(const uint32_t *g, const int32_t *perm, int tile, uint32_t slot_addr, int lane) {
  constexpr int U = fmt_u(F);
  const uint32_t *src = g + (size_t)tile * fmt_words(F);
  const int32_t *q = perm + (size_t)tile * (64 * U);
  if (F == 8) {
    lds_dma16(src + lane * 4, slot_addr);
    if (EXTRA) lds_dma16(q + lane * 4, slot_addr + 1024);
    return;
  }
  lds_dma4(src + lane, slot_addr);
  if (U == 4) {
    lds_dma16(src + 64 + lane * 4, slot_addr + 256);
    if (EXTRA) lds_dma16(q + lane * 4, slot_addr + 256 + 1024);
  } else if (U == 2) {
    lds_dma4(src + 64 + lane, slot_addr + 256);
    lds_dma4(src + 128 + lane, slot_addr + 512);
    if (EXTRA) { lds_dma4(q + lane, slot_addr + 768); lds_dma4(q + 64 + lane, slot_addr + 1024); }
  } else {
    lds_dma4(src + 64 + lane, slot_addr + 256);
    if (EXTRA) lds_dma4(q + lane, slot_addr + 512);
  }
}

// ---- loader wave ------------------------------------------------------------------
// flags (LDS words, all only grow): rland = raw tiles 0 .. rland-1 have landed in the
// staging ring; the decoder's `land` (tiles decoded) tells which staging slots are free.
// part 1 (kernel entry, before anything else): the first ring-full needs no hand-shake,
// so it is in flight while the workgroup initialises
template <bool EXTRA>
__device__ __forceinline__ void loader_start(int U, const uint32_t *g, const int32_t *perm, int n_tiles, uint32_t *raw,
                                             int RS, int lane) {
  constexpr uint32_t RB = (EXTRA ? kRawWordsX : kRawWords) * 4;
  const uint32_t raw_base = lds_addr(raw);
  const int n = min(n_tiles, RS);
  for (int d = 0; d < n; ++d) {
    if (U == 8) tile_issue<8, EXTRA>(g, perm, d, raw_base + d * RB, lane);
    else if (U == 4) tile_issue<4, EXTRA>(g, perm, d, raw_base + d * RB, lane);
    else if (U == 2) tile_issue<2, EXTRA>(g, perm, d, raw_base + d * RB, lane);
    else tile_issue<1, EXTRA>(g, perm, d, raw_base + d * RB, lane);
  }
}

// blocks until at most `tiles` tiles (OPS LDS-DMA instructions each) are in flight;
// vmcnt takes an immediate, hence the chain
template <int OPS, int MAXT>
__device__ __forceinline__ void wait_tiles_in_flight(int tiles) {
  if (MAXT > 0 && tiles >= MAXT) vm_wait<(OPS * MAXT > 63 ? 63 : OPS * MAXT)>();
  else if (MAXT > 0) wait_tiles_in_flight<OPS, (MAXT > 0 ? MAXT - 1 : 0)>(tiles);
  else vm_wait<0>();
}

// part 2: streams the rest of the tile program into the staging ring.  Copies complete
// in order, so "at most k tiles in flight" means tiles 0 .. issued-k-1 have landed: after
// an issue the loader waits with k = AHEAD; whenever it cannot issue (ring full, or the
// whole program issued) it publishes the oldest unpublished tile with the exact count.
// `land` is the decoder's progress: when it publishes tile t the raw words of tiles 0 .. t+1
// are in its registers, so the staging slot of tile i is certainly free once land >= i + 1.
template <int F, bool EXTRA, int AHEAD>
__device__ __forceinline__ void tile_loader(const uint32_t *g, const int32_t *perm, int n_tiles, uint32_t *raw, int RS,
                                            const int *land, int *rland, int lane) {
  constexpr int OPS = DmaOps<F, EXTRA>::value;
  static_assert(OPS * AHEAD <= 63, "vmcnt is 6 bits");
  constexpr uint32_t RB = (EXTRA ? kRawWordsX : kRawWords) * 4;
  const uint32_t raw_base = lds_addr(raw), raw_end = raw_base + RS * RB;
  int issued = min(n_tiles, RS);  // loader_start issued these
  uint32_t rb = raw_base;         // slot of tile `issued` (the ring has wrapped once)
  int freed = 0;                  // copy of the decoder's counter
  int pub = 0;                    // tiles published in rland
  // nothing to issue right now: wait until half of the unpublished tiles have landed and
  // publish those (then half of the rest, ...)
  auto publish_some = [&]() {
    const int keep = (issued - pub - 1) >> 1;  // tiles that may stay in flight
    wait_tiles_in_flight<OPS, AHEAD>(keep);
    pub = issued - keep;
    lds_flag_store(rland, pub);
  };
  while (issued < n_tiles) {
    if (__builtin_expect(issued - freed >= RS, 0)) {  // ring full: look at the decoder's progress
      freed = __builtin_amdgcn_readfirstlane(lds_flag_load(land));
      if (issued - freed >= RS) {
        if (pub < issued) publish_some();
        else __builtin_amdgcn_s_sleep(1);
      }
      continue;
    }
    tile_issue<F, EXTRA>(g, perm, issued, rb, lane);
    ++issued;
    rb = (rb + RB == raw_end) ? raw_base : rb + RB;
    if (issued - pub > AHEAD) {
      vm_wait<OPS * AHEAD>();
      pub = issued - AHEAD;
      lds_flag_store(rland, pub);
    }
  }
  while (pub < n_tiles) publish_some();
}

// ---- decoder wave -----------------------------------------------------------------
// flags: land = tiles 0 .. land-1 are decoded and in the ring (written here),
// prog = tiles 0 .. prog-1 are consumed by the sweep, their ring slots are free.
// LDS accesses of one wave execute in order and LDS is coherent within the CU, so
// "write slot -> store land" / "load land -> read slot" need no barrier.
// one tile in the decoder's registers: control word, byte offset of every operand's value,
// 8 x label of every record, canonical arcs (only with per-arc extras)
template <int U, bool EXTRA>
struct RawRegs {
  uint32_t ctl;
  uint32_t opoff[U];
  uint32_t lab8[U];
  int32_t pm[EXTRA ? U : 1];
};
template <int F, bool EXTRA>
__device__ __forceinline__ void raw_fetch(uint32_t rb, int lane, RawRegs<fmt_u(F), EXTRA> &w) {
  constexpr int U = fmt_u(F);
  uint32_t rc[U];
  if (F == 8) {
    const v4u x = *(const lds_v4u *)(uintptr_t)(rb + lane * 16);
    w.ctl = x.x;
    const uint32_t r0 = x.y, r1 = __builtin_amdgcn_alignbit(x.z, x.y, 24), r2 = __builtin_amdgcn_alignbit(x.w, x.z, 16),
                   r3 = x.w >> 8;
    const uint32_t r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int j = 0; j < U; ++j) {
      w.opoff[j] = (r[j % 4] << 3) & 0xfff8u;   // state (13 bits) x 8
      w.lab8[j] = (r[j % 4] >> 10) & 0x3ff8u;   // label (11 bits) x 8
    }
    if (EXTRA) {
      const v4u a = *(const lds_v4u *)(uintptr_t)(rb + 1024 + lane * 16);
      w.pm[0] = (int)a.x; w.pm[EXTRA ? 1 % U : 0] = (int)a.y; w.pm[EXTRA ? 2 % U : 0] = (int)a.z; w.pm[EXTRA ? 3 % U : 0] = (int)a.w;
    }
    return;
  }
  w.ctl = *(const lds_u32 *)(uintptr_t)(rb + lane * 4);
  if (U == 4) {
    const v4u v = *(const lds_v4u *)(uintptr_t)(rb + 256 + lane * 16);
    rc[0] = v.x; rc[1 % U] = v.y; rc[2 % U] = v.z; rc[3 % U] = v.w;
    if (EXTRA) {
      const v4u a = *(const lds_v4u *)(uintptr_t)(rb + 256 + 1024 + lane * 16);
      w.pm[0] = (int)a.x; w.pm[EXTRA ? 1 % U : 0] = (int)a.y; w.pm[EXTRA ? 2 % U : 0] = (int)a.z; w.pm[EXTRA ? 3 % U : 0] = (int)a.w;
    }
  } else if (U == 2) {
    const v2u v = *(const lds_v2u *)(uintptr_t)(rb + 256 + lane * 8);
    rc[0] = v.x; rc[1 % U] = v.y;
    if (EXTRA) {
      const v2u a = *(const lds_v2u *)(uintptr_t)(rb + 768 + lane * 8);
      w.pm[0] = (int)a.x; w.pm[EXTRA ? 1 % U : 0] = (int)a.y;
    }
  } else {
    rc[0] = *(const lds_u32 *)(uintptr_t)(rb + 256 + lane * 4);
    if (EXTRA) w.pm[0] = (int)*(const lds_u32 *)(uintptr_t)(rb + 512 + lane * 4);
  }
#pragma unroll
  for (int j = 0; j < U; ++j) {
    w.opoff[j] = rc[j] & 0xffffu;
    w.lab8[j] = (rc[j] >> 16) << 3;
  }
}

// SELF: the decoder also does the loader's job (kernels with two workgroups per CU run
// fewer, busier waves): it keeps AHEAD tiles in flight itself -- self_start() at kernel
// entry, one issue per iteration -- and a counted wait replaces the rland flag.  The
// staging ring then has AHEAD + 1 slots.
template <int F, bool EXTRA, int AHEAD>
__device__ __forceinline__ void self_start_u(const uint32_t *g, const int32_t *perm, int n_tiles, uint32_t *raw, int lane) {
  constexpr uint32_t RB = (EXTRA ? kRawWordsX : kRawWords) * 4;
  const uint32_t raw_base = lds_addr(raw);
  const int last = max(n_tiles - 1, 0);
#pragma unroll
  for (int d = 0; d < AHEAD; ++d)  // short programs copy their last tile again: the count stays constant
    tile_issue<F, EXTRA>(g, perm, min(d, last), raw_base + d * RB, lane);
}
template <bool EXTRA, int AHEAD>
__device__ __forceinline__ void self_start(int U, const uint32_t *g, const int32_t *perm, int n_tiles, uint32_t *raw,
                                           int lane) {
  if (U == 8) self_start_u<8, EXTRA, AHEAD>(g, perm, n_tiles, raw, lane);
  else if (U == 4) self_start_u<4, EXTRA, AHEAD>(g, perm, n_tiles, raw, lane);
  else if (U == 2) self_start_u<2, EXTRA, AHEAD>(g, perm, n_tiles, raw, lane);
  else self_start_u<1, EXTRA, AHEAD>(g, perm, n_tiles, raw, lane);
}

template <int F, bool EXTRA, bool SELF, int AHEAD>
__device__ __forceinline__ void tile_decoder(int n_tiles, const uint32_t *raw, int RS, const int *rland,
                                             const uint32_t *g, const int32_t *perm,
                                             uint32_t *ring, int R, const int *prog, int *land, const float2 *val,
                                             const float2 *th_, const Extra ex, int lane) {
  if (n_tiles <= 0) {
    if (SELF) vm_wait<0>();
    return;
  }
  constexpr int U = fmt_u(F);
  constexpr int OPS = DmaOps<F, EXTRA>::value;
  constexpr uint32_t RB = (EXTRA ? kRawWordsX : kRawWords) * 4;
  constexpr uint32_t SB = 64 * (1 + 3 * U) * 4;
  const uint32_t th_base = lds_addr(th_);
  const uint32_t val_base = lds_addr(val);
  const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
  const uint32_t raw_base = lds_addr(raw), raw_end = raw_base + RS * RB;
  uint32_t sb = ring_base;  // decoded slot of tile t
  uint32_t rb = raw_base;   // staging slot of the tile whose raw words are fetched next
  uint32_t rb_issue = raw_base + (SELF ? AHEAD * RB : 0);  // SELF: staging slot of the tile issued next
  int issue_next = AHEAD;                                   // SELF: that tile
  const int last = n_tiles - 1;
  int freed = 0, landed = 0;
  int prog_peek = 0;  // SELF: the sweep's counter as of the previous iteration (per-lane copy)
  // makes sure the raw words of tile need-1 are in the staging ring (called once per tile, in order)
  auto wait_raw = [&](int need) {
    if (SELF) {
      // one more tile goes in flight (past the end the last tile is copied again into a slot
      // nobody reads, so that the count stays exact); then at most AHEAD are
      tile_issue<F, EXTRA>(g, perm, min(issue_next, last), rb_issue, lane);
      ++issue_next;
      rb_issue = (rb_issue + RB == raw_end) ? raw_base : rb_issue + RB;
      vm_wait<OPS * AHEAD>();
      return;
    }
    while (__builtin_expect(landed < need, 0)) {
      landed = __builtin_amdgcn_readfirstlane(lds_flag_load(rland));
      if (landed < need) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  auto gather_weights = [&](const RawRegs<U, EXTRA> &w, v2f (&tw)[U]) {
#pragma unroll
    for (int j = 0; j < U; ++j) tw[j] = *(const lds_v2f *)(uintptr_t)(th_base + w.lab8[j]);
  };
  // iteration t: `cur` = raw words of tile t, `tw` = its label weights (LDS gathers issued
  // one iteration earlier); fetches the raw words of tile t+1 into `nxt` and, at the end,
  // issues the gathers of its label weights into `twn`
  auto step = [&](int t, const RawRegs<U, EXTRA> &cur, v2f (&tw)[U], RawRegs<U, EXTRA> &nxt, v2f (&twn)[U]) {
    // past the end this reads a stale staging slot whose contents are never used
    rb = (rb + RB == raw_end) ? raw_base : rb + RB;
    // the self-loading decoder reads the sweep's counter (LDS) one iteration ahead: in the common
    // case the check of the ring slot costs no LDS round trip (measured: +3 % arcs/s with two
    // workgroups per CU; with separate loader waves the extra read costs 1 %, so not there)
    if (SELF) freed = max(freed, __builtin_amdgcn_readfirstlane(prog_peek));
    wait_raw(min(t + 2, n_tiles));
    raw_fetch<F, EXTRA>(rb, lane, nxt);
    if (SELF) prog_peek = lds_flag_load(prog);
    asm volatile("" ::: "memory");
    // --- control word and operand addresses: the packer's byte offsets + the array's base
    const uint32_t w0 = cur.ctl + val_base;
    uint32_t oa[U];
#pragma unroll
    for (int j = 0; j < U; ++j) oa[j] = cur.opoff[j] + val_base;
    // --- the ring slot must be free: tile t - R consumed
    while (__builtin_expect(t - freed >= R, 0)) {
      freed = __builtin_amdgcn_readfirstlane(lds_flag_load(prog));
      if (t - freed >= R) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
    *(lds_u32 *)(uintptr_t)(sb + lane * 4) = w0;
    if (U == 4) *(lds_v4u *)(uintptr_t)(sb + 256 + lane * 16) = v4u{oa[0], oa[1 % U], oa[2 % U], oa[3 % U]};
    else if (U == 2) *(lds_v2u *)(uintptr_t)(sb + 256 + lane * 8) = v2u{oa[0], oa[1 % U]};
    else *(lds_u32 *)(uintptr_t)(sb + 256 + lane * 4) = oa[0];
    if (EXTRA) {
      // per-arc extras: plain loads, waited for in place (lattices with per-arc extras
      // decode at about one memory latency per tile)
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (cur.pm[j] >= 0) {
          const ME x = exp_split(ex.at(cur.pm[j]));
          tw[j].x *= x.m;
          tw[j].y = __int_as_float(__float_as_int(tw[j].y) + x.e);
        }
      }
    }
    if (U == 4) {
      *(lds_v4f *)(uintptr_t)(sb + 256 + 1024 + lane * 16) = v4f{tw[0].x, tw[0].y, tw[1 % U].x, tw[1 % U].y};
      *(lds_v4f *)(uintptr_t)(sb + 256 + 2048 + lane * 16) = v4f{tw[2 % U].x, tw[2 % U].y, tw[3 % U].x, tw[3 % U].y};
    } else if (U == 2) {
      *(lds_v4f *)(uintptr_t)(sb + 256 + 512 + lane * 16) = v4f{tw[0].x, tw[0].y, tw[1 % U].x, tw[1 % U].y};
    } else {
      *(lds_v2f *)(uintptr_t)(sb + 256 + 256 + lane * 8) = tw[0];
    }
    asm volatile("" ::: "memory");
    // tile t is decoded; the loader reads the same word: the raw words of tiles 0 .. t+1
    // are in registers
    lds_flag_store(land, t + 1);
    sb = (sb + SB == ring_end) ? ring_base : sb + SB;
    gather_weights(nxt, twn);
  };
  RawRegs<U, EXTRA> ra, rbb;
  v2f ta[U], tb[U];
  wait_raw(1);
  raw_fetch<F, EXTRA>(rb, lane, ra);
  gather_weights(ra, ta);
  // two iterations per trip so that the register roles alternate without copies
  for (int t = 0; t < n_tiles; t += 2) {
    step(t, ra, ta, rbb, tb);
    if (t + 1 >= n_tiles) break;
    step(t + 1, rbb, tb, ra, ta);
  }
  if (SELF) vm_wait<0>();  // nothing of the staging ring stays in flight
}

// decoded tile in the sweep wave's registers
template <int U>
struct TileDec {
  uint32_t w0;
  uint32_t opa[U];
  v2f tw[U];
};

template <int U>
__device__ __forceinline__ void dec_fetch(uint32_t sb, int lane, TileDec<U> &d) {
  d.w0 = *(const lds_u32 *)(uintptr_t)(sb + lane * 4);
  if (U == 4) {
    const v4u a = *(const lds_v4u *)(uintptr_t)(sb + 256 + lane * 16);
    const v4f p = *(const lds_v4f *)(uintptr_t)(sb + 256 + 1024 + lane * 16);
    const v4f q = *(const lds_v4f *)(uintptr_t)(sb + 256 + 2048 + lane * 16);
    d.opa[0] = a.x; d.opa[1 % U] = a.y; d.opa[2 % U] = a.z; d.opa[3 % U] = a.w;
    d.tw[0] = v2f{p.x, p.y}; d.tw[1 % U] = v2f{p.z, p.w}; d.tw[2 % U] = v2f{q.x, q.y}; d.tw[3 % U] = v2f{q.z, q.w};
  } else if (U == 2) {
    const v2u a = *(const lds_v2u *)(uintptr_t)(sb + 256 + lane * 8);
    const v4f p = *(const lds_v4f *)(uintptr_t)(sb + 256 + 512 + lane * 16);
    d.opa[0] = a.x; d.opa[1 % U] = a.y;
    d.tw[0] = v2f{p.x, p.y}; d.tw[1 % U] = v2f{p.z, p.w};
  } else {
    d.opa[0] = *(const lds_u32 *)(uintptr_t)(sb + 256 + lane * 4);
    d.tw[0] = *(const lds_v2f *)(uintptr_t)(sb + 256 + 256 + lane * 8);
  }
}

// One sum-product sweep, run by ONE wave over the decoded ring.  The sweep is one
// dependency chain (gather operands -> sum -> reduce over the state's lanes -> store ->
// next tile's gathers) and a single wave issues one instruction every ~4 cycles, a taken
// branch costs ~20 and a scalar use of a fresh vector result ~25: an iteration is
// straight-line code.  It starts with the operand gathers of its tile, fetches the next
// decoded tile and prepares the stage masks in their shadow, and ends by moving the next
// tile's wave-uniform flags to a scalar register.
template <int U, bool WIDE>
__device__ __forceinline__ void tile_sweep(int n_tiles, const uint32_t *ring, int R, int *prog, const int *land,
                                           int lane) {
  if (n_tiles <= 0) return;
  constexpr uint32_t SB = 64 * (1 + 3 * U) * 4;  // bytes per ring slot
  const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
  int landed = 0;  // wave-uniform copy of the decoder's counter, refreshed only when it runs out
  auto wait_landed = [&](int need) {
    while (__builtin_expect(landed < need, 0)) {
      landed = __builtin_amdgcn_readfirstlane(lds_flag_load(land));
      if (landed < need) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  uint32_t sb = ring_base;  // slot of the tile that is fetched next
  int land_peek = 0;        // the decoder's counter as of the previous iteration (per-lane copy of the LDS word)
  // iteration T: `cur` = tile T with its uniform flags in `cu`; `nxt` receives tile T+1
  auto step = [&](int T, const TileDec<U> &cur, uint32_t cu, TileDec<U> &nxt, uint32_t &cu_nxt) {
    // --- operand gathers: the head of the dependency chain
    v2f vv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) vv[j] = *(const lds_v2f *)(uintptr_t)cur.opa[j];
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);  // nothing is scheduled in front of the gathers
    // --- the next decoded tile.  Past the end of the program this reads a stale slot
    // whose contents are never used.
    sb = (sb + SB == ring_end) ? ring_base : sb + SB;
    // the decoder's counter was read (LDS) during the previous iteration: in the common case the
    // check costs no LDS round trip
    landed = max(landed, __builtin_amdgcn_readfirstlane(land_peek));
    wait_landed(min(T + 2, n_tiles));
    dec_fetch<U>(sb, lane, nxt);
    land_peek = lds_flag_load(land);
    asm volatile("" ::: "memory");
    // --- what only needs the tile's control word: stage masks (lanes whose state owns
    // more than 2^s lanes), leader lanes, store address
    const uint32_t w0 = cur.w0;
    const int gl = (int)((w0 >> 20) & 7u);
    lds_v2f *dst = (lds_v2f *)(uintptr_t)(w0 & 0xfffffu);
    const bool leader = (int)w0 < 0;
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(gl > 0), m1 = __builtin_amdgcn_ballot_w64(gl > 1),
                   m2 = __builtin_amdgcn_ballot_w64(gl > 2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    // --- this lane's partial sum with one shared exponent
    float mt[U];
    int et[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      mt[j] = cur.tw[j].x * vv[j].x;
      et[j] = __float_as_int(cur.tw[j].y) + __float_as_int(vv[j].y);
    }
    int E = et[0];
#pragma unroll
    for (int j = 1; j < U; ++j) E = max(E, et[j]);
    float M = ldexpf(mt[0], et[0] - E);
#pragma unroll
    for (int j = 1; j < U; ++j) M += ldexpf(mt[j], et[j] - E);
    // --- reduce over the state's lanes (max of exponents, one rescale, sum), normalise,
    // store.  Groups of up to 8 lanes run three stages under execution masks (a stage
    // nobody takes part in is an empty mask).  Only programs the packer marked WIDE have
    // tiles with larger groups (flagged wave-uniformly); those take the general path.
    if (WIDE && __builtin_expect((cu & (1u << 25)) != 0, 0)) {
      seg_reduce_n<6>(M, E, gl);
      if (leader) {
        const float2 r = me_pack(M, E);
        *dst = v2f{r.x, r.y};
      }
    } else {
      seg_reduce_exec<3>(M, E, m0, m1, m2);
      if (leader) {
        const float2 r = me_pack(M, E);
        *dst = v2f{r.x, r.y};
      }
    }
    // tiles 0 .. T+1 are consumed: the words of tile T+1 were read above
    lds_flag_store(prog, T + 2);  // every tile: the decoder's hand-shake latency matters more than the store
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (WIDE) cu_nxt = (uint32_t)__builtin_amdgcn_readfirstlane(nxt.w0);
  };
  wait_landed(1);
  TileDec<U> da, db;
  dec_fetch<U>(sb, lane, da);
  asm volatile("" ::: "memory");
  uint32_t ca = WIDE ? (uint32_t)__builtin_amdgcn_readfirstlane(da.w0) : 0u, cb = 0;
  // two iterations per trip so that the register roles alternate without copies
  for (int T = 0; T < n_tiles; T += 2) {
    step(T, da, ca, db, cb);
    if (T + 1 >= n_tiles) break;
    step(T + 1, db, cb, da, ca);
  }
}

// role dispatch: role 0 sweeps, role 1 decodes for it, role 2 loads for the decoder.
// flags: [0] prog [1] land [2] rland
template <bool EXTRA, bool SELF, int AHEAD>
__device__ __forceinline__ void run_sweep(int role, int U, bool wide, uint32_t *raw, int RS, const uint32_t *g,
                                          const int32_t *perm, int n_tiles, uint32_t *ring, int R, int *flags,
                                          float2 *val, const float2 *th, const Extra ex, int lane) {
  int *prog = flags, *land = flags + 1, *rland = flags + 2;
  if (role == 0) {
    if (U == 8) U = 4;  // the sweep only sees decoded tiles
    if (wide) {
      if (U == 4) tile_sweep<4, true>(n_tiles, ring, R, prog, land, lane);
      else if (U == 2) tile_sweep<2, true>(n_tiles, ring, R, prog, land, lane);
      else tile_sweep<1, true>(n_tiles, ring, R, prog, land, lane);
    } else {
      if (U == 4) tile_sweep<4, false>(n_tiles, ring, R, prog, land, lane);
      else if (U == 2) tile_sweep<2, false>(n_tiles, ring, R, prog, land, lane);
      else tile_sweep<1, false>(n_tiles, ring, R, prog, land, lane);
    }
  } else if (role == 1) {
    if (U == 8) tile_decoder<8, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, perm, ring, R, prog, land, val, th, ex, lane);
    else if (U == 4) tile_decoder<4, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, perm, ring, R, prog, land, val, th, ex, lane);
    else if (U == 2) tile_decoder<2, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, perm, ring, R, prog, land, val, th, ex, lane);
    else tile_decoder<1, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, perm, ring, R, prog, land, val, th, ex, lane);
  } else if (!SELF) {
    if (U == 8) tile_loader<8, EXTRA, AHEAD>(g, perm, n_tiles, raw, RS, land, rland, lane);
    else if (U == 4) tile_loader<4, EXTRA, AHEAD>(g, perm, n_tiles, raw, RS, land, rland, lane);
    else if (U == 2) tile_loader<2, EXTRA, AHEAD>(g, perm, n_tiles, raw, RS, land, rland, lane);
    else tile_loader<1, EXTRA, AHEAD>(g, perm, n_tiles, raw, RS, land, rland, lane);
  }
}

__device__ __forceinline__ void load_theta(float2 *th, const float *theta, int64_t stride, int b,
                                           int V, int tid, int nt) {
  const float *t = theta + (size_t)stride * b;
  for (int l = tid; l < V; l += nt) {
    ME x = exp_split(t[l]);
    th[l] = make_float2(x.m, __int_as_float(x.e));
  }
  if (tid == 0) {
    th[V] = make_float2(0.0f, __int_as_float(kEZero));  // the null label of empty slots
    th[V + 1] = make_float2(0.5f, __int_as_float(1));       // weight one: the carry record of a continuation piece
  }
}

// ------------------------------------------------------------------ LDS layout
// [alpha: rows2 float2][beta: rows2 float2][theta: v2 float2 (V labels + null + unit)]
// [label histogram: v4 float][per sweep: R decoded tiles, kRawSlots raw tiles][4 flag words per sweep]  (16-B aligned)
struct LdsPlan {
  int rows2, v2, v4;
  __host__ __device__ LdsPlan(int max_rows, int vocab)
      : rows2((max_rows + 1) & ~1), v2((vocab + 3) & ~1), v4((vocab + 3) & ~3) {}
  // words of one sweep's rings
  static __host__ __device__ int64_t sweep_words(int R, int RS, bool extra) {
    return (int64_t)R * kSlotWords + (int64_t)RS * (extra ? kRawWordsX : kRawWords);
  }
  __host__ __device__ int64_t fb_bytes(int R, int RS, bool extra) const {
    return ((int64_t)2 * rows2 + v2) * 8 + (int64_t)v4 * 4 + 2 * sweep_words(R, RS, extra) * 4 + 32;
  }
  __host__ __device__ int64_t bwd_bytes(int R, int RS, bool extra) const {
    return ((int64_t)rows2 + v2) * 8 + sweep_words(R, RS, extra) * 4 + 16;
  }
};

// Block size: wave 0 runs the beta sweep, wave 1 the alpha sweep; every wave helps with
// the initialisation, the row outputs and the posterior pass.  With at most one lattice
// per CU those phases are latency-bound and get 16 waves; with several lattices per CU the
// co-resident workgroups hide each other's latencies and 4 waves are cheaper.

// ------------------------------------------------------------------ backward only
// Wave 0 sweeps the by-source program from the sink, wave 1 decodes for it, wave 2 loads
// for the decoder; every wave helps with the initialisation and the outputs.
template <int NT, bool EXTRA>
__global__ __launch_bounds__(NT) void k_backward(nfst_batch lat, nfst_scores sc, int R, int RS, float *logbeta,
                                                 double *logz64, float *logz32, float2 *beta_me) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  float2 *beta = lds;
  float2 *th = lds + plan.rows2;
  uint32_t *ring = (uint32_t *)(th + plan.v2);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t *raw = ring + (size_t)R * kSlotWords;
  // NT = 512: the workgroup has the CU to itself: wave 2 loads for the decoder (deep staging
  // ring); NT = 256: two workgroups per CU, the decoder loads for itself
  constexpr bool kSelf = NT != 512;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  if (kSelf) {
    if (wv == 1) self_start<EXTRA, kAhead>(m.bwd_u, lat.bwd_stream + m.bwd_off, lat.bwd_perm + m.bwd_slot_off, m.bwd_tiles, raw, lane);
  } else if (wv == 2) {
    loader_start<EXTRA>(m.bwd_u, lat.bwd_stream + m.bwd_off, lat.bwd_perm + m.bwd_slot_off, m.bwd_tiles, raw, RS, lane);
  }
  for (int i = tid; i < m.n_rows; i += NT) beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT);
  __syncthreads();
  int *flags = (int *)(ring + LdsPlan::sweep_words(R, RS, EXTRA));
  if (tid == 0) {
    beta[m.sink] = make_float2(0.5f, __int_as_float(1));
    flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;
  }
  __syncthreads();
  if (wv < (kSelf ? 2 : 3))
    run_sweep<EXTRA, kSelf, kAhead>(wv, m.bwd_u, m.bwd_wide != 0, raw, RS, lat.bwd_stream + m.bwd_off, lat.bwd_perm + m.bwd_slot_off,
              m.bwd_tiles, ring, R, flags, beta, th, ex, lane);
  __syncthreads();
  if (tid == 0) {
    const double z = me_log64(beta[0]);
    if (logz64) logz64[b] = z;
    if (logz32) logz32[b] = (float)z;
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
    if (beta_me) beta_me[m.row_off + i] = beta[i];
  }
}

// ------------------------------------------------------------------ forward-backward
__device__ __forceinline__ float arc_posterior(const float2 av, const float2 bv, const float2 tw, float rz,
                                               int ez, bool has_extra, const Extra &ex, int a) {
  float mw = tw.x;
  int ew = __float_as_int(tw.y);
  if (has_extra) {
    ME x = exp_split(ex.at(a));
    mw *= x.m;
    ew += x.e;
  }
  const float mm = (av.x * mw) * (bv.x * rz);
  const int ee = __float_as_int(av.y) + ew + __float_as_int(bv.y) - ez;
  return ldexpf(mm, max(ee, -300));
}

// Wave 0 runs the beta sweep and wave 1 the alpha sweep, concurrently and without any
// synchronisation between them, fed by waves 2 and 3; after the one barrier that
// follows every wave of the block streams canonical arcs for the posteriors.
template <int NT, bool EXTRA>
__global__ __launch_bounds__(NT) void k_forward_backward(
    nfst_batch lat, nfst_scores sc, int R, int RS, float *__restrict__ logalpha, float *__restrict__ logbeta,
    double *__restrict__ logz64, float *__restrict__ logz32, float *__restrict__ posterior,
    float *__restrict__ grad_theta, float2 *__restrict__ beta_me) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  float2 *alpha = lds;
  float2 *beta = lds + plan.rows2;
  float2 *th = lds + 2 * plan.rows2;
  float *gth = (float *)(th + plan.v2);  // [V] label histogram (only if grad_theta)
  uint32_t *ring = (uint32_t *)(gth + plan.v4);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  constexpr bool has_extra = EXTRA;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // even waves work for the beta sweep, odd waves for alpha: waves 0 / 1 sweep, 2 / 3 decode,
  // 6 / 7 load (waves i, i+4, ... share a SIMD: the busy-polling loaders sit with the decoders,
  // the sweep waves share theirs only with waves that sleep at the barrier)
  const bool bwd_side = (wv & 1) == 0;
  const uint32_t *my_prog = bwd_side ? lat.bwd_stream + m.bwd_off : lat.fwd_stream + m.fwd_off;
  const int32_t *my_perm = bwd_side ? lat.bwd_perm + m.bwd_slot_off : lat.fwd_perm + m.fwd_slot_off;
  const int my_tiles = bwd_side ? m.bwd_tiles : m.fwd_tiles;
  const int my_u = bwd_side ? m.bwd_u : m.fwd_u;
  const bool my_wide = (bwd_side ? m.bwd_wide : m.fwd_wide) != 0;
  uint32_t *my_ring = bwd_side ? ring : ring + LdsPlan::sweep_words(R, RS, EXTRA);
  uint32_t *my_raw = my_ring + (size_t)R * kSlotWords;
  // NT = 1024: the workgroup has the CU to itself: waves 4 / 5 load for the decoders (deep
  // staging ring); otherwise two workgroups share a CU and the decoders load for themselves
  constexpr bool kSelf = NT != 1024;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  if (kSelf) {
    if (wv == 2 || wv == 3) self_start<EXTRA, kAhead>(my_u, my_prog, my_perm, my_tiles, my_raw, lane);
  } else if (wv == 6 || wv == 7) {
    loader_start<EXTRA>(my_u, my_prog, my_perm, my_tiles, my_raw, RS, lane);
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    alpha[i] = make_float2(0.0f, __int_as_float(kEZero));
    beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  }
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT);
  if (grad_theta) for (int l = tid; l < lat.vocab; l += NT) gth[l] = 0.0f;
  __syncthreads();
  int *flags = (int *)(ring + 2 * LdsPlan::sweep_words(R, RS, EXTRA));
  if (tid == 0) {
    beta[m.sink] = make_float2(0.5f, __int_as_float(1));
    alpha[0] = make_float2(0.5f, __int_as_float(1));
    for (int i = 0; i < 8; ++i) flags[i] = 0;
  }
  __syncthreads();
  const bool want_post = posterior != nullptr || grad_theta != nullptr;
  const int a_begin = m.arc_off, a_end = m.arc_off + m.n_arcs;
  // the posterior pass works on groups of 4 arcs (16-byte loads / stores) over the
  // aligned interior [v_begin, v_end) of the lattice's canonical arc range
  const int v_begin = (a_begin + 3) & ~3, v_end = a_end & ~3;
  // The waves beyond the first four have nothing to do during the sweeps: they fetch
  // their first kPre arc groups into registers now, so that after the sweeps the
  // posterior pass starts on data that is already there.
  constexpr int kSweepThreads = 256;
  constexpr int kHelpers = NT - kSweepThreads;
  constexpr int kPre = (kHelpers > 0) ? 7 : 0;  // 7 x 768 x 4 = 21.5k arcs: a whole BASELINE lattice
  // src | dst << 16 and the label of 4 consecutive canonical arcs: 16 + 8 bytes
  uint4 psd[kPre > 0 ? kPre : 1];
  uint2 plb[kPre > 0 ? kPre : 1];
  if (kPre > 0 && tid >= kSweepThreads && want_post) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
      if (a < v_end) {
        psd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        plb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
      }
    }
  }
  // waves 0 / 1 run the beta / alpha sweeps, waves 2 / 3 decode and waves 6 / 7 load for them
  if (wv < 4 || (!kSelf && (wv == 6 || wv == 7)))
    run_sweep<EXTRA, kSelf, kAhead>(wv < 4 ? wv >> 1 : 2, my_u, my_wide, my_raw, RS, my_prog, my_perm, my_tiles, my_ring, R,
                     bwd_side ? flags : flags + 4, bwd_side ? beta : alpha, th, ex, lane);
  __syncthreads();
  const float2 zme = beta[0];
  if (tid == 0) {
    const double z = me_log64(zme);
    if (logz64) logz64[b] = z;
    if (logz32) logz32[b] = (float)z;
  }
  const float rz = (zme.x > 0.0f) ? 1.0f / zme.x : 0.0f;
  const int ez = __float_as_int(zme.y);
  auto do_group = [&](const uint4 sd, const uint2 lb, int a) {
    const uint32_t sdv[4] = {sd.x, sd.y, sd.z, sd.w};
    const int ll[4] = {(int)(lb.x & 0xffffu), (int)(lb.x >> 16), (int)(lb.y & 0xffffu), (int)(lb.y >> 16)};
    float pp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s0 = (int)(sdv[q] & 0xffffu), d0 = (int)(sdv[q] >> 16);
      pp[q] = (s0 != d0) ? arc_posterior(alpha[s0], beta[d0], th[ll[q]], rz, ez, has_extra, ex, a + q) : 0.0f;
      if (grad_theta && pp[q] > 0.0f) atomicAdd(&gth[ll[q]], pp[q]);
    }
    if (posterior) *reinterpret_cast<float4 *>(posterior + a) = make_float4(pp[0], pp[1], pp[2], pp[3]);
  };
  if (kPre > 0 && tid >= kSweepThreads) {
    if (want_post) {
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
        if (a < v_end) do_group(psd[u], plb[u], a);
      }
    }
  } else {
    // the sweep waves (all waves when there are no helpers) write the row outputs
    constexpr int RT = (kPre > 0) ? kSweepThreads : NT;
    for (int i = tid; i < m.n_rows; i += RT) {
      if (logalpha) logalpha[m.row_off + i] = me_log32(alpha[i]);
      if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
      if (beta_me) beta_me[m.row_off + i] = beta[i];
    }
  }
  if (want_post) {
    // the arc groups that were not preloaded: kPB groups per iteration, all loads issued
    // before the first use
    constexpr int kPB = 4;
    for (int a0 = v_begin + 4 * (kPre * kHelpers + tid); a0 < v_end; a0 += NT * 4 * kPB) {
      uint4 sd[kPB];
      uint2 lb[kPB];
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = min(a0 + u * NT * 4, v_end - 4);  // clamped: always a valid group
        sd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        lb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
      }
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = a0 + u * NT * 4;
        if (a >= v_end) break;
        do_group(sd[u], lb[u], a);
      }
    }
    // unaligned head and tail (at most 3 arcs each)
    const int n_head = min(v_begin, a_end) - a_begin;
    const int n_tail = (v_end >= v_begin) ? a_end - v_end : 0;
    if (tid < n_head + n_tail) {
      const int a = tid < n_head ? a_begin + tid : v_end + (tid - n_head);
      const int s0 = lat.arc_src[a], d0 = lat.arc_dst[a], l0 = lat.arc_label[a];
      const float p = (s0 != d0) ? arc_posterior(alpha[s0], beta[d0], th[l0], rz, ez, has_extra, ex, a) : 0.0f;
      if (posterior) posterior[a] = p;
      if (grad_theta && p > 0.0f) atomicAdd(&gth[l0], p);
    }
    if (grad_theta) {
      __syncthreads();
      float *gout = grad_theta + (size_t)b * lat.vocab;
      for (int l = tid; l < lat.vocab; l += NT) gout[l] = gth[l];
    }
  }
}

// ------------------------------------------------------------------ Viterbi
// max-plus run of the by-source tile program by one wave (float32 values, back pointers
// -- canonical arc and next state -- in LDS; the program is read straight from global
// memory), then lane 0 walks the best path inside LDS.  Ties keep the arc with the smallest
// canonical id, i.e. the smallest label.
__device__ __forceinline__ void vit_take(float &bv, int &ba, int &bn, float ov, int oa, int on) {
  if (ov > bv || (ov == bv && oa < ba)) { bv = ov; ba = oa; bn = on; }  // bn: the arc's other end
}

// Wave 0 runs the program; waves 1 .. 3 run ahead of it and pull the tiles it will read
// (program words, slot -> arc map, per-arc extras) into the L2 cache, throttled by wave 0's
// progress counter in LDS, so that its dependent loads are L2 hits instead of HBM misses.
constexpr int kVitThreads = 256, kVitAhead = 12;
__global__ __launch_bounds__(kVitThreads) void k_viterbi(nfst_batch lat, nfst_scores sc, float *best,
                                                         int32_t *paths, int32_t *path_arcs,
                                                         int32_t *lengths, int max_len, int pad) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Meta m = load_meta(lat.meta, b);
  float *v = (float *)lds;
  int *bp = (int *)(v + lat.max_rows);       // back pointer: best arc out of the state
  int *ns = bp + lat.max_rows;               // ... and the state it leads to
  float *tl = (float *)(ns + lat.max_rows);  // [V] label scores
  int *progress = (int *)(tl + lat.vocab);
  const float *tg = sc.theta + (size_t)sc.theta_stride * b;
  for (int i = tid; i < lat.max_rows; i += kVitThreads) { v[i] = kNegInf; bp[i] = -1; }  // incl. scratch rows
  for (int i = tid; i < lat.vocab; i += kVitThreads) tl[i] = tg[i];
  if (tid == 0) *progress = 0;
  __syncthreads();
  if (tid == 0) v[m.sink] = 0.0f;
  __syncthreads();
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int F = m.bwd_u, U = fmt_u(F), ST = fmt_words(F);  // program format, slots per lane, words per tile
  const uint32_t *prog = lat.bwd_stream + m.bwd_off;
  const int32_t *perm = lat.bwd_perm + m.bwd_slot_off;
  constexpr int kNone = 0x7fffffff;
  if (wv > 0) {
    const bool extras = arc_w != nullptr || sc.arc_scores != nullptr;
    float sink_f = 0.0f;
    int sink_i = 0;
    for (int T = wv - 1; T < m.bwd_tiles; T += kVitThreads / 64 - 1) {
      while (T > lds_flag_load(progress) + kVitAhead) __builtin_amdgcn_s_sleep(8);
      // one 128-byte line per lane
      const int prog_lines = (ST * 4 + 127) / 128, perm_lines = (64 * U * 4 + 127) / 128;
      if (lane < prog_lines) sink_i += (int)prog[(size_t)T * ST + min(lane * 32, ST - 1)];
      if (!extras) {
        if (lane < perm_lines) sink_i += perm[(size_t)T * 64 * U + min(lane * 32, 64 * U - 1)];
      } else {
        for (int j = 0; j < U; ++j) {
          const int ca = perm[(size_t)T * 64 * U + lane * U + j];
          if (ca >= 0) {
            if (arc_w) sink_f += arc_w[ca];
            if (sc.arc_scores) sink_f += sc.arc_scores[ca];
          }
        }
      }
    }
    if (sink_f == 1.2345e-33f && sink_i == 0x12345678) best[b] = 0.0f;  // keeps the loads alive, never true
  } else {
  // the words of tile T+1 are loaded while tile T is computed
  struct VitTile { uint32_t ctl; int cas[4]; uint32_t rcs[4]; };
  auto load_tile = [&](int T, VitTile &t) {
    if (F == 8) {  // compact tile: control word + four 24-bit records per lane
      const uint4 x = *reinterpret_cast<const uint4 *>(prog + (size_t)T * ST + lane * 4);
      t.ctl = x.x;
      const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t.cas[j] = perm[(size_t)T * 256 + lane * 4 + j];
        t.rcs[j] = ((r[j] & 0x1fffu) << 3) | (((r[j] >> 13) & 0x7ffu) << 16);  // as a 32-bit record
      }
      return;
    }
    t.ctl = prog[(size_t)T * ST + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = min(j, U - 1);
      t.cas[j] = perm[(size_t)T * 64 * U + lane * U + jj];
      t.rcs[j] = prog[(size_t)T * ST + 64 + lane * U + jj];
    }
  };
  VitTile cur, nxt;
  if (m.bwd_tiles > 0) load_tile(0, cur);
  for (int T = 0; T < m.bwd_tiles; ++T) {
    load_tile(min(T + 1, m.bwd_tiles - 1), nxt);
    const uint32_t ctl = cur.ctl;
    int cas[4];
    uint32_t rcs[4];
    float xs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { cas[j] = (j < U) ? cur.cas[j] : -1; rcs[j] = cur.rcs[j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xs[j] = 0.0f;
      if (cas[j] >= 0) {
        if (arc_w) xs[j] += arc_w[cas[j]];
        if (sc.arc_scores) xs[j] += sc.arc_scores[cas[j]];
      }
    }
    float bv = kNegInf;
    int ba = kNone, bn = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int other = (int)((rcs[j] & 0xffffu) >> 3);
      if (cas[j] >= 0) vit_take(bv, ba, bn, tl[rcs[j] >> 16] + xs[j] + v[other], cas[j], other);
      // a unit-label record stands for what row `other` holds -- the state's own earlier pieces
      // (carry) or a scratch row of a partial group: its best arc competes as such
      else if (j < U && (int)(rcs[j] >> 16) == lat.vocab + 1 && bp[other] >= 0) vit_take(bv, ba, bn, v[other], bp[other], ns[other]);
    }
    const int gl = (int)((ctl >> 20) & 7u);
    const int gmax = (int)((__builtin_amdgcn_readfirstlane(ctl) >> 23) & 7u);
    // segmented max over the state's lanes: quad permutes and row mirrors (DPP), then the
    // two cross-row stages; every lane of a state ends with the same (value, arc, next state)
#define NFST_VIT_STAGE(ST, FV, FI)                                          \
    if (gmax > ST) {                                                        \
      const float ov = FV(bv);                                              \
      const int oa = FI(ba), on = FI(bn);                                   \
      if (gl > ST) vit_take(bv, ba, bn, ov, oa, on);                        \
    }
#define NFST_SHFL16(x) __shfl_xor(x, 16)
#define NFST_SHFL32(x) __shfl_xor(x, 32)
    NFST_VIT_STAGE(0, dpp_f<0xB1>, dpp_i<0xB1>)
    NFST_VIT_STAGE(1, dpp_f<0x4E>, dpp_i<0x4E>)
    NFST_VIT_STAGE(2, dpp_f<0x141>, dpp_i<0x141>)
    NFST_VIT_STAGE(3, dpp_f<0x140>, dpp_i<0x140>)
    NFST_VIT_STAGE(4, NFST_SHFL16, NFST_SHFL16)
    NFST_VIT_STAGE(5, NFST_SHFL32, NFST_SHFL32)
#undef NFST_VIT_STAGE
#undef NFST_SHFL16
#undef NFST_SHFL32
    if (ctl & (1u << 31)) {
      const uint32_t sid = (ctl & 0xffffu) >> 3;
      v[sid] = bv;
      bp[sid] = (ba == kNone) ? -1 : ba;
      ns[sid] = bn;
    }
    // LDS accesses of one wave execute in order: the next tile's loads see these stores
    asm volatile("" ::: "memory");
    if ((T & 3) == 3) lds_flag_store(progress, T);
    cur = nxt;
  }
  }
  // lane 0 walks the back pointers inside LDS (arc ids go to the list `pa`, which reuses the
  // value array); all threads then write the labels
  int *pa = (int *)v;
  int *res = progress;  // [0] length, [1] reached the sink
  if (tid == 0) {
    best[b] = v[0];
    int s0 = 0, len = 0;
    const int cap = min(max_len, m.n_rows);
    while (s0 != m.sink && len < cap) {
      const int a = bp[s0];
      if (a < 0) break;
      const int nx = ns[s0];
      pa[len++] = a;  // v[len-1] is dead: only v[0] was needed, and it has been read
      s0 = nx;
    }
    res[0] = len;
    res[1] = (s0 == m.sink) ? 1 : 0;
  }
  __syncthreads();
  const int len = res[0];
  if (tid == 0) lengths[b] = res[1] ? len : -1;
  for (int j = tid; j < max_len; j += kVitThreads) {
    const int a = j < len ? pa[j] : -1;
    paths[(size_t)b * max_len + j] = a >= 0 ? lat.arc_label[a] : pad;
    if (path_arcs) path_arcs[(size_t)b * max_len + j] = a;
  }
}

// ------------------------------------------------------------------ sampling
__device__ __forceinline__ uint32_t mulhilo(uint32_t a, uint32_t b, uint32_t *hi) {
  const uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  return (uint32_t)p;
}
// Philox4x32-10, counter (walk, step, 0, 0), key from seed; first output word -> [0,1)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t walk, uint32_t step) {
  uint32_t c0 = walk, c1 = step, c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0, hi1;
    const uint32_t lo0 = mulhilo(0xD2511F53u, c0, &hi0);
    const uint32_t lo1 = mulhilo(0xCD9E8D57u, c2, &hi1);
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float arc_score(const float *theta, const float *arc_w,
                                           const float *arc_scores, int l, int a) {
  float s = theta[l];
  if (arc_w) s += arc_w[a];
  if (arc_scores) s += arc_scores[a];
  return s;
}

// One walk per 16-lane row (a DPP "row"): the arcs of the current state are spread over the
// row's lanes, 16 at a time; probabilities p = w * beta[dst] / beta[state] come from the
// lattice's beta values staged in LDS, the CDF is an inclusive scan inside the row
// (row_shr 1, 2, 4, 8 with zero fill) and the first lane with u < cdf wins (arcs with p = 0
// never do).  A block is 16 walks of one lattice.
constexpr int kSampleThreads = 256, kWalksPerBlock = kSampleThreads / 16;
template <int SHIFT>
__device__ __forceinline__ float row_shr_zero(float v) {  // lane i gets lane i-SHIFT of its row, 0 if there is none
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + SHIFT, 0xf, 0xf, true));
}
__global__ __launch_bounds__(kSampleThreads) void k_sample(nfst_batch lat, nfst_scores sc,
                                                           const float2 *beta_me, const double *logz64, int K,
                                                           int max_len, const float *uniforms, uint64_t seed,
                                                           int pad, int stage_theta, int32_t *paths, int32_t *path_arcs,
                                                           int32_t *lengths, float *logq, int32_t *status) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r = tid & 15;                                // lane within the row
  const int k = blockIdx.y * kWalksPerBlock + (tid >> 4);  // this row's walk
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  float2 *bl = lds;                                      // beta (m, e) of the lattice's rows
  float *tls = (float *)(bl + lat.max_rows);             // label scores (staged unless the vocabulary is huge)
  for (int i = tid; i < m.n_rows; i += kSampleThreads) bl[i] = beta_me[m.row_off + i];
  if (stage_theta) for (int i = tid; i < lat.vocab; i += kSampleThreads) tls[i] = theta[i];
  __syncthreads();
  const float *tl = stage_theta ? (const float *)tls : theta;
  const bool live = k < K;
  const size_t walk = (size_t)b * K + (live ? k : 0);
  int32_t *out = paths + walk * max_len;
  int32_t *outa = path_arcs ? path_arcs + walk * max_len : nullptr;
  int s = 0, t = 0;
  float tot = 0.0f;
  bool ok = true;
  bool active = live;
  while (true) {
    if (active && s == m.sink) active = false;
    if (active && t >= max_len) { ok = false; active = false; }
    if (!__any(active)) break;
    float u = 0.0f;
    float2 bs = make_float2(1.0f, 0.0f);
    int a0 = 0, a1 = 0;
    if (active) {
      u = uniforms ? uniforms[walk * max_len + t] : philox_uniform(seed, (uint32_t)walk, (uint32_t)t);
      bs = bl[s];
      a0 = rp[s];
      a1 = rp[s + 1];
    }
    const float rs = 1.0f / bs.x;
    const int es = __float_as_int(bs.y);
    float cum_base = 0.0f, sc_ch = 0.0f, sc_last = 0.0f;
    int chosen = -1, last = -1, d_ch = 0, d_last = 0;
    bool more = active;
    for (int c = a0; __any(more); c += 16) {
      more = more && c < a1 && chosen < 0;
      const int a = c + r;
      float p = 0.0f, x = 0.0f;
      int d = 0;
      if (more && a < a1) {
        const uint32_t sd = lat.arc_sd[a];
        d = (int)(sd >> 16);
        if (d != s) {
          x = tl[lat.arc_l16[a]];
          if (arc_w) x += arc_w[a];
          if (sc.arc_scores) x += sc.arc_scores[a];
          const ME wgt = exp_split(x);
          const float2 bd = bl[d];
          p = ldexpf((wgt.m * bd.x) * rs, max(wgt.e + __float_as_int(bd.y) - es, -300));
        }
      }
      float v = p;
      v += row_shr_zero<1>(v);
      v += row_shr_zero<2>(v);
      v += row_shr_zero<4>(v);
      v += row_shr_zero<8>(v);
      const float cum = cum_base + v;
      const int sh = (int)(threadIdx.x & 48);  // first lane of this row within the wave
      const uint32_t hit = (uint32_t)(__ballot(more && p > 0.0f && u < cum) >> sh) & 0xffffu;
      const uint32_t pos = (uint32_t)(__ballot(more && p > 0.0f) >> sh) & 0xffffu;
      const int f = hit ? __builtin_ctz(hit) : 0, l = pos ? 31 - __builtin_clz(pos) : 0;
      const float x_f = __shfl(x, f, 16), x_l = __shfl(x, l, 16), c_end = __shfl(cum, 15, 16);
      const int d_f = __shfl(d, f, 16), d_l = __shfl(d, l, 16);
      if (more) {
        if (hit) { chosen = c + f; sc_ch = x_f; d_ch = d_f; }
        else {
          cum_base = c_end;
          if (pos) { last = c + l; sc_last = x_l; d_last = d_l; }
        }
      }
    }
    if (active) {
      if (chosen < 0) { chosen = last; sc_ch = sc_last; d_ch = d_last; }
      if (chosen < 0) { ok = false; active = false; }
      else {
        if (r == 0) {
          out[t] = lat.arc_l16[chosen];
          if (outa) outa[t] = chosen;
        }
        tot += sc_ch;
        s = d_ch;
        ++t;
      }
    }
  }
  if (!live) return;
  if (!ok && r == 0) atomicExch(status, NFST_ERR_LENGTH);
  if (r == 0) {
    lengths[walk] = ok ? t : -1;
    logq[walk] = ok ? (float)((double)tot - logz64[b]) : kNegInf;
  }
  for (int j = t + r; j < max_len; j += 16) { out[j] = pad; if (outa) outa[j] = -1; }
}

__device__ __forceinline__ int find_arc(const int32_t *arc_label, int r0, int r1, int label) {
  int lo = r0, hi = r1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (arc_label[mid] < label) lo = mid + 1; else hi = mid;
  }
  return (lo < r1 && arc_label[lo] == label) ? lo : -1;
}

__global__ __launch_bounds__(64) void k_score_paths(nfst_batch lat, nfst_scores sc,
                                                    const int32_t *marks, int K, int max_len,
                                                    float *path_score, int32_t *end_state) {
  const int b = blockIdx.x;
  const int k = blockIdx.y * 64 + threadIdx.x;
  if (k >= K) return;
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  const size_t walk = (size_t)b * K + k;
  const int32_t *mk = marks + walk * max_len;
  int s = 0;
  float tot = 0.0f;
  for (int t = 0; t < max_len; ++t) {
    const int l = mk[t];
    const int a = (l >= 0 && l < lat.vocab) ? find_arc(lat.arc_label, rp[s], rp[s + 1], l) : -1;
    if (a < 0) { tot = kNegInf; s = 0; break; }
    const int d = lat.arc_dst[a];
    if (d != s) tot += arc_score(theta, arc_w, sc.arc_scores, l, a);
    s = d;
  }
  path_score[walk] = tot;
  end_state[walk] = s;
}

// ------------------------------------------------------------------ per-step gathers
__global__ void k_step(nfst_batch lat, const int64_t *state, const int64_t *label, int64_t *next,
                       int K, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / K);
  const Meta m = load_meta(lat.meta, b);
  const int64_t s = state[i], l = label[i];
  int64_t r = 0;
  if (s >= 0 && s < m.n_rows && l >= 0 && l < lat.vocab) {
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    const int a = find_arc(lat.arc_label, rp[s], rp[s + 1], (int)l);
    if (a >= 0) r = lat.arc_dst[a];
  }
  next[i] = r;
}

// MODE 0: emission mask (0 / weight / -inf); MODE 1: values[row_off + transition[state, l]]
template <int MODE>
__global__ __launch_bounds__(64) void k_row_gather(nfst_batch lat, const int64_t *state,
                                                   const float *values, const int64_t *inp, int pad,
                                                   int bos, int eos, int has_to_end, float *out,
                                                   int K) {
  const int64_t i = blockIdx.x;
  const int b = (int)(i / K);
  const Meta m = load_meta(lat.meta, b);
  const int64_t s = state[i];
  int r0 = 0, r1 = 0;
  if (s >= 0 && s < m.n_rows) {
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    r0 = rp[s]; r1 = rp[s + 1];
  }
  float *o = out + (size_t)i * lat.vocab;
  for (int l = threadIdx.x; l < lat.vocab; l += 64) {
    const int a = find_arc(lat.arc_label, r0, r1, l);
    float v;
    if (MODE == 0) {
      v = (a < 0) ? kNegInf : (lat.weighted ? lat.arc_w[a] : 0.0f);
      if (inp) {
        const int64_t p = inp[i];
        const bool ended = (p == eos) || (p == pad);
        if (l == bos || (ended ? (l != pad) : (l == pad))) v = kNegInf;
        if (has_to_end && !ended && l != eos) v = kNegInf;
      }
    } else {
      v = values[m.row_off + (a < 0 ? 0 : lat.arc_dst[a])];
    }
    o[l] = v;
  }
}

__global__ void k_gather_label_scores(nfst_batch lat, nfst_scores sc, float *out) {
  const int b = blockIdx.y;
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  for (int a = m.arc_off + blockIdx.x * blockDim.x + threadIdx.x; a < m.arc_off + m.n_arcs;
       a += gridDim.x * blockDim.x)
    out[a] = arc_score(theta, arc_w, sc.arc_scores, lat.arc_label[a], a);
}

// ------------------------------------------------------------------ sequence scoring
// One 256-thread workgroup per sequence; each wave takes positions t = wave,
// wave+4, ...: masked (log-)softmax over V with wave64 shuffles, gather of the
// realised mark, pad positions contribute 0 (scorers.py:1564-1611).
__device__ __forceinline__ float seq_mask(int v, int t, int prev, int pad, int bos, int eos,
                                          int max_length) {
  float mk = 0.0f;
  if (t == 0) {
    if (v == bos || v == pad) mk = kNegInf;
    return mk;
  }
  const bool ended = (prev == eos) || (prev == pad);
  if (ended ? (v != pad) : (v == pad)) mk = kNegInf;
  if (v == bos) mk = kNegInf;
  if (max_length >= 0 && t > max_length && !ended && v != eos) mk = kNegInf;
  return mk;
}

// wave64 all-reduce without LDS traffic: quad permutes and row mirrors (DPP) reduce each
// row of 16 lanes, v_readlane collects the four row results
__device__ __forceinline__ float read_lane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  const float a = read_lane_f(v, 0), b = read_lane_f(v, 16), c = read_lane_f(v, 32), d = read_lane_f(v, 48);
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  const float a = read_lane_f(v, 0), b = read_lane_f(v, 16), c = read_lane_f(v, 32), d = read_lane_f(v, 48);
  return (a + b) + (c + d);
}

// Streaming version for V % 4 == 0, V <= 1024: every wave keeps RB rows in registers
// (16-byte loads, RB * NV of them in flight per lane -- the kernel is HBM-bound and would be
// latency-bound with one row at a time), two-pass softmax per row (max, then sum of exp).
template <int NV, int RB>
__global__ __launch_bounds__(256) void k_path_logprob_v4(const float *__restrict__ scores,
                                                         const int64_t *__restrict__ marks, int T, int V,
                                                         int pad, int bos, int eos, int max_length,
                                                         float temp, int normalize, float smoothing,
                                                         float *out) {
  __shared__ float part[4];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t *mk = marks + n * T;
  const float rtemp = 1.0f / temp;
  // Row-level legality (scorers.py:59-83) only depends on three row flags; which of this lane's
  // 4 NV columns are illegal under each is a lane constant, one bit per column:
  //   normal / first row: bos, pad;  after eos or pad: everything but pad;  forced end: everything but eos
  uint32_t m_norm = 0, m_end = 0, m_force = 0;
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = (c * 64 + lane) * 4 + k;
      const uint32_t bit = 1u << (c * 4 + k);
      const bool oob = col >= V;
      if (oob || col == bos || col == pad) m_norm |= bit;
      if (oob || col == bos || col != pad) m_end |= bit;
      if (oob || col == bos || col == pad || col != eos) m_force |= bit;
    }
  float acc = 0.0f;
  for (int t0 = wave * RB; t0 < T; t0 += 4 * RB) {
    float4 v[RB][NV];
    int lab[RB], prev[RB];
    float lraw[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int t = min(t0 + r, T - 1);  // clamped: rows past the end are loaded but not used
      const float4 *row = reinterpret_cast<const float4 *>(scores + ((size_t)n * T + t) * V);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const int q = c * 64 + lane;
        v[r][c] = (4 * q < V) ? row[q] : make_float4(kNegInf, kNegInf, kNegInf, kNegInf);
      }
      lab[r] = (int)mk[t];
      prev[r] = t > 0 ? (int)mk[t - 1] : -1;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)  // the realised marks' scores: all RB gathers in flight together (L2 hits)
      lraw[r] = scores[((size_t)n * T + min(t0 + r, T - 1)) * V + lab[r]];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int t = t0 + r;
      if (t >= T) break;
      float sel;
      const float lmsk = seq_mask(lab[r], t, prev[r], pad, bos, eos, max_length);
      const float lx = ((lab[r] == pad ? 0.0f : lraw[r]) + lmsk) / temp + lmsk;
      sel = lx;
      if (normalize || smoothing > 0.0f) {
        const bool first = t == 0;
        const bool ended = !first && (prev[r] == eos || prev[r] == pad);
        const bool force = !first && max_length >= 0 && t > max_length && !ended;
        const uint32_t bad = ended ? m_end : (force ? m_force : m_norm);
        float mx = kNegInf;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          float *e = reinterpret_cast<float *>(&v[r][c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // the pad column counts as score 0 (scorers.py:1679-1683); it is only legal in `ended` rows
            const float x = ended ? 0.0f : e[k] * rtemp;
            e[k] = (bad & (1u << (c * 4 + k))) ? kNegInf : x;
            mx = fmaxf(mx, e[k]);
          }
        }
        float lse = 0.0f;
        if (normalize) {
          mx = wave_max(mx);
          float sm = 0.0f;
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) sm += __expf(e[k] - mx);  // masked / padding columns: exp(-inf) = 0
          }
          sm = wave_sum(sm);
          lse = mx + logf(sm);
          sel = lx - lse;  // an all -inf row gives NaN, like the reference
        }
        if (smoothing > 0.0f) {
          // training: label-smoothed target (scorers.py:1502-1528, 1584-1592): weight 1 - s on
          // the realised mark, s / (cnt - 1) on every other legal mark, values clamped to +-1e9
          float sx = 0.0f, cnt = 0.0f;
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
            for (int k = 0; k < 4; ++k)
              if (e[k] > kNegInf) { sx += e[k]; cnt += 1.0f; }
          }
          sx = wave_sum(sx);
          cnt = wave_sum(cnt);
          const float own = fminf(fmaxf(sel, -10e8f), 10e8f);
          float rest = sx - cnt * lse;  // sum of the legal marks' values ...
          float others = cnt;
          if (lx > kNegInf) { rest -= sel; others -= 1.0f; }  // ... other than the realised one
          sel = (1.0f - smoothing) * own + (others > 0.0f ? (smoothing / (cnt - 1.0f)) * rest : 0.0f);
        }
      }
      acc += sel * (lab[r] != pad ? 1.0f : 0.0f);
    }
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ __launch_bounds__(256) void k_path_logprob(const float *scores, const int64_t *marks,
                                                      int T, int V, int pad, int bos, int eos,
                                                      int max_length, float temp, int normalize,
                                                      float smoothing, float *out) {
  __shared__ float part[4];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t *mk = marks + n * T;
  float acc = 0.0f;
  for (int t = wave; t < T; t += 4) {
    const float *row = scores + ((size_t)n * T + t) * V;
    const int prev = t > 0 ? (int)mk[t - 1] : -1;
    const int lab = (int)mk[t];
    float sel;
    if (normalize) {
      // online max / sum over the lane's slice, then a wave reduction
      float mx = kNegInf, sm = 0.0f;
      for (int v = lane; v < V; v += 64) {
        const float msk = seq_mask(v, t, prev, pad, bos, eos, max_length);
        const float x = ((v == pad ? 0.0f : row[v]) + msk) / temp + msk;
        if (x > mx) { sm = sm * expf(mx - x) + 1.0f; mx = x; }
        else if (x > kNegInf) sm += expf(x - mx);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        const float omx = __shfl_xor(mx, d), osm = __shfl_xor(sm, d);
        const float nm = fmaxf(mx, omx);
        const float a = (mx > kNegInf) ? sm * expf(mx - nm) : 0.0f;
        const float c = (omx > kNegInf) ? osm * expf(omx - nm) : 0.0f;
        sm = a + c;
        mx = nm;
      }
      const float msk = seq_mask(lab, t, prev, pad, bos, eos, max_length);
      const float x = ((lab == pad ? 0.0f : row[lab]) + msk) / temp + msk;
      sel = x - (mx + logf(sm));  // all -inf row: -inf - (-inf + log 0) = NaN, like the reference
    } else {
      const float msk = seq_mask(lab, t, prev, pad, bos, eos, max_length);
      sel = ((lab == pad ? 0.0f : row[lab]) + msk) / temp + msk;
    }
    if (smoothing > 0.0f) {
      // label-smoothed target (scorers.py:1502-1528, 1584-1592)
      float lse = 0.0f;
      const float lmsk = seq_mask(lab, t, prev, pad, bos, eos, max_length);
      const float lx = ((lab == pad ? 0.0f : row[lab]) + lmsk) / temp + lmsk;
      if (normalize) lse = lx - sel;
      float sx = 0.0f, cnt = 0.0f;
      for (int v = lane; v < V; v += 64) {
        const float msk = seq_mask(v, t, prev, pad, bos, eos, max_length);
        const float x = ((v == pad ? 0.0f : row[v]) + msk) / temp + msk;
        if (x > kNegInf) { sx += x; cnt += 1.0f; }
      }
      sx = wave_sum(sx);
      cnt = wave_sum(cnt);
      const float own = fminf(fmaxf(sel, -10e8f), 10e8f);
      float rest = sx - cnt * lse, others = cnt;
      if (lx > kNegInf) { rest -= sel; others -= 1.0f; }
      sel = (1.0f - smoothing) * own + (others > 0.0f ? (smoothing / (cnt - 1.0f)) * rest : 0.0f);
    }
    acc += sel * (lab != pad ? 1.0f : 0.0f);
  }
  // every lane of a wave holds the same acc; reduce the 4 waves in a fixed order
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ void k_iwae(const float *log_p, const float *log_q, int B, int K, float *log_w,
                       float *log_marginal) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = kNegInf;
  for (int k = 0; k < K; ++k) {
    const float w = log_p[(size_t)b * K + k] - log_q[(size_t)b * K + k];
    log_w[(size_t)b * K + k] = w;
    mx = fmaxf(mx, w);
  }
  float sm = 0.0f;
  for (int k = 0; k < K; ++k) sm += expf(log_w[(size_t)b * K + k] - mx);
  log_marginal[b] = (mx + logf(sm)) - logf((float)K);
}

// ------------------------------------------------------------------ host helpers
int check_batch(const nfst_batch *lat) {
  if (!lat || lat->n_lattices <= 0 || lat->vocab <= 0 || lat->max_rows <= 0) return NFST_ERR_ARG;
  if (!lat->meta || !lat->row_ptr || !lat->fwd_stream || !lat->bwd_stream) return NFST_ERR_ARG;
  if (lat->total_arcs > 0 && (!lat->arc_src || !lat->arc_dst || !lat->arc_label)) return NFST_ERR_ARG;
  if ((lat->fwd_slots > 0 && !lat->fwd_perm) || (lat->bwd_slots > 0 && !lat->bwd_perm)) return NFST_ERR_ARG;
  if (lat->weighted && !lat->arc_w) return NFST_ERR_ARG;
  if (lat->max_rows > NFST_MAX_ROWS || lat->vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  if (((uintptr_t)lat->fwd_stream | (uintptr_t)lat->bwd_stream) & 15) return NFST_ERR_ARG;
  return NFST_OK;
}
int check_scores(const nfst_batch *lat, const nfst_scores *sc) {
  if (!sc || !sc->theta) return NFST_ERR_ARG;
  if (sc->theta_stride != 0 && sc->theta_stride < lat->vocab) return NFST_ERR_ARG;
  return NFST_OK;
}
int hip_status(hipError_t e) { return e == hipSuccess ? NFST_OK : NFST_ERR_HIP; }

constexpr int64_t kMaxLds = 160 * 1024;

// Dynamic LDS above 64 KiB needs a per-kernel opt-in; it is sticky, so it is requested
// once per kernel and size (hipFuncSetAttribute is slow and not capturable in a graph).
template <class K>
int set_lds(K kernel, int64_t bytes) {
  if (bytes > kMaxLds) return NFST_ERR_LIMIT;
  if (bytes <= 64 * 1024) return NFST_OK;
  static const void *seen_fn[32];
  static int64_t seen_bytes[32];
  static int n_seen = 0;
  const void *fn = reinterpret_cast<const void *>(kernel);
  for (int i = 0; i < n_seen; ++i)
    if (seen_fn[i] == fn) {
      if (seen_bytes[i] >= bytes) return NFST_OK;
      int rc = hip_status(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      if (rc == NFST_OK) seen_bytes[i] = bytes;
      return rc;
    }
  int rc = hip_status(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (rc == NFST_OK && n_seen < 32) { seen_fn[n_seen] = fn; seen_bytes[n_seen] = bytes; ++n_seen; }
  return rc;
}

}  // namespace

extern "C" {

int nfst_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0 ? 1 : 0;
}

int64_t nfst_lds_bytes(const nfst_batch *lat) {
  if (!lat) return NFST_ERR_ARG;
  return LdsPlan(lat->max_rows, lat->vocab).fb_bytes(kMinRing, kRawSlotsShared, lat->weighted != 0);
}

// number of CUs of the current device (cached per process; 256 on MI355X)
static int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      n = v;
    else
      n = 256;
  }
  return n;
}

// Ring sizes per sweep from the LDS budget of one workgroup, and which kernel flavour runs.
//   deep   (at most one lattice per CU): loader + decoder + sweep waves, deep staging ring;
//   shared (more lattices than CUs): the decoder loads for itself, shallow staging ring, and
//          two workgroups share a CU's 160 KiB when the lattices are small enough.
// A lattice too large for the deep rings runs the shared flavour with the whole CU.
struct RingCfg { int R, RS; bool self; };
static bool ring_config(const LdsPlan &plan, bool fb, bool extra, bool deep, RingCfg *c) {
  const int n_rings = fb ? 2 : 1;
  const int64_t slot = (int64_t)kSlotWords * 4 * n_rings;
  auto fixed = [&](int RS) { return fb ? plan.fb_bytes(0, RS, extra) : plan.bwd_bytes(0, RS, extra); };
  auto clampr = [](int64_t r) { return (int)(r > kMaxRing ? kMaxRing : r); };
  if (deep) {
    const int64_t r = (kMaxLds - fixed(kRawSlotsDeep)) / slot;
    if (r >= kMinRing) { *c = {clampr(r), kRawSlotsDeep, false}; return true; }
  } else {
    const int64_t r = (kMaxLds / 2 - fixed(kRawSlotsShared)) / slot;
    if (r >= kMinRing + 1) { *c = {clampr(r), kRawSlotsShared, true}; return true; }
  }
  const int64_t r = (kMaxLds - fixed(kRawSlotsShared)) / slot;
  if (r < kMinRing) return false;
  *c = {clampr(r), kRawSlotsShared, true};
  return true;
}

int nfst_backward(const nfst_batch *lat, const nfst_scores *scores, float *logbeta, double *logz64,
                  float *logz32, float *beta_me, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  const bool extra = (lat->weighted && lat->arc_w) || scores->arc_scores;
  const LdsPlan plan(lat->max_rows, lat->vocab);
  RingCfg cfg;
  if (!ring_config(plan, false, extra, lat->n_lattices <= cu_count(), &cfg)) return NFST_ERR_LIMIT;
  const int R = cfg.R, RS = cfg.RS;
  const int64_t lds = plan.bwd_bytes(R, RS, extra);
#define NFST_LAUNCH_BWD(NT, EX)                                                                          \
  {                                                                                                    \
    if ((rc = set_lds(k_backward<NT, EX>, lds))) return rc;                                            \
    hipLaunchKernelGGL((k_backward<NT, EX>), dim3(lat->n_lattices), dim3(NT), (size_t)lds,             \
                       (hipStream_t)stream, *lat, *scores, R, RS, logbeta, logz64, logz32, (float2 *)beta_me); \
  }
  // 512 threads: loader + decoder + sweep (deep); 256 threads: self-loading decoder + sweep
  if (!cfg.self) { if (extra) NFST_LAUNCH_BWD(512, true) else NFST_LAUNCH_BWD(512, false) }
  else { if (extra) NFST_LAUNCH_BWD(256, true) else NFST_LAUNCH_BWD(256, false) }
#undef NFST_LAUNCH_BWD
  return hip_status(hipGetLastError());
}

int nfst_forward_backward(const nfst_batch *lat, const nfst_scores *scores, float *logalpha,
                          float *logbeta, double *logz64, float *logz32, float *posterior,
                          float *grad_theta, float *beta_me, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (posterior && ((uintptr_t)posterior & 15)) return NFST_ERR_ARG;
  if (!lat->arc_sd || !lat->arc_l16 || ((uintptr_t)lat->arc_sd & 15) || ((uintptr_t)lat->arc_l16 & 7)) return NFST_ERR_ARG;
  const bool extra = (lat->weighted && lat->arc_w) || scores->arc_scores;
  const LdsPlan plan(lat->max_rows, lat->vocab);
  const int cus = cu_count();
  RingCfg cfg;
  if (!ring_config(plan, true, extra, lat->n_lattices <= cus, &cfg)) return NFST_ERR_LIMIT;
  const int R = cfg.R, RS = cfg.RS;
  const int64_t lds = plan.fb_bytes(R, RS, extra);
#define NFST_LAUNCH_FB(NT, EX)                                                                            \
  {                                                                                                     \
    if ((rc = set_lds(k_forward_backward<NT, EX>, lds))) return rc;                                     \
    hipLaunchKernelGGL((k_forward_backward<NT, EX>), dim3(lat->n_lattices), dim3(NT), (size_t)lds,      \
                       (hipStream_t)stream, *lat, *scores, R, RS, logalpha, logbeta, logz64, logz32, posterior, \
                       grad_theta, (float2 *)beta_me);                                                  \
  }
  // 1024 threads: loaders + decoders + sweeps and 10 more waves for the posterior pass (deep);
  // 512 / 256 threads: self-loading decoders + sweeps, two workgroups per CU when they fit
  if (!cfg.self) { if (extra) NFST_LAUNCH_FB(1024, true) else NFST_LAUNCH_FB(1024, false) }
  else if (lat->n_lattices <= 2 * cus) { if (extra) NFST_LAUNCH_FB(512, true) else NFST_LAUNCH_FB(512, false) }
  else { if (extra) NFST_LAUNCH_FB(256, true) else NFST_LAUNCH_FB(256, false) }
#undef NFST_LAUNCH_FB
  return hip_status(hipGetLastError());
}

int nfst_viterbi(const nfst_batch *lat, const nfst_scores *scores, float *best, int32_t *paths,
                 int32_t *path_arcs, int32_t *lengths, int32_t max_len, int32_t pad, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!best || !paths || !lengths || max_len <= 0) return NFST_ERR_ARG;
  const int64_t lds = (int64_t)lat->max_rows * 12 + (int64_t)lat->vocab * 4 + 16;
  if ((rc = set_lds(k_viterbi, lds))) return rc;
  hipLaunchKernelGGL(k_viterbi, dim3(lat->n_lattices), dim3(kVitThreads), (size_t)lds, (hipStream_t)stream, *lat,
                     *scores, best, paths, path_arcs, lengths, (int)max_len, (int)pad);
  return hip_status(hipGetLastError());
}

int nfst_sample_paths(const nfst_batch *lat, const nfst_scores *scores, const float *beta_me,
                      const double *logz64, int32_t k, int32_t max_len, const float *uniforms,
                      uint64_t seed, int32_t pad, int32_t *paths, int32_t *path_arcs,
                      int32_t *lengths, float *logq, int32_t *status, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!beta_me || !logz64 || !paths || !lengths || !logq || !status || k <= 0 || max_len <= 0)
    return NFST_ERR_ARG;
  if (!lat->arc_sd || !lat->arc_l16) return NFST_ERR_ARG;
  const int stage_theta = (int64_t)lat->max_rows * 8 + (int64_t)lat->vocab * 4 <= 96 * 1024;
  const int64_t lds = (int64_t)lat->max_rows * 8 + (stage_theta ? (int64_t)lat->vocab * 4 : 0);
  if ((rc = set_lds(k_sample, lds))) return rc;
  hipLaunchKernelGGL(k_sample, dim3(lat->n_lattices, (k + kWalksPerBlock - 1) / kWalksPerBlock), dim3(kSampleThreads),
                     (size_t)lds, (hipStream_t)stream,
                     *lat, *scores, (const float2 *)beta_me, logz64, (int)k, (int)max_len, uniforms,
                     seed, (int)pad, stage_theta, paths, path_arcs, lengths, logq, status);
  return hip_status(hipGetLastError());
}

int nfst_score_paths(const nfst_batch *lat, const nfst_scores *scores, const int32_t *marks, int32_t k,
                     int32_t max_len, float *path_score, int32_t *end_state, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!marks || !path_score || !end_state || k <= 0 || max_len <= 0) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_score_paths, dim3(lat->n_lattices, (k + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, *lat, *scores, marks, (int)k, (int)max_len, path_score,
                     end_state);
  return hip_status(hipGetLastError());
}

int nfst_step(const nfst_batch *lat, const int64_t *state, const int64_t *label, int64_t *next,
              int32_t k, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!state || !label || !next || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *lat,
                     state, label, next, (int)k, n);
  return hip_status(hipGetLastError());
}

int nfst_emission_mask(const nfst_batch *lat, const int64_t *state, const int64_t *inp, int32_t pad,
                       int32_t bos, int32_t eos, int32_t has_to_end, float *out, int32_t k,
                       void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!state || !out || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_row_gather<0>, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, *lat, state,
                     (const float *)nullptr, inp, (int)pad, (int)bos, (int)eos, (int)has_to_end, out,
                     (int)k);
  return hip_status(hipGetLastError());
}

int nfst_beta_logits(const nfst_batch *lat, const float *values, const int64_t *state, float *out,
                     int32_t k, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!values || !state || !out || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_row_gather<1>, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, *lat, state,
                     values, (const int64_t *)nullptr, 0, 0, 0, 0, out, (int)k);
  return hip_status(hipGetLastError());
}

int nfst_gather_label_scores(const nfst_batch *lat, const nfst_scores *scores, float *out, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!out) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_gather_label_scores, dim3(8, lat->n_lattices), dim3(256), 0, (hipStream_t)stream,
                     *lat, *scores, out);
  return hip_status(hipGetLastError());
}

int nfst_path_logprob(const float *scores, const int64_t *marks, int64_t n, int32_t t, int32_t vocab,
                      int32_t pad, int32_t bos, int32_t eos, int32_t max_length, float temp,
                      int32_t normalize, float smoothing, float *out, void *stream) {
  if (!scores || !marks || !out || n <= 0 || t <= 0 || vocab <= 0 || !(temp > 0.0f)) return NFST_ERR_ARG;
  if (!(smoothing >= 0.0f && smoothing < 1.0f)) return NFST_ERR_ARG;  // scorers.py:1514
  if (n > 0x7fffffffll) return NFST_ERR_LIMIT;
#define NFST_LAUNCH_PLP(NV, RB)                                                                         \
  hipLaunchKernelGGL((k_path_logprob_v4<NV, RB>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,  \
                     scores, marks, (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length,   \
                     temp, (int)normalize, smoothing, out)
  if (vocab % 4 == 0 && vocab <= 256 && ((uintptr_t)scores & 15) == 0) NFST_LAUNCH_PLP(1, 8);
  else if (vocab % 4 == 0 && vocab <= 512 && ((uintptr_t)scores & 15) == 0) NFST_LAUNCH_PLP(2, 4);
  else if (vocab % 4 == 0 && vocab <= 1024 && ((uintptr_t)scores & 15) == 0) NFST_LAUNCH_PLP(4, 2);
  else
    hipLaunchKernelGGL(k_path_logprob, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, scores, marks,
                       (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length, temp,
                       (int)normalize, smoothing, out);
#undef NFST_LAUNCH_PLP
  return hip_status(hipGetLastError());
}

int nfst_iwae(const float *log_p, const float *log_q, int32_t b, int32_t k, float *log_w,
              float *log_marginal, void *stream) {
  if (!log_p || !log_q || !log_w || !log_marginal || b <= 0 || k <= 0) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_iwae, dim3((b + 127) / 128), dim3(128), 0, (hipStream_t)stream, log_p, log_q,
                     (int)b, (int)k, log_w, log_marginal);
  return hip_status(hipGetLastError());
}

}  // extern "C"
