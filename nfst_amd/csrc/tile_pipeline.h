// tile_pipeline.h -- tile programs: loader (LDS-DMA), decoder and the single-wave sweep
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
#pragma once

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
constexpr int kSlotWords = 64 * (1 + 3 * 4);    // decoded tile for U = 4: 3328 B
constexpr int kSlotWords2 = 1024;               // ring slot of the tile-wave pipeline: 64 lanes x 64 bytes
constexpr int kSlotWordsP = 1280;               // ... of its precise flavour (float64 mantissas): 64 lanes x 80 bytes
constexpr int kPreciseTiles = 192;              // programs with more tiles than this run the precise flavour (semiring.h)
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

// ---- fused sweep: one wave loads, decodes and sweeps ---------------------------------------
// Round-2 measurements: the loader -> decoder -> sweep pipeline spends most
// of a tile's 267 ns in its two hand-offs through LDS (an empty pipeline already needs 177 ns per
// tile), and an LDS access of a wave waits for that wave's own LDS-DMAs, so a wave cannot both
// stream by DMA and talk through LDS.  The fused sweep has no hand-off: ONE wave per direction
//   * keeps kFusedDepth tiles of its program in flight with plain global loads whose destinations
//     are accumulation registers (AGPRs) the compiler never allocates: the load and, kFusedDepth
//     tiles later, "wait + move to VGPRs" are single asm statements, so no compiler-made copy can
//     touch a register before its data has arrived (hipcc did copy VGPR destinations of asm loads
//     in front of a hand-placed wait);
//   * unpacks the raw tile of the NEXT iteration and gathers its label weights in the shadow of the
//     current tile's operand gathers;
//   * sums and reduces with tile_math (below).
// No staging ring, no decoded ring, no flags: a workgroup's LDS is alpha, beta and the label table,
// so four workgroups share a CU when there are more lattices than CUs.  Compact tile format only
// (16 bytes per lane: the common case); other formats and per-arc extras run the pipeline flavours.
constexpr int kFusedDepth = 8;
#define NFST_AGPR_SLOTS(X) X(0, "a[0:3]", "a0", "a1", "a2", "a3") X(1, "a[4:7]", "a4", "a5", "a6", "a7") \
  X(2, "a[8:11]", "a8", "a9", "a10", "a11") X(3, "a[12:15]", "a12", "a13", "a14", "a15")                  \
  X(4, "a[16:19]", "a16", "a17", "a18", "a19") X(5, "a[20:23]", "a20", "a21", "a22", "a23")              \
  X(6, "a[24:27]", "a24", "a25", "a26", "a27") X(7, "a[28:31]", "a28", "a29", "a30", "a31")
template <int J>
__device__ __forceinline__ void agpr_load(const void *p) {  // 16 bytes per lane -> AGPR set J
#define NFST_X(K, R, A0, A1, A2, A3) \
  if (J == K) asm volatile("global_load_dwordx4 " R ", %0, off nt" ::"v"(p) : "memory", A0, A1, A2, A3);
  NFST_AGPR_SLOTS(NFST_X)
#undef NFST_X
}
template <int J, int N>
__device__ __forceinline__ v4u agpr_take() {  // at most N loads stay in flight; AGPR set J -> VGPRs
  v4u r;
#define NFST_X(K, R, A0, A1, A2, A3)                                                                                      \
  if (J == K)                                                                                                             \
    asm volatile("s_waitcnt vmcnt(%4)\n\tv_accvgpr_read_b32 %0, " A0 "\n\tv_accvgpr_read_b32 %1, " A1                     \
                 "\n\tv_accvgpr_read_b32 %2, " A2 "\n\tv_accvgpr_read_b32 %3, " A3                                        \
                 : "=v"(r.x), "=v"(r.y), "=v"(r.z), "=v"(r.w)                                                             \
                 : "n"(N)                                                                                                 \
                 : "memory");
  NFST_AGPR_SLOTS(NFST_X)
#undef NFST_X
  return r;
}

// terms of a state aligned to a stale wave-uniform exponent, summed over the state's lanes by three
// v_fmac_f32_dpp with a 0 / 1 multiplier (no execution-mask changes)
__device__ __forceinline__ float seg_sum_fmac3(float M, float gf) {
  float k0, k1, k2;
  asm volatile(
      "v_max_f32_e64 %[k0], %[g], %[g] clamp\n\t"
      "v_add_f32_e64 %[k1], %[g], -1.0 clamp\n\t"
      "v_fmac_f32_dpp %[m], %[m], %[k0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_e64 %[k2], %[g], -2.0 clamp\n\t"
      "s_nop 0\n\t"
      "v_fmac_f32_dpp %[m], %[m], %[k1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %[m], %[m], %[k2] row_half_mirror row_mask:0xf bank_mask:0xf"
      : [m] "+v"(M), [k0] "=&v"(k0), [k1] "=&v"(k1), [k2] "=&v"(k2)
      : [g] "v"(gf));
  return M;
}

// One tile of a sum-product sweep: this lane's products, their sum, the sum over the state's
// lanes, normalisation, the leaders' stores.  `trash` is a per-lane LDS location the other lanes store
// to (no execution-mask change, no branch).
// Fast path (groups of up to 8 lanes).  Scaling by a power of two is exact, so the terms of a state
// may be aligned to ANY common exponent that keeps them inside float32's range, not only to their
// maximum.  The reference `ref` is wave-uniform and one tile old (the exponent lane 0 ended the
// previous tile with: values drift by a few binary orders per level): a term is one multiply, one
// three-operand add and one ldexp; no maximum of exponents, no rescale, and the K_s multiplier of
// v_fmac_f32_dpp (1 if the state owns more than 2^s lanes, else 0; partners always agree because
// groups are 2^g-aligned) replaces the execution masks.  A tile in which some lane's largest
// non-zero term lies more than 2^64 away from the reference is redone by the exact path (maximum of
// exponents over the state's lanes first); everywhere else the two paths give the same bits
// (terms more than 2^60 below a lane's largest cannot change a float32 sum).  Exact zeros carry the
// exponent kEZero (about -2^28): a lane whose terms are all zero is recognised by that and never
// forces the exact path.
template <int U, bool WIDE>
__device__ __forceinline__ void tile_math(const v2f (&tw)[U], const v2f (&vv)[U], uint32_t ctl, uint32_t dst_addr,
                                          uint32_t trash, bool wide_tile, int &ref) {
  const int gl = (int)((ctl >> 20) & 7u);
  const bool leader = (int)ctl < 0;
  const int nref = -ref;
  float mt[U];
  int d[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    mt[j] = tw[j].x * vv[j].x;
    d[j] = __float_as_int(tw[j].y) + __float_as_int(vv[j].y) + nref;
  }
  int dmax = d[0];
#pragma unroll
  for (int j = 1; j < U; ++j) dmax = max(dmax, d[j]);
  float M = ldexpf(mt[0], d[0]);
#pragma unroll
  for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j]);
  constexpr int kZeroish = -(1 << 27);
  const bool bad = ((uint32_t)(dmax + 64) > 128u) & (dmax > kZeroish);
  int E = ref;
  // the next tile's reference: where lane 0 stands now (kept if lane 0 holds nothing)
  const int e0 = __builtin_amdgcn_readfirstlane(dmax);
  const int ref_old = ref;
  ref = (e0 > kZeroish) ? e0 + ref_old : ref_old;
  if (__builtin_expect((WIDE && wide_tile) || __builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
    // exact path: this lane's sum relative to its largest term, then the maximum over the state's lanes
    M = ldexpf(mt[0], d[0] - dmax);
#pragma unroll
    for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j] - dmax);
    E = dmax + ref_old;
    if (WIDE && wide_tile) {
      seg_reduce_n<6>(M, E, gl);
    } else {
      const uint64_t m0 = __builtin_amdgcn_ballot_w64(gl > 0), m1 = __builtin_amdgcn_ballot_w64(gl > 1),
                     m2 = __builtin_amdgcn_ballot_w64(gl > 2);
      seg_reduce_exec<3>(M, E, m0, m1, m2);
    }
  } else {
    M = seg_sum_fmac3(M, (float)gl);
    E = (M == 0.0f) ? kEZero : E;  // an exact zero keeps the exponent that never wins a maximum
  }
  const float2 r = me_pack(M, E);
  *(lds_v2f *)(uintptr_t)(leader ? dst_addr : trash) = v2f{r.x, r.y};  // trash: this lane's own 8 bytes
}

template <bool WIDE>
struct FusedSweep {
  static constexpr int K = kFusedDepth;
  struct Dec { uint32_t ctl; uint32_t opa[4]; uint32_t lab8[4]; };  // unpacked tile: control word, operand LDS addresses, 8 x label
  // tiles 0 .. K-1 go in flight (short programs load their last tile again: the count stays K)
  __device__ __forceinline__ static void start(const uint32_t *g, int n_tiles, int lane) {
    const int last = max(n_tiles - 1, 0);
    const uint32_t *p = g + lane * 4;
    agpr_load<0>(p + (size_t)min(0, last) * 256); agpr_load<1>(p + (size_t)min(1, last) * 256);
    agpr_load<2>(p + (size_t)min(2, last) * 256); agpr_load<3>(p + (size_t)min(3, last) * 256);
    agpr_load<4>(p + (size_t)min(4, last) * 256); agpr_load<5>(p + (size_t)min(5, last) * 256);
    agpr_load<6>(p + (size_t)min(6, last) * 256); agpr_load<7>(p + (size_t)min(7, last) * 256);
  }
  __device__ __forceinline__ static void unpack(const v4u x, uint32_t val_base, Dec &w) {
    w.ctl = x.x;
    const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w.opa[j] = ((r[j] << 3) & 0xfff8u) + val_base;  // state (13 bits) x 8 + the array's base
      w.lab8[j] = (r[j] >> 10) & 0x3ff8u;             // label (11 bits) x 8
    }
  }
  // the oldest tile in flight (set J) -> w; set J then loads tile `tile + K` (past the end: the last
  // tile again, never used)
  template <int J>
  __device__ __forceinline__ static void take(const uint32_t *g, int tile, int last, int lane, uint32_t val_base, Dec &w) {
    const v4u x = agpr_take<J, K - 1>();
    agpr_load<J>(g + (size_t)min(tile + K, last) * 256 + lane * 4);
    unpack(x, val_base, w);
  }
  __device__ __forceinline__ static void run(const uint32_t *g, int n_tiles, const float2 *val, const float2 *th_, uint32_t trash,
                                             int lane) {
    if (n_tiles <= 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); return; }
    const uint32_t th_base = lds_addr(th_), val_base = lds_addr(val);
    const int last = n_tiles - 1;
    int ref = 0;
    auto gather_weights = [&](const Dec &w, v2f (&tw)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) tw[j] = *(const lds_v2f *)(uintptr_t)(th_base + w.lab8[j]);
    };
    // iteration t: `cur` = tile t unpacked, `tw` = its label weights (gathers issued one iteration
    // earlier); takes tile t+1 into `nxt` and, at the end, issues the gathers of its label weights
#define NFST_FUSED_STEP(JN, CUR, TW, NXT, TWN)                                                             \
    {                                                                                                     \
      v2f vv[4];                                                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) vv[j] = *(const lds_v2f *)(uintptr_t)CUR.opa[j];       \
      asm volatile("" ::: "memory");                                                                      \
      __builtin_amdgcn_sched_barrier(0);  /* nothing is scheduled in front of the operand gathers */      \
      take<JN>(g, t + 1, last, lane, val_base, NXT);                                                      \
      asm volatile("" ::: "memory");                                                                      \
      tile_math<4, WIDE>(TW, vv, CUR.ctl, (CUR.ctl & 0xffffu) + val_base, trash,                           \
                         WIDE && ((__builtin_amdgcn_readfirstlane(CUR.ctl) >> 25) & 1u), ref);            \
      asm volatile("" ::: "memory");                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                  \
      gather_weights(NXT, TWN);                                                                           \
    }
    Dec da, db;
    v2f ta[4], tb[4];
    take<0>(g, 0, last, lane, val_base, da);
    gather_weights(da, ta);
    for (int t0 = 0; t0 < n_tiles; t0 += K) {  // K iterations per trip: AGPR sets and register roles are compile-time
      int t = t0;
      NFST_FUSED_STEP(1, da, ta, db, tb) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(2, db, tb, da, ta) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(3, da, ta, db, tb) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(4, db, tb, da, ta) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(5, da, ta, db, tb) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(6, db, tb, da, ta) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(7, da, ta, db, tb) if (++t >= n_tiles) break;
      NFST_FUSED_STEP(0, db, tb, da, ta)
    }
#undef NFST_FUSED_STEP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing stays in flight
  }
};

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
template <int F>
struct DmaOps {
  static constexpr int value = F == 8 ? 1 : (F == 2 ? 3 : 2);
};

// raw staging slot: [64 control words][64 U records]; compact tiles: [64 x (control word, 3 record words)]
template <int F>
__device__ __forceinline__ void tile_issue(const uint32_t *g, int tile, uint32_t slot_addr, int lane) {
  constexpr int U = fmt_u(F);
  const uint32_t *src = g + (size_t)tile * fmt_words(F);
  if (F == 8) {
    lds_dma16(src + lane * 4, slot_addr);
    return;
  }
  lds_dma4(src + lane, slot_addr);
  if (U == 4) {
    lds_dma16(src + 64 + lane * 4, slot_addr + 256);
  } else if (U == 2) {
    lds_dma4(src + 64 + lane, slot_addr + 256);
    lds_dma4(src + 128 + lane, slot_addr + 512);
  } else {
    lds_dma4(src + 64 + lane, slot_addr + 256);
  }
}

// ---- loader wave ------------------------------------------------------------------
// flags (LDS words, all only grow): rland = raw tiles 0 .. rland-1 have landed in the
// staging ring; the decoder's `land` (tiles decoded) tells which staging slots are free.
// part 1 (kernel entry, before anything else): the first ring-full needs no hand-shake,
// so it is in flight while the workgroup initialises
__device__ __forceinline__ void loader_start(int U, const uint32_t *g, int n_tiles, uint32_t *raw, int RS, int lane) {
  constexpr uint32_t RB = kRawWords * 4;
  const uint32_t raw_base = lds_addr(raw);
  const int n = min(n_tiles, RS);
  for (int d = 0; d < n; ++d) {
    if (U == 8) tile_issue<8>(g, d, raw_base + d * RB, lane);
    else if (U == 4) tile_issue<4>(g, d, raw_base + d * RB, lane);
    else if (U == 2) tile_issue<2>(g, d, raw_base + d * RB, lane);
    else tile_issue<1>(g, d, raw_base + d * RB, lane);
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
template <int F, int AHEAD>
__device__ __forceinline__ void tile_loader(const uint32_t *g, int n_tiles, uint32_t *raw, int RS,
                                            const int *land, int *rland, int lane) {
  constexpr int OPS = DmaOps<F>::value;
  static_assert(OPS * AHEAD <= 63, "vmcnt is 6 bits");
  constexpr uint32_t RB = kRawWords * 4;
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
    tile_issue<F>(g, issued, rb, lane);
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
template <int U, int EXTRA>
struct RawRegs {
  uint32_t ctl;
  uint32_t opoff[U];
  uint32_t lab8[U];
};
template <int F, int EXTRA>
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
    return;
  }
  w.ctl = *(const lds_u32 *)(uintptr_t)(rb + lane * 4);
  if (U == 4) {
    const v4u v = *(const lds_v4u *)(uintptr_t)(rb + 256 + lane * 16);
    rc[0] = v.x; rc[1 % U] = v.y; rc[2 % U] = v.z; rc[3 % U] = v.w;
  } else if (U == 2) {
    const v2u v = *(const lds_v2u *)(uintptr_t)(rb + 256 + lane * 8);
    rc[0] = v.x; rc[1 % U] = v.y;
  } else {
    rc[0] = *(const lds_u32 *)(uintptr_t)(rb + 256 + lane * 4);
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
template <int F, int AHEAD>
__device__ __forceinline__ void self_start_u(const uint32_t *g, int n_tiles, uint32_t *raw, int lane) {
  constexpr uint32_t RB = kRawWords * 4;
  const uint32_t raw_base = lds_addr(raw);
  const int last = max(n_tiles - 1, 0);
#pragma unroll
  for (int d = 0; d < AHEAD; ++d)  // short programs copy their last tile again: the count stays constant
    tile_issue<F>(g, min(d, last), raw_base + d * RB, lane);
}
template <int AHEAD>
__device__ __forceinline__ void self_start(int U, const uint32_t *g, int n_tiles, uint32_t *raw, int lane) {
  if (U == 8) self_start_u<8, AHEAD>(g, n_tiles, raw, lane);
  else if (U == 4) self_start_u<4, AHEAD>(g, n_tiles, raw, lane);
  else if (U == 2) self_start_u<2, AHEAD>(g, n_tiles, raw, lane);
  else self_start_u<1, AHEAD>(g, n_tiles, raw, lane);
}

template <int F, int EXTRA, bool SELF, int AHEAD>
__device__ __forceinline__ void tile_decoder(int n_tiles, const uint32_t *raw, int RS, const int *rland,
                                             const uint32_t *g, const int *xland, int NE,
                                             uint32_t *ring, int R, const int *prog, int *land, const float2 *val,
                                             const float2 *th_, int lane) {
  if (n_tiles <= 0) {
    if (SELF) vm_wait<0>();
    return;
  }
  constexpr int U = fmt_u(F);
  constexpr int OPS = DmaOps<F>::value;
  constexpr uint32_t RB = kRawWords * 4;
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
      tile_issue<F>(g, min(issue_next, last), rb_issue, lane);
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
  // per-arc extras: the weight wave (t mod NE; NE is a power of two) writes the weights of tile t into the
  // ring slot and then publishes t + 1 in its flag; the flag is read at the start of the iteration
  const uint32_t xl_base = lds_addr(xland);
  int x_peek = 0;
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
    if (EXTRA) x_peek = (int)*(const volatile lds_u32 *)(uintptr_t)(xl_base + (uint32_t)(t & (NE - 1)) * 4);
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
      // the arc weights of this tile (label weight x per-arc extras) come from the weight waves: wait for them
      int seen = __builtin_amdgcn_readfirstlane(x_peek);
      while (__builtin_expect(seen < t + 1, 0)) {
        seen = __builtin_amdgcn_readfirstlane(*(const volatile lds_u32 *)(uintptr_t)(xl_base + (uint32_t)(t & (NE - 1)) * 4));
        if (seen < t + 1) __builtin_amdgcn_s_sleep(1);
      }
    } else if (U == 4) {
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
    if (!EXTRA) gather_weights(nxt, twn);
  };
  RawRegs<U, EXTRA> ra, rbb;
  v2f ta[U], tb[U];
  wait_raw(1);
  raw_fetch<F, EXTRA>(rb, lane, ra);
  if (!EXTRA) gather_weights(ra, ta);
  // two iterations per trip so that the register roles alternate without copies
  for (int t = 0; t < n_tiles; t += 2) {
    step(t, ra, ta, rbb, tb);
    if (t + 1 >= n_tiles) break;
    step(t + 1, rbb, tb, ra, ta);
  }
  if (SELF) vm_wait<0>();  // nothing of the staging ring stays in flight
}

// ---- weight waves (kernels with per-arc extras) -----------------------------------
// Per-arc extras (table weights, caller scores; canonical arc order in HBM) reach the sweeps without a
// pre-pass: NE otherwise idle waves per sweep compute the arc weights of the decoded tiles -- label
// weight x exp(extras) as (mantissa, exponent) pairs -- and write them straight into the decoded ring;
// the decoder then only writes control words and operand addresses (it is lighter than without extras)
// and publishes a tile when the weight wave's flag says its weights are there.  Wave ei takes tiles ei,
// ei + NE, ...: the tile's records and slot -> arc map of tile i+4 (its own numbering: two coalesced
// loads), the gathers of tile i+2 (after the first touch a lattice's 80 KB of scores are L2 hits) and the
// arithmetic of tile i are in flight together.  A ring slot is free when the sweep has consumed the
// tile R earlier (`prog`).  Empty slots and carry records (arc -1) load arc 0 and ignore it.
// Straight-line code (the range test of exp_split is a select; EXTRA = 1: one array, 2: table weights
// and caller scores): hipcc keeps counted s_waitcnt only where no branch lies between a load and its use.
// Measured at 256 lattices: gathering both arrays when only one exists doubled the vector-L1 lookups
// (64 per gather instruction) and slowed the tile stream of the loaders by a third.
#ifndef NFST_X_NAP
#define NFST_X_NAP 12
#endif
template <int F, int NE, int XM, bool FULL, bool PREC = false>  // XM: arrays of per-arc extras (0: none -- label weights only --, 1, 2; 3: their sum is staged in LDS)
struct WeightWave {
  static_assert(!PREC || (F == 8 && FULL && XM != 3), "precise flavour: compact tiles, tile waves, extras from HBM / L2");
  static constexpr bool GATHER = XM == 1 || XM == 2, CACHED = XM == 3;
  // FULL: the wave decodes its tiles completely -- control words and operand addresses as well -- straight
  // from HBM into the decoded ring: no loader, no staging ring, no decoder wave (tile waves of the
  // one-lattice-per-CU kernels); otherwise a decoder wave writes those and waits for this wave's flag.
  static constexpr int U = fmt_u(F);
  static constexpr bool BOTH = XM == 2;
  static constexpr int RW = F == 8 ? 4 : (FULL ? 1 : 0) + U;  // raw words of a tile per lane
  struct P { int a[XM != 0 ? U : 1]; uint32_t raw[RW]; };
  struct G { float w[GATHER ? U : 1], s[BOTH ? U : 1]; int a[CACHED ? U : 1]; uint32_t valid; uint32_t raw[RW]; };
  P p0, p1, p2;
  G g0, g1, g2;
  const float *bw, *bs;
  const int32_t *perm;
  const uint32_t *prog_words;
  int n_mine, last, ei;

  __device__ __forceinline__ P ld_tile(int i, int lane) const {
    P p;
    const int t = ei + min(i, max(n_mine - 1, 0)) * NE;  // past this wave's last tile: that tile again
    const int32_t *q = perm + (size_t)t * (64 * U) + lane * U;
    const uint32_t *g = prog_words + (size_t)t * fmt_words(F);
    if (XM == 0) p.a[0] = -1;
    if (U == 4) {
      if (XM != 0) {
        const int4 v = *reinterpret_cast<const int4 *>(q);
        p.a[0] = v.x; p.a[XM != 0 ? 1 : 0] = v.y; p.a[XM != 0 ? 2 : 0] = v.z; p.a[XM != 0 ? 3 : 0] = v.w;
      }
      const uint4 x = *reinterpret_cast<const uint4 *>(g + (F == 8 ? 0 : 64) + lane * 4);
      constexpr int o = (F != 8 && FULL) ? 1 : 0;
      p.raw[o] = x.x; p.raw[o + 1] = x.y; p.raw[o + 2] = x.z; p.raw[o + 3] = x.w;
    } else if (U == 2) {
      if (XM != 0) {
        const int2 v = *reinterpret_cast<const int2 *>(q);
        p.a[0] = v.x; p.a[XM != 0 ? 1 : 0] = v.y;
      }
      const uint2 x = *reinterpret_cast<const uint2 *>(g + 64 + lane * 2);
      constexpr int o = FULL ? 1 : 0;
      p.raw[o] = x.x; p.raw[o + 1] = x.y;
    } else {
      if (XM != 0) p.a[0] = q[0];
      p.raw[FULL ? 1 : 0] = g[64 + lane];
    }
    if (F != 8 && FULL) p.raw[0] = g[lane];  // the control word
    return p;
  }
  // raw words -> control word, 8 x operand state, 8 x label
  __device__ __forceinline__ static void unpack(const uint32_t (&raw)[RW], uint32_t &ctl, uint32_t (&opoff)[U], uint32_t (&lab8)[U]) {
    if (F == 8) {
      ctl = raw[0];
      const uint32_t r[4] = {raw[1], __builtin_amdgcn_alignbit(raw[2], raw[1], 24), __builtin_amdgcn_alignbit(raw[3], raw[2], 16),
                             raw[3] >> 8};
#pragma unroll
      for (int j = 0; j < U; ++j) {
        opoff[j] = (r[j % 4] << 3) & 0xfff8u;
        lab8[j] = (r[j % 4] >> 10) & 0x3ff8u;
      }
    } else {
      ctl = FULL ? raw[0] : 0u;
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const uint32_t rc = raw[(FULL ? 1 : 0) + j];
        opoff[j] = rc & 0xffffu;
        lab8[j] = (rc >> 16) << 3;
      }
    }
  }
  __device__ __forceinline__ G issue(const P &p) const {
    G g;
    g.valid = 0;
#pragma unroll
    for (int j = 0; j < (XM != 0 ? U : 0); ++j) {
      const int a = max(p.a[j], 0);
      if (GATHER) g.w[j] = bw[a];
      if (BOTH) g.s[j] = bs[a];
      if (CACHED) g.a[j] = a;
      g.valid |= (p.a[j] >= 0 ? 1u : 0u) << j;
    }
#pragma unroll
    for (int j = 0; j < RW; ++j) g.raw[j] = p.raw[j];
    return g;
  }
  // kernel entry (right behind the meta record): the records and maps of this wave's first tiles are in
  // flight while the workgroup initialises; the gathers of its first two tiles follow before the barrier
  __device__ __forceinline__ void start_maps(const uint32_t *prog_, const int32_t *perm_, int n_tiles, const Extra ex, int ei_, int lane) {
    perm = perm_; prog_words = prog_; ei = ei_;
    n_mine = n_tiles > ei ? (n_tiles - ei + NE - 1) / NE : 0;
    last = max(n_tiles - 1, 0);
    bw = ex.arc_w ? ex.arc_w : ex.arc_scores;
    bs = ex.arc_scores;
    // a wave without a tile of its own loads nothing: "tile ei" of a program shorter than ei tiles lies behind the
    // program -- for the batch's last lattice behind the slot -> arc map itself, and whatever is there would be used
    // as an arc index by the gathers (a memory fault in round 3's fuzz run: 3-tile program, table weights)
    if (n_tiles <= 0 || n_mine <= 0) { n_mine = 0; return; }
    p0 = ld_tile(0, lane); p1 = ld_tile(1, lane); p2 = ld_tile(2, lane);
  }
  __device__ __forceinline__ void start_gathers(int lane) {
    if (n_mine <= 0) return;
    g0 = issue(p0);
    p0 = ld_tile(3, lane);
    g1 = issue(p1);
  }
  // v2 (FULL, four slots per lane, narrow groups): the decoded tile of tile_sweep2 -- per lane the store address
  // (the state's value for a leader lane, 8 bytes of trash otherwise) and the three 0 / 1 stage multipliers of the
  // segmented sum instead of the control word.  Tile waves use ring slots of kSlotWords2 words for every format.
  // xc_base / xc_first (XM = 3): LDS address of the staged sums of the per-arc extras, and the canonical arc the first one belongs to
  __device__ __forceinline__ void run(uint32_t *ring, int R, const int *prog, int *xland, const float2 *th_, const float2 *val, bool v2,
                                      uint32_t trash, uint32_t xc_base, int xc_first, int lane) {
    const uint32_t xl_a = lds_addr(xland) + (uint32_t)ei * 4;
    if (n_mine <= 0) {
      if (FULL) *(volatile lds_u32 *)(uintptr_t)xl_a = 0x7fffffffu;  // "every tile of mine is there" (there is none)
      return;
    }
    constexpr uint32_t SB = FULL ? kSlotWords2 * 4 : 64 * (1 + 3 * U) * 4;
    const uint32_t ring_base = lds_addr(ring), prog_a = lds_addr(prog);
    const uint32_t th_base = lds_addr(th_), val_base = lds_addr(val);
    int prog_seen = 0;
    auto process = [&](int i, const auto &g) {  // g: G, or P in kernels without extras
      const int t = ei + min(i, n_mine - 1) * NE;
      uint32_t ctl, opoff[U], lab8[U];
      unpack(g.raw, ctl, opoff, lab8);
      if constexpr (PREC) {
        // the decoded tile of tile_sweep2p: values and label weights are 16-byte records (float64 mantissa, exponent):
        // [store address | g, largest g][3 stage multipliers: the high word of 1.0 or 0][4 operand addresses]
        // [4 weight mantissas, float64][4 weight exponents] = 80 bytes per lane
        double om[U];
        int oe[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const v4u tr = *(const lds_v4u *)(uintptr_t)(th_base + 2 * lab8[j]);
          const double tm = __hiloint2double((int)tr.y, (int)tr.x);
          if constexpr (XM == 0) {
            om[j] = tm;
            oe[j] = (int)tr.z;
          } else {
            const double xs = BOTH ? (double)g.w[GATHER ? j : 0] + (double)g.s[BOTH ? j : 0] : (double)g.w[GATHER ? j : 0];
            const ME64 x = exp_split64(((g.valid >> j) & 1u) ? xs : 0.0);
            om[j] = tm * x.m;
            oe[j] = (int)tr.z + x.e;
          }
        }
        while (__builtin_expect(prog_seen < t - R + 1, 0)) {  // the slot's previous tile is consumed
          prog_seen = __builtin_amdgcn_readfirstlane(*(const volatile lds_u32 *)(uintptr_t)prog_a);
          if (prog_seen < t - R + 1) { if (R >= 8) __builtin_amdgcn_s_sleep(NFST_X_NAP); else __builtin_amdgcn_s_sleep(1); }
        }
        asm volatile("" ::: "memory");
        const uint32_t sb = ring_base + (uint32_t)(t % R) * (kSlotWordsP * 4);
        const int gl = (int)((ctl >> 20) & 7u);
        const uint32_t dst = (((int)ctl < 0) ? 2 * (ctl & 0xffffu) + val_base : trash) | (ctl & 0x03f00000u);
        *(lds_v4u *)(uintptr_t)(sb + lane * 16) = v4u{dst, gl > 0 ? 0x3ff00000u : 0u, gl > 1 ? 0x3ff00000u : 0u, gl > 2 ? 0x3ff00000u : 0u};
        *(lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16) = v4u{2 * opoff[0] + val_base, 2 * opoff[1 % U] + val_base, 2 * opoff[2 % U] + val_base, 2 * opoff[3 % U] + val_base};
        *(lds_v4u *)(uintptr_t)(sb + 2048 + lane * 16) = v4u{(uint32_t)__double2loint(om[0]), (uint32_t)__double2hiint(om[0]),
                                                             (uint32_t)__double2loint(om[1 % U]), (uint32_t)__double2hiint(om[1 % U])};
        *(lds_v4u *)(uintptr_t)(sb + 3072 + lane * 16) = v4u{(uint32_t)__double2loint(om[2 % U]), (uint32_t)__double2hiint(om[2 % U]),
                                                             (uint32_t)__double2loint(om[3 % U]), (uint32_t)__double2hiint(om[3 % U])};
        *(lds_v4u *)(uintptr_t)(sb + 4096 + lane * 16) = v4u{(uint32_t)oe[0], (uint32_t)oe[1 % U], (uint32_t)oe[2 % U], (uint32_t)oe[3 % U]};
        asm volatile("" ::: "memory");
        *(volatile lds_u32 *)(uintptr_t)xl_a = (uint32_t)(t + 1);
        return;
      }
      v2f tw[U];
#pragma unroll
      for (int j = 0; j < U; ++j) tw[j] = *(const lds_v2f *)(uintptr_t)(th_base + lab8[j]);
      v2f o[U];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        if (XM == 0) { o[j] = tw[j]; continue; }
        if constexpr (XM != 0) {
          const float xs = CACHED ? *(const __attribute__((address_space(3))) float *)(uintptr_t)(xc_base + (uint32_t)(g.a[CACHED ? j : 0] - xc_first) * 4)
                                  : (BOTH ? g.w[GATHER ? j : 0] + g.s[BOTH ? j : 0] : g.w[GATHER ? j : 0]);
          const ME x = exp_split_nb(((g.valid >> j) & 1u) ? xs : 0.0f);
          o[j] = v2f{tw[j].x * x.m, __int_as_float(__float_as_int(tw[j].y) + x.e)};
        }
      }
      while (__builtin_expect(prog_seen < t - R + 1, 0)) {  // the slot's previous tile is consumed
        prog_seen = __builtin_amdgcn_readfirstlane(*(const volatile lds_u32 *)(uintptr_t)prog_a);
        // (a blocked wave has its tile ready and the sweep is R - NE tiles behind: long naps, few issue slots)
        if (prog_seen < t - R + 1) { if (R >= 8) __builtin_amdgcn_s_sleep(NFST_X_NAP); else __builtin_amdgcn_s_sleep(1); }
      }
      asm volatile("" ::: "memory");
      const uint32_t sb = ring_base + (uint32_t)(t % R) * SB;
      if (FULL && v2) {
        const int gl = (int)((ctl >> 20) & 7u);
        const uint32_t dst = ((int)ctl < 0) ? (ctl & 0xffffu) + val_base : trash;
        *(lds_v4u *)(uintptr_t)(sb + lane * 16) = v4u{dst, gl > 0 ? 0x3f800000u : 0u, gl > 1 ? 0x3f800000u : 0u, gl > 2 ? 0x3f800000u : 0u};
        if (U == 4) {
          *(lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16) = v4u{opoff[0] + val_base, opoff[1 % U] + val_base, opoff[2 % U] + val_base, opoff[3 % U] + val_base};
          *(lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16) = v4f{o[0].x, o[0].y, o[1 % U].x, o[1 % U].y};
          *(lds_v4f *)(uintptr_t)(sb + 3072 + lane * 16) = v4f{o[2 % U].x, o[2 % U].y, o[3 % U].x, o[3 % U].y};
        } else if (U == 2) {
          *(lds_v2u *)(uintptr_t)(sb + 1024 + lane * 8) = v2u{opoff[0] + val_base, opoff[1 % U] + val_base};
          *(lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16) = v4f{o[0].x, o[0].y, o[1 % U].x, o[1 % U].y};
        } else {
          *(lds_u32 *)(uintptr_t)(sb + 1024 + lane * 4) = opoff[0] + val_base;
          *(lds_v2f *)(uintptr_t)(sb + 2048 + lane * 8) = o[0];
        }
        asm volatile("" ::: "memory");
        *(volatile lds_u32 *)(uintptr_t)xl_a = (uint32_t)(t + 1);
        return;
      }
      if (FULL) {
        uint32_t oa[U];
#pragma unroll
        for (int j = 0; j < U; ++j) oa[j] = opoff[j] + val_base;
        *(lds_u32 *)(uintptr_t)(sb + lane * 4) = ctl + val_base;
        if (U == 4) *(lds_v4u *)(uintptr_t)(sb + 256 + lane * 16) = v4u{oa[0], oa[1 % U], oa[2 % U], oa[3 % U]};
        else if (U == 2) *(lds_v2u *)(uintptr_t)(sb + 256 + lane * 8) = v2u{oa[0], oa[1 % U]};
        else *(lds_u32 *)(uintptr_t)(sb + 256 + lane * 4) = oa[0];
      }
      if (U == 4) {
        *(lds_v4f *)(uintptr_t)(sb + 256 + 1024 + lane * 16) = v4f{o[0].x, o[0].y, o[1 % U].x, o[1 % U].y};
        *(lds_v4f *)(uintptr_t)(sb + 256 + 2048 + lane * 16) = v4f{o[2 % U].x, o[2 % U].y, o[3 % U].x, o[3 % U].y};
      } else if (U == 2) {
        *(lds_v4f *)(uintptr_t)(sb + 256 + 512 + lane * 16) = v4f{o[0].x, o[0].y, o[1 % U].x, o[1 % U].y};
      } else {
        *(lds_v2f *)(uintptr_t)(sb + 256 + 256 + lane * 8) = o[0];
      }
      asm volatile("" ::: "memory");
      *(volatile lds_u32 *)(uintptr_t)xl_a = (uint32_t)(t + 1);
    };
    // iteration i: the gathers of tile i+2 (its records and map were loaded two iterations ago), the
    // records and map of tile i+4, then tile i itself; register roles are compile-time: three
    // iterations per trip
#define NFST_X_STEP(PU, GN, PL, GC)  \
    GN = issue(PU);                  \
    PL = ld_tile(i + 4, lane);       \
    process(i, GC);
    // (no exit inside a trip: past its last tile the wave writes that tile again -- same slot, same words --, which
    // keeps the loop a single block per trip; with exits in between hipcc rotated registers through copies that
    // wait for the loads just issued)
    for (int i = 0; i < n_mine; i += 3) {
      NFST_X_STEP(p2, g2, p1, g0) ++i;
      NFST_X_STEP(p0, g0, p2, g1) ++i;
      NFST_X_STEP(p1, g1, p0, g2) i -= 2;
    }
#undef NFST_X_STEP
    // every tile of this wave is there: a sweep that looks one tile past the end of its program never waits
    if (FULL) *(volatile lds_u32 *)(uintptr_t)xl_a = 0x7fffffffu;
  }
};

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
// NEF = 0: one producer (the decoder wave) counts decoded tiles in `land`; NEF > 0 (a power of two): NEF
// tile waves, wave k decodes tiles k, k + NEF, ... and stores tile + 1 in land[k]
template <int U, bool WIDE, int NEF = 0>
__device__ __forceinline__ void tile_sweep(int n_tiles, const uint32_t *ring, int R, int *prog, const int *land,
                                           int lane) {
  if (n_tiles <= 0) return;
  constexpr uint32_t SB = NEF > 0 ? kSlotWords2 * 4 : 64 * (1 + 3 * U) * 4;  // bytes per ring slot
  const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
  int landed = 0;  // wave-uniform copy of the decoder's counter, refreshed only when it runs out
  auto wait_landed = [&](int need) {  // tiles 0 .. need-1 are decoded (NEF: tile need-1 is)
    if (NEF > 0) landed = 0;
    const int *flag = NEF > 0 ? land + ((need - 1) & (NEF - 1)) : land;
    while (__builtin_expect(landed < need, 0)) {
      landed = __builtin_amdgcn_readfirstlane(lds_flag_load(flag));
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
    if (NEF > 0) {  // the flag of tile T+1's wave was read during the previous iteration
      if (T + 1 < n_tiles && __builtin_amdgcn_readfirstlane(land_peek) < T + 2) wait_landed(T + 2);
    } else {
      landed = max(landed, __builtin_amdgcn_readfirstlane(land_peek));
      wait_landed(min(T + 2, n_tiles));
    }
    dec_fetch<U>(sb, lane, nxt);
    land_peek = lds_flag_load(NEF > 0 ? land + ((T + 2) & (NEF - 1)) : land);
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

// role dispatch: role 0 sweeps, role 1 decodes for it, role 2 loads for the decoder (the extras waves
// of kernels with per-arc extras are driven by the kernels: their loads start at kernel entry).
// flags: [0] prog [1] land [2] rland [4 .. 4 + NE) the extras waves' progress
constexpr int kSweepFlags = 8;
template <int EXTRA, bool SELF, int AHEAD, int NE>
__device__ __forceinline__ void run_sweep(int role, int U, bool wide, uint32_t *raw, int RS, const uint32_t *g,
                                          int n_tiles, uint32_t *ring, int R, int *flags,
                                          float2 *val, const float2 *th, int lane) {
  int *prog = flags, *land = flags + 1, *rland = flags + 2, *xland = flags + 4;
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
    if (U == 8) tile_decoder<8, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, xland, NE, ring, R, prog, land, val, th, lane);
    else if (U == 4) tile_decoder<4, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, xland, NE, ring, R, prog, land, val, th, lane);
    else if (U == 2) tile_decoder<2, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, xland, NE, ring, R, prog, land, val, th, lane);
    else tile_decoder<1, EXTRA, SELF, AHEAD>(n_tiles, raw, RS, rland, g, xland, NE, ring, R, prog, land, val, th, lane);
  } else if (role == 2) {
    if (!SELF) {
      if (U == 8) tile_loader<8, AHEAD>(g, n_tiles, raw, RS, land, rland, lane);
      else if (U == 4) tile_loader<4, AHEAD>(g, n_tiles, raw, RS, land, rland, lane);
      else if (U == 2) tile_loader<2, AHEAD>(g, n_tiles, raw, RS, land, rland, lane);
      else tile_loader<1, AHEAD>(g, n_tiles, raw, RS, land, rland, lane);
    }
  }
}

// ---- the sweep wave of the tile-wave pipeline, four slots per lane, narrow groups ---------------------
// A single wave executes in order: every instruction of its loop lies on the level-to-level chain, the
// bookkeeping as much as the arithmetic (measured: ~100 issue slots per tile in tile_sweep, ~4.3 cycles
// each, plus one exposed LDS round trip).  Here everything that does not depend on DP values was moved to the
// tile waves (store address incl. the leader select, the stage multipliers, no end-of-program tests: a
// finished tile wave publishes "infinity"), and the segmented sum is three v_fmac_f32_dpp on terms
// aligned to a stale wave-uniform exponent (tile_math): ~60 issue slots per tile.
template <int U>
struct Dec2 {
  uint32_t dst;
  float k0, k1, k2;
  uint32_t opa[U];
  v2f tw[U];
};
template <int U>
__device__ __forceinline__ void dec2_fetch(uint32_t sb, int lane, Dec2<U> &d) {
  const v4u h = *(const lds_v4u *)(uintptr_t)(sb + lane * 16);
  d.dst = h.x; d.k0 = __uint_as_float(h.y); d.k1 = __uint_as_float(h.z); d.k2 = __uint_as_float(h.w);
  if (U == 4) {
    const v4u a = *(const lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16);
    const v4f p = *(const lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16);
    const v4f q = *(const lds_v4f *)(uintptr_t)(sb + 3072 + lane * 16);
    d.opa[0] = a.x; d.opa[1 % U] = a.y; d.opa[2 % U] = a.z; d.opa[3 % U] = a.w;
    d.tw[0] = v2f{p.x, p.y}; d.tw[1 % U] = v2f{p.z, p.w}; d.tw[2 % U] = v2f{q.x, q.y}; d.tw[3 % U] = v2f{q.z, q.w};
  } else if (U == 2) {
    const v2u a = *(const lds_v2u *)(uintptr_t)(sb + 1024 + lane * 8);
    const v4f p = *(const lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16);
    d.opa[0] = a.x; d.opa[1 % U] = a.y;
    d.tw[0] = v2f{p.x, p.y}; d.tw[1 % U] = v2f{p.z, p.w};
  } else {
    d.opa[0] = *(const lds_u32 *)(uintptr_t)(sb + 1024 + lane * 4);
    d.tw[0] = *(const lds_v2f *)(uintptr_t)(sb + 2048 + lane * 8);
  }
}
// tile_math with the per-lane constants precomputed (same arithmetic, same bits)
template <int U>
__device__ __forceinline__ void tile_math2(const v2f (&tw)[U], const v2f (&vv)[U], const Dec2<U> &c, int &ref) {
  const int nref = -ref;
  float mt[U];
  int d[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    mt[j] = tw[j].x * vv[j].x;
    d[j] = __float_as_int(tw[j].y) + __float_as_int(vv[j].y) + nref;
  }
  int dmax = d[0];
#pragma unroll
  for (int j = 1; j < U; ++j) dmax = max(dmax, d[j]);
  float M = ldexpf(mt[0], d[0]);
#pragma unroll
  for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j]);
  constexpr int kZeroish = -(1 << 27);
  const bool bad = ((uint32_t)(dmax + 64) > 128u) & (dmax > kZeroish);
  int E = ref;
  const int e0 = __builtin_amdgcn_readfirstlane(dmax);
  const int ref_old = ref;
  ref = (e0 > kZeroish) ? e0 + ref_old : ref_old;
  if (__builtin_expect(__builtin_amdgcn_ballot_w64(bad) != 0, 0)) {
    M = ldexpf(mt[0], d[0] - dmax);
#pragma unroll
    for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j] - dmax);
    E = dmax + ref_old;
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(c.k0 != 0.0f), m1 = __builtin_amdgcn_ballot_w64(c.k1 != 0.0f),
                   m2 = __builtin_amdgcn_ballot_w64(c.k2 != 0.0f);
    seg_reduce_exec<3>(M, E, m0, m1, m2);
  } else {
    asm volatile(
        "s_nop 1\n\t"  // (a DPP read needs two wait states after the vector write of its source: the compiler does not look into this block)
        "v_fmac_f32_dpp %[m], %[m], %[k0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %[m], %[m], %[k1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_fmac_f32_dpp %[m], %[m], %[k2] row_half_mirror row_mask:0xf bank_mask:0xf"
        : [m] "+v"(M)
        : [k0] "v"(c.k0), [k1] "v"(c.k1), [k2] "v"(c.k2));
    E = (M == 0.0f) ? kEZero : E;
  }
  const float2 r = me_pack(M, E);
  *(lds_v2f *)(uintptr_t)c.dst = v2f{r.x, r.y};
}

// One tile of tile_sweep2: tile_math2's arithmetic (same bits) in an order that keeps the scalar uses of
// vector results -- the test for the exact path, the next reference exponent -- away from the instructions that
// produce them (a scalar use of a fresh vector result costs ~25 cycles): the fast result is computed
// unconditionally and replaced in the rare tile that needs the exact path.
template <int U>
__device__ __forceinline__ int tile_math3(const Dec2<U> &c, const v2f (&vv)[U], const int ref) {  // returns this lane's largest term exponent - ref
  const int nref = -ref;
  float mt[U];
  int d[U];
#pragma unroll
  for (int j = 0; j < U; ++j) {
    mt[j] = c.tw[j].x * vv[j].x;
    // (one three-operand add with the reference in a scalar register; hipcc splits the C expression in two)
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d[j]) : "v"(__float_as_int(c.tw[j].y)), "v"(__float_as_int(vv[j].y)), "s"(nref));
  }
  int dmax = d[0];
#pragma unroll
  for (int j = 1; j < U; ++j) dmax = max(dmax, d[j]);
  constexpr int kZeroish = -(1 << 27);
  const uint64_t bad = __builtin_amdgcn_ballot_w64(((uint32_t)(dmax + 64) > 128u) & (dmax > kZeroish));
  const int ref_old = ref;
  float M = ldexpf(mt[0], d[0]);
#pragma unroll
  for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j]);
  asm volatile(
      "s_nop 1\n\t"  // (a DPP read needs two wait states after the vector write of its source: the compiler does not look into this block)
      "v_fmac_f32_dpp %[m], %[m], %[k0] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %[m], %[m], %[k1] quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_fmac_f32_dpp %[m], %[m], %[k2] row_half_mirror row_mask:0xf bank_mask:0xf"
      : [m] "+v"(M)
      : [k0] "v"(c.k0), [k1] "v"(c.k1), [k2] "v"(c.k2));
  int E = ref_old;
  if (__builtin_expect(bad != 0, 0)) {
    M = ldexpf(mt[0], d[0] - dmax);
#pragma unroll
    for (int j = 1; j < U; ++j) M += ldexpf(mt[j], d[j] - dmax);
    E = dmax + ref_old;
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(c.k0 != 0.0f), m1 = __builtin_amdgcn_ballot_w64(c.k1 != 0.0f),
                   m2 = __builtin_amdgcn_ballot_w64(c.k2 != 0.0f);
    seg_reduce_exec<3>(M, E, m0, m1, m2);
  }
  // (mantissa, exponent): an exact zero keeps the exponent that never wins a maximum
  int ex;
  const float mant = frexpf(M, &ex);
  const int eo = (M != 0.0f) ? max(E + ex, kEZero) : kEZero;
  *(lds_v2f *)(uintptr_t)c.dst = v2f{mant, __int_as_float(eo)};
  return dmax;
}

// Four tiles per trip (the ring holds a multiple of four slots, so a trip's slots are consecutive and the fetch
// addresses are one base plus immediates; tile t's wave is t mod 4, so the four flags -- one 16-byte read -- are
// checked once per trip for the four tiles the trip fetches; `prog` is published every other tile).
#ifdef NFST_PROF
__device__ unsigned long long fb_prof[4096 * 8];  // per workgroup: stamps of the profiling build (profiles/tune/stamps.py)
#define NFST_STAMP(k) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096) fb_prof[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define NFST_STAMP(k) do { } while (0)
#endif
template <int U, int NEF>
__device__ __forceinline__ void tile_sweep2(int n_tiles, const uint32_t *ring, int R, int *prog, const int *land, int lane, int prof_slot = -1) {
  static_assert(NEF == 4, "four tile waves per sweep");
  (void)prof_slot;
  if (n_tiles <= 0) return;
  constexpr uint32_t SB = kSlotWords2 * 4;
  const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
  const uint32_t land_a = lds_addr(land), prog_a = lds_addr(prog);
  // the four flags in one read: .x = min over the tiles T+1 .. T+3 of (their wave's flag - their offset): all three are decoded
  // iff it exceeds T; .y = the same for tile T+4, which the trip's last step fetches (with a ring of four slots that tile can only
  // be written once the trip has consumed its first tile)
  // (the raw flag words are kept and the margins evaluated where they are tested, a tile or two later: evaluated at once
  // they made hipcc wait for the flag read -- the youngest of nine LDS reads in flight -- in front of the tile's arithmetic)
  auto flags_now = [&]() { return *(const volatile lds_v4u *)(uintptr_t)land_a; };
  // (every word of a copy is kept alive until the copy is looked at -- the empty asm statements below --: with dead words hipcc
  // reused their registers at once and put a full wait for the flag read, the youngest of nine LDS reads in flight, in front of
  // the tile's arithmetic)
  auto margin_x = [](const v4u f) { return min(min((int)f.y - 1, (int)f.z - 2), (int)f.w - 3); };
  auto margin_y = [](const v4u f) { return (int)f.x - 4; };
  auto wait_group = [&](int T, int which) {
    for (;;) {
      const v4u f = flags_now();
      if (__builtin_amdgcn_readfirstlane(which ? margin_y(f) : margin_x(f)) > T) break;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  while (__builtin_amdgcn_readfirstlane((int)*(const volatile lds_u32 *)(uintptr_t)land_a) <= 0) __builtin_amdgcn_s_sleep(1);  // tile 0
  asm volatile("" ::: "memory");
#ifdef NFST_PROF
  if (prof_slot >= 0) NFST_STAMP(prof_slot);
#endif
  uint32_t gbase = ring_base;  // slot of the trip's first tile
  v4u margin = {0u, 0u, 0u, 0u}, margin1 = {0u, 0u, 0u, 0u};  // the flag words as of an earlier tile (per-lane copies): nothing decoded beyond tile 0 is assumed
  int ref = 0;     // the exponent the terms of a tile are aligned to (wave-uniform, a few tiles old)
  int dmax0 = 0;
  Dec2<U> da, db;
  dec2_fetch<U>(gbase, lane, da);
  asm volatile("" ::: "memory");
  // tile T+K: `cur` in registers; fetches tile T+K+1 into `nxt`
#define NFST_S2_STEP(K, CUR, NXT)                                                                          \
  {                                                                                                       \
    v2f vv[U];                                                                                            \
    _Pragma("unroll") for (int j = 0; j < U; ++j) vv[j] = *(const lds_v2f *)(uintptr_t)CUR.opa[j];         \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0); /* nothing is scheduled in front of the operand gathers */         \
    if (K == 3) {                                                                                         \
      gbase = (gbase + 4 * SB == ring_end) ? ring_base : gbase + 4 * SB;                                  \
      asm volatile("" ::"v"(margin1));                                                                    \
      if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_y(margin1)) <= T, 0)) wait_group(T, 1);     \
    }                                                                                                     \
    dec2_fetch<U>(gbase + (K == 3 ? 0u : (K + 1) * SB), lane, NXT);                                       \
    if (K == 1) margin1 = flags_now();                                                                     \
    if (K == 3) margin = flags_now();                                                        \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    const int dm_ = tile_math3<U>(CUR, vv, ref);                                                          \
    if (K == 0) dmax0 = dm_;                                                                              \
    if (K & 1) *(volatile lds_u32 *)(uintptr_t)prog_a = (uint32_t)(T + K + 2); /* tiles 0 .. T+K+1 are consumed */ \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
  int T = 0;
  for (; T + 4 <= n_tiles; T += 4) {  // whole trips: no end-of-program test between the tiles
    asm volatile("" ::"v"(margin));
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_x(margin)) <= T, 0)) wait_group(T, 0);
    NFST_S2_STEP(0, da, db)
    NFST_S2_STEP(1, db, da)
    NFST_S2_STEP(2, da, db)
    NFST_S2_STEP(3, db, da)
    // the reference exponent follows lane 0 once per trip, from the trip's first tile: by now that vector result is old
    const int e0 = __builtin_amdgcn_readfirstlane(dmax0);
    ref = __builtin_amdgcn_readfirstlane((e0 > -(1 << 27)) ? e0 + ref : ref);  // (kept in a scalar register)
  }
  if (T < n_tiles) {  // the last one to three tiles
    asm volatile("" ::"v"(margin));
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_x(margin)) <= T, 0)) wait_group(T, 0);
    NFST_S2_STEP(0, da, db)
    if (T + 1 < n_tiles) {
      NFST_S2_STEP(1, db, da)
      if (T + 2 < n_tiles) NFST_S2_STEP(2, da, db)
    }
  }
#undef NFST_S2_STEP
}

// ---- the precise flavour of tile_sweep2: float64 mantissas (semiring.h) ---------------------------------------
// Same loop, same hand-shakes, same ring protocol; a value is a 16-byte record (float64 mantissa, exponent), a ring
// slot 80 bytes per lane (WeightWave<.., PREC>::process).  The terms of a tile are aligned to the stale wave-uniform
// exponent as in tile_math3 -- with float64's range the window is 2^+-900 instead of 2^+-64 --, the segmented sum is
// two v_mov_b32_dpp + one v_fma_f64 per stage.  Tiles with groups wider than 8 lanes (programs the packer marked
// wide) and tiles outside the window take the exact path: maximum of exponents over the state's lanes first.
struct Dec2P {
  uint32_t dst;         // [0:20) LDS address of the state's record (trash for non-leader lanes) | g << 20 | largest g << 23
  uint32_t k0, k1, k2;  // the high word of 1.0 if the state owns more than 2^s lanes, else 0
  uint32_t opa[4];
  double wm[4];
  int we[4];
};
__device__ __forceinline__ void dec2p_fetch(uint32_t sb, int lane, Dec2P &d) {
  const v4u h = *(const lds_v4u *)(uintptr_t)(sb + lane * 16);
  const v4u a = *(const lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16);
  const v4u p = *(const lds_v4u *)(uintptr_t)(sb + 2048 + lane * 16);
  const v4u q = *(const lds_v4u *)(uintptr_t)(sb + 3072 + lane * 16);
  const v4u e = *(const lds_v4u *)(uintptr_t)(sb + 4096 + lane * 16);
  d.dst = h.x; d.k0 = h.y; d.k1 = h.z; d.k2 = h.w;
  d.opa[0] = a.x; d.opa[1] = a.y; d.opa[2] = a.z; d.opa[3] = a.w;
  d.wm[0] = __hiloint2double((int)p.y, (int)p.x); d.wm[1] = __hiloint2double((int)p.w, (int)p.z);
  d.wm[2] = __hiloint2double((int)q.y, (int)q.x); d.wm[3] = __hiloint2double((int)q.w, (int)q.z);
  d.we[0] = (int)e.x; d.we[1] = (int)e.y; d.we[2] = (int)e.z; d.we[3] = (int)e.w;
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false));
}
// segmented sum of float64 partial sums that share an exponent: a lane whose state owns 2^g lanes takes part in stages 0 .. g-1
template <int STAGES>
__device__ __forceinline__ double seg_sum64(double M, int g) {
  if (STAGES >= 1) { const double o = dpp_d<0xB1>(M); M = (g >= 1) ? M + o : M; }
  if (STAGES >= 2) { const double o = dpp_d<0x4E>(M); M = (g >= 2) ? M + o : M; }
  if (STAGES >= 3) { const double o = dpp_d<0x141>(M); M = (g >= 3) ? M + o : M; }
  if (STAGES >= 4) { const double o = dpp_d<0x140>(M); M = (g >= 4) ? M + o : M; }
  if (STAGES >= 5) { const double o = __shfl_xor(M, 16); M = (g >= 5) ? M + o : M; }
  if (STAGES >= 6) { const double o = __shfl_xor(M, 32); M = (g >= 6) ? M + o : M; }
  return M;
}
__device__ __forceinline__ double ldexp_clamped(double m, int d) {  // (exponent differences may be near -2^28: zero then)
  return ldexp(m, max(d, -2000));
}
__device__ __forceinline__ int tile_math3p(const Dec2P &c, const Rec64 (&vv)[4], const int ref) {  // returns this lane's largest term exponent - ref
  const int nref = -ref;
  double mt[4];
  int d[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mt[j] = c.wm[j] * vv[j].m;
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(d[j]) : "v"(c.we[j]), "v"(vv[j].e), "s"(nref));
  }
  int dmax = d[0];
#pragma unroll
  for (int j = 1; j < 4; ++j) dmax = max(dmax, d[j]);
  constexpr int kZeroish = -(1 << 27);
  const uint32_t wide = (c.dst >> 25) & 1u;  // the tile's largest group exceeds 8 lanes (same in every lane)
  const uint64_t bad = __builtin_amdgcn_ballot_w64((((uint32_t)(dmax + 900) > 1800u) & (dmax > kZeroish)) | (wide != 0));
  // (v_ldexp_f64 takes any int32 exponent: an exact zero's exponent, about -2^28, gives zero)
#ifdef NFST_P_CLAMP
  double M = ldexp_clamped(mt[0], d[0]);
#pragma unroll
  for (int j = 1; j < 4; ++j) M += ldexp_clamped(mt[j], d[j]);
#else
  double M = ldexp(mt[0], d[0]);
#pragma unroll
  for (int j = 1; j < 4; ++j) M += ldexp(mt[j], d[j]);
#endif
  // the three stages: partner's value by two v_mov_b32_dpp (every lane has a partner: the destination needs no initial
  // value), times the 0 / 1 multiplier whose high word the tile wave left in the slot
#ifdef NFST_P_BUILTINDPP
  M = fma(dpp_d<0xB1>(M), __hiloint2double((int)c.k0, 0), M);
  M = fma(dpp_d<0x4E>(M), __hiloint2double((int)c.k1, 0), M);
  M = fma(dpp_d<0x141>(M), __hiloint2double((int)c.k2, 0), M);
#else
  {
    int plo, phi;
#define NFST_DPP_PAIR(CTRL)                                                                                              \
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 " CTRL      \
                 " row_mask:0xf bank_mask:0xf" : "=&v"(plo), "=&v"(phi) : "v"(__double2loint(M)), "v"(__double2hiint(M)))
    NFST_DPP_PAIR("quad_perm:[1,0,3,2]");
    M = fma(__hiloint2double(phi, plo), __hiloint2double((int)c.k0, 0), M);
    NFST_DPP_PAIR("quad_perm:[2,3,0,1]");
    M = fma(__hiloint2double(phi, plo), __hiloint2double((int)c.k1, 0), M);
    NFST_DPP_PAIR("row_half_mirror");
    M = fma(__hiloint2double(phi, plo), __hiloint2double((int)c.k2, 0), M);
#undef NFST_DPP_PAIR
  }
#endif
  int E = ref;
  if (__builtin_expect(bad != 0, 0)) {
    const int g = (int)((c.dst >> 20) & 7u);
    M = ldexp_clamped(mt[0], d[0] - dmax);
#pragma unroll
    for (int j = 1; j < 4; ++j) M += ldexp_clamped(mt[j], d[j] - dmax);
    E = dmax + ref;
    const int Em = seg_max<6>(E, g);
    M = seg_sum64<6>(ldexp_clamped(M, E - Em), g);
    E = Em;
  }
  const Rec64 r = me_pack64(M, E);
  *(lds_v4u *)(uintptr_t)(c.dst & 0xfffffu) = v4u{(uint32_t)__double2loint(r.m), (uint32_t)__double2hiint(r.m), (uint32_t)r.e, 0u};
  return dmax;
}
__device__ __forceinline__ Rec64 rec64_load(uint32_t a) {
  const v4u x = *(const lds_v4u *)(uintptr_t)a;
  // (the pad word is dead, so hipcc reads 8 + 4 bytes in two instructions -- and that is the fast form: kept alive as one
  // ds_read_b128 per operand these random gathers cost 70 ns more per tile, 209 against 156 us on the SNIPS-shaped batch;
  // profiles/tune/ab_precise.sh)
#ifdef NFST_P_B128
  asm volatile("" ::"v"(x));
#endif
  Rec64 r;
  r.m = __hiloint2double((int)x.y, (int)x.x);
  r.e = (int)x.z;
  r.pad = 0;
  return r;
}
template <int NEF>
__device__ __forceinline__ void tile_sweep2p(int n_tiles, const uint32_t *ring, int R, int *prog, const int *land, int lane) {
  static_assert(NEF == 4, "four tile waves per sweep");
  if (n_tiles <= 0) return;
  constexpr uint32_t SB = kSlotWordsP * 4;
  const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
  const uint32_t land_a = lds_addr(land), prog_a = lds_addr(prog);
  // (as in tile_sweep2)
  auto flags_now = [&]() { return *(const volatile lds_v4u *)(uintptr_t)land_a; };
  // (every word of a copy is kept alive until the copy is looked at -- the empty asm statements below --: with dead words hipcc
  // reused their registers at once and put a full wait for the flag read, the youngest of nine LDS reads in flight, in front of
  // the tile's arithmetic)
  auto margin_x = [](const v4u f) { return min(min((int)f.y - 1, (int)f.z - 2), (int)f.w - 3); };
  auto margin_y = [](const v4u f) { return (int)f.x - 4; };
  auto wait_group = [&](int T, int which) {
    for (;;) {
      const v4u f = flags_now();
      if (__builtin_amdgcn_readfirstlane(which ? margin_y(f) : margin_x(f)) > T) break;
      __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
  };
  while (__builtin_amdgcn_readfirstlane((int)*(const volatile lds_u32 *)(uintptr_t)land_a) <= 0) __builtin_amdgcn_s_sleep(1);  // tile 0
  asm volatile("" ::: "memory");
  uint32_t gbase = ring_base;
  v4u margin = {0u, 0u, 0u, 0u}, margin1 = {0u, 0u, 0u, 0u};
  int ref = 0;
  int dmax0 = 0;
  Dec2P da, db;
  dec2p_fetch(gbase, lane, da);
  asm volatile("" ::: "memory");
#define NFST_S2P_STEP(K, CUR, NXT)                                                                         \
  {                                                                                                       \
    Rec64 vv[4];                                                                                          \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) vv[j] = rec64_load(CUR.opa[j]);                          \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    if (K == 3) {                                                                                         \
      gbase = (gbase + 4 * SB == ring_end) ? ring_base : gbase + 4 * SB;                                  \
      asm volatile("" ::"v"(margin1));                                                                    \
      if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_y(margin1)) <= T, 0)) wait_group(T, 1);     \
    }                                                                                                     \
    dec2p_fetch(gbase + (K == 3 ? 0u : (K + 1) * SB), lane, NXT);                                         \
    if (K == 1) margin1 = flags_now();                                                                     \
    if (K == 3) margin = flags_now();                                                        \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
    const int dm_ = tile_math3p(CUR, vv, ref);                                                            \
    if (K == 0) dmax0 = dm_;                                                                              \
    if (K & 1) *(volatile lds_u32 *)(uintptr_t)prog_a = (uint32_t)(T + K + 2);                            \
    asm volatile("" ::: "memory");                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                    \
  }
  int T = 0;
  for (; T + 4 <= n_tiles; T += 4) {
    asm volatile("" ::"v"(margin));
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_x(margin)) <= T, 0)) wait_group(T, 0);
    NFST_S2P_STEP(0, da, db)
    NFST_S2P_STEP(1, db, da)
    NFST_S2P_STEP(2, da, db)
    NFST_S2P_STEP(3, db, da)
    const int e0 = __builtin_amdgcn_readfirstlane(dmax0);
    ref = __builtin_amdgcn_readfirstlane((e0 > -(1 << 27)) ? e0 + ref : ref);
  }
  if (T < n_tiles) {
    asm volatile("" ::"v"(margin));
    if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin_x(margin)) <= T, 0)) wait_group(T, 0);
    NFST_S2P_STEP(0, da, db)
    if (T + 1 < n_tiles) {
      NFST_S2P_STEP(1, db, da)
      if (T + 2 < n_tiles) NFST_S2P_STEP(2, da, db)
    }
  }
#undef NFST_S2P_STEP
}

// a weight wave's part of a sweep: compact programs were started at kernel entry (x8), the rarer
// formats start here
template <int NE, int XM, bool FULL, bool PREC = false>
__device__ __forceinline__ void run_weights(WeightWave<8, NE, XM, FULL, PREC> &x8, int F, const uint32_t *prog_words, const int32_t *perm,
                                            int n_tiles, const Extra ex, int ei, uint32_t *ring, int R, int *flags,
                                            const float2 *th, const float2 *val, bool v2, uint32_t trash, uint32_t xc_base, int xc_first,
                                            int lane) {
  const int *prog = flags;
  int *xland = flags + 4;
  if constexpr (PREC) {
    x8.run(ring, R, prog, xland, th, val, v2, trash, xc_base, xc_first, lane);
    return;
  }
  if (F == 8 || FULL) {  // (tile waves run all-compact batches only)
    x8.run(ring, R, prog, xland, th, val, v2, trash, xc_base, xc_first, lane);
  } else if (F == 4) {
    WeightWave<4, NE, XM, FULL> x;
    x.start_maps(prog_words, perm, n_tiles, ex, ei, lane);
    x.start_gathers(lane);
    x.run(ring, R, prog, xland, th, val, v2, trash, xc_base, xc_first, lane);
  } else if (F == 2) {
    WeightWave<2, NE, XM, FULL> x;
    x.start_maps(prog_words, perm, n_tiles, ex, ei, lane);
    x.start_gathers(lane);
    x.run(ring, R, prog, xland, th, val, v2, trash, xc_base, xc_first, lane);
  } else {
    WeightWave<1, NE, XM, FULL> x;
    x.start_maps(prog_words, perm, n_tiles, ex, ei, lane);
    x.start_gathers(lane);
    x.run(ring, R, prog, xland, th, val, v2, trash, xc_base, xc_first, lane);
  }
}

// `first` = theta[tid] loaded by the caller at kernel entry (beside the meta record, not behind it)
__device__ __forceinline__ void load_theta(float2 *th, const float *theta, int64_t stride, int b,
                                           int V, int tid, int nt, float first) {
  const float *t = theta + (size_t)stride * b;
  for (int l = tid; l < V; l += nt) {
    ME x = exp_split(l == tid ? first : t[l]);
    th[l] = make_float2(x.m, __int_as_float(x.e));
  }
  if (tid == 0) {
    th[V] = make_float2(0.0f, __int_as_float(kEZero));  // the null label of empty slots
    th[V + 1] = make_float2(0.5f, __int_as_float(1));       // weight one: the carry record of a continuation piece
  }
}
__device__ __forceinline__ void load_theta(Rec64 *th, const float *theta, int64_t stride, int b,
                                           int V, int tid, int nt, float first) {
  const float *t = theta + (size_t)stride * b;
  for (int l = tid; l < V; l += nt) {
    const ME64 x = exp_split64((double)(l == tid ? first : t[l]));
    th[l].m = x.m; th[l].e = x.e; th[l].pad = 0;
  }
  if (tid == 0) {
    th[V].m = 0.0; th[V].e = kEZero; th[V].pad = 0;     // the null label of empty slots
    th[V + 1].m = 0.5; th[V + 1].e = 1; th[V + 1].pad = 0;  // weight one: the carry record of a continuation piece
  }
}
