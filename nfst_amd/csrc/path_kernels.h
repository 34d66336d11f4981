// path_kernels.h -- Viterbi, posterior sampling, path scoring, per-step gathers, sequence scoring, IWAE
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
#pragma once

// ------------------------------------------------------------------ Viterbi
// max-plus run of the by-source tile program by one wave (float32 values, back pointers
// -- canonical arc and next state -- in LDS; the program is read straight from global
// memory), then lane 0 walks the best path inside LDS.  Ties keep the arc with the smallest
// canonical id, i.e. the smallest label.
__device__ __forceinline__ void vit_take(float &bv, int &ba, int &bn, float ov, int oa, int on, bool allowed = true) {
  const bool t = allowed & ((ov > bv) | ((ov == bv) & (oa < ba)));  // bitwise on purpose: selects, no branches
  bv = t ? ov : bv;
  ba = t ? oa : ba;
  bn = t ? on : bn;  // the arc's other end
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
  // The words of tile T+1 are loaded while tile T is computed: loads only, nothing is unpacked
  // before the tile's turn (a use would make the wave wait for the L2 round trip right away), and
  // the format is a compile-time constant of the loop (a branch around loads ends in a full wait).
  struct VitTile { uint4 x; uint32_t w[4]; int cas[4]; };
  auto sweep = [&](auto compact_tag, auto extra_tag) {
  constexpr bool kCompact = decltype(compact_tag)::value, kExtra = decltype(extra_tag)::value;
  auto load_tile = [&](int T, VitTile &t) {
    if (kCompact) {  // control word + four 24-bit records per lane
      t.x = *reinterpret_cast<const uint4 *>(prog + (size_t)T * ST + lane * 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) t.cas[j] = perm[(size_t)T * 256 + lane * 4 + j];
      return;
    }
    t.x.x = prog[(size_t)T * ST + lane];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int jj = min(j, U - 1);
      t.cas[j] = perm[(size_t)T * 64 * U + lane * U + jj];
      t.w[j] = prog[(size_t)T * ST + 64 + lane * U + jj];
    }
  };
  // iteration T: `cur` = tile T, `nxt` receives tile T+1
  auto step = [&](int T, const VitTile &cur, VitTile &nxt) {
    load_tile(min(T + 1, m.bwd_tiles - 1), nxt);
    const uint32_t ctl = cur.x.x;
    int cas[4];
    uint32_t rcs[4];
    float xs[4];
    if (kCompact) {
      const uint4 x = cur.x;
      const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
      for (int j = 0; j < 4; ++j) rcs[j] = ((r[j] & 0x1fffu) << 3) | (((r[j] >> 13) & 0x7ffu) << 16);  // as a 32-bit record
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) rcs[j] = cur.w[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) cas[j] = (j < U) ? cur.cas[j] : -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xs[j] = 0.0f;
      if (kExtra && cas[j] >= 0) {  // (a loop without extras has no load but the prefetch)
        if (arc_w) xs[j] += arc_w[cas[j]];
        if (sc.arc_scores) xs[j] += sc.arc_scores[cas[j]];
      }
    }
    float bv = kNegInf;
    int ba = kNone, bn = -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // straight-line: every slot reads its label score and operand value (an empty slot has
      // operand 0 and the null label, whose "score" is the word behind the table -- never used)
      const int other = (int)((rcs[j] & 0xffffu) >> 3);
      xs[j] += tl[rcs[j] >> 16] + v[other];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) vit_take(bv, ba, bn, xs[j], cas[j], (int)((rcs[j] & 0xffffu) >> 3), cas[j] >= 0);
    // a unit-label record stands for what row `other` holds -- the state's own earlier pieces
    // (carry) or a scratch row of a partial group: its best arc competes as such.  Rare.
    bool unit[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) unit[j] = (j < U) & (cas[j] < 0) & ((int)(rcs[j] >> 16) == lat.vocab + 1);
    if (__builtin_amdgcn_ballot_w64(unit[0] | unit[1] | unit[2] | unit[3])) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int other = (int)((rcs[j] & 0xffffu) >> 3);
        if (unit[j] && bp[other] >= 0) vit_take(bv, ba, bn, v[other], bp[other], ns[other]);
      }
    }
    const int gl = (int)((ctl >> 20) & 7u);
    const int gmax = (int)((__builtin_amdgcn_readfirstlane(ctl) >> 23) & 7u);
    // segmented max over the state's lanes: quad permutes and row mirrors (DPP), then the
    // two cross-row stages; every lane of a state ends with the same (value, arc, next state)
#define NFST_VIT_STAGE(ST, FV, FI)                                          \
    if (gmax > ST) {                                                        \
      const float ov = FV(bv);                                              \
      const int oa = FI(ba), on = FI(bn);                                   \
      vit_take(bv, ba, bn, ov, oa, on, gl > ST);                            \
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
  };
  VitTile ta, tb;
  if (m.bwd_tiles > 0) load_tile(0, ta);
  // two iterations per trip so that the register roles alternate without copies (a copy of
  // registers with a load in flight would wait for it)
  for (int T = 0; T < m.bwd_tiles; T += 2) {
    step(T, ta, tb);
    if (T + 1 >= m.bwd_tiles) break;
    step(T + 1, tb, ta);
  }
  };
  const bool extras = arc_w != nullptr || sc.arc_scores != nullptr;
  if (F == 8) { if (extras) sweep(std::true_type{}, std::true_type{}); else sweep(std::true_type{}, std::false_type{}); }
  else { if (extras) sweep(std::false_type{}, std::true_type{}); else sweep(std::false_type{}, std::false_type{}); }
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

// ------------------------------------------------------------------ Viterbi on tile waves
// The same max-plus sweep on the tile-wave pipeline of the sum-product kernels (tile_pipeline.h; all-compact
// batches): wave 0 sweeps, waves 1, 2, 3, 5 decode every fourth tile of the by-source program straight from HBM
// into a ring of decoded tiles in LDS -- per lane the store address (a leader's state record, 16 bytes of trash
// otherwise), the group size / largest group of the tile / which slots are unit-label records, four operand
// record addresses, four arc scores (label score + per-arc extras; -inf for an empty slot, 0 for a unit record)
// and the four canonical arcs.  A state's record in LDS is 16 bytes: best value, best arc, the record address
// of the arc's other end (the trace-back is an LDS walk), so a leader stores one ds_write_b128.
constexpr int kVitTwThreads = 512, kVitNE = 4;
struct VitLds {
  int rows, vocab;
  __host__ __device__ VitLds(int r, int v) : rows(r), vocab(v) {}
  // 16 B records | float label scores [V + 2] (null label: -inf, unit label: 0) | ring | flags | 64 x 16 B of trash
  __host__ __device__ int64_t fixed() const { return (int64_t)rows * 16 + (int64_t)((vocab + 2 + 3) & ~3) * 4 + kSweepFlags * 4 + 1024; }
};

template <int XM>  // arrays of per-arc extras: 0, 1, 2
struct VitWave {
  struct P { int a[4]; uint32_t raw[4]; };
  struct G { float w[XM != 0 ? 4 : 1], s[XM == 2 ? 4 : 1]; int a[4]; uint32_t raw[4]; };
  P p0, p1, p2;
  G g0, g1, g2;
  const float *bw, *bs;
  const int32_t *perm;
  const uint32_t *prog_words;
  int n_mine, ei;
  __device__ __forceinline__ P ld_tile(int i, int lane) const {
    P p;
    const int t = ei + min(i, max(n_mine - 1, 0)) * kVitNE;  // past this wave's last tile: that tile again
    const int4 v = *reinterpret_cast<const int4 *>(perm + (size_t)t * 256 + lane * 4);
    p.a[0] = v.x; p.a[1] = v.y; p.a[2] = v.z; p.a[3] = v.w;
    const uint4 x = *reinterpret_cast<const uint4 *>(prog_words + (size_t)t * 256 + lane * 4);
    p.raw[0] = x.x; p.raw[1] = x.y; p.raw[2] = x.z; p.raw[3] = x.w;
    return p;
  }
  __device__ __forceinline__ G issue(const P &p) const {
    G g;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int a = max(p.a[j], 0);
      if (XM != 0) g.w[j] = bw[a];
      if (XM == 2) g.s[j] = bs[a];
      g.a[j] = p.a[j];
      g.raw[j] = p.raw[j];
    }
    return g;
  }
  __device__ __forceinline__ void start(const uint32_t *prog_, const int32_t *perm_, int n_tiles, const Extra ex, int ei_, int lane) {
    perm = perm_; prog_words = prog_; ei = ei_;
    n_mine = n_tiles > ei ? (n_tiles - ei + kVitNE - 1) / kVitNE : 0;
    bw = ex.arc_w ? ex.arc_w : ex.arc_scores;
    bs = ex.arc_scores;
    if (n_mine <= 0) return;
    p0 = ld_tile(0, lane); p1 = ld_tile(1, lane); p2 = ld_tile(2, lane);
    g0 = issue(p0);
    p0 = ld_tile(3, lane);
    g1 = issue(p1);
  }
  __device__ __forceinline__ void run(uint32_t *ring, int R, const int *prog, int *xland, const float *tl, uint32_t rec_base,
                                      uint32_t trash, int vocab, int lane) {
    const uint32_t xl_a = lds_addr(xland) + (uint32_t)ei * 4;
    if (n_mine <= 0) {
      *(volatile lds_u32 *)(uintptr_t)xl_a = 0x7fffffffu;
      return;
    }
    constexpr uint32_t SB = kSlotWords2 * 4;
    const uint32_t ring_base = lds_addr(ring), prog_a = lds_addr(prog), tl_base = lds_addr(tl);
    int prog_seen = 0;
    auto process = [&](int i, const G &g) {
      const int t = ei + min(i, n_mine - 1) * kVitNE;
      const uint32_t ctl = g.raw[0];
      const uint32_t r[4] = {g.raw[1], __builtin_amdgcn_alignbit(g.raw[2], g.raw[1], 24), __builtin_amdgcn_alignbit(g.raw[3], g.raw[2], 16),
                             g.raw[3] >> 8};
      uint32_t opa[4], unit = 0;
      float sc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t lab = (r[j] >> 13) & 0x7ffu;
        opa[j] = ((r[j] & 0x1fffu) << 4) + rec_base;  // the operand state's record
        sc[j] = *(const __attribute__((address_space(3))) float *)(uintptr_t)(tl_base + lab * 4);  // (null and unit label: -inf)
        if (XM != 0 && g.a[j] >= 0) sc[j] += XM == 2 ? g.w[j] + g.s[XM == 2 ? j : 0] : g.w[j];
        unit |= (((g.a[j] < 0) & ((int)lab == vocab + 1)) ? 1u : 0u) << j;
      }
      const bool any_unit = __builtin_amdgcn_ballot_w64(unit != 0) != 0;
      while (__builtin_expect(prog_seen < t - R + 1, 0)) {  // the slot's previous tile is consumed
        prog_seen = __builtin_amdgcn_readfirstlane(*(const volatile lds_u32 *)(uintptr_t)prog_a);
        if (prog_seen < t - R + 1) __builtin_amdgcn_s_sleep(NFST_X_NAP);
      }
      asm volatile("" ::: "memory");
      const uint32_t sb = ring_base + (uint32_t)(t % R) * SB;
      const uint32_t dst = ((int)ctl < 0) ? ((ctl & 0xffffu) << 1) + rec_base : trash;  // 8 x state -> 16 x state
      // word 0: store address (18 bits) | group size g (3) | largest g of the tile (3) | unit slots (4) | the tile has unit records (1);
      // words 1 .. 3: all-ones where the state owns more than 2^s lanes (stage s of the segmented maximum)
      const uint32_t gl = (ctl >> 20) & 7u;
      const uint32_t w0 = dst | (gl << 18) | (((ctl >> 23) & 7u) << 21) | (unit << 24) | ((any_unit ? 1u : 0u) << 28);
      *(lds_v4u *)(uintptr_t)(sb + lane * 16) = v4u{w0, gl > 0 ? ~0u : 0u, gl > 1 ? ~0u : 0u, gl > 2 ? ~0u : 0u};
      *(lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16) = v4u{opa[0], opa[1], opa[2], opa[3]};
      *(lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16) = v4f{sc[0], sc[1], sc[2], sc[3]};
      // (the complement of the canonical arc: the low half of a 64-bit key whose maximum prefers the smaller arc)
      *(lds_v4u *)(uintptr_t)(sb + 3072 + lane * 16) = v4u{~(uint32_t)g.a[0], ~(uint32_t)g.a[1], ~(uint32_t)g.a[2], ~(uint32_t)g.a[3]};
      asm volatile("" ::: "memory");
      *(volatile lds_u32 *)(uintptr_t)xl_a = (uint32_t)(t + 1);
    };
#define NFST_V_STEP(PU, GN, PL, GC)  \
    GN = issue(PU);                  \
    PL = ld_tile(i + 4, lane);       \
    process(i, GC);
    for (int i = 0; i < n_mine; i += 3) {
      NFST_V_STEP(p2, g2, p1, g0) ++i;
      NFST_V_STEP(p0, g0, p2, g1) ++i;
      NFST_V_STEP(p1, g1, p0, g2) i -= 2;
    }
#undef NFST_V_STEP
    *(volatile lds_u32 *)(uintptr_t)xl_a = 0x7fffffffu;
  }
};

struct VitDec {
  uint32_t w0, m0, m1, m2;
  uint32_t opa[4];
  float sc[4];
  uint32_t narc[4];  // ~arc
};
__device__ __forceinline__ void vit_fetch(uint32_t sb, int lane, VitDec &d) {
  const v4u h = *(const lds_v4u *)(uintptr_t)(sb + lane * 16);
  const v4u a = *(const lds_v4u *)(uintptr_t)(sb + 1024 + lane * 16);
  const v4f s = *(const lds_v4f *)(uintptr_t)(sb + 2048 + lane * 16);
  const v4u c = *(const lds_v4u *)(uintptr_t)(sb + 3072 + lane * 16);
  d.w0 = h.x; d.m0 = h.y; d.m1 = h.z; d.m2 = h.w;
  d.opa[0] = a.x; d.opa[1] = a.y; d.opa[2] = a.z; d.opa[3] = a.w;
  d.sc[0] = s.x; d.sc[1] = s.y; d.sc[2] = s.z; d.sc[3] = s.w;
  d.narc[0] = c.x; d.narc[1] = c.y; d.narc[2] = c.z; d.narc[3] = c.w;
}

template <int XM>
__global__ __launch_bounds__(kVitTwThreads) void k_viterbi_tw(nfst_batch lat, nfst_scores sc, int R, float *best, int32_t *paths,
                                                              int32_t *path_arcs, int32_t *lengths, int max_len, int pad) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Meta m = load_meta(lat.meta, b);
  const int tlw = (lat.vocab + 2 + 3) & ~3;
  v4u *rec = (v4u *)lds;                       // per state: value bits, best arc, record address of its other end, -
  float *tl = (float *)(rec + lat.max_rows);   // label scores
  uint32_t *ring = (uint32_t *)(tl + tlw);
  int *flags = (int *)(ring + (size_t)R * kSlotWords2);
  const uint32_t rec_base = lds_addr(rec), trash = lds_addr(flags + kSweepFlags) + lane * 16;
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const bool x_wave = wv == 1 || wv == 2 || wv == 3 || wv == 5;
  VitWave<XM> xw;
  if (x_wave) xw.start(lat.bwd_stream + m.bwd_off, lat.bwd_perm + m.bwd_slot_off, m.bwd_tiles, ex, wv < 4 ? wv - 1 : 3, lane);
  const float *tg = sc.theta + (size_t)sc.theta_stride * b;
  for (int i = tid; i < lat.max_rows; i += kVitTwThreads) rec[i] = v4u{__float_as_uint(kNegInf), 0xffffffffu, 0u, 0u};  // incl. scratch rows
  for (int i = tid; i < lat.vocab; i += kVitTwThreads) tl[i] = tg[i];
  if (tid == 0) {
    tl[lat.vocab] = kNegInf;    // the null label of empty slots
    tl[lat.vocab + 1] = kNegInf;  // the unit label of carry / combine records: they compete through their operand's record
    for (int i = 0; i < kSweepFlags; ++i) flags[i] = 0;
  }
  __syncthreads();
  if (tid == 0) rec[m.sink].x = __float_as_uint(0.0f);
  __syncthreads();
  if (x_wave) {
    xw.run(ring, R, flags, flags + 4, tl, rec_base, trash, lat.vocab, lane);
  } else if (wv == 0 && m.bwd_tiles > 0) {
    __builtin_amdgcn_s_setprio(3);
    constexpr uint32_t SB = kSlotWords2 * 4;
    const uint32_t ring_base = lds_addr(ring), ring_end = ring_base + (uint32_t)R * SB;
    const uint32_t land_a = lds_addr(flags + 4), prog_a = lds_addr(flags);
    auto group_margin = [&]() {
      const v4u f = *(const volatile lds_v4u *)(uintptr_t)land_a;
      return min(min((int)f.y - 1, (int)f.z - 2), min((int)f.w - 3, (int)f.x - 4));
    };
    while (__builtin_amdgcn_readfirstlane((int)*(const volatile lds_u32 *)(uintptr_t)land_a) <= 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    uint32_t gbase = ring_base;
    int margin = 0;
    VitDec da, db;
    vit_fetch(gbase, lane, da);
    asm volatile("" ::: "memory");
    // tile T+K: `cur` in registers; fetches tile T+K+1 into `nxt`
    auto step = [&](auto wide_tag, int T, int K, const VitDec &cur, VitDec &nxt) {
      constexpr bool WIDE = decltype(wide_tag)::value;
      const uint32_t cu = (uint32_t)__builtin_amdgcn_readfirstlane((int)cur.w0);  // the tile-uniform bits, read before they are needed
      float xs[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xs[j] = *(const __attribute__((address_space(3))) float *)(uintptr_t)cur.opa[j];
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (K == 3) gbase = (gbase + 4 * SB == ring_end) ? ring_base : gbase + 4 * SB;
      vit_fetch(gbase + (K == 3 ? 0u : (uint32_t)(K + 1) * SB), lane, nxt);
      if (K == 3) margin = group_margin();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // (value, arc) as one 64-bit key -- order-preserving bits of the float above the complement of the arc -- so
      // that "greater value, or equal value and smaller arc" is ONE vector compare feeding selects (the boolean
      // form goes through scalar mask arithmetic: ~25 cycles per use of a fresh vector result)
      auto ord = [](float x) { const uint32_t u = __float_as_uint(x); return u ^ (uint32_t)(((int)u >> 31) | (int)0x80000000); };
      auto key = [](uint32_t hi, uint32_t lo) { return ((uint64_t)hi << 32) | lo; };
      uint32_t bh = ord(xs[0] + cur.sc[0]), bl = cur.narc[0], bn = cur.opa[0];
#pragma unroll
      for (int j = 1; j < 4; ++j) {
        const uint32_t oh = ord(xs[j] + cur.sc[j]);
        const bool t = key(oh, cur.narc[j]) > key(bh, bl);
        bh = t ? oh : bh; bl = t ? cur.narc[j] : bl; bn = t ? cur.opa[j] : bn;
      }
      // a unit-label record stands for what its operand row holds -- the state's own earlier pieces (carry) or a
      // scratch row of a partial group: that row's best arc competes as such.  Rare (tile-uniform flag).
      if (__builtin_expect((cu >> 28) & 1u, 0)) {
        const uint32_t unit = (cur.w0 >> 24) & 0xfu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if ((unit >> j) & 1u) {
            const v4u o = *(const lds_v4u *)(uintptr_t)cur.opa[j];
            const uint32_t oh = ord(__uint_as_float(o.x));
            if ((int)o.y >= 0 && key(oh, ~o.y) > key(bh, bl)) { bh = oh; bl = ~o.y; bn = o.z; }
          }
        }
      }
      // segmented maximum over the state's lanes: a partner outside the state's group is masked to the smallest key
#define NFST_VIT_STAGE2(MASK, FI)                                            \
      {                                                                     \
        const uint32_t oh = (uint32_t)FI((int)bh) & (MASK), ol = (uint32_t)FI((int)bl), on = (uint32_t)FI((int)bn); \
        const bool t = key(oh, ol) > key(bh, bl);                           \
        bh = t ? oh : bh; bl = t ? ol : bl; bn = t ? on : bn;               \
      }
      NFST_VIT_STAGE2(cur.m0, dpp_i<0xB1>)
      NFST_VIT_STAGE2(cur.m1, dpp_i<0x4E>)
      NFST_VIT_STAGE2(cur.m2, dpp_i<0x141>)
      if (WIDE) {
        const int gl = (int)((cur.w0 >> 18) & 7u), gmax = (int)((cu >> 21) & 7u);
#define NFST_SHFL16(x) __shfl_xor(x, 16)
#define NFST_SHFL32(x) __shfl_xor(x, 32)
        if (gmax > 3) NFST_VIT_STAGE2(gl > 3 ? ~0u : 0u, dpp_i<0x140>)
        if (gmax > 4) NFST_VIT_STAGE2(gl > 4 ? ~0u : 0u, NFST_SHFL16)
        if (gmax > 5) NFST_VIT_STAGE2(gl > 5 ? ~0u : 0u, NFST_SHFL32)
#undef NFST_SHFL16
#undef NFST_SHFL32
      }
#undef NFST_VIT_STAGE2
      const uint32_t vb = bh ^ (uint32_t)((~(int)bh >> 31) | (int)0x80000000);  // back to float bits
      *(lds_v4u *)(uintptr_t)(cur.w0 & 0x3ffffu) = v4u{vb, ~bl, bn, 0u};
      if (K & 1) *(volatile lds_u32 *)(uintptr_t)prog_a = (uint32_t)(T + K + 2);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
    auto wait_group = [&](int T) {
      while (__builtin_amdgcn_readfirstlane(group_margin()) <= T) __builtin_amdgcn_s_sleep(1);
      asm volatile("" ::: "memory");
    };
    auto sweep = [&](auto wide_tag) {
      for (int T = 0; T < m.bwd_tiles; T += 4) {
        if (__builtin_expect(__builtin_amdgcn_readfirstlane(margin) <= T, 0)) wait_group(T);
        step(wide_tag, T, 0, da, db); if (T + 1 >= m.bwd_tiles) break;
        step(wide_tag, T, 1, db, da); if (T + 2 >= m.bwd_tiles) break;
        step(wide_tag, T, 2, da, db); if (T + 3 >= m.bwd_tiles) break;
        step(wide_tag, T, 3, db, da);
      }
    };
    if (m.bwd_wide) sweep(std::true_type{}); else sweep(std::false_type{});
    __builtin_amdgcn_s_setprio(0);
  }
  __syncthreads();
  // lane 0 walks the back pointers inside LDS (arc ids go to the list `pa` in the ring, which is free now);
  // all threads then write the labels
  int *pa = (int *)ring;
  int *res = flags;  // [0] length, [1] reached the sink
  if (tid == 0) {
    best[b] = __uint_as_float(rec[0].x);
    uint32_t at = rec_base;
    const uint32_t sink_at = rec_base + (uint32_t)m.sink * 16;
    int len = 0;
    const int cap = min(min(max_len, m.n_rows), R * kSlotWords2);
    while (at != sink_at && len < cap) {
      const v4u o = *(const lds_v4u *)(uintptr_t)at;
      if ((int)o.y < 0) break;
      pa[len++] = (int)o.y;
      at = o.z;
    }
    res[0] = len;
    res[1] = (at == sink_at) ? 1 : 0;
  }
  __syncthreads();
  const int len = res[0];
  if (tid == 0) lengths[b] = res[1] ? len : -1;
  for (int j = tid; j < max_len; j += kVitTwThreads) {
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
// Philox4x32-10 block keyed by the seed, counter (walk, step / 4): four uniforms, one per step
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint32_t walk, uint32_t block, float (&u)[4]) {
  uint32_t c0 = walk, c1 = block, c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0, hi1;
    const uint32_t lo0 = mulhilo(0xD2511F53u, c0, &hi0);
    const uint32_t lo1 = mulhilo(0xCD9E8D57u, c2, &hi1);
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  u[0] = (float)(c0 >> 8) * (1.0f / 16777216.0f);
  u[1] = (float)(c1 >> 8) * (1.0f / 16777216.0f);
  u[2] = (float)(c2 >> 8) * (1.0f / 16777216.0f);
  u[3] = (float)(c3 >> 8) * (1.0f / 16777216.0f);
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

template <int SHIFT>
__device__ __forceinline__ float row_shr_zero(float v) {  // lane i gets lane i-SHIFT of its row, 0 if there is none
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + SHIFT, 0xf, 0xf, true));
}
__global__ __launch_bounds__(1024) void k_sample(nfst_batch lat, nfst_scores sc,
                                                           const float2 *beta_me, const double *logz64, int K,
                                                           int max_len, const float *uniforms, uint64_t seed,
                                                           int pad, int stage_theta, int lds_bytes, int32_t *paths,
                                                           int32_t *path_arcs, int32_t *lengths, float *logq, int32_t *status) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int r = tid & 15;                                // lane within the row
  const int nt = (int)blockDim.x;                        // 256 .. 1024 threads: 16 .. 64 walks of one lattice share the staged data
  const int k = blockIdx.y * (nt >> 4) + (tid >> 4);     // this row's walk
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  float2 *bl = lds;                                      // beta (m, e) of the lattice's rows
  float *tls = (float *)(bl + lat.max_rows);             // label scores (staged unless the vocabulary is huge)
  for (int i = tid; i < m.n_rows; i += nt) bl[i] = beta_me[m.row_off + i];
  if (stage_theta) for (int i = tid; i < lat.vocab; i += nt) tls[i] = theta[i];
  // Round 2: when the lattice's CSR (row pointers, src | dst << 16, 16-bit labels: 6 bytes per arc) fits the LDS the
  // launch was given, it is staged there: a step's dependent reads (row pointers -> arcs -> beta of their ends) are then
  // LDS round trips instead of L2 / HBM ones (BASELINE batch: k_sample 105 -> 88 us, rocprofv3)
  int *rps = (int *)(tls + (stage_theta ? ((lat.vocab + 3) & ~3) : 0));
  uint32_t *sds = (uint32_t *)(rps + ((m.n_rows + 1 + 3) & ~3));
  uint16_t *lbs = (uint16_t *)(sds + ((m.n_arcs + 3) & ~3));
  const bool staged = (char *)(lbs + m.n_arcs) - (char *)lds <= (ptrdiff_t)lds_bytes;
  // ... and when the caller takes the arcs of the paths (path_arcs), the same 6 bytes per arc hold the arcs' CUMULATIVE
  // probabilities within their state (float) and 16-bit destinations instead: every arc's probability is computed once per
  // block, arc-parallel, and a step is row pointers -> one compare of u against up to 16 cumulative values -> next state; the
  // labels and the path score are collected in a second pass over the chosen arcs (k_sample 88 -> 48 us at 16 walks per
  // lattice, 57 us at 64; rocprofv3)
  const bool precdf = staged && path_arcs != nullptr;
  if (staged) {
    for (int i = tid; i <= m.n_rows; i += nt) rps[i] = rp[i] - m.arc_off;
    if (!precdf)
      for (int i = tid; i < m.n_arcs; i += nt) { sds[i] = lat.arc_sd[m.arc_off + i]; lbs[i] = lat.arc_l16[m.arc_off + i]; }
  }
  __syncthreads();
  const float *tl = stage_theta ? (const float *)tls : theta;
  if (precdf) {
    float *cdf = (float *)sds;
    uint16_t *d16 = lbs;
    // the probability of every arc given its source state: w beta[dst] / beta[src]  (four arcs' loads in flight per thread)
    for (int i0 = tid; i0 < m.n_arcs; i0 += 4 * nt) {
      uint32_t sd4[4];
      float x4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int a = m.arc_off + min(i0 + q * nt, m.n_arcs - 1);
        sd4[q] = lat.arc_sd[a];
        x4[q] = tl[lat.arc_l16[a]];
        if (arc_w) x4[q] += arc_w[a];
        if (sc.arc_scores) x4[q] += sc.arc_scores[a];
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = i0 + q * nt;
        const int s0 = (int)(sd4[q] & 0xffffu), d0 = (int)(sd4[q] >> 16);
        const ME wgt = exp_split(x4[q]);
        const float2 bd = bl[d0], bs = bl[s0];
        const float p = ldexpf((wgt.m * bd.x) * (1.0f / bs.x), max(wgt.e + __float_as_int(bd.y) - __float_as_int(bs.y), -300));
        if (i < m.n_arcs) {
          cdf[i] = d0 != s0 ? p : 0.0f;
          d16[i] = (uint16_t)d0;
        }
      }
    }
    __syncthreads();
    for (int s0 = tid; s0 < m.n_rows; s0 += nt) {  // running sums within each state (its arcs are consecutive)
      float run = 0.0f;
      for (int i = rps[s0]; i < rps[s0 + 1]; ++i) { run += cdf[i]; cdf[i] = run; }
    }
    __syncthreads();
    const bool live = k < K;
    const size_t walk = (size_t)b * K + (live ? k : 0);
    int32_t *out = paths + walk * max_len, *outa = path_arcs + walk * max_len;
    int s = 0, t = 0;
    float ublk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    bool ok = true, active = live;
    while (true) {
      if (active && s == m.sink) active = false;
      if (active && t >= max_len) { ok = false; active = false; }
      if (!__any(active)) break;
      float u = 0.0f;
      int a0 = 0, a1 = 0;
      if (!uniforms) {
        // the same numbers as one block per four steps, off the walk's chain: every 64 steps lane r of the walk's 16
        // takes Philox block t / 4 + r, and a step fetches its uniform from the lane that holds it
        if ((t & 63) == 0) philox_uniform4(seed, (uint32_t)walk, (uint32_t)((t >> 2) + r), ublk);
        const float mine = (t & 3) == 0 ? ublk[0] : ((t & 3) == 1 ? ublk[1] : ((t & 3) == 2 ? ublk[2] : ublk[3]));
        u = __shfl(mine, (t >> 2) & 15, 16);
      }
      if (active) {
        if (uniforms) u = uniforms[walk * max_len + t];
        a0 = rps[s];
        a1 = rps[s + 1];
      }
      // The usual step, without ballots or shuffles: the cumulative values of a state are non-decreasing, so the arc
      // to take is the one lane of the row with prev <= u < cum (prev: the lane before, DPP); its (index, destination)
      // reaches the whole row as a row maximum (four DPP steps).  Rows that find none -- more than 16 arcs, or u beyond
      // the last cumulative value by rounding -- take the chunk loop below.
      const int i0 = min(a0 + r, max(m.n_arcs - 1, 0));
      const bool valid0 = active & (a0 + r < a1);
      const float cum0 = valid0 ? cdf[i0] : 0.0f;
      const int dv0 = (int)d16[i0];
      const float prev0 = row_shr_zero<1>(cum0);
      int key = (valid0 && u < cum0 && !(u < prev0)) ? (((r << 16) | dv0) + 1) : 0;
      key = max(key, dpp_i<0xB1>(key));
      key = max(key, dpp_i<0x4E>(key));
      key = max(key, dpp_i<0x141>(key));
      key = max(key, dpp_i<0x140>(key));
      float cum_base = 0.0f;
      int chosen = key ? a0 + ((key - 1) >> 16) : -1, last = -1, d_ch = (key - 1) & 0xffff, d_last = 0;
      bool more = active & (a0 < a1) & (key == 0);
      for (int c = a0; __any(more); c += 16) {
        const int i = c + r;
        const bool valid = more & (i < a1);
        const float cum = valid ? cdf[i] : 0.0f;
        const int d = valid ? (int)d16[i] : 0;
        const float shr = row_shr_zero<1>(cum);
        const float prev = (r == 0) ? cum_base : shr;
        const bool positive = valid & (cum > prev);  // the arc has a non-zero probability
        const int sh = (int)(threadIdx.x & 48);
        const uint32_t hit = (uint32_t)(__ballot(positive && u < cum) >> sh) & 0xffffu;
        const uint32_t pos = (uint32_t)(__ballot(positive) >> sh) & 0xffffu;
        const int f = hit ? __builtin_ctz(hit) : 0, l = pos ? 31 - __builtin_clz(pos) : 0;
        const float c_end = __shfl(cum, 15, 16);
        const int d_f = __shfl(d, f, 16), d_l = __shfl(d, l, 16);
        const bool take = more & (hit != 0), keep = more & (hit == 0), seen = keep & (pos != 0);
        chosen = take ? c + f : chosen;
        d_ch = take ? d_f : d_ch;
        cum_base = keep ? c_end : cum_base;
        last = seen ? c + l : last;
        d_last = seen ? d_l : d_last;
        more = more & (c + 16 < a1) & (chosen < 0);
      }
      if (active) {
        if (chosen < 0) { chosen = last; d_ch = d_last; }
        if (chosen < 0) { ok = false; active = false; }
        else {
          if (r == 0) outa[t] = chosen + m.arc_off;
          s = d_ch;
          ++t;
        }
      }
    }
    if (!live) return;
    if (!ok && r == 0) atomicExch(status, NFST_ERR_LENGTH);
    // second pass, 16 lanes over the walk's arcs: labels out, path score summed
    // (the arcs were stored by lane 0 of this same 16-lane group: a workgroup-scope fence is all the order needed -- an
    // agent-scope one writes the XCD's L2 back, once per wave: 100 us)
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    // (in float64: a path of a thousand arcs sums to several thousand, log q is what is left after log Z is taken off)
    double part = 0.0;
    for (int j = r; j < t; j += 16) {
      const int a = __hip_atomic_load(outa + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int lab = lat.arc_l16[a];
      part += (double)tl[lab];
      if (arc_w) part += (double)arc_w[a];
      if (sc.arc_scores) part += (double)sc.arc_scores[a];
      out[j] = lab;
    }
    part += __shfl_xor(part, 8, 16);
    part += __shfl_xor(part, 4, 16);
    part += __shfl_xor(part, 2, 16);
    part += __shfl_xor(part, 1, 16);
    if (r == 0) {
      lengths[walk] = ok ? t : -1;
      logq[walk] = ok ? (float)(part - logz64[b]) : kNegInf;
    }
    for (int j = t + r; j < max_len; j += 16) { out[j] = pad; outa[j] = -1; }
    return;
  }
  const bool live = k < K;
  const size_t walk = (size_t)b * K + (live ? k : 0);
  int32_t *out = paths + walk * max_len;
  int32_t *outa = path_arcs ? path_arcs + walk * max_len : nullptr;
  int s = 0, t = 0;
  double tot = 0.0;  // (float64: see the second pass of the precomputed mode)
  float ublk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
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
      if (uniforms) {
        u = uniforms[walk * max_len + t];
      } else {  // one Philox block serves four steps
        if ((t & 3) == 0) philox_uniform4(seed, (uint32_t)walk, (uint32_t)(t >> 2), ublk);
        u = (t & 3) == 0 ? ublk[0] : ((t & 3) == 1 ? ublk[1] : ((t & 3) == 2 ? ublk[2] : ublk[3]));
      }
      bs = bl[s];
      a0 = staged ? rps[s] + m.arc_off : rp[s];
      a1 = staged ? rps[s + 1] + m.arc_off : rp[s + 1];
    }
    const float rs = 1.0f / bs.x;
    const int es = __float_as_int(bs.y);
    float cum_base = 0.0f, sc_ch = 0.0f, sc_last = 0.0f;
    int chosen = -1, last = -1, d_ch = 0, d_last = 0;
    bool more = active & (a0 < a1);
    for (int c = a0; __any(more); c += 16) {  // `more` is updated at the end: no empty last round
      const int a = c + r;
      float p = 0.0f, x = 0.0f;
      int d = 0;
      if (more && a < a1) {
        const uint32_t sd = staged ? sds[a - m.arc_off] : lat.arc_sd[a];
        d = (int)(sd >> 16);
        if (d != s) {
          x = tl[staged ? lbs[a - m.arc_off] : lat.arc_l16[a]];
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
      // selects, no branches (hipcc turns && / || and small ifs into execution-mask branches)
      const bool take = more & (hit != 0), keep = more & (hit == 0), seen = keep & (pos != 0);
      chosen = take ? c + f : chosen;
      sc_ch = take ? x_f : sc_ch;
      d_ch = take ? d_f : d_ch;
      cum_base = keep ? c_end : cum_base;
      last = seen ? c + l : last;
      sc_last = seen ? x_l : sc_last;
      d_last = seen ? d_l : d_last;
      more = more & (c + 16 < a1) & (chosen < 0);
    }
    if (active) {
      if (chosen < 0) { chosen = last; sc_ch = sc_last; d_ch = d_last; }
      if (chosen < 0) { ok = false; active = false; }
      else {
        if (r == 0) {
          out[t] = staged ? lbs[chosen - m.arc_off] : lat.arc_l16[chosen];
          if (outa) outa[t] = chosen;
        }
        tot += (double)sc_ch;
        s = d_ch;
        ++t;
      }
    }
  }
  if (!live) return;
  if (!ok && r == 0) atomicExch(status, NFST_ERR_LENGTH);
  if (r == 0) {
    lengths[walk] = ok ? t : -1;
    logq[walk] = ok ? (float)(tot - logz64[b]) : kNegInf;
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
  double tot = 0.0;  // (float64 sum, rounded once: a float32 sum of a thousand arcs loses 1e-3)
  for (int t = 0; t < max_len; ++t) {
    const int l = mk[t];
    const int a = (l >= 0 && l < lat.vocab) ? find_arc(lat.arc_label, rp[s], rp[s + 1], l) : -1;
    if (a < 0) { tot = (double)kNegInf; s = 0; break; }
    const int d = lat.arc_dst[a];
    if (d != s) tot += (double)arc_score(theta, arc_w, sc.arc_scores, l, a);
    s = d;
  }
  path_score[walk] = (float)tot;
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

// MODE 0: emission mask (0 / weight / -inf); MODE 1: values[row_off + transition[state, l]].
// One wave per walker: the row of V outputs is built in LDS -- filled with the value of "no arc", then the state's arcs
// (a coalesced read of their labels and weights / destinations) scatter theirs -- and written out with the legality masks
// applied.  (Round 1 searched the state's arcs for every label: V x log2(degree) dependent loads per walker, 1.3 TB/s.)
template <int MODE>
__global__ __launch_bounds__(64) void k_row_gather(nfst_batch lat, const int64_t *state,
                                                   const float *values, const int64_t *inp, int pad,
                                                   int bos, int eos, int has_to_end, float *out,
                                                   int K) {
  extern __shared__ float2 lds[];
  float *row = (float *)lds;
  const int64_t i = blockIdx.x;
  const int b = (int)(i / K);
  const int lane = threadIdx.x;
  const Meta m = load_meta(lat.meta, b);
  const int64_t s = state[i];
  int r0 = 0, r1 = 0;
  if (s >= 0 && s < m.n_rows) {
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    r0 = rp[s]; r1 = rp[s + 1];
  }
  const float none = MODE == 0 ? kNegInf : values[m.row_off];  // a mark without an arc: -inf / the dense table's transition 0
  for (int l = lane; l < lat.vocab; l += 64) row[l] = none;
  for (int a = r0 + lane; a < r1; a += 64)  // (LDS accesses of one wave execute in order: no barrier)
    row[lat.arc_label[a]] = MODE == 0 ? (lat.weighted ? lat.arc_w[a] : 0.0f) : values[m.row_off + lat.arc_dst[a]];
  float *o = out + (size_t)i * lat.vocab;
  bool ended = false, check = false;
  if (MODE == 0 && inp) {
    const int64_t p = inp[i];
    ended = (p == eos) || (p == pad);
    check = true;
  }
  for (int l = lane; l < lat.vocab; l += 64) {
    float v = row[l];
    if (MODE == 0 && check) {
      if (l == bos || (ended ? (l != pad) : (l == pad))) v = kNegInf;
      if (has_to_end && !ended && l != eos) v = kNegInf;
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
// A row lies on L lanes: a quarter wave (L = 16, four rows side by side), a half wave (32) or a
// whole wave (64).  Narrow rows waste fewer of the 16-byte lane slots (V = 300: 75 of 80 with
// L = 16 and five slots per lane, 75 of 128 with a wave per row) and share the per-row
// instructions (reductions, masks) among the rows of a wave; the launcher uses 16 up to V = 512
// and 32 above.  Measured at V = 300: 3.1 TB/s with L = 64, 4.2 with L = 32, 5.3 with L = 16.
template <int L>
__device__ __forceinline__ float row_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));
  v = fmaxf(v, dpp_f<0x4E>(v));
  v = fmaxf(v, dpp_f<0x141>(v));
  v = fmaxf(v, dpp_f<0x140>(v));
  if (L == 16) return v;  // a DPP row: every lane holds its row's result
  const float a = read_lane_f(v, 0), b = read_lane_f(v, 16), c = read_lane_f(v, 32), d = read_lane_f(v, 48);
  if (L == 64) return fmaxf(fmaxf(a, b), fmaxf(c, d));
  return (threadIdx.x & 32) ? fmaxf(c, d) : fmaxf(a, b);
}
template <int L>
__device__ __forceinline__ float row_sum(float v) {
  v += dpp_f<0xB1>(v);
  v += dpp_f<0x4E>(v);
  v += dpp_f<0x141>(v);
  v += dpp_f<0x140>(v);
  if (L == 16) return v;
  const float a = read_lane_f(v, 0), b = read_lane_f(v, 16), c = read_lane_f(v, 32), d = read_lane_f(v, 48);
  if (L == 64) return (a + b) + (c + d);
  return (threadIdx.x & 32) ? (c + d) : (a + b);
}

// Row-level legality (scorers.py:59-83) only depends on three row flags; which of a lane's 4 NV
// columns are illegal under each is a lane constant, one bit per column:
//   normal / first row: bos, pad;  after eos or pad: everything but pad;  forced end: everything but eos.
// GPT2Wrapper mode has no legality masks (only the columns beyond V are "illegal"); m_pad marks the
// pad column, whose logit it replaces by -1e8.
template <int NV, int L>
__device__ __forceinline__ void lane_masks(int ll, int V, int pad, int bos, int eos, bool gpt2, uint32_t &m_norm,
                                           uint32_t &m_end, uint32_t &m_force, uint32_t &m_pad) {
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int col = (c * L + ll) * 4 + k;
      const uint32_t bit = 1u << (c * 4 + k);
      const bool oob = col >= V;
      if (gpt2) {
        if (oob) { m_norm |= bit; m_end |= bit; m_force |= bit; }
        if (col == pad) m_pad |= bit;
        continue;
      }
      if (oob || col == bos || col == pad) m_norm |= bit;
      if (oob || col == bos || col != pad) m_end |= bit;
      if (oob || col == bos || col == pad || col != eos) m_force |= bit;
    }
}

template <int NV, int RB, int L>
__global__ __launch_bounds__(256) void k_path_logprob_v4(const float *__restrict__ scores,
                                                         const int64_t *__restrict__ marks, int T, int V,
                                                         int pad, int bos, int eos, int max_length,
                                                         float temp, int normalize, float smoothing, int mode,
                                                         float *out) {
  __shared__ float part[4];
  constexpr int HR = 64 / L;  // rows side by side in a wave
  const bool gpt2 = mode == 1;  // GPT2Wrapper.forward (transformer.py:45-52): pad logit = -1e8, no legality masks
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ll = lane & (L - 1), hsel = lane / L;
  const int64_t *mk = marks + n * T;
  const float rtemp = 1.0f / temp;
  uint32_t m_norm = 0, m_end = 0, m_force = 0, m_pad = 0;
  lane_masks<NV, L>(ll, V, pad, bos, eos, gpt2, m_norm, m_end, m_force, m_pad);
  float acc = 0.0f;
  for (int t0 = wave * RB * HR; t0 < T; t0 += 4 * RB * HR) {
    float4 v[RB][NV];
    int lab[RB], prev[RB];
    float lraw[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int t = min(t0 + r * HR + hsel, T - 1);  // clamped: rows past the end are loaded but not used
      const float4 *row = reinterpret_cast<const float4 *>(scores + ((size_t)n * T + t) * V);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const int q = c * L + ll;
        v[r][c] = (4 * q < V) ? row[q] : make_float4(kNegInf, kNegInf, kNegInf, kNegInf);
      }
      lab[r] = (int)mk[t];
      prev[r] = t > 0 ? (int)mk[t - 1] : -1;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r)  // the realised marks' scores: all RB gathers in flight together (L2 hits)
      lraw[r] = scores[((size_t)n * T + min(t0 + r * HR + hsel, T - 1)) * V + lab[r]];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (t0 + r * HR >= T) break;  // wave-uniform: no row of this slot exists
      const int t = min(t0 + r * HR + hsel, T - 1);
      const bool valid = t0 + r * HR + hsel < T;  // the second half's row may be past the end
      float sel;
      const float lmsk = gpt2 ? 0.0f : seq_mask(lab[r], t, prev[r], pad, bos, eos, max_length);
      const float lx = ((lab[r] == pad ? (gpt2 ? -1.0e8f : 0.0f) : lraw[r]) + lmsk) / temp + lmsk;
      sel = lx;
      if (normalize || smoothing > 0.0f) {
        const bool first = t == 0;
        const bool ended = !gpt2 && !first && (prev[r] == eos || prev[r] == pad);
        const bool force = !gpt2 && !first && max_length >= 0 && t > max_length && !ended;
        const uint32_t bad = ended ? m_end : (force ? m_force : m_norm);
        float mx = kNegInf;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          float *e = reinterpret_cast<float *>(&v[r][c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // the pad column counts as score 0 (scorers.py:1679-1683); it is only legal in `ended` rows
            // (GPT2Wrapper: the pad logit is -1e8 in every row)
            const float x = ended ? 0.0f : ((m_pad & (1u << (c * 4 + k))) ? -1.0e8f * rtemp : e[k] * rtemp);
            e[k] = (bad & (1u << (c * 4 + k))) ? kNegInf : x;
            mx = fmaxf(mx, e[k]);
          }
        }
        float lse = 0.0f;
        if (normalize) {
          mx = row_max<L>(mx);
          float sm = 0.0f;
#pragma unroll
          for (int c = 0; c < NV; ++c) {
            const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
            for (int k = 0; k < 4; ++k) sm += __expf(e[k] - mx);  // masked / padding columns: exp(-inf) = 0
          }
          sm = row_sum<L>(sm);
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
          sx = row_sum<L>(sx);
          cnt = row_sum<L>(cnt);
          const float own = fminf(fmaxf(sel, -10e8f), 10e8f);
          float rest = sx - cnt * lse;  // sum of the legal marks' values ...
          float others = cnt;
          if (lx > kNegInf) { rest -= sel; others -= 1.0f; }  // ... other than the realised one
          sel = (1.0f - smoothing) * own + (others > 0.0f ? (smoothing / (cnt - 1.0f)) * rest : 0.0f);
        }
      }
      acc += (valid && lab[r] != pad) ? sel : 0.0f;
    }
  }
  // every lane of a row's lanes holds the same acc
  if (HR == 2) acc += read_lane_f(acc, 32);
  if (HR == 4) acc = (read_lane_f(acc, 0) + read_lane_f(acc, 16)) + (read_lane_f(acc, 32) + read_lane_f(acc, 48));
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = ((part[0] + part[1]) + part[2]) + part[3];
}

// value of column v of a row before the log_softmax: masks + pad handling + temperature
// (scorers.py:1564-1580; GPT2Wrapper, transformer.py:45-52: pad logit -1e8, no legality masks)
__device__ __forceinline__ float seq_value(const float *row, int v, int t, int prev, int pad, int bos, int eos,
                                           int max_length, float temp, bool gpt2) {
  if (gpt2) return (v == pad ? -1.0e8f : row[v]) / temp;
  const float msk = seq_mask(v, t, prev, pad, bos, eos, max_length);
  return ((v == pad ? 0.0f : row[v]) + msk) / temp + msk;
}

__global__ __launch_bounds__(256) void k_path_logprob(const float *scores, const int64_t *marks,
                                                      int T, int V, int pad, int bos, int eos,
                                                      int max_length, float temp, int normalize,
                                                      float smoothing, int mode, float *out) {
  __shared__ float part[4];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t *mk = marks + n * T;
  const bool gpt2 = mode == 1;
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
        const float x = seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2);
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
      const float x = seq_value(row, lab, t, prev, pad, bos, eos, max_length, temp, gpt2);
      sel = x - (mx + logf(sm));  // all -inf row: -inf - (-inf + log 0) = NaN, like the reference
    } else {
      sel = seq_value(row, lab, t, prev, pad, bos, eos, max_length, temp, gpt2);
    }
    if (smoothing > 0.0f) {
      // label-smoothed target (scorers.py:1502-1528, 1584-1592)
      float lse = 0.0f;
      const float lx = seq_value(row, lab, t, prev, pad, bos, eos, max_length, temp, gpt2);
      if (normalize) lse = lx - sel;
      float sx = 0.0f, cnt = 0.0f;
      for (int v = lane; v < V; v += 64) {
        const float x = seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2);
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

// ------------------------------------------------------------------ sequence scoring, backward
// d out[n] / d scores[n, t, :] of nfst_path_logprob, times grad_out[n]: the reference trains p~
// straight through evaluate_seq_with_temp (lightning.py:511-516).  With e = the masked, scaled
// row, p = softmax(e), the per-row value is sel = sum_v td[v] * clamp(final[v]), final = e - lse
// (normalize) or e; td = one-hot of the realised mark (evaluation) or the label-smoothed target
// (scorers.py:1502-1528).  So
//     d sel / d scores[u] = (tdc[u] - p[u] * W) * padmask[u] / temp,   tdc = td * clamp',  W = sum tdc
// (W = 0 without the log_softmax); rows whose mark is pad contribute nothing; the pad column never
// receives a gradient (pad_masking_3d multiplies it by zero; GPT2Wrapper overwrites it).  Same
// streaming layout as the forward kernel: a row is recomputed in registers and its gradient row
// is written with 16-byte stores -- N T V 4 bytes read, as many written.
template <int NV, int RB, int L>
__global__ __launch_bounds__(256) void k_path_logprob_bwd_v4(const float *__restrict__ scores,
                                                             const int64_t *__restrict__ marks,
                                                             const float *__restrict__ grad_out, int T, int V, int pad,
                                                             int bos, int eos, int max_length, float temp, int normalize,
                                                             float smoothing, int mode, float *__restrict__ grad) {
  constexpr int HR = 64 / L;
  const bool gpt2 = mode == 1;
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ll = lane & (L - 1), hsel = lane / L;
  const int64_t *mk = marks + n * T;
  const float rtemp = 1.0f / temp;
  const float g = grad_out[n];
  uint32_t m_norm = 0, m_end = 0, m_force = 0, m_pad = 0;
  lane_masks<NV, L>(ll, V, pad, bos, eos, gpt2, m_norm, m_end, m_force, m_pad);
  uint32_t m_padcol = 0;  // the pad column in either mode: no gradient
#pragma unroll
  for (int c = 0; c < NV; ++c)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if ((c * L + ll) * 4 + k == pad) m_padcol |= 1u << (c * 4 + k);
  for (int t0 = wave * RB * HR; t0 < T; t0 += 4 * RB * HR) {
    float4 v[RB][NV];
    int lab[RB], prev[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int t = min(t0 + r * HR + hsel, T - 1);
      const float4 *row = reinterpret_cast<const float4 *>(scores + ((size_t)n * T + t) * V);
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const int q = c * L + ll;
        v[r][c] = (4 * q < V) ? row[q] : make_float4(kNegInf, kNegInf, kNegInf, kNegInf);
      }
      lab[r] = (int)mk[t];
      prev[r] = t > 0 ? (int)mk[t - 1] : -1;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      if (t0 + r * HR >= T) break;  // wave-uniform
      const int t = min(t0 + r * HR + hsel, T - 1);
      const bool valid = t0 + r * HR + hsel < T;
      const bool first = t == 0;
      const bool ended = !gpt2 && !first && (prev[r] == eos || prev[r] == pad);
      const bool force = !gpt2 && !first && max_length >= 0 && t > max_length && !ended;
      const uint32_t bad = ended ? m_end : (force ? m_force : m_norm);
      float mx = kNegInf;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        float *e = reinterpret_cast<float *>(&v[r][c]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float x = ended ? 0.0f : ((m_pad & (1u << (c * 4 + k))) ? -1.0e8f * rtemp : e[k] * rtemp);
          e[k] = (bad & (1u << (c * 4 + k))) ? kNegInf : x;
          mx = fmaxf(mx, e[k]);
        }
      }
      float lse = 0.0f, rsm = 0.0f;
      if (normalize) {
        mx = row_max<L>(mx);
        float sm = 0.0f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) sm += __expf(e[k] - mx);
        }
        sm = row_sum<L>(sm);
        lse = mx + logf(sm);
        rsm = 1.0f / sm;
      }
      // the target's weights: realised mark / other legal marks
      float w_lab = 1.0f, w_other = 0.0f;
      if (smoothing > 0.0f) {
        float cnt = 0.0f;
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
          for (int k = 0; k < 4; ++k) cnt += (e[k] - lse > kNegInf) ? 1.0f : 0.0f;
        }
        cnt = row_sum<L>(cnt);
        w_lab = 1.0f - smoothing;
        w_other = cnt > 1.0f ? smoothing / (cnt - 1.0f) : 0.0f;
      }
      // tdc and its row sum W
      float tdc[NV][4];
      float W = 0.0f;
#pragma unroll
      for (int c = 0; c < NV; ++c) {
        const float *e = reinterpret_cast<const float *>(&v[r][c]);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int col = (c * L + ll) * 4 + k;
          const float fin = e[k] - lse;  // -inf for masked columns (and for columns beyond V)
          // the clamp's derivative (scorers.py:1591): 1 inside [-1e9, 1e9]; evaluation mode has no clamp,
          // but a masked realised mark still passes its one-hot weight (d(-inf + x)/dx = 1)
          const bool inside = smoothing > 0.0f ? (fin >= -10e8f && fin <= 10e8f) : true;
          const float td = (col == lab[r]) ? w_lab : ((fin > kNegInf) ? w_other : 0.0f);
          tdc[c][k] = (col < V && inside) ? td : 0.0f;
          W += tdc[c][k];
        }
      }
      W = normalize ? row_sum<L>(W) : 0.0f;
      const float coef = (valid && lab[r] != pad) ? g * rtemp : 0.0f;
      if (valid) {
        float4 *orow = reinterpret_cast<float4 *>(grad + ((size_t)n * T + t) * V);
#pragma unroll
        for (int c = 0; c < NV; ++c) {
          const float *e = reinterpret_cast<const float *>(&v[r][c]);
          float o[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float p = normalize ? __expf(e[k] - mx) * rsm : 0.0f;
            const float x = coef * (tdc[c][k] - p * W);
            // a row without a contribution is all zeros (never NaN from an all-masked row)
            o[k] = (coef == 0.0f || (m_padcol & (1u << (c * 4 + k)))) ? 0.0f : x;
          }
          const int q = c * L + ll;
          if (4 * q < V) orow[q] = make_float4(o[0], o[1], o[2], o[3]);
        }
      }
    }
  }
}

// scalar fallback (V % 4 != 0 or V > 1024): one wave per row, three passes over the row
__global__ __launch_bounds__(256) void k_path_logprob_bwd(const float *scores, const int64_t *marks, const float *grad_out,
                                                          int T, int V, int pad, int bos, int eos, int max_length,
                                                          float temp, int normalize, float smoothing, int mode,
                                                          float *grad) {
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t *mk = marks + n * T;
  const bool gpt2 = mode == 1;
  const float g = grad_out[n], rtemp = 1.0f / temp;
  for (int t = wave; t < T; t += 4) {
    const float *row = scores + ((size_t)n * T + t) * V;
    float *orow = grad + ((size_t)n * T + t) * V;
    const int prev = t > 0 ? (int)mk[t - 1] : -1;
    const int lab = (int)mk[t];
    const float coef = (lab != pad) ? g * rtemp : 0.0f;
    if (coef == 0.0f) {
      for (int v = lane; v < V; v += 64) orow[v] = 0.0f;
      continue;
    }
    float mx = kNegInf, sm = 0.0f, cnt = 0.0f;
    for (int v = lane; v < V; v += 64) {
      const float x = seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2);
      mx = fmaxf(mx, x);
      if (x > kNegInf) cnt += 1.0f;
    }
    mx = wave_max(mx);
    cnt = wave_sum(cnt);
    float lse = 0.0f, rsm = 0.0f;
    if (normalize) {
      for (int v = lane; v < V; v += 64) sm += __expf(seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2) - mx);
      sm = wave_sum(sm);
      lse = mx + logf(sm);
      rsm = 1.0f / sm;
    }
    const float w_lab = smoothing > 0.0f ? 1.0f - smoothing : 1.0f;
    const float w_other = (smoothing > 0.0f && cnt > 1.0f) ? smoothing / (cnt - 1.0f) : 0.0f;
    auto tdc_of = [&](int v, float x) {
      const float fin = x - lse;
      const bool inside = smoothing > 0.0f ? (fin >= -10e8f && fin <= 10e8f) : true;
      const float td = (v == lab) ? w_lab : ((fin > kNegInf) ? w_other : 0.0f);
      return inside ? td : 0.0f;
    };
    float W = 0.0f;
    if (normalize) {
      for (int v = lane; v < V; v += 64) W += tdc_of(v, seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2));
      W = wave_sum(W);
    }
    for (int v = lane; v < V; v += 64) {
      const float x = seq_value(row, v, t, prev, pad, bos, eos, max_length, temp, gpt2);
      const float p = normalize ? __expf(x - mx) * rsm : 0.0f;
      orow[v] = (v == pad) ? 0.0f : coef * (tdc_of(v, x) - p * W);
    }
  }
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

// ------------------------------------------------------------------ fused proposal step
// Fused lattice side of one proposal-sampler step (Sampler.stateful_sample, samplers.py:243-297):
// logits = (pad_masking(scores [+ values of the next states] [- insertion / length penalties])
// + emission row + legality masks) / temperature -> logsumexp, inverse-CDF sample on a supplied
// uniform (or the forced symbol), its log probability, next state.  One wave per walker; the
// walker's row of logits and next states lives in LDS (V <= kStepMaxVocab).
constexpr int kStepMaxVocab = 4096, kStepWaves = 4;
__global__ __launch_bounds__(64 * kStepWaves) void k_proposal_step(
    nfst_batch lat, const int64_t *state, const int64_t *inp, const float *scores, const float *values, int pad, int bos,
    int eos, int has_to_end, float temperature, const float *uniforms, const int64_t *forced, nfst_step_extras ex,
    int64_t *symbol, float *logq, float *logz, int64_t *next_state, float *logits_out, int K, int64_t n_walkers) {
  extern __shared__ float2 lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t n = (int64_t)blockIdx.x * kStepWaves + wave;
  if (n >= n_walkers) return;
  const int V = lat.vocab;
  const bool own_row = values && ex.value_state;  // the value gather reads another state's transition row
  const int per_wave = (own_row ? 3 : 2) * V;
  float *xs = (float *)lds + (size_t)wave * per_wave;  // logits of the walker's row
  int *nx = (int *)(xs + V);                            // next state per mark (-1: no arc)
  int *nv = own_row ? nx + V : nx;                      // next state per mark from the value state (0: no arc)
  const int b = (int)(n / K);
  const Meta m = load_meta(lat.meta, b);
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  const int64_t s = state[n];
  int r0 = 0, r1 = 0;
  if (s >= 0 && s < m.n_rows) { r0 = rp[s]; r1 = rp[s + 1]; }
  // LDS accesses of one wave execute in order: no barrier between the phases
  for (int v = lane; v < V; v += 64) { xs[v] = kNegInf; nx[v] = -1; }
  if (own_row) {
    for (int v = lane; v < V; v += 64) nv[v] = 0;  // like the dense gather: no arc reads row 0
    const int64_t vs = ex.value_state[n];
    if (vs >= 0 && vs < m.n_rows)
      for (int a = rp[vs] + lane; a < rp[vs + 1]; a += 64) nv[lat.arc_label[a]] = lat.arc_dst[a];
  }
  for (int a = r0 + lane; a < r1; a += 64) {  // the state's arcs carry distinct marks
    const int l = lat.arc_label[a];
    xs[l] = lat.weighted ? lat.arc_w[a] : 0.0f;
    nx[l] = lat.arc_dst[a];
  }
  const int64_t prev = inp ? inp[n] : -1;
  const bool ended = inp && (prev == eos || prev == pad);
  const float rt = 1.0f / temperature;
  // insertion / length penalties (scorers.py:654-677): the counters are updated with the previous
  // symbol first, then read
  float ins_pen = 0.0f, use_prev = 0.0f;
  bool len_pen = false;
  const float *vu = ex.vocab_use ? ex.vocab_use + (size_t)n * V : nullptr;
  if (inp && prev >= 0 && prev < V) {
    // every lane reads the old counters (one address each), then lane 0 stores the new ones: the
    // loop below never reads a location this wave has just written
    if (ex.accumulated) {
      const int64_t acc = ex.accumulated[n] + (prev == ex.insertion_mark ? 1 : 0);
      if (lane == 0) ex.accumulated[n] = acc;
      if (ex.insert_threshold > 0 && acc > ex.insert_threshold)
        ins_pen = ex.insert_penalty * (float)(ex.length - ex.insert_threshold);
    }
    if (vu) {
      use_prev = vu[prev] + 1.0f;
      if (lane == 0) ex.vocab_use[(size_t)n * V + prev] = use_prev;
      len_pen = 0 < ex.length_threshold && ex.length_threshold < ex.length;
    }
  }
  float mx = kNegInf;
  for (int v = lane; v < V; v += 64) {
    float x = xs[v];
    if (inp) {  // bos / pad / eos legality (scorers.py:59-83)
      if (v == bos || (ended ? (v != pad) : (v == pad))) x = kNegInf;
      if (has_to_end && !ended && v != eos) x = kNegInf;
    }
    if (x > kNegInf) {
      float sc = 0.0f;  // pad_masking of the summed scores (scorers.py:182-187, 357)
      if (v != pad) {
        sc = scores[(size_t)n * V + v];
        if (values) sc += values[m.row_off + (own_row ? nv[v] : nx[v])];
        if (v == ex.insertion_mark) sc -= ins_pen;
        if (len_pen) sc -= (v == (int)prev ? use_prev : vu[v]) * ex.length_penalty;
      }
      x = (sc + x) * rt;
    }
    xs[v] = x;
    mx = fmaxf(mx, x);
  }
  mx = wave_max(mx);
  float sm = 0.0f;
  if (mx > kNegInf)
    for (int v = lane; v < V; v += 64) sm += __expf(xs[v] - mx);
  sm = wave_sum(sm);
  const float lz = (mx > kNegInf) ? mx + logf(sm) : kNegInf;  // no legal mark: log z = -inf, symbol = pad, log q = -inf
  if (logits_out)
    for (int v = lane; v < V; v += 64) logits_out[(size_t)n * V + v] = xs[v];
  int sym = -1;
  if (!(mx > kNegInf)) {
    sym = -1;
  } else if (forced) {
    sym = (int)forced[n];
  } else {
    const float u = uniforms[n];
    float base = 0.0f;
    int last = -1;
    for (int c = 0; c < V && sym < 0; c += 64) {
      const int v = c + lane;
      const float p = (v < V) ? __expf(xs[v] - lz) : 0.0f;
      float cum = p;
      for (int d = 1; d < 64; d <<= 1) {
        const float o = __shfl_up(cum, d);
        if (lane >= d) cum += o;
      }
      cum += base;
      const uint64_t hit = __ballot(p > 0.0f && u < cum), pos = __ballot(p > 0.0f);
      if (hit) sym = c + __builtin_ctzll(hit);
      else {
        if (pos) last = c + 63 - __builtin_clzll(pos);
        base = __shfl(cum, 63);
      }
    }
    if (sym < 0) sym = last;  // u beyond the rounded total: the last legal mark
  }
  if (lane == 0) {
    const bool ok = sym >= 0 && sym < V;
    symbol[n] = ok ? sym : pad;
    if (ex.not_pad && ok && sym != pad) atomicAdd(ex.not_pad, 1);
    logq[n] = ok ? xs[sym] - lz : kNegInf;
    if (logz) logz[n] = lz;
    next_state[n] = (ok && nx[sym] >= 0) ? nx[sym] : 0;  // like nfst_step: 0 where the table has no arc
  }
}

// Backward of the fused step with respect to the proposal network's scores (and optionally the
// gathered values): the reference's Categorical.log_prob is differentiable in the logits
// (samplers.py:256-273).  logits = the masked, scaled row the forward pass wrote out;
//   d logq / d scores[v] = (onehot(symbol)[v] - p[v]) / T,   d logz / d scores[v] = p[v] / T
// for legal marks other than pad (pad_masking multiplies the pad column by zero), p = softmax(logits).
// grad_values (optional, row-indexed, zeroed by the caller) receives the same numbers through the
// next-state gather (float atomics).  One wave per walker.
__global__ __launch_bounds__(64 * kStepWaves) void k_proposal_step_bwd(
    nfst_batch lat, const int64_t *vstate, const float *logits, const int64_t *symbol, const float *logz, const float *g_logq,
    const float *g_logz, int pad, float temperature, float *grad_scores, float *grad_values, int K, int64_t n_walkers) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t n = (int64_t)blockIdx.x * kStepWaves + wave;
  if (n >= n_walkers) return;
  const int V = lat.vocab;
  const float rt = 1.0f / temperature;
  const float gq = g_logq ? g_logq[n] : 0.0f, gz = g_logz ? g_logz[n] : 0.0f;
  const float lz = logz[n];
  const int sym = (int)symbol[n];
  const float *x = logits + (size_t)n * V;
  auto grad_of = [&](int v) {
    const float xv = x[v];
    if (!(xv > kNegInf) || !(lz > kNegInf) || v == pad) return 0.0f;
    const float p = __expf(xv - lz);
    return rt * (gq * ((v == sym ? 1.0f : 0.0f) - p) + gz * p);
  };
  for (int v = lane; v < V; v += 64) grad_scores[(size_t)n * V + v] = grad_of(v);
  if (grad_values) {
    const int b = (int)(n / K);
    const Meta m = load_meta(lat.meta, b);
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    const int64_t s = vstate[n];
    // marks with an arc out of the value state add to that arc's destination row; every other legal
    // mark read row 0 in the forward pass.  Row 0 gets the sum over exactly those marks (a bit set per mark with
    // an arc, in LDS): "sum of all minus sum over the arcs" left a float32 rounding residue -- nondeterministic noise
    // on row 0 of every lattice -- where the true value is zero (value_state = the masks' state: every legal mark
    // has an arc).
    __shared__ uint32_t has_arc[kStepWaves][(kStepMaxVocab + 31) / 32];
    uint32_t *bits = has_arc[wave];
    for (int w = lane; w < (V + 31) / 32; w += 64) bits[w] = 0u;
    __builtin_amdgcn_wave_barrier();  // (one wave per walker: its LDS accesses execute in order)
    if (s >= 0 && s < m.n_rows)
      for (int a = rp[s] + lane; a < rp[s + 1]; a += 64) {
        const int l = lat.arc_label[a];
        const float gv = grad_of(l);
        if (gv != 0.0f) atomicAdd(grad_values + m.row_off + lat.arc_dst[a], gv);
        atomicOr(&bits[l >> 5], 1u << (l & 31));
      }
    __builtin_amdgcn_wave_barrier();
    float to_row0 = 0.0f;
    for (int v = lane; v < V; v += 64)
      if (!((bits[v >> 5] >> (v & 31)) & 1u)) to_row0 += grad_of(v);
    to_row0 = wave_sum(to_row0);
    if (lane == 0 && to_row0 != 0.0f) atomicAdd(grad_values + m.row_off, to_row0);
  }
}
