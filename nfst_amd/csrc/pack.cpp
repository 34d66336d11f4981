// pack.cpp -- host side of the lattice engine: dense tables / arc lists ->
// canonical CSR + level-scheduled sweep streams (K1 of SURVEY.md section 2).
//
// Replaces FSAGRUScorer.set_masks/set_k (/root/reference/src/modules/scorers.py:
// 877-918): instead of keeping (and K-fold copying) the dense [S+1,V] tables, the
// lattice is stored once as arcs; the per-call graph construction that the
// reference's beta sweep redoes with S*V .item() calls (scorers.py:704-716,
// 764-776) becomes this one-time schedule.  Format: DESIGN.md section 3.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <new>
#include <thread>
#include <string>
#include <vector>

#include "nfst_hip.h"

namespace {

struct Opts {
  int n_threads = 0;
  int slots_per_lane = 0;  // 0 = choose per lattice and direction
  int group_mode = 0;      // 0 = choose, 1 = narrow, 2 = wide
  bool no_compact = false; // keep 32-bit records
};

struct Lat {
  int n_rows = 0;
  // canonical arcs of reachable states, sorted by (src, label)
  std::vector<int32_t> src, label, dst;
  std::vector<float> w;
  // results
  std::vector<int32_t> row_ptr;  // n_rows + 1, relative
  std::vector<uint32_t> fwd, bwd;
  std::vector<int32_t> fwd_perm, bwd_perm;  // per tile slot: relative canonical arc id or -1
  bool fwd_compact = false, bwd_compact = false;
  int fwd_tiles = 0, bwd_tiles = 0, fwd_u = 4, bwd_u = 4, fwd_wide = 0, bwd_wide = 0, scratch_rows = 0, sink = 0, n_reach = 0, depth = 0, n_dp = 0;
  int err = NFST_OK;
};

inline int ceil_log2(int x) { int g = 0; while ((1 << g) < x) ++g; return g; }

// One sweep direction as a "tile program" (DESIGN.md section 3).  A tile is one
// wave-wide unit of work: 64 control words + 64*U arc records (U = slots per lane).
// A state with d arcs takes 2^g lanes, g = ceil(log2(ceil(d/U))), at a lane offset
// that is a multiple of 2^g; lane r of the group owns arcs [r*U, r*U+U).  Tiles
// never mix levels, so every record's operand was produced by an earlier tile.
//
// A state whose arcs do not fit the largest group (2^max_g lanes) is cut into pieces
// that go into successive tiles: every continuation piece starts with a CARRY record
// (operand = the state itself, label = vocab + 1, weight one), so its sum includes what
// the earlier pieces stored and the piece simply overwrites the state's value.  Its
// leader lane also carries the "accumulate" flag (the max-plus kernel keeps the earlier
// back pointer when the carry wins).  max_g = 3 ("narrow": groups of up to 8 lanes, the
// sweep needs no cross-row reduction stage) or 6 ("wide": up to the whole wave).
// A state with many more arcs than a group holds (three or more groups' worth) is summed as
// a tree instead of a chain: its arcs are spread over PARTIAL groups that write scratch rows
// (ids from n_rows up, reused from level to level) -- up to eight of them side by side in one
// tile -- and a COMBINE piece in a later tile adds the scratch rows up with unit-label records.
struct Piece {
  int32_t state;       // row that receives the sum (a scratch row for a partial group)
  int32_t begin, end;  // arcs [begin,end) of the state's list, or scratch rows [begin,end) for a combine piece
  bool accum;          // continuation piece: starts with the carry record
  bool units;          // combine piece: its records are unit-label records of scratch rows
};

struct TileCount { int tiles = 0, wide = 0, scratch = 0; };  // wide: tiles whose largest group exceeds 8 lanes

template <class ArcsOf, class Other>
void emit_level(const std::vector<int32_t> &states, int U, int max_g, bool compact, uint32_t null_label, int n_rows,
                ArcsOf arcs_of, Other other, const std::vector<int32_t> &list, const std::vector<int32_t> &label,
                std::vector<uint32_t> *stream, std::vector<int32_t> *perm, TileCount &count) {
  const int cap = (1 << max_g) * U;
  const uint32_t null_rec = null_label << 16, unit_label = null_label + 1;
  // pass k holds the k-th piece of every state of the level
  std::vector<std::vector<Piece>> passes(1);
  auto add = [&](size_t k, const Piece &p) {
    if (passes.size() <= k) passes.emplace_back();
    passes[k].push_back(p);
  };
  // a chain of pieces over the index range [b, e): the first takes `cap` slots, every
  // continuation piece one less (its carry), starting in pass k0
  auto chain = [&](int32_t state, int b, int e, bool units, size_t k0) {
    for (size_t k = k0;; ++k) {
      const int room = (k == k0) ? cap : cap - 1;
      const int stop = std::min(e, b + room);
      add(k, {state, b, stop, k > k0, units});
      b = stop;
      if (b >= e) break;
    }
  };
  int scratch = n_rows;  // next free scratch row of this level
  for (int32_t s : states) {
    auto r = arcs_of(s);
    const int d = r.second - r.first;
    if (max_g == 3 && d > 2 * cap && n_rows + (d + cap - 1) / cap + (scratch - n_rows) <= NFST_MAX_ROWS) {
      const int first = scratch;
      for (int b = r.first; b < r.second; b += cap) add(0, {scratch++, b, std::min(r.second, b + cap), false, false});
      chain(s, first, scratch, true, 1);
    } else {
      chain(s, r.first, r.second, false, 0);
    }
  }
  count.scratch = std::max(count.scratch, scratch - n_rows);
  auto lanes_of = [&](const Piece &p) { return std::max(1, (p.end - p.begin + (p.accum ? 1 : 0) + U - 1) / U); };
  for (auto &pieces : passes) {
    std::stable_sort(pieces.begin(), pieces.end(), [&](const Piece &a, const Piece &b) {
      return ceil_log2(lanes_of(a)) > ceil_log2(lanes_of(b));
    });
    size_t i = 0;
    while (i < pieces.size()) {
      uint32_t ctl[64];
      std::vector<uint32_t> rec;
      std::vector<int32_t> pm;
      if (stream) { rec.assign((size_t)64 * U, null_rec); pm.assign((size_t)64 * U, -1); }
      for (int l = 0; l < 64; ++l) ctl[l] = 0;
      int lane = 0, gmax = 0;
      bool any_accum = false;
      while (i < pieces.size()) {
        const Piece &p = pieces[i];
        const int g = ceil_log2(lanes_of(p));
        const int size = 1 << g;
        if (lane + size > 64) break;
        gmax = std::max(gmax, g);
        any_accum = any_accum || p.accum;
        if (stream) {
          int slot = 0;  // position among the piece's slots: the carry first, then the arcs
          const int n_slots = (p.end - p.begin) + (p.accum ? 1 : 0);
          for (int r = 0; r < size; ++r) {
            uint32_t c = ((uint32_t)p.state << 3) | ((uint32_t)g << 20);
            if (r == 0) c |= (1u << 31) | (p.accum ? (1u << 30) : 0u);
            ctl[lane + r] = c;
            for (int j = 0; j < U; ++j, ++slot) {
              if (slot >= n_slots) continue;
              const size_t at = (size_t)(lane + r) * U + j;
              if (p.accum && slot == 0) {
                rec[at] = ((uint32_t)p.state << 3) | (unit_label << 16);
              } else if (p.units) {
                rec[at] = ((uint32_t)(p.begin + slot - (p.accum ? 1 : 0)) << 3) | (unit_label << 16);
              } else {
                const int32_t arc = list[p.begin + slot - (p.accum ? 1 : 0)];
                rec[at] = ((uint32_t)other(arc) << 3) | ((uint32_t)label[arc] << 16);
                pm[at] = arc;
              }
            }
          }
        }
        lane += size;
        ++i;
      }
      if (stream) {
        for (int l = 0; l < 64; ++l) ctl[l] |= ((uint32_t)gmax << 23) | (any_accum ? (1u << 26) : 0u);
        if (!compact) {
          stream->insert(stream->end(), ctl, ctl + 64);
          stream->insert(stream->end(), rec.begin(), rec.end());
        } else {
          // compact tile (U = 4, labels < 2048): 16 bytes per lane = control word + four 24-bit
          // records (state id 13 bits | label 11 bits) -- one 16-byte LDS-DMA per tile
          for (int l = 0; l < 64; ++l) {
            uint32_t r24[4];
            for (int j = 0; j < 4; ++j) {
              const uint32_t x = rec[(size_t)l * 4 + j];
              r24[j] = ((x & 0xffffu) >> 3) | ((x >> 16) << 13);
            }
            stream->push_back(ctl[l]);
            stream->push_back(r24[0] | (r24[1] << 24));
            stream->push_back((r24[1] >> 8) | (r24[2] << 16));
            stream->push_back((r24[2] >> 16) | (r24[3] << 8));
          }
        }
        perm->insert(perm->end(), pm.begin(), pm.end());
      }
      ++count.tiles;
      if (gmax > 3) ++count.wide;
    }
  }
}

void schedule(Lat &L, int vocab, const Opts &o) {
  const int n = L.n_rows;
  const int A = (int)L.src.size();
  if (n > NFST_MAX_ROWS || vocab > NFST_MAX_VOCAB) { L.err = NFST_ERR_LIMIT; return; }
  L.row_ptr.assign(n + 1, 0);
  for (int a = 0; a < A; ++a) L.row_ptr[L.src[a] + 1]++;
  for (int s = 0; s < n; ++s) L.row_ptr[s + 1] += L.row_ptr[s];
  // reachable = states with a canonical row or reached by one (input is already
  // restricted to arcs of reachable states) + the start
  std::vector<uint8_t> reach(n, 0);
  reach[0] = 1;
  for (int a = 0; a < A; ++a) { reach[L.src[a]] = 1; reach[L.dst[a]] = 1; }
  std::vector<int32_t> indeg(n, 0), outdeg(n, 0);
  int n_dp = 0;
  for (int a = 0; a < A; ++a)
    if (L.src[a] != L.dst[a]) { indeg[L.dst[a]]++; outdeg[L.src[a]]++; ++n_dp; }
  L.n_dp = n_dp;
  int n_reach = 0, sinks = 0, sink = -1;
  for (int s = 0; s < n; ++s)
    if (reach[s]) { ++n_reach; if (outdeg[s] == 0) { ++sinks; sink = s; } }
  L.n_reach = n_reach;
  if (sinks != 1) { L.err = NFST_ERR_SINK; return; }
  L.sink = sink;
  // in-arc lists (CSC), stable in canonical order
  std::vector<int32_t> in_ptr(n + 1, 0), in_list(n_dp), out_list(n_dp), out_ptr(n + 1, 0);
  for (int s = 0; s < n; ++s) { in_ptr[s + 1] = in_ptr[s] + indeg[s]; out_ptr[s + 1] = out_ptr[s] + outdeg[s]; }
  {
    std::vector<int32_t> ip(in_ptr.begin(), in_ptr.end() - 1), op(out_ptr.begin(), out_ptr.end() - 1);
    for (int a = 0; a < A; ++a)
      if (L.src[a] != L.dst[a]) { in_list[ip[L.dst[a]]++] = a; out_list[op[L.src[a]]++] = a; }
  }
  // Kahn from the start; depth = longest path from 0
  std::vector<int32_t> order;
  order.reserve(n_reach);
  std::vector<int32_t> depth(n, 0), height(n, 0), rem(indeg);
  if (rem[0] != 0) { L.err = NFST_ERR_CYCLE; return; }
  order.push_back(0);
  for (size_t h = 0; h < order.size(); ++h) {
    int s = order[h];
    for (int q = out_ptr[s]; q < out_ptr[s + 1]; ++q) {
      int d = L.dst[out_list[q]];
      depth[d] = std::max(depth[d], depth[s] + 1);
      if (--rem[d] == 0) order.push_back(d);
    }
  }
  if ((int)order.size() != n_reach) { L.err = NFST_ERR_CYCLE; return; }
  for (int i = n_reach - 1; i >= 0; --i) {
    int s = order[i];
    for (int q = out_ptr[s]; q < out_ptr[s + 1]; ++q)
      height[s] = std::max(height[s], height[L.dst[out_list[q]]] + 1);
  }
  L.depth = depth[sink];
  const int D = L.depth;
  // levels
  std::vector<std::vector<int32_t>> by_depth(D + 1), by_height(D + 1);
  for (int s = 0; s < n; ++s)
    if (reach[s]) { by_depth[depth[s]].push_back(s); by_height[height[s]].push_back(s); }
  for (int t = 1; t <= D; ++t) {
    std::stable_sort(by_height[t].begin(), by_height[t].end(), [&](int a, int b) { return outdeg[a] > outdeg[b]; });
    std::stable_sort(by_depth[t].begin(), by_depth[t].end(), [&](int a, int b) { return indeg[a] > indeg[b]; });
  }
  auto out_of = [&](int s) { return std::make_pair(out_ptr[s], out_ptr[s + 1]); };
  auto in_of = [&](int s) { return std::make_pair(in_ptr[s], in_ptr[s + 1]); };
  auto dst_of = [&](int a) { return L.dst[a]; };
  auto src_of = [&](int a) { return L.src[a]; };
  const uint32_t null_label = (uint32_t)vocab;
  // Slots per lane U and the largest group (narrow / wide) per direction: the cheapest
  // program by a cost model of the sweep kernel -- cycles per tile as measured on MI355X
  // (one wave): ~330 + 55 U, and ~450 more for a tile on the general
  // path; a program without wide tiles also saves the per-tile test for them.
  // Round 2: where the labels fit the compact tile, four slots per lane always -- all-compact batches run
  // the tile-wave / fused kernels, whose tile costs ~370 cycles whatever its fill (measured, 256 lattices
  // of 2k states: 551 levels 131 -> 106 us, 277 levels 77 -> 59 us against the cost model's choice of U = 1, 2).
  const bool compact_ok = vocab + 2 <= 2048 && !o.no_compact;
  auto pick = [&](bool backward, int &u_out, int &wide_out) {
    const int us[3] = {1, 2, 4};
    double best = 0.0;
    bool have = false;
    for (int q = 0; q < 3; ++q) {
      if ((o.slots_per_lane == 1 || o.slots_per_lane == 2 || o.slots_per_lane == 4) && us[q] != o.slots_per_lane) continue;
      if (o.slots_per_lane == 0 && compact_ok && us[q] != 4) continue;
      for (int wide = 0; wide < 2; ++wide) {
        if ((o.group_mode == 1 && wide) || (o.group_mode == 2 && !wide)) continue;
        TileCount c;
        for (int t = 1; t <= D; ++t) {
          if (backward) emit_level(by_height[t], us[q], wide ? 6 : 3, false, null_label, n, out_of, dst_of, out_list, L.label, nullptr, nullptr, c);
          else emit_level(by_depth[t], us[q], wide ? 6 : 3, false, null_label, n, in_of, src_of, in_list, L.label, nullptr, nullptr, c);
        }
        if (wide && c.wide == 0 && o.group_mode != 2) continue;  // same program as the narrow one
        const double cost = (double)c.tiles * (330.0 + 55.0 * us[q] + (wide ? 60.0 : 0.0)) + 450.0 * c.wide;
        if (!have || cost < best) { have = true; best = cost; u_out = us[q]; wide_out = wide; }
      }
    }
  };
  pick(true, L.bwd_u, L.bwd_wide);
  pick(false, L.fwd_u, L.fwd_wide);
  L.fwd.clear(); L.bwd.clear(); L.fwd_perm.clear(); L.bwd_perm.clear();
  // programs with four slots per lane use the compact tile when the labels fit 11 bits
  L.bwd_compact = compact_ok && L.bwd_u == 4;
  L.fwd_compact = compact_ok && L.fwd_u == 4;
  TileCount cb, cf;
  for (int t = 1; t <= D; ++t) {
    emit_level(by_height[t], L.bwd_u, L.bwd_wide ? 6 : 3, L.bwd_compact, null_label, n, out_of, dst_of, out_list, L.label, &L.bwd, &L.bwd_perm, cb);
    emit_level(by_depth[t], L.fwd_u, L.fwd_wide ? 6 : 3, L.fwd_compact, null_label, n, in_of, src_of, in_list, L.label, &L.fwd, &L.fwd_perm, cf);
  }
  L.bwd_tiles = cb.tiles;
  L.fwd_tiles = cf.tiles;
  L.scratch_rows = std::max(cb.scratch, cf.scratch);
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  n_threads = std::max(1, std::min(n_threads, n));
  if (n_threads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); });
  for (auto &x : th) x.join();
}

Opts read_opts(const nfst_pack_opts *o) {
  Opts r;
  if (o) {
    r.n_threads = o->n_threads;
    r.slots_per_lane = o->slots_per_lane;
    r.group_mode = o->group_mode;
    r.no_compact = o->reserved1 == 1;
  }
  return r;
}

}  // namespace

struct nfst_packed {
  nfst_batch view{};
  std::vector<int32_t> meta, row_ptr, arc_src, arc_dst, arc_label, fwd_perm, bwd_perm;
  std::vector<uint32_t> arc_sd;
  std::vector<uint16_t> arc_l16;
  std::vector<float> arc_w;
  std::vector<uint32_t> fwd, bwd;
};

static int finish(std::vector<Lat> &lats, int vocab, bool weighted, const Opts &o, nfst_packed **out,
                  int32_t *err_lattice) {
  const int B = (int)lats.size();
  parallel_for(B, o.n_threads, [&](int b) { if (lats[b].err == NFST_OK) schedule(lats[b], vocab, o); });
  for (int b = 0; b < B; ++b)
    if (lats[b].err != NFST_OK) { if (err_lattice) *err_lattice = b; return lats[b].err; }
  nfst_packed *p = new (std::nothrow) nfst_packed();
  if (!p) return NFST_ERR_NOMEM;
  int64_t rows = 0, arcs = 0, dp = 0, fw = 0, bw = 0, fs = 0, bs = 0;
  int max_rows = 0, max_tiles = 0;
  p->meta.assign((size_t)B * NFST_META_WORDS, 0);
  for (int b = 0; b < B; ++b) {
    Lat &L = lats[b];
    int32_t *m = &p->meta[(size_t)b * NFST_META_WORDS];
    m[NFST_META_ROW_OFF] = (int32_t)rows; m[NFST_META_N_ROWS] = L.n_rows;
    m[NFST_META_ARC_OFF] = (int32_t)arcs; m[NFST_META_N_ARCS] = (int32_t)L.src.size();
    m[NFST_META_FWD_OFF] = (int32_t)fw; m[NFST_META_FWD_TILES] = L.fwd_tiles;
    m[NFST_META_BWD_OFF] = (int32_t)bw; m[NFST_META_BWD_TILES] = L.bwd_tiles;
    m[NFST_META_SINK] = L.sink; m[NFST_META_N_REACH] = L.n_reach; m[NFST_META_DEPTH] = L.depth;
    m[NFST_META_N_DP] = L.n_dp; m[NFST_META_FWD_U] = (L.fwd_compact ? 8 : L.fwd_u) | (L.fwd_wide << 8);
    m[NFST_META_BWD_U] = (L.bwd_compact ? 8 : L.bwd_u) | (L.bwd_wide << 8);
    m[NFST_META_FWD_SLOT_OFF] = (int32_t)fs; m[NFST_META_BWD_SLOT_OFF] = (int32_t)bs;
    rows += L.n_rows; arcs += (int64_t)L.src.size(); dp += L.n_dp;
    // tile sizes are multiples of 64 words, so every lattice's stream starts on a
    // 256-byte boundary (LDS-DMA chunks are 16 B per lane)
    fw += (int64_t)L.fwd.size(); bw += (int64_t)L.bwd.size();
    fs += (int64_t)L.fwd_perm.size(); bs += (int64_t)L.bwd_perm.size();
    max_rows = std::max(max_rows, L.n_rows + L.scratch_rows);
    max_tiles = std::max(max_tiles, std::max(L.fwd_tiles, L.bwd_tiles));
    if (arcs > 0x7fffff00ll || fw > 0x7ffff000ll || bw > 0x7ffff000ll || rows > 0x7fffff00ll ||
        fs > 0x7fffff00ll || bs > 0x7fffff00ll) {
      delete p; if (err_lattice) *err_lattice = b; return NFST_ERR_LIMIT;
    }
  }
  // slack at the end of each stream: an empty last lattice still owns valid memory
  const int64_t slack = 512;
  p->row_ptr.resize(rows + B); p->arc_src.resize(arcs); p->arc_dst.resize(arcs); p->arc_label.resize(arcs);
  if (weighted) p->arc_w.resize(arcs);
  p->arc_sd.assign(arcs + 8, 0); p->arc_l16.assign(arcs + 8, 0);
  p->fwd.assign(fw + slack, 0); p->bwd.assign(bw + slack, 0); p->fwd_perm.resize(fs); p->bwd_perm.resize(bs);
  parallel_for(B, o.n_threads, [&](int b) {
    Lat &L = lats[b];
    const int32_t *m = &p->meta[(size_t)b * NFST_META_WORDS];
    int32_t a0 = m[NFST_META_ARC_OFF];
    int32_t *rp = &p->row_ptr[(size_t)m[NFST_META_ROW_OFF] + b];
    for (int s = 0; s <= L.n_rows; ++s) rp[s] = a0 + L.row_ptr[s];
    size_t A = L.src.size();
    if (A) {
      std::memcpy(&p->arc_src[a0], L.src.data(), A * 4);
      std::memcpy(&p->arc_dst[a0], L.dst.data(), A * 4);
      std::memcpy(&p->arc_label[a0], L.label.data(), A * 4);
      if (weighted) std::memcpy(&p->arc_w[a0], L.w.data(), A * 4);
      for (size_t i = 0; i < A; ++i) {
        p->arc_sd[a0 + i] = (uint32_t)L.src[i] | ((uint32_t)L.dst[i] << 16);
        p->arc_l16[a0 + i] = (uint16_t)L.label[i];
      }
    }
    if (!L.fwd.empty()) std::memcpy(&p->fwd[m[NFST_META_FWD_OFF]], L.fwd.data(), L.fwd.size() * 4);
    if (!L.bwd.empty()) std::memcpy(&p->bwd[m[NFST_META_BWD_OFF]], L.bwd.data(), L.bwd.size() * 4);
    int32_t *fp = L.fwd_perm.empty() ? nullptr : &p->fwd_perm[m[NFST_META_FWD_SLOT_OFF]];
    int32_t *bp = L.bwd_perm.empty() ? nullptr : &p->bwd_perm[m[NFST_META_BWD_SLOT_OFF]];
    for (size_t i = 0; i < L.fwd_perm.size(); ++i) fp[i] = L.fwd_perm[i] < 0 ? -1 : a0 + L.fwd_perm[i];
    for (size_t i = 0; i < L.bwd_perm.size(); ++i) bp[i] = L.bwd_perm[i] < 0 ? -1 : a0 + L.bwd_perm[i];
    std::vector<int32_t>().swap(L.src); std::vector<uint32_t>().swap(L.fwd); std::vector<uint32_t>().swap(L.bwd);
  });
  nfst_batch &v = p->view;
  v.n_lattices = B; v.vocab = vocab; v.max_rows = max_rows; v.max_tiles = max_tiles;
  bool all_compact = true;
  for (const Lat &L : lats) all_compact = all_compact && L.fwd_compact && L.bwd_compact;
  int64_t max_arcs = 0;
  for (const Lat &L : lats) max_arcs = std::max<int64_t>(max_arcs, (int64_t)L.label.size());
  v.weighted = weighted ? 1 : 0;
  v.reserved0 = (all_compact ? NFST_BATCH_ALL_COMPACT : 0) | (int32_t)(std::min<int64_t>(max_arcs, NFST_BATCH_MAX_ARCS_CAP) << NFST_BATCH_MAX_ARCS_SHIFT);
  v.total_rows = rows; v.total_arcs = arcs; v.total_dp_arcs = dp;
  v.fwd_words = fw + slack; v.bwd_words = bw + slack; v.fwd_slots = fs; v.bwd_slots = bs;
  v.meta = p->meta.data(); v.row_ptr = p->row_ptr.data(); v.arc_src = p->arc_src.data();
  v.arc_dst = p->arc_dst.data(); v.arc_label = p->arc_label.data();
  v.arc_w = weighted ? p->arc_w.data() : nullptr;
  v.fwd_stream = p->fwd.data(); v.bwd_stream = p->bwd.data();
  v.fwd_perm = p->fwd_perm.data(); v.bwd_perm = p->bwd_perm.data();
  v.arc_sd = p->arc_sd.data(); v.arc_l16 = p->arc_l16.data();
  *out = p;
  return NFST_OK;
}

// ---------------------------------------------------------------- packed batches on the host: check, checksum, concatenate
namespace {
// CRC-32C (Castagnoli): the SSE4.2 instruction where the CPU has it (8 bytes per step, three streams interleaved
// would be faster still; this runs at ~8 GB/s), a table otherwise
uint32_t crc32c_table_at(uint32_t i) {
  uint32_t c = i;
  for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
  return c;
}
uint32_t crc32c_soft(uint32_t crc, const uint8_t *p, size_t n) {
  static uint32_t table[256];
  static std::atomic<int> ready(0);
  if (!ready.load(std::memory_order_acquire)) {
    for (uint32_t i = 0; i < 256; ++i) table[i] = crc32c_table_at(i);
    ready.store(1, std::memory_order_release);
  }
  for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
  return crc;
}
#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc32c_hw(uint32_t crc, const uint8_t *p, size_t n) {
  uint64_t c = crc;
  while (n && ((uintptr_t)p & 7)) { c = __builtin_ia32_crc32qi((uint32_t)c, *p++); --n; }
  for (; n >= 8; n -= 8, p += 8) { uint64_t v; std::memcpy(&v, p, 8); c = __builtin_ia32_crc32di(c, v); }
  while (n--) c = __builtin_ia32_crc32qi((uint32_t)c, *p++);
  return (uint32_t)c;
}
#endif
}  // namespace

extern "C" uint32_t nfst_crc32c(const void *data, int64_t n_bytes, uint32_t seed) {
  if (!data || n_bytes <= 0) return seed;
  uint32_t crc = ~seed;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("sse4.2")) return ~crc32c_hw(crc, (const uint8_t *)data, (size_t)n_bytes);
#endif
  return ~crc32c_soft(crc, (const uint8_t *)data, (size_t)n_bytes);
}

// Everything a kernel turns into an address without looking: offsets and counts of the meta records, row pointers,
// state / label / arc ids of the canonical arrays, of every tile's control words and records, of the slot -> arc maps.
// Host pointers.  O(words of the batch).  A sidecar file that passes this cannot make a kernel read or write outside
// the batch's arrays or outside the LDS rows the launchers size from max_rows / vocab.
extern "C" int nfst_validate_batch(const nfst_batch *lat, int32_t *err_lattice) {
  if (err_lattice) *err_lattice = -1;
  if (!lat || lat->n_lattices <= 0 || lat->vocab <= 0 || lat->vocab > NFST_MAX_VOCAB || lat->max_rows <= 0 || lat->max_rows > NFST_MAX_ROWS)
    return NFST_ERR_ARG;
  if (!lat->meta || !lat->row_ptr || !lat->fwd_stream || !lat->bwd_stream || !lat->arc_sd || !lat->arc_l16) return NFST_ERR_ARG;
  if (lat->total_arcs > 0 && (!lat->arc_src || !lat->arc_dst || !lat->arc_label)) return NFST_ERR_ARG;
  if ((lat->fwd_slots > 0 && !lat->fwd_perm) || (lat->bwd_slots > 0 && !lat->bwd_perm) || (lat->weighted && !lat->arc_w)) return NFST_ERR_ARG;
  if (lat->total_rows < 0 || lat->total_arcs < 0 || lat->fwd_words < 512 || lat->bwd_words < 512 || lat->fwd_slots < 0 || lat->bwd_slots < 0) return NFST_ERR_ARG;
  const int B = lat->n_lattices, V = lat->vocab;
  bool all_compact = true;
  int64_t max_arcs = 0, dp_total = 0;
  for (int b = 0; b < B; ++b) {
    auto fail = [&](int code) { if (err_lattice) *err_lattice = b; return code; };
    const int32_t *m = lat->meta + (size_t)b * NFST_META_WORDS;
    const int64_t row_off = m[NFST_META_ROW_OFF], n = m[NFST_META_N_ROWS], a0 = m[NFST_META_ARC_OFF], na = m[NFST_META_N_ARCS];
    if (row_off < 0 || n <= 0 || n > lat->max_rows || row_off + n > lat->total_rows) return fail(NFST_ERR_INDEX);
    if (a0 < 0 || na < 0 || a0 + na > lat->total_arcs) return fail(NFST_ERR_INDEX);
    if (m[NFST_META_SINK] < 0 || m[NFST_META_SINK] >= n || m[NFST_META_N_REACH] < 1 || m[NFST_META_N_REACH] > n) return fail(NFST_ERR_INDEX);
    if (m[NFST_META_N_DP] < 0 || m[NFST_META_N_DP] > na || m[NFST_META_DEPTH] < 0 || m[NFST_META_DEPTH] >= n) return fail(NFST_ERR_INDEX);
    max_arcs = std::max(max_arcs, na);
    dp_total += m[NFST_META_N_DP];
    // canonical arcs
    const int32_t *rp = lat->row_ptr + row_off + b;
    if (rp[0] != a0 || rp[n] != a0 + na) return fail(NFST_ERR_INDEX);
    for (int64_t s = 0; s < n; ++s)
      if (rp[s + 1] < rp[s]) return fail(NFST_ERR_INDEX);
    for (int64_t a = a0; a < a0 + na; ++a) {
      const int32_t s = lat->arc_src[a], d = lat->arc_dst[a], l = lat->arc_label[a];
      if (s < 0 || s >= n || d < 0 || d >= n || l < 0 || l >= V) return fail(NFST_ERR_INDEX);
      if (a < rp[s] || a >= rp[s + 1]) return fail(NFST_ERR_INDEX);
      if (lat->arc_sd[a] != ((uint32_t)s | ((uint32_t)d << 16)) || lat->arc_l16[a] != (uint16_t)l) return fail(NFST_ERR_INDEX);
    }
    // tile programs and slot -> arc maps
    for (int dir = 0; dir < 2; ++dir) {
      const int64_t off = m[dir ? NFST_META_BWD_OFF : NFST_META_FWD_OFF], tiles = m[dir ? NFST_META_BWD_TILES : NFST_META_FWD_TILES];
      const int code = m[dir ? NFST_META_BWD_U : NFST_META_FWD_U] & 0xff;
      const int64_t slot_off = m[dir ? NFST_META_BWD_SLOT_OFF : NFST_META_FWD_SLOT_OFF];
      const int64_t words = dir ? lat->bwd_words : lat->fwd_words, slots = dir ? lat->bwd_slots : lat->fwd_slots;
      const uint32_t *stream = dir ? lat->bwd_stream : lat->fwd_stream;
      const int32_t *perm = dir ? lat->bwd_perm : lat->fwd_perm;
      if (code != 1 && code != 2 && code != 4 && code != 8) return fail(NFST_ERR_ARG);
      all_compact = all_compact && code == 8;
      const int U = code == 8 ? 4 : code, tw = code == 8 ? 256 : 64 * (1 + code);
      if (tiles < 0 || tiles > lat->max_tiles || off < 0 || (off & 63) || off + tiles * tw > words - 512) return fail(NFST_ERR_INDEX);
      if (slot_off < 0 || slot_off + tiles * 64 * U > slots) return fail(NFST_ERR_INDEX);
      const uint32_t max_state = (uint32_t)lat->max_rows, max_label = (uint32_t)V + 1;
      for (int64_t t = 0; t < tiles; ++t) {
        const uint32_t *w = stream + off + t * tw;
        for (int l = 0; l < 64; ++l) {
          uint32_t ctl, rec_state[4], rec_label[4];
          if (code == 8) {
            const uint32_t *x = w + 4 * l;
            ctl = x[0];
            const uint32_t r[4] = {x[1] & 0xffffffu, ((x[1] >> 24) | (x[2] << 8)) & 0xffffffu, ((x[2] >> 16) | (x[3] << 16)) & 0xffffffu, x[3] >> 8};
            for (int j = 0; j < 4; ++j) { rec_state[j] = r[j] & 0x1fffu; rec_label[j] = r[j] >> 13; }
          } else {
            ctl = w[l];
            for (int j = 0; j < U; ++j) { const uint32_t r = w[64 + l * U + j]; rec_state[j] = (r & 0xffffu) >> 3; rec_label[j] = r >> 16; }
          }
          const uint32_t g = (ctl >> 20) & 7u, gmax = (ctl >> 23) & 7u;
          if (((ctl & 0xffffu) >> 3) >= max_state || (ctl & 7u) || g > 6 || gmax > 6 || g > gmax) return fail(NFST_ERR_INDEX);
          for (int j = 0; j < U; ++j)
            if (rec_state[j] >= max_state || rec_label[j] > max_label) return fail(NFST_ERR_INDEX);
        }
        const int32_t *pm = perm + slot_off + t * 64 * U;
        for (int q = 0; q < 64 * U; ++q)
          if (pm[q] != -1 && (pm[q] < a0 || pm[q] >= a0 + na)) return fail(NFST_ERR_INDEX);
      }
    }
  }
  if (((lat->reserved0 & NFST_BATCH_ALL_COMPACT) != 0) && !all_compact) return NFST_ERR_ARG;
  const int64_t rec_max = ((int64_t)lat->reserved0 >> NFST_BATCH_MAX_ARCS_SHIFT) & NFST_BATCH_MAX_ARCS_CAP;
  if (rec_max < std::min<int64_t>(max_arcs, NFST_BATCH_MAX_ARCS_CAP)) return NFST_ERR_ARG;  // (the launchers size LDS-resident per-arc data with it)
  if (dp_total != lat->total_dp_arcs) return NFST_ERR_ARG;
  return NFST_OK;
}

// Concatenation of packed batches without running the packer again (pack every example once, build a step's
// batch from sidecars): memcpy + offset fix-ups, parallel over the parts.  nfst_concat_sizes fills the scalar fields of
// *total; the caller allocates the arrays (ordinary or page-locked host memory), stores their addresses in *total and
// calls nfst_concat_packed, which writes through them.
extern "C" int nfst_concat_sizes(const nfst_batch *parts, int32_t n_parts, nfst_batch *total) {
  if (!parts || n_parts <= 0 || !total) return NFST_ERR_ARG;
  nfst_batch t{};
  t.vocab = parts[0].vocab; t.weighted = parts[0].weighted;
  int all_compact = 1;
  int64_t max_arcs = 0;
  const int64_t slack = 512;
  for (int i = 0; i < n_parts; ++i) {
    const nfst_batch &p = parts[i];
    if (p.vocab != t.vocab || p.weighted != t.weighted || p.n_lattices <= 0 || p.fwd_words < slack || p.bwd_words < slack) return NFST_ERR_ARG;
    t.n_lattices += p.n_lattices; t.max_rows = std::max(t.max_rows, p.max_rows); t.max_tiles = std::max(t.max_tiles, p.max_tiles);
    t.total_rows += p.total_rows; t.total_arcs += p.total_arcs; t.total_dp_arcs += p.total_dp_arcs;
    t.fwd_words += p.fwd_words - slack; t.bwd_words += p.bwd_words - slack; t.fwd_slots += p.fwd_slots; t.bwd_slots += p.bwd_slots;
    all_compact &= (p.reserved0 & NFST_BATCH_ALL_COMPACT) ? 1 : 0;
    max_arcs = std::max<int64_t>(max_arcs, ((int64_t)p.reserved0 >> NFST_BATCH_MAX_ARCS_SHIFT) & NFST_BATCH_MAX_ARCS_CAP);
  }
  if (std::max({t.total_arcs, t.fwd_words, t.bwd_words, t.fwd_slots, t.bwd_slots, t.total_rows}) > 0x7ffff000ll) return NFST_ERR_LIMIT;
  t.fwd_words += slack; t.bwd_words += slack;
  t.reserved0 = (all_compact ? NFST_BATCH_ALL_COMPACT : 0) | (int32_t)(max_arcs << NFST_BATCH_MAX_ARCS_SHIFT);
  *total = t;
  return NFST_OK;
}

extern "C" int nfst_concat_packed(const nfst_batch *parts, int32_t n_parts, const nfst_batch *out, int32_t n_threads) {
  if (!parts || n_parts <= 0 || !out) return NFST_ERR_ARG;
  nfst_batch want{};
  int rc = nfst_concat_sizes(parts, n_parts, &want);
  if (rc) return rc;
  if (want.n_lattices != out->n_lattices || want.total_rows != out->total_rows || want.total_arcs != out->total_arcs ||
      want.fwd_words != out->fwd_words || want.bwd_words != out->bwd_words || want.fwd_slots != out->fwd_slots ||
      want.bwd_slots != out->bwd_slots || want.vocab != out->vocab || want.weighted != out->weighted)
    return NFST_ERR_ARG;
  if (!out->meta || !out->row_ptr || !out->fwd_stream || !out->bwd_stream || !out->arc_sd || !out->arc_l16) return NFST_ERR_ARG;
  if (out->total_arcs > 0 && (!out->arc_src || !out->arc_dst || !out->arc_label || (out->weighted && !out->arc_w))) return NFST_ERR_ARG;
  if ((out->fwd_slots > 0 && !out->fwd_perm) || (out->bwd_slots > 0 && !out->bwd_perm)) return NFST_ERR_ARG;
  struct Off { int64_t lat, rows, arcs, fw, bw, fs, bs; };
  std::vector<Off> off((size_t)n_parts);
  Off o{0, 0, 0, 0, 0, 0, 0};
  const int64_t slack = 512;
  for (int i = 0; i < n_parts; ++i) {
    off[i] = o;
    const nfst_batch &p = parts[i];
    o.lat += p.n_lattices; o.rows += p.total_rows; o.arcs += p.total_arcs; o.fw += p.fwd_words - slack; o.bw += p.bwd_words - slack;
    o.fs += p.fwd_slots; o.bs += p.bwd_slots;
  }
  auto W32 = [](const int32_t *p) { return const_cast<int32_t *>(p); };
  parallel_for(n_parts, n_threads, [&](int i) {
    const nfst_batch &p = parts[i];
    const Off &f = off[i];
    int32_t *meta = W32(out->meta) + (size_t)f.lat * NFST_META_WORDS;
    std::memcpy(meta, p.meta, (size_t)p.n_lattices * NFST_META_WORDS * 4);
    for (int b = 0; b < p.n_lattices; ++b) {
      int32_t *m = meta + (size_t)b * NFST_META_WORDS;
      m[NFST_META_ROW_OFF] += (int32_t)f.rows; m[NFST_META_ARC_OFF] += (int32_t)f.arcs;
      m[NFST_META_FWD_OFF] += (int32_t)f.fw; m[NFST_META_BWD_OFF] += (int32_t)f.bw;
      m[NFST_META_FWD_SLOT_OFF] += (int32_t)f.fs; m[NFST_META_BWD_SLOT_OFF] += (int32_t)f.bs;
    }
    int32_t *rp = W32(out->row_ptr) + f.rows + f.lat;
    const int64_t nrp = p.total_rows + p.n_lattices;
    for (int64_t k = 0; k < nrp; ++k) rp[k] = p.row_ptr[k] + (int32_t)f.arcs;
    const size_t A = (size_t)p.total_arcs;
    if (A) {
      std::memcpy(W32(out->arc_src) + f.arcs, p.arc_src, A * 4);
      std::memcpy(W32(out->arc_dst) + f.arcs, p.arc_dst, A * 4);
      std::memcpy(W32(out->arc_label) + f.arcs, p.arc_label, A * 4);
      if (out->weighted) std::memcpy(const_cast<float *>(out->arc_w) + f.arcs, p.arc_w, A * 4);
      std::memcpy(const_cast<uint32_t *>(out->arc_sd) + f.arcs, p.arc_sd, A * 4);
      std::memcpy(const_cast<uint16_t *>(out->arc_l16) + f.arcs, p.arc_l16, A * 2);
    }
    std::memcpy(const_cast<uint32_t *>(out->fwd_stream) + f.fw, p.fwd_stream, (size_t)(p.fwd_words - slack) * 4);
    std::memcpy(const_cast<uint32_t *>(out->bwd_stream) + f.bw, p.bwd_stream, (size_t)(p.bwd_words - slack) * 4);
    int32_t *fp = W32(out->fwd_perm) + f.fs, *bp = W32(out->bwd_perm) + f.bs;
    for (int64_t k = 0; k < p.fwd_slots; ++k) fp[k] = p.fwd_perm[k] < 0 ? -1 : p.fwd_perm[k] + (int32_t)f.arcs;
    for (int64_t k = 0; k < p.bwd_slots; ++k) bp[k] = p.bwd_perm[k] < 0 ? -1 : p.bwd_perm[k] + (int32_t)f.arcs;
  });
  std::memset(const_cast<uint32_t *>(out->fwd_stream) + (out->fwd_words - slack), 0, slack * 4);
  std::memset(const_cast<uint32_t *>(out->bwd_stream) + (out->bwd_words - slack), 0, slack * 4);
  std::memset(const_cast<uint32_t *>(out->arc_sd) + out->total_arcs, 0, 8 * 4);
  std::memset(const_cast<uint16_t *>(out->arc_l16) + out->total_arcs, 0, 8 * 2);
  return NFST_OK;
}

// Offsets and header of a batch planned on the device (the counterpart of finish() above): meta holds the counts the
// planning kernel wrote; the offsets are prefix sums in lattice order, exactly as the host packer lays the arrays out.
extern "C" int nfst_pack_device_layout(int32_t *meta, const int32_t *status, const int32_t *scratch_rows, int32_t n_lattices,
                                       int32_t vocab, int32_t weighted, nfst_batch *header, int32_t *err_lattice) {
  if (!meta || !status || !scratch_rows || !header || n_lattices <= 0 || vocab <= 0) return NFST_ERR_ARG;
  for (int b = 0; b < n_lattices; ++b)
    if (status[b] != NFST_OK) { if (err_lattice) *err_lattice = b; return status[b]; }
  nfst_batch h{};
  int64_t rows = 0, arcs = 0, dp = 0, fw = 0, bw = 0, fs = 0, bs = 0, max_arcs = 0;
  int max_rows = 0, max_tiles = 0;
  for (int b = 0; b < n_lattices; ++b) {
    int32_t *m = meta + (size_t)b * NFST_META_WORDS;
    const int64_t ft = m[NFST_META_FWD_TILES], bt = m[NFST_META_BWD_TILES];
    m[NFST_META_ROW_OFF] = (int32_t)rows; m[NFST_META_ARC_OFF] = (int32_t)arcs;
    m[NFST_META_FWD_OFF] = (int32_t)fw; m[NFST_META_BWD_OFF] = (int32_t)bw;
    m[NFST_META_FWD_SLOT_OFF] = (int32_t)fs; m[NFST_META_BWD_SLOT_OFF] = (int32_t)bs;
    rows += m[NFST_META_N_ROWS]; arcs += m[NFST_META_N_ARCS]; dp += m[NFST_META_N_DP];
    fw += ft * 256; bw += bt * 256; fs += ft * 256; bs += bt * 256;  // compact tiles: 256 words, 256 slots each
    max_rows = std::max(max_rows, m[NFST_META_N_ROWS] + scratch_rows[b]);
    max_tiles = std::max<int64_t>(max_tiles, std::max(ft, bt));
    max_arcs = std::max<int64_t>(max_arcs, m[NFST_META_N_ARCS]);
    if (arcs > 0x7fffff00ll || fw > 0x7ffff000ll || bw > 0x7ffff000ll || rows > 0x7fffff00ll) { if (err_lattice) *err_lattice = b; return NFST_ERR_LIMIT; }
  }
  const int64_t slack = 512;
  h.n_lattices = n_lattices; h.vocab = vocab; h.max_rows = max_rows; h.max_tiles = max_tiles; h.weighted = weighted ? 1 : 0;
  h.reserved0 = NFST_BATCH_ALL_COMPACT | (int32_t)(std::min<int64_t>(max_arcs, NFST_BATCH_MAX_ARCS_CAP) << NFST_BATCH_MAX_ARCS_SHIFT);
  h.total_rows = rows; h.total_arcs = arcs; h.total_dp_arcs = dp;
  h.fwd_words = fw + slack; h.bwd_words = bw + slack; h.fwd_slots = fs; h.bwd_slots = bs;
  *header = h;
  return NFST_OK;
}

extern "C" {

int nfst_pack_dense(const void *emission, int emission_is_float, const int64_t *transition,
                    int32_t n_lattices, int32_t n_rows, int32_t vocab, const nfst_pack_opts *opts,
                    nfst_packed **out, int32_t *err_lattice) {
  if (!emission || !transition || !out || n_lattices <= 0 || n_rows <= 0 || vocab <= 0) return NFST_ERR_ARG;
  if (n_rows > NFST_MAX_ROWS || vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  Opts o = read_opts(opts);
  std::vector<Lat> lats((size_t)n_lattices);
  const size_t cells = (size_t)n_rows * vocab;
  const uint8_t *eb = (const uint8_t *)emission;
  const float *ef = (const float *)emission;
  parallel_for(n_lattices, o.n_threads, [&](int b) {
    Lat &L = lats[b];
    L.n_rows = n_rows;
    const int64_t *tr = transition + (size_t)b * cells;
    auto has = [&](size_t at) {
      return emission_is_float ? (ef[(size_t)b * cells + at] > -std::numeric_limits<float>::infinity())
                               : (eb[(size_t)b * cells + at] != 0);
    };
    std::vector<uint8_t> seen(n_rows, 0);
    std::vector<int32_t> stack;
    seen[0] = 1; stack.push_back(0);
    size_t n_arcs = 0;
    while (!stack.empty()) {
      int s = stack.back(); stack.pop_back();
      for (int l = 0; l < vocab; ++l) {
        size_t at = (size_t)s * vocab + l;
        if (!has(at)) continue;
        int64_t d = tr[at];
        if (d < 0 || d >= n_rows) { L.err = NFST_ERR_INDEX; return; }
        ++n_arcs;
        if (!seen[d]) { seen[d] = 1; stack.push_back((int32_t)d); }
      }
    }
    L.src.reserve(n_arcs); L.label.reserve(n_arcs); L.dst.reserve(n_arcs);
    if (emission_is_float) L.w.reserve(n_arcs);
    for (int s = 0; s < n_rows; ++s) {
      if (!seen[s]) continue;
      for (int l = 0; l < vocab; ++l) {
        size_t at = (size_t)s * vocab + l;
        if (!has(at)) continue;
        L.src.push_back(s); L.label.push_back(l); L.dst.push_back((int32_t)tr[at]);
        if (emission_is_float) L.w.push_back(ef[(size_t)b * cells + at]);
      }
    }
  });
  return finish(lats, vocab, emission_is_float != 0, o, out, err_lattice);
}

int nfst_pack_arcs(const int32_t *n_rows, const int64_t *arc_off, const int32_t *src,
                   const int32_t *label, const int32_t *dst, const float *arc_w, int32_t n_lattices,
                   int32_t vocab, const nfst_pack_opts *opts, nfst_packed **out, int32_t *err_lattice) {
  if (!n_rows || !arc_off || !out || n_lattices <= 0 || vocab <= 0) return NFST_ERR_ARG;
  if (vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  if (arc_off[n_lattices] > 0 && (!src || !label || !dst)) return NFST_ERR_ARG;
  Opts o = read_opts(opts);
  std::vector<Lat> lats((size_t)n_lattices);
  parallel_for(n_lattices, o.n_threads, [&](int b) {
    Lat &L = lats[b];
    const int n = n_rows[b];
    L.n_rows = n;
    if (n <= 0 || n > NFST_MAX_ROWS) { L.err = n <= 0 ? NFST_ERR_ARG : NFST_ERR_LIMIT; return; }
    const int64_t a0 = arc_off[b], a1 = arc_off[b + 1];
    if (a1 < a0) { L.err = NFST_ERR_ARG; return; }
    std::vector<int32_t> rp(n + 1, 0);
    for (int64_t a = a0; a < a1; ++a) {
      if (src[a] < 0 || src[a] >= n || dst[a] < 0 || dst[a] >= n || label[a] < 0 || label[a] >= vocab) {
        L.err = NFST_ERR_INDEX; return;
      }
      if (a > a0) {
        if (src[a] < src[a - 1]) { L.err = NFST_ERR_ARG; return; }
        if (src[a] == src[a - 1] && label[a] <= label[a - 1]) { L.err = NFST_ERR_DETERMINISM; return; }
      }
      rp[src[a] + 1]++;
    }
    for (int s = 0; s < n; ++s) rp[s + 1] += rp[s];
    std::vector<uint8_t> seen(n, 0);
    std::vector<int32_t> stack;
    seen[0] = 1; stack.push_back(0);
    while (!stack.empty()) {
      int s = stack.back(); stack.pop_back();
      for (int q = rp[s]; q < rp[s + 1]; ++q) {
        int d = dst[a0 + q];
        if (!seen[d]) { seen[d] = 1; stack.push_back(d); }
      }
    }
    for (int64_t a = a0; a < a1; ++a) {
      if (!seen[src[a]]) continue;
      L.src.push_back(src[a]); L.label.push_back(label[a]); L.dst.push_back(dst[a]);
      if (arc_w) L.w.push_back(arc_w[a]);
    }
  });
  return finish(lats, vocab, arc_w != nullptr, o, out, err_lattice);
}

int nfst_packed_view(const nfst_packed *p, nfst_batch *view) {
  if (!p || !view) return NFST_ERR_ARG;
  *view = p->view;
  return NFST_OK;
}

void nfst_packed_free(nfst_packed *p) { delete p; }

int nfst_abi_version(void) { return NFST_ABI_VERSION; }

int nfst_sizeof(const char *name) {
  if (!name) return -1;
  const std::string n(name);
  if (n == "nfst_batch") return (int)sizeof(nfst_batch);
  if (n == "nfst_scores") return (int)sizeof(nfst_scores);
  if (n == "nfst_chunks") return (int)sizeof(nfst_chunks);
  if (n == "nfst_chunk_opts") return (int)sizeof(nfst_chunk_opts);
  if (n == "nfst_pack_opts") return (int)sizeof(nfst_pack_opts);
  if (n == "nfst_step_extras") return (int)sizeof(nfst_step_extras);
  if (n == "nfst_arcs_device") return (int)sizeof(nfst_arcs_device);
  return -1;
}

const char *nfst_strerror(int code) {
  switch (code) {
    case NFST_OK: return "ok";
    case NFST_ERR_ARG: return "invalid argument (null pointer, bad size or unsorted arcs)";
    case NFST_ERR_INDEX: return "state or label index out of range";
    case NFST_ERR_CYCLE: return "lattice is not acyclic";
    case NFST_ERR_SINK: return "lattice must have exactly one final (sink) state";
    case NFST_ERR_DETERMINISM: return "two arcs leave one state with the same label";
    case NFST_ERR_LIMIT: return "lattice exceeds engine limits (rows <= 8192, vocab <= 32767)";
    case NFST_ERR_HIP: return "HIP runtime error";
    case NFST_ERR_NOMEM: return "out of memory";
    case NFST_ERR_LENGTH: return "ran out of length budget";
    default: return "unknown error";
  }
}

}  // extern "C"
