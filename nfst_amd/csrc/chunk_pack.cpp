// Host side of the chunked flavour for deep, narrow lattices (include/nfst_hip.h, "Chunked programs"): puts the states
// of a lattice in topological order per direction, cuts the positions into chunks and writes one entry per arc in the
// order a chunk's lanes walk them.  The kernels are in chunk_kernels.h.
//
// Reference shape this serves: the SNIPS tagging machines (/root/reference/src/main_snips.py, conf/train/lstm_snips.yaml:2
// max_length 750; decode/decoder.py:77-79 walks them state by state): a few tag states per token position.
#include "../../include/nfst_hip.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

struct nfst_chunks_host {
  nfst_chunks view{};
  std::vector<int32_t> meta, tab, pos;
  std::vector<uint32_t> stream;
  std::vector<uint16_t> label;
};

namespace {

constexpr int kCus = 256;        // MI355X
constexpr int kMaxReach = 63;    // an entry's operand slot has 6 bits
// cost model of the chunked sweep (cycles at 2.4 GHz, MI355X, measured with profiles/tune/chunk_stamps.py: DESIGN.md
// section 4.4): pass 1 is the longest chunk at one entry per kEntry cycles (an LDS round trip per entry on a lane's chain:
// 82 ns with one wave per SIMD, 88 ns with three), or -- a bound for workgroups full of lanes -- the whole program's entries
// x F / 64 lanes x kIssue / 4 SIMDs; pass 2 is C steps of step_cycles(F); kFixed: weights, initialisation, the tail of pass 3,
// the second kernel and the empty launch of the general one
constexpr double kEntry = 205.0, kIssue = 60.0, kFixed = 25000.0;
inline double step_cycles(int F) { return (F <= 2 ? 560.0 : F <= 4 ? 600.0 : 775.0) + 300.0 * ((F + 7) / 8 - 1); }  // (234 / 250 / 322 / 445 ns at F = 2 / 3 / 8 / 16)

struct Prog {
  int C = 0, F = 0, R = 0, npos = 0;
  std::vector<int32_t> tab, pos;
  std::vector<uint32_t> stream;
  double cycles = 0.0;           // cost model of the chunked sweep
};

inline int pow2_at_least(int x) { int r = 1; while (r < x) r <<= 1; return r; }

template <class Fn>
void parallel_for(int n, int n_threads, Fn f) {
  if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
  n_threads = std::max(1, std::min(n_threads, n));
  if (n_threads == 1) { for (int i = 0; i < n; ++i) f(i); return; }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; ++t)
    th.emplace_back([&] { for (int i; (i = next.fetch_add(1)) < n;) f(i); });
  for (auto &x : th) x.join();
}

// LDS of a workgroup that runs a program with C chunks of F right-hand sides and R ring slots: the rings (one padded
// block per chunk), the frontier values of every chunk (mantissa + exponent) and the chunks' first positions
inline int64_t lds_need(int C, int F, int R) {
  return (int64_t)C * (R * F + F) * 8 + 64 + (int64_t)C * F * 12 + (int64_t)(2 * C + 2) * 4 + 64;
}

// One direction of one lattice.  level[s]: longest path from the start (alpha) / to the sink (beta); operands of a
// state: the sources of its in-arcs (alpha) / the destinations of its out-arcs (beta), as lists of canonical arcs.
bool cut(int n_rows, const std::vector<int32_t> &level, const std::vector<uint8_t> &reach, const std::vector<int32_t> &ptr,
         const std::vector<int32_t> &list, const int32_t *operand_of_arc, int threads, int64_t lds_bytes, int max_chunks, Prog &out) {
  std::vector<int32_t> order;
  for (int s = 0; s < n_rows; ++s)
    if (reach[s]) order.push_back(s);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[a] < level[b]; });
  const int n = (int)order.size();
  if (n < 2) return false;
  std::vector<int32_t> posof(n_rows, -1);
  for (int p = 0; p < n; ++p) posof[order[p]] = p;
  // the longest reach of an arc, and per cut a (a chunk starting at position a) the deepest operand below it
  int W = 0;
  std::vector<int32_t> low(n + 1);
  for (int a = 0; a <= n; ++a) low[a] = a;  // low[a] = smallest operand position of an arc that crosses the cut
  std::vector<int64_t> pre(n + 1, 0);       // entries of positions [1, p)
  for (int p = 1; p < n; ++p) {
    const int s = order[p];
    const int cnt = ptr[s + 1] - ptr[s];
    pre[p + 1] = pre[p] + std::max(cnt, 1);
    for (int j = ptr[s]; j < ptr[s + 1]; ++j) {
      const int q = posof[operand_of_arc[list[j]]];
      if (q < 0 || q >= p) return false;  // (not a topological order: cannot happen for a packed batch)
      W = std::max(W, p - q);
      if (p - q > kMaxReach) return false;
      for (int a = q + 1; a <= p; ++a) low[a] = std::min(low[a], q);
    }
  }
  pre[1] = 0;
  const int R = std::max(4, pow2_at_least(W + 1));
  const int64_t total = pre[n];
  auto reach_of = [&](int a) { return a - low[a]; };  // frontier of a chunk that starts at a
  int Fb = 1;
  for (int a = 1; a < n; ++a) Fb = std::max(Fb, reach_of(a));
  // Cuts for a frontier of at most Ft states: as many chunks as the lanes, the LDS and the balance of pass 1 against pass 2
  // allow, about the same number of entries each; a cut whose frontier is wider than Ft moves to the nearest position
  // where it is not (or is dropped).  The cheapest plan over all Ft by the cost model wins.
  struct Plan { std::vector<int32_t> starts; int F = 0; double cycles = 0.0; };
  auto plan_for = [&](int Ft, int lane_cap, Plan &pl) {
    int C = std::min(std::min(threads, lane_cap) / Ft, n - 1);
    C = std::min(C, std::max(1, (int)std::ceil(std::sqrt((double)total * kEntry / step_cycles(Ft)))));
    while (C > 1 && lds_need(C, Ft, R) > lds_bytes) --C;
    if (max_chunks > 0) C = std::min(C, max_chunks);
    if (C < 1 || lds_need(C, Ft, R) > lds_bytes) return false;
    pl.starts.assign(1, 1);
    const int slack = std::max(1, (n / C) / 2);
    for (int c = 1; c < C; ++c) {
      const int64_t target = total * c / C;
      int a = (int)(std::upper_bound(pre.begin() + 1, pre.begin() + n + 1, target) - pre.begin()) - 1;
      a = std::max(a, pl.starts.back() + 1);
      if (a >= n) break;
      int best = -1;
      for (int d = 0; d <= slack && best < 0; ++d)
        for (int sgn = -1; sgn <= 1 && best < 0; sgn += 2) {
          const int b = a + sgn * d;
          if (b > pl.starts.back() && b < n && reach_of(b) <= Ft) best = b;
        }
      if (best > 0) pl.starts.push_back(best);
    }
    pl.F = 1;
    for (int a : pl.starts) pl.F = std::max(pl.F, reach_of(a));
    int64_t longest = 0;
    for (size_t c = 0; c < pl.starts.size(); ++c) {
      const int64_t cnt = pre[c + 1 < pl.starts.size() ? pl.starts[c + 1] : n] - pre[pl.starts[c]];
      longest = std::max(longest, (cnt + 7) / 8 * 8);
    }
    // an entry of pass 1 is an LDS round trip on its lane's chain while a SIMD holds at most two of the workgroup's waves
    // (82 .. 88 ns); with a third one the walks share its issue slots (88 .. 113 ns with nine waves, 112 .. 139 with twelve)
    const int waves = ((int)pl.starts.size() * pl.F + 63) / 64, per_simd = (waves + 3) / 4;
    const double entry = kEntry + 60.0 * std::max(0, per_simd - 2);
    pl.cycles = std::max((double)longest * entry, (double)total * pl.F / 64.0 * kIssue / 4.0) +
                (double)pl.starts.size() * step_cycles(pl.F) + kFixed;
    return true;
  };
  Plan best;
  for (int Ft = Fb; Ft >= 1; --Ft)
    for (int lane_cap : {1024, 512}) {  // (fewer chunks can be faster: two waves per SIMD)
      Plan pl;
      if (!plan_for(Ft, lane_cap, pl)) continue;
      if (best.F == 0 || pl.cycles < best.cycles) best = pl;
    }
  if (best.F == 0) return false;
  const std::vector<int32_t> &starts = best.starts;
  const int F = best.F;
  const int C = (int)starts.size();
  if (C * F > threads || lds_need(C, F, R) > lds_bytes || F > R) return false;
  out.C = C; out.F = F; out.R = R; out.npos = n;
  out.pos.assign(order.begin(), order.end());
  out.tab.clear(); out.stream.clear();
  for (int c = 0; c < C; ++c) {
    const int a = starts[c], b = c + 1 < C ? starts[c + 1] : n;
    const int64_t begin = (int64_t)out.stream.size();
    for (int p = a; p < b; ++p) {
      const int s = order[p];
      const int cnt = ptr[s + 1] - ptr[s];
      if (cnt == 0) out.stream.push_back(NFST_CHK_LAST | NFST_CHK_ZERO);
      for (int j = 0; j < cnt; ++j) {
        const int arc = list[j + ptr[s]];
        const int q = posof[operand_of_arc[arc]];
        out.stream.push_back((uint32_t)(q & (R - 1)) | (j + 1 == cnt ? NFST_CHK_LAST : 0u) | ((uint32_t)arc << 8));
      }
    }
    // (a chunk's entries are walked eight at a time without a bounds test: padded with zero-weight entries)
    while ((out.stream.size() - begin) % 8) out.stream.push_back(NFST_CHK_ZERO);
    const int64_t count = (int64_t)out.stream.size() - begin;
    out.tab.push_back(a); out.tab.push_back((int32_t)begin); out.tab.push_back((int32_t)count); out.tab.push_back(0);
  }
  out.cycles = best.cycles;
  return true;
}

}  // namespace

extern "C" int nfst_pack_chunks(const nfst_batch *hb, const nfst_chunk_opts *opts, nfst_chunks_host **out) {
  if (!hb || !out) return NFST_ERR_ARG;
  *out = nullptr;
  if (hb->n_lattices <= 0 || !hb->meta || !hb->arc_src || !hb->arc_dst || !hb->arc_label) return NFST_ERR_ARG;
  nfst_chunk_opts o{};
  if (opts) o = *opts;
  const int B = hb->n_lattices;
  const bool roomy = 2 * B <= kCus;  // every (lattice, direction) workgroup has a CU to itself
  const int threads = o.threads > 0 ? o.threads : (roomy ? 1024 : 512);
  const int64_t lds_bytes = o.lds_bytes > 0 ? o.lds_bytes : (roomy ? 152 * 1024 : 64 * 1024);
  if (threads < 64 || threads > 1024 || (threads & 63) || lds_bytes > 160 * 1024) return NFST_ERR_ARG;
  // a quick no: up to ~160 levels the general kernels are done before the fixed costs of this flavour are (the BASELINE shape:
  // 130 .. 150 tiles)
  if (!o.force && hb->max_tiles <= 160) return NFST_OK;
  nfst_chunks_host *h = new (std::nothrow) nfst_chunks_host();
  if (!h) return NFST_ERR_NOMEM;
  h->meta.assign((size_t)B * 2 * NFST_CHK_META_WORDS, 0);
  int64_t t_units = 0;
  double cycles_chunked = 0.0, cycles_general = 0.0;
  int64_t lds_used = 0;
  // per lattice, on host threads: both programs (or "cannot be cut"), then the programs are laid out one behind the other
  struct One { int err = NFST_OK; bool ok = false; Prog p[2]; };
  std::vector<One> ones(B);
  parallel_for(B, o.n_threads, [&](int b) {
    One &one = ones[b];
    const int32_t *m = hb->meta + (size_t)b * NFST_META_WORDS;
    const int n = m[NFST_META_N_ROWS], A = m[NFST_META_N_ARCS];
    const int32_t *src = hb->arc_src + m[NFST_META_ARC_OFF], *dst = hb->arc_dst + m[NFST_META_ARC_OFF];
    if (A >= (1 << 24)) return;
    std::vector<uint8_t> reach(n, 0);
    reach[0] = 1;
    std::vector<int32_t> in_ptr(n + 1, 0), out_ptr(n + 1, 0);
    for (int a = 0; a < A; ++a) {
      if (src[a] < 0 || src[a] >= n || dst[a] < 0 || dst[a] >= n) { one.err = NFST_ERR_INDEX; return; }
      reach[src[a]] = 1; reach[dst[a]] = 1;
      if (src[a] != dst[a]) { in_ptr[dst[a] + 1]++; out_ptr[src[a] + 1]++; }
    }
    for (int s = 0; s < n; ++s) { in_ptr[s + 1] += in_ptr[s]; out_ptr[s + 1] += out_ptr[s]; }
    std::vector<int32_t> in_list(in_ptr[n]), out_list(out_ptr[n]);
    {
      std::vector<int32_t> ip(in_ptr.begin(), in_ptr.end() - 1), op(out_ptr.begin(), out_ptr.end() - 1);
      for (int a = 0; a < A; ++a)
        if (src[a] != dst[a]) { in_list[ip[dst[a]]++] = a; out_list[op[src[a]]++] = a; }
    }
    // longest path from the start / to the sink (Kahn; the packer has already refused cycles)
    std::vector<int32_t> depth(n, 0), height(n, 0), rem(n), order;
    for (int s = 0; s < n; ++s) rem[s] = in_ptr[s + 1] - in_ptr[s];
    order.push_back(0);
    for (size_t i = 0; i < order.size(); ++i) {
      const int s = order[i];
      for (int j = out_ptr[s]; j < out_ptr[s + 1]; ++j) {
        const int d = dst[out_list[j]];
        depth[d] = std::max(depth[d], depth[s] + 1);
        if (--rem[d] == 0) order.push_back(d);
      }
    }
    int n_reach = 0;
    for (int s = 0; s < n; ++s) n_reach += reach[s];
    if ((int)order.size() != n_reach) { one.err = NFST_ERR_CYCLE; return; }
    for (int i = n_reach - 1; i >= 0; --i) {
      const int s = order[i];
      for (int j = out_ptr[s]; j < out_ptr[s + 1]; ++j) height[s] = std::max(height[s], height[dst[out_list[j]]] + 1);
    }
    for (int dir = 0; dir < 2; ++dir) {
      Prog &p = one.p[dir];
      const bool ok = dir == 0 ? cut(n, depth, reach, in_ptr, in_list, src, threads, lds_bytes, o.max_chunks, p)
                               : cut(n, height, reach, out_ptr, out_list, dst, threads, lds_bytes, o.max_chunks, p);
      // position 0 must be the start (alpha) / the sink (beta)
      if (!ok || p.pos[0] != (dir == 0 ? 0 : m[NFST_META_SINK])) return;
    }
    one.ok = true;
  });
  for (int b = 0; b < B; ++b) {
    if (ones[b].err != NFST_OK) { const int err = ones[b].err; delete h; return err; }
    if (!ones[b].ok) { delete h; return NFST_OK; }
  }
  for (int b = 0; b < B; ++b) {
    const int32_t *m = hb->meta + (size_t)b * NFST_META_WORDS;
    for (int dir = 0; dir < 2; ++dir) {
      const Prog &p = ones[b].p[dir];
      int32_t *cm = h->meta.data() + ((size_t)b * 2 + dir) * NFST_CHK_META_WORDS;
      cm[NFST_CHK_C] = p.C; cm[NFST_CHK_F] = p.F; cm[NFST_CHK_R] = p.R; cm[NFST_CHK_NPOS] = p.npos;
      cm[NFST_CHK_TAB_OFF] = (int32_t)(h->tab.size() / 4);
      cm[NFST_CHK_STREAM_OFF] = (int32_t)h->stream.size();
      cm[NFST_CHK_POS_OFF] = (int32_t)h->pos.size();
      cm[NFST_CHK_T_OFF] = (int32_t)t_units;
      t_units += ((int64_t)p.npos * p.F + 63) / 64;
      if (t_units > INT32_MAX || h->stream.size() + p.stream.size() > (size_t)INT32_MAX) { delete h; return NFST_OK; }
      h->tab.insert(h->tab.end(), p.tab.begin(), p.tab.end());
      h->stream.insert(h->stream.end(), p.stream.begin(), p.stream.end());
      h->pos.insert(h->pos.end(), p.pos.begin(), p.pos.end());
      lds_used = std::max(lds_used, lds_need(p.C, p.F, p.R));
      cycles_chunked = std::max(cycles_chunked, p.cycles * (roomy ? 1.0 : 2.0 * B / kCus));
    }
    // the general flavour: a chain of tiles (precise flavour beyond 192 tiles), one lattice per CU
    const int tiles = std::max(m[NFST_META_FWD_TILES], m[NFST_META_BWD_TILES]);
    cycles_general = std::max(cycles_general, tiles * (tiles > 192 ? 500.0 : 400.0) * std::max(1.0, (double)B / kCus) + 12000.0);
  }
  if (!o.force && cycles_chunked > 0.9 * cycles_general) { delete h; return NFST_OK; }  // (profiles/tune/chunk_auto.py: the choice against measurements)
  h->stream.resize(h->stream.size() + 64, 0);  // (slack: a lane of pass 1 reads up to 24 entries ahead)
  // the label of every entry's arc, beside the entry
  h->label.assign(h->stream.size(), 0);
  for (int b = 0; b < B; ++b) {
    const int32_t *m = hb->meta + (size_t)b * NFST_META_WORDS;
    for (int dir = 0; dir < 2; ++dir) {
      const int32_t *cm = h->meta.data() + ((size_t)b * 2 + dir) * NFST_CHK_META_WORDS;
      const int32_t *tb = h->tab.data() + (size_t)cm[NFST_CHK_TAB_OFF] * 4;
      const int64_t n_e = tb[(cm[NFST_CHK_C] - 1) * 4 + 1] + tb[(cm[NFST_CHK_C] - 1) * 4 + 2];
      for (int64_t k = 0; k < n_e; ++k) {
        const uint32_t e = h->stream[cm[NFST_CHK_STREAM_OFF] + k];
        if (!(e & NFST_CHK_ZERO)) h->label[cm[NFST_CHK_STREAM_OFF] + k] = (uint16_t)hb->arc_label[m[NFST_META_ARC_OFF] + (e >> 8)];
      }
    }
  }
  nfst_chunks &v = h->view;
  v.n_lattices = B; v.threads = threads; v.lds_bytes = (int32_t)((lds_used + 255) & ~(int64_t)255); v.launches = 0;
  v.n_tab = (int64_t)h->tab.size() / 4; v.n_stream = (int64_t)h->stream.size(); v.n_pos = (int64_t)h->pos.size();
  v.t_units = t_units; v.total_rows = hb->total_rows; v.total_arcs = hb->total_arcs;
  v.meta = h->meta.data(); v.tab = h->tab.data(); v.stream = h->stream.data(); v.pos = h->pos.data(); v.label = h->label.data();
  v.ws = nullptr; v.ws_bytes = 0;
  *out = h;
  return NFST_OK;
}

extern "C" int nfst_chunks_view(const nfst_chunks_host *c, nfst_chunks *view) {
  if (!c || !view) return NFST_ERR_ARG;
  *view = c->view;
  return NFST_OK;
}

extern "C" void nfst_chunks_free(nfst_chunks_host *c) { delete c; }

// scratch: the entries with their weights (16 B each), T, the (mantissa, exponent) values of every row per direction,
// Z per lattice, the flags
extern "C" int64_t nfst_chunks_ws_bytes(const nfst_chunks *c) {
  if (!c) return NFST_ERR_ARG;
  return c->n_stream * 16 + c->t_units * 64 * 8 + c->total_rows * 2 * 16 + (int64_t)c->n_lattices * 16 + (int64_t)c->n_lattices * 4 + 1024;
}
