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
#include <vector>

#include "nfst_hip.h"

namespace {

struct Opts {
  int n_threads = 0;
  int max_step_words = NFST_MAX_STEP_WORDS;
  int lanes_policy = 0;
  int sweep_waves = 4;
};

struct Lat {
  int n_rows = 0;
  // canonical arcs of reachable states, sorted by (src, label)
  std::vector<int32_t> src, label, dst;
  std::vector<float> w;
  // results
  std::vector<int32_t> row_ptr;  // n_rows + 1, relative
  std::vector<uint32_t> fwd, bwd;
  std::vector<int32_t> fwd_perm, bwd_perm;  // relative canonical arc ids
  int fwd_steps = 0, bwd_steps = 0, sink = 0, n_reach = 0, depth = 0, n_dp = 0;
  int err = NFST_OK;
};

// Lanes per state (2^kl) for a step of n_states states whose largest degree is
// maxdeg, run by W waves: minimise the instructions on one wave's critical path,
//   tiles(kl) * (iters(kl) * c_iter + kl * c_reduce + c_fixed),
// tiles = ceil(n_states * 2^kl / (64 W)), iters = ceil(maxdeg / 2^kl).
// lanes_policy 1 ("throughput") instead keeps lanes busy: the smallest kl with
// iters <= 4.
int choose_klog(int n_states, int maxdeg, const Opts &o) {
  auto iters = [&](int kl) { return (maxdeg + (1 << kl) - 1) >> kl; };
  if (o.lanes_policy == 1) {
    int kl = 0;
    while (kl < 6 && iters(kl) > 4) ++kl;
    return kl;
  }
  const int64_t lanes = 64 * (int64_t)o.sweep_waves;
  int best = 0;
  int64_t best_cost = -1;
  for (int kl = 0; kl <= 6; ++kl) {
    const int64_t tiles = (((int64_t)n_states << kl) + lanes - 1) / lanes;
    const int it = iters(kl);
    // the first two arcs of a lane share one rescale; later ones use the online rule
    const int64_t per_tile = 12 * std::min(it, 2) + 16 * std::max(it - 2, 0) + 3 * kl + (kl > 4 ? 12 * (kl - 4) : 0) + 24;
    const int64_t cost = tiles * per_tile;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = kl; }
  }
  return best;
}

// Emits the steps of one level.  states are sorted by degree (desc).  arcs_of(s)
// gives [begin,end) into `list` (arc ids), other(a) the state stored in the record.
template <class ArcsOf, class Other>
void emit_level(const std::vector<int32_t> &states, const Opts &o, ArcsOf arcs_of, Other other,
                const std::vector<int32_t> &list, const std::vector<int32_t> &label,
                std::vector<uint32_t> &stream, std::vector<int32_t> &perm, int &n_steps) {
  const int max_words = std::max(o.max_step_words, 16);
  const int max_chunk = std::min(max_words - 4, 65535);
  size_t i = 0, n = states.size();
  auto put_rec = [&](int32_t a) {
    stream.push_back((uint32_t)other(a) | ((uint32_t)label[a] << 16));
    perm.push_back(a);
  };
  while (i < n) {
    int s = states[i];
    auto r = arcs_of(s);
    int deg = r.second - r.first;
    if (deg > max_chunk) {
      // a state too large for one step: partial sums over consecutive steps, the
      // later ones flagged "accumulate" (bit 20)
      for (int c = 0; c < deg; c += max_chunk) {
        int cnt = std::min(max_chunk, deg - c);
        Opts oo = o;
        int kl = choose_klog(1, cnt, oo);
        stream.push_back(1u | ((uint32_t)kl << 16) | (c > 0 ? (1u << 20) : 0u));
        stream.push_back((uint32_t)cnt);
        stream.push_back((uint32_t)s | (0u << 16));
        stream.push_back(0xFFFFu | ((uint32_t)cnt << 16));
        for (int k = 0; k < cnt; ++k) put_rec(list[r.first + c + k]);
        ++n_steps;
      }
      ++i;
      continue;
    }
    size_t j = i;
    int words = 3, arcs = 0, maxdeg = 0;
    while (j < n) {
      auto rr = arcs_of(states[j]);
      int d = rr.second - rr.first;
      if (d > max_chunk) break;
      if (j > i && (words + 1 + d > max_words || arcs + d > 65535 || (j - i) >= 65535)) break;
      words += 1 + d;
      arcs += d;
      maxdeg = std::max(maxdeg, d);
      ++j;
    }
    int ns = (int)(j - i);
    int kl = choose_klog(ns, std::max(maxdeg, 1), o);
    stream.push_back((uint32_t)ns | ((uint32_t)kl << 16));
    stream.push_back((uint32_t)arcs);
    uint32_t off = 0;
    for (size_t q = i; q < j; ++q) {
      auto rr = arcs_of(states[q]);
      stream.push_back((uint32_t)states[q] | (off << 16));
      off += (uint32_t)(rr.second - rr.first);
    }
    stream.push_back(0xFFFFu | (off << 16));
    for (size_t q = i; q < j; ++q) {
      auto rr = arcs_of(states[q]);
      for (int a = rr.first; a < rr.second; ++a) put_rec(list[a]);
    }
    ++n_steps;
    i = j;
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
  L.fwd.clear(); L.bwd.clear(); L.fwd_perm.clear(); L.bwd_perm.clear();
  L.fwd.reserve(n_dp + 2 * n_reach + 4 * (D + 1));
  L.bwd.reserve(n_dp + 2 * n_reach + 4 * (D + 1));
  L.fwd_perm.reserve(n_dp); L.bwd_perm.reserve(n_dp);
  for (int t = 1; t <= D; ++t) {
    auto &sv = by_height[t];
    std::stable_sort(sv.begin(), sv.end(), [&](int a, int b) { return outdeg[a] > outdeg[b]; });
    emit_level(sv, o, [&](int s) { return std::make_pair(out_ptr[s], out_ptr[s + 1]); },
               [&](int a) { return L.dst[a]; }, out_list, L.label, L.bwd, L.bwd_perm, L.bwd_steps);
    auto &dv = by_depth[t];
    std::stable_sort(dv.begin(), dv.end(), [&](int a, int b) { return indeg[a] > indeg[b]; });
    emit_level(dv, o, [&](int s) { return std::make_pair(in_ptr[s], in_ptr[s + 1]); },
               [&](int a) { return L.src[a]; }, in_list, L.label, L.fwd, L.fwd_perm, L.fwd_steps);
  }
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
    if (o->max_step_words > 0) r.max_step_words = std::min<int>(o->max_step_words, NFST_MAX_STEP_WORDS);
    r.lanes_policy = o->lanes_policy;
    if (o->sweep_waves == 1 || o->sweep_waves == 2 || o->sweep_waves == 4) r.sweep_waves = o->sweep_waves;
  }
  return r;
}

}  // namespace

struct nfst_packed {
  nfst_batch view{};
  std::vector<int32_t> meta, row_ptr, arc_src, arc_dst, arc_label, fwd_perm, bwd_perm;
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
  int64_t rows = 0, arcs = 0, dp = 0, fw = 0, bw = 0;
  int max_rows = 0, max_steps = 0;
  p->meta.assign((size_t)B * NFST_META_WORDS, 0);
  for (int b = 0; b < B; ++b) {
    Lat &L = lats[b];
    int32_t *m = &p->meta[(size_t)b * NFST_META_WORDS];
    m[NFST_META_ROW_OFF] = (int32_t)rows; m[NFST_META_N_ROWS] = L.n_rows;
    m[NFST_META_ARC_OFF] = (int32_t)arcs; m[NFST_META_N_ARCS] = (int32_t)L.src.size();
    m[NFST_META_FWD_OFF] = (int32_t)fw; m[NFST_META_FWD_STEPS] = L.fwd_steps;
    m[NFST_META_BWD_OFF] = (int32_t)bw; m[NFST_META_BWD_STEPS] = L.bwd_steps;
    m[NFST_META_SINK] = L.sink; m[NFST_META_N_REACH] = L.n_reach; m[NFST_META_DEPTH] = L.depth;
    m[NFST_META_DP_OFF] = (int32_t)dp; m[NFST_META_N_DP] = L.n_dp;
    m[NFST_META_FWD_WORDS] = (int32_t)L.fwd.size(); m[NFST_META_BWD_WORDS] = (int32_t)L.bwd.size();
    rows += L.n_rows; arcs += (int64_t)L.src.size(); dp += L.n_dp;
    // every lattice's stream starts on a 256-byte boundary (LDS-DMA chunks are 16 B per lane)
    fw += ((int64_t)L.fwd.size() + 63) / 64 * 64; bw += ((int64_t)L.bwd.size() + 63) / 64 * 64;
    max_rows = std::max(max_rows, L.n_rows);
    max_steps = std::max(max_steps, std::max(L.fwd_steps, L.bwd_steps));
    if (arcs > 0x7fffff00ll || fw > 0x7fffff00ll || bw > 0x7fffff00ll || rows > 0x7fffff00ll) {
      delete p; if (err_lattice) *err_lattice = b; return NFST_ERR_LIMIT;
    }
  }
  // one LDS-DMA chunk (256 words) + a header of slack at the end of each stream: the
  // last chunk of the last lattice is read whole
  p->row_ptr.resize(rows + B); p->arc_src.resize(arcs); p->arc_dst.resize(arcs); p->arc_label.resize(arcs);
  if (weighted) p->arc_w.resize(arcs);
  p->fwd.assign(fw + 512, 0); p->bwd.assign(bw + 512, 0); p->fwd_perm.resize(dp); p->bwd_perm.resize(dp);
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
    }
    if (!L.fwd.empty()) std::memcpy(&p->fwd[m[NFST_META_FWD_OFF]], L.fwd.data(), L.fwd.size() * 4);
    if (!L.bwd.empty()) std::memcpy(&p->bwd[m[NFST_META_BWD_OFF]], L.bwd.data(), L.bwd.size() * 4);
    int32_t d0 = m[NFST_META_DP_OFF];
    for (int i = 0; i < L.n_dp; ++i) { p->fwd_perm[d0 + i] = a0 + L.fwd_perm[i]; p->bwd_perm[d0 + i] = a0 + L.bwd_perm[i]; }
    std::vector<int32_t>().swap(L.src); std::vector<uint32_t>().swap(L.fwd); std::vector<uint32_t>().swap(L.bwd);
  });
  nfst_batch &v = p->view;
  v.n_lattices = B; v.vocab = vocab; v.max_rows = max_rows; v.max_steps = max_steps;
  v.weighted = weighted ? 1 : 0; v.total_rows = rows; v.total_arcs = arcs; v.total_dp_arcs = dp;
  v.fwd_words = fw + 512; v.bwd_words = bw + 512;
  v.max_step_words = o.max_step_words;
  v.sweep_waves = o.sweep_waves;
  v.meta = p->meta.data(); v.row_ptr = p->row_ptr.data(); v.arc_src = p->arc_src.data();
  v.arc_dst = p->arc_dst.data(); v.arc_label = p->arc_label.data();
  v.arc_w = weighted ? p->arc_w.data() : nullptr;
  v.fwd_stream = p->fwd.data(); v.bwd_stream = p->bwd.data();
  v.fwd_perm = p->fwd_perm.data(); v.bwd_perm = p->bwd_perm.data();
  *out = p;
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

const char *nfst_strerror(int code) {
  switch (code) {
    case NFST_OK: return "ok";
    case NFST_ERR_ARG: return "invalid argument (null pointer, bad size or unsorted arcs)";
    case NFST_ERR_INDEX: return "state or label index out of range";
    case NFST_ERR_CYCLE: return "lattice is not acyclic";
    case NFST_ERR_SINK: return "lattice must have exactly one final (sink) state";
    case NFST_ERR_DETERMINISM: return "two arcs leave one state with the same label";
    case NFST_ERR_LIMIT: return "lattice exceeds engine limits (rows <= 8192, vocab <= 32768)";
    case NFST_ERR_HIP: return "HIP runtime error";
    case NFST_ERR_NOMEM: return "out of memory";
    case NFST_ERR_LENGTH: return "ran out of length budget";
    default: return "unknown error";
  }
}

}  // extern "C"
