// pack_kernels.h -- the packer on the device (K1 of SURVEY.md section 2, section 7 step 4: "pack on GPU")
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
//
// The reference hands set_masks tables that Lightning has already moved to the GPU
// (/root/reference/src/modules/lightning.py:417, 354-355 -> scorers.py:877-885).  The host packer (pack.cpp) made
// them travel GPU -> host -> packer -> GPU.  Here the same schedule is computed where the tables are:
//   k_dense_reach / k_dense_write   dense [B, R, V] tables -> arc lists of the rows reachable from state 0
//                                   (12 bytes per arc; collate padding rows never become arcs)
//   k_pack_lattice<EMIT>            arc lists -> canonical arcs + tile programs, ONE workgroup per lattice:
//       reachability, depth and height by iterated relaxation in LDS; canonical arcs, out- and in-arc lists by
//       block-wide scans (in-arc lists: counting placement + a rank inside every destination's segment, which is
//       the stable order); level orders and tile layouts by bitonic sorts of 64-bit keys in LDS + block-wide scans.
// The output is BIT-IDENTICAL to pack.cpp's for the formats the device packer supports: compact tiles (vocab + 2
// <= 2048, four slots per lane), narrow / wide groups chosen by the same cost model.  Two launches: a planning pass
// that counts (tiles, arcs, scratch rows per lattice -> the host sizes the arrays), and an emitting pass.
#pragma once

constexpr int kPkThreads = 1024;
constexpr int kPkRowArrays = 16;   // int32 workspace arrays of n_rows + 2 entries per lattice
constexpr int kPkArcArrays = 4;    // ... of one entry per input arc
constexpr int kPkMaxKeys = 16384;  // 64-bit sort keys that fit LDS (pieces of one direction; states of a lattice)
constexpr int kPkAuxArrays = 4;    // per-piece int32 arrays (scans of the tile layout, tile bits)
constexpr int64_t kPkLdsBytes = 152 * 1024;
// words of per-piece workspace in front of lattice b / in total: a direction has at most 2 n + A / 15 pieces
__host__ __device__ inline int64_t pk_aux_off(int64_t row_off, int64_t arc_off, int64_t b) { return 2 * row_off + (arc_off >> 3) + 64 * b; }
__host__ __device__ inline int64_t pk_ws_words(int64_t n_lattices, int64_t total_rows, int64_t total_arcs) {
  return kPkRowArrays * (total_rows + 2 * n_lattices) + kPkArcArrays * total_arcs +
         kPkAuxArrays * (pk_aux_off(total_rows, total_arcs, n_lattices) + 64);
}

struct PkArgs {
  const int32_t *n_rows;   // [B]
  const int64_t *row_off;  // [B + 1] prefix sums of n_rows (workspace layout)
  const int64_t *arc_off;  // [B + 1]
  const int32_t *src, *label, *dst;
  const float *w;          // or null
  int32_t vocab, group_mode;
  int32_t *ws;             // workspace: kPkRowArrays * (total rows + 2 B) + kPkArcArrays * total arcs int32
  int64_t total_rows, total_arcs;
  int32_t n_lattices;
  int32_t *meta;           // [B, 16]: plan writes the counts, emit reads counts + offsets
  int32_t *status;         // [B] error code per lattice (0 = ok)
  int32_t *scratch;        // [B] scratch rows of the chosen programs
  nfst_batch out;          // emit: destination arrays (device)
};

// ---- block-wide primitives (1024 threads, `red` = 64 words of LDS scratch) -----------------------------------------
__device__ __forceinline__ int pk_wave_incl_scan(int v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if ((int)(threadIdx.x & 63) >= d) v += o;
  }
  return v;
}
__device__ __forceinline__ int pk_wave_incl_max(int v) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(v, d);
    if ((int)(threadIdx.x & 63) >= d) v = max(v, o);
  }
  return v;
}
// exclusive scan of one value per thread; returns this thread's prefix, *total = the sum over the block
template <bool MAX>
__device__ int pk_block_excl(int v, int *red, int *total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int incl = MAX ? pk_wave_incl_max(v) : pk_wave_incl_scan(v);
  __syncthreads();  // (red may still be read by the previous call)
  if (lane == 63) red[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    const int x = lane < kPkThreads / 64 ? red[lane] : (MAX ? INT_MIN : 0);
    const int s = MAX ? pk_wave_incl_max(x) : pk_wave_incl_scan(x);
    if (lane < kPkThreads / 64) red[16 + lane] = s;
  }
  __syncthreads();
  const int base = wave > 0 ? red[16 + wave - 1] : (MAX ? INT_MIN : 0);
  *total = red[16 + kPkThreads / 64 - 1];
  const int prev = __shfl_up(incl, 1);
  const int excl_in_wave = lane > 0 ? prev : (MAX ? INT_MIN : 0);
  return MAX ? max(base, excl_in_wave) : base + excl_in_wave;
}
// exclusive scan of f(i), i in [0, n), into out[i] (thread t owns a contiguous chunk); returns the total
template <bool MAX, class F>
__device__ int pk_scan(int n, F f, int32_t *out, int *red) {
  const int chunk = (n + kPkThreads - 1) / kPkThreads;
  const int i0 = min((int)threadIdx.x * chunk, n), i1 = min(i0 + chunk, n);
  int acc = MAX ? INT_MIN : 0;
  for (int i = i0; i < i1; ++i) { const int x = f(i); acc = MAX ? max(acc, x) : acc + x; }
  int total;
  int run = pk_block_excl<MAX>(acc, red, &total);
  for (int i = i0; i < i1; ++i) {
    const int x = f(i);
    out[i] = run;
    run = MAX ? max(run, x) : run + x;
  }
  __syncthreads();
  return total;
}
template <class F>
__device__ int pk_count(int n, F f, int *red) {  // sum of f(i) over [0, n)
  int acc = 0;
  for (int i = threadIdx.x; i < n; i += kPkThreads) acc += f(i);
  int total;
  pk_block_excl<false>(acc, red, &total);
  __syncthreads();
  return total;
}
// bitonic sort of n 64-bit keys in LDS (ascending; the array holds np2 >= n entries, padded here with ~0)
__device__ void pk_sort(unsigned long long *key, int n) {
  int np2 = 1;
  while (np2 < n) np2 <<= 1;
  for (int i = n + threadIdx.x; i < np2; i += kPkThreads) key[i] = ~0ull;
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += kPkThreads) {
        const int x = i ^ j;
        if (x > i) {
          const unsigned long long a = key[i], b = key[x];
          if ((a > b) == ((i & k) == 0)) { key[i] = b; key[x] = a; }
        }
      }
      __syncthreads();
    }
}

__device__ __forceinline__ int pk_ceil_log2(int x) { return x <= 1 ? 0 : 32 - __builtin_clz(x - 1); }

// ---- pieces of a state (pack.cpp: emit_level), as closed formulas of (state, pass, partial index) -------------------
struct PkPiece {
  int row, begin, end, accum, units, g;
};
struct PkState {  // what the formulas need about one state of the level being laid out
  int s, b0, e0, tree, n_part, first;  // arcs [b0, e0) of its list; tree: partial groups write scratch rows first ..
};
__device__ __forceinline__ int pk_chain_pieces(int len, int cap) { return len <= cap ? 1 : 1 + (len - cap + cap - 2) / (cap - 1); }
__device__ __forceinline__ int pk_state_pieces(const PkState &st, int cap) {
  return st.tree ? st.n_part + pk_chain_pieces(st.n_part, cap) : pk_chain_pieces(st.e0 - st.b0, cap);
}
// the c-th piece of a chain over [b, e): the first takes `cap` slots, every continuation piece one less (its carry)
__device__ __forceinline__ void pk_chain_piece(int b, int e, int c, int cap, int &pb, int &pe) {
  pb = c == 0 ? b : b + cap + (c - 1) * (cap - 1);
  pe = min(e, pb + (c == 0 ? cap : cap - 1));
}
__device__ __forceinline__ PkPiece pk_piece(const PkState &st, int pass, int sub, int cap) {
  PkPiece p;
  if (st.tree && pass == 0) {
    p.row = st.first + sub; p.begin = st.b0 + sub * cap; p.end = min(st.e0, p.begin + cap); p.accum = 0; p.units = 0;
  } else if (st.tree) {
    pk_chain_piece(st.first, st.first + st.n_part, pass - 1, cap, p.begin, p.end);
    p.row = st.s; p.accum = pass > 1; p.units = 1;
  } else {
    pk_chain_piece(st.b0, st.e0, pass, cap, p.begin, p.end);
    p.row = st.s; p.accum = pass > 0; p.units = 0;
  }
  p.g = pk_ceil_log2(max(1, (p.end - p.begin + p.accum + 3) / 4));
  return p;
}
// sort key of a piece: tiles are laid out by level, then pass, then group size (largest first), then the order in
// which pack.cpp's emit_level met the pieces (position of the state in its level, partial index)
__device__ __forceinline__ unsigned long long pk_key(int level, int pass, int g, int pos, int sub) {
  return ((unsigned long long)level << 50) | ((unsigned long long)pass << 36) | ((unsigned long long)(7 - g) << 33) |
         ((unsigned long long)pos << 13) | (unsigned long long)sub;  // level 13 | pass 14 | 3 | pos 20 | sub 13 bits
}

// ---- one direction, one group mode: tile layout (count or emit) -------------------------------------------------------
struct PkDir {
  const int32_t *ord, *lvlp, *lev, *ptr, *list;  // level order, level starts, level of a state, list pointers, arc list (canonical ids)
  const int32_t *other;                         // other end of an arc, by canonical id (emit)
  const int32_t *lab;                           // label of an arc, by canonical id (emit)
  int n_reach, D, n_rows;
};
struct PkLayout {
  int tiles, wide, scratch, err;
};
// tmp_a, tmp_b, tmp_c: int32 workspace of n_reach + 2 entries each (global); keys: LDS, kPkMaxKeys entries;
// aux: LDS int32 [3 * kPkMaxKeys]?  -- no: the per-piece scans run through global arrays pa / pb / pc of kPkMaxKeys entries
template <bool EMIT>
__device__ PkLayout pk_layout(const PkDir &d, int max_g, int vocab, unsigned long long *keys, int32_t *tmp_a, int32_t *tmp_b,
                              int32_t *pa, int32_t *pb, int32_t *pc, int32_t *tile_bits, int aux_cap, int *red, uint32_t *stream,
                              int32_t *perm, int32_t perm_base) {
  PkLayout out{0, 0, 0, 0};
  const int cap = (1 << max_g) * 4;
  const int first_pos = d.lvlp[1], n_st = d.n_reach - first_pos;  // states of the levels 1 .. D (level 0 holds one state and emits nothing)
  if (n_st <= 0) return out;
  auto state_at = [&](int pos) {  // pos: position in the level order
    PkState st;
    st.s = d.ord[pos];
    st.b0 = d.ptr[st.s]; st.e0 = d.ptr[st.s + 1];
    const int deg = st.e0 - st.b0;
    st.tree = max_g == 3 && deg > 2 * cap;
    st.n_part = st.tree ? (deg + cap - 1) / cap : 0;
    st.first = 0;
    return st;
  };
  // scratch rows: tmp_a[i] = partial groups of the states before i (position first_pos + i) in the level order
  pk_scan<false>(n_st, [&](int i) { return state_at(first_pos + i).n_part; }, tmp_a, red);
  // pieces per state -> offsets of their keys
  const int n_pieces = pk_scan<false>(n_st, [&](int i) { return pk_state_pieces(state_at(first_pos + i), cap); }, tmp_b, red);
  if (n_pieces > kPkMaxKeys || n_pieces + 1 > aux_cap) { out.err = NFST_ERR_LIMIT; return out; }
  auto full_state = [&](int pos, int level) {
    PkState st = state_at(pos);
    const int i = pos - first_pos;
    st.first = d.n_rows + tmp_a[i] - tmp_a[d.lvlp[level] - first_pos];
    return st;
  };
  // scratch rows of the fullest level
  {
    int worst = 0;
    for (int t = 1 + threadIdx.x; t <= d.D; t += kPkThreads) {
      const int a = d.lvlp[t] - first_pos, b = d.lvlp[t + 1] - first_pos;
      const int end_sum = b < n_st ? tmp_a[b] : tmp_a[n_st - 1] + state_at(first_pos + n_st - 1).n_part;
      worst = max(worst, end_sum - tmp_a[a]);
    }
    int total;
    pk_block_excl<true>(worst, red, &total);
    __syncthreads();
    out.scratch = max(total, 0);
    if (d.n_rows + out.scratch > NFST_MAX_ROWS) { out.err = NFST_ERR_LIMIT; return out; }  // (pack.cpp then chains instead: host packer)
  }
  // keys
  for (int i = threadIdx.x; i < n_st; i += kPkThreads) {
    const int pos = first_pos + i, level = d.lev[d.ord[pos]];
    const PkState st = full_state(pos, level);
    int k = tmp_b[i];
    const int lp = pos - d.lvlp[level];
    if (st.tree) {
      for (int j = 0; j < st.n_part; ++j) keys[k++] = pk_key(level, 0, pk_piece(st, 0, j, cap).g, lp, j);
      const int nc = pk_chain_pieces(st.n_part, cap);
      for (int c = 0; c < nc; ++c) keys[k++] = pk_key(level, 1 + c, pk_piece(st, 1 + c, 0, cap).g, lp, 0);
    } else {
      const int nc = pk_chain_pieces(st.e0 - st.b0, cap);
      for (int c = 0; c < nc; ++c) keys[k++] = pk_key(level, c, pk_piece(st, c, 0, cap).g, lp, 0);
    }
  }
  __syncthreads();
  pk_sort(keys, n_pieces);
  // lane offsets: sizes are powers of two in falling order inside a (level, pass) segment, so a piece never straddles
  // a tile and "next fit" is the running sum.  pa = sum of sizes before i, pb = head index of i's segment,
  // pc = tiles of the segments that end before i
  auto size_of = [&](int i) { return 1 << (7 - (int)((keys[i] >> 33) & 7ull)); };
  auto seg_of = [&](int i) { return keys[i] >> 36; };
  pk_scan<false>(n_pieces, size_of, pa, red);
  pk_scan<true>(n_pieces + 1, [&](int i) { return (i < n_pieces && (i == 0 || seg_of(i) != seg_of(i - 1))) ? i : -1; }, pb, red);
  // pb[i + 1] = head of i's segment (an exclusive max scan shifted by one)
  auto head_of = [&](int i) { return pb[i + 1]; };
  auto seg_tiles_at_end = [&](int i) {  // for the last piece of a segment: the segment's tiles
    if (i + 1 < n_pieces && seg_of(i + 1) == seg_of(i)) return 0;
    const int h = head_of(i);
    return (pa[i] + size_of(i) - pa[h] + 63) >> 6;
  };
  out.tiles = pk_scan<false>(n_pieces, seg_tiles_at_end, pc, red);
  // tiles whose first piece is a wide group
  out.wide = pk_count(n_pieces, [&](int i) {
    const int h = head_of(i);
    return (((pa[i] - pa[h]) & 63) == 0 && (7 - (int)((keys[i] >> 33) & 7ull)) > 3) ? 1 : 0;
  }, red);
  if (!EMIT) return out;
  // ---- emission.  (a) every tile gets its default words: control 0, null records, no arcs
  const uint32_t null24 = (uint32_t)vocab << 13;
  const uint32_t n1 = null24 | (null24 << 24), n2 = (null24 >> 8) | (null24 << 16), n3 = (null24 >> 16) | (null24 << 8);
  for (int i = threadIdx.x; i < out.tiles * 64; i += kPkThreads) {
    reinterpret_cast<uint4 *>(stream)[i] = make_uint4(0u, n1, n2, n3);
    reinterpret_cast<int4 *>(perm)[i] = make_int4(-1, -1, -1, -1);
  }
  for (int i = threadIdx.x; i < out.tiles; i += kPkThreads) tile_bits[i] = 0;
  __syncthreads();
  // (b) the pieces
  const uint32_t unit_label = (uint32_t)vocab + 1;
  for (int i = threadIdx.x; i < n_pieces; i += kPkThreads) {
    const unsigned long long key = keys[i];
    const int level = (int)(key >> 50), pass = (int)((key >> 36) & 0x3fffull), lp = (int)((key >> 13) & 0xfffffull), sub = (int)(key & 0x1fffull);
    const PkState st = full_state(d.lvlp[level] + lp, level);
    const PkPiece p = pk_piece(st, pass, sub, cap);
    const int h = head_of(i), off = pa[i] - pa[h], tile = pc[h] + (off >> 6), lane0 = off & 63;
    if (lane0 == 0) atomicOr(&tile_bits[tile], p.g << 23);  // (falling sizes: the tile's first piece has its largest group)
    if (p.accum) atomicOr(&tile_bits[tile], 1 << 26);
    const int n_slots = p.end - p.begin + p.accum;
    int slot = 0;
    for (int r = 0; r < (1 << p.g); ++r) {
      uint32_t r24[4];
      int pm[4];
#pragma unroll
      for (int j = 0; j < 4; ++j, ++slot) {
        r24[j] = null24; pm[j] = -1;
        if (slot >= n_slots) continue;
        if (p.accum && slot == 0) {
          r24[j] = (uint32_t)p.row | (unit_label << 13);
        } else if (p.units) {
          r24[j] = (uint32_t)(p.begin + slot - p.accum) | (unit_label << 13);
        } else {
          const int arc = d.list[p.begin + slot - p.accum];
          r24[j] = (uint32_t)d.other[arc] | ((uint32_t)d.lab[arc] << 13);
          pm[j] = perm_base + arc;
        }
      }
      uint32_t c = ((uint32_t)p.row << 3) | ((uint32_t)p.g << 20);
      if (r == 0) c |= (1u << 31) | (p.accum ? (1u << 30) : 0u);
      const size_t at = (size_t)tile * 64 + lane0 + r;
      reinterpret_cast<uint4 *>(stream)[at] = make_uint4(c, r24[0] | (r24[1] << 24), (r24[1] >> 8) | (r24[2] << 16), (r24[2] >> 16) | (r24[3] << 8));
      reinterpret_cast<int4 *>(perm)[at] = make_int4(pm[0], pm[1], pm[2], pm[3]);
    }
  }
  __threadfence_block();
  __syncthreads();
  // (c) the tile-wide bits of every control word: largest group, "holds a continuation piece"
  for (int i = threadIdx.x; i < out.tiles * 64; i += kPkThreads) stream[(size_t)i * 4] |= (uint32_t)tile_bits[i >> 6];
  __syncthreads();
  return out;
}

// The planning pass only needs COUNTS: a (level, pass) segment of pieces takes ceil(sum of sizes / 64) tiles whatever
// the order inside it, its wide tiles are ceil(sum of the sizes of groups beyond 8 lanes / 64) (those come first), and
// the scratch rows of a level are the partial groups of its states.  One pass over the states with atomic adds into
// tables indexed by (level, pass) in LDS -- no sort, no scan.  `table`: LDS, `words` int32 of room.  Returns err =
// NFST_ERR_ARG when the tables do not fit (the caller then counts with pk_layout<false>).
__device__ PkLayout pk_count_layout(const PkDir &d, int max_g, int32_t *table, int words, int aux_cap, int *red) {
  PkLayout out{0, 0, 0, 0};
  const int cap = (1 << max_g) * 4;
  const int first_pos = d.lvlp[1], n_st = d.n_reach - first_pos;
  if (n_st <= 0) return out;
  auto state_at = [&](int pos) {
    PkState st;
    st.s = d.ord[pos];
    st.b0 = d.ptr[st.s]; st.e0 = d.ptr[st.s + 1];
    const int deg = st.e0 - st.b0;
    st.tree = max_g == 3 && deg > 2 * cap;
    st.n_part = st.tree ? (deg + cap - 1) / cap : 0;
    st.first = 0;
    return st;
  };
  // (the emitting pass sorts the pieces in LDS: what it cannot take is refused here, where the host still can fall back)
  const int n_pieces = pk_count(n_st, [&](int i) { return pk_state_pieces(state_at(first_pos + i), cap); }, red);
  if (n_pieces > kPkMaxKeys || n_pieces + 1 > aux_cap) { out.err = NFST_ERR_LIMIT; return out; }
  // passes of the longest chain
  int pmax_t = 1, P = 1;
  for (int i = threadIdx.x; i < n_st; i += kPkThreads) {
    const PkState st = state_at(first_pos + i);
    pmax_t = max(pmax_t, st.tree ? 1 + pk_chain_pieces(st.n_part, cap) : pk_chain_pieces(st.e0 - st.b0, cap));
  }
  pk_block_excl<true>(pmax_t, red, &P);
  __syncthreads();
  const int L = d.D + 1;
  if ((int64_t)L * (2 * P + 1) > words) { out.err = NFST_ERR_ARG; return out; }
  int32_t *t_size = table, *t_wide = table + L * P, *t_scr = table + 2 * L * P;
  for (int i = threadIdx.x; i < L * (2 * P + 1); i += kPkThreads) table[i] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < n_st; i += kPkThreads) {
    const PkState st = state_at(first_pos + i);
    const int level = d.lev[st.s];
    auto add = [&](int pass, const PkPiece &p) {
      atomicAdd(&t_size[level * P + pass], 1 << p.g);
      if (p.g > 3) atomicAdd(&t_wide[level * P + pass], 1 << p.g);
    };
    if (st.tree) {
      atomicAdd(&t_scr[level], st.n_part);
      for (int j = 0; j < st.n_part; ++j) add(0, pk_piece(st, 0, j, cap));
      const int nc = pk_chain_pieces(st.n_part, cap);
      for (int c = 0; c < nc; ++c) add(1 + c, pk_piece(st, 1 + c, 0, cap));
    } else {
      const int nc = pk_chain_pieces(st.e0 - st.b0, cap);
      for (int c = 0; c < nc; ++c) add(c, pk_piece(st, c, 0, cap));
    }
  }
  __syncthreads();
  out.tiles = pk_count(L * P, [&](int i) { return (t_size[i] + 63) >> 6; }, red);
  out.wide = pk_count(L * P, [&](int i) { return (t_wide[i] + 63) >> 6; }, red);
  int scr_t = 0;
  for (int i = threadIdx.x; i < L; i += kPkThreads) scr_t = max(scr_t, t_scr[i]);
  pk_block_excl<true>(scr_t, red, &out.scratch);
  __syncthreads();
  if (d.n_rows + out.scratch > NFST_MAX_ROWS) out.err = NFST_ERR_LIMIT;  // (pack.cpp then chains instead: host packer)
  return out;
}

#ifdef NFST_PK_STAMPS
__device__ unsigned long long pk_stamps[32];
#define PK_STAMP(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) pk_stamps[(EMIT ? 16 : 0) + (k)] = wall_clock64(); } while (0)
#else
#define PK_STAMP(k) do { } while (0)
#endif
// ---- the packer: one workgroup per lattice -----------------------------------------------------------------------------
template <bool EMIT>
__global__ __launch_bounds__(kPkThreads) void k_pack_lattice(PkArgs a) {
  extern __shared__ unsigned long long pk_lds[];
  __shared__ int red[64];
  __shared__ int sh[8];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = a.n_rows[b];
  const int64_t a0 = a.arc_off[b];
  const int A = (int)(a.arc_off[b + 1] - a0);
  int32_t *meta = a.meta + (size_t)b * NFST_META_WORDS;
  if (EMIT && a.status[b] != 0) return;  // (the host refuses a batch with a failed lattice before it launches this)
  auto fail = [&](int code) {
    if (tid == 0) { a.status[b] = code; }
  };
  if (n <= 0 || n > NFST_MAX_ROWS || A < 0) { fail(n <= 0 || A < 0 ? NFST_ERR_ARG : NFST_ERR_LIMIT); return; }
  const int32_t *src = a.src + a0, *lab = a.label + a0, *dst = a.dst + a0;
  // workspace of this lattice
  int32_t *wrow = a.ws + (size_t)kPkRowArrays * (a.row_off[b] + 2 * b);
  int32_t *warc = a.ws + (size_t)kPkRowArrays * (a.total_rows + 2 * a.n_lattices) + (size_t)kPkArcArrays * a0;
  const int rs = n + 2;
  int32_t *g_dep = wrow, *g_hei = wrow + rs, *g_in = wrow + 2 * rs, *g_out = wrow + 3 * rs, *in_ptr = wrow + 4 * rs, *out_ptr = wrow + 5 * rs,
          *ord_f = wrow + 6 * rs, *ord_b = wrow + 7 * rs, *lvlp_f = wrow + 8 * rs, *lvlp_b = wrow + 9 * rs, *tmp_a = wrow + 10 * rs,
          *tmp_b = wrow + 11 * rs, *cnt = wrow + 12 * rs;
  int32_t *canon = warc, *in_tmp = warc + A, *in_list = warc + 2 * (size_t)A, *out_list = warc + 3 * (size_t)A;
  // per-piece arrays of the tile layout (scans, tile bits)
  const int64_t aux_total = pk_aux_off(a.total_rows, a.total_arcs, a.n_lattices) + 64;
  const int64_t aux_at = pk_aux_off(a.row_off[b], a0, b);
  const int aux_cap = (int)(pk_aux_off(a.row_off[b + 1], a.arc_off[b + 1], b + 1) - aux_at);
  int32_t *aux = a.ws + (size_t)kPkRowArrays * (a.total_rows + 2 * a.n_lattices) + (size_t)kPkArcArrays * a.total_arcs + aux_at;
  int32_t *pa_ = aux, *pb_ = aux + aux_total, *pc_ = aux + 2 * aux_total, *tile_bits = aux + 3 * aux_total;

  // The emitting pass reads what the planning pass left in the workspace (depths, canonical ids, arc lists, level orders):
  // the same workspace, untouched in between.
  PK_STAMP(0);
  int n_reach = EMIT ? meta[NFST_META_N_REACH] : 0, D = EMIT ? meta[NFST_META_DEPTH] : 0, sink = 0, n_arcs = 0, n_dp = 0;
  if (!EMIT) {
  // ---- 0. the arcs are what the host packer accepts: ids in range, sorted by (src, label), one arc per (state, label)
  if (tid == 0) { sh[0] = 0x7fffffff; sh[1] = 0; }
  __syncthreads();
  for (int i = tid; i < A; i += kPkThreads) {
    int code = 0;
    if (src[i] < 0 || src[i] >= n || dst[i] < 0 || dst[i] >= n || lab[i] < 0 || lab[i] >= a.vocab) code = 1;  // NFST_ERR_INDEX
    else if (i > 0 && src[i] < src[i - 1]) code = 2;                                                            // NFST_ERR_ARG
    else if (i > 0 && src[i] == src[i - 1] && lab[i] <= lab[i - 1]) code = 3;                                  // NFST_ERR_DETERMINISM
    if (code) atomicMin(&sh[0], i * 4 + code);  // the first offending arc decides, as in pack.cpp
  }
  __syncthreads();
  if (sh[0] != 0x7fffffff) {
    const int code = sh[0] & 3;
    fail(code == 1 ? NFST_ERR_INDEX : code == 2 ? NFST_ERR_ARG : NFST_ERR_DETERMINISM);
    return;
  }
  PK_STAMP(1);
  // ---- 1. reachability, depth (longest path from 0) and height (longest path to the sink) by relaxation in LDS
  int *dep = reinterpret_cast<int *>(pk_lds), *hei = dep + n;
  uint32_t *sd = reinterpret_cast<uint32_t *>(hei + n);  // src | dst << 16 of every arc when they fit beside
  const bool sd_in_lds = (int64_t)A * 4 + (int64_t)n * 8 <= kPkLdsBytes;
  for (int s = tid; s < n; s += kPkThreads) { dep[s] = s == 0 ? 0 : -1; hei[s] = 0; }
  if (sd_in_lds) for (int i = tid; i < A; i += kPkThreads) sd[i] = (uint32_t)src[i] | ((uint32_t)dst[i] << 16);
  __syncthreads();
  // One sweep over the arcs moves depths forward and heights back by a level at least.  First with the heights of ALL
  // states (D + 2 sweeps on a lattice whose unreachable part is acyclic too: the heights of reachable states only depend
  // on reachable states); if that does not settle -- junk arcs among unreachable states may form cycles -- again with
  // the heights of reached states only (they start moving when their states are reached: 2 D + 2 sweeps at most).
  bool cyclic = false;
  for (int attempt = 0; attempt < 2; ++attempt) {
    const int cap_it = attempt == 0 ? n + 2 : 2 * n + 4;
    cyclic = false;
    if (attempt == 1) {
      for (int s = tid; s < n; s += kPkThreads) { dep[s] = s == 0 ? 0 : -1; hei[s] = 0; }
      __syncthreads();
    }
    if (tid == 0) { sh[4] = 0; sh[5] = 0; sh[6] = 0; }
    __syncthreads();
    for (int it = 0;; ++it) {  // (three "something moved" flags take turns: one barrier per sweep)
      int changed = 0;
      for (int i = tid; i < A; i += kPkThreads) {
        int s, d2;
        if (sd_in_lds) { const uint32_t x = sd[i]; s = (int)(x & 0xffffu); d2 = (int)(x >> 16); }
        else { s = src[i]; d2 = dst[i]; }
        if (s == d2) continue;
        const int ds = dep[s];
        if (ds >= 0 && dep[d2] < ds + 1) { atomicMax(&dep[d2], ds + 1); changed = 1; }
        if (attempt == 1 && ds < 0) continue;
        const int hd = hei[d2];
        if (hei[s] < hd + 1) { atomicMax(&hei[s], hd + 1); changed = 1; }
      }
      if (changed) sh[4 + it % 3] = 1;
      if (tid == 0) sh[4 + (it + 2) % 3] = 0;  // the flag of the sweep after the next: nobody reads or sets it now
      __syncthreads();
      const int any = sh[4 + it % 3];
      if (!any) break;
      if (it > cap_it) { cyclic = true; break; }
    }
    if (!cyclic) break;
  }
  // counters and the unordered in-arc lists live in LDS when they fit (global atomics on a few hot counters -- a state
  // with hundreds of in-arcs -- cost more than everything else of this pass together)
  const bool c_lds = (int64_t)5 * n * 4 <= kPkLdsBytes;
  int *c_out = c_lds ? hei + n : g_out, *c_in = c_lds ? hei + 2 * n : g_in, *c_cnt = c_lds ? hei + 3 * n : cnt;
  for (int s = tid; s < n; s += kPkThreads) { g_dep[s] = dep[s]; g_hei[s] = hei[s]; c_in[s] = 0; c_out[s] = 0; c_cnt[s] = 0; }
  __syncthreads();
  PK_STAMP(2);
  // ---- 2. degrees over the arcs of reachable states (self loops are not part of the sweeps), sink
  for (int i = tid; i < A; i += kPkThreads) {
    const int s = src[i], d2 = dst[i];
    if (dep[s] >= 0 && s != d2) { atomicAdd(&c_out[s], 1); atomicAdd(&c_in[d2], 1); }
  }
  __syncthreads();
  if (c_lds) for (int s = tid; s < n; s += kPkThreads) { g_out[s] = c_out[s]; g_in[s] = c_in[s]; }
  n_reach = pk_count(n, [&](int s) { return dep[s] >= 0 ? 1 : 0; }, red);
  const int sinks = pk_count(n, [&](int s) { return (dep[s] >= 0 && c_out[s] == 0) ? 1 : 0; }, red);
  if (sinks != 1) { fail(NFST_ERR_SINK); return; }
  if (cyclic) { fail(NFST_ERR_CYCLE); return; }
  if (tid == 0) sh[2] = 0;
  __syncthreads();
  for (int s = tid; s < n; s += kPkThreads)
    if (dep[s] >= 0 && c_out[s] == 0) sh[2] = s;
  __syncthreads();
  sink = sh[2];
  D = dep[sink];
  PK_STAMP(3);
  // ---- 3. canonical arcs (arcs of reachable states, input order), out- and in-arc lists of the sweeps
  n_arcs = pk_scan<false>(A, [&](int i) { return dep[src[i]] >= 0 ? 1 : 0; }, canon, red);
  n_dp = pk_scan<false>(A, [&](int i) { return (dep[src[i]] >= 0 && src[i] != dst[i]) ? 1 : 0; }, in_tmp, red);
  for (int i = tid; i < A; i += kPkThreads)
    if (dep[src[i]] >= 0 && src[i] != dst[i]) out_list[in_tmp[i]] = canon[i];
  pk_scan<false>(n + 1, [&](int s) { return s < n ? c_out[s] : 0; }, out_ptr, red);
  pk_scan<false>(n + 1, [&](int s) { return s < n ? c_in[s] : 0; }, in_ptr, red);
  PK_STAMP(4);
  // in-arcs: placed in any order inside their destination's segment, then ranked by input position = the stable order
  const bool t_lds = c_lds && ((int64_t)5 * n + n_dp) * 4 <= kPkLdsBytes;
  int *place = t_lds ? hei + 4 * n : in_tmp;
  for (int i = tid; i < A; i += kPkThreads)
    if (dep[src[i]] >= 0 && src[i] != dst[i]) place[in_ptr[dst[i]] + atomicAdd(&c_cnt[dst[i]], 1)] = i;
  __syncthreads();
  for (int p = tid; p < n_dp; p += kPkThreads) {
    const int i = place[p], d2 = dst[i];
    const int q0 = in_ptr[d2], q1 = in_ptr[d2 + 1];
    int rank = 0;
    for (int q = q0; q < q1; ++q) rank += place[q] < i ? 1 : 0;
    in_list[q0 + rank] = canon[i];
  }
  __syncthreads();
  PK_STAMP(5);
  // ---- 4. level orders: states by (depth, in-degree falling, id) and by (height, out-degree falling, id)
  if (n_reach > kPkMaxKeys) { fail(NFST_ERR_LIMIT); return; }
  for (int dir = 0; dir < 2; ++dir) {
    const int32_t *lev = dir ? g_hei : g_dep, *deg = dir ? g_out : g_in;
    int32_t *ord = dir ? ord_b : ord_f, *lvlp = dir ? lvlp_b : lvlp_f;
    // (compaction of the reachable states: tmp_a = index among them)
    pk_scan<false>(n, [&](int s) { return g_dep[s] >= 0 ? 1 : 0; }, tmp_a, red);
    for (int s = tid; s < n; s += kPkThreads)
      if (g_dep[s] >= 0)
        pk_lds[tmp_a[s]] = ((unsigned long long)lev[s] << 40) | ((unsigned long long)(0xffffff - deg[s]) << 16) | (unsigned long long)s;
    __syncthreads();
    pk_sort(pk_lds, n_reach);
    for (int i = tid; i < n_reach; i += kPkThreads) {
      const unsigned long long k = pk_lds[i];
      ord[i] = (int)(k & 0xffffull);
      const int l = (int)(k >> 40), lprev = i > 0 ? (int)(pk_lds[i - 1] >> 40) : -1;
      for (int t = lprev + 1; t <= l; ++t) lvlp[t] = i;  // (levels 0 .. D are all inhabited: one store per level)
    }
    if (tid == 0) lvlp[D + 1] = n_reach;
    __syncthreads();
  }
  PK_STAMP(6);
  }  // (!EMIT)
  // canonical-id indexed views of the arcs (the lists hold canonical ids): other end and label
  // (emit: the canonical arrays themselves; plan: not needed)
  const int32_t arc_base = EMIT ? meta[NFST_META_ARC_OFF] : 0;
  if (EMIT) {
    const int32_t row_base = meta[NFST_META_ROW_OFF];
    int32_t *o_src = const_cast<int32_t *>(a.out.arc_src) + arc_base, *o_dst = const_cast<int32_t *>(a.out.arc_dst) + arc_base,
            *o_lab = const_cast<int32_t *>(a.out.arc_label) + arc_base;
    uint32_t *o_sd = const_cast<uint32_t *>(a.out.arc_sd) + arc_base;
    uint16_t *o_l16 = const_cast<uint16_t *>(a.out.arc_l16) + arc_base;
    float *o_w = a.out.weighted ? const_cast<float *>(a.out.arc_w) + arc_base : nullptr;
    for (int i = tid; i < A; i += kPkThreads)
      if (g_dep[src[i]] >= 0) {
        const int c = canon[i];
        o_src[c] = src[i]; o_dst[c] = dst[i]; o_lab[c] = lab[i];
        o_sd[c] = (uint32_t)src[i] | ((uint32_t)dst[i] << 16);
        o_l16[c] = (uint16_t)lab[i];
        if (o_w) o_w[c] = a.w[a0 + i];
      }
    // row pointers: arcs of the states before s (absolute ids), n + 1 entries
    int32_t *rp = const_cast<int32_t *>(a.out.row_ptr) + row_base + b;
    int *rc = reinterpret_cast<int *>(pk_lds);  // arcs per row, counted in LDS
    for (int s = tid; s < n; s += kPkThreads) rc[s] = 0;
    __syncthreads();
    for (int i = tid; i < A; i += kPkThreads)
      if (g_dep[src[i]] >= 0) atomicAdd(&rc[src[i]], 1);
    __syncthreads();
    pk_scan<false>(n + 1, [&](int s) { return s < n ? rc[s] : 0; }, tmp_a, red);
    for (int s = tid; s <= n; s += kPkThreads) rp[s] = arc_base + tmp_a[s];
    __syncthreads();
  }
  PK_STAMP(7);
  // ---- 5. tile programs
  PkDir dirs[2];
  dirs[0] = PkDir{ord_f, lvlp_f, g_dep, in_ptr, in_list, EMIT ? a.out.arc_src + arc_base : nullptr, EMIT ? a.out.arc_label + arc_base : nullptr, n_reach, D, n};
  dirs[1] = PkDir{ord_b, lvlp_b, g_hei, out_ptr, out_list, EMIT ? a.out.arc_dst + arc_base : nullptr, EMIT ? a.out.arc_label + arc_base : nullptr, n_reach, D, n};
  int tiles[2], wide[2], scratch_rows = 0;
  for (int dir = 0; dir < 2; ++dir) {
    if (!EMIT) {
      // the cheapest of narrow (8-lane groups) and wide (64-lane) programs by pack.cpp's cost model
      PkLayout best{0, 0, 0, 0};
      int best_wide = 0;
      double best_cost = 0.0;
      bool have = false;
      // (a wide program differs from the narrow one only if some state has more arcs than a narrow group holds)
      int dmax_t = 0, dmax = 0;
      for (int s = tid; s < n; s += kPkThreads) dmax_t = max(dmax_t, dirs[dir].ptr[s + 1] - dirs[dir].ptr[s]);
      pk_block_excl<true>(dmax_t, red, &dmax);
      __syncthreads();
      for (int w = 0; w < 2; ++w) {
        if ((a.group_mode == 1 && w) || (a.group_mode == 2 && !w)) continue;
        if (w && dmax <= 32 && a.group_mode != 2) continue;  // no group beyond 8 lanes: the same program as the narrow one
        PkLayout l = pk_count_layout(dirs[dir], w ? 6 : 3, reinterpret_cast<int32_t *>(pk_lds), (int)(kPkLdsBytes / 4), aux_cap, red);
        if (l.err == NFST_ERR_ARG)  // (tables of levels x passes beyond LDS: count by laying the tiles out)
          l = pk_layout<false>(dirs[dir], w ? 6 : 3, a.vocab, pk_lds, tmp_a, tmp_b, pa_, pb_, pc_, tile_bits, aux_cap, red, nullptr, nullptr, 0);
        if (l.err) { fail(l.err); return; }
        if (w && l.wide == 0 && a.group_mode != 2) continue;  // the same program as the narrow one
        const double cost = (double)l.tiles * (330.0 + 55.0 * 4 + (w ? 60.0 : 0.0)) + 450.0 * l.wide;
        if (!have || cost < best_cost) { have = true; best_cost = cost; best = l; best_wide = w; }
      }
      tiles[dir] = best.tiles; wide[dir] = best_wide;
      scratch_rows = max(scratch_rows, best.scratch);
    } else {
      const int w = (meta[dir ? NFST_META_BWD_U : NFST_META_FWD_U] >> 8) & 1;
      uint32_t *stream = const_cast<uint32_t *>(dir ? a.out.bwd_stream : a.out.fwd_stream) + meta[dir ? NFST_META_BWD_OFF : NFST_META_FWD_OFF];
      int32_t *perm = const_cast<int32_t *>(dir ? a.out.bwd_perm : a.out.fwd_perm) + meta[dir ? NFST_META_BWD_SLOT_OFF : NFST_META_FWD_SLOT_OFF];
      const PkLayout l = pk_layout<true>(dirs[dir], w ? 6 : 3, a.vocab, pk_lds, tmp_a, tmp_b, pa_, pb_, pc_, tile_bits, aux_cap, red, stream, perm, arc_base);
      if (l.err || l.tiles != meta[dir ? NFST_META_BWD_TILES : NFST_META_FWD_TILES]) { fail(l.err ? l.err : NFST_ERR_ARG); return; }
    }
  }
  PK_STAMP(8);
  if (!EMIT && tid == 0) {
    meta[NFST_META_N_ROWS] = n; meta[NFST_META_N_ARCS] = n_arcs; meta[NFST_META_FWD_TILES] = tiles[0]; meta[NFST_META_BWD_TILES] = tiles[1];
    meta[NFST_META_SINK] = sink; meta[NFST_META_N_REACH] = n_reach; meta[NFST_META_DEPTH] = D; meta[NFST_META_N_DP] = n_dp;
    meta[NFST_META_FWD_U] = 8 | (wide[0] << 8); meta[NFST_META_BWD_U] = 8 | (wide[1] << 8);
    a.scratch[b] = scratch_rows;
    a.status[b] = 0;
  }
}

// ---- dense tables on the device -> arc lists ------------------------------------------------------------------------
// The reference's set_masks receives emission [B, R, V] (bool, or float log weights with -inf for "no arc") and
// transition [B, R, V] int64 (scorers.py:877-885), collated with pad-id padding rows (util/dataset_reader.py:175-186:
// bool rows all True, transition = pad).  Only rows reachable from state 0 become arcs.
// pass 1 (k_dense_reach): breadth-first over the rows in LDS; reach [B, R] bytes, row_cnt [B, R] arcs per reachable
// row, counts [B]; status [B] = NFST_ERR_INDEX for a destination outside [0, R).
template <bool FLOAT>
__device__ __forceinline__ bool pk_has(const void *em, size_t at) {
  if (FLOAT) return reinterpret_cast<const float *>(em)[at] > kNegInf;
  return reinterpret_cast<const uint8_t *>(em)[at] != 0;
}
template <bool FLOAT>
__global__ __launch_bounds__(kPkThreads) void k_dense_reach(const void *em, const int64_t *tr, int R, int V, uint8_t *reach, int32_t *row_cnt,
                                                           int32_t *counts, int32_t *status) {
  extern __shared__ unsigned long long pk_lds[];
  __shared__ int red[64];
  __shared__ int sh[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *seen = reinterpret_cast<int *>(pk_lds), *front = seen + R, *next = front + R;  // 12 R bytes <= 96 KiB
  const size_t base = (size_t)b * R * V;
  for (int s = tid; s < R; s += kPkThreads) seen[s] = s == 0;
  if (tid == 0) { front[0] = 0; sh[0] = 1; sh[1] = 0; sh[2] = 0; }
  __syncthreads();
  for (;;) {
    const int nf = sh[0];
    if (nf == 0) break;
    for (int f = wave; f < nf; f += kPkThreads / 64) {  // one wave per frontier row
      const int s = front[f];
      for (int l = lane; l < V; l += 64) {
        const size_t at = base + (size_t)s * V + l;
        if (!pk_has<FLOAT>(em, at)) continue;
        const int64_t d = tr[at];
        if (d < 0 || d >= R) { sh[2] = 1; continue; }
        if (atomicExch(&seen[(int)d], 1) == 0) next[atomicAdd(&sh[1], 1)] = (int)d;
      }
    }
    __syncthreads();
    if (tid == 0) { sh[0] = sh[1]; sh[1] = 0; }
    int *t = front; front = next; next = t;
    __syncthreads();
  }
  int acc = 0;
  for (int s = wave; s < R; s += kPkThreads / 64) {
    int c = 0;
    if (seen[s])
      for (int l = lane; l < V; l += 64) c += pk_has<FLOAT>(em, base + (size_t)s * V + l) ? 1 : 0;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d);
    if (lane == 0) { row_cnt[(size_t)b * R + s] = c; reach[(size_t)b * R + s] = (uint8_t)seen[s]; acc += c; }
  }
  int total;
  pk_block_excl<false>(acc, red, &total);
  if (tid == 0) { counts[b] = total; status[b] = sh[2] ? NFST_ERR_INDEX : 0; }
}
// pass 2 (k_dense_write): the arcs of lattice b at arc_off[b], in (row, label) order -- the canonical order
template <bool FLOAT>
__global__ __launch_bounds__(kPkThreads) void k_dense_write(const void *em, const int64_t *tr, int R, int V, const uint8_t *reach,
                                                           const int32_t *row_cnt, const int64_t *arc_off, int32_t *src, int32_t *label,
                                                           int32_t *dst, float *w) {
  extern __shared__ unsigned long long pk_lds[];
  __shared__ int red[64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int32_t *start = reinterpret_cast<int32_t *>(pk_lds);  // [R] first arc of every row
  const int32_t *rc = row_cnt + (size_t)b * R;
  pk_scan<false>(R, [&](int s) { return rc[s]; }, start, red);
  const size_t base = (size_t)b * R * V;
  const int64_t a0 = arc_off[b];
  for (int s = wave; s < R; s += kPkThreads / 64) {
    if (!reach[(size_t)b * R + s]) continue;
    int64_t at_arc = a0 + start[s];
    for (int l0 = 0; l0 < V; l0 += 64) {
      const int l = l0 + lane;
      const size_t at = base + (size_t)s * V + l;
      const bool has = l < V && pk_has<FLOAT>(em, at);
      const unsigned long long m = __ballot(has);
      if (has) {
        const int64_t p = at_arc + __builtin_popcountll(m & ((1ull << lane) - 1ull));
        src[p] = s; label[p] = l; dst[p] = (int32_t)tr[at];
        if (FLOAT) w[p] = reinterpret_cast<const float *>(em)[at];
      }
      at_arc += __builtin_popcountll(m);
    }
  }
}
