// chunk_kernels.h -- the chunked flavour of the alpha / beta sweeps for deep, narrow lattices (include/nfst_hip.h,
// "Chunked programs"; host side: chunk_pack.cpp).  Part of the single translation unit kernels.hip.
//
// The general kernels walk a lattice level by level: one ~200 ns tile per level whatever its fill, so a SNIPS-shaped
// tagging machine (750 levels of 2..8 states, /root/reference/src/conf/train/lstm_snips.yaml:2) is a 150 us chain on a
// CU that is almost idle.  Here the positions (states in topological order) are cut into C chunks that are swept at
// the same time, each with one right-hand side per frontier state; the chain that remains is C small matrix-vector
// steps.  Same function as the general kernels: sums over the same arcs, in float64.
#pragma once

struct ChkWs {
  uint4 *rec;     // [n_stream] the entries as pass 1 walks them: {weight (float64, two words), last-entry flag (sign bit), operand offset}
  double *T;      // [t_units * 64]
  Rec64 *me;      // [2][total_rows] alpha, beta of every row as (mantissa, exponent)
  Rec64 *zme;     // [n_lattices]
  int *flags;     // [n_lattices] == tag of the launch: the lattice left the float64 range, the general kernels run it
};
__host__ __device__ inline ChkWs chk_ws(const nfst_chunks &c) {
  char *p = (char *)c.ws;
  ChkWs w;
  w.rec = (uint4 *)p; p += c.n_stream * 16;
  w.T = (double *)p; p += c.t_units * 512;
  w.me = (Rec64 *)p; p += c.total_rows * 2 * 16;
  w.zme = (Rec64 *)p; p += (int64_t)c.n_lattices * 16;
  w.flags = (int *)p;
  return w;
}

constexpr int kChkAhead = 8;   // entries a lane of pass 1 keeps in flight (NFST_CHK_SLACK in chunk_pack.cpp: twice that)
constexpr int kChkNotYet = 0x7fffffff;  // vemax[c] before pass 2 has published chunk c's frontier values
constexpr int kChkRange = 480;  // |binary exponent| of every weight and partial sum of pass 1: products stay normal

// log weight of canonical arc `a` (absolute) of lattice b, in float64 (the oracle's sum)
__device__ __forceinline__ double chk_score(const nfst_batch &lat, const nfst_scores &sc, int b, int a) {
  double s = (double)sc.theta[(size_t)sc.theta_stride * b + lat.arc_label[a]];
  if (lat.weighted && lat.arc_w) s += (double)lat.arc_w[a];
  if (sc.arc_scores) s += (double)sc.arc_scores[a];
  return s;
}

__device__ __forceinline__ double chk_readlane(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// Pass 2 of k_chunk_sweep on one wave: the frontier values chained through the chunks.  Lane i holds the value at position
// a_c - 1 - i as (m, e); a step scales them to lane 0's exponent, multiplies with the F x F block of T the chunk's ring
// still holds, and normalises.  One wave, in order: what a step costs is its instruction count (~320 ns at F = 8, a third
// less at F <= 2: NB = the row entries a lane keeps, 2, 4 or 8; frontiers beyond 8 states take a tail loop).  The chain is
// ALU only -- the frontier values reach the other lanes by v_readlane (the scaled value of lane f is a scalar operand of
// lane i's multiply-add) -- and the first NB entries of a lane's row of T for the NEXT step are read from the ring while
// this step computes, from an offset looked up one step before that (roff, filled by all threads before pass 1).
// Returns non-zero when the frontier values of a step lie further apart than 2^250 (the lattice goes to the general kernels).
template <int NB>
__device__ __forceinline__ unsigned chk_chain(const double *ring, double *vs, int *vemax, const int *roff, int C, int F, int lane) {
  unsigned bad = 0;
  double m = (lane == 0) ? 1.0 : 0.0;  // chunk 0: position 0 (the start / the sink) has value one
  int e = (lane == 0) ? 0 : kEZero;
  const char *lds0 = (const char *)ring;
  const int fl = min(lane, F - 1);
  auto load = [&](int off, double (&r)[NB]) {
    const double *row = (const double *)(lds0 + max(off, 0));
#pragma unroll
    for (int j = 0; j < NB; ++j) r[j] = row[j];  // (entries beyond F meet the zero of an idle lane)
  };
  int off = roff[fl];                         // step 0
  int off1 = roff[min(1, C - 1) * F + fl];    // step 1
  double r[NB];
  load(off, r);
  for (int c = 0; c < C; ++c) {
    // a common exponent for the step: lane 0's (the frontier state next to the chunk); the others must lie within
    // 2^+-250 of it
    int eref = __builtin_amdgcn_readfirstlane(e);
    if (eref == kEZero)
      for (int f = 1; f < F; ++f) eref = max(eref, __builtin_amdgcn_readlane(e, f));
    const int de = e - eref;
    bad |= (e != kEZero) & ((unsigned)(de + 250) > 500u);
    const double s = (e == kEZero) ? 0.0 : ldexp(m, max(min(de, 1000), -1000));  // (lanes from F up: zero)
    if (lane < F) vs[c * F + lane] = s;
    asm volatile("" ::: "memory");  // (the exponent publishes the chunk: behind the values, and LDS executes a wave's accesses in order)
    if (lane == 0) vemax[c] = eref;
    if (c + 1 == C) break;
    // the next steps' operands (off the chain)
    const int off2 = roff[min(c + 2, C - 1) * F + fl];
    double rn[NB];
    load(off1, rn);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j & 3] = fma(r[j], chk_readlane(s, j), acc[j & 3]);
    if (NB == 8) {
      for (int f0 = 8; f0 < F; f0 += 8) {  // (frontiers beyond eight states: these reads are on the chain)
        load(off + f0 * 8, r);
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[j & 3] = fma(r[j], chk_readlane(s, f0 + j), acc[j & 3]);
      }
    }
    const double sum = NB == 2 ? acc[0] + acc[1] : (acc[0] + acc[1]) + (acc[2] + acc[3]);
    const bool inside = lane < F && off >= 0;
    double m_new = inside ? __builtin_amdgcn_frexp_mant(sum) : 0.0;
    int e_new = (inside && sum != 0.0) ? eref + __builtin_amdgcn_frexp_exp(sum) : kEZero;
    if (__any(lane < F && off <= -2)) {  // a chunk shorter than the frontier: positions below it pass through
      const bool below = lane < F && off <= -2;
      const int src_lane = below ? -2 - off : lane;
      const double m_pass = __shfl(m, src_lane);
      const int e_pass = __shfl(e, src_lane);
      if (below) { m_new = m_pass; e_new = e_pass; }
    }
    m = m_new;
    e = e_new;
    off = off1; off1 = off2;
#pragma unroll
    for (int j = 0; j < NB; ++j) r[j] = rn[j];
  }
  return bad;
}

// One workgroup per (lattice, direction).  n_dirs = 2: blockIdx = 2 * lattice + direction (0 alpha, 1 beta);
// n_dirs = 1: the beta programs only (nfst_backward).
__global__ __launch_bounds__(1024) void k_chunk_sweep(nfst_batch lat, nfst_scores sc, nfst_chunks ck, int tag, int n_dirs,
                                                       float *__restrict__ logalpha, float *__restrict__ logbeta,
                                                       double *__restrict__ logz64, float *__restrict__ logz32,
                                                       float *__restrict__ grad_theta, float2 *__restrict__ beta_me) {
  extern __shared__ double chk_lds[];
  const int tid = threadIdx.x, NT = blockDim.x, lane = tid & 63;
  const int b = n_dirs == 2 ? (int)(blockIdx.x >> 1) : (int)blockIdx.x;
  const int dir = n_dirs == 2 ? (int)(blockIdx.x & 1) : 1;
  const int32_t *cm = ck.meta + ((size_t)b * 2 + dir) * NFST_CHK_META_WORDS;
  const int C = cm[NFST_CHK_C], F = cm[NFST_CHK_F], R = cm[NFST_CHK_R], npos = cm[NFST_CHK_NPOS];
  const int32_t *tab = ck.tab + (size_t)cm[NFST_CHK_TAB_OFF] * 4;
  const int32_t *pos = ck.pos + cm[NFST_CHK_POS_OFF];
  const int32_t *lm = lat.meta + (size_t)b * NFST_META_WORDS;
  const int row_off = lm[NFST_META_ROW_OFF], n_rows = lm[NFST_META_N_ROWS], arc_off = lm[NFST_META_ARC_OFF];
  const ChkWs ws = chk_ws(ck);
  uint4 *rec = ws.rec + cm[NFST_CHK_STREAM_OFF];
  double *T = ws.T + (size_t)cm[NFST_CHK_T_OFF] * 64;
  Rec64 *me = ws.me + (size_t)dir * ck.total_rows + row_off;
  float *logout = dir == 0 ? logalpha : logbeta;
  // LDS: the rings (a padded block of R * F + F doubles per chunk), the scaled frontier values of every chunk, their
  // common exponents, the chunks' first positions
  const int ring_stride = R * F + F;
  double *ring = chk_lds;
  double *vs = ring + (size_t)C * ring_stride + 8;  // [C][F]  (8 doubles of zeros between: pass 2 reads rows 8 entries at a time)
  int *vemax = (int *)(vs + (size_t)C * F);         // [C]
  int *a_tab = vemax + C;                           // [C + 1]
  int *roff = a_tab + C + 1;                        // [C][F] pass 2: where lane i finds its row of T in step c (see below)
  const int n_entries = tab[(C - 1) * 4 + 1] + tab[(C - 1) * 4 + 2];
  unsigned bad = 0;
  if (tid == 0) NFST_STAMP(0);  // (profiling build: profiles/tune/chunk_stamps.py)
  // ---- the weights of this program's entries, in walking order: eight entries per thread and trip, every load of a
  // stage issued before the first use (entry and label, then the scores: dependent loads from HBM / L2)
  {
    const uint32_t *__restrict__ stream = ck.stream + cm[NFST_CHK_STREAM_OFF];
    const uint16_t *__restrict__ labels = ck.label + cm[NFST_CHK_STREAM_OFF];  // (beside the entries: entry -> arc -> label would be a third load in the chain)
    const bool has_w = lat.weighted && lat.arc_w, has_s = sc.arc_scores != nullptr;
    const float *__restrict__ th = sc.theta + (size_t)sc.theta_stride * b;
    constexpr int kW = 8;
    for (int k0 = tid; k0 < n_entries; k0 += kW * NT) {
      uint32_t e[kW];
      int arc[kW], lab[kW];
      float t[kW], xw[kW], xs[kW];
#pragma unroll
      for (int j = 0; j < kW; ++j) e[j] = stream[min(k0 + j * NT, n_entries - 1)];
#pragma unroll
      for (int j = 0; j < kW; ++j) { arc[j] = arc_off + (int)(e[j] >> 8); lab[j] = labels[min(k0 + j * NT, n_entries - 1)]; }
#pragma unroll
      for (int j = 0; j < kW; ++j) {
        t[j] = th[lab[j]];
        xw[j] = has_w ? lat.arc_w[arc[j]] : 0.0f;
        xs[j] = has_s ? sc.arc_scores[arc[j]] : 0.0f;
      }
#pragma unroll
      for (int j = 0; j < kW; ++j) {
        const int k = k0 + j * NT;
        if (k >= n_entries) break;
        double w = 0.0;
        if (!(e[j] & NFST_CHK_ZERO)) {
          const double sco = (double)t[j] + (double)xw[j] + (double)xs[j];
          const ME64 x = exp_split64(sco);
          bad |= (sco != sco) | ((x.e != kEZero) & ((unsigned)(x.e + kChkRange) > 2u * kChkRange));
          w = (x.e == kEZero) ? 0.0 : ldexp(x.m, max(min(x.e, 1000), -1000));
        }
        // .z: the sign bit says "last entry of its state" (one compare in the walk); .w: the byte offset of the operand's slot
        rec[k] = make_uint4((uint32_t)__double2loint(w), (uint32_t)__double2hiint(w), (e[j] & NFST_CHK_LAST) ? 0x80000000u : 0u,
                            (e[j] & 63u) * (uint32_t)(F * 8));
      }
    }
  }
  for (int i = tid; i < C * ring_stride + 8 + C * F; i += NT) ring[i] = 0.0;  // (rings, the gap, vs)
  for (int i = tid; i <= C; i += NT) a_tab[i] = i < C ? tab[i * 4] : npos;
  for (int i = tid; i < C; i += NT) vemax[i] = kChkNotYet;  // (pass 3 follows pass 2 chunk by chunk: "not published yet")
  // pass 2, step c (frontier of chunk c -> frontier of chunk c + 1): lane i computes the value at position p = a_{c+1} - 1 - i.
  // roff >= 0: the byte offset of p's row of T in the rings (p inside chunk c); -1: no such position (value zero);
  // -2 - f: p lies below chunk c (a chunk shorter than the frontier): the value of this step's frontier lane f passes through
  for (int i = tid; i < C * F; i += NT) {
    const int c = i / F, l = i - c * F;
    const int a_c = tab[c * 4], a_n = c + 1 < C ? tab[(c + 1) * 4] : npos;
    const int p = a_n - 1 - l;
    roff[i] = p >= a_c ? (c * ring_stride + (p & (R - 1)) * F) * 8 : (p >= 0 ? -2 - (a_c - 1 - p) : -1);
  }
  if (grad_theta && dir == 1)  // (k_chunk_post adds its workgroups' per-label sums into it)
    for (int l = tid; l < lat.vocab; l += NT) grad_theta[(size_t)b * lat.vocab + l] = 0.0f;
  // rows the program does not reach: -inf / zero (every reached row is overwritten below)
  float2 *me32 = (beta_me && dir == 1) ? beta_me + row_off : nullptr;  // beta as float32 (mantissa, exponent) pairs: what nfst_sample_paths reads
  for (int i = tid; i < n_rows; i += NT) {
    if (logout) logout[row_off + i] = kNegInf;
    me[i] = Rec64{0.0, kEZero, 0};
    if (me32) me32[i] = make_float2(0.0f, __int_as_float(kEZero));
  }
  // (what the lanes read below was written by this workgroup: a workgroup barrier orders it -- an agent-scope fence here
  // wrote the XCD's L2 back, 60 .. 100 us with 128 workgroups doing it)
  __syncthreads();
  if (tid == 0) NFST_STAMP(1);
  // ---- pass 1: lane (c, f) sweeps chunk c from the unit vector on frontier position a_c - 1 - f
  const int c1 = tid / F, f1 = tid - c1 * F;
  if (c1 < C) {
    const int a_c = tab[c1 * 4];
    double *ringc = ring + (size_t)c1 * ring_stride + f1;
    const int q = a_c - 1 - f1;
    if (q >= 0) ringc[(q & (R - 1)) * F] = 1.0;
    const uint4 *rp = rec + tab[c1 * 4 + 1];
    const int cnt = tab[c1 * 4 + 2];
    double *__restrict__ Tp = T + (size_t)a_c * F + f1;
    double acc = 0.0;
    char *rb = (char *)ringc;
    const int fstep = F * 8, wend = R * F * 8;
    int woff = (a_c & (R - 1)) * fstep;  // byte offset of the slot the state being summed is written to
    // One entry: operand from the lane's ring (byte offset precomputed in .w), multiply-add; on the last entry of a state
    // the sum goes to the ring and to T.  A chunk's entries are padded to a multiple of kChkAhead with zero-weight
    // entries (chunk_pack.cpp), so no entry needs a bounds test.  The walk is bound by instruction issue (two to four waves
    // per SIMD, all busy): ~12 instructions per entry.
#define NFST_CHK_ENTRY(E)                                                         \
    {                                                                             \
      const double w = __hiloint2double((int)(E).y, (int)(E).x);                  \
      acc = fma(*(const double *)(rb + (E).w), w, acc);                           \
      if ((int)(E).z < 0) {                                                       \
        *(double *)(rb + woff) = acc;                                             \
        *Tp = acc;                                                                \
        Tp += F;                                                                  \
        woff += fstep;                                                            \
        woff = woff == wend ? 0 : woff;                                           \
        acc = 0.0;                                                                \
      }                                                                           \
    }
    // the entries come from L2 (this workgroup wrote them a moment ago): kChkAhead of them are in flight while the
    // previous kChkAhead are walked (one entry ahead left the lane waiting ~300 ns per entry)
    uint4 ba[kChkAhead], bb[kChkAhead];
#pragma unroll
    for (int j = 0; j < kChkAhead; ++j) ba[j] = rp[j];
    for (int k0 = 0; k0 < cnt; k0 += 2 * kChkAhead) {
#pragma unroll
      for (int j = 0; j < kChkAhead; ++j) bb[j] = rp[k0 + kChkAhead + j];  // (the stream has slack behind its last entry)
#pragma unroll
      for (int j = 0; j < kChkAhead; ++j) NFST_CHK_ENTRY(ba[j])
#pragma unroll
      for (int j = 0; j < kChkAhead; ++j) ba[j] = rp[k0 + 2 * kChkAhead + j];
      if (k0 + kChkAhead < cnt) {
#pragma unroll
        for (int j = 0; j < kChkAhead; ++j) NFST_CHK_ENTRY(bb[j])
      }
    }
#undef NFST_CHK_ENTRY
  }
  __syncthreads();
  if (tid == 0) NFST_STAMP(2);
  // ---- pass 2 (wave 0): the frontier values chained through the chunks (chk_chain above)
  if (tid < 64) {
    __builtin_amdgcn_s_setprio(3);  // (the waves that follow with pass 3 share its SIMD)
    if (F <= 2) bad |= chk_chain<2>(ring, vs, vemax, roff, C, F, lane);
    else if (F <= 4) bad |= chk_chain<4>(ring, vs, vemax, roff, C, F, lane);
    else bad |= chk_chain<8>(ring, vs, vemax, roff, C, F, lane);
    __builtin_amdgcn_s_setprio(0);
  }
  if (tid == 0) NFST_STAMP(3);
  // ---- pass 3: every position's value from its row of T and its chunk's frontier values, one position per thread and
  // trip (its chunk by bisection in LDS).  The range test of pass 1's sums happens here, off its chain.  (Tried: rows read
  // eight entries at a time into two accumulators, and the positions of a chunk dealt to its pass-1 lanes -- 4 .. 7 us
  // slower per launch on the batch of 64 than this plain loop.)
  // The waves beside wave 0 do not wait for pass 2 to end: a position is computed as soon as its chunk's frontier values
  // are published (its exponent in vemax); positions go up with the trips, so does the chunk a thread waits for.  A workgroup of one wave
  // does the passes one after the other.
  const bool follow = NT > 64;
  for (int p = follow ? tid - 64 : tid; p < npos; p += follow ? NT - 64 : NT) {
    if (p < 0) break;  // (wave 0 of a larger workgroup: pass 2 was its part)
    Rec64 v;
    if (p == 0) {
      v = Rec64{0.5, 1, 0};  // the start / the sink
    } else {
      int lo = 0, hi = C;  // the chunk with a_tab[c] <= p < a_tab[c + 1]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (a_tab[mid] <= p) lo = mid; else hi = mid;
      }
      if (follow) {
        while (((volatile int *)vemax)[lo] == kChkNotYet) __builtin_amdgcn_s_sleep(4);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      }
      const double *row = T + (size_t)p * F;
      const double *sv = vs + lo * F;
      double acc = 0.0;
      for (int f = 0; f < F; ++f) {
        const double t = row[f];
        const unsigned ex = ((unsigned)__double2hiint(t) >> 20) & 0x7ffu;
        bad |= (t != 0.0) & (ex - (unsigned)(1023 - kChkRange) > 2u * kChkRange);
        acc = fma(t, sv[f], acc);
      }
      v = me_pack64(acc, vemax[lo]);
    }
    const int state = pos[p];
    me[state] = v;
    if (me32) me32[state] = me_f2(v);
    const double lg = me_log64(v);
    if (logout) logout[row_off + state] = (float)lg;
    if (dir == 1 && p == npos - 1) {
      ws.zme[b] = v;
      if (logz64) logz64[b] = lg;
      if (logz32) logz32[b] = (float)lg;
    }
  }
  if (bad) ws.flags[b] = tag;
#ifdef NFST_PROF
  __syncthreads();
  if (tid == 0) NFST_STAMP(4);
#endif
}

// Arc posteriors, the per-label sums and the batch total from the (m, e) values the sweeps left: `parts` workgroups of
// 256 threads per lattice, each with a slice of its arcs (about four arcs per thread: one trip, every load of a stage
// issued before the first use; 64 workgroups of 1024 threads took 15 us for 360k arcs).  grad_theta was zeroed by the
// sweep kernel.
__global__ __launch_bounds__(256) void k_chunk_post(nfst_batch lat, nfst_scores sc, nfst_chunks ck, int tag, int parts,
                                                      float *__restrict__ posterior, float *__restrict__ grad_theta,
                                                      double *__restrict__ logz_total, int total_slot) {
  extern __shared__ double chk_lds[];
  float *gth = (float *)chk_lds;  // [V] (only with grad_theta)
  const int tid = threadIdx.x, NT = blockDim.x, b = blockIdx.x / parts, part = blockIdx.x - b * parts;
  const ChkWs ws = chk_ws(ck);
  const bool flagged = ws.flags[b] == tag;
  const Rec64 z = ws.zme[b];
  if (tid == 0 && part == 0 && logz_total) {
    if (!flagged) atomicAdd(&logz_total[total_slot], me_log64(z));
    if (b == 0) logz_total[(total_slot + 1) % 3] = 0.0;
  }
  if (flagged || (!posterior && !grad_theta)) return;
  const int32_t *lm = lat.meta + (size_t)b * NFST_META_WORDS;
  const int row_off = lm[NFST_META_ROW_OFF], arc_off = lm[NFST_META_ARC_OFF], n_arcs = lm[NFST_META_N_ARCS];
  const int begin = arc_off + (int)((int64_t)n_arcs * part / parts), end = arc_off + (int)((int64_t)n_arcs * (part + 1) / parts);
  if (begin >= end) return;
  const Rec64 *__restrict__ al = ws.me + row_off, *__restrict__ be = ws.me + ck.total_rows + row_off;
  const float *__restrict__ th = sc.theta + (size_t)sc.theta_stride * b;
  const bool has_w = lat.weighted && lat.arc_w, has_s = sc.arc_scores != nullptr;
  if (grad_theta) {
    for (int l = tid; l < lat.vocab; l += NT) gth[l] = 0.0f;
    __syncthreads();
  }
  const double rz = z.m > 0.0 ? 1.0 / z.m : 0.0;
  for (int a0 = begin + tid; a0 < end; a0 += 4 * NT) {
    int a[4], s[4], d[4], lab[4];
    float xw[4], xs[4], t[4];
    Rec64 va[4], vb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = min(a0 + j * NT, end - 1);
      s[j] = lat.arc_src[a[j]]; d[j] = lat.arc_dst[a[j]]; lab[j] = lat.arc_label[a[j]];
      xw[j] = has_w ? lat.arc_w[a[j]] : 0.0f;
      xs[j] = has_s ? sc.arc_scores[a[j]] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { t[j] = th[lab[j]]; va[j] = al[s[j]]; vb[j] = be[d[j]]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (a0 + j * NT >= end) break;
      float p = 0.0f;
      if (s[j] != d[j]) {
        const ME64 x = exp_split64((double)t[j] + (double)xw[j] + (double)xs[j]);
        const double mm = va[j].m * x.m * vb[j].m * rz;
        const int ee = max(va[j].e + x.e + vb[j].e - z.e, -4000);  // (zeros carry kEZero: far below, never wraps)
        p = (float)ldexp(mm, min(ee, 4000));
      }
      if (posterior) __builtin_nontemporal_store(p, posterior + a[j]);
      if (grad_theta && p > 0.0f) atomicAdd(&gth[lab[j]], p);
    }
  }
  if (grad_theta) {
    __syncthreads();
    float *gout = grad_theta + (size_t)b * lat.vocab;
    for (int l = tid; l < lat.vocab; l += NT)
      if (gth[l] != 0.0f) atomicAdd(&gout[l], gth[l]);
  }
}
