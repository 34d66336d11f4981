/*
 * nfst_oracle.c -- CPU restatement of the reference's lattice algorithms.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: it
 * is imported only by tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg, and only as the checker (never as the thing measured as
 * the GPU result, never as a fallback).  The product path is nfst_amd/ + the
 * HIP library and fails loudly without it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks these functions
 * against the .npz files in tests/golden/, which tests/golden/make_golden.py produced by
 * running the reference's own code (/root/reference/src/modules) in the
 * development container.
 *
 * What is restated, and from where (all paths under /root/reference/src):
 *   orc_dense_to_arcs      the dense table encoding of a lattice --
 *                          modules/scorers.py:995-1035 (get_state_mask_pynini)
 *                          and the arc test used by the beta sweep,
 *                          scorers.py:705-716 (`to_state != 0 and != i`).
 *   orc_forward_backward   compute_beta_per_sample, scorers.py:692-751, with
 *                          Wh = 0 (arc weight = exp(score)); sums over paths in
 *                          float64 log space.  The alpha sweep, log-Z and arc
 *                          posteriors do not exist in the reference; they are
 *                          the same sum run from the start state and are tied
 *                          to beta by identities (tests/test_oracle_golden.py).
 *   orc_viterbi            max-plus version of the same recursion (float32 so
 *                          that the GPU result can be compared bit for bit).
 *   orc_sample_paths       ancestral walk of samplers.py:243-297 with the exact
 *                          posterior as proposal and inverse-CDF draws from
 *                          supplied uniforms.
 *   orc_score_paths        forced walk (samplers.py:208-218, 258-259).
 *   orc_beta_dense_frontier  compute_beta_parallel, scorers.py:753-856,
 *                          restated with its dense [S,S] edge table, its
 *                          frontier loop and its H-dimensional messages -- this
 *                          is "the reference CPU path" that bench.py times.
 *
 * Error codes: 0 ok, -1 cycle, -2 not exactly one sink, -3 bad index,
 *              -4 inconsistent table, -5 buffer too small.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_OK 0
#define ORC_ERR_CYCLE -1
#define ORC_ERR_SINK -2
#define ORC_ERR_INDEX -3
#define ORC_ERR_TABLE -4
#define ORC_ERR_SPACE -5

/* ------------------------------------------------------------------------- */
/* dense tables -> arc list, (state asc, label asc) over states reachable from 0.
 * emission_bool (uint8) or emission_w (float32, -inf = no arc): exactly one is
 * non-NULL.  The sink's pad self loop is kept (the sampler walks it); the DP
 * functions below skip self loops like scorers.py:711 does. */
int64_t orc_dense_to_arcs(const uint8_t *emission_bool, const float *emission_w,
                          const int64_t *transition, int n_rows, int V,
                          int32_t *src, int32_t *label, int32_t *dst, float *w,
                          int64_t cap) {
  uint8_t *seen = (uint8_t *)calloc((size_t)n_rows, 1);
  int32_t *stack = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int sp = 0;
  int64_t rc = 0;
  seen[0] = 1;
  stack[sp++] = 0;
  while (sp > 0) {
    int s = stack[--sp];
    for (int l = 0; l < V; ++l) {
      size_t at = (size_t)s * V + l;
      int has = emission_bool ? (emission_bool[at] != 0) : (emission_w[at] > -INFINITY);
      if (!has) continue;
      int64_t d = transition[at];
      if (d < 0 || d >= n_rows) { rc = ORC_ERR_INDEX; goto done; }
      if (!seen[d]) { seen[d] = 1; stack[sp++] = (int32_t)d; }
    }
  }
  {
    int64_t n = 0;
    for (int s = 0; s < n_rows; ++s) {
      if (!seen[s]) continue;
      for (int l = 0; l < V; ++l) {
        size_t at = (size_t)s * V + l;
        int has = emission_bool ? (emission_bool[at] != 0) : (emission_w[at] > -INFINITY);
        if (!has) continue;
        if (n >= cap) { rc = ORC_ERR_SPACE; goto done; }
        src[n] = s; label[n] = l; dst[n] = (int32_t)transition[at];
        if (w) w[n] = emission_w ? emission_w[at] : 0.0f;
        ++n;
      }
    }
    rc = n;
  }
done:
  free(seen); free(stack);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Topological order of the states reachable from 0 (self loops ignored).
 * order[] receives the states, start first; returns their number or an error. */
static int topo_order(int n_rows, int64_t A, const int32_t *src, const int32_t *dst,
                      int32_t *order, int32_t *indeg, int32_t *outdeg,
                      int64_t *row_ptr /* n_rows+1, CSR by src over all arcs */) {
  for (int s = 0; s <= n_rows; ++s) row_ptr[s] = 0;
  for (int64_t a = 0; a < A; ++a) {
    if (src[a] < 0 || src[a] >= n_rows || dst[a] < 0 || dst[a] >= n_rows) return ORC_ERR_INDEX;
    if (a > 0 && src[a] < src[a - 1]) return ORC_ERR_TABLE; /* must be sorted by src */
    row_ptr[src[a] + 1]++;
  }
  for (int s = 0; s < n_rows; ++s) row_ptr[s + 1] += row_ptr[s];
  /* reachability from 0 */
  uint8_t *seen = (uint8_t *)calloc((size_t)n_rows, 1);
  int n = 0, head = 0;
  seen[0] = 1; order[n++] = 0;
  while (head < n) {
    int s = order[head++];
    for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a)
      if (!seen[dst[a]]) { seen[dst[a]] = 1; order[n++] = dst[a]; }
  }
  int n_reach = n;
  for (int s = 0; s < n_rows; ++s) { indeg[s] = 0; outdeg[s] = 0; }
  for (int64_t a = 0; a < A; ++a)
    if (seen[src[a]] && src[a] != dst[a]) { indeg[dst[a]]++; outdeg[src[a]]++; }
  /* Kahn */
  n = 0; head = 0;
  if (indeg[0] != 0) { free(seen); return ORC_ERR_CYCLE; }
  order[n++] = 0;
  while (head < n) {
    int s = order[head++];
    for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
      int d = dst[a];
      if (d == s) continue;
      if (--indeg[d] == 0) order[n++] = d;
    }
  }
  free(seen);
  if (n != n_reach) return ORC_ERR_CYCLE;
  return n;
}

static inline double lse2(double a, double b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  double m = a > b ? a : b;
  return m + log(exp(a - m) + exp(b - m));
}

/* log-sum-exp of v[0..n) (two passes, float64) */
static double lse_n(const double *v, int64_t n) {
  double m = -INFINITY;
  for (int64_t i = 0; i < n; ++i) if (v[i] > m) m = v[i];
  if (m == -INFINITY) return -INFINITY;
  double s = 0.0;
  for (int64_t i = 0; i < n; ++i) s += exp(v[i] - m);
  return m + log(s);
}

/* Exact path sums over one lattice.  Arcs sorted by src (canonical order).
 * score[a] is the arc's log weight.  Unreachable states get -inf everywhere.
 * Any of logalpha/logbeta/posterior may be NULL. */
int orc_forward_backward(int n_rows, int64_t A, const int32_t *src, const int32_t *dst,
                         const double *score, double *logalpha, double *logbeta,
                         double *posterior, double *logZ) {
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int32_t *indeg = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int32_t *outdeg = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int64_t *row_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n_rows + 1));
  double *la = (double *)malloc(sizeof(double) * (size_t)n_rows);
  double *lb = (double *)malloc(sizeof(double) * (size_t)n_rows);
  double *tmp = (double *)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
  int rc = ORC_OK;
  int n = topo_order(n_rows, A, src, dst, order, indeg, outdeg, row_ptr);
  if (n < 0) { rc = n; goto done; }
  {
    /* scorers.py:715-720: exactly one state without (non-self) out arcs */
    int sinks = 0, sink = -1;
    for (int i = 0; i < n; ++i) if (outdeg[order[i]] == 0) { ++sinks; sink = order[i]; }
    if (sinks != 1) { rc = ORC_ERR_SINK; goto done; }
    for (int s = 0; s < n_rows; ++s) { la[s] = -INFINITY; lb[s] = -INFINITY; }
    /* beta: reverse topological order; beta(sink) = 1 (scorers.py:720) */
    lb[sink] = 0.0;
    for (int i = n - 1; i >= 0; --i) {
      int s = order[i];
      if (s == sink) continue;
      int64_t k = 0;
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a)
        if (dst[a] != s) tmp[k++] = score[a] + lb[dst[a]];
      lb[s] = lse_n(tmp, k);
    }
    /* alpha: push along the topological order */
    la[0] = 0.0;
    for (int i = 0; i < n; ++i) {
      int s = order[i];
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a)
        if (dst[a] != s) la[dst[a]] = lse2(la[dst[a]], la[s] + score[a]);
    }
    double z = lb[0];
    if (logZ) *logZ = z;
    if (posterior)
      for (int64_t a = 0; a < A; ++a) {
        double v = (src[a] == dst[a]) ? -INFINITY : la[src[a]] + score[a] + lb[dst[a]] - z;
        posterior[a] = (v == -INFINITY || isnan(v)) ? 0.0 : exp(v);
      }
    if (logalpha) memcpy(logalpha, la, sizeof(double) * (size_t)n_rows);
    if (logbeta) memcpy(logbeta, lb, sizeof(double) * (size_t)n_rows);
  }
done:
  free(order); free(indeg); free(outdeg); free(row_ptr); free(la); free(lb); free(tmp);
  return rc;
}

/* Batch driver used by bench.py's cpu_baseline: lattices are independent, so
 * they are spread over OpenMP threads.  Arc arrays are concatenated; arc_off and
 * row_off give each lattice's slice; scores are float32 label scores theta[V]
 * gathered per arc (the arc-score gather) plus the optional per-arc weight. */
int orc_forward_backward_batch(int B, const int32_t *n_rows, const int64_t *arc_off,
                               const int32_t *src, const int32_t *label, const int32_t *dst,
                               const float *arc_w, const float *theta, int n_threads,
                               double *logZ, double *posterior /* total arcs or NULL */) {
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
  for (int b = 0; b < B; ++b) {
    int64_t a0 = arc_off[b], A = arc_off[b + 1] - a0;
    double *sc = (double *)malloc(sizeof(double) * (size_t)(A > 0 ? A : 1));
    for (int64_t a = 0; a < A; ++a)
      sc[a] = (double)theta[label[a0 + a]] + (arc_w ? (double)arc_w[a0 + a] : 0.0);
    int rc = orc_forward_backward(n_rows[b], A, src + a0, dst + a0, sc, NULL, NULL,
                                  posterior ? posterior + a0 : NULL, &logZ[b]);
    if (rc != ORC_OK) {
#pragma omp atomic write
      bad = rc;
    }
    free(sc);
  }
  return bad;
}

/* ------------------------------------------------------------------------- */
/* Viterbi in float32 max-plus (value = score + best(dst); ties keep the arc
 * met first, i.e. the smallest label).  path receives the labels of the best
 * path from state 0 to the sink (bos .. eos), path_arcs the arc indices. */
int orc_viterbi(int n_rows, int64_t A, const int32_t *src, const int32_t *label,
                const int32_t *dst, const float *score, float *best, int32_t *path,
                int32_t *path_arcs, int max_len, int32_t *path_len) {
  int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int32_t *indeg = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int32_t *outdeg = (int32_t *)malloc(sizeof(int32_t) * (size_t)n_rows);
  int64_t *row_ptr = (int64_t *)malloc(sizeof(int64_t) * ((size_t)n_rows + 1));
  float *v = (float *)malloc(sizeof(float) * (size_t)n_rows);
  int64_t *bp = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_rows);
  int rc = ORC_OK;
  int n = topo_order(n_rows, A, src, dst, order, indeg, outdeg, row_ptr);
  if (n < 0) { rc = n; goto done; }
  {
    int sinks = 0, sink = -1;
    for (int i = 0; i < n; ++i) if (outdeg[order[i]] == 0) { ++sinks; sink = order[i]; }
    if (sinks != 1) { rc = ORC_ERR_SINK; goto done; }
    for (int s = 0; s < n_rows; ++s) { v[s] = -INFINITY; bp[s] = -1; }
    v[sink] = 0.0f;
    for (int i = n - 1; i >= 0; --i) {
      int s = order[i];
      if (s == sink) continue;
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
        if (dst[a] == s) continue;
        float c = score[a] + v[dst[a]];
        if (c > v[s]) { v[s] = c; bp[s] = a; }
      }
    }
    *best = v[0];
    int len = 0, s = 0;
    while (s != sink && bp[s] >= 0) {
      if (len >= max_len) { rc = ORC_ERR_SPACE; goto done; }
      int64_t a = bp[s];
      path[len] = label[a];
      if (path_arcs) path_arcs[len] = (int32_t)a;
      ++len;
      s = dst[a];
    }
    *path_len = len;
  }
done:
  free(order); free(indeg); free(outdeg); free(row_ptr); free(v); free(bp);
  return rc;
}

/* ------------------------------------------------------------------------- */
/* Exact posterior sampling: from state s take arc a with probability
 * exp(score[a] + logbeta[dst] - logbeta[s]); arcs are visited in label order and
 * the first arc whose cumulative probability exceeds u is taken (the last arc
 * with positive probability if rounding leaves u uncovered).  uniforms is
 * [K, max_len]; paths [K, max_len] is filled with `pad` after the path ends;
 * margin[k] = smallest |u - cdf boundary| met on walk k (for tolerance-aware
 * comparison with a float32 implementation); logq[k] = sum of log p. */
int orc_sample_paths(int n_rows, int64_t A, const int32_t *src, const int32_t *label,
                     const int32_t *dst, const double *score, const double *logbeta,
                     int K, int max_len, const double *uniforms, int32_t pad,
                     int32_t *paths, int32_t *path_arcs, int32_t *lengths,
                     double *logq, double *margin) {
  int64_t *row_ptr = (int64_t *)calloc((size_t)n_rows + 1, sizeof(int64_t));
  for (int64_t a = 0; a < A; ++a) row_ptr[src[a] + 1]++;
  for (int s = 0; s < n_rows; ++s) row_ptr[s + 1] += row_ptr[s];
  int rc = ORC_OK;
  for (int k = 0; k < K; ++k) {
    int s = 0, t = 0;
    double lq = 0.0, mg = 1.0;
    for (t = 0; t < max_len; ++t) {
      /* out arcs excluding self loops */
      int64_t chosen = -1, lastpos = -1;
      double cum = 0.0, pch = 0.0;
      double u = uniforms[(size_t)k * max_len + t];
      int any = 0;
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) {
        if (dst[a] == s) continue;
        any = 1;
        double p = exp(score[a] + logbeta[dst[a]] - logbeta[s]);
        if (!(p > 0.0)) continue;
        cum += p;
        lastpos = a;
        if (chosen < 0) {
          if (u < cum) { chosen = a; pch = p; }
          double d1 = fabs(u - cum);
          if (d1 < mg) mg = d1;
        }
      }
      if (!any) break; /* at the sink */
      if (chosen < 0) { chosen = lastpos; pch = exp(score[chosen] + logbeta[dst[chosen]] - logbeta[s]); }
      if (chosen < 0) { rc = ORC_ERR_TABLE; goto done; }
      paths[(size_t)k * max_len + t] = label[chosen];
      if (path_arcs) path_arcs[(size_t)k * max_len + t] = (int32_t)chosen;
      lq += log(pch);
      s = dst[chosen];
    }
    if (t == max_len) {
      /* ran out of length budget (samplers.py:299-302) unless we are at the sink */
      int any = 0;
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) if (dst[a] != s) any = 1;
      if (any) { rc = ORC_ERR_SPACE; goto done; }
    }
    lengths[k] = t;
    for (int j = t; j < max_len; ++j) {
      paths[(size_t)k * max_len + j] = pad;
      if (path_arcs) path_arcs[(size_t)k * max_len + j] = -1;
    }
    logq[k] = lq;
    if (margin) margin[k] = mg;
  }
done:
  free(row_ptr);
  return rc;
}

/* Forced walk: follow marks[k, :] from state 0; returns per-walk sum of arc
 * scores (path_score) and the end state; a mark that has no arc gives -inf and
 * end state 0 like the dense gather (transition == 0) does. */
int orc_score_paths(int n_rows, int64_t A, const int32_t *src, const int32_t *label,
                    const int32_t *dst, const double *score, int K, int max_len,
                    const int32_t *marks, double *path_score,
                    int32_t *end_state) {
  int64_t *row_ptr = (int64_t *)calloc((size_t)n_rows + 1, sizeof(int64_t));
  for (int64_t a = 0; a < A; ++a) row_ptr[src[a] + 1]++;
  for (int s = 0; s < n_rows; ++s) row_ptr[s + 1] += row_ptr[s];
  for (int k = 0; k < K; ++k) {
    int s = 0;
    double tot = 0.0;
    for (int t = 0; t < max_len; ++t) {
      int32_t mk = marks[(size_t)k * max_len + t];
      int64_t hit = -1;
      for (int64_t a = row_ptr[s]; a < row_ptr[s + 1]; ++a) if (label[a] == mk) { hit = a; break; }
      if (hit < 0) { tot = -INFINITY; s = 0; break; }
      if (dst[hit] != s) tot += score[hit]; /* the sink's pad self loop scores nothing */
      s = dst[hit];
    }
    path_score[k] = tot;
    end_state[k] = s;
  }
  free(row_ptr);
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* compute_beta_parallel (scorers.py:753-856) restated with its own data
 * structures, float32, probability domain:
 *   edges[dst][src] = label (one label per state pair: the LAST label written
 *   wins, while parents[src] counts every label -- scorers.py:775-776; this is
 *   the parallel-arc quirk and it is reproduced on purpose);
 *   frontier = states not yet visited whose remaining-parents count is zero;
 *   per frontier iteration the message tanh(Wh bhat + Wx e(label) + b) and the
 *   compatibility exp(W . msg) are recomputed for EVERY [S,S] cell (the
 *   reference's einsums over [B,S,S,H], scorers.py:823-838), then masked.
 * emb [V,H], Wx [H,H], Wh [H,H], W [H], bias [H]; transition [S,V] int64.
 * beta [S] float32 out.  Returns the number of frontier iterations. */
int orc_beta_dense_frontier(const int64_t *transition, int S, int V, int H,
                            const float *emb, const float *Wx, const float *Wh,
                            const float *W, const float *bias, float *beta, int n_threads) {
  if (n_threads < 1) n_threads = 1;  /* rows of the [S,S] cell table in parallel, like torch's intra-op threads */
  int32_t *edges = (int32_t *)calloc((size_t)S * S, sizeof(int32_t));
  float *parents = (float *)calloc((size_t)S, sizeof(float));
  uint8_t *visited = (uint8_t *)calloc((size_t)S, 1);
  uint8_t *cur = (uint8_t *)calloc((size_t)S, 1);
  float *bhat = (float *)calloc((size_t)S * H, sizeof(float));
  float *pmsg = (float *)calloc((size_t)S * S, sizeof(float));            /* [src-side row][col] */
  float *pmsg_emb = (float *)calloc((size_t)S * S * H, sizeof(float));
  float *xw = (float *)malloc(sizeof(float) * (size_t)V * H);             /* Wx e(l) per label */
  float *hw = (float *)malloc(sizeof(float) * (size_t)S * H);             /* Wh bhat(s) */
  for (int i = 0; i < S; ++i)
    for (int j = 0; j < V; ++j) {
      int64_t to = transition[(size_t)i * V + j];
      if (to != 0 && to != i) { parents[i] += 1.0f; edges[(size_t)to * S + i] = j; }
    }
  for (int l = 0; l < V; ++l)
    for (int e = 0; e < H; ++e) {
      float acc = 0.0f;
      for (int d = 0; d < H; ++d) acc += emb[(size_t)l * H + d] * Wx[(size_t)e * H + d];
      xw[(size_t)l * H + e] = acc;
    }
  for (int s = 0; s < S; ++s) beta[s] = 0.0f;
  int init = 1, iters = 0;
  for (;;) {
    int any = 0;
    for (int s = 0; s < S; ++s) { cur[s] = (!visited[s] && parents[s] == 0.0f); any |= cur[s]; }
    if (!any) break;
    ++iters;
    if (init) {
      for (int s = 0; s < S; ++s) if (cur[s]) beta[s] = 1.0f;
      init = 0;
    } else {
      /* beta[cur] = sum over the transposed message row; bhat = sum q * msg_emb */
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 8)
      for (int s = 0; s < S; ++s) {
        if (!cur[s]) continue;
        float b = 0.0f;
        for (int p = 0; p < S; ++p) b += pmsg[(size_t)p * S + s];
        beta[s] = b;
        for (int e = 0; e < H; ++e) {
          float acc = 0.0f;
          for (int p = 0; p < S; ++p)
            acc += pmsg_emb[((size_t)p * S + s) * H + e] * (pmsg[(size_t)p * S + s] / b);
          bhat[(size_t)s * H + e] = acc;
        }
      }
    }
    /* messages for every cell of the rows in the frontier */
    for (int s = 0; s < S; ++s)
      for (int e = 0; e < H; ++e) {
        float acc = 0.0f;
        for (int d = 0; d < H; ++d) acc += bhat[(size_t)s * H + d] * Wh[(size_t)e * H + d];
        hw[(size_t)s * H + e] = acc;
      }
#pragma omp parallel for num_threads(n_threads) schedule(static)
    for (int s = 0; s < S; ++s) {
      float msg[512];  /* H <= 512 (checked by the wrapper) */
      /* the reference evaluates all S x S cells each iteration; the cost model
       * is kept (the loop below touches every cell of every row) but only the
       * frontier rows are stored, exactly what scorers.py:828-843 keeps. */
      for (int c = 0; c < S; ++c) {
        int32_t lab = edges[(size_t)s * S + c];
        float comp = 0.0f;
        for (int e = 0; e < H; ++e) {
          msg[e] = tanhf(hw[(size_t)s * H + e] + xw[(size_t)lab * H + e] + bias[e]);
          comp += msg[e] * W[e];
        }
        if (!cur[s]) continue;
        if (lab != 0) {
          pmsg[(size_t)s * S + c] = expf(comp) * beta[s];
          for (int e = 0; e < H; ++e) pmsg_emb[((size_t)s * S + c) * H + e] = msg[e];
        } else {
          pmsg[(size_t)s * S + c] = 0.0f;
          for (int e = 0; e < H; ++e) pmsg_emb[((size_t)s * S + c) * H + e] = 0.0f;
        }
      }
    }
    /* release children: parents -= number of frontier rows with an edge to them */
    for (int s = 0; s < S; ++s) {
      if (!cur[s]) continue;
      visited[s] = 1;
      for (int c = 0; c < S; ++c) if (edges[(size_t)s * S + c] != 0) parents[c] -= 1.0f;
    }
  }
  free(edges); free(parents); free(visited); free(cur); free(bhat); free(pmsg);
  free(pmsg_emb); free(xw); free(hw);
  return iters;
}
