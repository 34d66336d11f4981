// semiring.h -- (mantissa, exponent) arithmetic, lattice meta data, per-arc extras
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
#pragma once

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

// the same numbers without a branch (the range test is a select): for code that keeps loads in flight
__device__ __forceinline__ ME exp_split_nb(float x) {
  const bool live = x > -9.0e7f;
  const float xc = fminf(live ? x : 0.0f, 9.0e7f);
  const float kf = rintf(xc * 1.44269504088896341f);
  float t = fmaf(-kf, 0.693145751953125f, xc);
  t = fmaf(-kf, 1.42860682030941723e-6f, t);
  float p = 1.0f / 5040.0f;
  p = fmaf(p, t, 1.0f / 720.0f);
  p = fmaf(p, t, 1.0f / 120.0f);
  p = fmaf(p, t, 1.0f / 24.0f);
  p = fmaf(p, t, 1.0f / 6.0f);
  p = fmaf(p, t, 0.5f);
  p = fmaf(p, t, 1.0f);
  p = fmaf(p, t, 1.0f);
  ME r;
  r.m = live ? p : 0.0f;
  r.e = live ? (int)kf : kEZero;
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

// natural log of an (m, e) pair.  float64: log(m) in float32 (m in [0.5, 1): |log m| <= 0.7, so one
// float32 ulp is 6e-8 absolute -- below what the float32 mantissa arithmetic of the sweeps leaves)
// plus e * ln 2 in float64; no double-precision log on the kernel's tail.  float32: two fused
// multiply-adds with ln 2 split in two (the product with the high part is exact for |e| < 2^11).
__device__ __forceinline__ double me_log64(float2 v) {
  if (!(v.x > 0.0f)) return -__builtin_huge_val();
  return (double)logf(v.x) + (double)__float_as_int(v.y) * 0.693147180559945309417232;
}
__device__ __forceinline__ float me_log32(float2 v) {
  if (!(v.x > 0.0f)) return kNegInf;
  const float e = (float)__float_as_int(v.y);
  return fmaf(e, 0.693145751953125f, fmaf(e, 1.42860682030941723e-6f, logf(v.x)));
}

// ---- float64 mantissas: the precise flavour of the sweeps (deep programs) ------------
// A float32 label weight carries a relative rounding error of ~3e-8 that is the SAME at every use of the
// label: along a path of L arcs over few distinct labels the errors add up linearly (L eps / sqrt(V)), not
// as a random walk -- 1.5e-5 on a 900-level chain with 24 labels (tests/golden/fuzz_deep_chain.npz).  Programs
// deeper than kPreciseTiles tiles therefore keep alpha, beta and every arc weight as (float64 mantissa,
// int32 exponent): same semiring, same tile programs, 16 bytes per value in LDS.  v_fma_f64 / v_mul_f64 issue
// at the float32 rate on gfx950, so the chain is as long as before; what grows is the LDS footprint.
struct Rec64 {  // one value in LDS (16 bytes, read and written as one b128)
  double m;
  int e;
  int pad;
};
struct ME64 {
  double m;
  int e;
};
// exp(x) = m * 2^e, m in [0.70, 1.42], relative error ~2e-16; x below -9e7 gives an exact zero
__device__ __forceinline__ ME64 exp_split64(double x) {
  const bool live = x > -9.0e7;
  const double xc = fmin(live ? x : 0.0, 9.0e7);
  const double kf = rint(xc * 1.44269504088896340736);
  double t = fma(-kf, 6.93147180369123816490e-01, xc);  // ln 2, high part (32 bits: the product is short)
  t = fma(-kf, 1.90821492927058770002e-10, t);          // low part
  // exp(t), |t| <= 0.3466: degree-13 Taylor (the first term left out is 4e-18)
  double p = 1.0 / 6227020800.0;
  p = fma(p, t, 1.0 / 479001600.0);
  p = fma(p, t, 1.0 / 39916800.0);
  p = fma(p, t, 1.0 / 3628800.0);
  p = fma(p, t, 1.0 / 362880.0);
  p = fma(p, t, 1.0 / 40320.0);
  p = fma(p, t, 1.0 / 5040.0);
  p = fma(p, t, 1.0 / 720.0);
  p = fma(p, t, 1.0 / 120.0);
  p = fma(p, t, 1.0 / 24.0);
  p = fma(p, t, 1.0 / 6.0);
  p = fma(p, t, 0.5);
  p = fma(p, t, 1.0);
  p = fma(p, t, 1.0);
  ME64 r;
  r.m = live ? p : 0.0;
  r.e = live ? (int)kf : kEZero;
  return r;
}
// (mantissa in [0.5, 1), exponent) of a sum M 2^E; an exact zero keeps the exponent that never wins a maximum
__device__ __forceinline__ Rec64 me_pack64(double M, int E) {
  Rec64 r;
  r.m = __builtin_amdgcn_frexp_mant(M);
  r.e = (M != 0.0) ? max(E + __builtin_amdgcn_frexp_exp(M), kEZero) : kEZero;
  r.pad = 0;
  return r;
}
__device__ __forceinline__ double me_log64(const Rec64 v) {
  if (!(v.m > 0.0)) return -__builtin_huge_val();
  return log(v.m) + (double)v.e * 0.693147180559945309417232;
}
__device__ __forceinline__ float2 me_f2(const Rec64 v) { return make_float2((float)v.m, __int_as_float(v.e)); }
__device__ __forceinline__ float2 me_f2(const float2 v) { return v; }
__device__ __forceinline__ float me_log32(const Rec64 v) { return me_log32(me_f2(v)); }
// value types of the two flavours
template <bool PREC> struct ValOf { typedef float2 T; };
template <> struct ValOf<true> { typedef Rec64 T; };
__device__ __forceinline__ void val_set(float2 &v, float m, int e) { v = make_float2(m, __int_as_float(e)); }
__device__ __forceinline__ void val_set(Rec64 &v, float m, int e) { v.m = (double)m; v.e = e; v.pad = 0; }

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
