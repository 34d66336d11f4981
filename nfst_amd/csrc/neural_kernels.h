// neural_kernels.h -- neuralised beta: the backward sweep whose arc weights depend on an
// H-dimensional summary of the destination state.
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
//
// Reference: FSAGRUScorer.compute_beta_per_sample, /root/reference/src/modules/scorers.py:692-751
// (compute_beta_parallel, 753-856, is the same recurrence minus its parallel-arc quirk):
//     t(arc)      = tanh(x[label] + Wh . beta_hat(dst)),   x[l] = Wx . e(l) + bias
//     msg(arc)    = exp(W . t(arc)) * beta(dst)
//     beta(s)     = sum msg,    beta_hat(s) = sum (msg / beta(s)) t(arc)
//     beta(sink)  = 1,          beta_hat(sink) = 0
//
// One workgroup (16 waves) per lattice walks the beta tile program tile by
// tile; the groups of a tile are independent, so every tile is
//   A  one wave per group, lanes over the H components.  Records go D at a time: all operand rows
//      (label table, u and beta_hat rows -- L2 hits) are requested first, tanh and W . t run per
//      record on all lanes, and what is one number per record (exp, times beta of the operand,
//      common exponent) runs once per batch with one record per lane.  The group's sum is a
//      (mantissa, exponent) pair plus an H-vector scaled by the same exponent: no exp/log of beta.
//   B  u(s) = Wh . beta_hat(s) for the states the tile finished, 16 of them against one pass over
//      Wh, as float32 MFMA (16x16x4): the per-state H x H product is what dominates at H = 256,
//      and Wh (256 KiB there) comes from L2 once per pass.  beta_hat rows go from phase A to
//      phase B through LDS.
// The last wave takes no part in A: it stages the next tile's words (an HBM miss) meanwhile.
// A unit-label record (carry of a continuation piece, or the scratch row of a partial group)
// contributes the row's own (beta, beta_hat): beta_hat is a beta-weighted mean, so pieces merge
// by weight.
#pragma once

#ifdef NFST_NEU_STAMPS
// (profiling build only: per-wave busy time of phase A, and where a tile's time goes, for workgroup 0; 100 MHz ticks)
__device__ unsigned long long neu_stamps[128];
#define NEU_STAMP_ADD(i, v) do { if (b == 0 && lane == 0) atomicAdd(&neu_stamps[i], (unsigned long long)(v)); } while (0)
#define NEU_NOW() wall_clock64()
#else
#define NEU_STAMP_ADD(i, v) do {} while (0)
#define NEU_NOW() 0ull
#endif
constexpr int kNeuThreads = 1024, kNeuWaves = kNeuThreads / 64, kNeuRows = 32, kNeuMaxHid = 512;
constexpr int kNeuStageWords = 64 + 256 + 256 + 64 + 4;  // ctl | rec | slot -> arc | leaders | their count

// Which group of a tile a wave takes first.  The packer lists a level's states by falling degree, so group 0 is the
// largest; wave w sits on SIMD w % 4 and SIMD 3 also hosts the (light) staging wave.  With group i on wave i SIMD 0 gets
// ranks 0, 4, 8, 12 -- the heaviest of every four -- and its waves end phase A last (in-kernel stamps at H = 256: 10.0 us
// for wave 0 against 5.6 for wave 3).  This order deals the ranks to the SIMDs like a snake, the largest group to SIMD 3:
// SIMD 3 <- 0, 7, 8; SIMD 0 <- 1, 6, 9, 14; SIMD 1 <- 2, 5, 10, 13; SIMD 2 <- 3, 4, 11, 12.  Later rounds (tiles with more
// than 15 groups): group 15 r + w.
__device__ __forceinline__ int neu_first_group(int wv) { return (int)((0xfcde8ba974560321ull >> (4 * wv)) & 15u); }
// Tiles with more than 15 groups (about half of them on the BASELINE batch: 14.6 groups on average): groups 15 .. 30 go
// to the staging wave first -- it is idle for most of the phase -- then to the waves in rising order of what they already
// hold: 15 + k on wave {15, 12, 13, 14, 11, 8, 9, 10, 7, 4, 5, 6, 3, 0, 1, 2}[k] (with group 15 r + w on wave w, wave 0 had
// two groups in every such tile and ended phase A 1.7 us after the wave with the largest group).  Groups from 31 on:
// 31 + 15 (round - 2) + w on waves 0 .. 14.
__device__ __forceinline__ int neu_group_of(int wv, int rnd) {
  if (rnd == 0) return wv < 15 ? neu_first_group(wv) : (1 << 20);
  if (rnd == 1) return 15 + (int)((0x032147658ba9cfedull >> (4 * wv)) & 15u);
  return wv < 15 ? 31 + (rnd - 2) * 15 + wv : (1 << 20);
}

// The (mantissa, exponent) rows come first in LDS, 8 bytes each: an odd row count would leave everything behind them
// -- the rows phase B reads 16 bytes at a time -- 8 bytes off, and a ds_read_b128 off its alignment is replayed at 64
// cycles per wave instruction (the BASELINE batch has 2221 rows: phase B took 8 us per tile instead of 4)
__host__ __device__ inline int neu_rows_al(int rows) { return (rows + 1) & ~1; }
struct NeuLds {
  int rows, hid;
  __host__ __device__ NeuLds(int r, int h) : rows(r), hid(h) {}
  // multiple of 16 (the K loop of phase B) + 4 floats so that the 16 rows of a pass start in different banks
  __host__ __device__ static int row_stride(int h) { return ((h + 15) & ~15) + 4; }
  // float2 beta[rows] | two staged tiles | float bh[32][row_stride]
  __host__ __device__ int64_t bytes0() const { return (int64_t)neu_rows_al(rows) * 8 + 2 * kNeuStageWords * 4 + (int64_t)kNeuRows * row_stride(hid) * 4; }
  // three bfloat16 planes (high / middle / low part of the 16 rows of a pass of phase B: neu_phase_b), rows of hid + 8 elements
  __host__ __device__ static int64_t planes_bytes(int h) { return 3 * 16 * (int64_t)(h + 8) * 2; }
  __host__ __device__ int64_t bytes() const { return ((bytes0() + 15) & ~(int64_t)15) + planes_bytes(hid); }
};

// tanh(x) = 1 - 2 / (e^2x + 1) on the hardware exp2 and reciprocal: absolute error below 2e-7 over
// the whole line (the messages are bounded by 1 and enter sums, so absolute error is what counts)
__device__ __forceinline__ float neu_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

__device__ __forceinline__ int wave_max_i(int v) {
  v = max(v, dpp_i<0xB1>(v));
  v = max(v, dpp_i<0x4E>(v));
  v = max(v, dpp_i<0x141>(v));
  v = max(v, dpp_i<0x140>(v));
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32),
            d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}

// 2^-d for d >= 0 (0 when the term is too small to matter)
__device__ __forceinline__ float neu_scale(int d) { return d > 120 ? 0.0f : __int_as_float((127 - d) << 23); }

// One wave stages a tile of a program for the phases that follow: control words, records as 32-bit
// words (8 x operand state | label << 16), slot -> canonical arc map, list of group leaders.  Its
// loads miss to HBM, and a wave's loads complete in order, so they must not sit in front of a
// computing wave's L2 hits: the last wave of the workgroup stages tile T+1 while the others run
// phase A of tile T.
// (arc_dst given: the destination state of every slot's arc as well, 256 more words -- the gradient kernel's groups need
// the state their arcs enter before anything else, and that load would otherwise head every group's chain)
__device__ __forceinline__ void neu_stage_tile(const uint32_t *prog, const int32_t *perm, int F, int T, uint32_t *st, int lane,
                                               const int32_t *arc_dst = nullptr) {
  const int U = fmt_u(F), ST = fmt_words(F);
  uint32_t *ctl = st, *rec = st + 64;
  int *cas = (int *)(st + 320), *lead = (int *)(st + 576), *nlead = (int *)(st + 640);
  uint32_t c;
  if (F == 8) {
    const uint4 x = *reinterpret_cast<const uint4 *>(prog + (size_t)T * ST + lane * 4);
    c = x.x;
    const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
    for (int j = 0; j < 4; ++j) rec[lane * 4 + j] = ((r[j] & 0x1fffu) << 3) | (((r[j] >> 13) & 0x7ffu) << 16);
  } else {
    c = prog[(size_t)T * ST + lane];
    for (int j = 0; j < U; ++j) rec[lane + 64 * j] = prog[(size_t)T * ST + 64 + lane + 64 * j];
  }
  for (int j = 0; j < U; ++j) {
    const int ca = perm[(size_t)T * 64 * U + lane + 64 * j];
    cas[lane + 64 * j] = ca;
    if (arc_dst) ((int *)(st + kNeuStageWords))[lane + 64 * j] = ca >= 0 ? arc_dst[ca] : -1;
  }
  ctl[lane] = c;
  const uint64_t leaders = __builtin_amdgcn_ballot_w64((c >> 31) != 0);
  if (c >> 31) lead[__builtin_popcountll(leaders & ((1ull << lane) - 1))] = lane;
  if (lane == 0) nlead[0] = __builtin_popcountll(leaders);
}

// a row another wave of this workgroup wrote earlier in the kernel: read past the CU's vector L1
// (scratch rows are rewritten level after level, a cached line may be stale)
__device__ __forceinline__ float neu_load_fresh(const float *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// a = hi + mid + lo with bfloat16 parts rounded to nearest (the upper 16 bits of a float32 each): |a - (hi + mid + lo)| <= 2^-27 |a|,
// and the products the split MFMAs drop (mid lo, lo mid, lo lo) are ~2^-26 of the result with random signs (truncated parts
// leave them at 2^-23, all of one sign: 3.9e-5 on log beta at H = 512 against 3e-5 asked)
__device__ __forceinline__ uint32_t neu_bf16_rn(float x) {
  const uint32_t b = __float_as_uint(x);
  return (b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u;
}
__device__ __forceinline__ void neu_split3(float a, uint32_t &hi, uint32_t &mid, uint32_t &lo) {  // results in the upper halves
  hi = neu_bf16_rn(a);
  const float r1 = a - __uint_as_float(hi);
  mid = neu_bf16_rn(r1);
  lo = neu_bf16_rn(r1 - __uint_as_float(mid));
}
// Phase B of a tile: out(s) = M . row(s) for the states the tile finished, 16 of them per pass over
// the matrix: D[16 states x 16 columns] += A[16 x 4] B[4 x 16] on the matrix cores in float32
// (v_mfma_f32_16x16x4_f32), one block of 16 columns per wave.  A comes from the LDS rows phase A
// filled (rows_s; groups beyond kNeuRows are fetched back through row_of), B straight from the
// matrix (L2), row-major [out, in].  put(state, column, value) stores a result.
template <class RowOf, class Put>
__device__ __forceinline__ void neu_phase_b(float *rows_s, int hs, int n_lead, const float *__restrict__ wh, int hid,
                                            const uint32_t *ctl_s, const int *lead_s, int tid, int wv, int lane,
                                            RowOf row_of, Put put, const uint4 *whb = nullptr, uint16_t *a3 = nullptr) {
  // ---- B: u = Wh . beta_hat for the states this tile wrote, 16 of them per pass over Wh:
  // D[16 states x 16 columns] += A[16 x 4] B[4 x 16] on the matrix cores in float32, one block of
  // 16 columns per wave.  A comes from the LDS rows phase A filled, B straight from Wh (L2).
  for (int g0 = 0; g0 < n_lead; g0 += 16) {
    const int n = min(16, n_lead - g0);
    const float *rows = rows_s + (size_t)g0 * hs;
    if (g0 + 16 > kNeuRows) {  // more groups in the tile than LDS rows: fetch theirs from the workspace
      __syncthreads();
      for (int i = tid; i < n * hid; i += kNeuThreads) {
        const int g = i / hid, h = i - g * hid;
        const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
        rows_s[g * hs + h] = neu_load_fresh(row_of(sid) + h);
      }
      __syncthreads();
      rows = rows_s;
    }
    const int li = lane & 15, kq = lane >> 4;
    if (whb) {
      // Both factors as three bfloat16 parts (high, middle, low, rounded to nearest: neu_split3) and six
      // v_mfma_f32_16x16x32_bf16 per 32 of K -- hh, hm, mh, hl, lh, mm; the dropped products are 2^-24 of the result -- instead
      // of eight float32 MFMAs: 96 instead of 256 matrix-pipe cycles (round 2 found phase B bound by the float32 MFMA rate:
      // 3.4 us of its ~7 us per tile at H = 256 with four waves per SIMD).  The matrix was split by k_pack_mfma_b3; the 16
      // rows of this pass are split here, once for all sixteen waves, into three planes in LDS.
      typedef float f4 __attribute__((ext_vector_type(4)));
      typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
      const int ps = hid + 8;  // plane row in elements: 16-byte aligned, and the 16 rows of a fragment read start 4 banks apart
      __syncthreads();         // (the planes of the previous pass have been read)
      for (int i = tid; i < 16 * hid; i += kNeuThreads) {
        const int g = i / hid, h = i - g * hid;
        const float a = g < n ? rows[g * hs + h] : 0.0f;
        uint32_t hb, mb, lb;
        neu_split3(a, hb, mb, lb);
        a3[g * ps + h] = (uint16_t)(hb >> 16);
        a3[(16 + g) * ps + h] = (uint16_t)(mb >> 16);
        a3[(32 + g) * ps + h] = (uint16_t)(lb >> 16);
      }
      __syncthreads();
      const int n_steps = hid >> 5;
      for (int ct = wv; ct * 16 < hid; ct += kNeuWaves) {
        const int col = ct * 16 + li;
        f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        const uint4 *bq = whb + (size_t)ct * n_steps * 3 * 64 + lane;
        const uint16_t *ap = a3 + li * ps + 8 * kq;
#pragma unroll 1  // (the gradient kernel is at its 128 registers: a second step in flight spills)
        for (int st = 0; st < n_steps; ++st) {
          const uint4 b0 = bq[(st * 3 + 0) * 64], b1 = bq[(st * 3 + 1) * 64], b2 = bq[(st * 3 + 2) * 64];
          const uint4 a0 = *reinterpret_cast<const uint4 *>(ap + 32 * st), a1 = *reinterpret_cast<const uint4 *>(ap + 16 * ps + 32 * st),
                      a2 = *reinterpret_cast<const uint4 *>(ap + 32 * ps + 32 * st);
#define NFST_MM(A, B) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc, 0, 0, 0)
          NFST_MM(a2, b0); NFST_MM(a0, b2); NFST_MM(a1, b1); NFST_MM(a1, b0); NFST_MM(a0, b1); NFST_MM(a0, b0);  // small terms first
#undef NFST_MM
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // lane holds D[4 kq + r][li]
          const int g = 4 * kq + r;
          if (g < n && col < hid) {
            const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
            put(sid, col, acc[r]);
          }
        }
      }
      continue;
    }
    for (int ct = wv; ct * 16 < hid; ct += kNeuWaves) {
      const int col = ct * 16 + li;
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
      if ((hid & 15) == 0) {
        const float *bp = wh + (size_t)col * hid + 4 * kq;
        const float *ap = rows + li * hs + 4 * kq;
        // B fragments from the matrix itself (16 rows x 64 bytes per load instruction: hidden sizes that are not a multiple of
        // 64, and the float32 reference path of the A/B switch `neu_bf16`; multiples of 64 take the split path above)
        const float4 *b4 = reinterpret_cast<const float4 *>(bp);
        constexpr int sd = 4, sh = 16;  // float4 strides per d and per 64 of K
        int h = 0;
        for (; h + 64 <= hid; h += 64) {  // four 16-byte loads of Wh in flight per lane
          float4 bq[4], aq[4];
#pragma unroll
          for (int d = 0; d < 4; ++d) bq[d] = b4[(h >> 6) * sh + d * sd];
#pragma unroll
          for (int d = 0; d < 4; ++d) aq[d] = *reinterpret_cast<const float4 *>(ap + h + 16 * d);
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d].x, bq[d].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d].y, bq[d].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d].z, bq[d].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[d].w, bq[d].w, acc, 0, 0, 0);
          }
        }
        for (; h < hid; h += 16) {
          const float4 bq = *reinterpret_cast<const float4 *>(bp + h);
          const float4 aq = *reinterpret_cast<const float4 *>(ap + h);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, bq.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, bq.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, bq.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, bq.w, acc, 0, 0, 0);
        }
      } else {
        for (int h = 0; h < hid; h += 4) {  // the LDS rows are zero beyond hid
          const float bv = (col < hid && h + kq < hid) ? wh[(size_t)col * hid + h + kq] : 0.0f;
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(rows[li * hs + h + kq], bv, acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {  // lane holds D[4 kq + r][li]
        const int g = 4 * kq + r;
        if (g < n && col < hid) {
          const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
          put(sid, col, acc[r]);
        }
      }
    }
  }
}

// The matrix of phase B split in three bfloat16 parts, in the fragment order of v_mfma_f32_16x16x32_bf16 (hid a multiple of 64):
// 16-byte word number (((ct * hid/32 + s) * 3 + part) * 64 + lane) holds part `part` of M[16 ct + (lane & 15)][32 s + 8 (lane >> 4) + 0 .. 7].
// One thread per (ct, s, lane); 384 KiB at H = 256, once per launch.
__global__ __launch_bounds__(256) void k_pack_mfma_b3(const float *__restrict__ mat, int hid, uint4 *__restrict__ out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= (hid >> 4) * (hid >> 5) * 64) return;
  const int lane = t & 63, cs = t >> 6, n_steps = hid >> 5;
  const int st = cs % n_steps, ct = cs / n_steps;
  const float *p = mat + (size_t)(ct * 16 + (lane & 15)) * hid + 32 * st + 8 * (lane >> 4);
  uint32_t part[3][4];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    uint32_t hb, mb, lb;
    neu_split3(p[j], hb, mb, lb);
    const uint32_t v[3] = {hb >> 16, mb >> 16, lb >> 16};
#pragma unroll
    for (int q = 0; q < 3; ++q) part[q][j >> 1] = (j & 1) ? (part[q][j >> 1] | (v[q] << 16)) : v[q];
  }
#pragma unroll
  for (int q = 0; q < 3; ++q) out[((size_t)cs * 3 + q) * 64 + lane] = make_uint4(part[q][0], part[q][1], part[q][2], part[q][3]);
}
// where the packed copy sits in a workspace of `used` floats (16-byte aligned)
__host__ __device__ inline int64_t neu_pack_off(int64_t used) { return (used + 3) & ~(int64_t)3; }

template <int HC>  // components per lane: hid <= 64 * HC
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural(nfst_batch lat, const float *__restrict__ label_x,
                                                                 const float *__restrict__ wh,
                                                                 const float *__restrict__ wvec, int hid,
                                                                 float *__restrict__ log_beta,
                                                                 float *beta_hat, float *ws, int wh_packed) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Meta m = load_meta(lat.meta, b);
  const float *wh_ws = ws + neu_pack_off(2 * (int64_t)lat.n_lattices * lat.max_rows * (hid + 1));
  const uint4 *whb = wh_packed ? reinterpret_cast<const uint4 *>(wh_ws) : nullptr;  // Wh as three bfloat16 parts, fragment order
  uint16_t *a3 = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lds) + ((NeuLds(lat.max_rows, hid).bytes0() + 15) & ~(int64_t)15));
  float2 *bme = lds;
  uint32_t *stage_s = (uint32_t *)(bme + neu_rows_al(lat.max_rows));  // two tiles: kNeuStageWords each
  float *bh_s = (float *)(stage_s + 2 * kNeuStageWords);
  // beta_hat of a real row lives in the output array, that of a scratch row (partial groups of the
  // tile program) in the workspace; u = Wh . beta_hat of the real rows in the workspace
  float *bh_w = ws + (size_t)b * lat.max_rows * hid;
  float *u_w = ws + ((size_t)lat.n_lattices + b) * lat.max_rows * hid;
  float *bh_out = beta_hat + (size_t)m.row_off * hid;
  auto bh_row = [&](int r) { return r < m.n_rows ? bh_out + (size_t)r * hid : bh_w + (size_t)r * hid; };
  const int F = m.bwd_u, U = fmt_u(F);
  const uint32_t *prog = lat.bwd_stream + m.bwd_off;
  const int32_t *perm = lat.bwd_perm + m.bwd_slot_off;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int V = lat.vocab;

  for (int i = tid; i < lat.max_rows; i += kNeuThreads) bme[i] = make_float2(0.0f, __int_as_float(kEZero));
  for (int i = tid; i < hid; i += kNeuThreads) {
    bh_out[(size_t)m.sink * hid + i] = 0.0f;
    u_w[(size_t)m.sink * hid + i] = 0.0f;
  }
  float wl[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) wl[c] = (c * 64 + lane < hid) ? wvec[c * 64 + lane] : 0.0f;
  __syncthreads();
  if (tid == 0) bme[m.sink] = make_float2(0.5f, __int_as_float(1));  // beta(sink) = 1
  __threadfence_block();
  __syncthreads();

  const int hs = NeuLds::row_stride(hid);  // LDS row of one state's beta_hat: zero beyond hid
  for (int i = tid; i < kNeuRows * hs; i += kNeuThreads) bh_s[i] = 0.0f;
  __syncthreads();

  auto stage_tile = [&](int T, uint32_t *st) { neu_stage_tile(prog, perm, F, T, st, lane); };
  if (wv == kNeuWaves - 1 && m.bwd_tiles > 0) stage_tile(0, stage_s);
  __syncthreads();

#ifdef NFST_NEU_STAMPS
  unsigned long long st_own = 0, st_bar1 = 0, st_b = 0, st_fence = 0, st_bar2 = 0, st_tiles = 0;
#endif
  for (int T = 0; T < m.bwd_tiles; ++T) {
    uint32_t *st = stage_s + (T & 1) * kNeuStageWords;
    const uint32_t *ctl_s = st, *rec_s = st + 64;
    const int *cas_s = (const int *)(st + 320), *lead_s = (const int *)(st + 576), *nlead_s = (const int *)(st + 640);
    const unsigned long long t_tile = NEU_NOW();
    (void)t_tile;
    if (wv == kNeuWaves - 1 && T + 1 < m.bwd_tiles) stage_tile(T + 1, stage_s + ((T + 1) & 1) * kNeuStageWords);
    const int n_lead = nlead_s[0];

    // ---- A: one wave per group.  The operands of the group's next record are in flight while
    // the current one is computed (they come from L2: label table, u and beta_hat rows).
    for (int rnd = (wv == kNeuWaves - 1), i = neu_group_of(wv, rnd); i < n_lead; ++rnd, i = neu_group_of(wv, rnd)) {
      const int l0 = __builtin_amdgcn_readfirstlane(lead_s[i]);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(ctl_s[l0]);
      const int sid = (int)((c0 & 0xffffu) >> 3), n_rec = (1 << ((c0 >> 20) & 7u)) * U;
      float macc = 0.0f, tacc[HC];
      int eacc = kEZero;
#pragma unroll
      for (int c = 0; c < HC; ++c) tacc[c] = 0.0f;
      for (int q0 = 0; q0 < n_rec; q0 += 64) {
        // lanes look at one record each: which of them carry anything
        uint32_t rc_l = 0;
        int ca_l = -1;
        if (q0 + lane < n_rec) { rc_l = rec_s[l0 * U + q0 + lane]; ca_l = cas_s[l0 * U + q0 + lane]; }
        uint64_t todo = __builtin_amdgcn_ballot_w64(ca_l >= 0 || (int)(rc_l >> 16) == V + 1);
        const float x_l = arc_w ? arc_w[max(ca_l, 0)] : 0.0f;  // table weight of this lane's record
        const int other_l = (int)((rc_l & 0xffffu) >> 3);
        // Records are taken D at a time.  Their operands (L2 hits: label table, u and beta_hat
        // rows) are all requested before the first is used -- one after the other each would cost
        // a round trip.  The H-wide part (tanh, W . t) runs per record on all lanes; everything
        // that is one number per record (exp, times beta of the operand, common exponent) runs
        // once per batch with one record per lane.
        struct Ops { float a[HC], b[HC]; int p, ca; };
        constexpr int D = HC <= 2 ? 6 : (HC <= 4 ? 4 : 2);
        // (straight-line on purpose: a conditional load makes the compiler wait for every load in
        // flight where the branches join; a lane beyond hid re-reads the last component, a record
        // without u reads one it ignores)
        auto issue = [&](int p) {
          Ops o;
          o.p = p;
          const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)rc_l, p);
          o.ca = __builtin_amdgcn_readlane(ca_l, p);
          const int other = (int)((rc & 0xffffu) >> 3), lab = (int)(rc >> 16);
          const float *pa = o.ca >= 0 ? label_x + (size_t)lab * hid : bh_row(other);
          const float *pb = u_w + (size_t)other * hid;
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const int h = min(c * 64 + lane, hid - 1);
            o.a[c] = pa[h];
            o.b[c] = pb[h];
          }
          return o;
        };
        while (todo) {
          Ops buf[D];
          uint64_t batch = 0;
          int nb = 0, p = 0;
#pragma unroll
          for (int k = 0; k < D; ++k) {  // past the last record: the last one again, ignored below
            p = todo ? __builtin_ctzll(todo) : p;
            nb += todo ? 1 : 0;
            batch |= todo & (0 - todo);
            todo &= todo - (todo ? 1 : 0);
            buf[k] = issue(p);
          }
          float vec[D][HC];
          float sc_l = 0.0f;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            if (k < nb) {
              if (buf[k].ca >= 0) {
                float part = 0.0f;
#pragma unroll
                for (int c = 0; c < HC; ++c) {
                  const float t = neu_tanh(buf[k].a[c] + buf[k].b[c]);
                  vec[k][c] = (c * 64 + lane < hid) ? t : 0.0f;
                  part = fmaf(wl[c], vec[k][c], part);
                }
                const float sc = wave_sum(part);
                sc_l = (lane == buf[k].p) ? sc : sc_l;
              } else {  // what the operand row holds: own earlier pieces, or a partial group's scratch row
#pragma unroll
                for (int c = 0; c < HC; ++c) vec[k][c] = (c * 64 + lane < hid) ? buf[k].a[c] : 0.0f;
              }
            } else {
#pragma unroll
              for (int c = 0; c < HC; ++c) vec[k][c] = 0.0f;
            }
          }
          // one record per lane
          const bool act = (batch >> lane) & 1;
          const float2 bo = bme[act ? other_l : 0];
          const ME w = exp_split(sc_l + x_l);
          float wm = act ? bo.x * (ca_l >= 0 ? w.m : 1.0f) : 0.0f;
          int we = max(__float_as_int(bo.y) + (ca_l >= 0 ? w.e : 0), kEZero);
          if (!(wm > 0.0f)) we = kEZero;
          const int en = max(eacc, wave_max_i(we));
          const float so = neu_scale(en - eacc);
          const float q_l = wm * neu_scale(en - we);
          macc = fmaf(macc, so, wave_sum(q_l));
          eacc = en;
#pragma unroll
          for (int c = 0; c < HC; ++c) tacc[c] *= so;
#pragma unroll
          for (int k = 0; k < D; ++k) {
            const float qk = k < nb ? read_lane_f(q_l, buf[k].p) : 0.0f;
#pragma unroll
            for (int c = 0; c < HC; ++c) tacc[c] = fmaf(qk, vec[k][c], tacc[c]);
          }
        }
      }
      const float inv = macc > 0.0f ? 1.0f / macc : 0.0f;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        const int h = c * 64 + lane;
        if (h < hid) {
          bh_row(sid)[h] = tacc[c] * inv;
          if (i < kNeuRows) bh_s[i * hs + h] = tacc[c] * inv;
        }
      }
      if (lane == 0) bme[sid] = me_pack(macc, eacc);
    }
    const unsigned long long t_own = NEU_NOW();
    __syncthreads();
    const unsigned long long t_a = NEU_NOW();
#ifdef NFST_NEU_STAMPS
    st_own += t_own - t_tile; st_bar1 += t_a - t_own;
#endif
    (void)t_own;

    // ---- B: u = Wh . beta_hat for the states this tile wrote (real rows only)
    neu_phase_b(bh_s, hs, n_lead, wh, hid, ctl_s, lead_s, tid, wv, lane,
                [&](int sid) { return (const float *)bh_row(sid); },
                [&](int sid, int col, float v) { if (sid < m.n_rows) u_w[(size_t)sid * hid + col] = v; }, whb, a3);
    const unsigned long long t_b = NEU_NOW();
    __threadfence_block();
    const unsigned long long t_f = NEU_NOW();
    __syncthreads();
#ifdef NFST_NEU_STAMPS
    st_b += t_b - t_a; st_fence += t_f - t_b; st_bar2 += NEU_NOW() - t_f; st_tiles += 1;
#endif
    (void)t_b; (void)t_f; (void)t_a;
  }
#ifdef NFST_NEU_STAMPS
  // per wave of workgroup 0: own phase A | wait at its barrier | phase B | fence | wait at the tile's last barrier | tiles
  NEU_STAMP_ADD(wv * 8 + 0, st_own); NEU_STAMP_ADD(wv * 8 + 1, st_bar1); NEU_STAMP_ADD(wv * 8 + 2, st_b);
  NEU_STAMP_ADD(wv * 8 + 3, st_fence); NEU_STAMP_ADD(wv * 8 + 4, st_bar2); NEU_STAMP_ADD(wv * 8 + 5, st_tiles);
#endif

  // ---- outputs: log beta; beta_hat is in place (rows the program never wrote: -inf, 0)
  float2 *bme_w = reinterpret_cast<float2 *>(ws + 2 * (size_t)lat.n_lattices * lat.max_rows * hid) + (size_t)b * lat.max_rows;
  for (int r = tid; r < m.n_rows; r += kNeuThreads) {
    log_beta[m.row_off + r] = me_log32(bme[r]);
    bme_w[r] = bme[r];  // (mantissa, exponent) pairs for nfst_backward_neural_grad
  }
  for (int r = wv; r < m.n_rows; r += kNeuWaves)
    if (!(bme[r].x > 0.0f))
      for (int h = lane; h < hid; h += 64) bh_out[(size_t)r * hid + h] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Small hidden sizes (hid <= LPR <= 32; round 2).  With lanes over the H components only, a wave of the kernel above
// works on one record at a time and 64 - H of its lanes idle; a tile then costs its longest group's records / D
// round trips to L2 plus the barriers of the second phase, whatever H is: 8.3 us per tile = 1.15 ms on the BASELINE
// batch at H = 8.  Here a wave holds 64 / LPR records side by side, LPR lanes each (the records of a group D x 64 / LPR
// at a time, their operand rows requested together), every record slot keeps its own running (mantissa, exponent,
// vector) sum -- no cross-lane step inside the loop but the W . t sum within a record's LPR lanes, on DPP -- and the
// slots are merged once per group.  u = Wh . beta_hat of the finished state is taken by the same wave straight away
// (row h of Wh lives in lane h's registers, beta_hat[j] by v_readlane): no second phase, no LDS rows for it, one
// barrier per tile.
template <int LPR>
__device__ __forceinline__ float neu_group_sum(float v) {  // sum over the LPR lanes of a record slot, in all of them
  v += dpp_f<0xB1>(v);   // lane ^ 1
  v += dpp_f<0x4E>(v);   // lane ^ 2
  if (LPR >= 8) v += dpp_f<0x141>(v);   // half mirror
  if (LPR >= 16) v += dpp_f<0x140>(v);  // row mirror
  if (LPR >= 32) v += __shfl_xor(v, 16);
  if (LPR >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <int LPR>
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural_small(nfst_batch lat, const float *__restrict__ label_x,
                                                                       const float *__restrict__ wh,
                                                                       const float *__restrict__ wvec, int hid,
                                                                       float *__restrict__ log_beta, float *beta_hat,
                                                                       float *ws) {
  extern __shared__ float2 lds[];
  constexpr int RPB = 64 / LPR;                                   // record slots of a wave
  constexpr int D = LPR <= 16 ? 4 : 6;                            // records per slot whose operands are in flight together
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane / LPR, h = lane % LPR;
  const bool hv = h < hid;
  const int hc = min(h, hid - 1);
  const Meta m = load_meta(lat.meta, b);
  float2 *bme = lds;
  uint32_t *stage_s = (uint32_t *)(bme + neu_rows_al(lat.max_rows));
  float *bh_w = ws + (size_t)b * lat.max_rows * hid;
  float *u_w = ws + ((size_t)lat.n_lattices + b) * lat.max_rows * hid;
  float *bh_out = beta_hat + (size_t)m.row_off * hid;
  auto bh_row = [&](int r) { return r < m.n_rows ? bh_out + (size_t)r * hid : bh_w + (size_t)r * hid; };
  const int F = m.bwd_u, U = fmt_u(F);
  const uint32_t *prog = lat.bwd_stream + m.bwd_off;
  const int32_t *perm = lat.bwd_perm + m.bwd_slot_off;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int V = lat.vocab;

  for (int i = tid; i < lat.max_rows; i += kNeuThreads) bme[i] = make_float2(0.0f, __int_as_float(kEZero));
  for (int i = tid; i < hid; i += kNeuThreads) {
    bh_out[(size_t)m.sink * hid + i] = 0.0f;
    u_w[(size_t)m.sink * hid + i] = 0.0f;
  }
  const float wl = hv ? wvec[h] : 0.0f;
  float whr[LPR];                                                 // row h of Wh ([out, in], row-major)
#pragma unroll
  for (int j = 0; j < LPR; ++j) whr[j] = (hv && j < hid) ? wh[(size_t)h * hid + j] : 0.0f;
  __syncthreads();
  if (tid == 0) bme[m.sink] = make_float2(0.5f, __int_as_float(1));  // beta(sink) = 1
  if (wv == kNeuWaves - 1 && m.bwd_tiles > 0) neu_stage_tile(prog, perm, F, 0, stage_s, lane);
  __threadfence_block();
  __syncthreads();

  for (int T = 0; T < m.bwd_tiles; ++T) {
    uint32_t *st = stage_s + (T & 1) * kNeuStageWords;
    const uint32_t *ctl_s = st, *rec_s = st + 64;
    const int *cas_s = (const int *)(st + 320), *lead_s = (const int *)(st + 576), *nlead_s = (const int *)(st + 640);
    if (wv == kNeuWaves - 1 && T + 1 < m.bwd_tiles)
      neu_stage_tile(prog, perm, F, T + 1, stage_s + ((T + 1) & 1) * kNeuStageWords, lane);
    const int n_lead = nlead_s[0];
    for (int rnd = (wv == kNeuWaves - 1), i = neu_group_of(wv, rnd); i < n_lead; ++rnd, i = neu_group_of(wv, rnd)) {
      const int l0 = __builtin_amdgcn_readfirstlane(lead_s[i]);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(ctl_s[l0]);
      const int sid = (int)((c0 & 0xffffu) >> 3), n_rec = (1 << ((c0 >> 20) & 7u)) * U;
      float macc = 0.0f, tacc = 0.0f;
      int eacc = kEZero;
      for (int q0 = 0; q0 < n_rec; q0 += D * RPB) {
        float a[D], bb[D], x[D];
        float2 bo[D];
        bool real[D], act[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {  // (straight-line: a slot past the group's records reads row 0 and counts for nothing)
          const int q = q0 + k * RPB + sub;
          const bool in = q < n_rec;
          const uint32_t rc = in ? rec_s[l0 * U + q] : 0u;
          const int ca = in ? cas_s[l0 * U + q] : -1;
          const int other = (int)((rc & 0xffffu) >> 3), lab = (int)(rc >> 16);
          real[k] = ca >= 0;
          act[k] = real[k] || lab == V + 1;
          const float *pa = real[k] ? label_x + (size_t)lab * hid : bh_row(other);
          a[k] = pa[hc];
          bb[k] = u_w[(size_t)other * hid + hc];
          x[k] = arc_w ? arc_w[max(ca, 0)] : 0.0f;
          bo[k] = bme[other];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float t = neu_tanh(a[k] + bb[k]);
          const float v = (hv && act[k]) ? (real[k] ? t : a[k]) : 0.0f;  // (what an idle slot read may be anything)
          const float sc = neu_group_sum<LPR>(wl * v);
          const ME w = exp_split_nb(sc + x[k]);
          float wm = act[k] ? bo[k].x * (real[k] ? w.m : 1.0f) : 0.0f;
          int we = max(__float_as_int(bo[k].y) + (real[k] ? w.e : 0), kEZero);
          if (!(wm > 0.0f)) { wm = 0.0f; we = kEZero; }
          const int en = max(eacc, we);
          const float so = neu_scale(en - eacc), q = wm * neu_scale(en - we);
          macc = fmaf(macc, so, q);
          tacc = fmaf(tacc, so, q * v);
          eacc = en;
        }
      }
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {  // merge the record slots
        const int e2 = __shfl_xor(eacc, off);
        const float m2 = __shfl_xor(macc, off), t2 = __shfl_xor(tacc, off);
        const int en = max(eacc, e2);
        const float s1 = neu_scale(en - eacc), s2 = neu_scale(en - e2);
        macc = fmaf(macc, s1, m2 * s2);
        tacc = fmaf(tacc, s1, t2 * s2);
        eacc = en;
      }
      const float inv = macc > 0.0f ? 1.0f / macc : 0.0f;
      const float bh = tacc * inv;
      float u = 0.0f;
#pragma unroll
      for (int j = 0; j < LPR; ++j) u = fmaf(whr[j], read_lane_f(bh, j), u);
      if (sub == 0 && hv) {
        bh_row(sid)[h] = bh;
        if (sid < m.n_rows) u_w[(size_t)sid * hid + h] = u;
      }
      if (lane == 0) bme[sid] = me_pack(macc, eacc);
    }
    __threadfence_block();
    __syncthreads();
  }

  float2 *bme_w = reinterpret_cast<float2 *>(ws + 2 * (size_t)lat.n_lattices * lat.max_rows * hid) + (size_t)b * lat.max_rows;
  for (int r = tid; r < m.n_rows; r += kNeuThreads) {
    log_beta[m.row_off + r] = me_log32(bme[r]);
    bme_w[r] = bme[r];  // (mantissa, exponent) pairs for nfst_backward_neural_grad
  }
  for (int r = wv; r < m.n_rows; r += kNeuWaves)
    if (!(bme[r].x > 0.0f))
      for (int hh = lane; hh < hid; hh += 64) bh_out[(size_t)r * hid + hh] = 0.0f;
}

// ---------------------------------------------------------------------------------------------
// Gradient of the neuralised beta sweep (tune_proposal differentiates log q through compute_beta,
// /root/reference/src/modules/lightning.py:339-406; parameters scorers.py:954-970).
// With z_a = W . t_a (+ arc_w) + log beta(d), p_a = exp(z_a - log beta(s)) and
// lambda(s) = dL/d log beta(s), eta(s) = dL/d beta_hat(s), for an arc a = (s -l-> d):
//     delta_a = p_a (lambda(s) + eta(s) . (t_a - beta_hat(s)))         = dL/dz_a
//     lambda(d) = g(d) + sum_{a into d} delta_a
//     tau_a = delta_a W + p_a eta(s),   rho_a = tau_a (1 - t_a^2)        = dL/d(x[l] + u(d))
//     gamma(d) = sum_{a into d} rho_a,   eta(d) = Wh^T gamma(d) (+ g_beta_hat(d))
//     dL/dx[l] += rho_a,   dL/dW += delta_a t_a,   dL/dWh = sum_d gamma(d) beta_hat(d)^T (host GEMM)
// lambda and gamma of a state are sums over its INCOMING arcs and need lambda, eta of the arcs'
// sources: exactly the alpha tile program (groups by destination, levels by depth), whose carry
// records and scratch rows add partial sums.  t_a is recomputed from u(d), which the forward pass
// left in its workspace together with beta as (mantissa, exponent) pairs -- p_a is a ratio of such
// pairs, no float32 log beta in it.  Same phases as the forward kernel: A one wave per group, lanes
// over H; B eta = Wh^T gamma on float32 MFMA for the states the tile finished.
constexpr int kNeuGradStageWords = kNeuStageWords + 256;  // ... | destination state of every slot's arc
struct NeuGradLds {
  int rows, hid;
  __host__ __device__ NeuGradLds(int r, int h) : rows(r), hid(h) {}
  // float2 beta[rows] | float lambda[rows] | two staged tiles | float gamma[32][row_stride]
  __host__ __device__ int64_t bytes0() const {
    return (int64_t)neu_rows_al(rows) * 8 + (int64_t)((rows + 3) & ~3) * 4 + 2 * kNeuGradStageWords * 4 + (int64_t)kNeuRows * NeuLds::row_stride(hid) * 4 + 16;
  }
  __host__ __device__ int64_t bytes() const { return ((bytes0() + 15) & ~(int64_t)15) + NeuLds::planes_bytes(hid); }
};

template <int HC>
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural_grad(
    nfst_batch lat, const float *__restrict__ label_x, const float *__restrict__ whT, const float *__restrict__ wvec, int hid,
    const float *__restrict__ beta_hat, const float *__restrict__ ws_fwd, const float *__restrict__ g_logbeta,
    const float *__restrict__ g_betahat, float *gamma, float *grad_label_x, float *grad_w, float *ws, int wh_packed) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Meta m = load_meta(lat.meta, b);
  const float *wh_ws = ws + neu_pack_off(2 * (int64_t)lat.n_lattices * lat.max_rows * hid);
  const uint4 *whb = wh_packed ? reinterpret_cast<const uint4 *>(wh_ws) : nullptr;
  uint16_t *a3 = reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(lds) + ((NeuGradLds(lat.max_rows, hid).bytes0() + 15) & ~(int64_t)15));
  float2 *bme = lds;
  float *lam = (float *)(bme + neu_rows_al(lat.max_rows));
  uint32_t *stage_s = (uint32_t *)(lam + ((lat.max_rows + 3) & ~3));
  float *gs = (float *)(stage_s + 2 * kNeuGradStageWords);
  const size_t plane = (size_t)lat.n_lattices * lat.max_rows * hid;
  const float *u_w = ws_fwd + plane + (size_t)b * lat.max_rows * hid;  // u = Wh . beta_hat of the real rows
  const float2 *bme_w = reinterpret_cast<const float2 *>(ws_fwd + 2 * plane) + (size_t)b * lat.max_rows;
  float *gam_w = ws + (size_t)b * lat.max_rows * hid;          // gamma of scratch rows
  float *eta_w = ws + plane + (size_t)b * lat.max_rows * hid;  // eta of real rows
  float *gam_out = gamma + (size_t)m.row_off * hid;
  const float *bh_in = beta_hat + (size_t)m.row_off * hid;
  auto gam_row = [&](int r) { return r < m.n_rows ? gam_out + (size_t)r * hid : gam_w + (size_t)r * hid; };
  const int F = m.fwd_u, U = fmt_u(F);
  const uint32_t *prog = lat.fwd_stream + m.fwd_off;
  const int32_t *perm = lat.fwd_perm + m.fwd_slot_off;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int V = lat.vocab;

  for (int i = tid; i < lat.max_rows; i += kNeuThreads) {
    bme[i] = i < m.n_rows ? bme_w[i] : make_float2(0.0f, __int_as_float(kEZero));
    lam[i] = 0.0f;
  }
  // the start state has no incoming arc: lambda(0) = g(0), eta(0) = g_beta_hat(0)
  for (int i = tid; i < hid; i += kNeuThreads) eta_w[i] = g_betahat ? g_betahat[(size_t)m.row_off * hid + i] : 0.0f;
  float wl[HC], gw[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) {
    wl[c] = (c * 64 + lane < hid) ? wvec[c * 64 + lane] : 0.0f;
    gw[c] = 0.0f;
  }
  const int hs = NeuLds::row_stride(hid);
  for (int i = tid; i < kNeuRows * hs; i += kNeuThreads) gs[i] = 0.0f;
  __syncthreads();
  if (tid == 0) lam[0] = g_logbeta[m.row_off];
  if (wv == kNeuWaves - 1 && m.fwd_tiles > 0) neu_stage_tile(prog, perm, F, 0, stage_s, lane, lat.arc_dst);
  __threadfence_block();
  __syncthreads();

  for (int T = 0; T < m.fwd_tiles; ++T) {
    uint32_t *st = stage_s + (T & 1) * kNeuGradStageWords;
    const uint32_t *ctl_s = st, *rec_s = st + 64;
    const int *cas_s = (const int *)(st + 320), *lead_s = (const int *)(st + 576), *nlead_s = (const int *)(st + 640);
    const int *dst_s = (const int *)(st + kNeuStageWords);
    if (wv == kNeuWaves - 1 && T + 1 < m.fwd_tiles)
      neu_stage_tile(prog, perm, F, T + 1, stage_s + ((T + 1) & 1) * kNeuGradStageWords, lane, lat.arc_dst);
    const int n_lead = nlead_s[0];

    // ---- A: one wave per group (a destination state, or a scratch row holding a partial sum)
    for (int rnd = (wv == kNeuWaves - 1), i = neu_group_of(wv, rnd); i < n_lead; ++rnd, i = neu_group_of(wv, rnd)) {
      const int l0 = __builtin_amdgcn_readfirstlane(lead_s[i]);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(ctl_s[l0]);
      const int sid = (int)((c0 & 0xffffu) >> 3), n_rec = (1 << ((c0 >> 20) & 7u)) * U;
      const bool continuation = (c0 >> 30) & 1u;
      float lacc = 0.0f, gacc[HC], ud[HC];
#pragma unroll
      for (int c = 0; c < HC; ++c) { gacc[c] = 0.0f; ud[c] = 0.0f; }
      float2 bd = make_float2(0.0f, __int_as_float(kEZero));  // beta of the destination state
      bool have_d = false;
      for (int q0 = 0; q0 < n_rec; q0 += 64) {
        uint32_t rc_l = 0;
        int ca_l = -1;
        int dst_l = 0;
        if (q0 + lane < n_rec) { rc_l = rec_s[l0 * U + q0 + lane]; ca_l = cas_s[l0 * U + q0 + lane]; dst_l = dst_s[l0 * U + q0 + lane]; }
        const uint64_t real = __builtin_amdgcn_ballot_w64(ca_l >= 0);
        uint64_t todo = real | __builtin_amdgcn_ballot_w64(ca_l < 0 && (int)(rc_l >> 16) == V + 1);
        if (real && !have_d) {  // every real record of a group enters the same state: the row of u and beta
          const int d = __builtin_amdgcn_readlane(dst_l, __builtin_ctzll(real));  // (staged with the tile)
          bd = bme[d];
#pragma unroll
          for (int c = 0; c < HC; ++c) ud[c] = u_w[(size_t)d * hid + min(c * 64 + lane, hid - 1)];
          have_d = true;
        }
        // one record per lane: source row, its beta and lambda, the table weight
        const int src_l = (int)((rc_l & 0xffffu) >> 3);
        const float x_l = (arc_w && ca_l >= 0) ? arc_w[ca_l] : 0.0f;
        const float2 bs_l = bme[src_l];
        const float lam_l = lam[src_l];
        // Records go D at a time, the operand rows of all of them requested before the first is used (round 2: one
        // record after the other with the next one's rows in flight cost a round trip to L2 per record -- 24 of them
        // for the longest group of a tile).  Straight-line: the row pointers are selected, not the loads (a
        // conditional load makes the compiler wait for every load in flight where the branches join); a partial-sum
        // record reads beta_hat / eta of row 0 and ignores them.
        struct Ops { float a[HC], bh[HC], et[HC]; int p, ca, lab; };
        constexpr int D = HC <= 2 ? 6 : (HC <= 4 ? 4 : 1);
        auto issue = [&](int p) {
          Ops o;
          o.p = p;
          const uint32_t rc = (uint32_t)__builtin_amdgcn_readlane((int)rc_l, p);
          o.ca = __builtin_amdgcn_readlane(ca_l, p);
          const int src = (int)((rc & 0xffffu) >> 3);
          o.lab = (int)(rc >> 16);
          const bool is_arc = o.ca >= 0;
          const int srow = is_arc ? src : 0;
          const float *pa = is_arc ? label_x + (size_t)o.lab * hid : (const float *)gam_row(src);  // (else: gamma of the operand row)
          const float *pb = bh_in + (size_t)srow * hid, *pe = eta_w + (size_t)srow * hid;
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const int h = min(c * 64 + lane, hid - 1);
            o.a[c] = neu_load_fresh(pa + h);
            o.bh[c] = pb[h];
            o.et[c] = neu_load_fresh(pe + h);
          }
          return o;
        };
        while (todo) {
          Ops buf[D];
          int nb = 0, p = 0;
#pragma unroll
          for (int k = 0; k < D; ++k) {  // past the last record: the last one again, ignored below
            p = todo ? __builtin_ctzll(todo) : p;
            nb += todo ? 1 : 0;
            todo &= todo - (todo ? 1 : 0);
            buf[k] = issue(p);
          }
#pragma unroll
          for (int k = 0; k < D; ++k) {
            if (k >= nb) continue;
            const Ops &cur = buf[k];
            if (cur.ca >= 0) {
              float t[HC], p1 = 0.0f, p2 = 0.0f;
#pragma unroll
              for (int c = 0; c < HC; ++c) {
                const bool in = c * 64 + lane < hid;
                t[c] = in ? neu_tanh(cur.a[c] + ud[c]) : 0.0f;
                p1 = fmaf(wl[c], t[c], p1);
                p2 = fmaf(in ? cur.et[c] : 0.0f, t[c] - (in ? cur.bh[c] : 0.0f), p2);
              }
              const float sc = wave_sum(p1), dot = wave_sum(p2);
              // wave-uniform: p = exp(sc + table weight) beta(d) / beta(s)
              const float xs = read_lane_f(x_l, cur.p), lam_s = read_lane_f(lam_l, cur.p);
              const float bsm = read_lane_f(bs_l.x, cur.p);
              const int bse = __builtin_amdgcn_readlane(__float_as_int(bs_l.y), cur.p);
              const ME w = exp_split(sc + xs);
              const float pm = bsm > 0.0f ? w.m * bd.x / bsm : 0.0f;
              const int pe = w.e + __float_as_int(bd.y) - bse;
              const float pa = ldexpf(pm, max(min(pe, 64), -300));
              const float delta = pa * (lam_s + dot);
              lacc += delta;
              float *gx = grad_label_x + (size_t)cur.lab * hid;
#pragma unroll
              for (int c = 0; c < HC; ++c) {
                const int h = c * 64 + lane;
                if (h < hid) {
                  const float tau = fmaf(delta, wl[c], pa * cur.et[c]);
                  const float rho = tau * fmaf(-t[c], t[c], 1.0f);
                  gacc[c] += rho;
                  gw[c] = fmaf(delta, t[c], gw[c]);
                  unsafeAtomicAdd(gx + h, rho);  // (without it: 2.31 / 3.13 / 5.95 ms at H = 8 / 64 / 256 instead of 2.41 / 3.33 / 6.79)
                }
              }
            } else {
              lacc += read_lane_f(lam_l, cur.p);
#pragma unroll
              for (int c = 0; c < HC; ++c) gacc[c] += (c * 64 + lane < hid) ? cur.a[c] : 0.0f;
            }
          }
        }
      }
      if (sid < m.n_rows && !continuation) lacc += g_logbeta[m.row_off + sid];
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        const int h = c * 64 + lane;
        if (h < hid) {
          gam_row(sid)[h] = gacc[c];
          if (i < kNeuRows) gs[i * hs + h] = gacc[c];
        }
      }
      if (lane == 0) lam[sid] = lacc;
    }
    __threadfence_block();
    __syncthreads();

    // ---- B: eta = Wh^T gamma (+ the caller's gradient w.r.t. beta_hat) for the real rows
    neu_phase_b(gs, hs, n_lead, whT, hid, ctl_s, lead_s, tid, wv, lane,
                [&](int sid) { return (const float *)gam_row(sid); },
                [&](int sid, int col, float v) {
                  if (sid < m.n_rows)
                    eta_w[(size_t)sid * hid + col] = v + (g_betahat ? g_betahat[((size_t)m.row_off + sid) * hid + col] : 0.0f);
                }, whb, a3);
    __threadfence_block();
    __syncthreads();
  }
  // dL/dW: one atomic add per wave and component
#pragma unroll
  for (int c = 0; c < HC; ++c)
    if (c * 64 + lane < hid && gw[c] != 0.0f) unsafeAtomicAdd(grad_w + c * 64 + lane, gw[c]);
}

// Small hidden sizes (hid <= LPR <= 32; round 2): the packed layout of k_backward_neural_small for the gradient.  A wave
// holds 64 / LPR records side by side (LPR lanes each), D per slot with all operand rows requested up front; every slot
// keeps its own sums (lambda, gamma, dL/dW), merged once per group; eta = Wh^T gamma of the finished state is taken by
// the same wave (row h of Wh^T in lane h's registers, gamma[j] by v_readlane): no second phase, one barrier per tile.
template <int LPR>
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural_grad_small(
    nfst_batch lat, const float *__restrict__ label_x, const float *__restrict__ whT, const float *__restrict__ wvec, int hid,
    const float *__restrict__ beta_hat, const float *__restrict__ ws_fwd, const float *__restrict__ g_logbeta,
    const float *__restrict__ g_betahat, float *gamma, float *grad_label_x, float *grad_w, float *ws, int gx_in_lds) {
  extern __shared__ float2 lds[];
  constexpr int RPB = 64 / LPR, D = 4;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int sub = lane / LPR, h = lane % LPR;
  const bool hv = h < hid;
  const int hc = min(h, hid - 1);
  const Meta m = load_meta(lat.meta, b);
  float2 *bme = lds;
  float *lam = (float *)(bme + neu_rows_al(lat.max_rows));
  uint32_t *stage_s = (uint32_t *)(lam + ((lat.max_rows + 3) & ~3));
  // dL/dx of this lattice is summed in LDS when [V, hid] floats fit (gx_in_lds) and added to the global table once at
  // the end: at H = 8 the whole batch's atomics would otherwise land on 64 cache lines
  float *gx_s = (float *)(stage_s + 2 * kNeuGradStageWords);
  const size_t plane = (size_t)lat.n_lattices * lat.max_rows * hid;
  const float *u_w = ws_fwd + plane + (size_t)b * lat.max_rows * hid;
  const float2 *bme_w = reinterpret_cast<const float2 *>(ws_fwd + 2 * plane) + (size_t)b * lat.max_rows;
  float *gam_w = ws + (size_t)b * lat.max_rows * hid;
  float *eta_w = ws + plane + (size_t)b * lat.max_rows * hid;
  float *gam_out = gamma + (size_t)m.row_off * hid;
  const float *bh_in = beta_hat + (size_t)m.row_off * hid;
  auto gam_row = [&](int r) { return r < m.n_rows ? gam_out + (size_t)r * hid : gam_w + (size_t)r * hid; };
  const int F = m.fwd_u, U = fmt_u(F);
  const uint32_t *prog = lat.fwd_stream + m.fwd_off;
  const int32_t *perm = lat.fwd_perm + m.fwd_slot_off;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int V = lat.vocab;

  for (int i = tid; i < lat.max_rows; i += kNeuThreads) {
    bme[i] = i < m.n_rows ? bme_w[i] : make_float2(0.0f, __int_as_float(kEZero));
    lam[i] = 0.0f;
  }
  if (gx_in_lds) for (int i = tid; i < V * hid; i += kNeuThreads) gx_s[i] = 0.0f;
  for (int i = tid; i < hid; i += kNeuThreads) eta_w[i] = g_betahat ? g_betahat[(size_t)m.row_off * hid + i] : 0.0f;
  const float wl = hv ? wvec[h] : 0.0f;
  float whr[LPR];  // row h of Wh^T
#pragma unroll
  for (int j = 0; j < LPR; ++j) whr[j] = (hv && j < hid) ? whT[(size_t)h * hid + j] : 0.0f;
  float gw = 0.0f;
  __syncthreads();
  if (tid == 0) lam[0] = g_logbeta[m.row_off];
  if (wv == kNeuWaves - 1 && m.fwd_tiles > 0) neu_stage_tile(prog, perm, F, 0, stage_s, lane, lat.arc_dst);
  __threadfence_block();
  __syncthreads();

  for (int T = 0; T < m.fwd_tiles; ++T) {
    uint32_t *st = stage_s + (T & 1) * kNeuGradStageWords;
    const uint32_t *ctl_s = st, *rec_s = st + 64;
    const int *cas_s = (const int *)(st + 320), *lead_s = (const int *)(st + 576), *nlead_s = (const int *)(st + 640);
    const int *dst_s = (const int *)(st + kNeuStageWords);
    if (wv == kNeuWaves - 1 && T + 1 < m.fwd_tiles)
      neu_stage_tile(prog, perm, F, T + 1, stage_s + ((T + 1) & 1) * kNeuGradStageWords, lane, lat.arc_dst);
    const int n_lead = nlead_s[0];
    for (int rnd = (wv == kNeuWaves - 1), i = neu_group_of(wv, rnd); i < n_lead; ++rnd, i = neu_group_of(wv, rnd)) {
      const int l0 = __builtin_amdgcn_readfirstlane(lead_s[i]);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(ctl_s[l0]);
      const int sid = (int)((c0 & 0xffffu) >> 3), n_rec = (1 << ((c0 >> 20) & 7u)) * U;
      const bool continuation = (c0 >> 30) & 1u;
      // the state the group's arcs enter (all the same): its u row and beta
      int d = -1;
      for (int q0 = 0; q0 < n_rec && d < 0; q0 += 64) {
        const int dl = q0 + lane < n_rec ? dst_s[l0 * U + q0 + lane] : -1;
        const uint64_t real = __builtin_amdgcn_ballot_w64(dl >= 0);
        if (real) d = __builtin_amdgcn_readlane(dl, __builtin_ctzll(real));
      }
      const float ud = d >= 0 ? u_w[(size_t)d * hid + hc] : 0.0f;
      const float2 bd = d >= 0 ? bme[d] : make_float2(0.0f, __int_as_float(kEZero));
      float lacc = 0.0f, gacc = 0.0f;
      for (int q0 = 0; q0 < n_rec; q0 += D * RPB) {
        float a[D], bh[D], et[D], x[D], lm[D];
        float2 bs[D];
        bool isarc[D], act[D];
        int lab[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {  // (straight-line: a slot past the group's records reads row 0 and counts for nothing)
          const int q = q0 + k * RPB + sub;
          const bool in = q < n_rec;
          const uint32_t rc = in ? rec_s[l0 * U + q] : 0u;
          const int ca = in ? cas_s[l0 * U + q] : -1;
          const int src = (int)((rc & 0xffffu) >> 3);
          lab[k] = (int)(rc >> 16);
          isarc[k] = ca >= 0;
          act[k] = isarc[k] || lab[k] == V + 1;
          const int srow = isarc[k] ? src : 0;
          const float *pa = isarc[k] ? label_x + (size_t)lab[k] * hid : (const float *)gam_row(src);
          a[k] = neu_load_fresh(pa + hc);
          bh[k] = bh_in[(size_t)srow * hid + hc];
          et[k] = neu_load_fresh(eta_w + (size_t)srow * hid + hc);
          x[k] = arc_w ? arc_w[max(ca, 0)] : 0.0f;
          bs[k] = bme[src];
          lm[k] = lam[src];
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
          const float t = hv ? neu_tanh(a[k] + ud) : 0.0f;
          const float sc = neu_group_sum<LPR>(wl * t);
          const float dot = neu_group_sum<LPR>(hv ? et[k] * (t - bh[k]) : 0.0f);
          const ME w = exp_split_nb(sc + x[k]);
          const float pm = bs[k].x > 0.0f ? w.m * bd.x / bs[k].x : 0.0f;
          const int pe = w.e + __float_as_int(bd.y) - __float_as_int(bs[k].y);
          const float pa = ldexpf(pm, max(min(pe, 64), -300));
          const float delta = pa * (lm[k] + dot);
          if (isarc[k]) {
            lacc += delta;
            if (hv) {
              const float tau = fmaf(delta, wl, pa * et[k]);
              const float rho = tau * fmaf(-t, t, 1.0f);
              gacc += rho;
              gw = fmaf(delta, t, gw);
              if (gx_in_lds) atomicAdd(gx_s + lab[k] * hid + h, rho);
              else unsafeAtomicAdd(grad_label_x + (size_t)lab[k] * hid + h, rho);
            }
          } else if (act[k]) {  // a partial sum: lambda and gamma of the operand row
            lacc += lm[k];
            gacc += hv ? a[k] : 0.0f;
          }
        }
      }
#pragma unroll
      for (int off = LPR; off < 64; off <<= 1) {  // merge the record slots
        lacc += __shfl_xor(lacc, off);
        gacc += __shfl_xor(gacc, off);
      }
      if (sid < m.n_rows && !continuation) lacc += g_logbeta[m.row_off + sid];
      float eta = 0.0f;
#pragma unroll
      for (int j = 0; j < LPR; ++j) eta = fmaf(whr[j], read_lane_f(gacc, j), eta);
      if (sub == 0 && hv) {
        gam_row(sid)[h] = gacc;
        if (sid < m.n_rows)
          eta_w[(size_t)sid * hid + h] = eta + (g_betahat ? g_betahat[((size_t)m.row_off + sid) * hid + h] : 0.0f);
      }
      if (lane == 0) lam[sid] = lacc;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (gx_in_lds)
    for (int i = tid; i < V * hid; i += kNeuThreads)
      if (gx_s[i] != 0.0f) unsafeAtomicAdd(grad_label_x + i, gx_s[i]);
  // dL/dW: the slots of a wave first, then one atomic add per wave and component
#pragma unroll
  for (int off = LPR; off < 64; off <<= 1) gw += __shfl_xor(gw, off);
  if (sub == 0 && hv && gw != 0.0f) unsafeAtomicAdd(grad_w + h, gw);
}
