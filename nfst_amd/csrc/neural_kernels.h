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
// One workgroup per lattice walks the beta tile program tile by tile; the
// groups of a tile are independent, so every tile is
//   A  one wave per group: lanes hold the H components, the group's records are summed into
//      (mantissa, exponent) + an H-vector scaled by the same exponent -- no exp/log of beta;
//   B  u(s) = Wh . beta_hat(s) for the states the tile finished, all of them against one pass
//      over Wh (columns on threads, up to 8 states per thread in registers, beta_hat broadcast
//      from LDS) -- the per-state H x H product is what dominates at H = 256.
// A unit-label record (carry of a continuation piece, or the scratch row of a partial group)
// contributes the row's own (beta, beta_hat): beta_hat is a beta-weighted mean, so pieces merge
// by weight.
#pragma once

constexpr int kNeuThreads = 1024, kNeuWaves = kNeuThreads / 64, kNeuChunk = 16, kNeuMaxHid = 512;

struct NeuLds {
  int rows, hid;
  __host__ __device__ NeuLds(int r, int h) : rows(r), hid(h) {}
  // float2 beta[rows] | u32 ctl[64] | u32 rec[256] | i32 cas[256] | i32 lead[64] | i32 n_lead[4] | float bh[16 * hid]
  __host__ __device__ int64_t bytes() const { return (int64_t)rows * 8 + (64 + 256 + 256 + 64 + 4) * 4 + (int64_t)kNeuChunk * hid * 4; }
};

// 2^-d for d >= 0 (0 when the term is too small to matter)
__device__ __forceinline__ float neu_scale(int d) { return d > 120 ? 0.0f : __int_as_float((127 - d) << 23); }

template <int HC>  // components per lane: hid <= 64 * HC
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural(nfst_batch lat, const float *__restrict__ label_x,
                                                                 const float *__restrict__ wh_t,
                                                                 const float *__restrict__ wvec, int hid,
                                                                 float *__restrict__ log_beta,
                                                                 float *__restrict__ beta_hat, float *ws) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const Meta m = load_meta(lat.meta, b);
  float2 *bme = lds;
  uint32_t *ctl_s = (uint32_t *)(bme + lat.max_rows);
  uint32_t *rec_s = ctl_s + 64;
  int *cas_s = (int *)(rec_s + 256);
  int *lead_s = cas_s + 256;
  int *nlead_s = lead_s + 64;
  float *bh_s = (float *)(nlead_s + 4);
  // workspace rows of this lattice: beta_hat (also of scratch rows), then u = Wh . beta_hat
  float *bh_w = ws + (size_t)b * lat.max_rows * hid;
  float *u_w = ws + ((size_t)lat.n_lattices + b) * lat.max_rows * hid;
  const int F = m.bwd_u, U = fmt_u(F), ST = fmt_words(F);
  const uint32_t *prog = lat.bwd_stream + m.bwd_off;
  const int32_t *perm = lat.bwd_perm + m.bwd_slot_off;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int V = lat.vocab;

  for (int i = tid; i < lat.max_rows; i += kNeuThreads) bme[i] = make_float2(0.0f, __int_as_float(kEZero));
  for (int i = tid; i < hid; i += kNeuThreads) {
    bh_w[(size_t)m.sink * hid + i] = 0.0f;
    u_w[(size_t)m.sink * hid + i] = 0.0f;
  }
  float wl[HC];
#pragma unroll
  for (int c = 0; c < HC; ++c) wl[c] = (c * 64 + lane < hid) ? wvec[c * 64 + lane] : 0.0f;
  __syncthreads();
  if (tid == 0) bme[m.sink] = make_float2(0.5f, __int_as_float(1));  // beta(sink) = 1
  __threadfence_block();
  __syncthreads();

  // phase-B thread layout: column k of Wh^T, thread row tr; thread rows share the chunk's states
  const int hp = (hid + 63) & ~63, TR = kNeuThreads / hp;
  const int kcol = tid % hp, trow = tid / hp;

  for (int T = 0; T < m.bwd_tiles; ++T) {
    // ---- stage the tile's words: control words, records as 32-bit words, slot -> arc map
    if (tid < 64) {
      if (F == 8) {
        const uint4 x = *reinterpret_cast<const uint4 *>(prog + (size_t)T * ST + tid * 4);
        ctl_s[tid] = x.x;
        const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
        for (int j = 0; j < 4; ++j) rec_s[tid * 4 + j] = ((r[j] & 0x1fffu) << 3) | (((r[j] >> 13) & 0x7ffu) << 16);
      } else {
        ctl_s[tid] = prog[(size_t)T * ST + tid];
      }
      const uint32_t c = ctl_s[tid];
      const uint64_t leaders = __builtin_amdgcn_ballot_w64((c >> 31) != 0);
      if (c >> 31) lead_s[__builtin_popcountll(leaders & ((1ull << tid) - 1))] = tid;
      if (tid == 0) nlead_s[0] = __builtin_popcountll(leaders);
    } else if (tid < 64 + 64 * U) {
      const int q = tid - 64;
      if (F != 8) rec_s[q] = prog[(size_t)T * ST + 64 + q];
    } else if (tid >= 512 && tid < 512 + 64 * U) {
      const int q = tid - 512;
      cas_s[q] = perm[(size_t)T * 64 * U + q];
    }
    __syncthreads();
    const int n_lead = nlead_s[0];

    // ---- A: one wave per group
    for (int i = wv; i < n_lead; i += kNeuWaves) {
      const int l0 = __builtin_amdgcn_readfirstlane(lead_s[i]);
      const uint32_t c0 = __builtin_amdgcn_readfirstlane(ctl_s[l0]);
      const int sid = (int)((c0 & 0xffffu) >> 3), n_rec = (1 << ((c0 >> 20) & 7u)) * U;
      float macc = 0.0f, tacc[HC];
      int eacc = kEZero;
#pragma unroll
      for (int c = 0; c < HC; ++c) tacc[c] = 0.0f;
      for (int r = 0; r < n_rec; ++r) {
        const int q = l0 * U + r;
        const uint32_t rc = __builtin_amdgcn_readfirstlane(rec_s[q]);
        const int ca = __builtin_amdgcn_readfirstlane(cas_s[q]);
        const int other = (int)((rc & 0xffffu) >> 3), lab = (int)(rc >> 16);
        float vec[HC];
        float wm;
        int we;
        if (ca >= 0) {
          float part = 0.0f;
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const int h = c * 64 + lane;
            float t = 0.0f;
            if (h < hid) t = tanhf(label_x[(size_t)lab * hid + h] + u_w[(size_t)other * hid + h]);
            vec[c] = t;
            part = fmaf(wl[c], t, part);
          }
          float score = wave_sum(part);
          if (arc_w) score += arc_w[ca];
          const ME w = exp_split(score);
          const float2 bo = bme[other];
          wm = w.m * bo.x;
          we = w.e + __float_as_int(bo.y);
        } else if (lab == V + 1) {  // what row `other` holds: own earlier pieces, or a partial group's scratch row
          const float2 bo = bme[other];
          wm = bo.x;
          we = __float_as_int(bo.y);
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const int h = c * 64 + lane;
            vec[c] = (h < hid) ? bh_w[(size_t)other * hid + h] : 0.0f;
          }
        } else {
          continue;  // empty slot
        }
        if (!(wm > 0.0f)) continue;
        we = max(we, kEZero);
        if (we > eacc) {
          const float s = neu_scale(we - eacc);
          macc *= s;
#pragma unroll
          for (int c = 0; c < HC; ++c) tacc[c] *= s;
          eacc = we;
        }
        wm *= neu_scale(eacc - we);
        macc += wm;
#pragma unroll
        for (int c = 0; c < HC; ++c) tacc[c] = fmaf(wm, vec[c], tacc[c]);
      }
      const float inv = macc > 0.0f ? 1.0f / macc : 0.0f;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        const int h = c * 64 + lane;
        if (h < hid) bh_w[(size_t)sid * hid + h] = tacc[c] * inv;
      }
      if (lane == 0) bme[sid] = me_pack(macc, eacc);
    }
    __threadfence_block();
    __syncthreads();

    // ---- B: u = Wh . beta_hat for the states this tile wrote (scratch rows need none)
    for (int g0 = 0; g0 < n_lead; g0 += kNeuChunk) {
      const int n = min(kNeuChunk, n_lead - g0);
      for (int i = tid; i < n * hid; i += kNeuThreads) {
        const int g = i / hid, h = i - g * hid;
        const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
        bh_s[g * hid + h] = bh_w[(size_t)sid * hid + h];
      }
      __syncthreads();
      if (kcol < hid && trow < TR) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
        for (int h = 0; h < hid; ++h) {
          const float w = wh_t[(size_t)h * hid + kcol];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int g = trow + j * TR;
            if (g < n) acc[j] = fmaf(w, bh_s[g * hid + h], acc[j]);
          }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int g = trow + j * TR;
          if (g < n) {
            const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
            if (sid < m.n_rows) u_w[(size_t)sid * hid + kcol] = acc[j];
          }
        }
      }
      __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
  }

  // ---- outputs: log beta and beta_hat of the real rows (rows the program never wrote: -inf, 0)
  for (int r = tid; r < m.n_rows; r += kNeuThreads) log_beta[m.row_off + r] = me_log32(bme[r]);
  for (int i = tid; i < m.n_rows * hid; i += kNeuThreads) {
    const int r = i / hid;
    beta_hat[(size_t)m.row_off * hid + i] = bme[r].x > 0.0f ? bh_w[i] : 0.0f;
  }
}
