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
//   B  u(s) = Wh . beta_hat(s) for the states the tile finished, 16 of them against one pass over
//      Wh, as float32 MFMA (16x16x4) -- the per-state H x H product is what dominates at H = 256,
//      and Wh (256 KiB there) has to come from L2 once per pass.
// A unit-label record (carry of a continuation piece, or the scratch row of a partial group)
// contributes the row's own (beta, beta_hat): beta_hat is a beta-weighted mean, so pieces merge
// by weight.
#pragma once

constexpr int kNeuThreads = 1024, kNeuWaves = kNeuThreads / 64, kNeuRows = 32, kNeuMaxHid = 512;

struct NeuLds {
  int rows, hid;
  __host__ __device__ NeuLds(int r, int h) : rows(r), hid(h) {}
  // multiple of 16 (the K loop of phase B) + 4 floats so that the 16 rows of a pass start in different banks
  __host__ __device__ static int row_stride(int h) { return ((h + 15) & ~15) + 4; }
  // float2 beta[rows] | u32 ctl[64] | u32 rec[256] | i32 cas[256] | i32 lead[64] | i32 n_lead[4] | float bh[32][row_stride]
  __host__ __device__ int64_t bytes() const { return (int64_t)rows * 8 + (64 + 256 + 256 + 64 + 4) * 4 + (int64_t)kNeuRows * row_stride(hid) * 4; }
};

// 2^-d for d >= 0 (0 when the term is too small to matter)
__device__ __forceinline__ float neu_scale(int d) { return d > 120 ? 0.0f : __int_as_float((127 - d) << 23); }

template <int HC>  // components per lane: hid <= 64 * HC
__global__ __launch_bounds__(kNeuThreads) void k_backward_neural(nfst_batch lat, const float *__restrict__ label_x,
                                                                 const float *__restrict__ wh,
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

  const int hs = NeuLds::row_stride(hid);  // LDS row of one state's beta_hat: zero beyond hid
  for (int i = tid; i < kNeuRows * hs; i += kNeuThreads) bh_s[i] = 0.0f;
  __syncthreads();

  // the words of tile T+1 are fetched while tile T is computed
  struct Fetch { uint4 x; uint32_t w; };
  auto fetch = [&](int T) {
    Fetch f;
    f.x = make_uint4(0, 0, 0, 0);
    f.w = 0;
    if (T >= m.bwd_tiles) return f;
    if (tid < 64) {
      if (F == 8) f.x = *reinterpret_cast<const uint4 *>(prog + (size_t)T * ST + tid * 4);
      else f.w = prog[(size_t)T * ST + tid];
    } else if (tid < 64 + 64 * U) {
      if (F != 8) f.w = prog[(size_t)T * ST + tid];
    } else if (tid >= 512 && tid < 512 + 64 * U) {
      f.w = (uint32_t)perm[(size_t)T * 64 * U + (tid - 512)];
    }
    return f;
  };
  Fetch nx = fetch(0);

  for (int T = 0; T < m.bwd_tiles; ++T) {
    // ---- stage the tile's words: control words, records as 32-bit words, slot -> arc map
    if (tid < 64) {
      uint32_t c = nx.w;
      if (F == 8) {
        const uint4 x = nx.x;
        c = x.x;
        const uint32_t r[4] = {x.y, __builtin_amdgcn_alignbit(x.z, x.y, 24), __builtin_amdgcn_alignbit(x.w, x.z, 16), x.w >> 8};
#pragma unroll
        for (int j = 0; j < 4; ++j) rec_s[tid * 4 + j] = ((r[j] & 0x1fffu) << 3) | (((r[j] >> 13) & 0x7ffu) << 16);
      }
      ctl_s[tid] = c;
      const uint64_t leaders = __builtin_amdgcn_ballot_w64((c >> 31) != 0);
      if (c >> 31) lead_s[__builtin_popcountll(leaders & ((1ull << tid) - 1))] = tid;
      if (tid == 0) nlead_s[0] = __builtin_popcountll(leaders);
    } else if (tid < 64 + 64 * U) {
      if (F != 8) rec_s[tid - 64] = nx.w;
    } else if (tid >= 512 && tid < 512 + 64 * U) {
      cas_s[tid - 512] = (int)nx.w;
    }
    __syncthreads();
    nx = fetch(T + 1);
    const int n_lead = nlead_s[0];

    // ---- A: one wave per group.  The operands of the group's next record are in flight while
    // the current one is computed (they come from L2: label table, u and beta_hat rows).
    for (int i = wv; i < n_lead; i += kNeuWaves) {
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
        struct Ops { float a[HC], b[HC]; uint32_t rc; int ca; };
        auto issue = [&](int p) {
          Ops o;
          o.rc = (uint32_t)__builtin_amdgcn_readlane((int)rc_l, p);
          o.ca = __builtin_amdgcn_readlane(ca_l, p);
          const int other = (int)((o.rc & 0xffffu) >> 3), lab = (int)(o.rc >> 16);
          const float *pa = o.ca >= 0 ? label_x + (size_t)lab * hid : bh_w + (size_t)other * hid;
          const float *pb = u_w + (size_t)other * hid;
#pragma unroll
          for (int c = 0; c < HC; ++c) {
            const int h = c * 64 + lane;
            o.a[c] = (h < hid) ? pa[h] : 0.0f;
            o.b[c] = (h < hid && o.ca >= 0) ? pb[h] : 0.0f;
          }
          return o;
        };
        Ops cur;
        if (todo) cur = issue(__builtin_ctzll(todo));
        while (todo) {
          todo &= todo - 1;
          Ops nxt;
          if (todo) nxt = issue(__builtin_ctzll(todo));
          const int other = (int)((cur.rc & 0xffffu) >> 3);
          const float2 bo = bme[other];
          float vec[HC];
          float wm = bo.x;
          int we = __float_as_int(bo.y);
          if (cur.ca >= 0) {
            float part = 0.0f;
#pragma unroll
            for (int c = 0; c < HC; ++c) {
              vec[c] = (c * 64 + lane < hid) ? tanhf(cur.a[c] + cur.b[c]) : 0.0f;
              part = fmaf(wl[c], vec[c], part);
            }
            float score = wave_sum(part);
            if (arc_w) score += arc_w[cur.ca];
            const ME w = exp_split(score);
            wm *= w.m;
            we += w.e;
          } else {  // what row `other` holds: own earlier pieces, or a partial group's scratch row
#pragma unroll
            for (int c = 0; c < HC; ++c) vec[c] = cur.a[c];
          }
          if (wm > 0.0f) {
            we = max(we, kEZero);
            if (we > eacc) {
              const float sc = neu_scale(we - eacc);
              macc *= sc;
#pragma unroll
              for (int c = 0; c < HC; ++c) tacc[c] *= sc;
              eacc = we;
            }
            wm *= neu_scale(eacc - we);
            macc += wm;
#pragma unroll
            for (int c = 0; c < HC; ++c) tacc[c] = fmaf(wm, vec[c], tacc[c]);
          }
          if (todo) cur = nxt;
        }
      }
      const float inv = macc > 0.0f ? 1.0f / macc : 0.0f;
#pragma unroll
      for (int c = 0; c < HC; ++c) {
        const int h = c * 64 + lane;
        if (h < hid) {
          bh_w[(size_t)sid * hid + h] = tacc[c] * inv;
          if (i < kNeuRows) bh_s[i * hs + h] = tacc[c] * inv;
        }
      }
      if (lane == 0) bme[sid] = me_pack(macc, eacc);
    }
    __syncthreads();

    // ---- B: u = Wh . beta_hat for the states this tile wrote, 16 of them per pass over Wh:
    // D[16 states x 16 columns] += A[16 x 4] B[4 x 16] on the matrix cores in float32, one block of
    // 16 columns per wave.  A comes from the LDS rows phase A filled, B straight from Wh (L2).
    for (int g0 = 0; g0 < n_lead; g0 += 16) {
      const int n = min(16, n_lead - g0);
      const float *rows = bh_s + (size_t)g0 * hs;
      if (g0 + 16 > kNeuRows) {  // more groups in the tile than LDS rows: fetch theirs from the workspace
        __syncthreads();
        for (int i = tid; i < n * hid; i += kNeuThreads) {
          const int g = i / hid, h = i - g * hid;
          const int sid = (int)((ctl_s[lead_s[g0 + g]] & 0xffffu) >> 3);
          bh_s[g * hs + h] = bh_w[(size_t)sid * hid + h];
        }
        __syncthreads();
        rows = bh_s;
      }
      const int li = lane & 15, kq = lane >> 4;
      for (int ct = wv; ct * 16 < hid; ct += kNeuWaves) {
        const int col = ct * 16 + li;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if ((hid & 15) == 0) {
          const float *bp = wh + (size_t)col * hid + 4 * kq;
          const float *ap = rows + li * hs + 4 * kq;
          int h = 0;
          for (; h + 64 <= hid; h += 64) {  // four 16-byte loads of Wh in flight per lane
            float4 bq[4], aq[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) bq[d] = *reinterpret_cast<const float4 *>(bp + h + 16 * d);
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
            if (sid < m.n_rows) u_w[(size_t)sid * hid + col] = acc[r];
          }
        }
      }
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
