// fb_kernels.h -- LDS layout, k_backward and k_forward_backward
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
#pragma once

// Per-arc extras (table weights + caller scores, canonical arc order) -> tile-slot order of the
// forward program, then of the backward program; empty slots and carry records get 0.
__global__ __launch_bounds__(256) void k_slot_extras(nfst_batch lat, nfst_scores sc, int64_t first, int64_t n) {
  const int64_t i = first + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int a = i < lat.fwd_slots ? lat.fwd_perm[i] : lat.bwd_perm[i - lat.fwd_slots];
  float x = 0.0f;
  if (a >= 0) {
    if (lat.weighted) x += lat.arc_w[a];
    if (sc.arc_scores) x += sc.arc_scores[a];
  }
  sc.slot_ws[i] = x;
}

// ------------------------------------------------------------------ LDS layout
// [alpha: rows2 float2][beta: rows2 float2][theta: v2 float2 (V labels + null + unit)]
// [label histogram: v4 float][per sweep: R decoded tiles, kRawSlots raw tiles][4 flag words per sweep]  (16-B aligned)
struct LdsPlan {
  int rows2, v2, v4;
  __host__ __device__ LdsPlan(int max_rows, int vocab)
      : rows2((max_rows + 1) & ~1), v2((vocab + 3) & ~1), v4((vocab + 3) & ~3) {}
  // words of one sweep's rings
  static __host__ __device__ int64_t sweep_words(int R, int RS, bool extra) {
    return (int64_t)R * kSlotWords + (int64_t)RS * (extra ? kRawWordsX : kRawWords);
  }
  __host__ __device__ int64_t fb_bytes(int R, int RS, bool extra) const {
    return ((int64_t)2 * rows2 + v2) * 8 + (int64_t)v4 * 4 + 2 * sweep_words(R, RS, extra) * 4 + 32 + 1024;  // + 8 flag words + 64 x 8 bytes of trash per fused sweep
  }
  __host__ __device__ int64_t bwd_bytes(int R, int RS, bool extra) const {
    return ((int64_t)rows2 + v2) * 8 + sweep_words(R, RS, extra) * 4 + 16;
  }
};

// Block size: wave 0 runs the beta sweep, wave 1 the alpha sweep; every wave helps with
// the initialisation, the row outputs and the posterior pass.  With at most one lattice
// per CU those phases are latency-bound and get 16 waves; with several lattices per CU the
// co-resident workgroups hide each other's latencies and 4 waves are cheaper.

// ------------------------------------------------------------------ backward only
// Wave 0 sweeps the by-source program from the sink, wave 1 decodes for it, wave 2 loads
// for the decoder; every wave helps with the initialisation and the outputs.
template <int NT, bool EXTRA>
__global__ __launch_bounds__(NT) void k_backward(nfst_batch lat, nfst_scores sc, int R, int RS, float *logbeta,
                                                 double *logz64, float *logz32, float2 *beta_me) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  // this thread's label score: requested before anything waits for the meta record
  const float theta_first = tid < lat.vocab ? sc.theta[(size_t)sc.theta_stride * b + tid] : 0.0f;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  float2 *beta = lds;
  float2 *th = lds + plan.rows2;
  uint32_t *ring = (uint32_t *)(th + plan.v2);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t *raw = ring + (size_t)R * kSlotWords;
  // NT = 512: the workgroup has the CU to itself: wave 2 loads for the decoder (deep staging
  // ring); NT = 256: two workgroups per CU, the decoder loads for itself
  // slot-ordered per-arc extras of the backward program (only read by the kernels with EXTRA)
  const int32_t *bwd_extras = (const int32_t *)(sc.slot_ws + lat.fwd_slots + m.bwd_slot_off);
  constexpr bool kSelf = NT != 512;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  if (kSelf) {
    if (wv == 1) self_start<EXTRA, kAhead>(m.bwd_u, lat.bwd_stream + m.bwd_off, bwd_extras, m.bwd_tiles, raw, lane);
  } else if (wv == 2) {
    loader_start<EXTRA>(m.bwd_u, lat.bwd_stream + m.bwd_off, bwd_extras, m.bwd_tiles, raw, RS, lane);
  }
  for (int i = tid; i < m.n_rows; i += NT) beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT, theta_first);
  __syncthreads();
  int *flags = (int *)(ring + LdsPlan::sweep_words(R, RS, EXTRA));
  if (tid == 0) {
    beta[m.sink] = make_float2(0.5f, __int_as_float(1));
    flags[0] = 0; flags[1] = 0; flags[2] = 0; flags[3] = 0;
  }
  __syncthreads();
  if (wv < (kSelf ? 2 : 3))
    run_sweep<EXTRA, kSelf, kAhead>(wv, m.bwd_u, m.bwd_wide != 0, raw, RS, lat.bwd_stream + m.bwd_off, bwd_extras,
              m.bwd_tiles, ring, R, flags, beta, th, ex, lane);
  __syncthreads();
  if (tid == 0) {
    const double z = me_log64(beta[0]);
    if (logz64) logz64[b] = z;
    if (logz32) logz32[b] = (float)z;
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
    if (beta_me) beta_me[m.row_off + i] = beta[i];
  }
}

// ------------------------------------------------------------------ forward-backward
__device__ __forceinline__ float arc_posterior(const float2 av, const float2 bv, const float2 tw, float rz,
                                               int ez, bool has_extra, const Extra &ex, int a) {
  float mw = tw.x;
  int ew = __float_as_int(tw.y);
  if (has_extra) {
    ME x = exp_split(ex.at(a));
    mw *= x.m;
    ew += x.e;
  }
  const float mm = (av.x * mw) * (bv.x * rz);
  const int ee = __float_as_int(av.y) + ew + __float_as_int(bv.y) - ez;
  return ldexpf(mm, max(ee, -300));
}

// Wave 0 runs the beta sweep and wave 1 the alpha sweep, concurrently and without any
// synchronisation between them, fed by waves 2 and 3; after the one barrier that
// follows every wave of the block streams canonical arcs for the posteriors.
// FUSED: waves 0 / 1 load, decode and sweep by themselves (FusedSweep, tile_pipeline.h): no loader,
// no decoder, no rings; every program of the batch is compact and there are no per-arc extras.
template <int NT, bool EXTRA, bool FUSED = false>
__global__ __launch_bounds__(NT) void k_forward_backward(
    nfst_batch lat, nfst_scores sc, int R, int RS, float *__restrict__ logalpha, float *__restrict__ logbeta,
    double *__restrict__ logz64, float *__restrict__ logz32, double *__restrict__ logz_total, int total_slot,
    float *__restrict__ posterior,
    float *__restrict__ grad_theta, float2 *__restrict__ beta_me) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  // this thread's label score: requested before anything waits for the meta record
  const float theta_first = tid < lat.vocab ? sc.theta[(size_t)sc.theta_stride * b + tid] : 0.0f;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  float2 *alpha = lds;
  float2 *beta = lds + plan.rows2;
  float2 *th = lds + 2 * plan.rows2;
  float *gth = (float *)(th + plan.v2);  // [V] label histogram (only if grad_theta)
  uint32_t *ring = (uint32_t *)(gth + plan.v4);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  constexpr bool has_extra = EXTRA;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // even waves work for the beta sweep, odd waves for alpha: waves 0 / 1 sweep, 2 / 3 decode,
  // 6 / 7 load (waves i, i+4, ... share a SIMD: the busy-polling loaders sit with the decoders,
  // the sweep waves share theirs only with waves that sleep at the barrier)
  const bool bwd_side = (wv & 1) == 0;
  const uint32_t *my_prog = bwd_side ? lat.bwd_stream + m.bwd_off : lat.fwd_stream + m.fwd_off;
  // slot-ordered per-arc extras of this sweep (only read by the kernels with EXTRA)
  const int32_t *my_perm = (const int32_t *)(sc.slot_ws + (bwd_side ? lat.fwd_slots + m.bwd_slot_off : m.fwd_slot_off));
  const int my_tiles = bwd_side ? m.bwd_tiles : m.fwd_tiles;
  const int my_u = bwd_side ? m.bwd_u : m.fwd_u;
  const bool my_wide = (bwd_side ? m.bwd_wide : m.fwd_wide) != 0;
  uint32_t *my_ring = bwd_side ? ring : ring + LdsPlan::sweep_words(R, RS, EXTRA);
  uint32_t *my_raw = my_ring + (size_t)R * kSlotWords;
  // NT = 1024: the workgroup has the CU to itself: waves 4 / 5 load for the decoders (deep
  // staging ring); otherwise two workgroups share a CU and the decoders load for themselves
  constexpr bool kSelf = NT != 1024;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  static_assert(!(FUSED && EXTRA), "the fused sweep takes no per-arc extras");
  if (FUSED) {
    if (wv < 2) FusedSweep<false>::start(my_prog, my_tiles, lane);  // (start() does not depend on WIDE)
  } else if (kSelf) {
    if (wv == 2 || wv == 3) self_start<EXTRA, kAhead>(my_u, my_prog, my_perm, my_tiles, my_raw, lane);
  } else if (wv == 6 || wv == 7) {
    loader_start<EXTRA>(my_u, my_prog, my_perm, my_tiles, my_raw, RS, lane);
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    alpha[i] = make_float2(0.0f, __int_as_float(kEZero));
    beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  }
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT, theta_first);
  if (grad_theta) for (int l = tid; l < lat.vocab; l += NT) gth[l] = 0.0f;
  __syncthreads();
  int *flags = (int *)(ring + 2 * LdsPlan::sweep_words(R, RS, EXTRA));
  if (tid == 0) {
    beta[m.sink] = make_float2(0.5f, __int_as_float(1));
    alpha[0] = make_float2(0.5f, __int_as_float(1));
    for (int i = 0; i < 8; ++i) flags[i] = 0;
  }
  __syncthreads();
  const bool want_post = posterior != nullptr || grad_theta != nullptr;
  const int a_begin = m.arc_off, a_end = m.arc_off + m.n_arcs;
  // the posterior pass works on groups of 4 arcs (16-byte loads / stores) over the
  // aligned interior [v_begin, v_end) of the lattice's canonical arc range
  const int v_begin = (a_begin + 3) & ~3, v_end = a_end & ~3;
  // The waves beyond the first four have nothing to do during the sweeps: they fetch
  // their first kPre arc groups into registers now, so that after the sweeps the
  // posterior pass starts on data that is already there.
  constexpr int kSweepThreads = 256;
  constexpr int kHelpers = NT - kSweepThreads;
  constexpr int kPre = (kHelpers > 0) ? 7 : 0;  // 7 x 768 x 4 = 21.5k arcs: a whole BASELINE lattice
  // src | dst << 16 and the label of 4 consecutive canonical arcs: 16 + 8 bytes
  uint4 psd[kPre > 0 ? kPre : 1];
  uint2 plb[kPre > 0 ? kPre : 1];
  if (kPre > 0 && tid >= kSweepThreads && want_post) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
      if (a < v_end) {
        psd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        plb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
      }
    }
  }
  // waves 0 / 1 run the beta / alpha sweeps, waves 2 / 3 decode and waves 6 / 7 load for them
  if constexpr (FUSED) {
    if (wv < 2) {
      const uint32_t trash = lds_addr(flags + 8) + (bwd_side ? 0 : 512) + lane * 8;
      if (my_wide) FusedSweep<true>::run(my_prog, my_tiles, bwd_side ? beta : alpha, th, trash, lane);
      else FusedSweep<false>::run(my_prog, my_tiles, bwd_side ? beta : alpha, th, trash, lane);
    }
  } else if (wv < 4 || (!kSelf && (wv == 6 || wv == 7)))
    run_sweep<EXTRA, kSelf, kAhead>(wv < 4 ? wv >> 1 : 2, my_u, my_wide, my_raw, RS, my_prog, my_perm, my_tiles, my_ring, R,
                     bwd_side ? flags : flags + 4, bwd_side ? beta : alpha, th, ex, lane);
  __syncthreads();
  const float2 zme = beta[0];
  if (tid == 0) {
    const double z = me_log64(zme);
    if (logz64) logz64[b] = z;
    if (logz_total) {
      atomicAdd(&logz_total[total_slot], z);
      if (b == 0) logz_total[(total_slot + 1) % 3] = 0.0;  // the slot of the next launch
    }
    if (logz32) logz32[b] = (float)z;
  }
  const float rz = (zme.x > 0.0f) ? 1.0f / zme.x : 0.0f;
  const int ez = __float_as_int(zme.y);
  auto do_group = [&](const uint4 sd, const uint2 lb, int a) {
    const uint32_t sdv[4] = {sd.x, sd.y, sd.z, sd.w};
    const int ll[4] = {(int)(lb.x & 0xffffu), (int)(lb.x >> 16), (int)(lb.y & 0xffffu), (int)(lb.y >> 16)};
    float pp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s0 = (int)(sdv[q] & 0xffffu), d0 = (int)(sdv[q] >> 16);
      pp[q] = (s0 != d0) ? arc_posterior(alpha[s0], beta[d0], th[ll[q]], rz, ez, has_extra, ex, a + q) : 0.0f;
      if (grad_theta && pp[q] > 0.0f) atomicAdd(&gth[ll[q]], pp[q]);
    }
    if (posterior) *reinterpret_cast<float4 *>(posterior + a) = make_float4(pp[0], pp[1], pp[2], pp[3]);
  };
  if (kPre > 0 && tid >= kSweepThreads && want_post) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
      if (a < v_end) do_group(psd[u], plb[u], a);
    }
  }
  // the row outputs: every thread (the sweep waves start here, the others come when their preloaded
  // arc groups are done)
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logalpha) logalpha[m.row_off + i] = me_log32(alpha[i]);
    if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
    if (beta_me) beta_me[m.row_off + i] = beta[i];
  }
  if (want_post) {
    // the arc groups that were not preloaded: kPB groups per iteration, all loads issued
    // before the first use
    constexpr int kPB = 4;
    for (int a0 = v_begin + 4 * (kPre * kHelpers + tid); a0 < v_end; a0 += NT * 4 * kPB) {
      uint4 sd[kPB];
      uint2 lb[kPB];
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = min(a0 + u * NT * 4, v_end - 4);  // clamped: always a valid group
        sd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        lb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
      }
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = a0 + u * NT * 4;
        if (a >= v_end) break;
        do_group(sd[u], lb[u], a);
      }
    }
    // unaligned head and tail (at most 3 arcs each)
    const int n_head = min(v_begin, a_end) - a_begin;
    const int n_tail = (v_end >= v_begin) ? a_end - v_end : 0;
    if (tid < n_head + n_tail) {
      const int a = tid < n_head ? a_begin + tid : v_end + (tid - n_head);
      const int s0 = lat.arc_src[a], d0 = lat.arc_dst[a], l0 = lat.arc_label[a];
      const float p = (s0 != d0) ? arc_posterior(alpha[s0], beta[d0], th[l0], rz, ez, has_extra, ex, a) : 0.0f;
      if (posterior) posterior[a] = p;
      if (grad_theta && p > 0.0f) atomicAdd(&gth[l0], p);
    }
    if (grad_theta) {
      __syncthreads();
      float *gout = grad_theta + (size_t)b * lat.vocab;
      for (int l = tid; l < lat.vocab; l += NT) gout[l] = gth[l];
    }
  }
}
