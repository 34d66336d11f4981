// fb_kernels.h -- LDS layout, k_backward and k_forward_backward
// Part of the single translation unit kernels.hip (device code in an anonymous namespace).
#pragma once

// ------------------------------------------------------------------ LDS layout
// [alpha: rows2 float2][beta: rows2 float2][theta: v2 float2 (V labels + null + unit)]
// [label histogram: v4 float][per sweep: R decoded tiles, RS raw tiles][kSweepFlags flag words per sweep]  (16-B aligned)
struct LdsPlan {
  int rows2, v2, v4;
  __host__ __device__ LdsPlan(int max_rows, int vocab)
      : rows2((max_rows + 1) & ~1), v2((vocab + 3) & ~1), v4((vocab + 3) & ~3) {}
  // the precise flavour (tile waves only: no staging rings): 16-byte values and label weights, ring slots of
  // kSlotWordsP words, 16 bytes of trash per lane and sweep
  __host__ __device__ int64_t fb_fixed_precise() const {
    return ((int64_t)2 * rows2 + v2) * 16 + (int64_t)v4 * 4 + 2 * kSweepFlags * 4 + 2048;
  }
  __host__ __device__ int64_t bwd_fixed_precise() const { return ((int64_t)rows2 + v2) * 16 + kSweepFlags * 4 + 1024; }
  // words of one sweep's rings
  static __host__ __device__ int64_t sweep_words(int R, int RS, bool extra) {
    (void)extra;
    return (int64_t)R * kSlotWords + (int64_t)RS * kRawWords;
  }
  __host__ __device__ int64_t fb_bytes(int R, int RS, bool extra) const {
    return ((int64_t)2 * rows2 + v2) * 8 + (int64_t)v4 * 4 + 2 * sweep_words(R, RS, extra) * 4 + 2 * kSweepFlags * 4 + 1024;  // + flag words + 64 x 8 bytes of trash per fused sweep
  }
  __host__ __device__ int64_t bwd_bytes(int R, int RS, bool extra) const {
    return ((int64_t)rows2 + v2) * 8 + sweep_words(R, RS, extra) * 4 + kSweepFlags * 4;
  }
};

// Block size: wave 0 runs the beta sweep, wave 1 the alpha sweep; every wave helps with
// the initialisation, the row outputs and the posterior pass.  With at most one lattice
// per CU those phases are latency-bound and get 16 waves; with several lattices per CU the
// co-resident workgroups hide each other's latencies and 4 waves are cheaper.

// ------------------------------------------------------------------ backward only
// Wave 0 sweeps the by-source program from the sink, wave 1 decodes for it, wave 2 loads
// for the decoder; with per-arc extras the last four (two) waves are the extras waves; every wave
// helps with the initialisation and the outputs.
// TW (512 threads, all-compact batches): wave 0 sweeps (tile_sweep2), waves 1, 2, 3, 5 are its tile waves.
template <int NT, int EXTRA, bool TW = false, bool PREC = false>  // EXTRA: 0 none, 1 table weights or caller scores, 2 both
__global__ __launch_bounds__(NT) void k_backward(nfst_batch lat, nfst_scores sc, int R, int RS, float *logbeta,
                                                 double *logz64, float *logz32, float2 *beta_me) {
  static_assert(!PREC || TW, "precise flavour: tile waves");
  typedef typename ValOf<PREC>::T VT;
  extern __shared__ float2 lds_raw[];
  VT *lds = reinterpret_cast<VT *>(lds_raw);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (lat.only && lat.only[b] != lat.only_tag) return;  // (the chunked flavour ran this lattice: nfst_batch.only)
  // this thread's label score: requested before anything waits for the meta record
  const float theta_first = tid < lat.vocab ? sc.theta[(size_t)sc.theta_stride * b + tid] : 0.0f;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  VT *beta = lds;
  VT *th = lds + plan.rows2;
  uint32_t *ring = (uint32_t *)(th + plan.v2);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t *raw = ring + (size_t)R * kSlotWords;
  constexpr int kTwSlot = PREC ? kSlotWordsP : kSlotWords2;
  // NT = 512: the workgroup has the CU to itself: wave 2 loads for the decoder (deep staging
  // ring); NT = 256: two workgroups per CU, the decoder loads for itself
  // slot -> canonical arc map of the backward program (only read by the kernels with EXTRA)
  const int32_t *bwd_perm = lat.bwd_perm + m.bwd_slot_off;
  constexpr bool kSelf = NT != 512;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  constexpr int kNE = kSelf ? 2 : 4, kFirstX = kSelf ? 2 : 4;  // extras waves: the last kNE waves of the block
  static_assert(!TW || NT == 512, "tile waves: the one-lattice-per-CU flavour");
  if (TW) {  // (the tile waves start below)
  } else if (kSelf) {
    if (wv == 1) self_start<kAhead>(m.bwd_u, lat.bwd_stream + m.bwd_off, m.bwd_tiles, raw, lane);
  } else if (wv == 2) {
    loader_start(m.bwd_u, lat.bwd_stream + m.bwd_off, m.bwd_tiles, raw, RS, lane);
  }
  // weight / tile waves: the records and maps of their first tiles go in flight now, the first gathers before the barrier
  const bool x_wave = TW ? (wv == 1 || wv == 2 || wv == 3 || wv == 5) : (EXTRA != 0 && wv >= kFirstX);
  const int x_index = TW ? (wv < 4 ? wv - 1 : 3) : wv - kFirstX;
  WeightWave<8, kNE, EXTRA, TW, PREC> xw8;
  if (x_wave && m.bwd_u == 8) xw8.start_maps(lat.bwd_stream + m.bwd_off, bwd_perm, m.bwd_tiles, ex, x_index, lane);
  for (int i = tid; i < m.n_rows; i += NT) val_set(beta[i], 0.0f, kEZero);
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT, theta_first);
  if (x_wave && m.bwd_u == 8) xw8.start_gathers(lane);
  __syncthreads();
  int *flags = (int *)(ring + (TW ? (int64_t)R * kTwSlot : LdsPlan::sweep_words(R, RS, EXTRA)));
  if (tid == 0) {
    val_set(beta[m.sink], 0.5f, 1);
    for (int i = 0; i < kSweepFlags; ++i) flags[i] = 0;
  }
  __syncthreads();
  const bool tw_v2 = TW && m.bwd_wide == 0;
  if (x_wave) {
    run_weights<kNE, EXTRA, TW, PREC>(xw8, m.bwd_u, lat.bwd_stream + m.bwd_off, bwd_perm, m.bwd_tiles, ex, x_index, ring, R, flags,
                                      (const float2 *)th, (const float2 *)beta, tw_v2, lds_addr(flags + kSweepFlags) + lane * (PREC ? 16 : 8), 0u, 0, lane);
  } else if constexpr (TW) {
    if (wv == 0) {
      __builtin_amdgcn_s_setprio(3);
      if constexpr (PREC) tile_sweep2p<kNE>(m.bwd_tiles, ring, R, flags, flags + 4, lane);
      else if (tw_v2) tile_sweep2<4, kNE>(m.bwd_tiles, ring, R, flags, flags + 4, lane);
      else tile_sweep<4, true, kNE>(m.bwd_tiles, ring, R, flags, flags + 4, lane);
      __builtin_amdgcn_s_setprio(0);
    }
  } else if (wv < (kSelf ? 2 : 3))
    run_sweep<EXTRA, kSelf, kAhead, kNE>(wv, m.bwd_u, m.bwd_wide != 0, raw, RS, lat.bwd_stream + m.bwd_off,
              m.bwd_tiles, ring, R, flags, (float2 *)beta, (const float2 *)th, lane);
  __syncthreads();
  if (tid == 0) {
    const double z = me_log64(beta[0]);
    if (logz64) logz64[b] = z;
    if (logz32) logz32[b] = (float)z;
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
    if (beta_me) beta_me[m.row_off + i] = me_f2(beta[i]);
  }
}

// ------------------------------------------------------------------ forward-backward
// exp of a per-arc extra as (mantissa, exponent) on the hardware exp2 after a Cody-Waite reduction (relative error
// ~1e-7: a posterior needs no more; the sweeps' weights come from exp_split).  Below -9e7: weight zero.
__device__ __forceinline__ void exp_me_fast(float extra, float &m, int &e) {
  const float x = fminf(fmaxf(extra, -1.0e8f), 9.0e7f);
  const float kf = rintf(x * 1.44269504088896341f);
  float t = fmaf(-kf, 0.693145751953125f, x);
  t = fmaf(-kf, 1.42860682030941723e-6f, t);
  m = __builtin_amdgcn_exp2f(t * 1.44269504088896341f);
  e = (int)kf;
}
__device__ __forceinline__ float arc_posterior(const float2 av, const float2 bv, const float2 tw, float rz,
                                               int ez, bool has_extra, float mx, int ex) {
  float mw = tw.x;
  int ew = __float_as_int(tw.y);
  if (has_extra) {
    mw *= mx;
    ew += ex;
  }
  const float mm = (av.x * mw) * (bv.x * rz);
  const int ee = __float_as_int(av.y) + ew + __float_as_int(bv.y) - ez;
  return ldexpf(mm, max(ee, -300));
}

// Wave 0 runs the beta sweep and wave 1 the alpha sweep, concurrently and without any
// synchronisation between them, fed by waves 2 and 3; after the one barrier that
// follows every wave of the block streams canonical arcs for the posteriors.
// With per-arc extras the waves from 8 (4 with 512 threads) on are the extras waves, even ones for
// beta, odd ones for alpha (tile_pipeline.h).
// FUSED: waves 0 / 1 load, decode and sweep by themselves (FusedSweep, tile_pipeline.h): no loader,
// no decoder, no rings; every program of the batch is compact and there are no per-arc extras.
// TW (1024 threads): tile waves -- no loader, no staging ring, no decoder: the eight waves 2, 3, 6, 7, 10, 11, 14, 15
// (the SIMDs the sweep waves are not on) decode every fourth tile of their sweep's program straight from
// HBM into the decoded ring, label weights and per-arc extras included (WeightWave<.., FULL>).
template <int NT, int EXTRA, bool FUSED = false, bool TW = false, bool PREC = false>
__global__ __launch_bounds__(NT) void k_forward_backward(
    nfst_batch lat, nfst_scores sc, int R, int RS, float *__restrict__ logalpha, float *__restrict__ logbeta,
    double *__restrict__ logz64, float *__restrict__ logz32, double *__restrict__ logz_total, int total_slot,
    float *__restrict__ posterior,
    float *__restrict__ grad_theta, float2 *__restrict__ beta_me) {
  static_assert(!PREC || (TW && EXTRA != 3), "precise flavour: tile waves, per-arc extras from HBM / L2");
  typedef typename ValOf<PREC>::T VT;
  extern __shared__ float2 lds_raw[];
  VT *lds = reinterpret_cast<VT *>(lds_raw);
  constexpr int kTwSlot = PREC ? kSlotWordsP : kSlotWords2;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (lat.only && lat.only[b] != lat.only_tag) return;  // (the chunked flavour ran this lattice: nfst_batch.only)
  if (tid == 0) NFST_STAMP(0);
  // this thread's label score: requested before anything waits for the meta record
  const float theta_first = tid < lat.vocab ? sc.theta[(size_t)sc.theta_stride * b + tid] : 0.0f;
  const Meta m = load_meta(lat.meta, b);
  const LdsPlan plan(lat.max_rows, lat.vocab);
  VT *alpha = lds;
  VT *beta = lds + plan.rows2;
  VT *th = lds + 2 * plan.rows2;
  float *gth = (float *)(th + plan.v2);  // [V] label histogram (only if grad_theta)
  uint32_t *ring = (uint32_t *)(gth + plan.v4);
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  constexpr bool has_extra = EXTRA != 0;
  // EXTRA = 3 (tile waves only): the lattice's per-arc extras -- table weights + caller scores, summed -- are staged in LDS at
  // kernel entry with coalesced reads (RS carries the room in floats); the tile waves and the posterior pass gather them there
  // instead of from HBM / L2 (the alpha side's random 4-byte gathers kept the CU's vector memory pipeline busy for longer than
  // a tile takes); needs room for eight ring slots per sweep beside the extras: lattices up to ~14k arcs
  constexpr bool kCached = EXTRA == 3;
  static_assert(!kCached || TW, "staged extras: tile-wave kernels");
  static_assert(!TW || (NT == 1024 && !FUSED), "tile waves: the one-lattice-per-CU flavour");
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // even waves work for the beta sweep, odd waves for alpha: waves 0 / 1 sweep, 2 / 3 decode,
  // 6 / 7 load (waves i, i+4, ... share a SIMD: the busy-polling loaders sit with the decoders,
  // the sweep waves share theirs only with waves that sleep at the barrier)
  const bool bwd_side = (wv & 1) == 0;
  const uint32_t *my_prog = bwd_side ? lat.bwd_stream + m.bwd_off : lat.fwd_stream + m.fwd_off;
  // slot -> canonical arc map of this sweep's program (only read by the kernels with EXTRA)
  const int32_t *my_perm = bwd_side ? lat.bwd_perm + m.bwd_slot_off : lat.fwd_perm + m.fwd_slot_off;
  const int my_tiles = bwd_side ? m.bwd_tiles : m.fwd_tiles;
  const int my_u = bwd_side ? m.bwd_u : m.fwd_u;
  const bool my_wide = (bwd_side ? m.bwd_wide : m.fwd_wide) != 0;
  uint32_t *my_ring = bwd_side ? ring : ring + (TW ? (int64_t)R * kTwSlot : LdsPlan::sweep_words(R, RS, EXTRA));
  const bool tw_v2 = TW && !my_wide;  // (the decoded-tile format of tile_sweep2: programs with narrow groups)
  uint32_t *my_raw = my_ring + (size_t)R * kSlotWords;
  // NT = 1024: the workgroup has the CU to itself: waves 4 / 5 load for the decoders (deep
  // staging ring); otherwise two workgroups share a CU and the decoders load for themselves
  constexpr bool kSelf = NT != 1024;
  constexpr int kAhead = kSelf ? kDmaAheadShared : kDmaAheadDeep;
  static_assert(!(FUSED && EXTRA), "the fused sweep takes no per-arc extras");
  static_assert(!(EXTRA && NT < 512), "the extras waves need idle waves: 512 threads at least");
#ifndef NFST_XW_MODE
#define NFST_XW_MODE 0
#endif
  // extras waves per sweep, the first of them (tuning: NFST_XW_MODE 1 = only the waves on the decoders' SIMDs, 2 = only
  // those on the sweep waves' SIMDs)
  constexpr int kNE = (TW || (NT == 1024 && NFST_XW_MODE == 0)) ? 4 : 2, kFirstX = NT == 1024 ? 8 : 4;
#ifndef NFST_TW_PLACE
#define NFST_TW_PLACE 1
#endif
  // tile waves: 1 = waves 2 .. 9 (two per SIMD: one SIMD cannot issue a whole sweep's decoding -- ~100 slots per
  // tile -- at the sweep's pace); 0 = the eight waves of SIMDs 2 and 3
  const bool x_wave = TW ? (NFST_TW_PLACE == 1 ? (wv >= 2 && wv < 10) : (wv & 2) != 0)
                         : EXTRA != 0 && wv >= kFirstX && (NT != 1024 || NFST_XW_MODE == 0 || ((wv >> 1) & 1) == (NFST_XW_MODE == 1 ? 1 : 0));
  const int x_index = TW ? (NFST_TW_PLACE == 1 ? (wv - 2) >> 1 : wv >> 2) : (NT == 1024 && NFST_XW_MODE != 0) ? (wv - kFirstX) >> 2 : (wv - kFirstX) >> 1;
  if (FUSED) {
    if (wv < 2) FusedSweep<false>::start(my_prog, my_tiles, lane);  // (start() does not depend on WIDE)
  } else if (TW) {  // (the tile waves start below)
  } else if (kSelf) {
    if (wv == 2 || wv == 3) self_start<kAhead>(my_u, my_prog, my_tiles, my_raw, lane);
  } else if (wv == 6 || wv == 7) {
    loader_start(my_u, my_prog, my_tiles, my_raw, RS, lane);
  }
  // extras waves: the slot -> arc maps of their first tiles go in flight now, the first gathers before the barrier
  WeightWave<8, kNE, EXTRA, TW, PREC> xw8;
  if (x_wave && my_u == 8) xw8.start_maps(my_prog, my_perm, my_tiles, ex, x_index, lane);
  for (int i = tid; i < m.n_rows; i += NT) {
    val_set(alpha[i], 0.0f, kEZero);
    val_set(beta[i], 0.0f, kEZero);
  }
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT, theta_first);
  if (grad_theta) for (int l = tid; l < lat.vocab; l += NT) gth[l] = 0.0f;
  if (x_wave && my_u == 8) xw8.start_gathers(lane);
  __syncthreads();
  int *flags = (int *)(ring + 2 * (TW ? (int64_t)R * kTwSlot : LdsPlan::sweep_words(R, RS, EXTRA)));
  float *xc = (float *)(flags + 2 * kSweepFlags + 256);  // (behind the flags and the 1 KiB of trash)
  constexpr int kTrash = PREC ? 16 : 8;  // bytes of trash per lane and sweep (the non-leader lanes' stores)
  const int xc_first = m.arc_off & ~3;
  if (kCached) {
    // 16 bytes at a time over the aligned interior, scalar at the ends (nothing is read outside the lattice's arcs)
    const int a_lo = m.arc_off, a_hi = m.arc_off + m.n_arcs;
    for (int a = xc_first + 4 * tid; a < a_hi; a += 4 * NT) {
      float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (a >= a_lo && a + 4 <= a_hi) {
        if (ex.arc_w) x = *reinterpret_cast<const float4 *>(ex.arc_w + a);
        if (ex.arc_scores) {
          const float4 y = *reinterpret_cast<const float4 *>(ex.arc_scores + a);
          x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
      } else {
        float t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int q = 0; q < 4; ++q)
          if (a + q >= a_lo && a + q < a_hi) t[q] = ex.at(a + q);
        x = make_float4(t[0], t[1], t[2], t[3]);
      }
      *reinterpret_cast<float4 *>(xc + (a - xc_first)) = x;
    }
  }
  if (tid == 0) {
    val_set(beta[m.sink], 0.5f, 1);
    val_set(alpha[0], 0.5f, 1);
    for (int i = 0; i < 2 * kSweepFlags; ++i) flags[i] = 0;
  }
  __syncthreads();
  if (tid == 0) NFST_STAMP(1);
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
  // exp of their per-arc extras as (mantissa, exponent), computed while the sweeps run: fetched and exponentiated after the
  // sweeps they doubled the posterior pass (7.5 -> 15 us)
  float4 pxm[kPre > 0 ? kPre : 1];
  int4 pxe[kPre > 0 ? kPre : 1];
  // the per-arc extras of 4 consecutive canonical arcs (16-byte loads: a is a multiple of 4)
  auto extras4 = [&](int a) {
    float4 x = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (kCached) return *reinterpret_cast<const float4 *>(xc + (a - xc_first));
    if (has_extra) {
      if (ex.arc_w) x = *reinterpret_cast<const float4 *>(ex.arc_w + a);
      if (ex.arc_scores) {
        const float4 y = *reinterpret_cast<const float4 *>(ex.arc_scores + a);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
    }
    return x;
  };
  auto preload_arcs = [&]() {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
      if (a < v_end) {
        psd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        plb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
        if (has_extra) pxm[u] = extras4(a);
      }
    }
    if (has_extra) {
#pragma unroll
      for (int u = 0; u < kPre; ++u) {
        const float4 x = pxm[u];
        exp_me_fast(x.x, pxm[u].x, pxe[u].x); exp_me_fast(x.y, pxm[u].y, pxe[u].y);
        exp_me_fast(x.z, pxm[u].z, pxe[u].z); exp_me_fast(x.w, pxm[u].w, pxe[u].w);
      }
    }
  };
  // (in program order behind the sweeps' dispatch: the idle waves get there at once, a tile / weight / loader
  // wave when its tiles are done -- HBM misses in front of its loads would hold them back, a wave's loads
  // complete in order -- and the preloaded registers are not live across the pipeline code)
  // waves 0 / 1 run the beta / alpha sweeps, waves 2 / 3 decode and waves 6 / 7 load for them
  if constexpr (FUSED) {
    if (wv < 2) {
      const uint32_t trash = lds_addr(flags + 2 * kSweepFlags) + (bwd_side ? 0 : 512) + lane * 8;
      if (my_wide) FusedSweep<true>::run(my_prog, my_tiles, (const float2 *)(bwd_side ? beta : alpha), (const float2 *)th, trash, lane);
      else FusedSweep<false>::run(my_prog, my_tiles, (const float2 *)(bwd_side ? beta : alpha), (const float2 *)th, trash, lane);
    }
  } else if (x_wave) {
    run_weights<kNE, EXTRA, TW, PREC>(xw8, my_u, my_prog, my_perm, my_tiles, ex, x_index, my_ring, R, bwd_side ? flags : flags + kSweepFlags,
                                      (const float2 *)th, (const float2 *)(bwd_side ? beta : alpha), tw_v2,
                                      lds_addr(flags + 2 * kSweepFlags) + (bwd_side ? 0 : 64 * kTrash) + lane * kTrash, lds_addr(xc), xc_first, lane);
    if (TW && wv == 2) NFST_STAMP(6);
  } else if constexpr (TW) {
    if (wv < 2) {
      int *fl = bwd_side ? flags : flags + kSweepFlags;
      __builtin_amdgcn_s_setprio(3);  // (the chain: 45.8 -> 44.3 us at 256 lattices)
      // (every program is compact: four slots per lane)
      if constexpr (PREC) tile_sweep2p<kNE>(my_tiles, my_ring, R, fl, fl + 4, lane);
      else if (tw_v2) tile_sweep2<4, kNE>(my_tiles, my_ring, R, fl, fl + 4, lane, bwd_side ? 5 : -1);
      else tile_sweep<4, true, kNE>(my_tiles, my_ring, R, fl, fl + 4, lane);
      __builtin_amdgcn_s_setprio(0);
      NFST_STAMP(bwd_side ? 2 : 3);
    }
  } else if (wv < 4 || (!kSelf && (wv == 6 || wv == 7))) {
    run_sweep<EXTRA, kSelf, kAhead, kNE>(wv < 4 ? wv >> 1 : 2, my_u, my_wide, my_raw, RS, my_prog,
                     my_tiles, my_ring, R, bwd_side ? flags : flags + kSweepFlags, (float2 *)(bwd_side ? beta : alpha), (const float2 *)th, lane);
  }
  if (kPre > 0 && want_post && tid >= kSweepThreads) preload_arcs();
  __syncthreads();
  if (tid == 0) NFST_STAMP(4);
  const float2 zme = me_f2(beta[0]);
  if (tid == 0) {
    const double z = me_log64(beta[0]);
    if (logz64) logz64[b] = z;
    if (logz_total) {
      atomicAdd(&logz_total[total_slot], z);
      if (b == 0) logz_total[(total_slot + 1) % 3] = 0.0;  // the slot of the next launch
    }
    if (logz32) logz32[b] = (float)z;
  }
  const float rz = (zme.x > 0.0f) ? 1.0f / zme.x : 0.0f;
  const int ez = __float_as_int(zme.y);
  auto do_group = [&](const uint4 sd, const uint2 lb, int a, const float4 xm, const int4 xe) {
    const float xmv[4] = {xm.x, xm.y, xm.z, xm.w};
    const int xev[4] = {xe.x, xe.y, xe.z, xe.w};
    const uint32_t sdv[4] = {sd.x, sd.y, sd.z, sd.w};
    const int ll[4] = {(int)(lb.x & 0xffffu), (int)(lb.x >> 16), (int)(lb.y & 0xffffu), (int)(lb.y >> 16)};
    float pp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int s0 = (int)(sdv[q] & 0xffffu), d0 = (int)(sdv[q] >> 16);
      pp[q] = (s0 != d0) ? arc_posterior(me_f2(alpha[s0]), me_f2(beta[d0]), me_f2(th[ll[q]]), rz, ez, has_extra, xmv[q], xev[q]) : 0.0f;
      if (grad_theta && pp[q] > 0.0f) atomicAdd(&gth[ll[q]], pp[q]);
    }
    if (posterior) {
      // non-temporal: nobody reads the 20 MB of posteriors back inside this launch, and as ordinary stores they went through
      // the L2 in one burst behind the barrier (39.6 -> 38.6 us cold, 36.0 -> 34.8 replayed; profiles/tune/ab_variants.sh)
      typedef float f4v __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(f4v{pp[0], pp[1], pp[2], pp[3]}, reinterpret_cast<f4v *>(posterior + a));
    }
  };
  if (kPre > 0 && tid >= kSweepThreads && want_post) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
      const int a = v_begin + 4 * (u * kHelpers + (tid - kSweepThreads));
      if (a < v_end) do_group(psd[u], plb[u], a, pxm[u], pxe[u]);
    }
  }
  // the row outputs: every thread (the sweep waves start here, the others come when their preloaded
  // arc groups are done)
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logalpha) __builtin_nontemporal_store(me_log32(alpha[i]), logalpha + m.row_off + i);
    if (logbeta) __builtin_nontemporal_store(me_log32(beta[i]), logbeta + m.row_off + i);
    if (beta_me) beta_me[m.row_off + i] = me_f2(beta[i]);
  }
  if (want_post) {
    // the arc groups that were not preloaded: kPB groups per iteration, all loads issued
    // before the first use
    constexpr int kPB = 4;
    for (int a0 = v_begin + 4 * (kPre * kHelpers + tid); a0 < v_end; a0 += NT * 4 * kPB) {
      uint4 sd[kPB];
      uint2 lb[kPB];
      float4 xe[kPB];
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = min(a0 + u * NT * 4, v_end - 4);  // clamped: always a valid group
        sd[u] = *reinterpret_cast<const uint4 *>(lat.arc_sd + a);
        lb[u] = *reinterpret_cast<const uint2 *>(lat.arc_l16 + a);
        xe[u] = extras4(a);
      }
#pragma unroll
      for (int u = 0; u < kPB; ++u) {
        const int a = a0 + u * NT * 4;
        if (a >= v_end) break;
        float4 gm = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
        int4 ge = make_int4(0, 0, 0, 0);
        if (has_extra) {
          exp_me_fast(xe[u].x, gm.x, ge.x); exp_me_fast(xe[u].y, gm.y, ge.y);
          exp_me_fast(xe[u].z, gm.z, ge.z); exp_me_fast(xe[u].w, gm.w, ge.w);
        }
        do_group(sd[u], lb[u], a, gm, ge);
      }
    }
    // unaligned head and tail (at most 3 arcs each)
    const int n_head = min(v_begin, a_end) - a_begin;
    const int n_tail = (v_end >= v_begin) ? a_end - v_end : 0;
    if (tid < n_head + n_tail) {
      const int a = tid < n_head ? a_begin + tid : v_end + (tid - n_head);
      const int s0 = lat.arc_src[a], d0 = lat.arc_dst[a], l0 = lat.arc_label[a];
      float hm = 1.0f;
      int he = 0;
      if (has_extra) exp_me_fast(kCached ? xc[a - xc_first] : ex.at(a), hm, he);
      const float p = (s0 != d0) ? arc_posterior(me_f2(alpha[s0]), me_f2(beta[d0]), me_f2(th[l0]), rz, ez, has_extra, hm, he) : 0.0f;
      if (posterior) posterior[a] = p;
      if (grad_theta && p > 0.0f) atomicAdd(&gth[l0], p);
    }
    if (grad_theta) {
      __syncthreads();
      float *gout = grad_theta + (size_t)b * lat.vocab;
      for (int l = tid; l < lat.vocab; l += NT) gout[l] = gth[l];
    }
  }
#ifdef NFST_PROF
  __syncthreads();
  if (tid == 0) NFST_STAMP(7);
#endif
}
