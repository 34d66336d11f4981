// kernels.hip -- gfx950 (MI355X) kernels of the nFST lattice engine and their
// C-ABI launchers.  Design: DESIGN.md.  One workgroup owns one lattice; alpha and
// beta of all its states live in LDS as (mantissa, exponent) pairs -- an
// extended-exponent probability semiring: exact path sums like the reference's
// probability-domain beta sweep (/root/reference/src/modules/scorers.py:692-751)
// but without its float32 overflow and without exp/log on
// the level-to-level critical path.  Arc records stream once per sweep from HBM in
// level order; per-state sums are reduced by 2^k neighbouring lanes with wave64
// shuffles.  No MFMA: this is a sparse gather/reduce.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cmath>

#include "nfst_hip.h"

__device__ unsigned long long g_dbg[8192];
extern "C" int nfst_debug_read(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dbg), sizeof(unsigned long long)*8192); }
namespace {
#define STAMP(slot) do { if (dbg_on && t >= 20 && t < 52) { unsigned long long c_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(c_) :: "memory"); if (lane == 0) g_dbg[dbg_base + (t-20)*8 + (slot)] = c_; } } while(0)

constexpr int kEZero = -(1 << 28);      // exponent of an exact zero
constexpr int kViterbiWaves = 4;        // waves of the Viterbi kernel
constexpr float kNegInf = -__builtin_huge_valf();

struct ME {
  float m;
  int e;
};

// exp(x) = m * 2^e with m in [0.70, 1.42]; x = -inf (or below -1e30) gives zero.
__device__ __forceinline__ ME exp_split(float x) {
  ME r;
  if (!(x > -1e30f)) { r.m = 0.0f; r.e = kEZero; return r; }
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

__device__ __forceinline__ void me_acc(float &M, int &E, float mt, int et) {
  if (et > E) { M = ldexpf(M, E - et); E = et; }
  M += ldexpf(mt, et - E);
}

__device__ __forceinline__ float2 me_pack(float M, int E) {
  if (!(M > 0.0f)) return make_float2(0.0f, __int_as_float(kEZero));
  int ex;
  float mant = frexpf(M, &ex);
  return make_float2(mant, __int_as_float(E + ex));
}

// natural log of an (m, e) pair in float64 / float32
__device__ __forceinline__ double me_log64(float2 v) {
  if (!(v.x > 0.0f)) return -__builtin_huge_val();
  return log((double)v.x) + (double)__float_as_int(v.y) * 0.693147180559945309417232;
}
__device__ __forceinline__ float me_log32(float2 v) {
  if (!(v.x > 0.0f)) return kNegInf;
  return (float)((double)logf(v.x) + (double)__float_as_int(v.y) * 0.693147180559945309417232);
}

struct Meta {
  int row_off, n_rows, arc_off, n_arcs, fwd_off, fwd_steps, bwd_off, bwd_steps, sink, n_reach,
      depth, dp_off, n_dp, fwd_words, bwd_words;
};
__device__ __forceinline__ Meta load_meta(const int32_t *meta, int b) {
  const int32_t *m = meta + (size_t)b * NFST_META_WORDS;
  Meta r;
  r.row_off = m[NFST_META_ROW_OFF]; r.n_rows = m[NFST_META_N_ROWS];
  r.arc_off = m[NFST_META_ARC_OFF]; r.n_arcs = m[NFST_META_N_ARCS];
  r.fwd_off = m[NFST_META_FWD_OFF]; r.fwd_steps = m[NFST_META_FWD_STEPS];
  r.bwd_off = m[NFST_META_BWD_OFF]; r.bwd_steps = m[NFST_META_BWD_STEPS];
  r.sink = m[NFST_META_SINK]; r.n_reach = m[NFST_META_N_REACH]; r.depth = m[NFST_META_DEPTH];
  r.dp_off = m[NFST_META_DP_OFF]; r.n_dp = m[NFST_META_N_DP];
  r.fwd_words = m[NFST_META_FWD_WORDS]; r.bwd_words = m[NFST_META_BWD_WORDS];
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

// ---------------------------------------------------------------- LDS ring (LDS-DMA)
// A sweep consumes its stream strictly front to back and the stream does not
// depend on the DP values, so it is prefetched far ahead: the sweep's W waves copy
// 1-KiB chunks straight into a 16 KiB LDS ring with global_load_lds_dwordx4 (no
// VGPR staging).  The ring is 8 blocks of 512 words (2 chunks each); chunk c is
// issued by wave c % W and lives in slot c % 16.  A step is at most
// NFST_MAX_STEP_WORDS = 512 words, so while the read offset is in block b a step
// (plus the next step's 2-word header) touches blocks b .. b+2 only.  Protocol, run
// (plus the static data of the following step and the header after that, which the
// software pipeline reads early) touches blocks b .. b+3 only.  Protocol, run by
// every wave when the offset enters block b ("crossing", at the top of a step, i.e.
// after the barrier that ended the previous step):
//   1. blocks < b are dead: issue the chunks this wave owns of block b+7 into them;
//   2. counted wait: all of this wave's chunks of blocks <= b+4 have landed
//      (blocks b+5 .. b+7 may stay in flight -- the constant vmcnt below);
//   3. the barrier that ends this step publishes block b+4, one crossing before
//      any wave can read it.
// The prologue issues blocks 0..7 and waits for blocks 0..4 with the same constant.
constexpr int kRingWords = 4096;
constexpr int kRingMask = kRingWords - 1;
constexpr int kChunkWords = 256;
constexpr int kBlockShift = 9;  // 512-word blocks

template <int W>
struct Ring {
  uint32_t *lds;      // ring base in LDS
  const uint32_t *g;  // this lattice's stream (256-byte aligned)
  int total_chunks;
  int blk;            // block holding the current read offset
  int w;              // this wave's index within the sweep
};

template <int W>
__device__ __forceinline__ void ring_issue_block(const Ring<W> &r, int block, int lane) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = 2 * block + q;
    if ((c % W) != r.w) continue;
    uint32_t *dst = r.lds + (c & 15) * kChunkWords;
    if (c < r.total_chunks) {
      const uint32_t *src = r.g + (size_t)c * kChunkWords + lane * 4;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    } else {
      // past the end of the stream: a 4-byte-per-lane placeholder load into the (dead)
      // slot keeps the wave's vmcnt sequence identical, so the constant waits stay exact
      const uint32_t *src = r.g + lane;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                       (__attribute__((address_space(3))) void *)dst, 4, 0, 0);
    }
  }
}

template <int W>
__device__ __forceinline__ void ring_wait() {
  // chunks a wave may leave in flight: its share of blocks b+5 .. b+7
  if (W == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if (W == 2) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
}

template <int W>
__device__ __forceinline__ void ring_start(Ring<W> &r, uint32_t *lds, const uint32_t *g, int words, int w,
                                           int lane) {
  r.lds = lds; r.g = g; r.total_chunks = (words + kChunkWords - 1) / kChunkWords; r.blk = 0; r.w = w;
  for (int b = 0; b < 8; ++b) ring_issue_block(r, b, lane);
  ring_wait<W>();
}

template <int W>
__device__ __forceinline__ void ring_advance(Ring<W> &r, int off, int lane) {
  const int nb = off >> kBlockShift;
  if (nb != r.blk) {  // a step is at most one block long: nb == blk + 1
    r.blk = nb;
    ring_issue_block(r, nb + 7, lane);
    ring_wait<W>();
  }
}

// workgroup barrier that does not drain the LDS-DMA queue (a __syncthreads() would
// wait vmcnt(0)): this wave's LDS writes are complete, then s_barrier.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

// All-reduce of an (M, E) partial sum over groups of 2^KL neighbouring lanes:
// max of the exponents, one rescale, then the sum.  Quad permutes and half-row /
// row mirrors are DPP modifiers (no LDS traffic); 32- and 64-lane groups finish with
// shuffles.  Every lane of a group ends with bitwise the same (M, E).
template <int KL>
__device__ __forceinline__ void group_reduce(float &M, int &E) {
  int Em = E;
  if (KL >= 1) Em = max(Em, dpp_i<0xB1>(Em));   // quad_perm [1,0,3,2]
  if (KL >= 2) Em = max(Em, dpp_i<0x4E>(Em));   // quad_perm [2,3,0,1]
  if (KL >= 3) Em = max(Em, dpp_i<0x141>(Em));  // row_half_mirror
  if (KL >= 4) Em = max(Em, dpp_i<0x140>(Em));  // row_mirror
  if (KL >= 5) Em = max(Em, __shfl_xor(Em, 16));
  if (KL >= 6) Em = max(Em, __shfl_xor(Em, 32));
  if (KL >= 1) {
    M = ldexpf(M, E - Em);
    E = Em;
  }
  if (KL >= 1) M += dpp_f<0xB1>(M);
  if (KL >= 2) M += dpp_f<0x4E>(M);
  if (KL >= 3) M += dpp_f<0x141>(M);
  if (KL >= 4) M += dpp_f<0x140>(M);
  if (KL >= 5) M += __shfl_xor(M, 16);
  if (KL >= 6) M += __shfl_xor(M, 32);
}

// ---------------------------------------------------------------- the sweep proper
constexpr int kUnroll = 4;  // arcs per lane handled by the straight-line path

// scalar description of one step (wave-uniform, lives in SGPRs)
struct StepHdr {
  int off, ns, kl, na, st, rec, next_off, arc_base;
  bool accum;
};
__device__ __forceinline__ StepHdr make_hdr(int off, uint32_t h0, uint32_t na, int arc_base) {
  StepHdr h;
  h.off = off; h.ns = (int)(h0 & 0xffffu); h.kl = (int)((h0 >> 16) & 0xfu); h.na = (int)na;
  h.accum = ((h0 >> 20) & 1u) != 0;
  h.st = off + 2; h.rec = h.st + h.ns + 1; h.next_off = h.rec + h.na; h.arc_base = arc_base;
  return h;
}

// per-lane static data of one tile: the lane's state, its arc range and its first
// kUnroll arc records.  Nothing here depends on DP values, so a tile's TileRegs are
// fetched from the ring while the previous tile is being computed.
struct TileRegs {
  uint32_t rc[kUnroll];
  uint32_t sid;
  int a0, a1;  // this lane's arcs: a0, a0 + k, ... < a1 (both 0 for an idle lane)
};

__device__ __forceinline__ void tile_fetch(const uint32_t *ring, const StepHdr &h, int base, int lane,
                                           TileRegs &tr) {
  const int k = 1 << h.kl;
  const int i = base + (lane >> h.kl);
  const int r = lane & (k - 1);
  const uint32_t w0 = ring[(h.st + i) & kRingMask], w1 = ring[(h.st + i + 1) & kRingMask];
  const bool act = i < h.ns;
  tr.sid = w0 & 0xffffu;
  tr.a0 = act ? (int)(w0 >> 16) + r : 0;
  tr.a1 = act ? (int)(w1 >> 16) : 0;
#pragma unroll
  for (int j = 0; j < kUnroll; ++j) {
    const int a = tr.a0 + j * k;
    const uint32_t v = ring[(h.rec + a) & kRingMask];  // always a valid LDS address
    tr.rc[j] = (a < tr.a1) ? v : 0u;
  }
}

// One sum-product sweep over one direction's stream, run by W waves (index w) of the
// workgroup.  With W > 1 the waves meet at one barrier per step and every wave of the
// workgroup must call lds_barrier() exactly n_barriers + 1 times; with W == 1 a sweep
// is a single wave, its LDS accesses are ordered, and there is no barrier at all.
// Software pipeline, per tile: gathers of this tile (theta, alpha/beta) are issued
// first, then the static reads of the wave's next tile; the sum, the cross-lane
// reduce and the LDS write of this tile run while those are in flight.
template <int W>
__device__ __forceinline__ void ring_sweep(const uint32_t *g, int words, uint32_t *ring_lds, int my_steps,
                                           int n_barriers, float2 *val, const float2 *th, const Extra ex,
                                           const int32_t *__restrict__ perm, int w, int lane) {
  Ring<W> rg;
  ring_start(rg, ring_lds, g, words, w, lane);
  if (W > 1) lds_barrier();
  const uint32_t *ring = ring_lds;
  const bool has_extra = ex.any();
  const bool dbg_on = (blockIdx.x == 7);
  const int dbg_base = (int)(threadIdx.x >> 6) * 512;
  StepHdr H0 = make_hdr(0, 0, 0, 0), H1 = H0;
  if (my_steps > 0)
    H0 = make_hdr(0, __builtin_amdgcn_readfirstlane(ring[0]), __builtin_amdgcn_readfirstlane(ring[1]), 0);
  if (my_steps > 1)
    H1 = make_hdr(H0.next_off, __builtin_amdgcn_readfirstlane(ring[H0.next_off & kRingMask]),
                  __builtin_amdgcn_readfirstlane(ring[(H0.next_off + 1) & kRingMask]), H0.na);
  TileRegs cur;
  bool have = false;
  if (my_steps > 0 && w * (64 >> H0.kl) < H0.ns) {
    tile_fetch(ring, H0, w * (64 >> H0.kl), lane, cur);
    have = true;
  }
  for (int t = 0; t < n_barriers; ++t) {
    if (t < my_steps) {
      STAMP(0);
      ring_advance(rg, H0.off, lane);
      // header of step t+2 (static data that has already landed)
      const bool v1 = t + 1 < my_steps, v2 = t + 2 < my_steps;
      const uint32_t f0 = ring[H1.next_off & kRingMask], f1 = ring[(H1.next_off + 1) & kRingMask];
      const int spw = 64 >> H0.kl;
      const int k = 1 << H0.kl;
      for (int base = w * spw; base < H0.ns; base += W * spw) {
        if (!have) tile_fetch(ring, H0, base, lane, cur);
        STAMP(1);
        // --- A: gathers of this tile
        float2 tw[kUnroll], vv[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
          tw[j] = th[cur.rc[j] >> 16];
          vv[j] = val[cur.rc[j] & 0xffffu];
        }
        // --- B: static data of this wave's next tile (same step, else the next step)
        const int nbase = base + W * spw;
        const bool same = nbase < H0.ns;
        const int nb1 = w * (64 >> H1.kl);
        const bool hv = same || (v1 && nb1 < H1.ns);
        const StepHdr Hn = same ? H0 : (hv ? H1 : H0);
        const int nb = same ? nbase : (hv ? nb1 : base);
        TileRegs nxt;
        tile_fetch(ring, Hn, nb, lane, nxt);
        STAMP(2);
        // --- C: this tile's sum with one shared exponent
        float mt[kUnroll];
        int et[kUnroll];
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
          const bool ok = cur.a0 + j * k < cur.a1;
          float mw = tw[j].x;
          int ew = __float_as_int(tw[j].y);
          if (has_extra) {
            const int a = ok ? cur.a0 + j * k : 0;
            ME x = exp_split(ex.at(perm[H0.arc_base + a]));
            mw *= x.m;
            ew += x.e;
          }
          mt[j] = ok ? mw * vv[j].x : 0.0f;
          et[j] = ok ? ew + __float_as_int(vv[j].y) : kEZero;
        }
        int E = max(max(et[0], et[1]), max(et[2], et[3]));
        float M = (ldexpf(mt[0], et[0] - E) + ldexpf(mt[1], et[1] - E)) +
                  (ldexpf(mt[2], et[2] - E) + ldexpf(mt[3], et[3] - E));
        STAMP(3);
        // --- E: lanes with more than kUnroll arcs (the packer avoids this when it can)
        if (__any(cur.a0 + kUnroll * k < cur.a1)) {
          for (int a = cur.a0 + kUnroll * k; a < cur.a1; a += k) {
            const uint32_t rc = ring[(H0.rec + a) & kRingMask];
            const float2 t2 = th[rc >> 16], v2f = val[rc & 0xffffu];
            float mw = t2.x;
            int ew = __float_as_int(t2.y);
            if (has_extra) {
              ME x = exp_split(ex.at(perm[H0.arc_base + a]));
              mw *= x.m;
              ew += x.e;
            }
            me_acc(M, E, mw * v2f.x, ew + __float_as_int(v2f.y));
          }
        }
        STAMP(4);
        // --- F: reduce over the state's lanes, normalise, store
        switch (H0.kl) {
          case 0: group_reduce<0>(M, E); break;
          case 1: group_reduce<1>(M, E); break;
          case 2: group_reduce<2>(M, E); break;
          case 3: group_reduce<3>(M, E); break;
          case 4: group_reduce<4>(M, E); break;
          case 5: group_reduce<5>(M, E); break;
          default: group_reduce<6>(M, E); break;
        }
        if ((lane & (k - 1)) == 0 && base + (lane >> H0.kl) < H0.ns) {
          if (H0.accum) {
            const float2 old = val[cur.sid];
            me_acc(M, E, old.x, __float_as_int(old.y));
          }
          val[cur.sid] = me_pack(M, E);
        }
        STAMP(5);
        cur = nxt;
        have = hv;
      }
      // rotate the headers
      H0 = H1;
      if (v2) H1 = make_hdr(H1.next_off, __builtin_amdgcn_readfirstlane(f0), __builtin_amdgcn_readfirstlane(f1),
                            H1.arc_base + H1.na);
    }
    if (W > 1) lds_barrier();
    STAMP(6);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // nothing of the ring stays in flight
}

__device__ __forceinline__ void load_theta(float2 *th, const float *theta, int64_t stride, int b,
                                           int V, int tid, int nt) {
  const float *t = theta + (size_t)stride * b;
  for (int l = tid; l < V; l += nt) {
    ME x = exp_split(t[l]);
    th[l] = make_float2(x.m, __int_as_float(x.e));
  }
}

// ------------------------------------------------------------------ backward only
// W waves sweep the by-source stream from the sink (block = max(W, 2) * 64 threads...
// exactly W * 64 threads).
template <int W>
__global__ __launch_bounds__(W * 64) void k_backward(nfst_batch lat, nfst_scores sc, float *logbeta,
                                                     double *logz64, float *logz32, float2 *beta_me) {
  extern __shared__ float2 lds[];
  constexpr int NT = W * 64;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const Meta m = load_meta(lat.meta, b);
  const int rows2 = (lat.max_rows + 1) & ~1;
  float2 *beta = lds;
  float2 *th = lds + rows2;
  uint32_t *ring_lds = (uint32_t *)(th + ((lat.vocab + 1) & ~1));
  for (int i = tid; i < m.n_rows; i += NT) beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT);
  __syncthreads();
  if (tid == 0) beta[m.sink] = make_float2(0.5f, __int_as_float(1));
  __syncthreads();
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  ring_sweep<W>(lat.bwd_stream + m.bwd_off, m.bwd_words, ring_lds, m.bwd_steps, m.bwd_steps, beta, th, ex,
                lat.bwd_perm + m.dp_off, w, lane);
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
// Waves [0, W) run the beta sweep and waves [W, 2W) the alpha sweep, concurrently;
// then every wave of the block streams the canonical arcs once for the posteriors.
// W = 1: 256-thread block, the two sweeps are single waves that never synchronise
// (the other two waves wait at the barrier before the posterior pass).
template <int W>
struct FbGeom {
  static constexpr int kThreads = (2 * W * 64 < 256) ? 256 : 2 * W * 64;
};

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

template <int W>
__global__ __launch_bounds__(FbGeom<W>::kThreads) void k_forward_backward(
    nfst_batch lat, nfst_scores sc, float *__restrict__ logalpha, float *__restrict__ logbeta,
    double *__restrict__ logz64, float *__restrict__ logz32, float *__restrict__ posterior,
    float *__restrict__ grad_theta, float2 *__restrict__ beta_me) {
  extern __shared__ float2 lds[];
  constexpr int NT = FbGeom<W>::kThreads;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const Meta m = load_meta(lat.meta, b);
  const int rows2 = (lat.max_rows + 1) & ~1;
  const int v4 = (lat.vocab + 3) & ~3;
  float2 *alpha = lds;
  float2 *beta = lds + rows2;
  float2 *th = lds + 2 * rows2;
  float *gth = (float *)(th + v4);             // [V] label histogram (only if grad_theta)
  uint32_t *ring_lds = (uint32_t *)(gth + v4);  // two 16 KiB rings: beta stream, alpha stream
  for (int i = tid; i < m.n_rows; i += NT) {
    alpha[i] = make_float2(0.0f, __int_as_float(kEZero));
    beta[i] = make_float2(0.0f, __int_as_float(kEZero));
  }
  load_theta(th, sc.theta, sc.theta_stride, b, lat.vocab, tid, NT);
  if (grad_theta) for (int l = tid; l < lat.vocab; l += NT) gth[l] = 0.0f;
  __syncthreads();
  if (tid == 0) {
    beta[m.sink] = make_float2(0.5f, __int_as_float(1));
    alpha[0] = make_float2(0.5f, __int_as_float(1));
  }
  __syncthreads();
  const Extra ex{lat.weighted ? lat.arc_w : nullptr, sc.arc_scores};
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int all_steps = max(m.fwd_steps, m.bwd_steps);
  if (wv < W) {
    ring_sweep<W>(lat.bwd_stream + m.bwd_off, m.bwd_words, ring_lds, m.bwd_steps, all_steps, beta, th, ex,
                  lat.bwd_perm + m.dp_off, wv, lane);
  } else if (wv < 2 * W) {
    ring_sweep<W>(lat.fwd_stream + m.fwd_off, m.fwd_words, ring_lds + kRingWords, m.fwd_steps, all_steps,
                  alpha, th, ex, lat.fwd_perm + m.dp_off, wv - W, lane);
  }
  __syncthreads();
  const float2 zme = beta[0];
  if (tid == 0) {
    const double z = me_log64(zme);
    if (logz64) logz64[b] = z;
    if (logz32) logz32[b] = (float)z;
  }
  for (int i = tid; i < m.n_rows; i += NT) {
    if (logalpha) logalpha[m.row_off + i] = me_log32(alpha[i]);
    if (logbeta) logbeta[m.row_off + i] = me_log32(beta[i]);
    if (beta_me) beta_me[m.row_off + i] = beta[i];
  }
  if (posterior || grad_theta) {
    const float rz = (zme.x > 0.0f) ? 1.0f / zme.x : 0.0f;
    const int ez = __float_as_int(zme.y);
    const bool has_extra = ex.any();
    const int a_begin = m.arc_off, a_end = m.arc_off + m.n_arcs;
    // 4 arcs per lane and iteration with 16-byte loads/stores on the aligned interior
    const int v_begin = (a_begin + 3) & ~3, v_end = a_end & ~3;
    for (int a = v_begin + tid * 4; a < v_end; a += NT * 4) {
      const int4 s4 = *reinterpret_cast<const int4 *>(lat.arc_src + a);
      const int4 d4 = *reinterpret_cast<const int4 *>(lat.arc_dst + a);
      const int4 l4 = *reinterpret_cast<const int4 *>(lat.arc_label + a);
      const int ss[4] = {s4.x, s4.y, s4.z, s4.w}, dd[4] = {d4.x, d4.y, d4.z, d4.w},
                ll[4] = {l4.x, l4.y, l4.z, l4.w};
      float pp[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pp[q] = (ss[q] != dd[q]) ? arc_posterior(alpha[ss[q]], beta[dd[q]], th[ll[q]], rz, ez, has_extra, ex, a + q)
                                 : 0.0f;
        if (grad_theta && pp[q] > 0.0f) atomicAdd(&gth[ll[q]], pp[q]);
      }
      if (posterior) *reinterpret_cast<float4 *>(posterior + a) = make_float4(pp[0], pp[1], pp[2], pp[3]);
    }
    // unaligned head and tail (at most 3 arcs each)
    const int n_head = min(v_begin, a_end) - a_begin;
    const int n_tail = (v_end >= v_begin) ? a_end - v_end : 0;
    if (tid < n_head + n_tail) {
      const int a = tid < n_head ? a_begin + tid : v_end + (tid - n_head);
      const int s = lat.arc_src[a], d = lat.arc_dst[a], l = lat.arc_label[a];
      const float p = (s != d) ? arc_posterior(alpha[s], beta[d], th[l], rz, ez, has_extra, ex, a) : 0.0f;
      if (posterior) posterior[a] = p;
      if (grad_theta && p > 0.0f) atomicAdd(&gth[l], p);
    }
    if (grad_theta) {
      __syncthreads();
      float *g = grad_theta + (size_t)b * lat.vocab;
      for (int l = tid; l < lat.vocab; l += NT) g[l] = gth[l];
    }
  }
}

// ------------------------------------------------------------------ Viterbi
// max-plus sweep over the by-source stream; float32 values, canonical arc back
// pointers; thread 0 then walks the best path.
__global__ __launch_bounds__(256) void k_viterbi(nfst_batch lat, nfst_scores sc, float *best,
                                                 int32_t *paths, int32_t *path_arcs,
                                                 int32_t *lengths, int max_len, int pad) {
  extern __shared__ float2 lds[];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const Meta m = load_meta(lat.meta, b);
  float *v = (float *)lds;
  int *bp = (int *)(v + lat.max_rows);
  float *th = (float *)(bp + lat.max_rows);
  for (int i = tid; i < m.n_rows; i += 256) { v[i] = kNegInf; bp[i] = -1; }
  const float *tg = sc.theta + (size_t)sc.theta_stride * b;
  for (int l = tid; l < lat.vocab; l += 256) th[l] = tg[l];
  __syncthreads();
  if (tid == 0) v[m.sink] = 0.0f;
  __syncthreads();
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const uint32_t *stream = lat.bwd_stream + m.bwd_off;
  const int32_t *perm = lat.bwd_perm + m.dp_off;
  int off = 0, arc_base = 0;
  for (int t = 0; t < m.bwd_steps; ++t) {
    const uint32_t *step = stream + off;
    const uint32_t h0 = __builtin_amdgcn_readfirstlane(step[0]);
    const int na = (int)__builtin_amdgcn_readfirstlane(step[1]);
    const int ns = (int)(h0 & 0xffffu), kl = (int)((h0 >> 16) & 0xfu);
    const bool accum = ((h0 >> 20) & 1u) != 0;
    const uint32_t *st = step + 2, *rec = st + ns + 1;
    const int spw = 64 >> kl, k = 1 << kl;
    for (int base = wave * spw; base < ns; base += kViterbiWaves * spw) {
      const int i = base + (lane >> kl), r = lane & (k - 1);
      float bv = kNegInf;
      int ba = 0x7fffffff;
      uint32_t sid = 0;
      if (i < ns) {
        const uint32_t w0 = st[i], w1 = st[i + 1];
        sid = w0 & 0xffffu;
        for (int a = (int)(w0 >> 16) + r; a < (int)(w1 >> 16); a += k) {
          const uint32_t rc = rec[a];
          const int ca = perm[arc_base + a];
          float s = th[rc >> 16];
          if (arc_w) s += arc_w[ca];
          if (sc.arc_scores) s += sc.arc_scores[ca];
          const float c = s + v[rc & 0xffffu];
          if (c > bv) { bv = c; ba = ca; }
        }
      }
      for (int d = 1; d < k; d <<= 1) {
        const float ov = __shfl_xor(bv, d);
        const int oa = __shfl_xor(ba, d);
        if (ov > bv || (ov == bv && oa < ba)) { bv = ov; ba = oa; }
      }
      if (i < ns && r == 0) {
        if (accum) {
          const float ov = v[sid];
          const int oa = bp[sid];
          if (ov > bv || (ov == bv && oa >= 0 && oa < ba)) { bv = ov; ba = oa; }
        }
        v[sid] = bv;
        bp[sid] = (ba == 0x7fffffff) ? -1 : ba;
      }
    }
    off += 2 + ns + 1 + na;
    arc_base += na;
    __syncthreads();
  }
  if (tid == 0) {
    best[b] = v[0];
    int s = 0, len = 0;
    while (s != m.sink && len < max_len) {
      const int a = bp[s];
      if (a < 0) break;
      paths[(size_t)b * max_len + len] = lat.arc_label[a];
      if (path_arcs) path_arcs[(size_t)b * max_len + len] = a;
      ++len;
      s = lat.arc_dst[a];
    }
    lengths[b] = (s == m.sink) ? len : -1;
    for (int j = len; j < max_len; ++j) {
      paths[(size_t)b * max_len + j] = pad;
      if (path_arcs) path_arcs[(size_t)b * max_len + j] = -1;
    }
  }
}

// ------------------------------------------------------------------ sampling
__device__ __forceinline__ uint32_t mulhilo(uint32_t a, uint32_t b, uint32_t *hi) {
  const uint64_t p = (uint64_t)a * b;
  *hi = (uint32_t)(p >> 32);
  return (uint32_t)p;
}
// Philox4x32-10, counter (walk, step, 0, 0), key from seed; first output word -> [0,1)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t walk, uint32_t step) {
  uint32_t c0 = walk, c1 = step, c2 = 0, c3 = 0;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  for (int i = 0; i < 10; ++i) {
    uint32_t hi0, hi1;
    const uint32_t lo0 = mulhilo(0xD2511F53u, c0, &hi0);
    const uint32_t lo1 = mulhilo(0xCD9E8D57u, c2, &hi1);
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return (float)(c0 >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float arc_score(const float *theta, const float *arc_w,
                                           const float *arc_scores, int l, int a) {
  float s = theta[l];
  if (arc_w) s += arc_w[a];
  if (arc_scores) s += arc_scores[a];
  return s;
}

__global__ __launch_bounds__(64) void k_sample(nfst_batch lat, nfst_scores sc,
                                               const float2 *beta_me, const double *logz64, int K,
                                               int max_len, const float *uniforms, uint64_t seed,
                                               int pad, int32_t *paths, int32_t *path_arcs,
                                               int32_t *lengths, float *logq, int32_t *status) {
  const int b = blockIdx.x;
  const int k = blockIdx.y * 64 + threadIdx.x;
  if (k >= K) return;
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  const float2 *bme = beta_me + m.row_off;
  const size_t walk = (size_t)b * K + k;
  int32_t *out = paths + walk * max_len;
  int32_t *outa = path_arcs ? path_arcs + walk * max_len : nullptr;
  int s = 0, t = 0;
  float tot = 0.0f;
  bool ok = true;
  while (s != m.sink) {
    if (t >= max_len) { ok = false; break; }
    const float u = uniforms ? uniforms[walk * max_len + t] : philox_uniform(seed, (uint32_t)walk, (uint32_t)t);
    const float2 bs = bme[s];
    const float rs = 1.0f / bs.x;
    const int es = __float_as_int(bs.y);
    float cum = 0.0f, sc_ch = 0.0f, sc_last = 0.0f;
    int chosen = -1, last = -1;
    for (int a = rp[s]; a < rp[s + 1]; ++a) {
      const int d = lat.arc_dst[a];
      if (d == s) continue;
      const float x = arc_score(theta, arc_w, sc.arc_scores, lat.arc_label[a], a);
      const ME wgt = exp_split(x);
      const float2 bd = bme[d];
      const float p = ldexpf((wgt.m * bd.x) * rs, max(wgt.e + __float_as_int(bd.y) - es, -300));
      if (p > 0.0f) {
        cum += p;
        last = a;
        sc_last = x;
        if (u < cum) { chosen = a; sc_ch = x; break; }
      }
    }
    if (chosen < 0) { chosen = last; sc_ch = sc_last; }
    if (chosen < 0) { ok = false; break; }
    out[t] = lat.arc_label[chosen];
    if (outa) outa[t] = chosen;
    tot += sc_ch;
    s = lat.arc_dst[chosen];
    ++t;
  }
  if (!ok) atomicExch(status, NFST_ERR_LENGTH);
  lengths[walk] = ok ? t : -1;
  for (int j = t; j < max_len; ++j) { out[j] = pad; if (outa) outa[j] = -1; }
  logq[walk] = ok ? (float)((double)tot - logz64[b]) : kNegInf;
}

// binary search of `label` in the canonical row [r0, r1); returns arc id or -1
__device__ __forceinline__ int find_arc(const int32_t *arc_label, int r0, int r1, int label) {
  int lo = r0, hi = r1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (arc_label[mid] < label) lo = mid + 1; else hi = mid;
  }
  return (lo < r1 && arc_label[lo] == label) ? lo : -1;
}

__global__ __launch_bounds__(64) void k_score_paths(nfst_batch lat, nfst_scores sc,
                                                    const int32_t *marks, int K, int max_len,
                                                    float *path_score, int32_t *end_state) {
  const int b = blockIdx.x;
  const int k = blockIdx.y * 64 + threadIdx.x;
  if (k >= K) return;
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  const int32_t *rp = lat.row_ptr + m.row_off + b;
  const size_t walk = (size_t)b * K + k;
  const int32_t *mk = marks + walk * max_len;
  int s = 0;
  float tot = 0.0f;
  for (int t = 0; t < max_len; ++t) {
    const int l = mk[t];
    const int a = (l >= 0 && l < lat.vocab) ? find_arc(lat.arc_label, rp[s], rp[s + 1], l) : -1;
    if (a < 0) { tot = kNegInf; s = 0; break; }
    const int d = lat.arc_dst[a];
    if (d != s) tot += arc_score(theta, arc_w, sc.arc_scores, l, a);
    s = d;
  }
  path_score[walk] = tot;
  end_state[walk] = s;
}

// ------------------------------------------------------------------ per-step gathers
__global__ void k_step(nfst_batch lat, const int64_t *state, const int64_t *label, int64_t *next,
                       int K, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int b = (int)(i / K);
  const Meta m = load_meta(lat.meta, b);
  const int64_t s = state[i], l = label[i];
  int64_t r = 0;
  if (s >= 0 && s < m.n_rows && l >= 0 && l < lat.vocab) {
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    const int a = find_arc(lat.arc_label, rp[s], rp[s + 1], (int)l);
    if (a >= 0) r = lat.arc_dst[a];
  }
  next[i] = r;
}

// MODE 0: emission mask (0 / weight / -inf); MODE 1: values[row_off + transition[state, l]]
template <int MODE>
__global__ __launch_bounds__(64) void k_row_gather(nfst_batch lat, const int64_t *state,
                                                   const float *values, const int64_t *inp, int pad,
                                                   int bos, int eos, int has_to_end, float *out,
                                                   int K) {
  const int64_t i = blockIdx.x;
  const int b = (int)(i / K);
  const Meta m = load_meta(lat.meta, b);
  const int64_t s = state[i];
  int r0 = 0, r1 = 0;
  if (s >= 0 && s < m.n_rows) {
    const int32_t *rp = lat.row_ptr + m.row_off + b;
    r0 = rp[s]; r1 = rp[s + 1];
  }
  float *o = out + (size_t)i * lat.vocab;
  for (int l = threadIdx.x; l < lat.vocab; l += 64) {
    const int a = find_arc(lat.arc_label, r0, r1, l);
    float v;
    if (MODE == 0) {
      v = (a < 0) ? kNegInf : (lat.weighted ? lat.arc_w[a] : 0.0f);
      if (inp) {
        const int64_t p = inp[i];
        const bool ended = (p == eos) || (p == pad);
        if (l == bos || (ended ? (l != pad) : (l == pad))) v = kNegInf;
        if (has_to_end && !ended && l != eos) v = kNegInf;
      }
    } else {
      v = values[m.row_off + (a < 0 ? 0 : lat.arc_dst[a])];
    }
    o[l] = v;
  }
}

__global__ void k_gather_label_scores(nfst_batch lat, nfst_scores sc, float *out) {
  const int b = blockIdx.y;
  const Meta m = load_meta(lat.meta, b);
  const float *theta = sc.theta + (size_t)sc.theta_stride * b;
  const float *arc_w = lat.weighted ? lat.arc_w : nullptr;
  for (int a = m.arc_off + blockIdx.x * blockDim.x + threadIdx.x; a < m.arc_off + m.n_arcs;
       a += gridDim.x * blockDim.x)
    out[a] = arc_score(theta, arc_w, sc.arc_scores, lat.arc_label[a], a);
}

// ------------------------------------------------------------------ sequence scoring
// One 256-thread workgroup per sequence; each wave takes positions t = wave,
// wave+4, ...: masked (log-)softmax over V with wave64 shuffles, gather of the
// realised mark, pad positions contribute 0 (scorers.py:1564-1611).
__device__ __forceinline__ float seq_mask(int v, int t, int prev, int pad, int bos, int eos,
                                          int max_length) {
  float mk = 0.0f;
  if (t == 0) {
    if (v == bos || v == pad) mk = kNegInf;
    return mk;
  }
  const bool ended = (prev == eos) || (prev == pad);
  if (ended ? (v != pad) : (v == pad)) mk = kNegInf;
  if (v == bos) mk = kNegInf;
  if (max_length >= 0 && t > max_length && !ended && v != eos) mk = kNegInf;
  return mk;
}

__global__ __launch_bounds__(256) void k_path_logprob(const float *scores, const int64_t *marks,
                                                      int T, int V, int pad, int bos, int eos,
                                                      int max_length, float temp, int normalize,
                                                      float *out) {
  __shared__ float part[4];
  const int64_t n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t *mk = marks + n * T;
  float acc = 0.0f;
  for (int t = wave; t < T; t += 4) {
    const float *row = scores + ((size_t)n * T + t) * V;
    const int prev = t > 0 ? (int)mk[t - 1] : -1;
    const int lab = (int)mk[t];
    float sel;
    if (normalize) {
      // online max / sum over the lane's slice, then a wave reduction
      float mx = kNegInf, sm = 0.0f;
      for (int v = lane; v < V; v += 64) {
        const float msk = seq_mask(v, t, prev, pad, bos, eos, max_length);
        const float x = ((v == pad ? 0.0f : row[v]) + msk) / temp + msk;
        if (x > mx) { sm = sm * expf(mx - x) + 1.0f; mx = x; }
        else if (x > kNegInf) sm += expf(x - mx);
      }
      for (int d = 32; d >= 1; d >>= 1) {
        const float omx = __shfl_xor(mx, d), osm = __shfl_xor(sm, d);
        const float nm = fmaxf(mx, omx);
        const float a = (mx > kNegInf) ? sm * expf(mx - nm) : 0.0f;
        const float c = (omx > kNegInf) ? osm * expf(omx - nm) : 0.0f;
        sm = a + c;
        mx = nm;
      }
      const float msk = seq_mask(lab, t, prev, pad, bos, eos, max_length);
      const float x = ((lab == pad ? 0.0f : row[lab]) + msk) / temp + msk;
      sel = x - (mx + logf(sm));  // all -inf row: -inf - (-inf + log 0) = NaN, like the reference
    } else {
      const float msk = seq_mask(lab, t, prev, pad, bos, eos, max_length);
      sel = ((lab == pad ? 0.0f : row[lab]) + msk) / temp + msk;
    }
    acc += sel * (lab != pad ? 1.0f : 0.0f);
  }
  // every lane of a wave holds the same acc; reduce the 4 waves in a fixed order
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[n] = ((part[0] + part[1]) + part[2]) + part[3];
}

__global__ void k_iwae(const float *log_p, const float *log_q, int B, int K, float *log_w,
                       float *log_marginal) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  float mx = kNegInf;
  for (int k = 0; k < K; ++k) {
    const float w = log_p[(size_t)b * K + k] - log_q[(size_t)b * K + k];
    log_w[(size_t)b * K + k] = w;
    mx = fmaxf(mx, w);
  }
  float sm = 0.0f;
  for (int k = 0; k < K; ++k) sm += expf(log_w[(size_t)b * K + k] - mx);
  log_marginal[b] = (mx + logf(sm)) - logf((float)K);
}

// ------------------------------------------------------------------ host helpers
int check_batch(const nfst_batch *lat) {
  if (!lat || lat->n_lattices <= 0 || lat->vocab <= 0 || lat->max_rows <= 0) return NFST_ERR_ARG;
  if (!lat->meta || !lat->row_ptr || !lat->fwd_stream || !lat->bwd_stream) return NFST_ERR_ARG;
  if (lat->total_arcs > 0 && (!lat->arc_src || !lat->arc_dst || !lat->arc_label)) return NFST_ERR_ARG;
  if (lat->total_dp_arcs > 0 && (!lat->fwd_perm || !lat->bwd_perm)) return NFST_ERR_ARG;
  if (lat->weighted && !lat->arc_w) return NFST_ERR_ARG;
  if (lat->max_rows > NFST_MAX_ROWS || lat->vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  if (lat->max_step_words <= 0 || lat->max_step_words > NFST_MAX_STEP_WORDS) return NFST_ERR_LIMIT;
  if (lat->sweep_waves != 1 && lat->sweep_waves != 2 && lat->sweep_waves != 4) return NFST_ERR_ARG;
  if (((uintptr_t)lat->fwd_stream | (uintptr_t)lat->bwd_stream) & 15) return NFST_ERR_ARG;
  return NFST_OK;
}
int check_scores(const nfst_batch *lat, const nfst_scores *sc) {
  if (!sc || !sc->theta) return NFST_ERR_ARG;
  if (sc->theta_stride != 0 && sc->theta_stride < lat->vocab) return NFST_ERR_ARG;
  return NFST_OK;
}
int hip_status(hipError_t e) { return e == hipSuccess ? NFST_OK : NFST_ERR_HIP; }

constexpr int64_t kMaxLds = 160 * 1024;

template <class K>
int set_lds(K kernel, int64_t bytes) {
  if (bytes > kMaxLds) return NFST_ERR_LIMIT;
  if (bytes > 64 * 1024)
    return hip_status(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return NFST_OK;
}

}  // namespace

extern "C" {

int nfst_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0 ? 1 : 0;
}

int64_t nfst_lds_bytes(const nfst_batch *lat) {
  if (!lat) return NFST_ERR_ARG;
  const int64_t rows2 = (lat->max_rows + 1) & ~1, v4 = (lat->vocab + 3) & ~3;
  return (2 * rows2 + v4) * 8 + v4 * 4 + 2 * (int64_t)kRingWords * 4;
}

int nfst_backward(const nfst_batch *lat, const nfst_scores *scores, float *logbeta, double *logz64,
                  float *logz32, float *beta_me, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  const int64_t lds = (((int64_t)lat->max_rows + 1) / 2 * 2 + ((int64_t)lat->vocab + 1) / 2 * 2) * 8 +
                      (int64_t)kRingWords * 4;
#define NFST_LAUNCH_BWD(W)                                                                              \
  {                                                                                                     \
    if ((rc = set_lds(k_backward<W>, lds))) return rc;                                                  \
    hipLaunchKernelGGL(k_backward<W>, dim3(lat->n_lattices), dim3(W * 64), (size_t)lds,                 \
                       (hipStream_t)stream, *lat, *scores, logbeta, logz64, logz32, (float2 *)beta_me); \
  }
  switch (lat->sweep_waves) {
    case 1: NFST_LAUNCH_BWD(1) break;
    case 2: NFST_LAUNCH_BWD(2) break;
    default: NFST_LAUNCH_BWD(4) break;
  }
#undef NFST_LAUNCH_BWD
  return hip_status(hipGetLastError());
}

int nfst_forward_backward(const nfst_batch *lat, const nfst_scores *scores, float *logalpha,
                          float *logbeta, double *logz64, float *logz32, float *posterior,
                          float *grad_theta, float *beta_me, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (posterior && ((uintptr_t)posterior & 15)) return NFST_ERR_ARG;
  if (((uintptr_t)lat->arc_src | (uintptr_t)lat->arc_dst | (uintptr_t)lat->arc_label) & 15) return NFST_ERR_ARG;
  const int64_t lds = nfst_lds_bytes(lat);
#define NFST_LAUNCH_FB(W)                                                                               \
  {                                                                                                     \
    if ((rc = set_lds(k_forward_backward<W>, lds))) return rc;                                          \
    hipLaunchKernelGGL(k_forward_backward<W>, dim3(lat->n_lattices), dim3(FbGeom<W>::kThreads),         \
                       (size_t)lds, (hipStream_t)stream, *lat, *scores, logalpha, logbeta, logz64,      \
                       logz32, posterior, grad_theta, (float2 *)beta_me);                               \
  }
  switch (lat->sweep_waves) {
    case 1: NFST_LAUNCH_FB(1) break;
    case 2: NFST_LAUNCH_FB(2) break;
    default: NFST_LAUNCH_FB(4) break;
  }
#undef NFST_LAUNCH_FB
  return hip_status(hipGetLastError());
}

int nfst_viterbi(const nfst_batch *lat, const nfst_scores *scores, float *best, int32_t *paths,
                 int32_t *path_arcs, int32_t *lengths, int32_t max_len, int32_t pad, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!best || !paths || !lengths || max_len <= 0) return NFST_ERR_ARG;
  const int64_t lds = (int64_t)lat->max_rows * 8 + (int64_t)lat->vocab * 4;
  if ((rc = set_lds(k_viterbi, lds))) return rc;
  hipLaunchKernelGGL(k_viterbi, dim3(lat->n_lattices), dim3(256), (size_t)lds, (hipStream_t)stream,
                     *lat, *scores, best, paths, path_arcs, lengths, (int)max_len, (int)pad);
  return hip_status(hipGetLastError());
}

int nfst_sample_paths(const nfst_batch *lat, const nfst_scores *scores, const float *beta_me,
                      const double *logz64, int32_t k, int32_t max_len, const float *uniforms,
                      uint64_t seed, int32_t pad, int32_t *paths, int32_t *path_arcs,
                      int32_t *lengths, float *logq, int32_t *status, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!beta_me || !logz64 || !paths || !lengths || !logq || !status || k <= 0 || max_len <= 0)
    return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_sample, dim3(lat->n_lattices, (k + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                     *lat, *scores, (const float2 *)beta_me, logz64, (int)k, (int)max_len, uniforms,
                     seed, (int)pad, paths, path_arcs, lengths, logq, status);
  return hip_status(hipGetLastError());
}

int nfst_score_paths(const nfst_batch *lat, const nfst_scores *scores, const int32_t *marks, int32_t k,
                     int32_t max_len, float *path_score, int32_t *end_state, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!marks || !path_score || !end_state || k <= 0 || max_len <= 0) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_score_paths, dim3(lat->n_lattices, (k + 63) / 64), dim3(64), 0,
                     (hipStream_t)stream, *lat, *scores, marks, (int)k, (int)max_len, path_score,
                     end_state);
  return hip_status(hipGetLastError());
}

int nfst_step(const nfst_batch *lat, const int64_t *state, const int64_t *label, int64_t *next,
              int32_t k, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!state || !label || !next || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_step, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *lat,
                     state, label, next, (int)k, n);
  return hip_status(hipGetLastError());
}

int nfst_emission_mask(const nfst_batch *lat, const int64_t *state, const int64_t *inp, int32_t pad,
                       int32_t bos, int32_t eos, int32_t has_to_end, float *out, int32_t k,
                       void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!state || !out || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_row_gather<0>, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, *lat, state,
                     (const float *)nullptr, inp, (int)pad, (int)bos, (int)eos, (int)has_to_end, out,
                     (int)k);
  return hip_status(hipGetLastError());
}

int nfst_beta_logits(const nfst_batch *lat, const float *values, const int64_t *state, float *out,
                     int32_t k, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!values || !state || !out || k <= 0) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_row_gather<1>, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, *lat, state,
                     values, (const int64_t *)nullptr, 0, 0, 0, 0, out, (int)k);
  return hip_status(hipGetLastError());
}

int nfst_gather_label_scores(const nfst_batch *lat, const nfst_scores *scores, float *out, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!out) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_gather_label_scores, dim3(8, lat->n_lattices), dim3(256), 0, (hipStream_t)stream,
                     *lat, *scores, out);
  return hip_status(hipGetLastError());
}

int nfst_path_logprob(const float *scores, const int64_t *marks, int64_t n, int32_t t, int32_t vocab,
                      int32_t pad, int32_t bos, int32_t eos, int32_t max_length, float temp,
                      int32_t normalize, float *out, void *stream) {
  if (!scores || !marks || !out || n <= 0 || t <= 0 || vocab <= 0 || !(temp > 0.0f)) return NFST_ERR_ARG;
  if (n > 0x7fffffffll) return NFST_ERR_LIMIT;
  hipLaunchKernelGGL(k_path_logprob, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, scores, marks,
                     (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length, temp,
                     (int)normalize, out);
  return hip_status(hipGetLastError());
}

int nfst_iwae(const float *log_p, const float *log_q, int32_t b, int32_t k, float *log_w,
              float *log_marginal, void *stream) {
  if (!log_p || !log_q || !log_w || !log_marginal || b <= 0 || k <= 0) return NFST_ERR_ARG;
  hipLaunchKernelGGL(k_iwae, dim3((b + 127) / 128), dim3(128), 0, (hipStream_t)stream, log_p, log_q,
                     (int)b, (int)k, log_w, log_marginal);
  return hip_status(hipGetLastError());
}

}  // extern "C"
