// kernels.hip -- gfx950 (MI355X) kernels of the nFST lattice engine and their
// C-ABI launchers.  Design: DESIGN.md.  One workgroup owns one lattice; alpha and
// beta of all its states live in LDS as (mantissa, exponent) pairs -- an
// extended-exponent probability semiring: exact path sums like the reference's
// probability-domain beta sweep (/root/reference/src/modules/scorers.py:692-751)
// but without its float32 overflow and without exp/log on
// the level-to-level critical path.  Arc records stream once per sweep from HBM in
// level order; per-state sums are reduced by 2^k neighbouring lanes with wave64
// shuffles.  The sweeps use no MFMA: they are sparse gather/reduce work; the neuralised
// beta sweep (neural_kernels.h) has one dense product per state and runs it on float32 MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "nfst_hip.h"

namespace {

#include "semiring.h"
#include "tile_pipeline.h"
#include "fb_kernels.h"
#include "path_kernels.h"
#include "neural_kernels.h"
#include "pack_kernels.h"
#include "chunk_kernels.h"

// ------------------------------------------------------------------ host helpers
int check_batch(const nfst_batch *lat) {
  if (!lat || lat->n_lattices <= 0 || lat->vocab <= 0 || lat->max_rows <= 0) return NFST_ERR_ARG;
  if (!lat->meta || !lat->row_ptr || !lat->fwd_stream || !lat->bwd_stream) return NFST_ERR_ARG;
  if (lat->total_arcs > 0 && (!lat->arc_src || !lat->arc_dst || !lat->arc_label)) return NFST_ERR_ARG;
  if ((lat->fwd_slots > 0 && !lat->fwd_perm) || (lat->bwd_slots > 0 && !lat->bwd_perm)) return NFST_ERR_ARG;
  if (lat->weighted && !lat->arc_w) return NFST_ERR_ARG;
  if (lat->max_rows > NFST_MAX_ROWS || lat->vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
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

// Dynamic LDS above 64 KiB needs a per-kernel opt-in; it is sticky per (device, kernel), so it is
// requested once per device, kernel and size (hipFuncSetAttribute is slow and not capturable in a
// graph).  The cache is the library's only process state: keyed by the current device and guarded
// by a mutex, so several devices in one process and launches from several host threads are safe.
struct LdsSeen { int dev; const void *fn; int64_t bytes; };
std::mutex g_lds_mutex;
std::vector<LdsSeen> g_lds_seen;

template <class K>
int set_lds(K kernel, int64_t bytes) {
  if (bytes > kMaxLds) return NFST_ERR_LIMIT;
  if (bytes <= 64 * 1024) return NFST_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NFST_ERR_HIP;
  const void *fn = reinterpret_cast<const void *>(kernel);
  std::lock_guard<std::mutex> lock(g_lds_mutex);
  for (LdsSeen &e : g_lds_seen)
    if (e.dev == dev && e.fn == fn) {
      if (e.bytes >= bytes) return NFST_OK;
      int rc = hip_status(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      if (rc == NFST_OK) e.bytes = bytes;
      return rc;
    }
  int rc = hip_status(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (rc == NFST_OK) g_lds_seen.push_back({dev, fn, bytes});
  return rc;
}

// number of CUs of the current device (cached per device; 256 on MI355X)
int cu_count() {
  static std::mutex mu;
  static std::vector<int> per_dev;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if ((int)per_dev.size() <= dev) per_dev.resize(dev + 1, 0);
  if (per_dev[dev] == 0) {
    int v = 0;
    per_dev[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return per_dev[dev];
}

// Switches of the launchers.  The environment is read ONCE, when the first launcher runs (never on the launch path
// afterwards); nfst_tuning_set() changes a switch from the host side (tests and A/B measurements flip flavours inside
// one process).  Every switch only selects among kernels that compute the same function.
struct Tuning {
  int64_t lds_reserve = 0;  // NFST_LDS_RESERVE_KB: LDS the one-lattice-per-CU flavours leave free on every CU
  int tw = 1;               // NFST_TW=0: loader + decoder + sweep instead of tile waves
  int fused = 1;            // NFST_NO_FUSED=1: never the fused sweeps
  int xcache = 1;           // NFST_XCACHE=0: per-arc extras gathered from HBM / L2, never staged in LDS
  int precise = -1;         // NFST_PRECISE: 0 never, 1 whenever it fits, unset: programs deeper than kPreciseTiles tiles
  int neu_pack = 1;         // NFST_NEU_PACK=0: phase B reads Wh from the matrix itself
  int neu_bf16 = 1;         // NFST_NEU_BF16=0: phase B on float32 MFMAs instead of three bfloat16 parts (hid a multiple of 64)
  int neu_small = 1;        // NFST_NEU_NO_SMALL=1: two-phase neural kernels for every hidden size
  int chunked = 1;          // NFST_CHUNKED=0: never the chunked flavour (a batch with chunked programs runs the general kernels)
};
Tuning &tuning() {
  static Tuning t = [] {
    Tuning r;
    auto num = [](const char *name, long dflt) { const char *e = getenv(name); return e && *e ? strtol(e, nullptr, 10) : dflt; };
    const long kb = num("NFST_LDS_RESERVE_KB", 0);
    r.lds_reserve = (kb > 0 && kb <= 96) ? kb * 1024 : 0;
    r.tw = num("NFST_TW", 1) != 0;
    r.fused = num("NFST_NO_FUSED", 0) != 1;
    r.xcache = num("NFST_XCACHE", 1) != 0;
    r.precise = (int)num("NFST_PRECISE", -1);
    r.neu_pack = num("NFST_NEU_PACK", 1) != 0;
    r.neu_bf16 = num("NFST_NEU_BF16", 1) != 0;
    r.neu_small = getenv("NFST_NEU_NO_SMALL") ? 0 : 1;
    r.chunked = num("NFST_CHUNKED", 1) != 0;
    return r;
  }();
  return t;
}
int64_t lds_reserve() { return tuning().lds_reserve; }

// the precise flavour (float64 mantissas, semiring.h) runs all-compact batches whose deepest program has more than
// kPreciseTiles tiles, when at least four ring slots per sweep fit beside the 16-byte values
int precise_ring(const nfst_batch *lat, int64_t fixed, int n_rings) {
  const Tuning &tu = tuning();
  if (!(lat->reserved0 & NFST_BATCH_ALL_COMPACT) || tu.precise == 0) return 0;
  if (tu.precise != 1 && lat->max_tiles <= kPreciseTiles) return 0;
  const int64_t r = (kMaxLds - lds_reserve() - fixed) / ((int64_t)kSlotWordsP * 4 * n_rings);
  const int R = (int)(r > kMaxRing ? kMaxRing : r) & ~3;
  return R >= 4 ? R : 0;
}


// The chunked flavour (chunk_kernels.h): sweeps (one workgroup per lattice and direction), then posteriors / totals.  Returns
// the device flags of the lattices that have to be run again by the general kernels (tagged *tag).
int chunked_launch(const nfst_batch *lat, const nfst_scores *scores, int n_dirs, float *logalpha, float *logbeta, double *logz64,
                   float *logz32, float *posterior, float *grad_theta, float *beta_me, double *logz_total, int total_slot, hipStream_t st,
                   const int32_t **flags, int *tag) {
  nfst_chunks *ck = const_cast<nfst_chunks *>(lat->chunks);
  if (ck->n_lattices != lat->n_lattices || ck->total_rows != lat->total_rows || ck->total_arcs != lat->total_arcs || !ck->meta ||
      !ck->tab || !ck->stream || !ck->pos || !ck->label || !ck->ws || ck->ws_bytes < nfst_chunks_ws_bytes(ck) || ck->threads < 64 ||
      ck->threads > 1024 || (ck->threads & 63) || ck->lds_bytes <= 0 || ck->lds_bytes > kMaxLds || ((uintptr_t)ck->ws & 15))
    return NFST_ERR_ARG;
  ck->launches = ck->launches >= INT_MAX - 1 ? 1 : ck->launches + 1;
  *tag = ck->launches;
  *flags = chk_ws(*ck).flags;
  int rc;
  if ((rc = set_lds(k_chunk_sweep, ck->lds_bytes))) return rc;
  hipLaunchKernelGGL(k_chunk_sweep, dim3(lat->n_lattices * n_dirs), dim3(ck->threads), (size_t)ck->lds_bytes, st, *lat, *scores, *ck,
                     *tag, n_dirs, logalpha, logbeta, logz64, logz32, grad_theta, (float2 *)beta_me);
  if (n_dirs == 2 && (posterior || grad_theta || logz_total)) {
    const size_t lds = grad_theta ? (size_t)lat->vocab * 4 : 0;
    // workgroups of 256 threads, about four arcs per thread of the largest lattice: a slice of a lattice's arcs each
    const int64_t max_arcs = ((int64_t)lat->reserved0 >> NFST_BATCH_MAX_ARCS_SHIFT) & NFST_BATCH_MAX_ARCS_CAP;
    const int parts = (posterior || grad_theta) ? (int)std::max<int64_t>(1, std::min<int64_t>(64, (max_arcs + 1023) / 1024)) : 1;
    hipLaunchKernelGGL(k_chunk_post, dim3(lat->n_lattices * parts), dim3(256), lds, st, *lat, *scores, *ck, *tag, parts, posterior,
                       grad_theta, logz_total, total_slot);
  }
  return hip_status(hipGetLastError());
}

}  // namespace

extern "C" {

int nfst_device_available(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return n > 0 ? 1 : 0;
}

int nfst_tuning_set(const char *name, int value) {
  if (!name) return NFST_ERR_ARG;
  Tuning &t = tuning();
  const std::string n(name);
  if (n == "tw") t.tw = value != 0;
  else if (n == "fused") t.fused = value != 0;
  else if (n == "xcache") t.xcache = value != 0;
  else if (n == "precise") t.precise = value < 0 ? -1 : (value != 0);
  else if (n == "neu_pack") t.neu_pack = value != 0;
  else if (n == "neu_bf16") t.neu_bf16 = value != 0;
  else if (n == "neu_small") t.neu_small = value != 0;
  else if (n == "chunked") t.chunked = value != 0;
  else if (n == "lds_reserve_kb") t.lds_reserve = (value > 0 && value <= 96) ? (int64_t)value * 1024 : 0;
  else return NFST_ERR_ARG;
  return NFST_OK;
}

// ---------------------------------------------------------------- the packer on the device
int nfst_dense_to_arcs_count(const void *emission, int emission_is_float, const int64_t *transition, int32_t n_lattices,
                             int32_t n_rows, int32_t vocab, uint8_t *reach, int32_t *row_cnt, int32_t *counts,
                             int32_t *status, void *stream) {
  if (!emission || !transition || !reach || !row_cnt || !counts || !status || n_lattices <= 0 || n_rows <= 0 || vocab <= 0) return NFST_ERR_ARG;
  if (n_rows > NFST_MAX_ROWS || vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  const int64_t lds = (int64_t)n_rows * 12;
  int rc;
  if (emission_is_float) {
    if ((rc = set_lds(k_dense_reach<true>, lds))) return rc;
    hipLaunchKernelGGL(k_dense_reach<true>, dim3(n_lattices), dim3(kPkThreads), (size_t)lds, (hipStream_t)stream, emission, transition,
                       (int)n_rows, (int)vocab, reach, row_cnt, counts, status);
  } else {
    if ((rc = set_lds(k_dense_reach<false>, lds))) return rc;
    hipLaunchKernelGGL(k_dense_reach<false>, dim3(n_lattices), dim3(kPkThreads), (size_t)lds, (hipStream_t)stream, emission, transition,
                       (int)n_rows, (int)vocab, reach, row_cnt, counts, status);
  }
  return hip_status(hipGetLastError());
}

int nfst_dense_to_arcs_write(const void *emission, int emission_is_float, const int64_t *transition, int32_t n_lattices,
                             int32_t n_rows, int32_t vocab, const uint8_t *reach, const int32_t *row_cnt,
                             const int64_t *arc_off, int32_t *src, int32_t *label, int32_t *dst, float *arc_w, void *stream) {
  if (!emission || !transition || !reach || !row_cnt || !arc_off || !src || !label || !dst || n_lattices <= 0 || n_rows <= 0 || vocab <= 0)
    return NFST_ERR_ARG;
  if (emission_is_float && !arc_w) return NFST_ERR_ARG;
  if (n_rows > NFST_MAX_ROWS || vocab > NFST_MAX_VOCAB) return NFST_ERR_LIMIT;
  const int64_t lds = (int64_t)n_rows * 4;
  if (emission_is_float)
    hipLaunchKernelGGL(k_dense_write<true>, dim3(n_lattices), dim3(kPkThreads), (size_t)lds, (hipStream_t)stream, emission, transition,
                       (int)n_rows, (int)vocab, reach, row_cnt, arc_off, src, label, dst, arc_w);
  else
    hipLaunchKernelGGL(k_dense_write<false>, dim3(n_lattices), dim3(kPkThreads), (size_t)lds, (hipStream_t)stream, emission, transition,
                       (int)n_rows, (int)vocab, reach, row_cnt, arc_off, src, label, dst, arc_w);
  return hip_status(hipGetLastError());
}

int64_t nfst_pack_device_ws_bytes(int32_t n_lattices, int64_t total_rows, int64_t total_arcs) {
  if (n_lattices <= 0 || total_rows < 0 || total_arcs < 0) return NFST_ERR_ARG;
  return 4 * pk_ws_words(n_lattices, total_rows, total_arcs);
}

static int pack_device_args(const nfst_arcs_device *arcs, const nfst_pack_opts *opts, void *ws, int64_t ws_bytes, PkArgs *a) {
  if (!arcs || !ws || arcs->n_lattices <= 0 || arcs->vocab <= 0 || !arcs->n_rows || !arcs->row_off || !arcs->arc_off) return NFST_ERR_ARG;
  if (arcs->total_arcs > 0 && (!arcs->src || !arcs->label || !arcs->dst)) return NFST_ERR_ARG;
  if (arcs->vocab + 2 > 2048) return NFST_ERR_LIMIT;  // compact tiles only: the host packer takes wider vocabularies
  if (opts && ((opts->slots_per_lane != 0 && opts->slots_per_lane != 4) || opts->reserved1 == 1)) return NFST_ERR_LIMIT;
  if (ws_bytes < 4 * pk_ws_words(arcs->n_lattices, arcs->total_rows, arcs->total_arcs) || ((uintptr_t)ws & 15)) return NFST_ERR_ARG;
  *a = PkArgs{};
  a->n_rows = arcs->n_rows; a->row_off = arcs->row_off; a->arc_off = arcs->arc_off;
  a->src = arcs->src; a->label = arcs->label; a->dst = arcs->dst; a->w = arcs->arc_w;
  a->vocab = arcs->vocab; a->group_mode = opts ? opts->group_mode : 0;
  a->ws = (int32_t *)ws; a->total_rows = arcs->total_rows; a->total_arcs = arcs->total_arcs; a->n_lattices = arcs->n_lattices;
  return NFST_OK;
}

int nfst_pack_device_plan(const nfst_arcs_device *arcs, const nfst_pack_opts *opts, void *ws, int64_t ws_bytes, int32_t *meta,
                          int32_t *status, int32_t *scratch_rows, void *stream) {
  PkArgs a;
  int rc = pack_device_args(arcs, opts, ws, ws_bytes, &a);
  if (rc) return rc;
  if (!meta || !status || !scratch_rows) return NFST_ERR_ARG;
  a.meta = meta; a.status = status; a.scratch = scratch_rows;
  if ((rc = set_lds(k_pack_lattice<false>, kPkLdsBytes))) return rc;
  hipLaunchKernelGGL(k_pack_lattice<false>, dim3(arcs->n_lattices), dim3(kPkThreads), (size_t)kPkLdsBytes, (hipStream_t)stream, a);
  return hip_status(hipGetLastError());
}

int nfst_pack_device_emit(const nfst_arcs_device *arcs, const nfst_pack_opts *opts, void *ws, int64_t ws_bytes,
                          const int32_t *meta, int32_t *status, const nfst_batch *out, void *stream) {
  PkArgs a;
  int rc = pack_device_args(arcs, opts, ws, ws_bytes, &a);
  if (rc) return rc;
  if (!meta || !status || !out || out->n_lattices != arcs->n_lattices) return NFST_ERR_ARG;
  if ((rc = check_batch(out))) return rc;
  if (!out->arc_sd || !out->arc_l16 || (arcs->arc_w && !out->arc_w)) return NFST_ERR_ARG;
  if ((((uintptr_t)out->fwd_perm | (uintptr_t)out->bwd_perm) & 15)) return NFST_ERR_ARG;
  a.meta = const_cast<int32_t *>(meta); a.status = status; a.scratch = nullptr; a.out = *out;
  // the slack behind the streams and the 8 spare entries of the 6-byte arc arrays are part of the format: zero
  const int64_t slack = 512;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(const_cast<uint32_t *>(out->fwd_stream) + (out->fwd_words - slack), 0, slack * 4, st) != hipSuccess ||
      hipMemsetAsync(const_cast<uint32_t *>(out->bwd_stream) + (out->bwd_words - slack), 0, slack * 4, st) != hipSuccess ||
      hipMemsetAsync(const_cast<uint32_t *>(out->arc_sd) + out->total_arcs, 0, 8 * 4, st) != hipSuccess ||
      hipMemsetAsync(const_cast<uint16_t *>(out->arc_l16) + out->total_arcs, 0, 8 * 2, st) != hipSuccess)
    return NFST_ERR_HIP;
  if ((rc = set_lds(k_pack_lattice<true>, kPkLdsBytes))) return rc;
  hipLaunchKernelGGL(k_pack_lattice<true>, dim3(arcs->n_lattices), dim3(kPkThreads), (size_t)kPkLdsBytes, (hipStream_t)stream, a);
  return hip_status(hipGetLastError());
}

int64_t nfst_lds_bytes(const nfst_batch *lat) {
  if (!lat) return NFST_ERR_ARG;
  return LdsPlan(lat->max_rows, lat->vocab).fb_bytes(kMinRing, kRawSlotsShared, lat->weighted != 0);
}

// Ring sizes per sweep from the LDS budget of one workgroup, and which kernel flavour runs.
//   deep   (at most one lattice per CU): loader + decoder + sweep waves, deep staging ring;
//   shared (more lattices than CUs): the decoder loads for itself, shallow staging ring, and
//          two workgroups share a CU's 160 KiB when the lattices are small enough.
// A lattice too large for the deep rings runs the shared flavour with the whole CU.
// NFST_LDS_RESERVE_KB (environment, read once): LDS the one-lattice-per-CU flavour leaves free on
// every CU, so that a small kernel of another stream -- RCCL's all-reduce of the loss -- finds a CU
// to run on beside a sweep workgroup instead of waiting for one to retire.  Costs ring depth only.
struct RingCfg { int R, RS; bool self; };
static bool ring_config(const LdsPlan &plan, bool fb, bool extra, bool deep, RingCfg *c) {
  const int n_rings = fb ? 2 : 1;
  const int64_t slot = (int64_t)kSlotWords * 4 * n_rings;
  auto fixed = [&](int RS) { return fb ? plan.fb_bytes(0, RS, extra) : plan.bwd_bytes(0, RS, extra); };
  auto clampr = [](int64_t r) { return (int)(r > kMaxRing ? kMaxRing : r); };
  if (deep) {
    const int64_t r = (kMaxLds - lds_reserve() - fixed(kRawSlotsDeep)) / slot;
    if (r >= kMinRing) { *c = {clampr(r), kRawSlotsDeep, false}; return true; }
  } else {
    const int64_t r = (kMaxLds / 2 - fixed(kRawSlotsShared)) / slot;
    if (r >= kMinRing + 1) { *c = {clampr(r), kRawSlotsShared, true}; return true; }
  }
  const int64_t r = (kMaxLds - fixed(kRawSlotsShared)) / slot;
  if (r < kMinRing) return false;
  *c = {clampr(r), kRawSlotsShared, true};
  return true;
}

int nfst_backward(const nfst_batch *lat, const nfst_scores *scores, float *logbeta, double *logz64,
                  float *logz32, float *beta_me, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (lat->chunks && tuning().chunked) {
    // deep, narrow lattices: the chunked sweeps; lattices whose numbers leave their range are flagged on the device and
    // run by the general kernels below (a launch that finds no flag set returns at once)
    nfst_batch rest = *lat;
    rest.chunks = nullptr;
    if ((rc = chunked_launch(lat, scores, 1, nullptr, logbeta, logz64, logz32, nullptr, nullptr, beta_me, nullptr, 0, (hipStream_t)stream,
                             &rest.only, &rest.only_tag)))
      return rc;
    return nfst_backward(&rest, scores, logbeta, logz64, logz32, beta_me, stream);
  }
  const bool extra = (lat->weighted && lat->arc_w) || scores->arc_scores;
  const LdsPlan plan(lat->max_rows, lat->vocab);
  RingCfg cfg;
  if (!ring_config(plan, false, extra, lat->n_lattices <= cu_count(), &cfg)) return NFST_ERR_LIMIT;
  if (extra && ((uintptr_t)lat->bwd_perm & 15)) return NFST_ERR_ARG;  // the extras waves read the slot -> arc map 16 bytes at a time
  const bool both = lat->weighted && lat->arc_w && scores->arc_scores;
  if (const int Rp = precise_ring(lat, plan.bwd_fixed_precise(), 1)) {
    const int64_t ldsp = plan.bwd_fixed_precise() + (int64_t)Rp * kSlotWordsP * 4;
#define NFST_LAUNCH_BWD_P(EX)                                                                            \
    {                                                                                                  \
      if ((rc = set_lds(k_backward<512, EX, true, true>, ldsp))) return rc;                            \
      hipLaunchKernelGGL((k_backward<512, EX, true, true>), dim3(lat->n_lattices), dim3(512), (size_t)ldsp, \
                         (hipStream_t)stream, *lat, *scores, Rp, 0, logbeta, logz64, logz32, (float2 *)beta_me); \
    }
    if (both) NFST_LAUNCH_BWD_P(2) else if (extra) NFST_LAUNCH_BWD_P(1) else NFST_LAUNCH_BWD_P(0)
#undef NFST_LAUNCH_BWD_P
    return hip_status(hipGetLastError());
  }
  // one lattice per CU, all-compact: tile waves (NFST_TW=0: loader + decoder + sweep, for A/B runs)
  const bool tw = !cfg.self && (lat->reserved0 & NFST_BATCH_ALL_COMPACT) && tuning().tw;
  const int64_t tw_fixed = plan.bwd_bytes(0, 0, extra) + 512;  // + 64 x 8 bytes of trash for the non-leader lanes' stores
  if (tw) {
    const int64_t r = (kMaxLds - lds_reserve() - tw_fixed) / ((int64_t)kSlotWords2 * 4);
    cfg = {(int)(r > kMaxRing ? kMaxRing : r) & ~3, 0, false};
    if (cfg.R < 4) return NFST_ERR_LIMIT;
  }
  const int R = cfg.R, RS = cfg.RS;
  const int64_t lds = tw ? tw_fixed + (int64_t)R * kSlotWords2 * 4 : plan.bwd_bytes(R, RS, extra);
#define NFST_LAUNCH_BWD_TW(EX)                                                                           \
  {                                                                                                    \
    if ((rc = set_lds(k_backward<512, EX, true>, lds))) return rc;                                     \
    hipLaunchKernelGGL((k_backward<512, EX, true>), dim3(lat->n_lattices), dim3(512), (size_t)lds,     \
                       (hipStream_t)stream, *lat, *scores, R, RS, logbeta, logz64, logz32, (float2 *)beta_me); \
  }
#define NFST_LAUNCH_BWD(NT, EX)                                                                          \
  {                                                                                                    \
    if ((rc = set_lds(k_backward<NT, EX>, lds))) return rc;                                            \
    hipLaunchKernelGGL((k_backward<NT, EX>), dim3(lat->n_lattices), dim3(NT), (size_t)lds,             \
                       (hipStream_t)stream, *lat, *scores, R, RS, logbeta, logz64, logz32, (float2 *)beta_me); \
  }
  // 512 threads: loader + decoder + sweep (deep); 256 threads: self-loading decoder + sweep
  if (tw) { if (both) NFST_LAUNCH_BWD_TW(2) else if (extra) NFST_LAUNCH_BWD_TW(1) else NFST_LAUNCH_BWD_TW(0) }
  else if (!cfg.self) { if (both) NFST_LAUNCH_BWD(512, 2) else if (extra) NFST_LAUNCH_BWD(512, 1) else NFST_LAUNCH_BWD(512, 0) }
  else { if (both) NFST_LAUNCH_BWD(256, 2) else if (extra) NFST_LAUNCH_BWD(256, 1) else NFST_LAUNCH_BWD(256, 0) }
#undef NFST_LAUNCH_BWD
#undef NFST_LAUNCH_BWD_TW
  return hip_status(hipGetLastError());
}

int nfst_forward_backward(const nfst_batch *lat, const nfst_scores *scores, float *logalpha,
                          float *logbeta, double *logz64, float *logz32, float *posterior,
                          float *grad_theta, float *beta_me, double *logz_total, int32_t total_slot, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (posterior && ((uintptr_t)posterior & 15)) return NFST_ERR_ARG;
  if (logz_total && (total_slot < 0 || total_slot > 2)) return NFST_ERR_ARG;
  if (!lat->arc_sd || !lat->arc_l16 || ((uintptr_t)lat->arc_sd & 15) || ((uintptr_t)lat->arc_l16 & 7)) return NFST_ERR_ARG;
  if (lat->chunks && tuning().chunked) {  // (as in nfst_backward)
    nfst_batch rest = *lat;
    rest.chunks = nullptr;
    if ((rc = chunked_launch(lat, scores, 2, logalpha, logbeta, logz64, logz32, posterior, grad_theta, beta_me, logz_total, (int)total_slot,
                             (hipStream_t)stream, &rest.only, &rest.only_tag)))
      return rc;
    return nfst_forward_backward(&rest, scores, logalpha, logbeta, logz64, logz32, posterior, grad_theta, beta_me, logz_total, total_slot,
                                 stream);
  }
  const bool extra = (lat->weighted && lat->arc_w) || scores->arc_scores;
  const bool both = lat->weighted && lat->arc_w && scores->arc_scores;
  const LdsPlan plan(lat->max_rows, lat->vocab);
  const int cus = cu_count();
  RingCfg cfg;
  if (extra && (((uintptr_t)lat->fwd_perm | (uintptr_t)lat->bwd_perm | (uintptr_t)lat->arc_w | (uintptr_t)scores->arc_scores) & 15))
    return NFST_ERR_ARG;  // (maps and extras are read 16 bytes at a time)
  // deep programs: the precise flavour (tile waves with float64 mantissas), whatever the number of lattices
  if (const int Rp = precise_ring(lat, plan.fb_fixed_precise(), 2)) {
    const int64_t ldsp = plan.fb_fixed_precise() + (int64_t)Rp * kSlotWordsP * 4 * 2;
#define NFST_LAUNCH_TWP(EX)                                                                               \
    {                                                                                                   \
      if ((rc = set_lds(k_forward_backward<1024, EX, false, true, true>, ldsp))) return rc;             \
      hipLaunchKernelGGL((k_forward_backward<1024, EX, false, true, true>), dim3(lat->n_lattices), dim3(1024), (size_t)ldsp, \
                         (hipStream_t)stream, *lat, *scores, Rp, 0, logalpha, logbeta, logz64, logz32, logz_total,        \
                         (int)total_slot, posterior, grad_theta, (float2 *)beta_me);                    \
    }
    if (both) NFST_LAUNCH_TWP(2) else if (extra) NFST_LAUNCH_TWP(1) else NFST_LAUNCH_TWP(0)
#undef NFST_LAUNCH_TWP
    return hip_status(hipGetLastError());
  }
  // every program compact and no per-arc extras: the fused sweeps (no rings at all)
  // Measured (profiles/r02_ab_fused.txt): 227 against 164 G arcs/s at 1024 lattices, 210 against 160 at 2048,
  // equal at 512; with one lattice per CU the three-wave pipeline is 10 % faster (46.8 against 51.8 us).
  const bool fused = !extra && (lat->reserved0 & NFST_BATCH_ALL_COMPACT) && lat->n_lattices > cus && tuning().fused;
  // one lattice per CU: tile waves instead of loader + decoder (NFST_TW=0: the three-wave pipeline, for A/B runs)
  bool tw = false, cached = false;
  if (fused) cfg = {0, 0, lat->n_lattices > cus};
  else if (!ring_config(plan, true, extra, lat->n_lattices <= cus, &cfg)) return NFST_ERR_LIMIT;
  else if (!cfg.self && (lat->reserved0 & NFST_BATCH_ALL_COMPACT) && tuning().tw) {
    tw = true;  // no staging ring; ring slots of kSlotWords2 words
    const int64_t slot = (int64_t)kSlotWords2 * 4 * 2;
    const int64_t r = (kMaxLds - lds_reserve() - plan.fb_bytes(0, 0, extra)) / slot;
    cfg = {(int)(r > kMaxRing ? kMaxRing : r) & ~3, 0, false};  // (tile_sweep2 takes four tiles per trip: a multiple of four slots)
    if (cfg.R < 4) return NFST_ERR_LIMIT;
    // per-arc extras staged in LDS (the sum of both arrays, 4 bytes per arc of the largest lattice) when a ring of at least
    // eight slots per sweep still fits beside them (lattices up to ~14k arcs at 2k states); RS carries the room in floats
    const int64_t max_arcs = ((int64_t)lat->reserved0 >> NFST_BATCH_MAX_ARCS_SHIFT) & NFST_BATCH_MAX_ARCS_CAP;
    if (extra && max_arcs > 0 && max_arcs < NFST_BATCH_MAX_ARCS_CAP && tuning().xcache) {  // (NFST_XCACHE=0: gather from HBM / L2 instead, for A/B runs)
      const int64_t words = (max_arcs + 8 + 3) & ~(int64_t)3;
      const int64_t rc2 = ((kMaxLds - lds_reserve() - plan.fb_bytes(0, 0, extra) - words * 4) / slot) & ~(int64_t)3;
      if (rc2 >= 8) {  // (with four slots per sweep the tile waves cannot run ahead: 69 us against 50 from HBM / L2 at 256 x 20k arcs)
        cached = true;
        cfg = {(int)(rc2 > kMaxRing ? kMaxRing : rc2), (int)words, false};
      }
    }
  }
  const int R = cfg.R, RS = cfg.RS;
  const int64_t lds = tw ? plan.fb_bytes(0, 0, extra) + (int64_t)R * kSlotWords2 * 4 * 2 + (cached ? (int64_t)RS * 4 : 0) : plan.fb_bytes(R, RS, extra);
#define NFST_LAUNCH_FB(NT, EX)                                                                            \
  {                                                                                                     \
    if ((rc = set_lds(k_forward_backward<NT, EX>, lds))) return rc;                                     \
    hipLaunchKernelGGL((k_forward_backward<NT, EX>), dim3(lat->n_lattices), dim3(NT), (size_t)lds,      \
                       (hipStream_t)stream, *lat, *scores, R, RS, logalpha, logbeta, logz64, logz32, logz_total,    \
                       (int)total_slot, posterior,                                                      \
                       grad_theta, (float2 *)beta_me);                                                  \
  }
#define NFST_LAUNCH_TW(EX)                                                                                 \
  {                                                                                                     \
    if ((rc = set_lds(k_forward_backward<1024, EX, false, true>, lds))) return rc;                      \
    hipLaunchKernelGGL((k_forward_backward<1024, EX, false, true>), dim3(lat->n_lattices), dim3(1024), (size_t)lds, \
                       (hipStream_t)stream, *lat, *scores, R, RS, logalpha, logbeta, logz64, logz32, logz_total,    \
                       (int)total_slot, posterior, grad_theta, (float2 *)beta_me);                      \
  }
#define NFST_LAUNCH_FUSED(NT)                                                                            \
  {                                                                                                     \
    if ((rc = set_lds(k_forward_backward<NT, 0, true>, lds))) return rc;                                \
    hipLaunchKernelGGL((k_forward_backward<NT, 0, true>), dim3(lat->n_lattices), dim3(NT), (size_t)lds,     \
                       (hipStream_t)stream, *lat, *scores, R, RS, logalpha, logbeta, logz64, logz32, logz_total,    \
                       (int)total_slot, posterior, grad_theta, (float2 *)beta_me);                      \
  }
  // 1024 threads: loaders + decoders + sweeps and 10 more waves for the posterior pass (deep);
  // 512 / 256 threads: self-loading decoders + sweeps, two workgroups per CU when they fit
  if (fused) {
    if (lds > kMaxLds) return NFST_ERR_LIMIT;
    // (512 threads at most: with 1024 the 128 registers a lane may have leave hipcc 64 VGPRs beside the
    // fused sweep's 32 AGPRs, and it then spills into AGPRs -- into the ones the sweep stages tiles in)
    if (lat->n_lattices <= 2 * cus) NFST_LAUNCH_FUSED(512)
    else NFST_LAUNCH_FUSED(256)
  } else if (!cfg.self && tw) {
    if (cached) NFST_LAUNCH_TW(3) else if (both) NFST_LAUNCH_TW(2) else if (extra) NFST_LAUNCH_TW(1) else NFST_LAUNCH_TW(0)
  } else if (!cfg.self) {
    if (both) NFST_LAUNCH_FB(1024, 2) else if (extra) NFST_LAUNCH_FB(1024, 1) else NFST_LAUNCH_FB(1024, 0)
  }
  else if (both) NFST_LAUNCH_FB(512, 2)  // (the weight waves are waves 4 .. 7)
  else if (extra) NFST_LAUNCH_FB(512, 1)
  else if (lat->n_lattices <= 2 * cus) NFST_LAUNCH_FB(512, 0)
  else NFST_LAUNCH_FB(256, 0)
#undef NFST_LAUNCH_FB
#undef NFST_LAUNCH_TW
#undef NFST_LAUNCH_FUSED
  return hip_status(hipGetLastError());
}

int nfst_viterbi(const nfst_batch *lat, const nfst_scores *scores, float *best, int32_t *paths,
                 int32_t *path_arcs, int32_t *lengths, int32_t max_len, int32_t pad, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if ((rc = check_scores(lat, scores))) return rc;
  if (!best || !paths || !lengths || max_len <= 0) return NFST_ERR_ARG;
  // all-compact batches: the tile-wave kernel (NFST_TW=0: one wave reading the program from global memory, for A/B runs)
  const bool extra = (lat->weighted && lat->arc_w) || scores->arc_scores;
  const bool both = lat->weighted && lat->arc_w && scores->arc_scores;
  const int64_t tw_fixed = VitLds(lat->max_rows, lat->vocab).fixed();
  int64_t tw_r = (kMaxLds - tw_fixed) / ((int64_t)kSlotWords2 * 4);
  tw_r = (tw_r > kMaxRing ? kMaxRing : tw_r) & ~(int64_t)3;
  if ((lat->reserved0 & NFST_BATCH_ALL_COMPACT) && tw_r >= 8 && tuning().tw &&  // (its trips check four tiles ahead: eight slots)
      (!extra || (((uintptr_t)lat->arc_w | (uintptr_t)scores->arc_scores) & 3) == 0) && ((uintptr_t)lat->bwd_perm & 15) == 0) {
    const int64_t lds = tw_fixed + tw_r * kSlotWords2 * 4;
#define NFST_LAUNCH_VIT(XM)                                                                                          \
    {                                                                                                                \
      if ((rc = set_lds(k_viterbi_tw<XM>, lds))) return rc;                                                          \
      hipLaunchKernelGGL(k_viterbi_tw<XM>, dim3(lat->n_lattices), dim3(kVitTwThreads), (size_t)lds, (hipStream_t)stream, \
                         *lat, *scores, (int)tw_r, best, paths, path_arcs, lengths, (int)max_len, (int)pad);        \
    }
    if (both) NFST_LAUNCH_VIT(2) else if (extra) NFST_LAUNCH_VIT(1) else NFST_LAUNCH_VIT(0)
#undef NFST_LAUNCH_VIT
    return hip_status(hipGetLastError());
  }
  const int64_t lds = (int64_t)lat->max_rows * 12 + (int64_t)lat->vocab * 4 + 16;
  if ((rc = set_lds(k_viterbi, lds))) return rc;
  hipLaunchKernelGGL(k_viterbi, dim3(lat->n_lattices), dim3(kVitThreads), (size_t)lds, (hipStream_t)stream, *lat,
                     *scores, best, paths, path_arcs, lengths, (int)max_len, (int)pad);
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
  if (!lat->arc_sd || !lat->arc_l16) return NFST_ERR_ARG;
  const int stage_theta = (int64_t)lat->max_rows * 8 + (int64_t)lat->vocab * 4 <= 96 * 1024;
  const int64_t lds_min = (int64_t)lat->max_rows * 8 + (stage_theta ? (int64_t)((lat->vocab + 3) & ~3) * 4 : 0);
  // room for a lattice's CSR (row pointers + 6 bytes per arc; a program's slots bound its arcs): the kernel stages it
  // when it fits what it was given
  const int64_t csr = ((int64_t)lat->max_rows + 8) * 4 + ((int64_t)lat->max_tiles * 256 + 8) * 6;
  const int64_t lds = lds_min + csr <= kMaxLds ? lds_min + csr : (lds_min + 64 * 1024 <= kMaxLds ? kMaxLds : lds_min);
  if ((rc = set_lds(k_sample, lds))) return rc;
  // 16 walks per 256 threads; up to 64 walks (1024 threads) of a lattice in one block share its staged data
  // (with the arcs' probabilities precomputed per block -- path_arcs given and the CSR fits -- every block has 1024
  // threads for that pass, whatever k)
  const bool precdf = path_arcs && lds > lds_min;  // (the kernel decides per lattice, from its own arc count)
  const int walks = (k >= 64 || precdf) ? 64 : ((k + 15) / 16) * 16;
  hipLaunchKernelGGL(k_sample, dim3(lat->n_lattices, (k + walks - 1) / walks), dim3(walks * 16),
                     (size_t)lds, (hipStream_t)stream,
                     *lat, *scores, (const float2 *)beta_me, logz64, (int)k, (int)max_len, uniforms,
                     seed, (int)pad, stage_theta, (int)lds, paths, path_arcs, lengths, logq, status);
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
  if ((rc = set_lds(k_row_gather<0>, (int64_t)lat->vocab * 4))) return rc;
  hipLaunchKernelGGL(k_row_gather<0>, dim3((unsigned)n), dim3(64), (size_t)lat->vocab * 4, (hipStream_t)stream, *lat, state,
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
  if ((rc = set_lds(k_row_gather<1>, (int64_t)lat->vocab * 4))) return rc;
  hipLaunchKernelGGL(k_row_gather<1>, dim3((unsigned)n), dim3(64), (size_t)lat->vocab * 4, (hipStream_t)stream, *lat, state,
                     values, (const int64_t *)nullptr, 0, 0, 0, 0, out, (int)k);
  return hip_status(hipGetLastError());
}

int nfst_proposal_step(const nfst_batch *lat, const int64_t *state, const int64_t *inp, const float *scores,
                       const float *values, int32_t pad, int32_t bos, int32_t eos, int32_t has_to_end, float temperature,
                       const float *uniforms, const int64_t *forced, const nfst_step_extras *extras, int64_t *symbol,
                       float *logq, float *logz, int64_t *next_state, float *logits_out, int32_t k, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!state || !scores || !symbol || !logq || !next_state || k <= 0 || !(temperature > 0.0f)) return NFST_ERR_ARG;
  if (!uniforms && !forced) return NFST_ERR_ARG;
  if (pad < 0 || pad >= lat->vocab) return NFST_ERR_ARG;
  if (lat->vocab > kStepMaxVocab) return NFST_ERR_LIMIT;
  nfst_step_extras ex = {};
  if (extras) {
    ex = *extras;
    if ((ex.accumulated || ex.vocab_use) && (ex.insertion_mark < 0 || ex.insertion_mark >= lat->vocab || ex.length < 1))
      return NFST_ERR_ARG;
  }
  const int64_t n = (int64_t)lat->n_lattices * k;
  const int64_t lds = (int64_t)kStepWaves * ((values && ex.value_state) ? 3 : 2) * lat->vocab * 4;
  if ((rc = set_lds(k_proposal_step, lds))) return rc;
  hipLaunchKernelGGL(k_proposal_step, dim3((unsigned)((n + kStepWaves - 1) / kStepWaves)), dim3(64 * kStepWaves), (size_t)lds,
                     (hipStream_t)stream, *lat, state, inp, scores, values, (int)pad, (int)bos, (int)eos, (int)has_to_end,
                     temperature, uniforms, forced, ex, symbol, logq, logz, next_state, logits_out, (int)k, n);
  return hip_status(hipGetLastError());
}

int nfst_proposal_step_backward(const nfst_batch *lat, const int64_t *value_state, const float *logits,
                                const int64_t *symbol, const float *logz, const float *g_logq, const float *g_logz,
                                int32_t pad, float temperature, float *grad_scores, float *grad_values, int32_t k,
                                void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!logits || !symbol || !logz || !grad_scores || k <= 0 || !(temperature > 0.0f)) return NFST_ERR_ARG;
  if (grad_values && !value_state) return NFST_ERR_ARG;
  if (!g_logq && !g_logz) return NFST_ERR_ARG;
  const int64_t n = (int64_t)lat->n_lattices * k;
  hipLaunchKernelGGL(k_proposal_step_bwd, dim3((unsigned)((n + kStepWaves - 1) / kStepWaves)), dim3(64 * kStepWaves), 0,
                     (hipStream_t)stream, *lat, value_state, logits, symbol, logz, g_logq, g_logz, (int)pad, temperature,
                     grad_scores, grad_values, (int)k, n);
  return hip_status(hipGetLastError());
}

int64_t nfst_neural_ws_floats(const nfst_batch *lat, int32_t hid) {
  if (!lat || hid <= 0) return NFST_ERR_ARG;
  // u and beta_hat planes, (mantissa, exponent) rows, and Wh as three bfloat16 parts in MFMA fragment order (k_pack_mfma_b3)
  return neu_pack_off(2 * (int64_t)lat->n_lattices * lat->max_rows * (hid + 1)) + 2 * (int64_t)hid * hid;  // (+ Wh: float32 or 3 x bfloat16)
}

int nfst_backward_neural(const nfst_batch *lat, const float *label_x, const float *wh, const float *w, int32_t hid,
                         float *log_beta, float *beta_hat, float *ws, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!label_x || !wh || !w || !log_beta || !beta_hat || !ws || hid <= 0) return NFST_ERR_ARG;
  if (hid > kNeuMaxHid) return NFST_ERR_LIMIT;
  const int64_t lds = NeuLds(lat->max_rows, hid).bytes();
    // phase B on three bfloat16 parts when hid is a multiple of 64 from 256 on (neu_pack = 0 or neu_bf16 = 0: float32 MFMAs from the matrix itself)
  const int wh_packed = tuning().neu_pack && tuning().neu_bf16 && hid % 64 == 0 && hid >= 256;  // (below 256 the split's two barriers per pass cost more than the matrix pipe gains: H = 128 1.41 against 1.39 ms)
  float *wh_ws = ws + neu_pack_off(2 * (int64_t)lat->n_lattices * lat->max_rows * (hid + 1));
  if (wh_packed)
    hipLaunchKernelGGL(k_pack_mfma_b3, dim3(((hid >> 4) * (hid >> 5) * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, wh, (int)hid,
                       reinterpret_cast<uint4 *>(wh_ws));
#define NFST_LAUNCH_NEU(HC)                                                                                   \
  do {                                                                                                        \
    if ((rc = set_lds(k_backward_neural<HC>, lds))) return rc;                                                \
    hipLaunchKernelGGL(k_backward_neural<HC>, dim3(lat->n_lattices), dim3(kNeuThreads), (size_t)lds,          \
                       (hipStream_t)stream, *lat, label_x, wh, w, (int)hid, log_beta, beta_hat, ws, wh_packed); \
  } while (0)
#define NFST_LAUNCH_NEU_SMALL(LPR)                                                                             \
  do {                                                                                                        \
    if ((rc = set_lds(k_backward_neural_small<LPR>, lds))) return rc;                                         \
    hipLaunchKernelGGL(k_backward_neural_small<LPR>, dim3(lat->n_lattices), dim3(kNeuThreads), (size_t)lds,   \
                       (hipStream_t)stream, *lat, label_x, wh, w, (int)hid, log_beta, beta_hat, ws);        \
  } while (0)
  const int no_small = !tuning().neu_small;  // (A/B against the two-phase kernel)
  // BASELINE batch, whole op: H = 8 0.52 against 1.19 ms, 16 0.58 / 1.18, 32 1.07 / 1.19; with a whole wave per
  // record (H = 64) the packed kernel has nothing to pack and loses to the two-phase one: 1.97 / 1.24
  if (hid <= 8 && !no_small) NFST_LAUNCH_NEU_SMALL(8);
  else if (hid <= 16 && !no_small) NFST_LAUNCH_NEU_SMALL(16);
  else if (hid <= 32 && !no_small) NFST_LAUNCH_NEU_SMALL(32);
  else if (hid <= 64) NFST_LAUNCH_NEU(1);
  else if (hid <= 128) NFST_LAUNCH_NEU(2);
  else if (hid <= 256) NFST_LAUNCH_NEU(4);
  else NFST_LAUNCH_NEU(8);
#undef NFST_LAUNCH_NEU
#undef NFST_LAUNCH_NEU_SMALL
  return hip_status(hipGetLastError());
}

#ifdef NFST_PROF
extern "C" int nfst_prof_read(unsigned long long *out, int n) {  // profiling build only
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(fb_prof), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif
#ifdef NFST_PK_STAMPS
extern "C" int nfst_debug_pk_stamps(unsigned long long *out) {  // profiling build only
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(pk_stamps), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : -1;
}
#endif
#ifdef NFST_NEU_STAMPS
extern "C" int nfst_debug_neu_stamps(unsigned long long *out, int reset) {  // profiling build only
  if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(neu_stamps), sizeof(unsigned long long) * 128) != hipSuccess) return -1;
  if (reset) { unsigned long long z[128] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(neu_stamps), z, sizeof(z)) != hipSuccess) return -1; }
  return 0;
}
#endif

int64_t nfst_neural_grad_ws_floats(const nfst_batch *lat, int32_t hid) {
  if (!lat || hid <= 0) return NFST_ERR_ARG;
  return neu_pack_off(2 * (int64_t)lat->n_lattices * lat->max_rows * hid) + 2 * (int64_t)hid * hid;  // ... + Wh^T in fragment order (float32 or 3 x bfloat16)
}

int nfst_backward_neural_grad(const nfst_batch *lat, const float *label_x, const float *wh_t, const float *w, int32_t hid,
                              const float *beta_hat, const float *ws_fwd, const float *g_log_beta, const float *g_beta_hat,
                              float *gamma, float *grad_label_x, float *grad_w, float *ws, void *stream) {
  int rc = check_batch(lat);
  if (rc) return rc;
  if (!label_x || !wh_t || !w || !beta_hat || !ws_fwd || !g_log_beta || !gamma || !grad_label_x || !grad_w || !ws || hid <= 0)
    return NFST_ERR_ARG;
  if (hid > kNeuMaxHid) return NFST_ERR_LIMIT;
  if (lat->fwd_slots > 0 && !lat->fwd_perm) return NFST_ERR_ARG;
  const int64_t lds = NeuGradLds(lat->max_rows, hid).bytes();
  // phase B on three bfloat16 parts when hid is a multiple of 64 from 256 on (neu_pack = 0 or neu_bf16 = 0: float32 MFMAs from the matrix itself)
  const int wh_packed = tuning().neu_pack && tuning().neu_bf16 && hid % 64 == 0 && hid >= 256;  // (below 256 the split's two barriers per pass cost more than the matrix pipe gains: H = 128 1.41 against 1.39 ms)
  float *wh_ws = ws + neu_pack_off(2 * (int64_t)lat->n_lattices * lat->max_rows * hid);
  if (wh_packed)
    hipLaunchKernelGGL(k_pack_mfma_b3, dim3(((hid >> 4) * (hid >> 5) * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, wh_t, (int)hid,
                       reinterpret_cast<uint4 *>(wh_ws));
#define NFST_LAUNCH_NEUG(HC)                                                                                       \
  do {                                                                                                             \
    if ((rc = set_lds(k_backward_neural_grad<HC>, lds))) return rc;                                                \
    hipLaunchKernelGGL(k_backward_neural_grad<HC>, dim3(lat->n_lattices), dim3(kNeuThreads), (size_t)lds,          \
                       (hipStream_t)stream, *lat, label_x, wh_t, w, (int)hid, beta_hat, ws_fwd, g_log_beta,         \
                       g_beta_hat, gamma, grad_label_x, grad_w, ws, wh_packed);                                    \
  } while (0)
  // (the packed kernel sums dL/dx per lattice in LDS when [V, hid] floats fit beside its rows and staged tiles)
  const int64_t lds_small0 = (int64_t)neu_rows_al(lat->max_rows) * 8 + (int64_t)((lat->max_rows + 3) & ~3) * 4 + 2 * kNeuGradStageWords * 4 + 16;
  // (BASELINE batch, whole gradient op: H = 8 1.34 ms with the LDS table against 1.72 with global atomics; H = 16 1.80
  // against 1.56, H = 32 3.13 against 2.56 -- the table is contended only when its rows are a few lanes wide)
  const int gx_in_lds = hid <= 8 && lds_small0 + (int64_t)lat->vocab * hid * 4 <= kMaxLds;
  const int64_t lds_small = lds_small0 + (gx_in_lds ? (int64_t)lat->vocab * hid * 4 : 0);
#define NFST_LAUNCH_NEUG_SMALL(LPR)                                                                               \
  do {                                                                                                             \
    if ((rc = set_lds(k_backward_neural_grad_small<LPR>, lds_small))) return rc;                                   \
    hipLaunchKernelGGL(k_backward_neural_grad_small<LPR>, dim3(lat->n_lattices), dim3(kNeuThreads), (size_t)lds_small, \
                       (hipStream_t)stream, *lat, label_x, wh_t, w, (int)hid, beta_hat, ws_fwd, g_log_beta,         \
                       g_beta_hat, gamma, grad_label_x, grad_w, ws, gx_in_lds);                                    \
  } while (0)
  const int no_small = !tuning().neu_small;  // (A/B against the two-phase kernel)
  if (hid <= 8 && !no_small) NFST_LAUNCH_NEUG_SMALL(8);
  else if (hid <= 16 && !no_small) NFST_LAUNCH_NEUG_SMALL(16);
  else if (hid <= 32 && !no_small) NFST_LAUNCH_NEUG_SMALL(32);
  else if (hid <= 64) NFST_LAUNCH_NEUG(1);
  else if (hid <= 128) NFST_LAUNCH_NEUG(2);
  else if (hid <= 256) NFST_LAUNCH_NEUG(4);
  else NFST_LAUNCH_NEUG(8);
#undef NFST_LAUNCH_NEUG
#undef NFST_LAUNCH_NEUG_SMALL
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

// shared argument checks and the variant table of the 16-byte streaming kernels: a row lies on a
// quarter wave (up to 128 slots of 16 bytes, 8 per lane) or a half wave, four or two rows side by
// side: the per-row instructions (two reductions, masks) are shared by the rows of a wave, and the
// slots a row wastes are at most 15 / 31.
static int plp_check(const void *scores, const void *marks, int64_t n, int32_t t, int32_t vocab, float temp,
                     float smoothing, int32_t mask_mode) {
  if (!scores || !marks || n <= 0 || t <= 0 || vocab <= 0 || !(temp > 0.0f)) return NFST_ERR_ARG;
  if (!(smoothing >= 0.0f && smoothing < 1.0f)) return NFST_ERR_ARG;  // scorers.py:1514
  if (mask_mode != NFST_MASK_STATICRNN && mask_mode != NFST_MASK_GPT2) return NFST_ERR_ARG;
  if (n > 0x7fffffffll) return NFST_ERR_LIMIT;
  return NFST_OK;
}
static int plp_variant(const void *p0, const void *p1, int32_t vocab) {
  const bool v4 = vocab % 4 == 0 && (((uintptr_t)p0 | (uintptr_t)p1) & 15) == 0 && vocab <= 1024;
  if (!v4) return 0;
  const int f4 = vocab / 4;
  return f4 <= 128 ? (f4 + 15) / 16 : 8 + (f4 + 31) / 32;
}
#define NFST_PLP_SWITCH(u, LAUNCH, FALLBACK) \
  switch (u) {                               \
    case 0: FALLBACK; break;                 \
    case 1: LAUNCH(1, 8, 16); break;         \
    case 2: LAUNCH(2, 4, 16); break;         \
    case 3: LAUNCH(3, 4, 16); break;         \
    case 4: LAUNCH(4, 2, 16); break;         \
    case 5: LAUNCH(5, 2, 16); break;         \
    case 6: LAUNCH(6, 2, 16); break;         \
    case 7: LAUNCH(7, 1, 16); break;         \
    case 8: LAUNCH(8, 1, 16); break;         \
    case 13: LAUNCH(5, 2, 32); break; /* 129 .. 160 slots */ \
    case 14: LAUNCH(6, 2, 32); break;        \
    case 15: LAUNCH(7, 1, 32); break;        \
    default: LAUNCH(8, 1, 32); break; /* 16: up to 256 slots */ \
  }

int nfst_path_logprob(const float *scores, const int64_t *marks, int64_t n, int32_t t, int32_t vocab,
                      int32_t pad, int32_t bos, int32_t eos, int32_t max_length, float temp,
                      int32_t normalize, float smoothing, int32_t mask_mode, float *out, void *stream) {
  int rc = plp_check(scores, marks, n, t, vocab, temp, smoothing, mask_mode);
  if (rc) return rc;
  if (!out) return NFST_ERR_ARG;
#define NFST_LAUNCH_PLP(NV, RB, L)                                                                         \
  hipLaunchKernelGGL((k_path_logprob_v4<NV, RB, L>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,  \
                     scores, marks, (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length,      \
                     temp, (int)normalize, smoothing, (int)mask_mode, out)
  NFST_PLP_SWITCH(plp_variant(scores, nullptr, vocab), NFST_LAUNCH_PLP,
                  hipLaunchKernelGGL(k_path_logprob, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, scores, marks,
                                     (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length, temp,
                                     (int)normalize, smoothing, (int)mask_mode, out))
#undef NFST_LAUNCH_PLP
  return hip_status(hipGetLastError());
}

int nfst_path_logprob_backward(const float *scores, const int64_t *marks, const float *grad_out, int64_t n, int32_t t,
                               int32_t vocab, int32_t pad, int32_t bos, int32_t eos, int32_t max_length, float temp,
                               int32_t normalize, float smoothing, int32_t mask_mode, float *grad_scores, void *stream) {
  int rc = plp_check(scores, marks, n, t, vocab, temp, smoothing, mask_mode);
  if (rc) return rc;
  if (!grad_out || !grad_scores) return NFST_ERR_ARG;
#define NFST_LAUNCH_PLPB(NV, RB, L)                                                                            \
  hipLaunchKernelGGL((k_path_logprob_bwd_v4<NV, RB, L>), dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream,  \
                     scores, marks, grad_out, (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length, \
                     temp, (int)normalize, smoothing, (int)mask_mode, grad_scores)
  NFST_PLP_SWITCH(plp_variant(scores, grad_scores, vocab), NFST_LAUNCH_PLPB,
                  hipLaunchKernelGGL(k_path_logprob_bwd, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, scores, marks,
                                     grad_out, (int)t, (int)vocab, (int)pad, (int)bos, (int)eos, (int)max_length, temp,
                                     (int)normalize, smoothing, (int)mask_mode, grad_scores))
#undef NFST_LAUNCH_PLPB
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
