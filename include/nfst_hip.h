/*
 * nfst_hip.h -- C ABI of the MI355X (gfx950) lattice engine for nFST.
 *
 * Drop-in boundary for the reference's lattice hot path.
 * The reference (steventan0110/nFST) is pure Python and has no FFI; the entry
 * points below are what a ctypes binding for that path binds, one per Python
 * call site it replaces (file:line into /root/reference/src):
 *
 *   nfst_pack_dense / nfst_pack_arcs   FSAGRUScorer.set_masks + set_k
 *                                      (modules/scorers.py:877-918), table format
 *                                      of get_state_mask_pynini (995-1035) and of
 *                                      the .npz files (preprocess/tr.py:182-190)
 *   nfst_backward                      FSAGRUScorer.compute_beta
 *                                      (scorers.py:692-751, 753-875)
 *   nfst_forward_backward              exact log-Z / alpha / arc posteriors: the
 *                                      quantity Estimators.iwae estimates
 *                                      (modules/estimatros.py:33-44), cf.
 *                                      JointProb.log_marginalize
 *                                      (modules/lightning.py:408-440)
 *   nfst_viterbi                       best_sample of JointProb.forward
 *                                      (lightning.py:474-479)
 *   nfst_sample_paths                  Sampler.sample / stateful_sample
 *                                      (modules/samplers.py:137-335)
 *   nfst_score_paths                   forced scoring (samplers.py:208-218)
 *   nfst_step                          FSAGRUScorer.update_fsa_state
 *                                      (scorers.py:683-690)
 *   nfst_emission_mask                 FSAGRUScorer.mask_out_invalid
 *                                      (scorers.py:1037-1054)
 *   nfst_beta_logits                   beta-logit gather (scorers.py:581-593)
 *   nfst_proposal_step (+ _backward)   one step of Sampler.stateful_sample on the lattice
 *                                      side (samplers.py:243-297; scorers.py:340-366, 630-690)
 *   nfst_backward_neural               compute_beta with Wh != 0 (scorers.py:692-856)
 *   nfst_backward_neural_grad          its gradient (tune_proposal, lightning.py:339-406)
 *   nfst_gather_label_scores           WFSTScorer (scorers.py:1671-1687)
 *   nfst_path_logprob (+ _backward)    StaticRNNScorer.evaluate_seq_with_temp
 *                                      gather (scorers.py:1564-1611) and its gradient
 *                                      (lightning.py:511-516 trains through it),
 *                                      GPT2Wrapper.forward (transformer.py:45-52)
 *   nfst_iwae                          Estimators.iwae (estimatros.py:11-44)
 *
 * Conventions
 *   - plain C: pointers and sizes only, no C++/torch types.
 *   - "device" pointers are HIP device memory owned by the caller (the Python
 *     host side passes torch tensors' data_ptr()); "host" pointers are ordinary
 *     memory.  The library allocates device memory nowhere; every launch goes to
 *     the hipStream_t passed as `stream` (void*) on the calling thread's current
 *     device.  Its only process state is a mutex-guarded cache, keyed by device,
 *     of the kernels' dynamic-LDS opt-ins and of the CU count: several devices in
 *     one process and launches from several host threads are safe.
 *   - every function returns NFST_OK (0) or a negative error code; the message
 *     is available from nfst_strerror().  Kernels are never launched on invalid
 *     shapes: all operands are validated on the host first.
 *   - state, arc and label indices are int32 and bit-exact with the reference's
 *     int64 tables; floating point is float32 unless stated (log Z also float64).
 */
#ifndef NFST_HIP_H
#define NFST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NFST_ABI_VERSION 7

/* error codes */
#define NFST_OK 0
#define NFST_ERR_ARG -1        /* null pointer / bad size */
#define NFST_ERR_INDEX -2      /* state or label index out of range */
#define NFST_ERR_CYCLE -3      /* lattice is not acyclic */
#define NFST_ERR_SINK -4       /* not exactly one final (sink) state */
#define NFST_ERR_DETERMINISM -5/* two arcs with the same (state, label) */
#define NFST_ERR_LIMIT -6      /* lattice exceeds the engine's limits */
#define NFST_ERR_HIP -7        /* HIP runtime error */
#define NFST_ERR_NOMEM -8
#define NFST_ERR_LENGTH -9     /* "ran out of length budget" (samplers.py:299-302) */

/* limits of the LDS-resident kernels */
#define NFST_MAX_ROWS 8192     /* states per lattice incl. sink (alpha+beta live in LDS) */
#define NFST_MAX_VOCAB 32767   /* labels + the null label are packed in 16 bits of an arc record */

/* per-lattice metadata: NFST_META_WORDS int32 each, see DESIGN.md */
#define NFST_META_WORDS 16
#define NFST_META_ROW_OFF 0    /* first row of this lattice in row-indexed arrays */
#define NFST_META_N_ROWS 1     /* rows of this lattice (S+1; padded rows included) */
#define NFST_META_ARC_OFF 2    /* first canonical arc */
#define NFST_META_N_ARCS 3     /* canonical arcs (incl. self loops) */
#define NFST_META_FWD_OFF 4    /* word offset of the alpha (by-destination) tile program */
#define NFST_META_FWD_TILES 5
#define NFST_META_BWD_OFF 6    /* word offset of the beta (by-source) tile program */
#define NFST_META_BWD_TILES 7
#define NFST_META_SINK 8       /* sink state id */
#define NFST_META_N_REACH 9    /* states reachable from 0 */
#define NFST_META_DEPTH 10     /* longest path, in arcs */
#define NFST_META_N_DP 11      /* non-self-loop arcs */
#define NFST_META_FWD_U 12     /* alpha program format: bits [0:8) 1, 2 or 4 arc slots per lane in 32-bit records,
                                  or 8 = four 24-bit records per lane; bit 8 = wide groups (up to 64 lanes) */
#define NFST_META_BWD_U 13
#define NFST_META_FWD_SLOT_OFF 14 /* first slot of this lattice in fwd_perm */
#define NFST_META_BWD_SLOT_OFF 15

/*
 * A packed batch of lattices.  All pointers are host memory when returned by
 * the packer and device memory when handed to a kernel entry point (the host
 * side copies the arrays once; K samples per lattice share them, K is a launch
 * parameter -- scorers.py:887-918 materialises K copies instead).
 *
 * Canonical arrays (user-facing): arcs of the states
 * reachable from state 0 in (state asc, label asc) order; values bit-exact with
 * nonzero(emission) / transition.  The sink's pad self loop is included.
 * Tile programs (engine-private): the level-scheduled sweeps, DESIGN.md section 3.
 */
/* nfst_batch.reserved0 bit: every tile program of the batch is in the compact format (code 8): the
 * fused sweep kernels apply (set by the packer; LatticeBatch.concat keeps it if all parts have it) */
#define NFST_BATCH_ALL_COMPACT 1
/* nfst_batch.reserved0 bits 8 .. 30: the largest number of canonical arcs of one lattice of the batch (capped: the cap
 * means "unknown, large"); the launchers size LDS-resident per-arc data with it */
#define NFST_BATCH_MAX_ARCS_SHIFT 8
#define NFST_BATCH_MAX_ARCS_CAP 0x7fffff

typedef struct nfst_batch {
  int32_t n_lattices;
  int32_t vocab;
  int32_t max_rows;        /* max over the batch of the LDS rows a lattice needs: n_rows + scratch rows of its tile programs */
  int32_t max_tiles;       /* max tiles of one program over the batch */
  int32_t weighted;        /* arc_w holds the table's float weights */
  int32_t reserved0;       /* NFST_BATCH_ALL_COMPACT | largest arc count of one lattice << NFST_BATCH_MAX_ARCS_SHIFT */
  int64_t total_rows;
  int64_t total_arcs;
  int64_t total_dp_arcs;
  int64_t fwd_words;
  int64_t bwd_words;
  int64_t fwd_slots;
  int64_t bwd_slots;
  const int32_t *meta;       /* [n_lattices * NFST_META_WORDS] */
  const int32_t *row_ptr;    /* [total_rows + n_lattices] absolute arc indices */
  const int32_t *arc_src;    /* [total_arcs] */
  const int32_t *arc_dst;    /* [total_arcs] */
  const int32_t *arc_label;  /* [total_arcs] */
  const float *arc_w;        /* [total_arcs] or NULL */
  const uint32_t *fwd_stream;/* [fwd_words] */
  const uint32_t *bwd_stream;/* [bwd_words] */
  const int32_t *fwd_perm;   /* [fwd_slots] tile slot -> canonical arc, -1 = empty slot */
  const int32_t *bwd_perm;   /* [bwd_slots] */
  const uint32_t *arc_sd;    /* [total_arcs + 8] src | dst << 16: the canonical arcs in 6 B/arc */
  const uint16_t *arc_l16;   /* [total_arcs + 8] label            for the posterior pass        */
  /* ABI 7 */
  const struct nfst_chunks *chunks; /* HOST pointer or NULL: the chunked programs of a batch of deep, narrow lattices
                                       (nfst_pack_chunks) with their device arrays and scratch; nfst_backward and
                                       nfst_forward_backward run the chunked flavour when it is there */
  const int32_t *only;       /* launcher-internal, callers pass NULL: device [n_lattices]; a sweep kernel skips lattice b */
  int32_t only_tag;          /*   unless only[b] == only_tag (the chunked flavour hands lattices whose numbers leave   */
  int32_t reserved2;         /*   its range back to the general kernels this way)                                      */
} nfst_batch;

/*
 * Chunked programs for deep, narrow lattices (SNIPS-shaped tagging machines, main_snips.py / lstm_snips.yaml:2
 * max_length 750: hundreds of levels of a few states each, where a level-by-level sweep is a chain of ~200 ns steps
 * with most of a wave idle).  Per lattice and direction the states are put in topological order ("positions":
 * by longest path from the start for alpha, to the sink for beta) and the positions are cut into C chunks.  A chunk's
 * values depend on at most F positions before it (its frontier; arcs reach at most R - 1 positions back), so
 *   pass 1  every chunk is swept at the same time with F right-hand sides (unit vectors on its frontier), one lane per
 *           (chunk, right-hand side): T[p][f] = total weight of the paths from frontier state f to position p inside the chunk;
 *   pass 2  the frontier values are chained through the chunks (C steps of an F x F product);
 *   pass 3  value[p] = sum_f T[p][f] * frontier[f] for every position, all at once.
 * Depth L becomes L / C + C.  Pass 1 computes in plain float64 (a chunk's products stay within 2^+-480 for any
 * sane scores); passes 2 and 3 in (float64 mantissa, int32 exponent).  A lattice whose numbers leave that range
 * (a weight or a partial sum beyond 2^+-480) is flagged on the device and run by the general kernels in the same call.
 * Results: the same function as the general kernels (<= 1e-9 from the float64 oracle on log Z).
 * (A launch captured into a HIP graph replays with the tag it was captured with: a lattice flagged in one replay stays
 * with the general kernels in the following ones -- same results, their speed.)
 */
#define NFST_CHK_META_WORDS 8
#define NFST_CHK_C 0          /* chunks */
#define NFST_CHK_F 1          /* right-hand sides per chunk (largest frontier) */
#define NFST_CHK_R 2          /* ring slots (power of two > the longest reach of an arc, in positions; >= F) */
#define NFST_CHK_NPOS 3       /* positions = reachable states; position 0 is the start (alpha) / the sink (beta) */
#define NFST_CHK_TAB_OFF 4    /* first chunk of this program in tab (4 int32 per chunk) */
#define NFST_CHK_STREAM_OFF 5 /* first entry of this program in stream */
#define NFST_CHK_POS_OFF 6    /* first position of this program in pos */
#define NFST_CHK_T_OFF 7      /* first row of this program's T matrix, in units of 64 doubles */
/* stream entry: operand ring slot (bits 0..5) | last entry of its state (bit 6) | weight zero (bit 7: the entry
 * of a state without arcs, and the entries that pad a chunk to a multiple of eight) | canonical arc, relative to the lattice (bits 8..31) */
#define NFST_CHK_LAST 0x40u
#define NFST_CHK_ZERO 0x80u

typedef struct nfst_chunks {
  int32_t n_lattices;
  int32_t threads;         /* workgroup size the programs were cut for: C * F <= threads for every program */
  int32_t lds_bytes;       /* dynamic LDS of the sweep kernel */
  int32_t launches;        /* launcher-internal: counts the launches (tags the device flags) */
  int64_t n_tab;           /* chunks over all programs */
  int64_t n_stream;        /* stream entries over all programs */
  int64_t n_pos;           /* positions over all programs */
  int64_t t_units;         /* T rows over all programs, in units of 64 doubles */
  int64_t total_rows;      /* of the batch the programs were cut from */
  int64_t total_arcs;
  const int32_t *meta;     /* [n_lattices * 2 * NFST_CHK_META_WORDS]: direction 0 = alpha, 1 = beta */
  const int32_t *tab;      /* [n_tab * 4]: first position, first entry (relative to the program), entries, 0 */
  const uint32_t *stream;  /* [n_stream] */
  const int32_t *pos;      /* [n_pos] position -> state */
  const uint16_t *label;   /* [n_stream] the label of every entry's arc (saves the sweep kernel a dependent load) */
  void *ws;                /* device scratch of nfst_chunks_ws_bytes() bytes, owned by the caller and zeroed by it once (when
                              `launches` is zero), used by every launch on this batch (one launch at a time per batch) */
  int64_t ws_bytes;
} nfst_chunks;

typedef struct nfst_chunk_opts {
  int32_t threads;         /* 0 = by batch size: 1024 when every program gets a CU to itself, else 512 */
  int32_t lds_bytes;       /* 0 = by batch size: 152 KiB / 64 KiB */
  int32_t force;           /* 1 = cut every batch that can be cut (testing); 0 = only when the cost model of the
                              two flavours says the chunked one is faster */
  int32_t max_chunks;      /* 0 = no limit beside threads / LDS (testing: small values) */
  int32_t n_threads;       /* host threads over lattices (0 = hardware) */
  int32_t reserved;
} nfst_chunk_opts;

typedef struct nfst_chunks_host nfst_chunks_host; /* opaque, owns host arrays */
/* Cut the chunked programs of a packed batch (host arrays, as nfst_packed_view returns them).  *out = NULL (and
 * NFST_OK) when the batch is not one for this flavour: an arc that reaches more than 63 positions back, a frontier
 * too wide for the workgroup, or no gain by the cost model. */
int nfst_pack_chunks(const nfst_batch *host_batch, const nfst_chunk_opts *opts, nfst_chunks_host **out);
int nfst_chunks_view(const nfst_chunks_host *c, nfst_chunks *view); /* host pointers; ws = NULL */
void nfst_chunks_free(nfst_chunks_host *c);
int64_t nfst_chunks_ws_bytes(const nfst_chunks *c);

const char *nfst_strerror(int code);
int nfst_abi_version(void);
/* sizeof the named struct of this header ("nfst_batch", "nfst_scores", "nfst_chunks", "nfst_chunk_opts", "nfst_pack_opts",
 * "nfst_step_extras", "nfst_arcs_device") as the library was compiled, -1 for another name: a binding checks its own
 * declarations against it (the entry points copy whole structs) */
int nfst_sizeof(const char *struct_name);
/* 1 when a HIP device is usable, 0 otherwise (never an error) */
int nfst_device_available(void);

/*
 * Launcher switches (tests and A/B measurements; every switch only selects among kernels that compute the same
 * function).  The environment variables of the same meaning -- NFST_TW, NFST_NO_FUSED, NFST_XCACHE, NFST_PRECISE,
 * NFST_NEU_PACK, NFST_NEU_BF16, NFST_NEU_NO_SMALL, NFST_LDS_RESERVE_KB -- are read once, when the first launcher runs, never on the
 * launch path.  name: "tw", "fused", "xcache", "neu_pack", "neu_bf16", "neu_small" (0 / 1), "precise" (0 never, 1 whenever it
 * fits, -1 by program depth: the default), "lds_reserve_kb" (0 .. 96).  Not thread-safe against concurrent launches.
 */
int nfst_tuning_set(const char *name, int value);

/* ---------------------------------------------------------------- packing (host) */
typedef struct nfst_packed nfst_packed; /* opaque, owns host arrays */

/* pack options; zero-initialise for defaults */
typedef struct nfst_pack_opts {
  int32_t n_threads;       /* host threads over lattices (0 = hardware) */
  int32_t slots_per_lane;  /* arc slots per lane of a tile: 1, 2 or 4 (0 = per lattice and direction, cheapest) */
  int32_t group_mode;      /* largest lane group of a state: 1 = narrow (8 lanes, more tiles for high-degree
                              states), 2 = wide (64 lanes), 0 = per lattice and direction, cheapest */
  int32_t reserved1;       /* 1 = never use the compact (24-bit record) tile format (testing) */
} nfst_pack_opts;

/*
 * Dense tables of the reference (scorers.py:995-1035; collated batches as in
 * util/dataset_reader.py:175-186): emission [B, n_rows, V] as uint8 bool
 * (emission_is_float = 0) or float32 log weight with -inf for "no arc"
 * (emission_is_float = 1); transition [B, n_rows, V] int64.  Rows that cannot be
 * reached from state 0 (collate padding, the machine's old final state) are
 * ignored.  Host pointers.  On success *out owns the packed batch.
 * err_lattice (may be NULL) receives the index of the offending lattice.
 */
int nfst_pack_dense(const void *emission, int emission_is_float, const int64_t *transition,
                    int32_t n_lattices, int32_t n_rows, int32_t vocab,
                    const nfst_pack_opts *opts, nfst_packed **out, int32_t *err_lattice);

/*
 * Arc lists (a CSR/COO sidecar of the same lattices): lattice b owns arcs
 * [arc_off[b], arc_off[b+1]) sorted by (src, label); n_rows[b] rows each.
 * arc_w may be NULL.  Host pointers.
 */
int nfst_pack_arcs(const int32_t *n_rows, const int64_t *arc_off, const int32_t *src,
                   const int32_t *label, const int32_t *dst, const float *arc_w,
                   int32_t n_lattices, int32_t vocab, const nfst_pack_opts *opts,
                   nfst_packed **out, int32_t *err_lattice);

/*
 * Packed batches on the host (sidecar files, DataLoader workers).
 *
 * nfst_validate_batch: everything a kernel turns into an address without looking -- offsets and counts of the meta
 * records, row pointers, the state / label / arc ids of the canonical arrays, of every tile's control words and records
 * and of the slot -> arc maps -- is inside the batch's arrays and inside the LDS rows the launchers size from max_rows
 * and vocab.  Host pointers, O(words of the batch).  A batch read back from a file is checked with this before it is
 * ever handed to a kernel (LatticeBatch.load); err_lattice (may be NULL) receives the offending lattice.
 *
 * nfst_crc32c: CRC-32C of a byte range (the checksum stored per array in sidecar files); chain calls through `seed`
 * (0 for the first).
 *
 * nfst_concat_sizes / nfst_concat_packed: one batch from already packed ones without running the packer again -- the
 * collate step of a loader that packs every example once (util/dataset_reader.py:175-186 pads and stacks dense tables
 * instead).  concat_sizes fills the scalar fields of *total; the caller allocates the arrays (ordinary or page-locked
 * host memory), stores their addresses in *total and calls concat_packed, which writes through those pointers:
 * memcpy + offset fix-ups, n_threads host threads over the parts (0 = hardware).  The parts must agree in vocabulary
 * and in being weighted; the result is bit-identical to packing the lattices as one batch.
 */
int nfst_validate_batch(const nfst_batch *lat, int32_t *err_lattice);
uint32_t nfst_crc32c(const void *data, int64_t n_bytes, uint32_t seed);
int nfst_concat_sizes(const nfst_batch *parts, int32_t n_parts, nfst_batch *total);
int nfst_concat_packed(const nfst_batch *parts, int32_t n_parts, const nfst_batch *out, int32_t n_threads);

/*
 * The packer on the device (SURVEY.md section 2 "K1", section 7 step 4).  The reference's trainer hands set_masks
 * tables that already live on the GPU (modules/lightning.py:417 -> scorers.py:877-885); these entry points pack them
 * there, and pack compact arc lists (12 bytes per arc) uploaded by a loader.  Output: the arrays of an nfst_batch in
 * device memory, BIT-IDENTICAL to nfst_pack_dense / nfst_pack_arcs on the same lattices.  Supported: vocab + 2 <= 2048
 * (compact tiles, four slots per lane), nfst_pack_opts.group_mode; slots_per_lane must be 0 or 4.  A lattice beyond
 * the device packer's limits (more than 16384 reachable states or pieces of one sweep direction) gets NFST_ERR_LIMIT in
 * its status word: pack that batch on the host.
 *
 * Dense tables -> arc lists, two launches around one small read-back:
 *   nfst_dense_to_arcs_count  reach [B, n_rows] bytes, row_cnt [B, n_rows], counts [B] = arcs of the rows reachable
 *                             from state 0 (collate padding rows never become arcs), status [B] (NFST_ERR_INDEX)
 *   (host: arc_off = prefix sums of counts; allocate src / label / dst (/ arc_w) of arc_off[B] entries)
 *   nfst_dense_to_arcs_write  the arcs of lattice b at arc_off[b] in (state, label) order
 * Arc lists -> packed batch, two launches around one small read-back:
 *   nfst_pack_device_plan     meta [B, 16] counts (rows, arcs, tiles, sink, depth, formats), status [B], scratch_rows [B]
 *   nfst_pack_device_layout   (host) offsets into *meta, sizes and flags into *header: allocate the arrays, store their
 *                             device addresses in the header, upload meta (it is the batch's meta array)
 *   nfst_pack_device_emit     writes every array of the batch
 * ws: device workspace of nfst_pack_device_ws_bytes() bytes.  The emitting pass reads what the planning pass of the same
 * batch left there: the same workspace, untouched in between; free afterwards.
 */
typedef struct nfst_arcs_device {
  const int32_t *n_rows;   /* [B] device */
  const int64_t *row_off;  /* [B + 1] device: prefix sums of n_rows */
  const int64_t *arc_off;  /* [B + 1] device */
  const int32_t *src;      /* [total_arcs] device, sorted by (src, label) inside a lattice */
  const int32_t *label;
  const int32_t *dst;
  const float *arc_w;      /* or NULL */
  int64_t total_rows;      /* = row_off[B] */
  int64_t total_arcs;      /* = arc_off[B] */
  int32_t n_lattices;
  int32_t vocab;
} nfst_arcs_device;

int nfst_dense_to_arcs_count(const void *emission, int emission_is_float, const int64_t *transition, int32_t n_lattices,
                             int32_t n_rows, int32_t vocab, uint8_t *reach, int32_t *row_cnt, int32_t *counts,
                             int32_t *status, void *stream);
int nfst_dense_to_arcs_write(const void *emission, int emission_is_float, const int64_t *transition, int32_t n_lattices,
                             int32_t n_rows, int32_t vocab, const uint8_t *reach, const int32_t *row_cnt,
                             const int64_t *arc_off, int32_t *src, int32_t *label, int32_t *dst, float *arc_w, void *stream);
int64_t nfst_pack_device_ws_bytes(int32_t n_lattices, int64_t total_rows, int64_t total_arcs);
int nfst_pack_device_plan(const nfst_arcs_device *arcs, const nfst_pack_opts *opts, void *ws, int64_t ws_bytes, int32_t *meta,
                          int32_t *status, int32_t *scratch_rows, void *stream);
int nfst_pack_device_layout(int32_t *meta, const int32_t *status, const int32_t *scratch_rows, int32_t n_lattices,
                            int32_t vocab, int32_t weighted, nfst_batch *header, int32_t *err_lattice);
int nfst_pack_device_emit(const nfst_arcs_device *arcs, const nfst_pack_opts *opts, void *ws, int64_t ws_bytes,
                          const int32_t *meta, int32_t *status, const nfst_batch *out, void *stream);

/* view of the arrays owned by a packed batch (host pointers, valid until free) */
int nfst_packed_view(const nfst_packed *p, nfst_batch *view);
void nfst_packed_free(nfst_packed *p);

/* ---------------------------------------------------------------- kernels (device) */

/* bytes of dynamic LDS the sweep kernels need for this batch (informational) */
int64_t nfst_lds_bytes(const nfst_batch *lat);

/*
 * Arc scores.  The log weight of canonical arc a of lattice b is
 *     theta[(theta_stride * b) + label[a]]  (+ arc_w[a] if the batch is weighted)
 *                                          (+ arc_scores[a] if arc_scores != NULL)
 * theta: device float32 [V] (theta_stride = 0, one table for the batch, the
 * WFSTScorer case) or [B, V] (theta_stride = V).  arc_scores: device float32
 * [total_arcs] in canonical order, or NULL.
 */
typedef struct nfst_scores {
  const float *theta;
  int64_t theta_stride;
  const float *arc_scores;
  float *reserved_ws;      /* unused since ABI 5 (was a workspace for slot-ordered extras: the sweeps now gather
                              per-arc extras inside the kernel, on otherwise idle waves); pass NULL */
  int64_t reserved_flag;   /* unused, pass 0 */
} nfst_scores;

/*
 * Backward (beta) sweep: log beta[row] for every row (float32, -inf for rows
 * unreachable from 0), log Z = log beta[start] per lattice.  Any output may be
 * NULL.  beta_me (optional) receives the raw (mantissa, exponent) pairs, 2
 * float32 words per row, consumed by nfst_sample_paths.
 * Replaces FSAGRUScorer.compute_beta (scorers.py:858-875); semantics follow
 * compute_beta_per_sample (692-751) with Wh = 0.
 */
int nfst_backward(const nfst_batch *lat, const nfst_scores *scores, float *logbeta,
                  double *logz64, float *logz32, float *beta_me, void *stream);
/* (lat->chunks != NULL: the chunked flavour, see nfst_chunks) */

/*
 * Full forward-backward: alpha and beta sweeps, log Z and arc posteriors
 * posterior[a] = exp(alpha[src] + score + beta[dst] - log Z) in canonical arc
 * order (= d log Z / d score[a]); grad_theta (optional, [B, V] float32) receives the
 * posteriors summed per label (d log Z / d theta[b, l]).
 * logz_total (optional, 3 doubles, zero-initialised once by the caller): the launch adds every
 * lattice's log Z to logz_total[total_slot] (atomic adds: the summation order is not fixed) and
 * clears logz_total[(total_slot + 1) % 3] for the next launch -- the loss of a training step
 * without a reduction kernel; use total_slot = step % 3.
 * Environment: NFST_LDS_RESERVE_KB=<0..96> (read once per process) makes nfst_backward and
 * nfst_forward_backward leave that much LDS free on every CU when they run one lattice per CU,
 * so that a small kernel of another stream (RCCL's all-reduce of the loss) can run beside them.
 */
int nfst_forward_backward(const nfst_batch *lat, const nfst_scores *scores, float *logalpha,
                          float *logbeta, double *logz64, float *logz32, float *posterior,
                          float *grad_theta, float *beta_me, double *logz_total, int32_t total_slot, void *stream);

/*
 * Viterbi: best[b] = max path score (float32), paths [B, max_len] int32 labels
 * of the best path (bos .. eos) padded with `pad`, lengths [B]; path_arcs
 * (optional) canonical arc ids, -1 padded.  Ties keep the smallest label.
 */
int nfst_viterbi(const nfst_batch *lat, const nfst_scores *scores, float *best, int32_t *paths,
                 int32_t *path_arcs, int32_t *lengths, int32_t max_len, int32_t pad,
                 void *stream);

/*
 * Exact posterior path sampling: K walks per lattice from state 0; at state s
 * arc a is taken with probability exp(score[a] + beta[dst] - beta[s]), chosen by
 * inverse CDF over the state's arcs in label order from uniforms [B, K, max_len]
 * (device float32 in [0,1)); if uniforms is NULL a Philox4x32-10 stream is used
 * (key = seed, counter = (walk, step / 4): one output word per step).  paths [B, K, max_len] labels padded with `pad`,
 * path_arcs (optional) canonical arc ids, lengths [B, K], logq [B, K] =
 * path score - log Z.  status (device int32, one word, zeroed by the caller)
 * becomes NFST_ERR_LENGTH if a walk does not reach the sink within max_len.
 */
int nfst_sample_paths(const nfst_batch *lat, const nfst_scores *scores, const float *beta_me,
                      const double *logz64, int32_t k, int32_t max_len, const float *uniforms,
                      uint64_t seed, int32_t pad, int32_t *paths, int32_t *path_arcs,
                      int32_t *lengths, float *logq, int32_t *status, void *stream);

/*
 * Forced walk of marks [B, K, max_len] (pad-terminated) from state 0:
 * path_score [B, K] = sum of arc scores (-inf if a mark has no arc),
 * end_state [B, K] (0 if the walk fell off the lattice, like the dense gather).
 */
int nfst_score_paths(const nfst_batch *lat, const nfst_scores *scores, const int32_t *marks,
                     int32_t k, int32_t max_len, float *path_score, int32_t *end_state,
                     void *stream);

/* state' = transition[state, label] for N = B*K walkers (walker n belongs to
 * lattice n / k); 0 where the arc does not exist.  int64 in/out like the
 * reference's LongTensors. */
int nfst_step(const nfst_batch *lat, const int64_t *state, const int64_t *label, int64_t *next,
              int32_t k, void *stream);

/* out [B*K, V] float32: 0 (or the table's weight) where the walker's state has
 * an arc with that label, -inf elsewhere.  If inp != NULL (the previously
 * emitted mark per walker) the legality masks of LeftToRightScorer.mask_out_invalid
 * (scorers.py:59-83, 314-338) are added in the same pass: bos never, pad only
 * after eos/pad, and -- when has_to_end -- only eos for walkers that have not
 * ended: the whole of FSAGRUScorer.mask_out_invalid in one kernel. */
int nfst_emission_mask(const nfst_batch *lat, const int64_t *state, const int64_t *inp,
                       int32_t pad, int32_t bos, int32_t eos, int32_t has_to_end, float *out,
                       int32_t k, void *stream);

/* out [B*K, V] = values[row_off(b) + transition[state, label]] with values a
 * row-indexed float32 array (beta in the probability domain in the reference,
 * any row-indexed array here); labels without an arc read row 0 like the dense
 * gather does. */
int nfst_beta_logits(const nfst_batch *lat, const float *values, const int64_t *state,
                     float *out, int32_t k, void *stream);

/*
 * Optional operands of nfst_proposal_step (zero-initialise; NULL pointers switch a part off).
 *   value_state [N]: the state whose transition row the `values` gather reads.  The reference gathers
 *     its beta "logits" from the state BEFORE the previous symbol was consumed (scorers.py:584-590 run
 *     inside super().actual_left_to_right_score, the advance is line 679) while the masks use the state
 *     after it: pass the previous step's `state` here to reproduce that (pinned by
 *     tests/golden/sampler_beta.npz); NULL = gather from `state`.
 *   accumulated [N] int64, vocab_use [N, V] float32 (in/out, zeroed by the caller before the first
 *     step): the counters of scorers.py:654-661, updated with the previous symbol `inp` first; then
 *       scores[:, insertion_mark] -= insert_penalty * (length - insert_threshold)
 *           where accumulated > insert_threshold             (if insert_threshold > 0; 663-669)
 *       scores -= vocab_use * length_penalty                  (if 0 < length_threshold < length; 671-677)
 *     `length` is the step's metadata["length"] (1 at the first step).
 *   not_pad: device int32 word (zeroed by the caller): the launch adds the number of walkers whose symbol is
 *     not `pad` -- zero means "every walker has ended" (Sampler.all_reached_eos, samplers.py:288-290) without a
 *     reduction kernel; a sampling loop reads it every few steps instead of synchronising with the device at
 *     every step.
 */
typedef struct nfst_step_extras {
  const int64_t *value_state;
  int64_t *accumulated;
  float *vocab_use;
  int32_t insertion_mark;
  int32_t insert_threshold;
  float insert_penalty;
  int32_t length_threshold;
  float length_penalty;
  int32_t length;
  int32_t *not_pad;
} nfst_step_extras;

/*
 * Fused lattice side of one proposal-sampler step (Sampler.stateful_sample, samplers.py:243-297;
 * left_to_right_score, scorers.py:340-366; mask_out_invalid, 1037-1054 + 59-83; beta-logit gather,
 * 581-593; penalties, 654-677; update_fsa_state, 683-690): for every walker n (lattice n / k)
 *   logits = (pad_masking(scores[n, :] [+ values[next state]] [- penalties]) + emission row + legality masks) / temperature
 *   symbol = forced[n], or the first mark (in id order) whose CDF exceeds uniforms[n]
 *   logq = logits[symbol] - logsumexp(logits), logz = that logsumexp (optional), next_state as nfst_step.
 * `state` is the state after the previous symbol `inp` was consumed.  inp may be NULL: no bos / pad /
 * eos legality masks and no counters then.  values is row-indexed [total_rows] (e.g. beta in the domain
 * the caller's scorer expects) or NULL.  extras may be NULL.  logits_out (optional, [N, V]) receives the
 * masked, scaled logits: what nfst_proposal_step_backward needs.  vocab <= 4096 (<= 3400 with
 * extras->value_state).
 */
int nfst_proposal_step(const nfst_batch *lat, const int64_t *state, const int64_t *inp, const float *scores,
                       const float *values, int32_t pad, int32_t bos, int32_t eos, int32_t has_to_end, float temperature,
                       const float *uniforms, const int64_t *forced, const nfst_step_extras *extras, int64_t *symbol,
                       float *logq, float *logz, int64_t *next_state, float *logits_out, int32_t k, void *stream);

/*
 * Backward of nfst_proposal_step: the reference's Categorical.log_prob is differentiable in the
 * proposal's logits (samplers.py:256-273; tune_proposal, lightning.py:339-406).  From the logits the
 * forward pass wrote: grad_scores [N, V] = (g_logq * (onehot(symbol) - softmax) + g_logz * softmax) /
 * temperature for legal marks other than pad, 0 elsewhere (g_logq, g_logz [N]; either may be NULL).
 * grad_values (optional, [total_rows], zeroed by the caller) accumulates the same numbers through the
 * next-state gather out of value_state (float atomics: the summation order is not fixed).
 */
int nfst_proposal_step_backward(const nfst_batch *lat, const int64_t *value_state, const float *logits,
                                const int64_t *symbol, const float *logz, const float *g_logq, const float *g_logz,
                                int32_t pad, float temperature, float *grad_scores, float *grad_values, int32_t k,
                                void *stream);

/*
 * Neuralised beta: FSAGRUScorer.compute_beta_per_sample with Wh != 0
 * (modules/scorers.py:692-751; compute_beta_parallel, 753-856, is the same
 * recurrence).  For every arc (s, l, s') with s' != s
 *     t   = tanh(label_x[l] + Wh . beta_hat(s')),  label_x[l] = Wx . e(l) + bias
 *     msg = exp(w . t (+ arc_w)) * beta(s')
 *     beta(s) = sum msg,  beta_hat(s) = sum (msg / beta(s)) t;  beta(sink) = 1, beta_hat(sink) = 0.
 * label_x: device float32 [vocab, hid]; wh: Wh as the reference holds it, [hid (out),
 * hid (in)] row-major; w: [hid]; hid <= 512.  Outputs: log_beta [total_rows] (natural log;
 * the reference returns exp of it), beta_hat [total_rows, hid].  ws: device
 * workspace of nfst_neural_ws_floats() floats, contents irrelevant.
 */
int64_t nfst_neural_ws_floats(const nfst_batch *lat, int32_t hid);
int nfst_backward_neural(const nfst_batch *lat, const float *label_x, const float *wh, const float *w,
                         int32_t hid, float *log_beta, float *beta_hat, float *ws, void *stream);

/*
 * Gradient of nfst_backward_neural: tune_proposal differentiates log q through compute_beta()
 * (modules/lightning.py:339-406) w.r.t. Wh, Wx, W, beta_bias and the embeddings
 * (scorers.py:954-970).  Inputs: the forward call's label_x, w, beta_hat and workspace (ws_fwd,
 * untouched since), wh_t = Wh transposed ([hid (in), hid (out)] row-major), g_log_beta
 * [total_rows] = dL / d log_beta, g_beta_hat [total_rows, hid] = dL / d beta_hat or NULL.
 * Outputs: grad_label_x [vocab, hid] and grad_w [hid] ACCUMULATE (float atomics: the caller zeroes
 * them, the summation order is not fixed); gamma [total_rows, hid] (zeroed by the caller) receives
 * dL / d (Wh . beta_hat(s)) per state, so that dL/dWh = gamma^T . beta_hat is one library GEMM on
 * the caller's side.  dL/dWx, dL/dbias and dL/d embeddings follow from grad_label_x through
 * label_x = emb . Wx^T + bias.  ws: nfst_neural_grad_ws_floats() floats, contents irrelevant.
 */
int64_t nfst_neural_grad_ws_floats(const nfst_batch *lat, int32_t hid);
int nfst_backward_neural_grad(const nfst_batch *lat, const float *label_x, const float *wh_t, const float *w,
                              int32_t hid, const float *beta_hat, const float *ws_fwd, const float *g_log_beta,
                              const float *g_beta_hat, float *gamma, float *grad_label_x, float *grad_w,
                              float *ws, void *stream);

/* out[a] = theta[(theta_stride * b) + label[a]] (+ arc_w[a]) (+ arc_scores[a]) */
int nfst_gather_label_scores(const nfst_batch *lat, const nfst_scores *scores, float *out,
                             void *stream);

/*
 * Fused log_softmax + gather + mask-sum over sequences: scores [N, T, V] float32,
 * marks [N, T] int64, out [N] float32.
 * mask_mode NFST_MASK_STATICRNN -- StaticRNNScorer.evaluate_seq_with_temp (scorers.py:1564-1611):
 * masks as in scorers.py:89-134: bos/pad illegal, after eos/pad only pad, beyond
 * max_length only eos (max_length < 0 disables); the pad column of scores is
 * zeroed (pad_masking_3d, scorers.py:189-192); positions holding pad contribute
 * 0.  normalize = 0 skips the log_softmax (self_normalized = False).
 * smoothing in [0,1): 0 = evaluation (gather of the realised mark); > 0 = the
 * training branch (scorers.py:1502-1528, 1584-1592): weight 1 - smoothing on the
 * realised mark, smoothing / (legal marks - 1) on the other legal marks, values
 * clamped to +-1e9.
 * mask_mode NFST_MASK_GPT2 -- GPT2Wrapper.forward (modules/transformer.py:45-52): scores are
 * the language model's logits for the bos-shifted input, marks is gold_x (the sequence with one
 * trailing pad); the pad logit is replaced by -1e8, there are no legality masks, positions
 * holding pad contribute 0 (bos, eos, max_length are ignored; temp = 1, normalize = 1 there).
 */
#define NFST_MASK_STATICRNN 0
#define NFST_MASK_GPT2 1
int nfst_path_logprob(const float *scores, const int64_t *marks, int64_t n, int32_t t,
                      int32_t vocab, int32_t pad, int32_t bos, int32_t eos, int32_t max_length,
                      float temp, int32_t normalize, float smoothing, int32_t mask_mode, float *out,
                      void *stream);

/*
 * Backward of nfst_path_logprob: grad_scores [N, T, V] = grad_out[n] * d out[n] / d scores[n, t, v]
 * (every element is written).  The reference trains p~ straight through this op:
 * p_loss = -(num_prob - denom_prob).mean() (modules/lightning.py:511-516) back-propagates through
 * evaluate_seq_with_temp.  Per row: (target - softmax * sum(target)) / temp with the one-hot or
 * label-smoothed target; the pad column and rows whose mark is pad get 0.  The row is recomputed
 * from scores (nothing is saved by the forward pass): N T V 4 bytes read, as many written.
 */
int nfst_path_logprob_backward(const float *scores, const int64_t *marks, const float *grad_out, int64_t n,
                               int32_t t, int32_t vocab, int32_t pad, int32_t bos, int32_t eos,
                               int32_t max_length, float temp, int32_t normalize, float smoothing,
                               int32_t mask_mode, float *grad_scores, void *stream);

/* log_w [B,K] = log_p - log_q ; log_marginal [B] = logsumexp_k(log_w) - log K */
int nfst_iwae(const float *log_p, const float *log_q, int32_t b, int32_t k, float *log_w,
              float *log_marginal, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NFST_HIP_H */
