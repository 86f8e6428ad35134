/*
 * dnastore_amd.h -- C ABI of the MI355X-native error decoder for ihh/dnastore.
 *
 * The reference has no FFI/plugin boundary: its Viterbi path is the C++ function
 *   vguard<FastSeq> decodeFastSeqs(const char*, const Machine&, const MutatorParams&)
 * (reference src/viterbi.h:108, src/viterbi.cpp:306-320) called from one arm of main
 * (t/dnastore.cpp:217-223).  This header is the boundary a maintainer would bind in its
 * place: plain pointers and sizes, opaque handles, int status returns, no exceptions,
 * no torch types.  Each entry point names the reference interface it replaces.
 *
 * Threading: a handle is thread-compatible (one host thread / one GPU per handle).
 * Every function returns DNAS_OK (0) or a negative DNAS_E_* code; dnas_last_error()
 * returns the message of the calling thread's last failure.
 */
#ifndef DNASTORE_AMD_H
#define DNASTORE_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNAS_OK 0
#define DNAS_E_INVALID (-1)   /* bad argument                                              */
#define DNAS_E_IO (-2)        /* file not found / unreadable  (reference: Fail -> exit(1))  */
#define DNAS_E_PARSE (-3)     /* malformed JSON / FASTA / contexts (verifyContexts)         */
#define DNAS_E_CYCLIC (-4)    /* "Transducer is cyclic, can't toposort" (trans.cpp:631-632) */
#define DNAS_E_NOT_DNA (-5)   /* "Not a DNA-outputting machine" (viterbi.cpp:27-28)         */
#define DNAS_E_BAD_BASE (-6)  /* non-ACGT read character (fastseq.cpp:30-35)                */
#define DNAS_E_DEVICE (-7)    /* HIP runtime failure / no GPU / extension not built         */
#define DNAS_E_NOMEM (-8)
#define DNAS_E_UNSUPPORTED (-9)

/* per-read status written by dnas_viterbi_batch */
#define DNAS_READ_OK 0
#define DNAS_READ_NO_PATH 1        /* loglike == -inf: empty string, "No valid Viterbi decoding found" (viterbi.cpp:198-201) */
#define DNAS_READ_OUT_OVERFLOW 2   /* decoded string longer than the caller's slot            */
#define DNAS_READ_TRACEBACK_FAIL 3 /* checkBest assertion (viterbi.cpp:230-237)               */

typedef struct dnas_machine dnas_machine; /* Machine, reference src/trans.h:82-126 */
typedef struct dnas_model dnas_model;     /* device-resident MachineScores + MutatorScores + InputModel */

/* MutatorParams, reference src/mutator.h:9-31 */
typedef struct dnas_mutator_params {
  double p_del_open, p_del_extend, p_tan_dup, p_transition, p_transversion;
  int32_t n_len;        /* pLen.size() = maxDupLen() */
  int32_t local;        /* 1 = local (partial reads allowed), 0 = --error-global */
  double p_len[32];
} dnas_mutator_params;

/*
 * The flattened model: everything ViterbiMatrix derives from (Machine, MutatorParams)
 * before touching a read -- InputModel (viterbi.cpp:6-14,309-310), MachineScores
 * (viterbi.cpp:23-60), MutatorScores (mutator.cpp:56-75), decoderToposort
 * (trans.cpp:604-634) -- as flat CSR arrays.  Edge order inside every row is the
 * reference's enumeration order (ascending source state, then transition order), which
 * is the traceback tie-break order.  All pointers are owned by the dnas_flat handle.
 */
typedef struct dnas_flat_model {
  int32_t n_states;      /* N                                                   */
  int32_t max_dup_len;   /* D = min(maxLeftContext, P), viterbi.cpp:63          */
  int32_t n_len;         /* P                                                   */
  int32_t local;
  int32_t n_emit, n_null;
  /* incoming edges per destination state */
  const int32_t *ein_ptr, *ein_src;   const double *ein_score; const uint8_t *ein_in, *ein_base;
  const int32_t *nin_ptr, *nin_src;   const double *nin_score; const uint8_t *nin_in;
  /* outgoing edges per source state */
  const int32_t *eout_ptr, *eout_dst; const double *eout_score;
  const int32_t *nout_ptr, *nout_dst; const double *nout_score;
  const uint8_t *mdl;                 /* [N]   maxDupLenAt, viterbi.h:104                      */
  const uint8_t *ctx;                 /* [N*D] ctx[j*D+k] = tanDupBase(j,k), viterbi.h:105     */
  const int32_t *topo;                /* [N]   decoderToposort order                           */
  double no_gap, del_open, del_extend, del_end, tan_dup;
  double sub[16];                     /* sub[base*4+observed]                                  */
  const double *len;                  /* [P]                                                   */
  char alphabet[64];                  /* inputAlphabet(Relaxed|Control|SEOF)                   */
  double sym_logp[128];               /* log P(input symbol), 0 where absent                   */
} dnas_flat_model;

typedef struct dnas_flat dnas_flat;

/* ---- host side: file formats (no GPU needed) ------------------------------------- */

/* Machine::fromFile / readJSON (trans.cpp:431-482). */
int dnas_machine_load_json(const char *path, dnas_machine **out);
int dnas_machine_parse_json(const char *text, size_t len, dnas_machine **out);
void dnas_machine_free(dnas_machine *m);
int32_t dnas_machine_n_states(const dnas_machine *m);
/* Machine::writeJSON (trans.cpp:402-429) into a malloc'd buffer the caller frees with dnas_free. */
int dnas_machine_write_json(const dnas_machine *m, char **out_text, size_t *out_len);

/* Exact encoder (reference Encoder<FastaWriter>, src/encoder.h:7-243): input symbols
 * ('^', '0', '1', '$', controls ...; SOF/EOF are added when missing, as the reference does)
 * or raw bytes (bits LSB first, encoder.h:222-231) -> DNA string, malloc'd, caller frees
 * with dnas_free.  Makes synthetic reads for bench.py and the parity tests. */
int dnas_encode_symbols(const dnas_machine *m, const char *symbols, size_t n_symbols, char **out_dna, size_t *out_len);
int dnas_encode_bytes(const dnas_machine *m, const uint8_t *bytes, size_t n_bytes, char **out_dna, size_t *out_len);

/* Machine::compose(first, second) (trans.cpp:505-602): first's output feeds second's input. */
int dnas_machine_compose(const dnas_machine *first, const dnas_machine *second, dnas_machine **out);

/* Exact decoder (reference Decoder<W>, src/decoder.h:7-191): DNA -> input symbols (malloc'd string),
 * and BinaryWriter (decoder.h:193-240): '0'/'1' symbols -> bytes, LSB first (malloc'd, *out_len bytes). */
int dnas_decode_exact(const dnas_machine *m, const char *dna, size_t n, char **out_symbols, size_t *out_len);
int dnas_symbols_to_bytes(const char *symbols, size_t n, uint8_t **out_bytes, size_t *out_len);

/* Error model from the CLI flags (t/dnastore.cpp:119-129): --error-sub-prob, --error-iv-ratio,
 * --error-dup-prob, --error-del-open, --error-del-ext, --error-global, --length. */
int dnas_mutator_params_from_flags(double sub_prob, double iv_ratio, double dup_prob, double del_open,
                                   double del_ext, int global, int length, dnas_mutator_params *out);
/* MutatorParams::fromFile (mutator.cpp:18-49), the --error-file format. */
int dnas_mutator_params_load_json(const char *path, dnas_mutator_params *out);

/* MachineScores + InputModel + MutatorScores + toposort, once per (machine, params). */
int dnas_flatten(const dnas_machine *m, const dnas_mutator_params *p, dnas_flat **out);
const dnas_flat_model *dnas_flat_view(const dnas_flat *f);
void dnas_flat_free(dnas_flat *f);

/* ---- device side ------------------------------------------------------------------ */

/* Upload the flattened model to GPU `device_id` and size the lattice arena.
 * arena_bytes = 0 picks a default (a fraction of free HBM). */
int dnas_model_create(const dnas_flat_model *fm, int device_id, size_t arena_bytes, dnas_model **out);
/* The same with options, "key=value,key=value" (NULL: none).  Keys: tier = A | B | C (force a fill kernel; failing
 * to provide it is then an error), cluster = work-groups per read for tier C, threads = 512 | 1024 per work-group
 * (default 1024 for tier A; tier C takes 512 when the machine then fits fewer work-groups), max_clusters, max_slots (reads per
 * fill launch), cluster_spread = 0 | 1 (tier C: the members of a cluster dealt over the XCDs -- their exchange then goes through
 * memory instead of one XCD's L2; default: when that fits a quarter more clusters on the chip), cluster_timeout_s (tier C
 * watchdog per lattice column, default 30 s) and cluster_arrive_s (how long the first barrier of a launch waits for work-groups
 * of a cluster that have not been STARTED yet because something else holds their CUs, default 120 s: the members of a cluster
 * wait for each other, so a launch needs all of them resident; when either time runs out the launch is abandoned and the call
 * returns DNAS_E_DEVICE -- until then the waiting work-groups keep their CUs), sync_place = 0 | 1 (tier C, default 1: the words
 * the members of a cluster agree through are put where the device-scope round trip from the cluster's XCD measures shortest --
 * the memory channel matters: one read alone fills in 42 ms or 54 ms --, a few milliseconds when the model is made; 0: wherever
 * the allocation starts), tb_threads = threads per block of the
 * thread-per-read traceback (multiple of 64, at most 256, default 128) and tb_lanes = reads per wave there (1 .. 64, default 16:
 * the lanes of a wave stand on different states, and a step costs the wave the union of what they do), checkpoint = auto | always | never and
 * segment = columns (bounded-memory decode of reads whose lattice -- the reference's ViterbiMatrix::cell,
 * viterbi.h:48-50 -- does not fit the arena: segments of the lattice are filled from checkpoints and traced back one
 * after the other; results are bit-identical), traceback = thread, arena_fraction, plan_order = 0 | 1 | 2 and plan_slack = 0 .. 8
 * (the row program: states dealt depth first / breadth first / by longest-path level, there with that many eighths of the
 * room between a state's earliest and latest level used.  None given: the machine's tuning record decides -- shipped in
 * <library dir>/tune/, or found in the kernel cache; records = 0: the default program; autotune = 1: a tier-A machine
 * without a record is timed when its first model is created and the verdict kept in the kernel cache).  dnas_model_tier
 * names the program and the record that chose it.  A key that is absent falls back to the environment variable
 * DNAS_<KEY IN UPPER CASE>. */
int dnas_model_create_ex(const dnas_flat_model *fm, int device_id, size_t arena_bytes, const char *options,
                         dnas_model **out);
void dnas_model_destroy(dnas_model *model);

/*
 * The hot path: decodeFastSeqs' per-read loop (viterbi.cpp:312-318) for a batch.
 *   read_offsets[n_reads+1]  prefix offsets into bases
 *   bases[...]               one byte per nucleotide, values 0..3 (A,C,G,T)
 *   out_sym                  caller buffer; read i's decoded symbols go to
 *                            out_sym[out_offsets[i] .. out_offsets[i+1]) (no terminator)
 *   out_len[n_reads]         decoded length (0 for DNAS_READ_NO_PATH)
 *   out_loglike[n_reads]     ViterbiMatrix::loglike(), fp64 (viterbi.h:102)
 *   out_status[n_reads]      DNAS_READ_*
 * Host pointers; the call copies in, runs fill + traceback kernels, copies out, and
 * returns after the stream has drained.
 */
int dnas_viterbi_batch(dnas_model *model, int64_t n_reads, const uint64_t *read_offsets, const uint8_t *bases,
                       char *out_sym, const uint64_t *out_offsets, uint32_t *out_len, double *out_loglike,
                       uint8_t *out_status);

/* Same, with bases and all outputs already resident in this GPU's HBM (device pointers);
 * read_offsets / out_offsets stay host arrays (they drive batching).  Asynchronous on the
 * model's own streams; dnas_model_sync waits.  The library does not know the caller's streams:
 * whatever produced d_bases (and any fill of the output buffers) must have completed before the
 * call, and the outputs may be read after dnas_model_sync. */
int dnas_viterbi_batch_device(dnas_model *model, int64_t n_reads, const uint64_t *read_offsets,
                              const uint8_t *d_bases, char *d_out_sym, const uint64_t *out_offsets,
                              uint32_t *d_out_len, double *d_out_loglike, uint8_t *d_out_status);
int dnas_model_sync(dnas_model *model);

/* Which fill kernel serves this model: "tier A: <shape>" (register/LDS-resident kernel, JIT-specialised
 * for the machine; "...W8 ... 2 work-groups per CU": a small row program compiled so that two reads share a CU), "tier C: <n>
 * work-groups per read, ..." (the same kernel on a cluster of work-groups) or "tier B: <reason>" (general global-memory
 * kernel); behind it the tuning record that chose the row program.  DNAS_TIER=B forces tier B. */
const char *dnas_model_tier(const dnas_model *model);
/* Keep the traceback's event log (see dnas_decode_fastseqs_ex) for the following calls; dnas_model_read_events
 * returns the events of read `read_index` of the last call (out may be NULL to ask for the count). */
int dnas_model_set_event_log(dnas_model *model, int on);
int dnas_model_read_events(dnas_model *model, int64_t read_index, uint64_t *out, int64_t cap, int64_t *n_events);
/* Specialise + compile the tier-A kernel for a machine ahead of time (no GPU needed). */
int dnas_tiera_precompile(const dnas_flat_model *fm, char *note, size_t note_cap);

/* Tier C: the same kernel run by a cluster of `members` work-groups per read (machines beyond one CU).
 * dnas_tierc_precompile compiles it ahead of time (members = 0: the smallest cluster that fits; no GPU needed).
 * dnas_tierc_plan is an analysis / test aid: the tables exactly as the kernel receives them.  info[8] =
 * {members, rows, threads, entries per member, S stripes, inbox rows, proxies, 0}; the other outputs may be NULL:
 * row_shapes[rows][6] = {entries, S stripe, kind, class, full, entries into another member's inbox: 0 none /
 * 1 all / 2 mixed}, entries[members][entries][threads], meta[members][rows][threads], member_of[n_states],
 * lds_index[n_states] = row*threads + lane inside the member, lattice_slot[n_states],
 * fold[members][inbox rows][threads] = the LDS cells behind every inbox cell (one cell per edge between two members).
 * dnas_tierc_plan_proxies: member and lds_index of the plan's proxies (places that hold no state of the machine: they
 * combine the null edges of one member into one state of another and forward the maximum over ONE edge), info[6] of them.
 * members = 1 describes the tier-A plan.  dnas_model_cluster_census: clusters that ran in the last call and how
 * many of them had members on more than one XCD (placement is a speed matter only). */
int dnas_tierc_precompile(const dnas_flat_model *fm, int32_t members, char *note, size_t note_cap);
int dnas_tierc_plan(const dnas_flat_model *fm, int32_t members, int32_t *info, int32_t *row_shapes, uint32_t *entries,
                    size_t entries_cap, uint32_t *meta, int32_t *member_of, int32_t *lds_index, int32_t *lattice_slot,
                    uint32_t *fold);
int dnas_tierc_plan_proxies(const dnas_flat_model *fm, int32_t members, int32_t *proxy_member, int32_t *proxy_lds_index, size_t cap);
int dnas_model_cluster_census(dnas_model *model, int32_t *clusters, int32_t *split);
/* The file name of a machine's row-program tuning record -- members = 1: as tier A (threads = 0: 1024); members = 0 or >= 2: as
 * tier C with the smallest / that cluster (threads as given to the model, 0 = the planner's choice) --: "tune_<hash>.txt", looked
 * for in the kernel cache and in <library dir>/tune/.  The name hashes the machine's graph, the work-group shape and the
 * planner version.  A record starts with "order=<0|1|2> slack=<0..8> kernel=<hash>" (options plan_order, plan_slack; the
 * hash of the kernel source the verdict was measured with, see dnas_kernel_source_hash). */
int dnas_tune_record_name(const dnas_flat_model *fm, int32_t members, int32_t threads, char *out, size_t cap);
/* The hash of the fill kernel's source as this library carries it, 16 hex digits: what the tuning records name as
 * "kernel=".  A record measured with another source is still followed; dnas_model_tier then says "stale". */
int dnas_kernel_source_hash(char *out, size_t cap);

/* Analysis / test aid: where tier A puts each state (lds_index = row*threads + lane; lattice_slot = its
 * position inside a lattice row).  No GPU needed.  DNAS_E_UNSUPPORTED when the machine does not fit tier A. */
int dnas_tiera_plan_slots(const dnas_flat_model *fm, int32_t *lds_index, int32_t *lattice_slot, int32_t *threads,
                          int32_t *rows);

/* Analysis / test aid: the tier-A tables exactly as the fill kernel receives them (layout:
 * dnastore_amd/csrc/viterbi_tiera.hip).  row_shapes[rows][2] = {out-edge entries, S stripe or -1};
 * entries[n_entries][threads]; meta[rows][threads].  Any output may be NULL. */
int dnas_tiera_plan_tables(const dnas_flat_model *fm, int32_t *row_shapes, uint32_t *entries, size_t entries_cap,
                           uint32_t *meta, int32_t *n_entries, int32_t *n_s_rows);

/* Diagnostic: 8 words of the kernel's rounds/stamp buffer (word 0 = total rounds; words 1-5 are filled only by
 * a -DDNAS_STAMP diagnostic build selected with DNAS_TIERA_DEFS). */
int dnas_model_debug_words(dnas_model *model, unsigned long long *out8);

/* Device-time accounting of the last batch call (HIP events on the model's stream). */
typedef struct dnas_batch_stats {
  double fill_ms, traceback_ms;   /* summed kernel durations          */
  int64_t fill_launches, columns; /* launches; sum over reads of L+1  */
  int64_t lattice_bytes;          /* 8*(D+2)*N*columns (algorithmic)  */
  int64_t rounds;                 /* relaxation rounds, summed        */
  int64_t checkpointed_reads;     /* reads decoded in segments (bounded-memory decode: option checkpoint=) */
} dnas_batch_stats;
int dnas_model_last_stats(const dnas_model *model, dnas_batch_stats *out);

/* Copy one read's lattice out of the arena after a single-read batch (testing aid):
 * layout [pos][lane][n_states], lanes S, D, T1..TD. */
int dnas_model_read_lattice(dnas_model *model, int64_t slot, int64_t len, double *out);

/* ---- forward-backward path ------------------------------------------------------------ */

/*
 * expectedCounts(params, db, ll, strict) (reference src/fwdback.cpp:190-209): the E-step of the
 * mutator pair-HMM over a database of (original, read) alignment pairs, one GPU thread per pair.
 * Pairs are concatenated: pair i is in_seqs[in_off[i]..in_off[i+1]) / out_seqs[...] (bases 0..3)
 * with the guide alignment given per sequence position as the cumulative match count at that
 * position's alignment column (GuideAlignmentEnvelope, alignpath.h:35-54):
 *   cm_in[cm_in_off[i] + ip],  ip = 0..inLen;   cm_out[cm_out_off[i] + op],  op = 0..outLen.
 * out_counts[21 + n_len]: nDelOpen, nTanDup, nNoGap, nDelExtend, nDelEnd, nSub[4][4], nLen[]
 * (MutatorCounts, mutator.h:43-50); *out_ll = sum of forward log-likelihoods; out_pair_ll
 * (optional, n_pairs) the per-pair values.  Host pointers.
 */
int dnas_fwdback_estep(const dnas_mutator_params *params, int strict, int64_t n_pairs, const int8_t *in_seqs,
                       const int64_t *in_off, const int8_t *out_seqs, const int64_t *out_off, const int32_t *cm_in,
                       const int64_t *cm_in_off, const int32_t *cm_out, const int64_t *cm_out_off, int device_id,
                       double *out_counts, double *out_ll, double *out_pair_ll);

/* The same E-step behind a persistent handle: the log-sum-exp table (per device) and the database (per load) go to
 * the GPU once; dnas_fb_estep then runs on the handle's own stream with buffers it keeps -- the EM loop calls it up
 * to 100 times.  Pairs whose envelope rows are at most 32 cells wide (and n_len <= 8) are served by the wavefront
 * kernels (a group of 8, 16 or 32 lanes per pair, neighbours by lane shuffle); the rest by the streaming kernel.  dnas_fwdback_estep and
 * dnas_baum_welch are built on this. */
typedef struct dnas_fb dnas_fb;
typedef struct dnas_fb_stats {
  double kernel_ms;                 /* E-step kernels of the last call (HIP events on the handle's stream) */
  int64_t pairs_onchip, pairs_streaming;
  int64_t lse_ops;                  /* log_sum_exp evaluations of the on-chip kernel (counted in the kernel) */
  int64_t out_nt;                   /* sum of the read (output) lengths */
  int64_t pairs_narrow;             /* ... of pairs_onchip: served with half as many lanes as the row capacity (alignments that run down a diagonal) */
} dnas_fb_stats;
int dnas_fb_create(int device_id, dnas_fb **out);
int dnas_fb_load_pairs(dnas_fb *h, int64_t n_pairs, const int8_t *in_seqs, const int64_t *in_off, const int8_t *out_seqs,
                       const int64_t *out_off, const int32_t *cm_in, const int64_t *cm_in_off, const int32_t *cm_out,
                       const int64_t *cm_out_off);
int dnas_fb_estep(dnas_fb *h, const dnas_mutator_params *params, int strict, double *out_counts, double *out_ll,
                  double *out_pair_ll);
int dnas_fb_last_stats(const dnas_fb *h, dnas_fb_stats *out);
void dnas_fb_destroy(dnas_fb *h);

/* baumWelchParams(init, Laplace prior, db, strict) (fwdback.cpp:211-230, dnastore.cpp:135-140):
 * EM on the host around the GPU E-step; at most 100 iterations, stops when the relative gain
 * of log(likelihood * prior) drops below 1e-3. */
int dnas_baum_welch(const dnas_mutator_params *init, int strict, int64_t n_pairs, const int8_t *in_seqs,
                    const int64_t *in_off, const int8_t *out_seqs, const int64_t *out_off, const int32_t *cm_in,
                    const int64_t *cm_in_off, const int32_t *cm_out, const int64_t *cm_out_off, int device_id,
                    dnas_mutator_params *out, int32_t *out_iterations);

/* Stockholm database of two-row (original, read) alignments, readStockholmDatabase + Alignment +
 * GuideAlignmentEnvelope (stockholm.cpp:154-167, alignpath.cpp:189-204,237-265), flattened into the
 * arrays dnas_fwdback_estep takes. */
typedef struct dnas_pairs dnas_pairs;
typedef struct dnas_pairs_view {
  int64_t n_pairs;
  const int8_t *in_seqs;  const int64_t *in_off;
  const int8_t *out_seqs; const int64_t *out_off;
  const int32_t *cm_in;   const int64_t *cm_in_off;
  const int32_t *cm_out;  const int64_t *cm_out_off;
} dnas_pairs_view;
int dnas_stockholm_read(const char *path, dnas_pairs **out);
const dnas_pairs_view *dnas_pairs_get(const dnas_pairs *p);
void dnas_pairs_free(dnas_pairs *p);

/* The JSON the reference prints for --fit-error (MutatorParams::writeJSON, mutator.cpp:6-16) and
 * --error-counts (MutatorCounts::writeJSON, mutator.cpp:108-124), NUL-terminated into buf. */
int dnas_mutator_params_json(const dnas_mutator_params *p, char *buf, size_t cap);
int dnas_mutator_counts_json(const double *counts, int32_t n_len, char *buf, size_t cap);

/* ---- convenience: the whole reference call ------------------------------------------ */

typedef struct dnas_decoded dnas_decoded; /* vguard<FastSeq> result of decodeFastSeqs */
/* decodeFastSeqs(filename, machine, params) (viterbi.cpp:306-320) on GPU `device_id`; device_id = -1: on every GPU of
 * the node -- the reads of the file are dealt over the devices (by length, snake order), one host thread and one
 * model per device, results in file order: the serial loop of viterbi.cpp:312-318 has no dependence between reads.
 * dnas_decode_fastseqs_ex with want_events != 0 also keeps what the traceback found (the reference's level-3 log
 * messages, viterbi.cpp:266-293): per read a list of 64-bit events in the order the traceback met them,
 * type << 62 | position << 32 | payload -- 1: substitution at position, payload = emitted base << 2 | read base;
 * 2: deletion between position-1 and position, payload = the deleted base; 3: duplication at position, payload =
 * count << 26 | the duplicated bases, 2 bits each, first one in the highest bits (a model with more than 13 duplication
 * lanes cannot log them: dnas_model_set_event_log then returns DNAS_E_UNSUPPORTED). */
int dnas_decode_fastseqs(const char *fasta_path, const dnas_machine *m, const dnas_mutator_params *p,
                         int device_id, dnas_decoded **out);
int dnas_decode_fastseqs_ex(const char *fasta_path, const dnas_machine *m, const dnas_mutator_params *p,
                            int device_id, int want_events, dnas_decoded **out);
const char *dnas_decoded_tier(const dnas_decoded *d);   /* which fill kernel served the machine ("tier A: ...") */
int dnas_decoded_devices(const dnas_decoded *d);         /* how many devices shared the reads */
int64_t dnas_decoded_events(const dnas_decoded *d, int64_t i, const uint64_t **events);
/* GPUs visible to the library. */
int dnas_device_count(void);
int64_t dnas_decoded_count(const dnas_decoded *d);
const char *dnas_decoded_name(const dnas_decoded *d, int64_t i);
const char *dnas_decoded_seq(const dnas_decoded *d, int64_t i);
double dnas_decoded_loglike(const dnas_decoded *d, int64_t i);
void dnas_decoded_free(dnas_decoded *d);

/* FASTA/FASTQ(.gz) reader, readFastSeqs (fastseq.cpp:123-148): names + sequences. */
typedef struct dnas_fastseqs dnas_fastseqs;
int dnas_fastseqs_read(const char *path, dnas_fastseqs **out);
int64_t dnas_fastseqs_count(const dnas_fastseqs *f);
const char *dnas_fastseqs_name(const dnas_fastseqs *f, int64_t i);
const char *dnas_fastseqs_seq(const dnas_fastseqs *f, int64_t i);
void dnas_fastseqs_free(dnas_fastseqs *f);

const char *dnas_last_error(void);
void dnas_free(void *p);
/* 1 when the library was built with its HIP kernels (always, for the shipped .so). */
int dnas_has_device_code(void);

#ifdef __cplusplus
}
#endif
#endif /* DNASTORE_AMD_H */
