/*
 * spacepharer_amd.h -- C ABI of the MI355X-native implementation of SpacePHARER's spacer -> phage-ORF search
 * hot path (the vendored MMseqs2 `prefilter`, `align` and `rescorediagonal` modules).
 *
 * Two levels:
 *
 *  (1) Module drop-ins.  Same argv as the reference modules receive (program and module name already
 *      stripped, lib/mmseqs/src/commons/Application.cpp:196), same on-disk DB triples in and out, return value
 *      = process exit status.  They replace
 *          int prefilter      (int argc, const char **argv, const Command&)   lib/mmseqs/src/CommandDeclarations.h:79
 *          int align          (int argc, const char **argv, const Command&)   lib/mmseqs/src/CommandDeclarations.h:5
 *          int rescorediagonal(int argc, const char **argv, const Command&)   lib/mmseqs/src/CommandDeclarations.h:86
 *      whose bodies are lib/mmseqs/src/prefiltering/Main.cpp:13-68, alignment/Main.cpp:12-32 and
 *      alignment/rescorediagonal.cpp:381-431.  INTEGRATION.md shows the three-line Command wrappers.
 *
 *  (2) In-memory search context.  What the module drivers, bench.py and the parity tests call: matrices ->
 *      target set (masking + k-mer index, built once, staged to HBM) -> query batch -> prefilter -> align.
 *      It corresponds to the objects the reference builds per process:
 *          Prefiltering::Prefiltering + getIndexTable   prefiltering/Prefiltering.cpp:22-221,499-537
 *          Prefiltering::runSplit / QueryMatcher        prefiltering/Prefiltering.cpp:709-867
 *          Alignment::run / Matcher::getSWResult        alignment/Alignment.cpp:255-549, Matcher.cpp:61-187
 *          doRescorediagonal                            alignment/rescorediagonal.cpp:45-379
 *
 * Plain pointers and sizes only; no memory ownership crosses the boundary (callers own every buffer they pass,
 * the context owns everything it allocates).  All functions return 0 on success and a negative code on error;
 * sp_last_error() describes the last failure of the calling thread.  There is no CPU fallback: every compute entry
 * point fails if no HIP device is available.
 */
#ifndef SPACEPHARER_AMD_H
#define SPACEPHARER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SP_OK 0
#define SP_ERR_ARG (-1)
#define SP_ERR_DEVICE (-2)
#define SP_ERR_STATE (-3)
#define SP_ERR_IO (-4)
#define SP_ERR_OVERFLOW (-5)
#define SP_ERR_UNSUPPORTED (-6)

/* ---- process level ---- */
int sp_init(int device_mask);              /* bit i = use HIP device i; 0 = device of LOCAL_RANK (or 0) */
/* the same on a thread of its own, plus the one-time device set-up of the process (HIP runtime, device context, this library's code object): returns at once; sp_init() /
   sp_ctx_create() wait for it.  A module front end calls it first and reads its DBs meanwhile (0.1-0.2 s of a fresh process); the reference has no counterpart. */
int sp_init_async(int device_mask);
void sp_shutdown(void);
const char *sp_last_error(void);
int sp_device_count(void);
int sp_device_memory(uint64_t *free_bytes, uint64_t *total_bytes);   /* HBM of the current device (hipMemGetInfo) */
const char *sp_version(void);

/* ---- (1) module drop-ins ---- */
int sp_prefilter_run(int argc, const char **argv);        /* <i:queryDB> <i:targetDB> <o:prefilterDB> [flags]              */
int sp_align_run(int argc, const char **argv);            /* <i:queryDB> <i:targetDB> <i:resultDB> <o:alignmentDB> [flags] */
int sp_rescorediagonal_run(int argc, const char **argv);  /* <i:queryDB> <i:targetDB> <i:resultDB> <o:resultDB> [flags]    */
/* prefilter + align fused, IN ADDITION to the three above: the argv of the reference's `search` workflow command for a sequence-sequence
   search (int search(int, const char **, const Command&), lib/mmseqs/src/CommandDeclarations.h:93; lib/mmseqs/src/workflow/Search.cpp),
   same alignment DB as `prefilter` followed by `align`, without the prefilter DB in between and with one resident context */
int sp_search_run(int argc, const char **argv);           /* <i:queryDB> <i:targetDB> <o:alignmentDB> <tmpDir> [flags]           */

/* ---- (1b) set-level steps around the search: host-side module drop-ins with the reference modules' argv.
   They need no device. ---- */
int sp_extractorfs_run(int argc, const char **argv);           /* <i:nuclDB> <o:orfDB> [flags]   replaces lib/mmseqs/src/util/extractorfs.cpp:18-160 (CommandDeclarations.h:42) */
int sp_translatenucs_run(int argc, const char **argv);         /* <i:nuclOrfDB> <o:aaDB> [flags]   replaces lib/mmseqs/src/util/translatenucs.cpp:16-115 (CommandDeclarations.h:117) */
int sp_reverseseqbycodon_run(int argc, const char **argv);     /* <i:nuclOrfDB> <o:reversedDB> [flags]   replaces src/util/ReverseSeqByCodon.cpp:13-62 (src/LocalCommandDeclarations.h) */
int sp_proteinaln2nucl_run(int argc, const char **argv);       /* <i:queryNuclDB> <i:targetNuclDB> <i:queryAADB> <i:targetAADB> <i:alnDB> <o:alnDB> [flags]   replaces lib/mmseqs/src/util/proteinaln2nucl.cpp:13-205 */
int sp_combineprotnuclaln_run(int argc, const char **argv);    /* <i:protAlnDB> <i:nuclAlnDB> <o:alnDB> [flags]   replaces src/util/CombineProtNuclAln.cpp:13-106 */
int sp_besthitperset_run(int argc, const char **argv);         /* <i:querySetDB> <i:targetSetDB> <i:resultDB> <o:resultDB> [flags]   replaces lib/mmseqs/src/multihit/besthitperset.cpp:122-129 */
int sp_mergeresultsbyset_run(int argc, const char **argv);     /* <i:setToMemberDB> <i:resultDB> <o:resultDB> [flags]   replaces lib/mmseqs/src/util/mergeresultsbyset.cpp:10-58 */
int sp_combinescore_run(int argc, const char **argv);          /* <i:querySetDB> <i:targetSetDB> <i:resultDB> <o:scoreDB> <tmpDir> [flags]   replaces src/util/CombineScore.cpp:106-113 */
int sp_filtermatchbyfdr_run(int argc, const char **argv);      /* <i:scoreDB> <i:controlScoreDB> <o:matchDB> [flags]   replaces src/util/FilterMatchbyFdr.cpp:13-264 */
int sp_summarizeresults_run(int argc, const char **argv);      /* <i:targetNuclDB> <i:matchDB> <i:alnDB> <o:report> [flags]   replaces src/util/SummarizeResults.cpp:13-197 */

/* ---- (2) in-memory search context ---- */
typedef struct sp_ctx sp_ctx;

typedef struct {
    const char *seed_sub_mat;      /* --seed-sub-mat (aa part), e.g. "VTML80.out"; name of an embedded table, a path, or NAME.out:DATA */
    const char *sub_mat;           /* --sub-mat (aa part), e.g. "VTML40.out" */
    float sensitivity;             /* -s            (5.7 in SpacePHARER, src/workflow/PredictMatch.cpp:9-31) */
    int kmer_score;                /* --k-score     (INT_MAX = derive from sensitivity) */
    int max_seqs;                  /* --max-seqs    (300) */
    int min_ungapped_score;        /* --min-ungapped-score (15) */
    int mask;                      /* --mask        (1 = tantan) */
    int gap_open;                  /* --gap-open aa part (16) */
    int gap_extend;                /* --gap-extend  (2)  */
    double evalue;                 /* -e            (200) */
    int threads;                   /* host threads for index construction / output formatting (0 = all) */
    unsigned bin_size;             /* 0 = derive from the host L2 size as the reference does (tie order at the --max-seqs cut) */
} sp_search_params;

void sp_default_params(sp_search_params *p);   /* SpacePHARER's predictmatch defaults */

typedef struct { uint32_t seq_id; int32_t score; int16_t diagonal; int16_t pad; } sp_hit;   /* hit_t, QueryMatcher.h:35-50 */

typedef struct {             /* Matcher::result_t, alignment/Matcher.h:32-90 (fields that reach the output DB) */
    uint32_t db_key;
    int32_t bit_score;
    int32_t raw_score;
    float seq_id;
    double evalue;
    int32_t q_start, q_end; uint32_t q_len;
    int32_t db_start, db_end; uint32_t db_len;
    uint32_t aln_len;
    uint32_t cigar_offset;   /* into the cigar pool returned next to the results (uncompressed M/I/D string, NUL-terminated) */
    int32_t passed;          /* Alignment::checkCriteria */
} sp_alignment;

typedef struct {             /* DistanceCalculator::LocalAlignment + statistics, rescorediagonal.cpp:227-252 */
    int32_t score, start_pos, end_pos, diagonal_len, diagonal; uint32_t dist_to_diagonal;
    double evalue; int32_t bit_score;
} sp_rescore;

sp_ctx *sp_ctx_create(const sp_search_params *p);
void sp_ctx_destroy(sp_ctx *c);

/* target set: ASCII residues concatenated, offsets[n+1]; keys may be NULL (key = index).  Builds the masked
   sequence store and the k-mer index on the host and stages both into HBM. */
int sp_ctx_set_targets(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n);
/* flags: bit 0 = do not build / stage the k-mer index (align and rescorediagonal only need the residues) */
int sp_ctx_set_targets_ex(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n, int flags);
/* c searches the target set `from` has staged: device tables and target set are shared between the contexts (same device, same matrices / sensitivity / masking), not
 * copied or rebuilt.  Several search contexts -- one per host thread, each with its own queries, streams and result buffers -- then hold ONE copy of the similar-k-mer
 * tables, the residues and the index.  The reference shares these the same way between its OpenMP threads (one IndexTable / SequenceLookup / ScoreMatrix per process,
 * prefiltering/Prefiltering.cpp:499-537; a QueryMatcher per thread, :716-740). */
int sp_ctx_share_targets(sp_ctx *c, sp_ctx *from);
/* query batch: ASCII residues concatenated, offsets[n+1]; uploaded once, stays resident until replaced */
int sp_ctx_set_queries(sp_ctx *c, const char *residues, const uint64_t *offsets, uint64_t n);

/* prefilter all resident queries; results stay on the device until fetched */
int sp_ctx_prefilter(sp_ctx *c);
/* hits: n_queries * max_seqs entries, counts: n_queries entries; seq_id = 0-based target index */
int sp_ctx_get_hits(sp_ctx *c, sp_hit *hits, uint32_t *counts);
int sp_ctx_max_seqs(const sp_ctx *c);
/* compact form: total = sp_ctx_num_hits(); hits[total] grouped by query in query order, offsets[n_queries + 1] */
uint64_t sp_ctx_num_hits(const sp_ctx *c);
int sp_ctx_get_hits_compact(sp_ctx *c, sp_hit *hits, uint64_t *offsets);
/* the same list without a copy: the page-locked buffers the device delivered it to (hits[*n] grouped by query,
   offsets[n_queries + 1]); the pointers stay valid until the next prefilter / merge call on this context */
typedef struct { uint32_t seq_id; uint16_t score; int16_t diagonal; } sp_hit_packed;   /* 8 bytes */
int sp_ctx_hits_view(sp_ctx *c, const sp_hit_packed **hits, const uint32_t **offsets, uint64_t *n);
/* deliver the hit lists of the FOLLOWING passes into caller-owned memory (page-locked: sp_host_alloc) instead of the context's
   own staging buffers: hits[cap_hits], offsets[n_queries + 1].  sp_ctx_hits_view() then returns these pointers; a pass with
   more than cap_hits hits falls back to the internal buffers (check the pointer).  NULL switches back.  Lets a caller keep
   the results of several passes (one buffer set per pass) without copying them. */
int sp_ctx_set_hit_sink(sp_ctx *c, sp_hit_packed *hits, uint32_t *offsets, uint64_t cap_hits);

/* align (query index, target index) pairs of the resident batch.  results[n]; cigars: caller buffer of cigar_cap
   bytes, *cigar_used returns the bytes needed/used (SP_ERR_OVERFLOW if cigar_cap is too small). */
int sp_ctx_align(sp_ctx *c, const uint32_t *query_idx, const uint32_t *target_idx, uint64_t n, sp_alignment *results,
                 char *cigars, uint64_t cigar_cap, uint64_t *cigar_used);
/* align every hit of the last sp_ctx_prefilter() without a host round trip in between; results[i] belongs to hit i of
   sp_ctx_get_hits_compact() (n = sp_ctx_num_hits()) */
int sp_ctx_align_hits(sp_ctx *c, sp_alignment *results, char *cigars, uint64_t cigar_cap, uint64_t *cigar_used);
/* sp_ctx_align_hits() in two halves: _begin returns when the kernels are done and the results are travelling to `results` /
   `cigars` (page-locked buffers from sp_host_alloc(); do not touch them), _end waits for them.  In between the caller may
   run the next sp_ctx_prefilter(): the copy overlaps it. */
int sp_ctx_align_hits_begin(sp_ctx *c, sp_alignment *results, char *cigars, uint64_t cigar_cap, uint64_t *cigar_used);
int sp_ctx_align_hits_end(sp_ctx *c);
/* A whole search step of the resident query batch -- sp_ctx_prefilter() + sp_ctx_align_hits() -- queued WITHOUT a host round trip (the
   reference's prefilter + align module pair for one batch, lib/mmseqs/src/prefiltering/Prefiltering.cpp:717-872 +
   lib/mmseqs/src/alignment/Alignment.cpp:255-549): the hit count stays on the device, every kernel of the align half reads it there, the hit
   list (hits[n], offsets[n_queries + 1]), the records (results[i] belongs to hit i) and the packed backtraces travel on a copy stream into the
   caller's page-locked buffers (sp_host_alloc; hits and results hold hit_cap entries, cigars cigar_cap bytes).  _begin returns as soon as the
   step is queued; up to two steps may be in flight per context (the second one's kernels run while the first one's results travel); _end
   waits for the OLDEST step's kernels, sends its results on their way (from device buffers the younger step does not write) and reports its counts.  Leave the buffers alone between _begin and _end; do not call sp_ctx_set_queries /
   sp_ctx_set_targets or the synchronous entry points while a step is in flight.  A step that meets a capacity (more hits than earlier passes
   delivered, a traceback that needs the host path) is repeated the synchronous way inside _end: the results are always complete. */
int sp_ctx_search_begin(sp_ctx *c, sp_hit_packed *hits, uint32_t *offsets, uint64_t hit_cap, sp_alignment *results, char *cigars, uint64_t cigar_cap);
int sp_ctx_search_end(sp_ctx *c, uint64_t *n_hits, uint64_t *cigar_used);
/* ---- multi-GPU: one process per GPU of a node (the reference's counterpart is its MPI mode: per-rank result files merged by rank 0,
 * lib/mmseqs/src/prefiltering/Prefiltering.cpp:258-362, slices by commons/DBReader.cpp:1171-1213) ----
 * The ranks of a job meet through a 128-byte id: one of them (or their launcher) makes it with sp_dist_unique_id and hands it to all of them by whatever control plane is
 * at hand (environment, file, MPI, torch store); sp_dist_init is collective.  SP_TRANSPORT_RCCL: device buffers travel by RCCL all-gather over xGMI, one rank per GPU
 * (the device this process selected with sp_init).  SP_TRANSPORT_HOST: through host memory and the bootstrap sockets -- for ranks that share a GPU (tests on a one-GPU
 * box; RCCL refuses two ranks on a device).  Small host values (counts, timings) always use the sockets. */
typedef struct sp_comm sp_comm;
#define SP_DIST_ID_BYTES 128
#define SP_TRANSPORT_RCCL 0
#define SP_TRANSPORT_HOST 1
int sp_dist_unique_id(unsigned char id[SP_DIST_ID_BYTES]);
sp_comm *sp_dist_init(const unsigned char id[SP_DIST_ID_BYTES], int rank, int world, int transport);      /* NULL on failure (sp_last_error) */
void sp_dist_destroy(sp_comm *comm);
int sp_dist_rank(const sp_comm *comm);
int sp_dist_world(const sp_comm *comm);
/* the transport the job ended up with: SP_TRANSPORT_RCCL, or SP_TRANSPORT_HOST -- asked for, or agreed on by all ranks because RCCL could not be set up on one of them
   (SP_DIST_STRICT=1 in the environment: sp_dist_init fails instead) */
int sp_dist_transport(const sp_comm *comm);
int sp_dist_barrier(sp_comm *comm);
/* equal-size all-gather: recv holds world x bytes, rank r's block at r x bytes; on_device: both buffers are memory of this rank's GPU */
int sp_dist_allgather(sp_comm *comm, const void *send, uint64_t bytes, void *recv, int on_device);
int sp_dist_allreduce_f64(sp_comm *comm, double *values, uint64_t n, int op /* 0 sum, 1 max, 2 min */);
/* every rank sends block r (send[r], send_bytes[r]) to rank r and receives rank r's block for it into recv[r] (recv_bytes[r], agreed on beforehand: e.g. by an
   all-gather of the sizes); device buffers over RCCL: one group of ncclSend / ncclRecv pairs; otherwise through rank 0 over the bootstrap sockets.  The probe lists of
   sp_ctx_probe_search travel this way. */
int sp_dist_alltoallv(sp_comm *comm, const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes, int on_device);
uint64_t sp_dist_bytes_moved(const sp_comm *comm);      /* bytes this rank has received through the communicator so far */
/* DBReader::decomposeDomainByAminoAcid: contiguous slices balanced by data size; entry_lengths = the index `length` fields (sequence length + 2) */
int sp_dist_decompose(const uint64_t *entry_lengths, uint64_t n, int rank, int world, uint64_t *start, uint64_t *count);
/* Target-sharded search (every rank staged ITS target range with sp_ctx_set_targets_shard and holds ALL queries): after sp_ctx_prefilter_candidates, the shards'
 * candidate lists are all-gathered (HBM to HBM) and the selection runs on the union for queries [q_from, q_to) -- the hits equal the unsplit search's.  exchange_s:
 * wall time of the exchange (counts + payload), union_rows: rows of the gathered list incl. padding. */
int sp_ctx_allgather_candidates(sp_ctx *c, sp_comm *comm, uint64_t q_from, uint64_t q_to, uint64_t *union_rows, double *exchange_s);
/* Query-partitioned search (every rank staged the whole target set and holds ITS slice of the queries, starting at global query `query_offset`): the slices' hit lists
 * are all-gathered; every rank receives all hits in global query order (offsets has nq_total + 1 entries).  my_hits / my_offsets: this rank's list as a search step
 * delivered it (sp_ctx_search_end, sp_ctx_hits_view: offsets of the resident queries + 1); both NULL: the last sp_ctx_prefilter's list, taken where it lies in HBM. */
int sp_ctx_allgather_hits(sp_ctx *c, sp_comm *comm, uint64_t query_offset, uint64_t nq_total, const sp_hit_packed *my_hits, const uint32_t *my_offsets,
                          sp_hit_packed *hits, uint64_t *offsets, uint64_t hit_cap, uint64_t *n_hits);

/* rescorediagonal core (alignment/DistanceCalculator.h:93-295): mode 0 = equal letters on the diagonal (the module's default), 1 = ungapped local score,
 * 2 = score + start/end, 3 = the whole diagonal (start/end = its first / last column, a '*' at either end skipped), 4 = the longest stretch with at most five
 * mismatches in any 20 columns.  evalue / bit_score are those of `score` (modes 1-4; the module ignores them in mode 0). */
int sp_ctx_rescore(sp_ctx *c, const uint32_t *query_idx, const uint32_t *target_idx, const int16_t *diagonal, uint64_t n,
                   int mode, sp_rescore *results);

/* ---- the chain after the search without DB round trips (SURVEY.md section 8 rows f1 + f2; data/predictmatch.sh:27-75) ----
 * Takes the alignment records of the last sp_ctx_prefilter() + sp_ctx_align_hits*() as they lie in host memory (aln[i] / cigars belong
 * to hit i) through: align-module text (Alignment.cpp:255-549) -> proteinaln2nucl (util/proteinaln2nucl.cpp:99-187) ->
 * combineprotnuclaln (src/util/CombineProtNuclAln.cpp:55-97) -> besthitperset (multihit/besthitperset.cpp:29-115) ->
 * mergeresultsbyset (util/mergeresultsbyset.cpp:32-50) -> combinescore (src/util/CombineScore.cpp:41-98).  The resident queries and
 * targets are the peptides of nucleotide ORFs; the layout says which ORF and which set each belongs to.  Set sizes are the member
 * counts (what createsetdb's result2stats counts), the nucleotide E-values use the sum of the target ORF lengths.
 * stage selects what is returned: 0 = combined score per query set (the cScore DB), 1 = align text, 2 = nucleotide alignments,
 * 3 = combined alignments, 4 = best hit per target set (1-4: one entry per query, key = query index), 5 = best hits merged per
 * query set.  Entries are what the reference modules write into the respective DB, NUL-terminated. */
typedef struct {
    const char *q_nucl; const uint64_t *q_nucl_off;   /* nucleotide ORF of resident query i: q_nucl[q_nucl_off[i] .. q_nucl_off[i+1]) */
    const char *t_nucl; const uint64_t *t_nucl_off;   /* nucleotide ORF of target i */
    const uint32_t *q_set;                            /* set key of query i (spacer array / file) */
    const uint32_t *t_set;                            /* set key of target i (phage genome) */
    const char *nucl_sub_mat;                         /* NULL = "nucleotide.out" */
    int nucl_gap_open, nucl_gap_extend;               /* 0 = 10 / 2 (predictmatch) */
    int simple_best_hit;                              /* --simple-best-hit (predictmatch: 1) */
} sp_set_layout;
typedef struct { uint64_t n; uint32_t *keys; uint64_t *offsets; char *text; } sp_text_db;   /* entry i: text + offsets[i], key keys[i]; offsets[n] = bytes */
int sp_ctx_set_scores(sp_ctx *c, const sp_set_layout *l, const sp_alignment *aln, const char *cigars, int stage, sp_text_db *out);
void sp_text_db_free(sp_text_db *db);

/* A sequence DB (MMseqs2 triple <path>, <path>.index, <path>.dbtype; split data files tolerated) read by the library's own reader and flattened into what
   sp_ctx_set_targets / sp_ctx_set_queries / sp_set_layout take: residues back to back (entry terminators removed), offsets[n + 1], keys[n], entries in the order of the data file (DBReader's LINEAR_ACCCESS order) -- and
   sets[n], the third column of <path>.lookup ("key\tname\tset", commons/DBReader.h) per entry, when that file exists (else NULL).  The arrays belong to the handle. */
typedef struct sp_flat_db sp_flat_db;
sp_flat_db *sp_flat_db_open(const char *path);      /* NULL + sp_last_error() on failure */
void sp_flat_db_close(sp_flat_db *f);
uint64_t sp_flat_db_entries(const sp_flat_db *f);
const char *sp_flat_db_residues(const sp_flat_db *f);
const uint64_t *sp_flat_db_offsets(const sp_flat_db *f);
const uint32_t *sp_flat_db_keys(const sp_flat_db *f);
const uint32_t *sp_flat_db_sets(const sp_flat_db *f);

/* ---- target-sharded search (one context per GPU) ----
 * Every GPU holds all target residues (align needs them) but masks and indexes only targets [shard_from, shard_to);
 * sequence ids stay global.  Per query batch: sp_ctx_prefilter_candidates() on every shard, gather the candidate lists of
 * all shards (RCCL all-gather of the device buffers, or any host transport), then sp_ctx_merge_candidates() runs the
 * selection tail of QueryMatcher::matchQuery (prefiltering/QueryMatcher.cpp:100-186) over the union -- the result equals
 * the unsplit search bit for bit (unlike Prefiltering::mergeTargetSplits, prefiltering/Prefiltering.cpp:300-420, whose
 * per-split quota changes the hit set).  Afterwards the context is in the state sp_ctx_prefilter() leaves:
 * sp_ctx_get_hits_compact() / sp_ctx_align_hits() work, restricted to queries [q_from, q_to). */
typedef struct { uint32_t query; uint32_t seq_id; uint32_t kmer_ordinal; uint16_t diagonal; uint16_t score; } sp_candidate;   /* 16 bytes */
int sp_ctx_set_targets_shard(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n,
                             uint64_t shard_from, uint64_t shard_to);
int sp_ctx_prefilter_candidates(sp_ctx *c);
uint64_t sp_ctx_num_candidates(const sp_ctx *c);
int sp_ctx_get_candidates(sp_ctx *c, sp_candidate *out);            /* copy of this shard's candidates; `out` may be host or
                                                                       device memory (e.g. the send buffer of the all-gather) */
const void *sp_ctx_candidates_device(const sp_ctx *c);              /* the same list in HBM (sp_candidate[num]) */
/* candidates: the union (host or device memory, on_device says which); records with query == 0xFFFFFFFF are padding */
int sp_ctx_merge_candidates(sp_ctx *c, const sp_candidate *candidates, uint64_t n, int on_device, uint64_t q_from, uint64_t q_to);
/* ---- the probe-list cut of the target-sharded search (opt-in; queries of up to 64 residues; at most SP_MAX_PROBE_SHARDS shards) ----
 * sp_ctx_prefilter_candidates() makes every shard enumerate the similar k-mers of EVERY query -- the same work on every GPU, and most of the matching kernel's time
 * (the reference's MPI target split does the same: prefiltering/Prefiltering.cpp:258-362, each rank runs QueryMatcher::match over all queries against its index split).
 * With the cut a query is enumerated once: every rank takes a slice of the batch, tests its similar k-mers against the union of the shards' occupancy maps
 * (SP_SHARD_MAP_BYTES each: one bit per k-mer slot) and sends every occupied probe -- a 16-byte sp_probe: query, position of the probe in the query's list on that
 * shard, ordinal of the k-mer in the query's enumeration, k-mer slot | window position << 26 -- to the shards whose own map holds the
 * slot, together with the number of probes per query; a shard places the probes it received by (query, position) -- no sort --, fetches their index lists and detects
 * the double hits.  It ends where sp_ctx_prefilter_candidates() ends (same candidates, bit for bit); the candidates' exchange and sp_ctx_merge_candidates() follow.
 *   sp_ctx_shard_occupancy     this shard's map -> out (host or device memory)
 *   sp_ctx_probe_set_shards    all shards' maps, in shard order, n_shards x SP_SHARD_MAP_BYTES
 *   sp_ctx_probe_enumerate     queries [q_from, q_to) of the resident batch -> probe lists by destination shard; counts[n_shards] = records per list
 *                              (incl. padding records, query 0xFFFFFFFF, that the fetch side skips)
 *   sp_ctx_probe_get           list `dest` -> records[count] and query_counts[q_to - q_from] (host or device memory); sp_ctx_probe_records_device /
 *                              sp_ctx_probe_query_counts_device: where they lie in HBM
 *   sp_ctx_probe_fetch         the lists addressed to this shard, back to back in any order of their senders, and query_counts[n_queries]: the senders' counts
 *                              in the order of their slices (the slices tile the batch) -> this shard's candidates
 *   sp_ctx_probe_search        collective, the steps above for rank r = shard r of `comm` (maps all-gathered once per target set, lists by all-to-all:
 *                              RCCL send/recv pairs over xGMI, or through host memory); exchange_s: wall time of the exchanges */
#define SP_SHARD_MAP_BYTES 8000000u
#define SP_MAX_PROBE_SHARDS 8
typedef struct { uint32_t query; uint32_t position; uint32_t kmer_ordinal; uint32_t slot_window; } sp_probe;      /* 16 bytes; slot_window = k-mer slot | window position << 26 */
int sp_ctx_shard_occupancy(sp_ctx *c, void *out, int on_device);
int sp_ctx_probe_set_shards(sp_ctx *c, const void *maps, int n_shards, int on_device);
int sp_ctx_probe_enumerate(sp_ctx *c, uint64_t q_from, uint64_t q_to, uint64_t *counts);
int sp_ctx_probe_get(sp_ctx *c, int dest, sp_probe *records, uint32_t *query_counts, int on_device);
const void *sp_ctx_probe_records_device(const sp_ctx *c, int dest);
const void *sp_ctx_probe_query_counts_device(const sp_ctx *c, int dest);
int sp_ctx_probe_fetch(sp_ctx *c, const sp_probe *records, uint64_t n, const uint32_t *query_counts, int on_device);
int sp_ctx_probe_search(sp_ctx *c, sp_comm *comm, uint64_t q_from, uint64_t q_to, double *exchange_s);
/* After sp_ctx_prefilter() / sp_ctx_prefilter_candidates() returned SP_ERR_OVERFLOW: the positions (in the batch) of the queries whose
 * hit stream does not fit one pass (more than 65 535 index hits; QueryMatcher's own buffer, prefiltering/QueryMatcher.cpp:39-50, holds
 * 2 x 10^6 and is evaluated in place when it fills, :254-283).  Returns their number, writes at most cap of them.  The module drop-ins
 * search the batch again without them and take them through target shards (exact union), so a module call never fails on them. */
uint64_t sp_ctx_overflow_queries(const sp_ctx *c, uint32_t *out, uint64_t cap);

/* counters / timings of the last call, by name: "kmers", "db_matches", "diagonals", "diag_residues",
   "hits", "index_entries", "masked_residues", "target_residues", "kmer_thr", "bin_size",
   "ms_prep", "ms_match", "ms_match_active" (the matching kernel from its first wave's start to its last wave's end, by the device wall clock),
   "ms_score", "ms_select", "ms_sw_forward", "ms_sw_reverse", "ms_traceback", "ms_assemble",
   "ms_index_build", "ms_upload" */
double sp_ctx_stat(const sp_ctx *c, const char *name);
/* Result-DB entries as TEXT, formatted on the device (csrc/sp_gpu_text.inc) -- what the `prefilter` and `align` / `search` modules write into their data files:
   one entry per query of the resident batch, in query order, each its lines followed by a NUL; entry_len[q] (n_queries values) = bytes of entry q, *text_used = their sum.
   text / entry_len: caller's buffers (page-locked for speed: sp_host_alloc).  SP_ERR_OVERFLOW: text_cap is too small, *text_used says what is needed (nothing copied).
     sp_ctx_hits_text        after sp_ctx_prefilter(): "<target key>\t<score>\t<diagonal>\n" per hit (QueryMatcher::prefilterHitToBuffer, prefiltering/QueryMatcher.h:114-126)
     sp_ctx_align_hits_text  aligns the resident hits like sp_ctx_align_hits() and writes, per query, the accepted alignments (Alignment::checkCriteria with
                             min_seq_id / min_aln_len, alignment/Alignment.cpp:555-574) sorted by Matcher::compareHits as Matcher::resultToBuffer lines with
                             backtrace (alignment/Matcher.cpp:211-230,323-370).  SP_ERR_UNSUPPORTED: this batch needs the host path (sp_ctx_align_hits + own formatting).
   Either call returns SP_ERR_UNSUPPORTED, before writing anything, when the batch's entries would add up to 4 GiB of text or more (the device keeps 32-bit offsets). */
/* ... and the INDEX text of those entries (commons/DBWriter.cpp:409-419: "<key>\t<offset>\t<length>\n" per entry, keys[e] = key of the batch's e-th query, offsets counted
   from `base`), formatted on the device from the lengths the last sp_ctx_hits_text / sp_ctx_align_hits_text call left there.  Same buffer rules. */
int sp_ctx_text_index(sp_ctx *c, const uint32_t *keys, uint64_t n, uint64_t base, char *text, uint64_t text_cap, uint64_t *text_used);
int sp_ctx_hits_text(sp_ctx *c, char *text, uint64_t text_cap, uint64_t *text_used, uint32_t *entry_len);
int sp_ctx_align_hits_text(sp_ctx *c, double min_seq_id, uint32_t min_aln_len, char *text, uint64_t text_cap, uint64_t *text_used, uint32_t *entry_len);
/* wait for everything queued on the context's streams (search + copy); SP_ERR_DEVICE with the HIP error text in sp_last_error() when a kernel or copy failed */
int sp_ctx_sync(sp_ctx *c);
/* HIP-event timing inside the pipeline (sp_ctx_stat "ms_*"): 0 = none, 1 = the k-mer matching kernel only (default; every
   event record between two kernels costs ~6 us of idle GPU), 2 = every stage.  Stages that are not timed report NaN. */
int sp_ctx_set_timing(sp_ctx *c, int level);
/* tuning knobs (never change a result).  "match_blocks_per_cu": resident blocks of the matching kernel per CU, 0 = as many as
   fit.  Fewer leave LDS for the aligner of another context on the same GPU -- better when several passes are in flight.
   "match_small": footprint of the matching kernel, -1 by index density (default), 0 large, 1 small.  "match_dense2": 1 = a dense index (target ids
   below 2^24) is searched by k_match_dense (64-byte double buckets, 5-byte hit stream; round 6, same records, slower) instead of k_match.  "match_heavy_threshold": similar
   k-mers from which a query is started first, -1 by batch size (default), 0 no list.  "match_weigh_exact": 1 = those k-mers counted exactly
   (two table rows read per window) instead of estimated from the window's slack over its threshold (default).  "match_chunk" (1..32) /
   "match_static" (0..8): queries per pop of the dynamically dealt part / eighths of the batch dealt round-robin.  "step_capacity": pairs the
   next sp_ctx_search_begin() is sized for (0 = estimate again). */
int sp_ctx_set_tuning(sp_ctx *c, const char *name, int value);
/* raw HIP stream the context launches on (hipStream_t), for callers that time with their own events */
void *sp_ctx_stream(const sp_ctx *c);
/* test hooks: digest of the index built on the device ((k-mer, seqId, pos) triples in table order, FNV-1a as in
   sp_host_index_digest) and the masked target residues as letters (out: target_residues bytes) */
int sp_ctx_index_digest(sp_ctx *c, uint64_t *entries, uint64_t *digest);
int sp_ctx_ext3_digest(sp_ctx *c, uint64_t *digest);     /* the similar-3-mer table in HBM, hashed like sp_host_ext3_digest */
int sp_ctx_get_masked(sp_ctx *c, char *out);
/* page-locked host memory for result buffers (hits, alignment records, cigars): device-to-host copies into such memory
   run at full PCIe rate; ordinary memory works too, only slower.  Needs an initialised device (sp_init). */
void *sp_host_alloc(uint64_t bytes);
void sp_host_free(void *p);
/* profiling aid: one streaming read of `bytes` with lane_bytes (4, 8 or 16) per lane per load on the context's stream;
   tools/pmc_traffic.py uses it to calibrate rocprofv3's FETCH_SIZE before quoting it for the search kernels */
int sp_ctx_diag_stream_read(sp_ctx *c, uint64_t bytes, int lane_bytes, double *ms);

/* ---- host-side diagnostics: run the set-up code (matrices, masking, index construction, E-value statistics) without
   a device so that it can be checked on any machine; digests are FNV-1a over the same byte streams the golden
   fixtures hash (tests/test_host_logic.py) ---- */
int sp_host_matrix(const char *spec, double bit_factor, double score_bias, int16_t *sub441, double *p_back21);
int sp_host_ext3_digest(const char *seed_spec, uint64_t *digest, int16_t *score_head64, uint32_t *index_head64, uint32_t row);
int sp_host_index_digest(const char *seed_spec, float sensitivity, int mask, const char *residues, const uint64_t *offsets, uint64_t n,
                         uint64_t *entries, uint64_t *masked_residues, uint64_t *digest, char *masked_out);
int sp_host_evalue(const char *matrix_name, int gap_open, int gap_extend, uint64_t db_residues, double score, double q_len,
                   double *evalue, double *bit_score);

#ifdef __cplusplus
}
#endif
#endif
