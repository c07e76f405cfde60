// C ABI (include/spacepharer_amd.h): process-level state + the in-memory search context.
#include "../../include/spacepharer_amd.h"

#include <algorithm>
#include <chrono>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "sp_gpu.h"
#include "sp_dist.h"
#include <hip/hip_runtime_api.h>
#include "sp_host.h"
#include "sp_modules.h"
#include "sp_sets.h"

namespace {
thread_local std::string g_err;
int g_device = -1;
std::mutex g_mu;

double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

namespace sp {
void setLastError(const std::string &e) { g_err = e; }
int currentDevice() {
    if (g_device >= 0) return g_device;
    const char *lr = getenv("LOCAL_RANK");
    return lr ? atoi(lr) : 0;
}
}  // namespace sp

struct sp_ctx {
    sp_search_params par;
    std::string seedName, subName;
    sp::SubMat seed, ungapped, aln;
    sp::Gpu gpu;
    sp::TargetSet targets;
    uint64_t indexEntries = 0;
    int kmerThr = 0;
    unsigned binSize = 2;
    int threads = 1;
    std::vector<uint8_t> qres; std::vector<uint32_t> qoff; std::vector<char> qascii; std::vector<uint64_t> qasciiOff;
    sp::Evaluer evGapped, evGapless;
    bool haveGapped = false, haveGapless = false;
    std::map<std::string, double> stats;
    std::map<uint32_t, std::vector<uint8_t>> gateRows;     // E-value gate rows per query length (for this target set / -e)
    std::map<uint32_t, std::vector<double>> evalRows;      // evalue(score, qLen) per query length
    std::vector<int32_t> bitRow;                           // int(bitScore(score) + 0.5)
    bool tablesUp = false, kmerTablesUp = false;
    sp::GateTable gate; bool gateValid = false; uint64_t gateSerial = 0;    // E-value tables of the resident query batch / target set
    std::vector<char> targetFirst;                         // first letter of every target (proteinaln2nucl's '*' test)
    sp::StepOut stepOut[2]; uint64_t stepsBegun = 0, stepsEnded = 0;      // result buffers of the device-driven steps in flight (sp_ctx_search_begin / _end)
    sp::AlignHost alnHost;                                 // pinned staging, reused; declared last = released before the device context
};

static uint64_t fnv1a(const void *p, size_t n, uint64_t h) {
    const unsigned char *c = (const unsigned char *) p;
    if (h == 0) h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ULL; }
    return h;
}
// digest of a packed similar-3-mer table over the byte streams the reference holds (short score[8000], uint index[8000] per row)
static uint64_t ext3Digest(const std::vector<uint32_t> &packed, int16_t *scoreHead64, uint32_t *indexHead64, uint32_t row) {
    uint64_t h = 0; std::vector<int16_t> sc(8000); std::vector<uint32_t> ix(8000);
    for (size_t r = 0; r < 8000; r++) {
        for (size_t z = 0; z < 8000; z++) { sc[z] = (int16_t) ((int) (packed[r * 8000 + z] >> 16) - 32768); ix[z] = packed[r * 8000 + z] & 0xFFFF; }
        h = fnv1a(sc.data(), 16000, h); h = fnv1a(ix.data(), 32000, h);
        if (scoreHead64 && r == row) for (int z = 0; z < 64; z++) { scoreHead64[z] = sc[z]; indexHead64[z] = ix[z]; }
    }
    return h;
}

extern "C" {

const char *sp_last_error(void) { return g_err.c_str(); }
int sp_device_memory(uint64_t *free_bytes, uint64_t *total_bytes) {
    size_t f = 0, t = 0;
    if (!sp::deviceMemory(f, t)) { g_err = "sp_device_memory: hipMemGetInfo failed (no device initialised?)"; return SP_ERR_DEVICE; }
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return SP_OK;
}
const char *sp_version(void) { return "spacepharer_amd 0.1 (gfx950)"; }

static int sp_init_sync(int device_mask);
// sp_init_async: the same on a thread of its own, plus the process's one-time device set-up (runtime, context, code object); whoever needs the device next -- sp_init,
// sp_ctx_create -- waits for it.  A front end calls it first thing and reads its DBs meanwhile: 0.1-0.2 s of a fresh module process.
static std::mutex g_asyncMu; static std::condition_variable g_asyncCv; static bool g_asyncRunning = false;
static void waitForAsyncInit() { std::unique_lock<std::mutex> lk(g_asyncMu); g_asyncCv.wait(lk, [] { return !g_asyncRunning; }); }
int sp_init_async(int device_mask) {
    {   std::lock_guard<std::mutex> lk(g_asyncMu);
        if (g_asyncRunning) return SP_OK;
        g_asyncRunning = true; }
    std::thread([device_mask] {
        if (sp_init_sync(device_mask) == SP_OK) { std::string err; (void) sp::Gpu::warmUp(sp::currentDevice(), err); }      // (a failure shows when the context is created)
        { std::lock_guard<std::mutex> lk(g_asyncMu); g_asyncRunning = false; }
        g_asyncCv.notify_all();
    }).detach();
    return SP_OK;
}
int sp_init(int device_mask) { waitForAsyncInit(); return sp_init_sync(device_mask); }
static int sp_init_sync(int device_mask) {
    std::lock_guard<std::mutex> lk(g_mu);
    sp::Gpu probe;
    int n = probe.deviceCount();
    if (n <= 0) { g_err = "sp_init: no HIP device available (this library has no CPU path)"; return SP_ERR_DEVICE; }
    int dev = -1;
    if (device_mask) { for (int i = 0; i < 32; i++) if (device_mask & (1 << i)) { dev = i; break; } }
    if (dev < 0) { const char *lr = getenv("LOCAL_RANK"); dev = lr ? atoi(lr) : 0; }
    g_device = dev % n;
    return SP_OK;
}

void sp_shutdown(void) { sp::releaseCachedContexts(); std::lock_guard<std::mutex> lk(g_mu); g_device = -1; }

int sp_device_count(void) { sp::Gpu probe; return probe.deviceCount(); }

void sp_default_params(sp_search_params *p) {
    memset(p, 0, sizeof(*p));
    p->seed_sub_mat = "VTML80.out";
    p->sub_mat = "VTML40.out";
    p->sensitivity = 5.7f;
    p->kmer_score = INT_MAX;
    p->max_seqs = 300;
    p->min_ungapped_score = 15;
    p->mask = 1;
    p->gap_open = 16;
    p->gap_extend = 2;
    p->evalue = 200.0;
    p->threads = 0;
    p->bin_size = 0;
}

sp_ctx *sp_ctx_create(const sp_search_params *p) {
    waitForAsyncInit();
    std::unique_ptr<sp_ctx> c(new sp_ctx);
    c->par = *p;
    c->seedName = p->seed_sub_mat ? p->seed_sub_mat : "VTML80.out";
    c->subName = p->sub_mat ? p->sub_mat : "VTML40.out";
    c->par.seed_sub_mat = c->seedName.c_str();
    c->par.sub_mat = c->subName.c_str();
    c->threads = p->threads > 0 ? p->threads : sp::defaultThreads();
    std::string err;
    const bool laps = getenv("SP_MODULE_TIMES") != nullptr; const double tc0 = nowMs();
    // bit factors / score biases: prefiltering/Prefiltering.cpp:63-67, alignment/Alignment.cpp:157-162
    if (!c->seed.load(c->seedName, (double) 8.0f, (double) -0.2f, err) || !c->ungapped.load(c->subName, (double) 2.0f, (double) -0.2f, err) ||
        !c->aln.load(c->subName, (double) 2.0f, (double) 0.0f, err)) { g_err = err; return nullptr; }
    c->kmerThr = sp::kmerThreshold(p->sensitivity, p->kmer_score, sp::kKmer);
    const double tc1 = nowMs();
    if (!c->gpu.init(sp::currentDevice(), err)) { g_err = "sp_ctx_create: " + err + " (this library has no CPU path)"; return nullptr; }
    if (laps) fprintf(stderr, "[sp] sp_ctx_create: matrices %.3f ms, device context %.3f ms\n", tc1 - tc0, nowMs() - tc1);
    c->stats["kmer_thr"] = c->kmerThr;
    return c.release();
}

void sp_ctx_destroy(sp_ctx *c) { delete c; }

int sp_ctx_max_seqs(const sp_ctx *c) { return (int) std::min<uint64_t>(c->targets.n, (uint64_t) c->par.max_seqs); }

int sp_ctx_set_targets(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n) {
    return sp_ctx_set_targets_ex(c, residues, offsets, keys, n, 0);
}

static int setTargets(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n, int flags, uint64_t from, uint64_t to);

int sp_ctx_set_targets_ex(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n, int flags) {
    return setTargets(c, residues, offsets, keys, n, flags, 0, n);
}

int sp_ctx_set_targets_shard(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n, uint64_t from, uint64_t to) {
    if (from > to || to > n) { g_err = "sp_ctx_set_targets_shard: bad shard range"; return SP_ERR_ARG; }
    return setTargets(c, residues, offsets, keys, n, 0, from, to);
}

static int setTargets(sp_ctx *c, const char *residues, const uint64_t *offsets, const uint32_t *keys, uint64_t n, int flags, uint64_t from, uint64_t to) {
    if (c && c->stepsBegun != c->stepsEnded) { g_err = "sp_ctx_set_targets: a search step is in flight (sp_ctx_search_end first)"; return SP_ERR_STATE; }
    if (c) c->gpu.makeCurrent();
    if (!c || !residues || !offsets) { g_err = "sp_ctx_set_targets: null argument"; return SP_ERR_ARG; }
    const bool noIndex = (flags & 1) != 0;
    std::string err;
    double t0 = nowMs();
    if (!c->tablesUp || (!noIndex && !c->kmerTablesUp)) {      // matrices always; the similar-k-mer tables only for a set that will be searched
        if (!c->gpu.uploadTables(c->seed, c->ungapped, c->aln, !noIndex, err)) { g_err = err; return SP_ERR_DEVICE; }
        c->tablesUp = true; c->kmerTablesUp = c->kmerTablesUp || !noIndex;
    }
    double t1 = nowMs();
    // the host keeps the target directory only; residue classes, tantan masking and the k-mer index are made on the device
    sp::TargetSet &ts = c->targets;
    ts = sp::TargetSet();
    ts.n = (size_t) n;
    ts.offsets.assign(offsets, offsets + n + 1);
    ts.keys.resize((size_t) n);
    for (size_t i = 0; i < (size_t) n; i++) { ts.keys[i] = keys ? keys[i] : (uint32_t) i; ts.maxLen = std::max<unsigned>(ts.maxLen, (unsigned) (offsets[i + 1] - offsets[i])); }
    c->targetFirst.resize((size_t) n);
    for (size_t i = 0; i < (size_t) n; i++) c->targetFirst[i] = offsets[i + 1] > offsets[i] ? residues[offsets[i]] : '\0';
    sp::Gpu::TargetBuild tb;
    if (!c->gpu.buildTargets(residues, ts, c->seed, c->kmerThr, c->par.mask != 0 && !noIndex, !noIndex, (size_t) from, (size_t) to, tb, err)) { g_err = err; return SP_ERR_DEVICE; }
    ts.maskedCount = tb.maskedResidues;
    c->indexEntries = tb.indexEntries;
    double t2 = nowMs(), t3 = t2;
    c->stats["ms_mask_gpu"] = tb.msMask; c->stats["ms_index_gpu"] = tb.msIndex; c->stats["index_slots"] = (double) tb.indexSlots;
    c->binSize = c->par.bin_size ? c->par.bin_size : sp::binSizeFor((size_t) n, sp::hostL2CacheSize());
    c->stats["ms_ext_table"] = t1 - t0;
    c->stats["ms_index_build"] = t2 - t1;
    c->stats["ms_upload"] = t3 - t2;
    c->stats["index_entries"] = (double) c->indexEntries;
    c->stats["masked_residues"] = (double) c->targets.maskedCount;
    c->stats["target_residues"] = (double) c->targets.residues();
    c->stats["bin_size"] = c->binSize;
    c->haveGapped = c->haveGapless = false;
    c->gateRows.clear(); c->evalRows.clear(); c->bitRow.clear(); c->gateValid = false;
    return SP_OK;
}

/* c searches the target set `from` has staged: the device tables (256 MB of similar 3-mers, the matrices) and the set itself (residues, masked residues, k-mer index) are
 * SHARED between the two contexts, not copied or rebuilt -- four contexts of one process keep one copy in HBM and in the caches.  Both contexts must sit on the same device
 * and have been created with the same matrices, sensitivity and masking.  sp_ctx_set_targets on either context later gives that context a set of its own again. */
int sp_ctx_share_targets(sp_ctx *c, sp_ctx *from) {
    if (!c || !from) { g_err = "sp_ctx_share_targets: null argument"; return SP_ERR_ARG; }
    if (c == from) return SP_OK;
    if (c->stepsBegun != c->stepsEnded) { g_err = "sp_ctx_share_targets: a search step is in flight (sp_ctx_search_end first)"; return SP_ERR_STATE; }
    if (from->targets.n == 0 || !from->tablesUp) { g_err = "sp_ctx_share_targets: the other context has no target set"; return SP_ERR_STATE; }
    if (c->seedName != from->seedName || c->subName != from->subName || c->kmerThr != from->kmerThr || (c->par.mask != 0) != (from->par.mask != 0)) {
        g_err = "sp_ctx_share_targets: the contexts differ in matrices, sensitivity or masking"; return SP_ERR_ARG;
    }
    c->gpu.makeCurrent();
    std::string err;
    if (!c->gpu.shareFrom(from->gpu, err)) { g_err = err; return SP_ERR_DEVICE; }
    c->targets = from->targets; c->targetFirst = from->targetFirst; c->indexEntries = from->indexEntries;
    c->tablesUp = from->tablesUp; c->kmerTablesUp = from->kmerTablesUp;
    c->binSize = c->par.bin_size ? c->par.bin_size : from->binSize;
    for (const char *k : {"ms_mask_gpu", "ms_index_gpu", "index_slots", "index_entries", "masked_residues", "target_residues"}) { auto it = from->stats.find(k); if (it != from->stats.end()) c->stats[k] = it->second; }
    c->stats["ms_ext_table"] = 0; c->stats["ms_index_build"] = 0; c->stats["ms_upload"] = 0; c->stats["bin_size"] = c->binSize;
    c->haveGapped = c->haveGapless = false;
    c->gateRows.clear(); c->evalRows.clear(); c->bitRow.clear(); c->gateValid = false;
    return SP_OK;
}

int sp_ctx_set_queries(sp_ctx *c, const char *residues, const uint64_t *offsets, uint64_t n) {
    if (c && c->stepsBegun != c->stepsEnded) { g_err = "sp_ctx_set_queries: a search step is in flight (sp_ctx_search_end first)"; return SP_ERR_STATE; }
    if (c) c->gpu.makeCurrent();
    if (!c || !residues || !offsets) { g_err = "sp_ctx_set_queries: null argument"; return SP_ERR_ARG; }
    if (offsets[n] >= (1ull << 32)) { g_err = "query batch larger than 4 GiB of residues"; return SP_ERR_ARG; }
    const bool laps = n > 100000 && getenv("SP_MODULE_TIMES") != nullptr; double tLap = laps ? nowMs() : 0;
    auto lap = [&](const char *what) { if (laps) { const double t = nowMs(); fprintf(stderr, "[sp]   set_queries: %-22s %8.3f ms\n", what, t - tLap); tLap = t; } };
    c->qres.resize(offsets[n]);
    c->qoff.resize(n + 1);
    c->gateValid = false;
    c->qascii.assign(residues, residues + offsets[n]);
    c->qasciiOff.assign(offsets, offsets + n + 1);
    for (uint64_t i = 0; i <= n; i++) c->qoff[i] = (uint32_t) offsets[i];
    lap("copies");
    {   // letters -> residue classes, a few threads for batches of millions of residues
        const uint64_t R = offsets[n];
        const size_t T = (size_t) std::max<uint64_t>(1, std::min<uint64_t>((uint64_t) std::min(16, std::max(1, c->threads)), R / (1u << 19) + 1));
        auto map = [&](size_t t) { for (uint64_t i = R * t / T, e = R * (t + 1) / T; i < e; i++) c->qres[i] = c->seed.aa2num[(uint8_t) residues[i]]; };
        if (T == 1) map(0);
        else sp::runTasks(T, map);
    }
    lap("letters -> classes");
    std::string err;
    if (!c->gpu.uploadQueries(c->qres.data(), c->qoff.data(), (size_t) n, err)) { g_err = err; return SP_ERR_DEVICE; }
    lap("to the device");
    return SP_OK;
}

static void prefilterStats(sp_ctx *c);

int sp_ctx_prefilter(sp_ctx *c) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_prefilter: no target set"; return SP_ERR_STATE; }
    if (c->indexEntries == 0) { g_err = "sp_ctx_prefilter: the target set has no k-mer index (no indexable k-mer, or staged without one)"; return SP_ERR_STATE; }
    std::string err;
    // maxResListLen = min(#targets, --max-seqs): prefiltering/Prefiltering.cpp:170
    const int maxHits = sp_ctx_max_seqs(c);
    if (!c->gpu.prefilter(c->kmerThr, c->binSize, maxHits, c->par.min_ungapped_score, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    prefilterStats(c);
    return SP_OK;
}

static void prefilterStats(sp_ctx *c) {
    sp::PrefilterCounters pc = c->gpu.counters();
    sp::KernelTimes kt = c->gpu.times();
    c->stats["kmers"] = (double) pc.kmers; c->stats["db_matches"] = (double) pc.dbMatches; c->stats["diagonals"] = (double) pc.diagonals;
    c->stats["diag_residues"] = (double) pc.diagResidues; c->stats["hits"] = (double) pc.hits;
    c->stats["ms_prep"] = kt.prep; c->stats["ms_match"] = kt.match; c->stats["ms_match_active"] = kt.matchActive; c->stats["ms_score"] = kt.score; c->stats["ms_select"] = kt.select;
}

int sp_ctx_prefilter_candidates(sp_ctx *c) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_prefilter_candidates: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!c->gpu.prefilterCandidates(c->kmerThr, c->par.min_ungapped_score, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    prefilterStats(c);
    c->stats["candidates"] = (double) c->gpu.numCandidates();
    return SP_OK;
}

struct sp_comm { sp::Comm *c; };
// ---- the probe-list cut (Gpu::probe*)
int sp_ctx_shard_occupancy(sp_ctx *c, void *out, int on_device) {
    if (!c || !out) { g_err = "sp_ctx_shard_occupancy: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    if (c->targets.n == 0) { g_err = "sp_ctx_shard_occupancy: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!c->gpu.shardOccupancy(out, on_device != 0, err)) { g_err = err; return SP_ERR_DEVICE; }
    return SP_OK;
}
int sp_ctx_probe_set_shards(sp_ctx *c, const void *maps, int n_shards, int on_device) {
    if (!c || !maps) { g_err = "sp_ctx_probe_set_shards: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    std::string err;
    if (!c->gpu.probeSetShardMaps(maps, n_shards, on_device != 0, err)) { g_err = err; return n_shards < 1 || n_shards > SP_MAX_PROBE_SHARDS ? SP_ERR_ARG : SP_ERR_DEVICE; }
    return SP_OK;
}
int sp_ctx_probe_enumerate(sp_ctx *c, uint64_t q_from, uint64_t q_to, uint64_t *counts) {
    if (!c) { g_err = "sp_ctx_probe_enumerate: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    std::string err;
    if (!c->gpu.probeEnumerate(c->kmerThr, (size_t) q_from, (size_t) q_to, counts, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    { sp::KernelTimes kt = c->gpu.times(); c->stats["ms_prep"] = kt.prep; c->stats["ms_match"] = kt.match; }
    return SP_OK;
}
int sp_ctx_probe_get(sp_ctx *c, int dest, sp_probe *records, uint32_t *query_counts, int on_device) {
    if (!c || !records || !query_counts || dest < 0 || dest >= SP_MAX_PROBE_SHARDS) { g_err = "sp_ctx_probe_get: bad argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    const size_t n = c->gpu.probeCount(dest), nq = c->gpu.probeSliceQueries();
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    if ((n && hipMemcpy(records, c->gpu.probeRecordsDevice(dest), n * sizeof(sp_probe), kind) != hipSuccess) ||
        (nq && hipMemcpy(query_counts, c->gpu.probeQueryCountsDevice(dest), nq * sizeof(uint32_t), kind) != hipSuccess)) { g_err = "sp_ctx_probe_get: copy failed"; return SP_ERR_DEVICE; }
    return SP_OK;
}
const void *sp_ctx_probe_records_device(const sp_ctx *c, int dest) { return c && dest >= 0 && dest < SP_MAX_PROBE_SHARDS ? c->gpu.probeRecordsDevice(dest) : nullptr; }
const void *sp_ctx_probe_query_counts_device(const sp_ctx *c, int dest) { return c && dest >= 0 && dest < SP_MAX_PROBE_SHARDS ? c->gpu.probeQueryCountsDevice(dest) : nullptr; }
int sp_ctx_probe_fetch(sp_ctx *c, const sp_probe *records, uint64_t n, const uint32_t *query_counts, int on_device) {
    if (!c || !query_counts || (n && !records)) { g_err = "sp_ctx_probe_fetch: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    if (c->targets.n == 0) { g_err = "sp_ctx_probe_fetch: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!c->gpu.probeFetch(records, (size_t) n, query_counts, on_device != 0, c->kmerThr, c->par.min_ungapped_score, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    prefilterStats(c);
    c->stats["candidates"] = (double) c->gpu.numCandidates();
    return SP_OK;
}
int sp_ctx_probe_search(sp_ctx *c, sp_comm *comm, uint64_t q_from, uint64_t q_to, double *exchange_s) {
    if (!c || !comm) { g_err = "sp_ctx_probe_search: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    if (c->targets.n == 0) { g_err = "sp_ctx_probe_search: no target set"; return SP_ERR_STATE; }
    std::string err; double ms = 0;
    if (!c->gpu.probeSearch(*comm->c, c->kmerThr, c->par.min_ungapped_score, (size_t) q_from, (size_t) q_to, &ms, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    if (exchange_s) *exchange_s = ms * 1e-3;
    prefilterStats(c);
    c->stats["candidates"] = (double) c->gpu.numCandidates(); c->stats["ms_probe_exchange"] = ms;
    return SP_OK;
}

uint64_t sp_ctx_num_candidates(const sp_ctx *c) { return (uint64_t) c->gpu.numCandidates(); }
uint64_t sp_ctx_overflow_queries(const sp_ctx *c, uint32_t *out, uint64_t cap) {
    const std::vector<uint32_t> &l = c->gpu.overflowQueries();
    for (uint64_t i = 0; i < l.size() && i < cap; i++) out[i] = l[i];
    return (uint64_t) l.size();
}
const void *sp_ctx_candidates_device(const sp_ctx *c) { return c->gpu.candidatesDevice(); }

int sp_ctx_get_candidates(sp_ctx *c, sp_candidate *out) {
    if (c) c->gpu.makeCurrent();
    static_assert(sizeof(sp_candidate) == sizeof(sp::Candidate), "candidate layout");
    std::string err;
    if (!c->gpu.downloadCandidates(reinterpret_cast<sp::Candidate *>(out), err)) { g_err = err; return SP_ERR_DEVICE; }
    return SP_OK;
}

int sp_ctx_merge_candidates(sp_ctx *c, const sp_candidate *cands, uint64_t n, int onDevice, uint64_t qFrom, uint64_t qTo) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_merge_candidates: no target set"; return SP_ERR_STATE; }
    if (n && !cands) { g_err = "sp_ctx_merge_candidates: null candidate list"; return SP_ERR_ARG; }
    std::string err;
    if (!c->gpu.mergeCandidates(reinterpret_cast<const sp::Candidate *>(cands), (size_t) n, onDevice != 0, (size_t) qFrom, (size_t) qTo, c->binSize,
                                sp_ctx_max_seqs(c), c->par.min_ungapped_score, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    sp::KernelTimes kt = c->gpu.times();
    c->stats["hits"] = (double) c->gpu.counters().hits; c->stats["ms_select"] = kt.select;
    return SP_OK;
}

/* ---- multi-GPU: one process per GPU, see include/spacepharer_amd.h ---- */
int sp_dist_unique_id(unsigned char *id) {
    std::string err;
    if (!id || !sp::Comm::uniqueId(id, err)) { g_err = id ? err : "sp_dist_unique_id: null argument"; return SP_ERR_ARG; }
    return SP_OK;
}
sp_comm *sp_dist_init(const unsigned char *id, int rank, int world, int transport) {
    if (!id) { g_err = "sp_dist_init: null id"; return nullptr; }
    std::string err;
    if (transport == SP_TRANSPORT_RCCL) (void) hipSetDevice(sp::currentDevice());      // the RCCL communicator binds to this rank's GPU
    sp::Comm *c = sp::Comm::create(id, rank, world, transport, 0, err);
    if (!c) { g_err = err; return nullptr; }
    return new sp_comm{c};
}
void sp_dist_destroy(sp_comm *comm) { if (comm) { delete comm->c; delete comm; } }
int sp_dist_rank(const sp_comm *comm) { return comm ? comm->c->rank() : 0; }
int sp_dist_world(const sp_comm *comm) { return comm ? comm->c->world() : 1; }
int sp_dist_transport(const sp_comm *comm) { return comm ? comm->c->transport() : SP_TRANSPORT_HOST; }
int sp_dist_barrier(sp_comm *comm) {
    std::string err;
    if (!comm || !comm->c->barrier(err)) { g_err = comm ? err : "sp_dist_barrier: null communicator"; return SP_ERR_DEVICE; }
    return SP_OK;
}
int sp_dist_allgather(sp_comm *comm, const void *send, uint64_t bytes, void *recv, int on_device) {
    std::string err;
    if (!comm || !comm->c->allgather(send, (size_t) bytes, recv, on_device != 0, err)) { g_err = comm ? err : "sp_dist_allgather: null communicator"; return SP_ERR_DEVICE; }
    return SP_OK;
}
int sp_dist_allreduce_f64(sp_comm *comm, double *values, uint64_t n, int op) {
    std::string err;
    if (!comm || !comm->c->allreduce(values, (size_t) n, op, err)) { g_err = comm ? err : "sp_dist_allreduce_f64: null communicator"; return SP_ERR_DEVICE; }
    return SP_OK;
}
int sp_dist_alltoallv(sp_comm *comm, const void *const *send, const uint64_t *send_bytes, void *const *recv, const uint64_t *recv_bytes, int on_device) {
    if (!comm || !send || !send_bytes || !recv || !recv_bytes) { g_err = "sp_dist_alltoallv: null argument"; return SP_ERR_ARG; }
    std::string err;
    const size_t W = (size_t) comm->c->world();
    std::vector<size_t> sb(W), rb(W);
    for (size_t k = 0; k < W; k++) { sb[k] = (size_t) send_bytes[k]; rb[k] = (size_t) recv_bytes[k]; }
    if (!comm->c->alltoallv(send, sb.data(), recv, rb.data(), on_device != 0, err)) { g_err = err; return SP_ERR_DEVICE; }
    return SP_OK;
}
uint64_t sp_dist_bytes_moved(const sp_comm *comm) { return comm ? comm->c->bytesMoved() : 0; }
int sp_dist_decompose(const uint64_t *entry_lengths, uint64_t n, int rank, int world, uint64_t *start, uint64_t *count) {
    size_t s = 0, k = 0;
    if (!entry_lengths || !start || !count || world < 1 || rank < 0 || rank >= world) { g_err = "sp_dist_decompose: bad argument"; return SP_ERR_ARG; }
    if (!sp::decomposeByResidues(entry_lengths, (size_t) n, rank, world, s, k)) { g_err = "sp_dist_decompose: more ranks than residues"; return SP_ERR_ARG; }
    *start = s; *count = k;
    return SP_OK;
}

int sp_ctx_allgather_candidates(sp_ctx *c, sp_comm *comm, uint64_t q_from, uint64_t q_to, uint64_t *union_rows, double *exchange_s) {
    if (!c || !comm) { g_err = "sp_ctx_allgather_candidates: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    if (c->targets.n == 0) { g_err = "sp_ctx_allgather_candidates: no target set"; return SP_ERR_STATE; }
    std::string err; size_t rows = 0;
    const double t0 = nowMs();
    if (!c->gpu.allgatherCandidates(*comm->c, rows, err)) { g_err = err; return SP_ERR_DEVICE; }
    const double t1 = nowMs();
    if (union_rows) *union_rows = rows;
    if (exchange_s) *exchange_s = (t1 - t0) * 1e-3;
    if (!c->gpu.mergeCandidates(c->gpu.candidateUnionDevice(), rows, true, (size_t) q_from, (size_t) q_to, c->binSize, sp_ctx_max_seqs(c), c->par.min_ungapped_score, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    sp::KernelTimes kt = c->gpu.times();
    c->stats["hits"] = (double) c->gpu.counters().hits; c->stats["ms_select"] = kt.select; c->stats["ms_exchange"] = t1 - t0;
    return SP_OK;
}

int sp_ctx_allgather_hits(sp_ctx *c, sp_comm *comm, uint64_t query_offset, uint64_t nq_total, const sp_hit_packed *my_hits, const uint32_t *my_offsets,
                          sp_hit_packed *hits, uint64_t *offsets, uint64_t hit_cap, uint64_t *n_hits) {
    if (!c || !comm || !hits || !offsets || !n_hits || ((my_hits == nullptr) != (my_offsets == nullptr))) { g_err = "sp_ctx_allgather_hits: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    std::string err; size_t n = 0;
    if (!c->gpu.allgatherHits(*comm->c, (size_t) query_offset, (size_t) nq_total, reinterpret_cast<const sp::GpuHit *>(my_hits), my_offsets, reinterpret_cast<sp::GpuHit *>(hits), offsets,
                              (size_t) hit_cap, n, err)) { g_err = err; return SP_ERR_DEVICE; }
    *n_hits = n;
    return SP_OK;
}

uint64_t sp_ctx_num_hits(const sp_ctx *c) { return (uint64_t) c->gpu.counters().hits; }

int sp_ctx_get_hits_compact(sp_ctx *c, sp_hit *hits, uint64_t *offsets) {
    if (c) c->gpu.makeCurrent();
    const sp::GpuHit *h = nullptr; const uint32_t *off = nullptr; size_t n = 0; std::string err;
    if (!c->gpu.hostHits(h, off, n, err)) { g_err = err; return SP_ERR_DEVICE; }
    const size_t nq = c->gpu.numQueries();
    for (size_t q = 0; q <= nq; q++) offsets[q] = off[q];
    for (size_t i = 0; i < n; i++) { hits[i].seq_id = h[i].seqId; hits[i].score = h[i].score; hits[i].diagonal = (int16_t) h[i].diag; hits[i].pad = 0; }
    return SP_OK;
}

int sp_ctx_hits_view(sp_ctx *c, const sp_hit_packed **hits, const uint32_t **offsets, uint64_t *n) {
    if (c) c->gpu.makeCurrent();
    static_assert(sizeof(sp_hit_packed) == sizeof(sp::GpuHit), "sp_hit_packed is the device hit record");
    const sp::GpuHit *h = nullptr; const uint32_t *off = nullptr; size_t cnt = 0; std::string err;
    if (!c->gpu.hostHits(h, off, cnt, err)) { g_err = err; return SP_ERR_DEVICE; }
    *hits = reinterpret_cast<const sp_hit_packed *>(h); *offsets = off; *n = cnt;
    return SP_OK;
}

int sp_ctx_set_hit_sink(sp_ctx *c, sp_hit_packed *hits, uint32_t *offsets, uint64_t cap_hits) {
    if (!c) { g_err = "sp_ctx_set_hit_sink: null context"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    c->gpu.setHitSink(reinterpret_cast<sp::GpuHit *>(hits), offsets, (size_t) cap_hits);
    return SP_OK;
}

int sp_ctx_get_hits(sp_ctx *c, sp_hit *hits, uint32_t *counts) {
    if (c) c->gpu.makeCurrent();
    const sp::GpuHit *h = nullptr; const uint32_t *off = nullptr; size_t n = 0; std::string err;
    if (!c->gpu.hostHits(h, off, n, err)) { g_err = err; return SP_ERR_DEVICE; }
    const size_t nq = c->gpu.numQueries(); const int mh = c->gpu.maxHits();
    for (size_t q = 0; q < nq; q++) {
        counts[q] = (uint32_t) (off[q + 1] - off[q]);
        for (uint64_t i = off[q]; i < off[q + 1]; i++) {
            sp_hit &o = hits[q * mh + (i - off[q])];
            o.seq_id = h[i].seqId; o.score = h[i].score; o.diagonal = (int16_t) h[i].diag; o.pad = 0;
        }
    }
    return SP_OK;
}

static bool ensureEvaluers(sp_ctx *c, bool gapped, std::string &err) {
    if (gapped && !c->haveGapped) { if (!c->evGapped.init(c->aln.name, c->par.gap_open, c->par.gap_extend, c->targets.residues(), err, sp::matrixDigest(c->aln))) return false; c->haveGapped = true; }
    if (!gapped && !c->haveGapless) { if (!c->evGapless.init(c->aln.name, -1, -1, c->targets.residues(), err, sp::matrixDigest(c->aln))) return false; c->haveGapless = true; }
    return true;
}

// E-value gate table for the query lengths of the resident batch: pass[slot(qLen)][score] = !(evalue(score, qLen) > -e)
// evaluated with the exact double arithmetic of EvalueComputation::computeEvalue; rows are cached per length
static void buildGate(sp_ctx *c) {
    if (c->gateValid) return;
    sp::GateTable &g = c->gate;
    const uint32_t maxL = c->gpu.maxQueryLen();
    g.stride = std::min<uint32_t>(65536u, maxL * 32u + 2u);
    g.slotOf.assign(maxL + 2, 0);
    std::vector<uint32_t> lens;
    { std::vector<uint8_t> seen(maxL + 2, 0); for (size_t q = 0; q + 1 < c->qoff.size(); q++) { uint32_t L = c->qoff[q + 1] - c->qoff[q]; if (!seen[L]) { seen[L] = 1; lens.push_back(L); } } }
    g.pass.assign((lens.size() + 1) * (size_t) g.stride, 0);
    g.evalue.assign((lens.size() + 1) * (size_t) g.stride, 0.0);
    for (size_t s = 0; s < lens.size(); s++) {
        const uint32_t L = lens[s];
        g.slotOf[L] = (uint16_t) s;
        std::vector<uint8_t> &row = c->gateRows[L];
        if (row.size() != g.stride) {
            row.assign(g.stride, 0);
            std::vector<double> &ev = c->evalRows[L];
            const uint32_t lim = std::min<uint32_t>(g.stride, L * 32u + 2u);       // scores above qLen * max matrix entry cannot occur
            ev.assign(lim, 0.0);
            for (uint32_t sc = 0; sc < lim; sc++) { ev[sc] = c->evGapped.evalue((double) sc, (double) L); row[sc] = !(ev[sc] > c->par.evalue); }
            for (uint32_t sc = lim; sc < g.stride; sc++) row[sc] = lim ? row[lim - 1] : 0;
        }
        memcpy(&g.pass[s * (size_t) g.stride], row.data(), g.stride);
        const std::vector<double> &ev = c->evalRows[L];
        for (uint32_t sc = 0; sc < g.stride; sc++) g.evalue[s * (size_t) g.stride + sc] = sc < ev.size() ? ev[sc] : c->evGapped.evalue((double) sc, (double) L);
    }
    if (c->bitRow.size() != g.stride) { c->bitRow.resize(g.stride); for (uint32_t sc = 0; sc < g.stride; sc++) c->bitRow[sc] = (int) (c->evGapped.bitScore((double) sc) + 0.5); }
    g.bits.assign(c->bitRow.begin(), c->bitRow.end());
    // "%.3E" of every table entry for the device-side result text (sp_gpu_text.inc): 16 bytes each; left empty -- the host formats then -- when the table is large
    // (long queries: stride = 32 x the longest query) or an entry does not fit (inf / nan / three-digit exponents)
    g.evalText.clear();
    if (g.evalue.size() <= ((size_t) 1 << 20)) {
        g.evalText.assign(g.evalue.size() * 16, 0);
        for (size_t i = 0; i < g.evalue.size(); i++) {
            char tmp[48]; const size_t n = (size_t) (sp::fmtE3(g.evalue[i], tmp) - tmp);
            if (n == 0 || n > 15) { g.evalText[i * 16] = 255; continue; }      // (the kernel reports such a record and the caller falls back)
            g.evalText[i * 16] = (unsigned char) n; memcpy(&g.evalText[i * 16 + 1], tmp, n);
        }
    }
    g.evalueThr = c->par.evalue;
    g.version = ++c->gateSerial;
    c->gateValid = true;
}

static int finishAlign(sp_ctx *c, const sp::AlignHost &ah, const std::vector<uint32_t> *order,
                       sp_alignment *res, char *cigars, uint64_t cigarCap, uint64_t *cigarUsed) {
    const sp::PinVec<sp::SwPair> &pairs = ah.pairs; const sp::PinVec<sp::SwForward> &fwd = ah.fwd; const std::vector<uint8_t> &pass = ah.pass;
    const sp::PinVec<sp::SwFull> &full = ah.full; const sp::PinVec<char> &bt = ah.bt;
    const size_t n = pairs.size();
    uint64_t used = 0;
    for (size_t i = 0; i < n; i++) if (pass[i]) used += (uint64_t) full[i].alnLen + 1;
    if (cigarUsed) *cigarUsed = used;
    if (used > cigarCap) { g_err = "sp_ctx_align: cigar buffer too small"; return SP_ERR_OVERFLOW; }
    std::vector<const std::vector<double> *> evRow(c->gpu.maxQueryLen() + 2, nullptr);
    for (auto &kv : c->evalRows) if (kv.first < evRow.size()) evRow[kv.first] = &kv.second;
    // cigar offsets first (serial, cheap), then the records in parallel
    std::vector<uint64_t> cigAt(n + 1, 0);
    for (size_t o = 0; o < n; o++) { const size_t i = order ? (*order)[o] : o; cigAt[o + 1] = cigAt[o] + (pass[i] ? (uint64_t) full[i].alnLen + 1 : 0); }
    auto fill = [&](size_t from, size_t to) {
        for (size_t o = from; o < to; o++) {
            const size_t i = order ? (*order)[o] : o;
            sp_alignment &r = res[o];
            memset(&r, 0, sizeof(r));
            const uint32_t q = pairs[i].query, t = pairs[i].target;
            const uint32_t qL = c->qoff[q + 1] - c->qoff[q];
            r.db_key = c->targets.keys[t];
            r.raw_score = fwd[i].score; r.q_len = qL; r.db_len = (uint32_t) (c->targets.offsets[t + 1] - c->targets.offsets[t]);
            r.q_start = -1; r.db_start = -1; r.q_end = fwd[i].readEnd; r.db_end = fwd[i].refEnd;
            const uint32_t sc = (uint32_t) fwd[i].score;
            r.bit_score = sc < c->bitRow.size() ? c->bitRow[sc] : (int) (c->evGapped.bitScore(fwd[i].score) + 0.5);
            if (fwd[i].refEnd == -1) continue;
            const std::vector<double> *er = qL < evRow.size() ? evRow[qL] : nullptr;
            r.evalue = (er && sc < er->size()) ? (*er)[sc] : c->evGapped.evalue(fwd[i].score, qL);
            if (!pass[i]) continue;
            r.q_start = full[i].qStart; r.db_start = full[i].dbStart; r.aln_len = (uint32_t) full[i].alnLen;
            r.seq_id = (float) full[i].identities / (float) full[i].alnLen;     // Util::computeSeqId, SEQ_ID_ALN_LEN
            r.cigar_offset = (uint32_t) cigAt[o];
            memcpy(cigars + cigAt[o], bt.data() + full[i].btOffset, (size_t) full[i].alnLen);
            cigars[cigAt[o] + full[i].alnLen] = 0;
            r.passed = (r.evalue <= c->par.evalue) && (r.seq_id >= 0.0);     // Alignment::checkCriteria, -c 0 --min-seq-id 0 --min-aln-len 0
        }
    };
    const size_t nThreads = n > 20000 ? (size_t) std::min(8, std::max(1, c->threads)) : 1;
    if (nThreads == 1) fill(0, n);
    else {
        std::vector<std::thread> pool; const size_t chunk = (n + nThreads - 1) / nThreads;
        for (size_t t = 0; t < nThreads; t++) { const size_t a = std::min(n, t * chunk), b = std::min(n, a + chunk); pool.emplace_back(fill, a, b); }
        for (auto &th : pool) th.join();
    }
    sp::KernelTimes kt = c->gpu.times();
    c->stats["ms_sw_forward"] = kt.swForward; c->stats["ms_sw_reverse"] = kt.swReverse; c->stats["ms_traceback"] = kt.traceback;
    return SP_OK;
}

// records assembled on the device went straight into the caller's buffers; otherwise (a pair needed the retry path)
// assemble on the host from the downloaded pieces
static int finishDirectOrHost(sp_ctx *c, const sp::AlignDirect &direct, sp_alignment *res, char *cigars, uint64_t cigarCap, uint64_t *cigarUsed) {
    sp::KernelTimes kt = c->gpu.times();
    c->stats["ms_sw_forward"] = kt.swForward; c->stats["ms_sw_reverse"] = kt.swReverse; c->stats["ms_traceback"] = kt.traceback; c->stats["ms_assemble"] = kt.assemble;
    c->stats["align_retries"] = (double) c->gpu.alignRetries();
    if (direct.overflow) { if (cigarUsed) *cigarUsed = direct.cigarUsed; g_err = "sp_ctx_align: cigar buffer too small"; return SP_ERR_OVERFLOW; }
    if (direct.done) { if (cigarUsed) *cigarUsed = direct.cigarUsed; return SP_OK; }
    return finishAlign(c, c->alnHost, nullptr, res, cigars, cigarCap, cigarUsed);
}

int sp_ctx_align(sp_ctx *c, const uint32_t *qidx, const uint32_t *tidx, uint64_t n, sp_alignment *res, char *cigars, uint64_t cigarCap, uint64_t *cigarUsed) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_align: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!ensureEvaluers(c, true, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    std::vector<sp::SwPair> pairs(n);
    for (uint64_t i = 0; i < n; i++) { pairs[i].query = qidx[i]; pairs[i].target = tidx[i]; }
    buildGate(c);
    static_assert(sizeof(sp::AlnRec) == sizeof(sp_alignment) && offsetof(sp::AlnRec, evalue) == offsetof(sp_alignment, evalue) &&
                  offsetof(sp::AlnRec, cigarOffset) == offsetof(sp_alignment, cigar_offset) && offsetof(sp::AlnRec, passed) == offsetof(sp_alignment, passed), "AlnRec mirrors sp_alignment");
    sp::AlignDirect direct; direct.records = res; direct.cigars = cigars; direct.cigarCap = cigarCap;
    if (!c->gpu.align(pairs.data(), (size_t) n, c->par.gap_open, c->par.gap_extend, c->gate, c->alnHost, &direct, err)) { g_err = err; return SP_ERR_DEVICE; }
    return finishDirectOrHost(c, direct, res, cigars, cigarCap, cigarUsed);
}

/* align every hit of the last sp_ctx_prefilter(): results[i] belongs to hit i of sp_ctx_get_hits_compact() */
int sp_ctx_align_hits(sp_ctx *c, sp_alignment *res, char *cigars, uint64_t cigarCap, uint64_t *cigarUsed) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_align_hits: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!ensureEvaluers(c, true, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    const bool dbg = getenv("SP_DEBUG") != nullptr;
    double t0 = nowMs();
    buildGate(c);
    double t1 = nowMs();
    sp::AlignDirect direct; direct.records = res; direct.cigars = cigars; direct.cigarCap = cigarCap;
    if (!c->gpu.align(nullptr, 0, c->par.gap_open, c->par.gap_extend, c->gate, c->alnHost, &direct, err)) { g_err = err; return SP_ERR_DEVICE; }
    double t2 = nowMs();
    int rc = finishDirectOrHost(c, direct, res, cigars, cigarCap, cigarUsed);
    if (dbg) fprintf(stderr, "[sp] align_hits: gate %.3f  gpu.align %.3f  finish %.3f ms\n", t1 - t0, t2 - t1, nowMs() - t2);
    return rc;
}

/* result-DB text formatted on the device: see include/spacepharer_amd.h */
int sp_ctx_hits_text(sp_ctx *c, char *text, uint64_t textCap, uint64_t *textUsed, uint32_t *entryLen) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0 || !textUsed || !entryLen) { g_err = "sp_ctx_hits_text: no target set / missing output"; return SP_ERR_STATE; }
    std::string err; uint64_t used = 0; bool declined = false;
    if (!c->gpu.hitsText(text, text ? textCap : 0, used, entryLen, declined, err)) { g_err = "sp_ctx_hits_text: " + err; return SP_ERR_DEVICE; }
    if (declined) { g_err = "sp_ctx_hits_text: the entries of this batch add up to 4 GiB of text or more (32-bit offsets on the device): format them on the host (sp_ctx_get_hits) or search smaller batches"; return SP_ERR_UNSUPPORTED; }
    *textUsed = used;
    return used > (text ? textCap : 0) ? SP_ERR_OVERFLOW : SP_OK;
}
int sp_ctx_text_index(sp_ctx *c, const uint32_t *keys, uint64_t n, uint64_t base, char *text, uint64_t textCap, uint64_t *textUsed) {
    if (c) c->gpu.makeCurrent();
    if (!c || !keys || !textUsed) { g_err = "sp_ctx_text_index: null argument"; return SP_ERR_ARG; }
    std::string err; uint64_t used = 0;
    if (!c->gpu.textIndex(keys, (size_t) n, base, text, text ? textCap : 0, used, err)) { g_err = "sp_ctx_text_index: " + err; return SP_ERR_STATE; }
    *textUsed = used;
    return used > (text ? textCap : 0) ? SP_ERR_OVERFLOW : SP_OK;
}
int sp_ctx_align_hits_text(sp_ctx *c, double minSeqId, uint32_t minAlnLen, char *text, uint64_t textCap, uint64_t *textUsed, uint32_t *entryLen) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0 || !textUsed || !entryLen) { g_err = "sp_ctx_align_hits_text: no target set / missing output"; return SP_ERR_STATE; }
    std::string err;
    if (!ensureEvaluers(c, true, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    buildGate(c);
    uint64_t used = 0; bool fellBack = false;
    if (!c->gpu.alignHitsText(c->par.gap_open, c->par.gap_extend, c->gate, c->alnHost, minSeqId, minAlnLen, text, text ? textCap : 0, used, entryLen, fellBack, err)) { g_err = "sp_ctx_align_hits_text: " + err; return SP_ERR_DEVICE; }
    if (fellBack) { g_err = "sp_ctx_align_hits_text: this batch needs the host's formatting (a pair took the host-driven traceback, an E-value has no table text, or the entries add up to 4 GiB of text): use sp_ctx_align_hits"; return SP_ERR_UNSUPPORTED; }
    *textUsed = used;
    return used > (text ? textCap : 0) ? SP_ERR_OVERFLOW : SP_OK;
}

/* split form of sp_ctx_align_hits(): _begin returns when the device work is done and the results are on their way to
   `res` / `cigars` (leave both alone), _end waits for them.  Between the two the caller may queue the next prefilter pass. */
int sp_ctx_align_hits_begin(sp_ctx *c, sp_alignment *res, char *cigars, uint64_t cigarCap, uint64_t *cigarUsed) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_align_hits_begin: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!ensureEvaluers(c, true, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    buildGate(c);
    sp::AlignDirect direct; direct.records = res; direct.cigars = cigars; direct.cigarCap = cigarCap; direct.async = true;
    if (!c->gpu.align(nullptr, 0, c->par.gap_open, c->par.gap_extend, c->gate, c->alnHost, &direct, err)) { g_err = err; return SP_ERR_DEVICE; }
    return finishDirectOrHost(c, direct, res, cigars, cigarCap, cigarUsed);
}
int sp_ctx_align_hits_end(sp_ctx *c) {
    if (c) c->gpu.makeCurrent();
    std::string err;
    if (!c->gpu.waitAlignCopy(err)) { g_err = err; return SP_ERR_DEVICE; }
    return SP_OK;
}

/* One whole search step (sp_ctx_prefilter + sp_ctx_align_hits) queued without a host round trip; see include/spacepharer_amd.h */
int sp_ctx_search_begin(sp_ctx *c, sp_hit_packed *hits, uint32_t *offsets, uint64_t hit_cap, sp_alignment *results, char *cigars, uint64_t cigar_cap) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_search_begin: no target set"; return SP_ERR_STATE; }
    if (c->indexEntries == 0) { g_err = "sp_ctx_search_begin: the target set has no k-mer index"; return SP_ERR_STATE; }
    if (!hits || !offsets || !results || !cigars) { g_err = "sp_ctx_search_begin: result buffers missing"; return SP_ERR_ARG; }
    std::string err;
    if (!ensureEvaluers(c, true, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    buildGate(c);
    sp::StepOut o; o.hits = reinterpret_cast<sp::GpuHit *>(hits); o.offsets = offsets; o.hitCap = hit_cap; o.records = reinterpret_cast<sp::AlnRec *>(results);
    o.cigars = cigars; o.recordCap = hit_cap; o.cigarCap = cigar_cap;
    if (!c->gpu.stepBegin(c->kmerThr, c->binSize, sp_ctx_max_seqs(c), c->par.min_ungapped_score, c->par.gap_open, c->par.gap_extend, c->gate, o, err)) { g_err = err; return SP_ERR_STATE; }
    c->stepOut[c->stepsBegun++ & 1] = o;
    return SP_OK;
}

int sp_ctx_search_end(sp_ctx *c, uint64_t *n_hits, uint64_t *cigar_used) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->stepsBegun == c->stepsEnded) { g_err = "sp_ctx_search_end: no step in flight"; return SP_ERR_STATE; }
    const sp::StepOut o = c->stepOut[c->stepsEnded++ & 1];
    sp::StepResult r; std::string err;
    if (!c->gpu.stepEnd(r, err)) { g_err = err; return SP_ERR_OVERFLOW; }
    c->stats["steps_repeated"] += r.repeat ? 1.0 : 0.0;
    if (r.repeat) {
        // rare: the step met a capacity.  Let everything queued finish (a newer step keeps its results), then take this step the synchronous way
        // into the same buffers; the context remembers the larger sizes.
        if (getenv("SP_DEBUG")) fprintf(stderr, "[sp] search step repeated: %s\n", r.why.c_str());
        if (!c->gpu.drain(err)) { g_err = err; return SP_ERR_DEVICE; }
        c->gpu.setHitSink(o.hits, o.offsets, (size_t) o.hitCap);
        if (!c->gpu.prefilter(c->kmerThr, c->binSize, sp_ctx_max_seqs(c), c->par.min_ungapped_score, err)) { c->gpu.setHitSink(nullptr, nullptr, 0); g_err = err; return SP_ERR_OVERFLOW; }
        const uint64_t n = c->gpu.counters().hits;
        if (n > o.hitCap) { c->gpu.setHitSink(nullptr, nullptr, 0); g_err = "sp_ctx_search_end: the pass delivers " + std::to_string(n) + " hits, the result buffers hold " + std::to_string(o.hitCap); return SP_ERR_OVERFLOW; }
        sp::AlignDirect direct; direct.records = o.records; direct.cigars = o.cigars; direct.cigarCap = o.cigarCap;
        uint64_t used = 0;
        bool ok = c->gpu.align(nullptr, 0, c->par.gap_open, c->par.gap_extend, c->gate, c->alnHost, &direct, err);
        int rc = ok ? finishDirectOrHost(c, direct, reinterpret_cast<sp_alignment *>(o.records), o.cigars, o.cigarCap, &used) : SP_ERR_DEVICE;
        const sp::GpuHit *h = nullptr; const uint32_t *off = nullptr; size_t cnt = 0;
        if (ok && rc == SP_OK && !c->gpu.hostHits(h, off, cnt, err)) { ok = false; rc = SP_ERR_DEVICE; }
        c->gpu.setHitSink(nullptr, nullptr, 0);
        if (!ok) g_err = err;
        if (rc != SP_OK) return rc;
        c->gpu.stepCapacityHint((size_t) (n + n / 8 + 4096));
        r.hits = n; r.cigarUsed = used;
    }
    prefilterStats(c);
    sp::KernelTimes kt = c->gpu.times();
    c->stats["ms_sw_forward"] = kt.swForward; c->stats["ms_sw_reverse"] = kt.swReverse; c->stats["ms_traceback"] = kt.traceback; c->stats["ms_assemble"] = kt.assemble;
    if (n_hits) *n_hits = r.hits;
    if (cigar_used) *cigar_used = r.cigarUsed;
    return SP_OK;
}

int sp_ctx_rescore(sp_ctx *c, const uint32_t *qidx, const uint32_t *tidx, const int16_t *diag, uint64_t n, int mode, sp_rescore *res) {
    if (c) c->gpu.makeCurrent();
    if (!c || c->targets.n == 0) { g_err = "sp_ctx_rescore: no target set"; return SP_ERR_STATE; }
    std::string err;
    if (!ensureEvaluers(c, false, err)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    std::vector<sp::SwPair> pairs(n); std::vector<uint16_t> d(n);
    for (uint64_t i = 0; i < n; i++) { pairs[i].query = qidx[i]; pairs[i].target = tidx[i]; d[i] = (uint16_t) diag[i]; }
    std::vector<sp::DiagRescore> out;
    if (!c->gpu.rescoreDiagonals(pairs.data(), d.data(), (size_t) n, mode, c->qascii.data(), out, err)) { g_err = err; return SP_ERR_DEVICE; }
    for (uint64_t i = 0; i < n; i++) {
        const uint32_t qL = c->qoff[qidx[i] + 1] - c->qoff[qidx[i]];
        res[i].score = out[i].score; res[i].start_pos = out[i].startPos; res[i].end_pos = out[i].endPos; res[i].diagonal_len = out[i].diagonalLen;
        res[i].diagonal = out[i].diagonal; res[i].dist_to_diagonal = out[i].distToDiagonal;
        res[i].evalue = c->evGapless.evalue(out[i].score, qL);
        res[i].bit_score = (int) (c->evGapless.bitScore(out[i].score) + 0.5);
    }
    return SP_OK;
}

/* the chain after the search on the records in host memory (rows f1 + f2): see include/spacepharer_amd.h */
int sp_ctx_set_scores(sp_ctx *c, const sp_set_layout *l, const sp_alignment *aln, const char *cigars, int stage, sp_text_db *out) {
    if (!c || !l || !out || !l->q_set || !l->t_set) { g_err = "sp_ctx_set_scores: null argument"; return SP_ERR_ARG; }
    c->gpu.makeCurrent();
    memset(out, 0, sizeof(*out));
    const sp::GpuHit *h = nullptr; const uint32_t *off = nullptr; size_t n = 0; std::string err;
    const bool laps = getenv("SP_MODULE_TIMES") != nullptr; double tLap = nowMs();
    auto lap = [&](const char *what) { if (laps) { const double t = nowMs(); fprintf(stderr, "[sp] set_scores: %-24s %8.3f ms\n", what, t - tLap); tLap = t; } };
    if (!c->gpu.hostHits(h, off, n, err)) { g_err = err; return SP_ERR_DEVICE; }
    lap("hit list on the host");
    if (n && (!aln || !cigars)) { g_err = "sp_ctx_set_scores: no alignment records"; return SP_ERR_ARG; }
    if (stage != 1 && (!l->q_nucl || !l->q_nucl_off || !l->t_nucl || !l->t_nucl_off)) { g_err = "sp_ctx_set_scores: the nucleotide ORFs are missing"; return SP_ERR_ARG; }
    sp::sets::SetInputs in;
    in.nq = c->gpu.numQueries(); in.nt = c->targets.n;
    in.aln = aln; in.cigars = cigars; in.hitOffsets = off;
    std::vector<uint32_t> hitTarget(n); for (size_t i = 0; i < n; i++) hitTarget[i] = h[i].seqId;
    in.hitTarget = hitTarget.data();
    in.qNucl = l->q_nucl; in.qNuclOff = l->q_nucl_off; in.tNucl = l->t_nucl; in.tNuclOff = l->t_nucl_off;
    in.qSet = l->q_set; in.tSet = l->t_set; in.tKeys = c->targets.keys.data();
    std::vector<char> qFirst(in.nq); for (size_t q = 0; q < in.nq; q++) qFirst[q] = c->qasciiOff[q + 1] > c->qasciiOff[q] ? c->qascii[c->qasciiOff[q]] : '\0';
    in.qFirst = qFirst.data(); in.tFirst = c->targetFirst.data();
    if (l->nucl_sub_mat) in.nuclMatrix = l->nucl_sub_mat;
    if (l->nucl_gap_open > 0) in.gapOpen = l->nucl_gap_open;
    if (l->nucl_gap_extend > 0) in.gapExtend = l->nucl_gap_extend;
    in.simpleBestHit = l->simple_best_hit != 0; in.threads = c->threads;
    std::vector<uint32_t> keys; std::vector<std::string> entries;
    lap("inputs");
    if (!sp::sets::setScores(in, stage, keys, entries, err)) { g_err = err; return SP_ERR_ARG; }
    lap("setScores");
    uint64_t total = 0; for (const std::string &e : entries) total += e.size() + 1;
    out->n = keys.size();
    out->keys = (uint32_t *) malloc(sizeof(uint32_t) * std::max<size_t>(1, keys.size()));
    out->offsets = (uint64_t *) malloc(sizeof(uint64_t) * (keys.size() + 1));
    out->text = (char *) malloc(std::max<uint64_t>(1, total));
    if (!out->keys || !out->offsets || !out->text) { sp_text_db_free(out); g_err = "sp_ctx_set_scores: out of memory"; return SP_ERR_ARG; }
    uint64_t at = 0;
    for (size_t i = 0; i < keys.size(); i++) { out->keys[i] = keys[i]; out->offsets[i] = at; memcpy(out->text + at, entries[i].c_str(), entries[i].size() + 1); at += entries[i].size() + 1; }
    out->offsets[keys.size()] = at;
    return SP_OK;
}
/* a sequence DB read and flattened by the library's own (threaded, memory-mapped) DB reader: see include/spacepharer_amd.h */
struct sp_flat_db { std::vector<char> res; std::vector<uint64_t> off; std::vector<uint32_t> keys, sets; };
sp_flat_db *sp_flat_db_open(const char *path) {
    if (!path) { g_err = "sp_flat_db_open: null path"; return nullptr; }
    sp_flat_db *f = new sp_flat_db; std::string err;
    if (!sp::loadFlatForAbi(path, f->res, f->off, f->keys, f->sets, err)) { g_err = "sp_flat_db_open: " + err; delete f; return nullptr; }
    return f;
}
void sp_flat_db_close(sp_flat_db *f) { delete f; }
uint64_t sp_flat_db_entries(const sp_flat_db *f) { return f ? f->keys.size() : 0; }
const char *sp_flat_db_residues(const sp_flat_db *f) { return f ? f->res.data() : nullptr; }
const uint64_t *sp_flat_db_offsets(const sp_flat_db *f) { return f ? f->off.data() : nullptr; }
const uint32_t *sp_flat_db_keys(const sp_flat_db *f) { return f ? f->keys.data() : nullptr; }
const uint32_t *sp_flat_db_sets(const sp_flat_db *f) { return f && !f->sets.empty() ? f->sets.data() : nullptr; }
void sp_text_db_free(sp_text_db *db) { if (!db) return; free(db->keys); free(db->offsets); free(db->text); memset(db, 0, sizeof(*db)); }

double sp_ctx_stat(const sp_ctx *c, const char *name) {
    auto it = c->stats.find(name);
    return it == c->stats.end() ? NAN : it->second;
}

void *sp_ctx_stream(const sp_ctx *c) { return c->gpu.stream(); }
/* test hooks for the device-built target store: FNV-1a over the (k-mer, seqId, pos) triples of the resident index in table
   order (the digest tests/golden holds for the reference's IndexTable) and the masked sequences as letters */
int sp_ctx_index_digest(sp_ctx *c, uint64_t *entries, uint64_t *digest) {
    std::vector<uint32_t> km; std::vector<uint64_t> en; std::string err;
    if (!c->gpu.downloadIndex(km, en, err)) { g_err = err; return SP_ERR_DEVICE; }
    uint64_t h = 0;
    for (size_t i = 0; i < km.size(); i++) { const uint32_t rec[3] = {km[i], (uint32_t) en[i], (uint32_t) (en[i] >> 32)}; h = fnv1a(rec, sizeof(rec), h); }
    if (km.empty()) h = 1469598103934665603ULL;
    *entries = km.size(); *digest = h;
    return SP_OK;
}
int sp_ctx_ext3_digest(sp_ctx *c, uint64_t *digest) {
    std::vector<uint32_t> packed; std::string err;
    if (!c->gpu.downloadExt3(packed, err)) { g_err = err; return SP_ERR_DEVICE; }
    *digest = ext3Digest(packed, nullptr, nullptr, 0);
    return SP_OK;
}
int sp_ctx_get_masked(sp_ctx *c, char *out) {
    std::vector<uint8_t> m; std::string err;
    if (!c->gpu.downloadMasked(m, err)) { g_err = err; return SP_ERR_DEVICE; }
    for (size_t i = 0; i < m.size(); i++) out[i] = c->seed.num2aa[m[i]];
    return SP_OK;
}

void *sp_host_alloc(uint64_t bytes) { return sp::pinnedAlloc((size_t) bytes); }
void sp_host_free(void *p) { sp::pinnedFree(p); }
int sp_ctx_sync(sp_ctx *c) { c->gpu.makeCurrent(); std::string err; if (!c->gpu.sync(err)) { g_err = "sp_ctx_sync: " + err; return SP_ERR_DEVICE; } return SP_OK; }
int sp_ctx_set_timing(sp_ctx *c, int level) { c->gpu.setTiming(level); return SP_OK; }
int sp_ctx_set_tuning(sp_ctx *c, const char *name, int value) {
    if (c && name && !strcmp(name, "match_blocks_per_cu")) { c->gpu.setMatchBlocksPerCU(value); return SP_OK; }
    if (c && name && c->gpu.setKnob(name, value)) return SP_OK;
    g_err = std::string("sp_ctx_set_tuning: unknown knob '") + (name ? name : "") + "'"; return SP_ERR_ARG;
}
int sp_ctx_diag_stream_read(sp_ctx *c, uint64_t bytes, int laneBytes, double *ms) {
    std::string err; double t = 0;
    if (!c->gpu.diagStreamRead(bytes, laneBytes, t, err)) { g_err = err; return SP_ERR_DEVICE; }
    if (ms) *ms = t;
    return SP_OK;
}

/* ---- host-side diagnostics (no device needed): used by the CPU test-suite to check the set-up code ---- */

int sp_host_matrix(const char *spec, double bitFactor, double scoreBias, int16_t *sub441, double *pBack21) {
    sp::SubMat m; std::string err;
    if (!m.load(spec, bitFactor, scoreBias, err)) { g_err = err; return SP_ERR_ARG; }
    for (int i = 0; i < 21; i++) { for (int j = 0; j < 21; j++) sub441[i * 21 + j] = m.sub[i][j]; pBack21[i] = m.pBack[i]; }
    return SP_OK;
}

int sp_host_ext3_digest(const char *seedSpec, uint64_t *digest, int16_t *scoreHead64, uint32_t *indexHead64, uint32_t row) {
    sp::SubMat m; std::string err;
    if (!m.load(seedSpec, (double) 8.0f, (double) -0.2f, err)) { g_err = err; return SP_ERR_ARG; }
    std::vector<uint32_t> packed; sp::buildExt3(m, packed, sp::defaultThreads());
    *digest = ext3Digest(packed, scoreHead64, indexHead64, row);
    return SP_OK;
}

int sp_host_index_digest(const char *seedSpec, float sensitivity, int mask, const char *residues, const uint64_t *offsets, uint64_t n,
                         uint64_t *entries, uint64_t *maskedResidues, uint64_t *digest, char *maskedOut) {
    sp::SubMat m; std::string err;
    if (!m.load(seedSpec, (double) 8.0f, (double) -0.2f, err)) { g_err = err; return SP_ERR_ARG; }
    const int threads = sp::defaultThreads();
    sp::TargetSet ts; sp::KmerIndex idx;
    sp::prepareTargets(m, residues, offsets, nullptr, (size_t) n, mask != 0, threads, ts);
    sp::buildIndex(m, ts, sp::kmerThreshold(sensitivity, INT_MAX, sp::kKmer), threads, idx);
    uint64_t h = 0;
    for (size_t k = 0; k < sp::kTableSize; k++) for (uint32_t e = idx.offsets[k]; e < idx.offsets[k + 1]; e++) {
        uint32_t rec[3] = {(uint32_t) k, (uint32_t) idx.entries[e], (uint32_t) (idx.entries[e] >> 32)};
        h = fnv1a(rec, sizeof(rec), h);
    }
    *entries = idx.entries.size(); *maskedResidues = ts.maskedCount; *digest = h;
    if (maskedOut) for (uint64_t i = 0; i < ts.residues(); i++) maskedOut[i] = m.num2aa[ts.masked[i]];
    return SP_OK;
}

int sp_host_evalue(const char *matrixName, int gapOpen, int gapExtend, uint64_t dbResidues, double score, double qLen, double *evalue, double *bitScore) {
    sp::Evaluer e; std::string err, merr;
    sp::SubMat m; const bool aa = m.load(matrixName, 2.0, 0.0, merr);      // (the align matrix as sp_ctx_create loads it; "nucleotide.out" is not an amino-acid table: name only)
    if (!e.init(aa ? m.name : std::string(matrixName), gapOpen, gapExtend, dbResidues, err, aa ? sp::matrixDigest(m) : 0)) { g_err = err; return SP_ERR_UNSUPPORTED; }
    *evalue = e.evalue(score, qLen); *bitScore = e.bitScore(score);
    return SP_OK;
}

int sp_prefilter_run(int argc, const char **argv) { return sp::runPrefilterModule(argc, argv); }
int sp_align_run(int argc, const char **argv) { return sp::runAlignModule(argc, argv); }
int sp_rescorediagonal_run(int argc, const char **argv) { return sp::runRescoreDiagonalModule(argc, argv); }
int sp_search_run(int argc, const char **argv) { return sp::runSearchModule(argc, argv); }
int sp_extractorfs_run(int argc, const char **argv) { return sp::sets::runExtractOrfs(argc, argv); }
int sp_translatenucs_run(int argc, const char **argv) { return sp::sets::runTranslateNucs(argc, argv); }
int sp_reverseseqbycodon_run(int argc, const char **argv) { return sp::sets::runReverseSeqByCodon(argc, argv); }
int sp_proteinaln2nucl_run(int argc, const char **argv) { return sp::sets::runProteinAln2Nucl(argc, argv); }
int sp_combineprotnuclaln_run(int argc, const char **argv) { return sp::sets::runCombineProtNuclAln(argc, argv); }
int sp_besthitperset_run(int argc, const char **argv) { return sp::sets::runBestHitPerSet(argc, argv); }
int sp_mergeresultsbyset_run(int argc, const char **argv) { return sp::sets::runMergeResultsBySet(argc, argv); }
int sp_combinescore_run(int argc, const char **argv) { return sp::sets::runCombineScore(argc, argv); }
int sp_filtermatchbyfdr_run(int argc, const char **argv) { return sp::sets::runFilterMatchByFdr(argc, argv); }
int sp_summarizeresults_run(int argc, const char **argv) { return sp::sets::runSummarizeResults(argc, argv); }

}  // extern "C"

namespace sp { AlignParams alignParamsOf(const sp_ctx *c) { return AlignParams{c->subName, c->par.gap_open, c->par.gap_extend, c->par.evalue}; } }
