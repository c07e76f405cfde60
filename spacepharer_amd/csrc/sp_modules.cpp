// Module drivers: `prefilter`, `align`, `rescorediagonal` with the reference's command lines on MMseqs2 DB triples
// (SURVEY.md Appendix A/B).  They replace the bodies of
//   lib/mmseqs/src/prefiltering/Main.cpp:13-68            (prefilter)
//   lib/mmseqs/src/alignment/Main.cpp:12-32               (align)
//   lib/mmseqs/src/alignment/rescorediagonal.cpp:381-431  (rescorediagonal)
// Host orchestration only: every score comes from the HIP kernels behind the in-memory C ABI.
#include "sp_modules.h"
#include "sp_gpu.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <fstream>
#include <map>
#include <functional>
#include <memory>
#include <set>
#include <sstream>
#include <string>
#include <dlfcn.h>
#include <signal.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/spacepharer_amd.h"
#include "sp_db.h"

extern char **environ;

namespace sp {

static const std::set<std::string> kPrefilterFlags = {
    "--sub-mat", "--seed-sub-mat", "-s", "-k", "--k-score", "--alph-size", "--max-seq-len", "--max-seqs", "--split", "--split-mode",
    "--split-memory-limit", "-c", "--cov-mode", "--comp-bias-corr", "--diag-score", "--exact-kmer-matching", "--mask", "--mask-lower-case",
    "--min-ungapped-score", "--add-self-matches", "--spaced-kmer-mode", "--db-load-mode", "--pca", "--pcb", "--spaced-kmer-pattern",
    "--local-tmp", "--threads", "--compressed", "-v"};
static const std::set<std::string> kAlignFlags = {
    "--sub-mat", "-a", "--alignment-mode", "--alignment-output-mode", "--wrapped-scoring", "-e", "--min-seq-id", "--min-aln-len", "--seq-id-mode",
    "--alt-ali", "-c", "--cov-mode", "--max-seq-len", "--comp-bias-corr", "--max-rejected", "--max-accept", "--add-self-matches", "--db-load-mode",
    "--pca", "--pcb", "--score-bias", "--realign", "--realign-score-bias", "--realign-max-seqs", "--gap-open", "--gap-extend", "--zdrop",
    "--threads", "--compressed", "-v"};
static const std::set<std::string> kRescoreFlags = {
    "--sub-mat", "--rescore-mode", "--wrapped-scoring", "--filter-hits", "-e", "-c", "-a", "--cov-mode", "--min-seq-id", "--min-aln-len",
    "--seq-id-mode", "--add-self-matches", "--sort-results", "--db-load-mode", "--threads", "--compressed", "-v"};

// A search context with its target set staged (index, masked residues, similar-k-mer tables in HBM).  In a one-shot module process
// it lives for the call.  In a serving process (`spacepharer-amd serve`, SP_CONTEXT_CACHE=<n>) the last n of them stay resident,
// keyed by the search parameters and the identity of the target DB files, so that the next call against the same target set -- the
// control search of predictmatch, the next batch of spacers -- finds its index in HBM: 288 GB hold hundreds of phage-scale sets.
// SP_ONESHOT (set by the front end when it runs a module in its own, fresh process -- never by a server): the process ends with this call
static bool oneShotProcess() { static const bool v = getenv("SP_ONESHOT") != nullptr && getenv("SP_CONTEXT_CACHE") == nullptr; return v; }

struct Ctx {
    sp_ctx *c = nullptr; bool cached = false; bool targetsStaged = false;
    // (also in a process that ends with this call: left to the driver at exit, the 10 GB of device buffers are reclaimed BEHIND the exit -- and the next fresh process, the
    // workflow's following module, then waits up to 2 s for its own first allocations; measured in round 6)
    ~Ctx() { if (c && !cached) sp_ctx_destroy(c); }
};
struct CachedCtx { std::string key; sp_ctx *c; uint64_t stamp; std::string target; bool indexed; };
static std::vector<CachedCtx> g_ctxCache; static uint64_t g_ctxStamp = 0; static std::mutex g_ctxMutex;
static size_t ctxCacheSize() { const char *e = getenv("SP_CONTEXT_CACHE"); return e ? (size_t) std::max(0, atoi(e)) : 0; }
static std::string fileIdentity(const std::string &path) {
    std::string id; struct stat st;
    for (const char *ext : {"", ".0", ".index", ".dbtype"}) if (stat((path + ext).c_str(), &st) == 0) id += std::to_string((long long) st.st_size) + ":" + std::to_string((long long) st.st_mtim.tv_sec) + "." + std::to_string((long long) st.st_mtim.tv_nsec) + ";";
    char real[4096]; return std::string(realpath(path.c_str(), real) ? real : path.c_str()) + "|" + id;
}
// Serving (`spacepharer-amd serve`): what the last `prefilter` call left behind.  The context it used still holds the query batch, the hit list and
// the targets in HBM; when the next call is `align` on exactly that result DB (same query / target / result files by path, size and mtime -- what
// blastp.sh:33,41 issues back to back) the hits are aligned where they are: no DB is read or parsed again, nothing is uploaded.  Any other use of the
// context forgets the memo.
struct PrefilterMemo { sp_ctx *c = nullptr; std::string query, target, result; std::vector<uint32_t> qkeys, tkeys; };
// (size and mtime miss a same-size rewrite within the clock's granularity: the result DB's identity also carries a digest of its index file)
static std::string resultIdentity(const std::string &path) {
    std::string id = fileIdentity(path);
    // (eight bytes per step, four independent lanes: the index of 6 x 10^5 entries is 12 MB and a byte-wise hash of it cost 9 ms of every served call -- twice)
    uint64_t h[4] = {1469598103934665603ull, 0x9E3779B97F4A7C15ull, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull};
    if (FILE *in = fopen((path + ".index").c_str(), "rb")) {
        std::vector<char> buf(1u << 20);
        size_t n;
        while ((n = fread(buf.data(), 1, buf.size(), in)) > 0) {
            size_t i = 0;
            for (; i + 32 <= n; i += 32) {
                uint64_t w[4]; memcpy(w, buf.data() + i, 32);
                for (int k = 0; k < 4; k++) { h[k] = (h[k] ^ w[k]) * 0x100000001B3ull; h[k] ^= h[k] >> 29; }
            }
            for (; i < n; i++) { h[0] = (h[0] ^ (unsigned char) buf[i]) * 0x100000001B3ull; }
            h[1] += n;
        }
        fclose(in);
    }
    const uint64_t all = (h[0] * 31 + h[1]) ^ (h[2] * 0x9E3779B97F4A7C15ull + h[3]);
    return id + "#" + std::to_string((unsigned long long) all);
}
static PrefilterMemo g_memo;
static void forgetPrefilter(const sp_ctx *c) { if (!c || g_memo.c == c) g_memo = PrefilterMemo(); }
static void rememberPrefilter(sp_ctx *c, const std::string &q, const std::string &t, const std::string &r, const std::vector<uint32_t> &qkeys, const std::vector<uint32_t> &tkeys) {
    g_memo.c = c; g_memo.query = fileIdentity(q); g_memo.target = fileIdentity(t); g_memo.result = resultIdentity(r); g_memo.qkeys = qkeys; g_memo.tkeys = tkeys;
}

// a context for (params, target DB): from the cache when serving, else a fresh one.  what = "index" (searchable) or "residues"
static bool acquireCtx(const sp_search_params &p, const std::string &targetPath, const char *what, Ctx &ctx, std::string &err) {
    const size_t cap = ctxCacheSize();
    std::string key;
    if (cap) {
        key = std::string(what) + "|" + p.seed_sub_mat + "|" + p.sub_mat + "|" + std::to_string(p.sensitivity) + "|" + std::to_string(p.kmer_score) + "|" + std::to_string(p.max_seqs) + "|" +
              std::to_string(p.min_ungapped_score) + "|" + std::to_string(p.mask) + "|" + std::to_string(p.gap_open) + "|" + std::to_string(p.gap_extend) + "|" + std::to_string(p.evalue) + "|" +
              std::to_string(p.threads) + "|" + fileIdentity(targetPath);
        std::lock_guard<std::mutex> lk(g_ctxMutex);
        for (CachedCtx &e : g_ctxCache) if (e.key == key) { e.stamp = ++g_ctxStamp; ctx.c = e.c; ctx.cached = true; ctx.targetsStaged = true; forgetPrefilter(e.c); return true; }
    }
    ctx.c = sp_ctx_create(&p);
    if (!ctx.c && cap) {      // serving: the resident target sets may be what fills the device -- let go of them and try once more
        releaseCachedContexts();
        ctx.c = sp_ctx_create(&p);
    }
    if (!ctx.c) { err = sp_last_error(); return false; }
    if (cap) {
        std::lock_guard<std::mutex> lk(g_ctxMutex);
        // the same target DB may already be resident for another parameter set or another module (`prefilter` staged it with its index, now `align` asks for the residues):
        // the new context SHARES that staged set (sp_ctx_share_targets) instead of reading, mapping and uploading the DB again
        const std::string tid = fileIdentity(targetPath); const bool wantIndex = std::string(what) == "index";
        for (CachedCtx &e : g_ctxCache) if (e.target == tid && (e.indexed || !wantIndex) && sp_ctx_share_targets(ctx.c, e.c) == SP_OK) { ctx.targetsStaged = true; break; }
        if (g_ctxCache.size() >= cap) {      // evict the least recently used
            size_t lru = 0; for (size_t i = 1; i < g_ctxCache.size(); i++) if (g_ctxCache[i].stamp < g_ctxCache[lru].stamp) lru = i;
            forgetPrefilter(g_ctxCache[lru].c);
            sp_ctx_destroy(g_ctxCache[lru].c); g_ctxCache.erase(g_ctxCache.begin() + (long) lru);
        }
        g_ctxCache.push_back({key, ctx.c, ++g_ctxStamp, tid, wantIndex});
        ctx.cached = true;
    }
    return true;
}
// a cached context whose staging failed must not be handed out again
static void dropCtx(Ctx &ctx) {
    if (!ctx.cached || !ctx.c) return;
    std::lock_guard<std::mutex> lk(g_ctxMutex);
    for (size_t i = 0; i < g_ctxCache.size(); i++) if (g_ctxCache[i].c == ctx.c) { g_ctxCache.erase(g_ctxCache.begin() + (long) i); break; }
    forgetPrefilter(ctx.c);
    sp_ctx_destroy(ctx.c); ctx.c = nullptr; ctx.cached = false;
}
void releaseCachedContexts() {
    std::lock_guard<std::mutex> lk(g_ctxMutex);
    forgetPrefilter(nullptr);
    for (CachedCtx &e : g_ctxCache) sp_ctx_destroy(e.c);
    g_ctxCache.clear();
}

static bool loadSeqDb(const std::string &path, Db &db, std::string &err) {
    if (!db.load(path, err)) return false;
    if (db.type() != 0) { err = path + ": only amino-acid sequence DBs (dbtype 0) are supported, got " + std::to_string(db.type()); return false; }
    if (db.compressed()) { err = path + ": compressed DBs are not supported (the workflows pass --compressed 0)"; return false; }
    return true;
}

static void flatten(const Db &db, std::vector<char> &res, std::vector<uint64_t> &off, std::vector<uint32_t> &keys) {
    // two passes over contiguous ranges of entries on a few threads -- lengths, then the copy: one entry at a time on one thread was 10 of the 38 ms a served
    // `prefilter` of 6 x 10^5 peptides spends in its process
    const size_t n = db.entries.size();
    const int threads = n >= (1u << 16) ? std::min(16, defaultThreads()) : 1;
    off.assign(n + 1, 0); keys.resize(n);
    parallelChunks(n, threads, [&](size_t, size_t from, size_t to) {
        for (size_t i = from; i < to; i++) {
            const char *p = db.ptr(i); uint64_t l = db.seqLen(i);                    // Sequence::mapSequence stops at \n / \0
            if (const void *nl = memchr(p, '\n', (size_t) l)) l = (uint64_t) ((const char *) nl - p);
            if (const void *z = memchr(p, '\0', (size_t) l)) l = (uint64_t) ((const char *) z - p);
            off[i + 1] = l; keys[i] = db.entries[i].key;
        }
    });
    for (size_t i = 0; i < n; i++) off[i + 1] += off[i];
    res.resize(off[n]);
    parallelChunks(n, threads, [&](size_t, size_t from, size_t to) { for (size_t i = from; i < to; i++) memcpy(res.data() + off[i], db.ptr(i), (size_t) (off[i + 1] - off[i])); });
    if (res.empty()) res.push_back('X');
}

// A sequence DB as the search takes it (residues back to back, offsets, keys).  A resident server keeps the last few by file identity (real path, sizes and
// modification times of the data, index and dbtype files -- what the staged target sets are keyed by): predictmatch searches the SAME query set against the targets
// and then against the control set, and reading + parsing + flattening 6 x 10^5 entries is 9 of the 27 ms of a served `prefilter` call.
struct FlatDb { std::string id; std::vector<char> res; std::vector<uint64_t> off; std::vector<uint32_t> keys; uint64_t stamp = 0; };
static std::vector<std::shared_ptr<FlatDb>> g_flatCache; static uint64_t g_flatStamp = 0;
static std::shared_ptr<const FlatDb> flatSeqDb(const std::string &path, std::string &err) {
    const bool serving = ctxCacheSize() > 0;
    const std::string id = serving ? fileIdentity(path) : std::string();
    if (serving) {
        std::lock_guard<std::mutex> lk(g_ctxMutex);
        for (auto &e : g_flatCache) if (e->id == id) { e->stamp = ++g_flatStamp; return e; }
    }
    Db db;
    if (!loadSeqDb(path, db, err)) return nullptr;
    auto f = std::make_shared<FlatDb>();
    flatten(db, f->res, f->off, f->keys);
    if (serving) {
        f->id = id;
        std::lock_guard<std::mutex> lk(g_ctxMutex);
        f->stamp = ++g_flatStamp;
        if (g_flatCache.size() >= 4) { size_t old = 0; for (size_t i = 1; i < g_flatCache.size(); i++) if (g_flatCache[i]->stamp < g_flatCache[old]->stamp) old = i; g_flatCache.erase(g_flatCache.begin() + (long) old); }
        g_flatCache.push_back(f);
    }
    return f;
}

// Path of the front-end binary that workers are started from: $SP_WORKER_EXE, else `spacepharer-amd` next to this library.
static std::string workerExe() {
    if (const char *e = getenv("SP_WORKER_EXE")) return e;
    Dl_info info;
    if (dladdr((const void *) &workerExe, &info) && info.dli_fname) {
        std::string p = info.dli_fname; const size_t s = p.rfind('/');
        p = (s == std::string::npos ? std::string(".") : p.substr(0, s)) + "/spacepharer-amd";
        if (access(p.c_str(), X_OK) == 0) return p;
    }
    return "";
}

static void removeOutputs(const std::string &out, int keepParts) {
    remove(out.c_str()); remove((out + ".index").c_str()); remove((out + ".dbtype").c_str());
    for (int i = 0;; i++) {
        const std::string dp = out + "." + std::to_string(i);
        struct stat st; const bool have = stat(dp.c_str(), &st) == 0;
        remove((dp + ".idxpart").c_str()); remove((dp + ".typepart").c_str());
        if (!have && i >= std::max(keepParts, 64)) break;      // parts of an earlier, wider split (a node has at most a few dozen GPUs)
        if (have) remove(dp.c_str());
    }
}

// SP_GPUS=N and not yet a worker: start one worker PROCESS per GPU, wait, merge.  Returns -1 when the call is not to be fanned
// out.  Workers are fresh processes of the front-end binary (posix_spawn: fork + exec in one step, the way a shell starts a
// child) with SP_PART=r/N and LOCAL_RANK=r -- never a bare fork(): the caller may already have initialised the HIP runtime
// (sp_init / a live sp_ctx), and a forked copy of an initialised runtime is undefined.
static int fanOut(const char *module, int argc, const char **argv, const std::string &out) {
    const char *e = getenv("SP_GPUS");
    const int n = e ? atoi(e) : 1;
    if (n <= 1 || getenv("SP_PART")) return -1;
    const std::string exe = workerExe();
    if (exe.empty()) return fail(std::string(module) + ": SP_GPUS=" + std::to_string(n) + " needs the worker binary (spacepharer-amd next to the library, or $SP_WORKER_EXE)");
    // leftovers of an earlier (crashed, or differently split) run must not be mixed with this run's parts: readers prefer an
    // unsplit <out> over <out>.0.., and parts with rank >= N would be appended to the data
    removeOutputs(out, n);
    std::string distId;
    {   unsigned char id[SP_DIST_ID_BYTES];
        if (sp_dist_unique_id(id) == SP_OK) { static const char *hx = "0123456789abcdef"; for (unsigned char b : id) { distId.push_back(hx[b >> 4]); distId.push_back(hx[b & 15]); } } }
    std::vector<pid_t> kids;
    auto reap = [&](bool kill9) { bool ok = true; for (pid_t k : kids) { if (kill9) kill(k, SIGKILL); int st = 0; if (waitpid(k, &st, 0) < 0 || !WIFEXITED(st) || WEXITSTATUS(st) != 0) ok = false; } kids.clear(); return ok; };
    for (int r = 0; r < n; r++) {
        std::vector<std::string> envs;
        for (char **ev = environ; *ev; ev++) if (strncmp(*ev, "SP_PART=", 8) && strncmp(*ev, "LOCAL_RANK=", 11) && strncmp(*ev, "SP_SERVER=", 10) && strncmp(*ev, "SP_DIST_ID=", 11)) envs.push_back(*ev);      // (a worker runs its part itself: no forwarding to a server)
        envs.push_back("SP_PART=" + std::to_string(r) + "/" + std::to_string(n));
        if (!distId.empty()) envs.push_back("SP_DIST_ID=" + distId);      // target shards (SP_SHARD=targets): the workers exchange candidates through sp_dist
        envs.push_back("LOCAL_RANK=" + std::to_string(r));          // device = LOCAL_RANK mod visible devices (sp_init)
        std::vector<char *> envp; for (std::string &s : envs) envp.push_back(&s[0]); envp.push_back(nullptr);
        std::vector<std::string> args = {exe, module}; for (int i = 0; i < argc; i++) args.push_back(argv[i]);
        std::vector<char *> av; for (std::string &s : args) av.push_back(&s[0]); av.push_back(nullptr);
        pid_t pid = 0;
        if (posix_spawn(&pid, exe.c_str(), nullptr, nullptr, av.data(), envp.data()) != 0) {
            reap(true);                                             // do not leave started workers writing behind a failed call
            removeOutputs(out, n);
            return fail(std::string(module) + ": cannot start worker " + exe);
        }
        kids.push_back(pid);
    }
    if (!reap(false)) { removeOutputs(out, n); return fail(std::string(module) + ": a per-GPU worker failed"); }
    // merge: parts are data files <out>.0 .. <out>.N-1 read back to back, so part r's offsets shift by the sizes before it
    std::vector<DbEntry> all; uint64_t shift = 0; int32_t dbtype = -1;
    for (int r = 0; r < n; r++) {
        const std::string dp = out + "." + std::to_string(r);
        std::ifstream ix(dp + ".idxpart");
        if (!ix.good()) return fail(std::string(module) + ": missing " + dp + ".idxpart");
        unsigned long long k, o, l;
        while (ix >> k >> o >> l) { DbEntry en; en.key = (uint32_t) k; en.offset = o + shift; en.length = l; all.push_back(en); }
        struct stat st; if (stat(dp.c_str(), &st) != 0) return fail(std::string(module) + ": missing " + dp);
        shift += (uint64_t) st.st_size;
        std::ifstream tp(dp + ".typepart", std::ios::binary); int32_t v = -1; tp.read((char *) &v, 4); if (r == 0) dbtype = v;
        remove((dp + ".idxpart").c_str()); remove((dp + ".typepart").c_str());
    }
    std::sort(all.begin(), all.end(), [](const DbEntry &a, const DbEntry &b) { return a.key < b.key; });
    { std::ofstream o(out + ".index", std::ios::binary); if (!o.good()) return fail(std::string(module) + ": cannot write " + out + ".index");
      for (const DbEntry &en : all) o << en.key << '\t' << en.offset << '\t' << en.length << '\n'; }
    { std::ofstream o(out + ".dbtype", std::ios::binary); o.write((const char *) &dbtype, 4); if (!o.good()) return fail(std::string(module) + ": cannot write " + out + ".dbtype"); }
    return 0;
}

// Alignment records + CIGAR text of a module call.  A one-shot process takes ordinary memory (page-locking tens of megabytes costs more, 0.23 ms per MB, than
// one run gains from the faster copy); a resident server keeps ONE page-locked pair for all its calls: the device writes into it at the link's rate (a copy into
// pageable memory is staged by the runtime at a fraction of it -- 12 of the 19 ms of a served `align` at 10^5 spacers) and no call touches fresh pages.
struct ResultBufs {
    sp_alignment *res = nullptr; char *cig = nullptr; uint64_t resCap = 0, cigCap = 0; bool pinned = false;
    void release() { if (pinned) { sp_host_free(res); sp_host_free(cig); } else { free(res); free(cig); } res = nullptr; cig = nullptr; resCap = cigCap = 0; }
    bool ensure(uint64_t nRes, uint64_t nCig) {
        if (nRes <= resCap && nCig <= cigCap) return true;
        const uint64_t r = std::max(resCap, nRes + nRes / 4 + 1024), g = std::max(cigCap, nCig + nCig / 4 + 1024);
        release();
        res = static_cast<sp_alignment *>(pinned ? sp_host_alloc(r * sizeof(sp_alignment)) : malloc(r * sizeof(sp_alignment)));
        cig = static_cast<char *>(pinned ? sp_host_alloc(g) : malloc(g));
        if (!res || !cig) { release(); sp::setLastError("out of memory for the alignment records"); return false; }
        resCap = r; cigCap = g;
        return true;
    }
};
static ResultBufs g_servedBufs;      // (a server runs one module call at a time)
struct ResultBufsRef {               // the served pair, or a pair of this call's own that goes with it
    ResultBufs own, *b;
    explicit ResultBufsRef(bool serving) : b(serving ? &g_servedBufs : &own) { if (serving) g_servedBufs.pinned = true; }
    ~ResultBufsRef() { own.release(); }
};

// Result text formatted on the device (sp_ctx_hits_text / sp_ctx_align_hits_text): the page-locked buffer it lands in is handed to the DbWriter as it is (addExternal)
// and must outlive the writer's close() -- it belongs to the module call, or to the server between calls.  SP_DEVICE_TEXT=0: the host formats (the round-4 path, also
// taken for whatever the device path declines: --max-accept / --max-rejected, runs of several batches, target shards, queries retried through shards, batches with a pair
// that needs the host-driven traceback).
// (a one-shot process takes ordinary memory here as for the records above: page-locking the 150-550 MB these buffers are sized for costs 40-130 ms, the staged copy of the
// text that is really there 20)
struct TextBufs {
    char *text = nullptr; uint32_t *lens = nullptr; uint64_t textCap = 0, lenCap = 0;
    char *index = nullptr; uint64_t indexCap = 0;
    const bool pinned = !oneShotProcess();
    void *take(uint64_t n) { return pinned ? sp_host_alloc(n) : malloc(n); }
    void give(void *p) { if (pinned) sp_host_free(p); else free(p); }
    void release() { give(text); give(lens); give(index); text = nullptr; lens = nullptr; index = nullptr; textCap = lenCap = indexCap = 0; }
    bool ensureIndex(uint64_t n) {
        if (n <= indexCap) return true;
        give(index); indexCap = n + n / 4 + 4096; index = static_cast<char *>(take(indexCap));
        if (!index) { indexCap = 0; sp::setLastError("out of page-locked memory for the index text"); return false; }
        return true;
    }
    bool ensure(uint64_t nText, uint64_t nLens) {
        if (nText > textCap) { give(text); textCap = nText + nText / 4 + 4096; text = static_cast<char *>(take(textCap)); if (!text) { textCap = 0; sp::setLastError("out of page-locked memory for the result text"); return false; } }
        if (nLens > lenCap) { give(lens); lenCap = nLens + nLens / 4 + 1024; lens = static_cast<uint32_t *>(take(lenCap * sizeof(uint32_t))); if (!lens) { lenCap = 0; sp::setLastError("out of page-locked memory for the result text"); return false; } }
        return true;
    }
};
static TextBufs g_servedText;
struct TextBufsRef {
    TextBufs own, *b;
    explicit TextBufsRef(bool serving) : b(serving ? &g_servedText : &own) {}
    ~TextBufsRef() { own.release(); }
};
static bool deviceTextEnabled() { const char *e = getenv("SP_DEVICE_TEXT"); return !(e && *e == '0'); }
// the entries of the context's resident batch as one external block of `w`: 1 = added, 0 = the device path declined (the caller formats on the host), -1 = error
static int deviceTextBlock(sp_ctx *c, bool alignments, double minSeqId, long minAlnLen, const std::vector<uint32_t> &keys, TextBufs &tb, DbWriter &w) {
    const uint64_t nHits = sp_ctx_num_hits(c);
    // (first guess: 40 / 20 bytes a line -- an alignment line of a spacer peptide has 27, a prefilter line 15; the formatter says what it needs when that is too little and the
    // second attempt below has it.  Sized for the worst line, 72 / 24 bytes, a resident server's first call page-locked 550 MB for 170 MB of text.)
    if (!tb.ensure(nHits * (alignments ? 40 : 20) + keys.size() + (1u << 16), keys.size() + 1)) return -1;
    for (int attempt = 0; attempt < 2; attempt++) {
        uint64_t used = 0;
        const int rc = alignments ? sp_ctx_align_hits_text(c, minSeqId, (uint32_t) std::max<long>(0, minAlnLen), tb.text, tb.textCap, &used, tb.lens)
                                  : sp_ctx_hits_text(c, tb.text, tb.textCap, &used, tb.lens);
        if (rc == SP_OK) {
            w.addExternal(keys.data(), tb.lens, keys.size(), tb.text, (size_t) used);
            // the index text too while the lengths are on the device (the writer takes it when this block turns out to be the whole DB with ascending keys)
            if (w.part.count <= 1 && tb.ensureIndex(keys.size() * 24 + 64)) {
                uint64_t iu = 0;
                int ri = sp_ctx_text_index(c, keys.data(), keys.size(), 0, tb.index, tb.indexCap, &iu);
                if (ri == SP_ERR_OVERFLOW && tb.ensureIndex(iu + 16)) ri = sp_ctx_text_index(c, keys.data(), keys.size(), 0, tb.index, tb.indexCap, &iu);
                if (ri == SP_OK) w.setExternalIndex(tb.index, (size_t) iu, keys.size());
            }
            return 1;
        }
        if (rc == SP_ERR_UNSUPPORTED) return 0;
        if (rc != SP_ERR_OVERFLOW || attempt == 1 || !tb.ensure(used + 16, keys.size() + 1)) return -1;
    }
    return -1;
}

// a sequence DB as the in-memory interface takes it (sp_flat_db_open): residues back to back, offsets, keys -- and, when <path>.lookup exists, the set number of
// every entry (third column, commons/DBReader.h lookup format "key\tname\tset") in entry order
bool loadFlatForAbi(const std::string &path, std::vector<char> &res, std::vector<uint64_t> &off, std::vector<uint32_t> &keys, std::vector<uint32_t> &sets, std::string &err) {
    Db db;
    if (!db.load(path, err)) return false;
    if (db.type() != 0 && db.type() != 1) { err = path + ": a sequence DB (dbtype 0 amino acids or 1 nucleotides) is expected, got " + std::to_string(db.type()); return false; }
    if (db.compressed()) { err = path + ": compressed DBs are not supported (the workflows pass --compressed 0)"; return false; }
    flatten(db, res, off, keys);
    sets.clear();
    MappedFile lk;
    if (lk.open(path + ".lookup")) {
        // lines "key\tname\tset"; entries of the DB in key order hold the sets in the same order when the lookup is keyed alike (createsetdb / extractorfs --create-lookup 1)
        std::vector<std::pair<uint32_t, uint32_t>> kv; kv.reserve(keys.size());
        const char *b = lk.p, *e = lk.p + lk.n;
        while (b < e) {
            const char *nl = static_cast<const char *>(memchr(b, '\n', (size_t) (e - b))); if (!nl) nl = e;
            const char *t1 = static_cast<const char *>(memchr(b, '\t', (size_t) (nl - b)));
            const char *t2 = t1 ? static_cast<const char *>(memchr(t1 + 1, '\t', (size_t) (nl - t1 - 1))) : nullptr;
            if (t1 && t2) kv.emplace_back((uint32_t) strtoul(b, nullptr, 10), (uint32_t) strtoul(t2 + 1, nullptr, 10));
            b = nl + 1;
        }
        const bool sortedKv = std::is_sorted(kv.begin(), kv.end());
        if (!sortedKv) std::sort(kv.begin(), kv.end());
        sets.resize(keys.size());
        for (size_t i = 0; i < keys.size(); i++) {
            auto it = std::lower_bound(kv.begin(), kv.end(), std::make_pair(keys[i], 0u));
            if (it == kv.end() || it->first != keys[i]) { err = path + ".lookup has no line for key " + std::to_string(keys[i]); return false; }
            sets[i] = it->second;
        }
    }
    return true;
}

// SP_MODULE_TIMES=1: wall-clock laps of a module run on stderr
struct Laps {
    bool on = getenv("SP_MODULE_TIMES") != nullptr; const char *module;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit Laps(const char *m) : module(m) {}
    void memory(const char *what) {      // device memory in use (all processes): what a fresh process asks the driver for, and what the driver clears behind it
        if (!on) return;
        size_t fr = 0, tot = 0;
        if (sp::deviceMemory(fr, tot)) fprintf(stderr, "[sp] %s: %-28s %8.2f GB of device memory in use\n", module, what, (double) (tot - fr) / 1e9);
    }
    void lap(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[sp] %s: %-28s %8.3f ms\n", module, what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// ---------------------------------------------------------------------------------------------------------------
// Queries whose hit stream does not fit one pass (sp_ctx_overflow_queries; never a spacer peptide -- a few hundred residues against a
// dense index) are searched against target shards: a second context indexes one target range after the other, the candidates of all
// ranges are merged by sp_ctx_merge_candidates, whose result equals the unsplit search bit for bit.  The context returned is in the
// state sp_ctx_prefilter() leaves for the sub-batch `which` (in that order); the caller formats / aligns from it and destroys it.
// ---------------------------------------------------------------------------------------------------------------
static sp_ctx *searchByShards(const sp_search_params &p, const std::vector<char> &tres, const std::vector<uint64_t> &toff, const std::vector<uint32_t> &tkeys,
                              const char *qres, const uint64_t *qoff /* offsets of the whole batch */, const std::vector<uint32_t> &which, std::string &err) {
    std::vector<char> sub; std::vector<uint64_t> subOff(1, 0);
    for (uint32_t q : which) { sub.insert(sub.end(), qres + qoff[q], qres + qoff[q + 1]); subOff.push_back(sub.size()); }
    if (sub.empty()) sub.push_back('X');
    const size_t nt = tkeys.size();
    for (size_t shards = 8; shards <= 512; shards *= 2) {
        sp_ctx *c = sp_ctx_create(&p);
        if (!c) { err = sp_last_error(); return nullptr; }
        std::vector<sp_candidate> all; bool again = false, failed = false;
        const uint64_t total = toff[nt];
        size_t from = 0;
        for (size_t s = 0; s < shards && !again && !failed; s++) {      // contiguous target ranges of about equal residue counts
            size_t to = from;
            const uint64_t upTo = total * (s + 1) / shards;
            while (to < nt && (toff[to + 1] <= upTo || to == from)) to++;
            if (s + 1 == shards) to = nt;
            if (to == from) continue;
            if (sp_ctx_set_targets_shard(c, tres.data(), toff.data(), tkeys.data(), nt, from, to) != SP_OK || sp_ctx_set_queries(c, sub.data(), subOff.data(), which.size()) != SP_OK) { failed = true; break; }
            const int rc = sp_ctx_prefilter_candidates(c);
            if (rc == SP_ERR_OVERFLOW) { again = true; break; }
            if (rc != SP_OK) { failed = true; break; }
            const size_t n = (size_t) sp_ctx_num_candidates(c), at = all.size();
            all.resize(at + n);
            if (n && sp_ctx_get_candidates(c, all.data() + at) != SP_OK) { failed = true; break; }
            from = to;
        }
        if (!again && !failed) {
            if (all.empty()) all.push_back(sp_candidate{0xFFFFFFFFu, 0, 0, 0, 0});      // (padding record: nothing matched)
            if (sp_ctx_merge_candidates(c, all.data(), all.size(), 0, 0, which.size()) == SP_OK) return c;
            failed = true;
        }
        if (failed) { err = sp_last_error(); sp_ctx_destroy(c); return nullptr; }
        sp_ctx_destroy(c);      // a range still overflowed: twice as many, narrower ranges
    }
    err = "a query exceeds the hit stream of one pass even against 1/512 of the target set";
    return nullptr;
}
// the batch without the refused queries: residues, offsets and the positions (in the batch) that remain
static void withoutQueries(const char *qres, const uint64_t *qoff, size_t n, const std::vector<uint32_t> &refused, std::vector<char> &res, std::vector<uint64_t> &off, std::vector<uint32_t> &kept) {
    std::vector<char> drop(n, 0); for (uint32_t q : refused) if (q < n) drop[q] = 1;
    res.clear(); off.assign(1, 0); kept.clear();
    for (size_t q = 0; q < n; q++) if (!drop[q]) { res.insert(res.end(), qres + qoff[q], qres + qoff[q + 1]); off.push_back(res.size()); kept.push_back((uint32_t) q); }
    if (res.empty()) res.push_back('X');
}

// ---------------------------------------------------------------------------------------------------------------
// prefilter <queryDB> <targetDB> <outDB>
// ---------------------------------------------------------------------------------------------------------------
int runPrefilterModule(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kPrefilterFlags, 3, f, err)) return fail("prefilter: " + err);
    if (!requireValue(f, "-k", {"6", "0"}, err) || !requireValue(f, "--alph-size", {"21"}, err) || !requireValue(f, "--comp-bias-corr", {"1"}, err) ||
        !requireValue(f, "--diag-score", {"1"}, err) || !requireValue(f, "--exact-kmer-matching", {"0"}, err) || !requireValue(f, "--spaced-kmer-mode", {"1"}, err) ||
        !requireValue(f, "--spaced-kmer-pattern", {"11011101", ""}, err) || !requireValue(f, "--mask-lower-case", {"0"}, err) || !requireValue(f, "--compressed", {"0"}, err) ||
        !requireValue(f, "--add-self-matches", {"0"}, err) || !requireValue(f, "--split", {"0", "1"}, err) || !requireValue(f, "-c", {"0", "0.0", "0.000"}, err))
        return fail("prefilter: " + err);
    { const int rc = fanOut("prefilter", argc, argv, f.positional[2]); if (rc >= 0) return rc; }
    Laps laps("prefilter");
    std::shared_ptr<const FlatDb> qf = flatSeqDb(f.positional[0], err), tf = qf ? flatSeqDb(f.positional[1], err) : nullptr;
    if (!qf || !tf) return fail("prefilter: " + err);
    laps.lap("read DBs");
    sp_search_params p; sp_default_params(&p);
    std::string seed = aaPart(f.str("--seed-sub-mat", "VTML80.out")), sub = aaPart(f.str("--sub-mat", "VTML40.out"));
    p.seed_sub_mat = seed.c_str(); p.sub_mat = sub.c_str();
    p.sensitivity = (float) f.real("-s", 5.7);
    p.kmer_score = (int) f.num("--k-score", INT_MAX);
    p.max_seqs = (int) f.num("--max-seqs", 300);
    p.min_ungapped_score = (int) f.num("--min-ungapped-score", 15);
    p.mask = (int) f.num("--mask", 1);
    p.threads = (int) f.num("--threads", 0);
    Ctx ctx;
    if (!acquireCtx(p, f.positional[1], "index", ctx, err)) return fail("prefilter: " + err);
    laps.lap("context (matrices, tables)");
    const std::vector<char> &tres = tf->res, &qres = qf->res; const std::vector<uint64_t> &toff = tf->off, &qoff = qf->off; const std::vector<uint32_t> &tkeys = tf->keys, &qkeys = qf->keys;
    if (!ctx.targetsStaged && sp_ctx_set_targets(ctx.c, tres.data(), toff.data(), tkeys.data(), tkeys.size()) != SP_OK) {
        // serving: cached target sets of earlier calls may be what fills the device; a fresh process would have succeeded, so drop them all and stage again
        std::string e = sp_last_error();
        const bool retry = ctx.cached;
        dropCtx(ctx);
        if (retry) {
            releaseCachedContexts();
            if (!acquireCtx(p, f.positional[1], "index", ctx, err)) return fail("prefilter: " + err);
            if (sp_ctx_set_targets(ctx.c, tres.data(), toff.data(), tkeys.data(), tkeys.size()) == SP_OK) e.clear(); else { e = sp_last_error(); dropCtx(ctx); }
        }
        if (!e.empty()) return fail("prefilter: " + e);
    }
    laps.lap("targets -> device, index");
    DbWriter w; w.path = f.positional[2]; w.dbtype = 7;      // DBTYPE_PREFILTER_RES
    const size_t batch = 1u << 20;
    size_t qFrom = 0, nq = qkeys.size();
    { std::vector<uint64_t> lens(qkeys.size()); for (size_t i = 0; i < lens.size(); i++) lens[i] = qoff[i + 1] - qoff[i] + 2; partRange(lens, w.part, qFrom, nq); }   // this GPU's queries
    // hits of the context's current batch -> one entry per query key (QueryMatcher::prefilterHitToBuffer, QueryMatcher.h:114-126), read from the
    // device's own packed records in place and formatted on --threads workers over contiguous query ranges (the blocks go to the writer in order)
    const int fmtThreads = (int) f.num("--threads", 0) > 0 ? (int) f.num("--threads", 0) : defaultThreads();
    auto writeBatch = [&](sp_ctx *c, const std::vector<uint32_t> &keys) -> bool {
        const sp_hit_packed *hits = nullptr; const uint32_t *ho = nullptr; uint64_t nHits = 0;
        if (sp_ctx_hits_view(c, &hits, &ho, &nHits) != SP_OK) return false;
        laps.lap("  hit list on the host");
        const size_t nChunks = numChunks(keys.size(), fmtThreads);
        std::vector<std::string> blocks = DbWriter::pooledBlocks(nChunks);
        static thread_local std::vector<std::vector<uint32_t>> keyPool; static thread_local std::vector<std::vector<uint64_t>> lenPool;      // (kept with their capacity, like the blocks)
        std::vector<std::vector<uint32_t>> &bKeys = keyPool; std::vector<std::vector<uint64_t>> &bLens = lenPool;      // the CALLING thread's pools: the workers below see them through these references
        bKeys.resize(std::max(bKeys.size(), nChunks)); bLens.resize(std::max(bLens.size(), nChunks));
        for (size_t ci = 0; ci < nChunks; ci++) { bKeys[ci].clear(); bLens[ci].clear(); }
        parallelChunks(keys.size(), fmtThreads, [&](size_t ci, size_t from, size_t to) {
            // (the chunk's buffers as LOCALS of the worker: the headers of neighbouring std::strings / vectors share cache lines, and every append
            // of one thread would pull the line out of its neighbour's cache -- 0.5 us per hit line on a two-socket host instead of 0.1)
            std::string block = std::move(blocks[ci]); std::vector<uint32_t> ck = std::move(bKeys[ci]); std::vector<uint64_t> cl = std::move(bLens[ci]); char buf[64];
            block.reserve((size_t) (ho[to] - ho[from]) * 20 + (to - from));
            ck.reserve(to - from); cl.reserve(to - from);
            for (size_t q = from; q < to; q++) {
                const size_t before = block.size();
                for (uint32_t h = ho[q]; h < ho[q + 1]; h++) {
                    char *e = u32toa(tkeys[hits[h].seq_id], buf); *e++ = '\t';
                    e = i32toa((int32_t) hits[h].score, e); *e++ = '\t';
                    e = i32toa((int32_t) hits[h].diagonal, e); *e++ = '\n';
                    block.append(buf, (size_t) (e - buf));
                }
                block.push_back('\0');      // every query key gets an entry, empty ones too
                ck.push_back(keys[q]); cl.push_back(block.size() - before);
            }
            blocks[ci] = std::move(block); bKeys[ci] = std::move(ck); bLens[ci] = std::move(cl);
        });
        laps.lap("  format (threads)");
        for (size_t ci = 0; ci < nChunks; ci++) w.addBlock(bKeys[ci], bLens[ci], std::move(blocks[ci]));
        return true;
    };
    bool oneBatch = false;
    TextBufsRef text(ctx.cached);
    for (size_t b0 = qFrom; b0 < nq; b0 += batch) {
        const size_t b1 = std::min(nq, b0 + batch);
        std::vector<uint64_t> off(qoff.begin() + b0, qoff.begin() + b1 + 1);
        for (uint64_t &o : off) o -= qoff[b0];
        std::vector<uint32_t> keys(qkeys.begin() + b0, qkeys.begin() + b1);
        if (sp_ctx_set_queries(ctx.c, qres.data() + qoff[b0], off.data(), b1 - b0) != SP_OK) return fail(std::string("prefilter: ") + sp_last_error());
        laps.lap("  queries -> device");
        int rc = sp_ctx_prefilter(ctx.c);
        std::vector<uint32_t> refused(rc == SP_ERR_OVERFLOW ? sp_ctx_overflow_queries(ctx.c, nullptr, 0) : 0);
        if (rc == SP_ERR_OVERFLOW && !refused.empty()) {      // search the batch without them, then them against target shards
            sp_ctx_overflow_queries(ctx.c, refused.data(), refused.size());
            std::vector<char> res2; std::vector<uint64_t> off2; std::vector<uint32_t> kept;
            withoutQueries(qres.data() + qoff[b0], off.data(), b1 - b0, refused, res2, off2, kept);
            if (sp_ctx_set_queries(ctx.c, res2.data(), off2.data(), kept.size()) != SP_OK || sp_ctx_prefilter(ctx.c) != SP_OK) return fail(std::string("prefilter: ") + sp_last_error());
            keys.clear(); for (uint32_t q : kept) keys.push_back(qkeys[b0 + q]);
            if (!writeBatch(ctx.c, keys)) return fail(std::string("prefilter: ") + sp_last_error());
            sp_ctx *r = searchByShards(p, tres, toff, tkeys, qres.data() + qoff[b0], off.data(), refused, err);
            if (!r) return fail("prefilter: " + err);
            keys.clear(); for (uint32_t q : refused) keys.push_back(qkeys[b0 + q]);
            const bool ok = writeBatch(r, keys);
            sp_ctx_destroy(r);
            if (!ok) return fail(std::string("prefilter: ") + sp_last_error());
            continue;
        }
        if (rc != SP_OK) return fail(std::string("prefilter: ") + sp_last_error());
        laps.lap("queries -> device, search");
        int viaDevice = 0;
        if (b0 == qFrom && b1 == nq && deviceTextEnabled()) {      // the whole run is this batch: its entries come as text from the device, in the buffer the writer reads
            viaDevice = deviceTextBlock(ctx.c, false, 0.0, 0, keys, *text.b, w);
            if (viaDevice < 0) return fail(std::string("prefilter: ") + sp_last_error());
            if (viaDevice) laps.lap("  entries formatted on the device");
        }
        if (!viaDevice && !writeBatch(ctx.c, keys)) return fail(std::string("prefilter: ") + sp_last_error());
        oneBatch = b0 == 0 && b1 == qkeys.size();
    }
    laps.lap("format hits");
    if (!w.close(err)) return fail("prefilter: " + err);
    laps.lap("write DB");
    // serving: the context still holds this batch and its hits -- an `align` call on exactly this result DB need not read it back
    if (ctx.cached && oneBatch && w.part.count <= 1) rememberPrefilter(ctx.c, f.positional[0], f.positional[1], f.positional[2], qkeys, tkeys);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// shared by align / rescorediagonal: parse a prefilter result DB into (query index, target index, diagonal) triples
// ---------------------------------------------------------------------------------------------------------------
struct Triples { std::vector<uint32_t> q, t; std::vector<int16_t> d; std::vector<uint64_t> perQuery; /* offsets per result entry */ std::vector<uint32_t> entryQuery; std::vector<uint32_t> entryKey; };

static bool collectTriples(const Db &res, const KeyMap &qIndex, const KeyMap &tIndex, Triples &tr, std::string &err) {
    size_t eFrom = 0, eTo = res.entries.size();
    { std::vector<uint64_t> lens(res.entries.size()); for (size_t i = 0; i < lens.size(); i++) lens[i] = res.entries[i].length; partRange(lens, currentPart(), eFrom, eTo); }   // this GPU's entries
    // entry ranges are parsed on a few threads (740 000 lines of a 100 000-spacer result DB: 30 ms on one), the pieces are joined in entry order
    const int threads = defaultThreads();
    const size_t n = eTo - eFrom, nChunks = numChunks(n, threads);
    std::vector<Triples> part(nChunks); std::vector<std::string> errs(nChunks);
    parallelChunks(n, threads, [&](size_t ci, size_t from, size_t to) {
        Triples p; std::string e;      // worker-local, moved into the shared slots at the end
        size_t bytes = 0; for (size_t i = eFrom + from; i < eFrom + to; i++) bytes += res.entries[i].length;
        p.q.reserve(bytes / 12); p.t.reserve(bytes / 12); p.d.reserve(bytes / 12); p.perQuery.reserve(to - from + 1); p.entryQuery.reserve(to - from); p.entryKey.reserve(to - from);
        p.perQuery.push_back(0);
        for (size_t i = eFrom + from; i < eFrom + to && e.empty(); i++) {
            const uint32_t key = res.entries[i].key;
            const char *c = res.ptr(i);
            uint32_t qi = 0xFFFFFFFFu;
            if (*c != '\0' && !qIndex.find(key, qi)) { e = "query " + std::to_string(key) + " of the result DB is not in the query DB"; break; }
            while (*c != '\0') {
                char *end; unsigned long tk = strtoul(c, &end, 10);
                long score = 0, diag = 0; int cols = 1;
                if (*end == '\t') { score = strtol(end + 1, &end, 10); cols++; }
                if (*end == '\t') { diag = strtol(end + 1, &end, 10); cols++; }
                (void) score;
                uint32_t ti = 0;
                if (!tIndex.find((uint32_t) tk, ti)) { e = "target " + std::to_string(tk) + " of the result DB is not in the target DB"; break; }
                p.q.push_back(qi); p.t.push_back(ti); p.d.push_back(cols == 3 ? (int16_t) diag : 0);
                while (*end != '\n' && *end != '\0') end++;
                c = (*end == '\n') ? end + 1 : end;
            }
            p.perQuery.push_back(p.q.size()); p.entryQuery.push_back(qi); p.entryKey.push_back(key);
        }
        part[ci] = std::move(p); errs[ci] = std::move(e);
    });
    for (const std::string &e : errs) if (!e.empty()) { err = e; return false; }
    size_t total = 0; for (const Triples &p : part) total += p.q.size();
    tr.perQuery.assign(1, 0); tr.q.reserve(total); tr.t.reserve(total); tr.d.reserve(total); tr.perQuery.reserve(n + 1); tr.entryQuery.reserve(n); tr.entryKey.reserve(n);
    for (const Triples &p : part) {
        const uint64_t base = tr.q.size();
        tr.q.insert(tr.q.end(), p.q.begin(), p.q.end()); tr.t.insert(tr.t.end(), p.t.begin(), p.t.end()); tr.d.insert(tr.d.end(), p.d.begin(), p.d.end());
        for (size_t k = 1; k < p.perQuery.size(); k++) tr.perQuery.push_back(base + p.perQuery[k]);
        tr.entryQuery.insert(tr.entryQuery.end(), p.entryQuery.begin(), p.entryQuery.end()); tr.entryKey.insert(tr.entryKey.end(), p.entryKey.begin(), p.entryKey.end());
    }
    return true;
}

// (everything the host can do without the device comes FIRST -- reading and flattening both DBs, and `beforeContext`, the caller's own host work (align: reading and
// parsing the result DB): in a fresh process the runtime is still coming up on its own thread meanwhile, sp_init_async, and the context waits for it)
static bool openSearch(const Flags &f, const char *module, bool gapped, Ctx &ctx, Db &qdb, Db &tdb, std::vector<uint32_t> &qkeys, std::vector<uint32_t> &tkeys,
                       std::vector<char> &qres, std::vector<uint64_t> &qoff, std::string &err, const std::function<bool(std::string &)> &beforeContext = nullptr) {
    Laps laps(module);
    if (!loadSeqDb(f.positional[0], qdb, err) || !loadSeqDb(f.positional[1], tdb, err)) return false;
    laps.lap("  read sequence DBs");
    std::vector<char> tres; std::vector<uint64_t> toff;
    flatten(tdb, tres, toff, tkeys);
    flatten(qdb, qres, qoff, qkeys);
    laps.lap("  flatten targets, queries");
    if (beforeContext && !beforeContext(err)) return false;
    sp_search_params p; sp_default_params(&p);
    static std::string sub; sub = aaPart(f.str("--sub-mat", "VTML40.out"));
    p.sub_mat = sub.c_str();
    p.evalue = f.real("-e", 0.001);
    p.threads = (int) f.num("--threads", 0);
    if (gapped) { p.gap_open = (int) strtol(aaPart(f.str("--gap-open", "11")).c_str(), NULL, 10); p.gap_extend = (int) strtol(aaPart(f.str("--gap-extend", "1")).c_str(), NULL, 10); }
    ctx.c = sp_ctx_create(&p);
    if (!ctx.c) { err = sp_last_error(); return false; }
    laps.lap("  context");
    if (sp_ctx_set_targets_ex(ctx.c, tres.data(), toff.data(), tkeys.data(), tkeys.size(), 1 /* no k-mer index needed */) != SP_OK) { err = sp_last_error(); return false; }
    laps.lap("  targets -> device");
    if (sp_ctx_set_queries(ctx.c, qres.data(), qoff.data(), qkeys.size()) != SP_OK) { err = sp_last_error(); return false; }
    laps.lap("  queries -> device");
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// align <queryDB> <targetDB> <prefilterDB> <outDB>          (alignment/Alignment.cpp:255-549)
// ---------------------------------------------------------------------------------------------------------------
int runAlignModule(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kAlignFlags, 4, f, err)) return fail("align: " + err);
    if (!requireValue(f, "--alignment-mode", {"0", "1", "2", "3"}, err) || !requireValue(f, "--comp-bias-corr", {"1"}, err) || !requireValue(f, "--compressed", {"0"}, err) ||
        !requireValue(f, "--alt-ali", {"0"}, err) || !requireValue(f, "--realign", {"0"}, err) || !requireValue(f, "--wrapped-scoring", {"0"}, err) ||
        !requireValue(f, "--add-self-matches", {"0"}, err) || !requireValue(f, "--alignment-output-mode", {"0"}, err) || !requireValue(f, "--score-bias", {"0", "0.0", "0.000"}, err) ||
        !requireValue(f, "-c", {"0", "0.0", "0.000"}, err) || !requireValue(f, "--seq-id-mode", {"0"}, err))
        return fail("align: " + err);
    if (f.num("-a", 0) != 1) return fail("align: only -a 1 (backtrace => score, coverage and sequence identity) is implemented, which is what predictmatch passes");
    { const int rc = fanOut("align", argc, argv, f.positional[3]); if (rc >= 0) return rc; }
    Laps laps("align");
    Ctx ctx; Db qdb, tdb, rdb; std::vector<uint32_t> qkeys, tkeys; std::vector<char> qres; std::vector<uint64_t> qoff;
    Triples tr;
    uint64_t used = 0;
    const double minSeqId = f.real("--min-seq-id", 0.0); const long minAlnLen = f.num("--min-aln-len", 0);
    const long maxAccept = f.num("--max-accept", INT_MAX), maxRejected = f.num("--max-rejected", INT_MAX);
    // serving, and this is the result DB the last `prefilter` call wrote from a context that still holds the batch: align its hits where they are
    bool resident = false;
    if (g_memo.c && currentPart().count <= 1 && g_memo.result == resultIdentity(f.positional[2]) && g_memo.query == fileIdentity(f.positional[0]) && g_memo.target == fileIdentity(f.positional[1])) {
        const AlignParams ap = alignParamsOf(g_memo.c);
        resident = ap.subMat == aaPart(f.str("--sub-mat", "VTML40.out")) && ap.gapOpen == (int) strtol(aaPart(f.str("--gap-open", "11")).c_str(), NULL, 10) &&
                   ap.gapExtend == (int) strtol(aaPart(f.str("--gap-extend", "1")).c_str(), NULL, 10) && ap.evalue == f.real("-e", 0.001);
    }
    ResultBufsRef bufs(g_memo.c != nullptr || getenv("SP_CONTEXT_CACHE") != nullptr);      // (serving)
    TextBufsRef text(g_memo.c != nullptr || getenv("SP_CONTEXT_CACHE") != nullptr);
    if (resident && deviceTextEnabled() && maxAccept == INT_MAX && maxRejected == INT_MAX) {
        // ... and write the entries where the records are: aligned, accepted, sorted and formatted on the device, one block of text for the writer
        DbWriter w; w.path = f.positional[3]; w.dbtype = 5;
        const int via = deviceTextBlock(g_memo.c, true, minSeqId, minAlnLen, g_memo.qkeys, *text.b, w);
        if (via < 0) return fail(std::string("align: ") + sp_last_error());
        if (via) {
            laps.lap("align + entries on the device (resident hits)");
            if (!w.close(err)) return fail("align: " + err);
            laps.lap("write DB");
            return 0;
        }
    }
    if (resident) {
        sp_ctx *c = g_memo.c;
        const sp_hit_packed *hits = nullptr; const uint32_t *ho = nullptr; uint64_t nHits = 0;
        if (sp_ctx_hits_view(c, &hits, &ho, &nHits) != SP_OK) return fail(std::string("align: ") + sp_last_error());
        if (!bufs.b->ensure(nHits + 16, 24 * (nHits + 16))) return fail(std::string("align: ") + sp_last_error());
        int rc = sp_ctx_align_hits(c, bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used);
        if (rc == SP_ERR_OVERFLOW && used > bufs.b->cigCap) { if (!bufs.b->ensure(nHits + 16, used + 16)) return fail(std::string("align: ") + sp_last_error()); rc = sp_ctx_align_hits(c, bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used); }
        if (rc != SP_OK) return fail(std::string("align: ") + sp_last_error());
        // entries in the order the ordinary path takes them -- the result DB's data order (Db::load), which is the batch order the prefilter call
        // wrote -- with the records of query q at the offsets of its hits
        tr.entryKey = g_memo.qkeys;
        tr.perQuery.assign(ho, ho + g_memo.qkeys.size() + 1);
        laps.lap("align on the device (resident hits)");
    } else {
        if (!openSearch(f, "align", true, ctx, qdb, tdb, qkeys, tkeys, qres, qoff, err, [&](std::string &e) {
                Laps inner("align");
                if (!rdb.load(f.positional[2], e)) return false;
                KeyMap qIndex, tIndex; qIndex.build(qkeys); tIndex.build(tkeys);
                const bool ok = collectTriples(rdb, qIndex, tIndex, tr, e);
                inner.lap("  read + parse result DB");
                return ok;
            })) return fail("align: " + err);
        laps.lap("read DBs + result DB, context, device");
        if (!bufs.b->ensure(tr.q.size() + 16, 24 * (tr.q.size() + 16))) return fail(std::string("align: ") + sp_last_error());
        int rc = sp_ctx_align(ctx.c, tr.q.data(), tr.t.data(), tr.q.size(), bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used);
        if (rc == SP_ERR_OVERFLOW && used > bufs.b->cigCap) { if (!bufs.b->ensure(tr.q.size() + 16, used + 16)) return fail(std::string("align: ") + sp_last_error()); rc = sp_ctx_align(ctx.c, tr.q.data(), tr.t.data(), tr.q.size(), bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used); }
        if (rc != SP_OK) return fail(std::string("align: ") + sp_last_error());
        laps.lap("align on the device");
    }
    const sp_alignment *res = bufs.b->res; const char *cig = bufs.b->cig;
    DbWriter w; w.path = f.positional[3]; w.dbtype = 5;       // DBTYPE_ALIGNMENT_RES
    // entries are formatted by --threads workers over contiguous ranges and appended in order
    const size_t nEntries = tr.entryKey.size();
    const int threads = (int) f.num("--threads", 0) > 0 ? (int) f.num("--threads", 0) : defaultThreads();
    const size_t nChunks = numChunks(nEntries, threads);
    std::vector<std::string> blocks = DbWriter::pooledBlocks(nChunks);
        static thread_local std::vector<std::vector<uint32_t>> keyPool; static thread_local std::vector<std::vector<uint64_t>> lenPool;      // (kept with their capacity, like the blocks)
        std::vector<std::vector<uint32_t>> &bKeys = keyPool; std::vector<std::vector<uint64_t>> &bLens = lenPool;      // the CALLING thread's pools: the workers below see them through these references
        bKeys.resize(std::max(bKeys.size(), nChunks)); bLens.resize(std::max(bLens.size(), nChunks));
        for (size_t ci = 0; ci < nChunks; ci++) { bKeys[ci].clear(); bLens[ci].clear(); }
    parallelChunks(nEntries, threads, [&](size_t ci, size_t from, size_t to) {
        std::string block = std::move(blocks[ci]); std::vector<uint32_t> ck = std::move(bKeys[ci]); std::vector<uint64_t> cl = std::move(bLens[ci]);      // worker-local: no shared cache lines
        ck.reserve(to - from); cl.reserve(to - from);
        std::vector<uint64_t> ok;
        for (size_t e = from; e < to; e++) {
            ok.clear();
            long passed = 0, rejected = 0;
            for (uint64_t i = tr.perQuery[e]; i < tr.perQuery[e + 1] && passed < maxAccept && rejected < maxRejected; i++) {
                const sp_alignment &r = res[i];
                const bool good = r.passed && r.seq_id >= minSeqId && (long) r.aln_len >= minAlnLen;    // Alignment::checkCriteria, Alignment.cpp:555-574
                if (good) { ok.push_back(i); passed++; rejected = 0; } else rejected++;
            }
            if (ok.size() > 1) std::sort(ok.begin(), ok.end(), [&](uint64_t a, uint64_t b) { return alnLess(res[a], res[b]); });
            const size_t before = block.size();
            for (uint64_t i : ok) appendAlnLine(block, res[i], cig + res[i].cigar_offset, true);
            block.push_back('\0');
            ck.push_back(tr.entryKey[e]); cl.push_back(block.size() - before);
        }
        blocks[ci] = std::move(block); bKeys[ci] = std::move(ck); bLens[ci] = std::move(cl);
    });
    for (size_t ci = 0; ci < nChunks; ci++) w.addBlock(bKeys[ci], bLens[ci], std::move(blocks[ci]));
    laps.lap("format records");
    if (!w.close(err)) return fail("align: " + err);
    laps.lap("write DB");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// search <queryDB> <targetDB> <alignmentDB> <tmpDir>: prefilter + align in one process on one resident context
// (lib/mmseqs/src/workflow/Search.cpp:200-420 writes blastp.sh, which calls the two modules through $MMSEQS --
// lib/mmseqs/data/workflow/blastp.sh:28-45 -- with the prefilter DB on disk in between).  Same flags as the reference
// `search` (Parameters::searchworkflow), same alignment DB as running `prefilter` then `align`; the prefilter DB is never
// formatted, written, read back or parsed, the target set and the queries are staged once, and the hits go from the
// prefilter kernels to the aligner without leaving the device.
// ---------------------------------------------------------------------------------------------------------------
static const std::set<std::string> kSearchOnlyFlags = {
    "--add-orf-stop", "--allow-deletion", "--chain-alignments", "--contig-end-mode", "--contig-start-mode", "--cov", "--create-lookup", "--diff", "--disk-space-limit",
    "--e-profile", "--exhaustive-search", "--exhaustive-search-filter", "--filter-hits", "--filter-msa", "--force-reuse", "--forward-frames", "--headers-split-mode", "--id-offset",
    "--lca-search", "--mask-profile", "--max-gaps", "--max-length", "--max-seq-id", "--merge-query", "--min-length", "--num-iterations", "--orf-start-mode", "--qid", "--qsc",
    "--remove-tmp-files", "--rescore-mode", "--reverse-frames", "--search-type", "--sens-steps", "--sequence-overlap", "--sequence-split-mode", "--sort-results", "--start-sens",
    "--strand", "--translate", "--translation-table", "--use-all-table-starts", "--wg", "--reuse-latest"};

int runSearchModule(int argc, const char **argv) {
    Flags f; std::string err;
    std::set<std::string> allowed = kPrefilterFlags; allowed.insert(kAlignFlags.begin(), kAlignFlags.end()); allowed.insert(kSearchOnlyFlags.begin(), kSearchOnlyFlags.end());
    if (!parseFlags(argc, argv, allowed, 4, f, err)) return fail("search: " + err);
    if (!requireValue(f, "-k", {"6", "0"}, err) || !requireValue(f, "--alph-size", {"21"}, err) || !requireValue(f, "--comp-bias-corr", {"1"}, err) ||
        !requireValue(f, "--diag-score", {"1"}, err) || !requireValue(f, "--exact-kmer-matching", {"0"}, err) || !requireValue(f, "--spaced-kmer-mode", {"1"}, err) ||
        !requireValue(f, "--spaced-kmer-pattern", {"11011101", ""}, err) || !requireValue(f, "--mask-lower-case", {"0"}, err) || !requireValue(f, "--compressed", {"0"}, err) ||
        !requireValue(f, "--add-self-matches", {"0"}, err) || !requireValue(f, "--split", {"0", "1"}, err) || !requireValue(f, "-c", {"0", "0.0", "0.000"}, err) ||
        !requireValue(f, "--alignment-mode", {"0", "1", "2", "3"}, err) || !requireValue(f, "--alt-ali", {"0"}, err) || !requireValue(f, "--realign", {"0"}, err) ||
        !requireValue(f, "--wrapped-scoring", {"0"}, err) || !requireValue(f, "--alignment-output-mode", {"0"}, err) || !requireValue(f, "--score-bias", {"0", "0.0", "0.000"}, err) ||
        !requireValue(f, "--seq-id-mode", {"0"}, err) || !requireValue(f, "--num-iterations", {"1"}, err) || !requireValue(f, "--search-type", {"0", "1"}, err) ||
        !requireValue(f, "--exhaustive-search", {"0"}, err) || !requireValue(f, "--lca-search", {"0"}, err) || !requireValue(f, "--start-sens", {"4", "4.0", "4.000"}, err) ||
        !requireValue(f, "--sens-steps", {"1"}, err) || !requireValue(f, "--translate", {"0"}, err))
        return fail("search: " + err);
    if (f.num("-a", 0) != 1) return fail("search: only -a 1 (backtrace => score, coverage and sequence identity) is implemented, which is what predictmatch passes");
    { const int rc = fanOut("search", argc, argv, f.positional[2]); if (rc >= 0) return rc; }
    Laps laps("search");
    Db tdb;
    std::shared_ptr<const FlatDb> qf = flatSeqDb(f.positional[0], err);
    if (!qf) return fail("search: " + err);
    laps.lap("read query DB");
    sp_search_params p; sp_default_params(&p);
    std::string seed = aaPart(f.str("--seed-sub-mat", "VTML80.out")), sub = aaPart(f.str("--sub-mat", "VTML40.out"));
    p.seed_sub_mat = seed.c_str(); p.sub_mat = sub.c_str();
    p.sensitivity = (float) f.real("-s", 5.7);
    p.kmer_score = (int) f.num("--k-score", INT_MAX);
    p.max_seqs = (int) f.num("--max-seqs", 300);
    p.min_ungapped_score = (int) f.num("--min-ungapped-score", 15);
    p.mask = (int) f.num("--mask", 1);
    p.evalue = f.real("-e", 0.001);
    p.gap_open = (int) strtol(aaPart(f.str("--gap-open", "11")).c_str(), NULL, 10);
    p.gap_extend = (int) strtol(aaPart(f.str("--gap-extend", "1")).c_str(), NULL, 10);
    p.threads = (int) f.num("--threads", 0);
    // SP_GPUS=N SP_SHARD=targets: the N workers each index 1/N of the TARGET set (for sets beyond one GPU's HBM), match ALL queries against their shard, all-gather the
    // candidates (sp_ctx_allgather_candidates: RCCL over xGMI) and finish -- selection on the union, alignment, formatting -- their slice of the queries; the entries equal
    // the unsplit run's.  Default (no SP_SHARD): the queries are partitioned and every worker holds the whole target set -- no exchange at all.
    const Part myPart = currentPart();
    const char *shardEnv = getenv("SP_SHARD");
    const bool shardTargets = myPart.count > 1 && shardEnv && std::string(shardEnv) == "targets";
    struct CommHolder { sp_comm *c = nullptr; ~CommHolder() { if (c) sp_dist_destroy(c); } } comm;
    if (shardTargets) {
        const char *hexId = getenv("SP_DIST_ID");
        if (!hexId || strlen(hexId) != 2 * SP_DIST_ID_BYTES) return fail("search: SP_SHARD=targets needs the workers' SP_DIST_ID (set by the SP_GPUS fan-out)");
        unsigned char id[SP_DIST_ID_BYTES];
        for (int i = 0; i < SP_DIST_ID_BYTES; i++) { unsigned v = 0; sscanf(hexId + 2 * i, "%2x", &v); id[i] = (unsigned char) v; }
        const char *tr = getenv("SP_DIST_TRANSPORT");
        const int transport = tr ? (std::string(tr) == "host" ? SP_TRANSPORT_HOST : SP_TRANSPORT_RCCL) : (sp_device_count() >= myPart.count ? SP_TRANSPORT_RCCL : SP_TRANSPORT_HOST);
        comm.c = sp_dist_init(id, myPart.rank, myPart.count, transport);
        if (!comm.c) return fail(std::string("search: ") + sp_last_error());
    }
    std::vector<char> tres; std::vector<uint64_t> toff; std::vector<uint32_t> tkeys;
    bool targetsRead = false;
    if (oneShotProcess() && !shardTargets) {      // (a fresh process: the runtime is still coming up on its own thread -- sp_init_async -- and the context below waits for it: the target DB is read and flattened meanwhile)
        if (!loadSeqDb(f.positional[1], tdb, err)) return fail("search: " + err);
        flatten(tdb, tres, toff, tkeys);
        targetsRead = true;
        laps.lap("read + flatten target DB");
    }
    Ctx ctx;
    if (shardTargets) { ctx.c = sp_ctx_create(&p); if (!ctx.c) return fail(std::string("search: ") + sp_last_error()); }      // (a shard is not a cacheable target set)
    else if (!acquireCtx(p, f.positional[1], "index", ctx, err)) return fail("search: " + err);
    laps.lap("context (matrices, tables)");
    const std::vector<char> &qres = qf->res; const std::vector<uint64_t> &qoff = qf->off; const std::vector<uint32_t> &qkeys = qf->keys;
    if (shardTargets) {
        if (!loadSeqDb(f.positional[1], tdb, err)) return fail("search: " + err);
        flatten(tdb, tres, toff, tkeys);
        std::vector<uint64_t> lens(tkeys.size()); for (size_t i = 0; i < lens.size(); i++) lens[i] = toff[i + 1] - toff[i] + 2;
        uint64_t tFrom = 0, tCnt = 0;
        if (sp_dist_decompose(lens.data(), lens.size(), myPart.rank, myPart.count, &tFrom, &tCnt) != SP_OK ||
            sp_ctx_set_targets_shard(ctx.c, tres.data(), toff.data(), tkeys.data(), tkeys.size(), tFrom, tFrom + tCnt) != SP_OK) return fail(std::string("search: ") + sp_last_error());
    } else if (!ctx.targetsStaged) {      // (a cached context -- the resident server -- holds the target set of this very DB: nothing to read)
        if (!targetsRead) {
            if (!loadSeqDb(f.positional[1], tdb, err)) { dropCtx(ctx); return fail("search: " + err); }
            flatten(tdb, tres, toff, tkeys);
        }
        if (sp_ctx_set_targets(ctx.c, tres.data(), toff.data(), tkeys.data(), tkeys.size()) != SP_OK) { const std::string e = sp_last_error(); dropCtx(ctx); return fail("search: " + e); }
    }
    laps.lap("targets -> device, index");
    laps.memory("after the target set");
    const double minSeqId = f.real("--min-seq-id", 0.0); const long minAlnLen = f.num("--min-aln-len", 0);
    const long maxAccept = f.num("--max-accept", INT_MAX), maxRejected = f.num("--max-rejected", INT_MAX);
    const int threads = (int) f.num("--threads", 0) > 0 ? (int) f.num("--threads", 0) : defaultThreads();
    DbWriter w; w.path = f.positional[2]; w.dbtype = 5;       // DBTYPE_ALIGNMENT_RES
    const size_t batch = 1u << 20;
    size_t qFrom = 0, nq = qkeys.size();
    if (!shardTargets) { std::vector<uint64_t> lens(qkeys.size()); for (size_t i = 0; i < lens.size(); i++) lens[i] = qoff[i + 1] - qoff[i] + 2; partRange(lens, w.part, qFrom, nq); }   // this GPU's queries
    ResultBufsRef bufs(ctx.cached);
    TextBufsRef text(ctx.cached);
    auto freeBufs = [&]() { };      // (the buffers go with `bufs`)
    // align the hits of the context's current batch, filter / sort / format them, one entry per key
    auto alignAndFormat = [&](sp_ctx *c, const std::vector<uint32_t> &keys, size_t entryFrom = 0) -> bool {      // keys[k] names query entryFrom + k of the resident batch
        const uint64_t nHits = sp_ctx_num_hits(c);
        if (!bufs.b->ensure(nHits + 16, 24 * (nHits + 16))) return false;
        uint64_t used = 0;
        int rc = sp_ctx_align_hits(c, bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used);
        if (rc == SP_ERR_OVERFLOW && used > bufs.b->cigCap) {
            if (!bufs.b->ensure(nHits + 16, used + 1024)) return false;
            rc = sp_ctx_align_hits(c, bufs.b->res, bufs.b->cig, bufs.b->cigCap, &used);
        }
        if (rc != SP_OK) return false;
        const sp_alignment *res = bufs.b->res; const char *cig = bufs.b->cig;
        const sp_hit_packed *hits = nullptr; const uint32_t *ho = nullptr; uint64_t nh = 0;
        if (sp_ctx_hits_view(c, &hits, &ho, &nh) != SP_OK) return false;
        laps.lap("align on the device");
        const size_t nEntries = keys.size();
        const size_t nChunks = numChunks(nEntries, threads);
        std::vector<std::string> blocks = DbWriter::pooledBlocks(nChunks);
        static thread_local std::vector<std::vector<uint32_t>> keyPool; static thread_local std::vector<std::vector<uint64_t>> lenPool;      // (kept with their capacity, like the blocks)
        std::vector<std::vector<uint32_t>> &bKeys = keyPool; std::vector<std::vector<uint64_t>> &bLens = lenPool;      // the CALLING thread's pools: the workers below see them through these references
        bKeys.resize(std::max(bKeys.size(), nChunks)); bLens.resize(std::max(bLens.size(), nChunks));
        for (size_t ci = 0; ci < nChunks; ci++) { bKeys[ci].clear(); bLens[ci].clear(); }
        parallelChunks(nEntries, threads, [&](size_t ci, size_t from, size_t to) {
            std::string block = std::move(blocks[ci]); std::vector<uint32_t> ck = std::move(bKeys[ci]); std::vector<uint64_t> cl = std::move(bLens[ci]);      // worker-local: no shared cache lines
            from += entryFrom; to += entryFrom;
            block.reserve((size_t) ((to - from) * 40 + (ho[to] - ho[from]) * 64));
            ck.reserve(to - from); cl.reserve(to - from);
            std::vector<uint64_t> ok;
            for (size_t e = from; e < to; e++) {
                ok.clear();
                long passed = 0, rejected = 0;
                for (uint64_t i = ho[e]; i < ho[e + 1] && passed < maxAccept && rejected < maxRejected; i++) {
                    const sp_alignment &r = res[i];
                    const bool good = r.passed && r.seq_id >= minSeqId && (long) r.aln_len >= minAlnLen;    // Alignment::checkCriteria, Alignment.cpp:555-574
                    if (good) { ok.push_back(i); passed++; rejected = 0; } else rejected++;
                }
                if (ok.size() > 1) std::sort(ok.begin(), ok.end(), [&](uint64_t a, uint64_t b) { return alnLess(res[a], res[b]); });
                const size_t before = block.size();
                for (uint64_t i : ok) appendAlnLine(block, res[i], cig + res[i].cigar_offset, true);
                block.push_back('\0');
                ck.push_back(keys[e - entryFrom]); cl.push_back(block.size() - before);
            }
            blocks[ci] = std::move(block); bKeys[ci] = std::move(ck); bLens[ci] = std::move(cl);
        });
        for (size_t ci = 0; ci < nChunks; ci++) w.addBlock(bKeys[ci], bLens[ci], std::move(blocks[ci]));
        laps.lap("format records");
        return true;
    };
    for (size_t b0 = qFrom; b0 < nq; b0 += batch) {
        const size_t b1 = std::min(nq, b0 + batch);
        std::vector<uint64_t> off(qoff.begin() + b0, qoff.begin() + b1 + 1);
        for (uint64_t &o : off) o -= qoff[b0];
        std::vector<uint32_t> keys(qkeys.begin() + b0, qkeys.begin() + b1);
        if (sp_ctx_set_queries(ctx.c, qres.data() + qoff[b0], off.data(), b1 - b0) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
        if (shardTargets) {
            // every worker matched the whole batch against its shard; this worker finishes slice `rank` of the batch
            std::vector<uint64_t> lens(b1 - b0); for (size_t i = 0; i < lens.size(); i++) lens[i] = off[i + 1] - off[i] + 2;
            uint64_t sFrom = 0, sCnt = 0, rows = 0; double secs = 0;
            const int rcC = sp_ctx_prefilter_candidates(ctx.c);
            // A query whose hit stream does not fit one pass over THIS rank's shard (QueryMatcher.cpp:254-283 is the reference's overflow path): the ranks agree on the union of
            // such queries, search the batch without them as usual, and each refused query goes -- on one rank, round robin -- through sub-shards of the WHOLE target set
            // (searchByShards: every rank has read the target DB), whose result equals the unsplit search.  Round 6; before, one rank's overflow ended the job on all ranks.
            std::vector<uint32_t> refusedHere(rcC == SP_ERR_OVERFLOW ? sp_ctx_overflow_queries(ctx.c, nullptr, 0) : 0);
            if (!refusedHere.empty()) sp_ctx_overflow_queries(ctx.c, refusedHere.data(), refusedHere.size());
            const std::string candErr = rcC == SP_OK ? std::string() : std::string(sp_last_error());
            std::vector<uint64_t> counts((size_t) myPart.count, 0);
            uint64_t mine = (rcC != SP_OK && refusedHere.empty()) ? ~0ull : (uint64_t) refusedHere.size();      // (~0: this rank failed for another reason)
            if (sp_dist_allgather(comm.c, &mine, sizeof(mine), counts.data(), 0) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
            uint64_t most = 0; bool someoneFailed = false;
            for (uint64_t n : counts) { if (n == ~0ull) someoneFailed = true; else most = std::max(most, n); }
            if (someoneFailed) { freeBufs(); return fail("search: " + (candErr.empty() ? std::string("the candidate pass failed on another rank") : candErr)); }
            std::vector<uint32_t> refused;
            if (most) {
                std::vector<uint32_t> padded((size_t) most, 0xFFFFFFFFu), all((size_t) most * (size_t) myPart.count);
                std::copy(refusedHere.begin(), refusedHere.end(), padded.begin());
                if (sp_dist_allgather(comm.c, padded.data(), most * sizeof(uint32_t), all.data(), 0) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
                for (uint32_t q : all) if (q != 0xFFFFFFFFu) refused.push_back(q);
                std::sort(refused.begin(), refused.end()); refused.erase(std::unique(refused.begin(), refused.end()), refused.end());
            }
            if (refused.empty()) {
                if (sp_dist_decompose(lens.data(), lens.size(), myPart.rank, myPart.count, &sFrom, &sCnt) != SP_OK ||
                    sp_ctx_allgather_candidates(ctx.c, comm.c, sFrom, sFrom + sCnt, &rows, &secs) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
                laps.lap("queries -> device, candidates, all-gather, selection");
                std::vector<uint32_t> sliceKeys(qkeys.begin() + b0 + sFrom, qkeys.begin() + b0 + sFrom + sCnt);
                if (!alignAndFormat(ctx.c, sliceKeys, (size_t) sFrom)) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
                continue;
            }
            std::vector<char> res2; std::vector<uint64_t> off2; std::vector<uint32_t> kept;
            withoutQueries(qres.data() + qoff[b0], off.data(), b1 - b0, refused, res2, off2, kept);
            std::vector<uint64_t> lens2(kept.size()); for (size_t i = 0; i < kept.size(); i++) lens2[i] = off2[i + 1] - off2[i] + 2;
            if (!kept.empty()) {      // (the same decision on every rank: `kept` follows from the agreed union)
                if (sp_ctx_set_queries(ctx.c, res2.data(), off2.data(), kept.size()) != SP_OK || sp_dist_decompose(lens2.data(), lens2.size(), myPart.rank, myPart.count, &sFrom, &sCnt) != SP_OK ||
                    sp_ctx_prefilter_candidates(ctx.c) != SP_OK || sp_ctx_allgather_candidates(ctx.c, comm.c, sFrom, sFrom + sCnt, &rows, &secs) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
                laps.lap("queries -> device, candidates, all-gather, selection (without the refused queries)");
                std::vector<uint32_t> sliceKeys; for (uint64_t i = sFrom; i < sFrom + sCnt; i++) sliceKeys.push_back(qkeys[b0 + kept[i]]);
                if (!alignAndFormat(ctx.c, sliceKeys, (size_t) sFrom)) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
            }
            std::vector<uint32_t> mineRefused;
            for (size_t i = 0; i < refused.size(); i++) if ((int) (i % (size_t) myPart.count) == myPart.rank) mineRefused.push_back(refused[i]);
            if (!mineRefused.empty()) {
                sp_ctx *r = searchByShards(p, tres, toff, tkeys, qres.data() + qoff[b0], off.data(), mineRefused, err);
                if (!r) { freeBufs(); return fail("search: " + err); }
                std::vector<uint32_t> rk; for (uint32_t q : mineRefused) rk.push_back(qkeys[b0 + q]);
                const bool ok = alignAndFormat(r, rk);
                const std::string e = ok ? std::string() : std::string(sp_last_error());
                sp_ctx_destroy(r);
                if (!ok) { freeBufs(); return fail("search: " + e); }
                laps.lap("refused queries through sub-shards of the whole target set");
            }
            continue;
        }
        const int rc = sp_ctx_prefilter(ctx.c);
        std::vector<uint32_t> refused(rc == SP_ERR_OVERFLOW ? sp_ctx_overflow_queries(ctx.c, nullptr, 0) : 0);
        if (rc == SP_ERR_OVERFLOW && !refused.empty()) {      // the batch without the queries one pass cannot hold, then those against target shards (searchByShards)
            sp_ctx_overflow_queries(ctx.c, refused.data(), refused.size());
            std::vector<char> res2; std::vector<uint64_t> off2; std::vector<uint32_t> kept;
            withoutQueries(qres.data() + qoff[b0], off.data(), b1 - b0, refused, res2, off2, kept);
            if (sp_ctx_set_queries(ctx.c, res2.data(), off2.data(), kept.size()) != SP_OK || sp_ctx_prefilter(ctx.c) != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
            keys.clear(); for (uint32_t q : kept) keys.push_back(qkeys[b0 + q]);
            if (!alignAndFormat(ctx.c, keys)) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
            if (tkeys.empty()) {      // (a cached context: the target DB was not read for staging)
                if (!loadSeqDb(f.positional[1], tdb, err)) { freeBufs(); return fail("search: " + err); }
                flatten(tdb, tres, toff, tkeys);
            }
            sp_ctx *r = searchByShards(p, tres, toff, tkeys, qres.data() + qoff[b0], off.data(), refused, err);
            if (!r) { freeBufs(); return fail("search: " + err); }
            keys.clear(); for (uint32_t q : refused) keys.push_back(qkeys[b0 + q]);
            const bool ok = alignAndFormat(r, keys);
            const std::string e = ok ? std::string() : std::string(sp_last_error());
            sp_ctx_destroy(r);
            if (!ok) { freeBufs(); return fail("search: " + e); }
            continue;
        }
        if (rc != SP_OK) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
        laps.lap("queries -> device, prefilter");
        int viaDevice = 0;
        if (b0 == qFrom && b1 == nq && deviceTextEnabled() && maxAccept == INT_MAX && maxRejected == INT_MAX) {      // the whole run is this batch: aligned, accepted, sorted, formatted on the device
            viaDevice = deviceTextBlock(ctx.c, true, minSeqId, minAlnLen, keys, *text.b, w);
            if (viaDevice < 0) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
            if (viaDevice) laps.lap("align + entries on the device");
        }
        if (!viaDevice && !alignAndFormat(ctx.c, keys)) { freeBufs(); return fail(std::string("search: ") + sp_last_error()); }
    }
    freeBufs();
    if (!w.close(err)) return fail("search: " + err);
    laps.lap("write DB");
    laps.memory("at the end of the call");
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// rescorediagonal <queryDB> <targetDB> <prefilterDB> <outDB>     (alignment/rescorediagonal.cpp:45-379), modes 1 and 2
// ---------------------------------------------------------------------------------------------------------------
int runRescoreDiagonalModule(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kRescoreFlags, 4, f, err)) return fail("rescorediagonal: " + err);
    // (--filter-hits 1 reads its score-per-column threshold from a table the reference generates at build time -- CovSeqidQscPercMinDiag.lib.h, rescorediagonal.cpp:9-10,98-103 --
    // which is not in its source tree: refused, like wrapped scoring, which is for nucleotide DBs)
    if (!requireValue(f, "--rescore-mode", {"0", "1", "2", "3", "4"}, err) || !requireValue(f, "--wrapped-scoring", {"0"}, err) || !requireValue(f, "--filter-hits", {"0"}, err) ||
        !requireValue(f, "--compressed", {"0"}, err) || !requireValue(f, "--add-self-matches", {"0"}, err) || !requireValue(f, "-c", {"0", "0.0", "0.000"}, err) ||
        !requireValue(f, "--seq-id-mode", {"0"}, err))
        return fail("rescorediagonal: " + err);
    const int mode = (int) f.num("--rescore-mode", 0);      // RESCORE_MODE_HAMMING is the module's default (commons/Parameters.cpp:2256)
    const bool alnFormat = mode >= 2;                        // modes 2, 3, 4 write alignment records (rescorediagonal.cpp:390-394), 0 and 1 prefilter-format lines
    { const int rc = fanOut("rescorediagonal", argc, argv, f.positional[3]); if (rc >= 0) return rc; }
    Ctx ctx; Db qdb, tdb, rdb; std::vector<uint32_t> qkeys, tkeys; std::vector<char> qres; std::vector<uint64_t> qoff;
    if (!openSearch(f, "rescorediagonal", false, ctx, qdb, tdb, qkeys, tkeys, qres, qoff, err)) return fail("rescorediagonal: " + err);
    if (!rdb.load(f.positional[2], err)) return fail("rescorediagonal: " + err);
    KeyMap qIndex, tIndex; qIndex.build(qkeys); tIndex.build(tkeys);
    Triples tr;
    if (!collectTriples(rdb, qIndex, tIndex, tr, err)) return fail("rescorediagonal: " + err);
    std::vector<sp_rescore> res(std::max<size_t>(1, tr.q.size()));
    if (sp_ctx_rescore(ctx.c, tr.q.data(), tr.t.data(), tr.d.data(), tr.q.size(), mode, res.data()) != SP_OK) return fail(std::string("rescorediagonal: ") + sp_last_error());
    const double evalThr = f.real("-e", 0.001), seqIdThr = f.real("--min-seq-id", 0.0); const long alnLenThr = f.num("--min-aln-len", 0);
    const bool addBt = f.num("-a", 0) == 1; const bool sortRes = f.num("--sort-results", 0) > 0;
    // target ASCII for the identity count of mode 2 (rescorediagonal.cpp:283-290 compares letters case-insensitively)
    std::vector<char> tres; std::vector<uint64_t> toff; std::vector<uint32_t> tk2; flatten(tdb, tres, toff, tk2);
    DbWriter w; w.path = f.positional[3]; w.dbtype = alnFormat ? 5 : (rdb.type() > 0 ? rdb.type() : 7);
    std::string payload; char buf[64];
    for (size_t e = 0; e + 1 < tr.perQuery.size(); e++) {
        std::vector<std::pair<sp_alignment, std::string>> alns;
        struct Short { uint32_t key; int score; int diag; }; std::vector<Short> shorts;
        for (uint64_t i = tr.perQuery[e]; i < tr.perQuery[e + 1]; i++) {
            const sp_rescore &r = res[i];
            const uint32_t qi = tr.q[i], ti = tr.t[i];
            const int qLen = (int) (qoff[qi + 1] - qoff[qi]), dbLen = (int) (toff[ti + 1] - toff[ti]);
            double seqId = 0; int alnLen = 0;
            // mode 0 has no E-value (it stays 0.0 and always passes -e >= 0): the score is the identity count, the aligned length the diagonal's (rescorediagonal.cpp:243-252)
            const bool hasEvalue = (mode == 0 ? 0.0 : r.evalue) <= evalThr;
            sp_alignment a; memset(&a, 0, sizeof(a));
            std::string bt;
            if (mode == 0) { alnLen = r.diagonal_len; seqId = (double) ((float) r.score / (float) r.diagonal_len); }
            if (alnFormat) {
                alnLen = (r.end_pos - r.start_pos) + 1;
                int qs, qe, ts, te;
                if (r.diagonal >= 0) { qs = r.start_pos + (int) r.dist_to_diagonal; qe = r.end_pos + (int) r.dist_to_diagonal; ts = r.start_pos; te = r.end_pos; }
                else { qs = r.start_pos; qe = r.end_pos; ts = r.start_pos + (int) r.dist_to_diagonal; te = r.end_pos + (int) r.dist_to_diagonal; }
                if (hasEvalue) {
                    int ids = 0;
                    for (int x = qs; x <= qe; x++) { char ql = qres[qoff[qi] + x] & (char) ~0x20, tl = tres[toff[ti] + ts + (x - qs)] & (char) ~0x20; ids += (ql == tl); }
                    seqId = (double) ((float) ids / (float) alnLen);
                }
                if (addBt) { char *e2 = i32toa(alnLen, buf); bt.assign(buf, e2 - buf); bt.push_back('M'); }
                a.db_key = tkeys[ti]; a.bit_score = r.bit_score; a.seq_id = (float) seqId; a.evalue = r.evalue; a.q_start = qs; a.q_end = qe; a.q_len = (uint32_t) qLen;
                a.db_start = ts; a.db_end = te; a.db_len = (uint32_t) dbLen; a.aln_len = (uint32_t) alnLen;
            }
            const bool hasSeqId = seqId >= (seqIdThr - 1.1920929e-07), hasAlnLen = alnLen >= alnLenThr;
            if (!(hasAlnLen && hasSeqId && hasEvalue)) continue;
            if (alnFormat) alns.push_back({a, bt});
            else shorts.push_back({tkeys[ti], mode == 0 ? (int) (100 * seqId) : r.bit_score, r.diagonal});      // (hit.prefScore = 100 * seqId: a double truncated to int, :357-358)
        }
        payload.clear();
        if (alnFormat) {
            if (sortRes) std::sort(alns.begin(), alns.end(), [](const std::pair<sp_alignment, std::string> &x, const std::pair<sp_alignment, std::string> &y) { return alnLess(x.first, y.first); });
            for (auto &x : alns) {
                // the reference prints the run-length string itself here (resultToBuffer(..., compress=false))
                std::string line; appendAlnLine(line, x.first, "", false);
                if (addBt) { line.pop_back(); line.push_back('\t'); line += x.second; line.push_back('\n'); }
                payload += line;
            }
        } else {
            if (sortRes) std::sort(shorts.begin(), shorts.end(), [](const Short &x, const Short &y) { int ax = abs(x.score), ay = abs(y.score); if (ax != ay) return ax > ay; return x.key < y.key; });
            for (const Short &s : shorts) {
                char *p = u32toa(s.key, buf); *p++ = '\t'; p = i32toa(s.score, p); *p++ = '\t'; p = i32toa((int32_t) (short) (uint16_t) s.diag, p); *p++ = '\n';
                payload.append(buf, p - buf);
            }
        }
        w.add(tr.entryKey[e], payload);
    }
    if (!w.close(err)) return fail("rescorediagonal: " + err);
    return 0;
}

}  // namespace sp
