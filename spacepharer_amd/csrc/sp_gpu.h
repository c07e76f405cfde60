// Device side of the search path: HBM-resident tables + the HIP kernels (sp_gpu.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "sp_host.h"

namespace sp {

struct GpuHit { uint32_t seqId; uint16_t score; uint16_t diag; };
// target-sharded search: best double-hit diagonal of one (query, target) as exchanged between shards.  ord = ordinal of
// the similar k-mer whose index hit recorded the diagonal, counted over the query's k-mer enumeration (shard independent).
struct Candidate { uint32_t query; uint32_t seqId; uint32_t ord; uint16_t diag; uint16_t exact; };

struct SwPair { uint32_t query; uint32_t target; };          // indices into the resident query batch / target set
struct SwForward { int32_t score; int32_t refEnd; int32_t readEnd; int32_t word; };
struct SwFull {
    int32_t qStart, dbStart;      // from the reverse pass
    int32_t alnLen;               // length of the backtrace
    int32_t identities;
    uint32_t btOffset;            // into the backtrace byte pool
    int32_t status;               // 0 ok
};
// E-value gate as a lookup table filled by the host with the exact double-precision test: pass[slotOf[qLen] * stride + min(score, stride-1)]
// evalue[slot * stride + score] and bits[score] are the statistics the result records carry (doubles computed by the host's libm);
// version changes whenever the content does, so that the device copy is refreshed only then
struct GateTable { uint32_t stride = 0; std::vector<uint16_t> slotOf; std::vector<uint8_t> pass; std::vector<double> evalue; std::vector<int32_t> bits;
                   std::vector<unsigned char> evalText;      // "%.3E" of every table entry, 16 bytes each (length, characters): the device formats result lines from it; empty when the table would be large
                   double evalueThr = 0; uint64_t version = 0; };
// device-assembled result record: layout of sp_alignment (include/spacepharer_amd.h), checked by a static_assert in sp_api.cpp
struct AlnRec { uint32_t dbKey; int32_t bitScore, rawScore; float seqId; double evalue; int32_t qStart, qEnd; uint32_t qLen; int32_t dbStart, dbEnd; uint32_t dbLen, alnLen, cigarOffset; int32_t passed; };
// where align() may put finished records directly (host memory of the caller, ideally page-locked: sp_host_alloc)
// async: return as soon as the copies are queued; the caller collects them with Gpu::waitAlignCopy()
struct AlignDirect { void *records = nullptr; char *cigars = nullptr; uint64_t cigarCap = 0; uint64_t cigarUsed = 0; bool done = false; bool overflow = false; bool async = false;
                     bool deviceOnly = false; };      // deviceOnly: assemble records + packed backtraces on the device and leave them there (the text kernels read them)
// one device-driven search step (Gpu::stepBegin / stepEnd): where its results go (caller-owned, page-locked: sp_host_alloc) ...
struct StepOut { GpuHit *hits = nullptr; uint32_t *offsets = nullptr; uint64_t hitCap = 0; AlnRec *records = nullptr; char *cigars = nullptr; uint64_t recordCap = 0, cigarCap = 0; };
// ... and what the host learns when it collects the step
struct StepResult { uint64_t hits = 0, pairs = 0, cigarUsed = 0; bool repeat = false; std::string why; };
struct DiagRescore { int32_t score, startPos, endPos, diagonalLen, diagonal; uint32_t distToDiagonal; };

// page-locked host memory for device<->host copies (pageable std::vector memory costs a staging copy per transfer)
bool deviceMemory(size_t &freeBytes, size_t &totalBytes);      // hipMemGetInfo of the current device
void *pinnedAlloc(size_t bytes, bool coherent = false);      // coherent: kernels store into it while the host polls (mailboxes); else DMA destinations the CPU reads cached
void pinnedFree(void *p);
template <typename T>
struct PinnedAlloc {
    using value_type = T;
    PinnedAlloc() = default;
    template <typename U> PinnedAlloc(const PinnedAlloc<U> &) {}
    T *allocate(size_t n) { return static_cast<T *>(pinnedAlloc(n * sizeof(T))); }
    void deallocate(T *p, size_t) { pinnedFree(p); }
    template <typename U> bool operator==(const PinnedAlloc<U> &) const { return true; }
    template <typename U> bool operator!=(const PinnedAlloc<U> &) const { return false; }
};
template <typename T> using PinVec = std::vector<T, PinnedAlloc<T>>;

// host side of one align() call; the buffers are reused from call to call
struct AlignHost {
    PinVec<SwPair> pairs; PinVec<SwForward> fwd; PinVec<SwFull> full; PinVec<char> bt;
    std::vector<uint8_t> pass;
};

struct PrefilterCounters {       // the K, M, D, H terms of the roofline formula
    uint64_t kmers;              // similar k-mers generated (scanned windows only)
    uint64_t dbMatches;          // index entries fetched
    uint64_t diagonals;          // double-hit diagonals scored
    uint64_t diagResidues;       // target residues read while scoring
    uint64_t hits;               // hits written
    uint64_t overflowQueries;    // queries whose hit stream exceeded the supported size
};

struct KernelTimes {             // HIP-event time of the last prefilter()/align() call, milliseconds
    float prep = 0, match = 0, score = 0, select = 0, swForward = 0, swReverse = 0, traceback = 0, assemble = 0;
    float matchActive = 0;       // k_match from the start of its first wave to the end of its last (the kernel's own wall clock): what a profiler reports as its duration;
                                 // the event time above also holds the wait for the first wave slots when other contexts' kernels fill the CUs
};

class Gpu {
public:
    Gpu();
    ~Gpu();
    bool init(int device, std::string &err);
    static bool warmUp(int device, std::string &err);      // runtime + device context + code object, nothing of a search context (sp_init_async)
    int deviceCount() const;
    const std::string &deviceName() const { return name_; }

    // matrices + the similar-3-mer table (built on the device)
    bool uploadTables(const SubMat &seed, const SubMat &ungapped, const SubMat &aln, bool kmerTables, std::string &err);
    bool downloadExt3(std::vector<uint32_t> &packed, std::string &err);     // test hook
    // the product path: residues -> classes, tantan masking and the k-mer index of targets [from, to) built on the device
    // (sp_gpu_index.inc).  ts carries n / offsets / keys; plain, masked and ascii stay empty on the host.
    struct TargetBuild { uint64_t maskedResidues = 0, indexEntries = 0, indexSlots = 0; float msMask = 0, msIndex = 0; };
    bool buildTargets(const char *ascii, const TargetSet &ts, const SubMat &seed, int kmerThr, bool mask, bool index, size_t from, size_t to,
                      TargetBuild &out, std::string &err);
    // test hook: (k-mer, seqId, pos) triples of the resident index in table order, and the masked residue classes
    bool downloadIndex(std::vector<uint32_t> &kmers, std::vector<uint64_t> &entries, std::string &err);
    bool downloadMasked(std::vector<uint8_t> &masked, std::string &err);
    // use the tables and the target set another context of the same device has staged (shared, not copied)
    bool shareFrom(const Gpu &other, std::string &err);
    // queries: numeric residues (seed-matrix numbering) concatenated; off has nq+1 entries
    bool uploadQueries(const uint8_t *res, const uint32_t *off, size_t nq, std::string &err);
    size_t numQueries() const { return nq_; }

    // k-mer prefilter + ungapped diagonal scoring + selection for every resident query
    bool prefilter(int kmerThr, unsigned binSize, int maxHits, int minDiagScore, std::string &err);
    // target-sharded search (every GPU indexes a target range, all queries resident): candidates of this shard, then the
    // selection over the union of all shards' candidates (gathered by the caller, e.g. RCCL all-gather)
    bool prefilterCandidates(int kmerThr, int minDiagScore, std::string &err);
    // ... its probe-list cut (sp_gpu.hip: "the probe-list cut"): this shard's occupancy map; all shards' maps; enumerate a query slice -> probes by destination shard;
    // fetch the probes addressed to this shard -> candidates; the four steps over a communicator
    bool shardOccupancy(void *out, bool onDevice, std::string &err);
    bool probeSetShardMaps(const void *maps, int nShards, bool onDevice, std::string &err);
    bool probeEnumerate(int kmerThr, size_t qFrom, size_t qTo, uint64_t *counts, std::string &err);
    size_t probeCount(int dest) const;
    size_t probeSliceQueries() const;
    const void *probeRecordsDevice(int dest) const;               // 16-byte records {query, position in the query's list, ordinal, slot rank | window position << 26}
    const uint32_t *probeQueryCountsDevice(int dest) const;       // probes per query of the enumerated slice
    bool probeFetch(const void *records, size_t n, const uint32_t *counts, bool onDevice, int kmerThr, int minDiagScore, std::string &err);
    bool probeSearch(class Comm &comm, int kmerThr, int minDiagScore, size_t qFrom, size_t qTo, double *exchangeMs, std::string &err);
    size_t numCandidates() const;
    const std::vector<uint32_t> &overflowQueries() const { return overflowList_; }      // positions (in the batch) of the queries the last pass refused
    const Candidate *candidatesDevice() const;
    bool downloadCandidates(Candidate *out, std::string &err);
    // multi-GPU exchange on device buffers (sp_dist.h): the shards' candidates -> candidateUnionDevice() on every rank; the slices' hit lists -> host arrays in query order
    bool allgatherCandidates(class Comm &comm, size_t &unionRows, std::string &err);
    const Candidate *candidateUnionDevice() const;
    bool allgatherHits(class Comm &comm, size_t queryOffset, size_t nqTotal, const GpuHit *srcHits, const uint32_t *srcOffsets, GpuHit *hits, uint64_t *offsets, size_t hitCap, size_t &nHits, std::string &err);
    bool mergeCandidates(const Candidate *src, size_t n, bool onDevice, size_t qFrom, size_t qTo, unsigned binSize, int maxHits, int minDiagScore, std::string &err);
    // hits grouped by query (query order); offsets has nq+1 entries
    bool hostHits(const GpuHit *&hits, const uint32_t *&offsets, size_t &n, std::string &err);
    // deliver the hit lists of the following passes into caller-owned (page-locked) memory: hits[capHits], offsets[nq + 1];
    // a pass with more hits than capHits falls back to the internal staging buffers.  nullptr switches back.
    void setHitSink(GpuHit *hits, uint32_t *offsets, size_t capHits);
    PrefilterCounters counters() const { return counters_; }
    KernelTimes times() const { return times_; }
    int maxHits() const { return maxHits_; }

    // retry path of align(): banded traceback with host-computed scratch offsets for the flagged pairs
    bool traceHostDriven(const uint8_t *passMask, std::vector<SwFull> &out, std::vector<char> &backtraces, std::string &err);

    // device-driven align: forward pass, E-value gate (table), 16-bit redo, reverse pass, banded traceback without host
    // round trips.  hostPairs == nullptr aligns every hit of the last prefilter(), in the order of downloadHits().
    // direct != nullptr: result records and backtraces are assembled on the device and copied straight into the caller's
    // buffers (direct->done); when a pair needs the host-driven retry path, `out` is filled instead (direct->done stays false)
    bool align(const SwPair *hostPairs, size_t n, int gapOpen, int gapExtend, const GateTable &gate, AlignHost &out, AlignDirect *direct, std::string &err);
    bool waitAlignCopy(std::string &err);
    // One whole search step of the resident queries -- prefilter, selection, align of every hit, record assembly, delivery of the hit list, the
    // records and the packed backtraces into `out` -- queued WITHOUT a host round trip: the hit count stays on the device (kernels sized by a
    // capacity read it themselves), results and counters travel on the copy stream.  stepBegin returns as soon as everything is queued; up to two
    // steps may be in flight; stepEnd waits for the OLDEST one.  r.repeat: the step ran into a capacity (more hits than estimated, a traceback that
    // needs the host path, backtraces beyond out.cigarCap): nothing is lost, the caller repeats the step through prefilter() + align().
    bool stepBegin(int kmerThr, unsigned binSize, int maxHits, int minDiagScore, int gapOpen, int gapExtend, const GateTable &gate, const StepOut &out, std::string &err);
    bool stepEnd(StepResult &r, std::string &err);
    size_t stepsInFlight() const;
    bool drain(std::string &err);      // both streams idle; uncollected steps keep their results (used before a step is repeated the synchronous way)
    void stepCapacityHint(size_t pairs);
    uint32_t maxQueryLen() const;
    size_t alignRetries() const { return alignRetries_; }      // pairs of the last align() that went through the host-driven traceback

    // rescorediagonal: ungapped score on one diagonal of the UNMASKED residues, align matrix, no composition bias
    // (modes 0, 3 and 4 compare letters: queryLetters = the resident queries' ASCII, concatenated like their residues)
    bool rescoreDiagonals(const SwPair *pairs, const uint16_t *diags, size_t n, int mode, const char *queryLetters, std::vector<DiagRescore> &out, std::string &err);

    // profiling aid: streaming read of `bytes` with laneBytes (4/8/16) per lane, to calibrate the FETCH_SIZE counter
    bool diagStreamRead(uint64_t bytes, int laneBytes, double &ms, std::string &err);

    bool sync(std::string &err);
    // result-DB text formatted on the device (sp_gpu_text.inc): entries of the resident batch in query order, each followed by a NUL; entryLen[q] = bytes of entry q.
    // hitsText: the prefilter entries of the last pass's hit list.  alignHitsText: aligns the resident hits and writes the alignment entries (accepted records, sorted);
    // fellBack = true (and nothing written): a pair needed the host-driven traceback or an E-value has no table text -- the caller formats on the host.
    // used > cap: nothing copied, `used` says how much room the text needs.
    bool hitsText(char *text, uint64_t cap, uint64_t &used, uint32_t *entryLen, bool &declined, std::string &err);      // declined: the text would not fit 32-bit offsets (nothing written)
    // the index text of the entries the last hitsText / alignHitsText call produced ("<key>\t<offset>\t<length>\n", offsets from `base`): keys[e] names entry e
    bool textIndex(const uint32_t *keys, size_t n, uint64_t base, char *text, uint64_t cap, uint64_t &used, std::string &err);
    bool alignHitsText(int gapOpen, int gapExtend, const GateTable &gate, AlignHost &host, double minSeqId, uint32_t minAlnLen, char *text, uint64_t cap, uint64_t &used, uint32_t *entryLen, bool &fellBack, std::string &err);
    void setTiming(int level);
    void setMatchBlocksPerCU(int n);   // 0 = as many as fit
    bool setKnob(const std::string &name, int value);      // "match_small", "match_dense2", "match_heavy_threshold"; false: unknown
    void makeCurrent();            // HIP's current device is per host thread: every entry point that may run on another thread calls this     // 0 none, 1 k_match only (default), 2 every stage
    void *stream() const { return stream_; }

private:
    bool prefetchHits(std::string &err);
    // recordsLeft / hitsLeft: HIP events (or nullptr) the kernels that rewrite the record / hit buffers wait for on the device
    bool queueAlign(size_t cap, const uint32_t *nDev, const SwPair *hostPairs, int gapOpen, int gapExtend, const GateTable &gate, bool assemble, bool prefetch, void *recordsLeft, std::string &err);
    bool queueSelect(unsigned binSize, int maxHits, int minDiagScore, void *hitsLeft, std::string &err);
    bool candidatesTail(int minDiagScore, std::string &err);
    bool matchAndScore(int kmerThr, bool withOrd, std::string &err, int probeMode = 0, size_t qFrom = 0, size_t qTo = 0);
    bool finishPrefilter(bool matched, std::string &err);
    struct Impl;
    Impl *d_;
    void *stream_ = nullptr;
    std::string name_;
    size_t nq_ = 0;
    int maxHits_ = 0;
    size_t alignRetries_ = 0;
    PrefilterCounters counters_{};
    std::vector<uint32_t> overflowList_;
    KernelTimes times_{};
};

}  // namespace sp
