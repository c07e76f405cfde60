// HIP kernels (gfx950 / MI355X) for SpacePHARER's spacer->phage-ORF search hot path, and the device context
// that keeps the target index, the target residues and the similar-k-mer tables resident in HBM.
//
// Integer scan/lookup work bounded by random HBM gathers -- no MFMA.  Design notes are in DESIGN.md; the
// reference semantics each kernel reproduces are cited next to it (paths under lib/mmseqs/src/).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <sched.h>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "sp_gpu.h"
#include "sp_dist.h"

// the reference's AVX2 build has no FMA contraction; a few float/double expressions below must round identically
#pragma clang fp contract(off)

namespace sp {


#define HIP_OK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            err = std::string(#call) + ": " + hipGetErrorString(e_);                                   \
            return false;                                                                              \
        }                                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int laneId() { return (int) (threadIdx.x & 63); }

// LDS traffic between lanes of ONE wave: order it without a workgroup barrier (the four waves of a block work
// on different queries and run different trip counts)
__device__ __forceinline__ void waveSync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// DPP data movement (gfx9 family): row_shr:n shifts inside a row of 16 lanes, row_bcast15/31 forward a row's last lane
// to the following rows, wave_shr:1 shifts the whole wave by one lane.  One VALU op each -- no trip through the LDS
// crossbar like ds_bpermute (__shfl*), which matters because the kernels here are latency-bound.
#define SP_DPP_ROW_SHR(n) (0x110 | (n))
#define SP_DPP_ROW_BCAST15 0x142
#define SP_DPP_ROW_BCAST31 0x143
#define SP_DPP_WAVE_SHR1 0x138
#define SP_DPP_WAVE_SHL1 0x130

__device__ __forceinline__ uint32_t waveInclusiveScan(uint32_t v, int lane) {
    (void) lane;
    int x = (int) v;
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_SHR(1), 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_SHR(2), 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_SHR(4), 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_SHR(8), 0xF, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_BCAST15, 0xA, 0xF, false);
    x += __builtin_amdgcn_update_dpp(0, x, SP_DPP_ROW_BCAST31, 0xC, 0xF, false);
    return (uint32_t) x;
}

// value of the previous lane (0 for lane 0)
__device__ __forceinline__ int waveShiftUp1(int v) { return __builtin_amdgcn_update_dpp(0, v, SP_DPP_WAVE_SHR1, 0xF, 0xF, false); }

// NOTE: every value that steers wave-uniform control flow is forced into an SGPR (readfirstlane/readlane).  Left
// in VGPRs, the compiler structurises such loops as divergent and lanes can leave a loop one by one, which
// deadlocks against the cross-lane operations inside it.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t) __builtin_amdgcn_readfirstlane((int) v); }
__device__ __forceinline__ uint32_t lane63(uint32_t v) { return (uint32_t) __builtin_amdgcn_readlane((int) v, 63); }

// work-queue pop without a divergent `if (lane == 0)`: every lane issues the add (0 for all but lane 0); the
// compiler's atomic optimiser folds it into one atomic per wave and lane 0 receives the old counter value
__device__ __forceinline__ uint32_t popQueue(uint32_t *queue, int lane) {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(queue, 1u);
    return uni(v);
}
// the same for an amount: lane 0 adds, everybody gets the old value (called from wave-uniform code)
__device__ __forceinline__ uint32_t waveAtomicAdd(uint32_t *p, uint32_t amount, int lane) {
    uint32_t v = 0;
    if (lane == 0) v = atomicAdd(p, amount);
    return uni(v);
}

__device__ __forceinline__ uint32_t waveSum(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d);
    return uni(v);
}

// LDS byte address of a __shared__ object (the destination base of an LDS-DMA, `global_load_lds_dwordx4`: lane l's 16 bytes land at base + 16 l, base wave-uniform
// in M0).  Such requests are invisible to the compiler's wait bookkeeping: the reader waits with its own s_waitcnt vmcnt(0) (memory operations retire in
// order, so the compiler's own counted waits only become stricter).  See `stage` in k_match.
__device__ __forceinline__ uint32_t ldsAddr(const void *p) { return (uint32_t) (uintptr_t) (const __attribute__((address_space(3))) void *) p; }

__device__ __forceinline__ uint64_t laneMaskLt(int lane) { return lane == 0 ? 0ull : (~0ull >> (64 - lane)); }
// number of set bits of a wave mask below this lane: v_mbcnt_lo + v_mbcnt_hi on the mask's scalar halves (popcount(m & laneMaskLt) builds the lane mask first: 4 instructions)
__device__ __forceinline__ uint32_t lanePrefixCount(uint64_t m) { return (uint32_t) __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u)); }

// bitonic sort of n (power of two) u64 keys by one wave; keys may live in LDS or in global scratch
__device__ void waveBitonicSort(uint64_t *keys, uint32_t n, int lane) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (n >> 1); t += 64) {
                // t-th compare-exchange of this stage: insert a zero bit at position log2(j)
                const uint32_t lo = t & (j - 1);
                const uint32_t i = ((t - lo) << 1) | lo;
                const uint32_t p = i | j;
                const uint64_t a = keys[i], b = keys[p];
                const bool up = ((i & k) == 0);
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            waveSync();
        }
    }
}


// Slot numbering of the device index.  The reference numbers a k-mer  first3mer + 8000 * second3mer  (Indexer).  What bounds the search is the number
// of different 128-byte lines of the occupancy bitmap that the similar k-mers of a window touch, so slots are numbered for that:
//  - 3-mers are renumbered along a Z-curve: residues in an order that keeps exchangeable ones next to each other (C | G A S T P | N D E Q | K R H | Y W F |
//    M L I V instead of the alphabet), the three positions' 5-bit digits bit-interleaved, the 8000 used codes ranked 0..7999.  The similar 3-mers of a
//    3-mer differ from it in a few positions by such residues: on the curve they fall into a few short runs.
//  - a k-mer (first 3-mer a', second 3-mer b', both renumbered) lies in the 32 x 32 tile (a' / 32, b' / 32), and a tile is the 1024 bits of one 128-byte
//    line: the similar k-mers of a window -- (run of a') x (run of b') -- cover a few tiles instead of one row of 8000 slots per a'.
// slot = tile x 1024 + (a' % 32) x 32 + b' % 32 = slotA(a') + slotB(b').  The renumbering lives in the tables (the partner indices stored in the
// similar-3-mer table and the keys of the index construction are renumbered; unperm3 / kmerOfSlot are test-time functions).
struct Perm3Tables { uint16_t fwd[8000], inv[8000]; };
static const Perm3Tables &perm3Host() {
    static const Perm3Tables t = [] {
        constexpr uint8_t pi[20] = {2, 0, 7, 8, 15, 1, 12, 18, 10, 17, 16, 6, 5, 9, 11, 3, 4, 19, 14, 13};     // A C D E F G H I K L M N P Q R S T V W Y -> position
        std::vector<std::pair<uint32_t, uint32_t>> code(8000);
        for (uint32_t idx = 0; idx < 8000; idx++) {
            const uint32_t d[3] = {pi[idx % 20u], pi[(idx / 20u) % 20u], pi[idx / 400u]};
            uint32_t z = 0;
            for (int bit = 0; bit < 5; bit++) for (int j = 0; j < 3; j++) z |= ((d[j] >> bit) & 1u) << (3 * bit + j);
            code[idx] = {z, idx};
        }
        std::sort(code.begin(), code.end());
        Perm3Tables r;
        for (uint32_t rank = 0; rank < 8000; rank++) { r.fwd[code[rank].second] = (uint16_t) rank; r.inv[rank] = (uint16_t) code[rank].second; }
        return r;
    }();
    return t;
}
__device__ uint16_t g_perm3[8000], g_unperm3[8000];        // filled per device by Gpu::init
__host__ __device__ __forceinline__ uint32_t perm3(uint32_t idx) {       // reference 3-mer number -> renumbered
#if defined(__HIP_DEVICE_COMPILE__)
    return g_perm3[idx];
#else
    return perm3Host().fwd[idx];
#endif
}
__host__ __device__ __forceinline__ uint32_t unperm3(uint32_t idx) {
#if defined(__HIP_DEVICE_COMPILE__)
    return g_unperm3[idx];
#else
    return perm3Host().inv[idx];
#endif
}
constexpr uint32_t kTilesPerSide = 250u;                    // 8000 / 32
__host__ __device__ __forceinline__ uint32_t slotA(uint32_t first3p) { return (first3p >> 5) * (kTilesPerSide * 1024u) + ((first3p & 31u) << 5); }
__host__ __device__ __forceinline__ uint32_t slotB(uint32_t second3p) { return ((second3p >> 5) << 10) | (second3p & 31u); }
__host__ __device__ __forceinline__ uint32_t slotOfKmer(uint32_t first3p, uint32_t second3p) { return slotA(first3p) + slotB(second3p); }      // renumbered 3-mers in
__host__ __device__ __forceinline__ uint32_t kmerOfSlot(uint32_t slot) {                                                                        // reference k-mer id out
    const uint32_t tile = slot >> 10, a3 = ((tile / kTilesPerSide) << 5) | ((slot >> 5) & 31u), b3 = ((tile % kTilesPerSide) << 5) | (slot & 31u);
    return unperm3(a3) + 8000u * unperm3(b3);
}

__device__ __forceinline__ int extScore(uint32_t packed) { return (int) (packed >> 16) - 32768; }
__device__ __forceinline__ uint32_t extIndex(uint32_t packed) { return packed & 0xFFFFu; }

#include "sp_gpu_prep.inc"
#include "sp_gpu_match.inc"
#include "sp_gpu_match_dense.inc"
#include "sp_gpu_match_grp.inc"
#include "sp_gpu_select.inc"
#include "sp_gpu_probe.inc"

#include "sp_gpu_sw.inc"
#include "sp_gpu_text.inc"
#include "sp_gpu_index.inc"

// ---------------------------------------------------------------------------------------------------------------
// device context
// ---------------------------------------------------------------------------------------------------------------
// Device blocks a one-shot process (a fresh module process: SP_ONESHOT, set by the front end) hands from buffers that are done to buffers that come later -- the index build's
// scratch (12 GB at 2x10^8 residues) is dead when the search starts, and what the search needs (diagonal records, per-wave scratch, Smith-Waterman pools, records) fits into it.
// A fresh process then asks the driver for ~20 GB instead of 32: on the pool's slow boxes the allocations of the first prefilter alone took 0.5 s (profiles/r06_fresh_process_pool.txt).
// A block taken from the pool is cleared like a new one (code that relies on what hipMalloc hands out must not see another buffer's content); blocks are split, never
// returned to the driver before the process ends.  Served / in-process contexts keep their buffers as before: the pool is off.
struct DevPool {
    std::mutex mu; std::vector<std::pair<void *, size_t>> blocks;
    static bool on() { static const bool v = getenv("SP_ONESHOT") != nullptr && getenv("SP_NO_DEVICE_POOL") == nullptr; return v; }
    static DevPool &get() { static DevPool *p = new DevPool; return *p; }      // (never destroyed: the driver reclaims the blocks with the process)
    static constexpr size_t kMin = (size_t) 32 << 20;
    void *take(size_t bytes, size_t &got) {      // the smallest block that holds `bytes`; a block more than twice as large is split, its tail stays in the pool
        std::lock_guard<std::mutex> lk(mu);
        size_t best = blocks.size();
        for (size_t i = 0; i < blocks.size(); i++) if (blocks[i].second >= bytes && (best == blocks.size() || blocks[i].second < blocks[best].second)) best = i;
        if (best == blocks.size()) return nullptr;
        void *p = blocks[best].first; got = blocks[best].second;
        const size_t piece = (bytes + ((size_t) 2 << 20) - 1) & ~(((size_t) 2 << 20) - 1);      // (pieces start on 2 MB boundaries of the block)
        if (got > bytes * 2 && got - piece >= kMin) { blocks[best] = {static_cast<char *>(p) + piece, got - piece}; got = piece; }
        else blocks.erase(blocks.begin() + (long) best);
        return p;
    }
    void give(void *p, size_t bytes) { std::lock_guard<std::mutex> lk(mu); blocks.emplace_back(p, bytes); }
};

template <typename T>
struct DevBuf {
    T *p = nullptr; size_t n = 0; size_t bytes = 0;
    bool ensure(size_t count, std::string &err) {
        if (count <= n && p) return true;
        release();
        size_t want = DevPool::on() ? count + 64 : count + count / 8 + 64;      // (room to grow into: a long-lived context's next batch; a one-shot process sizes exactly)
        static const bool trace = getenv("SP_MEM_TRACE") != nullptr;      // (every device allocation of 32 MB or more, with the element size: which buffers a call asks the driver for)
        if (DevPool::on() && want * sizeof(T) >= DevPool::kMin) {
            size_t got = 0;
            if (void *q = DevPool::get().take(want * sizeof(T), got)) {
                if (trace) fprintf(stderr, "[sp] recycled  %9.1f MB  for %.1f MB (%zu elements of %zu bytes)\n", (double) got / 1e6, (double) (want * sizeof(T)) / 1e6, want, sizeof(T));
                if (hipMemset(q, 0, want * sizeof(T)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) (void) hipGetLastError();      // (cleared before any stream of the process can touch it; a block that cannot be cleared is dropped -- it may be a piece of a larger allocation)
                else { p = (T *) q; bytes = got; n = got / sizeof(T); return true; }
            }
        }
        if (trace && want * sizeof(T) >= (32u << 20)) fprintf(stderr, "[sp] hipMalloc %9.1f MB  (%zu elements of %zu bytes)\n", (double) (want * sizeof(T)) / 1e6, want, sizeof(T));
        hipError_t e = hipMalloc((void **) &p, want * sizeof(T));
        if (e != hipSuccess) { p = nullptr; err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
        n = want; bytes = want * sizeof(T);
        return true;
    }
    void release() {
        if (p) { if (DevPool::on() && bytes >= DevPool::kMin) { (void) hipDeviceSynchronize(); DevPool::get().give(p, bytes); } else (void) hipFree(p); }      // (hipFree waits for the device; so does the hand-over)
        p = nullptr; n = 0; bytes = 0;
    }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }      // (a context that goes away gives its device memory back: a long-lived process creates and destroys many)
};

struct SwDev {
    DevBuf<SwPair> pairs; DevBuf<SwForward> fwd; DevBuf<uint32_t> list; DevBuf<int32_t> revRef, revRead, revScore;
    DevBuf<signed char> qcomp; DevBuf<int> qbias;
    DevBuf<uint64_t> dirOff, rowOff, btOff; DevBuf<uint32_t> dirCap, rowCap, btCap;
    DevBuf<signed char> dirPool; DevBuf<int> rowPool; DevBuf<char> btPool; DevBuf<SwFull> full;
    DevBuf<uint16_t> diag; DevBuf<DiagRescore> resc; DevBuf<char> qletters;
    DevBuf<uint32_t> passList, redoList, wordPass, counters; DevBuf<unsigned long long> cursors; DevBuf<uint8_t> gatePass; DevBuf<uint16_t> gateSlot;
    size_t preparedFor = (size_t) -1; const void *preparedPtr = nullptr;
    DevBuf<double> evalTab; DevBuf<int32_t> bitTab; uint64_t gateVersion = 0;
    DevBuf<uint32_t> cigLen, cigOff; DevBuf<AlnRec> rec; DevBuf<char> cig;
};

struct Gpu::Impl {
    hipStream_t stream = nullptr;
    hipEvent_t ev[2][10] = {};                   // stage timing events; the device-driven steps alternate between the two sets
    int evSet = 0;
    int numCU = 256;
    // environment knobs (diagnostics / experiments), read once per context instead of once per pass
    struct Knobs {
        int matchSmall = -1, matchPerCU = 0, matchHeavy = -1, matchStatic = 4, matchDChunk = 4, weighExact = -1, matchGroup = 1, matchOrder = 1, selChunkDense = 8, selDynamic = 0, probePerQuery = 192, matchDense2 = 0;
        bool debug = false, prepSplit = false, swRows32 = false, swRows16 = false, swOldForward = false;
        void read() {
            auto num = [](const char *k, int d) { const char *e = getenv(k); return e ? atoi(e) : d; };
            matchSmall = num("SP_MATCH_SMALL", -1); matchGroup = num("SP_MATCH_GROUP", 1); matchOrder = num("SP_MATCH_ORDER", 1); { const int we = num("SP_MATCH_WEIGH_EXACT", -1); weighExact = we < 0 ? -1 : (we ? 1 : 0); } matchPerCU = num("SP_MATCH_PERCU", 0); selChunkDense = num("SP_SEL_CHUNK", 8); selDynamic = num("SP_SEL_DYNAMIC", 0); matchHeavy = num("SP_MATCH_HEAVY", -1); probePerQuery = num("SP_PROBE_PER_QUERY", 192); matchDense2 = num("SP_MATCH_DENSE2", 0);
            matchStatic = std::min(8, std::max(0, num("SP_MATCH_STATIC", 4))); matchDChunk = std::min(32, std::max(1, num("SP_MATCH_DCHUNK", 4)));      // (a chunk's offsets live in the lanes of a wave)
            debug = getenv("SP_DEBUG") != nullptr; prepSplit = getenv("SP_PREP_SPLIT") != nullptr; swRows32 = getenv("SP_SW_ROWS32") != nullptr;
            swRows16 = getenv("SP_SW_ROWS16") != nullptr; swOldForward = getenv("SP_SW_OLD_FORWARD") != nullptr;
        }
    } knob;
    // What several search contexts of one device can SHARE (Gpu::shareFrom): the matrices and similar-k-mer tables, and the staged target set.  Both are written once,
    // by the context that builds them, and only read by the kernels; four contexts of one process (bench.py, a resident server) hold ONE copy -- 256 MB of similar
    // 3-mers, the index, the residues -- instead of four, in HBM and in the 256 MB Infinity Cache.
    struct Tables {
        DevBuf<uint32_t> ext3; DevBuf<uint16_t> ext3cnt; DevBuf<uint32_t> heavyEst; int cntMin = 0; DevBuf<short> seedSub; DevBuf<double> seedBack; DevBuf<signed char> ungSub; DevBuf<signed char> alnSub;
        DevBuf<short> alnSub16; DevBuf<double> alnBack;
        int ungMinAbs = 0, alnMinAbs = 0;
        std::mutex mu;      // a sharing context may add the similar-k-mer tables to a set that was staged without them
    };
    struct Targets {
        DevBuf<uint64_t> buckets; bool bucketsValid = false; DevBuf<uint64_t> buckets2; bool buckets2Valid = false;      // (buckets2: the 64-byte double buckets of k_match_dense)
        DevBuf<uint64_t> slotDirect; bool slotDirectValid = false; size_t indexSlots = 0; uint64_t buildSerial = 0;      // large-footprint layout of the slot table, derived from slots / ientries on first use
        DevBuf<uint64_t> pbits; DevBuf<uint16_t> prank; DevBuf<uint32_t> prow; DevBuf<uint64_t> slots; DevBuf<uint64_t> ientries; DevBuf<uint8_t> tmasked; DevBuf<uint8_t> tplain; DevBuf<char> tascii; DevBuf<uint64_t> toff; DevBuf<uint32_t> tkeys;
        size_t nTargets = 0; size_t indexEntries = 0; uint64_t maxTLen = 0;      // (maxTLen: longest target -- k_sw_fwd1p packs end columns in 16 bits)
        bool denseIndex = false;      // 1.5+ entries per occupied k-mer on average: k_match runs with its large footprint
        std::mutex mu;      // the bucket table is derived by whichever sharing context needs it first
    };
    std::shared_ptr<Tables> tab = std::make_shared<Tables>();
    std::shared_ptr<Targets> tgt = std::make_shared<Targets>();
    unsigned long long matchSpan[2] = {0, 0};      // wall clock at k_match's first wave start / last wave end of the last pass
    int wallKHz = 100000;            // rate of the device's constant wall clock (wall_clock64)
    bool swBiasFresh = false;        // k_prep8 of the current pass left the Smith-Waterman bias of the resident queries in swd.qcomp / swd.qbias
    // index construction scratch (kept for the next target set)
    DevBuf<uint32_t> ixKeys, ixKeys2, ixKeep, ixPos, ixKmer, ixHead, ixSlotOf, ixHeadPos, ixCnt, ixBefore; DevBuf<uint64_t> ixVals, ixVals2;
    DevBuf<unsigned char> ixTemp; DevBuf<float> ttProbs; DevBuf<double> ttScale, ttLr; DevBuf<uint64_t> ttPlane; DevBuf<unsigned long long> ixCounters;
    DevBuf<uint32_t> txLen, txOff, txLineOff, txLineLen, txFlags, txWaveList; DevBuf<char> txText, txIdxText; DevBuf<uint32_t> txKeys, txIdxLen, txIdxOff; DevBuf<unsigned long long> txTotal; DevBuf<unsigned char> txEval; uint64_t txEvalVersion = 0;      // device-side result text (sp_gpu_text.inc)
    // queries
    DevBuf<uint8_t> qres; DevBuf<uint32_t> qoff; DevBuf<float> biasf; DevBuf<short> qthr; DevBuf<signed char> qcorr; DevBuf<int> qbias; DevBuf<int> qself;
    size_t qTotal = 0; uint32_t maxQLen = 0;
    std::vector<uint32_t> hQoff;
    // match / select
    DevBuf<uint32_t> qweight, qorder, orderHist, orderHeader, orderQueues; DevBuf<uint64_t> spill; DevBuf<uint32_t> heavyList; DevBuf<uint32_t> heavyCnt; DevBuf<uint8_t> heavyFlag; DevBuf<unsigned long long> wavePartial; DevBuf<uint32_t> queue; DevBuf<DiagRec> diagRec; DevBuf<uint32_t> diagQuery; DevBuf<uint16_t> diagScore; DevBuf<uint8_t> diagFlag;
    DevBuf<uint32_t> qDiagBase; DevBuf<uint32_t> qDiagCnt; DevBuf<unsigned long long> counters;
    DevBuf<GpuHit> outHits; DevBuf<GpuHit> outStage; DevBuf<uint32_t> outQuery; DevBuf<uint32_t> outCount; DevBuf<uint32_t> outBase; DevBuf<uint32_t> outOffset;
    uint32_t matchBlocks = 0;
    size_t diagCap = 0;
    DevBuf<unsigned char> scanTemp; DevBuf<uint32_t> selWant, selWantOff;
    unsigned long long *mailbox = nullptr;       // page-locked landing area for counters
    int device = 0;
    bool searcher = false;                       // counted in g_searchers: this context holds a query batch
    int matchPerCU = 0;                          // resident k_match blocks per CU (0 = what the occupancy calculator says)
    char *ring[2] = {nullptr, nullptr}; hipEvent_t ringEv[2] = {nullptr, nullptr};      // staging of device -> host copies into ordinary memory (copyOut)
    bool copyOut(void *dst, const void *srcDev, size_t bytes, std::string &err);
    int timing = 1;                              // 0: no HIP events in the pipeline, 1: around k_match only (each record costs ~6 us of idle GPU), 2: every stage
    // the hit list starts travelling to the host on its own stream as soon as a prefilter pass is complete, so that the copy
    // overlaps whatever the caller queues next (typically the align kernels)
    hipStream_t copyStream = nullptr; hipEvent_t hitsOnHost = nullptr, alnOnHost = nullptr; bool alnInFlight = false;
    // Host-driven passes wait for their RECORDS on the copy stream, not on an event recorded behind them (round 6): the marker an event record puts behind a copy sits in a
    // hardware queue that other contexts' streams share and holds their kernels back until the copy is done (SP_COPY_EVENT_WAIT=1: the event, as before).  The copy stream
    // carries this context's result copies only, the records are the last of a pass.
    static bool copyEvents() { static const bool v = getenv("SP_COPY_EVENT_WAIT") != nullptr; return v; }
    hipError_t waitCopies(hipEvent_t e) { return copyEvents() ? hipEventSynchronize(e) : hipStreamSynchronize(copyStream); }
    PinVec<GpuHit> hitStage; PinVec<uint32_t> hitOffStage; bool hitsInFlight = false;
    GpuHit *sinkHits = nullptr; uint32_t *sinkOff = nullptr; size_t sinkCap = 0, sinkN = 0; bool sinkUsed = false;      // caller-owned delivery buffers (setHitSink)
    // exclusive scan of n counts into n + 1 offsets (offset[n] = total); count must have n + 1 readable entries.
    // Small inputs: one block (k_scan_counts); larger ones: rocPRIM's decoupled look-back scan.
    bool scanCounts(const uint32_t *count, uint32_t n, uint32_t *offset, std::string &err) {
        if (n <= 4096) { hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, count, n, offset); return true; }
        if (n <= 64u * kScanTile) { hipLaunchKernelGGL(k_scan_counts_tiles, dim3((n + kScanTile - 1) / kScanTile), dim3(256), 0, stream, count, n, offset); return true; }
        size_t bytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, count, offset, (int) (n + 1), stream) != hipSuccess) { err = "scan: temp size query failed"; return false; }
        if (!scanTemp.ensure(bytes + 256, err)) return false;
        bytes = scanTemp.n;
        if (hipcub::DeviceScan::ExclusiveSum(scanTemp.p, bytes, count, offset, (int) (n + 1), stream) != hipSuccess) { err = "scan failed"; return false; }
        return true;
    }
    // target-sharded search
    DevBuf<uint32_t> diagOrd; DevBuf<uint32_t> spillOrd; DevBuf<Candidate> cand; DevBuf<Candidate> candUnion; DevBuf<Candidate> candByQuery; DevBuf<Candidate> candPad;
    DevBuf<uint32_t> xOff, xOffAll; DevBuf<GpuHit> xHit, xHitAll;      // send / receive buffers of the hit-list all-gather (query partition)
    DevBuf<uint32_t> candCnt; DevBuf<uint32_t> candFill; DevBuf<uint32_t> candOffset; DevBuf<uint64_t> mergeScratch;
    // ... its probe-list cut: all shards' occupancy maps + their union; the probes this context enumerated, by destination shard; the probes it was sent
    DevBuf<uint64_t> shardMaps, unionMap; DevBuf<uint8_t> shardOwners; uint32_t nShardMaps = 0; uint64_t probeMapsSerial = 0;
    DevBuf<uint4> pbRec[kMaxProbeShards]; DevBuf<uint32_t> pbQCnt[kMaxProbeShards]; DevBuf<uint32_t> probeCursor; size_t probeCap = 0, probeSlice = 0; uint32_t probeCnt[kMaxProbeShards] = {};
    DevBuf<uint4> pinRec, pxRec; DevBuf<uint32_t> pinCnt, probeOff; DevBuf<unsigned long long> probeList;      // received records, probes per query, their prefix sums, the lists by query
    uint32_t nCand = 0;
    bool candFailed = false; std::string candError;      // the last candidate pass of this context failed (a query beyond the hit-stream limit ...): allgatherCandidates tells every rank
    SwState sw;
    SwDev swd;
    // device-driven steps (stepBegin / stepEnd): the whole pass -- prefilter, selection, align, record assembly, delivery -- is queued without a
    // host round trip; counts stay on the device, results and counters travel on the copy stream.  Up to two steps in flight.
    struct StepSlot { hipEvent_t selected = nullptr, assembled = nullptr, hitsOnHost = nullptr, alnOnHost = nullptr; size_t cap = 0; StepOut out;
                      bool late = false, copied = false; size_t nq = 0; const GpuHit *dHits = nullptr; const uint32_t *dOffsets = nullptr; const AlnRec *dRec = nullptr; const char *dCig = nullptr; size_t dCigN = 0; };
    // late copies (SP_STEP_LATE_COPY): a step's results leave the device when stepEnd() has seen its last kernel finish, from buffers the NEXT step does not write -- no copy is
    // queued behind an event that has not happened yet, no kernel waits for a copy (see stepBegin)
    DevBuf<GpuHit> outHitsAlt; DevBuf<uint32_t> outOffsetAlt; DevBuf<AlnRec> recAlt; DevBuf<char> cigAlt;
    hipError_t issueLateCopies(StepSlot &ss) {      // the results of a finished step, on the copy stream (stepEnd, or drain() for steps nobody has collected yet)
        if (!ss.late || ss.copied) return hipSuccess;
        hipError_t e = hipMemcpyAsync(ss.out.offsets, ss.dOffsets, (ss.nq + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, copyStream);
        if (e == hipSuccess) e = hipMemcpyAsync(ss.out.hits, ss.dHits, ss.cap * sizeof(GpuHit), hipMemcpyDeviceToHost, copyStream);
        if (e == hipSuccess) e = hipMemcpyAsync(ss.out.records, ss.dRec, ss.cap * sizeof(AlnRec), hipMemcpyDeviceToHost, copyStream);
        const size_t cigBytes = (size_t) std::min<uint64_t>(ss.out.cigarCap, (uint64_t) ss.dCigN);
        if (e == hipSuccess && cigBytes) e = hipMemcpyAsync(ss.out.cigars, ss.dCig, cigBytes, hipMemcpyDeviceToHost, copyStream);
        ss.copied = e == hipSuccess;
        return e;
    }
    StepSlot step[2];
    uint64_t stepsBegun = 0, stepsEnded = 0;
    bool drained = false;                        // every queued step has finished on the device and its results are in host memory (drain())
    size_t stepCap = 0;                          // pairs the device buffers of a step are sized for (grows with what the steps deliver)
    DevBuf<uint32_t> stepWords;                  // [0] hits of the pass, [1] pairs aligned = min(hits, capacity), [2] flags
    uint32_t *stepMail = nullptr;                // page-locked: 2 x 64 words, one block per step in flight
};

// contexts per device that hold a query batch (see matchAndScore: how the heavy queries are found)
static std::atomic<int> g_searchers[64];

Gpu::Gpu() : d_(new Impl) {}
// Stream pairs (search stream, copy stream) are created once per process and device, in order, and handed from context to context: the HIP runtime
// multiplexes streams onto 4 hardware queues in the order they are created, and how the streams of several contexts pair up on those queues is worth
// 10 % of the throughput of four contexts.  A pool keeps the pairing of the first contexts of a process for every later one -- a resident
// server, a bench that measures several configurations -- instead of leaving it to what was created and destroyed before.
struct StreamPair { hipStream_t main = nullptr, copy = nullptr; bool busy = false; };
static std::mutex g_streamPoolMutex;
static std::map<int, std::vector<StreamPair>> g_streamPool;
static bool takeStreams(int device, hipStream_t &mainS, hipStream_t &copyS) {
    std::lock_guard<std::mutex> lock(g_streamPoolMutex);
    std::vector<StreamPair> &pool = g_streamPool[device];
    for (StreamPair &p : pool) if (!p.busy) { p.busy = true; mainS = p.main; copyS = p.copy; return true; }
    StreamPair p;
    if (hipStreamCreateWithFlags(&p.main, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipStreamCreateWithFlags(&p.copy, hipStreamNonBlocking) != hipSuccess) { (void) hipStreamDestroy(p.main); return false; }
    p.busy = true; pool.push_back(p);
    mainS = p.main; copyS = p.copy;
    return true;
}
static void giveStreams(int device, hipStream_t mainS) {
    std::lock_guard<std::mutex> lock(g_streamPoolMutex);
    for (StreamPair &p : g_streamPool[device]) if (p.main == mainS) p.busy = false;
}

Gpu::~Gpu() {
    if (d_) {
        if (d_->searcher) g_searchers[d_->device & 63].fetch_sub(1, std::memory_order_relaxed);
        if (d_->stream) (void) hipStreamSynchronize(d_->stream);
        for (auto &set : d_->ev) for (hipEvent_t e : set) if (e) (void) hipEventDestroy(e);
        for (Impl::StepSlot &ss : d_->step) for (hipEvent_t e : {ss.selected, ss.assembled, ss.hitsOnHost, ss.alnOnHost}) if (e) (void) hipEventDestroy(e);
        if (d_->stepMail) pinnedFree(d_->stepMail);
        for (int k = 0; k < 2; k++) { if (d_->ring[k]) pinnedFree(d_->ring[k]); if (d_->ringEv[k]) (void) hipEventDestroy(d_->ringEv[k]); }
        if (d_->mailbox) pinnedFree(d_->mailbox);
        if (d_->copyStream) (void) hipStreamSynchronize(d_->copyStream);
        if (d_->stream) giveStreams(d_->device, d_->stream);      // (idle now: the next context of this device takes the pair)
        if (d_->hitsOnHost) (void) hipEventDestroy(d_->hitsOnHost);
        if (d_->alnOnHost) (void) hipEventDestroy(d_->alnOnHost);
        d_->hitStage = PinVec<GpuHit>(); d_->hitOffStage = PinVec<uint32_t>();
        delete d_;
    }
}

// The two host waits of a search step (hit count after the prefilter, record count after the alignment) are short and
// sit on the critical path: poll the stream instead of sleeping on an interrupt (tens of microseconds to wake up).
// Short waits on the critical path are polled; a poll that has not succeeded after a few hundred tries gives the core away between tries (several contexts per
// process, several processes per node: more pollers than the process may have CPUs -- cgroup quotas -- would otherwise starve the threads they wait for).
static inline void pollPause(unsigned &tries) {
    if (++tries < 256u) __builtin_ia32_pause(); else sched_yield();
}
static hipError_t spinSync(hipStream_t s) {
    unsigned tries = 0;
    for (;;) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
        pollPause(tries);
    }
}

// stage timing: an event record between two kernels opens a ~6 us bubble, so only the roofline kernel is bracketed by default
#define SP_EV(i, level) do { if (d.timing >= (level)) HIP_OK(hipEventRecord(d.ev[d.evSet][i], d.stream)); } while (0)
static float evMs(int timing, int level, hipEvent_t a, hipEvent_t b) {
    float t = 0; if (timing < level || hipEventElapsedTime(&t, a, b) != hipSuccess) return __builtin_nanf(""); return t;
}
void Gpu::setMatchBlocksPerCU(int n) { d_->matchPerCU = n < 0 ? -1 : n; }      // n > 0: at most n; 0: the library's choice (matchAndScore); -1: all that fit
// scheduling knobs of k_match (they change the order of the work, never a result): footprint -1 by index density / 0 large / 1 small;
// similar k-mers from which a query is started first, -1 = by batch size, 0 = no list.  Initial values: SP_MATCH_SMALL / SP_MATCH_HEAVY.
bool Gpu::setKnob(const std::string &name, int value) {
    if (name == "match_weigh_exact") { d_->knob.weighExact = value < 0 ? -1 : (value ? 1 : 0); return true; }      // the heavy-query list from exact counts (table rows read) or from the calibrated estimate; -1: exact while this is the only searching context of the device
    if (name == "match_chunk") { d_->knob.matchDChunk = std::min(32, std::max(1, value)); return true; }      // queries per pop of the dynamically dealt part
    if (name == "match_static") { d_->knob.matchStatic = std::min(8, std::max(0, value)); return true; }      // eighths of the queries dealt round-robin without atomics
    if (name == "match_small") { d_->knob.matchSmall = value < 0 ? -1 : (value ? 1 : 0); return true; }
    if (name == "match_dense2") { d_->knob.matchDense2 = value ? 1 : 0; return true; }
    if (name == "probe_per_query") { d_->knob.probePerQuery = std::max(64, value); return true; }      // probe records per query and destination the enumerating side of the probe-list cut has room for (SP_PROBE_PER_QUERY)
    if (name == "match_order") { d_->knob.matchOrder = value ? 1 : 0; return true; }      // k_match_grp takes the batch in the order of decreasing weight (list schedule); 0: heavy queries first, then shares
    if (name == "match_group") { d_->knob.matchGroup = value ? 1 : 0; return true; }      // k_match_grp (queries matched four at a time) where its preconditions hold; 0: always k_match
    if (name == "match_heavy_threshold") { d_->knob.matchHeavy = value < 0 ? -1 : value; return true; }
    if (name == "step_capacity") { d_->stepCap = value < 0 ? 0 : (size_t) value; return true; }      // pairs the next device-driven step is sized for (0 = estimate again); tests
    return false;
}
void Gpu::makeCurrent() { (void) hipSetDevice(d_->device); }
void Gpu::setTiming(int level) { d_->timing = level < 0 ? 0 : (level > 2 ? 2 : level); }

// several small fills in one launch (each hipMemsetAsync is a kernel of its own plus a gap)
struct FillArgs { void *p[4]; unsigned long long bytes[4]; uint32_t value[4]; };
__global__ void k_fill(FillArgs a) {
    for (int r = 0; r < 4; r++) {
        uint32_t *w = static_cast<uint32_t *>(a.p[r]); const unsigned long long n = a.bytes[r] / 4;
        for (unsigned long long i = (unsigned long long) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long) gridDim.x * blockDim.x) w[i] = a.value[r];
    }
}

// sum of n 32-bit lengths in 64 bits (the result text's offsets are 32-bit: a batch whose text would not fit them is declined -- Gpu::hitsText)
__global__ void k_sum_u64(const uint32_t *v, uint32_t n, unsigned long long *out) {
    unsigned long long s = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += v[i];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(out, s);
}

// small results for the host: one kernel stores them straight into the page-locked mailbox (each hipMemcpyAsync of a few
// bytes is a copy kernel of its own)
struct PublishArgs { const void *src[5]; uint32_t words[5]; uint32_t at[5]; uint32_t *mailbox; };
__global__ void k_publish(PublishArgs a) {
    for (int r = 0; r < 5; r++) {
        const uint32_t *s = static_cast<const uint32_t *>(a.src[r]);
        for (uint32_t i = threadIdx.x; i < a.words[r]; i += blockDim.x) a.mailbox[a.at[r] + i] = s[i];
    }
}

// a device-driven step learns its pair count here: hits of the selection (the scan leaves the total behind the last query's offset), clipped to
// the capacity the align buffers and the caller's result buffers were sized for; a clipped step is flagged and repeated by the host
struct StepCountArgs { const uint32_t *total; uint32_t cap; uint32_t *words; };
__global__ void k_step_counts(StepCountArgs a) {
    if (threadIdx.x == 0) { const uint32_t n = *a.total; a.words[0] = n; a.words[1] = n < a.cap ? n : a.cap; a.words[2] = n > a.cap ? 1u : 0u; }
}

int Gpu::deviceCount() const { int n = 0; if (hipGetDeviceCount(&n) != hipSuccess) return 0; return n; }

// What a process pays ONCE before its first kernel can run -- the HIP runtime, the device's context, this library's code object -- without anything that belongs to a search
// context: a front end starts it on a second thread while it reads its DBs (sp_init_async)
bool Gpu::warmUp(int device, std::string &err) {
    const bool times = getenv("SP_MODULE_TIMES") != nullptr;
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (n <= 0) { err = "no HIP device visible"; return false; }
    const double t1 = now();
    HIP_OK(hipSetDevice(device % n));
    HIP_OK(hipFree(nullptr));
    const double t2 = now();
    void *p = nullptr;      // (a first allocation and a first kernel: the memory pools and the code object are there when the search context asks)
    HIP_OK(hipMalloc(&p, 1 << 20));
    const double t3 = now();
    FillArgs fa{{p, nullptr, nullptr, nullptr}, {1 << 20, 0, 0, 0}, {0u, 0u, 0u, 0u}};
    hipLaunchKernelGGL(k_fill, dim3(4), dim3(256), 0, nullptr, fa);
    HIP_OK(hipDeviceSynchronize());
    const double t4 = now();
    HIP_OK(hipFree(p));
    if (times) fprintf(stderr, "[sp] runtime (second thread): device count %.1f ms, device + context %.1f ms, first allocation %.1f ms, code object + first kernel %.1f ms\n", t1 - t0, t2 - t1, t3 - t2, t4 - t3);
    return true;
}

bool Gpu::init(int device, std::string &err) {
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (n <= 0) { err = "no HIP device visible"; return false; }
    HIP_OK(hipSetDevice(device % n));
    d_->device = device % n;
    // (hipGetDeviceProperties takes milliseconds: asked once per process and device)
    static std::mutex propMutex; static std::map<int, std::pair<std::string, int>> propCache;
    {   std::lock_guard<std::mutex> lock(propMutex);
        auto it = propCache.find(device % n);
        if (it == propCache.end()) {
            hipDeviceProp_t prop;
            HIP_OK(hipGetDeviceProperties(&prop, device % n));
            it = propCache.emplace(device % n, std::make_pair(std::string(prop.name), (int) prop.multiProcessorCount)).first;
        }
        name_ = it->second.first; d_->numCU = it->second.second;
    }
    {   // the 3-mer renumbering tables of this device (once per process and device)
        static std::mutex permMutex; static std::set<int> permDone;
        std::lock_guard<std::mutex> lock(permMutex);
        if (!permDone.count(d_->device)) {
            HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_perm3), perm3Host().fwd, sizeof(g_perm3)));
            HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_unperm3), perm3Host().inv, sizeof(g_unperm3)));
            permDone.insert(d_->device);
        }
    }
    if (!takeStreams(d_->device, d_->stream, d_->copyStream)) { err = "hipStreamCreate failed"; return false; }
    { int khz = 0; if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, d_->device) == hipSuccess && khz > 0) d_->wallKHz = khz; }
    HIP_OK(hipEventCreateWithFlags(&d_->hitsOnHost, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&d_->alnOnHost, hipEventDisableTiming));
    for (auto &set : d_->ev) for (hipEvent_t &e : set) HIP_OK(hipEventCreate(&e));
    for (Impl::StepSlot &ss : d_->step) for (hipEvent_t *e : {&ss.selected, &ss.assembled, &ss.hitsOnHost, &ss.alnOnHost}) HIP_OK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    d_->knob.read();
    stream_ = d_->stream;
    return true;
}

// waits for the search stream AND the copy stream; a faulted kernel or copy comes back as an error text (sp_ctx_sync -> SP_ERR_DEVICE) instead of being swallowed
bool Gpu::sync(std::string &err) {
    hipError_t e = hipStreamSynchronize(d_->stream);
    if (e == hipSuccess && d_->copyStream) e = hipStreamSynchronize(d_->copyStream);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { err = std::string("the context's stream failed: ") + hipGetErrorString(e); return false; }
    return true;
}

// ---- counter calibration (profiles/ only): streaming read of a known byte count, W bytes per lane per load.  The
// rocprofv3 FETCH_SIZE counter is calibrated against these launches before it is quoted for k_match.
template <typename T>
__global__ __launch_bounds__(256) void k_diag_read(const T *src, size_t n, unsigned long long *sink) {
    unsigned long long acc = 0;
    for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t) gridDim.x * blockDim.x) {
        const T v = src[i];
        const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
#pragma unroll
        for (unsigned j = 0; j < sizeof(T) / 4; j++) acc += w[j];
    }
    if (acc == 0x9E3779B97F4A7C15ull) *sink = acc;     // never true for the zero-filled buffer; keeps the loads alive
}

bool Gpu::diagStreamRead(uint64_t bytes, int laneBytes, double &ms, std::string &err) {
    void *buf = nullptr; unsigned long long *sink = nullptr;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc((void **) &sink, 8) != hipSuccess) { err = "diagStreamRead: hipMalloc failed"; return false; }
    (void) hipMemsetAsync(buf, 0, bytes, d_->stream);
    hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
    const int blocks = 256 * 16;
    (void) hipEventRecord(e0, d_->stream);
    if (laneBytes == 16) hipLaunchKernelGGL(k_diag_read<uint4>, dim3(blocks), dim3(256), 0, d_->stream, (const uint4 *) buf, (size_t) (bytes / 16), sink);
    else if (laneBytes == 8) hipLaunchKernelGGL(k_diag_read<uint2>, dim3(blocks), dim3(256), 0, d_->stream, (const uint2 *) buf, (size_t) (bytes / 8), sink);
    else hipLaunchKernelGGL(k_diag_read<uint32_t>, dim3(blocks), dim3(256), 0, d_->stream, (const uint32_t *) buf, (size_t) (bytes / 4), sink);
    (void) hipEventRecord(e1, d_->stream);
    (void) hipStreamSynchronize(d_->stream);
    float t = 0; (void) hipEventElapsedTime(&t, e0, e1); ms = t;
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    (void) hipFree(buf); (void) hipFree(sink);
    if (hipGetLastError() != hipSuccess) { err = "diagStreamRead: launch failed"; return false; }
    return true;
}

static inline int extScoreHost(uint32_t packed) { return (int) (packed >> 16) - 32768; }

template <typename T>
static bool upload(DevBuf<T> &b, const T *src, size_t n, hipStream_t s, std::string &err) {
    if (!b.ensure(n ? n : 1, err)) return false;
    if (n) HIP_OK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, s));
    return true;
}

bool Gpu::uploadTables(const SubMat &seed, const SubMat &ung, const SubMat &aln, bool kmerTables, std::string &err) {
    Impl &d = *d_;
    std::lock_guard<std::mutex> shared(d.tab->mu);      // (contexts that share the tables write them one at a time -- the same content)
    if (kmerTables && !d.tab->ext3.p) {   // similar-3-mer table + count table, built on the device (sp_gpu_index.inc); not needed to align / rescore
        const size_t N2 = (size_t) 8000 * 8000;
        int lo1 = 1 << 20, hi1 = -(1 << 20);
        Ext3Args ea{};
        for (int i = 0; i < 20; i++) for (int j = 0; j < 20; j++) { ea.sub[i * 20 + j] = seed.sub[i][j]; lo1 = std::min<int>(lo1, seed.sub[i][j]); hi1 = std::max<int>(hi1, seed.sub[i][j]); }
        const int lo = 3 * lo1, hi = 3 * hi1;           // the three positions are independent: extremes of the sum
        if (hi - lo + 2 > kCntW || hi - lo >= 1024) { err = "seed matrix 3-mer score range " + std::to_string(hi - lo) + " exceeds the count-table width"; return false; }
        d.tab->cntMin = lo;
        DevBuf<uint32_t> k1, k2, v1, v2; DevBuf<unsigned char> tmp;
        const bool laps = getenv("SP_MODULE_TIMES") != nullptr;
        auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t0 = now();
        if (!d.tab->ext3.ensure(N2, err) || !d.tab->ext3cnt.ensure((size_t) 8000 * kCntW, err) || !k1.ensure(N2, err) || !k2.ensure(N2, err) || !v1.ensure(N2, err) || !v2.ensure(N2, err)) return false;
        ea.hi = hi; ea.keys = k1.p; ea.vals = v1.p;
        const double t1 = now();
        hipLaunchKernelGGL(k_ext3_fill, dim3((unsigned) (N2 / 256)), dim3(256), 0, d.stream, ea);
        size_t tempBytes = 0;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, k1.p, k2.p, v1.p, v2.p, (int) N2, 0, 23, d.stream));
        if (!tmp.ensure(tempBytes + 256, err)) return false;
        tempBytes = tmp.n;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(tmp.p, tempBytes, k1.p, k2.p, v1.p, v2.p, (int) N2, 0, 23, d.stream));
        hipLaunchKernelGGL(k_ext3_pack, dim3((unsigned) (N2 / 256)), dim3(256), 0, d.stream, k2.p, v2.p, hi, d.tab->ext3.p);
        hipLaunchKernelGGL(k_ext3_count, dim3(8000), dim3(kCntW), 0, d.stream, d.tab->ext3.p, lo, d.tab->ext3cnt.p);
        {   // similar k-mers of a window by the slack of its best pair (the order k_match takes the queries in; see k_heavy_calib)
            if (!d.tab->heavyEst.ensure(kHeavyEstW, err)) return false;
            HeavyCalibArgs ha{d.tab->ext3.p, d.tab->ext3cnt.p, lo, {}, d.tab->heavyEst.p};
            double tot = 0, run = 0;
            for (int i = 0; i < 20; i++) tot += seed.pBack[i];
            for (int i = 0; i < 20; i++) { run += seed.pBack[i]; ha.cum[i] = tot > 0 ? (float) (run / tot) : (float) (i + 1) / 20.0f; }
            hipLaunchKernelGGL(k_heavy_calib, dim3(kHeavyEstW), dim3(256), 0, d.stream, ha);
        }
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(d.stream));
        const double t2 = now();
        for (DevBuf<uint32_t> *b : {&k1, &k2, &v1, &v2}) b->release();
        tmp.release();
        if (laps) fprintf(stderr, "[sp]   similar-3-mer table: buffers %.1f ms, fill + sort + pack + counts %.1f ms, buffers returned %.1f ms\n", t1 - t0, t2 - t1, now() - t2);
    }
    short s16[kAlph * kAlph]; signed char u8[kAlph * kAlph], a8[kAlph * kAlph]; short a16[kAlph * kAlph];
    int mn = 0, mnAln = 0;
    for (int i = 0; i < kAlph; i++) for (int j = 0; j < kAlph; j++) {
        s16[i * kAlph + j] = seed.sub[i][j];
        u8[i * kAlph + j] = (signed char) ung.sub[i][j];
        a8[i * kAlph + j] = (signed char) aln.sub[i][j];
        a16[i * kAlph + j] = aln.sub[i][j];
        mn = std::min<int>(mn, ung.sub[i][j]);
        mnAln = std::min<int>(mnAln, aln.sub[i][j]);
    }
    d.tab->ungMinAbs = std::abs(mn);
    d.tab->alnMinAbs = std::abs(mnAln);
    if (!upload(d.tab->seedSub, s16, kAlph * kAlph, d.stream, err)) return false;
    if (!upload(d.tab->seedBack, seed.pBack, kAlph, d.stream, err)) return false;
    if (!upload(d.tab->ungSub, u8, kAlph * kAlph, d.stream, err)) return false;
    if (!upload(d.tab->alnSub, a8, kAlph * kAlph, d.stream, err)) return false;
    if (!upload(d.tab->alnSub16, a16, kAlph * kAlph, d.stream, err)) return false;
    if (!upload(d.tab->alnBack, aln.pBack, kAlph, d.stream, err)) return false;
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

// This context searches the target set `o` has staged -- same device, same matrices: the tables and the set are shared, not copied (one context writes them,
// all read); whatever this context had staged before goes back when its last user lets go.  A later buildTargets() on either context detaches it again.
bool Gpu::shareFrom(const Gpu &o, std::string &err) {
    Impl &d = *d_;
    if (&o == this) return true;
    if (d.device != o.d_->device) { err = "shareFrom: the contexts are on different devices"; return false; }
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "shareFrom: a device-driven step is in flight"; return false; }
    if (d.stream) HIP_OK(hipStreamSynchronize(d.stream));
    if (o.d_->stream) HIP_OK(hipStreamSynchronize(o.d_->stream));      // what `o` has queued for its tables / targets is complete
    d.tab = o.d_->tab; d.tgt = o.d_->tgt;
    return true;
}

bool Gpu::uploadQueries(const uint8_t *res, const uint32_t *off, size_t nq, std::string &err) {
    Impl &d = *d_;
    if (!d.searcher) { d.searcher = true; g_searchers[d.device & 63].fetch_add(1, std::memory_order_relaxed); }
    nq_ = nq;
    d.qTotal = off[nq];
    d.hQoff.assign(off, off + nq + 1);
    d.maxQLen = 0; d.swBiasFresh = false;
    for (size_t i = 0; i < nq; i++) d.maxQLen = std::max(d.maxQLen, off[i + 1] - off[i]);
    if (!upload(d.qres, res, d.qTotal, d.stream, err)) return false;
    if (!upload(d.qoff, off, nq + 1, d.stream, err)) return false;
    if (!d.biasf.ensure(d.qTotal + 1, err) || !d.qthr.ensure(d.qTotal + 1, err) || !d.qcorr.ensure(d.qTotal + 1, err)) return false;
    if (!d.qbias.ensure(nq + 1, err) || !d.qself.ensure(nq + 1, err) || !d.qDiagBase.ensure(nq + 1, err) || !d.qDiagCnt.ensure(nq + 1, err) || !d.outCount.ensure(nq + 1, err) || !d.outBase.ensure(nq + 1, err) || !d.outOffset.ensure(nq + 2, err)) return false;
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

#define SP_DBG_SYNC(tag) do { HIP_OK(hipGetLastError()); if (dbg) { HIP_OK(hipStreamSynchronize(d.stream)); fprintf(stderr, "[sp] %s done\n", tag); } } while (0)

// k_prep, k_match, k_score for every resident query; leaves the scored double-hit diagonals in HBM (events 0..3)
// probeMode 1: the enumerating side of the probe-list cut (queries [qFrom, qTo) against the union of the shards' maps; the occupied probes leave by destination
// shard, nothing is matched or scored); probeMode 2: its fetching side (the probes in probeList / probeOff, placed by k_probe_place, instead of an enumeration)
bool Gpu::matchAndScore(int kmerThr, bool withOrd, std::string &err, int probeMode, size_t qFrom, size_t qTo) {
    Impl &d = *d_;
    overflowList_.clear();
    if (probeMode && d.maxQLen > 64) { err = "the probe-list cut serves queries of up to 64 residues (spacer peptides)"; return false; }
    const size_t nqRun = probeMode == 1 ? qTo - qFrom : nq_;      // queries this pass's matching kernel takes
    if (!d.tab->ext3.p || !d.tgt->pbits.p) { err = "tables/targets not resident"; return false; }
    if (d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }     // the previous pass's hit list is still being copied
    // persistent match grid: as many blocks per CU as the LDS footprint admits.  512 hits per query live in LDS; more go to
    // the per-wave HBM scratch and come back through the pre-filter (faster than a larger LDS buffer even for dense indices:
    // 45 vs 51 ms at 2 x 10^8 residues, because three blocks per CU stay resident instead of one).
    // footprint by index density (SP_MATCH_SMALL=0/1 overrides): sparse -> small, dense -> large; the sharded search always small
    // (its ordinal arrays push the large footprint to two blocks per CU)
    const bool small = d.knob.matchSmall >= 0 ? d.knob.matchSmall != 0 : (withOrd || !d.tgt->denseIndex);
    // short queries against a sparse index, unsharded: the grouped kernel (sp_gpu_match_grp.inc) -- same records, four queries per pass of a wave
    // (with ordinals too since round 5: the target-sharded search -- of SPARSE shards: the kernel, its list schedule and its 512 MB direct slot table are built and measured for a sparse
    // index; a dense shard with ordinals -- 2e8 residues over two ranks -- keeps k_match<256, true, true>)
    const bool grouped = probeMode ? true : (small && d.knob.matchGroup != 0 && d.maxQLen <= 64 && d.tgt->nTargets < ((size_t) 1 << 30) && !(withOrd && d.tgt->denseIndex));
    // dense index, unsharded, target ids below 2^24, on request (SP_MATCH_DENSE2=1 / set_tuning("match_dense2", 1)): k_match_dense (sp_gpu_match_dense.inc: 64-byte double buckets, a
    // 5-byte hit stream read back by LDS-DMA) -- round 6's decomposition: the same records, 41 % fewer bytes written, and 8 % SLOWER than k_match<512, false, false>
    const bool dense2 = !probeMode && !small && !withOrd && d.knob.matchDense2 != 0 && d.tgt->nTargets <= ((size_t) 1 << 24);
    const void *fn = probeMode == 1 ? (const void *) k_match_grp<true, 1> : probeMode == 2 ? (const void *) k_match_grp<true, 2> : grouped ? (withOrd ? (const void *) k_match_grp<true> : (const void *) k_match_grp<false>) :
                     small ? (withOrd ? (const void *) k_match<256, true, true> : (const void *) k_match<256, false, true>)
                           : (withOrd ? (const void *) k_match<512, true, false> : dense2 ? (const void *) k_match_dense : (const void *) k_match<512, false, false>);
    int perCU = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, fn, 64 * kWavesPerBlock, 0) != hipSuccess || perCU < 1) perCU = 2;
    if (d.matchPerCU > 0) perCU = std::min(perCU, d.matchPerCU);
    // Several search contexts on one GPU: the grouped kernel at TWO blocks per CU leaves half of every CU -- LDS, wave slots, registers -- to the other contexts' Smith-Waterman,
    // traceback and selection kernels and loses 13 % alone (0.533 against 0.468 ms; k_match loses 22 %): four host-driven contexts reach 23.4 M spacer-queries/s with two blocks,
    // 21.2 M with three, 20.6 M with four (profiles/r04_blocks_sweep.txt).  A context that searches alone takes all that fit.
    else if (d.matchPerCU == 0 && grouped && g_searchers[d.device & 63].load(std::memory_order_relaxed) >= 2) perCU = std::min(perCU, 2);
    if (d.knob.matchPerCU > 0) perCU = d.knob.matchPerCU;
    const bool dbg = d.knob.debug;
    if (dbg) fprintf(stderr, "[sp] k_match footprint=%s ord=%d blocks/CU=%d\n", grouped ? "grouped" : (small ? "small" : (dense2 ? "dense" : "large")), (int) withOrd, perCU);
    const uint32_t blocks = (uint32_t) std::max<size_t>(1, std::min<size_t>((nqRun + kWavesPerBlock - 1) / kWavesPerBlock, (size_t) d.numCU * perCU));
    d.matchBlocks = blocks;
    if (!d.spill.ensure((size_t) blocks * kWavesPerBlock * (dense2 ? std::max<size_t>(kSpillWordsPerWave, kDScrBytesPerWave / 8) : kSpillWordsPerWave), err)) return false;
    if (!d.queue.ensure(8, err) || !d.counters.ensure(16, err) || !d.heavyList.ensure((size_t) kHeavyLists * nq_ + 1, err) || !d.heavyCnt.ensure(kHeavyLists * kHeavyStride, err) || !d.heavyFlag.ensure(nq_ + 1, err) || !d.wavePartial.ensure((size_t) blocks * kWavesPerBlock * 4, err)) return false;
    // diagonal buffer: generous, HBM is plentiful (64 M records = 512 MB + side arrays)
    const size_t diagCap = std::max<size_t>((size_t) 1 << 22, std::min<size_t>((size_t) 1 << 26, nq_ * 64));
    d.diagCap = diagCap;
    if (!d.diagRec.ensure(diagCap, err) || !d.diagQuery.ensure(diagCap, err) || !d.diagScore.ensure(diagCap, err) || !d.diagFlag.ensure(diagCap, err)) return false;
    if (withOrd && (!d.diagOrd.ensure(diagCap, err) || !d.spillOrd.ensure((size_t) blocks * kWavesPerBlock * 65536, err))) return false;
    {   FillArgs fa{{d.queue.p, d.counters.p, d.heavyCnt.p, nullptr}, {8 * sizeof(uint32_t), 16 * sizeof(unsigned long long), kHeavyLists * kHeavyStride * sizeof(uint32_t), 0}, {0u, 0u, 0u, 0u}};
        hipLaunchKernelGGL(k_fill, dim3(1), dim3(256), 0, d.stream, fa); }
    SP_EV(0, 2);
    PrepArgs pa{d.qres.p, d.qoff.p, (uint32_t) nq_, d.tab->seedSub.p, d.tab->seedBack.p, d.tab->ungSub.p, kmerThr, d.tab->ungMinAbs, d.biasf.p, d.qthr.p, d.qcorr.p, d.qbias.p, d.qself.p};
    // SP_MATCH_HEAVY: similar k-mers from which a query counts as heavy and is started first (0 = none)
    // (worth its 30 us per 60 000 queries only while a wave gets few queries: with more than ~30 per wave the heavy ones no
    // longer decide when the kernel ends)
    // (the grouped kernel's list schedule wants the weights of ANY batch: 6 x 10^5 peptides -- 146 per wave -- still end 5 % apart, 2.98 -> 2.82 ms for 0.07 ms of weighing
    // and sorting)
    const uint32_t heavyThr = d.knob.matchHeavy >= 0 ? (uint32_t) d.knob.matchHeavy
                            : ((nq_ <= (size_t) blocks * kWavesPerBlock * 32 || ((grouped || !small) && d.knob.matchOrder != 0 && d.maxQLen <= 64 && !d.knob.prepSplit)) ? 4000u : 0u);
    // exact weights order the heavy queries better (k_match alone 0.43 instead of 0.47 ms on the bench workload) and cost k_prep8 two table rows per window -- 20 us alone, a
    // multiple beside other contexts' matching kernels, where the estimate wins: exact while this context is the only one searching on its device
    // (... while the heavy queries were LISTED by a threshold.  With the batch sorted by weight -- `ordered` below, the grouped kernel -- the estimate orders it well enough: 0.353 ms
    // + 0.044 prep against 0.365 + 0.064 with exact counts; exact there only on request)
    const bool orderedPossible = (grouped || !small) && heavyThr != 0 && d.knob.matchOrder != 0 && d.maxQLen <= 64 && !d.knob.prepSplit;      // (k_match_grp, and k_match on the dense index)
    const bool weighExact = d.knob.weighExact > 0 || (d.knob.weighExact < 0 && !orderedPossible && g_searchers[d.device & 63].load(std::memory_order_relaxed) <= 1);
    // the grouped kernel takes its queries in the order of decreasing weight (k_order_*) instead of "heavy ones first, then shares": needs the weights (k_prep8)
    const bool ordered = orderedPossible;
    if (probeMode == 1 && !ordered) { err = "the probe-list cut enumerates its slice under the list schedule only (SP_MATCH_ORDER=0, SP_MATCH_HEAVY=0 and SP_PREP_SPLIT switch it off: the heavy-query lists are laid out for the whole batch)"; return false; }
    const uint32_t orderTiles = (uint32_t) ((nqRun + kOrderTile - 1) / kOrderTile);      // (the enumerating side of the probe-list cut orders ITS slice: positions relative to qFrom)
    const size_t qOff = probeMode == 1 ? qFrom : 0;
    if (ordered && (!d.qweight.ensure(nq_ + 1, err) || !d.qorder.ensure(nq_ + 1, err) || !d.orderHist.ensure((size_t) orderTiles * kOrderClasses, err) || !d.orderHeader.ensure(16, err) ||
                    !d.orderQueues.ensure(kOrderQueues * kOrderQueueStride, err))) return false;
    WeighArgs wa{d.qres.p, d.qoff.p, (uint32_t) nq_, d.qthr.p, d.tab->ext3.p, d.tab->ext3cnt.p, d.tab->cntMin, ordered ? 0xFFFFFFFEu : (heavyThr ? heavyThr : 0xFFFFFFFFu), d.heavyList.p, d.heavyFlag.p, d.heavyCnt.p,
                 weighExact ? nullptr : d.tab->heavyEst.p, ordered ? d.qweight.p : nullptr};
    const unsigned weighBlocks = (unsigned) std::min<size_t>((nq_ + 31) / 32, (size_t) d.numCU * 8);
    if (d.maxQLen <= 64 && !d.knob.prepSplit) {
        const uint32_t stride = std::max<uint32_t>(16u, (d.maxQLen + 7u) & ~7u);      // staging row per query (<= 64)
        // the Smith-Waterman bias of the same queries rides along when the align matrix is resident (queueAlign() then skips its own launch)
        SwDev &sd = d.swd;
        const bool withSw = d.tab->alnSub16.p && d.tab->alnBack.p && sd.qcomp.ensure(d.qTotal + 1, err) && sd.qbias.ensure(nq_ + 1, err);
        Prep8Args p8{pa, wa, stride, withSw ? SwBiasArgs{d.tab->alnSub16.p, d.tab->alnBack.p, d.tab->alnMinAbs, sd.qcomp.p, sd.qbias.p} : SwBiasArgs{nullptr, nullptr, 0, nullptr, nullptr}};
        d.swBiasFresh = withSw;
        if (p8.w.est || !heavyThr) hipLaunchKernelGGL(k_prep8<false>, dim3(weighBlocks), dim3(256), (size_t) 32 * stride * 12, d.stream, p8);
        else hipLaunchKernelGGL(k_prep8<true>, dim3(weighBlocks), dim3(256), (size_t) 32 * stride * 12, d.stream, p8);
        SP_DBG_SYNC("k_prep8");
        if (ordered) {
            hipLaunchKernelGGL(k_order_count, dim3(orderTiles), dim3(256), 0, d.stream, d.qweight.p + qOff, (uint32_t) nqRun, d.orderHist.p);
            hipLaunchKernelGGL(k_order_scatter, dim3(orderTiles), dim3(256), 0, d.stream, d.qweight.p + qOff, (uint32_t) nqRun, d.orderHist.p, orderTiles, d.qorder.p, reinterpret_cast<OrderHeader *>(d.orderHeader.p),
                               d.orderQueues.p, blocks * (uint32_t) kWavesPerBlock / (uint32_t) kOrderQueues, small ? 0u : 1u, (small && nqRun <= (size_t) blocks * kWavesPerBlock * 32) ? 36u : 44u);      // (finer tickets on the sparse index: 0.353 -> 0.360 ms -- its groups of four queries want whole tickets)
            SP_DBG_SYNC("k_order");
        }
    } else {
        d.swBiasFresh = false;
        hipLaunchKernelGGL(k_prep, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, pa);
        SP_DBG_SYNC("k_prep");
        if (heavyThr) hipLaunchKernelGGL(k_weigh, dim3(weighBlocks), dim3(256), 0, d.stream, wa);
        else HIP_OK(hipMemsetAsync(d.heavyFlag.p, 0, nq_, d.stream));
        SP_DBG_SYNC("k_weigh");
    }
    SP_EV(1, 1);
    if (grouped && probeMode != 1) {      // (the slot words by k-mer slot: once per target set, on the first pass of the grouped kernel -- of whichever context sharing the set gets here first)
      std::lock_guard<std::mutex> shared(d.tgt->mu);
      if (!d.tgt->slotDirectValid) {
        if (!d.tgt->slotDirect.ensure((size_t) kSlots, err)) return false;
        hipLaunchKernelGGL(k_idx_slot_direct, dim3((unsigned) ((kSlots + 255) / 256)), dim3(256), 0, d.stream, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->slotDirect.p);
        HIP_OK(hipGetLastError());
        HIP_OK(hipStreamSynchronize(d.stream));      // (complete before another context's stream reads it)
        d.tgt->slotDirectValid = true;
      }
    }
    if (dense2) {      // (the double-bucket table: once per target set, on the first pass of k_match_dense -- of whichever context sharing the set gets here first)
      std::lock_guard<std::mutex> shared(d.tgt->mu);
      if (!d.tgt->buckets2Valid) {
        if (!d.tgt->buckets2.ensure(8 * (size_t) kSlots, err)) return false;
        hipLaunchKernelGGL(k_idx_buckets2, dim3((unsigned) ((kSlots + 255) / 256)), dim3(256), 0, d.stream, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->ientries.p, d.tgt->buckets2.p);
        HIP_OK(hipStreamSynchronize(d.stream));      // (complete before another context's stream reads it)
        d.tgt->buckets2Valid = true;
      }
    } else if (!small) {      // (the bucket table: once per target set, on the first pass with the large footprint -- of whichever context sharing the set gets here first)
      std::lock_guard<std::mutex> shared(d.tgt->mu);
      if (!d.tgt->bucketsValid) {
        if (!d.tgt->buckets.ensure(4 * (size_t) kSlots, err)) return false;
        hipLaunchKernelGGL(k_idx_buckets, dim3((unsigned) ((kSlots + 255) / 256)), dim3(256), 0, d.stream, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->ientries.p, d.tgt->buckets.p);
        HIP_OK(hipStreamSynchronize(d.stream));      // (complete before another context's stream reads it)
        d.tgt->bucketsValid = true;
      }
    }
    MatchArgs ma{d.qres.p, d.qoff.p, (uint32_t) nqRun, d.qthr.p, d.tab->ext3.p, d.tab->ext3cnt.p, d.tab->cntMin, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->ientries.p, (grouped && probeMode != 1) ? d.tgt->slotDirect.p : nullptr, small ? nullptr : (dense2 ? d.tgt->buckets2.p : d.tgt->buckets.p), d.spill.p, d.queue.p,
                 d.diagRec.p, d.diagQuery.p, (uint32_t) diagCap, d.queue.p + 1, d.qDiagBase.p, d.qDiagCnt.p, d.counters.p,
                 withOrd ? d.diagOrd.p : nullptr, withOrd ? d.spillOrd.p : nullptr,
                 (uint32_t) d.knob.matchStatic, (uint32_t) d.knob.matchDChunk,
                 d.heavyList.p, d.heavyFlag.p, d.heavyCnt.p, d.wavePartial.p,
                 ordered ? d.qorder.p : nullptr, ordered ? reinterpret_cast<const OrderHeader *>(d.orderHeader.p) : nullptr, ordered ? d.orderQueues.p : nullptr};
    if (probeMode == 1) {
        // capacity per destination: whole chunks per wave on top of a generous share of the slice's similar k-mers (a full buffer is counted as an overflow and reported)
        const size_t cap = std::max<size_t>((size_t) 1 << 20, nqRun * (size_t) std::max(64, d.knob.probePerQuery)) + (size_t) blocks * kWavesPerBlock * kProbeChunk;
        if (cap >= ((size_t) 1 << 31)) { err = "probe-list cut: the slice is too large for one pass (" + std::to_string(nqRun) + " queries)"; return false; }
        if (!d.probeCursor.ensure(kMaxProbeShards * kProbeCursorStride, err)) return false;
        for (uint32_t s = 0; s < d.nShardMaps; s++) {
            if (!d.pbRec[s].ensure(cap, err) || !d.pbQCnt[s].ensure(nqRun + 1, err)) return false;
            HIP_OK(hipMemsetAsync(d.pbQCnt[s].p, 0, (nqRun + 1) * sizeof(uint32_t), d.stream));      // (a query without a scanned window never reports)
        }
        d.probeCap = cap; d.probeSlice = nqRun;
        HIP_OK(hipMemsetAsync(d.probeCursor.p, 0, kMaxProbeShards * kProbeCursorStride * sizeof(uint32_t), d.stream));
        ma.pbits = d.unionMap.p; ma.prank = nullptr; ma.prow = nullptr; ma.slots = nullptr; ma.ientries = nullptr;
        ma.qFirst = (uint32_t) qFrom; ma.shardOwners = d.shardOwners.p; ma.nShards = d.nShardMaps; ma.probeCursor = d.probeCursor.p; ma.probeCap = (uint32_t) cap;
        for (uint32_t s = 0; s < d.nShardMaps; s++) { ma.probeRec[s] = d.pbRec[s].p; ma.probeQCnt[s] = d.pbQCnt[s].p; }
        hipLaunchKernelGGL((k_match_grp<true, 1>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
        SP_DBG_SYNC("k_match_grp (probes out)");
        SP_EV(2, 1);
        return true;
    }
    if (probeMode == 2) {
        ma.probeOff = d.probeOff.p; ma.probeIn = d.probeList.p;
        hipLaunchKernelGGL((k_match_grp<true, 2>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    }
    else if (grouped) { if (withOrd) hipLaunchKernelGGL(k_match_grp<true>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma); else hipLaunchKernelGGL(k_match_grp<false>, dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma); }
    else if (small) {
        if (withOrd) hipLaunchKernelGGL((k_match<256, true, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
        else hipLaunchKernelGGL((k_match<256, false, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    } else if (withOrd) hipLaunchKernelGGL((k_match<512, true, false>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    else if (dense2) hipLaunchKernelGGL(k_match_dense, dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    else hipLaunchKernelGGL((k_match<512, false, false>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    SP_DBG_SYNC("k_match");
    SP_EV(2, 1);
    ScoreArgs sa{d.diagRec.p, d.diagQuery.p, d.queue.p + 1, (uint32_t) diagCap, d.qres.p, d.qoff.p, d.qcorr.p, d.tgt->tmasked.p, d.tgt->toff.p, d.tab->ungSub.p, d.diagScore.p, d.counters.p, d.wavePartial.p, blocks * (uint32_t) kWavesPerBlock};
    hipLaunchKernelGGL(k_score, dim3((unsigned) d.numCU * 8), dim3(256), 0, d.stream, sa);
    SP_DBG_SYNC("k_score");
    SP_EV(3, 2);
    return true;
}

// counters + HIP-event times after the selection kernels were queued (event 4 recorded)
bool Gpu::finishPrefilter(bool matched, std::string &err) {
    Impl &d = *d_;
    // small results land in one page-locked mailbox (copies into pageable memory are staged one by one)
    if (!d.mailbox) d.mailbox = static_cast<unsigned long long *>(pinnedAlloc(64 * sizeof(unsigned long long), true));
    unsigned long long *c = d.mailbox; uint32_t *qv = reinterpret_cast<uint32_t *>(d.mailbox + 16); uint32_t &totalHits = *reinterpret_cast<uint32_t *>(d.mailbox + 24);
    totalHits = 0;
    {   // counters -> mailbox words 0..31, queue -> 32..39, the number of hits (the scan leaves it behind the last query's offset) -> 48
        PublishArgs pa{{d.counters.p, d.queue.p, d.outOffset.p + nq_, nullptr, nullptr}, {32, 8, 1, 0, 0}, {0, 32, 48, 0, 0}, reinterpret_cast<uint32_t *>(d.mailbox)};
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, d.stream, pa);
    }
    HIP_OK(spinSync(d.stream));
    if (matched) { counters_.kmers = c[0]; counters_.dbMatches = c[1]; counters_.diagResidues = c[3]; counters_.diagonals = c[5]; d.matchSpan[0] = ~c[6]; d.matchSpan[1] = c[7]; }
    counters_.overflowQueries = c[2]; counters_.hits = totalHits;
    d.hitsInFlight = false;
    d.nCand = qv[4];
    if (matched && d.knob.debug && d.matchBlocks) {      // how evenly the matching kernel's waves were busy
        std::vector<unsigned long long> wp((size_t) d.matchBlocks * kWavesPerBlock * 4);
        if (hipMemcpy(wp.data(), d.wavePartial.p, wp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
            double sum = 0, mx = 0; std::vector<double> v;
            for (size_t w = 0; w < wp.size() / 4; w++) { const double t = (double) wp[w * 4 + 3] / (double) d.wallKHz * 1e3; sum += t; mx = std::max(mx, t); v.push_back(t); }
            std::sort(v.begin(), v.end());
            fprintf(stderr, "[sp] k_match waves %zu: busy mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f us; kernel span %.1f us\n", v.size(), sum / (double) v.size(), v[v.size() / 2],
                    v[v.size() * 9 / 10], v[v.size() * 99 / 100], mx, (double) (c[7] - ~c[6]) / (double) d.wallKHz * 1e3);
            std::vector<size_t> idx(wp.size() / 4); for (size_t w = 0; w < idx.size(); w++) idx[w] = w;
            std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return wp[x * 4 + 3] > wp[y * 4 + 3]; });
            double km = 0; for (size_t w = 0; w < idx.size(); w++) km += (double) wp[w * 4];
            fprintf(stderr, "[sp]   busiest waves (us, similar k-mers, index hits; the mean wave has %.0f k-mers):", km / (double) idx.size());
            for (size_t k = 0; k < 6 && k < idx.size(); k++) fprintf(stderr, "  %.0f us %llu k %llu h", (double) wp[idx[k] * 4 + 3] / (double) d.wallKHz * 1e3, wp[idx[k] * 4], wp[idx[k] * 4 + 1]);
            fprintf(stderr, "\n");
        }
    }
#ifdef SP_GRP_STATS
    if (matched) fprintf(stderr, "[sp] grp stats: groups %llu hits %llu rounds %llu survivors %llu passes %llu passHits %llu | waves %llu mean busy %.1f us of span %.1f us\n", c[8], c[9], c[10], c[11], c[12], c[13], c[15],
                         c[15] ? (double) c[14] / (double) c[15] / (double) d.wallKHz * 1e3 : 0.0, (double) (c[7] - ~c[6]) / (double) d.wallKHz * 1e3);
#endif
#ifdef SP_PHASE_TIMERS
    if (matched) {
        unsigned long long tot = 0; for (int ph = 0; ph < 8; ph++) tot += c[8 + ph];
        fprintf(stderr, "[sp] k_match phase share:");
        for (int ph = 0; ph < 8; ph++) fprintf(stderr, " p%d=%.1f%%", ph, tot ? 100.0 * (double) c[8 + ph] / (double) tot : 0.0);
        fprintf(stderr, "  (ticks/wave %.0f)\n", (double) tot / (double) (d.matchBlocks * kWavesPerBlock));
    }
#endif
    if (matched) {
        times_.prep = evMs(d.timing, 2, d.ev[d.evSet][0], d.ev[d.evSet][1]);
        times_.match = evMs(d.timing, 1, d.ev[d.evSet][1], d.ev[d.evSet][2]);
        times_.matchActive = d.matchSpan[1] > d.matchSpan[0] ? (float) ((double) (d.matchSpan[1] - d.matchSpan[0]) / (double) d.wallKHz) : 0.f;
        times_.score = evMs(d.timing, 2, d.ev[d.evSet][2], d.ev[d.evSet][3]);
    }
    times_.select = evMs(d.timing, 2, d.ev[d.evSet][3], d.ev[d.evSet][4]);
    if (counters_.overflowQueries) {
        // error path only: name the first offending query (k_match marks them in qDiagBase)
        std::vector<uint32_t> base(nq_); std::string which;
        overflowList_.clear();
        if (hipMemcpy(base.data(), d.qDiagBase.p, nq_ * sizeof(uint32_t), hipMemcpyDeviceToHost) == hipSuccess)
            for (size_t q = 0; q < nq_; q++) if (base[q] == 0xFFFFFFFFu) overflowList_.push_back((uint32_t) q);
        if (!overflowList_.empty()) which = ", first the query at position " + std::to_string(overflowList_[0]) + " of the batch";
        err = std::to_string(counters_.overflowQueries) + " queries produced more index hits (limit 65535 per query), diagonals or candidates than the device buffers hold" + which +
              "; search such queries in a batch of their own against target shards (sp_ctx_set_targets_shard / sp_ctx_merge_candidates)";
        return false;
    }
    return true;
}

// selection of the scored diagonals + hits into query order (queued behind matchAndScore)
bool Gpu::queueSelect(unsigned binSize, int maxHits, int minDiagScore, void *hitsLeft, std::string &err) {
    Impl &d = *d_;
    const bool dbg = d.knob.debug;
    // a query's selected hits are staged where its diagonal records sit (they are at most as many): no separate layout pass
    // (... and a query has at most as many hits as diagonal records: the whole list never exceeds the record buffer -- nq x --max-seqs, 180 M entries = 3.6 GB in three buffers for
    // 100 000 spacers, is a bound no batch reaches)
    const size_t stageCap = d.diagCap;
    if (!d.outHits.ensure(std::min<size_t>(nq_ * (size_t) maxHits, d.diagCap), err) || !d.outStage.ensure(stageCap, err) || !d.outQuery.ensure(stageCap, err)) return false;
    SelectArgs se{(uint32_t) nq_, d.diagRec.p, d.diagScore.p, d.qDiagBase.p, d.qDiagCnt.p, d.qbias.p, d.qself.p, d.qoff.p, d.diagFlag.p, d.spill.p, d.qDiagBase.p,
                  binSize - 1, maxHits, minDiagScore, d.outStage.p, d.outQuery.p, d.outCount.p, d.outBase.p, d.counters.p, d.tgt->denseIndex ? (uint32_t) std::max(1, d.knob.selChunkDense) : (uint32_t) kSelChunk,
                  (d.tgt->denseIndex && d.knob.selDynamic) ? d.queue.p + 2 : nullptr};      // (queue words 2, 3 are zeroed with the rest at the start of the pass: matchAndScore's k_fill)      // (dense index: every query takes the whole-wave path -- small chunks deal them evenly)
    hipLaunchKernelGGL(k_select, dim3(d.matchBlocks), dim3(64 * kWavesPerBlock), 0, d.stream, se);
    if (hitsLeft) HIP_OK(hipStreamWaitEvent(d.stream, static_cast<hipEvent_t>(hitsLeft), 0));      // the previous step's hit list has left outOffset / outHits
    if (!d.scanCounts(d.outCount.p, (uint32_t) nq_, d.outOffset.p, err)) return false;
    hipLaunchKernelGGL(k_reorder_hits, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, d.outStage.p, d.outBase.p, d.outCount.p, d.outOffset.p, (uint32_t) nq_, d.outHits.p, d.outQuery.p);
    SP_DBG_SYNC("k_select");
    SP_EV(4, 2);
    return true;
}

bool Gpu::prefilter(int kmerThr, unsigned binSize, int maxHits, int minDiagScore, std::string &err) {
    Impl &d = *d_;
    if (nq_ == 0) return true;
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "prefilter: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    maxHits_ = maxHits;
    if (!matchAndScore(kmerThr, false, err)) return false;
    if (!queueSelect(binSize, maxHits, minDiagScore, nullptr, err)) return false;
    return finishPrefilter(true, err);      // the hit list travels when somebody asks for it (hostHits) or underneath the align kernels (align)
}

// ---- device-driven steps ----------------------------------------------------------------------------------------------------------------
// wait until everything queued on both streams is done (steps not yet collected keep their results and counters)
bool Gpu::drain(std::string &err) {
    Impl &d = *d_;
    HIP_OK(hipStreamSynchronize(d.stream));
    for (uint64_t j = d.stepsEnded; j < d.stepsBegun; j++) HIP_OK(d.issueLateCopies(d.step[j & 1]));      // (the steps nobody has collected: their results travel now -- the caller may restage queries next)
    HIP_OK(hipStreamSynchronize(d.copyStream));
    d.drained = true;
    d.evSet = (int) (d.stepsEnded & 1) ^ 1;      // the event set no uncollected step reads its times from
    return true;
}
size_t Gpu::stepsInFlight() const { return (size_t) (d_->stepsBegun - d_->stepsEnded); }
void Gpu::stepCapacityHint(size_t pairs) { d_->stepCap = std::max(d_->stepCap, pairs); }

bool Gpu::stepBegin(int kmerThr, unsigned binSize, int maxHits, int minDiagScore, int gapOpen, int gapExtend, const GateTable &gate, const StepOut &out, std::string &err) {
    Impl &d = *d_;
    SwDev &s = d.swd;
    if (d.stepsBegun - d.stepsEnded >= 2) { err = "stepBegin: two steps are in flight, collect one with stepEnd first"; return false; }
    if (nq_ == 0) { err = "stepBegin: no queries resident"; return false; }
    if (!out.hits || !out.offsets || !out.records || !out.cigars) { err = "stepBegin: result buffers missing"; return false; }
    if (!d.stepMail) { d.stepMail = static_cast<uint32_t *>(pinnedAlloc(2 * 64 * sizeof(uint32_t), true)); memset(d.stepMail, 0, 2 * 64 * sizeof(uint32_t)); }
    const int slot = (int) (d.stepsBegun & 1);
    Impl::StepSlot &ss = d.step[slot];
    d.drained = false;
    // capacity of this step: what earlier passes delivered plus a margin, bounded by the caller's buffers
    if (d.stepCap == 0) d.stepCap = counters_.hits ? (size_t) counters_.hits + (size_t) counters_.hits / 8 + 4096 : nq_ * 4 + 4096;
    // (... and by what the selection can deliver at all: nq x maxHits is what queueSelect() sizes the hit buffer for -- a small batch with a generous caller buffer
    // would otherwise copy past it)
    const size_t cap = std::max<size_t>(1, std::min<size_t>({d.stepCap, (size_t) out.hitCap, (size_t) out.recordCap, (size_t) 0x7FFFFFF0u, nq_ * (size_t) std::max(1, maxHits),
                                                              std::max<size_t>((size_t) 1 << 22, std::min<size_t>((size_t) 1 << 26, nq_ * 64))}));      // (... or the diagonal-record buffer, whichever is smaller: matchAndScore's diagCap)
    ss.cap = cap; ss.out = out;
    maxHits_ = maxHits;
    if (!d.stepWords.ensure(4, err)) return false;
    // the step in front: its hit list / records are still leaving buffers this one rewrites near its end -- the kernels that rewrite them wait
    // for those copies on the device (k_reorder_hits, k_aln_assemble), everything before them runs underneath the copies
    const bool behind = d.stepsBegun != d.stepsEnded;
    const Impl::StepSlot &prev = d.step[slot ^ 1];
    if (!behind) {
        if (d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }
        if (!waitAlignCopy(err)) return false;
    }
    // (round 6, default; SP_STEP_LATE_COPY=0 = the round-4 scheme: copies queued ahead on the copy stream behind events, the next step's kernels waiting for them.  Those
    // barrier packets sat in hardware queues that other contexts' streams share and held their kernels back: four device-driven contexts 15.3 M spacer-queries/s, with late
    // copies 24.0 M -- level with four host-driven contexts; profiles/r06_step_late_copy.txt)
    static const bool lateCopy = !(getenv("SP_STEP_LATE_COPY") != nullptr && *getenv("SP_STEP_LATE_COPY") == '0');
    ss.late = lateCopy;
    if (lateCopy) {      // this step writes the buffer set the step in front of it does not use: that one's results may still be on the device, waiting for its stepEnd()
        std::swap(d.outHits.p, d.outHitsAlt.p); std::swap(d.outHits.n, d.outHitsAlt.n); std::swap(d.outHits.bytes, d.outHitsAlt.bytes);
        std::swap(d.outOffset.p, d.outOffsetAlt.p); std::swap(d.outOffset.n, d.outOffsetAlt.n); std::swap(d.outOffset.bytes, d.outOffsetAlt.bytes);
        std::swap(s.rec.p, d.recAlt.p); std::swap(s.rec.n, d.recAlt.n); std::swap(s.rec.bytes, d.recAlt.bytes);
        std::swap(s.cig.p, d.cigAlt.p); std::swap(s.cig.n, d.cigAlt.n); std::swap(s.cig.bytes, d.cigAlt.bytes);
        if (!d.outOffset.ensure(nq_ + 2, err)) return false;
    }
    d.evSet = slot;
    if (!matchAndScore(kmerThr, false, err)) return false;
    if (!queueSelect(binSize, maxHits, minDiagScore, (behind && !lateCopy) ? prev.hitsOnHost : nullptr, err)) return false;
    {   StepCountArgs ca{d.outOffset.p + nq_, (uint32_t) cap, d.stepWords.p};
        hipLaunchKernelGGL(k_step_counts, dim3(1), dim3(64), 0, d.stream, ca); }
    HIP_OK(hipEventRecord(ss.selected, d.stream));
    if (lateCopy) {
        if (!queueAlign(cap, d.stepWords.p + 1, nullptr, gapOpen, gapExtend, gate, true, false, nullptr, err)) return false;
        {   PublishArgs pa{{d.counters.p, d.queue.p, d.stepWords.p, s.counters.p, s.cigOff.p + cap}, {32, 8, 4, 8, 1}, {0, 32, 40, 44, 52}, d.stepMail + 64 * slot};
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, d.stream, pa); }
        SP_EV(9, 2);
        HIP_OK(hipEventRecord(ss.assembled, d.stream));
        ss.copied = false; ss.nq = nq_; ss.dHits = d.outHits.p; ss.dOffsets = d.outOffset.p; ss.dRec = s.rec.p; ss.dCig = s.cig.p; ss.dCigN = s.cig.n;
        HIP_OK(hipGetLastError());
        d.stepsBegun++;
        return true;
    }
    // the hit list travels underneath the align kernels: offsets exactly, hit records up to the capacity
    HIP_OK(hipStreamWaitEvent(d.copyStream, ss.selected, 0));
    HIP_OK(hipMemcpyAsync(out.offsets, d.outOffset.p, (nq_ + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, d.copyStream));
    HIP_OK(hipMemcpyAsync(out.hits, d.outHits.p, cap * sizeof(GpuHit), hipMemcpyDeviceToHost, d.copyStream));
    HIP_OK(hipEventRecord(ss.hitsOnHost, d.copyStream));
    if (!queueAlign(cap, d.stepWords.p + 1, nullptr, gapOpen, gapExtend, gate, true, false, behind ? prev.alnOnHost : nullptr, err)) return false;
    // counters of the step -> its mailbox block (page-locked): prefilter counters [0, 32), work queues [32, 40), step words [40, 44), align counters [44, 52), backtrace bytes [52]
    {   PublishArgs pa{{d.counters.p, d.queue.p, d.stepWords.p, s.counters.p, s.cigOff.p + cap}, {32, 8, 4, 8, 1}, {0, 32, 40, 44, 52}, d.stepMail + 64 * slot};
        hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, d.stream, pa); }
    SP_EV(9, 2);
    HIP_OK(hipEventRecord(ss.assembled, d.stream));
    HIP_OK(hipStreamWaitEvent(d.copyStream, ss.assembled, 0));
    HIP_OK(hipMemcpyAsync(out.records, s.rec.p, cap * sizeof(AlnRec), hipMemcpyDeviceToHost, d.copyStream));
    const size_t cigBytes = (size_t) std::min<uint64_t>(out.cigarCap, (uint64_t) s.cig.n);
    if (cigBytes) HIP_OK(hipMemcpyAsync(out.cigars, s.cig.p, cigBytes, hipMemcpyDeviceToHost, d.copyStream));
    HIP_OK(hipEventRecord(ss.alnOnHost, d.copyStream));
    HIP_OK(hipGetLastError());
    d.stepsBegun++;
    return true;
}

bool Gpu::stepEnd(StepResult &r, std::string &err) {
    Impl &d = *d_;
    r = StepResult();
    if (d.stepsBegun == d.stepsEnded) { err = "stepEnd: no step in flight"; return false; }
    const int slot = (int) (d.stepsEnded & 1);
    Impl::StepSlot &ss = d.step[slot];
    auto poll = [&](hipEvent_t e) -> bool {      // short waits on the critical path: poll
        unsigned tries = 0;
        for (;;) { const hipError_t q = hipEventQuery(e); if (q == hipSuccess) return true; if (q != hipErrorNotReady) { err = std::string("stepEnd: ") + hipGetErrorString(q); d.stepsEnded++; return false; } pollPause(tries); }
    };
    if (ss.late) {      // the step's kernels are done: NOW its results travel (nothing was queued behind an event that had not happened)
        if (!poll(ss.assembled)) return false;
        HIP_OK(d.issueLateCopies(ss));
        if (Impl::copyEvents()) { HIP_OK(hipEventRecord(ss.alnOnHost, d.copyStream)); if (!poll(ss.alnOnHost)) return false; }
        else {      // (no marker behind the copies: the copy stream itself is asked -- see Impl::waitCopies)
            unsigned tries = 0;
            for (;;) { const hipError_t q = hipStreamQuery(d.copyStream); if (q == hipSuccess) break; if (q != hipErrorNotReady) { err = std::string("stepEnd: ") + hipGetErrorString(q); d.stepsEnded++; return false; } pollPause(tries); }
        }
    } else {
        if (!poll(ss.hitsOnHost) || !poll(ss.alnOnHost)) return false;
    }
    d.stepsEnded++;
    const uint32_t *m = d.stepMail + 64 * slot;
    const unsigned long long *c = reinterpret_cast<const unsigned long long *>(m);
    counters_.kmers = c[0]; counters_.dbMatches = c[1]; counters_.overflowQueries = c[2]; counters_.diagResidues = c[3]; counters_.diagonals = c[5];
    d.matchSpan[0] = ~c[6]; d.matchSpan[1] = c[7];
    const uint32_t nHits = m[40], nPairs = m[41], clipped = m[42], needHost = m[44 + 2], cigTotal = m[52];
    counters_.hits = nHits;
    d.nCand = m[32 + 4];
    d.sw.n = nPairs;
    alignRetries_ = 0;
    times_ = KernelTimes();
    times_.prep = evMs(d.timing, 2, d.ev[slot][0], d.ev[slot][1]);
    times_.match = evMs(d.timing, 1, d.ev[slot][1], d.ev[slot][2]);
    times_.matchActive = d.matchSpan[1] > d.matchSpan[0] ? (float) ((double) (d.matchSpan[1] - d.matchSpan[0]) / (double) d.wallKHz) : 0.f;
    times_.score = evMs(d.timing, 2, d.ev[slot][2], d.ev[slot][3]);
    times_.select = evMs(d.timing, 2, d.ev[slot][3], d.ev[slot][4]);
    times_.swForward = evMs(d.timing, 2, d.ev[slot][5], d.ev[slot][6]);
    times_.swReverse = evMs(d.timing, 2, d.ev[slot][6], d.ev[slot][7]);
    times_.traceback = evMs(d.timing, 2, d.ev[slot][7], d.ev[slot][8]);
    times_.assemble = evMs(d.timing, 2, d.ev[slot][8], d.ev[slot][9]);
    r.hits = nHits; r.pairs = nPairs; r.cigarUsed = cigTotal;
    d.stepCap = std::max<size_t>(d.stepCap, (size_t) nHits + (size_t) nHits / 8 + 4096);
    if (counters_.overflowQueries) {
        err = std::to_string(counters_.overflowQueries) + " queries produced more index hits (limit 65535 per query), diagonals or candidates than the device buffers hold; "
              "run sp_ctx_prefilter() on this batch for the list of such queries and search them against target shards (sp_ctx_set_targets_shard / sp_ctx_merge_candidates)";
        return false;
    }
    if (clipped) { r.repeat = true; r.why = "more hits (" + std::to_string(nHits) + ") than the step was sized for (" + std::to_string(ss.cap) + ")"; }
    else if (needHost) { r.repeat = true; r.why = std::to_string(needHost) + " alignments need the host-driven traceback"; }
    else if (cigTotal > ss.out.cigarCap) { r.repeat = true; r.why = "backtraces (" + std::to_string(cigTotal) + " bytes) exceed the caller's buffer"; }
    return true;
}

// sharded search, step 1: this shard's candidates (per query and target the best double-hit diagonal with score >= min)
bool Gpu::prefilterCandidates(int kmerThr, int minDiagScore, std::string &err) {
    Impl &d = *d_;
    d.nCand = 0; d.candFailed = false; d.candError.clear();
    if (nq_ == 0) return true;
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "prefilterCandidates: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    // (a failure here is rank-local -- one shard's queries beyond the hit-stream limit, say: it is remembered so that the collective behind this call, allgatherCandidates,
    // fails on EVERY rank instead of leaving the others in a receive)
    if (!matchAndScore(kmerThr, true, err) || !candidatesTail(minDiagScore, err)) { d.candFailed = true; d.candError = err; d.nCand = 0; return false; }
    return true;
}
bool Gpu::candidatesTail(int minDiagScore, std::string &err) {
    Impl &d = *d_;
    const bool dbg = d.knob.debug;
    if (!d.cand.ensure(d.diagCap, err)) return false;
    CandArgs ca{d.diagRec.p, d.diagQuery.p, d.diagScore.p, d.diagOrd.p, d.queue.p + 1, (uint32_t) d.diagCap, d.qDiagBase.p, d.qDiagCnt.p, d.qbias.p, d.qoff.p, minDiagScore,
                d.cand.p, d.queue.p + 4, (uint32_t) d.diagCap};
    hipLaunchKernelGGL(k_candidates, dim3((unsigned) d.numCU * 8), dim3(256), 0, d.stream, ca);
    SP_DBG_SYNC("k_candidates");
    SP_EV(4, 2);
    if (!finishPrefilter(true, err)) return false;
    counters_.hits = 0;
    return true;
}

static bool agreeOk(Comm &comm, bool localOk, std::string &err);
// ---- the probe-list cut (opt-in: SP_SHARD=probes in bench.py, sp_ctx_probe_* in the ABI) ------------------------------------------------------------------------
// Target-sharded search as above, but the similar-k-mer enumeration -- the same on every shard, and most of the matching kernel's time -- is done ONCE per query: rank r
// enumerates its slice of the queries against the union of the shards' occupancy maps and sends every occupied probe (query, ordinal, slot, window position: 12 bytes) to
// the shards whose own map has the slot; a shard fetches the lists of the probes it received and detects the double hits.  Candidates, their exchange and the merge are
// unchanged, and so are the results: a shard sees exactly the probes its own enumeration would have found occupied, in the same order, with the same ordinals.
bool Gpu::shardOccupancy(void *out, bool onDevice, std::string &err) {
    Impl &d = *d_;
    if (!d.tgt->pbits.p) { err = "shardOccupancy: no target set resident"; return false; }
    HIP_OK(hipStreamSynchronize(d.stream));
    HIP_OK(hipMemcpy(out, d.tgt->pbits.p, kShardMapBytes, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost));
    return true;
}

bool Gpu::probeSetShardMaps(const void *maps, int nShards, bool onDevice, std::string &err) {
    Impl &d = *d_;
    if (nShards < 1 || nShards > kMaxProbeShards) { err = "probeSetShardMaps: 1 .. " + std::to_string(kMaxProbeShards) + " shards"; return false; }
    const size_t words = kTableSize / 64;
    if (!d.shardMaps.ensure(words * (size_t) nShards, err) || !d.unionMap.ensure(words, err) || !d.shardOwners.ensure((size_t) kTableSize, err)) return false;
    if (maps != d.shardMaps.p) HIP_OK(hipMemcpyAsync(d.shardMaps.p, maps, kShardMapBytes * (size_t) nShards, onDevice ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, d.stream));
    hipLaunchKernelGGL(k_probe_union, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, d.stream, d.shardMaps.p, (uint32_t) nShards, (uint32_t) words, d.unionMap.p);
    hipLaunchKernelGGL(k_probe_owners, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, d.stream, d.shardMaps.p, (uint32_t) nShards, (uint32_t) words, d.shardOwners.p);
    HIP_OK(hipGetLastError());
    HIP_OK(hipStreamSynchronize(d.stream));
    d.nShardMaps = (uint32_t) nShards;
    return true;
}

bool Gpu::probeEnumerate(int kmerThr, size_t qFrom, size_t qTo, uint64_t *counts, std::string &err) {
    Impl &d = *d_;
    if (!d.nShardMaps) { err = "probeEnumerate: the shards' occupancy maps are not set (probeSetShardMaps)"; return false; }
    if (qFrom > qTo || qTo > nq_) { err = "probeEnumerate: the slice ends behind the last query"; return false; }
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "probeEnumerate: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    for (int s = 0; s < kMaxProbeShards; s++) d.probeCnt[s] = 0;
    if (qTo > qFrom) {
        if (!matchAndScore(kmerThr, true, err, 1, qFrom, qTo)) return false;
        unsigned long long over = 0;
        HIP_OK(hipMemcpy2DAsync(d.probeCnt, sizeof(uint32_t), d.probeCursor.p, kProbeCursorStride * sizeof(uint32_t), sizeof(uint32_t), kMaxProbeShards, hipMemcpyDeviceToHost, d.stream));
        HIP_OK(hipMemcpyAsync(&over, d.counters.p + 2, sizeof(over), hipMemcpyDeviceToHost, d.stream));
        HIP_OK(hipStreamSynchronize(d.stream));
        times_.prep = evMs(d.timing, 2, d.ev[d.evSet][0], d.ev[d.evSet][1]);
        times_.match = evMs(d.timing, 1, d.ev[d.evSet][1], d.ev[d.evSet][2]);
        if (d.knob.debug) {      // how evenly the enumerating kernel's waves were busy
            std::vector<unsigned long long> wp((size_t) d.matchBlocks * kWavesPerBlock * 4);
            if (hipMemcpy(wp.data(), d.wavePartial.p, wp.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess) {
                std::vector<double> v; double sum = 0;
                for (size_t w = 0; w < wp.size() / 4; w++) { const double t = (double) wp[w * 4 + 3] / (double) d.wallKHz * 1e3; sum += t; v.push_back(t); }
                std::sort(v.begin(), v.end());
                fprintf(stderr, "[sp] probe enumeration, waves %zu: busy mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f us\n", v.size(), sum / (double) v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back());
            }
        }
        if (over) { err = "probeEnumerate: more occupied probes than the buffers hold (" + std::to_string(d.probeCap) + " per shard; SP_PROBE_PER_QUERY raises the room)"; return false; }
    }
    if (counts) for (uint32_t s = 0; s < d.nShardMaps; s++) counts[s] = d.probeCnt[s];
    return true;
}
size_t Gpu::probeCount(int dest) const { return dest >= 0 && dest < kMaxProbeShards ? d_->probeCnt[dest] : 0; }
size_t Gpu::probeSliceQueries() const { return d_->probeSlice; }
const void *Gpu::probeRecordsDevice(int dest) const { return d_->pbRec[dest].p; }
const uint32_t *Gpu::probeQueryCountsDevice(int dest) const { return d_->pbQCnt[dest].p; }

// the probes addressed to this shard -- the enumerating ranks' lists back to back, in any order of the ranks, and the number of probes each query of the batch has among
// them (the ranks' per-query counts in the order of their slices) -- -> this shard's candidates (as prefilterCandidates leaves them)
bool Gpu::probeFetch(const void *records, size_t n, const uint32_t *counts, bool onDevice, int kmerThr, int minDiagScore, std::string &err) {
    Impl &d = *d_;
    d.nCand = 0;
    if (nq_ == 0) return true;
    if (n >= ((size_t) 1 << 31)) { err = "probeFetch: more than 2^31 probes in one pass"; return false; }
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "probeFetch: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    if (!d.pinRec.ensure(n + 1, err) || !d.pinCnt.ensure(nq_ + 2, err) || !d.probeList.ensure(n + 1, err) || !d.probeOff.ensure(nq_ + 2, err) || !d.counters.ensure(16, err)) return false;
    const uint4 *rIn = static_cast<const uint4 *>(records); const uint32_t *cIn = counts;
    if (!onDevice) {
        if (n) HIP_OK(hipMemcpyAsync(d.pinRec.p, records, n * sizeof(uint4), hipMemcpyHostToDevice, d.stream));
        HIP_OK(hipMemcpyAsync(d.pinCnt.p, counts, nq_ * sizeof(uint32_t), hipMemcpyHostToDevice, d.stream));
        rIn = d.pinRec.p; cIn = d.pinCnt.p;
    }
    if (!d.scanCounts(cIn, (uint32_t) nq_, d.probeOff.p, err)) return false;
    HIP_OK(hipMemsetAsync(d.counters.p + 14, 0, 2 * sizeof(unsigned long long), d.stream));      // ([14] records placed, [15] records out of place)
    if (n) hipLaunchKernelGGL(k_probe_place, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, rIn, (uint32_t) n, (uint32_t) nq_, d.probeOff.p, d.probeList.p, d.counters.p + 15);
    HIP_OK(hipGetLastError());
    unsigned long long pb[2] = {0, 0}; uint32_t total = 0;
    HIP_OK(hipMemcpyAsync(pb, d.counters.p + 14, sizeof(pb), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipMemcpyAsync(&total, d.probeOff.p + nq_, sizeof(total), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
    const unsigned long long bad = pb[1] + (pb[0] != total ? 1ull : 0ull);      // (every position of every list must have received its record: a hole would be fetched as whatever the buffer held)
    if (bad || total > n) { err = "probeFetch: the probe records do not fit the per-query counts (" + std::to_string(bad) + " records out of place; lists and counts of different passes?)"; return false; }
    d.candFailed = false; d.candError.clear();
    if (!matchAndScore(kmerThr, true, err, 2) || !candidatesTail(minDiagScore, err)) { d.candFailed = true; d.candError = err; d.nCand = 0; return false; }
    return true;
}

// collective: rank r of `comm` holds shard r and enumerates queries [qFrom, qTo) (the ranks' slices tile the batch); leaves this shard's candidates
bool Gpu::probeSearch(Comm &comm, int kmerThr, int minDiagScore, size_t qFrom, size_t qTo, double *exchangeMs, std::string &err) {
    Impl &d = *d_;
    const int world = comm.world(), me = comm.rank();
    if (world > kMaxProbeShards) { err = "probeSearch: at most " + std::to_string(kMaxProbeShards) + " ranks"; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    double xMs = 0;
    // once per target set: everybody's occupancy map.  The ranks decide TOGETHER (a rank that alone thought its maps were stale would wait in an all-gather nobody else enters)
    bool needMaps = d.nShardMaps != (uint32_t) world || d.probeMapsSerial != d.tgt->buildSerial || !d.tgt->buildSerial;
    {   unsigned char mineFlag = needMaps ? 1 : 0; std::vector<unsigned char> flags((size_t) world, 0);
        if (!comm.allgather(&mineFlag, 1, flags.data(), false, err)) return false;
        for (unsigned char f : flags) needMaps = needMaps || f != 0; }
    if (needMaps) {
        const size_t words = kTableSize / 64;
        bool ok = d.tgt->pbits.p && d.shardMaps.ensure(words * (size_t) world, err) && d.unionMap.ensure(words, err);
        if (ok && hipStreamSynchronize(d.stream) != hipSuccess) { ok = false; err = "probeSearch: the search stream failed"; }
        if (!agreeOk(comm, ok, err)) return false;
        if (!comm.allgather(d.tgt->pbits.p, kShardMapBytes, d.shardMaps.p, true, err)) return false;
        if (!agreeOk(comm, probeSetShardMaps(d.shardMaps.p, world, true, err), err)) return false;
        d.probeMapsSerial = d.tgt->buildSerial;
    }
    xMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    uint64_t mine[kMaxProbeShards + 2] = {};      // records per destination, then this rank's slice
    const bool sliceOk = qFrom <= qTo && qTo <= nq_;
    std::string lerr;
    bool ok = sliceOk && probeEnumerate(kmerThr, qFrom, qTo, mine, lerr);
    if (!sliceOk) lerr = "probeSearch: the slice ends behind the last query";
    if (!ok) err = lerr;
    mine[kMaxProbeShards] = qFrom; mine[kMaxProbeShards + 1] = qTo;
    if (!agreeOk(comm, ok, err)) return false;
    const auto t1 = std::chrono::steady_clock::now();
    constexpr size_t kRow = kMaxProbeShards + 2;
    std::vector<uint64_t> all((size_t) world * kRow, 0);
    if (!comm.allgather(mine, sizeof(mine), all.data(), false, err)) return false;
    size_t recvTotal = 0, covered = 0; std::vector<size_t> recvCnt((size_t) world), sendCnt((size_t) world);
    bool tiles = true;
    for (int r = 0; r < world; r++) {
        recvCnt[(size_t) r] = (size_t) all[(size_t) r * kRow + (size_t) me]; recvTotal += recvCnt[(size_t) r]; sendCnt[(size_t) r] = (size_t) mine[r];
        const uint64_t f = all[(size_t) r * kRow + kMaxProbeShards], t = all[(size_t) r * kRow + kMaxProbeShards + 1];
        covered += (size_t) (t - f);
        for (int o = 0; o < r; o++) { const uint64_t f2 = all[(size_t) o * kRow + kMaxProbeShards], t2 = all[(size_t) o * kRow + kMaxProbeShards + 1]; if (f < t && f2 < t2 && f < t2 && f2 < t) tiles = false; }
    }
    if (!tiles || covered != nq_) { err = "probeSearch: the ranks' query slices do not tile the batch"; return false; }      // (every rank sees the same table)
    ok = recvTotal < ((size_t) 1 << 31) && d.pxRec.ensure(recvTotal + 1, err) && d.pinCnt.ensure(nq_ + 2, err);
    if (ok && recvTotal >= ((size_t) 1 << 31)) { ok = false; err = "probeSearch: more than 2^31 probes for one shard"; }
    if (!agreeOk(comm, ok, err)) return false;
    {   std::vector<const void *> sp((size_t) world); std::vector<void *> rp((size_t) world); std::vector<size_t> sb((size_t) world), rb((size_t) world);
        size_t at = 0;
        for (int r = 0; r < world; r++) { sp[(size_t) r] = d.pbRec[r].p; sb[(size_t) r] = sendCnt[(size_t) r] * sizeof(uint4); rp[(size_t) r] = d.pxRec.p + at; rb[(size_t) r] = recvCnt[(size_t) r] * sizeof(uint4); at += recvCnt[(size_t) r]; }
        if (!comm.alltoallv(sp.data(), sb.data(), rp.data(), rb.data(), true, err)) return false;
        // the per-query counts: rank r's cover its slice, they land at the slice's place in the batch
        for (int r = 0; r < world; r++) {
            const uint64_t f = all[(size_t) r * kRow + kMaxProbeShards], t = all[(size_t) r * kRow + kMaxProbeShards + 1];
            sp[(size_t) r] = d.pbQCnt[r].p; sb[(size_t) r] = (size_t) (qTo - qFrom) * sizeof(uint32_t); rp[(size_t) r] = d.pinCnt.p + f; rb[(size_t) r] = (size_t) (t - f) * sizeof(uint32_t);
        }
        if (!comm.alltoallv(sp.data(), sb.data(), rp.data(), rb.data(), true, err)) return false;
    }
    xMs += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count();
    if (exchangeMs) *exchangeMs = xMs;
    ok = probeFetch(d.pxRec.p, recvTotal, d.pinCnt.p, true, kmerThr, minDiagScore, lerr);
    if (!ok) err = lerr;
    return agreeOk(comm, ok, err);
}

const Candidate *Gpu::candidatesDevice() const { return d_->cand.p; }
size_t Gpu::numCandidates() const { return d_->nCand; }

// Target-sharded search: every rank has matched ALL queries against ITS target shard (prefilterCandidates); the shards' candidate lists are exchanged -- counts over the
// bootstrap sockets, the lists themselves padded to the longest one by RCCL all-gather, HBM to HBM over xGMI -- and land in candUnion, where mergeCandidates() reads them
// (rows of 0xFF are padding).  The reference merges the per-split result FILES of its MPI ranks (prefiltering/Prefiltering.cpp:258-362).
// The ranks of a job fail TOGETHER: a condition only one rank can see (a slice out of range, a buffer that cannot grow) is exchanged before the next collective, so that nobody
// is left waiting in an all-gather for a rank that has already returned an error.
static bool agreeOk(Comm &comm, bool localOk, std::string &err) {
    unsigned char mine = localOk ? 1 : 0; std::vector<unsigned char> all((size_t) comm.world(), 0);
    std::string xerr;
    if (!comm.allgather(&mine, 1, all.data(), false, xerr)) { if (localOk) err = xerr; return false; }
    for (int r = 0; r < comm.world(); r++) if (!all[(size_t) r]) { if (localOk) err = "rank " + std::to_string(r) + " of the exchange failed before the collective (see its own error)"; return false; }
    return true;
}

bool Gpu::allgatherCandidates(Comm &comm, size_t &unionRows, std::string &err) {
    Impl &d = *d_;
    const int world = comm.world();
    uint64_t mine = d.candFailed ? ~0ull : (uint64_t) d.nCand; std::vector<uint64_t> all((size_t) world, 0);
    if (!comm.allgather(&mine, sizeof(mine), all.data(), false, err)) return false;
    for (int r = 0; r < world; r++) if (all[(size_t) r] == ~0ull) {      // (every rank sees the same numbers: all of them stop here, together)
        err = "allgatherCandidates: the candidate pass of rank " + std::to_string(r) + " failed" + (d.candFailed ? ": " + d.candError : std::string(" (see its own error); no rank holds a complete candidate list"));
        return false;
    }
    uint64_t longest = 1; for (uint64_t x : all) longest = std::max(longest, x);
    if (longest * (uint64_t) world >= (1ull << 32)) { err = "allgatherCandidates: more than 2^32 candidate rows"; return false; }      // (every rank sees the same numbers)
    bool ok = d.candPad.ensure((size_t) longest, err) && d.candUnion.ensure((size_t) longest * world + 1, err);
    if (ok) {
        ok = hipMemsetAsync(d.candPad.p, 0xFF, (size_t) longest * sizeof(Candidate), d.stream) == hipSuccess &&
             (!d.nCand || hipMemcpyAsync(d.candPad.p, d.cand.p, (size_t) d.nCand * sizeof(Candidate), hipMemcpyDeviceToDevice, d.stream) == hipSuccess) &&
             hipStreamSynchronize(d.stream) == hipSuccess;
        if (!ok) err = "allgatherCandidates: staging the candidate rows failed";
    }
    if (!agreeOk(comm, ok, err)) return false;
    if (!comm.allgather(d.candPad.p, (size_t) longest * sizeof(Candidate), d.candUnion.p, true, err)) return false;
    unionRows = (size_t) longest * world;
    return true;
}
const Candidate *Gpu::candidateUnionDevice() const { return d_->candUnion.p; }

// Query-partitioned search: every rank has searched ITS slice of the queries against the whole target set; all ranks receive all hit lists in global query order.
// counts[q] (hits of global query q) and the 8-byte hit records are gathered by two padded all-gathers of device buffers; the caller gets them in host memory.
bool Gpu::allgatherHits(Comm &comm, size_t queryOffset, size_t nqTotal, const GpuHit *srcHits, const uint32_t *srcOffsets, GpuHit *hits, uint64_t *offsets, size_t hitCap, size_t &nHits, std::string &err) {
    Impl &d = *d_;
    const int world = comm.world();
    const bool sliceOk = queryOffset + nq_ <= nqTotal;      // (travels in the meta record: the others must not wait for a rank that stops here)
    // srcHits / srcOffsets: a delivered step's hit list in host memory (the context may already be searching the next batch); nullptr: the last pass's, still on the device
    const uint64_t myHits = srcOffsets ? (uint64_t) srcOffsets[nq_] : (uint64_t) counters_.hits;
    uint64_t mine[3] = {sliceOk ? (uint64_t) queryOffset : ~0ull, (uint64_t) nq_, myHits}; std::vector<uint64_t> all((size_t) world * 3, 0);
    if (!comm.allgather(mine, sizeof(mine), all.data(), false, err)) return false;
    for (int r = 0; r < world; r++) if (all[(size_t) r * 3] == ~0ull) { err = "allgatherHits: the slice of rank " + std::to_string(r) + " ends behind the last query"; return false; }
    uint64_t maxQ = 1, maxH = 1, totalH = 0, totalQ = 0;
    for (int r = 0; r < world; r++) { maxQ = std::max(maxQ, all[(size_t) r * 3 + 1]); maxH = std::max(maxH, all[(size_t) r * 3 + 2]); totalH += all[(size_t) r * 3 + 2]; totalQ += all[(size_t) r * 3 + 1]; }
    if (totalQ != nqTotal) { err = "allgatherHits: the ranks' slices hold " + std::to_string(totalQ) + " queries, not " + std::to_string(nqTotal); return false; }
    const bool roomOk = totalH <= hitCap;      // (the caller's buffer: a rank-local condition, agreed on below)
    if (!roomOk) err = "allgatherHits: " + std::to_string(totalH) + " hits, room for " + std::to_string(hitCap);
    if (!srcOffsets && d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }
    bool ok = roomOk && d.xOff.ensure((size_t) maxQ + 1, err) && d.xOffAll.ensure((size_t) (maxQ + 1) * world, err) && d.xHit.ensure((size_t) maxH, err) && d.xHitAll.ensure((size_t) maxH * world, err);
    hipStream_t xs = srcOffsets ? d.copyStream : d.stream;      // (a host source has nothing to wait for on the search stream)
    if (ok) {
        ok = hipMemsetAsync(d.xOff.p, 0, (size_t) (maxQ + 1) * sizeof(uint32_t), xs) == hipSuccess &&
             (!nq_ || hipMemcpyAsync(d.xOff.p, srcOffsets ? (const void *) srcOffsets : (const void *) d.outOffset.p, (nq_ + 1) * sizeof(uint32_t), hipMemcpyDefault, xs) == hipSuccess) &&
             (!myHits || hipMemcpyAsync(d.xHit.p, srcOffsets ? (const void *) srcHits : (const void *) d.outHits.p, (size_t) myHits * sizeof(GpuHit), hipMemcpyDefault, xs) == hipSuccess) &&
             hipStreamSynchronize(xs) == hipSuccess;
        if (!ok) err = "allgatherHits: staging the hit list failed";
    }
    if (!agreeOk(comm, ok, err)) return false;
    if (!comm.allgather(d.xOff.p, (size_t) (maxQ + 1) * sizeof(uint32_t), d.xOffAll.p, true, err)) return false;
    if (!comm.allgather(d.xHit.p, (size_t) maxH * sizeof(GpuHit), d.xHitAll.p, true, err)) return false;
    std::vector<uint32_t> offAll((size_t) (maxQ + 1) * world);
    HIP_OK(hipMemcpy(offAll.data(), d.xOffAll.p, offAll.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    // the slices in the order of their first query
    std::vector<int> order((size_t) world); for (int r = 0; r < world; r++) order[(size_t) r] = r;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return all[(size_t) x * 3] != all[(size_t) y * 3] ? all[(size_t) x * 3] < all[(size_t) y * 3] : x < y; });
    uint64_t at = 0, q = 0;
    for (int r : order) {
        const uint64_t qo = all[(size_t) r * 3], nq = all[(size_t) r * 3 + 1], nh = all[(size_t) r * 3 + 2];
        if (nq && qo != q) { err = "allgatherHits: the ranks' slices do not tile the queries"; return false; }
        const uint32_t *off = offAll.data() + (size_t) r * (maxQ + 1);
        for (uint64_t i = 0; i < nq; i++) offsets[q + i] = at + off[i];
        if (nh) HIP_OK(hipMemcpy(hits + at, d.xHitAll.p + (size_t) r * maxH, (size_t) nh * sizeof(GpuHit), hipMemcpyDeviceToHost));
        at += nh; q += nq;
    }
    offsets[nqTotal] = at;
    nHits = (size_t) at;
    return true;
}

// ---- result-DB text on the device (sp_gpu_text.inc) ---------------------------------------------------------------------------------------------------------
// Entry lengths, offsets and the text total of the device-side formatters are 32-bit.  A batch whose text would reach 4 GiB (2^20 queries x 300 hits of a saturated
// dense DB: 4.7 GB of prefilter lines, several times that in alignment lines) is DECLINED before anything is written -- the caller's host formatter takes it -- instead of
// wrapping the scan.  SP_TEXT_LIMIT (bytes) lowers the bound for tests.
static uint64_t deviceTextLimit() {
    static const uint64_t lim = [] { const char *e = getenv("SP_TEXT_LIMIT"); const uint64_t v = e ? strtoull(e, nullptr, 10) : 0; return v ? v : (1ull << 32) - (1ull << 20); }();
    return lim;
}
// Device -> host for a destination that may be ORDINARY memory (a one-shot process does not page-lock its result buffers: that costs 0.23 ms per MB).  The runtime's own
// staging of such a copy took anything between 40 ms and 2.3 s for 170 MB on the pool's boxes; here the bytes go through two page-locked 8 MB buffers of the context,
// the copy of chunk i + 1 running while the CPU moves chunk i to its place.  Page-locked destinations take the plain asynchronous copy.
bool Gpu::Impl::copyOut(void *dst, const void *srcDev, size_t bytes, std::string &err) {
    if (bytes == 0) return true;
    hipPointerAttribute_t at;
    const bool pinnedDst = hipPointerGetAttributes(&at, dst) == hipSuccess && at.type == hipMemoryTypeHost;
    (void) hipGetLastError();      // (an unregistered pointer is an "error" of the query, not of the stream)
    if (pinnedDst) { HIP_OK(hipMemcpyAsync(dst, srcDev, bytes, hipMemcpyDeviceToHost, stream)); return true; }
    constexpr size_t kChunk = 8u << 20;
    if (!ring[0]) { ring[0] = static_cast<char *>(pinnedAlloc(kChunk)); ring[1] = static_cast<char *>(pinnedAlloc(kChunk)); HIP_OK(hipEventCreateWithFlags(&ringEv[0], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&ringEv[1], hipEventDisableTiming)); }
    const size_t nChunks = (bytes + kChunk - 1) / kChunk;
    auto issue = [&](size_t c) -> hipError_t {
        const size_t off = c * kChunk, n = std::min(kChunk, bytes - off);
        hipError_t e = hipMemcpyAsync(ring[c & 1], static_cast<const char *>(srcDev) + off, n, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipEventRecord(ringEv[c & 1], stream);
    };
    HIP_OK(issue(0));
    for (size_t c = 0; c < nChunks; c++) {
        if (c + 1 < nChunks) HIP_OK(issue(c + 1));
        HIP_OK(hipEventSynchronize(ringEv[c & 1]));
        const size_t off = c * kChunk, n = std::min(kChunk, bytes - off);
        memcpy(static_cast<char *>(dst) + off, ring[c & 1], n);
    }
    return true;
}

bool Gpu::hitsText(char *text, uint64_t cap, uint64_t &used, uint32_t *entryLen, bool &declined, std::string &err) {
    Impl &d = *d_;
    used = 0; declined = false;
    if (nq_ == 0) return true;
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "hitsText: a device-driven step is in flight"; return false; }
    const size_t nHits = (size_t) counters_.hits;
    if (!d.txLen.ensure(nq_ + 1, err) || !d.txOff.ensure(nq_ + 2, err) || !d.txLineOff.ensure(nHits + 1, err)) return false;
    HitTextArgs a{d.outHits.p, d.outOffset.p, d.outQuery.p, (uint32_t) nq_, d.outOffset.p + nq_, d.tgt->tkeys.p, d.txLen.p, d.txOff.p, d.txLineOff.p, nullptr};
    if (nHits) hipLaunchKernelGGL(k_hits_text_linelen, dim3((unsigned) ((nHits + 255) / 256)), dim3(256), 0, d.stream, a);
    hipLaunchKernelGGL(k_hits_text_len, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, a);
    if (!d.txTotal.ensure(2, err)) return false;
    HIP_OK(hipMemsetAsync(d.txTotal.p, 0, sizeof(unsigned long long), d.stream));
    hipLaunchKernelGGL(k_sum_u64, dim3((unsigned) std::min<size_t>((nq_ + 255) / 256, 1024)), dim3(256), 0, d.stream, d.txLen.p, (uint32_t) nq_, d.txTotal.p);
    if (!d.scanCounts(d.txLen.p, (uint32_t) nq_, d.txOff.p, err)) return false;
    uint32_t total = 0; unsigned long long total64 = 0;
    HIP_OK(hipMemcpyAsync(&total, d.txOff.p + nq_, sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipMemcpyAsync(&total64, d.txTotal.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
    if (total64 >= deviceTextLimit()) { declined = true; return true; }
    used = total;
    if (used > cap) return true;
    if (!d.txText.ensure((size_t) total + 16, err)) return false;
    a.text = d.txText.p;
    if (nHits) hipLaunchKernelGGL(k_hits_text_write, dim3((unsigned) ((nHits + 255) / 256)), dim3(256), 0, d.stream, a);
    hipLaunchKernelGGL(k_text_entry_ends, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, d.txLen.p, d.txOff.p, (uint32_t) nq_, d.txText.p);
    if (!d.copyOut(text, d.txText.p, total, err) || !d.copyOut(entryLen, d.txLen.p, nq_ * sizeof(uint32_t), err)) return false;
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

bool Gpu::textIndex(const uint32_t *keys, size_t n, uint64_t base, char *text, uint64_t cap, uint64_t &used, std::string &err) {
    Impl &d = *d_;
    used = 0;
    if (n == 0) return true;
    if (n != nq_ || !d.txLen.p || !d.txOff.p) { err = "textIndex: no entries of this batch on the device (call hitsText / alignHitsText first)"; return false; }
    if ((uint64_t) n * 48ull >= deviceTextLimit()) { err = "textIndex: the index text of " + std::to_string(n) + " entries would not fit 32-bit offsets"; return false; }      // (key + offset + length + separators: at most 44 bytes a line)
    if (!upload(d.txKeys, keys, n, d.stream, err) || !d.txIdxLen.ensure(n + 1, err) || !d.txIdxOff.ensure(n + 2, err)) return false;
    IndexTextArgs a{d.txKeys.p, d.txLen.p, d.txOff.p, (uint32_t) n, base, d.txIdxLen.p, d.txIdxOff.p, nullptr};
    hipLaunchKernelGGL(k_index_text_len, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, a);
    if (!d.scanCounts(d.txIdxLen.p, (uint32_t) n, d.txIdxOff.p, err)) return false;
    uint32_t total = 0;
    HIP_OK(hipMemcpyAsync(&total, d.txIdxOff.p + n, sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
    used = total;
    if (used > cap) return true;
    if (!d.txIdxText.ensure((size_t) total + 16, err)) return false;
    a.text = d.txIdxText.p;
    hipLaunchKernelGGL(k_index_text_write, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, a);
    if (!d.copyOut(text, d.txIdxText.p, total, err)) return false;
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

bool Gpu::alignHitsText(int gapOpen, int gapExtend, const GateTable &gate, AlignHost &host, double minSeqId, uint32_t minAlnLen, char *text, uint64_t cap, uint64_t &used, uint32_t *entryLen,
                        bool &fellBack, std::string &err) {
    Impl &d = *d_;
    SwDev &s = d.swd;
    used = 0; fellBack = false;
    if (nq_ == 0) return true;
    if (gate.evalText.empty()) { fellBack = true; return true; }
    AlignDirect direct; direct.deviceOnly = true;
    if (!align(nullptr, 0, gapOpen, gapExtend, gate, host, &direct, err)) return false;
    if (!direct.done) { fellBack = true; return true; }      // (a pair took the host-driven traceback: the records are not all on the device)
    const size_t n = (size_t) counters_.hits;
    if (!d.txLen.ensure(nq_ + 1, err) || !d.txOff.ensure(nq_ + 2, err) || !d.txLineOff.ensure(n + 1, err) || !d.txLineLen.ensure(n + 1, err) || !d.txFlags.ensure(4, err)) return false;
    if (d.txEvalVersion != gate.version || !d.txEval.p) {
        if (!upload(d.txEval, gate.evalText.data(), gate.evalText.size(), d.stream, err)) return false;
        HIP_OK(hipStreamSynchronize(d.stream));
        d.txEvalVersion = gate.version;
    }
    HIP_OK(hipMemsetAsync(d.txFlags.p, 0, 4 * sizeof(uint32_t), d.stream));
    AlnTextArgs a{s.rec.p, s.cig.p, d.outOffset.p, d.outQuery.p, (uint32_t) nq_, d.outOffset.p + nq_, minSeqId, minAlnLen, d.txEval.p, s.gateSlot.p, gate.stride,
                  d.txLineLen.p, d.txLineOff.p, d.txLen.p, d.txOff.p, nullptr, d.txFlags.p, 0u, nullptr};
    if (!d.txWaveList.ensure(nq_ + 1, err)) return false;
    a.waveList = d.txWaveList.p;
    a.waveServes = n >= 2 * nq_ ? 1u : 0u;
    if (const char *e = getenv("SP_TEXT_WAVE")) a.waveServes = *e != '0' ? 1u : 0u;      // (tests: either kernel on any batch)      // (two or more alignments per query on average -- the dense index: a wave per query for those with 2 .. 64 records)
    if (n) hipLaunchKernelGGL(k_aln_text_linelen, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, a);
    hipLaunchKernelGGL(k_aln_text_measure, dim3((unsigned) ((nq_ + 63) / 64)), dim3(64), 0, d.stream, a);
    hipLaunchKernelGGL(k_aln_text_measure_wave, dim3((unsigned) std::min<size_t>((nq_ + 3) / 4, (size_t) d.numCU * (a.waveServes ? 8 : 2))), dim3(256), 0, d.stream, a);
    if (!d.txTotal.ensure(2, err)) return false;
    HIP_OK(hipMemsetAsync(d.txTotal.p, 0, sizeof(unsigned long long), d.stream));
    hipLaunchKernelGGL(k_sum_u64, dim3((unsigned) std::min<size_t>((nq_ + 255) / 256, 1024)), dim3(256), 0, d.stream, d.txLen.p, (uint32_t) nq_, d.txTotal.p);
    if (!d.scanCounts(d.txLen.p, (uint32_t) nq_, d.txOff.p, err)) return false;
    uint32_t back[2] = {0, 0}; unsigned long long total64 = 0;
    HIP_OK(hipMemcpyAsync(&back[0], d.txOff.p + nq_, sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipMemcpyAsync(&back[1], d.txFlags.p, sizeof(uint32_t), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipMemcpyAsync(&total64, d.txTotal.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
    if (back[1] != 0 || total64 >= deviceTextLimit()) { fellBack = true; return true; }      // (a text of 4 GiB or more does not fit the 32-bit offsets: the host formats)
    used = back[0];
    if (used > cap) return true;
    if (!d.txText.ensure((size_t) back[0] + 16, err)) return false;
    a.text = d.txText.p;
    if (n) hipLaunchKernelGGL(k_aln_text_write, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, a);
    hipLaunchKernelGGL(k_text_entry_ends, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, d.txLen.p, d.txOff.p, (uint32_t) nq_, d.txText.p);
    if (!d.copyOut(text, d.txText.p, back[0], err) || !d.copyOut(entryLen, d.txLen.p, nq_ * sizeof(uint32_t), err)) return false;
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

bool Gpu::downloadCandidates(Candidate *out, std::string &err) {
    Impl &d = *d_;
    if (d.nCand == 0) return true;
    HIP_OK(hipMemcpyAsync(out, d.cand.p, (size_t) d.nCand * sizeof(Candidate), hipMemcpyDefault, d.stream));     // host or device destination
    HIP_OK(hipStreamSynchronize(d.stream));
    return true;
}

// sharded search, step 2: the selection tail over the union of all shards' candidates, for queries [qFrom, qTo); leaves
// the context as prefilter() does (downloadHits(), align() work).  Needs the per-query bias / self scores of a preceding
// prefilterCandidates() on the same query batch.
bool Gpu::mergeCandidates(const Candidate *src, size_t n, bool onDevice, size_t qFrom, size_t qTo, unsigned binSize, int maxHits, int minDiagScore, std::string &err) {
    Impl &d = *d_;
    if (nq_ == 0) return true;
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "mergeCandidates: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    if (!d.qbias.p || !d.queue.p) { err = "mergeCandidates: run prefilterCandidates() on this query batch first"; return false; }
    if (d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }
    if (n >= (1ull << 32)) { err = "mergeCandidates: more than 2^32 candidates"; return false; }
    qTo = std::min(qTo, nq_); qFrom = std::min(qFrom, qTo);
    maxHits_ = maxHits;
    const bool dbg = d.knob.debug;
    const size_t stageCap = std::max<size_t>(nq_ * (size_t) maxHits, n + 1);      // a query's hits are staged where its candidates sit in the grouped list
    if (!d.outHits.ensure(nq_ * (size_t) maxHits, err) || !d.outStage.ensure(stageCap, err) || !d.outQuery.ensure(stageCap, err)) return false;
    if (!d.candCnt.ensure(nq_ + 1, err) || !d.candFill.ensure(nq_ + 1, err) || !d.candOffset.ensure(nq_ + 2, err) || !d.candByQuery.ensure(n + 1, err)) return false;
    const uint32_t blocks = (uint32_t) std::min<size_t>((nq_ + kWavesPerBlock - 1) / kWavesPerBlock, (size_t) d.numCU * 2);
    if (!d.mergeScratch.ensure((size_t) blocks * kWavesPerBlock * kMergeScratchWords, err)) return false;
    const Candidate *u = src;
    if (!onDevice) {
        if (!d.candUnion.ensure(n + 1, err)) return false;
        if (n) HIP_OK(hipMemcpyAsync(d.candUnion.p, src, n * sizeof(Candidate), hipMemcpyHostToDevice, d.stream));
        u = d.candUnion.p;
    }
    HIP_OK(hipMemsetAsync(d.queue.p + 2, 0, 2 * sizeof(uint32_t), d.stream));          // select queue + hit cursor
    HIP_OK(hipMemsetAsync(d.counters.p + 2, 0, sizeof(unsigned long long), d.stream));
    HIP_OK(hipMemsetAsync(d.candCnt.p, 0, (nq_ + 1) * sizeof(uint32_t), d.stream));
    HIP_OK(hipMemsetAsync(d.candFill.p, 0, (nq_ + 1) * sizeof(uint32_t), d.stream));
    SP_EV(3, 2);
    if (n) hipLaunchKernelGGL(k_cand_count, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, u, (uint32_t) n, (uint32_t) qFrom, (uint32_t) qTo, (uint32_t) std::min<size_t>(d.tgt->nTargets, 0xFFFFFFFFull), d.candCnt.p);
    if (!d.scanCounts(d.candCnt.p, (uint32_t) nq_, d.candOffset.p, err)) return false;
    if (n) hipLaunchKernelGGL(k_cand_scatter, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, u, (uint32_t) n, (uint32_t) qFrom, (uint32_t) qTo, (uint32_t) std::min<size_t>(d.tgt->nTargets, 0xFFFFFFFFull), d.candOffset.p, d.candFill.p, d.candByQuery.p);
    MergeArgs me{(uint32_t) nq_, d.candByQuery.p, d.candOffset.p, d.qbias.p, d.qself.p, d.qoff.p, d.mergeScratch.p, d.queue.p + 2, binSize - 1, maxHits, minDiagScore,
                 d.outStage.p, d.outQuery.p, d.outCount.p, d.outBase.p, d.queue.p + 3, d.counters.p};
    hipLaunchKernelGGL(k_select_merged, dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, me);
    if (!d.scanCounts(d.outCount.p, (uint32_t) nq_, d.outOffset.p, err)) return false;
    hipLaunchKernelGGL(k_reorder_hits, dim3((unsigned) ((nq_ + 255) / 256)), dim3(256), 0, d.stream, d.outStage.p, d.outBase.p, d.outCount.p, d.outOffset.p, (uint32_t) nq_, d.outHits.p, d.outQuery.p);
    SP_DBG_SYNC("k_select_merged");
    SP_EV(4, 2);
    return finishPrefilter(false, err);
}

bool Gpu::buildTargets(const char *ascii, const TargetSet &ts, const SubMat &seed, int kmerThr, bool mask, bool index, size_t from, size_t to,
                       TargetBuild &out, std::string &err) {
    Impl &d = *d_;
    const size_t n = ts.n;
    to = std::min(to, n); from = std::min(from, to);
    const uint64_t R = ts.offsets[n];
    if (d.tgt.use_count() > 1) d.tgt = std::make_shared<Impl::Targets>();      // a shared target set belongs to the others from here on: this context stages its own
    d.tgt->nTargets = n; d.tgt->indexEntries = 0; d.tgt->indexSlots = 0; d.tgt->bucketsValid = false; d.tgt->buckets2Valid = false; d.tgt->slotDirectValid = false; { static std::atomic<uint64_t> serial{0}; d.tgt->buildSerial = ++serial; }
    d.tgt->maxTLen = 0;
    for (size_t i = 0; i < n; i++) d.tgt->maxTLen = std::max<uint64_t>(d.tgt->maxTLen, ts.offsets[i + 1] - ts.offsets[i]);
    out = TargetBuild();
    if (!upload(d.tgt->tascii, ascii, (size_t) R, d.stream, err) || !upload(d.tgt->toff, ts.offsets.data(), n + 1, d.stream, err) ||
        !upload(d.tgt->tkeys, ts.keys.data(), n, d.stream, err)) return false;
    if (!d.tgt->tplain.ensure((size_t) R + 1, err) || !d.tgt->tmasked.ensure((size_t) R + 1, err) || !d.ixCounters.ensure(4, err)) return false;
    HIP_OK(hipMemcpyToSymbolAsync(HIP_SYMBOL(c_aa2num), seed.aa2num, 256, 0, hipMemcpyHostToDevice, d.stream));
    HIP_OK(hipMemsetAsync(d.ixCounters.p, 0, 4 * sizeof(unsigned long long), d.stream));
    HIP_OK(hipEventRecord(d.ev[0][0], d.stream));
    if (R) { MapArgs ma{d.tgt->tascii.p, R, d.tgt->tplain.p, d.tgt->tmasked.p}; hipLaunchKernelGGL(k_map_residues, dim3((unsigned) std::min<uint64_t>((R + 255) / 256, 65536)), dim3(256), 0, d.stream, ma); }
    if (mask && to > from) {
        TantanModel tm; tantanModel(seed, tm);
        if (!upload(d.ttLr, &tm.lr[0][0], (size_t) kAlph * kAlph, d.stream, err)) return false;
        // per block of 64 sequences a plane [position][lane] of probabilities (and one of rescaling factors at a sixteenth of its offset + 64 per block): as many
        // positions as the block's longest sequence has
        const size_t nTtBlocks = (to - from + 63) / 64;
        std::vector<uint64_t> planeBase(nTtBlocks + 1, 0);
        for (size_t b = 0; b < nTtBlocks; b++) {
            uint64_t longest = 0;
            for (size_t k = from + 64 * b; k < std::min<size_t>(to, from + 64 * (b + 1)); k++) longest = std::max<uint64_t>(longest, ts.offsets[k + 1] - ts.offsets[k]);
            planeBase[b + 1] = planeBase[b] + 64 * ((longest + 15) / 16 * 16);
        }
        if (!d.ttProbs.ensure((size_t) planeBase[nTtBlocks] + 64, err) || !d.ttScale.ensure((size_t) (planeBase[nTtBlocks] / 16 + 64 * nTtBlocks + 64), err) ||
            !upload(d.ttPlane, planeBase.data(), planeBase.size(), d.stream, err)) return false;
        TantanArgs ta{};
        ta.masked = d.tgt->tmasked.p; ta.toff = d.tgt->toff.p; ta.from = (uint32_t) from; ta.to = (uint32_t) to; ta.lr = d.ttLr.p;
        for (int i = 0; i < 50; i++) ta.b2f[i] = tm.b2f[i];
        ta.b2b = tm.b2b; ta.f2b = tm.f2b; ta.f2f0 = tm.f2f0; ta.probs = d.ttProbs.p; ta.scale = d.ttScale.p; ta.planeBase = d.ttPlane.p; ta.maskedCount = d.ixCounters.p;
        hipLaunchKernelGGL(k_tantan, dim3((unsigned) ((to - from + 63) / 64)), dim3(64), 0, d.stream, ta);
    }
    HIP_OK(hipEventRecord(d.ev[0][1], d.stream));
    const size_t words = kTableSize / 64;
    d.tgt->denseIndex = false;
    if (!d.tgt->pbits.ensure(words, err) || !d.tgt->prank.ensure(words, err) || !d.tgt->prow.ensure(kTableSize / 8000, err) || !d.tgt->slots.ensure(1, err) || !d.tgt->ientries.ensure(1, err)) return false;
    HIP_OK(hipMemsetAsync(d.tgt->pbits.p, 0, words * sizeof(uint64_t), d.stream));
    HIP_OK(hipMemsetAsync(d.tgt->prank.p, 0, words * sizeof(uint16_t), d.stream));
    HIP_OK(hipMemsetAsync(d.tgt->prow.p, 0, (kTableSize / 8000) * sizeof(uint32_t), d.stream));
    const uint64_t base = ts.offsets[from], Rs = ts.offsets[to] - base;
    if (index && Rs > 0) {
        if (Rs >= (1ull << 31)) { err = "index construction: more than 2^31 residues in one shard"; return false; }
        const uint32_t nItems = (uint32_t) Rs;
        if (!d.ixKeys.ensure(nItems, err) || !d.ixKeys2.ensure(nItems, err) || !d.ixVals.ensure(nItems, err) || !d.ixVals2.ensure(nItems, err)) return false;
        ExtractArgs ea{};
        ea.masked = d.tgt->tmasked.p; ea.toff = d.tgt->toff.p; ea.from = (uint32_t) from; ea.to = (uint32_t) to; ea.base = base;
        for (int x = 0; x < kAlph; x++) ea.self[x] = (signed char) seed.sub[x][x];      // getScoreLookup, IndexBuilder.cpp:8-19
        ea.kmerThr = kmerThr; ea.keys = d.ixKeys.p; ea.vals = d.ixVals.p; ea.counters = d.ixCounters.p;
        hipLaunchKernelGGL(k_idx_extract, dim3((unsigned) (to - from)), dim3(64), 0, d.stream, ea);
        // stable LSD radix sort on the 27 key bits: every list comes out in (seqId, pos) order, invalid windows last
        size_t tempBytes = 0;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(nullptr, tempBytes, d.ixKeys.p, d.ixKeys2.p, d.ixVals.p, d.ixVals2.p, (int) nItems, 0, 27, d.stream));
        size_t scanBytes = 0;
        HIP_OK(hipcub::DeviceScan::ExclusiveSum(nullptr, scanBytes, d.ixKeys.p, d.ixKeys.p, (int) std::max<size_t>(nItems, words), d.stream));
        if (!d.ixTemp.ensure(std::max(tempBytes, scanBytes) + 256, err)) return false;
        tempBytes = d.ixTemp.n;
        HIP_OK(hipcub::DeviceRadixSort::SortPairs(d.ixTemp.p, tempBytes, d.ixKeys.p, d.ixKeys2.p, d.ixVals.p, d.ixVals2.p, (int) nItems, 0, 27, d.stream));
        unsigned long long ctr[4];
        HIP_OK(hipMemcpyAsync(ctr, d.ixCounters.p, sizeof(ctr), hipMemcpyDeviceToHost, d.stream));
        HIP_OK(hipStreamSynchronize(d.stream));
        const uint32_t nValid = (uint32_t) ctr[1];
        if (DevPool::on()) {      // (one-shot process: the sort's inputs and the masking planes are done -- the next buffers of the build take their blocks)
            d.ixKeys.release(); d.ixVals.release(); d.ttProbs.release(); d.ttScale.release(); d.ttLr.release(); d.ttPlane.release();
        }
        if (nValid) {
            const unsigned gv = (nValid + 255) / 256;
            if (!d.ixKeep.ensure(nValid + 1, err) || !d.ixPos.ensure(nValid + 1, err)) return false;
            hipLaunchKernelGGL(k_idx_flags, dim3(gv), dim3(256), 0, d.stream, d.ixKeys2.p, d.ixVals2.p, nValid, d.ixKeep.p);
            size_t sb = d.ixTemp.n;
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(d.ixTemp.p, sb, d.ixKeep.p, d.ixPos.p, (int) nValid, d.stream));
            uint32_t lastPos = 0, lastKeep = 0;
            HIP_OK(hipMemcpyAsync(&lastPos, d.ixPos.p + (nValid - 1), 4, hipMemcpyDeviceToHost, d.stream));
            HIP_OK(hipMemcpyAsync(&lastKeep, d.ixKeep.p + (nValid - 1), 4, hipMemcpyDeviceToHost, d.stream));
            HIP_OK(hipStreamSynchronize(d.stream));
            const uint32_t E = lastPos + lastKeep;
            if (!d.tgt->ientries.ensure(E + 1, err) || !d.ixKmer.ensure(E + 1, err)) return false;
            hipLaunchKernelGGL(k_idx_compact, dim3(gv), dim3(256), 0, d.stream, d.ixKeys2.p, d.ixVals2.p, d.ixKeep.p, d.ixPos.p, nValid, d.ixKmer.p, d.tgt->ientries.p);
            if (DevPool::on()) { d.ixKeys2.release(); d.ixVals2.release(); d.ixKeep.release(); d.ixPos.release(); }      // (release() waits for the kernel above)
            if (!d.ixHead.ensure(E + 1, err) || !d.ixSlotOf.ensure(E + 1, err)) return false;
            const unsigned ge = (E + 255) / 256;
            hipLaunchKernelGGL(k_idx_heads, dim3(ge), dim3(256), 0, d.stream, d.ixKmer.p, E, d.ixHead.p);
            sb = d.ixTemp.n;
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(d.ixTemp.p, sb, d.ixHead.p, d.ixSlotOf.p, (int) E, d.stream));
            uint32_t lastSlot = 0, lastHead = 0;
            HIP_OK(hipMemcpyAsync(&lastSlot, d.ixSlotOf.p + (E - 1), 4, hipMemcpyDeviceToHost, d.stream));
            HIP_OK(hipMemcpyAsync(&lastHead, d.ixHead.p + (E - 1), 4, hipMemcpyDeviceToHost, d.stream));
            HIP_OK(hipStreamSynchronize(d.stream));
            const uint32_t nSlots = lastSlot + lastHead;
            if (!d.ixHeadPos.ensure(nSlots + 1, err) || !d.tgt->slots.ensure(nSlots + 1, err) || !d.ixCnt.ensure(words, err) || !d.ixBefore.ensure(words + 1, err)) return false;
            hipLaunchKernelGGL(k_idx_slots1, dim3(ge), dim3(256), 0, d.stream, d.ixKmer.p, d.ixHead.p, d.ixSlotOf.p, E, d.ixHeadPos.p, (unsigned long long *) d.tgt->pbits.p);
            hipLaunchKernelGGL(k_idx_slots2, dim3((nSlots + 255) / 256), dim3(256), 0, d.stream, d.ixHeadPos.p, nSlots, E, d.tgt->ientries.p, d.tgt->slots.p);
            hipLaunchKernelGGL(k_idx_popc, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, d.stream, (const unsigned long long *) d.tgt->pbits.p, (uint32_t) words, d.ixCnt.p);
            sb = d.ixTemp.n;
            HIP_OK(hipcub::DeviceScan::ExclusiveSum(d.ixTemp.p, sb, d.ixCnt.p, d.ixBefore.p, (int) words, d.stream));
            hipLaunchKernelGGL(k_idx_ranks, dim3((unsigned) ((words + 255) / 256)), dim3(256), 0, d.stream, d.ixBefore.p, (uint32_t) words, d.tgt->prank.p, d.tgt->prow.p);
            HIP_OK(hipGetLastError());
            d.tgt->denseIndex = (uint64_t) E * 2 >= (uint64_t) nSlots * 3;
            d.tgt->indexEntries = E; d.tgt->indexSlots = nSlots; out.indexEntries = E; out.indexSlots = nSlots;
        }
    }
    HIP_OK(hipEventRecord(d.ev[0][2], d.stream));
    unsigned long long ctr[4];
    HIP_OK(hipMemcpyAsync(ctr, d.ixCounters.p, sizeof(ctr), hipMemcpyDeviceToHost, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
    HIP_OK(hipGetLastError());
    out.maskedResidues = ctr[0];
    (void) hipEventElapsedTime(&out.msMask, d.ev[0][0], d.ev[0][1]);
    (void) hipEventElapsedTime(&out.msIndex, d.ev[0][1], d.ev[0][2]);
    if (DevPool::on()) {      // a one-shot process builds one target set: the build's scratch goes to the buffers of the search (DevPool)
        for (DevBuf<uint32_t> *b : {&d.ixKeys, &d.ixKeys2, &d.ixKeep, &d.ixPos, &d.ixKmer, &d.ixHead, &d.ixSlotOf, &d.ixHeadPos, &d.ixCnt, &d.ixBefore}) b->release();
        d.ixVals.release(); d.ixVals2.release(); d.ixTemp.release(); d.ttProbs.release(); d.ttScale.release(); d.ttLr.release(); d.ttPlane.release();
    }
    return true;
}

bool Gpu::downloadIndex(std::vector<uint32_t> &kmers, std::vector<uint64_t> &entries, std::string &err) {
    Impl &d = *d_;
    kmers.resize(d.tgt->indexEntries); entries.resize(d.tgt->indexEntries);
    if (d.tgt->indexEntries == 0) return true;
    if (!d.ixKmer.p) { err = "downloadIndex: the resident index was not built on the device"; return false; }
    std::vector<uint32_t> slot(d.tgt->indexEntries); std::vector<uint64_t> ent(d.tgt->indexEntries);
    HIP_OK(hipMemcpy(slot.data(), d.ixKmer.p, d.tgt->indexEntries * sizeof(uint32_t), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(ent.data(), d.tgt->ientries.p, d.tgt->indexEntries * sizeof(uint64_t), hipMemcpyDeviceToHost));
    // back to the reference's table order: k-mer ascending, lists untouched (a stable counting sort by k-mer id)
    std::vector<uint32_t> start((size_t) kTableSize + 1, 0);
    for (uint32_t s2 : slot) start[(size_t) kmerOfSlot(s2) + 1]++;
    for (size_t k = 0; k < kTableSize; k++) start[k + 1] += start[k];
    for (size_t i = 0; i < slot.size(); i++) { const uint32_t km = kmerOfSlot(slot[i]); const uint32_t at = start[km]++; kmers[at] = km; entries[at] = ent[i]; }
    return true;
}

bool Gpu::downloadExt3(std::vector<uint32_t> &packed, std::string &err) {
    Impl &d = *d_;
    if (!d.tab->ext3.p) { err = "downloadExt3: tables not resident"; return false; }
    packed.resize((size_t) 8000 * 8000);
    HIP_OK(hipMemcpy(packed.data(), d.tab->ext3.p, packed.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    for (uint32_t &v : packed) v = (v & 0xFFFF0000u) | unperm3(v & 0xFFFFu);      // the reference's partner numbers
    return true;
}

bool Gpu::downloadMasked(std::vector<uint8_t> &masked, std::string &err) {
    Impl &d = *d_;
    size_t R = 0;
    { uint64_t last = 0; if (d.tgt->nTargets) HIP_OK(hipMemcpy(&last, d.tgt->toff.p + d.tgt->nTargets, sizeof(uint64_t), hipMemcpyDeviceToHost)); R = (size_t) last; }
    masked.resize(R);
    if (R) HIP_OK(hipMemcpy(masked.data(), d.tgt->tmasked.p, R, hipMemcpyDeviceToHost));
    return true;
}

// Result buffers (hit lists, alignment records, backtraces) are written by DMA copies and read by the host after the copy's event: NON-coherent
// page-locked memory, which the CPU reads through its caches.  The default (coherent, fine-grained) kind is mapped uncached on this platform:
// formatting 740 000 hits straight out of it took 32 ms on 16 threads instead of 4.  Only the mailboxes kernels store into stay coherent.
void *pinnedAlloc(size_t bytes, bool coherent) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, coherent ? hipHostMallocDefault : hipHostMallocNonCoherent) != hipSuccess) { fprintf(stderr, "spacepharer_amd: hipHostMalloc(%zu) failed\n", bytes); abort(); }
    return p;
}
bool deviceMemory(size_t &freeBytes, size_t &totalBytes) { return hipMemGetInfo(&freeBytes, &totalBytes) == hipSuccess; }
void pinnedFree(void *p) { if (p) (void) hipHostFree(p); }

bool Gpu::waitAlignCopy(std::string &err) {
    Impl &d = *d_;
    if (d.alnInFlight) { HIP_OK(d.waitCopies(d.alnOnHost)); d.alnInFlight = false; }
    return true;
}

// start copying the (final) hit list of the pass that just completed; called with the main stream idle
void Gpu::setHitSink(GpuHit *hits, uint32_t *offsets, size_t capHits) {
    Impl &d = *d_;
    if (d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); }      // a copy into the previous destination is still running
    // the last pass's hit list may already sit in the OLD destination: forget that copy, so that hostHits() after a switch copies again (into the
    // new buffers, or the internal staging) instead of handing out the new pointers with the old count
    d.hitsInFlight = false; d.sinkUsed = false; d.sinkN = 0;
    d.sinkHits = hits; d.sinkOff = offsets; d.sinkCap = (hits && offsets) ? capHits : 0;
}

// start copying the (final) hit list of the pass that just completed; called with the main stream idle
bool Gpu::prefetchHits(std::string &err) {
    Impl &d = *d_;
    if (d.hitsInFlight) (void) hipEventSynchronize(d.hitsOnHost);      // the staging buffers are about to be rewritten
    const size_t nHits = (size_t) counters_.hits;
    d.sinkUsed = d.sinkCap >= nHits && d.sinkHits != nullptr;           // the caller's buffers when they are large enough
    GpuHit *dstHits; uint32_t *dstOff;
    if (d.sinkUsed) { dstHits = d.sinkHits; dstOff = d.sinkOff; d.sinkN = nHits; }
    else { d.hitOffStage.resize(nq_ + 1); d.hitStage.resize(nHits); dstHits = d.hitStage.data(); dstOff = d.hitOffStage.data(); }
    if (nq_ == 0) { d.hitsInFlight = false; return true; }
    HIP_OK(hipMemcpyAsync(dstOff, d.outOffset.p, (nq_ + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, d.copyStream));
    if (nHits) HIP_OK(hipMemcpyAsync(dstHits, d.outHits.p, nHits * sizeof(GpuHit), hipMemcpyDeviceToHost, d.copyStream));
    HIP_OK(hipEventRecord(d.hitsOnHost, d.copyStream));      // (the hit list keeps its event: who waits for it must not wait for the records queued behind it)
    d.hitsInFlight = true;
    return true;
}

// hits of the last pass grouped by query: offsets[nq + 1]; the pointers stay valid until the next pass
bool Gpu::hostHits(const GpuHit *&hits, const uint32_t *&offsets, size_t &n, std::string &err) {
    Impl &d = *d_;
    if (!d.hitsInFlight && !prefetchHits(err)) return false;
    if (d.hitsInFlight) HIP_OK(hipEventSynchronize(d.hitsOnHost));
    if (d.sinkUsed) { hits = d.sinkHits; offsets = d.sinkOff; n = d.sinkN; return true; }
    if (d.hitOffStage.empty()) d.hitOffStage.assign(1, 0);
    hits = d.hitStage.data(); offsets = d.hitOffStage.data(); n = d.hitStage.size();
    return true;
}

#include "sp_gpu_sw_host.inc"

}  // namespace sp
