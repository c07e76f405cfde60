// HIP kernels (gfx950 / MI355X) for SpacePHARER's spacer->phage-ORF search hot path, and the device context
// that keeps the target index, the target residues and the similar-k-mer tables resident in HBM.
//
// Integer scan/lookup work bounded by random HBM gathers -- no MFMA.  Design notes are in DESIGN.md; the
// reference semantics each kernel reproduces are cited next to it (paths under lib/mmseqs/src/).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <atomic>
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

// ---------------------------------------------------------------------------------------------------------------
// K_prep: per query -- local composition bias, per-window k-mer thresholds, ungapped-profile correction
//   SubstitutionMatrix::calcLocalAaBiasCorrection   commons/SubstitutionMatrix.cpp:85-114
//   QueryMatcher::match threshold                    prefiltering/QueryMatcher.cpp:205-219
//   UngappedAlignment::createProfile                 prefiltering/UngappedAlignment.cpp:295-355
// one thread per query (spacer peptides are 9..30 residues)
// ---------------------------------------------------------------------------------------------------------------
struct PrepArgs {
    const uint8_t *qres; const uint32_t *qoff; uint32_t nq;
    const short *seedSub;      // 21x21
    const double *seedBack;    // 21
    const signed char *ungSub; // 21x21
    int kmerThr; int ungMinAbs;   // |min S_ungapped|
    float *biasf; short *qthr; signed char *qcorr; int *qbias; int *qself;
};

__global__ void k_prep(PrepArgs a) {
    const uint32_t qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= a.nq) return;
    const uint32_t o = a.qoff[qi];
    const int L = (int) (a.qoff[qi + 1] - o);
    const uint8_t *q = a.qres + o;
    int aaBias = 0;
    for (int i = 0; i < L; i++) {
        const int minPos = max(0, i - 20), maxPos = min(L, i + 20);
        const short *row = a.seedSub + q[i] * kAlph;
        int sum = 0;
        for (int j = minPos; j < maxPos; j++) sum += row[q[j]];
        sum -= row[q[i]];
        float d = (float) sum;
        d = (float) ((double) d / (-1.0 * (double) (float) (maxPos - minPos)));
        for (int x = 0; x < kAlph; x++) d = (float) ((double) d + a.seedBack[x] * (double) (float) row[x]);
        a.biasf[o + i] = d;
        float c = d;
        c = (float) ((c < 0.0) ? (double) (c / 4) - 0.5 : (double) (c / 4) + 0.5);
        const signed char cc = (signed char) c;
        a.qcorr[o + i] = cc;
        aaBias = (cc < aaBias) ? cc : aaBias;
    }
    if (aaBias > 0) aaBias = 0;
    a.qbias[qi] = a.ungMinAbs + abs(aaBias);
    // windows
    for (int i = 0; i < L; i++) {
        short t = -1;
        if (i + kSpan <= L) {
            float bc = 0; bool hasX = false;
            const int pp[6] = {0, 1, 3, 4, 5, 7};
#pragma unroll
            for (int p = 0; p < 6; p++) { bc += a.biasf[o + i + pp[p]]; hasX |= (q[i + pp[p]] == kX); }
            if (!hasX) {
                const short b = (short) ((bc < 0.0) ? (double) bc - 0.5 : (double) bc + 0.5);
                int v = a.kmerThr - b; if (v < 0) v = 0;
                t = (short) v;
            }
        }
        a.qthr[o + i] = t;
    }
    // self score on diagonal 0 (QueryMatcher::rescoreHits, prefiltering/QueryMatcher.cpp:489-494)
    int mx = 0, sc = 0;
    for (int i = 0; i < L; i++) { sc += a.ungSub[q[i] * kAlph + q[i]] + a.qcorr[o + i]; sc = sc < 0 ? 0 : sc; mx = sc > mx ? sc : mx; }
    a.qself[qi] = mx;
}

// ---------------------------------------------------------------------------------------------------------------
// k_weigh: how many similar k-mers will a query generate?  The cost of a query in k_match is proportional to that
// number and very skewed (median 350, mean 870, one in a thousand above 11 000 for spacer peptides): a heavy query
// popped late keeps one wave busy long after the others have finished.  Per window the count is exact and cheap from
// the count tables: sum over the score values s of row A that reach the cut-off of  #a(score == s) * #b(score >= thr - s).
// Queries above `heavyThr` are listed; k_match starts with them (one per wave, spread over the CUs).  Only the ORDER
// of the work depends on this, never a result.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kCntW = 320;         // width of a row of the count table: cutoffs cntMin .. cntMin + 319
constexpr int kHeavyEstW = 64;     // slack values the per-window estimate of the similar k-mers is tabulated for (k_heavy_calib)
constexpr int kHeavyLists = 16;    // the heavy queries are collected in 16 lists with counters on separate cache lines (a block appends with one
                                   // atomic, and a thousand blocks finishing together on ONE counter would queue up for ~12 ns each)
constexpr int kHeavyStride = 32;   // words between two counters
struct WeighArgs {
    const uint8_t *qres; const uint32_t *qoff; uint32_t nq; const short *qthr;
    const uint32_t *ext3; const uint16_t *ext3cnt; int cntMin; uint32_t heavyThr;
    uint32_t *heavyList; uint8_t *heavyFlag; uint32_t *heavyCount;
    const uint32_t *est;            // k_prep8 only: similar k-mers of a window by the slack of its best pair (k_heavy_calib); nullptr = count exactly
    uint32_t *weight;               // k_prep8 only, optional: the count / estimate per query (k_order_* sort the batch by it; then nobody is listed as heavy)
};
__global__ __launch_bounds__(256) void k_weigh(WeighArgs a) {
    // 8 lanes per query share the score values of a window (the loads of one lane would be a long dependent chain); a block
    // walks a contiguous range of queries and appends its heavy ones with ONE atomic (atomics on one word serialise)
    __shared__ uint32_t s_list[1024]; __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t per = (a.nq + gridDim.x - 1) / gridDim.x, qFrom = blockIdx.x * per, qTo = min(a.nq, qFrom + per);
    const int sub = (int) (threadIdx.x & 7u);
    for (uint32_t q0 = qFrom; q0 < qTo; q0 += 32) {
        const uint32_t qi = q0 + (threadIdx.x >> 3);
        uint32_t K = 0;
        if (qi < qTo) {
            const uint32_t o = a.qoff[qi];
            const int L = (int) (a.qoff[qi + 1] - o);
            const uint8_t *q = a.qres + o;
            for (int i = 0; i < L - 8; i++) {
                const int thr = a.qthr[o + i];
                if (thr < 0) continue;
                const uint32_t ia = q[i] + 20u * q[i + 1] + 400u * q[i + 3], ib = q[i + 4] + 20u * q[i + 5] + 400u * q[i + 7];
                const int sA0 = extScore(a.ext3[(size_t) ia * 8000u]), sB0 = extScore(a.ext3[(size_t) ib * 8000u]);
                const uint16_t *cA = a.ext3cnt + (size_t) ia * kCntW, *cB = a.ext3cnt + (size_t) ib * kCntW;
                auto at = [&](const uint16_t *row, int c) -> uint32_t { int x = c - a.cntMin; x = x < 0 ? 0 : (x > kCntW - 1 ? kCntW - 1 : x); return row[x]; };
                const int lo = thr - sB0 > a.cntMin ? thr - sB0 : a.cntMin;
                for (int base = lo; base <= sA0; base += 64) {        // 64 score values per trip: 24 independent loads per lane
                    uint32_t nGe[8], nGt[8], nB[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int sc = base + sub + 8 * j; const bool ok = sc <= sA0;
                        nGe[j] = ok ? at(cA, sc) : 0u;
                        nGt[j] = ok && sc + 1 - a.cntMin <= kCntW - 1 ? at(cA, sc + 1) : 0u;
                        nB[j] = ok ? at(cB, thr - sc) : 0u;
                    }
#pragma unroll
                    for (int j = 0; j < 8; j++) K += (nGe[j] - nGt[j]) * nB[j];
                }
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) K += (uint32_t) __shfl_xor((int) K, d, 8);
        const bool heavy = qi < qTo && sub == 0 && K >= a.heavyThr;
        if (qi < qTo && sub == 0) a.heavyFlag[qi] = heavy ? 1 : 0;
        if (heavy) { const uint32_t at = atomicAdd(&s_n, 1u); if (at < 1024u) s_list[at] = qi; else { a.heavyFlag[qi] = 0; } }   // (a full list: the query keeps its normal place)
    }
    __syncthreads();
    const uint32_t n = min(s_n, 1024u);
    const uint32_t hl = blockIdx.x % kHeavyLists;
    if (threadIdx.x == 0 && n) s_base = atomicAdd(a.heavyCount + hl * kHeavyStride, n);
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n; x += blockDim.x) a.heavyList[(size_t) hl * a.nq + s_base + x] = s_list[x];
}

// k_prep + k_weigh in one launch for batches whose queries have at most 64 residues (every spacer batch): 8 lanes per query,
// residues / bias values / thresholds staged in LDS, so the chains of dependent global round trips of the one-lane-per-query
// kernels (store bias -> load it back per window, threshold -> table heads -> count rows) shrink to three.  Same
// arithmetic, in the same order, as k_prep; same counts as k_weigh.
// (+ the per-query part of the Smith-Waterman set-up -- composition bias of the align matrix and profile bias, k_sw_prep8 in sp_gpu_sw.inc -- when sw.sub16 is
// set: it needs the same staged residues, and as a launch of its own it cost the step 35-80 us beside another context's k_match)
struct SwBiasArgs { const short *sub16; const double *back; int matMinAbs; signed char *qcomp; int *qbias; };
struct Prep8Args { PrepArgs p; WeighArgs w; uint32_t stride; SwBiasArgs sw; };
template <bool kExactWeigh>      // (the exact weights' loops cost the kernel 40 registers: the estimating variant fits two waves per SIMD beside a matching kernel)
__global__ __launch_bounds__(256) void k_prep8(Prep8Args a8) {
    const PrepArgs &a = a8.p; const WeighArgs &wa = a8.w;
    __shared__ short s_seed[kAlph * kAlph]; __shared__ double s_back[kAlph]; __shared__ signed char s_ung[kAlph * kAlph];
    // per-query staging rows are as long as the longest query of the batch (rounded up to 8), not 64: a spacer batch needs 4 KB of LDS
    // per block instead of 24, so the kernel squeezes onto CUs that another context's k_match already occupies
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const int S = (int) a8.stride;
    float *const s_bA = reinterpret_cast<float *>(s_dyn);                       // [32][S]
    uint32_t *const s_topA = reinterpret_cast<uint32_t *>(s_bA + 32 * S);       // [32][S]
    short *const s_tA = reinterpret_cast<short *>(s_topA + 32 * S);             // [32][S]
    uint8_t *const s_qA = reinterpret_cast<uint8_t *>(s_tA + 32 * S);           // [32][S]
    signed char *const s_cA = reinterpret_cast<signed char *>(s_qA + 32 * S);   // [32][S]
    __shared__ uint32_t s_list[1024]; __shared__ uint32_t s_n, s_base;
    __shared__ short s_sw[kAlph * kAlph]; __shared__ double s_swBack[kAlph]; __shared__ uint32_t s_est[kHeavyEstW];
    if (a8.w.est && threadIdx.x < kHeavyEstW) s_est[threadIdx.x] = a8.w.est[threadIdx.x];
    const SwBiasArgs &sw = a8.sw;
    for (int x = threadIdx.x; x < kAlph * kAlph; x += blockDim.x) { s_seed[x] = a.seedSub[x]; s_ung[x] = a.ungSub[x]; if (sw.sub16) s_sw[x] = sw.sub16[x]; }
    if (threadIdx.x < kAlph) { s_back[threadIdx.x] = a.seedBack[threadIdx.x]; if (sw.sub16) s_swBack[threadIdx.x] = sw.back[threadIdx.x]; }
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t per = (a.nq + gridDim.x - 1) / gridDim.x, qFrom = blockIdx.x * per, qTo = min(a.nq, qFrom + per);
    const int sub = (int) (threadIdx.x & 7u), slot = (int) (threadIdx.x >> 3);
    for (uint32_t q0 = qFrom; q0 < qTo; q0 += 32) {
        const uint32_t qi = q0 + (uint32_t) slot;
        const bool have = qi < qTo;
        const uint32_t o = have ? a.qoff[qi] : 0u;
        const int L = have ? (int) (a.qoff[qi + 1] - o) : 0;
        for (int i = sub; i < L; i += 8) s_qA[slot * S + (i)] = a.qres[o + i];
        __syncthreads();
        // local amino-acid bias (SubstitutionMatrix::calcLocalAaBiasCorrection, SubstitutionMatrix.cpp:85-114)
        int aaBias = 0;
        for (int i = sub; i < L; i += 8) {
            const int minPos = max(0, i - 20), maxPos = min(L, i + 20);
            const short *row = s_seed + s_qA[slot * S + (i)] * kAlph;
            int sum = 0;
#pragma unroll 2
            for (int j = minPos; j < maxPos; j++) sum += row[s_qA[slot * S + (j)]];
            sum -= row[s_qA[slot * S + (i)]];
            float d = (float) sum;
            d = (float) ((double) d / (-1.0 * (double) (float) (maxPos - minPos)));
#pragma unroll 1
            for (int x = 0; x < kAlph; x++) d = (float) ((double) d + s_back[x] * (double) (float) row[x]);
            a.biasf[o + i] = d; s_bA[slot * S + (i)] = d;
            float c = d;
            c = (float) ((c < 0.0) ? (double) (c / 4) - 0.5 : (double) (c / 4) + 0.5);
            const signed char cc = (signed char) c;
            a.qcorr[o + i] = cc; s_cA[slot * S + (i)] = cc;
            aaBias = (cc < aaBias) ? cc : aaBias;
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) { const int ob = __shfl_xor(aaBias, d, 8); aaBias = ob < aaBias ? ob : aaBias; }
        if (have && sub == 0) a.qbias[qi] = a.ungMinAbs + abs(aaBias > 0 ? 0 : aaBias);
        if (sw.sub16) {      // SmithWaterman::ssw_init (StripedSmithWaterman.cpp:689-743): the statement of k_sw_prep8
            int cb = 0;
            for (int i = sub; i < L; i += 8) {
                const int minPos = max(0, i - 20), maxPos = min(L, i + 20);
                const short *row = s_sw + s_qA[slot * S + (i)] * kAlph;
                int sum = 0;
    #pragma unroll 2
            for (int j = minPos; j < maxPos; j++) sum += row[s_qA[slot * S + (j)]];
                sum -= row[s_qA[slot * S + (i)]];
                float d = (float) sum;
                d = (float) ((double) d / (-1.0 * (double) (float) (maxPos - minPos)));
#pragma unroll 1
                for (int x = 0; x < kAlph; x++) d = (float) ((double) d + s_swBack[x] * (double) (float) row[x]);
                const signed char c = (signed char) ((d < 0.0) ? (double) d - 0.5 : (double) d + 0.5);
                sw.qcomp[o + i] = c;
                cb = (cb < c) ? cb : c;
            }
#pragma unroll
            for (int dd = 1; dd < 8; dd <<= 1) { const int ob = __shfl_xor(cb, dd, 8); cb = ob < cb ? ob : cb; }
            if (cb > 0) cb = 0;
            if (have && sub == 0) sw.qbias[qi] = sw.matMinAbs + abs(cb);
        }
        __syncthreads();
        // k-mer thresholds per window (QueryMatcher.cpp:205-219)
        for (int i = sub; i < L; i += 8) {
            short t = -1;
            if (i + kSpan <= L) {
                float bc = 0; bool hasX = false;
                const int pp[6] = {0, 1, 3, 4, 5, 7};
#pragma unroll
                for (int p = 0; p < 6; p++) { bc += s_bA[slot * S + (i + pp[p])]; hasX |= (s_qA[slot * S + (i + pp[p])] == kX); }
                if (!hasX) {
                    const short b = (short) ((bc < 0.0) ? (double) bc - 0.5 : (double) bc + 0.5);
                    int v = a.kmerThr - b; if (v < 0) v = 0;
                    t = (short) v;
                }
            }
            a.qthr[o + i] = t; s_tA[slot * S + (i)] = t;
        }
        // self score on diagonal 0 (QueryMatcher::rescoreHits, prefiltering/QueryMatcher.cpp:489-494)
        if (have && sub == 0) {
            int mx = 0, sc = 0;
#pragma unroll 2
            for (int i = 0; i < L; i++) { sc += s_ung[s_qA[slot * S + (i)] * kAlph + s_qA[slot * S + (i)]] + s_cA[slot * S + (i)]; sc = sc < 0 ? 0 : sc; mx = sc > mx ? sc : mx; }
            a.qself[qi] = mx;
        }
        __syncthreads();
        // similar k-mers the query will generate (see k_weigh).  Estimated from the slack of every window's best pair over its threshold -- the best pair is
        // the k-mer itself, its score the sum of six diagonal entries -- when the calibration table is there: no table row is read, this kernel's chains
        // of dependent round trips (heads of two rows per window, then their count rows) shrink to the residues alone.
        const bool weighEst = !kExactWeigh && wa.heavyThr != 0xFFFFFFFFu && wa.est != nullptr;
        if (weighEst) {
            uint32_t Ke = 0;
            for (int i = sub; i < L - 8; i += 8) {
                const int thr = s_tA[slot * S + (i)];
                if (thr < 0) continue;
                const uint8_t *q = s_qA + slot * S;
                int self = 0;
                const int pp[6] = {0, 1, 3, 4, 5, 7};
#pragma unroll
                for (int p = 0; p < 6; p++) self += s_seed[q[i + pp[p]] * kAlph + q[i + pp[p]]];
                const int slack = self - thr;
                if (slack >= 0) Ke += s_est[slack < kHeavyEstW ? slack : kHeavyEstW - 1];
            }
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) Ke += (uint32_t) __shfl_xor((int) Ke, d, 8);
            const bool heavyE = have && sub == 0 && Ke >= wa.heavyThr;
            if (have && sub == 0 && wa.weight) wa.weight[qi] = Ke;
            if (have && sub == 0) wa.heavyFlag[qi] = heavyE ? 1 : 0;
            if (heavyE) { const uint32_t at = atomicAdd(&s_n, 1u); if (at < 1024u) s_list[at] = qi; else wa.heavyFlag[qi] = 0; }
            __syncthreads();
            continue;
        }
        const bool weigh = kExactWeigh && wa.heavyThr != 0xFFFFFFFFu;        // off for large batches (see matchAndScore)
        for (int i = sub; i < (weigh ? L - 8 : 0); i += 8) {
            const uint8_t *q = s_qA + slot * S;
            const uint32_t ia = q[i] + 20u * q[i + 1] + 400u * q[i + 3], ib = q[i + 4] + 20u * q[i + 5] + 400u * q[i + 7];
            s_topA[slot * S + (i)] = s_tA[slot * S + (i)] < 0 ? 0u : (wa.ext3[(size_t) ia * 8000u] >> 16) | (wa.ext3[(size_t) ib * 8000u] & 0xFFFF0000u);
        }
        __syncthreads();
        uint32_t K = 0;
        for (int i = 0; i < (weigh ? L - 8 : 0); i++) {
            const int thr = s_tA[slot * S + (i)];
            if (thr < 0) continue;
            const uint8_t *q = s_qA + slot * S;
            const uint32_t ia = q[i] + 20u * q[i + 1] + 400u * q[i + 3], ib = q[i + 4] + 20u * q[i + 5] + 400u * q[i + 7];
            const int sA0 = (int) (s_topA[slot * S + (i)] & 0xFFFFu) - 32768, sB0 = (int) (s_topA[slot * S + (i)] >> 16) - 32768;
            const uint16_t *cA = wa.ext3cnt + (size_t) ia * kCntW, *cB = wa.ext3cnt + (size_t) ib * kCntW;
            auto at = [&](const uint16_t *row, int c) -> uint32_t { int x = c - wa.cntMin; x = x < 0 ? 0 : (x > kCntW - 1 ? kCntW - 1 : x); return row[x]; };
            const int lo = thr - sB0 > wa.cntMin ? thr - sB0 : wa.cntMin;
            for (int base = lo; base <= sA0; base += 32) {
                uint32_t nGe[4], nGt[4], nB[4];
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int sc = base + sub + 8 * j; const bool ok = sc <= sA0;
                    nGe[j] = ok ? at(cA, sc) : 0u;
                    nGt[j] = ok && sc + 1 - wa.cntMin <= kCntW - 1 ? at(cA, sc + 1) : 0u;
                    nB[j] = ok ? at(cB, thr - sc) : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) K += (nGe[j] - nGt[j]) * nB[j];
            }
        }
#pragma unroll
        for (int d = 1; d < 8; d <<= 1) K += (uint32_t) __shfl_xor((int) K, d, 8);
        const bool heavy = have && sub == 0 && K >= wa.heavyThr;
        if (have && sub == 0 && wa.weight) wa.weight[qi] = K;
        if (have && sub == 0) wa.heavyFlag[qi] = heavy ? 1 : 0;
        if (heavy) { const uint32_t at = atomicAdd(&s_n, 1u); if (at < 1024u) s_list[at] = qi; else wa.heavyFlag[qi] = 0; }
        __syncthreads();
    }
    const uint32_t n = min(s_n, 1024u);
    const uint32_t hl = blockIdx.x % kHeavyLists;
    if (threadIdx.x == 0 && n) s_base = atomicAdd(wa.heavyCount + hl * kHeavyStride, n);
    __syncthreads();
    for (uint32_t x = threadIdx.x; x < n; x += blockDim.x) wa.heavyList[(size_t) hl * a.nq + s_base + x] = s_list[x];
}

// ---------------------------------------------------------------------------------------------------------------
// The batch in the order of decreasing weight (k_match_grp, short queries): a counting sort by 64 quarter-octave classes of the similar-k-mer count,
// tiles of kOrderTile queries.  The matching kernel hands the queries out in this order -- longest first, the classic list schedule: a wave that
// holds one of the heaviest queries (20 000 similar k-mers against 850 on average) takes nothing else meanwhile, and the light queries fill in at the
// end -- through tickets on kOrderQueues counters (one counter for 60 000 queries is a line that 4 096 waves queue on).  A ticket is one query
// while the queries are heavy, then two, four, eight: OrderHeader holds the positions where the size changes.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kOrderTile = 2048, kOrderClasses = 64, kOrderQueues = 16, kOrderQueueStride = 32;
struct OrderHeader { uint32_t p[4]; uint32_t t[4]; };      // p[k]: first position of step k + 1 (p[3] = nq); t[k]: tickets before step k + 1 (t[3] = all)
__device__ __forceinline__ uint32_t weightClass(uint32_t w) {
    if (w < 4u) return w;
    const uint32_t l = 31u - (uint32_t) __clz((int) w), c = 4u * l + ((w >> (l - 2u)) & 3u);
    return c < (uint32_t) kOrderClasses ? c : (uint32_t) kOrderClasses - 1u;
}
__global__ __launch_bounds__(256) void k_order_count(const uint32_t *weight, uint32_t nq, uint32_t *tileHist) {
    __shared__ uint32_t s_h[kOrderClasses];
    if (threadIdx.x < kOrderClasses) s_h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t) kOrderTile; i += 256) { const uint32_t q = blockIdx.x * kOrderTile + i; if (q < nq) atomicAdd(&s_h[weightClass(weight[q])], 1u); }
    __syncthreads();
    if (threadIdx.x < kOrderClasses) tileHist[blockIdx.x * kOrderClasses + threadIdx.x] = s_h[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_order_scatter(const uint32_t *weight, uint32_t nq, const uint32_t *tileHist, uint32_t nTiles, uint32_t *order, OrderHeader *header,
                                                       uint32_t *queues, uint32_t firstTickets) {
    __shared__ uint32_t s_tot[kOrderClasses], s_before[kOrderClasses], s_base[kOrderClasses], s_cur[kOrderClasses];
    if (threadIdx.x < kOrderClasses) {
        uint32_t tot = 0, before = 0;
        for (uint32_t t = 0; t < nTiles; t++) { const uint32_t v = tileHist[t * kOrderClasses + threadIdx.x]; tot += v; if (t < blockIdx.x) before += v; }
        s_tot[threadIdx.x] = tot; s_before[threadIdx.x] = before; s_cur[threadIdx.x] = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int c = kOrderClasses - 1; c >= 0; c--) { s_base[c] = run + s_before[c]; run += s_tot[c]; }      // descending classes
        if (blockIdx.x == 0) {
            // a ticket holds 1 query from 2 048 similar k-mers up (class 44), 2 from 1 024 (40), 4 from 512 (36), 8 below
            uint32_t p1 = 0, p2 = 0, p3 = 0;
            for (int c = kOrderClasses - 1; c >= 36; c--) { if (c >= 44) p1 += s_tot[c]; if (c >= 40) p2 += s_tot[c]; p3 += s_tot[c]; }
            OrderHeader h;
            h.p[0] = p1; h.p[1] = p2; h.p[2] = p3; h.p[3] = nq;
            h.t[0] = p1; h.t[1] = h.t[0] + (p2 - p1 + 1u) / 2u; h.t[2] = h.t[1] + (p3 - p2 + 3u) / 4u; h.t[3] = h.t[2] + (nq - p3 + 7u) / 8u;
            *header = h;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (uint32_t) kOrderQueues) queues[threadIdx.x * kOrderQueueStride] = firstTickets;      // (every wave starts with the ticket of its own number)
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < (uint32_t) kOrderTile; i += 256) {
        const uint32_t q = blockIdx.x * kOrderTile + i;
        if (q < nq) { const uint32_t c = weightClass(weight[q]); order[s_base[c] + atomicAdd(&s_cur[c], 1u)] = q; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// K_match: one wavefront per query (persistent waves pulling queries from an atomic counter)
//   similar k-mer lists        KmerGenerator::generateKmerList / calculateArrayProduct, KmerGenerator.cpp:107-216
//   index lookup + hit stream  QueryMatcher::match, QueryMatcher.cpp:188-307
//   double-hit detection       CacheFriendlyOperations::findDuplicates, CacheFriendlyOperations.cpp:146-235
//
// The reference scans the hit stream bin by bin with a byte array "previous 8-bit diagonal per target".  Per
// target that is a sequential automaton over the target's hits in stream order, so here the hits of a query
// are sorted by (target, stream position) in LDS and the automaton runs on adjacent elements:
//   recorded(h)  <=>  diag8(h) == diag8(previous hit of the same target)   (previous := 0 for the first hit)
//   kept(h)      <=>  recorded(h) and (no earlier recorded hit of the target, or diag8 differs from the last one)
// Only windows [0, lastWindow) are scanned (CacheFriendlyOperations.cpp:43 with QueryMatcher.cpp:285-291).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSpillCap = 65535;     // hits per query in the per-wave HBM scratch (16-bit stream index)
constexpr int kBTile = 512;
constexpr int kBStage = 256;         // partners of row B staged in LDS by the small footprint of k_match (one 16-byte LDS-DMA per lane); later ones are read in place
constexpr uint32_t kMaxKmerResult = 262144u * 32u;   // KmerGenerator.h:45
constexpr int kWavesPerBlock = 4;
constexpr uint32_t kDiagChunk = 64;   // diagonal records reserved per atomic in k_match
constexpr uint32_t kQueryChunk = 4;
constexpr int kDeferCap = 128;       // multi-entry index lists per wave whose entries are fetched together (k_match)
constexpr int kWalkStep = 4;        // list positions per step when long lists are walked on the spot (kOccRegs x kWalkStep entry loads in flight)
constexpr int kTailKeys = 3;        // keys per lane and step when the pre-filter reads spilled hits back from HBM
constexpr int kMapGroup = 4;         // rounds of the (a, b) mapping whose LDS lookups are issued together

struct DiagRec { uint32_t seqId; uint16_t diag; uint16_t idx; };

struct MatchArgs {
    const uint8_t *qres; const uint32_t *qoff; uint32_t nq; const short *qthr;
    const uint32_t *ext3;           // 8000 x 8000 packed (score+32768)<<16 | index
    const uint16_t *ext3cnt;        // 8000 x kCntW: [row][c - cntMin] = number of partners with score >= c
    int cntMin;
    // rank-compressed slot table.  k-mer = a + 8000 b: the 8000 slots of one b form a "row" of 125 words of 64 slots.
    const uint64_t *pbits;          // occupancy bits, one u64 per 64 slots (8 MB): all a miss needs; a 128-byte line covers 1 024 slots
    const uint16_t *prank;          // occupied slots of the row before the word (2 MB)
    const uint32_t *prow;           // occupied slots before the row (8 000 entries)
    const uint64_t *slots;          // per OCCUPIED slot (rank order): the entry itself (seqId | pos << 32) when the list has one
                                    // entry, else bit 63 | len << 32 | offset into ientries
    const uint64_t *ientries;       // seqId | pos << 32, lists concatenated in k-mer order
    const uint64_t *buckets;        // large footprint only: 32 bytes per k-mer SLOT (20^6 of them, 2 GB): up to five entries of a list, or three and where the rest is (k_idx_buckets)
    uint64_t *spill;                // per-wave scratch: kSpillCap+1 keys (+ bitmap) each
    uint32_t *queue;                // work counter
    DiagRec *diagRec; uint32_t *diagQuery; uint32_t diagCap; uint32_t *diagCursor;
    uint32_t *qDiagBase; uint32_t *qDiagCnt;
    unsigned long long *counters;   // [0]=kmers [1]=dbMatches [2]=overflow queries
    // target-sharded search only (kOrd): ordinal of the similar k-mer that produced the recording hit, counted over the
    // query's whole k-mer enumeration -- identical on every shard, so (ordinal, seqId) orders records globally
    uint32_t *diagOrd; uint32_t *spillOrd;
    uint32_t staticEighths, dynChunk;   // work distribution: share of the queries dealt round-robin, queries per atomic pop for the rest
    const uint32_t *heavyList; const uint8_t *heavyFlag; const uint32_t *heavyCount;   // k_weigh: queries to start with
    unsigned long long *wavePartial;   // [wave][4]: similar k-mers, index entries, diagonals of this wave (summed by k_score: 9 000 atomics on
                                       // one cache line at the end of the kernel, when all waves finish together, cost tens of microseconds)
    const uint32_t *order; const OrderHeader *orderHeader; uint32_t *orderQueues;   // k_match_grp, optional: the batch by decreasing weight and its tickets (k_order_*)
};

// phase timers (build with -DSP_PHASE_TIMERS): shader-clock ticks per phase, summed over waves into counters[8 + phase]
#ifdef SP_PHASE_TIMERS
#define SP_T_DECL unsigned long long tPh[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tLast = __builtin_amdgcn_s_memtime();
#define SP_T(ph) do { const unsigned long long tNow = __builtin_amdgcn_s_memtime(); tPh[ph] += tNow - tLast; tLast = tNow; } while (0)
#define SP_T_FLUSH(ctr) do { if (lane == 0) for (int ph = 0; ph < 8; ph++) atomicAdd(&(ctr)[8 + ph], tPh[ph]); } while (0)
#else
#define SP_T_DECL
#define SP_T(ph) do { } while (0)
#define SP_T_FLUSH(ctr) do { } while (0)
#endif

constexpr size_t kSpillKeysPerWave = 65536 + 512;             // (+ 512: the dense index writes its hits behind a gap the size of the LDS buffer, see kRing)
constexpr size_t kSpillWordsPerWave = kSpillKeysPerWave + 65536 / 64;   // keys + kept-bitmap (as u64 words)

__device__ void markKept(const uint64_t *keys, uint32_t M, uint32_t *bits, int lane) {
    for (uint32_t base = 0; base < M; base += 64) {
        const uint32_t t = base + lane;
        if (t < M) {
            const uint32_t id = (uint32_t) (keys[t] >> 32);
            const bool head = (t == 0) || ((uint32_t) (keys[t - 1] >> 32) != id);
            if (head) {
                uint32_t prev = 0, lastRec = 0; bool hasRec = false;
                for (uint32_t u = t; u < M; u++) {
                    const uint64_t k = keys[u];
                    if ((uint32_t) (k >> 32) != id) break;
                    const uint32_t d = (uint32_t) (k & 0xFF);
                    const bool rec = (d == prev);
                    prev = d;
                    if (rec) {
                        if (!hasRec || lastRec != d) atomicOr(&bits[u >> 5], 1u << (u & 31));
                        lastRec = d; hasRec = true;
                    }
                }
            }
        }
    }
}

// HITCAP = hits per query held in LDS; more spill to the per-wave HBM scratch.  Smaller HITCAP = more resident waves.
// kSmall: the footprint for a sparse index (phage scale: 1.1 entries per occupied k-mer, ~110 hits per query) -- super-rounds of 512
// similar k-mers, 256 hits in LDS, two slot registers per lane: 9 KB of LDS and <= 128 VGPRs per wave, i.e. FOUR waves per SIMD or three
// plus room for another context's aligner on the same CU.  Alone it runs as fast as the large footprint (what the fourth wave hides,
// the second super-round of a window above 512 k-mers costs); with two search contexts per GPU it is 6 % faster per step.  A dense index
// (2 200 hits per query) needs the large one: its pre-filter bitmaps and hit buffer are sized by kSuper / kHitCap.
template <int kHitCap, bool kOrd, bool kSmall>
__global__ __launch_bounds__(64 * kWavesPerBlock, kSmall ? (kOrd ? 3 : 4) : (kOrd ? 2 : 3)) void k_match(MatchArgs a) {
    constexpr int kSuper = kSmall ? 512 : 1024;         // similar k-mers per enumeration super-round
    constexpr int kOccRegs = 2;                          // occupied slots per lane and batch in the fetch phase: 64 x 2 slot words (small footprint) or 32-byte buckets (large) in flight per wave
    constexpr int kATile = kSmall ? 128 : 256;          // a's per tile of row A (a window of a spacer peptide has 20-60)
    static_assert(kHitCap == (kSmall ? 256 : 512), "hit buffer and footprint go together");
    __shared__ uint16_t s_occOrd[kWavesPerBlock][kOrd ? kSuper : 1];
    __shared__ uint32_t s_ordL[kWavesPerBlock][kOrd ? kHitCap : 1];
    __shared__ uint64_t s_keys[kWavesPerBlock][kHitCap];
    __shared__ uint32_t s_aStart[kWavesPerBlock][kATile + 1];
    __shared__ uint16_t s_aIdx[kWavesPerBlock][kATile];
    // small footprint: the rows a window starts from -- the head of row A, the head of row B and the count row -- arrive by LDS-DMA (global_load_lds,
    // no registers in between), requested while the PREVIOUS window's slots are still being fetched; the large footprint keeps the partner indices only
    __shared__ uint16_t s_bIdx[kWavesPerBlock][kSmall ? 1 : kBTile];
    __shared__ __attribute__((aligned(16))) uint32_t s_stA[kWavesPerBlock][kSmall ? kATile : 4];
    __shared__ __attribute__((aligned(16))) uint32_t s_stB[kWavesPerBlock][kSmall ? kBStage : 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_cntB[kWavesPerBlock][kCntW];
    __shared__ uint32_t s_bits[kWavesPerBlock][kHitCap / 32];
    __shared__ uint32_t s_occ[kWavesPerBlock][kSuper];
    __shared__ unsigned long long s_hflag[kWavesPerBlock][kSuper / 64];
    __shared__ uint64_t s_defSlot[kWavesPerBlock][kDeferCap];      // multi-entry lists whose walk is deferred (slot word)
    __shared__ uint32_t s_defMeta[kWavesPerBlock][kDeferCap];      // ... stream position of the list's first hit | window << 16
    __shared__ uint32_t s_defOrd[kWavesPerBlock][kOrd ? kDeferCap : 1];

    const int lane = laneId();
    const int w = __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));      // (wave-uniform, and the compiler should know: the work distribution below derives from it)
    uint64_t *keys = s_keys[w];
    uint32_t *aStart = s_aStart[w];
    uint16_t *aIdx = s_aIdx[w];
    uint16_t *bIdx = s_bIdx[w];
    uint32_t *stA = s_stA[w], *stB = s_stB[w];
    uint16_t *cntB = s_cntB[w];
    uint32_t *bitsL = s_bits[w];
    uint32_t *occList = s_occ[w];
    unsigned long long *hflag = s_hflag[w];
    uint16_t *occOrd = s_occOrd[w];
    uint32_t *ordL = s_ordL[w];
    uint64_t *defSlot = s_defSlot[w]; uint32_t *defMeta = s_defMeta[w]; uint32_t *defOrd = s_defOrd[w];
    const uint32_t waveGlobal = blockIdx.x * kWavesPerBlock + w;
    uint64_t *spillKeys = a.spill + (size_t) waveGlobal * kSpillWordsPerWave;
    uint32_t *spillBits = (uint32_t *) (spillKeys + kSpillKeysPerWave);
    uint32_t *spillOrd = kOrd ? a.spillOrd + (size_t) waveGlobal * 65536 : nullptr;
    unsigned long long kmersTotal = 0, matchesTotal = 0, diagTotal = 0;
    uint32_t dChunkBase = 0, dChunkUsed = 0, dChunkCap = 0;
    // slots of a reservation chunk that got no record are marked invalid when the wave leaves the chunk (instead of a
    // 16 MB fill of the whole array before every pass)
    auto markUnused = [&]() {
        for (uint32_t t = dChunkUsed + lane; t < dChunkCap; t += 64) if ((uint64_t) dChunkBase + t < a.diagCap) a.diagQuery[dChunkBase + t] = 0xFFFFFFFFu;
    };
    SP_T_DECL
    const unsigned long long tWaveStart = wall_clock64();
    if (threadIdx.x == 0) atomicMax(&a.counters[6], ~(unsigned long long) wall_clock64());      // (max of the complement = the earliest start; counters are zeroed per pass)

    // one atomic pops kQueryChunk queries: a single counter word serialises at ~12 ns per atomic (60 000 pops = 0.7 ms)
    // hybrid work distribution: the first staticEighths / 8 of the queries are dealt round-robin (no atomics), the rest are
    // popped dynChunk at a time to even out the tail (half / 4 measured best on the spacer workload: SP_MATCH_STATIC, SP_MATCH_DCHUNK)
    const uint32_t nWavesTot = gridDim.x * kWavesPerBlock;
    const uint32_t staticEnd = (uint32_t) ((uint64_t) a.nq * a.staticEighths / 8 / (nWavesTot * kQueryChunk)) * (nWavesTot * kQueryChunk);
    uint32_t sNext = waveGlobal * kQueryChunk;
    // the heavy queries first: the j-th of them goes to wave j / #blocks of block j % #blocks (one per block -- one per SIMD of
    // a CU -- before any block gets a second one); everybody skips them later
    // (16 lists: block b serves list b % 16; within a list the entries go to wave 0 of each of its blocks first, then wave 1 ...)
    const uint32_t myList = blockIdx.x % (uint32_t) kHeavyLists;
    const uint32_t listBlocks = (gridDim.x - myList + (uint32_t) kHeavyLists - 1u) / (uint32_t) kHeavyLists;
    const uint32_t nHeavy = uni(a.heavyCount[myList * kHeavyStride]);
    const uint32_t *heavyList = a.heavyList + (size_t) myList * a.nq;
    uint32_t hNext = (uint32_t) w * listBlocks + blockIdx.x / (uint32_t) kHeavyLists;
    // Work arrives in CHUNKS of consecutive queries (one heavy query, kQueryChunk dealt ones or dynChunk popped ones): one round trip brings the
    // offsets and the heavy flags of the whole chunk into lanes, and the residues / thresholds of a query (one lane per position, packed) are requested
    // while the query before it is still being matched -- the wave never waits for a header in the middle of a chunk.
    uint32_t c0 = 0, offL = 0; uint64_t todo = 0ull; int nxt = -1, pkN = -256, wN = -1; bool wNValid = false;
    auto windowWords = [&](int pk) -> int {
        const int res = pk & 0xFF;
        const int r1 = __builtin_amdgcn_update_dpp(0, res, SP_DPP_WAVE_SHL1, 0xF, 0xF, false);      // residue of lane + 1
        const int r2 = __builtin_amdgcn_update_dpp(0, r1, SP_DPP_WAVE_SHL1, 0xF, 0xF, false);
        const int r3 = __builtin_amdgcn_update_dpp(0, r2, SP_DPP_WAVE_SHL1, 0xF, 0xF, false);       // ... of lane + 3
        return (res + 20 * r1 + 400 * r3) | ((pk >> 8) << 16);      // (threshold -1: negative word; positions whose window leaves the query are never asked)
    };
    auto header = [&](int c) -> int {
        const uint32_t o = (uint32_t) __builtin_amdgcn_readlane((int) offL, c), Lc = (uint32_t) __builtin_amdgcn_readlane((int) offL, c + 1) - o;
        return (uint32_t) lane < Lc ? ((int) a.qres[o + lane] | ((int) a.qthr[o + lane] << 8)) : -256;      // residue | threshold << 8; beyond the query: threshold -1
    };
    // rows staged in LDS (small footprint): which (row A, row B) pair they are, or are on their way to be
    uint32_t stagedA = 0xFFFFFFFFu, stagedB = 0xFFFFFFFFu;
    bool stagedLanded = false;      // the staged rows are known to be in LDS (a wait went by since they were requested)
    const uint32_t stALds = uni(ldsAddr(stA)), stBLds = uni(ldsAddr(stB)), cntLds = uni(ldsAddr(cntB));
    // one statement: the three requests with scalar row bases and a lane offset, exec narrowed inside (32 lanes bring the head of row A, 64 the partners of
    // row B, 40 the count row), M0 and exec restored.  Called from wave-uniform code only (all lanes on).
    const uint32_t dmaOff = (uint32_t) lane * 16u;
    auto stage = [&](uint32_t ia, uint32_t ib) {
      if constexpr (kSmall) {
        const uint64_t bA = (uint64_t) (uintptr_t) a.ext3 + (uint64_t) ia * 32000ull, bB = (uint64_t) (uintptr_t) a.ext3 + (uint64_t) ib * 32000ull;
        const uint64_t bC = (uint64_t) (uintptr_t) a.ext3cnt + (uint64_t) ib * (uint64_t) (kCntW * 2);
        static_assert(kATile / 4 == 32 && kBStage / 4 == 64 && kCntW / 8 == 40, "lane counts of the three requests");
        uint32_t keepM0; uint64_t keepExec;
        asm volatile("s_waitcnt lgkmcnt(0)\n\t"                      // every read of the staging area issued so far has returned
                     "s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\t"
                     "s_mov_b32 m0, %6\n\ts_mov_b32 exec_hi, 0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\t"
                     "s_mov_b32 m0, %7\n\ts_mov_b32 exec_hi, -1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %4\n\t"
                     "s_mov_b32 m0, %8\n\ts_mov_b32 exec_hi, 0xff\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %5\n\t"
                     "s_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keepM0), "=&s"(keepExec) : "v"(dmaOff), "s"(bA), "s"(bB), "s"(bC), "s"(stALds), "s"(stBLds), "s"(cntLds) : "memory");
        stagedA = ia; stagedB = ib; stagedLanded = false;
      }
    };
    for (;;) {
        if (nxt < 0) {
            uint32_t nC; bool heavyChunk = false;
            if (hNext < nHeavy) { c0 = uni(heavyList[hNext]); nC = 1; hNext += listBlocks * kWavesPerBlock; heavyChunk = true; }
            else if (sNext < staticEnd) { c0 = sNext; sNext += nWavesTot * kQueryChunk; nC = kQueryChunk; }
            else { c0 = staticEnd + popQueue(a.queue, lane) * a.dynChunk; nC = a.dynChunk; if (c0 >= a.nq) break; }
            if (c0 + nC > a.nq) nC = a.nq - c0;
            if constexpr (kSmall) offL = (uint32_t) lane <= nC ? a.qoff[c0 + lane] : 0u;
            const uint32_t flag = (!heavyChunk && (uint32_t) lane < nC) ? (uint32_t) a.heavyFlag[c0 + lane] : (lane == 0 && heavyChunk ? 0u : 1u);
            todo = __ballot(flag == 0u);                  // (the heavy queries of a dealt chunk were served first, from the lists)
            if (todo == 0ull) continue;
            nxt = __builtin_ctzll(todo); todo &= todo - 1ull;
            if constexpr (kSmall) pkN = header(nxt);
        }
        const int cur = nxt;
        const uint32_t qi = c0 + (uint32_t) cur;
        const uint32_t qo = kSmall ? (uint32_t) __builtin_amdgcn_readlane((int) offL, cur) : uni(a.qoff[qi]);
        const int L = (int) ((kSmall ? (uint32_t) __builtin_amdgcn_readlane((int) offL, cur + 1) : uni(a.qoff[qi + 1])) - qo);
        const uint8_t *q = a.qres + qo;
        const int nScan = L - kSpan;        // windows 0 .. L-9 ; the last window (L-8) is never scanned
        uint32_t M = 0, ringBase = 0;
        uint32_t qK0 = 0;                   // similar k-mers of the windows before the current one
        bool overflow = false;
        // residues and thresholds of the first 64 positions live in lanes (readlane per window)
        // (the large footprint has no register to spare for the header of the query after this one: it fetches its own)
        const int myPk = kSmall ? pkN : (lane < L ? ((int) q[lane] | ((int) a.qthr[qo + lane] << 8)) : -256);
        // per position: number of the 3-mer (i, i+1, i+3) | threshold << 16 -- a window reads its two rows' numbers with two readlanes (the b half of window
        // i is the a half of position i + 4); built once per query from the residues in the lanes (wave_shl DPP)
        const int myW = (kSmall && wNValid) ? wN : windowWords(myPk);
        wNValid = false;
        nxt = todo ? __builtin_ctzll(todo) : -1;
        if (nxt >= 0) { todo &= todo - 1ull; if constexpr (kSmall) pkN = header(nxt); }
        // the windows whose 8 positions lie inside the first 64 (all of them for a spacer peptide) and that have a threshold
        const int nLaneWin = nScan < 65 - kSpan ? nScan : 65 - kSpan;
        uint64_t wm = __ballot(lane < nLaneWin && myW >= 0);
        int iLong = 65 - kSpan;             // (queries longer than 64: the remaining windows fetch their positions one by one)
        auto rowsOf = [&](int w, int i, uint32_t &ia, uint32_t &ib) {
            ia = (uint32_t) __builtin_amdgcn_readlane(w, i) & 0xFFFFu;
            ib = (uint32_t) __builtin_amdgcn_readlane(w, i + 4) & 0xFFFFu;
        };
        SP_T(0);

        // the exact pre-filter of the tail (below) works on two bitmaps of hashed target ids, "seen" and "seen again".  The large footprint (dense
        // index: 2 000+ hits per query, most of them spilled to HBM) marks them WHILE the hits are emitted -- the occupied-slot list whose LDS they
        // share is not used on that path -- so the tail reads the spilled hits once (to sift them) instead of twice (mark, then sift).
        uint32_t *seen = occList, *again = occList + kSuper / 2;     // 2 x kSuper / 2 words: 16 384 (large) or 8 192 (small) buckets each
        constexpr int kBktShift = kSuper == 1024 ? 18 : 19;
        constexpr uint32_t kMul0 = 2654435761u;                       // the first round's multiplier
        if constexpr (!kSmall) { for (uint32_t t = lane * 4; t < (uint32_t) kSuper; t += 256) *reinterpret_cast<uint4 *>(occList + t) = make_uint4(0u, 0u, 0u, 0u); waveSync(); }
        // Dense index: a query has thousands of hits and the LDS buffer holds kHitCap of them.  The buffer is a window [ringBase, ringBase + kHitCap) of the
        // hit stream: when the next batch of lists does not fit, the window's content goes to the per-wave HBM scratch in whole lines (8 bytes per lane,
        // consecutive) and the window moves on -- the hits themselves are emitted into LDS.  (Stored from the emission directly, a list entry per lane at a
        // stride of the list length, the same bytes cost three partial writes per line: 0.9 of 3.1 ms at 2 x 10^8 residues.)  In the scratch the stream lies
        // kHitCap entries behind its position, so that the pre-filter below can compact in place: LDS content first, then the scratch.
        constexpr bool kRing = !kSmall && !kOrd;
        auto emit = [&](uint64_t ent, uint32_t h, uint32_t ord, int win) {
            const uint32_t seqId = (uint32_t) ent;
            const uint32_t diag = (uint32_t) (win - (int) (uint32_t) ((ent >> 32) & 0xFFFF)) & 0xFFFFu;
            const uint64_t key = ((uint64_t) seqId << 32) | ((uint64_t) (h & 0xFFFFu) << 16) | diag;
            if (h - ringBase < (uint32_t) kHitCap) { keys[h - ringBase] = key; if (kOrd) ordL[h] = ord; }
#if !(defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 4)
            else if (h < (uint32_t) kSpillCap) { spillKeys[kRing ? h + (uint32_t) kHitCap : h] = key; if (kOrd) spillOrd[h] = ord; }
#endif
#if defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 6
            return;
#endif
            if constexpr (!kSmall) {
                if (h < (uint32_t) kSpillCap) { const uint32_t bkt = (seqId * kMul0) >> kBktShift, bit = 1u << (bkt & 31u); if (atomicOr(&seen[bkt >> 5], bit) & bit) atomicOr(&again[bkt >> 5], bit); }
            }
        };
        // Multi-entry lists are rare while the index is sparse (3 % of the occupied slots at phage scale) but walking them
        // right away costs every batch of slots another dependent memory round trip: their slot words are parked in LDS
        // (the stream positions of their hits are already assigned -- the slot word holds the length) and the entries of up
        // to kDeferCap lists are fetched together.
        uint32_t defN = 0;
        auto flushDeferred = [&]() {
            waveSync();
            uint64_t sw[kDeferCap / 64]; uint32_t mt[kDeferCap / 64], od[kDeferCap / 64], ln[kDeferCap / 64], lenMax = 0;
#pragma unroll
            for (int r = 0; r < kDeferCap / 64; r++) {
                const uint32_t t = 64u * r + lane;
                sw[r] = 0ull; mt[r] = 0u; od[r] = 0u; ln[r] = 0u;
                if (t < defN) { sw[r] = defSlot[t]; mt[r] = defMeta[t]; if (kOrd) od[r] = defOrd[t]; ln[r] = (uint32_t) (sw[r] >> 32) & 0x7FFFFFFFu; }
                lenMax = ln[r] > lenMax ? ln[r] : lenMax;
            }
            for (uint32_t e0 = kSmall ? 0u : 3u; e0 < lenMax; e0 += 4) {      // (bucket layout: the first three entries of a list were emitted with its bucket)
                uint64_t ent[4][kDeferCap / 64];
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < kDeferCap / 64; r++) ent[u][r] = (e0 + u < ln[r]) ? a.ientries[(uint32_t) sw[r] + e0 + u] : 0ull;
#pragma unroll
                for (int u = 0; u < 4; u++)
#pragma unroll
                    for (int r = 0; r < kDeferCap / 64; r++) if (e0 + u < ln[r]) emit(ent[u][r], (mt[r] & 0xFFFFu) + e0 + u, od[r], (int) (mt[r] >> 16));
            }
            defN = 0;
            waveSync();
        };
        auto moveRing = [&](uint32_t upTo) {      // the window's content [ringBase, upTo) -> scratch, in whole lines; the window continues at upTo
            waveSync();
            const uint32_t cnt = upTo - ringBase < (uint32_t) kHitCap ? upTo - ringBase : (uint32_t) kHitCap;      // (what lies beyond went to the scratch directly)
#if !(defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 7)
            for (uint32_t t = lane; t < cnt; t += 64) if (ringBase + t < (uint32_t) kSpillCap) spillKeys[(uint32_t) kHitCap + ringBase + t] = keys[t];
#endif
            waveSync();
            ringBase = upTo;
        };
        for (;;) {
            int i, thr; uint32_t ia, ib;
            if (wm) {
                i = __builtin_ctzll(wm); wm &= wm - 1ull;
                { const int v = __builtin_amdgcn_readlane(myW, i); thr = v >> 16; ia = (uint32_t) v & 0xFFFFu; ib = (uint32_t) __builtin_amdgcn_readlane(myW, i + 4) & 0xFFFFu; }
            } else if (iLong < nScan) {
                i = iLong++;
                thr = (int) uni((uint32_t) (int) a.qthr[qo + i]);
                if (thr < 0) continue;
                ia = uni(q[i] + 20u * q[i + 1] + 400u * q[i + 3]);
                ib = uni(q[i + 4] + 20u * q[i + 5] + 400u * q[i + 7]);
            } else break;
            const uint32_t *rowA = a.ext3 + (size_t) ia * 8000u;
            const uint32_t *rowB = a.ext3 + (size_t) ib * 8000u;
            // Small footprint: the rows of the NEXT window (of this query, or the first one of the next query of the chunk) are requested as soon as this
            // window has no more use for the staging area -- after its last mapping phase -- and land while its slots are fetched: a window starts from LDS.
            bool requested = false;
            auto requestNext = [&]() {
                if constexpr (kSmall) {
                    if (requested) return;
                    requested = true;
                    uint32_t na, nb;
                    if (wm) { rowsOf(myW, __builtin_ctzll(wm), na, nb); stage(na, nb); }
                    else if (iLong >= nScan && nxt >= 0) {
                        const int Ln = (int) ((uint32_t) __builtin_amdgcn_readlane((int) offL, nxt + 1) - (uint32_t) __builtin_amdgcn_readlane((int) offL, nxt));
                        const int nl = Ln - kSpan < 65 - kSpan ? Ln - kSpan : 65 - kSpan;
                        wN = windowWords(pkN); wNValid = true;
                        const uint64_t wn = __ballot(lane < nl && wN >= 0);
                        if (wn) { rowsOf(wN, __builtin_ctzll(wn), na, nb); stage(na, nb); }
                    }
                }
            };
            int sA0, sB0; uint32_t vA0[kATile / 64];
            if constexpr (kSmall) {
                if (stagedA != ia || stagedB != ib) stage(ia, ib);      // (the first window of a chunk, windows beyond position 64)
                if (!stagedLanded) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stagedLanded = true; }      // the LDS-DMA of this wave has landed
                waveSync();
                sB0 = (int) uni((uint32_t) extScore(stB[0]));
                sA0 = (int) uni((uint32_t) extScore(stA[0]));
#pragma unroll
                for (int j = 0; j < kATile / 64; j++) vA0[j] = 0u;
            } else {
                // ---- one batch of independent loads
                const uint16_t *cntRow = a.ext3cnt + (size_t) ib * kCntW;
#pragma unroll
                for (int j = 0; j < kATile / 64; j++) vA0[j] = rowA[j * 64 + lane];
                const uint32_t vB0 = rowB[lane], vB1 = rowB[64 + lane];
                uint16_t cw[kCntW / 64];
#pragma unroll
                for (int j = 0; j < kCntW / 64; j++) cw[j] = cntRow[j * 64 + lane];
                sB0 = (int) uni((uint32_t) extScore(vB0));
                sA0 = (int) uni((uint32_t) extScore(vA0[0]));
                // "how many b reach cutoff c": precomputed per-row count table
#pragma unroll
                for (int j = 0; j < kCntW / 64; j++) cntB[j * 64 + lane] = cw[j];
                bIdx[lane] = (uint16_t) extIndex(vB0); bIdx[64 + lane] = (uint16_t) extIndex(vB1);
            }
            const int cutoff1 = (short) (thr - sB0);
            if (sA0 < cutoff1) { requestNext(); continue; }
            const int minCut2 = (short) (thr - sA0);
            SP_T(1);
            waveSync();
            int cAt = minCut2 - a.cntMin; cAt = cAt < 0 ? 0 : (cAt >= kCntW ? kCntW - 1 : cAt);
            const uint32_t nBmax = uni((uint32_t) cntB[cAt]);
            if constexpr (!kSmall) { for (uint32_t p = 128 + lane; p < nBmax && p < (uint32_t) kBTile; p += 64) bIdx[p] = (uint16_t) extIndex(rowB[p]); }
            waveSync();
            // ---- tiles of row A, flattened (a, b) enumeration in the reference's list order
            uint32_t kBase = 0;
            bool lastTile = false;
            for (uint32_t a0 = 0; a0 < 8000u && !lastTile; a0 += kATile) {
                uint32_t tileK = 0, nATile = kATile;
                uint32_t vA[kATile / 64], nbR[kATile / 64], aStR[kATile / 64];
#pragma unroll
                for (int j = 0; j < kATile / 64; j++) { nbR[j] = 0; aStR[j] = 0; }
#pragma unroll
                for (int j = 0; j < kATile / 64; j++) { const uint32_t ai = a0 + j * 64 + lane; vA[j] = a0 == 0 ? (kSmall ? stA[j * 64 + lane] : vA0[j]) : (ai < 8000u ? rowA[ai] : 0u); }
#pragma unroll
                for (int j = 0; j < kATile / 64; j++) {
                    const uint32_t ai = a0 + j * 64 + lane;
                    const uint32_t v = vA[j];
                    const int s = ai < 8000u ? extScore(v) : -32768;
                    const bool ok = s >= cutoff1;
                    uint32_t nb = 0;
                    if (ok) { int at = (short) (thr - s) - a.cntMin; at = at < 0 ? 0 : (at >= kCntW ? kCntW - 1 : at); nb = cntB[at]; }
                    const uint32_t incl = waveInclusiveScan(nb, lane);
                    nbR[j] = nb; aStR[j] = tileK + incl - nb;
                    aStart[j * 64 + lane] = tileK + incl - nb;
                    aIdx[j * 64 + lane] = (uint16_t) extIndex(v);
                    tileK += lane63(incl);
                    const uint64_t okMask = __ballot(ok);
                    if (okMask != ~0ull) { lastTile = true; nATile = j * 64 + __popcll(okMask); break; }
                }
                if (lane == 0) aStart[nATile] = tileK;
                waveSync();
                if (kBase + tileK > kMaxKmerResult - 1) { tileK = kMaxKmerResult - 1 - kBase; lastTile = true; }
                SP_T(2);
                // Flattened (a, b) enumeration, 64 similar k-mers per round.  Lane L owns position k = k0 + L of the list and
                // needs the largest a with aStart[a] <= k.  Every a contributes >= 1 k-mer, so a(k) = (number of list starts at
                // positions <= k) - 1: the lanes that own the a's set one bit per start in a 1024-bit map in LDS, each round
                // broadcasts its 64-bit word of the map through a register lane, and a popcount gives the a -- no search and no
                // LDS round trip per round.
                // Enumeration in super-rounds of 1024 similar k-mers so that memory round trips are paid per 1024, not per 64:
                //   phase 1: map the 16 x 64 flattened positions to k-mer ids (registers), fetch their probe words with 16
                //            independent loads, compact the ranks of the occupied slots (14 % for a phage-scale index) into LDS
                //            in list order
                //   phase 2: 256 occupied slots at a time: 4 independent slot-word loads per lane; one-entry lists are
                //            finished by that load, longer lists walk the entry array
                for (uint32_t s0 = 0; s0 < tileK; s0 += kSuper) {
                    if (lane < kSuper / 64) hflag[lane] = 0ull;
                    waveSync();
                    uint32_t aBefore = 0;
#pragma unroll
                    for (int j = 0; j < kATile / 64; j++) {
                        const uint32_t pos = aStR[j] - s0;
                        if (nbR[j] > 0 && pos < (uint32_t) kSuper) atomicOr(&hflag[pos >> 6], 1ull << (pos & 63u));
                        aBefore += (uint32_t) __popcll(__ballot(nbR[j] > 0 && aStR[j] < s0));
                    }
                    waveSync();
                    const unsigned long long myW = lane < kSuper / 64 ? hflag[lane] : 0ull;
#if defined(SP_ABLATE) && SP_ABLATE >= 4
                    if (myW == 0x123456789abcull) a.counters[7] = 1;      // (keeps the head flags alive; no mapping rounds)
                    if (lastTile && s0 + (uint32_t) kSuper >= tileK) requestNext();
                    continue;
#endif
                    uint32_t carry = 0;
                    uint32_t kmerR[kSuper / 64];
                    // The rounds are mapped in groups of kMapGroup WITHOUT per-round branches: three dependent LDS lookups per round
                    // (list start of the a, partner index, a's 3-mer) cost a few hundred cycles when every round waits for its own;
                    // issued for kMapGroup rounds back to back they overlap.  Lanes beyond the list compute with the last a (their map
                    // bits are zero) and are masked at the end; partners beyond the LDS copy of row B (rare) are patched afterwards.
#pragma unroll
                    for (int g = 0; g < kSuper / 64; g += kMapGroup) {
                        if (g > 0 && s0 + 64u * (uint32_t) g >= tileK) {
#pragma unroll
                            for (int r = g; r < g + kMapGroup; r++) kmerR[r] = 0xFFFFFFFFu;
                            continue;
                        }
                        uint32_t aiG[kMapGroup], bbG[kMapGroup];
#pragma unroll
                        for (int r = 0; r < kMapGroup; r++) {
                            const unsigned long long word = ((unsigned long long) (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) (myW >> 32), g + r) << 32) |
                                                            (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) myW, g + r);
                            uint32_t ai = aBefore + carry + (uint32_t) __popcll(word & (~0ull >> (63 - lane))) - 1u;
                            carry += (uint32_t) __popcll(word);
                            aiG[r] = ai < nATile ? ai : nATile - 1u;         // (only lanes in front of the first list start can be out of range: never inside the list)
                        }
#pragma unroll
                        for (int r = 0; r < kMapGroup; r++) bbG[r] = aStart[aiG[r]];
                        constexpr uint32_t kBLds = kSmall ? kBStage : kBTile;      // partners whose index is in LDS
                        uint32_t far = 0;
#pragma unroll
                        for (int r = 0; r < kMapGroup; r++) {
                            const uint32_t k = s0 + 64u * (uint32_t) (g + r) + (uint32_t) lane;
                            bbG[r] = k - bbG[r];
                            far |= (k < tileK && bbG[r] >= kBLds) ? 1u : 0u;
                        }
                        uint32_t biG[kMapGroup], a3G[kMapGroup];
#pragma unroll
                        for (int r = 0; r < kMapGroup; r++) {
                            if constexpr (kSmall) biG[r] = extIndex(stB[bbG[r] < kBLds ? bbG[r] : 0u]); else biG[r] = bIdx[bbG[r] < kBLds ? bbG[r] : 0u];
                            a3G[r] = aIdx[aiG[r]];
                        }
                        if (__ballot(far != 0u) != 0ull) {
#pragma unroll
                            for (int r = 0; r < kMapGroup; r++) {
                                const uint32_t k = s0 + 64u * (uint32_t) (g + r) + (uint32_t) lane;
                                if (k < tileK && bbG[r] >= kBLds) biG[r] = extIndex(rowB[bbG[r]]);
                            }
                        }
#pragma unroll
                        for (int r = 0; r < kMapGroup; r++) {
                            const uint32_t k = s0 + 64u * (uint32_t) (g + r) + (uint32_t) lane;
                            kmerR[g + r] = k < tileK ? slotOfKmer(a3G[r], biG[r]) : 0xFFFFFFFFu;
                        }
                    }
                    SP_T(3);
                    if (lastTile && s0 + (uint32_t) kSuper >= tileK) requestNext();      // (the staging area is free from here on)
#if defined(SP_ABLATE) && SP_ABLATE >= 3
                    { uint32_t acc = 0;
#pragma unroll
                      for (int r = 0; r < kSuper / 64; r++) acc ^= kmerR[r];
                      if (acc == 0x12345u) a.counters[7] = 1; }      // (keeps the mapping alive)
                    continue;
#endif
                    if constexpr (!kSmall) {
                        // Dense index (the large footprint): a DIRECT table of 32-byte buckets, one per k-mer slot, read with the slot number itself.  At
                        // 2 x 10^8 residues 95 % of the slots are occupied and lists average three entries: the occupancy probe (a quarter of the L2 misses
                        // there: the 8 MB bit table does not fit one XCD's L2), the rank lookup, the compaction of the occupied slots and the second random
                        // access per list (slot word -> entry array) all go.  A bucket is 16 halfwords: h0 = length (0..5) and up to FIVE entries of 6 bytes
                        // {position, target id} -- or h0 = 0xFFFF, three entries, the length in words 5 and the list's offset in word 7 for the 9 % of the
                        // lists that are longer: only those cost a second, dependent access (with 8-byte entries three fit and 37 % of the lists paid it --
                        // a quarter of the kernel).
                        // (the buckets of the NEXT batch are in flight while this one is emitted: a super-round has up to eight batches)
                        uint64_t hdN[kOccRegs], e0N[kOccRegs], e1N[kOccRegs], e2N[kOccRegs];
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) {
                            hdN[r] = 0ull; e0N[r] = 0ull; e1N[r] = 0ull; e2N[r] = 0ull;
                            const uint32_t km = kmerR[r];
                            if (km != 0xFFFFFFFFu) {
                                const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(a.buckets + 4ull * km);
                                const ulonglong2 v0 = bp[0], v1 = bp[1];
                                hdN[r] = v0.x; e0N[r] = v0.y; e1N[r] = v1.x; e2N[r] = v1.y;
                            }
                        }
#pragma unroll
                        for (int g = 0; g < kSuper / 64; g += kOccRegs) {
                            if (s0 + 64u * (uint32_t) g >= tileK) break;                   // (uniform: the rounds beyond the list hold nothing)
                            uint64_t hdR[kOccRegs], e0R[kOccRegs], e1R[kOccRegs], e2R[kOccRegs]; uint32_t lenR[kOccRegs], hR[kOccRegs], ordR[kOccRegs];
#pragma unroll
                            for (int r = 0; r < kOccRegs; r++) {
                                hdR[r] = hdN[r]; e0R[r] = e0N[r]; e1R[r] = e1N[r]; e2R[r] = e2N[r];
                                ordR[r] = 0u;
                                if (kOrd) ordR[r] = qK0 + kBase + s0 + 64u * (uint32_t) (g + r) + (uint32_t) lane;
                            }
                            if (g + kOccRegs < kSuper / 64) {
                                if (s0 + 64u * (uint32_t) (g + kOccRegs) < tileK) {        // (uniform)
#pragma unroll
                                    for (int r = 0; r < kOccRegs; r++) {
                                        hdN[r] = 0ull; e0N[r] = 0ull; e1N[r] = 0ull; e2N[r] = 0ull;
                                        const uint32_t km = kmerR[(g + kOccRegs + r) % (kSuper / 64)];
                                        if (km != 0xFFFFFFFFu) {
                                            const ulonglong2 *bp = reinterpret_cast<const ulonglong2 *>(a.buckets + 4ull * km);
                                            const ulonglong2 v0 = bp[0], v1 = bp[1];
                                            hdN[r] = v0.x; e0N[r] = v0.y; e1N[r] = v1.x; e2N[r] = v1.y;
                                        }
                                    }
                                }
                            }
#if defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 2
                            { uint64_t acc = 0;
#pragma unroll
                              for (int r = 0; r < kOccRegs; r++) acc ^= hdR[r] ^ e0R[r] ^ e1R[r] ^ e2R[r];
                              if (acc == 0x123456789ull) a.counters[7] = 1; }
                            continue;
#endif
#ifdef SP_PHASE_SPLIT
                            SP_T(5); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); SP_T(0);      // p5: issue, p0: wait for the buckets
#endif
                            // unpack: D0..D7 = the bucket's words
                            uint64_t ent[kOccRegs][5]; uint32_t nIn[kOccRegs], offR[kOccRegs]; bool isLong[kOccRegs];
#pragma unroll
                            for (int r = 0; r < kOccRegs; r++) {
                                const uint32_t D0 = (uint32_t) hdR[r], D1 = (uint32_t) (hdR[r] >> 32), D2 = (uint32_t) e0R[r], D3 = (uint32_t) (e0R[r] >> 32);
                                const uint32_t D4 = (uint32_t) e1R[r], D5 = (uint32_t) (e1R[r] >> 32), D6 = (uint32_t) e2R[r], D7 = (uint32_t) (e2R[r] >> 32);
                                const uint32_t h0 = D0 & 0xFFFFu;
                                isLong[r] = h0 == 0xFFFFu;
                                lenR[r] = isLong[r] ? D5 : h0; nIn[r] = isLong[r] ? 3u : h0; offR[r] = D7;
                                ent[r][0] = ((uint64_t) (D0 >> 16) << 32) | D1;
                                ent[r][1] = ((uint64_t) (D2 & 0xFFFFu) << 32) | __builtin_amdgcn_alignbit(D3, D2, 16);
                                ent[r][2] = ((uint64_t) (D3 >> 16) << 32) | D4;
                                ent[r][3] = ((uint64_t) (D5 & 0xFFFFu) << 32) | __builtin_amdgcn_alignbit(D6, D5, 16);
                                ent[r][4] = ((uint64_t) (D6 >> 16) << 32) | D7;
                            }
                            uint32_t lenLong = 0, run = M;
#pragma unroll
                            for (int r = 0; r < kOccRegs; r++) {
                                hR[r] = run;
                                if (s0 + 64u * (uint32_t) (g + r) >= tileK) continue;      // (uniform)
                                const uint32_t incl = waveInclusiveScan(lenR[r], lane);
                                hR[r] = run + incl - lenR[r];
                                run += lane63(incl);
                                if (isLong[r] && lenR[r] > lenLong) lenLong = lenR[r];
                            }
                            if constexpr (kRing) {
                                // this batch does not fit behind what the window holds.  (Parked list tails whose positions lie in the window that leaves go to
                                // the scratch directly when they are fetched -- emit() -- : one list in eleven; flushing them first cost a memory round trip per move)
                                if (run - ringBase > (uint32_t) kHitCap && M > ringBase) moveRing(M);
                            }
#ifdef SP_PHASE_SPLIT
                            SP_T(1);      // p1: scans, window moves (with the flush of parked lists in front of them)
#endif
                            uint32_t nLong = 0;
#pragma unroll
                            for (int r = 0; r < kOccRegs; r++) {
#pragma unroll
                                for (int j = 0; j < 5; j++) if ((uint32_t) j < nIn[r]) emit(ent[r][j], hR[r] + (uint32_t) j, ordR[r], i);
                                nLong += (uint32_t) __popcll(__ballot(isLong[r]));
                            }
#if defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 3
                            nLong = 0;
#endif
#ifdef SP_PHASE_SPLIT
                            SP_T(2);      // p2: emission
#endif
                            // tails of lists longer than five entries: parked (their stream positions are assigned) and fetched together, kDeferCap lists
                            // at a time -- walking the ~20 long lists of a batch on the spot costs as many dependent round trips as the longest of them has steps
                            if (nLong > 0 && nLong <= (uint32_t) kDeferCap && defN + nLong > (uint32_t) kDeferCap) flushDeferred();
                            if (nLong > (uint32_t) kDeferCap) {
                                for (uint32_t e0 = 3; e0 < lenLong; e0 += kWalkStep) {      // (never with two buckets per lane: a batch has 128 lists)
                                    uint64_t entW[kWalkStep][kOccRegs];
#pragma unroll
                                    for (int u = 0; u < kWalkStep; u++)
#pragma unroll
                                        for (int r = 0; r < kOccRegs; r++) entW[u][r] = (isLong[r] && e0 + u < lenR[r]) ? a.ientries[offR[r] + e0 + u] : 0ull;
#pragma unroll
                                    for (int u = 0; u < kWalkStep; u++)
#pragma unroll
                                        for (int r = 0; r < kOccRegs; r++) if (isLong[r] && e0 + u < lenR[r]) emit(entW[u][r], hR[r] + e0 + u, ordR[r], i);
                                }
                            } else if (nLong > 0) {
#pragma unroll
                                for (int r = 0; r < kOccRegs; r++) {
                                    const bool longList = isLong[r] && hR[r] < (uint32_t) kSpillCap;
                                    const uint64_t m = __ballot(longList);
                                    if (longList) {
                                        const uint32_t at = defN + lanePrefixCount(m);
                                        defSlot[at] = (1ull << 63) | ((uint64_t) lenR[r] << 32) | offR[r]; defMeta[at] = hR[r] | ((uint32_t) i << 16); if (kOrd) defOrd[at] = ordR[r];
                                    }
                                    defN += (uint32_t) __popcll(m);
                                }
                            }
                            M = run;
#ifdef SP_PHASE_SPLIT
                            SP_T(3);      // p3: parking / flushing list tails
#endif
                        }
                        waveSync();
                        SP_T(5);
                        continue;
                    }
                    uint64_t prR[kSuper / 64];
#pragma unroll
                    for (int r = 0; r < kSuper / 64; r++) prR[r] = (kmerR[r] != 0xFFFFFFFFu) ? a.pbits[kmerR[r] >> 6] : 0ull;
                    uint32_t nOcc = 0;
#pragma unroll
                    for (int r = 0; r < kSuper / 64; r++) {
                        if (s0 + 64u * r >= tileK) break;                          // (uniform: the rounds beyond the list hold nothing)
                        const uint32_t bit = kmerR[r] & 63u;
                        const bool occ = kmerR[r] != 0xFFFFFFFFu && ((prR[r] >> bit) & 1ull);
                        const uint64_t m = __ballot(occ);
                        if (occ) {
                            const uint32_t at = nOcc + lanePrefixCount(m);
                            // slot number (< 2^26) | occupied slots of its word below it: the rest of the rank is looked up in phase 2
                            occList[at] = kmerR[r] | ((uint32_t) __popcll(prR[r] & ((1ull << bit) - 1ull)) << 26);
                            if (kOrd) occOrd[at] = (uint16_t) (r * 64 + lane);
                        }
                        nOcc += (uint32_t) __popcll(m);
                    }
                    waveSync();
                    SP_T(4);
#if defined(SP_ABLATE) && SP_ABLATE >= 2
                    if (nOcc == 0x7FFFFFFFu) a.counters[7] = 1;
                    continue;
#endif
                    for (uint32_t ob = 0; ob < nOcc; ob += 64u * kOccRegs) {
                        uint64_t slotR[kOccRegs]; uint32_t lenR[kOccRegs], hR[kOccRegs], ordR[kOccRegs], rankR[kOccRegs];
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) { slotR[r] = 0ull; lenR[r] = 0u; ordR[r] = 0u; rankR[r] = 0u; }
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) {
                            const uint32_t at = ob + 64u * r + lane;
                            if (at < nOcc) {
                                const uint32_t e = occList[at], km = e & 0x3FFFFFFu;
                                rankR[r] = a.prow[km / 8000u] + (uint32_t) a.prank[km >> 6] + (e >> 26);      // rank of the slot
                                if (kOrd) ordR[r] = qK0 + kBase + s0 + occOrd[at];
                            }
                        }
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) {
                            if (ob + 64u * r + lane < nOcc) { slotR[r] = a.slots[rankR[r]]; lenR[r] = (slotR[r] >> 63) ? ((uint32_t) (slotR[r] >> 32) & 0x7FFFFFFFu) : 1u; }
                        }
                        uint32_t lenMulti = 0, run = M;
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) {
                            hR[r] = run;
                            if (ob + 64u * r >= nOcc) continue;                    // (uniform: no slot in this register)
                            const uint32_t incl = waveInclusiveScan(lenR[r], lane);
                            hR[r] = run + incl - lenR[r];
                            run += lane63(incl);
                            if ((slotR[r] >> 63) && lenR[r] > lenMulti) lenMulti = lenR[r];
                        }
                        // one-entry lists (the rule while the index is sparse) are complete with the slot word
                        uint32_t nMulti = 0;
#pragma unroll
                        for (int r = 0; r < kOccRegs; r++) {
                            if (lenR[r] == 1u && !(slotR[r] >> 63)) emit(slotR[r], hR[r], ordR[r], i);
                            nMulti += (uint32_t) __popcll(__ballot((slotR[r] >> 63) != 0ull));
                        }
                        if (nMulti > (uint32_t) kDeferCap / 4 || defN + nMulti > (uint32_t) kDeferCap) {
                            // dense index (or the parking area is full): most lists are long, the batch amortises the walk by itself.  Four list positions per
                            // step: 16 independent entry loads in flight (lists of 3 on average, the longest of 256 decides the
                            // trip count)
                            for (uint32_t e0 = 0; e0 < lenMulti; e0 += kWalkStep) {
                                uint64_t ent[kWalkStep][kOccRegs];
#pragma unroll
                                for (int u = 0; u < kWalkStep; u++)
#pragma unroll
                                    for (int r = 0; r < kOccRegs; r++) ent[u][r] = ((slotR[r] >> 63) && e0 + u < lenR[r]) ? a.ientries[(uint32_t) slotR[r] + e0 + u] : 0ull;
#pragma unroll
                                for (int u = 0; u < kWalkStep; u++)
#pragma unroll
                                    for (int r = 0; r < kOccRegs; r++) if ((slotR[r] >> 63) && e0 + u < lenR[r]) emit(ent[u][r], hR[r] + e0 + u, ordR[r], i);
                            }
                        } else if (nMulti > 0) {
#pragma unroll
                            for (int r = 0; r < kOccRegs; r++) {
                                const bool multi = (slotR[r] >> 63) != 0ull && hR[r] < (uint32_t) kSpillCap;
                                const uint64_t m = __ballot(multi);
                                if (multi) {
                                    const uint32_t at = defN + lanePrefixCount(m);
                                    defSlot[at] = slotR[r]; defMeta[at] = hR[r] | ((uint32_t) i << 16); if (kOrd) defOrd[at] = ordR[r];
                                }
                                defN += (uint32_t) __popcll(m);
                            }
                        }
                        M = run;
                    }
                    waveSync();
                    SP_T(5);
                }
                kBase += tileK;
                waveSync();
            }
            requestNext();
            kmersTotal += kBase;
            qK0 += kBase;
        }
        if (defN) flushDeferred();
        if constexpr (kRing) { if (M - ringBase > (uint32_t) kHitCap) moveRing(M); }      // (an oversized last batch: the scratch holds [0, ringBase) without a gap)
        // The rows of the NEXT query's first window were requested during this query's last window; everything requested since has been waited for (memory
        // operations retire in order), so this wait costs nothing -- and it must stand HERE: the same wait at the next window's start would also wait for the
        // record stores of this query's tail, issued a few hundred cycles before it (gfx9 counts stores in vmcnt), a full write latency per query.
        if constexpr (kSmall) { if (!stagedLanded) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stagedLanded = true; } }
#if (defined(SP_ABLATE) && SP_ABLATE >= 1) || (defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE >= 3 && SP_ABLATE_DENSE <= 7)
        M = 0;
#endif
        matchesTotal += M;
        if (M > (uint32_t) kSpillCap) { overflow = true; M = 0; }
        uint32_t D = 0, dbase = 0;
        if (M > 0) {
            // Cheap exact pre-filter before the sort: a target hit only once in this query can only be recorded when its
            // 8-bit diagonal is 0 (previous := 0).  Two bitmaps of 16 384 hashed targets in LDS -- "seen" and "seen again" --
            // keep the hits whose bucket was hit at least twice or whose diag8 == 0; bucket collisions merely let a few extra
            // hits through to the exact automaton (0.7 % at 100 hits per query, 13 % at 2 000).  The filter is exact under ANY
            // hash, so while the survivors do not fit the register path it runs again on them with another hash: on a dense
            // index (2 000 - 60 000 hits per query) the chance collisions fall geometrically (2 200 -> 300 -> 120 -> the true
            // double hits) and nothing is ever sorted in HBM.
            // (hits beyond the LDS buffer live in the per-wave HBM scratch; the survivors usually fit LDS again)
#pragma nounroll
            for (int round = 0; round < 6; round++) {
                const uint32_t mul = uni(kMul0 + (uint32_t) round * 0x3C6EF372u);      // odd multipliers, a different one per round
                const bool marked = !kSmall && round == 0;      // (large footprint: the first round's bitmaps were filled while the hits were emitted)
                if (!marked) { for (uint32_t t = lane * 4; t < (uint32_t) kSuper; t += 256) *reinterpret_cast<uint4 *>(occList + t) = make_uint4(0u, 0u, 0u, 0u); }
                waveSync();
                // hits beyond the LDS buffer come from the per-wave HBM scratch: those chunks are read kTailKeys keys per lane at a
                // time (independent loads; one load per step would pay the memory latency 35 times for 2 200 hits)
                auto mark = [&](uint64_t k) {
                    const uint32_t bkt = ((uint32_t) (k >> 32) * mul) >> kBktShift, bit = 1u << (bkt & 31u);
                    if (atomicOr(&seen[bkt >> 5], bit) & bit) atomicOr(&again[bkt >> 5], bit);
                };
                // (kRing, first round: LDS holds the END of the stream, [ringBase, M), the scratch [0, ringBase) at kHitCap + position; the survivors are
                // compacted into the usual layout -- keys[0 .. kHitCap), then spillKeys[kHitCap ..) -- and a survivor's place is never ahead of the reader)
                const bool ring0 = kRing && round == 0;
                const uint32_t mLds = ring0 ? M - ringBase : (M < (uint32_t) kHitCap ? M : (uint32_t) kHitCap);
#if defined(SP_ABLATE_DENSE) && SP_ABLATE_DENSE == 8
                const uint32_t spEnd = (uint32_t) kHitCap;
#else
                const uint32_t spEnd = ring0 ? (uint32_t) kHitCap + ringBase : M;
#endif
                if (!marked) {
                    for (uint32_t t = lane; t < mLds; t += 64) mark(keys[t]);
                    for (uint32_t base = (uint32_t) kHitCap; base < spEnd; base += 64u * kTailKeys) {
                        uint64_t k4[kTailKeys];
#pragma unroll
                        for (int u = 0; u < kTailKeys; u++) { const uint32_t t = base + 64u * u + lane; k4[u] = t < spEnd ? spillKeys[t] : 0ull; }
#pragma unroll
                        for (int u = 0; u < kTailKeys; u++) if (base + 64u * u + lane < spEnd) mark(k4[u]);
                    }
                    waveSync();
                }
                uint32_t wr = 0;
                auto sift = [&](uint64_t k, bool valid) {
                    const uint32_t bkt = ((uint32_t) (k >> 32) * mul) >> kBktShift;
                    const bool keep = valid && (((again[bkt >> 5] >> (bkt & 31u)) & 1u) || (k & 0xFF) == 0);
                    const uint64_t m = __ballot(keep);
                    if (keep) { const uint32_t at = wr + lanePrefixCount(m); if (at < (uint32_t) kHitCap) keys[at] = k; else spillKeys[at] = k; }
                    wr += (uint32_t) __popcll(m);
                };
                // the scratch is read two steps ahead of the sifting (a step = kSiftKeys keys per lane): a query of the dense index has a dozen such steps, and
                // waiting out every one of them was a third of the kernel
                constexpr int kSiftKeys = kTailKeys;
                uint64_t kA[kSiftKeys], kB[kSiftKeys];
                auto fetch = [&](uint64_t (&k)[kSiftKeys], uint32_t base) {
#pragma unroll
                    for (int u = 0; u < kSiftKeys; u++) { const uint32_t t = base + 64u * u + lane; k[u] = t < spEnd ? spillKeys[t] : 0ull; }
                };
                auto siftStep = [&](const uint64_t (&k)[kSiftKeys], uint32_t base) {
#pragma unroll
                    for (int u = 0; u < kSiftKeys; u++) { if (base + 64u * u >= spEnd) break; sift(k[u], base + 64u * u + lane < spEnd); }
                };
                constexpr uint32_t kStep = 64u * kSiftKeys;
                if ((uint32_t) kHitCap < spEnd) fetch(kA, (uint32_t) kHitCap);
                if ((uint32_t) kHitCap + kStep < spEnd) fetch(kB, (uint32_t) kHitCap + kStep);
                for (uint32_t base = 0; base < mLds; base += 64) { const uint32_t t = base + lane; sift(t < mLds ? keys[t] : 0ull, t < mLds); }
                for (uint32_t base = (uint32_t) kHitCap; base < spEnd; base += 2u * kStep) {
                    siftStep(kA, base);
                    if (base + 2u * kStep < spEnd) fetch(kA, base + 2u * kStep);
                    if (base + kStep >= spEnd) break;
                    siftStep(kB, base + kStep);
                    if (base + 3u * kStep < spEnd) fetch(kB, base + 3u * kStep);
                }
                waveSync();
                const uint32_t before = M;
                M = wr;
                if (M <= 64u || (uint64_t) M * 8u > (uint64_t) before * 7u) break;      // fits the register path, or what is left are real multi-hit targets
            }
        }
        if constexpr (!kSmall) SP_T(4);      // (dense index: the pre-filter rounds; the sparse footprints count their occupancy probes here)
        const bool spilled = M > (uint32_t) kHitCap;
        // reserve room for D diagonal records: kDiagChunk at a time per wave (one atomic per query on a single word would
        // serialise); unused slots of a chunk stay marked invalid (diagQuery == ~0)
        auto reserve = [&](uint32_t need) {
            if (dChunkUsed + need > dChunkCap) {
                markUnused();
                dChunkCap = need > kDiagChunk ? need : kDiagChunk;
                dChunkBase = uni(atomicAdd(a.diagCursor, lane == 0 ? dChunkCap : 0u));
                dChunkUsed = 0;
            }
            if ((uint64_t) dChunkBase + dChunkCap > a.diagCap) { overflow = true; D = 0; dbase = 0; }      // (the chunk stays unused)
            else { dbase = dChunkBase + dChunkUsed; dChunkUsed += need; diagTotal += need; }
        };
        if (M > 0 && M <= 64u && !spilled) {
            // The usual case after the pre-filter: everything in registers, one hit per lane.  Sort by counting (the keys are
            // distinct), move each key to the lane of its rank, then the automaton is a compare with the left neighbour plus
            // "diag8 of the previous recorded hit of the same target", found with two ballot masks.
            const bool in = (uint32_t) lane < M;
            const uint64_t key = in ? keys[lane] : ~0ull;
            uint32_t rank = 0;
            for (uint32_t u = 0; u < M; u++) {
                const uint64_t ku = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (key >> 32), (int) u) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) key, (int) u);
                rank += ku < key ? 1u : 0u;
            }
            // ds_permute: lane t sends its key to lane rank(t) (lanes >= M keep ~0 by sending to themselves)
            const int dst = (int) ((in ? rank : (uint32_t) lane) << 2);
            const uint32_t sLo = (uint32_t) __builtin_amdgcn_ds_permute(dst, (int) (uint32_t) key), sHi = (uint32_t) __builtin_amdgcn_ds_permute(dst, (int) (key >> 32));
            const uint32_t id = sHi, d8 = sLo & 0xFFu;
            const uint32_t idPrev = (uint32_t) waveShiftUp1((int) id), d8Prev = (uint32_t) waveShiftUp1((int) d8);
            const bool head = in && (lane == 0 || idPrev != id);
            const bool rec = in && (d8 == (head ? 0u : d8Prev));
            const uint64_t recMask = __ballot(rec), headMask = __ballot(head);
            const uint64_t upTo = laneMaskLt(lane) | (1ull << lane);
            const int myHead = 63 - __builtin_clzll((headMask & upTo) | 1ull);
            const uint64_t prior = recMask & laneMaskLt(lane) & ~laneMaskLt(myHead);
            const int lastRecLane = prior ? 63 - __builtin_clzll(prior) : lane;
            const uint32_t lastD = (uint32_t) __shfl((int) d8, lastRecLane);
            const bool kept = rec && (prior == 0ull || lastD != d8);
            SP_T(6);
            const uint64_t keptMask = __ballot(kept);
            D = (uint32_t) __popcll(keptMask);
            if (D > 0) reserve(D);
            if (D > 0 && kept) {
                const uint32_t at = dbase + lanePrefixCount(keptMask);
                DiagRec r; r.seqId = id; r.diag = (uint16_t) (sLo & 0xFFFF); r.idx = (uint16_t) ((sLo >> 16) & 0xFFFF);
                a.diagRec[at] = r;
                a.diagQuery[at] = qi;
                if (kOrd) a.diagOrd[at] = r.idx < (uint32_t) kHitCap ? ordL[r.idx] : spillOrd[r.idx];
            }
        } else if (M > 0) {
            const bool inLds = !spilled;
            uint32_t n = 1; while (n < M) n <<= 1;
            uint64_t *K = keys; uint32_t *B = bitsL;
            if (!inLds) {
                for (uint32_t t = lane; t < (uint32_t) kHitCap; t += 64) spillKeys[t] = keys[t];
                K = spillKeys; B = spillBits;
            }
            for (uint32_t t = M + lane; t < n; t += 64) K[t] = ~0ull;
            for (uint32_t t = lane; t < (n + 31) / 32; t += 64) B[t] = 0;
            waveSync();
            if (inLds) waveBitonicSort(keys, n, lane); else waveBitonicSort(spillKeys, n, lane);
            if (inLds) markKept(keys, M, bitsL, lane); else markKept(spillKeys, M, spillBits, lane);
            waveSync();
            SP_T(6);
            uint32_t cnt = 0;
            for (uint32_t t = lane; t < (M + 31) / 32; t += 64) cnt += __popc(B[t]);
            D = waveSum(cnt);
            if (D > 0) reserve(D);
            if (D > 0) {
                uint32_t outPos = dbase;
                for (uint32_t base = 0; base < M; base += 64) {
                    const uint32_t t = base + lane;
                    const bool kept = (t < M) && ((B[t >> 5] >> (t & 31)) & 1u);
                    const uint64_t m = __ballot(kept);
                    if (kept) {
                        const uint64_t k = K[t];
                        const uint32_t at = outPos + lanePrefixCount(m);
                        DiagRec r; r.seqId = (uint32_t) (k >> 32); r.diag = (uint16_t) (k & 0xFFFF); r.idx = (uint16_t) ((k >> 16) & 0xFFFF);
                        a.diagRec[at] = r;
                        a.diagQuery[at] = qi;
                        if (kOrd) a.diagOrd[at] = r.idx < (uint32_t) kHitCap ? ordL[r.idx] : spillOrd[r.idx];
                    }
                    outPos += __popcll(m);
                }
            }
            waveSync();
        }
        if (lane == 0) {
            a.qDiagBase[qi] = overflow ? 0xFFFFFFFFu : dbase;      // (the marker names the query in the error text; it has no records)
            a.qDiagCnt[qi] = D;
            if (overflow) atomicAdd(&a.counters[2], 1ull);
        }
        SP_T(7);
    }
    SP_T_FLUSH(a.counters);
    if (lane == 0) atomicMax(&a.counters[7], (unsigned long long) wall_clock64());
    markUnused();
    if (lane == 0) { unsigned long long *wp = a.wavePartial + (size_t) waveGlobal * 4; wp[0] = kmersTotal; wp[1] = matchesTotal; wp[2] = diagTotal; wp[3] = wall_clock64() - tWaveStart; }
}

#include "sp_gpu_match_grp.inc"

// ---------------------------------------------------------------------------------------------------------------
// K_score: one lane per double-hit diagonal -- ungapped local score along the diagonal of the MASKED target
//   UngappedAlignment::scalarDiagonalScoring / computeSingelSequenceScores / computeLongScore
//   prefiltering/UngappedAlignment.cpp:37-51, :216-233, :368-382.  The AVX2 u8-saturating path (:67-121) equals
//   min(255 - bias, exact), so the exact score is kept and clamped where the reference clamps.
// ---------------------------------------------------------------------------------------------------------------
struct ScoreArgs {
    const DiagRec *diagRec; const uint32_t *diagQuery; const uint32_t *diagCursor; uint32_t diagCap;
    const uint8_t *qres; const uint32_t *qoff; const signed char *qcorr;
    const uint8_t *tmasked; const uint64_t *toff;
    const signed char *ungSub;
    uint16_t *diagScore;
    unsigned long long *counters;   // [3] += residues read
    const unsigned long long *wavePartial; uint32_t nWaves;      // k_match's per-wave counters, folded into counters[0], [1], [5] here
};

__device__ __forceinline__ int diagRun(const signed char *sub, const uint8_t *q, const signed char *corr, const uint8_t *db, uint32_t len) {
    int mx = 0, sc = 0;
    for (uint32_t p = 0; p < len; p++) {
        sc += sub[q[p] * kAlph + db[p]] + corr[p];
        sc = sc < 0 ? 0 : sc;
        mx = sc > mx ? sc : mx;
    }
    return mx;
}

__device__ int singleDiag(const signed char *sub, const uint8_t *q, const signed char *corr, int qL, const uint8_t *db, uint32_t dbLen,
                          int diagonal, uint32_t minDist, uint32_t &read) {
    if (diagonal >= 0 && minDist < (uint32_t) qL) {
        const uint32_t len = min(dbLen, (uint32_t) qL - minDist);
        read += len;
        return diagRun(sub, q + minDist, corr + minDist, db, len);
    } else if (diagonal < 0 && minDist < dbLen) {
        const uint32_t len = min(dbLen - minDist, (uint32_t) qL);
        read += len;
        return diagRun(sub, q, corr, db + minDist, len);
    }
    return 0;
}

__global__ void k_score(ScoreArgs a) {
    __shared__ signed char s_sub[kAlph * kAlph];
    for (int t = threadIdx.x; t < kAlph * kAlph; t += blockDim.x) s_sub[t] = a.ungSub[t];
    __syncthreads();
    if (blockIdx.x == gridDim.x - 1) {          // (the last block has the fewest diagonals to score)
        __shared__ unsigned long long s_part[3];
        if (threadIdx.x < 3) s_part[threadIdx.x] = 0;
        __syncthreads();
        unsigned long long p0 = 0, p1 = 0, p2 = 0;
        for (uint32_t wv = threadIdx.x; wv < a.nWaves; wv += blockDim.x) { const unsigned long long *wp = a.wavePartial + (size_t) wv * 4; p0 += wp[0]; p1 += wp[1]; p2 += wp[2]; }
        atomicAdd(&s_part[0], p0); atomicAdd(&s_part[1], p1); atomicAdd(&s_part[2], p2);
        __syncthreads();
        if (threadIdx.x == 0) { a.counters[0] += s_part[0]; a.counters[1] += s_part[1]; a.counters[5] += s_part[2]; }
    }
    uint32_t nDiag = *a.diagCursor; if (nDiag > a.diagCap) nDiag = a.diagCap;
    uint32_t read = 0;
    for (uint32_t r = blockIdx.x * blockDim.x + threadIdx.x; r < nDiag; r += gridDim.x * blockDim.x) {
        const uint32_t qi = a.diagQuery[r];
        if (qi == 0xFFFFFFFFu) continue;            // unused slot of a reservation chunk
        const DiagRec rec = a.diagRec[r];
        const uint32_t qo = a.qoff[qi];
        const int qL = (int) (a.qoff[qi + 1] - qo);
        const uint64_t to = a.toff[rec.seqId];
        const uint32_t dbLen = (uint32_t) (a.toff[rec.seqId + 1] - to);
        const uint8_t *db = a.tmasked + to;
        int mx;
        if (qL >= 32768 || dbLen >= 32768) {
            mx = 0;
            for (uint32_t d = 1; d <= 1 + dbLen / 32768; d++) {
                const int real = (int) (-(long long) d * 65536 + rec.diag);
                mx = max(mx, singleDiag(s_sub, a.qres + qo, a.qcorr + qo, qL, db, dbLen, real, (uint32_t) abs(real), read));
            }
            for (uint32_t d = 0; d <= (uint32_t) qL / 65536; d++) {
                const int real = (int) (d * 65536 + rec.diag);
                mx = max(mx, singleDiag(s_sub, a.qres + qo, a.qcorr + qo, qL, db, dbLen, real, (uint32_t) abs(real), read));
            }
        } else {
            const uint16_t nd = (uint16_t) (0 - rec.diag);
            const uint32_t minDist = nd < rec.diag ? nd : rec.diag;
            mx = singleDiag(s_sub, a.qres + qo, a.qcorr + qo, qL, db, dbLen, (int) (short) rec.diag, minDist, read);
        }
        a.diagScore[r] = (uint16_t) (mx > 65535 ? 65535 : mx);
    }
    // one atomic per block (thousands of them on one word serialise at ~12 ns each)
    __shared__ uint32_t s_read;
    if (threadIdx.x == 0) s_read = 0;
    __syncthreads();
    read = waveSum(read);
    if (laneId() == 0 && read) atomicAdd(&s_read, read);
    __syncthreads();
    if (threadIdx.x == 0 && s_read) atomicAdd(&a.counters[3], (unsigned long long) s_read);
}

// ---------------------------------------------------------------------------------------------------------------
// K_select: one wavefront per query -- the tail of QueryMatcher::matchQuery
//   min-score filter, per-target best, histogram threshold, top-N cut, final order
//   prefiltering/QueryMatcher.cpp:100-186,318-381,459-505; QueryMatcher.h:199-209;
//   CacheFriendlyOperations.cpp:311-341 (keepMaxElement)
// Equivalent order-free statement used here (derivation in DESIGN.md): per target keep the diagonal with the
// highest clamped score, ties -> earliest in the hit stream; candidates are ranked by
// (score desc, seqId & (BINSIZE-1) asc, stream position asc); the first maxHits with score >= threshold survive;
// output is sorted by (final score desc, seqId asc).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSelCap = 256;
constexpr int kSelChunk = 64;     // queries per chunk: one per lane
constexpr int kSelLane = 4;       // diagonals up to which a query is selected inside its lane

struct SelectArgs {
    uint32_t nq; const DiagRec *diagRec; const uint16_t *diagScore; const uint32_t *qDiagBase; const uint32_t *qDiagCnt;
    const int *qbias; const int *qself; const uint32_t *qoff;
    uint8_t *diagFlag;              // scratch, one byte per diagonal
    uint64_t *selScratch;           // per-wave HBM key scratch (65536 keys) for oversized candidate lists
    const uint32_t *wantOff;        // start of each query's range in the staging buffer (= qDiagBase: a query selects at most as many hits as it has diagonals)
    uint32_t binMask; int maxHits; int minDiagScore;
    GpuHit *outHits; uint32_t *outQuery; uint32_t *outCount; uint32_t *outBase;   // compact output: each query owns [outBase, outBase+outCount)
    unsigned long long *counters;   // [4] += hits
    uint32_t chunkQ;                // queries per chunk (<= kSelChunk): 64 while most queries are finished inside a lane, 16 when most need the whole wave (dense index)
};

__device__ __forceinline__ void selectOne(const SelectArgs &a, const uint32_t qi, const uint32_t obase, uint32_t *hist, uint64_t *keys, uint64_t *scratch, const int lane) {
    const uint32_t D = uni(a.qDiagCnt[qi]);
    if (D == 0) { if (lane == 0) { a.outCount[qi] = 0; a.outBase[qi] = 0; } return; }
    const uint32_t base = uni(a.qDiagBase[qi]);
    const DiagRec *rec = a.diagRec + base;
    const uint16_t *exact = a.diagScore + base;
    uint8_t *flag = a.diagFlag + base;
    const int qL = (int) uni(a.qoff[qi + 1] - a.qoff[qi]);
    const int bias = (int) uni((uint32_t) a.qbias[qi]);
    const int maxC = (qL >= 32768) ? 255 : 255 - bias;          // clamp applied by scoreDiagonalAndUpdateHits
    const int truncAt = 255 - bias;                             // UCHAR_MAX - queryBias
    for (int t = lane; t < 256; t += 64) hist[t] = 0;
    for (uint32_t t = lane; t < D; t += 64) flag[t] = 0;
    waveSync();
    // per-target best (records arrive sorted by (seqId, stream position))
    for (uint32_t b0 = 0; b0 < D; b0 += 64) {
        const uint32_t t = b0 + lane;
        if (t < D) {
            const uint32_t id = rec[t].seqId;
            if (t == 0 || rec[t - 1].seqId != id) {
                int bestC = a.minDiagScore - 1; uint32_t best = 0xFFFFFFFFu;
                for (uint32_t u = t; u < D && rec[u].seqId == id; u++) {
                    const int c = min((int) exact[u], maxC);
                    if (c > bestC) { bestC = c; best = u; }
                }
                if (best != 0xFFFFFFFFu) { flag[best] = 1; atomicAdd(&hist[bestC], 1u); }
            }
        }
    }
    waveSync();
    // QueryMatcher::computeScoreThreshold (QueryMatcher.h:199-209)
    uint32_t thr = 0;
    if (lane == 0) {
        uint32_t found = 0; int t;
        for (t = 255; t > 0; t--) { found += hist[t]; if (found >= (uint32_t) a.maxHits) break; }
        thr = (uint32_t) t;
        if (thr < (uint32_t) a.minDiagScore) thr = (uint32_t) a.minDiagScore;
    }
    thr = uni(thr);
    const bool truncated = (int) thr >= truncAt;
    uint32_t nC = 0;
    for (int t = lane; t < 256; t += 64) if ((uint32_t) t >= thr) nC += hist[t];
    nC = waveSum(nC);
    if (nC == 0) { if (lane == 0) { a.outCount[qi] = 0; a.outBase[qi] = 0; } return; }
    const bool inLds = nC <= (uint32_t) kSelCap;
    uint64_t *K = inLds ? keys : scratch;
    uint32_t n = 1; while (n < nC) n <<= 1;
    // rescale factor of the saturated branch (QueryMatcher::rescoreHits, QueryMatcher.cpp:486-505)
    int maxSelf = (int) uni((uint32_t) a.qself[qi]) - truncAt; if (maxSelf < 1) maxSelf = 1; if (maxSelf > 65535) maxSelf = 65535;
    const float fltMax = (float) maxSelf;
    uint32_t wr = 0;
    for (uint32_t b0 = 0; b0 < D; b0 += 64) {
        const uint32_t t = b0 + lane;
        bool take = false; uint64_t key = 0;
        if (t < D && flag[t]) {
            int c = min((int) exact[t], maxC);
            if ((uint32_t) c >= thr) {
                take = true;
                if (truncated) {
                    uint32_t ns = (uint32_t) exact[t] - (uint32_t) truncAt;
                    const float sc = (float) (ns < 65535u ? ns : 65535u);
                    c = (int) (unsigned char) ((double) ((sc / fltMax) * (float) 255) + 0.5);
                }
                key = ((uint64_t) (255 - c) << 48) | ((uint64_t) (rec[t].seqId & a.binMask) << 32) | ((uint64_t) rec[t].idx << 16) | (uint64_t) t;
            }
        }
        const uint64_t m = __ballot(take);
        if (take) K[wr + __popcll(m & laneMaskLt(lane))] = key;
        wr += __popcll(m);
    }
    for (uint32_t t = nC + lane; t < n; t += 64) K[t] = ~0ull;
    waveSync();
    if (inLds) waveBitonicSort(keys, n, lane); else waveBitonicSort(K, n, lane);
    const uint32_t nOut = min(nC, (uint32_t) a.maxHits);
    // final score + final order (score desc, seqId asc)
    for (uint32_t t = lane; t < n; t += 64) {
        uint64_t key = ~0ull;
        if (t < nOut) {
            const uint64_t k2 = K[t];
            const uint32_t r = (uint32_t) (k2 & 0xFFFF);
            const int c = 255 - (int) (k2 >> 48);
            int score;
            if (truncated) score = truncAt + (int) ((uint32_t) c * (uint32_t) maxSelf / 255u);
            else score = (c >= truncAt) ? (int) exact[r] : c;
            if (score > 65535) score = 65535;
            key = ((uint64_t) (65535 - score) << 48) | ((uint64_t) rec[r].seqId << 16) | (uint64_t) r;
        }
        // all lanes must have read K[t] of this round before anybody overwrites: rounds are disjoint in t
        K[t] = key;
    }
    waveSync();
    uint32_t n2 = 1; while (n2 < nOut) n2 <<= 1;
    if (inLds) waveBitonicSort(keys, n2, lane); else waveBitonicSort(K, n2, lane);
    for (uint32_t t = lane; t < nOut; t += 64) {
        const uint64_t k3 = K[t];
        const uint32_t r = (uint32_t) (k3 & 0xFFFF);
        GpuHit h; h.seqId = rec[r].seqId; h.score = (uint16_t) (65535 - (uint32_t) (k3 >> 48)); h.diag = rec[r].diag;
        a.outHits[(size_t) obase + t] = h;
        a.outQuery[(size_t) obase + t] = qi;
    }
    if (lane == 0) { a.outCount[qi] = nOut; a.outBase[qi] = obase; }
    waveSync();
}

// Queries with 2..64 double-hit diagonals and fewer diagonals than --max-seqs: everything in registers, one record per
// lane.  With fewer candidates than maxHits the histogram threshold is minDiagScore and nothing is cut, so the only work
// is the per-target best and the final (score desc, seqId asc) order -- both by broadcasting one lane at a time.
__device__ __forceinline__ void selectSmall(const SelectArgs &a, const uint32_t qi, const uint32_t D, const uint32_t obase, const int lane) {
    const uint32_t base = uni(a.qDiagBase[qi]);
    const int qL = (int) uni(a.qoff[qi + 1] - a.qoff[qi]);
    const int bias = (int) uni((uint32_t) a.qbias[qi]);
    const int maxC = (qL >= 32768) ? 255 : 255 - bias;
    const int truncAt = 255 - bias;
    const bool in = (uint32_t) lane < D;
    DiagRec r{0xFFFFFFFFu, 0, 0}; int ex = 0;
    if (in) { r = a.diagRec[base + lane]; ex = (int) a.diagScore[base + lane]; }
    const int c = min(ex, maxC);
    // per-target best: records arrive sorted by (seqId, stream position); ties -> the earliest
    bool best = in && c >= a.minDiagScore;
    for (uint32_t u = 0; u < D; u++) {
        const uint32_t su = (uint32_t) __builtin_amdgcn_readlane((int) r.seqId, (int) u);
        const int cu = __builtin_amdgcn_readlane(c, (int) u);
        if (su == r.seqId && ((u < (uint32_t) lane && cu >= c) || (u > (uint32_t) lane && cu > c))) best = false;
    }
    int score = (c >= truncAt) ? ex : c;
    if (score > 65535) score = 65535;
    const uint64_t key = ((uint64_t) (65535 - score) << 32) | r.seqId;
    uint64_t m = __ballot(best);
    const uint32_t nOut = (uint32_t) __popcll(m);
    uint32_t rank = 0;
    while (m) {
        const int u = __builtin_ctzll(m);
        m &= m - 1;
        const uint64_t ku = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (key >> 32), u) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) key, u);
        rank += ku < key ? 1u : 0u;
    }
    if (best) {
        GpuHit h; h.seqId = r.seqId; h.score = (uint16_t) score; h.diag = r.diag;
        a.outHits[(size_t) obase + rank] = h;
        a.outQuery[(size_t) obase + rank] = qi;
    }
    if (lane == 0) { a.outCount[qi] = nOut; a.outBase[qi] = obase; }
}

__global__ __launch_bounds__(64 * kWavesPerBlock) void k_select(SelectArgs a) {
    __shared__ uint64_t s_keys[kWavesPerBlock][kSelCap];
    __shared__ uint32_t s_hist[kWavesPerBlock][256];
    const int lane = laneId();
    const int w = (int) (threadIdx.x >> 6);
    uint64_t *scratch = a.selScratch + (size_t) (blockIdx.x * kWavesPerBlock + w) * kSpillWordsPerWave;
    // kSelChunk queries per pop.  Output space is reserved once per chunk (sum of min(D, maxHits) -- an upper bound, the
    // gaps are closed by k_reorder_hits): single-diagonal queries (the common case for spacer peptides) are finished
    // lane-parallel, small ones in registers by the whole wave, the rest through LDS.
    // No atomics: chunks are dealt round-robin to the waves (the work per chunk is even) and every query's output range is
    // known beforehand (the place of its diagonal records); two contended counters cost more than the kernel's work.
    const uint32_t nWaves = gridDim.x * kWavesPerBlock;
    static_assert(kSelChunk == 64, "one query per lane");
    for (uint32_t chunk = blockIdx.x * kWavesPerBlock + w; (uint64_t) chunk * a.chunkQ < a.nq; chunk += nWaves) {
        const uint32_t qi = chunk * a.chunkQ + lane;
        const bool mine = (uint32_t) lane < a.chunkQ && qi < a.nq;
        // everything a query's selection starts from in ONE round trip, the records of the small ones in a second: this kernel usually runs beside another
        // context's k_match with a single wave per SIMD, where only the number of dependent round trips counts
        const uint32_t D = mine ? a.qDiagCnt[qi] : 0;
        const uint32_t myBase = mine ? a.wantOff[qi] : 0;       // (= the place of its diagonal records)
        const uint32_t o0 = mine ? a.qoff[qi] : 0, o1 = mine ? a.qoff[qi + 1] : 0;
        const int bias = mine ? a.qbias[qi] : 0;
        const int qL = (int) (o1 - o0);
        const int maxC = (qL >= 32768) ? 255 : 255 - bias, truncAt = 255 - bias;
        // up to kSelLane diagonals: the whole selection of the query in its lane (the statement of selectSmall; D == 1 needs none of its conditions)
        const bool fast = mine && D >= 1 && D < (uint32_t) a.maxHits && (D == 1 || (D <= (uint32_t) kSelLane && a.minDiagScore < 255 - bias));
        if (fast) {
            DiagRec r[kSelLane]; int ex[kSelLane];
#pragma unroll
            for (int u = 0; u < kSelLane; u++) { r[u] = DiagRec{0xFFFFFFFFu, 0, 0}; ex[u] = 0; }
#pragma unroll
            for (int u = 0; u < kSelLane; u++) if ((uint32_t) u < D) { r[u] = a.diagRec[myBase + u]; ex[u] = (int) a.diagScore[myBase + u]; }
            int c[kSelLane]; bool best[kSelLane]; uint64_t key[kSelLane]; int score[kSelLane];
#pragma unroll
            for (int u = 0; u < kSelLane; u++) c[u] = min(ex[u], maxC);
#pragma unroll
            for (int u = 0; u < kSelLane; u++) {
                best[u] = (uint32_t) u < D && c[u] >= a.minDiagScore;
#pragma unroll
                for (int v = 0; v < kSelLane; v++) if (v != u && (uint32_t) v < D && r[v].seqId == r[u].seqId && ((v < u && c[v] >= c[u]) || (v > u && c[v] > c[u]))) best[u] = false;
                score[u] = (c[u] >= truncAt) ? ex[u] : c[u];
                if (score[u] > 65535) score[u] = 65535;
                key[u] = ((uint64_t) (65535 - score[u]) << 32) | r[u].seqId;
            }
            uint32_t cnt = 0;
#pragma unroll
            for (int u = 0; u < kSelLane; u++) {
                if (!best[u]) continue;
                uint32_t rank = 0;
#pragma unroll
                for (int v = 0; v < kSelLane; v++) if (v != u && best[v] && key[v] < key[u]) rank++;
                GpuHit h; h.seqId = r[u].seqId; h.score = (uint16_t) score[u]; h.diag = r[u].diag;
                a.outHits[(size_t) myBase + rank] = h; a.outQuery[(size_t) myBase + rank] = qi;
                cnt++;
            }
            a.outBase[qi] = myBase; a.outCount[qi] = cnt;
        } else if (mine && D == 0) { a.outCount[qi] = 0; a.outBase[qi] = 0; }
        uint64_t slow = __ballot(D > 0 && !fast);
        while (slow) {
            const int bit = __builtin_ctzll(slow);
            slow &= slow - 1;
            const uint32_t dq = (uint32_t) __builtin_amdgcn_readlane((int) D, bit);
            const uint32_t ob = (uint32_t) __builtin_amdgcn_readlane((int) myBase, bit);
            const int qbias = __builtin_amdgcn_readlane(bias, bit);
            if (dq <= 64u && dq < (uint32_t) a.maxHits && a.minDiagScore < 255 - qbias) selectSmall(a, chunk * a.chunkQ + (uint32_t) bit, dq, ob, lane);
            else selectOne(a, chunk * a.chunkQ + (uint32_t) bit, ob, s_hist[w], s_keys[w], scratch, lane);
        }
    }
}

__global__ void k_select_want(const uint32_t *qDiagCnt, uint32_t nq, uint32_t maxHits, uint32_t *want) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) want[q] = min(qDiagCnt[q], maxHits);
}

// query-ordered output: exclusive scan of the per-query hit counts (one block), then move the hits from the staging
// array (allocation order of k_select) to their final place; everything downstream (download, align) is sequential
__global__ __launch_bounds__(1024) void k_scan_counts(const uint32_t *count, uint32_t n, uint32_t *offset /* n+1 */) {
    // one block (small inputs): tiles of 1024 counts read coalesced, DPP scan per wave, the 16 wave totals through LDS
    __shared__ uint32_t s_wave[16];
    const int lane = laneId(), wv = (int) (threadIdx.x >> 6);
    uint32_t carry = 0;
    for (uint32_t t0 = 0; t0 < n; t0 += 1024) {
        const uint32_t i = t0 + threadIdx.x;
        const uint32_t v = i < n ? count[i] : 0u;
        const uint32_t incl = waveInclusiveScan(v, lane);
        if (lane == 63) s_wave[wv] = incl;
        __syncthreads();
        uint32_t before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; w++) { const uint32_t x = s_wave[w]; all += x; if (w < wv) before += x; }
        if (i < n) offset[i] = carry + before + incl - v;
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) offset[n] = carry;
}

__global__ void k_reorder_hits(const GpuHit *staging, const uint32_t *base, const uint32_t *count, const uint32_t *offset, uint32_t nq, GpuHit *out, uint32_t *outQuery) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    const uint32_t b = base[q], o = offset[q], c = count[q];
    for (uint32_t k = 0; k < c; k++) { out[o + k] = staging[b + k]; outQuery[o + k] = q; }
}

// ---------------------------------------------------------------------------------------------------------------
// Target-sharded search: every GPU matches all queries against the index of ITS target range and
// exports, per (query, target), the best double-hit diagonal as a 16-byte candidate; the union of all shards' candidates
// goes through the same selection tail as the unsplit search (NOT Prefiltering::mergeTargetSplits, which applies a
// per-split quota -- prefiltering/Prefiltering.cpp:343-346 -- and therefore differs from the unsplit result).
// ---------------------------------------------------------------------------------------------------------------
struct CandArgs {
    const DiagRec *diagRec; const uint32_t *diagQuery; const uint16_t *diagScore; const uint32_t *diagOrd; const uint32_t *diagCursor; uint32_t diagCap;
    const uint32_t *qDiagBase; const uint32_t *qDiagCnt; const int *qbias; const uint32_t *qoff; int minDiagScore;
    Candidate *out; uint32_t *outCursor; uint32_t outCap;
};

// one lane per diagonal record: keep it iff it is its target's best (clamped score, ties -> earliest in the stream)
__global__ void k_candidates(CandArgs a) {
    uint32_t nDiag = *a.diagCursor; if (nDiag > a.diagCap) nDiag = a.diagCap;
    const int lane = laneId();
    for (uint32_t r0 = (blockIdx.x * blockDim.x + threadIdx.x) & ~63u; r0 < nDiag; r0 += gridDim.x * blockDim.x) {
        const uint32_t r = r0 + lane;
        bool keep = false; Candidate c{0, 0, 0, 0, 0};
        if (r < nDiag) {
            const uint32_t qi = a.diagQuery[r];
            if (qi != 0xFFFFFFFFu) {
                const uint32_t base = a.qDiagBase[qi], end = base + a.qDiagCnt[qi];
                const int qL = (int) (a.qoff[qi + 1] - a.qoff[qi]);
                const int maxC = (qL >= 32768) ? 255 : 255 - a.qbias[qi];
                const DiagRec me = a.diagRec[r];
                const int cMe = min((int) a.diagScore[r], maxC);
                keep = cMe >= a.minDiagScore;
                for (uint32_t u = r; keep && u > base && a.diagRec[u - 1].seqId == me.seqId; u--) if (min((int) a.diagScore[u - 1], maxC) >= cMe) keep = false;
                for (uint32_t u = r + 1; keep && u < end && a.diagRec[u].seqId == me.seqId; u++) if (min((int) a.diagScore[u], maxC) > cMe) keep = false;
                c.query = qi; c.seqId = me.seqId; c.ord = a.diagOrd[r]; c.diag = me.diag; c.exact = a.diagScore[r];
            }
        }
        const uint64_t m = __ballot(keep);
        if (m) {
            const uint32_t ob = uni(atomicAdd(a.outCursor, lane == 0 ? (uint32_t) __popcll(m) : 0u));
            if (keep) { const uint32_t at = ob + (uint32_t) __popcll(m & laneMaskLt(lane)); if (at < a.outCap) a.out[at] = c; }
        }
    }
}

// union of candidates -> grouped by query (counting sort; the order inside a query is irrelevant, the selection ranks
// by a total order)
__global__ void k_cand_count(const Candidate *c, uint32_t n, uint32_t qFrom, uint32_t qTo, uint32_t *cnt) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t q = c[i].query; if (q >= qFrom && q < qTo) atomicAdd(&cnt[q], 1u); }
}
__global__ void k_cand_scatter(const Candidate *c, uint32_t n, uint32_t qFrom, uint32_t qTo, const uint32_t *offset, uint32_t *fill, Candidate *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const Candidate v = c[i]; if (v.query >= qFrom && v.query < qTo) out[offset[v.query] + atomicAdd(&fill[v.query], 1u)] = v; }
}

struct MergeArgs {
    uint32_t nq; const Candidate *cand; const uint32_t *candOffset;     // candidates grouped by query, candOffset[nq + 1]
    const int *qbias; const int *qself; const uint32_t *qoff;
    uint64_t *scratch;              // per wave: 65536 u64 keys + 65536 u32 indices
    uint32_t *queue;
    uint32_t binMask; int maxHits; int minDiagScore;
    GpuHit *outHits; uint32_t *outQuery; uint32_t *outCount; uint32_t *outBase; uint32_t *outCursor;
    unsigned long long *counters;   // [2] += queries with more candidates than the scratch holds
};
constexpr size_t kMergeScratchWords = 65536 + 65536 / 2;
constexpr int kMergeIdxLds = 1024;

// bitonic sort of candidate indices by the selection order (rank key desc, bin asc, k-mer ordinal asc, seqId asc)
template <typename Less>
__device__ __forceinline__ void waveBitonicSortIdx(uint32_t *idx, uint32_t n, int lane, Less less) {
    for (uint32_t k = 2; k <= n; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = lane; t < (n >> 1); t += 64) {
                const uint32_t lo = t & (j - 1);
                const uint32_t i = ((t - lo) << 1) | lo;
                const uint32_t p = i | j;
                const uint32_t x = idx[i], y = idx[p];
                const bool up = ((i & k) == 0);
                if (less(y, x) == up) { idx[i] = y; idx[p] = x; }
            }
            waveSync();
        }
    }
}

// the tail of QueryMatcher::matchQuery on the union of all shards' candidates; same statement as selectOne() with the
// global stream order (k-mer ordinal, seqId) in place of the shard-local stream index
__global__ __launch_bounds__(64 * kWavesPerBlock) void k_select_merged(MergeArgs a) {
    __shared__ uint64_t s_keys[kWavesPerBlock][kSelCap];
    __shared__ uint32_t s_idx[kWavesPerBlock][kMergeIdxLds];
    __shared__ uint32_t s_hist[kWavesPerBlock][256];
    const int lane = laneId();
    const int w = (int) (threadIdx.x >> 6);
    uint32_t *hist = s_hist[w];
    uint64_t *scratchK = a.scratch + (size_t) (blockIdx.x * kWavesPerBlock + w) * kMergeScratchWords;
    uint32_t *scratchI = (uint32_t *) (scratchK + 65536);
    // Like k_select: no atomics -- chunks of kSelChunk queries are dealt round-robin to the waves, and a query's hits are staged where
    // its candidates sit in the grouped list (it selects at most as many; k_reorder_hits closes the gaps).  A work-queue pop and an
    // output reservation per chunk were 75 000 returning atomics on two words: 0.9 of this kernel's 0.98 ms at 600 000 queries.
    // Queries with no candidate or a single one -- the rule for spacer peptides -- are finished a query per lane; queries whose union
    // holds fewer candidates than --max-seqs (and at most 64) need no histogram and no cut: one candidate per lane, final order only.
    const uint32_t nWaves = gridDim.x * kWavesPerBlock;
    for (uint32_t chunk = blockIdx.x * kWavesPerBlock + w; (uint64_t) chunk * kSelChunk < a.nq; chunk += nWaves) {
        const uint32_t ql = chunk * kSelChunk + (uint32_t) lane;
        const bool mine = lane < kSelChunk && ql < a.nq;
        const uint32_t myBase = mine ? a.candOffset[ql] : 0u;
        const uint32_t myD = mine ? a.candOffset[ql + 1] - myBase : 0u;
        bool done = !mine || myD == 0u;
        if (mine && myD == 0u) { a.outCount[ql] = 0; a.outBase[ql] = 0; }
        else if (mine && myD == 1u && a.maxHits > 1) {
            const int bias0 = a.qbias[ql];
            if (a.minDiagScore < 255 - bias0) {
                const int qL0 = (int) (a.qoff[ql + 1] - a.qoff[ql]);
                const int maxC0 = (qL0 >= 32768) ? 255 : 255 - bias0;
                const Candidate c = a.cand[myBase];
                const int v = min((int) c.exact, maxC0);
                int score = (v >= 255 - bias0) ? (int) c.exact : v;
                if (score > 65535) score = 65535;
                GpuHit h; h.seqId = c.seqId; h.score = (uint16_t) score; h.diag = c.diag;
                a.outHits[(size_t) myBase] = h;
                a.outQuery[(size_t) myBase] = ql;
                a.outCount[ql] = 1; a.outBase[ql] = myBase;
                done = true;
            }
        }
        uint64_t slow = __ballot(!done);
      while (slow) {
        const int slot = __builtin_ctzll(slow);
        slow &= slow - 1;
        const uint32_t qi = chunk * kSelChunk + (uint32_t) slot;
        uint32_t D = (uint32_t) __builtin_amdgcn_readlane((int) myD, slot);
        const uint32_t obase = (uint32_t) __builtin_amdgcn_readlane((int) myBase, slot);
        const uint32_t c0 = uni(a.candOffset[qi]);
        if (D > 65535u) { if (lane == 0) atomicAdd(&a.counters[2], 1ull); D = 0; }
        if (D == 0) { if (lane == 0) { a.outCount[qi] = 0; a.outBase[qi] = 0; } continue; }
        const Candidate *cand = a.cand + c0;
        if (D <= 64u && D < (uint32_t) a.maxHits) {
            const int bias0 = (int) uni((uint32_t) a.qbias[qi]);
            const int qL0 = (int) uni(a.qoff[qi + 1] - a.qoff[qi]);
            if (a.minDiagScore < 255 - bias0) {
                const int maxC0 = (qL0 >= 32768) ? 255 : 255 - bias0;
                const bool in = (uint32_t) lane < D;
                Candidate c{0, 0xFFFFFFFFu, 0, 0, 0};
                if (in) c = cand[lane];
                const int v = min((int) c.exact, maxC0);
                int score = (v >= 255 - bias0) ? (int) c.exact : v;
                if (score > 65535) score = 65535;
                const uint64_t key = ((uint64_t) (65535 - score) << 32) | c.seqId;
                uint32_t rank = 0;
                for (uint32_t u = 0; u < D; u++) {
                    const uint64_t ku = ((uint64_t) (uint32_t) __builtin_amdgcn_readlane((int) (key >> 32), (int) u) << 32) | (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) key, (int) u);
                    rank += ku < key ? 1u : 0u;
                }
                if (in) {
                    GpuHit h; h.seqId = c.seqId; h.score = (uint16_t) score; h.diag = c.diag;
                    a.outHits[(size_t) obase + rank] = h;
                    a.outQuery[(size_t) obase + rank] = qi;
                }
                if (lane == 0) { a.outCount[qi] = D; a.outBase[qi] = obase; }
                continue;
            }
        }
        const int qL = (int) uni(a.qoff[qi + 1] - a.qoff[qi]);
        const int bias = (int) uni((uint32_t) a.qbias[qi]);
        const int maxC = (qL >= 32768) ? 255 : 255 - bias;
        const int truncAt = 255 - bias;
        for (int t = lane; t < 256; t += 64) hist[t] = 0;
        waveSync();
        for (uint32_t t = lane; t < D; t += 64) atomicAdd(&hist[min((int) cand[t].exact, maxC)], 1u);
        waveSync();
        uint32_t thr = 0;
        if (lane == 0) {
            uint32_t found = 0; int t;
            for (t = 255; t > 0; t--) { found += hist[t]; if (found >= (uint32_t) a.maxHits) break; }
            thr = (uint32_t) t;
            if (thr < (uint32_t) a.minDiagScore) thr = (uint32_t) a.minDiagScore;
        }
        thr = uni(thr);
        const bool truncated = (int) thr >= truncAt;
        uint32_t nC = 0;
        for (int t = lane; t < 256; t += 64) if ((uint32_t) t >= thr) nC += hist[t];
        nC = waveSum(nC);
        if (nC == 0) { if (lane == 0) { a.outCount[qi] = 0; a.outBase[qi] = 0; } continue; }
        uint32_t n = 1; while (n < nC) n <<= 1;
        const bool idxLds = n <= (uint32_t) kMergeIdxLds;
        uint32_t *I = idxLds ? s_idx[w] : scratchI;
        int maxSelf = (int) uni((uint32_t) a.qself[qi]) - truncAt; if (maxSelf < 1) maxSelf = 1; if (maxSelf > 65535) maxSelf = 65535;
        const float fltMax = (float) maxSelf;
        // rank value of a candidate: its clamped score, or the rescaled one in the saturated branch (QueryMatcher.cpp:486-505)
        auto rankC = [&](const Candidate &c) -> int {
            int v = min((int) c.exact, maxC);
            if (truncated) {
                const uint32_t ns = (uint32_t) c.exact - (uint32_t) truncAt;
                const float sc = (float) (ns < 65535u ? ns : 65535u);
                v = (int) (unsigned char) ((double) ((sc / fltMax) * (float) 255) + 0.5);
            }
            return v;
        };
        uint32_t wr = 0;
        for (uint32_t b0 = 0; b0 < D; b0 += 64) {
            const uint32_t t = b0 + lane;
            const bool take = t < D && (uint32_t) min((int) cand[t].exact, maxC) >= thr;
            const uint64_t m = __ballot(take);
            if (take) I[wr + (uint32_t) __popcll(m & laneMaskLt(lane))] = t;
            wr += (uint32_t) __popcll(m);
        }
        for (uint32_t t = nC + lane; t < n; t += 64) I[t] = 0xFFFFFFFFu;
        waveSync();
        auto less = [&](uint32_t x, uint32_t y) -> bool {       // strict order; the padding index sorts last
            if (x == 0xFFFFFFFFu || y == 0xFFFFFFFFu) return x != 0xFFFFFFFFu && y == 0xFFFFFFFFu;
            const Candidate cx = cand[x], cy = cand[y];
            const int rx = rankC(cx), ry = rankC(cy);
            if (rx != ry) return rx > ry;
            const uint32_t bx = cx.seqId & a.binMask, by = cy.seqId & a.binMask;
            if (bx != by) return bx < by;
            if (cx.ord != cy.ord) return cx.ord < cy.ord;
            return cx.seqId < cy.seqId;
        };
        if (idxLds) waveBitonicSortIdx(s_idx[w], n, lane, less); else waveBitonicSortIdx(scratchI, n, lane, less);
        const uint32_t nOut = min(nC, (uint32_t) a.maxHits);
        uint32_t n2 = 1; while (n2 < nOut) n2 <<= 1;
        const bool keyLds = n2 <= (uint32_t) kSelCap;
        uint64_t *K = keyLds ? s_keys[w] : scratchK;
        for (uint32_t t = lane; t < n2; t += 64) {
            uint64_t key = ~0ull;
            if (t < nOut) {
                const Candidate c = cand[I[t]];
                const int v = rankC(c);
                int score;
                if (truncated) score = truncAt + (int) ((uint32_t) v * (uint32_t) maxSelf / 255u);
                else score = (v >= truncAt) ? (int) c.exact : v;
                if (score > 65535) score = 65535;
                key = ((uint64_t) (65535 - score) << 48) | ((uint64_t) c.seqId << 16) | (uint64_t) t;
            }
            K[t] = key;
        }
        waveSync();
        if (keyLds) waveBitonicSort(s_keys[w], n2, lane); else waveBitonicSort(scratchK, n2, lane);
        for (uint32_t t = lane; t < nOut; t += 64) {
            const uint64_t k3 = K[t];
            const Candidate c = cand[I[(uint32_t) (k3 & 0xFFFF)]];
            GpuHit h; h.seqId = c.seqId; h.score = (uint16_t) (65535 - (uint32_t) (k3 >> 48)); h.diag = c.diag;
            a.outHits[(size_t) obase + t] = h;
            a.outQuery[(size_t) obase + t] = qi;
        }
        if (lane == 0) { a.outCount[qi] = nOut; a.outBase[qi] = obase; }
        waveSync();
      }
    }
}

#include "sp_gpu_sw.inc"
#include "sp_gpu_index.inc"

// ---------------------------------------------------------------------------------------------------------------
// device context
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct DevBuf {
    T *p = nullptr; size_t n = 0;
    bool ensure(size_t count, std::string &err) {
        if (count <= n && p) return true;
        if (p) (void) hipFree(p);
        p = nullptr; n = 0;
        size_t want = count + count / 8 + 64;
        hipError_t e = hipMalloc((void **) &p, want * sizeof(T));
        if (e != hipSuccess) { err = std::string("hipMalloc: ") + hipGetErrorString(e); return false; }
        n = want;
        return true;
    }
    void release() { if (p) (void) hipFree(p); p = nullptr; n = 0; }
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
        int matchSmall = -1, matchPerCU = 0, matchHeavy = -1, matchStatic = 4, matchDChunk = 4, weighExact = -1, matchGroup = 1, matchOrder = 1;
        bool debug = false, prepSplit = false, swRows32 = false, swRows16 = false, swOldForward = false;
        void read() {
            auto num = [](const char *k, int d) { const char *e = getenv(k); return e ? atoi(e) : d; };
            matchSmall = num("SP_MATCH_SMALL", -1); matchGroup = num("SP_MATCH_GROUP", 1); matchOrder = num("SP_MATCH_ORDER", 1); { const int we = num("SP_MATCH_WEIGH_EXACT", -1); weighExact = we < 0 ? -1 : (we ? 1 : 0); } matchPerCU = num("SP_MATCH_PERCU", 0); matchHeavy = num("SP_MATCH_HEAVY", -1);
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
        DevBuf<uint64_t> buckets; bool bucketsValid = false; size_t indexSlots = 0;      // large-footprint layout of the slot table, derived from slots / ientries on first use
        DevBuf<uint64_t> pbits; DevBuf<uint16_t> prank; DevBuf<uint32_t> prow; DevBuf<uint64_t> slots; DevBuf<uint64_t> ientries; DevBuf<uint8_t> tmasked; DevBuf<uint8_t> tplain; DevBuf<char> tascii; DevBuf<uint64_t> toff; DevBuf<uint32_t> tkeys;
        size_t nTargets = 0; size_t indexEntries = 0;
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
    DevBuf<unsigned char> ixTemp; DevBuf<float> ttProbs; DevBuf<double> ttScale, ttLr; DevBuf<unsigned long long> ixCounters;
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
    int timing = 1;                              // 0: no HIP events in the pipeline, 1: around k_match only (each record costs ~6 us of idle GPU), 2: every stage
    // the hit list starts travelling to the host on its own stream as soon as a prefilter pass is complete, so that the copy
    // overlaps whatever the caller queues next (typically the align kernels)
    hipStream_t copyStream = nullptr; hipEvent_t hitsOnHost = nullptr, alnOnHost = nullptr; bool alnInFlight = false;
    PinVec<GpuHit> hitStage; PinVec<uint32_t> hitOffStage; bool hitsInFlight = false;
    GpuHit *sinkHits = nullptr; uint32_t *sinkOff = nullptr; size_t sinkCap = 0, sinkN = 0; bool sinkUsed = false;      // caller-owned delivery buffers (setHitSink)
    // exclusive scan of n counts into n + 1 offsets (offset[n] = total); count must have n + 1 readable entries.
    // Small inputs: one block (k_scan_counts); larger ones: rocPRIM's decoupled look-back scan.
    bool scanCounts(const uint32_t *count, uint32_t n, uint32_t *offset, std::string &err) {
        if (n <= 4096) { hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, stream, count, n, offset); return true; }
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
    uint32_t nCand = 0;
    SwState sw;
    SwDev swd;
    // device-driven steps (stepBegin / stepEnd): the whole pass -- prefilter, selection, align, record assembly, delivery -- is queued without a
    // host round trip; counts stay on the device, results and counters travel on the copy stream.  Up to two steps in flight.
    struct StepSlot { hipEvent_t selected = nullptr, assembled = nullptr, hitsOnHost = nullptr, alnOnHost = nullptr; size_t cap = 0; StepOut out; };
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

void Gpu::sync() { (void) hipStreamSynchronize(d_->stream); }

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
        if (!d.tab->ext3.ensure(N2, err) || !d.tab->ext3cnt.ensure((size_t) 8000 * kCntW, err) || !k1.ensure(N2, err) || !k2.ensure(N2, err) || !v1.ensure(N2, err) || !v2.ensure(N2, err)) return false;
        ea.hi = hi; ea.keys = k1.p; ea.vals = v1.p;
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
        for (DevBuf<uint32_t> *b : {&k1, &k2, &v1, &v2}) { (void) hipFree(b->p); b->p = nullptr; }
        (void) hipFree(tmp.p); tmp.p = nullptr;
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
bool Gpu::matchAndScore(int kmerThr, bool withOrd, std::string &err) {
    Impl &d = *d_;
    overflowList_.clear();
    if (!d.tab->ext3.p || !d.tgt->pbits.p) { err = "tables/targets not resident"; return false; }
    if (d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }     // the previous pass's hit list is still being copied
    // persistent match grid: as many blocks per CU as the LDS footprint admits.  512 hits per query live in LDS; more go to
    // the per-wave HBM scratch and come back through the pre-filter (faster than a larger LDS buffer even for dense indices:
    // 45 vs 51 ms at 2 x 10^8 residues, because three blocks per CU stay resident instead of one).
    // footprint by index density (SP_MATCH_SMALL=0/1 overrides): sparse -> small, dense -> large; the sharded search always small
    // (its ordinal arrays push the large footprint to two blocks per CU)
    const bool small = d.knob.matchSmall >= 0 ? d.knob.matchSmall != 0 : (withOrd || !d.tgt->denseIndex);
    // short queries against a sparse index, unsharded: the grouped kernel (sp_gpu_match_grp.inc) -- same records, four queries per pass of a wave
    const bool grouped = small && !withOrd && d.knob.matchGroup != 0 && d.maxQLen <= 64 && d.tgt->nTargets < ((size_t) 1 << 30);
    const void *fn = grouped ? (const void *) k_match_grp :
                     small ? (withOrd ? (const void *) k_match<256, true, true> : (const void *) k_match<256, false, true>)
                           : (withOrd ? (const void *) k_match<512, true, false> : (const void *) k_match<512, false, false>);
    int perCU = 2;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCU, fn, 64 * kWavesPerBlock, 0) != hipSuccess || perCU < 1) perCU = 2;
    if (d.matchPerCU > 0) perCU = std::min(perCU, d.matchPerCU);
    // Several search contexts on one GPU: the grouped kernel at TWO blocks per CU leaves half of every CU -- LDS, wave slots, registers -- to the other contexts' Smith-Waterman,
    // traceback and selection kernels and loses 13 % alone (0.533 against 0.468 ms; k_match loses 22 %): four host-driven contexts reach 23.4 M spacer-queries/s with two blocks,
    // 21.2 M with three, 20.6 M with four (profiles/r04_blocks_sweep.txt).  A context that searches alone takes all that fit.
    else if (d.matchPerCU == 0 && grouped && g_searchers[d.device & 63].load(std::memory_order_relaxed) >= 2) perCU = std::min(perCU, 2);
    if (d.knob.matchPerCU > 0) perCU = d.knob.matchPerCU;
    const bool dbg = d.knob.debug;
    if (dbg) fprintf(stderr, "[sp] k_match footprint=%s ord=%d blocks/CU=%d\n", grouped ? "grouped" : (small ? "small" : "large"), (int) withOrd, perCU);
    const uint32_t blocks = (uint32_t) std::min<size_t>((nq_ + kWavesPerBlock - 1) / kWavesPerBlock, (size_t) d.numCU * perCU);
    d.matchBlocks = blocks;
    if (!d.spill.ensure((size_t) blocks * kWavesPerBlock * kSpillWordsPerWave, err)) return false;
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
    const uint32_t heavyThr = d.knob.matchHeavy >= 0 ? (uint32_t) d.knob.matchHeavy : (nq_ <= (size_t) blocks * kWavesPerBlock * 32 ? 4000u : 0u);
    // exact weights order the heavy queries better (k_match alone 0.43 instead of 0.47 ms on the bench workload) and cost k_prep8 two table rows per window -- 20 us alone, a
    // multiple beside other contexts' matching kernels, where the estimate wins: exact while this context is the only one searching on its device
    // (... while the heavy queries were LISTED by a threshold.  With the batch sorted by weight -- `ordered` below, the grouped kernel -- the estimate orders it well enough: 0.353 ms
    // + 0.044 prep against 0.365 + 0.064 with exact counts; exact there only on request)
    const bool orderedPossible = grouped && heavyThr != 0 && d.knob.matchOrder != 0 && d.maxQLen <= 64 && !d.knob.prepSplit;
    const bool weighExact = d.knob.weighExact > 0 || (d.knob.weighExact < 0 && !orderedPossible && g_searchers[d.device & 63].load(std::memory_order_relaxed) <= 1);
    // the grouped kernel takes its queries in the order of decreasing weight (k_order_*) instead of "heavy ones first, then shares": needs the weights (k_prep8)
    const bool ordered = orderedPossible;
    const uint32_t orderTiles = (uint32_t) ((nq_ + kOrderTile - 1) / kOrderTile);
    if (ordered && (!d.qweight.ensure(nq_ + 1, err) || !d.qorder.ensure(nq_ + 1, err) || !d.orderHist.ensure((size_t) orderTiles * kOrderClasses, err) || !d.orderHeader.ensure(8, err) ||
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
            hipLaunchKernelGGL(k_order_count, dim3(orderTiles), dim3(256), 0, d.stream, d.qweight.p, (uint32_t) nq_, d.orderHist.p);
            hipLaunchKernelGGL(k_order_scatter, dim3(orderTiles), dim3(256), 0, d.stream, d.qweight.p, (uint32_t) nq_, d.orderHist.p, orderTiles, d.qorder.p, reinterpret_cast<OrderHeader *>(d.orderHeader.p),
                               d.orderQueues.p, blocks * (uint32_t) kWavesPerBlock / (uint32_t) kOrderQueues);
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
    if (!small) {      // (the bucket table: once per target set, on the first pass with the large footprint -- of whichever context sharing the set gets here first)
      std::lock_guard<std::mutex> shared(d.tgt->mu);
      if (!d.tgt->bucketsValid) {
        if (!d.tgt->buckets.ensure(4 * (size_t) kSlots, err)) return false;
        hipLaunchKernelGGL(k_idx_buckets, dim3((unsigned) ((kSlots + 255) / 256)), dim3(256), 0, d.stream, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->ientries.p, d.tgt->buckets.p);
        HIP_OK(hipStreamSynchronize(d.stream));      // (complete before another context's stream reads it)
        d.tgt->bucketsValid = true;
      }
    }
    MatchArgs ma{d.qres.p, d.qoff.p, (uint32_t) nq_, d.qthr.p, d.tab->ext3.p, d.tab->ext3cnt.p, d.tab->cntMin, d.tgt->pbits.p, d.tgt->prank.p, d.tgt->prow.p, d.tgt->slots.p, d.tgt->ientries.p, small ? nullptr : d.tgt->buckets.p, d.spill.p, d.queue.p,
                 d.diagRec.p, d.diagQuery.p, (uint32_t) diagCap, d.queue.p + 1, d.qDiagBase.p, d.qDiagCnt.p, d.counters.p,
                 withOrd ? d.diagOrd.p : nullptr, withOrd ? d.spillOrd.p : nullptr,
                 (uint32_t) d.knob.matchStatic, (uint32_t) d.knob.matchDChunk,
                 d.heavyList.p, d.heavyFlag.p, d.heavyCnt.p, d.wavePartial.p,
                 ordered ? d.qorder.p : nullptr, ordered ? reinterpret_cast<const OrderHeader *>(d.orderHeader.p) : nullptr, ordered ? d.orderQueues.p : nullptr};
    if (grouped) hipLaunchKernelGGL(k_match_grp, dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    else if (small) {
        if (withOrd) hipLaunchKernelGGL((k_match<256, true, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
        else hipLaunchKernelGGL((k_match<256, false, true>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
    } else if (withOrd) hipLaunchKernelGGL((k_match<512, true, false>), dim3(blocks), dim3(64 * kWavesPerBlock), 0, d.stream, ma);
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
    const size_t stageCap = std::max<size_t>(nq_ * (size_t) maxHits, d.diagCap);
    if (!d.outHits.ensure(nq_ * (size_t) maxHits, err) || !d.outStage.ensure(stageCap, err) || !d.outQuery.ensure(stageCap, err)) return false;
    SelectArgs se{(uint32_t) nq_, d.diagRec.p, d.diagScore.p, d.qDiagBase.p, d.qDiagCnt.p, d.qbias.p, d.qself.p, d.qoff.p, d.diagFlag.p, d.spill.p, d.qDiagBase.p,
                  binSize - 1, maxHits, minDiagScore, d.outStage.p, d.outQuery.p, d.outCount.p, d.outBase.p, d.counters.p, d.tgt->denseIndex ? 16u : (uint32_t) kSelChunk};
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
    const size_t cap = std::max<size_t>(1, std::min<size_t>({d.stepCap, (size_t) out.hitCap, (size_t) out.recordCap, (size_t) 0x7FFFFFF0u, nq_ * (size_t) std::max(1, maxHits)}));
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
    d.evSet = slot;
    if (!matchAndScore(kmerThr, false, err)) return false;
    if (!queueSelect(binSize, maxHits, minDiagScore, behind ? prev.hitsOnHost : nullptr, err)) return false;
    {   StepCountArgs ca{d.outOffset.p + nq_, (uint32_t) cap, d.stepWords.p};
        hipLaunchKernelGGL(k_step_counts, dim3(1), dim3(64), 0, d.stream, ca); }
    HIP_OK(hipEventRecord(ss.selected, d.stream));
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
    for (hipEvent_t e : {ss.hitsOnHost, ss.alnOnHost}) {      // short waits on the critical path: poll
        unsigned tries = 0;
        for (;;) { const hipError_t q = hipEventQuery(e); if (q == hipSuccess) break; if (q != hipErrorNotReady) { err = std::string("stepEnd: ") + hipGetErrorString(q); d.stepsEnded++; return false; } pollPause(tries); }
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
    d.nCand = 0;
    if (nq_ == 0) return true;
    if (d.stepsBegun != d.stepsEnded && !d.drained) { err = "prefilterCandidates: a device-driven step is in flight (collect it with stepEnd first)"; return false; }
    if (!matchAndScore(kmerThr, true, err)) return false;
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

const Candidate *Gpu::candidatesDevice() const { return d_->cand.p; }
size_t Gpu::numCandidates() const { return d_->nCand; }

// Target-sharded search: every rank has matched ALL queries against ITS target shard (prefilterCandidates); the shards' candidate lists are exchanged -- counts over the
// bootstrap sockets, the lists themselves padded to the longest one by RCCL all-gather, HBM to HBM over xGMI -- and land in candUnion, where mergeCandidates() reads them
// (rows of 0xFF are padding).  The reference merges the per-split result FILES of its MPI ranks (prefiltering/Prefiltering.cpp:258-362).
bool Gpu::allgatherCandidates(Comm &comm, size_t &unionRows, std::string &err) {
    Impl &d = *d_;
    const int world = comm.world();
    uint64_t mine = d.nCand; std::vector<uint64_t> all((size_t) world, 0);
    if (!comm.allgather(&mine, sizeof(mine), all.data(), false, err)) return false;
    uint64_t longest = 1; for (uint64_t x : all) longest = std::max(longest, x);
    if (longest * (uint64_t) world >= (1ull << 32)) { err = "allgatherCandidates: more than 2^32 candidate rows"; return false; }
    if (!d.candPad.ensure((size_t) longest, err) || !d.candUnion.ensure((size_t) longest * world + 1, err)) return false;
    HIP_OK(hipMemsetAsync(d.candPad.p, 0xFF, (size_t) longest * sizeof(Candidate), d.stream));
    if (d.nCand) HIP_OK(hipMemcpyAsync(d.candPad.p, d.cand.p, (size_t) d.nCand * sizeof(Candidate), hipMemcpyDeviceToDevice, d.stream));
    HIP_OK(hipStreamSynchronize(d.stream));
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
    if (queryOffset + nq_ > nqTotal) { err = "allgatherHits: this rank's slice ends behind the last query"; return false; }
    // srcHits / srcOffsets: a delivered step's hit list in host memory (the context may already be searching the next batch); nullptr: the last pass's, still on the device
    const uint64_t myHits = srcOffsets ? (uint64_t) srcOffsets[nq_] : (uint64_t) counters_.hits;
    uint64_t mine[3] = {(uint64_t) queryOffset, (uint64_t) nq_, myHits}; std::vector<uint64_t> all((size_t) world * 3, 0);
    if (!comm.allgather(mine, sizeof(mine), all.data(), false, err)) return false;
    uint64_t maxQ = 1, maxH = 1, totalH = 0, totalQ = 0;
    for (int r = 0; r < world; r++) { maxQ = std::max(maxQ, all[(size_t) r * 3 + 1]); maxH = std::max(maxH, all[(size_t) r * 3 + 2]); totalH += all[(size_t) r * 3 + 2]; totalQ += all[(size_t) r * 3 + 1]; }
    if (totalQ != nqTotal) { err = "allgatherHits: the ranks' slices hold " + std::to_string(totalQ) + " queries, not " + std::to_string(nqTotal); return false; }
    if (totalH > hitCap) { err = "allgatherHits: " + std::to_string(totalH) + " hits, room for " + std::to_string(hitCap); return false; }
    if (!srcOffsets && d.hitsInFlight) { (void) hipEventSynchronize(d.hitsOnHost); d.hitsInFlight = false; }
    if (!d.xOff.ensure((size_t) maxQ + 1, err) || !d.xOffAll.ensure((size_t) (maxQ + 1) * world, err) || !d.xHit.ensure((size_t) maxH, err) || !d.xHitAll.ensure((size_t) maxH * world, err)) return false;
    hipStream_t xs = srcOffsets ? d.copyStream : d.stream;      // (a host source has nothing to wait for on the search stream)
    HIP_OK(hipMemsetAsync(d.xOff.p, 0, (size_t) (maxQ + 1) * sizeof(uint32_t), xs));
    if (nq_) HIP_OK(hipMemcpyAsync(d.xOff.p, srcOffsets ? (const void *) srcOffsets : (const void *) d.outOffset.p, (nq_ + 1) * sizeof(uint32_t), hipMemcpyDefault, xs));
    if (myHits) HIP_OK(hipMemcpyAsync(d.xHit.p, srcOffsets ? (const void *) srcHits : (const void *) d.outHits.p, (size_t) myHits * sizeof(GpuHit), hipMemcpyDefault, xs));
    HIP_OK(hipStreamSynchronize(xs));
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
    if (n) hipLaunchKernelGGL(k_cand_count, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, u, (uint32_t) n, (uint32_t) qFrom, (uint32_t) qTo, d.candCnt.p);
    if (!d.scanCounts(d.candCnt.p, (uint32_t) nq_, d.candOffset.p, err)) return false;
    if (n) hipLaunchKernelGGL(k_cand_scatter, dim3((unsigned) ((n + 255) / 256)), dim3(256), 0, d.stream, u, (uint32_t) n, (uint32_t) qFrom, (uint32_t) qTo, d.candOffset.p, d.candFill.p, d.candByQuery.p);
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
    d.tgt->nTargets = n; d.tgt->indexEntries = 0; d.tgt->indexSlots = 0; d.tgt->bucketsValid = false;
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
        if (!d.ttProbs.ensure((size_t) R + 1, err) || !d.ttScale.ensure((size_t) (R / 16 + n + 2), err)) return false;
        TantanArgs ta{};
        ta.masked = d.tgt->tmasked.p; ta.toff = d.tgt->toff.p; ta.from = (uint32_t) from; ta.to = (uint32_t) to; ta.lr = d.ttLr.p;
        for (int i = 0; i < 50; i++) ta.b2f[i] = tm.b2f[i];
        ta.b2b = tm.b2b; ta.f2b = tm.f2b; ta.f2f0 = tm.f2f0; ta.probs = d.ttProbs.p; ta.scale = d.ttScale.p; ta.maskedCount = d.ixCounters.p;
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
            if (!d.tgt->ientries.ensure(E + 1, err) || !d.ixKmer.ensure(E + 1, err) || !d.ixHead.ensure(E + 1, err) || !d.ixSlotOf.ensure(E + 1, err)) return false;
            hipLaunchKernelGGL(k_idx_compact, dim3(gv), dim3(256), 0, d.stream, d.ixKeys2.p, d.ixVals2.p, d.ixKeep.p, d.ixPos.p, nValid, d.ixKmer.p, d.tgt->ientries.p);
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
    if (d.alnInFlight) { HIP_OK(hipEventSynchronize(d.alnOnHost)); d.alnInFlight = false; }
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
    HIP_OK(hipEventRecord(d.hitsOnHost, d.copyStream));
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
