// Host-side pieces of the MI355X-native spacer->phage-ORF search path: substitution matrices, target masking,
// k-mer index construction, E-value statistics.  Everything here is set-up work that runs once per target set
// (or once per process); the per-query hot path lives in sp_gpu.hip.
//
// Reference files restated (paths under /root/reference/lib/mmseqs/src/): see each function.
#pragma once
#include <cstdint>
#include <cstddef>
#include <functional>
#include <string>
#include <vector>

namespace sp {

constexpr int kAlph = 21;          // ACDEFGHIKLMNPQRSTVWYX
constexpr int kKmer = 6;
constexpr int kSpan = 8;           // spaced pattern 11011101
constexpr uint32_t kTableSize = 64000000u;   // 20^6
constexpr size_t kShardMapBytes = kTableSize / 8;      // a target shard's occupancy map: one bit per k-mer slot (the probe-list cut, sp_gpu.hip)
constexpr int kX = 20;
extern const int kPatternPos[kKmer];

struct SubMat {
    int alph = kAlph;
    short sub[kAlph][kAlph];
    double pBack[kAlph];
    double prob[kAlph][kAlph];
    double lambda = 0;
    uint8_t aa2num[256];
    char num2aa[kAlph + 1];
    std::string name;              // e.g. "VTML40.out" (EvalueComputation looks parameters up by name)

    // `spec` is an embedded matrix name ("VTML40.out"), a path to a .out file, or "NAME.out:DATA"
    // (commons/SubstitutionMatrix.cpp:13-51, BaseMatrix.cpp:186-211)
    bool load(const std::string &spec, double bitFactor, double scoreBias, std::string &err);
    bool parse(const std::string &text, double bitFactor, double scoreBias, std::string &err);
};

// Extended 3-mer table over the 20-letter alphabet (prefiltering/ExtendedSubstitutionMatrix.cpp:20-69).
// packed[row*8000 + rank] = (uint16 score biased by +32768) << 16 | partner 3-mer index; rows sorted by score
// descending, ties in enumeration order (first letter slowest).
void buildExt3(const SubMat &seed, std::vector<uint32_t> &packed, int threads);

// commons/SubstitutionMatrix.cpp:85-114
void localBias(const SubMat &m, const uint8_t *seq, int L, float *bias);

// commons/tantan.cpp (gap-less branch) with IndexBuilder.cpp:139-148's parameters; returns #masked residues
struct TantanModel { double lr[kAlph][kAlph]; double b2f[50]; double b2b, f2b, f2f0; };
void tantanModel(const SubMat &seed, TantanModel &tm);
int tantanMask(const TantanModel &tm, uint8_t *seq, int L, std::vector<float> &probs, std::vector<double> &scale);

struct TargetSet {
    size_t n = 0;
    std::vector<uint64_t> offsets;     // n+1, residue offsets
    std::vector<uint32_t> keys;        // DB keys (seqId -> key)
    std::vector<uint8_t> plain;        // numeric residues, unmasked (align / rescorediagonal read these)
    std::vector<uint8_t> masked;       // numeric residues after tantan (ungapped prefilter scoring reads these)
    std::vector<char> ascii;           // original letters (rescorediagonal works on ASCII)
    uint64_t maskedCount = 0;
    unsigned maxLen = 0;
    uint64_t residues() const { return offsets.empty() ? 0 : offsets.back(); }
};

struct KmerIndex {
    std::vector<uint32_t> offsets;     // kTableSize + 1
    std::vector<uint64_t> entries;     // seqId | (uint64)position << 32, lists sorted by (seqId, position)
};

// prefiltering/IndexBuilder.cpp:53-256 + IndexTable.h (addKmerCount / addSequence / sortDBSeqLists)
// [from, to) = the target range that is masked / indexed (a shard of a target-sharded search); seqIds stay global
void prepareTargets(const SubMat &seed, const char *residues, const uint64_t *offsets, const uint32_t *keys, size_t n,
                    bool mask, int threads, TargetSet &ts, size_t from = 0, size_t to = (size_t) -1);
void buildIndex(const SubMat &seed, const TargetSet &ts, int kmerThr, int threads, KmerIndex &idx, size_t from = 0, size_t to = (size_t) -1);

// Parallel sections of the host code: run f(0) .. f(tasks - 1) on the calling thread and the workers of ONE process-wide pool.  The workers are kept: a module
// call has eight or nine short parallel sections (index parse, flatten, residue mapping, two formatters, two index texts ...) and spawning 16 threads for each
// costs 0.3-0.5 ms a time (tools/micro/thread_spawn.cpp).  One section at a time: a section started while another one runs (or from inside one) runs on the
// calling thread alone.
void runTasks(size_t tasks, const std::function<void(size_t)> &f);

// host threads to use when the caller does not say: the CPUs this process may run on, capped by the cgroup CPU quota of its container
int defaultThreads();

// Prefiltering::getKmerThreshold (prefiltering/Prefiltering.cpp:987-1021), sequence search
int kmerThreshold(float sensitivity, int kmerScore, int kmerSize);

// QueryMatcher::initDiagonalMatcher (prefiltering/QueryMatcher.cpp:383-411) + Util::getL2CacheSize (Util.cpp:373-388)
uint64_t hostL2CacheSize();
unsigned binSizeFor(size_t dbSize, uint64_t l2);

// E-value statistics (alignment/EvalueComputation.h, lib/alp)
struct AlpParams { double lambda, K, a_I, b_I, alpha_I, beta_I, a_J, b_J, alpha_J, beta_J, sigma, tau; };
uint64_t matrixDigest(const SubMat &m);      // identifies the integer table of a matrix (the ALP table is keyed by name AND digest)

struct Evaluer {
    AlpParams p{};
    double dbResidues = 0;
    double logK = 0;
    bool ok = false;
    // gapOpen/gapExtend < 0 => gap-less statistics.  Parameters are the values ALP fits for the matrix/gap
    // combination; the fit itself (Monte-Carlo, lib/alp) is not rebuilt: the reference's bundled matrices over a grid of
    // gap penalties are tabulated (sp_alp_table.h, generated from the reference), anything else must be supplied through SP_ALP_PARAMS.
    bool init(const std::string &matrixName, int gapOpen, int gapExtend, uint64_t dbResidues, std::string &err, uint64_t matrixId = 0);      // matrixId: matrixDigest() of the matrix in use (0: not checked)
    double area(double score, double qLen) const;
    double evalue(double score, double qLen) const;
    double bitScore(double score) const;
};

}  // namespace sp
