// Set-level steps around the search hot path: see sp_sets.cpp.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

#include "sp_db.h"
#include "sp_host.h"

namespace sp {
namespace sets {

struct OrfLoc { size_t from, to; bool incStart, incEnd; int strand; };      // Orf::SequenceLocation
struct OrfFinder {
    std::string fwd, rev; size_t len = 0;
    void setSequence(const char *seq, size_t L);
    void findAll(size_t minLen, size_t maxLen, size_t maxGaps, unsigned fwdFrames, unsigned revFrames, unsigned startMode, std::vector<OrfLoc> &out) const;
private:
    void scan(const std::string &s, int strand, size_t minLen, size_t maxLen, size_t maxGaps, unsigned frames, unsigned startMode, std::vector<OrfLoc> &out) const;
};
std::string orfHeader(uint32_t key, size_t from, size_t to, bool incStart, bool incEnd);
void translate(const char *nucl, size_t len, char *aa);
size_t translatableLength(const char *data, size_t entryLen, size_t maxSeqLen);
void reverseByCodon(const char *seq, size_t L, std::string &out);

struct NuclScorer {
    int gapOpen = 10, gapExtend = 2; Evaluer ev;
    bool init(const std::string &matrixName, int gapOpen, int gapExtend, uint64_t dbResidues, std::string &err);
    int score(unsigned char a, unsigned char b) const { return score_[cls_[a]][cls_[b]]; }
private:
    unsigned char cls_[256]; int score_[5][5];
};

// key -> unsigned value tables (member -> set, set -> size)
struct SetLookup {
    KeyMap index; std::vector<uint32_t> values;
    bool fromDb(const Db &db);
    void fromArrays(const std::vector<uint32_t> &keys, const std::vector<uint32_t> &vals);
    bool find(uint32_t key, uint32_t &v) const;
};

bool proteinLineToNucl(const char *line, const NuclScorer &sc, const char *qNuc, uint32_t qNucLen, bool qStartCodon, const char *tNuc, uint32_t tNucLen, bool tStartCodon,
                       std::string &out, std::string &err);
bool combineProtNucl(const char *prot, const char *nucl, std::string &out, std::string &err);
bool bestHitPerSet(const char *entry, const SetLookup &memberToSet, const SetLookup &setSize, bool simple, std::string &out, std::string &err);
bool combineScore(const char *entry, uint32_t querySetSize, const SetLookup &memberToSet, std::string &out, std::string &err);
struct FdrResult { double threshold = 0; std::vector<double> uniqueScores, fdr; };
void fdrThreshold(std::vector<double> P, std::vector<double> N, bool haveControl, double cutoff, bool reportFdr, FdrResult &r);

// In-memory form of predictmatch's chain after the search (data/predictmatch.sh:27-75): alignment records of a query batch ->
// align-module text -> proteinaln2nucl -> combineprotnuclaln -> besthitperset -> mergeresultsbyset -> combinescore, no DB in between.
struct SetInputs {
    size_t nq = 0, nt = 0;
    const sp_alignment *aln = nullptr; const char *cigars = nullptr;      // records of hit i, uncompressed backtraces
    const uint32_t *hitOffsets = nullptr; const uint32_t *hitTarget = nullptr;      // hits of query q: [hitOffsets[q], hitOffsets[q+1]); target index of hit i
    const char *qNucl = nullptr; const uint64_t *qNuclOff = nullptr; const char *tNucl = nullptr; const uint64_t *tNuclOff = nullptr;
    const uint32_t *qSet = nullptr, *tSet = nullptr, *qKeys = nullptr, *tKeys = nullptr;      // set key per query / target; DB keys (nullptr: the index)
    const char *qFirst = nullptr, *tFirst = nullptr;      // first residue of every query / target peptide ('*' = ORF with a start codon marker)
    std::string nuclMatrix = "nucleotide.out"; int gapOpen = 10, gapExtend = 2; bool simpleBestHit = true; int threads = 1;
};
// stage: 0 cScore per query set | 1 align text | 2 nucleotide alignment | 3 combined | 4 best hit per set (1-4: per query) | 5 merged per query set
bool setScores(const SetInputs &in, int stage, std::vector<uint32_t> &keys, std::vector<std::string> &entries, std::string &err);

int runExtractOrfs(int argc, const char **argv);
int runTranslateNucs(int argc, const char **argv);
int runReverseSeqByCodon(int argc, const char **argv);
int runProteinAln2Nucl(int argc, const char **argv);
int runCombineProtNuclAln(int argc, const char **argv);
int runBestHitPerSet(int argc, const char **argv);
int runMergeResultsBySet(int argc, const char **argv);
int runCombineScore(int argc, const char **argv);
int runFilterMatchByFdr(int argc, const char **argv);
int runSummarizeResults(int argc, const char **argv);

}  // namespace sets
}  // namespace sp
