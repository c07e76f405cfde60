// Host set-up code of the search path (see sp_host.h).  Compile with -ffp-contract=off: several "integer"
// tables of the reference are rounded from double/float expressions (BaseMatrix.cpp:126,148-149;
// SubstitutionMatrix.cpp:104-110; tantan.cpp), and the reference's AVX2 build has no FMA contraction.
#include "sp_host.h"

#include <algorithm>
#include <atomic>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <sched.h>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

#include "sp_matrix_data.inc"

namespace sp {

const int kPatternPos[kKmer] = {0, 1, 3, 4, 5, 7};

// ---- the worker pool behind runTasks() ------------------------------------------------------------------------
namespace {
struct WorkPool {
    std::mutex m; std::condition_variable wake, finished;
    std::vector<std::thread> workers;
    const std::function<void(size_t)> *job = nullptr; size_t tasks = 0; std::atomic<size_t> next{0}; size_t pending = 0; uint64_t generation = 0; bool busy = false, stop = false;
    void workerLoop() {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *f; size_t n;
            {   std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation; f = job; n = tasks; }
            for (size_t t = next.fetch_add(1); t < n; t = next.fetch_add(1)) (*f)(t);
            {   std::lock_guard<std::mutex> lk(m); if (--pending == 0) finished.notify_all(); }
        }
    }
    void run(size_t n, const std::function<void(size_t)> &f) {
        if (n <= 1) { if (n) f(0); return; }
        std::unique_lock<std::mutex> lk(m);
        if (busy || stop) { lk.unlock(); for (size_t t = 0; t < n; t++) f(t); return; }      // (a section inside a section, or two callers at once)
        busy = true;
        const size_t want = std::min<size_t>(n - 1, 63);
        while (workers.size() < want) workers.emplace_back([this] { workerLoop(); });
        job = &f; tasks = n; next = 0; pending = workers.size(); generation++;
        lk.unlock();
        wake.notify_all();
        for (size_t t = next.fetch_add(1); t < n; t = next.fetch_add(1)) f(t);
        lk.lock();
        finished.wait(lk, [&] { return pending == 0; });      // (every worker has looked at this job: none is left inside f)
        job = nullptr; busy = false;
    }
    ~WorkPool() {
        { std::lock_guard<std::mutex> lk(m); stop = true; }
        wake.notify_all();
        for (std::thread &t : workers) t.join();
    }
};
WorkPool &workPool() { static WorkPool *p = new WorkPool; return *p; }      // (never destroyed: a process that leaves through _exit must not wait for it, and exit() need not)
}  // namespace
void runTasks(size_t tasks, const std::function<void(size_t)> &f) { workPool().run(tasks, f); }

template <typename F>
static void parallelFor(size_t n, int threads, F f) {
    threads = std::max(1, std::min<int>(threads, (int) std::max<size_t>(1, n)));
    if (threads == 1) { f(0, 0, n); return; }
    const size_t chunk = (n + threads - 1) / threads;
    runTasks((size_t) threads, [&](size_t t) { const size_t a = std::min(n, t * chunk), b = std::min(n, a + chunk); f((int) t, a, b); });
}

// ---------------------------------------------------------------------------------------------------------------
// substitution matrix
// ---------------------------------------------------------------------------------------------------------------
static std::string renderEmbedded(const EmbeddedMatrix &e) {
    std::ostringstream o;
    o << "# Background (precomputed optional):";
    for (int i = 0; i < kAlph; i++) o << " " << e.background[i];
    o << "\n# Lambda     (precomputed optional): " << e.lambda << "\n";
    for (int i = 0; i < kAlph; i++) o << "\t" << e.letters[i];
    o << "\n";
    for (int i = 0; i < kAlph; i++) {
        o << e.letters[i];
        for (int j = 0; j < kAlph; j++) o << "\t" << (int) e.scores[i][j];
        o << "\n";
    }
    return o.str();
}

bool SubMat::load(const std::string &spec, double bitFactor, double scoreBias, std::string &err) {
    // "NAME.out:DATA" carries the table itself (commons/BaseMatrix.cpp:186-211)
    size_t p = spec.find(".out:");
    if (p != std::string::npos) {
        name = spec.substr(0, p + 4);
        return parse(spec.substr(p + 5), bitFactor, scoreBias, err);
    }
    std::string base = spec;
    size_t slash = base.find_last_of("/\\");
    if (slash != std::string::npos) base = base.substr(slash + 1);
    std::ifstream in(spec);
    if (in.good()) {
        std::stringstream ss; ss << in.rdbuf();
        name = base;
        return parse(ss.str(), bitFactor, scoreBias, err);
    }
    for (const EmbeddedMatrix &e : kEmbeddedMatrices) {
        if (base == e.name) { name = e.name; return parse(renderEmbedded(e), bitFactor, scoreBias, err); }
    }
    err = "substitution matrix '" + spec + "' is neither a readable .out file nor one of the embedded tables (VTML40.out, VTML80.out)";
    return false;
}

static std::vector<std::string> splitWs(const std::string &line) {
    std::vector<std::string> w; size_t i = 0;
    while (i < line.size()) {
        while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) i++;
        if (i >= line.size()) break;
        size_t j = i;
        while (j < line.size() && line[j] != ' ' && line[j] != '\t') j++;
        w.push_back(line.substr(i, j - i)); i = j;
    }
    return w;
}

bool SubMat::parse(const std::string &text, double bitFactor, double scoreBias, std::string &err) {
    for (int i = 0; i < 256; i++) aa2num[i] = 255;
    memset(num2aa, 0, sizeof(num2aa));
    std::vector<std::string> lines; { std::istringstream in(text); std::string l; while (std::getline(in, l)) { if (!l.empty() && l.back() == '\r') l.pop_back(); lines.push_back(l); } }
    double raw[kAlph][kAlph] = {};
    bool header = false, hasBg = false, hasLambda = false;
    for (const std::string &l : lines) {
        std::vector<std::string> w = splitWs(l);
        if (!l.empty() && l[0] == '#') {
            if (l.rfind("# Background (precomputed optional):", 0) == 0) { for (size_t i = 4; i < w.size() && i - 4 < (size_t) kAlph; i++) pBack[i - 4] = strtod(w[i].c_str(), NULL); hasBg = true; }
            if (l.rfind("# Lambda     (precomputed optional):", 0) == 0 && w.size() > 4) { lambda = strtod(w[4].c_str(), NULL); hasLambda = true; }
            continue;
        }
        if (w.size() <= 1) continue;
        if (!header) {
            if ((int) w.size() != kAlph) { err = "substitution matrix must have 21 letters (20 amino acids + X)"; return false; }
            for (int i = 0; i < kAlph; i++) { int aa = toupper((unsigned char) w[i][0]); aa2num[aa] = (uint8_t) i; num2aa[i] = (char) aa; }
            header = true;
            continue;
        }
        int aa = aa2num[toupper((unsigned char) w[0][0])];
        if (aa >= kAlph || (int) w.size() < kAlph + 1) { err = "malformed substitution matrix row"; return false; }
        for (int i = 0; i < kAlph; i++) raw[aa][i] = strtod(w[i + 1].c_str(), NULL);
    }
    if (!header || aa2num['X'] >= kAlph) { err = "substitution matrix lacks the letter header / X"; return false; }
    if (!hasBg || !hasLambda) { err = "substitution matrix lacks precomputed background/lambda header lines"; return false; }
    const int x = aa2num['X'];
    bool xPos = false;
    for (int j = 0; j < kAlph; j++) if (raw[x][j] > 0 || raw[j][x] > 0) { xPos = true; break; }
    if (!xPos) for (int i = 0; i < kAlph - 1; i++) pBack[i] = pBack[i] * (1.0 - pBack[x]);      // SubstitutionMatrix.cpp:395-399
    for (int i = 0; i < kAlph; i++) for (int j = 0; j < kAlph; j++) prob[i][j] = std::exp(lambda * raw[i][j]) * pBack[i] * pBack[j];
    // letter folding (SubstitutionMatrix.cpp:258-299)
    for (int c = 0; c < 256; c++) {
        int up = c < 128 ? toupper(c) : c;
        uint8_t v;
        switch (up) {
            case 'A': case 'C': case 'D': case 'E': case 'F': case 'G': case 'H': case 'I': case 'K': case 'L': case 'M':
            case 'N': case 'P': case 'Q': case 'R': case 'S': case 'T': case 'V': case 'W': case 'Y': case 'X': v = aa2num[up]; break;
            case 'J': v = aa2num['L']; break;
            case 'Z': v = aa2num['E']; break;
            case 'B': v = aa2num['D']; break;
            default: v = (uint8_t) x; break;      // U, O, '*', digits, ... -> X
        }
        aa2num[c] = v;
    }
    // BaseMatrix::generateSubMatrix (BaseMatrix.cpp:96-156): background := row sums, X := 1e-5
    double pb[kAlph];
    for (int i = 0; i < kAlph; i++) { pb[i] = 0; for (int j = 0; j < kAlph; j++) pb[i] += prob[i][j]; }
    pb[kAlph - 1] = 1E-5;
    for (int i = 0; i < kAlph; i++) for (int j = 0; j < kAlph; j++) {
        double v = bitFactor * std::log2(prob[i][j] / (pb[i] * pb[j])) + scoreBias;
        sub[i][j] = (short) ((v < 0.0) ? v - 0.5 : v + 0.5);
    }
    return true;
}

// ---------------------------------------------------------------------------------------------------------------
// extended 3-mer table
// ---------------------------------------------------------------------------------------------------------------
void buildExt3(const SubMat &seed, std::vector<uint32_t> &packed, int threads) {
    const int A = 20, N = A * A * A;
    packed.assign((size_t) N * N, 0);
    // enumeration order e = l0*400 + l1*20 + l2 (first letter slowest); k-mer index = l0 + 20*l1 + 400*l2
    parallelFor(N, threads, [&](int, size_t ea, size_t eb) {
        std::vector<short> sc(N);
        std::vector<int> cnt;
        for (size_t ei = ea; ei < eb; ei++) {
            const int i0 = (int) ei / 400, i1 = ((int) ei / 20) % 20, i2 = (int) ei % 20;
            int lo = 1 << 20, hi = -(1 << 20);
            for (int j0 = 0, ej = 0; j0 < A; j0++) for (int j1 = 0; j1 < A; j1++) {
                const int s01 = seed.sub[i0][j0] + seed.sub[i1][j1];
                for (int j2 = 0; j2 < A; j2++, ej++) { int s = s01 + seed.sub[i2][j2]; sc[ej] = (short) s; lo = std::min(lo, s); hi = std::max(hi, s); }
            }
            cnt.assign(hi - lo + 2, 0);
            for (int ej = 0; ej < N; ej++) cnt[hi - sc[ej] + 1]++;
            for (int r = 0; r + 1 < (int) cnt.size(); r++) cnt[r + 1] += cnt[r];
            uint32_t *row = packed.data() + (size_t) (i0 + 20 * i1 + 400 * i2) * N;
            for (int ej = 0; ej < N; ej++) {
                const int j0 = ej / 400, j1 = (ej / 20) % 20, j2 = ej % 20;
                const int at = cnt[hi - sc[ej]]++;
                row[at] = ((uint32_t) (uint16_t) (sc[ej] + 32768) << 16) | (uint32_t) (j0 + 20 * j1 + 400 * j2);
            }
        }
    });
}

// ---------------------------------------------------------------------------------------------------------------
// composition bias
// ---------------------------------------------------------------------------------------------------------------
void localBias(const SubMat &m, const uint8_t *seq, int N, float *bias) {
    const int half = 20;    // windowSize 40
    for (int i = 0; i < N; i++) {
        const int minPos = std::max(0, i - half), maxPos = std::min(N, i + half);
        const short *row = m.sub[seq[i]];
        int sum = 0;
        for (int j = minPos; j < maxPos; j++) sum += row[seq[j]];
        sum -= row[seq[i]];
        float d = (float) sum;
        d = (float) ((double) d / (-1.0 * (double) (float) (maxPos - minPos)));
        for (int a = 0; a < m.alph; a++) d = (float) ((double) d + m.pBack[a] * (double) (float) row[a]);
        bias[i] = d;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// tantan
// ---------------------------------------------------------------------------------------------------------------
void tantanModel(const SubMat &seed, TantanModel &tm) {
    const double repeatProb = 0.005, repeatEndProb = 0.05, decay = 0.9;
    const int maxOff = 50;
    for (int i = 0; i < kAlph; i++) for (int j = 0; j < kAlph; j++) tm.lr[i][j] = seed.prob[i][j] / (seed.pBack[i] * seed.pBack[j]);
    tm.b2b = 1 - repeatProb; tm.f2b = repeatEndProb; tm.f2f0 = 1 - repeatEndProb;
    volatile double pm = decay;
    double first = (pm < 1 || pm > 1) ? (1 - pm) / (1 - std::pow((double) pm, (double) maxOff)) : 1.0 / maxOff;
    double p = repeatProb * first;
    for (int i = 0; i < maxOff; i++) { tm.b2f[i] = p; p *= decay; }
}

int tantanMask(const TantanModel &tm, uint8_t *seq, int L, std::vector<float> &probs, std::vector<double> &scale) {
    constexpr int MAXOFF = 50, STEP = 16;
    if (L <= 0) return 0;
    probs.resize(L); scale.assign(L / STEP + 1, 0.0);
    double fg[MAXOFF];
    for (double &v : fg) v = 0.0;
    double bg = 1.0;
    for (int pos = 0; pos < L; pos++) {
        const double b = bg; double fromFg = 0;
        const double *row = tm.lr[seq[pos]];
        const int maxOffset = std::min(pos, MAXOFF);
        for (int i = 0; i < maxOffset; i++) {
            const double f = fg[i];
            fromFg += f;
            fg[i] = (b * tm.b2f[i] + f * tm.f2f0) * row[seq[pos - i - 1]];
        }
        bg = b * tm.b2b + fromFg * tm.f2b;
        if (pos % STEP == STEP - 1) { const double s = 1 / bg; scale[pos / STEP] = s; bg *= s; for (double &v : fg) v *= s; }
        probs[pos] = (float) bg;
    }
    double acc = 0.0; for (double v : fg) acc += v;
    const double z = bg * tm.b2b + acc * tm.f2b;
    bg = tm.b2b;
    for (double &v : fg) v = tm.f2b;
    for (int pos = L - 1; pos >= 0; pos--) {
        const double nonRepeat = (double) probs[pos] * bg / z;
        probs[pos] = 1 - (float) nonRepeat;
        if (pos % STEP == STEP - 1) { const double s = scale[pos / STEP]; bg *= s; for (double &v : fg) v *= s; }
        const double toBg = tm.f2b * bg; double toFg = 0;
        const double *row = tm.lr[seq[pos]];
        const int maxOffset = std::min(pos, MAXOFF);
        for (int i = 0; i < maxOffset; i++) {
            const double f = fg[i] * row[seq[pos - i - 1]];
            toFg += tm.b2f[i] * f;
            fg[i] = toBg + tm.f2f0 * f;
        }
        bg = tm.b2b * bg + toFg;
    }
    int masked = 0;
    for (int pos = 0; pos < L; pos++) if ((double) probs[pos] >= 0.9) { seq[pos] = kX; masked++; }
    return masked;
}

// ---------------------------------------------------------------------------------------------------------------
// targets + index
// ---------------------------------------------------------------------------------------------------------------
void prepareTargets(const SubMat &seed, const char *residues, const uint64_t *offsets, const uint32_t *keys, size_t n,
                    bool mask, int threads, TargetSet &ts, size_t from, size_t to) {
    to = std::min(to, n); from = std::min(from, to);
    ts.n = n;
    ts.offsets.assign(offsets, offsets + n + 1);
    ts.keys.resize(n);
    for (size_t i = 0; i < n; i++) ts.keys[i] = keys ? keys[i] : (uint32_t) i;
    const uint64_t R = offsets[n];
    ts.ascii.assign(residues, residues + R);
    ts.plain.resize(R); ts.masked.resize(R);
    ts.maxLen = 0;
    for (size_t i = 0; i < n; i++) ts.maxLen = std::max<unsigned>(ts.maxLen, (unsigned) (offsets[i + 1] - offsets[i]));
    TantanModel tm; tantanModel(seed, tm);
    std::atomic<uint64_t> maskedTotal(0);
    parallelFor(n, threads, [&](int, size_t a, size_t b) {
        std::vector<float> probs; std::vector<double> scale;
        uint64_t local = 0;
        for (size_t t = a; t < b; t++) {
            const uint64_t o = offsets[t]; const int L = (int) (offsets[t + 1] - o);
            for (int i = 0; i < L; i++) ts.plain[o + i] = seed.aa2num[(uint8_t) residues[o + i]];
            memcpy(ts.masked.data() + o, ts.plain.data() + o, (size_t) L);
            if (mask && t >= from && t < to) local += (uint64_t) tantanMask(tm, ts.masked.data() + o, L, probs, scale);
        }
        maskedTotal += local;
    });
    ts.maskedCount = maskedTotal.load();
}

void buildIndex(const SubMat &seed, const TargetSet &ts, int kmerThr, int threads, KmerIndex &idx, size_t from, size_t to) {
    to = std::min(to, ts.n); from = std::min(from, to);
    struct KP { uint32_t kmer; uint16_t pos; };
    // per target: distinct indexable k-mers, each at its smallest position (IndexTable.h:130-170,341-394)
    std::vector<std::vector<uint64_t>> perThread(std::max(1, threads));     // kmer<<32 | seq-local (count<<16|pos) is not needed; store kmer<<16|pos
    std::vector<std::pair<size_t, size_t>> ranges(std::max(1, threads), {0, 0});
    std::vector<std::vector<uint32_t>> counts(std::max(1, threads));        // #entries per target of the range
    signed char self[kAlph];
    for (int a = 0; a < kAlph; a++) self[a] = (signed char) seed.sub[a][a];   // getScoreLookup, IndexBuilder.cpp:8-19
    parallelFor(to - from, threads, [&](int tid, size_t a, size_t b) {
        a += from; b += from;
        ranges[tid] = {a, b};
        std::vector<KP> buf;
        std::vector<uint64_t> &out = perThread[tid];
        std::vector<uint32_t> &cnt = counts[tid];
        cnt.reserve(b - a);
        for (size_t t = a; t < b; t++) {
            const uint8_t *s = ts.masked.data() + ts.offsets[t];
            const int L = (int) (ts.offsets[t + 1] - ts.offsets[t]);
            buf.clear();
            for (int i = 0; i + kSpan <= L; i++) {
                bool hasX = false; int selfScore = 0; uint32_t k = 0, pw = 1;
                for (int p = 0; p < kKmer; p++) { const uint8_t r = s[i + kPatternPos[p]]; hasX |= (r == kX); selfScore += self[r]; k += r * pw; pw *= 20; }
                if (hasX) continue;
                if (kmerThr > 0 && selfScore < kmerThr) continue;
                buf.push_back({k, (uint16_t) i});
            }
            std::sort(buf.begin(), buf.end(), [](const KP &x, const KP &y) { return x.kmer != y.kmer ? x.kmer < y.kmer : x.pos < y.pos; });
            uint32_t c = 0;
            for (size_t j = 0; j < buf.size(); j++) if (j == 0 || buf[j].kmer != buf[j - 1].kmer) { out.push_back(((uint64_t) buf[j].kmer << 16) | buf[j].pos); c++; }
            cnt.push_back(c);
        }
    });
    idx.offsets.assign((size_t) kTableSize + 1, 0);
    for (auto &v : perThread) for (uint64_t e : v) idx.offsets[(size_t) (e >> 16) + 1]++;
    for (size_t k = 0; k < kTableSize; k++) idx.offsets[k + 1] += idx.offsets[k];
    idx.entries.assign(idx.offsets[kTableSize], 0);
    std::vector<uint32_t> cursor(idx.offsets.begin(), idx.offsets.end() - 1);
    // scatter in target order so that every list comes out sorted by (seqId, position) (IndexTable.h:179-186)
    for (size_t tid = 0; tid < perThread.size(); tid++) {
        const std::vector<uint64_t> &v = perThread[tid];
        size_t at = 0;
        for (size_t t = ranges[tid].first, j = 0; t < ranges[tid].second; t++, j++) {
            for (uint32_t c = 0; c < counts[tid][j]; c++, at++) {
                const uint32_t k = (uint32_t) (v[at] >> 16); const uint16_t pos = (uint16_t) (v[at] & 0xFFFF);
                idx.entries[cursor[k]++] = (uint64_t) (uint32_t) t | ((uint64_t) pos << 32);
            }
        }
    }
}

int kmerThreshold(float sensitivity, int kmerScore, int kmerSize) {
    double best = kmerScore;
    if (kmerScore == 2147483647) {
        if (kmerSize == 5) { float base = 160.75; best = base - (sensitivity * 12.75); }
        else if (kmerSize == 6) { float base = 163.2; best = base - (sensitivity * 8.917); }
        else if (kmerSize == 7) { float base = 186.15; best = base - (sensitivity * 11.22); }
        else return -1;
    }
    return (int) best;
}

uint64_t hostL2CacheSize() {
#if defined(_SC_LEVEL2_CACHE_SIZE)
    long v = sysconf(_SC_LEVEL2_CACHE_SIZE);
    if (v > 0) return (uint64_t) v;
#endif
    return 262144;
}

unsigned binSizeFor(size_t dbSize, uint64_t l2) {
    for (unsigned b = 2; b <= 1024; b *= 2) if (dbSize / b < l2) return b;
    return 2048;
}

// ---------------------------------------------------------------------------------------------------------------
// E-values
// ---------------------------------------------------------------------------------------------------------------
struct AlpEntry { const char *matrix; int gapOpen, gapExtend; const char *hex[12]; };
// Gumbel parameters as fitted by ALP (lib/alp, seed 42) for the combinations SpacePHARER uses
// (PredictMatch.cpp: VTML40, gap open 16, extend 2; rescorediagonal: gap-less VTML40).  Captured from the
// reference (tests/golden/*.json.gz: alpGapped / alpGapless); identical in every run of the reference here.
static const AlpEntry kAlpTable[] = {
    {"VTML40.out", 16, 2, {"0x1.6076ff151a18p-2", "0x1.197881179aaf3p-2", "0x1.c2f26d4420b5cp-3", "0x1.ade01ae0e75ap-6", "0x1.7a804890a2d29p-3", "0x1.b672cb9380acdp+0",
                           "0x1.c2f26d4420b5cp-3", "0x1.ade01ae0e75ap-6", "0x1.7a804890a2d29p-3", "0x1.b672cb9380acdp+0", "0x1.7674244e3b2f3p-3", "0x1.c8a96ebe530cp+0"}},
    {"VTML40.out", -1, -1, {"0x1.614cc810f4edcp-2", "0x1.15f37f76ce8e9p-2", "0x1.c47089cdcc2e1p-3", "0x0p+0", "0x1.dbef207886873p-3", "0x0p+0",
                            "0x1.c47089cdcc2e1p-3", "0x0p+0", "0x1.dbef207886873p-3", "0x0p+0", "0x1.dbef207886873p-3", "0x0p+0"}},
    // proteinaln2nucl: NucleotideMatrix(nucleotide.out, 1.0, 0.0), gap open 10, extend 2 (tests/golden/examples.json.gz: info.nucl_evalue.alp)
    {"nucleotide.out", 10, 2, {"0x1.4469d1f7ec6c4p-1", "0x1.af9f955180647p-2", "0x1.5f922abcb8de4p-1", "0x1.838e08750576p-3", "0x1.6cacd8a69e238p-1", "0x1.73f45a27d096ep+1",
                               "0x1.5f922abcb8de4p-1", "0x1.838e08750576p-3", "0x1.6cacd8a69e238p-1", "0x1.73f45a27d096ep+1", "0x1.6cd5a205f28f5p-1", "0x1.72ffa1ebd61p+1"}},
    // the module's own defaults (Parameters.cpp: nucleotide gap open 5, extend 2), fitted by `oracle/_ref/ref_sets nuclalp <db> nucleotide.out 5 2`
    {"nucleotide.out", 5, 2, {"0x1.3de995e742594p-1", "0x1.6837f664724b1p-2", "0x1.7d956af2b60d6p-1", "-0x1.6ba8cc62c7c7ep-1", "0x1.037654c94bd2ep+0", "-0x1.42e626a2af9a8p+1",
                              "0x1.7d956af2b60d6p-1", "-0x1.6ba8cc62c7c7ep-1", "0x1.037654c94bd2ep+0", "-0x1.42e626a2af9a8p+1", "0x1.00d16215dfda5p+0", "-0x1.306383babbceap+1"}},
};

// The reference's bundled amino-acid matrices over a grid of gap penalties (gap open 5..20, extend 1..2, and gap-less), fitted by the reference's own EvalueComputation / ALP:
// tools/gen_alp_table.py.  An entry applies to a matrix of that NAME whose integer table (bit factor 2.0) has that digest.
struct AlpGridEntry { const char *matrix; uint64_t id; int gapOpen, gapExtend; double v[12]; };
static const AlpGridEntry kAlpGrid[] = {
#include "sp_alp_table.h"
};

uint64_t matrixDigest(const SubMat &m) {      // FNV-1a 64 over (letter i, letter j, score as int16 little endian); tools/gen_alp_table.py: matrix_id
    uint64_t h = 1469598103934665603ULL;
    for (int i = 0; i < m.alph; i++) for (int j = 0; j < m.alph; j++) {
        const uint16_t s = (uint16_t) m.sub[i][j];
        const unsigned char b[4] = {(unsigned char) m.num2aa[i], (unsigned char) m.num2aa[j], (unsigned char) (s & 0xFF), (unsigned char) (s >> 8)};
        for (unsigned char x : b) { h ^= x; h *= 1099511628211ULL; }
    }
    return h;
}

bool Evaluer::init(const std::string &matrixName, int gapOpen, int gapExtend, uint64_t dbRes, std::string &err, uint64_t matrixId) {
    dbResidues = (double) dbRes;
    double v[12]; bool found = false; bool otherMatrix = false;
    if (const char *env = getenv("SP_ALP_PARAMS")) {        // 12 comma separated numbers: lambda,K,a_I,b_I,alpha_I,beta_I,a_J,b_J,alpha_J,beta_J,sigma,tau
        std::string s(env); size_t at = 0; int k = 0;
        while (k < 12 && at < s.size()) { size_t c = s.find(',', at); if (c == std::string::npos) c = s.size(); v[k++] = strtod(s.substr(at, c - at).c_str(), NULL); at = c + 1; }
        found = (k == 12);
    }
    if (!found) for (const AlpEntry &e : kAlpTable) {
        if (matrixName == e.matrix && ((gapOpen < 0 && e.gapOpen < 0) || (gapOpen == e.gapOpen && gapExtend == e.gapExtend))) {
            for (int k = 0; k < 12; k++) v[k] = strtod(e.hex[k], NULL);
            found = true; break;
        }
    }
    if (!found) for (const AlpGridEntry &e : kAlpGrid) {
        if (matrixName != e.matrix || !((gapOpen < 0 && e.gapOpen < 0) || (gapOpen == e.gapOpen && gapExtend == e.gapExtend))) continue;
        if (matrixId != 0 && matrixId != e.id) { otherMatrix = true; continue; }      // (a file that shares a bundled matrix's name is not that matrix)
        for (int k = 0; k < 12; k++) v[k] = e.v[k];
        found = true; break;
    }
    if (!found) {
        err = "no tabulated ALP (Gumbel) parameters for matrix " + matrixName + " with gap open/extend " + std::to_string(gapOpen) + "/" + std::to_string(gapExtend) +
              (otherMatrix ? " (the table has this name with other scores)" : "") +
              ": the reference fits them at run time (lib/alp); tabulated are its bundled matrices with gap open 5..20, extend 1..2 where ALP can fit them (tools/gen_alp_table.py); "
              "supply others through SP_ALP_PARAMS";
        return false;
    }
    p = AlpParams{v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11]};
    logK = std::log(p.K);
    ok = true;
    return true;
}

// lib/alp/sls_pvalues.cpp:342-364 (thresholds) and :366-560 (area); normal_probability = sls_basic.hpp:195-198
double Evaluer::area(double y, double qLen) const {
    const double pi = 3.1415926535897932384626433832795;
    const double constVal = 1 / std::sqrt(2.0 * pi);
    const double m_ = dbResidues, n_ = qLen;
    const double viThr = std::max(2.0 * p.alpha_I / p.lambda, 0.0), vjThr = std::max(2.0 * p.alpha_J / p.lambda, 0.0), cThr = std::max(2.0 * p.sigma / p.lambda, 0.0);
    double tmp = p.a_I * y + p.b_I;
    const double mLi = m_ - tmp;
    const double viY = std::max(viThr, p.alpha_I * y + p.beta_I);
    const double sqrtVi = std::sqrt(viY);
    const double mF = (sqrtVi == 0.0) ? 1e100 : mLi / sqrtVi;
    const double PmF = 0.5 * std::erfc(-std::sqrt(0.5) * mF);
    const double EmF = -constVal * std::exp(-0.5 * mF * mF);
    const double p1 = mLi * PmF - sqrtVi * EmF;
    tmp = p.a_J * y + p.b_J;
    const double nLj = n_ - tmp;
    const double vjY = std::max(vjThr, p.alpha_J * y + p.beta_J);
    const double sqrtVj = std::sqrt(vjY);
    const double nF = (sqrtVj == 0.0) ? 1e100 : nLj / sqrtVj;
    const double PnF = 0.5 * std::erfc(-std::sqrt(0.5) * nF);
    const double EnF = -constVal * std::exp(-0.5 * nF * nF);
    const double p2 = nLj * PnF - sqrtVj * EnF;
    const double cY = std::max(cThr, p.sigma * y + p.tau);
    const double PP = PmF * PnF;
    const double cYPP = cY * PP;
    const double p1p2 = p1 * p2;
    return p1p2 + cYPP;
}

double Evaluer::evalue(double score, double qLen) const {
    const double epa = p.K * std::exp(-p.lambda * score);
    const double a = area(score, qLen);
    return epa * a;
}

double Evaluer::bitScore(double score) const { return (p.lambda * score - logK) / std::log(2.0); }

int defaultThreads() {
    static const int n = [] {
        int hw = (int) std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t set; CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) hw = std::min(hw, k); }
        std::ifstream f("/sys/fs/cgroup/cpu.max");       // "<quota> <period>" or "max <period>"
        std::string q; double period = 0;
        if (f.good() && (f >> q >> period) && q != "max" && period > 0) { const int k = (int) (strtod(q.c_str(), NULL) / period + 0.5); if (k >= 1) hw = std::min(hw, k); }
        return std::max(1, hw);
    }();
    return n;
}

}  // namespace sp
