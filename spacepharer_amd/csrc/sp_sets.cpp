// Set-level steps around the search hot path: host-side text / integer transformations on
// MMseqs2 DB triples, as module drop-ins (same argv as the reference modules) and as one in-memory pipeline fed with the
// alignment records the device delivers (sp_ctx_set_scores, sp_api.cpp).  They replace the bodies of
//   lib/mmseqs/src/util/extractorfs.cpp:18-160            (extractorfs; ORF finder: commons/Orf.cpp:150-330)
//   lib/mmseqs/src/util/translatenucs.cpp:16-115          (translatenucs; codon tables: commons/TranslateNucl.h:334-503)
//   src/util/ReverseSeqByCodon.cpp:13-62                  (reverseseqbycodon)
//   lib/mmseqs/src/util/proteinaln2nucl.cpp:13-205        (proteinaln2nucl)
//   src/util/CombineProtNuclAln.cpp:13-106                (combineprotnuclaln)
//   lib/mmseqs/src/multihit/besthitperset.cpp:10-129 + Aggregation.cpp:24-91   (besthitperset)
//   lib/mmseqs/src/util/mergeresultsbyset.cpp:10-58       (mergeresultsbyset)
//   src/util/CombineScore.cpp:12-113                      (combinescore)
//   src/util/FilterMatchbyFdr.cpp:13-264                  (filtermatchbyfdr)
//   src/util/SummarizeResults.cpp:13-197                  (summarizeresults; taxonomy columns from the NCBI dump files next to the target set)
// No kernels here: these steps move a few bytes per hit; what they gain on the MI355X path is that they can run on the
// records in host memory without the DB round trips in between.
#include "sp_sets.h"

#include <cfloat>
#include <climits>
#include <cmath>
#include <atomic>
#include <chrono>
#include <cstring>
#include <thread>

#include "sp_db.h"
#include "sp_host.h"

namespace sp {
namespace sets {

// ------------------------------------------------------------------------------------------------------------------
// small text helpers: the reference's number formats
// ------------------------------------------------------------------------------------------------------------------
static std::string fmtE(double x) { char b[40]; return std::string(b, (size_t) (fmtE3(x, b) - b)); }      // SSTR(double), Util.cpp:713-718
static inline bool isWs(char c) { return c == ' ' || c == '\t'; }
// Util::getWordsOfLine (Util.h:232-246): word starts of one line; words[n] = the line end
static size_t words(const char *p, const char **w, size_t maxW) {
    size_t n = 0;
    while (*p != '\n' && *p != '\0') {
        while (isWs(*p)) p++;
        w[n++] = p;
        if (n >= maxW) return n;
        while (*p != '\n' && *p != '\0' && !isWs(*p)) p++;
    }
    if (n < maxW) w[n] = p;
    return n;
}
static const char *skipLine(const char *p) { while (*p != '\n' && *p != '\0') p++; return *p == '\n' ? p + 1 : p; }
static unsigned long atou(const char *p) { unsigned long v = 0; while (*p >= '0' && *p <= '9') v = v * 10 + (unsigned long) (*p++ - '0'); return v; }
static long atoiS(const char *p) { bool neg = *p == '-'; if (neg || *p == '+') p++; unsigned long u = atou(p); if (u > (unsigned long) LONG_MAX) u = (unsigned long) LONG_MAX; const long v = (long) u; return neg ? -v : v; }

// ------------------------------------------------------------------------------------------------------------------
// ORFs (commons/Orf.cpp)
// ------------------------------------------------------------------------------------------------------------------
// IUPAC complement; anything else -> '.'  (Orf.cpp:46-50)
static char complementOf(unsigned char c) {
    static const char *upper = "TVGH..CD..M.KN...YSAABW.R";      // 'A' .. 'Y'
    if (c >= 'A' && c <= 'Y') return upper[c - 'A'];
    if (c >= 'a' && c <= 'y') { const char u = upper[c - 'a']; return u == '.' ? '.' : (char) (u | 0x20); }
    return '.';
}

// per byte: its upper-case form (the CHAR_MAX sentinel stays), whether it counts as a gap (N or not an IUPAC letter, Orf.cpp:268-272)
// and its complement on the reverse strand (non-IUPAC -> 'N')
struct BaseTables {
    unsigned char up[256], gap[256]; char comp[256];
    BaseTables() {
        for (int c = 0; c < 256; c++) {
            up[c] = c == CHAR_MAX ? (unsigned char) CHAR_MAX : (unsigned char) ((char) c & (char) ~0x20);
            const char k = complementOf((unsigned char) c); comp[c] = k == '.' ? 'N' : k;
        }
        for (int c = 0; c < 256; c++) gap[c] = (c == 'N' || complementOf((unsigned char) c) == '.') ? 1 : 0;      // indexed by the upper-cased byte
    }
};
static const BaseTables &baseTables() { static const BaseTables t; return t; }

void OrfFinder::setSequence(const char *seq, size_t L) {
    // Orf::setSequence (Orf.cpp:118-148): 'u' -> 't' (the 'U' -> 'T' assignment is overwritten there and has no effect);
    // both strands are followed by a few CHAR_MAX sentinels so that the last codons read "incomplete"
    const BaseTables &t = baseTables();
    len = L;
    fwd.resize(L + 8); rev.resize(L + 8);
    char *f = &fwd[0], *r = &rev[0];
    for (size_t i = 0; i < L; i++) f[i] = seq[i] == 'u' ? 't' : seq[i];
    for (size_t i = 0; i < L; i++) r[i] = t.comp[(unsigned char) f[L - 1 - i]];
    memset(f + L, CHAR_MAX, 8); memset(r + L, CHAR_MAX, 8);
}

// Orf::findForward (Orf.cpp:200-330), translation table 1: stop codons TAA TAG TGA, start codon ATG.  The reference walks the
// strand codon by codon and visits the three frames of each codon position; that is every position 0 .. 3 * ceil((len - 2) / 3) - 1
// in turn, frame = position % 3.  A codon is "incomplete" when it reaches into the sentinels, i.e. position + 2 >= len.
void OrfFinder::scan(const std::string &s, int strand, size_t minLen, size_t maxLen, size_t maxGaps, unsigned frames, unsigned startMode, std::vector<OrfLoc> &out) const {
    if (len < 3) return;
    const BaseTables &t = baseTables();
    const unsigned char *p = (const unsigned char *) s.data();
    bool inside[3] = {true, true, true}, hasStart[3] = {false, false, false};
    size_t gaps[3] = {0, 0, 0}, count[3] = {0, 0, 0}, from[3] = {0, 1, 2};
    const size_t end = ((len - 2 + 2) / 3) * 3;
    unsigned char c0 = t.up[p[0]], c1 = t.up[p[1]];
    unsigned fr = 0;
    for (size_t pos = 0; pos < end; pos++, fr = fr == 2 ? 0 : fr + 1) {
        const unsigned char a = c0, b = c1, c = t.up[p[pos + 2]];
        c0 = b; c1 = c;
        if (!(frames & (1u << fr))) continue;
        const bool last = pos + 2 < len && pos + 5 >= len;
        const bool isStartCodon = a == 'A' && b == 'T' && c == 'G';
        bool start;
        if (startMode == 0) start = !inside[fr] && isStartCodon;            // START_TO_STOP
        else if (startMode == 1) start = !inside[fr];                      // ANY_TO_STOP
        else start = isStartCodon;                                         // LAST_START_TO_STOP
        if (start) { inside[fr] = true; hasStart[fr] = true; from[fr] = pos; gaps[fr] = 0; count[fr] = 0; }
        if (!inside[fr]) continue;
        const bool stop = a == 'T' && ((b == 'A' && (c == 'A' || c == 'G')) || (b == 'G' && c == 'A'));
        if (!stop) count[fr]++;
        gaps[fr] += (size_t) (t.gap[a] | t.gap[b] | t.gap[c]);
        if (stop || last) {
            inside[fr] = false;
            if (count[fr] == 0 && stop) continue;
            const size_t to = (last && !stop) ? pos + 2 : pos - 1;
            if (gaps[fr] > maxGaps || count[fr] > maxLen || count[fr] < minLen) continue;
            out.push_back(OrfLoc{from[fr], to, !hasStart[fr], !stop, strand});
        }
    }
}

void OrfFinder::findAll(size_t minLen, size_t maxLen, size_t maxGaps, unsigned fwdFrames, unsigned revFrames, unsigned startMode, std::vector<OrfLoc> &out) const {
    if (fwdFrames) scan(fwd, 1, minLen, maxLen, maxGaps, fwdFrames, startMode, out);
    if (revFrames) scan(rev, -1, minLen, maxLen, maxGaps, revFrames, startMode, out);
}

// Orf::writeOrfHeader (Orf.cpp:437-457)
std::string orfHeader(uint32_t key, size_t from, size_t to, bool incStart, bool incEnd) {
    char b[96]; char *p = u32toa(key, b); *p++ = '\t';
    p = u32toa((uint32_t) from, p); *p++ = from < to ? '+' : '-';
    p = i32toa(abs((int) from - (int) to), p);
    const int complete = (incStart ? 1 : 0) | (incEnd ? 2 : 0);
    if (complete) { *p++ = '\t'; p = i32toa(complete, p); }
    *p++ = '\n';
    return std::string(b, (size_t) (p - b));
}

// ------------------------------------------------------------------------------------------------------------------
// translation (commons/TranslateNucl.h): a codon of IUPAC letters translates to the residue all its expansions agree on,
// B / Z / J for Asx / Glx / Xle, X otherwise (and for letters outside the IUPAC alphabet); a lower-case base gives a
// lower-case residue
// ------------------------------------------------------------------------------------------------------------------
struct CodonTable {
    unsigned char base[256]; char aa[16][16][16];
    CodonTable() {
        memset(base, 0, sizeof(base));
        const char *letters = "-ACMGRSVTWYHKDBN";      // ncbi4na: bit 0 = A, 1 = C, 2 = G, 3 = T
        for (int i = 0; i < 16; i++) { base[(unsigned char) letters[i]] = (unsigned char) i; base[(unsigned char) tolower(letters[i])] = (unsigned char) i; }
        base[(unsigned char) 'U'] = base[(unsigned char) 'u'] = 8; base[(unsigned char) 'X'] = base[(unsigned char) 'x'] = 15;
        for (int i = 0; i < 16; i++) base[i] = (unsigned char) i;
        const char *code = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";      // table 1, TCAG order
        const int tcag[9] = {0, 2, 1, 0, 3, 0, 0, 0, 0};      // bit value -> position in TCAG
        const int bits[4] = {1, 2, 4, 8};
        for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) for (int k = 0; k < 16; k++) {
            char a = 0;
            for (int x : bits) if (x & i) for (int y : bits) if (y & j) for (int z : bits) if (z & k) {
                const char ch = code[16 * tcag[x] + 4 * tcag[y] + tcag[z]];
                if (!a) a = ch;
                else if (a != ch) {
                    if ((a == 'B' || a == 'D' || a == 'N') && (ch == 'D' || ch == 'N')) a = 'B';
                    else if ((a == 'Z' || a == 'E' || a == 'Q') && (ch == 'E' || ch == 'Q')) a = 'Z';
                    else if ((a == 'J' || a == 'I' || a == 'L') && (ch == 'I' || ch == 'L')) a = 'J';
                    else a = 'X';
                }
            }
            aa[i][j][k] = a ? a : 'X';
        }
    }
};
static const CodonTable &codonTable() { static const CodonTable t; return t; }

void translate(const char *nucl, size_t len, char *aa) {
    const CodonTable &t = codonTable();
    for (size_t i = 0; i + 2 < len; i += 3) {
        const unsigned char a = (unsigned char) nucl[i], b = (unsigned char) nucl[i + 1], c = (unsigned char) nucl[i + 2];
        const char r = t.aa[t.base[a]][t.base[b]][t.base[c]];
        aa[i / 3] = (islower(a) || islower(b) || islower(c)) ? (char) tolower(r) : r;
    }
}

// translatenucs' length rule (translatenucs.cpp:77-92): data = the entry ("...\n\0"), entryLen includes the NUL; returns the
// number of nucleotides to translate, 0 = skip the entry
size_t translatableLength(const char *data, size_t entryLen, size_t maxSeqLen) {
    size_t length = entryLen - 1;
    if ((data[length] != '\n' && length % 3 != 0) && (data[length - 1] == '\n' && (length - 1) % 3 != 0)) length -= length % 3;
    if (length < 3) return 0;
    if (length > 3 * maxSeqLen) length = 3 * maxSeqLen;
    return length;
}

// ReverseSeqByCodon.cpp:39-52: codons in reverse order, every codon read forward; a trailing partial codon is dropped
void reverseByCodon(const char *seq, size_t L, std::string &out) {
    out.clear();
    for (size_t done = 0; done + 3 <= L; done += 3) out.append(seq + (L - done - 3), 3);
    out.push_back('\n');
}

// ------------------------------------------------------------------------------------------------------------------
// nucleotide scores + statistics (NucleotideMatrix.cpp, SubstitutionMatrix::createAsciiSubMat)
// ------------------------------------------------------------------------------------------------------------------
bool NuclScorer::init(const std::string &matrixName, int gapOpen_, int gapExtend_, uint64_t dbResidues, std::string &err) {
    std::string base = matrixName; const size_t slash = base.find_last_of("/\\"); if (slash != std::string::npos) base = base.substr(slash + 1);
    const size_t colon = base.find(".out:"); if (colon != std::string::npos) base = base.substr(0, colon + 4);
    if (base != "nucleotide.out") { err = "proteinaln2nucl: only the embedded nucleotide.out matrix is supported, got " + matrixName; return false; }
    // lib/mmseqs/data/nucleotide.out: letters A C T G X, background 0.2499975 x 4 + 0.00001, lambda 0.6337314, match 2, anything else -3
    const int n = 5; const double lambda = 0.6337314;
    double pBack[n] = {0.2499975, 0.2499975, 0.2499975, 0.2499975, 0.00001}, raw[n][n], prob[n][n];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) raw[i][j] = (i == j && i < 4) ? 2.0 : -3.0;
    for (int i = 0; i < n - 1; i++) pBack[i] = pBack[i] * (1.0 - pBack[n - 1]);                       // SubstitutionMatrix.cpp:395-399 (no positive X score)
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) prob[i][j] = std::exp(lambda * raw[i][j]) * pBack[i] * pBack[j];
    double pb[n];                                                                                   // BaseMatrix::generateSubMatrix, bit factor 1.0, bias 0.0
    for (int i = 0; i < n; i++) { pb[i] = 0; for (int j = 0; j < n; j++) pb[i] += prob[i][j]; }
    pb[n - 1] = 1E-5;
    int sub[n][n];
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { const double v = 1.0 * std::log2(prob[i][j] / (pb[i] * pb[j])) + 0.0; sub[i][j] = (short) ((v < 0.0) ? v - 0.5 : v + 0.5); }
    // NucleotideMatrix::setupLetterMapping (NucleotideMatrix.cpp:18-63): classes A=0 C=1 T=2 G=3 X=4
    for (int c = 0; c < 256; c++) {
        int cls;
        switch (c < 128 ? toupper(c) : c) {
            case 'A': cls = 0; break; case 'C': case 'M': case 'Y': case 'H': cls = 1; break;
            case 'T': case 'U': case 'W': cls = 2; break;
            case 'G': case 'K': case 'B': case 'D': case 'V': case 'R': case 'S': cls = 3; break;
            default: cls = 4; break;
        }
        cls_[c] = (unsigned char) cls;
    }
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) score_[i][j] = sub[i][j];
    gapOpen = gapOpen_; gapExtend = gapExtend_;
    return ev.init("nucleotide.out", gapOpen_, gapExtend_, dbResidues, err);
}

// ------------------------------------------------------------------------------------------------------------------
// f1  proteinaln2nucl: one alignment line of a protein search lifted to the nucleotide ORFs (proteinaln2nucl.cpp:99-187)
// ------------------------------------------------------------------------------------------------------------------
bool proteinLineToNucl(const char *line, const NuclScorer &sc, const char *qNuc, uint32_t qNucLen, bool qStartCodon, const char *tNuc, uint32_t tNucLen, bool tStartCodon,
                       std::string &out, std::string &err) {
    const char *w[16]; const size_t cols = words(line, w, 16);
    if (cols < 11) { err = "proteinaln2nucl: this module needs alignment records with a backtrace"; return false; }
    sp_alignment r; memset(&r, 0, sizeof(r));
    r.db_key = (uint32_t) atou(w[0]);
    long qs = atoiS(w[4]), qe = atoiS(w[5]), ts = atoiS(w[7]), te = atoiS(w[8]);
    if ((qStartCodon && qs == 0) || (tStartCodon && ts == 0)) { err = "proteinaln2nucl: alignment contains an unalignable character"; return false; }
    const long lim = 1L << 40;      // (positions of a damaged record: nothing below multiplies or adds past a long -- UBSan on the fuzzed DBs of tests/test_host_fuzz.py)
    if (qs < -lim || qs > lim || qe < -lim || qe > lim || ts < -lim || ts > lim || te < -lim || te > lim) { err = "proteinaln2nucl: invalid alignment result record"; return false; }
    ts = ts * 3 + (tStartCodon ? -3 : 0); te = te * 3 + 2 + (tStartCodon ? -3 : 0);
    qs = qs * 3 + (qStartCodon ? -3 : 0); qe = qe * 3 + 2 + (qStartCodon ? -3 : 0);
    size_t ids = 0, alnLen = 0; long qPos = qs, tPos = ts; int score = 0;
    std::string bt; char nb[16];
    for (const char *p = w[10]; *p != '\n' && *p != '\0' && !isWs(*p); p++) {
        int cnt = 0;
        if (*p >= '0' && *p <= '9') { const unsigned long c = atou(p); if (c > (1ul << 28)) { err = "proteinaln2nucl: invalid alignment result record"; return false; } cnt = (int) c; while (*p >= '0' && *p <= '9') p++; }
        const char op = *p;
        if (op == 'M') {
            if (qPos < 0 || tPos < 0 || qPos + (long) cnt * 3 > (long) qNucLen || tPos + (long) cnt * 3 > (long) tNucLen) { err = "proteinaln2nucl: an alignment runs past its nucleotide ORF (peptide and nucleotide DBs disagree)"; return false; }
            for (int b = 0; b < cnt * 3; b++) { const unsigned char a = (unsigned char) qNuc[qPos], c = (unsigned char) tNuc[tPos]; ids += (a == c); score += sc.score(a, c); qPos++; tPos++; }
        } else if (op == 'D') { tPos += cnt * 3; score -= sc.gapOpen + ((cnt - 1) * 3) * sc.gapExtend; }
        else if (op == 'I') { qPos += cnt * 3; score -= sc.gapOpen + ((cnt - 1) * 3) * sc.gapExtend; }
        else { if (op == '\n' || op == '\0') break; continue; }
        alnLen += (size_t) cnt * 3;
        char *e = i32toa(cnt * 3, nb); bt.append(nb, (size_t) (e - nb)); bt.push_back(op);
    }
    r.bit_score = (int) sc.ev.bitScore(score);      // `res.score = evaluer.computeBitScore(score)`: a double assigned to an int (truncation, no rounding)
    r.evalue = sc.ev.evalue(score, qNucLen);
    r.seq_id = (float) ids / (float) alnLen;
    r.q_start = (int) qs; r.q_end = (int) qe; r.q_len = qNucLen; r.db_start = (int) ts; r.db_end = (int) te; r.db_len = tNucLen;
    appendAlnLine(out, r, "", false);
    out.pop_back(); out.push_back('\t'); out += bt; out.push_back('\n');      // resultToBuffer(..., compress = false): the run-length string as it is
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// f2  combineprotnuclaln (CombineProtNuclAln.cpp:55-97): protein lines get the nucleotide identity and the combined E-value
// ------------------------------------------------------------------------------------------------------------------
bool combineProtNucl(const char *prot, const char *nucl, std::string &out, std::string &err) {
    const char *pe[16], *ne[16];
    while (*prot != '\0') {
        if (words(prot, pe, 16) < 10) { err = "combineprotnuclaln: invalid alignment result record"; return false; }
        const char *next = skipLine(prot);
        const unsigned long target = atou(pe[0]);
        const double protEval = strtod(pe[3], NULL);
        double updated = FLT_MAX, nuclSeqId = 0;
        for (const char *cur = nucl; *cur != '\0';) {
            if (atou(cur) != target) { cur = skipLine(cur); continue; }
            if (words(cur, ne, 16) < 10) { err = "combineprotnuclaln: invalid alignment result record"; return false; }
            cur = skipLine(cur);
            nuclSeqId = strtod(ne[2], NULL);
            const double nuclEval = strtod(ne[3], NULL);
            const double mean = std::log(protEval) * 0.5 + std::log(nuclEval) * 0.5;
            updated = (mean < std::log(nuclEval)) ? std::exp(mean) : nuclEval;
        }
        out.append(pe[0], (size_t) (pe[2] - pe[0]));
        out += fmtE(nuclSeqId); out.push_back('\t');
        out += fmtE(updated); out.push_back('\t');
        out.append(pe[4], (size_t) ((next - 1) - pe[4]));
        out.push_back('\n');
        prot = next;
    }
    return true;
}

// Aggregation::buildMap (Aggregation.cpp:24-45): lines grouped by the set of their first column, sets ascending, lines in order
// the non-empty lines of an entry with the set of their first column, ordered by set (lines of one set keep their order): what
// the reference collects into a std::map<set, vector<columns>> (besthitperset.cpp:35-52, CombineScore.cpp:47-62) -- kept as
// views into the entry, no per-column copies
struct SetLine { const char *p; uint32_t n; uint32_t set; };
static bool groupBySet(const char *data, const SetLookup &memberToSet, std::vector<SetLine> &lines, std::string &err) {
    lines.clear();
    bool ordered = true;
    while (*data != '\0') {
        const char *next = skipLine(data);
        size_t n = (size_t) (next - data); if (n && data[n - 1] == '\n') n--;
        if (n) {
            uint32_t set;
            if (!memberToSet.find((uint32_t) atou(data), set)) { size_t k = 0; while (k < n && data[k] != '\t') k++; err = "invalid target database key " + std::string(data, k); return false; }
            if (!lines.empty() && lines.back().set > set) ordered = false;
            lines.push_back(SetLine{data, (uint32_t) n, set});
        }
        data = next;
    }
    if (!ordered) std::stable_sort(lines.begin(), lines.end(), [](const SetLine &a, const SetLine &b) { return a.set < b.set; });
    return true;
}
// column i of a tab-separated line ("" when the line has fewer columns); the text ends at a tab / the line end, which stops strtod
static const char *column(const SetLine &l, size_t i) {
    const char *p = l.p, *e = l.p + l.n;
    for (; i > 0; i--) { while (p < e && *p != '\t') p++; if (p == e) return ""; p++; }
    return p;
}
static double columnValue(const SetLine &l, size_t i) {      // strtod must not run past the line into the next one when the column is the (empty) last one
    const char *c = column(l, i);
    return (c == l.p + l.n) ? 0.0 : strtod(c, NULL);
}

// besthitperset (besthitperset.cpp:29-115): per target set the line with the best E-value; column 1 becomes the log of the
// set-corrected P-value.  simple = --simple-best-hit (predictmatch: 1)
bool bestHitPerSet(const char *entry, const SetLookup &memberToSet, const SetLookup &setSize, bool simple, std::string &out, std::string &err) {
    std::vector<SetLine> lines; char nb[40];
    if (!groupBySet(entry, memberToSet, lines, err)) return false;
    for (size_t from = 0, to; from < lines.size(); from = to) {
        for (to = from + 1; to < lines.size() && lines[to].set == lines[from].set; to++) {}
        uint32_t nGenes;
        if (!setSize.find(lines[from].set, nGenes)) { err = "invalid target size database key " + std::to_string(lines[from].set); return false; }
        double bestScore = -DBL_MAX, second = -DBL_MAX, bestEval = DBL_MAX; const SetLine *best = nullptr;
        const bool single = simple || to - from < 2;
        for (size_t k = from; k < to; k++) {
            const SetLine &l = lines[k];
            const double eval = columnValue(l, 3);
            double pval = eval / nGenes; if (pval == 0) pval = DBL_MIN;
            const double score = -std::log(pval);
            if (single) { if (bestEval > eval) { bestEval = eval; best = &l; } }
            else if (score >= bestScore) { second = bestScore; bestScore = score; best = &l; }
            else if (score > second) second = score;
        }
        double logP;
        if (single) {
            if (bestEval == 0) logP = std::log(DBL_MIN);
            else if (bestEval > 0 && bestEval < 10e-4) logP = std::log(bestEval);
            else logP = std::log(1 - std::exp(-bestEval));
        } else logP = second - bestScore;
        if (best) {      // the line with column 1 replaced
            const char *b = best->p, *e = best->p + best->n, *c1 = b;
            while (c1 < e && *c1 != '\t') c1++;
            out.append(b, (size_t) (c1 - b));
            if (c1 < e) {
                out.push_back('\t'); out.append(nb, (size_t) (fmtE3(logP, nb) - nb));
                const char *c2 = c1 + 1; while (c2 < e && *c2 != '\t') c2++;
                out.append(c2, (size_t) (e - c2));
            }
        }
        out.push_back('\n');
    }
    return true;
}

// combinescore (CombineScore.cpp:41-98): per target set the truncated-product style score over a query set's best hits
bool combineScore(const char *entry, uint32_t querySetSize, const SetLookup &memberToSet, std::string &out, std::string &err) {
    std::vector<SetLine> lines;
    if (!groupBySet(entry, memberToSet, lines, err)) return false;
    const double logThr = std::log(1.0 / (querySetSize + 1));
    char nb[40];
    for (size_t from = 0, to; from < lines.size(); from = to) {
        for (to = from + 1; to < lines.size() && lines[to].set == lines[from].set; to++) {}
        double minLogP = 0, sumLogP = 0, minSeqId = 0, sumSeqId = 0; size_t k = 0;
        for (size_t i = from; i < to; i++) {
            const double logP = columnValue(lines[i], 1), seqId = columnValue(lines[i], 2);
            if (logP < minLogP) { minLogP = (logP == 0) ? logP : -logP; minSeqId = seqId; }
            if (logP < logThr) { sumLogP -= logP - logThr; sumSeqId += seqId; k++; }
        }
        char *e = u32toa(lines[from].set, nb); out.append(nb, (size_t) (e - nb)); out.push_back('\t');
        const double a = k == 0 ? minLogP : sumLogP - logThr, b = k == 0 ? minSeqId : sumSeqId / k;
        out.append(nb, (size_t) (fmtE3(a, nb) - nb)); out.push_back('\t'); out.append(nb, (size_t) (fmtE3(b, nb) - nb));
        out.push_back('\n');
    }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// f4  filtermatchbyfdr (FilterMatchbyFdr.cpp:49-215): score threshold at a false-discovery rate from target and control scores
// ------------------------------------------------------------------------------------------------------------------
// The step in four pieces.  The floating-point expressions keep the reference's operand order (they decide which score
// becomes the threshold); the cutoff is a FLOAT there (LocalParameters.h:57), so 0.05 is compared as 0.0500000007...
namespace {
struct RocCurve { std::vector<double> fp, tp; };      // per distinct target score, best first: control fraction / target fraction above it

// (:95-118) one point per distinct target score s: tp = #target scores > s / #target, fp = (#control scores > s + 0.5) / (#control + 1); closes at (1, 1)
RocCurve rocCurve(const std::vector<double> &target, const std::vector<double> &control, std::vector<double> &distinct) {
    RocCurve c; size_t above = 0; double last = DBL_MAX;
    for (size_t k = 0; k < target.size(); k++) {
        if (!(target[k] < last)) continue;
        last = target[k];
        while (above < control.size() && target[k] < control[above]) above++;
        distinct.push_back(target[k]);
        c.tp.push_back(1.0 * k / target.size());
        c.fp.push_back(1.0 * (above + 0.5) / (control.size() + 1));
    }
    c.fp.push_back(1.0); c.tp.push_back(1.0);
    return c;
}

// (:120-138) greedy upper hull: from vertex v the next one is the LAST point of the steepest slope seen (ties move on: >=)
void upperHull(const RocCurve &c, std::vector<size_t> &vertex, std::vector<double> &steep) {
    for (size_t v = 0; v + 1 < c.fp.size();) {
        size_t nxt = v + 1; double top = 0;
        for (size_t w = v + 1; w < c.fp.size(); w++) { const double sl = 1.0 * (c.tp[w] - c.tp[v]) / (c.fp[w] - c.fp[v]); if (sl >= top) { top = sl; nxt = w; } }
        vertex.push_back(nxt); steep.push_back(top); v = nxt;
    }
}
}  // namespace

void fdrThreshold(std::vector<double> P, std::vector<double> N, bool haveControl, double cutoffArg, bool reportFdr, FdrResult &r) {
    const double cutoff = (double) (float) cutoffArg;
    std::sort(P.begin(), P.end(), std::greater<double>());
    if (P.empty()) P.push_back(0);
    r = FdrResult();
    r.threshold = P.back();
    if (!haveControl) {
        // (:201-212) every match is printed; with --report-fdr the table lists each score above DBL_MIN with FDR 0 (the reference declares
        // its running score inside the loop, so nothing is merged); scores at or below DBL_MIN are looked up past the table there
        if (reportFdr) for (double p : P) if (DBL_MIN < p) { r.uniqueScores.push_back(p); r.fdr.push_back(0.0); }
        return;
    }
    std::sort(N.begin(), N.end(), std::greater<double>());
    const RocCurve c = rocCurve(P, N, r.uniqueScores);
    std::vector<size_t> vertex; std::vector<double> steep;
    upperHull(c, vertex, steep);
    // (:142-144) null proportion: mean of the last two hull slopes.  A hull of ONE segment (one distinct target score) makes the reference read in
    // front of its vector; that segment's slope stands in twice here
    const double pi0 = ((steep.size() >= 2 ? steep[steep.size() - 2] : steep.back()) + steep.back()) / 2;
    // (:146-155) first hull vertex whose estimated FDR exceeds the cutoff -- searched only when the LAST vertex reaches it
    size_t seg = 0;
    if (c.fp[vertex.back()] * pi0 / c.tp[vertex.back()] >= cutoff) { double est = 0; while (est <= cutoff) { est = c.fp[vertex[seg]] * pi0 / c.tp[vertex[seg]]; seg++; } }
    if (seg >= 2) {
        // (:166-180) walk the curve from the vertex before, interpolating along the hull segment, until the FDR passes the cutoff
        size_t at = vertex[seg - 2];
        double all = c.tp[at], null = c.fp[at] * pi0, est = 0;
        while (est <= cutoff) { at++; const double step = c.fp[at] - c.fp[at - 1]; all += step * steep[seg - 1]; null += step * pi0; est = null / all; }
        r.threshold = r.uniqueScores[at];
    }
    if (!reportFdr) return;
    // (:184-200) FDR of every distinct score: 0 in front of the first vertex, then accumulated segment by segment
    r.fdr.assign(vertex[0], 0.0);
    for (size_t h = 0; h + 1 < vertex.size(); h++) {
        const size_t from = vertex[h], to = vertex[h + 1];
        const bool vertical = std::isinf(steep[h]);
        double all = c.tp[from], null = c.fp[from] * pi0;
        for (size_t at = from; at < to; at++) {
            if (vertical) { r.fdr.push_back(c.fp[from] * pi0 / c.tp[from]); continue; }
            const double step = c.fp[at] - c.fp[at - 1];
            all += step * steep[h]; null += step * pi0;
            r.fdr.push_back(null / all);
        }
    }
}

static bool collectScores(const Db &db, std::vector<double> &v, std::string &err) {
    const char *w[8];
    for (size_t i = 0; i < db.entries.size(); i++)
        for (const char *p = db.ptr(i); *p != '\0'; p = skipLine(p)) {
            if (words(p, w, 8) < 2) { err = "invalid result record"; return false; }
            v.push_back(strtod(w[1], NULL));
        }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// the chain after the search on records in memory
// ------------------------------------------------------------------------------------------------------------------
bool setScores(const SetInputs &in, int stage, std::vector<uint32_t> &keys, std::vector<std::string> &entries, std::string &err) {
    keys.clear(); entries.clear();
    if (stage < 0 || stage > 5) { err = "set scores: unknown stage"; return false; }
    const size_t nq = in.nq, nt = in.nt;
    NuclScorer sc;
    if (stage != 1 && !sc.init(in.nuclMatrix, in.gapOpen, in.gapExtend, nt ? in.tNuclOff[nt] : 0, err)) return false;
    // set tables from the per-sequence set keys: member -> set, set -> number of members (what createsetdb's result2stats counts)
    SetLookup tMemberToSet, tSetSize;
    {   std::vector<uint32_t> k(nt), v(nt); std::map<uint32_t, uint32_t> cnt;
        for (size_t t = 0; t < nt; t++) { k[t] = in.tKeys ? in.tKeys[t] : (uint32_t) t; v[t] = in.tSet[t]; cnt[in.tSet[t]]++; }
        tMemberToSet.fromArrays(k, v);
        std::vector<uint32_t> sk, sv; for (auto &c : cnt) { sk.push_back(c.first); sv.push_back(c.second); }
        tSetSize.fromArrays(sk, sv); }
    std::vector<std::string> perQuery(nq);
    std::vector<std::string> errs((size_t) std::max(1, in.threads) + 64);
    parallelChunks(nq, in.threads, [&](size_t ci, size_t from, size_t to) {
        std::vector<std::pair<sp_alignment, const char *>> ok; std::string prot, nucl, comb; std::string &e = errs[ci];
        for (size_t q = from; q < to && e.empty(); q++) {
            ok.clear();
            for (uint32_t i = in.hitOffsets[q]; i < in.hitOffsets[q + 1]; i++) { const sp_alignment &r = in.aln[i]; if (r.passed && r.seq_id >= 0.0) ok.push_back({r, in.cigars + r.cigar_offset}); }
            std::sort(ok.begin(), ok.end(), [](const std::pair<sp_alignment, const char *> &a, const std::pair<sp_alignment, const char *> &b) { return alnLess(a.first, b.first); });
            prot.clear();
            for (auto &x : ok) appendAlnLine(prot, x.first, x.second, true);
            if (stage == 1) { perQuery[q] = prot; continue; }
            // the records carry target KEYS; the nucleotide ORFs are addressed by target index: look the index up among this query's hits
            nucl.clear();
            const char *p = prot.c_str();
            for (auto &x : ok) {
                uint32_t t = 0xFFFFFFFFu;
                for (uint32_t i = in.hitOffsets[q]; i < in.hitOffsets[q + 1]; i++) if ((in.tKeys ? in.tKeys[in.hitTarget[i]] : in.hitTarget[i]) == x.first.db_key) { t = in.hitTarget[i]; break; }
                if (t == 0xFFFFFFFFu) { e = "set scores: alignment record without a hit"; break; }
                if (!proteinLineToNucl(p, sc, in.qNucl + in.qNuclOff[q], (uint32_t) (in.qNuclOff[q + 1] - in.qNuclOff[q]), in.qFirst && in.qFirst[q] == '*',
                                       in.tNucl + in.tNuclOff[t], (uint32_t) (in.tNuclOff[t + 1] - in.tNuclOff[t]), in.tFirst && in.tFirst[t] == '*', nucl, e)) break;
                p = skipLine(p);
            }
            if (!e.empty()) break;
            if (stage == 2) { perQuery[q] = nucl; continue; }
            comb.clear();
            if (!combineProtNucl(prot.c_str(), nucl.c_str(), comb, e)) break;
            if (stage == 3) { perQuery[q] = comb; continue; }
            if (!bestHitPerSet(comb.c_str(), tMemberToSet, tSetSize, in.simpleBestHit, perQuery[q], e)) break;
        }
    });
    for (const std::string &e : errs) if (!e.empty()) { err = e; return false; }
    if (stage >= 1 && stage <= 4) {
        keys.resize(nq); entries.swap(perQuery);
        for (size_t q = 0; q < nq; q++) keys[q] = in.qKeys ? in.qKeys[q] : (uint32_t) q;
        return true;
    }
    // mergeresultsbyset + combinescore: query sets ascending, members in query order
    std::vector<std::pair<uint32_t, uint32_t>> bySet(nq);      // (set, query), sorted: the members of a set are consecutive and in query order
    for (size_t q = 0; q < nq; q++) bySet[q] = {in.qSet[q], (uint32_t) q};
    if (!std::is_sorted(bySet.begin(), bySet.end())) std::sort(bySet.begin(), bySet.end());
    std::vector<size_t> first;      // first[s] = where set number s starts in bySet
    for (size_t i = 0; i < nq; i++) if (i == 0 || bySet[i].first != bySet[i - 1].first) first.push_back(i);
    first.push_back(nq);
    const size_t ns = first.size() - 1;
    keys.resize(ns); entries.resize(ns);
    const size_t T = (size_t) std::max(1, std::min<int>(in.threads, (int) (ns / 16 + 1)));
    std::vector<std::string> setErrs(T);
    std::atomic<size_t> nextSet{0};
    auto work = [&](size_t t) {      // sets are handed out in runs of 16: a set's cost is the number of best-hit lines of its members
        std::string merged;
        for (;;) {
            const size_t from = nextSet.fetch_add(16), to = std::min(ns, from + 16);
            if (from >= ns || !setErrs[t].empty()) break;
            for (size_t s = from; s < to && setErrs[t].empty(); s++) {
                merged.clear();
                for (size_t i = first[s]; i < first[s + 1]; i++) merged += perQuery[bySet[i].second];
                keys[s] = bySet[first[s]].first;
                if (stage == 5) { entries[s] = merged; continue; }
                combineScore(merged.c_str(), (uint32_t) (first[s + 1] - first[s]), tMemberToSet, entries[s], setErrs[t]);
            }
        }
    };
    {   std::vector<std::thread> pool; for (size_t t = 1; t < T; t++) pool.emplace_back(work, t); work(0); for (auto &th : pool) th.join(); }
    for (const std::string &e : setErrs) if (!e.empty()) { err = e; return false; }
    return true;
}

// ------------------------------------------------------------------------------------------------------------------
// module drivers
// ------------------------------------------------------------------------------------------------------------------
static const std::set<std::string> kCommon = {"--threads", "--compressed", "-v"};
static std::set<std::string> with(std::initializer_list<const char *> more) { std::set<std::string> s = kCommon; for (const char *m : more) s.insert(m); return s; }

static bool loadDb(const std::string &path, Db &db, std::string &err) {
    if (!db.load(path, err)) return false;
    if (db.compressed()) { err = path + ": compressed DBs are not supported (the workflows pass --compressed 0)"; return false; }
    return true;
}
static bool uncompressedOnly(const Flags &f, std::string &err) { return requireValue(f, "--compressed", {"0"}, err); }

bool SetLookup::fromDb(const Db &db) {
    std::vector<uint32_t> keys(db.entries.size()); values.resize(db.entries.size());
    for (size_t i = 0; i < db.entries.size(); i++) { keys[i] = db.entries[i].key; values[i] = (uint32_t) atou(db.ptr(i)); }
    index.build(keys);
    return true;
}
bool SetLookup::find(uint32_t key, uint32_t &v) const { uint32_t pos; if (!index.find(key, pos)) return false; v = values[pos]; return true; }
void SetLookup::fromArrays(const std::vector<uint32_t> &keys, const std::vector<uint32_t> &vals) { values = vals; index.build(keys); }

// entries of `db` in ascending key order (the order DBReader::getId / NOSORT readers present them in)
static std::vector<size_t> byKey(const Db &db) {
    std::vector<size_t> o(db.entries.size());
    for (size_t i = 0; i < o.size(); i++) o[i] = i;
    std::sort(o.begin(), o.end(), [&](size_t a, size_t b) { return db.entries[a].key < db.entries[b].key; });
    return o;
}

// ------------------------------------------------------------------------------------------------------------------
// entry-parallel drivers: the input entries [0, n) are cut into chunks of about equal input bytes, worker threads take the chunks
// in turn and format their output entries into one block per chunk, the blocks go to the writer in chunk order (so the data
// file has the layout a single thread would have written)
// ------------------------------------------------------------------------------------------------------------------
struct ChunkOut {
    std::string block; std::vector<uint32_t> keys; std::vector<uint64_t> lengths; std::string err;
    void add(uint32_t key, const char *p, size_t n) { block.append(p, n); block.push_back('\0'); keys.push_back(key); lengths.push_back(n + 1); }
    void add(uint32_t key, const std::string &e) { add(key, e.data(), e.size()); }
};
static int threadsOf(const Flags &f) { return (int) std::max(1L, std::min(256L, f.num("--threads", defaultThreads()))); }
template <class Out, class W, class F>
static bool forChunks(size_t n, W weight, int threads, std::vector<Out> &outs, F fn, std::string &err) {
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += weight(i);
    const size_t T = (size_t) std::max(1, threads);
    const uint64_t per = std::max<uint64_t>(1u << 18, total / (T * 4) + 1);
    std::vector<size_t> cut{0}; uint64_t acc = 0;
    for (size_t i = 0; i < n; i++) { acc += weight(i); if (acc >= per) { cut.push_back(i + 1); acc = 0; } }
    if (cut.back() != n) cut.push_back(n);
    const size_t C = cut.size() - 1;
    outs.clear(); outs.resize(C);
    std::atomic<size_t> next{0};
    // (a chunk's output is built in a LOCAL of the worker and moved into its slot at the end: the headers of neighbouring strings / vectors in
    // `outs` share cache lines, and every append of one thread would pull the line out of its neighbour's cache)
    auto work = [&]() { for (;;) { const size_t c = next.fetch_add(1); if (c >= C) break; Out o = std::move(outs[c]); for (size_t i = cut[c]; i < cut[c + 1] && o.err.empty(); i++) fn(i, o); outs[c] = std::move(o); } };
    if (C <= 1 || T == 1) work();
    else { std::vector<std::thread> pool; for (size_t t = 1; t < std::min(T, C); t++) pool.emplace_back(work); work(); for (auto &th : pool) th.join(); }
    for (const Out &o : outs) if (!o.err.empty()) { err = o.err; return false; }
    return true;
}
static void flushChunks(std::vector<ChunkOut> &outs, DbWriter &w) { for (ChunkOut &o : outs) if (!o.keys.empty()) w.addBlock(o.keys, o.lengths, std::move(o.block)); }

// DBReader::softlinkDb(in, out, flags) for the files next to a sequence DB (commons/DBReader.cpp:1093-1163): every listed file that exists
// is linked under the output's name by its real path, replacing what is there (FileUtil::symlinkAbs, FileUtil.cpp:242-278).
// SEQUENCE_ANCILLARY = headers, lookup, source and the taxonomy files (DBReader.h:43-47).
static const char *const kSourceOnly[] = {".source", NULL};
static const char *const kSequenceAncillary[] = {"_h", "_h.index", "_h.dbtype", ".lookup", ".source", "_mapping", "_names.dmp", "_nodes.dmp", "_merged.dmp", NULL};
static bool linkAncillary(const std::string &in, const std::string &out, const char *const *suffixes, std::string &err) {
    for (const char *const *sfx = suffixes; *sfx; sfx++) {
        const std::string from = in + *sfx, to = out + *sfx;
        struct stat st;
        if (stat(from.c_str(), &st) != 0) continue;
        char *real = realpath(from.c_str(), NULL);
        if (!real) { err = "Could not get realpath of " + from; return false; }
        std::string dir = to, base = to;
        const size_t slash = to.rfind('/');
        if (slash == std::string::npos) dir = "."; else { dir = to.substr(0, slash ? slash : 1); base = to.substr(slash + 1); }
        char *realDir = realpath(dir.c_str(), NULL);
        if (!realDir) { free(real); err = "Could not get realpath of " + to; return false; }
        const std::string link = std::string(realDir) + "/" + base;
        free(realDir);
        if (link == real) { free(real); continue; }      // in == out
        (void) unlink(link.c_str());
        const int rc = symlink(real, link.c_str());
        free(real);
        if (rc != 0) { err = "Could not create symlink of " + from; return false; }
    }
    return true;
}

int runExtractOrfs(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--min-length", "--max-length", "--max-gaps", "--contig-start-mode", "--contig-end-mode", "--orf-start-mode", "--forward-frames",
                                      "--reverse-frames", "--translation-table", "--translate", "--use-all-table-starts", "--id-offset", "--create-lookup"}), 2, f, err))
        return fail("extractorfs: " + err);
    if (!uncompressedOnly(f, err) || !requireValue(f, "--translation-table", {"1"}, err) || !requireValue(f, "--translate", {"0"}, err) ||
        !requireValue(f, "--use-all-table-starts", {"0"}, err) || !requireValue(f, "--id-offset", {"0"}, err))
        return fail("extractorfs: " + err);
    const size_t minLen = (size_t) f.num("--min-length", 30), maxLen = (size_t) f.num("--max-length", 32734), maxGaps = (size_t) f.num("--max-gaps", INT_MAX);
    const int contigStart = (int) f.num("--contig-start-mode", 2), contigEnd = (int) f.num("--contig-end-mode", 2);
    const unsigned startMode = (unsigned) f.num("--orf-start-mode", 1);
    if (startMode == 1 && contigStart < 2) return fail("extractorfs: orf-start-mode 1 can only go with contig-start-mode 2");
    auto frames = [&](const char *flag) { unsigned r = 0; const std::string v = f.str(flag, "1,2,3"); for (int k = 1; k <= 3; k++) if (("," + v + ",").find("," + std::to_string(k) + ",") != std::string::npos) r |= 1u << (k - 1); return r; };
    const unsigned fwdFrames = frames("--forward-frames"), revFrames = frames("--reverse-frames");
    const bool tm = getenv("SP_MODULE_TIMING") != nullptr; auto t0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (tm) { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "extractorfs %-10s %.1f ms\n", what, std::chrono::duration<double, std::milli>(t1 - t0).count()); t0 = t1; } };
    Db in;
    if (!loadDb(f.positional[0], in, err)) return fail("extractorfs: " + err);
    lap("load");
    // lookup of the contigs (key, name, set) for --create-lookup 1
    // (contig key -> the text of its set column), sorted by key
    std::string lookupFile; std::vector<std::pair<uint32_t, std::pair<uint64_t, uint32_t>>> setOf;
    const bool createLookup = f.num("--create-lookup", 0) != 0;
    if (createLookup) {
        MappedFile lk;
        if (!lk.open(f.positional[0] + ".lookup")) return fail("extractorfs: cannot open " + f.positional[0] + ".lookup");
        lookupFile.assign(lk.p, lk.n);
        const char *b = lookupFile.data(), *e = b + lookupFile.size();
        for (const char *p = b; p < e;) {
            const char *eol = (const char *) memchr(p, '\n', (size_t) (e - p)); if (!eol) eol = e;
            const char *first = (const char *) memchr(p, '\t', (size_t) (eol - p)); const char *lastTab = first;
            if (first) for (const char *q = eol; q > first; q--) if (q[-1] == '\t') { lastTab = q - 1; break; }
            if (first && lastTab != first) setOf.push_back({(uint32_t) atou(p), {(uint64_t) (lastTab + 1 - b), (uint32_t) (eol - lastTab - 1)}});
            p = eol + 1;
        }
        bool sorted = true;
        for (size_t i = 1; i < setOf.size() && sorted; i++) sorted = setOf[i - 1].first < setOf[i].first;
        if (!sorted) std::stable_sort(setOf.begin(), setOf.end(), [](const auto &a, const auto &b) { return a.first < b.first; });      // a key listed twice: the last line wins, as with the map the reference fills
    }
    auto setText = [&](uint32_t key, std::string &to) {
        auto it = std::upper_bound(setOf.begin(), setOf.end(), key, [](uint32_t k, const auto &x) { return k < x.first; });
        if (it == setOf.begin() || (it - 1)->first != key) { to.push_back('0'); return; }
        to.append(lookupFile.data() + (it - 1)->second.first, (it - 1)->second.second);
    };
    lap("lookup");
    // contigs in key order; ORFs are numbered 0, 1, ... in (contig key, discovery) order (DBWriter::createRenumberedDB, DBWriter.cpp:681-739)
    DbWriter seqW, hdrW; seqW.path = f.positional[1]; seqW.dbtype = 1; hdrW.path = f.positional[1] + "_h"; hdrW.dbtype = 12;
    struct OrfChunk { ChunkOut seq, hdr; std::vector<uint32_t> contig; std::string err, lookup; OrfFinder orf; std::vector<OrfLoc> locs; };
    const std::vector<size_t> order = byKey(in);
    std::vector<OrfChunk> chunks;
    if (!forChunks(order.size(), [&](size_t k) { return in.entries[order[k]].length; }, threadsOf(f), chunks, [&](size_t k, OrfChunk &o) {
        const size_t i = order[k]; const uint32_t key = in.entries[i].key; const size_t L = (size_t) in.seqLen(i);
        if (L < 3) return;
        o.orf.setSequence(in.ptr(i), L);
        o.locs.clear();
        o.orf.findAll(minLen, maxLen, maxGaps, fwdFrames, revFrames, startMode, o.locs);
        for (const OrfLoc &loc : o.locs) {
            if (contigStart < 2 && (int) loc.incStart == contigStart) continue;
            if (contigEnd < 2 && (int) loc.incEnd == contigEnd) continue;
            size_t from = loc.from, to = loc.to;
            if (loc.strand < 0) { from = (L - 1) - loc.from; to = (L - 1) - loc.to; }
            const size_t n = loc.to - loc.from + 1;
            o.seq.block.append((loc.strand > 0 ? o.orf.fwd : o.orf.rev).data() + loc.from, n); o.seq.block.push_back('\n'); o.seq.block.push_back('\0'); o.seq.lengths.push_back(n + 2);
            o.hdr.add(0, orfHeader(key, from, to, loc.incStart, loc.incEnd));
            o.contig.push_back(key);
        }
    }, err)) return fail("extractorfs: " + err);
    lap("scan");
    // ORF ids: chunk bases first, then keys and lookup lines chunk by chunk in parallel
    std::vector<uint32_t> base(chunks.size() + 1, 0);
    for (size_t c = 0; c < chunks.size(); c++) base[c + 1] = base[c] + (uint32_t) chunks[c].contig.size();
    {   std::atomic<size_t> nextChunk{0};
        auto work = [&]() { char nb[16]; for (;;) { const size_t c = nextChunk.fetch_add(1); if (c >= chunks.size()) break; OrfChunk &o = chunks[c]; const size_t n = o.contig.size();
            o.seq.keys.resize(n); for (size_t k = 0; k < n; k++) o.seq.keys[k] = base[c] + (uint32_t) k;
            o.hdr.keys = o.seq.keys;
            if (createLookup) { size_t setAt = 0, setLen = 0; for (size_t k = 0; k < n; k++) {
                char *e = u32toa(base[c] + (uint32_t) k, nb); o.lookup.append(nb, (size_t) (e - nb)); o.lookup.push_back('\t'); e = u32toa(o.contig[k], nb); o.lookup.append(nb, (size_t) (e - nb)); o.lookup.push_back('\t');
                if (k == 0 || o.contig[k] != o.contig[k - 1]) { setAt = o.lookup.size(); setText(o.contig[k], o.lookup); setLen = o.lookup.size() - setAt; }
                else { o.lookup.reserve(o.lookup.size() + setLen + 1); o.lookup.append(o.lookup.data() + setAt, setLen); }
                o.lookup.push_back('\n'); } } } };
        std::vector<std::thread> pool; for (int t = 1; t < std::min<int>(threadsOf(f), (int) chunks.size()); t++) pool.emplace_back(work); work(); for (auto &th : pool) th.join(); }
    std::string lookup;
    for (OrfChunk &o : chunks) {
        if (o.contig.empty()) continue;
        lookup += o.lookup;
        seqW.addBlock(o.seq.keys, o.seq.lengths, std::move(o.seq.block));
        hdrW.addBlock(o.hdr.keys, o.hdr.lengths, std::move(o.hdr.block));
    }
    lap("number");
    if (createLookup) { std::ofstream o(f.positional[1] + ".lookup", std::ios::binary); o.write(lookup.data(), (std::streamsize) lookup.size()); if (!o.good()) return fail("extractorfs: cannot write the lookup file"); }
    if (!linkAncillary(f.positional[0], f.positional[1], kSourceOnly, err)) return fail("extractorfs: " + err);      // softlinkDb(db1, db2, SOURCE), extractorfs.cpp:156
    {   std::string hdrErr; bool hdrOk = true;
        std::thread hdrThread([&]() { hdrOk = hdrW.close(hdrErr); });
        const bool seqOk = seqW.close(err);
        hdrThread.join();
        if (!hdrOk) return fail("extractorfs: " + hdrErr);
        if (!seqOk) return fail("extractorfs: " + err); }
    lap("write");
    return 0;
}

int runTranslateNucs(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--translation-table", "--add-orf-stop"}), 2, f, err)) return fail("translatenucs: " + err);
    if (!uncompressedOnly(f, err) || !requireValue(f, "--translation-table", {"1"}, err) || !requireValue(f, "--add-orf-stop", {"0"}, err)) return fail("translatenucs: " + err);
    Db in;
    if (!loadDb(f.positional[0], in, err)) return fail("translatenucs: " + err);
    DbWriter w; w.path = f.positional[1]; w.dbtype = 0;
    std::vector<ChunkOut> outs;
    if (!forChunks(in.entries.size(), [&](size_t i) { return in.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        const char *data = in.ptr(i);
        if (*data == '\0') return;
        const size_t length = translatableLength(data, (size_t) in.entries[i].length, 65535);
        if (!length) return;
        const size_t at = o.block.size(), n = length / 3 + 1;
        o.block.append(n, '\n'); o.block.push_back('\0');
        translate(data, length, &o.block[at]);
        o.keys.push_back(in.entries[i].key); o.lengths.push_back(n + 1);
    }, err)) return fail("translatenucs: " + err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("translatenucs: " + err);
    if (!linkAncillary(f.positional[0], f.positional[1], kSequenceAncillary, err)) return fail("translatenucs: " + err);      // softlinkDb(db1, db2, SEQUENCE_ANCILLARY), translatenucs.cpp:103
    return 0;
}

int runReverseSeqByCodon(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kCommon, 2, f, err)) return fail("reverseseqbycodon: " + err);
    if (!uncompressedOnly(f, err)) return fail("reverseseqbycodon: " + err);
    Db in;
    if (!loadDb(f.positional[0], in, err)) return fail("reverseseqbycodon: " + err);
    DbWriter w; w.path = f.positional[1]; w.dbtype = in.dbtype;
    std::vector<ChunkOut> outs;
    if (!forChunks(in.entries.size(), [&](size_t i) { return in.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        const char *seq = in.ptr(i); const size_t L = (size_t) in.seqLen(i), at = o.block.size();
        for (size_t done = 0; done + 3 <= L; done += 3) o.block.append(seq + (L - done - 3), 3);
        o.block.push_back('\n'); o.block.push_back('\0');
        o.keys.push_back(in.entries[i].key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail("reverseseqbycodon: " + err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("reverseseqbycodon: " + err);
    if (!linkAncillary(f.positional[0], f.positional[1], kSequenceAncillary, err)) return fail("reverseseqbycodon: " + err);      // ReverseSeqByCodon.cpp:58
    return 0;
}

static bool keyIndex(const Db &db, KeyMap &m) { std::vector<uint32_t> k(db.entries.size()); for (size_t i = 0; i < k.size(); i++) k[i] = db.entries[i].key; m.build(k); return true; }
static uint64_t residues(const Db &db) { uint64_t r = 0; for (size_t i = 0; i < db.entries.size(); i++) r += db.seqLen(i); return r; }      // DBReader::getAminoAcidDBSize

int runProteinAln2Nucl(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--sub-mat", "--gap-open", "--gap-extend"}), 6, f, err)) return fail("proteinaln2nucl: " + err);
    if (!uncompressedOnly(f, err)) return fail("proteinaln2nucl: " + err);
    Db qNuc, tNuc, qAA, tAA, aln;
    if (!loadDb(f.positional[0], qNuc, err) || !loadDb(f.positional[1], tNuc, err) || !loadDb(f.positional[2], qAA, err) || !loadDb(f.positional[3], tAA, err) || !loadDb(f.positional[4], aln, err))
        return fail("proteinaln2nucl: " + err);
    if (qNuc.type() != 1 || tNuc.type() != 1 || qAA.type() != 0 || tAA.type() != 0) return fail("proteinaln2nucl: wrong query and target database input");
    auto nuclPart = [](const std::string &v, const char *d) { const size_t p = v.find("nucl:"); if (p == std::string::npos) return v.empty() ? std::string(d) : v; const size_t e = v.find(',', p); return v.substr(p + 5, e == std::string::npos ? std::string::npos : e - p - 5); };
    NuclScorer sc;
    if (!sc.init(nuclPart(f.str("--sub-mat", ""), "nucleotide.out"), (int) strtol(nuclPart(f.str("--gap-open", ""), "5").c_str(), NULL, 10),
                 (int) strtol(nuclPart(f.str("--gap-extend", ""), "2").c_str(), NULL, 10), residues(tNuc), err)) return fail("proteinaln2nucl: " + err);
    KeyMap qNucIdx, tNucIdx, qAAIdx, tAAIdx; keyIndex(qNuc, qNucIdx); keyIndex(tNuc, tNucIdx); keyIndex(qAA, qAAIdx); keyIndex(tAA, tAAIdx);
    DbWriter w; w.path = f.positional[5]; w.dbtype = 5;
    std::vector<ChunkOut> outs;
    if (!forChunks(aln.entries.size(), [&](size_t i) { return aln.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        const uint32_t key = aln.entries[i].key; uint32_t qn, qa;
        if (!qNucIdx.find(key, qn) || !qAAIdx.find(key, qa)) { o.err = "proteinaln2nucl: query " + std::to_string(key) + " is missing from the query DBs"; return; }
        const size_t at = o.block.size();
        for (const char *p = aln.ptr(i); *p != '\0'; p = skipLine(p)) {
            uint32_t tn, ta; const uint32_t tKey = (uint32_t) atou(p);
            if (!tNucIdx.find(tKey, tn) || !tAAIdx.find(tKey, ta)) { o.err = "proteinaln2nucl: target " + std::to_string(tKey) + " is missing from the target DBs"; return; }
            if (!proteinLineToNucl(p, sc, qNuc.ptr(qn), (uint32_t) qNuc.seqLen(qn), qAA.ptr(qa)[0] == '*', tNuc.ptr(tn), (uint32_t) tNuc.seqLen(tn), tAA.ptr(ta)[0] == '*', o.block, o.err)) return;
        }
        o.block.push_back('\0'); o.keys.push_back(key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail(err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("proteinaln2nucl: " + err);
    return 0;
}

int runCombineProtNuclAln(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kCommon, 3, f, err)) return fail("combineprotnuclaln: " + err);
    if (!uncompressedOnly(f, err)) return fail("combineprotnuclaln: " + err);
    Db prot, nucl;
    if (!loadDb(f.positional[0], prot, err) || !loadDb(f.positional[1], nucl, err)) return fail("combineprotnuclaln: " + err);
    KeyMap nIdx; keyIndex(nucl, nIdx);
    DbWriter w; w.path = f.positional[2]; w.dbtype = prot.dbtype;
    std::vector<ChunkOut> outs;
    if (!forChunks(prot.entries.size(), [&](size_t i) { return prot.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        uint32_t n;
        if (!nIdx.find(prot.entries[i].key, n)) return;      // "Missing nucleotide alignment result": the reference skips the key
        const size_t at = o.block.size();
        if (!combineProtNucl(prot.ptr(i), nucl.ptr(n), o.block, o.err)) return;
        o.block.push_back('\0'); o.keys.push_back(prot.entries[i].key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail(err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("combineprotnuclaln: " + err);
    return 0;
}

int runBestHitPerSet(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--simple-best-hit"}), 4, f, err)) return fail("besthitperset: " + err);
    if (!uncompressedOnly(f, err)) return fail("besthitperset: " + err);
    Db m2s, sizes, res;
    if (!loadDb(f.positional[1] + "_member_to_set", m2s, err) || !loadDb(f.positional[1] + "_set_size", sizes, err) || !loadDb(f.positional[2], res, err)) return fail("besthitperset: " + err);
    SetLookup memberToSet, setSize; memberToSet.fromDb(m2s); setSize.fromDb(sizes);
    const bool simple = f.num("--simple-best-hit", 1) != 0;
    DbWriter w; w.path = f.positional[3]; w.dbtype = 5;
    std::vector<ChunkOut> outs;
    if (!forChunks(res.entries.size(), [&](size_t i) { return res.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        const size_t at = o.block.size();
        if (!bestHitPerSet(res.ptr(i), memberToSet, setSize, simple, o.block, o.err)) return;
        o.block.push_back('\0'); o.keys.push_back(res.entries[i].key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail("besthitperset: " + err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("besthitperset: " + err);
    return 0;
}

int runMergeResultsBySet(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kCommon, 3, f, err)) return fail("mergeresultsbyset: " + err);
    if (!uncompressedOnly(f, err)) return fail("mergeresultsbyset: " + err);
    Db sets, res;
    if (!loadDb(f.positional[0], sets, err) || !loadDb(f.positional[1], res, err)) return fail("mergeresultsbyset: " + err);
    KeyMap rIdx; keyIndex(res, rIdx);
    DbWriter w; w.path = f.positional[2]; w.dbtype = res.dbtype;
    std::vector<ChunkOut> outs;
    if (!forChunks(sets.entries.size(), [&](size_t i) { return sets.entries[i].length * 16; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        const size_t at = o.block.size();
        for (const char *p = sets.ptr(i); *p != '\0'; p = skipLine(p)) {
            uint32_t r;
            if (!rIdx.find((uint32_t) atou(p), r)) { o.err = "mergeresultsbyset: invalid key " + std::to_string(atou(p)) + " in entry " + std::to_string(i); return; }
            o.block.append(res.ptr(r), (size_t) res.entries[r].length - 1);
        }
        o.block.push_back('\0'); o.keys.push_back(sets.entries[i].key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail(err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("mergeresultsbyset: " + err);
    return 0;
}

int runCombineScore(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, kCommon, 5, f, err)) return fail("combinescore: " + err);       // <queryDB> <targetDB> <resultDB> <outDB> <tmpDir>
    if (!uncompressedOnly(f, err)) return fail("combinescore: " + err);
    Db qSizes, m2s, res;
    if (!loadDb(f.positional[0] + "_set_size", qSizes, err) || !loadDb(f.positional[1] + "_member_to_set", m2s, err) || !loadDb(f.positional[2], res, err)) return fail("combinescore: " + err);
    SetLookup querySize, memberToSet; querySize.fromDb(qSizes); memberToSet.fromDb(m2s);
    DbWriter w; w.path = f.positional[3]; w.dbtype = 5;
    std::vector<ChunkOut> outs;
    if (!forChunks(res.entries.size(), [&](size_t i) { return res.entries[i].length; }, threadsOf(f), outs, [&](size_t i, ChunkOut &o) {
        uint32_t n;
        if (!querySize.find(res.entries[i].key, n)) { o.err = "query set " + std::to_string(res.entries[i].key) + " has no size entry"; return; }
        const size_t at = o.block.size();
        if (!combineScore(res.ptr(i), n, memberToSet, o.block, o.err)) return;
        o.block.push_back('\0'); o.keys.push_back(res.entries[i].key); o.lengths.push_back(o.block.size() - at);
    }, err)) return fail("combinescore: " + err);
    flushChunks(outs, w);
    if (!w.close(err)) return fail("combinescore: " + err);
    return 0;
}

int runFilterMatchByFdr(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--fdr", "--report-fdr"}), 3, f, err)) return fail("filtermatchbyfdr: " + err);
    if (!uncompressedOnly(f, err)) return fail("filtermatchbyfdr: " + err);
    Db pos, neg;
    if (!loadDb(f.positional[0], pos, err) || !loadDb(f.positional[1], neg, err)) return fail("filtermatchbyfdr: " + err);
    std::vector<double> P, N;
    if (!collectScores(pos, P, err) || !collectScores(neg, N, err)) return fail("filtermatchbyfdr: " + err);
    const bool report = f.num("--report-fdr", 0) != 0;
    FdrResult r;
    fdrThreshold(P, N, !neg.entries.empty(), f.real("--fdr", 0.05), report, r);
    DbWriter w; w.path = f.positional[2]; w.dbtype = 12;
    std::string out; const char *c[8];
    for (size_t i = 0; i < pos.entries.size(); i++) {
        out.clear();
        for (const char *p = pos.ptr(i); *p != '\0';) {
            if (words(p, c, 8) < 3) return fail("filtermatchbyfdr: invalid result record (a score line has three columns)");      // (found by tests/test_host_fuzz.py: c[3] of a shorter line is the previous line's)
            const double score = strtod(c[1], NULL);
            p = skipLine(p);
            if (score < r.threshold) continue;
            out.append(c[0], (size_t) (c[3] - c[0]));
            if (report) { out.push_back('\t'); const size_t at = (size_t) (std::find(r.uniqueScores.begin(), r.uniqueScores.end(), score) - r.uniqueScores.begin()); out += fmtE(at < r.fdr.size() ? r.fdr[at] : 0.0); }
            out.push_back('\n');
        }
        w.add(pos.entries[i].key, out);
    }
    if (!w.close(err)) return fail("filtermatchbyfdr: " + err);
    return 0;
}

// The taxonomy columns of the report (SummarizeResults.cpp:17-22,118-120,156-180) need three lookups of NcbiTaxonomy: node by id,
// names along the path to the root, rank tables.  This reads the three NCBI dump files the way taxonomy/NcbiTaxonomy.cpp:110-183,
// 429-455 does; the LCA machinery of that class is not needed for a report and stays out of scope.
struct TaxTable {
    struct Node { int taxId, parent; std::string rank, name; bool named = false; };
    std::vector<Node> nodes; std::vector<int> slot; int maxTaxId = 0;      // slot[taxId] = index into nodes, -1 = none (merged ids share a node)
    static std::vector<std::string> fields(const std::string &line, size_t maxCol) {      // NcbiTaxonomy.cpp:92-107: pieces between "\t|\t"
        std::vector<std::string> r; size_t prev = 0, pos = 0, n = 0;
        do { pos = line.find("\t|\t", prev); if (pos == std::string::npos) pos = line.size(); r.push_back(line.substr(prev, pos - prev)); prev = pos + 3; n++; } while (pos < line.size() && prev < line.size() && n < maxCol);
        return r;
    }
    bool exists(int id) const { return id >= 0 && id <= maxTaxId && slot[(size_t) id] != -1; }
    const Node *node(int id) const { return (id == 0 || !exists(id)) ? nullptr : &nodes[(size_t) slot[(size_t) id]]; }
    bool load(const std::string &prefix, std::string &err) {
        std::string nodesFile = prefix + "_nodes.dmp", namesFile = prefix + "_names.dmp", mergedFile = prefix + "_merged.dmp";
        struct stat st;
        if (stat(namesFile.c_str(), &st) != 0 || stat(mergedFile.c_str(), &st) != 0) {      // openTaxonomy falls back to the working directory (:510-525)
            if (stat("nodes.dmp", &st) == 0 && stat("names.dmp", &st) == 0 && stat("merged.dmp", &st) == 0) { nodesFile = "nodes.dmp"; namesFile = "names.dmp"; mergedFile = "merged.dmp"; }
            else { err = "names.dmp, nodes.dmp, merged.dmp from NCBI taxdump could not be found"; return false; }
        }
        std::string line;
        {   std::ifstream in(nodesFile);
            if (!in.good()) { err = "File " + nodesFile + " not found"; return false; }
            std::map<int, int> first;
            while (std::getline(in, line)) {
                const std::vector<std::string> c = fields(line, 3);
                if (c.size() < 3) { err = nodesFile + ": malformed line"; return false; }
                Node n; n.taxId = (int) strtol(c[0].c_str(), NULL, 10); n.parent = (int) strtol(c[1].c_str(), NULL, 10); n.rank = c[2];
                if (n.taxId < 0) { err = nodesFile + ": negative taxon id"; return false; }
                maxTaxId = std::max(maxTaxId, n.taxId);
                first.emplace(n.taxId, (int) nodes.size());
                nodes.push_back(n);
            }
            slot.assign((size_t) maxTaxId + 1, -1);
            for (const auto &kv : first) slot[(size_t) kv.first] = kv.second;
            for (const Node &n : nodes) if (!exists(n.parent)) { err = "Inconsistent nodes.dmp taxonomy file! Cannot find parent taxon with ID " + std::to_string(n.parent); return false; } }
        {   std::ifstream in(mergedFile);
            if (!in.good()) { err = "File " + mergedFile + " not found"; return false; }
            while (std::getline(in, line)) {
                const std::vector<std::string> c = fields(line, 2);
                if (c.size() != 2) { err = mergedFile + ": invalid entry"; return false; }
                const unsigned long oldId = strtoul(c[0].c_str(), NULL, 10), newId = strtoul(c[1].c_str(), NULL, 10);
                if (oldId > (unsigned long) maxTaxId) continue;      // (the reference writes past its table for such a line)
                if (!exists((int) oldId) && newId <= (unsigned long) maxTaxId && exists((int) newId)) slot[oldId] = slot[newId];
            } }
        {   std::ifstream in(namesFile);
            if (!in.good()) { err = "File " + namesFile + " not found"; return false; }
            while (std::getline(in, line)) {
                if (line.find("scientific name") == std::string::npos) continue;
                const std::vector<std::string> c = fields(line, 2);
                if (c.size() != 2) { err = namesFile + ": invalid name entry"; return false; }
                const int id = (int) strtol(c[0].c_str(), NULL, 10);
                if (!exists(id)) { err = "loadNames: Taxon " + std::to_string(id) + " not present in nodes file"; return false; }
                Node &n = nodes[(size_t) slot[(size_t) id]]; n.name = c[1]; n.named = true;
            } }
        return true;
    }
    // rank order and one-letter codes of taxonomy/NcbiTaxonomy.h:57-93
    static int rankIndex(const std::string &rank) {
        static const char *order[] = {"forma", "varietas", "subspecies", "species", "species subgroup", "species group", "subgenus", "genus", "subtribe", "tribe", "subfamily", "family",
                                      "superfamily", "parvorder", "infraorder", "suborder", "order", "superorder", "infraclass", "subclass", "class", "superclass", "subphylum", "phylum",
                                      "superphylum", "subkingdom", "kingdom", "superkingdom"};
        for (size_t i = 0; i < sizeof(order) / sizeof(order[0]); i++) if (rank == order[i]) return (int) i + 1;
        return -1;
    }
    static char shortRank(const std::string &rank) {
        static const std::pair<const char *, char> code[] = {{"species", 's'}, {"genus", 'g'}, {"family", 'f'}, {"order", 'o'}, {"class", 'c'}, {"phylum", 'p'}, {"kingdom", 'k'}, {"superkingdom", 'd'}};
        for (const auto &c : code) if (rank == c.first) return c.second;
        return '-';
    }
    // AtRanks (NcbiTaxonomy.cpp:312-338): the name at each requested rank on the path to the root (the lowest node of a rank wins), "uc_<name>" for
    // ranks below the node's own, "unknown" otherwise
    std::string atRanks(const Node *n, const std::vector<std::string> &levels) const {
        std::map<std::string, std::string> onPath;
        for (const Node *p = n;; p = node(p->parent)) {
            if (p->taxId == 1) { onPath.emplace(p->rank, p->name); break; }
            if (p->rank != "no_rank" && p->rank != "no rank") onPath.emplace(p->rank, p->name);
            if (p->parent == p->taxId) break;      // a root that is not taxon 1 (the reference never leaves this loop then)
        }
        const int base = rankIndex(n->rank); std::string out;
        for (size_t i = 0; i < levels.size(); i++) {
            if (i) out.push_back(';');
            const auto it = onPath.find(levels[i]);
            if (it != onPath.end()) out += it->second;
            else if (rankIndex(levels[i]) < base) out += "uc_" + n->name;
            else out += "unknown";
        }
        return out;
    }
    // taxLineage (NcbiTaxonomy.cpp:367-390): the nodes from below the root down to this one
    std::string lineage(const Node *n, bool names) const {
        std::vector<const Node *> path;
        do { path.push_back(n); n = node(n->parent); } while (n->parent != n->taxId);
        std::string out;
        for (size_t i = path.size(); i-- > 0;) {
            if (names) { out.push_back(shortRank(path[i]->rank)); out.push_back('_'); out += path[i]->name; } else out += std::to_string(path[i]->taxId);
            if (i > 0) out.push_back(';');
        }
        return out;
    }
};

struct SummaryOptions { int fmt = 1; bool reportFdr = false; const TaxTable *tax = nullptr; std::vector<std::string> ranks; int lineage = 0; };

// one match entry joined with the alignment lines of its query set (SummarizeResults.cpp:66-180); fmt: 0 short, 1 long, 2 long + alignment
static bool summarizeEntry(const char *match, const char *aln, const SummaryOptions &o, std::string &out, std::string &err) {
    const char *e[20]; std::string cScore, fdr, tmp; char nb[24];
    // a column without the separator behind it; a line that ends in white space has an EMPTY last word (words() is Util::getWordsOfLine), whose "length" would be -1
    // (tests/test_host_fuzz.py found it: basic_string::append threw)
    auto col = [](std::string &to, const char *from, const char *next) { if (next > from + 1) to.append(from, (size_t) (next - from - 1)); };
    for (const char *m = match; *m != '\0';) {
        const size_t matchColumns = words(m, e, 20);
        if (matchColumns < 2) { err = "summarizeresults: invalid match record"; return false; }
        if (o.reportFdr && matchColumns < 4) { err = "summarizeresults: --report-fdr 1 needs the FDR column filtermatchbyfdr --report-fdr 1 writes"; return false; }
        m = skipLine(m);
        const unsigned long targetSet = atou(e[0]);
        size_t lines = 0; int taxId = 0;
        for (const char *cur = aln; *cur != '\0';) {
            if (atou(cur) != targetSet) { cur = skipLine(cur); continue; }
            if (lines == 0) { cScore.assign(e[1], (size_t) (e[2] - e[1])); if (o.reportFdr) fdr.assign(e[3], (size_t) (e[4] - e[3])); }
            const size_t columns = words(cur, e, 20);
            if (columns < 10 || (o.fmt == 2 && columns < 12)) { err = "summarizeresults: invalid alignment result record"; return false; }
            cur = skipLine(cur);
            if (lines == 0) {
                out.push_back('#'); col(out, e[2], e[3]); out.push_back('\t'); col(out, e[3], e[4]); out.push_back('\t'); out += cScore;
                if (o.reportFdr) { out += fdr; out.push_back('\t'); }
                if (o.tax) {
                    if (columns < 13) { err = "summarizeresults: the target set has a taxonomy but the alignment records carry no taxid column"; return false; }
                    taxId = (int) strtol(e[12], NULL, 10);
                }
            }
            lines++;
            if (o.fmt != 1 && o.fmt != 2) continue;
            tmp.push_back('>'); col(tmp, e[1], e[2]);
            for (int c : {3, 4, 5, 6, 8, 9}) { tmp.push_back('\t'); col(tmp, e[c], e[c + 1]); }
            if (columns == 15) { tmp.push_back('\t'); col(tmp, e[13], e[14]); tmp.push_back('\t'); tmp.append(e[14], (size_t) (e[15] - e[14])); }
            if (o.fmt == 2) { tmp.push_back('\n'); col(tmp, e[10], e[11]); tmp.push_back('\n'); col(tmp, e[11], e[12]); }
            tmp.push_back('\n');
        }
        if (lines > 0) {
            char *x = u32toa((uint32_t) lines, nb); out.append(nb, (size_t) (x - nb));
            const TaxTable::Node *n = o.tax ? o.tax->node(taxId) : nullptr;
            if (n) {
                if (!n->named) { err = "summarizeresults: taxon " + std::to_string(n->taxId) + " has no scientific name in names.dmp"; return false; }
                out.push_back('\t'); out += std::to_string(n->taxId); out.push_back('\t'); out += n->rank; out.push_back('\t'); out += n->name;
                if (!o.ranks.empty()) { out.push_back('\t'); out += o.tax->atRanks(n, o.ranks); }
                if (o.lineage == 1 || o.lineage == 2) { out.push_back('\t'); out += o.tax->lineage(n, o.lineage == 1); }
            }
            out.push_back('\n'); out += tmp;
        }
        tmp.clear();
    }
    return true;
}

int runSummarizeResults(int argc, const char **argv) {
    Flags f; std::string err;
    if (!parseFlags(argc, argv, with({"--fmt", "--report-fdr", "--lca-ranks", "--tax-lineage", "--db-output"}), 4, f, err)) return fail("summarizeresults: " + err);
    if (!uncompressedOnly(f, err)) return fail("summarizeresults: " + err);
    SummaryOptions o; o.fmt = (int) f.num("--fmt", 1); o.reportFdr = f.num("--report-fdr", 0) != 0; o.lineage = (int) f.num("--tax-lineage", 0);
    {   const std::string ranks = f.str("--lca-ranks", "");      // NcbiTaxonomy::parseRanks: comma separated, every name a known rank
        for (size_t from = 0; from < ranks.size();) { size_t to = ranks.find(',', from); if (to == std::string::npos) to = ranks.size(); o.ranks.push_back(ranks.substr(from, to - from)); from = to + 1; }
        for (const std::string &r : o.ranks) if (TaxTable::rankIndex(r) == -1) return fail("summarizeresults: Invalid taxonomic rank " + r + " given"); }
    TaxTable tax;
    { struct stat st; if (stat((f.positional[0] + "_nodes.dmp").c_str(), &st) == 0) { if (!tax.load(f.positional[0], err)) return fail("summarizeresults: " + err); o.tax = &tax; } }
    Db match, aln;
    if (!loadDb(f.positional[1], match, err) || !loadDb(f.positional[2], aln, err)) return fail("summarizeresults: " + err);
    KeyMap aIdx; keyIndex(aln, aIdx);
    const bool dbOut = f.num("--db-output", 0) != 0;
    DbWriter w; w.path = f.positional[3]; w.dbtype = 12;
    std::string out, flat;
    for (size_t i = 0; i < match.entries.size(); i++) {
        uint32_t a;
        if (!aIdx.find(match.entries[i].key, a)) continue;
        out.clear();
        if (!summarizeEntry(match.ptr(i), aln.ptr(a), o, out, err)) return fail(err);
        if (dbOut) w.add(match.entries[i].key, out); else flat += out;
    }
    if (dbOut) { if (!w.close(err)) return fail("summarizeresults: " + err); return 0; }
    // --db-output 0 (the workflow's default): one flat text file, no index (SummarizeResults.cpp:188-191)
    std::ofstream o2(f.positional[3], std::ios::binary); o2.write(flat.data(), (std::streamsize) flat.size());
    if (!o2.good()) return fail("summarizeresults: cannot write " + f.positional[3]);
    return 0;
}

}  // namespace sets
}  // namespace sp
