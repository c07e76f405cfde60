// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// Second driver over the REAL reference sources (compiled in place from /root/reference by oracle/Makefile, outputs only
// under oracle/_ref/): the set-level steps around the search hot path on MMseqs2 DB
// triples, through the reference's own DBReader / DBWriter.  tests/golden/make_examples.py drives it on the bundled
// examples/ data and commits the resulting DBs as fixtures (tests/golden/examples.json.gz).
//
// What is the reference's own code here and what is a mirror:
//   * besthitperset, combinescore: the reference's aggregator CLASSES (BestHitBySetFilter in multihit/besthitperset.cpp,
//     PvalueAggregator in src/util/CombineScore.cpp) are compiled by #including those files and run through the real
//     Aggregation::run() (multihit/Aggregation.cpp:47-91).  Only their three-line entry functions, which parse the command
//     line with Parameters (Parameters.cpp needs cmake-generated headers and is NOT built), are bypassed.
//   * extractorfs, translatenucs, reverseseqbycodon, proteinaln2nucl, combineprotnuclaln, mergeresultsbyset,
//     filtermatchbyfdr, summarizeresults are free functions that begin with Parameters::parseParameters, so their bodies are
//     MIRRORED below on top of the real classes they use (Orf, TranslateNucl, Matcher::parseAlignmentRecord /
//     resultToBuffer, NucleotideMatrix, EvalueComputation, Util::getWordsOfLine, SSTR, DBWriter::createRenumberedDB).
//     Each mirror names the lines it follows.
//
// usage: ref_sets <step> <args...>   (steps and arguments: see main)
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "Command.h"
#include "DBReader.h"
#include "DBWriter.h"
#include "Debug.h"
// the evaluer member is private; the driver only READS the fitted Gumbel parameters out of it
#include "sls_alignment_evaluer.hpp"
#define private public
#include "EvalueComputation.h"
#undef private
#include "FileUtil.h"
#include "Matcher.h"
#include "NucleotideMatrix.h"
#include "Orf.h"
#include "SubstitutionMatrix.h"
#include "TranslateNucl.h"
#include "Util.h"
#include "itoa.h"

// the reference's aggregator classes, compiled as they are
#include "multihit/besthitperset.cpp"
#include "CombineScore.cpp"

const char *binary_name = "ref_sets";
const char *tool_name = "ref_sets";
const char *tool_introduction = "";
const char *main_author = "";
const char *show_extended_help = NULL;
const char *show_bash_info = NULL;
bool hide_base_commands = true;
std::vector<Command> commands;

typedef DBReader<unsigned int> Reader;
static std::string idx(const std::string &p) { return p + ".index"; }

// ---- extractorfs <nuclDB> <outDB> <minLength> <orfStartMode>   (util/extractorfs.cpp:18-160; workflow defaults:
// --max-length 32734 --max-gaps INT_MAX --contig-start-mode 2 --contig-end-mode 2 --forward/reverse-frames 1,2,3
// --translation-table 1 --use-all-table-starts 0 --translate 0 --create-lookup 1)
static int stepExtractOrfs(const std::string &in, const std::string &out, size_t minLen, unsigned startMode) {
    Reader reader(in.c_str(), idx(in).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA);
    reader.open(Reader::NOSORT);
    DBWriter seqW(out.c_str(), idx(out).c_str(), 1, 0, Parameters::DBTYPE_NUCLEOTIDES);
    seqW.open();
    const std::string hdr = out + "_h";
    DBWriter hdrW(hdr.c_str(), idx(hdr).c_str(), 1, 0, Parameters::DBTYPE_GENERIC_DB);
    hdrW.open();
    const unsigned frames = Orf::getFrames("1,2,3");
    Orf orf(1, false);
    std::vector<Orf::SequenceLocation> res;
    char buffer[1024];
    const char nl = '\n';
    for (size_t i = 0; i < reader.getSize(); i++) {
        const unsigned key = reader.getDbKey(i);
        const char *data = reader.getData(i, 0);
        const size_t L = reader.getSeqLen(i);
        if (!orf.setSequence(data, L)) continue;
        orf.findAll(res, minLen, 32734, INT_MAX, frames, frames, startMode);
        for (size_t k = 0; k < res.size(); k++) {
            const Orf::SequenceLocation loc = res[k];
            std::pair<const char *, size_t> s = orf.getSequence(loc);
            size_t from = loc.from, to = loc.to;
            if (loc.strand == Orf::STRAND_MINUS) { from = (L - 1) - loc.from; to = (L - 1) - loc.to; }
            Orf::writeOrfHeader(buffer, key, from, to, loc.hasIncompleteStart, loc.hasIncompleteEnd);
            seqW.writeStart(0);
            seqW.writeAdd(s.first, s.second, 0);
            seqW.writeAdd(&nl, 1, 0);
            seqW.writeEnd(key, 0);
            hdrW.writeData(buffer, strlen(buffer), key, 0);
        }
        res.clear();
    }
    hdrW.close(true);
    seqW.close(true);
    reader.close();
    DBWriter::createRenumberedDB(hdr, idx(hdr), "", "");
    DBWriter::createRenumberedDB(out, idx(out), in, idx(in));        // --create-lookup 1: <out>.lookup = new id, contig key, set
    return 0;
}

// ---- translatenucs <nuclOrfDB> <outDB>   (util/translatenucs.cpp:16-115, --translation-table 1 --add-orf-stop 0)
static int stepTranslate(const std::string &in, const std::string &out) {
    Reader reader(in.c_str(), idx(in).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA);
    reader.open(Reader::LINEAR_ACCCESS);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, Parameters::DBTYPE_AMINO_ACIDS);
    w.open();
    TranslateNucl tr(static_cast<TranslateNucl::GenCode>(1));
    const size_t maxSeqLen = 65535;
    std::vector<char> aa(maxSeqLen + 8);
    for (size_t i = 0; i < reader.getSize(); i++) {
        const unsigned key = reader.getDbKey(i);
        char *data = reader.getData(i, 0);
        if (*data == '\0') continue;
        size_t length = reader.getEntryLen(i) - 1;
        if ((data[length] != '\n' && length % 3 != 0) && (data[length - 1] == '\n' && (length - 1) % 3 != 0)) length -= length % 3;
        if (length < 3) continue;
        if (length > 3 * maxSeqLen) length = 3 * maxSeqLen;
        tr.translate(aa.data(), data, length);
        aa[length / 3] = '\n';
        w.writeData(aa.data(), length / 3 + 1, key, 0);
    }
    w.close(true);
    reader.close();
    return 0;
}

// ---- reverseseqbycodon <nuclOrfDB> <outDB>   (src/util/ReverseSeqByCodon.cpp:13-62)
static int stepReverse(const std::string &in, const std::string &out) {
    Reader reader(in.c_str(), idx(in).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA);
    reader.open(Reader::LINEAR_ACCCESS);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, reader.getDbtype());
    w.open();
    std::string rev, codon;
    for (size_t id = 0; id < reader.getSize(); id++) {
        const unsigned key = reader.getDbKey(id);
        const char *seq = reader.getData(id, 0);
        const size_t L = reader.getSeqLen(id);
        codon.clear();
        for (size_t i = 0; i < L; i++) {
            codon.push_back(seq[L - i - 1]);
            if (codon.size() == 3) { std::reverse(codon.begin(), codon.end()); rev.append(codon); codon.clear(); }
        }
        rev.push_back('\n');
        w.writeData(rev.c_str(), rev.size(), key, 0, true);
        rev.clear();
    }
    w.close(true);
    reader.close();
    return 0;
}

// ---- proteinaln2nucl <qNuclOrf> <tNuclOrf> <qAA> <tAA> <protAln> <out> <nucleotide.out> <gapOpen> <gapExtend>
// (util/proteinaln2nucl.cpp:13-205).  Also prints the ALP parameters EvalueComputation fitted for the nucleotide matrix.
static int stepProteinAln2Nucl(char **a) {
    const std::string qn = a[0], tn = a[1], qa = a[2], ta = a[3], aln = a[4], out = a[5];
    Reader qNuc(qn.c_str(), idx(qn).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); qNuc.open(Reader::NOSORT);
    Reader qAA(qa.c_str(), idx(qa).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); qAA.open(Reader::NOSORT);
    Reader tNuc(tn.c_str(), idx(tn).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); tNuc.open(Reader::NOSORT);
    Reader tAA(ta.c_str(), idx(ta).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); tAA.open(Reader::NOSORT);
    NucleotideMatrix subMat(a[6], 1.0, 0.0);
    SubstitutionMatrix::FastMatrix fast = SubstitutionMatrix::createAsciiSubMat(subMat);
    const int gapOpen = atoi(a[7]), gapExtend = atoi(a[8]);
    EvalueComputation evaluer(tNuc.getAminoAcidDBSize(), &subMat, gapOpen, gapExtend);
    Reader alnDb(aln.c_str(), idx(aln).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); alnDb.open(Reader::LINEAR_ACCCESS);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, Parameters::DBTYPE_ALIGNMENT_RES);
    w.open();
    std::vector<char> buffer(1024 + 32768 * 4);
    std::string result, bt;
    for (size_t i = 0; i < alnDb.getSize(); i++) {
        const unsigned key = alnDb.getDbKey(i);
        char *data = alnDb.getData(i, 0);
        const unsigned qId = qNuc.getId(key);
        const char *qSeq = qNuc.getData(qId, 0);
        const unsigned qLen = qNuc.getSeqLen(qId);
        const bool qStart = qAA.getDataByDBKey(key, 0)[0] == '*';
        while (*data != '\0') {
            Matcher::result_t r = Matcher::parseAlignmentRecord(data, true);
            data = Util::skipLine(data);
            if ((qStart && r.qStartPos == 0) || r.backtrace.empty()) { fprintf(stderr, "proteinaln2nucl: unsupported record\n"); return 1; }
            const unsigned tId = tNuc.getId(r.dbKey);
            const char *tSeq = tNuc.getData(tId, 0);
            const bool tStart = tAA.getDataByDBKey(r.dbKey, 0)[0] == '*';
            if (tStart && r.dbStartPos == 0) { fprintf(stderr, "proteinaln2nucl: unalignable character\n"); return 1; }
            r.dbStartPos = r.dbStartPos * 3 + (tStart ? -3 : 0);
            r.dbEndPos = r.dbEndPos * 3 + 2 + (tStart ? -3 : 0);
            r.dbLen = tNuc.getSeqLen(tId);
            r.qStartPos = r.qStartPos * 3 + (qStart ? -3 : 0);
            r.qEndPos = r.qEndPos * 3 + 2 + (qStart ? -3 : 0);
            r.qLen = qLen;
            size_t ids = 0, alnLen = 0;
            int qPos = r.qStartPos, tPos = r.dbStartPos, score = 0;
            for (size_t pos = 0; pos < r.backtrace.size(); pos++) {
                int cnt = 0;
                if (isdigit(r.backtrace[pos])) { cnt += Util::fast_atoi<int>(r.backtrace.c_str() + pos); while (isdigit(r.backtrace[pos])) pos++; }
                const char op = r.backtrace[pos];
                if (op == 'M') {
                    for (int b = 0; b < cnt * 3; b++) { ids += (qSeq[qPos] == tSeq[tPos]); score += fast.matrix[(int) qSeq[qPos]][(int) tSeq[tPos]]; tPos++; qPos++; }
                } else if (op == 'D') { tPos += cnt * 3; score -= gapOpen + ((cnt - 1) * 3) * gapExtend; }
                else if (op == 'I') { qPos += cnt * 3; score -= gapOpen + ((cnt - 1) * 3) * gapExtend; }
                else continue;
                alnLen += cnt * 3;
                bt.append(SSTR(cnt * 3)); bt.push_back(op);
            }
            r.score = evaluer.computeBitScore(score);
            r.eval = evaluer.computeEvalue(score, qLen);
            r.backtrace = bt;
            r.seqId = static_cast<float>(ids) / static_cast<float>(alnLen);
            const size_t len = Matcher::resultToBuffer(buffer.data(), r, true, false);
            result.append(buffer.data(), len);
            bt.clear();
        }
        w.writeData(result.c_str(), result.length(), key, 0);
        result.clear();
    }
    w.close();
    // probes of the nucleotide E-value statistics for the fixtures: evalue / bit score of a few raw scores at a few query lengths
    {   const Sls::ALP_set_of_parameters &p = evaluer.evaluer.parameters();
        printf("{\"alp\":[\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\"],", p.lambda, p.K, p.a_I, p.b_I, p.alpha_I, p.beta_I,
               p.a_J, p.b_J, p.alpha_J, p.beta_J, p.sigma, p.tau); }
    printf("\"dbResidues\":%zu,\"probes\":[", (size_t) tNuc.getAminoAcidDBSize());
    bool first = true;
    for (int s = -10; s <= 120; s += 7) for (int L = 27; L <= 60; L += 11) {
        printf("%s[%d,%d,\"%a\",%d]", first ? "" : ",", s, L, evaluer.computeEvalue(s, L), (int) evaluer.computeBitScore(s));
        first = false;
    }
    printf("]}\n");
    alnDb.close(); qNuc.close(); qAA.close(); tNuc.close(); tAA.close();
    delete[] fast.matrix; delete[] fast.matrixData;
    return 0;
}

// ---- combineprotnuclaln <protAln> <nuclAln> <out>   (src/util/CombineProtNuclAln.cpp:13-106)
static int stepCombineProtNucl(const std::string &prot, const std::string &nucl, const std::string &out) {
    Reader protR(prot.c_str(), idx(prot).c_str(), 1, Reader::USE_DATA | Reader::USE_INDEX); protR.open(Reader::LINEAR_ACCCESS);
    Reader nuclR(nucl.c_str(), idx(nucl).c_str(), 1, Reader::USE_DATA | Reader::USE_INDEX); nuclR.open(Reader::NOSORT);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, protR.getDbtype());
    w.open();
    const char *pe[255], *ne[255];
    std::string buffer;
    for (size_t id = 0; id < protR.getSize(); id++) {
        const unsigned key = protR.getDbKey(id);
        const size_t nId = nuclR.getId(key);
        if (nId == UINT_MAX) continue;
        char *nuclData = nuclR.getData(nId, 0), *protData = protR.getData(id, 0);
        while (*protData != '\0') {
            if (Util::getWordsOfLine(protData, pe, 255) < (size_t) Matcher::ALN_RES_WITHOUT_BT_COL_CNT) return 1;
            protData = Util::skipLine(protData);
            const size_t target = Util::fast_atoi<size_t>(pe[0]);
            const double protEval = strtod(pe[3], NULL);
            double updated = FLT_MAX, nuclSeqId = 0;
            for (char *cur = nuclData; *cur != '\0';) {
                const unsigned k = Util::fast_atoi<size_t>(cur);
                if (k != target) { cur = Util::skipLine(cur); continue; }
                if (Util::getWordsOfLine(cur, ne, 255) < (size_t) Matcher::ALN_RES_WITHOUT_BT_COL_CNT) return 1;
                cur = Util::skipLine(cur);
                nuclSeqId = strtod(ne[2], NULL);
                const double nuclEval = strtod(ne[3], NULL);
                updated = ((log(protEval) * 0.5 + log(nuclEval) * 0.5) < log(nuclEval)) ? exp(log(protEval) * 0.5 + log(nuclEval) * 0.5) : nuclEval;
            }
            buffer.append(pe[0], pe[2] - pe[0]);
            buffer.append(SSTR(nuclSeqId)); buffer.append("\t");
            buffer.append(SSTR(updated)); buffer.append("\t");
            buffer.append(pe[4], (protData - 1) - pe[4]);
            buffer.append("\n");
        }
        w.writeData(buffer.c_str(), buffer.length(), key, 0);
        buffer.clear();
    }
    w.close();
    protR.close(); nuclR.close();
    return 0;
}

// ---- mergeresultsbyset <setToMember> <resultDB> <out>   (util/mergeresultsbyset.cpp:10-58)
static int stepMergeBySet(const std::string &sets, const std::string &res, const std::string &out) {
    Reader setR(sets.c_str(), idx(sets).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); setR.open(Reader::LINEAR_ACCCESS);
    Reader resR(res.c_str(), idx(res).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); resR.open(Reader::NOSORT);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, resR.getDbtype());
    w.open();
    std::string buffer; char dbKey[255];
    for (size_t i = 0; i < setR.getSize(); i++) {
        for (char *data = setR.getData(i, 0); *data != '\0'; data = Util::skipLine(data)) {
            Util::parseKey(data, dbKey);
            const size_t id = resR.getId(Util::fast_atoi<unsigned int>(dbKey));
            if (id == UINT_MAX) { fprintf(stderr, "mergeresultsbyset: invalid key %s\n", dbKey); return 1; }
            buffer.append(resR.getData(id, 0));
        }
        w.writeData(buffer.c_str(), buffer.length(), setR.getDbKey(i), 0);
        buffer.clear();
    }
    w.close();
    resR.close(); setR.close();
    return 0;
}

// ---- filtermatchbyfdr <cScore> <cScoreControl> <out> <fdr> <reportFdr>   (src/util/FilterMatchbyFdr.cpp:13-264)
static void readScores(Reader &db, std::vector<double> &v) {
    const char *entry[255];
    for (size_t id = 0; id < db.getSize(); id++)
        for (char *data = db.getData(id, 0); *data != '\0'; data = Util::skipLine(data)) {
            if (Util::getWordsOfLine(data, entry, 255) < 2) { fprintf(stderr, "invalid record\n"); exit(1); }
            v.push_back(strtod(entry[1], NULL));
        }
}
static int stepFdr(const std::string &pos, const std::string &neg, const std::string &out, double fdrCutoff, int reportFdr) {
    Reader posDb(pos.c_str(), idx(pos).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); posDb.open(Reader::LINEAR_ACCCESS);
    std::vector<double> P, N;
    readScores(posDb, P);
    std::sort(P.rbegin(), P.rend());
    if (P.empty()) P.push_back(0);
    Reader negDb(neg.c_str(), idx(neg).c_str(), 1, Reader::USE_INDEX | Reader::USE_DATA); negDb.open(Reader::LINEAR_ACCCESS);
    double threshold = 0;
    std::vector<double> fdrList, uniq;
    if (negDb.getSize() > 0) {
        readScores(negDb, N);
        std::sort(N.rbegin(), N.rend());
        std::vector<double> x, y;
        double cur = DBL_MAX; size_t nc = 0, pc = 0;
        for (size_t i = 0; i < P.size(); i++) {
            if (P[pc] < cur) {
                cur = P[pc];
                while (nc < N.size() && cur < N[nc]) nc++;
                uniq.push_back(cur);
                y.push_back(1.0 * pc / P.size());
                x.push_back(1.0 * (nc + 0.5) / (N.size() + 1));
            }
            pc++;
        }
        x.push_back(1.0); y.push_back(1.0);
        std::vector<double> slopes; std::vector<size_t> hull;
        for (size_t i = 0; i < x.size() - 1;) {          // upper convex hull of the (x, y) curve, greedy by steepest slope
            size_t jMax = i + 1; double best = 0;
            for (size_t j = i + 1; j < x.size(); j++) { const double s = 1.0 * (y[j] - y[i]) / (x[j] - x[i]); if (s >= best) { jMax = j; best = s; } }
            i = jMax; slopes.push_back(best); hull.push_back(jMax);
        }
        const double pi0 = (slopes.end()[-2] + slopes.end()[-1]) / 2;
        size_t i = 0; double fdr = 0;
        if (x[hull.back()] * pi0 / y[hull.back()] >= fdrCutoff)
            while (fdr <= fdrCutoff) { fdr = x[hull[i]] * pi0 / y[hull[i]]; i++; }
        if (i < 2) threshold = P.back();
        else {
            size_t j = hull[i - 2];
            double TPFP = y[j], FP = x[j] * pi0;
            fdr = 0;
            while (fdr <= fdrCutoff) { j++; const double dx = x[j] - x[j - 1]; TPFP += dx * slopes[i - 1]; FP += dx * pi0; fdr = FP / TPFP; }
            threshold = uniq[j];
        }
        if (reportFdr) {
            for (size_t j = 0; j < hull[0]; j++) fdrList.push_back(0.0);
            for (size_t h = 0; h + 1 < hull.size(); h++) {
                double TPFP = y[hull[h]], FP = x[hull[h]] * pi0;
                for (size_t j = hull[h]; j < hull[h + 1]; j++) {
                    if (std::isinf(slopes[h])) fdrList.push_back(x[hull[h]] * pi0 / y[hull[h]]);
                    else { TPFP += (x[j] - x[j - 1]) * slopes[h]; FP += (x[j] - x[j - 1]) * pi0; fdrList.push_back(FP / TPFP); }
                }
            }
        }
    } else {
        if (reportFdr) { double s = DBL_MIN; for (size_t i = 0; i < P.size(); i++) if (s < P[i]) { uniq.push_back(P[i]); fdrList.push_back(0.0); s = P[i]; } }
        threshold = P.back();
    }
    negDb.close();
    printf("{\"threshold\":\"%a\"}\n", threshold);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, Parameters::DBTYPE_GENERIC_DB);
    w.open();
    const char *entry[255]; std::string buffer;
    for (size_t id = 0; id < posDb.getSize(); id++) {
        for (char *data = posDb.getData(id, 0); *data != '\0';) {
            Util::getWordsOfLine(data, entry, 255);
            const double score = strtod(entry[1], NULL);
            data = Util::skipLine(data);
            if (score >= threshold) {
                buffer.append(entry[0], entry[3] - entry[0]);
                if (reportFdr) { buffer.append("\t"); buffer.append(SSTR(fdrList[std::distance(uniq.begin(), std::find(uniq.begin(), uniq.end(), score))])); }
                buffer.append("\n");
            }
        }
        w.writeData(buffer.c_str(), buffer.length(), posDb.getDbKey(id), 0);
        buffer.clear();
    }
    w.close();
    posDb.close();
    return 0;
}

// ---- summarizeresults <matchDB> <alnDB> <out> <fmt>   (src/util/SummarizeResults.cpp:13-197; no taxonomy DB, --report-fdr 0,
// --db-output: the entries are written as a generic DB so that they can be compared per key)
static int stepSummarize(const std::string &match, const std::string &aln, const std::string &out, int fmt) {
    Reader matchR(match.c_str(), idx(match).c_str(), 1, Reader::USE_DATA | Reader::USE_INDEX); matchR.open(Reader::LINEAR_ACCCESS);
    Reader alnR(aln.c_str(), idx(aln).c_str(), 1, Reader::USE_DATA | Reader::USE_INDEX); alnR.open(Reader::NOSORT);
    DBWriter w(out.c_str(), idx(out).c_str(), 1, 0, Parameters::DBTYPE_GENERIC_DB);
    w.open();
    const char *e[255]; std::string cScore, buffer, tmp;
    for (size_t id = 0; id < matchR.getSize(); id++) {
        const unsigned key = matchR.getDbKey(id);
        const size_t alnId = alnR.getId(key);
        if (alnId == UINT_MAX) continue;
        char *alnData = alnR.getData(alnId, 0);
        for (char *m = matchR.getData(id, 0); *m != '\0';) {
            if (Util::getWordsOfLine(m, e, 255) < 2) return 1;
            m = Util::skipLine(m);
            const unsigned targetSet = Util::fast_atoi<unsigned int>(e[0]);
            size_t lines = 0;
            for (char *cur = alnData; *cur != '\0';) {
                if (Util::fast_atoi<unsigned int>(cur) != targetSet) { cur = Util::skipLine(cur); continue; }
                if (lines == 0) cScore.assign(e[1], e[2] - e[1]);
                const size_t columns = Util::getWordsOfLine(cur, e, 255);
                if (columns < 10) return 1;
                cur = Util::skipLine(cur);
                if (lines == 0) {
                    buffer.append("#"); buffer.append(e[2], e[3] - e[2] - 1); buffer.append("\t");
                    buffer.append(e[3], e[4] - e[3] - 1); buffer.append("\t"); buffer.append(cScore);
                }
                lines++;
                if (fmt != 1 && fmt != 2) continue;
                tmp.append(">"); tmp.append(e[1], e[2] - e[1] - 1);
                static const int cols[] = {3, 4, 5, 6, 8, 9};
                for (int c : cols) { tmp.append("\t"); tmp.append(e[c], e[c + 1] - e[c] - 1); }
                if (columns == 15) { tmp.append("\t"); tmp.append(e[13], e[14] - e[13] - 1); tmp.append("\t"); tmp.append(e[14], e[15] - e[14]); }
                if (fmt == 2) { tmp.append("\n"); tmp.append(e[10], e[11] - e[10] - 1); tmp.append("\n"); tmp.append(e[11], e[12] - e[11] - 1); }
                tmp.append("\n");
            }
            if (lines > 0) { buffer.append(SSTR(lines)); buffer.append(1, '\n'); buffer.append(tmp); }
            tmp.clear();
        }
        w.writeData(buffer.c_str(), buffer.length(), key, 0);
        buffer.clear();
    }
    w.close();
    matchR.close(); alnR.close();
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_sets <step> <args>\n"); return 2; }
    const std::string step = argv[1];
    char **a = argv + 2; const int n = argc - 2;
    if (step == "extractorfs" && n == 4) return stepExtractOrfs(a[0], a[1], (size_t) atoi(a[2]), (unsigned) atoi(a[3]));
    if (step == "translatenucs" && n == 2) return stepTranslate(a[0], a[1]);
    if (step == "reverseseqbycodon" && n == 2) return stepReverse(a[0], a[1]);
    if (step == "proteinaln2nucl" && n == 9) return stepProteinAln2Nucl(a);
    if (step == "combineprotnuclaln" && n == 3) return stepCombineProtNucl(a[0], a[1], a[2]);
    if (step == "besthitperset" && n == 4) {       // <targetDB> <resultDB> <out> <simpleBestHit>: the reference's class, as is
        BestHitBySetFilter agg(a[0], a[1], a[2], atoi(a[3]) != 0, 1, 0);
        return agg.run();
    }
    if (step == "combinescore" && n == 4) {        // <queryDB> <targetDB> <aggregateMerged> <out>: the reference's class, as is
        PvalueAggregator agg(a[0], a[1], a[2], a[3], 1, 0);
        return agg.run();
    }
    if (step == "mergeresultsbyset" && n == 3) return stepMergeBySet(a[0], a[1], a[2]);
    if (step == "filtermatchbyfdr" && n == 5) return stepFdr(a[0], a[1], a[2], strtod(a[3], NULL), atoi(a[4]));
    if (step == "summarizeresults" && n == 4) return stepSummarize(a[0], a[1], a[2], atoi(a[3]));
    fprintf(stderr, "ref_sets: unknown step / wrong argument count: %s\n", step.c_str());
    return 2;
}
