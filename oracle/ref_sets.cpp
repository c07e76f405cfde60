// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// Driver over the reference's OWN set-level modules (SURVEY.md section 8 rows f1-f4 and the target-side preparation):
//     extractorfs, translatenucs, reverseseqbycodon, proteinaln2nucl, combineprotnuclaln, besthitperset,
//     mergeresultsbyset, combinescore, filtermatchbyfdr, summarizeresults
// Their translation units are compiled where they lie under /root/reference, unmodified (oracle/Makefile, outputs only under
// oracle/_ref/), together with the classes they use (DBReader / DBWriter, Orf, TranslateNucl, Matcher, NucleotideMatrix,
// EvalueComputation + ALP, Aggregation, NcbiTaxonomy).  The one thing that is not the reference's is the parameter
// singleton: Parameters.cpp needs cmake-generated headers, so the modules read their settings from the plain record in
// oracle/ref_par.h (see there), and this file fills that record from its own command line -- the reference's flag names --
// before calling the module function the reference's command table would call.
// (Round 2 re-typed eight of the ten module bodies on top of the reference classes; that mirror is gone.)
//
// usage: ref_sets <module> <db> [<db> ...] [--flag value ...]
//        ref_sets nuclalp <nuclDB> <matrix.out> <gapOpen> <gapExtend>     E-value probes of the nucleotide statistics (fixtures)
// the evaluer member is private; `nuclalp` only READS the fitted Gumbel parameters out of it
#include "sls_alignment_evaluer.hpp"
#define private public
#include "EvalueComputation.h"
#undef private
#include "NucleotideMatrix.h"

#include "ref_par.h"

#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>

const char *binary_name = "ref_sets";
const char *tool_name = "ref_sets";
const char *tool_introduction = "";
const char *main_author = "";
const char *show_extended_help = NULL;
const char *show_bash_info = NULL;
bool hide_base_commands = true;
std::vector<Command> commands;

typedef int (*ModuleFn)(int, const char **, const Command &);
extern int extractorfs(int argc, const char **argv, const Command &command);
extern int translatenucs(int argc, const char **argv, const Command &command);
extern int reverseseqbycodon(int argc, const char **argv, const Command &command);
extern int proteinaln2nucl(int argc, const char **argv, const Command &command);
extern int combineprotnuclaln(int argc, const char **argv, const Command &command);
extern int besthitperset(int argc, const char **argv, const Command &command);
extern int mergeresultsbyset(int argc, const char **argv, const Command &command);
extern int combinescore(int argc, const char **argv, const Command &command);
extern int filtermatchbyfdr(int argc, const char **argv, const Command &command);
extern int summarizeresults(int argc, const char **argv, const Command &command);

static int nuclAlp(const char *db, const char *matrix, int gapOpen, int gapExtend) {
    DBReader<unsigned int> t(db, (std::string(db) + ".index").c_str(), 1, DBReader<unsigned int>::USE_INDEX | DBReader<unsigned int>::USE_DATA);
    t.open(DBReader<unsigned int>::NOSORT);
    NucleotideMatrix subMat(matrix, 1.0, 0.0);
    EvalueComputation evaluer(t.getAminoAcidDBSize(), &subMat, gapOpen, gapExtend);
    const Sls::ALP_set_of_parameters &p = evaluer.evaluer.parameters();
    printf("{\"alp\":[\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\",\"%a\"],", p.lambda, p.K, p.a_I, p.b_I, p.alpha_I, p.beta_I,
           p.a_J, p.b_J, p.alpha_J, p.beta_J, p.sigma, p.tau);
    printf("\"dbResidues\":%zu,\"probes\":[", (size_t) t.getAminoAcidDBSize());
    bool first = true;
    for (int s = -10; s <= 120; s += 7) for (int L = 27; L <= 60; L += 11) {
        printf("%s[%d,%d,\"%a\",%d]", first ? "" : ",", s, L, evaluer.computeEvalue(s, L), (int) evaluer.computeBitScore(s));
        first = false;
    }
    printf("]}\n");
    t.close();
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_sets <module> <db ...> [--flag value ...]\n"); return 2; }
    const std::string module = argv[1];
    if (module == "nuclalp" && argc == 6) return nuclAlp(argv[2], argv[3], atoi(argv[4]), atoi(argv[5]));
    static const std::map<std::string, std::pair<ModuleFn, int>> table = {      // module -> (function, number of DB arguments)
        {"extractorfs", {extractorfs, 2}}, {"translatenucs", {translatenucs, 2}}, {"reverseseqbycodon", {reverseseqbycodon, 2}},
        {"proteinaln2nucl", {proteinaln2nucl, 6}}, {"combineprotnuclaln", {combineprotnuclaln, 3}}, {"besthitperset", {besthitperset, 4}},
        {"mergeresultsbyset", {mergeresultsbyset, 3}}, {"combinescore", {combinescore, 4}}, {"filtermatchbyfdr", {filtermatchbyfdr, 3}},
        {"summarizeresults", {summarizeresults, 4}}};
    const auto it = table.find(module);
    if (it == table.end()) { fprintf(stderr, "ref_sets: unknown module %s\n", module.c_str()); return 2; }
    RefPar &par = RefPar::getInstance();
    std::vector<std::string> dbs;
    int i = 2;
    for (; i < argc && strncmp(argv[i], "--", 2) != 0; i++) dbs.push_back(argv[i]);
    if ((int) dbs.size() != it->second.second) { fprintf(stderr, "ref_sets: %s takes %d DB arguments\n", module.c_str(), it->second.second); return 2; }
    std::string *name[] = {&par.db1, &par.db2, &par.db3, &par.db4, &par.db5, &par.db6}, *index[] = {&par.db1Index, &par.db2Index, &par.db3Index, &par.db4Index, &par.db5Index, &par.db6Index};
    for (size_t k = 0; k < dbs.size(); k++) { *name[k] = dbs[k]; *index[k] = dbs[k] + ".index"; }
    par.hdr1 = par.db1 + "_h"; par.hdr1Index = par.hdr1 + ".index"; par.hdr2 = par.db2 + "_h"; par.hdr2Index = par.hdr2 + ".index";
    std::string subMat = "nucleotide.out";
    for (; i + 1 < argc; i += 2) {
        const std::string f = argv[i]; const char *v = argv[i + 1];
        if (f == "--min-length") par.orfMinLength = atoi(v);
        else if (f == "--max-length") par.orfMaxLength = atoi(v);
        else if (f == "--max-gaps") par.orfMaxGaps = atoi(v);
        else if (f == "--contig-start-mode") par.contigStartMode = atoi(v);
        else if (f == "--contig-end-mode") par.contigEndMode = atoi(v);
        else if (f == "--orf-start-mode") par.orfStartMode = atoi(v);
        else if (f == "--forward-frames") par.forwardFrames = v;
        else if (f == "--reverse-frames") par.reverseFrames = v;
        else if (f == "--translation-table") par.translationTable = atoi(v);
        else if (f == "--translate") par.translate = atoi(v);
        else if (f == "--use-all-table-starts") par.useAllTableStarts = atoi(v) != 0;
        else if (f == "--create-lookup") par.createLookup = atoi(v);
        else if (f == "--add-orf-stop") par.addOrfStop = atoi(v) != 0;
        else if (f == "--sub-mat") subMat = v;                                   // nucleotide matrix file (proteinaln2nucl)
        else if (f == "--gap-open") par.gapOpen = MultiParam<int>(11, atoi(v));
        else if (f == "--gap-extend") par.gapExtend = MultiParam<int>(1, atoi(v));
        else if (f == "--simple-best-hit") par.simpleBestHit = atoi(v) != 0;
        else if (f == "--fdr") par.fdrCutoff = strtof(v, NULL);                   // a float in the reference (LocalParameters.h:57)
        else if (f == "--report-fdr") par.reportFdr = atoi(v);
        else if (f == "--fmt") par.formatType = atoi(v);
        else if (f == "--db-output") par.dbOut = atoi(v) != 0;
        else if (f == "--lca-ranks") par.lcaRanks = v;
        else if (f == "--tax-lineage") par.showTaxLineage = atoi(v);
        else if (f == "--threads") par.threads = atoi(v);
        else { fprintf(stderr, "ref_sets: unknown flag %s\n", f.c_str()); return 2; }
    }
    if (i != argc) { fprintf(stderr, "ref_sets: flag without a value\n"); return 2; }
    par.scoringMatrixFile = MultiParam<char *>("blosum62.out", subMat.c_str());
    omp_set_num_threads(std::max(1, par.threads));      // the modules' OpenMP regions index their readers / writers by thread (Parameters::parseParameters does this, Parameters.cpp:2111-2121)
    const Command command = {module.c_str(), it->second.first, NULL, 0, "", "", "", "", 0, {}};
    return it->second.first(0, NULL, command);
}
