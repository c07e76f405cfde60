// TEST INFRASTRUCTURE ONLY -- never linked into or called by the product.
//
// Forced include (-include) for the reference's OWN module translation units (extractorfs.cpp, translatenucs.cpp,
// ReverseSeqByCodon.cpp, proteinaln2nucl.cpp, CombineProtNuclAln.cpp, besthitperset.cpp, mergeresultsbyset.cpp,
// CombineScore.cpp, FilterMatchbyFdr.cpp, SummarizeResults.cpp), which oracle/Makefile compiles where they lie under
// /root/reference, unmodified.
//
// Every one of those modules starts with
//     Parameters &par = Parameters::getInstance();  par.parseParameters(argc, argv, command, true, 0, 0);
// and Parameters.cpp cannot be built here (it includes headers the reference's cmake build generates from data files).
// Nothing of Parameters.cpp is re-typed or stood in for.  Instead the two NAMES `Parameters` / `LocalParameters` are
// redirected, for the module bodies only, to the plain record below: the reference headers are included first under their
// real names (their include guards make the modules' own #include lines no-ops), then the names are #define'd.  The record
// holds the handful of fields the ten modules read; oracle/ref_sets.cpp fills them from its own command line, which is
// what parseParameters would have done.  Everything else the modules execute -- their bodies, DBReader / DBWriter, Orf,
// TranslateNucl, Matcher, NucleotideMatrix, EvalueComputation + ALP, Aggregation, NcbiTaxonomy, Util -- is the
// reference's code, compiled in place.
#ifndef SP_REF_PAR_H
#define SP_REF_PAR_H

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <string>
#include <vector>
#include <unistd.h>

#include "Aggregation.h"
#include "Command.h"
#include "DBReader.h"
#include "DBWriter.h"
#include "Debug.h"
#include "FastSort.h"
#include "FileUtil.h"
#include "LocalParameters.h"
#include "Matcher.h"
#include "MultiParam.h"
#include "NcbiTaxonomy.h"
#include "Orf.h"
#include "Parameters.h"
#include "TranslateNucl.h"
#include "Util.h"
#include "itoa.h"

struct RefPar {
    // constants the module bodies name through the class: the reference's values, re-exported
    static const int DBTYPE_AMINO_ACIDS = ::Parameters::DBTYPE_AMINO_ACIDS;
    static const int DBTYPE_NUCLEOTIDES = ::Parameters::DBTYPE_NUCLEOTIDES;
    static const int DBTYPE_ALIGNMENT_RES = ::Parameters::DBTYPE_ALIGNMENT_RES;
    static const int DBTYPE_GENERIC_DB = ::Parameters::DBTYPE_GENERIC_DB;
    static const int DBTYPE_OMIT_FILE = ::Parameters::DBTYPE_OMIT_FILE;
    static const int FORMAT_TYPE_SHORT = ::LocalParameters::FORMAT_TYPE_SHORT;
    static const int FORMAT_TYPE_LONG = ::LocalParameters::FORMAT_TYPE_LONG;
    static const int FORMAT_TYPE_ALN = ::LocalParameters::FORMAT_TYPE_ALN;
    static bool isEqualDbtype(const int a, const int b) { return ::Parameters::isEqualDbtype(a, b); }

    // positional DB names as parseParameters leaves them (<name>, <name>.index, <name>_h, <name>_h.index)
    std::string db1, db1Index, hdr1, hdr1Index, db2, db2Index, hdr2, hdr2Index, db3, db3Index, db4, db4Index, db5, db5Index, db6, db6Index;
    int threads = 1;
    int compressed = 0;
    size_t maxSeqLen = 65535;
    // extractorfs / translatenucs (the reference's defaults, Parameters.cpp:2219-2248; the driver overrides)
    int orfMinLength = 30, orfMaxLength = 32734, orfMaxGaps = INT_MAX, contigStartMode = 2, contigEndMode = 2, orfStartMode = 1;
    std::string forwardFrames = "1,2,3", reverseFrames = "1,2,3";
    bool useAllTableStarts = false;
    int translate = 0, translationTable = 1, createLookup = 0;
    bool addOrfStop = false;
    // proteinaln2nucl
    MultiParam<char *> scoringMatrixFile = MultiParam<char *>("blosum62.out", "nucleotide.out");
    MultiParam<int> gapOpen = MultiParam<int>(11, 5), gapExtend = MultiParam<int>(1, 2);
    // besthitperset
    bool simpleBestHit = true;
    // filtermatchbyfdr / summarizeresults (LocalParameters.h:165-173)
    float fdrCutoff = 0.05f;
    int reportFdr = 0, formatType = 1, showTaxLineage = 0;
    bool dbOut = false;
    std::string lcaRanks;

    template <typename... A> void parseParameters(A &&...) {}       // the driver has filled the record
    static RefPar &getInstance() { static RefPar p; return p; }
    static RefPar &getLocalInstance() { return getInstance(); }
};

#define Parameters RefPar
#define LocalParameters RefPar

#endif
