// Front-end binary: `spacepharer-amd <module> <args...>` with module in {prefilter, align, rescorediagonal} and the set-level
// steps around them (extractorfs ... summarizeresults).
// Point $MMSEQS at a wrapper that sends these three module names here and every other one to the stock
// spacepharer binary, and the unmodified workflow scripts (lib/mmseqs/data/workflow/blastp.sh:33,41,48) land
// in the MI355X implementation.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/spacepharer_amd.h"

int main(int argc, const char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <prefilter|align|rescorediagonal|search|extractorfs|translatenucs|reverseseqbycodon|proteinaln2nucl|combineprotnuclaln|besthitperset|mergeresultsbyset|combinescore|filtermatchbyfdr|summarizeresults> <args>\n", argv[0]); return 1; }
    int rc;
    // the set-level steps are host-only: no device is touched for them
    struct HostModule { const char *name; int (*run)(int, const char **); };
    static const HostModule hostModules[] = {{"extractorfs", sp_extractorfs_run}, {"translatenucs", sp_translatenucs_run}, {"reverseseqbycodon", sp_reverseseqbycodon_run}, {"proteinaln2nucl", sp_proteinaln2nucl_run}, {"combineprotnuclaln", sp_combineprotnuclaln_run}, {"besthitperset", sp_besthitperset_run}, {"mergeresultsbyset", sp_mergeresultsbyset_run}, {"combinescore", sp_combinescore_run}, {"filtermatchbyfdr", sp_filtermatchbyfdr_run}, {"summarizeresults", sp_summarizeresults_run}};
    for (const HostModule &m : hostModules) if (!strcmp(argv[1], m.name)) { rc = m.run(argc - 2, argv + 2); if (rc != 0) fprintf(stderr, "%s\n", sp_last_error()); return rc; }
    // SP_GPUS=N: the module starts one worker process per GPU (this binary again, with SP_PART set); the parent only merges
    // the parts and has no use for a device
    const char *gpus = getenv("SP_GPUS");
    const bool fanOut = gpus && atoi(gpus) > 1;
    if (!fanOut && sp_init(0) != SP_OK) { fprintf(stderr, "%s\n", sp_last_error()); return 1; }
    if (!strcmp(argv[1], "prefilter")) rc = sp_prefilter_run(argc - 2, argv + 2);
    else if (!strcmp(argv[1], "align")) rc = sp_align_run(argc - 2, argv + 2);
    else if (!strcmp(argv[1], "rescorediagonal")) rc = sp_rescorediagonal_run(argc - 2, argv + 2);
    else if (!strcmp(argv[1], "search")) rc = sp_search_run(argc - 2, argv + 2);
    else { fprintf(stderr, "unknown module %s\n", argv[1]); return 1; }
    if (rc != 0) fprintf(stderr, "%s\n", sp_last_error());
    sp_shutdown();
    return rc;
}
