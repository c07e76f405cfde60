// Module drivers: the reference's `prefilter`, `align`, `rescorediagonal` command lines on MMseqs2 DB triples.
#pragma once
#include <string>
struct sp_ctx;
namespace sp {
void setLastError(const std::string &e);
int currentDevice();
int runPrefilterModule(int argc, const char **argv);
int runAlignModule(int argc, const char **argv);
int runRescoreDiagonalModule(int argc, const char **argv);
int runSearchModule(int argc, const char **argv);
void releaseCachedContexts();
struct AlignParams { std::string subMat; int gapOpen, gapExtend; double evalue; };
AlignParams alignParamsOf(const sp_ctx *c);      // what sp_ctx_align* of this context computes with
}  // namespace sp
