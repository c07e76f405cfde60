// Module drivers: the reference's `prefilter`, `align`, `rescorediagonal` command lines on MMseqs2 DB triples.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
struct sp_ctx;
namespace sp {
void setLastError(const std::string &e);
int currentDevice();
int runPrefilterModule(int argc, const char **argv);
int runAlignModule(int argc, const char **argv);
int runRescoreDiagonalModule(int argc, const char **argv);
int runSearchModule(int argc, const char **argv);
void releaseCachedContexts();
bool loadFlatForAbi(const std::string &path, std::vector<char> &res, std::vector<uint64_t> &off, std::vector<uint32_t> &keys, std::vector<uint32_t> &sets, std::string &err);
struct AlignParams { std::string subMat; int gapOpen, gapExtend; double evalue; };
AlignParams alignParamsOf(const sp_ctx *c);      // what sp_ctx_align* of this context computes with
}  // namespace sp
