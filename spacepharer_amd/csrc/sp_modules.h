// Module drivers: the reference's `prefilter`, `align`, `rescorediagonal` command lines on MMseqs2 DB triples.
#pragma once
#include <string>
namespace sp {
void setLastError(const std::string &e);
int currentDevice();
int runPrefilterModule(int argc, const char **argv);
int runAlignModule(int argc, const char **argv);
int runRescoreDiagonalModule(int argc, const char **argv);
int runSearchModule(int argc, const char **argv);
void releaseCachedContexts();
}  // namespace sp
