#include "sp_modules.h"
namespace sp {
int runPrefilterModule(int, const char **) { setLastError("not implemented"); return 1; }
int runAlignModule(int, const char **) { setLastError("not implemented"); return 1; }
int runRescoreDiagonalModule(int, const char **) { setLastError("not implemented"); return 1; }
}
