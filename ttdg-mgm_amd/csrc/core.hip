// Library-level entry points: version + thread-local error string.
#include "common.h"

thread_local char g_ttdg_err[512] = "";

extern "C" int ttdg_version(void) { return TTDG_VERSION; }
extern "C" const char* ttdg_last_error(void) { return g_ttdg_err; }
