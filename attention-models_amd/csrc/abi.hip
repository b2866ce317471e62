// libamk.so: version / architecture / error-string entry points (include/amk.h).
#include "amk_common.h"

static thread_local char g_err[512] = "";

void amk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" int amk_version(void) { return AMK_VERSION; }
extern "C" const char* amk_arch(void) { return "gfx950"; }
extern "C" const char* amk_last_error(void) { return g_err; }
