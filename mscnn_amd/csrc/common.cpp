#include "common.h"
#include <cstdlib>
#include <cstring>

namespace mscnn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
#ifdef MSCNN_TUNING_ENV
int tune_env(const char* name, int dflt) {
  const char* e = std::getenv(name);
  return e && *e ? std::atoi(e) : dflt;
}
#endif
}  // namespace mscnn

extern "C" const char* mscnn_last_error(void) { return mscnn::g_err; }
extern "C" const char* mscnn_version(void) { return "mscnn_hip 0.2 gfx950"; }
