#include "common.h"
#include <cstring>

namespace mscnn {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace mscnn

extern "C" const char* mscnn_last_error(void) { return mscnn::g_err; }
extern "C" const char* mscnn_version(void) { return "mscnn_hip 0.1 gfx950"; }
