// Error reporting + device query entry points of the C-ABI (include/branchgan.h).
#include <stdarg.h>

#include "common.h"
#include "../build/build_id.h"

namespace bg {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace bg

extern "C" const char* bg_last_error(void) { return bg::g_err; }
extern "C" int bg_version(void) { return 2; }
extern "C" const char* bg_build_id(void) { return BG_BUILD_ID; }

extern "C" int bg_device_arch(char* buf, size_t buf_len) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
    bg::set_error("bg_device_arch: no HIP device");
    return BG_ERR_LAUNCH;
  }
  if (buf && buf_len) {
    strncpy(buf, prop.gcnArchName, buf_len - 1);
    buf[buf_len - 1] = 0;
  }
  return BG_OK;
}
