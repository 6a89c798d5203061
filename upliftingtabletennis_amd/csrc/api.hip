// Library-level entry points: version, error string, device probe.
#include "common.h"

namespace ttup {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace ttup

extern "C" int ttup_version(void) { return 100; }
extern "C" const char* ttup_last_error(void) { return ttup::get_error(); }
extern "C" int ttup_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { ttup::set_error("hipGetDeviceCount failed"); return -1; }
    return n;
}
