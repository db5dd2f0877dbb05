// emd_version / emd_last_error and the thread-local error buffer.
#include "emd_common.hpp"

namespace emd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace emd

extern "C" int emd_version(void) { return EMD_VERSION; }

extern "C" const char* emd_last_error(void) { return emd::g_err; }
