// Library-wide pieces of libmkb_hip.so: ABI version and the thread-local error message.
#include "common.h"

namespace mkb {

static thread_local char g_err[512] = "";

int set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace mkb

extern "C" int mkb_abi_version(void) { return MKB_ABI_VERSION; }
extern "C" const char *mkb_last_error(void) { return mkb::g_err; }
