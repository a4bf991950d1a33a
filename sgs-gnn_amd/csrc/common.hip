// Error channel and ABI version of libsgs_hip.so.
#include "sgs_common.h"

namespace sgs {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace sgs

extern "C" {
int sgs_abi_version(void) { return SGS_ABI_VERSION; }
const char* sgs_last_error(void) { return sgs::g_err; }
}
