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
static const uint64_t* g_epoch = nullptr;
const uint64_t* epoch_ptr() { return g_epoch; }
void set_epoch_ptr(const uint64_t* p) { g_epoch = p; }
}  // namespace sgs

namespace sgs { void set_epoch_ptr(const uint64_t* p); }

extern "C" {
int sgs_rng_set_epoch_buffer(const uint64_t* epoch_dev) { sgs::set_epoch_ptr(epoch_dev); return SGS_OK; }
int sgs_abi_version(void) { return SGS_ABI_VERSION; }
const char* sgs_last_error(void) { return sgs::g_err; }
}
