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
namespace {
__global__ void zero_words(uint32_t* __restrict__ p, size_t n) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
__global__ void fill2_words(uint32_t* __restrict__ p1, size_t n1, uint32_t v1, uint32_t* __restrict__ p2, size_t n2, uint32_t v2) {
    const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n1) p1[i] = v1;
    else if (i - n1 < n2) p2[i - n1] = v2;
}
}  // namespace
int fill2_async(void* p1, size_t bytes1, uint32_t v1, void* p2, size_t bytes2, uint32_t v2, hipStream_t stream) {
    if (((bytes1 | bytes2) & 3) || ((reinterpret_cast<uintptr_t>(p1) | reinterpret_cast<uintptr_t>(p2)) & 3)) {
        set_error("fill2_async: unaligned span");
        return SGS_EINVAL;
    }
    const size_t n1 = bytes1 / 4, n2 = bytes2 / 4, n = n1 + n2;
    if (n == 0) return SGS_OK;
    hipLaunchKernelGGL(fill2_words, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, static_cast<uint32_t*>(p1), n1, v1,
                       static_cast<uint32_t*>(p2), n2, v2);
    return SGS_OK;
}
int zero_async(void* p, size_t bytes, hipStream_t stream) {
    if (bytes == 0) return SGS_OK;
    if ((bytes & 3) || (reinterpret_cast<uintptr_t>(p) & 3)) { set_error("zero_async: unaligned span"); return SGS_EINVAL; }
    const size_t n = bytes / 4;
    hipLaunchKernelGGL(zero_words, dim3(static_cast<unsigned>((n + 255) / 256)), dim3(256), 0, stream, static_cast<uint32_t*>(p), n);
    return SGS_OK;
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
