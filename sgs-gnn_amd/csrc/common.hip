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
static const int64_t* g_dyn_edges = nullptr;
const int64_t* dyn_edges_ptr() { return g_dyn_edges; }

// ---------------------------------------------------------------- batch staging (sgs_stage_segments)
// One launch copies up to kMaxSeg (source -> destination) byte spans and fills each destination's tail with a 32-bit word.
// HBM-bound streaming copy: 16 bytes per lane per step, a workgroup per 16 KiB chunk; segments are found by a linear scan
// of the (by-value) descriptor table's chunk prefix.
constexpr int kMaxSeg = 24;
constexpr int kStageChunk = 16384;
struct StageArgs {
    const char* src[kMaxSeg];
    char* dst[kMaxSeg];
    int64_t src_bytes[kMaxSeg];
    int64_t dst_bytes[kMaxSeg];
    uint32_t pad[kMaxSeg];
    int chunk0[kMaxSeg + 1];     // first chunk of each segment
    int n;
    int64_t* dims;               // optional: dims[0..n_dims) = dims_value[..]
    int64_t dims_value[4];
    int n_dims;
};
namespace {
__global__ void __launch_bounds__(256) stage_segments_kernel(const StageArgs a) {
    if (blockIdx.x == 0 && threadIdx.x < a.n_dims && a.dims) a.dims[threadIdx.x] = a.dims_value[threadIdx.x];
    int seg = 0;
    while (seg + 1 < a.n && static_cast<int>(blockIdx.x) >= a.chunk0[seg + 1]) ++seg;
    const int64_t off0 = static_cast<int64_t>(static_cast<int>(blockIdx.x) - a.chunk0[seg]) * kStageChunk;
    const char* __restrict__ src = a.src[seg];
    char* __restrict__ dst = a.dst[seg];
    const int64_t sb = a.src_bytes[seg], db = a.dst_bytes[seg];
    const uint32_t pw = a.pad[seg];
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < kStageChunk / (256 * 16); ++it) {
        const int64_t o = off0 + (static_cast<int64_t>(it) * 256 + threadIdx.x) * 16;
        if (o >= db) break;
        if (vec && o + 16 <= sb) {
            *reinterpret_cast<uint4*>(dst + o) = *reinterpret_cast<const uint4*>(src + o);
        } else {
            // span end / unaligned: 4-byte words (every span is a whole number of words)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int64_t ow = o + 4 * w;
                if (ow < db) *reinterpret_cast<uint32_t*>(dst + ow) = (ow < sb) ? *reinterpret_cast<const uint32_t*>(src + ow) : pw;
            }
        }
    }
}
}  // namespace
}  // namespace sgs

namespace sgs { void set_epoch_ptr(const uint64_t* p); }

extern "C" {
int sgs_rng_set_epoch_buffer(const uint64_t* epoch_dev) { sgs::set_epoch_ptr(epoch_dev); return SGS_OK; }
int sgs_dyn_edges_set(const int64_t* n_edges_dev) { sgs::g_dyn_edges = n_edges_dev; return SGS_OK; }
int sgs_stage_max_segments(void) { return sgs::kMaxSeg; }
int sgs_stage_segments(const int64_t* desc_host, int64_t n_segments, int64_t* dims_dev, const int64_t* dims_host, int64_t n_dims,
                       sgs_stream_t stream) {
    using namespace sgs;
    SGS_REQUIRE(n_segments >= 0 && n_segments <= kMaxSeg && n_dims >= 0 && n_dims <= 4 && (n_segments == 0 || desc_host) &&
                    (n_dims == 0 || (dims_dev && dims_host)), SGS_EINVAL, "sgs_stage_segments: bad arguments");
    StageArgs a{};
    int chunks = 0;
    int k = 0;
    for (int64_t i = 0; i < n_segments; ++i) {
        const int64_t* d = desc_host + 5 * i;
        const int64_t sb = d[2], db = d[3];
        SGS_REQUIRE(sb >= 0 && db >= sb && (sb & 3) == 0 && (db & 3) == 0 && ((d[0] | d[1]) & 3) == 0, SGS_EINVAL,
                    "sgs_stage_segments: segment %lld: spans must be 4-byte aligned whole words with dst_bytes >= src_bytes", (long long)i);
        if (db == 0) continue;
        SGS_REQUIRE(d[1] != 0 && (sb == 0 || d[0] != 0), SGS_EINVAL, "sgs_stage_segments: segment %lld: null pointer", (long long)i);
        a.src[k] = reinterpret_cast<const char*>(d[0]);
        a.dst[k] = reinterpret_cast<char*>(d[1]);
        a.src_bytes[k] = sb; a.dst_bytes[k] = db; a.pad[k] = static_cast<uint32_t>(d[4]);
        a.chunk0[k] = chunks;
        chunks += static_cast<int>((db + kStageChunk - 1) / kStageChunk);
        ++k;
    }
    a.chunk0[k] = chunks;
    a.n = k;
    a.dims = dims_dev;
    a.n_dims = static_cast<int>(n_dims);
    for (int64_t i = 0; i < n_dims; ++i) a.dims_value[i] = dims_host[i];
    if (chunks == 0) {
        if (n_dims == 0) return SGS_OK;
        chunks = 1; a.n = 1; a.chunk0[1] = 1;          // dims only: one workgroup, an empty span
        a.dst_bytes[0] = 0; a.src_bytes[0] = 0;
    }
    hipLaunchKernelGGL(stage_segments_kernel, dim3(static_cast<unsigned>(chunks)), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    SGS_LAUNCH_OK();
    return SGS_OK;
}
int sgs_abi_version(void) { return SGS_ABI_VERSION; }
const char* sgs_last_error(void) { return sgs::g_err; }
}
