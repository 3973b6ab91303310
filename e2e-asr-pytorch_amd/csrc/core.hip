// Error plumbing + identification entry points of libasr_hip.so.
#include "common.h"
#include "handoff.h"
#include <stdarg.h>
#include <algorithm>

static thread_local char g_err[512] = "";

void asr_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* asr_last_error(void) { return g_err; }
extern "C" int asr_version(void) { return 100; }
extern "C" const char* asr_device_arch(void) { return "gfx950"; }

// ---- test support ----------------------------------------------------------------------------------------------
// Holds `workgroups` compute units busy for `ticks` of the 100 MHz real-time counter: one 64-thread workgroup per CU
// that claims `lds_bytes` of LDS (the whole 160 KB makes it the only LDS-using resident of its CU) and sleeps until the
// deadline.  tests/test_persist_abort.py launches it on a second stream to starve a persistent launch of co-resident
// workgroups and expects the abort word + an exception, not numbers.  Bounded by construction: the loop ends at the
// deadline whatever happens around it.
namespace {
__global__ __launch_bounds__(64) void occupy_kernel(unsigned long long ticks, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hold[];
    if (threadIdx.x == 0) hold[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned n = 0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { __builtin_amdgcn_s_sleep(64); ++n; }
    if (sink && threadIdx.x == 0 && hold[0] == 0) sink[0] = n;      // keeps the LDS claim and the loop alive
}
}  // namespace

extern "C" int asr_debug_occupy(int workgroups, int lds_bytes, double seconds, asr_stream_t stream) {
    ASR_REQUIRE(workgroups > 0 && workgroups <= 1024 && lds_bytes >= 0 && lds_bytes <= 160 * 1024, ASR_E_ARG, "asr_debug_occupy: bad args");
    ASR_REQUIRE(seconds > 0.0 && seconds <= 20.0, ASR_E_ARG, "asr_debug_occupy: at most 20 s");
    hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(64), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)(seconds * 1e8), (unsigned*)nullptr);
    ASR_LAUNCH_CHECK("asr_debug_occupy");
    return ASR_OK;
}


// ---- CU-masked streams (include/asr_hip.h) -----------------------------------------------------------------------
extern "C" int asr_stream_create_cu_mask(int first, int count, asr_stream_t* stream) {
    ASR_REQUIRE(stream && first >= 0 && count > 0 && first + count <= 32, ASR_E_ARG, "asr_stream_create_cu_mask: units [first, first+count) must lie in 0..31");
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int u = first; u < first + count; ++u)
        for (int x = 0; x < 8; ++x) { const int bit = 8 * u + x; mask[bit >> 5] |= 1u << (bit & 31); }
    hipStream_t st = nullptr;
    const hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask);
    ASR_REQUIRE(e == hipSuccess, ASR_E_LAUNCH, "asr_stream_create_cu_mask: hipExtStreamCreateWithCUMask failed: %s", hipGetErrorString(e));
    *stream = (asr_stream_t)st;
    return ASR_OK;
}

extern "C" int asr_stream_destroy(asr_stream_t stream) {
    ASR_REQUIRE(stream, ASR_E_ARG, "asr_stream_destroy: null stream");
    const hipError_t e = hipStreamDestroy((hipStream_t)stream);
    ASR_REQUIRE(e == hipSuccess, ASR_E_LAUNCH, "asr_stream_destroy: %s", hipGetErrorString(e));
    return ASR_OK;
}


// ---- scrubbing a hand-off work area --------------------------------------------------------------------------------
// Exchange granules are stored L2-locally (`sc0`) by the persistent kernels and can outlive the launch that wrote them:
// in the L2 of the XCD that wrote them, and - when such a dirty line is evicted late - in HBM again, on top of a later zero
// fill.  Hand-off areas therefore live in a pool of their own that never goes back to the tensor allocator
// (src/hipabi.handoff_acquire), and an area is scrubbed whenever it changes hands: EVERY XCD writes zeros over the whole
// range with the same L2-local stores, so each L2 holds - and will evict - zeros only.
// Nothing is assumed about which workgroup runs on which XCD: a workgroup reads its XCC_ID and draws chunk numbers from
// THAT XCD's ticket counter until the counter has passed the last chunk, so any one workgroup that lands on an XCD covers the
// whole range for it.  The workgroup that finishes last checks that all eight counters did pass the last chunk; an XCD
// that received no workgroup at all (never seen; the grid has >= 64 workgroups) sets tickets[15], an abort word the caller
// registers like those of the persistent launches.
namespace {
__global__ __launch_bounds__(256) void scrub_kernel(unsigned long long* base, long n, long chunk, unsigned nchunk, unsigned* tickets) {
    __shared__ unsigned s_c;
    const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // hwreg(HW_REG_XCC_ID, 0, 4)
    while (true) {
        if (threadIdx.x == 0) s_c = __hip_atomic_fetch_add(&tickets[xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned c = s_c;
        __syncthreads();
        if (c >= nchunk) break;                                  // every chunk of this XCD has been handed out
        const long c0 = (long)c * chunk;
        for (long i = c0 + threadIdx.x; i < c0 + chunk && i < n; i += blockDim.x) st_gran_local(base + i, 0ull);
    }
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(&tickets[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (done == gridDim.x) {
            bool ok = true;
            for (int x = 0; x < 8; ++x) ok = ok && __hip_atomic_load(&tickets[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= nchunk;
            if (!ok) __hip_atomic_store(&tickets[15], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
}  // namespace

extern "C" int asr_scrub_workspace(void* ptr, size_t bytes, void* tickets, asr_stream_t stream) {
    ASR_REQUIRE(ptr && ((uintptr_t)ptr & 7) == 0, ASR_E_ARG, "asr_scrub_workspace: null or unaligned pointer");
    ASR_REQUIRE(tickets && ((uintptr_t)tickets & 3) == 0, ASR_E_ARG, "asr_scrub_workspace: tickets = 64 bytes of device memory owned by the calling stream");
    const long n = (long)(bytes / 8);
    if (n == 0) return ASR_OK;
    const long chunk = 4096;                                  // granules per ticket
    const long nchunk = (n + chunk - 1) / chunk;
    ASR_REQUIRE(nchunk <= 0x3fffffffL, ASR_E_ARG, "asr_scrub_workspace: range too large");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(tickets, 0, 9 * sizeof(unsigned), st);     // counters + census; the abort word (tickets[15]) is the caller's to clear
    const long grid = std::max(64L, std::min(8 * nchunk, 1024L));
    hipLaunchKernelGGL(scrub_kernel, dim3((unsigned)grid), dim3(256), 0, st, (unsigned long long*)ptr, n, chunk, (unsigned)nchunk, (unsigned*)tickets);
    ASR_LAUNCH_CHECK("asr_scrub_workspace");
    return ASR_OK;
}
