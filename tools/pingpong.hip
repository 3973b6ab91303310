// Micro-benchmark: granule ping-pong latency between two workgroups (one wave each polls/stores) on gfx950.
// Variants: producer store flavour (sc1 write-through vs plain) x placement (same XCD / different XCD).
// Build: hipcc --offload-arch=gfx950 -O3 tools/pingpong.hip -o gpurun_out/pingpong ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned long long u64;

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }

template <bool PLAIN>
__device__ __forceinline__ void st(u64* p, u64 v) {
    if (PLAIN) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid = 64 workgroups; roles are picked from the XCC id table built at start: A = first WG on xcd xa, B = first WG on xcd xb (B != A)
template <bool PLAIN>
__global__ void pingpong(u64* buf, unsigned* xtab, int* roles, int iters, u64* out, unsigned* fail) {
    const int wg = blockIdx.x;
    if (threadIdx.x == 0) { __hip_atomic_store(&xtab[wg], xcc_id() + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    const int a = roles[0], b = roles[1];
    if (wg != a && wg != b) return;
    const int lane = threadIdx.x;          // 64 lanes: each lane owns one granule (a 512-B edge)
    u64* mine = buf + (wg == a ? 0 : 4096) + lane;
    u64* other = buf + (wg == a ? 4096 : 0) + lane;
    u64 t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 17) t0 = __builtin_amdgcn_s_memrealtime();
        if (wg == a) st<PLAIN>(mine + (i & 1) * 64, ((u64)i << 32) | lane);
        int spins = 0;
        while (true) {
            u64 g = ld(other + (i & 1) * 64);
            if ((unsigned)(g >> 32) == (unsigned)i) break;
            if (++spins > (1 << 20)) { if (lane == 0) *fail = 1; return; }
        }
        if (wg == b) st<PLAIN>(mine + (i & 1) * 64, ((u64)i << 32) | lane);
    }
    u64 t1 = __builtin_amdgcn_s_memrealtime();
    if (wg == a && lane == 0) { out[0] = t1 - t0; out[1] = xcc_id(); }
    if (wg == b && lane == 0) { out[2] = xcc_id(); }
}

__global__ void census(unsigned* xtab) { if (threadIdx.x == 0) xtab[blockIdx.x] = xcc_id() + 1; }

int main() {
    u64 *buf, *out; unsigned *xtab, *fail; int* roles;
    hipMalloc(&buf, 8192 * 8 * 2); hipMalloc(&out, 64); hipMalloc(&xtab, 64 * 4); hipMalloc(&fail, 4); hipMalloc(&roles, 8);
    unsigned hx[64];
    hipMemset(xtab, 0, 256);
    hipLaunchKernelGGL(census, dim3(64), dim3(64), 0, 0, xtab);
    hipMemcpy(hx, xtab, 256, hipMemcpyDeviceToHost);
    printf("xcc of workgroups 0..63:"); for (int i = 0; i < 64; ++i) printf(" %u", hx[i] - 1); printf("\n");
    const int iters = 2000;
    for (int variant = 0; variant < 4; ++variant) {
        const bool plain = variant & 1, same = variant & 2;
        int r[2] = {0, same ? 8 : 1};      // under round-robin placement wg 0 and 8 share an XCD, 0 and 1 do not
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(buf, 0, 8192 * 16); hipMemset(fail, 0, 4); hipMemset(out, 0, 64);
            hipMemcpy(roles, r, 8, hipMemcpyHostToDevice);
            if (plain) hipLaunchKernelGGL(pingpong<true>, dim3(64), dim3(64), 0, 0, buf, xtab, roles, iters, out, fail);
            else hipLaunchKernelGGL(pingpong<false>, dim3(64), dim3(64), 0, 0, buf, xtab, roles, iters, out, fail);
            u64 ho[8]; unsigned hf;
            hipDeviceSynchronize();
            hipMemcpy(ho, out, 64, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost);
            printf("store=%s placement=%s (xcc %llu <-> %llu): round trip %.3f us (one hop %.3f us)%s\n", plain ? "plain" : "sc1  ",
                   same ? "same-xcd " : "cross-xcd", ho[1], ho[2], ho[0] * 0.01 / (iters - 16), ho[0] * 0.005 / (iters - 16),
                   hf ? "  ** TIMEOUT (stale) **" : "");
        }
    }
    return 0;
}
