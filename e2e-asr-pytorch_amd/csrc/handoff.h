// Inter-workgroup hand-off primitives of the persistent kernels (lstm_persist2.hip, decoder_persist.hip):
// data-tagged 8-byte granules, 16-byte polling sweeps, XCD placement consensus.  See cdna_hip_programming.md G16 form R2
// and DESIGN.md 4.2.  Everything here is `static` per translation unit.
#pragma once
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int SPIN_LIMIT2 = 1 << 22;

__device__ __forceinline__ u64 ld_gran(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// `global_store_dwordx2 sc0`: reaches the XCD's L2 and keeps the line there (an `sc1` store drops it), so a
// consumer on the SAME XCD is served by an L2 hit.  Only valid when producer and consumer share an XCD.
__device__ __forceinline__ void st_gran_local(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <bool LOCAL> __device__ __forceinline__ void publish(u64* p, u64 v) { if (LOCAL) st_gran_local(p, v); else st_gran(p, v); }

// Grid layout: workgroup i of the launch runs on XCD i % 8 under the dispatcher's round-robin placement, so the
// P workgroups of direction d are the ones with i % 8 == d (the others exit at once) and share one L2.
// Placement is never ASSUMED: every workgroup reads its XCC id, the ids are counted in one status word and only if all
// P workgroups of the direction report the same XCD the hand-off uses L2-local stores; otherwise write-through.
__device__ __forceinline__ bool xcd_consensus(u64* word, int P, int allow, unsigned* abort_flag) {
    __shared__ int s_local;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u;      // hwreg(HW_REG_XCC_ID, 0, 4)
        __hip_atomic_fetch_add(word, 1ull << (6 * xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int local = 0, spins = 0;
        while (true) {
            const u64 v = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int sum = 0, mx = 0;
            for (int k = 0; k < 8; ++k) { const int f = (int)((v >> (6 * k)) & 63u); sum += f; mx = max(mx, f); }
            if (sum >= P) { local = (mx == P) && allow; break; }
            if (++spins > SPIN_LIMIT2) { __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        // Every producer of the cluster has cleared its records by now.  One agent-scope release + acquire (`buffer_wbl2 sc1`,
        // `buffer_inv sc1`) before the first poll: whatever this XCD's L2 (and this CU's L1) still holds of EARLIER users of the
        // polled addresses - exchange granules of other launches on recycled allocator blocks, stored L2-locally by a cluster
        // that sat on this XCD - is written back / dropped, so the first polls are served by what the producers of THIS launch
        // wrote.  (Observed without it: about one full test-suite run in four accepted a few stale context granules in the
        // first decoder step of one utterance - logits off by 0.1, attention intact)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
        s_local = local;
        if ((blockIdx.x >> 3) == 0) reinterpret_cast<u64*>(abort_flag)[26 + (blockIdx.x & 7)] = (u64)local + 1;   // status: mode used (1 write-through, 2 XCD-local)
    }
    __syncthreads();
    return s_local != 0;
}
__device__ __forceinline__ unsigned seq_of(int s) { return (unsigned)((s >> 1) % 3) + 1u; }

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

// One polling round: CH `global_load_dwordx4 ... sc1` AND their `s_waitcnt vmcnt(0)` as ONE asm statement, for every CH.
// The destination registers are early-clobber outputs of that single statement, so for the compiler they come into
// existence already complete: no use, copy or spill of a destination can be scheduled between a load and the wait.
// (Round 2 had this form for CH <= 3 and 5 only; for the other widths the loads and the wait were separate statements and
// the tag checks were once scheduled in front of the wait, where the registers still held the previous poll - the
// intermittent decoder deviations of round 2, DESIGN.md section 2.  tests/test_handoff_isa.py disassembles the built
// objects and checks that nothing touches a poll destination between its load and the wait.)
// Inline asm because the loads must be re-issued every round; the gather waves have nothing else in their vector-memory
// queue, so `s_waitcnt vmcnt(0)` is exact.  2 * CH operands <= 20 of the 30 an asm statement may carry.
#define G16_LD(o, i) "global_load_dwordx4 %" #o ", %" #i ", off sc1\n\t"
template <int CH>
__device__ __forceinline__ void poll_round(const u64* const (&a)[CH], u32x4 (&v)[CH]) {
    static_assert(CH >= 1 && CH <= 10, "poll_round: add a case (one asm statement per width)");
    if constexpr (CH == 1) {
        asm volatile(G16_LD(0, 1) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0])
                     : "v"(a[0]) : "memory");
    } else if constexpr (CH == 2) {
        asm volatile(G16_LD(0, 2) G16_LD(1, 3) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1])
                     : "v"(a[0]), "v"(a[1]) : "memory");
    } else if constexpr (CH == 3) {
        asm volatile(G16_LD(0, 3) G16_LD(1, 4) G16_LD(2, 5) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]) : "memory");
    } else if constexpr (CH == 4) {
        asm volatile(G16_LD(0, 4) G16_LD(1, 5) G16_LD(2, 6) G16_LD(3, 7) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]) : "memory");
    } else if constexpr (CH == 5) {
        asm volatile(G16_LD(0, 5) G16_LD(1, 6) G16_LD(2, 7) G16_LD(3, 8) G16_LD(4, 9) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]) : "memory");
    } else if constexpr (CH == 6) {
        asm volatile(G16_LD(0, 6) G16_LD(1, 7) G16_LD(2, 8) G16_LD(3, 9) G16_LD(4, 10) G16_LD(5, 11) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]) : "memory");
    } else if constexpr (CH == 7) {
        asm volatile(G16_LD(0, 7) G16_LD(1, 8) G16_LD(2, 9) G16_LD(3, 10) G16_LD(4, 11) G16_LD(5, 12) G16_LD(6, 13) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]) : "memory");
    } else if constexpr (CH == 8) {
        asm volatile(G16_LD(0, 8) G16_LD(1, 9) G16_LD(2, 10) G16_LD(3, 11) G16_LD(4, 12) G16_LD(5, 13) G16_LD(6, 14) G16_LD(7, 15) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]) : "memory");
    } else if constexpr (CH == 9) {
        asm volatile(G16_LD(0, 9) G16_LD(1, 10) G16_LD(2, 11) G16_LD(3, 12) G16_LD(4, 13) G16_LD(5, 14) G16_LD(6, 15) G16_LD(7, 16) G16_LD(8, 17) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]) : "memory");
    } else if constexpr (CH == 10) {
        asm volatile(G16_LD(0, 10) G16_LD(1, 11) G16_LD(2, 12) G16_LD(3, 13) G16_LD(4, 14) G16_LD(5, 15) G16_LD(6, 16) G16_LD(7, 17) G16_LD(8, 18) G16_LD(9, 19) "s_waitcnt vmcnt(0)"
                     : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7]), "=&v"(v[8]), "=&v"(v[9])
                     : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]), "v"(a[8]), "v"(a[9]) : "memory");
    }
}
#undef G16_LD

template <int CH>
__device__ __forceinline__ bool poll_check(const u32x4 (&v)[CH], int n, u64 mask, u64 want, u64 (&lo)[CH], u64 (&hi)[CH]) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        lo[i] = (u64)v[i][0] | ((u64)v[i][1] << 32);
        hi[i] = (u64)v[i][2] | ((u64)v[i][3] << 32);
        ok = ok && ((i >= n) || (((lo[i] & mask) == want) && ((hi[i] & mask) == want)));
    }
    return ok;
}

// Polls the 16-byte granule PAIRS addr[0..n) (n <= CH, 16-byte aligned) until both halves of each carry `want` under
// `mask`.  One 16-byte load per pair: the bypass-load path of a CU moves ~10 B/clk however it is cut up, and 8-byte loads
// reach only 0.54-0.70x the 16-byte rate, so the sweep of the whole exchange vector - not the latency of one load - sets
// the length of a polling round.  Entries >= n re-read entry 0 and are not checked.
template <int CH>
__device__ __forceinline__ int gather16v(const u64* const (&addr)[CH], int n, u64 mask, u64 want, u64 (&lo)[CH], u64 (&hi)[CH],
                                         unsigned* abort_flag) {
    const u64* a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = addr[i < n ? i : 0];
    int spins = 0;
    while (true) {
        u32x4 v[CH];
        poll_round<CH>(a, v);
        if (poll_check<CH>(v, n, mask, want, lo, hi)) return spins;
        ++spins;
        if ((spins & 63) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return spins;
            if (spins > SPIN_LIMIT2) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return spins;
            }
        }
    }
}

// The same for the strided pairs base[0], base[stride], ... (u64 units).
template <int CH>
__device__ __forceinline__ int gather16(const u64* base, long stride, int n, u64 mask, u64 want, u64 (&lo)[CH], u64 (&hi)[CH],
                                        unsigned* abort_flag) {
    const u64* a[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) a[i] = base + (i < n ? i : 0) * stride;
    return gather16v<CH>(a, n, mask, want, lo, hi, abort_flag);
}

}  // namespace
