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

// Fetches the 16-byte granule PAIRS base[0], base[stride], ... (n <= CH of them, base in u64 units and 16-byte aligned)
// until both halves of each carry `want` under `mask`.  One `global_load_dwordx4 sc1` per pair: the bypass-load path of a
// CU moves ~10 B/clk however it is cut up, and 8-byte loads reach only 0.54-0.70x the 16-byte rate, so the sweep of the
// whole exchange vector - not the latency of one load - sets the length of a polling round.  All CH loads of a round are
// issued back to back (entries >= n re-read entry 0 and are not checked).  Inline asm because the loads must be re-issued
// every round; the gather waves have nothing else in their vector-memory queue, so `s_waitcnt vmcnt(0)` is exact.
template <int CH>
__device__ __forceinline__ int gather16(const u64* base, long stride, int n, u64 mask, u64 want, u64 (&lo)[CH], u64 (&hi)[CH],
                                        unsigned* abort_flag) {
    int spins = 0;
    while (true) {
        u32x4 v[CH];
        // the loads of a round and their wait in ONE asm statement where CH allows: between separate statements hipcc was
        // seen to place an `s_waitcnt vmcnt(0)` of its own behind the first load (one extra round trip per round)
        if constexpr (CH == 1) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v[0]) : "v"(base) : "memory");
        } else if constexpr (CH == 2) {
            const u64* a1 = base + (1 < n ? 1 : 0) * stride;
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]) : "v"(base), "v"(a1) : "memory");
        } else if constexpr (CH == 3) {
            const u64* a1 = base + (1 < n ? 1 : 0) * stride;
            const u64* a2 = base + (2 < n ? 2 : 0) * stride;
            asm volatile("global_load_dwordx4 %0, %3, off sc1\n\tglobal_load_dwordx4 %1, %4, off sc1\n\tglobal_load_dwordx4 %2, %5, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]) : "v"(base), "v"(a1), "v"(a2) : "memory");
        } else if constexpr (CH == 5) {
            const u64* a1 = base + (1 < n ? 1 : 0) * stride;
            const u64* a2 = base + (2 < n ? 2 : 0) * stride;
            const u64* a3 = base + (3 < n ? 3 : 0) * stride;
            const u64* a4 = base + (4 < n ? 4 : 0) * stride;
            asm volatile("global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
                         "global_load_dwordx4 %3, %8, off sc1\n\tglobal_load_dwordx4 %4, %9, off sc1\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]) : "v"(base), "v"(a1), "v"(a2), "v"(a3), "v"(a4) : "memory");
        } else {
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                const u64* a = base + (i < n ? i : 0) * stride;
                asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(a) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // The destination registers of the loads above are only DEFINED for the compiler, not "complete": nothing tells it
            // that their contents arrive at the s_waitcnt.  It is free to schedule the tag checks below - plain VALU reads of
            // v[i] - in front of the wait statement, where the registers still hold the previous poll's data.  When that was the
            // previous STEP's granule with the same step tag (steps 2k and 2k+1 share one) the check passed on stale registers
            // and the not-yet-published slot (zeros) was consumed: the intermittent decoder deviations of round 2 (DESIGN.md
            // section 2).  An empty volatile asm that "modifies" each register pins every later use behind the wait.
            // WHY NOT ONE STATEMENT HERE TOO (round 3, measured): as a single asm with early-clobber outputs a 10-wide round
            // needs its 40 destination and 20 address registers live at once; dec_bwd_persist then spills 216 registers instead
            // of 41 (dec_fwd_persist: 126 -> 168 registers + scratch), the spill traffic shares the polling waves' in-order
            // vector-memory queue, and the step lost 2.4 ms (17.7 -> 20.1; A/B in one gpurun call, profiles/r03_handoff_ab.txt).
            // What the pins cannot exclude in principle - a register copy or spill of a destination placed between its load
            // and the wait - is excluded on the BUILT code instead: tests/test_handoff_isa.py disassembles the objects of every
            // kernel that polls and fails if any instruction touches a poll destination between its load and the wait.
#pragma unroll
            for (int i = 0; i < CH; ++i) asm volatile("" : "+v"(v[i]));
        }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            lo[i] = (u64)v[i][0] | ((u64)v[i][1] << 32);
            hi[i] = (u64)v[i][2] | ((u64)v[i][3] << 32);
            ok = ok && ((i >= n) || (((lo[i] & mask) == want) && ((hi[i] & mask) == want)));
        }
        if (ok) return spins;
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


// Same polling sweep with an explicit address per entry (entries >= n re-read entry 0 and are not checked).
template <int CH>
__device__ __forceinline__ int gather16v(const u64* const (&addr)[CH], int n, u64 mask, u64 want, u64 (&lo)[CH], u64 (&hi)[CH],
                                         unsigned* abort_flag) {
    int spins = 0;
    while (true) {
        u32x4 v[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const u64* a = addr[i < n ? i : 0];
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(a) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < CH; ++i) asm volatile("" : "+v"(v[i]));        // uses of v[i] stay behind the wait (see gather16)
        bool ok = true;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            lo[i] = (u64)v[i][0] | ((u64)v[i][1] << 32);
            hi[i] = (u64)v[i][2] | ((u64)v[i][3] << 32);
            ok = ok && ((i >= n) || (((lo[i] & mask) == want) && ((hi[i] & mask) == want)));
        }
        if (ok) return spins;
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


}  // namespace
