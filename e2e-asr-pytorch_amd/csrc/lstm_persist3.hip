// Third-generation persistent LSTM recurrence (bf16 contraction mode, H % 16 == 0, H <= 512): the encoder's
// working point.  One launch walks all T steps of both directions (nn.LSTM recurrence + its BPTT, reference
// src/module.py:1023,1049) for up to 16 * (8 / ND) utterances.
//
// What changed against lstm_persist2.hip:
//  * BATCH-SLICED GROUPS, ONE PER XCD.  The recurrences of different utterances are independent, and a step is bound by
//    what one workgroup has to take in from its peers (the whole h_{t-1} of its batch rows, ~10 B/clk per CU on the
//    bypass-load path), not by arithmetic.  So the batch is cut into NS = 8/ND slices and the launch runs 8 independent
//    groups (direction x slice) of P = H/16 workgroups, group g on the workgroups with blockIdx % 8 == g (one XCD under
//    round-robin dispatch; verified at run time by the XCC-id consensus, never assumed).  At B = 16 a workgroup polls
//    2.5 KB per step instead of 10 KB, 160 of the 256 compute units work instead of 40, and B <= 64 fits one launch.
//  * GATE-MINOR STORAGE.  Pre-activations / activated gates / gate gradients are stored (B,T,ND,H,4) = [unit][i,f,g,o]
//    (the input-projection contraction writes that order because the bf16 copy of W_ih is stored row-permuted,
//    elementwise.hip: asr_lstm_pack_weights).  With the MFMA rows ordered unit*4+gate a lane's four accumulator
//    registers are the four gates of ONE (unit, batch row): the cell update is lane-local, the LDS gate rendez-vous and
//    the second barrier of a step are gone, and every bulk access of a lane is one 8-byte word.
//  * bf16 STORAGE of everything but the cell state: gates, h (y) and dy are bf16 in HBM (half the bytes of v2);
//    y is written with one zero row of padding on both sides in time ((B,T+2,ND*H), frame t at row t+1), which lets
//    the recurrent weight gradient read h_{t-1} / h_{t+1} as plain shifted rows.
//  * the launch epoch of the tags is extended by rotating the exchange region of a caller-owned, persistent workspace
//    (see asr_hip.h): a (region, tag) pair repeats only every 32 (forward) / 32 (backward) launches of that workspace.
// Hand-off primitives, bounded spins and the abort word: handoff.h (guide form R2).
#include "common.h"
#include <stdlib.h>
#include "handoff.h"

namespace {

#ifdef ASR_DIAG
#define DIAG3_DECL unsigned long long dg_t = __builtin_amdgcn_s_memrealtime(), dg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DIAG3_MARK(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); dg_acc[k] += n_ - dg_t; dg_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define DIAG3_DUMP(thr, word) { if (blockIdx.x == 0 && threadIdx.x == (thr)) { unsigned long long* o = (unsigned long long*)p.abort_flag + (word); for (int k = 0; k < 8; ++k) o[k] = dg_acc[k]; } }
#elif defined(ASR_JITTER)
// race-detector build (see decoder_persist.hip): a pseudo-random sleep at every phase boundary of every wave
__device__ __forceinline__ void jitter3(unsigned k, unsigned step, unsigned epoch) {
    unsigned h = (blockIdx.x * 0x9E3779B1u) ^ ((threadIdx.x >> 6) * 0x85EBCA6Bu) ^ (k * 0xC2B2AE35u) ^ (step * 0x27D4EB2Fu) ^ (epoch * 0x165667B1u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    h = __builtin_amdgcn_readfirstlane(h);
    if ((h & 7u) == 0u) {
        const unsigned n = (h >> 3) & 31u;
        for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }
}
#define DIAG3_DECL
#define DIAG3_MARK(k) jitter3(k, (unsigned)s, p.epoch);
#define DIAG3_DUMP(thr, word)
#else
#define DIAG3_DECL
#define DIAG3_MARK(k)
#define DIAG3_DUMP(thr, word)
#endif

constexpr int HDR_BYTES = 1024;         // status block: abort word, per-group consensus words (u64 8..15), modes (26..33), diag (64..)
// TWO status blocks per workspace, used by launch epoch parity: a launch works in block (epoch & 1) and clears the OTHER one for
// its successor (first workgroup, before anything else) - the per-launch `hipMemsetAsync` of the header is gone (eight fill
// launches per training step on the critical stream).  A fresh / scrubbed workspace is all zero.
constexpr int HDR_SLOTS = 2;
constexpr int FWD_REGIONS = 8, BWD_REGIONS = 2;

struct P3 {
    unsigned short* gates;   // (B,T,ND,H,4) bf16
    const float* whh;        // (ND,4H,H) fp32, reference row order [gate][unit]
    unsigned short* y;       // fwd: h out, padded (B,T+2,ND*H) bf16;  bwd: dy in, (B,T,ND*H) bf16
    float* c;                // (B,T,ND,H)
    u64* xbuf;               // exchange region of this launch
    unsigned* abort_flag;
    unsigned* clear_next;    // status block of the next launch on this workspace: cleared by workgroup 0
    int B, T, H, ND, P, NS, BS;
    int allow_local, poll_delay;
    unsigned epoch;
    unsigned char* dump;     // backward: 8 KB per workgroup where the memory operations of inactive lanes land (see lstm_bwd_p3)
    unsigned region_bytes;   // exchange region + dump area (buffer descriptor of the publishes)
};

__device__ __forceinline__ float fast_sigmoid3(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh3(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float bf2f(unsigned short v) { return __uint_as_float((unsigned)v << 16); }

// forward tag: bit 14 of each of the four bf16 values of a granule (0 for every |h| <= 1): step sequence in elements 0,1,
// launch epoch in elements 2,3
constexpr u64 FWD3_MASK = (1ull << 14) | (1ull << 30) | (1ull << 46) | (1ull << 62);
__device__ __forceinline__ u64 fwd3_want(unsigned seq, unsigned epoch) {
    return ((u64)(seq & 1u) << 14) | ((u64)(seq >> 1) << 30) | ((u64)(epoch & 1u) << 46) | ((u64)((epoch >> 1) & 1u) << 62);
}
// backward tag: three mantissa LSBs of both floats of a granule = 2-bit step sequence + 4-bit launch epoch
constexpr u64 BWD3_MASK = 7ull | (7ull << 32);
__device__ __forceinline__ u64 bwd3_want(unsigned seq, unsigned epoch) {
    epoch = 2u + epoch % 14u;             // epoch field 2..15: non-zero tag bits in BOTH words (see pair_want, decoder_persist.hip)
    const unsigned tag = ((epoch & 15u) << 2) | seq;
    return (u64)(tag & 7u) | ((u64)(tag >> 3) << 32);
}

// two adjacent granules (16-byte aligned pair) in ONE 16-byte store: `sc0` keeps the line in this XCD's L2 (consumers on the
// same XCD), `sc1` writes it through (any placement).  A 16-byte store is one fabric write like an 8-byte one.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_p3;
// (a buffer store the compiler counts: with hidden inline-asm stores in the queue its counted `s_waitcnt vmcnt(N)` for the
// operand prefetch waited for the publishes just issued - 0.45 us per step)
__device__ __forceinline__ void publish_pair(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off, u64 v0, u64 v1, bool local) {
    const u32x4_p3 v = {(unsigned)v0, (unsigned)(v0 >> 32), (unsigned)v1, (unsigned)(v1 >> 32)};
    if (local) __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, 1);        // sc0
    else __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, byte_off, 0, 16);             // sc1
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup pw of group (d, slice): hidden units u0 = 16 pw .. +15 for batch rows b0 .. b0+nb-1 (nb <= 16).
//   waves 0-3 (compute): wave w owns units u0+4w .. +3, ALL four gates.  MFMA D[row = 4*unit + gate][col = b]
//                        = sum_k W_hh[gate*H + u0+4w+unit][k] * h_{t-1}[b][k]; lane (n = b, q = unit) holds the four
//                        gates of its unit in the four accumulator registers -> cell update without leaving the lane;
//                        three lane shuffles collect the wave's four h of a batch row into one 8-byte granule.
//   waves 4-7 (gather):  poll the h_{t-1} granules of the group's P workgroups into the LDS operand tile (nothing else in
//                        their vector-memory queue, so a poll is never stuck behind bulk traffic).
// Exchange region: [group][parity][b (16)][H/4] granules.
template <int NKS, int CH>            // CH = 16-byte granule pairs per gather thread: ceil(rows * (H/8) / 256)
__global__ __launch_bounds__(512) void lstm_fwd_p3(P3 p) {
    if (blockIdx.x == 0 && threadIdx.x < HDR_BYTES / 4) p.clear_next[threadIdx.x] = 0u;      // the successor's status block (see HDR_SLOTS)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, T = p.T, ND = p.ND;
    const int gid = blockIdx.x & 7, pw = blockIdx.x >> 3;
    const int d = gid % ND, slice = gid / ND;
    if (slice >= p.NS) return;
    const int b0 = slice * p.BS, nb = min(p.BS, p.B - b0);
    if (nb <= 0) return;
    const int u0 = pw * 16;
    const int tid = threadIdx.x;
    constexpr int LD = NKS * 32 + 8;                         // bf16 elements per operand-tile row (16-byte pad)
    __bf16* tiles = reinterpret_cast<__bf16*>(smem);         // [2][16][LD]  h_{t-1}, double buffered
    for (int i = tid; i < 2 * 16 * LD / 2; i += 512) reinterpret_cast<unsigned*>(smem)[i] = 0u;
    const int HG = H >> 2;
    u64* xg = p.xbuf + (long)gid * 2 * 16 * HG;              // this group's [parity][16][HG]
    // clear this producer's granules (both parities) with L2-local stores before the consensus, behind which polling
    // starts: a line an earlier launch left in this XCD's L2 with a matching tag cannot survive that
    for (int i = tid; i < 2 * 16 * 4; i += 512) {
        const int parity = i >> 6, r = i & 63, bb = r >> 2, qq = r & 3;
        st_gran_local(xg + ((long)parity * 16 + bb) * HG + (u0 >> 2) + qq, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.abort_flag) + 8 + gid, p.P, p.allow_local, p.abort_flag);

    if (tid >= 256) {
        // ---- gather role ----
        const int gt = tid - 256;
        const int HG2 = HG >> 1, total2 = nb * HG2;             // 16-byte pairs per row / in all (rows are contiguous)
        int slot_off[CH], cnt = 0;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int idx = gt + 256 * i;
            slot_off[i] = (idx < total2) ? (idx / HG2) * LD + (idx % HG2) * 8 : -1;
            if (idx < total2) cnt = i + 1;
        }
        DIAG3_DECL
        for (int s = 0; s < T; ++s) {
            __bf16* tile = tiles + (s & 1) * 16 * LD;
            if (s > 0 && cnt > 0) {
                u64 glo[CH], ghi[CH];
                for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(2);
                const u64* src = xg + (long)((s - 1) & 1) * 16 * HG + 2 * gt;
                gather16<CH>(src, 512, cnt, FWD3_MASK, fwd3_want(seq_of(s - 1), p.epoch), glo, ghi, p.abort_flag);
                DIAG3_MARK(0)
#pragma unroll
                for (int i = 0; i < CH; ++i)
                    if (slot_off[i] >= 0) {
                        u64* dst = reinterpret_cast<u64*>(tile + slot_off[i]);
                        dst[0] = glo[i] & ~FWD3_MASK;
                        dst[1] = ghi[i] & ~FWD3_MASK;
                    }
            }
            DIAG3_MARK(1)
            __syncthreads();
            DIAG3_MARK(2)
        }
        DIAG3_DUMP(256, 72)
        return;
    }

    // ---- compute role ----
    const int lane = tid & 63, w = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    bf16x8 wreg[NKS];
    {
        // A operand row m = lane & 15 = 4*unit + gate
        const float* wrow = p.whh + ((long)d * 4 * H + (long)(n & 3) * H + u0 + 4 * w + (n >> 2)) * H;
        float wf[NKS][8];
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) wf[ks][e] = wrow[min(ks * 32 + 8 * q + e, H - 1)];      // all loads in flight
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) wreg[ks][e] = (__bf16)((ks * 32 + 8 * q + e < H) ? wf[ks][e] : 0.f);
    }
    const bool bok = n < nb;
    const int bg = b0 + (bok ? n : 0);
    const int unit = u0 + 4 * w + q;
    const long g_ts = (long)ND * 4 * H, c_ts = (long)ND * H;
    unsigned short* gs_base = p.gates + ((long)bg * T * ND + d) * 4 * H + (long)unit * 4;
    float* c_base = p.c + ((long)bg * T * ND + d) * H + unit;
    unsigned short* y_base = p.y + ((long)bg * (T + 2) + 1) * c_ts + (long)d * H + u0 + 4 * w;      // row t+1; 4 units of this wave
    // the time pads of y (rows 0 and T+1 = h_{-1} / h_T = 0, read by the shifted rows of the weight-gradient contraction) are
    // zeroed here by the lanes that store this wave's h, instead of two fill launches per layer from the host
    if (bok && q == 0) {
        *reinterpret_cast<uint2*>(y_base - c_ts) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(y_base + (long)T * c_ts) = make_uint2(0u, 0u);
    }
    auto tix = [&](int s_) { return (d == 0) ? s_ : T - 1 - s_; };
    auto ldx = [&](int s_) -> uint2 {
        if (s_ < T && bok) return *reinterpret_cast<const uint2*>(gs_base + (long)tix(s_) * g_ts);
        return make_uint2(0u, 0u);
    };
    // The pre-activations are requested four steps ahead into FOUR NAMED registers and the time loop is unrolled by four:
    // a rotation (xgA = xgB; ...) moves a register whose load has just been issued, so the compiler must wait for it -
    // `s_waitcnt vmcnt(0)` at the loop's back edge, i.e. every step drained the whole vector-memory queue (0.47 us/step).
    uint2 xg0 = ldx(0), xg1 = ldx(1), xg2 = ldx(2), xg3 = ldx(3);
    float cst = 0.f;
    DIAG3_DECL
    int s = 0;

#define FWD3_STEP(XG)                                                                                                       \
    {                                                                                                                       \
        const long t = tix(s);                                                                                              \
        const __bf16* tile = tiles + (s & 1) * 16 * LD;                                                                     \
        DIAG3_MARK(7)                                                                                                       \
        __syncthreads();                         /* h_{t-1} tile complete */                                                \
        DIAG3_MARK(0)                                                                                                       \
        f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};                                                        \
        if (s > 0) {                                                                                                        \
            bf16x8 hb[NKS];                                                                                                 \
            _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) hb[ks] = *reinterpret_cast<const bf16x8*>(tile + n * LD + ks * 32 + 8 * q); \
            _Pragma("unroll") for (int ks = 0; ks < NKS; ++ks) {         /* two independent accumulation chains */          \
                if (ks & 1) acc2 = mma16(wreg[ks], hb[ks], acc2); else acc = mma16(wreg[ks], hb[ks], acc);                   \
            }                                                                                                               \
        }                                                                                                                   \
        const float gi = fast_sigmoid3(acc[0] + acc2[0] + bf2f((unsigned short)(XG.x & 0xFFFFu)));                           \
        const float gf = fast_sigmoid3(acc[1] + acc2[1] + bf2f((unsigned short)(XG.x >> 16)));                               \
        const float gg = fast_tanh3(acc[2] + acc2[2] + bf2f((unsigned short)(XG.y & 0xFFFFu)));                              \
        const float go = fast_sigmoid3(acc[3] + acc2[3] + bf2f((unsigned short)(XG.y >> 16)));                               \
        cst = gf * cst + gi * gg;                                                                                           \
        const float hv = go * fast_tanh3(cst);                                                                              \
        DIAG3_MARK(1)                                                                                                       \
        /* the wave's four units of batch row n sit in lanes n, n+16, n+32, n+48: collect them in lane n                    \
           (bit 14 of a bf16 is clear for |x| < 2; clearing it keeps the tag bits of the granule intact) */                 \
        const unsigned hb16 = f2bf_bits(hv) & 0xBFFFu;                                                                      \
        const unsigned h1 = __shfl(hb16, n + 16), h2 = __shfl(hb16, n + 32), h3 = __shfl(hb16, n + 48);                      \
        if (q == 0 && bok) {                                                                                                \
            const u64 v = (u64)hb16 | ((u64)h1 << 16) | ((u64)h2 << 32) | ((u64)h3 << 48);                                   \
            if (s + 1 < T) {                                                                                                \
                u64* dst = xg + ((long)(s & 1) * 16 + n) * HG + (u0 >> 2) + w;                                              \
                if (local) publish<true>(dst, v | fwd3_want(seq_of(s), p.epoch));                                           \
                else publish<false>(dst, v | fwd3_want(seq_of(s), p.epoch));                                                \
            }                                                                                                               \
            *reinterpret_cast<u64*>(y_base + t * c_ts) = v;                                                                 \
        }                                                                                                                   \
        DIAG3_MARK(2)                                                                                                       \
        /* saved activated gates (for BPTT), cell state, and the pre-activations four steps ahead */                        \
        if (bok) {                                                                                                          \
            uint2 o;                                                                                                        \
            o.x = (unsigned)f2bf_bits(gi) | ((unsigned)f2bf_bits(gf) << 16);                                                \
            o.y = (unsigned)f2bf_bits(gg) | ((unsigned)f2bf_bits(go) << 16);                                                \
            *reinterpret_cast<uint2*>(gs_base + t * g_ts) = o;                                                              \
            c_base[t * c_ts] = cst;                                                                                         \
        }                                                                                                                   \
        XG = ldx(s + 4);                                                                                                    \
        DIAG3_MARK(3)                                                                                                       \
        if (++s >= T) break;                                                                                                \
    }

    for (;;) { FWD3_STEP(xg0) FWD3_STEP(xg1) FWD3_STEP(xg2) FWD3_STEP(xg3) }
#undef FWD3_STEP
    DIAG3_DUMP(0, 64)
}

// ------------------------------------------------------------------------------------------------
// backward (BPTT), reduce-scatter form
// ------------------------------------------------------------------------------------------------
// Workgroup (me) of group (d, slice) owns units j0 = 16*me .. +15 of its batch rows: it turns their dh into the four
// gate-pre-activation gradients (reduction index k = 4*unit + gate, 64 of them) and multiplies by its W_hh rows: a PARTIAL
// dh_{t-1}[b, all H].  MFMA D[row = k' (unit of output tile tcol)][col = b].  Output tile tcol belongs to workgroup tcol.
// Waves 0-3: cell backward, MFMA, publish, bulk traffic.  Waves 4-7: poll and sum the producers' partials.
// Exchange region: [group][parity][consumer][producer][b (BS)][8 granules of two fp32].
template <int NTO>
__global__ __launch_bounds__(512) void lstm_bwd_p3(P3 p) {
    if (blockIdx.x == 0 && threadIdx.x < HDR_BYTES / 4) p.clear_next[threadIdx.x] = 0u;      // the successor's status block (see HDR_SLOTS)
    __shared__ __attribute__((aligned(16))) __bf16 tile[16 * 72];     // [b][64 + 8] dgates of this slice, k = 4*unit + gate
    __shared__ __attribute__((aligned(16))) float s_part[4 * 256];   // [producer group][b][16]
    const int H = p.H, T = p.T, ND = p.ND, P = p.P, BS = p.BS;
    const int gid = blockIdx.x & 7, me = blockIdx.x >> 3;
    const int d = gid % ND, slice = gid / ND;
    if (slice >= p.NS) return;
    const int b0 = slice * BS, nb = min(BS, p.B - b0);
    if (nb <= 0) return;
    const int j0 = me * 16;
    const int tid = threadIdx.x;
    constexpr int LD = 72;
    for (int i = tid; i < 16 * LD / 2; i += 512) reinterpret_cast<unsigned*>(tile)[i] = 0u;
    for (int i = tid; i < 4 * 256; i += 512) s_part[i] = 0.f;
    const long per_par = (long)P * P * BS * 8;
    u64* xg = p.xbuf + (long)gid * 2 * per_par;
    // clear this producer's granules in the L2 (see lstm_fwd_p3): [parity][consumer][me][b][8]
    for (int i = tid; i < 2 * P * BS * 8; i += 512) {
        const int parity = i / (P * BS * 8), r = i - parity * (P * BS * 8), pc = r / (BS * 8), rr = r - pc * (BS * 8);
        st_gran_local(xg + (long)parity * per_par + (((long)pc * P + me) * BS * 8) + rr, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.abort_flag) + 8 + gid, P, p.allow_local, p.abort_flag);

    if (tid >= 256) {
        // ---- gather role: (row gb, unit quad g4) x producer group gq (NTO = ceil(P/4) producers each) ----
        const int gt = tid - 256;
        const int gslot = gt & 63, gb = gslot >> 2, g4 = gslot & 3, gq = gt >> 6;
        const int pp_lo = gq * NTO, cntp = max(0, min(P - pp_lo, NTO));
        DIAG3_DECL
        for (int s = 0; s < T; ++s) {
            if (s > 0) {
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                if (gb < nb && cntp > 0) {
                    for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(2);
                    const u64* src = xg + (long)((s - 1) & 1) * per_par + (((long)me * P + pp_lo) * BS + gb) * 8 + 2 * g4;
                    u64 glo[NTO], ghi[NTO];
                    gather16<NTO>(src, (long)BS * 8, cntp, BWD3_MASK, bwd3_want(seq_of(s - 1), p.epoch), glo, ghi, p.abort_flag);
#pragma unroll
                    for (int i = 0; i < NTO; ++i)
                        if (i < cntp) {
                            a0 += __uint_as_float((unsigned)glo[i] & ~7u);
                            a1 += __uint_as_float((unsigned)(glo[i] >> 32) & ~7u);
                            a2 += __uint_as_float((unsigned)ghi[i] & ~7u);
                            a3 += __uint_as_float((unsigned)(ghi[i] >> 32) & ~7u);
                        }
                }
                DIAG3_MARK(0)
                *reinterpret_cast<float4*>(s_part + gq * 256 + gb * 16 + 4 * g4) = make_float4(a0, a1, a2, a3);
            }
            DIAG3_MARK(1)
            __syncthreads();
            DIAG3_MARK(2)
            __syncthreads();
            DIAG3_MARK(3)
        }
        DIAG3_DUMP(256, 72)
        return;
    }

    // ---- compute role ----
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    // resident weights, A operand: row = output unit kp = 16*tcol + n, reduction index k = 32*ks + 8q + e = 4*unit + gate
    bf16x8 wreg[NTO][2];
    {
        float wf[NTO][2][8];
#pragma unroll
        for (int ot = 0; ot < NTO; ++ot) {
            const int tcol = min(wave + 4 * ot, P - 1);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = ks * 32 + 8 * q + e;
                    wf[ot][ks][e] = p.whh[((long)d * 4 * H + (long)(k & 3) * H + j0 + (k >> 2)) * H + tcol * 16 + n];
                }
        }
#pragma unroll
        for (int ot = 0; ot < NTO; ++ot)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) wreg[ot][ks][e] = (__bf16)wf[ot][ks][e];
    }

    // element owned by this thread in the cell backward: (row eb, unit j0 + ej)
    const int eb = tid >> 4, ej = tid & 15;
    const bool eok = eb < nb;
    const int ebg = b0 + (eok ? eb : 0);
    // Every vector-memory operation of the time loop is issued by EVERY lane on EVERY path (inactive lanes read a valid
    // address and write into this workgroup's dump area): with operations under lane- or step-dependent conditions the
    // compiler's count of outstanding operations differs per path and it falls back to `s_waitcnt vmcnt(0)`.
    unsigned char* dump_wg = p.dump + (long)blockIdx.x * 512 * 16;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.region_bytes, 0x00020000);
    const unsigned dump_off = (unsigned)(dump_wg - reinterpret_cast<unsigned char*>(p.xbuf)) + (unsigned)tid * 16u;
    const long g_ts = (long)ND * 4 * H, c_ts = (long)ND * H;
    unsigned short* ge = p.gates + ((long)ebg * T * ND + d) * 4 * H + (long)(j0 + ej) * 4;
    const long cy_e = ((long)ebg * T * ND + d) * H + j0 + ej;
    auto tix = [&](int s_) { return (d == 0) ? T - 1 - s_ : s_; };
    // raw operands of one step exactly as loaded (no arithmetic on them here: a conversion or a select right behind the
    // load makes the compiler wait for it at once and the prefetch distance collapses to zero)
    struct Raw { unsigned dy; float c, cp, cpm; uint2 g; };
    auto load_raw = [&](int s_) -> Raw {
        Raw r{0u, 0.f, 0.f, 0.f, make_uint2(0u, 0u)};
        {
            const int t = tix(min(s_, T - 1));
            const int tp = (d == 0) ? t - 1 : t + 1;
            const bool has_cp = (d == 0) ? (t > 0) : (t < T - 1);
            r.g = *reinterpret_cast<const uint2*>(ge + (long)t * g_ts);
            r.dy = p.y[cy_e + (long)t * c_ts];
            r.c = p.c[cy_e + (long)t * c_ts];
            r.cp = p.c[cy_e + (long)(has_cp ? tp : t) * c_ts];
            r.cpm = has_cp ? 1.f : 0.f;
        }
        return r;
    };
    struct Coef { float dy, c1, c2, c3, c4, c5, f; };
    auto make_coef = [&](const Raw& r) -> Coef {
        const float gi = bf2f((unsigned short)(r.g.x & 0xFFFFu)), gf = bf2f((unsigned short)(r.g.x >> 16));
        const float gg = bf2f((unsigned short)(r.g.y & 0xFFFFu)), go = bf2f((unsigned short)(r.g.y >> 16));
        const float tc = fast_tanh3(r.c);
        Coef k;
        k.dy = bf2f((unsigned short)r.dy);
        k.c1 = go * (1.f - tc * tc);          // d c / d h
        k.c2 = gg * gi * (1.f - gi);          // d i_pre / d c
        k.c3 = (r.cp * r.cpm) * gf * (1.f - gf);   // d f_pre / d c
        k.c4 = gi * (1.f - gg * gg);          // d g_pre / d c
        k.c5 = tc * go * (1.f - go);          // d o_pre / d h
        k.f = gf;
        return k;
    };

    // operands four steps ahead in FOUR NAMED register sets, time loop unrolled by four (no register rotation: see
    // lstm_fwd_p3); the coefficients of step s+1 are formed at the end of step s, off the hand-off's critical path
    Raw raw0 = load_raw(0), raw1 = load_raw(1), raw2 = load_raw(2), raw3 = load_raw(3);
    Coef coef = make_coef(raw0);
    float carry = 0.f;
    DIAG3_DECL
    int s = 0;

#define BWD3_STEP(RCUR, RNEXT)                                                                                              \
    {                                                                                                                       \
        DIAG3_MARK(7)                                                                                                       \
        __syncthreads();                         /* recurrent partial sums of step s are in s_part */                       \
        DIAG3_MARK(0)                                                                                                       \
        /* cell backward of the owned element -> bf16 operand tile (k = 4*unit + gate) and the saved-gates slot */          \
        uint2 dg16;                                                                                                         \
        {                                                                                                                   \
            float dh = coef.dy;                                                                                             \
            if (s > 0) dh += (s_part[tid] + s_part[256 + tid]) + (s_part[512 + tid] + s_part[768 + tid]);                    \
            const float dc = dh * coef.c1 + carry;                                                                          \
            const float d0 = dc * coef.c2, d1 = dc * coef.c3, d2 = dc * coef.c4, d3 = dh * coef.c5;                          \
            carry = dc * coef.f;                                                                                            \
            dg16.x = (unsigned)f2bf_bits(d0) | ((unsigned)f2bf_bits(d1) << 16);                                             \
            dg16.y = (unsigned)f2bf_bits(d2) | ((unsigned)f2bf_bits(d3) << 16);                                             \
            if (eok) *reinterpret_cast<uint2*>(tile + eb * LD + 4 * ej) = dg16;      /* LDS */                              \
        }                                                                                                                   \
        DIAG3_MARK(1)                                                                                                       \
        __syncthreads();                                                                                                    \
        DIAG3_MARK(2)                                                                                                       \
        /* partial dh_{prev}[b, k'] for every k', handed to the owner of k' */                                              \
        {                                                                                                                   \
            const bf16x8 bq0 = *reinterpret_cast<const bf16x8*>(tile + n * LD + 8 * q);                                     \
            const bf16x8 bq1 = *reinterpret_cast<const bf16x8*>(tile + n * LD + 32 + 8 * q);                                \
            u64* dst = xg + (long)(s & 1) * per_par;                                                                        \
            const u64 want = bwd3_want(seq_of(s), p.epoch);                                                                 \
            f32x4 acc[NTO];                                                                                                 \
            _Pragma("unroll") for (int ot = 0; ot < NTO; ++ot) {                                                            \
                acc[ot] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                      \
                acc[ot] = mma16(wreg[ot][0], bq0, acc[ot]);                                                                 \
                acc[ot] = mma16(wreg[ot][1], bq1, acc[ot]);                                                                 \
            }                                                                                                               \
            _Pragma("unroll") for (int ot = 0; ot < NTO; ++ot) {                                                            \
                const int tcol = wave + 4 * ot;                                                                             \
                const bool pok = tcol < P && n < nb && s + 1 < T;                                                           \
                u64* o = dst + (((long)min(tcol, P - 1) * P + me) * BS + min(n, BS - 1)) * 8 + 2 * q;                         \
                const unsigned off = pok ? (unsigned)(reinterpret_cast<unsigned char*>(o) - reinterpret_cast<unsigned char*>(p.xbuf)) : dump_off; \
                const u64 v0 = ((u64)(__float_as_uint(acc[ot][0]) & ~7u) | ((u64)(__float_as_uint(acc[ot][1]) & ~7u) << 32)) | want; \
                const u64 v1 = ((u64)(__float_as_uint(acc[ot][2]) & ~7u) | ((u64)(__float_as_uint(acc[ot][3]) & ~7u) << 32)) | want; \
                publish_pair(xrsrc, off, v0, v1, local);                                                                    \
            }                                                                                                               \
        }                                                                                                                   \
        DIAG3_MARK(3)                                                                                                       \
        /* gradients wrt the gate pre-activations replace the saved gates; next coefficients; operands four steps ahead */  \
        *(eok ? reinterpret_cast<uint2*>(ge + (long)tix(s) * g_ts) : reinterpret_cast<uint2*>(dump_wg + tid * 16)) = dg16;    \
        coef = make_coef(RNEXT);                                                                                            \
        RCUR = load_raw(s + 4);                                                                                             \
        DIAG3_MARK(4)                                                                                                       \
        if (++s >= T) break;                                                                                                \
    }

    for (;;) { BWD3_STEP(raw0, raw1) BWD3_STEP(raw1, raw2) BWD3_STEP(raw2, raw3) BWD3_STEP(raw3, raw0) }
#undef BWD3_STEP
    DIAG3_DUMP(0, 64)
}

// ------------------------------------------------------------------------------------------------
// backward, second form: ALL-GATHER of the gate gradients instead of a reduce-scatter of partial dh
// ------------------------------------------------------------------------------------------------
// lstm_bwd_p3 multiplies the workgroup's 64 gate gradients by its W_hh rows into a partial dh_{t-1} for ALL H units and hands
// 5 KB of fp32 partials per step to their owners, who sum P of them.  Here a workgroup publishes only its own 64 x nb gate
// gradients (one 16-byte pair per (row, unit) straight from the thread that computed them: 1/5 of the bytes), every
// workgroup gathers the gate gradients of the whole group (P x nb x 64 values, converted to a bf16 operand tile in LDS) and
// computes dh_{t-1} for its OWN 16 units against the matching W_hh columns: one 16 x 16 output tile, K = 4H split over the
// four compute waves, partial tiles summed through LDS - no P-way sum on the critical path, and MFMA + publish shrink from
// 0.39 us to 0.14.  Same tags, same barriers per step, same operand prefetch as lstm_bwd_p3.  NOT the default: the 4x larger
// gather costs more than the smaller publish saves (see bwd_form3); kept as the measured alternative.
// Exchange region: [group][parity][producer][row < nb][16 units] pairs of granules {d_i, d_f}, {d_g, d_o}.
template <int NTO>
__global__ __launch_bounds__(512) void lstm_bwd_p4(P3 p) {
    if (blockIdx.x == 0 && threadIdx.x < HDR_BYTES / 4) p.clear_next[threadIdx.x] = 0u;      // the successor's status block (see HDR_SLOTS)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem4[];
    __shared__ __attribute__((aligned(16))) float s_part[4 * 256];   // [compute wave][b][16 units] partial dh
    constexpr int KSW = 2 * NTO;                                      // k-steps of 32 gate rows per compute wave (4 KSW >= 2 P)
    constexpr int LDT = 4 * NTO * 64 + 8;                             // bf16 per operand-tile row
    __bf16* tile = reinterpret_cast<__bf16*>(smem4);                  // [16 rows][LDT]: k = 64 producer + 4 unit + gate
    const int H = p.H, T = p.T, ND = p.ND, P = p.P, BS = p.BS;
    const int gid = blockIdx.x & 7, me = blockIdx.x >> 3;
    const int d = gid % ND, slice = gid / ND;
    if (slice >= p.NS) return;
    const int b0 = slice * BS, nb = min(BS, p.B - b0);
    if (nb <= 0) return;
    const int j0 = me * 16;
    const int tid = threadIdx.x;
    for (int i = tid; i < 16 * LDT / 2; i += 512) reinterpret_cast<unsigned*>(tile)[i] = 0u;
    for (int i = tid; i < 4 * 256; i += 512) s_part[i] = 0.f;
    const long per_par = (long)P * BS * 32;                           // granules per parity
    u64* xg = p.xbuf + (long)gid * 2 * per_par;
    const int PR = nb * 16;                                           // pairs per producer
    // clear this producer's granules in the L2 (see lstm_fwd_p3)
    for (int i = tid; i < 2 * PR * 2; i += 512) {
        const int parity = i / (PR * 2), r = i - parity * (PR * 2);
        st_gran_local(xg + (long)parity * per_par + (long)me * PR * 2 + r, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.abort_flag) + 8 + gid, P, p.allow_local, p.abort_flag);

    if (tid >= 256) {
        // ---- gather role: pair i of the group (producer, row, unit) -> four bf16 of the operand tile ----
        const int gt = tid - 256;
        const int total2 = P * PR;
        DIAG3_DECL
        for (int s = 0; s < T; ++s) {
            if (s > 0) {
                for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(2);
                const u64* src = xg + (long)((s - 1) & 1) * per_par;
                const u64 want = bwd3_want(seq_of(s - 1), p.epoch);
                for (int i0 = gt; i0 < total2; i0 += 256 * NTO) {
                    int toff[NTO], cnt = 0;
#pragma unroll
                    for (int k = 0; k < NTO; ++k) {
                        const int pi = i0 + 256 * k;
                        if (pi < total2) cnt = k + 1;
                        const int pc = min(pi, total2 - 1);
                        const int prod = pc / PR, r = pc - prod * PR;
                        toff[k] = (r >> 4) * LDT + prod * 64 + 4 * (r & 15);
                    }
                    u64 glo[NTO], ghi[NTO];
                    gather16<NTO>(src + 2 * (long)i0, 2 * 256, cnt, BWD3_MASK, want, glo, ghi, p.abort_flag);
#pragma unroll
                    for (int k = 0; k < NTO; ++k)
                        if (k < cnt) {
                            const unsigned h0 = (unsigned)f2bf_bits(__uint_as_float((unsigned)glo[k] & ~7u)) |
                                                ((unsigned)f2bf_bits(__uint_as_float((unsigned)(glo[k] >> 32) & ~7u)) << 16);
                            const unsigned h1 = (unsigned)f2bf_bits(__uint_as_float((unsigned)ghi[k] & ~7u)) |
                                                ((unsigned)f2bf_bits(__uint_as_float((unsigned)(ghi[k] >> 32) & ~7u)) << 16);
                            *reinterpret_cast<uint2*>(tile + toff[k]) = make_uint2(h0, h1);
                        }
                }
                DIAG3_MARK(0)
            }
            DIAG3_MARK(1)
            __syncthreads();                     // A: the operand tile of step s is complete
            DIAG3_MARK(2)
            __syncthreads();                     // B: the tile has been read
            DIAG3_MARK(3)
        }
        DIAG3_DUMP(256, 72)
        return;
    }

    // ---- compute role ----
    const int lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    // resident weights, A operand: row = own unit j0 + n, reduction index kk = 32 ksg + 8 q + e = 64 producer + 4 unit' + gate
    bf16x8 wreg[KSW];
#pragma unroll
    for (int ksl = 0; ksl < KSW; ++ksl) {
        const int ksg = wave * KSW + ksl;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kk = 32 * ksg + 8 * q + e;
            const int prod = kk >> 6, kin = kk & 63, jp = min(prod, P - 1) * 16 + (kin >> 2), gate = kin & 3;
            const float w = p.whh[((long)d * 4 * H + (long)gate * H + jp) * H + j0 + n];
            wreg[ksl][e] = (__bf16)(prod < P ? w : 0.f);
        }
    }

    const int eb = tid >> 4, ej = tid & 15;
    const bool eok = eb < nb;
    const int ebg = b0 + (eok ? eb : 0);
    unsigned char* dump_wg = p.dump + (long)blockIdx.x * 512 * 16;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(p.xbuf, 0, p.region_bytes, 0x00020000);
    const unsigned dump_off = (unsigned)(dump_wg - reinterpret_cast<unsigned char*>(p.xbuf)) + (unsigned)tid * 16u;
    // own pair of parity 0 (a lane without an element publishes into the dump area)
    const unsigned pub_off0 = (unsigned)(reinterpret_cast<unsigned char*>(xg + ((long)me * PR + (long)min(eb, nb - 1) * 16 + ej) * 2) -
                                         reinterpret_cast<unsigned char*>(p.xbuf));
    const unsigned par_bytes = (unsigned)(per_par * sizeof(u64));
    const long g_ts = (long)ND * 4 * H, c_ts = (long)ND * H;
    unsigned short* ge = p.gates + ((long)ebg * T * ND + d) * 4 * H + (long)(j0 + ej) * 4;
    const long cy_e = ((long)ebg * T * ND + d) * H + j0 + ej;
    auto tix = [&](int s_) { return (d == 0) ? T - 1 - s_ : s_; };
    struct Raw { unsigned dy; float c, cp, cpm; uint2 g; };
    auto load_raw = [&](int s_) -> Raw {
        Raw r{0u, 0.f, 0.f, 0.f, make_uint2(0u, 0u)};
        {
            const int t = tix(min(s_, T - 1));
            const int tp = (d == 0) ? t - 1 : t + 1;
            const bool has_cp = (d == 0) ? (t > 0) : (t < T - 1);
            r.g = *reinterpret_cast<const uint2*>(ge + (long)t * g_ts);
            r.dy = p.y[cy_e + (long)t * c_ts];
            r.c = p.c[cy_e + (long)t * c_ts];
            r.cp = p.c[cy_e + (long)(has_cp ? tp : t) * c_ts];
            r.cpm = has_cp ? 1.f : 0.f;
        }
        return r;
    };
    struct Coef { float dy, c1, c2, c3, c4, c5, f; };
    auto make_coef = [&](const Raw& r) -> Coef {
        const float gi = bf2f((unsigned short)(r.g.x & 0xFFFFu)), gf = bf2f((unsigned short)(r.g.x >> 16));
        const float gg = bf2f((unsigned short)(r.g.y & 0xFFFFu)), go = bf2f((unsigned short)(r.g.y >> 16));
        const float tc = fast_tanh3(r.c);
        Coef k;
        k.dy = bf2f((unsigned short)r.dy);
        k.c1 = go * (1.f - tc * tc);
        k.c2 = gg * gi * (1.f - gi);
        k.c3 = (r.cp * r.cpm) * gf * (1.f - gf);
        k.c4 = gi * (1.f - gg * gg);
        k.c5 = tc * go * (1.f - go);
        k.f = gf;
        return k;
    };
    Raw raw0 = load_raw(0), raw1 = load_raw(1), raw2 = load_raw(2), raw3 = load_raw(3);
    Coef coef = make_coef(raw0);
    float carry = 0.f;
    DIAG3_DECL
    int s = 0;
    const __bf16* trow = tile + n * LDT + 32 * (wave * KSW) + 8 * q;

#define BWD4_STEP(RCUR, RNEXT)                                                                                              \
    {                                                                                                                       \
        DIAG3_MARK(7)                                                                                                       \
        __syncthreads();                         /* A: gate gradients of step s-1 of the whole group are in the tile */     \
        DIAG3_MARK(0)                                                                                                       \
        if (s > 0) {                                                                                                        \
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};                                                 \
            _Pragma("unroll") for (int ksl = 0; ksl < KSW; ksl += 2) {                                                      \
                acc0 = mma16(wreg[ksl], *reinterpret_cast<const bf16x8*>(trow + 32 * ksl), acc0);                           \
                acc1 = mma16(wreg[ksl + 1], *reinterpret_cast<const bf16x8*>(trow + 32 * ksl + 32), acc1);                  \
            }                                                                                                               \
            /* lane (n = row b, q): units 4q..4q+3 -> s_part[wave][b][16] */                                                \
            *reinterpret_cast<float4*>(s_part + wave * 256 + n * 16 + 4 * q) =                                              \
                make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);                    \
        }                                                                                                                   \
        DIAG3_MARK(1)                                                                                                       \
        __syncthreads();                         /* B */                                                                    \
        DIAG3_MARK(2)                                                                                                       \
        uint2 dg16;                                                                                                         \
        {                                                                                                                   \
            float dh = coef.dy;                                                                                             \
            if (s > 0) dh += (s_part[tid] + s_part[256 + tid]) + (s_part[512 + tid] + s_part[768 + tid]);                    \
            const float dc = dh * coef.c1 + carry;                                                                          \
            const float d0 = dc * coef.c2, d1 = dc * coef.c3, d2 = dc * coef.c4, d3 = dh * coef.c5;                          \
            carry = dc * coef.f;                                                                                            \
            dg16.x = (unsigned)f2bf_bits(d0) | ((unsigned)f2bf_bits(d1) << 16);                                             \
            dg16.y = (unsigned)f2bf_bits(d2) | ((unsigned)f2bf_bits(d3) << 16);                                             \
            const u64 want = bwd3_want(seq_of(s), p.epoch);                                                                 \
            const bool pok = eok && s + 1 < T;                                                                              \
            const unsigned off = pok ? pub_off0 + (unsigned)(s & 1) * par_bytes : dump_off;                                 \
            const u64 v0 = ((u64)(__float_as_uint(d0) & ~7u) | ((u64)(__float_as_uint(d1) & ~7u) << 32)) | want;            \
            const u64 v1 = ((u64)(__float_as_uint(d2) & ~7u) | ((u64)(__float_as_uint(d3) & ~7u) << 32)) | want;            \
            publish_pair(xrsrc, off, v0, v1, local);                                                                        \
        }                                                                                                                   \
        DIAG3_MARK(3)                                                                                                       \
        *(eok ? reinterpret_cast<uint2*>(ge + (long)tix(s) * g_ts) : reinterpret_cast<uint2*>(dump_wg + tid * 16)) = dg16;    \
        coef = make_coef(RNEXT);                                                                                            \
        RCUR = load_raw(s + 4);                                                                                             \
        DIAG3_MARK(4)                                                                                                       \
        if (++s >= T) break;                                                                                                \
    }

    for (;;) { BWD4_STEP(raw0, raw1) BWD4_STEP(raw1, raw2) BWD4_STEP(raw2, raw3) BWD4_STEP(raw3, raw0) }
#undef BWD4_STEP
    DIAG3_DUMP(0, 64)
}

int allow_local3() {
    static const int on = [] { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); return (e && e[0] == '0') ? 0 : 1; }();
    return on;
}
// which backward kernel: 3 = reduce-scatter of partial dh (lstm_bwd_p3, default), 4 = all-gather of the gate gradients
// (lstm_bwd_p4, ASR_LSTM3_BWD=4).  Measured on MI355X, B=16 x T=1200 x H=320: 1.36 vs 1.77 us per step (B=64: 1.65 vs 4.4) -
// the all-gather form publishes 1/5 of the bytes and has no P-way sum, but every workgroup polls 4x the bytes (the whole group's
// gate gradients instead of its own units' partials) and the poll is the critical path: 0.93 -> 1.46 us.
int bwd_form3() {
    static const int f = [] { const char* e = getenv("ASR_LSTM3_BWD"); return (e && e[0] == '4') ? 4 : 3; }();
    return f;
}
int poll_delay3(bool bwd) {
    static const int df = [] { const char* e = getenv("ASR_LSTM3_POLL_DELAY_FWD"); return e ? atoi(e) : 6; }();
    static const int db = [] { const char* e = getenv("ASR_LSTM3_POLL_DELAY_BWD"); return e ? atoi(e) : 4; }();
    return bwd ? db : df;
}

size_t fwd3_region_bytes(int H) { return (size_t)8 * 2 * 16 * (H / 4) * sizeof(u64); }
size_t bwd3_dump_bytes(int H) { return (size_t)8 * (H / 16) * 512 * 16; }
size_t bwd3_region_bytes(int H, int BS) { const size_t P = H / 16; return (size_t)8 * 2 * P * P * BS * 8 * sizeof(u64) + bwd3_dump_bytes(H); }
int slice_rows(int B, int ND) { const int NS = 8 / ND; return (B + NS - 1) / NS; }

// every workgroup of the launch must be resident at the same time: the grid against what the device can hold beside
// `reserved_cus` compute units that another stream may be using (data-parallel all-reduce kernels)
template <typename K>
bool fits_resident(K kernel, int grid_active, size_t lds, int reserved_cus) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 512, lds) != hipSuccess || per_cu < 1) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    if (per_cu > 2) per_cu = 2;          // the occupancy API can over-report by one block per CU (guide): stay well inside
    return (long)grid_active <= (long)(cus - reserved_cus) * per_cu;
}

}  // namespace

bool lstm3_shape_ok(int B, int H, int ND) { return H % 16 == 0 && H >= 16 && H <= 512 && B >= 1 && B <= 16 * (8 / ND) && (ND == 1 || ND == 2); }

size_t lstm_persist3_workspace_bytes(int B, int H, int ND, int bwd) {
    if (!lstm3_shape_ok(B, H, ND)) return 0;
    return HDR_SLOTS * HDR_BYTES + (bwd ? BWD_REGIONS * bwd3_region_bytes(H, slice_rows(B, ND)) : FWD_REGIONS * fwd3_region_bytes(H));
}

#define FWD3_LAUNCH(NKS_, CH_)                                                                                              \
    {                                                                                                                       \
        const size_t lds = 2 * 16 * (NKS_ * 32 + 8) * 2;                                                                    \
        if (!fits_resident(lstm_fwd_p3<NKS_, CH_>, groups * p.P, lds, reserved_cus)) return 1;                               \
        hipLaunchKernelGGL((lstm_fwd_p3<NKS_, CH_>), dim3(8 * p.P), dim3(512), lds, st, p);                                  \
        goto launched;                                                                                                      \
    }
#define FWD3_CASE(NKS_)                                                                                                     \
    if (nks <= NKS_) {                                                                                                      \
        constexpr int CHMAX = (NKS_ + 3) / 4;                                                                               \
        const int ch = (BS * (H / 8) + 255) / 256;                                                                          \
        if (ch <= 1) FWD3_LAUNCH(NKS_, 1)                                                                                   \
        else if (ch <= 2 || CHMAX <= 2) FWD3_LAUNCH(NKS_, (CHMAX < 2 ? CHMAX : 2))                                          \
        else FWD3_LAUNCH(NKS_, CHMAX)                                                                                       \
    }
#define BWD3_CASE(NTO_)                                                                                                     \
    if (nto <= NTO_) {                                                                                                      \
        if (bwd_form3() == 4) {                                                                                             \
            const size_t lds4 = (size_t)16 * (4 * NTO_ * 64 + 8) * 2;                                                       \
            static bool attr4 = false;                                                                                      \
            if (!attr4) { hipFuncSetAttribute((const void*)lstm_bwd_p4<NTO_>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024); attr4 = true; } \
            if (!fits_resident(lstm_bwd_p4<NTO_>, groups * p.P, lds4, reserved_cus)) return 1;                               \
            hipLaunchKernelGGL(lstm_bwd_p4<NTO_>, dim3(8 * p.P), dim3(512), lds4, st, p);                                    \
            goto launched;                                                                                                  \
        }                                                                                                                   \
        if (!fits_resident(lstm_bwd_p3<NTO_>, groups * p.P, 0, reserved_cus)) return 1;                                      \
        hipLaunchKernelGGL(lstm_bwd_p3<NTO_>, dim3(8 * p.P), dim3(512), 0, st, p);                                           \
        goto launched;                                                                                                      \
    }

// Return ASR_OK when launched, 1 when the shape has no plan (or the grid would not be resident), negative on error.
int lstm_fwd_persistent3(unsigned short* gates, const float* whh, unsigned short* y, float* c, int B, int T, int H, int ND,
                         void* ws, size_t ws_bytes, unsigned epoch, int reserved_cus, hipStream_t st) {
    if (!lstm3_shape_ok(B, H, ND) || !ws || ((uintptr_t)ws & 255) != 0) return 1;
    if (ws_bytes < lstm_persist3_workspace_bytes(B, H, ND, 0)) return 1;
    const int NS = 8 / ND, BS = slice_rows(B, ND);
    const int groups = ND * ((B + BS - 1) / BS);
    unsigned* hdr = (unsigned*)((char*)ws + (size_t)(epoch & 1u) * HDR_BYTES);
    unsigned* hdr_next = (unsigned*)((char*)ws + (size_t)((epoch + 1u) & 1u) * HDR_BYTES);
    u64* region = (u64*)((char*)ws + HDR_SLOTS * HDR_BYTES + (size_t)((epoch >> 2) % FWD_REGIONS) * fwd3_region_bytes(H));
    P3 p{gates, whh, y, c, region, hdr, hdr_next, B, T, H, ND, H / 16, NS, BS, allow_local3(), poll_delay3(false), epoch, nullptr, 0u};
    const int nks = (H + 31) / 32;
    FWD3_CASE(1) FWD3_CASE(2) FWD3_CASE(4) FWD3_CASE(6) FWD3_CASE(8) FWD3_CASE(10) FWD3_CASE(12) FWD3_CASE(16)
    return 1;
launched:
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_lstm3_fwd: launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

int lstm_bwd_persistent3(unsigned short* gates, const float* whh, const unsigned short* dy, const float* c, int B, int T, int H, int ND,
                         void* ws, size_t ws_bytes, unsigned epoch, int reserved_cus, hipStream_t st) {
    if (!lstm3_shape_ok(B, H, ND) || !ws || ((uintptr_t)ws & 255) != 0) return 1;
    if (ws_bytes < lstm_persist3_workspace_bytes(B, H, ND, 1)) return 1;
    const int NS = 8 / ND, BS = slice_rows(B, ND);
    const int groups = ND * ((B + BS - 1) / BS);
    unsigned* hdr = (unsigned*)((char*)ws + (size_t)(epoch & 1u) * HDR_BYTES);
    unsigned* hdr_next = (unsigned*)((char*)ws + (size_t)((epoch + 1u) & 1u) * HDR_BYTES);
    u64* region = (u64*)((char*)ws + HDR_SLOTS * HDR_BYTES + (size_t)((epoch >> 4) % BWD_REGIONS) * bwd3_region_bytes(H, BS));
    const size_t rbytes = bwd3_region_bytes(H, BS);
    if (rbytes >= (1ull << 31)) return 1;
    P3 p{gates, whh, const_cast<unsigned short*>(dy), const_cast<float*>(c), region, hdr, hdr_next, B, T, H, ND, H / 16, NS, BS,
         allow_local3(), poll_delay3(true), epoch, (unsigned char*)region + rbytes - bwd3_dump_bytes(H), (unsigned)rbytes};
    const int nto = (p.P + 3) / 4;
    BWD3_CASE(1) BWD3_CASE(2) BWD3_CASE(3) BWD3_CASE(4) BWD3_CASE(5) BWD3_CASE(6) BWD3_CASE(8)
    return 1;
launched:
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_lstm3_bwd: launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}
