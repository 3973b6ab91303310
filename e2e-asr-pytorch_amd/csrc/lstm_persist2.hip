// Second-generation persistent LSTM recurrence for the encoder's working point: bf16 matrix-core operands,
// B <= 16 rows (one MFMA tile), H a multiple of 16.  One launch walks all T steps of both directions
// (nn.LSTM recurrence + its BPTT, reference src/module.py:1023,1049).
//
// What changed against lstm_persist.hip (which stays for fp32 parity mode, B > 16 and odd shapes):
//  * finer cut: workgroup (p, dir) owns 16 hidden units (P = H/16 workgroups per direction), so the serial
//    part of a step (MFMA chain, activations) is 4x shorter and the resident W_hh slice fits in VGPRs
//    (v1 kept 160 registers of weights in AGPRs and copied 4 of them back before every MFMA);
//  * forward: wave g computes gate g (i,f,g,o) for the 16 units: 10 MFMAs at H=320, 4 activations per lane;
//    the gates meet in LDS and wave 0 does the cell update;
//  * the MFMA is issued with the weights as the A operand, so a lane's four accumulator registers are four
//    CONSECUTIVE hidden units of one batch row: every bulk load/store is a 16-byte access and granules pack
//    adjacent units without lane shuffles;
//  * denser granules, the tag costs no payload: forward packs four bf16 h per 8-byte granule and keeps a 2-bit
//    sequence number in bit 14 of the first two (|h| <= 1, so that exponent bit is always 0); backward packs
//    two fp32 partial sums per granule with the sequence number in their mantissa LSBs (2^-23 relative,
//    far below the bf16 operand rounding).  Sequence = ((step >> 1) mod 3) + 1: consecutive occupants of a
//    parity buffer differ, and the memset state 0 is never valid;
//  * bulk traffic of a step (saved gates / c / y stores, prefetch of the operands two steps ahead) is issued
//    right AFTER the first burst of polling loads: vmcnt retires in issue order, so anything issued before
//    the poll would sit on the hand-off's critical path.
// Hand-off, bounded spins and the abort word are as in lstm_persist.hip (guide form R2).
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int SPIN_LIMIT2 = 1 << 22;

struct P2 {
    float* gates;        // (B,T,ND,4H)
    const float* whh;    // (ND,4H,H)
    const float* bias2;  // fwd: (ND,4H) or null
    float* y;            // fwd: h out (B,T,ND*H);  bwd: dy in
    float* c;            // (B,T,ND,H)
    u64* xbuf;
    unsigned* abort_flag;
    int B, T, H, ND, P;
};

__device__ __forceinline__ u64 ld_gran(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_gran(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned seq_of(int s) { return (unsigned)((s >> 1) % 3) + 1u; }

// Fetches granules base[0], base[stride], ... (n <= CH of them) until all carry `want` under `mask`.
// First round: the CH loads, then the caller's deferred memory traffic `io`, then the check.
template <int CH, typename IO>
__device__ __forceinline__ void gather_seq(const u64* base, long stride, int n, u64 mask, u64 want, u64 (&g)[CH],
                                           unsigned* abort_flag, IO&& io) {
#pragma unroll
    for (int i = 0; i < CH; ++i) g[i] = ld_gran(base + (i < n ? i : 0) * stride);
    __builtin_amdgcn_sched_barrier(0);
    io();
    __builtin_amdgcn_sched_barrier(0);
    int spins = 0;
    while (true) {
        bool ok = true;
#pragma unroll
        for (int i = 0; i < CH; ++i) ok = ok && ((i >= n) || ((g[i] & mask) == want));
        if (ok) return;
        ++spins;
        if ((spins & 63) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (spins > SPIN_LIMIT2) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int i = 0; i < CH; ++i) g[i] = ld_gran(base + (i < n ? i : 0) * stride);
    }
}

__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup (p, dir): hidden units u0 = 16p .. 16p+15.  Wave g: gate g.  MFMA D[row = unit 4q+r][col = b = n]
// = sum_k W_hh[g*H + u0 + row][k] * h_{t-1}[b][k].  Exchange buffer xbuf[parity][dir][b][H/4].
constexpr u64 FWD_MASK = (1ull << 14) | (1ull << 30);
__device__ __forceinline__ u64 fwd_want(unsigned seq) { return ((u64)(seq & 1u) << 14) | ((u64)(seq >> 1) << 30); }

template <int NKS>
__global__ __launch_bounds__(256) void lstm_fwd_p2(P2 p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int H = p.H, T = p.T, ND = p.ND, B = p.B;
    const int d = blockIdx.y, u0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    constexpr int LD = NKS * 32 + 8;                         // bf16 elements per operand-tile row (16-byte pad)
    constexpr int CH = (NKS + 1) / 2;                        // granules per thread: 16 rows * (H/4) / 256
    __bf16* tiles = reinterpret_cast<__bf16*>(smem);         // [2][16][LD]  h_{t-1}, double buffered
    float* gbuf = reinterpret_cast<float*>(smem + 2 * 16 * LD * 2);   // [4][16][20] activated gates of this step
    for (int i = tid; i < 2 * 16 * LD / 2; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    bf16x8 w[NKS];
    {
        const float* wrow = p.whh + ((long)d * 4 * H + (long)g * H + u0 + n) * H;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ks * 32 + 8 * q + e;
                w[ks][e] = (__bf16)((k < H) ? wrow[k] : 0.f);
            }
    }
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = p.bias2 ? p.bias2[(long)d * 4 * H + (long)g * H + u0 + 4 * q + r] : 0.f;

    const bool bok = n < B;
    const int bc = bok ? n : 0;
    const long g_ts = (long)ND * 4 * H, c_ts = (long)ND * H;
    const float* xg_base = p.gates + ((long)bc * T * ND + d) * 4 * H + (long)g * H + u0 + 4 * q;
    float* gs_base = p.gates + ((long)bc * T * ND + d) * 4 * H + (long)g * H + u0 + 4 * q;
    const long cy_off = ((long)bc * T * ND + d) * H + u0 + 4 * q;
    auto tix = [&](int s_) { return (d == 0) ? s_ : T - 1 - s_; };
    auto ldx = [&](int s_) -> float4 {
        if (s_ < T && bok) return *reinterpret_cast<const float4*>(xg_base + (long)tix(s_) * g_ts);
        return make_float4(0.f, 0.f, 0.f, 0.f);
    };

    const int HG = H >> 2, total = B * HG;
    const long xstride = (long)ND * B * HG;
    int slot_off[CH], cnt = 0;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
        const int idx = tid + 256 * i;
        slot_off[i] = (idx < total) ? (idx / HG) * LD + (idx % HG) * 4 : -1;
        if (idx < total) cnt = i + 1;
    }
    float4 xgA = ldx(0), xgB = ldx(1), xgC = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 o_gate = xgC, o_c = xgC, o_h = xgC, cst = xgC;
    __syncthreads();

    for (int s = 0; s <= T; ++s) {
        // deferred bulk traffic: outputs of step s-1, operands of step s+2
        auto io = [&]() {
            if (s > 0 && bok) {
                const long tp = tix(s - 1);
                *reinterpret_cast<float4*>(gs_base + tp * g_ts) = o_gate;
                if (g == 0) {
                    *reinterpret_cast<float4*>(p.c + cy_off + tp * c_ts) = o_c;
                    *reinterpret_cast<float4*>(p.y + cy_off + tp * c_ts) = o_h;
                }
            }
            xgC = ldx(s + 2);
        };
        if (s == T) { io(); break; }
        __bf16* tile = tiles + (s & 1) * 16 * LD;
        if (s > 0 && cnt > 0) {
            u64 gr[CH];
            const u64* src = p.xbuf + (long)((s - 1) & 1) * xstride + (long)d * B * HG + tid;
            gather_seq<CH>(src, 256, cnt, FWD_MASK, fwd_want(seq_of(s - 1)), gr, p.abort_flag, io);
#pragma unroll
            for (int i = 0; i < CH; ++i)
                if (slot_off[i] >= 0) *reinterpret_cast<u64*>(tile + slot_off[i]) = gr[i] & ~FWD_MASK;
        } else {
            io();
        }
        __syncthreads();
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                acc = mma16(w[ks], *reinterpret_cast<const bf16x8*>(tile + n * LD + ks * 32 + 8 * q), acc);
        }
        float a[4];
        const float xr[4] = {xgA.x, xgA.y, xgA.z, xgA.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float pre = acc[r] + xr[r] + bias[r];
            a[r] = (g == 2) ? fast_tanh(pre) : fast_sigmoid(pre);
        }
        o_gate = make_float4(a[0], a[1], a[2], a[3]);
        *reinterpret_cast<float4*>(gbuf + (g * 16 + n) * 20 + 4 * q) = o_gate;
        __syncthreads();
        if (g == 0) {
            const float4 gi = *reinterpret_cast<const float4*>(gbuf + (0 * 16 + n) * 20 + 4 * q);
            const float4 gf = *reinterpret_cast<const float4*>(gbuf + (1 * 16 + n) * 20 + 4 * q);
            const float4 gg = *reinterpret_cast<const float4*>(gbuf + (2 * 16 + n) * 20 + 4 * q);
            const float4 go = *reinterpret_cast<const float4*>(gbuf + (3 * 16 + n) * 20 + 4 * q);
            cst.x = gf.x * cst.x + gi.x * gg.x; cst.y = gf.y * cst.y + gi.y * gg.y;
            cst.z = gf.z * cst.z + gi.z * gg.z; cst.w = gf.w * cst.w + gi.w * gg.w;
            o_c = cst;
            o_h = make_float4(go.x * fast_tanh(cst.x), go.y * fast_tanh(cst.y), go.z * fast_tanh(cst.z), go.w * fast_tanh(cst.w));
            if (s + 1 < T && bok) {
                // bit 14 of a bf16 is clear for every |x| < 2; clearing it (only NaN/Inf are affected, and those
                // still reach the loss through y) keeps the sequence bits intact in all cases
                const u64 b0 = f2bf_bits(o_h.x) & 0xBFFFu, b1 = f2bf_bits(o_h.y) & 0xBFFFu;
                const u64 b2 = f2bf_bits(o_h.z), b3 = f2bf_bits(o_h.w);
                const u64 v = b0 | (b1 << 16) | (b2 << 32) | (b3 << 48);
                // sequence bits live in elements 0 and 1 (bits 14 and 30)
                st_gran(p.xbuf + (long)(s & 1) * xstride + ((long)d * B + n) * HG + (u0 >> 2) + q, v | fwd_want(seq_of(s)));
            }
        }
        xgA = xgB; xgB = xgC;
    }
}

// ------------------------------------------------------------------------------------------------
// backward (BPTT), reduce-scatter form
// ------------------------------------------------------------------------------------------------
// Workgroup (me, dir) owns units j0 = 16*me .. +15: it turns their dh into the four gate-pre-activation
// gradients (K = 64 reduction index nl = g*16 + jl) and multiplies by its W_hh rows: a PARTIAL dh_{t-1}[b, all H].
// MFMA D[row = k' unit 4q+r of output tile][col = b = n].  Output tile tcol (16 units) belongs to workgroup tcol.
// Exchange buffer xbuf[parity][dir][consumer][producer][b][8 pairs of fp32].
constexpr u64 BWD_MASK = 1ull | (1ull << 32);
__device__ __forceinline__ u64 bwd_want(unsigned seq) { return (u64)(seq & 1u) | ((u64)(seq >> 1) << 32); }

template <int NTO>
__global__ __launch_bounds__(256) void lstm_bwd_p2(P2 p) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[16 * 72];     // [b][64 + 8] dgates of this slice
    __shared__ __attribute__((aligned(16))) float s_part[2 * 256];   // [producer half][b][16]
    const int H = p.H, T = p.T, ND = p.ND, B = p.B, P = p.P;
    const int d = blockIdx.y, me = blockIdx.x, j0 = me * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    constexpr int LD = 72;
    for (int i = tid; i < 16 * LD / 2; i += 256) reinterpret_cast<unsigned*>(tile)[i] = 0u;

    // resident weights, A operand: row = output unit kp = 16*tcol + n, k = nl = 32*ks + 8q + e
    bf16x8 w[NTO][2];
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
        const int tcol = wave + 4 * ot;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int nl = ks * 32 + 8 * q + e;
                float v = 0.f;
                if (tcol < P) v = p.whh[((long)d * 4 * H + (long)(nl >> 4) * H + j0 + (nl & 15)) * H + tcol * 16 + n];
                w[ot][ks][e] = (__bf16)v;
            }
    }

    // element owned by this thread in the cell backward: (row eb, unit j0 + ej)
    const int eb = tid >> 4, ej = tid & 15;
    const bool eok = eb < B;
    const int ebc = eok ? eb : 0;
    const long g_ts = (long)ND * 4 * H, c_ts = (long)ND * H;
    float* ge = p.gates + ((long)ebc * T * ND + d) * 4 * H + j0 + ej;
    const long cy_e = ((long)ebc * T * ND + d) * H + j0 + ej;
    auto tix = [&](int s_) { return (d == 0) ? T - 1 - s_ : s_; };
    struct Raw { float dy, gi, gf, gg, go, c, cp; };
    auto load_raw = [&](int s_) -> Raw {
        Raw r{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (s_ < T && eok) {
            const int t = tix(s_);
            const int tp = (d == 0) ? t - 1 : t + 1;
            const bool has_cp = (d == 0) ? (t > 0) : (t < T - 1);
            const float* gp = ge + (long)t * g_ts;
            r.dy = p.y[cy_e + (long)t * c_ts];
            r.gi = gp[0]; r.gf = gp[H]; r.gg = gp[2 * (long)H]; r.go = gp[3 * (long)H];
            r.c = p.c[cy_e + (long)t * c_ts];
            r.cp = has_cp ? p.c[cy_e + (long)tp * c_ts] : 0.f;
        }
        return r;
    };
    struct Coef { float dy, c1, c2, c3, c4, c5, f; };
    auto make_coef = [&](const Raw& r) -> Coef {
        const float tc = fast_tanh(r.c);
        Coef k;
        k.dy = r.dy;
        k.c1 = r.go * (1.f - tc * tc);          // d c / d h
        k.c2 = r.gg * r.gi * (1.f - r.gi);      // d i_pre / d c
        k.c3 = r.cp * r.gf * (1.f - r.gf);      // d f_pre / d c
        k.c4 = r.gi * (1.f - r.gg * r.gg);      // d g_pre / d c
        k.c5 = tc * r.go * (1.f - r.go);        // d o_pre / d h
        k.f = r.gf;
        return k;
    };

    // gather role: (row gb, unit pair gp) x producer half gh
    const int gslot = tid & 127, gb = gslot >> 3, gpr = gslot & 7, gh = tid >> 7;
    const int PH = (P + 1) >> 1;
    const int pp_lo = gh * PH, pp_hi = min(P, pp_lo + PH);
    const long per_par = (long)ND * P * P * B * 8;

    Coef coef = make_coef(load_raw(0));
    Raw rawB = load_raw(1);
    Raw rawC = rawB;
    float carry = 0.f;
    float dgv[4] = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    for (int s = 0; s <= T; ++s) {
        auto io = [&]() {
            if (s > 0 && eok) {
                float* gp = ge + (long)tix(s - 1) * g_ts;
                gp[0] = dgv[0]; gp[H] = dgv[1]; gp[2 * (long)H] = dgv[2]; gp[3 * (long)H] = dgv[3];
            }
            rawC = load_raw(s + 2);
        };
        if (s == T) { io(); break; }
        // 1. recurrent dh of the owned slice: sum of the producers' partials (sequence of step s-1)
        if (s > 0) {
            float a0 = 0.f, a1 = 0.f;
            if (gb < B && pp_lo < pp_hi) {
                const u64* src = p.xbuf + (long)((s - 1) & 1) * per_par + (((long)d * P + me) * P) * B * 8 + gb * 8 + gpr;
                const u64 want = bwd_want(seq_of(s - 1));
                bool first = true;
                for (int pp0 = pp_lo; pp0 < pp_hi; pp0 += 10) {
                    u64 gr[10];
                    const int c = min(10, pp_hi - pp0);
                    if (first) gather_seq<10>(src + (long)pp0 * B * 8, (long)B * 8, c, BWD_MASK, want, gr, p.abort_flag, io);
                    else gather_seq<10>(src + (long)pp0 * B * 8, (long)B * 8, c, BWD_MASK, want, gr, p.abort_flag, []() {});
                    first = false;
#pragma unroll
                    for (int i = 0; i < 10; ++i)
                        if (i < c) {
                            a0 += __uint_as_float((unsigned)gr[i] & ~1u);
                            a1 += __uint_as_float((unsigned)(gr[i] >> 32) & ~1u);
                        }
                }
            } else {
                io();
            }
            *reinterpret_cast<float2*>(s_part + gh * 256 + gb * 16 + 2 * gpr) = make_float2(a0, a1);
            __syncthreads();
        } else {
            io();
        }
        // 2. cell backward of the owned element -> bf16 operand tile
        {
            float dh = coef.dy;
            if (s > 0) dh += s_part[tid] + s_part[256 + tid];
            const float dc = dh * coef.c1 + carry;
            dgv[0] = dc * coef.c2; dgv[1] = dc * coef.c3; dgv[2] = dc * coef.c4; dgv[3] = dh * coef.c5;
            carry = dc * coef.f;
            if (eok) {
                tile[eb * LD + ej] = (__bf16)dgv[0];
                tile[eb * LD + 16 + ej] = (__bf16)dgv[1];
                tile[eb * LD + 32 + ej] = (__bf16)dgv[2];
                tile[eb * LD + 48 + ej] = (__bf16)dgv[3];
            }
        }
        __syncthreads();
        // 3. partial dh_{prev}[b, k'] for every k', handed to the owner of k'
        if (s + 1 < T) {
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(tile + n * LD + 8 * q);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(tile + n * LD + 32 + 8 * q);
            u64* dst = p.xbuf + (long)(s & 1) * per_par + (long)d * P * P * B * 8;
            const u64 want = bwd_want(seq_of(s));
#pragma unroll
            for (int ot = 0; ot < NTO; ++ot) {
                const int tcol = wave + 4 * ot;
                if (tcol < P) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                    acc = mma16(w[ot][0], b0, acc);
                    acc = mma16(w[ot][1], b1, acc);
                    if (n < B) {
                        u64* o = dst + (((long)tcol * P + me) * B + n) * 8 + 2 * q;
                        const u64 v0 = (u64)(__float_as_uint(acc[0]) & ~1u) | ((u64)(__float_as_uint(acc[1]) & ~1u) << 32);
                        const u64 v1 = (u64)(__float_as_uint(acc[2]) & ~1u) | ((u64)(__float_as_uint(acc[3]) & ~1u) << 32);
                        st_gran(o, v0 | want);
                        st_gran(o + 1, v1 | want);
                    }
                }
            }
        }
        // 4. coefficients of the next step from the operands requested one step ago
        coef = make_coef(rawB);
        rawB = rawC;
    }
}

template <typename KernelT>
int launch_p2(KernelT kernel, const P2& p, size_t lds, hipStream_t st, const char* name) {
    hipLaunchKernelGGL(kernel, dim3(p.P, p.ND), dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("%s: launch failed: %s", name, hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

size_t lstm_persist2_workspace_bytes(int B, int H, int ND) {
    if (H % 16 != 0 || B > 16 || H > 512) return 0;
    const size_t P = H / 16;
    const size_t fwd = 2 * (size_t)ND * B * (H / 4) * sizeof(u64);
    const size_t bwd = 2 * (size_t)ND * P * P * B * 8 * sizeof(u64);
    return 256 + (fwd > bwd ? fwd : bwd);
}

#define FWD2_CASE(NKS_) \
    if (nks <= NKS_) return launch_p2(lstm_fwd_p2<NKS_>, p, 2 * 16 * (NKS_ * 32 + 8) * 2 + 4 * 16 * 20 * 4, st, "asr_lstm_fwd(persistent v2)");
#define BWD2_CASE(NTO_) \
    if (nto <= NTO_) return launch_p2(lstm_bwd_p2<NTO_>, p, 0, st, "asr_lstm_bwd(persistent v2)");

// Return ASR_OK when launched, 1 when the shape/precision has no v2 plan, negative on error.
int lstm_fwd_persistent2(float* gates, const float* whh, const float* bias2, float* y, float* c,
                         int B, int T, int H, int ND, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
    if (prec != ASR_BF16 || B > 16 || H % 16 != 0 || H > 512 || !ws) return 1;
    if (!aligned16(gates) || !aligned16(y) || !aligned16(c)) return 1;
    const size_t need = 256 + 2 * (size_t)ND * B * (H / 4) * sizeof(u64);
    if (ws_bytes < need) return 1;
    hipMemsetAsync(ws, 0, need, st);
    P2 p{gates, whh, bias2, y, c, (u64*)((char*)ws + 256), (unsigned*)ws, B, T, H, ND, H / 16};
    const int nks = (H + 31) / 32;
    FWD2_CASE(1) FWD2_CASE(2) FWD2_CASE(4) FWD2_CASE(6) FWD2_CASE(8) FWD2_CASE(10) FWD2_CASE(12) FWD2_CASE(16)
    return 1;
}

int lstm_bwd_persistent2(float* gates, const float* whh, const float* dy, const float* c,
                         int B, int T, int H, int ND, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
    if (prec != ASR_BF16 || B > 16 || H % 16 != 0 || H > 512 || !ws) return 1;
    const size_t P = H / 16;
    const size_t need = 256 + 2 * (size_t)ND * P * P * B * 8 * sizeof(u64);
    if (ws_bytes < need) return 1;
    hipMemsetAsync(ws, 0, need, st);
    P2 p{gates, whh, nullptr, const_cast<float*>(dy), const_cast<float*>(c), (u64*)((char*)ws + 256), (unsigned*)ws,
         B, T, H, ND, (int)P};
    const int nto = ((int)P + 3) / 4;
    BWD2_CASE(1) BWD2_CASE(2) BWD2_CASE(3) BWD2_CASE(4) BWD2_CASE(5) BWD2_CASE(6) BWD2_CASE(8)
    return 1;
}
