// Persistent form of the encoder LSTM recurrence (forward and BPTT): ONE launch walks all T steps.
//
// Per direction the hidden units are cut into P slices of HS = 16*NT units; workgroup (p, dir) keeps the
// W_hh rows of its 4*HS gate columns in registers for the whole sequence (bf16 or fp32 fragments) and the
// cell state c of its slice in registers.  What crosses workgroups every step is only
//   forward : h_t            (B x H per direction)    — all-gather
//   backward: partial dh_{t-1} (B x H per workgroup)  — reduce-scatter (each slice owner sums P partials)
// exchanged through global memory as 8-byte {tag = step+1, 32-bit payload} granules written with
// agent-scope relaxed atomic stores (write-through, sc1) and polled with agent-scope relaxed atomic loads:
// the data is its own flag, no fences, no separate barrier (cdna_hip_programming.md G16 form R2).
// Two granule buffers alternate by step parity; a producer can never be two steps ahead of a consumer
// because producing step s+1 needs every workgroup's step-s data.  All polled words are zeroed by a
// memset node before each launch; every spin is bounded and a shared abort word releases all waiters.
//
// Same arithmetic as lstm.hip's launch-per-step kernels (which remain the fallback for shapes that do not
// fit: H not a multiple of 16, too many rows for the LDS tiles, or more workgroups than CUs).
#include "common.h"

namespace {

typedef unsigned long long u64;
constexpr int SPIN_LIMIT = 1 << 22;

// Diagnostic build (-DASR_DIAG): workgroup (0,0) accumulates the wall time (100 MHz s_memrealtime ticks) it
// spends in each phase of the forward step into the status block (u64 words 2..9 of the workspace).
#ifdef ASR_DIAG
#define DIAG_DECL unsigned long long dg_t = __builtin_amdgcn_s_memrealtime(), dg_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long dg_c0 = __builtin_amdgcn_s_memtime();
#define DIAG_MARK(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); dg_acc[k] += n_ - dg_t; dg_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define DIAG_DUMP { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { unsigned long long* o = (unsigned long long*)p.abort_flag + 2; for (int k = 0; k < 8; ++k) o[k] = dg_acc[k]; o[8] = __builtin_amdgcn_s_memtime() - dg_c0; } }
#else
#define DIAG_DECL
#define DIAG_MARK(k)
#define DIAG_DUMP
#endif

struct PersistP {
    float* gates;        // (B,T,ND,4H)
    const float* whh;    // (ND,4H,H)
    const float* bias2;  // fwd: (ND,4H) or null
    float* y;            // fwd: h out (B,T,ND*H);  bwd: dy in
    float* c;            // (B,T,ND,H)
    u64* xbuf;           // granule exchange buffers
    unsigned* abort_flag;
    int B, T, H, ND, P, HS, MT;
};

__device__ __forceinline__ u64 ld_granule(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(u64* p, unsigned tag, unsigned payload) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Waits until granules base[0], base[stride], ..., base[(n-1)*stride] (n <= CH) all carry `tag` and returns their
// payloads.  All CH loads are issued back to back with no control flow between them (entries >= n re-read
// entry 0 and are not checked) so that ONE memory round trip covers the burst; the addresses are recomputed
// from (base, stride) every round instead of being kept in registers across the spin.
// (Polling one sentinel granule first and fetching the rest afterwards was measured SLOWER: two dependent
// sc1 round trips instead of one.)
template <int CH>
__device__ __forceinline__ void gather_granules(const u64* base, long stride, int n, unsigned tag, unsigned (&val)[CH],
                                                unsigned* abort_flag) {
    int spins = 0;
    while (true) {
        u64 g[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) g[i] = ld_granule(base + (i < n ? i : 0) * stride);
        bool ok = true;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            val[i] = (unsigned)g[i];
            ok = ok && ((i >= n) || ((unsigned)(g[i] >> 32) == tag));
        }
        if (ok) return;
        ++spins;
        if ((spins & 63) == 0) {
            if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;
            if (spins > SPIN_LIMIT) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

template <bool BF16> struct Frag;
template <> struct Frag<true> {
    typedef bf16x8 T;
    static __device__ __forceinline__ T zero() { T v; for (int j = 0; j < 8; ++j) v[j] = (__bf16)0.f; return v; }
    static __device__ __forceinline__ void set(T& v, int j, float x) { v[j] = (__bf16)x; }
    static __device__ __forceinline__ f32x4 mma(const T& a, const T& b, f32x4 c) { return mma16(a, b, c); }
};
struct F8 { float v[8]; };
template <> struct Frag<false> {
    typedef F8 T;
    static __device__ __forceinline__ T zero() { T v; for (int j = 0; j < 8; ++j) v.v[j] = 0.f; return v; }
    static __device__ __forceinline__ void set(T& v, int j, float x) { v.v[j] = x; }
    // slot q of f32 k-step j holds k = 32*ks + 8*q + j on both operands, so 8 MFMAs cover the 32-block
    static __device__ __forceinline__ f32x4 mma(const T& a, const T& b, f32x4 c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = mma16(a.v[j], b.v[j], c);
        return c;
    }
};

// LDS operand tile [rows][K] in the MFMA operand type, row stride padded by 16 bytes
template <bool BF16> __device__ __forceinline__ int tile_ld(int K) { return BF16 ? K + 8 : K + 4; }
template <bool BF16> __device__ __forceinline__ void tile_store(void* tile, int ld, int row, int k, float v) {
    if (BF16) reinterpret_cast<__bf16*>(tile)[row * ld + k] = (__bf16)v;
    else reinterpret_cast<float*>(tile)[row * ld + k] = v;
}
template <bool BF16> __device__ __forceinline__ typename Frag<BF16>::T tile_frag(const void* tile, int ld, int row, int k) {
    typename Frag<BF16>::T f;
    if constexpr (BF16) {
        f = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(tile) + row * ld + k);
    } else {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(tile) + row * ld + k);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(tile) + row * ld + k + 4);
        f.v[0] = a.x; f.v[1] = a.y; f.v[2] = a.z; f.v[3] = a.w; f.v[4] = b.x; f.v[5] = b.y; f.v[6] = b.z; f.v[7] = b.w;
    }
    return f;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Workgroup (p, dir) owns hidden units [64p, 64p+64); wave w the 16 units j = 64p + 16w + (lane&15).
// Each wave runs FOUR accumulators (gates i,f,g,o) over the same 16 columns, so a lane ends up with all
// four gate pre-activations of its (row, unit): the cell update needs no cross-lane traffic at all.
// Exchange buffer: xbuf[parity][dir][b][j]  (one fp32 h per granule).
template <bool FAST> __device__ __forceinline__ float act_sigmoid(float x) {
    return FAST ? 1.f / (1.f + __expf(-x)) : sigmoidf_(x);
}
template <bool FAST> __device__ __forceinline__ float act_tanh(float x) {
    return FAST ? 1.f - 2.f / (1.f + __expf(2.f * x)) : tanhf(x);
}

template <bool BF16, int NKS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_fwd_persist(PersistP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Frag<BF16> FR;
    const int H = p.H, T = p.T, ND = p.ND, B = p.B;
    const int d = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int j = blockIdx.x * 64 + 16 * wave + n;      // hidden unit of this lane's column
    const bool jok = j < H;
    const int jc = jok ? j : 0;
    const int ld = tile_ld<BF16>(NKS * 32);            // K padded to whole MFMA k-steps; the pad stays zero
    const size_t tile_bytes = (size_t)MT * 16 * ld * (BF16 ? 2 : 4);
    for (size_t i = tid; i < 2 * tile_bytes / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // resident weights: w[g][ks] = 8 consecutive k of W_hh row (g*H + j)
    typename FR::T w[4][NKS];
    float bias[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float* wrow = p.whh + ((long)d * 4 * H + (long)g * H + jc) * H;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            w[g][ks] = FR::zero();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = ks * 32 + 8 * q + e;
                FR::set(w[g][ks], e, (jok && k < H) ? wrow[k] : 0.f);
            }
        }
        bias[g] = (p.bias2 && jok) ? p.bias2[(long)d * 4 * H + (long)g * H + j] : 0.f;
    }
    float cst[MT][4];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) cst[a][r] = 0.f;

    // granule slots swept by this thread (step-invariant): idx = tid + 256*i  ->  LDS element offset.
    // bf16 mode packs two adjacent hidden units (even j, j+1) per granule; fp32 mode one value per granule.
    constexpr int PK = BF16 ? 2 : 1;
    const int HG = (H + PK - 1) / PK;          // granules per row
    const int total = B * HG;
    constexpr int NSLOT = 10;
    int slot_off[NSLOT];
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) {
        const int idx = tid + 256 * i;
        slot_off[i] = (idx < total) ? (idx / HG) * ld + (idx % HG) * PK : -1;
    }
    const long xstride = (long)ND * B * HG;   // granules per parity buffer
    __syncthreads();

    // software pipeline: xg (input-projection pre-activations) of step s+1 is loaded, and the bulk outputs of
    // step s-1 are stored, right AFTER the gather of step s — vmcnt retires loads and stores in issue order,
    // so anything issued before the poll would sit on the critical path of the hand-off.
    float xg[MT][4][4];
    float o_g[MT][4][4], o_c[MT][4], o_h[MT][4];
    auto load_xg = [&](int t, float (&dst)[MT][4][4]) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = mt * 16 + 4 * q + r;
                const long gi = (((long)(b < B ? b : 0) * T + t) * ND + d) * 4 * H + jc;
#pragma unroll
                for (int g = 0; g < 4; ++g) dst[mt][g][r] = (b < B && jok) ? p.gates[gi + (long)g * H] : 0.f;
            }
    };
    auto store_out = [&](int t) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = mt * 16 + 4 * q + r;
                if (b < B && jok) {
                    const long gi = (((long)b * T + t) * ND + d) * 4 * H + j;
#pragma unroll
                    for (int g = 0; g < 4; ++g) p.gates[gi + (long)g * H] = o_g[mt][g][r];
                    p.c[(((long)b * T + t) * ND + d) * H + j] = o_c[mt][r];
                    p.y[((long)b * T + t) * ND * H + (long)d * H + j] = o_h[mt][r];
                }
            }
    };
    load_xg((d == 0) ? 0 : T - 1, xg);
    DIAG_DECL

    for (int s = 0; s < T; ++s) {
        const int t = (d == 0) ? s : T - 1 - s;
        void* tile = smem + (size_t)(s & 1) * tile_bytes;
        DIAG_MARK(7)
        if (s > 0) {
            // gather h_{s-1} of this direction (tag s) into the LDS operand tile
            const u64* src = p.xbuf + (long)((s - 1) & 1) * xstride + (long)d * B * HG;
            {
                unsigned val[NSLOT];
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) if (slot_off[i] >= 0) cnt = i + 1;
                if (cnt > 0) gather_granules<NSLOT>(src + tid, 256, cnt, (unsigned)s, val, p.abort_flag);
                DIAG_MARK(0)
#pragma unroll
                for (int i = 0; i < NSLOT; ++i)
                    if (slot_off[i] >= 0) {
                        if (BF16) *reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(tile) + slot_off[i]) = val[i];   // two bf16
                        else reinterpret_cast<float*>(tile)[slot_off[i]] = __uint_as_float(val[i]);
                    }
            }
            for (int c0 = 256 * NSLOT; c0 < total; c0 += 256 * NSLOT) {   // further chunks (large batches)
                unsigned val[NSLOT];
                int cnt = 0;
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) if (c0 + tid + 256 * i < total) cnt = i + 1;
                if (cnt > 0) gather_granules<NSLOT>(src + c0 + tid, 256, cnt, (unsigned)s, val, p.abort_flag);
#pragma unroll
                for (int i = 0; i < NSLOT; ++i) {
                    const int idx = c0 + tid + 256 * i;
                    if (idx < total) {
                        const int off = (idx / HG) * ld + (idx % HG) * PK;
                        if (BF16) *reinterpret_cast<unsigned*>(reinterpret_cast<__bf16*>(tile) + off) = val[i];
                        else reinterpret_cast<float*>(tile)[off] = __uint_as_float(val[i]);
                    }
                }
            }
            DIAG_MARK(1)
        }
        DIAG_MARK(2)
        __syncthreads();
        DIAG_MARK(3)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            f32x4 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) acc[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (s > 0) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const typename FR::T a = tile_frag<BF16>(tile, ld, mt * 16 + n, ks * 32 + 8 * q);
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = FR::mma(a, w[g][ks], acc[g]);
                }
            }
            asm volatile("" :: "v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]), "v"(acc[3][0]));
            DIAG_MARK(4)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int b = mt * 16 + 4 * q + r;
                const float gi_ = act_sigmoid<BF16>(xg[mt][0][r] + acc[0][r] + bias[0]);
                const float gf = act_sigmoid<BF16>(xg[mt][1][r] + acc[1][r] + bias[1]);
                const float gg = act_tanh<BF16>(xg[mt][2][r] + acc[2][r] + bias[2]);
                const float go = act_sigmoid<BF16>(xg[mt][3][r] + acc[3][r] + bias[3]);
                const float cn = gf * cst[mt][r] + gi_ * gg;
                cst[mt][r] = cn;
                const float hv = go * act_tanh<BF16>(cn);
                if (BF16) {
                    // even lanes publish (h_j, h_{j+1}) as two bf16; the odd neighbour's value comes over a lane swap
                    const float hn = __shfl_xor(hv, 1);
                    if (b < B && jok && s + 1 < T && (n & 1) == 0) {
                        const unsigned lo = f2bf_bits(hv), hi = (j + 1 < H) ? f2bf_bits(hn) : 0u;
                        st_granule(p.xbuf + (long)(s & 1) * xstride + ((long)d * B + b) * HG + (j >> 1), (unsigned)(s + 1), lo | (hi << 16));
                    }
                } else if (b < B && jok && s + 1 < T) {
                    st_granule(p.xbuf + (long)(s & 1) * xstride + ((long)d * B + b) * HG + j, (unsigned)(s + 1), __float_as_uint(hv));
                }
                o_g[mt][0][r] = gi_; o_g[mt][1][r] = gf; o_g[mt][2][r] = gg; o_g[mt][3][r] = go;
                o_c[mt][r] = cn; o_h[mt][r] = hv;
            }
            DIAG_MARK(5)
        }
        // bulk outputs of this step and the input-projection operands of the next one: issued between the
        // publish and the next poll, so their latency overlaps the hand-off instead of following it
        store_out(t);
        if (s + 1 < T) load_xg((d == 0) ? t + 1 : t - 1, xg);
        DIAG_MARK(6)
    }
    DIAG_DUMP
}

// ------------------------------------------------------------------------------------------------
// backward (BPTT)
// ------------------------------------------------------------------------------------------------
// Workgroup (p, dir): slice units jl in [0,HS).  Reduction index of its MFMA = its own gate columns
// n_local = g*HS + jl (K = 4*HS); output columns = all H hidden units k' (partial dh_{prev}[b,k']).
// Exchange: xbuf[parity][dir][consumer pc][producer pp][b][jl]  fp32 granules.
template <bool BF16, int NTO, int NKS, int NE>
__global__ __launch_bounds__(256, 1) void lstm_bwd_persist(PersistP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef Frag<BF16> FR;
    const int H = p.H, T = p.T, ND = p.ND, B = p.B, HS = p.HS, P = p.P;
    const int d = blockIdx.y, me = blockIdx.x, j0 = me * HS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, q = lane >> 4;
    const int K = 4 * HS;
    const int lg = 31 - __builtin_clz(HS);               // HS is a power of two
    const int ld = tile_ld<BF16>(K);
    void* tile = smem;                                                        // [MT*16][K] dgates of this slice
    float* s_dh = reinterpret_cast<float*>(smem + (size_t)p.MT * 16 * ld * (BF16 ? 2 : 4));   // [B][HS] recurrent dh

    // resident weights: output tile ot (wave-strided) covers k' = 16*(wave + 4*ot) + n;
    // fragment element e of k-step ks is W_hh[g*H + j0 + jl][k'] with n_local = 32*ks + 8*q + e = g*HS + jl
    typename FR::T w[NTO][NKS];
    const int ntiles = H / 16;
#pragma unroll
    for (int ot = 0; ot < NTO; ++ot) {
        const int tcol = wave + 4 * ot;
        const int kp = tcol * 16 + n;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
            w[ot][ks] = FR::zero();
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int nl = ks * 32 + 8 * q + e;
                float v = 0.f;
                if (tcol < ntiles && nl < K) v = p.whh[((long)d * 4 * H + (long)(nl >> lg) * H + j0 + (nl & (HS - 1))) * H + kp];
                FR::set(w[ot][ks], e, v);
            }
        }
    }
    for (int i = tid; i < p.MT * 16 * ld * (BF16 ? 2 : 4) / 4; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;

    // thread-owned elements e = tid + 256*i (i < NE): (row b, slice unit jl)
    const int nelem = B * HS;
    const long per_par = (long)ND * P * P * B * HS;
    float carry[NE], coef[NE][7], raw[NE][7], dgv[NE][4];
#pragma unroll
    for (int i = 0; i < NE; ++i) carry[i] = 0.f;

    // raw operands of the cell backward at time index t (everything that does not depend on the recurrent dh)
    auto load_raw = [&](int t) {
        const int tp = (d == 0) ? t - 1 : t + 1;
        const bool has_cprev = (d == 0) ? (t > 0) : (t < T - 1);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            if (e < nelem) {
                const int b = e >> lg, j = j0 + (e & (HS - 1));
                const long gi = (((long)b * T + t) * ND + d) * 4 * H + j;
                raw[i][0] = p.y[((long)b * T + t) * ND * H + (long)d * H + j];
                raw[i][1] = p.gates[gi]; raw[i][2] = p.gates[gi + H]; raw[i][3] = p.gates[gi + 2 * (long)H]; raw[i][4] = p.gates[gi + 3 * (long)H];
                raw[i][5] = p.c[(((long)b * T + t) * ND + d) * H + j];
                raw[i][6] = has_cprev ? p.c[(((long)b * T + tp) * ND + d) * H + j] : 0.f;
            }
        }
    };
    auto make_coef = [&]() {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const float gi_ = raw[i][1], gf = raw[i][2], gg = raw[i][3], go = raw[i][4];
            const float tc = act_tanh<BF16>(raw[i][5]);
            coef[i][0] = raw[i][0];
            coef[i][1] = go * (1.f - tc * tc);          // d c / d h
            coef[i][2] = gg * gi_ * (1.f - gi_);        // d i_pre / d c
            coef[i][3] = raw[i][6] * gf * (1.f - gf);   // d f_pre / d c
            coef[i][4] = gi_ * (1.f - gg * gg);         // d g_pre / d c
            coef[i][5] = tc * go * (1.f - go);          // d o_pre / d h
            coef[i][6] = gf;
        }
    };
    load_raw((d == 0) ? T - 1 : 0);
    make_coef();
    __syncthreads();

    for (int s = 0; s < T; ++s) {
        const int t = (d == 0) ? T - 1 - s : s;
        // 1. recurrent dh of this slice = sum over producers of their partials (tag s): thread-private LDS slots
        if (s > 0) {
            const u64* src = p.xbuf + (long)((s - 1) & 1) * per_par + (((long)d * P + me) * P) * B * HS;
            for (int e0 = tid; e0 < nelem; e0 += 256) {          // one owned element x up to 10 producers per burst
                float a0 = 0.f;
                for (int pp0 = 0; pp0 < P; pp0 += 10) {
                    unsigned val[10];
                    const int cnt = min(10, P - pp0);
                    gather_granules<10>(src + (long)pp0 * B * HS + e0, (long)B * HS, cnt, (unsigned)s, val, p.abort_flag);
#pragma unroll
                    for (int i = 0; i < 10; ++i) if (i < cnt) a0 += __uint_as_float(val[i]);
                }
                s_dh[e0] = a0;
            }
        }
        // operands of the NEXT step are requested now (after the poll, vmcnt retires in issue order)
        if (s + 1 < T) load_raw((d == 0) ? t - 1 : t + 1);
        // 2. cell backward for the owned elements -> LDS operand tile (global stores are deferred)
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            if (e < nelem) {
                const int b = e >> lg, jl = e & (HS - 1);
                float dh = coef[i][0];
                if (s > 0) dh += s_dh[e];
                const float dc = dh * coef[i][1] + carry[i];
                dgv[i][0] = dc * coef[i][2]; dgv[i][1] = dc * coef[i][3]; dgv[i][2] = dc * coef[i][4]; dgv[i][3] = dh * coef[i][5];
                carry[i] = dc * coef[i][6];
                tile_store<BF16>(tile, ld, b, jl, dgv[i][0]);
                tile_store<BF16>(tile, ld, b, HS + jl, dgv[i][1]);
                tile_store<BF16>(tile, ld, b, 2 * HS + jl, dgv[i][2]);
                tile_store<BF16>(tile, ld, b, 3 * HS + jl, dgv[i][3]);
            }
        }
        __syncthreads();
        // 3. partial dh_{prev}[b, k'] for all k', published to the owner of k'
        if (s + 1 < T) {
            u64* dst = p.xbuf + (long)(s & 1) * per_par + (long)d * P * P * B * HS;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                if (mt < p.MT) {
                    typename FR::T a[NKS];
#pragma unroll
                    for (int ks = 0; ks < NKS; ++ks) a[ks] = tile_frag<BF16>(tile, ld, mt * 16 + n, ks * 32 + 8 * q);
#pragma unroll
                    for (int ot = 0; ot < NTO; ++ot) {
                        const int tcol = wave + 4 * ot;
                        if (tcol < ntiles) {
                            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int ks = 0; ks < NKS; ++ks) acc = FR::mma(a[ks], w[ot][ks], acc);
                            const int kp = tcol * 16 + n;
                            const int pc = kp >> lg, jl = kp & (HS - 1);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int b = mt * 16 + 4 * q + r;
                                if (b < B)
                                    st_granule(dst + (((long)pc * P + me) * B + b) * HS + jl, (unsigned)(s + 1), __float_as_uint(acc[r]));
                            }
                        }
                    }
                }
            }
        }
        // 4. deferred: gradients wrt the gate pre-activations to global (in place of the saved gates),
        //    then the coefficients of the next step from the operands requested above
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = tid + 256 * i;
            if (e < nelem) {
                const int b = e >> lg, j = j0 + (e & (HS - 1));
                const long gi = (((long)b * T + t) * ND + d) * 4 * H + j;
                p.gates[gi] = dgv[i][0]; p.gates[gi + H] = dgv[i][1]; p.gates[gi + 2 * (long)H] = dgv[i][2]; p.gates[gi + 3 * (long)H] = dgv[i][3];
            }
        }
        if (s + 1 < T) make_coef();
        __syncthreads();   // the operand tile and s_dh are rewritten by the next step
    }
}

struct Plan { bool ok; int NT, NKS, P, HS, MT; size_t lds; };

Plan plan_fwd(int B, int H, bool bf16) {
    Plan pl{false, 0, 0, 0, 0, 0, 0};
    if (B > 64 || H > 320) return pl;
    const int nks = (H + 31) / 32;
    pl.NT = 4; pl.NKS = nks; pl.HS = 64; pl.P = (H + 63) / 64; pl.MT = (B + 15) / 16;
    if (pl.MT == 3) pl.MT = 4;
    const int ld = bf16 ? nks * 32 + 8 : nks * 32 + 4;
    pl.lds = 2 * (size_t)pl.MT * 16 * ld * (bf16 ? 2 : 4);
    pl.ok = pl.lds <= 150 * 1024;
    return pl;
}

Plan plan_bwd(int B, int H, bool bf16) {
    Plan pl{false, 0, 0, 0, 0, 0, 0};
    if (H % 16 != 0 || B > 64) return pl;
    // slice size HS: K = 4*HS must be a multiple of 32; output tiles per wave NTO = ceil(H/16/4)
    const int nto = (H / 16 + 3) / 4;
    int HS = 0;
    for (int c : {64, 32, 16, 8}) {
        const int nks = (4 * c) / 32;
        if (H % c == 0 && nks >= 1 && nto * nks <= 20) { HS = c; break; }
    }
    if (!HS) return pl;
    pl.HS = HS; pl.P = H / HS; pl.NT = nto; pl.NKS = (4 * HS) / 32; pl.MT = (B + 15) / 16;
    const int K = 4 * HS, ld = bf16 ? K + 8 : K + 4;
    pl.lds = (size_t)pl.MT * 16 * ld * (bf16 ? 2 : 4) + (size_t)B * HS * 4;
    pl.ok = pl.lds <= 150 * 1024 && pl.P * 2 <= 200 && B * HS <= 256 * 8;
    return pl;
}

template <typename KernelT>
int launch_persist(KernelT kernel, const PersistP& p, size_t lds, hipStream_t st, const char* name) {
    if (lds > 64 * 1024) hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kernel, dim3(p.P, p.ND), dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("%s: launch failed: %s", name, hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

#define FWD_CASE(BF, NKS_)                                                                                          \
    if (pl.NKS == NKS_) {                                                                                           \
        if (pl.MT == 1) return launch_persist(lstm_fwd_persist<BF, NKS_, 1>, p, pl.lds, st, "asr_lstm_fwd(persistent)"); \
        if (pl.MT == 2) return launch_persist(lstm_fwd_persist<BF, NKS_, 2>, p, pl.lds, st, "asr_lstm_fwd(persistent)"); \
        if (pl.MT <= 4) return launch_persist(lstm_fwd_persist<BF, NKS_, 4>, p, pl.lds, st, "asr_lstm_fwd(persistent)"); \
    }
#define BWD_CASE(BF, NTO_, NKS_)                                                                                              \
    if (pl.NT == NTO_ && pl.NKS == NKS_) {                                                                                    \
        const int ne = (p.B * p.HS + 255) / 256;                                                                              \
        if (ne <= 1) return launch_persist(lstm_bwd_persist<BF, NTO_, NKS_, 1>, p, pl.lds, st, "asr_lstm_bwd(persistent)");    \
        if (ne <= 2) return launch_persist(lstm_bwd_persist<BF, NTO_, NKS_, 2>, p, pl.lds, st, "asr_lstm_bwd(persistent)");    \
        if (ne <= 4) return launch_persist(lstm_bwd_persist<BF, NTO_, NKS_, 4>, p, pl.lds, st, "asr_lstm_bwd(persistent)");    \
        return launch_persist(lstm_bwd_persist<BF, NTO_, NKS_, 8>, p, pl.lds, st, "asr_lstm_bwd(persistent)");                 \
    }

template <bool BF>
int dispatch_fwd(const Plan& pl, const PersistP& p, hipStream_t st) {
    FWD_CASE(BF, 1) FWD_CASE(BF, 2) FWD_CASE(BF, 3) FWD_CASE(BF, 4) FWD_CASE(BF, 5) FWD_CASE(BF, 6) FWD_CASE(BF, 8) FWD_CASE(BF, 10)
    return 1;   // no instantiation: caller falls back
}
template <bool BF>
int dispatch_bwd(const Plan& pl, const PersistP& p, hipStream_t st) {
    // (NTO, NKS): H=16 -> (1,*), H=32 -> (1,*), H=128 -> (2,8), H=320 -> (5,8), H=256 -> (4,8), H=512 -> (8,8)
    BWD_CASE(BF, 1, 1) BWD_CASE(BF, 1, 2) BWD_CASE(BF, 1, 4) BWD_CASE(BF, 1, 8) BWD_CASE(BF, 2, 8) BWD_CASE(BF, 4, 4) BWD_CASE(BF, 5, 4)
    BWD_CASE(BF, 3, 4) BWD_CASE(BF, 2, 4)
    return 1;
}

}  // namespace

// Returns ASR_OK when the persistent kernel was launched, 1 when the shape has no persistent plan
// (caller uses the launch-per-step path), negative on error.
int lstm_fwd_persistent(float* gates, const float* whh, const float* bias2, float* y, float* c,
                        int B, int T, int H, int ND, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool bf = prec == ASR_BF16;
    Plan pl = plan_fwd(B, H, bf);
    if (!pl.ok || !ws) return 1;
    const size_t need = 256 + 2 * (size_t)ND * B * H * sizeof(u64);
    if (ws_bytes < need) return 1;
    hipMemsetAsync(ws, 0, need, st);
    PersistP p{gates, whh, bias2, y, c, (u64*)((char*)ws + 256), (unsigned*)ws, B, T, H, ND, pl.P, pl.HS, pl.MT};
    return bf ? dispatch_fwd<true>(pl, p, st) : dispatch_fwd<false>(pl, p, st);
}

int lstm_bwd_persistent(float* gates, const float* whh, const float* dy, const float* c,
                        int B, int T, int H, int ND, int prec, void* ws, size_t ws_bytes, hipStream_t st) {
    const bool bf = prec == ASR_BF16;
    Plan pl = plan_bwd(B, H, bf);
    if (!pl.ok || !ws) return 1;
    const size_t need = 256 + 2 * (size_t)ND * pl.P * pl.P * B * pl.HS * sizeof(u64);
    if (ws_bytes < need) return 1;
    hipMemsetAsync(ws, 0, need, st);
    PersistP p{gates, whh, nullptr, const_cast<float*>(dy), const_cast<float*>(c), (u64*)((char*)ws + 256), (unsigned*)ws,
               B, T, H, ND, pl.P, pl.HS, pl.MT};
    return bf ? dispatch_bwd<true>(pl, p, st) : dispatch_bwd<false>(pl, p, st);
}

size_t lstm_persist_workspace_bytes(int B, int H, int ND) {
    // upper bound of both passes: backward exchanges P partial slices per consumer (P*P*B*HS = P*B*H granules)
    const int Pmax = H / 8 > 0 ? H / 8 : 1;
    size_t fwd = 2 * (size_t)ND * B * H * sizeof(u64);
    Plan pb = plan_bwd(B, H, true);
    Plan pf = plan_bwd(B, H, false);
    int P = 1;
    if (pb.ok) P = pb.P;
    if (pf.ok && pf.P > P) P = pf.P;
    (void)Pmax;
    size_t bwd = 2 * (size_t)ND * P * B * H * sizeof(u64);
    return 256 + (fwd > bwd ? fwd : bwd);
}
