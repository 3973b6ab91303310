// Shared pieces of the decoder BACKWARD cluster kernels (decoder_persist.hip: dec_bwd_persist, tiles of <= 40 frames resident
// in LDS / registers; decoder_stream.hip: dec_bwd_stream, tiles of any size walked in 48-frame groups): constants, the
// transposed-weight product macros, the MFMA-layout energy-backward sweep.  `static` per translation unit.
#pragma once
#include "decoder_cluster.h"

namespace {

constexpr int NPB = 3;            // polling waves of the backward kernel
constexpr int RCB = 8, RPB = 9;   // rows of the transposed-weight product per compute wave / per polling wave
constexpr int RPWB = 12;          // (s_out is sized for RPWB * 8 >= RCB * ncw + RPB * NPB outputs)
constexpr int KCHB = 5;           // 4-column chunks of the gate-gradient vector per lane (4*Dd <= 1280)
constexpr int UQW = 4;            // hidden units per compute wave in the query-part product

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
__device__ __forceinline__ float dot2bf(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_, a), __builtin_bit_cast(bf16x2_, b), c, false);
}

struct PB {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const unsigned short* enc16;
    const int64_t* enc_len;
    const float* dhs;               // (B,L,Dd) gradient wrt h_t from the output layer
    float* dxin;                    // (B,L,Dd+E)  context part written here
    float* dq;                      // (B,L,A)     gradient wrt the query pre-activation
    float* dkey;                    // (B,T',A)    zero on entry, accumulated atomically
    float* slots;                   // (B*NT, slot)  d w_g [a], d W_proj [k][a], d b_g
    float* dgates;                  // (B,L,4Dd)   gate pre-activation gradients (the saved gates stay intact)
    const unsigned short* wcatT16;  // (Dd+E+Dd rows = input columns) x R4 bf16, row x = gradient weights of input column x
    const float* wqT;               // (Dd x A)
    u64* xbuf;
    unsigned* status;
    int slot, NT, TE, UPW, CPW, R4;
    int CG2, QG2, VG2, NG2;         // granules per producer record (even)
    int allow_local;
    int poll_delay;                 // s_sleep units before the Q/V polling starts
};

// LDS carve of dec_bwd_persist, shared by the kernel and the host plan (float offsets follow the bf16 arrays).
struct BCarve { int AP, DW, PADL, WT, AQ, key, dl, wp16, dg16, wq16, cvx, cvT, shorts; int wc, crec, qst, nrec, dcp, de, out, hq, pt, dcx, dq, floats; };
constexpr int SW_MT = 3;          // 16-frame tiles of the sweep (TE <= 40: the third one is ragged)
constexpr int SW_NU = 3;          // 16-column units of the sweep per compute wave (A <= 320 over ncw + 3 waves)
constexpr int SW_NUP = 2;         //   and per polling wave (unit u -> wave u % (ncw + 3): wave index >= ncw gets at most two)
constexpr int CVX_LD = 32;        // row of the split-bf16 conv tile = the K slots of one MFMA
constexpr int CVT_LD = 16 * SW_MT + 8;
__host__ __device__ inline BCarve bwd_carve(int TE, int KP, int A, int E, int Kn, int Ks, int NT, int UPW, int CG2, int QG2, int NG2) {
    BCarve c;
    int ap8 = 8 * ((A + 63) / 64); if ((ap8 & 1) == 0) ++ap8;
    c.AP = 8 * ap8;                                     // row stride of the [frame][a] tiles: >= 64*ceil(A/64), odd in 16-byte units
    c.PADL = Ks + 8 + ((4 - ((2 * Ks) & 3)) & 3);       // left zero pad of a dconv row: PADL + Ks is a multiple of 4
    c.DW = (c.PADL + NT * TE + Ks + 8 + 3) & ~3;        // zero-padded dconv row,
    if (((c.DW >> 2) & 1) == 0) c.DW += 4;              //   an odd number of 16-byte units (bank spread across the Kn rows)
    c.WT = (2 * Ks + 1 + 3) & ~3;                       // zero-padded filter row
    int o = 0;
    c.key = o; o += (TE * A + 7) & ~7;
    c.dl = o; o += TE * c.AP;
    c.wp16 = o; o += 16 * c.AP;
    c.dg16 = o; o += 64 * KCHB * 4;
    c.AQ = 64 * ((A + 63) / 64);                        // row of the resident W_q^T slice
    c.wq16 = o; o += ((UPW + 1) & ~1) * c.AQ;
    c.cvx = o; o += 16 * SW_MT * CVX_LD;                // [48][32] conv tile of the step, {hi, lo, hi} slots
    c.cvT = o; o += 16 * CVT_LD;                        // [16][56] the same tile transposed (hi only)
    c.shorts = o;
    o = 0;
    c.wc = o; o += Kn * c.WT;
    c.crec = o; o += NT * CG2 * 2;
    c.qst = o; o += NT * QG2 * 2;
    c.nrec = o; o += NT * NG2 * 2 + 8;
    c.dcp = o; o += Kn * c.DW;
    c.de = o; o += 16 * SW_MT;                          // [48], rows >= TE stay 0
    c.out = o; o += RPWB * 8;
    c.hq = o; o += 64;
    c.pt = o; o += 4 * Kn * TE;
    c.dcx = o; o += (E + 3) & ~3;
    c.dq = o; o += (A + 3) & ~3;
    c.floats = o;
    return c;
}

__device__ __forceinline__ void cbar(unsigned* cnt, unsigned& gen, int nw) {
    gen += nw;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// rows of [W_ih(ctx) | W_hh]^T (registers) . dgates (bf16, LDS): output oo = BASE + STRIDE*o, o < NR -> s_out[oo]
#define DPB_P1(NR, BASE, STRIDE)                                                                                       \
    {                                                                                                                  \
        float acc_[NR];                                                                                                \
        _Pragma("unroll") for (int o = 0; o < NR; ++o) acc_[o] = 0.f;                                                  \
        _Pragma("unroll") for (int k = 0; k < KCHB; ++k) {                                                             \
            const uint2 g_ = *reinterpret_cast<const uint2*>(s_dg16 + 4 * (lane + 64 * k));                            \
            _Pragma("unroll") for (int o = 0; o < NR; ++o) {                                                           \
                acc_[o] = dot2bf(wreg[o][k].x, g_.x, acc_[o]);                                                         \
                acc_[o] = dot2bf(wreg[o][k].y, g_.y, acc_[o]);                                                         \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int o = 0; o < NR; ++o) {                                                               \
            const float sv_ = wave_sum_dpp(acc_[o]);                                                                   \
            if (lane == 0 && (BASE) + (STRIDE) * o < nout) s_out[(BASE) + (STRIDE) * o] = sv_;                         \
        }                                                                                                              \
    }
// loads the rows of this wave (same mapping) into wreg[NR][KCHB]
#define DPB_WLOAD(NR, BASE, STRIDE)                                                                                    \
    _Pragma("unroll") for (int o = 0; o < NR; ++o) {                                                                   \
        const int oo = (BASE) + (STRIDE) * o;                                                                          \
        int x = (oo < p.CPW) ? Dd + min(c_base + oo, E - 1) : XW + min(u_base + (oo - p.CPW), Dd - 1);                  \
        if (oo >= nout) x = Dd;                                                                                        \
        _Pragma("unroll") for (int k = 0; k < KCHB; ++k) {                                                             \
            const int col = 4 * (lane + 64 * k);                                                                       \
            const uint2 v = *reinterpret_cast<const uint2*>(p.wcatT16 + (long)x * R4 + min(col, R4 - 4));              \
            wreg[o][k] = (col < R4) ? v : make_uint2(0u, 0u);                                                          \
        }                                                                                                              \
    }

// ---- energy-backward sweep on the matrix cores ---------------------------------------------------------------------
// The tile's (frame f, attention column a) plane is cut into 16 x 16 MFMA tiles; a wave owns up to SW_NU column units (unit
// u -> wave u % nw, all SW_MT frame tiles of it) for the whole launch, and a lane holds the MFMA result layout of each tile:
// column a = 16 u + (lane & 15), frames f = 16 mt + 4 (lane >> 4) + r, r < 4.  Per tile
//   lp   = conv(f, :) . W_proj(a, :)          one v_mfma_f32_16x16x32_bf16: the K slots carry {hi.hi, lo.hi, hi.lo} of the
//                                              split-bf16 operands, i.e. fp32-grade products from one instruction
//   loc = tanh(lp), u = tanh(key + q + loc), du = de w_g (1 - u^2), dl = du (1 - loc^2)          (4 elements per lane)
//   d W_proj(a, :) += dl(:, a)^T . conv       one v_mfma_f32_16x16x16_bf16: the four dl values of a lane ARE its A fragment
// and everything that was a per-step read-modify-write before stays in registers for all L steps: dkey (4 floats per tile -
// no atomics, one plain store at the end), d w_g, d W_proj (MFMA accumulators).  dl goes to s_dl (bf16) for the dconv
// product; the query gradient is summed over the lane's frames here and over the four lane groups by the caller.
typedef __attribute__((ext_vector_type(4))) short s16x4_;
template <int NU> struct Sweep {
    bf16x8 wpx[NU];                 // B fragment of the lp product: W_proj(a, :) in the {hi, hi, lo} slots
    f32x4 dwp[NU];                  // d W_proj accumulator: rows a = 16 u + 4 (lane >> 4) + r, column k = lane & 15
    float dk[NU][SW_MT][4];         // dkey of the lane's elements
    float dwg[NU], wg[NU];
};

template <int KNMAX, int NU>
__device__ __forceinline__ void sweep_init(Sweep<NU>& S, const float* __restrict__ Wproj, const float* __restrict__ wg, int A, int Kn, int wave, int nw, int lane) {
    const int q = lane >> 4, c = lane & 15;
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) {
        const int a = 16 * (wave + nw * nu) + c;
        const bool ok = a < A;
        const float* wr = Wproj + (long)min(a, A - 1) * Kn;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int slot = 8 * q + i;
            const int k = slot < KNMAX ? slot : (slot < 2 * KNMAX ? slot - KNMAX : slot - 2 * KNMAX);
            const float w = (ok && slot < 3 * KNMAX && k < Kn) ? wr[min(k, Kn - 1)] : 0.f;
            const __bf16 hi = (__bf16)w;
            S.wpx[nu][i] = (slot < 2 * KNMAX) ? hi : (__bf16)(w - (float)hi);
        }
        S.dwp[nu] = f32x4{0.f, 0.f, 0.f, 0.f};
        S.dwg[nu] = 0.f;
        S.wg[nu] = ok ? wg[min(a, A - 1)] : 0.f;                    // 0: the pad columns contribute nothing
#pragma unroll
        for (int mt = 0; mt < SW_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) S.dk[nu][mt][r] = 0.f;
    }
}

// one step; dq[nu] = sum of du over the lane's frames.  nu_cnt (units of this wave) and MT are wave-uniform.
template <int NU>
__device__ __forceinline__ void sweep_step(Sweep<NU>& S, const float (&qa)[NU], float (&dq)[NU], int nu_cnt, int wave, int nw, int MT, int TE, int A, int AP,
                                           const unsigned short* s_cvx, const unsigned short* s_cvT, const float* s_de, const unsigned short* s_key,
                                           unsigned short* s_dl, int lane) {
    // The LDS addresses below (36 s_dl writes, 9 key reads, ...) are loop invariants of the time loop: left alone, the compiler
    // hoists them into as many live registers and spills the accumulators instead.  An opaque zero ties them to the step.
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const int q = (lane >> 4) + opaque, c = lane & 15;
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) dq[nu] = 0.f;
#pragma unroll
    for (int mt = 0; mt < SW_MT; ++mt) {
        if (mt < MT) {
            const int f0 = 16 * mt + 4 * q;
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(s_cvx + (16 * mt + c) * CVX_LD + 8 * q);
            const s16x4_ bv = *reinterpret_cast<const s16x4_*>(s_cvT + c * CVT_LD + f0);
            const float4 de4 = *reinterpret_cast<const float4*>(s_de + f0);
            const float de[4] = {de4.x, de4.y, de4.z, de4.w};
            const int fg = min(f0, TE - 4) >> 2;                        // frames >= TE: de = 0, any finite key will do
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                if (nu < nu_cnt) {
                    const int a = 16 * (wave + nw * nu) + c;
                    const f32x4 lp = mma16(av, S.wpx[nu], f32x4{0.f, 0.f, 0.f, 0.f});
                    const uint2 kb = *reinterpret_cast<const uint2*>(s_key + ((long)fg * A + min(a, A - 1)) * 4);
                    const float key[4] = {__uint_as_float(kb.x << 16), __uint_as_float(kb.x & 0xffff0000u),
                                          __uint_as_float(kb.y << 16), __uint_as_float(kb.y & 0xffff0000u)};
                    bf16x4 dl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float loc = tanh_f(lp[r]);
                        const float u = tanh_f(key[r] + qa[nu] + loc);
                        const float du = de[r] * S.wg[nu] * (1.f - u * u);
                        S.dwg[nu] += de[r] * u;
                        dq[nu] += du;
                        S.dk[nu][mt][r] += du;
                        dl[r] = (__bf16)(du * (1.f - loc * loc));
                    }
                    const s16x4_ dls = __builtin_bit_cast(s16x4_, dl);
                    if (f0 < TE) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) s_dl[(f0 + r) * AP + a] = (unsigned short)dls[r];
                    }
                    S.dwp[nu] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dls, bv, S.dwp[nu], 0, 0, 0);
                }
            }
        }
    }
}

// sum over the four lane groups (lanes c, c+16, c+32, c+48); every lane gets the total
__device__ __forceinline__ float sum_groups(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// sweep of this wave's units, then the Q record: the tile's query-gradient partial times (1 - q^2), pairs (a, a+1) from even lanes
#define DPB_SWEEP_AND_PUBLISH(XB, NU)                                                                                      \
    {                                                                                                                  \
        float dqp_[NU];                                                                                                \
        if (tau0 < len) sweep_step(S, qa, dqp_, nu_cnt, wave, nw, MT, TE, A, AP, s_cvx, s_cvT, s_de, s_key, s_dl, lane); \
        else { _Pragma("unroll") for (int nu = 0; nu < NU; ++nu) dqp_[nu] = 0.f; }                                     \
        _Pragma("unroll") for (int nu = 0; nu < NU; ++nu) {                                                            \
            if (nu < nu_cnt) {                                                                                         \
                const int a_ = 16 * (wave + nw * nu) + csub;                                                           \
                const float tot_ = sum_groups(dqp_[nu]);                                                               \
                const float mine_ = (a_ < A) ? tot_ * (1.f - qa[nu] * qa[nu]) : 0.f;                                   \
                const float nb_ = __shfl_down(mine_, 1);                                                               \
                if (qsub == 0 && (lane & 1) == 0 && a_ < 2 * p.QG2) {                                                  \
                    u64* dst_ = (XB) + offQ + (long)j * p.QG2 + (a_ >> 1);                           \
                    if (local) publish<true>(dst_, pack2(mine_, nb_, want)); else publish<false>(dst_, pack2(mine_, nb_, want)); \
                }                                                                                                      \
            }                                                                                                          \
        }                                                                                                              \
    }
// end of the launch: d w_g and d W_proj of the wave's units -> the workgroup's slot, dkey -> HBM (plain stores, once)
#define DPB_SWEEP_RESULTS(NU)                                                                                          \
    {                                                                                                                  \
        float* sl_ = p.slots + ((long)b * NT + j) * p.slot;                                                            \
        _Pragma("unroll") for (int nu = 0; nu < NU; ++nu) {                                                            \
            if (nu < nu_cnt) {                                                                                         \
                const int u0_ = 16 * (wave + nw * nu);                                                                 \
                const float g_ = sum_groups(S.dwg[nu]);                                                                \
                if (qsub == 0 && u0_ + csub < A) sl_[u0_ + csub] = g_;                                                 \
                _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                        \
                    const int a_ = u0_ + 4 * qsub + r;                                                                 \
                    if (csub < Kn && a_ < A) sl_[A + csub * A + a_] = S.dwp[nu][r];                                    \
                }                                                                                                      \
                _Pragma("unroll") for (int mt = 0; mt < SW_MT; ++mt)                                                   \
                    _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                    \
                        const int f_ = 16 * mt + 4 * qsub + r;                                                         \
                        if (f_ < TE && tau0 + f_ < Tp && u0_ + csub < A)                                               \
                            p.dkey[((long)b * Tp + tau0 + f_) * A + u0_ + csub] = S.dk[nu][mt][r];                     \
                    }                                                                                                  \
            }                                                                                                          \
        }                                                                                                              \
    }

// conv value of (frame f, kernel k) into the two LDS images of the step's conv tile
template <int KNMAX>
__device__ __forceinline__ void put_cv(unsigned short* s_cvx, unsigned short* s_cvT, int f, int k, float v) {
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    const unsigned short h = __builtin_bit_cast(unsigned short, hi), l = __builtin_bit_cast(unsigned short, lo);
    unsigned short* r = s_cvx + f * CVX_LD;
    r[k] = h; r[KNMAX + k] = l; r[2 * KNMAX + k] = h;
    s_cvT[k * CVT_LD + f] = h;
}

__global__ void cast_rows_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int rows, int cols, int ldd) {
    const long total = (long)rows * ldd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / ldd), c = (int)(i - (long)r * ldd);
        dst[i] = (c < cols) ? f2bf_bits(src[(long)r * cols + c]) : (unsigned short)0;
    }
}

}  // namespace
