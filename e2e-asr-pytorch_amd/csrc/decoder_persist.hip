// Persistent, state-resident form of the teacher-forced decoder loop (forward): ONE launch walks all L steps
// (reference src/asr.py:123-175; the per-step kernels of decoder.hip stay for greedy/beam decoding, fp32 parity mode,
// multi-layer decoders and shapes outside the limits below, and produce the same saved state).
//
// The loop has no cross-utterance dependency except shared weights, so each utterance b gets a CLUSTER of NT workgroups
// (one per TE-frame tile of its encoder output) that runs its own L steps with intra-cluster hand-offs only:
//   * resident for all L steps: the tile's key and enc rows as bf16 in LDS (no HBM re-read per step), W_proj^T, the previous
//     attention row of the utterance, the workgroup's slice of the decoder state (c of UPW hidden units);
//   * per step three all-gathers inside the cluster (data-tagged granules, lstm_persist2.hip's scheme):
//       H: h_{t-1} slices (UPW units per workgroup)      -> query slice q[a], a in the workgroup's QPW outputs
//       Q: query slices                                   -> energies of the tile, local softmax statistics (m, s) and the
//                                                            partial context  sum_f exp(e_f - m) enc[f,:]
//       S: {e[TE], m, s, ctx_partial[E]} records          -> every workgroup rebuilds the utterance's attention row and context
//     then the LSTM cell for the workgroup's UPW units (4*UPW rows of [W_ih(ctx part) | W_hh] streamed as bf16 from L2;
//     the embedding part W_ih[:, :Dd] emb(token) is one batched contraction before the launch);
//   * cluster b lives on XCD b % 8 (workgroup id = 8*slot + xcd under round-robin dispatch); placement is verified by the
//     XCC-id consensus and only then the producers use L2-local stores.
// Waves 0-7 compute and store, waves 8-11 only poll (their vector-memory queue holds nothing else).  The compute waves
// synchronise among themselves through an LDS counter where the polling waves are busy; s_barrier is used where
// the polled data is handed over.
#include "common.h"
#include "handoff.h"
#include <stdlib.h>
#include <algorithm>

namespace {

// Tag = 6 bits in the three mantissa LSBs of both floats of a granule: 2-bit step sequence + 4-bit launch epoch.  (With a
// 2-bit tag any stale or foreign 8 bytes pass the check with probability 1/4; recycled allocator memory whose old lines
// still sit in this XCD's L2 did exactly that in the first step of a launch.)  Payload loses 3 of 24 mantissa bits.
constexpr u64 PAIR_MASK = 7ull | (7ull << 32);
__device__ __forceinline__ u64 pair_want(unsigned seq, unsigned epoch) {
    const unsigned tag = ((epoch & 15u) << 2) | seq;
    return (u64)(tag & 7u) | ((u64)(tag >> 3) << 32);
}
__device__ __forceinline__ u64 pack2(float a, float b, u64 want) {
    return ((u64)(__float_as_uint(a) & ~7u) | ((u64)(__float_as_uint(b) & ~7u) << 32)) | want;
}
__device__ __forceinline__ float lo_f(u64 g) { return __uint_as_float((unsigned)g & ~7u); }
__device__ __forceinline__ float hi_f(u64 g) { return __uint_as_float((unsigned)(g >> 32) & ~7u); }
constexpr float NEG_BIG = -1e30f;       // masked energy in the exchange records (-inf would turn into NaN under the tag bit)
constexpr int NCW = 8, NPW = 4;         // compute waves, polling waves
#ifdef ASR_DIAG
#define DP_DECL unsigned long long dg_t = __builtin_amdgcn_s_memrealtime(), dg_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DP_MARK(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); dg_acc[k] += n_ - dg_t; dg_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define DP_DUMP { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long* o = (unsigned long long*)p.status + 128; for (int k = 0; k < 16; ++k) o[k] = dg_acc[k]; } }
#else
#define DP_DECL
#define DP_MARK(k)
#define DP_DUMP
#endif
constexpr int RB = 5;                   // gate rows per batch of the cell contraction
inline size_t align_up256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ float bf2f_(unsigned short x) { return __uint_as_float((unsigned)x << 16); }
__device__ __forceinline__ float tanh_f(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

struct PD {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const float* enc;
    const int64_t* enc_len;
    const unsigned short* wcat16;   // (4Dd, KCP) bf16 rows [W_ih[:, Dd:Dd+E] | W_hh | 0-pad]
    const float* embproj;           // (B*L, 4Dd)  W_ih[:, :Dd] . emb(token)
    u64* xbuf;
    unsigned* status;
    int NT, TE, UPW, QPW, CPW;      // tiles per utterance, frames per tile, hidden units / query outputs / context columns per workgroup
    int HG2, QG2, SG2;              // granules per producer record (even)
    int KC, KCP;                    // E + Dd, padded to a multiple of 8
    int allow_local;
    unsigned epoch;                 // launch counter (tag bits)
};

// barrier among the NCW compute waves only (the polling waves are inside a spin loop at these points)
__device__ __forceinline__ void compute_barrier(unsigned* cnt, unsigned& gen) {
    gen += NCW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int KNMAX, int TPW>
__global__ __launch_bounds__(64 * (NCW + NPW)) void dec_fwd_persist(PD p) {
    constexpr int TE = 8 * TPW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_bar;
    __shared__ float s_scale[NCW][32];
    const asr_dec_dims_t& d = p.d;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int cb = slot / p.NT, j = slot - cb * p.NT;
    const int b = cb * 8 + xcd;
    if (b >= d.B) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NT = p.NT, A = d.A, E = d.E, Dd = d.Dd, Tp = d.Tp, Kn = d.Kn, Ks = d.Ks, L = d.L;
    const int taps = 2 * Ks + 1, XW = Dd + E;
    const int tau0 = j * TE;
    const int len = min((int)p.enc_len[b], Tp);
    const int tmax = max(len - 1, 0);
    // ---- LDS carve
    unsigned short* s_key = reinterpret_cast<unsigned short*>(smem);                 // [TE][A] bf16
    unsigned short* s_enc = s_key + ((TE * A + 7) & ~7);                              // [TE][E] bf16
    float* s_wpT = reinterpret_cast<float*>(s_enc + ((TE * E + 7) & ~7));             // [Kn][A]
    float* s_x2 = s_wpT + ((Kn * A + 3) & ~3);                                        // [2][KCP]  ctx_t | h_{t-1} | 0, by step parity
    float* s_q = s_x2 + 2 * p.KCP;                                                       // [A]
    float* s_wg = s_q + ((A + 3) & ~3);                                               // [A]
    float* s_attp = s_wg + ((A + 3) & ~3);                                             // [Ks + Tp + Ks + 4]  zero-padded previous attention
    float* s_wc = s_attp + ((Tp + 2 * Ks + 4 + 3) & ~3);                              // [Kn*taps]
    float* s_conv = s_wc + ((Kn * taps + 3) & ~3);                                    // [Kn][TE]
    float* s_e = s_conv + Kn * TE;                                                    // [TE]
    float* s_g = s_e + ((TE + 3) & ~3);                                               // [4*UPW]
    float* s_stage = s_g + ((4 * p.UPW + 3) & ~3);                                    // [NT][2*SG2]
    if (tid == 0) s_bar = 0u;
    const long region = (long)NT * (p.HG2 + p.QG2 + p.SG2);
    auto xb = [&](int parity) { return p.xbuf + ((long)parity * d.B + b) * region; };
    // The host memset reaches memory, but granules that an earlier launch stored L2-locally (sc0) can still sit in this
    // XCD's L2 with perfectly valid tags (observed: wrong data in the first step after a launch with other inputs; a
    // write-through `sc1` store of zeros did NOT displace the resident copy).  Every producer therefore clears its own
    // records with the same L2-local store flavour before it joins the consensus - the barrier behind which polling starts.
    for (int parity = 0; parity < 2; ++parity) {
        u64* base = xb(parity);
        for (int i = tid; i < p.HG2; i += blockDim.x) st_gran_local(base + (long)j * p.HG2 + i, 0ull);
        for (int i = tid; i < p.QG2; i += blockDim.x) st_gran_local(base + (long)NT * p.HG2 + (long)j * p.QG2 + i, 0ull);
        for (int i = tid; i < p.SG2; i += blockDim.x) st_gran_local(base + (long)NT * (p.HG2 + p.QG2) + (long)j * p.SG2 + i, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.status) + 64 + b, NT, p.allow_local, p.status);

    // ---- resident data
    for (int i = tid; i < TE * A; i += blockDim.x) {
        const int f = i / A, a = i - f * A;
        s_key[i] = f2bf_bits(p.s.key[((long)b * Tp + min(tau0 + f, tmax)) * A + a]);
    }
    for (int i = tid; i < TE * E; i += blockDim.x) {
        const int f = i / E, c = i - f * E;
        s_enc[i] = f2bf_bits(p.enc[((long)b * Tp + min(tau0 + f, tmax)) * E + c]);
    }
    for (int i = tid; i < Kn * A; i += blockDim.x) { const int a = i / Kn, k = i - a * Kn; s_wpT[k * A + a] = p.w.Wproj[i]; }
    for (int i = tid; i < Kn * taps; i += blockDim.x) s_wc[i] = p.w.Wconv[i];
    for (int i = tid; i < A; i += blockDim.x) s_wg[i] = p.w.wg[i];
    for (int i = tid; i < 2 * p.KCP; i += blockDim.x) s_x2[i] = 0.f;
    {
        const float uni = 1.f / (float)max(len, 1);
        for (int i = tid; i < Tp + 2 * Ks + 4; i += blockDim.x) {
            const int tau = i - Ks;
            s_attp[i] = (tau >= 0 && tau < len) ? uni : 0.f;       // initial attention: uniform over the valid frames
        }
    }
    __syncthreads();

    if (wave >= NCW) {
        // =========================== polling role ===========================
        const int gt = tid - 64 * NCW, np = 64 * NPW;
        for (int t = 0; t < L; ++t) {
            // H: h_{t-1} of the utterance -> s_x[E ..]
            if (t > 0) {
                const u64* src = xb((t - 1) & 1);
                const u64 want = pair_want(seq_of(t - 1), p.epoch);
                for (int i0 = gt; 2 * i0 < NT * p.HG2; i0 += np) {
                    u64 lo[1], hi[1];
                    gather16<1>(src + 2 * i0, 0, 1, PAIR_MASK, want, lo, hi, p.status);
                    const int g0 = 2 * i0, prod = g0 / p.HG2, gi = g0 - prod * p.HG2;
                    const int u = prod * p.UPW + 2 * gi;          // units 2gi, 2gi+1 (lo), 2gi+2, 2gi+3 (hi) of producer prod
                    const float v[4] = {lo_f(lo[0]), hi_f(lo[0]), lo_f(hi[0]), hi_f(hi[0])};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (2 * gi + k < p.UPW && u + k < Dd) s_x2[(t & 1) * p.KCP + E + u + k] = v[k];
                }
            }
            __syncthreads();                                            // B1
            // Q: query of step t -> s_q
            {
                const u64* src = xb(t & 1) + (long)NT * p.HG2;
                const u64 want = pair_want(seq_of(t), p.epoch);
                for (int i0 = gt; 2 * i0 < NT * p.QG2; i0 += np) {
                    u64 lo[1], hi[1];
                    gather16<1>(src + 2 * i0, 0, 1, PAIR_MASK, want, lo, hi, p.status);
                    const int g0 = 2 * i0, prod = g0 / p.QG2, gi = g0 - prod * p.QG2;
                    const int a = prod * p.QPW + 2 * gi;
                    const float v[4] = {lo_f(lo[0]), hi_f(lo[0]), lo_f(hi[0]), hi_f(hi[0])};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (2 * gi + k < p.QPW && a + k < A) s_q[a + k] = v[k];
                }
            }
            __syncthreads();                                            // B2
            // S: softmax records of all tiles -> s_stage (flat copy)
            {
                const u64* src = xb(t & 1) + (long)NT * (p.HG2 + p.QG2);
                const u64 want = pair_want(seq_of(t), p.epoch);
                const int npair = NT * p.SG2 / 2;
                for (int i0 = gt; i0 < npair; i0 += 10 * np) {
                    u64 lo[10], hi[10];
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) if (i0 + k * np < npair) cnt = k + 1;
                    gather16<10>(src + 2 * i0, 2 * np, cnt, PAIR_MASK, want, lo, hi, p.status);
#pragma unroll
                    for (int k = 0; k < 10; ++k)
                        if (k < cnt) {
                            float4* o = reinterpret_cast<float4*>(s_stage + 4 * (long)(i0 + k * np));
                            *o = make_float4(lo_f(lo[k]), hi_f(lo[k]), lo_f(hi[k]), hi_f(hi[k]));
                        }
                }
            }
            __syncthreads();                                            // B3
        }
        return;
    }

    // =========================== compute role ===========================
    unsigned gen = 0;
    const int u_base = j * p.UPW, q_base = j * p.QPW, c_base = j * p.CPW;
    const int SG2f = 2 * p.SG2;                                         // floats per staged record
    const float bg = p.w.bg[0];
    float c_state = 0.f;                                                // cell state of unit u_base + lane (wave 0, lane < UPW)
    const int RPW = (4 * p.UPW + NCW - 1) / NCW;                        // gate rows per wave
    DP_DECL

    for (int t = 0; t < L; ++t) {
        const long row = (long)b * L + t;
        float* s_x = s_x2 + (t & 1) * p.KCP;
        const u64 want = pair_want(seq_of(t), p.epoch);
        u64* out = xb(t & 1);
        // operands of the cell phase that do not depend on this step's hand-offs: requested now
        float add_r = 0.f;                                              // lane r of the wave: embproj + both biases of its gate row
        {
            const int r = wave * RPW + lane;
            if (lane < RPW && r < 4 * p.UPW) {
                const int g = r / p.UPW, ul = r - g * p.UPW, unit = u_base + ul;
                if (unit < Dd) {
                    const int grow = g * Dd + unit;
                    add_r = p.embproj[row * 4 * Dd + grow] + p.w.bih[0][grow] + p.w.bhh[0][grow];
                }
            }
        }
        DP_MARK(0)
        __syncthreads();                                                // B1: s_x holds h_{t-1}
        DP_MARK(1)
        // ---- query slice: outputs q_base + o, two per wave per round, lanes over the reduction
        for (int o0 = 2 * wave; o0 < p.QPW; o0 += 2 * NCW) {
            float acc0 = 0.f, acc1 = 0.f;
            const int a0 = min(q_base + o0, A - 1), a1 = min(q_base + o0 + 1, A - 1);
            if (t > 0) {
                float w0[5], w1[5];
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int kk = min(lane + 64 * k, Dd - 1);
                    w0[k] = p.w.Wq[(long)a0 * Dd + kk];
                    w1[k] = p.w.Wq[(long)a1 * Dd + kk];
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float hv = (lane + 64 * k < Dd) ? s_x[E + lane + 64 * k] : 0.f;
                    acc0 += w0[k] * hv; acc1 += w1[k] * hv;
                }
                for (int kk = lane + 320; kk < Dd; kk += 64) { const float hv = s_x[E + kk]; acc0 += p.w.Wq[(long)a0 * Dd + kk] * hv; acc1 += p.w.Wq[(long)a1 * Dd + kk] * hv; }
                acc0 = wave_sum(acc0); acc1 = wave_sum(acc1);
            }
            if (lane == 0) {
                const float q0 = tanhf(acc0 + p.w.bq[a0]), q1 = tanhf(acc1 + p.w.bq[a1]);
                if (q_base + o0 < A) p.s.q[row * A + q_base + o0] = q0;
                if (o0 + 1 < p.QPW && q_base + o0 + 1 < A) p.s.q[row * A + q_base + o0 + 1] = q1;
                u64* dst = out + (long)NT * p.HG2 + (long)j * p.QG2 + (o0 >> 1);
                if (local) publish<true>(dst, pack2(q0, q1, want)); else publish<false>(dst, pack2(q0, q1, want));
            }
        }
        // pad granule of an odd record length (the gather reads 16-byte pairs)
        if (tid == 0 && (p.QPW + 1) / 2 < p.QG2) {
            u64* dst = out + (long)NT * p.HG2 + (long)j * p.QG2 + p.QG2 - 1;
            if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
        }
        DP_MARK(2)
        // ---- location convolution of the tile from the previous attention row (runs while the query is gathered):
        //      item = (tap range, kernel, group of 4 frames) with a sliding register window, partial sums meet in LDS
        {
            constexpr int ngrp = TE / 4;
            const int nout = Kn * ngrp;
            const int parts = max(1, min(8, (64 * NCW) / nout));
            const int tp = (taps + parts - 1) / parts;
            float* s_part = s_stage;                                     // free until the S gather of this step
            for (int it = tid; it < parts * nout; it += 64 * NCW) {
                const int pz = it / nout, o = it - pz * nout, k = o / ngrp, ig = o - k * ngrp;
                const int j0 = pz * tp, j1 = min(taps, j0 + tp);
                const float* wk = s_wc + k * taps;
                const float* pa = s_attp + tau0 + 4 * ig;                // pa[i + jj] = prev_att[tau0 + 4ig + i + jj - Ks]
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                float p0 = pa[j0], p1 = pa[j0 + 1], p2 = pa[j0 + 2];
                for (int jj = j0; jj < j1; ++jj) {
                    const float wv = wk[jj], p3 = pa[jj + 3];
                    a0 += wv * p0; a1 += wv * p1; a2 += wv * p2; a3 += wv * p3;
                    p0 = p1; p1 = p2; p2 = p3;
                }
                float* o4 = s_part + (long)pz * Kn * TE + k * TE + 4 * ig;
                o4[0] = a0; o4[1] = a1; o4[2] = a2; o4[3] = a3;
            }
            compute_barrier(&s_bar, gen);
            for (int o = tid; o < Kn * TE; o += 64 * NCW) {
                float v = 0.f;
                for (int pz = 0; pz < parts; ++pz) v += s_part[(long)pz * Kn * TE + o];
                s_conv[o] = v;
                const int k = o / TE, i = o - k * TE;
                if (p.s.conv && tau0 + i < Tp) p.s.conv[(row * Kn + k) * Tp + tau0 + i] = v;
            }
        }
        DP_MARK(3)
        __syncthreads();                                                // B2: s_q holds q_t, s_conv the tile's conv
        DP_MARK(4)
        // ---- energies of the tile: wave w owns frames w*TPW.., lanes sweep the attention dimension
        {
            float cv[KNMAX][TPW];
#pragma unroll
            for (int k = 0; k < KNMAX; ++k)
#pragma unroll
                for (int i = 0; i < TPW; ++i) cv[k][i] = (k < Kn) ? s_conv[k * TE + wave * TPW + i] : 0.f;
            float e[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) e[i] = 0.f;
#pragma unroll 1
            for (int a = lane; a < A; a += 64) {                     // not unrolled: register budget of a 12-wave workgroup
                float wp[KNMAX];
#pragma unroll
                for (int k = 0; k < KNMAX; ++k) wp[k] = (k < Kn) ? s_wpT[k * A + a] : 0.f;
                const float qa = s_q[a], wg_ = s_wg[a];
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    float lp = 0.f;
#pragma unroll
                    for (int k = 0; k < KNMAX; ++k) lp += wp[k] * cv[k][i];
                    const float kvv = bf2f_(s_key[(wave * TPW + i) * A + a]);
                    e[i] += wg_ * tanh_f(kvv + qa + tanh_f(lp));
                }
            }
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const float sv = wave_sum(e[i]);
                const int f = wave * TPW + i;
                if (lane == 0) s_e[f] = (tau0 + f < len) ? (sv + bg) / d.temperature : NEG_BIG;
            }
        }
        DP_MARK(5)
        compute_barrier(&s_bar, gen);                                   // c3: s_e complete
        DP_MARK(6)
        // ---- local softmax statistics (every wave for itself) and the tile's partial context
        {
            const float ev = (lane < TE) ? s_e[lane] : NEG_BIG;
            const float m = wave_max(ev);
            const float wv = (ev > 0.5f * NEG_BIG) ? __expf(ev - m) : 0.f;
            const float ssum = wave_sum(wv);
            // thread c2: context columns 2*c2, 2*c2+1
            const int c2 = tid;
            float x0 = 0.f, x1 = 0.f;
            if (2 * c2 < E) {
#pragma unroll
                for (int f = 0; f < TE; ++f) {
                    const float wf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv), f));
                    const unsigned pr = *reinterpret_cast<const unsigned*>(s_enc + f * E + 2 * c2);
                    x0 += wf * __uint_as_float(pr << 16);
                    x1 += wf * __uint_as_float(pr & 0xffff0000u);
                }
            }
            u64* rec = out + (long)NT * (p.HG2 + p.QG2) + (long)j * p.SG2;
            const int ghead = (TE + 2) >> 1;                             // granules of e[TE], m, s
            if (2 * c2 < E) {
                if (local) publish<true>(rec + ghead + c2, pack2(x0, x1, want)); else publish<false>(rec + ghead + c2, pack2(x0, x1, want));
            }
            // e pairs, (m, s) and the pad granule by the last compute wave's lanes
            if (wave == NCW - 1) {
                if (lane < TE / 2) {
                    const float e0 = s_e[2 * lane], e1 = s_e[2 * lane + 1];
                    if (local) publish<true>(rec + lane, pack2(e0, e1, want)); else publish<false>(rec + lane, pack2(e0, e1, want));
                } else if (lane == TE / 2) {
                    if (local) publish<true>(rec + lane, pack2(m, ssum, want)); else publish<false>(rec + lane, pack2(m, ssum, want));
                } else if (lane == TE / 2 + 1 && ghead + E / 2 < p.SG2) {
                    u64* dst = rec + p.SG2 - 1;
                    if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
                }
            }
        }
        DP_MARK(7)
        __syncthreads();                                                // B3: s_stage holds every tile's record
        DP_MARK(8)
        // ---- attention row and context of the utterance
        {
            float mi = NEG_BIG, si = 0.f;
            if (lane < NT) { mi = s_stage[lane * SG2f + TE]; si = s_stage[lane * SG2f + TE + 1]; }
            const float M = wave_max(mi);
            const float wi = (lane < NT) ? si * __expf(mi - M) : 0.f;
            const float S = fmaxf(wave_sum(wi), 1e-30f);
            const float invS = 1.f / S;
            if (lane < 32) s_scale[wave][lane] = (lane < NT) ? __expf(mi - M) * invS : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int tau = tid; tau < Tp; tau += 64 * NCW) {
                const int i = tau / TE, f = tau - i * TE;
                const float ev = s_stage[i * SG2f + f];
                const float av = (ev > 0.5f * NEG_BIG) ? __expf(ev - M) * invS : 0.f;
                s_attp[Ks + tau] = av;
                if (i == j) p.s.att[row * Tp + tau] = av;
            }
            for (int c = tid; c < E; c += 64 * NCW) {
                float acc = 0.f;
                for (int i = 0; i < NT; ++i) acc += s_stage[i * SG2f + TE + 2 + c] * s_scale[wave][i];
                s_x[c] = acc;
                if (c >= c_base && c < c_base + p.CPW) p.s.xin[row * XW + Dd + c] = acc;
            }
        }
        DP_MARK(9)
        compute_barrier(&s_bar, gen);                                   // c5: s_x holds [ctx_t | h_{t-1}]
        DP_MARK(10)
        // ---- LSTM cell: gate rows r = g*UPW + ul of this workgroup, RPW rows per wave, lanes over 16-byte chunks
        {
            const int nchunk = p.KCP >> 3;
            float mine = 0.f;
#pragma unroll 1
            for (int bt = 0; bt * RB < RPW; ++bt) {
                float part[RB];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) part[rr] = 0.f;
                for (int ch0 = lane; ch0 < nchunk; ch0 += 128) {
                    uint4 wv[RB][2];
#pragma unroll
                    for (int rr = 0; rr < RB; ++rr) {
                        const int r = min(wave * RPW + bt * RB + rr, 4 * p.UPW - 1);
                        const int g = r / p.UPW, ul = r - g * p.UPW;
                        const long grow = (long)g * Dd + min(u_base + ul, Dd - 1);
#pragma unroll
                        for (int h2 = 0; h2 < 2; ++h2)
                            wv[rr][h2] = *reinterpret_cast<const uint4*>(p.wcat16 + grow * p.KCP + 8 * min(ch0 + 64 * h2, nchunk - 1));
                    }
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int ch = ch0 + 64 * h2;
                        if (ch < nchunk) {
                            const float4 xa = *reinterpret_cast<const float4*>(s_x + 8 * ch);
                            const float4 xb4 = *reinterpret_cast<const float4*>(s_x + 8 * ch + 4);
#pragma unroll
                            for (int rr = 0; rr < RB; ++rr) {
                                const uint4 w4 = wv[rr][h2];
                                part[rr] += __uint_as_float(w4.x << 16) * xa.x + __uint_as_float(w4.x & 0xffff0000u) * xa.y +
                                            __uint_as_float(w4.y << 16) * xa.z + __uint_as_float(w4.y & 0xffff0000u) * xa.w +
                                            __uint_as_float(w4.z << 16) * xb4.x + __uint_as_float(w4.z & 0xffff0000u) * xb4.y +
                                            __uint_as_float(w4.w << 16) * xb4.z + __uint_as_float(w4.w & 0xffff0000u) * xb4.w;
                            }
                        }
                    }
                }
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const float sv = wave_sum(part[rr]);
                    if (lane == bt * RB + rr) mine = sv;
                }
            }
            const int r = wave * RPW + lane;
            if (lane < RPW && r < 4 * p.UPW) s_g[r] = mine + add_r;
        }
        DP_MARK(11)
        compute_barrier(&s_bar, gen);                                   // c6: s_g holds the gate pre-activations
        DP_MARK(12)
        if (wave == 0) {
            const int unit = u_base + lane;
            const bool uok = lane < p.UPW && unit < Dd;
            float hv = 0.f;
            if (uok) {
                const float ai = sigmoidf_(s_g[lane]), af = sigmoidf_(s_g[p.UPW + lane]);
                const float ag = tanhf(s_g[2 * p.UPW + lane]), ao = sigmoidf_(s_g[3 * p.UPW + lane]);
                c_state = af * c_state + ai * ag;
                hv = ao * tanhf(c_state);
                float* go = p.s.gates + row * 4 * Dd;
                go[unit] = ai; go[Dd + unit] = af; go[2 * Dd + unit] = ag; go[3 * Dd + unit] = ao;
                p.s.cs[row * Dd + unit] = c_state;
                p.s.hs[row * Dd + unit] = hv;
            }
            const float hn = __shfl_down(hv, 1);
            if (t + 1 < L && (lane & 1) == 0 && lane < 2 * p.HG2) {
                u64* dst = out + (long)j * p.HG2 + (lane >> 1);
                if (local) publish<true>(dst, pack2(hv, hn, want)); else publish<false>(dst, pack2(hv, hn, want));
            }
        }
        DP_MARK(13)
    }
    DP_DUMP
}

__global__ void build_wcat16_kernel(const float* __restrict__ wih, const float* __restrict__ whh, unsigned short* __restrict__ out,
                                    int rows, int Dd, int E, int KCP) {
    const long total = (long)rows * KCP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / KCP), k = (int)(i - (long)r * KCP);
        float v = 0.f;
        if (k < E) v = wih[(long)r * (Dd + E) + Dd + k];
        else if (k < E + Dd) v = whh[(long)r * Dd + (k - E)];
        out[i] = f2bf_bits(v);
    }
}

struct PersistPlan { bool ok; int tpw, NT, TE, UPW, QPW, CPW, HG2, QG2, SG2, KC, KCP; size_t lds, status_bytes, xbuf_bytes, wcat_bytes, emb_bytes, total; };

PersistPlan persist_plan(const asr_dec_dims_t& d) {
    PersistPlan pl{};
    pl.ok = false;
    if (d.NL != 1 || d.B > 64 || d.A > 1024 || d.Kn > 10 || (d.E & 7) != 0 || d.Dd > 20 * 32 || d.Tp < 1) return pl;
    const int cpx = cdiv(d.B, 8);                       // clusters per XCD
    const int cand[] = {2, 4, 5, 8};
    int tpw = 0;
    for (int i = 0; i < 4; ++i) {
        const int nt = cdiv(d.Tp, 8 * cand[i]);
        if (nt <= 30 && cpx * nt <= 32) { tpw = cand[i]; break; }
    }
    if (!tpw) return pl;
    pl.tpw = tpw; pl.TE = 8 * tpw; pl.NT = cdiv(d.Tp, pl.TE);
    pl.UPW = cdiv(d.Dd, pl.NT); pl.QPW = cdiv(d.A, pl.NT); pl.CPW = cdiv(d.E, pl.NT);
    if (pl.UPW > 64 || cdiv(4 * pl.UPW, NCW) > 60) return pl;
    auto even = [](int x) { return (x + 1) & ~1; };
    pl.HG2 = even((pl.UPW + 1) / 2); pl.QG2 = even((pl.QPW + 1) / 2); pl.SG2 = even((pl.TE + 2 + d.E) / 2);
    pl.KC = d.E + d.Dd; pl.KCP = (pl.KC + 7) & ~7;
    const int taps = 2 * d.Ks + 1;
    size_t fl = 0;
    fl += ((d.Kn * d.A + 3) & ~3) + 2 * pl.KCP + 2 * ((d.A + 3) & ~3) + ((d.Tp + 2 * d.Ks + 4 + 3) & ~3) + ((d.Kn * taps + 3) & ~3) +
          (size_t)d.Kn * pl.TE + ((pl.TE + 3) & ~3) + ((4 * pl.UPW + 3) & ~3) +
          std::max((size_t)pl.NT * 2 * pl.SG2, (size_t)8 * d.Kn * pl.TE);      // the record stage doubles as the conv's partial-sum area
    pl.lds = 2 * (size_t)(((pl.TE * d.A + 7) & ~7) + ((pl.TE * d.E + 7) & ~7)) + 4 * fl;
    if (pl.lds > 156 * 1024) return pl;
    pl.status_bytes = 4096;
    pl.xbuf_bytes = align_up256(2 * (size_t)d.B * pl.NT * (pl.HG2 + pl.QG2 + pl.SG2) * sizeof(u64));
    pl.wcat_bytes = align_up256((size_t)4 * d.Dd * pl.KCP * 2);
    pl.emb_bytes = align_up256((size_t)d.B * d.L * 4 * d.Dd * sizeof(float));
    pl.total = pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes + pl.emb_bytes;
    pl.ok = true;
    return pl;
}

}  // namespace

size_t dec_fwd_persist_work_bytes(const asr_dec_dims_t& d) {
    const PersistPlan pl = persist_plan(d);
    return pl.ok ? pl.total : 0;
}

// Returns ASR_OK when the whole loop was launched, 1 when the configuration has no persistent plan, negative on error.
int dec_fwd_persistent(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const float* enc,
                       const int64_t* enc_len, void* work, size_t work_bytes, hipStream_t st) {
    static int enabled = -1;
    if (enabled < 0) { const char* e = getenv("ASR_DEC_PERSIST"); enabled = (e && e[0] == '0') ? 0 : 1; }
    const PersistPlan pl = persist_plan(d);
    if (!enabled || !pl.ok || !work || work_bytes < pl.total || ((uintptr_t)work & 255) != 0 || !s.conv) return 1;
    char* base = (char*)work;
    unsigned* status = (unsigned*)base;
    u64* xbuf = (u64*)(base + pl.status_bytes);
    unsigned short* wcat16 = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes);
    float* embproj = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes);
    hipMemsetAsync(work, 0, pl.status_bytes + pl.xbuf_bytes, st);
    hipLaunchKernelGGL(build_wcat16_kernel, dim3(512), dim3(256), 0, st, w.Wih[0], w.Whh[0], wcat16, 4 * d.Dd, d.Dd, d.E, pl.KCP);
    const int XW = d.Dd + d.E;
    int rc = asr_gemm(s.xin, w.Wih[0], embproj, nullptr, d.B * d.L, 4 * d.Dd, d.Dd, XW, XW, 4 * d.Dd, 1, 1, ASR_ACT_NONE, 0, 1, 1, 0, 0, 0,
                      0, 0, ASR_BF16, (asr_stream_t)st);
    if (rc != ASR_OK) return rc;
    static unsigned epoch_counter = 1;
    static int allow = -1;
    if (allow < 0) { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); allow = (e && e[0] == '0') ? 0 : 1; }
    PD p{d, w, s, enc, enc_len, wcat16, embproj, xbuf, status, pl.NT, pl.TE, pl.UPW, pl.QPW, pl.CPW, pl.HG2, pl.QG2, pl.SG2, pl.KC, pl.KCP, allow, epoch_counter++};
    const int cpx = cdiv(d.B, 8);
    const dim3 grid(8 * cpx * pl.NT), block(64 * (NCW + NPW));
#define DPF_LAUNCH(KN_, TPW_)                                                                                                   \
    {                                                                                                                           \
        static bool attr = false;                                                                                               \
        if (!attr) { hipFuncSetAttribute((const void*)dec_fwd_persist<KN_, TPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); attr = true; } \
        hipLaunchKernelGGL((dec_fwd_persist<KN_, TPW_>), grid, block, pl.lds, st, p);                                             \
    }
    if (d.Kn <= 4) {
        if (pl.tpw == 2) DPF_LAUNCH(4, 2) else if (pl.tpw == 4) DPF_LAUNCH(4, 4) else if (pl.tpw == 5) DPF_LAUNCH(4, 5) else DPF_LAUNCH(4, 8)
    } else {
        if (pl.tpw == 2) DPF_LAUNCH(10, 2) else if (pl.tpw == 4) DPF_LAUNCH(10, 4) else if (pl.tpw == 5) DPF_LAUNCH(10, 5) else return 1;
    }
#undef DPF_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_att_decoder_fwd(persistent): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}
