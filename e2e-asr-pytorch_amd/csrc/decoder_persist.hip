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

static int g_persist_fwd = -1, g_persist_bwd = -1;   // -1: from env ASR_DEC_PERSIST / ASR_DEC_PERSIST_BWD (default on)
static int g_stream_fwd = -1, g_stream_bwd = -1;     // 1: take the streamed-tile plan (decoder_stream.hip) even where the LDS-resident one exists; -1: env ASR_DEC_STREAM
// bit 0: persistent forward, bit 1: persistent backward, bit 2 / 3: prefer the streamed-tile plan forward / backward (tests, A/B
// runs; shapes without an LDS-resident plan take the streamed one anyway); returns the previous setting
extern "C" int asr_att_decoder_set_persistent(int flags) {
    const int old = (g_persist_fwd != 0 ? 1 : 0) | (g_persist_bwd != 0 ? 2 : 0) | (g_stream_fwd > 0 ? 4 : 0) | (g_stream_bwd > 0 ? 8 : 0);
    g_persist_fwd = flags & 1; g_persist_bwd = (flags >> 1) & 1; g_stream_fwd = (flags >> 2) & 1; g_stream_bwd = (flags >> 3) & 1;
    return old;
}
static void stream_env() {
    if (g_stream_fwd < 0) { const char* e = getenv("ASR_DEC_STREAM"); g_stream_fwd = (e && e[0] == '1') ? 1 : 0; }
    if (g_stream_bwd < 0) { const char* e = getenv("ASR_DEC_STREAM"); g_stream_bwd = (e && e[0] == '1') ? 1 : 0; }
}
// decoder_stream.hip
size_t dec_fwd_stream_work_bytes(const asr_dec_dims_t& d);
int dec_fwd_streamed(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const float* enc,
                     const int64_t* enc_len, void* work, size_t work_bytes, hipStream_t st);

#include "decoder_cluster.h"
#include "decoder_bwd_common.h"

namespace {

template <int KNMAX, int TPW>
__global__ __launch_bounds__(64 * (NCW + NPW)) void dec_fwd_persist(PD p) {
    constexpr int TE = 8 * TPW;
    constexpr int MT = (TE + 15) / 16;                                   // 16-frame MFMA tiles of the energy sweep
    constexpr int EPL = 16 * MT;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_bar;
    __shared__ float s_scale[NCW][32];
    const asr_dec_dims_t& d = p.d;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int cb = slot / p.NT, j = slot - cb * p.NT;
    const int b = cb * 8 + xcd;
    if (b >= d.B) return;
    const unsigned epoch_ = __builtin_amdgcn_readfirstlane(p.status[EPOCH_WORD]);     // launch epoch of this work area (decoder_cluster.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = p.NT, A = d.A, E = d.E, Dd = d.Dd, Tp = d.Tp, Kn = d.Kn, Ks = d.Ks, L = d.L;
    const int taps = 2 * Ks + 1, XW = Dd + E;
    const int tau0 = j * TE;
    const int len = min((int)p.enc_len[b], Tp);
    const int tmax = max(len - 1, 0);
    // ---- LDS carve
    unsigned short* s_key = reinterpret_cast<unsigned short*>(smem);                 // [TE/4][A][4] bf16 (four frames innermost)
    unsigned short* s_enc = s_key + ((TE * A + 7) & ~7);                              // [TE][E] bf16
    unsigned short* s_cvx = s_enc + ((TE * E + 7) & ~7);                              // [16*MT][32] bf16 conv tile of the step, slots {hi | lo | hi}
    const ConvGeo cg_ = conv_geo(EPL, Ks);
    unsigned short* s_ximg = s_cvx + EPL * FCVX_LD;                                   // window image of the attention row, {hi, lo} x 4 shifts
    unsigned short* s_wimg = s_ximg + 8 * cg_.IMG_LD;                                 // filter image {hi, lo} x 16 rows (decoder_cluster.h::conv_mfma)
    float* s_x2 = reinterpret_cast<float*>(s_wimg + 32 * cg_.WKP);                    // [2][KCP]  ctx_t | h_{t-1} | 0, by step parity
    float* s_q = s_x2 + 2 * p.KCP;                                                       // [A]
    float* s_wg = s_q + ((A + 3) & ~3);                                               // [A]
    const int WT = (taps + 3) & ~3;                                                    // zero-padded filter row (16-byte units)
    const int ATP = (NT * TE + 2 * Ks + 8 + 3) & ~3;
    float* s_attp = s_wg + ((A + 3) & ~3);                                             // [Ks + NT*TE + Ks + 8]  zero-padded previous attention
    float* s_epart = s_attp + ATP;                                                    // [NCW][16*MT] energy partials of the compute waves
    float* s_g = s_epart + NCW * EPL;                                                 // [4*UPW]
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
        s_key[((long)(f >> 2) * A + a) * 4 + (f & 3)] = f2bf_bits(p.s.key[((long)b * Tp + min(tau0 + f, tmax)) * A + a]);
    }
    for (int i = tid; i < TE * E; i += blockDim.x) {
        const int f = i / E, c = i - f * E;
        s_enc[i] = f2bf_bits(p.enc[((long)b * Tp + min(tau0 + f, tmax)) * E + c]);
    }
    for (int i = tid; i < EPL * FCVX_LD; i += blockDim.x) s_cvx[i] = 0;
    conv_build_wimg(p.w.Wconv, Kn, taps, cg_, s_wimg, tid, blockDim.x);
    for (int i = tid; i < A; i += blockDim.x) s_wg[i] = p.w.wg[i];
    for (int i = tid; i < 2 * p.KCP; i += blockDim.x) s_x2[i] = 0.f;
    {
        const float uni = 1.f / (float)max(len, 1);
        for (int i = tid; i < ATP; i += blockDim.x) {
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
                const u64 want = pair_want(seq_of(t - 1), epoch_);
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
            DP_JIT(0)
            __syncthreads();                                            // B1
            DP_JIT(1)
            // Q: query of step t -> s_q
            {
                const u64* src = xb(t & 1) + (long)NT * p.HG2;
                const u64 want = pair_want(seq_of(t), epoch_);
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
            DP_JIT(2)
            __syncthreads();                                            // B2
            DP_JIT(3)
            // S: softmax records of all tiles -> s_stage (flat copy)
            {
                const u64* src = xb(t & 1) + (long)NT * (p.HG2 + p.QG2);
                const u64 want = pair_want(seq_of(t), epoch_);
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
            DP_JIT(4)
            __syncthreads();                                            // B3
            DP_JIT(5)
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
    // sweep operands of this wave's column units (registers): W_proj(a, :) in the {hi, hi, lo} K slots, w_g(a)
    const int nu_cnt = (((A + 15) >> 4) - wave + NCW - 1) / NCW;
    bf16x8 wpx[FSW_NU];
    float wgu[FSW_NU];
    {
        const int q = lane >> 4, c = lane & 15;
#pragma unroll
        for (int nu = 0; nu < FSW_NU; ++nu) {
            const int a = 16 * (wave + NCW * nu) + c;
            const bool ok = a < A;
            const float* wr = p.w.Wproj + (long)min(a, A - 1) * Kn;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int slot = 8 * q + i;
                const int k = slot < KNMAX ? slot : (slot < 2 * KNMAX ? slot - KNMAX : slot - 2 * KNMAX);
                const float w = (ok && slot < 3 * KNMAX && k < Kn) ? wr[min(k, Kn - 1)] : 0.f;
                const __bf16 hi = (__bf16)w;
                wpx[nu][i] = (slot < 2 * KNMAX) ? hi : (__bf16)(w - (float)hi);
            }
            wgu[nu] = ok ? p.w.wg[min(a, A - 1)] : 0.f;
        }
    }
    constexpr int QR = 2;                                               // query rounds with register-resident W_q rows
    float wq0[QR][5], wq1[QR][5], bq0[QR], bq1[QR];
#pragma unroll
    for (int r2 = 0; r2 < QR; ++r2) {
        const int o0 = 2 * wave + 2 * NCW * r2;
        const int a0 = min(q_base + o0, A - 1), a1 = min(q_base + o0 + 1, A - 1);
        bq0[r2] = p.w.bq[a0]; bq1[r2] = p.w.bq[a1];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int kk = min(lane + 64 * k, Dd - 1);
            wq0[r2][k] = p.w.Wq[(long)a0 * Dd + kk];
            wq1[r2][k] = p.w.Wq[(long)a1 * Dd + kk];
        }
    }
    DP_DECL

    for (int t = 0; t < L; ++t) {
        // thread-index arithmetic of the step is loop invariant: hoisted it becomes dozens of live address registers (and
        // spills); an opaque copy of the indices ties it to the step
        int tz = tid, lz_ = lane;
        asm volatile("" : "+v"(tz), "+v"(lz_));
        const int lane = lz_;
        const long row = (long)b * L + t;
        float* s_x = s_x2 + (t & 1) * p.KCP;
        const u64 want = pair_want(seq_of(t), epoch_);
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
        // ---- query slice: outputs q_base + o, two per wave per round, lanes over the reduction; the W_q rows of the first QR
        //      rounds are register-resident for the whole launch (no L2 round trip on the step's critical path)
        for (int o0 = 2 * wave, rd = 0; o0 < p.QPW; o0 += 2 * NCW, ++rd) {
            float acc0 = 0.f, acc1 = 0.f;
            const int a0 = min(q_base + o0, A - 1), a1 = min(q_base + o0 + 1, A - 1);
            float b0 = 0.f, b1 = 0.f;                                    // bias: registers for the resident rounds (a load inside
            if (rd < QR) {                                               // the lane-0 branch below would be a round trip per round)
#pragma unroll
                for (int r2 = 0; r2 < QR; ++r2) if (r2 == rd) { b0 = bq0[r2]; b1 = bq1[r2]; }
            } else {
                b0 = p.w.bq[a0]; b1 = p.w.bq[a1];
            }
            if (t > 0) {
                float w0[5], w1[5];
                if (rd < QR) {
#pragma unroll
                    for (int r2 = 0; r2 < QR; ++r2)
                        if (r2 == rd) {
#pragma unroll
                            for (int k = 0; k < 5; ++k) { w0[k] = wq0[r2][k]; w1[k] = wq1[r2][k]; }
                        }
                } else {
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const int kk = min(lane + 64 * k, Dd - 1);
                        w0[k] = p.w.Wq[(long)a0 * Dd + kk];
                        w1[k] = p.w.Wq[(long)a1 * Dd + kk];
                    }
                }
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const float hv = (lane + 64 * k < Dd) ? s_x[E + lane + 64 * k] : 0.f;
                    acc0 += w0[k] * hv; acc1 += w1[k] * hv;
                }
                for (int kk = lane + 320; kk < Dd; kk += 64) { const float hv = s_x[E + kk]; acc0 += p.w.Wq[(long)a0 * Dd + kk] * hv; acc1 += p.w.Wq[(long)a1 * Dd + kk] * hv; }
                acc0 = wave_sum_dpp(acc0); acc1 = wave_sum_dpp(acc1);
            }
            if (lane == 0) {
                const float q0 = tanh_f(acc0 + b0), q1 = tanh_f(acc1 + b1);
                u64* dst = out + (long)NT * p.HG2 + (long)j * p.QG2 + (o0 >> 1);
                if (local) publish<true>(dst, pack2(q0, q1, want)); else publish<false>(dst, pack2(q0, q1, want));     // the hop first
                if (q_base + o0 < A) p.s.q[row * A + q_base + o0] = q0;
                if (o0 + 1 < p.QPW && q_base + o0 + 1 < A) p.s.q[row * A + q_base + o0 + 1] = q1;
            }
        }
        // pad granule of an odd record length (the gather reads 16-byte pairs)
        if (tz == 0 && (p.QPW + 1) / 2 < p.QG2) {
            u64* dst = out + (long)NT * p.HG2 + (long)j * p.QG2 + p.QG2 - 1;
            if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
        }
        DP_MARK(2)
        // ---- location convolution of the tile from the previous attention row (runs while the query is gathered): a Toeplitz
        //      product on the matrix cores (decoder_cluster.h::conv_mfma).  A lane ends up with four kernels of one frame: the
        //      {hi | lo | hi} slots of the sweep's operand row (the A operand of its MFMA) are written straight from the accumulator
        {
            conv_build_ximg(s_attp + tau0, ATP - tau0, cg_, s_ximg, tz, 64 * NCW);
            compute_barrier(&s_bar, gen);
            conv_mfma<NCW>(s_ximg, s_wimg, cg_, MT, wave, lane, [&](int mt, int n, int q4, const f32x4& acc) {
                const int i = 16 * mt + n;
                unsigned short* r = s_cvx + i * FCVX_LD;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int k0 = 4 * q4 + 2 * pp;
                    if (k0 < Kn && i < TE) {
                        const float v0 = acc[2 * pp], v1 = acc[2 * pp + 1];
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
                        const unsigned hh = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                        const unsigned ll = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                        *reinterpret_cast<unsigned*>(r + k0) = hh;
                        *reinterpret_cast<unsigned*>(r + KNMAX + k0) = ll;
                        *reinterpret_cast<unsigned*>(r + 2 * KNMAX + k0) = hh;
                        if (p.s.conv && tau0 + i < Tp) {
                            p.s.conv[(row * Kn + k0) * Tp + tau0 + i] = v0;
                            if (k0 + 1 < Kn) p.s.conv[(row * Kn + k0 + 1) * Tp + tau0 + i] = v1;
                        }
                    }
                }
            });
        }
        DP_MARK(3)
        __syncthreads();                                                // B2: s_q holds q_t, s_cvx the tile's conv
        DP_MARK(4)
        // ---- energies of the tile on the matrix cores (see the backward's sweep_step): wave w owns the 16-column units
        //      w, w + NCW, ...; lp = conv . W_proj as ONE split-bf16 MFMA per 16 x 16 tile, a lane holds column a = 16 u + (lane & 15)
        //      and frames 16 mt + 4 (lane >> 4) + r; its partial energies are summed over the 16 columns of the row with DPP adds
        //      and over the waves through s_epart (fixed order: deterministic)
        {
            int opaque = 0;
            asm volatile("" : "+v"(opaque));                            // keeps the LDS addresses below out of the loop-invariant registers
            const int q = (lane >> 4) + opaque, c = lane & 15;
            float ep[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ep[mt][r] = 0.f;
            float qv[FSW_NU];
#pragma unroll
            for (int nu = 0; nu < FSW_NU; ++nu) qv[nu] = s_q[min(16 * (wave + NCW * nu) + c, A - 1)];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int f0 = 16 * mt + 4 * q;
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(s_cvx + (16 * mt + c) * FCVX_LD + 8 * q);
                const int fg = min(f0, TE - 4) >> 2;
#pragma unroll
                for (int nu = 0; nu < FSW_NU; ++nu) {
                    if (nu < nu_cnt) {
                        const int a = min(16 * (wave + NCW * nu) + c, A - 1);
                        const f32x4 lp = mma16(av, wpx[nu], f32x4{0.f, 0.f, 0.f, 0.f});
                        const uint2 kb = *reinterpret_cast<const uint2*>(s_key + ((long)fg * A + a) * 4);
                        const float key[4] = {__uint_as_float(kb.x << 16), __uint_as_float(kb.x & 0xffff0000u),
                                              __uint_as_float(kb.y << 16), __uint_as_float(kb.y & 0xffff0000u)};
#pragma unroll
                        for (int r = 0; r < 4; ++r) ep[mt][r] += wgu[nu] * tanh_f(key[r] + qv[nu] + tanh_f(lp[r]));
                    }
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ep[mt][r];
#define DPF_STEP(CTRL) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
                    DPF_STEP(0xB1) DPF_STEP(0x4E) DPF_STEP(0x141) DPF_STEP(0x140)      // sum over the 16 lanes of the row
#undef DPF_STEP
                    if (c == 0) s_epart[wave * EPL + 16 * mt + 4 * q + r] = v;
                }
        }
        DP_MARK(5)
        compute_barrier(&s_bar, gen);                                   // c3: s_epart complete
        DP_MARK(6)
        // ---- local softmax statistics (every wave for itself) and the tile's partial context
        {
            float ev = NEG_BIG;
            if (lane < TE) {
                float sv = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < NCW; ++w8) sv += s_epart[w8 * EPL + lane];
                if (tau0 + lane < len) ev = (sv + bg) / d.temperature;
            }
            const float m = wave_max_dpp(ev);
            const float wv = (ev > 0.5f * NEG_BIG) ? __expf(ev - m) : 0.f;
            const float ssum = wave_sum_dpp(wv);
            // thread c2: context columns 2*c2, 2*c2+1
            const int c2 = tz;
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
            // context columns 4 at a time as halves (pack4h): the even threads take their neighbour's pair
            const float y0 = __shfl_down(x0, 1), y1 = __shfl_down(x1, 1);
            if (2 * c2 < E && (c2 & 1) == 0) {
                if (local) publish<true>(rec + ghead + (c2 >> 1), pack4h(x0, x1, y0, y1, want)); else publish<false>(rec + ghead + (c2 >> 1), pack4h(x0, x1, y0, y1, want));
            }
            // e pairs, (m, s) and the pad granule by the last compute wave's lanes
            if (wave == NCW - 1) {
                const float e0 = __shfl(ev, min(2 * lane, 63)), e1 = __shfl(ev, min(2 * lane + 1, 63));
                if (lane < TE / 2) {
                    if (local) publish<true>(rec + lane, pack2(e0, e1, want)); else publish<false>(rec + lane, pack2(e0, e1, want));
                } else if (lane == TE / 2) {
                    if (local) publish<true>(rec + lane, pack2(m, ssum, want)); else publish<false>(rec + lane, pack2(m, ssum, want));
                } else if (lane == TE / 2 + 1 && ghead + E / 4 < p.SG2) {
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
            const float M = wave_max_dpp(mi);
            const float wi = (lane < NT) ? si * __expf(mi - M) : 0.f;
            const float S = fmaxf(wave_sum_dpp(wi), 1e-30f);
            const float invS = 1.f / S;
            if (lane < 32) s_scale[wave][lane] = (lane < NT) ? __expf(mi - M) * invS : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int tau = tz; tau < Tp; tau += 64 * NCW) {
                const int i = tau / TE, f = tau - i * TE;
                const float ev = s_stage[i * SG2f + f];
                const float av = (ev > 0.5f * NEG_BIG) ? __expf(ev - M) * invS : 0.f;
                s_attp[Ks + tau] = av;
                if (i == j) p.s.att[row * Tp + tau] = av;
            }
            for (int c2_ = tz; 2 * c2_ < E; c2_ += 64 * NCW) {           // two columns per staged word (halves, pack4h)
                float a0 = 0.f, a1 = 0.f;
                for (int i = 0; i < NT; ++i) {
                    const unsigned w_ = __float_as_uint(s_stage[i * SG2f + TE + 2 + c2_]);
                    a0 += h2f_lo(w_) * s_scale[wave][i]; a1 += h2f_hi(w_) * s_scale[wave][i];
                }
                const int c = 2 * c2_;
                s_x[c] = a0; s_x[c + 1] = a1;
                if (c >= c_base && c < c_base + p.CPW) p.s.xin[row * XW + Dd + c] = a0;
                if (c + 1 >= c_base && c + 1 < c_base + p.CPW) p.s.xin[row * XW + Dd + c + 1] = a1;
            }
        }
        DP_MARK(9)
        compute_barrier(&s_bar, gen);                                   // c5: s_x holds [ctx_t | h_{t-1}]
        DP_MARK(10)
        // ---- LSTM cell: gate rows r = g*UPW + ul of this workgroup, RPW rows per wave, lanes over 16-byte chunks
        //      RBF = 10 rows per batch of loads: the 10 rows per wave of the bench shape (80 rows, 8 waves) are ONE L2 round trip, not
        //      two.  The phase stays near the L2's bandwidth: 32 workgroups per XCD x 151 KB of rows per step.  Keeping rows in registers
        //      across all twelve waves was tried (round 3): the 168-register budget holds 40 of the 80 rows without spilling
        //      (4 per polling wave beside its 10-wide polls, 3 per compute wave); cell rows 2.3 -> 1.9 us, but the compute waves' other
        //      phases lost as much (stats + ctx 1.2 -> 1.8 us) - not adopted.
        {
            constexpr int RBF = 10;
            const int nchunk = p.KCP >> 3;
            float mine = 0.f;
#pragma unroll 1
            for (int bt = 0; bt * RBF < RPW; ++bt) {
                float part[RBF];
#pragma unroll
                for (int rr = 0; rr < RBF; ++rr) part[rr] = 0.f;
                for (int ch0 = lane; ch0 < nchunk; ch0 += 128) {
                    uint4 wv[RBF][2];
#pragma unroll
                    for (int rr = 0; rr < RBF; ++rr) {
                        const int r = min(wave * RPW + bt * RBF + rr, 4 * p.UPW - 1);
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
                            for (int rr = 0; rr < RBF; ++rr) {
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
                for (int rr = 0; rr < RBF; ++rr) {
                    const float sv = wave_sum_dpp(part[rr]);
                    if (lane == bt * RBF + rr) mine = sv;
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
            // the hop starts at the publish: the activations use the fast exp / rcp forms (1 - 2 ulp) and the granule leaves
            // BEFORE the six stores of the step's saved state (one in-order vector-memory queue per wave)
            float ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f;
            if (uok) {
                ai = sigm_f(s_g[lane]); af = sigm_f(s_g[p.UPW + lane]);
                ag = tanh_f(s_g[2 * p.UPW + lane]); ao = sigm_f(s_g[3 * p.UPW + lane]);
                c_state = af * c_state + ai * ag;
                hv = ao * tanh_f(c_state);
            }
            const float hn = __shfl_down(hv, 1);
            if (t + 1 < L && (lane & 1) == 0 && lane < 2 * p.HG2) {
                u64* dst = out + (long)j * p.HG2 + (lane >> 1);
                if (local) publish<true>(dst, pack2(hv, hn, want)); else publish<false>(dst, pack2(hv, hn, want));
            }
            if (uok) {
                float* go = p.s.gates + row * 4 * Dd;
                go[unit] = ai; go[Dd + unit] = af; go[2 * Dd + unit] = ag; go[3 * Dd + unit] = ao;
                p.s.cs[row * Dd + unit] = c_state;
                p.s.hs[row * Dd + unit] = hv;
            }
        }
        DP_MARK(13)
    }
    DP_DUMP
}

struct PersistPlan { bool ok; int tpw, NT, TE, UPW, QPW, CPW, HG2, QG2, SG2, KC, KCP; size_t lds, status_bytes, xbuf_bytes, wcat_bytes, emb_bytes, total; };

PersistPlan persist_plan(const asr_dec_dims_t& d) {
    PersistPlan pl{};
    pl.ok = false;
    if (d.NL != 1 || d.B > 64 || d.A > 16 * FSW_NU * NCW || d.Kn > 10 || (d.E & 7) != 0 || d.Dd > 20 * 32 || d.Tp < 1) return pl;
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
    pl.HG2 = even((pl.UPW + 1) / 2); pl.QG2 = even((pl.QPW + 1) / 2); pl.SG2 = even((pl.TE + 2) / 2 + d.E / 4);      // record: e pairs, (m, s), context as halves
    pl.KC = d.E + d.Dd; pl.KCP = (pl.KC + 7) & ~7;
    const int taps = 2 * d.Ks + 1;
    size_t fl = 0;
    const int epl = 16 * cdiv(pl.TE, 16);
    fl += 2 * pl.KCP + 2 * ((d.A + 3) & ~3) + ((pl.NT * pl.TE + 2 * d.Ks + 8 + 3) & ~3) +
          (size_t)NCW * epl + ((4 * pl.UPW + 3) & ~3) + (size_t)pl.NT * 2 * pl.SG2;
    pl.lds = 2 * (size_t)(((pl.TE * d.A + 7) & ~7) + ((pl.TE * d.E + 7) & ~7) + epl * FCVX_LD + conv_img_shorts(conv_geo(epl, d.Ks))) + 4 * fl;
    if (getenv("ASR_DEC_PLAN_DEBUG")) fprintf(stderr, "[asr] resident fwd plan B=%d T'=%d: NT=%d TE=%d UPW=%d LDS=%zu\n", d.B, d.Tp, pl.NT, pl.TE, pl.UPW, pl.lds);
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

// 0: per-step kernels, 1: LDS-resident tiles (this file), 2: streamed tiles (decoder_stream.hip)
int dec_fwd_plan_kind(const asr_dec_dims_t& d) {
    stream_env();
    const PersistPlan pl = persist_plan(d);
    if (pl.ok && !g_stream_fwd) return 1;
    return dec_fwd_stream_work_bytes(d) ? 2 : (pl.ok ? 1 : 0);
}
size_t dec_fwd_persist_work_bytes(const asr_dec_dims_t& d) {
    const PersistPlan pl = persist_plan(d);                 // sized for whichever plan may be taken (the preference can be switched)
    return std::max(pl.ok ? pl.total : (size_t)0, dec_fwd_stream_work_bytes(d));
}

// Returns ASR_OK when the whole loop was launched, 1 when the configuration has no persistent plan, negative on error.
int dec_fwd_persistent(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const float* enc,
                       const int64_t* enc_len, void* work, size_t work_bytes, hipStream_t st) {
    if (g_persist_fwd < 0) { const char* e = getenv("ASR_DEC_PERSIST"); g_persist_fwd = (e && e[0] == '0') ? 0 : 1; }
    if (!g_persist_fwd) return 1;
    if (dec_fwd_plan_kind(d) == 2) return dec_fwd_streamed(d, w, s, enc, enc_len, work, work_bytes, st);
    const PersistPlan pl = persist_plan(d);
    if (!pl.ok || !work || work_bytes < pl.total || ((uintptr_t)work & 255) != 0 || !s.conv) return 1;
    char* base = (char*)work;
    unsigned* status = (unsigned*)base;
    u64* xbuf = (u64*)(base + pl.status_bytes);
    unsigned short* wcat16 = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes);
    float* embproj = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes);
    clear_work(work, pl.xbuf_bytes, st);
    hipLaunchKernelGGL(build_wcat16_kernel, dim3(512), dim3(256), 0, st, w.Wih[0], w.Whh[0], wcat16, 4 * d.Dd, d.Dd, d.E, pl.KCP);
    const int XW = d.Dd + d.E;
    int rc = asr_gemm(s.xin, w.Wih[0], embproj, nullptr, d.B * d.L, 4 * d.Dd, d.Dd, XW, XW, 4 * d.Dd, 1, 1, ASR_ACT_NONE, 0, 1, 1, 0, 0, 0,
                      0, 0, ASR_BF16, (asr_stream_t)st);
    if (rc != ASR_OK) return rc;
    static int allow = -1;
    if (allow < 0) { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); allow = (e && e[0] == '0') ? 0 : 1; }
    PD p{d, w, s, enc, enc_len, wcat16, embproj, xbuf, status, pl.NT, pl.TE, pl.UPW, pl.QPW, pl.CPW, pl.HG2, pl.QG2, pl.SG2, pl.KC, pl.KCP, allow};
    const int cpx = cdiv(d.B, 8);
    const dim3 grid(8 * cpx * pl.NT), block(64 * (NCW + NPW));
#define DPF_LAUNCH(KN_, TPW_)                                                                                                   \
    {                                                                                                                           \
        static unsigned char attr_[32];                                                                                         \
        if (first_on_device(attr_)) hipFuncSetAttribute((const void*)dec_fwd_persist<KN_, TPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); \
        if (!grid_resident(dec_fwd_persist<KN_, TPW_>, (int)grid.x, (int)block.x, pl.lds)) return 1;                              \
        hipLaunchKernelGGL((dec_fwd_persist<KN_, TPW_>), grid, block, pl.lds, st, p);                                             \
        hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(1), 0, st, status);                                                   \
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

// =================================================================================================
// Backward of the teacher-forced decoder loop as ONE persistent launch (same cluster-per-utterance scheme as the
// forward; bf16 mode, one decoder layer).  Workgroup (tile j of utterance b) keeps for all L steps
//   * in REGISTERS: its rows of [W_ih(ctx) | W_hh]^T (bf16, 8 rows x 20 columns per lane over all 8 waves), its rows of
//     W_q^T, and the accumulators the per-step kernels had to read-modify-write in HBM every step: d w_g[a], d W_proj[a,:],
//     the cell-state carry;
//   * in LDS: the key tile (bf16), W_proj (bf16, MFMA operand), W_conv;
//   * dkey is accumulated with no-return atomics (one add per address per step, issued by the same thread: deterministic).
// Every workgroup recomputes the (elementwise) cell backward of ALL hidden units from the full dh vector, so a step
// needs three all-gathers inside the cluster (tagged granules):
//   C  {dctx slice, recurrent part of dh_{t-1} for the workgroup's units}
//   Q  per-tile partial of the query gradient,  V  the tile's dconv
//   N  {the tile's gradient wrt the previous attention, query part of dh_{t-1} for the workgroup's units}
// Waves 0..ncw-1 (ncw = ceil(A/64)) compute, three more waves poll; all eight take part in the transposed-weight product.
// =================================================================================================
namespace {

// (constants, PB, the LDS carve, P1 macros, the sweep: decoder_bwd_common.h)
// TEC: frames per tile as a compile-time constant (the bench shape), 0: taken from the plan (any multiple of 4 up to 40)
template <int KNMAX, int TEC>
__global__ __launch_bounds__(512) void dec_bwd_persist(PB p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_bar;
    __shared__ float s_red[8];
    constexpr int KP = (KNMAX + 3) & ~3;
    const int TE = TEC ? TEC : p.TE;
    const int MT = (TE + 15) / 16;
    const asr_dec_dims_t& d = p.d;
    const int id = blockIdx.x, xcd = id & 7, slot_id = id >> 3;
    const int cb = slot_id / p.NT, j = slot_id - cb * p.NT;
    const int b = cb * 8 + xcd;
    if (b >= d.B) return;
    const unsigned epoch_ = __builtin_amdgcn_readfirstlane(p.status[EPOCH_WORD]);     // launch epoch of this work area (decoder_cluster.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = p.NT, A = d.A, E = d.E, Dd = d.Dd, Tp = d.Tp, Kn = d.Kn, Ks = d.Ks, L = d.L;
    const int ncw = (A + 63) >> 6, nct = 64 * ncw, nw = ncw + NPB;       // compute waves / threads, all waves
    const int taps = 2 * Ks + 1, XW = Dd + E, R4 = p.R4;
    const int tau0 = j * TE;
    const int len = min((int)p.enc_len[b], Tp);
    const int tmax = max(len - 1, 0);
    const BCarve cv_ = bwd_carve(TE, KP, A, E, Kn, Ks, NT, p.UPW, p.CG2, p.QG2, p.NG2);
    const int AP = cv_.AP, DW = cv_.DW, PADL = cv_.PADL, WT = cv_.WT, AQ = cv_.AQ;
    const int TW = NT * TE;                                             // padded attention row
    const int CG2f = 2 * p.CG2, NG2f = 2 * p.NG2, QG2f = 2 * p.QG2;
    // ---- LDS carve
    unsigned short* s_sh = reinterpret_cast<unsigned short*>(smem);
    unsigned short* s_key = s_sh + cv_.key;                                              // [TE/4][A][4] bf16 (four frames innermost)
    unsigned short* s_dl = s_sh + cv_.dl;                                                // [TE][AP] bf16  d loc pre-activation
    unsigned short* s_wp16 = s_sh + cv_.wp16;                                            // [16][AP] bf16  W_proj^T, zero padded
    unsigned short* s_dg16 = s_sh + cv_.dg16;                                            // [1280] bf16 dgates of the utterance
    unsigned short* s_wq16 = s_sh + cv_.wq16;                                            // [UPW][AQ] bf16 rows of W_q^T of the own units
    unsigned short* s_cvx = s_sh + cv_.cvx;                                              // [48][32] bf16 conv tile, slots {hi | lo | hi}
    unsigned short* s_cvT = s_sh + cv_.cvT;                                              // [16][CVT_LD] bf16 conv tile transposed
    float* s_f = reinterpret_cast<float*>(s_sh + cv_.shorts);
    float* s_wc = s_f + cv_.wc;                                                          // [Kn*taps]
    float* s_crec = s_f + cv_.crec;                                                      // [NT][CG2*2] records {dctx slice, dh_rec slice}
    float* s_qst = s_f + cv_.qst;                                                        // [NT][QG2*2] dq partials
    float* s_nrec = s_f + cv_.nrec;                                                      // [NT][NG2*2] records {datt_next tile, dh_q slice}
    float* s_dcp = s_f + cv_.dcp;                                                        // [Kn][DW] zero-padded dconv rows
    float* s_de = s_f + cv_.de;                                                          // [48], zero beyond TE
    float* s_out = s_f + cv_.out;                                                        // [RPWB * 8] products of P1
    float* s_hq = s_f + cv_.hq;                                                          // [UPW] query part of dh_{t-1}
    float* s_pt = s_f + cv_.pt;                                                          // [4*Kn*TE] partial sums of datt_next (tap ranges)
    float* s_dcx = s_f + cv_.dcx;                                                        // [E] dctx, contiguous
    float* s_dq = s_f + cv_.dq;                                                          // [A] dq of the utterance
    const long region = (long)NT * (p.CG2 + p.QG2 + p.VG2 + p.NG2);
    auto xb = [&](int parity) { return p.xbuf + ((long)parity * d.B + b) * region; };
    const long offC = 0, offQ = offC + (long)NT * p.CG2, offV = offQ + (long)NT * p.QG2, offN = offV + (long)NT * p.VG2;
    if (tid == 0) s_bar = 0u;
    // clear this producer's records in the L2 (see the forward kernel)
    for (int parity = 0; parity < 2; ++parity) {
        u64* base = xb(parity);
        for (int i = tid; i < p.CG2; i += blockDim.x) st_gran_local(base + offC + (long)j * p.CG2 + i, 0ull);
        for (int i = tid; i < p.QG2; i += blockDim.x) st_gran_local(base + offQ + (long)j * p.QG2 + i, 0ull);
        for (int i = tid; i < p.VG2; i += blockDim.x) st_gran_local(base + offV + (long)j * p.VG2 + i, 0ull);
        for (int i = tid; i < p.NG2; i += blockDim.x) st_gran_local(base + offN + (long)j * p.NG2 + i, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.status) + 64 + b, NT, p.allow_local, p.status);

    // ---- resident data
    for (int i = tid; i < TE * A; i += blockDim.x) {
        const int f = i / A, a = i - f * A;
        s_key[((long)(f >> 2) * A + a) * 4 + (f & 3)] = f2bf_bits(p.s.key[((long)b * Tp + min(tau0 + f, tmax)) * A + a]);
    }
    for (int i = tid; i < 16 * AP; i += blockDim.x) { const int k = i / AP, a = i - k * AP; s_wp16[i] = (k < Kn && a < A) ? f2bf_bits(p.w.Wproj[a * Kn + k]) : (unsigned short)0; }
    for (int i = tid; i < Kn * WT; i += blockDim.x) { const int k = i / WT, jj = i - k * WT; s_wc[i] = (jj < taps) ? p.w.Wconv[k * taps + jj] : 0.f; }
    for (int i = tid; i < 64 * KCHB * 4; i += blockDim.x) s_dg16[i] = 0;
    for (int i = tid; i < NT * NG2f + 8; i += blockDim.x) s_nrec[i] = 0.f;
    for (int i = tid; i < NT * CG2f; i += blockDim.x) s_crec[i] = 0.f;
    for (int i = tid; i < Kn * DW; i += blockDim.x) s_dcp[i] = 0.f;
    for (int i = tid; i < TE * AP; i += blockDim.x) s_dl[i] = 0;
    for (int i = tid; i < 16 * SW_MT * CVX_LD; i += blockDim.x) s_cvx[i] = 0;
    for (int i = tid; i < 16 * CVT_LD; i += blockDim.x) s_cvT[i] = 0;
    for (int i = tid; i < 16 * SW_MT; i += blockDim.x) s_de[i] = 0.f;
    const int u_base = j * p.UPW, c_base = j * p.CPW;
    for (int i = tid; i < p.UPW * AQ; i += blockDim.x) {
        const int ul = i / AQ, aa = i - ul * AQ;
        s_wq16[i] = (aa < A && u_base + ul < Dd) ? f2bf_bits(p.wqT[(long)(u_base + ul) * A + aa]) : (unsigned short)0;
    }
    const int nout = p.CPW + p.UPW;
    // sweep: 16-column units over ALL waves (unit u -> wave u % nw), see sweep_step
    const int nunits = (A + 15) >> 4;
    const int nu_cnt = (nunits - wave + nw - 1) / nw;              // units of this wave (<= SW_NU by the plan)
    const int qsub = lane >> 4, csub = lane & 15;
    // rows of the transposed cell weights are register-resident: compute wave w has outputs w + ncw*o (o < RCB), polling
    // wave pw has RCB*ncw + pw + NPB*o (o < RPB)
    __syncthreads();

    if (wave >= ncw) {
        // =========================== polling role ===========================
        const int gt = tid - nct, np = 64 * NPB;
        const int vp0 = max(0, (tau0 - Ks) / TE), vp1 = min(NT - 1, (tau0 + TE - 1 + Ks) / TE);   // tiles whose dconv this tile's datt_next reads
        const int obase = RCB * ncw + (wave - ncw);
        uint2 wreg[RPB][KCHB];
        DPB_WLOAD(RPB, obase, NPB)
        Sweep<SW_NUP> S;                                                // the polling waves own at most two units (plan)
        sweep_init<KNMAX>(S, p.w.Wproj, p.w.wg, A, Kn, wave, nw, lane);
        for (int t = L - 1; t >= 0; --t) {
            const int s = L - 1 - t;                                     // step counter of this launch (tags, parity)
            const u64 want = pair_want(seq_of(s), epoch_);
            const u64* base = xb(s & 1);
            DP_JIT(6)
            __syncthreads();                                            // Ba: s_dg16 holds the gate gradients of step t
            DPB_P1(RPB, obase, NPB)
            __syncthreads();                                            // Bb: s_out complete
            poll_copy<4>(base + offC, NT * p.CG2 / 2, s_crec, gt, np, want, p.status);
            float qa[SW_NUP];
#pragma unroll
            for (int nu = 0; nu < SW_NUP; ++nu) qa[nu] = p.s.q[((long)b * L + t) * A + min(16 * (wave + nw * nu) + csub, A - 1)];
            DP_JIT(7)
            __syncthreads();                                            // H2
            DP_JIT(8)
            __syncthreads();                                            // X1: s_de, s_cvx, s_cvT complete
            DPB_SWEEP_AND_PUBLISH(const_cast<u64*>(base), SW_NUP)
            __syncthreads();                                            // X2: s_dl complete
            for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(127);
            {   // Q records of all tiles (flat copy) and the dconv tiles V of the tiles within reach of the location filter
                // (they go straight into the zero-padded per-kernel rows s_dcp[k][PADL + prod*TE + f]), as ONE polling sweep
                const int nq = NT * p.QG2 / 2, nv = (vp1 - vp0 + 1) * (p.VG2 / 2);
                const u64* vbase = base + offV + (long)vp0 * p.VG2;
                for (int i0 = gt; i0 < nq + nv; i0 += 10 * np) {
                    u64 lo[10], hi[10];
                    const u64* addr[10];
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) {
                        const int idx = i0 + k * np;
                        if (idx < nq + nv) cnt = k + 1;
                        addr[k] = (idx < nq) ? base + offQ + 2 * (long)idx : vbase + 2 * (long)min(idx - nq, nv - 1);
                    }
                    gather16v<10>(addr, cnt, PAIR_MASK, want, lo, hi, p.status);
#pragma unroll
                    for (int k = 0; k < 10; ++k)
                        if (k < cnt) {
                            const int idx = i0 + k * np;
                            const float v[4] = {lo_f(lo[k]), hi_f(lo[k]), lo_f(hi[k]), hi_f(hi[k])};
                            if (idx < nq) {
                                *reinterpret_cast<float4*>(s_qst + 4 * (long)idx) = make_float4(v[0], v[1], v[2], v[3]);
                            } else {
                                const int g0 = vp0 * p.VG2 + 2 * (idx - nq), prod = g0 / p.VG2, r0 = 2 * (g0 - prod * p.VG2);
#pragma unroll
                                for (int q4 = 0; q4 < 4; ++q4) {
                                    const int r = r0 + q4;
                                    if (r < Kn * TE) { const int kk = r / TE, f = r - kk * TE; s_dcp[kk * DW + PADL + prod * TE + f] = v[q4]; }
                                }
                            }
                        }
                }
            }
            DP_JIT(9)
            __syncthreads();                                            // H3
            DP_JIT(10)
            if (t > 0) poll_copy<4>(base + offN, NT * p.NG2 / 2, s_nrec, gt, np, want, p.status);
            __syncthreads();                                            // H4
        }
        DPB_SWEEP_RESULTS(SW_NUP)
        return;
    }

    // =========================== compute role ===========================
    unsigned gen = 0;
    uint2 wreg[RCB][KCHB];
    DPB_WLOAD(RCB, wave, ncw)
    Sweep<SW_NU> S;
    sweep_init<KNMAX>(S, p.w.Wproj, p.w.wg, A, Kn, wave, nw, lane);
    const int a = tid;                                                  // attention column of this thread in the sweep
    const bool aok = a < A;
    const int ac = aok ? a : A - 1;
    const bool uok = tid < Dd;                                          // hidden unit of this thread in the cell backward
    const int uc = uok ? tid : Dd - 1;
    const int ui = uc / p.UPW, uul = uc - ui * p.UPW;                   // its producer and position in the records
    const bool uown = uok && ui == j;
    float dbg = 0.f, dc_carry = 0.f;
    // operands of the cell backward of step L-1 (later steps: requested one step ahead)
    float pgi, pgf, pgg, pgo, pct, pcp, pdh;
    {
        const long r0 = (long)b * L + (L - 1);
        const float* g = p.s.gates + r0 * 4 * Dd + uc;
        pgi = g[0]; pgf = g[Dd]; pgg = g[2 * Dd]; pgo = g[3 * Dd];
        pct = p.s.cs[r0 * Dd + uc];
        pcp = (L > 1) ? p.s.cs[(r0 - 1) * Dd + uc] : 0.f;
        pdh = p.dhs[r0 * Dd + uc];
    }
    DP_DECL

    for (int t = L - 1; t >= 0; --t) {
        // Thread-index arithmetic of the step is loop invariant; hoisted, it becomes ~100 live address registers and the
        // accumulators get spilled instead.  An opaque copy of the index ties it to the step (a few dozen VALU per step).
        int tz = tid, lz_ = lane;
        asm volatile("" : "+v"(tz), "+v"(lz_));
        const int lane = lz_;
        const int s = L - 1 - t;
        const long row = (long)b * L + t;
        const u64 want = pair_want(seq_of(s), epoch_);
        u64* out = xb(s & 1);
        DP_MARK(0)
        // ---- S1: cell backward of ALL hidden units (thread per unit; every workgroup of the cluster computes the same)
        {
            float dh = pdh;
            if (s > 0) dh += s_crec[ui * CG2f + p.CPW + uul] + s_nrec[ui * NG2f + TE + uul];
            const float tc = tanhf(pct);
            const float dc = dh * pgo * (1.f - tc * tc) + dc_carry;
            const float d0 = dc * pgg * pgi * (1.f - pgi), d1 = dc * pcp * pgf * (1.f - pgf);
            const float d2 = dc * pgi * (1.f - pgg * pgg), d3 = dh * tc * pgo * (1.f - pgo);
            dc_carry = dc * pgf;
            if (uok) {
                s_dg16[tz] = f2bf_bits(d0); s_dg16[Dd + tz] = f2bf_bits(d1);
                s_dg16[2 * Dd + tz] = f2bf_bits(d2); s_dg16[3 * Dd + tz] = f2bf_bits(d3);
            }
            if (uown) {
                float* go = p.dgates + row * 4 * Dd + tz;
                go[0] = d0; go[Dd] = d1; go[2 * Dd] = d2; go[3 * Dd] = d3;
            }
        }
        DP_MARK(1)
        __syncthreads();                                                // Ba
        DP_MARK(2)
        // ---- P1: dctx slice and the recurrent part of dh_{t-1} for the own units
        DPB_P1(RCB, wave, ncw)
        DP_MARK(3)
        __syncthreads();                                                // Bb
        DP_MARK(4)
        // ---- C record {dctx slice | dh_rec slice} + global dxin (context part)
        if (tz < p.CG2) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * tz + h;
                const bool ok = (i < p.CPW) ? (c_base + i < E) : (i < nout && u_base + (i - p.CPW) < Dd);
                v[h] = ok ? s_out[min(i, RPWB * 8 - 1)] : 0.f;
                if (i < p.CPW && c_base + i < E) p.dxin[row * XW + Dd + c_base + i] = v[h];
            }
            u64* dst = out + offC + (long)j * p.CG2 + tz;
            if (local) publish<true>(dst, pack2(v[0], v[1], want)); else publish<false>(dst, pack2(v[0], v[1], want));
        }
        // ---- operands of P2/P3 that do not depend on the hand-offs: requested now, used behind H2
        float qa[SW_NU];
#pragma unroll
        for (int nu = 0; nu < SW_NU; ++nu) qa[nu] = p.s.q[row * A + min(16 * (wave + nw * nu) + csub, A - 1)];
        const int f2 = tz >> 3, part = tz & 7;                         // P2: 8 threads per frame
        uint4 x[10];
        {
            const unsigned short* er = p.enc16 + ((long)b * Tp + min(tau0 + min(f2, TE - 1), tmax)) * E;
#pragma unroll
            for (int u = 0; u < 10; ++u) x[u] = *reinterpret_cast<const uint4*>(er + 8 * min(part + 8 * u, (E >> 3) - 1));
        }
        float ctx2[2], attv[4];
#pragma unroll
        for (int h = 0; h < 2; ++h) ctx2[h] = p.s.xin[row * XW + Dd + min(tz + nct * h, E - 1)];
#pragma unroll
        for (int h = 0; h < 4; ++h) attv[h] = p.s.att[row * Tp + min(tz + nct * h, Tp - 1)];
        const float attf = p.s.att[row * Tp + min(tau0 + min(f2, TE - 1), Tp - 1)];
        float c0, c1;                                                   // conv tile of the step (stored to LDS behind H2)
        const int cvi0 = tz, cvi1 = tz + nct;
        const int cvk0 = min(cvi0, Kn * TE - 1) / TE, cvf0 = min(cvi0, Kn * TE - 1) - cvk0 * TE;
        const int cvk1 = min(cvi1, Kn * TE - 1) / TE, cvf1 = min(cvi1, Kn * TE - 1) - cvk1 * TE;
        c0 = p.s.conv[(row * Kn + cvk0) * Tp + min(tau0 + cvf0, Tp - 1)];
        c1 = p.s.conv[(row * Kn + cvk1) * Tp + min(tau0 + cvf1, Tp - 1)];
        DP_MARK(5)
        __syncthreads();                                                // H2: s_crec holds the C records of all workgroups
        DP_MARK(6)
        // ---- P2: dattn of the tile, dot over the utterance, de
        {
            for (int e = tz; e < E; e += nct) { const int i = e / p.CPW; s_dcx[e] = s_crec[i * CG2f + (e - i * p.CPW)]; }
            if (cvi0 < Kn * TE) put_cv<KNMAX>(s_cvx, s_cvT, cvf0, cvk0, (tau0 + cvf0 < Tp) ? c0 : 0.f);
            if (cvi1 < Kn * TE) put_cv<KNMAX>(s_cvx, s_cvT, cvf1, cvk1, (tau0 + cvf1 < Tp) ? c1 : 0.f);
            cbar(&s_bar, gen, ncw);
            float v = 0.f;
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                const int ch = part + 8 * u;
                if (8 * ch < E) {
                    const float4 da = *reinterpret_cast<const float4*>(s_dcx + 8 * ch), db = *reinterpret_cast<const float4*>(s_dcx + 8 * ch + 4);
                    v += __uint_as_float(x[u].x << 16) * da.x + __uint_as_float(x[u].x & 0xffff0000u) * da.y +
                         __uint_as_float(x[u].y << 16) * da.z + __uint_as_float(x[u].y & 0xffff0000u) * da.w +
                         __uint_as_float(x[u].z << 16) * db.x + __uint_as_float(x[u].z & 0xffff0000u) * db.y +
                         __uint_as_float(x[u].w << 16) * db.z + __uint_as_float(x[u].w & 0xffff0000u) * db.w;
                }
            }
            v = sum8_dpp(v);
            float dot = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) if (tz + nct * h < E) dot += ctx2[h] * s_dcx[tz + nct * h];
            if (s > 0) {
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int tau = tz + nct * h;
                    if (tau < len) { const int i = tau / TE; dot += attv[h] * s_nrec[i * NG2f + (tau - i * TE)]; }
                }
            }
            dot = wave_sum_dpp(dot);
            if (lane == 0) s_red[wave] = dot;
            cbar(&s_bar, gen, ncw);
            dot = 0.f;
            for (int w = 0; w < ncw; ++w) dot += s_red[w];
            if (f2 < TE && part == 0) {
                const int tau = tau0 + f2;
                const float dat = v + ((s > 0) ? s_nrec[j * NG2f + f2] : 0.f);
                const float dev = (tau < len) ? attf * (dat - dot) / d.temperature : 0.f;
                s_de[f2] = dev;
                dbg += dev;                                             // d b_g: per-thread partial, added to the slot at the end
            }
        }
        __syncthreads();                                                // X1: s_de, s_cvx, s_cvT complete (the polling waves join the sweep)
        DP_MARK(7)
        // ---- P3: energy backward sweep of this wave's column units + their query-gradient partials (Q record)
        DPB_SWEEP_AND_PUBLISH(out, SW_NU)
        __syncthreads();                                                // X2: s_dl complete
        DP_MARK(8)
        // ---- P4: dconv (TE x Kn) = dl (TE x A) . W_proj (A x Kn) on the matrix cores, one 16-frame tile per wave
        if (wave < MT) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int m = lane & 15, kq = lane >> 4;
            const unsigned short* ar = s_dl + min(16 * wave + m, TE - 1) * AP + 8 * kq;
            const unsigned short* br = s_wp16 + m * AP + 8 * kq;
            const int nks = (A + 31) >> 5;
            for (int ks = 0; ks < nks; ++ks) {
                const bf16x8 av = *reinterpret_cast<const bf16x8*>(ar + 32 * ks);
                const bf16x8 bv = *reinterpret_cast<const bf16x8*>(br + 32 * ks);
                acc = mma16(av, bv, acc);
            }
            if (m < Kn) {
#pragma unroll
                for (int i = 0; i < 4; i += 2) {
                    const int f = 16 * wave + 4 * kq + i;
                    if (f < TE) {
                        const float v0 = (tau0 + f < len) ? acc[i] : 0.f, v1 = (tau0 + f + 1 < len) ? acc[i + 1] : 0.f;
                        // dconv over the saved conv (for d W_conv after the loop)
                        if (tau0 + f < Tp) p.s.conv[(row * Kn + m) * Tp + tau0 + f] = v0;
                        if (tau0 + f + 1 < Tp) p.s.conv[(row * Kn + m) * Tp + tau0 + f + 1] = v1;
                        u64* dst = out + offV + (long)j * p.VG2 + ((m * TE + f) >> 1);      // record order [k][f]
                        if (local) publish<true>(dst, pack2(v0, v1, want)); else publish<false>(dst, pack2(v0, v1, want));
                    }
                }
            }
        }
        if (tz == nct - 1 && (TE * Kn + 1) / 2 < p.VG2) {
            u64* dst = out + offV + (long)j * p.VG2 + p.VG2 - 1;
            if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
        }
        DP_MARK(9)
        __syncthreads();                                                // H3: dq partials and dconv tiles of all workgroups
        DP_MARK(10)
        // ---- P5: dq (sum over tiles), its part of dh_{t-1}, datt_next of the tile
        {
            float dqv = 0.f;
            if (aok) for (int i = 0; i < NT; ++i) dqv += s_qst[i * QG2f + a];
            if (aok) s_dq[a] = dqv;
            const int asl = (A + NT - 1) / NT;
            if (aok && a >= j * asl && a < (j + 1) * asl) p.dq[row * A + a] = dqv;     // each workgroup saves a slice
        }
        DP_MARK(13)
        if (t > 0) {
            // datt_next[tau'] = sum_k sum_j W_conv[k][j] * dconv[k][tau' - j + Ks] for the tile's frames:
            // item = (tap range, kernel, group of 4 outputs); four taps per round from three 16-byte LDS reads
            const int ngrp = TE / 4;
            const int nitem = Kn * ngrp;
            const int parts = max(1, min(4, nct / nitem));
            const int gpp = (WT / 4 + parts - 1) / parts;                 // tap groups per part
            for (int it = tz; it < parts * nitem; it += nct) {
                const int pz = it / nitem, o = it - pz * nitem, k = o / ngrp, ig = o - k * ngrp;
                const int g0 = pz * gpp, g1 = min(WT / 4, g0 + gpp);
                // 16-byte units throughout (s_wc rows, the dconv rows and PADL + Ks + tau0 are multiples of 4 floats)
                const float4* wk4 = reinterpret_cast<const float4*>(s_wc) + (k * WT) / 4;
                const float4* q4 = reinterpret_cast<const float4*>(s_dcp) + (k * DW + PADL + Ks + tau0) / 4 + ig;   // q4[0] = q[0..3]
                float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 3
                for (int g = g0; g < g1; ++g) {
                    const float4 w4 = wk4[g];
                    const float4 lo = q4[-g - 1];                       // q[-4..-1] of this tap group
                    const float4 hi = q4[-g];                           // q[0..3]
                    a0 += w4.x * hi.x + w4.y * lo.w + w4.z * lo.z + w4.w * lo.y;
                    a1 += w4.x * hi.y + w4.y * hi.x + w4.z * lo.w + w4.w * lo.z;
                    a2 += w4.x * hi.z + w4.y * hi.y + w4.z * hi.x + w4.w * lo.w;
                    a3 += w4.x * hi.w + w4.y * hi.z + w4.z * hi.y + w4.w * hi.x;
                }
                *reinterpret_cast<float4*>(s_pt + (long)(pz * Kn + k) * TE + 4 * ig) = make_float4(a0, a1, a2, a3);
            }
            DP_MARK(14)
            cbar(&s_bar, gen, ncw);                                     // s_dq, s_pt complete
            // query part of dh_{t-1}: sum_a dq[a] * W_q[a][unit] for the own units (bf16 rows in LDS)
            {
                float dqr[5];
#pragma unroll
                for (int k5 = 0; k5 < 5; ++k5) dqr[k5] = (lane + 64 * k5 < A) ? s_dq[min(lane + 64 * k5, A - 1)] : 0.f;
                for (int ul = wave; ul < p.UPW; ul += ncw) {
                    const unsigned short* wr = s_wq16 + ul * AQ + lane;
                    float acc = 0.f;
#pragma unroll
                    for (int k5 = 0; k5 < 5; ++k5) if (64 * k5 < AQ) acc += dqr[k5] * bf2f_(wr[64 * k5]);
                    acc = wave_sum_dpp(acc);
                    if (lane == 0) s_hq[ul] = acc;
                }
            }
            // datt_next of the tile: the tap-range partial sums, 8 threads per frame (s_de is free after the sweep)
            {
                const int i = tz >> 3, sub = tz & 7;
                float sv = 0.f;
                if (i < TE) for (int r = sub; r < parts * Kn; r += 8) sv += s_pt[(long)r * TE + i];
                sv = sum8_dpp(sv);
                if (i < TE && sub == 0) s_de[i] = sv;
            }
            DP_MARK(15)
            cbar(&s_bar, gen, ncw);                                     // s_hq, datt_next complete
            // N record {datt_next tile | dh_q slice}
            if (tz < p.NG2) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * tz + h;
                    v[h] = (i < TE) ? s_de[i] : ((i - TE < p.UPW) ? s_hq[i - TE] : 0.f);
                }
                u64* dst = out + offN + (long)j * p.NG2 + tz;
                if (local) publish<true>(dst, pack2(v[0], v[1], want)); else publish<false>(dst, pack2(v[0], v[1], want));
            }
            // operands of the next step's cell backward (they arrive while the N records are gathered)
            const long r1 = row - 1;
            const float* g = p.s.gates + r1 * 4 * Dd + uc;
            pgi = g[0]; pgf = g[Dd]; pgg = g[2 * Dd]; pgo = g[3 * Dd];
            pct = p.s.cs[r1 * Dd + uc];
            pcp = (t > 1) ? p.s.cs[(r1 - 1) * Dd + uc] : 0.f;
            pdh = p.dhs[r1 * Dd + uc];
        }
        DP_MARK(11)
        __syncthreads();                                                // H4: s_nrec holds the N records for the next step
        DP_MARK(12)
    }
    DP_DUMP
    // ---- results that were accumulated on chip: the slot of this workgroup, the dkey tile
    DPB_SWEEP_RESULTS(SW_NU)
    if (dbg != 0.f) atomicAdd(p.slots + ((long)b * NT + j) * p.slot + A * (1 + Kn), dbg);     // slots are zero on entry
}

struct PersistPlanB { bool ok; int TE, NT, UPW, CPW, R4, CG2, QG2, VG2, NG2; size_t lds, status_bytes, xbuf_bytes, w16_bytes, dg_bytes, total; };

PersistPlanB persist_plan_b(const asr_dec_dims_t& d) {
    PersistPlanB pl{};
    pl.ok = false;
    if (d.NL != 1 || d.B > 64 || d.A > 320 || d.A < 16 || d.Kn > 10 || (d.E & 7) != 0 || (d.A & 1) != 0 || d.Dd > 64 * KCHB || d.L < 1) return pl;
    const int cpx = cdiv(d.B, 8);
    const int ncw = cdiv(d.A, 64), nct = 64 * ncw;
    if (d.Dd > nct || d.E > 2 * nct || d.E > 640 || d.Tp > 4 * nct) return pl;
    pl.R4 = (4 * d.Dd + 7) & ~7;
    auto even = [](int x) { return (x + 1) & ~1; };
    // frames per tile: the smallest multiple of 4 (most tiles, fewest weight rows and sweep frames per workgroup) whose tiles
    // fit the XCD (32 CUs per XCD, ceil(B/8) clusters each), the register-resident weight rows and the LDS
    // Second pass (short encoder outputs, e.g. T' = 150 behind a VGG front-end): more workgroups than the frames need.  The
    // weight rows of the cell and the context columns are spread over ALL tiles of an utterance, so a short utterance can run
    // out of register rows before it runs out of frames; tiles past T' hold rows only (every frame access of the kernel is
    // clamped to the utterance and every frame store guarded by T', as for the tiles past a short utterance of a ragged batch).
    bool found = false;
    for (int pass = 0; pass < 2 && !found; ++pass)
    for (int TE = 8; TE <= 40 && !found; TE += 4) {
        const int nt_frames = cdiv(d.Tp, TE);
        const int nt_hi = pass == 0 ? nt_frames : std::min(30, 32 / cpx);
        for (int nt = nt_frames + pass; nt <= nt_hi && !found; ++nt) {
            if (nt > 30 || cpx * nt > 32 || 8 * TE > nct || d.Kn * TE > 2 * nct) continue;
            pl.TE = TE; pl.NT = nt;
            pl.UPW = cdiv(d.Dd, nt); pl.CPW = cdiv(d.E, nt);
            if (pl.UPW > 64 || pl.UPW + pl.CPW > RCB * ncw + RPB * NPB || (TE + pl.UPW + 1) / 2 + 1 > nct) continue;
            pl.CG2 = even((pl.CPW + pl.UPW + 1) / 2); pl.QG2 = even(d.A / 2); pl.VG2 = even((TE * d.Kn + 1) / 2); pl.NG2 = even((TE + pl.UPW + 1) / 2);
            if (pl.NT * pl.QG2 * 2 < NPB * 64 * 11) continue;       // s_qst doubles as the stage of the polling waves' partial accumulators
            const BCarve cv = bwd_carve(TE, 12, d.A, d.E, d.Kn, d.Ks, pl.NT, pl.UPW, pl.CG2, pl.QG2, pl.NG2);
            pl.lds = 2 * (size_t)cv.shorts + 4 * (size_t)cv.floats;
            if (getenv("ASR_DEC_PLAN_DEBUG")) fprintf(stderr, "[asr] resident bwd plan B=%d T'=%d: NT=%d TE=%d LDS=%zu\n", d.B, d.Tp, pl.NT, pl.TE, pl.lds);
            if (pl.lds > 160 * 1024 - 4096) continue;
            found = true;
        }
    }
    if (!found) return pl;
    pl.status_bytes = 4096;
    pl.xbuf_bytes = align_up256(2 * (size_t)d.B * pl.NT * (pl.CG2 + pl.QG2 + pl.VG2 + pl.NG2) * sizeof(u64));
    pl.w16_bytes = align_up256((size_t)(d.Dd + d.E + d.Dd) * pl.R4 * 2);
    pl.dg_bytes = align_up256((size_t)d.B * d.L * 4 * d.Dd * sizeof(float));
    pl.total = pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes + pl.dg_bytes;
    pl.ok = true;
    return pl;
}


}  // namespace

// decoder_stream.hip
size_t dec_bwd_stream_work_bytes(const asr_dec_dims_t& d);
int dec_bwd_stream_tiles(const asr_dec_dims_t& d);
float* dec_bwd_stream_dgates(const asr_dec_dims_t& d, void* work);
int dec_bwd_streamed(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const int64_t* enc_len,
                     const float* dhs, float* dxin, float* dq, float* dkey, float* slots, int slot, const float* wcatT, const float* wqT,
                     void* work, size_t work_bytes, float** dgates_out, hipStream_t st);

// 0: per-step kernels, 1: tiles resident on chip (this file), 2: streamed tiles (decoder_stream.hip)
int dec_bwd_plan_kind(const asr_dec_dims_t& d) {
    stream_env();
    const PersistPlanB pl = persist_plan_b(d);
    if (pl.ok && !g_stream_bwd) return 1;
    return dec_bwd_stream_work_bytes(d) ? 2 : (pl.ok ? 1 : 0);
}
float* dec_bwd_persist_dgates(const asr_dec_dims_t& d, void* work) {
    if (dec_bwd_plan_kind(d) == 2) return dec_bwd_stream_dgates(d, work);
    const PersistPlanB pl = persist_plan_b(d);
    return (float*)((char*)work + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes);
}
// the work area is sized for whichever plan may be taken for the shape (the preference can be switched between calls)
size_t dec_bwd_persist_work_bytes(const asr_dec_dims_t& d) {
    const PersistPlanB pl = persist_plan_b(d);
    return std::max(pl.ok ? pl.total : (size_t)0, dec_bwd_stream_work_bytes(d));
}
int dec_bwd_persist_tiles(const asr_dec_dims_t& d) {
    const int kind = dec_bwd_plan_kind(d);
    if (kind == 2) return dec_bwd_stream_tiles(d);
    return kind == 1 ? persist_plan_b(d).NT : 0;
}
// slots are sized for the larger tile count of the two plans
int dec_bwd_persist_tiles_max(const asr_dec_dims_t& d) {
    const PersistPlanB pl = persist_plan_b(d);
    return std::max(pl.ok ? pl.NT : 0, dec_bwd_stream_tiles(d));
}

// Returns ASR_OK when the whole backward loop was launched, 1 when there is no persistent plan, negative on error.
// dhs: (B,L,Dd) gradient wrt h from the output layer; wcatT: ((Dd+E+Dd) x 4Dd) fp32 transposed [W_ih | W_hh]; wqT: (Dd x A).
// *dgates_out: (B*L, 4Dd) gate pre-activation gradients inside `work`.
int dec_bwd_persistent(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const int64_t* enc_len,
                       const float* dhs, float* dxin, float* dq, float* dkey, float* slots, int slot, const float* wcatT, const float* wqT,
                       void* work, size_t work_bytes, float** dgates_out, hipStream_t st) {
    if (g_persist_bwd < 0) { const char* e = getenv("ASR_DEC_PERSIST_BWD"); g_persist_bwd = (e && e[0] == '0') ? 0 : 1; }
    if (!g_persist_bwd) return 1;
    if (dec_bwd_plan_kind(d) == 2) return dec_bwd_streamed(d, w, s, enc_len, dhs, dxin, dq, dkey, slots, slot, wcatT, wqT, work, work_bytes, dgates_out, st);
    const PersistPlanB pl = persist_plan_b(d);
    if (!pl.ok || !work || work_bytes < pl.total || ((uintptr_t)work & 255) != 0 || !s.conv || !s.enc16) return 1;
    char* base = (char*)work;
    unsigned* status = (unsigned*)base;
    u64* xbuf = (u64*)(base + pl.status_bytes);
    unsigned short* w16 = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes);
    float* dgates = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes);
    *dgates_out = dgates;
    clear_work(work, pl.xbuf_bytes, st);
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3(512), dim3(256), 0, st, wcatT, w16, d.Dd + d.E + d.Dd, 4 * d.Dd, pl.R4);
    static int allow = -1, delay = -1;
    if (allow < 0) { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); allow = (e && e[0] == '0') ? 0 : 1; }
    if (delay < 0) { const char* e = getenv("ASR_DEC_BWD_POLL_DELAY"); delay = e ? atoi(e) : 0; }
    PB p{d, w, s, (const unsigned short*)s.enc16, enc_len, dhs, dxin, dq, dkey, slots, dgates, w16, wqT, xbuf, status,
         slot, pl.NT, pl.TE, pl.UPW, pl.CPW, pl.R4, pl.CG2, pl.QG2, pl.VG2, pl.NG2, allow, delay};
    const int cpx = cdiv(d.B, 8), ncw = cdiv(d.A, 64);
    const dim3 grid(8 * cpx * pl.NT), block(64 * (ncw + NPB));
#define DPB_LAUNCH(KN_, TE_)                                                                                                    \
    {                                                                                                                           \
        static unsigned char attr_[32];                                                                                         \
        if (first_on_device(attr_)) hipFuncSetAttribute((const void*)dec_bwd_persist<KN_, TE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); \
        if (!grid_resident(dec_bwd_persist<KN_, TE_>, (int)grid.x, (int)block.x, pl.lds)) return 1;                               \
        hipLaunchKernelGGL((dec_bwd_persist<KN_, TE_>), grid, block, pl.lds, st, p);                                              \
        hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(1), 0, st, status);                                                   \
    }
    if (d.Kn <= 4) { if (pl.TE == 40) DPB_LAUNCH(4, 40) else DPB_LAUNCH(4, 0) }
    else { if (pl.TE == 40) DPB_LAUNCH(10, 40) else DPB_LAUNCH(10, 0) }
#undef DPB_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_att_decoder_bwd(persistent): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}
