// Location-aware attention + LSTM decoder ("speller") of the joint CTC-attention model: the whole
// decode loop of ASR.forward (reference src/asr.py:123-175) and its backward as two C-ABI calls.
//   per step t:  query = tanh(W_q hcat_{t-1} + b_q)                                     (src/asr.py:337)
//                loc   = tanh(W_proj conv1d(prev_att))                                  (src/module.py:1163)
//                e     = w_g . tanh(key + query + loc) + b_g ; /temperature ; mask ; softmax
//                ctx   = attn . enc                                                     (src/module.py:1110-1117,1168)
//                (h,c) = LSTM([emb(prev token), ctx])  per decoder layer                (src/asr.py:141-143,259-266)
// after the loop: logits = W_c h_top + b_c for all steps at once (same arithmetic as the per-step
// char_trans, src/asr.py:265).  Greedy mode (teacher == NULL) also evaluates logits/argmax per step.
//
// Step kernels are bandwidth/latency bound: key (B,T',A) and enc (B,T',E) are re-read every step and
// stay resident in the 256 MB Infinity Cache between steps; one launch covers the whole batch.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int TT = 16;   // encoder frames per workgroup in the energy kernels (4 waves x TPW frames)
constexpr int TPW = TT / 4;

// tanh via one v_exp and one v_rcp: 1 - 2/(1+e^{2x}); absolute error ~2e-7 (the energy kernels evaluate ~6 M of these per step)
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// A batch of U strided global reads per thread whose values are consumed LATER (usually stored to LDS): the loads are
// unconditional from clamped indices, so several batches can be put in flight back to back and cost one round trip in
// all.  (A plain `for (i = tid; i < n; i += nthr) lds[i] = g[i];` costs one serialized round trip per iteration: the
// compiler branches around each bounds-checked load and waits for it.)  Elements beyond U*nthr: `rest`.
template <int U> struct Stage {
    float v[U];
    template <typename L> __device__ __forceinline__ void load(int n, int tid, int nthr, L ld) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld(max(min(tid + u * nthr, n - 1), 0));
    }
    template <typename S> __device__ __forceinline__ void store(int n, int tid, int nthr, S st) const {
#pragma unroll
        for (int u = 0; u < U; ++u) if (tid + u * nthr < n) st(tid + u * nthr, v[u]);
    }
    template <typename L, typename S> __device__ __forceinline__ void rest(int n, int tid, int nthr, L ld, S st) const {
        for (int i = tid + U * nthr; i < n; i += nthr) st(i, ld(i));
    }
};

__device__ __forceinline__ float bf2f(unsigned short x) { return __uint_as_float((unsigned)x << 16); }
// four consecutive bf16 (8 bytes, 8-byte aligned) -> float4
__device__ __forceinline__ float4 ld4_bf16(const unsigned short* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__global__ void cast_bf16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = f2bf_bits(src[i]);
}

// i / n for 0 <= i < 2^22, n <= 2^10 through a float reciprocal (a runtime integer division costs ~40 VALU instructions;
// the index maps of the staging loops do a dozen of them per thread per launch)
__device__ __forceinline__ int fdiv(int i, int n, float inv) {
    int q = (int)((float)i * inv);
    const int r = i - q * n;
    q += (r >= n) - (r < 0);
    return q;
}

struct DecP {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const float* enc;
    const int64_t* enc_len;
};

// ------------------------------------------------------------------------------------------------
// token embedding rows:  xin[b,t,0:Dd] = emb[tokens[b,t]]
// ------------------------------------------------------------------------------------------------
__global__ void shift_tokens_kernel(const int64_t* __restrict__ teacher, int64_t* __restrict__ tokens, int B, int L, int Lt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * L) return;
    const int b = i / L, t = i % L;
    tokens[i] = (t == 0) ? 0 : teacher[(long)b * Lt + (t - 1)];
}

__global__ void embed_kernel(const float* __restrict__ emb, const int64_t* __restrict__ tokens, float* __restrict__ xin,
                             int B, int L, int Dd, int XW, int t0, int nt, int V) {
    // rows (b, t) for t in [t0, t0+nt)
    const long total = (long)B * nt * Dd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Dd);
        const int t = t0 + (int)((i / Dd) % nt);
        const int b = (int)(i / ((long)Dd * nt));
        long tok = tokens[(long)b * L + t];
        tok = tok < 0 ? 0 : (tok >= V ? V - 1 : tok);
        xin[((long)b * L + t) * XW + k] = emb[tok * Dd + k];
    }
}

// ------------------------------------------------------------------------------------------------
// K1: query projection  q[b,t,:] = tanh(W_q hcat_{t-1}[b] + b_q);  one wave per 16 output columns
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void dec_query_kernel(DecP p, int t) {
    __shared__ float red[4][256];
    const asr_dec_dims_t& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int a = blockIdx.x * 16 + n;
    const bool aok = a < d.A;
    const float* wrow = p.w.Wq + (long)(aok ? a : 0) * d.Q;
    const bool vec = (d.Q % 4) == 0;
    for (int m0 = 0; m0 < d.B; m0 += 16) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (t > 0) {
            const int ab = m0 + n;
            const bool rok = ab < d.B;
            const float* hrow = p.s.hs + ((long)(rok ? ab : 0) * d.L + (t - 1)) * d.Q;
            acc = dot_rows<BF16>(hrow, rok, wrow, aok, d.Q, wave, 4, vec, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * 16 + n] = acc[r];
        __syncthreads();
        const int row = tid >> 4, col = tid & 15;
        const int b = m0 + row, ac = blockIdx.x * 16 + col;
        if (b < d.B && ac < d.A)
            p.s.q[((long)b * d.L + t) * d.A + ac] = tanhf(red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] + p.w.bq[ac]);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// shared tile prologue of the energy kernels: previous attention window + location convolution
//   s_conv[k*TT + i] = sum_j Wconv[k][j] * prev_att[tau0 + i + j - Ks]
// prev_att for t == 0 is the uniform initialisation 1/len over valid frames (src/module.py:1157-1160).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void conv_tile(const DecP& p, int b, int t, int tau0, float* s_pa, float* s_wc, float* s_conv) {
    const asr_dec_dims_t& d = p.d;
    const int taps = 2 * d.Ks + 1;
    const int len = (int)p.enc_len[b];
    const int win = TT + 2 * d.Ks;
    const float* prev = (t > 0) ? p.s.att + ((long)b * d.L + (t - 1)) * d.Tp : nullptr;
    for (int i = threadIdx.x; i < win; i += blockDim.x) {
        const int tau = tau0 + i - d.Ks;
        float v = 0.f;
        if (tau >= 0 && tau < d.Tp) v = prev ? prev[tau] : (tau < len ? 1.f / (float)len : 0.f);
        s_pa[i] = v;
    }
    for (int i = threadIdx.x; i < d.Kn * taps; i += blockDim.x) s_wc[i] = p.w.Wconv[i];
    __syncthreads();
    // one (frame, kernel) output per thread
    for (int o = threadIdx.x; o < d.Kn * TT; o += blockDim.x) {
        const int i = o % TT, k = o / TT;
        const float* wk = s_wc + k * taps;
        float a0 = 0.f, a1 = 0.f;
        int j = 0;
        for (; j + 1 < taps; j += 2) { a0 += wk[j] * s_pa[i + j]; a1 += wk[j + 1] * s_pa[i + j + 1]; }
        if (j < taps) a0 += wk[j] * s_pa[i + j];
        s_conv[o] = a0 + a1;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// K2: energies.  grid (ceil(T'/TE), B), TE = 8 waves x TPW frames; wave w owns frames w*TPW.., lanes sweep the
// attention dimension.  TPW is chosen by the host so that the whole batch is ONE round of workgroups (<= 1 per CU):
// the kernel is a chain of dependent round trips (window -> conv -> keys -> reduce), not bandwidth, so a second
// round of workgroups doubles its time.  Every global operand of the sweep is requested before the location
// convolution is computed; the convolution itself is spread over all 512 threads (tap ranges x 4-frame groups).
// conv (B,L,Kn,T') is kept for the backward pass when state.conv != NULL.
// ------------------------------------------------------------------------------------------------
template <int KNMAX, int TPW, bool H16>
__global__ __launch_bounds__(512) void att_energy_kernel(DecP p, int t) {
    constexpr int TE = 8 * TPW, NA = 5;                 // NA: attention columns per lane held in flight
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.y, tau0 = blockIdx.x * TE;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = 2 * d.Ks + 1, win = TE + 2 * d.Ks;
    const int len = min((int)p.enc_len[b], d.Tp);
    float* s_pa = smem_f;                               // [win]            previous attention window
    float* s_wc = s_pa + ((win + 3) & ~3);              // [Kn*taps]        location filters
    float* s_conv = s_wc + ((d.Kn * taps + 3) & ~3);    // [Kn*TE]          conv output of the tile
    float* s_wp = s_conv + d.Kn * TE;                   // [Kn][A]          W_proj transposed
    float* s_part = s_wp + ((d.Kn * d.A + 3) & ~3);     // [parts][Kn*TE]   partial conv sums
    // ---- stage 0: every global read that does not depend on the convolution, all in flight together
    const float* prev = (t > 0) ? p.s.att + ((long)b * d.L + (t - 1)) * d.Tp : nullptr;
    const float uni = 1.f / (float)max(len, 1);
    auto ld_pa = [&](int i) { return prev[min(max(tau0 + i - d.Ks, 0), d.Tp - 1)]; };
    auto st_pa = [&](int i, float v) {
        const int tau = tau0 + i - d.Ks;
        s_pa[i] = (tau >= 0 && tau < d.Tp) ? (prev ? v : (tau < len ? uni : 0.f)) : 0.f;
    };
    auto ld_wc = [&](int i) { return p.w.Wconv[i]; };
    auto st_wc = [&](int i, float v) { s_wc[i] = v; };
    auto ld_wp = [&](int i) { return p.w.Wproj[i]; };
    const float inv_kn = 1.f / (float)d.Kn;
    auto st_wp = [&](int i, float v) { const int a = fdiv(i, d.Kn, inv_kn), k = i - a * d.Kn; s_wp[k * d.A + a] = v; };
    Stage<2> g_pa; Stage<4> g_wc; Stage<6> g_wp;
    g_pa.v[0] = g_pa.v[1] = 0.f;
    if (prev) g_pa.load(win, tid, 512, ld_pa);
    g_wc.load(d.Kn * taps, tid, 512, ld_wc);
    g_wp.load(d.Kn * d.A, tid, 512, ld_wp);
    const float* qrow = p.s.q + ((long)b * d.L + t) * d.A;
    const int tmax = max(len - 1, 0);
    float kv[NA][TPW], qa[NA], wga[NA];
    auto load_cols = [&](int a0) {
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int a = min(a0 + 64 * c + lane, d.A - 1);
            qa[c] = qrow[a]; wga[c] = p.w.wg[a];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const long ki = ((long)b * d.Tp + min(tau0 + wave * TPW + i, tmax)) * d.A + a;
                kv[c][i] = H16 ? bf2f(reinterpret_cast<const unsigned short*>(p.s.key16)[ki]) : p.s.key[ki];
            }
        }
    };
    load_cols(0);
    g_pa.store(win, tid, 512, st_pa);
    g_wc.store(d.Kn * taps, tid, 512, st_wc);
    g_wp.store(d.Kn * d.A, tid, 512, st_wp);
    g_pa.rest(win, tid, 512, [&](int i) { return prev ? ld_pa(i) : 0.f; }, st_pa);
    g_wc.rest(d.Kn * taps, tid, 512, ld_wc, st_wc);
    g_wp.rest(d.Kn * d.A, tid, 512, ld_wp, st_wp);
    __syncthreads();
    // ---- stage 1: conv[k][i] = sum_j Wconv[k][j] * pa[i + j];  item = (tap range, k, group of 4 frames)
    {
        const int ngrp = TE / 4, nout = d.Kn * ngrp;
        const int parts = max(1, min(8, 512 / nout));
        const int tp = (taps + parts - 1) / parts;
        const float inv_nout = 1.f / (float)nout;
        for (int it = tid; it < parts * nout; it += 512) {
            const int pz = fdiv(it, nout, inv_nout), o = it - pz * nout, k = o / ngrp, ig = o - k * ngrp;   // ngrp is a constant
            const int j0 = pz * tp, j1 = min(taps, j0 + tp);
            const float* wk = s_wc + k * taps;
            const float* pa = s_pa + 4 * ig;
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            float p0 = pa[j0], p1 = pa[j0 + 1], p2 = pa[j0 + 2];
            for (int j = j0; j < j1; ++j) {
                const float w = wk[j], p3 = pa[j + 3];
                a0 += w * p0; a1 += w * p1; a2 += w * p2; a3 += w * p3;
                p0 = p1; p1 = p2; p2 = p3;
            }
            float* o4 = s_part + (long)pz * d.Kn * TE + k * TE + 4 * ig;
            o4[0] = a0; o4[1] = a1; o4[2] = a2; o4[3] = a3;
        }
        __syncthreads();
        for (int o = tid; o < d.Kn * TE; o += 512) {
            float v = 0.f;
            for (int pz = 0; pz < parts; ++pz) v += s_part[(long)pz * d.Kn * TE + o];
            s_conv[o] = v;
            const int k = o / TE, tau = tau0 + (o - k * TE);
            if (p.s.conv && tau < d.Tp) p.s.conv[(((long)b * d.L + t) * d.Kn + k) * d.Tp + tau] = v;
        }
        __syncthreads();
    }
    // ---- stage 2: sweep.  The wave's conv values are the same for every lane: keep them in registers.
    float cv[KNMAX][TPW];
#pragma unroll
    for (int k = 0; k < KNMAX; ++k)
#pragma unroll
        for (int i = 0; i < TPW; ++i) cv[k][i] = (k < d.Kn) ? s_conv[k * TE + wave * TPW + i] : 0.f;
    float e[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) e[i] = 0.f;
    for (int a0 = 0; a0 < d.A; a0 += 64 * NA) {
        if (a0 > 0) load_cols(a0);
#pragma unroll
        for (int c = 0; c < NA; ++c) {
            const int a = a0 + 64 * c + lane;
            const bool aok = a < d.A;
            const int ac = aok ? a : d.A - 1;
            float wp[KNMAX];
#pragma unroll
            for (int k = 0; k < KNMAX; ++k) wp[k] = (k < d.Kn) ? s_wp[k * d.A + ac] : 0.f;
            const float wg_ = aok ? wga[c] : 0.f;
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                float lp = 0.f;
#pragma unroll
                for (int k = 0; k < KNMAX; ++k) lp += wp[k] * cv[k][i];
                e[i] += wg_ * tanh_fast(kv[c][i] + qa[c] + tanh_fast(lp));
            }
        }
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const float s = wave_sum(e[i]);
        const int tau = tau0 + wave * TPW + i;
        if (lane == 0 && tau < d.Tp)
            p.s.energy[(long)b * d.Tp + tau] = (tau < len) ? (s + p.w.bg[0]) / d.temperature : -INFINITY;
    }
}

// block-wide reductions over 256 threads (4 waves)
__device__ __forceinline__ float block_max(float v, float* s4) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s4[3]));
}
__device__ __forceinline__ float block_sum(float v, float* s4) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = v;
    __syncthreads();
    return s4[0] + s4[1] + s4[2] + s4[3];
}

// ------------------------------------------------------------------------------------------------
// K3: softmax over T' + context.  grid (ceil(E/64), B); every block redoes the (cheap) softmax of its
// utterance and owns 64 columns of the context vector; block x == 0 also stores the attention row.
// ------------------------------------------------------------------------------------------------
template <bool H16>
__global__ __launch_bounds__(256) void att_softmax_ctx_kernel(DecP p, int t) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    __shared__ float s4[4];
    __shared__ float red[4][64];
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.y, e0 = blockIdx.x * 64;
    const int len = min((int)p.enc_len[b], d.Tp);
    float* s_att = smem_f;  // [Tp]
    const float* en = p.s.energy + (long)b * d.Tp;
    float m = -INFINITY;
    {
        Stage<4> g_en;                     // the whole energy row in one round trip (T' <= 1024; the rest in a plain loop)
        g_en.load(d.Tp, threadIdx.x, 256, [&](int i) { return en[i]; });
        g_en.store(d.Tp, threadIdx.x, 256, [&](int i, float v) { s_att[i] = v; m = fmaxf(m, v); });
        g_en.rest(d.Tp, threadIdx.x, 256, [&](int i) { return en[i]; }, [&](int i, float v) { s_att[i] = v; m = fmaxf(m, v); });
    }
    m = block_max(m, s4);
    float sum = 0.f;
    for (int i = threadIdx.x; i < d.Tp; i += 256) { const float v = expf(s_att[i] - m); s_att[i] = v; sum += v; }
    sum = block_sum(sum, s4);
    const float inv = 1.f / sum;
    for (int i = threadIdx.x; i < d.Tp; i += 256) {
        const float a = s_att[i] * inv;
        s_att[i] = a;
        if (blockIdx.x == 0) p.s.att[((long)b * d.L + t) * d.Tp + i] = a;
    }
    __syncthreads();
    float* xrow = p.s.xin + ((long)b * d.L + t) * (d.Dd + d.E) + d.Dd;
    if ((d.E & 3) == 0) {
        // 16 lanes x float4 cover the block's 64 columns; 16 frame groups, 4 independent loads in flight each
        const int c4 = threadIdx.x & 15, grp = threadIdx.x >> 4;
        const int ecol = e0 + 4 * c4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ecol < d.E) {
            const float* ep = p.enc + (long)b * d.Tp * d.E + ecol;
            const unsigned short* ep16 = reinterpret_cast<const unsigned short*>(p.s.enc16) + (long)b * d.Tp * d.E + ecol;
            // 16 frame groups x 16 float4 columns; each thread walks frames grp, grp+16, ... ten at a time in flight
            // (clamped addresses, weights of frames >= len are read as 0 from the padded attention row)
            for (int tau = grp; tau < len; tau += 160) {
                float4 v[10];
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const long off = (long)min(tau + 16 * u, len - 1) * d.E;
                    v[u] = H16 ? ld4_bf16(ep16 + off) : *reinterpret_cast<const float4*>(ep + off);
                }
#pragma unroll
                for (int u = 0; u < 10; ++u) {
                    const float a = (tau + 16 * u < len) ? s_att[tau + 16 * u] : 0.f;
                    acc.x += a * v[u].x; acc.y += a * v[u].y; acc.z += a * v[u].z; acc.w += a * v[u].w;
                }
            }
        }
        __shared__ float4 red4[16][16];
        red4[grp][c4] = acc;
        __syncthreads();
        if (threadIdx.x < 16 && e0 + 4 * threadIdx.x < d.E) {
            float4 r = red4[0][threadIdx.x];
#pragma unroll
            for (int g = 1; g < 16; ++g) { const float4 v = red4[g][threadIdx.x]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
            float* o = xrow + e0 + 4 * threadIdx.x;
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
        }
    } else {
        const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
        const int ecol = e0 + col;
        float acc = 0.f;
        if (ecol < d.E) {
            const float* ep = p.enc + (long)b * d.Tp * d.E + ecol;
            for (int tau = grp; tau < len; tau += 4) acc += s_att[tau] * ep[(long)tau * d.E];
        }
        red[grp][col] = acc;
        __syncthreads();
        if (grp == 0 && ecol < d.E) xrow[ecol] = red[0][col] + red[1][col] + red[2][col] + red[3][col];
    }
}

// ------------------------------------------------------------------------------------------------
// K4: one decoder LSTM layer, one step.  Same wave layout as the encoder step: column n of the tile is
// gate n>>2 of hidden unit 4*blockIdx.x + (n&3).
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void dec_cell_fwd_kernel(DecP p, int t, int l) {
    // block = 4 hidden units x 4 gates (16 columns); the K = Kx + Dd reduction is split over the 4 waves
    // (interleaved k-steps) and summed through LDS, so one memory round trip per wave covers ~K/4.
    __shared__ float red[4][256];
    const asr_dec_dims_t& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int g = n >> 2;
    const int j = blockIdx.x * 4 + (n & 3);
    const bool jok = j < d.Dd;
    const int XW = d.Dd + d.E;
    const int Kx = (l == 0) ? XW : d.Dd;
    const float* wih = p.w.Wih[l] + ((long)g * d.Dd + (jok ? j : 0)) * Kx;
    const float* whh = p.w.Whh[l] + ((long)g * d.Dd + (jok ? j : 0)) * d.Dd;
    const long SW = (long)d.NL * d.Dd;  // row width of hs/cs
    for (int m0 = 0; m0 < d.B; m0 += 16) {
        const int ab = m0 + n;
        const bool rok = ab < d.B;
        const long rowi = (long)(rok ? ab : 0) * d.L + t;
        const float* xrow = (l == 0) ? p.s.xin + rowi * XW : p.s.hs + rowi * SW + (long)(l - 1) * d.Dd;
        const float* hrow = p.s.hs + (rowi - (t > 0 ? 1 : 0)) * SW + (long)l * d.Dd;
        // the epilogue's operands (biases, previous cell state) are requested together with the contraction's
        const int erow = tid >> 2, ejj = tid & 3;
        const int eb = m0 + erow, eju = blockIdx.x * 4 + ejj;
        const bool eok = tid < 64 && eb < d.B && eju < d.Dd;
        const long eri = (long)(eok ? eb : 0) * d.L + t;
        float bsum[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
        if (eok) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) bsum[gg] = p.w.bih[l][gg * d.Dd + eju] + p.w.bhh[l][gg * d.Dd + eju];
            if (t > 0) cp = p.s.cs[(eri - 1) * SW + (long)l * d.Dd + eju];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = dot_rows_cat<BF16, 10>(xrow, wih, Kx, (Kx % 4) == 0 && (d.Dd % 4) == 0, hrow, whh, (t > 0) ? d.Dd : 0, (d.Dd % 4) == 0,
                                     rok, jok, wave, 4, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * 16 + n] = acc[r];
        __syncthreads();
        if (eok) {
            // thread (row, unit): gathers its four gate columns
            float pre[4];
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const int c = erow * 16 + gg * 4 + ejj;
                pre[gg] = red[0][c] + red[1][c] + red[2][c] + red[3][c] + bsum[gg];
            }
            const float ai = sigmoidf_(pre[0]), af = sigmoidf_(pre[1]), ag = tanhf(pre[2]), ao = sigmoidf_(pre[3]);
            float* go = p.s.gates + (eri * d.NL + l) * 4 * d.Dd;
            go[eju] = ai; go[d.Dd + eju] = af; go[2 * d.Dd + eju] = ag; go[3 * d.Dd + eju] = ao;
            const float cn = af * cp + ai * ag;
            p.s.cs[eri * SW + (long)l * d.Dd + eju] = cn;
            p.s.hs[eri * SW + (long)l * d.Dd + eju] = ao * tanhf(cn);
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// K5 (greedy decoding only): logits of step t, argmax, next input token.  One block per utterance.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_greedy_kernel(DecP p, int t) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];  // [V]
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* h = p.s.hs + (((long)b * d.L + t) * d.NL + (d.NL - 1)) * d.Dd;
    for (int v = wave; v < d.V; v += 4) {
        float acc = 0.f;
        for (int k = lane; k < d.Dd; k += 64) acc += h[k] * p.w.Wc[(long)v * d.Dd + k];
        acc = wave_sum(acc);
        if (lane == 0) {
            acc += p.w.bc[v];
            smem_f[v] = acc;
            p.s.logits[((long)b * d.L + t) * d.V + v] = acc;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && t + 1 < d.L) {
        int best = 0;
        float bv = smem_f[0];
        for (int v = 1; v < d.V; ++v) if (smem_f[v] > bv) { bv = smem_f[v]; best = v; }
        p.s.tokens[(long)b * d.L + t + 1] = best;
    }
}

__global__ __launch_bounds__(256) void dec_logits_kernel(DecP p, int t) {
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* h = p.s.hs + (((long)b * d.L + t) * d.NL + (d.NL - 1)) * d.Dd;
    for (int v = wave; v < d.V; v += 4) {
        float acc = 0.f;
        for (int k = lane; k < d.Dd; k += 64) acc += h[k] * p.w.Wc[(long)v * d.Dd + k];
        acc = wave_sum(acc);
        if (lane == 0) p.s.logits[((long)b * d.L + t) * d.V + v] = acc + p.w.bc[v];
    }
}

// =================================================================================================
// backward
// =================================================================================================
struct DecB {
    DecP f;
    asr_dec_grads_t g;
    float* dhs;        // (B,L,NL,Dd)   gradient wrt every h (accumulated in place)
    float* dxin;       // (B,L,Dd+E)
    float* dq;         // (B,L,A)       gradient wrt the query pre-activation (per-tile partials added atomically)
    float* dkey;       // (B,T',A)
    float* datt_next;  // (B,T')        gradient flowing into attn_t from step t+1's location conv
    float* dcf;        // (NL,B,Dd)
    float* wcatT[ASR_MAX_DEC_LAYERS];  // ((Kx+Dd) x 4Dd) transposed [W_ih ; W_hh]
    float* wqT;        // (Q x A)
    float* slots;      // (B*nte, slot) per-workgroup partial sums of d w_g, d W_proj ([k][a]), d b_g
    int nte, slot;     // energy-backward tiles per utterance, floats per slot
};

// cell backward, elementwise part: dgates (in place over the activated gates) and the dc*f carry
__global__ void dec_cell_bwd_elem_kernel(DecB p, int t, int l, int last) {
    const asr_dec_dims_t& d = p.f.d;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.B * d.Dd) return;
    const int b = i / d.Dd, j = i % d.Dd;
    const long ri = (long)b * d.L + t;
    const long SW = (long)d.NL * d.Dd;
    float* g = p.f.s.gates + (ri * d.NL + l) * 4 * d.Dd;
    const float gi = g[j], gf = g[d.Dd + j], gg = g[2 * d.Dd + j], go = g[3 * d.Dd + j];
    const float ct = p.f.s.cs[ri * SW + (long)l * d.Dd + j];
    const float cp = (t > 0) ? p.f.s.cs[(ri - 1) * SW + (long)l * d.Dd + j] : 0.f;
    const float dh = p.dhs[ri * SW + (long)l * d.Dd + j];
    const long ci = ((long)l * d.B + b) * d.Dd + j;
    const float carry = last ? 0.f : p.dcf[ci];
    const float tc = tanhf(ct);
    const float dc = dh * go * (1.f - tc * tc) + carry;
    g[j] = dc * gg * gi * (1.f - gi);
    g[d.Dd + j] = dc * cp * gf * (1.f - gf);
    g[2 * d.Dd + j] = dc * gi * (1.f - gg * gg);
    g[3 * d.Dd + j] = dh * tc * go * (1.f - go);
    p.dcf[ci] = dc * gf;
}

// cell backward, contraction part: [d_input | d_h_prev] = dgates (B x 4Dd) * [W_ih | W_hh]
// block = 4 waves (split over the 4Dd reduction) x 16 output columns of the concatenated width Kx+Dd.
template <bool BF16>
__global__ __launch_bounds__(256) void dec_cell_bwd_mm_kernel(DecB p, int t, int l) {
    __shared__ float red[4][256];
    const asr_dec_dims_t& d = p.f.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int XW = d.Dd + d.E;
    const int Kx = (l == 0) ? XW : d.Dd;
    const int K = 4 * d.Dd;
    const int c0 = blockIdx.x * 16;
    const bool cok = (c0 + n) < Kx + d.Dd;
    const float* wrow = p.wcatT[l] + (long)(cok ? c0 + n : 0) * K;
    const long SW = (long)d.NL * d.Dd;
    for (int m0 = 0; m0 < d.B; m0 += 16) {
        const int ab = m0 + n;
        const bool rok = ab < d.B;
        const float* grow = p.f.s.gates + ((((long)(rok ? ab : 0) * d.L + t) * d.NL) + l) * K;
        // accumulate targets are read before the contraction (one round trip for everything)
        const int row = tid >> 4, col = tid & 15;
        const int b = m0 + row, c = c0 + col;
        const bool ook = b < d.B && c < Kx + d.Dd;
        const long ri = (long)(ook ? b : 0) * d.L + t;
        float* dst = nullptr;
        bool add = false;
        if (ook) {
            if (c < Kx) {
                if (l == 0) dst = p.dxin + ri * XW + c;
                else { dst = p.dhs + ri * SW + (long)(l - 1) * d.Dd + c; add = true; }
            } else if (t > 0) {
                dst = p.dhs + (ri - 1) * SW + (long)l * d.Dd + (c - Kx); add = true;
            }
        }
        const float oldv = (dst && add) ? *dst : 0.f;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = dot_rows<BF16, 10>(grow, rok, wrow, cok, K, wave, 4, true, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * 16 + n] = acc[r];
        __syncthreads();
        if (dst) *dst = oldv + red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        __syncthreads();
    }
}

// B2: attention backward of one (utterance, TE-frame tile) in ONE kernel:
//   dattn[tau] = dctx . enc[tau] + datt_next[tau]                      (context + next step's location path)
//   softmax backward: de = attn * (dattn - dot) / temperature, with dot = sum_tau attn*dattn over the WHOLE utterance
//     = dctx . ctx + sum_tau attn*datt_next   (ctx = sum attn*enc is the saved context), so no workgroup needs another
//     tile's dattn and the former separate dattn kernel + buffer are gone;
//   energy backward: recompute loc/u from the saved conv, accumulate dkey (RMW), per-workgroup partial sums of
//     d w_g / d W_proj / d b_g in private slots, d query (x (1-q^2), atomically into dq[b,t,:]) and
//     dconv[tau,k] = sum_a dl[tau,a] W_proj[a,k], written over the saved conv of this step.
// Threads: NG frame groups x ceil(A/64) waves; wave (g, wa) owns attention dims a = 64*wa + lane for the frames of
// group g, so the per-a sums have one writer per group and meet in LDS.  TE is chosen by the host so that the batch is
// one round of workgroups; all global operands are requested before the first barrier.
template <int KNMAX, bool H16>
__global__ __launch_bounds__(640) void att_bwd_energy_kernel(DecB p, int t, int TE, int NG, int last) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    __shared__ float s_red[16];
    constexpr int KP = (KNMAX + 3) & ~3, CHK = 10;
    const asr_dec_dims_t& d = p.f.d;
    const int b = blockIdx.y, tau0 = blockIdx.x * TE;
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int nwa = (d.A + 63) >> 6, g = wave / nwa, wa = wave - g * nwa;
    const int FG = TE / NG;                              // frames per group
    int ap4 = (d.A + 3) >> 2; if ((ap4 & 1) == 0) ++ap4;
    const int AP = 4 * ap4;                              // row stride of the [frame][a] / [k][a] tiles: 16B rows, odd in float4 units
    const int XW = d.Dd + d.E;
    float* s_cv = smem_f;                                // [TE][KP]   conv of the tile
    float* s_de = s_cv + TE * KP;                        // [TE]
    float* s_dat = s_de + TE;                            // [TE]
    float* s_dctx = s_dat + TE;                          // [E]
    float* s_wpT = s_dctx + ((d.E + 3) & ~3);            // [Kn][AP]
    float* s_dl = s_wpT + d.Kn * AP;                     // [TE][AP]
    float* s_acc = s_dl + TE * AP;                       // [NG][Kn+2][A]
    const int len = min((int)p.f.enc_len[b], d.Tp);
    const int tmax = max(len - 1, 0);
    const long row = (long)b * d.L + t;
    float* convrow = p.f.s.conv + row * d.Kn * d.Tp;     // conv in, dconv out
    const float* att = p.f.s.att + row * d.Tp;
    const float* dnext = p.datt_next + (long)b * d.Tp;

    // ---- stage 0: all global reads, in flight together (consumed after the key/dkey chunk has been requested too)
    auto ld_dc = [&](int e) { return p.dxin[row * XW + d.Dd + e]; };
    auto ld_cx = [&](int e) { return p.f.s.xin[row * XW + d.Dd + e]; };
    auto ld_at = [&](int i) { return att[i]; };
    auto ld_dn = [&](int i) { return dnext[i]; };
    const float inv_te = 1.f / (float)TE, inv_kn = 1.f / (float)d.Kn;
    auto ld_cv = [&](int i) { const int k = fdiv(i, TE, inv_te), ti = i - k * TE; return convrow[(long)k * d.Tp + min(tau0 + ti, d.Tp - 1)]; };
    auto st_cv = [&](int i, float v) { const int k = fdiv(i, TE, inv_te), ti = i - k * TE; s_cv[ti * KP + k] = (tau0 + ti < d.Tp) ? v : 0.f; };
    auto ld_wp = [&](int i) { return p.f.w.Wproj[i]; };
    auto st_wp = [&](int i, float v) { const int a_ = fdiv(i, d.Kn, inv_kn), k = i - a_ * d.Kn; s_wpT[k * AP + a_] = v; };
    Stage<2> g_dc, g_cx, g_at, g_dn, g_cv; Stage<6> g_wp;
    g_dc.load(d.E, tid, nthr, ld_dc);
    g_cx.load(d.E, tid, nthr, ld_cx);
    g_at.load(max(len, 1), tid, nthr, ld_at);
    g_dn.load(max(len, 1), tid, nthr, ld_dn);
    g_cv.load(d.Kn * TE, tid, nthr, ld_cv);
    g_wp.load(d.Kn * d.A, tid, nthr, ld_wp);
    const int a = 64 * wa + lane;
    const bool aok = a < d.A;
    const int ac = aok ? a : d.A - 1;
    const float qa = p.f.s.q[row * d.A + ac], wga = p.f.w.wg[ac];
    float* slot = p.slots + ((long)b * p.nte + blockIdx.x) * p.slot;
    float old[KNMAX + 1];
    if (g == 0) {
        old[KNMAX] = slot[ac];
#pragma unroll
        for (int k = 0; k < KNMAX; ++k) old[k] = (k < d.Kn) ? slot[d.A + k * d.A + ac] : 0.f;
    }
    float kv[CHK], dk[CHK];
    const int f_beg = g * FG, f_end = f_beg + FG;
    auto load_chunk = [&](int f0) {
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
            const long ki = ((long)b * d.Tp + min(tau0 + f0 + i, tmax)) * d.A + ac;
            kv[i] = H16 ? bf2f(reinterpret_cast<const unsigned short*>(p.f.s.key16)[ki]) : p.f.s.key[ki];
            dk[i] = p.dkey[ki];
        }
    };
    load_chunk(f_beg);
    float dot = 0.f;
    {
        g_dc.store(d.E, tid, nthr, [&](int e, float v) { s_dctx[e] = v; });
#pragma unroll
        for (int u = 0; u < 2; ++u) if (tid + u * nthr < d.E) dot += g_dc.v[u] * g_cx.v[u];
        for (int e = tid + 2 * nthr; e < d.E; e += nthr) { const float dc = ld_dc(e); s_dctx[e] = dc; dot += dc * ld_cx(e); }
        if (!last) {
#pragma unroll
            for (int u = 0; u < 2; ++u) if (tid + u * nthr < len) dot += g_at.v[u] * g_dn.v[u];
            for (int i = tid + 2 * nthr; i < len; i += nthr) dot += att[i] * dnext[i];
        }
        g_cv.store(d.Kn * TE, tid, nthr, st_cv);
        g_cv.rest(d.Kn * TE, tid, nthr, ld_cv, st_cv);
        g_wp.store(d.Kn * d.A, tid, nthr, st_wp);
        g_wp.rest(d.Kn * d.A, tid, nthr, ld_wp, st_wp);
        for (int i = tid; i < d.Kn * (AP - d.A); i += nthr) { const int k = i / (AP - d.A); s_wpT[k * AP + d.A + (i - k * (AP - d.A))] = 0.f; }
    }
    __syncthreads();
    // ---- stage 1: dattn of the tile (16 lanes per frame, float4 over E), block-wide dot, de
    {
        const int part = tid & 15;
        const bool vec = (d.E & 3) == 0;
        for (int f = tid >> 4; f < TE; f += nthr >> 4) {
            const int tau = tau0 + f;
            const float* er = p.f.enc + ((long)b * d.Tp + min(tau, tmax)) * d.E;
            const unsigned short* er16 = reinterpret_cast<const unsigned short*>(p.f.s.enc16) + ((long)b * d.Tp + min(tau, tmax)) * d.E;
            float v = 0.f;
            if (vec) {
                // E/64 float4 per lane, ten at a time in flight (clamped address, contribution zeroed by the LDS operand)
                for (int e0 = 4 * part; e0 < d.E; e0 += 640) {
                    float4 x[10];
#pragma unroll
                    for (int u = 0; u < 10; ++u) x[u] = H16 ? ld4_bf16(er16 + min(e0 + 64 * u, d.E - 4)) : *reinterpret_cast<const float4*>(er + min(e0 + 64 * u, d.E - 4));
#pragma unroll
                    for (int u = 0; u < 10; ++u) {
                        if (e0 + 64 * u < d.E) {
                            const float4 c = *reinterpret_cast<const float4*>(s_dctx + e0 + 64 * u);
                            v += x[u].x * c.x + x[u].y * c.y + x[u].z * c.z + x[u].w * c.w;
                        }
                    }
                }
            } else {
                for (int e = part; e < d.E; e += 16) v += er[e] * s_dctx[e];
            }
            v += __shfl_xor(v, 8); v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
            if (part == 0) s_dat[f] = (tau < len) ? v + (last ? 0.f : dnext[tau]) : 0.f;
        }
        dot = wave_sum(dot);
        if (lane == 0) s_red[wave] = dot;
    }
    __syncthreads();
    dot = 0.f;
    for (int w = 0; w < (nthr >> 6); ++w) dot += s_red[w];
    for (int f = tid; f < TE; f += nthr) {
        const int tau = tau0 + f;
        s_de[f] = (tau < len) ? att[tau] * (s_dat[f] - dot) / d.temperature : 0.f;
    }
    __syncthreads();
    // ---- stage 2: sweep over the group's frames
    float wp[KNMAX], dwp[KNMAX];
#pragma unroll
    for (int k = 0; k < KNMAX; ++k) { wp[k] = (k < d.Kn) ? s_wpT[k * AP + ac] : 0.f; dwp[k] = 0.f; }
    float dwg = 0.f, dqa = 0.f;
    for (int f0 = f_beg; f0 < f_end; f0 += CHK) {
        float kc[CHK], dc_[CHK];
#pragma unroll
        for (int i = 0; i < CHK; ++i) { kc[i] = kv[i]; dc_[i] = dk[i]; }
        if (f0 + CHK < f_end) load_chunk(f0 + CHK);          // next chunk in flight during this one
#pragma unroll
        for (int i = 0; i < CHK; ++i) {
            const int f = f0 + i, tau = tau0 + f;
            if (f < f_end) {
                float cvv[KP];
#pragma unroll
                for (int k4 = 0; k4 < KP; k4 += 4) {
                    const float4 c4 = *reinterpret_cast<const float4*>(s_cv + f * KP + k4);
                    cvv[k4] = c4.x; cvv[k4 + 1] = c4.y; cvv[k4 + 2] = c4.z; cvv[k4 + 3] = c4.w;
                }
                float lp = 0.f;
#pragma unroll
                for (int k = 0; k < KNMAX; ++k) lp += wp[k] * cvv[k];
                const float loc = tanh_fast(lp);
                const float u = tanh_fast(kc[i] + qa + loc);
                const float de = s_de[f];                        // 0 for tau >= len
                const float du = de * wga * (1.f - u * u);
                const float dl = du * (1.f - loc * loc);
                dwg += de * u;
                dqa += du;
#pragma unroll
                for (int k = 0; k < KNMAX; ++k) dwp[k] += dl * cvv[k];
                if (aok && tau < len) p.dkey[((long)b * d.Tp + tau) * d.A + a] = dc_[i] + du;
                if (aok) s_dl[f * AP + a] = dl;
            }
        }
    }
    if (wa == nwa - 1) {                                     // zero the pad columns of the dl tile (read as float4 below)
        for (int f = f_beg; f < f_end; ++f)
            for (int x = d.A + lane; x < AP; x += 64) s_dl[f * AP + x] = 0.f;
    }
    if (NG > 1 && aok) {
        float* acc = s_acc + (long)g * (d.Kn + 2) * d.A;
        acc[a] = dwg; acc[d.A + a] = dqa;
#pragma unroll
        for (int k = 0; k < KNMAX; ++k) if (k < d.Kn) acc[(2 + k) * d.A + a] = dwp[k];
    }
    __syncthreads();
    if (g == 0 && aok) {
        for (int gg = 1; gg < NG; ++gg) {
            const float* acc = s_acc + (long)gg * (d.Kn + 2) * d.A;
            dwg += acc[a]; dqa += acc[d.A + a];
#pragma unroll
            for (int k = 0; k < KNMAX; ++k) if (k < d.Kn) dwp[k] += acc[(2 + k) * d.A + a];
        }
        slot[a] = old[KNMAX] + dwg;
#pragma unroll
        for (int k = 0; k < KNMAX; ++k) if (k < d.Kn) slot[d.A + k * d.A + a] = old[k] + dwp[k];
        atomicAdd(&p.dq[row * d.A + a], dqa * (1.f - qa * qa));
    }
    if (tid == 0) {
        float sde = 0.f;
        for (int f = 0; f < TE; ++f) sde += s_de[f];
        slot[d.A * (1 + d.Kn)] += sde;
    }
    // ---- stage 3: dconv[tau,k] = sum_a dl[tau,a] * W_proj[a,k]  (thread per (frame, k), float4 over a)
    for (int o = tid; o < TE * d.Kn; o += nthr) {
        const int k = fdiv(o, TE, inv_te), f = o - k * TE;
        const float4* dl4 = reinterpret_cast<const float4*>(s_dl + f * AP);
        const float4* w4 = reinterpret_cast<const float4*>(s_wpT + k * AP);
        float v0 = 0.f, v1 = 0.f;
        int x = 0;
        for (; x + 1 < ap4; x += 2) {
            const float4 d0 = dl4[x], w0 = w4[x], d1 = dl4[x + 1], w1 = w4[x + 1];
            v0 += d0.x * w0.x + d0.y * w0.y + d0.z * w0.z + d0.w * w0.w;
            v1 += d1.x * w1.x + d1.y * w1.y + d1.z * w1.z + d1.w * w1.w;
        }
        if (x < ap4) { const float4 d0 = dl4[x], w0 = w4[x]; v0 += d0.x * w0.x + d0.y * w0.y + d0.z * w0.z + d0.w * w0.w; }
        const int tau = tau0 + f;
        if (tau < d.Tp) convrow[(long)k * d.Tp + tau] = (tau < len) ? v0 + v1 : 0.f;
    }
}

// B3: gradient through the location convolution wrt attn_{t-1}:
//   datt_next[tau'] = sum_k sum_j W[k][j] * dconv[k][tau' - j + Ks]      (dconv = this step's rows of state.conv)
// grid (ceil(T'/TC), B), 512 threads; item = (tap range, k, group of 4 outputs) with a sliding register window.
// (d W_conv needs no place in the sequential chain: wconv_grad_kernel after the loop.)
__device__ __forceinline__ void att_bwd_conv_body(const DecB& p, int t, int TC, int bx, int by) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const asr_dec_dims_t& d = p.f.d;
    const int b = by, tau0 = bx * TC, tid = threadIdx.x;
    const int taps = 2 * d.Ks + 1, win = TC + 2 * d.Ks;
    float* s_wc = smem_f;                                // [Kn*taps]
    float* s_dc = s_wc + ((d.Kn * taps + 3) & ~3);       // [Kn][win+4]  dconv window tau0-Ks .. tau0+TC+Ks
    const int winp = win + 4;
    float* s_part = s_dc + d.Kn * winp;                  // [parts*Kn][TC]
    const float* dconv = p.f.s.conv + ((long)b * d.L + t) * d.Kn * d.Tp;
    for (int i = tid; i < d.Kn * taps; i += 512) s_wc[i] = p.f.w.Wconv[i];
    for (int i = tid; i < d.Kn * winp; i += 512) {
        const int k = i / winp, x = i - k * winp, tau = tau0 + x - d.Ks;
        s_dc[i] = (x < win && tau >= 0 && tau < d.Tp) ? dconv[(long)k * d.Tp + tau] : 0.f;
    }
    __syncthreads();
    const int ngrp = TC / 4, nout = d.Kn * ngrp;
    const int parts = max(1, min(8, 512 / nout));
    const int tp = (taps + parts - 1) / parts;
    for (int it = tid; it < parts * nout; it += 512) {
        const int pz = it / nout, o = it - pz * nout, k = o / ngrp, ig = o - k * ngrp;
        const int j0 = pz * tp, j1 = min(taps, j0 + tp);
        const float* wk = s_wc + k * taps;
        const float* dc = s_dc + k * winp + 4 * ig + 2 * d.Ks;      // dc[i - j] for output i of the group
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        float p1 = 0.f, p2 = 0.f, p3 = 0.f;
        if (j0 < j1) { p1 = dc[1 - j0]; p2 = dc[2 - j0]; p3 = dc[3 - j0]; }
        for (int j = j0; j < j1; ++j) {
            const float w = wk[j], p0 = dc[-j];
            a0 += w * p0; a1 += w * p1; a2 += w * p2; a3 += w * p3;
            p3 = p2; p2 = p1; p1 = p0;
        }
        float* o4 = s_part + (long)(pz * d.Kn + k) * TC + 4 * ig;
        o4[0] = a0; o4[1] = a1; o4[2] = a2; o4[3] = a3;
    }
    __syncthreads();
    for (int i = tid; i < TC; i += 512) {
        float v = 0.f;
        for (int r = 0; r < parts * d.Kn; ++r) v += s_part[(long)r * TC + i];
        if (tau0 + i < d.Tp) p.datt_next[(long)b * d.Tp + tau0 + i] = v;
    }
}

// d W_conv[k][j] += sum_{l,b} sum_tau dconv_l[b,k,tau] * prev_att_l[b, tau + j - Ks]   after the loop, all steps at once.
// grid (NCH, B): workgroup (c, b) walks the steps l = c, c+NCH, ...; thread (k, group of 4 taps) slides over tau.
// Partial sums per workgroup land in `out` (gridDim.x*gridDim.y, Kn*taps) and are reduced by slot_reduce_kernel.
__global__ __launch_bounds__(512) void wconv_grad_kernel(DecP p, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.y, tid = threadIdx.x;
    const int taps = 2 * d.Ks + 1;
    const int tg = (taps + 3) / 4, nitem = d.Kn * tg;
    const int len = min((int)p.enc_len[b], d.Tp);
    const int wpa = d.Tp + 2 * d.Ks + 8;
    float* s_pa = smem_f;                                // [Tp + 2Ks + 8]  zero-padded previous attention
    float* s_dc = s_pa + ((wpa + 3) & ~3);               // [Kn][Tp]
    float acc[4][4];                                     // up to 4 items per thread
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.f; }
    for (int l = blockIdx.x; l < d.L; l += gridDim.x) {
        __syncthreads();
        const float* prev = (l > 0) ? p.s.att + ((long)b * d.L + (l - 1)) * d.Tp : nullptr;
        const float uni = 1.f / (float)max(len, 1);
        for (int i = tid; i < wpa; i += 512) {
            const int tau = i - d.Ks;
            float v = 0.f;
            if (tau >= 0 && tau < d.Tp) v = prev ? prev[tau] : (tau < len ? uni : 0.f);
            s_pa[i] = v;
        }
        const float* dconv = p.s.conv + ((long)b * d.L + l) * d.Kn * d.Tp;
        for (int i = tid; i < d.Kn * d.Tp; i += 512) s_dc[i] = dconv[i];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int it = tid + 512 * u;
            if (it < nitem) {
                const int k = it / tg, j0 = 4 * (it - k * tg);
                const float* dc = s_dc + k * d.Tp;
                const float* pa = s_pa + j0;                 // pa[tau + jj] = prev_att[tau + j0 + jj - Ks]
                float q0 = pa[0], q1 = pa[1], q2 = pa[2];
                float a0 = acc[u][0], a1 = acc[u][1], a2 = acc[u][2], a3 = acc[u][3];
                for (int tau = 0; tau < len; ++tau) {
                    const float w = dc[tau], q3 = pa[tau + 3];
                    a0 += w * q0; a1 += w * q1; a2 += w * q2; a3 += w * q3;
                    q0 = q1; q1 = q2; q2 = q3;
                }
                acc[u][0] = a0; acc[u][1] = a1; acc[u][2] = a2; acc[u][3] = a3;
            }
        }
    }
    float* o = out + ((long)blockIdx.y * gridDim.x + blockIdx.x) * d.Kn * taps;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int it = tid + 512 * u;
        if (it < nitem) {
            const int k = it / tg, j0 = 4 * (it - k * tg);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) if (j0 + jj < taps) o[k * taps + j0 + jj] = acc[u][jj];
        }
    }
}

// The same gradient on the matrix cores (bf16 mode, Kn <= 16): per (b, l) the product  dconv (Kn x T')  x  Toeplitz(prev_att)
// (T' x taps),  D[k][j] += sum_tau dconv[k][tau] * pa[tau + j]  with pa the zero-padded previous attention.  A tile = the
// dconv rows of a 32-frame chunk (loaded once per chunk), B tile = 32 x 16 window of pa for taps j0..j0+15 (eight LDS words
// per lane); 16-tap tiles are dealt round-robin to the 4 waves, whose accumulators run over all frames and all steps of
// the workgroup.  Same grid, slots and reduction as wconv_grad_kernel.
__global__ __launch_bounds__(256) void wconv_grad_mfma_kernel(DecP p, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const asr_dec_dims_t& d = p.d;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int taps = 2 * d.Ks + 1;
    const int ntile = (taps + 15) >> 4;                   // 16-tap output tiles
    const int len = min((int)p.enc_len[b], d.Tp);
    // frames are walked in segments of <= 768 (one LDS image of the dconv rows and of the attention window per segment), so any
    // T' is covered (round 2 took this kernel only up to T' = 768 and fell back to the VALU kernel beyond: 14 ms at config 5)
    const int TpP = 32 * ((d.Tp + 31) >> 5);
    const int SEG = min(TpP, 768);
    const int wpa = SEG + 16 * ntile + 16;
    float* s_pa = smem_f;                                // [SEG + 16*ntile + 16]  zero-padded previous attention: s_pa[i] = att[seg + i - Ks]
    float* s_dc = s_pa + ((wpa + 3) & ~3);               // [16][SEG]  dconv rows of the segment, zero beyond Kn / len
    constexpr int MAXT = 4;                              // tiles per wave (taps <= 256)
    f32x4 acc[MAXT];
#pragma unroll
    for (int u = 0; u < MAXT; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int m = lane & 15, q = lane >> 4;
    for (int l = blockIdx.x; l < d.L; l += gridDim.x) {
        const float* prev = (l > 0) ? p.s.att + ((long)b * d.L + (l - 1)) * d.Tp : nullptr;
        const float uni = 1.f / (float)max(len, 1);
        const float* dconv = p.s.conv + ((long)b * d.L + l) * d.Kn * d.Tp;
        for (int seg = 0; seg < len; seg += SEG) {
            __syncthreads();
            // staging with all loads of a batch in flight (clamped addresses, zeroing by selects): SEG <= 768, wpa <= 1024
            {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int tau = seg + tid + 256 * j - d.Ks;
                    v[j] = prev ? prev[min(max(tau, 0), d.Tp - 1)] : uni;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = tid + 256 * j, tau = seg + i - d.Ks;
                    if (i < wpa) s_pa[i] = (tau >= 0 && tau < (prev ? d.Tp : len)) ? v[j] : 0.f;
                }
            }
#pragma unroll 1
            for (int k0 = 0; k0 < 16; k0 += 4) {
                float v[4][3];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        v[kk][j] = dconv[(long)min(k0 + kk, d.Kn - 1) * d.Tp + min(seg + tid + 256 * j, d.Tp - 1)];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int ti = tid + 256 * j;
                        if (ti < SEG) s_dc[(k0 + kk) * SEG + ti] = (k0 + kk < d.Kn && seg + ti < len) ? v[kk][j] : 0.f;
                    }
            }
            __syncthreads();
            const int nchunk = (min(len - seg, SEG) + 31) >> 5;
            for (int c = 0; c < nchunk; ++c) {
                const float* ar = s_dc + m * SEG + 32 * c + 8 * q;
                const float4 a0 = *reinterpret_cast<const float4*>(ar), a1 = *reinterpret_cast<const float4*>(ar + 4);
                bf16x8 av;
                av[0] = (__bf16)a0.x; av[1] = (__bf16)a0.y; av[2] = (__bf16)a0.z; av[3] = (__bf16)a0.w;
                av[4] = (__bf16)a1.x; av[5] = (__bf16)a1.y; av[6] = (__bf16)a1.z; av[7] = (__bf16)a1.w;
#pragma unroll
                for (int u = 0; u < MAXT; ++u) {
                    const int tile = wave + 4 * u;
                    if (tile < ntile) {
                        const float* br = s_pa + 32 * c + 8 * q + 16 * tile + m;      // pa[tau + j], tau = seg + 32c + 8q + e, j = 16*tile + m
                        bf16x8 bv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) bv[e] = (__bf16)br[e];
                        acc[u] = mma16(av, bv, acc[u]);
                    }
                }
            }
        }
    }
    // D[row = 4q + r = k][col = m = tap within the tile]
    float* o = out + ((long)blockIdx.y * gridDim.x + blockIdx.x) * d.Kn * taps;
#pragma unroll
    for (int u = 0; u < MAXT; ++u) {
        const int tile = wave + 4 * u, j = 16 * tile + m;
        if (tile < ntile && j < taps) {
#pragma unroll
            for (int r = 0; r < 4; ++r) if (4 * q + r < d.Kn) o[(4 * q + r) * taps + j] = acc[u][r];
        }
    }
}

// B4: query backward.  dq[b,t,:] already holds the gradient wrt the query pre-activation;  dhs[b,t-1,:] += dq * W_q.
template <bool BF16>
__device__ __forceinline__ void dec_query_bwd_body(const DecB& p, int t, int fuse_elem, int qblock) {
    // fuse_elem (single-layer decoders): dhs[b,t-1,:] is complete once this kernel has added its part, so the same
    // thread goes on with the cell backward of step t-1 for its (row, unit) - the elementwise kernel of that step and
    // its launch boundary are saved.  Operands of that epilogue are requested before the contraction.
    __shared__ float red[4][256];
    const asr_dec_dims_t& d = p.f.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 15, q = lane >> 4;
    const int c = qblock * 16 + n;
    const bool cok = c < d.Q;
    const float* wrow = p.wqT + (long)(cok ? c : 0) * d.A;
    for (int m0 = 0; m0 < d.B; m0 += 16) {
        const int ab = m0 + n;
        const bool rok = ab < d.B;
        const float* drow = p.dq + ((long)(rok ? ab : 0) * d.L + t) * d.A;
        const int row = tid >> 4, col = tid & 15;
        const int b = m0 + row, cc = qblock * 16 + col;
        const bool ook = b < d.B && cc < d.Q;
        const long ri = (long)(ook ? b : 0) * d.L + (t - 1);
        float* dh = p.dhs + ri * d.Q + (ook ? cc : 0);
        const float dh_old = *dh;
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f, ct = 0.f, cp = 0.f, carry = 0.f;
        float* g = p.f.s.gates + ri * 4 * d.Dd + (ook ? cc : 0);          // NL == 1 when fuse_elem
        const long ci = (long)(ook ? b : 0) * d.Dd + (ook ? cc : 0);
        if (fuse_elem) {
            gi = g[0]; gf = g[d.Dd]; gg = g[2 * d.Dd]; go = g[3 * d.Dd];
            ct = p.f.s.cs[ri * d.Dd + (ook ? cc : 0)];
            cp = p.f.s.cs[(ri - (t > 1 ? 1 : 0)) * d.Dd + (ook ? cc : 0)];
            if (t <= 1) cp = 0.f;
            carry = p.dcf[ci];
        }
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = dot_rows<BF16>(drow, rok, wrow, cok, d.A, wave, 4, (d.A % 4) == 0, acc);
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][(4 * q + r) * 16 + n] = acc[r];
        __syncthreads();
        if (ook) {
            const float dhv = dh_old + red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
            *dh = dhv;
            if (fuse_elem) {
                const float tc = tanhf(ct);
                const float dc = dhv * go * (1.f - tc * tc) + carry;
                g[0] = dc * gg * gi * (1.f - gi);
                g[d.Dd] = dc * cp * gf * (1.f - gf);
                g[2 * d.Dd] = dc * gi * (1.f - gg * gg);
                g[3 * d.Dd] = dhv * tc * go * (1.f - go);
                p.dcf[ci] = dc * gf;
            }
        }
        __syncthreads();
    }
}

// B3 + B4 in one launch: the location-conv gradient and the query backward both consume the attention backward's outputs
// and do not depend on each other, so their workgroups share a launch (one dependent kernel boundary per step less; the
// launch lasts as long as the longer of the two).  blockIdx.x < nconv: conv tile (bx, by); else query block.
template <bool BF16>
__global__ __launch_bounds__(512) void att_bwd_tail_kernel(DecB p, int t, int TC, int gx, int nconv, int fuse_elem) {
    if ((int)blockIdx.x < nconv) {
        att_bwd_conv_body(p, t, TC, blockIdx.x % gx, blockIdx.x / gx);
    } else {
        if (threadIdx.x >= 256) return;                 // the matvec body is written for 4 waves
        dec_query_bwd_body<BF16>(p, t, fuse_elem, blockIdx.x - nconv);
    }
}

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int R, int C, long ld_dst, long col0) {
    // src (R x C) -> dst[(c + 0) * ld_dst + col0 + r]
    const long total = (long)R * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / C), c = (int)(i % C);
        dst[(long)c * ld_dst + col0 + r] = src[i];
    }
}

// reduce the per-workgroup slots into the parameter gradients: one wave per output element
__global__ __launch_bounds__(256) void slot_reduce_kernel(const float* __restrict__ slots, int nslots, int slot, float* __restrict__ out,
                                                          int off, int n, int tr_rows, int tr_cols) {
    // tr_rows > 0: the slot holds this block transposed ([col][row]) and `out` is [row][col]
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n) return;
    const int src = tr_rows > 0 ? (i % tr_cols) * tr_rows + (i / tr_cols) : i;
    float acc = 0.f;
    for (int s = lane; s < nslots; s += 64) acc += slots[(long)s * slot + off + src];
    acc = wave_sum(acc);
    if (lane == 0) out[i] += acc;
}

// embedding gradient: dE[v,:] += sum over (b,t) with tokens[b,t] == v of dxin[b,t,0:Dd]   (deterministic)
// grid (V, ceil(Dd/64)); wave g owns the positions i = g (mod 4): it first compacts the matching ones, in order, into an LDS
// list (coalesced token scan, ballot + prefix count), then sums those rows with independent loads; lanes over 64 columns
constexpr int EMB_LIST_MAX = 3584;          // positions per wave list and chunk (4 lists of 14 KB in LDS)
__global__ __launch_bounds__(256) void embed_bwd_kernel(const float* __restrict__ dxin, const int64_t* __restrict__ tokens,
                                                        float* __restrict__ demb, int B, int L, int Dd, int XW, int V) {
    extern __shared__ int s_rows[];                       // 4 lists of `cper` positions, rebuilt per chunk of 4*cper positions
    __shared__ float red[4][64];
    const int v = blockIdx.x, lane = threadIdx.x & 63, k = blockIdx.y * 64 + lane, grp = threadIdx.x >> 6;
    const int BL = B * L, per = (BL + 3) / 4;
    const int cper = min(per, EMB_LIST_MAX);
    int* list = s_rows + grp * cper;
    float acc = 0.f;
    const int kc = min(k, Dd - 1);
    for (int p0 = 0; p0 < per; p0 += cper) {             // positions 4*(p0 + j) + grp, j < cper: ascending per wave, so the sum order is fixed
        int n = 0;
        const int pend = min(per, p0 + cper);
        for (int base = p0; base < pend; base += 64) {
            const int i = 4 * (base + lane) + grp;
            const bool m = (base + lane < pend) && i < BL && tokens[min(i, BL - 1)] == v;
            const unsigned long long bal = __ballot(m);
            const int before = __builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0));
            if (m) list[n + before] = i;
            n += __popcll(bal);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");     // the list is read by the wave that wrote it
        int q = 0;
        for (; q + 4 <= n; q += 4) {
            const float x0 = dxin[(long)list[q] * XW + kc], x1 = dxin[(long)list[q + 1] * XW + kc];
            const float x2 = dxin[(long)list[q + 2] * XW + kc], x3 = dxin[(long)list[q + 3] * XW + kc];
            acc += x0; acc += x1; acc += x2; acc += x3;
        }
        for (; q < n; ++q) acc += dxin[(long)list[q] * XW + kc];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the list is rewritten by the next chunk
    }
    red[grp][lane] = acc;
    __syncthreads();
    if (grp == 0 && k < Dd) demb[(long)v * Dd + k] += red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
}

inline size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

size_t dec_bwd_persist_work_bytes_fw(const asr_dec_dims_t& d);
int dec_bwd_persist_tiles_fw(const asr_dec_dims_t& d);
int dec_bwd_persist_tiles_max_fw(const asr_dec_dims_t& d);
struct BwdLayout {
    size_t dhs, dxin, dq, dkey, datt_next, dcf, wq_t, slots, wslots, dkeypre, wcat[ASR_MAX_DEC_LAYERS], pwork, pwork_bytes, total;
    int ntp, ntp_max;       // tiles per utterance of the persistent backward under the current plan preference (0: no plan) / the larger of its two plans
    int nte, slot;          // energy-backward tiles per utterance, floats per slot
    int TE, NG;             // frames per energy-backward workgroup, frame groups
    int TC;                 // outputs per conv-backward workgroup
    int nch;                // step chunks of the W_conv gradient kernel
    size_t lds_e, lds_c, lds_w;
};
BwdLayout bwd_layout(const asr_dec_dims_t& d) {
    BwdLayout o;
    size_t off = 0;
    auto take = [&](size_t nfloat) { size_t r = off; off += align_up(nfloat * sizeof(float)); return r; };
    const int XW = d.Dd + d.E;
    const int taps = 2 * d.Ks + 1;
    // energy backward: NG frame groups x ceil(A/64) waves (<= 640 threads); the smallest tile that keeps the batch in
    // one round of workgroups and fits the LDS budget
    const int nwa = cdiv(d.A, 64);
    o.NG = nwa <= 2 ? 4 : (nwa <= 5 ? 2 : 1);
    int ap4 = (d.A + 3) / 4; if ((ap4 & 1) == 0) ++ap4;
    const int AP = 4 * ap4, KP = 16;
    auto lds_of = [&](int te) {
        return sizeof(float) * ((size_t)te * KP + 2 * (size_t)te + ((d.E + 3) & ~3) + (size_t)d.Kn * AP + (size_t)te * AP +
                                (size_t)o.NG * (d.Kn + 2) * d.A);
    };
    const int te_cand[] = {8, 16, 24, 32, 40, 48, 64};
    o.TE = 8;
    for (int i = 0; i < 7; ++i) {
        const int te = te_cand[i];
        if (lds_of(te) > 150 * 1024) break;
        o.TE = te;
        if ((long)d.B * cdiv(d.Tp, te) <= 256) break;
    }
    o.lds_e = lds_of(o.TE);
    o.nte = cdiv(d.Tp, o.TE);
    o.slot = d.A * (1 + d.Kn) + 1;
    // conv backward
    const int tc_cand[] = {8, 16, 24, 32, 40, 48, 64};
    o.TC = 64;
    for (int i = 0; i < 7; ++i) if ((long)d.B * cdiv(d.Tp, tc_cand[i]) <= 256) { o.TC = tc_cand[i]; break; }
    {
        const int nout = d.Kn * (o.TC / 4);
        const int parts = nout >= 512 ? 1 : (512 / nout > 8 ? 8 : 512 / nout);
        o.lds_c = sizeof(float) * ((size_t)((d.Kn * taps + 3) & ~3) + (size_t)d.Kn * (o.TC + 2 * d.Ks + 4) + (size_t)parts * d.Kn * o.TC);
    }
    // W_conv gradient
    o.nch = (int)(256 / d.B > 0 ? 256 / d.B : 1);
    if (o.nch > d.L) o.nch = d.L;
    o.lds_w = sizeof(float) * ((size_t)((d.Tp + 2 * d.Ks + 8 + 3) & ~3) + (size_t)d.Kn * d.Tp);
    // the zero-initialised accumulators first and back to back, the persistent launch's status block right behind them: ONE fill
    // from `dhs` to `pwork + 256` at the start of a call (five separate fills before)
    o.dhs = take((size_t)d.B * d.L * d.NL * d.Dd);
    o.dq = take((size_t)d.B * d.L * d.A);
    o.dkey = take((size_t)d.B * d.Tp * d.A);
    o.ntp = dec_bwd_persist_tiles_fw(d);
    o.ntp_max = dec_bwd_persist_tiles_max_fw(d);
    o.slots = take((size_t)d.B * (o.nte > o.ntp_max ? o.nte : o.ntp_max) * o.slot);
    o.pwork_bytes = dec_bwd_persist_work_bytes_fw(d);
    o.pwork = take(o.pwork_bytes / sizeof(float) + 64);
    o.dxin = take((size_t)d.B * d.L * XW);
    o.dkeypre = take((size_t)d.B * d.Tp * d.A);
    o.datt_next = take((size_t)d.B * d.Tp);
    o.dcf = take((size_t)d.NL * d.B * d.Dd);
    o.wq_t = take((size_t)d.Q * d.A);
    o.wslots = take((size_t)d.B * o.nch * d.Kn * taps);
    for (int l = 0; l < d.NL; ++l) o.wcat[l] = take((size_t)((l == 0 ? XW : d.Dd) + d.Dd) * 4 * d.Dd);
    o.total = off;
    return o;
}

int check_dims(const asr_dec_dims_t& d, const char* who) {
    ASR_REQUIRE(d.B > 0 && d.Tp > 0 && d.E > 0 && d.A > 0 && d.Dd > 0 && d.V > 1 && d.L > 0, ASR_E_ARG, "%s: bad dims", who);
    ASR_REQUIRE(d.NL >= 1 && d.NL <= ASR_MAX_DEC_LAYERS, ASR_E_UNSUPPORTED, "%s: %d decoder layers (max %d)", who, d.NL, ASR_MAX_DEC_LAYERS);
    ASR_REQUIRE(d.Q == d.Dd * d.NL, ASR_E_ARG, "%s: Q must equal Dd*NL", who);
    ASR_REQUIRE(d.Kn >= 1 && d.Kn <= 16, ASR_E_UNSUPPORTED, "%s: loc_kernel_num %d not in [1,16]", who, d.Kn);
    ASR_REQUIRE(d.Ks >= 0 && d.Ks <= 512, ASR_E_UNSUPPORTED, "%s: loc_kernel_size %d too large", who, d.Ks);
    ASR_REQUIRE(d.temperature > 0.f, ASR_E_ARG, "%s: temperature must be > 0", who);
    ASR_REQUIRE(d.Tp <= 12000, ASR_E_UNSUPPORTED, "%s: T'=%d exceeds the LDS row budget", who, d.Tp);
    return ASR_OK;
}

// Launch geometry of the forward energy kernel: the smallest frames-per-wave count that puts the whole batch in one
// round of workgroups (<= 256), within the register budget of the (KNMAX, TPW) instantiations.
struct EnergyPlan { int tpw; dim3 grid; size_t lds; };
EnergyPlan energy_plan(const asr_dec_dims_t& d) {
    const int cand_small[] = {2, 4, 5, 8}, cand_big[] = {2, 4, 5};
    const int* cand = (d.Kn <= 4) ? cand_small : cand_big;
    const int nc = (d.Kn <= 4) ? 4 : (d.Kn <= 10 ? 3 : 2);
    int tpw = cand[nc - 1];
    for (int i = 0; i < nc; ++i)
        if ((long)d.B * cdiv(d.Tp, 8 * cand[i]) <= 256) { tpw = cand[i]; break; }
    EnergyPlan pl;
    pl.tpw = tpw;
    const int TE = 8 * tpw, taps = 2 * d.Ks + 1, win = TE + 2 * d.Ks;
    pl.grid = dim3(cdiv(d.Tp, TE), d.B);
    const int nout = d.Kn * (TE / 4);
    const int parts = nout >= 512 ? 1 : (512 / nout > 8 ? 8 : 512 / nout);
    pl.lds = sizeof(float) * ((size_t)((win + 3) & ~3) + ((d.Kn * taps + 3) & ~3) + (size_t)d.Kn * TE + ((d.Kn * d.A + 3) & ~3) +
                              (size_t)parts * d.Kn * TE);
    return pl;
}
template <int KNMAX, bool H16>
void launch_energy_k(const DecP& p, int t, const EnergyPlan& pl, hipStream_t st) {
    switch (pl.tpw) {
        case 2: hipLaunchKernelGGL((att_energy_kernel<KNMAX, 2, H16>), pl.grid, dim3(512), pl.lds, st, p, t); break;
        case 4: hipLaunchKernelGGL((att_energy_kernel<KNMAX, 4, H16>), pl.grid, dim3(512), pl.lds, st, p, t); break;
        case 5: if constexpr (KNMAX <= 10) { hipLaunchKernelGGL((att_energy_kernel<KNMAX, 5, H16>), pl.grid, dim3(512), pl.lds, st, p, t); } break;
        default: if constexpr (KNMAX <= 4) { hipLaunchKernelGGL((att_energy_kernel<KNMAX, 8, H16>), pl.grid, dim3(512), pl.lds, st, p, t); } break;
    }
}
void launch_energy(const DecP& p, int t, const EnergyPlan& pl, hipStream_t st) {
    const bool h16 = p.s.key16 != nullptr;
    if (p.d.Kn <= 4) { if (h16) launch_energy_k<4, true>(p, t, pl, st); else launch_energy_k<4, false>(p, t, pl, st); }
    else if (p.d.Kn <= 10) { if (h16) launch_energy_k<10, true>(p, t, pl, st); else launch_energy_k<10, false>(p, t, pl, st); }
    else { if (h16) launch_energy_k<16, true>(p, t, pl, st); else launch_energy_k<16, false>(p, t, pl, st); }
}

}  // namespace

// decoder_persist.hip
size_t dec_fwd_persist_work_bytes(const asr_dec_dims_t& d);
int dec_fwd_persistent(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const float* enc,
                       const int64_t* enc_len, void* work, size_t work_bytes, hipStream_t st);

size_t dec_bwd_persist_work_bytes(const asr_dec_dims_t& d);
int dec_bwd_persist_tiles(const asr_dec_dims_t& d);
int dec_bwd_persist_tiles_max(const asr_dec_dims_t& d);
int dec_bwd_plan_kind(const asr_dec_dims_t& d);
namespace { size_t dec_bwd_persist_work_bytes_fw(const asr_dec_dims_t& d) { return dec_bwd_persist_work_bytes(d); }
            int dec_bwd_persist_tiles_fw(const asr_dec_dims_t& d) { return dec_bwd_persist_tiles(d); }
            int dec_bwd_persist_tiles_max_fw(const asr_dec_dims_t& d) { return dec_bwd_persist_tiles_max(d); } }
int dec_bwd_persistent(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const int64_t* enc_len,
                       const float* dhs, float* dxin, float* dq, float* dkey, float* slots, int slot, const float* wcatT, const float* wqT,
                       void* work, size_t work_bytes, float** dgates_out, hipStream_t st);

int dec_fwd_plan_kind(const asr_dec_dims_t& d);
extern "C" size_t asr_att_decoder_fwd_work_bytes(const asr_dec_dims_t* dims) {
    return dims ? dec_fwd_persist_work_bytes(*dims) : 0;
}
extern "C" int asr_att_decoder_fwd_plan(const asr_dec_dims_t* dims) { return dims ? dec_fwd_plan_kind(*dims) : 0; }
extern "C" int asr_att_decoder_bwd_plan(const asr_dec_dims_t* dims) { return dims ? dec_bwd_plan_kind(*dims) : 0; }

extern "C" int asr_att_decoder_fwd(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights,
                                   const float* enc, const int64_t* enc_len, const int64_t* teacher, int teacher_ld,
                                   const asr_dec_state_t* state, int prec, asr_stream_t stream) {
    ASR_REQUIRE(dims && weights && enc && enc_len && state, ASR_E_ARG, "asr_att_decoder_fwd: null pointer");
    const asr_dec_dims_t& d = *dims;
    int rc = check_dims(d, "asr_att_decoder_fwd");
    if (rc != ASR_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    DecP p{d, *weights, *state, enc, enc_len};
    const int XW = d.Dd + d.E;
    const bool bf = (prec == ASR_BF16);
    // the abort word (first word of `work`) is defined after EVERY call, whichever kernels run: callers fold it into
    // their status word without knowing whether the persistent launch was taken
    if (state->work && state->work_bytes >= 256) hipMemsetAsync(state->work, 0, 256, st);

    // key = tanh(enc W_k^T + b_k), once per batch (src/asr.py:345)
    rc = asr_gemm(enc, weights->Wk, state->key, weights->bk, d.B * d.Tp, d.A, d.E, d.E, d.E, d.A, 1, 1, ASR_ACT_TANH, 0, 1,
                  1, 0, 0, 0, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;
    // bf16 working copies of the two tensors every step re-reads (bf16 contraction mode only; the caller provides the
    // buffers): they halve the loop's HBM traffic and bring its per-XCD working set under the 4 MB L2
    if (state->key16) hipLaunchKernelGGL(cast_bf16_kernel, dim3(1024), dim3(256), 0, st, state->key, (unsigned short*)state->key16, (long)d.B * d.Tp * d.A);
    if (state->enc16) hipLaunchKernelGGL(cast_bf16_kernel, dim3(1024), dim3(256), 0, st, enc, (unsigned short*)state->enc16, (long)d.B * d.Tp * d.E);

    if (teacher) {
        hipLaunchKernelGGL(shift_tokens_kernel, dim3(cdiv(d.B * d.L, 256)), dim3(256), 0, st, teacher, state->tokens, d.B, d.L, teacher_ld);
        hipLaunchKernelGGL(embed_kernel, dim3(cdiv((long)d.B * d.L * d.Dd, 256)), dim3(256), 0, st, weights->emb, state->tokens,
                           state->xin, d.B, d.L, d.Dd, XW, 0, d.L, d.V);
    } else {
        hipMemsetAsync(state->tokens, 0, sizeof(int64_t) * d.B * d.L, st);
    }

    const EnergyPlan epl = energy_plan(d);
    ASR_REQUIRE(epl.lds <= 64 * 1024, ASR_E_UNSUPPORTED, "asr_att_decoder_fwd: energy tile needs %zu B of LDS", epl.lds);
    int t_begin = 0;
    if (teacher && bf && state->work) {
        // the whole teacher-forced loop as one persistent launch (decoder_persist.hip) when the shape has a plan
        rc = dec_fwd_persistent(d, *weights, *state, enc, enc_len, state->work, state->work_bytes, st);
        if (rc < 0) return rc;
        if (rc == ASR_OK) t_begin = d.L;
    }
    for (int t = t_begin; t < d.L; ++t) {
        if (!teacher)
            hipLaunchKernelGGL(embed_kernel, dim3(cdiv((long)d.B * d.Dd, 256)), dim3(256), 0, st, weights->emb, state->tokens,
                               state->xin, d.B, d.L, d.Dd, XW, t, 1, d.V);
        if (bf) hipLaunchKernelGGL(dec_query_kernel<true>, dim3(cdiv(d.A, 16)), dim3(256), 0, st, p, t);
        else    hipLaunchKernelGGL(dec_query_kernel<false>, dim3(cdiv(d.A, 16)), dim3(256), 0, st, p, t);
        launch_energy(p, t, epl, st);
        if (state->enc16 && (d.E & 3) == 0) hipLaunchKernelGGL(att_softmax_ctx_kernel<true>, dim3(cdiv(d.E, 64), d.B), dim3(256), sizeof(float) * d.Tp, st, p, t);
        else hipLaunchKernelGGL(att_softmax_ctx_kernel<false>, dim3(cdiv(d.E, 64), d.B), dim3(256), sizeof(float) * d.Tp, st, p, t);
        for (int l = 0; l < d.NL; ++l) {
            if (bf) hipLaunchKernelGGL(dec_cell_fwd_kernel<true>, dim3(cdiv(d.Dd, 4)), dim3(256), 0, st, p, t, l);
            else    hipLaunchKernelGGL(dec_cell_fwd_kernel<false>, dim3(cdiv(d.Dd, 4)), dim3(256), 0, st, p, t, l);
        }
        if (!teacher) hipLaunchKernelGGL(dec_greedy_kernel, dim3(d.B), dim3(256), sizeof(float) * d.V, st, p, t);
    }
    ASR_LAUNCH_CHECK("asr_att_decoder_fwd");
    if (teacher) {
        // logits[b,t,:] = W_c h_top[b,t] + b_c for all steps in one contraction
        rc = asr_gemm(state->hs + (size_t)(d.NL - 1) * d.Dd, weights->Wc, state->logits, weights->bc, d.B * d.L, d.V, d.Dd,
                      (long)d.NL * d.Dd, d.Dd, d.V, 1, 1, ASR_ACT_NONE, 0, 1, 1, 0, 0, 0, 0, 0, prec, stream);
        if (rc != ASR_OK) return rc;
    }
    return ASR_OK;
}

// One decode step t for every row of the state (beam search: rows = live hypotheses, all at the same step).
// The caller has set state->tokens[:, t] (input token of the step) and, for t > 0, the step t-1 entries of
// hs / cs / att of every row (parents' values after pruning).  Produces att[:, t], xin[:, t], gates/cs/hs[:, t]
// and logits[:, t].  (state->key must hold tanh(proj_k(enc)) — asr_att_decoder_keys.)
extern "C" int asr_att_decoder_keys(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const float* enc,
                                    float* key, int prec, asr_stream_t stream) {
    ASR_REQUIRE(dims && weights && enc && key, ASR_E_ARG, "asr_att_decoder_keys: null pointer");
    const asr_dec_dims_t& d = *dims;
    return asr_gemm(enc, weights->Wk, key, weights->bk, d.B * d.Tp, d.A, d.E, d.E, d.E, d.A, 1, 1, ASR_ACT_TANH, 0, 1, 1, 0, 0, 0, 0, 0,
                    prec, stream);
}

extern "C" int asr_att_decoder_step(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights,
                                    const float* enc, const int64_t* enc_len, const asr_dec_state_t* state, int t,
                                    int prec, asr_stream_t stream) {
    ASR_REQUIRE(dims && weights && enc && enc_len && state, ASR_E_ARG, "asr_att_decoder_step: null pointer");
    const asr_dec_dims_t& d = *dims;
    int rc = check_dims(d, "asr_att_decoder_step");
    if (rc != ASR_OK) return rc;
    ASR_REQUIRE(t >= 0 && t < d.L, ASR_E_ARG, "asr_att_decoder_step: step %d outside [0,%d)", t, d.L);
    hipStream_t st = (hipStream_t)stream;
    DecP p{d, *weights, *state, enc, enc_len};
    const int XW = d.Dd + d.E;
    const bool bf = (prec == ASR_BF16);
    const EnergyPlan epl = energy_plan(d);
    ASR_REQUIRE(epl.lds <= 64 * 1024, ASR_E_UNSUPPORTED, "asr_att_decoder_step: energy tile needs %zu B of LDS", epl.lds);
    hipLaunchKernelGGL(embed_kernel, dim3(cdiv((long)d.B * d.Dd, 256)), dim3(256), 0, st, weights->emb, state->tokens, state->xin,
                       d.B, d.L, d.Dd, XW, t, 1, d.V);
    if (bf) hipLaunchKernelGGL(dec_query_kernel<true>, dim3(cdiv(d.A, 16)), dim3(256), 0, st, p, t);
    else    hipLaunchKernelGGL(dec_query_kernel<false>, dim3(cdiv(d.A, 16)), dim3(256), 0, st, p, t);
    launch_energy(p, t, epl, st);
    if (state->enc16 && (d.E & 3) == 0) hipLaunchKernelGGL(att_softmax_ctx_kernel<true>, dim3(cdiv(d.E, 64), d.B), dim3(256), sizeof(float) * d.Tp, st, p, t);
    else hipLaunchKernelGGL(att_softmax_ctx_kernel<false>, dim3(cdiv(d.E, 64), d.B), dim3(256), sizeof(float) * d.Tp, st, p, t);
    for (int l = 0; l < d.NL; ++l) {
        if (bf) hipLaunchKernelGGL(dec_cell_fwd_kernel<true>, dim3(cdiv(d.Dd, 4)), dim3(256), 0, st, p, t, l);
        else    hipLaunchKernelGGL(dec_cell_fwd_kernel<false>, dim3(cdiv(d.Dd, 4)), dim3(256), 0, st, p, t, l);
    }
    // logits only (the arg-max side effect is redirected to a scratch row: t+1 >= L never written)
    DecP pl = p;
    hipLaunchKernelGGL(dec_logits_kernel, dim3(d.B), dim3(256), sizeof(float) * d.V, st, pl, t);
    ASR_LAUNCH_CHECK("asr_att_decoder_step");
    return ASR_OK;
}

// offset of the persistent backward's status block inside the workspace (diagnostics: tools/diag_dec.py)
extern "C" size_t asr_att_decoder_bwd_status_offset(const asr_dec_dims_t* dims) {
    if (!dims) return 0;
    return bwd_layout(*dims).pwork;
}

// tiles per utterance of the persistent backward launch (0: the shape has no plan and the per-step kernels run)
extern "C" int asr_att_decoder_bwd_persistent_tiles(const asr_dec_dims_t* dims) {
    if (!dims) return 0;
    return bwd_layout(*dims).ntp;
}

extern "C" size_t asr_att_decoder_bwd_workspace_bytes(const asr_dec_dims_t* dims) {
    if (!dims) return 0;
    return bwd_layout(*dims).total;
}

// reduction slices of a weight-gradient contraction (out_rows x out_cols, reduction over `depth` rows): these outputs are a
// handful of 128x128 tiles with a reduction thousands of rows deep, i.e. a few workgroups walking ~100 dependent k-steps;
// slices of >= 256 rows spread them over about one round of workgroups (fp32 atomics into the gradient)
static int wgrad_slices(int out_rows, int out_cols, int depth) {
    const int tiles = cdiv(out_rows, 128) * cdiv(out_cols, 128);
    int s = 768 / tiles;
    if (s > depth / 256) s = depth / 256;
    return s < 1 ? 1 : (s > 64 ? 64 : s);
}

extern "C" int asr_att_decoder_bwd_params(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                                          const float* enc, const int64_t* enc_len, const asr_dec_state_t* state, const float* dlogits,
                                          void* workspace, size_t workspace_bytes, int looped, int prec, asr_stream_t stream);
// pointer to the gate gradients the persistent backward leaves in its work area (decoder_persist.hip)
float* dec_bwd_persist_dgates(const asr_dec_dims_t& d, void* work);

extern "C" int asr_att_decoder_bwd_ex(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                                      const float* enc, const int64_t* enc_len, const asr_dec_state_t* state,
                                      const float* dlogits, float* denc,
                                      void* workspace, size_t workspace_bytes, int prec, int defer_params, int* looped_out,
                                      asr_stream_t stream) {
    ASR_REQUIRE(dims && weights && grads && enc && enc_len && state && dlogits && denc && workspace, ASR_E_ARG,
                "asr_att_decoder_bwd: null pointer");
    const asr_dec_dims_t& d = *dims;
    int rc = check_dims(d, "asr_att_decoder_bwd");
    if (rc != ASR_OK) return rc;
    const BwdLayout lay = bwd_layout(d);
    ASR_REQUIRE(workspace_bytes >= lay.total, ASR_E_ARG, "asr_att_decoder_bwd: workspace %zu < %zu", workspace_bytes, lay.total);
    ASR_REQUIRE(((uintptr_t)workspace & 255) == 0, ASR_E_ARG, "asr_att_decoder_bwd: workspace must be 256B aligned");
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int XW = d.Dd + d.E;
    const bool bf = (prec == ASR_BF16);
    const long SW = (long)d.NL * d.Dd;
    const int BL = d.B * d.L;

    DecB p;
    p.f = DecP{d, *weights, *state, enc, enc_len};
    p.g = *grads;
    p.dhs = (float*)(ws + lay.dhs); p.dxin = (float*)(ws + lay.dxin); p.dq = (float*)(ws + lay.dq);
    p.dkey = (float*)(ws + lay.dkey); p.datt_next = (float*)(ws + lay.datt_next);
    p.dcf = (float*)(ws + lay.dcf); p.wqT = (float*)(ws + lay.wq_t);
    p.slots = (float*)(ws + lay.slots); p.nte = lay.nte; p.slot = lay.slot;
    float* wslots = (float*)(ws + lay.wslots);
    float* dkeypre = (float*)(ws + lay.dkeypre);
    for (int l = 0; l < ASR_MAX_DEC_LAYERS; ++l) p.wcatT[l] = (l < d.NL) ? (float*)(ws + lay.wcat[l]) : nullptr;

    // zero-initialised accumulators dhs, dq, dkey, slots (dxin is fully written by the loop) and the abort word of the persistent
    // launch's status block (defined, 0, after every call - see asr_att_decoder_fwd): contiguous in the layout, one fill
    hipMemsetAsync(ws + lay.dhs, 0, lay.pwork + 256 - lay.dhs, st);

    // transposed weights so that every per-step contraction is K-contiguous
    for (int l = 0; l < d.NL; ++l) {
        const int Kx = (l == 0) ? XW : d.Dd;
        hipLaunchKernelGGL(transpose_kernel, dim3(256), dim3(256), 0, st, weights->Wih[l], p.wcatT[l], 4 * d.Dd, Kx, (long)4 * d.Dd, 0L);
        // rows Kx.. of wcatT hold W_hh^T
        hipLaunchKernelGGL(transpose_kernel, dim3(256), dim3(256), 0, st, weights->Whh[l], p.wcatT[l] + (size_t)Kx * 4 * d.Dd, 4 * d.Dd, d.Dd,
                           (long)4 * d.Dd, 0L);
    }
    hipLaunchKernelGGL(transpose_kernel, dim3(256), dim3(256), 0, st, weights->Wq, p.wqT, d.A, d.Q, (long)d.A, 0L);

    // output layer: dh_top = dlogits W_c ; dW_c += dlogits^T h_top ; db_c += colsum(dlogits)
    rc = asr_gemm(dlogits, weights->Wc, p.dhs + (size_t)(d.NL - 1) * d.Dd, nullptr, BL, d.Dd, d.V, d.V, d.Dd, SW, 1, 0,
                  ASR_ACT_NONE, 0, 1, 1, 0, 0, 0, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;

    const int taps = 2 * d.Ks + 1;
    const int nw_e = cdiv(d.A, 64);
    ASR_REQUIRE(state->conv, ASR_E_ARG, "asr_att_decoder_bwd: state->conv (saved location convolution) is NULL");
    ASR_REQUIRE(nw_e * lay.NG <= 10, ASR_E_UNSUPPORTED, "asr_att_decoder_bwd: attention dim %d > 640", d.A);
    ASR_REQUIRE(d.Kn * ((taps + 3) / 4) <= 2048, ASR_E_UNSUPPORTED, "asr_att_decoder_bwd: location filter bank %d x %d too large", d.Kn, taps);
    ASR_REQUIRE(lay.lds_e <= 150 * 1024 && lay.lds_c <= 148 * 1024 && lay.lds_w <= 150 * 1024, ASR_E_UNSUPPORTED,
                "asr_att_decoder_bwd: shape needs %zu / %zu / %zu B of LDS", lay.lds_e, lay.lds_c, lay.lds_w);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<10, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<16, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<10, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        hipFuncSetAttribute((const void*)att_bwd_energy_kernel<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        // (4 KB of static LDS of the query part count against the 160 KB)
        hipFuncSetAttribute((const void*)att_bwd_tail_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
        hipFuncSetAttribute((const void*)att_bwd_tail_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192);
        hipFuncSetAttribute((const void*)wconv_grad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
        attr_set = true;
    }
    bool looped = false;
    float* pdg = nullptr;         // gate gradients of the persistent backward (it leaves the saved gates intact)
    if (bf && lay.ntp > 0 && state->enc16 && d.NL == 1) {
        // the whole loop as one persistent, cluster-per-utterance launch (decoder_persist.hip)
        rc = dec_bwd_persistent(d, *weights, *state, enc_len, p.dhs, p.dxin, p.dq, p.dkey, p.slots, lay.slot, p.wcatT[0], p.wqT,
                                ws + lay.pwork, lay.pwork_bytes, &pdg, st);
        if (rc < 0) return rc;
        if (rc == ASR_OK) {
            looped = true;
            // (the kernel wrote the context part of dxin; the embedding part is only read by the embedding gradient and is
            // computed with it, asr_att_decoder_bwd_params)
        }
    }
    const dim3 grid_e(lay.nte, d.B), block_e(64 * nw_e * lay.NG);
    const bool h16 = state->key16 != nullptr && state->enc16 != nullptr && (d.E & 3) == 0;
    const dim3 grid_c(cdiv(d.Tp, lay.TC), d.B);
    for (int t = looped ? -1 : d.L - 1; t >= 0; --t) {
        const int last = (t == d.L - 1);
        const int fuse = (d.NL == 1) ? 1 : 0;    // the cell backward's elementwise part rides on the previous query backward
        for (int l = d.NL - 1; l >= 0; --l) {
            if (!fuse || last)
                hipLaunchKernelGGL(dec_cell_bwd_elem_kernel, dim3(cdiv(d.B * d.Dd, 256)), dim3(256), 0, st, p, t, l, last);
            const int width = ((l == 0) ? XW : d.Dd) + d.Dd;
            if (bf) hipLaunchKernelGGL(dec_cell_bwd_mm_kernel<true>, dim3(cdiv(width, 16)), dim3(256), 0, st, p, t, l);
            else    hipLaunchKernelGGL(dec_cell_bwd_mm_kernel<false>, dim3(cdiv(width, 16)), dim3(256), 0, st, p, t, l);
        }
        if (h16) {
            if (d.Kn <= 4)       hipLaunchKernelGGL((att_bwd_energy_kernel<4, true>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
            else if (d.Kn <= 10) hipLaunchKernelGGL((att_bwd_energy_kernel<10, true>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
            else                 hipLaunchKernelGGL((att_bwd_energy_kernel<16, true>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
        } else {
            if (d.Kn <= 4)       hipLaunchKernelGGL((att_bwd_energy_kernel<4, false>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
            else if (d.Kn <= 10) hipLaunchKernelGGL((att_bwd_energy_kernel<10, false>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
            else                 hipLaunchKernelGGL((att_bwd_energy_kernel<16, false>), grid_e, block_e, lay.lds_e, st, p, t, lay.TE, lay.NG, last);
        }
        if (t > 0) {
            // the location path reaches attn_{t-1}; step 0 convolves the constant initial attention
            const int nconv = (int)(grid_c.x * grid_c.y), nq = cdiv(d.Q, 16);
            if (bf) hipLaunchKernelGGL(att_bwd_tail_kernel<true>, dim3(nconv + nq), dim3(512), lay.lds_c, st, p, t, lay.TC, (int)grid_c.x, nconv, fuse);
            else    hipLaunchKernelGGL(att_bwd_tail_kernel<false>, dim3(nconv + nq), dim3(512), lay.lds_c, st, p, t, lay.TC, (int)grid_c.x, nconv, fuse);
        }
    }
    ASR_LAUNCH_CHECK("asr_att_decoder_bwd");

    // ---- the rest of the critical path: gradient wrt the encoder output ----------------------------------------
    // context: denc[b] += attn[b]^T (T' x L) dctx[b] (L x E)
    rc = asr_gemm(state->att, p.dxin + d.Dd, denc, nullptr, d.Tp, d.E, d.L, d.Tp, XW, d.E, 0, 0, ASR_ACT_NONE, 1, 1, d.B,
                  (long)d.L * d.Tp, (long)d.L * XW, (long)d.Tp * d.E, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;
    // key projection backward
    rc = asr_act_bwd(p.dkey, state->key, dkeypre, (long)d.B * d.Tp * d.A, ASR_ACT_TANH, stream);
    if (rc != ASR_OK) return rc;
    rc = asr_gemm(dkeypre, weights->Wk, denc, nullptr, d.B * d.Tp, d.E, d.A, d.A, d.E, d.E, 1, 0, ASR_ACT_NONE, 1, 1, 1, 0, 0, 0, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;
    ASR_LAUNCH_CHECK("asr_att_decoder_bwd(tail)");
    if (looped_out) *looped_out = looped ? 1 : 0;
    if (defer_params) return ASR_OK;
    return asr_att_decoder_bwd_params(dims, weights, grads, enc, enc_len, state, dlogits, workspace, workspace_bytes, looped ? 1 : 0, prec, stream);
}

extern "C" int asr_att_decoder_bwd(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                                   const float* enc, const int64_t* enc_len, const asr_dec_state_t* state,
                                   const float* dlogits, float* denc,
                                   void* workspace, size_t workspace_bytes, int prec, asr_stream_t stream) {
    return asr_att_decoder_bwd_ex(dims, weights, grads, enc, enc_len, state, dlogits, denc, workspace, workspace_bytes, prec, 0, nullptr, stream);
}

// Parameter gradients of the decoder from what asr_att_decoder_bwd_ex left in the workspace and the saved state: nothing on
// the path to the encoder gradient depends on them, so the caller may run this on another stream beside the encoder's BPTT.
extern "C" int asr_att_decoder_bwd_params(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                                          const float* enc, const int64_t* enc_len, const asr_dec_state_t* state, const float* dlogits,
                                          void* workspace, size_t workspace_bytes, int looped, int prec, asr_stream_t stream) {
    ASR_REQUIRE(dims && weights && grads && enc && enc_len && state && dlogits && workspace, ASR_E_ARG, "asr_att_decoder_bwd_params: null pointer");
    const asr_dec_dims_t& d = *dims;
    int rc = check_dims(d, "asr_att_decoder_bwd_params");
    if (rc != ASR_OK) return rc;
    const BwdLayout lay = bwd_layout(d);
    ASR_REQUIRE(workspace_bytes >= lay.total, ASR_E_ARG, "asr_att_decoder_bwd_params: workspace %zu < %zu", workspace_bytes, lay.total);
    hipStream_t st = (hipStream_t)stream;
    char* ws = (char*)workspace;
    const int XW = d.Dd + d.E;
    const bool bf = (prec == ASR_BF16);
    const long SW = (long)d.NL * d.Dd;
    const int BL = d.B * d.L;
    const int taps = 2 * d.Ks + 1;
    DecB p;
    p.f = DecP{d, *weights, *state, enc, enc_len};
    p.g = *grads;
    p.dxin = (float*)(ws + lay.dxin); p.dq = (float*)(ws + lay.dq);
    p.slots = (float*)(ws + lay.slots); p.nte = lay.nte; p.slot = lay.slot;
    float* wslots = (float*)(ws + lay.wslots);
    float* dkeypre = (float*)(ws + lay.dkeypre);
    const float* pdg = looped ? dec_bwd_persist_dgates(d, ws + lay.pwork) : nullptr;
    const int nslots_used = d.B * (looped ? lay.ntp : lay.nte);
    // output layer: dW_c += dlogits^T h_top ; db_c += colsum(dlogits)
    rc = asr_gemm(dlogits, state->hs + (size_t)(d.NL - 1) * d.Dd, grads->Wc, nullptr, d.V, d.Dd, BL, d.V, SW, d.Dd, 0, 0,
                  ASR_ACT_NONE, 1, wgrad_slices(d.V, d.Dd, BL), 1, 0, 0, 0, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;
    rc = asr_colsum(dlogits, d.V, BL, d.V, grads->bc, stream);
    if (rc != ASR_OK) return rc;
    // ---- batched parameter gradients -----------------------------------------------------------
    for (int l = 0; l < d.NL; ++l) {
        const float* dg = looped ? pdg : state->gates + (size_t)l * 4 * d.Dd;   // rows (b,t), stride NL*4Dd
        const long ldg = (long)d.NL * 4 * d.Dd;
        const int Kx = (l == 0) ? XW : d.Dd;
        const float* xl = (l == 0) ? state->xin : state->hs + (size_t)(l - 1) * d.Dd;
        const long ldx = (l == 0) ? XW : SW;
        rc = asr_gemm(dg, xl, grads->Wih[l], nullptr, 4 * d.Dd, Kx, BL, ldg, ldx, Kx, 0, 0, ASR_ACT_NONE, 1, wgrad_slices(4 * d.Dd, Kx, BL), 1, 0, 0, 0, 0,
                      0, prec, stream);
        if (rc != ASR_OK) return rc;
        rc = asr_gemm(dg, state->hs + (size_t)l * d.Dd, grads->Whh[l], nullptr, 4 * d.Dd, d.Dd, BL, ldg, SW, d.Dd, 0, 0, ASR_ACT_NONE,
                      1, wgrad_slices(4 * d.Dd, d.Dd, BL), 1, 0, 0, 0, d.L, -1, prec, stream);
        if (rc != ASR_OK) return rc;
        rc = asr_colsum2(dg, ldg, BL, 4 * d.Dd, grads->bih[l], grads->bhh[l], stream);
        if (rc != ASR_OK) return rc;
    }
    // query projection: dW_q += dqpre^T hcat_{t-1}, db_q += colsum(dqpre)
    rc = asr_gemm(p.dq, state->hs, grads->Wq, nullptr, d.A, d.Q, BL, d.A, d.Q, d.Q, 0, 0, ASR_ACT_NONE, 1, wgrad_slices(d.A, d.Q, BL), 1, 0, 0, 0, d.L, -1,
                  prec, stream);
    if (rc != ASR_OK) return rc;
    rc = asr_colsum(p.dq, d.A, BL, d.A, grads->bq, stream);
    if (rc != ASR_OK) return rc;
    // embedding.  After the persistent backward the embedding part of dxin is still to be formed:
    // dgates (B*L x 4Dd) . W_ih[:, :Dd]
    if (looped) {
        rc = asr_gemm(pdg, weights->Wih[0], p.dxin, nullptr, BL, d.Dd, 4 * d.Dd, 4 * d.Dd, XW, XW, 1, 0, ASR_ACT_NONE, 0, 1, 1,
                      0, 0, 0, 0, 0, prec, stream);
        if (rc != ASR_OK) return rc;
    }
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(d.V, cdiv(d.Dd, 64)), dim3(256), (size_t)4 * std::min(cdiv(BL, 4), EMB_LIST_MAX) * sizeof(int), st,
                       p.dxin, state->tokens, grads->emb, d.B, d.L, d.Dd, XW, d.V);
    // key projection
    const int M = d.B * d.Tp;
    const int splits = wgrad_slices(d.A, d.E, M);
    rc = asr_gemm(dkeypre, enc, grads->Wk, nullptr, d.A, d.E, M, d.A, d.E, d.E, 0, 0, ASR_ACT_NONE, 1, splits, 1, 0, 0, 0, 0, 0, prec, stream);
    if (rc != ASR_OK) return rc;
    rc = asr_colsum(dkeypre, d.A, M, d.A, grads->bk, stream);
    if (rc != ASR_OK) return rc;
    // slot partials -> d w_g, d W_proj, d b_g;  d W_conv from the saved dconv of every step
    const int nslots = nslots_used;
    hipLaunchKernelGGL(slot_reduce_kernel, dim3(cdiv(d.A, 4)), dim3(256), 0, st, p.slots, nslots, lay.slot, grads->wg, 0, d.A, 0, 0);
    hipLaunchKernelGGL(slot_reduce_kernel, dim3(cdiv(d.A * d.Kn, 4)), dim3(256), 0, st, p.slots, nslots, lay.slot, grads->Wproj, d.A, d.A * d.Kn, d.A, d.Kn);
    hipLaunchKernelGGL(slot_reduce_kernel, dim3(1), dim3(256), 0, st, p.slots, nslots, lay.slot, grads->bg, d.A * (1 + d.Kn), 1, 0, 0);
    {
        const int ntile = (taps + 15) / 16, SEG = std::min(32 * cdiv(d.Tp, 32), 768);
        const size_t lds_m = sizeof(float) * ((size_t)((SEG + 16 * ntile + 16 + 3) & ~3) + (size_t)16 * SEG);
        if (bf && d.Kn <= 16 && ntile <= 16 && lds_m <= 150 * 1024 && SEG + 16 * ntile + 16 <= 1024) {
            static bool attr_m = false;
            if (!attr_m) { hipFuncSetAttribute((const void*)wconv_grad_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); attr_m = true; }
            hipLaunchKernelGGL(wconv_grad_mfma_kernel, dim3(lay.nch, d.B), dim3(256), lds_m, st, p.f, wslots);
        } else {
            hipLaunchKernelGGL(wconv_grad_kernel, dim3(lay.nch, d.B), dim3(512), lay.lds_w, st, p.f, wslots);
        }
    }
    hipLaunchKernelGGL(slot_reduce_kernel, dim3(cdiv(d.Kn * taps, 4)), dim3(256), 0, st, wslots, d.B * lay.nch, d.Kn * taps, grads->Wconv,
                       0, d.Kn * taps, 0, 0);
    ASR_LAUNCH_CHECK("asr_att_decoder_bwd_params");
    return ASR_OK;
}


// ---- embedding gradient as a stand-alone entry (RNN-LM training, src/lm.py) ----------------------------------------
// demb[v, :] += sum over rows r with idx[r] == v of dy[r, :width]   (deterministic: fixed summation order per row)
extern "C" int asr_embedding_bwd(const float* dy, long dy_ld, const int64_t* idx, float* demb, int rows, int width, int V, asr_stream_t stream) {
    ASR_REQUIRE(dy && idx && demb && rows > 0 && width > 0 && V > 0 && dy_ld >= width && dy_ld <= 0x7fffffffL, ASR_E_ARG, "asr_embedding_bwd: bad args");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(V, cdiv(width, 64)), dim3(256), (size_t)4 * std::min(cdiv(rows, 4), EMB_LIST_MAX) * sizeof(int),
                       (hipStream_t)stream, dy, idx, demb, rows, 1, width, (int)dy_ld, V);
    ASR_LAUNCH_CHECK("asr_embedding_bwd");
    return ASR_OK;
}
