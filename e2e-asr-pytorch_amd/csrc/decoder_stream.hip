// Cluster-per-utterance decoder loop with STREAMED tiles: the plan for batches the LDS-resident kernels of decoder_persist.hip
// cannot hold (reference src/asr.py:123-175, src/module.py:1152-1173; batch shapes of src/collect_batch.py:21-24).
//
// decoder_persist.hip keeps the tile's key and enc rows in LDS for all L steps; that caps a tile at 40 frames and the batch at
// B <= 16 x T' <= 640 (B <= 8 up to T' = 750 / 850).  The reference's batches reach T' = 1 700 at B = 8, and BASELINE config 5
// is B = 64 x T' = 1 500.  Here the cluster scheme and the three all-gathers per step are the same, but
//   * a tile is TEB = 16 * ceil(T' / (16 NT)) frames of ANY size: NT = 32 / ceil(B/8) workgroups per utterance, one per CU of
//     the XCD the utterance lives on (B = 64: 4 tiles of 384 frames, B = 16: 16 tiles of <= 80, B = 8: 30 tiles of <= 64);
//   * the tile's key rows are read every step from a bf16 image in the sweep's own fragment order ([frame / 4][A][4], built
//     once per launch), its enc rows from the bf16 copy of the state; both come from the XCD's L2 when the utterances of an
//     XCD fit it (B <= 16: <= 4.6 MB per XCD) and from the Infinity Cache / HBM otherwise (config 5: 180 MB per step - the
//     step is then bound by that stream, ~30 us, not by the hand-offs);
//   * the energy sweep walks the tile in 16-frame MFMA tiles with the next tile's key fragment in flight, tiles past the
//     utterance's length are skipped, and the softmax statistics / partial context are two passes over the tile
//     (energies -> (m, s) -> weights in LDS -> weighted sum of the streamed enc rows), so nothing scales with TEB but LDS rows.
// Exchange records, tags, epochs, XCD-local consensus, poll / compute wave roles: decoder_cluster.h, as in decoder_persist.hip.
#include "decoder_cluster.h"

namespace {

struct PSF {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const float* enc;
    const int64_t* enc_len;
    const unsigned short* wcat16;   // (4Dd, KCP) bf16 rows [W_ih[:, Dd:Dd+E] | W_hh | 0-pad]
    const float* embproj;           // (B*L, 4Dd)  W_ih[:, :Dd] . emb(token)
    const unsigned short* key16t;   // [B][NT][pair][q][A][2][4] bf16: a lane's fragments of two 16-frame tiles side by side (build_key16p_kernel)
    u64* xbuf;
    unsigned* status;
    int NT, TEB, UPW, QPW, CPW;
    int HG2, QG2, SG2;
    int KC, KCP;
    int allow_local;
};

// LDS carve (floats behind the bf16 conv tile), shared by kernel and host plan
struct FCarve { int cvx_shorts, img, shorts, WT, ATP, NG, x2, q, wg, attp, wc, epart, e, w, g, cpart, stage, stage_floats, floats; };
__host__ __device__ inline FCarve fwd_carve(int TEB, int NT, int A, int E, int Kn, int Ks, int KCP, int UPW, int SG2) {
    FCarve c;
    c.cvx_shorts = TEB * FCVX_LD;
    c.img = (c.cvx_shorts + 7) & ~7;                          // bf16 images of the convolution (conv_geo)
    c.shorts = (c.img + conv_img_shorts(conv_geo(TEB, Ks)) + 7) & ~7;
    c.WT = (2 * Ks + 1 + 3) & ~3;
    c.ATP = (NT * TEB + 2 * Ks + 8 + 3) & ~3;
    const int nch = E >> 3;
    c.NG = (64 * NCW) / nch; if (c.NG > 8) c.NG = 8;          // frame groups of the partial-context pass
    int o = 0;
    c.x2 = o; o += 2 * KCP;
    c.q = o; o += (A + 3) & ~3;
    c.wg = o; o += (A + 3) & ~3;
    c.attp = o; o += c.ATP;
    c.wc = o;                                             // (the fp32 filter rows of dec_fwd_persist live in the bf16 filter image here)
    c.epart = o; o += NCW * TEB;
    c.e = o; o += TEB;
    c.w = o; o += TEB;
    c.g = o; o += (4 * UPW + 3) & ~3;
    c.cpart = o; o += c.NG * E;
    c.stage = o;
    c.stage_floats = NT * 2 * SG2; if (c.stage_floats < Kn * TEB) c.stage_floats = Kn * TEB;      // doubles as the conv's partial sums
    o += c.stage_floats;
    c.floats = o;
    return c;
}

template <int KNMAX>
__global__ __launch_bounds__(64 * (NCW + NPW)) void dec_fwd_stream(PSF p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_bar;
    __shared__ float s_scale[NCW][32];
    __shared__ float s_red[2][NCW];
    const asr_dec_dims_t& d = p.d;
    const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
    const int cb = slot / p.NT, j = slot - cb * p.NT;
    const int b = cb * 8 + xcd;
    if (b >= d.B) return;
    const unsigned epoch_ = __builtin_amdgcn_readfirstlane(p.status[EPOCH_WORD]);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = p.NT, TEB = p.TEB, A = d.A, E = d.E, Dd = d.Dd, Tp = d.Tp, Kn = d.Kn, Ks = d.Ks, L = d.L;
    const int taps = 2 * Ks + 1, XW = Dd + E;
    const int tau0 = j * TEB;
    const int len = min((int)p.enc_len[b], Tp);
    const int nf = max(0, min(len - tau0, TEB));                        // valid frames of this tile
    const int MTV = (nf + 15) >> 4;                                      // 16-frame MFMA tiles that hold a valid frame
    const FCarve cv_ = fwd_carve(TEB, NT, A, E, Kn, Ks, p.KCP, p.UPW, p.SG2);
    const int WT = cv_.WT, ATP = cv_.ATP, NG = cv_.NG;
    unsigned short* s_cvx = reinterpret_cast<unsigned short*>(smem);                     // [TEB][32] bf16 conv tile of the step, slots {hi | lo | hi}
    const ConvGeo cg_ = conv_geo(TEB, Ks);
    unsigned short* s_ximg = s_cvx + cv_.img;                                           // window image of the attention row, {hi, lo} x 4 shifts
    unsigned short* s_wimg = s_ximg + 8 * cg_.IMG_LD;                                   // filter image {hi, lo} x 16 rows
    float* s_f = reinterpret_cast<float*>(s_cvx + cv_.shorts);
    float* s_x2 = s_f + cv_.x2;                                                         // [2][KCP]  ctx_t | h_{t-1} | 0, by step parity
    float* s_q = s_f + cv_.q;                                                           // [A]
    float* s_wg = s_f + cv_.wg;                                                         // [A]
    float* s_attp = s_f + cv_.attp;                                                     // [Ks + NT*TEB + Ks + 8] zero-padded previous attention
    float* s_epart = s_f + cv_.epart;                                                   // [NCW][TEB] energy partials of the compute waves
    float* s_e = s_f + cv_.e;                                                           // [TEB] masked energies of the tile
    float* s_w = s_f + cv_.w;                                                           // [TEB] exp(e - m)
    float* s_g = s_f + cv_.g;                                                           // [4*UPW]
    float* s_cpart = s_f + cv_.cpart;                                                   // [NG][E] partial context per frame group
    float* s_stage = s_f + cv_.stage;                                                   // [NT][2*SG2]
    if (tid == 0) s_bar = 0u;
    const long region = (long)NT * (p.HG2 + p.QG2 + p.SG2);
    auto xb = [&](int parity) { return p.xbuf + ((long)parity * d.B + b) * region; };
    // every producer clears its own records with the L2-local store flavour before the consensus (decoder_persist.hip)
    for (int parity = 0; parity < 2; ++parity) {
        u64* base = xb(parity);
        for (int i = tid; i < p.HG2; i += blockDim.x) st_gran_local(base + (long)j * p.HG2 + i, 0ull);
        for (int i = tid; i < p.QG2; i += blockDim.x) st_gran_local(base + (long)NT * p.HG2 + (long)j * p.QG2 + i, 0ull);
        for (int i = tid; i < p.SG2; i += blockDim.x) st_gran_local(base + (long)NT * (p.HG2 + p.QG2) + (long)j * p.SG2 + i, 0ull);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const bool local = xcd_consensus(reinterpret_cast<u64*>(p.status) + 64 + b, NT, p.allow_local, p.status);

    // ---- resident data (small: nothing here scales with the tile but LDS rows)
    for (int i = tid; i < TEB * FCVX_LD; i += blockDim.x) s_cvx[i] = 0;
    conv_build_wimg(p.w.Wconv, Kn, taps, cg_, s_wimg, tid, blockDim.x);
    for (int i = tid; i < A; i += blockDim.x) s_wg[i] = p.w.wg[i];
    for (int i = tid; i < 2 * p.KCP; i += blockDim.x) s_x2[i] = 0.f;
    {
        const float uni = 1.f / (float)max(len, 1);
        for (int i = tid; i < ATP; i += blockDim.x) {
            const int tau = i - Ks;
            s_attp[i] = (tau >= 0 && tau < len) ? uni : 0.f;           // initial attention: uniform over the valid frames
        }
    }
    __syncthreads();

    if (wave >= NCW) {
        // =========================== polling role (as in dec_fwd_persist) ===========================
        const int gt = tid - 64 * NCW, np = 64 * NPW;
        for (int t = 0; t < L; ++t) {
            if (t > 0) {
                const u64* src = xb((t - 1) & 1);
                const u64 want = pair_want(seq_of(t - 1), epoch_);
                for (int i0 = gt; 2 * i0 < NT * p.HG2; i0 += np) {
                    u64 lo[1], hi[1];
                    gather16<1>(src + 2 * i0, 0, 1, PAIR_MASK, want, lo, hi, p.status);
                    const int g0 = 2 * i0, prod = g0 / p.HG2, gi = g0 - prod * p.HG2;
                    const int u = prod * p.UPW + 2 * gi;
                    const float v[4] = {lo_f(lo[0]), hi_f(lo[0]), lo_f(hi[0]), hi_f(hi[0])};
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (2 * gi + k < p.UPW && u + k < Dd) s_x2[(t & 1) * p.KCP + E + u + k] = v[k];
                }
            }
            __syncthreads();                                            // B1
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
            __syncthreads();                                            // B2
            poll_copy<10>(xb(t & 1) + (long)NT * (p.HG2 + p.QG2), NT * p.SG2 / 2, s_stage, gt, np, pair_want(seq_of(t), epoch_), p.status);
            __syncthreads();                                            // B3
        }
        return;
    }

    // =========================== compute role ===========================
    unsigned gen = 0;
    const int u_base = j * p.UPW, q_base = j * p.QPW, c_base = j * p.CPW;
    const int SG2f = 2 * p.SG2;
    const float bg = p.w.bg[0];
    float c_state = 0.f;                                                // cell state of unit u_base + tid (waves 0, 1; tid < UPW)
    const int RPW = (4 * p.UPW + NCW - 1) / NCW;                        // gate rows per wave (<= 60)
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
                const int sl = 8 * q + i;
                const int k = sl < KNMAX ? sl : (sl < 2 * KNMAX ? sl - KNMAX : sl - 2 * KNMAX);
                const float w = (ok && sl < 3 * KNMAX && k < Kn) ? wr[min(k, Kn - 1)] : 0.f;
                const __bf16 hi = (__bf16)w;
                wpx[nu][i] = (sl < 2 * KNMAX) ? hi : (__bf16)(w - (float)hi);
            }
            wgu[nu] = ok ? p.w.wg[min(a, A - 1)] : 0.f;
        }
    }
    const int G4 = (NT * TEB) >> 2;                                     // frame groups per utterance in key16t
    const int nch = E >> 3;                                             // 16-byte chunks of an enc row
    DP_DECL

    for (int t = 0; t < L; ++t) {
        int tz = tid, lz_ = lane;
        asm volatile("" : "+v"(tz), "+v"(lz_));                        // ties the index arithmetic to the step (decoder_persist.hip)
        const int lane = lz_;
        const long row = (long)b * L + t;
        float* s_x = s_x2 + (t & 1) * p.KCP;
        const u64 want = pair_want(seq_of(t), epoch_);
        u64* out = xb(t & 1);
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
            const float b0 = p.w.bq[a0], b1 = p.w.bq[a1];
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
        if (tz == 0 && (p.QPW + 1) / 2 < p.QG2) {                       // pad granule of an odd record length
            u64* dst = out + (long)NT * p.HG2 + (long)j * p.QG2 + p.QG2 - 1;
            if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
        }
        DP_MARK(2)
        // ---- location convolution of the tile from the previous attention row (runs while the query is gathered): a Toeplitz
        //      product on the matrix cores (decoder_cluster.h::conv_mfma).  A lane ends up with four kernels of one frame: the
        //      {hi | lo | hi} slots of the sweep's operand row are written straight from the accumulator, so are the values the
        //      backward pass reads (64-byte runs per kernel)
        {
            conv_build_ximg(s_attp + tau0, ATP - tau0, cg_, s_ximg, tz, 64 * NCW);
            compute_barrier(&s_bar, gen);
            conv_mfma<NCW>(s_ximg, s_wimg, cg_, TEB >> 4, wave, lane, [&](int mt, int n, int q4, const f32x4& acc) {
                const int i = 16 * mt + n;
                unsigned short* r = s_cvx + i * FCVX_LD;
#pragma unroll
                for (int pp = 0; pp < 2; ++pp) {
                    const int k0 = 4 * q4 + 2 * pp;
                    if (k0 < Kn) {
                        const float v0 = acc[2 * pp], v1 = acc[2 * pp + 1];
                        const __bf16 h0 = (__bf16)v0, h1 = (__bf16)v1;
                        const __bf16 l0 = (__bf16)(v0 - (float)h0), l1 = (__bf16)(v1 - (float)h1);
                        const unsigned hh = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
                        const unsigned ll = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
                        *reinterpret_cast<unsigned*>(r + k0) = hh;
                        *reinterpret_cast<unsigned*>(r + KNMAX + k0) = ll;
                        *reinterpret_cast<unsigned*>(r + 2 * KNMAX + k0) = hh;
                        if (tau0 + i < Tp) {
                            p.s.conv[(row * Kn + k0) * Tp + tau0 + i] = v0;
                            if (k0 + 1 < Kn) p.s.conv[(row * Kn + k0 + 1) * Tp + tau0 + i] = v1;
                        }
                    }
                }
            });
        }
        DP_MARK(3)
        // key fragments of the first KPF pairs of 16-frame tiles: requested before the barrier (they do not depend on the query),
        // so the sweep starts on data that is already there; inside the sweep a pair's slot is re-requested for pair P + KPF as
        // soon as pair P has been consumed.  The image holds a lane's two quads of a pair side by side, [pair][q][a][2][4]: one
        // 16-byte load per (pair, unit) - 8-byte loads reach 0.54 - 0.70 of the 16-byte rate - and 3 pairs x 3 units x 16 B x 512
        // lanes = 73 KB per CU in flight against the ~2 us of an HBM / Infinity-Cache miss at config 5.
        constexpr int KPF = 3;
        uint4 kr[KPF][FSW_NU];
        int acol[FSW_NU];
        const int q_ = lane >> 4, c_ = lane & 15;
#pragma unroll
        for (int nu = 0; nu < FSW_NU; ++nu) acol[nu] = min(16 * (wave + NCW * nu) + c_, A - 1);
        const int NPR = (TEB + 31) >> 5;                                 // pairs per tile
        const int NPV = (MTV + 1) >> 1;                                  // pairs that hold a valid frame
        const unsigned short* kb0 = p.key16t + ((((long)b * NT + j) * NPR * 4 + q_) * A) * 8;      // + (P*4*A + a) * 8
        if (NPV > 0) {
#pragma unroll
            for (int u = 0; u < KPF; ++u)
#pragma unroll
                for (int nu = 0; nu < FSW_NU; ++nu)
                    kr[u][nu] = *reinterpret_cast<const uint4*>(kb0 + ((long)4 * min(u, NPV - 1) * A + acol[nu]) * 8);
        }
        __syncthreads();                                                // B2: s_q holds q_t, s_cvx the tile's conv
        DP_MARK(4)
        // ---- energies of the tile on the matrix cores, 16 frames at a time
        {
            int opaque = 0;
            asm volatile("" : "+v"(opaque));
            const int q = q_ + opaque, c = c_;
            float qv[FSW_NU];
#pragma unroll
            for (int nu = 0; nu < FSW_NU; ++nu) qv[nu] = s_q[acol[nu]];
            for (int P0 = 0; P0 < NPV; P0 += KPF) {
#pragma unroll
                for (int u = 0; u < KPF; ++u) {
                    const int P = P0 + u;
                    if (P < NPV) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int mt = 2 * P + h;
                            if (mt < MTV) {
                                const bf16x8 av = *reinterpret_cast<const bf16x8*>(s_cvx + (16 * mt + c) * FCVX_LD + 8 * q);
                                float ep[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                for (int nu = 0; nu < FSW_NU; ++nu) {
                                    if (nu < nu_cnt) {
                                        const f32x4 lp = mma16(av, wpx[nu], f32x4{0.f, 0.f, 0.f, 0.f});
                                        const unsigned k0 = h ? kr[u][nu].z : kr[u][nu].x, k1 = h ? kr[u][nu].w : kr[u][nu].y;
                                        const float key[4] = {__uint_as_float(k0 << 16), __uint_as_float(k0 & 0xffff0000u),
                                                              __uint_as_float(k1 << 16), __uint_as_float(k1 & 0xffff0000u)};
#pragma unroll
                                        for (int r = 0; r < 4; ++r) ep[r] += wgu[nu] * tanh_f(key[r] + qv[nu] + tanh_f(lp[r]));
                                    }
                                }
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    float v = ep[r];
#define DPF_STEP(CTRL) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
                                    DPF_STEP(0xB1) DPF_STEP(0x4E) DPF_STEP(0x141) DPF_STEP(0x140)      // sum over the 16 lanes of the row
#undef DPF_STEP
                                    if (c == 0) s_epart[wave * TEB + 16 * mt + 4 * q + r] = v;
                                }
                            }
                        }
                        if (P + KPF < NPV) {
#pragma unroll
                            for (int nu = 0; nu < FSW_NU; ++nu)
                                kr[u][nu] = *reinterpret_cast<const uint4*>(kb0 + ((long)4 * (P + KPF) * A + acol[nu]) * 8);
                        }
                    }
                }
            }
        }
        DP_MARK(5)
        compute_barrier(&s_bar, gen);                                   // c3: s_epart complete
        DP_MARK(6)
        // ---- softmax statistics of the tile: masked energies, (m, s), weights exp(e - m) in LDS
        float m, ssum;
        {
            float mloc = NEG_BIG;
            for (int f = tz; f < TEB; f += 64 * NCW) {
                float sv = 0.f;
#pragma unroll
                for (int w8 = 0; w8 < NCW; ++w8) sv += s_epart[w8 * TEB + f];
                const float ev = (f < nf) ? (sv + bg) / d.temperature : NEG_BIG;
                s_e[f] = ev;
                mloc = fmaxf(mloc, ev);
            }
            mloc = wave_max_dpp(mloc);
            if (lane == 0) s_red[0][wave] = mloc;
            compute_barrier(&s_bar, gen);
            m = s_red[0][0];
#pragma unroll
            for (int w8 = 1; w8 < NCW; ++w8) m = fmaxf(m, s_red[0][w8]);
            float sloc = 0.f;
            for (int f = tz; f < TEB; f += 64 * NCW) {
                const float ev = s_e[f];
                const float wv = (ev > 0.5f * NEG_BIG) ? __expf(ev - m) : 0.f;
                s_w[f] = wv;
                sloc += wv;
            }
            sloc = wave_sum_dpp(sloc);
            if (lane == 0) s_red[1][wave] = sloc;
            compute_barrier(&s_bar, gen);                               // s_w, s_red[1] complete
            ssum = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < NCW; ++w8) ssum += s_red[1][w8];
        }
        DP_MARK(7)
        // ---- partial context of the tile: sum_f w[f] enc[f, :], the enc rows streamed as 16-byte chunks; thread = (chunk,
        //      frame group), eight rows in flight (61 KB per CU), the frame groups meet in LDS
        {
            const int cg = tz % nch, fg = tz / nch;
            if (fg < NG) {
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const unsigned short* er = reinterpret_cast<const unsigned short*>(p.s.enc16) + ((long)b * Tp + tau0) * E + 8 * cg;
                for (int f = fg; f < nf; f += 8 * NG) {
                    uint4 x[8];
                    float wv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int fu = min(f + u * NG, nf - 1);
                        x[u] = *reinterpret_cast<const uint4*>(er + (long)fu * E);
                        wv[u] = (f + u * NG < nf) ? s_w[fu] : 0.f;
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        acc[0] += wv[u] * __uint_as_float(x[u].x << 16); acc[1] += wv[u] * __uint_as_float(x[u].x & 0xffff0000u);
                        acc[2] += wv[u] * __uint_as_float(x[u].y << 16); acc[3] += wv[u] * __uint_as_float(x[u].y & 0xffff0000u);
                        acc[4] += wv[u] * __uint_as_float(x[u].z << 16); acc[5] += wv[u] * __uint_as_float(x[u].z & 0xffff0000u);
                        acc[6] += wv[u] * __uint_as_float(x[u].w << 16); acc[7] += wv[u] * __uint_as_float(x[u].w & 0xffff0000u);
                    }
                }
                float4* o4 = reinterpret_cast<float4*>(s_cpart + fg * E + 8 * cg);
                o4[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                o4[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
            compute_barrier(&s_bar, gen);                               // s_cpart complete
            u64* rec = out + (long)NT * (p.HG2 + p.QG2) + (long)j * p.SG2;
            const int ghead = (TEB + 2) >> 1;                            // granules of e[TEB], m, s
            const int c4 = tz;                                           // four context columns per granule, as halves (pack4h)
            if (4 * c4 < E) {
                float x0 = 0.f, x1 = 0.f, x2 = 0.f, x3 = 0.f;
                for (int g = 0; g < NG; ++g) { const float4 v = *reinterpret_cast<const float4*>(s_cpart + g * E + 4 * c4); x0 += v.x; x1 += v.y; x2 += v.z; x3 += v.w; }
                if (local) publish<true>(rec + ghead + c4, pack4h(x0, x1, x2, x3, want)); else publish<false>(rec + ghead + c4, pack4h(x0, x1, x2, x3, want));
            }
            for (int i = tz; i < (TEB >> 1); i += 64 * NCW) {
                const float2 e2 = *reinterpret_cast<const float2*>(s_e + 2 * i);
                if (local) publish<true>(rec + i, pack2(e2.x, e2.y, want)); else publish<false>(rec + i, pack2(e2.x, e2.y, want));
            }
            if (tz == 64 * NCW - 1) {
                if (local) publish<true>(rec + (TEB >> 1), pack2(m, ssum, want)); else publish<false>(rec + (TEB >> 1), pack2(m, ssum, want));
                if (ghead + E / 4 < p.SG2) {
                    if (local) publish<true>(rec + p.SG2 - 1, pack2(0.f, 0.f, want)); else publish<false>(rec + p.SG2 - 1, pack2(0.f, 0.f, want));
                }
            }
        }
        DP_MARK(8)
        __syncthreads();                                                // B3: s_stage holds every tile's record
        DP_MARK(9)
        // ---- attention row and context of the utterance (online-softmax combine of the tiles)
        {
            float mi = NEG_BIG, si = 0.f;
            if (lane < NT) { mi = s_stage[lane * SG2f + TEB]; si = s_stage[lane * SG2f + TEB + 1]; }
            const float M = wave_max_dpp(mi);
            const float wi = (lane < NT) ? si * __expf(mi - M) : 0.f;
            const float S = fmaxf(wave_sum_dpp(wi), 1e-30f);
            const float invS = 1.f / S;
            if (lane < 32) s_scale[wave][lane] = (lane < NT) ? __expf(mi - M) * invS : 0.f;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            for (int tau = tz; tau < Tp; tau += 64 * NCW) {
                const int i = tau / TEB, f = tau - i * TEB;
                const float ev = s_stage[i * SG2f + f];
                const float av = (ev > 0.5f * NEG_BIG) ? __expf(ev - M) * invS : 0.f;
                s_attp[Ks + tau] = av;
                if (i == j) p.s.att[row * Tp + tau] = av;
            }
            for (int c2_ = tz; 2 * c2_ < E; c2_ += 64 * NCW) {           // two columns per staged word (halves, pack4h)
                float a0 = 0.f, a1 = 0.f;
                for (int i = 0; i < NT; ++i) {
                    const unsigned w_ = __float_as_uint(s_stage[i * SG2f + TEB + 2 + c2_]);
                    a0 += h2f_lo(w_) * s_scale[wave][i]; a1 += h2f_hi(w_) * s_scale[wave][i];
                }
                const int c = 2 * c2_;
                s_x[c] = a0; s_x[c + 1] = a1;
                if (c >= c_base && c < c_base + p.CPW) p.s.xin[row * XW + Dd + c] = a0;
                if (c + 1 >= c_base && c + 1 < c_base + p.CPW) p.s.xin[row * XW + Dd + c + 1] = a1;
            }
        }
        DP_MARK(10)
        compute_barrier(&s_bar, gen);                                   // c5: s_x holds [ctx_t | h_{t-1}]
        DP_MARK(11)
        // ---- LSTM cell: gate rows r = g*UPW + ul of this workgroup, RPW rows per wave, lanes over 16-byte chunks; RBS = 8 rows
        //      per batch of loads (each batch is one exposed L2 round trip: 38 rows per wave at config 5 are 5 batches, not 8)
        {
            constexpr int RBS = 8;
            const int nchunk = p.KCP >> 3;
            float mine = 0.f;
#pragma unroll 1
            for (int bt = 0; bt * RBS < RPW; ++bt) {
                float part[RBS];
#pragma unroll
                for (int rr = 0; rr < RBS; ++rr) part[rr] = 0.f;
                for (int ch0 = lane; ch0 < nchunk; ch0 += 128) {
                    uint4 wv[RBS][2];
#pragma unroll
                    for (int rr = 0; rr < RBS; ++rr) {
                        const int r = min(wave * RPW + bt * RBS + rr, 4 * p.UPW - 1);
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
                            for (int rr = 0; rr < RBS; ++rr) {
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
                for (int rr = 0; rr < RBS; ++rr) {
                    const float sv = wave_sum_dpp(part[rr]);
                    if (lane == bt * RBS + rr) mine = sv;
                }
            }
            const int r = wave * RPW + lane;
            if (lane < RPW && r < 4 * p.UPW) s_g[r] = mine + add_r;
        }
        DP_MARK(12)
        compute_barrier(&s_bar, gen);                                   // c6: s_g holds the gate pre-activations
        DP_MARK(13)
        if (wave < 2) {                                                 // UPW <= 128 units: thread per unit over two waves
            const int ul = tz, unit = u_base + ul;
            const bool uok = ul < p.UPW && unit < Dd;
            float hv = 0.f;
            // the hop starts at the publish: the activations use the fast exp / rcp forms (1 - 2 ulp) and the granule leaves
            // BEFORE the six stores of the step's saved state (one in-order vector-memory queue per wave)
            float ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f;
            if (uok) {
                ai = sigm_f(s_g[ul]); af = sigm_f(s_g[p.UPW + ul]);
                ag = tanh_f(s_g[2 * p.UPW + ul]); ao = sigm_f(s_g[3 * p.UPW + ul]);
                c_state = af * c_state + ai * ag;
                hv = ao * tanh_f(c_state);
            }
            const float hn = __shfl_down(hv, 1);                        // pairs never straddle a wave (64 is even)
            if (t + 1 < L && (ul & 1) == 0 && ul < 2 * p.HG2) {
                u64* dst = out + (long)j * p.HG2 + (ul >> 1);
                if (local) publish<true>(dst, pack2(hv, hn, want)); else publish<false>(dst, pack2(hv, hn, want));
            }
            if (uok) {
                float* go = p.s.gates + row * 4 * Dd;
                go[unit] = ai; go[Dd + unit] = af; go[2 * Dd + unit] = ag; go[3 * Dd + unit] = ao;
                p.s.cs[row * Dd + unit] = c_state;
                p.s.hs[row * Dd + unit] = hv;
            }
        }
    }
    DP_DUMP
}

// key (B, T', A) fp32 -> [B][G4][A][4] bf16, four frames innermost; frames past T' repeat the last row (finite, masked later)
__global__ void build_key16t_kernel(const float* __restrict__ key, unsigned short* __restrict__ out, int B, int Tp, int A, int G4) {
    const long total = (long)B * G4 * A;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const long bg = i / A;
        const int g = (int)(bg % G4), b = (int)(bg / G4);
        unsigned short v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = f2bf_bits(key[((long)b * Tp + min(4 * g + r, Tp - 1)) * A + a]);
        *reinterpret_cast<uint2*>(out + 4 * i) = make_uint2((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16));
    }
}

// forward image of the keys: [B][NT][pairs of 16-frame tiles][q][A][2][4] bf16 - a lane's quads (frames 16 mt + 4 q + r) of the two
// tiles of a pair side by side; frames past T' repeat the last row (finite, masked later)
__global__ void build_key16p_kernel(const float* __restrict__ key, unsigned short* __restrict__ out, int B, int Tp, int A, int NT, int TEB) {
    const int NPR = (TEB + 31) >> 5;
    const long total = (long)B * NT * NPR * 4 * A;                       // one 16-byte element (pair, q, a) per thread
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        long r_ = i / A;
        const int q = (int)(r_ % 4); r_ /= 4;
        const int P = (int)(r_ % NPR); r_ /= NPR;
        const int j = (int)(r_ % NT), b = (int)(r_ / NT);
        unsigned short v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f = j * TEB + 16 * (2 * P + h) + 4 * q + r;
                v[4 * h + r] = f2bf_bits(key[((long)b * Tp + min(f, Tp - 1)) * A + a]);
            }
        *reinterpret_cast<uint4*>(out + 8 * i) = make_uint4((unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                                            (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16));
    }
}

struct StreamPlanF { bool ok; int NT, TEB, UPW, QPW, CPW, HG2, QG2, SG2, KC, KCP; size_t lds, status_bytes, xbuf_bytes, wcat_bytes, emb_bytes, key_bytes, total; };

StreamPlanF stream_plan_f(const asr_dec_dims_t& d) {
    StreamPlanF pl{};
    pl.ok = false;
    if (d.NL != 1 || d.B > 64 || d.B < 1 || d.A > 16 * FSW_NU * NCW || d.Kn > 10 || (d.E & 7) != 0 || d.E > 8 * 64 * NCW || d.Dd > 20 * 32 || d.Tp < 1) return pl;
    const int cpx = cdiv(d.B, 8);
    pl.NT = std::min(30, 32 / cpx);
    pl.TEB = 16 * cdiv(d.Tp, 16 * pl.NT);
    pl.UPW = cdiv(d.Dd, pl.NT); pl.QPW = cdiv(d.A, pl.NT); pl.CPW = cdiv(d.E, pl.NT);
    if (pl.UPW > 120 || cdiv(4 * pl.UPW, NCW) > 60) return pl;
    auto even = [](int x) { return (x + 1) & ~1; };
    pl.HG2 = even((pl.UPW + 1) / 2); pl.QG2 = even((pl.QPW + 1) / 2); pl.SG2 = even((pl.TEB + 2) / 2 + d.E / 4);      // record: e pairs, (m, s), context as halves
    pl.KC = d.E + d.Dd; pl.KCP = (pl.KC + 7) & ~7;
    const FCarve c = fwd_carve(pl.TEB, pl.NT, d.A, d.E, d.Kn, d.Ks, pl.KCP, pl.UPW, pl.SG2);
    if (c.NG < 1) return pl;
    pl.lds = 2 * (size_t)c.shorts + 4 * (size_t)c.floats;
    if (getenv("ASR_DEC_PLAN_DEBUG")) fprintf(stderr, "[asr] streamed fwd plan B=%d T'=%d: NT=%d TEB=%d UPW=%d LDS=%zu\n", d.B, d.Tp, pl.NT, pl.TEB, pl.UPW, pl.lds);
    if (pl.lds > 156 * 1024) return pl;
    pl.status_bytes = 4096;
    pl.xbuf_bytes = align_up256(2 * (size_t)d.B * pl.NT * (pl.HG2 + pl.QG2 + pl.SG2) * sizeof(u64));
    pl.wcat_bytes = align_up256((size_t)4 * d.Dd * pl.KCP * 2);
    pl.emb_bytes = align_up256((size_t)d.B * d.L * 4 * d.Dd * sizeof(float));
    pl.key_bytes = align_up256((size_t)d.B * pl.NT * ((pl.TEB + 31) / 32) * 4 * d.A * 16);      // pair image (odd tile counts are padded)
    pl.total = pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes + pl.emb_bytes + pl.key_bytes;
    pl.ok = true;
    return pl;
}

}  // namespace

size_t dec_fwd_stream_work_bytes(const asr_dec_dims_t& d) {
    const StreamPlanF pl = stream_plan_f(d);
    return pl.ok ? pl.total : 0;
}

// Returns ASR_OK when the whole loop was launched, 1 when the configuration has no streamed plan, negative on error.
int dec_fwd_streamed(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const float* enc,
                     const int64_t* enc_len, void* work, size_t work_bytes, hipStream_t st) {
    const StreamPlanF pl = stream_plan_f(d);
    if (!pl.ok || !work || work_bytes < pl.total || ((uintptr_t)work & 255) != 0 || !s.conv || !s.enc16) return 1;
    char* base = (char*)work;
    unsigned* status = (unsigned*)base;
    u64* xbuf = (u64*)(base + pl.status_bytes);
    unsigned short* wcat16 = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes);
    float* embproj = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes);
    unsigned short* key16t = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes + pl.wcat_bytes + pl.emb_bytes);
    clear_work(work, pl.xbuf_bytes, st);
    hipLaunchKernelGGL(build_wcat16_kernel, dim3(512), dim3(256), 0, st, w.Wih[0], w.Whh[0], wcat16, 4 * d.Dd, d.Dd, d.E, pl.KCP);
    hipLaunchKernelGGL(build_key16p_kernel, dim3(1024), dim3(256), 0, st, s.key, key16t, d.B, d.Tp, d.A, pl.NT, pl.TEB);
    const int XW = d.Dd + d.E;
    int rc = asr_gemm(s.xin, w.Wih[0], embproj, nullptr, d.B * d.L, 4 * d.Dd, d.Dd, XW, XW, 4 * d.Dd, 1, 1, ASR_ACT_NONE, 0, 1, 1, 0, 0, 0,
                      0, 0, ASR_BF16, (asr_stream_t)st);
    if (rc != ASR_OK) return rc;
    static int allow = -1;
    if (allow < 0) { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); allow = (e && e[0] == '0') ? 0 : 1; }
    PSF p{d, w, s, enc, enc_len, wcat16, embproj, key16t, xbuf, status, pl.NT, pl.TEB, pl.UPW, pl.QPW, pl.CPW, pl.HG2, pl.QG2, pl.SG2, pl.KC, pl.KCP, allow};
    const int cpx = cdiv(d.B, 8);
    const dim3 grid(8 * cpx * pl.NT), block(64 * (NCW + NPW));
#define DSF_LAUNCH(KN_)                                                                                                         \
    {                                                                                                                           \
        static unsigned char attr_[32];                                                                                         \
        if (first_on_device(attr_)) hipFuncSetAttribute((const void*)dec_fwd_stream<KN_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); \
        if (!grid_resident(dec_fwd_stream<KN_>, (int)grid.x, (int)block.x, pl.lds)) return 1;                                   \
        hipLaunchKernelGGL((dec_fwd_stream<KN_>), grid, block, pl.lds, st, p);                                                  \
        hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(1), 0, st, status);                                                 \
    }
    if (d.Kn <= 4) DSF_LAUNCH(4) else DSF_LAUNCH(10)
#undef DSF_LAUNCH
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_att_decoder_fwd(streamed): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

// =================================================================================================
// Backward of the teacher-forced decoder loop with streamed tiles: ONE persistent launch, the cluster / record scheme of
// dec_bwd_persist (decoder_persist.hip) with tiles of any size.  What that kernel keeps on chip per tile and what becomes of it here:
//   key tile (LDS)                  -> the bf16 fragment image key16t, read per 48-frame group with the next group's fragment in flight
//   dkey tile (registers, 4 x 9)    -> fp32 read-modify-write of dkT [B][frame / 4][A][4] (16 bytes per lane, the sweep's own layout;
//                                      L2-resident for B <= 16, an HBM stream at config 5), un-transposed into dkey after the launch
//   enc rows of P2 (one pass)       -> passes of nct / 8 frames, the next pass's rows in flight
//   conv tile of the step (48 x Kn) -> one 48-frame group at a time, staged while the previous group's dconv product runs
//   rows of [W_ih(ctx) | W_hh]^T    -> the first RCB ncw + RPB NPB outputs of a workgroup stay register-resident, the rest (B > 16:
//                                      more rows per workgroup than registers) are streamed from L2 as bf16, four rows in flight
//   rows of W_q^T (LDS)             -> streamed from L2 (fp32), three rows in flight
//   dconv rows of the utterance     -> only the window [tau0 - Ks, tau0 + TEB + Ks) of the location filter's reach
// A step costs two workgroup barriers per 48-frame group on top of dec_bwd_persist's; groups past the utterance's length are
// skipped (their dconv is published as zeros).
// =================================================================================================
#include "decoder_bwd_common.h"

namespace {

struct PSB {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const unsigned short* enc16;
    const int64_t* enc_len;
    const float* dhs;               // (B,L,Dd) gradient wrt h_t from the output layer
    float* dxin;                    // (B,L,Dd+E)  context part written here
    float* dq;                      // (B,L,A)     gradient wrt the query pre-activation
    float* dkT;                     // [B][NT*TEB/4][A][4] fp32, zero on entry: dkey in the sweep's layout
    float* slots;                   // (B*NT, slot)  d w_g [a], d W_proj [k][a], d b_g
    float* dgates;                  // (B,L,4Dd)
    const unsigned short* wcatT16;  // (Dd+E+Dd rows = input columns) x R4 bf16
    const float* wqT;               // (Dd x A)
    const unsigned short* key16t;   // [B][NT*TEB/4][A][4] bf16
    u64* xbuf;
    unsigned* status;
    int slot, NT, TEB, UPW, CPW, R4;
    int CG2, QG2, VG2, NG2;
    int allow_local;
    int poll_delay;
};

struct SBCarve { int AP, DW, PADL, WT, parts, dl, wp16, dg16, cvx, cvT, shorts; int wc, crec, qst, nrec, dcp, de, out, hq, pt, dcx, dq, floats; };
__host__ __device__ inline SBCarve sbwd_carve(int TEB, int A, int E, int Kn, int Ks, int NT, int UPW, int CPW, int CG2, int QG2, int NG2) {
    SBCarve c;
    const int nct = 64 * ((A + 63) / 64);
    int ap8 = 8 * ((A + 63) / 64); if ((ap8 & 1) == 0) ++ap8;
    c.AP = 8 * ap8;
    c.PADL = Ks + 8 + ((4 - ((2 * Ks) & 3)) & 3);       // PADL + Ks is a multiple of 4
    c.DW = (c.PADL + TEB + Ks + 8 + 3) & ~3;            // zero-padded dconv window of the tile: frames tau0 - PADL ..
    if (((c.DW >> 2) & 1) == 0) c.DW += 4;
    c.WT = (2 * Ks + 1 + 3) & ~3;
    const int nitem = Kn * (TEB >> 4);                   // (kernel, 16-frame group) items of the transposed convolution
    c.parts = nct / nitem; if (c.parts > 4) c.parts = 4; if (c.parts < 1) c.parts = 1;
    int o = 0;
    c.dl = o; o += 16 * SW_MT * c.AP;
    c.wp16 = o; o += 16 * c.AP;
    c.dg16 = o; o += 64 * KCHB * 4;
    c.cvx = o; o += 16 * SW_MT * CVX_LD;
    c.cvT = o; o += 16 * CVT_LD;
    c.shorts = (o + 7) & ~7;
    o = 0;
    c.wc = o; o += Kn * c.WT;
    c.crec = o; o += NT * CG2 * 2;
    c.qst = o; o += NT * QG2 * 2;
    c.nrec = o; o += NT * NG2 * 2 + 8;
    c.dcp = o; o += Kn * c.DW;
    c.de = o; o += 16 * SW_MT * ((TEB + 16 * SW_MT - 1) / (16 * SW_MT)) + 8;
    const int nout = CPW + UPW;
    c.out = o; o += (nout > RPWB * 8 ? ((nout + 3) & ~3) : RPWB * 8);
    c.hq = o; o += (UPW > 64 ? ((UPW + 3) & ~3) : 64);
    c.pt = o; o += c.parts * Kn * TEB;
    c.dcx = o; o += (E + 3) & ~3;
    c.dq = o; o += (A + 3) & ~3;
    c.floats = o;
    return c;
}

// one 48-frame group of the energy-backward sweep (decoder_bwd_common.h::sweep_step with the key fragment in registers): the dkey
// contribution of a (unit, 16-frame tile) is added to the value read ahead (`old`) and stored at once, and the tile's key
// fragment is re-requested for the NEXT group as soon as it has been consumed - neither lives longer than it must
template <int NU>
__device__ __forceinline__ void sweep_group(Sweep<NU>& S, const float (&qa)[NU], float (&dq)[NU], float4 (&old)[NU][SW_MT], uint2 (&kf)[NU][SW_MT],
                                            float* dk_g, const float* dk_next, const unsigned short* key_next, const int (&acol)[NU], int A, int MTB_left_next,
                                            int nu_cnt, int wave, int nw, int MTg, int AP, const unsigned short* s_cvx, const unsigned short* s_cvT,
                                            const float* s_de, unsigned short* s_dl, int lane) {
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const int q = (lane >> 4) + opaque, c = lane & 15;
#pragma unroll
    for (int mt = 0; mt < SW_MT; ++mt) {
        if (mt < MTg) {
            const int f0 = 16 * mt + 4 * q;
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(s_cvx + (16 * mt + c) * CVX_LD + 8 * q);
            const s16x4_ bv = *reinterpret_cast<const s16x4_*>(s_cvT + c * CVT_LD + f0);
            const float4 de4 = *reinterpret_cast<const float4*>(s_de + f0);
            const float de[4] = {de4.x, de4.y, de4.z, de4.w};
#pragma unroll
            for (int nu = 0; nu < NU; ++nu) {
                if (nu < nu_cnt) {
                    const int a = 16 * (wave + nw * nu) + c;
                    const f32x4 lp = mma16(av, S.wpx[nu], f32x4{0.f, 0.f, 0.f, 0.f});
                    const uint2 kb = kf[nu][mt];
                    const float key[4] = {__uint_as_float(kb.x << 16), __uint_as_float(kb.x & 0xffff0000u),
                                          __uint_as_float(kb.y << 16), __uint_as_float(kb.y & 0xffff0000u)};
                    if (key_next) kf[nu][mt] = *reinterpret_cast<const uint2*>(key_next + ((long)4 * min(mt, MTB_left_next - 1) * A + acol[nu]) * 4);
                    bf16x4 dl;
                    float du4[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float loc = tanh_f(lp[r]);
                        const float u = tanh_f(key[r] + qa[nu] + loc);
                        const float du = de[r] * S.wg[nu] * (1.f - u * u);
                        S.dwg[nu] += de[r] * u;
                        dq[nu] += du;
                        du4[r] = du;
                        dl[r] = (__bf16)(du * (1.f - loc * loc));
                    }
                    {
                        const float4 o = old[nu][mt];
                        if (a < A) *reinterpret_cast<float4*>(dk_g + ((long)4 * mt * A + acol[nu]) * 4) = make_float4(o.x + du4[0], o.y + du4[1], o.z + du4[2], o.w + du4[3]);
                        // the register is free: the NEXT group's value of this (unit, tile) is requested into it (one group of
                        // look-ahead at no extra registers; a second array for it cost 36 registers and spilled)
                        if (dk_next) old[nu][mt] = *reinterpret_cast<const float4*>(dk_next + ((long)4 * min(mt, MTB_left_next - 1) * A + acol[nu]) * 4);
                    }
                    const s16x4_ dls = __builtin_bit_cast(s16x4_, dl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_dl[(f0 + r) * AP + a] = (unsigned short)dls[r];
                    S.dwp[nu] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dls, bv, S.dwp[nu], 0, 0, 0);
                }
            }
        }
    }
}

// All 48-frame groups of the tile for one step (both roles: the polling waves own sweep units too).  COMPUTE waves also stage
// the next group's conv tile and run the dconv product of the group (P4).  Returns the tile's query-gradient partial in dqt.
template <int KNMAX, int NU, bool COMPUTE, bool DKPF>
__device__ __forceinline__ void sweep_tile(const PSB& p, Sweep<NU>& S, const float (&qa)[NU], float (&dqt)[NU], int nu_cnt, int wave, int nw, int lane, int tz,
                                           int nct, int b, int j, int tau0, int nf, int len, long row, int AP,
                                           unsigned short* s_cvx, unsigned short* s_cvT, const float* s_de, unsigned short* s_dl,
                                           const unsigned short* s_wp16, u64* out, long offV, bool local, u64 want) {
    const int A = p.d.A, Kn = p.d.Kn, Tp = p.d.Tp, TEB = p.TEB;
    const long G4 = ((long)p.NT * TEB) >> 2;
    const int MTB = TEB >> 4;
    const int q = lane >> 4, c = lane & 15;
    const int NGRP = (nf + 16 * SW_MT - 1) / (16 * SW_MT);             // groups that hold a valid frame
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) dqt[nu] = 0.f;
    int acol[NU];
#pragma unroll
    for (int nu = 0; nu < NU; ++nu) acol[nu] = min(16 * (wave + nw * nu) + c, A - 1);
    const long g0 = (long)b * G4 + (tau0 >> 2) + q;                      // frame group of (tile, lane row q), + 4 * (3 g + mt)
    uint2 kf[NU][SW_MT];
    if (NGRP > 0) {
#pragma unroll
        for (int nu = 0; nu < NU; ++nu)
#pragma unroll
            for (int mt = 0; mt < SW_MT; ++mt)
                kf[nu][mt] = *reinterpret_cast<const uint2*>(p.key16t + ((g0 + 4 * min(mt, MTB - 1)) * A + acol[nu]) * 4);
    }
    const int cvi0 = tz, cvi1 = tz + nct;                                // conv elements (k, f) of a group staged by this thread
    const int GF = 16 * SW_MT;
    const int cvk0 = min(cvi0, Kn * GF - 1) / GF, cvf0 = min(cvi0, Kn * GF - 1) - cvk0 * GF;
    const int cvk1 = min(cvi1, Kn * GF - 1) / GF, cvf1 = min(cvi1, Kn * GF - 1) - cvk1 * GF;
    // dkey of the group's elements.  DKPF (weight rows streamed, registers to spare): read ahead - the first group's here, every
    // later group's inside the sweep of the group before it, into the registers that sweep has just consumed (sweep_group).
    // Otherwise (register-resident weight rows): read at the group's start into registers that live for that group only.
    float4 old[NU][SW_MT];
    if (DKPF && NGRP > 0) {
#pragma unroll
        for (int nu = 0; nu < NU; ++nu)
#pragma unroll
            for (int mt = 0; mt < SW_MT; ++mt)
                old[nu][mt] = *reinterpret_cast<const float4*>(p.dkT + ((g0 + 4 * min(mt, MTB - 1)) * A + acol[nu]) * 4);
    }
    for (int g = 0; g < NGRP; ++g) {
        const int MTg = min(SW_MT, MTB - SW_MT * g);
        const int fg0 = GF * g;                                          // first frame of the group inside the tile
        float cn0 = 0.f, cn1 = 0.f;
        if (COMPUTE && g + 1 < NGRP) {
            const int fa = tau0 + fg0 + GF + cvf0, fb = tau0 + fg0 + GF + cvf1;
            cn0 = p.s.conv[(row * Kn + cvk0) * Tp + min(fa, Tp - 1)];
            cn1 = p.s.conv[(row * Kn + cvk1) * Tp + min(fb, Tp - 1)];
            if (fa >= Tp) cn0 = 0.f;
            if (fb >= Tp) cn1 = 0.f;
        }
        float dqg[NU];
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) dqg[nu] = 0.f;
        const unsigned short* key_next = (g + 1 < NGRP) ? p.key16t + (g0 + 4 * SW_MT * (g + 1)) * A * 4 : nullptr;
        if constexpr (DKPF) {
            const float* dk_next = (g + 1 < NGRP) ? p.dkT + (g0 + 4 * SW_MT * (g + 1)) * A * 4 : nullptr;
            sweep_group<NU>(S, qa, dqg, old, kf, p.dkT + (g0 + 4 * SW_MT * g) * A * 4, dk_next, key_next, acol, A, MTB - SW_MT * (g + 1),
                            nu_cnt, wave, nw, MTg, AP, s_cvx, s_cvT, s_de + fg0, s_dl, lane);
        } else {
            float4 og[NU][SW_MT];
#pragma unroll
            for (int nu = 0; nu < NU; ++nu)
#pragma unroll
                for (int mt = 0; mt < SW_MT; ++mt)
                    og[nu][mt] = *reinterpret_cast<const float4*>(p.dkT + ((g0 + 4 * min(SW_MT * g + mt, MTB - 1)) * A + acol[nu]) * 4);
            sweep_group<NU>(S, qa, dqg, og, kf, p.dkT + (g0 + 4 * SW_MT * g) * A * 4, nullptr, key_next, acol, A, MTB - SW_MT * (g + 1),
                            nu_cnt, wave, nw, MTg, AP, s_cvx, s_cvT, s_de + fg0, s_dl, lane);
        }
#pragma unroll
        for (int nu = 0; nu < NU; ++nu) dqt[nu] += dqg[nu];
        __syncthreads();                                                // G1: s_dl of the group complete
        if (COMPUTE) {
            // P4: dconv (16 x Kn) = dl (16 x A) . W_proj (A x Kn) on the matrix cores, one 16-frame tile per wave
            if (wave < MTg) {
                f32x4 acc = {0.f, 0.f, 0.f, 0.f};
                const int m = lane & 15, kq = lane >> 4;
                const unsigned short* ar = s_dl + (16 * wave + m) * AP + 8 * kq;
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
                        const int f = fg0 + 16 * wave + 4 * kq + i;
                        const float v0 = (tau0 + f < len) ? acc[i] : 0.f, v1 = (tau0 + f + 1 < len) ? acc[i + 1] : 0.f;
                        if (tau0 + f < Tp) p.s.conv[(row * Kn + m) * Tp + tau0 + f] = v0;           // dconv over the saved conv (d W_conv after the loop)
                        if (tau0 + f + 1 < Tp) p.s.conv[(row * Kn + m) * Tp + tau0 + f + 1] = v1;
                        u64* dst = out + offV + (long)j * p.VG2 + ((m * TEB + f) >> 1);           // record order [k][f]
                        if (local) publish<true>(dst, pack2(v0, v1, want)); else publish<false>(dst, pack2(v0, v1, want));
                    }
                }
            }
            if (g + 1 < NGRP) {
                if (cvi0 < Kn * GF) put_cv<KNMAX>(s_cvx, s_cvT, cvf0, cvk0, cn0);
                if (cvi1 < Kn * GF) put_cv<KNMAX>(s_cvx, s_cvT, cvf1, cvk1, cn1);
            }
        }
        __syncthreads();                                                // G2: the next group's conv tile is staged, s_dl is free
    }
    if (COMPUTE) {
        // frames of the tile past the last valid group: their dconv is zero - for the neighbours' gathers and for d W_conv
        const int fz = GF * NGRP;                                        // multiple of 16
        const int nz = TEB - fz;
        for (int i = tz; i < Kn * (nz >> 1); i += nct) {
            const int k = i / (nz >> 1), f = fz + 2 * (i - k * (nz >> 1));
            if (tau0 + f < Tp) p.s.conv[(row * Kn + k) * Tp + tau0 + f] = 0.f;
            if (tau0 + f + 1 < Tp) p.s.conv[(row * Kn + k) * Tp + tau0 + f + 1] = 0.f;
            u64* dst = out + offV + (long)j * p.VG2 + ((k * TEB + f) >> 1);
            if (local) publish<true>(dst, pack2(0.f, 0.f, want)); else publish<false>(dst, pack2(0.f, 0.f, want));
        }
    }
}

// the rows of the transposed cell weights beyond the register-resident ones: streamed from L2, four rows in flight
#define DSB_P1_EXTRA(WIDX, NWV)                                                                                        \
    for (int oo0_ = RES + (WIDX); oo0_ < nout; oo0_ += 4 * (NWV)) {                                                    \
        uint2 wv_[4][KCHB];                                                                                            \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                             \
            const int oo = oo0_ + (NWV) * r_;                                                                          \
            int x = (oo < p.CPW) ? Dd + min(c_base + oo, E - 1) : XW + min(u_base + (oo - p.CPW), Dd - 1);              \
            if (oo >= nout) x = Dd;                                                                                    \
            _Pragma("unroll") for (int k = 0; k < KCHB; ++k) {                                                         \
                const int col = 4 * (lane + 64 * k);                                                                   \
                const uint2 v = *reinterpret_cast<const uint2*>(p.wcatT16 + (long)x * R4 + min(col, R4 - 4));          \
                wv_[r_][k] = (col < R4) ? v : make_uint2(0u, 0u);                                                      \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int r_ = 0; r_ < 4; ++r_) {                                                             \
            float acc_ = 0.f;                                                                                          \
            _Pragma("unroll") for (int k = 0; k < KCHB; ++k) {                                                         \
                const uint2 g_ = *reinterpret_cast<const uint2*>(s_dg16 + 4 * (lane + 64 * k));                        \
                acc_ = dot2bf(wv_[r_][k].x, g_.x, acc_);                                                               \
                acc_ = dot2bf(wv_[r_][k].y, g_.y, acc_);                                                               \
            }                                                                                                          \
            const float sv_ = wave_sum_dpp(acc_);                                                                      \
            if (lane == 0 && oo0_ + (NWV) * r_ < nout) s_out[oo0_ + (NWV) * r_] = sv_;                                 \
        }                                                                                                              \
    }

// RESIDENT: the workgroup's CPW + UPW rows of the transposed cell weights fit the registers (B <= 16: at most 59 rows); otherwise
// every row is streamed from L2 and the 80 registers go to prefetching (dkey one group ahead)
// RC / RP: register-resident rows per compute / polling wave: (4, 5) = the 33 rows of a B <= 8 plan (NT = 30), (0, 0) = every row
// streamed from L2 (the 8 + 9 rows of dec_bwd_persist would spill 110 registers around the sweep here and lose to streaming).
template <int KNMAX, int RC, int RP>
__global__ __launch_bounds__(512) void dec_bwd_stream(PSB p) {
    constexpr bool RESIDENT = RC > 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned s_bar;
    __shared__ float s_red[8];
    const asr_dec_dims_t& d = p.d;
    const int id = blockIdx.x, xcd = id & 7, slot_id = id >> 3;
    const int cb = slot_id / p.NT, j = slot_id - cb * p.NT;
    const int b = cb * 8 + xcd;
    if (b >= d.B) return;
    const unsigned epoch_ = __builtin_amdgcn_readfirstlane(p.status[EPOCH_WORD]);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = p.NT, TEB = p.TEB, A = d.A, E = d.E, Dd = d.Dd, Tp = d.Tp, Kn = d.Kn, Ks = d.Ks, L = d.L;
    const int ncw = (A + 63) >> 6, nct = 64 * ncw, nw = ncw + NPB;
    const int taps = 2 * Ks + 1, XW = Dd + E, R4 = p.R4;
    const int tau0 = j * TEB;
    const int len = min((int)p.enc_len[b], Tp);
    const int nf = max(0, min(len - tau0, TEB));                        // valid frames of this tile
    const SBCarve cv_ = sbwd_carve(TEB, A, E, Kn, Ks, NT, p.UPW, p.CPW, p.CG2, p.QG2, p.NG2);
    const int AP = cv_.AP, DW = cv_.DW, PADL = cv_.PADL, WT = cv_.WT;
    const int CG2f = 2 * p.CG2, NG2f = 2 * p.NG2, QG2f = 2 * p.QG2;
    const int GF = 16 * SW_MT;
    unsigned short* s_sh = reinterpret_cast<unsigned short*>(smem);
    unsigned short* s_dl = s_sh + cv_.dl;                                                // [48][AP] bf16  d loc pre-activation of the group
    unsigned short* s_wp16 = s_sh + cv_.wp16;                                            // [16][AP] bf16  W_proj^T, zero padded
    unsigned short* s_dg16 = s_sh + cv_.dg16;                                            // [1280] bf16 dgates of the utterance
    unsigned short* s_cvx = s_sh + cv_.cvx;                                              // [48][32] bf16 conv tile of the group, slots {hi | lo | hi}
    unsigned short* s_cvT = s_sh + cv_.cvT;                                              // [16][CVT_LD] the same transposed
    float* s_f = reinterpret_cast<float*>(s_sh + cv_.shorts);
    float* s_wc = s_f + cv_.wc;
    float* s_crec = s_f + cv_.crec;
    float* s_qst = s_f + cv_.qst;
    float* s_nrec = s_f + cv_.nrec;
    float* s_dcp = s_f + cv_.dcp;                                                        // [Kn][DW] dconv window: index PADL + (frame - tau0)
    float* s_de = s_f + cv_.de;                                                          // [TEB rounded up to 48], zero past the valid frames
    float* s_out = s_f + cv_.out;
    float* s_hq = s_f + cv_.hq;
    float* s_pt = s_f + cv_.pt;
    float* s_dcx = s_f + cv_.dcx;
    float* s_dq = s_f + cv_.dq;
    const int NDE = GF * ((TEB + GF - 1) / GF) + 8;
    const long region = (long)NT * (p.CG2 + p.QG2 + p.VG2 + p.NG2);
    auto xb = [&](int parity) { return p.xbuf + ((long)parity * d.B + b) * region; };
    const long offC = 0, offQ = offC + (long)NT * p.CG2, offV = offQ + (long)NT * p.QG2, offN = offV + (long)NT * p.VG2;
    if (tid == 0) s_bar = 0u;
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

    for (int i = tid; i < 16 * AP; i += blockDim.x) { const int k = i / AP, a = i - k * AP; s_wp16[i] = (k < Kn && a < A) ? f2bf_bits(p.w.Wproj[a * Kn + k]) : (unsigned short)0; }
    for (int i = tid; i < Kn * WT; i += blockDim.x) { const int k = i / WT, jj = i - k * WT; s_wc[i] = (jj < taps) ? p.w.Wconv[k * taps + jj] : 0.f; }
    for (int i = tid; i < 64 * KCHB * 4; i += blockDim.x) s_dg16[i] = 0;
    for (int i = tid; i < NT * NG2f + 8; i += blockDim.x) s_nrec[i] = 0.f;
    for (int i = tid; i < NT * CG2f; i += blockDim.x) s_crec[i] = 0.f;
    for (int i = tid; i < Kn * DW; i += blockDim.x) s_dcp[i] = 0.f;
    for (int i = tid; i < GF * AP; i += blockDim.x) s_dl[i] = 0;
    for (int i = tid; i < GF * CVX_LD; i += blockDim.x) s_cvx[i] = 0;
    for (int i = tid; i < 16 * CVT_LD; i += blockDim.x) s_cvT[i] = 0;
    for (int i = tid; i < NDE; i += blockDim.x) s_de[i] = 0.f;
    const int u_base = j * p.UPW, c_base = j * p.CPW;
    const int nout = p.CPW + p.UPW;
    const int RES = RC * ncw + RP * NPB;                                  // register-resident outputs of P1
    const int nunits = (A + 15) >> 4;
    const int nu_cnt = (nunits - wave + nw - 1) / nw;
    const int qsub = lane >> 4, csub = lane & 15;
    // dconv window of this tile in utterance frames, multiples of 4
    const int Ks4 = (Ks + 3) & ~3;
    const int wlo = max(0, tau0 - Ks4), whi = min(NT * TEB, tau0 + TEB + Ks4);
    __syncthreads();

    if (wave >= ncw) {
        // =========================== polling role ===========================
        const int gt = tid - nct, np = 64 * NPB;
        const int obase = RC * ncw + (wave - ncw);
        uint2 wreg[RESIDENT ? RP : 1][KCHB];
        if (RESIDENT) { DPB_WLOAD((RESIDENT ? RP : 1), obase, NPB) }
        Sweep<SW_NUP> S;
        sweep_init<KNMAX>(S, p.w.Wproj, p.w.wg, A, Kn, wave, nw, lane);
        for (int t = L - 1; t >= 0; --t) {
            const int s = L - 1 - t;
            const long row = (long)b * L + t;
            const u64 want = pair_want(seq_of(s), epoch_);
            u64* base = xb(s & 1);
            __syncthreads();                                            // Ba: s_dg16 holds the gate gradients of step t
            if (RESIDENT) { DPB_P1((RESIDENT ? RP : 1), obase, NPB) }
            DSB_P1_EXTRA(wave, nw)
            __syncthreads();                                            // Bb: s_out complete
            poll_copy<4>(base + offC, NT * p.CG2 / 2, s_crec, gt, np, want, p.status);
            float qa[SW_NUP];
#pragma unroll
            for (int nu = 0; nu < SW_NUP; ++nu) qa[nu] = p.s.q[row * A + min(16 * (wave + nw * nu) + csub, A - 1)];
            __syncthreads();                                            // H2
            __syncthreads();                                            // X1: s_de, the first group's conv tile complete
            float dqt[SW_NUP];
            sweep_tile<KNMAX, SW_NUP, false, !RESIDENT>(p, S, qa, dqt, nu_cnt, wave, nw, lane, tid, nct, b, j, tau0, nf, len, row, AP,
                                              s_cvx, s_cvT, s_de, s_dl, s_wp16, base, offV, local, want);
#pragma unroll
            for (int nu = 0; nu < SW_NUP; ++nu) {
                if (nu < nu_cnt) {
                    const int a_ = 16 * (wave + nw * nu) + csub;
                    const float tot_ = sum_groups(dqt[nu]);
                    const float mine_ = (a_ < A) ? tot_ * (1.f - qa[nu] * qa[nu]) : 0.f;
                    const float nb_ = __shfl_down(mine_, 1);
                    if (qsub == 0 && (lane & 1) == 0 && a_ < 2 * p.QG2) {
                        u64* dst_ = base + offQ + (long)j * p.QG2 + (a_ >> 1);
                        if (local) publish<true>(dst_, pack2(mine_, nb_, want)); else publish<false>(dst_, pack2(mine_, nb_, want));
                    }
                }
            }
            for (int z = 0; z < p.poll_delay; ++z) __builtin_amdgcn_s_sleep(127);
            {   // Q records of all tiles (flat copy) and the dconv frames [wlo, whi) of the utterance (all Kn rows), ONE polling sweep
                const int nq = NT * p.QG2 / 2, npk = (whi - wlo) >> 2, nv = Kn * npk;
                for (int i0 = gt; i0 < nq + nv; i0 += 10 * np) {
                    u64 lo[10], hi[10];
                    const u64* addr[10];
                    int cnt = 0;
#pragma unroll
                    for (int k = 0; k < 10; ++k) {
                        const int idx = i0 + k * np;
                        if (idx < nq + nv) cnt = k + 1;
                        const int iv = min(max(idx - nq, 0), nv - 1);
                        const int kk = iv / npk, tau = wlo + 4 * (iv - kk * npk);
                        const int prod = tau / TEB, f = tau - prod * TEB;
                        addr[k] = (idx < nq) ? base + offQ + 2 * (long)idx : base + offV + (long)prod * p.VG2 + ((kk * TEB + f) >> 1);
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
                                const int iv = idx - nq, kk = iv / npk, tau = wlo + 4 * (iv - kk * npk);
                                float* o = s_dcp + kk * DW + PADL + (tau - tau0);
                                o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
                            }
                        }
                }
            }
            __syncthreads();                                            // H3
            if (t > 0) poll_copy<4>(base + offN, NT * p.NG2 / 2, s_nrec, gt, np, want, p.status);
            __syncthreads();                                            // H4
        }
        {   // d w_g and d W_proj of the wave's units -> the workgroup's slot
            float* sl_ = p.slots + ((long)b * NT + j) * p.slot;
#pragma unroll
            for (int nu = 0; nu < SW_NUP; ++nu) {
                if (nu < nu_cnt) {
                    const int u0_ = 16 * (wave + nw * nu);
                    const float g_ = sum_groups(S.dwg[nu]);
                    if (qsub == 0 && u0_ + csub < A) sl_[u0_ + csub] = g_;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int a_ = u0_ + 4 * qsub + r;
                        if (csub < Kn && a_ < A) sl_[A + csub * A + a_] = S.dwp[nu][r];
                    }
                }
            }
        }
        return;
    }

    // =========================== compute role ===========================
    unsigned gen = 0;
    uint2 wreg[RESIDENT ? RC : 1][KCHB];
    if (RESIDENT) { DPB_WLOAD((RESIDENT ? RC : 1), wave, ncw) }
    Sweep<SW_NU> S;
    sweep_init<KNMAX>(S, p.w.Wproj, p.w.wg, A, Kn, wave, nw, lane);
    const int a = tid;
    const bool aok = a < A;
    const bool uok = tid < Dd;
    const int uc = uok ? tid : Dd - 1;
    const int ui = uc / p.UPW, uul = uc - ui * p.UPW;
    const bool uown = uok && ui == j;
    float dbg = 0.f, dc_carry = 0.f;
    float pgi, pgf, pgg, pgo, pct, pcp, pdh;
    {
        const long r0 = (long)b * L + (L - 1);
        const float* g = p.s.gates + r0 * 4 * Dd + uc;
        pgi = g[0]; pgf = g[Dd]; pgg = g[2 * Dd]; pgo = g[3 * Dd];
        pct = p.s.cs[r0 * Dd + uc];
        pcp = (L > 1) ? p.s.cs[(r0 - 1) * Dd + uc] : 0.f;
        pdh = p.dhs[r0 * Dd + uc];
    }
    const int FPP = nct >> 3;                                           // frames per pass of P2 (8 threads per frame)
    const int NATT = (Tp + nct - 1) / nct;                              // attention-row elements per thread (<= 6 by the plan)
    DP_DECL

    for (int t = L - 1; t >= 0; --t) {
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
            if (s > 0) dh += s_crec[ui * CG2f + p.CPW + uul] + s_nrec[ui * NG2f + TEB + uul];
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
        if (RESIDENT) { DPB_P1((RESIDENT ? RC : 1), wave, ncw) }
        DSB_P1_EXTRA(wave, nw)
        DP_MARK(3)
        __syncthreads();                                                // Bb
        DP_MARK(4)
        // ---- C record {dctx slice | dh_rec slice} + global dxin (context part)
        for (int i2 = tz; i2 < p.CG2; i2 += nct) {
            float v[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = 2 * i2 + h;
                const bool ok = (i < p.CPW) ? (c_base + i < E) : (i < nout && u_base + (i - p.CPW) < Dd);
                v[h] = ok ? s_out[min(i, nout - 1)] : 0.f;
                if (i < p.CPW && c_base + i < E) p.dxin[row * XW + Dd + c_base + i] = v[h];
            }
            u64* dst = out + offC + (long)j * p.CG2 + i2;
            if (local) publish<true>(dst, pack2(v[0], v[1], want)); else publish<false>(dst, pack2(v[0], v[1], want));
        }
        // ---- operands of P2/P3 that do not depend on the hand-offs: requested now, used behind H2
        float qa[SW_NU];
#pragma unroll
        for (int nu = 0; nu < SW_NU; ++nu) qa[nu] = p.s.q[row * A + min(16 * (wave + nw * nu) + csub, A - 1)];
        const int f2 = tz >> 3, part = tz & 7;                         // P2: 8 threads per frame
        const unsigned short* erow = p.enc16 + ((long)b * Tp + min(tau0, Tp - 1)) * E;       // (tiles past T' hold no frame: nf = 0, row 0 is only a valid address)
        uint4 x[10];
        {
            const unsigned short* er = erow + (long)min(f2, max(nf - 1, 0)) * E;
#pragma unroll
            for (int u = 0; u < 10; ++u) x[u] = *reinterpret_cast<const uint4*>(er + 8 * min(part + 8 * u, (E >> 3) - 1));
        }
        float ctx2[2], attv[6];
#pragma unroll
        for (int h = 0; h < 2; ++h) ctx2[h] = p.s.xin[row * XW + Dd + min(tz + nct * h, E - 1)];
#pragma unroll
        for (int h = 0; h < 6; ++h) attv[h] = (h < NATT) ? p.s.att[row * Tp + min(tz + nct * h, Tp - 1)] : 0.f;
        // conv tile of the first group (stored to LDS behind H2)
        const int cvi0 = tz, cvi1 = tz + nct;
        const int cvk0 = min(cvi0, Kn * GF - 1) / GF, cvf0 = min(cvi0, Kn * GF - 1) - cvk0 * GF;
        const int cvk1 = min(cvi1, Kn * GF - 1) / GF, cvf1 = min(cvi1, Kn * GF - 1) - cvk1 * GF;
        const float c0 = p.s.conv[(row * Kn + cvk0) * Tp + min(tau0 + cvf0, Tp - 1)];
        const float c1 = p.s.conv[(row * Kn + cvk1) * Tp + min(tau0 + cvf1, Tp - 1)];
        DP_MARK(5)
        __syncthreads();                                                // H2: s_crec holds the C records of all workgroups
        DP_MARK(6)
        // ---- P2: dot over the utterance, dattn of the tile's valid frames, de
        {
            for (int e = tz; e < E; e += nct) { const int i = e / p.CPW; s_dcx[e] = s_crec[i * CG2f + (e - i * p.CPW)]; }
            if (cvi0 < Kn * GF) put_cv<KNMAX>(s_cvx, s_cvT, cvf0, cvk0, (tau0 + cvf0 < Tp) ? c0 : 0.f);
            if (cvi1 < Kn * GF) put_cv<KNMAX>(s_cvx, s_cvT, cvf1, cvk1, (tau0 + cvf1 < Tp) ? c1 : 0.f);
            cbar(&s_bar, gen, ncw);
            float dot = 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h) if (tz + nct * h < E) dot += ctx2[h] * s_dcx[tz + nct * h];
            if (s > 0) {
#pragma unroll
                for (int h = 0; h < 6; ++h) {
                    const int tau = tz + nct * h;
                    if (h < NATT && tau < len) { const int i = tau / TEB; dot += attv[h] * s_nrec[i * NG2f + (tau - i * TEB)]; }
                }
            }
            dot = wave_sum_dpp(dot);
            if (lane == 0) s_red[wave] = dot;
            cbar(&s_bar, gen, ncw);
            dot = 0.f;
            for (int w = 0; w < ncw; ++w) dot += s_red[w];
            for (int fp = 0; fp < nf; fp += FPP) {
                const int f = fp + f2;
                // streamed-from-HBM tiles (ten passes): the next pass's rows are requested before this pass is consumed; small
                // tiles (one or two passes, L2-resident) re-request in place and keep the 40 registers
                uint4 xn[RESIDENT ? 1 : 10];
                if (!RESIDENT) {
                    const unsigned short* er = erow + (long)min(f + FPP, nf - 1) * E;
#pragma unroll
                    for (int u = 0; u < 10; ++u) xn[RESIDENT ? 0 : u] = *reinterpret_cast<const uint4*>(er + 8 * min(part + 8 * u, (E >> 3) - 1));
                } else if (fp > 0) {
                    const unsigned short* er = erow + (long)min(f, nf - 1) * E;
#pragma unroll
                    for (int u = 0; u < 10; ++u) x[u] = *reinterpret_cast<const uint4*>(er + 8 * min(part + 8 * u, (E >> 3) - 1));
                }
                const float attf = p.s.att[row * Tp + min(tau0 + min(f, nf - 1), Tp - 1)];
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
                if (f < nf && part == 0) {
                    const float dat = v + ((s > 0) ? s_nrec[j * NG2f + f] : 0.f);
                    const float dev = attf * (dat - dot) / d.temperature;
                    s_de[f] = dev;
                    dbg += dev;
                }
                if (!RESIDENT) {
#pragma unroll
                    for (int u = 0; u < 10; ++u) x[u] = xn[RESIDENT ? 0 : u];
                }
            }
            for (int f = nf + tz; f < NDE; f += nct) s_de[f] = 0.f;     // frames past the utterance (and what P5 left there)
        }
        __syncthreads();                                                // X1: s_de, s_cvx, s_cvT complete (the polling waves join the sweep)
        DP_MARK(7)
        // ---- P3 / P4: energy backward sweep and dconv of every 48-frame group, then this wave's query-gradient partials (Q record)
        {
            float dqt[SW_NU];
            sweep_tile<KNMAX, SW_NU, true, !RESIDENT>(p, S, qa, dqt, nu_cnt, wave, nw, lane, tz, nct, b, j, tau0, nf, len, row, AP,
                                            s_cvx, s_cvT, s_de, s_dl, s_wp16, out, offV, local, want);
#pragma unroll
            for (int nu = 0; nu < SW_NU; ++nu) {
                if (nu < nu_cnt) {
                    const int a_ = 16 * (wave + nw * nu) + csub;
                    const float tot_ = sum_groups(dqt[nu]);
                    const float mine_ = (a_ < A) ? tot_ * (1.f - qa[nu] * qa[nu]) : 0.f;
                    const float nb_ = __shfl_down(mine_, 1);
                    if (qsub == 0 && (lane & 1) == 0 && a_ < 2 * p.QG2) {
                        u64* dst_ = out + offQ + (long)j * p.QG2 + (a_ >> 1);
                        if (local) publish<true>(dst_, pack2(mine_, nb_, want)); else publish<false>(dst_, pack2(mine_, nb_, want));
                    }
                }
            }
        }
        DP_MARK(8)
        __syncthreads();                                                // H3: dq partials of all workgroups, the dconv window
        DP_MARK(10)
        // ---- P5: dq (sum over tiles), its part of dh_{t-1}, datt_next of the tile
        {
            float dqv = 0.f;
            if (aok) for (int i = 0; i < NT; ++i) dqv += s_qst[i * QG2f + a];
            if (aok) s_dq[a] = dqv;
            const int asl = (A + NT - 1) / NT;
            if (aok && a >= j * asl && a < (j + 1) * asl) p.dq[row * A + a] = dqv;
        }
        if (t > 0) {
            // datt_next[tau'] = sum_k sum_jj W_conv[k][jj] * dconv[k][tau' - jj + Ks] for the tile's frames (window-relative rows)
            // item = (tap range, kernel, group of 16 frames): 64 FMAs per filter read + five window reads.  (With 4 frames per item
            // as in dec_bwd_persist a read fed 5 FMAs, and the 384-frame tile of config 5 was bound by the LDS port.)
            const int ngrp = TEB >> 4;
            const int nitem = Kn * ngrp;
            const int parts = cv_.parts;
            const int gpp = (WT / 4 + parts - 1) / parts;
            for (int it = tz; it < parts * nitem; it += nct) {
                const int pz = it / nitem, o = it - pz * nitem, k = o / ngrp, ig = o - k * ngrp;
                const int g0 = pz * gpp, g1 = min(WT / 4, g0 + gpp);
                const float4* wk4 = reinterpret_cast<const float4*>(s_wc) + (k * WT) / 4;
                const float4* q4 = reinterpret_cast<const float4*>(s_dcp) + (k * DW + PADL + Ks) / 4 + 4 * ig;
                float acc[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 2
                for (int g = g0; g < g1; ++g) {
                    const float4 w4 = wk4[g];
                    float qq[20];
#pragma unroll
                    for (int v = 0; v < 5; ++v) {
                        const float4 t4 = q4[v - 1 - g];
                        qq[4 * v] = t4.x; qq[4 * v + 1] = t4.y; qq[4 * v + 2] = t4.z; qq[4 * v + 3] = t4.w;
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] += w4.x * qq[i + 4] + w4.y * qq[i + 3] + w4.z * qq[i + 2] + w4.w * qq[i + 1];
                }
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    *reinterpret_cast<float4*>(s_pt + (long)(pz * Kn + k) * TEB + 16 * ig + 4 * v) = make_float4(acc[4 * v], acc[4 * v + 1], acc[4 * v + 2], acc[4 * v + 3]);
            }
            cbar(&s_bar, gen, ncw);                                     // s_dq, s_pt complete
            // query part of dh_{t-1}: sum_a dq[a] * W_q[a][unit] for the own units, rows of W_q^T streamed three at a time
            {
                float dqr[5];
#pragma unroll
                for (int k5 = 0; k5 < 5; ++k5) dqr[k5] = (lane + 64 * k5 < A) ? s_dq[min(lane + 64 * k5, A - 1)] : 0.f;
                for (int ul0 = wave; ul0 < p.UPW; ul0 += 3 * ncw) {
                    float wr[3][5];
#pragma unroll
                    for (int r3 = 0; r3 < 3; ++r3) {
                        const int unit = min(u_base + min(ul0 + r3 * ncw, p.UPW - 1), Dd - 1);
#pragma unroll
                        for (int k5 = 0; k5 < 5; ++k5) wr[r3][k5] = p.wqT[(long)unit * A + min(lane + 64 * k5, A - 1)];
                    }
#pragma unroll
                    for (int r3 = 0; r3 < 3; ++r3) {
                        float acc = 0.f;
#pragma unroll
                        for (int k5 = 0; k5 < 5; ++k5) acc += dqr[k5] * wr[r3][k5];
                        acc = wave_sum_dpp(acc);
                        const int ul = ul0 + r3 * ncw;
                        if (lane == 0 && ul < p.UPW) s_hq[ul] = (u_base + ul < Dd) ? acc : 0.f;
                    }
                }
            }
            // datt_next of the tile: the tap-range partial sums, 8 threads per frame (s_de is free after the sweep)
            for (int i0 = 0; i0 < TEB; i0 += FPP) {
                const int i = i0 + (tz >> 3), sub = tz & 7;
                float sv = 0.f;
                if (i < TEB) for (int r = sub; r < parts * Kn; r += 8) sv += s_pt[(long)r * TEB + i];
                sv = sum8_dpp(sv);
                if (i < TEB && sub == 0) s_de[i] = sv;
            }
            cbar(&s_bar, gen, ncw);                                     // s_hq, datt_next complete
            // N record {datt_next tile | dh_q slice}
            for (int i2 = tz; i2 < p.NG2; i2 += nct) {
                float v[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int i = 2 * i2 + h;
                    v[h] = (i < TEB) ? s_de[i] : ((i - TEB < p.UPW) ? s_hq[i - TEB] : 0.f);
                }
                u64* dst = out + offN + (long)j * p.NG2 + i2;
                if (local) publish<true>(dst, pack2(v[0], v[1], want)); else publish<false>(dst, pack2(v[0], v[1], want));
            }
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
    {   // results accumulated on chip: d w_g, d W_proj of the wave's units -> the workgroup's slot
        float* sl_ = p.slots + ((long)b * NT + j) * p.slot;
#pragma unroll
        for (int nu = 0; nu < SW_NU; ++nu) {
            if (nu < nu_cnt) {
                const int u0_ = 16 * (wave + nw * nu);
                const float g_ = sum_groups(S.dwg[nu]);
                if (qsub == 0 && u0_ + csub < A) sl_[u0_ + csub] = g_;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int a_ = u0_ + 4 * qsub + r;
                    if (csub < Kn && a_ < A) sl_[A + csub * A + a_] = S.dwp[nu][r];
                }
            }
        }
    }
    if (dbg != 0.f) atomicAdd(p.slots + ((long)b * NT + j) * p.slot + A * (1 + Kn), dbg);     // slots are zero on entry
}

// dkey (B, T', A) = dkT [B][G4][A][4] un-transposed (plain stores: dkey is written once)
__global__ void dkey_untranspose_kernel(const float* __restrict__ dkT, float* __restrict__ dkey, int B, int Tp, int A, long G4) {
    const long total = (long)B * Tp * A;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const long bt = i / A;
        const int tau = (int)(bt % Tp), b = (int)(bt / Tp);
        dkey[i] = dkT[(((long)b * G4 + (tau >> 2)) * A + a) * 4 + (tau & 3)];
    }
}

struct StreamPlanB { bool ok; int TEB, NT, UPW, CPW, R4, CG2, QG2, VG2, NG2; size_t lds, status_bytes, xbuf_bytes, w16_bytes, dg_bytes, key_bytes, dkt_bytes, total; };

StreamPlanB stream_plan_b(const asr_dec_dims_t& d) {
    StreamPlanB pl{};
    pl.ok = false;
    if (d.NL != 1 || d.B > 64 || d.B < 1 || d.A > 320 || d.A < 16 || d.Kn > 10 || (d.E & 7) != 0 || (d.A & 1) != 0 || d.Dd > 64 * KCHB || d.L < 1 || d.Tp < 1) return pl;
    const int cpx = cdiv(d.B, 8);
    const int ncw = cdiv(d.A, 64), nct = 64 * ncw;
    if (d.Dd > nct || d.E > 2 * nct || d.E > 640 || d.Tp > 6 * nct || d.Kn * 16 * SW_MT > 2 * nct) return pl;
    if (cdiv(d.A, 16) > SW_NU * ncw + SW_NUP * NPB) return pl;           // sweep units over all waves
    pl.R4 = (4 * d.Dd + 7) & ~7;
    pl.NT = std::min(30, 32 / cpx);
    pl.TEB = 16 * cdiv(d.Tp, 16 * pl.NT);
    pl.UPW = cdiv(d.Dd, pl.NT); pl.CPW = cdiv(d.E, pl.NT);
    auto even = [](int x) { return (x + 1) & ~1; };
    pl.CG2 = even((pl.CPW + pl.UPW + 1) / 2); pl.QG2 = even(d.A / 2); pl.VG2 = even((pl.TEB * d.Kn + 1) / 2); pl.NG2 = even((pl.TEB + pl.UPW + 1) / 2);
    const SBCarve cv = sbwd_carve(pl.TEB, d.A, d.E, d.Kn, d.Ks, pl.NT, pl.UPW, pl.CPW, pl.CG2, pl.QG2, pl.NG2);
    pl.lds = 2 * (size_t)cv.shorts + 4 * (size_t)cv.floats;
    if (getenv("ASR_DEC_PLAN_DEBUG")) fprintf(stderr, "[asr] streamed bwd plan B=%d T'=%d: NT=%d TEB=%d UPW=%d LDS=%zu\n", d.B, d.Tp, pl.NT, pl.TEB, pl.UPW, pl.lds);
    if (pl.lds > 160 * 1024 - 4096) return pl;
    pl.status_bytes = 4096;
    pl.xbuf_bytes = align_up256(2 * (size_t)d.B * pl.NT * (pl.CG2 + pl.QG2 + pl.VG2 + pl.NG2) * sizeof(u64));
    pl.w16_bytes = align_up256((size_t)(d.Dd + d.E + d.Dd) * pl.R4 * 2);
    pl.dg_bytes = align_up256((size_t)d.B * d.L * 4 * d.Dd * sizeof(float));
    pl.key_bytes = align_up256((size_t)d.B * pl.NT * pl.TEB * d.A * 2);
    pl.dkt_bytes = align_up256((size_t)d.B * pl.NT * pl.TEB * d.A * 4);
    pl.total = pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes + pl.dg_bytes + pl.key_bytes + pl.dkt_bytes;
    pl.ok = true;
    return pl;
}

}  // namespace

size_t dec_bwd_stream_work_bytes(const asr_dec_dims_t& d) { const StreamPlanB pl = stream_plan_b(d); return pl.ok ? pl.total : 0; }
int dec_bwd_stream_tiles(const asr_dec_dims_t& d) { const StreamPlanB pl = stream_plan_b(d); return pl.ok ? pl.NT : 0; }
float* dec_bwd_stream_dgates(const asr_dec_dims_t& d, void* work) {
    const StreamPlanB pl = stream_plan_b(d);
    return (float*)((char*)work + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes);
}

// Same contract as dec_bwd_persistent (decoder_persist.hip).
int dec_bwd_streamed(const asr_dec_dims_t& d, const asr_dec_weights_t& w, const asr_dec_state_t& s, const int64_t* enc_len,
                     const float* dhs, float* dxin, float* dq, float* dkey, float* slots, int slot, const float* wcatT, const float* wqT,
                     void* work, size_t work_bytes, float** dgates_out, hipStream_t st) {
    const StreamPlanB pl = stream_plan_b(d);
    if (!pl.ok || !work || work_bytes < pl.total || ((uintptr_t)work & 255) != 0 || !s.conv || !s.enc16) return 1;
    char* base = (char*)work;
    unsigned* status = (unsigned*)base;
    u64* xbuf = (u64*)(base + pl.status_bytes);
    unsigned short* w16 = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes);
    float* dgates = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes);
    unsigned short* key16t = (unsigned short*)(base + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes + pl.dg_bytes);
    float* dkT = (float*)(base + pl.status_bytes + pl.xbuf_bytes + pl.w16_bytes + pl.dg_bytes + pl.key_bytes);
    *dgates_out = dgates;
    clear_work(work, pl.xbuf_bytes, st);
    hipMemsetAsync(dkT, 0, pl.dkt_bytes, st);
    hipLaunchKernelGGL(cast_rows_bf16_kernel, dim3(512), dim3(256), 0, st, wcatT, w16, d.Dd + d.E + d.Dd, 4 * d.Dd, pl.R4);
    hipLaunchKernelGGL(build_key16t_kernel, dim3(1024), dim3(256), 0, st, s.key, key16t, d.B, d.Tp, d.A, pl.NT * pl.TEB / 4);
    static int allow = -1, delay = -1;
    if (allow < 0) { const char* e = getenv("ASR_LSTM_XCD_LOCAL"); allow = (e && e[0] == '0') ? 0 : 1; }
    if (delay < 0) { const char* e = getenv("ASR_DEC_BWD_POLL_DELAY"); delay = e ? atoi(e) : 0; }
    PSB p{d, w, s, (const unsigned short*)s.enc16, enc_len, dhs, dxin, dq, dkT, slots, dgates, w16, wqT, key16t, xbuf, status,
          slot, pl.NT, pl.TEB, pl.UPW, pl.CPW, pl.R4, pl.CG2, pl.QG2, pl.VG2, pl.NG2, allow, delay};
    const int cpx = cdiv(d.B, 8), ncw = cdiv(d.A, 64);
    const dim3 grid(8 * cpx * pl.NT), block(64 * (ncw + NPB));
#define DSB_LAUNCH(KN_, RC_, RP_)                                                                                               \
    {                                                                                                                           \
        static unsigned char attr_[32];                                                                                         \
        if (first_on_device(attr_)) hipFuncSetAttribute((const void*)dec_bwd_stream<KN_, RC_, RP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 4096); \
        if (!grid_resident(dec_bwd_stream<KN_, RC_, RP_>, (int)grid.x, (int)block.x, pl.lds)) return 1;                         \
        hipLaunchKernelGGL((dec_bwd_stream<KN_, RC_, RP_>), grid, block, pl.lds, st, p);                                        \
        hipLaunchKernelGGL(bump_epoch_kernel, dim3(1), dim3(1), 0, st, status);                                                 \
    }
    const int nrows = pl.CPW + pl.UPW;
    // measured (tools/diag_dec_stream.py, us per step): B=8 x T'=1225 half-resident 29.8 / all rows streamed 30.1 / the full resident
    // set of dec_bwd_persist (8 + 9 rows per wave, 110 spilled registers here) 34.5; B=16 x T'=1225 streamed 35.2 / full set 37.8
    int res = (nrows <= 4 * ncw + 5 * NPB) ? 1 : 0;
    if (const char* e = getenv("ASR_DEC_STREAM_RES")) { if (atoi(e) == 0) res = 0; }        // experiments: stream every row
    if (d.Kn <= 4) { if (res == 1) DSB_LAUNCH(4, 4, 5) else DSB_LAUNCH(4, 0, 0) }
    else { if (res == 1) DSB_LAUNCH(10, 4, 5) else DSB_LAUNCH(10, 0, 0) }
#undef DSB_LAUNCH
    hipLaunchKernelGGL(dkey_untranspose_kernel, dim3(2048), dim3(256), 0, st, dkT, dkey, d.B, d.Tp, d.A, (long)pl.NT * pl.TEB / 4);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_att_decoder_bwd(streamed): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}
