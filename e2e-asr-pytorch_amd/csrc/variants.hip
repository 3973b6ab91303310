// Model variants outside the shipped configs: the pieces the YAML surface accepts beside the
// LSTM / location-aware / single-head path that the persistent kernels cover -
//   * scaled dot-product and multi-head attention, value projection  (reference src/module.py:1084-1132, src/asr.py:299-355)
//   * location-aware attention with num_head > 1                     (src/module.py:1135-1189)
//   * GRU cells: nn.GRU as the encoder recurrence (src/module.py:1023) and as the decoder cell (src/asr.py:203-204)
//   * the LSTM cell with its backward (multi-layer decoders with inter-layer dropout, src/asr.py:204,262-268)
// Every entry point is one small kernel; the host composes them step by step (src/variants.py) exactly as the reference's Python
// loop does.  They are written for correctness first: a variant step is bound by its launches, not by these kernels.
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += s_red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* s_red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[w] = v;
    __syncthreads();
    float t = -INFINITY;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, s_red[i]);
    return t;
}

// attn[r, t] = softmax_t(energy[r, t] / temperature) over t < len[r / NH], 0 beyond (BaseAttention._attend, src/module.py:1110-1118)
__global__ __launch_bounds__(256) void masked_softmax_fwd_kernel(const float* __restrict__ e, const int64_t* __restrict__ len, int NH, int T,
                                                                float inv_temp, float* __restrict__ attn) {
    __shared__ float s_red[4];
    const int r = blockIdx.x;
    const int L = (int)min((long)len[r / NH], (long)T);
    const float* er = e + (long)r * T;
    float m = -INFINITY;
    for (int t = threadIdx.x; t < L; t += blockDim.x) m = fmaxf(m, er[t] * inv_temp);
    m = block_max(m, s_red);
    float s = 0.f;
    for (int t = threadIdx.x; t < L; t += blockDim.x) s += expf(er[t] * inv_temp - m);
    s = block_sum(s, s_red);
    const float inv = (s > 0.f) ? 1.f / s : 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) attn[(long)r * T + t] = (t < L) ? expf(er[t] * inv_temp - m) * inv : 0.f;
}

// de = attn * (dattn - sum_s attn dattn) / temperature
__global__ __launch_bounds__(256) void masked_softmax_bwd_kernel(const float* __restrict__ attn, const float* __restrict__ dattn, int T,
                                                                float inv_temp, float* __restrict__ de) {
    __shared__ float s_red[4];
    const int r = blockIdx.x;
    const float* a = attn + (long)r * T;
    const float* g = dattn + (long)r * T;
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) s += a[t] * g[t];
    s = block_sum(s, s_red);
    for (int t = threadIdx.x; t < T; t += blockDim.x) de[(long)r * T + t] = a[t] * (g[t] - s) * inv_temp;
}

// energy[r, t] = sum_d wg[d] tanh(key[r, t, d] + q[r, d] + tanh(loc_pre[r / NH, t, d])) + bg      (src/module.py:1176-1182)
// one wave per (r, t)
__global__ __launch_bounds__(256) void loc_energy_fwd_kernel(const float* __restrict__ key, const float* __restrict__ q, const float* __restrict__ loc_pre,
                                                            const float* __restrict__ wg, const float* __restrict__ bg, int NH, int T, int D,
                                                            float* __restrict__ energy) {
    const int r = blockIdx.y, t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= T) return;
    const int b = r / NH;
    const float* k_ = key + ((long)r * T + t) * D;
    const float* l_ = loc_pre + ((long)b * T + t) * D;
    const float* q_ = q + (long)r * D;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) acc += wg[d] * tanhf(k_[d] + q_[d] + tanhf(l_[d]));
    acc = wave_sum(acc);
    if (lane == 0) energy[(long)r * T + t] = acc + bg[0];
}

// backward of the above for a block of (b, 32 frames): all heads of the utterance, threads over d.
// dkey is accumulated in place (the key is the same tensor at every decoder step); dq / dwg / dbg are added atomically
// (zeroed by the caller / parameter gradients); dloc_pre is written (the sum over the heads).
constexpr int LE_TCH = 32;
__global__ __launch_bounds__(256) void loc_energy_bwd_kernel(const float* __restrict__ key, const float* __restrict__ q, const float* __restrict__ loc_pre,
                                                            const float* __restrict__ wg, const float* __restrict__ de, int NH, int T, int D,
                                                            float* __restrict__ dkey, float* __restrict__ dq, float* __restrict__ dloc_pre,
                                                            float* __restrict__ dwg, float* __restrict__ dbg) {
    __shared__ float s_red[4];
    const int b = blockIdx.y, t0 = blockIdx.x * LE_TCH, t1 = min(T, t0 + LE_TCH);
    float gsum = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        const float w = wg[d];
        float dw = 0.f;
        for (int h = 0; h < NH; ++h) {
            const int r = b * NH + h;
            const float qd = q[(long)r * D + d];
            float dqa = 0.f;
            for (int t = t0; t < t1; ++t) {
                const float v = tanhf(loc_pre[((long)b * T + t) * D + d]);
                const float u = tanhf(key[((long)r * T + t) * D + d] + qd + v);
                const float g = de[(long)r * T + t];
                const float du = g * w * (1.f - u * u);
                dkey[((long)r * T + t) * D + d] += du;
                dqa += du;
                dw += g * u;
                const float dl = du * (1.f - v * v);
                float* dst = dloc_pre + ((long)b * T + t) * D + d;
                if (h == 0) *dst = dl; else *dst += dl;
            }
            atomicAdd(dq + (long)r * D + d, dqa);
        }
        atomicAdd(dwg + d, dw);
    }
    for (int h = 0; h < NH; ++h)
        for (int t = t0 + threadIdx.x; t < t1; t += blockDim.x) gsum += de[((long)b * NH + h) * T + t];
    gsum = block_sum(gsum, s_red);
    if (threadIdx.x == 0) atomicAdd(dbg, gsum);
}

// out[b, t, k] = sum_h sum_j W[k, h, j] prev[b, h, t + j - Ks]       (nn.Conv1d(NH, Kn, 2 Ks + 1, padding Ks, bias False), then transpose)
__global__ void loc_conv_fwd_kernel(const float* __restrict__ prev, const float* __restrict__ W, int B, int NH, int T, int Kn, int Ks,
                                    float* __restrict__ out) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * T * Kn) return;
    const int k = (int)(i % Kn), t = (int)((i / Kn) % T), b = (int)(i / ((long)Kn * T));
    const int taps = 2 * Ks + 1;
    float acc = 0.f;
    for (int h = 0; h < NH; ++h) {
        const float* p = prev + ((long)b * NH + h) * T;
        const float* w = W + ((long)k * NH + h) * taps;
        const int j0 = max(0, Ks - t), j1 = min(taps, T + Ks - t);
        for (int j = j0; j < j1; ++j) acc += w[j] * p[t + j - Ks];
    }
    out[i] = acc;
}
// dprev[b, h, t'] = sum_k sum_j W[k, h, j] dout[b, t' - j + Ks, k]
__global__ void loc_conv_bwd_data_kernel(const float* __restrict__ dout, const float* __restrict__ W, int B, int NH, int T, int Kn, int Ks,
                                         float* __restrict__ dprev) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)B * NH * T) return;
    const int tp = (int)(i % T), h = (int)((i / T) % NH), b = (int)(i / ((long)T * NH));
    const int taps = 2 * Ks + 1;
    float acc = 0.f;
    for (int k = 0; k < Kn; ++k) {
        const float* w = W + ((long)k * NH + h) * taps;
        const int j0 = max(0, tp + Ks - (T - 1)), j1 = min(taps, tp + Ks + 1);
        for (int j = j0; j < j1; ++j) acc += w[j] * dout[((long)b * T + (tp - j + Ks)) * Kn + k];
    }
    dprev[i] = acc;
}
// dW[k, h, j] += sum_b sum_t dout[b, t, k] prev[b, h, t + j - Ks]; one wave per filter element
__global__ __launch_bounds__(64) void loc_conv_bwd_weight_kernel(const float* __restrict__ dout, const float* __restrict__ prev, int B, int NH, int T,
                                                                 int Kn, int Ks, float* __restrict__ dW) {
    const int taps = 2 * Ks + 1;
    const int i = blockIdx.x;
    const int j = i % taps, h = (i / taps) % NH, k = i / (taps * NH);
    float acc = 0.f;
    for (int b = 0; b < B; ++b) {
        const int ta = max(0, Ks - j), tb = min(T, T + Ks - j);
        for (int t = ta + threadIdx.x; t < tb; t += 64) acc += dout[((long)b * T + t) * Kn + k] * prev[((long)b * NH + h) * T + t + j - Ks];
    }
    acc = wave_sum(acc);
    if (threadIdx.x == 0) dW[i] += acc;
}

// nn.LSTM cell on the two pre-activation halves (gate order i, f, g, o; biases already inside gx / gh)
__global__ void lstm_cell_train_fwd_kernel(const float* __restrict__ gx, const float* __restrict__ gh, const float* __restrict__ c_prev, int N, int D,
                                           float* __restrict__ act, float* __restrict__ h, float* __restrict__ c) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)N * D) return;
    const int n = (int)(i / D), d = (int)(i % D);
    const long g0 = (long)n * 4 * D + d;
    const float gi = sigmoidf_(gx[g0] + gh[g0]), gf = sigmoidf_(gx[g0 + D] + gh[g0 + D]);
    const float gg = tanhf(gx[g0 + 2 * D] + gh[g0 + 2 * D]), go = sigmoidf_(gx[g0 + 3 * D] + gh[g0 + 3 * D]);
    const float cp = c_prev ? c_prev[i] : 0.f;
    const float cn = gf * cp + gi * gg;
    act[g0] = gi; act[g0 + D] = gf; act[g0 + 2 * D] = gg; act[g0 + 3 * D] = go;
    c[i] = cn;
    h[i] = go * tanhf(cn);
}
__global__ void lstm_cell_train_bwd_kernel(const float* __restrict__ act, const float* __restrict__ c_prev, const float* __restrict__ c,
                                           const float* __restrict__ dh, const float* __restrict__ dc, int N, int D,
                                           float* __restrict__ dg, float* __restrict__ dc_prev) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)N * D) return;
    const int n = (int)(i / D), d = (int)(i % D);
    const long g0 = (long)n * 4 * D + d;
    const float gi = act[g0], gf = act[g0 + D], gg = act[g0 + 2 * D], go = act[g0 + 3 * D];
    const float tc = tanhf(c[i]);
    const float dhv = dh ? dh[i] : 0.f;
    const float dcn = (dc ? dc[i] : 0.f) + dhv * go * (1.f - tc * tc);
    const float cp = c_prev ? c_prev[i] : 0.f;
    dg[g0] = dcn * gg * gi * (1.f - gi);
    dg[g0 + D] = dcn * cp * gf * (1.f - gf);
    dg[g0 + 2 * D] = dcn * gi * (1.f - gg * gg);
    dg[g0 + 3 * D] = dhv * tc * go * (1.f - go);
    dc_prev[i] = dcn * gf;
}

// nn.GRU cell: r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h = (1 - z) n + z h_prev; saved: r | z | n | gh_n
__global__ void gru_cell_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh, const float* __restrict__ h_prev, int N, int D,
                                    float* __restrict__ saved, float* __restrict__ h) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)N * D) return;
    const int n_ = (int)(i / D), d = (int)(i % D);
    const long g0 = (long)n_ * 3 * D + d, s0 = (long)n_ * 4 * D + d;
    const float r = sigmoidf_(gi[g0] + gh[g0]), z = sigmoidf_(gi[g0 + D] + gh[g0 + D]);
    const float ghn = gh[g0 + 2 * D];
    const float nn_ = tanhf(gi[g0 + 2 * D] + r * ghn);
    const float hp = h_prev ? h_prev[i] : 0.f;
    saved[s0] = r; saved[s0 + D] = z; saved[s0 + 2 * D] = nn_; saved[s0 + 3 * D] = ghn;
    h[i] = (1.f - z) * nn_ + z * hp;
}
__global__ void gru_cell_bwd_kernel(const float* __restrict__ saved, const float* __restrict__ h_prev, const float* __restrict__ dh, int N, int D,
                                    float* __restrict__ dgi, float* __restrict__ dgh, float* __restrict__ dh_prev) {
    const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i >= (long)N * D) return;
    const int n_ = (int)(i / D), d = (int)(i % D);
    const long g0 = (long)n_ * 3 * D + d, s0 = (long)n_ * 4 * D + d;
    const float r = saved[s0], z = saved[s0 + D], nn_ = saved[s0 + 2 * D], ghn = saved[s0 + 3 * D];
    const float hp = h_prev ? h_prev[i] : 0.f;
    const float g = dh[i];
    const float dpn = g * (1.f - z) * (1.f - nn_ * nn_);
    const float dpz = g * (hp - nn_) * z * (1.f - z);
    const float dpr = dpn * ghn * r * (1.f - r);
    dgi[g0] = dpr; dgi[g0 + D] = dpz; dgi[g0 + 2 * D] = dpn;
    dgh[g0] = dpr; dgh[g0 + D] = dpz; dgh[g0 + 2 * D] = dpn * r;
    dh_prev[i] = g * z;
}

// ---- nn.GRU over a whole (padded) sequence: one workgroup per (batch row, direction), h in LDS, W_hh^T streamed from L2 ----------
// gi (B, T, ND, 3H) = x W_ih^T + b_ih (asr_gemm); whhT (ND, H, 3H) = weight_hh transposed; bhh (ND, 3H)
// y (B, T, ND*H); saved (B, T, ND, 4H) = r | z | n | gh_n (bias included)
__global__ __launch_bounds__(1024) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whhT, const float* __restrict__ bhh,
                                                          int T, int Hd, int ND, float* __restrict__ y, float* __restrict__ saved) {
    extern __shared__ float s_h[];            // [H] h_{t-1} | [3H] gh
    float* s_gh = s_h + Hd;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int G = 3 * Hd;
    const float* W = whhT + (long)dir * Hd * G;
    for (int k = tid; k < Hd; k += blockDim.x) s_h[k] = 0.f;
    __syncthreads();
    for (int s = 0; s < T; ++s) {
        const int t = dir ? (T - 1 - s) : s;
        for (int row = tid; row < G; row += blockDim.x) {
            float acc = bhh[dir * G + row];
            for (int k = 0; k < Hd; ++k) acc += W[(long)k * G + row] * s_h[k];
            s_gh[row] = acc;
        }
        __syncthreads();
        const float* gi_ = gi + (((long)b * T + t) * ND + dir) * G;
        float* sv = saved + (((long)b * T + t) * ND + dir) * 4 * Hd;
        float hn[2];
        for (int k = tid, it = 0; k < Hd; k += blockDim.x, ++it) {
            const float r = sigmoidf_(gi_[k] + s_gh[k]), z = sigmoidf_(gi_[Hd + k] + s_gh[Hd + k]);
            const float ghn = s_gh[2 * Hd + k];
            const float nn_ = tanhf(gi_[2 * Hd + k] + r * ghn);
            sv[k] = r; sv[Hd + k] = z; sv[2 * Hd + k] = nn_; sv[3 * Hd + k] = ghn;
            const float hv = (1.f - z) * nn_ + z * s_h[k];
            y[((long)b * T + t) * ND * Hd + dir * Hd + k] = hv;
            if (it < 2) hn[it] = hv;
        }
        __syncthreads();
        for (int k = tid, it = 0; k < Hd; k += blockDim.x, ++it) s_h[k] = hn[it];      // Hd <= 2 * blockDim.x (checked by the host)
        __syncthreads();
    }
}

// backward: dy (B, T, ND*H) -> dgi, dgh (B, T, ND, 3H); whh (ND, 3H, H) row-major (dh_{t-1}[k] = sum_row dgh[row] whh[row][k])
__global__ __launch_bounds__(1024) void gru_seq_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ saved,
                                                          const float* __restrict__ whh, int T, int Hd, int ND,
                                                          float* __restrict__ dgi, float* __restrict__ dgh) {
    extern __shared__ float s_m[];            // [H] dh carry | [3H] dgh of the step
    float* s_dg = s_m + Hd;
    const int b = blockIdx.x, dir = blockIdx.y, tid = threadIdx.x;
    const int G = 3 * Hd;
    const float* W = whh + (long)dir * G * Hd;
    for (int k = tid; k < Hd; k += blockDim.x) s_m[k] = 0.f;
    __syncthreads();
    for (int s = T - 1; s >= 0; --s) {
        const int t = dir ? (T - 1 - s) : s;                 // forward step s visited frame t; its predecessor is t -/+ 1
        const int tprev = dir ? t + 1 : t - 1;
        const bool has_prev = (s > 0);
        const float* sv = saved + (((long)b * T + t) * ND + dir) * 4 * Hd;
        float* gi_ = dgi + (((long)b * T + t) * ND + dir) * G;
        float* gh_ = dgh + (((long)b * T + t) * ND + dir) * G;
        float carry[2];
        for (int k = tid, it = 0; k < Hd; k += blockDim.x, ++it) {
            const float r = sv[k], z = sv[Hd + k], nn_ = sv[2 * Hd + k], ghn = sv[3 * Hd + k];
            const float hp = has_prev ? y[((long)b * T + tprev) * ND * Hd + dir * Hd + k] : 0.f;
            const float g = dy[((long)b * T + t) * ND * Hd + dir * Hd + k] + s_m[k];
            const float dpn = g * (1.f - z) * (1.f - nn_ * nn_);
            const float dpz = g * (hp - nn_) * z * (1.f - z);
            const float dpr = dpn * ghn * r * (1.f - r);
            gi_[k] = dpr; gi_[Hd + k] = dpz; gi_[2 * Hd + k] = dpn;
            gh_[k] = dpr; gh_[Hd + k] = dpz; gh_[2 * Hd + k] = dpn * r;
            s_dg[k] = dpr; s_dg[Hd + k] = dpz; s_dg[2 * Hd + k] = dpn * r;
            if (it < 2) carry[it] = g * z;
        }
        __syncthreads();
        for (int k = tid, it = 0; k < Hd; k += blockDim.x, ++it) {
            float acc = carry[it];
            for (int row = 0; row < G; ++row) acc += s_dg[row] * W[(long)row * Hd + k];
            carry[it] = acc;
        }
        __syncthreads();
        for (int k = tid, it = 0; k < Hd; k += blockDim.x, ++it) s_m[k] = carry[it];
        __syncthreads();
    }
}

}  // namespace

extern "C" int asr_masked_softmax_fwd(const float* energy, const int64_t* len, int rows, int NH, int T, float temperature, float* attn,
                                      asr_stream_t stream) {
    ASR_REQUIRE(energy && len && attn && rows > 0 && NH > 0 && T > 0 && temperature > 0.f && rows % NH == 0, ASR_E_ARG, "asr_masked_softmax_fwd: bad args");
    hipLaunchKernelGGL(masked_softmax_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, energy, len, NH, T, 1.f / temperature, attn);
    ASR_LAUNCH_CHECK("asr_masked_softmax_fwd");
    return ASR_OK;
}
extern "C" int asr_masked_softmax_bwd(const float* attn, const float* dattn, int rows, int T, float temperature, float* denergy,
                                      asr_stream_t stream) {
    ASR_REQUIRE(attn && dattn && denergy && rows > 0 && T > 0 && temperature > 0.f, ASR_E_ARG, "asr_masked_softmax_bwd: bad args");
    hipLaunchKernelGGL(masked_softmax_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, attn, dattn, T, 1.f / temperature, denergy);
    ASR_LAUNCH_CHECK("asr_masked_softmax_bwd");
    return ASR_OK;
}
extern "C" int asr_loc_energy_fwd(const float* key, const float* q, const float* loc_pre, const float* wg, const float* bg,
                                  int B, int NH, int T, int D, float* energy, asr_stream_t stream) {
    ASR_REQUIRE(key && q && loc_pre && wg && bg && energy && B > 0 && NH > 0 && T > 0 && D > 0, ASR_E_ARG, "asr_loc_energy_fwd: bad args");
    hipLaunchKernelGGL(loc_energy_fwd_kernel, dim3(cdiv(T, 4), B * NH), dim3(256), 0, (hipStream_t)stream, key, q, loc_pre, wg, bg, NH, T, D, energy);
    ASR_LAUNCH_CHECK("asr_loc_energy_fwd");
    return ASR_OK;
}
extern "C" int asr_loc_energy_bwd(const float* key, const float* q, const float* loc_pre, const float* wg, const float* denergy,
                                  int B, int NH, int T, int D, float* dkey, float* dq, float* dloc_pre, float* dwg, float* dbg,
                                  asr_stream_t stream) {
    ASR_REQUIRE(key && q && loc_pre && wg && denergy && dkey && dq && dloc_pre && dwg && dbg && B > 0 && NH > 0 && T > 0 && D > 0, ASR_E_ARG,
                "asr_loc_energy_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dq, 0, sizeof(float) * (size_t)B * NH * D, st) != hipSuccess) { asr_set_error("asr_loc_energy_bwd: memset failed"); return ASR_E_LAUNCH; }
    hipLaunchKernelGGL(loc_energy_bwd_kernel, dim3(cdiv(T, LE_TCH), B), dim3(256), 0, st, key, q, loc_pre, wg, denergy, NH, T, D, dkey, dq, dloc_pre, dwg, dbg);
    ASR_LAUNCH_CHECK("asr_loc_energy_bwd");
    return ASR_OK;
}
extern "C" int asr_loc_conv_fwd(const float* prev_att, const float* W, int B, int NH, int T, int Kn, int Ks, float* out, asr_stream_t stream) {
    ASR_REQUIRE(prev_att && W && out && B > 0 && NH > 0 && T > 0 && Kn > 0 && Ks >= 0, ASR_E_ARG, "asr_loc_conv_fwd: bad args");
    hipLaunchKernelGGL(loc_conv_fwd_kernel, dim3(cdiv((long)B * T * Kn, 256)), dim3(256), 0, (hipStream_t)stream, prev_att, W, B, NH, T, Kn, Ks, out);
    ASR_LAUNCH_CHECK("asr_loc_conv_fwd");
    return ASR_OK;
}
extern "C" int asr_loc_conv_bwd(const float* dout, const float* prev_att, const float* W, int B, int NH, int T, int Kn, int Ks,
                                float* dprev, float* dW, asr_stream_t stream) {
    ASR_REQUIRE(dout && prev_att && W && dW && B > 0 && NH > 0 && T > 0 && Kn > 0 && Ks >= 0, ASR_E_ARG, "asr_loc_conv_bwd: bad args");
    hipStream_t st = (hipStream_t)stream;
    if (dprev) hipLaunchKernelGGL(loc_conv_bwd_data_kernel, dim3(cdiv((long)B * NH * T, 256)), dim3(256), 0, st, dout, W, B, NH, T, Kn, Ks, dprev);
    hipLaunchKernelGGL(loc_conv_bwd_weight_kernel, dim3(Kn * NH * (2 * Ks + 1)), dim3(64), 0, st, dout, prev_att, B, NH, T, Kn, Ks, dW);
    ASR_LAUNCH_CHECK("asr_loc_conv_bwd");
    return ASR_OK;
}
extern "C" int asr_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, int N, int D, float* act, float* h, float* c,
                                 asr_stream_t stream) {
    ASR_REQUIRE(gx && gh && act && h && c && N > 0 && D > 0, ASR_E_ARG, "asr_lstm_cell_fwd: bad args");
    hipLaunchKernelGGL(lstm_cell_train_fwd_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, (hipStream_t)stream, gx, gh, c_prev, N, D, act, h, c);
    ASR_LAUNCH_CHECK("asr_lstm_cell_fwd");
    return ASR_OK;
}
extern "C" int asr_lstm_cell_bwd(const float* act, const float* c_prev, const float* c, const float* dh, const float* dc, int N, int D,
                                 float* dgates, float* dc_prev, asr_stream_t stream) {
    ASR_REQUIRE(act && c && dgates && dc_prev && (dh || dc) && N > 0 && D > 0, ASR_E_ARG, "asr_lstm_cell_bwd: bad args");
    hipLaunchKernelGGL(lstm_cell_train_bwd_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, (hipStream_t)stream, act, c_prev, c, dh, dc, N, D, dgates, dc_prev);
    ASR_LAUNCH_CHECK("asr_lstm_cell_bwd");
    return ASR_OK;
}
extern "C" int asr_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, int N, int D, float* saved, float* h, asr_stream_t stream) {
    ASR_REQUIRE(gi && gh && saved && h && N > 0 && D > 0, ASR_E_ARG, "asr_gru_cell_fwd: bad args");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, (hipStream_t)stream, gi, gh, h_prev, N, D, saved, h);
    ASR_LAUNCH_CHECK("asr_gru_cell_fwd");
    return ASR_OK;
}
extern "C" int asr_gru_cell_bwd(const float* saved, const float* h_prev, const float* dh, int N, int D, float* dgi, float* dgh, float* dh_prev,
                                asr_stream_t stream) {
    ASR_REQUIRE(saved && dh && dgi && dgh && dh_prev && N > 0 && D > 0, ASR_E_ARG, "asr_gru_cell_bwd: bad args");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, (hipStream_t)stream, saved, h_prev, dh, N, D, dgi, dgh, dh_prev);
    ASR_LAUNCH_CHECK("asr_gru_cell_bwd");
    return ASR_OK;
}
extern "C" int asr_gru_fwd(const float* gi, const float* whhT, const float* bhh, int B, int T, int H, int ND, float* y, float* saved,
                           asr_stream_t stream) {
    ASR_REQUIRE(gi && whhT && bhh && y && saved && B > 0 && T > 0 && H > 0 && (ND == 1 || ND == 2), ASR_E_ARG, "asr_gru_fwd: bad args");
    ASR_REQUIRE(H <= 2048, ASR_E_UNSUPPORTED, "asr_gru_fwd: hidden size %d above 2048", H);
    hipLaunchKernelGGL(gru_seq_fwd_kernel, dim3(B, ND), dim3(1024), sizeof(float) * 4 * H, (hipStream_t)stream, gi, whhT, bhh, T, H, ND, y, saved);
    ASR_LAUNCH_CHECK("asr_gru_fwd");
    return ASR_OK;
}
extern "C" int asr_gru_bwd(const float* dy, const float* y, const float* saved, const float* whh, int B, int T, int H, int ND,
                           float* dgi, float* dgh, asr_stream_t stream) {
    ASR_REQUIRE(dy && y && saved && whh && dgi && dgh && B > 0 && T > 0 && H > 0 && (ND == 1 || ND == 2), ASR_E_ARG, "asr_gru_bwd: bad args");
    ASR_REQUIRE(H <= 2048, ASR_E_UNSUPPORTED, "asr_gru_bwd: hidden size %d above 2048", H);
    hipLaunchKernelGGL(gru_seq_bwd_kernel, dim3(B, ND), dim3(1024), sizeof(float) * 4 * H, (hipStream_t)stream, dy, y, saved, whh, T, H, ND, dgi, dgh);
    ASR_LAUNCH_CHECK("asr_gru_bwd");
    return ASR_OK;
}
