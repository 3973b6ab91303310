// Kernels of the beam-search inference path (reference src/decode.py:65-183, src/ctc.py:68-107, src/lm.py:27-37),
// batched over live hypotheses:
//   asr_ctc_prefix_score : CTCPrefixScore.cheap_compute for N hypotheses x C candidate tokens in one launch
//   asr_ctc_prefix_init  : CTCPrefixScore.init_state (running blank-only path)
//   asr_lstm_cell        : pointwise LSTM cell on pre-computed gate sums (RNN-LM step; the two projections are asr_gemm)
//   asr_gather_rows      : row gather (embedding lookup, state re-ordering after pruning)
#include "common.h"

namespace {

constexpr float LOGZERO = -100000000.0f;   // src/ctc.py:12

// numpy float32 logaddexp (finite "log zero", no -inf handling needed)
__device__ __forceinline__ float lae(float a, float b) {
    const float m = fmaxf(a, b);
    return m + log1pf(expf(-fabsf(a - b)));
}

__global__ void ctc_prefix_init_kernel(const float* __restrict__ x, float* __restrict__ r, int T, int V) {
    // r (T,2): r[:,0] = logzero ; r[t,1] = cumulative blank log-prob
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float acc = 0.f;
        for (int t = 0; t < T; ++t) {
            acc = (t == 0) ? x[0] : acc + x[(long)t * V];
            r[2 * t] = LOGZERO;
            r[2 * t + 1] = acc;
        }
    }
}

// one thread per (hypothesis n, candidate c)
__global__ void ctc_prefix_score_kernel(const float* __restrict__ x, const float* __restrict__ r_prev, const int* __restrict__ cand,
                                        const int* __restrict__ prefix_len, const int* __restrict__ last_tok,
                                        float* __restrict__ psi_out, float* __restrict__ r_out, int N, int C, int T, int V) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * C) return;
    const int n = i / C;
    const int tok = cand[i];
    const int plen = prefix_len[n], last = last_tok[n];
    const float* rp = r_prev + (long)n * T * 2;
    float* ro = r_out + (long)i * T * 2;
    const int start = max(1, plen);
    for (int t = 0; t < start && t < T; ++t) { ro[2 * t] = LOGZERO; ro[2 * t + 1] = LOGZERO; }
    if (plen == 0) ro[0] = x[tok];
    float r0 = ro[2 * (start - 1)], r1 = ro[2 * (start - 1) + 1];
    float psi = r0;
    const bool same = (plen > 0 && tok == last);
    for (int t = start; t < T; ++t) {
        const float p0 = rp[2 * (t - 1)], p1 = rp[2 * (t - 1) + 1];
        const float phi = same ? p1 : lae(p0, p1);
        const float xt = x[(long)t * V + tok];
        const float n0 = lae(r0, phi) + xt;
        const float n1 = lae(r1, r0) + x[(long)t * V];
        psi = lae(psi, phi + xt);
        r0 = n0; r1 = n1;
        ro[2 * t] = r0; ro[2 * t + 1] = r1;
    }
    if (tok == 1) psi = lae(rp[2 * (T - 1)], rp[2 * (T - 1) + 1]);   // <eos>: probability of the prefix itself
    psi_out[i] = psi;
}

__global__ void lstm_cell_kernel(const float* __restrict__ pre, const float* __restrict__ bih, const float* __restrict__ bhh,
                                 const float* __restrict__ c_prev, float* __restrict__ h, float* __restrict__ c, int N, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * D) return;
    const int n = i / D, j = i % D;
    const float* p = pre + (long)n * 4 * D;
    const float gi = sigmoidf_(p[j] + bih[j] + bhh[j]);
    const float gf = sigmoidf_(p[D + j] + bih[D + j] + bhh[D + j]);
    const float gg = tanhf(p[2 * D + j] + bih[2 * D + j] + bhh[2 * D + j]);
    const float go = sigmoidf_(p[3 * D + j] + bih[3 * D + j] + bhh[3 * D + j]);
    const float cn = gf * (c_prev ? c_prev[i] : 0.f) + gi * gg;
    c[i] = cn;
    h[i] = go * tanhf(cn);
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int64_t* __restrict__ idx, float* __restrict__ dst,
                                   int rows, int width, long src_ld, long dst_ld, int nsrc) {
    const long total = (long)rows * width;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / width), k = (int)(i % width);
        long s = idx[r];
        s = s < 0 ? 0 : (s >= nsrc ? nsrc - 1 : s);
        dst[(long)r * dst_ld + k] = src[s * src_ld + k];
    }
}

}  // namespace

extern "C" int asr_ctc_prefix_init(const float* logp, float* r, int T, int V, asr_stream_t stream) {
    ASR_REQUIRE(logp && r && T > 0 && V > 1, ASR_E_ARG, "asr_ctc_prefix_init: bad args");
    hipLaunchKernelGGL(ctc_prefix_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, logp, r, T, V);
    ASR_LAUNCH_CHECK("asr_ctc_prefix_init");
    return ASR_OK;
}

extern "C" int asr_ctc_prefix_score(const float* logp, const float* r_prev, const int* candidates, const int* prefix_len,
                                    const int* last_token, float* psi, float* r_out, int N, int C, int T, int V,
                                    asr_stream_t stream) {
    ASR_REQUIRE(logp && r_prev && candidates && prefix_len && last_token && psi && r_out, ASR_E_ARG, "asr_ctc_prefix_score: null pointer");
    ASR_REQUIRE(N > 0 && C > 0 && T > 0 && V > 1, ASR_E_ARG, "asr_ctc_prefix_score: bad dims");
    hipLaunchKernelGGL(ctc_prefix_score_kernel, dim3(cdiv((long)N * C, 64)), dim3(64), 0, (hipStream_t)stream, logp, r_prev, candidates,
                       prefix_len, last_token, psi, r_out, N, C, T, V);
    ASR_LAUNCH_CHECK("asr_ctc_prefix_score");
    return ASR_OK;
}

extern "C" int asr_lstm_cell(const float* gates_pre, const float* bias_ih, const float* bias_hh, const float* c_prev,
                             float* h, float* c, int N, int D, asr_stream_t stream) {
    ASR_REQUIRE(gates_pre && bias_ih && bias_hh && h && c && N > 0 && D > 0, ASR_E_ARG, "asr_lstm_cell: bad args");
    hipLaunchKernelGGL(lstm_cell_kernel, dim3(cdiv((long)N * D, 256)), dim3(256), 0, (hipStream_t)stream, gates_pre, bias_ih, bias_hh,
                       c_prev, h, c, N, D);
    ASR_LAUNCH_CHECK("asr_lstm_cell");
    return ASR_OK;
}

extern "C" int asr_gather_rows(const float* src, const int64_t* idx, float* dst, int rows, int width, long src_ld, long dst_ld,
                               int nsrc, asr_stream_t stream) {
    ASR_REQUIRE(src && idx && dst && rows > 0 && width > 0 && nsrc > 0, ASR_E_ARG, "asr_gather_rows: bad args");
    long g = ((long)rows * width + 255) / 256; if (g > 4096) g = 4096;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, src, idx, dst, rows, width, src_ld, dst_ld, nsrc);
    ASR_LAUNCH_CHECK("asr_gather_rows");
    return ASR_OK;
}
