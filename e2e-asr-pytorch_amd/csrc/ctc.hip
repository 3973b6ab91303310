// CTC loss (blank = 0, reduction 'mean', zero_infinity = False) forward + gradient in one launch —
// stands in for torch.nn.CTCLoss as the reference calls it (bin/train_asr.py:135,237).
//
// One workgroup per utterance; the 2L+1 states of the extended label sequence are striped over the
// threads, alpha_{t-1} lives in an LDS double buffer, the time loop is sequential with one barrier per
// frame.  alpha is spilled to HBM (workspace, (B,T,2L+1) fp32) and re-read by the beta sweep, which
// also forms the gradient.  All of it is fp32 log-space (log-sum-exp), whatever `prec` the model runs in.
//
// Gradient convention = the one torch returns for log-softmax inputs: for t < input_len
//   g[b,t,v] = gscale / (B * max(target_len_b,1)) * ( exp(lp[b,t,v]) - sum_{s: ext_s = v} exp(alpha+beta-lp+nll) )
// and exactly 0 for t >= input_len.  An infeasible alignment gives nll = +inf and NaN gradients.
#include "common.h"

namespace {

struct CtcP {
    const float* lp;        // (B,T,V) log-probs
    const int64_t* tgt;     // (B,L) padded with 0
    const int64_t* in_len;  // (B)
    const int64_t* tgt_len; // (B)
    float* nll;             // (B)
    float* loss;            // scalar, += nll_b / (max(tl,1) * B)
    float* grad;            // (B,T,V)
    float* alpha;           // workspace (B,T,Smax)
    int B, T, V, L, Smax;
    float gscale;
};

__global__ __launch_bounds__(256) void ctc_loss_kernel(CtcP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, NTH = blockDim.x;
    const int V = p.V, T = p.T;
    int tl = (int)p.tgt_len[b];
    tl = max(0, min(tl, p.L));
    const int S = 2 * tl + 1;
    int Tin = (int)p.in_len[b];
    Tin = max(0, min(Tin, T));

    int* ext = reinterpret_cast<int*>(smem);                     // [Smax]
    float* buf0 = reinterpret_cast<float*>(ext + p.Smax);         // [Smax]
    float* buf1 = buf0 + p.Smax;                                  // [Smax]
    float* acc = buf1 + p.Smax;                                   // [V]

    const float* lp = p.lp + (long)b * T * V;
    float* grad = p.grad + (long)b * T * V;
    float* alpha = p.alpha + (long)b * T * p.Smax;

    for (int s = tid; s < S; s += NTH) ext[s] = (s & 1) ? (int)p.tgt[(long)b * p.L + (s >> 1)] : 0;
    // zero the whole gradient slab of this utterance first (covers t >= Tin and the Tin == 0 case)
    for (long i = tid; i < (long)T * V; i += NTH) grad[i] = 0.f;
    __syncthreads();

    if (Tin == 0) {
        // no frames: only the empty target is feasible
        if (tid == 0) {
            float n = (tl == 0) ? 0.f : INFINITY;
            p.nll[b] = n;
            atomicAdd(p.loss, n / (float)(max(tl, 1) * p.B));
        }
        return;
    }

    // ---- alpha sweep ---------------------------------------------------------------------------
    float* prev = buf0;
    float* cur = buf1;
    for (int s = tid; s < S; s += NTH) {
        float a = -INFINITY;
        if (s == 0) a = lp[0];
        else if (s == 1) a = lp[ext[1]];
        prev[s] = a;
        alpha[s] = a;
    }
    __syncthreads();
    for (int t = 1; t < Tin; ++t) {
        const float* lpt = lp + (long)t * V;
        for (int s = tid; s < S; s += NTH) {
            const int e = ext[s];
            float a = prev[s];
            if (s >= 1) a = logaddexpf_(a, prev[s - 1]);
            if (s >= 2 && e != 0 && e != ext[s - 2]) a = logaddexpf_(a, prev[s - 2]);
            a += lpt[e];
            cur[s] = a;
            alpha[(long)t * p.Smax + s] = a;
        }
        __syncthreads();
        float* tmp = prev; prev = cur; cur = tmp;
    }
    float ll = prev[S - 1];
    if (S > 1) ll = logaddexpf_(ll, prev[S - 2]);
    const float nll = -ll;
    __syncthreads();
    if (tid == 0) {
        p.nll[b] = nll;
        atomicAdd(p.loss, nll / (float)(max(tl, 1) * p.B));
    }
    const float scale = p.gscale / (float)(max(tl, 1) * p.B);

    // ---- beta sweep + gradient -----------------------------------------------------------------
    // prev <- beta_{t+1}
    for (int t = Tin - 1; t >= 0; --t) {
        const float* lpt = lp + (long)t * V;
        for (int v = tid; v < V; v += NTH) acc[v] = 0.f;
        __syncthreads();
        for (int s = tid; s < S; s += NTH) {
            const int e = ext[s];
            float bt;
            if (t == Tin - 1) {
                bt = (s == S - 1 || s == S - 2) ? 0.f : -INFINITY;
            } else {
                bt = prev[s];
                if (s + 1 < S) bt = logaddexpf_(bt, prev[s + 1]);
                if (s + 2 < S && ext[s + 2] != 0 && ext[s + 2] != e) bt = logaddexpf_(bt, prev[s + 2]);
            }
            const float lpe = lpt[e];
            bt += lpe;
            cur[s] = bt;
            // occupancy term: exp(alpha + beta - lp + nll); NaN when nll = +inf (infeasible), as in the reference
            const float term = expf(alpha[(long)t * p.Smax + s] + bt - lpe + nll);
            atomicAdd(&acc[e], term);
        }
        __syncthreads();
        for (int v = tid; v < V; v += NTH) grad[(long)t * V + v] = scale * (expf(lpt[v]) - acc[v]);
        float* tmp = prev; prev = cur; cur = tmp;
        __syncthreads();
    }
}

}  // namespace

extern "C" size_t asr_ctc_loss_workspace_bytes(int B, int T, int L) {
    return (size_t)B * T * (2 * (size_t)L + 1) * sizeof(float);
}

extern "C" int asr_ctc_loss(const float* logp, const int64_t* targets, const int64_t* input_len, const int64_t* target_len,
                            float* nll, float* loss, float* grad, int B, int T, int V, int L, float gscale,
                            void* workspace, size_t workspace_bytes, asr_stream_t stream) {
    ASR_REQUIRE(logp && targets && input_len && target_len && nll && loss && grad && workspace, ASR_E_ARG, "asr_ctc_loss: null pointer");
    ASR_REQUIRE(B > 0 && T > 0 && V > 1 && L > 0, ASR_E_ARG, "asr_ctc_loss: bad dims");
    ASR_REQUIRE(workspace_bytes >= asr_ctc_loss_workspace_bytes(B, T, L), ASR_E_ARG, "asr_ctc_loss: workspace too small");
    CtcP p{logp, targets, input_len, target_len, nll, loss, grad, (float*)workspace, B, T, V, L, 2 * L + 1, gscale};
    size_t lds = (size_t)p.Smax * 12 + (size_t)V * 4;
    ASR_REQUIRE(lds <= 64 * 1024, ASR_E_UNSUPPORTED, "asr_ctc_loss: 2L+1=%d states and V=%d classes exceed the LDS budget", p.Smax, V);
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(loss, 0, sizeof(float), st);
    hipLaunchKernelGGL(ctc_loss_kernel, dim3(B), dim3(256), lds, st, p);
    ASR_LAUNCH_CHECK("asr_ctc_loss");
    return ASR_OK;
}
