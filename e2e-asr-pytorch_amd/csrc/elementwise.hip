// HBM-bound glue kernels of the encoder layers and heads: dropout + time down-sampling
// (reference src/module.py:1059-1076), activation backward, bias-gradient column sums, row-wise
// log-softmax (src/asr.py:120), LayerNorm over the last axis (src/module.py:1031,1057; 546-550).
#include "common.h"

namespace {

// ---- dropout + down-sampling ------------------------------------------------------------------
// style 0 ('drop'): z[b,t2,k] = drop(y)[b, t2*rate, k];  style 1 ('concat'): z[b,t2,i*D+k] = drop(y)[b,t2*rate+i,k]
// The keep decision of element (b,t,k) of y is Philox(seed, flat index in y), independent of the style.
struct DsP { const float* src; float* dst; int B, T, D, T2, rate, style; float scale; uint32_t thresh; uint64_t seed; };

template <bool BWD>
__global__ void dropout_downsample_kernel(DsP p) {
    // forward: src = y (B,T,D), dst = z;  backward: src = dz, dst = dy (B,T,D) fully written (zeros where dropped)
    const long total = (long)p.B * p.T * p.D;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % p.D);
        const int t = (int)((i / p.D) % p.T);
        const int b = (int)(i / ((long)p.D * p.T));
        long zi = -1;
        if (p.style == 0) {
            if (t % p.rate == 0 && t / p.rate < p.T2) zi = ((long)b * p.T2 + t / p.rate) * p.D + k;
        } else {
            if (t < p.T2 * p.rate) zi = ((long)b * p.T2 + t / p.rate) * ((long)p.D * p.rate) + (long)(t % p.rate) * p.D + k;
        }
        if (BWD) {
            float v = 0.f;
            if (zi >= 0) {
                const bool keep = p.thresh == 0 || dropout_keep(p.seed, (uint64_t)i, p.thresh);
                v = keep ? p.src[zi] * p.scale : 0.f;
            }
            p.dst[i] = v;
        } else if (zi >= 0) {
            const bool keep = p.thresh == 0 || dropout_keep(p.seed, (uint64_t)i, p.thresh);
            p.dst[zi] = keep ? p.src[i] * p.scale : 0.f;
        }
    }
}

// Same map for D % 4 == 0 and 16-byte aligned buffers: a thread owns four consecutive elements of a row of y - ONE Philox
// block (dropout_keep draws element idx from word idx & 3 of block idx >> 2), one 16-byte load, one 16-byte store;
// the row decomposition is done in 32-bit arithmetic per (b,t) row.
template <bool BWD>
__global__ void dropout_downsample_vec4_kernel(DsP p) {
    const int D4 = p.D >> 2;
    const long rows = (long)p.B * p.T, total4 = rows * D4;
    for (long j = blockIdx.x * (long)blockDim.x + threadIdx.x; j < total4; j += (long)gridDim.x * blockDim.x) {
        const long row = j / D4;
        const int k4 = (int)(j - row * D4);
        const int b = (int)(row / p.T), t = (int)(row - (long)b * p.T);
        long zi = -1;
        if (p.style == 0) {
            if (t % p.rate == 0 && t / p.rate < p.T2) zi = ((long)b * p.T2 + t / p.rate) * p.D + 4 * k4;
        } else {
            if (t < p.T2 * p.rate) zi = ((long)b * p.T2 + t / p.rate) * ((long)p.D * p.rate) + (long)(t % p.rate) * p.D + 4 * k4;
        }
        const long i = 4 * j;                                            // flat index in y: a multiple of 4 = one Philox block
        if (!BWD && zi < 0) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (zi >= 0) {
            const float4 x = *reinterpret_cast<const float4*>(p.src + (BWD ? zi : i));
            uint32_t r[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (p.thresh != 0) {
                const uint64_t blk = (uint64_t)i >> 2;
                philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, (uint32_t)p.seed, (uint32_t)(p.seed >> 32), r);
            }
            v.x = (r[0] >= p.thresh) ? x.x * p.scale : 0.f;
            v.y = (r[1] >= p.thresh) ? x.y * p.scale : 0.f;
            v.z = (r[2] >= p.thresh) ? x.z * p.scale : 0.f;
            v.w = (r[3] >= p.thresh) ? x.w * p.scale : 0.f;
        }
        *reinterpret_cast<float4*>(p.dst + (BWD ? i : zi)) = v;
    }
}

__global__ void dropout_mask_kernel(float* mask, long n, uint32_t thresh, uint64_t seed) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        mask[i] = (thresh == 0 || dropout_keep(seed, (uint64_t)i, thresh)) ? 1.f : 0.f;
}

__global__ void act_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out, float* __restrict__ dpre,
                               long n, int act) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float o = out[i], g = dout[i];
        dpre[i] = (act == ASR_ACT_TANH) ? g * (1.f - o * o) : ((o > 0.f) ? g : 0.f);
    }
}

// column sums of a (M,N) matrix with row stride lda, atomically added into out[N]
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ A, long lda, int M, int N,
                                                     float* __restrict__ out, float* __restrict__ out2, int rows_per_block) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int grp = threadIdx.x >> 6;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    if (col < N) {
        float s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int r = r0 + grp;
        for (; r + 12 < r1; r += 16) {                      // four independent loads in flight
            const float a0 = A[(long)r * lda + col], a1 = A[(long)(r + 4) * lda + col];
            const float a2 = A[(long)(r + 8) * lda + col], a3 = A[(long)(r + 12) * lda + col];
            s += a0; s1 += a1; s2 += a2; s3 += a3;
        }
        for (; r < r1; r += 4) s += A[(long)r * lda + col];
        s += s1 + s2 + s3;
    }
    red[grp][threadIdx.x & 63] = s;
    __syncthreads();
    if (grp == 0 && col < N) {
        const float v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        atomicAdd(out + col, v);
        if (out2) atomicAdd(out2 + col, v);
    }
}

// one wave per row: out = x - logsumexp(x)
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ out, long R, int V) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + row * V;
    float m = -INFINITY;
    for (int v = lane; v < V; v += 64) m = fmaxf(m, xr[v]);
    m = wave_max(m);
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += expf(xr[v] - m);
    s = wave_sum(s);
    const float lse = m + logf(s);
    for (int v = lane; v < V; v += 64) out[row * V + v] = xr[v] - lse;
}

__global__ __launch_bounds__(256) void logsoftmax_relu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ lp,
                                                                  const float* __restrict__ act, float* __restrict__ dpre, long R, int V) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    float s = 0.f;
    for (int v = lane; v < V; v += 64) s += g[row * V + v];
    s = wave_sum(s);
    for (int v = lane; v < V; v += 64) {
        const long i = row * V + v;
        dpre[i] = (act[i] > 0.f) ? (g[i] - expf(lp[i]) * s) : 0.f;
    }
}

// LayerNorm over the last axis (n <= a few thousand): one wave per row.
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y,
                                                            float* __restrict__ stats, long R, int n, float eps, int relu) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float* xr = x + row * n;
    float s = 0.f;
    for (int i = lane; i < n; i += 64) s += xr[i];
    const float mean = wave_sum(s) / n;
    float v = 0.f;
    for (int i = lane; i < n; i += 64) { float d = xr[i] - mean; v += d * d; }
    const float rstd = rsqrtf(wave_sum(v) / n + eps);
    for (int i = lane; i < n; i += 64) {
        float o = (xr[i] - mean) * rstd * w[i] + b[i];
        y[row * n + i] = relu ? fmaxf(o, 0.f) : o;
    }
    if (lane == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
}

// dx for LayerNorm(+optional ReLU); dw/db accumulated with atomics into (n) buffers.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ stats, float* __restrict__ dx,
                                                            float* __restrict__ dw, float* __restrict__ db,
                                                            long R, int n, int relu) {
    const long row = blockIdx.x * 4L + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= R) return;
    const float mean = stats[2 * row], rstd = stats[2 * row + 1];
    const float* xr = x + row * n;
    const float* gr = dy + row * n;
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < n; i += 64) {
        const float xh = (xr[i] - mean) * rstd;
        float g = gr[i];
        if (relu && (xh * w[i] + b[i]) <= 0.f) g = 0.f;
        const float gw = g * w[i];
        s1 += gw; s2 += gw * xh;
        atomicAdd(dw + i, g * xh);
        atomicAdd(db + i, g);
    }
    s1 = wave_sum(s1) / n; s2 = wave_sum(s2) / n;
    for (int i = lane; i < n; i += 64) {
        const float xh = (xr[i] - mean) * rstd;
        float g = gr[i];
        if (relu && (xh * w[i] + b[i]) <= 0.f) g = 0.f;
        dx[row * n + i] = rstd * (g * w[i] - s1 - xh * s2);
    }
}

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 4096 ? 4096 : g)); }

inline uint32_t drop_thresh(float p) {
    if (p <= 0.f) return 0u;
    double v = (double)p * 4294967296.0;
    return v >= 4294967295.0 ? 4294967295u : (uint32_t)v;
}

}  // namespace

extern "C" int asr_dropout_downsample_fwd(const float* y, float* z, int B, int T, int D, int T2, int rate, int style,
                                          float p, uint64_t seed, asr_stream_t stream) {
    ASR_REQUIRE(y && z && B > 0 && T > 0 && D > 0 && T2 > 0 && rate >= 1, ASR_E_ARG, "asr_dropout_downsample_fwd: bad args");
    ASR_REQUIRE(p >= 0.f && p < 1.f, ASR_E_ARG, "asr_dropout_downsample_fwd: p must be in [0,1)");
    DsP a{y, z, B, T, D, T2, rate, style, 1.f / (1.f - p), drop_thresh(p), seed};
    if ((D & 3) == 0 && (((uintptr_t)y | (uintptr_t)z) & 15) == 0)
        hipLaunchKernelGGL(dropout_downsample_vec4_kernel<false>, dim3(grid_for((long)B * T * (D / 4))), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(dropout_downsample_kernel<false>, dim3(grid_for((long)B * T * D)), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_dropout_downsample_fwd");
    return ASR_OK;
}

extern "C" int asr_dropout_downsample_bwd(const float* dz, float* dy, int B, int T, int D, int T2, int rate, int style,
                                          float p, uint64_t seed, asr_stream_t stream) {
    ASR_REQUIRE(dz && dy && B > 0 && T > 0 && D > 0 && T2 > 0 && rate >= 1, ASR_E_ARG, "asr_dropout_downsample_bwd: bad args");
    DsP a{dz, dy, B, T, D, T2, rate, style, 1.f / (1.f - p), drop_thresh(p), seed};
    if ((D & 3) == 0 && (((uintptr_t)dz | (uintptr_t)dy) & 15) == 0)
        hipLaunchKernelGGL(dropout_downsample_vec4_kernel<true>, dim3(grid_for((long)B * T * (D / 4))), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(dropout_downsample_kernel<true>, dim3(grid_for((long)B * T * D)), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_dropout_downsample_bwd");
    return ASR_OK;
}

extern "C" int asr_dropout_mask(float* mask, long n, float p, uint64_t seed, asr_stream_t stream) {
    ASR_REQUIRE(mask && n > 0, ASR_E_ARG, "asr_dropout_mask: bad args");
    hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, mask, n, drop_thresh(p), seed);
    ASR_LAUNCH_CHECK("asr_dropout_mask");
    return ASR_OK;
}

extern "C" int asr_act_bwd(const float* dout, const float* out, float* dpre, long n, int act, asr_stream_t stream) {
    ASR_REQUIRE(dout && out && dpre && n > 0, ASR_E_ARG, "asr_act_bwd: bad args");
    ASR_REQUIRE(act == ASR_ACT_TANH || act == ASR_ACT_RELU, ASR_E_ARG, "asr_act_bwd: bad act");
    hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dout, out, dpre, n, act);
    ASR_LAUNCH_CHECK("asr_act_bwd");
    return ASR_OK;
}

extern "C" int asr_colsum2(const float* A, long lda, int M, int N, float* out, float* out2, asr_stream_t stream) {
    ASR_REQUIRE(A && out && M > 0 && N > 0 && lda >= N, ASR_E_ARG, "asr_colsum: bad args");
    int rows_per_block = 512;                               // small matrices: more, shorter row blocks (about one round of workgroups)
    while (rows_per_block > 32 && (long)cdiv(N, 64) * cdiv(M, rows_per_block) < 1024) rows_per_block >>= 1;
    dim3 grid(cdiv(N, 64), cdiv(M, rows_per_block));
    hipLaunchKernelGGL(colsum_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, lda, M, N, out, out2, rows_per_block);
    ASR_LAUNCH_CHECK("asr_colsum");
    return ASR_OK;
}
extern "C" int asr_colsum(const float* A, long lda, int M, int N, float* out, asr_stream_t stream) {
    return asr_colsum2(A, lda, M, N, out, nullptr, stream);
}

extern "C" int asr_log_softmax(const float* x, float* out, long rows, int V, asr_stream_t stream) {
    ASR_REQUIRE(x && out && rows > 0 && V > 0, ASR_E_ARG, "asr_log_softmax: bad args");
    hipLaunchKernelGGL(log_softmax_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, out, rows, V);
    ASR_LAUNCH_CHECK("asr_log_softmax");
    return ASR_OK;
}

extern "C" int asr_logsoftmax_relu_bwd(const float* dlogp, const float* logp, const float* act, float* dpre,
                                       long rows, int V, asr_stream_t stream) {
    ASR_REQUIRE(dlogp && logp && act && dpre && rows > 0 && V > 0, ASR_E_ARG, "asr_logsoftmax_relu_bwd: bad args");
    hipLaunchKernelGGL(logsoftmax_relu_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dlogp, logp, act, dpre, rows, V);
    ASR_LAUNCH_CHECK("asr_logsoftmax_relu_bwd");
    return ASR_OK;
}

extern "C" int asr_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats,
                                 long rows, int n, float eps, int relu, asr_stream_t stream) {
    ASR_REQUIRE(x && w && b && y && stats && rows > 0 && n > 0, ASR_E_ARG, "asr_layernorm_fwd: bad args");
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, stats, rows, n, eps, relu);
    ASR_LAUNCH_CHECK("asr_layernorm_fwd");
    return ASR_OK;
}

extern "C" int asr_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* stats,
                                 float* dx, float* dw, float* db, long rows, int n, int relu, asr_stream_t stream) {
    ASR_REQUIRE(dy && x && w && b && stats && dx && dw && db && rows > 0 && n > 0, ASR_E_ARG, "asr_layernorm_bwd: bad args");
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, dy, x, w, b, stats, dx, dw, db, rows, n, relu);
    ASR_LAUNCH_CHECK("asr_layernorm_bwd");
    return ASR_OK;
}
