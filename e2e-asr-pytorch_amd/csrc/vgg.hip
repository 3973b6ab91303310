// Support kernels of the VGG front-ends (reference VGGExtractor src/module.py:659-716 and VGGExtractor_LN
// :582-657): layout permutes to/from the channel-last image the implicit-GEMM convolution (asr_conv3x3,
// gemm.hip) works on, 2x2 max pooling (ceil or floor mode) and LayerNorm over the frequency axis (+ReLU).
// All HBM-bound, channel-contiguous accesses.
#include "common.h"

namespace {

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

// out[r, b, a] = in[r, a, b]   (rows r, inner dims A x Bd)
__global__ void permute_last2_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int A, int Bd) {
    const long total = rows * A * Bd;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int a = (int)(i % A);
        const int b = (int)((i / A) % Bd);
        const long r = i / ((long)A * Bd);
        out[i] = in[(r * A + a) * Bd + b];
    }
}

// mode 0: dst (Co, 9*Ci)[co][tap*Ci+ci]  = src (Co,Ci,3,3)[co][ci][tap]                (forward operand)
// mode 1: dst (Ci, 9*Co)[ci][tap*Co+co]  = src[co][ci][8-tap]                           (input-gradient operand)
// mode 2: dst (Co,Ci,3,3)[co][ci][tap]  += src (Co, 9*Ci)[co][tap*Ci+ci]                (weight gradient back)
__global__ void conv_weight_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int mode) {
    const long total = (long)Co * Ci * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9);
        const int ci = (int)((i / 9) % Ci);
        const int co = (int)(i / (9L * Ci));
        if (mode == 0) dst[((long)co * 9 + tap) * Ci + ci] = src[i];
        else if (mode == 1) dst[((long)ci * 9 + (8 - tap)) * Co + co] = src[i];
        else dst[i] += src[((long)co * 9 + tap) * Ci + ci];
    }
}

__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, unsigned char* __restrict__ idx,
                                   int B, int T, int F, int C, int T2, int F2) {
    const long total = (long)B * T2 * F2 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int f2 = (int)((i / C) % F2);
        const int t2 = (int)((i / ((long)C * F2)) % T2);
        const int b = (int)(i / ((long)C * F2 * T2));
        float best = -INFINITY;
        int bi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int t = 2 * t2 + (k >> 1), f = 2 * f2 + (k & 1);
            if (t < T && f < F) {
                const float v = x[(((long)b * T + t) * F + f) * C + c];
                if (v > best) { best = v; bi = k; }
            }
        }
        y[i] = best;
        idx[i] = (unsigned char)bi;
    }
}

__global__ void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx, float* __restrict__ dx,
                                   int B, int T, int F, int C, int T2, int F2) {
    // one thread per INPUT element: gradient arrives iff it was the arg-max of its window
    const long total = (long)B * T * F * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int f = (int)((i / C) % F);
        const int t = (int)((i / ((long)C * F)) % T);
        const int b = (int)(i / ((long)C * F * T));
        const int t2 = t >> 1, f2 = f >> 1;
        float v = 0.f;
        if (t2 < T2 && f2 < F2) {
            const long o = (((long)b * T2 + t2) * F2 + f2) * C + c;
            if (idx[o] == (unsigned char)(((t & 1) << 1) | (f & 1))) v = dy[o];
        }
        dx[i] = v;
    }
}

// LayerNorm over F for every (b,t,c) of a channel-last image (B*T, F, C); affine per f; optional ReLU.
__global__ __launch_bounds__(256) void ln_f_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bia,
                                                       float* __restrict__ y, float* __restrict__ stats, long rows, int F, int C,
                                                       float eps, int relu) {
    const long total = rows * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / C;
        const int c = (int)(i % C);
        const float* xp = x + r * F * C + c;
        float s = 0.f;
        for (int f = 0; f < F; ++f) s += xp[(long)f * C];
        const float mean = s / F;
        float v = 0.f;
        for (int f = 0; f < F; ++f) { const float d = xp[(long)f * C] - mean; v += d * d; }
        const float rstd = rsqrtf(v / F + eps);
        float* yp = y + r * F * C + c;
        for (int f = 0; f < F; ++f) {
            const float o = (xp[(long)f * C] - mean) * rstd * w[f] + bia[f];
            yp[(long)f * C] = relu ? fmaxf(o, 0.f) : o;
        }
        stats[2 * i] = mean;
        stats[2 * i + 1] = rstd;
    }
}

template <int FMAX>
__global__ __launch_bounds__(256) void ln_f_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bia, const float* __restrict__ stats,
                                                       float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                                       long rows, int F, int C, int relu) {
    __shared__ float s_dw[FMAX], s_db[FMAX];
    for (int f = threadIdx.x; f < F; f += 256) { s_dw[f] = 0.f; s_db[f] = 0.f; }
    __syncthreads();
    const long total = rows * C;
    for (long i0 = blockIdx.x * 256L; i0 < total; i0 += (long)gridDim.x * 256) {
        const long i = i0 + threadIdx.x;
        const bool ok = i < total;
        const long r = ok ? i / C : 0;
        const int c = ok ? (int)(i % C) : 0;
        const float mean = ok ? stats[2 * i] : 0.f, rstd = ok ? stats[2 * i + 1] : 0.f;
        const float* xp = x + r * F * C + c;
        const float* gp = dy + r * F * C + c;
        float s1 = 0.f, s2 = 0.f;
        for (int f = 0; f < F; ++f) {
            float g = 0.f, xh = 0.f;
            if (ok) {
                xh = (xp[(long)f * C] - mean) * rstd;
                g = gp[(long)f * C];
                if (relu && (xh * w[f] + bia[f]) <= 0.f) g = 0.f;
            }
            s1 += g * w[f];
            s2 += g * w[f] * xh;
            // per-f parameter gradients: wave reduction, then one LDS atomic per wave
            const float gw = wave_sum(g * xh), gb = wave_sum(g);
            if ((threadIdx.x & 63) == 0) { atomicAdd(&s_dw[f], gw); atomicAdd(&s_db[f], gb); }
        }
        if (ok) {
            s1 /= F; s2 /= F;
            float* dp = dx + r * F * C + c;
            for (int f = 0; f < F; ++f) {
                const float xh = (xp[(long)f * C] - mean) * rstd;
                float g = gp[(long)f * C];
                if (relu && (xh * w[f] + bia[f]) <= 0.f) g = 0.f;
                dp[(long)f * C] = rstd * (g * w[f] - s1 - xh * s2);
            }
        }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) { atomicAdd(dw + f, s_dw[f]); atomicAdd(db + f, s_db[f]); }
}

}  // namespace

extern "C" int asr_permute_last2(const float* in, float* out, long rows, int A, int Bd, asr_stream_t stream) {
    ASR_REQUIRE(in && out && rows > 0 && A > 0 && Bd > 0, ASR_E_ARG, "asr_permute_last2: bad args");
    hipLaunchKernelGGL(permute_last2_kernel, dim3(grid_for(rows * A * Bd)), dim3(256), 0, (hipStream_t)stream, in, out, rows, A, Bd);
    ASR_LAUNCH_CHECK("asr_permute_last2");
    return ASR_OK;
}

extern "C" int asr_conv_weight_permute(const float* src, float* dst, int Co, int Ci, int mode, asr_stream_t stream) {
    ASR_REQUIRE(src && dst && Co > 0 && Ci > 0 && mode >= 0 && mode <= 2, ASR_E_ARG, "asr_conv_weight_permute: bad args");
    hipLaunchKernelGGL(conv_weight_permute_kernel, dim3(grid_for((long)Co * Ci * 9)), dim3(256), 0, (hipStream_t)stream, src, dst, Co, Ci, mode);
    ASR_LAUNCH_CHECK("asr_conv_weight_permute");
    return ASR_OK;
}

extern "C" int asr_maxpool2x2_fwd(const float* x, float* y, unsigned char* idx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream) {
    ASR_REQUIRE(x && y && idx && B > 0 && T > 0 && F > 0 && C > 0 && T2 > 0 && F2 > 0, ASR_E_ARG, "asr_maxpool2x2_fwd: bad args");
    ASR_REQUIRE(2 * T2 - 1 <= T && 2 * F2 - 1 <= F, ASR_E_ARG, "asr_maxpool2x2_fwd: output larger than ceil(T/2) x ceil(F/2)");
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for((long)B * T2 * F2 * C)), dim3(256), 0, (hipStream_t)stream, x, y, idx, B, T, F, C, T2, F2);
    ASR_LAUNCH_CHECK("asr_maxpool2x2_fwd");
    return ASR_OK;
}

extern "C" int asr_maxpool2x2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int T, int F, int C, int T2, int F2,
                                  asr_stream_t stream) {
    ASR_REQUIRE(dy && idx && dx && B > 0 && T > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_maxpool2x2_bwd: bad args");
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(grid_for((long)B * T * F * C)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, B, T, F, C, T2, F2);
    ASR_LAUNCH_CHECK("asr_maxpool2x2_bwd");
    return ASR_OK;
}

extern "C" int asr_ln_freq_fwd(const float* x, const float* w, const float* b, float* y, float* stats, long rows, int F, int C,
                               float eps, int relu, asr_stream_t stream) {
    ASR_REQUIRE(x && w && b && y && stats && rows > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_ln_freq_fwd: bad args");
    hipLaunchKernelGGL(ln_f_fwd_kernel, dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream, x, w, b, y, stats, rows, F, C, eps, relu);
    ASR_LAUNCH_CHECK("asr_ln_freq_fwd");
    return ASR_OK;
}

extern "C" int asr_ln_freq_bwd(const float* dy, const float* x, const float* w, const float* b, const float* stats,
                               float* dx, float* dw, float* db, long rows, int F, int C, int relu, asr_stream_t stream) {
    ASR_REQUIRE(dy && x && w && b && stats && dx && dw && db && rows > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_ln_freq_bwd: bad args");
    ASR_REQUIRE(F <= 128, ASR_E_UNSUPPORTED, "asr_ln_freq_bwd: F=%d > 128", F);
    long g = (rows * C + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(ln_f_bwd_kernel<128>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, dy, x, w, b, stats, dx, dw, db, rows, F, C, relu);
    ASR_LAUNCH_CHECK("asr_ln_freq_bwd");
    return ASR_OK;
}
