// bf16 VGG front-end on zero-bordered channel-last images (bf16 contraction mode; reference VGGExtractor src/module.py:659-716,
// VGGExtractor_LN :582-657).  Round 2 ran the 3x3 convolutions on the fp32-operand contraction kernel (global -> VGPR -> convert
// -> LDS, ~120 TFLOP/s); here they are implicit GEMMs on the direct-to-LDS bf16 kernel of gemm16.hip:
//   * every activation of the stack is a bf16 image P(T, F, C) = (B, T+2, F+2, C) whose border pixels are ZERO, so the tap
//     (dt, df) of a 3x3 convolution is a CONSTANT row shift dt*(F+2) + df of the pixel-row matrix: k-step `kt` of the
//     9*C-deep reduction adds one scalar to the per-lane source offsets (gemm16_nt_kernel<.., CONV>), no bounds tests;
//     outputs of border rows are stored as zeros, which is the next layer's padding;
//   * input gradient = the same kernel with the flipped weight copy; weight gradient = nine shifted-row TN contractions
//     (one launch of gemm16_tn_kernel, the tap as a grid dimension) over the same images;
//   * the first layer (C = input channels = 4: K = 36) reads an explicit patch matrix over the same pixel grid (61 MB);
//   * pooling, LayerNorm-over-frequency and the layout changes at both ends work on the bordered bf16 images, 16 bytes per lane.
#include "common.h"

int gemm16_conv3x3(const void* img, const void* W, void* out, const float* bias, int B, int T, int F, int C, int N, int K, int implicit, int act,
                   int out_f32, hipStream_t st);
int gemm16_tn_taps(const void* A, const void* B, float* C, int I, int J, int R, long lda, long ldb, long ldc, int splits, int perm_h,
                   int seqT, int bshift, int padded, int tapF2, hipStream_t st);

namespace {

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }
__device__ __forceinline__ float bfv(unsigned short x) { return __uint_as_float((unsigned)x << 16); }

// feature (B, T, Cin*F) fp32, channel-major -> patch matrix X1 (B*(T+2)*(F+2), Kp) bf16 over the bordered pixel grid:
// X1[(b,tp,fp)][tap*Cin + ci] = feature[b, tp-1+dt, ci*F + fp-1+df] (0 outside the image; border rows all zero; columns >= 9*Cin zero)
__global__ void vgg16_im2col_kernel(const float* __restrict__ feat, unsigned short* __restrict__ X1, int B, int T, int F, int Cin, int Kp) {
    const int T2 = T + 2, F2 = F + 2;
    const long total = (long)B * T2 * F2 * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp);
        const long m = i / Kp;
        const int fp = (int)(m % F2), tp = (int)((m / F2) % T2), b = (int)(m / ((long)F2 * T2));
        float v = 0.f;
        if (k < 9 * Cin && fp >= 1 && fp <= F && tp >= 1 && tp <= T) {
            const int tap = k / Cin, ci = k - tap * Cin;
            const int t = tp - 1 + tap / 3 - 1, f = fp - 1 + tap % 3 - 1;
            if (t >= 0 && t < T && f >= 0 && f < F) v = feat[((long)b * T + t) * ((long)Cin * F) + (long)ci * F + f];
        }
        X1[i] = f2bf_bits(v);
    }
}

// mode 0: dst (Co, Kp)[co][tap*Ci+ci] = src (Co,Ci,3,3)[co][ci][tap]          forward operand (Kp >= 9*Ci, zero padded)
// mode 1: dst (Ci, Kp)[ci][tap*Co+co] = src[co][ci][8-tap]                     input-gradient operand (Kp >= 9*Co)
__global__ void conv_weight_pack16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int Co, int Ci, int Kp, int mode) {
    const int rows = mode == 0 ? Co : Ci, inner = mode == 0 ? Ci : Co;
    const long total = (long)rows * Kp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)(i % Kp), r = (int)(i / Kp);
        float v = 0.f;
        if (k < 9 * inner) {
            const int tap = k / inner, c = k - tap * inner;
            v = mode == 0 ? src[((long)r * Ci + c) * 9 + tap] : src[((long)c * Ci + r) * 9 + (8 - tap)];
        }
        dst[i] = f2bf_bits(v);
    }
}

// dst (Co,Ci,3,3)[co][ci][tap] += src (Co, ld)[co][tap*Ci+ci]
__global__ void conv_weight_fold_kernel(const float* __restrict__ src, float* __restrict__ dst, int Co, int Ci, int ld) {
    const long total = (long)Co * Ci * 9;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tap = (int)(i % 9), ci = (int)((i / 9) % Ci), co = (int)(i / (9L * Ci));
        dst[i] += src[(long)co * ld + tap * Ci + ci];
    }
}

typedef __attribute__((ext_vector_type(8))) unsigned short us8;

// 2x2 max pooling, stride 2 (ceil mode: windows cut at the image edge) on bordered images, 8 channels per thread
__global__ void maxpool16_fwd_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, unsigned char* __restrict__ idx,
                                     int B, int T, int F, int C, int T2, int F2) {
    const int C8 = C >> 3;
    const long total = (long)B * (T2 + 2) * (F2 + 2) * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long m = i / C8;
        const int fp = (int)(m % (F2 + 2)), tp = (int)((m / (F2 + 2)) % (T2 + 2)), b = (int)(m / ((long)(F2 + 2) * (T2 + 2)));
        us8 o = {0, 0, 0, 0, 0, 0, 0, 0}, bi = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fp >= 1 && fp <= F2 && tp >= 1 && tp <= T2) {
            float best[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) best[j] = -INFINITY;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int t = 2 * (tp - 1) + (k >> 1), f = 2 * (fp - 1) + (k & 1);
                if (t < T && f < F) {
                    const us8 v = *reinterpret_cast<const us8*>(x + ((((long)b * (T + 2) + t + 1) * (F + 2) + f + 1) * C + 8 * c8));
#pragma unroll
                    for (int j = 0; j < 8; ++j) { const float fv = bfv(v[j]); if (fv > best[j]) { best[j] = fv; o[j] = v[j]; bi[j] = (unsigned short)k; } }
                }
            }
        }
        *reinterpret_cast<us8*>(y + m * C + 8 * c8) = o;
        unsigned long long pk = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) pk |= (unsigned long long)(bi[j] & 3) << (8 * j);
        *reinterpret_cast<unsigned long long*>(idx + m * C + 8 * c8) = pk;
    }
}

__global__ void maxpool16_bwd_kernel(const unsigned short* __restrict__ dy, const unsigned char* __restrict__ idx, unsigned short* __restrict__ dx,
                                     int B, int T, int F, int C, int T2, int F2) {
    const int C8 = C >> 3;
    const long total = (long)B * (T + 2) * (F + 2) * C8;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        const long m = i / C8;
        const int fp = (int)(m % (F + 2)), tp = (int)((m / (F + 2)) % (T + 2)), b = (int)(m / ((long)(F + 2) * (T + 2)));
        us8 o = {0, 0, 0, 0, 0, 0, 0, 0};
        if (fp >= 1 && fp <= F && tp >= 1 && tp <= T) {
            const int t = tp - 1, f = fp - 1, t2 = t >> 1, f2 = f >> 1;
            if (t2 < T2 && f2 < F2) {
                const long mo = (((long)b * (T2 + 2) + t2 + 1) * (F2 + 2) + f2 + 1) * C + 8 * c8;
                const us8 g = *reinterpret_cast<const us8*>(dy + mo);
                const unsigned long long pk = *reinterpret_cast<const unsigned long long*>(idx + mo);
                const unsigned me = (unsigned)(((t & 1) << 1) | (f & 1));
#pragma unroll
                for (int j = 0; j < 8; ++j) if (((pk >> (8 * j)) & 3ull) == me) o[j] = g[j];
            }
        }
        *reinterpret_cast<us8*>(dx + m * C + 8 * c8) = o;
    }
}

// CNNLayerNorm: LayerNorm over the F interior pixels of every (b, t, c) of a bordered image, affine per f, + ReLU; borders zero
__global__ __launch_bounds__(256) void ln_f16_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bia,
                                                         unsigned short* __restrict__ y, float* __restrict__ stats, int B, int T, int F, int C,
                                                         float eps, int relu) {
    const int T2 = T + 2, F2 = F + 2;
    const long total = (long)B * T2 * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long r = i / C;                                   // (b, tp)
        const int tp = (int)(r % T2);
        const float* xp = x + r * F2 * C + c;
        unsigned short* yp = y + r * F2 * C + c;
        if (tp == 0 || tp == T2 - 1) {
            for (int fp = 0; fp < F2; ++fp) yp[(long)fp * C] = 0;
            stats[2 * i] = 0.f; stats[2 * i + 1] = 0.f;
            continue;
        }
        float s = 0.f;
        for (int f = 1; f <= F; ++f) s += xp[(long)f * C];
        const float mean = s / F;
        float v = 0.f;
        for (int f = 1; f <= F; ++f) { const float d = xp[(long)f * C] - mean; v += d * d; }
        const float rstd = rsqrtf(v / F + eps);
        yp[0] = 0; yp[(long)(F2 - 1) * C] = 0;
        for (int f = 1; f <= F; ++f) {
            const float o = (xp[(long)f * C] - mean) * rstd * w[f - 1] + bia[f - 1];
            yp[(long)f * C] = f2bf_bits(relu ? fmaxf(o, 0.f) : o);
        }
        stats[2 * i] = mean;
        stats[2 * i + 1] = rstd;
    }
}

template <int FMAX>
__global__ __launch_bounds__(256) void ln_f16_bwd_kernel(const unsigned short* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bia, const float* __restrict__ stats, unsigned short* __restrict__ dx,
                                                         float* __restrict__ dw, float* __restrict__ db, float* __restrict__ dcb,
                                                         int B, int T, int F, int C, int relu) {
    // dcb: gradient of the bias of the convolution in front (per channel: the sum of dx over all pixels) - analytically ZERO
    // (the normalisation removes a constant over f), so it is summed here from the fp32 values; summed from the bf16-rounded dx
    // it would be rounding noise five orders of magnitude above the fp32 path's, which Adadelta normalises into real steps
    __shared__ float s_dw[FMAX], s_db[FMAX], s_cb[512];
    for (int f = threadIdx.x; f < F; f += 256) { s_dw[f] = 0.f; s_db[f] = 0.f; }
    for (int c = threadIdx.x; c < C && c < 512; c += 256) s_cb[c] = 0.f;
    __syncthreads();
    const int T2 = T + 2, F2 = F + 2;
    const long total = (long)B * T2 * C;
    for (long i0 = blockIdx.x * 256L; i0 < total; i0 += (long)gridDim.x * 256) {
        const long i = i0 + threadIdx.x;
        const bool in = i < total;
        const long r = in ? i / C : 0;
        const int c = in ? (int)(i % C) : 0;
        const int tp = (int)(r % T2);
        const bool ok = in && tp >= 1 && tp <= T;
        const float mean = ok ? stats[2 * i] : 0.f, rstd = ok ? stats[2 * i + 1] : 0.f;
        const float* xp = x + r * F2 * C + c;
        const unsigned short* gp = dy + r * F2 * C + c;
        float s1 = 0.f, s2 = 0.f;
        for (int f = 1; f <= F; ++f) {
            float g = 0.f, xh = 0.f;
            if (ok) {
                xh = (xp[(long)f * C] - mean) * rstd;
                g = bfv(gp[(long)f * C]);
                if (relu && (xh * w[f - 1] + bia[f - 1]) <= 0.f) g = 0.f;
            }
            s1 += g * w[f - 1];
            s2 += g * w[f - 1] * xh;
            const float gw = wave_sum(g * xh), gb = wave_sum(g);
            if ((threadIdx.x & 63) == 0) { atomicAdd(&s_dw[f - 1], gw); atomicAdd(&s_db[f - 1], gb); }
        }
        if (in) {
            unsigned short* dp = dx + r * F2 * C + c;
            dp[0] = 0; dp[(long)(F2 - 1) * C] = 0;
            s1 /= F; s2 /= F;
            float cb = 0.f;
            for (int f = 1; f <= F; ++f) {
                float o = 0.f;
                if (ok) {
                    const float xh = (xp[(long)f * C] - mean) * rstd;
                    float g = bfv(gp[(long)f * C]);
                    if (relu && (xh * w[f - 1] + bia[f - 1]) <= 0.f) g = 0.f;
                    o = rstd * (g * w[f - 1] - s1 - xh * s2);
                }
                cb += o;
                dp[(long)f * C] = f2bf_bits(o);
            }
            if (dcb && ok) atomicAdd(&s_cb[c], cb);
        }
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += 256) { atomicAdd(dw + f, s_dw[f]); atomicAdd(db + f, s_db[f]); }
    if (dcb) for (int c = threadIdx.x; c < C; c += 256) { const float v_ = s_cb[c]; if (v_ != 0.f) atomicAdd(dcb + c, v_); }
}

// bordered image P(T, F, C) -> encoder input (B, T, C*F) bf16, channel-major (reference: view of the NCHW tensor transposed);
// and its adjoint: gradient (B, T, C*F) bf16 -> bordered image (borders zero)
__global__ void vgg16_out_kernel(const unsigned short* __restrict__ img, unsigned short* __restrict__ out, int B, int T, int F, int C) {
    const long total = (long)B * T * C * F;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int f = (int)(i % F), c = (int)((i / F) % C);
        const long bt = i / ((long)F * C);
        const int t = (int)(bt % T), b = (int)(bt / T);
        out[i] = img[(((long)b * (T + 2) + t + 1) * (F + 2) + f + 1) * C + c];
    }
}
__global__ void vgg16_out_bwd_kernel(const unsigned short* __restrict__ dout, unsigned short* __restrict__ g, int B, int T, int F, int C) {
    const long total = (long)B * (T + 2) * (F + 2) * C;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long m = i / C;
        const int fp = (int)(m % (F + 2)), tp = (int)((m / (F + 2)) % (T + 2)), b = (int)(m / ((long)(F + 2) * (T + 2)));
        unsigned short v = 0;
        if (fp >= 1 && fp <= F && tp >= 1 && tp <= T) v = dout[((long)b * T + tp - 1) * ((long)C * F) + (long)c * F + fp - 1];
        g[i] = v;
    }
}

}  // namespace

extern "C" int asr_vgg16_im2col(const float* feature, void* x1, int B, int T, int F, int Cin, int Kp, asr_stream_t stream) {
    ASR_REQUIRE(feature && x1 && B > 0 && T > 0 && F > 0 && Cin > 0 && Kp >= 9 * Cin && Kp % 8 == 0, ASR_E_ARG, "asr_vgg16_im2col: bad args");
    hipLaunchKernelGGL(vgg16_im2col_kernel, dim3(grid_for((long)B * (T + 2) * (F + 2) * Kp)), dim3(256), 0, (hipStream_t)stream, feature,
                       (unsigned short*)x1, B, T, F, Cin, Kp);
    ASR_LAUNCH_CHECK("asr_vgg16_im2col");
    return ASR_OK;
}

extern "C" int asr_conv_weight_pack16(const float* src, void* dst, int Co, int Ci, int Kp, int mode, asr_stream_t stream) {
    ASR_REQUIRE(src && dst && Co > 0 && Ci > 0 && (mode == 0 || mode == 1) && Kp >= 9 * (mode == 0 ? Ci : Co), ASR_E_ARG, "asr_conv_weight_pack16: bad args");
    hipLaunchKernelGGL(conv_weight_pack16_kernel, dim3(grid_for((long)(mode == 0 ? Co : Ci) * Kp)), dim3(256), 0, (hipStream_t)stream, src,
                       (unsigned short*)dst, Co, Ci, Kp, mode);
    ASR_LAUNCH_CHECK("asr_conv_weight_pack16");
    return ASR_OK;
}

extern "C" int asr_conv_weight_fold(const float* src, float* dst, int Co, int Ci, int ld, asr_stream_t stream) {
    ASR_REQUIRE(src && dst && Co > 0 && Ci > 0 && ld >= 9 * Ci, ASR_E_ARG, "asr_conv_weight_fold: bad args");
    hipLaunchKernelGGL(conv_weight_fold_kernel, dim3(grid_for((long)Co * Ci * 9)), dim3(256), 0, (hipStream_t)stream, src, dst, Co, Ci, ld);
    ASR_LAUNCH_CHECK("asr_conv_weight_fold");
    return ASR_OK;
}

extern "C" int asr_conv3x3_16(const void* img, const void* w, void* out, const float* bias, int B, int T, int F, int C, int N, int K, int implicit,
                              int act, int out_f32, asr_stream_t stream) {
    ASR_REQUIRE(img && w && out && B > 0 && T > 0 && F > 0 && N > 0 && K > 0, ASR_E_ARG, "asr_conv3x3_16: bad args");
    const int rc = gemm16_conv3x3(img, w, out, bias, B, T, F, C, N, K, implicit, act, out_f32, (hipStream_t)stream);
    ASR_REQUIRE(rc != 1, ASR_E_UNSUPPORTED, "asr_conv3x3_16: shape not covered (C %% 64, K %% 8, N %% 8, 16-byte alignment): C=%d N=%d K=%d", C, N, K);
    return rc;
}

// dw (N, ldw)[n][tap*C + ci] += sum over the bordered pixel rows of dout[row, n] * img[row + shift(tap), ci]
extern "C" int asr_conv3x3_16_wgrad(const void* img, const void* dout, float* dw, int B, int T, int F, int C, int N, int ldw, int splits,
                                    asr_stream_t stream) {
    ASR_REQUIRE(img && dout && dw && B > 0 && T > 0 && F > 0 && C % 8 == 0 && N % 8 == 0 && ldw >= 9 * C, ASR_E_ARG, "asr_conv3x3_16_wgrad: bad args");
    const long M = (long)B * (T + 2) * (F + 2);
    ASR_REQUIRE(M < (1L << 31), ASR_E_ARG, "asr_conv3x3_16_wgrad: image too large");
    // ONE launch for the nine taps (tap = a grid dimension of gemm16_tn_kernel): 9 x splits x tiles workgroups fill the chip with
    // a ninth of the split-K atomics nine separate launches would need; shifted rows outside the image read as zeros
    const int rc = gemm16_tn_taps(dout, img, dw, N, C, (int)M, N, C, ldw, splits, 0, 0, 0, 0, F + 2, (hipStream_t)stream);
    ASR_REQUIRE(rc != 1, ASR_E_UNSUPPORTED, "asr_conv3x3_16_wgrad: shape not covered by the bf16 TN contraction");
    return rc;
}

extern "C" int asr_maxpool2x2_16_fwd(const void* x, void* y, unsigned char* idx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream) {
    ASR_REQUIRE(x && y && idx && B > 0 && T > 0 && F > 0 && C > 0 && C % 8 == 0 && T2 > 0 && F2 > 0, ASR_E_ARG, "asr_maxpool2x2_16_fwd: bad args");
    ASR_REQUIRE(2 * T2 - 1 <= T && 2 * F2 - 1 <= F, ASR_E_ARG, "asr_maxpool2x2_16_fwd: output larger than ceil(T/2) x ceil(F/2)");
    hipLaunchKernelGGL(maxpool16_fwd_kernel, dim3(grid_for((long)B * (T2 + 2) * (F2 + 2) * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)x, (unsigned short*)y, idx, B, T, F, C, T2, F2);
    ASR_LAUNCH_CHECK("asr_maxpool2x2_16_fwd");
    return ASR_OK;
}

extern "C" int asr_maxpool2x2_16_bwd(const void* dy, const unsigned char* idx, void* dx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream) {
    ASR_REQUIRE(dy && idx && dx && B > 0 && T > 0 && F > 0 && C > 0 && C % 8 == 0, ASR_E_ARG, "asr_maxpool2x2_16_bwd: bad args");
    hipLaunchKernelGGL(maxpool16_bwd_kernel, dim3(grid_for((long)B * (T + 2) * (F + 2) * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)dy, idx, (unsigned short*)dx, B, T, F, C, T2, F2);
    ASR_LAUNCH_CHECK("asr_maxpool2x2_16_bwd");
    return ASR_OK;
}

extern "C" int asr_ln_freq16_fwd(const float* x, const float* w, const float* b, void* y, float* stats, int B, int T, int F, int C, float eps, int relu,
                                 asr_stream_t stream) {
    ASR_REQUIRE(x && w && b && y && stats && B > 0 && T > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_ln_freq16_fwd: bad args");
    hipLaunchKernelGGL(ln_f16_fwd_kernel, dim3(grid_for((long)B * (T + 2) * C)), dim3(256), 0, (hipStream_t)stream, x, w, b,
                       (unsigned short*)y, stats, B, T, F, C, eps, relu);
    ASR_LAUNCH_CHECK("asr_ln_freq16_fwd");
    return ASR_OK;
}

extern "C" int asr_ln_freq16_bwd(const void* dy, const float* x, const float* w, const float* b, const float* stats, void* dx, float* dw, float* db,
                                 float* dconv_bias, int B, int T, int F, int C, int relu, asr_stream_t stream) {
    ASR_REQUIRE(dy && x && w && b && stats && dx && dw && db && B > 0 && T > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_ln_freq16_bwd: bad args");
    ASR_REQUIRE(F <= 128 && C <= 512, ASR_E_UNSUPPORTED, "asr_ln_freq16_bwd: F=%d > 128 or C=%d > 512", F, C);
    long g = ((long)B * (T + 2) * C + 255) / 256; if (g > 2048) g = 2048;
    hipLaunchKernelGGL(ln_f16_bwd_kernel<128>, dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)dy, x, w, b,
                       stats, (unsigned short*)dx, dw, db, dconv_bias, B, T, F, C, relu);
    ASR_LAUNCH_CHECK("asr_ln_freq16_bwd");
    return ASR_OK;
}

extern "C" int asr_vgg16_output(const void* img, void* out, int B, int T, int F, int C, asr_stream_t stream) {
    ASR_REQUIRE(img && out && B > 0 && T > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_vgg16_output: bad args");
    hipLaunchKernelGGL(vgg16_out_kernel, dim3(grid_for((long)B * T * F * C)), dim3(256), 0, (hipStream_t)stream, (const unsigned short*)img,
                       (unsigned short*)out, B, T, F, C);
    ASR_LAUNCH_CHECK("asr_vgg16_output");
    return ASR_OK;
}

extern "C" int asr_vgg16_output_bwd(const void* dout, void* g, int B, int T, int F, int C, asr_stream_t stream) {
    ASR_REQUIRE(dout && g && B > 0 && T > 0 && F > 0 && C > 0, ASR_E_ARG, "asr_vgg16_output_bwd: bad args");
    hipLaunchKernelGGL(vgg16_out_bwd_kernel, dim3(grid_for((long)B * (T + 2) * (F + 2) * C)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)dout, (unsigned short*)g, B, T, F, C);
    ASR_LAUNCH_CHECK("asr_vgg16_output_bwd");
    return ASR_OK;
}
