// Post-backward step on the flat parameter / gradient buffers (all parameters of the model live in one
// contiguous fp32 buffer, gradients in another — also the RCCL all-reduce bucket):
//   asr_sumsq        : sum of squares (for the global L2 norm of clip_grad_norm_, src/solver.py:97)
//   asr_scale        : in-place scale (averaging after the gradient all-reduce)
//   asr_adadelta_step: clip-by-global-norm + NaN guard + torch.optim.Adadelta update fused, reading the
//                      squared norm from device memory so the host never synchronises
//                      (src/solver.py:96-103 + src/optim.py:29,53-54)
// Pure HBM streaming: 12-19 M parameters, 4 tensors read + 3 written per step.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long n, double* __restrict__ out) {
    __shared__ double s4[4];
    double acc = 0.0;
    const long n4 = n >> 2;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    // four loads in flight per lane and four independent fp64 chains (one load + a 4-deep dependent chain per trip left
    // the 48 MB read at 1.5 TB/s)
    const long stride = (long)gridDim.x * blockDim.x;
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
        acc += ((double)v0.x * v0.x + (double)v0.y * v0.y) + ((double)v0.z * v0.z + (double)v0.w * v0.w);
        a1 += ((double)v1.x * v1.x + (double)v1.y * v1.y) + ((double)v1.z * v1.z + (double)v1.w * v1.w);
        a2 += ((double)v2.x * v2.x + (double)v2.y * v2.y) + ((double)v2.z * v2.z + (double)v2.w * v2.w);
        a3 += ((double)v3.x * v3.x + (double)v3.y * v3.y) + ((double)v3.z * v3.z + (double)v3.w * v3.w);
    }
    for (; i < n4; i += stride) {
        const float4 v = x4[i];
        acc += ((double)v.x * v.x + (double)v.y * v.y) + ((double)v.z * v.z + (double)v.w * v.w);
    }
    acc = (acc + a1) + (a2 + a3);
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float v = x[(n4 << 2) + threadIdx.x]; acc += (double)v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) s4[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, s4[0] + s4[1] + s4[2] + s4[3]);
}

__global__ void scale_kernel(float* x, long n, float k) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= k;
}

__global__ __launch_bounds__(256) void adadelta_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                       float* __restrict__ sq, float* __restrict__ ad, long n,
                                                       float lr, float rho, float eps, float wd, float clip,
                                                       const double* __restrict__ normsq, float gmul,
                                                       const unsigned* __restrict__ status) {
    // a persistent launch of this step gave up a hand-off (asr_status_collect): its results are not to be trusted,
    // the update is refused just like a NaN gradient; the word stays set until the host has read and cleared it
    if (status && *status != 0u) return;
    // gmul: extra factor applied to the stored gradient before clipping (1/world_size when the buffer holds a sum)
    float coef = gmul;
    if (normsq) {
        const double nrm = sqrt(*normsq) * (double)gmul;
        if (!(nrm == nrm) || nrm == INFINITY) return;      // NaN/inf gradient norm: skip the update (src/solver.py:99-103)
        if (clip > 0.f) {
            const double c = (double)clip / (nrm + 1e-6);
            if (c < 1.0) coef *= (float)c;
        }
    }
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float s = rho * sq[i] + (1.f - rho) * gi * gi;
        const float a = ad[i];
        const float delta = sqrtf(a + eps) / sqrtf(s + eps) * gi;
        sq[i] = s;
        ad[i] = rho * a + (1.f - rho) * delta * delta;
        p[i] = pi - lr * delta;
    }
}

// torch.optim.Adam (L2 weight decay, optional amsgrad) with the same clip / NaN guard / status refusal in front
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float* __restrict__ vmax, long n, float lr, float b1, float b2,
                                                   float eps, float wd, int step_host, const unsigned long long* __restrict__ step_dev, float clip,
                                                   const double* __restrict__ normsq, float gmul, const unsigned* __restrict__ status) {
    if (status && *status != 0u) return;
    // bias corrections in double (1 - beta^step in float32 is off by ~6e-5 relative at step 1); the step is the number of
    // APPLIED updates so far + 1: with a device counter (step_dev) a refused update does not advance it, as in torch
    const double stepd = step_dev ? (double)(*step_dev + 1ull) : (double)step_host;
    const float bc1 = (float)(1.0 - pow((double)b1, stepd)), bc2_sqrt = (float)sqrt(1.0 - pow((double)b2, stepd));
    float coef = gmul;
    if (normsq) {
        const double nrm = sqrt(*normsq) * (double)gmul;
        if (!(nrm == nrm) || nrm == INFINITY) return;
        if (clip > 0.f) {
            const double c = (double)clip / (nrm + 1e-6);
            if (c < 1.0) coef *= (float)c;
        }
    }
    const float step_size = lr / bc1;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] * coef;
        const float pi = p[i];
        if (wd != 0.f) gi += wd * pi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        if (vmax) { const float vm = fmaxf(vmax[i], vi); vmax[i] = vm; vi = vm; }
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

// advances the device-side count of applied Adam updates under exactly the guard of adam_kernel
__global__ void adam_count_kernel(unsigned long long* step_dev, const double* __restrict__ normsq, float gmul, const unsigned* __restrict__ status) {
    if (status && *status != 0u) return;
    if (normsq) { const double nrm = sqrt(*normsq) * (double)gmul; if (!(nrm == nrm) || nrm == INFINITY) return; }
    *step_dev = *step_dev + 1ull;
}

// ORs "abort word != 0" of up to 32 persistent-launch workspaces into the sticky status word
struct CollectP { const unsigned* w[32]; int n; };
__global__ void status_collect_kernel(CollectP p, unsigned* status) {
    const int i = threadIdx.x;
    if (i < p.n && p.w[i] != nullptr) {
        const unsigned v = __hip_atomic_load(p.w[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != 0u) atomicOr(status, 1u << (i & 31));
    }
}

inline int grid_for(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g)); }

}  // namespace

extern "C" int asr_sumsq(const float* x, long n, double* out, asr_stream_t stream) {
    ASR_REQUIRE(x && out && n > 0, ASR_E_ARG, "asr_sumsq: bad args");
    ASR_REQUIRE(((uintptr_t)x & 15) == 0, ASR_E_ARG, "asr_sumsq: x must be 16B aligned");
    hipStream_t st = (hipStream_t)stream;
    hipMemsetAsync(out, 0, sizeof(double), st);
    // 512 workgroups: every workgroup ends in one fp64 atomic on the same address, and 2048 of them were most of the kernel's time
    hipLaunchKernelGGL(sumsq_kernel, dim3(std::min(grid_for(n / 4 + 1), 512)), dim3(256), 0, st, x, n, out);
    ASR_LAUNCH_CHECK("asr_sumsq");
    return ASR_OK;
}

namespace {
// out[i] = in[i] * alpha[0] with the factor in DEVICE memory: the backward of a loss kernel under the loss mix (grad_output is a
// device scalar; torch's `grad * gout` was the last elementwise ATen arithmetic of the step)
__global__ void scale_dev_kernel(const float* __restrict__ in, float* __restrict__ out, long n, const float* __restrict__ alpha) {
    const float a = alpha[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i] * a;
}
// out[0] = a[0] * wa[0] (+ b[0] * wb[0]): total = w ctc + (1 - w) att (bin/train_asr.py:246) and its backward, scalars on the device
__global__ void loss_mix_kernel(const float* a, const float* wa, const float* b, const float* wb, float* out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = a[0] * wa[0] + (b ? b[0] * wb[0] : 0.f);
}
}  // namespace
extern "C" int asr_scale_dev(const float* in, float* out, long n, const float* alpha, asr_stream_t stream) {
    ASR_REQUIRE(in && out && alpha && n > 0, ASR_E_ARG, "asr_scale_dev: bad args");
    hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, in, out, n, alpha);
    ASR_LAUNCH_CHECK("asr_scale_dev");
    return ASR_OK;
}
extern "C" int asr_loss_mix(const float* a, const float* wa, const float* b, const float* wb, float* out, asr_stream_t stream) {
    ASR_REQUIRE(a && wa && out && (!b || wb), ASR_E_ARG, "asr_loss_mix: bad args");
    hipLaunchKernelGGL(loss_mix_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, wa, b, wb, out);
    ASR_LAUNCH_CHECK("asr_loss_mix");
    return ASR_OK;
}
extern "C" int asr_scale(float* x, long n, float k, asr_stream_t stream) {
    ASR_REQUIRE(x && n > 0, ASR_E_ARG, "asr_scale: bad args");
    hipLaunchKernelGGL(scale_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, n, k);
    ASR_LAUNCH_CHECK("asr_scale");
    return ASR_OK;
}

extern "C" int asr_adadelta_step(float* param, const float* grad, float* square_avg, float* acc_delta, long n,
                                 float lr, float rho, float eps, float weight_decay, float clip,
                                 const double* normsq, float grad_mul, const unsigned* status, asr_stream_t stream) {
    ASR_REQUIRE(param && grad && square_avg && acc_delta && n > 0, ASR_E_ARG, "asr_adadelta_step: bad args");
    hipLaunchKernelGGL(adadelta_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, square_avg,
                       acc_delta, n, lr, rho, eps, weight_decay, clip, normsq, grad_mul, status);
    ASR_LAUNCH_CHECK("asr_adadelta_step");
    return ASR_OK;
}

extern "C" int asr_status_collect(const void* const* abort_words, int n, unsigned* status, asr_stream_t stream) {
    ASR_REQUIRE(abort_words && status && n >= 0 && n <= 32, ASR_E_ARG, "asr_status_collect: bad args (n = %d, at most 32)", n);
    if (n == 0) return ASR_OK;
    CollectP p;
    for (int i = 0; i < 32; ++i) p.w[i] = i < n ? (const unsigned*)abort_words[i] : nullptr;
    p.n = n;
    hipLaunchKernelGGL(status_collect_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, p, status);
    ASR_LAUNCH_CHECK("asr_status_collect");
    return ASR_OK;
}


extern "C" int asr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int step, float clip,
                             const double* normsq, float grad_mul, const unsigned* status, unsigned long long* step_counter,
                             asr_stream_t stream) {
    ASR_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && (step >= 1 || step_counter), ASR_E_ARG, "asr_adam_step: bad args");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, max_exp_avg_sq, n,
                       lr, beta1, beta2, eps, weight_decay, step, (const unsigned long long*)step_counter, clip, normsq, grad_mul, status);
    if (step_counter) hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_counter, normsq, grad_mul, status);
    ASR_LAUNCH_CHECK("asr_adam_step");
    return ASR_OK;
}
