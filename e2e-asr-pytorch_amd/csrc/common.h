// Shared device/host helpers for libasr_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include "../../include/asr_hip.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define WAVE 64

// ---- error plumbing (thread-local message, negative return codes) -------------------------
void asr_set_error(const char* fmt, ...);

#define ASR_REQUIRE(cond, code, ...)                                          \
    do {                                                                      \
        if (!(cond)) {                                                        \
            asr_set_error(__VA_ARGS__);                                       \
            return (code);                                                    \
        }                                                                     \
    } while (0)

#define ASR_LAUNCH_CHECK(name)                                                \
    do {                                                                      \
        hipError_t e_ = hipGetLastError();                                    \
        if (e_ != hipSuccess) {                                               \
            asr_set_error("%s: launch failed: %s", name, hipGetErrorString(e_)); \
            return ASR_E_LAUNCH;                                              \
        }                                                                     \
    } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// true the first time it is called for the CURRENT device with this flag array: kernel attributes (dynamic-LDS limits) are
// per device, so the "already set" marks are too (a process-wide bool would leave the second GPU of a process unset).  The
// marks are write-once and the guarded calls idempotent: two threads racing here set the attribute twice, nothing else.
static inline bool first_on_device(unsigned char (&done)[32]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return true;
    dev &= 31;
    if (done[dev]) return false;
    done[dev] = 1;
    return true;
}

// ---- device math --------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ unsigned short f2bf_bits(float f) {
    __bf16 v = (__bf16)f;
    return __builtin_bit_cast(unsigned short, v);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// log(exp(a)+exp(b)) that tolerates -inf on either side.
__device__ __forceinline__ float logaddexpf_(float a, float b) {
    float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + log1pf(expf(-fabsf(a - b)));
}

// MFMA wrappers: 16x16 output tile, f32 accumulate.
//   bf16: K-step 32, lane l holds A[row l&15][k = 8*(l>>4)+j], B[k = 8*(l>>4)+j][col l&15], j=0..7
//   f32 : K-step 4,  lane l holds A[row l&15][k = l>>4],       B[k = l>>4][col l&15]
//   C/D : col = l&15, row = 4*(l>>4) + reg
__device__ __forceinline__ f32x4 mma16(bf16x8 a, bf16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 mma16(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Philox4x32-10 counter RNG (one call -> 4 x u32). Used for dropout masks.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// keep(element idx) for dropout probability p: uniform u32 >= p * 2^32
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    uint32_t r[4];
    uint64_t blk = idx >> 2;
    philox4x32((uint32_t)blk, (uint32_t)(blk >> 32), 0u, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), r);
    return r[idx & 3] >= thresh;
}

// ---- small-M MFMA row dot products (recurrent / per-step matvecs) ------------------------------
// These kernels are latency-bound, so the loads of U k-steps must be in flight TOGETHER.  A load under a lane-dependent
// condition is compiled into a branch plus its own `s_waitcnt vmcnt(0)` (U serialized round trips), and so is a load
// whose value is converted in the same basic block behind a uniform branch; therefore: one uniform branch on `vec`
// around the whole batch, every load unconditional from a clamped address, zeroing by selects after the batch.
struct Raw8 { float4 lo, hi; };
// rows 16-byte aligned, kend % 4 == 0
__device__ __forceinline__ Raw8 load8_raw(const float* p, int k, int kend) {
    Raw8 r;
    r.lo = *reinterpret_cast<const float4*>(p + max(min(k, kend - 4), 0));
    r.hi = *reinterpret_cast<const float4*>(p + max(min(k + 4, kend - 4), 0));
    return r;
}
__device__ __forceinline__ bf16x8 cvt8(const Raw8& r, int k, int kend, bool ok) {
    const bool oa = ok && k < kend, ob = ok && k + 4 < kend;
    bf16x8 v;
    v[0] = (__bf16)(oa ? r.lo.x : 0.f); v[1] = (__bf16)(oa ? r.lo.y : 0.f); v[2] = (__bf16)(oa ? r.lo.z : 0.f); v[3] = (__bf16)(oa ? r.lo.w : 0.f);
    v[4] = (__bf16)(ob ? r.hi.x : 0.f); v[5] = (__bf16)(ob ? r.hi.y : 0.f); v[6] = (__bf16)(ob ? r.hi.z : 0.f); v[7] = (__bf16)(ob ? r.hi.w : 0.f);
    return v;
}
// generic (unaligned / ragged) rows: scalar loads from clamped addresses
__device__ __forceinline__ bf16x8 load8_slow(const float* p, int k, int kend, bool ok) {
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x = p[max(min(k + j, kend - 1), 0)];
        v[j] = (__bf16)((ok && k + j < kend) ? x : 0.f);
    }
    return v;
}

// acc += [a1 | a2] . [b1 | b2] over the k-steps ks_beg, ks_beg + ks_stride, ... of the concatenated reduction
// (segment 2 optional: K2 = 0).  Lane (l&15) addresses row `a*` of A and column-row `b*` of B, both K-contiguous.
template <bool BF16, int U>
__device__ __forceinline__ f32x4 dot_rows_cat(const float* a1, const float* b1, int K1, bool vec1,
                                              const float* a2, const float* b2, int K2, bool vec2,
                                              bool aok, bool bok, int ks_beg, int ks_stride, f32x4 acc) {
    const int q = (threadIdx.x & 63) >> 4;
    if (BF16) {
        const int n1 = (K1 + 31) >> 5, n2 = (K2 + 31) >> 5, nks = n1 + n2;
        if (vec1 && (vec2 || K2 == 0)) {
            for (int ks0 = ks_beg; ks0 < nks; ks0 += U * ks_stride) {
                Raw8 ra[U], rb[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ks = ks0 + u * ks_stride;
                    const bool s2 = ks >= n1 && K2 > 0;
                    const int k = (s2 ? ks - n1 : min(ks, n1 - 1)) * 32 + 8 * q;
                    ra[u] = load8_raw(s2 ? a2 : a1, k, s2 ? K2 : K1);
                    rb[u] = load8_raw(s2 ? b2 : b1, k, s2 ? K2 : K1);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ks = ks0 + u * ks_stride;
                    const bool in = ks < nks, s2 = ks >= n1 && K2 > 0;
                    const int k = (s2 ? ks - n1 : min(ks, n1 - 1)) * 32 + 8 * q;
                    acc = mma16(cvt8(ra[u], k, s2 ? K2 : K1, aok && in), cvt8(rb[u], k, s2 ? K2 : K1, bok && in), acc);
                }
            }
        } else {
            for (int ks0 = ks_beg; ks0 < nks; ks0 += U * ks_stride) {
                bf16x8 a[U], b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ks = ks0 + u * ks_stride;
                    const bool in = ks < nks, s2 = ks >= n1 && K2 > 0;
                    const int k = (s2 ? ks - n1 : min(ks, n1 - 1)) * 32 + 8 * q;
                    a[u] = load8_slow(s2 ? a2 : a1, k, s2 ? K2 : K1, aok && in);
                    b[u] = load8_slow(s2 ? b2 : b1, k, s2 ? K2 : K1, bok && in);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc = mma16(a[u], b[u], acc);
            }
        }
    } else {
        const int n1 = (K1 + 3) >> 2, n2 = (K2 + 3) >> 2, nks = n1 + n2;
        constexpr int UF = 4 * U;
        for (int ks0 = ks_beg; ks0 < nks; ks0 += UF * ks_stride) {
            float a[UF], b[UF];
#pragma unroll
            for (int u = 0; u < UF; ++u) {
                const int ks = ks0 + u * ks_stride;
                const bool in = ks < nks, s2 = ks >= n1 && K2 > 0;
                const int k = (s2 ? ks - n1 : min(ks, n1 - 1)) * 4 + q;
                const int K = s2 ? K2 : K1;
                const float av = (s2 ? a2 : a1)[max(min(k, K - 1), 0)], bv = (s2 ? b2 : b1)[max(min(k, K - 1), 0)];
                a[u] = (aok && in && k < K) ? av : 0.f;
                b[u] = (bok && in && k < K) ? bv : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UF; ++u) acc = mma16(a[u], b[u], acc);
        }
    }
    return acc;
}

// single-segment form
template <bool BF16, int U = 8>
__device__ __forceinline__ f32x4 dot_rows(const float* arow, bool aok, const float* brow, bool bok, int K,
                                          int ks_beg, int ks_stride, bool vec, f32x4 acc) {
    return dot_rows_cat<BF16, U>(arow, brow, K, vec, arow, brow, 0, vec, aok, bok, ks_beg, ks_stride, acc);
}
