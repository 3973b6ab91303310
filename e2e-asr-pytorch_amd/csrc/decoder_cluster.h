// Shared pieces of the cluster-per-utterance decoder kernels (decoder_persist.hip: tiles resident in LDS; decoder_stream.hip:
// tiles streamed from L2 / HBM): tagged exchange granules, DPP wave reductions, the compute-wave barrier, residency check,
// the device-resident launch epoch.  Everything is `static` per translation unit.
#pragma once
#include "common.h"
#include "handoff.h"
#include <stdlib.h>
#include <algorithm>

namespace {


// Tag = 6 bits in the three mantissa LSBs of both floats of a granule: 2-bit step sequence + 4-bit launch epoch.  (With a
// 2-bit tag any stale or foreign 8 bytes pass the check with probability 1/4; recycled allocator memory whose old lines
// still sit in this XCD's L2 did exactly that in the first step of a launch.)  Payload loses 3 of 24 mantissa bits.
constexpr u64 PAIR_MASK = 7ull | (7ull << 32);
__device__ __forceinline__ u64 pair_want(unsigned seq, unsigned epoch) {
    // the epoch field takes the values 2..15 only: then BOTH words of a granule carry non-zero tag bits (low: seq != 0, high:
    // epoch >> 1 != 0).  With epoch 0 / 1 the high word's tag was 0, and any 8 bytes whose second word ends in three zero bits
    // and whose first word ends in the step tag passed - an int64 token id or length (5 = 0x0000000000000005) is exactly a
    // valid "zero payload" granule of (epoch 1, first step).  Seen on first launches (epoch 1) on memory recycled from such
    // tensors: whole records accepted as zeros, attention rows of 1e30.
    epoch = 2u + epoch % 14u;
    const unsigned tag = ((epoch & 15u) << 2) | seq;
    return (u64)(tag & 7u) | ((u64)(tag >> 3) << 32);
}
__device__ __forceinline__ u64 pack2(float a, float b, u64 want) {
    return ((u64)(__float_as_uint(a) & ~7u) | ((u64)(__float_as_uint(b) & ~7u) << 32)) | want;
}
__device__ __forceinline__ float lo_f(u64 g) { return __uint_as_float((unsigned)g & ~7u); }
__device__ __forceinline__ float hi_f(u64 g) { return __uint_as_float((unsigned)(g >> 32) & ~7u); }
// four values as IEEE halves in one granule (the context partial sums of the forward records, |x| <= tile frames): the tag keeps
// its place - bits 0..2 of both words, i.e. the three mantissa LSBs of values 0 and 2 (they keep 8 significant bits, like a
// bf16; values 1 and 3 all 11) - so gather16 / poll_copy and their masks are unchanged.  A staged word (tag bits cleared by the
// copy) holds two values: h2f_lo / h2f_hi.
__device__ __forceinline__ unsigned f2h_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (_Float16)x); }
__device__ __forceinline__ u64 pack4h(float a, float b, float c, float d, u64 want) {
    const unsigned lo = (f2h_bits(a) & ~7u) | (f2h_bits(b) << 16), hi = (f2h_bits(c) & ~7u) | (f2h_bits(d) << 16);
    return ((u64)lo | ((u64)hi << 32)) | want;
}
__device__ __forceinline__ float h2f_lo(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w & 0xffffu)); }
__device__ __forceinline__ float h2f_hi(unsigned w) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(w >> 16)); }
constexpr float NEG_BIG = -1e30f;       // masked energy in the exchange records (-inf would turn into NaN under the tag bit)
constexpr int NCW = 8, NPW = 4;         // compute waves, polling waves
constexpr int FSW_NU = 3;               // 16-column units of the forward energy sweep per compute wave (A <= 384)
constexpr int FCVX_LD = 32;             // row of the split-bf16 conv tile = the K slots of one MFMA
#ifdef ASR_DIAG
#define DP_DECL unsigned long long dg_t = __builtin_amdgcn_s_memrealtime(), dg_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define DP_MARK(k) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); dg_acc[k] += n_ - dg_t; dg_t = n_; __builtin_amdgcn_sched_barrier(0); }
#define DP_DUMP { if (blockIdx.x == 0 && threadIdx.x == 0) { unsigned long long* o = (unsigned long long*)p.status + 128; for (int k = 0; k < 16; ++k) o[k] = dg_acc[k]; } }
#define DP_JIT(k)
#elif defined(ASR_JITTER)
// Race-detector build (`make jitter`, tools/jitter_dec.py): every phase boundary of both roles sleeps for a pseudo-random time
// that depends on (workgroup, wave, step, boundary, launch epoch), so each launch runs under a different interleaving of its
// waves and workgroups.  A result that changes with the jitter is an ordering bug (a missing barrier, a slot reused too early).
__device__ __forceinline__ void dp_jitter(unsigned k, unsigned step, unsigned epoch) {
    unsigned h = (blockIdx.x * 0x9E3779B1u) ^ ((threadIdx.x >> 6) * 0x85EBCA6Bu) ^ (k * 0xC2B2AE35u) ^ (step * 0x27D4EB2Fu) ^ (epoch * 0x165667B1u);
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    h = __builtin_amdgcn_readfirstlane(h);
    if ((h & 3u) == 0u) {                                  // one boundary in four: up to ~8 us
        const unsigned n = (h >> 2) & 63u;
        for (unsigned i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(4);
    }
}
#define DP_DECL
#define DP_MARK(k) dp_jitter(k, (unsigned)t, epoch_);
#define DP_JIT(k) dp_jitter(32 + k, (unsigned)t, epoch_);
#define DP_DUMP
#else
#define DP_DECL
#define DP_MARK(k)
#define DP_JIT(k)
#define DP_DUMP
#endif
constexpr int RB = 5;                   // gate rows per batch of the cell contraction
inline size_t align_up256(size_t x) { return (x + 255) & ~(size_t)255; }

__device__ __forceinline__ float bf2f_(unsigned short x) { return __uint_as_float((unsigned)x << 16); }
__device__ __forceinline__ float tanh_f(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }
__device__ __forceinline__ float sigm_f(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }

// wave-wide sum on the DPP path (quad swaps, mirrors, row broadcasts: 6 VALU steps, no LDS permutes); uniform result
__device__ __forceinline__ float wave_sum_dpp(float v) {
#define DPB_STEP(CTRL, RMASK) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false));
    DPB_STEP(0xB1, 0xf)      // quad_perm [1,0,3,2]
    DPB_STEP(0x4E, 0xf)      // quad_perm [2,3,0,1]
    DPB_STEP(0x141, 0xf)     // row_half_mirror
    DPB_STEP(0x140, 0xf)     // row_mirror
    DPB_STEP(0x142, 0xa)     // row_bcast15 -> rows 1, 3
    DPB_STEP(0x143, 0xc)     // row_bcast31 -> rows 2, 3
#undef DPB_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// sum over aligned groups of 8 lanes (every lane of the group gets it): quad swaps + half-row mirror, no LDS permutes
__device__ __forceinline__ float sum8_dpp(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, false));
    return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
#define DPB_STEP(CTRL, RMASK) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, v), __builtin_bit_cast(int, v), CTRL, RMASK, 0xf, false)));
    DPB_STEP(0xB1, 0xf)
    DPB_STEP(0x4E, 0xf)
    DPB_STEP(0x141, 0xf)
    DPB_STEP(0x140, 0xf)
    DPB_STEP(0x142, 0xa)
    DPB_STEP(0x143, 0xc)
#undef DPB_STEP
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

struct PD {
    asr_dec_dims_t d;
    asr_dec_weights_t w;
    asr_dec_state_t s;
    const float* enc;
    const int64_t* enc_len;
    const unsigned short* wcat16;   // (4Dd, KCP) bf16 rows [W_ih[:, Dd:Dd+E] | W_hh | 0-pad]
    const float* embproj;           // (B*L, 4Dd)  W_ih[:, :Dd] . emb(token)
    u64* xbuf;
    unsigned* status;
    int NT, TE, UPW, QPW, CPW;      // tiles per utterance, frames per tile, hidden units / query outputs / context columns per workgroup
    int HG2, QG2, SG2;              // granules per producer record (even)
    int KC, KCP;                    // E + Dd, padded to a multiple of 8
    int allow_local;
    unsigned epoch;                 // launch counter (tag bits)
};

// ---- location convolution on the matrix cores -----------------------------------------------------------------------------------
// conv[k][tau] = sum_jj W_conv[k][jj] * att[tau + jj - Ks] (reference src/module.py:1161-1189, Conv1d(1, Kn, 2 Ks + 1, padding Ks)) is a
// Toeplitz product: for a 16-frame tile, A[m][jj] = P[16 mt + m + jj] (P = the zero-padded attention row from the tile's first frame
// on) times B[jj][k] = W_conv[k][jj].  The A fragment of lane (m, q) is 8 CONSECUTIVE elements of P starting at an offset whose low
// two bits are m & 3, so the bf16 image of P is kept in four copies shifted by 0..3 elements: every fragment is then two aligned
// 8-byte LDS reads.  Precision: both operands are split hi + lo (bf16 + bf16 of the remainder) and three products are kept
// (hi hi, hi lo, lo hi): ~2^-16 relative, the fp32 VALU loop it replaces needed 16 FMAs per three 16-byte LDS reads and was bound
// by the LDS port and the FMA rate (16.7 us per step on the 384-frame tiles of config 5, tools/diag_dec_stream.py).
struct ConvGeo { int NKS, WKP, IMG_LD, need; };
__host__ __device__ inline ConvGeo conv_geo(int TEB, int Ks) {
    ConvGeo g;
    g.NKS = (2 * Ks + 1 + 31) >> 5;                       // 32-tap steps of the contraction
    g.WKP = 32 * g.NKS + 8;                               // row of the filter image: 2 WKP bytes = an odd multiple of 16 modulo 256
    g.need = ((TEB + 15) & ~15) + 32 * g.NKS + 8;         // window elements a tile can touch
    g.IMG_LD = ((g.need - 32 + 127) / 128) * 128 + 32;    // 2 IMG_LD = 64 modulo 256: the four copies sit in disjoint banks
    return g;
}
__host__ __device__ inline int conv_img_shorts(const ConvGeo& g) { return 8 * g.IMG_LD + 32 * g.WKP; }     // {hi, lo} x 4 copies | {hi, lo} x 16 rows

// filter image: rows n < Kn hold W_conv[n][0 .. taps) split hi | lo, everything else zero (once per launch)
__device__ inline void conv_build_wimg(const float* __restrict__ Wconv, int Kn, int taps, const ConvGeo& g, unsigned short* wimg, int tid, int nthr) {
    for (int i = tid; i < 16 * g.WKP; i += nthr) {
        const int n = i / g.WKP, x = i - n * g.WKP;
        const float v = (n < Kn && x < taps) ? Wconv[n * taps + x] : 0.f;
        const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
        wimg[i] = __builtin_bit_cast(unsigned short, hi);
        wimg[16 * g.WKP + i] = __builtin_bit_cast(unsigned short, lo);
    }
}

// window image of the step: src[i] for 0 <= i < navail (zero beyond), copy c holds element i at position i - c
__device__ inline void conv_build_ximg(const float* src, int navail, const ConvGeo& g, unsigned short* ximg, int tz, int nthr) {
    const int ngrp = g.IMG_LD >> 2;
    for (int it = tz; it < 4 * ngrp; it += nthr) {
        const int c = it / ngrp, gq = it - c * ngrp;
        unsigned short h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * gq + c + e;
            const float v = (i < navail && i < g.need) ? src[i] : 0.f;
            const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
            h[e] = __builtin_bit_cast(unsigned short, hi); l[e] = __builtin_bit_cast(unsigned short, lo);
        }
        *reinterpret_cast<uint2*>(ximg + (long)c * g.IMG_LD + 4 * gq) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        *reinterpret_cast<uint2*>(ximg + (long)(4 + c) * g.IMG_LD + 4 * gq) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
}

// conv[k][16 mt + n] for the MTB 16-frame tiles of the window, tiles dealt to the NW waves three at a time.  The filter is the
// row operand, so lane (n = lane & 15, q = lane >> 4) ends up with kernels 4 q .. 4 q + 3 of frame 16 mt + n: epi(mt, n, q, acc)
template <int NW, class Epi>
__device__ inline void conv_mfma(const unsigned short* ximg, const unsigned short* wimg, const ConvGeo& g, int MTB, int wave, int lane, Epi epi) {
    constexpr int CT = 3;
    const int m = lane & 15, q = lane >> 4, cc = m & 3;
    const unsigned short* xh = ximg + (long)cc * g.IMG_LD + (m - cc) + 8 * q;
    const unsigned short* xl = xh + 4 * g.IMG_LD;
    const unsigned short* wh = wimg + m * g.WKP + 8 * q;
    const unsigned short* wl = wh + 16 * g.WKP;
    for (int mt0 = wave; mt0 < MTB; mt0 += NW * CT) {
        f32x4 acc[CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int ks = 0; ks < g.NKS; ++ks) {
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(wh + 32 * ks), bl = *reinterpret_cast<const bf16x8*>(wl + 32 * ks);
#pragma unroll
            for (int c = 0; c < CT; ++c) {
                const int mt = mt0 + NW * c;
                if (mt < MTB) {
                    const unsigned short* ph = xh + 16 * mt + 32 * ks;
                    const unsigned short* pl = xl + 16 * mt + 32 * ks;
                    const uint2 h0 = *reinterpret_cast<const uint2*>(ph), h1 = *reinterpret_cast<const uint2*>(ph + 4);
                    const uint2 l0 = *reinterpret_cast<const uint2*>(pl), l1 = *reinterpret_cast<const uint2*>(pl + 4);
                    const bf16x8 ah = __builtin_bit_cast(bf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
                    const bf16x8 al = __builtin_bit_cast(bf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
                    acc[c] = mma16(bh, ah, acc[c]);
                    acc[c] = mma16(bl, ah, acc[c]);
                    acc[c] = mma16(bh, al, acc[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
            const int mt = mt0 + NW * c;
            if (mt < MTB) epi(mt, m, q, acc[c]);
        }
    }
}

// barrier among the NCW compute waves only (the polling waves are inside a spin loop at these points)
__device__ __forceinline__ void compute_barrier(unsigned* cnt, unsigned& gen) {
    gen += NCW;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < gen) __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__global__ void build_wcat16_kernel(const float* __restrict__ wih, const float* __restrict__ whh, unsigned short* __restrict__ out,
                                    int rows, int Dd, int E, int KCP) {
    const long total = (long)rows * KCP;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / KCP), k = (int)(i - (long)r * KCP);
        float v = 0.f;
        if (k < E) v = wih[(long)r * (Dd + E) + Dd + k];
        else if (k < E + Dd) v = whh[(long)r * Dd + (k - E)];
        out[i] = f2bf_bits(v);
    }
}

// Every workgroup of a cluster waits for its peers, so the whole grid has to be resident at once: checked, not assumed
// (occupancy of THIS kernel at its block size and LDS x compute units; the API can over-report, so at most one workgroup per
// CU is counted - each one needs most of a CU's LDS anyway).  Not resident -> the caller falls back to the per-step kernels.
template <typename K>
bool grid_resident(K kernel, int grid, int block, size_t lds) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block, lds) != hipSuccess || per_cu < 1) return false;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
    return grid <= cus;
}

// flat copy of one gathered exchange region into LDS (poll role): n16 16-byte pairs
template <int CH>
__device__ __forceinline__ void poll_copy(const u64* src, int n16, float* dst, int gt, int np, u64 want, unsigned* status) {
    for (int i0 = gt; i0 < n16; i0 += CH * np) {
        u64 lo[CH], hi[CH];
        int cnt = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) if (i0 + k * np < n16) cnt = k + 1;
        gather16<CH>(src + 2 * i0, 2 * np, cnt, PAIR_MASK, want, lo, hi, status);
#pragma unroll
        for (int k = 0; k < CH; ++k)
            if (k < cnt) *reinterpret_cast<float4*>(dst + 4 * (long)(i0 + k * np)) = make_float4(lo_f(lo[k]), hi_f(lo[k]), lo_f(hi[k]), hi_f(hi[k]));
    }
}

// ---- launch epoch (4 tag bits), kept IN the work area -------------------------------------------------------------------
// Consecutive launches on the same work area carry consecutive epochs, so the tags a slot can still hold - those of the
// previous launch on that area - never match, and a tag recurs only after 14 launches that each rewrote every slot.  The
// counter is a word of the area itself (the last 64 bytes of its 4 KB status block): the kernels read it at their start, a
// one-thread kernel behind each launch advances it.  No host-side state: the library keeps nothing per process, device or
// work area (round 2 kept an unsynchronised std::unordered_map keyed by pointer).  A fresh area (zeros) starts at epoch 0.
constexpr int EPOCH_WORD = 1008;                 // unsigned index into the status block: bytes 4032..4035
constexpr size_t STATUS_CLEAR_BYTES = 4032;     // what a launcher clears of the status block
__global__ void bump_epoch_kernel(unsigned* status) { status[EPOCH_WORD] = status[EPOCH_WORD] + 1u; }
// clears [0, STATUS_CLEAR_BYTES) and [4096, 4096 + xbuf_bytes) of a work area in one launch (the epoch word survives)
__global__ void clear_work_kernel(uint4* work, long n16) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x)
        if (i < (long)(STATUS_CLEAR_BYTES / 16) || i >= 256) work[i] = make_uint4(0u, 0u, 0u, 0u);
}
inline void clear_work(void* work, size_t xbuf_bytes, hipStream_t st) {
    const long n16 = (long)((4096 + xbuf_bytes) / 16);
    hipLaunchKernelGGL(clear_work_kernel, dim3((unsigned)std::min<long>((n16 + 255) / 256, 2048)), dim3(256), 0, st, (uint4*)work, n16);
}

}  // namespace
