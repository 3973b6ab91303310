// bf16 x bf16 -> bf16 contraction of the encoder stack (bf16 contraction mode), "NT" form: both operands K-contiguous,
//   C[m,n] = act( sum_k X[m,k] W[n,k] + bias[n] )       X (M,K) activations, W (N,K) weights (or a transposed weight copy)
// = nn.Linear forward (reference src/module.py:1078-1079 `pj`, the input half of nn.LSTM src/module.py:1023) and the
// input-gradient contractions of both (with the transposed bf16 weight copies of asr_rnn_pack_weights).
//
// 128 x 128 x 64 tiles, 256 threads (4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles of v_mfma_f32_16x16x32_bf16).
//  * operands go global -> LDS directly (buffer_load ... lds, 16 bytes per lane, no VGPR staging, no conversion): rows
//    outside the matrix and chunks beyond K are addressed past the buffer's num_records and come back as zeros;
//  * LDS image: [row][8 chunks of 16 B], chunk index XOR-ed with (row & 7): the direct-to-LDS destination is lane-linear,
//    so the swizzle is applied to the SOURCE address (which chunk a lane fetches) and again on the fragment read
//    (ds_read_b128 then runs conflict-free);
//  * the MFMA takes the WEIGHT fragment as its A operand, so a lane's four accumulator registers are four consecutive
//    output columns of one row: the epilogue (bias, tanh / ReLU, bf16 rounding) writes 8 bytes per lane per tile;
//  * 32 KB of LDS and <= 168 VGPRs: three workgroups per CU overlap each other's load / MFMA / store phases (K is short in
//    this model - 160 .. 2560 - so a deep per-workgroup pipeline would spend its time in prologue and epilogue);
//  * tile order: each XCD gets a contiguous run of tiles, N fastest, so the X rows of a run stay in that XCD's L2 and the
//    weight matrix (<= 3.3 MB) is shared by all of its workgroups.
#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64, NTH = 256;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;     // one k-step of both operands in LDS (32 KB)
constexpr unsigned OOB = 0x80000000u;        // byte offset beyond any operand (num_records < 2^31): the load returns zeros

struct G16P {
    const unsigned short* X; const unsigned short* W; unsigned short* C; const float* bias;
    int M, N, K;
    long ldx, ldw, ldc;
    int act;
    unsigned xbytes, wbytes;
    int ntx, nty;            // tiles along N, along M
    // CONV instantiation only: rows are the pixels of a zero-bordered channel-last image (B, T2, F2, C) and k-step kt of the
    // 9*C-deep reduction reads the rows shifted by its tap (convC > 0), border rows are stored as zeros (maskF2 > 0)
    int convC, maskF2, maskT2;
    float* C32;              // CONV only: when set, the result goes here as fp32 (pre-activations that a LayerNorm normalises next)
};

typedef __attribute__((address_space(3))) void lds_void;
// tanh through one exp and one reciprocal (relative error ~1e-6, far below the bf16 rounding of the stored result)
__device__ __forceinline__ float tanh_fast(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// Two tilings of the same kernel (NWM x NWN waves, each MT x 4 MFMA tiles = 16 MT rows x 64 columns):
//   <2,2,4>  128 x 128, 256 threads, 2 x 32 KB LDS, two workgroups per CU - short problems, narrow outputs;
//   <2,4,8>  256 x 256, 512 threads, 2 x 64 KB LDS, one workgroup per CU - a wave's 128 x 64 block reads 12 fragments for
//            32 MFMAs (8 for 16 in the small tiling, where the LDS read port is as busy as the MFMA pipe: 64 KB of fragment
//            reads = 512 clk per 512 clk of MFMA) and the L2 -> LDS bytes per flop are halved.
template <int NWM, int NWN, int MT, bool CONV = false>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN > 4) ? 1 : 2) void gemm16_nt_kernel(G16P p) {
    constexpr int BM = NWM * MT * 16, BN = NWN * 64, NW = NWM * NWN;
    constexpr int XP = BM / 8 / NW, WP = BN / 8 / NW;          // 8-row staging pieces per wave and operand
    constexpr int STAGE_BYTES = (BM + BN) * BK * 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // X tile 16 KB | W tile 16 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile of this workgroup: XCD x (= blockIdx % 8 under round-robin dispatch; speed only) walks a contiguous run
    const int ntiles = p.ntx * p.nty;
    int id = blockIdx.x;
    {
        const int qq = ntiles >> 3, rr = ntiles & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int bx = id % p.ntx, by = id / p.ntx;
    const int m0 = by * BM, n0 = bx * BN;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.X), 0, p.xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.W), 0, p.wbytes, 0x00020000);

    // staging map: wave w, piece i (0..3) fills rows (4w+i)*8 .. +7 of a tile; lane -> row +lane/8, LDS slot lane%8, which
    // holds the row's chunk (lane%8) ^ (lane/8)   [row & 7 == lane / 8]
    const int srow = lane >> 3, schunk = (lane & 7) ^ srow;
    unsigned xoff[XP], woff[WP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
        const int r = (XP * w + i) * 8 + srow;
        xoff[i] = (m0 + r) < p.M ? (unsigned)(((long)(m0 + r) * p.ldx + schunk * 8) * 2) : OOB;
    }
#pragma unroll
    for (int i = 0; i < WP; ++i) {
        const int r = (WP * w + i) * 8 + srow;
        woff[i] = (n0 + r) < p.N ? (unsigned)(((long)(n0 + r) * p.ldw + schunk * 8) * 2) : OOB;
    }
    const int kchunk = schunk * 8;             // first k of this lane's chunk within a k-step

    const int wr = w / NWN, wc = w % NWN;
    const int fr = lane & 15, fq = lane >> 4;
    f32x4 acc[4][MT];                           // [n tile][m tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes) within a tile for k-half ks: row*128 + ((4*ks + fq) ^ (row & 7)) * 16; row & 7 == fr & 7
    const int xrow = wr * (16 * MT) + fr, wrow = wc * 64 + fr;

    // Two LDS stages: the loads of k-step kt+1 are in flight while k-step kt is multiplied.  The direct-to-LDS loads are
    // retired by a COUNTED wait (8 loads per stage and wave) and the barriers are raw `s_barrier`s: `__syncthreads()` would
    // drain the loads in flight (its fence waits vmcnt(0) while an LDS-DMA is pending).
    const int nk = (p.K + BK - 1) / BK;
    auto stage = [&](int kt, int buf) {
        const int k0 = kt * BK;
        const bool kok = (k0 + kchunk) < p.K;
        unsigned char* base = smem + buf * STAGE_BYTES;
        // implicit 3x3 convolution: this k-step lies inside ONE tap (C % 64 == 0); its rows are the pixel rows shifted by
        // (dt, df) in the zero-bordered image, i.e. by a constant number of rows - a negative total offset wraps to a huge
        // unsigned value and the load returns zeros like any other access past num_records
        unsigned kadd = (unsigned)(k0 * 2);
        if (CONV && p.convC > 0) {
            const int tap = k0 / p.convC, ci0 = k0 - tap * p.convC;
            const int rowoff = (tap / 3 - 1) * p.maskF2 + (tap % 3 - 1);
            kadd = (unsigned)(((long)rowoff * p.convC + ci0) * 2);
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const unsigned vx = (xoff[i] != OOB && kok) ? xoff[i] + kadd : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_void*)(base + (XP * w + i) * 1024), 16, vx, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WP; ++i) {
            const unsigned vw = (woff[i] != OOB && kok) ? woff[i] + (unsigned)(k0 * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void*)(base + BM * BK * 2 + (WP * w + i) * 1024), 16, vw, 0, 0, 0);
        }
    };
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) {
            stage(kt + 1, buf ^ 1);
            static_assert(XP + WP == 8, "the counted wait below assumes 8 loads per stage and wave");
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const unsigned char* Xs = smem + buf * STAGE_BYTES;
        const unsigned char* Ws = Xs + BM * BK * 2;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 xf[MT], wf[4];
            const int sl = ((4 * ks + fq) ^ (fr & 7)) * 16;
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const bf16x8*>(Xs + (xrow + 16 * t) * 128 + sl);
#pragma unroll
            for (int t = 0; t < 4; ++t) wf[t] = *reinterpret_cast<const bf16x8*>(Ws + (wrow + 16 * t) * 128 + sl);
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] = mma16(wf[a], xf[b], acc[a][b]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();            // everyone has read stage `buf` before the next iteration refills it
    }

    // epilogue: acc[a][b][r] = C[m = m0 + wr*64 + 16b + fr][n = n0 + wc*64 + 16a + 4*fq + r]: four consecutive columns per
    // lane and tile = 8 bytes of bf16.  Two neighbouring column tiles (a, a+1) are exchanged between the lane rows
    // (v_permlane16_swap: odd rows of the first operand <-> even rows of the second) so that every lane ends up with 8
    // consecutive columns of ONE tile = one 16-byte store (half the store instructions of a store-issue-bound tail):
    //   lane row fq: tile a + (fq & 1), columns 8*(fq >> 1) .. +7
#pragma unroll
    for (int a = 0; a < 4; a += 2) {
        float4 bva = make_float4(0.f, 0.f, 0.f, 0.f), bvb = bva;
        const int na = n0 + wc * 64 + 16 * a + 4 * fq, nb_ = na + 16;
        if (p.bias) {
            if (na < p.N) bva = *reinterpret_cast<const float4*>(p.bias + na);
            if (nb_ < p.N) bvb = *reinterpret_cast<const float4*>(p.bias + nb_);
        }
        const int nst = n0 + wc * 64 + 16 * (a + (fq & 1)) + 8 * (fq >> 1);        // first of this lane's 8 output columns
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = m0 + wr * (16 * MT) + 16 * b + fr;
            bool border = false;
            if (CONV && p.maskF2 > 0) {
                const int rowi = m / p.maskF2, fp = m - rowi * p.maskF2, tp = rowi % p.maskT2;
                border = fp == 0 || fp == p.maskF2 - 1 || tp == 0 || tp == p.maskT2 - 1;
            }
            if (CONV && p.C32) {
                // fp32 result (no activation): four consecutive columns per lane and tile
                if (m < p.M) {
                    if (na < p.N) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc + na) = border ? make_float4(0.f, 0.f, 0.f, 0.f) :
                        make_float4(acc[a][b][0] + bva.x, acc[a][b][1] + bva.y, acc[a][b][2] + bva.z, acc[a][b][3] + bva.w);
                    if (nb_ < p.N) *reinterpret_cast<float4*>(p.C32 + (long)m * p.ldc + nb_) = border ? make_float4(0.f, 0.f, 0.f, 0.f) :
                        make_float4(acc[a + 1][b][0] + bvb.x, acc[a + 1][b][1] + bvb.y, acc[a + 1][b][2] + bvb.z, acc[a + 1][b][3] + bvb.w);
                }
                continue;
            }
            unsigned oa[2], ob[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float a0 = acc[a][b][2 * h] + (h ? bva.z : bva.x), a1 = acc[a][b][2 * h + 1] + (h ? bva.w : bva.y);
                float b0 = acc[a + 1][b][2 * h] + (h ? bvb.z : bvb.x), b1 = acc[a + 1][b][2 * h + 1] + (h ? bvb.w : bvb.y);
                if (p.act == ASR_ACT_TANH) { a0 = tanh_fast(a0); a1 = tanh_fast(a1); b0 = tanh_fast(b0); b1 = tanh_fast(b1); }
                else if (p.act == ASR_ACT_RELU) { a0 = fmaxf(a0, 0.f); a1 = fmaxf(a1, 0.f); b0 = fmaxf(b0, 0.f); b1 = fmaxf(b1, 0.f); }
                oa[h] = (unsigned)f2bf_bits(a0) | ((unsigned)f2bf_bits(a1) << 16);
                ob[h] = (unsigned)f2bf_bits(b0) | ((unsigned)f2bf_bits(b1) << 16);
            }
            const auto s0 = __builtin_amdgcn_permlane16_swap(oa[0], ob[0], false, false);
            const auto s1 = __builtin_amdgcn_permlane16_swap(oa[1], ob[1], false, false);
            // rows 0,2 keep tile a's own words and receive its neighbour row's; rows 1,3 hold tile a+1 likewise
            uint4 o = make_uint4(s0[0], s1[0], s0[1], s1[1]);
            if (CONV && border) o = make_uint4(0u, 0u, 0u, 0u);
            if (m < p.M && nst < p.N) *reinterpret_cast<uint4*>(p.C + (long)m * p.ldc + nst) = o;
        }
    }
}

}  // namespace

// Returns ASR_OK when launched, 1 when the shape does not qualify (the caller uses the generic kernel).
int gemm16_nt(const void* X, const void* W, void* C, const float* bias, int M, int N, int K, long ldx, long ldw, long ldc,
              int act, hipStream_t st) {
    if (K % 8 != 0 || N % 8 != 0 || ldx % 8 != 0 || ldw % 8 != 0 || ldc % 8 != 0) return 1;
    if ((((uintptr_t)X | (uintptr_t)W | (uintptr_t)C) & 15) != 0 || (bias && ((uintptr_t)bias & 15) != 0)) return 1;
    const long xb = ((long)(M - 1) * ldx + K) * 2, wb = ((long)(N - 1) * ldw + K) * 2;
    if (xb >= (1L << 31) || wb >= (1L << 31)) return 1;
    if (xb + 2L * K >= (long)OOB || wb + 2L * K >= (long)OOB) return 1;   // a valid byte offset never equals the OOB marker
    // The 256 x 256 tiling is selected by ASR_GEMM16_BIG=1 only: measured on MI355X it is no faster on any shape of the
    // encoder (78.7 vs 80.1 us on 19200 x 2560 x 640, slower wherever the tile count drops under two rounds) - the main
    // loop, not the fragment traffic, holds both tilings near 0.3 of the MFMA peak.
    static const bool big = [] { const char* e = getenv("ASR_GEMM16_BIG"); return e && atoi(e) > 0; }();
    const int bm = big ? 256 : BM, bn = big ? 256 : BN;
    G16P p{(const unsigned short*)X, (const unsigned short*)W, (unsigned short*)C, bias, M, N, K, ldx, ldw, ldc, act,
           (unsigned)xb, (unsigned)wb, cdiv(N, bn), cdiv(M, bm), 0, 0, 0, nullptr};
    const long ntiles = (long)p.ntx * p.nty;
    if (ntiles >= (1L << 31)) return 1;
    if (big) {
        static const bool once = [] {
            hipFuncSetAttribute((const void*)gemm16_nt_kernel<2, 4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            return true;
        }();
        (void)once;
        hipLaunchKernelGGL((gemm16_nt_kernel<2, 4, 8>), dim3((unsigned)ntiles), dim3(512), 128 * 1024, st, p);
    } else {
        hipLaunchKernelGGL((gemm16_nt_kernel<2, 2, 4>), dim3((unsigned)ntiles), dim3(NTH), 2 * STAGE_BYTES, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_gemm16(nt): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

// 3x3 convolution (stride 1, zero padding 1) as an implicit GEMM on the NT kernel, nn.Conv2d of the VGG front-ends (reference
// src/module.py:599-614,670-681).  img: zero-bordered channel-last bf16 image (B, T+2, F+2, C); W: (N, 9*C) bf16 with
// k = tap*C + ci; out: (B, T+2, F+2, N) bf16, borders written as zeros (ready to be the next layer's input).  C % 64 == 0.
// out_f32: the result is stored as fp32 (B, T+2, F+2, N) without activation (the input of a CNNLayerNorm).
// convC == 0: `img` is an explicit (rows, K) patch matrix over the same zero-bordered pixel grid (first layer: C = 4), only the
// border mask applies.  Returns 1 when the shape does not qualify.
int gemm16_conv3x3(const void* img, const void* W, void* out, const float* bias, int B, int T, int F, int C, int N, int K, int implicit, int act,
                   int out_f32, hipStream_t st) {
    const int T2 = T + 2, F2 = F + 2;
    const long M = (long)B * T2 * F2;
    const long ldx = implicit ? C : K;
    if (implicit && (C % 64 != 0 || K != 9 * C)) return 1;
    if (K % 8 != 0 || N % 8 != 0 || M >= (1L << 31)) return 1;
    if ((((uintptr_t)img | (uintptr_t)W | (uintptr_t)out) & 15) != 0 || (bias && ((uintptr_t)bias & 15) != 0)) return 1;
    const long xb = M * ldx * 2, wb = (long)N * K * 2;
    if (xb + 2L * K >= (long)OOB || wb + 2L * K >= (long)OOB) return 1;
    G16P p{(const unsigned short*)img, (const unsigned short*)W, (unsigned short*)out, bias, (int)M, N, K, ldx, (long)K, (long)N, act,
           (unsigned)xb, (unsigned)wb, cdiv(N, BN), cdiv(M, BM), implicit ? C : 0, F2, T2, out_f32 ? (float*)out : nullptr};
    if (out_f32 && act != ASR_ACT_NONE) return 1;
    const long ntiles = (long)p.ntx * p.nty;
    if (ntiles >= (1L << 31)) return 1;
    hipLaunchKernelGGL((gemm16_nt_kernel<2, 2, 4, true>), dim3((unsigned)ntiles), dim3(NTH), 2 * STAGE_BYTES, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_conv3x3_16: launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}

// =================================================================================================================
// "TN" form: weight gradients.  C[i,j] += sum_r A[r,i] * B[r,j]   (A = gate / pre-activation gradients (R,I), B = layer
// input / h (R,J), both bf16 with the REDUCTION index as the row; C fp32, accumulated with atomics over `splits`
// slices of R).  nn.Linear / nn.LSTM weight-gradient autograd (reference src/module.py:1023,1078).
//
// Same 128 x 128 x 64 tile and direct-to-LDS staging as the NT kernel; the LDS image keeps the memory order [r][128
// columns] (one wave instruction = 4 rows x 256 B) and the MFMA fragments - 8 consecutive r for one column - come out of
// it through the hardware-transposed read ds_read_b64_tr_b16 (4 r-rows x 16 columns per 16-lane group).  Rows are 256 B =
// all 64 banks, so without a swizzle the 8 r-rows a 32-lane half touches collide 8-way: the 16-byte chunk index is
// XOR-ed with 2 * f(r), f(r) = (r & 3) | ((r >> 3) & 1) << 2, distinct for those 8 rows (applied to the SOURCE chunk of the
// direct-to-LDS load and again on the read).
//   bshift / seqT / padded: row r = (b,t) of B is read at padded row b*(seqT+2) + t + 1 + bshift (h_{t-1} / h_{t+1} of the
//   time-padded y16) - per-lane source addresses make that free;   permH: gate-minor -> reference row order of C.
struct T16P {
    const unsigned short* A; const unsigned short* B; float* C;
    int I, J, R;                 // C is (I,J); reduction over R rows
    long lda, ldb, ldc;
    unsigned abytes, bbytes;
    int nti, ntj, splits, per;   // tiles along I and J, reduction slices, k-steps per slice
    int seqT, bshift, permH;
    float inv_seqT;
    int tapF2;                   // > 0: NINE contractions in one launch, the taps of a 3x3 convolution's weight gradient: tap
                                 // (dt, df) reads B at row r + dt*tapF2 + df (rows outside [0, R) are zeros) and writes C at column tap*J
};

__device__ __forceinline__ int swz_f(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int STAGES>
__global__ __launch_bounds__(NTH, STAGES == 1 ? 3 : (STAGES == 2 ? 2 : 1)) void gemm16_tn_kernel(T16P p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // A tile [64][256 B] | B tile [64][256 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = p.nti * p.ntj;
    const int ntap = p.tapF2 > 0 ? 9 : 1;
    const int total = ntiles * p.splits * ntap;
    int id = blockIdx.x;
    {
        const int qq = total >> 3, rr = total & 7, xcd = id & 7, loc = id >> 3;
        id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + loc;
    }
    const int zt = id / ntiles, tile = id - zt * ntiles;
    const int tap = zt % ntap, z = zt / ntap;
    const int rowshift = p.tapF2 > 0 ? (tap / 3 - 1) * p.tapF2 + (tap % 3 - 1) : 0;
    const int bj = tile % p.ntj, bi = tile / p.ntj;
    const int i0 = bi * BM, j0 = bj * BN;
    const int nk = (p.R + BK - 1) / BK;
    const int kt0 = z * p.per, kt1 = min(nk, kt0 + p.per);
    if (kt0 >= kt1) return;

    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.A), 0, p.abytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(p.B), 0, p.bbytes, 0x00020000);

    // staging: wave w, piece i (0..3) fills r-rows (4w+i)*4 .. +3 of a tile; lane -> row + lane/16, LDS slot lane%16
    const int srow = lane >> 4, sslot = lane & 15;
    int colA[4], colB[4];                     // first column (element) of the chunk this lane fetches, per piece
    bool caok[4], cbok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (4 * w + i) * 4 + srow;
        const int chunk = sslot ^ (2 * swz_f(r));
        colA[i] = i0 + chunk * 8; colB[i] = j0 + chunk * 8;
        caok[i] = colA[i] < p.I; cbok[i] = colB[i] < p.J;          // I, J % 8 == 0: a chunk is all in or all out
    }

    const int wr = w >> 1, wc = w & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int tq = fr >> 2, tp = fr & 3;      // transposed read: lane 4q+p of a 16-lane group supplies row q, columns 4p..4p+3
    f32x4 acc[4][4];                          // [i tile][j tile]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef __attribute__((address_space(3))) bf16x4 lds_bf16x4;

    // ONE LDS stage at four workgroups per CU: measured against two stages at two per CU (as in gemm16_nt_kernel) the weight-
    // gradient shapes lose 20 % with the deeper pipeline - many short split-K workgroups hide each other's loads better
    auto stage = [&](int kt, int buf) {
        const int r0 = kt * BK;
        unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + (4 * w + i) * 4 + srow;
            const bool rok = r < p.R;
            long rsrc = r;
            if (p.seqT > 0) {
                int q = (int)((float)r * p.inv_seqT);
                if (q * p.seqT > r) --q; else if ((q + 1) * p.seqT <= r) ++q;
                rsrc = (long)r + 2 * q + 1 + p.bshift;
            }
            if (p.tapF2 > 0) rsrc = (long)r + rowshift;
            const unsigned va = (rok && caok[i]) ? (unsigned)(((long)r * p.lda + colA[i]) * 2) : OOB;
            const unsigned vb = (rok && cbok[i] && rsrc >= 0 && (p.tapF2 == 0 || rsrc < p.R)) ? (unsigned)((rsrc * p.ldb + colB[i]) * 2) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void*)(base + (4 * w + i) * 1024), 16, va, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void*)(base + BK * 256 + (4 * w + i) * 1024), 16, vb, 0, 0, 0);
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* As = smem + buf * STAGE_BYTES;
        const unsigned char* Bs = As + BK * 256;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            // fragment of column tile c0 (16 columns) for this k-half: lo = r-rows 32ks + 8fq + tq, hi = +4
            bf16x8 af[4], bf[4];
            const int rlo = 32 * ks + 8 * fq + tq, rhi = rlo + 4;
            const int slo = 2 * swz_f(rlo), shi = 2 * swz_f(rhi);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ca = wr * 64 + 16 * t + 4 * tp, cb = wc * 64 + 16 * t + 4 * tp;      // element column within the tile
                const bf16x4 alo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(As + rlo * 256 + (((ca >> 3) ^ slo) * 16) + (ca & 7) * 2));
                const bf16x4 ahi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(As + rhi * 256 + (((ca >> 3) ^ shi) * 16) + (ca & 7) * 2));
                const bf16x4 blo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Bs + rlo * 256 + (((cb >> 3) ^ slo) * 16) + (cb & 7) * 2));
                const bf16x4 bhi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4*)(Bs + rhi * 256 + (((cb >> 3) ^ shi) * 16) + (cb & 7) * 2));
                af[t] = (bf16x8){alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
                bf[t] = (bf16x8){blo[0], blo[1], blo[2], blo[3], bhi[0], bhi[1], bhi[2], bhi[3]};
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = mma16(af[a], bf[b], acc[a][b]);
        }
    };
    if constexpr (STAGES == 1) {
        for (int kt = kt0; kt < kt1; ++kt) {
            stage(kt, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    } else {
        // STAGES LDS stages, STAGES - 1 k-steps of direct-to-LDS loads in flight (one workgroup per CU at 4 stages): a
        // single workgroup's k-step is otherwise one exposed L2 round trip (~1 us) per 0.2 us of MFMA work - measured
        // 1.4 us per k-step at one workgroup per CU - and many short split-K slices pay for that hiding with fp32 atomics
        // (900 workgroups x 64 KB on the 2560 x 640 gradient).  Counted waits: 8 loads per stage and wave.
        constexpr int AHEAD = STAGES - 1;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) if (kt0 + i < kt1) stage(kt0 + i, i);
        for (int kt = kt0; kt < kt1; ++kt) {
            const int rel = kt - kt0;
            if (kt + AHEAD < kt1) stage(kt + AHEAD, (rel + AHEAD) % STAGES);
            const int inflight = min(AHEAD, kt1 - 1 - kt);          // k-steps requested behind this one
            if (inflight >= 3) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else if (inflight == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (inflight == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            compute(rel % STAGES);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
    }

    // epilogue: acc[a][b][r] = C[i = i0 + wr*64 + 16a + 4*fq + r][j = j0 + wc*64 + 16b + fr]
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int i = i0 + wr * 64 + 16 * a + 4 * fq + r;
            if (i >= p.I) continue;
            if (p.permH > 0) { const int h4 = 4 * p.permH, blk = i / h4, rr = i - blk * h4; i = blk * h4 + (rr & 3) * p.permH + (rr >> 2); }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int j = j0 + wc * 64 + 16 * b + fr;
                if (j < p.J) atomicAdd(p.C + (long)i * p.ldc + (long)tap * p.J + j, acc[a][b][r]);
            }
        }
    }
}

// Returns ASR_OK when launched, 1 when the shape does not qualify.
int gemm16_tn_taps(const void* A, const void* B, float* C, int I, int J, int R, long lda, long ldb, long ldc, int splits, int perm_h,
                   int seqT, int bshift, int padded, int tapF2, hipStream_t st);
int gemm16_tn(const void* A, const void* B, float* C, int I, int J, int R, long lda, long ldb, long ldc, int splits, int perm_h,
              int seqT, int bshift, int padded, hipStream_t st) {
    return gemm16_tn_taps(A, B, C, I, J, R, lda, ldb, ldc, splits, perm_h, seqT, bshift, padded, 0, st);
}
int gemm16_tn_taps(const void* A, const void* B, float* C, int I, int J, int R, long lda, long ldb, long ldc, int splits, int perm_h,
                   int seqT, int bshift, int padded, int tapF2, hipStream_t st) {
    if (I % 8 != 0 || J % 8 != 0 || lda % 8 != 0 || ldb % 8 != 0) return 1;
    if ((((uintptr_t)A | (uintptr_t)B) & 15) != 0) return 1;
    if (seqT > 0 && !padded) return 1;                        // unpadded shifted rows need masking: generic kernel
    const long rows_b = seqT > 0 ? (long)(R / seqT) * (seqT + 2) : R;
    const long ab = ((long)(R - 1) * lda + I) * 2, bb = ((rows_b - 1) * ldb + J) * 2;
    if (ab >= (1L << 31) || bb >= (1L << 31) || (seqT > 0 && R % seqT != 0)) return 1;
    const int nk = cdiv(R, BK);
    if (splits < 1) splits = 1;
    if (splits > nk) splits = nk;
    T16P p{(const unsigned short*)A, (const unsigned short*)B, C, I, J, R, lda, ldb, ldc, (unsigned)ab, (unsigned)bb,
           cdiv(I, BM), cdiv(J, BN), splits, cdiv(nk, splits), seqT, bshift, perm_h, seqT > 0 ? 1.0f / (float)seqT : 0.f, tapF2};
    const long total = (long)p.nti * p.ntj * p.splits * (tapF2 > 0 ? 9 : 1);
    if (total >= (1L << 31)) return 1;
    // ASR_GEMM16_TN_STAGES=2: the double-buffered instantiation (two workgroups per CU); default one stage, three per CU
    static const int stages = [] { const char* e = getenv("ASR_GEMM16_TN_STAGES"); return e ? atoi(e) : 1; }();
    if (stages == 4) {
        static unsigned char attr4_[32];
        if (first_on_device(attr4_)) hipFuncSetAttribute((const void*)gemm16_tn_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * STAGE_BYTES);
        hipLaunchKernelGGL(gemm16_tn_kernel<4>, dim3((unsigned)total), dim3(NTH), 4 * STAGE_BYTES, st, p);
    } else if (stages == 2) {
        static const bool once = [] { hipFuncSetAttribute((const void*)gemm16_tn_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_BYTES); return true; }();
        (void)once;
        hipLaunchKernelGGL(gemm16_tn_kernel<2>, dim3((unsigned)total), dim3(NTH), 2 * STAGE_BYTES, st, p);
    } else {
        hipLaunchKernelGGL(gemm16_tn_kernel<1>, dim3((unsigned)total), dim3(NTH), STAGE_BYTES, st, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { asr_set_error("asr_gemm16(tn): launch failed: %s", hipGetErrorString(e)); return ASR_E_LAUNCH; }
    return ASR_OK;
}
