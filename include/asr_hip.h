/*
 * libasr_hip.so — C ABI of the MI355X (gfx950) joint CTC-attention ASR training hot path.
 *
 * The reference (DanielLin94144/E2E-ASR-Pytorch) has no FFI: every entry point below replaces a
 * PyTorch call site on the path  src/asr.py:89-177 (ASR.forward) -> bin/train_asr.py:229-253 (losses,
 * backward); the citation next to each function names the reference lines it stands in for.
 *
 * Conventions
 *   - plain C types only; every tensor pointer is a DEVICE pointer to dense row-major fp32 (or the
 *     stated integer type) owned by the caller and valid until `stream` reaches the call;
 *   - no entry point allocates device memory or synchronises the stream: scratch space is passed in
 *     by the caller (sizes are documented per call);
 *   - returns ASR_OK (0) or a negative ASR_E_* code; asr_last_error() gives a thread-local message;
 *   - `prec` selects the matrix-core type of the contractions: ASR_BF16 (bf16 MFMA operands, fp32
 *     accumulate, fp32 everywhere else) or ASR_F32 (exact fp32-input MFMA, parity mode);
 *   - `stream` is a hipStream_t passed as void*.
 */
#ifndef ASR_HIP_H
#define ASR_HIP_H
#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASR_OK 0
#define ASR_E_ARG (-1)
#define ASR_E_LAUNCH (-2)
#define ASR_E_UNSUPPORTED (-3)

#define ASR_F32 0
#define ASR_BF16 1

#define ASR_ACT_NONE 0
#define ASR_ACT_TANH 1
#define ASR_ACT_RELU 2

typedef void* asr_stream_t;

const char* asr_last_error(void);
int asr_version(void);
/* "gfx950" — the only architecture this library carries code objects for. */
const char* asr_device_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Contractions (replace nn.Linear / torch.mm / bmm and their autograd: src/module.py:1078-1079 `pj`,
 * src/asr.py:29-32 `ctc_layer`, src/asr.py:292-293,345 `proj_q/proj_k`, src/asr.py:213 `char_trans`,
 * the input half of nn.LSTM src/module.py:1023, and torch.bmm src/module.py:1114).
 *   C[i,j] (+)= act( sum_r opA(i,r) opB(r,j) + bias[j] ),  i<M, j<N, r<K
 *   a_kc=1: A element (i,r) at A[i*lda+r];   a_kc=0: at A[r*lda+i]
 *   b_kc=1: B element (r,j) at B[j*ldb+r];   b_kc=0: at B[r*ldb+j]
 *   accum=1: add into C;  splits>1: the reduction is cut into `splits` slices added with fp32 atomics
 *            (needs accum=1, act=NONE);  batch>1 with element strides sA/sB/sC.
 *   seqT>0 (b_kc=0 only): reduction index r enumerates (b,t) with t = r % seqT; B is read at row
 *            r+bshift and treated as zero when t+bshift falls outside [0,seqT) — the h_{t-1}/h_{t+1}
 *            operand of the recurrent weight gradient.
 */
int asr_gemm(const float* A, const float* B, float* C, const float* bias,
             int M, int N, int K, long lda, long ldb, long ldc,
             int a_kc, int b_kc, int act, int accum, int splits,
             int batch, long sA, long sB, long sC, int seqT, int bshift,
             int prec, asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
