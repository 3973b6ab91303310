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

/* ------------------------------------------------------------------------------------------------
 * Encoder BiLSTM time recurrence (the sequential half of nn.LSTM(bidirectional, batch_first),
 * src/module.py:1023,1049; zero initial state; padded frames are processed like the reference does).
 *   gates (B,T,ND,4H): in  = x W_ih^T + b_ih + b_hh for every frame (asr_gemm, gate order i,f,g,o)
 *                      out = activated gates (saved for backward)
 *   whh   (ND,4H,H)  : weight_hh_l0 [, weight_hh_l0_reverse]
 *   bias2 (ND,4H)    : optional second bias (bias_hh) added here when `gates` only carries bias_ih
 *   y     (B,T,ND*H) : h of both directions (forward half first);  c (B,T,ND,H): cell states
 * Backward consumes dy (gradient wrt y) and overwrites `gates` with the gradient wrt the gate
 * pre-activations, from which the caller forms dW_ih, dW_hh (shifted rows), db and dx with asr_gemm.
 * workspace: asr_lstm_workspace_bytes(B,H,ND), 256B aligned.  With a workspace both passes run as ONE persistent
 * launch each (weights resident in registers, h / partial-dh exchanged between workgroups through tagged
 * granules in the workspace; its first 32-bit word is an abort flag that stays 0 on success); shapes without a
 * persistent plan, a NULL workspace (forward) or ASR_LSTM_PERSIST=0 use one launch per time step.
 */
size_t asr_lstm_workspace_bytes(int B, int H, int ND);
/* 1 (default, or env ASR_LSTM_PERSIST) = persistent single-launch recurrence, 0 = one launch per step,
 * 2 = persistent but first-generation kernels only (A/B measurements); returns the old value. */
int asr_lstm_set_persistent(int on);
/* which implementation the two calls below take for this shape: 0 = launch per time step, 1 = first-generation
 * persistent kernel, 2 = second-generation persistent kernel (tests assert the plan the bench shape must get) */
int asr_lstm_plan(int B, int T, int H, int ND, int prec);
int asr_lstm_fwd(float* gates, const float* whh, const float* bias2, float* y, float* c,
                 int B, int T, int H, int ND, int prec,
                 void* workspace, size_t workspace_bytes, asr_stream_t stream);
int asr_lstm_bwd(float* gates, const float* whh, const float* dy, const float* c,
                 int B, int T, int H, int ND, int prec,
                 void* workspace, size_t workspace_bytes, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * bf16-STORAGE ENCODER STACK (bf16 contraction mode; the same reference call sites as above: nn.LSTM + Dropout +
 * down-sampling + tanh(Linear) of RNNLayer.forward, src/module.py:1040-1081, and their autograd).  Activations, gate
 * pre-activations / gradients and h live in HBM as bf16, the cell state and every parameter gradient as fp32:
 *   gates16 (B,T,ND,H,4) bf16, GATE-MINOR: [unit][i,f,g,o]; in = x W_ih^T + b_ih + b_hh, out = activated gates (forward);
 *           in = activated gates, out = gradient wrt the pre-activations (backward)
 *   y16     (B,T+2,ND*H) bf16: h of frame t at time row t+1; rows 0 and T+1 (h_{-1} / h_T) are ZEROED by asr_lstm16_fwd:
 *           the recurrent weight gradient reads h_{t-1} / h_{t+1} as shifted rows of this buffer
 *   dy16    (B,T,ND*H) bf16;  c (B,T,ND,H) fp32;  whh (ND,4H,H) fp32 in the REFERENCE row order [gate][unit]
 * asr_rnn_pack_weights builds, once per step, the bf16 contraction operands from the fp32 master weights: W_ih with rows
 * re-ordered gate-minor (so that the input projection writes gates16 directly) and its transpose, b_ih + b_hh in that
 * order, the projection weight and its transpose.  Weight gradients computed in gate-minor order are stored at the
 * reference's rows (perm_h of asr_gemm16 / asr_colsum16).
 * The recurrence runs as ONE persistent launch of 8 independent groups (direction x batch slice, one per XCD) of H/16
 * workgroups: B <= 16 * (8/ND), H % 16 == 0, H <= 512 (asr_lstm16_workspace_bytes returns 0 otherwise: use the fp32
 * entry points).  workspace: CALLER-OWNED and PERSISTENT per (layer, pass), zero-filled once at allocation, 256B aligned;
 * `epoch` = number of launches this workspace has seen (the caller increments it): it is folded into the granule tags
 * and selects the exchange region, so a (region, tag) pair repeats only every 32 launches of the same workspace.
 * `reserved_cus`: compute units another stream may occupy meanwhile (data-parallel all-reduce): the launch is refused
 * (ASR_E_UNSUPPORTED) unless all its workgroups fit beside them (occupancy query).  The workspace starts with TWO 1 KB status
 * blocks used by launch parity: a launch reports in block (epoch & 1) - its first 32-bit word is the abort word (see "Status
 * word" below) - and clears the other block for its successor, so consecutive launches on a workspace MUST carry consecutive
 * epochs and an uncollected abort word of launch k-1 must be folded before launch k+1 starts (it lives in the block k+1 clears).
 */
size_t asr_lstm16_workspace_bytes(int B, int H, int ND, int backward);
int asr_lstm16_fwd(void* gates16, const float* whh, void* y16, float* c, int B, int T, int H, int ND,
                   void* workspace, size_t workspace_bytes, unsigned epoch, int reserved_cus, asr_stream_t stream);
int asr_lstm16_bwd(void* gates16, const float* whh, const void* dy16, const float* c, int B, int T, int H, int ND,
                   void* workspace, size_t workspace_bytes, unsigned epoch, int reserved_cus, asr_stream_t stream);
/* asr_gemm on bf16 operands in HBM (same index conventions).  c_bf16 = 1: C bf16 = act(sum + bias), written;
 * c_bf16 = 0: C fp32, accumulated (accum / splits as asr_gemm), output row i stored at the reference row
 * (i / 4h)*4h + (i & 3)*h + (i % 4h >> 2) when perm_h = h > 0.  b_time_padded = 1 (with seqT, bshift): the reduction rows
 * of B live in a time-padded buffer (.,seqT+2,.), row (b,t) is read at padded row b*(seqT+2) + t + 1 + bshift.
 * Contiguous extents and row strides must be multiples of 8 elements, bases 16-byte aligned. */
int asr_gemm16(const void* A, const void* B, void* C, const float* bias, int M, int N, int K, long lda, long ldb, long ldc,
               int a_kc, int b_kc, int act, int accum, int splits, int c_bf16, int perm_h,
               int seqT, int bshift, int b_time_padded, asr_stream_t stream);
int asr_rnn_pack_weights(const float* w_ih, const float* b_ih, const float* b_hh, const float* pj,
                         void* w_ih16, void* w_ihT16, float* bias, void* pj16, void* pjT16,
                         int H, int ND, int Din, int D, asr_stream_t stream);
/* fp32 <-> bf16 (accum = 1: dst += src) */
int asr_cast_bf16(const float* src, void* dst, long n, asr_stream_t stream);
int asr_cast_f32(const void* src, float* dst, long n, int accum, asr_stream_t stream);
/* the bf16 forms of asr_dropout_downsample_*, asr_act_bwd and asr_colsum2 (same Philox mask: flat index (b*T+t)*D+k);
 * y is addressed as y[y_off + b*y_bstride + t*D + k] (elements), which covers the time-padded y16 */
int asr_dropout_downsample16_fwd(const void* y, long y_bstride, long y_off, void* z, int B, int T, int D, int T2, int rate,
                                 int style, float p, uint64_t seed, asr_stream_t stream);
int asr_dropout_downsample16_bwd(const void* dz, void* dy, int B, int T, int D, int T2, int rate, int style,
                                 float p, uint64_t seed, asr_stream_t stream);
int asr_act_bwd16(const void* dout, const void* out, void* dpre, long n, int act, asr_stream_t stream);
int asr_colsum16(const void* A, long lda, int M, int N, float* out, float* out2, int perm_h, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dropout + time down-sampling between the LSTM and `pj` (src/module.py:1059-1076).
 *   style 0 'drop'  : z[b,t2,:]      = drop(y)[b, t2*rate, :]          z is (B,T2,D)
 *   style 1 'concat': z[b,t2,i*D+:]  = drop(y)[b, t2*rate+i, :]        z is (B,T2,D*rate)
 * keep(element) = Philox4x32-10(seed, flat index in y) >= p*2^32, kept values scaled by 1/(1-p);
 * asr_dropout_mask exports the same {0,1} mask (tests feed it to the oracle).  p = 0: no dropout.
 * Backward writes the full dy (B,T,D), zeros where nothing flowed.
 */
int asr_dropout_downsample_fwd(const float* y, float* z, int B, int T, int D, int T2, int rate, int style,
                               float p, uint64_t seed, asr_stream_t stream);
int asr_dropout_downsample_bwd(const float* dz, float* dy, int B, int T, int D, int T2, int rate, int style,
                               float p, uint64_t seed, asr_stream_t stream);
int asr_dropout_mask(float* mask, long n, float p, uint64_t seed, asr_stream_t stream);

/* dpre = dout * act'(out) for act in {TANH, RELU} (autograd of torch.tanh / nn.ReLU on the path). */
int asr_act_bwd(const float* dout, const float* out, float* dpre, long n, int act, asr_stream_t stream);
/* out[j] += sum_i A[i*lda + j]  (bias gradients). */
int asr_colsum(const float* A, long lda, int M, int N, float* out, asr_stream_t stream);
/* the same sums added to two vectors (an LSTM's bias_ih and bias_hh gradients are the same column sums); out2 may be NULL */
int asr_colsum2(const float* A, long lda, int M, int N, float* out, float* out2, asr_stream_t stream);
/* row-wise log_softmax (src/asr.py:120) and the backward of log_softmax(ReLU(.)) of the CTC head
 * (src/asr.py:29-32,120): dpre = (act > 0) ? dlogp - exp(logp) * rowsum(dlogp) : 0. */
int asr_log_softmax(const float* x, float* out, long rows, int V, asr_stream_t stream);
int asr_logsoftmax_relu_bwd(const float* dlogp, const float* logp, const float* act, float* dpre,
                            long rows, int V, asr_stream_t stream);
/* LayerNorm over the last axis (+ optional fused ReLU): nn.LayerNorm in RNNLayer.ln (src/module.py:1031,1057)
 * and CNNLayerNorm (src/module.py:546-550).  stats (rows,2) = mean, rstd.  dw/db are accumulated. */
int asr_layernorm_fwd(const float* x, const float* w, const float* b, float* y, float* stats,
                      long rows, int n, float eps, int relu, asr_stream_t stream);
int asr_layernorm_bwd(const float* dy, const float* x, const float* w, const float* b, const float* stats,
                      float* dx, float* dw, float* db, long rows, int n, int relu, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * CTC loss, torch.nn.CTCLoss(blank=0, zero_infinity=False) with reduction 'mean' as called at
 * bin/train_asr.py:135,237 (inputs are batch-major here: logp (B,T,V)).
 *   nll (B) per-utterance negative log-likelihood; *loss = mean_b nll_b / max(target_len_b,1)
 *   grad (B,T,V) = gscale * d loss / d logits in the folded form torch returns for log-softmax inputs
 *   (exp(logp) - posterior), exactly 0 for t >= input_len; +inf / NaN for infeasible alignments.
 * workspace: asr_ctc_loss_workspace_bytes(B,T,L) (alpha lattice).
 * Limit: 2*L+1 <= 1024 lattice states (L = padded target width <= 511 tokens), one state per thread of a workgroup;
 * longer targets return ASR_E_UNSUPPORTED.
 */
size_t asr_ctc_loss_workspace_bytes(int B, int T, int L);
int asr_ctc_loss(const float* logp, const int64_t* targets, const int64_t* input_len, const int64_t* target_len,
                 float* nll, float* loss, float* grad, int B, int T, int V, int L, float gscale,
                 void* workspace, size_t workspace_bytes, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Sequence loss of the attention decoder + gradient (bin/train_asr.py:131-134,245; src/util.py:11-25).
 *   mode 0: CrossEntropyLoss(ignore_index=0);  mode 1: LabelSmoothingLoss(classes, smoothing)
 *   logits (B,L,V); targets (B,target_ld) int64; dlogits = gscale * d loss / d logits; accum2: scratch of 4 floats, 8-byte aligned (loss sum as 64-bit fixed point: order-independent, + count).
 */
int asr_xent(const float* logits, const int64_t* targets, long target_ld, float* dlogits, float* loss,
             float* accum2, int B, int L, int V, int mode, int classes, float smoothing, float gscale,
             asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attention decoder loop (src/asr.py:123-175; Attention.forward :333-364; LocationAwareAttention
 * src/module.py:1152-1173; Decoder.forward src/asr.py:259-266; pre_embed :35,128-133), num_head = 1,
 * v_proj = False, teacher forcing (tf_rate = 1) when `teacher` is given, greedy argmax otherwise.
 */
#define ASR_MAX_DEC_LAYERS 4
typedef struct {
    int B, Tp, E;       /* batch, encoder frames T', encoder feature dim */
    int A, Q;           /* attention dim, query dim = Dd*NL */
    int Dd, NL, V;      /* decoder LSTM dim, layers, vocabulary */
    int Kn, Ks;         /* loc_kernel_num, loc_kernel_size (taps = 2*Ks+1) */
    int L;              /* decode steps */
    float temperature;
} asr_dec_dims_t;

typedef struct {        /* reference state_dict tensors */
    const float *Wq, *bq;            /* attention.proj_q  (A,Q),(A) */
    const float *Wk, *bk;            /* attention.proj_k  (A,E),(A) */
    const float *Wconv;              /* attention.att_layer.loc_conv.weight (Kn,1,2Ks+1) */
    const float *Wproj;              /* attention.att_layer.loc_proj.weight (A,Kn) */
    const float *wg, *bg;            /* attention.att_layer.gen_energy (1,A),(1) */
    const float *emb;                /* pre_embed.weight (V,Dd) */
    const float *Wih[ASR_MAX_DEC_LAYERS], *Whh[ASR_MAX_DEC_LAYERS];   /* decoder.layers.weight_{ih,hh}_l* */
    const float *bih[ASR_MAX_DEC_LAYERS], *bhh[ASR_MAX_DEC_LAYERS];
    const float *Wc, *bc;            /* decoder.char_trans (V,Dd),(V) */
} asr_dec_weights_t;

typedef struct {        /* gradient accumulators, same shapes (+=) */
    float *Wq, *bq, *Wk, *bk, *Wconv, *Wproj, *wg, *bg, *emb;
    float *Wih[ASR_MAX_DEC_LAYERS], *Whh[ASR_MAX_DEC_LAYERS], *bih[ASR_MAX_DEC_LAYERS], *bhh[ASR_MAX_DEC_LAYERS];
    float *Wc, *bc;
} asr_dec_grads_t;

typedef struct {        /* forward outputs / saved activations, caller-allocated */
    float* key;         /* (B,Tp,A)   tanh(proj_k(enc)) */
    float* att;         /* (B,L,Tp)   attention weights per step (= att_seq of the reference, num_head 1) */
    float* q;           /* (B,L,A)    tanh(proj_q(.)) */
    float* xin;         /* (B,L,Dd+E) decoder input [embedding | context] */
    float* gates;       /* (B,L,NL,4Dd) activated gates; overwritten with pre-activation gradients by backward */
    float* cs;          /* (B,L,NL,Dd) */
    float* hs;          /* (B,L,NL,Dd) */
    float* logits;      /* (B,L,V)    att_output */
    float* energy;      /* (B,Tp)     scratch */
    float* conv;        /* (B,L,Kn,Tp) location-convolution output per step, kept for backward; may be NULL for inference */
    void* key16;        /* (B,Tp,A) bf16 working copy of key for the step kernels; NULL = read the fp32 tensor (fp32 mode) */
    void* enc16;        /* (B,Tp,E) bf16 working copy of enc, filled by asr_att_decoder_fwd; NULL = read enc */
    void* work;         /* optional scratch (256B aligned, asr_att_decoder_fwd_work_bytes) enabling the single-launch forward */
    size_t work_bytes;
    int64_t* tokens;    /* (B,L)      input token of each step (<sos>=0 first) */
} asr_dec_state_t;

int asr_att_decoder_fwd(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights,
                        const float* enc, const int64_t* enc_len, const int64_t* teacher, int teacher_ld,
                        const asr_dec_state_t* state, int prec, asr_stream_t stream);
size_t asr_att_decoder_bwd_workspace_bytes(const asr_dec_dims_t* dims);
/* bit 0 / bit 1: run the teacher-forced forward / backward loop as one persistent launch where the shape has a plan
 * (default on, env ASR_DEC_PERSIST / ASR_DEC_PERSIST_BWD); bit 2 / bit 3: prefer the streamed-tile plan (csrc/decoder_stream.hip)
 * forward / backward even where the LDS-resident plan exists (env ASR_DEC_STREAM=1); returns the previous flags.  For A/B tests. */
int asr_att_decoder_set_persistent(int flags);
/* Which plan asr_att_decoder_fwd / _bwd take for this shape (reference loop: src/asr.py:123-175): 0 = per-step kernels,
 * 1 = one persistent launch with the key / enc tiles resident in LDS (B <= 16 x T' <= 640, B <= 8 up to T' = 750 / 850),
 * 2 = one persistent launch with the tiles streamed from L2 / HBM (any T', B <= 64: the reference's B = 8 batches of up to
 * 3 400 frames, src/collect_batch.py:21-24, and BASELINE config 5). */
int asr_att_decoder_fwd_plan(const asr_dec_dims_t* dims);
int asr_att_decoder_bwd_plan(const asr_dec_dims_t* dims);
/* tiles per utterance when asr_att_decoder_bwd runs the whole loop as ONE persistent launch (0: no plan for this shape,
 * the per-step kernels run); offset of that launch's 4 KB status block (abort word first) inside the workspace. */
int asr_att_decoder_bwd_persistent_tiles(const asr_dec_dims_t* dims);
size_t asr_att_decoder_bwd_status_offset(const asr_dec_dims_t* dims);
/* bytes of state->work that let asr_att_decoder_fwd run the teacher-forced loop as ONE persistent launch (0: shape has no plan) */
size_t asr_att_decoder_fwd_work_bytes(const asr_dec_dims_t* dims);
/* dlogits (B,L,V) in; denc (B,Tp,E) accumulated (+=); parameter gradients accumulated into `grads`.
 * workspace must be 256B aligned. */
int asr_att_decoder_bwd(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                        const float* enc, const int64_t* enc_len, const asr_dec_state_t* state,
                        const float* dlogits, float* denc,
                        void* workspace, size_t workspace_bytes, int prec, asr_stream_t stream);

/* The same in two halves, for callers that overlap the parameter gradients with later work: asr_att_decoder_bwd_ex with
 * defer_params != 0 stops when the gradient wrt the encoder output is complete (denc; the critical path of the backward
 * pass) and reports in *looped_out which loop implementation ran; asr_att_decoder_bwd_params then computes every parameter
 * gradient of the decoder from the SAME workspace and saved state - on any stream, once the first half has completed
 * there.  asr_att_decoder_bwd = _ex(defer_params 0).  (Reference: one autograd backward, /root/reference/src/asr.py:155-249.) */
int asr_att_decoder_bwd_ex(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                           const float* enc, const int64_t* enc_len, const asr_dec_state_t* state,
                           const float* dlogits, float* denc,
                           void* workspace, size_t workspace_bytes, int prec, int defer_params, int* looped_out,
                           asr_stream_t stream);
int asr_att_decoder_bwd_params(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const asr_dec_grads_t* grads,
                               const float* enc, const int64_t* enc_len, const asr_dec_state_t* state, const float* dlogits,
                               void* workspace, size_t workspace_bytes, int looped, int prec, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-buffer step: global-norm clip + NaN guard (src/solver.py:96-103) fused with torch.optim.Adadelta
 * (src/optim.py:29,53-54).  normsq: device pointer to the sum of squares of `grad` (asr_sumsq);
 * grad_mul is applied to the stored gradient first (1/world_size when the buffer holds an all-reduced sum).
 *
 * Status word.  A persistent launch (asr_lstm_*, asr_att_decoder_*) whose workgroups could not hand data to each other
 * within a bounded number of polls (workgroups not co-resident, e.g. the GPU shared with another stream's kernels) sets
 * the abort word = the first 32-bit word of its workspace / status block and returns garbage.  The entry points never
 * synchronise, so the caller folds those words into ONE sticky device word after the step's launches:
 *   asr_status_collect(abort_words[n <= 32] (HOST array of device pointers), n, status): bit i of *status is set when
 *   abort word i is non-zero.  asr_adadelta_step(..., status, ...) refuses the update while *status != 0 (like the
 *   NaN guard); the host reads the word wherever it synchronises anyway and raises (src/step.py, src/solver.py).
 */
int asr_sumsq(const float* x, long n, double* out, asr_stream_t stream);
int asr_scale(float* x, long n, float k, asr_stream_t stream);
/* The loss mix with every scalar on the device (bin/train_asr.py:229-248: total = w ctc + (1 - w) att, then backward):
 *   asr_loss_mix  : out[0] = a[0] wa[0] + b[0] wb[0]  (b, wb may be NULL) - the forward and, with a = grad_output, each branch of the backward;
 *   asr_scale_dev : out[i] = in[i] alpha[0] - a loss kernel's stored gradient times its grad_output (autograd of CTCLoss / CrossEntropy). */
int asr_scale_dev(const float* in, float* out, long n, const float* alpha, asr_stream_t stream);
int asr_loss_mix(const float* a, const float* wa, const float* b, const float* wb, float* out, asr_stream_t stream);
int asr_status_collect(const void* const* abort_words, int n, unsigned* status, asr_stream_t stream);
int asr_adadelta_step(float* param, const float* grad, float* square_avg, float* acc_delta, long n,
                      float lr, float rho, float eps, float weight_decay, float clip,
                      const double* normsq, float grad_mul, const unsigned* status, asr_stream_t stream);
/* torch.optim.Adam (the reference's LM trainer: /root/reference/bin/train_lm.py:38, src/optim.py:27-28) behind the same clip /
 * NaN guard / status refusal; bias corrections in double.  `step_counter` (device, may be NULL): number of updates APPLIED so
 * far - the kernel uses *step_counter + 1 and the call advances it only when the update was applied (torch advances its step
 * on applied steps only); with NULL, `step` = 1-based update count from the host.  max_exp_avg_sq: amsgrad state or NULL. */
int asr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, long n,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int step, float clip,
                  const double* normsq, float grad_mul, const unsigned* status, unsigned long long* step_counter, asr_stream_t stream);
/* Embedding gradient (nn.Embedding backward of the RNN-LM, /root/reference/src/lm.py:16,28): demb[v,:] += sum of the rows r
 * of dy (rows x width, leading dimension dy_ld) with idx[r] == v; fixed summation order. */
int asr_embedding_bwd(const float* dy, long dy_ld, const int64_t* idx, float* demb, int rows, int width, int V, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Acoustic front-end, batched on the GPU (the reference runs it per utterance in DataLoader workers).
 *   asr_fbank: ExtractAudioFeature.forward (src/audio.py:158-171, 231-244).  wav (B,N) fp32 zero-padded,
 *     wav_len (B) samples; out (B,T,nmel), T >= 1 + (max(wav_len)-1)/hop, frames beyond an utterance are 0.
 *     dft_table (2*(n_fft/2+1), win): rows f = cos, rows nb+f = -sin of 2*pi*f*(m + (n_fft-win)/2)/n_fft;
 *     mel_fb (nmel, n_fft/2+1); window (win).  Tables are built by the host (src/audio.py).
 *   asr_delta_stack: Delta + Postprocess (src/audio.py:59-93, 108-121).  x (B,T,F), lens (B) frames,
 *     filters (channels, taps) as Delta._create_filters builds them; out (B,T,channels*F) channel-major.
 *   asr_specaug: Augment (src/audio.py:364-406), in place on x (B,T,D) using lens.  draws_in (B,6) int32 =
 *     {t, t0, tend, f, f0, fend} in the reference's draw order, or NULL to draw on the device
 *     (Philox, seed); draws_out (B,6) optional.
 */
size_t asr_fbank_workspace_bytes(int B, int T, int win, int n_fft);
int asr_fbank(const float* wav, const int64_t* wav_len, float* out, const float* dft_table, const float* mel_fb,
              const float* window, int B, int N, int T, int win, int hop, int n_fft, int nmel,
              float preemph, float ref_db, float min_db,
              void* workspace, size_t workspace_bytes, asr_stream_t stream);
int asr_delta_stack(const float* x, const int64_t* lens, float* out, const float* filters,
                    int B, int T, int F, int channels, int taps, asr_stream_t stream);
int asr_specaug(float* x, const int64_t* lens, const int* draws_in, int* draws_out, int B, int T, int D,
                int time_width, int freq_width, uint64_t seed, asr_stream_t stream);
/* the same with the reductions spread over 16 workgroups per utterance (two launches; workspace of asr_specaug_workspace_bytes(B)) */
size_t asr_specaug_workspace_bytes(int B);
int asr_specaug_ws(float* x, const int64_t* lens, const int* draws_in, int* draws_out, int B, int T, int D,
                   int time_width, int freq_width, uint64_t seed, void* workspace, size_t workspace_bytes, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * VGG front-ends (VGGExtractor src/module.py:659-716, VGGExtractor_LN :582-657) on channel-last images
 * (B,T,F,C).  asr_conv3x3 = nn.Conv2d(k=3, stride 1, pad 1) as an implicit GEMM on MFMA:
 *   mode 0: out[(b,t,f), n] (+)= act( sum_{tap,ci} img[(b,t+dt,f+df), ci] * w[n][tap*C+ci] + bias[n] )
 *           (forward with the (Co, 9*Ci) weight copy; input gradient with the flipped (Ci, 9*Co) copy)
 *   mode 1: dw[n][tap*C+ci] += sum_pixels dout[pixel, n] * img[pixel shifted by tap, ci]   (w_or_dout = dout)
 * asr_conv_weight_permute builds those copies from / folds the gradient back into the reference's
 * (Co,Ci,3,3) tensors (modes 0,1,2).  asr_maxpool2x2_* = nn.MaxPool2d(2, stride 2[, ceil_mode]); idx keeps
 * the arg-max (0..3).  asr_ln_freq_* = CNNLayerNorm (LayerNorm over F, affine per f) + optional ReLU;
 * stats (rows*C, 2).  asr_permute_last2: out[r,b,a] = in[r,a,b] (channel-major <-> channel-last).
 */
int asr_conv3x3(const float* img, const float* w_or_dout, float* out, const float* bias,
                int B, int T, int F, int C, int N, int mode, int act, int accum, int prec, asr_stream_t stream);

/* ---- bf16 VGG front-end (bf16 contraction mode), csrc/vgg16.hip ----------------------------------------------------
 * The same layers - nn.Conv2d(k 3, stride 1, pad 1) + ReLU / CNNLayerNorm, nn.MaxPool2d(2, 2[, ceil_mode]) of
 * /root/reference/src/module.py:599-614 (VGGExtractor_LN) and :670-681 (VGGExtractor) - on ZERO-BORDERED channel-last bf16
 * images P(T,F,C) = (B, T+2, F+2, C): a tap of the 3x3 stencil is then a constant row shift of the pixel-row matrix and the
 * convolution is an implicit GEMM on the direct-to-LDS bf16 contraction kernel (no bounds tests, no fp32 staging).
 *   asr_vgg16_im2col      feature (B,T,Cin*F) fp32 -> patch matrix (B*(T+2)*(F+2), Kp) bf16 of the FIRST layer (Cin = 4: K = 36 -> Kp 40)
 *   asr_conv_weight_pack16 (Co,Ci,3,3) fp32 -> (Co, Kp) bf16 [tap*Ci+ci] (mode 0) / flipped (Ci, Kp) [tap*Co+co] (mode 1: input gradient)
 *   asr_conv3x3_16        out P(T,F,N) = act(conv(img) + bias), borders written as zeros; implicit = 1: img = P(T,F,C), C % 64 == 0,
 *                         K = 9*C; implicit = 0: img = an explicit (rows, K) patch matrix over the same pixel grid
 *   asr_conv3x3_16_wgrad  dw (N, ldw)[n][tap*C+ci] += sum_rows dout[row,n] img[row+shift(tap),ci]: nine shifted-row TN contractions in
 *                         one launch (rows shifted outside the image read as zeros)
 *   asr_conv_weight_fold  (Co,Ci,3,3) += (Co, ld)[tap*Ci+ci]
 *   asr_maxpool2x2_16_*   on bordered images; idx: one byte per element of the bordered output
 *                         out_f32 = 1: `out` is fp32 (no activation) - the pre-activations a CNNLayerNorm normalises next (rounded to bf16
 *                         first, their rounding error would be amplified by mean / std)
 *   asr_ln_freq16_*       CNNLayerNorm (LayerNorm over the F interior pixels, affine per f) + ReLU: x fp32 bordered in, y / dy / dx bf16
 *                         bordered; stats (B*(T+2)*C, 2); dconv_bias (C, may be NULL) += per-channel sum of dx in fp32 = the gradient of
 *                         the bias of the convolution in front (analytically zero)
 *   asr_vgg16_output[_bwd] P(T,F,C) <-> (B, T, C*F) bf16, the encoder layout (channel-major), and its adjoint */
int asr_vgg16_im2col(const float* feature, void* x1, int B, int T, int F, int Cin, int Kp, asr_stream_t stream);
int asr_conv_weight_pack16(const float* src, void* dst, int Co, int Ci, int Kp, int mode, asr_stream_t stream);
int asr_conv_weight_fold(const float* src, float* dst, int Co, int Ci, int ld, asr_stream_t stream);
int asr_conv3x3_16(const void* img, const void* w, void* out, const float* bias, int B, int T, int F, int C, int N, int K, int implicit,
                   int act, int out_f32, asr_stream_t stream);
int asr_conv3x3_16_wgrad(const void* img, const void* dout, float* dw, int B, int T, int F, int C, int N, int ldw, int splits,
                         asr_stream_t stream);
int asr_maxpool2x2_16_fwd(const void* x, void* y, unsigned char* idx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream);
int asr_maxpool2x2_16_bwd(const void* dy, const unsigned char* idx, void* dx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream);
int asr_ln_freq16_fwd(const float* x, const float* w, const float* b, void* y, float* stats, int B, int T, int F, int C, float eps, int relu,
                      asr_stream_t stream);
int asr_ln_freq16_bwd(const void* dy, const float* x, const float* w, const float* b, const float* stats, void* dx, float* dw, float* db,
                      float* dconv_bias, int B, int T, int F, int C, int relu, asr_stream_t stream);
int asr_vgg16_output(const void* img, void* out, int B, int T, int F, int C, asr_stream_t stream);
int asr_vgg16_output_bwd(const void* dout, void* g, int B, int T, int F, int C, asr_stream_t stream);
int asr_conv_weight_permute(const float* src, float* dst, int Co, int Ci, int mode, asr_stream_t stream);
int asr_maxpool2x2_fwd(const float* x, float* y, unsigned char* idx, int B, int T, int F, int C, int T2, int F2, asr_stream_t stream);
int asr_maxpool2x2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int T, int F, int C, int T2, int F2,
                       asr_stream_t stream);
int asr_ln_freq_fwd(const float* x, const float* w, const float* b, float* y, float* stats, long rows, int F, int C,
                    float eps, int relu, asr_stream_t stream);
int asr_ln_freq_bwd(const float* dy, const float* x, const float* w, const float* b, const float* stats,
                    float* dx, float* dw, float* db, long rows, int F, int C, int relu, asr_stream_t stream);
int asr_permute_last2(const float* in, float* out, long rows, int A, int Bd, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Model variants outside the shipped configs (csrc/variants.hip, composed step by step by src/variants.py
 * as the reference's own Python loop does, src/asr.py:131-170).  All fp32, all row-major.
 *   asr_masked_softmax_fwd/bwd : BaseAttention._attend (src/module.py:1110-1118): attn = softmax(energy / temperature) over
 *                                t < len[row / NH], exactly 0 beyond; rows = B * NH ordered b * NH + head (src/module.py:1107).
 *   asr_loc_energy_fwd/bwd     : LocationAwareAttention.forward (src/module.py:1176-1182) for any num_head:
 *                                energy[r,t] = gen_energy(tanh(key[r,t,:] + q[r,:] + tanh(loc_pre[r / NH,t,:]))); loc_pre (B,T,D) is the
 *                                loc_proj output BEFORE its tanh.  bwd: dkey += (same key at every decoder step), dq / dloc_pre written,
 *                                dwg / dbg += (parameter gradients).
 *   asr_loc_conv_fwd/bwd       : loc_conv = nn.Conv1d(NH, Kn, 2 Ks + 1, padding Ks, bias False) on prev_att (B,NH,T), output already
 *                                transposed to (B,T,Kn) (src/module.py:1176); bwd: dprev written (may be NULL), dW +=.
 *   asr_lstm_cell_fwd/bwd      : nn.LSTM cell on gx + gh (both (N,4D), biases inside; gate order i,f,g,o) with its backward
 *                                (decoder layers, src/asr.py:204,262); act (N,4D) = activated gates, dgates (N,4D) serves both halves.
 *   asr_gru_cell_fwd/bwd       : nn.GRU cell (decoder module GRU, src/asr.py:203-204): gi, gh (N,3D) gate order r,z,n; saved (N,4D) =
 *                                r | z | n | gh_n; bwd: dgi, dgh (N,3D) and the direct part of dh_prev (= dh * z).
 *   asr_gru_fwd/bwd            : nn.GRU(bidirectional, batch_first) time recurrence of an encoder layer (src/module.py:1023,1049; zero
 *                                initial state, padded frames processed like the reference does): gi (B,T,ND,3H) = x W_ih^T + b_ih,
 *                                whhT (ND,H,3H) = weight_hh transposed, bhh (ND,3H); y (B,T,ND*H); saved (B,T,ND,4H).
 *                                bwd: whh (ND,3H,H) as stored; dgi / dgh (B,T,ND,3H) from which the caller forms dW_ih, db_ih, dx and
 *                                dW_hh (shifted rows, asr_gemm seqT), db_hh.
 */
int asr_masked_softmax_fwd(const float* energy, const int64_t* len, int rows, int NH, int T, float temperature, float* attn, asr_stream_t stream);
int asr_masked_softmax_bwd(const float* attn, const float* dattn, int rows, int T, float temperature, float* denergy, asr_stream_t stream);
int asr_loc_energy_fwd(const float* key, const float* q, const float* loc_pre, const float* wg, const float* bg,
                       int B, int NH, int T, int D, float* energy, asr_stream_t stream);
int asr_loc_energy_bwd(const float* key, const float* q, const float* loc_pre, const float* wg, const float* denergy,
                       int B, int NH, int T, int D, float* dkey, float* dq, float* dloc_pre, float* dwg, float* dbg, asr_stream_t stream);
int asr_loc_conv_fwd(const float* prev_att, const float* W, int B, int NH, int T, int Kn, int Ks, float* out, asr_stream_t stream);
int asr_loc_conv_bwd(const float* dout, const float* prev_att, const float* W, int B, int NH, int T, int Kn, int Ks,
                     float* dprev, float* dW, asr_stream_t stream);
int asr_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, int N, int D, float* act, float* h, float* c, asr_stream_t stream);
int asr_lstm_cell_bwd(const float* act, const float* c_prev, const float* c, const float* dh, const float* dc, int N, int D,
                      float* dgates, float* dc_prev, asr_stream_t stream);
int asr_gru_cell_fwd(const float* gi, const float* gh, const float* h_prev, int N, int D, float* saved, float* h, asr_stream_t stream);
int asr_gru_cell_bwd(const float* saved, const float* h_prev, const float* dh, int N, int D, float* dgi, float* dgh, float* dh_prev,
                     asr_stream_t stream);
int asr_gru_fwd(const float* gi, const float* whhT, const float* bhh, int B, int T, int H, int ND, float* y, float* saved, asr_stream_t stream);
int asr_gru_bwd(const float* dy, const float* y, const float* saved, const float* whh, int B, int T, int H, int ND,
                float* dgi, float* dgh, asr_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Beam-search inference (BeamDecoder.forward src/decode.py:65-183), batched over live hypotheses.
 *   asr_att_decoder_keys : key = tanh(proj_k(enc)) once per utterance (src/asr.py:345).
 *   asr_att_decoder_step : ONE decode step t for all dims->B rows of `state` (each row = one hypothesis at step t):
 *                          the caller sets state->tokens[:,t] and, for t>0, the step t-1 entries of hs/cs/att of
 *                          every row; outputs att/xin/gates/cs/hs/logits[:,t]   (src/decode.py:107-116).
 *   asr_ctc_prefix_init / asr_ctc_prefix_score : CTCPrefixScore.init_state / cheap_compute (src/ctc.py:19-27,68-107)
 *                          for N hypotheses x C candidates: logp (T,V); r_prev (N,T,2); candidates (N,C) int32;
 *                          prefix_len, last_token (N) int32; psi (N,C); r_out (N,C,T,2).  fp32, logzero = -1e8.
 *   asr_lstm_cell        : pointwise LSTM cell on gate pre-activations (N,4D) + both biases (RNNLM step, src/lm.py:27-37;
 *                          the projections are asr_gemm);  c_prev may be NULL (zero state).
 *   asr_gather_rows      : dst[r,:] = src[idx[r],:]  (embedding lookup, state re-ordering after pruning).
 */
int asr_att_decoder_keys(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights, const float* enc,
                         float* key, int prec, asr_stream_t stream);
int asr_att_decoder_step(const asr_dec_dims_t* dims, const asr_dec_weights_t* weights,
                         const float* enc, const int64_t* enc_len, const asr_dec_state_t* state, int t,
                         int prec, asr_stream_t stream);
int asr_ctc_prefix_init(const float* logp, float* r, int T, int V, asr_stream_t stream);
int asr_ctc_prefix_score(const float* logp, const float* r_prev, const int* candidates, const int* prefix_len,
                         const int* last_token, float* psi, float* r_out, int N, int C, int T, int V,
                         asr_stream_t stream);
int asr_lstm_cell(const float* gates_pre, const float* bias_ih, const float* bias_hh, const float* c_prev,
                  float* h, float* c, int N, int D, asr_stream_t stream);
int asr_gather_rows(const float* src, const int64_t* idx, float* dst, int rows, int width, long src_ld, long dst_ld,
                    int nsrc, asr_stream_t stream);

/* Device-side beam bookkeeping: one output position of BeamDecoder.forward (src/decode.py:104-177) for U utterances x
 * `beam` hypothesis rows (row = u*beam + i) with NO device-to-host copy.
 *   asr_beam_candidates : candidates (rows,C) = att_logp.topk(C) per row (the CTC scorer's candidates, :129), descending.
 *   asr_ctc_prefix_{init,score}_batched : the prefix scorer over a batch: row n belongs to utterance n / rows_per_utt with
 *       log-probs logp[utt] (Tmax,V) and tlen[utt] valid frames; r (rows,Tmax,2), r_out (N,C,Tmax,2).
 *   asr_beam_step : score fusion (1-w)*att + w*(psi - ctc_prob) on the candidates / LOG_ZERO elsewhere, blank excluded
 *       (:131-141), + lm_weight*lm (:152); per live hypothesis top-`beam` and the <eos> rule of Hypothesis.addTopk
 *       (att[eos] > eos_threshold * max(att[2:]) ends it with that score, :235-242); children pruned to the `beam` best by
 *       average token score in stable order (:175-177); finished hypotheses kept as the `beam` best per utterance, sorted
 *       (:179-183, incl. the survivors when t+1 reaches max_len[u], and the beam-1 early return).  State is double
 *       buffered (in -> out); outputs for the next step: last_token (rows), parent (rows: source row of each new row),
 *       ctc_index (rows: parent*C + candidate slot, row index into r_out viewed (N*C, Tmax*2)), and column t+1 of the
 *       decoder's token table.  beam <= 16, V <= 2048, one wave per utterance. */
typedef struct {
    const float *att_logp, *lm_logp, *psi;  /* (R,V); (R,V) or NULL; (R,C) or NULL */
    const int* candidates;                  /* (R,C) or NULL */
    const int* alive_in; const float* sum_in; const float* ctcp_in; const int* len_in; const int* seq_in; const float* score_in;
    int* alive_out; float* sum_out; float* ctcp_out; int* len_out; int* seq_out; float* score_out;   /* seq/score: (R,Lmax) */
    int* last_token; int64_t* parent; int64_t* ctc_index;
    int64_t* tokens; long tokens_ld;        /* decoder token table (R, tokens_ld) */
    const int *min_len, *max_len; int* done;                   /* (U) */
    int* fin_n; int* fin_len; float* fin_avg; int* fin_seq; float* fin_score;   /* (U), (U,beam), (U,beam), (U,beam,Lmax+1) x2 */
    int U, beam, V, C, Lmax, t;
    float ctc_weight, lm_weight, eos_threshold;
} asr_beam_step_t;
/* Scheduled sampling (src/asr.py:151-158): out[row*out_ld] ~ Categorical(softmax(logits[row*ld : row*ld+V])), one Philox
 * uniform per (seed, row). */
int asr_sample_tokens(const float* logits, long ld, int64_t* out, long out_ld, int rows, int V, uint64_t seed, asr_stream_t stream);
int asr_beam_candidates(const float* att_logp, int* candidates, int rows, int V, int C, asr_stream_t stream);
int asr_beam_step(const asr_beam_step_t* args, asr_stream_t stream);
int asr_ctc_prefix_init_batched(const float* logp, const int* tlen, float* r, int rows, int rows_per_utt, int Tmax, int V,
                                asr_stream_t stream);
int asr_ctc_prefix_score_batched(const float* logp, const int* tlen, const float* r_prev, const int* candidates,
                                 const int* prefix_len, const int* last_token, float* psi, float* r_out,
                                 int N, int C, int Tmax, int V, int rows_per_utt, asr_stream_t stream);

/* Test support: keeps `workgroups` compute units busy (one 64-thread workgroup each holding `lds_bytes` of LDS) for
 * `seconds` (<= 20) on `stream`; tests/test_persist_abort.py uses it to starve a persistent launch of co-residency. */
int asr_debug_occupy(int workgroups, int lds_bytes, double seconds, asr_stream_t stream);

/* ---- CU-masked streams --------------------------------------------------------------------------------------------
 * The persistent recurrences keep 40 of the 256 CUs busy; the parameter-gradient contractions of the layer above are off
 * the critical path and can run beside them - but only if they never land on the recurrence's CUs (a shared CU stretches
 * every hand-off of the chain).  asr_stream_create_cu_mask creates a HIP stream restricted to `count` compute units PER XCD
 * starting at per-XCD unit `first` (0..31).  Mask layout measured on MI355X (tools/probe/cumask.hip): mask bit i = XCD i % 8,
 * unit i / 8 of that XCD (shader engine (i / 8) % 4); a mask that leaves an XCD empty is ignored by the driver.
 * (No reference counterpart: the reference trains on one CUDA stream.) */
/* Zero a hand-off work area from EVERY XCD with L2-local stores (whenever an area of the caller's hand-off pool changes hands):
 * see csrc/core.hip.  `tickets`: 64 bytes of device memory private to the calling stream (per-XCD chunk counters, cleared
 * by the call; word 15 is set when an XCD received no workgroup - register it like an abort word).  Placement-independent:
 * workgroups draw chunks from the counter of the XCD they find themselves on.  No reference counterpart. */
int asr_scrub_workspace(void* ptr, size_t bytes, void* tickets, asr_stream_t stream);
int asr_stream_create_cu_mask(int first, int count, asr_stream_t* stream);
int asr_stream_destroy(asr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ASR_HIP_H */
