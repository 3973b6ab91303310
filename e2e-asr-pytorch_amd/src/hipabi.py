"""ctypes binding of libasr_hip.so (the C ABI declared in include/asr_hip.h).

There is no CPU fallback: importing this module without the built library raises, and every
wrapper refuses non-CUDA tensors.  PyTorch is used for device memory and streams only.
"""
import atexit
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ASR_HIP_LIB') or os.path.join(os.path.dirname(_HERE), 'lib', 'libasr_hip.so')      # ASR_HIP_LIB: A/B builds (tools/), never a fallback

F32, BF16 = 0, 1
DEBUG_KEEP = os.environ.get('ASR_DEBUG_KEEP', '0') == '1'      # layers keep references to their saved activations (tools/dbg_*.py)
ACT_NONE, ACT_TANH, ACT_RELU = 0, 1, 2
MAX_DEC_LAYERS = 4

_vp, _i, _l, _f, _u64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_uint64, ctypes.c_size_t


class DecDims(ctypes.Structure):
    _fields_ = [(n, _i) for n in ('B', 'Tp', 'E', 'A', 'Q', 'Dd', 'NL', 'V', 'Kn', 'Ks', 'L')] + [('temperature', _f)]


_DEC_W_FIELDS = ['Wq', 'bq', 'Wk', 'bk', 'Wconv', 'Wproj', 'wg', 'bg', 'emb']


class DecWeights(ctypes.Structure):
    _fields_ = [(n, _vp) for n in _DEC_W_FIELDS] + \
               [(n, _vp * MAX_DEC_LAYERS) for n in ('Wih', 'Whh', 'bih', 'bhh')] + [('Wc', _vp), ('bc', _vp)]


class DecState(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in ('key', 'att', 'q', 'xin', 'gates', 'cs', 'hs', 'logits', 'energy', 'conv', 'key16', 'enc16', 'work')]
                + [('work_bytes', ctypes.c_size_t), ('tokens', _vp)])


class BeamStep(ctypes.Structure):
    _fields_ = ([(n, _vp) for n in ('att_logp', 'lm_logp', 'psi', 'candidates', 'alive_in', 'sum_in', 'ctcp_in', 'len_in', 'seq_in', 'score_in',
                                    'alive_out', 'sum_out', 'ctcp_out', 'len_out', 'seq_out', 'score_out', 'last_token', 'parent', 'ctc_index',
                                    'tokens')]
                + [('tokens_ld', _l)]
                + [(n, _vp) for n in ('min_len', 'max_len', 'done', 'fin_n', 'fin_len', 'fin_avg', 'fin_seq', 'fin_score')]
                + [(n, _i) for n in ('U', 'beam', 'V', 'C', 'Lmax', 't')]
                + [(n, _f) for n in ('ctc_weight', 'lm_weight', 'eos_threshold')])


_P = ctypes.POINTER

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    'asr_gemm': [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _i, _i, _l, _l, _l, _i, _i, _i, _vp],
    'asr_lstm_fwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    'asr_lstm_bwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _vp],
    'asr_dropout_downsample_fwd': [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u64, _vp],
    'asr_dropout_downsample_bwd': [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u64, _vp],
    'asr_dropout_mask': [_vp, _l, _f, _u64, _vp],
    'asr_act_bwd': [_vp, _vp, _vp, _l, _i, _vp],
    'asr_colsum': [_vp, _l, _i, _i, _vp, _vp],
    'asr_colsum2': [_vp, _l, _i, _i, _vp, _vp, _vp],
    'asr_log_softmax': [_vp, _vp, _l, _i, _vp],
    'asr_logsoftmax_relu_bwd': [_vp, _vp, _vp, _vp, _l, _i, _vp],
    'asr_layernorm_fwd': [_vp, _vp, _vp, _vp, _vp, _l, _i, _f, _i, _vp],
    'asr_layernorm_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _vp],
    'asr_ctc_loss': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _sz, _vp],
    'asr_xent': [_vp, _vp, _l, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _vp],
    'asr_att_decoder_fwd': [_P(DecDims), _P(DecWeights), _vp, _vp, _vp, _i, _P(DecState), _i, _vp],
    'asr_att_decoder_bwd': [_P(DecDims), _P(DecWeights), _P(DecWeights), _vp, _vp, _P(DecState), _vp, _vp, _vp, _sz, _i, _vp],
    'asr_att_decoder_bwd_ex': [_P(DecDims), _P(DecWeights), _P(DecWeights), _vp, _vp, _P(DecState), _vp, _vp, _vp, _sz, _i, _i,
                               ctypes.POINTER(_i), _vp],
    'asr_att_decoder_bwd_params': [_P(DecDims), _P(DecWeights), _P(DecWeights), _vp, _vp, _P(DecState), _vp, _vp, _sz, _i, _i, _vp],
    'asr_fbank': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _f, _f, _f, _vp, _sz, _vp],
    'asr_delta_stack': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'asr_specaug': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _u64, _vp],
    'asr_specaug_ws': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _u64, _vp, _sz, _vp],
    'asr_conv3x3': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'asr_conv_weight_permute': [_vp, _vp, _i, _i, _i, _vp],
    'asr_vgg16_im2col': [_vp, _vp, _i, _i, _i, _i, _i, _vp],
    'asr_conv_weight_pack16': [_vp, _vp, _i, _i, _i, _i, _vp],
    'asr_conv_weight_fold': [_vp, _vp, _i, _i, _i, _vp],
    'asr_conv3x3_16': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'asr_conv3x3_16_wgrad': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp],
    'asr_maxpool2x2_16_fwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'asr_maxpool2x2_16_bwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'asr_ln_freq16_fwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i, _vp],
    'asr_ln_freq16_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'asr_vgg16_output': [_vp, _vp, _i, _i, _i, _i, _vp],
    'asr_vgg16_output_bwd': [_vp, _vp, _i, _i, _i, _i, _vp],
    'asr_maxpool2x2_fwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'asr_maxpool2x2_bwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    'asr_ln_freq_fwd': [_vp, _vp, _vp, _vp, _vp, _l, _i, _i, _f, _i, _vp],
    'asr_ln_freq_bwd': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _l, _i, _i, _i, _vp],
    'asr_permute_last2': [_vp, _vp, _l, _i, _i, _vp],
    'asr_att_decoder_keys': [_P(DecDims), _P(DecWeights), _vp, _vp, _i, _vp],
    'asr_att_decoder_step': [_P(DecDims), _P(DecWeights), _vp, _vp, _P(DecState), _i, _i, _vp],
    'asr_ctc_prefix_init': [_vp, _vp, _i, _i, _vp],
    'asr_ctc_prefix_score': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'asr_lstm_cell': [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    'asr_gather_rows': [_vp, _vp, _vp, _i, _i, _l, _l, _i, _vp],
    'asr_masked_softmax_fwd': [_vp, _vp, _i, _i, _i, _f, _vp, _vp],
    'asr_masked_softmax_bwd': [_vp, _vp, _i, _i, _f, _vp, _vp],
    'asr_loc_energy_fwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp],
    'asr_loc_energy_bwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    'asr_loc_conv_fwd': [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    'asr_loc_conv_bwd': [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp],
    'asr_lstm_cell_fwd': [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    'asr_lstm_cell_bwd': [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    'asr_gru_cell_fwd': [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    'asr_gru_cell_bwd': [_vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp],
    'asr_gru_fwd': [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    'asr_gru_bwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    'asr_sumsq': [_vp, _l, _vp, _vp],
    'asr_scale': [_vp, _l, _f, _vp],
    'asr_scale_dev': [_vp, _vp, _l, _vp, _vp],
    'asr_loss_mix': [_vp, _vp, _vp, _vp, _vp, _vp],
    'asr_adadelta_step': [_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _vp, _f, _vp, _vp],
    'asr_adam_step': [_vp, _vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp, _f, _vp, _vp, _vp],
    'asr_embedding_bwd': [_vp, _l, _vp, _vp, _i, _i, _i, _vp],
    'asr_status_collect': [ctypes.POINTER(_vp), _i, _vp, _vp],
    'asr_beam_candidates': [_vp, _vp, _i, _i, _i, _vp],
    'asr_sample_tokens': [_vp, _l, _vp, _l, _i, _i, _u64, _vp],
    'asr_beam_step': [ctypes.POINTER(BeamStep), _vp],
    'asr_ctc_prefix_init_batched': [_vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'asr_ctc_prefix_score_batched': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp],
    'asr_gemm16': [_vp, _vp, _vp, _vp, _i, _i, _i, _l, _l, _l, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    'asr_lstm16_fwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, ctypes.c_uint, _i, _vp],
    'asr_lstm16_bwd': [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _sz, ctypes.c_uint, _i, _vp],
    'asr_cast_bf16': [_vp, _vp, _l, _vp],
    'asr_cast_f32': [_vp, _vp, _l, _i, _vp],
    'asr_rnn_pack_weights': [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    'asr_dropout_downsample16_fwd': [_vp, _l, _l, _vp, _i, _i, _i, _i, _i, _i, _f, _u64, _vp],
    'asr_dropout_downsample16_bwd': [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _u64, _vp],
    'asr_act_bwd16': [_vp, _vp, _vp, _l, _i, _vp],
    'asr_colsum16': [_vp, _l, _i, _i, _vp, _vp, _i, _vp],
    'asr_debug_occupy': [_i, _i, ctypes.c_double, _vp],
    'asr_scrub_workspace': [_vp, _sz, _vp, _vp],
    'asr_stream_create_cu_mask': [_i, _i, ctypes.POINTER(_vp)],
    'asr_stream_destroy': [_vp],
}
_RESTYPES = {
    'asr_last_error': (ctypes.c_char_p, []),
    'asr_device_arch': (ctypes.c_char_p, []),
    'asr_version': (ctypes.c_int, []),
    'asr_lstm_workspace_bytes': (_sz, [_i, _i, _i]),
    'asr_specaug_workspace_bytes': (_sz, [_i]),
    'asr_lstm_set_persistent': (ctypes.c_int, [_i]),
    'asr_lstm_plan': (ctypes.c_int, [_i, _i, _i, _i, _i]),
    'asr_lstm16_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'asr_ctc_loss_workspace_bytes': (_sz, [_i, _i, _i]),
    'asr_fbank_workspace_bytes': (_sz, [_i, _i, _i, _i]),
    'asr_att_decoder_bwd_workspace_bytes': (_sz, [_P(DecDims)]),
    'asr_att_decoder_fwd_work_bytes': (_sz, [_P(DecDims)]),
    'asr_att_decoder_set_persistent': (ctypes.c_int, [_i]),
    'asr_att_decoder_bwd_status_offset': (_sz, [_P(DecDims)]),
    'asr_att_decoder_bwd_persistent_tiles': (ctypes.c_int, [_P(DecDims)]),
    'asr_att_decoder_fwd_plan': (ctypes.c_int, [_P(DecDims)]),
    'asr_att_decoder_bwd_plan': (ctypes.c_int, [_P(DecDims)]),
}


class HipLibraryMissing(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing(
            'libasr_hip.so not found at %s — build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(there is no CPU fallback for the HIP path)' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, (rt, at) in _RESTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = at
        fn.restype = rt
    return lib


_lib = _load()


def lib():
    return _lib


def exported_symbols():
    return list(SIGNATURES.keys()) + list(_RESTYPES.keys())


def stream_ptr():
    """HIP stream the kernels are enqueued on: torch's current stream of the CURRENT device.  The C ABI launches on
    the HIP current device, so every tensor handed to it must live there (checked in ptr())."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# ---- abort words of the persistent launches -> one sticky status word per device -------------------------------
# A persistent kernel whose workgroups cannot reach each other within its spin bound sets the first word of its
# workspace and returns garbage (include/asr_hip.h, "Status word").  Every wrapper that may have started such a launch
# registers that word here; collect_status() folds the registered words into the device's status word with ONE tiny
# kernel (no host sync), the optimizer kernel refuses the update while the word is set, and raise_if_aborted() - called
# wherever the host synchronises anyway - turns it into an exception.
_watch = {}            # device index -> list of (tensor kept alive, device address of the abort word)
_status = {}           # device index -> int32 tensor (1,)


def status_word(device=None):
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev not in _status:
        _status[dev] = torch.zeros(1, dtype=torch.int32, device='cuda:%d' % dev)
    return _status[dev]


def watch_abort(t, offset=0, release=False):
    """Registers the abort word at byte `offset` of workspace tensor `t` (kept alive until it has been collected).
    release=True: `t` came from handoff_acquire() for this one launch and goes back to the pool once collected."""
    if t is None:
        return
    lst = _watch.setdefault(t.device.index, [])
    lst.append((t, t.data_ptr() + int(offset), release))
    if len(lst) >= 32:
        collect_status()


def abort_guard(t, offset=0):
    """Call BEFORE a persistent launch on workspace `t`: the launchers clear the abort word of their workspace, so a word
    that is still registered from an earlier launch on the same workspace (validation loops reuse the per-shape areas many
    times between two optimizer steps) is folded into the status word first - an abort is never erased uncollected."""
    if t is None:
        return
    addr = t.data_ptr() + int(offset)
    for _, a, _r in _watch.get(t.device.index, ()):
        if a == addr:
            collect_status()
            return


def collect_status():
    """Folds every registered abort word into the current device's status word (asr_status_collect).  No sync."""
    dev = torch.cuda.current_device()
    lst = _watch.get(dev)
    st = status_word(dev)
    while lst:
        chunk, lst[:] = lst[:32], lst[32:]
        arr = (_vp * len(chunk))(*[a for _, a, _r in chunk])
        call('asr_status_collect', arr, len(chunk), ptr(st), stream_ptr())
        for t, _, rel in chunk:
            if rel:
                handoff_release(t)
    return st


# ---- hand-off areas: a pool of their own -----------------------------------------------------------------------------
# Workspaces of the persistent launches carry exchange granules that the kernels store L2-locally; such a line can be
# evicted from an XCD's L2 AFTER the launch, on top of whatever the address holds by then.  The areas therefore never go
# back to torch's caching allocator (where the late write-back would land in an unrelated tensor): they are allocated once,
# kept for the life of the process, handed out by size class and scrubbed from every XCD whenever one changes hands
# (asr_scrub_workspace), so the only thing a late eviction can write over a hand-off area is zeros or its own exchange data.
_pool = {'free': {}, 'all': [], 'tickets': {}}


def _size_class(nbytes):
    n = max(4096, int(nbytes))
    step = 1 << max(12, n.bit_length() - 3)          # eight classes per power of two: at most 12.5 % slack
    return (n + step - 1) // step * step


def scrub(ws):
    """Zeroes `ws` from every XCD with L2-local stores (one launch on the current stream, placement-independent)."""
    key = (ws.device.index, torch.cuda.current_stream().cuda_stream)
    tk = _pool['tickets'].get(key)
    if tk is None:
        tk = _pool['tickets'][key] = torch.zeros(16, dtype=torch.int32, device=ws.device)
    call('asr_scrub_workspace', ptr(ws), ws.numel(), ptr(tk), stream_ptr())
    watch_abort(tk, 60)


def handoff_acquire(nbytes, device):
    """A zeroed, scrubbed uint8 area of at least nbytes from the hand-off pool (stream-ordered on the current stream)."""
    device = torch.device(device)
    size = _size_class(nbytes)
    lst = _pool['free'].get((device.index, size))
    if lst:
        ws = lst.pop()
    else:
        ws = torch.zeros(size, dtype=torch.uint8, device=device)
        _pool['all'].append(ws)
    scrub(ws)
    return ws


def handoff_release(ws):
    """Back to the pool (never to the allocator); the next handoff_acquire scrubs it."""
    _pool['free'].setdefault((ws.device.index, ws.numel()), []).append(ws)


class PersistentLaunchAborted(RuntimeError):
    pass


def raise_if_aborted():
    """Host-synchronising check (one 4-byte read): raises when a persistent launch since the last check gave up."""
    st = collect_status()
    v = int(st.item())
    if v:
        st.zero_()
        raise PersistentLaunchAborted(
            'a persistent HIP launch timed out waiting for its peer workgroups (status 0x%x): the results of that '
            'step are invalid and the optimizer refused the update.  Typical cause: the GPU is shared with other '
            "work so the launch's workgroups were not co-resident; ASR_LSTM_PERSIST=0 / ASR_DEC_PERSIST=0 select the "
            'launch-per-step kernels.' % v)


# ---- side stream for work that is off the backward pass's dependency chain ------------------------
# Weight-gradient contractions and bias column sums of a layer are needed only by the optimizer, while the chain
# continues with a 40-workgroup persistent recurrence that leaves most of the chip idle: they run on a second HIP
# stream beside it.  The backward pass joins the side stream once, when the autograd engine has finished
# (engine callback), so every reader of .grad on the current stream sees complete gradients.
# Measured on MI355X (bench.py, 1 GPU): the contractions do run beside the recurrence, but they slow it by 0.7 ms, run
# 25-40 % slower themselves and the step gains nothing (34.2 vs 34.0 ms) - so this is OFF unless ASR_SIDE_STREAM=1.
_side = {'stream': None, 'pending': False, 'enabled': os.environ.get('ASR_SIDE_STREAM', '0') == '1', 'deferred': [], 'keep': []}

# Round 2: with bf16 operands AND CU-masked streams the overlap pays.  The recurrence of the bf16 encoder path is launched on
# a stream restricted to REC_UNITS compute units per XCD, the deferred parameter-gradient work on a stream restricted to the
# other ones (asr_stream_create_cu_mask), so the two never share a CU.  ASR_OVERLAP=0 switches it off (A/B runs).
# compute units per XCD of the recurrence stream; the side stream gets the other 32 - REC_UNITS.  One workgroup of a
# recurrence group per CU: H/16 of them per XCD (20 at the C2 width); configure_rec_units() derives it from the model.
REC_UNITS = int(os.environ.get('ASR_REC_UNITS', '20'))
_masked = {}


def rec_units_for(hidden):
    """CUs per XCD the persistent recurrence of an H-wide layer wants: its H/16 workgroups per group, one per CU, while at
    least 8 units stay with the side stream; wider layers (H = 512: 32 workgroups) sit two per CU (lstm_persist3.hip
    `fits_resident` admits two)."""
    p = max(1, (int(hidden) + 15) // 16)
    return p if p <= 24 else min(24, (p + 1) // 2)


def configure_rec_units(hidden_dims):
    """Called by the model constructor with the widths of its recurrent layers: the CU split between the recurrence stream
    and the side stream follows the widest layer (ASR_REC_UNITS overrides).  Changing the split drops the cached streams."""
    global REC_UNITS
    if 'ASR_REC_UNITS' in os.environ or not hidden_dims:
        return REC_UNITS
    units = max(rec_units_for(h) for h in hidden_dims)
    if units != REC_UNITS:
        if _side['stream'] is not None or _masked:
            release_streams()
        REC_UNITS = units
    return REC_UNITS


def overlap_enabled():
    return os.environ.get('ASR_OVERLAP', '1') != '0'


def overlap_dp_enabled():
    """The same overlap under data parallelism (the bucket signals move into the deferred work, so an all-reduce still starts
    only when its bucket is final).  OFF by default: validated with the recording reducer on one GPU (tests/test_dp_hooks.py),
    not yet on a multi-GPU node with RCCL kernels beside the masked streams.  ASR_OVERLAP_DP=1 switches it on."""
    return overlap_enabled() and os.environ.get('ASR_OVERLAP_DP', '0') == '1'


def ctc_side_enabled():
    """CTC head + loss on the side stream beside the decoder forward (side_branch).  OFF by default: measured on MI355X the
    16 CTC workgroups and the decoder's 240 (one per CU) need every compute unit at once; whenever three CTC workgroups share
    an XCD one decoder workgroup waits for them and the whole cluster launch with it (dec_fwd_persist 3.24 -> 3.39 ms), which
    cancels the 0.3 ms saved: 18.05 vs 18.06 ms per step.  ASR_CTC_SIDE=1 switches it on."""
    return os.environ.get('ASR_CTC_SIDE', '0') == '1'


_work = {}


def work_stream():
    """The non-default stream training runs on while the overlap is enabled.  Streams made by hipExtStreamCreateWithCUMask
    are BLOCKING streams: they serialise against the legacy default stream in both directions, so with the step on the
    default stream nothing overlaps (measured: the side work started when the recurrence had finished).  The training
    loops (bin/train_asr.py, bench.py) run inside `with torch.cuda.stream(H.work_stream())`; src.step.train_step moves a
    step that arrives on the default stream over (two cross-stream waits per step, ~0.25 ms)."""
    dev = torch.cuda.current_device()
    if dev not in _work:
        _work[dev] = torch.cuda.Stream()
    return _work[dev]


def masked_stream(first, count):
    """torch stream restricted to per-XCD compute units [first, first+count) (cached per device)."""
    key = (torch.cuda.current_device(), first, count)
    if key not in _masked:
        out = _vp()
        call('asr_stream_create_cu_mask', first, count, ctypes.byref(out))
        _masked[key] = (torch.cuda.ExternalStream(out.value), out.value)
    return _masked[key][0]


def release_streams():
    """Destroys the CU-masked streams deterministically: synchronise, drop the (non-owning) ExternalStream wrappers and
    everything queued for them, then asr_stream_destroy on each handle.  Registered with atexit: streams made by
    hipExtStreamCreateWithCUMask that are still alive when the HIP runtime's own finalisers run brought every
    `rocprofv3 -- python3 bench.py` of round 2 down with a SIGSEGV inside __cxa_finalize AFTER the tool had written its
    files; released here, before interpreter shutdown, the process exits 0."""
    if not _masked:
        return
    handles = []
    try:
        for (dev, _f, _c), (_wrap, h) in list(_masked.items()):
            with torch.cuda.device(dev):
                torch.cuda.synchronize()
            handles.append((dev, h))
    except Exception:                  # the runtime is already gone: nothing left to release
        _masked.clear()
        return
    _side['stream'], _side['deferred'], _side['pending'] = None, [], False
    del _side['keep'][:]
    _masked.clear()
    for dev, h in handles:
        with torch.cuda.device(dev):
            _lib.asr_stream_destroy(ctypes.c_void_p(h))


atexit.register(release_streams)


class on_rec_stream:
    """with on_rec_stream(): the kernels launched inside run on the recurrence stream (REC_UNITS compute units per XCD) behind
    everything issued so far on the current stream, and the current stream continues behind them."""

    def __enter__(self):
        self.cur = torch.cuda.current_stream()
        rec = masked_stream(0, REC_UNITS)
        rec.wait_stream(self.cur)
        self.rec = rec
        self.ctx = torch.cuda.stream(rec)
        self.ctx.__enter__()
        return rec

    def __exit__(self, *exc):
        r = self.ctx.__exit__(*exc)
        self.cur.wait_stream(self.rec)
        return r


def fast16_enabled():
    """bf16-storage encoder path (src/functions.RNNLayerFastFn); ASR_FAST16=0 keeps the fp32-storage kernels (A/B runs)."""
    return os.environ.get('ASR_FAST16', '1') != '0'


def side_enabled():
    return _side['enabled']


class on_side_stream:
    """with on_side_stream(event, t1, t2, ...): kernels launched inside run on the side stream once `event` (recorded on the
    producing stream; None = everything issued so far on the current stream) has completed; the listed tensors are kept
    alive (references in _side['keep']) until the current stream has joined the side stream (join_side / join_branch).
    NOT caching-allocator record_stream: a block with a recorded stream makes the allocator record an event on that stream
    when the tensor is freed - for tensors that die at interpreter shutdown that is a HIP call on a CU-masked stream after
    release_streams() has destroyed it (SIGSEGV at exit, round 3's first profile run)."""

    def __init__(self, event, *tensors):
        self.event = event
        self.tensors = [t for t in tensors if t is not None]

    def __enter__(self):
        if _side['stream'] is None:
            _side['stream'] = masked_stream(REC_UNITS, 32 - REC_UNITS) if overlap_enabled() else torch.cuda.Stream()
        side = _side['stream']
        if self.event is not None:
            side.wait_event(self.event)
        else:
            side.wait_stream(torch.cuda.current_stream())
        _side['keep'].extend(self.tensors)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


class side_branch:
    """with side_branch(wait=True, t1, ...): a whole BRANCH of the graph (the CTC head and loss, 0.3 ms on 16 workgroups) runs on
    the side stream beside the decoder loop, which leaves 16 compute units idle.  wait=True: behind everything issued so far on
    the current stream (the encoder output); wait=False: a continuation of the branch, ordered by the side stream itself.
    Autograd runs the backward of what was recorded here on the same stream.  The caller joins with join_branch()."""

    def __init__(self, wait, *tensors):
        self.wait, self.tensors = wait, [t for t in tensors if t is not None]

    def __enter__(self):
        if _side['stream'] is None:
            _side['stream'] = masked_stream(REC_UNITS, 32 - REC_UNITS) if overlap_enabled() else torch.cuda.Stream()
        side = _side['stream']
        if self.wait:
            side.wait_stream(torch.cuda.current_stream())
        _side['keep'].extend(self.tensors)
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

    def __exit__(self, *exc):
        _side['pending'] = True
        return self.ctx.__exit__(*exc)


def join_branch(*tensors):
    """The current stream waits for the side stream; the listed tensors (made on the side stream) may then be used here."""
    cur = torch.cuda.current_stream()
    if _side['stream'] is not None:
        cur.wait_stream(_side['stream'])
    _side['keep'].extend(t for t in tensors if t is not None)      # made on the side stream: freed only behind the next join_side


def defer_side(fn, *tensors):
    """Queues fn() for the side stream.  Its inputs are complete at this point of the current stream (an event is recorded
    here); it is issued by the next flush_side() - placed right behind the launch of a long, narrow kernel on the current
    stream, so that the two really run side by side - or by join_side()."""
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream())
    _side['deferred'].append((fn, ev, tensors))
    _side['pending'] = True
    if not _side.get('callback'):
        try:
            torch.autograd.Variable._execution_engine.queue_callback(join_side)
            _side['callback'] = True
        except RuntimeError:
            pass                      # not inside a backward pass: the caller joins explicitly


def begin_forward():
    """Called at the start of every model forward.  A backward pass that died with an exception leaves its deferred closures
    and the 'engine callback registered' mark behind; the next backward would then neither register its join nor run its own
    deferred parameter gradients (seen as wrong gradients of the first recurrent layer in the test AFTER a failing one).
    No backward is in flight when a forward starts, so whatever is still queued here is stale."""
    if _side['deferred'] or _side.get('callback'):
        _side['deferred'] = []
        _side['callback'] = False
        del _side['keep'][:]


def flush_side(after=None):
    """Issues the deferred work on the side stream.  `after`: an event of the current stream that the side stream waits for
    too - recorded just before a long narrow kernel is launched, it makes the deferred work start WITH that kernel
    instead of as soon as its own inputs are ready (when it would only compete with the wide kernels in between)."""
    todo, _side['deferred'] = _side['deferred'], []
    for fn, ev, tensors in todo:
        with on_side_stream(ev, *tensors) as side:
            if after is not None:
                side.wait_event(after)
            fn()


def join_side():
    """Runs what is still deferred IN LINE on the current stream (nothing is left to run beside it, and the side stream only
    owns part of the chip), then the current stream waits for everything on the side stream."""
    todo, _side['deferred'] = _side['deferred'], []
    cur = torch.cuda.current_stream()
    for fn, ev, tensors in todo:
        cur.wait_event(ev)
        fn()
    if _side['stream'] is not None and _side['pending']:
        torch.cuda.current_stream().wait_stream(_side['stream'])
    _side['pending'] = False
    _side['callback'] = False
    del _side['keep'][:]                 # the current stream is ordered behind every side-stream user of these tensors now


def ptr(t):
    """Device pointer of a CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError('HIP path got a non-CUDA tensor; there is no CPU fallback')
    if t.device.index != torch.cuda.current_device():
        raise RuntimeError('tensor on cuda:%d but the current device is cuda:%d: the HIP kernels launch on the current '
                           'device (call torch.cuda.set_device first)' % (t.device.index, torch.cuda.current_device()))
    return ctypes.c_void_p(t.data_ptr())


def check(rc, what):
    if rc != 0:
        raise RuntimeError('%s failed (%d): %s' % (what, rc, _lib.asr_last_error().decode()))


def call(name, *args):
    check(getattr(_lib, name)(*args), name)


def _f32c(t):
    assert t.dtype == torch.float32, t.dtype
    return t


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kc=1, b_kc=1, bias=None, act=ACT_NONE, accum=0, splits=1,
         batch=1, sA=0, sB=0, sC=0, seqT=0, bshift=0, prec=BF16):
    """Raw contraction on (views of) fp32 CUDA tensors; see include/asr_hip.h::asr_gemm."""
    call('asr_gemm', ptr(_f32c(A)), ptr(_f32c(B)), ptr(_f32c(C)), ptr(bias), M, N, K, lda, ldb, ldc,
         a_kc, b_kc, act, accum, splits, batch, sA, sB, sC, seqT, bshift, prec, stream_ptr())


def gemm16(A, B, C, M, N, K, lda, ldb, ldc, a_kc=1, b_kc=1, bias=None, act=ACT_NONE, accum=0, splits=1, perm_h=0,
           seqT=0, bshift=0, b_time_padded=0, a_off=0, b_off=0):
    """Contraction on bf16 CUDA tensors (see include/asr_hip.h::asr_gemm16); C bf16 (written) or fp32 (accumulated).
    a_off / b_off: element offsets into A / B (column blocks of a wider matrix)."""
    assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and C.dtype in (torch.bfloat16, torch.float32)
    pa = ctypes.c_void_p(ptr(A).value + 2 * a_off)
    pb = ctypes.c_void_p(ptr(B).value + 2 * b_off)
    call('asr_gemm16', pa, pb, ptr(C), ptr(bias), M, N, K, lda, ldb, ldc, a_kc, b_kc, act, accum, splits,
         1 if C.dtype == torch.bfloat16 else 0, perm_h, seqT, bshift, b_time_padded, stream_ptr())


def wgrad_splits(rows, out_rows=None, out_cols=None):
    """Number of reduction slices for a weight-gradient contraction over `rows` = B*T rows into an (out_rows, out_cols)
    matrix: enough 128x128 output tiles x slices to fill the chip ONCE at its residency of ~3.5 workgroups per CU (a
    second, partial round of workgroups costs as much as a full one: measured 247 us at 9 slices vs 296 us at 18 for
    the 2560x640 gradient), each slice at least 512 rows deep."""
    if out_rows is None:
        return max(1, min(32, rows // 1024))
    tiles = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    return max(1, min(64, rows // 512, int(round(900.0 / tiles))))


def linear_fwd(x2d, W, b, out2d, act=ACT_NONE, prec=BF16):
    """out = act(x W^T + b): x (M,K) contiguous rows, W (N,K)."""
    M, K = x2d.shape
    N = W.shape[0]
    gemm(x2d, W, out2d, M, N, K, x2d.stride(0), W.stride(0), out2d.stride(0), 1, 1, bias=b, act=act, prec=prec)


def linear_bwd(x2d, W, dy2d, dW, db, dx2d=None, prec=BF16, accum_dx=0):
    """dW += dy^T x ; db += colsum(dy) ; dx (=|+=) dy W."""
    M, K = x2d.shape
    N = W.shape[0]
    gemm(dy2d, x2d, dW, N, K, M, dy2d.stride(0), x2d.stride(0), dW.stride(0), 0, 0, accum=1, splits=wgrad_splits(M, N, K), prec=prec)
    if db is not None:
        call('asr_colsum', ptr(dy2d), dy2d.stride(0), M, N, ptr(db), stream_ptr())
    if dx2d is not None:
        gemm(dy2d, W, dx2d, M, K, N, dy2d.stride(0), W.stride(0), dx2d.stride(0), 1, 0, accum=accum_dx, prec=prec)


def linear_bwd_input(W, dy2d, dx2d, prec=BF16, accum_dx=0):
    """dx (=|+=) dy W alone (the parameter gradients of the layer are issued elsewhere)."""
    M, N = dy2d.shape
    K = W.shape[1]
    gemm(dy2d, W, dx2d, M, K, N, dy2d.stride(0), W.stride(0), dx2d.stride(0), 1, 0, accum=accum_dx, prec=prec)


def dec_weights_struct(tensors, nl):
    """tensors: dict with keys of _DEC_W_FIELDS + Wc, bc + lists Wih/Whh/bih/bhh."""
    w = DecWeights()
    for n in _DEC_W_FIELDS + ['Wc', 'bc']:
        setattr(w, n, tensors[n].data_ptr())
    for n in ('Wih', 'Whh', 'bih', 'bhh'):
        arr = getattr(w, n)
        for l in range(nl):
            arr[l] = tensors[n][l].data_ptr()
    return w


def dec_state_struct(tensors):
    s = DecState()
    for n, _ in DecState._fields_:
        if n == 'work_bytes':
            w = tensors.get('work')
            s.work_bytes = int(w.numel() * w.element_size()) if w is not None else 0
            continue
        t = tensors.get(n)
        setattr(s, n, t.data_ptr() if t is not None else None)
    return s
