"""Autograd bookkeeping around the HIP kernels (libasr_hip.so).

Each torch.autograd.Function below covers one coarse stage of the reference graph and calls the C ABI
for all arithmetic.  Parameter gradients are accumulated by the kernels directly into the model's flat
gradient buffer (the views held in `param.grad`); the Functions therefore return gradients only for
activations.  `anchor` is a dummy requires-grad tensor that keeps autograd calling `backward` even when
the acoustic features themselves need no gradient.
"""
import ctypes
import weakref

import torch

from src import hipabi as H


def _empty(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# --------------------------------------------------------------------------------------------------
# encoder RNN layer: BiLSTM -> [LayerNorm] -> dropout -> time down-sampling -> tanh(Linear)
# (reference RNNLayer.forward, src/module.py:1040-1081)
# --------------------------------------------------------------------------------------------------
class RNNLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, layer, train, seed, prec):
        x = x.contiguous()
        B, T, Din = x.shape
        Hd, ND = layer.dim, layer.nd
        G = ND * 4 * Hd
        D = ND * Hd
        st = H.stream_ptr()
        gates = _empty((B, T, ND, 4 * Hd), x)
        H.gemm(x, layer.w_ih_cat, gates, B * T, G, Din, Din, Din, G, 1, 1, bias=layer.b_ih_cat, prec=prec)
        y = _empty((B, T, D), x)
        c = _empty((B, T, ND, Hd), x)
        nbytes = H.lib().asr_lstm_workspace_bytes(B, Hd, ND)
        ws = H.handoff_acquire(nbytes, x.device)          # pool area for this one launch, back to the pool once collected
        H.call('asr_lstm_fwd', H.ptr(gates), H.ptr(layer.w_hh_cat), H.ptr(layer.b_hh_cat), H.ptr(y), H.ptr(c),
               B, T, Hd, ND, prec, H.ptr(ws), nbytes, st)
        layer.last_ws = ws
        H.watch_abort(ws, release=True)
        yn, stats = y, None
        if layer.layer_norm:
            yn = _empty((B, T, D), x)
            stats = _empty((B * T, 2), x)
            H.call('asr_layernorm_fwd', H.ptr(y), H.ptr(layer.ln.weight), H.ptr(layer.ln.bias), H.ptr(yn), H.ptr(stats),
                   B * T, D, 1e-5, 0, st)
        p = float(layer.dropout) if train else 0.0
        r, style = layer.sample_rate, (0 if layer.sample_style == 'drop' else 1)
        if r == 1:
            T2, Dz = T, D
        elif style == 0:
            T2, Dz = (T + r - 1) // r, D
        else:
            T2, Dz = T // r, D * r
        alias = (r == 1 and p == 0.0)
        if alias:
            z = yn
        else:
            z = _empty((B, T2, Dz), x)
            H.call('asr_dropout_downsample_fwd', H.ptr(yn), H.ptr(z), B, T, D, T2, r, style, p, seed, st)
        if layer.proj:
            out = _empty((B, T2, Dz), x)
            H.linear_fwd(z.view(B * T2, Dz), layer.pj.weight, layer.pj.bias, out.view(B * T2, Dz), act=H.ACT_TANH, prec=prec)
        else:
            out = z
        ctx.layer, ctx.prec, ctx.meta = layer, prec, (B, T, Din, T2, Dz, p, seed, alias, style)
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, gates, c, y, z, out, *( [stats] if stats is not None else []))
        return out

    @staticmethod
    def backward(ctx, dout):
        layer, prec = ctx.layer, ctx.prec
        B, T, Din, T2, Dz, p, seed, alias, style = ctx.meta
        saved = ctx.saved_tensors
        x, gates, c, y, z, out = saved[:6]
        stats = saved[6] if len(saved) > 6 else None
        Hd, ND = layer.dim, layer.nd
        G, D = ND * 4 * Hd, ND * Hd
        st = H.stream_ptr()
        dout = dout.contiguous()
        if layer.dp is not None:
            # every consumer of the encoder output has finished its backward: the heads/decoder bucket(s) can go - once whatever
            # was deferred to the side stream (the decoder's parameter gradients under ASR_OVERLAP_DP) has been issued and joined
            H.join_side()
            for i in range(layer.bucket - 0):
                layer.dp.bucket_ready(i)
        if layer.proj:
            dpre = _empty((B * T2, Dz), x)
            H.call('asr_act_bwd', H.ptr(dout), H.ptr(out), H.ptr(dpre), B * T2 * Dz, H.ACT_TANH, st)
            dz = _empty((B, T2, Dz), x)
            H.linear_bwd(z.view(B * T2, Dz), layer.pj.weight, dpre, layer.pj.weight.grad, layer.pj.bias.grad,
                         dz.view(B * T2, Dz), prec=prec)
        else:
            dz = dout
        if alias:
            dyn = dz
        else:
            dyn = _empty((B, T, D), x)
            H.call('asr_dropout_downsample_bwd', H.ptr(dz), H.ptr(dyn), B, T, D, T2, layer.sample_rate, style, p, seed, st)
        if layer.layer_norm:
            dy = _empty((B, T, D), x)
            H.call('asr_layernorm_bwd', H.ptr(dyn), H.ptr(y), H.ptr(layer.ln.weight), H.ptr(layer.ln.bias), H.ptr(stats),
                   H.ptr(dy), H.ptr(layer.ln.weight.grad), H.ptr(layer.ln.bias.grad), B * T, D, 0, st)
        else:
            dy = dyn
        nbytes = H.lib().asr_lstm_workspace_bytes(B, Hd, ND)
        ws = H.handoff_acquire(nbytes, x.device)
        pre = torch.cuda.Event()
        pre.record(torch.cuda.current_stream())
        H.call('asr_lstm_bwd', H.ptr(gates), H.ptr(layer.w_hh_cat), H.ptr(dy), H.ptr(c), B, T, Hd, ND, prec,
               H.ptr(ws), nbytes, st)
        layer.last_ws_bwd = ws
        H.watch_abort(ws, release=True)
        H.flush_side(after=pre)       # the upper layer's parameter gradients run beside this recurrence (40 workgroups)
        # gates now holds the gradient wrt the gate pre-activations
        g2 = gates.view(B * T, G)
        x2 = x.view(B * T, Din)
        y2 = y.view(B * T, D)
        splits_ih = H.wgrad_splits(B * T, G, Din)
        splits_hh = H.wgrad_splits(B * T, 4 * Hd, Hd)

        def weight_grads():
            s_ = H.stream_ptr()
            H.gemm(g2, x2, layer.g_w_ih_cat, G, Din, B * T, G, Din, Din, 0, 0, accum=1, splits=splits_ih, prec=prec)
            H.call('asr_colsum2', H.ptr(g2), G, B * T, G, H.ptr(layer.g_b_ih_cat), H.ptr(layer.g_b_hh_cat), s_)
            for d in range(ND):
                H.gemm(g2[:, d * 4 * Hd:], y2[:, d * Hd:], layer.g_w_hh_cat[d], 4 * Hd, Hd, B * T, G, D, Hd, 0, 0,
                       accum=1, splits=splits_hh, seqT=T, bshift=(-1 if d == 0 else 1), prec=prec)

        # the input gradient continues the chain; the parameter gradients of this layer are needed only by the optimizer
        # and run beside the next layer's recurrence on the side stream (single-process runs; under data parallelism they
        # stay in order so that the bucket's all-reduce can start right behind them)
        use_side = H.side_enabled() and layer.dp is None
        if use_side:
            H.defer_side(weight_grads, gates, x, y)      # issued behind the NEXT layer's recurrence launch (flush_side below)
        dx = None
        if ctx.need_dx:
            dx = _empty((B, T, Din), x)
            H.gemm(g2, layer.w_ih_cat, dx, B * T, Din, G, G, Din, Din, 1, 0, prec=prec)
        if not use_side:
            weight_grads()
        if layer.dp is not None:
            layer.dp.bucket_ready(layer.bucket)
        return None, dx, None, None, None, None


# --------------------------------------------------------------------------------------------------
# the same layer on bf16 storage (bf16 contraction mode, the encoder's working point): gate-minor bf16 gates,
# time-padded bf16 h, batch-sliced persistent recurrence (csrc/lstm_persist3.hip), bf16 contraction operands
# --------------------------------------------------------------------------------------------------
def _empty16(shape, like):
    return torch.empty(shape, dtype=torch.bfloat16, device=like.device)


def to_bf16(x):
    """fp32 -> bf16 copy through asr_cast_bf16 (bf16 tensors pass through)."""
    if x.dtype == torch.bfloat16:
        return x.contiguous()
    x = x.contiguous()
    out = _empty16(x.shape, x)
    H.call('asr_cast_bf16', H.ptr(x), H.ptr(out), x.numel(), H.stream_ptr())
    return out


def to_f32(x):
    if x.dtype == torch.float32:
        return x.contiguous()
    x = x.contiguous()
    out = _empty(x.shape, x)
    H.call('asr_cast_f32', H.ptr(x), H.ptr(out), x.numel(), 0, H.stream_ptr())
    return out


class CastF32Fn(torch.autograd.Function):
    """bf16 encoder output -> fp32 for the CTC head and the decoder; the gradient goes back as bf16."""
    @staticmethod
    def forward(ctx, x):
        return to_f32(x)

    @staticmethod
    def backward(ctx, g):
        return to_bf16(g)


def to_f32_fn(x):
    """Differentiable fp32 view of a layer input (no-op for fp32 tensors)."""
    return CastF32Fn.apply(x) if x.dtype == torch.bfloat16 else x


def rnn_fast_ok(layer, x, prec):
    """The bf16-storage path covers: bf16 contractions, LSTM cell, H % 16 == 0 <= 512, no LayerNorm, 'drop' down-sampling
    (or none), input width a multiple of 8, B <= 16 * (8 / directions)."""
    if prec != H.BF16 or layer.layer_norm or not H.fast16_enabled():
        return False
    if layer.sample_rate > 1 and layer.sample_style != 'drop':
        return False
    B, T, Din = x.shape
    if Din % 8 != 0 or (layer.nd * layer.dim) % 8 != 0:
        return False
    return int(H.lib().asr_lstm16_workspace_bytes(B, layer.dim, layer.nd, 0)) > 0


def _ws16(layer, B, bwd):
    """Persistent, zero-initialised workspace of the recurrence per (layer, batch, pass) + its launch counter."""
    key = (B, bwd)
    cache = layer.__dict__.setdefault('_ws16_cache', {})
    dev = layer.w_hh_cat.device
    if key not in cache or cache[key][0].device != dev:
        n = int(H.lib().asr_lstm16_workspace_bytes(B, layer.dim, layer.nd, bwd))
        ws = H.handoff_acquire(n, dev)                    # pool area: scrubbed from every XCD, never returned to the allocator
        weakref.finalize(layer, H.handoff_release, ws)    # the layer's areas go back to the POOL when the layer dies
        cache[key] = [ws, 0]
    ent = cache[key]
    ent[1] += 1
    return ent[0], ent[1]


def prepack16(layers, B, prec):
    """The bf16 weight copies of the later encoder layers, made on the side stream beside the first layer's projection and
    recurrence instead of in front of their own (4 x 13 us of small launches on the step's critical path).  Called once per
    forward by the Encoder, before the first layer; a marker left behind by a forward that never reached its layer is
    dropped here."""
    for l in layers:
        l.__dict__.pop('_pack16_ev', None)
    if not (H.overlap_enabled() and prec == H.BF16 and H.fast16_enabled() and torch.is_grad_enabled()):
        return
    if any(getattr(l, 'dp', None) is not None for l in layers) and not H.overlap_dp_enabled():
        return                # data parallel: no CU-masked stream beside RCCL's kernels until that has been run
    todo = [l for l in layers[1:] if not l.layer_norm and (l.sample_rate == 1 or l.sample_style == 'drop')
            and l.w_ih_cat.shape[1] % 8 == 0 and (l.nd * l.dim) % 8 == 0
            and int(H.lib().asr_lstm16_workspace_bytes(B, l.dim, l.nd, 0)) > 0]
    if not todo:
        return
    for l in todo:
        if l.__dict__.get('_pack16') is None:
            return                                        # first step: the copies are allocated on the layers' own stream
    with H.on_side_stream(None):
        for l in todo:
            _packed16(l, side=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            l.__dict__['_pack16_ev'] = ev


def _packed16(layer, side=False):
    """bf16 operand copies of the layer's contraction weights, rebuilt from the fp32 master by ONE kernel per call."""
    Hd, ND = layer.dim, layer.nd
    G, Din, D = ND * 4 * Hd, layer.w_ih_cat.shape[1], ND * Hd
    dev = layer.w_ih_cat.device
    pk = layer.__dict__.get('_pack16')
    if pk is None or pk['wih'].device != dev:
        b16 = lambda *s_: torch.empty(s_, dtype=torch.bfloat16, device=dev)
        pk = {'wih': b16(G, Din), 'wihT': b16(Din, G), 'bias': torch.empty(G, dtype=torch.float32, device=dev),
              'pj': b16(D, D) if layer.proj else None, 'pjT': b16(D, D) if layer.proj else None}
        layer.__dict__['_pack16'] = pk
    ev = layer.__dict__.pop('_pack16_ev', None)
    if ev is not None and not side:
        torch.cuda.current_stream().wait_event(ev)       # packed ahead on the side stream (prepack16) in this forward
        return pk
    H.call('asr_rnn_pack_weights', H.ptr(layer.w_ih_cat), H.ptr(layer.b_ih_cat), H.ptr(layer.b_hh_cat),
           H.ptr(layer.pj.weight) if layer.proj else None, H.ptr(pk['wih']), H.ptr(pk['wihT']), H.ptr(pk['bias']),
           H.ptr(pk['pj']), H.ptr(pk['pjT']), Hd, ND, Din, D, H.stream_ptr())
    return pk


class RNNLayerFastFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, layer, train, seed):
        in_dtype = x.dtype
        x16 = to_bf16(x)
        B, T, Din = x16.shape
        Hd, ND = layer.dim, layer.nd
        G, D = ND * 4 * Hd, ND * Hd
        st = H.stream_ptr()
        pk = _packed16(layer)
        gates = _empty16((B, T, ND, Hd, 4), x16)
        H.gemm16(x16, pk['wih'], gates, B * T, G, Din, Din, Din, G, 1, 1, bias=pk['bias'])
        y = _empty16((B, T + 2, D), x16)          # rows 0 and T+1 (time pads) are zeroed by the recurrence kernel
        c = _empty((B, T, ND, Hd), x16)
        ws, epoch = _ws16(layer, B, 0)
        # two status blocks by launch parity (include/asr_hip.h): this launch reports in block epoch & 1 and clears the other one
        H.abort_guard(ws, ((epoch + 1) & 1) * 1024)
        reserved = 64 if (layer.dp is not None and layer.dp.world > 1) else 0
        H.call('asr_lstm16_fwd', H.ptr(gates), H.ptr(layer.w_hh_cat), H.ptr(y), H.ptr(c), B, T, Hd, ND,
               H.ptr(ws), ws.numel(), epoch, reserved, st)
        H.watch_abort(ws, (epoch & 1) * 1024)
        layer.last_ws = ws
        p = float(layer.dropout) if train else 0.0
        r = layer.sample_rate
        T2 = T if r == 1 else (T + r - 1) // r
        z = _empty16((B, T2, D), x16)
        H.call('asr_dropout_downsample16_fwd', H.ptr(y), (T + 2) * D, D, H.ptr(z), B, T, D, T2, r, 0, p, seed, st)
        if layer.proj:
            out = _empty16((B, T2, D), x16)
            H.gemm16(z, pk['pj'], out, B * T2, D, D, D, D, D, 1, 1, bias=layer.pj.bias, act=H.ACT_TANH)
        else:
            out = z
        ctx.layer, ctx.meta = layer, (B, T, Din, T2, p, seed, reserved, in_dtype)
        ctx.need_dx = x.requires_grad
        ctx.pk = pk
        ctx.save_for_backward(x16, gates, c, y, z, out)
        if H.DEBUG_KEEP:
            layer._dbg = {'x': x16, 'gates': gates, 'c': c, 'y': y, 'z': z, 'out': out}
        return out

    @staticmethod
    def backward(ctx, dout):
        layer, pk = ctx.layer, ctx.pk
        B, T, Din, T2, p, seed, reserved, in_dtype = ctx.meta
        x16, gates, c, y, z, out = ctx.saved_tensors
        Hd, ND = layer.dim, layer.nd
        G, D = ND * 4 * Hd, ND * Hd
        st = H.stream_ptr()
        dout = to_bf16(dout)
        # Parameter gradients are off the critical path (only the optimizer reads them): they are deferred to the CU-masked side
        # stream and start together with the NEXT recurrence of the backward pass, which runs on the complementary CU mask
        # (H.on_rec_stream) - 160 workgroups that leave 96 compute units idle.  Under data parallelism a bucket may be signalled
        # only when everything that writes into it has been issued: in line (default) the signals stay where they were; with
        # ASR_OVERLAP_DP=1 they are deferred too, FIFO behind the work they depend on, and fire on the side stream.
        dp = layer.dp
        overlap = H.overlap_enabled() and (dp is None or H.overlap_dp_enabled())
        if dp is not None:
            def earlier_buckets():          # every consumer of this layer's output has finished its backward: heads / decoder / upper layers
                for i in range(layer.bucket):
                    dp.bucket_ready(i)
            if overlap:
                H.defer_side(earlier_buckets)
            else:
                earlier_buckets()
        if layer.proj:
            dpre = _empty16((B * T2, D), x16)
            H.call('asr_act_bwd16', H.ptr(dout), H.ptr(out), H.ptr(dpre), B * T2 * D, H.ACT_TANH, st)

            def pj_grads():
                H.gemm16(dpre, z, layer.pj.weight.grad, D, D, B * T2, D, D, D, 0, 0, accum=1, splits=H.wgrad_splits(B * T2, D, D))
                H.call('asr_colsum16', H.ptr(dpre), D, B * T2, D, H.ptr(layer.pj.bias.grad), None, 0, H.stream_ptr())
            if overlap:
                H.defer_side(pj_grads, dpre, z)
            else:
                pj_grads()
            dz = _empty16((B, T2, D), x16)
            H.gemm16(dpre, pk['pjT'], dz, B * T2, D, D, D, D, D, 1, 1)
        else:
            dz = dout
        dy = _empty16((B, T, D), x16)
        H.call('asr_dropout_downsample16_bwd', H.ptr(dz), H.ptr(dy), B, T, D, T2, layer.sample_rate, 0, p, seed, st)
        ws, epoch = _ws16(layer, B, 1)
        H.abort_guard(ws, ((epoch + 1) & 1) * 1024)
        if overlap:
            pre = torch.cuda.Event()
            pre.record(torch.cuda.current_stream())
            with H.on_rec_stream():
                H.call('asr_lstm16_bwd', H.ptr(gates), H.ptr(layer.w_hh_cat), H.ptr(dy), H.ptr(c), B, T, Hd, ND,
                       H.ptr(ws), ws.numel(), epoch, 256 - 8 * H.REC_UNITS, H.stream_ptr())
            H.flush_side(after=pre)       # the deferred gradients (layer above, this layer's projection) start with this recurrence
        else:
            H.call('asr_lstm16_bwd', H.ptr(gates), H.ptr(layer.w_hh_cat), H.ptr(dy), H.ptr(c), B, T, Hd, ND,
                   H.ptr(ws), ws.numel(), epoch, reserved, st)
        H.watch_abort(ws, (epoch & 1) * 1024)
        layer.last_ws_bwd = ws
        # gates now holds the gradient wrt the gate pre-activations (gate-minor); parameter gradients in reference row order
        dx = None
        if ctx.need_dx:
            dx = _empty16((B, T, Din), x16)
            H.gemm16(gates, pk['wihT'], dx, B * T, Din, G, G, G, Din, 1, 1)
            if in_dtype == torch.float32:
                dx = to_f32(dx)

        def weight_grads():
            H.gemm16(gates, x16, layer.g_w_ih_cat, G, Din, B * T, G, Din, Din, 0, 0, accum=1,
                     splits=H.wgrad_splits(B * T, G, Din), perm_h=Hd)
            H.call('asr_colsum16', H.ptr(gates), G, B * T, G, H.ptr(layer.g_b_ih_cat), H.ptr(layer.g_b_hh_cat), Hd, H.stream_ptr())
            splits_hh = H.wgrad_splits(B * T, 4 * Hd, Hd)
            for d in range(ND):
                H.gemm16(gates, y, layer.g_w_hh_cat[d], 4 * Hd, Hd, B * T, G, D, Hd, 0, 0, accum=1, splits=splits_hh,
                         perm_h=Hd, seqT=T, bshift=(-1 if d == 0 else 1), b_time_padded=1, a_off=d * 4 * Hd, b_off=d * Hd)
        if overlap:
            H.defer_side(weight_grads, gates, x16, y)
            if dp is not None:
                H.defer_side(lambda: dp.bucket_ready(layer.bucket))
        else:
            weight_grads()
            if dp is not None:
                dp.bucket_ready(layer.bucket)
        return None, dx, None, None, None


# --------------------------------------------------------------------------------------------------
# CTC head: log_softmax(ReLU(Linear(enc)))   (src/asr.py:29-32,116-120)
# --------------------------------------------------------------------------------------------------
class CTCHeadFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, enc, lin, prec, get_logit):
        enc = enc.contiguous()
        B, T, E = enc.shape
        V = lin.weight.shape[0]
        act = _empty((B, T, V), enc)
        H.linear_fwd(enc.view(B * T, E), lin.weight, lin.bias, act.view(B * T, V), act=H.ACT_RELU, prec=prec)
        ctx.lin, ctx.prec, ctx.get_logit = lin, prec, get_logit
        if get_logit:
            ctx.save_for_backward(enc, act)
            return act
        logp = _empty((B, T, V), enc)
        H.call('asr_log_softmax', H.ptr(act), H.ptr(logp), B * T, V, H.stream_ptr())
        ctx.save_for_backward(enc, act, logp)
        return logp

    @staticmethod
    def backward(ctx, g):
        lin, prec = ctx.lin, ctx.prec
        g = g.contiguous()
        if ctx.get_logit:
            enc, act = ctx.saved_tensors
            B, T, E = enc.shape
            V = act.shape[-1]
            dpre = _empty((B * T, V), enc)
            H.call('asr_act_bwd', H.ptr(g), H.ptr(act), H.ptr(dpre), B * T * V, H.ACT_RELU, H.stream_ptr())
        else:
            enc, act, logp = ctx.saved_tensors
            B, T, E = enc.shape
            V = act.shape[-1]
            dpre = _empty((B * T, V), enc)
            H.call('asr_logsoftmax_relu_bwd', H.ptr(g), H.ptr(logp), H.ptr(act), H.ptr(dpre), B * T, V, H.stream_ptr())
        denc = _empty((B, T, E), enc)
        if getattr(lin, '_asr_defer', False) and H.overlap_enabled():
            # input gradient now; the head's own gradients (a split reduction over B*T rows + a column sum) go to the side
            # stream beside the encoder's BPTT, like the decoder's (AttDecoderFn.backward)
            H.linear_bwd_input(lin.weight, dpre, denc.view(B * T, E), prec=prec)
            x2d = enc.view(B * T, E)
            H.defer_side(lambda: H.linear_bwd(x2d, lin.weight, dpre, lin.weight.grad, lin.bias.grad, None, prec=prec), x2d, dpre)
        else:
            H.linear_bwd(enc.view(B * T, E), lin.weight, dpre, lin.weight.grad, lin.bias.grad, denc.view(B * T, E), prec=prec)
        return None, denc, None, None, None


def scale_by_device_scalar(x, alpha):
    """x * alpha with alpha a one-element device tensor (the grad_output of a loss): asr_scale_dev."""
    x = x.contiguous()
    out = torch.empty_like(x)
    a = alpha.reshape(1).to(torch.float32).contiguous()
    H.call('asr_scale_dev', H.ptr(x), H.ptr(out), x.numel(), H.ptr(a), H.stream_ptr())
    return out


_LOSS_W = {}


def loss_weight(value, device):
    """A cached one-element device tensor holding a host constant (loss weights)."""
    key = (float(value), str(device))
    if key not in _LOSS_W:
        _LOSS_W[key] = torch.full((1,), float(value), dtype=torch.float32, device=device)
    return _LOSS_W[key]


class LossMixFn(torch.autograd.Function):
    """total = wa a + wb b (bin/train_asr.py:238,246) with the weights as one-element device tensors (under data parallelism the
    attention weight comes out of an all-reduce); forward and backward are the one-thread kernel asr_loss_mix."""

    @staticmethod
    def forward(ctx, a, wa, b, wb):
        out = torch.empty((), dtype=torch.float32, device=a.device)
        H.call('asr_loss_mix', H.ptr(a), H.ptr(wa), H.ptr(b), H.ptr(wb) if b is not None else None, H.ptr(out), H.stream_ptr())
        ctx.has_b = b is not None
        ctx.save_for_backward(wa, *( [wb] if b is not None else []))
        return out

    @staticmethod
    def backward(ctx, g):
        sv = ctx.saved_tensors
        g = g.reshape(1).to(torch.float32).contiguous()
        ga = torch.empty((), dtype=torch.float32, device=g.device)
        H.call('asr_loss_mix', H.ptr(g), H.ptr(sv[0]), None, None, H.ptr(ga), H.stream_ptr())
        gb = None
        if ctx.has_b:
            gb = torch.empty((), dtype=torch.float32, device=g.device)
            H.call('asr_loss_mix', H.ptr(g), H.ptr(sv[1]), None, None, H.ptr(gb), H.stream_ptr())
        return ga, None, gb, None


# --------------------------------------------------------------------------------------------------
# CTC loss (torch.nn.CTCLoss(blank=0, zero_infinity=False), bin/train_asr.py:135,237)
# --------------------------------------------------------------------------------------------------
class CTCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp_btv, targets, input_len, target_len):
        logp = logp_btv.contiguous()
        B, T, V = logp.shape
        targets = targets.contiguous()
        L = targets.shape[1]
        if 2 * L + 1 > 1024:
            # the kernel holds the 2L+1 lattice states of an utterance in one workgroup; what counts is the longest TARGET, not
            # the padded width of the batch (one device read-back, on this rare path only)
            L = max(1, int(target_len.max()))
            targets = targets[:, :L].contiguous()
        dev = logp.device
        nll = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        grad = _empty((B, T, V), logp)
        nbytes = H.lib().asr_ctc_loss_workspace_bytes(B, T, L)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        # locals keep the converted tensors alive until the kernel has been enqueued on this stream
        targets = targets.to(dev, torch.int64)
        il = input_len.to(dev, torch.int64).contiguous()
        tl = target_len.to(dev, torch.int64).contiguous()
        H.call('asr_ctc_loss', H.ptr(logp), H.ptr(targets), H.ptr(il), H.ptr(tl), H.ptr(nll), H.ptr(loss), H.ptr(grad),
               B, T, V, L, 1.0, H.ptr(ws), nbytes, H.stream_ptr())
        ctx.save_for_backward(grad)
        ctx.nll = nll
        return loss

    @staticmethod
    def backward(ctx, gout):
        (grad,) = ctx.saved_tensors
        return scale_by_device_scalar(grad, gout), None, None, None


# --------------------------------------------------------------------------------------------------
# sequence loss on decoder logits (CrossEntropyLoss(ignore_index=0) / LabelSmoothingLoss)
# --------------------------------------------------------------------------------------------------
class SeqLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits_rv, target_r, mode, classes, smoothing):
        logits = logits_rv.contiguous()
        R, V = logits.shape
        tgt = target_r.to(logits_rv.device, torch.int64).contiguous()
        dev = logits.device
        dl = _empty((R, V), logits)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        acc = torch.empty(4, dtype=torch.float32, device=dev)
        H.call('asr_xent', H.ptr(logits), H.ptr(tgt), R, H.ptr(dl), H.ptr(loss), H.ptr(acc), 1, R, V, mode, classes,
               smoothing, 1.0, H.stream_ptr())
        ctx.save_for_backward(dl)
        return loss

    @staticmethod
    def backward(ctx, gout):
        (dl,) = ctx.saved_tensors
        return scale_by_device_scalar(dl, gout), None, None, None, None


# --------------------------------------------------------------------------------------------------
# attention decoder loop (src/asr.py:123-175)
# --------------------------------------------------------------------------------------------------
def _dec_dims(model, B, Tp, L):
    att, dec = model.attention, model.decoder
    d = H.DecDims()
    d.B, d.Tp, d.E = B, Tp, model.encoder.out_dim
    d.A, d.Q = att.dim, dec.dim * dec.layer
    d.Dd, d.NL, d.V = dec.dim, dec.layer, model.vocab_size
    d.Kn, d.Ks = att.att_layer.kernel_num, att.att_layer.kernel_size
    d.L = L
    d.temperature = float(att.att_layer.temperature)
    return d


def _dec_tensors(model, grads):
    att, dec, al = model.attention, model.decoder, model.attention.att_layer
    g = (lambda p: p.grad) if grads else (lambda p: p)
    nl = dec.layer
    return {
        'Wq': g(att.proj_q.weight), 'bq': g(att.proj_q.bias), 'Wk': g(att.proj_k.weight), 'bk': g(att.proj_k.bias),
        'Wconv': g(al.loc_conv.weight), 'Wproj': g(al.loc_proj.weight), 'wg': g(al.gen_energy.weight), 'bg': g(al.gen_energy.bias),
        'emb': g(model.pre_embed.weight),
        'Wih': [g(getattr(dec.layers, 'weight_ih_l%d' % l)) for l in range(nl)],
        'Whh': [g(getattr(dec.layers, 'weight_hh_l%d' % l)) for l in range(nl)],
        'bih': [g(getattr(dec.layers, 'bias_ih_l%d' % l)) for l in range(nl)],
        'bhh': [g(getattr(dec.layers, 'bias_hh_l%d' % l)) for l in range(nl)],
        'Wc': g(dec.char_trans.weight), 'bc': g(dec.char_trans.bias),
    }


def _dec_state(d, dev, save_conv=True, half_copies=False):
    """save_conv: keep the location-convolution output of every step (B,L,Kn,Tp) for the backward pass.
    half_copies: bf16 working copies of key and enc for the step kernels (bf16 contraction mode, E % 4 == 0)."""
    f = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    h = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)
    return {
        'conv': f(d.B, d.L, d.Kn, d.Tp) if save_conv else None,
        'key16': h(d.B, d.Tp, d.A) if half_copies else None,
        'enc16': h(d.B, d.Tp, d.E) if half_copies else None,
        'key': f(d.B, d.Tp, d.A), 'att': f(d.B, d.L, d.Tp), 'q': f(d.B, d.L, d.A), 'xin': f(d.B, d.L, d.Dd + d.E),
        'gates': f(d.B, d.L, d.NL, 4 * d.Dd), 'cs': f(d.B, d.L, d.NL, d.Dd), 'hs': f(d.B, d.L, d.NL, d.Dd),
        'logits': f(d.B, d.L, d.V), 'energy': f(d.B, d.Tp),
        'tokens': torch.empty((d.B, d.L), dtype=torch.int64, device=dev),
    }


def att_decoder_forward_sampled(model, enc, enc_len, L, teacher, prec, tf_rate, decisions=None):
    """Scheduled sampling (reference src/asr.py:145-158): after every step ONE uniform draw decides for the whole batch
    whether the next input token is the teacher's (probability tf_rate) or a sample from softmax(logits) of this step.
    The loop runs through the per-step C-ABI call; the token table records what was fed, so the backward pass (tokens
    are data) is the ordinary one.  decisions: optional list of booleans (True = teacher) replacing the host draws (tests)."""
    B, Tp, _ = enc.shape
    d = _dec_dims(model, B, Tp, L)
    st = _dec_state(d, enc.device, save_conv=True, half_copies=False)
    st['tokens'].zero_()
    w = H.dec_weights_struct(_dec_tensors(model, False), d.NL)
    s = H.dec_state_struct(st)
    sp = H.stream_ptr()
    teacher = teacher.contiguous()
    H.call('asr_att_decoder_keys', ctypes.byref(d), ctypes.byref(w), H.ptr(enc), H.ptr(st['key']), prec, sp)
    model._ss_counter = getattr(model, '_ss_counter', 0)
    st['tf_decisions'] = []
    for t in range(L):
        H.call('asr_att_decoder_step', ctypes.byref(d), ctypes.byref(w), H.ptr(enc), H.ptr(enc_len), ctypes.byref(s), t, prec, sp)
        if t + 1 < L:
            use_teacher = bool(decisions[t]) if decisions is not None else (tf_rate == 1 or torch.rand(1).item() <= tf_rate)
            st['tf_decisions'].append(use_teacher)
            if use_teacher:
                st['tokens'][:, t + 1] = teacher[:, t]
            else:
                model._ss_counter += 1
                seed = (model.seed * 7919 + model._ss_counter) & 0xFFFFFFFFFFFF
                H.call('asr_sample_tokens', H.ptr(st['logits'][:, t]), st['logits'].stride(0), H.ptr(st['tokens'][:, t + 1]),
                       st['tokens'].stride(0), B, d.V, seed, sp)
    return d, st


_DEC_WS = {}          # (kind, decoder dims, device) -> uint8 tensor; process-wide, LRU of 32 shapes per pass


def _dec_workspace(model, kind, key, nbytes, device):
    """Workspaces of the persistent decoder launches live for the whole PROCESS, one per (pass, decoder dimensions, shape):
    fixed address ranges instead of recycled allocator blocks.  On a recycled block an exchange granule of ANY earlier launch
    of the process can sit at a polled address with a matching 6-bit tag (4-bit launch epoch + step sequence); in an XCD's L2
    it survives the host memset, and the producers' own clear covers only clusters that land on the same XCD again.  With a
    fixed range and the per-range launch epoch of csrc/decoder_persist.hip (`next_epoch`) the only older tags a slot can hold are
    those of the previous launch on that range, whose epoch differs by one.  Keyed by shape, not by model: tests build and drop
    many models of the same shape, which is exactly the recycling this avoids.  Concurrent use by two models of the same shape
    cannot happen on one stream: a launch has finished with the range when the next one starts, except the backward's
    parameter half on the side stream - joined before the next backward is issued (src/step.py, engine callback)."""
    dd = _dec_dims(model, *key)
    k = (kind, tuple(getattr(dd, f) for f, _ in dd._fields_), str(device))
    ws = _DEC_WS.pop(k, None)
    if ws is not None and ws.numel() < nbytes:
        H.handoff_release(ws)
        ws = None
    if ws is None:
        while len(_DEC_WS) >= 64:                  # variable-length training: keep the 32 most recent shapes (x 2 passes)
            H.handoff_release(_DEC_WS.pop(next(iter(_DEC_WS))))      # evicted areas go back to the hand-off POOL, not to the allocator
        ws = H.handoff_acquire(nbytes, device)     # zeroed + scrubbed from every XCD whenever an area changes hands
    _DEC_WS[k] = ws                                # most recently used last
    return ws


def att_decoder_forward(model, enc, enc_len, L, teacher, prec):
    """Runs the decode loop; returns (dims, state dict)."""
    B, Tp, _ = enc.shape
    d = _dec_dims(model, B, Tp, L)
    st = _dec_state(d, enc.device, half_copies=(prec == H.BF16 and d.E % 4 == 0))
    if prec == H.BF16 and teacher is not None:
        nwork = int(H.lib().asr_att_decoder_fwd_work_bytes(ctypes.byref(d)))     # 0: no single-launch plan for this shape
        if nwork:
            st['work'] = _dec_workspace(model, 'fwd', (d.B, d.Tp, d.L), nwork, enc.device)
            H.abort_guard(st['work'])
    w = H.dec_weights_struct(_dec_tensors(model, False), d.NL)
    s = H.dec_state_struct(st)
    t_ptr, t_ld = (None, 0)
    if teacher is not None:
        teacher = teacher.contiguous()
        assert teacher.shape[1] >= L - 1
        t_ptr, t_ld = H.ptr(teacher), teacher.shape[1]
    H.call('asr_att_decoder_fwd', ctypes.byref(d), ctypes.byref(w), H.ptr(enc), H.ptr(enc_len), t_ptr, t_ld,
           ctypes.byref(s), prec, H.stream_ptr())
    H.watch_abort(st.get('work'))
    return d, st


class AttDecoderFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, enc, enc_len, teacher, L, model, prec, tf_rate=1.0):
        enc = enc.contiguous()
        enc_len = enc_len.to(enc.device, torch.int64).contiguous()
        if teacher is not None and tf_rate != 1:
            d, st = att_decoder_forward_sampled(model, enc, enc_len, L, teacher, prec, tf_rate, getattr(model, '_tf_decisions', None))
            model._last_tokens = st['tokens']
        else:
            d, st = att_decoder_forward(model, enc, enc_len, L, teacher, prec)
        ctx.model, ctx.prec, ctx.d, ctx.st = model, prec, d, st
        ctx.save_for_backward(enc, enc_len)
        att_seq = st['att'].view(d.B, 1, d.L, d.Tp)
        ctx.mark_non_differentiable(att_seq)
        return st['logits'], att_seq, st['hs']

    @staticmethod
    def backward(ctx, dlogits, _datt, _dhs):
        model, prec, d, st = ctx.model, ctx.prec, ctx.d, ctx.st
        enc, enc_len = ctx.saved_tensors
        dlogits = dlogits.contiguous()
        denc = torch.zeros_like(enc)
        w = H.dec_weights_struct(_dec_tensors(model, False), d.NL)
        g = H.dec_weights_struct(_dec_tensors(model, True), d.NL)
        s = H.dec_state_struct(st)
        nbytes = H.lib().asr_att_decoder_bwd_workspace_bytes(ctypes.byref(d))
        ws = _dec_workspace(model, 'bwd', (d.B, d.Tp, d.L), nbytes, enc.device)
        model._last_dec_bwd_ws = ws          # kept for diagnostics (tools/diag_dec.py)
        persistent = int(H.lib().asr_att_decoder_bwd_persistent_tiles(ctypes.byref(d))) > 0
        status_off = int(H.lib().asr_att_decoder_bwd_status_offset(ctypes.byref(d))) if persistent else 0
        if persistent:
            H.abort_guard(ws, status_off)
        overlap = H.overlap_enabled() and (getattr(model, '_dp', None) is None or H.overlap_dp_enabled())
        looped = ctypes.c_int(0)
        H.call('asr_att_decoder_bwd_ex', ctypes.byref(d), ctypes.byref(w), ctypes.byref(g), H.ptr(enc), H.ptr(enc_len),
               ctypes.byref(s), H.ptr(dlogits), H.ptr(denc), H.ptr(ws), nbytes, prec, 1 if overlap else 0, ctypes.byref(looped),
               H.stream_ptr())
        if persistent:
            H.watch_abort(ws, status_off)
        if overlap:
            # the decoder's parameter gradients (15 launches, ~0.7 ms) are off the path to the encoder gradient: they run on the
            # CU-masked side stream beside the encoder's BPTT (issued at its first recurrence, RNNLayerFastFn.backward)
            keep = [t for t in st.values() if torch.is_tensor(t)]

            def param_grads():
                H.call('asr_att_decoder_bwd_params', ctypes.byref(d), ctypes.byref(w), ctypes.byref(g), H.ptr(enc), H.ptr(enc_len),
                       ctypes.byref(s), H.ptr(dlogits), H.ptr(ws), nbytes, looped.value, prec, H.stream_ptr())
            H.defer_side(param_grads, enc, enc_len, dlogits, ws, *keep)
        ctx.st = None
        return None, denc, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------------
# RNN language model, training forward/backward (reference src/lm.py:27-38): embedding -> dropout -> LSTM stack -> dropout ->
# output projection.  The LSTM layers go through RNNLayerFn (one direction, dropout on each layer's output: nn.LSTM's
# inter-layer dropout plus the model's dp2 on the last layer).
# --------------------------------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, tokens, emb):
        tok = tokens.reshape(-1).contiguous()
        V, D = emb.weight.shape
        out = torch.empty((tok.numel(), D), dtype=torch.float32, device=tok.device)
        H.call('asr_gather_rows', H.ptr(emb.weight), H.ptr(tok), H.ptr(out), tok.numel(), D, D, D, V, H.stream_ptr())
        ctx.emb = emb
        ctx.save_for_backward(tok)
        return out.view(*tokens.shape, D)

    @staticmethod
    def backward(ctx, dout):
        (tok,) = ctx.saved_tensors
        emb = ctx.emb
        V, D = emb.weight.shape
        dout = dout.contiguous().view(-1, D)
        H.call('asr_embedding_bwd', H.ptr(dout), D, H.ptr(tok), H.ptr(emb.weight.grad), tok.numel(), D, V, H.stream_ptr())
        return None, None, None


class DropoutFn(torch.autograd.Function):
    """nn.Dropout on a (B,T,D) tensor with the counter-based generator of the encoder layers (mask = f(seed, index))."""

    @staticmethod
    def forward(ctx, x, p, seed):
        x = x.contiguous()
        B, T, D = x.shape
        z = torch.empty_like(x)
        H.call('asr_dropout_downsample_fwd', H.ptr(x), H.ptr(z), B, T, D, T, 1, 0, float(p), int(seed), H.stream_ptr())
        ctx.meta = (B, T, D, float(p), int(seed))
        return z

    @staticmethod
    def backward(ctx, dz):
        B, T, D, p, seed = ctx.meta
        dz = dz.contiguous()
        dx = torch.empty_like(dz)
        H.call('asr_dropout_downsample_bwd', H.ptr(dz), H.ptr(dx), B, T, D, T, 1, 0, p, seed, H.stream_ptr())
        return dx, None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with the parameter gradients accumulated into W.grad / b.grad (flat storage); b may be None."""

    @staticmethod
    def forward(ctx, anchor, x, weight, bias, prec):
        x = x.contiguous()
        K = x.shape[-1]
        x2 = x.view(-1, K)
        N = weight.shape[0]
        out = torch.empty((x2.shape[0], N), dtype=torch.float32, device=x.device)
        H.linear_fwd(x2, weight, bias, out, prec=prec)
        ctx.weight, ctx.bias, ctx.prec = weight, bias, prec
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x2)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dout):
        (x2,) = ctx.saved_tensors
        weight, bias, prec = ctx.weight, ctx.bias, ctx.prec
        N = weight.shape[0]
        dy = dout.contiguous().view(-1, N)
        dx = torch.empty_like(x2) if ctx.need_dx else None
        H.linear_bwd(x2, weight, dy, weight.grad, bias.grad if bias is not None else None, dx, prec=prec)
        return None, (dx.view(*dout.shape[:-1], x2.shape[1]) if dx is not None else None), None, None, None
