"""Model variants outside the shipped configs: scaled-dot / multi-head / value-projected attention, GRU or
multi-layer decoders with dropout, embedding dropout, GRU encoder layers.

The shipped configs (LSTM, location-aware attention, one head, no decoder dropout) run on the persistent kernels
(src/functions.py::AttDecoderFn).  Everything else the reference's YAML surface accepts runs HERE: the reference's own step loop
(src/asr.py:131-170) with every arithmetic step a HIP kernel of csrc/variants.hip / the contraction kernels, chained by autograd.
torch contributes the tape, views, `cat` / `permute().contiguous()` copies and the accumulation of a gradient that has several
consumers - no arithmetic of the model.  A variant step is bound by its launches, like the reference's; it is a correctness path
(parity: tests/test_variants.py against the CPU restatement and the reference's fixtures), not the measured one.
"""
import torch
import torch.nn as nn

from src import hipabi as H


def _f32(shape, like):
    return torch.empty(shape, dtype=torch.float32, device=like.device)


# ---------------------------------------------------------------------------------------------------------------------
# a tensor that many steps read and whose gradient they accumulate IN PLACE (the attention key / value): the steps take the
# `token` as an input so that the holder's backward runs after all of theirs, and add into `grad` themselves
# ---------------------------------------------------------------------------------------------------------------------
class _Holder(object):
    def __init__(self, data):
        self.data = data.detach()
        self.grad = torch.zeros_like(self.data)


class HoldFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, box):
        box.append(_Holder(x.contiguous()))
        ctx.box = box
        return torch.zeros(1, dtype=torch.float32, device=x.device)

    @staticmethod
    def backward(ctx, dtoken):
        return ctx.box[0].grad.view_as(ctx.box[0].data), None


def hold(x):
    box = []
    token = HoldFn.apply(x, box)
    return box[0], token


class LinearActFn(torch.autograd.Function):
    """y = act(x W^T + b); parameter gradients are accumulated into W.grad / b.grad (flat storage); b may be None."""

    @staticmethod
    def forward(ctx, anchor, x, weight, bias, act, prec):
        x = x.contiguous()
        K = x.shape[-1]
        x2 = x.view(-1, K)
        N = weight.shape[0]
        out = _f32((x2.shape[0], N), x)
        H.linear_fwd(x2, weight, bias, out, act=act, prec=prec)
        ctx.weight, ctx.bias, ctx.act, ctx.prec = weight, bias, act, prec
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x2, out)
        return out.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dout):
        x2, out = ctx.saved_tensors
        weight, bias, prec = ctx.weight, ctx.bias, ctx.prec
        N = weight.shape[0]
        dy = dout.contiguous().view(-1, N)
        if ctx.act != H.ACT_NONE:
            dpre = torch.empty_like(dy)
            H.call('asr_act_bwd', H.ptr(dy), H.ptr(out), H.ptr(dpre), dy.numel(), ctx.act, H.stream_ptr())
            dy = dpre
        dx = torch.empty_like(x2) if ctx.need_dx else None
        H.linear_bwd(x2, weight, dy, weight.grad, bias.grad if bias is not None else None, dx, prec=prec)
        return None, (dx.view(*dout.shape[:-1], x2.shape[1]) if dx is not None else None), None, None, None, None


def linear(anchor, x, lin, prec, act=H.ACT_NONE):
    return LinearActFn.apply(anchor, x, lin.weight, lin.bias, act, prec)


class MaskedSoftmaxFn(torch.autograd.Function):
    """BaseAttention._attend's softmax (reference src/module.py:1110-1113): energy (R,T), R = B * NH rows ordered b * NH + head."""

    @staticmethod
    def forward(ctx, energy, enc_len, NH, temperature):
        energy = energy.contiguous()
        R, T = energy.shape
        attn = torch.empty_like(energy)
        H.call('asr_masked_softmax_fwd', H.ptr(energy), H.ptr(enc_len), R, NH, T, float(temperature), H.ptr(attn), H.stream_ptr())
        ctx.temperature = float(temperature)
        ctx.save_for_backward(attn)
        return attn

    @staticmethod
    def backward(ctx, dattn):
        (attn,) = ctx.saved_tensors
        R, T = attn.shape
        de = torch.empty_like(attn)
        H.call('asr_masked_softmax_bwd', H.ptr(attn), H.ptr(dattn.contiguous()), R, T, ctx.temperature, H.ptr(de), H.stream_ptr())
        return de, None, None, None


class DotEnergyFn(torch.autograd.Function):
    """ScaleDotAttention's energy (reference src/module.py:1127): energy[r,t] = q[r,:] . key[r,t,:]."""

    @staticmethod
    def forward(ctx, token, key, q, prec):
        q = q.contiguous()
        R, T, D = key.data.shape
        e = _f32((R, T), q)
        H.gemm(q, key.data, e, 1, T, D, D, D, T, 1, 1, batch=R, sA=D, sB=T * D, sC=T, prec=prec)
        ctx.key, ctx.prec = key, prec
        ctx.save_for_backward(q)
        return e

    @staticmethod
    def backward(ctx, de):
        (q,) = ctx.saved_tensors
        key, prec = ctx.key, ctx.prec
        R, T, D = key.data.shape
        de = de.contiguous()
        dq = torch.empty_like(q)
        H.gemm(de, key.data, dq, 1, D, T, T, D, D, 1, 0, batch=R, sA=T, sB=T * D, sC=D, prec=prec)
        H.gemm(de, q, key.grad, T, D, 1, T, D, D, 0, 0, accum=1, batch=R, sA=T, sB=D, sC=T * D, prec=prec)
        return None, None, dq, None


class LocConvFn(torch.autograd.Function):
    """loc_conv(prev_att).transpose(1,2) (reference src/module.py:1176): prev_att (B,NH,T) -> (B,T,Kn)."""

    @staticmethod
    def forward(ctx, anchor, prev_att, conv):
        prev_att = prev_att.contiguous()
        B, NH, T = prev_att.shape
        Kn, _, taps = conv.weight.shape
        out = _f32((B, T, Kn), prev_att)
        H.call('asr_loc_conv_fwd', H.ptr(prev_att), H.ptr(conv.weight), B, NH, T, Kn, (taps - 1) // 2, H.ptr(out), H.stream_ptr())
        ctx.conv = conv
        ctx.need_dprev = prev_att.requires_grad
        ctx.save_for_backward(prev_att)
        return out

    @staticmethod
    def backward(ctx, dout):
        (prev_att,) = ctx.saved_tensors
        conv = ctx.conv
        B, NH, T = prev_att.shape
        Kn, _, taps = conv.weight.shape
        dprev = torch.empty_like(prev_att) if ctx.need_dprev else None
        H.call('asr_loc_conv_bwd', H.ptr(dout.contiguous()), H.ptr(prev_att), H.ptr(conv.weight), B, NH, T, Kn, (taps - 1) // 2,
               H.ptr(dprev), H.ptr(conv.weight.grad), H.stream_ptr())
        return None, dprev, None


class LocEnergyFn(torch.autograd.Function):
    """gen_energy(tanh(k + q + tanh(loc_pre))) (reference src/module.py:1176-1182) for any number of heads."""

    @staticmethod
    def forward(ctx, anchor, token, key, q, loc_pre, gen):
        q, loc_pre = q.contiguous(), loc_pre.contiguous()
        R, T, D = key.data.shape
        B = loc_pre.shape[0]
        NH = R // B
        e = _f32((R, T), q)
        H.call('asr_loc_energy_fwd', H.ptr(key.data), H.ptr(q), H.ptr(loc_pre), H.ptr(gen.weight), H.ptr(gen.bias), B, NH, T, D, H.ptr(e),
               H.stream_ptr())
        ctx.key, ctx.gen, ctx.dims = key, gen, (B, NH, T, D)
        ctx.save_for_backward(q, loc_pre)
        return e

    @staticmethod
    def backward(ctx, de):
        q, loc_pre = ctx.saved_tensors
        key, gen = ctx.key, ctx.gen
        B, NH, T, D = ctx.dims
        dq, dloc = torch.empty_like(q), torch.empty_like(loc_pre)
        H.call('asr_loc_energy_bwd', H.ptr(key.data), H.ptr(q), H.ptr(loc_pre), H.ptr(gen.weight), H.ptr(de.contiguous()), B, NH, T, D,
               H.ptr(key.grad), H.ptr(dq), H.ptr(dloc), H.ptr(gen.weight.grad), H.ptr(gen.bias.grad), H.stream_ptr())
        return None, None, None, dq, dloc, None


class AttendFn(torch.autograd.Function):
    """context[r,:] = attn[r,:] . value[vi(r)] (reference src/module.py:1114).  value holds `Bv` utterances; row r reads value
    r % Bv: with one head or projected values Bv = R; with several heads and NO value projection the reference repeats the
    encoder output head-major (`value.repeat(num_head,1,1)`, src/asr.py:354) while its rows are batch-major - row r = b * NH + n
    attends over utterance r % B.  Reproduced as it is."""

    @staticmethod
    def forward(ctx, token, value, attn, prec):
        attn = attn.contiguous()
        R, T = attn.shape
        Bv, _, Dv = value.data.shape
        out = _f32((R, Dv), attn)
        for g in range(R // Bv):
            H.gemm(attn[g * Bv:], value.data, out[g * Bv:], 1, Dv, T, T, Dv, Dv, 1, 0, batch=Bv, sA=T, sB=T * Dv, sC=Dv, prec=prec)
        ctx.value, ctx.prec = value, prec
        ctx.save_for_backward(attn)
        return out

    @staticmethod
    def backward(ctx, dout):
        (attn,) = ctx.saved_tensors
        value, prec = ctx.value, ctx.prec
        R, T = attn.shape
        Bv, _, Dv = value.data.shape
        dout = dout.contiguous()
        dattn = torch.empty_like(attn)
        for g in range(R // Bv):
            H.gemm(dout[g * Bv:], value.data, dattn[g * Bv:], 1, T, Dv, Dv, Dv, T, 1, 1, batch=Bv, sA=Dv, sB=T * Dv, sC=T, prec=prec)
            H.gemm(attn[g * Bv:], dout[g * Bv:], value.grad, T, Dv, 1, T, Dv, Dv, 0, 0, accum=1, batch=Bv, sA=T, sB=Dv, sC=T * Dv, prec=prec)
        return None, None, dattn, None


class LSTMCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, gh, c_prev):
        gx, gh = gx.contiguous(), gh.contiguous()
        N, D4 = gx.shape
        D = D4 // 4
        act, h, c = torch.empty_like(gx), _f32((N, D), gx), _f32((N, D), gx)
        cp = c_prev.contiguous() if c_prev is not None else None
        H.call('asr_lstm_cell_fwd', H.ptr(gx), H.ptr(gh), H.ptr(cp), N, D, H.ptr(act), H.ptr(h), H.ptr(c), H.stream_ptr())
        ctx.has_cp = cp is not None
        ctx.save_for_backward(act, c, *([cp] if cp is not None else []))
        return h, c

    @staticmethod
    def backward(ctx, dh, dc):
        saved = ctx.saved_tensors
        act, c = saved[0], saved[1]
        cp = saved[2] if ctx.has_cp else None
        N, D4 = act.shape
        D = D4 // 4
        dg, dcp = torch.empty_like(act), torch.empty_like(c)
        H.call('asr_lstm_cell_bwd', H.ptr(act), H.ptr(cp), H.ptr(c), H.ptr(dh.contiguous() if dh is not None else None),
               H.ptr(dc.contiguous() if dc is not None else None), N, D, H.ptr(dg), H.ptr(dcp), H.stream_ptr())
        return dg, dg, (dcp if ctx.has_cp else None)


class GRUCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, gh, h_prev):
        gi, gh = gi.contiguous(), gh.contiguous()
        N, D3 = gi.shape
        D = D3 // 3
        saved, h = _f32((N, 4 * D), gi), _f32((N, D), gi)
        hp = h_prev.contiguous() if h_prev is not None else None
        H.call('asr_gru_cell_fwd', H.ptr(gi), H.ptr(gh), H.ptr(hp), N, D, H.ptr(saved), H.ptr(h), H.stream_ptr())
        ctx.has_hp = hp is not None
        ctx.save_for_backward(saved, *([hp] if hp is not None else []))
        return h

    @staticmethod
    def backward(ctx, dh):
        sv = ctx.saved_tensors
        saved = sv[0]
        hp = sv[1] if ctx.has_hp else None
        N, D4 = saved.shape
        D = D4 // 4
        dgi, dgh, dhp = _f32((N, 3 * D), saved), _f32((N, 3 * D), saved), _f32((N, D), saved)
        H.call('asr_gru_cell_bwd', H.ptr(saved), H.ptr(hp), H.ptr(dh.contiguous()), N, D, H.ptr(dgi), H.ptr(dgh), H.ptr(dhp), H.stream_ptr())
        return dgi, dgh, (dhp if ctx.has_hp else None)


def dropout(x, p, seed):
    """nn.Dropout on (N, D) with the counter-based generator of the encoder layers (mask = f(seed, flat index))."""
    from src.functions import DropoutFn
    return DropoutFn.apply(x.unsqueeze(1), p, seed).squeeze(1)


# ---------------------------------------------------------------------------------------------------------------------
# parameter containers with the reference's names
# ---------------------------------------------------------------------------------------------------------------------
class RNNParams(nn.Module):
    """nn.LSTM / nn.GRU parameter names (weight_ih_l0[_reverse], ...); `gates` = 4 (LSTM) or 3 (GRU)."""

    def __init__(self, module, input_dim, dim, bidirectional, num_layers=1):
        super().__init__()
        self.module, self.input_dim, self.dim, self.bidirectional, self.num_layers = module.upper(), input_dim, dim, bidirectional, num_layers
        self.gates = {'LSTM': 4, 'GRU': 3}[self.module]
        for l in range(num_layers):
            din = input_dim if l == 0 else dim * (2 if bidirectional else 1)
            for sfx in ([''] + (['_reverse'] if bidirectional else [])):
                self.register_parameter('weight_ih_l%d%s' % (l, sfx), nn.Parameter(torch.empty(self.gates * dim, din)))
                self.register_parameter('weight_hh_l%d%s' % (l, sfx), nn.Parameter(torch.empty(self.gates * dim, dim)))
                self.register_parameter('bias_ih_l%d%s' % (l, sfx), nn.Parameter(torch.empty(self.gates * dim)))
                self.register_parameter('bias_hh_l%d%s' % (l, sfx), nn.Parameter(torch.empty(self.gates * dim)))


class ScaleDotAttention(nn.Module):
    """Parameter-free (reference src/module.py:1121-1132)."""

    def __init__(self, temperature, num_head):
        super().__init__()
        self.temperature, self.num_head = temperature, num_head

    def reset_mem(self):
        pass

    def set_mem(self, prev_att):
        pass


# ---------------------------------------------------------------------------------------------------------------------
# the step loop (reference ASR.forward src/asr.py:124-175, Attention.forward :331-364, Decoder.forward :262-270)
# ---------------------------------------------------------------------------------------------------------------------
def variant_decoder(model, anchor, enc, enc_len, decode_step, teacher, tf_rate, ctx):
    from src.functions import EmbeddingFn
    att, dec, prec = model.attention, model.decoder, model.prec
    B, T, _ = enc.shape
    NH, adim = att.num_head, att.dim
    train = model.training
    dev = enc.device
    enc_len = enc_len.to(dev).contiguous()
    # ---- memory of the attention (computed once, reference :340-355)
    key = LinearActFn.apply(anchor, enc, att.proj_k.weight, att.proj_k.bias, H.ACT_TANH, prec)            # (B,T,NH*adim)
    if att.v_proj:
        value = LinearActFn.apply(anchor, enc, att.proj_v.weight, att.proj_v.bias, H.ACT_TANH, prec)      # (B,T,NH*v_dim)
    else:
        value = enc
    if NH > 1:
        key = key.view(B, T, NH, adim).permute(0, 2, 1, 3).contiguous().view(B * NH, T, adim)
        if att.v_proj:
            value = value.view(B, T, NH, att.v_dim).permute(0, 2, 1, 3).contiguous().view(B * NH, T, att.v_dim)
    key_h, key_tok = hold(key)
    val_h, val_tok = hold(value)
    prev_att = None
    if att.mode == 'loc':
        import numpy as np
        pa = np.zeros((B, NH, T), dtype=np.float32)                      # uniform over the valid frames (reference :1169-1173), built on the host
        for i, sl in enumerate(enc_len.cpu().tolist()):
            pa[i, :, :sl] = 1.0 / sl
        prev_att = torch.from_numpy(pa).to(dev)
    # ---- decoder state (reference Decoder.init_state :228-236)
    NL, dim = dec.layer, dec.dim
    hs = [torch.zeros((B, dim), dtype=torch.float32, device=dev) for _ in range(NL)]
    cs = [None] * NL
    is_lstm = dec.layers.module == 'LSTM'
    emb_p = float(getattr(model, 'emb_drop', 0.0)) if train else 0.0
    dec_p = float(dec.dropout) if train else 0.0
    tokens0 = torch.zeros((B,), dtype=torch.int64, device=dev)
    last_char = EmbeddingFn.apply(anchor, tokens0, model.pre_embed)
    teacher_emb = None
    if teacher is not None:
        teacher_emb = EmbeddingFn.apply(anchor, teacher.contiguous(), model.pre_embed)                     # (B,L,dim)
        if emb_p > 0:
            from src.functions import DropoutFn
            teacher_emb = DropoutFn.apply(teacher_emb, emb_p, ctx.next_seed())
    outputs, att_seq, states = [], [], []
    for t in range(int(decode_step)):
        # ---- attend (the query is the state of all layers, :251-257)
        query = hs[0] if NL == 1 else torch.cat(hs, dim=-1)
        q = LinearActFn.apply(anchor, query, att.proj_q.weight, att.proj_q.bias, H.ACT_TANH, prec).view(B * NH, adim)
        if att.mode == 'dot':
            energy = DotEnergyFn.apply(key_tok, key_h, q, prec)
        else:
            loc = LocConvFn.apply(anchor, prev_att, att.att_layer.loc_conv)
            loc_pre = LinearActFn.apply(anchor, loc, att.att_layer.loc_proj.weight, None, H.ACT_NONE, prec)
            energy = LocEnergyFn.apply(anchor, key_tok, key_h, q, loc_pre, att.att_layer.gen_energy)
        attn = MaskedSoftmaxFn.apply(energy, enc_len, NH, att.att_layer.temperature)                         # (B*NH,T)
        context = AttendFn.apply(val_tok, val_h, attn, prec)                                               # (B*NH,Dv)
        attn = attn.view(B, NH, T)
        if att.mode == 'loc':
            prev_att = attn
        if NH > 1:
            context = LinearActFn.apply(anchor, context.view(B, NH * att.v_dim), att.merge_head.weight, att.merge_head.bias, H.ACT_NONE, prec)
        # ---- decode (:262-270): nn.LSTM / nn.GRU over one time step, dropout between the layers, final dropout before char_trans
        x = torch.cat([last_char, context], dim=-1)
        for l in range(NL):
            P = dec.layers
            gx = LinearActFn.apply(anchor, x, getattr(P, 'weight_ih_l%d' % l), getattr(P, 'bias_ih_l%d' % l), H.ACT_NONE, prec)
            gh = LinearActFn.apply(anchor, hs[l], getattr(P, 'weight_hh_l%d' % l), getattr(P, 'bias_hh_l%d' % l), H.ACT_NONE, prec)
            if is_lstm:
                hs[l], cs[l] = LSTMCellFn.apply(gx, gh, cs[l])
            else:
                hs[l] = GRUCellFn.apply(gx, gh, hs[l])
            x = hs[l]
            if l + 1 < NL and dec_p > 0:
                x = dropout(x, dec_p, ctx.next_seed())
        d_state = x
        xo = dropout(x, dec_p, ctx.next_seed()) if dec_p > 0 else x
        cur_char = LinearActFn.apply(anchor, xo, dec.char_trans.weight, dec.char_trans.bias, H.ACT_NONE, prec)
        # ---- next input (:145-166)
        if teacher is not None:
            if tf_rate == 1 or torch.rand(1).item() <= tf_rate:
                last_char = teacher_emb[:, t, :]
            else:
                with torch.no_grad():
                    sampled = torch.empty((B,), dtype=torch.int64, device=dev)
                    model._ss_counter = getattr(model, '_ss_counter', 0) + 1
                    seed = (model.seed * 7919 + model._ss_counter) & 0xFFFFFFFFFFFF
                    H.call('asr_sample_tokens', H.ptr(cur_char.detach()), cur_char.shape[1], H.ptr(sampled), 1, B, cur_char.shape[1], seed,
                           H.stream_ptr())
                last_char = EmbeddingFn.apply(anchor, sampled, model.pre_embed)
                if emb_p > 0:
                    last_char = dropout(last_char, emb_p, ctx.next_seed())
        else:
            with torch.no_grad():
                cand = torch.empty((B, 1), dtype=torch.int32, device=dev)
                H.call('asr_beam_candidates', H.ptr(cur_char.detach().contiguous()), H.ptr(cand), B, cur_char.shape[1], 1, H.stream_ptr())
            last_char = EmbeddingFn.apply(anchor, cand.view(B).long(), model.pre_embed)
        outputs.append(cur_char)
        att_seq.append(attn)
        states.append(d_state)
    att_output = torch.stack(outputs, dim=1)          # (B,L,V)
    att_seq = torch.stack(att_seq, dim=2)             # (B,NH,L,T)
    return att_output, att_seq, torch.stack(states, dim=1)


# ---------------------------------------------------------------------------------------------------------------------
# GRU encoder layer: nn.GRU(bidirectional) -> [LayerNorm] -> dropout -> time down-sampling -> tanh(Linear)
# (reference RNNLayer src/module.py:1003-1081 with module = GRU)
# ---------------------------------------------------------------------------------------------------------------------
class GRUSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, layer, prec):
        x = x.contiguous()
        B, T, Din = x.shape
        Hd, ND = layer.dim, layer.nd
        G = ND * 3 * Hd
        st = H.stream_ptr()
        gi = _f32((B, T, ND, 3 * Hd), x)
        H.gemm(x, layer.w_ih_cat, gi, B * T, G, Din, Din, Din, G, 1, 1, bias=layer.b_ih_cat, prec=prec)
        whhT = _f32((ND, Hd, 3 * Hd), x)
        H.call('asr_permute_last2', H.ptr(layer.w_hh_cat), H.ptr(whhT), ND, 3 * Hd, Hd, st)
        y = _f32((B, T, ND * Hd), x)
        saved = _f32((B, T, ND, 4 * Hd), x)
        H.call('asr_gru_fwd', H.ptr(gi), H.ptr(whhT), H.ptr(layer.b_hh_cat), B, T, Hd, ND, H.ptr(y), H.ptr(saved), st)
        ctx.layer, ctx.prec = layer, prec
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, y, saved)
        return y

    @staticmethod
    def backward(ctx, dy):
        layer, prec = ctx.layer, ctx.prec
        x, y, saved = ctx.saved_tensors
        B, T, Din = x.shape
        Hd, ND = layer.dim, layer.nd
        G, D = ND * 3 * Hd, ND * Hd
        st = H.stream_ptr()
        dgi, dgh = _f32((B, T, ND, 3 * Hd), x), _f32((B, T, ND, 3 * Hd), x)
        H.call('asr_gru_bwd', H.ptr(dy.contiguous()), H.ptr(y), H.ptr(saved), H.ptr(layer.w_hh_cat), B, T, Hd, ND, H.ptr(dgi), H.ptr(dgh), st)
        gi2, gh2, x2, y2 = dgi.view(B * T, G), dgh.view(B * T, G), x.view(B * T, Din), y.view(B * T, D)
        H.gemm(gi2, x2, layer.g_w_ih_cat, G, Din, B * T, G, Din, Din, 0, 0, accum=1, splits=H.wgrad_splits(B * T, G, Din), prec=prec)
        H.call('asr_colsum', H.ptr(gi2), G, B * T, G, H.ptr(layer.g_b_ih_cat), st)
        H.call('asr_colsum', H.ptr(gh2), G, B * T, G, H.ptr(layer.g_b_hh_cat), st)
        for d in range(ND):
            H.gemm(gh2[:, d * 3 * Hd:], y2[:, d * Hd:], layer.g_w_hh_cat[d], 3 * Hd, Hd, B * T, G, D, Hd, 0, 0, accum=1,
                   splits=H.wgrad_splits(B * T, 3 * Hd, Hd), seqT=T, bshift=(-1 if d == 0 else 1), prec=prec)
        dx = None
        if ctx.need_dx:
            dx = _f32((B, T, Din), x)
            H.gemm(gi2, layer.w_ih_cat, dx, B * T, Din, G, G, Din, Din, 1, 0, prec=prec)
        return None, dx, None, None


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, y, ln):
        y = y.contiguous()
        B, T, D = y.shape
        yn, stats = torch.empty_like(y), _f32((B * T, 2), y)
        H.call('asr_layernorm_fwd', H.ptr(y), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(yn), H.ptr(stats), B * T, D, 1e-5, 0, H.stream_ptr())
        ctx.ln = ln
        ctx.save_for_backward(y, stats)
        return yn

    @staticmethod
    def backward(ctx, dyn):
        y, stats = ctx.saved_tensors
        ln = ctx.ln
        B, T, D = y.shape
        dy = torch.empty_like(y)
        H.call('asr_layernorm_bwd', H.ptr(dyn.contiguous()), H.ptr(y), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(stats), H.ptr(dy),
               H.ptr(ln.weight.grad), H.ptr(ln.bias.grad), B * T, D, 0, H.stream_ptr())
        return None, dy, None


class DropDownFn(torch.autograd.Function):
    """dropout then time down-sampling ('drop' keeps every rate-th frame, 'concat' stacks `rate` frames) - reference :1059-1076."""

    @staticmethod
    def forward(ctx, y, p, seed, rate, style):
        y = y.contiguous()
        B, T, D = y.shape
        if rate == 1:
            T2, Dz = T, D
        elif style == 0:
            T2, Dz = (T + rate - 1) // rate, D
        else:
            T2, Dz = T // rate, D * rate
        z = _f32((B, T2, Dz), y)
        H.call('asr_dropout_downsample_fwd', H.ptr(y), H.ptr(z), B, T, D, T2, rate, style, float(p), int(seed), H.stream_ptr())
        ctx.meta = (B, T, D, T2, rate, style, float(p), int(seed))
        return z

    @staticmethod
    def backward(ctx, dz):
        B, T, D, T2, rate, style, p, seed = ctx.meta
        dy = _f32((B, T, D), dz)
        H.call('asr_dropout_downsample_bwd', H.ptr(dz.contiguous()), H.ptr(dy), B, T, D, T2, rate, style, p, seed, H.stream_ptr())
        return dy, None, None, None, None


def gru_layer_forward(layer, x, ctx, train, seed):
    from src.functions import to_f32_fn
    x = to_f32_fn(x)
    y = GRUSeqFn.apply(ctx.anchor, x, layer, ctx.prec)
    if layer.layer_norm:
        y = LayerNormFn.apply(ctx.anchor, y, layer.ln)
    p = float(layer.dropout) if train else 0.0
    if p > 0 or layer.sample_rate > 1:
        y = DropDownFn.apply(y, p, seed, layer.sample_rate, 0 if layer.sample_style == 'drop' else 1)
    if layer.proj:
        y = LinearActFn.apply(ctx.anchor, y, layer.pj.weight, layer.pj.bias, H.ACT_TANH, ctx.prec)
    return y
