"""Joint CTC-attention ASR model on the MI355X HIP path.

Drop-in for the reference's `src/asr.ASR` (constructor kwargs = the YAML `model:` block, same
`forward` signature and return tuple, same attribute and state_dict names: reference src/asr.py:13-177),
but every tensor operation on the path runs in libasr_hip.so.  The nn.Modules only hold parameters, which
all live in ONE flat fp32 buffer (plus one flat gradient buffer = the data-parallel all-reduce bucket).
"""
import math

import torch
import torch.nn as nn

from src import functions as F_hip
from src import hipabi as H
from src.module import RNNLayer, Downsampler, LocationAwareAttention, LSTMParams
from src.util import init_weights_, init_gate_

ALIGN = 64  # floats; every parameter group starts on a 256-byte boundary of the flat buffer


class _RunCtx(object):
    """Per-forward bookkeeping handed to the layers: autograd anchor, precision, dropout seeds."""

    def __init__(self, model):
        self.anchor = model._anchor
        self.prec = model.prec
        self._model = model

    def next_seed(self):
        self._model._drop_counter += 1
        return (self._model.seed * 1000003 + self._model._drop_counter) & 0xFFFFFFFFFFFF


class Encoder(nn.Module):
    """Listener: optional VGG / down-sampler front-end + stacked (pyramidal) BiLSTM layers
    (reference src/asr.py:390-476)."""

    def __init__(self, input_size, batch_size, vgg, vgg_freq, vgg_low_filt, module, bidirection, dim, dropout,
                 layer_norm, proj, sample_rate, sample_style):
        super().__init__()
        self.vgg, self.vgg_freq, self.vgg_low_filt = vgg, vgg_freq, vgg_low_filt
        self.sample_rate = 1
        assert len(sample_rate) == len(dropout), 'Number of layer mismatch'
        assert len(dropout) == len(dim), 'Number of layer mismatch'
        layers = []
        input_dim = input_size
        if vgg > 0:
            if vgg == 1:
                from src.vgg import VGGExtractor
                ext = VGGExtractor(input_size)
            elif vgg == 2:
                from src.vgg import FreqVGGExtractor
                ext = FreqVGGExtractor(input_size, vgg_freq, vgg_low_filt)
            elif vgg == 3:
                from src.vgg import VGGExtractor2
                ext = VGGExtractor2(input_size)
            elif vgg == 4:
                from src.vgg import FreqVGGExtractor2
                ext = FreqVGGExtractor2(input_size, vgg_freq, vgg_low_filt)
            elif vgg == 5:
                from src.vgg import VGGExtractor_LN
                ext = VGGExtractor_LN(input_size)
            elif vgg == 6:
                ext = Downsampler(input_size)
            elif vgg == 7:
                from src.module import Featemb_Extractor
                ext = Featemb_Extractor(input_size)
            else:
                raise NotImplementedError('vgg = {} is not available on the HIP path'.format(vgg))
            layers.append(ext)
            input_dim = ext.out_dim
            self.sample_rate = 1 if vgg == 7 else self.sample_rate * (4 if (vgg < 3 or vgg == 6) else 2)
        if module not in ('LSTM', 'GRU'):
            raise NotImplementedError('encoder module %s is not available on the HIP path' % module)
        for l in range(len(dim)):
            layers.append(RNNLayer(input_dim, module, dim[l], bidirection, dropout[l], layer_norm[l], sample_rate[l],
                                   sample_style, proj[l], batch_size))
            input_dim = layers[-1].out_dim
            self.sample_rate = self.sample_rate * sample_rate[l]
        self.in_dim, self.out_dim = input_size, input_dim
        self.layers = nn.ModuleList(layers)

    def forward(self, input_x, enc_len, ctx=None):
        if ctx is not None and input_x.is_cuda:
            H.begin_forward()
            # CU split between the recurrence stream and the side stream follows the widest recurrent layer of THIS model
            # (eval-mode forwards are followed by backward passes too: tests, gradient checks)
            H.configure_rec_units([m.dim for m in self.layers if isinstance(m, RNNLayer) and m.module == 'LSTM'] or [320])
        if self.training and ctx is not None and input_x.is_cuda:
            F_hip.prepack16([m for m in self.layers if isinstance(m, RNNLayer) and m.module == 'LSTM'], input_x.shape[0], ctx.prec)
        for layer in self.layers:
            input_x, enc_len = layer(input_x, enc_len, ctx)
        return input_x, enc_len


class Decoder(nn.Module):
    """Speller parameters: `layers` (LSTM weights), `char_trans` (reference src/asr.py:183-270)."""

    def __init__(self, batch_size, input_dim, vocab_size, module, dim, layer, dropout):
        super().__init__()
        if module not in ('LSTM', 'GRU'):
            raise NotImplementedError('decoder module %s is not available on the HIP path' % module)
        from src.variants import RNNParams
        self.in_dim, self.layer, self.dim, self.dropout = input_dim, layer, dim, dropout
        self.enable_cell = module == 'LSTM'
        self.layers = RNNParams(module, input_dim, dim, False, num_layers=layer)
        # the persistent / per-step decoder kernels cover the shipped shape of this module; the rest runs in src/variants.py
        self.fast = module == 'LSTM' and dropout == 0 and layer <= H.MAX_DEC_LAYERS
        self.char_trans = nn.Linear(dim, vocab_size)
        self.hidden_state = None


class Attention(nn.Module):
    """Attention parameters: proj_q, proj_k, att_layer (reference src/asr.py:273-364)."""

    def __init__(self, v_dim, q_dim, mode, dim, num_head, temperature, v_proj, loc_kernel_size, loc_kernel_num):
        super().__init__()
        self.v_dim, self.dim, self.mode, self.num_head, self.v_proj = v_dim, dim, mode.lower(), num_head, v_proj
        self.proj_q = nn.Linear(q_dim, dim * num_head)
        self.proj_k = nn.Linear(v_dim, dim * num_head)
        if v_proj:
            self.proj_v = nn.Linear(v_dim, v_dim * num_head)
        if self.mode == 'dot':
            from src.variants import ScaleDotAttention
            self.att_layer = ScaleDotAttention(temperature, num_head)
        elif self.mode == 'loc':
            self.att_layer = LocationAwareAttention(loc_kernel_size, loc_kernel_num, dim, num_head, temperature)
        else:
            raise NotImplementedError
        if num_head > 1:
            self.merge_head = nn.Linear(v_dim * num_head, v_dim)
        self.fast = self.mode == 'loc' and num_head == 1 and not v_proj


class ASR(nn.Module):
    ''' ASR model, including Encoder/Decoder(s) — HIP implementation '''

    def __init__(self, input_size, vocab_size, batch_size, ctc_weight, encoder, attention=None, decoder=None, emb_drop=0.0,
                 init_adadelta=True, prec='bf16', seed=0):
        super().__init__()
        assert 0 <= ctc_weight <= 1
        self.emb_drop = float(emb_drop)
        self.vocab_size = vocab_size
        self.ctc_weight = ctc_weight
        self.enable_ctc = ctc_weight > 0
        self.enable_att = ctc_weight != 1
        self.lm = None
        self.prec = H.BF16 if str(prec).lower() in ('bf16', '1') else H.F32
        self.seed = int(seed)
        self._drop_counter = 0

        self.encoder = Encoder(input_size, batch_size, **encoder)
        if self.enable_ctc:
            self.ctc_layer = nn.Sequential(nn.Linear(self.encoder.out_dim, vocab_size), nn.ReLU())
        if self.enable_att:
            self.dec_dim = decoder['dim']
            self.pre_embed = nn.Embedding(vocab_size, self.dec_dim)
            self.decoder = Decoder(batch_size, self.encoder.out_dim + self.dec_dim, vocab_size, **decoder)
            self.attention = Attention(self.encoder.out_dim, self.dec_dim * self.decoder.layer, **attention)

        # same end state as the reference's init (src/asr.py:44-50, src/util.py:60-88)
        init_weights_(self)
        if self.enable_att:
            for l in range(self.decoder.layer):
                init_gate_(getattr(self.decoder.layers, 'bias_ih_l{}'.format(l)))

        self.flat_param = None
        self.flat_grad = None
        self._anchor = None
        self._flatten()

    # ---- flat parameter storage ---------------------------------------------------------------------
    def _param_groups_for_flat(self):
        """Layout of the flat buffer: every RNN layer's parameters back to back (one bucket per layer),
        then the front-end (VGG) parameters, then everything else (heads, embedding, decoder, attention)."""
        sections, seen = [], set()
        rnn_layers = [m for m in self.encoder.layers if isinstance(m, RNNLayer)]
        for m in rnn_layers:
            g = m.flat_groups()
            sections.append(('rnn', g))
            seen.update(id(p) for grp in g for p in grp)
        front = [m for m in self.encoder.layers if not isinstance(m, RNNLayer)]
        fg = [[p] for m in front for p in m.parameters()]
        seen.update(id(p) for grp in fg for p in grp)
        sections.append(('front', fg))
        sections.append(('rest', [[p] for p in self.parameters() if id(p) not in seen]))
        return sections

    def _flatten(self):
        params = list(self.parameters())
        if not params:
            return
        dev = params[0].device
        sections = self._param_groups_for_flat()
        offs, off, ranges = {}, 0, []
        for kind, groups in sections:
            start = (off + ALIGN - 1) // ALIGN * ALIGN
            for g in groups:
                off = (off + ALIGN - 1) // ALIGN * ALIGN
                for p in g:
                    offs[id(p)] = off
                    off += p.numel()
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            ranges.append((kind, start, off))
        total = off
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        grad = torch.zeros(total, dtype=torch.float32, device=dev)
        for p in params:
            o, n = offs[id(p)], p.numel()
            flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = flat[o:o + n].view(p.shape)
            p.grad = grad[o:o + n].view(p.shape)
            p._asr_flat = (flat, grad, offs)
        self.flat_param, self.flat_grad, self._offsets, self._ranges = flat, grad, offs, ranges
        self._anchor = torch.zeros(1, dtype=torch.float32, device=dev, requires_grad=True)

        def view_of(ps, shape, is_grad):
            o = offs[id(ps[0])]
            n = sum(p.numel() for p in ps)
            return (grad if is_grad else flat)[o:o + n].view(shape)
        for m in self.modules():
            if hasattr(m, 'bind_flat'):
                m.bind_flat(view_of)

    def attach_data_parallel(self, group=None, reducer_cls=None):
        """Creates the flat-bucket gradient reducer (src/dist.py) with buckets in backward-completion order:
        heads/decoder/attention first, then the encoder RNN layers top-down, the front-end last.
        reducer_cls: a FlatDataParallel subclass (tests/test_dp_hooks.py records the hook order with one)."""
        from src.dist import FlatDataParallel
        if reducer_cls is not None:
            FlatDataParallel = reducer_cls
        rnn = [(s, e) for k, s, e in self._ranges if k == 'rnn']
        front = [(s, e) for k, s, e in self._ranges if k == 'front' and e > s]
        rest = [(s, e) for k, s, e in self._ranges if k == 'rest' and e > s]
        buckets = rest + rnn[::-1] + front
        dp = FlatDataParallel(self.flat_param, self.flat_grad, buckets, group)
        layers = [m for m in self.encoder.layers if isinstance(m, RNNLayer)]
        for i, m in enumerate(layers[::-1]):
            m.dp, m.bucket = dp, len(rest) + i
        for m in self.encoder.layers:
            if not isinstance(m, RNNLayer) and hasattr(m, 'bucket') and front:
                m.dp, m.bucket = dp, len(buckets) - 1
        self._dp, self._n_rest = dp, len(rest)
        return dp

    def _apply(self, fn, *args, **kwargs):
        super()._apply(fn, *args, **kwargs)
        self._flatten()
        return self

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()

    # ---- reference surface --------------------------------------------------------------------------
    def set_state(self, prev_state, prev_attn):
        self.decoder.hidden_state = prev_state
        self.attention.att_layer.set_mem(prev_attn)

    def create_msg(self):
        msg = ['Model spec.| Encoder\'s downsampling rate of time axis is {}.'.format(self.encoder.sample_rate)]
        if self.encoder.vgg == 1:
            msg.append('           | VGG Extractor w/ time downsampling rate = 4 in encoder enabled.')
        if self.encoder.vgg == 5:
            msg.append('           | VGG Extractor w/ Layer normalization and downsampling rate = 4.')
        if self.enable_ctc:
            msg.append('           | CTC training on encoder enabled ( lambda = {}).'.format(self.ctc_weight))
        if self.enable_att:
            msg.append('           | {} attention decoder enabled ( lambda = {}).'.format(self.attention.mode, 1 - self.ctc_weight))
        msg.append('           | HIP path (gfx950), contraction precision = {}.'.format('bf16' if self.prec == H.BF16 else 'fp32'))
        return msg

    def forward(self, audio_feature, feature_len, decode_step, tf_rate=0.0, teacher=None,
                emb_decoder=None, get_dec_state=False, get_logit=False, ctc_async=False):
        '''Same contract as the reference (src/asr.py:89-177):
            audio_feature [B,T,D] fp32, feature_len [B], decode_step int, teacher [B,L] token ids or None.
           Returns ctc_output [B,T',V] (log-probs), encode_len [B], att_output [B,L,V] (logits),
                   att_seq [B,1,L,T'], dec_state.'''
        if not audio_feature.is_cuda:
            raise RuntimeError('ASR (HIP path) needs CUDA tensors; there is no CPU fallback')
        if emb_decoder is not None:
            raise NotImplementedError('embedding-regulariser plugin is outside the HIP path')
        ctx = _RunCtx(self)
        feature_len = feature_len.to(audio_feature.device)
        ctc_output, att_output, att_seq, dec_state = None, None, None, None
        encode_feature, encode_len = self.encoder(audio_feature.float(), feature_len, ctx)
        encode_feature = F_hip.to_f32_fn(encode_feature)      # the bf16-storage encoder stack hands over bf16
        # single-process training: the CTC branch (head here, loss in the step function) runs on the side stream beside the
        # decoder loop - see hipabi.side_branch; the caller joins before it uses the loss (`_asr_side` marks the output)
        side_ctc = (self.enable_ctc and self.enable_att and self.training and torch.is_grad_enabled() and H.overlap_enabled()
                    and H.ctc_side_enabled() and getattr(self, '_dp', None) is None)
        if self.enable_ctc:
            # the head's parameter gradients may run beside the BPTT when no gradient bucket is signalled before them
            self.ctc_layer[0]._asr_defer = getattr(self, '_dp', None) is None or H.overlap_dp_enabled()
            if side_ctc:
                with H.side_branch(True, encode_feature, encode_len):
                    ctc_output = F_hip.CTCHeadFn.apply(ctx.anchor, encode_feature, self.ctc_layer[0], self.prec, get_logit)
                ctc_output._asr_side = bool(ctc_async)      # the step function continues the branch and joins (src/step.py)
            else:
                ctc_output = F_hip.CTCHeadFn.apply(ctx.anchor, encode_feature, self.ctc_layer[0], self.prec, get_logit)
        if self.enable_att:
            L = int(decode_step)
            if not (self.decoder.fast and self.attention.fast and self.emb_drop == 0.0):
                from src.variants import variant_decoder
                att_output, att_seq, hs = variant_decoder(self, ctx.anchor, encode_feature, encode_len, L, teacher,
                                                          float(tf_rate) if teacher is not None else 1.0, ctx)
                if get_dec_state:
                    dec_state = hs
                if side_ctc and not ctc_async:
                    H.join_branch(ctc_output)
                return ctc_output, encode_len, att_output, att_seq, dec_state
            att_output, att_seq, hs = F_hip.AttDecoderFn.apply(ctx.anchor, encode_feature, encode_len, teacher, L, self, self.prec,
                                                               float(tf_rate) if teacher is not None else 1.0)
            if get_dec_state:
                dec_state = hs[:, :, -1, :]
        if side_ctc and not ctc_async:
            H.join_branch(ctc_output)        # ordinary callers get an output that is ordered on their stream
        return ctc_output, encode_len, att_output, att_seq, dec_state

    def fix_ctc_layer(self):
        for param in self.ctc_layer.parameters():
            param.requires_grad = False
