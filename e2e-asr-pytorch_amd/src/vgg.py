"""VGG front-ends of the encoder on the HIP path, with the reference's module/parameter names
(VGGExtractor src/module.py:659-716, VGGExtractor_LN src/module.py:582-657; `extractor.<i>.weight` keys).

Activations are channel-last images (B,T,F,C); every convolution is asr_conv3x3 (implicit GEMM on MFMA:
forward, input gradient with the flipped weight copy, weight gradient), pooling and LayerNorm-over-frequency
are the HBM-bound kernels of csrc/vgg.hip."""
import torch
import torch.nn as nn

from src import hipabi as H

FBANK_SIZE = 40


class CNNLayerNorm(nn.Module):
    def __init__(self, n_feats):
        super().__init__()
        self.layer_norm = nn.LayerNorm(n_feats)


def _e(shape, dev, dtype=torch.float32):
    return torch.empty(shape, dtype=dtype, device=dev)


class _VGGFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, feature, mod, prec):
        st = H.stream_ptr()
        dev = feature.device
        div = getattr(mod, 'time_div', 4)
        if feature.shape[1] % div != 0:
            feature = feature[:, :-(feature.shape[1] % div), :]
        fs = getattr(mod, 'freq_slice', None)
        if fs is not None:                       # one band of a frequency-split extractor: columns f0..f1 of every channel (a copy)
            feature = feature.reshape(feature.shape[0], feature.shape[1], mod.in_channel, -1)[..., fs[0]:fs[1]]
        feature = feature.contiguous()
        B, T = feature.shape[0], feature.shape[1]
        Cin, F = mod.in_channel, mod.freq_dim
        x = _e((B, T, F, Cin), dev)
        H.call('asr_permute_last2', H.ptr(feature), H.ptr(x), B * T, Cin, F, st)      # (.., C, F) -> (.., F, C)
        freq_only2 = getattr(mod, 'pool2_freq_only', False)
        saved = {'x': [], 'pre': [], 'stats': [], 'idx': [], 'dims': []}
        cur, t, f = x, T, F
        for li, (conv, ln) in enumerate(mod.conv_layers()):
            Co, Ci = conv.weight.shape[0], conv.weight.shape[1]
            wf = _e((Co, 9 * Ci), dev)
            H.call('asr_conv_weight_permute', H.ptr(conv.weight), H.ptr(wf), Co, Ci, 0, st)
            out = _e((B, t, f, Co), dev)
            saved['x'].append(cur)
            saved['dims'].append((t, f, Ci, Co))
            if ln is None:
                H.call('asr_conv3x3', H.ptr(cur), H.ptr(wf), H.ptr(out), H.ptr(conv.bias), B, t, f, Ci, Co, 0, H.ACT_RELU, 0, prec, st)
                saved['pre'].append(None); saved['stats'].append(None)
                act = out
            else:
                H.call('asr_conv3x3', H.ptr(cur), H.ptr(wf), H.ptr(out), H.ptr(conv.bias), B, t, f, Ci, Co, 0, H.ACT_NONE, 0, prec, st)
                act = _e((B, t, f, Co), dev)
                stats = _e((B * t * Co, 2), dev)
                H.call('asr_ln_freq_fwd', H.ptr(out), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(act), H.ptr(stats), B * t, f, Co, 1e-5, 1, st)
                saved['pre'].append(out); saved['stats'].append(stats)
            if li == 3 and freq_only2:
                # MaxPool2d((1, 2)): the 2 x 2 kernel on the image seen as B*t images of ONE row
                f2 = f // 2
                pooled = _e((B, t, f2, Co), dev)
                idx = _e((B, t, f2, Co), dev, torch.uint8)
                H.call('asr_maxpool2x2_fwd', H.ptr(act), H.ptr(pooled), H.ptr(idx), B * t, 1, f, Co, 1, f2, st)
                saved['idx'].append((idx, act, t, f, t, f2))
                cur, f = pooled, f2
            elif li in (1, 3):
                t2, f2 = ((t + 1) // 2, (f + 1) // 2) if mod.ceil_mode else (t // 2, f // 2)
                pooled = _e((B, t2, f2, Co), dev)
                idx = _e((B, t2, f2, Co), dev, torch.uint8)
                H.call('asr_maxpool2x2_fwd', H.ptr(act), H.ptr(pooled), H.ptr(idx), B, t, f, Co, t2, f2, st)
                saved['idx'].append((idx, act, t, f, t2, f2))
                cur, t, f = pooled, t2, f2
            else:
                saved['idx'].append(None)
                cur = act
            saved.setdefault('act', []).append(act)
        Co = cur.shape[-1]
        out = _e((B, t, Co * f), dev)
        H.call('asr_permute_last2', H.ptr(cur), H.ptr(out), B * t, f, Co, st)           # (.., F, C) -> (.., C, F)
        ctx.mod, ctx.prec, ctx.saved, ctx.B = mod, prec, saved, B
        ctx.final = (t, f, Co)
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, prec, sv, B = ctx.mod, ctx.prec, ctx.saved, ctx.B
        st = H.stream_ptr()
        dev = dout.device
        t, f, Co = ctx.final
        dout = dout.contiguous()
        g = _e((B, t, f, Co), dev)
        H.call('asr_permute_last2', H.ptr(dout), H.ptr(g), B * t, Co, f, st)            # (.., C, F) -> (.., F, C)
        layers = list(mod.conv_layers())
        for li in range(len(layers) - 1, -1, -1):
            conv, ln = layers[li]
            t_l, f_l, Ci, Co = sv['dims'][li]
            if sv['idx'][li] is not None:
                idx, act, tt, ff, t2, f2 = sv['idx'][li]
                gp = _e((B, tt, ff, Co), dev)
                if t2 == tt and li == 3 and getattr(mod, 'pool2_freq_only', False):
                    H.call('asr_maxpool2x2_bwd', H.ptr(g), H.ptr(idx), H.ptr(gp), B * tt, 1, ff, Co, 1, f2, st)
                else:
                    H.call('asr_maxpool2x2_bwd', H.ptr(g), H.ptr(idx), H.ptr(gp), B, tt, ff, Co, t2, f2, st)
                g = gp
            act = sv['act'][li]
            dpre = _e((B, t_l, f_l, Co), dev)
            if ln is None:
                H.call('asr_act_bwd', H.ptr(g), H.ptr(act), H.ptr(dpre), g.numel(), H.ACT_RELU, st)
            else:
                H.call('asr_ln_freq_bwd', H.ptr(g), H.ptr(sv['pre'][li]), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(sv['stats'][li]),
                       H.ptr(dpre), H.ptr(ln.weight.grad), H.ptr(ln.bias.grad), B * t_l, f_l, Co, 1, st)
            xin = sv['x'][li]
            dwf = torch.zeros((Co, 9 * Ci), dtype=torch.float32, device=dev)
            H.call('asr_conv3x3', H.ptr(xin), H.ptr(dpre), H.ptr(dwf), None, B, t_l, f_l, Ci, Co, 1, H.ACT_NONE, 1, prec, st)
            H.call('asr_conv_weight_permute', H.ptr(dwf), H.ptr(conv.weight.grad), Co, Ci, 2, st)
            H.call('asr_colsum', H.ptr(dpre), Co, B * t_l * f_l, Co, H.ptr(conv.bias.grad), st)
            if li > 0:
                wd = _e((Ci, 9 * Co), dev)
                H.call('asr_conv_weight_permute', H.ptr(conv.weight), H.ptr(wd), Co, Ci, 1, st)
                gin = _e((B, t_l, f_l, Ci), dev)
                H.call('asr_conv3x3', H.ptr(dpre), H.ptr(wd), H.ptr(gin), None, B, t_l, f_l, Co, Ci, 0, H.ACT_NONE, 0, prec, st)
                g = gin
        owner = getattr(mod, 'owner', None)
        if owner is not None:                    # a band of a frequency-split extractor: the bucket goes when both bands are done
            owner._bands_done += 1
            if owner._bands_done == 2 and owner.dp is not None:
                owner.dp.bucket_ready(owner.bucket)
        elif mod.dp is not None:
            mod.dp.bucket_ready(mod.bucket)
        ctx.saved = None
        return None, None, None, None


def _b16(shape, dev):
    return torch.empty(shape, dtype=torch.bfloat16, device=dev)


def vgg16_ok(mod, prec):
    """The bf16 front-end (csrc/vgg16.hip) covers bf16 contraction mode with channel counts that are multiples of 64 behind the
    first layer (both reference extractors: 128/256 and 64/128) and at most 4 input channels per tap group (9*Cin <= 40)."""
    import os
    if prec != H.BF16 or os.environ.get('ASR_VGG16', '1') == '0':
        return False
    return mod.init_dim % 64 == 0 and mod.hide_dim % 64 == 0 and 9 * mod.in_channel <= 40


class _VGG16Fn(torch.autograd.Function):
    """The VGG front-end on zero-bordered channel-last bf16 images P(T,F,C) = (B, T+2, F+2, C): every convolution is an implicit
    GEMM on the direct-to-LDS bf16 contraction kernel (asr_conv3x3_16), the weight gradient one launch of nine shifted-row TN
    contractions, pooling / CNNLayerNorm / layout changes run on the bordered images (reference src/module.py:582-716)."""

    @staticmethod
    def forward(ctx, anchor, feature, mod, prec):
        st = H.stream_ptr()
        dev = feature.device
        if feature.shape[1] % 4 != 0:
            feature = feature[:, :-(feature.shape[1] % 4), :]
        feature = feature.contiguous().float()
        B, T, _ = feature.shape
        Cin, F = mod.in_channel, mod.freq_dim
        K1p = (9 * Cin + 7) // 8 * 8
        sv = {'x': [], 'act': [], 'pre': [], 'stats': [], 'pool': [], 'dims': []}
        cur, t, f = None, T, F
        for li, (conv, ln) in enumerate(mod.conv_layers()):
            Co, Ci = conv.weight.shape[0], conv.weight.shape[1]
            Mp = B * (t + 2) * (f + 2)
            f32 = 0 if ln is None else 1           # pre-activations of a CNNLayerNorm stay fp32 (see asr_conv3x3_16)
            out = _b16((Mp, Co), dev) if ln is None else _e((Mp, Co), dev)
            act_code = H.ACT_RELU if ln is None else H.ACT_NONE
            if li == 0:
                x1 = _b16((Mp, K1p), dev)
                H.call('asr_vgg16_im2col', H.ptr(feature), H.ptr(x1), B, t, f, Cin, K1p, st)
                w16 = _b16((Co, K1p), dev)
                H.call('asr_conv_weight_pack16', H.ptr(conv.weight), H.ptr(w16), Co, Ci, K1p, 0, st)
                H.call('asr_conv3x3_16', H.ptr(x1), H.ptr(w16), H.ptr(out), H.ptr(conv.bias), B, t, f, Ci, Co, K1p, 0, act_code, f32, st)
                sv['x'].append(x1)
            else:
                w16 = _b16((Co, 9 * Ci), dev)
                H.call('asr_conv_weight_pack16', H.ptr(conv.weight), H.ptr(w16), Co, Ci, 9 * Ci, 0, st)
                H.call('asr_conv3x3_16', H.ptr(cur), H.ptr(w16), H.ptr(out), H.ptr(conv.bias), B, t, f, Ci, Co, 9 * Ci, 1, act_code, f32, st)
                sv['x'].append(cur)
            sv['dims'].append((t, f, Ci, Co))
            if ln is None:
                act = out
                sv['pre'].append(None); sv['stats'].append(None)
            else:
                act = _b16((Mp, Co), dev)
                stats = _e((B * (t + 2) * Co, 2), dev)
                H.call('asr_ln_freq16_fwd', H.ptr(out), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(act), H.ptr(stats), B, t, f, Co, 1e-5, 1, st)
                sv['pre'].append(out); sv['stats'].append(stats)
            sv['act'].append(act)
            if li in (1, 3):
                t2, f2 = ((t + 1) // 2, (f + 1) // 2) if mod.ceil_mode else (t // 2, f // 2)
                Mp2 = B * (t2 + 2) * (f2 + 2)
                pooled = _b16((Mp2, Co), dev)
                idx = torch.empty((Mp2, Co), dtype=torch.uint8, device=dev)
                H.call('asr_maxpool2x2_16_fwd', H.ptr(act), H.ptr(pooled), H.ptr(idx), B, t, f, Co, t2, f2, st)
                sv['pool'].append((idx, t, f, t2, f2))
                cur, t, f = pooled, t2, f2
            else:
                sv['pool'].append(None)
                cur = act
        Co = cur.shape[-1]
        out = _b16((B, t, Co * f), dev)
        H.call('asr_vgg16_output', H.ptr(cur), H.ptr(out), B, t, f, Co, st)
        ctx.mod, ctx.saved, ctx.B, ctx.final, ctx.K1p = mod, sv, B, (t, f, Co), K1p
        return out

    @staticmethod
    def backward(ctx, dout):
        mod, sv, B, K1p = ctx.mod, ctx.saved, ctx.B, ctx.K1p
        st = H.stream_ptr()
        dev = dout.device
        t, f, Co = ctx.final
        dout = dout.contiguous()
        if dout.dtype != torch.bfloat16:
            d16 = _b16(tuple(dout.shape), dev)
            H.call('asr_cast_bf16', H.ptr(dout), H.ptr(d16), dout.numel(), st)
            dout = d16
        g = _b16((B * (t + 2) * (f + 2), Co), dev)
        H.call('asr_vgg16_output_bwd', H.ptr(dout), H.ptr(g), B, t, f, Co, st)
        layers = list(mod.conv_layers())
        for li in range(len(layers) - 1, -1, -1):
            conv, ln = layers[li]
            t_l, f_l, Ci, Co = sv['dims'][li]
            Mp = B * (t_l + 2) * (f_l + 2)
            if sv['pool'][li] is not None:
                idx, tt, ff, t2, f2 = sv['pool'][li]
                gp = _b16((Mp, Co), dev)
                H.call('asr_maxpool2x2_16_bwd', H.ptr(g), H.ptr(idx), H.ptr(gp), B, tt, ff, Co, t2, f2, st)
                g = gp
            dpre = _b16((Mp, Co), dev)
            if ln is None:
                H.call('asr_act_bwd16', H.ptr(g), H.ptr(sv['act'][li]), H.ptr(dpre), Mp * Co, H.ACT_RELU, st)
            else:
                H.call('asr_ln_freq16_bwd', H.ptr(g), H.ptr(sv['pre'][li]), H.ptr(ln.weight), H.ptr(ln.bias), H.ptr(sv['stats'][li]),
                       H.ptr(dpre), H.ptr(ln.weight.grad), H.ptr(ln.bias.grad), H.ptr(conv.bias.grad), B, t_l, f_l, Co, 1, st)
            xin = sv['x'][li]
            if li == 0:
                dwf = torch.zeros((Co, K1p), dtype=torch.float32, device=dev)
                H.gemm16(dpre, xin, dwf, Co, K1p, Mp, Co, K1p, K1p, 0, 0, accum=1, splits=min(64, max(1, Mp // 4096)))
                H.call('asr_conv_weight_fold', H.ptr(dwf), H.ptr(conv.weight.grad), Co, Ci, K1p, st)
            else:
                dwf = torch.zeros((Co, 9 * Ci), dtype=torch.float32, device=dev)
                tiles = ((Co + 127) // 128) * ((Ci + 127) // 128)
                splits = max(1, min(Mp // 2048, 96 // tiles))          # x 9 taps: ~800 workgroups per launch
                H.call('asr_conv3x3_16_wgrad', H.ptr(xin), H.ptr(dpre), H.ptr(dwf), B, t_l, f_l, Ci, Co, 9 * Ci, splits, st)
                H.call('asr_conv_weight_fold', H.ptr(dwf), H.ptr(conv.weight.grad), Co, Ci, 9 * Ci, st)
            if ln is None:         # (behind a CNNLayerNorm the bias gradient comes out of asr_ln_freq16_bwd, in fp32)
                H.call('asr_colsum16', H.ptr(dpre), Co, Mp, Co, H.ptr(conv.bias.grad), None, 0, st)
            if li > 0:
                wd = _b16((Ci, 9 * Co), dev)
                H.call('asr_conv_weight_pack16', H.ptr(conv.weight), H.ptr(wd), Co, Ci, 9 * Co, 1, st)
                gin = _b16((Mp, Ci), dev)
                H.call('asr_conv3x3_16', H.ptr(dpre), H.ptr(wd), H.ptr(gin), None, B, t_l, f_l, Co, Ci, 9 * Co, 1, H.ACT_NONE, 0, st)
                g = gin
        if mod.dp is not None:
            mod.dp.bucket_ready(mod.bucket)
        ctx.saved = None
        return None, None, None, None


class _VGGBase(nn.Module):
    def check_dim(self, input_dim):
        if input_dim % FBANK_SIZE != 0:
            raise ValueError('HIP VGG front-end expects 40-bin fbank channels (input dim %d)' % input_dim)
        return input_dim // FBANK_SIZE, FBANK_SIZE, (FBANK_SIZE // 4) * self.hide_dim

    def forward(self, feature, feat_len, ctx=None):
        fn = _VGG16Fn if vgg16_ok(self, ctx.prec) else _VGGFn
        out = fn.apply(ctx.anchor, feature, self, ctx.prec)
        return out, feat_len // 4


class VGGExtractor(_VGGBase):
    ''' VGG extractor (reference src/module.py:659-716): 2x(conv3x3+ReLU) -> maxpool(ceil) -> 2x(conv3x3+ReLU) -> maxpool(ceil) '''

    def __init__(self, input_dim):
        super().__init__()
        self.init_dim, self.hide_dim, self.ceil_mode = 128, 256, True
        self.in_channel, self.freq_dim, self.out_dim = self.check_dim(input_dim)
        self.dp, self.bucket = None, None
        self.extractor = nn.Sequential(
            nn.Conv2d(self.in_channel, self.init_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(self.init_dim, self.init_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2, ceil_mode=True),
            nn.Conv2d(self.init_dim, self.hide_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(self.hide_dim, self.hide_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2, ceil_mode=True))

    def conv_layers(self):
        return [(self.extractor[i], None) for i in (0, 2, 5, 7)]


class VGGExtractor_LN(_VGGBase):
    ''' VGG + LayerNorm over frequency (reference src/module.py:582-657), floor-mode pooling '''

    def __init__(self, input_dim):
        super().__init__()
        self.init_dim, self.hide_dim, self.ceil_mode = 64, 128, False
        self.in_channel, self.freq_dim, self.out_dim = self.check_dim(input_dim)
        self.upstream = False
        self.dp, self.bucket = None, None
        n = FBANK_SIZE
        self.extractor = nn.Sequential(
            nn.Conv2d(self.in_channel, self.init_dim, 3, stride=1, padding=1), CNNLayerNorm(n), nn.ReLU(),
            nn.Conv2d(self.init_dim, self.init_dim, 3, stride=1, padding=1), CNNLayerNorm(n), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(self.init_dim, self.hide_dim, 3, stride=1, padding=1), CNNLayerNorm(n // 2), nn.ReLU(),
            nn.Conv2d(self.hide_dim, self.hide_dim, 3, stride=1, padding=1), CNNLayerNorm(n // 2), nn.ReLU(),
            nn.MaxPool2d(2, stride=2))

    def conv_layers(self):
        return [(self.extractor[i], self.extractor[i + 1].layer_norm) for i in (0, 3, 7, 10)]


class VGGExtractor2(_VGGBase):
    """VGG extractor that halves time once (reference src/module.py:843-905): 64 / 128 channels, floor-mode 2 x 2 pooling,
    then MaxPool2d((1, 2)) - frequency only."""

    def __init__(self, input_dim):
        super().__init__()
        self.init_dim, self.hide_dim, self.ceil_mode = 64, 128, False
        self.time_div, self.pool2_freq_only = 2, True
        self.in_channel, self.freq_dim, self.out_dim = self.check_dim(input_dim)
        self.dp, self.bucket = None, None
        self.extractor = nn.Sequential(
            nn.Conv2d(self.in_channel, self.init_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(self.init_dim, self.init_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d(2, stride=2),
            nn.Conv2d(self.init_dim, self.hide_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.Conv2d(self.hide_dim, self.hide_dim, 3, stride=1, padding=1), nn.ReLU(),
            nn.MaxPool2d((1, 2), stride=(1, 2)))

    def conv_layers(self):
        return [(self.extractor[i], None) for i in (0, 2, 5, 7)]

    def forward(self, feature, feat_len, ctx=None):
        return _VGGFn.apply(ctx.anchor, feature, self, ctx.prec), feat_len // 2


class _Band(object):
    """One band (low / high frequencies) of a frequency-split extractor as _VGGFn sees it."""

    def __init__(self, owner, seq, in_channel, f0, f1, time_div, pool2_freq_only):
        self.owner, self.seq, self.in_channel, self.freq_slice, self.freq_dim = owner, seq, in_channel, (f0, f1), f1 - f0
        self.time_div, self.pool2_freq_only, self.ceil_mode, self.dp, self.bucket = time_div, pool2_freq_only, False, None, None

    def conv_layers(self):
        return [(self.seq[i], None) for i in (0, 2, 5, 7)]


class FreqVGGExtractor(nn.Module):
    """Frequency-split VGG (reference src/module.py:746-841; `pool2_freq_only` = FreqVGGExtractor2, :907-1001): the bins below
    `split_freq` go through a narrow stack (low_dim, 2 low_dim channels), the rest through a wide one (64 - low_dim,
    128 - 2 low_dim); the two outputs are concatenated."""

    def __init__(self, input_dim, split_freq, low_dim=4, pool2_freq_only=False):
        super().__init__()
        if input_dim % FBANK_SIZE != 0:
            raise ValueError('HIP VGG front-end expects 40-bin fbank channels (input dim %d)' % input_dim)
        self.split_freq, self.in_channel, self.freq_dim = split_freq, input_dim // FBANK_SIZE, FBANK_SIZE
        assert split_freq % 4 == 0 and 0 < split_freq < self.freq_dim
        li, lh, hi, hh = low_dim, 2 * low_dim, 64 - low_dim, 128 - 2 * low_dim
        self.low_out_dim, self.high_out_dim = split_freq // 4 * lh, (self.freq_dim - split_freq) // 4 * hh
        self.out_dim = self.low_out_dim + self.high_out_dim
        self.time_div = 2 if pool2_freq_only else 4
        self.dp, self.bucket, self._bands_done = None, None, 0
        last = (lambda: nn.MaxPool2d((1, 2), stride=(1, 2))) if pool2_freq_only else (lambda: nn.MaxPool2d(2, stride=2))

        def stack(c1, c2):
            return nn.Sequential(nn.Conv2d(self.in_channel, c1, 3, stride=1, padding=1), nn.ReLU(),
                                 nn.Conv2d(c1, c1, 3, stride=1, padding=1), nn.ReLU(), nn.MaxPool2d(2, stride=2),
                                 nn.Conv2d(c1, c2, 3, stride=1, padding=1), nn.ReLU(),
                                 nn.Conv2d(c2, c2, 3, stride=1, padding=1), nn.ReLU(), last())
        self.low_extractor, self.high_extractor = stack(li, lh), stack(hi, hh)
        self._bands = (_Band(self, self.low_extractor, self.in_channel, 0, split_freq, self.time_div, pool2_freq_only),
                       _Band(self, self.high_extractor, self.in_channel, split_freq, self.freq_dim, self.time_div, pool2_freq_only))

    def forward(self, feature, feat_len, ctx=None):
        self._bands_done = 0
        lo = _VGGFn.apply(ctx.anchor, feature, self._bands[0], ctx.prec)
        hi = _VGGFn.apply(ctx.anchor, feature, self._bands[1], ctx.prec)
        return torch.cat((lo, hi), dim=-1), feat_len // self.time_div


class FreqVGGExtractor2(FreqVGGExtractor):
    def __init__(self, input_dim, split_freq, low_dim=4):
        super().__init__(input_dim, split_freq, low_dim, pool2_freq_only=True)
