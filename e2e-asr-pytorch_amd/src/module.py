"""Encoder / attention building blocks with the reference's module and parameter names
(src/module.py of DanielLin94144/E2E-ASR-Pytorch), so that its checkpoints load unchanged.

The nn.Modules here only OWN parameters; all arithmetic goes through the HIP kernels via
src/functions.py.  Supported on the MI355X path: RNNLayer (LSTM, uni/bidirectional, LayerNorm, dropout,
'drop'/'concat' down-sampling, projection), Downsampler (vgg 6), VGGExtractor (vgg 1), VGGExtractor_LN
(vgg 5), LocationAwareAttention; GRU layers and the other attention variants: src/variants.py.
"""
import torch
import torch.nn as nn

from src import functions as F_hip
from src import hipabi as H

FBANK_SIZE = 40


class LSTMParams(nn.Module):
    """Parameter container with nn.LSTM's names (weight_ih_l0[_reverse], ...), single layer."""

    def __init__(self, input_dim, dim, bidirectional, num_layers=1):
        super().__init__()
        self.input_dim, self.dim, self.bidirectional, self.num_layers = input_dim, dim, bidirectional, num_layers
        for l in range(num_layers):
            din = input_dim if l == 0 else dim * (2 if bidirectional else 1)
            for sfx in ([''] + (['_reverse'] if bidirectional else [])):
                self.register_parameter('weight_ih_l%d%s' % (l, sfx), nn.Parameter(torch.empty(4 * dim, din)))
                self.register_parameter('weight_hh_l%d%s' % (l, sfx), nn.Parameter(torch.empty(4 * dim, dim)))
                self.register_parameter('bias_ih_l%d%s' % (l, sfx), nn.Parameter(torch.empty(4 * dim)))
                self.register_parameter('bias_hh_l%d%s' % (l, sfx), nn.Parameter(torch.empty(4 * dim)))


class RNNLayer(nn.Module):
    """BiLSTM + LayerNorm/dropout/down-sampling/projection wrapper (reference src/module.py:1003-1081)."""

    def __init__(self, input_dim, module, dim, bidirection, dropout, layer_norm, sample_rate, sample_style, proj, batch_size=None):
        super().__init__()
        if module.upper() not in ('LSTM', 'GRU'):
            raise NotImplementedError('HIP path implements LSTM and GRU encoder cells (got %s)' % module)
        self.module = module.upper()
        if sample_style not in ('drop', 'concat'):
            raise ValueError('Unsupported Sample Style: ' + sample_style)
        self.dim, self.nd = dim, (2 if bidirection else 1)
        rnn_out_dim = self.nd * dim
        self.out_dim = sample_rate * rnn_out_dim if (sample_rate > 1 and sample_style == 'concat') else rnn_out_dim
        self.dropout, self.layer_norm, self.sample_rate, self.sample_style, self.proj = dropout, layer_norm, sample_rate, sample_style, proj
        if self.module == 'LSTM':
            self.layer = LSTMParams(input_dim, dim, bidirection)
        else:
            from src.variants import RNNParams
            self.layer = RNNParams('GRU', input_dim, dim, bidirection)     # persistent kernels are LSTM-only: src/variants.py
        if layer_norm:
            self.ln = nn.LayerNorm(rnn_out_dim)
        if proj:
            self.pj = nn.Linear(rnn_out_dim, rnn_out_dim)
        # concatenated views into the flat buffers, set by ASR._flatten()
        self.w_ih_cat = self.w_hh_cat = self.b_ih_cat = self.b_hh_cat = None
        self.g_w_ih_cat = self.g_w_hh_cat = self.g_b_ih_cat = self.g_b_hh_cat = None
        self.dp, self.bucket = None, None      # data-parallel hook: (FlatDataParallel, bucket index)

    def flat_groups(self):
        """Parameter groups that must be laid out back to back in the flat buffer."""
        sfx = [''] + (['_reverse'] if self.nd == 2 else [])
        groups = [[getattr(self.layer, n + s) for s in sfx] for n in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0')]
        if self.layer_norm:
            groups += [[self.ln.weight], [self.ln.bias]]
        if self.proj:
            groups += [[self.pj.weight], [self.pj.bias]]
        return groups

    def bind_flat(self, view_of):
        """view_of(params, shape, grad) -> tensor view over consecutive params."""
        H4, Din = self.layer.weight_ih_l0.shape
        groups = self.flat_groups()
        self.w_ih_cat = view_of(groups[0], (self.nd * H4, Din), False)
        self.w_hh_cat = view_of(groups[1], (self.nd, H4, self.dim), False)
        self.b_ih_cat = view_of(groups[2], (self.nd * H4,), False)
        self.b_hh_cat = view_of(groups[3], (self.nd * H4,), False)
        self.g_w_ih_cat = view_of(groups[0], (self.nd * H4, Din), True)
        self.g_w_hh_cat = view_of(groups[1], (self.nd, H4, self.dim), True)
        self.g_b_ih_cat = view_of(groups[2], (self.nd * H4,), True)
        self.g_b_hh_cat = view_of(groups[3], (self.nd * H4,), True)

    def forward(self, input_x, x_len, ctx=None):
        train = self.training and self.dropout > 0
        seed = ctx.next_seed() if (ctx is not None and train) else 0
        if self.module == 'GRU':
            from src.variants import gru_layer_forward
            out = gru_layer_forward(self, input_x, ctx, train, seed)
        elif F_hip.rnn_fast_ok(self, input_x, ctx.prec):
            out = F_hip.RNNLayerFastFn.apply(ctx.anchor, input_x, self, train, seed)       # bf16 out
        else:
            out = F_hip.RNNLayerFn.apply(ctx.anchor, F_hip.to_f32_fn(input_x), self, train, seed, ctx.prec)
        if self.sample_rate > 1:
            x_len = x_len // self.sample_rate
        return out, x_len


class Downsampler(nn.Module):
    """vgg: 6 — keep every 4th frame (reference src/module.py:719-729)."""

    def __init__(self, input_dim):
        super().__init__()
        self.sample_rate = 4
        self.out_dim = input_dim

    def forward(self, feature, feat_len, ctx=None):
        feature = feature.contiguous()
        B, T, D = feature.shape
        T2 = (T + self.sample_rate - 1) // self.sample_rate
        out = torch.empty((B, T2, D), dtype=torch.float32, device=feature.device)
        H.call('asr_dropout_downsample_fwd', H.ptr(feature), H.ptr(out), B, T, D, T2, self.sample_rate, 0, 0.0, 0,
               H.stream_ptr())
        return out, feat_len // self.sample_rate


class Featemb_Extractor(nn.Module):
    """vgg: 7 - one Linear(input, 256) in front of the recurrent layers (reference src/module.py:732-742,
    config/librispeech_asr_upstream.yaml)."""

    def __init__(self, input_dim):
        super().__init__()
        self.emb_dim = self.out_dim = 256
        self.dense = nn.Linear(input_dim, self.emb_dim)

    def forward(self, feature, feat_len, ctx=None):
        from src.variants import LinearActFn
        return LinearActFn.apply(ctx.anchor, feature, self.dense.weight, self.dense.bias, H.ACT_NONE, ctx.prec), feat_len


class LocationAwareAttention(nn.Module):
    """Parameter container of the location-aware attention (reference src/module.py:1135-1173)."""

    def __init__(self, kernel_size, kernel_num, dim, num_head, temperature):
        super().__init__()
        self.kernel_size, self.kernel_num, self.dim, self.num_head, self.temperature = kernel_size, kernel_num, dim, num_head, temperature
        self.loc_conv = nn.Conv1d(num_head, kernel_num, kernel_size=2 * kernel_size + 1, padding=kernel_size, bias=False)
        self.loc_proj = nn.Linear(kernel_num, dim, bias=False)
        self.gen_energy = nn.Linear(dim, 1)
        self.prev_att = None

    def reset_mem(self):
        self.prev_att = None

    def set_mem(self, prev_att):
        self.prev_att = prev_att
