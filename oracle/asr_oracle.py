"""CPU oracle for the joint CTC-attention training/decoding hot path.  TEST INFRASTRUCTURE ONLY.

A from-scratch fp32 restatement (plain torch/numpy on the CPU, explicit loops for the recurrences) of
the algorithm of DanielLin94144/E2E-ASR-Pytorch along the path named in SURVEY.md §8(a).  It is the
checker for the HIP kernels: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import it; the product path (e2e-asr-pytorch_amd/) never does.

Pinned: tests/test_oracle_golden.py compares every function here with tests/golden/*.npz, which were
produced by importing the genuine reference in the build container (tests/golden/gen_golden.py).
Third-party arithmetic the reference delegates to (not vendored there): torch.nn.LSTM / Conv2d /
CTCLoss (`torch>=1.2.0`, reference requirements.txt:8; goldens made with torch 2.10.0) and
torchaudio's Spectrogram/MelScale (unpinned) — the latter restated from its documented definition and
pinned only by self-consistency goldens (STFT/mel parity unpinned).

All parameters are passed as a dict keyed by the reference's state_dict names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NEG_INF = float('-inf')


# -------------------------------------------------------------------------------------------------
# configuration
# -------------------------------------------------------------------------------------------------
class ModelCfg(object):
    """Flattened view of the YAML `model:` block (reference src/asr.py:15, :393, :281, :186)."""

    def __init__(self, model_cfg, input_size, vocab_size):
        enc, att, dec = model_cfg['encoder'], model_cfg.get('attention'), model_cfg.get('decoder')
        self.input_size = input_size
        self.vocab_size = vocab_size
        self.ctc_weight = float(model_cfg['ctc_weight'])
        self.enable_ctc = self.ctc_weight > 0
        self.enable_att = self.ctc_weight != 1
        self.vgg = int(enc['vgg'])
        self.vgg_freq = int(enc.get('vgg_freq', -1))
        self.vgg_low_filt = int(enc.get('vgg_low_filt', -1))
        self.enc_dim = list(enc['dim'])
        self.enc_dropout = list(enc['dropout'])
        self.enc_layer_norm = list(enc['layer_norm'])
        self.enc_proj = list(enc['proj'])
        self.enc_sample_rate = list(enc['sample_rate'])
        self.enc_sample_style = enc['sample_style']
        self.bidirection = bool(enc['bidirection'])
        self.enc_module = enc['module'].upper()
        assert self.enc_module in ('LSTM', 'GRU'), 'oracle covers LSTM and GRU encoders'
        self.emb_drop = float(model_cfg.get('emb_drop', 0.0))
        if self.enable_att:
            self.att_dim = int(att['dim'])
            self.att_temperature = float(att['temperature'])
            self.loc_kernel_size = int(att['loc_kernel_size'])
            self.loc_kernel_num = int(att['loc_kernel_num'])
            self.att_mode = att['mode'].lower()
            self.num_head = int(att['num_head'])
            self.v_proj = bool(att['v_proj'])
            assert self.att_mode in ('loc', 'dot')
            self.dec_dim = int(dec['dim'])
            self.dec_layer = int(dec['layer'])
            self.dec_dropout = float(dec['dropout'])
            self.dec_module = dec['module'].upper()
            assert self.dec_module in ('LSTM', 'GRU')

    # --- shapes ---------------------------------------------------------------------------------
    def vgg_out_dim(self):
        if self.vgg == 0:
            return self.input_size
        if self.vgg == 1:
            return (40 // 4) * 256
        if self.vgg in (3, 5):
            return (40 // 4) * 128
        if self.vgg in (2, 4):
            low, sf = self.vgg_low_filt, self.vgg_freq
            return sf // 4 * (2 * low) + (40 - sf) // 4 * (128 - 2 * low)
        if self.vgg == 6:
            return self.input_size
        if self.vgg == 7:
            return 256
        raise NotImplementedError('vgg=%d' % self.vgg)

    def enc_out_dim(self):
        d = self.vgg_out_dim()
        for l, h in enumerate(self.enc_dim):
            d = (2 if self.bidirection else 1) * h
            if self.enc_sample_rate[l] > 1 and self.enc_sample_style == 'concat':
                d *= self.enc_sample_rate[l]
        return d


def param_shapes(cfg):
    """state_dict key -> shape, in the reference's registration order (verified in gen_golden.py)."""
    s = {}
    li = 0
    d = cfg.input_size
    if cfg.vgg in (1, 5):
        c_in = cfg.input_size // 40
        c1, c2 = (128, 256) if cfg.vgg == 1 else (64, 128)
        if cfg.vgg == 1:
            idx = [0, 2, 5, 7]
            chans = [(c_in, c1), (c1, c1), (c1, c2), (c2, c2)]
            for i, (ci, co) in zip(idx, chans):
                s['encoder.layers.0.extractor.%d.weight' % i] = (co, ci, 3, 3)
                s['encoder.layers.0.extractor.%d.bias' % i] = (co,)
        else:
            idx = [0, 3, 7, 10]
            chans = [(c_in, c1), (c1, c1), (c1, c2), (c2, c2)]
            lnd = [40, 40, 20, 20]
            for i, (ci, co), n in zip(idx, chans, lnd):
                s['encoder.layers.0.extractor.%d.weight' % i] = (co, ci, 3, 3)
                s['encoder.layers.0.extractor.%d.bias' % i] = (co,)
                s['encoder.layers.0.extractor.%d.layer_norm.weight' % (i + 1)] = (n,)
                s['encoder.layers.0.extractor.%d.layer_norm.bias' % (i + 1)] = (n,)
        d = cfg.vgg_out_dim()
        li = 1
    elif cfg.vgg in (2, 3, 4):
        c_in = cfg.input_size // 40
        towers = [('extractor', 64, 128)] if cfg.vgg == 3 else \
                 [('low_extractor', cfg.vgg_low_filt, 2 * cfg.vgg_low_filt), ('high_extractor', 64 - cfg.vgg_low_filt, 128 - 2 * cfg.vgg_low_filt)]
        for tw, c1, c2 in towers:
            for i, (ci, co) in zip([0, 2, 5, 7], [(c_in, c1), (c1, c1), (c1, c2), (c2, c2)]):
                s['encoder.layers.0.%s.%d.weight' % (tw, i)] = (co, ci, 3, 3)
                s['encoder.layers.0.%s.%d.bias' % (tw, i)] = (co,)
        d = cfg.vgg_out_dim()
        li = 1
    elif cfg.vgg == 6:
        li = 1
    elif cfg.vgg == 7:                       # Featemb_Extractor: one Linear(input, 256), src/module.py:732-742
        s['encoder.layers.0.dense.weight'] = (256, cfg.input_size)
        s['encoder.layers.0.dense.bias'] = (256,)
        d = 256
        li = 1
    for l, h in enumerate(cfg.enc_dim):
        pre = 'encoder.layers.%d.' % (li + l)
        ng = 4 if cfg.enc_module == 'LSTM' else 3
        for sfx in ([''] + (['_reverse'] if cfg.bidirection else [])):
            s[pre + 'layer.weight_ih_l0' + sfx] = (ng * h, d)
            s[pre + 'layer.weight_hh_l0' + sfx] = (ng * h, h)
            s[pre + 'layer.bias_ih_l0' + sfx] = (ng * h,)
            s[pre + 'layer.bias_hh_l0' + sfx] = (ng * h,)
        out = (2 if cfg.bidirection else 1) * h
        if cfg.enc_layer_norm[l]:
            s[pre + 'ln.weight'] = (out,)
            s[pre + 'ln.bias'] = (out,)
        if cfg.enc_proj[l]:
            s[pre + 'pj.weight'] = (out, out)
            s[pre + 'pj.bias'] = (out,)
        d = out * (cfg.enc_sample_rate[l] if (cfg.enc_sample_rate[l] > 1 and cfg.enc_sample_style == 'concat') else 1)
    enc_out = d
    if cfg.enable_ctc:
        s['ctc_layer.0.weight'] = (cfg.vocab_size, enc_out)
        s['ctc_layer.0.bias'] = (cfg.vocab_size,)
    if cfg.enable_att:
        s['pre_embed.weight'] = (cfg.vocab_size, cfg.dec_dim)
        dg = 4 if cfg.dec_module == 'LSTM' else 3
        for l in range(cfg.dec_layer):
            din = enc_out + cfg.dec_dim if l == 0 else cfg.dec_dim
            s['decoder.layers.weight_ih_l%d' % l] = (dg * cfg.dec_dim, din)
            s['decoder.layers.weight_hh_l%d' % l] = (dg * cfg.dec_dim, cfg.dec_dim)
            s['decoder.layers.bias_ih_l%d' % l] = (dg * cfg.dec_dim,)
            s['decoder.layers.bias_hh_l%d' % l] = (dg * cfg.dec_dim,)
        s['decoder.char_trans.weight'] = (cfg.vocab_size, cfg.dec_dim)
        s['decoder.char_trans.bias'] = (cfg.vocab_size,)
        qd = cfg.dec_dim * cfg.dec_layer
        nh = cfg.num_head
        s['attention.proj_q.weight'] = (cfg.att_dim * nh, qd)
        s['attention.proj_q.bias'] = (cfg.att_dim * nh,)
        s['attention.proj_k.weight'] = (cfg.att_dim * nh, enc_out)
        s['attention.proj_k.bias'] = (cfg.att_dim * nh,)
        if cfg.v_proj:
            s['attention.proj_v.weight'] = (enc_out * nh, enc_out)
            s['attention.proj_v.bias'] = (enc_out * nh,)
        if cfg.att_mode == 'loc':
            s['attention.att_layer.loc_conv.weight'] = (cfg.loc_kernel_num, nh, 2 * cfg.loc_kernel_size + 1)
            s['attention.att_layer.loc_proj.weight'] = (cfg.att_dim, cfg.loc_kernel_num)
            s['attention.att_layer.gen_energy.weight'] = (1, cfg.att_dim)
            s['attention.att_layer.gen_energy.bias'] = (1,)
        if nh > 1:
            s['attention.merge_head.weight'] = (enc_out, enc_out * nh)
            s['attention.merge_head.bias'] = (enc_out,)
    return s


def seeded_state_dict(shapes, seed, bias_scale=0.1):
    """Deterministic weights for fixtures: N(0, 1/sqrt(fan_in)) matrices, small non-zero vectors
    (the reference zero-initialises biases, src/util.py:60-83; non-zero ones exercise more code)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for k, shp in shapes.items():
        if len(shp) == 1:
            v = rng.standard_normal(shp) * bias_scale
            if k.endswith('layer_norm.weight') or k.endswith('ln.weight'):
                v = 1.0 + v
        else:
            fan_in = int(np.prod(shp[1:]))
            v = rng.standard_normal(shp) / math.sqrt(fan_in)
        sd[k] = torch.from_numpy(v.astype(np.float32))
    return sd


# -------------------------------------------------------------------------------------------------
# encoder  (reference src/module.py:1040-1081, src/asr.py:459-464)
# -------------------------------------------------------------------------------------------------
def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse, return_gates=False):
    """One direction of nn.LSTM(batch_first) with zero initial state; gate order i,f,g,o.
    Padded frames are processed like any other (the reference never packs: src/module.py:1047-1054)."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xg = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        g = xg[:, t] + h @ w_hh.t()
        i, f, gg, o = g.chunk(4, dim=-1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[t] = h
    return torch.stack(out, dim=1)


def gru_direction(x, w_ih, w_hh, b_ih, b_hh, reverse):
    """One direction of nn.GRU(batch_first), zero initial state; gate order r,z,n; n = tanh(W_in x + b_in + r (W_hn h + b_hn))."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    xg = x @ w_ih.t() + b_ih
    h = x.new_zeros(B, H)
    out = [None] * T
    order = range(T - 1, -1, -1) if reverse else range(T)
    for t in order:
        hg = h @ w_hh.t() + b_hh
        xr, xz, xn = xg[:, t].chunk(3, dim=-1)
        hr, hz, hn = hg.chunk(3, dim=-1)
        r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
        n = torch.tanh(xn + r * hn)
        h = (1 - z) * n + z * h
        out[t] = h
    return torch.stack(out, dim=1)


def bigru(x, P, pre, bidirection=True):
    outs = [gru_direction(x, P[pre + 'weight_ih_l0'], P[pre + 'weight_hh_l0'], P[pre + 'bias_ih_l0'], P[pre + 'bias_hh_l0'], False)]
    if bidirection:
        outs.append(gru_direction(x, P[pre + 'weight_ih_l0_reverse'], P[pre + 'weight_hh_l0_reverse'],
                                  P[pre + 'bias_ih_l0_reverse'], P[pre + 'bias_hh_l0_reverse'], True))
    return torch.cat(outs, dim=-1)


def bilstm(x, P, pre, bidirection=True):
    outs = [lstm_direction(x, P[pre + 'weight_ih_l0'], P[pre + 'weight_hh_l0'],
                           P[pre + 'bias_ih_l0'], P[pre + 'bias_hh_l0'], False)]
    if bidirection:
        outs.append(lstm_direction(x, P[pre + 'weight_ih_l0_reverse'], P[pre + 'weight_hh_l0_reverse'],
                                   P[pre + 'bias_ih_l0_reverse'], P[pre + 'bias_hh_l0_reverse'], True))
    return torch.cat(outs, dim=-1)


def bilstm_aten(x, P, pre, bidirection=True):
    """Same map through torch's fused LSTM (used for the timed CPU baseline and as a cross-check)."""
    H = P[pre + 'weight_hh_l0'].shape[1]
    names = ['weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0']
    if bidirection:
        names += [n + '_reverse' for n in names]
    flat = [P[pre + n] for n in names]
    B = x.shape[0]
    nd = 2 if bidirection else 1
    hx = (x.new_zeros(nd, B, H), x.new_zeros(nd, B, H))
    out, _, _ = torch._VF.lstm(x, hx, flat, True, 1, 0.0, False, bidirection, True)
    return out


def rnn_layer(x, x_len, P, pre, cfg, l, drop_mask=None, lstm_impl=bilstm):
    """RNNLayer.forward. drop_mask: None (eval) or a {0,1} float tensor of the LSTM output's shape."""
    y = (bigru if cfg.enc_module == 'GRU' else lstm_impl)(x, P, pre + 'layer.', cfg.bidirection)
    if cfg.enc_layer_norm[l]:
        y = F.layer_norm(y, (y.shape[-1],), P[pre + 'ln.weight'], P[pre + 'ln.bias'])
    if drop_mask is not None:
        y = y * drop_mask / (1.0 - cfg.enc_dropout[l])
    r = cfg.enc_sample_rate[l]
    if r > 1:
        x_len = x_len // r
        if cfg.enc_sample_style == 'drop':
            y = y[:, ::r, :].contiguous()
        else:
            B, T, D = y.shape
            if T % r != 0:
                y = y[:, :-(T % r), :]
            y = y.contiguous().view(B, T // r, D * r)
    if cfg.enc_proj[l]:
        y = torch.tanh(y @ P[pre + 'pj.weight'].t() + P[pre + 'pj.bias'])
    return y, x_len


def vgg_view_input(x, x_len, c_in, freq):
    """src/module.py:694-705: len//4, crop T to a multiple of 4, (B,T,C*F) -> (B,C,T,F)."""
    x_len = x_len // 4
    if x.shape[1] % 4 != 0:
        x = x[:, :-(x.shape[1] % 4), :].contiguous()
    B, T, _ = x.shape
    return x.view(B, T, c_in, freq).transpose(1, 2), x_len


def vgg_extractor(x, x_len, P, pre, layer_norm):
    """VGGExtractor (src/module.py:659-716) / VGGExtractor_LN (src/module.py:582-657)."""
    c_in = x.shape[-1] // 40
    y, x_len = vgg_view_input(x, x_len, c_in, 40)
    if not layer_norm:
        for i in (0, 2):
            y = F.relu(F.conv2d(y, P[pre + 'extractor.%d.weight' % i], P[pre + 'extractor.%d.bias' % i], padding=1))
        y = F.max_pool2d(y, 2, stride=2, ceil_mode=True)
        for i in (5, 7):
            y = F.relu(F.conv2d(y, P[pre + 'extractor.%d.weight' % i], P[pre + 'extractor.%d.bias' % i], padding=1))
        y = F.max_pool2d(y, 2, stride=2, ceil_mode=True)
    else:
        def block(y, i):
            y = F.conv2d(y, P[pre + 'extractor.%d.weight' % i], P[pre + 'extractor.%d.bias' % i], padding=1)
            n = y.shape[-1]
            y = F.layer_norm(y, (n,), P[pre + 'extractor.%d.layer_norm.weight' % (i + 1)],
                             P[pre + 'extractor.%d.layer_norm.bias' % (i + 1)])
            return F.relu(y)
        y = block(block(y, 0), 3)
        y = F.max_pool2d(y, 2, stride=2)
        y = block(block(y, 7), 10)
        y = F.max_pool2d(y, 2, stride=2)
    y = y.transpose(1, 2)
    y = y.contiguous().view(y.shape[0], y.shape[1], -1)
    return y, x_len


def vgg_tower(y, P, pre, freq_only_second_pool):
    """2 x (conv3x3 + ReLU) -> maxpool 2x2 -> 2 x (conv3x3 + ReLU) -> maxpool 2x2 or (1,2) (floor mode) on (B,C,T,F)."""
    for i in (0, 2):
        y = F.relu(F.conv2d(y, P[pre + '%d.weight' % i], P[pre + '%d.bias' % i], padding=1))
    y = F.max_pool2d(y, 2, stride=2)
    for i in (5, 7):
        y = F.relu(F.conv2d(y, P[pre + '%d.weight' % i], P[pre + '%d.bias' % i], padding=1))
    y = F.max_pool2d(y, (1, 2), stride=(1, 2)) if freq_only_second_pool else F.max_pool2d(y, 2, stride=2)
    y = y.transpose(1, 2)
    return y.contiguous().view(y.shape[0], y.shape[1], -1)


def vgg_extractor_variants(x, x_len, P, pre, cfg):
    """FreqVGGExtractor (vgg 2, src/module.py:746-841), VGGExtractor2 (vgg 3, :843-905), FreqVGGExtractor2 (vgg 4, :907-1001):
    time is cropped to a multiple of 4 (vgg 2) or 2 (vgg 3, 4) and the length divided likewise."""
    div = 4 if cfg.vgg == 2 else 2
    x_len = x_len // div
    if x.shape[1] % div != 0:
        x = x[:, :-(x.shape[1] % div), :].contiguous()
    B, T, _ = x.shape
    y = x.view(B, T, x.shape[-1] // 40, 40).transpose(1, 2)
    if cfg.vgg == 3:
        return vgg_tower(y, P, pre + 'extractor.', True), x_len
    sf = cfg.vgg_freq
    lo = vgg_tower(y[:, :, :, :sf], P, pre + 'low_extractor.', cfg.vgg == 4)
    hi = vgg_tower(y[:, :, :, sf:], P, pre + 'high_extractor.', cfg.vgg == 4)
    return torch.cat((lo, hi), dim=-1), x_len


def encoder(x, x_len, P, cfg, drop_masks=None, lstm_impl=bilstm, return_all=False):
    """Encoder.forward (src/asr.py:459-464). drop_masks: list (one per RNN layer) or None."""
    li = 0
    acts = []
    if cfg.vgg in (1, 5):
        x, x_len = vgg_extractor(x, x_len, P, 'encoder.layers.0.', cfg.vgg == 5)
        li = 1
        acts.append(x)
    elif cfg.vgg in (2, 3, 4):
        x, x_len = vgg_extractor_variants(x, x_len, P, 'encoder.layers.0.', cfg)
        li = 1
        acts.append(x)
    elif cfg.vgg == 7:
        x = x @ P['encoder.layers.0.dense.weight'].t() + P['encoder.layers.0.dense.bias']
        li = 1
        acts.append(x)
    elif cfg.vgg == 6:
        x_len = x_len // 4                     # Downsampler, src/module.py:725-729
        x = x[:, ::4, :]
        li = 1
        acts.append(x)
    for l in range(len(cfg.enc_dim)):
        m = None if drop_masks is None else drop_masks[l]
        x, x_len = rnn_layer(x, x_len, P, 'encoder.layers.%d.' % (li + l), cfg, l, m, lstm_impl)
        acts.append(x)
    if return_all:
        return x, x_len, acts
    return x, x_len


def ctc_head(enc, P):
    """log_softmax(ReLU(Linear(enc))) — src/asr.py:29-32,116-120 (the ReLU on logits is as written)."""
    return F.log_softmax(F.relu(enc @ P['ctc_layer.0.weight'].t() + P['ctc_layer.0.bias']), dim=-1)


# -------------------------------------------------------------------------------------------------
# attention decoder  (src/asr.py:123-175,227-266,333-364; src/module.py:1101-1117,1152-1173)
# -------------------------------------------------------------------------------------------------
def attention_keys(enc, P):
    return torch.tanh(enc @ P['attention.proj_k.weight'].t() + P['attention.proj_k.bias'])


def loc_attention_step(q_in, key, value, prev_att, mask, P, cfg):
    """One LocationAwareAttention step. q_in: concat of decoder h (B, dim*layers)."""
    query = torch.tanh(q_in @ P['attention.proj_q.weight'].t() + P['attention.proj_q.bias'])
    conv = F.conv1d(prev_att.unsqueeze(1), P['attention.att_layer.loc_conv.weight'], padding=cfg.loc_kernel_size)
    loc = torch.tanh(conv.transpose(1, 2) @ P['attention.att_layer.loc_proj.weight'].t())
    u = torch.tanh(key + query.unsqueeze(1) + loc)
    energy = (u @ P['attention.att_layer.gen_energy.weight'].t()).squeeze(-1) + P['attention.att_layer.gen_energy.bias']
    e = (energy / cfg.att_temperature).masked_fill(mask, NEG_INF)
    attn = torch.softmax(e, dim=-1)
    ctx = torch.bmm(attn.unsqueeze(1), value).squeeze(1)
    return attn, ctx


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i, f, gg, o = g.chunk(4, dim=-1)
    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h = torch.sigmoid(o) * torch.tanh(c)
    return h, c


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    xr, xz, xn = (x @ w_ih.t() + b_ih).chunk(3, dim=-1)
    hr, hz, hn = (h @ w_hh.t() + b_hh).chunk(3, dim=-1)
    r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
    return (1 - z) * torch.tanh(xn + r * hn) + z * h


def att_decoder_variants(enc, enc_len, P, cfg, decode_step, teacher=None, masks=None):
    """The reference's step loop for every attention / decoder variant (src/asr.py:124-175, Attention.forward :331-364,
    ScaleDotAttention / LocationAwareAttention src/module.py:1121-1189, Decoder.forward src/asr.py:262-270), teacher-forced
    or greedy.  masks: None (eval) or a dict of {0,1} float tensors - 'emb' (B,L,dec_dim), 'layer' [t][l] (B,dec_dim) for the
    inter-layer dropout of nn.LSTM / nn.GRU, 'final' [t] (B,dec_dim).  Returns logits (B,L,V), att (B,NH,L,T')."""
    B, Tp, Dv = enc.shape
    dim, nl, nh, ad = cfg.dec_dim, cfg.dec_layer, cfg.num_head, cfg.att_dim
    lstm = cfg.dec_module == 'LSTM'
    h = [enc.new_zeros(B, dim) for _ in range(nl)]
    c = [enc.new_zeros(B, dim) for _ in range(nl)]
    ar = torch.arange(Tp)[None, :]
    mask = (ar >= enc_len[:, None])[:, None, :].expand(B, nh, Tp).reshape(B * nh, Tp)          # rows b * nh + head (:1107)
    key = torch.tanh(enc @ P['attention.proj_k.weight'].t() + P['attention.proj_k.bias'])
    value = torch.tanh(enc @ P['attention.proj_v.weight'].t() + P['attention.proj_v.bias']) if cfg.v_proj else enc
    if nh > 1:
        key = key.view(B, Tp, nh, ad).permute(0, 2, 1, 3).reshape(B * nh, Tp, ad)
        if cfg.v_proj:
            value = value.view(B, Tp, nh, Dv).permute(0, 2, 1, 3).reshape(B * nh, Tp, Dv)
        else:
            value = value.repeat(nh, 1, 1)          # head-major rows against batch-major keys: as the reference has it (:354)
    prev_att = None
    if cfg.att_mode == 'loc':
        prev_att = torch.where(ar >= enc_len[:, None], torch.zeros(()), (1.0 / enc_len.float())[:, None].expand(B, Tp))
        prev_att = prev_att[:, None, :].expand(B, nh, Tp)
    E = P['pre_embed.weight']
    last = E[torch.zeros(B, dtype=torch.long)]
    temb = None
    if teacher is not None:
        temb = E[teacher]
        if masks is not None and cfg.emb_drop > 0:
            temb = temb * masks['emb'] / (1.0 - cfg.emb_drop)
    logits_seq, att_seq = [], []
    for t in range(decode_step):
        q_in = torch.cat(h, dim=-1)
        q = torch.tanh(q_in @ P['attention.proj_q.weight'].t() + P['attention.proj_q.bias']).view(B * nh, ad)
        if cfg.att_mode == 'dot':
            energy = torch.bmm(q.unsqueeze(1), key.transpose(1, 2)).squeeze(1)
        else:
            conv = F.conv1d(prev_att, P['attention.att_layer.loc_conv.weight'], padding=cfg.loc_kernel_size)
            loc = torch.tanh(conv.transpose(1, 2) @ P['attention.att_layer.loc_proj.weight'].t())          # (B,T,ad)
            loc = loc.unsqueeze(1).repeat(1, nh, 1, 1).view(B * nh, Tp, ad)
            u = torch.tanh(key + q.unsqueeze(1) + loc)
            energy = (u @ P['attention.att_layer.gen_energy.weight'].t()).squeeze(-1) + P['attention.att_layer.gen_energy.bias']
        attn = torch.softmax((energy / cfg.att_temperature).masked_fill(mask, NEG_INF), dim=-1)
        ctx = torch.bmm(attn.unsqueeze(1), value).squeeze(1)
        attn = attn.view(B, nh, Tp)
        if cfg.att_mode == 'loc':
            prev_att = attn
        if nh > 1:
            ctx = ctx.view(B, nh * Dv) @ P['attention.merge_head.weight'].t() + P['attention.merge_head.bias']
        x = torch.cat([last, ctx], dim=-1)
        for l in range(nl):
            W = [P['decoder.layers.%s_l%d' % (n, l)] for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            if lstm:
                h[l], c[l] = lstm_cell(x, h[l], c[l], *W)
            else:
                h[l] = gru_cell(x, h[l], *W)
            x = h[l]
            if l + 1 < nl and masks is not None and cfg.dec_dropout > 0:
                x = x * masks['layer'][t][l] / (1.0 - cfg.dec_dropout)
        if masks is not None and cfg.dec_dropout > 0:
            x = x * masks['final'][t] / (1.0 - cfg.dec_dropout)
        logit = x @ P['decoder.char_trans.weight'].t() + P['decoder.char_trans.bias']
        last = temb[:, t] if teacher is not None else E[logit.argmax(dim=-1)]
        logits_seq.append(logit)
        att_seq.append(attn)
    return torch.stack(logits_seq, dim=1), torch.stack(att_seq, dim=2)


def is_variant(cfg):
    return (cfg.att_mode != 'loc' or cfg.num_head != 1 or cfg.v_proj or cfg.dec_module != 'LSTM' or cfg.dec_dropout > 0 or cfg.emb_drop > 0)


def att_decoder(enc, enc_len, P, cfg, decode_step, teacher=None, state=None, return_state=False):
    """Teacher-forced (tf_rate=1) or greedy attention decoding; returns logits (B,L,V), att (B,1,L,T')."""
    B, Tp, _ = enc.shape
    dim, nl = cfg.dec_dim, cfg.dec_layer
    h = [enc.new_zeros(B, dim) for _ in range(nl)]
    c = [enc.new_zeros(B, dim) for _ in range(nl)]
    ar = torch.arange(Tp)[None, :]
    mask = ar >= enc_len[:, None]
    key = attention_keys(enc, P)
    prev_att = torch.where(mask, torch.zeros(()), (1.0 / enc_len.float())[:, None].expand(B, Tp))
    E = P['pre_embed.weight']
    last = E[torch.zeros(B, dtype=torch.long)]
    logits_seq, att_seq = [], []
    for t in range(decode_step):
        q_in = torch.cat(h, dim=-1)
        attn, ctx = loc_attention_step(q_in, key, enc, prev_att, mask, P, cfg)
        prev_att = attn
        x = torch.cat([last, ctx], dim=-1)
        for l in range(nl):
            h[l], c[l] = lstm_cell(x, h[l], c[l], P['decoder.layers.weight_ih_l%d' % l], P['decoder.layers.weight_hh_l%d' % l],
                                   P['decoder.layers.bias_ih_l%d' % l], P['decoder.layers.bias_hh_l%d' % l])
            x = h[l]
        logit = x @ P['decoder.char_trans.weight'].t() + P['decoder.char_trans.bias']
        if teacher is not None:
            last = E[teacher[:, t]]
        else:
            last = E[logit.argmax(dim=-1)]
        logits_seq.append(logit)
        att_seq.append(attn)
    out = torch.stack(logits_seq, dim=1)
    att = torch.stack(att_seq, dim=1).unsqueeze(1)
    return out, att


# -------------------------------------------------------------------------------------------------
# losses  (bin/train_asr.py:130-135,229-248; src/util.py:11-25)
# -------------------------------------------------------------------------------------------------
def ctc_loss_aten(logp_btv, txt, enc_len, txt_len):
    return F.ctc_loss(logp_btv.transpose(0, 1), txt, enc_len, txt_len, blank=0, reduction='mean', zero_infinity=False)


def ctc_nll_restated(logp_tv, labels):
    """-log p(labels | x) for ONE utterance by the textbook alpha recursion (Graves 2006), float64.
    logp_tv: (T,V) numpy log-probs, labels: 1-D ints (no blanks).  Returns (nll, grad wrt logp in the
    folded form exp(logp) - posterior that torch returns)."""
    lp = np.asarray(logp_tv, dtype=np.float64)
    T, V = lp.shape
    L = len(labels)
    S = 2 * L + 1
    ext = np.zeros(S, dtype=np.int64)
    ext[1::2] = labels
    la = np.full((T, S), -np.inf)
    lb = np.full((T, S), -np.inf)
    la[0, 0] = lp[0, 0]
    if S > 1:
        la[0, 1] = lp[0, ext[1]]
    for t in range(1, T):
        for s in range(S):
            v = la[t - 1, s]
            if s >= 1:
                v = np.logaddexp(v, la[t - 1, s - 1])
            if s >= 2 and ext[s] != 0 and ext[s] != ext[s - 2]:
                v = np.logaddexp(v, la[t - 1, s - 2])
            la[t, s] = v + lp[t, ext[s]]
    ll = la[T - 1, S - 1]
    if S > 1:
        ll = np.logaddexp(ll, la[T - 1, S - 2])
    lb[T - 1, S - 1] = lp[T - 1, 0]
    if S > 1:
        lb[T - 1, S - 2] = lp[T - 1, ext[S - 2]]
    for t in range(T - 2, -1, -1):
        for s in range(S):
            v = lb[t + 1, s]
            if s + 1 < S:
                v = np.logaddexp(v, lb[t + 1, s + 1])
            if s + 2 < S and ext[s + 2] != 0 and ext[s + 2] != ext[s]:
                v = np.logaddexp(v, lb[t + 1, s + 2])
            lb[t, s] = v + lp[t, ext[s]]
    nll = -ll
    post = np.full((T, V), -np.inf)
    for t in range(T):
        for s in range(S):
            post[t, ext[s]] = np.logaddexp(post[t, ext[s]], la[t, s] + lb[t, s])
    with np.errstate(invalid='ignore', over='ignore'):
        grad = np.exp(lp) - np.exp(post + nll - lp)
    return nll, grad


def label_smoothing_loss(logits, target, classes=31, smoothing=0.1):
    """src/util.py:11-25: mean over ALL rows (pad rows included), classes hard-coded by the caller."""
    lp = F.log_softmax(logits, dim=-1)
    true = torch.full_like(lp, smoothing / (classes - 1))
    true.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    return torch.mean(torch.sum(-true * lp, dim=-1))


def seq_loss(att_logits, txt, label_smoothing):
    b, t, v = att_logits.shape
    if label_smoothing:
        return label_smoothing_loss(att_logits.reshape(b * t, v), txt.reshape(-1))
    return F.cross_entropy(att_logits.reshape(b * t, v), txt.reshape(-1), ignore_index=0)


def asr_forward(feat, feat_len, P, cfg, decode_step, teacher=None, drop_masks=None, lstm_impl=bilstm, dec_masks=None):
    """ASR.forward (src/asr.py:89-177) with tf_rate=1 (teacher given) or greedy decoding."""
    enc, enc_len = encoder(feat, feat_len, P, cfg, drop_masks, lstm_impl)
    ctc_out = ctc_head(enc, P) if cfg.enable_ctc else None
    att_out, att_seq = (None, None)
    if cfg.enable_att:
        if is_variant(cfg):
            att_out, att_seq = att_decoder_variants(enc, enc_len, P, cfg, decode_step, teacher, dec_masks)
        else:
            att_out, att_seq = att_decoder(enc, enc_len, P, cfg, decode_step, teacher)
    return ctc_out, enc_len, att_out, att_seq


def asr_losses(feat, feat_len, txt, P, cfg, label_smoothing=False, drop_masks=None, lstm_impl=bilstm, dec_masks=None):
    """One training forward as the Solver wires it (bin/train_asr.py:204-248). Returns dict of tensors."""
    txt_len = (txt != 0).sum(dim=-1)
    L = int(txt_len.max())
    ctc_out, enc_len, att_out, att_seq = asr_forward(feat, feat_len, P, cfg, L, teacher=txt,
                                                     drop_masks=drop_masks, lstm_impl=lstm_impl, dec_masks=dec_masks)
    res = {'enc_len': enc_len, 'ctc_output': ctc_out, 'att_output': att_out, 'att_seq': att_seq}
    total = 0.0
    if ctc_out is not None:
        res['ctc_loss'] = ctc_loss_aten(ctc_out, txt, enc_len, txt_len)
        total = total + res['ctc_loss'] * cfg.ctc_weight
    if att_out is not None:
        res['att_loss'] = seq_loss(att_out, txt[:, :L], label_smoothing)
        total = total + res['att_loss'] * (1 - cfg.ctc_weight)
    res['total_loss'] = total
    return res


def clip_grad_norm(grads, max_norm=5.0):
    """torch.nn.utils.clip_grad_norm_ semantics (src/solver.py:97): returns (total_norm, scale)."""
    total = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    coef = max_norm / (total + 1e-6)
    return total, min(coef, 1.0)
