#!/usr/bin/env python
"""Generates tests/golden/*.npz by running the GENUINE reference (imported read-only from
/root/reference) on seeded inputs and seeded weights.  Run in the build container only:

    python tests/golden/gen_golden.py

The reference never travels to the GPU box; only the .npz fixtures (inputs + expected outputs) and this
script are committed.  Weights are not stored: both sides rebuild them with
oracle.asr_oracle.seeded_state_dict(param_shapes(cfg), seed).
"""
import os
import sys
import types

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference'
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import asr_oracle as O  # noqa: E402
sys.path.insert(0, HERE)
from batchgen import make_batch  # noqa: E402

torch.set_num_threads(4)


def small_model_cfg(vgg=0, ctc_weight=0.5, dims=(16, 16), rates=(1, 2), dec_layer=1, dec_dim=12, drop=0.0,
                    layer_norm=False, style='drop', proj=None):
    n = len(dims)
    proj = [True] * n if proj is None else list(proj)
    return {
        'ctc_weight': ctc_weight,
        'encoder': {'vgg': vgg, 'vgg_freq': -1, 'vgg_low_filt': -1, 'module': 'LSTM', 'bidirection': True,
                    'dim': list(dims), 'dropout': [drop] * n, 'layer_norm': [layer_norm] * n, 'proj': proj,
                    'sample_rate': list(rates), 'sample_style': style},
        'attention': {'mode': 'loc', 'dim': 12, 'num_head': 1, 'v_proj': False, 'temperature': 0.5,
                      'loc_kernel_size': 3, 'loc_kernel_num': 4},
        'decoder': {'module': 'LSTM', 'dim': dec_dim, 'layer': dec_layer, 'dropout': 0},
    }


class MaskDrop(torch.nn.Module):
    """Stands in for nn.Dropout inside the reference instance so that the mask is known."""

    def __init__(self, mask, p):
        super().__init__()
        self.mask, self.p = mask, p

    def forward(self, x):
        return x * self.mask / (1.0 - self.p)


class MaskDropSeq(torch.nn.Module):
    """A dropout module that is called once per decoder step: the t-th call applies the t-th mask."""

    def __init__(self, masks, p):
        super().__init__()
        self.masks, self.p, self.t = masks, p, 0

    def forward(self, x):
        m = self.masks[self.t]
        self.t += 1
        return x * m / (1.0 - self.p)


def run_reference_model(model_cfg, D, V, seed, batch, label_smoothing=False, drop_masks=None, store_all=True, dec_masks=None):
    from src.asr import ASR
    from src.util import LabelSmoothingLoss
    cfg = O.ModelCfg(model_cfg, D, V)
    shapes = O.param_shapes(cfg)
    model = ASR(D, V, 4, **model_cfg)
    ref_sd = model.state_dict()
    assert list(ref_sd.keys()) == list(shapes.keys()), (list(ref_sd.keys()), list(shapes.keys()))
    for k in shapes:
        assert tuple(ref_sd[k].shape) == tuple(shapes[k]), (k, ref_sd[k].shape, shapes[k])
    model.load_state_dict(O.seeded_state_dict(shapes, seed))
    model.eval()
    if drop_masks is not None:
        li = 1 if cfg.vgg > 0 else 0
        for l, m in enumerate(drop_masks):
            model.encoder.layers[li + l].dp = MaskDrop(torch.from_numpy(m), cfg.enc_dropout[l])
    feat, lens, txt = [torch.from_numpy(x) for x in batch]
    txt_len = (txt != 0).sum(-1)
    L = int(txt_len.max())
    if dec_masks is not None:
        # embedding dropout (src/asr.py:36,128) and the decoder's final dropout (src/asr.py:214,268) with KNOWN masks
        model.embed_drop = MaskDrop(torch.from_numpy(dec_masks['emb']), cfg.emb_drop)
        model.decoder.final_dropout = MaskDropSeq([torch.from_numpy(m) for m in dec_masks['final']], cfg.dec_dropout)
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, lens, L, tf_rate=1.0, teacher=txt)
    if dec_masks is not None:
        model.embed_drop = torch.nn.Identity()
        model.decoder.final_dropout = torch.nn.Identity()
    out = {'enc_len': enc_len.numpy()}
    total = 0
    if ctc_out is not None:
        ctc = torch.nn.CTCLoss(blank=0, zero_infinity=False)(ctc_out.transpose(0, 1), txt, enc_len, txt_len)
        total = total + ctc * model.ctc_weight
        out['ctc_loss'] = ctc.item()
        out['ctc_output'] = ctc_out.detach().numpy()
    if att_out is not None:
        b, t, _ = att_out.shape
        crit = LabelSmoothingLoss(31, 0.1) if label_smoothing else torch.nn.CrossEntropyLoss(ignore_index=0)
        att = crit(att_out.view(b * t, -1), txt[:, :L].reshape(-1))
        total = total + att * (1 - model.ctc_weight)
        out['att_loss'] = att.item()
        out['att_output'] = att_out.detach().numpy()
        out['att_seq'] = att_seq.detach().numpy()
    out['total_loss'] = total.item()
    total.backward()
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
    out['grad_norm'] = float(gn)
    # note: clip_grad_norm_ rescaled .grad in place when gn > 5; undo so fixtures hold raw grads
    scale = min(1.0, 5.0 / (float(gn) + 1e-6))
    for k, p in model.named_parameters():
        g = p.grad / scale if p.grad is not None else torch.zeros_like(p)
        if store_all:
            out['grad.' + k] = g.numpy()
        else:
            out['gradnorm.' + k] = float(g.norm())
            out['gradhead.' + k] = g.reshape(-1)[:8].numpy().copy()
    # greedy decoding (validation path, bin/train_asr.py:337-341)
    with torch.no_grad():
        _, _, g_out, _, _ = model(feat, lens, int(L * 1.2))
    if g_out is not None:
        out['greedy_argmax'] = g_out.argmax(-1).numpy()
    return out


def save(name, meta, arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, meta=np.array(yaml.safe_dump(meta)), **arrays)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))


def gen_models():
    V = 31
    # G1: down-scaled librispeech_asr.yaml (vgg 0, 2 layers, rates 1,2), eval mode, all activations + grads
    mc = small_model_cfg()
    batch = make_batch(101, 3, 37, 20, 7, V)
    out = run_reference_model(mc, 20, V, 7, batch)
    save('g1_small_c2', {'model': mc, 'D': 20, 'V': V, 'wseed': 7, 'label_smoothing': False},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))

    # G1d: same with known dropout masks (pins dropout placement / scaling), 'concat' style not used here
    mc = small_model_cfg(drop=0.25)
    batch = make_batch(102, 3, 37, 20, 7, V)
    g = np.random.Generator(np.random.PCG64(55))
    masks = [(g.random((3, 37, 32)) >= 0.25).astype(np.float32), (g.random((3, 37, 32)) >= 0.25).astype(np.float32)]
    out = run_reference_model(mc, 20, V, 8, batch, drop_masks=masks)
    save('g1_small_c2_dropout', {'model': mc, 'D': 20, 'V': V, 'wseed': 8, 'label_smoothing': False},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], mask0=masks[0], mask1=masks[1], **out))

    # G3a: debug.yaml-like (vgg 6 downsampler, att-only, label smoothing, 2-layer decoder)
    mc = small_model_cfg(vgg=6, ctc_weight=0.0, dims=(16,), rates=(1,), dec_layer=2, dec_dim=12)
    batch = make_batch(103, 3, 43, 20, 6, V)
    out = run_reference_model(mc, 20, V, 9, batch, label_smoothing=True)
    save('g3_small_debug', {'model': mc, 'D': 20, 'V': V, 'wseed': 9, 'label_smoothing': True},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))

    # G3b/c: VGG front-ends (channel counts are fixed by the reference; keep T and the LSTM small)
    for vgg, name in ((1, 'g3_small_vgg1'), (5, 'g3_small_vgg5')):
        mc = small_model_cfg(vgg=vgg, dims=(16,), rates=(1,))
        batch = make_batch(104 + vgg, 2, 43, 80, 4, V)
        out = run_reference_model(mc, 80, V, 10 + vgg, batch, store_all=False)
        save(name, {'model': mc, 'D': 80, 'V': V, 'wseed': 10 + vgg, 'label_smoothing': False},
             dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))

    # G3d: layer-norm + concat downsampling variant of RNNLayer (the reference's `pj` is sized for the
    # un-concatenated width, src/module.py:1038, so concat with rate>1 only runs with proj off)
    mc = small_model_cfg(layer_norm=True, style='concat', proj=[True, False])
    batch = make_batch(110, 3, 37, 20, 7, V)
    out = run_reference_model(mc, 20, V, 21, batch)
    save('g3_small_ln_concat', {'model': mc, 'D': 20, 'V': V, 'wseed': 21, 'label_smoothing': False},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))

    # G2: full-size config/librispeech_asr.yaml, B=4, T=203, L=11 — summaries only
    mc = yaml.safe_load(open(os.path.join(REF, 'config', 'librispeech_asr.yaml')))['model']
    for e in mc['encoder']['dropout']:
        pass
    batch = make_batch(120, 4, 203, 160, 11, V)
    out = run_reference_model(mc, 160, V, 31, batch, store_all=False)
    keep = {k: v for k, v in out.items() if not k.endswith('_output') and k != 'att_seq'}
    keep['ctc_output_head'] = out['ctc_output'][:, :4, :]
    keep['att_output_head'] = out['att_output'][:, :4, :]
    keep['att_seq_head'] = out['att_seq'][:, :, :4, :]
    save('g2_full_c2', {'model': mc, 'D': 160, 'V': V, 'wseed': 31, 'label_smoothing': False},
         dict(feat_seed=120, feat_len=batch[1], txt=batch[2], **keep))


def gen_variants():
    """G10: the model variants of SURVEY 8 row f-4 run by the genuine reference (eval mode; dropout through known masks)."""
    V, D = 31, 20

    def cfg_of(att=None, dec=None, enc_module='LSTM', emb_drop=0.0):
        mc = small_model_cfg()
        mc['encoder']['module'] = enc_module
        mc['attention'].update(att or {})
        mc['decoder'].update(dec or {})
        if emb_drop:
            mc['emb_drop'] = emb_drop
        return mc
    cases = [
        ('g10_dot', cfg_of(att={'mode': 'dot'}), 71),
        ('g10_loc_mh_vproj', cfg_of(att={'num_head': 2, 'v_proj': True}), 72),
        ('g10_dot_mh', cfg_of(att={'mode': 'dot', 'num_head': 2}), 73),
        ('g10_gru', cfg_of(dec={'module': 'GRU', 'layer': 2}, enc_module='GRU'), 74),
    ]
    for i, (name, mc, seed) in enumerate(cases):
        batch = make_batch((160 + i) if name != 'g10_gru' else 101, 3, 37, D, 7, V)       # (batch 163 has an utterance CTC cannot align)
        out = run_reference_model(mc, D, V, seed, batch)
        assert np.isfinite(out['total_loss']), name
        save(name, {'model': mc, 'D': D, 'V': V, 'wseed': seed, 'label_smoothing': False},
             dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))
    # the other VGG front-ends: FreqVGGExtractor (vgg 2), VGGExtractor2 (vgg 3), FreqVGGExtractor2 (vgg 4)
    for vgg in (2, 3, 4):
        mc = small_model_cfg(vgg=vgg, dims=(16,), rates=(1,))
        mc['encoder'].update({'vgg_freq': 12, 'vgg_low_filt': 4})
        batch = make_batch(180 + vgg, 2, 43, 80, 4, V)
        out = run_reference_model(mc, 80, V, 80 + vgg, batch, store_all=False)
        save('g10_vgg%d' % vgg, {'model': mc, 'D': 80, 'V': V, 'wseed': 80 + vgg, 'label_smoothing': False},
             dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))
    # Featemb_Extractor (vgg 7, config/librispeech_asr_upstream.yaml: CTC only)
    mc = small_model_cfg(vgg=7, ctc_weight=1.0, dims=(16,), rates=(1,))
    batch = make_batch(101, 3, 37, D, 7, V)
    out = run_reference_model(mc, D, V, 87, batch)
    save('g10_vgg7', {'model': mc, 'D': D, 'V': V, 'wseed': 87, 'label_smoothing': False},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], **out))
    # decoder dropout + embedding dropout with known masks (one decoder layer: nn.LSTM's own inter-layer dropout cannot be pinned)
    mc = cfg_of(dec={'dropout': 0.25}, emb_drop=0.2)
    batch = make_batch(170, 3, 37, D, 7, V)
    L = int((batch[2] != 0).sum(-1).max())
    g = np.random.Generator(np.random.PCG64(77))
    dm = {'emb': (g.random((3, batch[2].shape[1], 12)) >= 0.2).astype(np.float32),
          'final': [(g.random((3, 12)) >= 0.25).astype(np.float32) for _ in range(L)]}
    out = run_reference_model(mc, D, V, 75, batch, dec_masks=dm)
    save('g10_decdrop', {'model': mc, 'D': D, 'V': V, 'wseed': 75, 'label_smoothing': False},
         dict(feat=batch[0], feat_len=batch[1], txt=batch[2], mask_emb=dm['emb'], mask_final=np.stack(dm['final'], 0), **out))


def gen_ctc():
    """G4: torch.nn.CTCLoss(blank=0, zero_infinity=False) — loss and folded gradient, incl. repeated labels
    and an infeasible alignment (inf loss, nan grad: SURVEY V5)."""
    g = np.random.Generator(np.random.PCG64(77))
    T, B, V, L = 12, 4, 6, 5
    logits = g.standard_normal((B, T, V)).astype(np.float32)
    txt = np.array([[2, 2, 3, 1, 0], [4, 1, 0, 0, 0], [3, 3, 3, 3, 1], [5, 4, 1, 0, 0]], dtype=np.int64)
    in_len = np.array([12, 9, 12, 7], dtype=np.int64)
    lp = torch.log_softmax(torch.from_numpy(logits), -1).requires_grad_(True)
    tl = torch.from_numpy((txt != 0).sum(-1))
    per = torch.nn.functional.ctc_loss(lp.transpose(0, 1), torch.from_numpy(txt), torch.from_numpy(in_len), tl,
                                       blank=0, reduction='none', zero_infinity=False)
    loss = torch.nn.CTCLoss(blank=0, zero_infinity=False)(lp.transpose(0, 1), torch.from_numpy(txt),
                                                          torch.from_numpy(in_len), tl)
    loss.backward()
    arrays = dict(logits=logits, txt=txt, in_len=in_len, nll=per.detach().numpy(), loss=loss.item(), grad=lp.grad.numpy())
    # infeasible: labels (2,2,<eos>) need >= 4 frames (blank between the repeat) but only 3 are given
    lp2 = torch.log_softmax(torch.from_numpy(logits[:1, :4]), -1).requires_grad_(True)
    txt2 = torch.tensor([[2, 2, 1]])
    l2 = torch.nn.CTCLoss(blank=0, zero_infinity=False)(lp2.transpose(0, 1), txt2, torch.tensor([3]), torch.tensor([3]))
    l2.backward()
    arrays.update(inf_loss=l2.item(), inf_grad_isnan=np.array(bool(torch.isnan(lp2.grad).any())))
    save('g4_ctc', {'V': V}, arrays)


def gen_decode():
    """G7/G8: BeamDecoder (src/decode.py) hypotheses and CTCPrefixScore (src/ctc.py) states."""
    from src.asr import ASR
    from src.ctc import CTCPrefixScore
    from src.decode import BeamDecoder
    from src.lm import RNNLM
    V, D = 31, 20
    mc = small_model_cfg()
    cfg = O.ModelCfg(mc, D, V)
    shapes = O.param_shapes(cfg)
    model = ASR(D, V, 4, **mc)
    model.load_state_dict(O.seeded_state_dict(shapes, 41))
    model.eval()
    g = np.random.Generator(np.random.PCG64(130))
    feat = g.random((1, 61, D), dtype=np.float32)
    flen = np.array([61], dtype=np.int64)
    # LM: small tied RNNLM with seeded weights
    lm_cfg = {'emb_tying': True, 'emb_dim': 16, 'module': 'LSTM', 'dim': 16, 'n_layers': 2, 'dropout': 0.0}
    lm = RNNLM(V, **lm_cfg)
    lm_shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
    lm_sd = O.seeded_state_dict(lm_shapes, 43)
    arrays = dict(feat=feat, feat_len=flen)
    for tag, ctc_w, lm_w in (('att', 0.0, 0.0), ('ctc', 0.3, 0.0), ('ctc_lm', 0.3, 0.5)):
        dec = BeamDecoder(model, None, beam_size=4, min_len_ratio=0.01, max_len_ratio=0.12, ctc_weight=ctc_w)
        if lm_w > 0:
            dec.apply_lm, dec.lm_w, dec.lm = True, lm_w, lm
            lm.load_state_dict(lm_sd)
            lm.eval()
        with torch.no_grad():
            hyps = dec(torch.from_numpy(feat), torch.from_numpy(flen))
        arrays['n_' + tag] = np.array(len(hyps))
        for i, h in enumerate(hyps):
            arrays['%s_seq%d' % (tag, i)] = np.array(h.outIndex, dtype=np.int64)
            arrays['%s_score%d' % (tag, i)] = np.array([float(s) for s in h.output_scores], dtype=np.float32)
            arrays['%s_avg%d' % (tag, i)] = np.array(float(h.avgScore()), dtype=np.float32)
    # prefix scorer on the model's CTC posteriors
    with torch.no_grad():
        enc, _ = model.encoder(torch.from_numpy(feat), torch.from_numpy(flen))
        lp = torch.log_softmax(model.ctc_layer(enc), -1)
    ps = CTCPrefixScore(lp)
    r0 = ps.init_state()
    cand = [1, 5, 7, 2, 9, 4]
    psi1, r1 = ps.cheap_compute([], r0, cand)
    psi2, r2 = ps.cheap_compute([7], r1[cand.index(7)], cand)
    psi3, r3 = ps.cheap_compute([7, 7], r2[cand.index(7)], [7, 1, 3])
    arrays.update(ctc_logp=lp.numpy()[0], r0=r0, cand=np.array(cand), psi1=psi1, r1=r1, psi2=psi2, r2=r2,
                  psi3=psi3, r3=r3)
    save('g7_decode', {'model': mc, 'D': D, 'V': V, 'wseed': 41, 'lm': lm_cfg, 'lm_wseed': 43,
                       'beam': 4, 'min_len_ratio': 0.01, 'max_len_ratio': 0.12}, arrays)



def gen_decode_config4():
    """G7b: BASELINE config 4 at ITS size (VERDICT r02 next #4b): config/librispeech_asr.yaml model (12 M parameters), beam 8,
    ctc_weight 0.3, the 4 x 1024 tied RNN-LM of config/librispeech_lm.yaml with weight 0.3, T = 400 frames, max_len_ratio 0.05
    (about a second on the CPU).  Stores seeds, the input and the reference's hypotheses + scores, not weights."""
    from src.asr import ASR
    from src.decode import BeamDecoder
    from src.lm import RNNLM
    V, D, T = 31, 160, 400
    mc = yaml.safe_load(open(os.path.join(REF, 'config', 'librispeech_asr.yaml')))['model']
    lm_cfg = yaml.safe_load(open(os.path.join(REF, 'config', 'librispeech_lm.yaml')))['model']
    cfg = O.ModelCfg(mc, D, V)
    model = ASR(D, V, 8, **mc)
    model.load_state_dict(O.seeded_state_dict(O.param_shapes(cfg), 51))
    model.eval()
    lm = RNNLM(V, **lm_cfg)
    lm_shapes = {k: tuple(v.shape) for k, v in lm.state_dict().items()}
    lm.load_state_dict(O.seeded_state_dict(lm_shapes, 53))
    lm.eval()
    g = np.random.Generator(np.random.PCG64(150))
    arrays = {}
    nutt = 3
    for u in range(nutt):
        Tu = T - 37 * u
        feat = g.random((1, Tu, D), dtype=np.float32)
        flen = np.array([Tu], dtype=np.int64)
        dec = BeamDecoder(model, None, beam_size=8, min_len_ratio=0.01, max_len_ratio=0.05, ctc_weight=0.3)
        dec.apply_lm, dec.lm_w, dec.lm = True, 0.3, lm
        with torch.no_grad():
            hyps = dec(torch.from_numpy(feat), torch.from_numpy(flen))
        arrays['feat%d' % u], arrays['feat_len%d' % u] = feat, flen
        arrays['n%d' % u] = np.array(len(hyps))
        for i, h in enumerate(hyps):
            arrays['u%d_seq%d' % (u, i)] = np.array(h.outIndex, dtype=np.int64)
            arrays['u%d_score%d' % (u, i)] = np.array([float(s) for s in h.output_scores], dtype=np.float32)
            arrays['u%d_avg%d' % (u, i)] = np.array(float(h.avgScore()), dtype=np.float32)
        print('utt', u, [(h.outIndex, round(float(h.avgScore()), 4)) for h in hyps])
    save('g7b_decode_config4', {'model': mc, 'D': D, 'V': V, 'wseed': 51, 'lm': lm_cfg, 'lm_wseed': 53, 'beam': 8, 'min_len_ratio': 0.01,
                                'max_len_ratio': 0.05, 'ctc_weight': 0.3, 'lm_weight': 0.3, 'nutt': nutt}, arrays)

def gen_frontend():
    """G5/G6: Delta / Postprocess / Augment / mel filterbank from src/audio.py (pure torch/numpy parts).
    torchaudio, audiomentations and librosa are not installed here; the classes below never touch them,
    so they are imported with inert placeholders for those three names.
    ExtractAudioFeature itself needs torchaudio arithmetic and is NOT covered: STFT/mel parity unpinned."""
    for name in ('torchaudio', 'torchaudio.transforms', 'audiomentations', 'librosa', 'librosa.feature', 'librosa.util'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == 'audiomentations':
                for c in ('Compose', 'AddGaussianNoise', 'TimeStretch', 'PitchShift', 'Shift', 'FrequencyMask', 'TimeMask'):
                    setattr(m, c, object)
            sys.modules[name] = m
            if '.' in name:
                setattr(sys.modules[name.split('.')[0]], name.split('.')[1], m)
    import random
    import src.audio as A
    g = np.random.Generator(np.random.PCG64(140))
    mel = g.random((1, 80, 57), dtype=np.float32)            # (C=1, F, T) as ExtractAudioFeature returns it
    arrays = dict(mel=mel)
    for order in (1, 2):
        d = A.Delta(order, 2)(torch.from_numpy(mel))
        arrays['delta%d' % order] = d.numpy()
        arrays['post%d' % order] = A.Postprocess()(d).numpy()
        arrays['filters%d' % order] = A.Delta(order, 2).filters.numpy()
    fb = A.create_mel_filterbank(16000, 1025, n_mels=80)
    arrays['melfb'] = fb.astype(np.float32)
    # SpecAugment with recorded draws: replay torch.randint / random.randrange streams
    aug = A.Augment()
    x0 = torch.from_numpy(g.random((123, 160), dtype=np.float32))
    draws = []
    for trial in range(6):
        torch.manual_seed(1000 + trial)
        random.seed(2000 + trial)
        x = x0.clone()
        y = aug(x)
        # recover the draws by replaying the same streams
        torch.manual_seed(1000 + trial)
        random.seed(2000 + trial)
        T = x0.shape[0]
        t = torch.randint(0, 40, (1,)).item()
        t0 = torch.randint(0, T - t, (1,)).item()
        tend = torch.randint(t0, t0 + t, (1,)).item() if t > 0 else t0
        f = random.randrange(0, 27)
        f0 = random.randrange(0, 160 - f)
        fend = random.randrange(f0, f0 + f) if f > 0 else f0
        draws.append([t, t0, tend, f, f0, fend])
        arrays['aug_out%d' % trial] = y.numpy().copy()
    arrays['aug_in'] = x0.numpy()
    arrays['aug_draws'] = np.array(draws, dtype=np.int64)
    save('g5_frontend', {}, arrays)


def gen_ckpt():
    """G9: a checkpoint WRITTEN BY THE REFERENCE's classes in the reference's layout (src/solver.py:176-189:
    {model, optimizer, global_step, <metric>: score}) after two optimizer steps of the reference's training step
    (bin/train_asr.py:229-253 + src/solver.py:88-106: losses, backward, clip 5.0, Adadelta via src/optim.Optimizer),
    plus what a third step and a beam decode give from that state - the interop targets of tests/test_checkpoint_interop.py."""
    from src.asr import ASR
    from src.decode import BeamDecoder
    from src.optim import Optimizer
    V, D = 31, 20
    mc = small_model_cfg()
    cfg = O.ModelCfg(mc, D, V)
    model = ASR(D, V, 4, **mc)
    model.load_state_dict(O.seeded_state_dict(O.param_shapes(cfg), 51))
    model.eval()                                   # dropout is 0 in this config; eval keeps the fixture deterministic
    opt = Optimizer(model.parameters(), optimizer='Adadelta', lr=1.0, eps=1e-8, lr_scheduler='fixed', tf_start=1, tf_end=1, tf_step=1)
    batch = make_batch(150, 3, 37, D, 7, V)
    feat, lens, txt = [torch.from_numpy(x) for x in batch]
    txt_len = (txt != 0).sum(-1)
    L = int(txt_len.max())
    ctc_crit, att_crit = torch.nn.CTCLoss(blank=0, zero_infinity=False), torch.nn.CrossEntropyLoss(ignore_index=0)

    def step():
        opt.pre_step(0)
        ctc_out, enc_len, att_out, _, _ = model(feat, lens, L, tf_rate=1.0, teacher=txt)
        loss = 0.5 * ctc_crit(ctc_out.transpose(0, 1), txt, enc_len, txt_len) + \
            0.5 * att_crit(att_out.view(-1, V), txt[:, :L].reshape(-1))
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 5.0)
        opt.step()
        return float(loss), float(gn)
    losses = [step(), step()]
    path = os.path.join(HERE, 'g9_ref_ckpt.pth')
    torch.save({'model': model.state_dict(), 'optimizer': opt.get_opt_state_dict(), 'global_step': 2, 'wer': 0.4375}, path)
    print('wrote %s (%.1f KB)' % (path, os.path.getsize(path) / 1024.0))
    arrays = dict(feat=batch[0], feat_len=batch[1], txt=batch[2], loss0=losses[0][0], loss1=losses[1][0])
    # beam decode (beam 4, CTC 0.3) of the first utterance from the saved state
    dec = BeamDecoder(model, None, beam_size=4, min_len_ratio=0.01, max_len_ratio=0.2, ctc_weight=0.3)
    with torch.no_grad():
        hyps = dec(feat[:1, :int(lens[0])], lens[:1])
    arrays['n_hyp'] = np.array(len(hyps))
    for i, h in enumerate(hyps):
        arrays['hyp_seq%d' % i] = np.array(h.outIndex, dtype=np.int64)
        arrays['hyp_avg%d' % i] = np.array(float(h.avgScore()), dtype=np.float32)
    # third step from the saved state (resume target): loss, grad norm and every parameter after it
    l3, g3 = step()
    arrays['loss2'], arrays['gradnorm2'] = l3, g3
    for k, v in model.state_dict().items():
        arrays['after.' + k] = v.numpy().copy()
    save('g9_ckpt', {'model': mc, 'D': D, 'V': V, 'wseed': 51, 'beam': 4, 'min_len_ratio': 0.01, 'max_len_ratio': 0.2,
                     'ctc_weight': 0.3}, arrays)


if __name__ == '__main__':
    which = sys.argv[1:] or ['models', 'variants', 'ctc', 'decode', 'decode4', 'frontend', 'ckpt']
    if 'ckpt' in which:
        gen_ckpt()
    if 'models' in which:
        gen_models()
    if 'variants' in which:
        gen_variants()
    if 'ctc' in which:
        gen_ctc()
    if 'decode' in which:
        gen_decode()
    if 'decode4' in which:
        gen_decode_config4()
    if 'frontend' in which:
        gen_frontend()
