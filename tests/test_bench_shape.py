"""Oracle parity of the kernels the BENCH actually runs.

`bench.py` runs config/librispeech_asr.yaml at B=16, T=1200, L=180 in bf16 contraction mode with dropout on: that
shape takes the second-generation persistent LSTM recurrence (4 layers, forward + BPTT), the persistent decoder
forward (tile 40 x 15 workgroups per utterance, two clusters per XCD) and the persistent decoder backward - launches
no small fixture reaches.  Here one full training step at exactly that shape is compared with the CPU oracle
(oracle/asr_oracle.py, pinned to the reference by tests/test_oracle_golden.py) on the same seeded batch, weights and
dropout masks (the Philox masks are exported with asr_dropout_mask and fed to the oracle), and the test asserts that
the persistent plans were the ones taken.

Tolerances, bf16 contraction mode: losses rel 2e-2, ctc_output / att_output abs 5e-2 (att_seq abs 5e-2
on probabilities), per-parameter gradient cosine >= 0.99 for parameters with a non-negligible gradient.
"""
import ctypes
import os

import numpy as np
import pytest
import torch
import yaml

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')


def _setup(B, T, L, seed, train, vgg=0):
    from src.asr import ASR
    from src.synthetic import librispeech_shaped_batch
    mc = yaml.safe_load(open(CFG))['model']
    mc['encoder']['vgg'] = vgg
    D, V = 160, 31
    cfg = O.ModelCfg(mc, D, V)
    sd = O.seeded_state_dict(O.param_shapes(cfg), 11)
    model = ASR(D, V, B, prec='bf16', seed=5, **mc)
    model.load_state_dict(sd)
    model = model.cuda()
    model.train() if train else model.eval()
    feat, lens, txt = librispeech_shaped_batch(B, T, D, L, V, seed=seed)
    return mc, cfg, sd, model, feat, lens, txt


def _hip_step(model, feat, lens, txt):
    from src.util import CTCLoss, CrossEntropyLoss
    from src import hipabi as H
    feat, lens, txt = feat.cuda(), lens.cuda(), txt.cuda()
    txt_len = (txt != 0).sum(-1)
    L = int(txt.shape[1])
    model.zero_grad()
    model._drop_counter = 0
    ctc_out, enc_len, att_out, att_seq, _ = model(feat, lens, L, tf_rate=1.0, teacher=txt)
    ctc = CTCLoss()(ctc_out.transpose(0, 1), txt, enc_len, txt_len)
    att = CrossEntropyLoss()(att_out.view(-1, att_out.shape[-1]), txt.reshape(-1))
    total = 0.5 * ctc + 0.5 * att
    total.backward()
    H.raise_if_aborted()            # synchronises; a timed-out persistent launch must raise, not return numbers
    return {'ctc_output': ctc_out, 'att_output': att_out, 'att_seq': att_seq, 'ctc_loss': ctc, 'att_loss': att,
            'total_loss': total, 'enc_len': enc_len}


def _dropout_masks(model, cfg, B, T):
    """The masks the HIP step used: layer l's LSTM output has T_l frames (T, T, T/2, T/2 for rates 1,2,1,1; T = the frames
    that reach the first recurrent layer: a quarter of the input behind a VGG front-end)."""
    from src import hipabi as H
    masks, Tl = [], T
    rates = cfg.enc_sample_rate
    for l, p in enumerate(cfg.enc_dropout):
        seed = (model.seed * 1000003 + l + 1) & 0xFFFFFFFFFFFF
        width = (2 if cfg.bidirection else 1) * cfg.enc_dim[l]
        m = torch.empty(B * Tl * width, device='cuda')
        H.call('asr_dropout_mask', H.ptr(m), m.numel(), float(p), seed, H.stream_ptr())
        masks.append(m.view(B, Tl, width).cpu())
        Tl = Tl // rates[l] if rates[l] > 1 else Tl
    return masks


def _compare(model, res, ref, P, report):
    for key in ('ctc_output', 'att_output', 'att_seq'):
        got = res[key].detach().float().cpu().numpy()
        want = ref[key].detach().numpy()
        err = float(np.abs(got - want).max())
        report.append((key, err, 5e-2, err <= 5e-2))
    for key in ('ctc_loss', 'att_loss', 'total_loss'):
        r = float(ref[key])
        err = abs(float(res[key].detach()) - r)
        tol = 2e-2 * max(1.0, abs(r))
        report.append((key, err, tol, err <= tol))
    ref_g = {k: P[k].grad.numpy().astype(np.float64) for k in P}
    gmax = max(float(np.linalg.norm(g)) for g in ref_g.values())
    for k, p in model.named_parameters():
        g = p.grad.detach().cpu().numpy().astype(np.float64)
        r = ref_g[k]
        rn = np.linalg.norm(r)
        if rn > 1e-3 * gmax:
            cos = float((g * r).sum() / (np.linalg.norm(g) * rn + 1e-30))
            report.append(('gradcos.' + k, 1 - cos, 0.01, cos >= 0.99))
            ratio = float(np.linalg.norm(g) / rn)
            report.append(('gradnorm.' + k, abs(ratio - 1), 0.1, abs(ratio - 1) <= 0.1))


def _finish(report):
    bad = [r for r in report if not r[3]]
    msg = '\n'.join('%-60s err %.3e tol %.3e %s' % (n, e, t, 'ok' if ok else 'FAIL') for n, e, t, ok in report)
    assert not bad, '\n' + msg


def _plans(model, B, T, Tp, L):
    from src import functions as F_hip
    from src import hipabi as H
    d = F_hip._dec_dims(model, B, Tp, L)
    return {'lstm': [int(H.lib().asr_lstm_plan(B, t, 320, 2, H.BF16)) for t in (T, T, T // 2, T // 2)],
            'lstm16': int(H.lib().asr_lstm16_workspace_bytes(B, 320, 2, 0)) if H.fast16_enabled() else -1,
            'dec_fwd_work': int(H.lib().asr_att_decoder_fwd_work_bytes(ctypes.byref(d))),
            'dec_bwd_tiles': int(H.lib().asr_att_decoder_bwd_persistent_tiles(ctypes.byref(d)))}


def test_bench_shape_train_step_vs_oracle():
    """B=16, T=1200, L=180, dropout on: every persistent launch of the bench, against the oracle."""
    B, T, L = 16, 1200, 180
    mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=1234, train=True)
    plans = _plans(model, B, T, T // 2, L)
    assert min(plans['lstm']) >= 2 and plans['lstm16'] != 0, plans      # persistent recurrence; bf16-storage plan unless ASR_FAST16=0
    assert plans['dec_fwd_work'] > 0 and plans['dec_bwd_tiles'] > 0, plans
    res = _hip_step(model, feat, lens, txt)
    masks = _dropout_masks(model, cfg, B, T)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.asr_losses(feat, lens, txt, P, cfg, label_smoothing=False, drop_masks=masks, lstm_impl=O.bilstm_aten)
    ref['total_loss'].backward()
    assert np.array_equal(res['enc_len'].cpu().numpy(), ref['enc_len'].numpy())
    report = []
    _compare(model, res, ref, P, report)
    _finish(report)


@pytest.mark.parametrize('vgg', [1, 5])
def test_bench_shape_vgg_train_step_vs_oracle(vgg):
    """The vgg 1 (VGGExtractor) and vgg 5 (VGGExtractor_LN) bench lines (B=16 x T=1200 x L=100, T' = T/8 = 150) against the
    oracle: the implicit-GEMM convolutions at (16, C, 1200, 40), the recurrences at T/4 and T/8 frames, and the persistent
    decoder plans for short encoder outputs - 150 frames only get an on-chip backward plan through the frame-less tiles of
    round 2 (tiles past T' that carry weight rows only), which no small fixture reaches.
    Reference: src/module.py:582-716, src/asr.py:459-464."""
    from src import functions as F_hip
    from src import hipabi as H
    B, T, L = 16, 1200, 100
    mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=99 + vgg, train=True, vgg=vgg)
    d = F_hip._dec_dims(model, B, T // 8, L)
    assert int(H.lib().asr_att_decoder_fwd_plan(ctypes.byref(d))) == 1
    assert int(H.lib().asr_att_decoder_bwd_plan(ctypes.byref(d))) == 1, 'T\' = 150 must take the on-chip backward plan (frame-less tiles)'
    tiles = int(H.lib().asr_att_decoder_bwd_persistent_tiles(ctypes.byref(d)))
    assert tiles >= 15 and tiles * 8 <= 8 * 16, tiles            # 15+ workgroups per utterance for 150 frames: some carry weight rows only
    res = _hip_step(model, feat, lens, txt)
    masks = _dropout_masks(model, cfg, B, T // 4)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.asr_losses(feat, lens, txt, P, cfg, label_smoothing=False, drop_masks=masks, lstm_impl=O.bilstm_aten)
    ref['total_loss'].backward()
    assert np.array_equal(res['enc_len'].cpu().numpy(), ref['enc_len'].numpy())
    report = []
    _compare(model, res, ref, P, report)
    _finish(report)


def test_config5_like_batch64_vs_oracle():
    """BASELINE config 5 shape class (B=64, long utterances, SpecAugment-sized batches) at a length the CPU oracle
    finishes in seconds: B=64 puts 16 batch rows into every group of the batch-sliced recurrence (three polling loads
    per thread), the decoder runs its per-step kernels (64 clusters do not fit the chip), CTC has 2L+1 = 241 states."""
    B, T, L = 64, 640, 120
    mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=4321, train=True)
    plans = _plans(model, B, T, T // 2, L)
    assert plans['lstm16'] != 0, plans
    res = _hip_step(model, feat, lens, txt)
    masks = _dropout_masks(model, cfg, B, T)
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.asr_losses(feat, lens, txt, P, cfg, label_smoothing=False, drop_masks=masks, lstm_impl=O.bilstm_aten)
    ref['total_loss'].backward()
    report = []
    _compare(model, res, ref, P, report)
    _finish(report)


@pytest.mark.parametrize('B,T,L', [(2, 460, 40), (5, 800, 100)])
def test_persistent_decoder_backward_vs_oracle(B, T, L):
    """Smaller shapes that still have a persistent decoder backward plan (run-time tile sizes, ragged last tile,
    one cluster per XCD), eval mode, directly against the oracle."""
    mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=77 + B, train=False)
    plans = _plans(model, B, T, T // 2, L)
    assert plans['dec_bwd_tiles'] > 0 and plans['dec_fwd_work'] > 0, plans
    res = _hip_step(model, feat, lens, txt)
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = O.asr_losses(feat, lens, txt, P, cfg, label_smoothing=False, lstm_impl=O.bilstm_aten)
    ref['total_loss'].backward()
    report = []
    _compare(model, res, ref, P, report)
    _finish(report)


def test_step_repeatability():
    """Two runs of the same step (same weights, batch, Philox seed): the forward results are bitwise equal; the
    gradients are equal up to the summation order of the split reductions that add with fp32 atomics (weight-gradient
    contractions) - bounded here at 1e-5 of the gradient norm per parameter."""
    B, T, L = 16, 1200, 180
    mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=1234, train=True)
    runs = []
    for _ in range(2):
        res = _hip_step(model, feat, lens, txt)
        runs.append(({k: res[k].detach().clone() for k in ('ctc_output', 'att_output', 'att_seq', 'total_loss')},
                     model.flat_grad.clone()))
    for k in runs[0][0]:
        assert torch.equal(runs[0][0][k], runs[1][0][k]), k
    report = []
    gmax = max(float(runs[0][1][model._offsets[id(p)]:model._offsets[id(p)] + p.numel()].double().norm()) for p in model.parameters())
    for k, p in model.named_parameters():
        o, n = model._offsets[id(p)], p.numel()
        a, b = runs[0][1][o:o + n].double(), runs[1][1][o:o + n].double()
        # relative to the parameter's own gradient norm, floored at 1e-3 of the largest one (gen_energy.bias has a
        # mathematically zero gradient - a constant added to all energies leaves the softmax unchanged: pure rounding noise)
        err = float((a - b).norm() / max(float(a.norm()), 1e-3 * gmax))
        report.append(('repeat.' + k, err, 1e-5, err <= 1e-5))
    _finish(report)


def test_overlapped_parameter_gradients_equal_inline(monkeypatch):
    """The parameter gradients that run on the CU-masked side stream beside the BPTT (encoder layers, decoder) against
    the same step with everything in line (ASR_OVERLAP=0), at the bench shape through src.step.train_step (work stream, deferred
    launches, joins): forward results bitwise equal, gradients equal up to the order of the fp32 atomics (1e-5 of the norm), and
    the optimizer sees complete gradients (parameters after the step agree)."""
    from src import hipabi as H
    from src.optim import Optimizer
    from src.step import train_step
    from src.util import CTCLoss, CrossEntropyLoss
    B, T, L = 16, 1200, 180
    runs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('ASR_OVERLAP', mode)
        mc, cfg, sd, model, feat, lens, txt = _setup(B, T, L, seed=77, train=True)
        opt = Optimizer(model.parameters(), 'Adadelta', 1.0, 1e-8)
        feat, lens, txt = feat.cuda(), lens.cuda(), txt.cuda()
        model._drop_counter = 0
        out = train_step(model, opt, CTCLoss(), CrossEntropyLoss(), feat, lens, txt, L, tf_rate=1.0, clip=5.0, optimize=False)
        torch.cuda.synchronize()
        H.raise_if_aborted()
        grads = model.flat_grad.clone()
        opt.opt.step(clip=5.0, use_norm=True)
        torch.cuda.synchronize()
        runs[mode] = (float(out['total_loss']), out['att_output'].detach().clone(), grads, model.flat_param.clone(), model)
    assert runs['1'][0] == runs['0'][0]
    assert torch.equal(runs['1'][1], runs['0'][1])
    model = runs['1'][4]
    report = []
    gmax = max(float(runs['0'][2][model._offsets[id(p)]:model._offsets[id(p)] + p.numel()].double().norm()) for p in model.parameters())
    for k, p in model.named_parameters():
        o, n = model._offsets[id(p)], p.numel()
        a, b = runs['1'][2][o:o + n].double(), runs['0'][2][o:o + n].double()
        err = float((a - b).norm() / max(float(b.norm()), 1e-3 * gmax))
        report.append(('overlap.' + k, err, 1e-5, err <= 1e-5))
    _finish(report)
    assert float((runs['1'][3] - runs['0'][3]).abs().max()) < 1e-5
