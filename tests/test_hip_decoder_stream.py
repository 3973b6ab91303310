"""Streamed-tile plan of the persistent decoder (csrc/decoder_stream.hip): the shapes the LDS-resident plan cannot hold - the
reference's B = 8 batches of up to 3 400 frames (T' = 1 700, src/collect_batch.py:21-24), B = 16 up to T' = 1 225, BASELINE
config 5 (B = 64) - must reproduce the per-step kernels' saved state and gradients (same arithmetic up to summation order
and the tag bits of the exchanged values; bf16 contraction mode), and, forced onto small shapes, the LDS-resident plan's."""
import ctypes
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _model(cfg_name='librispeech_asr.yaml'):
    from src.asr import ASR
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'e2e-asr-pytorch_amd')
    config = yaml.safe_load(open(os.path.join(root, 'config', cfg_name)))
    torch.manual_seed(3)
    return ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()


def _inputs(B, Tp, L, E=640):
    g = torch.Generator().manual_seed(B + Tp)
    enc = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
    enc_len = torch.randint(max(Tp // 3, 1), Tp + 1, (B,), generator=g)
    enc_len[0] = Tp
    if B > 2:
        enc_len[1] = max(1, Tp // 7)                # a short utterance: most of its tiles hold no valid frame
    teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
    return enc, enc_len.cuda(), teacher


# (B, T', L, plan the shape takes by itself: 1 LDS-resident (forced to 2 here), 2 streamed)
FWD_SHAPES = [(16, 600, 5, 1), (3, 170, 6, 1), (8, 1300, 4, 2), (16, 1225, 3, 2), (8, 1700, 3, 2), (64, 200, 3, 2), (24, 333, 4, 1),
              (5, 77, 4, 1), (64, 1500, 2, 2)]


@pytest.mark.parametrize('B,Tp,L,native', FWD_SHAPES)
def test_streamed_forward_matches_step_kernels(B, Tp, L, native):
    from src import hipabi as H
    from src import functions as F
    model = _model()
    enc, enc_len, teacher = _inputs(B, Tp, L)
    d = F._dec_dims(model, B, Tp, L)
    old = H.lib().asr_att_decoder_set_persistent(3)
    try:
        assert int(H.lib().asr_att_decoder_fwd_plan(ctypes.byref(d))) == native
        H.lib().asr_att_decoder_set_persistent(3 | 4)             # prefer the streamed plan
        assert int(H.lib().asr_att_decoder_fwd_plan(ctypes.byref(d))) == 2
        _, st_p = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
        assert st_p.get('work') is not None
        torch.cuda.synchronize()
        assert int(st_p['work'][:4].view(torch.int32).item()) == 0, 'streamed decoder raised its abort word'
        H.lib().asr_att_decoder_set_persistent(2)                 # forward on the per-step kernels
        _, st_s = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
        torch.cuda.synchronize()
    finally:
        H.lib().asr_att_decoder_set_persistent(old)
    for name, tol in (('q', 2e-3), ('att', 2e-3), ('xin', 5e-3), ('hs', 5e-3), ('cs', 1e-2), ('gates', 5e-3), ('logits', 2e-2)):
        a, b_ = st_p[name].float().cpu(), st_s[name].float().cpu()
        assert torch.isfinite(a).all(), name
        err = (a - b_).abs().max().item()
        assert err < tol, '%s differs by %g' % (name, err)
    att = st_p['att'].float().cpu()
    for bi in range(B):
        n = int(enc_len[bi])
        assert float(att[bi, :, n:].abs().max()) == 0.0 if n < Tp else True       # exactly 0 past enc_len
        assert float((att[bi].sum(-1) - 1).abs().max()) < 1e-4
        err = (st_p['conv'][bi, :, :, :n] - st_s['conv'][bi, :, :, :n]).abs().max().item()
        assert err < 2e-3, 'conv row %d differs by %g' % (bi, err)


BWD_SHAPES = [(16, 600, 4, 1), (3, 170, 5, 1), (8, 1300, 3, 2), (16, 1225, 2, 2), (8, 1700, 2, 2), (64, 200, 3, 2), (24, 333, 3, 2),
              (5, 77, 4, 1), (2, 18, 3, 1), (64, 1500, 2, 2)]


@pytest.mark.parametrize('B,Tp,L,native', BWD_SHAPES)
def test_streamed_backward_matches_step_kernels(B, Tp, L, native):
    """Gradients of the decoder (all parameters + encoder output) with the backward loop as ONE streamed-tile launch vs the
    per-step kernels, from the same (streamed) forward state: 48-frame groups with a ragged last group, tiles past the
    utterance, one to eight clusters per XCD, more weight rows per workgroup than registers (B = 64)."""
    from src import hipabi as H
    from src import functions as F
    model = _model()
    enc0, enc_len, teacher = _inputs(B, Tp, L)
    g = torch.Generator().manual_seed(11 * B + Tp)
    dlog = (torch.randn(B, L, 31, generator=g) * 0.1).cuda()
    names = [n for n, _ in model.named_parameters() if n.startswith(('decoder', 'attention', 'pre_embed'))]
    d = F._dec_dims(model, B, Tp, L)
    out = {}
    old = H.lib().asr_att_decoder_set_persistent(3)
    try:
        assert int(H.lib().asr_att_decoder_bwd_plan(ctypes.byref(d))) == native
        off = int(H.lib().asr_att_decoder_bwd_status_offset(ctypes.byref(d)))
        for mode in (3 | 4 | 8, 1 | 4):
            H.lib().asr_att_decoder_set_persistent(mode)
            assert int(H.lib().asr_att_decoder_bwd_plan(ctypes.byref(d))) == 2 or mode == 5
            model.zero_grad()
            enc = enc0.clone().requires_grad_(True)
            logits, _, _ = F.AttDecoderFn.apply(model._anchor, enc, enc_len, teacher, L, model, H.BF16)
            (logits * dlog).sum().backward()
            torch.cuda.synchronize()
            out[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if n in names}
            out[mode]['enc'] = enc.grad.detach().clone()
            if mode != 5:
                assert int(model._last_dec_bwd_ws[off:off + 4].view(torch.int32).item()) == 0, 'streamed backward raised its abort word'
    finally:
        H.lib().asr_att_decoder_set_persistent(old)
    for n in out[15]:
        a, b_ = out[15][n].double(), out[5][n].double()
        assert torch.isfinite(a).all(), n
        if n.endswith('gen_energy.bias'):
            assert float(a.abs().max()) < 3e-3 and float(b_.abs().max()) < 3e-3      # analytically zero: rounding noise on both sides (see test_hip_decoder_persist.py)
            continue
        rel = float((a - b_).norm() / (b_.norm() + 1e-12))
        assert rel < 2e-2, '%s: relative difference %g' % (n, rel)
