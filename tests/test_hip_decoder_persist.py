"""The single-launch (persistent, state-resident) decoder forward must reproduce the per-step kernels' saved state:
same arithmetic up to summation order and the 1-ulp tag bits of the exchanged values (bf16 contraction mode)."""
import ctypes
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu


def _model(cfg_name):
    from src.asr import ASR
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'e2e-asr-pytorch_amd')
    config = yaml.safe_load(open(os.path.join(root, 'config', cfg_name)))
    torch.manual_seed(3)
    return ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()


@pytest.mark.parametrize('B,Tp,L', [(16, 600, 12), (3, 170, 9), (9, 333, 7)])
def test_persistent_forward_matches_step_kernels(B, Tp, L):
    from src import hipabi as H
    from src import functions as F
    model = _model('librispeech_asr.yaml')
    g = torch.Generator().manual_seed(B + Tp)
    E = 640
    enc = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
    enc_len = torch.randint(max(Tp // 3, 1), Tp + 1, (B,), generator=g)
    enc_len[0] = Tp
    enc_len = enc_len.cuda()
    teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
    d = F._dec_dims(model, B, Tp, L)
    assert int(H.lib().asr_att_decoder_fwd_work_bytes(ctypes.byref(d))) > 0, 'no persistent plan for the reference shape'
    _, st_p = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
    assert st_p.get('work') is not None
    torch.cuda.synchronize()
    assert int(st_p['work'][:4].view(torch.int32).item()) == 0, 'persistent decoder raised its abort word'
    orig = F._dec_state

    def no_work(*a, **k):
        s = orig(*a, **k)
        return s
    real = H.lib().asr_att_decoder_fwd_work_bytes
    try:
        H.lib().asr_att_decoder_fwd_work_bytes = lambda *_: 0
        _, st_s = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
    finally:
        H.lib().asr_att_decoder_fwd_work_bytes = real
    assert st_s.get('work') is None
    torch.cuda.synchronize()
    for name, tol in (('q', 2e-3), ('att', 2e-3), ('xin', 5e-3), ('hs', 5e-3), ('cs', 1e-2), ('gates', 5e-3), ('logits', 2e-2)):
        a, b_ = st_p[name].float().cpu(), st_s[name].float().cpu()
        assert torch.isfinite(a).all(), name
        err = (a - b_).abs().max().item()
        assert err < tol, '%s differs by %g' % (name, err)
    # conv only where frames are valid (tiles past the utterance are not computed identically)
    for bi in range(B):
        n = int(enc_len[bi])
        err = (st_p['conv'][bi, :, :, :n] - st_s['conv'][bi, :, :, :n]).abs().max().item()
        assert err < 2e-3, 'conv row %d differs by %g' % (bi, err)


# (B, T', L, whether the shape has a persistent plan; since round 3 every shape here has one: (8, 1000) and (64, 750) take the
# streamed-tile plan of csrc/decoder_stream.hip, the others the on-chip plan)
@pytest.mark.parametrize('B,Tp,L,tiles', [(16, 600, 10, True), (16, 577, 4, True), (16, 400, 5, True), (16, 300, 4, True), (5, 640, 1, True),
                                          (2, 230, 5, True), (8, 1000, 6, True), (3, 170, 9, True), (9, 333, 7, True), (4, 18, 5, True),
                                          (16, 150, 6, True), (16, 75, 3, True), (64, 750, 2, True)])
def test_persistent_backward_matches_step_kernels(B, Tp, L, tiles):
    """Gradients of the decoder (all parameters + encoder output) with the loop as one persistent launch vs the per-step
    kernels, from the same forward state.  Covers the compile-time tile size (40 frames) and run-time ones (8..28), a ragged
    last tile, one cluster per XCD and two, a single step, short outputs whose plan has more tiles than frames (T' = 150 / 75 / 18:
    tiles past T' carry weight rows only), and shapes without a plan."""
    from src import hipabi as H
    from src import functions as F
    model = _model('librispeech_asr.yaml')
    g = torch.Generator().manual_seed(7 * B + Tp)
    E = 640
    enc0 = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
    enc_len = torch.randint(max(Tp // 3, 1), Tp + 1, (B,), generator=g)
    enc_len[0] = Tp
    enc_len = enc_len.cuda()
    teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
    dlog = (torch.randn(B, L, 31, generator=g) * 0.1).cuda()
    names = [n for n, _ in model.named_parameters() if n.startswith(('decoder', 'attention', 'pre_embed'))]
    out = {}
    d = F._dec_dims(model, B, Tp, L)
    assert (int(H.lib().asr_att_decoder_bwd_persistent_tiles(ctypes.byref(d))) > 0) == tiles
    off = int(H.lib().asr_att_decoder_bwd_status_offset(ctypes.byref(d)))
    old = H.lib().asr_att_decoder_set_persistent(3)
    try:
        for mode in (3, 1):
            H.lib().asr_att_decoder_set_persistent(mode)
            model.zero_grad()
            enc = enc0.clone().requires_grad_(True)
            logits, _, _ = F.AttDecoderFn.apply(model._anchor, enc, enc_len, teacher, L, model, H.BF16)
            (logits * dlog).sum().backward()
            torch.cuda.synchronize()
            out[mode] = {n: p.grad.detach().clone() for n, p in model.named_parameters() if n in names}
            out[mode]['enc'] = enc.grad.detach().clone()
            if mode == 3 and tiles:
                assert int(model._last_dec_bwd_ws[off:off + 4].view(torch.int32).item()) == 0, 'persistent backward raised its abort word'
    finally:
        H.lib().asr_att_decoder_set_persistent(old)
    for n in out[3]:
        a, b_ = out[3][n].double(), out[1][n].double()
        assert torch.isfinite(a).all(), n
        if n.endswith('gen_energy.bias'):
            # analytically zero (softmax is shift invariant): sum_t (1 - sum attn_t) dot_t, i.e. fp32 rounding of the attention
            # rows times the size of dctx . ctx - both values are rounding noise whose sample changes with any rounding upstream
            # (seen 0.8e-3 .. 1.7e-3 over the forward variants of round 3)
            assert float(a.abs().max()) < 3e-3 and float(b_.abs().max()) < 3e-3
            continue
        rel = float((a - b_).norm() / (b_.norm() + 1e-12))
        assert rel < 2e-2, '%s: relative difference %g' % (n, rel)


def test_training_trajectory_persistent_vs_step_kernels():
    """Six optimizer steps of the full-size model (config/librispeech_asr.yaml, dropout off) at a shape with a persistent
    plan, once with both decoder loops as single launches and once with the per-step kernels: the loss trajectory and the
    final parameters must agree.  Exercises repeated launches (tag epochs, recycled exchange buffers) through the whole
    path: front of the model, both losses, clip, Adadelta."""
    from src import hipabi as H
    from src.asr import ASR
    from src.optim import Optimizer
    from src.step import train_step
    from src.synthetic import librispeech_shaped_batch
    from src.util import CTCLoss, CrossEntropyLoss
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'e2e-asr-pytorch_amd')
    config = yaml.safe_load(open(os.path.join(root, 'config', 'librispeech_asr.yaml')))
    mc = config['model']
    mc['encoder']['dropout'] = [0.0] * len(mc['encoder']['dropout'])
    mc['decoder']['dropout'] = 0.0
    B, T, L = 4, 1200, 24
    feat, feat_len, txt = librispeech_shaped_batch(B, T, 160, L, 31, seed=11, device='cuda')
    runs = {}
    old = H.lib().asr_att_decoder_set_persistent(3)
    try:
        for flags in (3, 0):
            H.lib().asr_att_decoder_set_persistent(flags)
            torch.manual_seed(0)
            model = ASR(160, 31, B, prec='bf16', seed=21, **mc).cuda().train()
            if flags == 3:
                d = __import__('src.functions', fromlist=['_dec_dims'])._dec_dims(model, B, T // 2, L)
                assert int(H.lib().asr_att_decoder_bwd_persistent_tiles(ctypes.byref(d))) > 0
                assert int(H.lib().asr_att_decoder_fwd_work_bytes(ctypes.byref(d))) > 0
            opt = Optimizer(model.parameters(), 'Adadelta', 1.0, 1e-8)
            ctc, att = CTCLoss(blank=0, zero_infinity=False), CrossEntropyLoss(ignore_index=0)
            losses = []
            for _ in range(6):
                out = train_step(model, opt, ctc, att, feat, feat_len, txt, L, tf_rate=1.0, clip=5.0)
                losses.append(float(out['total_loss']))
            runs[flags] = (losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).double().cpu())
    finally:
        H.lib().asr_att_decoder_set_persistent(old)
    la, lb = runs[3][0], runs[0][0]
    assert all(abs(x) < 1e4 and x == x for x in la), la
    assert la[-1] < la[0], 'loss did not go down: %s' % la
    for x, y in zip(la, lb):
        assert abs(x - y) <= 5e-3 * abs(y), 'loss trajectories differ: %s vs %s' % (la, lb)
    pa, pb = runs[3][1], runs[0][1]
    cos = float((pa * pb).sum() / (pa.norm() * pb.norm()))
    rel = float((pa - pb).norm() / pb.norm())
    assert cos > 0.99999 and rel < 2e-3, 'parameters after 6 steps: cosine %.7f, relative difference %.3g' % (cos, rel)
