"""Kernel-level parity of the VGG pieces (implicit-GEMM conv3x3 fwd / dgrad / wgrad, max-pool, LayerNorm over
frequency, layout permutes) against torch on the CPU."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('prec', [0, 1])
@pytest.mark.parametrize('B,T,Fq,Ci,Co', [(2, 12, 8, 4, 16), (1, 9, 5, 3, 7), (2, 16, 40, 8, 32)])
def test_conv3x3_all_passes(B, T, Fq, Ci, Co, prec):
    from src import hipabi as H
    g = torch.Generator().manual_seed(B + T + Ci + Co)
    x = torch.randn(B, Ci, T, Fq, generator=g)
    w = torch.randn(Co, Ci, 3, 3, generator=g) / (Ci * 9) ** 0.5
    bias = torch.randn(Co, generator=g) * 0.1
    dy = torch.randn(B, Co, T, Fq, generator=g)
    if prec == 1:
        bf = lambda t: t.to(torch.bfloat16).float()
        xr, wr, dyr = bf(x), bf(w), bf(dy)
    else:
        xr, wr, dyr = x, w, dy
    xr = xr.clone().requires_grad_(True)
    wr = wr.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, bias, padding=1)
    st = H.stream_ptr()
    x_cl = x.permute(0, 2, 3, 1).contiguous().cuda()          # (B,T,F,Ci)
    wd, bd = w.cuda(), bias.cuda()
    wf = torch.empty(Co, 9 * Ci, device='cuda')
    H.call('asr_conv_weight_permute', H.ptr(wd), H.ptr(wf), Co, Ci, 0, st)
    y = torch.empty(B, T, Fq, Co, device='cuda')
    H.call('asr_conv3x3', H.ptr(x_cl), H.ptr(wf), H.ptr(y), H.ptr(bd), B, T, Fq, Ci, Co, 0, 0, 0, prec, st)
    tol = 1e-4 if prec == 0 else 2e-2
    assert (y.cpu() - y_ref.detach().permute(0, 2, 3, 1)).abs().max().item() < tol
    # gradients (for bf16 the reference uses bf16-rounded operands of each contraction separately)
    (F.conv2d(xr, wr, None, padding=1) * dyr).sum().backward()
    dy_cl = dy.permute(0, 2, 3, 1).contiguous().cuda()
    wdg = torch.empty(Ci, 9 * Co, device='cuda')
    H.call('asr_conv_weight_permute', H.ptr(wd), H.ptr(wdg), Co, Ci, 1, st)
    dx = torch.empty(B, T, Fq, Ci, device='cuda')
    H.call('asr_conv3x3', H.ptr(dy_cl), H.ptr(wdg), H.ptr(dx), None, B, T, Fq, Co, Ci, 0, 0, 0, prec, st)
    assert (dx.cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() < (2e-4 if prec == 0 else 5e-2)
    dwf = torch.zeros(Co, 9 * Ci, device='cuda')
    H.call('asr_conv3x3', H.ptr(x_cl), H.ptr(dy_cl), H.ptr(dwf), None, B, T, Fq, Ci, Co, 1, 0, 1, prec, st)
    dw = torch.zeros(Co, Ci, 3, 3, device='cuda')
    H.call('asr_conv_weight_permute', H.ptr(dwf), H.ptr(dw), Co, Ci, 2, st)
    scale = wr.grad.abs().max().item()
    assert (dw.cpu() - wr.grad).abs().max().item() < (2e-4 if prec == 0 else 3e-2) * max(scale, 1.0)


@pytest.mark.parametrize('ceil', [False, True])
def test_maxpool_and_layernorm_freq(ceil):
    from src import hipabi as H
    g = torch.Generator().manual_seed(5)
    B, T, Fq, C = 2, 9, 10, 6
    x = torch.randn(B, C, T, Fq, generator=g, requires_grad=True)
    y_ref = F.max_pool2d(x, 2, stride=2, ceil_mode=ceil)
    T2, F2 = y_ref.shape[2], y_ref.shape[3]
    dy = torch.randn(B, C, T2, F2, generator=g)
    (y_ref * dy).sum().backward()
    st = H.stream_ptr()
    x_cl = x.detach().permute(0, 2, 3, 1).contiguous().cuda()
    y = torch.empty(B, T2, F2, C, device='cuda')
    idx = torch.empty(B, T2, F2, C, dtype=torch.uint8, device='cuda')
    H.call('asr_maxpool2x2_fwd', H.ptr(x_cl), H.ptr(y), H.ptr(idx), B, T, Fq, C, T2, F2, st)
    assert torch.equal(y.cpu(), y_ref.detach().permute(0, 2, 3, 1))
    dx = torch.empty(B, T, Fq, C, device='cuda')
    dyd = dy.permute(0, 2, 3, 1).contiguous().cuda()
    H.call('asr_maxpool2x2_bwd', H.ptr(dyd), H.ptr(idx), H.ptr(dx), B, T, Fq, C, T2, F2, st)
    assert torch.allclose(dx.cpu(), x.grad.permute(0, 2, 3, 1), atol=1e-6)
    # LayerNorm over F (+ReLU)
    xl = torch.randn(B, C, T, Fq, generator=g, requires_grad=True)
    w = (1 + 0.1 * torch.randn(Fq, generator=g)).requires_grad_(True)
    b = (0.1 * torch.randn(Fq, generator=g)).requires_grad_(True)
    yl = F.relu(F.layer_norm(xl, (Fq,), w, b))
    dyl = torch.randn(B, C, T, Fq, generator=g)
    (yl * dyl).sum().backward()
    xd = xl.detach().permute(0, 2, 3, 1).contiguous().cuda()
    out = torch.empty_like(xd)
    stats = torch.empty(B * T * C, 2, device='cuda')
    wdv, bdv = w.detach().cuda(), b.detach().cuda()
    H.call('asr_ln_freq_fwd', H.ptr(xd), H.ptr(wdv), H.ptr(bdv), H.ptr(out), H.ptr(stats), B * T, Fq, C, 1e-5, 1, st)
    assert torch.allclose(out.cpu(), yl.detach().permute(0, 2, 3, 1), atol=1e-5)
    dxl = torch.empty_like(xd)
    dw, db = torch.zeros(Fq, device='cuda'), torch.zeros(Fq, device='cuda')
    dyld = dyl.permute(0, 2, 3, 1).contiguous().cuda()
    H.call('asr_ln_freq_bwd', H.ptr(dyld), H.ptr(xd), H.ptr(wdv), H.ptr(bdv), H.ptr(stats), H.ptr(dxl), H.ptr(dw), H.ptr(db),
           B * T, Fq, C, 1, st)
    assert torch.allclose(dxl.cpu(), xl.grad.permute(0, 2, 3, 1), atol=2e-5)
    assert torch.allclose(dw.cpu(), w.grad, atol=1e-4) and torch.allclose(db.cpu(), b.grad, atol=1e-4)


def test_permute_last2():
    from src import hipabi as H
    x = torch.randn(7, 3, 5).cuda()
    y = torch.empty(7, 5, 3, device='cuda')
    H.call('asr_permute_last2', H.ptr(x), H.ptr(y), 7, 3, 5, H.stream_ptr())
    assert torch.equal(y, x.transpose(1, 2).contiguous())


@pytest.mark.parametrize('kind,B,T', [('vgg1', 2, 52), ('vgg5', 3, 40), ('vgg1', 1, 7), ('vgg5', 2, 203)])
def test_bf16_bordered_front_end_vs_torch(kind, B, T):
    """The bf16 front-end on zero-bordered images (csrc/vgg16.hip: implicit-GEMM convolutions on the direct-to-LDS kernel, the
    nine-tap weight gradient, bordered pooling / CNNLayerNorm) against the SAME layers in plain torch on the CPU (fp32):
    output and every parameter / input gradient.  Odd T (trimmed to a multiple of 4), ceil- and floor-mode pooling.
    Reference: src/module.py:582-716."""
    import torch.nn as nn
    from src import hipabi as H
    from src.vgg import VGGExtractor, VGGExtractor_LN, _VGG16Fn, vgg16_ok
    D = 160
    torch.manual_seed(7 + T)
    mod = (VGGExtractor if kind == 'vgg1' else VGGExtractor_LN)(D)
    for p in mod.parameters():
        if p.dim() == 1:
            p.data.uniform_(-0.2, 0.2).add_(1.0 if p.numel() in (40, 20) else 0.0)       # LayerNorm gains near 1, biases small
    ref = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    g = torch.Generator().manual_seed(B * 100 + T)
    x = torch.rand(B, T, D, generator=g)
    dy_full = None

    def torch_forward(xin):
        Tt = xin.shape[1] - xin.shape[1] % 4
        h = xin[:, :Tt].view(B, Tt, 4, 40).transpose(1, 2)                  # (B, C, T, F)
        ext = mod.extractor
        if kind == 'vgg1':
            h = ext(h)
        else:
            for m in ext:
                if isinstance(m, nn.Conv2d) or isinstance(m, (nn.ReLU, nn.MaxPool2d)):
                    h = m(h)
                else:                                                        # CNNLayerNorm: LayerNorm over F of (B, C, T, F)
                    h = m.layer_norm(h)
        h = h.transpose(1, 2)
        return h.contiguous().view(B, h.shape[1], -1)

    xr = x.clone().requires_grad_(True)
    y_ref = torch_forward(xr)
    dy = torch.randn(y_ref.shape, generator=g).to(torch.bfloat16).float()      # what the bf16 recurrent layer behind hands back
    (y_ref * dy).sum().backward()
    g_ref = {k: p.grad.detach().clone() for k, p in mod.named_parameters()}
    for p in mod.parameters():
        p.grad = None
    modc = mod.cuda()
    assert vgg16_ok(modc, H.BF16)
    from src.vgg import _VGGFn
    rep = {}
    for name, fn in (('bordered bf16', _VGG16Fn), ('round-2 path (fp32 images, bf16 operands)', _VGGFn)):
        for p in modc.parameters():
            p.grad = torch.zeros_like(p)
        anchor = torch.zeros(1, device='cuda', requires_grad=True)
        y = fn.apply(anchor, x.cuda(), modc, H.BF16)
        assert tuple(y.shape) == tuple(y_ref.shape)
        err = (y.float().cpu() - y_ref.detach()).abs().max().item()
        scale = y_ref.detach().abs().max().item()
        assert err < 3e-2 * max(1.0, scale), (name, err, scale)
        y.backward(dy.cuda().to(y.dtype))
        torch.cuda.synchronize()
        gmax = max(float(v.norm()) for v in g_ref.values())
        rows = []
        for k, p in modc.named_parameters():
            a, r = p.grad.double().cpu().reshape(-1), g_ref[k].double().reshape(-1)
            if float(r.norm()) < 1e-4 * gmax:
                # analytically zero (the bias of a convolution in front of a CNNLayerNorm): both sides are rounding noise
                rows.append((k, None, float(a.norm()) / gmax))
                continue
            cos = float((a * r).sum() / (a.norm() * r.norm() + 1e-30))
            rows.append((k, round(cos, 5), round(float(a.norm() / (r.norm() + 1e-30)), 4)))
        rep[name] = rows
    # The bound: the SURVEY 8d tolerance for bf16 compute (cosine >= 0.99) wherever the round-2 path (fp32 images, operands rounded
    # to bf16 when staged - the same number of roundings per contraction) meets it itself, and never worse than that path by more
    # than 0.008 (norm ratio: within 2e-2 of that path's): a dense random output gradient through four bf16 contractions leaves the FIRST layer's weight gradient (36
    # inputs, a sum over every pixel) near 0.987 on either path.
    new_rows, old_rows = rep['bordered bf16'], rep['round-2 path (fp32 images, bf16 operands)']
    bad = []
    for rn, ro in zip(new_rows, old_rows):
        if rn[1] is None:
            if rn[2] > 1e-3:
                bad.append((rn, ro))
        elif rn[1] < min(0.99, ro[1] - 0.008) or rn[1] < 0.98 or abs(rn[2] - 1) > max(4e-2, abs(ro[2] - 1) + 2e-2):
            bad.append((rn, ro))
    if os.environ.get('ASR_DUMP_DIR'):
        import json
        json.dump(rep, open(os.path.join(os.environ['ASR_DUMP_DIR'], 'vgg16_%s_%d_%d.json' % (kind, B, T)), 'w'), indent=1)
    assert y_ref is not None and not bad, bad
