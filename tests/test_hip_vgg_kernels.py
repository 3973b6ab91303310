"""Kernel-level parity of the VGG pieces (implicit-GEMM conv3x3 fwd / dgrad / wgrad, max-pool, LayerNorm over
frequency, layout permutes) against torch on the CPU."""
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
