"""Times asr_lstm_fwd / asr_lstm_bwd (bf16, B=16, T=1200, H=320, 2 directions) through the C ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch
from src import hipabi as H
B, T, Hd, ND = 16, 1200, 320, 2
g = torch.Generator().manual_seed(0)
gates = (torch.randn(B, T, ND, 4 * Hd, generator=g) * 0.5).cuda()
whh = (torch.randn(ND, 4 * Hd, Hd, generator=g) / Hd ** 0.5).cuda()
bhh = torch.zeros(ND * 4 * Hd).cuda()
dy = (torch.randn(B, T, ND * Hd, generator=g) * 0.1).cuda()
y = torch.empty(B, T, ND * Hd).cuda(); c = torch.empty(B, T, ND, Hd).cuda()
nb = H.lib().asr_lstm_workspace_bytes(B, Hd, ND)
ws = torch.zeros(nb, dtype=torch.uint8).cuda()
for it in range(4):
    g2 = gates.clone()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    H.call('asr_lstm_fwd', H.ptr(g2), H.ptr(whh), H.ptr(bhh), H.ptr(y), H.ptr(c), B, T, Hd, ND, 1, H.ptr(ws), nb, H.stream_ptr())
    e[1].record()
    H.call('asr_lstm_bwd', H.ptr(g2), H.ptr(whh), H.ptr(dy), H.ptr(c), B, T, Hd, ND, 1, H.ptr(ws), nb, H.stream_ptr())
    e[2].record(); torch.cuda.synchronize()
    print('fwd %.3f us/step   bwd %.3f us/step   abort=%d' % (e[0].elapsed_time(e[1]) * 1e3 / T, e[1].elapsed_time(e[2]) * 1e3 / T,
                                                              int(ws[:4].view(torch.int32).item())))
