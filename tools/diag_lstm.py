import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch
from src import hipabi as H
H._lib = None
lib = ctypes.CDLL(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'diag', 'libasr_hip_diag.so'))
for name, argtypes in H.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
for name, (rt, at) in H._RESTYPES.items():
    fn = getattr(lib, name); fn.argtypes = at; fn.restype = rt
H._lib = lib
B, T, Hd, ND = 16, 1200, 320, 2
g = torch.Generator().manual_seed(0)
gates = (torch.randn(B, T, ND, 4 * Hd, generator=g) * 0.5).cuda()
whh = (torch.randn(ND, 4 * Hd, Hd, generator=g) / Hd ** 0.5).cuda()
bhh = torch.zeros(ND * 4 * Hd).cuda()
y = torch.empty(B, T, ND * Hd).cuda(); c = torch.empty(B, T, ND, Hd).cuda()
nb = lib.asr_lstm_workspace_bytes(B, Hd, ND)
ws = torch.zeros(nb, dtype=torch.uint8).cuda()
for it in range(3):
    g2 = gates.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    H.call('asr_lstm_fwd', H.ptr(g2), H.ptr(whh), H.ptr(bhh), H.ptr(y), H.ptr(c), B, T, Hd, ND, 1, H.ptr(ws), nb, H.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    st = ws[:256].view(torch.int64).cpu().tolist()
    ms = e0.elapsed_time(e1)
    names = ['gather', 'lds-write+spill', 'store_out+load_xg issue', 'barrier', 'mfma', 'act+publish', 'copy', 'looptop']
    print('iter %d: %.2f ms total (%.2f us/step), abort=%d' % (it, ms, ms * 1e3 / T, st[0] & 0xffffffff))
    tot = sum(st[2:10])
    for k, nm in enumerate(names):
        print('   %-26s %8.2f us/step  (%4.1f%%)' % (nm, st[2 + k] * 0.01 / T, 100.0 * st[2 + k] / max(tot, 1)))
    print('   shader clock estimate: %.0f MHz' % (st[10] / max(tot, 1) * 100.0))
