"""Phase timers of the persistent LSTM kernels (needs `make -C e2e-asr-pytorch_amd/csrc diag`).
usage: python tools/diag_lstm.py [mode]   mode 1 = second-generation kernels (default), 2 = first generation"""
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
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
lib.asr_lstm_set_persistent(mode)
B, T, Hd, ND = 16, 1200, 320, 2
g = torch.Generator().manual_seed(0)
gates = (torch.randn(B, T, ND, 4 * Hd, generator=g) * 0.5).cuda()
whh = (torch.randn(ND, 4 * Hd, Hd, generator=g) / Hd ** 0.5).cuda()
bhh = torch.zeros(ND * 4 * Hd).cuda()
dy = (torch.randn(B, T, ND * Hd, generator=g) * 0.1).cuda()
y = torch.empty(B, T, ND * Hd).cuda(); c = torch.empty(B, T, ND, Hd).cuda()
nb = lib.asr_lstm_workspace_bytes(B, Hd, ND)
ws = torch.zeros(nb, dtype=torch.uint8).cuda()
NAMES = {
    ('fwd', 1): ['C: wait tile (barrier1)', 'C: mfma+act', 'C: barrier2', 'C: cell+publish', 'C: bulk io', '-', '-', 'C: looptop',
                 'G: poll', 'G: lds write', 'G: barrier1', 'G: barrier2', '-', '-', '-', 'G: extra poll rounds x100'],
    ('bwd', 1): ['C: wait sums (barrierA)', 'C: cell bwd+tile', 'C: barrierB', 'C: mfma+publish', 'C: bulk io+coef', '-', '-', 'C: looptop',
                 'G: poll+sum', 'G: lds write', 'G: barrierA', 'G: barrierB', '-', '-', '-', '-'],
    ('fwd', 2): ['gather', 'lds-write+spill', 'store_out+load_xg issue', 'barrier', 'mfma', 'act+publish', 'copy', 'looptop'],
    ('bwd', 2): ['-'] * 8,
}


def report(tag, ms):
    st = ws[:256].view(torch.int64).cpu().tolist()
    print('%s: %.2f ms total (%.2f us/step), abort=%d, hand-off mode per direction (1 write-through, 2 XCD-local): %s' % (tag, ms, ms * 1e3 / T, st[0] & 0xffffffff, st[26:28]))
    for k, nm in enumerate(NAMES[(tag, mode)]):
        if nm != '-':
            print('   %-26s %8.2f us/step' % (nm, st[2 + k] * 0.01 / T))


for it in range(2):
    g2 = gates.clone()
    e0, e1, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e0.record()
    H.call('asr_lstm_fwd', H.ptr(g2), H.ptr(whh), H.ptr(bhh), H.ptr(y), H.ptr(c), B, T, Hd, ND, 1, H.ptr(ws), nb, H.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    report('fwd', e0.elapsed_time(e1))
    e1.record()
    H.call('asr_lstm_bwd', H.ptr(g2), H.ptr(whh), H.ptr(dy), H.ptr(c), B, T, Hd, ND, 1, H.ptr(ws), nb, H.stream_ptr())
    e2.record(); torch.cuda.synchronize()
    report('bwd', e1.elapsed_time(e2))
