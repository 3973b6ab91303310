"""Timing and in-kernel phase timers of the bf16-storage persistent LSTM (csrc/lstm_persist3.hip).
usage: python tools/diag_lstm3.py [--diag] [B] [T]
  --diag : load lib/diag/libasr_hip_diag.so (make -C e2e-asr-pytorch_amd/csrc diag) and print the phase timers
Poll delays: env ASR_LSTM3_POLL_DELAY_FWD / _BWD (x128 clocks), hand-off mode: ASR_LSTM_XCD_LOCAL=0 forces write-through."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch
from src import hipabi as H
args = [a for a in sys.argv[1:] if not a.startswith('--')]
diag = '--diag' in sys.argv
if diag:
    lib = ctypes.CDLL(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'diag', 'libasr_hip_diag.so'))
    for name, argtypes in H.SIGNATURES.items():
        fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
    for name, (rt, at) in H._RESTYPES.items():
        fn = getattr(lib, name); fn.argtypes = at; fn.restype = rt
    H._lib = lib
lib = H.lib()
B = int(args[0]) if args else 16
T = int(args[1]) if len(args) > 1 else 1200
Hd, ND = 320, 2
g = torch.Generator().manual_seed(0)
gates = (torch.randn(B, T, ND, Hd, 4, generator=g) * 0.5).to(torch.bfloat16).cuda()
whh = (torch.randn(ND, 4 * Hd, Hd, generator=g) / Hd ** 0.5).cuda()
dy = (torch.randn(B, T, ND * Hd, generator=g) * 0.1).to(torch.bfloat16).cuda()
y = torch.zeros(B, T + 2, ND * Hd, dtype=torch.bfloat16).cuda(); c = torch.empty(B, T, ND, Hd).cuda()
nf, nb = lib.asr_lstm16_workspace_bytes(B, Hd, ND, 0), lib.asr_lstm16_workspace_bytes(B, Hd, ND, 1)
wsf, wsb = torch.zeros(nf, dtype=torch.uint8).cuda(), torch.zeros(nb, dtype=torch.uint8).cuda()
NAMES = {'fwd': ['C: wait tile (barrier)', 'C: lds read+mfma+cell', 'C: shuffle+publish+y', 'C: bulk io', '-', '-', '-', 'C: looptop',
                 'G: sleep+poll', 'G: lds write', 'G: barrier', '-', '-', '-', '-', '-'],
         'bwd': ['C: wait sums (barrierA)', 'C: cell bwd+tile', 'C: barrierB', 'C: mfma+publish', 'C: bulk io+coef', '-', '-', 'C: looptop',
                 'G: sleep+poll+sum', 'G: lds write', 'G: barrierA', 'G: barrierB', '-', '-', '-', '-']}


def report(tag, ms, ws, epoch):
    st = ws[(epoch & 1) * 1024:][:1024].view(torch.int64).cpu().tolist()          # the launch's status block (parity of its epoch)
    print('%s: %.3f ms (%.3f us/step) abort=%d modes=%s' % (tag, ms, ms * 1e3 / T, st[0] & 0xffffffff, st[26:34]))
    if diag:
        for k, nm in enumerate(NAMES[tag]):
            if nm != '-':
                print('   %-26s %8.3f us/step' % (nm, st[64 + k] * 0.01 / T))


for it in range(3):
    g2 = gates.clone()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    H.call('asr_lstm16_fwd', H.ptr(g2), H.ptr(whh), H.ptr(y), H.ptr(c), B, T, Hd, ND, H.ptr(wsf), nf, it + 1, 0, H.stream_ptr())
    e[1].record()
    H.call('asr_lstm16_bwd', H.ptr(g2), H.ptr(whh), H.ptr(dy), H.ptr(c), B, T, Hd, ND, H.ptr(wsb), nb, it + 1, 0, H.stream_ptr())
    e[2].record(); torch.cuda.synchronize()
    if it > 0:
        report('fwd', e[0].elapsed_time(e[1]), wsf, it + 1)
        report('bwd', e[1].elapsed_time(e[2]), wsb, it + 1)
