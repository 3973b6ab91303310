"""Phase timers of the persistent decoder forward (needs `make -C e2e-asr-pytorch_amd/csrc diag`)."""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch, yaml
from src import hipabi as H
H._lib = None
lib = ctypes.CDLL(os.environ.get('ASR_DIAG_LIB', os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'diag', 'libasr_hip_diag.so')))
for name, argtypes in H.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
for name, (rt, at) in H._RESTYPES.items():
    fn = getattr(lib, name); fn.argtypes = at; fn.restype = rt
H._lib = lib
from src import functions as F
from src.asr import ASR
config = yaml.safe_load(open(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')))
model = ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()
B, Tp, L, E = 16, 600, 180, 640
g = torch.Generator().manual_seed(1)
enc = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
enc_len = torch.randint(250, Tp + 1, (B,), generator=g); enc_len[0] = Tp; enc_len = enc_len.cuda()
teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
NAMES = ['wait H (B1)', 'B1 barrier', 'query', 'conv', 'wait Q (B2)', 'sweep', 'c3 barrier', 'stats+ctx+publish', 'wait S (B3)',
         'combine', 'c5 barrier', 'cell rows', 'c6 barrier', 'cell update+publish', '-', '-']
for it in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _, st = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
    e1.record(); torch.cuda.synchronize()
    w = st['work'][:4096].view(torch.int64).cpu().tolist()
    print('decoder forward call: %.2f ms, abort=%d, cluster modes %s' % (e0.elapsed_time(e1), w[0] & 0xffffffff, w[26:34]))
    # note: mark k accumulates the time BEFORE it since the previous mark
    ticks = w[128:144]
    order = [(0, 'cell update+publish .. wait H'), (1, 'B1 barrier'), (2, 'query'), (3, 'conv (incl. its barrier)'), (4, 'wait Q (B2)'), (5, 'sweep'),
             (6, 'c3 barrier'), (7, 'stats+ctx+publish'), (8, 'wait S (B3)'), (9, 'combine'), (10, 'c5 barrier'), (11, 'cell rows'),
             (12, 'c6 barrier'), (13, 'cell update+publish')]
    tot = 0.0
    for k, nm in order:
        us = ticks[k] * 0.01 / L
        tot += us
        print('   %-32s %7.2f us/step' % (nm, us))
    print('   %-32s %7.2f us/step' % ('sum', tot))

# ---- backward
from src import functions as F2
NB = ['wait H4 .. S1 start', 'S1 cell bwd (all units)', 'Ba barrier', 'P1 transposed weights (registers) . dgates', 'Bb barrier',
      'C publish + operand requests', 'wait C (H2)', 'P2 dattn + dot + de', 'P3 sweep + Q publish', 'P4 dconv (MFMA) + V publish',
      'wait Q,V (H3)', 'P5d barrier, N publish, operand requests', 'wait N (H4)', 'P5a dq sum', 'P5b transposed conv partials',
      'P5c barrier, query part, tile sums']
model.zero_grad()
enc2 = enc.clone().requires_grad_(True)
logits, _, _ = F2.AttDecoderFn.apply(model._anchor, enc2, enc_len, teacher, L, model, H.BF16)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
gout = torch.randn_like(logits) * 0.1
torch.cuda.synchronize()
e0.record()
logits.backward(gout)
e1.record(); torch.cuda.synchronize()
print('decoder backward: %.2f ms' % e0.elapsed_time(e1))
d = F2._dec_dims(model, B, Tp, L)
off = int(H.lib().asr_att_decoder_bwd_status_offset(ctypes.byref(d)))
w = model._last_dec_bwd_ws[off:off + 4096].view(torch.int64).cpu().tolist()
print('abort=%d modes %s' % (w[0] & 0xffffffff, w[26:34]))
ticks = w[128:144]
tot = 0.0
for k, nm in enumerate(NB):
    us = ticks[k] * 0.01 / L
    tot += us
    print('   %-48s %7.2f us/step' % (nm, us))
print('   %-48s %7.2f us/step' % ('sum', tot))
