"""Phase timers of the STREAMED-tile decoder kernels (csrc/decoder_stream.hip; needs `make -C e2e-asr-pytorch_amd/csrc diag`).
usage: python tools/diag_dec_stream.py [B Tp L]      default 64 1500 40 (BASELINE config 5's decoder shape, 40 of its 400 tokens)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch, yaml
from src import hipabi as H
lib = ctypes.CDLL(os.environ.get('ASR_DIAG_LIB', os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'diag', 'libasr_hip_diag.so')))
for name, argtypes in H.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
for name, (rt, at) in H._RESTYPES.items():
    fn = getattr(lib, name); fn.argtypes = at; fn.restype = rt
H._lib = lib
from src import functions as F
from src.asr import ASR
B, Tp, L = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (64, 1500, 40)
config = yaml.safe_load(open(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')))
model = ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()
lib.asr_att_decoder_set_persistent(3 | 4 | 8)
g = torch.Generator().manual_seed(1)
enc = torch.tanh(torch.randn(B, Tp, 640, generator=g)).cuda()
enc_len = torch.randint(int(0.6 * Tp), Tp + 1, (B,), generator=g); enc_len[0] = Tp; enc_len = enc_len.cuda()
teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
FW = ['cell update+publish .. wait H', 'B1 barrier', 'query', 'conv (incl. its barrier)', 'key prefetch + wait Q (B2)', 'sweep', 'c3 barrier',
      'softmax statistics (2 barriers)', 'partial context + publish', 'wait S (B3)', 'combine', 'c5 barrier', 'cell rows', 'c6 barrier']
for it in range(2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    d, st = F.att_decoder_forward(model, enc, enc_len, L, teacher, H.BF16)
    e1.record(); torch.cuda.synchronize()
    w = st['work'][:4096].view(torch.int64).cpu().tolist()
print('B=%d Tp=%d L=%d  plan %d' % (B, Tp, L, lib.asr_att_decoder_fwd_plan(ctypes.byref(d))))
print('decoder forward call: %.2f ms (%.1f us per token), abort=%d' % (e0.elapsed_time(e1), e0.elapsed_time(e1) * 1e3 / L, w[0] & 0xffffffff))
tot = 0.0
for k, nm in enumerate(FW):
    us = w[128 + k] * 0.01 / L
    tot += us
    print('   %-36s %7.2f us/step' % (nm, us))
print('   %-36s %7.2f us/step   (workgroup 0 of utterance 0)' % ('sum', tot))
BW = ['wait H4 .. S1 start', 'S1 cell bwd', 'Ba barrier', 'P1 transposed weights . dgates', 'Bb barrier', 'C publish + operand requests', 'wait C (H2)',
      'P2 dot + dattn passes + de', 'P3/P4 sweep groups + Q publish', '-', 'wait Q,V (H3)', 'P5 dq, transposed conv, W_q rows, N publish', 'wait N (H4)']
model.zero_grad()
enc2 = enc.clone().requires_grad_(True)
logits, _, _ = F.AttDecoderFn.apply(model._anchor, enc2, enc_len, teacher, L, model, H.BF16)
gout = torch.randn_like(logits) * 0.1
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
logits.backward(gout)
e1.record(); torch.cuda.synchronize()
off = int(lib.asr_att_decoder_bwd_status_offset(ctypes.byref(d)))
w = model._last_dec_bwd_ws[off:off + 4096].view(torch.int64).cpu().tolist()
print('decoder backward (whole call incl. tail contractions): %.2f ms, plan %d, abort=%d' % (e0.elapsed_time(e1), lib.asr_att_decoder_bwd_plan(ctypes.byref(d)), w[0] & 0xffffffff))
tot = 0.0
for k, nm in enumerate(BW):
    us = w[128 + k] * 0.01 / L
    tot += us
    print('   %-44s %7.2f us/step' % (nm, us))
print('   %-44s %7.2f us/step' % ('sum', tot))
