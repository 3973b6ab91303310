"""TFLOP/s of the encoder stack's bf16 contractions (asr_gemm16) at the shapes of config/librispeech_asr.yaml, B=16, T=1200.
usage: python tools/bench_gemm16.py [--json out.json]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch
from src import hipabi as H

SHAPES = [  # (name, form, M, N, K)
    ('L0 input projection', 'nt', 19200, 2560, 160), ('L1 input projection', 'nt', 19200, 2560, 640),
    ('L2/L3 input projection', 'nt', 9600, 2560, 640), ('L0 projection', 'nt', 19200, 640, 640),
    ('L1-L3 projection', 'nt', 9600, 640, 640), ('L1 input gradient', 'nt', 19200, 640, 2560),
    ('L2/L3 input gradient', 'nt', 9600, 640, 2560),
    ('L0 dW_ih', 'tn', 2560, 160, 19200), ('L1 dW_ih', 'tn', 2560, 640, 19200), ('L2/L3 dW_ih', 'tn', 2560, 640, 9600),
    ('L0/L1 dW_hh (one direction)', 'tn', 1280, 320, 19200), ('L0 dW_pj', 'tn', 640, 640, 19200), ('L1-L3 dW_pj', 'tn', 640, 640, 9600),
]
res = []
if '--tn-only' in sys.argv:
    SHAPES = [x for x in SHAPES if x[1] == 'tn']
for name, form, M, N, K in SHAPES:
    g = torch.Generator().manual_seed(1)
    if form == 'nt':
        A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda(); B = torch.randn(N, K, generator=g).to(torch.bfloat16).cuda()
        C = torch.empty(M, N, dtype=torch.bfloat16, device='cuda'); bias = torch.zeros(N, device='cuda')
        run = lambda: H.gemm16(A, B, C, M, N, K, K, K, N, 1, 1, bias=bias)
    else:
        A = torch.randn(K, M, generator=g).to(torch.bfloat16).cuda(); B = torch.randn(K, N, generator=g).to(torch.bfloat16).cuda()
        C = torch.zeros(M, N, device='cuda'); sp = H.wgrad_splits(K, M, N)
        if os.environ.get('ASR_TN_SPLITS'):
            sp = int(os.environ['ASR_TN_SPLITS'])
        run = lambda: H.gemm16(A, B, C, M, N, K, M, N, N, 0, 0, accum=1, splits=sp)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    tf = 2.0 * M * N * K / us / 1e6
    res.append({'name': name, 'form': form, 'M': M, 'N': N, 'K': K, 'splits': (sp if form == 'tn' else None), 'us': us, 'tflops': tf, 'frac_of_2500': tf / 2500.0})
    print('%-32s %s  M=%6d N=%5d K=%6d  %8.1f us  %7.1f TFLOP/s  (%.3f of the bf16 peak)' % (name, form, M, N, K, us, tf, tf / 2500.0))
if '--json' in sys.argv:
    json.dump(res, open(sys.argv[sys.argv.index('--json') + 1], 'w'), indent=1)
