"""Times asr_gemm on the encoder's shapes through the C ABI (bf16 contraction mode).  usage: python tools/bench_gemm.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch
from src import hipabi as H

def run(name, M, N, K, a_kc, b_kc, splits=1, accum=0, reps=5):
    dev = 'cuda'
    A = torch.randn((M, K) if a_kc else (K, M), device=dev)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev)
    C = torch.zeros(M, N, device=dev)
    lda = K if a_kc else M
    ldb = K if b_kc else N
    def call():
        H.call('asr_gemm', H.ptr(A), H.ptr(B), H.ptr(C), None, M, N, K, lda, ldb, N, a_kc, b_kc, 0, accum, splits, 1, 0, 0, 0, 0, 0, 1, H.stream_ptr())
    call(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    gb = 4.0 * (M * K + N * K + M * N) / 1e9
    print('%-34s M=%6d N=%5d K=%6d splits=%2d : %8.1f us  %7.1f TFLOP/s  (operands+output %.0f MB -> %.0f us at 5 TB/s)'
          % (name, M, N, K, splits, us, 2.0 * M * N * K / us / 1e6, gb * 1e3, gb / 5e3 * 1e6))

run('fwd in-proj L1 (x W^T)', 19200, 2560, 640, 1, 1)
run('fwd in-proj L0', 19200, 2560, 160, 1, 1)
run('fwd in-proj L2', 9600, 2560, 640, 1, 1)
run('fwd proj', 19200, 640, 640, 1, 1)
run('dgrad in-proj L1 (dy W)', 19200, 640, 2560, 1, 0)
run('dgrad in-proj L2', 9600, 640, 2560, 1, 0)
run('wgrad in-proj L1 (dy^T x)', 2560, 640, 19200, 0, 0, splits=18, accum=1)
run('wgrad in-proj L2', 2560, 640, 9600, 0, 0, splits=9, accum=1)
run('wgrad W_hh L1', 1280, 320, 19200, 0, 0, splits=18, accum=1)
run('wgrad proj', 640, 640, 19200, 0, 0, splits=18, accum=1)
print('--- split sweep, wgrad in-proj L1')
for sp in (2, 4, 6, 9, 12, 18, 24, 32):
    run('wgrad in-proj L1', 2560, 640, 19200, 0, 0, splits=sp, accum=1)
for sp in (4, 9, 18, 32):
    run('wgrad W_hh L1', 1280, 320, 19200, 0, 0, splits=sp, accum=1)
