"""Race detector for the persistent decoder kernels (needs `make -C e2e-asr-pytorch_amd/csrc jitter`).

The jitter build sleeps for a pseudo-random time at every phase boundary of every wave; each launch (new epoch) runs under a
different interleaving.  The same decoder forward (+ backward) is launched N times and compared with the launch-per-step
kernels; anything beyond bf16 rounding that CHANGES from launch to launch is an ordering bug.
usage: python tools/jitter_dec.py [N] [small|bench|both|stream]      (stream: the streamed-tile plans, forced, at B=8 x T'=1225 and B=64 x T'=1500)"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, yaml
from src import hipabi as H
lib = ctypes.CDLL(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'lib', 'diag_jitter', 'libasr_hip_jitter.so'))
for name, argtypes in H.SIGNATURES.items():
    fn = getattr(lib, name); fn.argtypes = argtypes; fn.restype = ctypes.c_int
for name, (rt, at) in H._RESTYPES.items():
    fn = getattr(lib, name); fn.argtypes = at; fn.restype = rt
H._lib = lib
from src import functions as F
from src.asr import ASR
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
which = sys.argv[2] if len(sys.argv) > 2 else 'both'


def run(model, B, Tp, L, E, tag, pflags=3):
    g = torch.Generator().manual_seed(3)
    enc0 = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
    enc_len = torch.randint(max(Tp // 2, 1), Tp + 1, (B,), generator=g); enc_len[0] = Tp; enc_len = enc_len.cuda()
    teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
    dlog = (torch.randn(B, L, 31, generator=g) * 0.1).cuda()
    names = [n for n, _ in model.named_parameters() if n.startswith(('decoder', 'attention', 'pre_embed'))]

    def one(flags):
        lib.asr_att_decoder_set_persistent(flags)
        model.zero_grad()
        enc = enc0.clone().requires_grad_(True)
        logits, att, hs = F.AttDecoderFn.apply(model._anchor, enc, enc_len, teacher, L, model, H.BF16)
        (logits * dlog).sum().backward()
        torch.cuda.synchronize()
        H.raise_if_aborted()
        gr = torch.cat([p.grad.detach().reshape(-1) for n, p in model.named_parameters() if n in names] + [enc.grad.reshape(-1)])
        return logits.detach().clone(), att.detach().clone(), gr.clone()
    ref = one(0)
    first = one(pflags)
    base = [float((first[0] - ref[0]).abs().max()), float((first[1] - ref[1]).abs().max()), float((first[2] - ref[2]).norm() / ref[2].norm())]
    print('%s: persistent vs per-step kernels: logits %.2e  att %.2e  grad rel %.2e' % (tag, *base))
    worst = [0.0, 0.0, 0.0]
    for it in range(N):
        cur = one(pflags)
        d = [float((cur[0] - first[0]).abs().max()), float((cur[1] - first[1]).abs().max()), float((cur[2] - first[2]).norm() / first[2].norm())]
        worst = [max(a, b) for a, b in zip(worst, d)]
        if d[0] > 1e-6 or d[1] > 1e-6 or d[2] > 1e-4:
            idx = np.unravel_index(int((cur[0] - first[0]).abs().argmax()), cur[0].shape)
            print('  launch %d DIFFERS from launch 0: logits %.3e (at b,t,v = %s) att %.3e grad rel %.3e' % (it, d[0], tuple(int(i) for i in idx), d[1], d[2]), flush=True)
    print('%s: %d jittered launches, worst change vs launch 0: logits %.2e  att %.2e  grad rel %.2e' % (tag, N, *worst), flush=True)


if which in ('small', 'both'):
    import test_hip_model as T
    meta, z = T.load(os.path.join(ROOT, 'tests', 'golden'), 'g3_small_ln_concat')
    cfg, sd, model = T.build(meta, 'bf16'); model.train()
    E = 2 * meta['model']['encoder']['dim'][-1] if meta['model']['encoder'].get('bidirection', True) else meta['model']['encoder']['dim'][-1]
    E = model.decoder_input_dim if hasattr(model, 'decoder_input_dim') else int(z['ctc_output'].shape[0] and model.attention.proj_k.weight.shape[1])
    run(model, 3, 18, 7, E, 'small (g3 fixture dims)')
if which in ('bench', 'both'):
    config = yaml.safe_load(open(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')))
    model = ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()
    run(model, 16, 600, 60, 640, 'bench shape (B=16, T\'=600, L=60)')
if which == 'stream':
    config = yaml.safe_load(open(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')))
    model = ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()
    run(model, 8, 1225, 24, 640, "streamed plans (B=8, T'=1225, L=24)", pflags=3 | 4 | 8)
    run(model, 64, 1500, 8, 640, "streamed plans (B=64, T'=1500, L=8)", pflags=3 | 4 | 8)
