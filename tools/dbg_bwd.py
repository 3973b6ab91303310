import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'e2e-asr-pytorch_amd'))
import torch, yaml
from src import hipabi as H
from src import functions as F
from src.asr import ASR
config = yaml.safe_load(open(os.path.join(ROOT, 'e2e-asr-pytorch_amd', 'config', 'librispeech_asr.yaml')))
torch.manual_seed(3)
model = ASR(160, 31, 16, prec='bf16', seed=5, **config['model']).cuda().train()
B, Tp, L, E = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), 640
g = torch.Generator().manual_seed(7 * B + Tp)
enc0 = torch.tanh(torch.randn(B, Tp, E, generator=g)).cuda()
enc_len = torch.randint(max(Tp // 3, 1), Tp + 1, (B,), generator=g); enc_len[0] = Tp; enc_len = enc_len.cuda()
teacher = torch.randint(2, 31, (B, L), generator=g).cuda()
dlog = (torch.randn(B, L, 31, generator=g) * 0.1).cuda()
res = {}
for mode in (1, 3):
    H.lib().asr_att_decoder_set_persistent(mode)
    model.zero_grad()
    enc = enc0.clone().requires_grad_(True)
    holder = {}
    orig = F.AttDecoderFn.backward
    logits, _, hs = F.AttDecoderFn.apply(model._anchor, enc, enc_len, teacher, L, model, H.BF16)
    fn = logits.grad_fn
    st = fn.st
    (logits * dlog).sum().backward()
    torch.cuda.synchronize()
    res[mode] = {'dgates': st['gates'].clone(), 'dconv': st['conv'].clone(), 'denc': enc.grad.clone(),
                 'wg': model.attention.att_layer.gen_energy.weight.grad.clone(), 'wq': model.attention.proj_q.weight.grad.clone(),
                 'wproj': model.attention.att_layer.loc_proj.weight.grad.clone(), 'wk': model.attention.proj_k.weight.grad.clone()}
a, b_ = res[3], res[1]
for name in a:
    x, y = a[name].float(), b_[name].float()
    print(name, 'finite', bool(torch.isfinite(x).all()), 'rel', float((x - y).norm() / (y.norm() + 1e-12)))
dg, dr = a['dgates'].float(), b_['dgates'].float()
for t in range(L - 1, max(L - 5, -1), -1):
    print(' t=%d dgates rel %.3g finite %s | dconv rel %.3g' % (t, float((dg[:, t] - dr[:, t]).norm() / (dr[:, t].norm() + 1e-12)), bool(torch.isfinite(dg[:, t]).all()),
          float((a['dconv'][:, t] - b_['dconv'][:, t]).norm() / (b_['dconv'][:, t].norm() + 1e-12))))
