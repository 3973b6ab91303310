"""The data-parallel bucket hooks of the HIP autograd functions (src/functions.py: RNNLayerFn.backward calls
`dp.bucket_ready` from INSIDE backward so that a bucket's all-reduce overlaps the remaining BPTT), driven on ONE GPU
with a recording reducer that stands in for two identical replicas (VERDICT r01 missing #7 / ADVICE):

  * every bucket is signalled exactly once, in backward-completion order (heads/decoder, then RNN layers top-down);
  * nothing writes into a bucket after it was signalled: a stream-ordered snapshot taken at the signal equals the
    bucket at the end of backward (an all-reduce started there would have reduced final values);
  * the "reduced" gradient (x2 = SUM over two identical replicas) with grad_mul = 1/2 gives the same clip decision and
    the same Adadelta update as the single-process step.
The communicator itself (torch.distributed all_reduce, token-count weighting) is covered by tests/test_dp_gloo.py.
"""
import numpy as np
import pytest
import torch

from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu

MC = {'ctc_weight': 0.5,
      'encoder': {'vgg': 0, 'vgg_freq': -1, 'vgg_low_filt': -1, 'module': 'LSTM', 'bidirection': True, 'dim': [32, 32, 32],
                  'dropout': [0.1, 0.1, 0.1], 'layer_norm': [False, False, False], 'proj': [True, True, True],
                  'sample_rate': [1, 2, 1], 'sample_style': 'drop'},
      'attention': {'mode': 'loc', 'dim': 24, 'num_head': 1, 'v_proj': False, 'temperature': 0.5, 'loc_kernel_size': 5,
                    'loc_kernel_num': 4},
      'decoder': {'module': 'LSTM', 'dim': 24, 'layer': 1, 'dropout': 0}}


def _model(prec):
    from src.asr import ASR
    cfg = O.ModelCfg(MC, 40, 31)
    sd = O.seeded_state_dict(O.param_shapes(cfg), 3)
    model = ASR(40, 31, 4, prec=prec, seed=9, **MC)
    model.load_state_dict(sd)
    return model.cuda().train()


def _recording_reducer():
    from src.dist import FlatDataParallel

    class Recording(FlatDataParallel):
        def __init__(self, flat_param, flat_grad, buckets=None, group=None):
            super().__init__(flat_param, flat_grad, buckets, group)
            self.world = 2                       # two identical replicas
            self.order, self.snaps = [], {}

        def ce_weight(self, n_tokens_local):
            return 1.0                           # n * 2 / (n + n)

        def bucket_ready(self, idx):
            if idx in self._done:
                return
            self._done.add(idx)
            self.order.append(idx)
            a, b = self.buckets[idx]
            self.flat_grad[a:b].mul_(2.0)        # all-reduce(SUM) over the two replicas, ordered on the compute stream
            self.snaps[idx] = self.flat_grad[a:b].clone()

        def finish(self):
            for i in range(len(self.buckets)):
                self.bucket_ready(i)
            self._done = set()
    return Recording


@pytest.mark.parametrize('overlap_dp', ['0', '1'])
@pytest.mark.parametrize('prec', ['fp32', 'bf16'])
def test_bucket_hooks_inside_backward(prec, overlap_dp, monkeypatch):
    # overlap_dp = 1: the parameter gradients AND the bucket signals are deferred to the CU-masked side stream (ASR_OVERLAP_DP);
    # the same properties must hold - in particular no write into a bucket after its signal
    monkeypatch.setenv('ASR_OVERLAP_DP', overlap_dp)
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from batchgen import make_batch
    from src.optim import Optimizer
    from src.step import train_step
    from src.util import CTCLoss, CrossEntropyLoss
    from src import hipabi as H
    feat, lens, txt = [torch.from_numpy(x).cuda() for x in make_batch(5, 4, 50, 40, 8, 31)]

    ref = _model(prec)
    opt_r = Optimizer(ref.parameters(), 'Adadelta', 1.0, 1e-8)
    out_r = train_step(ref, opt_r, CTCLoss(), CrossEntropyLoss(), feat, lens, txt, 8, clip=0.05)   # clip active
    g_ref = ref.flat_grad.clone()

    model = _model(prec)
    dp = model.attach_data_parallel(reducer_cls=_recording_reducer())
    opt = Optimizer(model.parameters(), 'Adadelta', 1.0, 1e-8)
    out = train_step(model, opt, CTCLoss(), CrossEntropyLoss(), feat, lens, txt, 8, dp=dp, clip=0.05)
    H.raise_if_aborted()
    nb = len(dp.buckets)
    assert dp.order == list(range(nb)), dp.order               # each bucket once, in backward-completion order
    assert nb == 1 + 3                                          # heads+decoder+attention, then 3 RNN layers (no front-end)
    # the RNN buckets follow the layers top-down: bucket 1 = last layer, ..., bucket 3 = first layer
    rnn = [(s, e) for k, s, e in model._ranges if k == 'rnn']
    assert [tuple(b) for b in dp.buckets[1:]] == rnn[::-1]
    for i, (a, b) in enumerate(dp.buckets):
        assert torch.equal(dp.snaps[i], model.flat_grad[a:b]), 'bucket %d was written after it had been signalled' % i
    covered = sum(b - a for a, b in dp.buckets)
    assert covered == model.flat_grad.numel()
    # the reduced buffer holds the SUM over two identical replicas; same clip decision and update as one process
    # both modes: equal up to the order of the split reductions (fp32 atomics).  Two MODEL INSTANCES are compared here; round 2
    # excused a 2e-2 deviation of single decoder logits between instances in bf16 mode - that was the polling defect of
    # csrc/handoff.h, fixed there and pinned by tests/test_handoff_isa.py, so the excuse is gone
    rel = float((model.flat_grad - 2.0 * g_ref).norm() / (2.0 * g_ref).norm())
    assert rel < (1e-5 if prec == 'fp32' else 1e-4), rel
    assert abs(float(out['total_loss']) - float(out_r['total_loss'])) < 1e-6 * max(1.0, abs(float(out_r['total_loss'])))
    n_dp, n_ref = float(out['grad_normsq'].sqrt()) * 0.5, float(out_r['grad_normsq'].sqrt())
    assert abs(n_dp - n_ref) < 1e-4 * n_ref
    drel = float((model.flat_param - ref.flat_param).norm() / (ref.flat_param.norm() + 1e-30))
    assert drel < (1e-6 if prec == 'fp32' else 1e-5), drel
