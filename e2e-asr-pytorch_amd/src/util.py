"""Host-side helpers mirroring the reference's src/util.py surface (init, losses, timer, error rates).
The loss modules call the HIP kernels; nothing here computes on the CPU except logging/metrics."""
import math
import time

import torch
from torch import nn

from src import functions as F_hip


def init_weights_(model):
    """End state of the reference's `self.apply(init_weights)` (src/asr.py:45-46, src/util.py:60-83):
    every 1-D parameter 0; 2-D N(0, 1/sqrt(fan_in)); 3/4-D N(0, 1/sqrt(C_in*k...)).  (The Embedding
    special case is overridden by the root call)"""
    for p in model.parameters():
        d = p.data
        if d.dim() == 1:
            d.zero_()
        elif d.dim() == 2:
            d.normal_(0, 1.0 / math.sqrt(d.size(1)))
        elif d.dim() in (3, 4):
            n = d.size(1)
            for k in d.size()[2:]:
                n *= k
            d.normal_(0, 1.0 / math.sqrt(n))
        else:
            raise NotImplementedError


def init_gate_(bias):
    """Forget-gate bias = 1 (src/util.py:84-88)."""
    n = bias.size(0)
    bias.data[n // 4:n // 2].fill_(1.)
    return bias


class CTCLoss(nn.Module):
    """torch.nn.CTCLoss(blank=0, zero_infinity=False) call-compatible (bin/train_asr.py:135,237): takes
    log-probs as (T,B,V) — normally a transposed view of the model's (B,T,V) output — padded targets
    (B,L), input lengths, target lengths.  Runs asr_ctc_loss."""

    def __init__(self, blank=0, zero_infinity=False):
        super().__init__()
        assert blank == 0 and not zero_infinity

    def forward(self, log_probs_tbv, targets, input_lengths, target_lengths):
        if not torch.is_tensor(input_lengths):
            input_lengths = torch.as_tensor(input_lengths)
        if not torch.is_tensor(target_lengths):
            target_lengths = torch.as_tensor(target_lengths)
        return F_hip.CTCLossFn.apply(log_probs_tbv.transpose(0, 1), targets, input_lengths, target_lengths)


class CrossEntropyLoss(nn.Module):
    """torch.nn.CrossEntropyLoss(ignore_index=0) call-compatible on (N,V) logits / (N) targets."""

    def __init__(self, ignore_index=0):
        super().__init__()
        assert ignore_index == 0

    def forward(self, logits, target):
        return F_hip.SeqLossFn.apply(logits, target, 0, logits.shape[-1], 0.0)


class LabelSmoothingLoss(nn.Module):
    """Reference src/util.py:11-25 (mean over all rows, off-target mass smoothing/(classes-1))."""

    def __init__(self, classes, smoothing=0.0, dim=-1):
        super().__init__()
        self.cls, self.smoothing = classes, smoothing

    def forward(self, pred, target):
        return F_hip.SeqLossFn.apply(pred, target, 1, self.cls, float(self.smoothing))


class Timer():
    ''' Wall-clock split of a step into rd / fw / bw, like the reference's Timer (src/util.py:30-57). '''

    def __init__(self):
        self.prev_t = time.time()
        self.clear()

    def set(self):
        self.prev_t = time.time()

    def cnt(self, mode):
        self.time_table[mode] += time.time() - self.prev_t
        self.set()
        if mode == 'bw':
            self.click += 1

    def show(self):
        total = sum(self.time_table.values())
        n = max(self.click, 1)
        msg = '{:.3f} sec/step (rd {:.1f}% | fw {:.1f}% | bw {:.1f}%)'.format(
            total / n, *[100 * self.time_table[k] / max(total, 1e-9) for k in ('rd', 'fw', 'bw')])
        self.clear()
        return msg

    def clear(self):
        self.time_table = {'rd': 0, 'fw': 0, 'bw': 0}
        self.click = 0


def human_format(num):
    magnitude = 0
    while num >= 1000:
        magnitude += 1
        num /= 1000.0
    return '{:3.1f}{}'.format(num, [' ', 'K', 'M', 'G', 'T', 'P'][magnitude])


def edit_distance(a, b):
    """Levenshtein distance between two sequences (stands in for the `editdistance` package)."""
    if len(a) < len(b):
        a, b = b, a
    prev = list(range(len(b) + 1))
    for i, x in enumerate(a, 1):
        cur = [i]
        for j, y in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
        prev = cur
    return prev[-1]


def cal_er(tokenizer, pred, truth, mode='wer', ctc=False):
    """Batch error rate as the reference logs it (src/util.py:123-139)."""
    if pred is None:
        return float('nan')
    if len(pred.shape) >= 3:
        pred = pred.argmax(dim=-1)
    er = []
    for p, t in zip(pred.tolist(), truth.tolist()):
        p = tokenizer.decode(p, ignore_repeat=ctc)
        t = tokenizer.decode(t)
        if mode in ('wer', 'per'):
            p, t = p.split(' '), t.split(' ')
        er.append(1. if len(t) == 0 else float(edit_distance(p, t)) / len(t))
    return sum(er) / len(er)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
