"""LibriSpeech-shaped synthetic batches (no corpus is available offline): the batch contract of the
reference's collate — zero-padded fp32 features (B,T,D), int64 lengths sorted descending, 0-padded int64
labels (B,L) each ending in <eos>=1 (src/collect_batch.py:44-48; SURVEY §8d for the length model)."""
import numpy as np
import torch


def librispeech_shaped_batch(B, T, D, L, V=31, seed=1234, ragged=True, device='cpu'):
    g = np.random.Generator(np.random.PCG64(seed))
    if ragged:
        lens = np.clip(np.round(g.normal(0.83 * T, 0.28 * T, size=B)), min(150, T), T).astype(np.int64)
        lens = np.sort(lens)[::-1].copy()
        lens[0] = T
    else:
        lens = np.full(B, T, dtype=np.int64)
    feat = g.random((B, T, D), dtype=np.float32)
    for b in range(B):
        feat[b, lens[b]:] = 0.0
    tl = np.clip(np.round(lens * (float(L) / T)), 5, L).astype(np.int64)
    tl[0] = L
    txt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        txt[b, :tl[b] - 1] = g.integers(3, V, size=tl[b] - 1)
        txt[b, tl[b] - 1] = 1
    return (torch.from_numpy(feat).to(device), torch.from_numpy(lens).to(device), torch.from_numpy(txt).to(device))


HALF_BATCHSIZE_AUDIO_LEN = 800     # reference src/collect_batch.py:6


def librispeech_length_batches(n, D, V=31, seed=1234, batch_size=16, halve=True, device='cpu'):
    """n batches whose lengths follow the train-clean-100 model of SURVEY 8d: T_b ~ clip(N(1270, 480), 150, 2450) frames, sorted
    descending inside the bucket, text length clip(round(0.14 T_b), 5, 400) tokens U{3..V-1} + <eos>; a bucket whose longest
    utterance exceeds 800 frames keeps every second utterance (the reference's halving rule, src/collect_batch.py:21-24).
    Returns a list of (feat (B,T,D), feat_len (B), txt (B,L)) with T = the bucket's longest utterance, L its longest text."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = []
    for _ in range(n):
        lens = np.clip(np.round(g.normal(1270.0, 480.0, size=batch_size)), 150, 2450).astype(np.int64)
        lens = np.sort(lens)[::-1].copy()
        if halve and lens[0] > HALF_BATCHSIZE_AUDIO_LEN:
            lens = lens[::2].copy()
        B, T = len(lens), int(lens[0])
        tl = np.clip(np.round(lens * 0.14), 5, 400).astype(np.int64)
        L = int(tl.max())
        feat = g.random((B, T, D), dtype=np.float32)
        txt = np.zeros((B, L), dtype=np.int64)
        for b in range(B):
            feat[b, lens[b]:] = 0.0
            txt[b, :tl[b] - 1] = g.integers(3, V, size=tl[b] - 1)
            txt[b, tl[b] - 1] = 1
        out.append((torch.from_numpy(feat).to(device), torch.from_numpy(lens).to(device), torch.from_numpy(txt).to(device)))
    return out
