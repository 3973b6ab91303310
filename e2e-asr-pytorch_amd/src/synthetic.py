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
