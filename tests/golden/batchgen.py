"""Seeded synthetic batches shared by the fixture generator and the tests (inputs by seed, so that large
feature tensors need not be stored)."""
import numpy as np


def make_batch(seed, B, T, D, L, V, min_frac=0.5):
    """Zero-padded (B,T,D) features in [0,1), descending ragged lengths, (B,L) 0-padded labels ending in
    <eos>=1 — the batch contract of the reference's collate (src/collect_batch.py:44-48)."""
    g = np.random.Generator(np.random.PCG64(seed))
    feat = g.random((B, T, D), dtype=np.float32)
    lens = np.sort(g.integers(int(T * min_frac), T + 1, size=B))[::-1].copy()
    lens[0] = T
    for b in range(B):
        feat[b, lens[b]:] = 0.0
    tl = g.integers(max(1, L // 2), L + 1, size=B)
    tl[0] = L
    txt = np.zeros((B, L), dtype=np.int64)
    for b in range(B):
        txt[b, :tl[b] - 1] = g.integers(3, V, size=tl[b] - 1)
        txt[b, tl[b] - 1] = 1  # <eos>
    return feat, lens.astype(np.int64), txt
