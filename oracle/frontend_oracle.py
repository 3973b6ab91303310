"""CPU oracle of the acoustic front-end.  TEST INFRASTRUCTURE ONLY (same rules as asr_oracle.py).

Delta / Postprocess / Augment restate src/audio.py:40-121,355-406 and are pinned by tests/golden/g5_frontend.npz
(generated from the genuine reference).  `fbank` restates ExtractAudioFeature.forward (src/audio.py:124-171,231-244)
on top of torch.stft as torchaudio.transforms.Spectrogram documents it (n_fft 1025, hann 400, hop 160, center,
reflect padding, power 2 then sqrt); torchaudio is not vendored by the reference and is absent here, so that part
is pinned only by self-consistency: STFT/mel parity unpinned."""
import numpy as np
import torch
import torch.nn.functional as F


def delta(mel_cft, filters):
    """mel (C=1,F,T) -> (order+1,F,T): conv2d with (order+1,1,1,taps) filters, zero padding along time."""
    taps = filters.shape[-1]
    w = torch.as_tensor(filters).reshape(filters.shape[0], 1, 1, taps)
    return F.conv2d(torch.as_tensor(mel_cft).unsqueeze(0), w, padding=(0, (taps - 1) // 2))[0]


def postprocess(x_cft):
    return x_cft.permute(2, 0, 1).reshape(x_cft.shape[2], -1)


def augment(x_td, draws):
    """SpecAugment with explicit draws [t, t0, tend, f, f0, fend] on a (T,D) array (returns a copy)."""
    x = torch.as_tensor(x_td).clone().t()      # (D,T) view as in the reference
    t, t0, tend, f, f0, fend = [int(v) for v in draws]
    if t != 0:
        x[:, t0:tend] = x.mean()
    if f != 0:
        x[f0:fend, :] = x.mean()
    return x.t()


def fbank(wav_1n, mel_fb, n_fft=1025, win=400, hop=160, preemph=0.97, ref_db=20.0, min_db=-100.0):
    wav = torch.as_tensor(wav_1n, dtype=torch.float32)
    wav = torch.cat([wav[:, :1], wav[:, 1:] - preemph * wav[:, :-1]], dim=-1)
    spec = torch.stft(wav, n_fft=n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=True,
                      pad_mode='reflect', normalized=False, onesided=True, return_complex=True)
    mag = spec.abs().pow(2).sqrt()                                        # (1, 513, T)
    mel = torch.matmul(torch.as_tensor(mel_fb), mag)                      # (1, 80, T)
    db = 20 * torch.log10(torch.clamp(mel, min=1e-5)) - ref_db
    return torch.clamp((db - min_db) / -min_db, min=0, max=1)
