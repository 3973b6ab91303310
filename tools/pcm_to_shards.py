#!/usr/bin/env python
"""Real audio in: turns already-decoded PCM (RIFF/WAVE files, or raw .npy sample arrays) into the waveform-shard layout that
src/data.py trains from, reproducing what the reference does to a waveform before the feature extractor sees it.

    python tools/pcm_to_shards.py --src /data/LibriSpeech-wav --out /data/shards --split train-clean-100

The reference reads every utterance with `librosa.load(filepath)` (src/audio.py:283-309): mono, float32 in [-1, 1], RESAMPLED TO
librosa's default 22 050 Hz - and then frames those samples with its 16 kHz parameters (400-sample window, 160-sample hop,
mel bank built for 16 000 Hz: src/audio.py:130-171).  A LibriSpeech second therefore becomes 22 050 samples =
137.8 frames instead of 100.  A model trained by the reference has only ever seen that; shards written here carry the same
samples (`--sr 22050`, the default, marked in the manifest header), `--sr 16000` gives the nominally correct rate instead.
FLAC itself cannot be decoded offline in this build (no codec): decode once elsewhere (e.g. `flac -d`), keep the directory
layout; transcripts are read as the reference reads them (corpus/preprocess_librispeech.py:18-27: `<spk>-<chap>.trans.txt`
beside the audio, one line `utt_id TRANSCRIPT` per utterance).

Resampling: scipy.signal.resample_poly with a Kaiser window (librosa's default `kaiser_best` is a Kaiser-windowed sinc too; the
two agree to ~1e-3 of full scale - no reference fixture pins this leg: the reference holds no test audio).
"""
import argparse
import os
import sys
import wave
from fractions import Fraction

import numpy as np


def read_pcm(path):
    """-> (float32 mono samples in [-1, 1], sample rate).  RIFF/WAVE PCM 8/16/24/32 bit, or .npy (int16 / float32; rate from --in-sr)."""
    if path.endswith('.npy'):
        a = np.load(path)
        if a.dtype == np.int16:
            a = a.astype(np.float32) / 32768.0
        if a.ndim > 1:
            a = a.mean(axis=-1 if a.shape[-1] <= 8 else 0)
        return a.astype(np.float32), None
    with wave.open(path, 'rb') as w:
        nch, width, sr, n = w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        a = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        a = np.frombuffer(raw, dtype='<i2').astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        a = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        a = np.frombuffer(raw, dtype='<i4').astype(np.float32) / 2147483648.0
    else:
        raise ValueError('%s: unsupported sample width %d' % (path, width))
    if nch > 1:
        a = a.reshape(-1, nch).mean(axis=1)          # librosa.load(mono=True): mean of the channels
    return a.astype(np.float32), sr


def resample(x, sr_in, sr_out):
    if sr_in == sr_out:
        return x
    from scipy.signal import resample_poly
    fr = Fraction(sr_out, sr_in)
    return resample_poly(x.astype(np.float64), fr.numerator, fr.denominator, window=('kaiser', 14.769656459379492)).astype(np.float32)


def read_transcripts(root):
    """utt_id -> text from every *.trans.txt under root (reference read_text, corpus/preprocess_librispeech.py:18-27)."""
    text = {}
    for d, _, files in os.walk(root):
        for f in files:
            if f.endswith('.trans.txt'):
                for line in open(os.path.join(d, f), encoding='utf-8'):
                    parts = line.rstrip('\n').split(' ', 1)
                    if len(parts) == 2:
                        text[parts[0]] = parts[1]
    return text


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--src', required=True, help='directory tree of .wav / .npy utterances with LibriSpeech-style *.trans.txt files')
    ap.add_argument('--out', required=True)
    ap.add_argument('--split', default='train-clean-100')
    ap.add_argument('--sr', type=int, default=22050, help="sample rate of the stored waveforms; 22050 = what the reference's librosa.load hands to its 16 kHz front-end")
    ap.add_argument('--in-sr', type=int, default=16000, help='sample rate of .npy inputs (WAVE files carry their own)')
    ap.add_argument('--int16', action='store_true', help='store int16 PCM instead of float32')
    ap.add_argument('--subset', type=int, default=None)
    a = ap.parse_args(argv)
    text = read_transcripts(a.src)
    files = []
    for d, _, fs in os.walk(a.src):
        for f in sorted(fs):
            if f.endswith(('.wav', '.WAV', '.npy')):
                files.append(os.path.join(d, f))
    files.sort()
    if a.subset:
        files = files[:a.subset]
    if not files:
        sys.exit('no .wav / .npy files under %s' % a.src)
    sdir = os.path.join(a.out, a.split)
    os.makedirs(sdir, exist_ok=True)
    rows, missing = [], 0
    for path in files:
        uid = os.path.splitext(os.path.basename(path))[0]
        if uid not in text:
            missing += 1
            continue
        x, sr = read_pcm(path)
        x = resample(x, sr or a.in_sr, a.sr)
        arr = np.clip(np.round(x * 32767.0), -32768, 32767).astype(np.int16) if a.int16 else x
        np.save(os.path.join(sdir, uid + '.npy'), arr)
        rows.append((uid, uid + '.npy', len(arr), text[uid]))
    with open(os.path.join(sdir, 'manifest.tsv'), 'w', encoding='utf-8') as f:
        for r in rows:
            f.write('%s\t%s\t%d\t%s\n' % r)
    with open(os.path.join(sdir, 'README'), 'w') as f:
        f.write('waveforms stored at %d Hz; the front-end frames them with its 16 kHz parameters (reference src/audio.py:283-309: '
                'librosa.load default rate)\n' % a.sr)
    print('wrote %d utterances to %s (%d files without a transcript skipped); stored rate %d Hz' % (len(rows), sdir, missing, a.sr))


if __name__ == '__main__':
    main()
