"""ORACLE (test infrastructure, NOT product code) -- log-VQT audio front end, CPU fp64 restatement.

Follows ``/root/reference/osu_fusion/scripts/dataset_creator.py:36-55`` (load_audio's feature step):
    np.log(np.abs(librosa.vqt(y, sr=22050, hop_length=176, fmin=C0, n_bins=96, bins_per_octave=12)) + 1e-10)
The arithmetic lives in **librosa==0.10.1** (requirements.txt:6), absent from /root/reference and from this image, and the
reference holds no spectrogram fixtures -> **parity unpinned**.  What is restated here is librosa's published definition:
  filters.wavelet_lengths   lengths = Q*sr / (f + gamma/alpha), Q = 1/alpha, alpha = (r^2-1)/(r^2+1), r = 2^(1/bpo),
                            gamma = 24.7*alpha/0.108 (vqt's gamma=None default)
  filters.wavelet           phasor(arange(-len//2, len//2) * 2 pi f / sr) * periodic hann, L1-normalised, centre-padded to n_fft
  core.constantq vqt        response = fft_basis . stft(y, n_fft, hop, window=ones, center=True, pad_mode="constant"),
                            fft_basis = fft(basis * lengths/n_fft), then V /= sqrt(lengths)   (scale=True)
evaluated the way librosa's __cqt_response does (spectral product of the centred frames) but at the FULL sample rate for every
octave; librosa itself halves the rate between octaves (soxr_hq), keeps the one-sided spectrum and drops the smallest 1 % of each
filter's spectral mass (sparsity=0.01) -- approximations of this same quantity (`one_sided=True` reproduces the second of these).
"""
from __future__ import annotations

import numpy as np

SR = 22050
HOP_LENGTH = 176
FMIN = 440.0 * 2.0 ** ((12 - 69) / 12)          # note_to_hz("C0")
N_BINS = 96
BPO = 12


def wavelet_lengths(freqs: np.ndarray, sr: float, bpo: int = BPO):
    r = 2.0 ** (1.0 / bpo)
    alpha = (r * r - 1.0) / (r * r + 1.0)
    gamma = 24.7 * alpha / 0.108
    q = 1.0 / alpha
    return q * sr / (freqs + gamma / alpha)


def wavelets(freqs: np.ndarray, sr: float, bpo: int = BPO):
    out = []
    for ilen, f in zip(wavelet_lengths(freqs, sr, bpo), freqs):
        t = np.arange(-ilen // 2, ilen // 2, dtype=float)
        sig = np.cos(2 * np.pi * f / sr * t) + 1j * np.sin(2 * np.pi * f / sr * t)
        n = len(sig)
        sig = sig * (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n) / n))
        out.append(sig / np.sum(np.abs(sig)))
    return out


def pad_center(v: np.ndarray, size: int) -> np.ndarray:
    lpad = int((size - len(v)) // 2)
    out = np.zeros(size, dtype=v.dtype)
    out[lpad:lpad + len(v)] = v
    return out


def vqt(y: np.ndarray, sr: float = SR, hop: int = HOP_LENGTH, fmin: float = FMIN, n_bins: int = N_BINS, bpo: int = BPO,
        one_sided: bool = False) -> np.ndarray:
    """Complex (n_bins, 1 + len(y)//hop) variable-Q transform, fp64."""
    y = np.asarray(y, dtype=np.float64)
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bpo)
    lengths = wavelet_lengths(freqs, sr, bpo)
    n_fft = int(2.0 ** np.ceil(np.log2(lengths.max())))
    n_fft = max(n_fft, int(2.0 ** (1 + np.ceil(np.log2(hop)))))
    basis = np.stack([pad_center(w, n_fft) for w in wavelets(freqs, sr, bpo)]) * (lengths[:, None] / n_fft)
    fft_basis = np.fft.fft(basis, axis=1)
    ypad = np.pad(y, n_fft // 2)
    frames = 1 + len(y) // hop
    out = np.empty((n_bins, frames), dtype=np.complex128)
    for t0 in range(0, frames, 256):                       # bounded memory
        t1 = min(frames, t0 + 256)
        seg = np.stack([ypad[t * hop:t * hop + n_fft] for t in range(t0, t1)], axis=1)
        d = np.fft.fft(seg, axis=0)
        if one_sided:
            out[:, t0:t1] = fft_basis[:, :n_fft // 2 + 1] @ d[:n_fft // 2 + 1]
        else:
            out[:, t0:t1] = fft_basis @ d
    return out / np.sqrt(lengths)[:, None]


def log_vqt(y: np.ndarray, **kw) -> np.ndarray:
    return np.log(np.abs(vqt(y, **kw)) + 1e-10)
