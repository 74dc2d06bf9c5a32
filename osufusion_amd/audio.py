"""Audio front end: waveform -> (96, n_frames) log-magnitude variable-Q spectrogram, on the GPU.

Host-side mirror of ``osu_fusion/scripts/dataset_creator.py:17-55`` (same constant names, same ``load_audio`` entry point):
the reference computes ``np.log(np.abs(librosa.vqt(y, sr=22050, hop_length=176, fmin=C0, n_bins=96, bins_per_octave=12)) + 1e-10)``
on the CPU.  Here the wavelet bank is built once on the host (numpy, float64 -> float32) following librosa 0.10.1's published filter
definition -- lengths ``Q*sr/(f + gamma/alpha)`` with the ERB default ``gamma = 24.7*alpha/0.108``, periodic Hann window, L1
normalisation, ``sqrt(length)`` output scale -- and the transform itself is ONE strided-row fp32 MFMA GEMM over the zero-padded
waveform plus a fused |.|/log/transpose kernel (``csrc/audio.hip``).

librosa evaluates the same filters through an octave-recursive FFT approximation (soxr resampling between octaves, 1 % sparsified
one-sided spectra); this module evaluates them directly at the full sample rate.  librosa is absent from the build image and the
reference holds no spectrogram fixtures, so the front end is **parity unpinned** against the reference; it is
checked against ``oracle/vqt_oracle.py`` (fp64, FFT-domain evaluation) and analytic known answers.
"""
from __future__ import annotations

import functools
import math
import wave as _wave
from pathlib import Path
from typing import NamedTuple, Union

import numpy as np
import torch

from . import ops

SR = 22050                                        # dataset_creator.py:17
MS_PER_FRAME = 8                                  # :18
HOP_LENGTH = (SR // 1000) * MS_PER_FRAME          # :19  = 176 samples (7.98 ms)
FMIN = 440.0 * 2.0 ** ((12 - 69) / 12)            # :21  librosa.note_to_hz("C0") = 16.3516 Hz
N_OCTAVES = 8                                     # :22
OCTAVE_BINS = 12                                  # :23
AUDIO_DIM = N_OCTAVES * OCTAVE_BINS               # :24
LOG_EPS = 1e-10                                   # :52


class VQTBank(NamedTuple):
    bank: np.ndarray        # (2*bins, K) float32: rows [0, bins) real parts, [bins, 2*bins) imaginary parts, correlation form
    scale: np.ndarray       # (bins,) float32 = sqrt(filter length)
    left_pad: int           # zeros in front of the waveform so that frame t starts at sample t*hop of the padded signal
    lengths: np.ndarray     # (bins,) float64 fractional filter lengths


@functools.lru_cache(maxsize=8)
def vqt_bank(sr: int = SR, fmin: float = FMIN, n_bins: int = AUDIO_DIM, bins_per_octave: int = OCTAVE_BINS) -> VQTBank:
    """Time-domain wavelet bank in the layout csrc/audio.hip consumes."""
    freqs = fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)
    r2 = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r2 - 1.0) / (r2 + 1.0)                               # relative bandwidth of an equal-tempered bin
    shift = 24.7 / 0.108                                          # gamma / alpha for the ERB default gamma
    lengths = sr / (alpha * (freqs + shift))                      # Q * sr / (f + gamma/alpha), Q = 1/alpha
    wavelets = []
    for flen, f in zip(lengths, freqs):
        idx = np.arange(math.floor(-flen / 2), math.floor(flen / 2), dtype=np.float64)
        n = idx.size
        win = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)   # periodic Hann
        sig = np.exp(1j * 2.0 * np.pi * f / sr * idx) * win
        wavelets.append(sig / np.abs(sig).sum())
    n_fft = 1 << int(math.ceil(math.log2(lengths.max())))
    # librosa multiplies spectra (a circular convolution against the centred frame): the value at frame centre c is
    # sum_i h[i] * y[c + n_fft/2 - lpad - i]; rewritten as a correlation with the padded signal y_pad[p] = y[p - left_pad]:
    offs = [n_fft // 2 - (n_fft - h.size) // 2 for h in wavelets]
    left_pad = max(h.size - 1 - o for h, o in zip(wavelets, offs))
    K = -(-(left_pad + max(offs) + 1) // 32) * 32
    bank = np.zeros((2 * n_bins, K), dtype=np.float64)
    for k, (h, o) in enumerate(zip(wavelets, offs)):
        pos = left_pad + o - np.arange(h.size)
        bank[k, pos] = h.real
        bank[n_bins + k, pos] = h.imag
    return VQTBank(bank.astype(np.float32), np.sqrt(lengths).astype(np.float32), int(left_pad), lengths)


_DEVICE_BANKS = {}


def _device_bank(device: torch.device):
    key = str(device)
    if key not in _DEVICE_BANKS:
        b = vqt_bank()
        _DEVICE_BANKS[key] = (torch.from_numpy(b.bank).to(device), torch.from_numpy(b.scale).to(device), b.left_pad)
    return _DEVICE_BANKS[key]


def n_frames(n_samples: int, hop_length: int = HOP_LENGTH) -> int:
    """Frame count of the centred transform (librosa.stft, center=True)."""
    return 1 + n_samples // hop_length


def log_vqt(wave: Union[np.ndarray, torch.Tensor], device: Union[str, torch.device] = "cuda") -> torch.Tensor:
    """(n_samples,) mono waveform at SR -> (AUDIO_DIM, 1 + n_samples // HOP_LENGTH) fp32 log-VQT on `device` (GPU only)."""
    y = torch.as_tensor(wave)
    if y.dim() != 1:
        raise ValueError(f"expected a mono waveform, got shape {tuple(y.shape)}")
    if y.numel() == 0:
        raise ValueError("Empty audio")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("osufusion_amd.audio.log_vqt runs on the GPU only (no CPU fallback)")
    bank, scale, left_pad = _device_bank(device)
    K = bank.shape[1]
    frames = n_frames(y.numel())
    n_pad = max((frames - 1) * HOP_LENGTH + K, left_pad + y.numel())
    y_pad = torch.zeros(n_pad, dtype=torch.float32, device=device)
    y_pad[left_pad:left_pad + y.numel()] = y.to(device=device, dtype=torch.float32)
    return ops.log_vqt(y_pad, bank, scale, HOP_LENGTH, frames, LOG_EPS)


def read_wave(audio_file: Union[str, Path]) -> np.ndarray:
    """Decode a PCM .wav (or a .npy waveform already at SR) to mono float32 at SR.  The reference decodes any container through
    ffmpeg (audioread) and resamples with resampy's kaiser_best (dataset_creator.py:37-38); neither exists in this image, so
    other rates go through scipy's polyphase resampler and other containers are rejected."""
    path = Path(audio_file)
    if path.suffix == ".npy":
        return np.asarray(np.load(path), dtype=np.float32).reshape(-1)
    if path.suffix.lower() != ".wav":
        raise ValueError(f"unsupported audio container {path.suffix!r}: decode to PCM .wav first (no ffmpeg in this build)")
    with _wave.open(str(path), "rb") as f:
        ch, width, rate, n = f.getnchannels(), f.getsampwidth(), f.getframerate(), f.getnframes()
        raw = f.readframes(n)
    if width == 1:
        data = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        data = (v - ((v & 0x800000) << 1)).astype(np.float32) / 8388608.0
    elif width == 4:
        data = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    data = data.reshape(-1, ch).mean(axis=1) if ch > 1 else data
    if rate != SR:
        from scipy.signal import resample_poly
        g = math.gcd(SR, rate)
        data = resample_poly(data.astype(np.float64), SR // g, rate // g).astype(np.float32)
    return np.ascontiguousarray(data, dtype=np.float32)


def load_audio(audio_file: Union[str, Path], device: Union[str, torch.device] = "cuda") -> torch.Tensor:
    """dataset_creator.load_audio: file -> (96, n_frames) log-VQT.  Returns a device tensor (call .cpu().numpy() for the
    reference's ndarray)."""
    wave = read_wave(audio_file)
    if wave.shape[0] == 0:
        raise ValueError(f"Empty audio file: {audio_file}")
    return log_vqt(wave, device)
