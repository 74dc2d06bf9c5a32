"""Audio front end: waveform -> (96, n_frames) log-magnitude variable-Q spectrogram, on the GPU.

Host-side mirror of ``osu_fusion/scripts/dataset_creator.py:17-55`` (same constant names, same ``load_audio`` entry point):
the reference computes ``np.log(np.abs(librosa.vqt(y, sr=22050, hop_length=176, fmin=C0, n_bins=96, bins_per_octave=12)) + 1e-10)``
on the CPU.  Here the wavelet bank is built once on the host (numpy, float64 -> float32) following librosa 0.10.1's published filter
definition -- lengths ``Q*sr/(f + gamma/alpha)`` with the ERB default ``gamma = 24.7*alpha/0.108``, periodic Hann window, L1
normalisation, ``sqrt(length)`` output scale -- and the transform itself is ONE strided-row fp32 MFMA GEMM over the zero-padded
waveform plus a fused |.|/log/transpose kernel (``csrc/audio.hip``).

``load_audio`` / ``log_vqt`` follow librosa 0.10.1's ACTUAL evaluation of these filters (``librosa_plan`` below; oracle:
``oracle.vqt_oracle.vqt_recursive``): octave by octave from the top, each octave's wavelets built at the current sample rate, kept
as one-sided spectra with the smallest 1 % of their L1 mass dropped (``util.sparsify_rows``) and applied to centred rectangular
frames; between octaves, while the hop is even (176 -> 88 -> 44 -> 22 -> 11), the signal is low-passed and decimated by 2 with a
sqrt(2) gain (``resample(orig_sr=2, target_sr=1, res_type="soxr_hq", scale=True)``).  A product of one-sided spectra is a
correlation of the frame with the kernel g[n] = sum_{k <= n_fft/2} B[k] e^{-2 pi i k n / n_fft}; those kernels are built on the host
(numpy fp64) and every octave group becomes one strided-row fp32 MFMA GEMM (``osuf_log_vqt``), after ``osuf_fir_decimate2``.
soxr itself is not in this image: the half-band filter is a Kaiser-windowed sinc to soxr's published HQ recipe (passband to 0.913
of the new Nyquist, ~125 dB) -- every octave's filters sit below 0.36 of its Nyquist, where any such filter is flat, so its exact
taps move the result by ~1e-5 of the peak.  ``log_vqt_direct`` keeps round 1's full-rate direct form (the quantity the recursion
approximates: differs by up to 2 % of the peak in the three lowest octaves).  librosa is absent from the build image and the
reference holds no spectrogram fixtures, so the front end is **parity unpinned** against the reference.
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


class OctaveGroup(NamedTuple):
    decimations: int        # halvings of the sample rate before this group
    hop: int                # hop at that rate
    n_fft: int              # frame length (= K of the GEMM)
    bin0: int               # first output bin (row) of the group
    bank: np.ndarray        # (2*nb, n_fft) float32: real rows then imaginary rows of the correlation kernels, 1/sqrt(length) folded in


class LibrosaPlan(NamedTuple):
    groups: tuple           # OctaveGroup, top octave first
    taps: np.ndarray        # half-band decimation filter (float32, odd length, includes sqrt(2))


def halfband_taps(passband: float = 0.913, stopband: float = 1.0, atten_db: float = 125.0) -> np.ndarray:
    """Zero-phase decimate-by-2 low-pass to soxr's HQ recipe (edges as fractions of the new Nyquist): Kaiser-windowed sinc."""
    width = (stopband - passband) * 0.25
    n = int(math.ceil((atten_db - 8.0) / (2.285 * 2.0 * math.pi * width))) | 1
    fc = 0.5 * (passband + stopband) * 0.25
    m = np.arange(n) - (n - 1) / 2
    h = 2.0 * fc * np.sinc(2.0 * fc * m) * np.kaiser(n, 0.1102 * (atten_db - 8.7))
    return h / h.sum() * math.sqrt(2.0)


def _sparsify_rows(x: np.ndarray, quantile: float) -> np.ndarray:
    out = np.zeros_like(x)
    mags = np.abs(x)
    srt = np.sort(mags, axis=1)
    cum = np.cumsum(srt / mags.sum(axis=1, keepdims=True), axis=1)
    for i, j in enumerate(np.argmin(cum < quantile, axis=1)):
        keep = mags[i] >= srt[i, j]
        out[i, keep] = x[i, keep]
    return out


@functools.lru_cache(maxsize=4)
def librosa_plan(sr: int = SR, hop: int = HOP_LENGTH, fmin: float = FMIN, n_bins: int = AUDIO_DIM, bins_per_octave: int = OCTAVE_BINS,
                 sparsity: float = 0.01) -> LibrosaPlan:
    """The octave recursion of librosa.vqt as GEMM operands: consecutive octaves that share (rate, hop, n_fft) form one group."""
    freqs = fmin * 2.0 ** (np.arange(n_bins, dtype=np.float64) / bins_per_octave)
    r2 = 2.0 ** (2.0 / bins_per_octave)
    alpha = (r2 - 1.0) / (r2 + 1.0)
    shift = 24.7 / 0.108

    def lengths_at(f, rate):
        return rate / (alpha * (f + shift))

    n_oct = -(-n_bins // bins_per_octave)
    nf = min(bins_per_octave, n_bins)
    groups, dec, my_hop = [], 0, hop
    for i in range(n_oct):
        lo, hi = max(0, n_bins - nf * (i + 1)), n_bins - nf * i
        my_sr = sr / 2.0 ** dec
        f_oct = freqs[lo:hi]
        lens = lengths_at(f_oct, my_sr)
        n_fft = 1 << int(math.ceil(math.log2(lens.max())))
        basis = np.zeros((hi - lo, n_fft), dtype=np.complex128)
        for k, (flen, f) in enumerate(zip(lens, f_oct)):
            idx = np.arange(math.floor(-flen / 2), math.floor(flen / 2), dtype=np.float64)
            n = idx.size
            sig = np.exp(1j * 2.0 * np.pi * f / my_sr * idx) * (0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))
            sig /= np.abs(sig).sum()
            lp = (n_fft - n) // 2
            basis[k, lp:lp + n] = sig * (flen / n_fft)
        fb = np.fft.fft(basis, axis=1)[:, : n_fft // 2 + 1]
        if sparsity > 0:
            fb = _sparsify_rows(fb, sparsity)
        fb = fb * math.sqrt(sr / my_sr) / np.sqrt(lengths_at(f_oct, sr))[:, None]            # + the final V /= sqrt(lengths)
        kk, nn = np.arange(n_fft // 2 + 1)[:, None], np.arange(n_fft)[None, :]
        g = fb @ np.exp(-2j * np.pi * kk * nn / n_fft)                                        # (nb, n_fft) correlation kernels
        last = groups[-1] if groups else None
        if last is not None and (last["dec"], last["hop"], last["n_fft"]) == (dec, my_hop, n_fft):
            last["g"] = np.concatenate([g, last["g"]], axis=0)                                # lower octave = lower bins: prepend
            last["bin0"] = lo
        else:
            groups.append(dict(dec=dec, hop=my_hop, n_fft=n_fft, bin0=lo, g=g))
        if my_hop % 2 == 0:
            my_hop //= 2
            dec += 1
    out = tuple(OctaveGroup(d["dec"], d["hop"], d["n_fft"], d["bin0"],
                            np.concatenate([d["g"].real, d["g"].imag], axis=0).astype(np.float32)) for d in groups)
    return LibrosaPlan(out, halfband_taps().astype(np.float32))


_DEVICE_PLANS = {}


def _device_plan(device: torch.device):
    key = str(device)
    if key not in _DEVICE_PLANS:
        plan = librosa_plan()
        _DEVICE_PLANS[key] = ([(g, torch.from_numpy(g.bank).to(device), torch.ones(g.bank.shape[0] // 2, dtype=torch.float32, device=device))
                               for g in plan.groups], torch.from_numpy(plan.taps).to(device))
    return _DEVICE_PLANS[key]


def log_vqt(wave: Union[np.ndarray, torch.Tensor], device: Union[str, torch.device] = "cuda") -> torch.Tensor:
    """(n_samples,) mono waveform at SR -> (AUDIO_DIM, 1 + n_samples // HOP_LENGTH) fp32 log-VQT on `device` (GPU only), evaluated
    as librosa.vqt evaluates it (octave recursion; see the module docstring)."""
    y = torch.as_tensor(wave)
    if y.dim() != 1:
        raise ValueError(f"expected a mono waveform, got shape {tuple(y.shape)}")
    if y.numel() == 0:
        raise ValueError("Empty audio")
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("osufusion_amd.audio.log_vqt runs on the GPU only (no CPU fallback)")
    groups, taps = _device_plan(device)
    frames = n_frames(y.numel())                                   # the top octave has the fewest frames (__trim_stack)
    out = torch.empty((AUDIO_DIM, frames), dtype=torch.float32, device=device)
    sig = y.to(device=device, dtype=torch.float32).contiguous()
    level = 0
    for g, bank, ones in groups:
        while level < g.decimations:                               # resample(orig_sr=2, target_sr=1, "soxr_hq", scale=True)
            sig = ops.fir_decimate2(sig, taps)
            level += 1
        n, K = sig.numel(), g.n_fft
        n_pad = (frames - 1) * g.hop + K                           # centred frames: n_fft/2 zeros in front, zeros behind
        pad = torch.zeros(max(n_pad, K // 2 + n), dtype=torch.float32, device=device)
        pad[K // 2:K // 2 + n] = sig
        nb = bank.shape[0] // 2
        rows = out[g.bin0:g.bin0 + nb]
        if g.hop % 4 == 0:
            ops.log_vqt(pad, bank, ones, g.hop, frames, LOG_EPS, out=rows)
        else:                                                      # hop 22 / 11: frame explicitly, then a dense GEMM
            ops.log_vqt(ops.frame_rows(pad, g.hop, K, frames).reshape(-1), bank, ones, K, frames, LOG_EPS, out=rows)
    return out


def log_vqt_direct(wave: Union[np.ndarray, torch.Tensor], device: Union[str, torch.device] = "cuda") -> torch.Tensor:
    """The same wavelets evaluated directly at the full rate for every bin (one GEMM; the quantity librosa's recursion approximates)."""
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
