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
Two evaluations live here:
  vqt_recursive   librosa 0.10.1's ACTUAL algorithm (core/constantq.py vqt): the octaves are processed from the top down, each with
                  its own wavelet bank built at the CURRENT sample rate (filters.wavelet, pad_fft=True -> n_fft = next power of two of
                  the octave's longest filter), multiplied as ONE-SIDED spectra (fft(basis * lengths / n_fft)[:, :n_fft//2+1],
                  util.sparsify_rows(quantile=0.01): the smallest entries holding 1 % of each row's L1 mass are zeroed) with the
                  rectangular-window centred STFT of the current signal (__cqt_response), scaled by sqrt(sr / my_sr); after an
                  octave, while the hop is even, hop and rate are halved and the signal is decimated by 2
                  (audio.resample(orig_sr=2, target_sr=1, res_type="soxr_hq", scale=True), i.e. low-pass, keep every other sample,
                  multiply by sqrt(2)); __early_downsample is a no-op for hop 176 / 8 octaves (num_two_factors = 4 < 8);
                  __trim_stack stacks the octaves lowest first and cuts all to the shortest frame count; V /= sqrt(lengths).
                  For hop 176: rates 22050, 11025, 5512.5, 2756.25, then 1378.125 for octaves 4-7 (hop 11 is odd).
                  The one thing that cannot be restated bit for bit is soxr's HQ low-pass itself (libsoxr is closed to this image):
                  it is a linear-phase FIR with passband to 0.913 of the new Nyquist and stopband from 1.0 at ~20-bit rejection
                  (soxr's published HQ recipe); `halfband_taps` designs a Kaiser-windowed sinc to that specification.  Every octave's
                  filters lie below 0.36 of its Nyquist, where any filter meeting the specification is flat and zero-phase, so the
                  choice moves the result by less than the stop-band leakage (measured here: 1.8e-5 of the peak between a 100 dB
                  and a 140 dB design).
  vqt             the same filters evaluated directly at the FULL sample rate for every octave (no decimation, two-sided spectra,
                  no sparsification) -- the quantity librosa's recursion approximates; kept as an independent cross-check of the
                  recursive form (they agree to ~1e-3 of the peak above the lowest two octaves, a few % below: DESIGN section 6d).
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


# ---- librosa's octave recursion --------------------------------------------------------------------------------------
def halfband_taps(passband: float = 0.913, stopband: float = 1.0, atten_db: float = 125.0) -> np.ndarray:
    """Zero-phase decimate-by-2 low-pass to soxr's HQ specification (band edges as fractions of the NEW Nyquist), Kaiser-windowed
    sinc, odd length; includes the sqrt(2) of librosa's resample(scale=True)."""
    new_nyq = 0.25                                            # cycles / sample at the old rate
    width = (stopband - passband) * new_nyq
    n = int(np.ceil((atten_db - 8.0) / (2.285 * 2.0 * np.pi * width))) | 1
    beta = 0.1102 * (atten_db - 8.7)
    fc = 0.5 * (passband + stopband) * new_nyq
    m = np.arange(n) - (n - 1) / 2
    h = 2.0 * fc * np.sinc(2.0 * fc * m) * np.kaiser(n, beta)
    return h / h.sum() * np.sqrt(2.0)


def decimate2(y: np.ndarray, taps: np.ndarray) -> np.ndarray:
    """out[m] = sum_j taps[j] * y[2m + j - (len(taps)-1)/2], m < ceil(len(y)/2) (librosa.resample's output length), zeros outside."""
    c = (len(taps) - 1) // 2
    ypad = np.concatenate([np.zeros(c), y, np.zeros(c + 1)])
    full = np.convolve(ypad, taps[::-1], mode="valid")       # full[i] = sum_j taps[j] * ypad[i + j] = response centred on y[i]
    return full[: len(y): 2][: int(np.ceil(len(y) / 2))]


def sparsify_rows(x: np.ndarray, quantile: float = 0.01) -> np.ndarray:
    """librosa.util.sparsify_rows: per row, zero the smallest-magnitude entries that together hold < quantile of the row's L1 mass."""
    out = np.zeros_like(x)
    mags = np.abs(x)
    norms = mags.sum(axis=1, keepdims=True)
    mag_sort = np.sort(mags, axis=1)
    cumulative = np.cumsum(mag_sort / norms, axis=1)
    thr = np.argmin(cumulative < quantile, axis=1)
    for i, j in enumerate(thr):
        keep = mags[i] >= mag_sort[i, j]
        out[i, keep] = x[i, keep]
    return out


def octave_plan(hop: int = HOP_LENGTH, n_bins: int = N_BINS, bpo: int = BPO):
    """(decimations before octave i, hop at octave i) for i = 0 (top octave) .. n_octaves-1."""
    n_oct = int(np.ceil(n_bins / bpo))
    plan, dec, h = [], 0, hop
    for _ in range(n_oct):
        plan.append((dec, h))
        if h % 2 == 0:
            h //= 2
            dec += 1
    return plan


def octave_fft_basis(freqs_oct: np.ndarray, my_sr: float, sr: float, bpo: int = BPO, sparsity: float = 0.01):
    """__vqt_filter_fft + the sqrt(sr / my_sr) rescale: one-sided sparsified spectra (n_filters, n_fft//2 + 1) and n_fft."""
    lengths = wavelet_lengths(freqs_oct, my_sr, bpo)
    n_fft = int(2.0 ** np.ceil(np.log2(lengths.max())))
    basis = np.stack([pad_center(w, n_fft) for w in wavelets(freqs_oct, my_sr, bpo)]) * (lengths[:, None] / n_fft)
    fb = np.fft.fft(basis, axis=1)[:, : n_fft // 2 + 1]
    if sparsity > 0:
        fb = sparsify_rows(fb, sparsity)
    return fb * np.sqrt(sr / my_sr), n_fft


def vqt_recursive(y: np.ndarray, sr: float = SR, hop: int = HOP_LENGTH, fmin: float = FMIN, n_bins: int = N_BINS, bpo: int = BPO,
                  sparsity: float = 0.01, taps: np.ndarray = None) -> np.ndarray:
    """librosa.vqt(y, sr, hop_length=hop, fmin, n_bins, bins_per_octave=bpo) restated (see the module docstring), fp64."""
    y = np.asarray(y, dtype=np.float64)
    taps = halfband_taps() if taps is None else taps
    freqs = fmin * 2.0 ** (np.arange(n_bins) / bpo)
    n_filters = min(bpo, n_bins)
    resp = []
    my_y, my_sr, my_hop = y, float(sr), hop
    for i in range(int(np.ceil(n_bins / bpo))):
        sl = slice(-n_filters, None) if i == 0 else slice(-n_filters * (i + 1), -n_filters * i)
        fb, n_fft = octave_fft_basis(freqs[sl], my_sr, sr, bpo, sparsity)
        ypad = np.pad(my_y, n_fft // 2)
        frames = 1 + len(my_y) // my_hop
        out = np.empty((fb.shape[0], frames), dtype=np.complex128)
        for t0 in range(0, frames, 1024):
            t1 = min(frames, t0 + 1024)
            seg = np.stack([ypad[t * my_hop:t * my_hop + n_fft] for t in range(t0, t1)], axis=1)
            out[:, t0:t1] = fb @ np.fft.rfft(seg, axis=0)
        resp.append(out)
        if my_hop % 2 == 0:
            my_hop //= 2
            my_sr /= 2.0
            my_y = decimate2(my_y, taps)
    max_col = min(r.shape[1] for r in resp)                  # __trim_stack
    V = np.empty((n_bins, max_col), dtype=np.complex128)
    end = n_bins
    for r in resp:
        n_oct = r.shape[0]
        if end < n_oct:
            V[:end] = r[-end:, :max_col]
        else:
            V[end - n_oct:end] = r[:, :max_col]
        end -= n_oct
    return V / np.sqrt(wavelet_lengths(freqs, sr, bpo))[:, None]


def log_vqt_recursive(y: np.ndarray, **kw) -> np.ndarray:
    return np.log(np.abs(vqt_recursive(y, **kw)) + 1e-10)
