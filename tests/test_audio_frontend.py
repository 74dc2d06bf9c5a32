"""Audio front end: waveform -> (96, n_frames) log-VQT.  Reference: scripts/dataset_creator.py:17-55
(np.log(np.abs(librosa.vqt(...)) + 1e-10)); librosa is absent and the reference has no spectrogram fixtures, so the oracle
(oracle/vqt_oracle.py, fp64: `vqt_recursive` = librosa 0.10.1's octave recursion restated, `vqt` = the same filters evaluated
directly at the full rate) is checked against analytic known answers and against itself here, and the HIP paths against the
oracle: A.log_vqt / A.load_audio (the product path) vs vqt_recursive, A.log_vqt_direct vs vqt.  Tolerances: linear magnitudes
within 2e-5 of the spectrogram's peak, log features within 1e-3 wherever the magnitude is above 1e-3 of the peak (fp32
accumulation)."""
import wave

import numpy as np
import pytest
import torch

from oracle import vqt_oracle as VO
from osufusion_amd import audio as A


def _tone(k, seconds=1.0):
    f = VO.FMIN * 2.0 ** (k / 12)
    t = np.arange(int(VO.SR * seconds)) / VO.SR
    return np.cos(2 * np.pi * f * t), f


def test_constants_match_the_reference_front_end():
    assert (A.SR, A.HOP_LENGTH, A.AUDIO_DIM, A.N_OCTAVES, A.OCTAVE_BINS) == (22050, 176, 96, 8, 12)     # dataset_creator.py:17-24
    assert abs(A.FMIN - 16.351597831287414) < 1e-12                                                       # note_to_hz("C0")
    assert A.n_frames(4096 * 176 - 1) == 4096 and A.n_frames(0) == 1


@pytest.mark.parametrize("k", [36, 48, 90])
def test_oracle_known_answer_pure_tone(k):
    """A unit cosine at a bin centre answers sqrt(length)/2 in that bin (L1-normalised wavelet, sqrt(length) scale; half the
    energy sits at -f) and that bin is the spectrogram's maximum."""
    y, f = _tone(k)
    v = np.abs(VO.vqt(y))
    mid = v.shape[1] // 2
    want = np.sqrt(VO.wavelet_lengths(np.array([f]), VO.SR))[0] / 2
    assert v[:, mid].argmax() == k
    assert abs(v[k, mid] - want) / want < 2e-3


def test_oracle_filter_lengths_and_frame_count():
    freqs = VO.FMIN * 2.0 ** (np.arange(96) / 12)
    lens = VO.wavelet_lengths(freqs, VO.SR)
    assert abs(lens[0] - 1559.4911) < 1e-3 and abs(lens[-1] - 91.43124) < 1e-4 and np.all(np.diff(lens) < 0)
    for n in (1, 175, 176, 177, 1000):
        assert VO.vqt(np.ones(n)).shape == (96, 1 + n // 176)


def test_oracle_hop_shift_equivariance():
    rng = np.random.default_rng(3)
    y = rng.standard_normal(8000)
    v0 = VO.vqt(np.concatenate([y, np.zeros(176 * 3)]))
    v1 = VO.vqt(np.concatenate([np.zeros(176 * 3), y]))
    np.testing.assert_allclose(v1[:, 3:], v0[:, :-3], atol=1e-12)


@pytest.mark.parametrize("k", [7, 36, 48, 90])
def test_recursive_oracle_known_answer_pure_tone(k):
    """librosa's recursion (decimated octaves, one-sided 1 %-sparsified spectra) answers the same sqrt(length)/2 to 1 %."""
    y, f = _tone(k, 2.0)
    v = np.abs(VO.vqt_recursive(y))
    mid = v.shape[1] // 2
    want = np.sqrt(VO.wavelet_lengths(np.array([f]), VO.SR))[0] / 2
    assert v[:, mid].argmax() == k
    assert abs(v[k, mid] - want) / want < 1e-2


def test_recursive_oracle_structure():
    """Octave plan for hop 176 (176 -> 88 -> 44 -> 22 -> 11, then no further halving), frame counts, hop-shift equivariance for
    shifts that are multiples of 16 samples (the total decimation), and agreement with the direct evaluation."""
    assert VO.octave_plan() == [(0, 176), (1, 88), (2, 44), (3, 22), (4, 11), (4, 11), (4, 11), (4, 11)]
    for n in (1, 175, 176, 177, 1000):
        assert VO.vqt_recursive(np.ones(n)).shape == (96, 1 + n // 176)
    taps = VO.halfband_taps()
    assert len(taps) % 2 == 1 and abs(taps.sum() - np.sqrt(2.0)) < 1e-12 and np.allclose(taps, taps[::-1])
    w = np.fft.rfft(taps, 8192) / np.sqrt(2.0)                    # passband flat to 0.913 of the new Nyquist, stopband from 1.0
    fgrid = np.arange(len(w)) / 8192.0
    assert np.abs(np.abs(w[fgrid <= 0.913 * 0.25]) - 1).max() < 1e-5 and np.abs(w[fgrid >= 0.25]).max() < 10 ** (-120 / 20)
    rng = np.random.default_rng(3)
    y = rng.standard_normal(12000)
    v0 = VO.vqt_recursive(np.concatenate([y, np.zeros(176 * 3)]))
    v1 = VO.vqt_recursive(np.concatenate([np.zeros(176 * 3), y]))
    # (interior frames: at the buffer's ends the decimated signals are cut to ceil(n/2) samples, as librosa.resample cuts them)
    np.testing.assert_allclose(v1[:, 3 + 30:-30], v0[:, 30:-3 - 30], atol=1e-9)
    direct = VO.vqt(y)
    rec = VO.vqt_recursive(y)
    peak = np.abs(direct).max()
    # librosa's recursion approximates the direct evaluation: 1 % of every filter's spectral mass is dropped (sparsity=0.01) and the
    # negative-frequency half of the spectrum is ignored (the lowest octave's filters straddle 0 Hz) -- on white noise a few % of
    # the peak, more in the first / last frames where the decimated signals are cut
    assert np.abs(rec - direct)[:, 10:-10].max() < 5e-2 * peak


def test_plan_is_the_recursive_oracle_in_correlation_form():
    """Host logic: the per-octave-group fp32 correlation kernels and the decimation taps the GPU path consumes reproduce the
    recursive oracle (emulated here in numpy, fp64 accumulation)."""
    plan = A.librosa_plan()
    assert [(g.decimations, g.hop, g.n_fft, g.bin0, g.bank.shape[0] // 2) for g in plan.groups] == \
        [(0, 176, 256, 84, 12), (1, 88, 256, 72, 12), (2, 44, 128, 60, 12), (3, 22, 128, 48, 12), (4, 11, 128, 0, 48)]
    assert np.allclose(plan.taps, VO.halfband_taps(), atol=1e-7)
    rng = np.random.default_rng(1)
    n = 9000
    y = rng.standard_normal(n) * 0.1 + np.sin(2 * np.pi * 220 * np.arange(n) / VO.SR)
    ref = VO.vqt_recursive(y)
    frames = 1 + n // 176
    out = np.zeros((96, frames), dtype=np.complex128)
    sig, level = y.copy(), 0
    for g in plan.groups:
        while level < g.decimations:
            sig, level = VO.decimate2(sig, plan.taps.astype(np.float64)), level + 1
        K = g.n_fft
        pad = np.zeros(max((frames - 1) * g.hop + K, K // 2 + len(sig)))
        pad[K // 2:K // 2 + len(sig)] = sig
        nb = g.bank.shape[0] // 2
        kern = g.bank[:nb].astype(np.float64) + 1j * g.bank[nb:].astype(np.float64)
        out[g.bin0:g.bin0 + nb] = kern @ np.stack([pad[t * g.hop:t * g.hop + K] for t in range(frames)], axis=1)
    assert np.abs(out - ref).max() < 1e-6 * np.abs(ref).max()


def test_bank_is_the_oracle_transform_in_correlation_form():
    """Host logic: the fp32 time-domain bank the GEMM consumes reproduces the oracle's spectral-product evaluation."""
    rng = np.random.default_rng(0)
    y = rng.standard_normal(5000) * 0.1
    b = A.vqt_bank()
    assert b.bank.shape == (192, 1568) and b.bank.dtype == np.float32 and b.left_pad == 778
    K, hop = b.bank.shape[1], A.HOP_LENGTH
    frames = A.n_frames(len(y))
    ypad = np.zeros(max((frames - 1) * hop + K, b.left_pad + len(y)))
    ypad[b.left_pad:b.left_pad + len(y)] = y
    rows = np.stack([ypad[t * hop:t * hop + K] for t in range(frames)])
    s = rows @ b.bank.astype(np.float64).T
    got = (s[:, :96] + 1j * s[:, 96:]).T * b.scale[:, None].astype(np.float64)
    want = VO.vqt(y)
    assert np.abs(got - want).max() / np.abs(want).max() < 1e-6


def test_wave_reader(tmp_path):
    rng = np.random.default_rng(1)
    pcm = (rng.uniform(-0.5, 0.5, (44100, 2)) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "s.wav"), "wb") as f:
        f.setnchannels(2)
        f.setsampwidth(2)
        f.setframerate(44100)
        f.writeframes(pcm.tobytes())
    y = A.read_wave(tmp_path / "s.wav")
    assert y.dtype == np.float32 and y.shape == (22050,) and np.abs(y).max() < 1.0
    mono = (rng.uniform(-1, 1, 3000) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "m.wav"), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(22050)
        f.writeframes(mono.tobytes())
    np.testing.assert_array_equal(A.read_wave(tmp_path / "m.wav"), mono.astype(np.float32) / 32768.0)
    np.save(tmp_path / "w.npy", y)
    np.testing.assert_array_equal(A.read_wave(tmp_path / "w.npy"), y)
    with pytest.raises(ValueError, match="unsupported audio container"):
        A.read_wave(tmp_path / "song.mp3")


def test_errors_without_gpu_work():
    with pytest.raises(ValueError, match="Empty audio"):
        A.log_vqt(np.zeros(0, np.float32))
    with pytest.raises(ValueError, match="mono"):
        A.log_vqt(np.zeros((2, 10), np.float32))
    with pytest.raises(RuntimeError, match="GPU only"):
        A.log_vqt(np.zeros(10, np.float32), device="cpu")


# ------------------------------------------------------------------------------------------------------------------ GPU
def _check(y, direct=False):
    got = (A.log_vqt_direct(y) if direct else A.log_vqt(y)).cpu().numpy().astype(np.float64)
    v = np.abs(VO.vqt(y) if direct else VO.vqt_recursive(y))
    assert got.shape == v.shape
    peak = v.max()
    mag = np.exp(got) - 1e-10
    assert np.abs(mag - v).max() <= 2e-5 * peak                          # linear magnitudes
    big = v > 1e-3 * peak
    assert np.abs(got[big] - np.log(v[big] + 1e-10)).max() < 1e-3        # log features where they carry signal
    return got


@pytest.mark.gpu
def test_hip_log_vqt_matches_oracle_noise_and_tones():
    rng = np.random.default_rng(7)
    n = 30000
    y = 0.05 * rng.standard_normal(n)
    for k, amp in ((5, 0.5), (40, 0.3), (77, 0.2)):
        y[: n] += amp * _tone(k, n / VO.SR)[0][:n]
    y[12000:12010] += 0.8                                                # a click
    _check(y.astype(np.float32))                                         # the product path: librosa's recursion
    _check(y.astype(np.float32), direct=True)                            # the full-rate direct form


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 175, 176, 177, 2047, 2048, 11264])
def test_hip_log_vqt_ragged_lengths(n):
    rng = np.random.default_rng(n)
    got = _check(rng.standard_normal(n).astype(np.float32))
    assert got.shape == (96, 1 + n // 176)
    _check(rng.standard_normal(n).astype(np.float32), direct=True)


@pytest.mark.gpu
def test_hip_log_vqt_silence_and_tone_known_answers():
    out = A.log_vqt(np.zeros(5000, np.float32))
    assert (out - float(np.log(1e-10))).abs().max() < 1e-5                            # silence -> log(1e-10) = -23.02585
    y, f = _tone(60, 1.0)
    v = A.log_vqt(y.astype(np.float32)).cpu().numpy()
    mid = v.shape[1] // 2
    assert v[:, mid].argmax() == 60
    assert abs(v[60, mid] - np.log(np.sqrt(VO.wavelet_lengths(np.array([f]), VO.SR))[0] / 2)) < 1e-2


@pytest.mark.gpu
def test_hip_log_vqt_full_song_properties():
    """Three minutes of audio (size-independent checks): frame count, finiteness, hop-shift equivariance, and agreement with
    the oracle on a window of frames cut from the middle."""
    g = torch.Generator().manual_seed(5)
    n = 180 * A.SR
    y = (torch.randn(n, generator=g) * 0.1).numpy()
    out = A.log_vqt(y)
    assert out.shape == (96, 1 + n // 176) and torch.isfinite(out).all()
    shifted = A.log_vqt(np.concatenate([np.zeros(176 * 5, np.float32), y]))
    nf = out.shape[1]
    assert torch.allclose(shifted[:, 5 + 10:5 + nf - 10], out[:, 10:nf - 10], atol=2e-4)
    t0, nt, lead = 9000, 40, 40                         # oracle on a cut that gives frames t0..t0+nt their full support: the longest
    seg = y[(t0 - lead) * 176:(t0 + nt + lead) * 176]   # frame (2,048 samples) + the four decimation filters' tails (187 x 15 samples)
    ref = np.log(np.abs(VO.vqt_recursive(seg)) + 1e-10)
    np.testing.assert_allclose(out[:, t0:t0 + nt].cpu().numpy(), ref[:, lead:lead + nt], atol=2e-3)


@pytest.mark.gpu
def test_load_audio_feeds_the_denoiser_shape(tmp_path):
    rng = np.random.default_rng(2)
    pcm = (rng.uniform(-0.3, 0.3, 22050 * 2) * 32767).astype("<i2")
    with wave.open(str(tmp_path / "a.wav"), "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(22050)
        f.writeframes(pcm.tobytes())
    a = A.load_audio(tmp_path / "a.wav")
    assert a.is_cuda and a.shape == (96, 1 + 44100 // 176) and a.dtype == torch.float32
    with pytest.raises(ValueError, match="Empty audio file"):
        with wave.open(str(tmp_path / "e.wav"), "wb") as f:
            f.setnchannels(1)
            f.setsampwidth(2)
            f.setframerate(22050)
            f.writeframes(b"")
        A.load_audio(tmp_path / "e.wav")
