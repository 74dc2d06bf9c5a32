"""decode_beatmap and its helpers (reference: osu_fusion/library/osu/data/decode.py:62-237,
data/hit.py:24-28,57-74, data/fit_bezier.py).  The reference's decode path cannot be imported here (it needs the `bezier` package)
and ships no fixtures, so these are known-answer and property tests: the peak pickers against scipy.signal.find_peaks (which the
reference calls), Bezier evaluation / length against closed forms, the curve fit against its own error bound, tempo recovery on a
synthetic onset train, and an end-to-end decode of a hand-encoded map (circle, repeating slider, spinner)."""
import numpy as np
import pytest
from scipy import signal

from osufusion_amd import decode as D


def test_peak_pickers_match_scipy():
    rng = np.random.default_rng(0)
    for _ in range(20):
        x = np.round(rng.standard_normal(200), 1)                     # rounding creates plateaus
        assert D._peaks_above(x, 0.5) == signal.find_peaks(x, height=0.5)[0].tolist()
        assert D._peaks_at_distance(x, 7.3) == signal.find_peaks(x, distance=7.3)[0].tolist()
    flips = np.array([-1.0] * 5 + [1.0] * 7 + [-1.0] * 3 + [1.0] * 4)
    g = np.gradient(flips)
    want = sorted(signal.find_peaks(g, height=0.5)[0].tolist() + signal.find_peaks(-g, height=0.5)[0].tolist())
    assert D.decode_flips(flips) == want == [4, 11, 14]


def test_decode_extents_pairs_runs():
    x = np.array([-1, -1, 1, 1, 1, -1, -1, 1, 1, -1, 1, 1], dtype=float)
    starts, ends = D.decode_extents(x)                                 # the last run never closes: its start is kept, unpaired
    assert (starts, ends) == ([1, 6, 9], [4, 8])
    assert D.decode_extents(np.array([1, 1, -1, -1, 1, 1, -1.0])) == ([3], [5])      # a run already open at frame 0 has no start
    assert D.decode_extents(-np.ones(8)) == ([], [])


def test_bezier_evaluation_and_length():
    line = np.array([[0.0, 0.0], [30.0, 40.0]])
    assert np.allclose(D.bezier_points(line, np.array([0.0, 0.5, 1.0])), [[0, 0], [15, 20], [30, 40]])
    assert abs(D.get_segment_length(line) - 50.0) < 1e-12
    cubic_line = np.array([[0.0, 0.0], [10.0, 0.0], [20.0, 0.0], [30.0, 0.0]])
    assert abs(D.get_segment_length(cubic_line) - 30.0) < 1e-12
    ctrl = np.array([[0.0, 0.0], [100.0, 200.0], [250.0, -80.0], [300.0, 120.0]])
    t = np.linspace(0, 1, 7)
    bern = np.stack([(1 - t) ** 3, 3 * (1 - t) ** 2 * t, 3 * (1 - t) * t ** 2, t ** 3], axis=1) @ ctrl
    assert np.allclose(D.bezier_points(ctrl, t), bern)
    fine = D.bezier_points(ctrl, np.linspace(0, 1, 200001))
    assert abs(D.get_segment_length(ctrl) - np.linalg.norm(np.diff(fine, axis=0), axis=1).sum()) < 1e-6


def test_fit_bezier_line_arc_and_split():
    t = np.linspace(0, 1, 40)
    line = np.stack([10 + 300 * t, 20 + 100 * t], axis=1)
    (seg,) = D.fit_bezier(line, max_err=50.0)
    assert seg.shape == (2, 2) and np.allclose(seg, line[[0, -1]])                  # a straight run collapses to its end points
    arc = 100 * np.stack([np.cos(t * np.pi / 2), np.sin(t * np.pi / 2)], axis=1) + 200
    segs = D.fit_bezier(arc, max_err=50.0)
    assert 1 <= len(segs) <= 2 and all(s.shape == (4, 2) for s in segs)
    dense = np.concatenate([D.bezier_points(s, np.linspace(0, 1, 400)) for s in segs])
    dist = np.min(np.linalg.norm(arc[:, None] - dense[None], axis=2), axis=1)
    assert dist.max() ** 2 < 50.0                                                     # the fit's own bound (squared osu!pixels)
    wiggle = np.stack([500 * t, 150 * np.sin(6 * np.pi * t) + 190], axis=1)           # three periods: one cubic cannot follow
    segs = D.fit_bezier(wiggle, max_err=50.0)
    assert len(segs) >= 3 and np.allclose(segs[0][0], wiggle[0]) and np.allclose(segs[-1][-1], wiggle[-1])
    for s0, s1 in zip(segs[:-1], segs[1:]):
        assert np.allclose(s0[-1], s1[0])                                             # pieces join at a data point
    assert D.fit_bezier(wiggle[:1], 50.0) == []


def _timing_with_scipy(hit_times):
    """decode.py:89-121 written with the scipy calls the reference makes (the lag axis of its autocorrelation counts ONSETS, which it
    then reads as milliseconds -- that is the reference's estimator, quirk included)."""
    diffs = np.diff(hit_times)
    ac = signal.correlate(diffs, diffs, mode="full")
    ac = ac[len(ac) // 2:]
    periods = 60000 / np.arange(1, 301)
    peaks, _ = signal.find_peaks(ac, distance=periods.min())
    valid = peaks[(periods.min() * 0.95 <= peaks) & (peaks <= periods.max() * 1.05)]
    if len(valid) == 0:
        return None
    bpm0 = 60000 / valid[np.argmax(ac[valid])]
    scan = np.linspace(bpm0 * 0.95, bpm0 * 1.05, 1000)
    scores = [np.histogram(hit_times % (60000 / b), bins=100, range=(0, 60000 / b))[0].max() for b in scan]
    return 60000 / scan[int(np.argmax(scores))]


def test_timing_point_estimator_and_snap():
    rng = np.random.default_rng(4)
    beat = 60000 / 180.0
    ticks = np.sort(rng.choice(np.arange(0, 3000), size=1500, replace=False))          # onsets on a quarter-beat grid, 1,500 of them
    hit_times = 37.0 + ticks * (beat / 4) + rng.uniform(-2, 2, size=len(ticks))
    want = _timing_with_scipy(hit_times)
    snap, tp = D.calculate_timing_point(hit_times, True, verbose=False)
    assert want is not None and snap and tp.meter == 4 and abs(tp.beat_length - want) < 1e-9 * want
    short = hit_times[:150]                                                            # fewer onsets than the smallest admissible lag (190)
    assert _timing_with_scipy(short) is None
    assert D.calculate_timing_point(short, True, verbose=False) == (False, D.TimingPoint(0, 300.0, 4))
    assert D.calculate_timing_point(hit_times, False) == (False, D.TimingPoint(0, 300.0, 4))
    ok, tp2 = D.get_timings(hit_times, beat / 4)                                       # given the tempo, the grid offset is recovered
    hist, edges = np.histogram(hit_times % (beat / 4), bins=100, range=(0, beat / 4))
    assert ok and tp2.t == edges[np.argmax(hist)] and abs(tp2.t - 37.0) < 3.0
    assert D.snap_to_beat(1003.0, 1510.0, 0.0, 320.0) == (1000.0, 1520.0)              # 1/16 of 320 ms = 20 ms ticks


def _encode(L, circles, sliders, spinners, combos, path):
    """A hand-rolled encoder following data/hit.py / data/encode.py: flips toggle at every onset, extents are +1 while they last."""
    enc = -np.ones((6, L))
    state = -1.0
    for f in sorted(circles + [s[0] for s in sliders] + [s[0] for s in spinners]):
        state = -state
        enc[D.HIT, f:] = state
    cstate = -1.0
    for f in sorted(combos):
        cstate = -cstate
        enc[D.COMBO, f:] = cstate
    for s, e, first in sliders:
        enc[D.SUSTAIN, s:e] = 1.0
        enc[D.SLIDER, s:first] = 1.0
    for s, e in spinners:
        enc[D.SUSTAIN, s:e] = 1.0
    enc[D.CURSOR_X], enc[D.CURSOR_Y] = path
    return enc


def test_decode_beatmap_end_to_end():
    L = 1000
    frame_times = np.arange(L) * 8.0                                                  # 8 ms frames
    t = np.arange(L)
    path = (np.clip(-0.8 + t / 600.0, -1, 1), 0.5 * np.sin(t / 90.0))
    enc = _encode(L, circles=[100, 200], sliders=[(300, 380, 340)], spinners=[(600, 700)], combos=[100, 600], path=path)
    enc[:4] *= 0.7                                                                    # the sampler's output is not exactly +-1
    meta = D.Metadata("audio.mp3", "Song", "Artist", "Version", cs=4.0, ar=9.0, od=8.0, hp=5.0)
    text = D.decode_beatmap(meta, enc, frame_times, bpm=187.5, verbose=False)         # 320 ms beats: 20 ms snap ticks
    assert text.startswith("osu file format v14\n\n[General]\nAudioFilename: audio.mp3\nAudioLeadIn: 0\nMode: 0\n\n[Metadata]\nTitle: Song\n")
    assert "[Difficulty]\nHPDrainRate: 5.0\nCircleSize: 4.0\nOverallDifficulty: 8.0\nApproachRate: 9.0\nSliderMultiplier: 1\nSliderTickRate: 1\n" in text
    tps = text.split("[TimingPoints]\n")[1].split("\n\n[HitObjects]")[0].splitlines()
    hos = text.split("[HitObjects]\n")[1].strip().splitlines()
    assert tps[0].endswith(",320.0,4,0,0,50,1,0") and len(tps) == 2 and tps[1].endswith(",4,0,0,50,0,0")   # + one inherited point: the slider
    assert len(hos) == 4
    kinds = [int(h.split(",")[3]) for h in hos]
    assert kinds == [1 + 4, 1, 2, 8 + 4]                                              # circle (new combo), circle, slider, spinner (new combo)
    times = [float(h.split(",")[2]) for h in hos]
    offset = float(tps[0].split(",")[0])
    for got, frame in zip(times, (100, 200, 300, 600)):
        assert abs(got - frame * 8.0) <= 10.0 + 8.0 and abs(((got - offset) / 20.0) - round((got - offset) / 20.0)) < 1e-6
    x0, y0 = (int(v) for v in hos[0].split(",")[:2])
    assert (x0, y0) == (round((path[0][99] + 1) / 2 * 512), round((path[1][99] + 1) / 2 * 384)) or abs(x0 - (path[0][100] + 1) * 256) <= 2
    slider = hos[2].split(",")
    assert slider[5].startswith("B|") and int(slider[6]) == 2 and float(slider[7]) > 10.0
    assert hos[3].startswith("256,192,") and float(hos[3].split(",")[5]) > times[3]
    # beat detection path (bpm=None) on the same signal still yields a well-formed map
    auto = D.decode_beatmap(meta, enc, frame_times, bpm=None, allow_beat_snap=False, verbose=False)
    assert auto.split("[TimingPoints]\n")[1].startswith("0,300.0,4,0,0,50,1,0") and len(auto.split("[HitObjects]\n")[1].strip().splitlines()) == 4
