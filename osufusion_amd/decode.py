"""Sampled signal -> ``.osu`` text: host-side mirror of ``osu_fusion/library/osu/data/decode.py`` (``Metadata``, ``decode_beatmap``)
and of the two helpers it pulls in, ``data/hit.py:24-28,57-74`` (``decode_flips`` / ``decode_extents``) and
``data/fit_bezier.py`` (Schneider's curve fit as the reference adapts it from volkerp/fitCurves).

This is the CPU post-processing that ``inference_gradio.py:149-163`` runs on every sample AFTER the GPU sampler returned
(SURVEY section 8f row 4): a few thousand frames of 6 channels per song -- numpy on the host, as in the reference; nothing here is
on the timed path and nothing here needs the GPU.  The reference evaluates Bezier points and arc lengths through the ``bezier``
package (absent from this image, so the reference module cannot even be imported here: **parity unpinned**, the reference holds no
decoded fixtures either); the same quantities are computed below from their definitions -- de Casteljau / Bernstein evaluation,
and the arc length by composite 16-point Gauss-Legendre quadrature of |B'(t)| (agrees with adaptive quadrature to ~1e-12 on
the cubic segments that occur).

Signal layout (``data/encode.py:10-26``): rows HIT, SUSTAIN, SLIDER, COMBO, CURSOR_X, CURSOR_Y in [-1, 1]; HIT / COMBO flip sign at
every (new-combo) hit object, SUSTAIN / SLIDER are positive while a slider or spinner (resp. the first slide of a slider) lasts.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass
from typing import List, NamedTuple, Optional, Sequence, Tuple

import numpy as np

HIT, SUSTAIN, SLIDER, COMBO, CURSOR_X, CURSOR_Y = range(6)        # data/encode.py:10-23 (BeatmapEncoding)
TOTAL_DIM = 6

BEAT_DIVISOR = 16                                                  # decode.py:13-16
SLIDER_MULT = 1.0
MIN_BPM, MAX_BPM = 1, 300
PLAYFIELD = (512.0, 384.0)                                          # osu!pixels; cursor channels map [-1, 1] onto it (decode.py:152)


@dataclass
class Metadata:                                                     # decode.py:19-28
    audio_filename: str
    title: str
    artist: str
    version: str
    cs: float
    ar: float
    od: float
    hp: float


class TimingPoint(NamedTuple):                                      # the fields of beatmap.TimingPoint that decode.py uses
    t: float
    beat_length: float
    meter: int = 4


# ---- hit signals (data/hit.py) -----------------------------------------------------------------------------------------------
def _peaks_above(x: np.ndarray, height: float) -> List[int]:
    """Indices of strict local maxima (plateaus: their middle sample) of x that reach `height` -- what scipy.signal.find_peaks(x,
    height=height) returns; written out so that decoding needs numpy only."""
    out, i, n = [], 1, len(x)
    while i < n - 1:
        if x[i - 1] < x[i]:
            j = i
            while j < n - 1 and x[j + 1] == x[i]:
                j += 1
            if j < n - 1 and x[j + 1] < x[i]:
                if x[i] >= height:
                    out.append((i + j) // 2)
                i = j
        i += 1
    return out


def decode_flips(flips: np.ndarray) -> List[int]:
    """hit.py:24-28: frame indices at which a +-1 flip signal changes sign (peaks of |gradient| above 0.5), ascending."""
    grad = np.gradient(np.asarray(flips, dtype=np.float64))
    return sorted(_peaks_above(grad, 0.5) + _peaks_above(-grad, 0.5))


def decode_extents(extents: np.ndarray) -> Tuple[List[int], List[int]]:
    """hit.py:57-74: (starts, ends) of the positive runs of a +-1 signal -- index of the last non-positive sample before a run and of
    the last positive sample of it; ends that do not come after their start are dropped, unmatched starts are cut off."""
    x = np.asarray(extents)
    low_before, low_after = x[:-1] <= 0, x[1:] <= 0
    starts = np.flatnonzero(low_before & ~low_after).tolist()
    ends = np.flatnonzero(~low_before & low_after).tolist()
    cursor = 0
    for cursor, s in enumerate(starts):                              # ends at or before this start belong to no run that starts here
        while cursor < len(ends) and s >= ends[cursor]:
            ends.pop(cursor)
        if cursor >= len(ends):                                      # a run still open at the end of the signal: stop pairing
            break
    paired = cursor + 1
    return starts[:paired], ends[:paired]


# ---- Bezier pieces (data/fit_bezier.py; evaluation and length restated without the `bezier` package) ----------------------------
def bezier_points(ctrl: np.ndarray, t: np.ndarray) -> np.ndarray:
    """Points of the Bezier curve with control points ctrl (n+1, 2) at parameters t (m,) -> (m, 2), by de Casteljau."""
    t = np.asarray(t, dtype=np.float64)[:, None, None]
    pts = np.broadcast_to(np.asarray(ctrl, dtype=np.float64)[None], (t.shape[0],) + np.shape(ctrl)).copy()
    while pts.shape[1] > 1:
        pts = (1.0 - t) * pts[:, :-1] + t * pts[:, 1:]
    return pts[:, 0]


def _hodograph(ctrl: np.ndarray) -> np.ndarray:
    """Control points of the derivative curve.  (The reference's `hodo` multiplies by the NUMBER of control points, degree + 1,
    where the derivative of a degree-n curve has the factor n (fit_bezier.py:10-11); the Newton step below divides one such factor
    by the other, so the quirk changes its step length -- kept, it is part of how the reference re-parameterises.)"""
    ctrl = np.asarray(ctrl, dtype=np.float64)
    return ctrl.shape[0] * (ctrl[1:] - ctrl[:-1])


_GL_X, _GL_W = np.polynomial.legendre.leggauss(16)


def get_segment_length(ctrl: np.ndarray, pieces: int = 8) -> float:
    """Arc length of a Bezier segment: integral of |B'(t)| over [0, 1] (what bezier.Curve.length integrates adaptively)."""
    ctrl = np.asarray(ctrl, dtype=np.float64)
    if ctrl.shape[0] < 2:
        return 0.0
    deriv = (ctrl.shape[0] - 1) * (ctrl[1:] - ctrl[:-1])             # the true derivative's control points
    edges = np.linspace(0.0, 1.0, pieces + 1)
    total = 0.0
    for lo, hi in zip(edges[:-1], edges[1:]):
        t = 0.5 * (hi - lo) * _GL_X + 0.5 * (hi + lo)
        speed = np.linalg.norm(bezier_points(deriv, t), axis=1) if deriv.shape[0] > 1 else np.full(t.shape, np.linalg.norm(deriv[0]))
        total += 0.5 * (hi - lo) * float(np.dot(_GL_W, speed))
    return total


def _unit(v: np.ndarray) -> np.ndarray:
    n = float(np.sqrt(np.dot(v, v)))
    return v if n < np.finfo(float).eps else v / n


def _worst_point(ctrl: np.ndarray, pts: np.ndarray, u: np.ndarray) -> Tuple[float, int]:
    err = ((bezier_points(ctrl, u) - pts) ** 2).sum(axis=1)
    k = int(err.argmax())
    return float(err[k]), k


def _cubic_through(pts: np.ndarray, u: np.ndarray, tan_l: np.ndarray, tan_r: np.ndarray) -> np.ndarray:
    """Least-squares cubic with fixed end points and end tangent directions (fit_bezier.py:105-150)."""
    cub = np.array([pts[0], pts[0], pts[-1], pts[-1]], dtype=np.float64)
    basis = (3.0 * (1.0 - u) * u)[:, None] * np.stack([1.0 - u, u], axis=1)          # B1(u), B2(u)
    a = basis[:, :, None] * np.stack([tan_l, tan_r])[None]                           # (m, 2, xy)
    c = np.einsum("lix,ljx->ij", a, a)
    x = np.einsum("lix,lx->i", a, pts - bezier_points(cub, u))
    det = c[0, 0] * c[1, 1] - c[1, 0] * c[0, 1]
    alpha_l = 0.0 if abs(det) < 1e-5 else (x[0] * c[1, 1] - x[1] * c[0, 1]) / det
    alpha_r = 0.0 if abs(det) < 1e-5 else (c[0, 0] * x[1] - c[1, 0] * x[0]) / det
    chord = float(np.linalg.norm(pts[0] - pts[-1]))
    if alpha_l < 1e-6 * chord or alpha_r < 1e-6 * chord:                             # Wu / Barsky fallback
        alpha_l = alpha_r = chord / 3.0
    cub[1] += tan_l * alpha_l
    cub[2] += tan_r * alpha_r
    return cub


def _reparameterise(ctrl: np.ndarray, pts: np.ndarray, u: np.ndarray) -> np.ndarray:
    """One Newton step per point towards the parameter of its foot point on the curve (fit_bezier.py:153-173)."""
    d1 = _hodograph(ctrl)
    d2 = _hodograph(d1)
    gap = bezier_points(ctrl, u) - pts
    vel = bezier_points(d1, u)
    acc = bezier_points(d2, u)
    num = (gap * vel).sum(axis=1)
    den = (vel ** 2 + gap * acc).sum(axis=1)
    step = np.zeros_like(num)
    np.divide(num, den, out=step, where=den != 0)
    return u - step


def fit_bezier(points: np.ndarray, max_err: float, left_tangent: Optional[np.ndarray] = None,
               right_tangent: Optional[np.ndarray] = None) -> List[np.ndarray]:
    """fit_bezier.py:50-102: one or more Bezier segments (2 control points for a straight run, else 4) through `points` (m, 2) with
    squared error below max_err; end tangents from geometrically weighted chords of up to 5 neighbours; up to 32 rounds of
    re-parameterisation, then a split at the worst point."""
    points = np.asarray(points, dtype=np.float64)
    m = len(points)
    if m < 2:
        return []
    taps = min(5, m - 2)
    k = np.arange(1, taps + 1)
    weights = (2.0 ** -k) / (1.0 - 2.0 ** -float(taps)) if taps > 0 else np.zeros(0)
    if left_tangent is None:
        left_tangent = _unit((points[2:2 + taps] - points[1]).T @ weights) if taps > 0 else np.zeros(2)
    if right_tangent is None:
        back = points[-3:-3 - taps:-1] if taps > 0 else np.zeros((0, 2))
        right_tangent = _unit((back - points[-2]).T @ weights[:len(back)]) if taps > 0 else np.zeros(2)
    if m == 2:
        return [points]
    u = np.concatenate([[0.0], np.cumsum(np.linalg.norm(np.diff(points, axis=0), axis=1))])
    u = u / u[-1]
    split = m // 2
    for _ in range(32):
        cub = _cubic_through(points, u, left_tangent, right_tangent)
        err, split = _worst_point(cub, points, u)
        if err < max_err:
            ends = cub[[0, -1]]
            return [ends] if _worst_point(ends, points, u)[0] < max_err else [cub]
        u = _reparameterise(cub, points, u)
    centre = _unit(points[split - 1] - points[split + 1])
    return (fit_bezier(points[:split + 1], max_err, left_tangent, centre)
            + fit_bezier(points[split:], max_err, -centre, right_tangent))


def slider_decoder(cursor: np.ndarray, start_idx: int, end_idx: int, num_repeats: int) -> Tuple[float, List[np.ndarray]]:
    """decode.py:62-79: control points (rounded to osu!pixels) and length of the first slide of a slider."""
    first_slide_end = round(start_idx + (end_idx - start_idx) / num_repeats)
    control, length = [], 0.0
    for seg in fit_bezier(cursor.T[start_idx:first_slide_end + 1], max_err=50.0):
        seg = np.round(seg)
        control.extend(seg)
        length += get_segment_length(seg)
    return length, control


# ---- timing (decode.py:82-130) --------------------------------------------------------------------------------------------------
def get_timings(hit_times: np.ndarray, beat_length: float) -> Tuple[bool, TimingPoint]:
    """Offset of the beat grid = the fullest of 100 phase bins of hit_times modulo one beat."""
    hist, edges = np.histogram(np.asarray(hit_times) % beat_length, bins=100, range=(0, beat_length))
    return True, TimingPoint(float(edges[int(np.argmax(hist))]), beat_length, 4)


def calculate_timing_point(hit_times: np.ndarray, allow_beat_snap: bool, verbose: bool = True) -> Tuple[bool, TimingPoint]:
    """Tempo from the autocorrelation of the inter-onset intervals, refined by a 1,000-point scan of +-5 % for the tempo whose
    phase histogram is the most peaked (decode.py:89-121); (False, 200 BPM at 0 ms) when snapping is off or no lag qualifies."""
    fallback = (False, TimingPoint(0, 60000 / 200, 4))
    if not allow_beat_snap:
        return fallback
    gaps = np.diff(np.asarray(hit_times, dtype=np.float64))
    corr = np.correlate(gaps, gaps, mode="full")[len(gaps) - 1:] if len(gaps) else np.zeros(0)
    periods = 60000.0 / np.arange(MIN_BPM, MAX_BPM + 1)
    cand = np.asarray(_peaks_at_distance(corr, periods.min()), dtype=int)
    cand = cand[(cand >= periods.min() * 0.95) & (cand <= periods.max() * 1.05)]
    if len(cand) == 0:
        if verbose:
            print("Warning: no valid BPM found within the range, disabling beat snap")
        return fallback
    lag = cand[int(np.argmax(corr[cand]))]
    bpm0 = 60000.0 / lag
    scan = np.linspace(bpm0 * 0.95, bpm0 * 1.05, 1000)
    score = [np.histogram(np.asarray(hit_times) % (60000.0 / b), bins=100, range=(0, 60000.0 / b))[0].max() for b in scan]
    return get_timings(hit_times, 60000.0 / scan[int(np.argmax(score))])


def _peaks_at_distance(x: np.ndarray, distance: float) -> List[int]:
    """scipy.signal.find_peaks(x, distance=distance): local maxima, the higher ones suppressing neighbours closer than ceil(distance)."""
    peaks = np.asarray(_peaks_above(x, -np.inf), dtype=int)
    if len(peaks) == 0:
        return []
    dist = int(np.ceil(distance))
    keep = np.ones(len(peaks), dtype=bool)
    for j in np.argsort(x[peaks])[::-1]:                             # highest first
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < dist:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < len(peaks) and peaks[k] - peaks[j] < dist:
            keep[k] = False
            k += 1
    return peaks[keep].tolist()


def snap_to_beat(t: float, u: float, beat_offset: float, beat_length: float) -> Tuple[float, float]:
    """decode.py:124-128: both times to the nearest 1/16 beat of the grid."""
    tick = beat_length / BEAT_DIVISOR
    return (round((t - beat_offset) / tick) * tick + beat_offset, round((u - beat_offset) / tick) * tick + beat_offset)


# ---- .osu text (decode.py:31-59,131-237) ----------------------------------------------------------------------------------------
def _osu_text(meta: Metadata, timing_points: Sequence[str], hit_objects: Sequence[str]) -> str:
    m = asdict(meta)
    sections = [
        ("General", [f"AudioFilename: {m['audio_filename']}", "AudioLeadIn: 0", "Mode: 0"]),
        ("Metadata", [f"Title: {m['title']}", f"TitleUnicode: {m['title']}", f"Artist: {m['artist']}", f"ArtistUnicode: {m['artist']}",
                      "Creator: OsuFusion", f"Version: {m['version']}", "Tags: OsuFusion"]),
        ("Difficulty", [f"HPDrainRate: {m['hp']}", f"CircleSize: {m['cs']}", f"OverallDifficulty: {m['od']}", f"ApproachRate: {m['ar']}",
                        "SliderMultiplier: 1", "SliderTickRate: 1"]),
        ("TimingPoints", list(timing_points)),
        ("HitObjects", list(hit_objects)),
    ]
    out = ["osu file format v14", ""]
    for name, lines in sections:
        out.append(f"[{name}]")
        out.append("\n".join(lines))
        out.append("")
    return "\n".join(out)


def decode_beatmap(metadata: Metadata, encoded_beatmap: np.ndarray, frame_times: np.ndarray, bpm: Optional[float],
                   allow_beat_snap: bool = True, verbose: bool = True) -> str:
    """decode.py:133-237.  encoded_beatmap: (6, L) sample of the diffusion model (any real values: the four hit channels are
    discretised by sign), frame_times: (L,) milliseconds.  Returns the text of an ``.osu`` file (format v14)."""
    enc = np.asarray(encoded_beatmap, dtype=np.float64)
    frame_times = np.asarray(frame_times)
    hit_sig = np.where(enc[[HIT, SUSTAIN, SLIDER, COMBO]] > 0, 1.0, -1.0)
    cursor = (enc[[CURSOR_X, CURSOR_Y]] + 1.0) / 2.0 * np.array([[PLAYFIELD[0]], [PLAYFIELD[1]]])

    onsets = decode_flips(hit_sig[0])
    onset_of_frame = np.full(len(frame_times), -1, dtype=int)
    for i, frame in enumerate(onsets):
        onset_of_frame[frame] = i
    new_combo = [False] * len(onsets)
    for frame in decode_flips(hit_sig[3]):
        new_combo[onset_of_frame[frame]] = True                     # (a combo flip away from any onset marks the LAST object, as index -1 does)

    def run_ends(channel: int) -> List[int]:
        ends = [-1] * len(onsets)
        for start, end in zip(*decode_extents(hit_sig[channel])):
            if onset_of_frame[start] != -1:
                ends[onset_of_frame[start]] = end
        return ends

    sustain_end, slider_end = run_ends(1), run_ends(2)
    hit_times = frame_times[onsets]
    snap, tp = get_timings(hit_times, 60000 / bpm) if bpm is not None else calculate_timing_point(hit_times, allow_beat_snap, verbose)
    slider_vel_base = SLIDER_MULT * 100 / tp.beat_length
    timing_lines = [f"{tp.t},{tp.beat_length},{tp.meter},0,0,50,1,0"]
    objects: List[str] = []
    for frame, combo, s_end, l_end in zip(onsets, new_combo, sustain_end, slider_end):
        x, y = np.round(cursor[:, frame]).astype(int)
        t, u = frame_times[frame], frame_times[s_end]
        combo_bit = 4 if combo else 0
        if snap:
            t, u = snap_to_beat(t, u, tp.t, tp.beat_length)
        circle = f"{x},{y},{t},{1 + combo_bit},0,0:0:0:0:"
        if s_end == -1 or s_end - frame < 4:                        # no sustain, or one too short to be anything but a circle
            objects.append(circle)
            continue
        if l_end == -1:                                             # sustained without a slide: spinner
            objects.append(f"256,192,{t},{8 + combo_bit},0,{u}")
            continue
        if l_end - frame < 4:
            objects.append(circle)
            continue
        slides = max(1, round((s_end - frame) / (l_end - frame)))
        length, ctrl = slider_decoder(cursor, frame, s_end, slides)
        if length == 0:
            objects.append(circle)                                  # (the reference then still emits the slider line below)
        x1, y1 = ctrl[0]
        curve = "|".join(f"{px}:{py}" for px, py in ctrl[1:])
        objects.append(f"{x1},{y1},{t},{2 + combo_bit},0,B|{curve},{slides},{length}")
        ratio = (length * slides / (u - t)) / slider_vel_base
        ratio = 1 if ratio == 0 else ratio
        if (ratio > 10 or ratio < 0.1) and verbose:
            print(f"Warning: slider velocity {ratio} is out of bounds, slider will not be good")
        timing_lines.append(f"{t},{-100 / ratio},4,0,0,50,0,0")
    return _osu_text(metadata, timing_lines, objects)
