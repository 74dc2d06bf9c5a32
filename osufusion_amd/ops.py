"""Raw (non-autograd) wrappers: torch tensors in, one C-ABI kernel launch each.  No fallbacks.

"rows" tensors are channels-last activations: contiguous (B, L, C) or (M, C); only the last dim must be
contiguous, the row stride may exceed the width (column slices of wider buffers are fine).
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import numpy as np
import torch

from . import _lib

F32, BF16 = 0, 1
_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _rows(t: torch.Tensor) -> Tuple[int, int, int]:
    """(M, cols, row stride) of a rows tensor."""
    assert t.is_cuda, "osufusion_amd kernels run on the GPU only (no CPU fallback)"
    assert t.stride(-1) == 1, "last dim must be contiguous"
    cols = t.shape[-1]
    if t.dim() == 2:
        return t.shape[0], cols, t.stride(0)
    assert t.dim() == 3
    B, Lr = t.shape[0], t.shape[1]
    ld = t.stride(1)
    assert t.stride(0) == Lr * ld or B == 1, "batch dim must be densely stacked rows"
    return B * Lr, cols, ld


def dt_of(t: torch.Tensor) -> int:
    return _DT[t.dtype]


F32X3 = 2                  # OSUF_DT_F32X3: fp32 storage, GEMM products as three bf16 MFMAs on split operands
F32_MATMUL = "exact"       # how the fp32 compute mode multiplies: "exact" (v_mfma_f32_32x32x2_f32, a bitwise fmaf chain, 1/16 of the bf16
#                            MFMA rate) or "x3" (a = a_hi + a_lo in bf16, three bf16 MFMAs: inputs kept to ~17 bits, 3/16 of the cost)


def set_f32_matmul(mode: str) -> str:
    """-> the previous setting.  Only the conv / linear GEMMs (osuf_gemm_nt / osuf_gemm_tn) have the x3 form; the embedding-sized
    linears (csrc/skinny.hip), attention (bf16 as the reference casts it) and every norm / elementwise kernel are untouched."""
    global F32_MATMUL
    assert mode in ("exact", "x3")
    prev, F32_MATMUL = F32_MATMUL, mode
    return prev


def gemm_dt(t: torch.Tensor) -> int:
    return F32X3 if (t.dtype == torch.float32 and F32_MATMUL == "x3") else _DT[t.dtype]


class LaunchSize(int):
    """what a timed launch records: the sequence length N (as an int), with the launch's batch as .b"""
    def __new__(cls, n, b=0):
        x = int.__new__(cls, n)
        x.b = b
        return x


class KernelTimer:
    """HIP-event timing of selected C-ABI launches on the stream they run on (bench.py's roofline leg)."""

    def __init__(self, names) -> None:
        self.names = set(names)
        self.records = {n: [] for n in names}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for n, recs in self.records.items():
            if recs:
                out[n] = dict(launches=len(recs), total_ms=sum(s.elapsed_time(e) for s, e, _ in recs), sizes=[m for _, _, m in recs])
        return out


_TIMER: Optional[KernelTimer] = None

# Reproducible mode (the sampling loop): GroupNorm statistics and GlobalContext pooling by fixed-order two-stage reductions
# (osuf_gn_stats, osuf_wcolsum with a partial buffer) instead of the atomics fused into the training step's kernels.
_REPRODUCIBLE = False


def reproducible() -> bool:
    return _REPRODUCIBLE


class reproducible_mode:
    def __init__(self, flag: bool = True) -> None:
        self.flag = bool(flag)

    def __enter__(self):
        global _REPRODUCIBLE
        self.prev, _REPRODUCIBLE = _REPRODUCIBLE, self.flag
        return self

    def __exit__(self, *exc):
        global _REPRODUCIBLE
        _REPRODUCIBLE = self.prev
        return False


def set_kernel_timer(t: Optional[KernelTimer]) -> None:
    global _TIMER
    _TIMER = t


class ZeroArena:
    """Pre-zeroed scratch for one train step.  The kernels' accumulators (GroupNorm statistics and backward sums, pooling / bias
    column sums, the loss) each used to be a `torch.zeros` = an allocation plus a fill launch -- ~380 fills per step.  Inside a
    Trainer step they are bump-allocated from one device buffer that is cleared by ONE fill at the start of the next step (up to the
    previous high-water mark); nothing handed out is reused before then, so tensors saved for backward stay valid."""

    ALIGN = 256

    def __init__(self, device, nbytes: int = 192 << 20) -> None:
        self.buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
        self.off = 0
        self.dirty = 0                                     # bytes handed out since the last clear
        self.misses = 0

    def begin(self) -> None:
        if self.dirty:
            self.buf[: self.dirty].zero_()
        self.off = self.dirty = 0

    def take(self, shape, dtype: torch.dtype) -> Optional[torch.Tensor]:
        n = 1
        for d in (shape if isinstance(shape, (tuple, list)) else (shape,)):
            n *= int(d)
        nbytes = n * torch.empty(0, dtype=dtype).element_size()
        end = self.off + (nbytes + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        if end > self.buf.numel():
            self.misses += 1
            return None
        t = self.buf[self.off: self.off + nbytes].view(dtype).view(shape)
        self.off = self.dirty = end
        return t


_ARENA: Optional[ZeroArena] = None


def set_zero_arena(arena: Optional[ZeroArena]) -> None:
    """Trainer.step activates its arena for the duration of forward + backward; None = plain torch.zeros."""
    global _ARENA
    _ARENA = arena


def zeros(shape, dtype: torch.dtype, device) -> torch.Tensor:
    """A zero-initialised accumulator: from the active step arena when there is one (and it is on `device`), else torch.zeros."""
    if _ARENA is not None and _ARENA.buf.device == torch.device(device):
        t = _ARENA.take(shape, dtype)
        if t is not None:
            return t
    return torch.zeros(shape, dtype=dtype, device=device)


_TIMED_AS = {"osuf_mqa_fwd_qs": "osuf_mqa_fwd", "osuf_mqa_fwd_zdq": "osuf_mqa_fwd", "osuf_mqa_fwd_rope": "osuf_mqa_fwd", "osuf_mqa_bwd_fused_qs": "osuf_mqa_bwd_fused"}    # pre-scaled-query forms: timed under the plain name


def call(name: str, *args, meta=None) -> None:
    lib = _lib.load()
    tname = _TIMED_AS.get(name, name)
    if _TIMER is not None and tname in _TIMER.names:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        _lib.check(getattr(lib, name)(*args), name)
        e.record()
        _TIMER.records[tname].append((s, e, meta))
        return
    _lib.check(getattr(lib, name)(*args), name)


# ---------------------------------------------------------------------------------------------------------
# GEMMs
# ---------------------------------------------------------------------------------------------------------
def gemm_nt(a: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, *, taps: int = 1, n_out: Optional[int] = None,
            lin: Optional[int] = None, lout: Optional[int] = None, stride: int = 1, pad: int = 0, mode: int = 0, act: int = 0,
            residual: Optional[torch.Tensor] = None, rscale: Optional[torch.Tensor] = None, dact: Optional[torch.Tensor] = None,
            stats: Optional[torch.Tensor] = None, want_pre: bool = False, out: Optional[torch.Tensor] = None,
            out_shape: Optional[Tuple[int, ...]] = None):
    """C = epilogue(sum_t A[rowmap(m,t)] @ W[t].T).  a: rows [Min][K]; w: [taps][N][K] (same dtype).  See gemm.hip."""
    Min, K, lda = _rows(a)
    assert w.dtype == a.dtype and w.is_contiguous()
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    assert w3.shape[0] == taps and w3.shape[2] == K, (tuple(w3.shape), taps, K)
    N = w3.shape[1] if n_out is None else n_out
    if lin is None:
        lin = lout = Min                      # plain GEMM: one "sample" of Min rows
    nb = Min // lin
    M = nb * lout
    if out is None:
        shape = out_shape if out_shape is not None else (M, N)
        out = torch.empty(shape, dtype=a.dtype, device=a.device)
    Mo, No, ldc = _rows(out)
    assert Mo == M and No >= N
    pre = torch.empty_like(out) if want_pre else None
    ldr = _rows(residual)[2] if residual is not None else 0
    ldu = _rows(dact)[2] if dact is not None else 0
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() >= N
    call("osuf_gemm_nt", gemm_dt(a), _p(a), lda, _p(w3), K, N * K if taps > 1 else 0, _p(out), ldc, _p(pre), ldc, _p(residual), ldr,
         _p(dact), ldu, _p(bias), _p(rscale), _p(stats), M, N, K, taps, lin, lout, stride, pad, mode, act, _stream())
    return (out, pre) if want_pre else out


def gemm_nt_rowdot(a: torch.Tensor, w: torch.Tensor, o: torch.Tensor, L: int, heads: int):
    """(C, delta): C = a @ w[0].T (rows [M][heads*64], a's dtype) and delta[b][h][l] = sum_d bf16(C) * o over each head's 64 columns."""
    M, K, lda = _rows(a)
    w3 = w if w.dim() == 3 else w.unsqueeze(0)
    assert w3.dtype == a.dtype and w3.is_contiguous() and w3.shape[0] == 1 and w3.shape[2] == K and w3.shape[1] == heads * 64
    Mo, No, ldo = _rows(o)
    assert Mo == M and No == heads * 64 and o.dtype == a.dtype and M % L == 0
    out = torch.empty(o.shape, dtype=a.dtype, device=a.device)
    delta = torch.empty((M // L, heads, L), dtype=torch.float32, device=a.device)
    call("osuf_gemm_nt_rowdot", gemm_dt(a), _p(a), lda, _p(w3), K, _p(out), heads * 64, _p(o), ldo, _p(delta), M, heads * 64, K, L, heads, _stream())
    return out, delta


def gemm_tn(dy: torch.Tensor, x: torch.Tensor, *, taps: int = 1, lin: Optional[int] = None, lout: Optional[int] = None, stride: int = 1,
            pad: int = 0, mode: int = 0, n1: Optional[int] = None, out: Optional[torch.Tensor] = None, conv_layout: bool = False,
            accumulate: bool = False, bias_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """sum_m dY[m][n1] * X[rowmap(m,t)][n2] -> fp32.  Default result [taps][N1][N2]; conv_layout=True -> (N1, N2, taps), the layout
    of a torch Conv1d weight.  `out` (dense fp32) is overwritten, or added into when accumulate=True (e.g. a parameter's .grad).
    bias_out (fp32 (N1,)): += the column sums of dy, i.e. the layer's bias gradient, from the same pass (osuf_gemm_tn_bias)."""
    M, N1, ldy = _rows(dy)
    Mx, N2, ldx = _rows(x)
    if n1 is not None:
        N1 = n1
    if lin is None:
        lin = lout = M
    assert dy.dtype == x.dtype and Mx // lin == M // lout
    shape = (N1, N2, taps) if conv_layout else (taps, N1, N2)
    if out is None:
        assert not accumulate
        out = torch.empty(shape, dtype=torch.float32, device=dy.device)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() == taps * N1 * N2
    need = _lib.load().osuf_gemm_tn_workspace_bytes(gemm_dt(dy), M, N1, N2, taps)   # the dtype code the launch uses (x3 has split plans, exact f32 none)
    ws = _workspace(need, dy.device) if need > 0 else None
    args = (gemm_dt(dy), _p(dy), ldy, _p(x), ldx, _p(out), N2, N1 * N2, M, N1, N2, taps, lin, lout, stride, pad, mode, 0,
            1 if conv_layout else 0, 1 if accumulate else 0, _p(ws), need if ws is not None else 0)
    if bias_out is not None:
        assert bias_out.dtype == torch.float32 and bias_out.is_contiguous() and bias_out.numel() == N1
        call("osuf_gemm_tn_bias", *args, _p(bias_out), _stream())
    else:
        call("osuf_gemm_tn", *args, _stream())
    return out


_WS = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Persistent fp32 scratch for split-wgrad partial tiles (grown geometrically, one per device; kernels on one stream
    are ordered, so consecutive wgrads can share it)."""
    key = str(device)
    cur = _WS.get(key)
    if cur is None or cur.numel() * 4 < nbytes:
        cur = torch.empty(max(nbytes // 4, 1 << 24) * 5 // 4, dtype=torch.float32, device=device)
        _WS[key] = cur
    return cur


def colsum(y: torch.Tensor, n: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[n] += sum_m y[m][n]  (`out` given: accumulated into, e.g. a bias .grad; else a fresh zeroed tensor)."""
    M, N, ld = _rows(y)
    N = N if n is None else n
    if out is None:
        out = zeros(N, torch.float32, y.device)
    call("osuf_colsum", dt_of(y), _p(y), ld, M, N, _p(out), _stream())
    return out


# ---------------------------------------------------------------------------------------------------------
# norms / gating
# ---------------------------------------------------------------------------------------------------------
def gn_finalize(stats: torch.Tensor, count: int) -> torch.Tensor:
    B = stats.shape[0]
    mr = torch.empty((B, 2), dtype=torch.float32, device=stats.device)
    call("osuf_gn_finalize", _p(stats), _p(mr), B, count, _stream())
    return mr


def gn_stats(y: torch.Tensor, L: int) -> torch.Tensor:
    """(mean, rstd) per sample of GroupNorm(1, C) over rows y, by fixed-order reductions (bit-reproducible)."""
    M, C, ld = _rows(y)
    B = M // L
    need = _lib.load().osuf_gn_stats_workspace_bytes(M, C, L)
    part = torch.empty(max(need // 8, 1), dtype=torch.float64, device=y.device)
    mr = torch.empty((B, 2), dtype=torch.float32, device=y.device)
    call("osuf_gn_stats", dt_of(y), _p(y), ld, _p(part), _p(mr), M, C, L, _stream())
    return mr


def gn_apply_reproducible(y: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ss: Optional[torch.Tensor], L: int):
    """Bit-reproducible statistics + apply in two launches (the sampler): per-chunk partial sums, then the apply kernel adds them in a fixed
    order itself -> (h, mean_rstd (B, 2)).  Same result on every call; not the same bits as gn_stats' serial second stage."""
    M, C, ld = _rows(y)
    B = M // L
    need = _lib.load().osuf_gn_stats_workspace_bytes(M, C, L)
    part = torch.empty(max(need // 8, 1), dtype=torch.float64, device=y.device)
    h = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    mr = torch.empty((B, 2), dtype=torch.float32, device=y.device)
    call("osuf_gn_stats_parts", dt_of(y), _p(y), ld, _p(part), M, C, L, _stream())
    call("osuf_gn_apply_fwd_parts", dt_of(y), _p(y), ld, _p(h), C, _p(part), _p(mr), _p(gamma), _p(beta), _p(ss), M, C, L, _stream())
    return h, mr


def gn_apply_from_stats(y: torch.Tensor, stats: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ss: Optional[torch.Tensor], L: int):
    """gn_finalize + gn_apply in one launch: stats (B, 2) fp64 raw sums from the GEMM epilogue -> (h, mean_rstd (B, 2))."""
    M, C, ld = _rows(y)
    h = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    mr = torch.empty((M // L, 2), dtype=torch.float32, device=y.device)
    call("osuf_gn_apply_fwd_stats", dt_of(y), _p(y), ld, _p(h), C, _p(stats), L * C, _p(mr), _p(gamma), _p(beta), _p(ss), M, C, L, _stream())
    return h, mr


def gn_apply(y: torch.Tensor, mr: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ss: Optional[torch.Tensor], L: int) -> torch.Tensor:
    M, C, ld = _rows(y)
    h = torch.empty(y.shape, dtype=y.dtype, device=y.device)
    call("osuf_gn_apply_fwd", dt_of(y), _p(y), ld, _p(h), C, _p(mr), _p(gamma), _p(beta), _p(ss), M, C, L, _stream())
    return h


def gn_bwd(dh: torch.Tensor, y: torch.Tensor, mr: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, ss: Optional[torch.Tensor], L: int,
           dgamma_out: Optional[torch.Tensor] = None, dbeta_out: Optional[torch.Tensor] = None, dbias_out: Optional[torch.Tensor] = None,
           dyy_out: Optional[torch.Tensor] = None, identity_norm: bool = False):
    """dgamma_out / dbeta_out (fp32 (C,)) are accumulated into when given (parameter .grad buffers); dbias_out (fp32 (C,)) receives
    (+=) the gradient of the bias of the conv that produced y, in closed form from the per-(b, c) sums (no pass over dy); dyy_out
    (fp32 (C,), needs dbias_out) likewise the column sums of dy * y (DoRA magnitude gradient)."""
    M, C, ldy = _rows(y)
    B = M // L
    dev = y.device
    dy = torch.empty(y.shape, dtype=y.dtype, device=dev)
    T12 = zeros((B, 4, C), torch.float32, dev)
    S = torch.empty((B, 2), dtype=torch.float32, device=dev)
    dss = torch.empty((B, 2 * C), dtype=torch.float32, device=dev) if ss is not None else None
    if identity_norm:                                      # Block(norm=False): no statistics, no gamma / beta
        dgamma_out = dbeta_out = None
    elif dgamma_out is None or dbeta_out is None:
        dgb = zeros((2, C), torch.float32, dev)
        dgamma_out, dbeta_out = dgb[0], dgb[1]
    call("osuf_gn_bwd", dt_of(y), _p(dh), _rows(dh)[2], _p(y), ldy, _p(dy), C, _p(mr), _p(gamma), _p(beta), _p(ss), _p(T12), _p(S),
         _p(dss), _p(dgamma_out), _p(dbeta_out), _p(dbias_out), _p(dyy_out), M, C, L, _stream())
    return dy, dgamma_out, dbeta_out, dss


def ln_fwd(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    M, C, ld = _rows(x)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    mr = torch.empty((M, 2), dtype=torch.float32, device=x.device)
    call("osuf_ln_fwd", dt_of(x), _p(x), ld, _p(out), C, _p(mr), _p(gamma), _p(beta), M, C, _stream())
    return out, mr


def ln_bwd(dy: torch.Tensor, x: torch.Tensor, mr: torch.Tensor, gamma: torch.Tensor, dgamma_out: Optional[torch.Tensor] = None,
           dbeta_out: Optional[torch.Tensor] = None):
    M, C, ld = _rows(x)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    if dgamma_out is None or dbeta_out is None:
        dgb = torch.zeros((2, C), dtype=torch.float32, device=x.device)
        dgamma_out, dbeta_out = dgb[0], dgb[1]
    call("osuf_ln_bwd", dt_of(x), _p(dy), _rows(dy)[2], _p(x), ld, _p(dx), C, _p(mr), _p(gamma), _p(dgamma_out), _p(dbeta_out), M, C, _stream())
    return dx, dgamma_out, dbeta_out


def rowdot(h: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], L: int, per_sample: bool = False) -> torch.Tensor:
    M, C, ld = _rows(h)
    out = torch.empty(M, dtype=torch.float32, device=h.device)
    call("osuf_rowdot", dt_of(h), _p(h), ld, _p(w), C if per_sample else 0, _p(bias), _p(out), M, C, L, _stream())
    return out


def gca_pool_fused_ok() -> bool:
    return os.environ.get("OSUF_GCA_NO_FUSED_POOL") != "1"


def gca_pool(h: torch.Tensor, wk: torch.Tensor, bk: Optional[torch.Tensor], L: int):
    """GlobalContext pooling in one pass over h: returns (pooled (B, C) fp32, p (B*L,) fp32 = softmax over each sample's L logits h . wk + bk).
    A running softmax per workgroup, the workgroups' partials added in order by a second kernel: no atomics (bit-reproducible)."""
    M, C, ld = _rows(h)
    B = M // L
    need = _lib.load().osuf_gca_pool_workspace_bytes(M, C, L)
    part = torch.empty(max(need // 4, 1), dtype=torch.float32, device=h.device)
    p = torch.empty(M, dtype=torch.float32, device=h.device)
    pooled = torch.empty((B, C), dtype=torch.float32, device=h.device)
    call("osuf_gca_pool", dt_of(h), _p(h), ld, _p(wk), _p(bk), _p(part), _p(p), _p(pooled), M, C, L, _stream())
    return pooled, p


def softmax_rows_(p: torch.Tensor, B: int, L: int) -> torch.Tensor:
    call("osuf_softmax_rows", _p(p), B, L, _stream())
    return p


def wcolsum(a: torch.Tensor, bmul: Optional[torch.Tensor], w: Optional[torch.Tensor], B: int, L: int) -> torch.Tensor:
    M, C, ld = _rows(a)
    part = None
    if _REPRODUCIBLE:
        out = torch.empty((B, C), dtype=torch.float32, device=a.device)
        part = torch.empty((B, (L + 63) // 64, C), dtype=torch.float32, device=a.device)
    else:
        out = zeros((B, C), torch.float32, a.device)
    call("osuf_wcolsum", dt_of(a), _p(a), ld, _p(bmul), _rows(bmul)[2] if bmul is not None else 0, _p(w), _p(out), B, C, L, _p(part), _stream())
    return out


def gate_residual(h: torch.Tensor, gate: torch.Tensor, res: Optional[torch.Tensor], L: int) -> torch.Tensor:
    """out = h * gate[b] (+ res)."""
    M, C, ld = _rows(h)
    out = torch.empty(h.shape, dtype=h.dtype, device=h.device)
    call("osuf_gate_residual", dt_of(h), _p(h), ld, _p(gate), _p(res), _rows(res)[2] if res is not None else 0, _p(out), C, M, C, L, _stream())
    return out


def gca_bwd_apply(dout, h, p, gate, dpooled, sdot, wk, L: int, dwk_out: Optional[torch.Tensor] = None,
                  dbk_out: Optional[torch.Tensor] = None):
    """dwk_out (C,) / dbk_out (1,) fp32, optional: accumulated into (+= sum_m dlogit[m] h[m][:], += sum_m dlogit[m])."""
    M, C, ld = _rows(h)
    dh = torch.empty(h.shape, dtype=h.dtype, device=h.device)
    dlogit = torch.empty(M, dtype=torch.float32, device=h.device)
    ws, need = None, 0
    if dwk_out is not None:
        need = _lib.load().osuf_gca_bwd_apply_workspace_bytes(M, C)
        ws = torch.empty(need // 4, dtype=torch.float32, device=h.device)
    call("osuf_gca_bwd_apply", dt_of(h), _p(dout), _rows(dout)[2], _p(h), ld, _p(dh), C, _p(p), _p(gate), _p(dpooled), _p(sdot), _p(wk),
         _p(dlogit), M, C, L, _p(dwk_out), _p(dbk_out), _p(ws), need, _stream())
    return dh, dlogit


# ---------------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------------
LOG2E = 1.4426950408889634


def q_prescale_ok(head_dim: int, variant: int) -> bool:
    """May AttentionFn fold the softmax scale into the queries' bf16 rounding (osuf_rope_cast_qs -> osuf_mqa_fwd_qs -> osuf_mqa_bwd_fused_qs)?
    Only the 64-wide kernels and the fused backward sweeps have the pre-scaled form; OSUF_ATTN_NO_QS=1 switches it off (A/B runs)."""
    import os
    return head_dim == 64 and variant in _FUSED_DQ_MODE and not os.environ.get("OSUF_ATTN_NO_QS")


def rope_cast(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, N: int, n_rot: int, n_heads: int, head_dim: int,
              q_mul: float = 1.0, n_q_heads: int = 0) -> torch.Tensor:
    """RoPE + bf16 cast of q | k | v rows.  q_mul != 1: the first n_q_heads heads (the queries) are multiplied by q_mul = scale * log2 e
    before their one bf16 rounding -- the attention kernels' *_qs forms then take the scores straight as exponents."""
    M, W, ld = _rows(qkv)
    out = torch.empty(qkv.shape, dtype=torch.bfloat16, device=qkv.device)
    if q_mul != 1.0:
        call("osuf_rope_cast_qs", dt_of(qkv), _p(qkv), ld, _p(out), W, _p(cos), _p(sin), M, N, n_rot, n_heads, head_dim, q_mul, n_q_heads, _stream())
    else:
        call("osuf_rope_cast", dt_of(qkv), _p(qkv), ld, _p(out), W, _p(cos), _p(sin), M, N, n_rot, n_heads, head_dim, _stream())
    return out


def rope_bwd(dqkv32: torch.Tensor, out_dtype: torch.dtype, cos, sin, N: int, n_rot: int, n_heads: int, head_dim: int) -> torch.Tensor:
    M, W, ld = _rows(dqkv32)
    out = torch.empty(dqkv32.shape, dtype=out_dtype, device=dqkv32.device)
    call("osuf_rope_bwd", _DT[out_dtype], _p(dqkv32), ld, _p(out), W, _p(cos), _p(sin), M, N, n_rot, n_heads, head_dim, _stream())
    return out


DQ_PREZEROED = 0x100                                             # OSUF_DQ_PREZEROED


def fused_bwd_workspace(B: int, N: int, H: int, D: int, out_dtype: torch.dtype, device, variant: Optional[int] = None, qsplit: int = 0):
    """A PRIVATE workspace for one layer's fused attention backward, to be zero-filled by that layer's forward (mqa_fwd(zero_dq=...)) and handed to
    mqa_bwd(workspace=...): the dQ accumulator's memset (66 us per N = 4096 layer in front of the backward sweep) then rides the forward
    kernel, which is bound by the vector pipe and leaves HBM idle.  None where the path does not apply (head dim != 64, non-atomic dQ, OSUF_ATTN_NO_ZDQ=1)."""
    variant = ATTN_BWD_DEFAULT if variant is None else variant
    if D != 64 or variant not in _FUSED_DQ_MODE or variant in (ATTN_FUSED_SLABS, ATTN_FUSED512_TIMING) or os.environ.get("OSUF_ATTN_NO_ZDQ") == "1":
        return None
    need = _lib.load().osuf_mqa_bwd_fused_workspace_bytes(B, H, N, _DT[out_dtype], qsplit, _FUSED_DQ_MODE[variant])
    if need <= 0:
        return None
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    ws.fused_variant = variant                              # the layout (and the zero-filled part) belong to this variant's dq_mode
    return ws


def mqa_fwd(qkv: torch.Tensor, B: int, N: int, H: int, D: int, out_dtype: torch.dtype, scale: float, kv_heads: int = 1, qs: bool = False,
            zero_dq: Optional[torch.Tensor] = None):
    """qkv: bf16 rows [B*N][(H+2G)*D] (q heads | G k heads | G v heads).  Returns o rows [B*N][H*D] and lse2 [B][H][N].
    G = kv_heads > 1 (grouped-query attention, unet.py:135): the q heads are laid out GROUP-MAJOR -- heads g*H/G .. (g+1)*H/G - 1
    share K/V head g -- and every group is one launch of the one-K/V-head kernels on its column block; lse2 is then [G][B][H/G][N].
    qs: the q columns hold queries pre-scaled by scale * log2 e (rope_cast(q_mul=...)).  zero_dq (fused_bwd_workspace): its first B*N*H*D floats
    -- the backward's dQ accumulator -- are zero-filled by this launch (one K/V head, head dim 64)."""
    M, W, ld = _rows(qkv)
    G = kv_heads
    assert qkv.dtype == torch.bfloat16 and W == (H + 2 * G) * D and H % G == 0
    assert zero_dq is None or (G == 1 and D == 64 and zero_dq.dtype == torch.float32 and zero_dq.numel() >= B * N * H * D)
    r = H // G
    o = torch.empty((B, N, H * D), dtype=out_dtype, device=qkv.device)
    lse = torch.empty((B, H, N) if G == 1 else (G, B, r, N), dtype=torch.float32, device=qkv.device)
    base, eo = qkv.data_ptr(), o.element_size()
    if zero_dq is not None:
        call("osuf_mqa_fwd_zdq", base, ld, base + 2 * H * D, ld, base + 2 * (H + 1) * D, ld, o.data_ptr(), H * D, _DT[out_dtype], lse.data_ptr(), B, H, N, D, scale,
             1 if qs else 0, zero_dq.data_ptr(), _stream(), meta=LaunchSize(N, B))
        return o, lse
    for g in range(G):
        call("osuf_mqa_fwd_qs" if qs else "osuf_mqa_fwd", base + 2 * g * r * D, ld, base + 2 * (H + g) * D, ld, base + 2 * (H + G + g) * D, ld, o.data_ptr() + eo * g * r * D,
             H * D, _DT[out_dtype], lse.data_ptr() + 4 * g * B * r * N, B, r, N, D, scale, _stream(), meta=LaunchSize(N, B))
    return o, lse


def fwd_rope_ok(qkv: torch.Tensor, head_dim: int, kv_heads: int, qs: bool) -> bool:
    """The forward that rotates its own queries (mqa_fwd_rope) applies: bf16 projections, 64-wide heads, one K/V head, pre-scaled-query kernels."""
    return qs and head_dim == 64 and kv_heads == 1 and qkv.dtype == torch.bfloat16 and os.environ.get("OSUF_ATTN_NO_FWD_ROPE") != "1"


def mqa_fwd_rope(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, B: int, N: int, H: int, D: int, out_dtype: torch.dtype, scale: float,
                 write_q: bool = True, zero_dq: Optional[torch.Tensor] = None):
    """RoPE + attention forward from the raw bf16 q|kv projections [B*N][(H+2)*D]: K is rotated and V copied by a rope_cast over THEIR two head blocks
    (128 of the (H + 2) * 64 columns); the queries are rotated, multiplied by scale * log2 e and rounded once inside the attention kernel, which
    stores them for the backward when write_q (else the q columns of the returned rows stay unwritten: inference).  Returns (qkv_r, o, lse2)."""
    M, W, ld = _rows(qkv)
    assert qkv.dtype == torch.bfloat16 and D == 64 and W == (H + 2) * D
    qkv_r = torch.empty(qkv.shape, dtype=torch.bfloat16, device=qkv.device)
    kv_in, kv_out = qkv.data_ptr() + 2 * H * D, qkv_r.data_ptr() + 2 * H * D
    call("osuf_rope_cast", _DT[torch.bfloat16], kv_in, ld, kv_out, W, _p(cos), _p(sin), M, N, 1, 2, D, _stream())     # k: rotate; v: copy
    o = torch.empty((B, N, H * D), dtype=out_dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    assert zero_dq is None or (zero_dq.dtype == torch.float32 and zero_dq.numel() >= B * N * H * D)
    call("osuf_mqa_fwd_rope", qkv.data_ptr(), ld, kv_out, W, kv_out + 2 * D, W, o.data_ptr(), H * D, _DT[out_dtype], lse.data_ptr(), B, H, N, D, scale,
         _p(cos), _p(sin), scale * LOG2E, qkv_r.data_ptr() if write_q else None, W, _p(zero_dq), _stream(), meta=LaunchSize(N, B))
    return qkv_r, o, lse


def mqa_fwd_masked(qkv: torch.Tensor, mask4: torch.Tensor, B: int, N: int, H: int, D: int, out_dtype: torch.dtype, scale: float) -> torch.Tensor:
    """Attend with attn_mask (attention.py:77-99): mask4 = the bf16 mask expanded (as a view: stride 0 on broadcast dims) to (B, H, N, N);
    one K/V head (G = 1) -- the caller repeats K / V heads itself as the reference's Attention does."""
    M, W, ld = _rows(qkv)
    assert qkv.dtype == torch.bfloat16 and W == (H + 2) * D and mask4.dtype == torch.bfloat16 and tuple(mask4.shape) == (B, H, N, N)
    o = torch.empty((B, N, H * D), dtype=out_dtype, device=qkv.device)
    lse = torch.empty((B, H, N), dtype=torch.float32, device=qkv.device)
    base = qkv.data_ptr()
    sb, sh, sq, sk = mask4.stride()
    call("osuf_mqa_fwd_masked", base, ld, base + 2 * H * D, ld, base + 2 * (H + 1) * D, ld, o.data_ptr(), H * D, _DT[out_dtype], lse.data_ptr(),
         mask4.data_ptr(), sb, sh, sq, sk, B, H, N, D, scale, _stream())
    return o


ATTN_AUTO, ATTN_PLAIN, ATTN_PIPE, ATTN_FUSED, ATTN_FUSED_SLABS = 0, 1, 2, 3, 4
ATTN_FUSED256, ATTN_FUSED512, ATTN_FUSED512_TIMING = 5, 6, 7     # force the 256- / 512-key sweep of ATTN_FUSED; 512 without atomics (timing only)
ATTN_FUSED512A = 8                                               # the 512-key sweep with the generated, hand-placed loop
_FUSED_DQ_MODE = {ATTN_FUSED: 0, ATTN_FUSED_SLABS: 1, ATTN_FUSED256: 2, ATTN_FUSED512: 3, ATTN_FUSED512_TIMING: 4, ATTN_FUSED512A: 5}      # OSUF_DQ_*
# What AttentionFn.backward asks for: ATTN_FUSED = one key-stationary sweep, dQ by fp32 atomics (fastest at every UNet shape, measured
# round 2); ATTN_FUSED_SLABS = the same sweep with a fixed-order dQ sum (bit-reproducible); ATTN_AUTO = the dQ + dK/dV kernel pair.
ATTN_BWD_DEFAULT = ATTN_FUSED
FUSE_ROWDOT = True      # AttentionFn.backward: sum_d dO * O from the to_out dgrad GEMM's epilogue (False: the stand-alone osuf_attn_delta pass)


def mqa_bwd(qkv: torch.Tensor, o: torch.Tensor, do: torch.Tensor, lse: torch.Tensor, B: int, N: int, H: int, D: int, scale: float,
            out_dtype: torch.dtype = torch.float32, cos: Optional[torch.Tensor] = None, sin: Optional[torch.Tensor] = None,
            variant: int = ATTN_AUTO, qsplit: int = 0, delta: Optional[torch.Tensor] = None, kv_heads: int = 1, qs: bool = False,
            workspace: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gradients laid out like qkv, [B*N][(H+2G)*D], in out_dtype.  With the RoPE tables (N, D/2) the q / k gradients are those of
    the un-rotated projections (the rotation's transpose is applied in the kernels' epilogues).  kv_heads = G > 1: one launch set
    per group on its column blocks (see mqa_fwd; lse / delta are [G][B][H/G][N]).  workspace: a fused_bwd_workspace whose dQ accumulator this
    layer's forward zero-filled (mqa_fwd(zero_dq=...)): used instead of the shared workspace, without the memset."""
    M, W, ld = _rows(qkv)
    G = kv_heads
    r = H // G
    assert do.dtype == torch.bfloat16 and W == (H + 2 * G) * D and (G == 1 or delta is None)
    if D != 64:
        # head dims 16 / 32 / 128 (csrc/attn_generic.hpp): the dQ + dK/dV kernel pair gives the gradients of the ROTATED q / k; the RoPE
        # transpose is its own pass (the 64-wide kernels fuse it into their epilogues)
        if cos is not None:
            raw = mqa_bwd(qkv, o, do, lse, B, N, H, D, scale, torch.float32, None, None, ATTN_AUTO, 0, delta, kv_heads)
            return rope_bwd(raw, out_dtype, cos, sin, N, H + G, H + 2 * G, D)
        variant, qsplit = ATTN_AUTO, 0
    dqkv = torch.empty((B, N, W), dtype=out_dtype, device=qkv.device)
    es, ldo_, ldo2 = dqkv.element_size(), _rows(do)[2], _rows(o)[2]
    if delta is None:                                      # (B, H, N) sum_d dO * O: given when the to_out dgrad GEMM produced it
        delta = torch.empty((B, H, N) if G == 1 else (G, B, r, N), dtype=torch.float32, device=qkv.device)
        for g in range(G):
            call("osuf_attn_delta", do.data_ptr() + 2 * g * r * D, ldo_, o.data_ptr() + o.element_size() * g * r * D, ldo2, _DT[o.dtype],
                 delta.data_ptr() + 4 * g * B * r * N, B, r, N, D, _stream())
    for g in range(G):
        q_, k_, v_ = qkv.data_ptr() + 2 * g * r * D, qkv.data_ptr() + 2 * (H + g) * D, qkv.data_ptr() + 2 * (H + G + g) * D
        dq_, dk_, dv_ = dqkv.data_ptr() + es * g * r * D, dqkv.data_ptr() + es * (H + g) * D, dqkv.data_ptr() + es * (H + G + g) * D
        do_, lse_, delta_ = do.data_ptr() + 2 * g * r * D, lse.data_ptr() + 4 * g * B * r * N, delta.data_ptr() + 4 * g * B * r * N
        if variant in _FUSED_DQ_MODE:
            mode = _FUSED_DQ_MODE[variant]
            need = _lib.load().osuf_mqa_bwd_fused_workspace_bytes(B, r, N, _DT[out_dtype], qsplit, mode)
            if workspace is not None and G == 1 and getattr(workspace, "fused_variant", None) == variant and workspace.numel() * 4 >= need:
                ws, mode = workspace, mode | DQ_PREZEROED
            else:
                ws = _workspace(need, qkv.device)
            call("osuf_mqa_bwd_fused_qs" if qs else "osuf_mqa_bwd_fused", q_, ld, k_, ld, v_, ld, do_, ldo_, lse_, delta_, dq_, W, dk_, dv_, W, B, r, N, D, scale, _DT[out_dtype],
                 _p(cos), _p(sin), _p(ws), need, qsplit, mode, _stream(), meta=N)
            continue
        assert not qs, "pre-scaled queries: only the fused sweeps have that form"
        call("osuf_mqa_bwd_dq", q_, ld, k_, ld, v_, ld, do_, ldo_, lse_, delta_, dq_, W, B, r, N, D, scale, _DT[out_dtype],
             _p(cos), _p(sin), variant, _stream(), meta=N)
        need = _lib.load().osuf_mqa_bwd_dkv_workspace_bytes(B, N, qsplit)  # > 0: short sequence (or forced), the query range is split
        ws = _workspace(need, qkv.device) if need > 0 else None
        call("osuf_mqa_bwd_dkv", q_, ld, k_, ld, v_, ld, do_, ldo_, lse_, delta_, dk_, dv_, W,
             B, r, N, D, scale, _DT[out_dtype], _p(cos), _p(sin), _p(ws), need if ws is not None else 0, qsplit, variant, _stream(), meta=N)
    return dqkv


# ---------------------------------------------------------------------------------------------------------
# layout / scheduler / optimizer
# ---------------------------------------------------------------------------------------------------------
def ncl_to_rows(x: torch.Tensor, dtype: torch.dtype, width: int, kt: int = 1) -> torch.Tensor:
    """(B, C, L) fp32 contiguous -> rows (B, L, width) [im2col over kt taps when kt > 1]."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.is_cuda
    B, C, Lx = x.shape
    out = torch.empty((B, Lx, width), dtype=dtype, device=x.device)
    call("osuf_ncl_to_rows", _DT[dtype], _p(x), _p(out), width, width, B, C, Lx, kt, _stream())
    return out


def rows_to_ncl(rows: torch.Tensor, C: int) -> torch.Tensor:
    """rows (B, L, >=C) -> (B, C, L) fp32 contiguous."""
    B, Lx = rows.shape[0], rows.shape[1]
    out = torch.empty((B, C, Lx), dtype=torch.float32, device=rows.device)
    call("osuf_rows_to_ncl", dt_of(rows), _p(rows), _rows(rows)[2], _p(out), B, C, Lx, _stream())
    return out


def copy2d(src: torch.Tensor, dst: torch.Tensor, cols: Optional[int] = None) -> torch.Tensor:
    M, W, lds = _rows(src)
    Md, Wd, ldd = _rows(dst)
    cols = W if cols is None else cols
    assert M == Md and Wd >= cols
    call("osuf_copy2d", dt_of(src), _p(src), lds, dt_of(dst), _p(dst), ldd, M, cols, _stream())
    return dst


def cast_rows(src: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    if src.dtype == dtype:
        return src
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    return copy2d(src, dst)


def add_rows(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    M, W, lda = _rows(a)
    out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    call("osuf_add2d", dt_of(a), _p(a), lda, _p(b), _rows(b)[2], _p(out), W, M, W, _stream())
    return out


def axpby_rows(x: torch.Tensor, y: torch.Tensor, ca: torch.Tensor, cb: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    call("osuf_axpby_rows", _p(x), _p(y), _p(ca), _p(cb), _p(out), x.shape[0], x[0].numel(), _stream())
    return out


def ddim_step(x: torch.Tensor, cond: torch.Tensor, null: Optional[torch.Tensor], cond_scale: float, coef: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(x)
    call("osuf_ddim_step", _p(x), _p(cond), _p(null), float(cond_scale), _p(coef), _p(out), x.shape[0], x[0].numel(), _stream())
    return out


def mse(pred: torch.Tensor, target: torch.Tensor, orig_len: Optional[torch.Tensor], want_grad: bool):
    B, Dc, Lx = pred.shape
    grad = torch.empty_like(pred) if want_grad else None
    acc = zeros(1, torch.float64, pred.device)
    ol = orig_len.to(device=pred.device, dtype=torch.int32).contiguous() if orig_len is not None else None
    call("osuf_mse", _p(pred), _p(target), _p(ol), _p(grad), _p(acc), B, Dc, Lx, _stream())
    return acc, grad


def sqnorm(flat: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    call("osuf_sqnorm", _p(flat), flat.numel(), _p(out), _stream())
    return out


def adamw(p, g, m, v, lr, beta1, beta2, eps, wd, step, gscale=None) -> None:
    call("osuf_adamw", _p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, wd, step, _p(gscale), _stream())


def clip_coef(sumsq: torch.Tensor, max_norm: float, base: float, coef: torch.Tensor, total_norm: Optional[torch.Tensor]) -> None:
    call("osuf_clip_coef", _p(sumsq), float(max_norm), float(base), _p(coef), _p(total_norm), _stream())


ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2


def skinny_fwd(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], mode_dtype: torch.dtype, in_act: int = 0, out_act: int = 0):
    """y = out_act(in_act(x) W^T + b) for (M, K) fp32 rows and an fp32 master weight (N, K[, 1]) read in place."""
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.is_cuda
    assert w.dtype == torch.float32 and w.is_contiguous()
    M, K = x.shape
    N = w.shape[0]
    assert w.numel() == N * K
    y = torch.empty((M, N), dtype=torch.float32, device=x.device)
    call("osuf_skinny_fwd", _DT[mode_dtype], _p(x), x.stride(0), _p(w), _p(bias), _p(y), N, M, N, K, in_act, out_act, _stream())
    return y


def skinny_bwd(dy: torch.Tensor, y: Optional[torch.Tensor], x: torch.Tensor, w: torch.Tensor, mode_dtype: torch.dtype, in_act: int, out_act: int,
               want_dx: bool, dw_out: Optional[torch.Tensor], db_out: Optional[torch.Tensor], accumulate: bool):
    """-> dx (or None).  dw_out (N, K[, 1]) fp32 is written (or += when accumulate); db_out (N,) is always added into."""
    M, K = x.shape
    N = w.shape[0]
    assert dy.dtype == torch.float32 and dy.stride(1) == 1 and dy.shape == (M, N)
    dx = torch.empty((M, K), dtype=torch.float32, device=x.device) if want_dx else None
    call("osuf_skinny_bwd", _DT[mode_dtype], _p(dy), dy.stride(0), _p(y), N if y is not None else 0, _p(x), x.stride(0), _p(w), _p(dx), K,
         _p(dw_out), _p(db_out), M, N, K, in_act, out_act, 1 if accumulate else 0, _stream())
    return dx


# osuf_linear_desc (include/osufusion_hip.h), 56 bytes
LINEAR_DESC = np.dtype([("W", "<u8"), ("bias", "<u8"), ("y", "<u8"), ("dy", "<u8"), ("ldy", "<i8"), ("lddy", "<i8"), ("N", "<i4"), ("block0", "<i4")])


def _upload_table(tab: np.ndarray, device) -> torch.Tensor:
    """Descriptor table -> device without stalling the host: pinned staging (the caching host allocator keeps the block alive
    until the copy has run) + an asynchronous copy on the current stream."""
    host = torch.empty(tab.nbytes, dtype=torch.uint8, pin_memory=True)
    host.numpy()[:] = tab.view(np.uint8).reshape(-1)
    return host.to(device, non_blocking=True)


def skinny_group_ok(x: torch.Tensor, ws) -> bool:
    """The group kernels' vector-path conditions (include/osufusion_hip.h)."""
    return (x.dtype == torch.float32 and x.dim() == 2 and x.is_cuda and x.stride(1) == 1 and x.shape[1] % 8 == 0 and x.stride(0) % 4 == 0 and
            x.data_ptr() % 16 == 0 and all(w.dtype == torch.float32 and w.is_contiguous() and w.shape[0] % 8 == 0 and w.data_ptr() % 16 == 0 and
                                           w.numel() == w.shape[0] * x.shape[1] for w in ws))


def skinny_fwd_group(x: torch.Tensor, ws, bs, mode_dtype: torch.dtype, in_act: int):
    """[in_act(x) W_i^T + b_i for i] for fp32 master weights (N_i, K[, 1]) that share the input rows x (M, K): one launch."""
    assert skinny_group_ok(x, ws)
    M, K = x.shape
    tab = np.zeros(len(ws), dtype=LINEAR_DESC)
    ys, blocks = [], 0
    for j, (w, b) in enumerate(zip(ws, bs)):
        N = w.shape[0]
        y = torch.empty((M, N), dtype=torch.float32, device=x.device)
        ys.append(y)
        tab[j] = (w.data_ptr(), b.data_ptr() if b is not None else 0, y.data_ptr(), 0, N, 0, N, blocks)
        blocks += -(-N // 32)
    dev = _upload_table(tab, x.device)
    call("osuf_skinny_fwd_group", _DT[mode_dtype], _p(x), x.stride(0), dev.data_ptr(), len(ws), blocks, M, K, in_act, _stream())
    return ys


def skinny_dx_group(dys, ws, x: torch.Tensor, mode_dtype: torch.dtype, in_act: int) -> torch.Tensor:
    """dx = in_act'(x) * sum_i dy_i W_i over the linears of a group (dys[i] None: that output had no gradient)."""
    M, K = x.shape
    live = [(dy, w) for dy, w in zip(dys, ws) if dy is not None]
    dx = torch.empty((M, K), dtype=torch.float32, device=x.device)
    if not live:
        return dx.zero_()
    tab = np.zeros(len(live), dtype=LINEAR_DESC)
    slices = 0
    for j, (dy, w) in enumerate(live):
        N = w.shape[0]
        assert dy.dtype == torch.float32 and dy.shape == (M, N) and dy.is_contiguous() and dy.data_ptr() % 16 == 0
        tab[j] = (w.data_ptr(), 0, 0, dy.data_ptr(), 0, N, N, slices)
        slices += -(-N // 512)
    dev = _upload_table(tab, x.device)
    call("osuf_skinny_dx_group", _DT[mode_dtype], dev.data_ptr(), len(live), slices, _p(x), x.stride(0), _p(dx), K, M, K, in_act, _stream())
    return dx


_DKIND = {"same": 0, "down": 1, "up": 2}


def _pack_geometry(w: torch.Tensor, dtype: torch.dtype, kind: str, want_fwd: bool, want_dgrad: bool, fwd, dgrad, row_offset: int):
    """Destinations (allocated when not given) and the pointer / stride arguments osuf_pack_weight takes for them."""
    assert w.dtype == torch.float32 and w.is_contiguous() and w.is_cuda
    O, I = w.shape[0], w.shape[1]
    k = w.shape[2] if w.dim() == 3 else 1
    kd = k if kind == "same" else 4
    if want_fwd and fwd is None:
        fwd = torch.empty((k, O, I), dtype=dtype, device=w.device)
    if want_dgrad and dgrad is None:
        dgrad = torch.empty((kd, I, O), dtype=dtype, device=w.device)
    pf = pd = None
    f_ld = f_ts = d_ld = d_ts = 0
    if want_fwd:
        assert fwd.dtype == dtype and fwd.is_contiguous() and fwd.shape[0] == k and fwd.shape[2] == I
        f_ld, f_ts = I, fwd.shape[1] * I
        pf = fwd.data_ptr() + row_offset * I * fwd.element_size()
    if want_dgrad:
        assert dgrad.dtype == dtype and dgrad.is_contiguous() and dgrad.shape[0] == kd and dgrad.shape[1] == I
        d_ld, d_ts = dgrad.shape[2], I * dgrad.shape[2]
        pd = dgrad.data_ptr() + row_offset * dgrad.element_size()
    return fwd, dgrad, (O, I, k, pf, f_ld, f_ts, pd, d_ld, d_ts, _DKIND[kind])


def pack_weight(w: torch.Tensor, dtype: torch.dtype, kind: str = "same", want_fwd: bool = True, want_dgrad: bool = True,
                fwd: Optional[torch.Tensor] = None, dgrad: Optional[torch.Tensor] = None, row_offset: int = 0, adapt=None):
    """fp32 (O, I[, k]) master -> (fwd [k][O][I], dgrad [k'][I][O]) GEMM operands in `dtype`, one launch.  With fwd / dgrad
    given, this weight's rows are written at row_offset of a larger stacked operand (the fused q|kv projection).
    adapt = (lora_A, lora_B, g or None, scaling): pack the LoRA / DoRA effective weight g*(w + s*BA) instead of w."""
    fwd, dgrad, (O, I, k, pf, f_ld, f_ts, pd, d_ld, d_ts, dk) = _pack_geometry(w, dtype, kind, want_fwd, want_dgrad, fwd, dgrad, row_offset)
    if adapt is not None:
        a, b, g, scaling = adapt
        r = a.shape[0]
        for t in (a, b):
            assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
        assert a.numel() == r * I * k and b.numel() == O * r and (g is None or (g.dtype == torch.float32 and g.numel() == O))
        call("osuf_pack_weight_adapted", _p(w), _p(a), _p(b), _p(g), float(scaling), r, O, I, k, 1 if dtype == torch.bfloat16 else 0,
             pf, f_ld, f_ts, pd, d_ld, d_ts, dk, _stream())
    else:
        call("osuf_pack_weight", _p(w), O, I, k, 1 if dtype == torch.bfloat16 else 0, pf, f_ld, f_ts, pd, d_ld, d_ts, dk, _stream())
    return (fwd if want_fwd else None), (dgrad if want_dgrad else None)


# osuf_pack_desc (include/osufusion_hip.h), 80 bytes
PACK_DESC = np.dtype([("w", "<u8"), ("F", "<u8"), ("D", "<u8"), ("f_ld", "<i8"), ("f_ts", "<i8"), ("d_ld", "<i8"), ("d_ts", "<i8"),
                      ("O", "<i4"), ("I", "<i4"), ("k", "<i4"), ("dkind", "<i4"), ("block0", "<i4"), ("reserved", "<i4")])


def pack_desc_table(items, dtype: torch.dtype, device) -> Tuple[torch.Tensor, int, int]:
    """items: (w, kind, fwd, dgrad, row_offset) per weight, destinations already allocated (as pack_weight wrote them before).
    -> (device table of osuf_pack_desc, n, total_blocks) for pack_weight_group."""
    tab = np.zeros(len(items), dtype=PACK_DESC)
    blocks = 0
    for j, (w, kind, fwd, dgrad, row_offset) in enumerate(items):
        _, _, (O, I, k, pf, f_ld, f_ts, pd, d_ld, d_ts, dk) = _pack_geometry(w, dtype, kind, fwd is not None, dgrad is not None, fwd, dgrad, row_offset)
        tab[j] = (w.data_ptr(), pf or 0, pd or 0, f_ld, f_ts, d_ld, d_ts, O, I, k, dk, blocks, 0)
        blocks += -(-O // 32) * -(-I // 32)
    dev = torch.from_numpy(tab.view(np.uint8)).to(device)
    return dev, len(items), blocks


def pack_weight_group(table: torch.Tensor, n: int, total_blocks: int, dtype: torch.dtype) -> None:
    """osuf_pack_weight for every descriptor of the table in one launch."""
    assert table.is_cuda and table.dtype == torch.uint8 and table.numel() == n * PACK_DESC.itemsize
    call("osuf_pack_weight_group", table.data_ptr(), n, total_blocks, 1 if dtype == torch.bfloat16 else 0, _stream())


def dora_gain(w: torch.Tensor, a: torch.Tensor, b: torch.Tensor, mag: Optional[torch.Tensor], scaling: float):
    """-> (g (O,), (s g B)^T as [1][r][O] f32, the same in bf16) for the frozen base w (O, I[, k]), lora_A (r, I[, k]), lora_B (O, r[, 1])
    and the DoRA magnitude (O values) or None (plain LoRA: g = 1)."""
    for t in (w, a, b):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    O, I, r = w.shape[0], w.shape[1], a.shape[0]
    k = w.shape[2] if w.dim() == 3 else 1
    assert a.numel() == r * I * k and b.numel() == O * r
    part = None
    if mag is not None:
        assert mag.dtype == torch.float32 and mag.is_contiguous() and mag.numel() == O
        part = torch.empty(((I + 31) // 32, O), dtype=torch.float32, device=w.device)
    g = torch.empty(O, dtype=torch.float32, device=w.device)
    t32 = torch.empty((1, r, O), dtype=torch.float32, device=w.device)
    t16 = torch.empty((1, r, O), dtype=torch.bfloat16, device=w.device)
    call("osuf_dora_gain", _p(w), _p(a), _p(b), _p(mag), O, I, k, r, float(scaling), _p(part), _p(g), _p(t32), _p(t16), _stream())
    return g, t32, t16


def dora_effective(w: torch.Tensor, a: torch.Tensor, b: torch.Tensor, mag: Optional[torch.Tensor], scaling: float):
    """(Weff shaped like w, g (O,)) from the frozen base w (O, I[, k]), lora_A (r, I[, k]), lora_B (O, r[, 1]) and the DoRA magnitude
    (O values, any shape) or None for plain LoRA."""
    for t in (w, a, b):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    O, r = w.shape[0], a.shape[0]
    IK = w.numel() // O
    assert a.numel() == r * IK and b.numel() == O * r
    if mag is not None:
        assert mag.dtype == torch.float32 and mag.is_contiguous() and mag.numel() == O
    weff = torch.empty_like(w)
    g = torch.empty(O, dtype=torch.float32, device=w.device)
    call("osuf_dora_effective", _p(w), _p(a), _p(b), _p(mag), O, IK, r, float(scaling), _p(weff), _p(g), _stream())
    return weff, g


def adapter_finish(tb, sg, db_out, gt, da_out, s0, s1, bias, m, dm_out, O: int, I: int, k: int, r: int, accumulate: bool) -> None:
    """dB (+)= sg*tb; dA (+)= flipped / transposed gt; dm (+)= (s0 - bias*s1)/m -- see osuf_adapter_finish."""
    for t in (tb, sg, db_out, gt, da_out, s0, s1, bias, m, dm_out):
        assert t is None or (t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda)
    assert tb.numel() == O * r and db_out.numel() == O * r and gt.numel() == k * I * r and da_out.numel() == r * I * k and sg.numel() == O
    call("osuf_adapter_finish", _p(tb), _p(sg), _p(db_out), _p(gt), _p(da_out), _p(s0), _p(s1), _p(bias), _p(m), _p(dm_out), O, I, k, r,
         1 if accumulate else 0, _stream())


def cast_f32_bf16(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    call("osuf_cast_f32_bf16", _p(src), _p(dst), src.numel(), _stream())
    return dst


def fir_decimate2(sig: torch.Tensor, taps: torch.Tensor) -> torch.Tensor:
    """out[m] = sum_j taps[j] sig[2m + j - (len(taps)-1)/2], m < ceil(len(sig)/2) (zero extension)."""
    assert sig.is_cuda and sig.dtype == torch.float32 and sig.dim() == 1 and sig.is_contiguous() and taps.dtype == torch.float32
    n_out = (sig.numel() + 1) // 2
    out = torch.empty(n_out, dtype=torch.float32, device=sig.device)
    call("osuf_fir_decimate2", _p(sig), sig.numel(), _p(taps), taps.numel(), _p(out), n_out, _stream())
    return out


def frame_rows(sig: torch.Tensor, hop: int, K: int, frames: int) -> torch.Tensor:
    """(frames, K) rows sig[t*hop : t*hop + K] (zeros past the end)."""
    assert sig.is_cuda and sig.dtype == torch.float32 and sig.dim() == 1 and sig.is_contiguous()
    out = torch.empty((frames, K), dtype=torch.float32, device=sig.device)
    call("osuf_frame_rows", _p(sig), sig.numel(), hop, K, _p(out), frames, _stream())
    return out


def log_vqt(wave_pad: torch.Tensor, bank: torch.Tensor, scale: torch.Tensor, hop: int, frames: int, eps: float = 1e-10,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out[k][t] = log(scale[k] |sum_n wave_pad[t*hop+n] (bank[k][n] + i bank[bins+k][n])| + eps) -> (bins, frames) fp32 (audio.hip);
    `out`: rows of a wider (bins_total, frames) result (row stride = its stride)."""
    assert wave_pad.is_cuda and wave_pad.dtype == torch.float32 and wave_pad.dim() == 1 and wave_pad.is_contiguous()
    assert bank.dtype == torch.float32 and bank.dim() == 2 and bank.is_contiguous() and bank.shape[0] % 2 == 0
    bins, K = bank.shape[0] // 2, bank.shape[1]
    assert scale.dtype == torch.float32 and scale.numel() == bins and hop % 4 == 0 and K % 4 == 0
    assert (frames - 1) * hop + K <= wave_pad.numel()
    ws = torch.empty((frames, 2 * bins), dtype=torch.float32, device=wave_pad.device)
    if out is None:
        out = torch.empty((bins, frames), dtype=torch.float32, device=wave_pad.device)
    assert out.dtype == torch.float32 and out.shape == (bins, frames) and out.stride(1) == 1
    call("osuf_log_vqt", _p(wave_pad), wave_pad.numel(), _p(bank), K, bins, hop, _p(scale), eps, _p(ws), _p(out), out.stride(0), frames, _stream())
    return out
