"""CPU check of every GLOBAL address the generated attention-backward loop forms (csrc/attn_bwd512_asm.inc, mqa_bwd_fused512a_kernel).

The loop carries 32-bit running byte offsets (v228/v229: dQ atomics, v252/v253: Q / dO LDS-DMA requests, v254: row constants) that a scalar
state machine advances head by head and wraps block by block; a wrong delta, a missed stop past the last pair or a 32-bit wrap-around is a
wild address on the GPU (round 4 saw one memory-access fault during the bring-up of the list-scheduled stream and had no tool that would
have named the access).  This script shares no code with tools/gen_attn_bwd512.py: it READS THE EMITTED TEXT, executes its scalar
instructions and the vector adds on the offset registers for all 4 x 64 lanes of a workgroup -- starting from the register values the C++
prologue of the kernel hands to the asm statement, restated here from csrc/attn.hip -- and checks at every vector-memory instruction that
  * each lane's address is exactly the element the algorithm wants there (Q / dO piece, lse / delta row constant, dQ element of the pair
    the iteration finishes), hence inside the tensor part the workgroup owns;
  * the LDS destination of every LDS-DMA (m0) is one of the two stage slots;
  * every register the text writes that is not a '+' / '=' operand of the asm statement is in its clobber list (m0).
Shapes: every (N, H, qsplit, row strides) family the launcher admits, including H == 1, the minimum trip count and strides near the 32-bit
guard of fused512a_ok().       python tools/check_bwd512a_addresses.py [--quick] [file.inc]   (default: attn_bwd512_asm.inc; the pre-scaled-query
variant attn_bwd512qs_asm.inc forms the same addresses and is checked the same way)"""
from __future__ import annotations

import re
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
INC = ROOT / "osufusion_amd" / "csrc" / "attn_bwd512_asm.inc"
D = 64
K_STAGE = 4096 + 4096 + 256
U32 = 0xFFFFFFFF


def inc_lines(path):
    out, clob = [], None
    for l in open(path):
        m = re.match(r'\s*"(.*)\\n\\t" \\$', l)
        if m:
            out.append(m.group(1))
        if re.match(r"#define OSUF_BWD512A(QS)?_CLOBBERS", l):
            clob = set(re.findall(r'"([^"]+)"', l))
    return out, clob


# ---------------------------------------------------------------------------------------------------------------------------------
# what the asm statement may write: its '=' / '+' operands (csrc/attn.hip: v[0:255] outputs, "+{s60}", "+{s63}") and its clobbers
# ---------------------------------------------------------------------------------------------------------------------------------
def check_clobbers(lines, clob):
    allowed = {f"v{i}" for i in range(256)} | {"s60", "s63"} | clob
    bad = []
    for l in lines:
        if l.endswith(":"):
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")] if rest else []
        written = []
        if op.startswith("s_") and not op.startswith(("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_setprio", "s_sleep")):
            written = [toks[0]]
            if op.startswith(("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr")):
                written.append("scc")
        elif op.startswith("s_cmp"):
            written = ["scc"]
        elif op.startswith("v_accvgpr_write"):
            written = [toks[0]]
        elif op.startswith(("v_cmp", )):
            written = ["vcc"]
        for w in written:
            regs = []
            m = re.match(r"([sva])\[(\d+):(\d+)\]", w)
            if m:
                regs = [f"{m.group(1)}{i}" for i in range(int(m.group(2)), int(m.group(3)) + 1)]
            else:
                regs = [w]
            for r in regs:
                if r not in allowed:
                    bad.append((r, l))
    return bad


# ---------------------------------------------------------------------------------------------------------------------------------
# the kernel's prologue (csrc/attn.hip, mqa_bwd_fused512a_kernel), restated: register values at the asm statement's entry
# ---------------------------------------------------------------------------------------------------------------------------------
class Shape:
    def __init__(self, B, N, H, qsplit, ldq=None, lddo=None, b=0, part=0):
        self.B, self.N, self.H, self.qsplit, self.b, self.part = B, N, H, qsplit, b, part
        self.ldq = ldq if ldq is not None else (H + 2) * D           # q | k | v in one row (elements)
        self.lddo = lddo if lddo is not None else H * D
        self.nqb = N // 32
        self.qb_per = self.nqb // qsplit
        self.qb_begin = part * self.qb_per
        self.niter = self.qb_per * H
        # distinct, far-apart fake base addresses (bytes) so that a mix-up of bases shows
        self.q, self.dout, self.lse2, self.delta, self.dq32 = 0x1_0000_0000, 0x3_0000_0000, 0x5_0000_0000, 0x6_0000_0000, 0x8_0000_0000
        self.sb = 0                                                      # LDS base of the dynamic array

    def guard(self):                                                     # fused512a_ok() of csrc/attn.hip (host side)
        ok = self.N % 512 == 0 and self.N >= 512 and self.nqb % self.qsplit == 0 and self.niter % 2 == 0 and self.niter >= 2
        ok = ok and self.N * self.ldq * 2 < (1 << 31) and self.N * self.lddo * 2 < (1 << 31)
        ok = ok and (self.N + 32) * self.H * D * 4 < (1 << 31) and self.H * self.N * 4 < (1 << 31)
        return ok


def entry_state(sh: Shape, wave: int):
    lane = np.arange(64, dtype=np.int64)
    H, N = sh.H, sh.N
    s = {}
    v = {}
    row = 8 * wave + (lane >> 3)
    x = (row >> 1) & 7
    f = ((x & 1) << 2) | (x >> 1)
    chunk = (lane & 7) ^ f
    g4, ip = lane >> 4, lane & 15
    aoffb = ((4 * g4) * (H * D) + wave * 16 + ip) * 4
    rowb = H * D * 4
    v[228] = aoffb & U32
    v[229] = (aoffb + 16 * rowb) & U32
    wq = (32 * sh.ldq - (H - 1) * D) * 2
    wdo = (32 * sh.lddo - (H - 1) * D) * 2
    wls = (32 - (H - 1) * N) * 4
    wdq = (32 * H * D - (H - 1) * D) * 4
    v[252] = ((row * sh.ldq + chunk * 8) * 2 + (D * 2 if H > 1 else wq)) & U32
    v[253] = ((row * sh.lddo + chunk * 8) * 2 + (D * 2 if H > 1 else wdo)) & U32
    v[254] = ((lane & 31) * 4 + (N * 4 if H > 1 else wls)) & U32
    hn = sh.b * H * N
    first_row = sh.b * N + sh.qb_begin * 32

    def pair(base, val):
        s[base], s[base + 1] = val & U32, (val >> 32) & U32
    pair(48, sh.q + first_row * sh.ldq * 2)
    pair(50, sh.dout + first_row * sh.lddo * 2)
    pair(52, sh.lse2 + (hn + sh.qb_begin * 32) * 4)
    pair(54, sh.delta + (hn + sh.qb_begin * 32) * 4)
    pair(56, sh.dq32 + first_row * (H * D) * 4)
    s[60] = sh.niter // 2
    s[63] = (sh.qb_per - 1 - (0 if H > 1 else 1)) & U32
    s[61], s[65] = H, (N * 4) & U32
    s[66], s[68], s[70], s[72] = wq & U32, wdo & U32, wls & U32, wdq & U32
    s[74], s[75] = 0x3E000000, 0xC1000000                                # c, -1/c: any float bits (not address relevant)
    s[76] = sh.sb + wave * 1024
    s[77] = 2 * K_STAGE + wave * 16384
    s[80] = rowb & U32
    return s, v, dict(row=row, chunk=chunk, g4=g4, ip=ip, lane=lane)


# ---------------------------------------------------------------------------------------------------------------------------------
# expected addresses
# ---------------------------------------------------------------------------------------------------------------------------------
def pair_of(sh, p):
    p = min(max(p, 0), sh.niter - 1)
    return sh.qb_begin + p // sh.H, p % sh.H


def expect_q(sh, L, wave, p, which):
    qb, h = pair_of(sh, p)
    base, ld = (sh.q, sh.ldq) if which == "q" else (sh.dout, sh.lddo)
    return base + ((sh.b * sh.N + qb * 32 + L["row"]) * ld + h * D + L["chunk"] * 8) * 2


def expect_c(sh, L, p, which):
    qb, h = pair_of(sh, p)
    base = sh.lse2 if which == "lse" else sh.delta
    return base + ((sh.b * sh.H + h) * sh.N + qb * 32 + (L["lane"] & 31)) * 4


def expect_dq(sh, L, wave, p, i):
    """atomic i (0..7): row i & 3 of query half i >> 2 of the pair's 32 x 64 dQ tile (this wave: columns 16 wave ..)"""
    qb, h = pair_of(sh, p)
    r = (i & 3) + 4 * L["g4"] + 16 * (i >> 2)
    return sh.dq32 + ((sh.b * sh.N + qb * 32 + r) * (sh.H * D) + h * D + wave * 16 + L["ip"]) * 4


# ---------------------------------------------------------------------------------------------------------------------------------
# interpreter of the scalar / offset-register subset
# ---------------------------------------------------------------------------------------------------------------------------------
TRACKED_V = (228, 229, 252, 253, 254)


def sval(s, tok):
    tok = tok.strip()
    if re.fullmatch(r"s\d+", tok):
        return s[int(tok[1:])]
    if tok == "m0":
        return s["m0"]
    return int(tok, 0) & U32


def run(lines, sh: Shape, wave: int):
    s, v, L = entry_state(sh, wave)
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    loop_top = labels[".Lloop%="]
    scc, pc, steps = 0, 0, 0
    n_req = n_c = 0                                                   # requests issued so far (the k-th asks for pair k + 1)
    body = -1                                                         # bodies started (a body = one (head, query block) pair)
    atom_in_body = 0
    errors, checked = [], 0
    in_tail = False
    # a body starts at the loop top and at the second copy inside the two-body text: detect bodies by the DMA requests (one Q request each)
    while pc < len(lines):
        steps += 1
        assert steps < 5_000_000, "runaway interpretation"
        l = lines[pc]
        pc += 1
        if l.endswith(":"):
            if l[:-1] == ".Lend%=":
                break
            continue
        op, _, rest = l.partition(" ")
        t = [x.strip() for x in rest.split(",")] if rest else []
        if op in ("s_mov_b32", "s_movk_i32"):
            val = sval(s, t[1]) if op == "s_mov_b32" else (int(t[1], 0) & 0xFFFF) | (0xFFFF0000 if int(t[1], 0) & 0x8000 else 0)
            s["m0" if t[0] == "m0" else int(t[0][1:])] = val & U32
        elif op == "s_mov_b64":
            d, sr = int(re.match(r"s\[(\d+)", t[0]).group(1)), int(re.match(r"s\[(\d+)", t[1]).group(1))
            s[d], s[d + 1] = s[sr], s[sr + 1]
        elif op == "s_add_u32":
            r = sval(s, t[1]) + sval(s, t[2])
            scc = int(r > U32)
            s["m0" if t[0] == "m0" else int(t[0][1:])] = r & U32
        elif op == "s_addc_u32":
            r = sval(s, t[1]) + sval(s, t[2]) + scc
            scc = int(r > U32)
            s[int(t[0][1:])] = r & U32
        elif op == "s_sub_u32":
            a, b_ = sval(s, t[1]), sval(s, t[2])
            scc = int(b_ > a)
            s[int(t[0][1:])] = (a - b_) & U32
        elif op == "s_cmp_eq_u32":
            scc = int(sval(s, t[0]) == sval(s, t[1]))
        elif op == "s_cmp_lg_u32":
            scc = int(sval(s, t[0]) != sval(s, t[1]))
        elif op == "s_cselect_b32":
            s[int(t[0][1:])] = sval(s, t[1]) if scc else sval(s, t[2])
        elif op == "s_cbranch_scc1":
            if scc:
                pc = labels[t[0]]
        elif op == "s_branch":
            pc = labels[t[0]]
        elif op.startswith("s_"):
            assert op in ("s_nop", "s_waitcnt", "s_barrier"), f"scalar instruction the checker does not know: {l}"
        elif op == "v_add_u32_e32" and t[0] in {f"v{r}" for r in TRACKED_V}:
            d = int(t[0][1:])
            assert t[2] == t[0] and re.fullmatch(r"s\d+", t[1]), l
            v[d] = (v[d] + sval(s, t[1])) & U32
        elif op.startswith("v_accvgpr_read"):
            in_tail = True                                              # the accumulators go back to v0..v255: the offsets are dead from here
        elif op.startswith("global_"):
            if in_tail:
                errors.append(f"vector-memory instruction behind the accumulator hand-back: {l}")
                continue
            base = int(re.search(r"s\[(\d+):", t[-1]).group(1))
            sbase = s[base] | (s[base + 1] << 32)
            if op == "global_load_lds_dwordx4":
                off = v[int(t[0][1:])]
                which = "q" if base == 48 else "do"
                assert base in (48, 50), l
                if which == "q":
                    body += 1                                           # (one Q request per body, issued near its head)
                    atom_in_body = 0
                want = expect_q(sh, L, wave, body + 1, which)
                got = sbase + off
                slot = (s["m0"] - sh.sb - wave * 1024)
                if slot not in ((0, K_STAGE) if which == "q" else (4096, K_STAGE + 4096)):
                    errors.append(f"body {body}: LDS-DMA destination m0 - wave base = {slot}: {l}")
                if body + 1 < sh.niter:
                    if not np.array_equal(got, want):
                        errors.append(f"body {body} wave {wave}: {which} request of pair {body + 1} off by {int((got - want)[np.argmax(got != want)])} bytes: {l}")
                else:                                                    # past the last pair the request repeats a valid one
                    lo, hi = (sh.q, sh.ldq) if which == "q" else (sh.dout, sh.lddo)
                    part_lo = lo + (sh.b * sh.N + sh.qb_begin * 32) * hi * 2
                    part_hi = lo + (sh.b * sh.N + (sh.qb_begin + sh.qb_per) * 32) * hi * 2
                    if got.min() < part_lo or got.max() + 16 > part_hi:
                        errors.append(f"body {body} wave {wave}: {which} request past the last pair leaves the part: {l}")
                checked += 1
            elif op == "global_load_dword":
                off = v[int(t[1][1:])]
                which = "lse" if base == 52 else "delta"
                assert base in (52, 54), l
                got = sbase + off
                if body + 1 < sh.niter:
                    want = expect_c(sh, L, body + 1, which)
                    if not np.array_equal(got, want):
                        errors.append(f"body {body} wave {wave}: {which} load of pair {body + 1} off by {int((got - want)[np.argmax(got != want)])} bytes: {l}")
                else:
                    lo = sh.lse2 if which == "lse" else sh.delta
                    if got.min() < lo + sh.b * sh.H * sh.N * 4 or got.max() + 4 > lo + (sh.b + 1) * sh.H * sh.N * 4:
                        errors.append(f"body {body} wave {wave}: {which} load past the last pair leaves the sample: {l}")
                checked += 1
            elif op == "global_atomic_add_f32":
                off = v[int(t[0][1:])]
                row = (base - 82) // 2
                half = 0 if t[0] == "v228" else 1
                i = row + 4 * half
                # body i finishes the dQ of pair i - 1 (body 0 adds the zero image to pair 0); the tail behind the loop finishes the last pair
                p = sh.niter - 1 if body >= sh.niter - 1 and atom_in_body >= 8 else max(body - 1, 0)
                want = expect_dq(sh, L, wave, p, i)
                got = sbase + off
                if not np.array_equal(got, want):
                    errors.append(f"body {body} wave {wave}: dQ atomic {i} of pair {p} off by {int((got - want)[np.argmax(got != want)])} bytes: {l}")
                atom_in_body += 1
                checked += 1
            else:
                errors.append(f"vector-memory instruction the checker does not know: {l}")
        elif op.startswith("v_") or op.startswith("ds_"):
            dst = t[0] if t else ""
            if dst in {f"v{r}" for r in TRACKED_V}:
                errors.append(f"an offset register is written by an instruction the checker does not model: {l}")
    if body + 1 != sh.niter:
        errors.append(f"{body + 1} bodies executed, {sh.niter} pairs")
    return errors, checked


SHAPES = [
    # (B, N, H, qsplit, ldq, lddo)            UNet shapes: q|k|v rows of (H + 2) * 64, dO rows of H * 64
    (1, 512, 2, 16, None, None),              # minimum trip count: one block of two heads
    (2, 512, 16, 4, None, None),
    (2, 1024, 16, 2, None, None),
    (9, 1024, 4, 2, None, None),
    (3, 1024, 3, 1, None, None),              # odd H (niter even through the block count)
    (2, 1024, 1, 1, None, None),              # H == 1: every pair opens a block
    (2, 2048, 1, 2, 64, 64),                  # H == 1 on a separate q tensor
    (8, 2048, 16, 1, None, None),
    (32, 4096, 16, 1, None, None),            # the headline shape
    (2, 4096, 16, 4, None, None),
    (2, 8192, 16, 1, None, None),             # config 4's length
    (1, 4096, 16, 1, 262136, 262136),         # row strides at the launcher's 2^31-byte guard (N * ld * 2 just under it)
    (1, 8192, 32, 1, None, None),             # dQ part of 8192 * 32 * 64 * 4 = 64 MiB, offsets far from the guard but past 2^26
]


def check_text(lines, clob, quick=False, verbose=True):
    """(findings, vector-memory instructions checked).  quick: every shape family, but the long sweeps (more than 600 pairs per part) on
    wave 0 of the last part only -- a few seconds, for the CPU test suite."""
    bad = check_clobbers(lines, clob)
    if verbose:
        for r, l in bad[:10]:
            print(f"CLOBBER: {r} is written but is neither an output operand nor a clobber: {l}")
    total_err, total_chk = len(bad), 0
    for B, N, H, qs, ldq, lddo in SHAPES:
        long_sweep = (N // 32 // qs) * H > 600
        for b, part in sorted({(0, 0), (B - 1, qs - 1)}):
            if quick and long_sweep and (b, part) != (B - 1, qs - 1):
                continue
            sh = Shape(B, N, H, qs, ldq, lddo, b=b, part=part)
            assert sh.guard(), ("shape refused by the launcher's guard", B, N, H, qs, ldq, lddo)
            for wave in range(4):
                if quick and long_sweep and wave:
                    continue
                errs, chk = run(lines, sh, wave)
                total_chk += chk
                if verbose:
                    for e_ in errs[:4]:
                        print(f"B={B} N={N} H={H} qsplit={qs} ldq={sh.ldq} b={b} part={part}: {e_}")
                total_err += len(errs)
    return total_err, total_chk


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = Path(args[0]) if args else INC
    lines, clob = inc_lines(path)
    total_err, total_chk = check_text(lines, clob, quick="--quick" in sys.argv)
    print(f"{path.name}: {total_chk} vector-memory instructions checked lane by lane over {len(SHAPES)} shapes, {total_err} finding(s)")
    return total_err


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
