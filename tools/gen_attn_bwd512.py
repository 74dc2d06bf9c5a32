#!/usr/bin/env python3
"""Generator of the hand-placed main loop of the 512-key fused attention backward (csrc/attn.hip, mqa_bwd_fused512a_kernel).

The loop of mqa_bwd_fused512_kernel as hipcc schedules it issues 764 instructions per (head, 32-query block) pair and wave for 96 MFMAs
and is bound by that issue (profiles/r03_sq_attn_bwd/summary.txt).  Here the same algorithm -- same LDS images, same fragment maps, same
order of every accumulation, so dK / dV come out bit-identical -- is emitted as ONE inline-asm statement with fixed physical registers:

  * every LDS address is a loop-invariant lane offset (VGPR) + an immediate: the loop is unrolled by two so that the stage slot and the
    dS image slot are compile-time constants (no address adds);
  * the transposed dO / Q fragments of a pair are read once and kept for its four key tiles (the compiled loop re-reads them per tile);
  * Q / dO tiles arrive by LDS-DMA one pair ahead (no staging registers, no ds_write_b128); the float atomics stay in flight behind
    counted waits: the only vmcnt(0) of an iteration sits before the row constants of the next stage, a full iteration after them;
  * software pipeline across key tiles: the matrix instructions are issued in the order dV(t) | S(t+1) | dK(t) | dP(t+1), so that the
    exp2 of tile t+1 runs beside dK(t) / dP(t+1) and dS = p dP' of tile t beside dV(t); one dQ k-step (2 x 16x16x32) after every group
    of four 32x32x16 MFMAs;
  * s_waitcnt lgkmcnt(N) / s_nop are inserted by this script from a model of the in-order LDS queue and of the MFMA result latencies
    (cdna_hip_programming.md 5.7: hipcc pads nothing around an asm statement).

Output: osufusion_amd/csrc/attn_bwd512_asm.inc (macros OSUF_BWD512A_ASM / _CLOBBERS; the register map is shared with the C++ side through
the constants printed at its top).   usage: python tools/gen_attn_bwd512.py [--stats]
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
OUT = ROOT / "osufusion_amd" / "csrc" / "attn_bwd512_asm.inc"

# ---- register map (arch VGPRs) -----------------------------------------------------------------------------------------------
VF, QA, DA, TRD, TRQ, S, DP, PF, DF, KF = 0, 64, 80, 96, 112, 128, 144, 160, 168, 176
DQB, DQA0, DQA1, ACC0, ACC1, AO, CL, CD, TMP0, TMP1 = 192, 196, 200, 204, 208, 212, 220, 221, 222, 223
RS, RK, T, KO, EO, EW, CR, CW, QOFF, DOOFF, COFF = 224, 228, 232, 236, 238, 242, 246, 247, 248, 249, 250
# AGPRs: dK^T tile t, head-dim half dt at a[32 t + 16 dt ..+15]; dV^T at a[128 + 32 t + 16 dt ..]
def DK(t, dt): return 32 * t + 16 * dt
def DV(t, dt): return 128 + 32 * t + 16 * dt
# SGPRs
SQ, SDO, SLS, SDL, SDQC, SDQP = 48, 50, 52, 54, 56, 58          # 64-bit pointers (pairs): stage Q / dO / lse2 / delta, dQ rows of this / the previous pair
SCNT, SH, SHS, SREM, SHD, SN4 = 60, 61, 62, 63, 64, 65
SWQ, SWDO, SWLS, SWDQ = 66, 68, 70, 72                         # 64-bit wrap deltas (bytes) at the last head of a query block
SC, SNRC, SM0, STMP, SMASK = 74, 75, 76, 77, 78                  # c = scale log2 e, -1/c, LDS byte address of this wave's 1-KiB DMA piece, tmp, lane mask (pair)
# LDS map (bytes): [2 x (Q 4096 | dO 4096 | -lse/c 128 | -delta 128)] | K image 65536 | 2 x dS image 32768
STAGE, KIMG, EIMG = 8448, 16896, 82432


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n):
    return f"a[{base}:{base + n - 1}]"


def vset(base, n=1):
    return {("v", base + i) for i in range(n)}


def aset(base, n):
    return {("a", base + i) for i in range(n)}


class Emitter:
    """Collects instructions; inserts s_waitcnt lgkmcnt / vmcnt for register dependences on outstanding LDS reads / global loads and s_nop
    for the MFMA hazards hipcc would have padded."""
    MFMA32_WS, MFMA16_WS, VALU_TO_MFMA_WS = 12, 8, 2

    def __init__(self):
        self.lines = []
        self.lds = []            # outstanding DS ops in issue order: set of written regs (empty for stores)
        self.vm = []             # outstanding VMEM ops in issue order: set of written regs (empty for DMA / atomics)
        self.ws = 0              # wait-state clock (1 per instruction, N + 1 per s_nop N)
        self.mfma_ready = {}     # reg -> ws at which a non-MFMA reader / an MFMA A,B reader may issue
        self.mfma_chain = {}     # reg -> (kind) of the MFMA that wrote it last (back-to-back srcC == vdst chains need no padding)
        self.valu_ws = {}        # reg -> ws of the last VALU write
        self.count = {}
        self.nops = 0

    # -- low level
    def raw(self, text, kind):
        self.lines.append(text)
        self.count[kind] = self.count.get(kind, 0) + 1
        self.ws += 1

    def nop(self, n):
        while n > 0:
            k = min(n, 16)
            self.lines.append(f"s_nop {k - 1}")
            self.count["s_nop"] = self.count.get("s_nop", 0) + 1
            self.ws += k
            self.nops += k
            n -= k

    def _wait_regs(self, regs):
        need_l = None
        for i, w in enumerate(self.lds):
            if w & regs:
                need_l = len(self.lds) - 1 - i
        need_v = None
        for i, w in enumerate(self.vm):
            if w & regs:
                need_v = len(self.vm) - 1 - i
        self.wait(need_l, need_v)

    def wait(self, lgkm=None, vm=None):
        parts = []
        if vm is not None and vm < len(self.vm):
            parts.append(f"vmcnt({min(vm, 63)})")
            del self.vm[:len(self.vm) - min(vm, 63)]
        if lgkm is not None and lgkm < len(self.lds):
            parts.append(f"lgkmcnt({min(lgkm, 15)})")
            del self.lds[:len(self.lds) - min(lgkm, 15)]
        if parts:
            self.lines.append("s_waitcnt " + " ".join(parts))
            self.count["s_waitcnt"] = self.count.get("s_waitcnt", 0) + 1
            self.ws += 1

    def _mfma_result_hazard(self, regs):
        """non-MFMA access (or MFMA A/B read) of registers an MFMA wrote: pad until the result is written back"""
        t = max((self.mfma_ready.get(r, 0) for r in regs), default=0)
        if t > self.ws:
            self.nop(t - self.ws)

    # -- instruction classes
    def valu(self, text, reads, writes, kind="valu"):
        self._wait_regs(reads | writes)
        self._mfma_result_hazard(reads | writes)
        self.raw(text, kind)
        for r in writes:
            self.valu_ws[r] = self.ws

    def salu(self, text, cond=False):
        """cond: inside a scalar branch that may be skipped -- counts as no wait state for the hazard clock"""
        self.raw(text, "salu")
        if cond:
            self.ws -= 1

    def ds_read(self, text, addr, writes, kind):
        self._wait_regs(writes)              # (a register with a load still in flight is never re-targeted; this documents it)
        self._mfma_result_hazard(writes)
        self.raw(text, kind)
        self.lds.append(set(writes))

    def ds_write(self, text, reads, kind="ds_write"):
        self._wait_regs(reads)
        self._mfma_result_hazard(reads)
        self.raw(text, kind)
        self.lds.append(set())

    def vmem(self, text, reads, writes, kind):
        self._wait_regs(reads | writes)
        self._mfma_result_hazard(reads | writes)
        self.raw(text, kind)
        self.vm.append(set(writes))

    def mfma(self, big, dst, a, b, c, text):
        """dst / c: register sets (c may be empty: inline 0)"""
        self._wait_regs(dst | a | b | c)
        self._mfma_result_hazard(a | b)
        if c and c != dst:
            self._mfma_result_hazard(c)
        elif c:
            # srcC == vdst: back-to-back with the MFMA of the same shape that wrote it is supported; anything else waits for the write-back
            if any(self.mfma_chain.get(r) not in (None, big) for r in c):
                self._mfma_result_hazard(c)
        t = max((self.valu_ws.get(r, -99) for r in a | b | c), default=-99)
        if self.ws - t < self.VALU_TO_MFMA_WS:
            self.nop(self.VALU_TO_MFMA_WS - (self.ws - t))
        self.raw(text, "mfma32" if big else "mfma16")
        for r in dst:
            self.mfma_ready[r] = self.ws + (self.MFMA32_WS if big else self.MFMA16_WS)
            self.mfma_chain[r] = big

    # -- the kernel's vocabulary
    def rd128(self, dst, addr, imm):
        assert 0 <= imm <= 65535 - 15, imm
        self.ds_read(f"ds_read_b128 {vr(dst, 4)}, {vr(addr)} offset:{imm}", addr, vset(dst, 4), "ds_read_b128")

    def rdtr(self, dst, addr, imm):
        assert 0 <= imm <= 65535 - 7, imm
        self.ds_read(f"ds_read_b64_tr_b16 {vr(dst, 2)}, {vr(addr)} offset:{imm}", addr, vset(dst, 2), "ds_read_tr")

    def mfma32(self, dst, a, b, agpr=False, c=None):
        d = ar(dst, 16) if agpr else vr(dst, 16)
        dset = aset(dst, 16) if agpr else vset(dst, 16)
        self.mfma(True, dset, vset(a, 4), vset(b, 4), dset, f"v_mfma_f32_32x32x16_bf16 {d}, {vr(a, 4)}, {vr(b, 4)}, {d}")

    def mfma16(self, dst, a, b, zero):
        d = vr(dst, 4)
        self.mfma(False, vset(dst, 4), vset(a, 4), vset(b, 4), set() if zero else vset(dst, 4),
                  f"v_mfma_f32_16x16x32_bf16 {d}, {vr(a, 4)}, {vr(b, 4)}, {'0' if zero else d}")


# ---- pieces of an iteration ----------------------------------------------------------------------------------------------------
def tr_frag(e, dst, tile_imm, s2, dt):
    """lds_tr_frag(tile, lo, rowbase = 16 s2, dt): low half T[dt][0] + rowbase * 128, high half T[dt][1] + (rowbase + 8) * 128"""
    e.rdtr(dst, T + 2 * dt, tile_imm + 2048 * s2)
    e.rdtr(dst + 2, T + 2 * dt + 1, tile_imm + 2048 * s2 + 1024)


def dq_reads(e, er, ks):
    """operands of dQ k-step ks (keys 32 ks .. 32 ks + 31 of the previous pair's dS image, this wave's 16 head-dim columns of K)"""
    e.rdtr(DQB, KO, ks * 4096)
    e.rdtr(DQB + 2, KO + 1, ks * 4096)
    e.rdtr(DQA0, EO, er + ks * 2048)
    e.rdtr(DQA0 + 2, EO + 1, er + ks * 2048)
    e.rdtr(DQA1, EO + 2, er + ks * 2048)
    e.rdtr(DQA1 + 2, EO + 3, er + ks * 2048)


def dq_mma(e, first):
    e.mfma16(ACC0, DQA0, DQB, first)
    e.mfma16(ACC1, DQA1, DQB, first)


def mul_exp(e, r):
    e.valu(f"v_mul_f32_e32 {vr(S + r)}, s{SC}, {vr(S + r)}", vset(S + r), vset(S + r))


def exp(e, r):
    e.valu(f"v_exp_f32_e32 {vr(S + r)}, {vr(S + r)}", vset(S + r), vset(S + r), "exp")


def cvt(e, dst, src):
    e.valu(f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(src)}, {vr(src + 1)}", vset(src, 2), vset(dst), "cvt")


def ds_mul(e, r):
    e.valu(f"v_mul_f32_e32 {vr(DP + r)}, {vr(S + r)}, {vr(DP + r)}", vset(S + r) | vset(DP + r), vset(DP + r))


def const_s(e, st, g):
    e.rd128(S + 4 * g, CR, st + 32 * g)


def const_dp(e, st, g):
    e.rd128(DP + 4 * g, CR, st + 128 + 32 * g)


def spread(spine, fillers):
    """spine: list of callables (one matrix instruction each); fillers: list of lists (one list per gap, after spine[i])"""
    assert len(fillers) == len(spine)
    for m, fl in zip(spine, fillers):
        m()
        for f in fl:
            f()


def chunks(lst, n):
    """lst cut into n consecutive pieces, sizes as even as possible"""
    k, r = divmod(len(lst), n)
    out, i = [], 0
    for j in range(n):
        sz = k + (1 if j < r else 0)
        out.append(lst[i:i + sz])
        i += sz
    return out


def merge(*gaplists):
    """element-wise concatenation of per-gap filler lists"""
    n = len(gaplists[0])
    return [sum((g[i] for g in gaplists), []) for i in range(n)]


def advance_stage_pointers(e, tag):
    """stage pointers -> the pair after the one just requested (clamped at the last pair)"""
    L = f"{tag}"
    e.salu(f"s_cmp_eq_u32 s{SREM}, 0", cond=True)
    e.salu(f"s_cbranch_scc1 .Lsd{L}%=", cond=True)
    e.salu(f"s_sub_u32 s{SREM}, s{SREM}, 1", cond=True)
    e.salu(f"s_add_u32 s{SHS}, s{SHS}, 1", cond=True)
    e.salu(f"s_cmp_eq_u32 s{SHS}, s{SH}", cond=True)
    e.salu(f"s_cbranch_scc1 .Lsw{L}%=", cond=True)
    for p, d in ((SQ, "128"), (SDO, "128"), (SLS, f"s{SN4}"), (SDL, f"s{SN4}")):
        e.salu(f"s_add_u32 s{p}, s{p}, {d}", cond=True)
        e.salu(f"s_addc_u32 s{p + 1}, s{p + 1}, 0", cond=True)
    e.salu(f"s_branch .Lsd{L}%=", cond=True)
    e.lines.append(f".Lsw{L}%=:")
    e.salu(f"s_mov_b32 s{SHS}, 0", cond=True)
    for p, w in ((SQ, SWQ), (SDO, SWDO), (SLS, SWLS), (SDL, SWLS)):
        e.salu(f"s_add_u32 s{p}, s{p}, s{w}", cond=True)
        e.salu(f"s_addc_u32 s{p + 1}, s{p + 1}, s{w + 1}", cond=True)
    e.lines.append(f".Lsd{L}%=:")


def advance_dq_pointers(e, tag):
    L = f"{tag}"
    e.salu(f"s_mov_b64 s[{SDQP}:{SDQP + 1}], s[{SDQC}:{SDQC + 1}]")
    e.salu(f"s_add_u32 s{SHD}, s{SHD}, 1", cond=True)
    e.salu(f"s_cmp_eq_u32 s{SHD}, s{SH}", cond=True)
    e.salu(f"s_cbranch_scc1 .Lqw{L}%=", cond=True)
    e.salu(f"s_add_u32 s{SDQC}, s{SDQC}, 256", cond=True)
    e.salu(f"s_addc_u32 s{SDQC + 1}, s{SDQC + 1}, 0", cond=True)
    e.salu(f"s_branch .Lqd{L}%=", cond=True)
    e.lines.append(f".Lqw{L}%=:")
    e.salu(f"s_mov_b32 s{SHD}, 0", cond=True)
    e.salu(f"s_add_u32 s{SDQC}, s{SDQC}, s{SWDQ}", cond=True)
    e.salu(f"s_addc_u32 s{SDQC + 1}, s{SDQC + 1}, s{SWDQ + 1}", cond=True)
    e.lines.append(f".Lqd{L}%=:")


def atomics(e):
    for i in range(8):
        acc = (ACC0 if i < 4 else ACC1) + (i & 3)
        e.vmem(f"global_atomic_add_f32 {vr(AO + i)}, {vr(acc)}, s[{SDQP}:{SDQP + 1}]", vset(AO + i) | vset(acc), set(), "atomic")


def dk3_mfma(i):
    s2, dt = i >> 1, i & 1
    return lambda e: e.mfma32(DK(3, dt), TRQ + 8 * s2 + 4 * dt, DF + 4 * s2, agpr=True)


def body(e, sg, tag):
    """one iteration = one (head, 32-query block) pair; sg = parity of the pair (stage slot, dS image slot)"""
    st, stw = sg * STAGE, (1 - sg) * STAGE
    ew, er = sg * 32768, (1 - sg) * 32768
    F = lambda fn, *a: (lambda: fn(e, *a))

    # ---- head (after the barrier): the last two dK MFMAs of the previous pair cover the first operand reads of this one
    rd_s = [F(const_s, st, g) for g in range(4)]
    rd_qk = []
    for ks in range(4):
        rd_qk.append(lambda ks=ks: e.rd128(QA + 4 * ks, RS + ks, st))
        rd_qk.append(lambda ks=ks: e.rd128(KF + 4 * ks, RK + ks, 0))
    rd_dp = [F(const_dp, st, g) for g in range(4)] + [(lambda ks=ks: e.rd128(DA + 4 * ks, RS + ks, st + 4096)) for ks in range(4)]
    dq_reads(e, er, 0)                                                                      # (first: k-step 0 runs right behind the two MFMAs)
    for f in rd_s + rd_qk[:4]:
        f()
    spread([lambda: dk3_mfma(2)(e), lambda: dk3_mfma(3)(e)], [rd_qk[4:] + rd_dp[:2], rd_dp[2:]])
    # request the NEXT pair's stage (LDS-DMA, 1 KiB of Q and of dO per wave) and its row constants; older than this iteration's atomics
    e.salu(f"s_add_u32 m0, s{SM0}, {stw}")
    e.salu("s_nop 0")
    e.vmem(f"global_load_lds_dwordx4 {vr(QOFF)}, s[{SQ}:{SQ + 1}]", vset(QOFF), set(), "dma")
    e.salu(f"s_add_u32 m0, s{SM0}, {stw + 4096}")
    e.salu("s_nop 0")
    e.vmem(f"global_load_lds_dwordx4 {vr(DOOFF)}, s[{SDO}:{SDO + 1}]", vset(DOOFF), set(), "dma")
    e.vmem(f"global_load_dword {vr(CL)}, {vr(COFF)}, s[{SLS}:{SLS + 1}]", vset(COFF), vset(CL), "gload")
    e.vmem(f"global_load_dword {vr(CD)}, {vr(COFF)}, s[{SDL}:{SDL + 1}]", vset(COFF), vset(CD), "gload")
    dq_mma(e, True)                                                                         # k-step 0
    dq_reads(e, er, 1)

    # ---- S(0), dP(0) with the transposed fragments of the pair and exp2 of tile 0 beside them
    trd = [(lambda s2=s2, dt=dt: tr_frag(e, TRD + 8 * s2 + 4 * dt, st + 4096, s2, dt)) for s2 in range(2) for dt in range(2)]
    trq = [(lambda s2=s2, dt=dt: tr_frag(e, TRQ + 8 * s2 + 4 * dt, st, s2, dt)) for s2 in range(2) for dt in range(2)]
    spread([(lambda ks=ks: e.mfma32(S, QA + 4 * ks, KF + 4 * ks)) for ks in range(4)], [[trd[0]], [trd[1]], [trd[2]], [trd[3]]])
    advance_stage_pointers(e, tag)
    dq_mma(e, False)                                                                        # k-step 1
    dq_reads(e, er, 2)
    b1 = [[F(mul_exp, 4 * g + i) for i in range(4)] + [F(exp, 4 * g + i) for i in range(4)] for g in range(4)]
    b1[2] += [F(cvt, PF + j, S + 2 * j) for j in range(4)]
    spread([(lambda ks=ks: e.mfma32(DP, DA + 4 * ks, VF + 4 * ks)) for ks in range(4)], merge(b1, [[trq[0]], [trq[1]], [trq[2]], [trq[3]]]))
    dq_mma(e, False)                                                                        # k-step 2
    dq_reads(e, er, 3)
    kstep = 3

    for t in range(4):
        last = t == 3
        # ---- G1(t): dV(t) | cvt of P's second half, dS = p dP', the next tile's row constants as soon as their registers are free
        pf1 = [F(cvt, PF + 4 + j, S + 8 + 2 * j) for j in range(4)]
        muls = [F(ds_mul, r) for r in range(16)]
        cdf = [F(cvt, DF + j, DP + 2 * j) for j in range(8)]
        g1 = [pf1 + muls[0:2], muls[2:8], muls[8:14], muls[14:16] + cdf[0:4]]
        if not last:
            g1[1] += [F(const_s, st, 0)]
            g1[2] += [F(const_s, st, 1), F(const_s, st, 2)]
            g1[3] += [F(const_s, st, 3)]
            for ks in range(4):
                g1[ks] += [lambda ks=ks: e.rd128(KF + 4 * ks, RK + ks, (t + 1) * 4096)]
        spread([(lambda s2=s2, dt=dt: e.mfma32(DV(t, dt), TRD + 8 * s2 + 4 * dt, PF + 4 * s2, agpr=True)) for s2 in range(2) for dt in range(2)], g1)
        dq_mma(e, False)
        kstep += 1
        if kstep < 16:
            dq_reads(e, er, kstep)
        for f in cdf[4:8]:
            f()
        wr = [(lambda g=g: e.ds_write(f"ds_write_b64 {vr(EW + g)}, {vr(DF + 2 * g, 2)} offset:{ew + t * 2048}", vset(EW + g) | vset(DF + 2 * g, 2))) for g in range(4)]
        if last:
            break
        # ---- G2(t): S(t+1) | dS rows -> image, dP row constants
        g2 = [[wr[0], F(const_dp, st, 0)], [wr[1], F(const_dp, st, 1)], [wr[2], F(const_dp, st, 2)], [wr[3], F(const_dp, st, 3)]]
        spread([(lambda ks=ks: e.mfma32(S, QA + 4 * ks, KF + 4 * ks)) for ks in range(4)], g2)
        dq_mma(e, False)
        kstep += 1
        dq_reads(e, er, kstep)
        # ---- G3(t): dK(t) | exp2, first half of tile t+1
        g3 = [[F(mul_exp, 0), F(mul_exp, 1), F(exp, 0), F(exp, 1)], [F(mul_exp, 2), F(mul_exp, 3), F(exp, 2), F(exp, 3)],
              [F(mul_exp, 4), F(mul_exp, 5), F(exp, 4), F(exp, 5)], [F(mul_exp, 6), F(mul_exp, 7), F(exp, 6), F(exp, 7)]]
        spread([(lambda s2=s2, dt=dt: e.mfma32(DK(t, dt), TRQ + 8 * s2 + 4 * dt, DF + 4 * s2, agpr=True)) for s2 in range(2) for dt in range(2)], g3)
        dq_mma(e, False)
        kstep += 1
        dq_reads(e, er, kstep)
        # ---- G4(t): dP(t+1) | exp2, second half; first half of P -> bf16
        g4 = [[F(mul_exp, 8), F(mul_exp, 9), F(exp, 8), F(exp, 9)], [F(mul_exp, 10), F(mul_exp, 11), F(exp, 10), F(exp, 11)],
              [F(mul_exp, 12), F(mul_exp, 13), F(exp, 12), F(exp, 13), F(cvt, PF, S), F(cvt, PF + 1, S + 2)],
              [F(mul_exp, 14), F(mul_exp, 15), F(exp, 14), F(exp, 15), F(cvt, PF + 2, S + 4), F(cvt, PF + 3, S + 6)]]
        spread([(lambda ks=ks: e.mfma32(DP, DA + 4 * ks, VF + 16 * (t + 1) + 4 * ks)) for ks in range(4)], g4)
        dq_mma(e, False)
        kstep += 1
        dq_reads(e, er, kstep)
    assert kstep == 16, kstep

    # ---- end of the iteration: first two dK MFMAs of tile 3 beside the dS rows, the next stage's row constants and the atomics
    dk3_mfma(0)(e)
    for f in wr:
        f()
    e.wait(vm=0)                                                                            # the row constants requested at the head (and with
    e.valu(f"v_mul_f32_e32 {vr(TMP0)}, s{SNRC}, {vr(CL)}", vset(CL), vset(TMP0))            #  them every older operation: last iteration's atomics)
    e.valu(f"v_mul_f32_e32 {vr(TMP1)}, -1.0, {vr(CD)}", vset(CD), vset(TMP1))
    e.valu(f"v_cndmask_b32_e64 {vr(TMP0)}, {vr(TMP1)}, {vr(TMP0)}, s[{SMASK}:{SMASK + 1}]", vset(TMP0) | vset(TMP1), vset(TMP0))
    e.ds_write(f"ds_write_b32 {vr(CW)}, {vr(TMP0)} offset:{stw}", vset(CW) | vset(TMP0), "ds_write")
    dk3_mfma(1)(e)
    atomics(e)
    advance_dq_pointers(e, tag)
    e.wait(lgkm=0)
    e.salu("s_barrier")


def generate():
    e = Emitter()
    # ---- prologue: accumulators, loop state
    for i in range(256):
        e.raw(f"v_accvgpr_write_b32 a{i}, 0", "init")
    for r in list(range(DF, DF + 8)) + list(range(TRQ, TRQ + 16)):                          # the head's dK MFMAs of a pair that does not exist: 0 x 0
        e.raw(f"v_mov_b32_e32 {vr(r)}, 0", "init")
    e.salu(f"s_mov_b64 s[{SDQP}:{SDQP + 1}], s[{SDQC}:{SDQC + 1}]")
    e.salu(f"s_mov_b32 s{SHD}, 0")
    e.salu(f"s_mov_b32 s{SHS}, 1")
    e.salu(f"s_cmp_eq_u32 s{SH}, 1")
    e.salu(f"s_cselect_b32 s{SHS}, 0, s{SHS}")
    e.salu(f"s_lshl_b32 s{SREM}, s{SCNT}, 1")
    e.salu(f"s_sub_u32 s{SREM}, s{SREM}, 2")
    e.salu(f"s_mov_b32 s{SMASK}, -1")
    e.salu(f"s_mov_b32 s{SMASK + 1}, 0")
    # The loop's text must be valid for the state the back edge arrives in (MFMA results of the previous iteration still settling), which
    # needs at least the padding of the first entry: emit the two bodies from the post-iteration state until the text repeats.
    import copy

    def two_bodies(state):
        x = copy.deepcopy(state)
        x.lines, x.count, x.nops = [], {}, 0
        body(x, 0, "a")
        body(x, 1, "b")
        x.salu(f"s_sub_u32 s{SCNT}, s{SCNT}, 1")
        x.salu(f"s_cmp_lg_u32 s{SCNT}, 0")
        x.salu("s_cbranch_scc1 .Lloop%=")
        return x

    first = two_bodies(e)
    second = two_bodies(first)
    third = two_bodies(second)
    assert second.lines == third.lines, "the loop text does not reach a fixed point over the back edge"
    entry_only = [l for l in first.lines if l.startswith("s_nop")]
    steady = [l for l in second.lines if l.startswith("s_nop")]
    assert sum(int(l.split()[1]) + 1 for l in entry_only) <= sum(int(l.split()[1]) + 1 for l in steady)
    e.lines.append(".Lloop%=:")
    e.lines += second.lines
    loop_len, loop_count, chk = len(second.lines), dict(second.count), second
    for k in ("lds", "vm", "ws", "mfma_ready", "mfma_chain", "valu_ws"):
        setattr(e, k, copy.deepcopy(getattr(second, k)))
    e.nops += second.nops
    # ---- tail: the last pair's remaining dK MFMAs and its dQ
    dk3_mfma(2)(e)
    dk3_mfma(3)(e)
    dq_reads(e, 32768, 0)
    for ks in range(16):
        dq_mma(e, ks == 0)
        if ks < 15:
            dq_reads(e, 32768, ks + 1)
    atomics(e)
    e.nop(20)                                                                               # every accumulator written back ...
    for i in range(256):                                                                    # ... then handed to the C++ epilogue in v0..v255 (a 1024-bit
        e.raw(f"v_accvgpr_read_b32 v{i}, a{i}", "fini")                                     #     AGPR tuple as an asm output makes hipcc 7.2 emit an illegal copy)
    return e, loop_len, loop_count, chk


def main():
    e, loop_len, loop_count, chk = generate()
    text = e.lines
    with open(OUT, "w") as f:
        f.write("// GENERATED by tools/gen_attn_bwd512.py -- do not edit.  Main loop of mqa_bwd_fused512a_kernel (csrc/attn.hip) as one asm statement.\n")
        f.write("// register map: see the generator; the C++ side binds its operands to the same physical registers.\n")
        f.write("#define OSUF_BWD512A_ASM \\\n")
        for l in text:
            f.write(f'  "{l}\\n\\t" \\\n')
        f.write('  ""\n')
        cl = ", ".join(f'"v{i}"' for i in range(64, 192)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
        sc = ", ".join(f'"s{i}"' for i in (SDQP, SDQP + 1, SHS, SREM, SHD, STMP, SMASK, SMASK + 1))
        f.write(f'#define OSUF_BWD512A_CLOBBERS "memory", "vcc", "scc", {sc}, {cl}\n')
    if "--stats" in sys.argv:
        per_pair = {k: v / 2 for k, v in loop_count.items() if k not in ("init", "fini")}
        tot = sum(v for k, v in per_pair.items() if not k.startswith("mfma"))
        print("per pair:", {k: per_pair[k] for k in sorted(per_pair)})
        print(f"non-MFMA instructions per pair: {tot:.0f}; s_nop wait states inserted in total: {e.nops}")
    print("wrote", OUT, len(text), "lines")


if __name__ == "__main__":
    main()
