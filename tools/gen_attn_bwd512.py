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

  * the list-scheduled body is checked against the hand-ordered one (--manual; pinned on the GPU bit for bit against the compiled kernel) by
    symbolic execution: same value in every register after four iterations, same LDS stores / atomics / DMA requests.

Output: osufusion_amd/csrc/attn_bwd512_asm.inc (macros OSUF_BWD512A_ASM / _CLOBBERS; the C++ side binds its operands to the physical registers of
the map below).   usage: python tools/gen_attn_bwd512.py [--stats] [--gaps] [--manual]
Triage / tuning knobs (timing-only or A/B builds through tools/build_variants.sh): --drop valu,atomics,dq,dqreads,lds,barrier,dma,vmwait,atomstore
--pad N --padlds --e64 --pkc --pkds --lat N --ldscap B,S --budget B,S --look N --ksp N --cwlate --out FILE
"""
from __future__ import annotations

import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
# --qs: the variant for queries that arrive PRE-SCALED by c = scale log2 e (osuf_rope_cast_qs folds the softmax scale into the bf16 rounding of
# the rotated q): S' = Qs K^T - lse2 leaves the MFMA chain already in the log2 domain, so the 64 `v_mul_f32 S, c, S` of a pair are not emitted
# (the row constant is -lse2: the C++ side passes -1 for -1/c).  Second output file / macro pair, same register map.
QS = "--qs" in sys.argv
OUT = ROOT / "osufusion_amd" / "csrc" / ("attn_bwd512qs_asm.inc" if QS else "attn_bwd512_asm.inc")
# timing-only triage builds (wrong results, never shipped): --drop valu,atomics,dq,lds,barrier,dma  --out <file>
DROP = set()
for i, a in enumerate(sys.argv):
    if a == "--drop":
        DROP = set(sys.argv[i + 1].split(","))
    if a == "--out":
        OUT = Path(sys.argv[i + 1])

# ---- register map (arch VGPRs) -----------------------------------------------------------------------------------------------
VF, QA, DA, TRD, TRQ, S, DP, PF, DF, KF = 0, 64, 80, 96, 112, 128, 144, 160, 168, 176
DQB, DQA0, DQA1, ACC0, ACC1 = 192, 196, 200, 204, 208
CS = 212                          # -lse / c of the pair's 32 queries as this lane's accumulator rows: the C operand of every S chain's first MFMA
AO, CL, CD = 228, 230, 231        # byte offsets of the atomics of the two query halves (rows: scalar bases); row constants of the next stage in flight
RK, T, KO, EO, EW, CR, CW, QOFF, DOOFF, COFF, RTMP = 232, 236, 240, 242, 246, 250, 251, 252, 253, 254, 255
# AGPRs: dK^T tile t, head-dim half dt at a[32 t + 16 dt ..+15]; dV^T at a[128 + 32 t + 16 dt ..]
def DK(t, dt): return 32 * t + 16 * dt
def DV(t, dt): return 128 + 32 * t + 16 * dt
# SGPRs
SQ, SDO, SLS, SDL, SDQC = 48, 50, 52, 54, 56                   # 64-bit bases (pairs): Q / dO / lse2 / delta of the part's SECOND pair, dQ rows of its first
SCNT, SH, SHS, SREM, SHD, SN4 = 60, 61, 62, 63, 64, 65          # loop trips, heads, heads left (stage), blocks left (stage), heads left (dQ), 4 N
SWQ, SWDO, SWLS, SWDQ = 66, 68, 70, 72                         # wrap deltas (bytes, low words used) at the last head of a query block
SD128, SDN4, SDQS = 58, 59, 90                                 # per-head steps of QOFF / DOOFF, of COFF, of AO (0 once past the last pair / in the first iteration)
SC, SNRC, SM0, SKOF, SMASK = 74, 75, 76, 77, 78                  # c = scale log2 e, -1/c, LDS byte address of this wave's 1-KiB DMA piece, RK - RS, lane mask (pair)
SROWB, SROW = 80, 82                                             # bytes per dQ row; s[82:89]: scalar bases of dQ rows 0..3 of the part's first pair
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


# 4-byte encodings (VOP1 / VOP2 _e32) leave the 8-byte instructions behind them (MFMA, DS, VOP3) on addresses = 4 mod 8
E64 = "--e64" in sys.argv


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
        # symbolic dataflow (which value sits in which register), compared between the hand-ordered body and the list-scheduled one
        self.val = {}            # reg -> value id
        self.look = set()        # registers the next few matrix instructions read (wait merging)
        self.cold = []           # rarely taken scalar-branch targets, emitted behind the kernel's last instruction
        self.sinks = []          # LDS stores / atomics: (kind, ids ...)
        self.epoch = 0           # barriers passed (LDS contents differ between iterations)
        self.pepoch = [0, 0]     # advances of the stage pointers / of the dQ pointers so far

    def vid(self, r):
        return self.val.get(r, ("init", r))

    def define(self, op, dsts, srcs, extra=()):
        import hashlib
        key = repr((op, tuple(self.vid(r) for r in sorted(srcs)), extra))
        for j, r in enumerate(sorted(dsts)):
            self.val[r] = hashlib.md5((key + str(j)).encode()).hexdigest()[:16]

    def sink(self, kind, srcs, extra=()):
        self.sinks.append((kind, tuple(self.vid(r) for r in sorted(srcs)), extra))

    # -- low level
    def raw(self, text, kind):
        if E64 and text.startswith("v_") and "_e32 " in text:    # every vector instruction 8 bytes long (see E64)
            text = text.replace("_e32 ", "_e64 ", 1)
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
        if self.look and (any(w & regs for w in self.lds)):
            regs = regs | self.look                               # a wait is due anyway: let it cover what the next matrix instructions read
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
        if "valu" in DROP:
            return
        self._wait_regs(reads | writes)
        self._mfma_result_hazard(reads | writes)
        self.raw(text, kind)
        self.define(text.split()[0] + "|" + "|".join(t for t in text.replace(",", " ").split()[1:] if not t.startswith("v")), writes, reads)
        for r in writes:
            self.valu_ws[r] = self.ws

    def salu(self, text, cond=False):
        """cond: inside a scalar branch that may be skipped -- counts as no wait state for the hazard clock"""
        if "barrier" in DROP and text == "s_barrier":
            return
        self.raw(text, "salu")
        if text == "s_barrier":
            self.epoch += 1
        if cond:
            self.ws -= 1

    def ds_read(self, text, addr, writes, kind):
        if "lds" in DROP:
            return
        self._wait_regs(writes)              # (a register with a load still in flight is never re-targeted; this documents it)
        self._mfma_result_hazard(writes)
        self.raw(text, kind)
        # (the K image is written once, before the loop: its reads mean the same in every iteration)
        self.define(kind + "|" + text.split("offset:")[1], writes, vset(addr), () if addr in (RK, RK + 1, RK + 2, RK + 3, KO, KO + 1) else (self.epoch,))
        self.lds.append(set(writes))

    def ds_write(self, text, reads, kind="ds_write"):
        if "lds" in DROP:
            return
        self._wait_regs(reads)
        self._mfma_result_hazard(reads)
        self.raw(text, kind)
        self.sink(text.split()[0] + "|" + text.split("offset:")[1], reads, (self.epoch,))
        self.lds.append(set())

    def vmem(self, text, reads, writes, kind):
        if (kind == "atomic" and "atomics" in DROP) or (kind in ("dma", "gload") and "dma" in DROP):
            return
        self._wait_regs(reads | writes)
        self._mfma_result_hazard(reads | writes)
        self.raw(text, kind)
        sbase = text.split("s[")[1].split("]")[0]
        if writes:
            self.define(kind + "|" + sbase, writes, reads, (self.epoch, tuple(self.pepoch)))
        else:
            self.sink(kind + "|" + sbase, reads, (self.epoch, tuple(self.pepoch), self.lines[-3] if kind == "dma" else ""))
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
        # (A and B fragments are register quadruples at distinct bases: tag each source with its role)
        import hashlib
        key = repr(("mfma", big, tuple(self.vid(r) for r in sorted(a)), tuple(self.vid(r) for r in sorted(b))))
        csrc = sorted(c) if c else []
        for j, r in enumerate(sorted(dst)):
            cj = self.vid(csrc[j]) if c else 0
            self.val[r] = hashlib.md5((key + repr(cj) + str(j)).encode()).hexdigest()[:16]
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
        cs = dset if c is None else vset(c, 16)
        self.mfma(True, dset, vset(a, 4), vset(b, 4), cs, f"v_mfma_f32_32x32x16_bf16 {d}, {vr(a, 4)}, {vr(b, 4)}, {d if c is None else vr(c, 16)}")

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
    if "dq" in DROP:
        return
    e.rdtr(DQB, KO, ks * 4096)
    e.rdtr(DQB + 2, KO + 1, ks * 4096)
    e.rdtr(DQA0, EO, er + ks * 2048)
    e.rdtr(DQA0 + 2, EO + 1, er + ks * 2048)
    e.rdtr(DQA1, EO + 2, er + ks * 2048)
    e.rdtr(DQA1 + 2, EO + 3, er + ks * 2048)


def dq_mma(e, first):
    if "dq" in DROP:
        return
    e.mfma16(ACC0, DQA0, DQB, first)
    e.mfma16(ACC1, DQA1, DQB, first)


PKC, PKDS = "--pkc" in sys.argv, "--pkds" in sys.argv          # packed forms of the two elementwise products (two per instruction; same fp32 results)
SC2 = 92                                                         # s[92:93] = (c, c) for the packed form


def mul_exp(e, r):
    if QS:                                                       # pre-scaled queries: the chain's result is the exponent itself
        return
    if PKC:
        if r % 2 == 0:
            e.valu(f"v_pk_mul_f32 {vr(S + r, 2)}, {vr(S + r, 2)}, s[{SC2}:{SC2 + 1}]", vset(S + r, 2), vset(S + r, 2))
        return
    e.valu(f"v_mul_f32_e32 {vr(S + r)}, s{SC}, {vr(S + r)}", vset(S + r), vset(S + r))


def exp(e, r):
    e.valu(f"v_exp_f32_e32 {vr(S + r)}, {vr(S + r)}", vset(S + r), vset(S + r), "exp")


def cvt(e, dst, src):
    e.valu(f"v_cvt_pk_bf16_f32 {vr(dst)}, {vr(src)}, {vr(src + 1)}", vset(src, 2), vset(dst), "cvt")


def ds_mul(e, r):
    if PKDS:
        if r % 2 == 0:
            e.valu(f"v_pk_mul_f32 {vr(DP + r, 2)}, {vr(S + r, 2)}, {vr(DP + r, 2)}", vset(S + r, 2) | vset(DP + r, 2), vset(DP + r, 2))
        return
    e.valu(f"v_mul_f32_e32 {vr(DP + r)}, {vr(S + r)}, {vr(DP + r)}", vset(S + r) | vset(DP + r), vset(DP + r))


def const_s(e, st, g):
    e.rd128(CS + 4 * g, CR, st + 32 * g)


def rs_read(e, dst, ks, imm):
    """row fragment ks of a stage tile: the lane offset is RK[ks] less the K rows' base (a register pair saved per fragment)"""
    e.valu(f"v_subrev_u32_e32 {vr(RTMP)}, s{SKOF}, {vr(RK + ks)}", vset(RK + ks), vset(RTMP))
    e.rd128(dst, RTMP, imm)


def s_mfma(e, ks):
    e.mfma32(S, QA + 4 * ks, KF + 4 * ks, c=CS if ks == 0 else None)


def const_dp(e, st, g):
    e.rd128(DP + 4 * g, CR, st + 128 + 32 * g)


def spread(spine, fillers):
    """spine: list of callables (one matrix instruction each); fillers: list of lists (one list per gap, after spine[i])"""
    assert len(fillers) == len(spine)
    for m, fl in zip(spine, fillers):
        m()
        for f in fl:
            f()


def merge(*gaplists):
    """element-wise concatenation of per-gap filler lists"""
    n = len(gaplists[0])
    return [sum((g[i] for g in gaplists), []) for i in range(n)]


def advance_stage_pointers(e, tag):
    """The request offsets (QOFF, DOOFF: this lane's bytes of the DMA pieces; COFF: its row constant) move on to the pair after the one just
    requested: +1 head, or at a block's last head on to the next query block; past the last pair they stop (deltas 0)."""
    e.pepoch[0] += 1
    L = f"{tag}"
    e.salu(f"s_sub_u32 s{SHS}, s{SHS}, 1")                         # heads left in the block; borrow -> next block
    e.salu(f"s_cbranch_scc1 .Lsw{L}%=")
    e.valu(f"v_add_u32_e32 {vr(QOFF)}, s{SD128}, {vr(QOFF)}", vset(QOFF), vset(QOFF))
    e.valu(f"v_add_u32_e32 {vr(DOOFF)}, s{SD128}, {vr(DOOFF)}", vset(DOOFF), vset(DOOFF))
    e.valu(f"v_add_u32_e32 {vr(COFF)}, s{SDN4}, {vr(COFF)}", vset(COFF), vset(COFF))
    e.cold.append([f".Lsw{L}%=:",
                   f"s_sub_u32 s{SREM}, s{SREM}, 1",               # blocks left; borrow -> that was the last pair
                   f"s_cbranch_scc1 .Lse{L}%=",
                   f"s_sub_u32 s{SHS}, s{SH}, 1",
                   f"v_add_u32_e32 {vr(QOFF)}, s{SWQ}, {vr(QOFF)}",
                   f"v_add_u32_e32 {vr(DOOFF)}, s{SWDO}, {vr(DOOFF)}",
                   f"v_add_u32_e32 {vr(COFF)}, s{SWLS}, {vr(COFF)}",
                   f"s_branch .Lsd{L}%=",
                   f".Lse{L}%=:",
                   f"s_mov_b32 s{SD128}, 0",
                   f"s_mov_b32 s{SDN4}, 0",
                   f"s_mov_b32 s{SHS}, 0x7fffffff",
                   f"s_branch .Lsd{L}%="])
    e.lines.append(f".Lsd{L}%=:")


def advance_dq_pointers(e, tag):
    """AO (this lane's byte offsets of the atomics) moves on to the pair whose dQ the next iteration finishes; in the very first iteration the
    step is 0 (its atomics added the zero image to the first pair's rows, which the second iteration adds to for real)."""
    L = f"{tag}"
    e.pepoch[1] += 1
    e.salu(f"s_sub_u32 s{SHD}, s{SHD}, 1")
    e.salu(f"s_cbranch_scc1 .Lqw{L}%=")
    e.valu(f"v_add_u32_e32 {vr(AO)}, s{SDQS}, {vr(AO)}", vset(AO), vset(AO))
    e.valu(f"v_add_u32_e32 {vr(AO + 1)}, s{SDQS}, {vr(AO + 1)}", vset(AO + 1), vset(AO + 1))
    e.salu(f"s_movk_i32 s{SDQS}, 256")
    e.cold.append([f".Lqw{L}%=:",
                   f"s_sub_u32 s{SHD}, s{SH}, 1",
                   f"v_add_u32_e32 {vr(AO)}, s{SWDQ}, {vr(AO)}",
                   f"v_add_u32_e32 {vr(AO + 1)}, s{SWDQ}, {vr(AO + 1)}",
                   f"s_branch .Lqd{L}%="])
    e.lines.append(f".Lqd{L}%=:")


def atomic(e, i):
    """row i & 3 of query half i >> 2"""
    acc = (ACC0 if i < 4 else ACC1) + (i & 3)
    base = SROW + 2 * (i & 3)
    if "atomstore" in DROP:                                       # triage: plain stores of the same shape instead of the float atomics
        e.vmem(f"global_store_dword {vr(AO + (i >> 2))}, {vr(acc)}, s[{base}:{base + 1}]", vset(AO + (i >> 2)) | vset(acc), set(), "atomic")
        return
    e.vmem(f"global_atomic_add_f32 {vr(AO + (i >> 2))}, {vr(acc)}, s[{base}:{base + 1}]", vset(AO + (i >> 2)) | vset(acc), set(), "atomic")


def atomics(e):
    for i in range(8):
        atomic(e, i)


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
        rd_qk.append(lambda ks=ks: rs_read(e, QA + 4 * ks, ks, st))
        rd_qk.append(lambda ks=ks: e.rd128(KF + 4 * ks, RK + ks, 0))
    rd_dp = [F(const_dp, st, g) for g in range(4)] + [(lambda ks=ks: rs_read(e, DA + 4 * ks, ks, st + 4096)) for ks in range(4)]
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
    spread([(lambda ks=ks: s_mfma(e, ks)) for ks in range(4)], [[trd[0]], [trd[1]], [trd[2]], [trd[3]]])
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
        spread([(lambda ks=ks: s_mfma(e, ks)) for ks in range(4)], g2)
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
    e.valu(f"v_mul_f32_e32 {vr(CL)}, s{SNRC}, {vr(CL)}", vset(CL), vset(CL))                #  them every older operation: last iteration's atomics)
    e.valu(f"v_mul_f32_e32 {vr(CD)}, -1.0, {vr(CD)}", vset(CD), vset(CD))
    e.valu(f"v_cndmask_b32_e64 {vr(CL)}, {vr(CD)}, {vr(CL)}, s[{SMASK}:{SMASK + 1}]", vset(CL) | vset(CD), vset(CL))
    e.ds_write(f"ds_write_b32 {vr(CW)}, {vr(CL)} offset:{stw}", vset(CW) | vset(CL), "ds_write")
    dk3_mfma(1)(e)
    atomics(e)
    advance_dq_pointers(e, tag)
    e.wait(lgkm=0)
    e.salu("s_barrier")


class Item:
    __slots__ = ("name", "fn", "cost", "after", "before", "deps", "order", "cls")

    def __init__(self, name, fn, cost, cls, after=-1, before=10 ** 6, deps=()):
        self.name, self.fn, self.cost, self.cls, self.after, self.before, self.deps = name, fn, cost, cls, after, before, tuple(deps)


# issue-cost model of the placement (cycles): an MFMA holds the wave's issue for 8 of its 32 (16x16x32: 8 of 16) cycles, so a gap hides
# 24 (8) cycles of other instructions (MI355X_MICROARCH.md, "Per-instruction cycle constants"); LDS instructions measured at ~4.8 (SQ counters).
# The four waves of a workgroup run this one stream in near lock-step (one barrier per pair), so a burst of LDS instructions in the stream is a
# 4-wave burst at the one LDS: SQ_WAIT_INST_LDS was 14 % of the wave cycles with six-read bursts -> at most LDS_CAP LDS instructions per gap.
C_V, C_EXP, C_L, C_W = 4, 8, 5, 6
BUDGET_BIG, BUDGET_SMALL = 30, 10
LDS_CAP_BIG, LDS_CAP_SMALL = 3, 1
LAT = 3                                                           # an LDS read sits at least this many slots ahead of its consumer
KSP = 4                                                           # 32x32x16 MFMAs per dQ k-step
CWLATE = "--cwlate" in sys.argv
for _i, _a in enumerate(sys.argv):
    if _a == "--ksp":
        KSP = int(sys.argv[_i + 1])                                # tuning knobs of the placement (tools/build_variants.sh name=@"--lat 5 --ldscap 4,2")
    if _a == "--lat":
        LAT = int(sys.argv[_i + 1])
    if _a == "--ldscap":
        LDS_CAP_BIG, LDS_CAP_SMALL = map(int, sys.argv[_i + 1].split(","))
    if _a == "--budget":
        BUDGET_BIG, BUDGET_SMALL = map(int, sys.argv[_i + 1].split(","))
GAPLOG = []


def body_sched(e, sg, tag):
    """One iteration (pair parity sg) as a spine of matrix instructions with every other instruction assigned to a gap by a list scheduler:
    earliest / latest gap and producer items per instruction, filled towards ~30 (10) cycles of issue per 32x32x16 (16x16x32) gap."""
    st, stw = sg * STAGE, (1 - sg) * STAGE
    ew, er = sg * 32768, (1 - sg) * 32768
    slots = []                                                    # (name, emit function, gap budget)

    sregs = []                                                    # per slot: the registers its matrix instruction reads out of LDS loads

    bigs = []

    def big(name, fn, regs=frozenset()):
        bigs.append((name, fn, set(regs)))

    def kstep(j):
        slots.append((f"Q{j}a", lambda: None if "dq" in DROP else e.mfma16(ACC0, DQA0, DQB, j == 0), BUDGET_SMALL))
        sregs.append(vset(DQA0, 4) | vset(DQB, 4))
        slots.append((f"Q{j}b", lambda: None if "dq" in DROP else e.mfma16(ACC1, DQA1, DQB, j == 0), BUDGET_SMALL))
        sregs.append(vset(DQA1, 4) | vset(DQB, 4))

    def group_s(t):
        for ks in range(4):
            big(f"S{t}.{ks}", lambda ks=ks: s_mfma(e, ks), vset(QA + 4 * ks, 4) | vset(KF + 4 * ks, 4) | (vset(CS, 16) if ks == 0 else set()))

    def group_p(t):
        for ks in range(4):
            big(f"P{t}.{ks}", lambda ks=ks: e.mfma32(DP, DA + 4 * ks, VF + 16 * t + 4 * ks), vset(DA + 4 * ks, 4) | (vset(DP, 16) if ks == 0 else set()))

    def group_v(t):
        for i in range(4):
            s2, dt = i >> 1, i & 1
            big(f"V{t}.{i}", lambda s2=s2, dt=dt: e.mfma32(DV(t, dt), TRD + 8 * s2 + 4 * dt, PF + 4 * s2, agpr=True), vset(TRD + 8 * s2 + 4 * dt, 4))

    def group_k(t):
        for i in range(4):
            s2, dt = i >> 1, i & 1
            big(f"K{t}.{i}", lambda s2=s2, dt=dt: e.mfma32(DK(t, dt), TRQ + 8 * s2 + 4 * dt, DF + 4 * s2, agpr=True), vset(TRQ + 8 * s2 + 4 * dt, 4))

    big("K3c", lambda: dk3_mfma(2)(e))
    big("K3d", lambda: dk3_mfma(3)(e))
    group_s(0)
    group_p(0)
    for t in range(4):
        group_v(t)
        if t < 3:
            group_s(t + 1)
            group_k(t)
            group_p(t + 1)
    big("K3a", lambda: dk3_mfma(0)(e))
    big("K3b", lambda: dk3_mfma(1)(e))
    # one dQ k-step (two 16x16x32) behind every KSP-th 32x32x16: KSP = 4 spreads the sixteen k-steps over the whole iteration, KSP = 3 ends them
    # after 48 of its 66 matrix instructions, so that the eight float atomics of the finished tiles can leave one at a time behind the rest
    kpos = {2 + KSP * q for q in range(16)}
    j = 0
    for n, (name, fn, regs) in enumerate(bigs, 1):
        slots.append((name, fn, BUDGET_BIG))
        sregs.append(regs)
        if n in kpos:
            kstep(j); j += 1
    assert j == 16, j
    idx = {n: i for i, (n, _, _) in enumerate(slots)}
    items = []

    def add(name, fn, cost, cls, after=-1, before=10 ** 6, deps=()):
        items.append(Item(name, fn, cost, cls, after, before, deps))

    def add_tr(name, dst, tile_imm, s2, dt, **kw):
        """the two transposed reads of lds_tr_frag(tile, rowbase = 16 s2, dt)"""
        add(name + "l", (lambda: e.rdtr(dst, T + 2 * dt, tile_imm + 2048 * s2)), C_L, "L", **kw)
        add(name + "h", (lambda: e.rdtr(dst + 2, T + 2 * dt + 1, tile_imm + 2048 * s2 + 1024)), C_L, "L", **kw)

    def add_dq(jq, after, before_a, before_b):
        if "dqreads" in DROP:                                     # triage: the dQ MFMAs on stale operands
            return
        for i, (dst, adr, imm) in enumerate(((DQB, KO, jq * 4096), (DQA0, EO, er + jq * 2048), (DQA1, EO + 2, er + jq * 2048))):
            for h in range(2):
                add(f"dqr{jq}.{i}{h}", (lambda dst=dst, adr=adr, imm=imm, h=h: e.rdtr(dst + 2 * h, adr + h, imm)), C_L, "L",
                    after=after[1] if i == 2 else after[0], before=before_b if i == 2 else before_a)

    # ---- head: operands of the pair (after the barrier)
    if "dq" not in DROP:
        add_dq(0, (-1, -1), idx["Q0a"], idx["Q0b"])
    for g in range(4):
        add(f"cs0.{g}", (lambda g=g: const_s(e, st, g)), C_L, "L", before=idx["S0.0"] - LAT)
    for ks in range(4):
        add(f"qa.{ks}", (lambda ks=ks: rs_read(e, QA + 4 * ks, ks, st)), C_L + C_V, "L", before=idx[f"S0.{ks}"] - LAT)
    for g in range(4):
        add(f"cd0.{g}", (lambda g=g: const_dp(e, st, g)), C_L, "L", before=idx["P0.0"] - LAT)
    for ks in range(4):
        add(f"da.{ks}", (lambda ks=ks: rs_read(e, DA + 4 * ks, ks, st + 4096)), C_L + C_V, "L", before=idx[f"P0.{ks}"] - LAT)
    for i in range(4):
        s2, dt = i >> 1, i & 1
        add_tr(f"trd.{i}", TRD + 8 * s2 + 4 * dt, st + 4096, s2, dt, before=idx[f"V0.{i}"] - LAT)
        add_tr(f"trq.{i}", TRQ + 8 * s2 + 4 * dt, st, s2, dt, after=idx["K3d"], before=idx[f"K0.{i}"] - LAT)

    def dma():
        e.salu(f"s_add_u32 m0, s{SM0}, {stw}")
        e.salu("s_nop 0")
        e.vmem(f"global_load_lds_dwordx4 {vr(QOFF)}, s[{SQ}:{SQ + 1}]", vset(QOFF), set(), "dma")
        e.salu(f"s_add_u32 m0, s{SM0}, {stw + 4096}")
        e.salu("s_nop 0")
        e.vmem(f"global_load_lds_dwordx4 {vr(DOOFF)}, s[{SDO}:{SDO + 1}]", vset(DOOFF), set(), "dma")
        e.vmem(f"global_load_dword {vr(CL)}, {vr(COFF)}, s[{SLS}:{SLS + 1}]", vset(COFF), vset(CL), "gload")
        e.vmem(f"global_load_dword {vr(CD)}, {vr(COFF)}, s[{SDL}:{SDL + 1}]", vset(COFF), vset(CD), "gload")
    add("dma", dma, 36, "O", after=idx["S0.1"], before=idx["P0.2"])
    add("adv", (lambda: advance_stage_pointers(e, tag)), 12, "O", deps=("dma",), before=idx["V0.2"])
    # ---- per key tile
    for t in range(4):
        s_done, p_done = idx[f"S{t}.3"] + 3, idx[f"P{t}.3"] + 3     # first gap in which a VALU reader of the chain's result may sit (the emitter pads to 12 wait states)
        vdead = lambda r: idx[f"V{t}.0"] if r < 8 else idx[f"V{t}.2"]
        snext = idx[f"S{t + 1}.0"] if t < 3 else 10 ** 6          # readers of S sit before the next tile's S chain
        for jj in range(8):
            add(f"mm{t}.{jj}", (lambda jj=jj: (mul_exp(e, 2 * jj), mul_exp(e, 2 * jj + 1))), 0 if QS else 2 * C_V, "V", after=s_done, before=vdead(2 * jj))
            add(f"ee{t}.{jj}", (lambda jj=jj: (exp(e, 2 * jj), exp(e, 2 * jj + 1))), 2 * C_EXP, "V", deps=(f"mm{t}.{jj}",), before=vdead(2 * jj))
        for q in range(4):
            add(f"pf{t}.{q}", (lambda q=q: (cvt(e, PF + 2 * q, S + 4 * q), cvt(e, PF + 2 * q + 1, S + 4 * q + 2))), 2 * C_V, "V",
                deps=(f"ee{t}.{2 * q}", f"ee{t}.{2 * q + 1}"), before=vdead(4 * q))
        kname = (lambda i: f"K{t}.{i}") if t < 3 else (lambda i: "K3a" if i == 0 else "K3b")
        kdead = lambda g: idx[kname(0)] if g < 2 else (idx[kname(2)] if t < 3 else idx["K3b"])
        for g in range(4):
            for h in range(2):
                add(f"dm{t}.{g}{h}", (lambda g=g, h=h: (ds_mul(e, 4 * g + 2 * h), ds_mul(e, 4 * g + 2 * h + 1))), 2 * C_V, "V", after=p_done,
                    deps=(f"ee{t}.{2 * g + h}",), before=min(kdead(g), snext))
            dfd = [f"dm{t}.{g}0", f"dm{t}.{g}1"] + ([f"dw{t - 1}.{g}"] if t > 0 else [])
            add(f"df{t}.{g}", (lambda g=g: [cvt(e, DF + 2 * g + q, DP + 4 * g + 2 * q) for q in range(2)]), 2 * C_V, "V", deps=dfd, before=kdead(g))
            add(f"dw{t}.{g}", (lambda g=g, t=t: e.ds_write(f"ds_write_b64 {vr(EW + g)}, {vr(DF + 2 * g, 2)} offset:{ew + t * 2048}", vset(EW + g) | vset(DF + 2 * g, 2))),
                C_W, "L", deps=(f"df{t}.{g}",))
            if t < 3:
                add(f"cd{t + 1}.{g}", (lambda g=g: const_dp(e, st, g)), C_L, "L", deps=(f"df{t}.{g}",), before=idx[f"P{t + 1}.0"] - LAT)
        for ks in range(4):                                       # K rows of the next tile (of tile 0 again for the next pair: the image is resident)
            tn = (t + 1) & 3
            add(f"kf{t + 1}.{ks}", (lambda ks=ks, tn=tn: e.rd128(KF + 4 * ks, RK + ks, tn * 4096)), C_L, "L", after=idx[f"S{t}.{ks}"],
                before=(idx[f"S{t + 1}.{ks}"] - LAT) if t < 3 else 10 ** 6)
    # ---- dQ k-steps 1..15: operands are re-read into the same registers right behind the MFMAs that consumed them
    if "dq" not in DROP:
        for jq in range(1, 16):
            add_dq(jq, (idx[f"Q{jq - 1}b"], idx[f"Q{jq - 1}b"]), idx[f"Q{jq}a"] - 1, idx[f"Q{jq}a"] - 1)
    # ---- end of the iteration

    def cw():
        if "vmwait" in DROP:                                                                    # triage: the wait alone (stale row constants)
            e.vm.clear()
        e.wait(vm=0)                                                                            # the row constants requested at the head (and with them
        e.valu(f"v_mul_f32_e32 {vr(CL)}, s{SNRC}, {vr(CL)}", vset(CL), vset(CL))                #  every older operation: last iteration's atomics)
        e.valu(f"v_mul_f32_e32 {vr(CD)}, -1.0, {vr(CD)}", vset(CD), vset(CD))
        e.valu(f"v_cndmask_b32_e64 {vr(CL)}, {vr(CD)}, {vr(CL)}, s[{SMASK}:{SMASK + 1}]", vset(CL) | vset(CD), vset(CL))
        e.ds_write(f"ds_write_b32 {vr(CW)}, {vr(CL)} offset:{stw}", vset(CW) | vset(CL), "ds_write")
    q15 = idx["Q15b"]
    add("cw", cw, 22, "O", after=(len(slots) - 3) if CWLATE else max(q15 - 4, idx["P0.3"]), before=q15 + 2 if not CWLATE else 10 ** 6)
    nleft = len(slots) - 1 - q15                                  # gaps behind the last k-step: the atomics leave one per gap where there are enough
    for i in range(8):
        add(f"atom.{i}", (lambda i=i: atomic(e, i)), 5, "O", after=q15 + 1 + (i * max(nleft - 2, 0)) // 8, deps=("cw",))
    add("advq", (lambda: advance_dq_pointers(e, tag)), 8, "O", deps=[f"atom.{i}" for i in range(8)])
    for n, it in enumerate(items):
        it.order = n
    byname = {it.name: it for it in items}
    for _ in range(6):                                            # a producer is due no later than its consumers
        for it in reversed(items):
            for d in it.deps:
                byname[d].before = min(byname[d].before, it.before)

    placed = set()
    pending = list(items)

    def fill(k, budget):
        used, nlds, last = 0, 0, None
        cap = LDS_CAP_BIG if budget == BUDGET_BIG else LDS_CAP_SMALL
        while True:
            cands = [it for it in pending if it.after <= k and all(d in placed for d in it.deps)]
            if not cands:
                break
            cands.sort(key=lambda it: (it.before, it.order))
            pick = None
            if cands[0].before <= k + 1:
                pick = cands[0]                                   # due now: placed whatever it costs
            else:
                fit = [c for c in cands if used + c.cost <= budget and not (c.cls == "L" and nlds >= cap)]
                alt = [c for c in fit if c.cls != last]           # alternate LDS and vector instructions where the deadlines allow
                if alt and alt[0].before <= fit[0].before + 12:
                    pick = alt[0]
                elif fit:
                    pick = fit[0]
            if pick is None:
                break
            pick.fn()
            placed.add(pick.name)
            pending.remove(pick)
            used += pick.cost
            nlds += pick.cls == "L"
            last = pick.cls
        return used

    pad = int(sys.argv[sys.argv.index("--pad") + 1]) if "--pad" in sys.argv else 0     # triage: extra vector instructions per 32x32x16 gap
    LOOK = int(sys.argv[sys.argv.index("--look") + 1]) if "--look" in sys.argv else 2
    for k, (name, fn, budget) in enumerate(slots):
        e.look = set().union(*sregs[k + 1:k + 1 + LOOK]) if LOOK else set()
        fn()
        if budget == BUDGET_BIG:
            for _ in range(pad):
                e.raw(f"v_mov_b32_e32 {vr(RTMP)}, {vr(RTMP)}", "pad")
            if "--padlds" in sys.argv:                            # triage: what an LDS instruction costs the issue (no data moved)
                e.raw("ds_nop", "pad")
                e.lds.append(set())
        u = fill(k, budget)
        GAPLOG.append((tag, name, budget, u))
    while pending:                                                # (what has no deadline inside the iteration: the last dS rows)
        ready = [it for it in pending if all(d in placed for d in it.deps)]
        assert ready, [it.name for it in pending]
        ready[0].fn(); placed.add(ready[0].name); pending.remove(ready[0])
        GAPLOG.append((tag, "flush:" + ready[0].name, 0, ready[0].cost))
    e.look = set()
    e.wait(lgkm=0)
    e.salu("s_barrier")


BODY = body if "--manual" in sys.argv else body_sched


def generate():
    e = Emitter()
    # ---- prologue: accumulators, loop state
    for i in range(256):
        e.raw(f"v_accvgpr_write_b32 a{i}, 0", "init")
    for r in list(range(DF, DF + 8)) + list(range(TRQ, TRQ + 16)):                          # the head's dK MFMAs of a pair that does not exist: 0 x 0
        e.raw(f"v_mov_b32_e32 {vr(r)}, 0", "init")
    for ks in range(4):                                                                     # K rows of tile 0 (each iteration re-reads them for the next pair)
        e.rd128(KF + 4 * ks, RK + ks, 0)
    e.salu(f"s_mov_b64 s[{SROW}:{SROW + 1}], s[{SDQC}:{SDQC + 1}]")                          # scalar bases of the four dQ rows a lane's atomics touch per query half
    for r in range(1, 4):
        e.salu(f"s_add_u32 s{SROW + 2 * r}, s{SROW + 2 * r - 2}, s{SROWB}")
        e.salu(f"s_addc_u32 s{SROW + 2 * r + 1}, s{SROW + 2 * r - 1}, 0")
    e.salu(f"s_mov_b32 s{SHD}, s{SH}")                                                       # dQ offsets: first step 0, then H - 1 steps of one head, then a block step
    e.salu(f"s_mov_b32 s{SDQS}, 0")
    e.salu(f"s_movk_i32 s{SD128}, 128")
    e.salu(f"s_mov_b32 s{SDN4}, s{SN4}")
    e.salu(f"s_sub_u32 s{SHS}, s{SH}, 2")                                                    # stage offsets sit at the second pair (head 1): H - 2 head steps left in its block
    e.salu(f"s_cmp_eq_u32 s{SH}, 1")                                                         # (one head: the second pair opens the second block)
    e.salu(f"s_cselect_b32 s{SHS}, 0, s{SHS}")
    e.salu(f"s_mov_b32 s{SMASK}, -1")
    e.salu(f"s_mov_b32 s{SMASK + 1}, 0")
    e.salu(f"s_mov_b32 s{SC2}, s{SC}")
    e.salu(f"s_mov_b32 s{SC2 + 1}, s{SC}")
    # The loop's text must be valid for the state the back edge arrives in (MFMA results of the previous iteration still settling), which
    # needs at least the padding of the first entry: emit the two bodies from the post-iteration state until the text repeats.
    import copy

    def two_bodies(state):
        x = copy.deepcopy(state)
        x.lines, x.count, x.nops, x.cold = [], {}, 0, []
        BODY(x, 0, "a")
        BODY(x, 1, "b")
        x.salu(f"s_sub_u32 s{SCNT}, s{SCNT}, 1")
        x.salu(f"s_cmp_lg_u32 s{SCNT}, 0")
        x.salu("s_cbranch_scc1 .Lloop%=")
        return x

    # the list-scheduled body must compute, register by register and store by store, what the hand-ordered body computes (that one
    # is pinned on the GPU: dK / dV bit-identical to the compiled kernel): same symbolic value in every register after two iterations,
    # same set of LDS stores / atomics / DMA requests per iteration
    if BODY is not body:
        def trace(fn):
            x = copy.deepcopy(e)
            for tg in ("a", "b", "c", "d"):
                fn(x, 0 if tg in "ac" else 1, tg)
            return x
        xm, xs = trace(body), trace(BODY)
        # (not compared: KF -- the scheduled body leaves the NEXT pair's K rows there -- and the address / scratch registers from v192 up)
        bad = [r for r in set(xm.val) | set(xs.val) if xm.vid(r) != xs.vid(r) and (r[0] == "a" or (r[1] < 192 and not KF <= r[1] < KF + 16))]
        assert not bad, ("dataflow differs from the hand-ordered body in", sorted(bad)[:12])
        assert sorted(map(repr, xm.sinks)) == sorted(map(repr, xs.sinks)), "stores / atomics differ from the hand-ordered body"
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
    e.cold = second.cold
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
    e.lines.append("s_branch .Lend%=")
    for blk in e.cold:                                                                      # block / end-of-part steps of the running offsets (once per H pairs)
        e.lines += blk
    e.lines.append(".Lend%=:")
    return e, loop_len, loop_count, chk


def main():
    e, loop_len, loop_count, chk = generate()
    text = e.lines
    with open(OUT, "w") as f:
        macro = "OSUF_BWD512AQS" if QS else "OSUF_BWD512A"
        f.write(f"// GENERATED by tools/gen_attn_bwd512.py{' --qs' if QS else ''} -- do not edit.  Main loop of mqa_bwd_fused512a_kernel (csrc/attn.hip) as one asm statement.\n")
        f.write("// register map: see the generator; the C++ side binds its operands to the same physical registers.\n")
        f.write(f"#define {macro}_ASM \\\n")
        for l in text:
            f.write(f'  "{l}\\n\\t" \\\n')
        f.write('  ""\n')
        cl = ", ".join(f'"v{i}"' for i in range(64, 192)) + ", " + ", ".join(f'"a{i}"' for i in range(256))
        sc = ", ".join(f'"s{i}"' for i in (SD128, SDN4, SDQS, SHS, SHD, SMASK, SMASK + 1) + tuple(range(SROW, SROW + 8)) + (SC2, SC2 + 1))
        f.write(f'#define {macro}_CLOBBERS "memory", "vcc", "scc", "m0", {sc}, {cl}\n')     # m0: the LDS-DMA destinations
    if "--stats" in sys.argv:
        per_pair = {k: v / 2 for k, v in loop_count.items() if k not in ("init", "fini")}
        tot = sum(v for k, v in per_pair.items() if not k.startswith("mfma"))
        print("per pair:", {k: per_pair[k] for k in sorted(per_pair)})
        print(f"non-MFMA instructions per pair: {tot:.0f}; s_nop wait states inserted in total: {e.nops}")
    if "--gaps" in sys.argv:
        seen = set()
        for tag, name, budget, used in GAPLOG[-2 * 200:]:
            if tag == "a" and (tag, name) not in seen:
                seen.add((tag, name))
                print(f"  {name:12s} budget {budget:3d} used {used:4d} {'#' * (used // 2)}")
        tot = sum(max(32 if b == BUDGET_BIG else 16, 8 + u) for t, n, b, u in GAPLOG[-200:] if t == "a") 
        print("issue-model cycles per iteration (sum over gaps of max(MFMA, 8 + fillers)):", tot)
    print("wrote", OUT, len(text), "lines")


if __name__ == "__main__":
    main()
