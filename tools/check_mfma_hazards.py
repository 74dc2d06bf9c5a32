"""Lint for the inline-asm MFMA kernels (csrc/attn.hip, mqa_bwd_fused512_kernel): hipcc pads no hazards around an `asm` statement
(cdna_hip_programming.md 5.7), so the emitted code is checked here instead.
  * a VALU / VMEM-free rule: no VALU instruction that WRITES a register an MFMA reads (A, B or C) within the two instructions before it,
    unless an s_nop >= 1 sits between them;
  * no compiler-generated v_accvgpr_* (outside ;;#ASMSTART / ;;#ASMEND) and no scratch access inside the main loop (a spill there would wait for vmcnt(0), i.e. for the float atomics in flight).
Usage: python tools/check_mfma_hazards.py <kernel-symbol-substring> <file.s>     (file.s from hipcc -save-temps)
"""
import re
import sys


def regs(tok):
    """'v[18:33]' / 'v7' / 'a[0:15]' -> set of ('v', n)"""
    out = set()
    for kind, lo, hi in re.findall(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(kind, i) for i in range(int(lo), int(hi) + 1)}
    for kind, n in re.findall(r"\b([va])(\d+)\b", tok):
        out.add((kind, int(n)))
    return out


def main():
    sym, path = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    body = [l.strip() for l in lines[start:end] if l.strip() and not l.strip().startswith((";", "."))]
    bad = 0
    # Round 4: the fences BEHIND the asm MFMAs are checked too -- a non-MFMA instruction (or an MFMA through its A / B operand)
    # that touches a register an MFMA wrote needs 12 wait states behind a 32x32x16 and 8 behind a 16x16x32 (same rule as check_inc below)
    ws, ready, shape = 0, {}, {}
    for l in body:
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")] if rest else []
        if op == "s_nop":
            ws += int(toks[0]) + 1 if toks and toks[0].isdigit() else 1
            continue
        if op.startswith("v_mfma"):
            big = "32x32" in op
            dst = regs(toks[0])
            for t in toks[1:3]:
                for r in regs(t):
                    if ready.get(r, 0) > ws:
                        print(f"HAZARD: MFMA operand {r} read {ready[r] - ws} wait state(s) before its MFMA result is written back: {l}"); bad += 1
            c = regs(toks[3]) if len(toks) > 3 else set()
            for r in c:
                if ready.get(r, 0) > ws and not (c == dst and shape.get(r) == big):
                    print(f"HAZARD: MFMA C operand {r} from an MFMA of another shape / register: {l}"); bad += 1
            ws += 1
            for r in dst:
                ready[r] = ws + (12 if big else 8)
                shape[r] = big
            continue
        if op.startswith(("v_", "ds_", "global_", "scratch_", "buffer_")):
            for t in toks:
                for r in regs(t.split(" ")[0]):
                    if ready.get(r, 0) > ws:
                        print(f"HAZARD: {op} touches {r} {ready[r] - ws} wait state(s) before its MFMA result is written back: {l}"); bad += 1
        ws += 1                                             # (straight-line scan: the fall-through of every branch is checked with everything in flight)
    hist = []                                               # (op, written regs, is_valu, nop states)
    for l in body:
        op, _, rest = l.partition(" ")
        ops = [t.strip() for t in rest.split(",")]
        if op.startswith("v_mfma"):
            src = set().union(*[regs(t) for t in ops[1:]]) if len(ops) > 1 else set()
            states = 0
            for pop, pw, pvalu, pn in reversed(hist[-3:]):
                if pop.startswith("s_nop"):
                    states += pn
                    continue
                if states >= 2:
                    break
                if pvalu and (pw & src):
                    print(f"HAZARD: {pop} writes {sorted(pw & src)[:4]} {states} state(s) before: {l}")
                    bad += 1
                states += 1
        is_valu = op.startswith("v_") and not op.startswith("v_mfma")
        written = regs(ops[0]) if ops and (is_valu or op.startswith(("ds_read", "global_load", "scratch_load"))) else set()
        nop = int(ops[0]) + 1 if op == "s_nop" and ops and ops[0].isdigit() else 0
        hist.append((op, written, is_valu, nop))
    # main loop = between the first "Loop Header" comment and the following back-branch: approximate by the region holding MFMAs and a barrier
    text = "\n".join(lines[start:end])
    m = re.search(r"Loop Header.*?s_barrier", text, re.S)
    loop = m.group(0) if m else ""
    loop = re.sub(r";;#ASMSTART.*?;;#ASMEND", "", loop, flags=re.S)      # statements of our own may move asm-owned accumulators (cold paths)
    for pat in ("v_accvgpr", "scratch_"):
        n = len(re.findall(pat, loop))
        if n:
            print(f"LOOP: {n} x {pat} inside the main loop")
            bad += n
    nm = len(re.findall(r"v_mfma", text))
    print(f"{sym}: {nm} MFMAs checked, {bad} finding(s)")
    sys.exit(1 if bad else 0)


if __name__ == "__main__" and not (len(sys.argv) > 2 and sys.argv[1] == "--inc"):
    main()


# ------------------------------------------------------------------------------------------------------------------------
# Round 4: the generated loop of mqa_bwd_fused512a_kernel (csrc/attn_bwd512_asm.inc).  The generator pads its own hazards; this is
# an independent re-check of the emitted TEXT (it shares no state with tools/gen_attn_bwd512.py), plus the instruction census the
# round-3 review asked for.      python tools/check_mfma_hazards.py --inc osufusion_amd/csrc/attn_bwd512_asm.inc
#   * a non-MFMA instruction (or an MFMA through its A / B operand) that touches a register an MFMA wrote needs 12 wait states behind a
#     32x32x16 and 8 behind a 16x16x32 (one per instruction, N + 1 per s_nop N; the stream has no skippable region: its rarely taken
#     branches leave to out-of-line blocks that execute MORE instructions than the in-line path they replace); an MFMA may take the
#     previous result of the SAME shape as its C operand back to back;
#   * an MFMA reads no register a vector instruction wrote within the two preceding wait states.
def _inc_lines(path):
    out = []
    for l in open(path):
        m = re.match(r'\s*"(.*)\\n\\t" \\$', l)
        if m:
            out.append(m.group(1))
    return out


def check_inc(path, max_non_mfma=600):
    lines = _inc_lines(path)
    i0 = next(i for i, l in enumerate(lines) if l.startswith(".Lloop"))
    i1 = next(i for i, l in enumerate(lines) if l.startswith("s_cbranch_scc1 .Lloop"))
    ready, shape, valu_at = {}, {}, {}
    ws, bad = 0, 0
    census = {}
    for n, l in enumerate(lines):
        if l.endswith(":"):
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")] if rest else []
        if i0 < n <= i1:
            key = "mfma" if op.startswith("v_mfma") else op.split("_")[0] + ("_" + op.split("_")[1] if op.startswith(("ds_", "global_")) else "")
            census[key] = census.get(key, 0) + 1
        if op == "s_nop":
            ws += int(toks[0]) + 1
            continue
        if op.startswith("v_mfma"):
            big = "32x32" in op
            dst, a, b = regs(toks[0]), regs(toks[1]), regs(toks[2])
            c = regs(toks[3]) if len(toks) > 3 and not toks[3].strip().isdigit() else set()
            for r in a | b:
                if ready.get(r, 0) > ws:
                    print(f"HAZARD line {n}: MFMA operand {r} not written back ({ready[r] - ws} wait states short): {l}"); bad += 1
                if ws - valu_at.get(r, -99) < 2:
                    print(f"HAZARD line {n}: MFMA reads {r} {ws - valu_at[r]} wait state(s) behind a vector write: {l}"); bad += 1
            for r in c:
                if ready.get(r, 0) > ws and not (c == dst and shape.get(r) == big):
                    print(f"HAZARD line {n}: MFMA C operand {r} from an MFMA of another shape / register: {l}"); bad += 1
                if ws - valu_at.get(r, -99) < 2:
                    print(f"HAZARD line {n}: MFMA C operand {r} behind a vector write: {l}"); bad += 1
            ws += 1
            for r in dst:
                ready[r] = ws + (12 if big else 8)
                shape[r] = big
            continue
        touched = set()
        if op.startswith(("v_", "ds_", "global_")):
            touched = set().union(*[regs(t.split(" ")[0]) for t in toks]) if toks else set()
        for r in touched:
            if ready.get(r, 0) > ws:
                print(f"HAZARD line {n}: {op} touches {r} {ready[r] - ws} wait state(s) before its MFMA result is written back: {l}"); bad += 1
        ws += 1
        if op.startswith("v_") and toks:
            for r in regs(toks[0]):
                valu_at[r] = ws
    pairs = 2
    non_mfma = sum(v for k, v in census.items() if k != "mfma") / pairs
    print("per (head, 32-query block) pair:", {k: v / pairs for k, v in sorted(census.items())})
    print(f"{path}: {census.get('mfma', 0) / pairs:.0f} MFMAs and {non_mfma:.0f} other instructions per pair (limit {max_non_mfma}), {bad} hazard finding(s)")
    return bad == 0 and non_mfma <= max_non_mfma


if __name__ == "__main__" and len(sys.argv) > 2 and sys.argv[1] == "--inc":
    sys.exit(0 if check_inc(sys.argv[2]) else 1)
