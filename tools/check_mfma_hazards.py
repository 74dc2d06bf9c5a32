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


if __name__ == "__main__":
    main()
