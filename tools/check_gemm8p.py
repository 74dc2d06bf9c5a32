"""The 8-phase 256^2 NT GEMM (gemm_nt_big8_kernel, the default; OSUF_GEMM_NO8P=1 selects the old loop) against the one-barrier-per-K-step 256^2 kernel it replaces: bit-identical
outputs on chip-filling shapes over repeated launches (a race screen: a fragment read that overtakes its DMA shows up as a changed tile), then
the timing of both, interleaved in ONE process (cdna_hip_programming.md 5.4 rule 24).
    python tools/check_gemm8p.py [--rounds 5] [--iters 20]"""
import argparse, os, sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops

ap = argparse.ArgumentParser(); ap.add_argument("--rounds", type=int, default=5); ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--screen", type=int, default=30); args = ap.parse_args()
M = 131072
SHAPES = [(256, 256, 1), (1152, 256, 1), (1024, 256, 1), (256, 1024, 1), (256, 64, 1), (256, 128, 1), (512, 512, 1), (1024, 1024, 1), (1024, 2048, 1), (256, 256, 3), (768, 768, 3)]


VAL = os.environ.get("CHECK_VAL", "1")                    # value the switch is set to (e.g. CHECK_VAR=OSUF_GEMM_BIG_MIN_TILES CHECK_VAL=off: the 128^2 kernel)
VAR = os.environ.get("CHECK_VAR", "OSUF_GEMM_NO8P")      # the switch under test: "plain" column = switch set (NO8P) / for other switches: unset


def run(x, w, out, p8, taps=1, L=None):
    if VAR == "OSUF_GEMM_NO8P":
        if p8: os.environ.pop("OSUF_GEMM_NO8P", None)
        else: os.environ["OSUF_GEMM_NO8P"] = "1"
    else:
        if p8: os.environ[VAR] = VAL
        else: os.environ.pop(VAR, None)
    if taps == 1:
        ops.gemm_nt(x, w, None, out=out)
    else:
        ops.gemm_nt(x, w, None, taps=taps, lin=L, lout=L, stride=1, pad=taps // 2, out=out)


bad = 0
# (N, K, taps, plain): plain = the k = 3 row compares the two PLAIN kernels (OSUF_GEMM_NOHALO); otherwise the two shared-panel kernels
HALO = [(256, 256, 3, False), (512, 512, 3, False), (768, 768, 3, False), (1024, 1024, 3, False), (256, 512, 3, False), (1024, 2048, 3, False), (256, 64, 3, False)]
for N, K, taps, plain in [(n_, k_, t_, True) for n_, k_, t_ in SHAPES] + HALO:
    if plain: os.environ["OSUF_GEMM_NOHALO"] = "1"
    else: os.environ.pop("OSUF_GEMM_NOHALO", None)
    m = M if N * K * taps <= 1024 * 2048 else M // 4
    L = 4096 if K <= 256 else 1024
    x = torch.randn(m, K, device="cuda").bfloat16(); w = (torch.randn(taps, N, K, device="cuda") * 0.05).bfloat16()
    ref = torch.empty(m, N, device="cuda", dtype=torch.bfloat16); out = torch.empty_like(ref)
    run(x, w, ref, False, taps, L)
    diff = 0
    for _ in range(args.screen):
        out.zero_()
        run(x, w, out, True, taps, L)
        diff += int((out != ref).sum().item())
    bad += diff
    ts = {False: [], True: []}
    for r in range(args.rounds):
        for p8 in (False, True):
            for _ in range(2): run(x, w, out, p8, taps, L)
            torch.cuda.synchronize()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(args.iters): run(x, w, out, p8, taps, L)
            e.record(); torch.cuda.synchronize()
            ts[p8].append(s.elapsed_time(e) / args.iters)
    fl = 2.0 * m * N * K * taps
    t0, t1 = sorted(ts[False])[len(ts[False]) // 2], sorted(ts[True])[len(ts[True]) // 2]
    ksteps = K * taps // 64
    tiles = -(-m // 256) * -(-N // 256)
    rounds_ = -(-tiles // 256)
    print(f"M={m:6d} N={N:5d} K={K:5d} taps={taps}{' ' if plain else 'h'} mismatches over {args.screen} launches: {diff:6d}   plain {t0 * 1e3:8.1f} us {fl / t0 / 1e9:6.0f} TF/s   "
          f"8-phase {t1 * 1e3:8.1f} us {fl / t1 / 1e9:6.0f} TF/s  ({t0 / t1:5.2f}x; min {min(ts[False]) * 1e3:.1f} / {min(ts[True]) * 1e3:.1f};  "
          f"us per K-step and tile round: {t0 * 1e3 / rounds_ / ksteps:.2f} -> {t1 * 1e3 / rounds_ / ksteps:.2f})", flush=True)
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
