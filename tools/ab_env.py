"""In-process A/B of an environment switch the library re-reads on every call (OSUF_GEMM_NO8P, OSUF_GEMM_NOHALO, ...) on the headline train
step: blocks of steps alternate between the two settings in ONE process on ONE device (cdna_hip_programming.md 5.4 rule 24), so box-to-box
and run-to-run spread cancels.      python tools/ab_env.py OSUF_GEMM_NO8P [--rounds 4] [--steps 6]"""
import argparse, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
import bench
from osufusion_amd import ops
from osufusion_amd.train import Trainer

ap = argparse.ArgumentParser(); ap.add_argument("var"); ap.add_argument("--rounds", type=int, default=4); ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
model = bench.build_model("cuda", bench.DIM_H)
trainer = Trainer(model, lr=1e-4, clip_grad_norm=1.0, compute_dtype=torch.bfloat16)
x, a, c, noise, t = bench.synth_batch(0, "cuda", bench.BATCH, bench.LENGTH)
for _ in range(3):
    trainer.step(x, a, c, noise, t)
torch.cuda.synchronize()
res = {False: [], True: []}
att = {False: [], True: []}
for r in range(args.rounds):
    for on in (False, True):
        if on: os.environ[args.var] = "1"
        else: os.environ.pop(args.var, None)
        trainer.step(x, a, c, noise, t)
        torch.cuda.synchronize()
        prof = ops.KernelTimer(list(bench.ATTN_UNITS)); ops.set_kernel_timer(prof)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.step(x, a, c, noise, t)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps * 1e3
        ops.set_kernel_timer(None)
        st = prof.summary()
        res[on].append(dt)
        att[on].append({k: round(v["total_ms"] / args.steps, 2) for k, v in st.items()})
        print(f"round {r} {args.var}={'1' if on else 'unset'}: {dt:7.2f} ms/step  attention {att[on][-1]}", flush=True)
os.environ.pop(args.var, None)
med = lambda v: sorted(v)[len(v) // 2]
print(f"median ms/step: unset {med(res[False]):.2f}   {args.var}=1 {med(res[True]):.2f}   (min {min(res[False]):.2f} / {min(res[True]):.2f})")
