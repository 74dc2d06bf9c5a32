"""Time the pre-scaled-query attention backward (ATTN_FUSED512A, qs=True; B=32, H=16) under library variants (tools/build_qs_variants.sh), rounds interleaved.
    python tools/time_qs_variants.py 4096 base lat4 ..."""
import os, subprocess, sys
N = sys.argv[1]
CODE = r'''
import sys, torch
sys.path.insert(0, "/root/repo")
from osufusion_amd import ops
B, H, D, N = 32, 16, 64, int(sys.argv[1])
qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
qkv[..., : H * D] = (qkv[..., : H * D].float() * (D ** -0.5 * ops.LOG2E)).to(torch.bfloat16)
o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5, qs=True)
do = torch.randn(B, N, H * D, device="cuda").to(torch.bfloat16)
delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, torch.cuda.current_stream().cuda_stream)
fn = lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, None, None, variant=ops.ATTN_FUSED512A, delta=delta, qs=True)
ref = fn().float()
for _ in range(3): fn()
torch.cuda.synchronize()
best = 1e9
for r in range(3):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(6): fn()
    e.record(); torch.cuda.synchronize()
    best = min(best, s.elapsed_time(e) / 6)
print(f"{best:.3f} {ref.norm().item():.4f}")
'''
res = {}
for rnd in range(2):
    for name in sys.argv[2:]:
        env = dict(os.environ)
        if name != "base":
            env["OSUF_HIP_LIB"] = f"/root/repo/osufusion_amd/csrc/libosuf_hip_{name}.so"
        r = subprocess.run([sys.executable, "-c", CODE, N], env=env, capture_output=True, text=True, timeout=300)
        try:
            t, nrm = r.stdout.split()
            res.setdefault(name, []).append((float(t), nrm))
        except ValueError:
            print(name, "FAILED", r.stderr[-300:])
for name, ts in res.items():
    print(f"N={N} {name:12s} min {min(t for t, _ in ts):.3f} ms   all {[t for t, _ in ts]}  |dqkv| {ts[0][1]}", flush=True)
