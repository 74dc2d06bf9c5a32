"""Time ATTN_FUSED512A (B=32, H=16, N from argv) under each library variant built by tools/build_variants.sh (timing-only triage).
    python tools/time_variants.py 4096 base noatom novalu ...     ('base' = the product library)"""
import os
import subprocess
import sys
N = sys.argv[1]
CODE = r'''
import sys, os, torch
sys.path.insert(0, "/root/repo")
from osufusion_amd import ops
B, H, D, N = 32, 16, 64, int(sys.argv[1])
qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
do = torch.randn(B, N, H * D, device="cuda").to(torch.bfloat16)
delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, torch.cuda.current_stream().cuda_stream)
fn = lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, None, None, variant=ops.ATTN_FUSED512A, delta=delta)
for _ in range(3): fn()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10): fn()
e.record(); torch.cuda.synchronize()
print(f"{s.elapsed_time(e) / 10:.3f}")
'''
for name in sys.argv[2:]:
    env = dict(os.environ)
    if name != "base":
        env["OSUF_HIP_LIB"] = f"/root/repo/osufusion_amd/csrc/libosuf_hip_{name}.so"
    r = subprocess.run([sys.executable, "-c", CODE, N], env=env, capture_output=True, text=True, timeout=300)
    print(f"N={N} {name:12s} {r.stdout.strip() or ('FAILED: ' + r.stderr[-300:])} ms", flush=True)
