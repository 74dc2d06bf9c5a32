"""Forward attention (osuf_mqa_fwd) at B=32 (N <= 4096) / B=32 of the sampler's 2B batch (N = 8192), H=16, D=64, under the product library and under
OSUF_HIP_LIB variants (same-box A/B).   python tools/time_fwd.py base <variant> ..."""
import os
import subprocess
import sys
CODE = r'''
import sys, torch
sys.path.insert(0, "/root/repo")
from osufusion_amd import ops
H, D = 16, 64
out = []
for N, B in ((8192, 32), (4096, 32), (2048, 32), (1024, 32), (1000, 32)):
    qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
    fn = lambda: ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 10
    out.append(f"N={N}: {t:.3f} ms ({4.0 * B * H * N * N * D / t / 1e9:.0f} TF/s)")
print("  ".join(out))
'''
for name in sys.argv[1:]:
    env = dict(os.environ)
    if name != "base":
        env["OSUF_HIP_LIB"] = f"/root/repo/osufusion_amd/csrc/libosuf_hip_{name}.so"
    r = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=300)
    print(f"{name:10s} {r.stdout.strip() or ('FAILED: ' + r.stderr[-300:])}", flush=True)
