"""The hand-placed 512-key attention backward (ATTN_FUSED512A) against the compiled 512-key sweep (ATTN_FUSED512): dK / dV must be bit-identical
(same fragment maps, same order of every accumulation), dQ equal up to the order of its float atomics; then timings.
    python tools/check_bwd512a.py [--time]"""
import os
import sys
os.environ["OSUF_ALLOW_TIMING_BUILDS"] = "1"
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops

D = 64


def run(B, N, H, var, rope, qsplit=0, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    qkv = torch.randn(B, N, (H + 2) * D, device="cuda", generator=g).to(torch.bfloat16)
    o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
    do = torch.randn(B, N, H * D, device="cuda", generator=g).to(torch.bfloat16)
    cos = sin = None
    if rope:
        ang = torch.rand(N, D // 2, device="cuda", generator=g) * 6.28
        cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    out = ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.float32, cos, sin, variant=var, qsplit=qsplit)
    torch.cuda.synchronize()
    return out


def main():
    ok = True
    for (B, N, H, rope, qs) in [(1, 512, 2, False, 0), (2, 1024, 16, True, 0), (9, 1024, 4, False, 2), (8, 2048, 16, True, 0), (3, 4096, 16, False, 0), (2, 1024, 3, False, 1)]:
        ref = run(B, N, H, ops.ATTN_FUSED512, rope, qs)
        got = run(B, N, H, ops.ATTN_FUSED512A, rope, qs)
        dq_r, dk_r, dv_r = ref[..., :H * D], ref[..., H * D:(H + 1) * D], ref[..., (H + 1) * D:]
        dq_g, dk_g, dv_g = got[..., :H * D], got[..., H * D:(H + 1) * D], got[..., (H + 1) * D:]
        e_dq = ((dq_g - dq_r).norm() / dq_r.norm()).item()
        bit_k, bit_v = torch.equal(dk_g, dk_r), torch.equal(dv_g, dv_r)
        e_dk = ((dk_g - dk_r).norm() / dk_r.norm()).item()
        e_dv = ((dv_g - dv_r).norm() / dv_r.norm()).item()
        fin = bool(torch.isfinite(got).all())
        good = fin and bit_k and bit_v and e_dq < 1e-5
        ok &= good
        print(f"B={B} N={N} H={H} rope={rope} qsplit={qs}: dq rel {e_dq:.2e}  dk bit-equal {bit_k} ({e_dk:.2e})  dv bit-equal {bit_v} ({e_dv:.2e})  finite {fin}  {'ok' if good else 'MISMATCH'}", flush=True)
    if "--time" in sys.argv and ok:
        B, H = 32, 16
        for N in (4096, 2048, 1024):
            qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
            o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
            do = torch.randn(B, N, H * D, device="cuda").to(torch.bfloat16)
            delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
            ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, torch.cuda.current_stream().cuda_stream)
            f = 8.0 * B * H * N * N * D
            row = []
            for name, var in (("fused512", ops.ATTN_FUSED512), ("fused512a", ops.ATTN_FUSED512A), ("fused512-noatomics", ops.ATTN_FUSED512_TIMING)):
                fn = lambda: ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, None, None, variant=var, delta=delta)
                for _ in range(2): fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(8): fn()
                e.record(); torch.cuda.synchronize()
                t = s.elapsed_time(e) / 8
                row.append(f"{name} {t:7.3f} ms ({f / t / 1e9:5.0f} alg TF/s)")
            print(f"N={N:5d}  " + " | ".join(row), flush=True)
    sys.exit(0 if ok else 1)


main()
