"""ATTN_FUSED512A alone at B=32, H=16, N from argv (for counter passes; OSUF_HIP_LIB selects a variant library)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from osufusion_amd import ops
B, H, D, N = 32, 16, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
qkv = torch.randn(B, N, (H + 2) * D, device="cuda").to(torch.bfloat16)
o, lse = ops.mqa_fwd(qkv, B, N, H, D, torch.bfloat16, D ** -0.5)
do = torch.randn(B, N, H * D, device="cuda").to(torch.bfloat16)
delta = torch.empty(B, H, N, dtype=torch.float32, device="cuda")
ops.call("osuf_attn_delta", do.data_ptr(), H * D, o.data_ptr(), H * D, 1, delta.data_ptr(), B, H, N, D, torch.cuda.current_stream().cuda_stream)
for _ in range(6):
    ops.mqa_bwd(qkv, o, do, lse, B, N, H, D, D ** -0.5, torch.bfloat16, None, None, variant=ops.ATTN_FUSED512A, delta=delta)
torch.cuda.synchronize()
