"""Round 4 GPU checks that have no home in the earlier files."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from osufusion_amd import ops
from tests.test_hip_parity import DEV, rell2, report


@pytest.mark.parametrize("M,L,N1,N2,taps,conv_layout", [(16384, 4096, 256, 256, 3, True),      # merged-taps kernel, 128^2 tiles, many splits
                                                         (16384, 4096, 1152, 256, 1, False),    # 256^2 kernel, kernel's own layout
                                                         (8192, 512, 1024, 1024, 3, True),      # deep level: few splits, large tiles
                                                         (16384, 4096, 256, 512, 1, True)])
def test_bf16_pair_partial_tiles_of_the_split_weight_gradient(monkeypatch, M, L, N1, N2, taps, conv_layout):
    """The bf16 weight-gradient kernels leave their per-split partial tiles as bf16 pairs of rows (half the write-out and half the reduce's
    reads) and wgrad_reduce_pk_kernel sums them in fp32 in a fixed order: against the fp32 partial tiles (OSUF_WGRAD_F32_PARTIALS=1) the result
    moves by the rounding of one bf16 per split partial -- on these random operands, where the partials of different splits are uncorrelated and
    the total is no larger than its parts, that is the full 2^-9 / sqrt(3) ~ 1.7e-3 of the result (the worst case; real gradients sum correlated
    partials) -- against 1.1e-2 for the bf16 train step's gradients as a whole (tests/test_full_size.py holds those to the oracle's own
    bf16-autocast floor parameter by parameter, with these tiles in the path); twice == bit-equal;
    accumulate=True adds into the target (residual.py:70,115, unet.py:118-123,149-156 under autocast)."""
    torch.manual_seed(3)
    dy = (torch.randn(M, N1, device=DEV) * 0.5).to(torch.bfloat16)
    x = torch.randn(M, N2, device=DEV).to(torch.bfloat16)
    kw = dict(taps=taps, lin=L, lout=L, stride=1, pad=taps // 2, mode=0, conv_layout=conv_layout)
    got = ops.gemm_tn(dy, x, **kw)
    again = ops.gemm_tn(dy, x, **kw)
    assert torch.equal(got, again)
    monkeypatch.setenv("OSUF_WGRAD_F32_PARTIALS", "1")
    ref = ops.gemm_tn(dy, x, **kw)
    monkeypatch.delenv("OSUF_WGRAD_F32_PARTIALS")
    e = rell2(got, ref)
    base = torch.full_like(got, 0.25)
    acc = ops.gemm_tn(dy, x, out=base.clone(), accumulate=True, **kw)
    e_acc = rell2(acc - 0.25, got)
    # fp64 product of the middle tap on a slice, for scale: what the bf16 rounding of a partial is small against
    t = taps // 2
    want = (dy[:, :64].double().T @ x.double())                                   # (64, N2): tap t of a stride-1 'same' conv is the unshifted product
    mine = (got[:64, :, t] if conv_layout else got[t, :64, :]).double()
    ref64 = (ref[:64, :, t] if conv_layout else ref[t, :64, :]).double()
    report(f"wgrad_bf16_partials/M{M}N{N1}x{N2}t{taps}", vs_fp32_partials=e, accumulate=e_acc, vs_fp64=rell2(mine, want), fp32_partials_vs_fp64=rell2(ref64, want))
    assert e < 3e-3, e
    assert e_acc < 1e-5, e_acc
    assert rell2(mine, want) < 3e-3
