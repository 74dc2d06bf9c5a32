"""CPU tests of the host logic around the kernels: weight packing + row-map geometry (checked against torch conv1d
autograd through a numpy emulation of the tap-GEMM contract in include/osufusion_hip.h), state-dict parity, C-ABI
surface, DDIM schedule, loud failure without a GPU."""
import json
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from osufusion_amd import _lib
from osufusion_amd import functional as Fn
from osufusion_amd import ops

ROOT = Path(__file__).resolve().parent.parent


# ---- numpy emulation of the osuf_gemm_nt / osuf_gemm_tn contracts (row-map modes 0..3) ----------------------
def map_row(i, t, Lin, Lout, stride, pad, mode):
    if mode == 3:
        if t == 3:
            return Lin - 1 if i == Lout - 2 else -1
        e = i - t
        if e < 0 or e % 2:
            return -1
        e //= 2
        return e if e < Lin else -1
    s = i * stride + t - pad
    if mode == 2:
        return s >> 1 if 0 <= s < 2 * Lin else -1
    if mode == 1 and s == Lin:
        s = Lin - 2
    return s if 0 <= s < Lin else -1


def emu_gemm_nt(a, w, bias=None, *, taps=1, n_out=None, lin=None, lout=None, stride=1, pad=0, mode=0, act=0, residual=None, rscale=None,
                dact=None, stats=None, want_pre=False, out=None, out_shape=None):
    A = a.reshape(-1, a.shape[-1]).double()
    W = (w if w.dim() == 3 else w.unsqueeze(0)).double()
    if lin is None:
        lin = lout = A.shape[0]
    nb = A.shape[0] // lin
    M, N = nb * lout, W.shape[1]
    C = torch.zeros(M, N, dtype=torch.float64)
    for m in range(M):
        b, i = divmod(m, lout)
        for t in range(taps):
            s = map_row(i, t, lin, lout, stride, pad, mode)
            if s >= 0:
                C[m] += W[t] @ A[b * lin + s]
    if bias is not None:
        C += bias.double()[:N]
    if residual is not None:
        C += residual.reshape(M, -1).double()
    C = C.float()
    return C.reshape(out_shape) if out_shape is not None else C


def emu_gemm_tn(dy, x, *, taps=1, lin=None, lout=None, stride=1, pad=0, mode=0, n1=None, out=None, conv_layout=False, accumulate=False,
                bias_out=None):
    Y = dy.reshape(-1, dy.shape[-1]).double()
    if bias_out is not None:                                   # osuf_gemm_tn_bias: += the column sums of dy
        bias_out += Y.sum(0).float()
    X = x.reshape(-1, x.shape[-1]).double()
    if lin is None:
        lin = lout = Y.shape[0]
    M = Y.shape[0]
    G = torch.zeros(taps, Y.shape[1], X.shape[1], dtype=torch.float64)
    for m in range(M):
        b, i = divmod(m, lout)
        for t in range(taps):
            s = map_row(i, t, lin, lout, stride, pad, mode)
            if s >= 0:
                G[t] += torch.outer(Y[m], X[b * lin + s])
    G = G.float()
    res = G.permute(1, 2, 0).contiguous() if conv_layout else G
    if out is not None:
        out.copy_(out + res.reshape(out.shape) if accumulate else res.reshape(out.shape))
        return out
    return res


def emu_pack_weight(w, dtype, kind="same", want_fwd=True, want_dgrad=True, fwd=None, dgrad=None, row_offset=0):
    """osuf_pack_weight's contract (include/osufusion_hip.h), element by element."""
    w3 = w if w.dim() == 3 else w.unsqueeze(-1)
    O, I, k = w3.shape
    kd = k if kind == "same" else 4
    if want_fwd and fwd is None:
        fwd = torch.zeros(k, O, I, dtype=dtype)
    if want_dgrad and dgrad is None:
        dgrad = torch.zeros(kd, I, O, dtype=dtype)
    for o in range(O):
        for i in range(I):
            taps = [float(w3[o, i, t]) for t in range(k)]
            if want_fwd:
                for t in range(k):
                    fwd[t, row_offset + o, i] = taps[t]
            if want_dgrad:
                if kind == "same":
                    vals = [taps[k - 1 - t] for t in range(k)]
                elif kind == "down":
                    vals = [taps[0], taps[1], taps[2], taps[2]]
                else:
                    vals = [taps[2], taps[1] + taps[2], taps[0] + taps[1], taps[0]]
                for t, v in enumerate(vals):
                    dgrad[t, i, row_offset + o] = v
    return (fwd if want_fwd else None), (dgrad if want_dgrad else None)


@pytest.fixture()
def emulated(monkeypatch):
    monkeypatch.setattr(ops, "gemm_nt", emu_gemm_nt)
    monkeypatch.setattr(ops, "gemm_tn", emu_gemm_tn)
    monkeypatch.setattr(ops, "pack_weight", emu_pack_weight)


@pytest.mark.parametrize("kind,k,L", [("same", 3, 12), ("same", 1, 9), ("same", 7, 10), ("same", 15, 20), ("down", 3, 12), ("down", 3, 2),
                                      ("up", 3, 7), ("up", 3, 1)])
def test_conv_geometry_and_packs_match_torch_autograd(emulated, kind, k, L):
    torch.manual_seed(0)
    Bn, Cin, Cout = 2, 5, 4
    x = torch.randn(Bn, Cin, L, requires_grad=True)
    w = torch.randn(Cout, Cin, k, requires_grad=True)
    b = torch.randn(Cout)
    if kind == "same":
        ref = F.conv1d(x, w, b, padding=k // 2)
    elif kind == "down":
        ref = F.conv1d(F.pad(x, (0, 1), mode="reflect"), w, b, stride=2)
    else:
        ref = F.conv1d(F.interpolate(x, scale_factor=2.0, mode="nearest"), w, b, padding=1)
    g = torch.randn_like(ref)
    ref.backward(g)
    rows = x.detach().permute(0, 2, 1).contiguous()
    cache = Fn.PackCache()
    out = Fn.conv_forward(rows, w.detach(), b, cache, kind)
    assert torch.allclose(out.permute(0, 2, 1), ref, atol=1e-5)
    dy = g.permute(0, 2, 1).contiguous()
    dx = Fn.conv_dgrad(dy, w.detach(), cache, kind, L)
    assert torch.allclose(dx.permute(0, 2, 1), x.grad, atol=1e-5)
    dw = Fn.conv_wgrad(dy, rows, w.detach(), kind)
    assert torch.allclose(dw, w.grad, atol=1e-5)


def test_linear_packs(emulated):
    torch.manual_seed(1)
    x = torch.randn(1, 6, 10, requires_grad=True)
    w = torch.randn(7, 10, requires_grad=True)
    ref = F.linear(x, w)
    g = torch.randn_like(ref)
    ref.backward(g)
    cache = Fn.PackCache()
    assert torch.allclose(Fn.conv_forward(x.detach(), w.detach(), None, cache, "same"), ref, atol=1e-5)
    assert torch.allclose(Fn.conv_dgrad(g, w.detach(), cache, "same", 6), x.grad, atol=1e-5)
    assert torch.allclose(Fn.conv_wgrad(g, x.detach(), w.detach(), "same"), w.grad, atol=1e-5)


def test_cross_embed_merge_matches_reference_semantics(emulated, monkeypatch):
    """Merged zero-padded-tap stem == three separate convs concatenated (unet.py:42-58), both stem layouts."""
    from osufusion_amd.modules import unet as U
    from osufusion_amd import runtime as rt
    monkeypatch.setattr(rt, "require_gpu", lambda t: None)

    def emu_ncl_to_rows(x, dtype, width, kt=1):
        Bn, C, L = x.shape
        out = torch.zeros(Bn, L, width)
        for t in range(kt):
            for n in range(L):
                s = n + t - kt // 2
                if 0 <= s < L:
                    out[:, n, t * C:(t + 1) * C] = x[:, :, s]
        return out
    monkeypatch.setattr(ops, "ncl_to_rows", emu_ncl_to_rows)
    torch.manual_seed(2)
    for dim, dout in ((6, 24), (16, 40)):
        m = U.CrossEmbedLayer(dim, dout, (3, 7, 15))
        x = torch.randn(2, dim, 11)
        ref = torch.cat([F.conv1d(x, c.weight, c.bias, padding=c.kernel_size[0] // 2) for c in m.convs], 1)
        with torch.no_grad():
            got = m.forward_rows(x, torch.float32)
        assert torch.allclose(got.permute(0, 2, 1), ref, atol=1e-5), dim


def test_parallel_merge(emulated, monkeypatch):
    from osufusion_amd.modules import unet as U
    torch.manual_seed(3)
    m = U.Parallel(torch.nn.Conv1d(8, 16, 3, padding=1), torch.nn.Conv1d(8, 16, 1))
    x = torch.randn(2, 8, 9)
    ref = m.fns[0](x) + m.fns[1](x)
    with torch.no_grad():
        got = m.forward_rows(x.permute(0, 2, 1).contiguous())
    assert torch.allclose(got.permute(0, 2, 1), ref, atol=1e-5)


# ---- state dict / API surface ----------------------------------------------------------------------------------
def test_state_dict_matches_reference_inventory(golden_dir):
    from osufusion_amd.models.diffusion import OsuFusion
    inv = json.loads((golden_dir / "state_dict_dim256.json").read_text())
    with torch.device("meta"):
        model = OsuFusion(256)
    sd = model.state_dict()
    assert len(sd) == 1239 and {k[len("unet."):] for k in sd} == set(inv)
    for k, s in inv.items():
        assert tuple(sd["unet." + k].shape) == tuple(s), k
    assert sum(p.numel() for p in model.parameters()) == 343_493_297
    assert float(torch.nn.Conv1d(4, 4, 1).weight.abs().sum()) > 0                  # sanity: default init is not zero ...
    real = OsuFusion(32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,), attn_heads=2)
    assert float(real.unet.final_conv.weight.abs().sum()) == 0.0                   # ... except final_conv (unet.py:354)


def test_api_surface_mirrors_reference():
    import inspect
    from osufusion_amd.models.diffusion import OsuFusion
    from osufusion_amd.modules import attention, residual, unet, utils
    sig = inspect.signature(unet.UNet.__init__)
    assert list(sig.parameters)[1:] == ["dim_in_x", "dim_in_a", "dim_in_c", "dim_h", "dim_h_mult", "num_layer_blocks", "num_middle_transformers",
                                        "cross_embed_kernel_sizes", "attn_dim_head", "attn_heads", "attn_kv_heads", "attn_context_len"]
    assert list(inspect.signature(unet.UNet.forward).parameters)[1:] == ["x", "a", "t", "c", "cond_drop_prob"]
    assert list(inspect.signature(OsuFusion.forward).parameters)[1:] == ["x", "a", "c", "orig_len"]
    assert list(inspect.signature(OsuFusion.sample).parameters)[1:] == ["a", "c", "x", "cond_scale"]
    assert inspect.signature(OsuFusion.sample).parameters["cond_scale"].default == 7.0
    for mod, names in ((unet, ["SinusoidalPositionEmbedding", "CrossEmbedLayer", "Upsample", "Downsample", "Parallel", "Attention", "FeedForward",
                               "TransformerBlock", "UNetBlock", "AudioEncoder", "UNet"]),
                       (residual, ["GlobalContext", "Block", "ResidualBlock"]), (attention, ["RotaryPositionEmbedding", "Attend"]),
                       (utils, ["prob_mask_like", "rotate_half", "apply_rotary_pos_emb", "right_pad_dims_to"])):
        for n in names:
            assert hasattr(mod, n), n
    m = OsuFusion(32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,), attn_heads=2)
    assert m.sampling_timesteps == 35 and m.cond_drop_prob == 0.5 and hasattr(m, "unet")
    with pytest.raises(AssertionError):                                            # trainer.py:296-299 catches exactly this type
        m(torch.zeros(1, 6, 32), torch.zeros(1, 96, 16), torch.zeros(1, 5))


def test_rectified_flow_api_and_oracle_selfcheck():
    """API mirror of models/rectified_flow.py + the oracle's midpoint rule on a linear ODE (dy/dt = -y: one exact check of the
    restated torchdiffeq fixed-grid midpoint step: y1 = y0 * (1 - dt + dt^2 / 2))."""
    import inspect
    from oracle import rectified_flow_oracle as RO
    from osufusion_amd.models.rectified_flow import OsuFusion as RF, cosmap
    assert list(inspect.signature(RF.sample).parameters)[1:] == ["a", "c", "x", "cond_scale"]
    assert inspect.signature(RF.sample).parameters["cond_scale"].default == 2.0
    assert inspect.signature(RF.__init__).parameters["sampling_timesteps"].default == 16
    t = torch.tensor([0.0, 0.5, 1.0])
    assert torch.allclose(cosmap(t)[:2], torch.tensor([0.0, 0.5]), atol=1e-6) and abs(cosmap(t)[2].item() - 1.0) < 1e-6
    assert torch.allclose(RO.cosmap(t), cosmap(t))
    with pytest.raises(AssertionError):
        RF(32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,), attn_heads=2)(
            torch.zeros(1, 6, 32), torch.zeros(1, 96, 16), torch.zeros(1, 5))


def test_product_path_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    from osufusion_amd.modules import residual
    blk = residual.Block(8, 8)
    with pytest.raises(RuntimeError, match="no CPU"):
        blk(torch.zeros(1, 8, 16))


def test_prob_mask_like_semantics():
    from osufusion_amd.modules.utils import prob_mask_like
    assert prob_mask_like((4,), 1.0, "cpu").all() and not prob_mask_like((4,), 0.0, "cpu").any()
    torch.manual_seed(0)
    m = prob_mask_like((10000,), 0.3, "cpu")
    assert 0.27 < m.float().mean() < 0.33


def test_ddim_schedule_known_answers():
    from osufusion_amd.models.diffusion import DDIMSchedule
    s = DDIMSchedule()
    for i, v in ((0, 0.99989998), (1, 0.99978006), (500, 0.07779665), (999, 4.0358304e-05)):
        assert abs(s.alphas_cumprod[i].item() - v) / v < 2e-6
    s.set_timesteps(35)
    assert s.timesteps[:2].tolist() == [952, 924] and s.timesteps[-2:].tolist() == [28, 0]
    s.set_timesteps(50)
    assert s.timesteps[:2].tolist() == [980, 960]
    c = s.step_coefficients(0)
    assert c[2] == 1.0 and c[3] == 0.0                                              # set_alpha_to_one


# ---- C ABI ------------------------------------------------------------------------------------------------------
def _header_decls():
    src = (ROOT / "include" / "osufusion_hip.h").read_text()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"\b(?:int|long)\s+(osuf_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S)


def test_capi_exports_every_declared_symbol():
    lib_path = _lib.LIB_PATH
    if not lib_path.exists():
        from osufusion_amd.csrc import build
        build.build()
    decls = _header_decls()
    assert len(decls) >= 29
    syms = subprocess.run(["nm", "-D", "--defined-only", str(lib_path)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (osuf_\w+)", syms))
    for name, _ in decls:
        assert name in exported, f"{name} declared in include/osufusion_hip.h but not exported by libosuf_hip.so"
    assert set(_lib.SIGNATURES) == {n for n, _ in decls}


def test_capi_ctypes_signatures_match_header():
    from ctypes import c_float, c_int, c_long, c_void_p
    for name, params in _header_decls():
        ps = [p.strip() for p in params.split(",")] if params.strip() != "void" else []
        want = []
        for p in ps:
            if "*" in p or "hipStream_t" in p:
                want.append(c_void_p)
            elif p.startswith("long"):
                want.append(c_long)
            elif p.startswith("int"):
                want.append(c_int)
            elif p.startswith("float"):
                want.append(c_float)
            else:
                raise AssertionError(f"unparsed parameter {p!r} of {name}")
        assert _lib.SIGNATURES[name] == want, name


def test_capi_descriptor_structs_match_their_numpy_layouts():
    """osuf_pack_desc / osuf_linear_desc (device tables of the grouped entry points): the header's field order and C layout (LP64:
    8-byte pointers and longs, 4-byte ints, natural alignment) against the numpy structured dtypes ops.py fills them through."""
    src = (ROOT / "include" / "osufusion_hip.h").read_text()
    for cname, dt in (("osuf_pack_desc", ops.PACK_DESC), ("osuf_linear_desc", ops.LINEAR_DESC)):
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", src, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields, off = [], 0
        for decl in (d.strip() for d in body.split(";") if d.strip()):
            size = 8 if ("*" in decl or decl.startswith("long")) else 4
            assert "*" in decl or decl.split()[0] in ("long", "int", "const"), decl
            names = [n.strip().lstrip("*") for n in decl.replace("*", " * ").split("*")[-1].split(",")] if "*" in decl else \
                    [n.strip() for n in decl.split(None, 1)[1].split(",")]
            for n in names:
                off = (off + size - 1) // size * size
                fields.append((n, off, size))
                off += size
        total = (off + 7) // 8 * 8
        assert total == dt.itemsize, (cname, total, dt.itemsize)
        assert len(fields) == len(dt.names)
        alias = {"f_tapstride": "f_ts", "d_tapstride": "d_ts"}
        for (n, o, sz), dn in zip(fields, dt.names):
            assert alias.get(n, n) == dn and dt.fields[dn][1] == o and dt.fields[dn][0].itemsize == sz, (cname, n, dn, o)


def test_capi_library_loads_and_reports_version():
    lib = _lib.load()
    assert lib.osuf_version() == 1          # host-only entry point: no GPU needed
