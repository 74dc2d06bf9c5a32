"""Data formats either side of the hot path: dataset npz layout, padding collate, context normalisation,
model files, and checkpoint.pt interchange with torch.optim.AdamW (reference trainer.py:74-95,143-203; library/dataset.py:25-37;
scripts/dataset_creator.py:57-78,114,180)."""
import numpy as np
import pytest
import torch

from osufusion_amd import data as D


def test_dataset_files_round_trip_and_nan_guard(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.normal(-10, 3, (96, 300)).astype(np.float32)
    for i, L in enumerate((300, 300)):
        x = rng.uniform(-1, 1, (6, L)).astype(np.float32)
        c = D.normalize_context(np.array([4.0, 9.5, 9.5, 4.0, 6.0], dtype=np.float32))
        D.save_tensor(tmp_path / f"m{i}.map.npz", x, c, a)
        x2, a2, c2 = D.load_tensor(tmp_path / f"m{i}.map.npz")
        assert x2.dtype == a2.dtype == c2.dtype == torch.float32
        assert torch.equal(x2, torch.from_numpy(x)) and torch.equal(a2, torch.from_numpy(a)) and torch.equal(c2, torch.from_numpy(c))
    assert sorted(p.name for p in tmp_path.iterdir()) == ["m0.map.npz", "m1.map.npz", "spec.npz"]      # one spec per set
    bad = rng.uniform(-1, 1, (6, 10)).astype(np.float32)
    bad[2, 3] = np.nan
    (tmp_path / "s2").mkdir()
    D.save_tensor(tmp_path / "s2" / "b.map.npz", bad, np.zeros(5, np.float32), a[:, :10])
    with pytest.raises(ValueError, match="Invalid values"):
        D.load_tensor(tmp_path / "s2" / "b.map.npz")
    assert [p.name for p in D.filter_dataset(sorted(tmp_path.glob("*.map.npz")), 299)] == []
    assert len(D.filter_dataset(sorted(tmp_path.glob("*.map.npz")), 300)) == 2


def test_collate_pads_like_the_reference_trainer():
    g = torch.Generator().manual_seed(1)
    batch = [(torch.rand(6, L, generator=g), torch.rand(96, L, generator=g), torch.rand(5, generator=g)) for L in (7, 12, 1)]
    x, a, c, orig = D.collate_fn(batch)
    assert x.shape == (3, 6, 12) and a.shape == (3, 96, 12) and c.shape == (3, 5) and orig.tolist() == [7, 12, 1]
    for i, (xi, ai, ci) in enumerate(batch):
        L = xi.shape[1]
        assert torch.equal(x[i, :, :L], xi) and torch.equal(a[i, :, :L], ai) and torch.equal(c[i], ci)
        assert (x[i, :, L:] == -1.0).all() and (a[i, :, L:] == -23.0).all()       # trainer.py:84-85


def test_context_normalisation_is_its_own_inverse():
    c = np.array([4.0, 9.5, 9.5, 4.0, 6.0], dtype=np.float32)
    n = D.normalize_context(c.copy())
    assert np.allclose(n, [-0.2, 0.9, 0.9, -0.2, -0.4])
    assert np.allclose(D.unnormalize_context(n.copy()), c)
    assert torch.allclose(D.unnormalize_context(torch.from_numpy(n.copy())), torch.from_numpy(c))


def test_model_files_load_with_reference_key_names(tmp_path):
    from osufusion_amd.modules import unet as U
    kw = dict(dim_h=32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,),
              attn_dim_head=64, attn_heads=2, attn_kv_heads=1, attn_context_len=256)
    net = U.UNet(6, 96, 5, **kw)
    D.save_model_sd(net, tmp_path / "model.safetensors")
    torch.save({"model_state_dict": net.state_dict(), "optimizer_state_dict": {}, "scheduler_state_dict": {}, "rng_state": torch.get_rng_state()},
               tmp_path / "checkpoint.pt")
    for name in ("model.safetensors", "checkpoint.pt"):
        other = U.UNet(6, 96, 5, **kw)
        res = D.load_model_sd(other, tmp_path / name)
        assert res == {"missing": [], "unexpected": []}
        for (k1, v1), (k2, v2) in zip(net.state_dict().items(), other.state_dict().items()):
            assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.gpu
def test_checkpoint_interchange_with_torch_adamw(golden_dir):
    """Two fused steps here, export checkpoint.pt-style state, resume in a plain torch.optim.AdamW on a copy of the model (and
    the other way round): the next update is the same in both -- the fused kernel implements torch's AdamW and its state maps one
    to one onto torch's per-parameter layout."""
    import copy
    import osufusion_amd as oa
    from osufusion_amd import functional as Fn
    from osufusion_amd.models.diffusion import OsuFusion
    from osufusion_amd.pattern import synth_inputs
    from osufusion_amd.train import Trainer
    kw = dict(dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,),
              attn_dim_head=64, attn_heads=2, attn_kv_heads=1, attn_context_len=256)
    torch.manual_seed(0)
    model = OsuFusion(32, **kw).cuda()
    with torch.no_grad():
        model.unet.final_conv.weight.normal_(0.0, 0.02)
    x, a, c, t, noise = (torch.from_numpy(v).cuda() for v in synth_inputs("ckpt", 2, 256))
    try:
        tr = Trainer(model, lr=1e-3, weight_decay=1e-2, compute_dtype=torch.float32)
        for _ in range(2):
            tr.step(x, a, c, noise, t)
        ck = tr.state_dict()
        assert set(ck) == {"model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "rng_state"}
        # resume in torch.optim.AdamW on a structural copy
        ref = OsuFusion(32, **kw).cuda()
        ref.load_state_dict(ck["model_state_dict"])
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2)
        opt.load_state_dict(copy.deepcopy(ck["optimizer_state_dict"]))
        # identical gradients for the third step: take them from the fused trainer's backward
        tr.flat.zero_grad()
        with oa.forced_compute_dtype(torch.float32):
            model.loss_with(x, a, c, noise, t).backward()
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        for k, p in ref.named_parameters():
            p.grad = grads[k].clone()
        opt.step()
        tr.opt.step()
        worst = max(((p.detach() - dict(ref.named_parameters())[k].detach()).abs().max() / (p.detach().abs().max() + 1e-12)).item()
                    for k, p in model.named_parameters())
        assert worst < 1e-5, worst
        # and back: torch's state into a fresh fused trainer
        ck2 = {"model_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(), "scheduler_state_dict": {}, "rng_state": None}
        Fn.enable_direct_grads(False)
        model2 = OsuFusion(32, **kw).cuda()
        tr2 = Trainer(model2, lr=1e-3, weight_decay=1e-2, compute_dtype=torch.float32)
        tr2.load_state_dict(ck2)
        assert tr2.opt.step_count == 3
        o = tr2.flat.offsets[0]
        p0 = tr2.flat.params[0]
        k0 = [k for k, p in model2.named_parameters() if p is p0][0]
        idx = [k for k, _ in ref.named_parameters()].index(k0)
        assert torch.equal(tr2.opt.exp_avg[o:o + p0.numel()].view_as(p0), opt.state_dict()["state"][idx]["exp_avg"])
    finally:
        Fn.enable_direct_grads(False)
