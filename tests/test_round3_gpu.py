"""GPU tests added in round 3 (run with -m gpu on an MI355X), all through the C ABI:
  * cached GEMM operand packs never go stale across optimizer steps when derived weights were packed under no_grad
    (reentrant activation checkpointing, trainer.py:229; a sample() between two train steps, trainer.py:344-356);
  * a ResidualBlock applied twice in one forward (residual.py:118-137) completes its FiLM gradient once, after both uses.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    import osufusion_amd as oa
    from osufusion_amd import functional as Fn
    from osufusion_amd import ops

from tests.test_hip_parity import DEV, T, load_pattern, rell2, report

_TINY = dict(dim_h_mult=(1, 2), num_layer_blocks=(2, 2), num_middle_transformers=1, cross_embed_kernel_sizes=(3,), attn_dim_head=64,
             attn_context_len=512)


def _fresh_copy(model, kw):
    """A new model object (empty pack caches) holding the same parameter values."""
    from osufusion_amd.models.diffusion import OsuFusion
    twin = OsuFusion(32, **kw).to(DEV)
    twin.load_state_dict({k: v.detach().clone() for k, v in model.state_dict().items()})
    return twin


@pytest.mark.parametrize("heads,kv_heads", [(2, 1), (4, 2)])
@pytest.mark.parametrize("scenario", ["checkpointing", "sample_between_steps"])
def test_packs_follow_the_weights_after_no_grad_forwards(heads, kv_heads, scenario):
    """under no_grad every tensor DERIVED from parameters (merged CrossEmbed stem, Parallel's c3 + pad(c1), the
    padded final_conv, the GQA-permuted to_q / to_out views) is a leaf, so `is_leaf` let the grouped refresh register a job on such a
    temporary copy and re-pack from it after the next optimizer step: forward / dgrad then ran on pre-step weights for good.
    Here: train steps with (a) reentrant checkpointing, (b) a no_grad sample() in between; afterwards every registered job reads a
    real nn.Parameter, and the model's forward equals that of a fresh model (empty caches) holding the same weights -- bit for bit
    in the reproducible sampling mode."""
    from osufusion_amd.models.diffusion import OsuFusion
    from osufusion_amd.pattern import synth_inputs
    from osufusion_amd.train import Trainer
    kw = dict(_TINY, attn_heads=heads, attn_kv_heads=kv_heads)
    model = OsuFusion(32, **kw).to(DEV)
    load_pattern(model.unet)
    x, a, c, t, noise = (T(v) for v in synth_inputs(f"r3packs{heads}{kv_heads}", 2, 256))
    try:
        if scenario == "checkpointing":
            model.unet.set_gradient_checkpointing(True)
        tr = Trainer(model, lr=3e-3, compute_dtype=torch.bfloat16)
        model.sampling_timesteps = 2
        for step in range(3):
            tr.step(x, a, c, noise, t)
            if scenario == "sample_between_steps":
                with torch.no_grad():
                    model.sample(a, c, x=noise.clone(), cond_scale=2.0)          # rebuilds stale stem / head / GQA packs under no_grad
        params = {id(p) for p in model.parameters()}
        jobs = [j for j in Fn._PACK_JOBS.values() if j.cache() is not None]
        assert len(jobs) >= 20
        for j in jobs:
            assert all(isinstance(w, torch.nn.Parameter) and id(w) in params for w in j.ws), j.key
        # one more optimizer step: the grouped refresh now marks its entries valid for the NEW weights
        tr.step(x, a, c, noise, t)
        twin = _fresh_copy(model, kw)
        with torch.no_grad(), ops.reproducible_mode(True), oa.forced_compute_dtype(torch.bfloat16):
            got = model.unet(x, a, t, c)
            ref = twin.unet(x, a, t, c)
        report(f"packs_after_no_grad/{scenario}/h{heads}kv{kv_heads}", rel_l2=rell2(got, ref))
        assert torch.equal(got, ref), f"cached operand packs differ from the current weights: rel-L2 {rell2(got, ref):.3e}"
    finally:
        Fn.enable_direct_grads(False)


def test_resblock_applied_twice_reports_film_gradient_once(golden_dir):
    """a ResidualBlock used twice in one forward takes two taps of the grouped FiLM output; its `mlp.1` parameters
    are reported complete ONCE (after both uses) and carry the sum of both uses' gradients."""
    from osufusion_amd.modules.residual import ResidualBlock
    from osufusion_amd import runtime as rt
    from osufusion_amd.train import FlatParameters, GradReducer
    torch.manual_seed(5)
    blk = ResidualBlock(32, 32, dim_time=64, dim_cond=64).to(DEV)
    for p in blk.parameters():
        p.data.normal_(0, 0.2)
    x = torch.randn(2, 32, 64, device=DEV)
    te, ce = torch.randn(2, 64, device=DEV), torch.randn(2, 64, device=DEV)

    def run(group):
        for p in blk.parameters():
            p.grad = None
        rt.clear_shared_cat()
        with oa.forced_compute_dtype(torch.float32):
            rows = rt.to_rows(x, torch.float32)
            if group:
                emb = rt.shared_cat(te, ce)
                assert rt.film_prepare(emb, [blk.mlp[1]])
            y = blk.forward_rows(blk.forward_rows(rows, te, ce), te, ce)
            y.float().square().mean().backward()
        rt.clear_shared_cat()
        return {k: p.grad.detach().clone() for k, p in blk.named_parameters()}

    ref = run(False)
    flat = FlatParameters(blk)
    red = GradReducer(flat)
    Fn.enable_direct_grads(True, red.param_ready, red.param_complete)
    try:
        flat.zero_grad()
        red.begin()
        rt.clear_shared_cat()
        with oa.forced_compute_dtype(torch.float32):
            rows = rt.to_rows(x, torch.float32)
            emb = rt.shared_cat(te, ce)
            assert rt.film_prepare(emb, [blk.mlp[1]])
            blk.forward_rows(blk.forward_rows(rows, te, ce), te, ce).float().square().mean().backward()
        red.finish()
        rt.clear_shared_cat()
        assert red.duplicate_reports == 0
        assert sorted(red.order_log) == list(range(len(flat.params))), "every parameter exactly once"
        got = {k: p.grad.detach().clone() for k, p in blk.named_parameters()}
    finally:
        Fn.enable_direct_grads(False)
    gmax = max(v.abs().max().item() for v in ref.values())
    for k in ref:
        e = (got[k] - ref[k]).abs().max().item() / (ref[k].abs().max().item() + 1e-4 * gmax)
        assert e < 2e-2, (k, e)
