"""GPU, world_size 2 (both ranks share cuda:0, gloo transport): the data-parallel train path end to end through the HIP
kernels -- direct gradient accumulation into the flat buffer, bucketed all-reduce fired from the kernels' "gradient ready"
notifications and autograd hooks, sum-then-scale.  Criterion: averaged N-rank gradient == the
1-process gradient on the concatenated batch."""
import json
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _model():
    from osufusion_amd.models.diffusion import OsuFusion
    from osufusion_amd.pattern import param_pattern
    model = OsuFusion(32, dim_h_mult=(1, 2), num_layer_blocks=(1, 1), num_middle_transformers=1, cross_embed_kernel_sizes=(3,),
                      attn_dim_head=64, attn_heads=2, attn_kv_heads=1, attn_context_len=256).cuda()
    sd = {k: torch.from_numpy(param_pattern(k, tuple(v.shape))).cuda() for k, v in model.unet.state_dict().items()}
    model.unet.load_state_dict(sd)
    return model


def _batch():
    from osufusion_amd.pattern import synth_inputs
    return tuple(torch.from_numpy(v).cuda() for v in synth_inputs("ddp", 4, 256))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from osufusion_amd import forced_compute_dtype
        from osufusion_amd.train import Trainer
        model = _model()
        trainer = Trainer(model, bucket_mib=0.25, compute_dtype=torch.float32)
        assert trainer.reducer.enabled and len(trainer.reducer.bounds) >= 3
        x, a, c, t, noise = (v.chunk(world)[rank].contiguous() for v in _batch())
        for _ in range(2):                                      # twice: bucket counters must re-arm
            trainer.flat.zero_grad()
            with forced_compute_dtype(torch.float32):
                model.loss_with(x, a, c, noise, t, cond_drop_prob=0.0).backward()
            trainer.reducer.finish()
        torch.cuda.synchronize()
        out[rank] = (trainer.flat.grad / world).cpu()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_gradient_matches_single_process():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    from osufusion_amd import forced_compute_dtype
    from osufusion_amd import functional as Fn
    from osufusion_amd.train import Trainer
    try:
        model = _model()
        trainer = Trainer(model, compute_dtype=torch.float32)
        x, a, c, t, noise = _batch()
        trainer.flat.zero_grad()
        with forced_compute_dtype(torch.float32):
            # mean over the full batch == mean of the two half-batch means (equal halves)
            model.loss_with(x, a, c, noise, t, cond_drop_prob=0.0).backward()
        torch.cuda.synchronize()
        ref = trainer.flat.grad.cpu()
    finally:
        Fn.enable_direct_grads(False)
    assert torch.equal(out[0], out[1])
    scale = ref.abs().max()
    err = ((out[0] - ref).abs().max() / scale).item()
    assert err < 5e-3, err                                       # fp32 kernels; bf16 attention noise floor only


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world", [2, 4])                    # (the box admits 6 GPU processes: this test + 4 ranks)
def test_bench_main_with_ranks(tmp_path, world):
    """bench.py's world > 1 branch (process-group init, per-rank data, barrier-bracketed timing, MAX over ranks, rank-0 JSON line)
    launched exactly as the driver launches it -- `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` -- on this
    one-GPU box: gloo transport and both ranks on cuda:0 (OSUF_DIST_BACKEND / OSUF_SINGLE_DEVICE rehearsal knobs), a small UNet.
    What it cannot show is RCCL itself (trainer.py:264-269,301 -> one process per GPU over xGMI): that needs the 8-GPU node."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, OSUF_DIST_BACKEND="gloo", OSUF_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), str(root / "bench.py"), "--gpus", str(world), "--steps", "2", "--warmup", "2",
           "--dim-h", "96", "--length", "256", "--batch", "2"]
    r = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == world and out["steps"] == 2 and out["warmup"] == 2 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == f"dp{world}" and out["config"]["global_batch"] == 2 * world
    assert out["value"] > 0 and out["ms_per_step"] > 0 and out["higher_is_better"] is True
    assert abs(out["value"] - world * 2 / (out["ms_per_step"] * 2 / 1e3) * (2 / 32)) < 1e-2 * out["value"] + 1e-3     # whole-job aggregate, in B=32 steps
    assert "cpu_baseline" not in out and "secondary" not in out  # single-GPU legs only
    import math
    assert math.isfinite(out["loss"]) and math.isfinite(out["grad_norm"])
    comm = out["comm"]                                           # the self-diagnosis a first real multi-GPU run prints
    assert comm["world"] == world and comm["backend"] == "gloo" and comm["fingerprints_agree"] and len(comm["per_rank"]) == world
    assert comm["allreduce_calls"] == comm["buckets"] >= 1 and comm["allreduce_bytes"] > 0
    assert all(r["allreduce_bytes"] == comm["allreduce_bytes"] for r in comm["per_rank"])
    assert comm["allreduce_alone_ms"] > 0 and comm["order_disagreements"] == 0


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_comm", ["fp32", "bf16"])
def test_bench_one_rank_rccl_rehearsal(grad_comm):
    """The RCCL leg itself on this one-GPU box: OSUF_DIST_REHEARSE=1 makes bench.py open a ONE-rank `nccl` (= RCCL) process group and run the
    N > 1 code path unchanged -- communicator set-up with device_id, the bucketed async all-reduces issued beside the backward from the
    gradient-complete hooks, finish()'s waits, the comm diagnostics and the stand-alone all-reduce timing.  No byte crosses a link, so this says
    nothing about xGMI rates; it does say that every call the 8-GPU run makes into torch.distributed's RCCL backend is accepted and ordered
    (trainer.py:264-269,301).  The same small step without the group gives the same loss."""
    import math
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    args = [sys.executable, str(root / "bench.py"), "--steps", "2", "--warmup", "2", "--dim-h", "96", "--length", "256", "--batch", "2",
            "--no-sampler", "--no-cpu-baseline", "--no-config5", "--no-fp32-mode", "--grad-comm", grad_comm]

    def run(extra):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), **extra)
        env.pop("OSUF_DIST_BACKEND", None)
        if not extra:
            env.pop("OSUF_DIST_REHEARSE", None)
        r = subprocess.run(args, env=env, cwd=root, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]
        return json.loads(lines[0])

    out, plain = run({"OSUF_DIST_REHEARSE": "1"}), run({})
    comm = out["comm"]
    assert "comm" not in plain and out["n_gpus"] == 1 and comm["rehearsal"] is True
    assert comm["backend"] == "nccl" and comm["world"] == 1 and comm["fingerprints_agree"]
    assert comm["comm_dtype"] == ("bfloat16" if grad_comm == "bf16" else "float32")
    assert comm["allreduce_calls"] == comm["buckets"] >= 1 and comm["out_of_order_completions"] == 0 and comm["order_disagreements"] == 0
    per_el = 2 if grad_comm == "bf16" else 4
    assert comm["allreduce_bytes"] % per_el == 0 and comm["allreduce_bytes"] > 0 and comm["allreduce_alone_ms"] > 0
    assert math.isfinite(out["loss"]) and math.isfinite(out["grad_norm"])
    # a one-rank SUM is the identity (bf16 buckets: one rounding of every gradient element): same step as without the group
    assert abs(out["loss"] - plain["loss"]) <= 2e-3 * abs(plain["loss"]) + 1e-5
    assert abs(out["grad_norm"] - plain["grad_norm"]) <= (2e-2 if grad_comm == "bf16" else 1e-2) * plain["grad_norm"]
