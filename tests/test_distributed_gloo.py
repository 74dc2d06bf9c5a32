"""world_size-2 gloo test (CPU) of the data-parallel path used by bench.py for N > 1: flat parameter/gradient buffers,
bucketed all-reduce fired from post-accumulate-grad hooks during backward, sum-then-scale averaging.
Correctness criterion: the N-rank averaged gradient equals the 1-process gradient on the
concatenated batch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from osufusion_amd.train import FlatParameters, GradReducer, cosine_warmup_lr


def _model():
    torch.manual_seed(7)
    return nn.Sequential(nn.Linear(16, 64), nn.SiLU(), nn.Linear(64, 64), nn.SiLU(), nn.Linear(64, 8))


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(8, 16, generator=g), torch.randn(8, 8, generator=g)


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        model = _model()
        flat = FlatParameters(model, align=4)
        red = GradReducer(flat, bucket_mib=0.001)                 # ~260 floats per bucket -> several buckets
        assert len(red.bounds) >= 3 and red.enabled
        x, y = _data()
        xs, ys = x.chunk(world)[rank], y.chunk(world)[rank]
        for _ in range(2):                                        # two steps: hooks/pending counters must re-arm
            flat.zero_grad()
            loss = ((model(xs) - ys) ** 2).mean()
            loss.backward()
            red.finish()
        out[rank] = (flat.grad / world).clone()
        # every parameter's .grad must still alias the flat buffer
        for p, o in zip(flat.params, flat.offsets):
            assert p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(120)
def test_bucketed_allreduce_matches_single_process_gradient():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    model = _model()
    flat = FlatParameters(model, align=4)
    x, y = _data()
    ((model(x) - y) ** 2).mean().backward()
    ref = flat.grad
    for r in range(world):
        assert torch.allclose(out[r], ref, rtol=1e-5, atol=1e-7), f"rank {r}"
    assert torch.equal(out[0], out[1])


def test_flat_parameters_alias_and_order():
    model = _model()
    before = [p.detach().clone() for p in model.parameters()]
    flat = FlatParameters(model, align=64)
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p, b)
    # reverse registration order: the last layer's bias is first in the flat buffer
    assert flat.params[0] is list(model.parameters())[-1]
    flat.data.mul_(2.0)
    for p, b in zip(model.parameters(), before):
        assert torch.equal(p, 2 * b)
    for p in model.parameters():
        p.grad = None                                            # a foreign zero_grad(set_to_none=True)
    flat.zero_grad()
    assert all(p.grad is not None for p in model.parameters())


def test_cosine_warmup_schedule():
    assert cosine_warmup_lr(0, 1.0, 10, 100) == 0.0
    assert abs(cosine_warmup_lr(5, 1.0, 10, 100) - 0.5) < 1e-12
    assert abs(cosine_warmup_lr(10, 1.0, 10, 100) - 1.0) < 1e-12
    assert abs(cosine_warmup_lr(55, 1.0, 10, 100) - 0.5) < 1e-12
    assert cosine_warmup_lr(100, 1.0, 10, 100) < 1e-12
