"""CPU, world_size 2, gloo: the N>1 path (flat gradient buffer + bucketed asynchronous all-reduce + averaging)
gives the same parameter update as a single process on the concatenated batch for a loss that is a mean
over samples (dist.py is device-agnostic torch.distributed plumbing)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

import adyolo_amd  # noqa: F401
from adyolo_amd.dist import BucketedAllReduce, FlatParameters


def _model():
    torch.manual_seed(0)
    return nn.Sequential(nn.Linear(16, 32), nn.Tanh(), nn.Linear(32, 8), nn.Tanh(), nn.Linear(8, 4))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, xs, ys, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = _model()
    flat = FlatParameters(model)
    red = BucketedAllReduce(flat, n_buckets=3)
    assert len(red.buckets) >= 2
    for step in range(2):
        flat.zero_grad()
        x, y = xs[step][rank::world], ys[step][rank::world]
        loss = ((model(x) - y) ** 2).mean()
        loss.backward()
        scale = red.finish()
        assert scale == 1.0 / world
        with torch.no_grad():
            flat.flat.add_(flat.flat_grad, alpha=-0.1 * scale)
    if rank == 0:
        out.put(flat.flat.clone())
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_matches_single_process():
    torch.manual_seed(1)
    xs = [torch.randn(8, 16) for _ in range(2)]
    ys = [torch.randn(8, 4) for _ in range(2)]
    ref = _model()
    flat_ref = FlatParameters(ref)
    for step in range(2):
        flat_ref.zero_grad()
        ((ref(xs[step]) - ys[step]) ** 2).mean().backward()
        with torch.no_grad():
            flat_ref.flat.add_(flat_ref.flat_grad, alpha=-0.1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, xs, ys, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.testing.assert_close(got, flat_ref.flat, rtol=1e-5, atol=1e-6)


def test_flat_parameters_keep_module_semantics():
    m = _model()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParameters(m)
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    m(torch.randn(3, 16)).sum().backward()
    assert flat.flat_grad.abs().sum() > 0
    for p in m.parameters():
        assert p.grad.data_ptr() >= flat.flat_grad.data_ptr()
    flat.zero_grad()
    assert float(flat.flat_grad.abs().sum()) == 0.0
