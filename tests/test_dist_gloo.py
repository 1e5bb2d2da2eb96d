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
        out.put(flat.flat.detach().numpy().copy())      # plain bytes: a tensor would travel as a handle into this process
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
    got = torch.from_numpy(q.get(timeout=120))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    torch.testing.assert_close(got, flat_ref.flat.detach(), rtol=1e-5, atol=1e-6)


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


# ---------------------------------------------------------------------------------------------- start-up / sharding / state
def _worker_startup(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adyolo_amd.rng import DropoutStream
    torch.manual_seed(100 + 7 * rank)                  # ranks that (wrongly) seeded differently ...
    model = nn.Sequential(nn.Linear(6, 5), nn.BatchNorm1d(5), nn.Linear(5, 3))
    model[1].running_mean.fill_(float(rank + 1))
    flat = FlatParameters(model)
    flat.broadcast(0)                                  # ... start from rank 0's parameters and buffers anyway
    red = BucketedAllReduce(flat, n_buckets=2)
    # only the LAST layer gets a gradient: the other bucket never fires from a hook and finish() must launch it
    flat.zero_grad()
    model[2](torch.ones(2, 5) * (rank + 1)).sum().backward()
    scale = red.finish()
    torch.manual_seed(100)
    stream = DropoutStream(0x5EED)
    model[1].running_mean.fill_(float(rank + 1))
    flat.average_buffers()
    # (numpy copies, not tensors: a tensor in a multiprocessing queue is a handle the parent must fetch from this process
    #  while it is still alive)
    out.put((rank, flat.flat.detach().numpy().copy(), (flat.flat_grad.detach() * scale).numpy().copy(), red.fired_from_hooks,
             red.fired_from_finish, stream.seed, model[1].running_mean.detach().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_startup_broadcast_leftover_buckets_and_rank_seeds():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_startup, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, p0, g0, h0, f0, s0, rm0), (_, p1, g1, h1, f1, s1, rm1) = got
    assert (p0 == p1).all(), "ranks must start from rank 0's parameters"
    assert (g0 == g1).all(), "averaged gradients must be identical on every rank"
    assert h0 + f0 == 2 and f0 >= 1 and (h0, f0) == (h1, f1)
    assert s0 != s1, "dropout streams must differ between the data-parallel ranks"
    assert (rm0 == rm1).all() and float(rm0[0]) == 1.5            # running statistics averaged over the ranks


def test_fused_adam_state_dict_uses_module_parameter_order():
    """ONE optimizer-state format: torch.optim.Adam's, indexed like model.parameters() (reference train.py:149, 236)."""
    from adyolo_amd import checkpoint

    class _Opt:                                        # FusedAdam without the HIP step: only the state containers
        def __init__(self, flat):
            self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, 1e-3, (0.9, 0.999), 1e-8, 0.0
            self.exp_avg, self.exp_avg_sq, self.step_count = torch.zeros_like(flat.flat), torch.zeros_like(flat.flat), 0
    m = _model()
    ref_params = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    adam = torch.optim.Adam(ref_params, lr=1e-3)
    for p in ref_params:
        p.grad = torch.randn_like(p)
    adam.step()
    flat = FlatParameters(m)
    opt = _Opt(flat)
    checkpoint.load_optimizer_state_dict(opt, None, adam.state_dict())
    back = checkpoint.optimizer_state_dict(opt, None)
    ref_sd = adam.state_dict()
    assert back["param_groups"][0]["params"] == ref_sd["param_groups"][0]["params"]
    for i, p in enumerate(ref_params):
        assert back["state"][i]["exp_avg"].shape == p.shape
        torch.testing.assert_close(back["state"][i]["exp_avg"], ref_sd["state"][i]["exp_avg"])
        torch.testing.assert_close(back["state"][i]["exp_avg_sq"], ref_sd["state"][i]["exp_avg_sq"])
    adam2 = torch.optim.Adam(ref_params, lr=1e-3)
    adam2.load_state_dict(back)                        # and torch's own optimizer accepts what we write


class _TinyEncoder(nn.Module):
    """A module holding a DropoutStream and a BatchNorm buffer, like the encoders (rng.collect / rng.restore walk vars())."""

    def __init__(self):
        super().__init__()
        from adyolo_amd.rng import DropoutStream
        self.lin = nn.Linear(4, 4)
        self.bn = nn.BatchNorm1d(4)
        self.dropout_stream = DropoutStream(0x5EED)


class _HostAdam:                                       # FusedAdam without the HIP step: only the state containers
    def __init__(self, flat):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, 1e-3, (0.9, 0.999), 1e-8, 0.0
        self.exp_avg, self.exp_avg_sq, self.step_count = torch.zeros_like(flat.flat), torch.zeros_like(flat.flat), 3


def _worker_resume(rank, world, port, path, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from adyolo_amd import checkpoint
    torch.manual_seed(100)
    model = _TinyEncoder()
    flat = FlatParameters(model)
    opt = _HostAdam(flat)
    s_before = model.dropout_stream.seed
    model.dropout_stream.draw(1000 + rank)                     # ranks are at different offsets when the checkpoint is cut
    model.bn.running_mean.fill_(float(rank + 1))
    # every rank calls save_checkpoint (it averages the BatchNorm buffers over the ranks: a collective); rank 0 writes
    checkpoint.save_checkpoint(path, model, opt, 7, 0.5, {}, [], "cpu")
    assert float(model.bn.running_mean[0]) == float(rank + 1)  # saving has no side effect on the live statistics (round 4)
    dist.barrier()
    assert os.path.exists(path)
    torch.manual_seed(12345)                                   # a resumed process starts from some other seed
    model2 = _TinyEncoder()
    opt2 = _HostAdam(FlatParameters(model2))
    ck = checkpoint.load_checkpoint(path, model2, opt2, device="cpu")
    st = model2.dropout_stream
    out.put((rank, s_before, st.seed, st.offset, float(model2.bn.running_mean[0]), ck["start_epoch_nb"], opt2.step_count))
    dist.barrier()
    dist.destroy_process_group()


def test_resume_keeps_rank_distinct_dropout_seeds_and_saves_averaged_buffers(tmp_path):
    """the ONE rank-0 checkpoint is restored on every rank; the dropout stream stores its rank-independent
    base seed and re-applies the rank on restore, so resumed ranks keep drawing different masks (and each continues its
    own pre-checkpoint seed); offsets are rank 0's on every rank; the saved running statistics are the rank average."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    path = str(tmp_path / "model_ckpt.h5")
    procs = [ctx.Process(target=_worker_resume, args=(r, 2, port, path, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, b0, s0, o0, rm0, e0, c0), (_, b1, s1, o1, rm1, e1, c1) = got
    assert s0 != s1, "resumed ranks must not share one dropout seed"
    assert (s0, s1) == (b0, b1), "each rank continues the seed it had before the checkpoint"
    assert o0 == o1 == 1000, "offsets come from the rank-0 checkpoint"
    assert rm0 == rm1 == 1.5 and e0 == e1 == 7 and c0 == c1 == 3


def test_dropout_stream_accepts_round2_state():
    """Checkpoints written before the base seed was stored carry the rank-mixed ``seed``: still restorable (verbatim)."""
    from adyolo_amd.rng import DropoutStream
    s = DropoutStream(1)
    s.set_state({"seed": 1234567, "offset": 8})
    assert s.seed == 1234567 and s.offset == 8
    torch.manual_seed(5)
    a, b = DropoutStream(2), DropoutStream(2)
    a.draw(16)
    b.set_state(a.state())
    assert b.seed == a.seed and b.offset == 16


def test_dropout_stream_round2_state_survives_a_second_checkpoint(monkeypatch):
    """a stream restored from a round-2 state ({'seed', 'offset'}) used to write base_seed = None into the
    NEXT checkpoint, which could not be loaded, and every resumed rank kept rank 0's seed.  Now: the legacy seed (written by
    rank 0, where the rank mix is the identity) is the base, state() round-trips, and ranks draw different masks again."""
    from adyolo_amd.rng import DropoutStream
    s = DropoutStream(1)
    s.set_state({"seed": 1234567, "offset": 8})
    st = s.state()                                        # what the next checkpoint stores
    assert st["base_seed"] == 1234567 and st["offset"] == 8
    t = DropoutStream(1)
    t.set_state(st)                                       # ... and the resume after that
    assert t.seed == s.seed == 1234567 and t.offset == 8
    t.set_state({"base_seed": None, "seed": 99, "offset": 3})      # a file written by the broken version: base_seed None
    assert t.seed == 99 and t.offset == 3
    monkeypatch.setenv("RANK", "1")                       # the same legacy state restored on rank 1: another seed
    r1 = DropoutStream(1)
    r1.set_state({"seed": 1234567, "offset": 8})
    assert r1.seed != 1234567 and r1.state()["base_seed"] == 1234567


class _SinkLinear(torch.autograd.Function):
    """A node that behaves like the gfx950 gradient producers under ``functional.GradSink``: it writes the weight gradient
    straight into the parameter's slice of the flat gradient buffer, tells the reducer by hand and hands autograd None."""

    @staticmethod
    def forward(ctx, x, w, red, idx):
        ctx.save_for_backward(x, w)
        ctx.red, ctx.idx = red, idx
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        w.grad.add_(dy.t() @ x)                  # "kernel" writing into the flat buffer's view
        ctx.red.notify(ctx.idx)
        return dy @ w, None, None, None


def _worker_sink(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    ws = nn.ParameterList([nn.Parameter(torch.randn(6, 6) * 0.3) for _ in range(6)])
    flat = FlatParameters(ws)
    red = BucketedAllReduce(flat, n_buckets=3)
    index = {id(p): i for i, p in enumerate(flat.params)}
    torch.manual_seed(10 + rank)
    x = torch.randn(5, 6)
    early = []
    launch = red._launch

    def spy(b):                                  # at launch time every member's gradient must already be in the buffer
        s, e, mem = red.buckets[b]
        early.append(any(float(flat.params[i].grad.abs().sum()) == 0.0 for i in mem))
        launch(b)
    red._launch = spy
    flat.zero_grad()
    h = x
    for k, w in enumerate(ws):                   # every second layer is "sunk" (each bucket's FIRST gradient), the others go through autograd
        h = _SinkLinear.apply(h, w, red, index[id(w)]) if k % 2 == 1 else h @ w.t()
        h = torch.tanh(h)
    h.sum().backward()
    scale = red.finish()
    out.put((rank, (flat.flat_grad * scale).numpy().copy(), early, red.fired_from_hooks, red.fired_from_finish))
    dist.barrier()
    dist.destroy_process_group()


def test_sunk_gradients_are_counted_once_per_step():
    """Round-3 bug: a parameter whose gradient bypasses autograd (GradSink) was counted twice (by ``notify`` and by its
    post-accumulate-grad hook, which PyTorch fires even for a None gradient), so buckets were all-reduced when half of
    their gradients were still unwritten.  World size 2: no bucket may be launched with an empty member, and the result
    must be the mean of the two ranks' gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sink, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, g0, early0, h0, f0), (_, g1, early1, h1, f1) = got
    assert not any(early0) and not any(early1), "a bucket was reduced before all of its gradients were written"
    assert (g0 == g1).all() and h0 + f0 == 3 and (h0, f0) == (h1, f1)
    # reference: the two ranks' gradients by plain autograd, averaged
    refs = []
    for rank in range(2):
        torch.manual_seed(0)
        ws = [torch.nn.Parameter(torch.randn(6, 6) * 0.3) for _ in range(6)]
        torch.manual_seed(10 + rank)
        h = torch.randn(5, 6)
        for w in ws:
            h = torch.tanh(h @ w.t())
        h.sum().backward()
        refs.append(torch.cat([w.grad.reshape(-1) for w in reversed(ws)]))
    ref = ((refs[0] + refs[1]) / 2).numpy()
    assert abs(g0[:ref.size] - ref).max() < 1e-6
