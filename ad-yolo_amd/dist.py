"""Data parallelism over RCCL/xGMI: one process per GPU, gradients live in ONE flat fp32 buffer that is
cut into a few contiguous buckets laid out in backward order; each bucket is all-reduced asynchronously
(torch.distributed backend "nccl" == RCCL on ROCm; its internal stream overlaps with the rest of the
backward pass) as soon as the last gradient of the bucket has been accumulated.

The reference has no distributed code at all (single-process train loop, /root/reference/src/train.py:40-62),
so this layer has no reference counterpart.  Semantics chosen (documented in DESIGN.md): DDP-conventional --
every rank normalises its loss terms and BatchNorm statistics over its own micro-batch, gradients are
averaged over ranks.  Payload: 6 682 093 fp32 = 26.7 MB per step; with 7 x ~153 GB/s xGMI links per GPU
the ring time is ~0.3 ms against a >100 ms step, so 4 buckets are plenty.

Device-agnostic on purpose (only torch.distributed plumbing): the same code runs under gloo on CPU in
tests/test_dist_gloo.py with world_size 2.
"""
import os

import torch
import torch.distributed as dist


def _params_changed():
    """parameters / buffers were written in place by a collective: evaluation-mode caches keyed on ops.PARAMS_EPOCH expire"""
    from . import ops
    ops.params_changed()


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or os.environ.get("ADYOLO_FORCE_DP_HOOKS") == "1") and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


class FlatParameters:
    """Re-homes every parameter of ``module`` into one flat buffer (and every .grad into another), so that
    one fused-Adam launch updates the model and gradient buckets are contiguous slices."""

    def __init__(self, module, reverse=True):
        params = [p for p in module.parameters() if p.requires_grad]
        self.module_params = params             # module.parameters() order (= torch.optim state_dict indexing)
        self.buffers = [b for b in module.buffers() if b.is_floating_point()]      # BatchNorm running statistics
        # backward produces gradients roughly in reverse registration order: put the last layers first so
        # that bucket 0 fills first
        self.params = list(reversed(params)) if reverse else params
        total = sum(p.numel() for p in self.params)
        pad = (-total) % 4
        dev, dt = self.params[0].device, self.params[0].dtype
        self.flat = torch.zeros(total + pad, dtype=dt, device=dev)
        self.flat_grad = torch.zeros(total + pad, dtype=dt, device=dev)
        self.numel = total
        self.offsets = []
        off = 0
        for p in self.params:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view_as(p.data)
            p.grad = self.flat_grad[off:off + n].view_as(p.data)
            self.offsets.append((off, n))
            off += n

    def broadcast(self, src=0, group=None):
        """Start-up consistency under data parallelism: every rank takes rank ``src``'s parameters and floating-point
        buffers (BatchNorm running statistics), so a differing seed or a partial load cannot diverge silently."""
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.flat, src, group=group)
            for b in self.buffers:
                dist.broadcast(b, src, group=group)
            _params_changed()

    def average_buffers(self, group=None):
        """BatchNorm running statistics are per-rank under DDP-conventional semantics; average them over the ranks (e.g.
        before rank 0 writes a checkpoint) so the saved statistics describe the whole data stream, not one shard."""
        if dist.is_initialized() and dist.get_world_size(group) > 1:
            w = dist.get_world_size(group)
            for b in self.buffers:
                dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)
                b.div_(w)
            _params_changed()

    def zero_grad(self):
        self.flat_grad.zero_()
        for p, (off, n) in zip(self.params, self.offsets):      # re-attach if someone set .grad = None
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + off * self.flat_grad.element_size():
                p.grad = self.flat_grad[off:off + n].view_as(p.data)


class BucketedAllReduce:
    """Asynchronous bucketed gradient averaging over a FlatParameters layout."""

    def __init__(self, flat: FlatParameters, n_buckets=4, group=None):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # ADYOLO_FORCE_DP_HOOKS=1 exercises the bucketed RCCL path even with one rank (1-GPU smoke test of the N>1 code)
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("ADYOLO_FORCE_DP_HOOKS") == "1")
        total = flat.numel
        target = max(1, (total + n_buckets - 1) // n_buckets)
        self.buckets = []           # (start, end, [param indices])
        start, members, acc = 0, [], 0
        for i, (off, n) in enumerate(flat.offsets):
            members.append(i)
            acc += n
            if acc >= target or i == len(flat.offsets) - 1:
                self.buckets.append((start, off + n, members))
                start, members, acc = off + n, [], 0
        self._bucket_of = {}
        for b, (_, _, mem) in enumerate(self.buckets):
            for i in mem:
                self._bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._hooks = []
        self.fired_from_hooks = 0            # buckets whose all-reduce was issued by a gradient hook (overlapped with backward)
        self.fired_from_finish = 0           # buckets finish() had to launch (their gradients never all arrived)
        if self.active:
            for i, p in enumerate(flat.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        self.reset()

    def reset(self):
        self._pending = [len(mem) for (_, _, mem) in self.buckets]
        self._arrived = [False] * len(self.flat.params)
        self._works = []

    def _arrive(self, idx):
        """Gradient ``idx`` is complete.  Counted ONCE per step: a parameter whose gradient a kernel wrote straight into
        the flat buffer (functional.GradSink -> ``notify``) is reported a second time by autograd itself, because the
        post-accumulate-grad hook of a parameter also fires when the node handed back None for it (PyTorch 2.10).  Counting
        both made every bucket look complete after HALF of its members -- its all-reduce then summed gradients that had
        not been written yet and wrote those zeros back over the real ones (found by the two-rank test on the real model,
        tests/test_gpu_parity_scale.py; with one rank the in-place reduction is an identity and hid it)."""
        if self._arrived[idx]:
            return
        self._arrived[idx] = True
        b = self._bucket_of[idx]
        self._pending[b] -= 1
        if self._pending[b] == 0:
            self.fired_from_hooks += 1
            self._launch(b)

    def _make_hook(self, idx):
        def hook(_param):
            if self.active:                 # (bench.py switches the reducer off for its exposed-all-reduce measurement)
                self._arrive(idx)
        return hook

    def notify(self, idx):
        """Gradient ``idx`` (position in flat.params) is complete although autograd never saw it (functional.GradSink)."""
        if self.active:
            self._arrive(idx)

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        view = self.flat.flat_grad[s:e]
        self._works.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Wait for all bucket reductions (launching any that never fired, e.g. unused parameters) and return
        the factor the optimizer must apply to the summed gradients (1/world)."""
        if self.active:
            for b, left in enumerate(self._pending):          # bucket-index order: identical on every rank
                if left > 0:
                    self._pending[b] = 0
                    self.fired_from_finish += 1
                    self._launch(b)
            for w in self._works:
                w.wait()
        self.reset()
        return 1.0 / self.world

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
