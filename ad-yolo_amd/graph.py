"""hipGraph capture of the hot path: one graph per input shape, replayed step after step.

Why: a train step of se-resnet34 + adyolo is ~770 kernel launches issued through ctypes from Python autograd nodes
(~15-25 us of host time each).  At the benchmark shape (64 x 60 s) 150 ms of kernel time hide that; at the REFERENCE's own
shapes -- 16 x 20 s training chunks (src/configs/hyp_train.yaml:3), one 60 s clip at evaluation (src/train.py:130-133,
src/test.py:81) -- the GPU work is shorter than its launch sequence and the step is host-bound.  Every entry point of
libadyolo_hip.so takes its stream, never allocates and never synchronises, so the whole step (K1 features -> forward ->
loss -> backward -> Adam) records into ONE hipGraph (``torch.cuda.CUDAGraph`` is hipGraph on ROCm; PyTorch is plumbing:
the private memory pool and the capture stream).

What had to move to the device for that: nothing in a recorded launch may change from step to step, so the Adam step
counter (``adyolo_adam_step_dev``) and the running offset of the dropout stream (``adyolo_dropout_apply_dev`` +
``adyolo_counter_add``) live in device memory and are advanced by the graph itself; the AD-YOLO target list (M rows, M
varies from batch to batch) is padded to a fixed capacity with rows the assignment kernel skips (batch index -1).

Results are bit-identical to the eager path (tests/test_gpu_graph.py compares losses and parameters with torch.equal).
"""
import collections
import contextlib
import gc
import os

import torch

from . import functional as Fn
from . import ops
from . import rng as _rng

TARGET_QUANTUM = 4096          # AD-YOLO target rows are padded up to a multiple of this (bounds the number of graphs)
MAX_GRAPHS = int(os.environ.get("ADYOLO_MAX_GRAPHS", "8"))     # recorded graphs kept per StepGraphs / ForwardGraphs (least recently
#                                used one evicted): every graph owns a private memory pool with the activations of its shape, and a
#                                data set with many distinct clip lengths / target capacities would otherwise grow without bound


def _lru_insert(entries, key, ent, limit):
    """entries: OrderedDict, most recently used last.  Evicts (outside any capture: the caller is not recording) until the new
    entry fits; the evicted graphs and their pools are released by reference counting."""
    evicted = 0
    while len(entries) >= max(1, limit):
        entries.popitem(last=False)
        evicted += 1
    entries[key] = ent
    return evicted


@contextlib.contextmanager
def _quiet_collector():
    """No garbage collection while a stream is capturing: a recorded graph whose last reference dies in a collection that
    happens to run DURING another capture is destroyed there (``~CUDAGraph`` -> hipGraphDestroy), which HIP refuses with
    "operation not permitted when stream is capturing" and the process aborts (seen once in ten runs of bench.py's
    extra_configs, round 4).  Collect first, then keep the collector off until the capture has ended."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class _Entry:
    __slots__ = ("graph", "audio", "target", "loss", "deltas", "n_replays")


class StepGraphs:
    """The captured steps of one ``train.TrainStep`` (single process).  ``step(audio, target)``: the first call at a new
    (audio shape, target capacity) runs eagerly (lazy initialisations, allocator warm-up -- and it IS a real step, nothing
    is discarded), the second records the graph, every call from then on copies the inputs into the graph's static
    buffers (skipped when the caller already works in them: ``static_inputs``) and replays it."""

    def __init__(self, trainer, warm_calls=1):
        self.trainer = trainer
        self.warm_calls = warm_calls
        self.entries = collections.OrderedDict()      # most recently used last; at most MAX_GRAPHS
        self.seen = {}
        self.eager_only = set()        # input shapes whose step could not be recorded (host-computed values inside it)
        self.evictions = 0
        self.streams = _rng.streams(trainer.model)
        self.captures = self.replays = self.eager_steps = 0
        self.switches = ops.switch_stamp()     # the recorded steps' kernel choice: graphs recorded under another table are dropped

    # ---------------------------------------------------------------------------------------- input handling
    @staticmethod
    def _capacity(target):
        if target.dim() == 2 and target.shape[1] == 7:        # AD-YOLO rows (M, 7): M varies per batch
            return ((target.shape[0] + TARGET_QUANTUM - 1) // TARGET_QUANTUM) * TARGET_QUANTUM
        return None

    def _key(self, audio, target):
        cap = self._capacity(target)
        return (tuple(audio.shape), cap if cap is not None else tuple(target.shape))

    def static_inputs(self, audio_shape, target_like):
        """(audio, target) buffers of the graph for this shape once it exists (else None): a producer may write the next
        batch straight into them and pass them to ``step`` (no copy then)."""
        cap = self._capacity(target_like)
        ent = self.entries.get((tuple(audio_shape), cap if cap is not None else tuple(target_like.shape)))
        return (ent.audio, ent.target) if ent is not None else None

    def _load(self, ent, audio, target):
        if audio.data_ptr() != ent.audio.data_ptr():
            ent.audio.copy_(audio, non_blocking=True)
        if target.data_ptr() != ent.target.data_ptr():
            if self._capacity(target) is not None:
                m = target.shape[0]
                ent.target[:m].copy_(target, non_blocking=True)
                if m < ent.target.shape[0]:
                    ent.target[m:, 0].fill_(-1.0)           # rows the assignment kernel skips (csrc/loss.hip: b < 0)
            else:
                ent.target.copy_(target, non_blocking=True)

    # ---------------------------------------------------------------------------------------- capture / replay
    def _capture(self, key, audio, target):
        tr = self.trainer
        dev = audio.device
        ent = _Entry()
        ent.audio = torch.empty_like(audio)
        cap = self._capacity(target)
        if cap is not None:
            ent.target = torch.full((cap, 7), -1.0, dtype=torch.float32, device=dev)
        else:
            ent.target = torch.empty(tuple(target.shape), dtype=torch.float32, device=dev)
        ent.n_replays = 0
        self._load(ent, audio, target)
        # host-side state the captured step advances: put back afterwards (recording runs nothing), re-applied per replay
        step0 = tr.optimizer.step_count
        tr.optimizer.sync_device_step()
        for s in self.streams:
            s.sync_device()
            s.begin_capture(dev)
            s.sync_device()
        graph = torch.cuda.CUDAGraph()
        try:
            with _quiet_collector(), torch.cuda.graph(graph):
                loss = tr.step_eager(ent.audio, ent.target)
                for s in self.streams:
                    ops.counter_add_(s.dev, s.offset - s.capture_base)
        except BaseException:
            # ``torch.cuda.graph.__exit__`` ends the capture also when its body raised; should the capture stream still be
            # recording (its capture_end failed too), end it here -- the caller falls back to eager launches on this stream's
            # device and a stream left capturing would refuse them
            try:
                if torch.cuda.is_current_stream_capturing():
                    graph.capture_end()
            except Exception:                                                    # noqa: BLE001
                pass
            raise
        finally:
            ent.deltas = [s.end_capture() for s in self.streams]
            tr.optimizer.step_count = step0
            tr.optimizer._dev_step_value = step0
        ent.graph, ent.loss = graph, loss
        self.evictions += _lru_insert(self.entries, key, ent, MAX_GRAPHS)
        self.captures += 1
        return ent

    def _mark_eager(self, key, why):
        import warnings
        warnings.warn("adyolo: train step not hipGraph-capturable (%s); running it eagerly" % why)
        self.eager_only.add(key)

    @staticmethod
    def _capture_related(e):
        """Is ``e`` an error CAUSED by recording (the step itself is fine when launched eagerly)?  ``NotImplementedError``: a
        host-computed value asked for under capture (``rng.DropoutStream``); a ``RuntimeError`` whose text names the capture -- HIP's
        "operation not permitted when stream is capturing" / "... capture invalidated", torch's "... during CUDA graph capture".
        Everything else (out of memory, a bug in a kernel wrapper, a failing assertion) is NOT: falling back to eager launches
        for it would turn a real failure into a silent, permanent slowdown (round 5, ADVICE) -- it propagates."""
        if isinstance(e, NotImplementedError):
            return True
        oom = getattr(torch, "OutOfMemoryError", ())
        return isinstance(e, RuntimeError) and not isinstance(e, oom) and "captur" in str(e).lower()

    def _reset_step_state(self):
        """Host-side state a step that died half way (inside backward, inside the optimizer) can leave behind, put back before the
        step runs again: the gradient sink (views / reducer hook), pending BatchNorm counter bumps, the reducer's bucket
        counters.  (The RNG offsets and the optimizer's step count are restored by ``_capture`` itself; the block links that
        carry per-patch sums and ReLU-mask bits between blocks are made anew by every forward pass.)"""
        Fn.SINK.end()
        del Fn._COUNTER_SCOPE[:]
        red = getattr(self.trainer, "reducer", None)
        if red is not None and getattr(red, "active", False):
            red.reset()

    def step(self, audio, target):
        tr = self.trainer
        target = target.to(torch.float32)
        key = self._key(audio, target)
        sw = ops.switch_stamp()
        if sw != self.switches:                # ``ops.reload_thresholds()`` / ADYOLO_CONV_ALGO moved the dispatch: record again
            self.entries.clear()
            self.seen.clear()
            self.switches = sw
        if key in self.eager_only:
            self.eager_steps += 1
            return tr.step_eager(audio, target)
        ent = self.entries.get(key)
        if ent is None:
            n = self.seen.get(key, 0)
            self.seen[key] = n + 1
            if n < self.warm_calls:
                self.eager_steps += 1
                before = sum(s.host_draws for s in self.streams)
                loss = tr.step_eager(audio, target)
                if sum(s.host_draws for s in self.streams) != before:
                    # the step drew host-computed values (the ResNet-Conformer's attention-dropout seeds, a dropout mask
                    # override): it cannot be replayed -- decided here, on the warm-up step, without attempting a capture
                    self._mark_eager(key, "its dropout draws host-computed values")
                return loss
            try:
                ent = self._capture(key, audio, target)
            except Exception as e:                                               # noqa: BLE001
                # second line of defence.  ``_capture`` has ended the capture and put the RNG offsets / step count back; whatever
                # else the half-recorded step touched on the host is reset here.  Only an error CAUSED by recording makes the
                # shape eager-only (round 4: NotImplementedError from a host draw; round 5: a RuntimeError that names the
                # capture); anything else -- out of memory, a real bug -- is re-raised to the caller with the trainer left usable
                self._reset_step_state()
                if not self._capture_related(e):
                    self.seen[key] = n                                           # (the next call at this shape tries again)
                    raise
                self._mark_eager(key, "%s: %s" % (type(e).__name__, e))
                self.eager_steps += 1
                return tr.step_eager(audio, target)
        else:
            self._load(ent, audio, target)
            self.entries.move_to_end(key)
        tr.optimizer.sync_device_step()
        for s in self.streams:
            s.sync_device()
        ent.graph.replay()
        tr.optimizer.replayed()
        for s, d in zip(self.streams, ent.deltas):
            s.replayed(d)
        ent.n_replays += 1
        self.replays += 1
        return ent.loss.clone()


class ForwardGraphs:
    """Evaluation forward (``test_epoch``, reference src/test.py:33-60: one clip at a time) as one hipGraph per clip length:
    K1 features -> encoder + head (eval mode) -> AD-YOLO decode.  ``__call__(audio (B, n, 4))`` -> (logits, decoded or
    None); both are the graph's static outputs, valid until the next call with the same shape."""

    def __init__(self, model, features, postprocessor=None, warm_calls=1):
        self.model, self.features, self.post = model, features, postprocessor
        self.warm_calls = warm_calls
        self.entries, self.seen = collections.OrderedDict(), {}
        self.captures = self.replays = self.evictions = 0
        self._tensors, self._calls, self._tensors_epoch = None, 0, None
        self.epoch = self._stamp()
        self.primed = None

    def _stamp(self):
        """What the recorded graphs depend on besides the input shape: the parameter / buffer epoch of in-place kernels
        (``ops.PARAMS_EPOCH``), the version counters of every parameter and buffer (``p.copy_()``, a torch optimizer step or an
        EMA swap in evaluation mode bump these, not the epoch -- the eager caches honour them, a recorded graph would replay
        stale affines and packed filters; round 4, ADVICE) and the dispatch switch table (``ops.switch_stamp``: the algorithm, the
        F(4x4) thresholds and the persistent / narrow / 1-D switches -- a graph recorded before ``ops.reload_thresholds()`` moved
        one would keep the old kernel choice; round 5, ADVICE)."""
        # (the tensor list is cached: walking the module tree costs ~0.3 ms per call, a tenth of a one-clip forward.  It is rebuilt
        #  whenever ops.PARAMS_EPOCH moved since it was made -- .to(), load_state_dict(assign=True) and every in-place kernel bump
        #  the epoch, and after a replaced Parameter object later copy_() / optimizer writes land in the NEW objects, whose version
        #  counters a stale list would never see (round 5, ADVICE) -- and every 256 calls regardless)
        ts = self._tensors
        if ts is None or self._tensors_epoch != ops.PARAMS_EPOCH[0] or self._ntensors_check():
            ts = self._tensors = list(self.model.parameters()) + list(self.model.buffers())
            self._tensors_epoch = ops.PARAMS_EPOCH[0]
        ver = 0
        for t in ts:
            ver += t._version
        return (ops.PARAMS_EPOCH[0], ver, len(ts), ops.switch_stamp())

    def _ntensors_check(self):
        """every 256 calls the cached tensor list is rebuilt (a module that replaced a parameter object without moving the model)"""
        self._calls += 1
        return (self._calls & 255) == 0

    def _run(self, audio):
        out = self.model(self.features(audio, channels_last8=True), channels_last8=True)
        dec = None
        if self.post is not None:
            p = self.post
            dec = ops.yolo_decode(out.contiguous(), p.nb_classes, p.nb_grids, p.nb_anchors, p.grid_size, p.g_overlap)
        return out, dec

    def __call__(self, audio):
        if self.model.training:
            raise RuntimeError("ForwardGraphs records the evaluation forward: call model.eval() first")
        stamp = self._stamp()
        if self.epoch != stamp:                        # parameters / buffers were written since the graphs were recorded: the
            self.entries.clear()                      # evaluation-mode BatchNorm affines inside them are stale -- record again
            self.seen.clear()
            self.epoch = stamp
        key = tuple(audio.shape)
        ent = self.entries.get(key)
        with torch.no_grad():
            if ent is None:
                n = self.seen.get(key, 0)
                self.seen[key] = n + 1
                if n < self.warm_calls:
                    self.primed = stamp
                    return self._run(audio)
                if self.primed != stamp:               # nothing has run eagerly under these parameters yet: the packed filters
                    self._run(audio)                   # and evaluation affines are built on first use with host->device copies,
                    self.primed = stamp                # which a capturing stream refuses -- build them outside the capture
                static = audio.clone()
                graph = torch.cuda.CUDAGraph()
                with _quiet_collector(), torch.cuda.graph(graph):
                    outs = self._run(static)
                # the recorded kernels read the evaluation-mode BatchNorm affines (``_BNState.eval_affine``) by address: they were
                # built outside the capture and belong to the modules' caches -- hold a reference for as long as the graph lives,
                # so that a dropped or replaced cache entry cannot hand their memory to somebody else under a replay (round 5: a
                # test that pops the caches between replays read freed memory once the allocation pattern changed)
                keep = [m.__dict__["_adyolo_eval_affine"] for m in self.model.modules() if "_adyolo_eval_affine" in m.__dict__]
                ent = (graph, static, outs, keep)
                self.evictions += _lru_insert(self.entries, key, ent, MAX_GRAPHS)
                self.captures += 1
            else:
                self.entries.move_to_end(key)
            graph, static, outs = ent[0], ent[1], ent[2]
            if audio.data_ptr() != static.data_ptr():
                static.copy_(audio, non_blocking=True)
            graph.replay()
            self.replays += 1
            return outs
