"""Train step of the AD-YOLO hot path: features (K1) -> encoder+head forward -> AD-YOLO loss ->
backward -> [bucketed RCCL all-reduce] -> fused Adam.  Mirror of ``get_optimizers`` / ``train_one_epoch``
(/root/reference/src/train.py:29-62) with the data-parallel layer the reference lacks.

FusedAdam keeps torch.optim.Adam's hyper-parameters and produces a ``state_dict`` in the stock Adam
format (train.py:149,236 save/restore it) while running one HIP launch over the flat parameter buffer.
"""
import os

import torch

from . import functional as Fn
from . import ops
from .dist import BucketedAllReduce, FlatParameters


class FusedAdam:
    """Adam over a FlatParameters buffer (csrc/optim.hip); lr 1e-3, betas (0.9,0.999), eps 1e-8, wd 0 by
    default like the reference config (src/configs/hyp_train.yaml:7-9)."""

    def __init__(self, flat: FlatParameters, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.flat = flat
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        # the step counter lives ON THE DEVICE (incremented by the Adam launch itself), so that no kernel argument changes
        # from step to step and a whole train step can be replayed from a hipGraph; `step_count` is the host's mirror of
        # it (checkpoints, torch.optim.Adam's state_dict layout)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=flat.flat.device)
        self.bc_dev = torch.zeros(2, dtype=torch.float32, device=flat.flat.device)
        self._step_count = 0
        self._dev_step_value = 0

    @property
    def step_count(self):
        return self._step_count

    @step_count.setter
    def step_count(self, v):
        self._step_count = int(v)

    def sync_device_step(self):
        if self._dev_step_value != self._step_count:
            self.step_dev.fill_(self._step_count)
            self._dev_step_value = self._step_count

    def replayed(self):
        """A captured step (graph.py) was replayed: the Adam launch inside it advanced the device counter."""
        self._step_count += 1
        self._dev_step_value = self._step_count
        ops.params_changed()

    def zero_grad(self, set_to_none=False):
        self.flat.zero_grad()

    def step(self, grad_scale=1.0):
        self.sync_device_step()
        ops.adam_step_dev(self.flat.flat, self.flat.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_dev, self.bc_dev,
                          self.lr, self.betas, self.eps, self.weight_decay, grad_scale)
        self._step_count += 1
        self._dev_step_value = self._step_count
        ops.params_changed()

    def state_dict(self):
        """torch.optim.Adam's layout with parameter indices in ``model.parameters()`` order -- the ONE format this build
        reads and writes (``checkpoint.optimizer_state_dict``), interchangeable with the reference's
        ``optimizer.state_dict()`` / ``load_state_dict`` (train.py:149, 236)."""
        from . import checkpoint
        return checkpoint.optimizer_state_dict(self, None)

    def load_state_dict(self, sd):
        from . import checkpoint
        checkpoint.load_optimizer_state_dict(self, None, sd)


def get_optimizers(params: dict, flat: FlatParameters):
    """reference train.py:29-37 (only Adam is on the gfx950 path)."""
    tc = params["train_config"]
    if tc.get("optim", "Adam") == "Adam":
        return FusedAdam(flat, lr=tc.get("lr", 1e-3), weight_decay=tc.get("weight_decay", 0.0))
    raise NotImplementedError(tc["optim"])


class TrainStep:
    """One data-parallel optimisation step on raw audio.

    step(audio (B, n_samples, 4) float32 on the GPU, target (M,7)) -> loss tensor (1,) on the device (no host sync).


    graph=True (default: the ADYOLO_GRAPH environment variable, off when unset): the whole step is recorded ONCE per input
    shape in a hipGraph and replayed (``graph.StepGraphs``) -- ~770 kernel launches per step leave the Python / ctypes /
    autograd path, which is what bounds the reference's own shapes (16 x 20 s chunks: the GPU work of a step is shorter than
    its launch sequence).  The first step at a new shape runs eagerly, the second is captured; results are bit-identical to
    the eager path.  Single-process only (under data parallelism the RCCL hooks stay eager).
    """

    def __init__(self, model, criterion, feature_extractor, params=None, n_buckets=4, lr=1e-3, graph=None, exact=None):
        self.model, self.criterion, self.features = model, criterion, feature_extractor
        self.flat = FlatParameters(model)
        self.flat.broadcast(0)                   # no-op on one rank: all ranks start from rank 0's parameters / buffers
        self.optimizer = get_optimizers(params, self.flat) if params is not None else FusedAdam(self.flat, lr=lr)
        self.reducer = BucketedAllReduce(self.flat, n_buckets=n_buckets)
        # exact=True (default: ADYOLO_DP_EXACT=1): batch statistics and loss normalisers over the batch of ALL ranks
        # (ops.ExactDP) -- N ranks on N equal shards == one device on the concatenated batch; gradients are summed
        self.exact = (os.environ.get("ADYOLO_DP_EXACT", "0") == "1") if exact is None else bool(exact)
        if graph is None:
            graph = os.environ.get("ADYOLO_GRAPH", "0") == "1"
        self.graphs = None
        if graph and not self.reducer.active:
            from .graph import StepGraphs
            self.graphs = StepGraphs(self)

    def step(self, audio, target):
        if self.graphs is not None:
            return self.graphs.step(audio, target)
        return self.step_eager(audio, target)

    def step_eager(self, audio, target):
        self.model.train()
        exact = self.exact and ops.EXACT.enable(self.reducer.group)
        try:
            feat = self.features(audio, channels_last8=True)
            output = self.model(feat, channels_last8=True)
            self.optimizer.zero_grad()
            loss = self.criterion(output, target)
            Fn.SINK.begin(self.flat, self.reducer)       # weight / BatchNorm / SE gradients go straight into the flat buffer
            try:
                loss.backward()
            finally:
                Fn.SINK.end()
        finally:
            ops.EXACT.disable()
        scale = self.reducer.finish()
        # exact + AD-YOLO: the loss kernel normalises by the counts of ALL ranks, so the ranks' gradients ADD UP to the batch's.
        # The class-wise losses (seddoa / masked-seddoa / accdoa / adpit: means over the local rows, ops.*_loss) keep their
        # local normaliser: with equal shards the batch mean is the mean of the ranks' means, i.e. gradients are AVERAGED and
        # the reported loss is the ranks' average (round 4, ADVICE: they used to be summed -- world times too large)
        summed = exact and getattr(self.criterion, "loss_nm", "adyolo") == "adyolo"
        self.optimizer.step(grad_scale=1.0 if summed else scale)
        if exact and not summed:
            loss = ops.EXACT_world_mean(loss.detach().clone(), self.reducer.group)
        return loss.detach()


def train_one_epoch(params: dict, dataloader, model, optimizer, criterion, device):
    """Mirror of ``train_one_epoch`` (/root/reference/src/train.py:40-62) for loaders that yield pre-computed features:
    ``for feat (B,7,T,64), label in dataloader`` -> forward, zero_grad, loss, backward, step; returns the mean loss.
    Works with ``torch.optim.Adam`` or ``FusedAdam`` (``TrainStep`` is the faster raw-audio + flat-buffer path).
    The per-step ``loss.item()`` of the reference (a device sync every iteration, train.py:57) is replaced by one
    device-side accumulation and a single sync at the end of the epoch."""
    model.train()
    total = None
    n = 0
    for i, (feat, label) in enumerate(dataloader):
        feat = feat.to(device).float()
        output = model(feat)
        optimizer.zero_grad()
        loss = criterion(output, label)
        loss.backward()
        optimizer.step()
        total = loss.detach().reshape(-1)[:1].clone() if total is None else total + loss.detach().reshape(-1)[:1]
        n = i + 1
        if params.get("args", {}).get("quick_test") and i == 4:
            break
    return float(total) / max(n, 1) if total is not None else 0.0


def train_one_epoch_audio(params: dict, dataloader, trainer, stager=None, rotate=True):
    """The raw-audio epoch: ``dataloader`` yields ``audio_collate_fn`` batches (pcm int16 (B,T,4), comb_nos, target); each is
    staged to the GPU (int16 over PCIe on a side stream, double-buffered), converted, rotated and handed to
    ``TrainStep.step`` (features + forward + loss + backward [+ all-reduce] + Adam).  Returns the mean loss with ONE
    device sync at the end of the epoch (the reference syncs every iteration, train.py:57)."""
    from .augmentations import rotate_audio
    from .datasets import AudioStager
    total, n = None, 0
    it = iter(dataloader)
    try:
        pcm, combs, target = next(it)
    except StopIteration:
        return 0.0
    if stager is None:
        stager = AudioStager(pcm.shape[0], pcm.shape[1], trainer.flat.flat.device)
    stager.stage(pcm)
    while True:
        audio = stager.get()
        cur_combs, cur_target = combs, target
        nxt = next(it, None)
        if nxt is not None:                       # the next batch crosses PCIe while this step runs
            pcm, combs, target = nxt
            stager.stage(pcm)
        if rotate and any(int(c) != 0 for c in cur_combs):
            audio = rotate_audio(audio, cur_combs)
        loss = trainer.step(audio, cur_target)
        total = loss.detach().reshape(-1)[:1].clone() if total is None else total + loss.detach().reshape(-1)[:1]
        n += 1
        if nxt is None or (params.get("args", {}).get("quick_test") and n == 5):
            break
    return float(total) / max(n, 1)
