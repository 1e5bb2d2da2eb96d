"""Counter-based dropout streams of the encoders (the reference draws its dropout masks from torch's global generator,
src/models/backbones/resnet.py:153, resnet_conformer.py:46-47,199-206; its checkpoints restore that generator,
src/utils/utility.py:32-50).  A stream here is (seed, running offset): the seed follows ``torch.initial_seed()`` (what
``torch.manual_seed`` / the reference's ``seed_init`` set) and the data-parallel rank, so ranks draw different masks;
``state()`` / ``set_state()`` go into the checkpoint's ``rng_state`` so a resumed run replays the same masks."""
import os

import torch
import torch.distributed as dist

_MASK64 = (1 << 64) - 1


class DropoutStream:
    def __init__(self, salt):
        self.salt = int(salt)
        self.offset = 0
        self._base = None          # rank-independent: torch.initial_seed() mixed with the salt (what a checkpoint stores)
        self._seed = None          # the base with this process's data-parallel rank mixed in (what the kernels get)
        # hipGraph support (graph.py): while a step is being CAPTURED, draws are relative to `capture_base` and the kernels
        # add the device-side counter `dev` (which holds the stream's offset at the start of the step being replayed)
        self.dev = None
        self.dev_value = None      # what the host knows `dev` to hold
        self.capture_base = None
        self.host_draws = 0        # mask() calls: values computed on the host, which a recorded step cannot replay

    @staticmethod
    def _rank():
        return dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", "0"))

    @property
    def seed(self):
        """Bound at first use: torch.initial_seed() mixed with the stream's salt and the data-parallel rank."""
        if self._seed is None:
            if self._base is None:
                self._base = (int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + self.salt) & _MASK64
            self._seed = (self._base ^ ((self._rank() * 0xD1B54A32D192ED03) & _MASK64)) & _MASK64
        return self._seed

    def begin_capture(self, device):
        if self.dev is None or self.dev.device != torch.device(device):
            self.dev = torch.zeros(1, dtype=torch.int64, device=device)
            self.dev_value = 0
        self.capture_base = self.offset

    def end_capture(self):
        """-> the number of values one replay of the captured step draws; the host offset is put back (capturing ran nothing)."""
        delta = self.offset - self.capture_base
        self.offset, self.capture_base = self.capture_base, None
        return delta

    def sync_device(self):
        """Make the device counter hold the host offset (no-op when it already does; one async fill otherwise)."""
        if self.dev is not None and self.dev_value != self.offset:
            self.dev.fill_(self.offset - (1 << 64) if self.offset >= (1 << 63) else self.offset)
            self.dev_value = self.offset

    def replayed(self, delta):
        """A captured step was replayed: its last node advanced the device counter by ``delta``."""
        self.offset += int(delta)
        self.dev_value = self.offset

    def _no_capture(self, what):
        if self.capture_base is not None:
            raise NotImplementedError("DropoutStream.%s is not hipGraph-capturable (host-computed values); "
                                      "only draw() is" % what)

    def mask(self, like, p):
        from . import ops
        self._no_capture("mask")
        self.host_draws += 1
        m = ops.dropout_mask(like, p, self.seed, self.offset)
        self.offset += like.numel()
        return m

    def draw(self, n):
        """(seed, offset) for ``n`` values generated inside a kernel (``ops.dropout_apply``); advances the stream by n.
        The values are the ones ``mask`` would have written."""
        if self.capture_base is not None:
            key = (self.seed, self.offset - self.capture_base, self.dev)
        else:
            key = (self.seed, self.offset)
        self.offset += int(n)
        return key

    def seed32(self, n):
        """A 32-bit seed for a kernel that draws ``n`` values from its own stateless hash; advances the stream by n.
        Eager: an int computed here.  While a step is being captured (round 4): a one-element device tensor that
        ``ops.seed32_dev`` fills with the same value derived from the device-side counter, so a replay draws what the eager
        step at that offset would."""
        if self.capture_base is not None:
            from . import ops
            t = ops.seed32_dev(self.seed, self.offset - self.capture_base, self.dev)
            self.offset += int(n)
            return t
        x = (self.seed ^ ((self.offset * 0x9E3779B97F4A7C15) & _MASK64)) & _MASK64
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & _MASK64
        self.offset += int(n)
        return int((x ^ (x >> 31)) & 0xFFFFFFFF)

    def state(self):
        """What goes into the checkpoint's rng_state: the RANK-INDEPENDENT base seed and the offset (the one rank-0
        checkpoint is restored on every rank; each re-applies its own rank, so resumed ranks keep drawing different masks).
        A stream restored from a round-2 checkpoint (which stored the rank-mixed seed of the rank that wrote it) has no base
        yet: it is recovered by taking this process's rank back out (round 4, ADVICE: it used to write base_seed = None, which
        the next resume could not load)."""
        if self._base is None and self._seed is not None:
            self._base = (self._seed ^ ((self._rank() * 0xD1B54A32D192ED03) & _MASK64)) & _MASK64
        _ = self.seed
        return {"base_seed": self._base, "offset": self.offset}

    def set_state(self, st):
        self.offset = int(st["offset"])
        if st.get("base_seed") is not None:
            self._base, self._seed = int(st["base_seed"]), None          # rank mixed in again at the next use
        else:
            # round-2 checkpoints stored the rank-mixed seed of the WRITING rank (rank 0: the mix is the identity there, so the
            # stored value is the base); treat it as the base so that every resumed rank mixes its own rank in again
            self._base, self._seed = int(st["seed"]), None


def streams(model):
    """Every DropoutStream attribute in ``model``, in module order."""
    return [val for _, mod in model.named_modules() for val in vars(mod).values() if isinstance(val, DropoutStream)]


def collect(model):
    """{module path: stream state} of every DropoutStream attribute in ``model`` (for the checkpoint's rng_state)."""
    out = {}
    for name, mod in model.named_modules():
        for attr, val in vars(mod).items():
            if isinstance(val, DropoutStream):
                out["%s.%s" % (name, attr)] = val.state()
    return out


def restore(model, states):
    for name, mod in model.named_modules():
        for attr, val in vars(mod).items():
            key = "%s.%s" % (name, attr)
            if isinstance(val, DropoutStream) and key in states:
                val.set_state(states[key])
