"""FOA rotation augmentation.  Mirror of ``RotationAug`` (/root/reference/src/utils/augmentations.py:36-111): 16
combinations of channel sign flips / X-Y swap with the matching azimuth / elevation remapping of the labels.
``rotate_labels`` is the host half (label dict); ``rotate_audio`` applies the channel transform to a whole batch of raw
audio on the GPU (csrc/optim.hip ``foa_rotate_kernel``).  ``SpecAug`` (augmentations.py:6-33) masks the GPU feature
tensor (csrc/aug.hip); its random draw restates torchaudio 0.10, which is absent here (parity of the stream unpinned)."""
import random

import torch

from . import _lib
from .ops import _p, _stream

# (yzx sign weights, xy swap, azimuth weight, azimuth offset, elevation weight) -- augmentations.py:45-68
COMBINATIONS = [
    ((1, 1, 1), False, 1, 0, 1), ((1, -1, 1), False, 1, 0, -1),
    ((-1, 1, 1), False, -1, 0, 1), ((-1, -1, 1), False, -1, 0, -1),
    ((-1, 1, -1), False, 1, 180, 1), ((-1, -1, -1), False, 1, 180, -1),
    ((1, 1, -1), False, -1, 180, 1), ((1, -1, -1), False, -1, 180, -1),
    ((-1, 1, 1), True, 1, 90, 1), ((-1, -1, 1), True, 1, 90, -1),
    ((1, 1, 1), True, -1, 90, 1), ((1, -1, 1), True, -1, 90, -1),
    ((1, 1, -1), True, 1, -90, 1), ((1, -1, -1), True, 1, -90, -1),
    ((-1, 1, -1), True, -1, -90, 1), ((-1, -1, -1), True, -1, -90, -1),
]


def rotate_labels(label: dict, comb_no: int):
    """label {frame: [[cls, src, az, el], ...]} -> rotated copy (augmentations.py:98-109)."""
    _, _, pw, dpi, tw = COMBINATIONS[int(comb_no)]
    out = {}
    for frame, events in label.items():
        rows = []
        for ev in events:
            pi = ev[-2] * pw + dpi
            if pi < -180:
                pi += 360
            elif pi > 180:
                pi -= 360
            rows.append(list(ev[:-2]) + [pi, ev[-1] * tw])
        out[frame] = rows
    return out


def rotate_audio(audio, comb_nos):
    """audio (B, n_samples, 4) float32 on the GPU, comb_nos: B combination indices -> rotated audio (new tensor)."""
    if not audio.is_cuda or audio.dtype != torch.float32 or not audio.is_contiguous():
        raise _lib.AdyoloHipError("rotate_audio needs contiguous float32 audio (B, n_samples, 4) on the GPU")
    b, n, _ = audio.shape
    cfg = torch.tensor([[*COMBINATIONS[int(c)][0], float(COMBINATIONS[int(c)][1])] for c in comb_nos],
                       dtype=torch.float32).to(audio.device)
    out = torch.empty_like(audio)
    _lib.call("adyolo_foa_rotate", _p(audio), _p(out), _p(cfg), b, n, _stream())
    return out


class RotationAug:
    """Same surface as the reference class: ``augment(audio (T,4) on the GPU as (1,T,4) or labels only, label)``."""

    def __init__(self, params: dict, is_valid: bool):
        self.apply_augment = bool(params["aug_config"]["rotation_augment"]) and not is_valid

    def augment(self, audio, label, comb_no=None):
        if not self.apply_augment:
            return audio, label
        if comb_no is None:
            comb_no = int(random.uniform(0, 16))
        return rotate_audio(audio.view(1, -1, 4), [comb_no]).view_as(audio), rotate_labels(label, comb_no)


class SpecAug:
    """Feature-wise spec-augmentation (reference src/utils/augmentations.py:6-33) on the GPU feature tensor.

    The reference applies torchaudio 0.10's ``TimeMasking`` / ``FrequencyMasking`` to a (C, T, F) tensor, i.e. (quirk)
    its "time" mask (axis 2) zeroes MEL BINS and its "frequency" mask (axis 1) zeroes FRAMES; both with probability
    ``spec_augment_thresh`` per sample, identity when ``spec_augment`` is false (the default) or on validation data.
    ``mask_along_axis``: value = U(0,1) * mask_param, start = U(0,1) * (size - value), mask [int(start), int(start+value)).
    torchaudio is not installed in this image, so the draw order is restated from its published 0.10 source (parity of the
    random stream unpinned); the masking itself is exact and tested against NumPy slicing.
    ``augment(feat)``: feat (B, T, 64, 8) float32 channels-last on the GPU, masked in place per sample.
    """

    def __init__(self, params: dict, is_valid: bool):
        a = params.get("aug_config", {})
        self.apply_augment = bool(a.get("spec_augment", False)) and not is_valid
        self.thresh = a.get("spec_augment_thresh", 0.5)
        self.time_mask_param = a.get("spec_augment_time_mask_param", 0)
        self.freq_mask_param = a.get("spec_augment_freq_mask_param", 0)

    @staticmethod
    def _range(mask_param, size):
        value = random.random() * mask_param
        start = random.random() * (size - value)
        return int(start), int(start + value)

    def draw(self, batch, t, f):
        """-> int32 (batch, 4) host tensor {t0, t1, f0, f1}; empty ranges where a mask is not applied."""
        rng = torch.zeros((batch, 4), dtype=torch.int32)
        for b in range(batch):
            if random.random() <= self.thresh:                       # reference "time_masking": last axis = mel bins
                rng[b, 2], rng[b, 3] = self._range(self.time_mask_param, f)
            if random.random() <= self.thresh:                       # reference "frequency_masking": axis 1 = frames
                rng[b, 0], rng[b, 1] = self._range(self.freq_mask_param, t)
        return rng

    def augment(self, feat, ranges=None):
        if not self.apply_augment:
            return feat
        from . import ops
        b, t, f, _ = feat.shape
        if ranges is None:
            ranges = self.draw(b, t, f)
        return ops.mask_ranges_(feat, ranges.to(feat.device, non_blocking=True).contiguous())
