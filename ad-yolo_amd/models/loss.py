"""AD-YOLO loss plugin.  Mirror of ``ADYOLOloss`` (/root/reference/src/models/loss.py:156-251): plain
callable built from the whole ``params`` dict, ``__call__(logit (B,T',K), target (M,7)) -> tensor (1,)``
supporting ``.backward()`` / ``.item()``.  Like the reference it owns its device placement
(``params['args']['device']``, loss.py:159,200) and moves the CPU target tensor there itself; unlike the
reference it makes no host round trips: one fused HIP pass (csrc/loss.hip)."""
import math

import torch

from .. import functional as Fn


class ADYOLOloss(object):
    def __init__(self, params: dict):
        self.device = torch.device(params["args"]["device"])
        self.nb_classes = params["data_config"]["nb_classes"]
        tc = params["train_config"]
        gs = [float(v) for v in tc["grid_size"]]
        self.cfg = {
            "nb_classes": self.nb_classes,
            "grid": (int(math.ceil(360.0 / gs[0])), int(math.ceil(180.0 / gs[1]))),
            "anchors": int(tc["nb_anchors"]),
            "thr": tuple(float(v) for v in tc["train_unify"]),
            "gains": (float(tc["loss_gains"]["angular_gain"]), float(tc["loss_gains"]["object_gain"]),
                      float(tc["loss_gains"]["nonobj_gain"]), float(tc["loss_gains"]["class_gain"])),
            "grid_size": tuple(gs),
            "g_overlap": float(tc["g_overlap"]),
        }
        if len(self.cfg["thr"]) != 3:
            raise NotImplementedError("adyolo loss kernel is built for 3 train_unify thresholds")

    def __call__(self, logit: torch.Tensor, target: torch.Tensor):
        if not logit.is_cuda:
            raise RuntimeError("ADYOLOloss (adyolo_amd) runs on MI355X only; logits must live on a HIP device")
        target = target.to(logit.device, torch.float32).contiguous()
        return Fn.ADYOLOLossFn.apply(logit.contiguous(), target, self.cfg)


def _need_gpu(t, who):
    if not t.is_cuda:
        raise RuntimeError("%s (adyolo_amd) runs on MI355X only; the network output must live on a HIP device" % who)


class SEDDOAloss(object):
    """Mirror of /root/reference/src/models/loss.py:32-54: BCE(sed) + 1000 * (masked) MSE(doa)."""

    def __init__(self, nb_classes, masked_mse=True):
        self.nb_classes = nb_classes
        self.masked_mse = masked_mse

    def __call__(self, output, target):
        _need_gpu(output, "SEDDOAloss")
        target = target.to(output.device, torch.float32).contiguous()
        cfg = {"nsed": self.nb_classes, "masked": int(self.masked_mse), "w_bce": 1.0, "w_mse": 1000.0}
        return Fn._FusedLossFn.apply(output, target, "seddoa", cfg)


class ACCDOAloss(object):
    """Mirror of loss.py:57-67: plain MSE."""

    def __init__(self, nb_classes):
        self.nb_classes = nb_classes

    def __call__(self, output, target):
        _need_gpu(output, "ACCDOAloss")
        target = target.to(output.device, torch.float32).contiguous()
        cfg = {"nsed": 0, "masked": 0, "w_bce": 0.0, "w_mse": 1.0}
        return Fn._FusedLossFn.apply(output, target, "accdoa", cfg)


class ADPITloss(object):
    """Mirror of loss.py:70-153 (multi-ACCDOA, 3 tracks): output (B,T,9*C), target (B,T,6,4,C)."""

    def __init__(self, nb_classes):
        self.nb_classes = nb_classes

    def __call__(self, output, target):
        _need_gpu(output, "ADPITloss")
        target = target.to(output.device, torch.float32).contiguous()
        return Fn._FusedLossFn.apply(output, target, "adpit", {"nb_classes": self.nb_classes})
