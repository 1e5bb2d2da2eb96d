"""Oracle (test infrastructure): the other plugin losses, heads and label encoders of the reference's --loss switch,
PyTorch-CPU float32 / NumPy.

Restates /root/reference/src/models/loss.py:32-54 (SEDDOAloss), :57-67 (ACCDOAloss), :70-153 (ADPITloss),
/root/reference/src/models/linearheads.py:26-86 (head activations) and /root/reference/src/datasets.py:296-455
(get_seddoa_label / get_accdoa_label / get_adpit_label, with utils/seld_metrics.py:50-65 for polar->Cartesian).
Pinned by tests/golden/other_losses.npz (generated from the real reference).
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F


def seddoa_loss(output, target, nb_classes, masked_mse=True):
    c = nb_classes
    sed = F.binary_cross_entropy(output[..., :c], target[..., :c])
    doa_out = output[..., c:]
    if masked_mse:
        doa_out = doa_out * target[..., :c].repeat(1, 1, 3)
    return sed + 1000.0 * F.mse_loss(doa_out, target[..., c:])


def accdoa_loss(output, target):
    return F.mse_loss(output, target)


def adpit_loss(output, target, nb_classes):
    """output (B,T,9C), target (B,T,6,4,C)."""
    b, t = output.shape[:2]
    v = target[:, :, :, 0:1, :] * target[:, :, :, 1:, :]               # (B,T,6,3,C): act * xyz
    a0, b0, b1, c0, c1, c2 = (v[:, :, i] for i in range(6))

    def cat3(x, y, z):
        return torch.cat([x, y, z], dim=2)                              # (B,T,9,C)
    aaa = cat3(a0, a0, a0)
    b_perms = [cat3(*p) for p in ((b0, b0, b1), (b0, b1, b0), (b0, b1, b1), (b1, b0, b0), (b1, b0, b1), (b1, b1, b0))]
    c_perms = [cat3(*p) for p in itertools.permutations((c0, c1, c2))]
    cands = [aaa + (b_perms[0] + c_perms[0])] + [p + (aaa + c_perms[0]) for p in b_perms] + \
            [p + (aaa + b_perms[0]) for p in c_perms]
    out = output.reshape(b, t, 9, nb_classes)
    losses = torch.stack([((out - cand) ** 2).mean(dim=2) for cand in cands], dim=0)   # (13,B,T,C)
    return losses.min(dim=0).values.mean()


def head_activation(raw, n_sigmoid_cols):
    return torch.cat([torch.sigmoid(raw[..., :n_sigmoid_cols]), torch.tanh(raw[..., n_sigmoid_cols:])], dim=-1)


def _xyz(az, el):
    e, a = el * np.pi / 180.0, az * np.pi / 180
    return np.cos(a) * np.cos(e), np.sin(a) * np.cos(e), np.sin(e)


def seddoa_label(label, nb_frames, nb_classes):
    """datasets.py:296-321 -> (T', 4C) [se | x | y | z]."""
    out = np.zeros((4, nb_frames, nb_classes))
    for frame, events in label.items():
        if frame < nb_frames:
            for ev in events:
                out[:, frame, ev[0]] = (1.0,) + _xyz(ev[2], ev[3])
    return np.concatenate(list(out), axis=1).astype(np.float32)


def accdoa_label(label, nb_frames, nb_classes):
    """datasets.py:323-348 -> (T', 3C)."""
    s = seddoa_label(label, nb_frames, nb_classes)
    c = nb_classes
    return (np.tile(s[:, :c], 3) * s[:, c:]).astype(np.float32)


def adpit_label(label, nb_frames, nb_classes):
    """datasets.py:350-455 -> (T', 6, 4, C)."""
    out = np.zeros((nb_frames, 6, 4, nb_classes), dtype=np.float64)
    for frame, events in label.items():
        if frame >= nb_frames:
            continue
        evs = sorted(events, key=lambda e: e[0])
        for cls in sorted(set(e[0] for e in evs)):
            same = [e for e in evs if e[0] == cls]
            base = {1: 0, 2: 1}.get(len(same), 3)
            for k, e in enumerate(same[:3 if base == 3 else len(same)]):
                out[frame, base + k, :, cls] = (1.0,) + _xyz(e[2], e[3])
    return out.astype(np.float32)
