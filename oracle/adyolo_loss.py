"""Oracle (test infrastructure): AD-YOLO angular-distance responsibility-assignment loss,
PyTorch-CPU float32 with autograd (the gradient oracle is ``torch.autograd`` of this).

Restates ``/root/reference/src/models/loss.py:156-251`` (``ADYOLOloss``):
  * ``__init__`` :157-180   grid [8,4], grid_offset (i*45-180+22.5, j*45-90+22.5), gains, train_unify
  * decode     :193-213   sigmoid on [obj, cls x C], tanh on [u, v]; UV = tanh*(0.5+g_overlap)*grid+offset;
                           V clamp [-90, 90]; U >= 180 -> -360; U < -180 -> +360
  * distance   :182-187   great-circle distance in degrees, acos argument clipped to +-(1 - 1e-7)
  * assignment :222-229   per threshold: D < thr  OR  arg-min anchor of the target's cell
  * BCE terms  :231-239   class BCE over positive anchors, objectness BCE over positives / negatives
                           (``nn.BCELoss``: log clamped at -100)
  * total      :241-251   i == 0 adds angular_gain * mean(D[mask] / 180) over (target, anchor) pairs;
                           every i adds (object*pos + nonobj*neg + class*cls) / len(train_unify)
Pinned by ``tests/golden/adyolo_loss.npz`` (loss value and dlogits of the real reference for three cases; the distance
matrix D and the masks are not stored -- they are pinned indirectly through the loss and its gradient).
"""
import math
import torch

DEFAULT_GAINS = {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0, "class_gain": 3.0}


def grid_geometry(grid_size=(45.0, 45.0)):
    n_az = int(math.ceil(360.0 / grid_size[0]))
    n_el = int(math.ceil(180.0 / grid_size[1]))
    gs = torch.tensor([float(grid_size[0]), float(grid_size[1])])
    ii, jj = torch.meshgrid(torch.arange(n_az), torch.arange(n_el), indexing="ij")
    offset = torch.stack([ii, jj], dim=-1).float() * gs - torch.tensor([180.0, 90.0]) + gs * 0.5
    return n_az, n_el, gs, offset


def decode(logit, nb_classes, grid_size=(45.0, 45.0), nb_anchors=5, g_overlap=0.5):
    """loss.py:193-213 -> prob (B,T,Gi,Gj,A,C+1), uv (B,T,Gi,Gj,A,2) in degrees."""
    b, t, _ = logit.shape
    n_az, n_el, gs, offset = grid_geometry(grid_size)
    out = logit.reshape(b, t, n_az, n_el, nb_anchors, nb_classes + 3)
    prob = torch.sigmoid(out[..., :nb_classes + 1])
    uv = torch.tanh(out[..., nb_classes + 1:]) * (0.5 + g_overlap) * gs + offset[None, None, :, :, None, :]
    u = uv[..., 0]
    v = torch.clamp(uv[..., 1], -90.0, 90.0)
    u = torch.where(u >= 180.0, u - 360.0, u)
    u = torch.where(u < -180.0, u + 360.0, u)
    return prob, torch.stack([u, v], dim=-1)


def angular_distance_deg(uv_a, uv_b):
    """loss.py:182-187."""
    a, b = torch.deg2rad(uv_a), torch.deg2rad(uv_b)
    c = torch.sin(a[..., 1]) * torch.sin(b[..., 1]) + \
        torch.cos(a[..., 1]) * torch.cos(b[..., 1]) * torch.cos(torch.abs(a[..., 0] - b[..., 0]))
    return torch.rad2deg(torch.acos(torch.clip(c, -1 + 1e-7, 1 - 1e-7)))


def _bce_mean(p, y):
    """nn.BCELoss(reduction='mean') (loss.py:180): forward logs clamped at -100; backward is
    (p - y) / max(p (1 - p), 1e-12), which stays finite when the sigmoid saturates to exactly 0 or 1."""
    return torch.nn.functional.binary_cross_entropy(p, y, reduction="mean")


def adyolo_loss(logit, target, nb_classes, grid_size=(45.0, 45.0), nb_anchors=5, g_overlap=0.5,
                train_unify=(45.0, 25.0, 10.0), gains=None, return_aux=False):
    """logit (B,T,G*A*(C+3)); target (M,7) [b, frame, Gi, Gj, cls, U, V] -> loss tensor of shape (1,)."""
    gains = gains or DEFAULT_GAINS
    prob, uv = decode(logit, nb_classes, grid_size, nb_anchors, g_overlap)
    b, t, n_az, n_el, a, _ = prob.shape
    m = target.shape[0]
    tb, tt, gi, gj, tc = (target[:, k].long() for k in range(5))
    cell = ((tb * t + tt) * n_az + gi) * n_el + gj                        # (M,)
    uv_cell = uv.reshape(-1, a, 2)[cell]                                   # (M,A,2)
    dist = angular_distance_deg(uv_cell, target[:, None, 5:7].expand(m, a, 2))   # (M,A)
    nearest = dist.argmin(dim=1)
    flat_obj = prob[..., 0].reshape(-1)                                    # (B*T*G*A,)
    flat_cls = prob[..., 1:].reshape(-1, nb_classes)
    anchor_ids = cell[:, None] * a + torch.arange(a)[None, :]              # (M,A)

    total = torch.zeros(1)
    masks = []
    for i, thr in enumerate(train_unify):
        mask = dist < thr
        mask[torch.arange(m), nearest] = True
        masks.append(mask)
        pos = torch.zeros(flat_obj.shape[0], dtype=torch.bool)
        pos[anchor_ids[mask]] = True
        cls_t = torch.zeros(flat_obj.shape[0], nb_classes)
        cls_t[anchor_ids[mask], tc[:, None].expand(m, a)[mask]] = 1.0
        cls_term = _bce_mean(flat_cls[pos], cls_t[pos])
        pos_term = _bce_mean(flat_obj[pos], torch.ones(int(pos.sum())))
        neg_term = _bce_mean(flat_obj[~pos], torch.zeros(int((~pos).sum())))
        if i == 0:
            total = total + (dist[mask] / 180.0).mean() * gains["angular_gain"]
        total = total + (pos_term * gains["object_gain"] + neg_term * gains["nonobj_gain"]
                         + cls_term * gains["class_gain"]) / len(train_unify)
    if return_aux:
        return total, {"D": dist.detach(), "masks": torch.stack(masks, 0), "uv": uv.detach(),
                       "prob": prob.detach()}
    return total
