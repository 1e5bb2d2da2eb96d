"""Oracle (test infrastructure): AD-YOLO label encoder and batch collation.

Restates ``/root/reference/src/datasets.py:219-238`` (grid constants),
``:457-482`` (``get_yolo_label``) and ``:164-184`` (``collate_fn``).
Pinned by ``tests/golden/labels.npz`` (generated from the reference itself).
"""
import math
import numpy as np


class YoloGrid:
    """Overlapping azimuth/elevation grid, datasets.py:219-238."""

    def __init__(self, grid_size=(45, 45), g_overlap=0.5):
        gs = np.asarray(grid_size, dtype=np.float64)
        self.n_az = int(math.ceil(360.0 / gs[0]))
        self.n_el = int(math.ceil(180.0 / gs[1]))
        ii, jj = np.meshgrid(np.arange(self.n_az), np.arange(self.n_el), indexing="ij")
        centre = np.stack([ii, jj], axis=-1) * gs - np.array([180.0, 90.0]) + gs * 0.5
        half = gs * (0.5 + g_overlap)
        self.centre = centre
        self.lb = centre - half
        self.ub = centre + half
        self.lb[..., 1] = np.clip(self.lb[..., 1], -90, 90)
        self.ub[..., 1] = np.clip(self.ub[..., 1], -90, 90)

    def cells(self, az, el):
        """Responsible cells of one event (datasets.py:470-478); returns (az_used, [(gi, gj)...])."""
        if az == 180:
            az = -180.0
        el_ok = (self.lb[..., 1] <= el) & (el < self.ub[..., 1])
        resp = (self.lb[..., 0] <= az) & (az < self.ub[..., 0]) & el_ok
        resp |= (az + 360 < self.ub[..., 0]) & el_ok
        resp |= (self.lb[..., 0] < az - 360) & el_ok
        gi, gj = np.where(resp)
        return az, list(zip(gi.tolist(), gj.tolist()))


def yolo_label(label, nb_label_frames, grid=None):
    """datasets.py:457-482.  label: {frame: [[cls, src, az, el], ...]} -> rows [frame,Gi,Gj,cls,U,V]."""
    grid = grid or YoloGrid()
    rows = []
    for frame, events in label.items():
        if frame >= nb_label_frames:
            continue
        for ev in events:
            az, cells = grid.cells(ev[2], ev[3])
            for gi, gj in cells:
                rows.append([frame, gi, gj, ev[0], az, ev[3]])
    return rows


def collate(feats, labels):
    """datasets.py:164-184.  -> feat (B,C,T,F) float32, target (M,7) float32 [b,frame,Gi,Gj,cls,U,V].

    Like the reference, raises when every sample has an empty label list
    (``torch.cat([])`` at datasets.py:184).
    """
    parts = []
    for b, rows in enumerate(labels):
        if len(rows) == 0:
            continue
        r = np.asarray(rows, dtype=np.float32).reshape(len(rows), 6)
        parts.append(np.concatenate([np.full((len(rows), 1), b, dtype=np.float32), r], axis=1))
    if not parts:
        raise RuntimeError("collate: every sample of the batch has an empty label list")
    return np.stack([np.asarray(f, dtype=np.float32) for f in feats], 0), np.concatenate(parts, 0)
