"""Oracle (test infrastructure): AD-YOLO inference decode, NumPy float32.
Restates the decode half of ``LabelPostProcessor.get_yolo_output`` (/root/reference/src/datasets.py:752-771):
sigmoid / tanh, grid offset, V clamp to [-90, 90 - 1e-7], U wrap, class-confidence = class prob * objectness.
Pinned (together with the product's host NMS) by tests/golden/postprocess.npz generated from the real reference."""
import numpy as np


def decode(logit, nb_classes, grid=(8, 4), anchors=5, grid_size=(45.0, 45.0), g_overlap=0.5):
    t = logit.shape[-2] if logit.ndim == 3 else logit.shape[0]
    x = np.asarray(logit, dtype=np.float32).reshape(t, grid[0], grid[1], anchors, nb_classes + 3)
    out = np.empty_like(x)
    sig = 1.0 / (1.0 + np.exp(-x[..., :nb_classes + 1].astype(np.float64)))
    out[..., 0] = sig[..., 0]
    out[..., 1:nb_classes + 1] = (sig[..., 1:] * sig[..., :1]).astype(np.float32)
    gi = np.arange(grid[0], dtype=np.float32)[None, :, None, None]
    gj = np.arange(grid[1], dtype=np.float32)[None, None, :, None]
    u = np.tanh(x[..., -2]) * np.float32(0.5 + g_overlap) * np.float32(grid_size[0]) + (gi * grid_size[0] - 180.0 + grid_size[0] * 0.5)
    v = np.tanh(x[..., -1]) * np.float32(0.5 + g_overlap) * np.float32(grid_size[1]) + (gj * grid_size[1] - 90.0 + grid_size[1] * 0.5)
    v = np.clip(v, -90.0, 90.0 - 1e-7)
    u = np.where(u >= 180.0, u - 360.0, u)
    u = np.where(u < -180.0, u + 360.0, u)
    out[..., -2], out[..., -1] = u, v
    return out.astype(np.float32)
