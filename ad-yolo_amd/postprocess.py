"""Inference post-processing of the AD-YOLO output: decode (GPU kernel) -> confidence / class thresholds -> per-class
connectivity soft-merge NMS (host) -> {frame: [[class, x, y, z], ...]} and the DCASE CSV writer.

Mirror of ``LabelPostProcessor`` for ``--loss adyolo`` (/root/reference/src/datasets.py:485-534, ``get_yolo_output``
:741-855, helpers :858-919) and ``write_seld_output_file`` (/root/reference/src/test.py:26-30).  The decode is the same
arithmetic as the loss (csrc/loss.hip ``yolo_decode_kernel``); the NMS is tiny, data dependent and stays on the host
(NumPy float32), exactly where the reference runs it (``postprocessor.postprocess(output.detach().cpu())``, test.py:52).
Like the reference it handles one clip at a time (B = 1, datasets.py:752-753).
"""
import math

import numpy as np

F32 = np.float32


def _ang_dist_deg(a, b):
    """datasets.py:858-871 (acos argument clipped to [-1, 1])."""
    a, b = np.deg2rad(a).astype(F32), np.deg2rad(b).astype(F32)
    d = np.sin(a[..., 1]) * np.sin(b[..., 1]) + np.cos(a[..., 1]) * np.cos(b[..., 1]) * np.cos(np.abs(a[..., 0] - b[..., 0]))
    return np.rad2deg(np.arccos(np.clip(d, -1, 1))).astype(F32)


def _to_xyz(uv):
    r = np.deg2rad(uv).astype(F32)
    return np.stack([np.cos(r[:, 0]) * np.cos(r[:, 1]), np.sin(r[:, 0]) * np.cos(r[:, 1]), np.sin(r[:, 1])], axis=1).astype(F32)


def _single(rows):
    """datasets.py:874-890: rows [cls, conf, U, V] -> [cls, x, y, z]."""
    return np.concatenate([rows[:, :1], _to_xyz(rows[:, 2:4])], axis=1)


def _voted(rows, thresh):
    """datasets.py:893-919: confidence-weighted vote of one cluster (weights = softmax(exp(conf^2 / thresh)))."""
    e = np.exp(rows[:, 1].astype(F32) ** 2 / F32(thresh)).astype(F32)
    w = np.exp(e - e.max())
    w = (w / w.sum()).astype(F32)
    v = (_to_xyz(rows[:, 2:4]) * w[:, None]).sum(axis=0, keepdims=True)
    v = v / np.sqrt((v ** 2).sum())
    return np.concatenate([rows[:1, :1], v.astype(F32)], axis=1)


def nms_frame(det, nms, unify_thresh, clss_thresh):
    """det: (K, 4) [class, class_conf, U, V] sorted by descending class_conf -> list of [class, x, y, z]."""
    out = []
    for cls in np.unique(det[:, 0]):
        rows = det[det[:, 0] == cls]
        if len(rows) == 1:
            out.append(_single(rows))
            continue
        if nms == "conn-merge":
            dist = _ang_dist_deg(rows[None, :, 2:4].repeat(len(rows), 0), rows[:, None, 2:4].repeat(len(rows), 1))
            ref = dist < unify_thresh
            while rows.shape[0]:
                prev = np.zeros(len(rows), dtype=bool)
                cur = ref[0].copy()
                while not (prev == cur).all():
                    if cur.sum() == 1:
                        break
                    prev = cur.copy()
                    cur |= ref[cur].sum(axis=0).astype(bool)
                out.append(_voted(rows[cur], clss_thresh))
                rows = rows[~cur]
                ref = ref[~cur][:, ~cur]
        elif nms == "soft-merge":
            reference = rows.copy()
            while rows.shape[0]:
                d = _ang_dist_deg(rows[:1, 2:4], reference[:, 2:4])
                out.append(_voted(reference[d <= unify_thresh], clss_thresh))
                if len(rows) == 1:
                    break
                d = _ang_dist_deg(rows[:1, 2:4], rows[1:, 2:4])
                rows = rows[1:][d > unify_thresh]
        else:
            while rows.shape[0]:
                out.append(_single(rows[:1]))
                if len(rows) == 1:
                    break
                d = _ang_dist_deg(rows[:1, 2:4], rows[1:, 2:4])
                rows = rows[1:][d > unify_thresh]
    return np.concatenate(out, axis=0).tolist() if out else []


def nms_decoded(decoded, nb_classes, conf_thresh, clss_thresh, unify_thresh, nms="conn-merge"):
    """decoded: (T, Gaz, Gel, A, C+3) float32 [conf, class_conf x C, U, V] -> {frame: [[class, x, y, z], ...]}."""
    t = decoded.shape[0]
    flat = np.asarray(decoded, dtype=F32).reshape(t, -1, nb_classes + 3)
    out = {}
    for frame in range(t):
        fo = flat[frame]
        fo = fo[fo[:, 0] > conf_thresh]
        if len(fo) == 0:
            continue
        i, j = np.nonzero(fo[:, 1:nb_classes + 1] > clss_thresh)
        det = np.concatenate([j.astype(F32)[:, None], fo[:, 1:nb_classes + 1][i, j][:, None], fo[i, -2:]], axis=1)
        det = det[np.argsort(-det[:, 1], kind="stable")]
        res = nms_frame(det, nms, unify_thresh, clss_thresh)
        if len(res):
            out[frame] = res
    return out


class LabelPostProcessor:
    """``LabelPostProcessor(params).postprocess(output)`` for the adyolo head; ``output`` (1, T', K) logits on the GPU."""

    def __init__(self, params):
        tc = params["train_config"]
        self.nb_classes = params["data_config"]["nb_classes"]
        self.loss = params["args"]["loss"]
        if self.loss != "adyolo":
            raise NotImplementedError("postprocess: {} (only the adyolo decode + NMS is built)".format(self.loss))
        self.grid_size = [float(v) for v in tc["grid_size"]]
        self.nb_anchors = int(tc["nb_anchors"])
        self.nb_grids = (int(math.ceil(360.0 / self.grid_size[0])), int(math.ceil(180.0 / self.grid_size[1])))
        self.conf_thresh = tc["conf_thresh"]
        self.clss_thresh = tc["clss_thresh"]
        self.unify_thresh = tc["unify_thresh"]
        self.g_overlap = tc["g_overlap"]
        self.nms = tc["nms"]

    def get_conf_thresh(self):
        return self.conf_thresh

    def set_conf_thresh(self, thresh):          # datasets.py:532-534 rewrites both thresholds
        self.conf_thresh = thresh
        self.clss_thresh = thresh

    def decode(self, output, borrow=False):
        """GPU half of ``postprocess`` (threshold-free): logits (1, T', K) -> decoded predictions as a host array.
        borrow: return a view of the page-locked staging buffer (valid until the next decode of that shape) instead of a copy."""
        from . import ops
        if output.shape[0] != 1:
            raise ValueError("postprocess handles one clip at a time (B = 1), like the reference (datasets.py:752-753)")
        dec = ops.yolo_decode(output.contiguous(), self.nb_classes, self.nb_grids, self.nb_anchors, self.grid_size,
                              self.g_overlap)
        host = ops.to_host(dec).numpy()
        return host if borrow else host.copy()

    def select(self, decoded):
        """Host half: confidence / class thresholds + conn-merge NMS on a ``decode`` result."""
        return nms_decoded(decoded, self.nb_classes, self.conf_thresh, self.clss_thresh, self.unify_thresh, self.nms)

    def postprocess(self, output):
        return self.select(self.decode(output, borrow=True))       # consumed at once: no second host copy


def write_seld_output_file(file_pth, output: dict):
    """reference test.py:26-30: rows ``frame,class,0,x,y,z``."""
    with open(file_pth, "w") as f:
        for frame_idx in output.keys():
            for [class_idx, x, y, z] in output[frame_idx]:
                f.write("{},{},{},{},{},{}\n".format(int(frame_idx), int(class_idx), 0, float(x), float(y), float(z)))
