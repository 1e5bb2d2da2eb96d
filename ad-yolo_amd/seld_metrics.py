"""DCASE SELD evaluation (ER / F / LE / LR / SELD score), host NumPy.

Mirror of ``ComputeSELDResults`` / ``SELDMetrics`` (/root/reference/src/utils/seld_metrics.py:188-519; itself adapted
from the DCASE challenge baseline, README.md:156): location-sensitive detection (20 degree threshold) and
class-sensitive localisation over 1-second segments with Hungarian track association, macro-averaged over classes.
This is the CPU evaluator that turns the CSV files written by ``test_epoch`` into the scores the reference prints; it is
host logic on tiny data, restated here so that SELD parity can be checked
without the reference.  Reference quirks kept: the number of frames of a recording is ``max(frame index)`` of its
reference CSV (seld_metrics.py:398); a class matched in no common frame books ``nb_pred`` (not ``nb_ref``) false
negatives (:339-342); LE of a class without matches is 180.
"""
import math
import os

import numpy as np
from scipy.optimize import linear_sum_assignment

EPS = np.finfo(float).eps


def load_output_format_file(path):
    """seld_metrics.py:13-33: rows ``frame,class,source,az,el`` (polar) or ``frame,class,source,x,y,z``."""
    out = {}
    with open(path, "r") as f:
        for line in f:
            w = line.strip().split(",")
            if len(w) < 5:
                continue
            out.setdefault(int(w[0]), []).append([int(w[1]), int(w[2])] + [float(v) for v in w[3:]])
    return out


def cartesian_to_polar(d):
    """seld_metrics.py:67-80."""
    out = {}
    for frame, vals in d.items():
        out[frame] = [[v[0], v[1], math.atan2(v[3], v[2]) * 180 / math.pi,
                       math.atan2(v[4], math.sqrt(v[2] ** 2 + v[3] ** 2)) * 180 / math.pi] for v in vals]
    return out


def _great_circle_deg(az1, el1, az2, el2):
    d = np.sin(el1) * np.sin(el2) + np.cos(el1) * np.cos(el2) * np.cos(np.abs(az1 - az2))
    return np.arccos(np.clip(d, -1, 1)) * 180 / np.pi


def segment(labels, max_frames, frames_per_block):
    """seld_metrics.py:478-519 -> {block: {class: {frame_in_block: [[src, az, el], ...]}}} (insertion ordered)."""
    nb_blocks = int(math.ceil(max_frames / float(frames_per_block)))
    out = {b: {} for b in range(nb_blocks)}
    for start in range(0, max_frames, frames_per_block):
        blk = out[start // frames_per_block]
        for fr in range(start, start + frames_per_block):
            for v in labels.get(fr, ()):
                blk.setdefault(v[0], {}).setdefault(fr - start, []).append(v[1:])
    return out


class SELDScorer:
    def __init__(self, nb_classes, doa_threshold=20.0):
        c = nb_classes
        self.nb_classes, self.thr = c, doa_threshold
        self.TP, self.FP, self.FP_spatial, self.FN, self.Nref = (np.zeros(c) for _ in range(5))
        self.total_DE, self.DE_TP, self.DE_FP, self.DE_FN = (np.zeros(c) for _ in range(4))
        self.S = self.D = self.I = 0

    def update(self, pred, ref):
        """pred / ref: outputs of ``segment`` for one recording (seld_metrics.py:289-395)."""
        for blk in range(len(ref)):
            loc_fn = loc_fp = 0
            pb, rb = pred.get(blk, {}), ref[blk]
            for c in range(self.nb_classes):
                n_ref = max(len(v) for v in rb[c].values()) if c in rb else None
                n_pred = max(len(v) for v in pb[c].values()) if c in pb else None
                if n_ref is not None:
                    self.Nref[c] += n_ref
                if c in rb and c in pb:
                    tracks = {}
                    for fr, ref_vals in rb[c].items():
                        if fr not in pb[c]:
                            continue
                        g = np.asarray(ref_vals, dtype=float)[:, 1:] * np.pi / 180.0
                        p = np.asarray(pb[c][fr], dtype=float)[:, 1:] * np.pi / 180.0
                        cost = _great_circle_deg(g[:, None, 0], g[:, None, 1], p[None, :, 0], p[None, :, 1])
                        ri, ci = linear_sum_assignment(cost)
                        for r, cc in zip(ri, ci):
                            tracks.setdefault(int(r), []).append(cost[r, cc])
                    if not tracks:
                        loc_fn += n_pred
                        self.FN[c] += n_pred
                        self.DE_FN[c] += n_pred
                    else:
                        for dists in tracks.values():
                            avg = sum(dists) / len(dists)
                            self.total_DE[c] += avg
                            self.DE_TP[c] += 1
                            if avg <= self.thr:
                                self.TP[c] += 1
                            else:
                                loc_fp += 1
                                self.FP_spatial[c] += 1
                        if n_pred > n_ref:
                            loc_fp += n_pred - n_ref
                            self.FP[c] += n_pred - n_ref
                            self.DE_FP[c] += n_pred - n_ref
                        elif n_pred < n_ref:
                            loc_fn += n_ref - n_pred
                            self.FN[c] += n_ref - n_pred
                            self.DE_FN[c] += n_ref - n_pred
                elif c in rb:
                    loc_fn += n_ref
                    self.FN[c] += n_ref
                    self.DE_FN[c] += n_ref
                elif c in pb:
                    loc_fp += n_pred
                    self.FP[c] += n_pred
                    self.DE_FP[c] += n_pred
            self.S += min(loc_fp, loc_fn)
            self.D += max(0, loc_fn - loc_fp)
            self.I += max(0, loc_fp - loc_fn)

    def scores(self):
        """Macro average (seld_metrics.py:260-287) -> ER, F, LE, LR, SELD, classwise (5, C)."""
        er = (self.S + self.D + self.I) / (self.Nref.sum() + EPS)
        f = self.TP / (EPS + self.TP + self.FP_spatial + 0.5 * (self.FP + self.FN))
        le = self.total_DE / (self.DE_TP + EPS)
        le[self.DE_TP == 0] = 180.0
        lr = self.DE_TP / (EPS + self.DE_TP + self.DE_FN)
        er_c = np.repeat(er, self.nb_classes)
        seld = np.mean([er_c, 1 - f, le / 180, 1 - lr], 0)
        classwise = np.array([er_c, f, le, lr, seld])
        return er, f.mean(), le.mean(), lr.mean(), seld.mean(), classwise


def jackknife_estimation(global_value, partial_estimates, significance_level=0.05):
    """seld_metrics.py:149-186: bias-corrected jackknife estimate, bias, standard error and the t-test confidence interval
    of a statistic from its leave-one-out estimates."""
    from scipy import stats
    partial_estimates = np.asarray(partial_estimates, dtype=float)
    mean_jack = np.mean(partial_estimates)
    n = len(partial_estimates)
    bias = (n - 1) * (mean_jack - global_value)
    std_err = np.sqrt((n - 1) * np.mean((partial_estimates - mean_jack) * (partial_estimates - mean_jack), axis=0))
    estimate = global_value - bias
    if not (0 < significance_level < 1):
        raise ValueError("confidence level must be in (0, 1).")
    t_value = stats.t.ppf(1 - significance_level / 2, n - 1)
    return estimate, bias, std_err, estimate + t_value * np.array((-std_err, std_err))


class ComputeSELDResults(object):
    """``ComputeSELDResults(params, ref_files_folder).get_SELD_Results(pred_files_path[, is_jackknife])`` like the reference
    (seld_metrics.py:374-476)."""

    def __init__(self, params, ref_files_folder=None):
        dc = params["data_config"]
        self._nb_classes = dc["nb_classes"]
        self._fpb = int(dc["sr"] / float(int(dc["sr"] * dc["label_hop_len_s"])))
        self._ref = {}
        for name in os.listdir(ref_files_folder):
            gt = load_output_format_file(os.path.join(ref_files_folder, name))
            nb = max(list(gt.keys()))
            self._ref[name] = (segment(gt, nb, self._fpb), nb)

    def _pred_labels(self, pred_files_path, name):
        """Segmented predictions of one file, or None when the file does not take part in this evaluation."""
        pred = cartesian_to_polar(load_output_format_file(os.path.join(pred_files_path, name)))
        return segment(pred, self._ref[name][1], self._fpb)

    def get_SELD_Results(self, pred_files_path, is_jackknife=False):
        scorer = SELDScorer(self._nb_classes, 20.0)
        kept = {}
        for name in os.listdir(pred_files_path):
            labels = self._pred_labels(pred_files_path, name)
            if labels is None:
                continue
            scorer.update(labels, self._ref[name][0])
            kept[name] = labels
        out = scorer.scores()
        if not is_jackknife:
            return out
        return self._jackknife(out, kept)

    def _jackknife(self, global_scores, kept):
        """Leave-one-file-out confidence intervals (seld_metrics.py:441-476 / :640-676).  Reference quirk kept: the point
        values returned next to the intervals are those of the LAST leave-one-out pass (the loop reuses the variable
        names of the global scores), not the global scores; the intervals themselves are built around the global ones."""
        er, f, le, lr, seld, cw = global_scores
        global_values = [er, f, le, lr, seld] + cw.reshape(-1).tolist()
        partial, last = [], global_scores
        names = list(kept.keys())
        for leave in names:
            scorer = SELDScorer(self._nb_classes, 20.0)
            for name in names:
                if name != leave:
                    scorer.update(kept[name], self._ref[name][0])
            last = scorer.scores()
            partial.append([last[0], last[1], last[2], last[3], last[4]] + last[5].reshape(-1).tolist())
        partial = np.asarray(partial)
        conf = [jackknife_estimation(global_values[i], partial[:, i], 0.05)[3] for i in range(len(global_values))]
        return ([last[0], conf[0]], [last[1], conf[1]], [last[2], conf[2]], [last[3], conf[3]], [last[4], conf[4]],
                [last[5], np.array(conf)[5:].reshape(5, self._nb_classes, 2)])


class ComputeSELDResultsFromEventOverlap(ComputeSELDResults):
    """Scores restricted to the reference frames with overlapping events (seld_metrics.py:522-717; printed by the
    reference's test.py:125-133 as "class-independent polyphony" and, with ``classwise_overlap_test=True``, as
    "class-homogenous polyphony"): a frame counts when it holds more than one event (or more than one event of the SAME
    class); reference files without such a frame are left out, predictions are cut down to those frames, the recording
    length stays ``max(frame index)`` of the full reference file."""

    def __init__(self, params, ref_files_folder=None, use_polar_format=True, classwise_overlap_test=False):
        if not use_polar_format:
            raise NotImplementedError("cartesian reference format (seld_metrics.py:547-548) is not used by the reference's callers")
        dc = params["data_config"]
        self._nb_classes = dc["nb_classes"]
        self._fpb = int(dc["sr"] / float(int(dc["sr"] * dc["label_hop_len_s"])))
        self._ref, self._ov_frames = {}, {}
        for name in os.listdir(ref_files_folder):
            gt = load_output_format_file(os.path.join(ref_files_folder, name))
            nb = max(list(gt.keys()))
            keep = {}
            for frame, events in gt.items():
                if classwise_overlap_test:
                    cnt = np.zeros(self._nb_classes)
                    for ev in events:
                        cnt[ev[0]] += 1
                    hit = cnt.max() > 1
                else:
                    hit = len(events) > 1
                if hit:
                    keep[frame] = events
            self._ov_frames[name] = list(keep.keys())
            if keep:
                self._ref[name] = (segment(keep, nb, self._fpb), nb)
        self.nb_overlap_files = len(self._ref)
        self.nb_overlap_frames = sum(len(v) for v in self._ov_frames.values())

    def _pred_labels(self, pred_files_path, name):
        if name not in self._ref:
            return None
        pred = cartesian_to_polar(load_output_format_file(os.path.join(pred_files_path, name)))
        pred = {fr: pred[fr] for fr in self._ov_frames[name] if fr in pred}
        return segment(pred, self._ref[name][1], self._fpb)
