"""Evaluation loop.  Mirror of ``test_epoch`` / ``write_seld_output_file`` (/root/reference/src/test.py:26-60):
per file (one full clip, batch 1): eval-mode forward, loss, post-processing (GPU decode + host NMS), one DCASE CSV per
file under ``output_pth``.  The SELD scores come from ``seld_metrics.ComputeSELDResults(ref_dir).get_SELD_Results(output_pth)``."""
import os
import shutil

import torch

from .postprocess import write_seld_output_file


def delete_and_create_folder(dir_pth):
    if os.path.exists(dir_pth) and os.path.isdir(dir_pth):
        shutil.rmtree(dir_pth)
    os.makedirs(dir_pth, exist_ok=True)


def test_epoch(dataloader, filelist, model, criterion, postprocessor, device, output_pth):
    """dataloader yields (feat (1,7,T,64), label) in the order of ``filelist`` (names without extension)."""
    model.eval()
    delete_and_create_folder(output_pth)
    total = None
    n = 0
    with torch.no_grad():
        for i, (feat, label) in enumerate(dataloader):
            output = model(feat.to(device).float())
            loss = criterion(output, label)
            total = loss.reshape(-1)[:1].clone() if total is None else total + loss.reshape(-1)[:1]
            write_seld_output_file(os.path.join(output_pth, filelist[i] + ".csv"), postprocessor.postprocess(output))
            n = i + 1
    return float(total) / max(n, 1) if total is not None else 0.0


def sweep_conf_thresh(dataloader, filelist, model, criterion, postprocessor, scorer, device, output_pth,
                      thresholds=None):
    """The reference's periodic threshold reset (src/train.py:178-203): try conf_thresh 0.1 .. 0.9, keep the first one
    with the lowest validation SELD score, leave it set on the post-processor (which rewrites conf AND class threshold,
    datasets.py:532-534).  The reference re-runs the whole validation epoch for each of the nine thresholds; the network
    output does not depend on the threshold, so here the model runs ONCE per file and only the host-side selection/NMS,
    the CSV files and the metrics are redone per threshold -- same files, same scores, a ninth of the forward passes.
    -> (new_thresh, [[ER, F, LE, LR, SELD] per threshold], mean validation loss)"""
    import numpy as np
    if thresholds is None:
        thresholds = np.arange(0.1, 1.0, 0.1)
    model.eval()
    decoded, total, n = [], None, 0
    with torch.no_grad():
        for i, (feat, label) in enumerate(dataloader):
            output = model(feat.to(device).float())
            loss = criterion(output, label)
            total = loss.reshape(-1)[:1].clone() if total is None else total + loss.reshape(-1)[:1]
            decoded.append(postprocessor.decode(output))
            n = i + 1
    new_thresh, best, table = postprocessor.get_conf_thresh(), 9999.0, []
    for th in thresholds:
        postprocessor.set_conf_thresh(th)
        delete_and_create_folder(output_pth)
        for i, dec in enumerate(decoded):
            write_seld_output_file(os.path.join(output_pth, filelist[i] + ".csv"), postprocessor.select(dec))
        er, f, le, lr, seld = scorer.get_SELD_Results(output_pth)[:5]
        table.append([er, f, le, lr, seld])
        if seld < best:
            new_thresh, best = th, seld
    postprocessor.set_conf_thresh(new_thresh)
    return new_thresh, table, (float(total) / max(n, 1) if total is not None else 0.0)


def test_epoch_audio(dataset, model, features, criterion, postprocessor, device, output_pth, batch_size=1, forward=None):
    """``test_epoch`` for a raw-audio ``FoaDataset`` split ('valid' / 'test' / 'infer'): int16 audio normalised on the GPU, K1
    features, encoder + head, loss, decode + NMS, one CSV per clip named after the file.  Returns the mean loss (0 for
    'infer', which has no labels).

    batch_size = 1 is the reference's loop (one clip per forward pass, test.py:33-60, train.py:130-133).  In evaluation mode a
    clip's output does not depend on what else is in the batch (BatchNorm uses its running statistics), so consecutive clips of
    EQUAL length may share one forward pass (batch_size > 1: same CSV files; the 60 s clips of a DCASE split all qualify) --
    3 ms per clip at B = 1 against ~1 ms at B = 8 on MI355X.  The loss stays per clip (its normalisers are per call), averaged
    over the clips like the reference's.  forward: optional ``graph.ForwardGraphs`` (K1 + model + decode replayed from a
    hipGraph per clip length); default: eager calls."""
    from . import ops
    from .datasets import audio_collate_fn
    model.eval()
    delete_and_create_folder(output_pth)
    total, n = None, 0
    names = dataset.get_filelist()
    i = 0
    with torch.no_grad():
        while i < len(dataset):
            items = []
            for j in range(i, min(i + max(1, int(batch_size)), len(dataset))):
                pcm, _, rows = dataset[j]
                t = (pcm.shape[0] // 600) * 600               # whole hops, like nb_feature_frames in datasets.py:283-286
                if items and t != items[0][1]:
                    break                                     # a clip of another length starts the next batch
                items.append((pcm, t, rows))
            t = items[0][1]
            pcm_b = torch.from_numpy(__import__("numpy").stack([it[0][:t] for it in items])).to(device).contiguous()
            audio = ops.pcm16_to_f32(pcm_b).view(len(items), t, 4)
            decoded = None
            if forward is not None:
                output, dec = forward(audio)
                if dec is not None:                           # the graph decoded the whole batch: ONE page-locked copy to the host
                    decoded = ops.to_host(dec).numpy()
            else:
                output = model(features(audio, channels_last8=True), channels_last8=True)
            for b, (pcm, _, rows) in enumerate(items):
                out_b = output[b:b + 1]
                if rows:
                    target = audio_collate_fn([(pcm, 0, rows)])[2]
                    loss = criterion(out_b, target)
                    total = loss.reshape(-1)[:1].clone() if total is None else total + loss.reshape(-1)[:1]
                    n += 1
                tp = output.shape[1]                          # decoded: [B * T'][Gaz][Gel][A][C+3], clip after clip
                rows_out = postprocessor.select(decoded[b * tp:(b + 1) * tp]) if decoded is not None else postprocessor.postprocess(out_b)
                write_seld_output_file(os.path.join(output_pth, names[i + b] + ".csv"), rows_out)
            i += len(items)
    return float(total) / max(n, 1) if total is not None else 0.0


def score_output_folder(params, ref_dir, output_pth, is_jackknife=False):
    """The three score sets the reference prints for one evaluated folder (src/test.py:104-133): all frames, frames with
    overlapping events ("class-independent polyphony") and frames with overlapping events of one class
    ("class-homogenous polyphony").  -> {'all' | 'polyphony' | 'homogenous': (ER, F, LE, LR, SELD, classwise)}."""
    from .seld_metrics import ComputeSELDResults, ComputeSELDResultsFromEventOverlap
    return {
        "all": ComputeSELDResults(params, ref_dir).get_SELD_Results(output_pth, is_jackknife),
        "polyphony": ComputeSELDResultsFromEventOverlap(params, ref_dir).get_SELD_Results(output_pth, is_jackknife),
        "homogenous": ComputeSELDResultsFromEventOverlap(params, ref_dir, classwise_overlap_test=True)
        .get_SELD_Results(output_pth, is_jackknife),
    }
