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
