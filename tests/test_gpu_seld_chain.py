"""GPU: the whole evaluation chain against the run of the REAL reference stored in ``golden/seld_chain.npz``
(make_golden.py::gen_seld_chain): WAV files + DCASE metadata on disk -> ``FoaDataset('test')`` ->
int16 normalise + K1 features + encoder + head (eval) on the GPU -> loss -> decode + NMS -> CSV -> SELD scores."""
import os
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops as _ops
    return _ops


def _params(data_pth, g):
    return {"args": {"device": "cuda:0", "encoder": "se-resnet34", "loss": "adyolo"},
            "data_config": {"nb_classes": 12, "sr": 24000, "label_hop_len_s": 0.1, "data_pth": str(data_pth)},
            "aug_config": {"rotation_augment": False, "spec_augment": False},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "conf_thresh": float(g["conf_thresh"]), "clss_thresh": float(g["clss_thresh"]),
                             "unify_thresh": float(g["unify_thresh"]), "nms": "conn-merge",
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0, "class_gain": 3.0}}}


def _rows(path):
    rows = [[float(v) for v in line.strip().split(",")] for line in open(path) if line.strip()]
    return np.asarray(rows, dtype=np.float64).reshape(len(rows), 6)


@pytest.mark.parametrize("algo", ["winograd4", "direct"])
@pytest.mark.parametrize("mode", ["eager", "graphs", "graphs-batched"])
def test_seld_chain_matches_the_reference_run(ops, tmp_path, monkeypatch, mode, algo):
    """Same CSV rows as the reference wrote (frame / class exact and in the same order per frame and class up to
    confidence near-ties -- compared sorted; unit vectors within 1e-3), the same mean loss (1e-3 relative) and
    ER / F / LE / LR / SELD within 0.01 (measured: see DESIGN.md), for the eager loop, for ``ForwardGraphs`` one clip per
    replay and for ``ForwardGraphs`` with equal-length clips batched; with the default F(4x4)/F(2x2) Winograd convolutions
    and with the direct kernel."""
    monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
    sys.path.insert(0, G)
    from scipy.io import wavfile
    from oracle.filler import fill_module_
    from seld_chain_inputs import CLIPS, chain_clip, crc
    from adyolo_amd import test as atest
    from adyolo_amd.datasets import FoaDataset
    from adyolo_amd.features import FeatureExtractor, load_scaler_npz
    from adyolo_amd.graph import ForwardGraphs
    from adyolo_amd.postprocess import LabelPostProcessor
    from adyolo_amd.seld_metrics import ComputeSELDResults
    from adyolo_amd.wrapper import WrapperCriterion, WrapperModel
    g = np.load(os.path.join(G, "seld_chain.npz"))
    wdir, cdir = os.path.join(tmp_path, "foa_dev", "dev-test"), os.path.join(tmp_path, "metadata_dev", "dev-test")
    os.makedirs(wdir), os.makedirs(cdir)
    for i, (name, seed, n) in enumerate(CLIPS):
        pcm = chain_clip(seed, n)
        assert crc(pcm) == int(g["crc32"][i])
        wavfile.write(os.path.join(wdir, name + ".wav"), 24000, pcm)
        with open(os.path.join(cdir, name + ".csv"), "w") as f:
            for r in g["ref_" + name]:
                f.write("%d,%d,%d,%d,%d\n" % tuple(int(v) for v in r))
    prm = _params(tmp_path, g)
    model = WrapperModel((1, 7, 400, 64), (), prm)
    fill_module_(model)
    model = model.to("cuda:0").eval()
    fx = FeatureExtractor(load_scaler_npz(os.path.join(G, "scaler_DCASE2021.npz")), "cuda:0")
    crit, post = WrapperCriterion(prm), LabelPostProcessor(prm)
    ds = FoaDataset(prm, "test", is_valid=True)
    out = os.path.join(tmp_path, "output_test")
    if mode == "eager":
        loss = atest.test_epoch_audio(ds, model, fx, crit, post, "cuda:0", out)
    else:
        fg = ForwardGraphs(model, fx, post, warm_calls=0)
        loss = atest.test_epoch_audio(ds, model, fx, crit, post, "cuda:0", out, batch_size=1 if mode == "graphs" else 4, forward=fg)
        assert fg.captures >= 2 and fg.replays >= 2
    assert abs(loss - float(g["mean_loss"])) <= 1e-3 * float(g["mean_loss"]), (loss, float(g["mean_loss"]))
    worst = 0.0
    for name, _, _ in CLIPS:
        got, ref = _rows(os.path.join(out, name + ".csv")), g["pred_" + name]
        assert got.shape == ref.shape, "%s: %d rows, the reference wrote %d" % (name, len(got), len(ref))
        got, ref = np.asarray(sorted(got.tolist())), np.asarray(sorted(ref.tolist()))
        np.testing.assert_array_equal(got[:, :3], ref[:, :3], err_msg=name)              # frame, class, track 0
        worst = max(worst, float(np.abs(got[:, 3:] - ref[:, 3:]).max()))
    assert worst <= 1e-3, "unit vectors differ by %.3e" % worst
    res = ComputeSELDResults(prm, cdir).get_SELD_Results(out)
    got = np.asarray([float(v) for v in res[:5]])
    assert np.all(np.abs(got - g["scores"]) <= 0.01), (got, g["scores"])
    print("seld chain %s/%s: worst xyz diff %.2e, loss %.6f (ref %.6f), scores %s (ref %s)"
          % (mode, algo, worst, loss, float(g["mean_loss"]), got, g["scores"]))
