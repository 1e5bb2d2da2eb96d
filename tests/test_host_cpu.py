"""CPU: host-side logic of the product package (label encoder, collate, synthetic workload, parameter
naming / initialisation) and the C-ABI surface (library loads, exports every declared symbol).
No compute kernels are launched here (no GPU in this container)."""
import os
import re

import numpy as np
import pytest
import torch

import adyolo_amd  # noqa: F401  (import shim at the repo root)
from adyolo_amd import _lib
from adyolo_amd.datasets import YoloLabelEncoder, collate_fn, synthetic_audio, synthetic_targets
from oracle import labels as olab
from oracle import seresnet as onet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_header_symbols_are_exported_and_bound():
    """Every function declared in include/adyolo_hip.h is exported by the .so and bound in _lib.SIGNATURES."""
    hdr = open(os.path.join(ROOT, "include", "adyolo_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(adyolo_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                       # raises if the library or a symbol is missing
    assert lib.adyolo_abi_version() == 1
    assert lib.adyolo_loss_workspace_words(4, 32, 5, 10) > 6 * 4 * 32 * 5


def test_ops_refuse_cpu_tensors():
    from adyolo_amd import ops
    with pytest.raises(_lib.AdyoloHipError):
        ops.add(torch.zeros(8), torch.zeros(8))
    from adyolo_amd.models.backbones.resnet import SEResnet34
    enc = SEResnet34((1, 7, 16, 64), (), {"data_config": {"nb_classes": 12}})
    with pytest.raises(RuntimeError):
        enc(torch.zeros(1, 7, 16, 64))


def test_label_encoder_matches_reference_golden():
    g = np.load(os.path.join(G, "labels.npz"))
    enc = YoloLabelEncoder()
    got = []
    for i, (az, el) in enumerate(g["sweep_in"]):
        for r in enc.get_yolo_label({0: [[1, 0, float(az), float(el)]]}, 1):
            got.append([i] + [float(v) for v in r])
    np.testing.assert_array_equal(np.asarray(got), g["sweep_rows"])


def test_label_encoder_and_collate_match_oracle():
    events = {0: [[3, 0, 10.0, 5.0]], 1: [[3, 0, 10.0, 5.0], [7, 1, -170.0, 40.0]], 2: [[0, 0, 180.0, -30.0]],
              4: [[5, 0, 44.9, -90.0], [2, 2, 47.0, -85.0]], 9: [[6, 0, 20.0, 20.0]]}
    enc = YoloLabelEncoder()
    rows = enc.get_yolo_label({k: [list(e) for e in v] for k, v in events.items()}, 8)
    ref = olab.yolo_label({k: [list(e) for e in v] for k, v in events.items()}, 8)
    np.testing.assert_array_equal(np.asarray(rows, dtype=np.float64), np.asarray(ref, dtype=np.float64))
    feats = [torch.zeros(7, 8, 4), torch.ones(7, 8, 4)]
    feat, target = collate_fn(list(zip(feats, [rows, []])))
    _, tref = olab.collate([f.numpy() for f in feats], [ref, []])
    np.testing.assert_array_equal(target.numpy(), tref)
    assert feat.shape == (2, 7, 8, 4)
    with pytest.raises(RuntimeError):
        collate_fn(list(zip(feats, [[], []])))


def test_synthetic_workload_shapes():
    a = synthetic_audio(2, 1200, seed=1)
    assert a.shape == (2, 1200, 4) and a.dtype == torch.float32
    pcm = (a.double() - 1e-8) * 32768.0
    assert float((pcm - pcm.round()).abs().max()) < 1e-3 and float(a.std()) == pytest.approx(0.1, rel=0.1)
    t = synthetic_targets(4, 50, 12, seed=2)
    assert t.shape[1] == 7 and t.dtype == torch.float32
    assert int(t[:, 0].max()) <= 3 and int(t[:, 1].max()) <= 49 and int(t[:, 4].max()) <= 11
    assert 2.5 < t.shape[0] / (4 * 50) < 4.5          # ~3.4 rows per (sample, frame), SURVEY 8d


def _params():
    return {"args": {"device": "cpu", "encoder": "se-resnet34", "loss": "adyolo"}, "data_config": {"nb_classes": 12},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0,
                                            "class_gain": 3.0}}}


def test_state_dict_is_abi_compatible_with_reference():
    from adyolo_amd.wrapper import WrapperModel
    model = WrapperModel((1, 7, 800, 64), (), _params())
    sd = model.state_dict()
    spec = dict(onet.state_dict_spec())
    assert set(sd) == set(spec)
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(spec[k]), k
    assert sum(p.numel() for p in model.parameters()) == 6682093


def test_default_init_is_bit_identical_to_reference_under_seed_100():
    """Same seed -> same initial weights as the reference modules (fingerprints from make_golden.gen_init)."""
    from adyolo_amd.wrapper import WrapperModel
    g = np.load(os.path.join(G, "init_seed100.npz"))
    torch.manual_seed(100)
    sd = WrapperModel((1, 7, 800, 64), (), _params()).state_dict()
    for name, s, f in zip(g["names"], g["sums"], g["first"]):
        v = sd[str(name)]
        assert float(v.reshape(-1)[0]) == float(f), name
        assert float(v.double().sum()) == pytest.approx(float(s), rel=1e-12, abs=1e-12), name


def test_wrapper_rejects_unknown_names():
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    p = _params()
    p["args"]["encoder"] = "nope"
    with pytest.raises(NotImplementedError):
        WrapperModel((1, 7, 8, 64), (), p)
    p = _params()
    p["args"]["loss"] = "nope"
    with pytest.raises(NotImplementedError):
        WrapperCriterion(p)


def test_conformer_state_dict_and_init_match_reference():
    """549 keys / shapes of the reference ResnetConformer, and bit-identical default init under seed 100."""
    from adyolo_amd.models.backbones.resnet_conformer import ResnetConformer
    g = np.load(os.path.join(G, "conformer.npz"))
    torch.manual_seed(100)
    m = ResnetConformer((1, 7, 64, 64), (), {"data_config": {"nb_classes": 12}})
    sd = m.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["all_keys"]] or set(sd.keys()) == set(str(k) for k in g["all_keys"])
    assert len(sd) == 549 and sum(p.numel() for p in m.parameters()) == 32386752
    for name, shp, s, f in zip(g["names"], g["shapes"], g["sums"], g["first"]):
        v = sd[str(name)]
        assert str(tuple(v.shape)) == str(shp), name
        assert float(v.reshape(-1)[0]) == float(f), name
        assert float(v.double().sum()) == pytest.approx(float(s), rel=1e-12, abs=1e-12), name


def test_seld_metrics_match_reference(tmp_path):
    """ER / F / LE / LR / SELD and the class-wise table vs the reference evaluator on the same CSV folders."""
    from adyolo_amd.seld_metrics import ComputeSELDResults
    g = np.load(os.path.join(G, "metrics.npz"))
    ref_dir, pred_dir = tmp_path / "ref", tmp_path / "pred"
    ref_dir.mkdir(); pred_dir.mkdir()
    for i, name in enumerate(g["names"]):
        with open(ref_dir / str(name), "w") as f:
            for r in g["ref_%d" % i]:
                f.write("%d,%d,%d,%d,%d\n" % tuple(int(v) for v in r))
        with open(pred_dir / str(name), "w") as f:
            for r in g["pred_%d" % i]:
                f.write("{},{},{},{},{},{}\n".format(int(r[0]), int(r[1]), 0, float(r[3]), float(r[4]), float(r[5])))
    prm = {"data_config": {"nb_classes": 12, "sr": 24000, "label_hop_len_s": 0.1}}
    res = ComputeSELDResults(prm, str(ref_dir)).get_SELD_Results(str(pred_dir))
    np.testing.assert_allclose(np.asarray([float(v) for v in res[:5]]), g["scores"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(res[5], g["classwise"], rtol=1e-9, atol=1e-9)
    # overlap-only variants the reference prints at test time (test.py:125-133; seld_metrics.py:522-717)
    from adyolo_amd.seld_metrics import ComputeSELDResultsFromEventOverlap, jackknife_estimation
    for tag, flag in (("poly", False), ("homog", True)):
        obj = ComputeSELDResultsFromEventOverlap(prm, str(ref_dir), classwise_overlap_test=flag)
        assert obj.nb_overlap_files == int(g["ov_%s_nfiles" % tag]) and obj.nb_overlap_frames == int(g["ov_%s_nframes" % tag])
        r = obj.get_SELD_Results(str(pred_dir))
        np.testing.assert_allclose(np.asarray([float(v) for v in r[:5]]), g["ov_%s_scores" % tag], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(r[5], g["ov_%s_classwise" % tag], rtol=1e-9, atol=1e-9)
    # jackknife (seld_metrics.py:149-186, :441-476): the intervals do not depend on the file order, the point values the
    # reference returns next to them are those of its LAST leave-one-out pass (a reference quirk, kept)
    est = jackknife_estimation(0.37, np.asarray([0.35, 0.36, 0.41, 0.39]), 0.05)
    np.testing.assert_allclose([est[0], est[1], est[2], est[3][0], est[3][1]], g["jk_unit"], rtol=1e-12, atol=1e-12)
    jk = ComputeSELDResults(prm, str(ref_dir)).get_SELD_Results(str(pred_dir), is_jackknife=True)
    np.testing.assert_allclose(np.asarray([np.asarray(jk[i][1]) for i in range(5)]), g["jk_conf"], rtol=1e-8, atol=1e-9)
    np.testing.assert_allclose(jk[5][1], g["jk_classwise_conf"], rtol=1e-8, atol=1e-9)
    from adyolo_amd.test import score_output_folder
    three = score_output_folder(prm, str(ref_dir), str(pred_dir))
    np.testing.assert_allclose([float(v) for v in three["polyphony"][:5]], g["ov_poly_scores"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose([float(v) for v in three["homogenous"][:5]], g["ov_homog_scores"], rtol=1e-9, atol=1e-9)
    if list(os.listdir(pred_dir)) == [str(n) for n in g["jk_order"]]:
        np.testing.assert_allclose([float(jk[i][0]) for i in range(5)], g["jk_points"], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(jk[5][0], g["jk_classwise"], rtol=1e-9, atol=1e-9)


def test_rotation_labels_match_reference():
    from adyolo_amd.augmentations import rotate_labels
    g = np.load(os.path.join(G, "rotation.npz"))
    label = {0: [[3, 0, 10.0, 5.0]], 4: [[1, 0, -170.0, 40.0], [2, 1, 180.0, -30.0]],
             7: [[5, 0, -95.0, -60.0], [5, 1, 135.0, 0.0]]}
    for c in range(16):
        lab = rotate_labels(label, c)
        rows = np.asarray([[fr] + [float(v) for v in ev] for fr, evs in lab.items() for ev in evs])
        np.testing.assert_array_equal(rows, g["label_rot"][c])


def _cpu_params():
    return {"args": {"device": "cpu", "encoder": "se-resnet34", "loss": "adyolo"}, "data_config": {"nb_classes": 12},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


def test_checkpoint_files_follow_the_reference_format(tmp_path):
    """optim_state_dict is torch.optim.Adam's layout in model.parameters() order (what the reference's
    optimizer.state_dict() writes, train.py:239-247), whatever the flat buffer's internal order; model_ckpt.h5 round-trips
    model, optimizer moments, step count and all four RNG streams (utility.py:32-50) without touching torch.cuda on CPU."""
    import random
    from adyolo_amd import checkpoint as ck
    from adyolo_amd.dist import FlatParameters
    from adyolo_amd.train import FusedAdam
    from adyolo_amd.wrapper import WrapperModel
    torch.manual_seed(3)
    model = WrapperModel((1, 7, 64, 64), (), _cpu_params())
    twin = [torch.nn.Parameter(p.detach().clone()) for p in model.parameters()]
    adam = torch.optim.Adam(twin, lr=2e-3, betas=(0.8, 0.95), eps=1e-7)
    for _ in range(2):
        for p in twin:
            p.grad = torch.randn_like(p)
        adam.step()
    ref = adam.state_dict()
    flat = FlatParameters(model)                                    # reversed internal order
    opt = FusedAdam(flat)
    ck.load_optimizer_state_dict(opt, model, ref)
    assert (opt.lr, opt.betas, opt.eps, opt.step_count) == (2e-3, (0.8, 0.95), 1e-7, 2)
    mine = ck.optimizer_state_dict(opt, model)
    assert mine["param_groups"][0]["params"] == ref["param_groups"][0]["params"]
    for i, st in ref["state"].items():
        assert int(mine["state"][i]["step"]) == int(st["step"])
        assert torch.equal(mine["state"][i]["exp_avg"], st["exp_avg"])
        assert torch.equal(mine["state"][i]["exp_avg_sq"], st["exp_avg_sq"])
    # file round trip incl. RNG
    path = os.path.join(tmp_path, "model_ckpt.h5")
    os.environ["PYTHONHASHSEED"] = "100"
    ck.save_checkpoint(path, model, opt, 7, 0.4, {"best_epoch": 3, "best_conf_thresh": 0.4}, ["a.wav", "b.wav"], "cpu")
    expect = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    saved = torch.load(path, map_location="cpu", weights_only=False)
    assert sorted(saved) == sorted(["start_epoch_nb", "model_state_dict", "optim_state_dict", "confidence_thresh",
                                    "rng_state", "best_log", "train_remaining_file"])
    assert list(saved["model_state_dict"]) == list(model.state_dict())
    model2 = WrapperModel((1, 7, 64, 64), (), _cpu_params())
    opt2 = FusedAdam(FlatParameters(model2))
    got = ck.load_checkpoint(path, model2, opt2, device="cpu")
    assert got["start_epoch_nb"] == 7 and got["train_remaining_file"] == ["a.wav", "b.wav"]
    assert (random.random(), float(np.random.rand()), float(torch.rand(1))) == expect
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.step_count == 2 and torch.equal(opt2.exp_avg.sort().values, opt.exp_avg.sort().values)
    # torch.optim.Adam itself accepts the written optimizer state (a reference-side resume)
    adam2 = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in model.parameters()])
    adam2.load_state_dict(saved["optim_state_dict"])
    assert adam2.state_dict()["state"][0]["exp_avg"].shape == ref["state"][0]["exp_avg"].shape


def test_specaug_policy():
    """identity by default and on validation data; drawn ranges follow torchaudio 0.10's mask_along_axis bounds; the
    reference's (C, T, F) call makes its "time" mask act on mel bins and its "frequency" mask on frames."""
    import random
    from adyolo_amd.augmentations import SpecAug
    off = SpecAug({"aug_config": {"spec_augment": False}}, is_valid=False)
    val = SpecAug({"aug_config": {"spec_augment": True, "spec_augment_thresh": 1.0}}, is_valid=True)
    x = torch.zeros(1)
    assert off.augment(x) is x and val.augment(x) is x
    sa = SpecAug({"aug_config": {"spec_augment": True, "spec_augment_thresh": 1.0, "spec_augment_time_mask_param": 10,
                                 "spec_augment_freq_mask_param": 30}}, is_valid=False)
    random.seed(5)
    r = sa.draw(64, 800, 64).numpy()
    assert r.shape == (64, 4) and (r[:, 0] >= 0).all() and (r[:, 1] <= 800).all() and (r[:, 3] <= 64).all()
    assert (r[:, 1] - r[:, 0]).max() <= 30 and (r[:, 3] - r[:, 2]).max() <= 10        # frames <- freq param, bins <- time param
    assert (r[:, 1] - r[:, 0]).max() > 10                                            # really the 30-wide one
    random.seed(5)
    never = SpecAug({"aug_config": {"spec_augment": True, "spec_augment_thresh": -1.0, "spec_augment_time_mask_param": 10,
                                    "spec_augment_freq_mask_param": 30}}, is_valid=False).draw(8, 800, 64)
    assert int(never.abs().sum()) == 0


def test_threshold_sweep_runs_the_model_once_and_keeps_the_first_minimum(tmp_path):
    """reference train.py:178-203 semantics (strict '<': the first of equal minima wins; both thresholds are rewritten)
    with one forward pass per file instead of nine."""
    from adyolo_amd import test as atest

    class Model(torch.nn.Module):
        calls = 0

        def forward(self, x):
            Model.calls += 1
            return x.sum(dim=(1, 2, 3), keepdim=False).reshape(1, 1, 1)

    class Post:
        def __init__(self):
            self.conf_thresh = self.clss_thresh = 0.5

        def get_conf_thresh(self):
            return self.conf_thresh

        def set_conf_thresh(self, t):
            self.conf_thresh = self.clss_thresh = t

        def decode(self, output):
            return float(output)

        def select(self, dec):
            return {0: [[int(round(self.conf_thresh * 10)), 1.0, 0.0, 0.0]]}

    class Scorer:
        def get_SELD_Results(self, pth):
            names = sorted(os.listdir(pth))
            assert names == ["a.csv", "b.csv"]
            cls = int(open(os.path.join(pth, names[0])).read().split(",")[1])      # = round(10 * threshold)
            seld = {3: 0.25, 6: 0.25}.get(cls, 0.5 + 0.01 * cls)                    # two equal minima: 0.3 must win
            return 0.1, 0.2, 3.0, 0.4, seld, None

    loader = [(torch.ones(1, 7, 4, 64), None), (torch.ones(1, 7, 4, 64) * 2, None)]
    post = Post()
    new, table, loss = atest.sweep_conf_thresh(loader, ["a", "b"], Model(), lambda o, l: o.reshape(1), post, Scorer(), "cpu",
                                               os.path.join(tmp_path, "out"))
    assert Model.calls == 2 and len(table) == 9
    assert abs(new - 0.3) < 1e-9 and abs(post.conf_thresh - 0.3) < 1e-9 and abs(post.clss_thresh - 0.3) < 1e-9
    assert abs(loss - (7 * 4 * 64 * 1.5)) < 1e-3


def _write_split(root, sub, names, n_samples, rs):
    from scipy.io import wavfile
    wdir, cdir = os.path.join(root, "foa_dev", sub), os.path.join(root, "metadata_dev", sub)
    os.makedirs(wdir), os.makedirs(cdir)
    for nm in names:
        wavfile.write(os.path.join(wdir, nm + ".wav"), 24000, rs.randint(-3000, 3000, size=(n_samples, 4)).astype(np.int16))
        with open(os.path.join(cdir, nm + ".csv"), "w") as f:
            for fr in range(0, n_samples // 2400, 3):
                f.write("%d,%d,0,%d,%d\n" % (fr, fr % 12, (fr * 37) % 360 - 180, (fr * 11) % 160 - 80))


def test_raw_audio_dataset_mirrors_the_reference_harness(tmp_path):
    """FoaDataset: reference directory layout, per-epoch sampling without replacement incl. the three refill branches of
    datasets.py:67-91, resume hooks, CSV reader, label rotation + AD-YOLO encoding on the host; audio stays int16."""
    import random
    from adyolo_amd.augmentations import rotate_labels
    from adyolo_amd.datasets import FoaDataset, audio_collate_fn
    rs = np.random.RandomState(0)
    names = ["clip%02d" % i for i in range(10)]
    _write_split(str(tmp_path), "dev-train-chunked_1s_1s", names, 24000, rs)
    _write_split(str(tmp_path), "dev-test", ["t0", "t1"], 48000, rs)
    params = {"args": {"loss": "adyolo"}, "aug_config": {"rotation_augment": True},
              "data_config": {"data_pth": str(tmp_path), "chunk_window_s": 1, "chunk_stride_s": 1, "nb_classes": 12},
              "train_config": {"batch_size": 2, "nb_iters": 2, "grid_size": [45, 45], "g_overlap": 0.5}}
    random.seed(4)
    ds = FoaDataset(params, "train")
    assert len(ds) == 4 and len(ds.get_remaining_file()) == 6 and not set(ds.get_filelist()) & set(ds.get_remaining_file())
    ds.sample_filelist_for_train_iter()                                  # 6 left -> 4 taken
    assert len(ds.get_remaining_file()) == 2
    rest = list(ds.get_remaining_file())
    ds.sample_filelist_for_train_iter()                                  # 2 left: both kept, 2 drawn from the refilled list
    assert set(rest) <= set(ds.get_filelist()) and len(ds.get_filelist()) == 4 and len(ds.get_remaining_file()) == 8
    ds.init_remaining_file_from_list([])
    ds.sample_filelist_for_train_iter()                                  # empty: refill first
    assert len(ds.get_remaining_file()) == 6
    # one item: int16 audio untouched, label rotated with the drawn combination and encoded
    random.seed(9)
    pcm, comb, rows = ds[0]
    random.seed(9)
    assert comb == int(random.uniform(0, 16)) and pcm.dtype == np.int16 and pcm.shape == (24000, 4)
    lab = FoaDataset.load_csv2dict(os.path.join(ds.csv_pth, ds.get_filelist()[0] + ".csv"))
    assert rows == ds.encoder.get_yolo_label(rotate_labels(lab, comb), 10)
    # evaluation split: no sampling, no rotation; collate builds the (M, 7) target like the reference collate_fn
    te = FoaDataset(params, "test", is_valid=True)
    assert sorted(te.get_filelist()) == ["t0", "t1"] and te[0][1] == 0
    pcm_b, combs, target = audio_collate_fn([te[0], te[1]])
    assert pcm_b.shape == (2, 48000, 4) and pcm_b.dtype == torch.int16 and combs == [0, 0]
    assert target.shape[1] == 7 and set(target[:, 0].tolist()) == {0.0, 1.0}


def test_raw_audio_dataset_shards_by_rank(tmp_path):
    """Data parallelism: every rank performs the SAME global draw (so ``remaining_file`` agrees everywhere) and keeps the
    files rank, rank + world, ...: disjoint shards whose union is the single-process draw of batch_size * world files."""
    import random
    from adyolo_amd.datasets import FoaDataset
    rs = np.random.RandomState(0)
    _write_split(str(tmp_path), "dev-train-chunked_1s_1s", ["clip%02d" % i for i in range(12)], 24000, rs)
    _write_split(str(tmp_path), "dev-test", ["t0", "t1", "t2"], 24000, rs)
    params = {"args": {"loss": "adyolo"}, "aug_config": {"rotation_augment": False},
              "data_config": {"data_pth": str(tmp_path), "chunk_window_s": 1, "chunk_stride_s": 1, "nb_classes": 12},
              "train_config": {"batch_size": 2, "nb_iters": 2, "grid_size": [45, 45], "g_overlap": 0.5}}
    shards, rests = [], []
    for rank in range(2):
        random.seed(4)
        ds = FoaDataset(params, "train", rank=rank, world=2)
        shards.append(ds.get_filelist())
        rests.append(sorted(ds.get_remaining_file()))
    assert len(shards[0]) == len(shards[1]) == 4 and not set(shards[0]) & set(shards[1])
    assert rests[0] == rests[1] and len(rests[0]) == 4
    random.seed(4)
    single = dict(params, train_config=dict(params["train_config"], batch_size=4))
    whole = FoaDataset(single, "train", rank=0, world=1).get_filelist()
    assert whole[0::2] == shards[0] and whole[1::2] == shards[1]
    te = [FoaDataset(params, "test", is_valid=True, rank=r, world=2).get_filelist() for r in range(2)]
    assert te == [["t0", "t2"], ["t1"]]


def test_dispatch_thresholds_and_parameter_epoch(monkeypatch):
    """Host logic that needs no GPU: the F(4x4) dispatch thresholds are read once and follow ``reload_thresholds()`` (the
    conftest fixture calls it when a test moves an ADYOLO_W4_* variable); the weight-gradient form is decided in one place;
    the parameter epoch that evaluation-mode caches are keyed on moves when asked to; a state load on the wrapper moves it too
    (the hook ``WrapperModel`` registers).  (The bf16x3 math mode this test used to cover left the product in round 5.)"""
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops
    for k in ("ADYOLO_W4_MIN_K", "ADYOLO_W4_MIN_K_ADDEND", "ADYOLO_W4_MIN_WGS"):
        monkeypatch.delenv(k, raising=False)
    assert ops.reload_thresholds() == {"min_k": 32, "min_k_addend": 32, "min_wgs": 200, "min_wgrad_work": 6000000, "min_k_32": 32}
    assert ops._w4_eligible(64, 64) and ops._w4_eligible(32, 64) and not ops._w4_eligible(64, 32) and not ops._w4_eligible(1024, 64)
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "64")                     # (the fixture reloads)
    assert ops.W4_THRESHOLDS["min_k"] == 64 and not ops._w4_eligible(32, 64)
    monkeypatch.delenv("ADYOLO_W4_MIN_K")                            # (the fixture reloads on delenv too)
    assert ops.W4_THRESHOLDS["min_k"] == 32 and ops.reload_thresholds()["min_k"] == 32
    # the on / off switches live in the same table, and the table is what recorded graphs are keyed on
    for k in ("ADYOLO_W4_PERSIST", "ADYOLO_W4_NARROW", "ADYOLO_WINO1D"):
        monkeypatch.delenv(k, raising=False)
    t0 = ops.switch_table()
    assert t0["persist"] and t0["narrow"] and t0["wino1d"] and t0["wgrad_algo"] is None and t0["conv_algo"] == ops.conv_algo()
    s0 = ops.switch_stamp()
    monkeypatch.setenv("ADYOLO_W4_PERSIST", "0")
    assert not ops.W4_THRESHOLDS["persist"] and ops.switch_stamp() != s0 and not ops.w4_narrow_ok(64, 64)
    import os as _os
    _os.environ["ADYOLO_WINO1D"] = "0"                               # a direct write is NOT seen until reload_thresholds()
    assert ops.W4_THRESHOLDS["wino1d"] and ops.wino1d_ok(32, 800, 64, 64)
    ops.reload_thresholds()
    assert not ops.W4_THRESHOLDS["wino1d"] and not ops.wino1d_ok(32, 800, 64, 64)
    del _os.environ["ADYOLO_WINO1D"]
    monkeypatch.delenv("ADYOLO_W4_PERSIST")
    assert ops.switch_stamp() == s0
    assert not hasattr(ops, "math_mode")
    monkeypatch.delenv("ADYOLO_WGRAD_ALGO", raising=False)
    monkeypatch.delenv("ADYOLO_CONV_ALGO", raising=False)
    assert ops.wgrad_form(64, 128) == ("wino_wgrad_kernel", 16.0 / 36.0) and ops.wgrad_form(8, 32) == ("conv3x3_wgrad_kernel", 1.0)
    assert ops.wgrad_form(64, 128, "direct")[1] == 1.0
    e0 = ops.PARAMS_EPOCH[0]
    ops.params_changed()
    assert ops.PARAMS_EPOCH[0] == e0 + 1
    from adyolo_amd.wrapper import WrapperModel
    torch.manual_seed(0)
    prm = {"args": {"encoder": "se-resnet34", "loss": "adyolo"}, "data_config": {"nb_classes": 12},
           "train_config": {"grid_size": [45, 45], "nb_anchors": 5}}
    model = WrapperModel((1, 7, 80, 64), (), prm)
    e1 = ops.PARAMS_EPOCH[0]
    model.load_state_dict(model.state_dict())
    assert ops.PARAMS_EPOCH[0] == e1 + 1
