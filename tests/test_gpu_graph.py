"""GPU: the hipGraph-replayed train step and evaluation forward (ad-yolo_amd/graph.py) against the eager path.

The graph path must be the SAME arithmetic: losses and parameters are compared with ``torch.equal`` (bit for bit), with
the inter-layer GRU dropout active (its stream offset lives on the device in graph mode), with AD-YOLO target lists of
different lengths from step to step (padded to the graph's capacity with rows the assignment kernel skips) and across a
checkpoint of the optimizer / dropout state."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops as _ops
    return _ops


def _params(nb_classes=12, loss="adyolo"):
    return {"args": {"device": "cuda:0", "encoder": "se-resnet34", "loss": loss},
            "data_config": {"nb_classes": nb_classes},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "conf_thresh": 0.5, "clss_thresh": 0.5, "unify_thresh": 15.0, "nms": "conn-merge",
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0, "class_gain": 3.0},
                             "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


def _trainer(graph, loss="adyolo", t=80):
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.train import TrainStep
    torch.manual_seed(100)
    prm = _params(loss=loss)
    model = WrapperModel((1, 7, t, 64), (), prm).to("cuda:0")
    return TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=graph)


def test_adam_device_step_matches_torch(ops):
    """adyolo_adam_step_dev: the step counter and the bias corrections live on the device (what a replayed graph needs)."""
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(4099, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    pg = p0.to("cuda:0")
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    step_dev = torch.zeros(1, dtype=torch.int64, device="cuda:0")
    bc = torch.zeros(2, device="cuda:0")
    for _ in range(7):
        grad = torch.randn(p0.numel(), generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step_dev(pg, (grad * 4.0).to("cuda:0"), m, v, step_dev, bc, grad_scale=0.25)
    torch.cuda.synchronize()
    assert int(step_dev) == 7
    err = float((pg.cpu() - ref.detach()).abs().max())
    assert err <= 1e-6, err


def test_graphed_train_step_is_bit_identical_to_eager(ops):
    """Six steps (2 clips x 2 s, dropout 0.3 active) with target lists of different lengths: eager trainer vs graph trainer
    (step 0 eager warm-up, step 1 recorded + replayed, steps 2-5 replayed) -- identical losses and parameters, bit for bit;
    host mirrors (Adam step count, dropout offset) equal too."""
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    audios = [synthetic_audio(2, 24000 * 2, seed=70 + i).to("cuda:0") for i in range(3)]
    targets = [synthetic_targets(2, 20, 12, seed=80 + i) for i in range(6)]
    targets[3] = targets[3][: targets[3].shape[0] // 2].contiguous()         # a much shorter list: padding rows in play
    assert len({t.shape[0] for t in targets}) > 1
    te, tg = _trainer(False), _trainer(True)
    assert tg.graphs is not None and te.graphs is None
    le, lg = [], []
    for i in range(6):
        le.append(te.step(audios[i % 3], targets[i]).clone())
        lg.append(tg.step(audios[i % 3], targets[i]).clone())
    torch.cuda.synchronize()
    assert tg.graphs.captures == 1 and tg.graphs.replays == 5 and tg.graphs.eager_steps == 1
    for i, (a, b) in enumerate(zip(le, lg)):
        assert torch.equal(a, b), "loss of step %d: eager %r graph %r" % (i, float(a), float(b))
    assert torch.equal(te.flat.flat, tg.flat.flat), "parameters after 6 steps"
    assert torch.equal(te.optimizer.exp_avg_sq, tg.optimizer.exp_avg_sq)
    assert te.optimizer.step_count == tg.optimizer.step_count == 6
    assert int(tg.optimizer.step_dev) == 6
    se, sg = te.model.encoder.dropout_stream, tg.model.encoder.dropout_stream
    assert se.offset == sg.offset > 0 and int(sg.dev) == sg.offset
    for k, v in te.model.state_dict().items():                                # BatchNorm running statistics and counters
        assert torch.equal(v, tg.model.state_dict()[k]), k
    assert float(le[-1]) < float(le[0])


def test_graphed_step_survives_a_checkpoint_round_trip(ops, tmp_path):
    """Optimizer state and dropout stream are restored from a checkpoint INTO a trainer that already holds a recorded
    graph: the device-side counters are re-synchronised from the host mirrors and the next steps equal the eager run."""
    from adyolo_amd import checkpoint
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    audio = synthetic_audio(2, 24000 * 2, seed=91).to("cuda:0")
    targets = [synthetic_targets(2, 20, 12, seed=92 + i) for i in range(6)]
    te, tg = _trainer(False), _trainer(True)
    for i in range(3):
        te.step(audio, targets[i])
        tg.step(audio, targets[i])
    path = str(tmp_path / "model_ckpt.h5")
    checkpoint.save_checkpoint(path, te.model, te.optimizer, 1, 0.5, {}, [], "cuda:0")
    for i in range(3, 5):                                                     # the graph trainer runs ahead ...
        tg.step(audio, targets[i])
    checkpoint.load_checkpoint(path, tg.model, tg.optimizer, device="cuda:0")      # ... and is rolled back to step 3
    assert tg.optimizer.step_count == 3
    for i in range(3, 6):
        a, b = te.step(audio, targets[i]), tg.step(audio, targets[i])
        assert torch.equal(a, b), i
    assert torch.equal(te.flat.flat, tg.flat.flat)


def test_graphed_step_with_a_fixed_shape_target(ops):
    """ADPIT (config 5): the target is a dense (B, T', 6, 4, C) tensor -- copied into the graph's static buffer as is."""
    from adyolo_amd.datasets import synthetic_audio, ClasswiseLabelEncoder
    enc = ClasswiseLabelEncoder(12)
    ev = {0: [[3, 0, 10.0, 5.0]], 2: [[3, 0, 10.0, 5.0], [3, 1, -170.0, 40.0]], 5: [[1, 0, 0.0, 0.0], [2, 1, 90.0, 10.0]]}
    target = torch.stack([enc.get_adpit_label(ev, 20), enc.get_adpit_label({}, 20)]).to("cuda:0")
    audio = synthetic_audio(2, 24000 * 2, seed=95).to("cuda:0")
    te, tg = _trainer(False, "adpit"), _trainer(True, "adpit")
    for i in range(4):
        a, b = te.step(audio, target), tg.step(audio, target)
        assert torch.equal(a.reshape(-1), b.reshape(-1)), i
    assert tg.graphs.replays == 3 and torch.equal(te.flat.flat, tg.flat.flat)


def test_graphed_evaluation_forward_matches_eager(ops):
    """test_epoch's forward (B = 1): K1 -> encoder + head (eval) -> decode, replayed from a graph, equals the eager calls
    bit for bit, for two clip lengths (two graphs) and changing audio."""
    from adyolo_amd.wrapper import WrapperModel
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.postprocess import LabelPostProcessor
    from adyolo_amd.graph import ForwardGraphs
    from adyolo_amd.datasets import synthetic_audio
    torch.manual_seed(100)
    prm = _params()
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    model.eval()
    fx = FeatureExtractor(None, "cuda:0")
    post = LabelPostProcessor(prm)
    fg = ForwardGraphs(model, fx, post)
    for seconds in (2, 3):
        for i in range(4):
            audio = synthetic_audio(1, 24000 * seconds, seed=100 + 10 * seconds + i).to("cuda:0")
            with torch.no_grad():
                ref = model(fx(audio, channels_last8=True), channels_last8=True)
            dec_ref = post.decode(ref)
            out, dec = fg(audio)
            assert torch.equal(out, ref), "clip of %d s, call %d: max abs diff %.3e" % (seconds, i, float((out - ref).abs().max()))
            assert np.array_equal(dec.cpu().numpy(), dec_ref)
    assert fg.captures == 2 and fg.replays == 6
    sd = model.state_dict()
    assert int(sd["encoder.bn1.num_batches_tracked"]) == 0                    # eval mode: nothing was updated


def test_batched_and_graphed_evaluation_writes_the_same_files(ops, tmp_path):
    """``test_epoch_audio(batch_size=4, forward=ForwardGraphs(...))`` against the reference's one-clip-per-pass loop
    (batch_size=1, eager): five clips, four of one length and one longer (its own batch and its own graph) -- identical CSV
    files (text) and the same mean loss.  Evaluation-mode outputs do not depend on the batch a clip travels in."""
    import os
    from scipy.io import wavfile
    from adyolo_amd import test as atest
    from adyolo_amd.datasets import FoaDataset
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.graph import ForwardGraphs
    from adyolo_amd.postprocess import LabelPostProcessor
    from adyolo_amd.wrapper import WrapperCriterion, WrapperModel
    rs = np.random.RandomState(5)
    wdir, cdir = os.path.join(tmp_path, "foa_dev", "dev-test"), os.path.join(tmp_path, "metadata_dev", "dev-test")
    os.makedirs(wdir), os.makedirs(cdir)
    for i, n in enumerate([48000, 48000 + 77, 48000, 72000, 48000 + 401]):
        wavfile.write(os.path.join(wdir, "t%d.wav" % i), 24000, rs.randint(-8000, 8000, size=(n, 4)).astype(np.int16))
        with open(os.path.join(cdir, "t%d.csv" % i), "w") as f:
            for fr in range(i, 20, 3):
                f.write("%d,%d,0,%d,%d\n" % (fr, (fr + i) % 12, (fr * 53) % 360 - 180, (fr * 9) % 100 - 50))
    prm = _params()
    prm["data_config"]["data_pth"] = str(tmp_path)
    prm["train_config"].update({"conf_thresh": 0.3, "clss_thresh": 0.3})
    torch.manual_seed(8)
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    crit, post, fx = WrapperCriterion(prm), LabelPostProcessor(prm), FeatureExtractor(None, "cuda:0")
    ds = FoaDataset(prm, "test", is_valid=True)
    out_a, out_b = os.path.join(tmp_path, "out_one"), os.path.join(tmp_path, "out_batched")
    loss_a = atest.test_epoch_audio(ds, model, fx, crit, post, "cuda:0", out_a)
    fg = ForwardGraphs(model.eval(), fx, None, warm_calls=0)
    loss_b = atest.test_epoch_audio(ds, model, fx, crit, post, "cuda:0", out_b, batch_size=4, forward=fg)
    assert fg.captures >= 2 and abs(loss_a - loss_b) <= 1e-6 * abs(loss_a)
    names = ds.get_filelist()
    assert len(names) == 5
    for nm in names:
        a, b = open(os.path.join(out_a, nm + ".csv")).read(), open(os.path.join(out_b, nm + ".csv")).read()
        assert a == b and len(a) > 0, nm


def test_evaluation_caches_expire_when_parameters_change(ops):
    """Evaluation-mode BatchNorm affines are computed once and kept (``_BNState.eval_affine``), and ``ForwardGraphs`` records
    them into its graphs; both must expire when parameters or running statistics change -- through a train step (kernels
    write in place, no torch version bump) or through ``load_state_dict``.  Reference behaviour: evaluation after every
    epoch sees the weights of that epoch (src/train.py:178-215)."""
    from adyolo_amd import functional as Fn
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.graph import ForwardGraphs
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    torch.manual_seed(100)
    prm = _params()
    n = 24000 * 2
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    fx = FeatureExtractor(None, "cuda:0")
    tr = TrainStep(model, WrapperCriterion(prm), fx, prm, graph=False)
    audio = synthetic_audio(2, n, seed=5).to("cuda:0")
    target = synthetic_targets(2, n // 2400, 12, seed=5).to("cuda:0")
    clip = synthetic_audio(1, n, seed=6).to("cuda:0")

    def uncached():
        """the evaluation forward with every cache dropped first"""
        for m in model.modules():
            m.__dict__.pop("_adyolo_eval_affine", None)
        with torch.no_grad():
            return model(fx(clip, channels_last8=True), channels_last8=True).clone()

    model.eval()
    fg = ForwardGraphs(model, fx, None, warm_calls=1)
    outs = [fg(clip)[0].clone() for _ in range(3)]              # warm call, recorded call, replay
    assert fg.captures == 1 and torch.equal(outs[0], outs[2]) and torch.equal(outs[2], uncached())
    launches = []
    real = ops.bn_eval_stats
    ops.bn_eval_stats = lambda *a, **k: (launches.append(1), real(*a, **k))[1]
    try:
        with torch.no_grad():
            again = model(fx(clip, channels_last8=True), channels_last8=True)
        assert not launches and torch.equal(again, outs[2])     # served from the cache: no BatchNorm launch at all
        model.train()
        tr.step(audio, target)                                  # parameters and running statistics move (in place, by kernels)
        model.eval()
        after_step = fg(clip)[0].clone()                        # must NOT be the stale graph
        assert len(launches) == 36                              # every BatchNorm recomputed its affine once
        assert torch.equal(after_step, uncached()) and not torch.equal(after_step, outs[2])
        for _ in range(2):
            assert torch.equal(fg(clip)[0], after_step)         # recorded again, replayed
        caps = fg.captures
        # load_state_dict (copy_ into the same storage) expires the eager caches through the tensors' version counters and the
        # recorded graphs through WrapperModel's post-load hook
        sd = {k: (v * 0.5 if k.endswith("bn1.weight") else v) for k, v in model.state_dict().items()}
        model.load_state_dict(sd)
        for _ in range(3):
            loaded = fg(clip)[0].clone()
            assert torch.equal(loaded, uncached()) and not torch.equal(loaded, after_step)
        assert fg.captures == caps + 1
        # a parameter written in evaluation mode WITHOUT ops.params_changed() (p.copy_(): an EMA swap, a torch optimizer step)
        # only bumps the tensor's version counter: the recorded graph must expire as well (round 4, ADVICE)
        caps = fg.captures
        with torch.no_grad():
            w = model.encoder.layer4[0].conv1.weight
            w.copy_(w * 1.25)
        for _ in range(3):
            swapped = fg(clip)[0].clone()
            assert torch.equal(swapped, uncached()) and not torch.equal(swapped, loaded)
        assert fg.captures == caps + 1
    finally:
        ops.bn_eval_stats = real
    assert Fn is not None


def test_uncapturable_step_falls_back_to_eager(ops, monkeypatch):
    """``TrainStep(graph=True)`` on a model whose step draws host-computed values: the shape is marked eager-only with ONE
    warning and every step runs eagerly -- the losses of a trainer built with graph=False (round 4, ADVICE: it used to raise
    on every call).  Since the Conformer's attention-dropout seeds are derived on the device (below) nothing in the tree is
    uncapturable any more, so the case is made: ``DropoutStream.seed32`` is patched back to a host-side draw."""
    import warnings
    from adyolo_amd import rng
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    device_seed32 = rng.DropoutStream.seed32

    def host_seed32(self, n):
        self._no_capture("seed32")
        self.host_draws += 1
        return device_seed32(self, n)
    monkeypatch.setattr(rng.DropoutStream, "seed32", host_seed32)
    n = 24000 * 2
    audio = synthetic_audio(2, n, seed=9).to("cuda:0")
    target = synthetic_targets(2, n // 2400, 12, seed=9).to("cuda:0")
    losses = {}
    for graph in (False, True):
        torch.manual_seed(100)
        prm = _params()
        prm["args"]["encoder"] = "resnet-conformer"
        model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
        tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=graph)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            losses[graph] = [float(tr.step(audio, target)) for _ in range(4)]
        if graph:
            assert tr.graphs.captures == 0 and len(tr.graphs.eager_only) == 1 and tr.graphs.eager_steps == 4
            assert sum("not hipGraph-capturable" in str(w.message) for w in caught) == 1
    assert losses[True] == losses[False], losses


def test_runtime_error_inside_a_capture_falls_back_to_eager(ops, monkeypatch):
    """A RuntimeError raised WHILE the step is being recorded (a HIP error, an op that refuses capture, ...) must not leave the
    stream capturing or the trainer unusable: the capture is ended, the shape is remembered as eager-only with one warning, the
    step runs eagerly and the losses equal those of a trainer that never records."""
    import warnings
    from adyolo_amd import ops as _ops
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    n = 24000 * 2
    audio = synthetic_audio(2, n, seed=11).to("cuda:0")
    target = synthetic_targets(2, n // 2400, 12, seed=11).to("cuda:0")
    real_loss = _ops.adyolo_loss

    def loss_that_refuses_capture(*a, **kw):
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("injected failure inside the capture")
        return real_loss(*a, **kw)
    losses = {}
    for graph in (False, True):
        torch.manual_seed(100)
        prm = _params()
        model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
        tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=graph)
        if graph:
            monkeypatch.setattr(_ops, "adyolo_loss", loss_that_refuses_capture)
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            losses[graph] = [float(tr.step(audio, target)) for _ in range(4)]
        if graph:
            assert not torch.cuda.is_current_stream_capturing()
            assert tr.graphs.captures == 0 and len(tr.graphs.eager_only) == 1 and tr.graphs.eager_steps == 4
            assert sum("not hipGraph-capturable" in str(w.message) and "injected failure" in str(w.message) for w in caught) == 1
    assert losses[True] == losses[False], losses


def test_failure_inside_backward_of_a_capture(ops, monkeypatch):
    """Round 5 ADVICE: (a) an error CAUSED by recording that strikes in the middle of backward (gradient sink open, BatchNorm
    counters pending) still ends in a clean eager fallback with the eager trainer's losses; (b) an error that has nothing to do
    with recording (a bug, out of memory) is NOT swallowed into a permanent eager fallback: it propagates, the shape is not
    marked eager-only, the stream is not left capturing, and the trainer records and replays normally afterwards -- with the
    losses of a trainer that never failed (a recording runs nothing, so the failed attempt must not have advanced any state)."""
    import warnings
    from adyolo_amd import ops as _ops, functional as Fn
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    n = 24000 * 2
    audio = synthetic_audio(2, n, seed=13).to("cuda:0")
    target = synthetic_targets(2, n // 2400, 12, seed=13).to("cuda:0")
    real_wgrad = _ops.conv3x3_wgrad

    def make(graph):
        torch.manual_seed(100)
        prm = _params()
        model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
        return TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=graph)
    ref = make(False)
    want = [float(ref.step(audio, target)) for _ in range(4)]

    # (a) capture-related, raised by the 5th weight-gradient launch of the recorded backward
    calls = {"n": 0}

    def wgrad_refusing_capture(*a, **kw):
        if torch.cuda.is_current_stream_capturing():
            calls["n"] += 1
            if calls["n"] == 5:
                raise RuntimeError("HIP error: operation not permitted when stream is capturing (injected)")
        return real_wgrad(*a, **kw)
    tr = make(True)
    monkeypatch.setattr(_ops, "conv3x3_wgrad", wgrad_refusing_capture)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got = [float(tr.step(audio, target)) for _ in range(4)]
    assert got == want, (got, want)
    assert not torch.cuda.is_current_stream_capturing() and Fn.SINK.views is None and not Fn._COUNTER_SCOPE
    assert tr.graphs.captures == 0 and len(tr.graphs.eager_only) == 1
    assert sum("not hipGraph-capturable" in str(w.message) for w in caught) == 1

    # (b) NOT capture-related: propagates; afterwards the trainer captures and replays as if nothing had happened
    calls["n"] = 0

    def wgrad_with_a_bug(*a, **kw):
        if torch.cuda.is_current_stream_capturing():
            calls["n"] += 1
            if calls["n"] == 5:
                raise RuntimeError("injected bug: shapes do not match")
        return real_wgrad(*a, **kw)
    tr = make(True)
    monkeypatch.setattr(_ops, "conv3x3_wgrad", wgrad_with_a_bug)
    got = [float(tr.step(audio, target))]                        # warm-up step, eager
    with pytest.raises(RuntimeError, match="injected bug"):
        tr.step(audio, target)
    assert not torch.cuda.is_current_stream_capturing() and Fn.SINK.views is None and not Fn._COUNTER_SCOPE
    assert not tr.graphs.eager_only and tr.graphs.captures == 0
    monkeypatch.setattr(_ops, "conv3x3_wgrad", real_wgrad)
    got += [float(tr.step(audio, target)) for _ in range(3)]
    assert tr.graphs.captures == 1 and tr.graphs.replays == 3
    assert got == want, (got, want)


def test_recorded_graphs_are_bounded_lru(ops, monkeypatch):
    """``StepGraphs`` / ``ForwardGraphs`` keep at most ``graph.MAX_GRAPHS`` recorded graphs (each owns a private pool with its
    shape's activations): the least recently used one is evicted, an evicted shape is simply recorded again, and the outputs
    stay those of the eager path."""
    from adyolo_amd import graph as G
    from adyolo_amd.wrapper import WrapperModel
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio
    monkeypatch.setattr(G, "MAX_GRAPHS", 2)
    torch.manual_seed(100)
    prm = _params()
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    model.eval()
    fx = FeatureExtractor(None, "cuda:0")
    fg = G.ForwardGraphs(model, fx)
    clips = {sec: synthetic_audio(1, 24000 * sec, seed=sec).to("cuda:0") for sec in (1, 2, 3)}
    ref = {}
    with torch.no_grad():
        for sec, a in clips.items():
            ref[sec] = model(fx(a, channels_last8=True), channels_last8=True).clone()
    for sec in (1, 2, 3, 1, 1, 2, 3, 3, 1):                    # warm call + capture per shape, with evictions in between
        for _ in range(2):
            out, _dec = fg(clips[sec])
            assert torch.equal(out, ref[sec]), sec
        assert len(fg.entries) <= 2
    assert fg.evictions >= 2 and fg.captures >= 5 and list(fg.entries)[-1] == tuple(clips[1].shape)


def test_conformer_step_replays_from_a_graph(ops):
    """The ResNet-Conformer train step (config 4) recorded once and replayed: the attention-dropout seed, the only host-computed
    value of its step, is derived on the device from the stream's counter (``ops.seed32_dev``), and the max-pool backward is
    a gather (no atomics), so replayed steps equal eager steps bit for bit -- losses of five steps and the parameters after them."""
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    n = 24000 * 4
    audio = synthetic_audio(3, n, seed=19).to("cuda:0")
    target = synthetic_targets(3, n // 2400, 12, seed=19).to("cuda:0")
    out = {}
    for graph in (False, True):
        torch.manual_seed(100)
        prm = _params()
        prm["args"]["encoder"] = "resnet-conformer"
        model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
        tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=graph)
        losses = [float(tr.step(audio, target)) for _ in range(5)]
        out[graph] = (losses, tr.flat.flat.clone())
        if graph:
            assert tr.graphs.captures == 1 and tr.graphs.replays == 4 and not tr.graphs.eager_only
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert torch.equal(out[True][1], out[False][1])
