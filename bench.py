#!/usr/bin/env python3
"""Headline benchmark: train-step audio-seconds per second of the AD-YOLO hot path on MI355X.

A "step" is one full optimisation step on raw audio already resident in HBM:
  K1 features (STFT -> log-mel + intensity vector) -> SE-ResNet34+BiGRU encoder + AD-YOLO head forward ->
  AD-YOLO loss -> backward -> [bucketed RCCL all-reduce when N > 1] -> fused Adam.
Workload (BASELINE.json configs[1]): synthetic 4-ch 24 kHz 60 s clips, batch 64 per GPU, se-resnet34 + adyolo,
12 classes, random-init weights (seed 100), fp32 arithmetic (exact-fp32 MFMA).  Weak scaling: every rank
processes its own 64 clips.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seconds S]

With --gpus N > 1 and no RANK in the environment this process only LAUNCHES: it starts N fresh worker processes
(one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays rank 0's JSON line
and exits with the worst worker return code -- it never touches the GPU itself and never exec()s.  Under
`python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the workers are the torchrun children.

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events (torch.cuda.Event on the stream the
kernels are launched on) around every launch of the dominant kernel family (conv3x3 forward / data-gradient)
inside the timed region: `achieved` = matrix FLOPs ISSUED per second (Winograd issues 16/36 of the algorithmic
convolution FLOPs), `frac` = achieved / the exact-fp32 MFMA peak, `algorithmic_tflops` = the direct-convolution FLOPs
per second the same launches stand for.  `stages` (HBM-bound passes: GB/s against 8 TB/s) are timed with HIP events in
two extra, separately instrumented steps AFTER the timed region, so their events do not perturb `value`.
`cpu_baseline` times the CPU oracle (a port of the reference path, `oracle/`) on the host cores on a bounded sample
(8 x 20 s clips, 3 warm-ups, median of 5 steps; N = 1, rank 0 only).
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
PEAK_HBM_GBS = 8000.0


def params(device, nb_classes=12):
    return {"args": {"device": device, "encoder": "se-resnet34", "loss": "adyolo"},
            "data_config": {"nb_classes": nb_classes},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0,
                                            "class_gain": 3.0},
                             "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


# ---------------------------------------------------------------------------------------------- launcher (N > 1, no RANK)
def self_launch(n_gpus, argv):
    """Start one fresh worker per GPU as CHILD processes and relay rank 0's line.  Nothing in this process has
    touched (or will touch) the GPU: no torch.cuda call, no HIP call, no exec."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n_gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # poll ALL children: when any rank dies, the others would sit in RCCL init / a collective until the launcher's
    # deadline -- terminate them (exactly the PIDs started here) as soon as one exits non-zero
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("ADYOLO_BENCH_LAUNCH_TIMEOUT", "1500"))
    failed = False
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    out = b"".join(c for c in chunks if c)
    rcs = [p.returncode if p.returncode is not None else -9 for p in procs]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return 0 if not bad else (max(abs(rc) for rc in bad) or 1)


# ---------------------------------------------------------------------------------------------- HIP-event timers
class TimingEvent:
    """A HIP event made for TIMING: ``hipEventDisableSystemFence`` (the runtime accepts one of its three release flags per event).  ``torch.cuda.Event`` is a default
    event: when it is recorded the runtime performs a system-scope release -- the L2 is written back and invalidated -- which costs
    the command processor microseconds per record and leaves the next kernel a cold cache (hip_runtime_api.h: "On some AMD GPU
    devices this can improve the accuracy of timing measurements by avoiding the cost of cache writeback and invalidation, and the
    performance impact of those actions on the execution of following work").  Same API underneath (hipEventRecord on the stream
    the kernels are launched on, hipEventElapsedTime), created through the HIP runtime PyTorch has already loaded."""
    _hip = None
    FLAGS = 0x20000000

    @classmethod
    def runtime(cls):
        if cls._hip is None:
            path = None
            with open("/proc/self/maps") as f:                       # the ONE runtime of this process (PyTorch ships its own copy)
                for ln in f:
                    if "libamdhip64" in ln:
                        path = ln.split()[-1]
                        break
            if path is None:
                raise RuntimeError("no HIP runtime is mapped (import torch and touch the GPU first)")
            hip = ctypes.CDLL(path)
            hip.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
            hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
            hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
            hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
            cls._hip = hip
        return cls._hip

    def __init__(self, torch):
        self.torch = torch
        self.h = ctypes.c_void_p()
        rc = self.runtime().hipEventCreateWithFlags(ctypes.byref(self.h), self.FLAGS)
        if rc != 0:
            raise RuntimeError("hipEventCreateWithFlags failed (%d)" % rc)

    def record(self):
        rc = self._hip.hipEventRecord(self.h, ctypes.c_void_p(self.torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError("hipEventRecord failed (%d)" % rc)

    def elapsed_time(self, other):
        self._hip.hipEventSynchronize(other.h)
        ms = ctypes.c_float()
        rc = self._hip.hipEventElapsedTime(ctypes.byref(ms), self.h, other.h)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed (%d)" % rc)
        return float(ms.value)

    def __del__(self):
        try:
            if self.h:
                self._hip.hipEventDestroy(self.h)
        except Exception:                                               # noqa: BLE001  (interpreter shutdown)
            pass


EVENT_KIND_USED = {"kind": None, "fallback": None}


def make_event(torch, kind="timing"):
    """kind 'timing': TimingEvent (no system-scope fence); 'torch': torch.cuda.Event(enable_timing=True) -- the A/B switch
    ``--event-kind`` of the bench.  Should the runtime refuse the flagged event (another ROCm release), every later call takes
    torch.cuda.Event and the line says so (``event_kind``)."""
    if kind == "timing" and EVENT_KIND_USED["fallback"] is None:
        try:
            ev = TimingEvent(torch)
            EVENT_KIND_USED["kind"] = "hipEventDisableSystemFence"
            return ev
        except Exception as exc:                                        # noqa: BLE001
            EVENT_KIND_USED["fallback"] = "%s: %s" % (type(exc).__name__, exc)
    EVENT_KIND_USED["kind"] = "torch.cuda.Event"
    return torch.cuda.Event(enable_timing=True)


class KernelTimer:
    """HIP-event timing of selected op families on the launch stream, only while `active`."""

    def __init__(self, torch, kind="timing"):
        self.torch = torch
        self.kind = kind
        self.active = False
        self.records = {}          # family -> list of (start, end, work)

    def wrap(self, module, name, family, work_fn, gate=None):
        orig = getattr(module, name)

        def timed(*a, **kw):
            if not (self.active if gate is None else gate()):
                return orig(*a, **kw)
            s, e = make_event(self.torch, self.kind), make_event(self.torch, self.kind)
            s.record()
            out = orig(*a, **kw)
            e.record()
            self.records.setdefault(family, []).append((s, e, work_fn(*a, **kw)))
            return out
        setattr(module, name, timed)

    def summary(self, family, tag=None):
        """-> (launches, total ms, sum of work[0], sum of work[1]); tag: only the records whose work tuple carries it as third item"""
        recs = [r for r in self.records.get(family, []) if tag is None or (len(r[2]) > 2 and r[2][2] == tag)]
        if not recs:
            return 0, 0.0, 0.0, 0.0
        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
        return len(recs), ms, float(sum(w[0] for _, _, w in recs)), float(sum(w[1] for _, _, w in recs))

    def tags(self, family):
        return sorted({r[2][2] for r in self.records.get(family, []) if len(r[2]) > 2})


_KERNEL_OF = {36: "wino4_fwd_kernel", 16: "wino_fwd_kernel"}       # (F(4x4,3x3) one-patch form / F(2x2,3x3)); the persistent
#                                                                       F(4x4) form, wino4p_fwd_kernel, is recognised after the launch


def _issued_share(wpk):
    """matrix FLOPs issued / direct-convolution FLOPs for a packed filter: Winograd F(4x4,3x3) 36 multiplies per 144 (first
    axis 36), F(2x2,3x3) 16 per 36 (first axis 16), direct form 1"""
    if not hasattr(wpk, "dim") or wpk.dim() != 4:
        return 1.0
    return 9.0 / 36.0 if wpk.shape[0] == 36 else 16.0 / 36.0


def _picked(ops_mod, x, wpk, cout, addend=False):
    """the packed form ``ops.conv3x3`` will launch with (``ops.DualPack``: F(4x4) or F(2x2) by the size of the launch)"""
    return wpk.pick(x.shape[0], x.shape[1], x.shape[2], cout, addend) if isinstance(wpk, ops_mod.DualPack) else wpk


def load_traffic(algo):
    """HBM bytes per launch of the dominant kernel family from the committed PMC passes (profiles/r0N_traffic.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE over this same command, corrected as MI355X_MICROARCH.md prescribes).  Counters
    cannot be collected inside the timed run: the figure is READ from the committed file, whose name is returned beside it
    (``roofline.traffic_source``); (None, None) when no PMC summary matches the algorithm."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            v = t.get(algo, {}).get("hbm_bytes_per_launch")
            if v is not None:
                return v, "profiles/" + name
        except (OSError, ValueError):
            continue
    return None, None


# ---------------------------------------------------------------------------------------------- CPU baseline (oracle)
def cpu_baseline(batch=8, clip_seconds=20, warmups=3, timed=5, budget_s=75.0):
    """Oracle (CPU port of the reference path) train step, BASELINE.md section 3: `batch` x 20 s clips per step
    (BASELINE config 1 = the reference's own CPU-runnable case), `warmups` discarded steps, median of `timed` steps,
    feature / model split, thread count chosen by a quick sweep.  Bounded: the sweep and the step counts shrink when a
    step is slower than the budget allows."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from oracle import features as ofeat, seresnet as onet, adyolo_loss as oloss
    from oracle.filler import fill_state_dict
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    t_begin = time.time()
    torch.manual_seed(100)
    ncpu = os.cpu_count() or 1
    sd = fill_state_dict(onet.state_dict_spec())
    plist = [v.requires_grad_(True) for k, v in sd.items()
             if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.Adam(plist, lr=1e-3)
    n = 24000 * clip_seconds
    audio = synthetic_audio(batch, n, seed=1234)
    target = synthetic_targets(batch, n // 2400, 12, seed=1234)
    audio64 = [audio[b].double().numpy() for b in range(batch)]
    mel = ofeat.mel_filterbank()
    pool = ThreadPoolExecutor(max_workers=min(batch, ncpu))      # the reference computes features in 16 loader workers

    def features():
        return np.stack(list(pool.map(lambda a: ofeat.get_feature(a, None, mel)[0], audio64)))

    def model_step(feats):
        logits = onet.model_forward(sd, torch.from_numpy(feats), training=True, update_stats=True)
        opt.zero_grad()
        loss = oloss.adyolo_loss(logits, target, 12)
        loss.backward()
        opt.step()
        return float(loss.detach())

    feats = features()
    torch.set_num_threads(min(ncpu, 32))
    model_step(feats)                                            # allocator / oneDNN primitive caches
    # thread sweep: one model step per candidate, keep the fastest (oversubscription hurts on 128+ hardware threads)
    # (ascending, stop at the first candidate that is slower than the best so far: 256 threads on B = 8 took 15 minutes)
    cands = sorted({c for c in (8, 16, 32, 64) if 1 <= c <= ncpu})
    sweep = {}
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.time()
        model_step(feats)
        sweep[c] = time.time() - t0
        if sweep[c] > 1.15 * min(sweep.values()) or time.time() - t_begin > 0.4 * budget_s:
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    left = budget_s - (time.time() - t_begin)
    est = sweep[best] * 1.3 + 1.0
    n_timed = max(1, min(timed, int(left / est) - 1))
    n_warm = max(0, min(warmups, int(left / est) - n_timed))
    for _ in range(n_warm):
        model_step(features())
    tf, tm = [], []
    for _ in range(n_timed):
        t0 = time.time()
        feats = features()
        t1 = time.time()
        model_step(feats)
        t2 = time.time()
        tf.append(t1 - t0)
        tm.append(t2 - t1)
    pool.shutdown()
    med = lambda v: float(sorted(v)[len(v) // 2])      # noqa: E731
    step_s = med([a + b for a, b in zip(tf, tm)])
    return {"value": round(batch * clip_seconds / step_s, 3), "unit": "audio-s/s", "cores": best, "kind": "port",
            "features_s": round(med(tf), 4), "model_s": round(med(tm), 4), "step_s": round(step_s, 4),
            "thread_sweep_s": {str(k): round(v, 3) for k, v in sweep.items()},
            "sample": "median of %d train steps (after %d warm-ups) of %d x %d s clips (features on %d host threads + "
                      "fwd+loss+bwd+Adam on %d torch threads), PyTorch-CPU/NumPy oracle, %d host cpus"
                      % (n_timed, n_warm + 1, batch, clip_seconds, min(batch, ncpu), best, ncpu)
                      + "; ONE process (the reference would run 16 DataLoader workers for the features beside the model, "
                        "src/configs/hyp_train.yaml:4)"}


# ---------------------------------------------------------------------------------------------- the reference's own shapes
# Small shapes are launch-bound in eager mode (~770 launches per step through ctypes + autograd); they run from a hipGraph
# (adyolo_amd/graph.py).  kernel_ms = sum of kernel durations per step from the committed rocprofv3 kernel trace of
# `bench.py --only-extra <name>` (profiles/r03_small_shapes.json, tools/small_shapes_profile.sh); wall / kernel says how
# much of the step is still not GPU work.
EXTRA_CONFIGS = {
    "train_bs16x20s": dict(kind="train", encoder="se-resnet34", batch=16, seconds=20, steps=10,
                           ref="the reference's training shape: batch_size 16 (src/configs/hyp_train.yaml:3) x 20 s chunks"),
    "train_bs8x20s": dict(kind="train", encoder="se-resnet34", batch=8, seconds=20, steps=10,
                          ref="BASELINE.json configs[0]: bs = 8 x 20 s (the CPU-runnable plumbing case; cpu_baseline's shape)"),
    "eval_bs1x60s": dict(kind="eval", encoder="se-resnet34", batch=1, seconds=60, steps=20,
                         ref="test_epoch: one 60 s clip at a time (src/train.py:130-133, src/test.py:81): K1 + forward + decode"),
    "eval_bs8x60s": dict(kind="eval", encoder="se-resnet34", batch=8, seconds=60, steps=10,
                         ref="the same evaluation with eight equal-length clips per forward pass (test.test_epoch_audio(batch_size=8): "
                             "evaluation-mode outputs do not depend on the batch, same CSV files)"),
    "config5_mic_adpit_bs64x20s": dict(kind="train", encoder="se-resnet34", batch=64, seconds=20, steps=4, loss="adpit", mic=True,
                                       ref="BASELINE.json configs[4]: se-resnet34 + multi-ACCDOA (ADPIT) loss, MIC-format audio -> "
                                           "4 log-mel + 6 GCC-PHAT features (10-channel stem), bs = 64 x 20 s; GCC-PHAT is not in the "
                                           "reference (parity unpinned)"),
    "conformer_bs32x20s": dict(kind="train", encoder="resnet-conformer", batch=32, seconds=20, steps=4,
                               ref="BASELINE.json configs[3]: resnet-conformer + adyolo, bs = 32 x 20 s"),
}


def _small_shape_kernel_ms(name):
    """-> (sum of the kernel durations of one step from the committed rocprofv3 run, the file it was read from)"""
    for fn in ("r06_small_shapes.json", "r05_small_shapes.json", "r04_small_shapes.json", "r03_small_shapes.json"):
        try:
            with open(os.path.join(ROOT, "profiles", fn)) as f:
                v = json.load(f).get(name, {}).get("kernel_ms_per_step")
            if v is not None:
                return v, "profiles/" + fn
        except (OSError, ValueError):
            pass
    return None, None


def run_extra_config(name, torch, modes="both"):
    return _run_extra_config(name, EXTRA_CONFIGS[name], torch, modes)


def _run_extra_config(name, cfg, torch, modes):
    import adyolo_amd  # noqa: F401
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    from adyolo_amd.graph import ForwardGraphs
    from adyolo_amd.postprocess import LabelPostProcessor
    device = "cuda:%d" % torch.cuda.current_device()
    B, n = cfg["batch"], 24000 * cfg["seconds"]
    T = n // 600
    prm = params(device)
    prm["args"]["encoder"] = cfg["encoder"]
    prm["args"]["loss"] = cfg.get("loss", "adyolo")
    prm["train_config"].update(conf_thresh=0.5, clss_thresh=0.5, unify_thresh=15.0, nms="conn-merge")
    n_feat = 7
    fx = FeatureExtractor(None, device)
    audio = synthetic_audio(B, n, seed=4321).to(device)
    target = synthetic_targets(B, T // 4, 12, seed=4321).to(device)
    if cfg.get("mic"):                      # MIC feature set: 4 log-mel + 6 GCC-PHAT channels in 32-channel pixels
        from adyolo_amd.features import MicFeatureExtractor
        mfx = MicFeatureExtractor(None, device)
        fx = lambda a, channels_last8=True: mfx(a, channels_last=channels_last8)        # noqa: E731
        n_feat = 10
    if cfg.get("loss") == "adpit":          # dense (B, T', 6, 4, C) activity / direction targets
        import numpy as _np
        rng = _np.random.default_rng(4321)
        tgt = _np.zeros((B, T // 4, 6, 4, 12), dtype=_np.float32)
        act = rng.random((B, T // 4, 12)) < 0.05
        xyz = rng.normal(size=(B, T // 4, 3, 12)).astype(_np.float32)
        xyz /= _np.linalg.norm(xyz, axis=2, keepdims=True)
        tgt[:, :, 0, 0, :] = act
        tgt[:, :, 0, 1:, :] = xyz * act[:, :, None, :]
        target = torch.from_numpy(tgt).to(device)
    graphable = True                        # (round 4: the Conformer step records too -- attention-dropout seeds derived on the device)
    staged = cfg["encoder"] != "se-resnet34"    # the Conformer entry carries a stage table from two instrumented eager steps
    ent = {"workload": cfg["ref"], "batch": B, "clip_seconds": cfg["seconds"], "steps": cfg["steps"]}

    def timed(fn, k, warm):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            out = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3, out

    if cfg["kind"] == "train":
        res = {}
        extra_steps = 0
        want = ("hipgraph", "eager") if modes == "both" else (modes,)
        for mode in ([m for m in want if graphable or m == "eager"] or ["eager"]):
            torch.manual_seed(100)
            model = WrapperModel((1, n_feat, T, 64), (), prm).to(device)
            tr = TrainStep(model, WrapperCriterion(prm), fx, prm, graph=(mode == "hipgraph"))
            ms, loss = timed(lambda: tr.step(audio, target), cfg["steps"], 3)
            res[mode] = (ms, float(loss.reshape(-1)[0]))
            del tr, model
        if staged and modes != "hipgraph":
            # resnet-conformer: where the step goes, family by family (HIP events in two extra steps; FLOPs = the matrix
            # products each launch stands for)
            from adyolo_amd import ops as _ops
            kt = KernelTimer(torch)
            fl = lambda v: (float(v), 1.0)                                              # noqa: E731
            kt.wrap(_ops, "conv_gemm", "implicit-GEMM convolutions (fwd / dgrad / wgrad)",
                    lambda mode_, src, other, n_, h, w, cin, cout, kh, kw, sh, sw, ph, pw:
                    fl(2.0 * n_ * _ops.conv_out_hw(h, w, kh, kw, sh, sw, ph, pw)[0] * _ops.conv_out_hw(h, w, kh, kw, sh, sw, ph, pw)[1]
                       * cout * kh * kw * cin))
            kt.wrap(_ops, "gemm", "plain GEMMs (linear layers, projections)", lambda a_, b_, m, n_, k_, *r, **kw: fl(2.0 * m * n_ * k_))
            kt.wrap(_ops, "attn_fwd", "attention forward", lambda q, k, v, heads, *a, **kw: fl(4.0 * q.shape[0] * q.shape[1] ** 2 * q.shape[2]))
            kt.wrap(_ops, "attn_bwd", "attention backward (7 products)",
                    lambda q, *a, **kw: fl(14.0 * q.shape[0] * q.shape[1] ** 2 * q.shape[2]))
            kt.wrap(_ops, "conv3x3", "Winograd 3x3 forward / dgrad [issued FLOPs]",
                    lambda x, wpk, cout, **kw: fl(2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * x.shape[3] * (_issued_share(_picked(_ops, x, wpk, cout, kw.get("addend") is not None)))))
            kt.wrap(_ops, "conv3x3_wgrad", "Winograd 3x3 weight-gradient [issued FLOPs]",
                    lambda x, dy, cin_real, **kw: fl(2.0 * x.shape[0] * x.shape[1] * x.shape[2] * dy.shape[3] * 9 * x.shape[3]
                                                     * _ops.wgrad_form(x.shape[3], dy.shape[3], kw.get("algo"), (x.shape[0], x.shape[1], x.shape[2]))[1]))
            torch.manual_seed(100)
            model = WrapperModel((1, 7, T, 64), (), prm).to(device)
            tr = TrainStep(model, WrapperCriterion(prm), fx, prm, graph=False)
            for _ in range(2):
                tr.step(audio, target)
            torch.cuda.synchronize()
            kt.active = True
            for _ in range(2):
                tr.step(audio, target)
            torch.cuda.synchronize()
            kt.active = False
            fam = {}
            for name_ in kt.records:
                n_l, ms_f, work, _ = kt.summary(name_)
                fam[name_] = {"launches_per_step": n_l // 2, "ms_per_step": round(ms_f / 2, 3),
                              "tflops": round(work / (ms_f * 1e-3) / 1e12, 1) if ms_f > 0 else 0.0,
                              "frac_of_mfma_peak": round(work / (ms_f * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3) if ms_f > 0 else 0.0}
            ent["stages"] = fam
            extra_steps = 4                     # (the two warm-up and two instrumented steps above: same kernels as a timed step)
            del tr, model
        mode = "hipgraph" if "hipgraph" in res else "eager"
        ms = res[mode][0]
        ent["steps_executed"] = {m: cfg["steps"] + 3 + (extra_steps if m == "eager" else 0) for m in res}
        ent.update({"mode": mode, "ms_per_step": round(ms, 3), "audio_s_per_s": round(B * cfg["seconds"] / (ms * 1e-3), 1),
                    "final_loss": round(res[mode][1], 6)})
        if len(res) == 2:
            ent["eager_ms_per_step"] = round(res["eager"][0], 3)
            ent["graph_equals_eager_loss"] = res["hipgraph"][1] == res["eager"][1]
    else:
        torch.manual_seed(100)
        model = WrapperModel((1, 7, T, 64), (), prm).to(device)
        model.eval()
        post = LabelPostProcessor(prm)
        fg = ForwardGraphs(model, fx, post)
        from adyolo_amd import ops as _ops

        def graphed():
            o, d = fg(audio)
            return o, _ops.to_host(d)                # the decoded tensor goes to the host for the NMS (page-locked staging, like postprocess.decode)
        ms, outs = timed(graphed, cfg["steps"], 3)

        def eager():
            with torch.no_grad():
                o = model(fx(audio, channels_last8=True), channels_last8=True)
                d = _ops.yolo_decode(o.contiguous(), post.nb_classes, post.nb_grids, post.nb_anchors, post.grid_size, post.g_overlap)
                return o, _ops.to_host(d)
        ms_e, outs_e = timed(eager, max(3, cfg["steps"] // 2), 2)
        ent["steps_executed"] = {"hipgraph": cfg["steps"] + 3, "eager": max(3, cfg["steps"] // 2) + 2}
        ent.update({"mode": "hipgraph", "ms_per_step": round(ms, 3), "clips_per_s": round(B / (ms * 1e-3), 1),
                    "audio_s_per_s": round(B * cfg["seconds"] / (ms * 1e-3), 1), "eager_ms_per_step": round(ms_e, 3),
                    "graph_equals_eager_logits": bool(torch.equal(outs[0], outs_e[0])),
                    "note": "both figures include the device-to-host copy of the decoded tensor through a page-locked staging buffer (the NMS runs on the host)"})
    k_ms, k_src = _small_shape_kernel_ms(name)
    # NOT measured in this run: kernel time of one step summed from a committed rocprofv3 trace (possibly of an earlier round)
    ent["kernel_ms_per_step"] = {"value": k_ms, "source": k_src}
    ent["wall_over_kernel"] = round(ent["ms_per_step"] / k_ms, 3) if k_ms else None
    torch.cuda.empty_cache()
    return ent


# ---------------------------------------------------------------------------------------------- worker
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--seconds", type=int, default=60, help="clip length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stages", action="store_true", help="skip the two extra instrumented steps")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs block (the reference's own shapes)")
    ap.add_argument("--only-extra", default=None, help="run ONE extra config by name and print its entry (profiling)")
    ap.add_argument("--extra-mode", default="both", choices=["both", "hipgraph", "eager"],
                    help="with --only-extra: run only the graph or only the eager variant (profiling: a known step count)")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the parity gate (profiling: its 8-clip launches would enter the per-kernel averages of the trace)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false",
                    help="skip the host-fed variant of the step (int16 clips through AudioStager), reported as `pipeline`")
    ap.add_argument("--graph", action="store_true", help="replay the headline step from a hipGraph too (no per-kernel events)")
    ap.add_argument("--event-kind", default="timing", choices=["timing", "torch"],
                    help="timing: HIP events created with hipEventDisableSystemFence (no L2 write-back / invalidate "
                         "per record); torch: torch.cuda.Event (default flags: a system-scope release per record)")
    ap.add_argument("--events", default="dominant", choices=["dominant", "all"],
                    help="HIP events inside the TIMED steps: around every 3x3 forward / data-gradient launch (the family of the dominant kernel: "
                         "`roofline`) or also around the weight gradients and K1 (`all`: +~100 launches x 2 events x ~6.5 us of GPU idle per step)")
    ap.add_argument("--encoder", default="se-resnet34", choices=["se-resnet34", "resnet-conformer"],
                    help="se-resnet34 = the headline workload (BASELINE configs[1]); resnet-conformer = config 4")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops, dist as adist
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    import torch.distributed as dist

    if args.only_extra:
        torch.cuda.set_device(0)
        print(json.dumps(run_extra_config(args.only_extra, torch, args.extra_mode)), flush=True)
        return

    # ADYOLO_DIST_BACKEND=gloo ADYOLO_BENCH_ONE_DEVICE=1: every rank on cuda:0 over gloo -- a FUNCTIONAL run of the N > 1 path
    # on a single-GPU box (RCCL refuses two ranks on one device); the numbers of such a run mean nothing
    one_device = os.environ.get("ADYOLO_BENCH_ONE_DEVICE") == "1"
    if one_device:
        os.environ["LOCAL_RANK"] = "0"
    rank, world, local_rank = adist.init_from_env(os.environ.get("ADYOLO_DIST_BACKEND", "nccl"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    B, n_samples = args.batch, 24000 * args.seconds
    T = n_samples // 600
    torch.manual_seed(100)
    prm = params(device)
    prm["args"]["encoder"] = args.encoder
    model = WrapperModel((1, 7, T, 64), (), prm).to(device)
    criterion = WrapperCriterion(prm)
    fx = FeatureExtractor(None, device)
    audio = synthetic_audio(B, n_samples, seed=1234 + rank).to(device)
    target = synthetic_targets(B, T // 4, 12, seed=1234 + rank).to(device)
    conv_algo = ops.conv_algo()
    wino = conv_algo in ("winograd", "winograd4")

    # parity gate at the benchmark's clip shape on a SLICE of the batch (12 clips: every layer still takes the kernel the full
    # batch takes -- ops.DualPack picks by launch size, the 128 -> 64 data-gradient needs 11 clips; round 3 ran it on all 64
    # clips, 40 GB of allocator churn before the timed region): the first forward loss with the benchmarked (Winograd) convolutions must equal
    # the direct implicit-GEMM path within 1e-3 (both are checked against torch / the oracle in tests/)
    parity = None
    if args.encoder == "se-resnet34" and wino and world == 1 and not args.no_parity:      # (N > 1: a rank-0-only assert would strand the other ranks)
        model.train()
        vals, outs = {}, {}
        nb = min(B, 12)
        tgt_slice = synthetic_targets(nb, T // 4, 12, seed=1234 + rank).to(device)
        with torch.no_grad():
            feat = fx(audio[:nb], channels_last8=True)
            for algo in (conv_algo, "direct"):
                os.environ["ADYOLO_CONV_ALGO"] = algo
                model.encoder.dropout_stream.offset = 0    # same inter-layer GRU dropout mask for both runs
                outs[algo] = model(feat, channels_last8=True)
                vals[algo] = float(criterion(outs[algo], tgt_slice))
            del feat
            logit_rel = float((outs[conv_algo] - outs["direct"]).abs().max() / outs["direct"].abs().max())
            del outs
        os.environ["ADYOLO_CONV_ALGO"] = conv_algo
        rel = abs(vals[conv_algo] - vals["direct"]) / abs(vals["direct"])
        parity = {"algo": conv_algo, "clips": nb, "first_loss": round(vals[conv_algo], 6), "first_loss_direct": round(vals["direct"], 6),
                  "rel_diff": float("%.3g" % rel), "tol": 1e-3, "logits_max_diff_of_absmax": float("%.3g" % logit_rel), "logits_tol": 3e-4}
        assert rel <= 1e-3 and logit_rel <= 3e-4, "%s vs direct at the bench shape: %r, logits %.3e" % (conv_algo, vals, logit_rel)
        del tgt_slice
        torch.manual_seed(100)                     # BatchNorm running statistics moved: rebuild the model
        model = WrapperModel((1, 7, T, 64), (), prm).to(device)
        torch.cuda.empty_cache()
    trainer = TrainStep(model, criterion, fx, prm, graph=args.graph)

    timer = KernelTimer(torch, args.event_kind)
    # work = (algorithmic FLOPs of the 3x3 convolution, matrix FLOPs actually issued: 16/36 of that in Winograd form)

    from adyolo_amd import _lib as _alib

    def conv_work(x, wpk, cout, **kw):
        # (called right after the launch: which F(4x4) form it took is asked from the library)
        alg = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * x.shape[3]
        pk = wpk.pick(x.shape[0], x.shape[1], x.shape[2], cout, kw.get("addend") is not None) if isinstance(wpk, ops.DualPack) else wpk
        tag = _KERNEL_OF.get(pk.shape[0] if hasattr(pk, "dim") and pk.dim() == 4 else 0, "conv3x3_fwd_kernel")
        if tag == "wino4_fwd_kernel" and _alib.load().adyolo_wino4_last_form() == 2:
            tag = "wino4p_fwd_kernel"
        return alg, alg * _issued_share(pk), tag

    def wgrad_work(x, dy, cin_real, **kw):
        alg = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * dy.shape[3] * 9 * x.shape[3]
        form, share = ops.wgrad_form(x.shape[3], dy.shape[3], kw.get("algo"), (x.shape[0], x.shape[1], x.shape[2]))
        return alg, alg * share, form
    # HIP events cost the GPU's command processor ~6.5 us each (a barrier packet with a completion signal: 170-230 idle gaps of that
    # size per step in profiles/r06_bench_b64x60s_kernel_trace gaps, one per recorded event = 1.2-1.5 ms per step).  The TIMED region
    # therefore records what the contract asks for -- the forward / data-gradient family with the dominant kernel, every launch --
    # and the weight gradients and K1 are timed in the extra instrumented steps with the HBM-bound passes (--events all: as before)
    stage = {"on": False}
    ev_all = args.events == "all"
    timer.wrap(ops, "conv3x3", "conv3x3_fwd_dgrad", conv_work)
    timer.wrap(ops, "conv3x3_wgrad", "conv3x3_wgrad", wgrad_work, None if ev_all else (lambda: stage["on"]))
    feat_call = trainer.features.__call__
    k1_rec = []

    def timed_features(a, channels_last8=True):
        if not (timer.active if ev_all else stage["on"]):
            return feat_call(a, channels_last8)
        s, e = make_event(torch, args.event_kind), make_event(torch, args.event_kind)
        s.record()
        out = feat_call(a, channels_last8)
        e.record()
        k1_rec.append((s, e))
        return out
    trainer.features = timed_features

    # HBM-bound passes, timed only in the extra instrumented steps (work = (algorithmic bytes, 0)): every distinct operand
    # tensor read once + every output written once, whatever number of passes the implementation makes
    gate = lambda: stage["on"]                                                        # noqa: E731
    nb = lambda t: float(t.numel() * 4)                                               # noqa: E731
    timer.wrap(ops, "bn_bwd", "bn_bwd (reduce + apply)", lambda dy, x, *a, **k: (3 * nb(x), 0.0), gate)
    timer.wrap(ops, "se_tail_fwd", "se_tail_fwd", lambda c, r, *a, **k: (3 * nb(c), 0.0), gate)
    timer.wrap(ops, "se_tail_bwd", "se_tail_bwd (reduce + fc + apply)",
               lambda de, e, c, *a, **k: ((5 if k.get("want_dr", True) else 4) * nb(c), 0.0), gate)
    timer.wrap(ops, "affine", "affine", lambda x, *a, **k: (2 * nb(x), 0.0), gate)
    timer.wrap(ops, "avgpool2", "avgpool2_fwd", lambda x, *a, **k: (1.25 * nb(x), 0.0), gate)
    timer.wrap(ops, "avgpool2_bwd", "avgpool2_bwd", lambda dy, *a, **k: (5 * nb(dy), 0.0), gate)
    timer.wrap(ops, "adyolo_loss", "adyolo_loss (assign + main + final)",
               lambda logit, tgt, *a, **k: (2 * nb(logit) + nb(tgt), 0.0), gate)
    timer.wrap(ops, "adam_step_dev", "adam", lambda p, *a, **k: (7 * nb(p), 0.0), gate)
    timer.wrap(ops, "bn_stats_tiles", "bn_stats_tiles", lambda st, *a, **k: (nb(st), 0.0), gate)
    timer.wrap(ops, "sap_fwd", "sap_fwd", lambda x, *a, **k: (nb(x), 0.0), gate)
    timer.wrap(ops, "sap_bwd", "sap_bwd", lambda dy, x, *a, **k: (2 * nb(x), 0.0), gate)
    timer.wrap(ops, "gemm", "gemm (1x1 conv, GRU / head projections) [FLOPs]",
               lambda a_, b_, m, n, k_, *r, **kw: (2.0 * m * n * k_, 1.0), gate)
    timer.wrap(ops, "gru_fwd", "gru_fwd", lambda gx, *a, **k: (nb(gx), 0.0), gate)
    timer.wrap(ops, "gru_bwd", "gru_bwd", lambda d, g_, *a, **k: (nb(g_), 0.0), gate)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.step(audio, target)
    sync()
    timer.active = not args.graph                  # (HIP events cannot be recorded around launches that are replayed from a graph)
    fired0 = (trainer.reducer.fired_from_hooks, trainer.reducer.fired_from_finish)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(audio, target)
    sync()
    dt = time.perf_counter() - t0
    timer.active = False
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)
    loss_val = float(loss.detach())
    step_ms = dt / args.steps * 1e3

    def timed_steps(step_fn, k, w):
        """w untimed + k timed calls, barrier + synchronize on both sides, max over ranks -> ms per step"""
        for i in range(w):
            step_fn(i)
        sync()
        t_a = time.perf_counter()
        for i in range(k):
            step_fn(w + i)
        sync()
        d = time.perf_counter() - t_a
        if world > 1:
            tm = torch.tensor([d], device=device, dtype=torch.float64)
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            d = float(tm)
        return d / k * 1e3

    # ---- N > 1: what the process group and the gradient reducer actually did -- read it from the line,
    # not from prose: the backend torch.distributed runs on, the world it sees, a one-element all-reduce that must sum to N,
    # how many bucket all-reduces per step were issued from gradient hooks (overlapped with the backward pass) against from
    # finish() (exposed), and the same step with the reducer switched off (its difference to ms_per_step = exposed all-reduce)
    rccl = None
    if world > 1:
        red = trainer.reducer
        ones = torch.ones(1, device=device)
        dist.all_reduce(ones)
        hooks_n, finish_n = red.fired_from_hooks - fired0[0], red.fired_from_finish - fired0[1]
        off_ms = None                               # (measured at the very end: see "reducer off" below)
        try:
            lib_ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:                                                              # noqa: BLE001
            lib_ver = None
        rccl = {"backend": dist.get_backend(), "library_version": lib_ver, "world_size": dist.get_world_size(),
                "all_reduce_of_ones": float(ones), "all_reduce_ok": bool(float(ones) == world),
                "buckets": len(red.buckets), "bucket_MB": [round((e_ - s_) * 4 / 1e6, 2) for s_, e_, _ in red.buckets],
                "buckets_fired_from_hooks_per_step": round(hooks_n / args.steps, 2),
                "buckets_fired_from_finish_per_step": round(finish_n / args.steps, 2),
                "ms_per_step_reducer_off": None, "exposed_allreduce_ms": None}

    # ---- the same step fed the way a data loader feeds it (second weak-scaling mode): int16 clips in host memory ->
    # AudioStager's page-locked buffer (a worker thread, as DataLoader workers would) -> PCIe copy on a side stream, overlapped
    # with the step on the previous batch -> int16 -> float32 on the device -> the step.  `value` stays the resident-input rate.
    pipeline = None
    if args.pipeline:
        from adyolo_amd.datasets import AudioStager
        host = [torch.clamp(torch.round((audio.double() - 1e-8) * 32768.0), -32768, 32767).to(torch.int16).cpu()]   # the int16 the clips came from
        host.append(torch.flip(host[0], dims=[0]).contiguous())                    # a second batch: the same clips in reverse order
        stager = AudioStager(B, n_samples, device)
        stager.stage(host[0])

        # ONE long-lived staging thread (a thread's first HIP call binds it to the device: ~0.1 s, which a thread per step
        # would pay every step -- invisible behind a 133 ms step, 4 x a 23 ms one); the page-locked copy inside is split over 4 more
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=1)

        def piped(i):
            a = stager.get()
            fut = pool.submit(stager.stage, host[(i + 1) & 1], 4)
            out = trainer.step(a, target)
            fut.result()
            return out
        # (a launch-bound step keeps the GIL busy: with the default 5 ms switch interval every Python-level operation of the staging
        #  thread waits that long for it -- 20-50 ms per step at small shapes; 0.2 ms here, as a threaded input pipeline would set it)
        old_si = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
        try:
            pipe_ms = timed_steps(piped, args.steps, max(1, args.warmup))
        finally:
            sys.setswitchinterval(old_si)
        pipeline = {"input": "int16 clips in host memory -> pinned staging buffer (4 worker threads) -> H2D on a side stream -> "
                             "adyolo_pcm16_to_f32 -> step; copy of batch k+1 overlaps step k",
                    "ms_per_step": round(pipe_ms, 3), "value": round(world * B * args.seconds / (pipe_ms * 1e-3), 2), "unit": "audio-s/s",
                    "h2d_MB_per_step_per_gpu": round(B * n_samples * 4 * 2 / 1e6, 1), "vs_resident": round(step_ms / pipe_ms, 4)}
        pool.shutdown()
        del stager, host

    # ---- which kernel every 3x3 convolution of one step launches: counted in one extra step
    # (every rank runs these extra steps -- with N > 1 the reducer's all-reduces need all of them; rank 0 keeps the books)
    dispatch = None
    if not args.graph:
        ops.DISPATCH_LOG = {}
        trainer.step(audio, target)
        torch.cuda.synchronize()
        log, ops.DISPATCH_LOG = ops.DISPATCH_LOG, None
        by_name = {}
        for (name, cin, cout, epi), cnt in sorted(log.items()):
            by_name.setdefault(name, {"launches_per_step": 0, "by_shape": {}})
            by_name[name]["launches_per_step"] += cnt
            key = "%d->%d ops%d" % (cin, cout, epi)
            by_name[name]["by_shape"][key] = by_name[name]["by_shape"].get(key, 0) + cnt
        dispatch = {"conv3x3": by_name, "thresholds": ops.switch_table(),
                    "env": {k: os.environ.get(k) for k in ("ADYOLO_LIB", "ADYOLO_CONV_ALGO", "ADYOLO_W4_PERSIST", "ADYOLO_W4_MIN_K",
                                                           "ADYOLO_W4_MIN_K_ADDEND", "ADYOLO_W4_MIN_WGS",
                                                           "ADYOLO_WGRAD_ALGO", "ADYOLO_GEMM_TILE")},
                    "library": os.path.relpath(_alib.LIB_PATH, ROOT)}

    # ---- the encoder + head FORWARD alone, training mode (north_star: ">= 40 % MFMA utilisation on the SE-ResNet forward";
    # 2.269 GFLOP per audio-second algorithmic): HIP events around model(feat) on precomputed features
    encoder_fwd = None
    if not args.graph and args.encoder == "se-resnet34":
        model.train()
        feat = fx(audio, channels_last8=True)
        n_f0 = len(timer.records.get("conv3x3_fwd_dgrad", []))
        fwd_ms = []
        for i in range(5):
            timer.active = i >= 2
            s_e, e_e = make_event(torch, args.event_kind), make_event(torch, args.event_kind)
            s_e.record()
            out = model(feat, channels_last8=True)
            e_e.record()
            torch.cuda.synchronize()
            if i >= 2:
                fwd_ms.append(s_e.elapsed_time(e_e))
            del out
        timer.active = False
        recs = timer.records.get("conv3x3_fwd_dgrad", [])[n_f0:]
        del timer.records["conv3x3_fwd_dgrad"][n_f0:]                 # (they must not enter the step's roofline below)
        conv_alg = sum(w[0] for _, _, w in recs) / 3.0
        conv_issued = sum(w[1] for _, _, w in recs) / 3.0
        conv_ms = sum(a_.elapsed_time(b_) for a_, b_, _ in recs) / 3.0
        alg = 2.269e9 * B * args.seconds
        ms = sorted(fwd_ms)[1]
        issued_f = conv_issued + max(0.0, alg - conv_alg)             # GEMMs / GRU / stem issue what they compute
        encoder_fwd = {"ms": round(ms, 3), "algorithmic_TFLOP": round(alg / 1e12, 3),
                       "algorithmic_tflops": round(alg / (ms * 1e-3) / 1e12, 2),
                       "issued_tflops": round(issued_f / (ms * 1e-3) / 1e12, 2),
                       "frac_of_mfma_peak_issued": round(issued_f / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                       "frac_of_mfma_peak_algorithmic": round(alg / (ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                       "conv3x3_ms": round(conv_ms, 3), "conv3x3_issued_frac": round(conv_issued / (conv_ms * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
                       if conv_ms > 0 else 0.0,
                       "note": "median of 3 training-mode forward passes (encoder + head, batch statistics, activations kept for "
                               "backward) on precomputed features; issued = Winograd-reduced matrix FLOPs actually executed"}
        del feat
        torch.cuda.empty_cache()

    stages = {}
    if not args.no_stages and not args.graph:
        n_inst = 2
        trainer.step(audio, target)            # (untimed: the allocator refills its pool after the empty_cache() above)
        torch.cuda.synchronize()
        stage["on"] = True
        t1 = time.perf_counter()
        for _ in range(n_inst):
            trainer.step(audio, target)
        torch.cuda.synchronize()
        inst_ms = (time.perf_counter() - t1) / n_inst * 1e3
        stage["on"] = False
        ew_ms = 0.0
        for fam in list(timer.records):
            if fam.startswith("conv3x3"):
                continue
            n_l, ms, work, is_flops = timer.summary(fam)
            per_step = ms / n_inst
            ent = {"launches_per_step": n_l // n_inst, "ms_per_step": round(per_step, 3),
                   "share_of_step": round(per_step / step_ms, 4)}
            if is_flops > 0:
                ent["achieved_tflops"] = round(work / (ms * 1e-3) / 1e12, 2) if ms > 0 else 0.0
                ent["frac_of_mfma_peak"] = round(ent["achieved_tflops"] / PEAK_FP32_MFMA_TFLOPS, 4)
            else:
                gbs = work / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                ent.update({"algorithmic_GB_per_step": round(work / n_inst / 1e9, 3), "achieved_GBps": round(gbs, 1),
                            "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)})
                if not fam.startswith("gru"):
                    ew_ms += per_step
            stages[fam] = ent
        stages["_elementwise_total"] = {"ms_per_step": round(ew_ms, 3), "share_of_step": round(ew_ms / step_ms, 4),
                                        "instrumented_step_ms": round(inst_ms, 3)}

    if world > 1:
        # "reducer off": the same step without the gradient all-reduce (its difference to ms_per_step = exposed all-reduce).  LAST
        # measurement of the run: the replicas (parameters AND Adam moments) drift apart while nothing is averaged, and nothing
        # that assumes identical replicas follows (round 4 ran it before the pipeline / stage passes; ADVICE)
        trainer.reducer.active = False
        off_ms = timed_steps(lambda i: trainer.step(audio, target), args.steps, 1)
        trainer.reducer.active = True
        rccl["ms_per_step_reducer_off"] = round(off_ms, 3)
        rccl["exposed_allreduce_ms"] = round(step_ms - off_ms, 3)

    if rank == 0:
        n_f, ms_f, fl_f, ex_f = timer.summary("conv3x3_fwd_dgrad")
        n_w, ms_w, fl_w, ex_w = timer.summary("conv3x3_wgrad")
        # (the weight gradients' share of the step: over the region they were recorded in -- the timed steps with --events all, else the
        #  instrumented steps; no record at all with --no-stages)
        wg_region_ms = dt * 1e3 if args.events == "all" else (stages.get("_elementwise_total", {}).get("instrumented_step_ms", 0.0) * 2)
        issued = ex_f / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
        algorithmic = fl_f / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
        k1_ms = sum(s.elapsed_time(e) for s, e in k1_rec) / max(1, len(k1_rec))
        k1_bytes = FeatureExtractor.algorithmic_bytes(B, n_samples)
        value = world * B * args.seconds * args.steps / dt
        traffic, traffic_src = load_traffic(conv_algo)
        by_kernel = {}
        for tag in timer.tags("conv3x3_fwd_dgrad"):       # the family's kernels one by one (names as rocprofv3 lists them)
            n_t, ms_t, fl_t, ex_t = timer.summary("conv3x3_fwd_dgrad", tag)
            by_kernel[tag] = {"launches": n_t, "avg_launch_ms": round(ms_t / max(1, n_t), 4), "share_of_step": round(ms_t / (dt * 1e3), 4),
                              "issued_tflops": round(ex_t / (ms_t * 1e-3) / 1e12, 2), "frac": round(ex_t / (ms_t * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                              "algorithmic_tflops": round(fl_t / (ms_t * 1e-3) / 1e12, 2)}
        wg_kernels = {}
        for tag in timer.tags("conv3x3_wgrad"):
            n_t, ms_t, fl_t, ex_t = timer.summary("conv3x3_wgrad", tag)
            wg_kernels[tag] = {"launches": n_t, "avg_launch_ms": round(ms_t / max(1, n_t), 4), "share_of_step": round(ms_t / max(wg_region_ms, 1e-9), 4),
                               "issued_tflops": round(ex_t / (ms_t * 1e-3) / 1e12, 2), "frac": round(ex_t / (ms_t * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4),
                               "algorithmic_tflops": round(fl_t / (ms_t * 1e-3) / 1e12, 2)}
        dom = max(by_kernel, key=lambda k_: by_kernel[k_]["share_of_step"]) if by_kernel else None
        line = {
            "metric": "train-step audio-sec/s (4ch, %s+adyolo)" % args.encoder,
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.encoder + " + adyolo loss, synthetic 4ch 24kHz %ds clips, bs=%d per GPU, "
                                   "12 classes, features+fwd+loss+bwd+allreduce+Adam" % (args.seconds, B),
                       "global_batch": world * B, "clip_seconds": args.seconds,
                       "parallelism": "dp%d" % world + (" (functional run: all ranks on one device over gloo)" if one_device else "")},
            # the ONE dominant kernel of the step (largest share of step among the 3x3 forward / data-gradient kernels):
            # achieved = matrix FLOPs it ISSUES per second (what the MFMA pipe executes: Winograd F(4x4) issues 9/36 of the
            # direct-convolution FLOPs its launches stand for, algorithmic_tflops); the whole family is stages.conv3x3_fwd_dgrad
            "roofline": {"bound": "mfma", "kernel": dom,
                         "achieved": by_kernel[dom]["issued_tflops"] if dom else 0.0, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": by_kernel[dom]["frac"] if dom else 0.0, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_tflops": by_kernel[dom]["algorithmic_tflops"] if dom else 0.0,
                         "launches": by_kernel[dom]["launches"] if dom else 0,
                         "avg_launch_ms": by_kernel[dom]["avg_launch_ms"] if dom else 0.0,
                         "share_of_step": by_kernel[dom]["share_of_step"] if dom else 0.0},
            "stages": {
                "conv3x3_fwd_dgrad": {"kernels": by_kernel, "launches": n_f, "avg_launch_ms": round(ms_f / max(1, n_f), 4),
                                      "mfma_issued_tflops": round(issued, 2), "frac_of_mfma_peak": round(issued / PEAK_FP32_MFMA_TFLOPS, 4),
                                      "algorithmic_tflops": round(algorithmic, 2), "share_of_step": round(ms_f / (dt * 1e3), 4)},
                "conv3x3_wgrad": {"kernels": wg_kernels, "launches": n_w, "avg_launch_ms": round(ms_w / max(1, n_w), 4),
                                  "mfma_issued_tflops": round(ex_w / (ms_w * 1e-3) / 1e12, 2) if ms_w > 0 else 0.0,
                                  "frac_of_mfma_peak": round(ex_w / (ms_w * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if ms_w > 0 else 0.0,
                                  "algorithmic_tflops": round(fl_w / (ms_w * 1e-3) / 1e12, 2) if ms_w > 0 else 0.0,
                                  "share_of_step": round(ms_w / max(wg_region_ms, 1e-9), 4),
                                  "recorded_in": "timed steps" if args.events == "all" else "2 instrumented steps after the timed region"},
                "k1_features": {"ms": round(k1_ms, 4), "algorithmic_GB": round(k1_bytes / 1e9, 4),
                                "achieved_GBps": round(k1_bytes / (k1_ms * 1e-3) / 1e9, 1) if k1_ms > 0 else 0.0,
                                "frac_of_hbm_peak": round(k1_bytes / (k1_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if k1_ms > 0 else 0.0},
            },
            "final_loss": round(loss_val, 6),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
            # how the per-launch durations of `roofline` / `stages` were taken (HIP events on the launch stream; --events / --event-kind)
            "timing": {"events_in_timed_steps": args.events, "event_kind": EVENT_KIND_USED["kind"],
                       "event_fallback": EVENT_KIND_USED["fallback"]},
        }
        line["stages"].update(stages)
        if encoder_fwd is not None:
            line["stages"]["encoder_fwd"] = encoder_fwd
        if dispatch is not None:
            line["dispatch"] = dispatch
        if parity is not None:
            line["parity_check"] = parity
        if rccl is not None:
            line["rccl"] = rccl
        if pipeline is not None:
            line["pipeline"] = pipeline
        if world == 1 and not args.no_extra and args.encoder == "se-resnet34":
            del trainer, model                          # the headline model's 40 GB of cached activations go back first
            torch.cuda.empty_cache()
            line["extra_configs"] = {name: run_extra_config(name, torch) for name in EXTRA_CONFIGS}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
