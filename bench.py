#!/usr/bin/env python3
"""Headline benchmark: train-step audio-seconds per second of the AD-YOLO hot path on MI355X.

A "step" is one full optimisation step on raw audio already resident in HBM:
  K1 features (STFT -> log-mel + intensity vector) -> SE-ResNet34+BiGRU encoder + AD-YOLO head forward ->
  AD-YOLO loss -> backward -> [bucketed RCCL all-reduce when N > 1] -> fused Adam.
Workload (BASELINE.json configs[1]): synthetic 4-ch 24 kHz 60 s clips, batch 64 per GPU, se-resnet34 + adyolo,
12 classes, random-init weights (seed 100), fp32 arithmetic (exact-fp32 MFMA).  Weak scaling: every rank
processes its own 64 clips.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seconds S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` is measured live with HIP events (torch.cuda.Event on the stream the
kernels are launched on) around every launch of the dominant kernel family (conv3x3 forward / data-gradient)
inside the timed region; `cpu_baseline` times the CPU oracle (a port of the reference path, `oracle/`) on the
host cores on a bounded sample of the same workload (N = 1, rank 0 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, exact fp32
PEAK_HBM_GBS = 8000.0


def params(device, nb_classes=12):
    return {"args": {"device": device, "encoder": "se-resnet34", "loss": "adyolo"},
            "data_config": {"nb_classes": nb_classes},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0,
                                            "class_gain": 3.0},
                             "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


class KernelTimer:
    """HIP-event timing of selected op families on the launch stream, only while `active`."""

    def __init__(self):
        self.active = False
        self.records = {}          # family -> list of (start, end, work)

    def wrap(self, module, name, family, work_fn):
        orig = getattr(module, name)

        def timed(*a, **kw):
            if not self.active:
                return orig(*a, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = orig(*a, **kw)
            e.record()
            self.records.setdefault(family, []).append((s, e, work_fn(*a, **kw)))
            return out
        setattr(module, name, timed)

    def summary(self, family):
        recs = self.records.get(family, [])
        if not recs:
            return 0, 0.0, 0.0, 0.0
        ms = sum(s.elapsed_time(e) for s, e, _ in recs)
        return len(recs), ms, float(sum(w[0] for _, _, w in recs)), float(sum(w[1] for _, _, w in recs))


def load_traffic(wino):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_traffic.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE over this same command, corrected as MI355X_MICROARCH.md prescribes);
    counters cannot be collected inside the timed run, so this is null when no PMC summary matches the algorithm."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        return t.get("winograd" if wino else "direct", {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def cpu_baseline(seconds_budget=12.0, clip_seconds=20, batch=2):
    """Oracle (CPU port of the reference path) train step on a bounded sample: `batch` x 20 s chunks per step."""
    import numpy as np
    from oracle import features as ofeat, seresnet as onet, adyolo_loss as oloss
    from oracle.filler import fill_state_dict
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    torch.manual_seed(100)
    threads = torch.get_num_threads()
    sd = fill_state_dict(onet.state_dict_spec())
    plist = [v.requires_grad_(True) for k, v in sd.items()
             if v.is_floating_point() and not k.endswith(("running_mean", "running_var"))]
    opt = torch.optim.Adam(plist, lr=1e-3)
    n = 24000 * clip_seconds
    audio = synthetic_audio(batch, n, seed=4321)
    target = synthetic_targets(batch, n // 2400, 12, seed=4321)
    mel = ofeat.mel_filterbank()

    def step():
        feats = np.stack([ofeat.get_feature(audio[b].double().numpy(), None, mel)[0] for b in range(batch)])
        logits = onet.model_forward(sd, torch.from_numpy(feats), training=True, update_stats=True)
        opt.zero_grad()
        loss = oloss.adyolo_loss(logits, target, 12)
        loss.backward()
        opt.step()
        return float(loss)

    step()                                   # warm-up (allocator / oneDNN primitive caches)
    t0, n_steps = time.time(), 0
    while True:
        step()
        n_steps += 1
        if time.time() - t0 >= seconds_budget or n_steps >= 20:
            break
    dt = time.time() - t0
    return {"value": round(batch * clip_seconds * n_steps / dt, 3), "unit": "audio-s/s", "cores": threads,
            "kind": "port",
            "sample": "%d train steps of %d x %d s clips (features+fwd+loss+bwd+Adam), PyTorch-CPU/NumPy oracle, "
                      "%d threads of %d host cpus" % (n_steps, batch, clip_seconds, threads, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="clips per GPU")
    ap.add_argument("--seconds", type=int, default=60, help="clip length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--encoder", default="se-resnet34", choices=["se-resnet34", "resnet-conformer"],
                    help="se-resnet34 = the headline workload (BASELINE configs[1]); resnet-conformer = config 4")
    args = ap.parse_args()

    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops, dist as adist
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    import adyolo_amd.functional as Fn
    import torch.distributed as dist

    rank, world, local_rank = adist.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank

    B, n_samples = args.batch, 24000 * args.seconds
    T = n_samples // 600
    torch.manual_seed(100)
    prm = params(device)
    prm["args"]["encoder"] = args.encoder
    model = WrapperModel((1, 7, T, 64), (), prm).to(device)
    trainer = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, device), prm)
    audio = synthetic_audio(B, n_samples, seed=1234 + rank).to(device)
    target = synthetic_targets(B, T // 4, 12, seed=1234 + rank).to(device)

    timer = KernelTimer()
    # work = (algorithmic FLOPs of the 3x3 convolution, matrix FLOPs actually issued: 16/36 of that in Winograd form)
    wino = ops.conv_algo() == "winograd"

    def conv_work(x, wpk, cout, **kw):
        alg = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * cout * 9 * x.shape[3]
        return alg, alg * (16.0 / 36.0 if wpk.dim() == 4 else 1.0)

    def wgrad_work(x, dy, cin_real, **kw):
        alg = 2.0 * x.shape[0] * x.shape[1] * x.shape[2] * dy.shape[3] * 9 * x.shape[3]
        return alg, alg * (16.0 / 36.0 if (wino and x.shape[3] % 32 == 0 and dy.shape[3] % 32 == 0) else 1.0)
    timer.wrap(ops, "conv3x3", "conv3x3_fwd_dgrad", conv_work)
    timer.wrap(ops, "conv3x3_wgrad", "conv3x3_wgrad", wgrad_work)
    feat_call = trainer.features.__call__
    k1_rec = []

    def timed_features(a, channels_last8=True):
        if not timer.active:
            return feat_call(a, channels_last8)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = feat_call(a, channels_last8)
        e.record()
        k1_rec.append((s, e))
        return out
    trainer.features = timed_features

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = trainer.step(audio, target)
    sync()
    timer.active = True
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
    loss_val = float(loss)

    if rank == 0:
        n_f, ms_f, fl_f, ex_f = timer.summary("conv3x3_fwd_dgrad")
        n_w, ms_w, fl_w, ex_w = timer.summary("conv3x3_wgrad")
        executed = ex_f / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
        k1_ms = sum(s.elapsed_time(e) for s, e in k1_rec) / max(1, len(k1_rec))
        k1_bytes = FeatureExtractor.algorithmic_bytes(B, n_samples)
        achieved = fl_f / (ms_f * 1e-3) / 1e12 if ms_f > 0 else 0.0
        value = world * B * args.seconds * args.steps / dt
        line = {
            "metric": "train-step audio-sec/s (4ch, %s+adyolo)" % args.encoder,
            "value": round(value, 2), "unit": "audio-s/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.encoder + " + adyolo loss, synthetic 4ch 24kHz %ds clips, bs=%d per GPU, "
                                   "12 classes, features+fwd+loss+bwd+allreduce+Adam" % (args.seconds, B),
                       "global_batch": world * B, "clip_seconds": args.seconds, "parallelism": "dp%d" % world},
            # achieved = ALGORITHMIC 3x3-convolution FLOPs / time; the Winograd kernels issue 16/36 of
            # them, so frac can exceed 1 -- `mfma_issued` / `mfma_util` is the matrix-pipe utilisation proper
            "roofline": {"bound": "mfma",
                         "kernel": ("wino_fwd_kernel (Winograd F(2x2,3x3)" if wino else "conv3x3_fwd_kernel (direct") +
                                   "; forward + data-gradient launches)",
                         "achieved": round(achieved, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_FP32_MFMA_TFLOPS, 4), "traffic": load_traffic(wino),
                         "mfma_issued": round(executed, 2), "mfma_util": round(executed / PEAK_FP32_MFMA_TFLOPS, 4),
                         "launches": n_f, "avg_launch_ms": round(ms_f / max(1, n_f), 4),
                         "share_of_step": round(ms_f / (dt * 1e3), 4)},
            "stages": {
                "conv3x3_wgrad": {"launches": n_w, "avg_launch_ms": round(ms_w / max(1, n_w), 4),
                                  "achieved_tflops": round(fl_w / (ms_w * 1e-3) / 1e12, 2) if ms_w > 0 else 0.0,
                                  "mfma_issued_tflops": round(ex_w / (ms_w * 1e-3) / 1e12, 2) if ms_w > 0 else 0.0,
                                  "share_of_step": round(ms_w / (dt * 1e3), 4)},
                "k1_features": {"ms": round(k1_ms, 4), "algorithmic_GB": round(k1_bytes / 1e9, 4),
                                "achieved_GBps": round(k1_bytes / (k1_ms * 1e-3) / 1e9, 1) if k1_ms > 0 else 0.0,
                                "frac_of_hbm_peak": round(k1_bytes / (k1_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if k1_ms > 0 else 0.0},
            },
            "final_loss": round(loss_val, 6),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
