"""Two ranks (gloo, one GPU) vs the sequential emulation on the real model: per-parameter difference of the reduced
gradients of step 0.  (GPU box)"""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path.insert(0, os.environ["ADYOLO_REPO"])
import torch, torch.distributed as dist
import adyolo_amd, bench
from adyolo_amd import dist as adist, functional as Fn
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep
mode = os.environ["MODE"]; b, n = 2, 24000 * 4
def make():
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    model.encoder.lstm.dropout = 0.0
    return TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
def data(r):
    return synthetic_audio(b, n, seed=30 + r).to("cuda:0"), synthetic_targets(b, n // 2400, 12, seed=40 + r).to("cuda:0")
if mode == "rank":
    rank, world, _ = adist.init_from_env("gloo")
    tr = make()
    audio, target = data(rank)
    opt_step = tr.optimizer.step
    saved = []
    def spy(grad_scale=1.0):
        saved.append(tr.flat.flat_grad.clone())
        opt_step(grad_scale=grad_scale)
    tr.optimizer.step = spy
    if os.environ.get("SYNC_HOOKS") == "1":
        launch = tr.reducer._launch
        def synced(bk):
            torch.cuda.synchronize()
            launch(bk)
        tr.reducer._launch = synced
    for _ in range(2):
        tr.step(audio, target)
    torch.cuda.synchronize()
    torch.save({"g": [t.cpu() for t in saved]}, "/tmp/dp2_rank%d.pt" % rank)
    dist.barrier(); dist.destroy_process_group()
else:
    trs = [make(), make()]; dat = [data(0), data(1)]; saved = []; locs = []
    for _ in range(2):
        for r, tr in enumerate(trs):
            tr.model.train()
            out = tr.model(tr.features(dat[r][0], channels_last8=True), channels_last8=True)
            tr.optimizer.zero_grad()
            loss = tr.criterion(out, dat[r][1])
            Fn.SINK.begin(tr.flat, tr.reducer)
            try: loss.backward()
            finally: Fn.SINK.end()
        total = trs[0].flat.flat_grad + trs[1].flat.flat_grad
        saved.append(total.clone()); locs.append([t.flat.flat_grad.clone().cpu() for t in trs])
        for tr in trs:
            tr.flat.flat_grad.copy_(total); tr.optimizer.step(grad_scale=0.5)
    names = {id(p): k for k, p in trs[0].model.named_parameters()}
    layout = [(names[id(p)], off, cnt) for p, (off, cnt) in zip(trs[0].flat.params, trs[0].flat.offsets)]
    torch.save({"g": [t.cpu() for t in saved], "loc": locs, "layout": layout, "buckets": [(s, e) for s, e, _ in trs[0].reducer.buckets]}, "/tmp/dp2_emu.pt")
'''


def main():
    sync = sys.argv[1] if len(sys.argv) > 1 else "0"
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, ADYOLO_REPO=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SYNC_HOOKS=sync)
    ps = [subprocess.Popen([sys.executable, "-c", CHILD], env=dict(base, MODE="rank", WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0")) for r in range(2)]
    for p in ps:
        assert p.wait(timeout=600) == 0
    subprocess.run([sys.executable, "-c", CHILD], env=dict(base, MODE="emu", WORLD_SIZE="1", RANK="0"), check=True)
    import torch
    r0, r1, emu = (torch.load("/tmp/dp2_%s.pt" % k) for k in ("rank0", "rank1", "emu"))
    print("sync_hooks =", sync, "buckets", emu["buckets"])
    for step in range(2):
        g0, g1, ge = r0["g"][step], r1["g"][step], emu["g"][step]
        print("step %d: rank0==rank1 %s, rank0==emu %s" % (step, bool(torch.equal(g0, g1)), bool(torch.equal(g0, ge))))
        if step == 0:
            shown = 0
            for name, off, cnt in emu["layout"]:
                a, e = g0[off:off + cnt], ge[off:off + cnt]
                if not torch.equal(a, e):
                    la, lb = emu["loc"][0][0][off:off + cnt], emu["loc"][0][1][off:off + cnt]
                    d = float((a - e).abs().max())
                    print("  %-44s off %8d n %7d  |dp-emu| %.3e  absmax %.3e  dp==locA %s dp==locB %s dp==0 %s" %
                          (name, off, cnt, d, float(e.abs().max()), bool(torch.equal(a, la)), bool(torch.equal(a, lb)), bool((a == 0).all())))
                    shown += 1
                    if shown > 40:
                        break


if __name__ == "__main__":
    main()
