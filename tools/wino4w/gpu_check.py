#!/usr/bin/env python3
"""GPU check of the F(4x4)-domain weight-gradient kernel (csrc/wino4w.hip) against a float64 weight gradient and the F(2x2)-domain
kernel (csrc/wino.hip), then a same-process A/B timing of the two at the bench workload's shapes.
usage: python tools/wino4w/gpu_check.py [--skip-check] [--skip-bench] [--batch 64] [--iters 5] [--stages 2,3,4,12,23,34]"""
import argparse
import os
import sys

os.environ["ADYOLO_W4W_MIN_WORK"] = "1"           # the F(4x4)-domain kernel at every shape it takes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

DEV = "cuda:0"


def ref64(x, dy, aff):
    """x [N][H][W][Ci], dy [N][H][W][Co] (channels last) -> dw [Co][Ci][3][3] in float64; aff: x' = scale x + shift inside the image"""
    xd = x.double()
    if aff is not None:
        xd = xd * aff[0].double() + aff[1].double()
    xn = xd.permute(0, 3, 1, 2)
    dn = dy.double().permute(0, 3, 1, 2)
    w = torch.zeros(dy.shape[3], x.shape[3], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    y = F.conv2d(xn, w, None, padding=1)
    (g,) = torch.autograd.grad(y, w, dn)
    return g


def check(n, h, w, cin, cout, affine):
    g = torch.Generator(device=DEV).manual_seed(n * 100 + h + cin)
    x = torch.randn(n, h, w, cin, device=DEV, generator=g) * 0.7 + 0.3
    dy = torch.randn(n, h, w, cout, device=DEV, generator=g) * 1e-2
    aff = (torch.rand(cin, device=DEV, generator=g) + 0.5, torch.randn(cin, device=DEV, generator=g)) if affine else None
    ref = ref64(x, dy, aff)
    am = float(ref.abs().max())
    out = {}
    for algo in ("winograd4", "winograd"):
        form = ops.wgrad_form(cin, cout, algo, (n, h, w))[0]
        dw = ops.conv3x3_wgrad(x, dy, cin, in_affine=aff, algo=algo)
        torch.cuda.synchronize()
        out[algo] = (form, float((dw.double() - ref).abs().max()) / am)
    ok = out["winograd4"][0] == "wino4_wgrad_kernel" and out["winograd4"][1] < 5e-5
    print("N=%d %dx%d %d->%d affine=%d:  %s %.2e   %s %.2e  %s" % (n, h, w, cin, cout, affine, out["winograd4"][0], out["winograd4"][1],
                                                                   out["winograd"][0], out["winograd"][1], "" if ok else "<-- FAIL"), flush=True)
    return ok


SHAPES = {1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128), 4: (600, 16, 256, 256), 12: (1200, 32, 32, 64), 23: (600, 16, 64, 128),
          34: (600, 16, 128, 256)}


def bench(batch, iters, stages, frames=2400):
    for st in stages:
        h, w, cin, cout = SHAPES[st]
        h = h * frames // 2400
        x = torch.randn(batch, h, w, cin, device=DEV)
        dy = torch.randn(batch, h, w, cout, device=DEV)
        aff = (torch.rand(cin, device=DEV) + 0.5, torch.randn(cin, device=DEV))
        flops = 2.0 * batch * h * w * cout * 9 * cin
        for affine in (False, True):
            times = {}
            for rep in range(2):
                for algo in ("winograd", "winograd4"):
                    fn = lambda: ops.conv3x3_wgrad(x, dy, cin, in_affine=aff if affine else None, algo=algo)   # noqa: E731
                    fn()
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(iters):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    times.setdefault(algo, []).append(s.elapsed_time(e) / iters)
            f2, f4 = min(times["winograd"]), min(times["winograd4"])
            print("stage %2d B=%d %dx%d %d->%d affine=%d:  F(2x2) domain %.3f ms (issued %.2f of peak)   F(4x4) domain %.3f ms (issued %.2f)   "
                  "%.2fx" % (st, batch, h, w, cin, cout, affine, f2, flops * 16 / 36 / f2 / 1e9 / 157.3, f4, flops * 9 / 36 / f4 / 1e9 / 157.3,
                             f2 / f4), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stages", default="1,2,3,4,12,23,34")
    ap.add_argument("--frames", type=int, default=2400, help="input frames of the timed shapes (2400 = 60 s clips, 800 = 20 s)")
    a = ap.parse_args()
    if not a.skip_check:
        ok = True
        # (N, H, W, Cin, Cout, affine): both widths, odd run counts (a half-empty pair), several segments per pair, 1 ... 32 channel blocks
        for shp in [(2, 8, 16, 32, 64, 0), (1, 8, 16, 32, 64, 1), (3, 40, 16, 64, 64, 1), (2, 64, 32, 32, 64, 0), (5, 36, 32, 64, 128, 1),
                    (1, 100, 48, 32, 64, 1), (2, 24, 64, 64, 64, 0), (9, 600, 16, 128, 128, 1), (4, 300, 32, 64, 64, 1),
                    (8, 152, 16, 256, 256, 0), (3, 28, 16, 96, 192, 1), (2, 64, 64, 32, 32, 1), (3, 20, 48, 64, 32, 0), (1, 12, 16, 32, 96, 1)]:
            ok = check(*shp) and ok
        print("CHECK %s" % ("OK" if ok else "FAIL"), flush=True)
    if not a.skip_bench:
        bench(a.batch, a.iters, [int(s) for s in a.stages.split(",")], a.frames)


if __name__ == "__main__":
    main()
