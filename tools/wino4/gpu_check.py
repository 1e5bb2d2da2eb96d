#!/usr/bin/env python3
"""One-shot GPU check of the F(4x4,3x3) convolution kernel (csrc/wino4.hip): correctness against a float64 convolution and
against the F(2x2) kernel's fused epilogues, then a same-process A/B timing against F(2x2) at the bench workload's shapes.
usage: python tools/wino4/gpu_check.py [--skip-check] [--skip-bench] [--batch 64] [--iters 5]"""
import argparse
import os
import sys

os.environ.setdefault("ADYOLO_W4_MIN_K", "32")      # the F(4x4) kernel at every eligible shape, not only where the library would choose it

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402


def dev(t):
    return t.to("cuda:0").contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def relerr(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    return float((got - ref).abs().max() / max(1e-30, ref.abs().max()))


def check_plain(n, h, w, cin, cout, seed=0):
    g = torch.Generator().manual_seed(seed + n * 1000 + h * 10 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    ref = F.conv2d(x.double(), wt.double(), None, padding=1)
    out = {}
    for algo in ("winograd4", "winograd"):
        wpk, wpd = ops.pack_w3x3(dev(wt), cin, want_dgrad=True, algo=algo)
        y = ops.conv3x3(dev(nhwc(x)), wpk, cout)
        torch.cuda.synchronize()
        out[algo] = relerr(nchw(y), ref)
        if algo == "winograd4":
            assert wpk.shape[0] == (36 if cout % 64 == 0 else 16), wpk.shape
        # data-gradient: dx = conv(dy, flipped transposed w)
        dy = torch.randn(n, cout, h, w, generator=torch.Generator().manual_seed(5))
        refd = F.conv_transpose2d(dy.double(), wt.double(), None, padding=1)
        dx = ops.conv3x3(dev(nhwc(dy)), wpd, cin)
        torch.cuda.synchronize()
        out[algo + "_dgrad"] = relerr(nchw(dx), refd)
    return out


def check_fused(n, h, w, cin, cout):
    """every fused operand of the epilogue / staging against the F(2x2) kernel (itself tested against torch)"""
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, h, w, cin, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    bias = torch.randn(cout, generator=g)
    add, mask = torch.randn(n, h, w, cout, generator=g), torch.randn(n, h, w, cout, generator=g)
    aux = torch.randn(n, h, w, cout, generator=g)
    mean, invstd = torch.randn(cout, generator=g) * 0.1, torch.rand(cout, generator=g) + 0.5
    res = {}
    outs = {}
    for algo in ("winograd4", "winograd"):
        wpk, _ = ops.pack_w3x3(dev(wt), cin, want_dgrad=False, algo=algo)
        y1, st1 = ops.conv3x3(dev(x), wpk, cout, bias=dev(bias), addend=dev(add), addend_mask=dev(mask), relu=True,
                              in_affine=(dev(scale), dev(shift)), want_stats=True)
        y2, st2 = ops.conv3x3(dev(x), wpk, cout, addend=dev(add), want_stats=True, stat_bn=(dev(aux), dev(mean), dev(invstd)),
                              stat_mask=dev(mask))
        torch.cuda.synchronize()
        outs[algo] = (y1, st1.sum(1), y2, st2.sum(1))
    names = ["affine+bias+masked addend+relu", "its patch sums", "addend + bn-backward stats (masked)", "its patch sums"]
    for nm, a, b in zip(names, outs["winograd4"], outs["winograd"]):
        res[nm] = relerr(a, b)
    # reference for the affine path from torch as well (padding stays zero after the affine)
    xa = x * scale + shift
    ref = F.relu(F.conv2d(xa.permute(0, 3, 1, 2).double(), wt.double(), bias.double(), padding=1)
                 + (add * (mask > 0)).permute(0, 3, 1, 2).double())
    res["affine path vs float64"] = relerr(nchw(outs["winograd4"][0]), ref)
    return res


SHAPES = {1: (2400, 64, 32, 32), 2: (1200, 32, 64, 64), 3: (600, 16, 128, 128), 4: (600, 16, 256, 256),
          12: (1200, 32, 32, 64), 23: (600, 16, 64, 128), 34: (600, 16, 128, 256)}


def bench(batch, iters, stages, fused):
    for st in stages:
        h, w, cin, cout = SHAPES[st]
        x = torch.randn(batch, h, w, cin, device="cuda:0")
        wt = torch.randn(cout, cin, 3, 3, device="cuda:0") * 0.05
        dy = torch.randn(batch, h, w, cout, device="cuda:0")
        aff = (torch.rand(cin, device="cuda:0") + 0.5, torch.randn(cin, device="cuda:0"))
        flops = 2.0 * batch * h * w * cout * 9 * cin
        line = "stage %2d B=%d %dx%d %d->%d %s:" % (st, batch, h, w, cin, cout, "fused" if fused else "plain")
        times = {}
        for rep in range(2):
            for algo in ("winograd", "winograd4"):
                wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo=algo)
                if algo == "winograd4" and wpk.shape[0] != 36:
                    continue
                if fused:
                    fn = lambda: ops.conv3x3(x, wpk, cout, relu=True, in_affine=aff, want_stats=True, addend=dy, addend_mask=dy)  # noqa: E731
                else:
                    fn = lambda: ops.conv3x3(x, wpk, cout)  # noqa: E731
                fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                times.setdefault(algo, []).append(s.elapsed_time(e) / iters)
        for algo, ts in times.items():
            ms = min(ts)
            issued = flops * (16 if algo == "winograd" else 9) / 36.0
            line += "  %s %.3f ms (issued %.1f TF = %.2f of peak)" % ("F2" if algo == "winograd" else "F4", ms, issued / ms / 1e9,
                                                                     issued / ms / 1e9 / 157.3)
        if "winograd4" in times:
            line += "  speed-up %.2fx" % (min(times["winograd"]) / min(times["winograd4"]))
        print(line, flush=True)


def bench_variants(batch, iters, stages):
    """The four operand combinations the SE-ResNet block actually launches (functional.py): forward conv1 / conv2, data-gradient
    of conv2 and of conv1 (identity shortcut, ReLU-mask bits)."""
    for st in stages:
        h, w, cin, cout = SHAPES[st]
        g = lambda *shape: torch.randn(*shape, device="cuda:0")                                         # noqa: E731
        x, dy, aux = g(batch, h, w, cin), g(batch, h, w, cout), g(batch, h, w, cout)
        wt = g(cout, cin, 3, 3) * 0.05
        aff = (torch.rand(cin, device="cuda:0") + 0.5, g(cin))
        mean, invstd = g(cout) * 0.1, torch.rand(cout, device="cuda:0") + 0.5
        bits = torch.randint(-2 ** 62, 2 ** 62, (batch * h * w * cout // 64,), dtype=torch.int64, device="cuda:0")
        combos = {
            "fwd conv1 (affine, relu, stats)": dict(relu=True, in_affine=aff, want_stats=True),
            "fwd conv2 (affine, stats)": dict(in_affine=aff, want_stats=True),
            "dgrad conv2 (stats vs bn aux)": dict(want_stats=True, stat_bn=(aux, mean, invstd)),
            "dgrad conv1 (addend+mask bits, stats vs aux, stat mask bits)": dict(addend=dy, addend_mask=bits, want_stats=True,
                                                                                 stat_bn=(aux, mean, invstd), stat_mask=bits),
            "plain": dict(),
        }
        for name, kw in combos.items():
            times = {}
            for rep in range(2):
                for algo in ("winograd", "winograd4"):
                    wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo=algo)
                    if algo == "winograd4" and wpk.shape[0] != 36:
                        continue
                    fn = lambda: ops.conv3x3(x, wpk, cout, **kw)                                       # noqa: E731
                    fn()
                    torch.cuda.synchronize()
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(iters):
                        fn()
                    e.record()
                    torch.cuda.synchronize()
                    times.setdefault(algo, []).append(s.elapsed_time(e) / iters)
            f2, f4 = min(times["winograd"]), min(times.get("winograd4", [float("nan")]))
            print("stage %2d %-62s F2 %.3f ms  F4 %.3f ms  speed-up %.2fx" % (st, name, f2, f4, f2 / f4), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", action="store_true", help="time the operand combinations of the real network instead")
    ap.add_argument("--skip-check", action="store_true")
    ap.add_argument("--skip-bench", action="store_true")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--stages", default="2,3,4,23,34")
    a = ap.parse_args()
    if not a.skip_check:
        worst = 0.0
        for shp in [(1, 16, 16, 32, 64), (2, 32, 16, 64, 64), (1, 40, 16, 128, 128), (2, 18, 16, 64, 128), (1, 10, 16, 256, 256),
                    (1, 24, 32, 64, 64), (2, 13, 32, 32, 64), (1, 33, 40, 32, 64), (1, 5, 5, 32, 64), (3, 7, 50, 64, 192),
                    (2, 35, 64, 32, 64), (5, 8, 16, 512, 64), (2, 67, 16, 128, 256), (3, 16, 16, 48 + 16, 128)]:
            r = check_plain(*shp)
            worst = max(worst, r["winograd4"], r["winograd4_dgrad"])
            print("plain %-28s F4 %.2e (dgrad %.2e)   F2 %.2e (dgrad %.2e)" % (shp, r["winograd4"], r["winograd4_dgrad"],
                                                                            r["winograd"], r["winograd_dgrad"]), flush=True)
        for shp in [(2, 20, 16, 64, 64), (3, 17, 16, 64, 128), (2, 9, 32, 32, 64), (2, 40, 64, 32, 64)]:
            r = check_fused(*shp)
            for k, v in r.items():
                worst = max(worst, v)
                print("fused %-24s %-40s %.2e" % (shp, k, v), flush=True)
        print("WORST relative error %.3e %s" % (worst, "OK" if worst < 2e-5 else "FAIL"), flush=True)
    if a.variants:
        bench_variants(a.batch, a.iters, [int(s) for s in a.stages.split(",")])
        return
    if not a.skip_bench:
        stages = [int(s) for s in a.stages.split(",")]
        bench(a.batch, a.iters, stages, fused=False)
        bench(a.batch, a.iters, stages, fused=True)


if __name__ == "__main__":
    main()
