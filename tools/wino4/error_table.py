#!/usr/bin/env python3
"""Error of every 3x3-convolution kernel against a float64 convolution at the bench workload's layer shapes (VERDICT round 3,
item 1a): forward and data-gradient, F(4x4,3x3) / F(2x2,3x3) / direct, per stage.  The error does not depend on the batch, so
two clips per shape are used (the float64 reference runs on the host).  Gate: F(4x4) <= 2e-4 of the output's absmax.
usage: python tools/wino4/error_table.py [--out gpurun_out/r04_conv_error_table]   (writes .txt and .json)"""
import argparse
import json
import os
import sys

os.environ.setdefault("ADYOLO_W4_MIN_K", "32")
os.environ.setdefault("ADYOLO_W4_MIN_WGS", "1")     # the F(4x4) kernel whatever the size of the launch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402
import adyolo_amd  # noqa: F401,E402
from adyolo_amd import ops  # noqa: E402

SHAPES = [("stage 1", 2400, 64, 32, 32), ("stage 1->2", 1200, 32, 32, 64), ("stage 2", 1200, 32, 64, 64),
          ("stage 2->3", 600, 16, 64, 128), ("stage 3", 600, 16, 128, 128), ("stage 3->4", 600, 16, 128, 256),
          ("stage 4", 600, 16, 256, 256)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r04_conv_error_table")
    ap.add_argument("--batch", type=int, default=2)
    a = ap.parse_args()
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    rows, worst4 = [], 0.0
    for name, h, w, cin, cout in SHAPES:
        g = torch.Generator().manual_seed(h + cin)
        x = torch.randn(a.batch, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, 3, 3, generator=g) * float(np.sqrt(2.0 / (9 * cin)))       # He initialisation
        dy = torch.randn(a.batch, cout, h, w, generator=g)
        ref_f = F.conv2d(x.double(), wt.double(), None, padding=1)
        ref_d = F.conv_transpose2d(dy.double(), wt.double(), None, padding=1)
        xg = x.permute(0, 2, 3, 1).contiguous().to("cuda:0")
        dyg = dy.permute(0, 2, 3, 1).contiguous().to("cuda:0")
        for algo in ("winograd4", "winograd", "direct"):
            wpk, wpd = ops.pack_w3x3(wt.to("cuda:0"), cin, want_dgrad=True, algo=algo)
            for direction, src, pk, n_out, ref in (("forward", xg, wpk, cout, ref_f), ("data-gradient", dyg, wpd, cin, ref_d)):
                if isinstance(pk, ops.DualPack):
                    pk = pk.pick(a.batch, h, w, n_out)
                kernel = {36: "F(4x4,3x3)", 16: "F(2x2,3x3)"}.get(pk.shape[0] if pk.dim() == 4 else 0, "direct")
                if algo == "winograd4" and kernel != "F(4x4,3x3)":
                    continue                                  # not eligible (output channels not a multiple of 64): same as the F(2x2) row
                y = ops.conv3x3(src, pk, n_out)
                torch.cuda.synchronize()
                got = y.permute(0, 3, 1, 2).double().cpu()
                err = (got - ref).abs()
                rel_max = float(err.max() / ref.abs().max())
                rel_rms = float(err.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
                rows.append({"shape": name, "H": h, "W": w, "Cin": cin, "Cout": cout, "direction": direction, "kernel": kernel,
                             "max_err_of_absmax": rel_max, "rms_err_of_rms": rel_rms})
                if kernel == "F(4x4,3x3)":
                    worst4 = max(worst4, rel_max)
    lines = ["%-11s %5s x %-3s %4s -> %-4s %-14s %-11s %12s %12s" % ("shape", "H", "W", "Cin", "Cout", "direction", "kernel",
                                                                      "max/absmax", "rms/rms")]
    for r in rows:
        lines.append("%-11s %5d x %-3d %4d -> %-4d %-14s %-11s %12.3e %12.3e" % (r["shape"], r["H"], r["W"], r["Cin"], r["Cout"], r["direction"],
                                                                             r["kernel"], r["max_err_of_absmax"], r["rms_err_of_rms"]))
    lines.append("worst F(4x4,3x3) error %.3e of absmax -- gate 2e-4: %s" % (worst4, "OK" if worst4 <= 2e-4 else "FAIL"))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out + ".txt", "w") as f:
        f.write("\n".join(lines) + "\n")
    with open(a.out + ".json", "w") as f:
        json.dump({"rows": rows, "worst_f4": worst4, "gate": 2e-4}, f, indent=1)
    print("\n".join(lines))
    return 0 if worst4 <= 2e-4 else 1


if __name__ == "__main__":
    sys.exit(main())
