"""Forward / data-gradient Winograd launches of the bench shapes (B = 64 x 60 s: stages 2-4), exact-fp32 kernel against the
bf16x3 kernel (ADYOLO_MATH=bf16x3, csrc/wino_b3.hip).  usage (GPU box): python3 tools/b3_bench.py [batch]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import adyolo_amd  # noqa: E402,F401
from adyolo_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SHAPES = [("stage1 32->32", 2400, 64, 32, 32), ("stage2.0 32->64", 1200, 32, 32, 64), ("stage2 64->64", 1200, 32, 64, 64), ("stage3 128->128", 600, 16, 128, 128), ("stage4 256->256", 600, 16, 256, 256),
          ("stage3.0 64->128", 600, 16, 64, 128), ("stage4.0 128->256", 600, 16, 128, 256)]


def timeit(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


for name, h, w, cin, cout in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(B, h, w, cin, device="cuda", generator=g)
    wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (9 * cin) ** 0.5
    sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda")
    res = {}
    for math in ("f32", "bf16x3"):
        wpk, _ = ops.pack_w3x3(wt, cin, want_dgrad=False, algo="winograd", math=math)
        t_plain = timeit(lambda: ops.conv3x3(x, wpk, cout))
        t_fused = timeit(lambda: ops.conv3x3(x, wpk, cout, relu=True, in_affine=(sc, sh), want_stats=True))
        y = ops.conv3x3(x, wpk, cout)
        res[math] = (t_plain, t_fused, y)
    d = float((res["f32"][2] - res["bf16x3"][2]).abs().max()) / float(res["f32"][2].abs().max())
    flops = 2.0 * B * h * w * cin * cout * 9
    print("%-18s plain %.3f -> %.3f ms (%.2fx, %.0f -> %.0f TFLOP/s direct-equivalent)   affine+relu+stats %.3f -> %.3f ms (%.2fx)   max rel diff %.1e"
          % (name, res["f32"][0], res["bf16x3"][0], res["f32"][0] / res["bf16x3"][0], flops / res["f32"][0] / 1e9,
             flops / res["bf16x3"][0] / 1e9, res["f32"][1], res["bf16x3"][1], res["f32"][1] / res["bf16x3"][1], d))
