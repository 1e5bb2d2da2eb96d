"""OPT-IN math mode ADYOLO_MATH=bf16x3 (csrc/wino_b3.hip): the Winograd forward / data-gradient GEMMs on the bf16 MFMA with
every fp32 operand split exactly into three bf16 terms.  Checked against float64 convolutions with the SAME tolerance as the
exact-fp32 kernels (tests/test_gpu_kernels.py: 2e-5 of the output scale), and against the fp32 Winograd kernel's own error."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops as _ops
    return _ops


def dev(t):
    return t.to("cuda:0", torch.float32).contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def _err(got, ref64):
    return float((got.detach().double().cpu() - ref64).abs().max()) / max(1.0, float(ref64.abs().max()))


@pytest.mark.parametrize("n,h,w,cin,cout,relu,bias,addend", [
    (1, 24, 32, 64, 64, False, False, True),      # stage 2
    (2, 18, 16, 64, 128, False, False, False),
    (1, 16, 16, 128, 128, True, False, False),    # stage 3
    (1, 10, 16, 256, 256, False, False, True),    # stage 4
    (3, 7, 50, 64, 96, False, True, True),        # 3 channel blocks of 32 (NT = 1 kernel), odd sizes
    (5, 8, 16, 512, 64, True, False, False),      # deepest reduction
    (2, 13, 37, 96, 64, False, True, False),      # three chunks, ragged both ways
    (2, 16, 64, 32, 32, False, False, True),      # stage 1: one chunk, one channel tile
    (7, 24, 16, 32, 128, False, False, False),    # one chunk, 21 patches on two channel blocks
])
def test_b3_forward_matches_float64_like_the_fp32_kernel(ops, monkeypatch, n, h, w, cin, cout, relu, bias, addend):
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) if bias else None
    add = torch.randn(n, cout, h, w, generator=g) if addend else None
    ref = F.conv2d(x.double(), wt.double(), b.double() if bias else None, padding=1)
    if add is not None:
        ref = ref + add.double()
    if relu:
        ref = F.relu(ref)
    errs = {}
    if cin == 32:
        monkeypatch.setenv("ADYOLO_B3_MIN_K", "32")           # one-chunk contractions: not the default, still a supported shape
    for math in ("f32", "bf16x3"):
        wpk, _ = ops.pack_w3x3(dev(wt), cin, want_dgrad=False, algo="winograd", math=math)
        assert (wpk.shape[-1] == 768) == (math == "bf16x3")
        y = ops.conv3x3(dev(nhwc(x)), wpk, cout, bias=dev(b) if bias else None, addend=dev(nhwc(add)) if addend else None,
                        relu=relu)
        torch.cuda.synchronize()
        errs[math] = _err(nchw(y), ref)
    assert errs["bf16x3"] <= 2e-5, errs
    assert errs["bf16x3"] <= 3.0 * errs["f32"] + 1e-7, errs        # the same order as the fp32 MFMA's own rounding


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 18, 16, 64, 128), (1, 9, 16, 256, 256), (2, 21, 19, 96, 64), (2, 13, 32, 32, 64)])
def test_b3_data_gradient(ops, n, h, w, cin, cout):
    g = torch.Generator().manual_seed(7 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    dy = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(x, wt.double(), None, padding=1).backward(dy.double())
    uf, ud = ops.pack_w3x3(dev(wt), cin, want_dgrad=True, algo="winograd", math="bf16x3")
    assert ud.shape[-1] == 768                       # the data-gradient contracts over Cout >= 64
    assert (uf.shape[-1] == 768) == (cin > 32)       # one-chunk forward stays on the fp32 kernel by default
    dx = ops.conv3x3(dev(nhwc(dy)), ud, cin)
    torch.cuda.synchronize()
    assert _err(nchw(dx), x.grad) <= 2e-5


def test_b3_fused_affine_mask_stats(ops):
    """the training epilogue / staging operands on the bf16x3 kernel: producer affine, masked addend, statistics"""
    n, h, w, cin, cout = 3, 17, 16, 64, 128
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    add, mask = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
    xa = x.double() * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    ref = F.relu(F.conv2d(xa, wt.double(), None, padding=1) + add.double() * (mask > 0))
    wpk, _ = ops.pack_w3x3(dev(wt), cin, want_dgrad=False, algo="winograd", math="bf16x3")
    y, st = ops.conv3x3(dev(nhwc(x)), wpk, cout, addend=dev(nhwc(add)), addend_mask=dev(nhwc(mask)), relu=True,
                        in_affine=(dev(scale), dev(shift)), want_stats=True)
    ssum, mean, invstd = ops.bn_stats_tiles(st, n, h * w)
    torch.cuda.synchronize()
    assert _err(nchw(y), ref) <= 2e-5
    assert _err(ssum, ref.sum(dim=(2, 3))) <= 2e-5
    assert _err(mean, ref.mean(dim=(0, 2, 3))) <= 2e-5
    assert _err(invstd, 1.0 / torch.sqrt(ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5)) <= 5e-5


def test_b3_training_trajectory_matches_the_fp32_mode(ops, monkeypatch):
    """Whole model (K1 -> SE-ResNet34 + BiGRU -> head -> AD-YOLO loss -> backward -> Adam), six steps, ADYOLO_MATH=f32 against
    ADYOLO_MATH=bf16x3 from the same seed: the first loss agrees to 1e-6, the six-step loss trajectories to the north-star
    tolerance 1e-3.  First-step gradients: ANY two arithmetics differ by ~1e-3 in norm on this model (ReLUs whose
    pre-activation is within rounding of zero switch; measured 1.18e-3 between the fp32 Winograd and the fp32 direct
    convolutions, tools/b3_grad_probe.py), so the bf16x3 difference is bounded by that of the direct fp32 convolutions."""
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    b, n = 4, 24000 * 8
    audio = synthetic_audio(b, n, seed=77).to("cuda:0")
    target = synthetic_targets(b, n // 2400, 12, seed=77).to("cuda:0")
    runs = {}
    for name, algo, math, steps in (("f32", "winograd", "f32", 6), ("bf16x3", "winograd", "bf16x3", 6), ("direct", "direct", "f32", 1)):
        monkeypatch.setenv("ADYOLO_MATH", math)
        monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
        torch.manual_seed(100)
        prm = bench.params("cuda:0")
        model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
        tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, graph=False)
        losses = [float(tr.step(audio, target))]
        g0 = tr.flat.flat_grad.double().clone()
        for _ in range(steps - 1):
            losses.append(float(tr.step(audio, target)))
        nb3 = 0
        if algo == "winograd":
            nb3 = sum(1 for uf, ud in model.encoder._packs.packs for u in (uf, ud) if u.shape[-1] == 768)
        runs[name] = (losses, g0, nb3)
        del tr, model
        torch.cuda.empty_cache()
    assert runs["f32"][2] == 0 and runs["bf16x3"][2] >= 50, (runs["f32"][2], runs["bf16x3"][2])     # the mode really switched kernels
    lf, lb = runs["f32"][0], runs["bf16x3"][0]
    assert abs(lf[0] - lb[0]) <= 1e-6 * abs(lf[0]), (lf, lb)
    assert all(abs(a - c) <= 1e-3 * abs(a) for a, c in zip(lf, lb)), (lf, lb)
    assert lb[-1] < lb[0]
    gf = runs["f32"][1]
    rel = {k: float((gf - runs[k][1]).norm() / gf.norm()) for k in ("bf16x3", "direct")}
    print("first-step gradient, relative L2 difference to the fp32 Winograd run:", rel)
    assert rel["bf16x3"] <= 1.5 * rel["direct"] + 1e-5 and rel["bf16x3"] <= 5e-3, rel


@pytest.mark.parametrize("cin,cout,h,w", [(64, 64, 1200, 32), (128, 128, 600, 16), (256, 256, 600, 16), (64, 128, 600, 16)])
def test_b3_at_bench_shape_slices(ops, cin, cout, h, w):
    """Forward and data-gradient on the bf16x3 kernel launched at the FULL benchmark shape (64 clips x 60 s: stages 2-4), compared
    on two clips x 64 rows with a float64 convolution of the slice -- same windows and the same 2e-5 bound as
    test_conv3x3_at_bench_shape_slices uses for the fp32 kernels."""
    n = 64
    gen = torch.Generator(device="cuda:0").manual_seed(cin * 7 + cout)
    x = torch.randn(n, h, w, cin, generator=gen, device="cuda:0")
    dy = torch.randn(n, h, w, cout, generator=gen, device="cuda:0")
    wt = (torch.randn(cout, cin, 3, 3, generator=gen, device="cuda:0") / np.sqrt(9 * cin)).contiguous()
    wf, wd = ops.pack_w3x3(wt, cin, algo="winograd", math="bf16x3")
    assert wf.shape[-1] == 768 and wd.shape[-1] == 768
    y = ops.conv3x3(x, wf, cout)
    dx = ops.conv3x3(dy, wd, cin)
    torch.cuda.synchronize()
    wc = wt.cpu()
    wflip = wc.flip(2, 3).permute(1, 0, 2, 3).contiguous()
    for clip, r0 in ((0, 0), (n - 1, h // 2 - 29)):
        lo, hi = max(r0 - 1, 0), min(r0 + 65, h)
        for src, kern, out, what in ((x, wc, y, "forward"), (dy, wflip, dx, "data-gradient")):
            xin = src[clip, lo:hi].cpu().permute(2, 0, 1)[None]
            ref = F.conv2d(xin.double(), kern.double(), padding=1)[0].permute(1, 2, 0)
            top = r0 - lo
            e = _err(out[clip, r0:r0 + 64], ref[top:top + 64])
            assert e <= 2e-5, "%s clip %d rows %d..%d: %.2e of absmax" % (what, clip, r0, r0 + 64, e)
