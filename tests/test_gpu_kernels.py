"""GPU parity tests (run with ``-m gpu`` on an MI355X): every hand-written kernel, through the C ABI
(ctypes -> libadyolo_hip.so), against the CPU oracle / a plain PyTorch-CPU fp32 reference of the same op on the
same seeded inputs.  Tolerances: 1e-3 (BASELINE.json north_star) or tighter, written at each assert."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import adyolo_amd  # noqa: F401
    from adyolo_amd import ops as _ops
    return _ops


def dev(t):
    return t.to("cuda:0").contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def assert_close(got, ref, tol, what):
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, tuple(got.shape), tuple(ref.shape))
    scale = max(1.0, float(ref.abs().max()))
    err = float((got - ref).abs().max())
    assert np.isfinite(err) and err <= tol * scale, "%s: max abs err %.3e (ref absmax %.3e, tol %.1e)" % (what, err, scale, tol)


# ------------------------------------------------------------------------------------------------ conv
@pytest.mark.parametrize("n,h,w,cin,cout,relu,bias,addend", [
    (2, 20, 64, 7, 32, True, True, False),       # stem shape family (Cin 7 padded to 8), W=64 -> TW=32
    (1, 9, 16, 7, 32, False, True, False),       # ragged H, TW=16 path
    (3, 13, 45, 7, 32, True, True, False),       # stem kernel: ragged rows and columns (2 column patches, the 2nd 13 wide)
    (2, 11, 32, 7, 32, False, False, False),     # stem kernel without bias / ReLU
    (2, 16, 64, 32, 32, False, False, True),     # layer1
    (2, 13, 32, 32, 64, True, False, False),     # layer2.0 conv1, ragged H
    (1, 24, 32, 64, 64, False, False, True),
    (2, 18, 16, 64, 128, False, False, False),   # TW=16
    (1, 16, 16, 128, 128, True, False, False),
    (1, 8, 16, 128, 256, False, False, False),
    (1, 10, 16, 256, 256, False, False, True),
    (1, 5, 5, 32, 32, False, False, False),      # tile much larger than the image
    (1, 33, 40, 32, 64, True, False, False),     # ragged both ways, TW=32
    (3, 7, 50, 64, 96, False, True, True),       # 3 channel blocks (no XCD mapping), odd sizes, 3 x 4 x 1 patches
    (5, 8, 16, 512, 64, True, False, False),     # deepest reduction the Winograd affine table allows (Cin = 512)
    (7, 24, 16, 32, 128, False, False, False),   # one-chunk variant, 21 patches on 4 channel blocks (ragged block deal)
])
@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4"])
def test_conv3x3_forward(ops, monkeypatch, n, h, w, cin, cout, relu, bias, addend, algo):
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")      # winograd4: the F(4x4) kernel wherever the shape allows it (else F(2x2))
    g = torch.Generator().manual_seed(n * 1000 + h * 10 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    b = torch.randn(cout, generator=g) if bias else None
    add = torch.randn(n, cout, h, w, generator=g) if addend else None
    ref = F.conv2d(x, wt, b, padding=1)
    if add is not None:
        ref = ref + add
    if relu:
        ref = F.relu(ref)
    cin_p = 8 if cin == 7 else cin
    xg = dev(nhwc(x))
    if cin_p != cin:
        xg = dev(F.pad(nhwc(x), (0, cin_p - cin)))
    wpk, _ = ops.pack_w3x3(dev(wt), cin_p, want_dgrad=False, algo=algo)
    assert (wpk.dim() == 4) == (algo != "direct" and cin != 7)
    assert (wpk.shape[0] == 36) == (algo == "winograd4" and cin != 7 and cout % 64 == 0)
    y = ops.conv3x3(xg, wpk, cout, bias=dev(b) if bias else None, addend=dev(nhwc(add)) if addend else None, relu=relu)
    torch.cuda.synchronize()
    assert_close(nchw(y), ref, 2e-5, "conv3x3 fwd (%s)" % algo)


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4"])
def test_conv3x3_without_relu_propagates_nan(ops, monkeypatch, algo, persist):
    """A NaN in the input must come out as NaN, not as -inf: round 4's branch-free ReLU (a maximum against a -inf floor for launches
    without ReLU) turned NaN results into -inf in the two Winograd epilogues; now a maximum + wave-uniform select
    (one-patch kernels) / a per-round wave-uniform branch (persistent F(4x4) kernel).  The other sample stays finite."""
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")
    monkeypatch.setenv("ADYOLO_W4_PERSIST", persist)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 16, 16, 64, generator=g)
    x[1, 5, 7, 3] = float("nan")
    wt = torch.randn(64, 64, 3, 3, generator=g) / 24.0
    wpk, _ = ops.pack_w3x3(dev(wt), 64, want_dgrad=False, algo=algo)
    for relu in (False, True):
        y = ops.conv3x3(dev(x), wpk, 64, relu=relu).cpu()
        assert torch.isfinite(y[0]).all()
        assert not torch.isinf(y).any(), "NaN became an infinity (relu=%s)" % relu
        if not relu:
            assert torch.isnan(y[1, 4:7, 6:9]).all()         # the 3 x 3 neighbourhood every filter tap reaches


@pytest.mark.parametrize("n,h,w,cin,cout", [
    (2, 16, 64, 32, 32), (2, 13, 32, 32, 64), (1, 18, 16, 64, 128), (1, 9, 16, 256, 256), (2, 20, 64, 7, 32),
    (3, 40, 64, 32, 32), (2, 150, 24, 32, 64), (1, 67, 37, 64, 32),
    (6, 132, 16, 256, 256),      # Winograd wgrad: 18 work items on 16 slabs -> several items per workgroup, ragged segment
    (2, 21, 19, 96, 32),         # odd H and W (general masking path), 3 input-channel blocks
    (3, 11, 37, 7, 32), (1, 5, 64, 7, 32), (5, 3, 6, 7, 32),    # stem weight-gradient kernel: odd W, fewer rows than waves
])
@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4"])
def test_conv3x3_dgrad_wgrad(ops, monkeypatch, n, h, w, cin, cout, algo):
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")
    g = torch.Generator().manual_seed(7 + cin + cout)
    x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)).requires_grad_(True)
    dy = torch.randn(n, cout, h, w, generator=g)
    F.conv2d(x, wt, None, padding=1).backward(dy)
    cin_p = 8 if cin == 7 else cin
    xg = dev(F.pad(nhwc(x.detach()), (0, cin_p - cin)))
    dyg = dev(nhwc(dy))
    dw = ops.conv3x3_wgrad(xg, dyg, cin, algo=algo)
    torch.cuda.synchronize()
    assert_close(dw, wt.grad, 2e-5, "conv3x3 wgrad (%s)" % algo)
    if cin != 7:
        _, wpk_d = ops.pack_w3x3(dev(wt.detach()), cin_p, want_dgrad=True, algo=algo)
        dx = ops.conv3x3(dyg, wpk_d, cin)
        torch.cuda.synchronize()
        assert_close(nchw(dx), x.grad, 2e-5, "conv3x3 dgrad (%s)" % algo)


@pytest.mark.parametrize("n,h,w,cin,cout,affine", [
    (2, 8, 16, 32, 64, 0), (1, 8, 16, 32, 64, 1),           # one pair / half a pair, a single step
    (3, 40, 16, 64, 64, 1), (5, 36, 32, 64, 128, 1),         # odd run counts: the last pair is half empty
    (2, 64, 32, 32, 64, 0), (1, 100, 48, 32, 64, 1), (2, 24, 64, 64, 64, 0),      # widths of 2, 3, 4 runs
    (4, 300, 32, 64, 64, 1), (6, 132, 16, 256, 256, 0),      # several segments per pair, several items per workgroup
    (3, 28, 16, 96, 192, 1),                                 # 3 x 3 channel blocks
    (2, 64, 64, 32, 32, 1), (3, 20, 48, 64, 32, 0), (1, 12, 16, 32, 96, 1),       # 32-channel output blocks (NB = 1)
    (5, 36, 4, 64, 64, 1), (7, 64, 4, 128, 128, 0), (3, 40, 8, 64, 128, 0), (2, 24, 8, 64, 64, 1), (9, 16, 4, 32, 32, 1),   # narrow maps: runs of 4 / 2 samples
])
def test_wino4_wgrad_every_geometry(ops, monkeypatch, n, h, w, cin, cout, affine):
    """The F(4x4)-domain weight-gradient kernel (csrc/wino4w.hip) forced onto shapes far below its dispatch threshold: every
    branch of its work split (pairs of runs, segments, half-empty pairs), both block widths and the fused BatchNorm affine,
    against a float64 weight gradient.  Measured: <= 1.6e-6 of absmax (tools/wino4w/gpu_check.py)."""
    monkeypatch.setenv("ADYOLO_W4W_MIN_WORK", "1")
    assert ops.wgrad_form(cin, cout, "winograd4", (n, h, w))[0] == "wino4_wgrad_kernel"
    g = torch.Generator().manual_seed(n * 100 + h + cin)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64) * 0.7 + 0.3
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64) * 1e-2
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    xa = x * scale.double()[None, :, None, None] + shift.double()[None, :, None, None] if affine else x
    wt = torch.zeros(cout, cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xa, wt, None, padding=1).backward(dy)
    dw = ops.conv3x3_wgrad(dev(nhwc(x.float())), dev(nhwc(dy.float())), cin,
                           in_affine=(dev(scale), dev(shift)) if affine else None, algo="winograd4")
    torch.cuda.synchronize()
    err = float((dw.double().cpu() - wt.grad).abs().max()) / float(wt.grad.abs().max())
    assert err < 2e-5, "F(4x4)-domain wgrad: %.2e of absmax" % err


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 20, 64, 32, 32), (3, 17, 16, 64, 128), (2, 9, 32, 32, 64), (2, 37, 16, 128, 64)])
@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4"])
def test_conv3x3_fused_affine_mask_stats(ops, monkeypatch, n, h, w, cin, cout, algo):
    """in_affine (producer BN affine applied while staging; padding stays zero), masked addend, epilogue statistics."""
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")
    g = torch.Generator().manual_seed(h * 7 + cin)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / np.sqrt(9 * cin)
    scale, shift = torch.rand(cin, generator=g) + 0.5, torch.randn(cin, generator=g)
    add, mask = torch.randn(n, cout, h, w, generator=g), torch.randn(n, cout, h, w, generator=g)
    xa = x * scale[None, :, None, None] + shift[None, :, None, None]
    ref = F.relu(F.conv2d(xa, wt, None, padding=1) + add * (mask > 0))
    wpk, _ = ops.pack_w3x3(dev(wt), cin, want_dgrad=False, algo=algo)
    y, st = ops.conv3x3(dev(nhwc(x)), wpk, cout, addend=dev(nhwc(add)), addend_mask=dev(nhwc(mask)), relu=True,
                        in_affine=(dev(scale), dev(shift)), want_stats=True)
    ssum, mean, invstd = ops.bn_stats_tiles(st, n, h * w)
    torch.cuda.synchronize()
    assert_close(nchw(y), ref, 2e-5, "fused conv")
    assert_close(ssum, ref.sum(dim=(2, 3)), 2e-5, "per-sample sums from the conv epilogue")
    assert_close(mean, ref.mean(dim=(0, 2, 3)), 2e-5, "batch mean from the conv epilogue")
    assert_close(invstd, 1.0 / torch.sqrt(ref.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 5e-5, "batch invstd")
    # weight gradient with x seen through the same affine
    dy = torch.randn(n, cout, h, w, generator=g)
    xr = xa.clone().requires_grad_(False)
    wr = wt.clone().requires_grad_(True)
    F.conv2d(xr, wr, None, padding=1).backward(dy)
    dw = ops.conv3x3_wgrad(dev(nhwc(x)), dev(nhwc(dy)), cin, in_affine=(dev(scale), dev(shift)), algo=algo)
    torch.cuda.synchronize()
    assert_close(dw, wr.grad, 2e-5, "wgrad through the fused affine")


@pytest.mark.parametrize("n,h,cin,cout", [(2, 40, 64, 128), (1, 8, 128, 64), (3, 100, 64, 64), (5, 4, 192, 64)])
def test_conv3x1_winograd_1d(ops, n, h, cin, cout):
    """1-D Winograd F(4, 3) along the time axis (csrc/wino1d.hip, the ResNet-Conformer's one-bin-wide stages): forward, data
    gradient and weight gradient of the 3 x 1 convolution against float64; zero padding at both ends of every sample."""
    from adyolo_amd import functional as Fn
    g = torch.Generator().manual_seed(11 * n + h + cin)
    x = torch.randn(n, cin, h, 1, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(cout, cin, 3, 1, generator=g, dtype=torch.float64) / np.sqrt(3 * cin)).requires_grad_(True)
    dy = torch.randn(n, cout, h, 1, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, w, None, padding=(1, 0))
    y_ref.backward(dy)
    xg = dev(nhwc(x.detach().float())).requires_grad_(True)
    wg = dev(w.detach().float()).requires_grad_(True)
    y = Fn.Conv3x1WinoFn.apply(xg, wg)
    y.backward(dev(nhwc(dy.float())))
    torch.cuda.synchronize()
    for got, ref, what in ((nchw(y), y_ref, "y"), (nchw(xg.grad), x.grad, "dx"), (wg.grad, w.grad, "dw")):
        err = float((got.detach().double().cpu() - ref.detach()).abs().max()) / float(ref.detach().abs().max())
        assert err < 2e-5, "1-D Winograd %s: %.2e of absmax" % (what, err)


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 70, 4, 64, 64), (3, 133, 8, 64, 128), (1, 128, 4, 128, 64), (2, 9, 8, 64, 64), (2, 260, 3, 64, 64),
                                            (2, 100, 7, 128, 128)])
def test_conv3x3_narrow_maps_on_the_persistent_kernel(ops, monkeypatch, n, h, w, cin, cout):
    """Maps at most 8 bins wide (the ResNet-Conformer's middle stages): plain launches of adyolo_wino4_fwd take patches one / two
    tiles wide on the persistent kernel.  Forward, data gradient (F(4x4)) and weight gradient (implicit GEMM) of
    functional.Conv3x3NarrowFn against float64, ragged heights and widths included."""
    from adyolo_amd import functional as Fn
    g = torch.Generator().manual_seed(h * 3 + w + cin)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, dtype=torch.float64) / np.sqrt(9 * cin)).requires_grad_(True)
    dy = torch.randn(n, cout, h, w, generator=g, dtype=torch.float64)
    y_ref = F.conv2d(x, wt, None, padding=1)
    y_ref.backward(dy)
    assert ops.w4_narrow_ok(cin, cout)
    xg = dev(nhwc(x.detach().float())).requires_grad_(True)
    wg = dev(wt.detach().float()).requires_grad_(True)
    y = Fn.Conv3x3NarrowFn.apply(xg, wg)
    assert ops._lib.load().adyolo_wino4_last_form() == 2, "not the persistent kernel"
    y.backward(dev(nhwc(dy.float())))
    torch.cuda.synchronize()
    for got, ref, what in ((nchw(y), y_ref, "y"), (nchw(xg.grad), x.grad, "dx"), (wg.grad, wt.grad, "dw")):
        err = float((got.detach().double().cpu() - ref.detach()).abs().max()) / float(ref.detach().abs().max())
        assert err < 2e-5, "narrow-map 3x3 convolution %s: %.2e of absmax" % (what, err)


# ------------------------------------------------------------------------------------------------ gemm
@pytest.mark.parametrize("m,n,k,ta,tb,bias,splits", [
    (300, 2400, 256, False, False, True, 1),      # head
    (130, 70, 36, False, False, False, 1),        # ragged everywhere
    (512, 256, 2400, False, True, False, 1),      # dX = dY W
    (384, 256, 3000, True, True, False, 4),       # dW = dY^T X, split-K
    (64, 8, 64, False, False, True, 1),           # SE-sized
    (260, 132, 520, True, False, False, 3),
])
def test_gemm(ops, m, n, k, ta, tb, bias, splits):
    g = torch.Generator().manual_seed(m + n + k)
    a = torch.randn(k, m, generator=g) if ta else torch.randn(m, k, generator=g)
    b = torch.randn(k, n, generator=g) if tb else torch.randn(n, k, generator=g)
    bv = torch.randn(n, generator=g) if bias else None
    ref = (a.t() if ta else a) @ (b if tb else b.t())
    if bias:
        ref = ref + bv
    out = ops.gemm(dev(a), dev(b), m, n, k, a.shape[1], b.shape[1], trans_a=ta, trans_b=tb,
                   bias=dev(bv) if bias else None, splits=splits)
    torch.cuda.synchronize()
    assert_close(out, ref, 3e-5, "gemm")
    acc = ops.gemm(dev(a), dev(b), m, n, k, a.shape[1], b.shape[1], trans_a=ta, trans_b=tb, out=out.clone(),
                   accumulate=True, splits=splits)
    torch.cuda.synchronize()
    assert_close(acc, 2 * ref - (bv if bias else 0), 3e-5, "gemm accumulate (bias only in the first call)")


def test_colsum_and_elementwise(ops):
    g = torch.Generator().manual_seed(3)
    a = torch.randn(1000, 768, generator=g)
    s = ops.colsum(dev(a)[:, 384:])
    torch.cuda.synchronize()
    assert_close(s, a[:, 384:].sum(0), 1e-5, "colsum strided")
    b = torch.randn(1000, 768, generator=g)
    assert_close(ops.add(dev(a), dev(b)), a + b, 1e-6, "add")
    assert_close(ops.mul(dev(a), dev(b)), a * b, 1e-6, "mul")
    assert_close(ops.scale_dev(dev(a), dev(torch.tensor([0.25]))), a * 0.25, 1e-6, "scale_dev")


# ------------------------------------------------------------------------------------------------ norm
@pytest.mark.parametrize("n,h,w,c", [(3, 20, 16, 32), (2, 7, 8, 64), (1, 50, 16, 256), (4, 5, 4, 128)])
def test_batchnorm_train_fwd_bwd(ops, n, h, w, c):
    g = torch.Generator().manual_seed(c + n)
    x = (torch.randn(n, c, h, w, generator=g) * 2 + 0.5).relu().requires_grad_(True)    # BN input is a ReLU output
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g).requires_grad_(True)
    rm, rv = torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5)
    dy = torch.randn(n, c, h, w, generator=g)
    y_ref.backward(dy)
    xg = dev(nhwc(x.detach()))
    rmg, rvg = dev(rm), dev(rv)
    ssum, mean, invstd = ops.bn_stats(xg, rmg, rvg, 0.1, 1e-5)
    scale, shift = ops.bn_scale_shift(dev(gamma.detach()), dev(beta.detach()), mean, invstd)
    y = ops.affine(xg, scale, shift)
    dx, dgamma, dbeta = ops.bn_bwd(dev(nhwc(dy)), xg, dev(gamma.detach()), mean, invstd, relu_mask=False)
    torch.cuda.synchronize()
    assert_close(nchw(y), y_ref, 2e-5, "bn fwd")
    assert_close(rmg, rm_ref, 1e-5, "running_mean")
    assert_close(rvg, rv_ref, 1e-5, "running_var")
    assert_close(ssum, x.detach().sum(dim=(2, 3)), 1e-5, "per-sample sums")
    assert_close(nchw(dx), x.grad, 5e-5, "bn dx")
    assert_close(dgamma, gamma.grad, 5e-5, "bn dgamma")
    assert_close(dbeta, beta.grad, 5e-5, "bn dbeta")
    dxm, _, _ = ops.bn_bwd(dev(nhwc(dy)), xg, dev(gamma.detach()), mean, invstd, relu_mask=True)
    assert_close(nchw(dxm), x.grad * (x.detach() > 0), 5e-5, "bn dx with relu mask")


@pytest.mark.parametrize("n,h,w,c,r_affine,want_mask", [
    (2, 8, 64, 32, False, True), (3, 6, 32, 64, True, True), (1, 4, 16, 128, False, False), (2, 64, 16, 32, True, True),
])
def test_se_tail_fwd_pooled_equals_tail_then_avgpool(ops, n, h, w, c, r_affine, want_mask):
    """The tail in front of a pooled stage boundary (reference resnet.py:29,40: the next block's AvgPool2d(2, 2)) writes
    avgpool2(e) and the ReLU-mask bits of e in one pass: both bit-identical to se_tail_fwd followed by avgpool2."""
    import torch
    torch.manual_seed(5)
    cc, r = dev(torch.randn(n, h, w, c)), dev(torch.randn(n, h, w, c))
    scale, shift = dev(torch.rand(c) + 0.5), dev(torch.randn(c))
    s = dev(torch.rand(n, c))
    raff = (dev(torch.rand(c) + 0.5), dev(torch.randn(c))) if r_affine else None
    assert ops.se_tail_pool_ok(h, w, c)
    if want_mask:
        e, bits = ops.se_tail_fwd(cc, r, scale, shift, s, want_mask=True, r_affine=raff)
    else:
        e, bits = ops.se_tail_fwd(cc, r, scale, shift, s, r_affine=raff), None
    ref = ops.avgpool2(e)
    got, gbits = ops.se_tail_fwd(cc, r, scale, shift, s, want_mask=want_mask, r_affine=raff, pool_hw=(h, w))
    assert got.shape == ref.shape and torch.equal(got, ref)
    if want_mask:
        assert bits is not None and torch.equal(gbits, bits)
    else:
        assert gbits is None
    # shapes the fused form does not take are reported, not mangled
    assert not ops.se_tail_pool_ok(7, 64, 32) and not ops.se_tail_pool_ok(8, 6, 32) and not ops.se_tail_pool_ok(8, 64, 256)


@pytest.mark.parametrize("n,h,w,c,want_dr", [(2, 8, 64, 32, False), (3, 6, 32, 64, True), (2, 64, 16, 32, False)])
def test_se_tail_bwd_from_the_pooled_gradient_equals_avgpool_bwd_then_tail_bwd(ops, n, h, w, c, want_dr):
    """Backward of a block whose output was avgpool2(e) (reference resnet.py:29: AvgPool2d's backward in front of the tail's):
    the reduction and the apply pass take the POOLED gradient and the apply pass writes the gradient of e -- every output
    bit-identical to avgpool2_bwd followed by the ordinary passes."""
    import torch
    torch.manual_seed(7)
    cr = max(c // 8, 4)
    cc, r = dev(torch.randn(n, h, w, c)), dev(torch.randn(n, h, w, c))
    gamma, beta = dev(torch.rand(c) + 0.5), dev(torch.randn(c))
    mean, invstd = dev(torch.randn(c) * 0.1), dev(torch.rand(c) + 0.5)
    scale, shift = ops.bn_scale_shift(gamma, beta, mean, invstd)
    fw1, fb1, fw2, fb2 = dev(torch.randn(cr, c) * 0.1), dev(torch.randn(cr) * 0.1), dev(torch.randn(c, cr) * 0.1), dev(torch.randn(c) * 0.1)
    ssum = cc.sum(dim=(1, 2)).contiguous()
    pooled, hid, s = ops.se_fc_fwd(ssum, scale, shift, fw1, fb1, fw2, fb2, h * w)
    _, bits = ops.se_tail_fwd(cc, r, scale, shift, s, want_mask=True)
    dpo = dev(torch.randn(n, h // 2, w // 2, c))
    de = ops.avgpool2_bwd(dpo, h, w)
    ref = ops.se_tail_bwd(de, None, cc, gamma, beta, mean, invstd, ssum, pooled, hid, s, fw1, fw2, want_dr=want_dr, mask=bits)
    de_out = torch.empty_like(cc)
    got = ops.se_tail_bwd(dpo, None, cc, gamma, beta, mean, invstd, ssum, pooled, hid, s, fw1, fw2, want_dr=want_dr, mask=bits,
                          pooled_hw=(h, w), de_out=de_out)
    torch.cuda.synchronize()
    assert torch.equal(de_out, de)
    for a, b, what in zip(got, ref, ("dc", "dr", "dgamma", "dbeta", "dw1", "db1", "dw2", "db2")):
        if a is None or b is None:
            assert a is None and b is None, what
        else:
            assert torch.equal(a, b), what


def test_avgpool(ops):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 32, 12, 16, generator=g, requires_grad=True)
    y_ref = F.avg_pool2d(x, 2, 2)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    y = ops.avgpool2(dev(nhwc(x.detach())))
    dx = ops.avgpool2_bwd(dev(nhwc(dy)), 12, 16)
    assert_close(nchw(y), y_ref, 1e-6, "avgpool fwd")
    assert_close(nchw(dx), x.grad, 1e-6, "avgpool bwd")


# ------------------------------------------------------------------------------------------------ blocks vs oracle
def _block_sd(prefix, cin, c, down, seed):
    from oracle.filler import fill_value
    names = [("conv1.weight", (c, cin, 3, 3)), ("bn1.weight", (c,)), ("bn1.bias", (c,)), ("bn1.running_mean", (c,)),
             ("bn1.running_var", (c,)), ("conv2.weight", (c, c, 3, 3)), ("bn2.weight", (c,)), ("bn2.bias", (c,)),
             ("bn2.running_mean", (c,)), ("bn2.running_var", (c,)), ("se.fc.0.weight", (c // 8, c)),
             ("se.fc.0.bias", (c // 8,)), ("se.fc.2.weight", (c, c // 8)), ("se.fc.2.bias", (c,))]
    if down:
        names += [("downsample.0.weight", (c, cin, 1, 1)), ("downsample.1.weight", (c,)), ("downsample.1.bias", (c,)),
                  ("downsample.1.running_mean", (c,)), ("downsample.1.running_var", (c,))]
    return {prefix + "." + k: fill_value("seed%d.%s.%s" % (seed, prefix, k), s) for k, s in names}


@pytest.mark.parametrize("cin,c,pool,n,h,w,training", [
    (32, 32, False, 2, 12, 64, True),
    (32, 64, True, 2, 16, 64, True),
    (64, 128, True, 2, 16, 32, True),
    (128, 256, False, 1, 10, 16, True),
    (256, 256, False, 2, 6, 16, True),
    (32, 64, True, 2, 16, 64, False),
    (64, 64, False, 1, 8, 32, False),
])
def test_se_basic_block_matches_oracle(ops, cin, c, pool, n, h, w, training):
    from oracle import seresnet as onet
    from adyolo_amd.models.backbones.resnet import SEBasicBlock, ConvParams, BatchNormParams
    import torch.nn as nn
    down = cin != c
    sd = _block_sd("blk", cin, c, down, seed=cin + c)
    dmod = nn.ModuleDict({"0": ConvParams(cin, c, 1, False), "1": BatchNormParams(c)}) if down else None
    blk = SEBasicBlock(cin, c, dmod, 2 if pool else None)
    blk.load_state_dict({k[4:]: v for k, v in sd.items()}, strict=False)
    blk = blk.to("cuda:0")
    blk.train(training)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g).relu()
    probe = torch.randn(n, c, h // 2 if pool else h, w // 2 if pool else w, generator=g)
    # oracle
    osd = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in sd.items()}
    xo = x.clone().requires_grad_(True)
    yo = onet.se_basic_block(osd, "blk", xo, (2, 2) if pool else None, training, update_stats=True)
    # device
    xg = dev(nhwc(x)).requires_grad_(True)
    yg = blk(xg)
    torch.cuda.synchronize()
    assert_close(nchw(yg), yo, 1e-4, "block forward")
    if training:
        (yo * probe).sum().backward()
        (yg * dev(nhwc(probe))).sum().backward()
        torch.cuda.synchronize()
        assert_close(nchw(xg.grad), xo.grad, 1e-3, "block dx")
        for name, prm in blk.named_parameters():
            assert_close(prm.grad, osd["blk." + name].grad, 1e-3, "block grad " + name)
        for name, buf in blk.named_buffers():
            if "running" in name:
                assert_close(buf, osd["blk." + name], 1e-4, "block buffer " + name)


def test_sap_gru_ln_linear_match_oracle(ops):
    from oracle import seresnet as onet
    from oracle.filler import fill_state_dict
    from adyolo_amd import functional as Fn
    enc, head = onet.split_state_dict(fill_state_dict(onet.state_dict_spec()))
    g = torch.Generator().manual_seed(31)
    bsz, t, f = 3, 11, 16
    x = torch.randn(bsz, t, f, 256, generator=g)
    # --- SAP
    xo = x.clone().requires_grad_(True)
    wo = enc["attention.W.weight"].clone().requires_grad_(True)
    bo = enc["attention.W.bias"].clone().requires_grad_(True)
    yo = onet.self_attention_pooling({"attention.W.weight": wo, "attention.W.bias": bo}, xo)
    probe = torch.randn(bsz, t, 256, generator=g)
    (yo * probe).sum().backward()
    xg, wg, bg = dev(x).requires_grad_(True), dev(wo.detach()).requires_grad_(True), dev(bo.detach()).requires_grad_(True)
    yg = Fn.SAPFn.apply(xg, wg, bg)
    (yg * dev(probe)).sum().backward()
    assert_close(yg, yo, 1e-5, "sap fwd")
    assert_close(xg.grad, xo.grad, 1e-4, "sap dx")
    assert_close(wg.grad, wo.grad, 1e-4, "sap dW")
    assert_close(bg.grad, bo.grad, 1e-4, "sap db")
    # --- BiGRU layer 0 and 1 (+ explicit mask in between), LN+tanh, head
    names = ["weight_ih", "weight_hh", "bias_ih", "bias_hh"]
    s = torch.randn(bsz, t, 256, generator=g)
    mask = (torch.rand(bsz, t, 256, generator=g) > 0.3).float() / 0.7
    osd = {k: v.clone().requires_grad_(True) for k, v in enc.items() if k.startswith(("lstm.", "norm."))}
    so = s.clone().requires_grad_(True)
    go = onet.bigru(osd, so, dropout_mask=mask)
    lo = torch.tanh(F.layer_norm(go, (256,), osd["norm.weight"], osd["norm.bias"], 1e-5))
    hsd = {k: v.clone().requires_grad_(True) for k, v in head.items()}
    ho = onet.adyolo_head(hsd, lo)
    probe2 = torch.randn(bsz, t, 2400, generator=g)
    (ho * probe2).sum().backward()
    gsd = {k: dev(v.detach()).requires_grad_(True) for k, v in osd.items()}
    ghd = {k: dev(v.detach()).requires_grad_(True) for k, v in hsd.items()}
    sg = dev(s).requires_grad_(True)

    def lp(layer):
        return [gsd["lstm.%s_l%d%s" % (nm, layer, sfx)] for sfx in ("", "_reverse") for nm in names]
    y0 = Fn.BiGRULayerFn.apply(sg, *lp(0), True)
    y0 = Fn.DropoutFn.apply(y0, dev(mask))
    y1 = Fn.BiGRULayerFn.apply(y0, *lp(1), True)
    yl = Fn.LNTanhFn.apply(y1, gsd["norm.weight"], gsd["norm.bias"], 1e-5)
    yh = Fn.LinearFn.apply(Fn.LinearFn.apply(yl, ghd["yolo_head.0.weight"], ghd["yolo_head.0.bias"]),
                           ghd["yolo_head.1.weight"], ghd["yolo_head.1.bias"])
    (yh * dev(probe2)).sum().backward()
    torch.cuda.synchronize()
    assert_close(y1, go, 1e-4, "bigru fwd")
    assert_close(yl, lo, 1e-4, "ln tanh fwd")
    assert_close(yh, ho, 1e-3, "head fwd")
    assert_close(sg.grad, so.grad, 1e-3, "gru dx")
    for k in osd:
        assert_close(gsd[k].grad, osd[k].grad, 1e-3, "grad " + k)
    for k in hsd:
        assert_close(ghd[k].grad, hsd[k].grad, 1e-3, "grad " + k)


def test_dropout_mask_statistics(ops):
    x = torch.ones(1 << 20, device="cuda:0")
    m = ops.dropout_mask(x, 0.3, 1234, 0)
    m2 = ops.dropout_mask(x, 0.3, 1234, 0)
    m3 = ops.dropout_mask(x, 0.3, 1234, 1 << 20)
    torch.cuda.synchronize()
    keep = float((m > 0).float().mean())
    assert abs(keep - 0.7) < 5e-3, keep
    assert torch.equal(m, m2) and not torch.equal(m, m3)
    vals = torch.unique(m).cpu().tolist()
    assert len(vals) == 2 and abs(vals[1] - 1 / 0.7) < 1e-6 and vals[0] == 0.0


def test_dropout_apply_regenerates_the_mask(ops):
    """The mask-free dropout (forward and backward) multiplies by exactly the values adyolo_dropout_mask writes for the same
    (seed, offset): bit-equal to x * mask, including a non-zero stream offset and a grid-stride tail."""
    from adyolo_amd import functional as Fn
    from adyolo_amd.rng import DropoutStream
    g = torch.Generator().manual_seed(9)
    for n, off in ((1 << 20, 0), (12 * 257 * 4, 777), (3 * 1000 * 1000 + 4, (1 << 33) + 5)):
        x = dev(torch.randn(n, generator=g))
        m = ops.dropout_mask(x, 0.2, 0xABCDEF0123, off)
        assert torch.equal(ops.dropout_apply(x, 0.2, 0xABCDEF0123, off), x * m)
    torch.manual_seed(5)
    s1, s2 = DropoutStream(0x11), DropoutStream(0x11)
    x = dev(torch.randn(4, 50, 256, generator=g)).requires_grad_(True)
    s1.draw(1000)
    s2.mask(torch.empty(1000, device="cuda:0"), 0.2)                     # both streams advanced by the same count
    y = Fn.DropoutHashFn.apply(x, 0.2, *s1.draw(x.numel()))
    mask = s2.mask(x, 0.2)
    assert s1.offset == s2.offset and torch.equal(y, x.detach() * mask)
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad, mask)


# ------------------------------------------------------------------------------------------------ loss
@pytest.mark.parametrize("tag,nb_classes", [("c12", 12), ("c13", 13), ("sat", 12)])
def test_loss_matches_golden_and_oracle(ops, tag, nb_classes):
    from oracle import adyolo_loss as oloss
    g = np.load(os.path.join(G, "adyolo_loss.npz"))
    logit = torch.from_numpy(g[tag + "_logit"])
    target = torch.from_numpy(g[tag + "_target"])
    loss, dlogit, dist = ops.adyolo_loss(dev(logit), dev(target), nb_classes, want_dist=True)
    torch.cuda.synchronize()
    lo = logit.clone().requires_grad_(True)
    ref, aux = oloss.adyolo_loss(lo, target, nb_classes, return_aux=True)
    assert_close(dist, aux["D"], 2e-5, "angular distances")
    assert_close(loss, torch.from_numpy(g[tag + "_loss"]), 1e-5, "loss vs golden")
    gold = torch.from_numpy(g[tag + "_dlogit"])
    err = float((dlogit.cpu() - gold).abs().max())
    assert err <= 1e-3 * float(gold.abs().max()), "dlogit abs err %.3e vs absmax %.3e" % (err, float(gold.abs().max()))


def test_loss_large_random_vs_oracle(ops):
    from oracle import adyolo_loss as oloss
    from adyolo_amd.datasets import synthetic_targets
    b, t, c = 4, 50, 12
    g = torch.Generator().manual_seed(77)
    logit = torch.randn(b, t, 8 * 4 * 5 * (c + 3), generator=g) * 1.5
    target = synthetic_targets(b, t, c, seed=5)
    lo = logit.clone().requires_grad_(True)
    ref = oloss.adyolo_loss(lo, target, c)
    ref.backward()
    loss, dlogit, _ = ops.adyolo_loss(dev(logit), dev(target), c)
    torch.cuda.synchronize()
    assert_close(loss, ref.detach(), 1e-5, "loss")
    err = float((dlogit.cpu() - lo.grad).abs().max())
    assert err < 2e-7 + 1e-3 * float(lo.grad.abs().max()), "dlogit abs err %.3e" % err


def test_other_losses_and_heads_match_reference_golden(ops):
    """SEDDOA / masked-SEDDOA / ACCDOA / ADPIT losses (value + gradient) and the head activations."""
    from adyolo_amd.models.loss import SEDDOAloss, ACCDOAloss, ADPITloss
    from adyolo_amd import functional as Fn
    from oracle import other_losses as ool
    g = np.load(os.path.join(G, "other_losses.npz"))
    cases = [("seddoa", SEDDOAloss(12, masked_mse=False), "sed_target"), ("masked", SEDDOAloss(12, masked_mse=True), "sed_target"),
             ("accdoa", ACCDOAloss(12), "accdoa_target"), ("adpit", ADPITloss(12), "adpit_target")]
    for tag, crit, tkey in cases:
        o = dev(torch.from_numpy(g[tag + "_out"])).requires_grad_(True)
        loss = crit(o, torch.from_numpy(g[tkey]))
        (loss * 1.0).backward()
        torch.cuda.synchronize()
        assert_close(loss, torch.from_numpy(g[tag + "_loss"]), 1e-5, tag + " loss")
        gold = torch.from_numpy(g[tag + "_dout"])
        err = float((o.grad.cpu() - gold).abs().max())
        assert err <= 1e-4 * float(gold.abs().max()) + 1e-9, "%s dout err %.3e (absmax %.3e)" % (tag, err, float(gold.abs().max()))
    gen = torch.Generator().manual_seed(8)
    raw = torch.randn(3, 5, 48, generator=gen)
    probe = torch.randn(3, 5, 48, generator=gen)
    ro = raw.clone().requires_grad_(True)
    (ool.head_activation(ro, 12) * probe).sum().backward()
    rg = dev(raw).requires_grad_(True)
    yg = Fn.ActFn.apply(rg, 12)
    (yg * dev(probe)).sum().backward()
    assert_close(yg, ool.head_activation(raw, 12), 1e-6, "head activation")
    assert_close(rg.grad, ro.grad, 1e-6, "head activation grad")


@pytest.mark.parametrize("loss_nm", ["adpit", "accdoa", "masked-seddoa"])
def test_other_loss_plugins_train_end_to_end(ops, loss_nm):
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.datasets import ClasswiseLabelEncoder
    torch.manual_seed(100)
    prm = _params()
    prm["args"]["loss"] = loss_nm
    model = WrapperModel((1, 7, 32, 64), (), prm).to("cuda:0")
    crit = WrapperCriterion(prm)
    enc = ClasswiseLabelEncoder(12)
    events = {0: [[3, 0, 10.0, 5.0]], 2: [[3, 0, 10.0, 5.0], [3, 1, -170.0, 40.0]], 5: [[1, 0, 0.0, 0.0], [2, 1, 90.0, 10.0]]}
    lab = {"adpit": enc.get_adpit_label, "accdoa": enc.get_accdoa_label, "masked-seddoa": enc.get_seddoa_label}[loss_nm]
    target = torch.stack([lab(events, 8), lab({}, 8)])
    x = torch.randn(2, 7, 32, 64).to("cuda:0")
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for _ in range(6):
        opt.zero_grad()
        loss = crit(model(x), target)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses


# ------------------------------------------------------------------------------------------------ features
def test_features_match_oracle(ops):
    from oracle import features as ofeat
    from adyolo_amd.features import FeatureExtractor, load_scaler_npz
    from adyolo_amd.datasets import synthetic_audio
    scaler = load_scaler_npz(os.path.join(G, "scaler_DCASE2021.npz"))
    audio = synthetic_audio(2, 24000 * 2, seed=9)                    # 2 clips x 2 s -> T = 80
    # make the second clip quiet in one channel so the top_db clip actually bites
    audio[1, :, 2] *= 1e-4
    fx = FeatureExtractor(scaler, "cuda:0")
    out_ref_layout = fx(dev(audio), channels_last8=False).cpu()
    out_cl8 = fx(dev(audio), channels_last8=True).cpu()
    torch.cuda.synchronize()
    for b in range(2):
        ref, _ = ofeat.get_feature(audio[b].double().numpy(), scaler)
        ref = torch.from_numpy(ref)
        assert_close(out_ref_layout[b, :4], ref[:4], 1e-3, "log-mel (clip %d)" % b)
        err_iv = float((out_ref_layout[b, 4:] - ref[4:]).abs().max())
        assert err_iv < 1e-3, "IV abs err %.3e" % err_iv
        assert_close(out_cl8[b, :, :, :7].permute(2, 0, 1), ref, 1e-3, "channels-last layout")
        assert float(out_cl8[b, :, :, 7].abs().max()) == 0.0


def test_features_match_reference_made_golden(ops):
    """K1 against ``features_ref.npz``: the features the reference's own ``get_feature`` (datasets.py:281-292, real scaler)
    produced for this int16 clip, with only librosa's three calls shimmed when the fixture was made -- a check of the GPU
    path that does not go through oracle/features.py's restatement of the intensity vector / z-score at all."""
    from adyolo_amd.features import FeatureExtractor, load_scaler_npz
    g = np.load(os.path.join(G, "features_ref.npz"))
    scaler = load_scaler_npz(os.path.join(G, "scaler_DCASE2021.npz"))
    pcm = torch.from_numpy(g["audio_pcm16"]).to("cuda:0").contiguous()
    audio = ops.pcm16_to_f32(pcm).view(1, -1, 4)
    got = FeatureExtractor(scaler, "cuda:0")(audio, channels_last8=False)[0].cpu()
    ref = torch.from_numpy(g["audio_feat"])
    assert_close(got[:4], ref[:4], 1e-3, "log-mel vs the reference-made golden")
    err_iv = float((got[4:] - ref[4:]).abs().max())
    assert err_iv < 1e-3, "IV abs err %.3e" % err_iv                   # absolute: z-scored IV reaches |x| ~ 30


def test_chunk_features_match_oracle_per_window(ops):
    """Offline chunking on the GPU (reference src/preprocess.py:13-84: windows of chunk_window_s at chunk_stride_s, each
    written as its own file and featurised on its own): ``chunk_offsets`` computes every window's features from the
    recording in place.  Each window must equal (a) the same kernel run on the materialised window, bit for bit, and
    (b) the float64 oracle applied to that window (own reflect padding at the window start, own top_db reference), 1e-3."""
    from oracle import features as ofeat
    from adyolo_amd.features import FeatureExtractor, load_scaler_npz
    from adyolo_amd.datasets import synthetic_audio
    scaler = load_scaler_npz(os.path.join(G, "scaler_DCASE2021.npz"))
    audio = synthetic_audio(2, 24000 * 6, seed=21)                   # 2 recordings x 6 s
    audio[0, 24000 * 2:24000 * 3] *= 30.0                            # a loud second: windows see different top_db references
    audio[1, :, 1] *= 1e-3
    win, stride, n = 24000 * 2, 24000, audio.shape[1]
    offs = [rec * n + o for rec in range(2) for o in range(0, n - win + 1, stride)]       # 5 windows per recording
    fx = FeatureExtractor(scaler, "cuda:0")
    dev_audio = dev(audio)
    got = fx(dev_audio, channels_last8=False, chunk_offsets=torch.tensor(offs, dtype=torch.int64, device="cuda:0"),
             chunk_samples=win)
    torch.cuda.synchronize()
    assert got.shape == (len(offs), 7, win // 600, 64)
    flat = audio.view(-1, 4)
    for i, o in enumerate(offs):
        chunk = flat[o:o + win].contiguous()
        same = fx(dev(chunk[None]), channels_last8=False)[0]
        assert torch.equal(got[i], same), "window %d differs from the kernel on the materialised window" % i
        ref = torch.from_numpy(ofeat.get_feature(chunk.double().numpy(), scaler)[0])
        assert_close(got[i, :4].cpu(), ref[:4], 1e-3, "log-mel of window %d" % i)
        assert float((got[i, 4:].cpu() - ref[4:]).abs().max()) < 1e-3
    # the windows are NOT slices of the recording's features: frame 0 is reflect-padded at the window start and the
    # top_db clip follows the window's own maximum (chunking)
    whole = fx(dev_audio, channels_last8=False)
    assert not torch.equal(got[1][:, :40], whole[0][:, 40:80])


# ------------------------------------------------------------------------------------------------ whole model
def _params(nb_classes=12):
    return {"args": {"device": "cuda:0", "encoder": "se-resnet34", "loss": "adyolo"},
            "data_config": {"nb_classes": nb_classes},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0, "class_gain": 3.0},
                             "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4", "winograd4-all"])
def test_wrapper_model_matches_reference_golden(ops, monkeypatch, algo):
    """Forward outputs (eval and train mode) hold the 1e-3 bar with either convolution algorithm, and every kernel on
    its own holds 2e-5 against torch.  Whole-model gradients on this golden (random filler weights, 34 normalised
    layers, 2 x 64 x 64 input) are ill-conditioned: the reference's OWN fp32 gradients deviate from an fp64 evaluation of
    the same graph by up to 2.6e-3 of absmax.  Measured on MI355X in round 1 (the mechanism -- ReLU mask flips -- is pinned down
    and removed from the comparison by test_gpu_parity_scale.py::test_seed100_training_step_matches_reference):
      direct    -- worst tensor 3.3e-4 of absmax vs fp64 (better than the reference's own fp32 path); bound: max(1e-3, 2x ref)
      winograd  -- activations differ from the direct run by <= 7e-6, which flips two ReLU masks of pre-activations
                   within 1e-6 of zero; a flipped mask is a different (equally valid) subgradient and moves the weight
                   gradients of those blocks by up to 1.7e-2 of absmax.  Bound: 5e-2 of absmax and 0.999 cosine."""
    if algo == "winograd4-all":                      # the F(4x4) kernel on every eligible layer, not only from 128 channels on
        monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")
        algo = "winograd4"
    monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
    from oracle.filler import fill_module_
    from adyolo_amd.wrapper import WrapperModel
    g = np.load(os.path.join(G, "encoder.npz"))
    model = WrapperModel((1, 7, 64, 64), (), _params())
    fill_module_(model)
    model = model.to("cuda:0")
    x = torch.from_numpy(g["x"])
    model.eval()
    with torch.no_grad():
        y = model.encoder(dev(x))
        hy = model.head(y)
        y1 = model.encoder(dev(x[:1]))
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g["y_eval"]), 1e-3, "encoder eval output vs reference")
    assert_close(y1, torch.from_numpy(g["y_eval_b1"]), 1e-3, "encoder eval output (batch 1) vs reference")
    assert_close(hy, torch.from_numpy(g["head_eval"]), 1e-3, "head output vs reference")
    # train mode (dropout off like the golden), gradients of <out, probe>
    model.train()
    model.encoder.lstm.dropout = 0.0
    xg = dev(x)
    y = model.encoder(xg)
    (y * dev(torch.from_numpy(g["probe"]))).sum().backward()
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g["y_train"]), 1e-3, "encoder train output vs reference")
    named = dict(model.encoder.named_parameters())
    sd = model.encoder.state_dict()
    # Gradients of the early layers are ill-conditioned in fp32 (34 normalised layers on a 2 x 64 x 64 input): the
    # reference's OWN fp32 gradients deviate from an fp64 evaluation of the same graph by up to 2.6e-3 of their
    # absmax.  So: (a) vs the reference golden, and (b) vs an fp64 evaluation of the oracle (bounds in the docstring).
    from oracle import seresnet as onet
    from oracle.filler import fill_state_dict
    enc64, _ = onet.split_state_dict(fill_state_dict(onet.state_dict_spec()))
    enc64 = {k: (v.double() if v.is_floating_point() else v) for k, v in enc64.items()}
    for k, v in enc64.items():
        if v.is_floating_point() and "running" not in k:
            v.requires_grad_(True)
    y64 = onet.encoder_forward(enc64, x.double(), training=True)
    (y64 * torch.from_numpy(g["probe"]).double()).sum().backward()
    worst = 0.0
    bad = []
    for key in g.files:
        if key.startswith("grad_"):
            ref = torch.from_numpy(g[key].reshape(-1))
            got = named[key[5:]].grad.reshape(-1)[:ref.numel()].cpu()
            t64 = enc64[key[5:]].grad.reshape(-1)[:ref.numel()]
            am = float(t64.abs().max())
            if am < 1e-9:
                continue
            ref_noise = float((ref.double() - t64).abs().max()) / am
            mine = float((got.double() - t64).abs().max()) / am
            worst = max(worst, mine)
            cos = float(torch.dot(got.double(), t64) / (got.double().norm() * t64.norm()))
            limit = max(1e-3, 2.0 * ref_noise) if algo == "direct" else 5e-2
            if mine > limit or cos < 0.999:
                bad.append("%s: %.2e of absmax vs fp64, cosine %.6f (reference fp32: %.2e)" % (key, mine, cos, ref_noise))
            assert_close(got, ref, 1e-2 if algo == "direct" else 5e-2, key + " vs reference golden")
        if key.startswith("stat_") and not key.endswith("num_batches_tracked"):
            assert_close(sd[key[5:]], torch.from_numpy(g[key]), 1e-4, key)
    assert int(sd["bn1.num_batches_tracked"]) == int(g["stat_bn1.num_batches_tracked"])
    print("[%s] worst gradient deviation from fp64: %.2e of absmax" % (algo, worst))
    assert not bad, "; ".join(bad)


def test_conformer_pieces_match_torch(ops):
    """implicit-GEMM convolution (7x7 s(1,2), 3x3 s(1,2), 1x1 s(1,2), the narrow stride-1 maps of the deep stages, a
    stride-(2,2) case with odd sizes for the transposed gather), max-pool, attention core, conv module pieces."""
    from adyolo_amd import functional as Fn
    g = torch.Generator().manual_seed(61)
    for (cin, cout, k, stride, pad, h, w) in [(8, 64, 7, (1, 2), (3, 3), 12, 64), (64, 128, 3, (1, 2), (1, 1), 10, 16),
                                              (128, 256, 1, (1, 2), (0, 0), 9, 8), (256, 512, 3, (1, 2), (1, 1), 6, 2),
                                              (256, 256, 3, (1, 1), (1, 1), 37, 2), (512, 512, (3, 1), (1, 1), (1, 0), 70, 1),
                                              (128, 128, 3, (1, 1), (1, 1), 33, 4), (12, 20, 3, (2, 2), (1, 1), 9, 7),
                                              (16, 36, (5, 3), (2, 1), (2, 0), 11, 5)]:
        kh_, kw_ = (k, k) if isinstance(k, int) else k
        x = torch.randn(2, cin, h, w, generator=g)
        wt = torch.randn(cout, cin, kh_, kw_, generator=g) / np.sqrt(cin * kh_ * kw_)
        xo, wo = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        yo = F.conv2d(xo, wo, None, stride=stride, padding=pad)
        probe = torch.randn_like(yo)
        (yo * probe).sum().backward()
        xg, wg = dev(nhwc(x)).requires_grad_(True), dev(wt).requires_grad_(True)
        yg = Fn.ConvFn.apply(xg, wg, stride, pad)
        (yg * dev(nhwc(probe))).sum().backward()
        assert_close(nchw(yg), yo, 3e-5, "strided conv fwd k=%s" % (k,))
        assert_close(nchw(xg.grad), xo.grad, 3e-5, "strided conv dx k=%s" % (k,))
        assert_close(wg.grad, wo.grad, 3e-5, "strided conv dw k=%s" % (k,))
    # the stride-1 3x3 router of the deep, narrow stages (W = 4: implicit GEMM, W = 2: bins folded into channels,
    # W = 1: centre column only) against conv2d, including the exactly-zero gradient of the taps that only see padding
    from adyolo_amd.models.backbones.resnet_conformer import _conv3x3_s1
    for (c, h, w) in [(128, 21, 4), (64, 19, 2), (96, 17, 1), (64, 16, 16)]:
        x = torch.randn(2, c, h, w, generator=g)
        wt = torch.randn(c, c, 3, 3, generator=g) / np.sqrt(9 * c)
        xo, wo = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        yo = F.conv2d(xo, wo, None, padding=1)
        probe = torch.randn_like(yo)
        (yo * probe).sum().backward()
        xg, wg = dev(nhwc(x)).requires_grad_(True), dev(wt).requires_grad_(True)
        yg = _conv3x3_s1(xg, wg)
        (yg * dev(nhwc(probe))).sum().backward()
        assert_close(nchw(yg), yo, 3e-5, "narrow conv fwd W=%d" % w)
        assert_close(nchw(xg.grad), xo.grad, 3e-5, "narrow conv dx W=%d" % w)
        assert_close(wg.grad, wo.grad, 3e-5, "narrow conv dw W=%d" % w)
        if w == 1:
            assert float(wg.grad[..., 0].abs().max()) == 0.0 and float(wg.grad[..., 2].abs().max()) == 0.0
    # max-pool 3x3 s(1,2) p1
    x = torch.randn(2, 64, 9, 32, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = F.max_pool2d(xo, 3, stride=(1, 2), padding=1)
    probe = torch.randn_like(yo)
    (yo * probe).sum().backward()
    xg = dev(nhwc(x)).requires_grad_(True)
    yg = Fn.MaxPool3Fn.apply(xg)
    (yg * dev(nhwc(probe))).sum().backward()
    assert_close(nchw(yg), yo, 1e-6, "maxpool fwd")
    assert_close(nchw(xg.grad), xo.grad, 1e-6, "maxpool bwd")
    # attention core
    b, t, heads, d = 2, 36, 4, 64
    q, k, v = (torch.randn(b, t, heads * d, generator=g) for _ in range(3))
    qo, ko, vo = (z.clone().requires_grad_(True) for z in (q, k, v))
    qh, kh, vh = (z.view(b, t, heads, d).transpose(1, 2) for z in (qo, ko, vo))
    ctx_o = (torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1) @ vh).transpose(1, 2).reshape(b, t, heads * d)
    probe = torch.randn(b, t, heads * d, generator=g)
    (ctx_o * probe).sum().backward()
    qg, kg, vg = (dev(z).requires_grad_(True) for z in (q, k, v))
    ctx_g = Fn.AttentionCoreFn.apply(qg, kg, vg, heads, d ** -0.5, None)
    (ctx_g * dev(probe)).sum().backward()
    assert_close(ctx_g, ctx_o, 2e-5, "attention fwd")
    for name, a_, b_ in (("dq", qg, qo), ("dk", kg, ko), ("dv", vg, vo)):
        assert_close(a_.grad, b_.grad, 5e-5, "attention " + name)
    # GLU, Swish, depthwise conv, avg-pool, LayerNorm
    x = torch.randn(3, 20, 512, generator=g)
    xo = x.clone().requires_grad_(True)
    yo = F.glu(xo, dim=-1)
    probe = torch.randn_like(yo)
    (yo * probe).sum().backward()
    xg = dev(x).requires_grad_(True)
    yg = Fn.GLUFn.apply(xg)
    (yg * dev(probe)).sum().backward()
    assert_close(yg, yo, 1e-6, "glu fwd")
    assert_close(xg.grad, xo.grad, 1e-6, "glu bwd")
    for dil in (1, 4, 16):
        x = torch.randn(2, 40, 256, generator=g)
        w = torch.randn(256, 1, 3, generator=g)
        bias = torch.randn(256, generator=g)
        xo, wo, bo = (z.clone().requires_grad_(True) for z in (x, w, bias))
        yo = F.conv1d(xo.transpose(1, 2), wo, bo, padding=dil, dilation=dil, groups=256).transpose(1, 2)
        probe = torch.randn(2, 40, 256, generator=g)
        (yo * probe).sum().backward()
        xg, wg, bg = (dev(z).requires_grad_(True) for z in (x, w, bias))
        yg = Fn.DWConv3Fn.apply(xg, wg, bg, dil)
        (yg * dev(probe)).sum().backward()
        assert_close(yg, yo, 1e-5, "dwconv fwd")
        assert_close(xg.grad, xo.grad, 1e-5, "dwconv dx")
        assert_close(wg.grad, wo.grad, 1e-4, "dwconv dw")
        assert_close(bg.grad, bo.grad, 1e-4, "dwconv db")
    x = torch.randn(2, 40, 256, generator=g)
    xo = x.clone().requires_grad_(True)
    gam, bet = torch.rand(256, generator=g) + 0.5, torch.randn(256, generator=g)
    go, bo = gam.clone().requires_grad_(True), bet.clone().requires_grad_(True)
    yo = F.layer_norm(xo * torch.sigmoid(xo), (256,), go, bo, 1e-5)
    yo = (F.avg_pool1d(yo.transpose(1, 2), 4) * 2).transpose(1, 2)
    probe = torch.randn_like(yo)
    (yo * probe).sum().backward()
    xg, gg, bg = dev(x).requires_grad_(True), dev(gam).requires_grad_(True), dev(bet).requires_grad_(True)
    yg = Fn.AvgPool1dFn.apply(Fn.LNFn.apply(Fn.SwishFn.apply(xg), gg, bg, 1e-5), 4, 2.0)
    (yg * dev(probe)).sum().backward()
    assert_close(yg, yo, 1e-5, "swish+ln+avgpool fwd")
    assert_close(xg.grad, xo.grad, 1e-4, "swish+ln+avgpool dx")
    assert_close(gg.grad, go.grad, 1e-4, "ln dgamma")
    assert_close(bg.grad, bo.grad, 1e-4, "ln dbeta")


def test_resnet_conformer_matches_reference_golden(ops):
    from oracle.filler import fill_module_
    from adyolo_amd.wrapper import WrapperModel
    g = np.load(os.path.join(G, "conformer.npz"))
    prm = _params()
    prm["args"]["encoder"] = "resnet-conformer"
    model = WrapperModel((1, 7, 32, 64), (), prm)
    fill_module_(model)
    enc = model.encoder.to("cuda:0")
    x = torch.from_numpy(g["x"])
    enc.eval()
    with torch.no_grad():
        y = enc(dev(x))
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g["y_eval"]), 1e-3, "conformer eval output vs reference")
    enc.train()
    for m in enc.modules():
        if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
            m.p = 0.0
        if hasattr(m, "p2"):
            m.p2 = 0.0
    y = enc(dev(x))
    (y * dev(torch.from_numpy(g["probe"]))).sum().backward()
    torch.cuda.synchronize()
    assert_close(y, torch.from_numpy(g["y_train"]), 1e-3, "conformer train output vs reference")
    named = dict(enc.named_parameters())
    for key in g.files:
        if key.startswith("grad_"):
            ref = torch.from_numpy(g[key].reshape(-1))
            got = named[key[5:]].grad.reshape(-1)[:ref.numel()]
            assert_close(got, ref, 1e-2, key + " vs reference golden")


@pytest.mark.parametrize("nms", ["conn-merge", "soft-merge", "default"])
def test_postprocess_matches_reference_golden(ops, nms, tmp_path):
    from adyolo_amd.postprocess import LabelPostProcessor, write_seld_output_file
    from oracle import postprocess as opp
    g = np.load(os.path.join(G, "postprocess.npz"))
    prm = _params()
    prm["train_config"].update(conf_thresh=0.5, clss_thresh=0.5, unify_thresh=15.0, nms=nms)
    pp = LabelPostProcessor(prm)
    logit = dev(torch.from_numpy(g["logit"]))
    dec = ops.yolo_decode(logit, 12)
    assert_close(dec, torch.from_numpy(opp.decode(g["logit"], 12)), 1e-5, "decode vs oracle")
    res = pp.postprocess(logit)
    rows = np.asarray([[fr] + [float(x) for x in d] for fr, dets in res.items() for d in dets], dtype=np.float64)
    ref = g["rows_" + nms]
    assert rows.shape == ref.shape
    np.testing.assert_array_equal(rows[:, :2], ref[:, :2])
    np.testing.assert_allclose(rows[:, 2:], ref[:, 2:], rtol=0, atol=5e-5)
    f = tmp_path / "out.csv"
    write_seld_output_file(f, res)
    assert len(open(f).read().strip().splitlines()) == len(rows)


@pytest.mark.parametrize("b,t,p", [(2, 36, 0.0), (1, 50, 0.0), (3, 131, 0.2), (2, 800, 0.2), (1, 2400, 0.0)])
def test_flash_attention_matches_torch(ops, b, t, p):
    """Flash-style attention core (csrc/attention.hip) against the materialised PyTorch-CPU form of
    MultiHeadAttention.forward (resnet_conformer.py:57-85: softmax(q k^T d^-1/2) -> dropout -> @ v), forward 2e-5 and
    dq / dk / dv 5e-5 of absmax: ragged lengths (T not a multiple of the 32-key block or the 128-query tile), the
    training length 800 with dropout (the in-kernel mask is materialised by adyolo_attn_dropout_mask for the reference)
    and the evaluation length 2400."""
    from adyolo_amd import functional as Fn
    heads, d = 4, 64
    g = torch.Generator().manual_seed(b * 1000 + t)
    q, k, v = (torch.randn(b, t, heads * d, generator=g) for _ in range(3))
    probe = torch.randn(b, t, heads * d, generator=g)
    seed = 0xC0FFEE + t
    mask = None
    if p > 0:
        mask = ops.attn_dropout_mask(dev(q), b, t, heads, p, seed).cpu()
        keep = float((mask > 0).float().mean())
        assert abs(keep - (1 - p)) < 0.01 and set(torch.unique(mask).tolist()) <= {0.0, float(np.float32(1.0 / (1.0 - p)))}
    qo, ko, vo = (z.clone().requires_grad_(True) for z in (q, k, v))
    qh, kh, vh = (z.view(b, t, heads, d).transpose(1, 2) for z in (qo, ko, vo))
    w = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, -1)
    if mask is not None:
        w = w * mask
    ctx_o = (w @ vh).transpose(1, 2).reshape(b, t, heads * d)
    (ctx_o * probe).sum().backward()
    qg, kg, vg = (dev(z).requires_grad_(True) for z in (q, k, v))
    ctx_g = Fn.AttentionCoreFn.apply(qg, kg, vg, heads, d ** -0.5, (p, seed) if p > 0 else None)
    (ctx_g * dev(probe)).sum().backward()
    torch.cuda.synchronize()
    assert_close(ctx_g, ctx_o, 2e-5, "attention fwd (T=%d)" % t)
    for name, a_, b_ in (("dq", qg, qo), ("dk", kg, ko), ("dv", vg, vo)):
        assert_close(a_.grad, b_.grad, 5e-5, "attention %s (T=%d)" % (name, t))


def test_train_one_epoch_and_test_epoch_entry_points(ops, tmp_path):
    """train_one_epoch / test_epoch mirrors on a tiny in-memory loader; CSVs are scored by the SELD evaluator."""
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.train import train_one_epoch
    from adyolo_amd.test import test_epoch
    from adyolo_amd.postprocess import LabelPostProcessor
    from adyolo_amd.seld_metrics import ComputeSELDResults
    from adyolo_amd.datasets import synthetic_targets
    torch.manual_seed(100)
    prm = _params()
    prm["args"]["quick_test"] = False
    prm["data_config"].update(sr=24000, label_hop_len_s=0.1)
    prm["train_config"].update(conf_thresh=0.3, clss_thresh=0.3, unify_thresh=15.0, nms="conn-merge")
    model = WrapperModel((1, 7, 40, 64), (), prm).to("cuda:0")
    crit = WrapperCriterion(prm)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(5)
    loader = [(torch.randn(2, 7, 40, 64, generator=g), synthetic_targets(2, 10, 12, seed=s)) for s in range(3)]
    l1 = train_one_epoch(prm, loader, model, opt, crit, "cuda:0")
    l2 = train_one_epoch(prm, loader, model, opt, crit, "cuda:0")
    assert np.isfinite(l1) and np.isfinite(l2) and l2 < l1
    files = ["fold6_room1_mix001", "fold6_room1_mix002"]
    eval_loader = [(torch.randn(1, 7, 40, 64, generator=g), synthetic_targets(1, 10, 12, seed=9 + i)) for i in range(2)]
    out_dir, ref_dir = tmp_path / "output_val", tmp_path / "ref"
    ref_dir.mkdir()
    for name in files:
        with open(ref_dir / (name + ".csv"), "w") as f:
            for fr in range(10):
                f.write("%d,%d,0,%d,%d\n" % (fr, fr % 12, 10 * fr - 40, 5))
    loss = test_epoch(eval_loader, files, model, crit, LabelPostProcessor(prm), "cuda:0", str(out_dir))
    assert np.isfinite(loss) and sorted(os.listdir(out_dir)) == [n + ".csv" for n in files]
    er, f1, le, lr, seld, cw = ComputeSELDResults(prm, str(ref_dir)).get_SELD_Results(str(out_dir))
    assert 0.0 <= seld <= 1.5 and cw.shape == (5, 12)


def test_rotation_audio_matches_reference(ops):
    from adyolo_amd.augmentations import rotate_audio
    g = np.load(os.path.join(G, "rotation.npz"))
    audio = torch.from_numpy(g["audio"].astype(np.float32))
    batch = dev(audio[None].repeat(16, 1, 1))
    out = rotate_audio(batch, list(range(16))).cpu().numpy()
    np.testing.assert_array_equal(out, g["audio_rot"].astype(np.float32))


def test_adam_matches_torch(ops):
    g = torch.Generator().manual_seed(2)
    p0 = torch.randn(10001 + 3, generator=g)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3)
    pg, m, v = dev(p0), torch.zeros(p0.numel(), device="cuda:0"), torch.zeros(p0.numel(), device="cuda:0")
    for step in range(1, 6):
        grad = torch.randn(p0.numel(), generator=g)
        ref.grad = grad.clone()
        opt.step()
        ops.adam_step(pg, dev(grad * 2.0), m, v, step, grad_scale=0.5)
    torch.cuda.synchronize()
    assert_close(pg, ref.detach(), 1e-6, "adam params")


def test_train_step_runs_and_learns(ops):
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    torch.manual_seed(100)
    prm = _params()
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    step = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
    audio = dev(synthetic_audio(2, 24000 * 2, seed=3))
    target = synthetic_targets(2, 20, 12, seed=3)
    losses = [float(step.step(audio, target)) for _ in range(8)]
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses


# ------------------------------------------------------------------------------------------------ input pipeline (8f rows 2-4)
@pytest.mark.parametrize("n", [8 * 1000, 8 * 1000 + 5, 3])
def test_pcm16_to_f32_matches_numpy(ops, n):
    rs = np.random.RandomState(n)
    pcm = rs.randint(-32768, 32768, size=n).astype(np.int16)
    got = ops.pcm16_to_f32(torch.from_numpy(pcm).to("cuda:0"))
    torch.cuda.synchronize()
    ref = (pcm.astype(np.float64) / 32768.0 + 1e-8).astype(np.float32)       # datasets.py:105, cast like torch.Tensor(...)
    assert np.array_equal(got.cpu().numpy(), ref)


def test_audio_stager_double_buffers(ops):
    from adyolo_amd.datasets import AudioStager
    rs = np.random.RandomState(0)
    st = AudioStager(3, 2400, "cuda:0")
    batches = [rs.randint(-32768, 32768, size=(3, 2400, 4)).astype(np.int16) for _ in range(4)]
    for i, pcm in enumerate(batches):
        st.stage(list(pcm), workers=1 + i)            # (1, 2, 3 staging threads; 4 > batch: clamped)
        a = st.get()
        torch.cuda.synchronize()
        assert a.shape == (3, 2400, 4) and a.dtype == torch.float32
        assert np.array_equal(a.cpu().numpy(), (pcm.astype(np.float64) / 32768.0 + 1e-8).astype(np.float32))


def test_audio_stager_reraises_worker_exceptions(ops):
    """An exception inside a staging worker thread (here: a clip that cannot be converted to int16 samples) must surface in
    ``stage()`` -- it used to die with the thread and the previous batch's audio was shipped again (round 4, ADVICE)."""
    from adyolo_amd.datasets import AudioStager
    st = AudioStager(4, 240, "cuda:0")
    good = [np.zeros((240, 4), dtype=np.int16) for _ in range(4)]
    st.stage(good, workers=2)

    class Broken:
        shape = (240, 4)

        def __array__(self, *a, **kw):
            raise TypeError("clip cannot be read")
    bad = list(good)
    bad[3] = Broken()
    with pytest.raises(TypeError, match="clip cannot be read"):
        st.stage(bad, workers=2)


def test_specaug_masking_matches_numpy(ops):
    from adyolo_amd.augmentations import SpecAug
    g = torch.Generator().manual_seed(9)
    x = torch.randn(5, 40, 64, 8, generator=g)
    rng = torch.tensor([[3, 11, 0, 0], [0, 0, 60, 64], [0, 40, 5, 6], [0, 0, 0, 0], [39, 40, 0, 64]], dtype=torch.int32)
    ref = x.clone()
    for b, (t0, t1, f0, f1) in enumerate(rng.tolist()):
        ref[b, t0:t1] = 0.0
        ref[b, :, f0:f1] = 0.0
    sa = SpecAug({"aug_config": {"spec_augment": True, "spec_augment_thresh": 1.0, "spec_augment_time_mask_param": 8,
                                 "spec_augment_freq_mask_param": 8}}, is_valid=False)
    got = sa.augment(dev(x), ranges=rng)
    torch.cuda.synchronize()
    assert torch.equal(got.cpu(), ref)
    again = sa.augment(dev(x))                  # drawn ranges: only zeros are introduced
    torch.cuda.synchronize()
    a = again.cpu()
    assert bool(((a == x) | (a == 0)).all())


def test_scaler_fitter_matches_numpy_and_oracle(ops):
    """mean / std / max / min of the unscaled features over all frames (preprocess.py:117-126, population std)."""
    from adyolo_amd.datasets import synthetic_audio
    from adyolo_amd.features import FeatureExtractor, ScalerFitter
    from oracle import features as ofeat
    audio = synthetic_audio(3, 24000, seed=77)
    fit = ScalerFitter("cuda:0")
    fit.partial_fit(dev(audio[:2].contiguous()))
    fit.partial_fit(dev(audio[2:].contiguous()))
    sc = fit.finalize()
    torch.cuda.synchronize()
    feat = FeatureExtractor(None, "cuda:0")(dev(audio), channels_last8=True).cpu().double().numpy()    # (3, 40, 64, 8)
    flat = feat.reshape(-1, 64, 8)
    for key, sl in (("MEL", slice(0, 4)), ("IV", slice(4, 7))):
        assert sc[key]["mean"].shape == (1, 64, sl.stop - sl.start)
        np.testing.assert_allclose(sc[key]["mean"][0], flat[:, :, sl].mean(0), rtol=0, atol=2e-6 * np.abs(flat).max())
        np.testing.assert_allclose(sc[key]["std"][0], flat[:, :, sl].std(0), rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(sc[key]["max"][0], flat[:, :, sl].max(0))        # K1 is bit-reproducible
        np.testing.assert_array_equal(sc[key]["min"][0], flat[:, :, sl].min(0))
    # against the float64 CPU oracle of the reference feature code (features differ at the fp32 FFT level)
    mel = ofeat.mel_filterbank()
    o = np.stack([ofeat.get_feature(audio[b].double().numpy(), None, mel)[0] for b in range(3)])        # (3, 7, 40, 64)
    om = o.transpose(0, 2, 3, 1).reshape(-1, 64, 7)
    np.testing.assert_allclose(sc["MEL"]["mean"][0], om[:, :, :4].mean(0), atol=2e-3)
    np.testing.assert_allclose(sc["MEL"]["std"][0], om[:, :, :4].std(0), atol=2e-3)
    np.testing.assert_allclose(sc["IV"]["mean"][0], om[:, :, 4:].mean(0), atol=2e-4)
    np.testing.assert_allclose(sc["IV"]["std"][0], om[:, :, 4:].std(0), atol=2e-4)


def test_resume_from_a_torch_adam_checkpoint_continues_identically(ops, tmp_path):
    """A reference-side run (torch.optim.Adam on the CPU) checkpointed after 2 steps and resumed on the fused gfx950 Adam
    takes the same third step (train.py:148-150 resume path)."""
    from adyolo_amd import checkpoint as ck
    from adyolo_amd.dist import FlatParameters
    from adyolo_amd.train import FusedAdam
    from adyolo_amd.wrapper import WrapperModel
    torch.manual_seed(11)
    model = WrapperModel((1, 7, 64, 64), (), _params()).to("cuda:0")
    twin = [torch.nn.Parameter(p.detach().cpu().clone()) for p in model.parameters()]
    adam = torch.optim.Adam(twin, lr=1e-3)
    grads = [[torch.randn_like(p) for p in twin] for _ in range(3)]
    for s in range(2):
        for p, g_ in zip(twin, grads[s]):
            p.grad = g_.clone()
        adam.step()
    path = os.path.join(tmp_path, "model_ckpt.h5")
    sd = dict(zip(model.state_dict().keys(), [None] * 1000))
    msd = model.state_dict()
    pnames = [k for k, _ in model.named_parameters()]
    for k, p in zip(pnames, twin):
        msd[k] = p.detach().clone()
    torch.save({"start_epoch_nb": 2, "model_state_dict": {k: v.cpu() for k, v in msd.items()},
                "optim_state_dict": adam.state_dict(), "confidence_thresh": 0.5, "rng_state": None, "best_log": {},
                "train_remaining_file": []}, path)
    flat = FlatParameters(model)
    opt = FusedAdam(flat)
    ck.load_checkpoint(path, model, opt, device="cuda:0")
    for p, g_ in zip(model.parameters(), grads[2]):
        p.grad.copy_(g_.to("cuda:0"))
    opt.step()
    for p, g_ in zip(twin, grads[2]):
        p.grad = g_.clone()
    adam.step()
    torch.cuda.synchronize()
    for (k, p), q in zip(model.named_parameters(), twin):
        assert_close(p, q, 1e-6, "parameter %s after the resumed step" % k)


def test_raw_audio_epoch_end_to_end(ops, tmp_path):
    """WAV/CSV files -> FoaDataset -> DataLoader(audio_collate_fn) -> staged int16 -> GPU normalise / rotate / K1 / encoder /
    loss / backward / Adam: the epoch equals the same steps driven by hand on host-converted, host-rotated audio."""
    import random
    from scipy.io import wavfile
    from adyolo_amd.augmentations import rotate_audio
    from adyolo_amd.datasets import FoaDataset, audio_collate_fn
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.train import TrainStep, train_one_epoch_audio
    from adyolo_amd.wrapper import WrapperCriterion, WrapperModel
    rs = np.random.RandomState(1)
    sub = "dev-train-chunked_2s_1s"
    wdir, cdir = os.path.join(tmp_path, "foa_dev", sub), os.path.join(tmp_path, "metadata_dev", sub)
    os.makedirs(wdir), os.makedirs(cdir)
    for i in range(4):
        wavfile.write(os.path.join(wdir, "c%d.wav" % i), 24000, rs.randint(-8000, 8000, size=(48000, 4)).astype(np.int16))
        with open(os.path.join(cdir, "c%d.csv" % i), "w") as f:
            for fr in range(0, 20, 2):
                f.write("%d,%d,0,%d,%d\n" % (fr, (fr + i) % 12, (fr * 41 + i * 90) % 360 - 180, (fr * 7) % 120 - 60))
    prm = _params()
    prm["aug_config"] = {"rotation_augment": True}
    prm["data_config"].update({"data_pth": str(tmp_path), "chunk_window_s": 2, "chunk_stride_s": 1})
    prm["train_config"].update({"batch_size": 2, "nb_iters": 2})

    def make():
        torch.manual_seed(5)
        model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
        model.encoder.lstm.dropout = 0.0
        return TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)

    random.seed(3)
    ds = FoaDataset(prm, "train")
    files = list(ds.get_filelist())
    random.seed(21)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, collate_fn=audio_collate_fn, num_workers=0)
    tr = make()
    mean_loss = train_one_epoch_audio(prm, loader, tr)
    torch.cuda.synchronize()
    # by hand: same items (same random stream for the rotation draw), host float conversion
    random.seed(21)
    tr2 = make()
    losses = []
    for b0 in (0, 2):
        items = [ds[b0], ds[b0 + 1]]
        pcm, combs, target = audio_collate_fn(items)
        audio = (pcm.double() / 32768.0 + 1e-8).float().to("cuda:0")
        audio = rotate_audio(audio.contiguous(), combs)
        losses.append(float(tr2.step(audio, target)))
    torch.cuda.synchronize()
    assert ds.get_filelist() == files
    assert abs(mean_loss - sum(losses) / 2) <= 1e-3 * abs(mean_loss), (mean_loss, losses)
    # The whole step is bit-reproducible (K1 combines its mel pieces in a fixed order, the loss accumulates the angular
    # gradient in fixed point, every reduction is staged), so the two drives of the same steps agree exactly.
    assert mean_loss == sum(losses) / 2 or abs(mean_loss - sum(losses) / 2) <= 1e-7 * abs(mean_loss)
    for (k, p), (_, q) in zip(tr.model.named_parameters(), tr2.model.named_parameters()):
        assert torch.equal(p, q), "parameter %s differs by %.2e" % (k, float((p - q).abs().max()))


def test_raw_audio_evaluation_writes_the_same_files_as_the_feature_path(ops, tmp_path):
    """test_epoch_audio (WAV -> GPU normalise -> K1 -> model -> decode/NMS -> CSV) against test_epoch fed with features
    computed from host-normalised audio: same loss, same prediction files."""
    from scipy.io import wavfile
    from adyolo_amd import test as atest
    from adyolo_amd.datasets import FoaDataset, audio_collate_fn
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.postprocess import LabelPostProcessor
    from adyolo_amd.wrapper import WrapperCriterion, WrapperModel
    rs = np.random.RandomState(2)
    wdir, cdir = os.path.join(tmp_path, "foa_dev", "dev-test"), os.path.join(tmp_path, "metadata_dev", "dev-test")
    os.makedirs(wdir), os.makedirs(cdir)
    for i in range(2):
        wavfile.write(os.path.join(wdir, "t%d.wav" % i), 24000, rs.randint(-8000, 8000, size=(48000 + 123 * i, 4)).astype(np.int16))
        with open(os.path.join(cdir, "t%d.csv" % i), "w") as f:
            for fr in range(0, 20, 3):
                f.write("%d,%d,0,%d,%d\n" % (fr, fr % 12, (fr * 53) % 360 - 180, (fr * 9) % 100 - 50))
    prm = _params()
    prm["data_config"]["data_pth"] = str(tmp_path)
    prm["train_config"].update({"conf_thresh": 0.3, "clss_thresh": 0.3, "unify_thresh": 15.0, "nms": "conn-merge"})
    torch.manual_seed(8)
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    crit, post, fx = WrapperCriterion(prm), LabelPostProcessor(prm), FeatureExtractor(None, "cuda:0")
    ds = FoaDataset(prm, "test", is_valid=True)
    out_a, out_b = os.path.join(tmp_path, "out_audio"), os.path.join(tmp_path, "out_feat")
    loss_a = atest.test_epoch_audio(ds, model, fx, crit, post, "cuda:0", out_a)
    loader = []
    for i in range(len(ds)):
        pcm, _, rows = ds[i]
        t = (pcm.shape[0] // 600) * 600
        audio = torch.from_numpy((pcm[:t].astype(np.float64) / 32768.0 + 1e-8).astype(np.float32)).to("cuda:0").view(1, t, 4)
        feat = fx(audio.contiguous(), channels_last8=False)                   # reference layout (1, 7, T, 64)
        loader.append((feat, audio_collate_fn([(pcm, 0, rows)])[2]))
    loss_b = atest.test_epoch(loader, ds.get_filelist(), model, crit, post, "cuda:0", out_b)
    assert abs(loss_a - loss_b) <= 1e-4 * abs(loss_b)
    for nm in ds.get_filelist():
        a = open(os.path.join(out_a, nm + ".csv")).read().splitlines()
        b = open(os.path.join(out_b, nm + ".csv")).read().splitlines()
        assert len(a) == len(b)
        for la, lb in zip(a, b):
            fa, fb = la.split(","), lb.split(",")
            assert fa[:3] == fb[:3] and np.allclose([float(v) for v in fa[3:]], [float(v) for v in fb[3:]], atol=1e-4)


# ------------------------------------------------------------------------------------------------ MIC features (config 5)
def _mic_audio(b, n, seed):
    """Four microphones = one noise source seen with small integer delays + independent sensor noise (so the GCC-PHAT has
    real peaks and no numerically empty bins)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((b, n, 4), dtype=np.float64)
    for i in range(b):
        base = rng.normal(0.0, 0.1, size=n + 64)
        delays = rng.integers(-10, 11, size=4)
        for c in range(4):
            out[i, :, c] = base[32 - delays[c]:32 - delays[c] + n] + rng.normal(0.0, 0.01, size=n)
    pcm = np.clip(np.round(out * 32768.0), -32768, 32767)
    return torch.from_numpy((pcm / 32768.0 + 1e-8).astype(np.float32))


def test_mic_gcc_phat_features_match_oracle(ops):
    """K1m (csrc/features_mic.hip): the MIC-format feature set of BASELINE config 5 -- four log-mel + six GCC-PHAT channels --
    against the float64 oracle (oracle/features.py::get_feature_mic; PARITY UNPINNED: GCC-PHAT is not in the reference, the
    oracle restates the DCASE2022 baseline definition and is pinned by its delay property in test_oracle_golden.py)."""
    from oracle import features as ofeat
    from adyolo_amd.features import MicFeatureExtractor
    audio = _mic_audio(2, 24000 * 2, seed=11)
    rng = np.random.default_rng(3)
    scaler = {"MEL": {"mean": rng.normal(-40, 5, (1, 64, 4)), "std": rng.uniform(5, 15, (1, 64, 4))},
              "GCC": {"mean": rng.normal(0, 0.01, (1, 64, 6)), "std": rng.uniform(0.05, 0.2, (1, 64, 6))}}
    fx = MicFeatureExtractor(scaler, "cuda:0")
    got32 = fx(dev(audio))
    got = fx(dev(audio), channels_last=False).cpu()
    torch.cuda.synchronize()
    assert got32.shape == (2, 80, 64, 32) and float(got32[..., 10:].abs().max()) == 0.0
    for b in range(2):
        ref = torch.from_numpy(ofeat.get_feature_mic(audio[b].double().numpy(), scaler)[0])
        assert_close(got[b, :4], ref[:4], 1e-3, "log-mel of the microphones (clip %d)" % b)
        err = float((got[b, 4:] - ref[4:]).abs().max())
        assert err < 1e-3, "GCC-PHAT abs err %.3e (z-scored values reach %.1f)" % (err, float(ref[4:].abs().max()))
        assert float(ref[4:].abs().max()) > 3.0                          # the delay peaks are there
        assert torch.equal(got32[b, :, :, :10].permute(2, 0, 1).cpu(), got[b])


def test_config5_mic_model_train_step_matches_oracle(ops):
    """BASELINE config 5 end to end at test size: MIC audio -> (10, T, 64) features -> SE-ResNet34 with a 10-channel stem
    (32-channel pixels, Winograd stem convolution) -> ADPIT head + loss, one training step against the oracle on the same
    features: output 1e-3, loss 1e-3, stem weight gradient (Cin_real = 10 of 32 padded channels) and head gradients."""
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import MicFeatureExtractor
    from adyolo_amd.datasets import ClasswiseLabelEncoder
    from oracle import features as ofeat, seresnet as onet, other_losses as ol
    prm = _params()
    prm["args"]["loss"] = "adpit"
    torch.manual_seed(100)
    model = WrapperModel((1, 10, 80, 64), (), prm).to("cuda:0")
    assert model.encoder.conv1.weight.shape == (32, 10, 3, 3)
    crit = WrapperCriterion(prm)
    model.train()
    model.encoder.lstm.dropout = 0.0
    audio = _mic_audio(3, 24000 * 2, seed=12)
    enc = ClasswiseLabelEncoder(12)
    ev = {0: [[3, 0, 10.0, 5.0]], 2: [[3, 0, 10.0, 5.0], [3, 1, -170.0, 40.0]], 5: [[1, 0, 0.0, 0.0], [2, 1, 90.0, 10.0]]}
    target = torch.stack([enc.get_adpit_label(ev, 20), enc.get_adpit_label({}, 20), enc.get_adpit_label({7: [[4, 0, 1.0, 2.0]]}, 20)])
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    feat = MicFeatureExtractor(None, "cuda:0")(dev(audio))
    out = model(feat, channels_last8=True)
    loss = crit(out, target)
    loss.backward()
    torch.cuda.synchronize()
    f_ref = torch.stack([torch.from_numpy(ofeat.get_feature_mic(audio[b].double().numpy())[0]) for b in range(3)])
    assert float((feat[..., :10].permute(0, 3, 1, 2).cpu() - f_ref).abs().max()) < 1e-3
    names = ["encoder.conv1.weight", "head.adpit_head.0.weight", "head.adpit_head.1.weight"]
    for n in names:
        sd[n].requires_grad_(True)
    enc_sd, _ = onet.split_state_dict(sd)
    y = onet.encoder_forward(enc_sd, f_ref, training=True)
    raw = F.linear(F.linear(y, sd["head.adpit_head.0.weight"], sd["head.adpit_head.0.bias"]),
                   sd["head.adpit_head.1.weight"], sd["head.adpit_head.1.bias"])
    out_ref = torch.tanh(raw)
    loss_ref = ol.adpit_loss(out_ref, target, 12)
    loss_ref.backward()
    assert float((out.detach().cpu() - out_ref).abs().max()) <= 1e-3
    assert abs(float(loss) - float(loss_ref)) <= 1e-3 * abs(float(loss_ref))
    named = dict(model.named_parameters())
    for n in names:
        got, ref = named[n].grad.cpu(), sd[n].grad
        cos = float(torch.dot(got.reshape(-1).double(), ref.reshape(-1).double()) / (got.double().norm() * ref.double().norm()))
        lim = 5e-2 if n.startswith("encoder") else 1e-3          # (toy-size encoder gradients: the bound of DESIGN section 7)
        dev_ = float((got - ref).abs().max()) / float(ref.abs().max())
        assert cos >= 0.999 and dev_ <= lim, "%s: cosine %.6f, max dev %.2e" % (n, cos, dev_)


def test_fused_dropout_residual_equals_the_two_separate_ops(ops):
    """DropoutAxpbyFn (a * dropout(x) + z in one pass; Conformer residual branches, reference resnet_conformer.py:98 over a
    sub-module ending in nn.Dropout) against AxpbyFn(DropoutHashFn(x), z, a, 1): same mask stream, bit-equal forward and
    gradients; a unit factor hands the incoming gradient on without a copy."""
    from adyolo_amd import functional as Fn
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(3, 50, 256, generator=g)
    z0 = torch.randn(3, 50, 256, generator=g)
    dy = dev(torch.randn(3, 50, 256, generator=g))
    for a, p in ((0.5, 0.1), (1.0, 0.2)):
        outs = []
        for fused in (True, False):
            x, z = dev(x0).requires_grad_(True), dev(z0).requires_grad_(True)
            if fused:
                y = Fn.DropoutAxpbyFn.apply(x, z, a, 1.0, p, 1234, 96)
            else:
                y = Fn.AxpbyFn.apply(Fn.DropoutHashFn.apply(x, p, 1234, 96), z, a, 1.0)
            y.backward(dy)
            outs.append((y.detach(), x.grad, z.grad))
        for got, ref, what in zip(outs[0], outs[1], ("forward", "dx", "dz")):
            assert torch.equal(got, ref), (a, p, what)
        keep = (outs[0][1] != 0).float().mean().item()
        assert abs(keep - (1.0 - p)) < 0.02                      # the mask really is a dropout mask
        assert torch.equal(outs[0][2], dy)                        # b = 1: the residual gradient is the incoming one


@pytest.mark.parametrize("n,h,w,cin,cout,kh,kw,sh,sw,ph,pw", [
    (2, 37, 4, 64, 96, 3, 3, 1, 1, 1, 1),        # descriptor fetch of the gathered operand (channels % 32 == 0), ragged M / N tiles
    (3, 50, 1, 128, 64, 3, 1, 1, 1, 1, 0),       # 3 x 1 taps on a width-1 map (the Conformer's narrow stages)
    (2, 21, 6, 32, 32, 1, 1, 1, 1, 0, 0),        # 1 x 1
    (2, 40, 8, 64, 128, 3, 3, 1, 2, 1, 1),       # strided: forward on the fast path, data-gradient on the general one
    (2, 19, 5, 32, 64, 5, 3, 1, 1, 2, 1),        # wide taps: negative tap displacements up to 2 rows
    (1, 33, 16, 8, 64, 7, 7, 1, 2, 3, 3),        # 8 input channels: the general path everywhere
    (2, 16, 4, 48, 40, 3, 3, 1, 1, 1, 1),        # channels not a multiple of the K tile
])
def test_conv_gemm_all_modes_match_torch(ops, n, h, w, cin, cout, kh, kw, sh, sw, ph, pw):
    """adyolo_conv_gemm (implicit GEMM, no column buffer) forward / data-gradient / weight-gradient against torch's convolution
    in float64, over shapes that take the buffer-descriptor fetch (round 4) and shapes that take the general path."""
    g = torch.Generator().manual_seed(n * 100 + h + cin + kh)
    x = torch.randn(n, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, kh, kw, generator=g) / np.sqrt(cin * kh * kw)
    xo, wo = x.double().requires_grad_(True), wt.double().requires_grad_(True)
    yo = F.conv2d(xo, wo, None, stride=(sh, sw), padding=(ph, pw))
    dy = torch.randn(yo.shape, generator=g)
    (yo * dy.double()).sum().backward()
    geom = (n, h, w, cin, cout, kh, kw, sh, sw, ph, pw)
    x8, dy8 = dev(x.permute(0, 2, 3, 1)), dev(dy.permute(0, 2, 3, 1))
    y = ops.conv_gemm(0, x8, ops.pack_wk(dev(wt)), *geom)
    dx = ops.conv_gemm(1, dy8, ops.pack_wk(dev(wt.transpose(0, 1))), *geom)
    dw = ops.unpack_wk(ops.conv_gemm(2, x8, dy8, *geom), cout, cin, kh, kw)
    torch.cuda.synchronize()
    assert_close(y.permute(0, 3, 1, 2), yo.detach().float(), 2e-5, "conv_gemm forward")
    assert_close(dx.permute(0, 3, 1, 2), xo.grad.float(), 2e-5, "conv_gemm data-gradient")
    assert_close(dw, wo.grad.float(), 5e-5, "conv_gemm weight-gradient")
