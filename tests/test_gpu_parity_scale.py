"""GPU parity tests at REALISTIC shapes (run with ``-m gpu`` on an MI355X).

tests/test_gpu_kernels.py compares every kernel with the oracle at sizes the oracle finishes instantly (B <= 7, T <= 150).
This file covers what those sizes cannot see: the reference's own training shape (2, 7, 800, 64) under its default
initialisation (golden from the REAL reference, float32 and float64), the reference's evaluation shape (1, 7, 2400, 64),
and the benchmark shape (64 clips x 60 s: ~1 M convolution patches per launch, XCD block dealing, adaptive
weight-gradient segments) through slice checks against torch on the host.  Everything goes through the C ABI.
"""
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


def _params(nb_classes=12):
    return {"args": {"device": "cuda:0", "encoder": "se-resnet34", "loss": "adyolo"},
            "data_config": {"nb_classes": nb_classes},
            "train_config": {"grid_size": [45, 45], "nb_anchors": 5, "train_unify": [45.0, 25.0, 10.0], "g_overlap": 0.5,
                             "loss_gains": {"angular_gain": 5.0, "object_gain": 1.0, "nonobj_gain": 5.0, "class_gain": 3.0},
                             "optim": "Adam", "lr": 1e-3, "weight_decay": 0.0}}


def strided_sample(numel, n=4096):
    """Same rule as tests/golden/make_golden.py::strided_sample."""
    step = max(1, numel // n)
    return np.arange(0, numel, step)[:n]


def _rel(got, ref):
    return float((got.double() - ref.double()).abs().max()) / max(float(ref.double().abs().max()), 1e-300)


# ------------------------------------------------------------------------------ the reference's training shape, default init
class _ShapeOnly:
    """Stand-in for a ReLU output that was never written (the tail in front of a pooled stage boundary stores avgpool2(e) and
    the mask bits of e, functional.FUSE_POOL): its mask lives in the bits alone."""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def numel(self):
        return int(np.prod(self.shape))


def _align_relu_masks(model, captured, g, bits=None):
    """Give the fragile ReLU elements (golden: float64 pre-activation within 1e-4 x absmax of zero) the float64 mask.
    The build derives every ReLU mask in backward from a saved ReLU OUTPUT (`a > 0`, `e > 0`) -- or, for the SE tail, from
    the mask BITS the forward stored (``bits[site]``) -- so a fragile element that must count as positive becomes 1e-30
    (bit set) and one that must not becomes 0 (bit cleared): a change of at most the activation round-off in value, and
    exactly the reference's subgradient in backward.  Returns the number of masks that had to change."""
    flips = 0
    for site, t in captured.items():
        idx = torch.from_numpy(g["fragile_idx_" + site].astype(np.int64))
        pos = torch.from_numpy(np.unpackbits(g["fragile_pos_" + site])[:idx.numel()].astype(bool))
        n, h, w, c = t.shape                                   # ours: channels-last; golden indices: flat NCHW
        ww, hh, cc, nn_ = idx % w, (idx // w) % h, (idx // (w * h)) % c, idx // (w * h * c)
        flat = ((nn_ * h + hh) * w + ww) * c + cc
        if not isinstance(t, _ShapeOnly):
            flat = flat.to(t.device)
            pos = pos.to(t.device)
            v = t.data.view(-1)
            cur = v[flat]
            flips += int(((cur > 0) != pos).sum())
            v[flat] = torch.where(pos, torch.where(cur > 0, cur, torch.full_like(cur, 1e-30)), torch.zeros_like(cur))
        if bits is not None and site in bits:                  # float4 i = flat >> 2 owns bit (i & 63) of word (i >> 6) * 4 + k
            words = bits[site].cpu().numpy().view(np.uint64)
            f = flat.cpu().numpy().astype(np.uint64)
            i4, k = f >> np.uint64(2), f & np.uint64(3)
            w = ((i4 >> np.uint64(6)) * np.uint64(4) + k).astype(np.int64)
            one = np.uint64(1) << (i4 & np.uint64(63))
            p = pos.cpu().numpy()
            if isinstance(t, _ShapeOnly):                      # no values: the bits ARE the mask
                flips += int((((words[w] & one) != 0) != p).sum())
            np.bitwise_or.at(words, w[p], one[p])
            np.bitwise_and.at(words, w[~p], ~one[~p])
            bits[site].data.copy_(torch.from_numpy(words.view(np.int64)))      # .data: no version bump on a saved tensor
    return flips


# tensors allowed above 5 x the reference's own float32 deviation in the UN-ALIGNED gradient comparison of the seed-100 step, with
# the factor measured for them (see the comment at the comparison): {algorithm: {parameter name: factor}}
UNALIGNED_WHITELIST = {"winograd4": {"encoder.layer1.2.se.fc.0.weight": 5.5}}       # measured 5.13 x (gpurun_out/r06/pytest1.txt)


@pytest.mark.parametrize("algo", ["direct", "winograd", "winograd4"])
def test_seed100_training_step_matches_reference(ops, monkeypatch, algo):
    """One training step at the reference's training shape (2, 7, 800, 64) with the reference's default initialisation
    under torch.manual_seed(100) (src/main.py:47; the build's init is bit-identical, test_host_cpu.py), against
    tests/golden/seed100_train.npz = the REAL reference in float32 and in float64 (resnet.py:180-199, linearheads.py:101-104,
    loss.py:189-251, train.py:49-55).  Both convolution algorithms (Winograd is the default and the benchmarked one)
    meet the same bars:

      * encoder output and logits 1e-3 (north_star), loss 1e-3 relative (1e-4 vs float64), running statistics 1e-4;
      * gradients with the ReLU masks of the fragile elements aligned to the float64 reference: EVERY sampled tensor
        within 1e-4 of its absmax of float64, cosine >= 0.999999 (measured ~1e-5; the reference's own float32 run,
        aligned the same way, gives 0.5-12e-6);
      * gradients as they come (no alignment): cosine >= 0.9999 and max deviation <= max(1e-3, 6 x the reference's own
        float32-vs-float64 deviation), with at most 100 flipped masks.

    Why two gradient checks: hooking the REAL reference (float32 vs float64, this shape) shows the gradient at the
    output of layer4.2 agreeing to 4e-6 (relative L2) and, right behind that block's ReLU, single elements off by 10 % of
    absmax: pre-activations within float32 round-off (5e-6) of zero get the other mask -- a different, equally valid
    subgradient.  23 such flips (of 44 M ReLU elements) move the reference's own gradients by 2e-3 .. 9e-2 of absmax
    depending on which elements they hit (oneDNN vs native convolution back end); with the 23 masks aligned the same
    float32 run agrees with float64 to 1.2e-5.  Which elements flip is a property of each implementation's rounding, so
    the un-aligned comparison can only be statistical; the aligned one is exact arithmetic parity."""
    monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    g = np.load(os.path.join(G, "seed100_train.npz"))
    prm = _params()
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 800, 64), (), prm).to("cuda:0")
    model.train()
    model.encoder.lstm.dropout = 0.0
    x = torch.randn(2, 7, 800, 64, generator=torch.Generator().manual_seed(int(g["x_seed"])))
    assert abs(float(x.double().sum()) - float(g["x_sum"])) < 1e-6 and np.array_equal(x.reshape(-1)[:16].numpy(), g["x_head"])
    target = torch.from_numpy(g["target"])
    captured, bits, hooks = {}, {}, []

    from adyolo_amd import functional as Fn

    def grab(site_a, site_e, is_first):
        def hook(mod, inp, out):
            sv = Fn.saved(out.grad_fn)                         # SEBlockFn's saved tensors by name
            if sv["e"] is None:                                # the block output is avgpool2(e) (functional.FUSE_POOL): bits only
                assert "ebits" in sv and out.shape[1] * 2 == sv["cc"].shape[1]
                captured[site_a], captured[site_e] = sv["a"], _ShapeOnly(sv["cc"].shape)
            else:
                assert sv["e"].data_ptr() == out.data_ptr()
                captured[site_a], captured[site_e] = sv["a"], out  # a = relu(conv1(x)), e = the block output
            if "ebits" in sv:
                bits[site_e] = sv["ebits"]                     # (e > 0) as bits, read by the SE-tail backward
            if is_first:
                captured["stem"] = Fn.saved(inp[0].grad_fn)["a"]      # StemFn: a = relu(conv + bias)
        return hook
    for li in range(1, 5):
        for bi, blk in enumerate(getattr(model.encoder, "layer%d" % li)):
            hooks.append(blk.register_forward_hook(grab("layer%d.%d.a" % (li, bi), "layer%d.%d.e" % (li, bi), li == 1 and bi == 0)))
    y = model.encoder(x.to("cuda:0"))
    logit = model.head(y)
    loss = WrapperCriterion(prm)(logit, target)
    for h in hooks:
        h.remove()
    assert len(captured) == 33 and len(bits) == 16
    torch.cuda.synchronize()
    y_ref = torch.from_numpy(g["y_train"])
    assert float((y.detach().cpu() - y_ref).abs().max()) <= 1e-3, "encoder output (tanh range) vs reference"
    li = strided_sample(logit.numel(), 65536)
    lg = logit.detach().cpu().reshape(-1)[li]
    lref = torch.from_numpy(g["logit_sample"])
    assert float((lg - lref).abs().max()) <= 1e-3 * max(1.0, float(lref.abs().max())), "logits vs reference"
    lv = float(loss.detach())
    assert abs(lv - float(g["loss"][0])) <= 1e-3 * abs(float(g["loss"][0])), (lv, float(g["loss"][0]))
    assert abs(lv - float(g["loss64"][0])) <= 1e-4 * abs(float(g["loss64"][0]))
    sd = model.state_dict()
    for key in g.files:
        if key.startswith("stat_"):
            ref = torch.from_numpy(g[key])
            assert float((sd[key[5:]].cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), key

    named = dict(model.named_parameters())

    def compare(tag, max_limit, cos_limit, whitelist=None):
        bad, report = [], []
        for key in g.files:
            if not key.startswith("grad64_"):
                continue
            name = key[len("grad64_"):]
            t64 = torch.from_numpy(g[key])
            idx = strided_sample(named[name].numel())
            got = named[name].grad.reshape(-1).cpu()[idx].double()
            ref_noise = float(g["gnoise_" + name])
            mine = float((got - t64).abs().max()) / float(g["gabs64_" + name])
            cos = float(torch.dot(got, t64) / (got.norm() * t64.norm()))
            limit = max_limit(ref_noise)
            if whitelist and name in whitelist:
                limit = max(limit, whitelist[name] * ref_noise)
            report.append("%-42s %.2e (reference float32, un-aligned: %.2e = %.2f x) cos %.8f" % (name, mine, ref_noise, mine / max(ref_noise, 1e-30), cos))
            if mine > limit or cos < cos_limit:
                bad.append("%s: %.2e of absmax vs float64 (limit %.2e), cosine %.8f" % (name, mine, limit, cos))
        print("[%s, %s]\n" % (algo, tag) + "\n".join(report))
        assert not bad, "%s gradients: " % tag + "; ".join(bad)

    loss.backward(retain_graph=True)
    torch.cuda.synchronize()
    # 5 x the reference's own float32 deviation for every tensor.  ONE tensor is whitelisted BY NAME with its measured ratio
    # (round 5 ADVICE: the bound had been widened to 6 x for everybody): since stage 1 runs on the F(4x4) kernels an SE weight whose
    # reference deviation happens to be the smallest of its group (1.9e-3) sits at 5.1 x under winograd4.  That this is ReLU-mask
    # flips and not kernel rounding is what the SECOND comparison shows: with the fragile masks aligned to the float64 run the
    # same tensor, like every other, is within 1e-4 of absmax.
    compare("as they come", lambda ref_noise: max(1e-3, 5.0 * ref_noise), 0.9999, whitelist=UNALIGNED_WHITELIST.get(algo))
    for p in model.parameters():
        p.grad = None
    flips = _align_relu_masks(model, captured, g, bits)
    loss.backward()
    torch.cuda.synchronize()
    print("[%s] %d of the %d fragile ReLU masks differed from float64" % (algo, flips, sum(v.numel() for v in captured.values()) and
          sum(int(g["fragile_idx_" + s_].shape[0]) for s_ in captured)))
    assert flips <= 100
    compare("masks aligned", lambda ref_noise: 1e-4, 0.999999)


def test_training_trajectories_direct_vs_winograd(ops, monkeypatch):
    """40 Adam steps on a fixed synthetic batch (8 x 10 s raw audio, K1 included) with the direct and the Winograd
    convolutions: identical loss at step 0 (1e-5), both decreasing, and never further apart than round-off alone drives
    two runs of the SAME algorithm: the yardstick is the direct path re-run with the BatchNorm-backward sums taken by the
    separate reduction pass instead of the dgrad epilogue (same arithmetic, other summation order); the bound is 3x that
    gap with a floor of 2e-3 (reference loop: src/train.py:40-62)."""
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep

    import adyolo_amd.functional as Fn

    def run(algo, steps=40, fuse_bnbwd=True):
        monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
        monkeypatch.setattr(Fn, "FUSE_BNBWD", fuse_bnbwd)
        torch.manual_seed(100)
        prm = bench.params("cuda:0")
        b, n = 8, 24000 * 10
        t = n // 600
        model = WrapperModel((1, 7, t, 64), (), prm).to("cuda:0")
        model.encoder.lstm.dropout = 0.0
        tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
        audio = synthetic_audio(b, n, seed=7).to("cuda:0")
        target = synthetic_targets(b, t // 4, 12, seed=7).to("cuda:0")
        losses = torch.stack([tr.step(audio, target).reshape(()) for _ in range(steps)])
        return losses.cpu().double().numpy()

    a, a2 = run("direct"), run("direct", fuse_bnbwd=False)
    noise = np.abs(a - a2) / np.abs(a)
    assert abs(a[0] - a2[0]) <= 1e-6 * abs(a[0])
    for algo in ("winograd", "winograd4"):          # F(2x2,3x3) everywhere; F(4x4,3x3) from 128 channels on (round 4)
        w = run(algo)
        assert np.all(np.isfinite(a)) and np.all(np.isfinite(w))
        assert abs(a[0] - w[0]) <= 1e-5 * abs(a[0])
        rel = np.abs(a - w) / np.abs(a)
        print("direct %.5f -> %.5f, %s %.5f -> %.5f, worst relative gap %.2e (same-algorithm round-off yardstick %.2e)"
              % (a[0], a[-1], algo, w[0], w[-1], rel.max(), noise.max()))
        assert rel.max() <= max(2e-3, 3 * noise.max()), (algo, rel, noise)
        assert a[-1] < 0.7 * a[0] and w[-1] < 0.7 * w[0]


# ------------------------------------------------------------------------------ the reference's evaluation shape
def test_eval_forward_at_reference_test_shape(ops):
    """Whole-clip evaluation forward, B = 1, T = 2400 (60 s; src/test.py:81 feeds one file at a time), default (Winograd)
    convolutions, against the float32 CPU oracle on the same seed-100 weights: encoder output and logits 1e-3."""
    from adyolo_amd.wrapper import WrapperModel
    from oracle import seresnet as onet
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 2400, 64), (), _params()).to("cuda:0")
    # non-trivial running statistics (fresh BatchNorm buffers are 0 / 1)
    gen = torch.Generator().manual_seed(3)
    for k, v in model.state_dict().items():
        if k.endswith("running_mean"):
            v.copy_((torch.rand(v.shape, generator=gen) * 0.2 - 0.1).to(v.device))
        elif k.endswith("running_var"):
            v.copy_((torch.rand(v.shape, generator=gen) * 0.5 + 0.75).to(v.device))
    model.eval()
    x = torch.randn(1, 7, 2400, 64, generator=gen)
    with torch.no_grad():
        y = model.encoder(x.to("cuda:0"))
        logit = model.head(y)
    torch.cuda.synchronize()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    enc, head = onet.split_state_dict(sd)
    with torch.no_grad():
        y_ref = onet.encoder_forward(enc, x, training=False)
        l_ref = onet.adyolo_head(head, y_ref)
    assert y.shape == (1, 600, 256) and logit.shape == (1, 600, 2400)
    e_y = float((y.cpu() - y_ref).abs().max())
    e_l = float((logit.cpu() - l_ref).abs().max())
    print("T=2400 eval: encoder err %.2e, logit err %.2e (absmax %.2f)" % (e_y, e_l, float(l_ref.abs().max())))
    assert e_y <= 1e-3 and e_l <= 1e-3 * max(1.0, float(l_ref.abs().max()))


def test_conformer_eval_forward_at_training_shape(ops):
    """ResNet-Conformer (config 4) at its training length, B = 1, T = 800 (20 s): the implicit-GEMM convolutions (7x7 s(1,2)
    stem, strided first blocks, the W = 4 / 2 / 1 stride-1 maps with the folded W = 2 form), the Winograd ones (W = 8) and
    the flash-style attention at T = 800 against the float32 CPU oracle (oracle/conformer.py, itself pinned by the
    reference golden at T = 32) with random default-init weights: encoder output 1e-3."""
    from adyolo_amd.wrapper import WrapperModel
    from oracle import conformer as ocf
    prm = _params()
    prm["args"]["encoder"] = "resnet-conformer"
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 800, 64), (), prm).to("cuda:0")
    gen = torch.Generator().manual_seed(4)
    for k, v in model.state_dict().items():
        if k.endswith("running_mean"):
            v.copy_((torch.rand(v.shape, generator=gen) * 0.2 - 0.1).to(v.device))
        elif k.endswith("running_var"):
            v.copy_((torch.rand(v.shape, generator=gen) * 0.5 + 0.75).to(v.device))
    model.eval()
    x = torch.randn(1, 7, 800, 64, generator=gen)
    with torch.no_grad():
        y = model.encoder(x.to("cuda:0"))
    torch.cuda.synchronize()
    sd = {k[len("encoder."):]: v.detach().cpu() for k, v in model.state_dict().items() if k.startswith("encoder.")}
    with torch.no_grad():
        y_ref = ocf.encoder_forward(sd, x, training=False)
    assert y.shape == (1, 200, 256)
    err = float((y.cpu() - y_ref).abs().max())
    print("conformer T=800 eval: encoder err %.2e (absmax %.2f)" % (err, float(y_ref.abs().max())))
    assert err <= 1e-3 * max(1.0, float(y_ref.abs().max()))
    # training mode (batch statistics through 36 BatchNorms + 8 conformer blocks), dropout switched off on both sides
    enc = model.encoder
    enc.train()
    for m in enc.modules():
        if hasattr(m, "p") and isinstance(getattr(m, "p"), float):
            m.p = 0.0
        if hasattr(m, "p2"):
            m.p2 = 0.0
    with torch.no_grad():
        yt = enc(x.to("cuda:0"))
        yt_ref = ocf.encoder_forward(sd, x, training=True)
    err = float((yt.cpu() - yt_ref).abs().max())
    print("conformer T=800 train-mode forward: encoder err %.2e" % err)
    assert err <= 1e-3 * max(1.0, float(yt_ref.abs().max()))


# ------------------------------------------------------------------------------ benchmark shape: convolution slice checks
BENCH_STAGES = [  # (Cin, Cout, H, W) of the 3x3 convolutions of stages 1-4 at 64 clips x 60 s
    (32, 32, 2400, 64), (64, 64, 1200, 32), (128, 128, 600, 16), (256, 256, 600, 16), (32, 64, 1200, 32)]


@pytest.mark.parametrize("cin,cout,h,w", BENCH_STAGES)
@pytest.mark.parametrize("algo", ["winograd4", "winograd", "direct"])
def test_conv3x3_at_bench_shape_slices(ops, monkeypatch, algo, cin, cout, h, w):
    """conv3x3 forward, data-gradient and weight-gradient launched at the FULL benchmark shape (N = 64 clips); forward and
    data-gradient are compared on two clips x 64 rows (first clip/top rows incl. the zero padding, last clip/an interior
    window crossing patch boundaries) with F.conv2d on the slice (2e-5 of absmax), the weight-gradient on an 8 x 8
    (Cout, Cin) block against a float64 contraction over ALL N*H*W pixels (5e-5 of absmax: 9.8 M-term fp32 sums)."""
    monkeypatch.setenv("ADYOLO_W4_MIN_K", "32")      # winograd4: the F(4x4) kernel at every stage it can run (Cout % 64 == 0)
    if algo == "winograd4" and cout % 64:
        pytest.skip("the F(4x4) kernel needs 64 output channels per workgroup: this stage runs the F(2x2) kernel")
    n = 64
    gen = torch.Generator(device="cuda:0").manual_seed(cin * 7 + cout)
    x = torch.randn(n, h, w, cin, generator=gen, device="cuda:0")
    dy = torch.randn(n, h, w, cout, generator=gen, device="cuda:0")
    wt = (torch.randn(cout, cin, 3, 3, generator=gen, device="cuda:0") / np.sqrt(9 * cin)).contiguous()
    wf, wd = ops.pack_w3x3(wt, cin, algo=algo)
    y = ops.conv3x3(x, wf, cout)
    dx = ops.conv3x3(dy, wd, cin)
    dw = ops.conv3x3_wgrad(x, dy, cin, algo=algo)
    torch.cuda.synchronize()
    wc = wt.cpu()
    wflip = wc.flip(2, 3).permute(1, 0, 2, 3).contiguous()            # data-gradient = convolution with flipped, transposed taps
    for clip, r0 in ((0, 0), (n - 1, h // 2 - 29)):
        lo, hi = max(r0 - 1, 0), min(r0 + 65, h)                      # one halo row each side where the image has one
        for src, kern, out, what in ((x, wc, y, "forward"), (dy, wflip, dx, "data-gradient")):
            xin = src[clip, lo:hi].cpu().permute(2, 0, 1)[None]       # (1, C, rows, W)
            ref = F.conv2d(xin.double(), kern.double(), padding=1)[0].permute(1, 2, 0)
            top = r0 - lo
            ref = ref[top:top + 64]
            # (the slice's own zero padding is only right at image borders: interior windows carry a halo row, dropped above)
            got = out[clip, r0:r0 + 64].cpu()
            e = _rel(got, ref)
            assert e <= 2e-5, "%s %s clip %d rows %d..%d: %.2e of absmax" % (algo, what, clip, r0, r0 + 64, e)
    # weight gradient block [co0:co0+8, ci0:ci0+8] over every pixel of the batch, float64 on the host
    co0, ci0 = cout - 8, cin // 2
    ref = torch.zeros(8, 8, 3, 3, dtype=torch.float64)
    for b0 in range(0, n, 8):
        xs = x[b0:b0 + 8, :, :, ci0:ci0 + 8].cpu().double()
        ds = dy[b0:b0 + 8, :, :, co0:co0 + 8].cpu().double()
        xp = F.pad(xs, (0, 0, 1, 1, 1, 1))
        for kh in range(3):
            for kw in range(3):
                ref[:, :, kh, kw] += torch.einsum("nhwo,nhwi->oi", ds, xp[:, kh:kh + h, kw:kw + w, :])
    e = _rel(dw[co0:co0 + 8, ci0:ci0 + 8].cpu(), ref)
    assert e <= 5e-5, "%s weight-gradient block: %.2e of absmax" % (algo, e)


def test_stem_conv_at_bench_shape_slices(ops):
    """The 8-channel stem (direct kernel, bias + ReLU + per-patch BatchNorm sums) at N = 64, H = 2400, W = 64."""
    n, h, w = 64, 2400, 64
    gen = torch.Generator(device="cuda:0").manual_seed(5)
    x = torch.randn(n, h, w, 8, generator=gen, device="cuda:0")
    x[..., 7] = 0.0
    wt = (torch.randn(32, 7, 3, 3, generator=gen, device="cuda:0") / np.sqrt(63)).contiguous()
    b = torch.randn(32, generator=gen, device="cuda:0")
    wpk, _ = ops.pack_w3x3(wt, 8, want_dgrad=False)
    y, st = ops.conv3x3(x, wpk, 32, bias=b, relu=True, want_stats=True)
    torch.cuda.synchronize()
    for clip, r0 in ((0, 0), (n - 1, h - 64)):
        lo, hi = max(r0 - 1, 0), min(r0 + 65, h)
        xin = x[clip, lo:hi, :, :7].cpu().permute(2, 0, 1)[None].double()
        ref = F.relu(F.conv2d(xin, wt.cpu().double(), b.cpu().double(), padding=1))[0].permute(1, 2, 0)
        ref = ref[r0 - lo:r0 - lo + 64]
        assert _rel(y[clip, r0:r0 + 64].cpu(), ref) <= 2e-5
    # per-patch sums add up to the channel sums of the whole output (the BatchNorm statistics come from them)
    tot = st[0].double().sum(0).cpu()
    ref = y.double().sum(dim=(0, 1, 2)).cpu()
    assert float((tot - ref).abs().max()) <= 1e-6 * float(ref.abs().max())
    # weight gradient at full size (stem kernel: rows dealt to 512 workgroups): with dy non-zero only in three row bands
    # (first rows, last rows, a band across a sample boundary) it must equal the float64 weight gradient of those bands
    dy = torch.zeros(n, h, w, 32, device="cuda:0")
    bands = ((0, 0, 40), (n - 1, h - 40, h), (17, h - 3, h), (18, 0, 5))
    ref_dw = torch.zeros(32, 7, 3, 3, dtype=torch.float64)
    for clip, r0, r1 in bands:
        dy[clip, r0:r1] = torch.randn(r1 - r0, w, 32, generator=gen, device="cuda:0")
        lo, hi = max(r0 - 1, 0), min(r1 + 1, h)
        xin = x[clip, lo:hi, :, :7].cpu().permute(2, 0, 1)[None].double()
        wd_ = wt.cpu().double().requires_grad_(True)
        out = F.conv2d(xin, wd_, None, padding=1)
        gsl = torch.zeros_like(out)
        gsl[0, :, r0 - lo:r1 - lo] = dy[clip, r0:r1].cpu().double().permute(2, 0, 1)
        out.backward(gsl)
        ref_dw += wd_.grad
    dw = ops.conv3x3_wgrad(x, dy, 7)
    assert _rel(dw.cpu(), ref_dw) <= 2e-5


# ------------------------------------------------------------------------------ benchmark shape: loss
def test_adyolo_loss_at_bench_shape(ops):
    """AD-YOLO loss + dlogit at B = 64, T' = 600 (M ~ 137 k rows, 683 k (target, anchor) pairs, 92 M logits) against the CPU
    oracle (loss.py:189-251): loss 1e-5 relative, dlogit 1e-3 of absmax.

    With 683 k pairs the discrete decisions of the loss meet float32 round-off, in the reference as here; the anchors they
    touch are compared loosely (bounded by the largest regular entry) and must be rare (< 0.5 % of the anchors):
      * D within 1e-3 degree of a threshold 45 / 25 / 10 (loss.py:225 `D < thr`): the pair is positive or not;
      * the two closest anchors of a target within 1e-3 degree (loss.py:226 arg-min; typical at the poles, where the clamped
        elevation makes the azimuth irrelevant): a different anchor is forced positive;
      * D within 0.5 degree of 0 or 180 (loss.py:187 clips the acos argument at +-(1 - 1e-7)): 1 - |cos D| < 4e-5 carries
        round-off of its own order, so the direction of dD/d(u, v) is decided by rounding."""
    from adyolo_amd.datasets import synthetic_targets
    from oracle import adyolo_loss as oloss
    b, t, a, ch = 64, 600, 5, 15
    gen = torch.Generator().manual_seed(17)
    logit = torch.randn(b, t, 2400, generator=gen) * 1.5
    target = synthetic_targets(b, t, 12, seed=17)
    loss, dlogit, _ = ops.adyolo_loss(logit.to("cuda:0"), target.to("cuda:0"), 12)
    torch.cuda.synchronize()
    lo = logit.clone().requires_grad_(True)
    ref, aux = oloss.adyolo_loss(lo, target, 12, return_aux=True)
    ref.backward()
    print("M = %d rows; loss %.6f vs oracle %.6f" % (target.shape[0], float(loss), float(ref)))
    assert abs(float(loss) - float(ref)) <= 1e-5 * abs(float(ref))
    g = lo.grad.view(-1, ch)                                            # [anchor][obj, cls x 12, u, v]
    got = dlogit.cpu().view(-1, ch)
    d = aux["D"]                                                        # (M, A) degrees
    tb, tt, gi, gj = (target[:, k].long() for k in range(4))
    cell = ((tb * t + tt) * 8 + gi) * 4 + gj
    anchors = cell[:, None] * a + torch.arange(a)[None, :]             # (M, A) anchor rows of every pair
    near_thr = ((d - 45.0).abs() < 1e-3) | ((d - 25.0).abs() < 1e-3) | ((d - 10.0).abs() < 1e-3)
    srt, _ = d.sort(dim=1)
    tie = ((srt[:, 1] - srt[:, 0]) < 1e-3)[:, None].expand_as(d)
    singular = (d < 0.5) | (d > 179.5)
    fragile = torch.zeros(g.shape[0], dtype=torch.bool)
    fragile[anchors[near_thr | tie | singular]] = True
    n_frag = int(fragile.sum())
    assert n_frag <= 5e-3 * g.shape[0], n_frag
    am = float(g[~fragile].abs().max())
    err = float((got[~fragile] - g[~fragile]).abs().max())
    print("%d fragile anchors of %d (threshold %d, arg-min tie %d, singular %d pairs); regular entries: %.2e of absmax %.2e"
          % (n_frag, g.shape[0], int(near_thr.sum()), int(tie[:, 0].sum()), int(singular.sum()), err / am, am))
    assert err <= 1e-3 * am
    assert float(got[fragile].abs().max()) <= 1.5 * float(g.abs().max())


def test_bench_shape_step_loss_direct_vs_winograd(ops, monkeypatch):
    """First training-step loss at the benchmark workload (64 clips x 60 s raw audio, K1 -> encoder -> head -> loss) with
    the Winograd (default, benchmarked) and the direct convolutions: 1e-3 relative (measured ~1e-6)."""
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    b, n = 64, 24000 * 60
    t = n // 600
    audio = synthetic_audio(b, n, seed=1234).to("cuda:0")
    target = synthetic_targets(b, t // 4, 12, seed=1234).to("cuda:0")
    fx = FeatureExtractor(None, "cuda:0")
    vals = {}
    for algo in ("winograd4", "winograd", "direct"):
        monkeypatch.setenv("ADYOLO_CONV_ALGO", algo)
        torch.manual_seed(100)
        prm = bench.params("cuda:0")
        model = WrapperModel((1, 7, t, 64), (), prm).to("cuda:0")
        model.train()
        with torch.no_grad():
            out = model(fx(audio, channels_last8=True), channels_last8=True)
            vals[algo] = float(WrapperCriterion(prm)(out, target))
        del model, out
        torch.cuda.empty_cache()
    print(vals)
    for algo in ("winograd4", "winograd"):
        assert np.isfinite(vals[algo]) and abs(vals[algo] - vals["direct"]) <= 1e-3 * abs(vals["direct"]), (algo, vals)


def test_dispatch_table_at_the_bench_shape(ops):
    """Which kernel every 3x3 convolution of one training step launches at the benchmark's clip shape with the DEFAULT thresholds
    (16 clips x 60 s: every layer takes the kernel the 64-clip batch takes -- ``ops.DualPack`` picks by launch size: the smallest
    F(4x4) forward launch of the step, the 128 -> 64 data-gradient at 600 x 16 pixels, needs 11 clips for its 200 work items, the
    smallest weight gradients (32 -> 64, 64 -> 128) 14 clips for ADYOLO_W4W_MIN_WORK).  A threshold or dispatch regression cannot hide behind green parity tests: the parity tests force
    ADYOLO_W4_MIN_K=32, this one asserts the table bench.py reports as ``dispatch``.
    SE-ResNet34 (reference resnet.py:126-199): 16 blocks x 2 convolutions, forward + data-gradient = 64 launches + the 7 -> 32 stem.
    F(4x4,3x3) in its persistent form takes ALL 64 block launches (round 5: also stage 1's 32 -> 32 layers, with 32-channel output
    blocks; round 6: the 32 -> 64 stage transition -- ADYOLO_W4_MIN_K 64 -> 32 -- and the first block's data-gradient, operand
    combination 15); the stem is the direct kernel.  The weight gradients of all 32 block convolutions run in the F(4x4) domain
    (csrc/wino4w.hip)."""
    import bench
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from adyolo_amd.train import TrainStep
    assert ops.reload_thresholds() == {"min_k": 32, "min_k_addend": 32, "min_wgs": 200, "min_wgrad_work": 6000000, "min_k_32": 32}
    sw = ops.switch_table()
    assert sw["conv_algo"] == "winograd4" and sw["persist"] and sw["narrow"] and sw["wino1d"] and sw["wgrad_algo"] is None
    b, n = 16, 24000 * 60
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
    audio = synthetic_audio(b, n, seed=1234).to("cuda:0")
    target = synthetic_targets(b, n // 2400, 12, seed=1234).to("cuda:0")
    tr.step(audio, target)
    ops.DISPATCH_LOG = {}
    try:
        loss = float(tr.step(audio, target))
        torch.cuda.synchronize()
    finally:
        log, ops.DISPATCH_LOG = ops.DISPATCH_LOG, None
    assert np.isfinite(loss)
    per_kernel, per_wgrad = {}, {}
    for (name, cin, cout, epi), cnt in list(log.items()):
        if "wgrad" in name:                        # the weight-gradient launches are logged beside the forward / data-gradient ones
            per_wgrad[name] = per_wgrad.get(name, 0) + cnt
            del log[(name, cin, cout, epi)]
        else:
            per_kernel[name] = per_kernel.get(name, 0) + cnt
    print(sorted(log.items()), per_wgrad)
    assert per_kernel == {"wino4p_fwd_kernel": 64, "conv3x3_fwd_kernel": 1}, per_kernel
    # (until round 6 F(2x2) kept two launches: the 32 -> 64 forward -- a 32-channel contraction was below ADYOLO_W4_MIN_K -- and
    #  the data-gradient of the very first block: addend + mask + statistics against the stem's BatchNorm input, no statistics
    #  mask = operand combination 15, which the persistent kernel was not built for)
    assert not [k for k in log if k[0] == "wino_fwd_kernel"]
    # the weight gradients: the F(4x4) domain for every block convolution (a launch needs ADYOLO_W4W_MIN_WORK: 16 clips have it,
    # and so has the benchmark's batch -- asked from the one function that decides)
    assert per_wgrad == {"wino4_wgrad_kernel": 32, "conv3x3_wgrad_kernel": 1}, per_wgrad
    layers = [(32, 32, 2400, 64)] * 6 + [(32, 64, 1200, 32)] + [(64, 64, 1200, 32)] * 7 + [(64, 128, 600, 16)] + \
             [(128, 128, 600, 16)] * 11 + [(128, 256, 600, 16)] + [(256, 256, 600, 16)] * 5
    assert len(layers) == 32
    assert all(ops.wgrad_form(ci, co, None, (64, h, w)) == ("wino4_wgrad_kernel", 9.0 / 36.0) for ci, co, h, w in layers)
    assert ops.wgrad_form(8, 32, None, (64, 2400, 64))[0] == "conv3x3_wgrad_kernel"
    assert [(cin, cout) for (name, cin, cout, _), _ in log.items() if name == "conv3x3_fwd_kernel"] == [(8, 32)]
    # operand combinations of the persistent launches: forward (statistics), the three data-gradient forms of conv1 / conv2
    assert {epi for (name, _, _, epi), _ in log.items() if name == "wino4p_fwd_kernel"} == {1, 2, 9, 15, 27, 31}


def test_bench_timing_events_measure_what_torch_events_measure(ops):
    """``bench.TimingEvent`` (HIP events created with hipEventDisableSystemFence through the runtime PyTorch has loaded; round 6)
    around the same launches as ``torch.cuda.Event``: the two clocks agree (a 3x3 convolution of ~1 ms: within 10 %), the events
    are recorded on the current stream, and ``make_event`` reports which kind it handed out."""
    import bench
    x = torch.randn(16, 600, 16, 128, device="cuda:0")
    wt = torch.randn(128, 128, 3, 3, device="cuda:0") * 0.05
    wpk, _ = ops.pack_w3x3(wt, 128, want_dgrad=False)
    for _ in range(3):
        ops.conv3x3(x, wpk, 128)
    torch.cuda.synchronize()
    a0, a1 = bench.make_event(torch, "timing"), bench.make_event(torch, "timing")
    assert bench.EVENT_KIND_USED["fallback"] is None and bench.EVENT_KIND_USED["kind"] == "hipEventDisableSystemFence"
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    a0.record()
    for _ in range(8):
        ops.conv3x3(x, wpk, 128)
    a1.record()
    t1.record()
    torch.cuda.synchronize()
    ms_a, ms_t = a0.elapsed_time(a1), t0.elapsed_time(t1)
    assert ms_a > 0.2 and abs(ms_a - ms_t) <= 0.1 * ms_t, (ms_a, ms_t)
    assert isinstance(bench.make_event(torch, "torch"), torch.cuda.Event)


# ------------------------------------------------------------------------------ data-parallel path on one GPU (RCCL, 1 rank)
_DP_CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, os.environ["ADYOLO_REPO"])
import torch
import adyolo_amd
import bench
from adyolo_amd import dist as adist
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep
rank, world, local = adist.init_from_env("nccl")
torch.manual_seed(100)
prm = bench.params("cuda:0")
b, n = 4, 24000 * 4
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)
audio = synthetic_audio(b, n, seed=3).to("cuda:0")
target = synthetic_targets(b, n // 2400, 12, seed=3).to("cuda:0")
fired = []
losses = []
for _ in range(3):
    h0, f0 = tr.reducer.fired_from_hooks, tr.reducer.fired_from_finish
    losses.append(float(tr.step(audio, target)))
    fired.append((tr.reducer.fired_from_hooks - h0, tr.reducer.fired_from_finish - f0))
torch.cuda.synchronize()
digest = hashlib.sha256(tr.flat.flat.cpu().numpy().tobytes()).hexdigest()
print(json.dumps({"digest": digest, "fired": fired, "buckets": len(tr.reducer.buckets), "active": tr.reducer.active,
                  "losses": losses, "dist": torch.distributed.is_initialized()}))
if torch.distributed.is_initialized():
    torch.distributed.destroy_process_group()
"""


def test_bucketed_allreduce_hooks_on_the_real_model(ops):
    """The N > 1 code path on ONE GPU: a FRESH child process with WORLD_SIZE=1 ADYOLO_FORCE_DP_HOOKS=1 initialises RCCL
    (backend nccl), registers the post-accumulate-grad hooks on the real SE-ResNet34 + AD-YOLO model and runs 3 TrainSteps:
    all 4 buckets must be launched from hooks (overlapped with backward; none left to finish()'s straggler path) and the
    parameters must be BIT-equal to a child without the hooks (the step is bit-reproducible, see test_raw_audio_epoch)."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]

    def child(force):
        env = dict(os.environ, ADYOLO_REPO=repo, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("ADYOLO_FORCE_DP_HOOKS", None)
        if force:
            env["ADYOLO_FORCE_DP_HOOKS"] = "1"
        r = subprocess.run([sys.executable, "-c", _DP_CHILD], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"digest"')]
        assert lines, "child printed no result line: %r / %r" % (r.stdout[-1000:], r.stderr[-1000:])
        return json.loads(lines[-1])
    hooked, plain = child(True), child(False)
    assert hooked["dist"] and hooked["active"] and not plain["active"]
    assert hooked["buckets"] == 4 and all(f == [4, 0] for f in hooked["fired"]), hooked["fired"]
    assert hooked["losses"] == plain["losses"] and hooked["digest"] == plain["digest"]


def test_grad_sink_equals_autograd_accumulation(ops):
    """TrainStep lets the convolution / BatchNorm / SE gradient kernels write straight into the flat gradient buffer
    (functional.GradSink) instead of returning tensors that autograd adds into it: both routes must give the same bits, and
    every parameter must have received its gradient exactly once."""
    import bench
    from adyolo_amd import functional as Fn
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.dist import FlatParameters
    from adyolo_amd.datasets import synthetic_targets
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, 160, 64), (), prm).to("cuda:0")
    model.train()
    model.encoder.lstm.dropout = 0.0
    flat = FlatParameters(model)
    crit = WrapperCriterion(prm)
    x = torch.randn(3, 7, 160, 64, generator=torch.Generator().manual_seed(4)).to("cuda:0")
    target = synthetic_targets(3, 40, 12, seed=4)
    grads = []
    for use_sink in (False, True):
        flat.zero_grad()
        loss = crit(model(x), target)
        if use_sink:
            Fn.SINK.begin(flat)
        try:
            loss.backward()
        finally:
            Fn.SINK.end()
        torch.cuda.synchronize()
        grads.append(flat.flat_grad.clone())
    assert torch.equal(grads[0], grads[1])
    # (a gradient may legitimately be all zero -- an SE bottleneck whose ReLU is dead for the whole batch -- but not many)
    nonzero = sum(float(grads[1][off:off + n].abs().sum()) > 0 for off, n in flat.offsets)
    assert nonzero >= 0.95 * len(flat.offsets), (nonzero, len(flat.offsets))


@pytest.mark.parametrize("training", [True, False])
def test_pooled_stage_boundaries_equal_the_separate_calls(ops, monkeypatch, training):
    """functional.FUSE_POOL / FUSE_POOL_BWD (round 6): in front of a pooled stage boundary (reference resnet.py:29,40,158-164:
    the next SEBasicBlock starts with AvgPool2d(2, 2)) the tail writes avgpool2(e) and the mask bits instead of e, and its backward
    works from the pooled gradient.  Output, loss and EVERY parameter gradient of the whole model must equal the unfused route
    bit for bit, in training mode (with and without the backward half) and in evaluation mode."""
    import bench
    from adyolo_amd import functional as Fn
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.dist import FlatParameters
    from adyolo_amd.datasets import synthetic_targets
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, 160, 64), (), prm).to("cuda:0")
    model.train(training)
    model.encoder.lstm.dropout = 0.0
    flat = FlatParameters(model)
    crit = WrapperCriterion(prm)
    x = torch.randn(3, 7, 160, 64, generator=torch.Generator().manual_seed(4)).to("cuda:0")
    target = synthetic_targets(3, 40, 12, seed=4)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    seen = []
    real = ops.se_tail_fwd

    def spy(*a, **kw):
        seen.append(kw.get("pool_hw"))
        return real(*a, **kw)
    monkeypatch.setattr(ops, "se_tail_fwd", spy)
    runs = []
    for fuse, fuse_bwd in ((False, False), (True, False), (True, True)):
        monkeypatch.setattr(Fn, "FUSE_POOL", fuse)
        monkeypatch.setattr(Fn, "FUSE_POOL_BWD", fuse_bwd)
        model.load_state_dict(sd0)                       # (training mode moves the running statistics)
        del seen[:]
        if training:
            flat.zero_grad()
            out = model(x)
            loss = crit(out, target)
            loss.backward()
            torch.cuda.synchronize()
            runs.append((out.detach().clone(), loss.detach().clone(), flat.flat_grad.clone()))
        else:
            with torch.no_grad():
                out = model(x)
            torch.cuda.synchronize()
            runs.append((out.clone(),))
        pooled = [hw for hw in seen if hw is not None]
        assert len(seen) == 16 and len(pooled) == (2 if fuse else 0), (fuse, seen)
    for other in runs[1:]:
        for a, b in zip(runs[0], other):
            assert torch.equal(a, b)


# ------------------------------------------------------------------------------ BASELINE configs 3 and 5 on the real model
def test_config3_dcase2022_c13_train_step_matches_oracle(ops):
    """BASELINE config 3's model on one device: DCASE2022 = 13 classes (src/configs/hyp_data_DCASE2022.yaml:3) -> head
    256 -> 8*4*5*16 = 2560 logits, C = 13 loss and label rows, features z-scored with the shipped DCASE2022 scaler
    (tests/golden/scaler_DCASE2022.npz = the reference's data/DCASE2022_SELD/scaler_wts.pkl).  One training step from raw
    int16-range audio: features, logits, loss (1e-3) and parameter gradients (stem, a stage-3 block, GRU, the 2560-wide
    head layer) against the CPU oracle on the same inputs."""
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor, load_scaler_npz
    from adyolo_amd.datasets import synthetic_audio, synthetic_targets
    from oracle import features as ofeat, seresnet as onet, adyolo_loss as oloss
    scaler = load_scaler_npz(os.path.join(G, "scaler_DCASE2022.npz"))
    prm = _params(13)
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 80, 64), (), prm).to("cuda:0")
    assert model.head.yolo_head[1].weight.shape == (2560, 256)
    crit = WrapperCriterion(prm)
    model.train()
    model.encoder.lstm.dropout = 0.0
    audio = synthetic_audio(3, 24000 * 2, seed=33)                      # 3 clips x 2 s -> T = 80, T' = 20
    target = synthetic_targets(3, 20, 13, seed=33)
    assert int(target[:, 4].max()) == 12                                # class 12 only exists with 13 classes
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    feat = FeatureExtractor(scaler, "cuda:0")(audio.to("cuda:0"), channels_last8=True)
    logit = model(feat, channels_last8=True)
    loss = crit(logit, target)
    loss.backward()
    torch.cuda.synchronize()
    f_ref = torch.stack([torch.from_numpy(ofeat.get_feature(audio[b].double().numpy(), scaler)[0]) for b in range(3)])
    names = ["encoder.conv1.weight", "encoder.layer3.2.conv2.weight", "encoder.layer3.2.se.fc.0.weight",
             "encoder.lstm.weight_hh_l1_reverse", "head.yolo_head.0.weight", "head.yolo_head.1.weight", "head.yolo_head.1.bias"]
    for n in names:
        sd[n].requires_grad_(True)
    logit_ref = onet.model_forward(sd, f_ref, training=True)
    loss_ref = oloss.adyolo_loss(logit_ref, target, 13)
    loss_ref.backward()
    assert logit_ref.shape == (3, 20, 2560)
    assert float((feat[..., :7].permute(0, 3, 1, 2).cpu() - f_ref).abs().max()) < 1e-3
    assert float((logit.detach().cpu() - logit_ref).abs().max()) <= 1e-3 * max(1.0, float(logit_ref.abs().max()))
    assert abs(float(loss) - float(loss_ref)) <= 1e-3 * abs(float(loss_ref)), (float(loss), float(loss_ref))
    named = dict(model.named_parameters())
    for n in names:
        got, ref = named[n].grad.cpu(), sd[n].grad
        cos = float(torch.dot(got.reshape(-1).double(), ref.reshape(-1).double()) / (got.double().norm() * ref.double().norm()))
        # toy shapes (T = 80) are ill-conditioned through 16 BatchNorm'd blocks: the statistical bound of DESIGN section 7
        assert cos >= 0.999 and _rel(got, ref) <= 5e-2, "%s: cosine %.6f, max dev %.2e of absmax" % (n, cos, _rel(got, ref))
    for n in ("head.yolo_head.1.weight", "head.yolo_head.1.bias", "head.yolo_head.0.weight"):
        assert _rel(named[n].grad.cpu(), sd[n].grad) <= 1e-3, n         # the C = 13 loss gradient itself, through the head only


def test_config5_adpit_bs64_train_step_matches_oracle(ops):
    """BASELINE config 5's plugin surface at its batch size: se-resnet34 + ADPIT head (256 -> 9 * 12, tanh) + ADPIT loss
    (linearheads.py:70-86, loss.py:70-153) on 64 x 20 s chunks.  The training-mode forward loss of the whole model vs the
    oracle run on the same K1 features (1e-3; features are compared with the oracle elsewhere), then one ``TrainStep``
    (gradient sink, fused Adam) on raw audio: same loss value, every parameter finite and moved."""
    import time
    from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
    from adyolo_amd.features import FeatureExtractor
    from adyolo_amd.datasets import synthetic_audio, ClasswiseLabelEncoder
    from adyolo_amd.train import TrainStep
    from oracle import seresnet as onet, other_losses as ol
    B, n = 64, 24000 * 20
    prm = _params(12)
    prm["args"]["loss"] = "adpit"
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 800, 64), (), prm).to("cuda:0")
    crit = WrapperCriterion(prm)
    model.train()
    model.encoder.lstm.dropout = 0.0
    audio = synthetic_audio(B, n, seed=55).to("cuda:0")
    rng = np.random.default_rng(55)
    enc = ClasswiseLabelEncoder(12)
    labels = []
    for b in range(B):
        ev = {}
        for fr in range(200):
            k = rng.choice(4, p=[0.4, 0.35, 0.2, 0.05])
            if k:        # same-class overlaps included: they exercise the B / C permutation targets of ADPIT
                ev[fr] = [[int(rng.integers(0, 4)), j, float(rng.uniform(-180, 180)), float(rng.uniform(-60, 60))] for j in range(k)]
        labels.append(enc.get_adpit_label(ev, 200))
    target = torch.stack(labels)                                          # (64, 200, 6, 4, 12)
    fx = FeatureExtractor(None, "cuda:0")
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        feat = fx(audio, channels_last8=True)
        out = model(feat, channels_last8=True)
        loss_fwd = float(crit(out, target))
        f_cpu = feat[..., :7].permute(0, 3, 1, 2).contiguous().cpu()
    enc_sd, _ = onet.split_state_dict(sd)
    t0 = time.time()
    with torch.no_grad():
        y = onet.encoder_forward(enc_sd, f_cpu, training=True)
        raw = torch.nn.functional.linear(torch.nn.functional.linear(y, sd["head.adpit_head.0.weight"], sd["head.adpit_head.0.bias"]),
                                         sd["head.adpit_head.1.weight"], sd["head.adpit_head.1.bias"])
        out_ref = torch.tanh(raw)
        loss_ref = float(ol.adpit_loss(out_ref, target, 12))
    print("oracle forward at 64 x 20 s: %.1f s" % (time.time() - t0))
    assert out.shape == (B, 200, 108)
    assert float((out.cpu() - out_ref).abs().max()) <= 1e-3
    assert abs(loss_fwd - loss_ref) <= 1e-3 * abs(loss_ref), (loss_fwd, loss_ref)
    # the BatchNorm running statistics moved in the no_grad forward above: rebuild so TrainStep starts from the same state
    torch.manual_seed(100)
    model = WrapperModel((1, 7, 800, 64), (), prm).to("cuda:0")
    model.encoder.lstm.dropout = 0.0
    trainer = TrainStep(model, crit, fx, prm)
    before = trainer.flat.flat.clone()
    tgt = target.to("cuda:0")
    ls = [float(trainer.step(audio, tgt)) for _ in range(6)]
    torch.cuda.synchronize()
    assert abs(ls[0] - loss_ref) <= 1e-3 * abs(loss_ref), (ls[0], loss_ref)
    # (the very first Adam step moves every parameter by lr whatever its gradient: the loss may go up once)
    assert all(np.isfinite(ls)) and min(ls[2:]) < ls[0], ls
    moved = (trainer.flat.flat != before)[:trainer.flat.numel]
    assert bool(torch.isfinite(trainer.flat.flat).all()) and float(moved.float().mean()) > 0.99


# ------------------------------------------------------------------------------ N = 2 on the real model (two processes, one GPU)
_DP2_CHILD = r"""
import hashlib, json, os, sys
sys.path.insert(0, os.environ["ADYOLO_REPO"])
import torch
import torch.distributed as dist
import adyolo_amd
import bench
from adyolo_amd import dist as adist, functional as Fn
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep

mode = os.environ["ADYOLO_DP2_MODE"]                    # "rank": one of two real processes; "emulate": both halves in one
b, n = 2, 24000 * 4                                     # clips per rank


def make():
    torch.manual_seed(100)
    prm = bench.params("cuda:0")
    model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
    model.encoder.lstm.dropout = 0.0
    return TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm)


def data(r):
    return synthetic_audio(b, n, seed=30 + r).to("cuda:0"), synthetic_targets(b, n // 2400, 12, seed=40 + r).to("cuda:0")


def digest(t):
    return hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()


if mode == "rank":
    rank, world, _ = adist.init_from_env("gloo")        # RCCL refuses two ranks on one device; gloo stages through the host
    try:
        probe = torch.ones(4, device="cuda:0")
        dist.all_reduce(probe)
        assert float(probe[0]) == world
    except Exception as exc:                            # a gloo build without device-tensor support
        print(json.dumps({"skip": repr(exc)[:300]}))
        sys.exit(0)
    tr = make()
    audio, target = data(rank)
    losses, fired, gdig = [], [], []
    opt_step = tr.optimizer.step

    def spy(grad_scale=1.0):                            # the reduced gradients, bucket by bucket, as Adam sees them
        gdig.append([digest(tr.flat.flat_grad[s:e]) for s, e, _ in tr.reducer.buckets])
        opt_step(grad_scale=grad_scale)
    tr.optimizer.step = spy
    for _ in range(3):
        h0, f0 = tr.reducer.fired_from_hooks, tr.reducer.fired_from_finish
        losses.append(float(tr.step(audio, target)))
        fired.append((tr.reducer.fired_from_hooks - h0, tr.reducer.fired_from_finish - f0))
    torch.cuda.synchronize()
    print(json.dumps({"rank": rank, "world": world, "active": tr.reducer.active, "buckets": len(tr.reducer.buckets),
                      "fired": fired, "losses": losses, "params": digest(tr.flat.flat), "grads": gdig,
                      "bn": digest(tr.model.encoder.bn1.running_mean)}))
    dist.barrier()
    dist.destroy_process_group()
else:
    trs = [make(), make()]
    dat = [data(0), data(1)]
    losses = [[], []]
    gdig = []
    for _ in range(3):
        for r, tr in enumerate(trs):                    # what rank r computes before the exchange
            tr.model.train()
            out = tr.model(tr.features(dat[r][0], channels_last8=True), channels_last8=True)
            tr.optimizer.zero_grad()
            loss = tr.criterion(out, dat[r][1])
            Fn.SINK.begin(tr.flat, tr.reducer)
            try:
                loss.backward()
            finally:
                Fn.SINK.end()
            losses[r].append(float(loss))
        total = trs[0].flat.flat_grad + trs[1].flat.flat_grad      # the all-reduce (sum of two floats: order-free)
        gdig.append([digest(total[s:e]) for s, e, _ in trs[0].reducer.buckets])
        for tr in trs:
            tr.flat.flat_grad.copy_(total)
            tr.optimizer.step(grad_scale=0.5)
    torch.cuda.synchronize()
    print(json.dumps({"emulated": True, "losses": losses, "params": [digest(t.flat.flat) for t in trs], "grads": gdig,
                      "bn": [digest(t.model.encoder.bn1.running_mean) for t in trs]}))
"""


def test_two_ranks_on_the_real_model_match_the_sequential_emulation(ops):
    """Data parallelism with N = 2 on the REAL model and the real hook / gradient-sink path: two processes (both on this
    one GPU; backend gloo, because RCCL refuses two ranks on one device) run 3 TrainSteps on different half batches.
    Checked: every bucket's all-reduce is issued from a hook / ``GradSink.notify`` during backward on BOTH ranks (none by
    finish()), both ranks end with bit-identical parameters, and these equal a single-process emulation that computes the
    two half-batch gradients one after the other, adds them and applies the same fused Adam with 1/2 -- i.e. the bucket
    layout, the sink's write-through into the flat buffer and the launch order are consistent across ranks.  BatchNorm
    running statistics stay per rank (DDP-conventional semantics): rank r's equal the emulation's r."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, ADYOLO_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    base.pop("ADYOLO_FORCE_DP_HOOKS", None)
    procs = [subprocess.Popen([sys.executable, "-c", _DP2_CHILD], env=dict(base, ADYOLO_DP2_MODE="rank", WORLD_SIZE="2",
                                                                           RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith("{")][-1]))
    if any("skip" in o for o in outs):
        pytest.skip("gloo cannot reduce device tensors in this build: %r" % outs)
    r = subprocess.run([sys.executable, "-c", _DP2_CHILD], env=dict(base, ADYOLO_DP2_MODE="emulate", WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    emu = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    outs.sort(key=lambda o: o["rank"])
    for o in outs:
        assert o["world"] == 2 and o["active"] and o["buckets"] == 4
        assert all(f == [4, 0] for f in o["fired"]), o["fired"]          # all four buckets overlapped with backward
    where = [(st, bk, outs[0]["grads"][st][bk] == outs[1]["grads"][st][bk], outs[0]["grads"][st][bk] == emu["grads"][st][bk])
             for st in range(3) for bk in range(4)]
    assert all(a and b for _, _, a, b in where), "reduced gradients (step, bucket, rank0 == rank1, rank0 == emulation): %r" % where
    assert outs[0]["params"] == outs[1]["params"], "ranks diverged"
    assert outs[0]["params"] == emu["params"][0] == emu["params"][1], "two ranks != sequential emulation"
    for rk in range(2):
        assert outs[rk]["losses"] == emu["losses"][rk]
        assert outs[rk]["bn"] == emu["bn"][rk]
    assert outs[0]["bn"] != outs[1]["bn"]                                 # per-rank running statistics (different shards)


_DPX_CHILD = r"""
import json, os, sys
sys.path.insert(0, os.environ["ADYOLO_REPO"])
import torch
import torch.distributed as dist
import adyolo_amd
import bench
from adyolo_amd import dist as adist
from adyolo_amd.wrapper import WrapperModel, WrapperCriterion
from adyolo_amd.features import FeatureExtractor
from adyolo_amd.datasets import synthetic_audio, synthetic_targets
from adyolo_amd.train import TrainStep

mode = os.environ["ADYOLO_DPX_MODE"]                    # "rank": one of two exact-mode processes; "single": the whole batch
b, n = 2, 24000 * 4                                     # clips per rank


loss_nm = os.environ.get("ADYOLO_DPX_LOSS", "adyolo")


def data(r):
    if loss_nm == "adpit":                              # dense (B, T', 6, 4, C) activity / direction targets
        import numpy as np
        rng = np.random.default_rng(60 + r)
        tp = n // 2400
        tgt = np.zeros((b, tp, 6, 4, 12), dtype=np.float32)
        act = rng.random((b, tp, 12)) < 0.2
        xyz = rng.normal(size=(b, tp, 3, 12)).astype(np.float32)
        xyz /= np.linalg.norm(xyz, axis=2, keepdims=True)
        tgt[:, :, 0, 0, :] = act
        tgt[:, :, 0, 1:, :] = xyz * act[:, :, None, :]
        return synthetic_audio(b, n, seed=50 + r), torch.from_numpy(tgt)
    return synthetic_audio(b, n, seed=50 + r), synthetic_targets(b, n // 2400, 12, seed=60 + r)


if mode == "rank":
    rank, world, _ = adist.init_from_env("gloo")        # two ranks on ONE GPU: gloo (RCCL refuses that)
    audio, target = data(rank)
else:
    rank, world = 0, 1
    (a0, t0), (a1, t1) = data(0), data(1)
    t1 = t1.clone()
    if loss_nm == "adyolo":
        t1[:, 0] += b                                   # rank 1's clips are samples b .. 2b-1 of the concatenated batch
    audio, target = torch.cat([a0, a1]), torch.cat([t0, t1])
torch.manual_seed(100)
prm = bench.params("cuda:0")
prm["args"]["loss"] = loss_nm
model = WrapperModel((1, 7, n // 600, 64), (), prm).to("cuda:0")
model.encoder.lstm.dropout = 0.0
tr = TrainStep(model, WrapperCriterion(prm), FeatureExtractor(None, "cuda:0"), prm, exact=True)
audio, target = audio.to("cuda:0"), target.to("cuda:0")
grads, losses = [], []
params0 = tr.flat.flat.detach().cpu().clone()
opt_step = tr.optimizer.step


def spy(grad_scale=1.0):
    grads.append(tr.flat.flat_grad.clone() * grad_scale)          # what Adam sees
    # exact mode + AD-YOLO (and one device): gradients are sums, never averaged; the class-wise losses normalise over the local
    # rows, so their rank gradients are AVERAGED (ADVICE round 3: they used to be summed)
    assert grad_scale == (1.0 if loss_nm == "adyolo" or world == 1 else 1.0 / world), grad_scale
    opt_step(grad_scale=grad_scale)
tr.optimizer.step = spy
bn_first = None
for i in range(3):
    losses.append(float(tr.step(audio, target)))
    if i == 0:
        bn_first = {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items() if "running_" in k}
torch.cuda.synchronize()
sd = tr.model.state_dict()
torch.save({"grad0": grads[0].cpu(), "params": tr.flat.flat.cpu(), "params0": params0, "losses": losses, "bn_first": bn_first,
            "bn": {k: v.cpu() for k, v in sd.items() if "running_" in k},
            "layout": [(off, cnt) for off, cnt in tr.flat.offsets]}, os.environ["ADYOLO_DPX_OUT"])
print(json.dumps({"ok": True, "rank": rank, "losses": losses}))
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
"""


@pytest.mark.parametrize("loss_nm", ["adyolo", "adpit"])
def test_exact_data_parallel_equals_one_device_on_the_concatenated_batch(ops, tmp_path, loss_nm):
    """``TrainStep(exact=True)`` (ops.ExactDP; SURVEY 8e "optional"): two ranks on two different 2-clip shards (two processes
    on this one GPU, gloo) against ONE process on the 4-clip concatenation, real SE-ResNet34 + AD-YOLO model, 3 Adam steps.
    BatchNorm statistics are formed over all ranks' samples by the same finishing kernel on the gathered per-sample sums
    -> running statistics BIT-identical after the first step (the whole forward pass is) and within 1e-3 after 3 Adam steps; the loss counts are all-reduced -> the first loss value agrees to
    1e-6 (measured 7e-8), the later ones to 1e-4 (measured 3e-6); the gradients Adam sees in step 0 (summed over the ranks, not averaged) agree per parameter tensor
    to 2e-5 of its absmax (only summation orders differ); both ranks hold the same parameters, and their 3-step update is
    the one-device update (cosine >= 0.999).  loss_nm = adpit (round 4): a class-wise loss, whose normaliser is the local row
    count -- there the exact step averages the ranks' gradients and reports the ranks' mean loss."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    base = dict(os.environ, ADYOLO_REPO=repo, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0",
                ADYOLO_DPX_LOSS=loss_nm)
    base.pop("ADYOLO_FORCE_DP_HOOKS", None)
    outs = [str(tmp_path / ("r%d.pt" % r)) for r in range(2)] + [str(tmp_path / "single.pt")]
    procs = [subprocess.Popen([sys.executable, "-c", _DPX_CHILD],
                              env=dict(base, ADYOLO_DPX_MODE="rank", WORLD_SIZE="2", RANK=str(r), LOCAL_RANK="0", ADYOLO_DPX_OUT=outs[r]),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, e[-3000:]
    r = subprocess.run([sys.executable, "-c", _DPX_CHILD], env=dict(base, ADYOLO_DPX_MODE="single", WORLD_SIZE="1", RANK="0", ADYOLO_DPX_OUT=outs[2]),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    r0, r1, one = (torch.load(f) for f in outs)
    assert torch.equal(r0["params"], r1["params"]) and torch.equal(r0["grad0"], r1["grad0"]), "ranks diverged"
    for i, (a, c) in enumerate(zip(r0["losses"], one["losses"])):      # step 0: only summation orders differ; later steps
        assert abs(a - c) <= (1e-6 if i == 0 else 1e-4) * abs(c), (r0["losses"], one["losses"])      # carry Adam's amplification
    worst = 0.0
    for off, cnt in one["layout"]:
        g, h = r0["grad0"][off:off + cnt].double(), one["grad0"][off:off + cnt].double()
        am = float(h.abs().max())
        if am > 0:
            worst = max(worst, float((g - h).abs().max()) / am)
    # (adpit: measured 2.05e-5 -- its 13-way arg-min picks are all-or-nothing per frame; bound 5e-5)
    assert worst <= (2e-5 if loss_nm == "adyolo" else 5e-5), "step-0 gradients: worst tensor deviates by %.2e of its absmax" % worst
    for k, v in one["bn_first"].items():                  # the whole first forward pass is bit-identical
        assert torch.equal(r0["bn_first"][k], v) and torch.equal(r1["bn_first"][k], v), k
    for k, v in one["bn"].items():                        # after 3 Adam steps: round-off amplified by Adam's normalisation
        d = float((r0["bn"][k] - v).abs().max())
        assert d <= 1e-3 * max(1.0, float(v.abs().max())), (k, d)
    # Adam divides every gradient element by its own magnitude, so an element whose gradient is round-off-sized moves by +-lr
    # either way: compare the 3-step UPDATES as vectors (measured: mean deviation 7e-4 of the mean parameter magnitude)
    assert torch.equal(r0["params0"], one["params0"])
    ua, ub = (r0["params"] - r0["params0"]).double(), (one["params"] - one["params0"]).double()
    cos = float(torch.dot(ua, ub) / (ua.norm() * ub.norm()))
    assert cos >= 0.999, "3-step parameter updates: cosine %.6f" % cos
    rel = float((r0["params"] - one["params"]).abs().mean()) / float(one["params"].abs().mean())
    assert rel <= 3e-3, "parameters after 3 steps: mean deviation %.2e of the mean magnitude" % rel


def test_bench_two_ranks_functional_run_on_one_device(ops):
    """``bench.py --gpus 2`` end to end on this single-GPU box: the launcher starts two workers, both on cuda:0 over gloo
    (ADYOLO_DIST_BACKEND / ADYOLO_BENCH_ONE_DEVICE: RCCL refuses two ranks on one device), barrier + max-over-ranks timing,
    rank 0 prints ONE JSON line with n_gpus = 2 and the whole-job aggregate.  A functional check of the N > 1 path of the
    benchmark (launcher, rank env, collectives, JSON contract) -- its numbers mean nothing."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ADYOLO_DIST_BACKEND="gloo", ADYOLO_BENCH_ONE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ADYOLO_FORCE_DP_HOOKS"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--seconds", "4", "--no-cpu-baseline", "--no-stages", "--no-extra"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["config"]["global_batch"] == 4 and "dp2" in d["config"]["parallelism"]
    assert abs(d["value"] - 2 * 2 * 4 / (d["ms_per_step"] * 1e-3)) <= 0.01 * d["value"]     # whole-job audio-seconds per second
    assert np.isfinite(d["final_loss"]) and d["roofline"]["launches"] > 0 and "cpu_baseline" not in d
    # the N > 1 evidence block (round 4): the process group's own view of the job and what the gradient reducer did per step
    rc = d["rccl"]
    assert rc["backend"] == "gloo" and rc["world_size"] == 2 and rc["all_reduce_of_ones"] == 2.0 and rc["all_reduce_ok"] is True
    assert rc["buckets"] >= 2 and len(rc["bucket_MB"]) == rc["buckets"] and abs(sum(rc["bucket_MB"]) - 26.73) < 0.1
    assert rc["buckets_fired_from_hooks_per_step"] + rc["buckets_fired_from_finish_per_step"] == rc["buckets"]
    assert rc["buckets_fired_from_hooks_per_step"] >= rc["buckets"] - 1           # at most the last bucket waits for finish()
    assert rc["ms_per_step_reducer_off"] > 0 and abs(rc["exposed_allreduce_ms"] - (d["ms_per_step"] - rc["ms_per_step_reducer_off"])) < 0.01
    # the host-fed variant of the same step (int16 clips through AudioStager)
    pl = d["pipeline"]
    assert pl["ms_per_step"] > 0 and abs(pl["value"] - 2 * 2 * 4 / (pl["ms_per_step"] * 1e-3)) <= 0.01 * pl["value"]
    assert abs(pl["h2d_MB_per_step_per_gpu"] - 2 * 96000 * 4 * 2 / 1e6) < 0.11
