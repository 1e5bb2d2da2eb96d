"""ResNet-Conformer encoder (BASELINE config 4) on the gfx950 kernels.

Host-side mirror of ``ResnetConformer`` (/root/reference/src/models/backbones/resnet_conformer.py:342-447) with its
building blocks ``MultiHeadAttention`` :25-85, ``ConformerConvModule`` :154-178, ``FeedForwardModule`` :181-212,
``ConformerBlock`` :215-282, ``PoolingModule`` :285-297 and torchvision's ``BasicBlock`` (conv3x3(stride)-BN-ReLU-conv3x3-BN
+ identity/downsample -> ReLU; torchvision==0.11 is not vendored in the reference: "parity unpinned" for that block).
Same constructor ``(in_feat_shape, out_shape, params)``, ``enc_out_dim`` attribute, forward ``(B,7,T,F) -> (B,T//4,256)``
and the same 549 ``state_dict`` keys.  Quirks kept: ReLU *before* BatchNorm after the 7x7 stem (:423-425), layer3 has 5
blocks (:373-384), ``PoolingModule.max_pool`` is an average pool (:289) so pooling is 2 x avg, attention ``mask`` path dead.

Strided convolutions (and the stride-1 ones on maps narrower than 8 bins) run on the implicit-GEMM convolution
(``adyolo_conv_gemm``: no column buffer), the other stride-1 3x3 convolutions on the Winograd kernels; attention is the flash-style fp32-MFMA kernel of csrc/attention.hip
(scores never written to HBM).
"""
import math

import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from ...rng import DropoutStream
from .resnet import BatchNormParams, ConvParams, LayerNormParams, LinearParams


def _dropout(x, p, training, rng):
    if not training or p <= 0.0:
        return x
    if x.numel() % 4:
        return Fn.DropoutFn.apply(x, rng.mask(x, p))
    return Fn.DropoutHashFn.apply(x, p, *rng.draw(x.numel()))


def _dropout_residual(y, z, a, p, training, rng):
    """a * dropout(y) + z: fused into one pass when the stateless mask applies (training, p > 0, element count a multiple
    of 4); the stream is drawn exactly as ``_dropout`` would draw it."""
    if training and p > 0.0 and y.numel() % 4 == 0:
        return Fn.DropoutAxpbyFn.apply(y, z, a, 1.0, p, *rng.draw(y.numel()))
    return Fn.AxpbyFn.apply(_dropout(y, p, training, rng), z, a, 1.0)


def _conv3x3_s1(x, w):
    """Stride-1 3x3 convolution.  Winograd kernels on wide maps (8 bins: patches two tiles wide); the deep stages of this ResNet
    have strided the frequency axis down to 4, 2 and 1 bins: 4 bins take the F(4x4) kernel with patches ONE tile wide (round 5),
    2 and 1 bins ARE 3 x 1 convolutions along time (1-D Winograd F(4, 3) when the time axis is long, csrc/wino1d.hip; else the
    implicit-GEMM convolution without the taps that only ever meet the zero padding):
    W == 1: only the centre kernel column meets data, the convolution IS the 3 x 1 one on w[:, :, :, 1:2];
    W == 2: every (input bin, output bin) pair is within one tap, so the two bins fold into the channel axis
            (free views [N][H][1][2 C]) and the convolution IS a 3 x 1 one with the 2 Cout x 2 Cin block filter
            w2[wo*Cout + co][wi*Cin + ci][kh] = w[co][ci][kh][wi - wo + 1]  -- 2/3 of the multiplies of the padded form.
    The filter rearrangements are differentiable torch views / copies of the (tiny) weight, so the untouched taps get
    exactly zero gradient, as in the reference."""
    n, h, wd, cin = x.shape
    cout = w.shape[0]
    if wd == 1:
        w1 = w[:, :, :, 1:2].contiguous()
        if ops.wino1d_ok(n, h, cin, cout):        # long time axis: 1-D Winograd F(4, 3), half the matrix work (csrc/wino1d.hip)
            return Fn.Conv3x1WinoFn.apply(x, w1)
        return Fn.ConvFn.apply(x, w1, (1, 1), (1, 0))
    if wd == 2:
        w2 = torch.stack((w[..., 1:3], w[..., 0:2]), 0)                     # [wo][co][ci][kh][wi]
        w2 = w2.permute(0, 1, 4, 2, 3).reshape(2 * cout, 2 * cin, 3, 1)
        if ops.wino1d_ok(n, h, 2 * cin, 2 * cout):
            return Fn.Conv3x1WinoFn.apply(x.view(n, h, 1, 2 * cin), w2).view(n, h, 2, cout)
        return Fn.ConvFn.apply(x.view(n, h, 1, 2 * cin), w2, (1, 1), (1, 0)).view(n, h, 2, cout)
    if wd <= 4:
        if ops.w4_narrow_ok(cin, cout):           # F(4x4) with patches one tile wide for forward / data gradient (wino4p.hpp)
            return Fn.Conv3x3NarrowFn.apply(x, w)
        return Fn.ConvFn.apply(x, w, (1, 1), (1, 1))
    return Fn.Conv3x3S1Fn.apply(x, w)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=(1, 1), downsample=None):
        super().__init__()
        self.stride = tuple(stride)
        self.conv1 = ConvParams(inplanes, planes, 3, bias=False)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, bias=False)
        self.bn2 = BatchNormParams(planes)
        self.downsample = downsample

    def forward(self, x):
        if self.stride == (1, 1):
            out = _conv3x3_s1(x, self.conv1.weight)
        else:
            out = Fn.ConvFn.apply(x, self.conv1.weight, self.stride, (1, 1))
        out = Fn.BatchNormFn.apply(out, self.bn1.weight, self.bn1.bias, self.bn1, self.training, True, None)
        out = _conv3x3_s1(out, self.conv2.weight)
        if self.downsample is not None:
            idn = Fn.ConvFn.apply(x, self.downsample["0"].weight, self.stride, (0, 0))
            d = self.downsample["1"]
            idn = Fn.BatchNormFn.apply(idn, d.weight, d.bias, d, self.training, False, None)
        else:
            idn = x
        return Fn.BatchNormFn.apply(out, self.bn2.weight, self.bn2.bias, self.bn2, self.training, True, idn)


def _make_layer(inplanes, planes, nblocks):
    down = nn.ModuleDict({"0": ConvParams(inplanes, planes, 1, bias=False), "1": BatchNormParams(planes)})
    blocks = [BasicBlock(inplanes, planes, (1, 2), down)]
    blocks += [BasicBlock(planes, planes) for _ in range(1, nblocks)]
    return nn.ModuleList(blocks)


class Conv1dParams(nn.Module):
    """weight (Cout, Cin/groups, k) + bias, initialised like nn.Conv1d."""

    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin // groups, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(cin // groups * k), 1.0 / math.sqrt(cin // groups * k))


class FeedForwardModule(nn.Module):
    """LN -> Linear(d, 4d) -> Swish -> Dropout -> Linear(4d, d) -> Dropout; keys sequential.{0,1,4}."""

    def __init__(self, dim, expansion, p):
        super().__init__()
        self.p = p
        self.sequential = nn.ModuleDict({"0": LayerNormParams(dim), "1": LinearParams(dim, dim * expansion),
                                         "4": LinearParams(dim * expansion, dim)})

    def forward(self, x, rng, residual=None):
        """residual=(z, a): returns a * module(x) + z with the final Dropout fused into the mix"""
        s = self.sequential
        y = Fn.LNFn.apply(x, s["0"].weight, s["0"].bias, s["0"].eps)
        y = Fn.LinearFn.apply(y, s["1"].weight, s["1"].bias)
        y = _dropout(Fn.SwishFn.apply(y), self.p, self.training, rng)
        y = Fn.LinearFn.apply(y, s["4"].weight, s["4"].bias)
        if residual is not None:
            return _dropout_residual(y, residual[0], residual[1], self.p, self.training, rng)
        return _dropout(y, self.p, self.training, rng)


class MultiHeadAttention(nn.Module):
    def __init__(self, emb_dim, num_heads, p):
        super().__init__()
        self.emb_dim, self.num_heads, self.p = emb_dim, num_heads, p
        self.scaling = (emb_dim // num_heads) ** -0.5
        self.value = LinearParams(emb_dim, emb_dim)
        self.key = LinearParams(emb_dim, emb_dim)
        self.query = LinearParams(emb_dim, emb_dim)
        self.linear = LinearParams(emb_dim, emb_dim)

    def forward(self, x, rng):
        q = Fn.LinearFn.apply(x, self.query.weight, self.query.bias)
        k = Fn.LinearFn.apply(x, self.key.weight, self.key.bias)
        v = Fn.LinearFn.apply(x, self.value.weight, self.value.bias)
        drop = None
        if self.training and self.p > 0.0:
            b, t, _ = x.shape
            drop = (self.p, rng.seed32(b * self.num_heads * t * t))       # stateless in-kernel mask: no (B, H, T, T) tensor
        ctx = Fn.AttentionCoreFn.apply(q, k, v, self.num_heads, self.scaling, drop)
        return Fn.LinearFn.apply(ctx, self.linear.weight, self.linear.bias)


class ConformerConvModule(nn.Module):
    """LN -> pw conv (d -> 2d) -> BN -> GLU -> depthwise conv k3 dil -> BN -> Swish -> pw conv -> Dropout;
    keys conv.{0,2,3,5,6,8}."""

    def __init__(self, dim, dilation, p=0.2):
        super().__init__()
        self.dilation, self.p = dilation, p
        self.conv = nn.ModuleDict({"0": LayerNormParams(dim), "2": Conv1dParams(dim, 2 * dim, 1),
                                   "3": BatchNormParams(2 * dim), "5": Conv1dParams(dim, dim, 3, groups=dim),
                                   "6": BatchNormParams(dim), "8": Conv1dParams(dim, dim, 1)})

    def forward(self, x, rng, residual=None):
        c = self.conv
        y = Fn.LNFn.apply(x, c["0"].weight, c["0"].bias, c["0"].eps)
        y = Fn.LinearFn.apply(y, c["2"].weight.view(c["2"].weight.shape[0], -1), c["2"].bias)
        y = Fn.BatchNormFn.apply(y, c["3"].weight, c["3"].bias, c["3"], self.training, False, None)
        y = Fn.GLUFn.apply(y)
        y = Fn.DWConv3Fn.apply(y, c["5"].weight, c["5"].bias, self.dilation)
        y = Fn.BatchNormFn.apply(y, c["6"].weight, c["6"].bias, c["6"], self.training, False, None)
        y = Fn.SwishFn.apply(y)
        y = Fn.LinearFn.apply(y, c["8"].weight.view(c["8"].weight.shape[0], -1), c["8"].bias)
        if residual is not None:
            return _dropout_residual(y, residual[0], residual[1], self.p, self.training, rng)
        return _dropout(y, self.p, self.training, rng)


class _Residual(nn.Module):
    """keys ``<idx>.module.*`` like the reference's ResidualConnectionModule."""

    def __init__(self, module, factor):
        super().__init__()
        self.module = module
        self.factor = factor


class _AttnBranch(nn.ModuleDict):
    pass


class ConformerBlock(nn.Module):
    def __init__(self, dim, heads, expansion, p1, p2, dilation):
        super().__init__()
        self.p2 = p2
        self.sequential = nn.ModuleList([
            _Residual(FeedForwardModule(dim, expansion, p1), 0.5),
            _Residual(_AttnBranch({"0": LayerNormParams(dim), "1": MultiHeadAttention(dim, heads, p1)}), 0.5),
            _Residual(ConformerConvModule(dim, dilation), 1.0),
            _Residual(FeedForwardModule(dim, expansion, p1), 0.5),
            LayerNormParams(dim),
        ])

    def forward(self, x, rng):
        s = self.sequential
        x = s[0].module(x, rng, residual=(x, s[0].factor))          # x + 0.5 * FFN(x), final Dropout fused into the mix
        a = s[1].module
        y = Fn.LNFn.apply(x, a["0"].weight, a["0"].bias, a["0"].eps)
        x = _dropout_residual(a["1"](y, rng), x, s[1].factor, self.p2, self.training, rng)
        x = s[2].module(x, rng, residual=(x, s[2].factor))
        x = s[3].module(x, rng, residual=(x, s[3].factor))
        return Fn.LNFn.apply(x, s[4].weight, s[4].bias, s[4].eps)


class ConformerEncoder(nn.Module):
    def __init__(self, n_layers, dim, heads, expansion, p1, p2):
        super().__init__()
        self.encoder_module = nn.ModuleList([ConformerBlock(dim, heads, expansion, p1, p2, 2 ** i)
                                             for i in range(n_layers)])

    def forward(self, x, rng):
        for blk in self.encoder_module:
            x = blk(x, rng)
        return x


class PoolingModule(nn.Module):
    """avg_pool(k) + "max_pool"(k) where both are AvgPool1d in the reference (:288-289) -> 2 x mean, then LayerNorm."""

    def __init__(self, pool, dim):
        super().__init__()
        self.pool = pool
        self.norm = LayerNormParams(dim)

    def forward(self, x):
        y = Fn.AvgPool1dFn.apply(x, self.pool, 2.0)
        return Fn.LNFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps)


class ResnetConformer(nn.Module):
    def __init__(self, in_feat_shape, out_shape, params):
        super().__init__()
        self.in_channels = in_feat_shape[1]
        self.conv1 = ConvParams(self.in_channels, 64, 7, bias=False)
        self.bn1 = BatchNormParams(64)
        self.layer1 = _make_layer(64, 64, 3)
        self.layer2 = _make_layer(64, 128, 4)
        self.layer3 = _make_layer(128, 256, 5)
        self.layer4 = _make_layer(256, 512, 3)
        self.bottleneck = nn.Module()
        self.bottleneck.weight = nn.Parameter(torch.empty(256, 512))
        nn.init.kaiming_uniform_(self.bottleneck.weight, a=math.sqrt(5))
        self.emb_dim = 256
        self.conformer = ConformerEncoder(8, 256, 4, 4, 0.2, 0.2)
        self.t_pooling = PoolingModule(4, 256)
        self.enc_out_dim = 256
        self._rng = DropoutStream(0xC0F0)

    def forward(self, x, channels_last8=False):
        """x: (B, 7, T, F) float32 on the GPU (reference layout), or (B, T, F, 8) when ``channels_last8``."""
        if not x.is_cuda:
            raise RuntimeError("ResnetConformer (adyolo_amd) runs on MI355X only; move the model/input to a HIP device")
        with Fn.bn_counter_scope():          # the BatchNorm step counters of this forward: one launch on exit
            return self._forward(x, channels_last8)

    def _forward(self, x, channels_last8):
        x8 = x if channels_last8 else ops.nchw_to_nhwc8(x.contiguous().float())
        w = self.conv1.weight
        if w.shape[1] < 8:          # activations are padded 7 -> 8 channels: pad the weight with zero planes (copy only)
            w = torch.cat([w, torch.zeros(w.shape[0], 8 - w.shape[1], 7, 7, dtype=w.dtype, device=w.device)], dim=1)
        y = Fn.ConvFn.apply(x8, w, (1, 2), (3, 3))
        y = Fn.ReluFn.apply(y)                                   # ReLU BEFORE BatchNorm (reference :423-425)
        y = Fn.BatchNormFn.apply(y, self.bn1.weight, self.bn1.bias, self.bn1, self.training, False, None)
        y = Fn.MaxPool3Fn.apply(y)
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            for blk in layer:
                y = blk(y)
        b, t, f, c = y.shape
        if f != 1:
            raise RuntimeError("ResnetConformer expects 64 mel bins (frequency axis must collapse to 1, got %d)" % f)
        y = y.view(b, t, c)
        y = Fn.LinearFn.apply(y, self.bottleneck.weight, None)
        y = self.conformer(y, self._rng)
        return self.t_pooling(y)
