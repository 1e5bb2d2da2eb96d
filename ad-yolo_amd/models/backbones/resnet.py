"""SE-ResNet34 + self-attention pooling + BiGRU encoder on hand-written gfx950 kernels.

Host-side mirror of the reference plugin ``SEResnet34`` (/root/reference/src/models/backbones/resnet.py:126-199):
same constructor ``(in_feat_shape, out_shape, params)``, same ``enc_out_dim`` attribute, same
``forward(x (B,7,T,F)) -> (B, T//4, 256)`` and -- because checkpoints are loaded with ``strict=True``
(reference train.py:148, test.py:90) -- the same ``state_dict`` keys and shapes (301 entries).
The modules below only HOLD parameters/buffers under the reference names; all arithmetic is in
``functional.py`` (autograd nodes) -> ``ops.py`` -> ``libadyolo_hip.so``.  There is no CPU path.
"""
import math

import torch
import torch.nn as nn

from ... import functional as Fn
from ... import ops
from ...rng import DropoutStream

LAYERS = (3, 4, 6, 3)
WIDTHS = (32, 64, 128, 256)


def _uniform_(t, bound):
    return nn.init.uniform_(t, -bound, bound)


class ConvParams(nn.Module):
    """weight (Cout,Cin,k,k) [+ bias]; initialised like nn.Conv2d (kaiming_uniform(a=sqrt(5)))."""

    def __init__(self, cin, cout, k, bias):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, k, k))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if bias:
            self.bias = nn.Parameter(torch.empty(cout))
            _uniform_(self.bias, 1.0 / math.sqrt(cin * k * k))
        else:
            self.register_parameter("bias", None)


class LinearParams(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.empty(cout))
        _uniform_(self.bias, 1.0 / math.sqrt(cin))


class BatchNormParams(nn.Module):
    momentum = 0.1
    eps = 1e-5

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class SEParams(nn.Module):
    """keys se.fc.0.{weight,bias}, se.fc.2.{weight,bias} (reference resnet.py:95-100)."""

    def __init__(self, c, reduction=8):
        super().__init__()
        self.fc = nn.ModuleDict({"0": LinearParams(c, c // reduction), "2": LinearParams(c // reduction, c)})


class SEBasicBlock(nn.Module):
    def __init__(self, inplanes, planes, downsample, pool):
        super().__init__()
        self.pool = pool is not None
        self.conv1 = ConvParams(inplanes, planes, 3, bias=False)
        self.bn1 = BatchNormParams(planes)
        self.conv2 = ConvParams(planes, planes, 3, bias=False)
        self.bn2 = BatchNormParams(planes)
        self.se = SEParams(planes)
        if downsample:
            self.downsample = downsample
        else:
            self.downsample = None

    def forward(self, x, link_in=None, link_out=None, in_affine=None, stem_holder=None, packs=None, pool_next=False):
        """link_in / link_out: ``functional.BlockLink`` shared with the block below / above (see FUSE_SEBWD);
        in_affine = (scale, shift): x is seen through this per-channel affine (the stem's un-materialised BatchNorm);
        packs = the Winograd-packed filters of conv1 / conv2 from the encoder's ``ops.WinoPackSet`` (one launch for all);
        pool_next: the block that consumes our output starts with AvgPool2d(2, 2) -- we may hand it the pooled tensor
        (``functional.FUSE_POOL``; ``link_out.prepooled`` tells it)."""
        fc0, fc2 = self.se.fc["0"], self.se.fc["2"]
        args = [x, self.training, self.pool,
                (self.bn1, self.bn2, self.downsample["1"] if self.downsample is not None else None, link_in, link_out,
                 in_affine, stem_holder, packs, pool_next),
                self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight, self.bn2.weight, self.bn2.bias,
                fc0.weight, fc0.bias, fc2.weight, fc2.bias]
        if self.downsample is not None:
            args += [self.downsample["0"].weight, self.downsample["1"].weight, self.downsample["1"].bias]
        return Fn.SEBlockFn.apply(*args)


class AttentionParams(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.W = LinearParams(c, 1)


class GRUParams(nn.Module):
    """Parameter names of nn.GRU(256, 128, num_layers=2, bidirectional=True) (attribute ``lstm`` upstream)."""

    dropout = 0.3

    def __init__(self, cin=256, hidden=128, layers=2):
        super().__init__()
        bound = 1.0 / math.sqrt(hidden)
        for layer in range(layers):
            lin = cin if layer == 0 else 2 * hidden
            for sfx in ("", "_reverse"):
                for name, shape in (("weight_ih", (3 * hidden, lin)), ("weight_hh", (3 * hidden, hidden)),
                                    ("bias_ih", (3 * hidden,)), ("bias_hh", (3 * hidden,))):
                    prm = nn.Parameter(torch.empty(*shape))
                    _uniform_(prm, bound)
                    setattr(self, "%s_l%d%s" % (name, layer, sfx), prm)

    def layer_params(self, layer):
        names = ("weight_ih", "weight_hh", "bias_ih", "bias_hh")
        return [getattr(self, "%s_l%d%s" % (n, layer, sfx)) for sfx in ("", "_reverse") for n in names]


class LayerNormParams(nn.Module):
    eps = 1e-5

    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class SEResnet34(nn.Module):
    def __init__(self, in_feat_shape, out_shape, params):
        super().__init__()
        n_in = in_feat_shape[1]
        self.nb_classes = params["data_config"]["nb_classes"]
        self.enc_out_dim = WIDTHS[-1]
        self.in_channels = n_in
        self.conv1 = ConvParams(n_in, WIDTHS[0], 3, bias=True)
        self.bn1 = BatchNormParams(WIDTHS[0])
        inpl = WIDTHS[0]
        for li, (nblk, planes, pool) in enumerate(zip(LAYERS, WIDTHS, (None, 2, 2, None)), start=1):
            down = None
            if inpl != planes:          # created before the block, like _make_layer (resnet.py:157-164)
                down = nn.ModuleDict({"0": ConvParams(inpl, planes, 1, bias=False), "1": BatchNormParams(planes)})
            blocks = [SEBasicBlock(inpl, planes, down, pool)]
            inpl = planes
            blocks += [SEBasicBlock(inpl, planes, None, None) for _ in range(1, nblk)]
            setattr(self, "layer%d" % li, nn.ModuleList(blocks))
        self.attention = AttentionParams(WIDTHS[-1])
        self.lstm = GRUParams(WIDTHS[-1], WIDTHS[-1] // 2, 2)
        self.norm = LayerNormParams(WIDTHS[-1])
        # dropout stream of the inter-layer GRU dropout (counter based: seed from torch.initial_seed() and the rank,
        # running offset; saved / restored with the checkpoint's rng_state)
        self.dropout_stream = DropoutStream(0x5EED)
        self.dropout_mask_override = None       # tests inject a (B,T',256) mask here
        self._packs = ops.WinoPackSet()          # Winograd forms of the 32 block filters, refreshed by one launch per forward

    def _dropout(self, y):
        p = self.lstm.dropout
        if not self.training or p <= 0.0:
            return y
        if self.dropout_mask_override is not None:
            mask = self.dropout_mask_override.to(y.device, torch.float32).contiguous()
        elif y.numel() % 4 == 0:
            return Fn.DropoutHashFn.apply(y, p, *self.dropout_stream.draw(y.numel()))
        else:
            mask = self.dropout_stream.mask(y, p)
        return Fn.DropoutFn.apply(y, mask)

    def forward(self, x, channels_last8=False):
        """x: (B, 7, T, F) float32 on the GPU (reference layout), or (B, T, F, 8) when ``channels_last8``."""
        if not x.is_cuda:
            raise RuntimeError("SEResnet34 (adyolo_amd) runs on MI355X only; move the model/input to a HIP device")
        with Fn.bn_counter_scope():          # the BatchNorm step counters of this forward: one launch on exit
            return self._forward(x, channels_last8)

    def _forward(self, x, channels_last8):
        if channels_last8:
            x8 = x                               # (B, T, F, 8) for up to 8 features, (B, T, F, 32) for more (MIC: 10)
        elif x.shape[1] <= 8:
            x8 = ops.nchw_to_nhwc8(x.contiguous().float())
        else:                                    # (plumbing copy) channels-last pixels padded to 32: the Winograd stem's input
            x8 = torch.nn.functional.pad(x.float().permute(0, 2, 3, 1), (0, 32 - x.shape[1])).contiguous()
        if x8.shape[-1] < self.in_channels or x8.shape[-1] not in (8, 32):
            raise RuntimeError("SEResnet34: %d input features need %d-channel pixels (got %d)"
                               % (self.in_channels, 8 if self.in_channels <= 8 else 32, x8.shape[-1]))
        first = self.layer1[0]
        holder = Fn.BlockLink() if (Fn.FUSE_STEM_AFFINE and not first.pool and first.downsample is None) else None
        y = Fn.StemFn.apply(x8, self.conv1.weight, self.conv1.bias, self.bn1.weight, self.bn1.bias, self.bn1,
                            self.training, holder)
        stem_affine = holder.affine if holder is not None else None     # the stem's BatchNorm is applied by its consumer
        link = None                              # BlockLink between consecutive blocks (functional.FUSE_SEBWD)
        blocks = [blk for li in range(1, 5) for blk in getattr(self, "layer%d" % li)]
        packs = None
        if ops.conv_algo() in ("winograd", "winograd4"):     # every block filter packed by ONE launch (two with winograd4)
            packs = self._packs.refresh([w for blk in blocks for w in (blk.conv1.weight, blk.conv2.weight)],
                                        frozen=not self.training)
        for bi, blk in enumerate(blocks):
            nxt = Fn.BlockLink()
            pk = (packs.get(2 * bi) + packs.get(2 * bi + 1)) if packs is not None else None
            y = blk(y, link_in=link, link_out=nxt, in_affine=stem_affine,
                    stem_holder=holder if stem_affine is not None else None, packs=pk,
                    pool_next=bi + 1 < len(blocks) and bool(blocks[bi + 1].pool))
            stem_affine = None
            link = nxt
        y = Fn.SAPFn.apply(y, self.attention.W.weight, self.attention.W.bias)
        save = self.training and torch.is_grad_enabled()
        y = Fn.BiGRULayerFn.apply(y, *self.lstm.layer_params(0), save)
        y = self._dropout(y)
        y = Fn.BiGRULayerFn.apply(y, *self.lstm.layer_params(1), save)
        return Fn.LNTanhFn.apply(y, self.norm.weight, self.norm.bias, self.norm.eps)
