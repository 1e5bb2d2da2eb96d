"""Oracle (test infrastructure): ResNet-Conformer encoder, PyTorch-CPU float32, functional over a reference-shaped
state_dict.  Restates /root/reference/src/models/backbones/resnet_conformer.py: MultiHeadAttention :25-85,
ConformerConvModule :154-178, FeedForwardModule :181-212, ConformerBlock :215-282, PoolingModule :285-297,
ResnetConformer :342-447.  The residual blocks of the front end are torchvision==0.11 ``BasicBlock``s
(README.md:20; not vendored in /root/reference): their published definition (conv3x3(stride) -> BN -> ReLU -> conv3x3 ->
BN, + identity or downsample(x), ReLU) is restated here -- PARITY UNPINNED for that block beyond the stub used to
generate tests/golden/conformer.npz.  Dropout (p = 0.2) is omitted: goldens are generated with dropout disabled.
"""
import torch
import torch.nn.functional as F

LAYER_BLOCKS = (3, 4, 5, 3)          # layer3 has 5 blocks, not 6 (resnet_conformer.py:373-384)
LAYER_WIDTHS = (64, 128, 256, 512)


def _bn(sd, p, x, training):
    rm, rv = sd[p + ".running_mean"].clone(), sd[p + ".running_var"].clone()
    return F.batch_norm(x, rm, rv, sd[p + ".weight"], sd[p + ".bias"], training=training, momentum=0.1, eps=1e-5)


def basic_block(sd, p, x, stride, training):
    out = F.conv2d(x, sd[p + ".conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(_bn(sd, p + ".bn1", out, training))
    out = _bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], None, padding=1), training)
    if (p + ".downsample.0.weight") in sd:
        idn = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride=stride), training)
    else:
        idn = x
    return F.relu(out + idn)


def _ln(sd, p, x):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], 1e-5)


def _swish(x):
    return x * torch.sigmoid(x)


def feed_forward(sd, p, x):
    y = _ln(sd, p + ".sequential.0", x)
    y = _swish(F.linear(y, sd[p + ".sequential.1.weight"], sd[p + ".sequential.1.bias"]))
    return F.linear(y, sd[p + ".sequential.4.weight"], sd[p + ".sequential.4.bias"])


def attention(sd, p, x, heads=4):
    b, t, e = x.shape
    d = e // heads
    q, k, v = (F.linear(x, sd[p + "." + n + ".weight"], sd[p + "." + n + ".bias"]).view(b, t, heads, d).transpose(1, 2)
               for n in ("query", "key", "value"))
    w = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
    ctx = (w @ v).transpose(1, 2).reshape(b, t, e)
    return F.linear(ctx, sd[p + ".linear.weight"], sd[p + ".linear.bias"])


def conv_module(sd, p, x, dilation, training):
    y = _ln(sd, p + ".conv.0", x).transpose(1, 2)
    y = F.conv1d(y, sd[p + ".conv.2.weight"], sd[p + ".conv.2.bias"])
    y = F.glu(_bn(sd, p + ".conv.3", y, training), dim=1)
    y = F.conv1d(y, sd[p + ".conv.5.weight"], sd[p + ".conv.5.bias"], padding=dilation, dilation=dilation, groups=y.shape[1])
    y = _swish(_bn(sd, p + ".conv.6", y, training))
    return F.conv1d(y, sd[p + ".conv.8.weight"], sd[p + ".conv.8.bias"]).transpose(1, 2)


def conformer_block(sd, p, x, dilation, training):
    x = feed_forward(sd, p + ".sequential.0.module", x) * 0.5 + x
    x = attention(sd, p + ".sequential.1.module.1", _ln(sd, p + ".sequential.1.module.0", x)) * 0.5 + x
    x = conv_module(sd, p + ".sequential.2.module", x, dilation, training) + x
    x = feed_forward(sd, p + ".sequential.3.module", x) * 0.5 + x
    return _ln(sd, p + ".sequential.4", x)


def encoder_forward(sd, x, training=False, taps=None):
    """resnet_conformer.py:419-447.  x (B,7,T,64) -> (B, T//4, 256)."""
    y = F.conv2d(x, sd["conv1.weight"], None, stride=(1, 2), padding=3)
    y = _bn(sd, "bn1", F.relu(y), training)
    y = F.max_pool2d(y, 3, stride=(1, 2), padding=1)
    if taps is not None:
        taps["stem"] = y
    for li, nblk in enumerate(LAYER_BLOCKS, start=1):
        for bi in range(nblk):
            y = basic_block(sd, "layer%d.%d" % (li, bi), y, (1, 2) if bi == 0 else 1, training)
    if taps is not None:
        taps["layer4"] = y
    y = y.permute(0, 2, 1, 3).squeeze(-1)
    y = F.linear(y, sd["bottleneck.weight"])
    for i in range(8):
        y = conformer_block(sd, "conformer.encoder_module.%d" % i, y, 2 ** i, training)
        if taps is not None and i == 0:
            taps["block0"] = y
    y = y.transpose(1, 2)
    y = (F.avg_pool1d(y, 4) + F.avg_pool1d(y, 4)).transpose(1, 2)
    return _ln(sd, "t_pooling.norm", y)
