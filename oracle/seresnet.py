"""Oracle (test infrastructure): SE-ResNet34 + SAP + BiGRU encoder and AD-YOLO head,
PyTorch-CPU float32, functional over a reference-shaped ``state_dict``.

Restates ``/root/reference/src/models/backbones/resnet.py:7-47`` (SEBasicBlock),
``:91-106`` (SELayer), ``:109-123`` (SelfAttentionPooling), ``:126-199``
(SEResnet34) and ``/root/reference/src/models/linearheads.py:88-104`` (ADYOLOhead).
Pinned by ``tests/golden/encoder.npz`` / ``head.npz`` (outputs of the real
reference modules under the name-seeded weight filler, ``oracle/filler.py``).

Quirks kept on purpose: ReLU *before* BatchNorm after the
stem conv and after each block's conv1; AvgPool2d(2,2) applied inside the first
block of layer2/layer3 before the residual is taken; the attribute called
``lstm`` is a 2-layer bidirectional GRU; the head is two Linears with no
non-linearity in between.
"""
import torch
import torch.nn.functional as F

LAYERS = (3, 4, 6, 3)
WIDTHS = (32, 64, 128, 256)
POOLS = (None, (2, 2), (2, 2), None)
BN_EPS = 1e-5
BN_MOM = 0.1


def _bn(sd, prefix, x, training, update_stats):
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if training and not update_stats:
        rm, rv = rm.clone(), rv.clone()
    return F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training=training, momentum=BN_MOM, eps=BN_EPS)


def se_layer(sd, prefix, x):
    """resnet.py:102-106."""
    y = x.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[prefix + ".fc.0.weight"], sd[prefix + ".fc.0.bias"]))
    y = torch.sigmoid(F.linear(y, sd[prefix + ".fc.2.weight"], sd[prefix + ".fc.2.bias"]))
    return x * y[:, :, None, None]


def se_basic_block(sd, prefix, x, pool, training, update_stats=False):
    """resnet.py:25-47."""
    if pool is not None:
        x = F.avg_pool2d(x, kernel_size=pool, stride=pool)
    out = F.conv2d(x, sd[prefix + ".conv1.weight"], None, stride=1, padding=1)
    out = _bn(sd, prefix + ".bn1", F.relu(out), training, update_stats)
    out = F.conv2d(out, sd[prefix + ".conv2.weight"], None, stride=1, padding=1)
    out = _bn(sd, prefix + ".bn2", out, training, update_stats)
    out = se_layer(sd, prefix + ".se", out)
    if (prefix + ".downsample.0.weight") in sd:
        res = F.conv2d(x, sd[prefix + ".downsample.0.weight"], None)
        res = _bn(sd, prefix + ".downsample.1", res, training, update_stats)
    else:
        res = x
    return F.relu(out + res)


def self_attention_pooling(sd, x):
    """resnet.py:115-123.  x (B,T,F,C) -> (B,T,C)."""
    attn = F.linear(x, sd["attention.W.weight"], sd["attention.W.bias"]).squeeze(-1)
    attn = F.softmax(attn, dim=-1).unsqueeze(-1)
    return (x * attn).sum(dim=2)


def gru_cell_steps(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """Explicit single-direction GRU (gate order r,z,n; torch.nn.GRU semantics) for cross-checks."""
    b, t, _ = x.shape
    h = x.new_zeros(b, w_hh.shape[1])
    hs = [None] * t
    order = range(t - 1, -1, -1) if reverse else range(t)
    for i in order:
        gx = F.linear(x[:, i], w_ih, b_ih)
        gh = F.linear(h, w_hh, b_hh)
        xr, xz, xn = gx.chunk(3, dim=1)
        hr, hz, hn = gh.chunk(3, dim=1)
        r = torch.sigmoid(xr + hr)
        z = torch.sigmoid(xz + hz)
        n = torch.tanh(xn + r * hn)
        h = (1.0 - z) * n + z * h
        hs[i] = h
    return torch.stack(hs, dim=1)


def _bigru_layer(sd, layer, x):
    names = ["lstm.weight_ih_l%d", "lstm.weight_hh_l%d", "lstm.bias_ih_l%d", "lstm.bias_hh_l%d"]
    flat = [sd[n % layer] for n in names] + [sd[(n % layer) + "_reverse"] for n in names]
    h0 = x.new_zeros(2, x.shape[0], flat[1].shape[1])
    out, _ = torch._VF.gru(x, h0, flat, True, 1, 0.0, False, True, True)
    return out


def bigru(sd, x, dropout_mask=None):
    """resnet.py:153,195: 2-layer bidirectional GRU(256 -> 2x128), inter-layer dropout 0.3.

    ``dropout_mask`` (B,T,256), already scaled by 1/(1-p), multiplies the layer-0 output
    (torch applies dropout between layers in training mode only); None = no dropout.
    """
    y = _bigru_layer(sd, 0, x)
    if dropout_mask is not None:
        y = y * dropout_mask
    return _bigru_layer(sd, 1, y)


def encoder_forward(sd, x, training=False, update_stats=False, dropout_mask=None, taps=None):
    """resnet.py:180-199.  x (B,7,T,F) float32 -> (B, T//4, 256).  ``sd`` keys have no 'encoder.' prefix."""
    out = F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], stride=1, padding=1)
    out = _bn(sd, "bn1", F.relu(out), training, update_stats)
    if taps is not None:
        taps["stem"] = out
    for li, (nblk, pool) in enumerate(zip(LAYERS, POOLS), start=1):
        for bi in range(nblk):
            out = se_basic_block(sd, "layer%d.%d" % (li, bi), out, pool if bi == 0 else None,
                                 training, update_stats)
        if taps is not None:
            taps["layer%d" % li] = out
    out = self_attention_pooling(sd, out.permute(0, 2, 3, 1))
    if taps is not None:
        taps["sap"] = out
    out = bigru(sd, out, dropout_mask)
    if taps is not None:
        taps["gru"] = out
    out = F.layer_norm(out, (out.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-5)
    return torch.tanh(out)


def adyolo_head(sd, x):
    """linearheads.py:101-104.  ``sd`` keys have no 'head.' prefix."""
    x = F.linear(x, sd["yolo_head.0.weight"], sd["yolo_head.0.bias"])
    return F.linear(x, sd["yolo_head.1.weight"], sd["yolo_head.1.bias"])


def split_state_dict(sd):
    """WrapperModel state_dict ('encoder.*', 'head.*', wrapper.py:26-47) -> (encoder sd, head sd)."""
    enc = {k[len("encoder."):]: v for k, v in sd.items() if k.startswith("encoder.")}
    head = {k[len("head."):]: v for k, v in sd.items() if k.startswith("head.")}
    return enc, head


def model_forward(sd, x, training=False, update_stats=False, dropout_mask=None):
    """wrapper.py:52-57 for --encoder se-resnet34 --loss adyolo."""
    enc, head = split_state_dict(sd)
    return adyolo_head(head, encoder_forward(enc, x, training, update_stats, dropout_mask))


def state_dict_spec(nb_classes=12, in_ch=7, grid=(8, 4), anchors=5):
    """Names/shapes of the reference WrapperModel state_dict (se-resnet34 + adyolo); 301 + 4 keys."""
    spec = []

    def bn(p, c):
        spec.extend([(p + ".weight", (c,)), (p + ".bias", (c,)), (p + ".running_mean", (c,)),
                     (p + ".running_var", (c,)), (p + ".num_batches_tracked", ())])

    spec.append(("encoder.conv1.weight", (WIDTHS[0], in_ch, 3, 3)))
    spec.append(("encoder.conv1.bias", (WIDTHS[0],)))
    bn("encoder.bn1", WIDTHS[0])
    inpl = WIDTHS[0]
    for li, (nblk, c) in enumerate(zip(LAYERS, WIDTHS), start=1):
        for bi in range(nblk):
            p = "encoder.layer%d.%d" % (li, bi)
            spec.append((p + ".conv1.weight", (c, inpl if bi == 0 else c, 3, 3)))
            bn(p + ".bn1", c)
            spec.append((p + ".conv2.weight", (c, c, 3, 3)))
            bn(p + ".bn2", c)
            spec.extend([(p + ".se.fc.0.weight", (c // 8, c)), (p + ".se.fc.0.bias", (c // 8,)),
                         (p + ".se.fc.2.weight", (c, c // 8)), (p + ".se.fc.2.bias", (c,))])
            if bi == 0 and inpl != c:
                spec.append((p + ".downsample.0.weight", (c, inpl, 1, 1)))
                bn(p + ".downsample.1", c)
        inpl = c
    spec.extend([("encoder.attention.W.weight", (1, 256)), ("encoder.attention.W.bias", (1,))])
    for layer in range(2):
        for sfx in ("", "_reverse"):
            spec.extend([("encoder.lstm.weight_ih_l%d%s" % (layer, sfx), (384, 256)),
                         ("encoder.lstm.weight_hh_l%d%s" % (layer, sfx), (384, 128)),
                         ("encoder.lstm.bias_ih_l%d%s" % (layer, sfx), (384,)),
                         ("encoder.lstm.bias_hh_l%d%s" % (layer, sfx), (384,))])
    spec.extend([("encoder.norm.weight", (256,)), ("encoder.norm.bias", (256,))])
    k = grid[0] * grid[1] * anchors * (nb_classes + 3)
    spec.extend([("head.yolo_head.0.weight", (256, 256)), ("head.yolo_head.0.bias", (256,)),
                 ("head.yolo_head.1.weight", (k, 256)), ("head.yolo_head.1.bias", (k,))])
    return spec
