"""Output heads.  Mirror of /root/reference/src/models/linearheads.py (ADYOLOhead :88-104, and the
ACCDOA / ADPIT / SEDDOA heads :26-86 so the ``wrapper.py`` dispatch surface is complete); Linear layers
run on the fp32-MFMA GEMM kernel.  State-dict keys: ``yolo_head.{0,1}.{weight,bias}`` etc."""
import math

import torch
import torch.nn as nn

from .. import functional as Fn


class _Linear(nn.Module):
    """Xavier-uniform weight, zero bias (reference init_head, linearheads.py:5-11) after the default init."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.uniform_(self.bias, -1.0 / math.sqrt(cin), 1.0 / math.sqrt(cin))

    def forward(self, x):
        return Fn.LinearFn.apply(x, self.weight, self.bias)


def _init_head(seq):
    for layer in seq:
        nn.init.xavier_uniform_(layer.weight)
        layer.bias.data.fill_(0.0)


def _mlp(cin, ffn, cout):
    return nn.Sequential(_Linear(cin, ffn), _Linear(ffn, cout))


class ADYOLOhead(nn.Module):
    def __init__(self, enc_out_dim, ffn_dim, nb_classes, grid_size, nb_anchors):
        super().__init__()
        self.nb_classes = nb_classes
        self.nb_grids = (math.ceil(360 / grid_size[0]), math.ceil(180 / grid_size[1]))
        self.nb_anchors = nb_anchors
        self.yolo_head = _mlp(enc_out_dim, ffn_dim, self.nb_grids[0] * self.nb_grids[1] * nb_anchors * (nb_classes + 3))
        _init_head(self.yolo_head)

    def forward(self, x):
        return self.yolo_head(x)          # raw logits, no activation (linearheads.py:101-104)


class ACCDOAhead(nn.Module):
    def __init__(self, enc_out_dim, ffn_dim, nb_classes):
        super().__init__()
        self.accdoa_head = _mlp(enc_out_dim, ffn_dim, 3 * nb_classes)
        _init_head(self.accdoa_head)

    def forward(self, x):
        return Fn.ActFn.apply(self.accdoa_head(x), 0)


class ADPIThead(nn.Module):
    def __init__(self, enc_out_dim, ffn_dim, nb_classes, n_tracks=3):
        super().__init__()
        self.adpit_head = _mlp(enc_out_dim, ffn_dim, n_tracks * 3 * nb_classes)
        _init_head(self.adpit_head)

    def forward(self, x):
        return Fn.ActFn.apply(self.adpit_head(x), 0)


class SEDDOAhead(nn.Module):
    def __init__(self, enc_out_dim, ffn_dim, nb_classes):
        super().__init__()
        self.sed_head = _mlp(enc_out_dim, ffn_dim, nb_classes)
        self.doa_head = _mlp(enc_out_dim, ffn_dim, 3 * nb_classes)
        _init_head(self.sed_head)
        _init_head(self.doa_head)

    def forward(self, x):
        # cat is a copy (plumbing); the activations run in one HIP pass: sigmoid on the SED columns, tanh on the DOA ones
        raw = torch.cat([self.sed_head(x), self.doa_head(x)], dim=-1)
        return Fn.ActFn.apply(raw, self.sed_head[1].weight.shape[0])
