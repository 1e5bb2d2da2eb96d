"""Plugin surface: name-keyed dispatch of encoder / head / loss.  Mirror of /root/reference/src/wrapper.py
(WrapperModel :10-57, WrapperCriterion :62-88): same constructor arguments, same attribute names
(``encoder``, ``head`` -> state_dict prefixes), same NotImplementedError behaviour for unknown names.
``--encoder se-resnet34 --loss adyolo`` is the MI355X hot path; the remaining names of the reference's
CLI (resnet-conformer, seddoa/masked-seddoa/accdoa/adpit losses) are outside SURVEY.md section 8's first rows
and raise NotImplementedError naming what is missing rather than silently falling back."""
import torch.nn as nn

from .models.backbones.resnet import SEResnet34
from .models.linearheads import ADYOLOhead, ACCDOAhead, ADPIThead, SEDDOAhead
from .models.loss import ADYOLOloss


class WrapperModel(nn.Module):
    def __init__(self, in_feat_shape, out_shape, params: dict):
        super().__init__()
        self.nb_classes = params["data_config"]["nb_classes"]
        self.encoder_nm = params["args"]["encoder"]
        self.loss_nm = params["args"]["loss"]
        if self.encoder_nm == "se-resnet34":
            self.encoder = SEResnet34(in_feat_shape, out_shape, params)
        elif self.encoder_nm == "resnet-conformer":
            raise NotImplementedError("encoder: resnet-conformer is not built yet on the gfx950 path (SURVEY 8a24)")
        else:
            raise NotImplementedError("encoder: {}".format(self.encoder_nm))
        d = self.encoder.enc_out_dim
        if self.loss_nm in ("seddoa", "masked-seddoa"):
            self.head = SEDDOAhead(d, d, self.nb_classes)
        elif self.loss_nm == "accdoa":
            self.head = ACCDOAhead(d, d, self.nb_classes)
        elif self.loss_nm == "adpit":
            self.head = ADPIThead(d, d, self.nb_classes)
        elif self.loss_nm == "adyolo":
            self.grid_size = params["train_config"]["grid_size"]
            self.nb_anchors = params["train_config"]["nb_anchors"]
            self.head = ADYOLOhead(d, d, self.nb_classes, self.grid_size, self.nb_anchors)
        else:
            raise NotImplementedError("head: {}".format(self.loss_nm))

    def forward(self, x, channels_last8=False):
        """x : (B, C, T, F)  ->  (B, T//4, K)"""
        return self.head(self.encoder(x, channels_last8=channels_last8))


class WrapperCriterion(object):
    def __init__(self, params):
        self.nb_classes = params["data_config"]["nb_classes"]
        self.loss_nm = params["args"]["loss"]
        if self.loss_nm == "adyolo":
            self.loss = ADYOLOloss(params)
        elif self.loss_nm in ("seddoa", "masked-seddoa", "accdoa", "adpit"):
            raise NotImplementedError("loss: {} is not built yet on the gfx950 path (SURVEY 8a25 / section 2 #15)"
                                      .format(self.loss_nm))
        else:
            raise NotImplementedError("loss: {}".format(self.loss_nm))

    def __call__(self, output, target):
        return self.loss(output, target)
