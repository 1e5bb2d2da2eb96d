"""Plugin surface: name-keyed dispatch of encoder / head / loss.  Mirror of /root/reference/src/wrapper.py
(WrapperModel :10-57, WrapperCriterion :62-88): same constructor arguments, same attribute names
(``encoder``, ``head`` -> state_dict prefixes), same NotImplementedError behaviour for unknown names.
Both encoders (``se-resnet34``, ``resnet-conformer``) and every ``--loss`` of the reference's CLI (adyolo, adpit,
accdoa, seddoa, masked-seddoa) run on the gfx950 kernels; there is no CPU / eager fallback."""
import torch.nn as nn

from . import ops

from .models.backbones.resnet import SEResnet34
from .models.backbones.resnet_conformer import ResnetConformer
from .models.linearheads import ADYOLOhead, ACCDOAhead, ADPIThead, SEDDOAhead
from .models.loss import ADYOLOloss, SEDDOAloss, ACCDOAloss, ADPITloss


class WrapperModel(nn.Module):
    def __init__(self, in_feat_shape, out_shape, params: dict):
        super().__init__()
        self.nb_classes = params["data_config"]["nb_classes"]
        self.encoder_nm = params["args"]["encoder"]
        self.loss_nm = params["args"]["loss"]
        if self.encoder_nm == "se-resnet34":
            self.encoder = SEResnet34(in_feat_shape, out_shape, params)
        elif self.encoder_nm == "resnet-conformer":
            self.encoder = ResnetConformer(in_feat_shape, out_shape, params)
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
        # evaluation-mode caches (BatchNorm affines, packed filters, recorded forward graphs) expire on a state load; code that
        # writes parameters by hand in evaluation mode calls ops.params_changed() itself
        self.register_load_state_dict_post_hook(lambda module, incompatible_keys: ops.params_changed())

    def forward(self, x, channels_last8=False):
        """x : (B, C, T, F)  ->  (B, T//4, K)"""
        return self.head(self.encoder(x, channels_last8=channels_last8))


class WrapperCriterion(object):
    def __init__(self, params):
        self.nb_classes = params["data_config"]["nb_classes"]
        self.loss_nm = params["args"]["loss"]
        if self.loss_nm == "adyolo":
            self.loss = ADYOLOloss(params)
        elif self.loss_nm == "seddoa":
            self.loss = SEDDOAloss(self.nb_classes, masked_mse=False)
        elif self.loss_nm == "masked-seddoa":
            self.loss = SEDDOAloss(self.nb_classes, masked_mse=True)
        elif self.loss_nm == "accdoa":
            self.loss = ACCDOAloss(self.nb_classes)
        elif self.loss_nm == "adpit":
            self.loss = ADPITloss(self.nb_classes)
        else:
            raise NotImplementedError("loss: {}".format(self.loss_nm))

    def __call__(self, output, target):
        return self.loss(output, target)
