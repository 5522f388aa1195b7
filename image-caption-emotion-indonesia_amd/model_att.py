"""Attention-path encoder on the MI355X kernels (stylenet/model_att.py:11-29).

EncoderCNN(encoded_image_size=14): ResNet-152 children[:-2] under no_grad, AdaptiveAvgPool2d
to 14x14 (an exact 2x replication of the 7x7 map), permuted to NHWC. The trunk already produces
NHWC, so the permute costs nothing here. The attention decoder is not built yet.
"""
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import CapnetError, check, current_stream, ptr
from .model import _Marker, _TrunkRunner, _resnet152_children


class EncoderCNN(nn.Module):

    def __init__(self, encoded_image_size=14):
        super(EncoderCNN, self).__init__()
        self.resnet = _resnet152_children(with_avgpool=False)
        self.adaptive_pool = _Marker()   # parameter-less, as nn.AdaptiveAvgPool2d
        self.encoded_image_size = encoded_image_size
        self._runner = [None]

    def _trunk(self):
        if self._runner[0] is None:
            self._runner[0] = _TrunkRunner(self.resnet)
        return self._runner[0]

    def forward(self, images):
        with torch.no_grad():
            _, fmap = self._trunk().forward(images, self.training, False, True)
            b, side = fmap.shape[0], fmap.shape[1]
            out_side = self.encoded_image_size
            if out_side == side:
                return fmap
            if out_side % side != 0:
                raise CapnetError("encoded_image_size %d must be a multiple of the trunk's %d"
                                  % (out_side, side))
            out = torch.empty((b, out_side, out_side, 2048), dtype=torch.float32, device=fmap.device)
            check(_lib.lib().capnet_adaptive_pool_replicate(ptr(fmap), ptr(out), b, side, out_side,
                                                            2048, current_stream()),
                  "capnet_adaptive_pool_replicate")
        return out
