"""Host-side helpers of the training loop (stylenet/utils.py:51-60,93-140)."""
import torch

from . import _lib
from ._lib import check, current_stream, ptr
from .optim import Adam


def clip_gradient(optimizer, grad_clip):
    """Element-wise clamp of every gradient to [-grad_clip, grad_clip] (stylenet/utils.py:51-60).
    With capnet.optim.Adam the clamp is fused into the next step(); with any other optimiser
    it runs now, as a HIP kernel per tensor."""
    if isinstance(optimizer, Adam):
        optimizer.set_pending_clip(grad_clip)
        return
    for group in optimizer.param_groups:
        for param in group['params']:
            if param.grad is not None:
                g = param.grad
                if not g.is_contiguous():
                    param.grad = g = g.contiguous()
                check(_lib.lib().capnet_clamp(ptr(g), g.numel(), -grad_clip, grad_clip,
                                              current_stream()), "capnet_clamp")


class AverageMeter(object):
    """stylenet/utils.py:93-111."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def adjust_learning_rate(optimizer, shrink_factor):
    """stylenet/utils.py:114-124."""
    for param_group in optimizer.param_groups:
        param_group['lr'] = param_group['lr'] * shrink_factor
