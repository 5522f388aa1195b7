"""Host-side helpers of the training loop (stylenet/utils.py:51-60,93-140)."""
import torch

from . import _lib, ops
from ._lib import CapnetError, check, current_stream, ptr
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


def save_checkpoint(folder, data_name, mode, epoch, epochs_since_improvement, encoder, decoder,
                    optimizer, lang_optimizer, bleu4, is_best):
    """stylenet/utils.py:63-90, same arguments and file names
    (`<mode>_checkpoint_<data_name>.pth.tar`, plus `<mode>_BEST_checkpoint_...` when is_best).
    The reference pickles the module and optimiser OBJECTS; this writes their state_dicts (keys
    identical to the reference's classes, tests/test_checkpoint_cpu.py), which load with
    `torch.load(..., weights_only=True)` and into the reference's own classes."""
    state = {
        'epoch': epoch,
        'epochs_since_improvement': epochs_since_improvement,
        'bleu-4': bleu4,
        'encoder': encoder.state_dict(),
        'decoder': decoder.state_dict(),
        'optimizer': optimizer.state_dict() if optimizer is not None else None,
        'lang_optimizer': lang_optimizer.state_dict() if lang_optimizer is not None else None,
    }
    filename = folder + '/' + mode + '_checkpoint_' + data_name + '.pth.tar'
    torch.save(state, filename)
    if is_best:
        filename = folder + '/' + mode + '_BEST_checkpoint_' + data_name + '.pth.tar'
        torch.save(state, filename)


def load_checkpoint(path, encoder=None, decoder=None, optimizer=None, lang_optimizer=None,
                    map_location="cpu"):
    """Restore a checkpoint written by save_checkpoint; returns its scalar fields."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    if encoder is not None:
        encoder.load_state_dict(state['encoder'])
    if decoder is not None:
        decoder.load_state_dict(state['decoder'])
    if optimizer is not None and state.get('optimizer') is not None:
        optimizer.load_state_dict(state['optimizer'])
    if lang_optimizer is not None and state.get('lang_optimizer') is not None:
        lang_optimizer.load_state_dict(state['lang_optimizer'])
    return {k: state[k] for k in ('epoch', 'epochs_since_improvement', 'bleu-4')}


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


def accuracy(scores, targets, k):
    """Top-k accuracy in percent (stylenet/utils.py:127-140): one kernel counts the rows whose
    target is among the k largest scores; the `.item()` is the same sync the reference has."""
    if not scores.is_cuda or scores.dtype != torch.float32 or targets.dtype != torch.int64:
        raise CapnetError("accuracy: scores must be CUDA float32 and targets int64")
    scores = scores.detach()
    scores = scores if scores.is_contiguous() else scores.contiguous()
    targets = targets.reshape(-1).contiguous()
    batch_size = targets.size(0)
    count = torch.empty(1, dtype=torch.int32, device=scores.device)
    check(_lib.lib().capnet_topk_correct(ptr(scores), scores.shape[1], batch_size, scores.shape[1],
                                         ptr(targets), int(k), ptr(count),
                                         ptr(ops.err_flag(scores.device)), current_stream()),
          "capnet_topk_correct")
    return count.item() * (100.0 / batch_size)
