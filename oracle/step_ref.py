"""CPU restatement of one training step. TEST INFRASTRUCTURE.

  train_step  <- train_factual / train_emotion, stylenet/train_multitask.py:373-389,527-537
  clip_gradient (element-wise clamp) <- stylenet/utils.py:51-60
  Adam <- torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8), stylenet/train_multitask.py:166-167
"""
import torch
import torch.nn.functional as Fn

from . import decoders_ref as D


def clip_gradient_(grads, clip):
    for g in grads:
        if g is not None:
            g.clamp_(-clip, clip)


class AdamRef:
    """torch.optim.Adam semantics (single tensor path), parameters held as a name->tensor dict.
    Parameters whose gradient is None are skipped and keep their step count (torch >= 2.0
    zero_grad(set_to_none=True) behaviour)."""

    def __init__(self, lr, betas=(0.9, 0.999), eps=1e-8):
        self.lr, self.betas, self.eps = lr, betas, eps
        self.state = {}

    def step(self, params, grads):
        b1, b2 = self.betas
        for name, p in params.items():
            g = grads.get(name)
            if g is None:
                continue
            st = self.state.setdefault(name, {"step": 0, "m": torch.zeros_like(p), "v": torch.zeros_like(p)})
            st["step"] += 1
            st["m"].lerp_(g, 1 - b1)
            st["v"].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** st["step"]
            bc2 = 1 - b2 ** st["step"]
            denom = (st["v"].sqrt() / (bc2 ** 0.5)).add_(self.eps)
            p.addcdiv_(st["m"], denom, value=-(self.lr / bc1))


def decoder_loss_and_grads(forward, p, captions, lengths, features, tf_mask, **kw):
    """loss = CrossEntropyLoss(decoder(captions, lengths, features), packed targets); returns
    (loss, {name: grad}, dfeatures)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    feat = None
    if features is not None:
        feat = features.detach().clone().requires_grad_(True)
    logits = forward(leaves, captions, lengths, feat, tf_mask, **kw)
    loss = Fn.cross_entropy(logits, D.packed_targets(captions, lengths))
    loss.backward()
    grads = {k: v.grad for k, v in leaves.items()}
    return loss.detach(), grads, (feat.grad if feat is not None else None), logits.detach()
