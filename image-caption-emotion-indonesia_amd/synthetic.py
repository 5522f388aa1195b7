"""Synthetic inputs and deterministic parameter tensors for benchmarks and parity tests.

Follows SURVEY.md 8(d): the reference ships no dataset, vocabulary or weights, so every run
uses seeded CPU generators that produce the same tensors on any machine with this torch build.
Output contract of the batch = stylenet/data_loader.py:116-145 (lengths sorted decreasing,
0-padded int64 captions) with the special ids of stylenet/build_vocab.py:53-56
(<pad>=0, <start>=1, <end>=2, <unk>=3).
"""
import math
import zlib

import torch

IMAGENET_MEAN = (0.485, 0.456, 0.406)   # stylenet/train_multitask.py:68
IMAGENET_STD = (0.229, 0.224, 0.225)


def make_batch(batch, vocab_size, seed=0, image_size=224, min_len=8, max_len=24, images=True):
    """(images [B,3,S,S] float32 or None, captions [B,T] int64, lengths list[int])."""
    g = torch.Generator().manual_seed(seed)
    imgs = None
    if images:
        imgs = torch.rand(batch, 3, image_size, image_size, generator=g)
        mean = torch.tensor(IMAGENET_MEAN).view(1, 3, 1, 1)
        std = torch.tensor(IMAGENET_STD).view(1, 3, 1, 1)
        imgs = (imgs - mean) / std
    else:
        # the same draws, discarded: captions and lengths of a seed must not depend on whether the images were asked for
        # (bench.py rebuilds every rank's lengths with images=False)
        for _ in range(batch):
            torch.rand(3, image_size, image_size, generator=g)
    lengths = torch.randint(min_len, max_len + 1, (batch,), generator=g)
    lengths = sorted(lengths.tolist(), reverse=True)
    T = lengths[0]
    captions = torch.randint(4, vocab_size, (batch, T), generator=g)
    captions[:, 0] = 1
    for i, l in enumerate(lengths):
        captions[i, l - 1] = 2
        captions[i, l:] = 0
    return imgs, captions, lengths


def param_tensor(name, shape, seed=1234, kind=None, bias_range=0.0):
    """Deterministic tensor for a parameter called `name`.

    kind: 'xavier' (>=2-D default: U(+-sqrt(6/(fan_in+fan_out))), stylenet/model.py:99-105),
          'embed'  (U(-0.1, 0.1), model.py:111-113), 'kaiming_out' (conv: N(0, sqrt(2/fan_out))),
          'ones' / 'zeros', 'bias' (U(+-bias_range); 0 reproduces the reference's zero biases).
    """
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7FFFFFFF)
    shape = tuple(shape)
    if kind is None:
        kind = 'xavier' if len(shape) >= 2 else 'bias'
    if kind == 'xavier':
        rf = 1
        for s in shape[2:]:
            rf *= s
        bound = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if kind == 'embed':
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.1
    if kind == 'kaiming_out':
        fan_out = shape[0] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
    if kind == 'ones':
        return torch.ones(shape)
    if kind == 'zeros':
        return torch.zeros(shape)
    if kind == 'bias':
        if bias_range == 0.0:
            return torch.zeros(shape)
        return (torch.rand(shape, generator=g) * 2 - 1) * bias_range
    raise ValueError(kind)


def decoder_state(state_dict_like, seed=1234, bias_range=0.0):
    """name -> tensor for every floating-point entry of a decoder state_dict (shapes only are
    read from `state_dict_like`). Embedding / output projection weights use U(-0.1, 0.1)."""
    out = {}
    for name, t in state_dict_like.items():
        if not torch.is_floating_point(t):
            continue
        if name in ("B.weight", "C.weight", "embed.weight", "linear.weight"):
            kind = 'embed'
        else:
            kind = None
        out[name] = param_tensor(name, t.shape, seed, kind, bias_range)
    return out


def trunk_state(state_dict_like, seed=1234):
    """Deterministic ResNet-152 trunk weights: kaiming-normal(fan_out) convolutions, BN weight 1,
    bias 0, fresh running stats (torchvision's own initialisation)."""
    out = {}
    for name, t in state_dict_like.items():
        if name.endswith("num_batches_tracked"):
            out[name] = torch.zeros_like(t)
        elif name.endswith("running_mean"):
            out[name] = torch.zeros_like(t)
        elif name.endswith("running_var"):
            out[name] = torch.ones_like(t)
        elif t.dim() == 4:
            out[name] = param_tensor(name, t.shape, seed, 'kaiming_out')
        elif name.endswith(".weight"):
            out[name] = torch.ones_like(t)
        else:
            out[name] = torch.zeros_like(t)
    return out


def encoder_state(state_dict_like, seed=1234):
    """Deterministic EncoderCNN state (stylenet/model.py:11-27): trunk_state for `resnet.*`, a xavier Linear with small
    biases and a BatchNorm1d with non-trivial weight / bias, fresh running statistics."""
    sd = state_dict_like
    new = trunk_state({k: v for k, v in sd.items() if k.startswith("resnet.")}, seed=seed)
    new["linear.weight"] = param_tensor("linear.weight", sd["linear.weight"].shape, seed, "xavier")
    new["linear.bias"] = param_tensor("linear.bias", sd["linear.bias"].shape, seed, "bias", 0.05)
    new["bn.weight"] = param_tensor("bn.weight", sd["bn.weight"].shape, seed, "bias", 0.5) + 1.0
    new["bn.bias"] = param_tensor("bn.bias", sd["bn.bias"].shape, seed, "bias", 0.2)
    for k in ("bn.running_mean", "bn.running_var", "bn.num_batches_tracked"):
        new[k] = sd[k].clone()
    return new
