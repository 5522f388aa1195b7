"""Input pipeline pieces next to the hot path.

* collate_fn / collate_fn_styled: the batch contract of stylenet/data_loader.py:116-160 (sort by
  caption length, stack images, zero-pad captions to LongTensor [B, T], lengths as a Python list).
* The torchvision transform chain of stylenet/train_multitask.py:62-69 on the GPU:
  Resize((336, 336)) -> RandomCrop(224) -> RandomHorizontalFlip() -> ToTensor() -> Normalize().
  `gpu_resize` is Pillow's antialiased two-pass resample (bit-identical to
  PIL.Image.resize(BILINEAR), which is what torchvision 0.2.2's Resize calls); `GpuTransform`
  draws the crop offsets and the flip from Python's `random` in torchvision 0.2.2's order
  (RandomCrop.get_params: randint for the row, randint for the column; RandomHorizontalFlip:
  random() < p), so it consumes the global RNG the way the reference's loader does.
* TransformCache: the class-level `__cache` of FlickrDataset (data_loader.py:11,57-62): the
  transformed image is computed once per file name and reused by every later epoch (random
  crop/flip included), here kept resident in HBM.
No CPU fallback: the image work runs through libcapnet_hip.so.
"""
import math
import random

import torch

from . import _lib
from ._lib import CapnetError, check, current_stream, ptr

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
_PRECISION_BITS = 32 - 8 - 2


def _bilinear(x):
    x = -x if x < 0.0 else x
    return 1.0 - x if x < 1.0 else 0.0


def resample_tables(in_size, out_size):
    """Filter tables of one resample pass, as Pillow's precompute_coeffs + normalize_coeffs_8bpc
    build them (libImaging/Resample.c; bilinear filter, support 1): for every output coordinate
    the first input tap, the number of taps, and the taps' weights in 22-bit fixed point.
    Returns (bounds [out, 2] int32, coef [out, kmax] int32, kmax)."""
    scale = float(in_size) / float(out_size)
    filterscale = scale if scale > 1.0 else 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = torch.zeros((out_size, 2), dtype=torch.int32)
    coef = torch.zeros((out_size, ksize), dtype=torch.int32)
    ss = 1.0 / filterscale
    one = float(1 << _PRECISION_BITS)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for w in k:
            ww += w
        if ww != 0.0:
            k = [w / ww for w in k]
        bounds[xx, 0], bounds[xx, 1] = xmin, xmax
        for x, w in enumerate(k):
            coef[xx, x] = int(-0.5 + w * one) if w < 0 else int(0.5 + w * one)
    return bounds, coef, ksize


_table_cache = {}


def _tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _table_cache.get(key)
    if t is None:
        b, c, k = resample_tables(in_size, out_size)
        t = (b.to(device), c.to(device), k)
        _table_cache[key] = t
    return t


def gpu_resize(image, size):
    """image: uint8 CUDA tensor [H, W, 3]; size: (out_h, out_w). Returns uint8 [out_h, out_w, 3],
    bit-identical to PIL.Image.fromarray(image).resize((out_w, out_h), PIL.Image.BILINEAR)."""
    if not image.is_cuda or image.dtype != torch.uint8 or image.dim() != 3 or image.shape[2] != 3:
        raise CapnetError("gpu_resize: expected a CUDA uint8 [H, W, 3] image")
    image = image.contiguous()
    Hs, Ws = int(image.shape[0]), int(image.shape[1])
    Ho, Wo = int(size[0]), int(size[1])
    bh, ch, kh = _tables(Ws, Wo, image.device)
    bv, cv, kv = _tables(Hs, Ho, image.device)
    tmp = torch.empty((Hs, Wo, 3), dtype=torch.uint8, device=image.device)
    out = torch.empty((Ho, Wo, 3), dtype=torch.uint8, device=image.device)
    check(_lib.lib().capnet_resize_u8(ptr(image), Hs, Ws, ptr(tmp), ptr(out), Ho, Wo, ptr(bh), ptr(ch),
                                      kh, ptr(bv), ptr(cv), kv, current_stream()), "capnet_resize_u8")
    return out


def crop_flip_normalize(images, params, crop, mean=IMAGENET_MEAN, std=IMAGENET_STD):
    """images: uint8 CUDA [B, H, W, 3]; params: int [B, 3] rows (top, left, flip); crop: (h, w).
    Returns fp32 [B, 3, h, w] = Normalize(ToTensor(flip(crop(image))))."""
    if not images.is_cuda or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
        raise CapnetError("crop_flip_normalize: expected a CUDA uint8 [B, H, W, 3] batch")
    images = images.contiguous()
    B, Hs, Ws = int(images.shape[0]), int(images.shape[1]), int(images.shape[2])
    params = torch.as_tensor(params, dtype=torch.int32).reshape(B, 3)
    Hc, Wc = int(crop[0]), int(crop[1])
    for top, left, _ in params.tolist():
        if top < 0 or left < 0 or top + Hc > Hs or left + Wc > Ws:
            raise CapnetError("crop_flip_normalize: crop window outside the image")
    params = params.to(images.device)
    out = torch.empty((B, 3, Hc, Wc), dtype=torch.float32, device=images.device)
    C = _lib.C
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    check(_lib.lib().capnet_crop_flip_normalize(ptr(images), B, Hs, Ws, ptr(params), ptr(out), Hc, Wc,
                                                m, s, current_stream()), "capnet_crop_flip_normalize")
    return out


def draw_crop_flip(h, w, th, tw, p=0.5):
    """(top, left, flip) from Python's `random`, in torchvision 0.2.2's order: RandomCrop.get_params
    (no draw when the sizes match) and then RandomHorizontalFlip."""
    if w == tw and h == th:
        top, left = 0, 0
    else:
        top = random.randint(0, h - th)
        left = random.randint(0, w - tw)
    flip = 1 if random.random() < p else 0
    return top, left, flip


class GpuTransform(object):
    """The transform chain of stylenet/train_multitask.py:62-69 for a list of uint8 [H, W, 3]
    images (any sizes) -> fp32 [B, 3, crop, crop] on `device`."""

    def __init__(self, device, resize=(336, 336), crop_size=224, mean=IMAGENET_MEAN,
                 std=IMAGENET_STD, flip_p=0.5):
        self.device, self.resize, self.crop_size = torch.device(device), tuple(resize), int(crop_size)
        self.mean, self.std, self.flip_p = mean, std, flip_p

    def __call__(self, images):
        resized, params = [], []
        for im in images:
            im = torch.as_tensor(im)
            resized.append(gpu_resize(im.to(self.device), self.resize))
            params.append(draw_crop_flip(self.resize[0], self.resize[1], self.crop_size, self.crop_size,
                                         self.flip_p))
        batch = torch.stack(resized, 0)
        return crop_flip_normalize(batch, params, (self.crop_size, self.crop_size), self.mean, self.std)


class TransformCache(object):
    """FlickrDataset.__cache (data_loader.py:11,57-62): transformed images by file name, computed
    on first use and then frozen (the random crop / flip of the first epoch included)."""

    def __init__(self, transform):
        self.transform, self._cache = transform, {}

    def get(self, name, load_u8):
        """load_u8(): uint8 [H, W, 3] array of the decoded file, only called on a miss."""
        t = self._cache.get(name)
        if t is None:
            t = self.transform([load_u8()])[0]
            self._cache[name] = t
        return t

    def __len__(self):
        return len(self._cache)


def collate_fn(data):
    """List of (image [3, S, S], caption 1-D tensor, all_captions) -> (images [B, 3, S, S],
    targets LongTensor [B, T] zero padded, lengths list, all_captions tuple), sorted by caption
    length, longest first (stylenet/data_loader.py:116-145)."""
    data.sort(key=lambda item: len(item[1]), reverse=True)
    images, captions, all_captions = zip(*data)
    images = torch.stack(images, 0)
    lengths = [len(c) for c in captions]
    targets = torch.zeros(len(captions), max(lengths)).long()
    for row, c in enumerate(captions):
        targets[row, :lengths[row]] = c[:lengths[row]]
    return images, targets, lengths, all_captions


def collate_fn_styled(captions):
    """List of caption tensors -> (targets [B, T], lengths), longest first
    (stylenet/data_loader.py:148-160)."""
    captions.sort(key=lambda c: len(c), reverse=True)
    lengths = [len(c) for c in captions]
    targets = torch.zeros(len(captions), max(lengths)).long()
    for row, c in enumerate(captions):
        targets[row, :lengths[row]] = c[:lengths[row]]
    return targets, lengths
