"""GPU parity of the input pipeline kernels (bit-exact: integer/byte work and two fp32 roundings)."""
import random

import numpy as np
import pytest
import torch

import capnet
from capnet import data
from helpers import load_golden
from oracle import image_ref as R

pytestmark = pytest.mark.gpu

Z = load_golden("image_tiny.npz")


@pytest.mark.parametrize("name", [str(n) for n in Z["resize_cases"]])
def test_gpu_resize_matches_pillow_fixture(dev, name):
    src, want = Z["resize.%s.in" % name], Z["resize.%s.out" % name]
    got = data.gpu_resize(torch.from_numpy(src).to(dev), want.shape[:2])
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize("h,w", [(375, 500), (500, 333), (120, 90), (336, 336), (1024, 768)])
def test_gpu_resize_to_336_matches_oracle(dev, h, w):
    a = (np.random.RandomState(h + w).rand(h, w, 3) * 255).astype(np.uint8)
    got = data.gpu_resize(torch.from_numpy(a).to(dev), (336, 336)).cpu().numpy()
    assert np.array_equal(got, R.resize_bilinear_u8(a, 336, 336))
    try:
        from PIL import Image
    except ImportError:
        return
    assert np.array_equal(got, np.asarray(Image.fromarray(a).resize((336, 336), Image.BILINEAR)))


def test_crop_flip_normalize_is_bit_exact(dev):
    img = Z["resize.down.out"]
    batch = torch.from_numpy(np.stack([img, img])).to(dev)
    params = [Z["norm.a.params"].tolist(), Z["norm.b.params"].tolist()]
    out = data.crop_flip_normalize(batch, params, (32, 32)).cpu()
    assert torch.equal(out[0], torch.from_numpy(Z["norm.a.out"]))
    assert torch.equal(out[1], torch.from_numpy(Z["norm.b.out"]))
    with pytest.raises(capnet.CapnetError):
        data.crop_flip_normalize(batch, [[40, 0, 0], [0, 0, 0]], (32, 32))


def test_transform_chain_and_cache(dev):
    rs = np.random.RandomState(11)
    imgs = [(rs.rand(h, w, 3) * 255).astype(np.uint8) for h, w in [(300, 400), (500, 375), (336, 336)]]
    tf = data.GpuTransform(dev)
    random.seed(21)
    out = tf(imgs).cpu()
    assert out.shape == (3, 3, 224, 224) and out.dtype == torch.float32
    random.seed(21)
    for i, im in enumerate(imgs):
        top, left, flip = data.draw_crop_flip(336, 336, 224, 224)
        want = R.crop_flip_normalize(R.resize_bilinear_u8(im, 336, 336), top, left, flip, 224, 224,
                                     data.IMAGENET_MEAN, data.IMAGENET_STD)
        assert torch.equal(out[i], want)
    cache = data.TransformCache(tf)
    calls = []

    def load():
        calls.append(1)
        return imgs[0]
    a = cache.get("x.jpg", load)
    b = cache.get("x.jpg", load)
    assert len(calls) == 1 and a is b and len(cache) == 1     # frozen after the first epoch
