"""Input-pipeline host logic and the image oracle (no GPU): Pillow-produced resize fixtures,
filter tables, RNG draw order, collate contract."""
import random

import numpy as np
import pytest
import torch

import capnet
from capnet import data
from helpers import load_golden
from oracle import image_ref as R

Z = load_golden("image_tiny.npz")


@pytest.mark.parametrize("name", [str(n) for n in Z["resize_cases"]])
def test_oracle_resize_matches_pillow_fixture(name):
    src, want = Z["resize.%s.in" % name], Z["resize.%s.out" % name]
    got = R.resize_bilinear_u8(src, want.shape[0], want.shape[1])
    assert np.array_equal(got, want)


def test_oracle_resize_matches_pillow_live():
    Image = pytest.importorskip("PIL.Image")
    rs = np.random.RandomState(3)
    for h, w, oh, ow in [(375, 500, 336, 336), (120, 90, 336, 336), (64, 1000, 20, 33)]:
        a = (rs.rand(h, w, 3) * 255).astype(np.uint8)
        want = np.asarray(Image.fromarray(a).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(R.resize_bilinear_u8(a, oh, ow), want)


@pytest.mark.parametrize("n_in,n_out", [(500, 336), (90, 336), (336, 336), (7, 3), (3, 7)])
def test_resample_tables_equal_the_oracle_tables(n_in, n_out):
    bounds, coef, kmax = data.resample_tables(n_in, n_out)
    ob, ok = R.precompute_coeffs(n_in, n_out)
    assert bounds.tolist() == [list(b) for b in ob]
    assert coef.tolist() == ok and kmax == len(ok[0])
    # weights are a partition of unity in 22-bit fixed point (up to rounding of each tap)
    assert all(abs(sum(row) - (1 << 22)) <= len(row) for row in ok)


def test_oracle_normalize_matches_torch_fixture():
    img = Z["resize.down.out"]
    for tag in ("a", "b"):
        top, left, flip = Z["norm.%s.params" % tag].tolist()
        got = R.crop_flip_normalize(img, top, left, flip, 32, 32, data.IMAGENET_MEAN, data.IMAGENET_STD)
        assert torch.equal(got, torch.from_numpy(Z["norm.%s.out" % tag]))


def test_draw_order_follows_torchvision():
    """RandomCrop.get_params draws the row, then the column; RandomHorizontalFlip one uniform."""
    random.seed(5)
    want_top, want_left = random.randint(0, 336 - 224), random.randint(0, 336 - 224)
    want_flip = 1 if random.random() < 0.5 else 0
    after = random.random()
    random.seed(5)
    assert data.draw_crop_flip(336, 336, 224, 224) == (want_top, want_left, want_flip)
    assert random.random() == after
    random.seed(5)
    assert data.draw_crop_flip(224, 224, 224, 224)[:2] == (0, 0)     # no crop draw when sizes match


def test_collate_fn_contract():
    caps = [torch.tensor([1., 5., 2.]), torch.tensor([1., 7., 8., 9., 2.]), torch.tensor([1., 2.])]
    items = [(torch.full((3, 4, 4), float(i)), c, [c]) for i, c in enumerate(caps)]
    images, targets, lengths, all_caps = data.collate_fn(list(items))
    assert lengths == [5, 3, 2] and targets.dtype == torch.int64
    assert targets.tolist() == [[1, 7, 8, 9, 2], [1, 5, 2, 0, 0], [1, 2, 0, 0, 0]]
    assert images.shape == (3, 3, 4, 4) and images[:, 0, 0, 0].tolist() == [1.0, 0.0, 2.0]
    assert len(all_caps) == 3
    t2, l2 = data.collate_fn_styled([c.clone() for c in caps])
    assert l2 == [5, 3, 2] and t2.tolist() == targets.tolist()


def test_image_ops_refuse_cpu_tensors():
    with pytest.raises(capnet.CapnetError):
        data.gpu_resize(torch.zeros(4, 4, 3, dtype=torch.uint8), (2, 2))
    with pytest.raises(capnet.CapnetError):
        data.crop_flip_normalize(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), [[0, 0, 0]], (2, 2))
