"""csrc/fused_block.hip: conv3 + bn3 + residual + ReLU of a bottleneck and the next block's conv1 without y3
(torchvision Bottleneck.forward via stylenet/model.py:15-18,24), against the same arithmetic in float64.
 * the BatchNorm statistics of y3 come from the Gram matrix of conv3's INPUT: (scale, shift), running statistics;
 * the fused kernel: out, y1 and y1's statistics partials; ragged tiles, a BatchNorm on the identity branch, prescales."""
import pytest
import torch

import capnet  # noqa: F401
from capnet import ops
from helpers import rel_err

pytestmark = pytest.mark.gpu


def _case(M, MID, ds, seed, in_scale=1.0):
    g = torch.Generator().manual_seed(seed)
    C = 4 * MID
    y2 = torch.randn(M, MID, generator=g) * torch.exp(0.5 * torch.randn(M, MID, generator=g)) * in_scale
    s2 = (torch.rand(MID, generator=g) + 0.5) / in_scale
    t2 = torch.randn(MID, generator=g) * 0.5
    w3 = torch.randn(C, MID, 1, 1, generator=g) * (2.0 / MID) ** 0.5
    w1 = torch.randn(MID, C, 1, 1, generator=g) * (2.0 / C) ** 0.5
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.3
    res = torch.randn(M, C, generator=g)
    sd, td = (torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)) if ds else (None, None)
    return y2, s2, t2, w3, w1, gamma, beta, res, sd, td


def _reference(y2, s2, t2, w3, w1, gamma, beta, res, sd, td, eps=1e-5):
    D = torch.float64
    a2 = torch.relu(y2.to(D) * s2.to(D) + t2.to(D))
    y3 = a2 @ w3.reshape(w3.shape[0], -1).to(D).t()
    mean, var = y3.mean(0), y3.var(0, unbiased=False)
    scale = gamma.to(D) / torch.sqrt(var + eps)
    shift = beta.to(D) - mean * scale
    idn = res.to(D) if sd is None else res.to(D) * sd.to(D) + td.to(D)
    out = torch.relu(y3 * scale + shift + idn)
    y1 = out @ w1.reshape(w1.shape[0], -1).to(D).t()
    return mean, var, scale, shift, out, y1


@pytest.mark.parametrize("M,MID,ds", [(588, 256, False), (12544, 256, False), (1000, 256, True), (2352, 128, False),
                                      (777, 128, True), (9408, 64, False), (100, 64, True), (64, 256, False), (49, 64, False)])
def test_statistics_from_the_gram_matrix_and_the_fused_kernel(dev, M, MID, ds):
    y2, s2, t2, w3, w1, gamma, beta, res, sd, td = _case(M, MID, ds, 7 * M + MID)
    mean, var, scale, shift, out_r, y1_r = _reference(y2, s2, t2, w3, w1, gamma, beta, res, sd, td)
    d = lambda t: None if t is None else t.to(dev)
    y2d, s2d, t2d, resd = d(y2), d(s2), d(t2), d(res)
    img3, img1 = ops.pack_fused_block_weight(d(w3), 0), ops.pack_fused_block_weight(d(w1), 1)
    rm, rv = torch.zeros(4 * MID, device=dev), torch.ones(4 * MID, device=dev)
    sc, sh = ops.fused_block_stats(y2d, s2d, t2d, img3, d(gamma), d(beta), rm, rv, momentum=0.1)
    # the statistics: as close to float64 as bn_finalize's double reduction of fp32 partial sums is
    assert rel_err(sc, scale) < 2e-6, rel_err(sc, scale)
    assert (sh.double().cpu() - shift).abs().max().item() < 4e-6 * max(1.0, shift.abs().max().item())
    assert rel_err(rm, 0.1 * mean) < 2e-6
    unb = var * (M / max(M - 1, 1))
    assert rel_err(rv, 0.9 + 0.1 * unb) < 2e-6
    out, y1, ps, pq = ops.fused_block_forward(y2d, s2d, t2d, img3, sc, sh, resd, img1, d(sd), d(td))
    # with the SAME (scale, shift) the kernel was given
    idn = res.double() if sd is None else res.double() * sd.double() + td.double()
    a2 = torch.relu(y2.double() * s2.double() + t2.double())
    y3 = a2 @ w3.reshape(4 * MID, MID).double().t()
    out_same = torch.relu(y3 * sc.double().cpu() + sh.double().cpu() + idn)
    assert rel_err(out, out_same) < 3e-6, rel_err(out, out_same)
    y1_same = out.double().cpu() @ w1.reshape(MID, 4 * MID).double().t()
    assert rel_err(y1, y1_same) < 3e-6, rel_err(y1, y1_same)
    assert rel_err(ps.sum(0), y1_same.sum(0)) < 1e-5
    assert rel_err(pq.sum(0), (y1_same ** 2).sum(0)) < 1e-5
    # and end to end against the all-float64 block
    assert rel_err(out, out_r) < 1e-5 and rel_err(y1, y1_r) < 1e-5
    ops.check_device_errors()


@pytest.mark.parametrize("e3,e1,in_scale", [(8, 3, 2.0 ** -8), (-6, 2, 2.0 ** 6), (12, -4, 2.0 ** -12)])
def test_prescales_keep_fp32_grade_results(dev, e3, e1, in_scale):
    """conv3's input at another scale with the matching power-of-two prescale: the same relative bounds."""
    M, MID = 700, 128
    y2, s2, t2, w3, w1, gamma, beta, res, sd, td = _case(M, MID, False, 99)
    # a2 = relu(y2 s2 + t2) scaled by in_scale: scale s2 and t2
    s2s, t2s = s2 * in_scale, t2 * in_scale
    _, _, scale, shift, out_r, y1_r = _reference(y2, s2s, t2s, w3, w1, gamma, beta, res, None, None, eps=1e-5 * in_scale ** 2)
    d = lambda t: t.to(dev)
    img3, img1 = ops.pack_fused_block_weight(d(w3), 0), ops.pack_fused_block_weight(d(w1), 1)
    sc, sh = ops.fused_block_stats(d(y2), d(s2s), d(t2s), img3, d(gamma), d(beta), eps=1e-5 * in_scale ** 2, in_exp=e3)
    assert rel_err(sc, scale) < 2e-6
    out, y1, _, _ = ops.fused_block_forward(d(y2), d(s2s), d(t2s), img3, sc, sh, d(res), img1, e3=e3, e1=e1)
    assert rel_err(out, out_r) < 1e-5 and rel_err(y1, y1_r) < 1e-5
    ops.check_device_errors()


@pytest.mark.parametrize("env", [{"CAPNET_FB_WIDE": "0", "CAPNET_FB_NW": "4", "CAPNET_FB_RS": "2"}, {"CAPNET_FB_WIDE": "0", "CAPNET_FB_NW": "4", "CAPNET_FB_RS": "1"},
                                 {"CAPNET_FB_WIDE": "0"}, {"CAPNET_FB_WIDE": "1"}])
@pytest.mark.parametrize("M,MID,ds", [(588, 256, True), (2352, 128, False), (1000, 64, False)])
def test_the_other_tile_shapes_are_working_configurations(dev, monkeypatch, env, M, MID, ds):
    """CAPNET_FB_WIDE (read when the weight images are packed and at launch: 32-row strips on 32x32x16 MFMAs, opt-in,
    or the default 16-row strips on 16x16x32 ones) and CAPNET_FB_NW / CAPNET_FB_RS (read at launch: the 16-row form
    with four waves of two or one strips instead of eight waves of one): the A/B arms of DESIGN 4m stay correct."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    y2, s2, t2, w3, w1, gamma, beta, res, sd, td = _case(M, MID, ds, 3 * M + MID)
    _, _, _, _, out_r, y1_r = _reference(y2, s2, t2, w3, w1, gamma, beta, res, sd, td)
    d = lambda t: None if t is None else t.to(dev)
    img3, img1 = ops.pack_fused_block_weight(d(w3), 0), ops.pack_fused_block_weight(d(w1), 1)
    sc, sh = ops.fused_block_stats(d(y2), d(s2), d(t2), img3, d(gamma), d(beta))
    out, y1, ps, pq = ops.fused_block_forward(d(y2), d(s2), d(t2), img3, sc, sh, d(res), img1, d(sd), d(td))
    assert rel_err(out, out_r) < 1e-5 and rel_err(y1, y1_r) < 1e-5
    assert rel_err(ps.sum(0), y1_r.sum(0)) < 2e-5
    ops.check_device_errors()


def test_fused_kernel_is_exact_beside_other_work(dev):
    """The kernel's counted waits and its two groups of waves one phase apart must not depend on how fast its DMAs land:
    the same launch alone and beside a stream that keeps the chip's memory system busy, bit for bit (a wait that is too
    lenient by one group of DMAs passed every isolated test in round 4 and produced non-finite values in the pipelined
    step)."""
    M, MID = 12544, 256
    y2, s2, t2, w3, w1, gamma, beta, res, sd, td = _case(M, MID, False, 77)
    d = lambda t: None if t is None else t.to(dev)
    y2d, s2d, t2d, resd = d(y2), d(s2), d(t2), d(res)
    img3, img1 = ops.pack_fused_block_weight(d(w3), 0), ops.pack_fused_block_weight(d(w1), 1)
    sc, sh = ops.fused_block_stats(y2d, s2d, t2d, img3, d(gamma), d(beta))
    out0, y10, _, _ = ops.fused_block_forward(y2d, s2d, t2d, img3, sc, sh, resd, img1)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    big = torch.randn(64 * 1024 * 1024, device=dev)
    worst = 0
    for rep in range(6):
        with torch.cuda.stream(side):
            for _ in range(8):
                big = big * 1.0001 + 0.5                      # 512 MB of traffic per launch beside the kernel
        out, y1, _, _ = ops.fused_block_forward(y2d, s2d, t2d, img3, sc, sh, resd, img1)
        torch.cuda.synchronize()
        worst = max(worst, int((out != out0).sum().item()) + int((y1 != y10).sum().item()))
    assert worst == 0
    ops.check_device_errors()


def test_inference_form_flags_non_finite_outputs(dev):
    M, MID = 300, 64
    y2, s2, t2, w3, w1, gamma, beta, res, sd, td = _case(M, MID, False, 5)
    d = lambda t: t.to(dev)
    img3, img1 = ops.pack_fused_block_weight(d(w3), 0), ops.pack_fused_block_weight(d(w1), 1)
    sc, sh = torch.ones(4 * MID, device=dev), torch.zeros(4 * MID, device=dev)
    out, y1, ps, pq = ops.fused_block_forward(d(y2), d(s2), d(t2), img3, sc, sh, d(res), img1, stats=False)
    assert ps is None and torch.isfinite(y1).all()
    ops.check_device_errors()
    bad = d(res).clone()
    bad[17, 3] = float("inf")
    ops.fused_block_forward(d(y2), d(s2), d(t2), img3, sc, sh, bad, img1, stats=False)
    with pytest.raises(capnet.CapnetError, match="non-finite"):
        ops.check_device_errors()
