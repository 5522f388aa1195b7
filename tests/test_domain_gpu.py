"""The split-f16 convolutions have the DOMAIN of the fp32 arithmetic they stand in for.

An fp32 operand is split into two f16 pieces; unscaled, an activation above 65 504 becomes inf and a tensor far below 1
loses its low pieces to f16's subnormals (absolute 2^-25, not relative). Every kernel therefore takes a per-tensor power
of two (in_exp): the input is multiplied by it on its way into the f16 planes -- folded into the BatchNorm's scale /
shift where there is one -- and the accumulators are scaled back, both exactly. The trunk derives the exponents from the
BatchNorm parameters (capnet.model._TrunkRunner._input_exponents: rigorous bounds in train mode), and what no bound can
cover (inference with running statistics far from the data; a genuine fp32 overflow) raises bit 3 of the device error
word instead of propagating inf silently.

Here: the three kernels at input scales 2^-20 ... 2^15 with the exponent the rule gives, held to the SAME relative bound
as at scale 1; without the exponent the error at the small scales is shown to be what the fix removes; the flag."""
import math

import pytest
import torch

import capnet
from capnet import ops, synthetic
from capnet._lib import check, current_stream, lib, ptr
from helpers import rel_err

pytestmark = pytest.mark.gpu
SCALES = [-20, -10, 0, 10, 15]


def _exp_for(amax):
    """the trunk's rule: max |x| 2^e in [2^14, 2^15)"""
    return max(-40, min(40, 15 - math.frexp(float(amax))[1]))


def _rms(y, ref):
    return (((y.double().cpu() - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt()).item()


@pytest.mark.parametrize("lg", SCALES)
@pytest.mark.parametrize("pre", [False, True])
def test_conv1x1_f16x3_relative_accuracy_at_every_input_scale(dev, lg, pre):
    B, H, W, Cin, Cout, bn = 3, 7, 7, 128, 128, 128
    g = torch.Generator().manual_seed(lg + 100)
    x = torch.randn(B, H, W, Cin, generator=g) * torch.exp(torch.randn(B, H, W, Cin, generator=g)) * 2.0 ** lg
    w = torch.randn(Cout, Cin, generator=g) * 0.1
    M = B * H * W
    if pre:
        scale = (torch.rand(Cin, generator=g) + 0.2) * 2.0 ** -lg      # a BatchNorm that brings the tensor back to O(1) ...
        shift = torch.randn(Cin, generator=g)
        xin = torch.relu(x * scale + shift)
        post_scale = 2.0 ** lg                                          # ... times a gamma of the tensor's scale
        scale, shift, xin = scale * post_scale, shift * post_scale, xin * post_scale
    else:
        scale = shift = None
        xin = x
    ref = xin.reshape(M, Cin).double() @ w.double().t()
    e = _exp_for(xin.abs().max())
    L = lib()
    xd, wd = x.to(dev), w.to(dev)
    img = torch.empty(L.capnet_conv1x1_f16x3_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    check(L.capnet_conv1x1_f16x3_pack(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
    t = L.capnet_conv1x1_tiles_m(M)
    sd, hd = (scale.to(dev), shift.to(dev)) if pre else (None, None)

    def run(in_exp):
        y = torch.full((M, Cout), float("nan"), device=dev)
        ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
        check(L.capnet_conv2d_fwd_f16x3_scaled(ptr(xd), H * W * Cin, W * Cin, Cin, ptr(img), bn, ptr(y), ptr(sd), ptr(hd), int(pre),
                                               ptr(ps), ptr(pq), B, H, W, Cin, Cout, 1, 1, 0, in_exp, current_stream()))
        return y
    y = run(e)
    err, rms = rel_err(y, ref), _rms(y, ref)
    y0 = run(0)
    rms0 = _rms(y0, ref) if torch.isfinite(y0).all() else float("inf")
    print("scale 2^%d (exp %d): max-norm %.2e rms %.2e | without the prescale: rms %.2e" % (lg, e, err, rms, rms0))
    assert err < 3e-6 and rms < 6e-7            # the bounds of tests/test_kernels_gpu.py at scale 1
    if lg <= -10:
        assert rms0 > 20 * rms                   # what the prescale removes: residuals lost to f16's subnormals
    if lg == 15:
        assert not torch.isfinite(y0).all()      # ... and values beyond 65 504 that became inf


@pytest.mark.parametrize("lg", SCALES)
def test_conv3x3_patch_relative_accuracy_at_every_input_scale(dev, lg):
    B, H, W, C, bn = 2, 14, 14, 64, 64
    g = torch.Generator().manual_seed(lg + 200)
    x = torch.randn(B, C, H, W, generator=g) * 2.0 ** lg
    w = torch.randn(C, C, 3, 3, generator=g) * 0.05
    scale = torch.rand(C, generator=g) + 0.5
    shift = torch.randn(C, generator=g) * 2.0 ** lg
    xin = torch.relu(x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))
    ref = torch.nn.functional.conv2d(xin.double(), w.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, C)
    e = _exp_for(xin.abs().max())
    L = lib()
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    img = ops.pack_conv_weight_f16x3(w.to(dev), bn)
    M = B * H * W
    t = L.capnet_conv1x1_tiles_m(M)
    y = torch.full((M, C), float("nan"), device=dev)
    ps, pq = torch.empty(t, C, device=dev), torch.empty(t, C, device=dev)
    sd, hd = scale.to(dev), shift.to(dev)
    check(L.capnet_conv3x3_fwd_patch_scaled(ptr(xd), ptr(img), bn, ptr(y), ptr(sd), ptr(hd), 1, ptr(ps), ptr(pq), B, H, W, C, C, 1, e,
                                            current_stream()))
    print("scale 2^%d (exp %d): max-norm %.2e rms %.2e" % (lg, e, rel_err(y, ref), _rms(y, ref)))
    assert rel_err(y, ref) < 4e-6 and _rms(y, ref) < 7e-7


@pytest.mark.parametrize("lg", SCALES)
def test_conv1x1_tail_relative_accuracy_at_every_input_scale(dev, lg):
    M, Cin, Cout, bn = 300, 256, 128, 128
    g = torch.Generator().manual_seed(lg + 300)
    s = 2.0 ** lg
    y3 = torch.randn(M, Cin, generator=g)
    res = torch.randn(M, Cin, generator=g).abs() * s
    s1, t1 = (torch.rand(Cin, generator=g) + 0.5) * s, torch.randn(Cin, generator=g) * s
    w = torch.randn(Cout, Cin, generator=g) * 0.05
    tail = torch.relu(y3 * s1 + t1 + res)                              # fp32, as the kernel forms it
    ref = tail.double() @ w.double().t()
    e = _exp_for(tail.abs().max())
    L = lib()
    img = torch.empty(L.capnet_conv1x1_f16x3_weight_words(Cin, Cout), dtype=torch.int32, device=dev)
    wd = w.to(dev)
    check(L.capnet_conv1x1_f16x3_pack(ptr(wd), ptr(img), Cout, Cin, bn, current_stream()))
    t = L.capnet_conv1x1_tiles_m(M)
    out = torch.full((M, Cin), float("nan"), device=dev)
    y = torch.full((M, Cout), float("nan"), device=dev)
    ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
    y3d, resd, s1d, t1d = y3.to(dev), res.to(dev), s1.to(dev), t1.to(dev)
    check(L.capnet_conv1x1_fwd_tail_scaled(ptr(y3d), ptr(s1d), ptr(t1d), ptr(resd), None, None, ptr(out), ptr(img), bn, ptr(y), ptr(ps),
                                           ptr(pq), M, Cin, Cout, e, current_stream()))
    assert rel_err(out, tail.double()) < 2e-7                          # the written tail is unscaled
    print("scale 2^%d (exp %d): max-norm %.2e rms %.2e" % (lg, e, rel_err(y, ref), _rms(y, ref)))
    assert rel_err(y, ref) < 3e-6 and _rms(y, ref) < 6e-7


def test_trunk_exponents_follow_the_batchnorm_parameters_and_overflow_is_flagged(dev, monkeypatch):
    """The whole trunk with the BatchNorm weights in front of conv2 / conv3 scaled by 2^12 and by 2^-12: the exponents
    move with them, and the features stay as close to the same network on the f32-MFMA kernels (CAPNET_NO_H3=1, which
    have fp32's range by construction) as they are at scale 1. Then an inference pass whose running statistics are
    absurd (a running mean of -1e8): activations beyond any bound -- the error word says so."""
    from capnet.model import EncoderCNN
    from test_encoder_gpu import _encoder_state
    imgs = synthetic.make_batch(4, 100, seed=5)[0].to(dev)

    def features(k, f32):
        if f32:
            monkeypatch.setenv("CAPNET_NO_H3", "1")
        else:
            monkeypatch.delenv("CAPNET_NO_H3", raising=False)
        enc = EncoderCNN(300)
        st = _encoder_state(enc)
        for name in list(st):
            if name.startswith("resnet.") and (name.endswith("bn1.weight") or name.endswith("bn2.weight")):
                st[name] = st[name] * 2.0 ** k             # conv2 / conv3 inputs at scale 2^k
        enc.load_state_dict(st)
        enc.to(dev).train()
        runner = enc._trunk()
        pooled, _ = runner.forward(imgs, True, True, False)
        plan = runner._plan(4, 224, 224, dev)
        return pooled, plan["exps"][2]
    errs = {}
    e0 = None
    for k in (0, 12, -12):
        ph, eh = features(k, False)
        pf, _ = features(k, True)
        errs[k] = rel_err(ph, pf)
        if k == 0:
            e0 = eh
        else:
            moved = [b - a for a, b in zip(e0, eh)]
            assert set(moved) <= {0, -k} and moved.count(-k) == 100      # the 50 conv2 and 50 conv3 inputs
    print("split-f16 trunk vs f32-MFMA trunk, BatchNorm weights x 2^k:", errs)
    # (train-mode BatchNorm over 4 images amplifies rounding differences: 1e-3 at scale 1, tests/test_encoder_gpu.py)
    assert errs[0] < 3e-3 and errs[12] < 3e-3 and errs[-12] < 3e-3
    assert max(errs[12], errs[-12]) < 3 * errs[0] + 1e-4
    ops.check_device_errors()
    # inference with absurd running statistics: post-BN values of 1e8, far beyond the bound the exponents allow for
    monkeypatch.delenv("CAPNET_NO_H3", raising=False)
    enc = EncoderCNN(300)
    st = _encoder_state(enc)
    st["resnet.4.0.bn1.running_mean"] = torch.full_like(st["resnet.4.0.bn1.running_mean"], -1e8)
    enc.load_state_dict(st)
    enc.to(dev).eval()
    with torch.no_grad():
        enc._trunk().forward(imgs, False, True, False)
    torch.cuda.synchronize()
    with pytest.raises(capnet.CapnetError, match="non-finite"):
        ops.check_device_errors()
