"""Edge cases of the decoder drivers against the CPU oracle: a single sample, single-step
captions, equal lengths, very ragged lengths, the step cap, a free-running first step
(the input is B(<start>) instead of the image feature), no-feature calls."""
import random

import pytest
import torch

import capnet
from capnet import ops, synthetic
from capnet.model import DecoderFactoredLSTM
from capnet.model_att import DecoderFactoredLSTMAtt
from capnet.nic_model import DecoderRNN
from helpers import rel_err
from oracle import decoders_ref as D
from oracle import step_ref as S

pytestmark = pytest.mark.gpu


def _captions(lengths, V, seed):
    g = torch.Generator().manual_seed(seed)
    B, T = len(lengths), max(lengths)
    c = torch.randint(4, V, (B, T), generator=g)
    c[:, 0] = 1
    for i, l in enumerate(lengths):
        c[i, l - 1] = 2 if l > 1 else 1
        c[i, l:] = 0
    return c


def _check(dec, p, forward, captions, lengths, feats, tf, dev, **kw):
    loss_r, grads_r, dfeat_r, logits_r = S.decoder_loss_and_grads(forward, p, captions, lengths, feats, tf, **kw)
    dec.zero_grad()
    f = feats.to(dev).requires_grad_(True) if feats is not None else None
    out = dec(captions.to(dev), lengths, f, tf_mask=tf, **kw)
    loss = ops.cross_entropy(out, D.packed_targets(captions, lengths).to(dev))
    loss.backward()
    ops.check_device_errors()
    assert out.shape == logits_r.shape
    assert rel_err(out, logits_r) < 5e-5
    assert abs(loss.item() - loss_r.item()) / abs(loss_r.item()) < 1e-5
    for k, prm in dec.named_parameters():
        if grads_r[k] is None:
            assert prm.grad is None, k
        else:
            assert rel_err(prm.grad, grads_r[k]) < 2e-4 or float(grads_r[k].abs().max()) < 1e-7, k
    if dfeat_r is not None and f is not None and float(dfeat_r.abs().max()) > 0:
        assert rel_err(f.grad, dfeat_r) < 2e-4


LENGTH_SETS = {
    "single_sample": [5],
    "single_step": [1, 1, 1],
    "one_long_many_short": [17, 2, 2, 1, 1, 1],
    "all_equal": [6, 6, 6, 6],
    "two_rows_len2": [2, 2],
}


@pytest.mark.parametrize("name", list(LENGTH_SETS))
@pytest.mark.parametrize("tf_kind", ["all", "none", "mixed"])
def test_factored_edge_lengths(dev, name, tf_kind):
    lengths = LENGTH_SETS[name]
    E, H, F, V = 20, 32, 24, 57
    dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=len(lengths), bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    captions = _captions(lengths, V, 3)
    feats = torch.randn(len(lengths), E, generator=torch.Generator().manual_seed(4))
    random.seed(9)
    T = max(lengths)
    tf = {"all": [True] * T, "none": [False] * T,
          "mixed": [random.random() < 0.5 for _ in range(T)]}[tf_kind]
    _check(dec, p, D.factored_lstm_forward, captions, lengths, feats, tf, dev, mode="factual")


def test_factored_without_features(dev):
    lengths = [7, 4, 2]
    E, H, F, V = 20, 32, 24, 57
    dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=2, bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    _check(dec, p, D.factored_lstm_forward, _captions(lengths, V, 5), lengths, None,
           [True, False, True, True, False, True, True], dev, mode="sad")


@pytest.mark.parametrize("name", ["single_sample", "single_step", "one_long_many_short"])
def test_nic_edge_lengths(dev, name):
    lengths = LENGTH_SETS[name]
    E, H, V = 20, 32, 57
    dec = DecoderRNN(E, H, V, 1, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=7, bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    feats = torch.randn(len(lengths), E, generator=torch.Generator().manual_seed(8))
    tf = [(i % 3) != 1 for i in range(max(lengths))]
    _check(dec, p, D.lstm_forward, _captions(lengths, V, 6), lengths, feats, tf, dev)


@pytest.mark.parametrize("lengths", [[5], [3, 2, 2], [9, 1]])
def test_attention_edge_lengths(dev, lengths):
    A, E, H, F, V, P, Cf = 16, 12, 16, 16, 37, 4, 512
    dec = DecoderFactoredLSTMAtt(A, E, H, F, V, 1, feature_size=Cf, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=11, bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    captions = _captions(lengths, V, 12)
    feats = torch.randn(len(lengths), P, Cf, generator=torch.Generator().manual_seed(13)).abs()
    tf = [(i % 2) == 0 for i in range(max(lengths))]
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    logits_r, alphas_r = D.factored_att_forward(pr, captions, lengths, feats, tf, mode="factual")
    loss_r = D.att_loss(logits_r, alphas_r, D.packed_targets(captions, lengths))
    loss_r.backward()
    dec.zero_grad()
    out, alphas = dec(captions.to(dev), lengths, feats.to(dev), tf_mask=tf, mode="factual")
    loss = ops.cross_entropy(out, D.packed_targets(captions, lengths).to(dev)) + \
        ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    loss.backward()
    assert rel_err(out, logits_r) < 5e-5 and rel_err(alphas, alphas_r) < 5e-5
    assert abs(loss.item() - loss_r.item()) / loss_r.item() < 1e-5
    for k, prm in dec.named_parameters():
        if pr[k].grad is not None:
            g = pr[k].grad
            assert (prm.grad.cpu().double() - g.double()).abs().max().item() <= 2e-4 * g.abs().max().item() + 1e-6, k


def test_step_cap_and_bad_lengths_raise(dev):
    E, H, F, V = 12, 16, 16, 37
    dec = DecoderFactoredLSTM(E, H, F, V, 1, dropout=0.0).to(dev).train()
    long_len = [200]
    with pytest.raises(capnet.CapnetError):
        dec(_captions(long_len, V, 1).to(dev), long_len, None)              # > 128 steps
    with pytest.raises((capnet.CapnetError, ValueError)):
        dec(_captions([3, 5], V, 1).to(dev), [3, 5], None)                   # not sorted
    with pytest.raises((capnet.CapnetError, ValueError)):
        dec(_captions([3, 2], V, 1).to(dev), [3, 0], None)                   # zero length
    bad = _captions([4, 3], V, 1)
    bad[0, 1] = V + 5                                                        # token id out of range
    dec(bad.to(dev), [4, 3], None, tf_mask=[True] * 4)
    with pytest.raises(capnet.CapnetError):
        ops.check_device_errors()


# ---- the device error word: dropped steps are taken out of Adam's step counts; the word crosses ranks ----
def test_dropped_steps_leave_parameters_moments_and_step_counts_consistent(dev):
    """While the error word is set capnet_clamp_adam drops its update; the host-side step counts (bias correction of
    torch.optim.Adam, train_multitask.py:389) must not run ahead of the moments. Two good steps, two dropped ones, the
    check (raises, rolls the counts back), one more good step == three good steps of an untouched twin."""
    from capnet.optim import Adam
    torch.manual_seed(0)
    w0 = torch.randn(300, 17)
    gs = [torch.randn(300, 17) * 0.1 for _ in range(5)]

    def run(fault):
        p = torch.nn.Parameter(w0.clone().to(dev))
        opt = Adam([p], lr=1e-2)
        for i, g in enumerate(gs):
            if fault and i in (2, 3):
                if i == 2:
                    ops.err_flag(dev).fill_(1)
            elif not fault and i in (2, 3):
                continue
            p.grad = g.clone().to(dev)
            opt.step()
            if fault and i == 3:
                before = p.detach().clone()
                with pytest.raises(capnet.CapnetError, match="2 optimizer steps"):
                    ops.check_device_errors()
                assert opt.state[p]["step"] == 2 and torch.equal(before, p.detach())
        ops.check_device_errors()
        return p.detach().cpu(), opt.state[p]["step"], opt.state[p]["exp_avg"].cpu()
    a, b = run(True), run(False)
    assert a[1] == b[1] == 3
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2])


def test_recurrent_weight_beyond_the_persistent_kernels_range_is_diagnosed(dev):
    """The persistent LSTM kernel's f16 weight image covers |w| < 64 (2^10 scale, lstm_persist.hip); a larger recurrent
    weight used to end in an undiagnosed NaN loss. The pack marks the image, the kernel raises bit 5 of the error word, and
    the launch-per-step path (f32 operands) takes the same weights."""
    from capnet._lib import lib
    E, H, V, B = 32, 512, 101, 6
    _, captions, lengths = synthetic.make_batch(B, V, seed=1, images=False, min_len=4, max_len=9)
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(2)).to(dev)

    def run(dec):
        random.seed(3)
        out = dec(captions.to(dev), lengths, feats, teacher_forcing_ratio=1.0)
        return out

    dec = DecoderFactoredLSTM(E, H, 32, V, 1, dropout=0.0).to(dev).train()
    if not lib().capnet_lstm_persist_supported(B, H):
        pytest.skip("persistent LSTM path not available on this device")
    with torch.no_grad():
        dec.W_i.weight[3, 5] = 100.0
    run(dec)
    with pytest.raises(capnet.CapnetError, match="beyond .w. < 64"):
        ops.check_device_errors()
    old = lib().capnet_lstm_persist_set_mode(1)
    try:
        out = run(dec)
        ops.check_device_errors()
        assert torch.isfinite(out).all()
    finally:
        lib().capnet_lstm_persist_set_mode(old)
    with torch.no_grad():
        dec.W_i.weight[3, 5] = 40.0                  # inside the range (round 3's documented limit of 32 was conservative)
    out2 = run(dec)
    ops.check_device_errors()
    assert torch.isfinite(out2).all()


def test_error_word_rides_behind_the_flat_gradient(dev):
    slot = torch.full((1,), 7.0, device=dev)
    err = ops.err_flag(dev)
    ops.err_word_exchange(slot, 0)
    assert slot.item() == 0.0 and err.item() == 0
    err.fill_(2)
    ops.err_word_exchange(slot, 0)
    assert slot.item() == 1.0
    err.zero_()
    slot.fill_(3.0)                    # "three ranks had a fault" after the SUM all-reduce
    ops.err_word_exchange(slot, 1)
    assert err.item() == 16
    with pytest.raises(capnet.CapnetError, match="another rank"):
        ops.check_device_errors()
    slot.zero_()
    ops.err_word_exchange(slot, 1)
    assert err.item() == 0
    ops.check_device_errors()


def test_native_rccl_allreduce_single_rank(dev):
    """capnet_comm_create / capnet_allreduce_grads (include/capnet.h): an RCCL communicator of one rank, the flat
    gradient's size, on a side stream; a one-rank SUM leaves the buffer as it is. (Two ranks need two GPUs: RCCL refuses
    two ranks per device; the N > 1 logic runs over gloo in tests/test_parallel_cpu.py and tests/test_bench_gpu.py.)"""
    from capnet.parallel import GradAllReducer
    red = GradAllReducer(native=True)
    assert red.native.world == 1
    gs = [torch.randn(1000, 300, device=dev), torch.randn(7, device=dev), torch.randn(8192, 512, device=dev)]
    want = [g.clone() for g in gs]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        red(gs, 1.0)
        red(gs, 0.5)
    torch.cuda.current_stream().wait_stream(side)
    for g, w in zip(gs, want):
        assert torch.equal(g, w * 0.5)
    red.native.destroy()
    ops.check_device_errors()
