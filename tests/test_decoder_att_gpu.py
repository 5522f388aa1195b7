"""GPU parity of the attention decoder (DecoderFactoredLSTMAtt through the C ABI) against the
fixture produced by the reference's own class and against the CPU oracle at larger sizes."""
import random

import pytest
import torch

import capnet
from capnet import ops, synthetic
from capnet.model_att import DecoderFactoredLSTMAtt
from helpers import golden_case, golden_params, load_golden, rel_err, t
from oracle import decoders_ref as D

pytestmark = pytest.mark.gpu


def grad_close(a, b, rtol):
    """max|a-b| <= rtol*max|b| + 1e-6: full_att.bias has an exactly-zero gradient (softmax is
    shift invariant), so both sides hold ~1e-8 rounding noise there."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return (a - b).abs().max().item() <= rtol * b.abs().max().item() + 1e-6


@pytest.fixture(params=[-1, 1], ids=["chain-3-products", "chain-1-product"])
def chain(request):
    """Both forms of the factored input product U (S (V x)) inside capnet_att_seq_forward/backward: three products per step
    each way, and one against U S V with the intermediate rows formed for all steps at once (the default for <= 16 rows)."""
    from capnet._lib import lib
    old = lib().capnet_att_set_chain_mode(request.param)
    yield request.param
    lib().capnet_att_set_chain_mode(old)


def _step(dec, captions, lengths, feats, seed, ratio, mode, dev):
    dec.zero_grad()
    lens = [l - 1 for l in lengths]
    targets = D.packed_targets(captions[:, 1:], lens).to(dev)
    random.seed(seed)
    out, alphas = dec(captions[:, :-1].contiguous().to(dev), lens, feats.to(dev),
                      teacher_forcing_ratio=ratio, mode=mode)
    loss = ops.cross_entropy(out, targets) + 1.0 * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    loss.backward()
    ops.check_device_errors()
    return out, alphas, loss


@pytest.mark.parametrize("cname,seed,ratio", [("tf1_factual", 100, 1.0), ("tf0_happy", 101, 0.0),
                                              ("tfmix_factual", 3, 0.6), ("tfmix_sad", 5, 0.6)])
def test_attention_decoder_matches_reference_fixture(dev, chain, cname, seed, ratio):
    z = load_golden("decoder_att_tiny.npz")
    A, E, H, F, V, Cf, P = z["dims"].tolist()
    dec = DecoderFactoredLSTMAtt(A, E, H, F, V, 1, feature_size=Cf, dropout=0.0)
    dec.load_state_dict(golden_params(z))
    dec.to(dev).train()
    c = golden_case(z, cname)
    out, alphas, loss = _step(dec, t(z["captions"]), z["lengths"].tolist(), t(z["features"]), seed,
                              ratio, str(c["mode"]), dev)
    assert rel_err(out, c["logits"]) < 2e-5
    assert rel_err(alphas, c["alphas"]) < 2e-5
    assert abs(loss.item() - float(c["loss"])) / float(c["loss"]) < 2e-6
    n = 0
    for k, prm in dec.named_parameters():
        key = "grad." + k
        if key in c:
            assert prm.grad is not None, k
            assert grad_close(prm.grad, c[key], 1e-4), k
            n += 1
        else:
            assert prm.grad is None, k
    assert n > 30


@pytest.mark.parametrize("B,V,A,E,F,H,P,mode,ratio", [
    (5, 203, 24, 20, 24, 28, 9, "angry", 0.7),
    (1, 203, 24, 20, 24, 28, 9, "happy", 0.7),                # one caption
    (20, 203, 72, 68, 64, 64, 9, "sad", 0.7),                 # 17 .. 128 rows: the 64-row K-split kernels' partials
    (40, 1000, 512, 300, 512, 512, 196, "factual", 0.8),      # the full cell at 40 rows
    (12, 1000, 512, 300, 512, 512, 196, "factual", 0.8),     # BASELINE config 4 cell at 12/GPU
])
def test_attention_decoder_matches_oracle_seeded(dev, chain, B, V, A, E, F, H, P, mode, ratio):
    Cf = 512 if P < 100 else 2048
    dec = DecoderFactoredLSTMAtt(A, E, H, F, V, 1, feature_size=Cf, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=B, bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    _, captions, lengths = synthetic.make_batch(B, V, seed=40 + B, images=False, min_len=4, max_len=11)
    g = torch.Generator().manual_seed(B)
    feats = torch.randn(B, P, Cf, generator=g).abs() * 0.5      # post-ReLU-like features
    lens = [l - 1 for l in lengths]
    seed = 7
    random.seed(seed)
    tf = [random.random() < ratio for _ in range(max(lens))]
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    logits_r, alphas_r = D.factored_att_forward(pr, captions[:, :-1], lens, feats, tf, mode=mode)
    loss_r = D.att_loss(logits_r, alphas_r, D.packed_targets(captions[:, 1:], lens))
    loss_r.backward()
    out, alphas, loss = _step(dec, captions, lengths, feats, seed, ratio, mode, dev)
    assert rel_err(out, logits_r) < 1e-4
    assert rel_err(alphas, alphas_r) < 1e-4
    assert abs(loss.item() - loss_r.item()) / loss_r.item() < 1e-5
    for k, prm in dec.named_parameters():
        gr = pr[k].grad
        if gr is None:
            assert prm.grad is None, k
        else:
            assert grad_close(prm.grad, gr, 5e-4), k


# ---- nic DecoderRNNAtt (nic/model_att.py) ------------------------------------------------
def _nic_step(dec, captions, lengths, feats, seed, ratio, dev):
    dec.zero_grad()
    lens = [l - 1 for l in lengths]
    targets = D.packed_targets(captions[:, 1:], lens).to(dev)
    random.seed(seed)
    out, alphas = dec(captions[:, :-1].contiguous().to(dev), lens, feats.to(dev),
                      teacher_forcing_ratio=ratio)
    loss = ops.cross_entropy(out, targets) + 1.0 * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    loss.backward()
    ops.check_device_errors()
    return out, alphas, loss


@pytest.mark.parametrize("cname,seed,ratio", [("tf1", 100, 1.0), ("tf0", 101, 0.0), ("tfmix", 3, 0.6)])
def test_nic_attention_decoder_matches_reference_fixture(dev, cname, seed, ratio):
    from capnet.nic_model_att import DecoderRNNAtt
    z = load_golden("decoder_nic_att_tiny.npz")
    A, E, H, V, Cf, P = z["dims"].tolist()
    dec = DecoderRNNAtt(A, E, H, V, 1, feature_size=Cf, dropout=0.0)
    dec.load_state_dict(golden_params(z))
    dec.to(dev).train()
    c = golden_case(z, cname)
    out, alphas, loss = _nic_step(dec, t(z["captions"]), z["lengths"].tolist(), t(z["features"]),
                                  seed, ratio, dev)
    assert rel_err(out, c["logits"]) < 2e-5
    assert rel_err(alphas, c["alphas"]) < 2e-5
    assert abs(loss.item() - float(c["loss"])) / float(c["loss"]) < 2e-6
    n = 0
    for k, prm in dec.named_parameters():
        assert prm.grad is not None, k
        assert grad_close(prm.grad, c["grad." + k], 1e-4), k
        n += 1
    assert n == 19


@pytest.mark.parametrize("B", [12, 33])          # <= 16 rows: the one-launch products; above: the 64-row K-split kernels
def test_nic_attention_decoder_matches_oracle_full_size(dev, B):
    from capnet.nic_model_att import DecoderRNNAtt
    V, A, E, H, P, Cf, ratio = 1000, 512, 300, 512, 196, 2048, 0.8
    dec = DecoderRNNAtt(A, E, H, V, 1, feature_size=Cf, dropout=0.0)
    p = synthetic.decoder_state(dec.state_dict(), seed=3, bias_range=0.05)
    dec.load_state_dict(p)
    dec.to(dev).train()
    _, captions, lengths = synthetic.make_batch(B, V, seed=52, images=False, min_len=4, max_len=11)
    feats = torch.randn(B, P, Cf, generator=torch.Generator().manual_seed(9)).abs() * 0.5
    lens = [l - 1 for l in lengths]
    seed = 11
    random.seed(seed)
    tf = [random.random() < ratio for _ in range(max(lens))]
    pr = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    logits_r, alphas_r = D.lstm_att_forward(pr, captions[:, :-1], lens, feats, tf)
    loss_r = D.att_loss(logits_r, alphas_r, D.packed_targets(captions[:, 1:], lens))
    loss_r.backward()
    out, alphas, loss = _nic_step(dec, captions, lengths, feats, seed, ratio, dev)
    assert rel_err(out, logits_r) < 1e-4
    assert rel_err(alphas, alphas_r) < 1e-4
    assert abs(loss.item() - loss_r.item()) / loss_r.item() < 1e-5
    for k, prm in dec.named_parameters():
        assert grad_close(prm.grad, pr[k].grad, 5e-4), k


def test_nic_attention_sample_matches_reference_sequence(dev):
    from capnet.nic_model_att import DecoderRNNAtt
    z = load_golden("decoder_nic_att_tiny.npz")
    A, E, H, V, Cf, P = z["dims"].tolist()
    dec = DecoderRNNAtt(A, E, H, V, 1, feature_size=Cf, dropout=0.0)
    dec.load_state_dict({k[len("sample.param."):]: t(z[k]) for k in z.files
                         if k.startswith("sample.param.")})
    dec.to(dev).eval()
    seq = dec.sample(t(z["sample.features"]).to(dev), 1, 2, k=int(z["sample.k"]))
    assert seq.cpu().tolist() == z["sample.seq"].tolist()
