"""The CPU oracle against the fixtures produced by the reference's own classes
(tools/gen_golden.py ran stylenet/model.py and nic/model.py verbatim). No GPU needed."""
import random

import pytest
import torch

from helpers import golden_case, golden_params, load_golden, rel_err, t
from oracle import decoders_ref as D
from oracle import step_ref as S

TOL = 2e-6


def _check(forward, z, cname, **kw):
    c = golden_case(z, cname)
    p = golden_params(z)
    captions, lengths = t(z["captions"]), z["lengths"].tolist()
    feats = t(z["features"]) if c.get("with_features", 1) else None
    tf = [bool(x) for x in c["tf_mask"]]
    loss, grads, dfeat, logits = S.decoder_loss_and_grads(forward, p, captions, lengths, feats, tf, **kw)
    assert rel_err(logits, c["logits"]) < TOL
    assert abs(loss.item() - float(c["loss"])) < 1e-6
    n = 0
    for k, g in grads.items():
        key = "grad." + k
        if key in c:
            assert rel_err(g, c[key]) < 1e-5, k
            n += 1
        else:
            assert g is None or float(g.abs().max()) == 0.0, k
    assert n > 0
    if feats is not None:
        assert rel_err(dfeat if dfeat is not None else torch.zeros_like(feats), c["dfeatures"]) < 1e-5 \
            or float(abs(c["dfeatures"]).max()) == 0.0


@pytest.mark.parametrize("cname", ["tf1_factual", "tf0_factual", "tfmix_factual", "tfmix_happy",
                                   "tfmix_angry_nofeat"])
def test_factored_forward_backward_matches_reference(cname):
    z = load_golden("decoder_factored_tiny.npz")
    _check(D.factored_lstm_forward, z, cname, mode=str(z["case.%s.mode" % cname]))


@pytest.mark.parametrize("cname", ["tf1", "tf0", "tfmix"])
def test_nic_forward_backward_matches_reference(cname):
    z = load_golden("decoder_nic_tiny.npz")
    _check(D.lstm_forward, z, cname)


def test_forward_step_matches_reference():
    z = load_golden("decoder_factored_tiny.npz")
    p = golden_params(z)
    x, h0, c0 = t(z["step.x"]), t(z["step.h0"]), t(z["step.c0"])
    for mode in ("factual", "happy", "sad", "angry"):
        h, c = D.factored_step(p, x, h0, c0, mode)
        assert rel_err(h, z["step.%s.h" % mode]) < TOL
        assert rel_err(c, z["step.%s.c" % mode]) < TOL


def test_tf_draws_follow_python_random():
    # the fixtures' masks are `random.random() < ratio` draws, one per step (model.py:181)
    z = load_golden("decoder_factored_tiny.npz")
    random.seed(3)
    assert [random.random() < 0.6 for _ in range(7)] == [bool(x) for x in z["case.tfmix_factual.tf_mask"]]


def test_clamp_adam_steps_match_reference():
    z = load_golden("decoder_factored_tiny.npz")
    a = load_golden("decoder_factored_tiny_adam.npz")
    p = {k: v.clone() for k, v in golden_params(z).items()}
    captions, lengths, feats = t(z["captions"]), z["lengths"].tolist(), t(z["features"])
    opt = S.AdamRef(lr=float(a["lr"]))
    for it, mode in enumerate([str(m) for m in a["modes"]]):
        random.seed(int(a["seeds"][it]))
        tf = [random.random() < float(a["ratio"]) for _ in range(max(lengths))]
        loss, grads, _, _ = S.decoder_loss_and_grads(D.factored_lstm_forward, p, captions, lengths,
                                                     feats, tf, mode=mode)
        assert abs(loss.item() - float(a["losses"][it])) < 2e-6
        S.clip_gradient_(grads.values(), float(a["clip"]))
        opt.step(p, grads)
    for k, v in p.items():
        assert rel_err(v, a["final." + k]) < 1e-5, k


def test_packed_targets_match_torch():
    from torch.nn.utils.rnn import pack_padded_sequence
    z = load_golden("decoder_factored_tiny.npz")
    captions, lengths = t(z["captions"]), z["lengths"].tolist()
    assert torch.equal(D.packed_targets(captions, lengths),
                       pack_padded_sequence(captions, lengths, batch_first=True)[0])


def test_trunk_oracle_fp32_against_fp64_fixture():
    """oracle/resnet152_ref.py in fp32 vs its own fp64 fixture: structure check (58.1 M conv+BN
    parameters, 155 convolutions) and a measurement of the fp32 noise floor at B=3."""
    import capnet  # noqa: F401
    from capnet import synthetic
    from oracle.resnet152_ref import resnet152_children
    z = load_golden("trunk_b3.npz")
    net = resnet152_children(True)
    sd = net.state_dict()
    assert sum(v.numel() for v in sd.values() if v.dim() == 4) == 57992384   # conv weights
    assert sum(1 for v in sd.values() if v.dim() == 4) == 155
    st = synthetic.trunk_state({"resnet." + k: v for k, v in sd.items()}, seed=1234)
    net.load_state_dict({k[len("resnet."):]: v for k, v in st.items()})
    net.train()
    imgs = synthetic.make_batch(int(z["B"]), 100, seed=0)[0]
    with torch.no_grad():
        y = net(imgs).reshape(int(z["B"]), -1)
    e = rel_err(y, z["pooled_train"])
    assert 1e-6 < e < 2e-3, e


@pytest.mark.parametrize("cname", ["tf1_factual", "tf0_happy", "tfmix_factual", "tfmix_sad"])
def test_attention_decoder_matches_reference(cname):
    """oracle factored_att_forward vs DecoderFactoredLSTMAtt run verbatim (fixture)."""
    z = load_golden("decoder_att_tiny.npz")
    c = golden_case(z, cname)
    p = {k: v.clone().requires_grad_(True) for k, v in golden_params(z).items()}
    captions, lengths, feats = t(z["captions"]), z["lengths"].tolist(), t(z["features"])
    lens = [l - 1 for l in lengths]
    tf = [bool(x) for x in c["tf_mask"]]
    logits, alphas = D.factored_att_forward(p, captions[:, :-1], lens, feats, tf, mode=str(c["mode"]))
    loss = D.att_loss(logits, alphas, D.packed_targets(captions[:, 1:], lens))
    loss.backward()
    assert rel_err(logits, c["logits"]) < TOL
    assert rel_err(alphas, c["alphas"]) < TOL
    assert abs(loss.item() - float(c["loss"])) < 2e-6
    n = 0
    for k, v in p.items():
        key = "grad." + k
        if key in c:
            assert rel_err(v.grad, c[key]) < 2e-5, k
            n += 1
        else:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
    assert n > 30


# ---- beam search (sample()) -----------------------------------------------------------
_SZ = load_golden("sample_tiny.npz")


@pytest.mark.parametrize("name", [str(c) for c in _SZ["cases"]])
def test_beam_search_matches_reference_sequences(name):
    """oracle/beam_ref.py against sequences produced by the reference's own sample() methods."""
    from oracle import beam_ref
    pre = "case.%s." % name
    c = {k[len(pre):]: _SZ[k] for k in _SZ.files if k.startswith(pre)}
    p = {k[len("param."):]: t(v) for k, v in c.items() if k.startswith("param.")}
    start, end = [int(v) for v in _SZ["start_end"]]
    dims = [int(v) for v in c["dims"]]
    E, H, F, V, k, maxlen = dims[:6]
    kind = str(c["kind"])
    with torch.no_grad():
        if kind == "factored":
            seq = beam_ref.sample_factored(p, H, start, end, k=k, mode=str(c["mode"]), max_seq_length=maxlen)
        elif kind == "nic":
            seq = beam_ref.sample_lstm(p, H, start, end, k=k, max_seq_length=maxlen)
        else:
            seq = beam_ref.sample_factored_att(p, t(c["features"]), start, end, k=k, mode=str(c["mode"]),
                                               max_seq_length=maxlen)
    assert seq.tolist() == c["seq"].tolist()


# ---- nic DecoderRNNAtt -----------------------------------------------------------------
@pytest.mark.parametrize("cname", ["tf1", "tf0", "tfmix"])
def test_nic_attention_decoder_matches_reference(cname):
    """oracle lstm_att_forward vs nic DecoderRNNAtt run verbatim (fixture)."""
    z = load_golden("decoder_nic_att_tiny.npz")
    c = golden_case(z, cname)
    p = {k: v.clone().requires_grad_(True) for k, v in golden_params(z).items()}
    captions, lengths, feats = t(z["captions"]), z["lengths"].tolist(), t(z["features"])
    lens = [l - 1 for l in lengths]
    tf = [bool(x) for x in c["tf_mask"]]
    logits, alphas = D.lstm_att_forward(p, captions[:, :-1], lens, feats, tf)
    loss = D.att_loss(logits, alphas, D.packed_targets(captions[:, 1:], lens))
    loss.backward()
    assert rel_err(logits, c["logits"]) < TOL
    assert rel_err(alphas, c["alphas"]) < TOL
    assert abs(loss.item() - float(c["loss"])) < 2e-6
    n = 0
    for k, v in p.items():
        key = "grad." + k
        if key in c:
            assert rel_err(v.grad, c[key]) < 2e-5 or float(abs(c[key]).max()) < 1e-6, k
            n += 1
    assert n >= 18


def test_nic_attention_beam_search_matches_reference_sequence():
    from oracle import beam_ref
    z = load_golden("decoder_nic_att_tiny.npz")
    p = {k[len("sample.param."):]: t(z[k]) for k in z.files if k.startswith("sample.param.")}
    with torch.no_grad():
        seq = beam_ref.sample_lstm_att(p, t(z["sample.features"]), 1, 2, k=int(z["sample.k"]))
    assert seq.tolist() == z["sample.seq"].tolist()


# ---- BLEU-4 (host metric of the validation loop) ---------------------------------------------
def test_corpus_bleu_known_values():
    from capnet.metrics import corpus_bleu
    ref = [[1, 2, 3, 4, 5, 6, 7]]
    assert corpus_bleu([ref], [[1, 2, 3, 4, 5, 6, 7]]) == pytest.approx(1.0)
    assert corpus_bleu([ref], [[9, 9, 9, 9]]) == 0.0
    # hand computation: hyp = 1 2 3 4 5 9 7 vs ref: p1 = 6/7, p2 = 4/6, p3 = 3/5, p4 = 2/4, bp = 1
    want = (6 / 7 * 4 / 6 * 3 / 5 * 2 / 4) ** 0.25
    assert corpus_bleu([ref], [[1, 2, 3, 4, 5, 9, 7]]) == pytest.approx(want)
    # brevity penalty: hyp of 5 tokens against a 7-token reference, all n-grams matching
    import math
    assert corpus_bleu([ref], [[1, 2, 3, 4, 5]]) == pytest.approx(math.exp(1 - 7 / 5))
    # clipping and multiple references (Papineni et al. example: "the the the ..." -> 2/7 unigrams)
    refs = [[1, 5, 6, 7, 1, 8, 9], [10, 6, 11, 5, 7, 1, 8]]
    hyp = [1, 1, 1, 1, 1, 1, 1]
    assert corpus_bleu([refs], [hyp], weights=(1.0,)) == pytest.approx(2 / 7)
