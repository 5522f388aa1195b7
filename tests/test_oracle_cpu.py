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


def test_corpus_bleu_matches_nltk_doctest_values():
    """The worked example in the docstring of nltk.translate.bleu_score.corpus_bleu (nltk 3.4.1, the
    version stylenet/requirements.txt pins; nltk itself is not installable here): two hypotheses,
    three + one references. Published values: corpus 0.5920..., sentence scores 0.5045... and
    0.7400... (their mean, 0.6223..., is the docstring's second number). This pins the restatement
    to nltk's own output on a multi-reference corpus with clipping and closest-length selection."""
    from capnet.metrics import corpus_bleu
    words = {}

    def ids(s):
        return [words.setdefault(w, len(words)) for w in s.split()]
    hyp1 = ids("It is a guide to action which ensures that the military always obeys the commands of the party")
    ref1a = ids("It is a guide to action that ensures that the military will forever heed Party commands")
    ref1b = ids("It is the guiding principle which guarantees the military forces always being under the "
                "command of the Party")
    ref1c = ids("It is the practical guide for the army always to heed the directions of the party")
    hyp2 = ids("he read the book because he was interested in world history")
    ref2a = ids("he was interested in world history because he read the book")
    corpus = corpus_bleu([[ref1a, ref1b, ref1c], [ref2a]], [hyp1, hyp2])
    s1 = corpus_bleu([[ref1a, ref1b, ref1c]], [hyp1])
    s2 = corpus_bleu([[ref2a]], [hyp2])
    assert abs(corpus - 0.5920) < 1e-4 and int(corpus * 1e4) == 5920
    assert int(s1 * 1e4) == 5045 and int((s1 + s2) / 2 * 1e4) == 6223


def test_corpus_bleu_hand_computed_corpus_cases():
    """Counts pooled over the corpus BEFORE the ratio (micro-average), brevity penalty from the summed
    closest reference lengths, zero higher-order matches."""
    import math
    from capnet.metrics import corpus_bleu
    # two segments. A: hyp 1 2 3 4 5 (5 tok) vs refs [1 2 3 4 5 6] and [1 2 3 9]: closest length 6 (|6-5| = |4-5|
    # -> the shorter one, 4, by nltk's (abs diff, length) tie-break). B: hyp 7 8 9 10 vs ref 7 8 9 11 12 13.
    refs_a = [[1, 2, 3, 4, 5, 6], [1, 2, 3, 9]]
    hyp_a = [1, 2, 3, 4, 5]
    refs_b = [[7, 8, 9, 11, 12, 13]]
    hyp_b = [7, 8, 9, 10]
    # n-gram matches A: 5/5, 4/4, 3/3, 2/2 ; B: 3/4, 2/3, 1/2, 0/1
    p = [(5 + 3) / 9, (4 + 2) / 7, (3 + 1) / 5, (2 + 0) / 3]
    hyp_len, ref_len = 5 + 4, 4 + 6
    bp = math.exp(1 - ref_len / hyp_len)
    want = bp * math.exp(sum(0.25 * math.log(x) for x in p))
    assert corpus_bleu([refs_a, refs_b], [hyp_a, hyp_b]) == pytest.approx(want, rel=1e-12)
    # a corpus without any 4-gram match scores 0 (nltk: ~1e-77 through its float_info.min guard)
    assert corpus_bleu([refs_b], [hyp_b]) == 0.0
    # an empty hypothesis adds its reference length and, as in nltk's modified_precision
    # (denominator max(1, count)), ONE to every n-gram denominator
    want = math.exp(1 - (4 + 6) / 5) * (5 / 6 * 4 / 5 * 3 / 4 * 2 / 3) ** 0.25
    assert corpus_bleu([refs_a, refs_b], [hyp_a, []]) == pytest.approx(want, rel=1e-12)
