"""Fixture drift guards. No GPU needed.

tests/golden/decoder_factored_full_scalars.npz holds only scalars; its inputs are re-drawn from capnet.synthetic by
the generator (tools/gen_golden.py gen_factored_full, which runs the reference's own DecoderFactoredLSTM +
utils.clip_gradient + torch.optim.Adam, stylenet/model.py:157-196, utils.py:51-60) and by the GPU test. In round 3 a
change of synthetic.make_batch's draw order left that fixture stale and only the GPU run could see it. Here the
container's CPU suite (a) compares digests of the re-drawn inputs with the digests stored in the fixture and (b) runs
the oracle at full size against every stored scalar. (The other fixture with re-drawn inputs, trunk_b3.npz, is held
by tests/test_oracle_cpu.py::test_trunk_oracle_fp32_against_fp64_fixture the same way.)"""
import random
import zlib

import numpy as np
import torch

import capnet  # noqa: F401
from capnet import synthetic
from helpers import load_golden
from oracle import decoders_ref as D
from oracle import step_ref as S


def _digest(*tensors):
    c = 0
    for x in tensors:
        a = np.ascontiguousarray(x.numpy() if torch.is_tensor(x) else np.asarray(x))
        c = zlib.crc32(a.tobytes(), c)
    return c


def _full_inputs(z):
    E, H, F, V, B = z["dims"].tolist()
    shapes = {"B.weight": (V, E), "C.weight": (V, H), "C.bias": (V,)}
    for g in "ifoc":
        shapes.update({"U_%s.weight" % g: (H, F), "U_%s.bias" % g: (H,), "V_%s.weight" % g: (F, E),
                       "V_%s.bias" % g: (F,), "W_%s.weight" % g: (H, H), "W_%s.bias" % g: (H,)})
        for m in ("f", "happy_", "sad_", "angry_"):
            name = "S_%s%s" % (m, g)
            shapes.update({name + ".weight": (F, F), name + ".bias": (F,)})
    state = synthetic.decoder_state({k: torch.empty(s) for k, s in shapes.items()}, seed=1234)
    _, captions, lengths = synthetic.make_batch(B, V, seed=0, images=False)
    feats = torch.randn(B, E, generator=torch.Generator().manual_seed(77))
    return state, captions, lengths, feats


def test_factored_full_fixture_inputs_have_not_drifted():
    z = load_golden("decoder_factored_full_scalars.npz")
    state, captions, lengths, feats = _full_inputs(z)
    assert _digest(captions, lengths) == int(z["digest_batch"]), \
        "synthetic.make_batch no longer draws what the fixture was generated from: re-run tools/gen_golden.py factored_full"
    assert _digest(feats) == int(z["digest_features"])
    assert _digest(*[state[k] for k in sorted(state)]) == int(z["digest_params"])
    random.seed(0)
    for it in range(4):
        assert [random.random() < 0.8 for _ in range(lengths[0])] == [bool(x) for x in z["tf_masks"][it]]


def test_batch_of_a_seed_does_not_depend_on_the_images_flag():
    a = synthetic.make_batch(5, 50, seed=3, image_size=8)
    b = synthetic.make_batch(5, 50, seed=3, image_size=8, images=False)
    assert a[2] == b[2] and torch.equal(a[1], b[1]) and b[0] is None


def test_oracle_at_full_size_matches_the_reference_scalars():
    """Four clamp+Adam steps of configs[1]'s decoder (B=64, V=8192) through oracle/ against the reference's own run:
    losses, logits checksums and every parameter's gradient norm at every step, 1e-5 relative."""
    torch.set_num_threads(8)
    z = load_golden("decoder_factored_full_scalars.npz")
    p, captions, lengths, feats = _full_inputs(z)
    names = [str(s) for s in z["grad_names"]]
    assert sorted(names) == sorted(p.keys())
    opt = S.AdamRef(lr=2e-4)
    for it in range(4):
        tf = [bool(x) for x in z["tf_masks"][it]]
        loss, grads, _, logits = S.decoder_loss_and_grads(D.factored_lstm_forward, p, captions, lengths, feats, tf,
                                                          mode="factual")
        assert abs(loss.item() - float(z["losses"][it])) <= 1e-5 * float(z["losses"][it]), it
        assert abs(float(logits.double().abs().sum()) - float(z["logits_abs_sum"][it])) \
            <= 1e-5 * float(z["logits_abs_sum"][it]), it
        assert abs(float(logits.double().sum()) - float(z["logits_sum"][it])) <= 1e-6 * float(z["logits_abs_sum"][it]), it
        for k, want in zip(names, z["grad_norms"][it].tolist()):
            g = grads[k]
            got = float(g.double().norm()) if g is not None else 0.0
            assert abs(got - want) <= 1e-5 * want + 1e-12, (it, k, got, want)
        S.clip_gradient_(grads.values(), 0.5)
        opt.step(p, grads)
