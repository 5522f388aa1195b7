"""Checkpoint compatibility: the mirror classes expose the reference's state_dict
keys, in the reference's order and shapes (fixture generated from the reference's own classes,
tools/gen_golden.py state_keys), and save_checkpoint / load_checkpoint round-trip."""
import json
import os

import torch

import capnet
from capnet.model import DecoderFactoredLSTM
from capnet.model_att import DecoderFactoredLSTMAtt
from capnet.nic_model import DecoderRNN
from capnet.nic_model_att import DecoderRNNAtt
from capnet.utils import load_checkpoint, save_checkpoint
from helpers import GOLDEN

KEYS = json.load(open(os.path.join(GOLDEN, "state_dict_keys.json")))


def _kv(module):
    return [[k, list(v.shape)] for k, v in module.state_dict().items()]


def test_factored_state_dict_matches_reference():
    assert _kv(DecoderFactoredLSTM(300, 512, 512, 1000, 1)) == \
        KEYS["stylenet.DecoderFactoredLSTM(300,512,512,1000,1)"]


def test_factored_att_state_dict_matches_reference():
    assert _kv(DecoderFactoredLSTMAtt(512, 300, 512, 512, 1000, 1)) == \
        KEYS["stylenet.DecoderFactoredLSTMAtt(512,300,512,512,1000,1)"]


def test_nic_state_dict_matches_reference():
    assert _kv(DecoderRNN(300, 512, 1000, 1)) == KEYS["nic.DecoderRNN(300,512,1000,1)"]


def test_nic_att_state_dict_matches_reference():
    assert _kv(DecoderRNNAtt(512, 300, 512, 1000, 1)) == KEYS["nic.DecoderRNNAtt(512,300,512,1000,1)"]


def test_transfer_learning_parameter_subsets_exist():
    """stylenet/train_transfer.py:94-115 builds its optimiser from these attributes."""
    dec = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    for emo in ("happy", "sad", "angry"):
        for g in "ifoc":
            assert len(list(getattr(dec, "S_%s_%s" % (emo, g)).parameters())) == 2
    assert len(list(dec.C.parameters())) == 2 and len(list(dec.B.parameters())) == 1
    att = DecoderFactoredLSTMAtt(8, 12, 16, 16, 37, 1, feature_size=512)
    for name in ("attention_happy", "attention_sad", "attention_angry", "f_beta", "init_h", "init_c"):
        assert len(list(getattr(att, name).parameters())) >= 2


def _fake_steps(opt, n):
    """Optimiser state as n GPU steps would leave it (capnet.optim.Adam.step needs the GPU; the
    layout of its state is what this CPU test is about)."""
    g = torch.Generator().manual_seed(7)
    for grp in opt.param_groups:
        for p in grp["params"][::2]:          # every other parameter: the rest never saw a gradient
            opt.state[p] = {"step": n, "exp_avg": torch.randn(p.shape, generator=g),
                            "exp_avg_sq": torch.rand(p.shape, generator=g)}


def test_save_and_load_checkpoint_round_trip(tmp_path):
    """utils.save_checkpoint / load_checkpoint with capnet.optim.Adam for both optimisers
    (stylenet/train_multitask.py:163-177: `optimizer` over decoder + head, `lang_optimizer` over the
    decoder): the files load with weights_only=True and restore moments, step counts and lr."""
    from capnet.optim import Adam
    dec = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    enc = torch.nn.Linear(3, 2)            # any module with a state_dict stands in for the encoder
    opt = Adam(list(dec.parameters()) + list(enc.parameters()), lr=2e-4)
    lang = Adam(dec.parameters(), lr=5e-4)
    _fake_steps(opt, 3)
    _fake_steps(lang, 5)
    lang.param_groups[0]["lr"] = 5e-4 * 0.8          # adjust_learning_rate happened
    save_checkpoint(str(tmp_path), "toy", "factual", 3, 1, enc, dec, opt, lang, 0.25, True)
    for name in ("factual_checkpoint_toy.pth.tar", "factual_BEST_checkpoint_toy.pth.tar"):
        assert os.path.exists(os.path.join(str(tmp_path), name))
    dec2 = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    enc2 = torch.nn.Linear(3, 2)
    opt2 = Adam(list(dec2.parameters()) + list(enc2.parameters()), lr=1.0)
    lang2 = Adam(dec2.parameters(), lr=1.0)
    meta = load_checkpoint(os.path.join(str(tmp_path), "factual_checkpoint_toy.pth.tar"), enc2, dec2,
                           opt2, lang2)
    assert meta["epoch"] == 3 and meta["epochs_since_improvement"] == 1 and meta["bleu-4"] == 0.25
    for (k, a), (_, b) in zip(dec.state_dict().items(), dec2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == 2e-4 and abs(lang2.param_groups[0]["lr"] - 4e-4) < 1e-12
    assert opt2.param_groups[0]["betas"] == (0.9, 0.999)
    for o, o2, n in ((opt, opt2, 3), (lang, lang2, 5)):
        ps, ps2 = o.param_groups[0]["params"], o2.param_groups[0]["params"]
        assert len(o2.state) == len(o.state) == (len(ps) + 1) // 2
        for p, p2 in zip(ps, ps2):
            if p in o.state:
                assert o2.state[p2]["step"] == n
                assert torch.equal(o2.state[p2]["exp_avg"], o.state[p]["exp_avg"])
                assert torch.equal(o2.state[p2]["exp_avg_sq"], o.state[p]["exp_avg_sq"])
            else:
                assert p2 not in o2.state


def test_adam_state_dict_has_torch_layout():
    """The same dict loads into torch.optim.Adam and back (a reference-side script that still uses
    torch's optimiser can resume from a capnet checkpoint and vice versa)."""
    from capnet.optim import Adam
    dec = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    opt = Adam(dec.parameters(), lr=2e-4)
    _fake_steps(opt, 4)
    sd = opt.state_dict()
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(list(dec.parameters()))))
    topt = torch.optim.Adam(dec.parameters(), lr=1.0)
    topt.load_state_dict(sd)
    p0 = opt.param_groups[0]["params"][0]
    assert float(topt.state[p0]["step"]) == 4.0 and torch.equal(topt.state[p0]["exp_avg"], opt.state[p0]["exp_avg"])
    assert topt.param_groups[0]["lr"] == 2e-4
    back = Adam(dec.parameters(), lr=1.0)
    back.load_state_dict(topt.state_dict())
    assert back.state[p0]["step"] == 4 and torch.equal(back.state[p0]["exp_avg_sq"], opt.state[p0]["exp_avg_sq"])
    assert back.param_groups[0]["lr"] == 2e-4
