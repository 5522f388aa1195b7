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


def test_save_and_load_checkpoint_round_trip(tmp_path):
    dec = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    enc = torch.nn.Linear(3, 2)            # any module with a state_dict stands in for the encoder
    opt = torch.optim.Adam(dec.parameters(), lr=1e-3)
    save_checkpoint(str(tmp_path), "toy", "factual", 3, 1, enc, dec, opt, None, 0.25, True)
    for name in ("factual_checkpoint_toy.pth.tar", "factual_BEST_checkpoint_toy.pth.tar"):
        assert os.path.exists(os.path.join(str(tmp_path), name))
    dec2 = DecoderFactoredLSTM(12, 16, 16, 37, 1)
    enc2 = torch.nn.Linear(3, 2)
    meta = load_checkpoint(os.path.join(str(tmp_path), "factual_checkpoint_toy.pth.tar"), enc2, dec2)
    assert meta["epoch"] == 3 and meta["epochs_since_improvement"] == 1 and meta["bleu-4"] == 0.25
    for (k, a), (_, b) in zip(dec.state_dict().items(), dec2.state_dict().items()):
        assert torch.equal(a, b), k
