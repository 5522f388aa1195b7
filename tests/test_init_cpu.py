"""a14: the constructors draw what the reference's constructors draw.

tests/golden/init_stats.json holds, for each reference decoder class constructed after
torch.manual_seed(s) (tools/gen_golden.py init), every parameter's shape, range, mean, std and
first values (stylenet/model.py:99-113, stylenet/model_att.py:169-183, nic/model.py:58-72,
nic/model_att.py reset_parameters/init_weights). Two levels:
  * distribution: 1-D parameters are zero, >= 2-D ones are xavier-uniform over the WHOLE tensor
    (LSTMCell.weight_ih as one [4H, E] matrix, App. A-12), embedding and output projection are
    U(-0.1, 0.1) -- checked against the reference's statistics;
  * stream: with the same seed and the same torch version the mirror classes consume the generator
    in the same order (constructor draws first, then reset_parameters in registration order), so the
    first values agree exactly."""
import json
import math
import os

import pytest
import torch

import capnet  # noqa: F401
from capnet.model import DecoderFactoredLSTM
from capnet.model_att import DecoderFactoredLSTMAtt
from capnet.nic_model import DecoderRNN
from capnet.nic_model_att import DecoderRNNAtt
from helpers import GOLDEN

STATS = json.load(open(os.path.join(GOLDEN, "init_stats.json")))
CLASSES = {"stylenet.DecoderFactoredLSTM": DecoderFactoredLSTM,
           "stylenet.DecoderFactoredLSTMAtt": DecoderFactoredLSTMAtt,
           "nic.DecoderRNN": DecoderRNN, "nic.DecoderRNNAtt": DecoderRNNAtt}


def _parse(key):
    name, rest = key.split("[", 1)
    args, seed = rest.split("] seed=")
    return name, [int(x) for x in args.split(",")], int(seed)


@pytest.mark.parametrize("key", sorted(STATS))
def test_constructor_init_matches_reference(key):
    name, args, seed = _parse(key)
    torch.manual_seed(seed)
    m = CLASSES[name](*args)
    sd = m.state_dict()
    ref = STATS[key]
    assert list(sd.keys()) == list(ref.keys())
    for k, v in sd.items():
        r = ref[k]
        assert list(v.shape) == r["shape"], k
        v64 = v.double()
        if v.dim() == 1:
            assert r["min"] == 0.0 and r["max"] == 0.0          # the reference zeroes every 1-D parameter
            assert float(v64.abs().max()) == 0.0, k
            continue
        # distribution: same support and spread as the reference's draw
        fan_out, fan_in = v.shape[0], v.shape[1]
        special = k in ("B.weight", "C.weight", "embed.weight", "linear.weight")
        bound = 0.1 if special else math.sqrt(6.0 / (fan_in + fan_out))
        assert r["max"] <= bound and r["min"] >= -bound, ("reference outside the stated bound", k)
        assert float(v64.max()) <= bound and float(v64.min()) >= -bound, k
        n = v.numel()
        tol = 4.0 / math.sqrt(n)            # ~4 sigma of the sample std / mean of a uniform
        assert abs(float(v64.std()) - bound / math.sqrt(3)) < tol * bound, k
        assert abs(r["std"] - bound / math.sqrt(3)) < tol * bound, k
        assert abs(float(v64.mean())) < tol * bound, k
        # stream: identical draws
        head = [float(x) for x in v.reshape(-1)[:8]]
        assert head == r["head"], (k, head, r["head"])
        assert float(v64.min()) == r["min"] and float(v64.max()) == r["max"], k
