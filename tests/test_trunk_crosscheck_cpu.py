"""Independent cross-check of the trunk oracle's ARCHITECTURE restatement.

torchvision (the reference's source of resnet152, stylenet/model.py:15) is not installed and its
weights need a network fetch, so the trunk stays "parity unpinned" against the reference itself.
What can be checked here: oracle/resnet152_ref.py against a second, independently written
implementation of the same published architecture (ResNet-152 "v1.5", stride on the 3x3,
bottleneck x [3, 8, 36, 3]) -- transformers.ResNetModel, which happens to be installed. The same
seeded weights go into both (name mapping below), both run in train mode (batch statistics) on
the same images, and the pooled features and a BatchNorm's running statistics must agree.
Skipped when transformers is not importable."""
import pytest
import torch

from capnet import synthetic
from oracle.resnet152_ref import resnet152_children

transformers = pytest.importorskip("transformers")


def _map_key(k):
    """oracle (torchvision-style nn.Sequential children) key -> transformers.ResNetModel key."""
    parts = k.split(".")
    norm = {"weight": "weight", "bias": "bias", "running_mean": "running_mean",
            "running_var": "running_var", "num_batches_tracked": "num_batches_tracked"}
    if parts[0] == "0":
        return "embedder.embedder.convolution.weight"
    if parts[0] == "1":
        return "embedder.embedder.normalization." + norm[parts[1]]
    stage, block = int(parts[0]) - 4, int(parts[1])
    pre = "encoder.stages.%d.layers.%d." % (stage, block)
    if parts[2] == "downsample":
        return pre + ("shortcut.convolution.weight" if parts[3] == "0"
                      else "shortcut.normalization." + norm[parts[4]])
    idx = int(parts[2][-1]) - 1
    if parts[2].startswith("conv"):
        return pre + "layer.%d.convolution.weight" % idx
    return pre + "layer.%d.normalization.%s" % (idx, norm[parts[3]])


def test_oracle_trunk_matches_an_independent_resnet152():
    from transformers import ResNetConfig, ResNetModel
    torch.manual_seed(0)
    net = resnet152_children(True)
    st = synthetic.trunk_state({"resnet." + k: v for k, v in net.state_dict().items()}, seed=1234)
    st = {k[len("resnet."):]: v for k, v in st.items()}
    net.load_state_dict(st)
    cfg = ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048],
                       depths=[3, 8, 36, 3], layer_type="bottleneck", hidden_act="relu",
                       downsample_in_first_stage=False, downsample_in_bottleneck=False)
    other = ResNetModel(cfg)
    mapped = {_map_key(k): v for k, v in st.items()}
    assert set(mapped) == set(other.state_dict()), "the two implementations disagree on the tensor set"
    other.load_state_dict(mapped)
    net.train()
    other.train()
    imgs = synthetic.make_batch(2, 100, seed=0)[0]
    with torch.no_grad():
        a = net(imgs).reshape(2, -1)
        b = other(pixel_values=imgs).pooler_output.reshape(2, -1)
    assert a.shape == b.shape == (2, 2048)
    err = ((a - b).abs().max() / b.abs().max()).item()
    assert err < 1e-5, err
    # the same BatchNorm saw the same batch statistics
    sd_o = other.state_dict()
    for k in ("7.2.bn3.running_mean", "7.2.bn3.running_var", "1.running_var"):
        x, y = net.state_dict()[k], sd_o[_map_key(k)]
        assert ((x - y).abs().max() / y.abs().max()).item() < 1e-5, k
