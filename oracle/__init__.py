"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A plain torch-CPU restatement of the reference's training step (deryrahman/
image-caption-emotion-indonesia), used as the checker for the HIP path. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(image-caption-emotion-indonesia_amd/) never does, and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * decoders (all four: DecoderFactoredLSTM, DecoderRNN, DecoderFactoredLSTMAtt, DecoderRNNAtt),
    their beam search, loss, clamp+Adam: pinned against outputs of the reference's own classes
    (stylenet/model.py, model_att.py, nic/model.py, model_att.py imported verbatim in the build
    container by tools/gen_golden.py; fixtures tests/golden/decoder_*.npz, sample_tiny.npz,
    state_dict_keys.json).
  * image transform chain: pinned against Pillow's own resize output and torch arithmetic
    (tests/golden/image_tiny.npz).
  * ResNet-152 trunk: torchvision is not installed and the reference holds no fixture for it,
    so the trunk restatement follows the published torchvision 0.2.2 architecture and is
    PARITY UNPINNED against the reference; tests/golden/trunk_*.npz pin the oracle to itself and
    tests/test_trunk_crosscheck_cpu.py checks it against transformers.ResNetModel (an independent
    implementation of the same architecture) with identical weights.
"""
