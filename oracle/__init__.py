"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

A plain torch-CPU restatement of the reference's training step (deryrahman/
image-caption-emotion-indonesia), used as the checker for the HIP path. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product
(image-caption-emotion-indonesia_amd/) never does, and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * decoders / loss / clamp+Adam: pinned against outputs of the reference's own classes
    (stylenet/model.py, nic/model.py imported verbatim in the build container by
    tools/gen_golden.py; fixtures in tests/golden/decoder_*.npz).
  * ResNet-152 trunk: torchvision is not installed and the reference holds no fixture for it,
    so the trunk restatement follows the published torchvision 0.2.2 architecture and is
    PARITY UNPINNED against the reference; tests/golden/trunk_*.npz pin the oracle to itself.
"""
