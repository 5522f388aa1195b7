"""capnet: MI355X-native training step of the StyleNet / NIC captioning models.

Python mirror of the reference's class surface (stylenet/model.py, nic/model.py,
stylenet/train_multitask.py) over the C ABI of libcapnet_hip.so (include/capnet.h).
Import as `import capnet` (see capnet.py at the repository root).
"""
from ._lib import CapnetError, LIB_PATH, SIGNATURES, lib  # noqa: F401

__all__ = ["CapnetError", "LIB_PATH", "SIGNATURES", "lib"]
