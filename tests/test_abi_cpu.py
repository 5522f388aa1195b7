"""The C-ABI library builds, loads on a CPU-only host and exports every symbol that
include/capnet.h declares (no compute calls here)."""
import os
import re

import pytest

import capnet
from capnet import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "capnet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(capnet_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = capnet.lib()
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.SIGNATURES) == names
    assert lib.capnet_abi_version() == 1


def test_host_side_argument_checks_need_no_gpu():
    lib = capnet.lib()
    # negative dimension is rejected before any device work
    rc = lib.capnet_sgemm(0, 0, -1, 4, 4, None, 4, None, 4, None, 4, None, 0, 1, 0, 0, 0, 0, 0, None)
    assert rc != 0 and b"negative" in lib.capnet_last_error()
    import ctypes as C
    h = C.c_void_p()
    assert lib.capnet_trunk_create(2, 100, 224, C.byref(h)) != 0       # not a multiple of 32
    assert lib.capnet_trunk_create(2, 224, 224, C.byref(h)) == 0
    assert lib.capnet_trunk_num_convs(h) == 155
    assert lib.capnet_trunk_final_side(h) == 7
    # 11.512 GMAC per image
    assert abs(lib.capnet_trunk_flops(h) / 2 / 2 / 1e9 - 11.512) < 0.01
    lib.capnet_trunk_destroy(h)


def test_ops_refuse_cpu_tensors():
    import torch
    from capnet import ops
    with pytest.raises(capnet.CapnetError):
        ops.sgemm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_batch_sizes_contract():
    from capnet import ops
    assert ops.batch_sizes_from_lengths([5, 3, 3, 1]) == [4, 3, 3, 1, 1]
    with pytest.raises(capnet.CapnetError):
        ops.batch_sizes_from_lengths([3, 5])
    with pytest.raises(capnet.CapnetError):
        ops.batch_sizes_from_lengths([])
