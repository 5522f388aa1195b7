"""Static checks of the gfx950 assembly that ships (tools/isa_inflight_check.py), for every kernel file that issues
vector-memory instructions from inline asm:

* no instruction touches the destination registers of a hand-issued load before the source declares it landed
  (a compiler copy, spill or re-use of a register a load is still going to write: wrong data, or a wild address);
* no inline-asm vector-memory instruction reads an SGPR that a VALU instruction (v_readlane_b32 = the reload of a
  spilled SGPR, v_readfirstlane_b32, a carry-out) wrote fewer than five wait states earlier -- the hazard the compiler
  pads for its own instructions but not inside an asm statement. This was the cause of round 2's GPU memory faults
  that "came and went with register allocation" (DESIGN.md 4i).

hipcc cross-compiles without a GPU; the files are compiled with the Makefile's flags."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_inflight_check as chk  # noqa: E402

FILES = ["conv_f16x3.hip", "conv_f32_v2.hip", "lstm_persist.hip", "gemm_dma.hip", "conv3x3_patch.hip", "conv1x1_areg.hip", "gemm_f32.hip", "fused_block.hip"]


@pytest.fixture(scope="module")
def isa():
    with ThreadPoolExecutor(4) as ex:
        outs = list(ex.map(lambda f: chk.build_isa(os.path.join(chk.CSRC, f), os.path.join(ROOT, "build", "isa")), FILES))
    return dict(zip(FILES, outs))


def test_every_file_with_asm_memory_instructions_is_listed():
    import re
    pat = re.compile(r'asm volatile\([^;]*?(global_load|global_store|buffer_load|buffer_store)', re.S)
    mine = set()
    for f in os.listdir(chk.CSRC):
        if f.endswith((".hip", ".h", ".cpp")) and pat.search(open(os.path.join(chk.CSRC, f)).read()):
            mine.add(f)
    users = {f for f in os.listdir(chk.CSRC) if f.endswith(".hip") and
             any(k in open(os.path.join(chk.CSRC, f)).read() for k in ("gload16(", "glds16(", "gstore32(", "glds16_imm<"))}
    assert (mine | users) - {"mfma_core.h"} <= set(FILES), sorted((mine | users) - set(FILES))


@pytest.mark.parametrize("name", FILES)
def test_inflight_registers_and_sgpr_hazards(isa, name):
    kernels = chk.parse_kernels(isa[name])
    assert kernels
    loads = 0
    for kname, insts in kernels.items():
        hz = chk.check_sgpr_hazard(kname, insts)
        assert not hz, (kname, sorted(hz.items())[:3])
        v, nload, marks = chk.check_kernel(kname, insts)
        assert not v, (kname, sorted(v.items())[:3])
        loads += nload
    if name not in ("gemm_dma.hip", "conv3x3_patch.hip", "conv1x1_areg.hip", "gemm_f32.hip"):   # (their asm loads are LDS-DMA: no register destination; gemm_f32's asm is a store)
        assert loads > 0


def test_the_checker_sees_what_it_is_for(tmp_path):
    """Known-bad assembly: a copy of an in-flight register, and a v_readlane-fed base."""
    bad = tmp_path / "bad.s"
    bad.write_text("""
k1:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\t;;#ASMSTART
\ts_nop 4
\tglobal_load_dwordx4 v[4:7], v1, s[0:1]
\t;;#ASMEND
\tv_mov_b32_e32 v9, v5
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\ts_endpgm
.Lfunc_end0:
k2:
\tv_readlane_b32 s0, v40, 3
\tv_readlane_b32 s1, v40, 4
\t;;#ASMSTART
\tglobal_store_dword v2, v3, s[0:1]
\t;;#ASMEND
\ts_endpgm
.Lfunc_end1:
k3:
\tv_readlane_b32 s0, v40, 3
\tv_readlane_b32 s1, v40, 4
\t;;#ASMSTART
\ts_nop 4
\tglobal_load_dwordx4 v[4:7], v1, s[0:1]
\t;;#ASMEND
\t;;#ASMSTART
\ts_waitcnt vmcnt(1)
\t;;#ASMEND
\t;;#ASMSTART
\t; capnet.landed v[4:7]
\t;;#ASMEND
\tv_mov_b32_e32 v9, v5
\ts_endpgm
.Lfunc_end2:
""")
    ks = chk.parse_kernels(str(bad))
    assert set(ks) == {"k1", "k2", "k3"}
    assert chk.check_kernel("k1", ks["k1"])[0] and not chk.check_sgpr_hazard("k1", ks["k1"])
    assert chk.check_sgpr_hazard("k2", ks["k2"])
    assert not chk.check_kernel("k3", ks["k3"])[0] and not chk.check_sgpr_hazard("k3", ks["k3"])
