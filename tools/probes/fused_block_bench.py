"""Times csrc/fused_block.hip's launches alone on the trunk's stage shapes (batch 64) beside the launches they replace:
conv3 (conv_f16x3_kernel) + bn_finalize + tail/conv1 (conv1x1_tail_kernel).   python tools/probes/fused_block_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402,F401
from capnet import ops  # noqa: E402
from capnet._lib import check, current_stream, lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
L = lib()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for side, MID in ((56, 64), (28, 128), (14, 256)):
    M, C = B * side * side, 4 * MID
    g = torch.Generator().manual_seed(MID)
    y2 = torch.randn(M, MID, generator=g).to(dev)
    s2, t2 = (torch.rand(MID, generator=g) + 0.5).to(dev), (torch.randn(MID, generator=g) * 0.5).to(dev)
    w3 = (torch.randn(C, MID, 1, 1, generator=g) * (2.0 / MID) ** 0.5).to(dev)
    w1 = (torch.randn(MID, C, 1, 1, generator=g) * (2.0 / C) ** 0.5).to(dev)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    res = torch.randn(M, C, generator=g).to(dev)
    img3, img1 = ops.pack_fused_block_weight(w3, 0), ops.pack_fused_block_weight(w1, 1)
    sc, sh = ops.fused_block_stats(y2, s2, t2, img3, gamma, beta)
    work = torch.empty(L.capnet_fused_block_stats_floats(M, MID), device=dev)
    out, y1 = torch.empty(M, C, device=dev), torch.empty(M, MID, device=dev)
    tiles = L.capnet_fused_block_tiles(M, MID)
    ps, pq = torch.empty(tiles, MID, device=dev), torch.empty(tiles, MID, device=dev)
    err = ops.err_flag(dev)
    st = current_stream()

    def stats():
        check(L.capnet_fused_block_stats(ptr(y2), ptr(s2), ptr(t2), ptr(img3), M, MID, 0, ptr(gamma), ptr(beta), None, None, 0.1,
                                         1e-5, ptr(sc), ptr(sh), ptr(work), ptr(err), st))

    def fused():
        check(L.capnet_fused_block_forward(ptr(y2), ptr(s2), ptr(t2), ptr(img3), ptr(sc), ptr(sh), ptr(res), None, None, ptr(out),
                                           ptr(img1), ptr(y1), ptr(ps), ptr(pq), M, MID, 0, 0, ptr(err), st))
    # the launches they replace
    bn3 = 128 if C % 128 == 0 else 64
    old3 = ops.pack_conv_weight_f16x3(w3, bn3)
    bn1 = 256 if (MID % 256 == 0 and (M + 127) // 128 >= 64) else (128 if MID % 128 == 0 else 64)
    old1 = ops.pack_conv_weight_f16x3(w1, bn1)
    y3 = torch.empty(M, C, device=dev)
    t3 = L.capnet_conv1x1_tiles_m(M)
    p3s, p3q = torch.empty(t3, C, device=dev), torch.empty(t3, C, device=dev)
    p1s, p1q = torch.empty(t3, MID, device=dev), torch.empty(t3, MID, device=dev)

    def conv3():
        check(L.capnet_conv2d_fwd_f16x3(ptr(y2), MID, MID, MID, ptr(old3), bn3, ptr(y3), ptr(s2), ptr(t2), 1, ptr(p3s), ptr(p3q),
                                        1, M, 1, MID, C, 1, 1, 0, None, None, None, 0, st))

    def fin3():
        check(L.capnet_bn_finalize(ptr(p3s), ptr(p3q), t3, C, M, ptr(gamma), ptr(beta), None, None, 0.1, 1e-5, ptr(sc), ptr(sh), st))

    def tail():
        check(L.capnet_conv1x1_fwd_tail(ptr(y3), ptr(sc), ptr(sh), ptr(res), None, None, ptr(out), ptr(old1), bn1, ptr(y1),
                                        ptr(p1s), ptr(p1q), M, C, MID, st))
    ts, tf = timed(stats), timed(fused)
    t_c3, t_f3, t_tl = timed(conv3), timed(fin3), timed(tail)
    flops = 2.0 * M * C * MID * 2
    print("stage %dx%d MID %3d M %6d: stats %6.1f us + fused %6.1f us = %6.1f us (%.0f TF/s algorithmic) | conv3 %6.1f + finalize %5.1f + "
          "tail/conv1 %6.1f = %6.1f us" % (side, side, MID, M, ts, tf, ts + tf, flops / (ts + tf) / 1e6, t_c3, t_f3, t_tl,
                                         t_c3 + t_f3 + t_tl))
ops.check_device_errors()
