"""The trunk's stride-1 3x3 shapes at B = 64: patch-in-LDS kernel against the implicit-GEMM split-f16 kernel.
    python tools/probes/patch_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr, current_stream
dev = torch.device("cuda:0"); L = lib()
B = 64
for side, ch in ((56, 64), (28, 128), (14, 256), (7, 512)):
    M = B * side * side
    bn = L.capnet_conv1x1_f16x3_bn(M, ch)
    bnp = 256 if (ch % 256 == 0 and not os.environ.get('P3_NARROW')) else bn
    x = torch.randn(M, ch, device=dev); w = torch.randn(ch, ch, 3, 3, device=dev) * 0.05
    sc = torch.rand(ch, device=dev) + 0.5; sh = torch.randn(ch, device=dev)
    img = ops.pack_conv_weight_f16x3(w, bn)
    imgp = img if bnp == bn else ops.pack_conv_weight_f16x3(w, bnp)
    tiles = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(tiles, ch, device=dev), torch.empty(tiles, ch, device=dev)
    y1, y2 = torch.empty(M, ch, device=dev), torch.empty(M, ch, device=dev)
    def patch():
        check(L.capnet_conv3x3_fwd_patch(ptr(x), ptr(imgp), bnp, ptr(y1), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq), B, side, side, ch, ch, int(os.environ.get("P3_SHARED", "0")), current_stream()))
    def gemm():
        check(L.capnet_conv2d_fwd_f16x3(ptr(x), side * side * ch, side * ch, ch, ptr(img), bn, ptr(y2), ptr(sc), ptr(sh), 1, ptr(ps), ptr(pq),
                                        B, side, side, ch, ch, 3, 1, 1, None, None, None, 0, current_stream()))
    res = []
    for f in (patch, gemm):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / 20)
    print("%2d x %2d x %3d (bn %3d): patch %.1f us   implicit GEMM %.1f us   max |diff| %.2e of %.2e" %
          (side, side, ch, bn, res[0], res[1], (y1 - y2).abs().max().item(), y2.abs().max().item()))
