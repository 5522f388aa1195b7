"""A block's tail fused into the next conv1 (capnet_conv1x1_fwd_tail) alone on the chip at B = 64, against the two
launches it replaces (capnet_bn_add_relu + capnet_conv1x1_fwd_f16x3).   python tools/probes/tail_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, lib, ptr, current_stream
dev = torch.device("cuda:0"); L = lib(); B = 64
for side, cin, cout in ((56, 256, 64), (28, 512, 128), (14, 1024, 256), (7, 2048, 512)):
    M = B * side * side
    bn = L.capnet_conv1x1_f16x3_bn(M, cout)
    bn_t = 256 if (cout % 256 == 0 and not os.environ.get('TAIL_NARROW')) else bn
    y3 = torch.randn(M, cin, device=dev); res = torch.randn(M, cin, device=dev)
    s1 = torch.rand(cin, device=dev) + 0.5; t1 = torch.randn(cin, device=dev)
    w = torch.randn(cout, cin, 1, 1, device=dev) * 0.05
    img = ops.pack_conv_weight_f16x3(w, bn)
    img_t = img if bn_t == bn else ops.pack_conv_weight_f16x3(w, bn_t)
    tiles = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(tiles, cout, device=dev), torch.empty(tiles, cout, device=dev)
    out = torch.empty(M, cin, device=dev); y = torch.empty(M, cout, device=dev)
    def fused():
        check(L.capnet_conv1x1_fwd_tail(ptr(y3), ptr(s1), ptr(t1), ptr(res), None, None, ptr(out), ptr(img_t), bn_t, ptr(y), ptr(ps), ptr(pq), M, cin, cout, current_stream()))
    def two():
        check(L.capnet_bn_add_relu(ptr(y3), ptr(s1), ptr(t1), ptr(res), None, None, ptr(out), M, cin, current_stream()))
        check(L.capnet_conv1x1_fwd_f16x3(ptr(out), side * side * cin, side * cin, cin, ptr(img), bn, ptr(y), None, None, 0, ptr(ps), ptr(pq),
                                         B, side, side, cin, cout, 1, None, None, None, 0, current_stream()))
    r = []
    for f in (fused, two):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        r.append(e0.elapsed_time(e1) * 1e3 / 20)
    traffic = (3 * M * cin + M * cout) * 4 / 1e6
    print("%2d x %2d  %4d -> %3d: fused %.1f us (%.0f MB = %.2f TB/s)   bn_add_relu + conv1 %.1f us" % (side, side, cin, cout, r[0], traffic, traffic / r[0], r[1]))
