"""Micro-benchmark of the implicit-GEMM conv kernel on the ResNet-152 layer shapes (B=64).
    python tools/probes/conv_bench.py [--tile CODE] [--iters N] [--shapes s3c1,s3c2,...]
Prints TFLOP/s per shape; run under rocprofv3 --pmc for counters."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402
from capnet import ops  # noqa: E402
from capnet._lib import check, current_stream, lib, ptr  # noqa: E402

SHAPES = {  # name: (H, Cin, Cout, k, stride)
    "s1c1": (56, 256, 64, 1, 1), "s1c2": (56, 64, 64, 3, 1), "s1c3": (56, 64, 256, 1, 1),
    "s2c1": (28, 512, 128, 1, 1), "s2c2": (28, 128, 128, 3, 1), "s2c3": (28, 128, 512, 1, 1),
    "s3c1": (14, 1024, 256, 1, 1), "s3c2": (14, 256, 256, 3, 1), "s3c3": (14, 256, 1024, 1, 1),
    "s4c1": (7, 2048, 512, 1, 1), "s4c2": (7, 512, 512, 3, 1), "s4c3": (7, 512, 2048, 1, 1),
    "s2c2s": (56, 128, 128, 3, 2), "s3c2s": (28, 256, 256, 3, 2), "s4c2s": (14, 512, 512, 3, 2),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--shapes", default=",".join(SHAPES))
    ap.add_argument("--no-pre", action="store_true")
    ap.add_argument("--no-tail", action="store_true", help="disable the K-sliced tail balancing")
    ap.add_argument("--h3", action="store_true", help="split-f16 kernel (1x1 and 3x3 shapes with Cin % 64 == 0)")
    ap.add_argument("--v1", action="store_true", help="row-major-weight kernel (conv_f32.hip)")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    B = args.batch
    tot_f = tot_t = 0.0
    for name in args.shapes.split(","):
        H, Cin, Cout, k, stride = SHAPES[name]
        pad = 1 if k == 3 else 0
        OH = (H + 2 * pad - k) // stride + 1
        M = B * OH * OH
        x = torch.randn(B, H, H, Cin, device=dev)
        w = torch.randn(Cout, Cin, k, k, device=dev) * 0.05
        Kw = (k * k * Cin + 15) // 16 * 16
        wp = ops.pack_conv_weight(w, Kw, kmajor=not args.v1)
        y = torch.empty(M, Cout, device=dev)
        tiles = max(lib().capnet_conv_tiles_m(M, Cout, args.tile),
                    lib().capnet_conv_kmajor_tiles_m(M, Cout, Kw, args.tile))
        slabs = torch.empty(max(1, lib().capnet_conv_kmajor_slab_floats(M, Cout, Kw, args.tile)), device=dev)
        ps = torch.empty(tiles, Cout, device=dev)
        pq = torch.empty(tiles, Cout, device=dev)
        sc = torch.rand(Cin, device=dev) + 0.5
        sh = torch.randn(Cin, device=dev)

        use_h3 = args.h3 and Cin % 64 == 0 and Cout % 64 == 0 and k in (1, 3)
        if use_h3:
            bn = 128 if Cout % 128 == 0 else 64
            img = ops.pack_conv_weight_f16x3(w, bn)
            t1 = lib().capnet_conv1x1_tiles_m(M)
            ps = torch.empty(t1, Cout, device=dev)
            pq = torch.empty(t1, Cout, device=dev)

        # (the split-f16 kernel keeps a folded input's scale / shift in LDS: Cin <= 512; the wider 1x1 inputs of the trunk
        #  arrive activated, from the tail-absorbing conv1)
        no_pre = args.no_pre or (use_h3 and Cin > 512)

        def run():
            if use_h3:
                check(lib().capnet_conv2d_fwd_f16x3(ptr(x), H * H * Cin, H * Cin, Cin, ptr(img), bn, ptr(y),
                                                    None if no_pre else ptr(sc), None if no_pre else ptr(sh),
                                                    0 if no_pre else 1, ptr(ps), ptr(pq), B, H, H, Cin, Cout, k, stride, pad,
                                                    None, None, None, 0, current_stream()))
                return
            if not args.v1:
                check(lib().capnet_conv2d_fwd_kmajor(ptr(x), H * H * Cin, H * Cin, Cin, ptr(wp), Kw, ptr(y),
                                                     None if args.no_pre else ptr(sc), None if args.no_pre else ptr(sh),
                                                     0 if args.no_pre else 1, ptr(ps), ptr(pq), B, H, H, Cin, Cout,
                                                     k, k, stride, pad, args.tile,
                                                     None if args.no_tail else ptr(slabs), current_stream()))
                return
            check(lib().capnet_conv2d_fwd(ptr(x), H * H * Cin, H * Cin, Cin, 1, ptr(wp), Kw, ptr(y),
                                          None if args.no_pre else ptr(sc), None if args.no_pre else ptr(sh),
                                          0 if args.no_pre else 1, ptr(ps), ptr(pq), B, H, H, Cin, Cout,
                                          k, k, stride, pad, args.tile, current_stream()))
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / args.iters
        fl = 2.0 * M * Cout * k * k * Cin
        tot_f += fl
        tot_t += us
        print("%s M=%6d N=%4d K=%5d  %8.1f us  %6.1f TF/s" % (name, M, Cout, k * k * Cin, us, fl / us / 1e6))
    print("sum: %.1f us, %.1f TF/s" % (tot_t, tot_f / tot_t / 1e6))


if __name__ == "__main__":
    main()
