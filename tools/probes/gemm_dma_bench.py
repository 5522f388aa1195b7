"""Throughput of the LDS-DMA NT core (csrc/gemm_dma.hip) on the trunk's 1x1 shapes and the vocabulary
projection, next to the kernels it replaces."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet
from capnet import ops
from capnet._lib import check, current_stream, lib, ptr
dev = torch.device("cuda:0")
L = lib()

def timed(fn, iters=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

B = 64
SH = {"s1c1": (56, 256, 64, 1), "s1c3": (56, 64, 256, 1), "s2c1": (28, 512, 128, 1), "s2c3": (28, 128, 512, 1),
      "s3c1": (14, 1024, 256, 1), "s3c3": (14, 256, 1024, 1), "s4c1": (7, 2048, 512, 1), "s4c3": (7, 512, 2048, 1),
      "s2ds": (56, 256, 512, 2), "s3ds": (28, 512, 1024, 2), "s4ds": (14, 1024, 2048, 2)}
tot = [0.0, 0.0, 0.0]
for name, (H, Cin, Cout, stride) in SH.items():
    OH = (H - 1) // stride + 1
    M = B * OH * OH
    x = torch.randn(B, H, H, Cin, device=dev)
    w = torch.randn(Cout, Cin, device=dev) * 0.05
    y = torch.empty(M, Cout, device=dev)
    t = L.capnet_conv1x1_tiles_m(M)
    ps, pq = torch.empty(t, Cout, device=dev), torch.empty(t, Cout, device=dev)
    us = timed(lambda: check(L.capnet_conv1x1_fwd_dma(ptr(x), H * H * Cin, H * Cin, Cin, ptr(w), ptr(y), ptr(ps), ptr(pq),
                                                      B, H, H, Cin, Cout, stride, None, None, None, 0, current_stream())))
    # old kernel, no prologue
    wk = ops.pack_conv_weight(w.reshape(Cout, Cin, 1, 1), Cin, kmajor=True)
    tiles = L.capnet_conv_kmajor_tiles_m(M, Cout, Cin, 0)
    ps2, pq2 = torch.empty(tiles, Cout, device=dev), torch.empty(tiles, Cout, device=dev)
    slabs = torch.empty(max(1, L.capnet_conv_kmajor_slab_floats(M, Cout, Cin, 0)), device=dev)
    us_old = timed(lambda: check(L.capnet_conv2d_fwd_kmajor(ptr(x), H * H * Cin, H * Cin, Cin, ptr(wk), Cin, ptr(y), None, None, 0,
                                                            ptr(ps2), ptr(pq2), B, H, H, Cin, Cout, 1, 1, stride, 0, 0, ptr(slabs), current_stream())))
    fl = 2.0 * M * Cout * Cin
    tot[0] += fl; tot[1] += us; tot[2] += us_old
    print("%s M=%6d N=%4d K=%4d s%d: dma %7.1f us %6.1f TF/s | k-major %7.1f us %6.1f TF/s" % (name, M, Cout, Cin, stride, us, fl / us / 1e6, us_old, fl / us_old / 1e6))
print("sum: dma %.1f us %.1f TF/s | k-major %.1f us %.1f TF/s" % (tot[1], tot[0] / tot[1] / 1e6, tot[2], tot[0] / tot[2] / 1e6))
for (M, N, K) in ((999, 8192, 512), (1037, 8192, 512), (2352, 512, 2048), (12544, 512, 2048)):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    us = timed(lambda: check(L.capnet_sgemm_nt_dma(M, N, K, ptr(A), K, ptr(W), ptr(out), ptr(b), current_stream())))
    us_old = timed(lambda: check(L.capnet_sgemm(0, 1, M, N, K, ptr(A), K, ptr(W), K, ptr(out), N, ptr(b), 0, 1, 0, 0, 0, 0, 128, current_stream())))
    fl = 2.0 * M * N * K
    print("gemm %5d x %5d x %4d: dma %7.1f us %6.1f TF/s | gemm_f32<128,128> %7.1f us %6.1f TF/s" % (M, N, K, us, fl / us / 1e6, us_old, fl / us_old / 1e6))
