"""csrc/gemm_b3.hip (three bf16 pieces, six products) against capnet_sgemm's f32-MFMA kernels on the decoders' large products:
time and error against float64.   python tools/probes/b3_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402,F401
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


SHAPES = [  # (ta, tb, M, N, K, what)
    (0, 1, 1344, 8192, 512, "logits = H C^T"),
    (0, 0, 1344, 512, 8192, "dH = dlogits C"),
    (1, 0, 8192, 512, 1344, "dC = dlogits^T H"),
    (0, 1, 1344, 2048, 512, "A2 / gates (N x 2048 x 512)"),
    (1, 0, 2048, 512, 1344, "dWcat = dPre^T Hprev"),
    (0, 0, 1344, 2048, 2048, "NN 1344 x 2048 x 2048"),
    (0, 1, 12544, 512, 2048, "encoder_att, 64 x 196 pixels"),
    (0, 1, 2352, 512, 2048, "encoder_att, 12 x 196 pixels"),
    (1, 0, 512, 2048, 12544, "dWe = datt1^T feat"),
    (0, 1, 2016, 4096, 1024, "stacked: N x 4F x F"),
]
for ta, tb, M, N, K, what in SHAPES:
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g).to(dev)
    B = torch.randn((N, K) if tb else (K, N), generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ref = ((A.double().t() if ta else A.double()) @ (B.double().t() if tb else B.double())) + bias.double()
    C1, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    ws = torch.empty(16 << 20, device=dev)
    st = current_stream()

    def f32():
        check(L.capnet_sgemm(ta, tb, M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1], ptr(C1), N, ptr(bias), 0, 1, 0, 0, 0, 0, 0, st))

    def b3():
        check(L.capnet_sgemm_b3(ta, tb, M, N, K, ptr(A), A.shape[1], ptr(B), B.shape[1], ptr(C2), N, ptr(bias), 0, 1, 0, 0, 0, 0, ptr(ws), ws.numel(), st))
    os.environ["CAPNET_NO_B3"] = "1"
    t1, t2 = timed(f32), timed(b3)
    sc = ref.abs().max().item()
    e1, e2 = (C1.double() - ref).abs().max().item() / sc, (C2.double() - ref).abs().max().item() / sc
    gf = 2.0 * M * N * K / 1e6
    print("%-34s %d%d %6d x %5d x %5d: f32 %7.1f us (%5.1f TF/s, err %.1e)   b3 %7.1f us (%5.1f TF/s, err %.1e)   x%.2f" %
          (what, ta, tb, M, N, K, t1, gf / t1, e1, t2, gf / t2, e2, t1 / t2))
