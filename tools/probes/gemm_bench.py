"""Micro-benchmark of the plain GEMM kernel on the decoder's all-rows shapes (N = 1037 packed rows).
    python tools/probes/gemm_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402
from capnet._lib import check, current_stream, lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
L = lib()
# name: (ta, tb, M, N, K, batch)
SHAPES = {
    "chain S/U fwd (NT b4)": (0, 1, 1037, 512, 512, 4),
    "chain V fwd (NT)": (0, 1, 1037, 2048, 300, 1),
    "dA2/dA1 (NN b4)": (0, 0, 1037, 512, 512, 4),
    "dX (NN)": (0, 0, 1037, 300, 2048, 1),
    "dH = dlogits.C (NN)": (0, 0, 1037, 512, 8192, 1),
    "vocab fwd (NT)": (0, 1, 1037, 8192, 512, 1),
    "dC (TN)": (1, 0, 8192, 512, 1037, 1),
    "dU/dS (TN b4)": (1, 0, 512, 512, 1037, 4),
    "dW (TN)": (1, 0, 2048, 512, 1037, 1),
}
for name, (ta, tb, M, N, K, batch) in SHAPES.items():
    A = torch.randn(batch, K, M, device=dev) if ta else torch.randn(batch, M, K, device=dev)
    B = torch.randn(batch, N, K, device=dev) if tb else torch.randn(batch, K, N, device=dev)
    Cc = torch.empty(batch, M, N, device=dev)
    res = []
    for tile in (64, 6432, 128):
        def run():
            check(L.capnet_sgemm(ta, tb, M, N, K, ptr(A), A.shape[2], ptr(B), B.shape[2], ptr(Cc), N, None, 0,
                                 batch, A.shape[1] * A.shape[2], B.shape[1] * B.shape[2], M * N, 0, tile,
                                 current_stream()))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append("%5d: %7.1f us %5.1f TF/s" % (tile, us, 2.0 * M * N * K * batch / us / 1e6))
    print("%-26s M=%5d N=%5d K=%5d b=%d | %s" % (name, M, N, K, batch, " | ".join(res)))
