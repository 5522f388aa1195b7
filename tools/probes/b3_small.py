"""gemm_b3_kernel against the f32 kernels on batched products with few 128 x 128 tiles (the stacked decoder's chains over one run
of steps): where the dispatch threshold of capnet_sgemm should sit.   python tools/probes/b3_small.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import capnet  # noqa: E402,F401
from capnet._lib import check, current_stream, lib, ptr  # noqa: E402

dev = torch.device("cuda:0")
L = lib()


def timed(fn, reps=30):
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


os.environ["CAPNET_NO_B3"] = "1"
for M in (96, 192, 288, 480, 960):
    for (N, K, batch, tb) in ((1024, 1024, 4, 1), (512, 1024, 4, 1), (512, 512, 4, 1), (4096, 512, 1, 1), (1024, 1024, 4, 0)):
        A = torch.randn(M, batch * K, device=dev)
        B = torch.randn(batch, N, K, device=dev) if tb else torch.randn(batch, K, N, device=dev)
        C1, C2 = torch.empty(M, batch * N, device=dev), torch.empty(M, batch * N, device=dev)
        st = current_stream()
        ldb = K if tb else N

        def f32():
            check(L.capnet_sgemm(0, tb, M, N, K, ptr(A), batch * K, ptr(B), ldb, ptr(C1), batch * N, None, 0, batch, K, N * K, N, 0, 0, st))

        def b3():
            check(L.capnet_sgemm_b3(0, tb, M, N, K, ptr(A), batch * K, ptr(B), ldb, ptr(C2), batch * N, None, 0, batch, K, N * K, N, 0, None, 0, st))
        t1, t2 = timed(f32), timed(b3)
        tiles = ((M + 127) // 128) * ((N + 127) // 128) * batch
        print("M %4d N %4d K %4d batch %d tb %d: tiles %4d  f32 %6.1f us  b3 %6.1f us  x%.2f  (diff %.1e)" %
              (M, N, K, batch, tb, tiles, t1, t2, t1 / t2, (C1 - C2).abs().max().item() / C1.abs().max().item()))
