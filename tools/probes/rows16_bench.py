"""gemm_rows16_kernel (csrc/gemm_f32.hip) on the step products of a 12-image attention batch, against the two-launch
path (skinny kernel + slab reduce): us per call in a back-to-back loop, same box.
    python tools/probes/rows16_bench.py"""
import os, subprocess, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SHAPES = [(12, 4608, 512, 1), (12, 2048, 2348, 1), (12, 512, 4608, 0), (12, 2348, 2048, 0), (12, 8192, 512, 1), (12, 512, 512, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch, capnet
    from capnet._lib import lib, check, ptr, current_stream
    dev = torch.device("cuda:0")
    fused = sys.argv[2] == "1"
    out = {}
    for M, N, K, tb in SHAPES:
        A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev) if tb else torch.randn(K, N, device=dev)
        C = torch.zeros(M, N, device=dev); ws = torch.empty(40 * 16 * 8192, device=dev); ctr = torch.zeros(1024, dtype=torch.int32, device=dev)
        def run():
            if fused:
                check(lib().capnet_sgemm_splitk_fused(0, tb, M, N, K, ptr(A), K, ptr(B), B.shape[1], ptr(C), N, None, 0, ptr(ws), ws.numel(), ptr(ctr), 1024, current_stream()))
            else:
                check(lib().capnet_sgemm_splitk(0, tb, M, N, K, ptr(A), K, ptr(B), B.shape[1], ptr(C), N, None, 0, ptr(ws), ws.numel(), current_stream()))
        for _ in range(20): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): run()
        e1.record(); torch.cuda.synchronize()
        out["%dx%dx%d %s" % (M, N, K, "TB" if tb else "NN")] = round(e0.elapsed_time(e1) * 1e3 / 200, 2)
    print(json.dumps(out))
    sys.exit(0)
def child(fused):
    o = subprocess.run([sys.executable, __file__, "child", "1" if fused else "0"], capture_output=True, text=True)
    return json.loads(o.stdout.strip().splitlines()[-1])
print("two launches (skinny + reduce):", child(False))
print("one launch (rows16):           ", child(True))
