"""How many streams run concurrently: N streams, one single-workgroup spin kernel each (torch.cuda._sleep), total time against
one kernel's. GPU_MAX_HW_QUEUES=8 python tools/probes/hw_queues.py   (the runtime reads the variable at start)"""
import os
import time

import torch

dev = torch.device("cuda:0")
torch.cuda._sleep(1000)
torch.cuda.synchronize()
CYC = 20_000_000


def run(streams):
    best = 1e9
    for _ in range(4):                      # (the first round pays for the streams' queues)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in streams:
            with torch.cuda.stream(s):
                torch.cuda._sleep(CYC)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) * 1e3)
    return best


one = run([torch.cuda.Stream()])
print("GPU_MAX_HW_QUEUES=%s  one spin kernel: %.2f ms" % (os.environ.get("GPU_MAX_HW_QUEUES"), one))
for n in (2, 3, 4, 5, 6, 8):
    ss = [torch.cuda.Stream() for _ in range(n)]
    t = run(ss)
    print("  %d normal streams: %.2f ms = %.1f kernels deep" % (n, t, t / one))
for n in (3, 4):
    ss = [torch.cuda.Stream() for _ in range(n)] + [torch.cuda.Stream(priority=-1)]
    t = run(ss)
    print("  %d normal + 1 high-priority stream: %.2f ms = %.1f deep" % (n, t, t / one))
ss = [torch.cuda.Stream() for _ in range(3)] + [torch.cuda.Stream(priority=-1), torch.cuda.default_stream()]
print("  3 normal + 1 high + the default stream: %.1f deep" % (run(ss) / one))
ss = [torch.cuda.Stream() for _ in range(4)] + [torch.cuda.Stream(priority=-1), torch.cuda.default_stream()]
print("  4 normal + 1 high + the default stream: %.1f deep" % (run(ss) / one))
