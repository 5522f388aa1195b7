"""Phase timing of the persistent LSTM sequence kernel (diagnostic build path: s_memtime stamps
written to a buffer nothing else reads). Per step and workgroup: start -> h_{t-1} flags seen ->
MFMAs done -> K-reduced (LDS, barrier) -> stores + flag issued."""
import sys
import torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream, int_array
L = lib(); dev = torch.device('cuda:0')
H = 512
b = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W = torch.randn(4 * H, H, device=dev) * 0.05
img = torch.empty(L.capnet_lstm_persist_w_floats(), device=dev)
check(L.capnet_lstm_persist_pack(W.data_ptr(), img.data_ptr(), 0, current_stream()))
G0 = torch.randn(T * b, 4 * H, device=dev)
Cst = torch.empty(T * b, H, device=dev); hid = torch.empty(T * b, H, device=dev)
ctl = torch.zeros(L.capnet_lstm_persist_ctl_ints(), dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
st = torch.zeros(T * 256 * 8 + 512, dtype=torch.int64, device=dev)
bs = int_array([b] * T)
# the product instantiation (no stamps), timed with events on the launch stream
for rep in range(6):
    G = G0.clone(); ctl.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(L.capnet_lstm_persist_run(img.data_ptr(), G.data_ptr(), Cst.data_ptr(), hid.data_ptr(), bs, 0, T, H, 0, 1,
                                    ctl.data_ptr(), err.data_ptr(), None, current_stream()))
    e1.record()
    torch.cuda.synchronize()
    if rep >= 4:
        print("product kernel: launch of %d steps %.1f us = %.2f us per step" % (T, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e1) * 1e3 / T))
# a burst of back-to-back product launches: does the per-launch time change as the clock ramps?
nb = 300
evs = [torch.cuda.Event(enable_timing=True) for _ in range(nb + 1)]
G = G0.clone()
evs[0].record()
for k in range(nb):
    ctl.zero_()
    check(L.capnet_lstm_persist_run(img.data_ptr(), G.data_ptr(), Cst.data_ptr(), hid.data_ptr(), bs, 0, T, H, 0, 1,
                                    ctl.data_ptr(), err.data_ptr(), None, current_stream()))
    evs[k + 1].record()
torch.cuda.synchronize()
dt = [evs[k].elapsed_time(evs[k + 1]) * 1e3 for k in range(nb)]
print("burst of %d launches (us each, incl. the memset between): first 5 %s ... 100-105 %s ... last 5 %s" %
      (nb, [round(x, 1) for x in dt[:5]], [round(x, 1) for x in dt[100:105]], [round(x, 1) for x in dt[-5:]]))
for rep in range(4):
    G = G0.clone(); ctl.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(L.capnet_lstm_persist_run(img.data_ptr(), G.data_ptr(), Cst.data_ptr(), hid.data_ptr(), bs, 0, T, H, 0, 1,
                                    ctl.data_ptr(), err.data_ptr(), st.data_ptr(), current_stream()))
    e1.record()
    torch.cuda.synchronize()
    clk = st.cpu()[T * 256 * 8:].reshape(256, 2).double()
    s = st.cpu()[:T * 256 * 8].reshape(T, 256, 8).double()
    d = (s[1:-1, :, 1:6] - s[1:-1, :, 0:5]).mean((0, 1))
    step = (s[2:, :, 0] - s[1:-1, :, 0]).mean()
    print("b %d, %d steps: launch %.1f us; cycles per step %.0f = poll %.0f + h loads, mfma %.0f + LDS exchange %.0f + "
          "gates, h store, ack %.0f + flag %.0f (+ loop, stamps) ; shader clock %.0f MHz ; err %d"
          % (b, T, e0.elapsed_time(e1) * 1e3, step, d[0], d[1], d[2], d[3], d[4],
             (clk[:, 0] / clk[:, 1]).median() * 100.0, int(err.item())))
    # per-step wait by shard (is one shard late?)
    if rep == 3:
        w = (s[1:, :, 1] - s[1:, :, 0]).mean(0).reshape(32, 8).mean(0)
        print("mean wait per shard:", [int(x) for x in w.tolist()])
        print("mode per shard (1 local, 2 safe):", ctl[1281:1289].tolist())
        print("step 0 (no product): %.0f cycles" % (s[0, :, 5] - s[0, :, 0]).mean())
