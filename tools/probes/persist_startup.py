"""Cost of one launch of the persistent LSTM kernel as a function of its number of steps: bursts of
back-to-back launches (no host sync in between), events around the burst."""
import sys
import torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream, int_array
L = lib(); dev = torch.device('cuda:0')
H, b = 512, int(sys.argv[1]) if len(sys.argv) > 1 else 64
W = torch.randn(4 * H, H, device=dev) * 0.05
img = torch.empty(L.capnet_lstm_persist_w_floats(), device=dev)
check(L.capnet_lstm_persist_pack(W.data_ptr(), img.data_ptr(), 0, current_stream()))
Tmax = 49
G = torch.randn(Tmax * b, 4 * H, device=dev)
Cst = torch.zeros(Tmax * b, H, device=dev); hid = torch.zeros(Tmax * b, H, device=dev)
nb = 200
ctls = torch.zeros(nb, L.capnet_lstm_persist_ctl_ints(), dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
bs = int_array([b] * Tmax)
for t0, T in ((0, 1), (1, 2), (1, 3), (1, 5), (1, 9), (1, 25), (1, 49)):
    res = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctls.zero_()
        e0.record()
        for k in range(nb):      # a fresh (zeroed) control block per launch: flags of an earlier launch must not pass
            check(L.capnet_lstm_persist_run(img.data_ptr(), G.data_ptr(), Cst.data_ptr(), hid.data_ptr(), bs, t0, T, H, 0,
                                            1, ctls[k].data_ptr(), err.data_ptr(), None, current_stream()))
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / nb)
    print("steps [%d, %d): %.2f us per launch (%d recurrent steps), err %d" % (t0, T, min(res), T - max(t0, 1) + (0 if t0 else 0), int(err.item())))
