"""20 launches of the persistent LSTM kernel over 24 steps at b = 64 (for the TCC counter passes)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream, int_array
L = lib(); dev = torch.device('cuda:0')
H, b, T, n = 512, 64, 24, 20
W = torch.randn(4 * H, H, device=dev) * 0.05
img = torch.empty(L.capnet_lstm_persist_w_floats(), device=dev)
check(L.capnet_lstm_persist_pack(W.data_ptr(), img.data_ptr(), 0, current_stream()))
G = torch.randn(T * b, 4 * H, device=dev)
Cst = torch.zeros(T * b, H, device=dev); hid = torch.zeros(T * b, H, device=dev)
ctls = torch.zeros(n, L.capnet_lstm_persist_ctl_ints(), dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
bs = int_array([b] * T)
for k in range(n):
    check(L.capnet_lstm_persist_run(img.data_ptr(), G.data_ptr(), Cst.data_ptr(), hid.data_ptr(), bs, 0, T, H, 0, 1,
                                    ctls[k].data_ptr(), err.data_ptr(), None, current_stream()))
torch.cuda.synchronize()
assert int(err.item()) == 0
