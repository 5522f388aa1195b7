"""Phase timing of the fused LSTM step kernel (diagnostic; s_memtime stamps)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream
L = lib(); dev = torch.device('cuda:0')
H, b = 512, 64
hp = torch.randn(b, H, device=dev) * 0.1; W = torch.randn(4 * H, H, device=dev) * 0.05
G = torch.randn(b, 4 * H, device=dev); cp = torch.randn(b, H, device=dev)
wf = torch.empty(L.capnet_lstm_wfrag_floats(H), device=dev)
check(L.capnet_lstm_pack_wfrag(W.data_ptr(), wf.data_ptr(), H, 0, current_stream()))
co, ho = torch.empty(b, H, device=dev), torch.empty(b, H, device=dev)
st = torch.zeros(H // 4 * 2 * 5, dtype=torch.int64, device=dev)
for rep in range(4):
    check(L.capnet_lstm_step_fused_stamped(hp.data_ptr(), wf.data_ptr(), G.data_ptr(), 4 * H, cp.data_ptr(), co.data_ptr(), ho.data_ptr(), b, H, st.data_ptr(), current_stream()))
    torch.cuda.synchronize()
    s = st.cpu().reshape(-1, 5).double()
    d = (s[:, 1:] - s[:, :-1]).mean(0)
    print("cycles: stage %.0f  mfma %.0f  reduce %.0f  epilogue %.0f  total %.0f ; spread of starts %.0f" % (d[0], d[1], d[2], d[3], (s[:, 4] - s[:, 0]).mean(), s[:, 0].max() - s[:, 0].min()))
