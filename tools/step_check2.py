import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream
L = lib(); dev = torch.device('cuda:0')
H, b = 16, 2
W = torch.arange(4 * H * H, dtype=torch.float32).reshape(4 * H, H) * 0.001
wf = torch.zeros(L.capnet_lstm_wfrag_floats(H), device=dev)
Wd = W.to(dev)
check(L.capnet_lstm_pack_wfrag(Wd.data_ptr(), wf.data_ptr(), H, 0, current_stream()))
torch.cuda.synchronize()
wfc = wf.cpu().reshape(2, 4, 64, 4)   # ug, wave, lane, e
print("wfrag[ug1,w0,lane0..3]:", wfc[1, 0, :4], "expect W[8,0]=", W[8, 0].item(), W[8, 2].item())
print("expect W[0,0],W[0,2]:", W[0, 0].item(), W[0, 2].item(), " lane32:", wfc[0, 0, 32], "expect", W[0, 1].item(), W[0, 3].item())
for k in range(H):
    hp = torch.zeros(b, H); hp[0, k] = 1.0
    G = torch.zeros(b, 4 * H, device=dev); cp = torch.zeros(b, H, device=dev)
    co, ho = torch.empty(b, H, device=dev), torch.empty(b, H, device=dev)
    hpd = hp.to(dev)
    check(L.capnet_lstm_step_fused(hpd.data_ptr(), wf.data_ptr(), G.data_ptr(), 4 * H, cp.data_ptr(), co.data_ptr(), ho.data_ptr(), b, H, 0, current_stream()))
    torch.cuda.synchronize()
    # gate i = sigmoid(pre) -> pre = logit
    gi = G[0, :H].cpu(); pre = torch.log(gi / (1 - gi))
    ref = W[:H, k]
    if k in (0, 5): print(k, "err per unit", [round(x, 4) for x in (pre - ref).abs().tolist()])
