import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet._lib import lib, check, current_stream
L = lib()
dev = torch.device('cuda:0')
for H, b in [(16, 4), (16, 64), (16, 32), (32, 4), (32, 64), (64, 3), (96, 40), (96, 17), (128, 33), (256, 64), (512, 64), (512, 9)]:
    g = torch.Generator().manual_seed(H + b)
    hp = torch.randn(b, H, generator=g); W = torch.randn(4 * H, H, generator=g) * 0.2
    G0 = torch.randn(b, 4 * H, generator=g); cp = torch.randn(b, H, generator=g)
    pre = G0.double() + hp.double() @ W.double().t()
    i, f, o, gt = [pre[:, k * H:(k + 1) * H] for k in range(4)]
    i, f, o, gt = torch.sigmoid(i), torch.sigmoid(f), torch.sigmoid(o), torch.tanh(gt)
    c = f * cp.double() + i * gt; h = o * c
    hpd, Wd, Gd, cpd = hp.to(dev), W.to(dev), G0.to(dev), cp.to(dev)
    wf = torch.zeros(L.capnet_lstm_wfrag_floats(H), device=dev)
    check(L.capnet_lstm_pack_wfrag(Wd.data_ptr(), wf.data_ptr(), H, 0, current_stream()))
    co, ho = torch.empty(b, H, device=dev), torch.empty(b, H, device=dev)
    check(L.capnet_lstm_step_fused(hpd.data_ptr(), wf.data_ptr(), Gd.data_ptr(), 4 * H, cpd.data_ptr(), co.data_ptr(), ho.data_ptr(), b, H, 0, current_stream()))
    torch.cuda.synchronize()
    print(H, b, "h err", (ho.cpu().double() - h).abs().max().item(), "c err", (co.cpu().double() - c).abs().max().item(), "gate err", (Gd.cpu().double() - torch.cat([i, f, o, gt], 1)).abs().max().item())
