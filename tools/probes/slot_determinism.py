import sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att
dev = torch.device('cuda:0')
enc = model_att.EncoderCNN(14)
enc.load_state_dict(synthetic.trunk_state(enc.state_dict(), seed=1234))
enc.to(dev).train()
for B in (4, 8):
    imgs = synthetic.make_batch(B, 100, seed=3)[0].to(dev)
    f0 = enc(imgs)
    f0b = enc(imgs)
    f1, ap1 = enc(imgs, slot=1, defer_stats=True)
    f2, ap2 = enc(imgs, slot=2, defer_stats=True)
    f1b, _ = enc(imgs, slot=1, defer_stats=True)
    torch.cuda.synchronize()
    print("B", B, "slot0 twice equal:", torch.equal(f0, f0b), "| slot0 vs slot1:", torch.equal(f0, f1), (f0 - f1).abs().max().item(),
          "| slot1 vs slot2:", torch.equal(f1, f2), "| slot1 twice:", torch.equal(f1, f1b))
