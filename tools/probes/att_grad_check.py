"""Diagnostic: attention decoder gradients on trunk features, GPU vs CPU oracle fed the SAME features."""
import random, sys, torch
sys.path.insert(0, '/root/repo')
import capnet
from capnet import synthetic, model_att, ops
from oracle import decoders_ref as D
torch.set_num_threads(16)
dev = torch.device('cuda:0')
V, B = 8192, 12
enc = model_att.EncoderCNN(14)
enc.load_state_dict(synthetic.trunk_state(enc.state_dict(), seed=1234))
dec = model_att.DecoderFactoredLSTMAtt(512, 300, 512, 512, V, 1, dropout=0.0)
p = synthetic.decoder_state(dec.state_dict(), seed=1234)
dec.load_state_dict(p)
imgs, captions, lengths = synthetic.make_batch(B, V, seed=0)
enc.to(dev).train(); dec.to(dev).train()
feats = enc(imgs.to(dev))
print("features: shape", tuple(feats.shape), "max %.3f mean %.4f frac zero %.3f" % (feats.max().item(), feats.mean().item(), (feats == 0).float().mean().item()))
lens1 = [l - 1 for l in lengths]
tf = [True] * max(lens1)
targets = D.packed_targets(captions[:, 1:], lens1)
out, alphas = dec(captions[:, :-1].contiguous().to(dev), lens1, feats, tf_mask=tf)
loss = ops.cross_entropy(out, targets.to(dev)) + ((1.0 - alphas.sum(dim=1)) ** 2).mean()
loss.backward()
leaves = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
lo, al = D.factored_att_forward(leaves, captions[:, :-1], lens1, feats.cpu(), tf, "factual")
lref = D.att_loss(lo, al, targets, 1.0)
lref.backward()
print("loss gpu %.7f cpu %.7f" % (loss.item(), lref.item()))
print("logits rel %.2e alphas rel %.2e" % ((out.cpu() - lo).abs().max() / lo.abs().max(), (alphas.cpu() - al).abs().max() / al.abs().max()))
rows = []
for k, prm in dec.named_parameters():
    gr = leaves[k].grad
    if gr is None:
        continue
    g = prm.grad.cpu()
    err = (g - gr).abs()
    rms = gr.pow(2).mean().sqrt().item()
    rows.append((err.max().item() / (gr.abs().max().item() + 1e-30), (err.pow(2).mean().sqrt().item()) / (rms + 1e-30), k, rms, gr.abs().max().item()))
for r in sorted(rows, reverse=True)[:14]:
    print("max-rel %.2e rms-rel %.2e  %-32s rms %.2e max %.2e" % r)
