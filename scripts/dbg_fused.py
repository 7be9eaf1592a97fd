import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, mau_amd as mau
from mau_amd import functional as F_
prec = os.environ.get("PREC", "fp16")
g = torch.Generator().manual_seed(91)
Hh, Ww = 64, 48
x, ts, md = torch.randn(2, 6, Hh, Ww, generator=g).cuda(), torch.randn(2, 10, generator=g).cuda(), torch.randn(2, 4, generator=g).cuda()
tgt = torch.randn(2, 2, Hh, Ww, generator=g).cuda()
def run(fused_bn, fused_up):
    F_._FUSED_BN, F_._FUSED_UP = fused_bn, fused_up
    torch.manual_seed(90)
    net = mau.UrbanPredictor("unet", 6, 10, 16, 4, 16, 24, 2, base_filters=16, temporal_embeddings=False, metadata_embeddings=True).cuda().set_precision(prec).train()
    out = net(x, ts, md); loss = mau.compute_loss_mse(out, tgt)["total"]; loss.backward()
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
ref = run(False, False)
for name, cfg in (("all fused", (True, True)), ("fused, no up", (True, False))):
    got = run(*cfg)
    bad = [(k, float((got[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in ref if not torch.equal(got[k], ref[k])]
    print(name, "differing:", len(bad), "of", len(ref))
    for k, d, m in bad[-12:]: print("   ", k, d, m)
