import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_oracle as O
from shapegan_amd.model.progressive_gan import Discriminator
import numpy as np
torch.manual_seed(23)
d = Discriminator().cuda(); d.set_iteration(3); d.fade_in_progress = 0.5
sd = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "modules.npz"))
x = torch.from_numpy(z["progressive_it3_fade05/in0"])
B = x.shape[0]
print("input", tuple(x.shape), float(x.min()), float(x.max()))
wts = torch.randn((B,), generator=torch.Generator().manual_seed(99))
out = d(x.cuda()); (out * wts.cuda()).sum().backward()
P = O.clone_state({k: v.double() if v.is_floating_point() else v for k, v in sd.items()})
o = O.progressive_forward(P, x.double(), 3, 0.5); (o * wts.double()).sum().backward()
print("forward err", float((out.cpu().double() - o.detach()).abs().max()))
for k, p in d.named_parameters():
    if P[k].grad is None or p.grad is None: continue
    err = (p.grad.cpu().double() - P[k].grad).abs(); scale = float(P[k].grad.abs().mean())
    bad = err > 1e-3 * scale + 1e-9
    info = ""
    if p.grad.dim() == 5 and bad.any():
        info = "bad co: %s" % bad.flatten(1).any(1).nonzero().flatten().tolist()[:12]
    print("%-28s max err %.3e scale %.3e bad frac %.4f %s" % (k, float(err.max()), scale, float(bad.double().mean()), info))
