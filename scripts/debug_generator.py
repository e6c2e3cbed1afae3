"""Diagnostic: full-tensor comparison HIP vs CPU oracle (fp32 and fp64) for the Generator at B=3 (golden case)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch_oracle as O
from shapegan_amd.model.gan import Generator
z = np.load("tests/golden/modules.npz")
torch.manual_seed(11)
g = Generator(); g.train()
sd = {k: v.detach().cpu().clone() for k, v in g.state_dict().items()}
zin = torch.from_numpy(z["generator/in0"])
w = torch.randn(3, 1, 32, 32, 32, generator=torch.Generator().manual_seed(99))
def run_oracle(dtype):
    P = O.clone_state({k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()})
    out = O.generator_forward(P, zin.to(dtype), True)
    (out * w.to(dtype)).sum().backward()
    return out.detach(), {k: v.grad for k, v in P.items() if v.requires_grad}
o32, g32 = run_oracle(torch.float32)
o64, g64 = run_oracle(torch.float64)
out = g(zin.cuda())
(out * w.cuda()).sum().backward()
print("fwd max abs err hip-vs-64 %.3e  ref32-vs-64 %.3e" % ((out.detach().cpu().double() - o64).abs().max(), (o32.double() - o64).abs().max()))
for k, p in g.named_parameters():
    h = p.grad.detach().cpu().double(); r64 = g64[k]; r32 = g32[k].double()
    eh, er = (h - r64).abs(), (r32 - r64).abs()
    scale = r64.abs().mean()
    bad = eh > 1e-4 * scale + 1e-4 * r64.abs()
    print("%-18s scale %.3e  hip max err %.3e (%.2e rel)  ref32 max err %.3e  bad %d/%d  ref32-bad-at-same %d" % (
        k, scale, eh.max(), eh.max() / scale, er.max(), int(bad.sum()), bad.numel(), int((bad & (er > 1e-4 * scale)).sum())))
