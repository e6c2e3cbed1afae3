import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from shapegan_amd import ops, lib as L
torch.manual_seed(0)
N, C, O = 64, 64, 16
dy = torch.randn(N, C, O, O, O, device="cuda")
w = torch.randn(C, 1, 4, 4, 4, device="cuda") * 0.1
out = ops.conv_dgrad_raw(dy, w, None, 1)
ref = F.conv_transpose3d(dy, w, None, stride=2, padding=1)
print("out err", float((out - ref).abs().max()), float(ref.abs().mean()))
ws = [v for k, v in L._workspaces.items() if k[2] == "dgrad"][0]
S = ws[: N * 64 * O ** 3 * 4].view(torch.float32).reshape(N, 64, O ** 3)
Sref = torch.einsum('ck,ncq->nkq', w.reshape(C, 64), dy.reshape(N, C, -1))
err = (S - Sref).abs()
print("S err", float(err.max()), "per-tap max err:", err.amax(dim=(0, 2))[:8].tolist())
print("S[0,:4,:4]", S[0, :4, :4].tolist()); print("Sref[0,:4,:4]", Sref[0, :4, :4].tolist())
