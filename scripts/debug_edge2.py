import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from shapegan_amd import ops
torch.manual_seed(0)
N = 64
x = torch.randn(N, 1, 32, 32, 32, device="cuda")
w = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
ref = F.conv3d(x.cpu(), w.cpu(), None, stride=2, padding=1).cuda()
for trial in range(4):
    y = ops.conv_fwd_raw(x, w, None)
    bad = ((y - ref).abs() > 1e-4)
    idx = bad.nonzero()
    print("trial", trial, "fwd alone: mismatches", int(bad.sum()), idx[:5].tolist() if len(idx) else "")
dy = torch.randn(N, 64, 16, 16, 16, device="cuda")
for trial in range(3):
    y = ops.conv_fwd_raw(x, w, None)
    dw = ops.conv_wgrad_raw(dy, x, 1)
    bad = ((y - ref).abs() > 1e-4)
    print("trial", trial, "fwd then wgrad: mismatches", int(bad.sum()))
