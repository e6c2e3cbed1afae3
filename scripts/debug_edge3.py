import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from shapegan_amd import ops
torch.manual_seed(0)
for (N, Co, R) in ((74, 64, 32), (64, 64, 32), (2, 32, 64), (3, 32, 64), (16, 32, 64), (17, 64, 32), (33, 24, 32)):
    x = torch.randn(N, 1, R, R, R, device="cuda")
    w = torch.randn(Co, 1, 4, 4, 4, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda") * 0.1
    for (bias, act) in ((None, 0), (b, 0), (b, 1)):
        ref = F.conv3d(x, w, bias, stride=2, padding=1)
        if act: ref = F.leaky_relu(ref, 0.2)
        y = ops.conv_fwd_raw(x, w, bias, act, 0.2)
        bad = ((y - ref).abs() > 1e-4)
        idx = bad.nonzero()
        print(N, Co, R, "bias" if bias is not None else "nobias", "act", act, "mismatches", int(bad.sum()), "of", y.numel(),
              "samples:", sorted(set(idx[:, 0].tolist()))[:10] if len(idx) else "", "chan:", sorted(set(idx[:, 1].tolist()))[:8] if len(idx) else "")
