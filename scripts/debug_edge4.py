import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from shapegan_amd import ops
torch.manual_seed(0)
for (N, Co, R) in ((2, 32, 64), (3, 32, 64), (16, 64, 32)):
    x = torch.randn(N, 1, R, R, R, device="cuda")
    w = torch.randn(Co, 1, 4, 4, 4, device="cuda") * 0.1
    b = torch.randn(Co, device="cuda") * 0.1
    ref = F.leaky_relu(F.conv3d(x.double(), w.double(), b.double(), stride=2, padding=1), 0.2)
    y = ops.conv_fwd_raw(x, w, b, 1, 0.2)
    err = (y.double() - ref).abs()
    print(N, Co, R, "max err", float(err.max()), "per-channel max:", [round(float(v) * 1e6, 2) for v in err.amax(dim=(0, 2, 3, 4))[:Co]])
