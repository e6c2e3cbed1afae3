"""The one-channel kernels alone at the critic's shapes, a few cold calls each — target of counter passes:
    bash scripts/kernel_pmc.sh conv_fwd_c1 python scripts/edge_target.py fwd"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
nb = 128
w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
b1 = torch.zeros(64, device="cuda")
xs = [torch.randn(nb, 1, 32, 32, 32, device="cuda") for _ in range(6)]
keep = []
if what == "fwd":
    for x in xs:
        keep.append(ops.conv_fwd_raw(x, w1, b1, 1, 0.2))
elif what == "convT":
    ys = [torch.randn(64, 64, 16, 16, 16, device="cuda") for _ in range(8)]
    for y in ys:
        keep.append(ops.conv_dgrad_raw(y, w1, None, 1))
torch.cuda.synchronize()
