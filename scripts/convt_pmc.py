"""Counter target: the ConvTranspose3d(64 -> 1) forward alone at 64 / 32 / 256 samples (scripts/convt_pmc.sh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
for nb in (64, 32, 256):
    y = torch.randn(nb, 64, 16, 16, 16, device="cuda")
    for _ in range(6):
        ops.conv_dgrad_raw(y, w1, None, 1)
    torch.cuda.synchronize()
