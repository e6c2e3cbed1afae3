"""gemm128p_kernel at the PointNet GAN's forward shape, a few calls — counter target: bash scripts/kernel_pmc.sh gemm128 python scripts/gemm_target.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402
P = 196608
x = torch.randn(P, 256, device="cuda"); w = torch.randn(256, 256, device="cuda") * 0.05; b = torch.zeros(256, device="cuda")
for _ in range(6):
    ops.gemm_raw(x, False, w, True, bias_j=b)
    ops.gemm_raw(x, False, w, False)
torch.cuda.synchronize()
