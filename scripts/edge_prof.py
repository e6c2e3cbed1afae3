import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
nb = 128
x32 = torch.randn(nb, 1, 32, 32, 32, device="cuda"); w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
b1 = torch.zeros(64, device="cuda"); y16 = torch.randn(nb, 64, 16, 16, 16, device="cuda"); y16g = y16[:64].contiguous()
for _ in range(20):
    ops.conv_fwd_raw(x32, w1, b1, 1, 0.2); ops.conv_wgrad_raw(y16, x32, 1); ops.conv_dgrad_raw(y16g, w1, None, 1)
torch.cuda.synchronize()
