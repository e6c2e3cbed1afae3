import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
B, Ci, Co, R = 128, 64, 128, 16
dy = torch.randn(B, Co, R // 2, R // 2, R // 2, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
x = torch.randn(B, Ci, R, R, R, device="cuda"); b = torch.zeros(Co, device="cuda")
for _ in range(4):
    ops.conv_dgrad_halo_raw(dy, w, None, Ci)
    ops.conv_fwd_raw(x, w, b, 1, 0.2)
    ops.conv_wgrad_halo_raw(dy, x, Ci)
torch.cuda.synchronize()
