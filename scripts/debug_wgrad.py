import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.nn.functional as F
from shapegan_amd import ops, lib as L
torch.manual_seed(0)
for (N, Ci, Co, R) in ((2, 32, 64, 32), (3, 32, 64, 32), (2, 64, 128, 16)):
    x = torch.randn(N, Ci, R, R, R, device="cuda"); dy = torch.randn(N, Co, R // 2, R // 2, R // 2, device="cuda")
    ref = torch.nn.grad.conv3d_weight(x.double(), (Co, Ci, 4, 4, 4), dy.double(), stride=2, padding=1)
    for big in (False, True):
        L._workspaces.clear()
        if big:
            L.workspace("splitk", 64 << 20, x.device)      # a large cached scratch, as after an earlier big call
        dw = ops.conv_wgrad_raw(dy, x, Ci)
        err = (dw.double() - ref).abs()
        print(N, Ci, Co, R, "big ws" if big else "fresh ws", "max err %.3e" % float(err.max()), "scale %.3e" % float(ref.abs().mean()),
              "bad co:", (err.amax(dim=(1, 2, 3, 4)) > 1e-3).nonzero().flatten().tolist()[:10])
    try:
        dwh = ops.conv_wgrad_halo_raw(dy, x, Ci)
        print("   forced halo: max err %.3e" % float((dwh.double() - ref).abs().max()))
    except RuntimeError as e:
        print("   forced halo: not eligible", str(e)[:60])
