"""SDFNet fused forward (inference, per-shape latents) at 8 x 32^3 and 16 x 64^3 points."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.sdf_net import SDFNet
from shapegan_amd.util import get_voxel_coordinates
net = SDFNet().cuda()
for S, R in ((8, 32), (16, 64)):
    grid = torch.tensor(get_voxel_coordinates(R)).cuda().repeat((S, 1))
    z = torch.randn(S, 128, device="cuda")
    with torch.no_grad():
        for _ in range(5): net.forward_shapes(grid, z, R ** 3)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): net.forward_shapes(grid, z, R ** 3)
        e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    n = S * R ** 3
    print("fwd %d x %d^3: %.3f ms  %.1f Mpoints/s  executed %.1f TF" % (S, R, ms, n / ms / 1e3, n * 790016 / ms / 1e9), flush=True)
