"""A/B timing of the SDFNet kernels on one box (GPU): fused inference forward at 8 x 32^3 and 16 x 64^3, the 200 000-point /
latent-256 training step, alternating several rounds.   SHAPEGAN_HIP_LIB=<variant> python scripts/sdf_ab.py"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.sdf_net import SDFNet
from shapegan_amd.train_steps import SDFAutoDecoderTrainer
from shapegan_amd.util import get_voxel_coordinates


def t_ms(fn, iters):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


torch.manual_seed(0)
net = SDFNet()
grid = torch.tensor(get_voxel_coordinates(32)).cuda().repeat((8, 1))
z = torch.randn(8, 128, device="cuda")
pc, shapes, lat, npts = 200000, 64, 256, 200000
pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), torch.randn(shapes, lat, device="cuda") * 1e-2, pts, sdf, pointcloud_size=pc)
idx = torch.randint(0, shapes * pc, (npts,), device="cuda")
out = {"lib": os.path.basename(os.environ.get("SHAPEGAN_HIP_LIB", "default"))}
with torch.no_grad():
    out["infer_8x32^3_ms"] = round(t_ms(lambda: net.forward_shapes(grid, z, 32768), 20), 4)
out["train_200k_L256_ms"] = round(t_ms(lambda: tr.step(idx), 20), 4)
print(json.dumps(out))
