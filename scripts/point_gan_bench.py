"""train_point_gan.py step times at the script's own (num_points, batch) stages: critic update (with GP) and generator update."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
from shapegan_amd.train_steps import PointGANTrainer

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
G, D = SDFGenerator(128, 256, 8, True).cuda(), PointNet(1).cuda()
tr = PointGANTrainer(G, D)
for P, B in ((1024, 32), (4096, 32), (16384, 12), (32768, 6)):
    u = torch.cat([torch.rand(B, P, 3) * 2 - 1, torch.rand(B, P, 1) * 0.2 - 0.1], -1).cuda()
    z, a = torch.randn(B, 128, device="cuda"), torch.rand(B, 1, 1, device="cuda")
    tr = PointGANTrainer(G, D)        # (fresh graphs per stage: every update is one captured graph launch)
    tc = timeit(lambda: tr.critic_step_graphed(u, z, a))
    tg = timeit(lambda: tr.generator_step_graphed(u, z))
    # FLOP per point, forward: SDFGenerator 2 (3 256 + 3 256^2 + 259 256 + 2 256^2 + 256) = 0.790 M, PointNet 2 (4 64 + 64 128
    # + 128 256 + 256 512) = 0.345 M.
    # REFERENCE arithmetic (dense autograd, train_point_gan.py:52-83): critic update = G forward + D(real) + D(fake) + D(interpolated)
    # forward, their weight / input gradients (2x a forward each) and the penalty's double backward through D (another 2x of one
    # pass): 0.790 + 0.345 (3 + 6 + 2) = 4.58 MFLOP / point; generator update 3 x (G + D) = 3.40 MFLOP / point.
    # EXECUTED arithmetic for clouds of >= 1024 points (the max's adjoint is sparse: everything behind the plain passes runs on the
    # 512 selected points of a cloud): critic update G + 3 D forward + (512 / P) x the rest; generator update G + D forward +
    # (512 / P) x (the recorded G and D passes and their backward).
    fc, fg = 4.58e6 * B * P, 3.40e6 * B * P
    sp = min(1.0, 512.0 / P) if P >= 1024 else 1.0
    ec = (0.790e6 + 3 * 0.345e6 + sp * 0.345e6 * (3 + 6 + 2)) * B * P if P >= 1024 else fc
    eg = (0.790e6 + 0.345e6 + sp * 3 * (0.790e6 + 0.345e6)) * B * P if P >= 1024 else fg
    print("P=%5d B=%2d  critic+GP %.2f ms = %.2f Mpoints/s (reference arithmetic %.1f TFLOP/s = %.3f of the f32 MFMA peak; executed %.1f "
          "TFLOP/s = %.3f)  generator %.2f ms = %.2f Mpoints/s (reference arithmetic %.1f TFLOP/s = %.3f; executed %.1f = %.3f)"
          % (P, B, tc, B * P / tc / 1e3, fc / tc / 1e9, fc / tc / 1e9 / 157.3, ec / tc / 1e9, ec / tc / 1e9 / 157.3,
             tg, B * P / tg / 1e3, fg / tg / 1e9, fg / tg / 1e9 / 157.3, eg / tg / 1e9, eg / tg / 1e9 / 157.3), flush=True)
