"""rocprofv3 target: N critic (or generator) updates of train_point_gan.py at one (num_points, batch) stage.
usage: point_gan_prof.py critic|generator [P B [iters]]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
from shapegan_amd.train_steps import PointGANTrainer

which = sys.argv[1] if len(sys.argv) > 1 else "critic"
P, B = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 12)
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 10
torch.manual_seed(0)
G, D = SDFGenerator(128, 256, 8, True).cuda(), PointNet(1).cuda()
tr = PointGANTrainer(G, D)
u = torch.cat([torch.rand(B, P, 3) * 2 - 1, torch.rand(B, P, 1) * 0.2 - 0.1], -1).cuda()
z, a = torch.randn(B, 128, device="cuda"), torch.rand(B, 1, 1, device="cuda")
step = (lambda: tr.critic_step(u, z, a)) if which == "critic" else (lambda: tr.generator_step(u, z))
for _ in range(iters):
    step()
torch.cuda.synchronize()
