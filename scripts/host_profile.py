"""cProfile of the host side of the 20 000-point auto-decoder step (launch-bound: where do the ~26 us per launch go?)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.sdf_net import SDFNet
from shapegan_amd.train_steps import SDFAutoDecoderTrainer
torch.manual_seed(0)
pc, shapes, L = 200000, 64, 128
pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=L), torch.randn(shapes, L, device="cuda") * 1e-2, pts, sdf, pointcloud_size=pc)
idx = torch.randint(0, shapes * pc, (20000,), device="cuda")
for _ in range(20): tr.step(idx)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200): tr.step(idx)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
