"""Configs 4 and 5 at full size: hybrid WGAN (B=8, 32^3) and hybrid progressive GAN iteration 3 (B=16, 64^3, WGAN-GP)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.gan import Discriminator
from shapegan_amd.model.progressive_gan import Discriminator as ProgD
from shapegan_amd.model.sdf_net import SDFNet
from shapegan_amd.train_steps import HybridWGANTrainer, HybridProgressiveGANTrainer
from shapegan_amd.util import get_voxel_coordinates
def timeit(fn, iters=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3
torch.manual_seed(0)
g, c = SDFNet(), Discriminator()
tr = HybridWGANTrainer(g, c, torch.tensor(get_voxel_coordinates(32)).cuda())
real = (torch.rand(8, 32, 32, 32, device="cuda") * 0.2 - 0.1); z = torch.randn(8, 128, device="cuda")
print("config5 critic step %.2f ms, generator step %.2f ms" % (timeit(lambda: tr.critic_step(real, z)), timeit(lambda: tr.generator_step(z))), flush=True)
for it, B in ((2, 16), (3, 16)):
    R = 8 * 2 ** it
    g, d = SDFNet(), ProgD().cuda(); d.set_iteration(it)
    tr = HybridProgressiveGANTrainer(g, d, torch.tensor(get_voxel_coordinates(R)).cuda(), R)
    real = torch.rand(B, R, R, R, device="cuda") * 2 - 1; z = torch.randn(B, 128, device="cuda"); alpha = torch.rand(B, 1, 1, 1, device="cuda")
    tg = timeit(lambda: tr.generator_step(z)); td = timeit(lambda: tr.discriminator_step(real, z, alpha))
    print("config4 it=%d (%d^3, B=%d): generator step %.2f ms, discriminator+GP step %.2f ms, peak mem %.1f GB" % (it, R, B, tg, td, torch.cuda.max_memory_allocated() / 2**30), flush=True)
