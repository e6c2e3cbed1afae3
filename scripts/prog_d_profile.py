"""Config 4 (hybrid progressive WGAN-GP, 64^3, B=16): discriminator step only, for rocprofv3 --stats."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.progressive_gan import Discriminator as ProgD
from shapegan_amd.model.sdf_net import SDFNet
from shapegan_amd.train_steps import HybridProgressiveGANTrainer
from shapegan_amd.util import get_voxel_coordinates
torch.manual_seed(0)
it, B = 3, 16
R = 8 * 2 ** it
g, d = SDFNet(), ProgD().cuda(); d.set_iteration(it)
tr = HybridProgressiveGANTrainer(g, d, torch.tensor(get_voxel_coordinates(R)).cuda(), R)
real = torch.rand(B, R, R, R, device="cuda") * 2 - 1; z = torch.randn(B, 128, device="cuda"); alpha = torch.rand(B, 1, 1, 1, device="cuda")
with torch.no_grad():
    fake = tr.generate(z)
tr.generate = lambda zz: fake          # profile the discriminator work only
for _ in range(4):
    tr.discriminator_step(real, z, alpha)
torch.cuda.synchronize()
