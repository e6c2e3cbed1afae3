"""Does the roofline kernel run slower after the training steps (clock / power state) than in a fresh process?"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from shapegan_amd.model.gan import Discriminator, Generator
from shapegan_amd.train_steps import WGANTrainer
print("fresh      ", bench.roofline_conv()["achieved"], flush=True)
torch.manual_seed(0)
g, c = Generator(), Discriminator()
tr = WGANTrainer(g, c)
reals = [torch.rand(64, 32, 32, 32, device="cuda") * 2 - 1 for _ in range(5)]
zs = [torch.randn(64, 128, device="cuda") for _ in range(5)]
zg = torch.randn(64, 128, device="cuda")
for _ in range(25): tr.step(reals, zs, zg)
torch.cuda.synchronize()
print("after steps", bench.roofline_conv()["achieved"], flush=True)
print("again      ", bench.roofline_conv()["achieved"], flush=True)
del tr, g, c, reals
torch.cuda.empty_cache()
print("after free ", bench.roofline_conv()["achieved"], flush=True)
