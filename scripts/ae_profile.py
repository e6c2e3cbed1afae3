import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.autoencoder import Autoencoder
from shapegan_amd.train_steps import AutoencoderTrainer
torch.manual_seed(0)
tr = AutoencoderTrainer(Autoencoder(is_variational=False))
x = (torch.rand(4, 32, 32, 32, device="cuda") * 2 - 1)
for _ in range(10): tr.step(x)
torch.cuda.synchronize()
