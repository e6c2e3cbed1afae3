"""train_autoencoder.py step time (classic AE and VAE) at the config batch (4) and the script's default (32)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.autoencoder import Autoencoder
from shapegan_amd.train_steps import AutoencoderTrainer
def timeit(fn, iters=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for variational in (False, True):
    for B in (4, 32, 256):
        torch.manual_seed(0)
        tr = AutoencoderTrainer(Autoencoder(is_variational=variational))
        x = (torch.rand(B, 32, 32, 32, device="cuda") * 2 - 1)
        print("%s B=%3d: %.3f ms/step  %.0f grids/s" % ("VAE" if variational else "AE ", B, timeit(lambda: tr.step(x)), B / timeit(lambda: tr.step(x)) * 1e3), flush=True)
