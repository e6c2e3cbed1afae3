"""Classic autoencoder step (train_autoencoder.py, BASELINE configs[0] shape) at small batches: eager vs one captured graph."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.autoencoder import Autoencoder
from shapegan_amd.train_steps import AutoencoderTrainer
def timeit(fn, iters=30):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
for B in (4, 32, 256):
    x = torch.rand(B, 32, 32, 32, device="cuda") * 2 - 1
    torch.manual_seed(0); tr = AutoencoderTrainer(Autoencoder(is_variational=False))
    torch.manual_seed(0); trg = AutoencoderTrainer(Autoencoder(is_variational=False), capturable=True)
    print("batch %3d: eager %.3f ms/step, graphed %.3f ms/step" % (B, timeit(lambda: tr.step(x)), timeit(lambda: trg.step_graphed(x))), flush=True)
