"""Sustained vs burst rate of the forward halo kernel (power / clock behaviour under a long run)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
x = torch.randn(128, 64, 16, 16, 16, device="cuda"); w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
b = torch.zeros(128, device="cuda")
flop = 2.0 * 128 * 128 * 512 * 64 * 64
fn = lambda: ops.conv_fwd_raw(x, w, b, 1, 0.2)
for _ in range(5): fn()
torch.cuda.synchronize()
for chunk in range(12):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(400): fn()
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 400
    print("chunk %2d (%.0f ms of work): %.3f ms  %.1f TF" % (chunk, t * 400, t, flop / t / 1e9), flush=True)
