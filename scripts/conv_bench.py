"""Per-kernel timing of the conv implicit GEMMs at the WGAN (config 2) and progressive-D (config 4) layer shapes."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops

def _warm_gpu():
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(200): a = (a @ a) * 1e-4
    torch.cuda.synchronize()
_warm_gpu()
def timeit(fn, iters=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

which = sys.argv[1] if len(sys.argv) > 1 else "all"
shapes = [(64, 64, 128, 16), (64, 128, 256, 8), (64, 1, 64, 32), (16, 32, 64, 32), (16, 64, 128, 16)]
if which == "fwd1":
    shapes = shapes[:1]
for (B, Ci, Co, R) in shapes:
    x = torch.randn(B, Ci, R, R, R, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda")
    y = ops.conv_fwd_raw(x, w, b, 1, 0.2)
    dy = torch.randn_like(y)
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    t = timeit(lambda: ops.conv_fwd_raw(x, w, b, 1, 0.2)); print("fwd   B%d %d->%d @%d: %.3f ms %.1f TF" % (B, Ci, Co, R, t, flop / t / 1e9), flush=True)
    if which == "fwd1": break
    t = timeit(lambda: ops.conv_dgrad_raw(dy, w, None, Ci)); print("dgrad B%d %d->%d @%d: %.3f ms %.1f TF" % (B, Ci, Co, R, t, flop / t / 1e9), flush=True)
    t = timeit(lambda: ops.conv_wgrad_raw(dy, x, Ci)); print("wgrad B%d %d->%d @%d: %.3f ms %.1f TF" % (B, Ci, Co, R, t, flop / t / 1e9), flush=True)
