import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
def timeit(fn, iters=40):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for (B, Ci, Co, R) in [(128, 128, 256, 8), (64, 128, 256, 8), (128, 128, 256, 8)]:
    x = torch.randn(B, Ci, R, R, R, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda")
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    for impl in (0, 1):
        t = timeit(lambda: ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, impl, 0))
        print("fwd B%d %d->%d@%d impl%d %.3f ms %.1f TF" % (B, Ci, Co, R, impl, t, flop / t / 1e9), flush=True)

for (B, Ci, Co, R) in [(128, 128, 256, 8), (64, 128, 256, 8), (128, 128, 256, 8)]:
    x = torch.randn(B, Ci, R, R, R, device="cuda"); dy = torch.randn(B, Co, R // 2, R // 2, R // 2, device="cuda")
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    t = timeit(lambda: ops.conv_wgrad_halo_raw(dy, x, Ci))
    print("wgrad halo4 B%d %d->%d@%d %.3f ms %.1f TF" % (B, Ci, Co, R, t, flop / t / 1e9), flush=True)
