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
for (B, Ci, Co, R) in [(64, 64, 128, 16), (128, 64, 128, 16), (64, 128, 256, 8), (64, 64, 128, 16)]:
    dy = torch.randn(B, Co, R // 2, R // 2, R // 2, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    ref = None
    for impl in (1, 3):
        t = timeit(lambda: ops.conv_dgrad_halo_raw(dy, w, None, Ci, impl=impl))
        out = ops.conv_dgrad_halo_raw(dy, w, None, Ci, impl=impl)
        if ref is None: ref = out
        err = ((out - ref).abs().max() / ref.abs().max()).item()
        print("dgrad B%d %d->%d@%d impl%d %.3f ms %.1f TF (diff %.1e)" % (B, Ci, Co, R, impl, t, flop / t / 1e9, err), flush=True)
