"""A/B of the parities-per-workgroup choice of conv_dgrad_halo_kernel<0> at the generator's shapes (GPU box)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops


def t_us(fn, iters=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.2:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return round(a.elapsed_time(b) / iters * 1e3, 1)


out = {}
w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
for n in (64, 128, 256):
    dy = torch.randn(n, 128, 8, 8, 8, device="cuda")
    out["auto_%d" % n] = t_us(lambda: ops.conv_dgrad_raw(dy, w, None, 64))
    for ppw in (1, 2, 4, 8):
        out["ppw%d_%d" % (ppw, n)] = t_us(lambda: ops.conv_dgrad_halo_raw(dy, w, None, 64, impl=(3 if ppw == 1 else 1 + 4 * ppw)))
w4 = torch.randn(256, 128, 4, 4, 4, device="cuda") * 0.02
for n in (64, 256):
    dy = torch.randn(n, 256, 4, 4, 4, device="cuda")
    out["d1_auto_%d" % n] = t_us(lambda: ops.conv_dgrad_raw(dy, w4, None, 128))
    for ppw in (1, 2, 4, 8):
        out["d1_ppw%d_%d" % (ppw, n)] = t_us(lambda: ops.conv_dgrad_halo_raw(dy, w4, None, 128, impl=(3 if ppw == 1 else 1 + 4 * ppw)))
print(json.dumps(out))
