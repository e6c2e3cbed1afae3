"""A/B timing of the two Conv3d forward implementations + halo-kernel ablations (debug flags)."""
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
for (B, Ci, Co, R) in [(64, 64, 128, 16), (16, 64, 128, 32), (16, 32, 64, 32)]:
    x = torch.randn(B, Ci, R, R, R, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda")
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    for impl, dbg, name in ((0, 0, "gather"), (1, 0, "halo auto"), (1, 32, "halo 8w"), (1, 16, "halo 4w"), (1, 48, "halo 64rows"),
                            (1, 32 + 8, "halo 8w spread"), (1, 16 + 8, "halo 4w spread")):
        t = timeit(lambda: ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, impl, dbg))
        if impl == 1:
            ref = ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, 0, 0); got = ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, impl, dbg)
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            assert err < 1e-5, (name, err)
        print("B%d %d->%d @%d %-16s %.3f ms  %.1f TF" % (B, Ci, Co, R, name, t, flop / t / 1e9), flush=True)
