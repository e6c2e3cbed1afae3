"""Auto dispatch against the forced LDS-halo kernels at the small batches of BASELINE configs[3] / [4] (GPU box):
the critic of train_hybrid_wgan.py sees 16 samples of 32^3, the progressive discriminator 32 samples of 64^3 .. 8^3."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops


def t_us(fn, iters=20):
    try:
        fn()
    except RuntimeError as e:
        return "n/a (%s)" % str(e)[-40:]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return round(a.elapsed_time(b) / iters * 1e3, 1)



def main():
    out = {}
    for n in (16, 32):
        for (ci, co, r) in ((64, 128, 16), (128, 256, 8), (32, 64, 32), (16, 32, 64)):
            if n * ci * r ** 3 * 4 > (1 << 31):
                continue
            x = torch.randn(n, ci, r, r, r, device="cuda")
            w = torch.randn(co, ci, 4, 4, 4, device="cuda") * 0.02
            dy = torch.randn(n, co, r // 2, r // 2, r // 2, device="cuda")
            key = "n%d_%dto%d_%d" % (n, ci, co, r)
            gf = 2.0 * n * (r // 2) ** 3 * co * ci * 64 / 1e9
            out[key] = {"gflop": round(gf, 2),
                        "fwd_auto": t_us(lambda: ops.conv_fwd_raw(x, w, None)), "fwd_halo": t_us(lambda: ops.conv_fwd_impl_raw(x, w, None, 0, 0.0, 1)),
                        "dgrad_auto": t_us(lambda: ops.conv_dgrad_raw(dy, w, None, ci)),
                        "dgrad_halo_ppw1": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w, None, ci, impl=3)),
                        "dgrad_halo_ppw2": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w, None, ci, impl=9)),
                        "dgrad_halo": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w, None, ci, impl=1)),
                        "wgrad_auto": t_us(lambda: ops.conv_wgrad_raw(dy, x, ci)), "wgrad_halo": t_us(lambda: ops.conv_wgrad_halo_raw(dy, x, ci))}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
