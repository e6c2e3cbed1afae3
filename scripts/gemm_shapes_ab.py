"""sg_gemm at the PointNet GAN's Linear shapes (196 608 points x 256 x 256): forward x W^T, input gradient g W, weight gradient
g^T x — fraction of the fp32 MFMA peak.  A library built with -DSG_GEMM128=0 (scripts/ab_build.sh x gemm.hip -DSG_GEMM128=0; SHAPEGAN_HIP_LIB=scripts/_abl/x.so) times the generic skeleton instead."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402


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
    return a.elapsed_time(b) / iters * 1e3


out = {"lib": os.path.basename(os.environ.get("SHAPEGAN_HIP_LIB", "default"))}
for P in (196608, 32768):
    x = torch.randn(P, 256, device="cuda")
    g = torch.randn(P, 256, device="cuda")
    w = torch.randn(256, 256, device="cuda") * 0.05
    b = torch.zeros(256, device="cuda")
    flop = 2.0 * P * 256 * 256
    for name, fn in (("fwd  x W^T", lambda: ops.gemm_raw(x, False, w, True, bias_j=b)),
                     ("dgrad g W", lambda: ops.gemm_raw(g, False, w, False)),
                     ("wgrad g^T x", lambda: ops.gemm_raw(g, True, x, False))):
        us = t_us(fn)
        out["%s P=%d" % (name, P)] = {"us": round(us, 1), "frac_f32_mfma": round(flop / (us * 1e-6) / 157.3e12, 3)}
print(json.dumps(out))
