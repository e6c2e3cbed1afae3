"""The two LDS-halo input-gradient kernels alone at the WGAN step's shapes — target of counter passes (FETCH_SIZE / WRITE_SIZE /
MFMA busy) and of A/B timing with SHAPEGAN_HIP_LIB variants:   python scripts/dgrad_target.py [time]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402

w0 = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
w1 = torch.randn(256, 128, 4, 4, 4, device="cuda") * 0.02
shapes = [(n, 128, 8, w0, 64) for n in (64, 128, 256)] + [(n, 256, 4, w1, 128) for n in (64, 128, 256)] + [(16, 256, 4, w1, 128), (32, 256, 4, w1, 128)]
out = {}
for n, co, o, w, cin in shapes:
    dy = torch.randn(n, co, o, o, o, device="cuda")
    fn = lambda: ops.conv_dgrad_raw(dy, w, None, cin)
    if len(sys.argv) > 1 and sys.argv[1] == "time":
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.2:
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record(); torch.cuda.synchronize()
        us = a.elapsed_time(b) / 20 * 1e3
        out["%dx%d@%d^3" % (n, co, o)] = {"us": round(us, 1), "frac_f32_mfma": round(2.0 * co * cin * 64 * o ** 3 * n / (us * 1e-6) / 157.3e12, 4)}
    else:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
print(json.dumps(out))
