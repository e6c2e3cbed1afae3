"""A/B timing of the one-channel conv kernels (GPU box): each C-ABI entry timed with HIP events at the critic's shapes.
    SHAPEGAN_HIP_LIB=<variant .so> python scripts/edge_ab.py"""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops


def t_us(fn, iters=30):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


nb = 128
x32 = torch.randn(nb, 1, 32, 32, 32, device="cuda")
w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
b1 = torch.zeros(64, device="cuda")
y16 = torch.randn(nb, 64, 16, 16, 16, device="cuda")
yact = torch.randn(nb, 64, 16, 16, 16, device="cuda")
y16g = y16[:64].contiguous()
out = {"lib": os.environ.get("SHAPEGAN_HIP_LIB", "default"),
       "conv1_fwd_128": round(t_us(lambda: ops.conv_fwd_raw(x32, w1, b1, 1, 0.2)), 1),
       "conv1_wgrad_128": round(t_us(lambda: ops.conv_wgrad_raw(y16, x32, 1)), 1),
       "conv1_wgrad_act_128": round(t_us(lambda: ops.conv_wgrad_act_raw(y16, yact, x32, 1, 0.2)), 1),
       "convT_fwd_64": round(t_us(lambda: ops.conv_dgrad_raw(y16g, w1, None, 1)), 1)}
for n2 in (32, 64, 256, 512):
    xs = torch.randn(n2, 1, 32, 32, 32, device="cuda")
    ys = torch.randn(n2, 64, 16, 16, 16, device="cuda")
    out["conv1_fwd_%d" % n2] = round(t_us(lambda: ops.conv_fwd_raw(xs, w1, b1, 1, 0.2)), 1)
    out["conv1_wgrad_%d" % n2] = round(t_us(lambda: ops.conv_wgrad_raw(ys, xs, 1)), 1)
    if n2 <= 256:
        out["convT_fwd_%d" % n2] = round(t_us(lambda: ops.conv_dgrad_raw(ys, w1, None, 1)), 1)
    del xs, ys
print(json.dumps(out))
