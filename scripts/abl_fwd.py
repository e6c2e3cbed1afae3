import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
def _warm_gpu():
    a = torch.randn(4096, 4096, device="cuda")
    for _ in range(200): a = (a @ a) * 1e-4
    torch.cuda.synchronize()
_warm_gpu()
def timeit(fn, iters=40):
    for _ in range(15): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
variants = [(int(v), "dbg%s" % v) for v in sys.argv[1:]] or [(32, "8w"), (16, "4w")]
for (B, Ci, Co, R) in [(64, 64, 128, 16), (128, 64, 128, 16), (16, 64, 128, 32), (64, 32, 64, 32)]:
    x = torch.randn(B, Ci, R, R, R, device="cuda"); w = torch.randn(Co, Ci, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(Co, device="cuda")
    flop = 2.0 * B * Co * (R // 2) ** 3 * Ci * 64
    for dbg, name in variants:
        t = timeit(lambda: ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, 1, dbg))
        print("%s B%d %d->%d@%d %s %.3f ms %.1f TF" % (os.environ.get("SHAPEGAN_HIP_LIB", "base")[-8:], B, Ci, Co, R, name, t, flop / t / 1e9), flush=True)
