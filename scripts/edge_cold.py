"""Cold and warm timing of the one-channel conv kernels at the critic's shapes (GPU box), for A/B of library variants:
    SHAPEGAN_HIP_LIB=<variant .so> python scripts/edge_cold.py [fwd|all]
cold = every call reads the next of K operand sets and writes a fresh output block, the rotation covering > 640 MB (the Infinity Cache
is 256 MB); warm = the same buffers again and again (what rounds 1 - 4 reported)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402


def t_us(fn, iters):
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
    return round(a.elapsed_time(b) / iters * 1e3, 1)


def both(nbytes, make, call):
    k = max(4, int((640 << 20) // nbytes) + 1)
    sets, alive, st = [make() for _ in range(k)], [None] * k, {"i": 0}

    def cold():
        i = st["i"] = (st["i"] + 1) % k
        alive[i] = call(sets[i])
    c = t_us(cold, 4 * k)
    w = t_us(lambda: call(sets[0]), 20)
    return {"cold_us": c, "warm_us": w, "cold_frac_8TBs": round(nbytes / (c * 1e-6) / 8e12, 3)}


what = sys.argv[1] if len(sys.argv) > 1 else "all"
w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
b1 = torch.zeros(64, device="cuda")
out = {"lib": os.path.basename(os.environ.get("SHAPEGAN_HIP_LIB", "default"))}
for nb in (() if what == 'convT' else (128, 64, 16) if what != 'all' else (128, 64)):
    mk_x = lambda: torch.randn(nb, 1, 32, 32, 32, device="cuda")
    mk_y = lambda: torch.randn(nb, 64, 16, 16, 16, device="cuda")
    nx, ny = nb * 32768, nb * 64 * 4096
    out["fwd_%d" % nb] = both(4.0 * (nx + ny), mk_x, lambda x: ops.conv_fwd_raw(x, w1, b1, 1, 0.2))
    if what == "all":
        out["wgrad_%d" % nb] = both(4.0 * (nx + ny), lambda: (mk_y(), mk_x()), lambda s: ops.conv_wgrad_raw(s[0], s[1], 1))
        out["wgrad_act_%d" % nb] = both(4.0 * (nx + 2 * ny), lambda: (mk_y(), mk_y(), mk_x()), lambda s: ops.conv_wgrad_act_raw(s[0], s[1], s[2], 1, 0.2))
        out["convT_%d" % nb] = both(4.0 * (nx + ny), mk_y, lambda y: ops.conv_dgrad_raw(y, w1, None, 1))
if what == "convT":
    for nb in (256, 64, 32, 16):
        mk_y = lambda: torch.randn(nb, 64, 16, 16, 16, device="cuda")
        out["convT_%d" % nb] = both(4.0 * (nb * 32768 + nb * 64 * 4096), mk_y, lambda y: ops.conv_dgrad_raw(y, w1, None, 1))
if what == "convT_forms":
    # every kernel form of sg_convT3d_k4s2p1_to1_pre_impl (identity input transform + tanh), cold and warm
    wt = torch.randn(64, 1, 4, 4, 4, device="cuda") / 23.0
    bt = torch.randn(1, device="cuda")
    sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
    for nb in (256, 128, 64, 48):
        mk_y = lambda: torch.randn(nb, 64, 16, 16, 16, device="cuda")
        for form in (1, 3, 5, 6, 7, 8):
            out["convT_%d_form%d" % (nb, form)] = both(4.0 * (nb * 32768 + nb * 64 * 4096), mk_y,
                                                       lambda y: ops.conv_transpose3d_to1_pre_raw(y, sc, sh, 1, 0.2, wt, bt, 3, 0.0, form=form))
print(json.dumps(out))
