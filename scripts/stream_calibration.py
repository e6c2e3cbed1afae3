"""What this box's HBM actually streams (GPU box): write-only, read-only, copy and mixed passes over 1 GiB tensors, timed with
HIP events on the launch stream.  The edge-layer argument of DESIGN.md 3.2 (achieved GB/s of the one-channel conv kernels
against what a pure stream of the same read / write mix reaches) rests on these numbers, so they are written as JSON:

    python scripts/stream_calibration.py > gpurun_out/stream_calibration.json      (copied to profiles/<round>_stream_calibration.json)
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import lib as L  # noqa: E402
from shapegan_amd.lib import check, ptr, stream  # noqa: E402

N = 1 << 28        # floats: 1 GiB per tensor


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    lib = L.load()
    x = torch.randn(N, device="cuda")
    y = torch.randn(N, device="cuda")
    out = torch.empty(N, device="cuda")
    scalar = torch.empty((), device="cuda")
    ws = torch.empty(lib.sg_reduce_workspace_bytes(), dtype=torch.uint8, device="cuda")
    gib = float(N * 4)
    rows = []

    def add(name, kernel, read, write, fn):
        us = timed(fn)
        rows.append({"pass": name, "kernel": kernel, "read_bytes": read, "write_bytes": write, "us": round(us, 1),
                     "tb_per_s": round((read + write) / us / 1e6, 3)})

    add("write-only 1 GiB", "ATen fill (vectorized_elementwise_kernel<FillFunctor>)", 0.0, gib, lambda: out.fill_(1.5))
    add("write-only 1 GiB", "hipMemsetAsync (tensor.zero_)", 0.0, gib, lambda: out.zero_())
    add("read-only 1 GiB", "sg::sum_partial_kernel (sg_reduce_sum)", gib, 0.0,
        lambda: check(lib.sg_reduce_sum(ptr(x), ptr(scalar), N, 1.0, ptr(ws), ws.numel(), stream()), "reduce_sum"))
    add("read-only 1 GiB", "ATen sum", gib, 0.0, lambda: torch.sum(x))
    add("copy 1 GiB -> 1 GiB", "__amd_rocclr_copyBuffer (tensor.copy_)", gib, gib, lambda: out.copy_(x))
    add("1 read + 1 write (b128)", "sg::act_fwd_kernel (LeakyReLU)", gib, gib,
        lambda: check(lib.sg_act_fwd(ptr(x), ptr(out), N, 1, 0.2, stream()), "act_fwd"))
    add("2 reads + 1 write (b32)", "sg::axpby_kernel", 2 * gib, gib,
        lambda: check(lib.sg_axpby(ptr(x), ptr(y), ptr(out), N, 1.0, 1.0, stream()), "axpby"))
    add("2 reads + 1 write", "sg::act_bwd_kernel", 2 * gib, gib,
        lambda: check(lib.sg_act_bwd(ptr(x), ptr(y), ptr(out), N, 1, 0.2, stream()), "act_bwd"))
    # the sizes the edge layers actually move (L2 / MALL effects included): 134 MB written (conv1 forward output at 128 samples),
    # 151 MB read (its weight gradient), 67 MB read + 8 MB written (fused ConvT at 64 samples)
    small_w = torch.empty(128 * 64 * 4096, device="cuda")
    add("write-only 134 MB (conv1 forward's output)", "ATen fill", 0.0, small_w.numel() * 4.0, lambda: small_w.fill_(0.5))
    small_r = torch.randn(128 * 64 * 4096 + 128 * 32768, device="cuda")
    add("read-only 151 MB (conv1 weight gradient's operands)", "sg::sum_partial_kernel", small_r.numel() * 4.0, 0.0,
        lambda: check(lib.sg_reduce_sum(ptr(small_r), ptr(scalar), small_r.numel(), 1.0, ptr(ws), ws.numel(), stream()), "reduce_sum"))
    small_c = torch.randn(64 * 64 * 4096, device="cuda")
    add("read-only 67 MB (fused ConvT's dy at 64 samples)", "sg::sum_partial_kernel", small_c.numel() * 4.0, 0.0,
        lambda: check(lib.sg_reduce_sum(ptr(small_c), ptr(scalar), small_c.numel(), 1.0, ptr(ws), ws.numel(), stream()), "reduce_sum"))
    # bias-gradient row sums at the shapes of BASELINE configs[3] (progressive discriminator, batch 32 at 64^3) and of the WGAN critic
    for nrows, length, what in ((32 * 32, 32768, "configs[3] stage 3: [32,32,32^3]"), (32 * 64, 4096, "configs[3] stage 2: [32,64,16^3]"),
                                (32 * 128, 512, "configs[3] stage 1: [32,128,8^3]"), (128 * 128, 512, "WGAN critic layer 2: [128,128,8^3]"),
                                (128 * 256, 64, "WGAN critic layer 3: [128,256,4^3]")):
        g = torch.randn(nrows, length, device="cuda")
        o = torch.empty(nrows, device="cuda")
        add("row sums " + what, "sg::rowsum_kernel / rowsum_wave_kernel (sg_rowsum)", g.numel() * 4.0, nrows * 4.0,
            lambda: check(lib.sg_rowsum(ptr(g), ptr(o), nrows, length, length, stream()), "rowsum"))
    print(json.dumps({"device": torch.cuda.get_device_name(0), "hbm_peak_tb_per_s": 8.0, "passes": rows}, indent=1))


if __name__ == "__main__":
    main()
