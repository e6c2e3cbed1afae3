#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on MI355X: GAN train steps/sec @32^3 voxels (+ SDFNet Mpoints/sec).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the train_wgan.py cadence over synthetic data that is already resident in HBM: five critic
updates (generator forward, two critic forward+backward, fused RMSprop + weight clip) and one generator update
(train_wgan.py:39,60-84), 32^3 voxel grids, batch 64 per GPU, fp32.  With N GPUs every rank runs the same per-GPU
workload (weak scaling) and each optimizer update all-reduces one flat gradient buffer over RCCL.
Rank 0 prints ONE JSON line (schema in the task contract) including `roofline` for the dominant kernel (the
64->128 channel Conv3d forward implicit GEMM, timed alone with HIP events on the launch stream), the SDFNet
Mpoints/s figures, and — at N=1 — `cpu_baseline`: the CPU oracle's time for the same step on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak (= fp32 vector peak)
BATCH = 64
# forward GFLOP of the dominant kernel: Conv3d 64->128, k4 s2 p1, 16^3 -> 8^3, per sample
CONV2_FLOP_PER_SAMPLE = 2.0 * 128 * 512 * 64 * 64


def event_time_ms(fn, iters):
    """Average duration of fn() in ms, measured with HIP events on the stream the kernels are launched on."""
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()   # warm up for ~0.2 s of work: the first launches pay code load and the clock ramp from idle
    while time.perf_counter() - t0 < 0.2:
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        fn()
    stop.record()
    torch.cuda.synchronize()
    return start.elapsed_time(stop) / iters


def roofline_conv():
    from shapegan_amd import ops
    # the critic runs fake and real as one 2*BATCH pass (5 of the step's 7 launches of this layer); the timed call is
    # the C-ABI entry, i.e. the weight-image pack (~1 % of the time) plus the MFMA kernel
    nb = 2 * BATCH
    x = torch.randn(nb, 64, 16, 16, 16, device="cuda")
    w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(128, device="cuda")
    ms = event_time_ms(lambda: ops.conv_fwd_raw(x, w, b, 1, 0.2), 30)
    flop = CONV2_FLOP_PER_SAMPLE * nb
    achieved = flop / (ms * 1e-3) / 1e12
    # HBM bytes per launch of this kernel: PMC counters cannot be read in-process, so the figure measured with
    # rocprofv3 (--pmc FETCH_SIZE / WRITE_SIZE, separate passes, guide corrections) is read from profiles/
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_conv_fwd_halo_hbm.json")) as fh:
            traffic = float(json.load(fh)["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        pass
    return {"bound": "mfma", "kernel": "conv_fwd_halo_kernel (Conv3d 64->128 k4 s2 p1 forward, 16^3 -> 8^3, 128 samples = critic pass over fake+real; the largest GEMM of the step)",
            "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "launch_ms": round(ms, 4),
            "flop_per_launch": flop}


def sdfnet_numbers():
    """SDFNet half of the metric.  Forward: a 32^3 x 8 grid pass (config 5's generator forward).  Training: one
    auto-decoder step (train_sdf_autodecoder.py:77-91: gather, fused forward, L1+reg loss, fused backward, weight-grad
    GEMMs, two Adam updates) at the reference's 20 000 points/step with latent 128, and at BASELINE configs[2]'s
    200 000 points/step with latent 256.  FLOP/point from SURVEY.md 8d (921 088 fwd at L=128, 1 052 160 at L=256,
    ~3x for a training step)."""
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(0)
    net = SDFNet()
    grid = torch.tensor(get_voxel_coordinates(32)).cuda().repeat((8, 1))
    z = torch.randn(8, 128, device="cuda")
    with torch.no_grad():
        ms_fwd = event_time_ms(lambda: net.forward_shapes(grid, z, 32768), 10)
    n_fwd = 8 * 32768
    fwd_tflops = n_fwd * 921088 / (ms_fwd * 1e-3) / 1e12
    out = {"fwd_mpoints_per_s": round(n_fwd / ms_fwd / 1e3, 2), "fwd_tflops": round(fwd_tflops, 2),
           "fwd_frac_of_f32_mfma_peak": round(fwd_tflops / F32_MFMA_PEAK_TFLOPS, 4)}
    pc, shapes = 200000, 64
    pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
    sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
    for tag, npts, lat, flop_pt in (("train_ref_20k_L128", 20000, 128, 3 * 921088), ("train_cfg_200k_L256", 200000, 256, 3 * 1052160)):
        table = torch.randn(shapes, lat, device="cuda") * 1e-2
        tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), table, pts, sdf, pointcloud_size=pc)
        idx = torch.randint(0, shapes * pc, (npts,), device="cuda")
        ms = event_time_ms(lambda: tr.step(idx), 10)
        out[tag] = {"mpoints_per_s": round(npts / ms / 1e3, 3), "ms_per_step": round(ms, 3),
                    "tflops": round(npts * flop_pt / (ms * 1e-3) / 1e12, 2)}
    return out


def cpu_baseline(reals, zs, zg, g_state, c_state):
    """The CPU oracle (torch fp32 ops = the reference's own arithmetic engine, all host cores) on the same step."""
    from oracle import torch_oracle as O
    # oneDNN's conv3d backward stops scaling past ~32 threads on the GPU box's 256-core host (measured: 0.49 s per
    # critic update at 16-32 threads, 1.6 s at 128, 15 s at 256): use the fastest setting and report it as `cores`
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    orc = O.WGANOracle(g_state, c_state)
    reals = [r.cpu() for r in reals]
    zs = [z.cpu() for z in zs]
    zg = zg.cpu()
    t0 = time.perf_counter()
    orc.critic_step(reals[0], zs[0])            # warm-up (also sizes the sample)
    warm = time.perf_counter() - t0
    nsteps = max(1, min(6, int(15.0 / max(warm * 7, 1e-3))))   # ~15 s of CPU work (a 5+1 step is ~7 critic-step equivalents)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        orc.step(reals, zs, zg)
    dt = time.perf_counter() - t0
    return {"value": round(nsteps / dt, 4), "unit": "steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d full 5+1 WGAN step(s) at batch 64 after 1 warm-up critic update, oracle/torch_oracle.py "
                      "WGANOracle on torch CPU fp32" % nsteps}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / SDFNet side measurements")
    args = ap.parse_args()

    from shapegan_amd import parallel
    rank, world, local = parallel.init_distributed()
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    torch.cuda.set_device(local % torch.cuda.device_count())
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer

    torch.manual_seed(0)                        # identical replicas on every rank
    generator, critic = Generator(), Discriminator()
    g_state = {k: v.detach().cpu().clone() for k, v in generator.state_dict().items()}
    c_state = {k: v.detach().cpu().clone() for k, v in critic.state_dict().items()}
    trainer = WGANTrainer(generator, critic)
    gen = torch.Generator().manual_seed(1000 + rank)   # per-rank synthetic data (batch axis sharded)
    reals = [(torch.rand(BATCH, 32, 32, 32, generator=gen) * 2 - 1).cuda() for _ in range(5)]
    zs = [torch.randn(BATCH, 128, generator=gen).cuda() for _ in range(5)]
    zg = torch.randn(BATCH, 128, generator=gen).cuda()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(reals, zs, zg)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step(reals, zs, zg)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        line = {
            "metric": "GAN train steps/sec @32^3 voxels (train_wgan.py 5 critic + 1 generator updates, batch 64/GPU)",
            "value": round(world * args.steps / elapsed, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "train_wgan.py 32^3 voxel WGAN, fp32, batch=64 synthetic SDF grids (BASELINE configs[1])",
                       "global_batch": BATCH * world, "critic_updates_per_step": 5, "generator_updates_per_step": 1,
                       "parallelism": "dp%d" % world},
            "critic_updates_per_s": round(world * args.steps * 5 / elapsed, 3),
        }
        if not args.no_extras:
            line["roofline"] = roofline_conv()
            line["sdfnet"] = sdfnet_numbers()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(reals, zs, zg, g_state, c_state)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
