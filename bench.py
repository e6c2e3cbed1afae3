#!/usr/bin/env python
"""bench.py — BASELINE.json's headline metric on MI355X: GAN train steps/sec @32^3 voxels (+ SDFNet Mpoints/sec).

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W [--config wgan|hybrid_progressive|hybrid_wgan|sdf]      (self-launching)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W [--config ...]                                             (launcher form)

Both multi-GPU forms run the same thing, one process per GPU: without WORLD_SIZE in the environment `--gpus N > 1` makes this
process the launcher (`self_launch`: N copies of itself with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free
MASTER_PORT; rank 0's single JSON line is passed through on stdout, every rank's stderr is prefixed `[rank r]`, the first rank
that fails stops the others and its exit code is returned) — the reference's one multi-GPU script needs no launcher either
(train_hybrid_progressive_gan.py:62-68).

Default workload (`--config wgan`, BASELINE configs[1], the configuration the metric is quoted on): one "step" is one pass
of the train_wgan.py cadence over synthetic data already resident in HBM — five critic updates (generator forward, one
critic forward+backward over fake+real, fused RMSprop + weight clip) and one generator update (train_wgan.py:39,60-84),
32^3 voxel grids, batch 64 per GPU, fp32.  With N GPUs every rank runs the same per-GPU workload (weak scaling) and each
optimizer update all-reduces one flat gradient buffer over RCCL.  The other configs are the remaining BASELINE workloads
behind the same harness (the multi-GPU ones are `hybrid_progressive` = configs[3] and `hybrid_wgan` = configs[4]):
    hybrid_progressive  train_hybrid_progressive_gan.py iteration 3 (64^3), batch 16/GPU, WGAN-GP: 5 discriminator + 1 generator updates
    hybrid_wgan         train_hybrid_wgan.py 32^3, batch 8/GPU: 5 critic + 1 generator updates
    sdf                 train_sdf_autodecoder.py, latent 256, 200 000 points per step per GPU (value in Mpoints/s)

Rank 0 prints ONE JSON line (schema in the task contract).  At N=1 with the default config it carries
  `roofline`      the kernel that takes the largest share of the step (Conv3d 128->64 input-gradient / ConvTranspose3d forward
                  implicit GEMM, `conv_dgrad_halo_kernel<0>`: 22 % of the step), timed alone with HIP events on the launch
                  stream at the critic's 128-sample shape, plus `kernels`: the same measurement for every conv form of the
                  step (MFMA-bound ones against the fp32 MFMA peak, Cin=1 / Cout=1 edge layers against HBM),
  `sdfnet`        the SDFNet half of the metric (fused forward, both auto-decoder training configs) with algorithmic AND
                  executed FLOP rates,
  `other_configs` BASELINE configs[2] / [3] / [4] on the same GPU (a few timed steps each: value, ms per step, the SDFNet share)
  and `point_gan` (critic / generator update times of train_point_gan.py at 12 x 16 384 points),
  `cpu_baseline`  the same step on the host cores — the reference's own modules where the checkout exists (`kind: "reference"`),
                  else the CPU restatement (`kind: "port"`, `reference_present: false`) — and the GPU-vs-oracle loss agreement;
                  `value` = ALL host cores, filled with concurrent 32-thread replicas of the step (one process stops scaling at
                  ~32 threads), `single_process` = one of them alone,
  `dropin_loop`   what the reference's own loop body (train_wgan.py:60-84 over the module-level surface: generate(), two critic
                  calls, stock torch.optim, clip_weights, .item()) reaches on the same modules, next to `WGANTrainer.step`.
With N > 1 the line carries `comm`: what exchanged the gradients (`native-rccl`: the C-ABI exchange, ranks / RCCL version read
back from the communicator; `torch-nccl`: torch.distributed, with the reason) and the all-reduced bytes per step.
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
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s (about 6.3 TB/s achievable)
BATCH = 64


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


PROFILE_TAGS = ("r06", "r05", "r04", "r03", "r02")     # newest committed profile pass first (scripts/profile_round.sh + collect_profiles.py)
COLD_BYTES = 640 << 20     # > 2 x the 256 MB Infinity Cache: what a rotation of operand sets must cover to be cold


def conv_kernel_table():
    """Every conv form of the WGAN step at the shape it runs with in the critic update (fake+real = 128 samples) or the
    generator (64 samples), timed through the C-ABI entry (weight-image packs included).  flop = 2*Cout*Cin*64 per output
    voxel; bytes = each operand once.

    The HBM-bound rows are timed COLD: a replay loop over one set of buffers keeps a working set of 75 - 151 MB inside the 256 MB
    Infinity Cache and reports a cache rate, not an HBM rate (conv_fwd_c1 44.9 us warm, 54 - 57 us in every rocprof
    measurement).  Each call of such a row therefore reads the next of K distinct operand sets and writes a fresh output block
    (the last K results stay alive, so the allocator cannot hand the same block back), K chosen so that the rotation covers
    > 640 MB; `us_warm` keeps the replay figure next to it, `us_in_step` the average duration of that kernel at that shape inside
    the profiled training step (`profiles/rNN_wgan_step_timeline.txt`) where the committed profile has it."""
    from shapegan_amd import ops
    nb = 2 * BATCH
    rows = []

    def mfma(name, kernel, flop, fn):
        ms = event_time_ms(fn, 20)
        tf = flop / (ms * 1e-3) / 1e12
        rows.append({"name": name, "kernel": kernel, "bound": "mfma", "flop": flop, "us": round(ms * 1e3, 1),
                     "tflops": round(tf, 1), "frac": round(tf / F32_MFMA_PEAK_TFLOPS, 4)})

    def hbm(name, kernel, nbytes, flop, make, call, in_step=None):
        """make() -> one operand set; call(set) -> the result tensor."""
        k = max(4, int(COLD_BYTES // nbytes) + 1)
        sets = [make() for _ in range(k)]
        alive = [None] * k
        state = {"i": 0}

        def cold():
            i = state["i"] = (state["i"] + 1) % k
            alive[i] = call(sets[i])
        ms = event_time_ms(cold, 4 * k)
        ms_warm = event_time_ms(lambda: call(sets[0]), 20)
        gbs = nbytes / (ms * 1e-3) / 1e9
        row = {"name": name, "kernel": kernel, "bound": "hbm", "bytes": nbytes, "flop": flop, "us": round(ms * 1e3, 1),
               "gb_per_s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "timing": "cold: %d operand sets in rotation" % k,
               "us_warm": round(ms_warm * 1e3, 1)}
        us_step = step_kernel_us(kernel, in_step)
        if us_step is not None:
            row["us_in_step"], row["frac_in_step"] = us_step, round(nbytes / (us_step * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
        rows.append(row)
        del sets, alive

    x16 = torch.randn(nb, 64, 16, 16, 16, device="cuda")
    w2 = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
    b2 = torch.zeros(128, device="cuda")
    y8 = torch.randn(nb, 128, 8, 8, 8, device="cuda")
    f2 = 2.0 * 128 * 64 * 64 * 512 * nb
    mfma("Conv3d 64->128 input-gradient / ConvT forward, 8^3 -> 16^3, 128 samples", "conv_dgrad_halo_kernel<0>", f2,
         lambda: ops.conv_dgrad_raw(y8, w2, None, 64))
    mfma("Conv3d 64->128 forward, 16^3 -> 8^3, 128 samples", "conv_fwd_halo_kernel<1,8>", f2,
         lambda: ops.conv_fwd_raw(x16, w2, b2, 1, 0.2))
    mfma("Conv3d 64->128 weight-gradient, 128 samples", "conv_wgrad_halo_kernel", f2, lambda: ops.conv_wgrad_raw(y8, x16, 64))
    x8 = torch.randn(nb, 128, 8, 8, 8, device="cuda")
    w3 = torch.randn(256, 128, 4, 4, 4, device="cuda") * 0.02
    b3 = torch.zeros(256, device="cuda")
    y4 = torch.randn(nb, 256, 4, 4, 4, device="cuda")
    f3 = 2.0 * 256 * 128 * 64 * 64 * nb
    mfma("Conv3d 128->256 forward, 8^3 -> 4^3, 128 samples", "conv_fwd_halo4_kernel", f3, lambda: ops.conv_fwd_raw(x8, w3, b3, 1, 0.2))
    mfma("Conv3d 128->256 input-gradient / ConvT forward, 4^3 -> 8^3, 128 samples", "conv_dgrad_halo_kernel<1>", f3,
         lambda: ops.conv_dgrad_raw(y4, w3, None, 128))
    mfma("Conv3d 128->256 weight-gradient, 128 samples", "conv_wgrad_halo4_kernel", f3, lambda: ops.conv_wgrad_raw(y4, x8, 128))
    del x16, y8, x8, y4
    # HBM-bound edge layers (one channel on one side)
    w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
    b1 = torch.zeros(64, device="cuda")
    nx, ny = nb * 32768, nb * 64 * 4096
    f1 = 2.0 * 64 * 64 * 4096 * nb
    mk_x = lambda: torch.randn(nb, 1, 32, 32, 32, device="cuda")
    mk_y = lambda n=nb: torch.randn(n, 64, 16, 16, 16, device="cuda")
    hbm("Conv3d 1->64 forward, 32^3 -> 16^3, 128 samples", "conv_fwd_c1_lds_kernel<2,1,32>", 4.0 * (nx + ny + w1.numel()), f1,
        mk_x, lambda x: ops.conv_fwd_raw(x, w1, b1, 1, 0.2), in_step="largest")
    hbm("Conv3d 1->64 weight-gradient, 128 samples", "conv_wgrad_c1_kernel<2,0>", 4.0 * (nx + ny + w1.numel()), f1,
        lambda: (mk_y(), mk_x()), lambda s: ops.conv_wgrad_raw(s[0], s[1], 1))
    hbm("Conv3d 1->64 weight + bias gradient through LeakyReLU (the critic's first layer: reads dy and y), 128 samples",
        "conv_wgrad_c1_kernel<2,LEAKY>", 4.0 * (nx + 2 * ny + w1.numel()), f1,
        lambda: (mk_y(), mk_y(), mk_x()), lambda s: ops.conv_wgrad_act_raw(s[0], s[1], s[2], 1, 0.2), in_step="largest")
    hbm("ConvT 64->1 forward / Conv3d 1->64 input-gradient, 16^3 -> 32^3, 64 samples", "convT_c1_stream_kernel<true,false,true,0>",
        4.0 * (ny // 2 + BATCH * 32768 + w1.numel()), f1 / 2, lambda: mk_y(BATCH), lambda y: ops.conv_dgrad_raw(y, w1, None, 1), in_step="smallest")
    # the launch the 5+1 step actually runs for its four grouped inference generator passes: 256 samples, the last BatchNorm +
    # LeakyReLU folded into the loads, tanh (the largest one-channel launch of the step was not in the table)
    g4 = 4 * BATCH
    wt = torch.randn(64, 1, 4, 4, 4, device="cuda") / 23.0
    bt = torch.randn(1, device="cuda")
    sc, sh = torch.rand(64, device="cuda") + 0.5, torch.randn(64, device="cuda") * 0.1
    hbm("ConvT 64->1 forward with the input transform (BatchNorm + LeakyReLU in the loads) + tanh, 16^3 -> 32^3, 256 samples (the "
        "grouped generator pass of WGANTrainer.step)", "convT_c1_all_kernel<true,true,true,3>",
        4.0 * (g4 * 64 * 4096 + g4 * 32768 + wt.numel()), 2.0 * 64 * 64 * 4096 * g4, lambda: mk_y(g4),
        lambda y: ops.conv_transpose3d_to1_pre_raw(y, sc, sh, 1, 0.2, wt, bt, 3, 0.0), in_step="largest")
    return rows


def step_kernel_us(kernel, which):
    """Average duration (us) of `kernel`'s launches at the table's shape inside the profiled 5+1 training step of the newest
    committed `profiles/rNN_wgan_step_timeline.txt` (one line per launch: start, gap, duration, name).  The step runs the
    one-channel kernels at two batch sizes (128 samples in the critic updates, 64 in the generator update): `which` = "largest" /
    "smallest" picks the launches within 35 % of the longest / shortest one.  None: the profile has no such kernel (or the step
    does not run it at the table's shape — the plain one-channel weight gradient only runs at 64 samples there)."""
    import re
    if which is None:
        return None
    want = kernel.replace(" ", "").replace("LEAKY", "1")
    for tag in PROFILE_TAGS:
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_wgan_step_timeline.txt")) as fh:
                lines = fh.read().splitlines()
        except OSError:
            continue
        durs = []
        for ln in lines:
            m = re.match(r"\s*[-\d.]+\s+\+\s+[-\d.]+ gap\s+([\d.]+) us\s+(.*)$", ln)
            if m and m.group(2).replace(" ", "") == want:
                durs.append(float(m.group(1)))
        if not durs:
            continue
        ref = max(durs) if which == "largest" else min(durs)
        pick = [d for d in durs if abs(d - ref) <= 0.35 * ref]
        return round(sum(pick) / len(pick), 1)
    return None


def dominant_kernel(rows):
    """The row of `rows` whose kernel has the largest share of the WGAN step's kernel time according to the newest committed
    `profiles/rNN_wgan_step_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of the step: "Name", ..., "Percentage"), and that
    share.  The table's kernel names are matched against the profiler's (e.g. "conv_dgrad_halo_kernel<0>" inside
    "void sg::conv_dgrad_halo_kernel<0>(sg::HaloDgradArgs)"); without a profile file, the first row."""
    import csv
    for tag in PROFILE_TAGS:
        try:
            with open(os.path.join(ROOT, "profiles", tag + "_wgan_step_kernel_stats.csv")) as fh:
                stats = sorted(((float(r["Percentage"]), r["Name"]) for r in csv.DictReader(fh)), reverse=True)
        except (OSError, KeyError, ValueError):
            continue
        for share, name in stats:
            squeezed = name.replace(" ", "")
            for row in rows:
                if "sg::" + row["kernel"].replace(" ", "") + "(" in squeezed:
                    return row, share, tag
            break      # the top kernel is not in the table: say so rather than report a smaller one as dominant
        return rows[0], None, tag
    return rows[0], None, None


def roofline_and_kernels():
    rows = conv_kernel_table()
    dom, share, tag = dominant_kernel(rows)
    traffic = None   # HBM bytes per launch from the rocprofv3 PMC passes (cannot be read in-process)
    for t in PROFILE_TAGS:
        try:
            with open(os.path.join(ROOT, "profiles", t + "_dominant_kernel_hbm.json")) as fh:
                rec = json.load(fh)
            if dom["kernel"].replace(" ", "") in rec.get("kernel", "").replace(" ", ""):
                traffic = float(rec["hbm_bytes_per_launch"])
            break
        except (OSError, KeyError, ValueError):
            pass
    why = "the largest share of the step's kernel time" + ("" if share is None else ": %.1f %% in profiles/%s_wgan_step_kernel_stats.csv" % (share, tag))
    roof = {"bound": dom["bound"], "kernel": dom["kernel"] + " (" + dom["name"] + "; " + why + ")",
            "achieved": dom["tflops"] if dom["bound"] == "mfma" else dom["gb_per_s"],
            "peak": F32_MFMA_PEAK_TFLOPS if dom["bound"] == "mfma" else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if dom["bound"] == "mfma" else "GB/s", "frac": dom["frac"], "traffic": traffic,
            "launch_ms": round(dom["us"] / 1e3, 4), "flop_per_launch": dom["flop"]}
    return roof, rows


def sdf_flops(latent):
    """(algorithmic, executed) forward FLOP per point: SURVEY.md 8d's count of the reference's cat+Linear layers, and what the
    kernels execute once the latent columns of layers1.0 / layers2.0 are folded into per-shape bias vectors."""
    algorithmic = 2 * ((3 + latent) * 256 + 3 * 256 * 256 + (259 + latent) * 256 + 2 * 256 * 256 + 256)
    executed = algorithmic - 2 * 2 * latent * 256
    return algorithmic, executed


def sdfnet_numbers():
    """SDFNet half of the metric.  Forward: a 32^3 x 8 grid pass (config 5's generator forward).  Training: one
    auto-decoder step (train_sdf_autodecoder.py:77-91: gather, fused forward, L1+reg loss, fused backward, weight-grad
    GEMMs, two Adam updates) at the reference's 20 000 points/step with latent 128, and at BASELINE configs[2]'s
    200 000 points/step with latent 256.  Rates are given twice: against the reference's algorithmic FLOP count (SURVEY.md
    8d: forward, x3 for a training step) and against the FLOPs the kernels execute (both steps take the shape-sorted data
    flow, whose per-shape latent fold removes the latent columns of two layers)."""
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
    alg, exe = sdf_flops(128)
    out = {"fwd_mpoints_per_s": round(n_fwd / ms_fwd / 1e3, 2),
           "fwd_tflops_algorithmic": round(n_fwd * alg / (ms_fwd * 1e-3) / 1e12, 2),
           "fwd_tflops_executed": round(n_fwd * exe / (ms_fwd * 1e-3) / 1e12, 2),
           # (the algorithmic count includes the latent columns of layers1.0 / layers2.0 that the per-shape fold never multiplies:
           # against the reference's FLOP count the kernel is this factor "faster" than its executed rate — a speed-up of the data
           # flow, not a fraction of the matrix peak)
           "speedup_from_latent_fold": round(alg / exe, 4),
           "fwd_frac_of_f32_mfma_peak_executed": round(n_fwd * exe / (ms_fwd * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    pc, shapes = 200000, 64
    pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
    sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
    # The reference's 20 000-point step is 23 launches of which 20 take a few microseconds: launched one by one it is paced by the
    # host (Python + ctypes + autograd per launch), so its figure is that of SDFAutoDecoderTrainer.step_graphed — the same step
    # as one captured graph launch; the eager figure is reported next to it.
    for tag, npts, lat, folded, graphed in (("train_ref_20k_L128", 20000, 128, True, True),
                                            ("train_ref_20k_L128_eager", 20000, 128, True, False),
                                            ("train_cfg_200k_L256", 200000, 256, True, False)):
        table = torch.randn(shapes, lat, device="cuda") * 1e-2
        tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), table, pts, sdf, pointcloud_size=pc, capturable=graphed)
        idx = torch.randint(0, shapes * pc, (npts,), device="cuda")
        # graphed: the whole step (sort, fused forward / backward, weight gradients, two Adam updates) as one captured launch
        ms = event_time_ms((lambda: tr.step_graphed(idx)) if graphed else (lambda: tr.step(idx)), 10)
        alg, exe = sdf_flops(lat)
        exe = exe if folded else alg
        out[tag] = {"launch": "step_graphed (one captured graph per step)" if graphed else "step (eager launches)",
                    "mpoints_per_s": round(npts / ms / 1e3, 3), "ms_per_step": round(ms, 3),
                    "tflops_algorithmic": round(npts * 3 * alg / (ms * 1e-3) / 1e12, 2),
                    "tflops_executed": round(npts * 3 * exe / (ms * 1e-3) / 1e12, 2),
                    "frac_of_f32_mfma_peak_executed": round(npts * 3 * exe / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    return out


def _count_launches(fn):
    """GPU kernel launches of one call of fn() as the torch profiler's device-side activity sees them (None if it cannot trace)."""
    try:
        from torch.profiler import ProfilerActivity, profile
        fn()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            fn()
            torch.cuda.synchronize()
        n = sum(1 for e in prof.events() if str(getattr(e, "device_type", "")).endswith("CUDA") and "memcpy" not in e.name.lower()
                and "memset" not in e.name.lower())
        return n or None
    except Exception:       # noqa: BLE001 — a diagnostic column, never a reason to lose the line
        return None


def dropin_loop_numbers(steps=10, warmup=3):
    """What the reference's OWN loop achieves on the drop-in surface: the body of train_wgan.py:60-84 over the
    module-level API only — `generator.generate()` (latents drawn on the CPU and moved, model/gan.py:31-34), two `critic(...)`
    calls per update, `loss.backward()` through the autograd engine, stock `torch.optim.RMSprop`, `critic.clip_weights()`, the
    three `.item()` reads of the generator update — on the same modules the headline trains, at batch 64, five critic updates and
    one generator update per step.  `WGANTrainer.step` (the headline) is the same arithmetic with the step-level fusions a
    script cannot express through that surface; the ratio is reported so that nobody mistakes one for the other."""
    from shapegan_amd.model.gan import Discriminator, Generator
    torch.manual_seed(0)
    generator, critic = Generator(), Discriminator()
    critic.use_sigmoid = False
    g_opt = torch.optim.RMSprop(generator.parameters(), lr=0.00005)
    c_opt = torch.optim.RMSprop(critic.parameters(), lr=0.00005)
    gen = torch.Generator().manual_seed(1000)
    host = [(torch.rand(BATCH, 32, 32, 32, generator=gen) * 2 - 1) for _ in range(5)]      # what a DataLoader hands the loop
    resident = [b.cuda() for b in host]
    logged = []

    def unit(batches, to_device):
        for batch_index, batch in enumerate(batches):
            generator.zero_grad()
            critic.zero_grad()
            fake_sample = generator.generate(sample_size=batch.shape[0]).detach()
            fake_out = critic(fake_sample)
            valid_out = critic(batch.cuda() if to_device else batch)
            critic_loss = torch.mean(fake_out) - torch.mean(valid_out)
            critic_loss.backward()
            c_opt.step()
            critic.clip_weights(0.01)
            if batch_index % 5 == 0:
                generator.zero_grad()
                critic.zero_grad()
                fake_out = critic(generator.generate(sample_size=BATCH))
                generator_loss = -torch.mean(fake_out)
                generator_loss.backward()
                g_opt.step()
                del logged[:]
                logged.extend((torch.mean(fake_out).item(), torch.mean(valid_out).item()))

    out = {"what": "train_wgan.py:60-84 restated over the module-level surface only (generate(), two critic calls, autograd, stock "
                   "torch.optim.RMSprop, clip_weights, .item() logging), batch 64, 5 critic + 1 generator updates per step"}
    for key, batches, to_device in (("resident_batches", resident, False), ("host_batches_copied_in_the_loop", host, True)):
        for _ in range(warmup):
            unit(batches, to_device)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            unit(batches, to_device)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[key] = {"steps_per_s": round(1.0 / dt, 3), "ms_per_step": round(dt * 1e3, 3)}
    out["launches_per_step"] = _count_launches(lambda: unit(resident, False))
    return out


class _ReferenceWGAN(object):
    """train_wgan.py:37-46,60-84 on the REFERENCE's OWN module classes (model.gan.Generator / Discriminator imported from the
    reference checkout by oracle/ref_import.py), stock torch.optim.RMSprop, on the CPU — BASELINE.md section 2's baseline.  Only
    possible where the checkout exists (the authoring container); the driver's GPU box has none."""

    def __init__(self, ref, g_state, c_state, lr=0.00005, clip=0.01):
        self.g, self.c = ref.Generator().cpu(), ref.Discriminator().cpu()
        self.g.load_state_dict(g_state)
        self.c.load_state_dict(c_state)
        self.c.use_sigmoid = False
        self.g_opt = torch.optim.RMSprop(self.g.parameters(), lr=lr)
        self.c_opt = torch.optim.RMSprop(self.c.parameters(), lr=lr)
        self.clip = clip

    def critic_step(self, real, z):
        self.g.zero_grad()
        self.c.zero_grad()
        fake = self.g(z).detach()
        out_fake, out_real = self.c(fake), self.c(real)
        loss = torch.mean(out_fake) - torch.mean(out_real)
        loss.backward()
        self.c_opt.step()
        self.c.clip_weights(self.clip)
        return loss.detach(), out_fake.detach(), out_real.detach()

    def generator_step(self, z):
        self.g.zero_grad()
        self.c.zero_grad()
        loss = -torch.mean(self.c(self.g(z)))
        loss.backward()
        self.g_opt.step()

    def step(self, reals, zs, zg):
        for i, (real, z) in enumerate(zip(reals, zs)):
            self.critic_step(real, z)
            if i == 0:
                self.generator_step(zg)


def usable_cores():
    """Cores this process can really run on: the affinity mask, capped by the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us` /
    `cpu.cfs_period_us`) — os.cpu_count() reports the host's logical CPUs whatever the container is allowed to use."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            q, period = fh.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, period = float(fq.read()), float(fp.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = max(1, min(n, int(quota + 0.5)))
    return n


def _cpu_replica_main(argv):
    """`python bench.py --cpu-replica <state file> <steps> <threads> <index>`: one replica of the CPU baseline (no GPU, no
    shapegan_amd): loads state and inputs, pins itself to its core set, runs `steps` 5+1 steps, prints its elapsed seconds."""
    path, steps, threads, index = argv[0], int(argv[1]), int(argv[2]), int(argv[3])
    try:
        os.sched_setaffinity(0, range(index * threads, (index + 1) * threads))
    except (AttributeError, OSError):
        pass
    torch.set_num_threads(threads)
    from oracle import ref_import
    from oracle import torch_oracle as O
    d = torch.load(path)
    orc = _ReferenceWGAN(ref_import.load(), d["g"], d["c"]) if ref_import.available() else O.WGANOracle(d["g"], d["c"])
    orc.critic_step(d["reals"][0], d["zs"][0])          # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        orc.step(d["reals"], d["zs"], d["zg"])
    print(json.dumps({"elapsed": time.perf_counter() - t0, "steps": steps}), flush=True)


def _cpu_replicas(replicas, threads, steps, reals, zs, zg, g_state, c_state, budget_s=60.0):
    """`replicas` concurrent copies of the CPU step, `threads` threads each, pinned to disjoint core sets.  Gives up (all replicas
    stopped by PID) when they are not done after `budget_s`: the default bench run must stay within minutes."""
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "state.pt")
        torch.save({"g": g_state, "c": c_state, "reals": reals, "zs": zs, "zg": zg}, path)
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-replica", path, str(steps), str(threads), str(i)],
                                  env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(replicas)]
        t0 = time.perf_counter()
        while any(p.poll() is None for p in procs) and time.perf_counter() - t0 < budget_s:
            time.sleep(0.2)
        late = [p for p in procs if p.poll() is None]
        for p in late:
            p.kill()
        times = []
        for p in procs:
            out, _ = p.communicate()
            lines = [ln for ln in (out or "").splitlines() if ln.startswith("{")]
            if p.returncode == 0 and lines:
                times.append(json.loads(lines[-1])["elapsed"])
    if late or len(times) != replicas:
        return {"replicas": replicas, "threads_each": threads, "steps_each": steps, "steps_per_s": 0.0,
                "note": "%d of %d replicas not done after %.0f s (stopped)" % (len(late), replicas, budget_s) if late else "a replica failed"}
    return {"replicas": replicas, "threads_each": threads, "steps_each": steps, "slowest_s": round(max(times), 3),
            "fastest_s": round(min(times), 3), "steps_per_s": replicas * steps / max(times)}


def cpu_baseline(reals, zs, zg, g_state, c_state):
    """The same 5+1 step on the host cores, plus the agreement of the GPU path with it on the first critic update.  Where the
    reference checkout is present the step runs on the reference's own modules (`kind: "reference"`); on a box without it — the
    driver's GPU box: /root/reference does not travel — on oracle/torch_oracle.py, the CPU restatement of the same step on the
    same torch fp32 ops (`kind: "port"`, `reference_present: false`)."""
    from oracle import ref_import
    from oracle import torch_oracle as O
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    # one process: min(32, usable cores) threads (the GPU box: 256 logical CPUs behind a cgroup quota of 16 cores, see below)
    torch.set_num_threads(max(1, min(32, usable_cores())))
    have_ref = ref_import.available()
    if have_ref:
        orc, kind, who = _ReferenceWGAN(ref_import.load(), g_state, c_state), "reference", "the reference's model.gan modules (imported from the checkout)"
    else:
        orc, kind, who = O.WGANOracle(g_state, c_state), "port", "oracle/torch_oracle.py WGANOracle"
    reals = [r.cpu() for r in reals]
    zs = [z.cpu() for z in zs]
    zg = zg.cpu()
    t0 = time.perf_counter()
    ref = orc.critic_step(reals[0], zs[0])      # warm-up (also sizes the sample) and the parity reference
    warm = time.perf_counter() - t0
    # the same update on the GPU from the same initial state: loss and the 2 x 64 critic scores
    g, c = Generator(), Discriminator()
    g.load_state_dict(g_state)
    c.load_state_dict(c_state)
    got = WGANTrainer(g, c).critic_step(reals[0].cuda(), zs[0].cuda())
    scores_ref = torch.cat([ref[1], ref[2]])
    scores = torch.cat([got[1], got[2]]).cpu()
    rel = float((scores - scores_ref).abs().max() / scores_ref.abs().mean())
    loss_rel = abs(float(got[0]) - float(ref[0])) / max(abs(float(ref[0])), 1e-30)
    nsteps = max(1, min(6, int(15.0 / max(warm * 7, 1e-3))))   # ~15 s of CPU work (a 5+1 step is ~7 critic-step equivalents)
    t0 = time.perf_counter()
    for _ in range(nsteps):
        orc.step(reals, zs, zg)
    dt = time.perf_counter() - t0
    threads = torch.get_num_threads()
    single = {"value": round(nsteps / dt, 4), "cores": threads}
    # ALL usable host cores.  "Usable" is what the scheduler grants, not os.cpu_count(): the GPU box shows
    # 256 logical CPUs behind a cgroup quota of 16 cores (cpu.max = 1600000 100000) — which is why rounds 1 - 4 found the step
    # "stops scaling" between 16 and 32 threads and collapses at 256 (15 s per critic update), and why eight concurrent 32-thread
    # replicas took 86 - 100 s per step instead of 3 (0.08 steps/s aggregate against 0.35 for one process).  The baseline therefore
    # runs on min(32, usable) threads; only a host that really grants more than twice that is filled with independent pinned
    # replicas of the same step (the data-parallel way to use a many-core host), whose aggregate is then the value.
    usable = usable_cores()
    replicas = max(1, usable // threads)
    agg = _cpu_replicas(replicas, threads, max(1, nsteps // 2), reals, zs, zg, g_state, c_state) if replicas > 1 else None
    if agg and agg["steps_per_s"] <= single["value"]:
        agg["note"] = "slower than one process: not used as the value"
    use_agg = bool(agg) and agg["steps_per_s"] > single["value"]
    value, cores = (agg["steps_per_s"], replicas * threads) if use_agg else (single["value"], threads)
    return {"value": round(value, 4), "unit": "steps/s", "cores": cores, "kind": kind,
            "reference_present": have_ref, "host_cores": os.cpu_count(), "usable_cores": usable, "single_process": single,
            "all_cores": agg,
            "sample": "%d full 5+1 WGAN step(s) at batch 64 after 1 warm-up critic update, %s on torch CPU fp32%s"
                      % (nsteps, who, "" if not use_agg else "; value = %d concurrent replicas x %d threads, %d step(s) each, steps "
                         "summed over the replicas / the slowest replica's time" % (replicas, threads, agg["steps_each"])),
            "why_port": None if have_ref else "the reference checkout (/root/reference) does not exist on this box, so its module "
                        "classes cannot be imported; the port runs the same torch.nn.functional calls from the same state_dict",
            "gpu_vs_oracle": {"critic_scores_max_err_over_mean_abs": rel, "critic_loss_rel_err": loss_rel,
                              "what": "first critic update at batch 64 from the same initial state and inputs"}}


# ---- workloads -----------------------------------------------------------------------------------------------------------
def _local_batch(global_batch, world, strong, what):
    """Per-GPU share of one step.  weak: every GPU runs the reference's own batch (SURVEY 8e's headline); strong: the reference's
    batch is split over the GPUs (the batch axis shards, every loss is a batch mean: averaged shard gradients = the full-batch
    gradient up to summation order)."""
    if not strong or world == 1:
        return global_batch
    if global_batch % world:
        sys.exit("bench.py --scaling strong: %s batch %d does not split over %d GPUs" % (what, global_batch, world))
    return global_batch // world


def make_wgan(rank, world=1, strong=False):
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    B = _local_batch(BATCH, world, strong, "train_wgan.py")
    torch.manual_seed(0)                        # identical replicas on every rank
    generator, critic = Generator(), Discriminator()
    state = ({k: v.detach().cpu().clone() for k, v in generator.state_dict().items()},
             {k: v.detach().cpu().clone() for k, v in critic.state_dict().items()})
    trainer = WGANTrainer(generator, critic)
    gen = torch.Generator().manual_seed(1000 + rank)   # per-rank synthetic data (batch axis sharded)
    # the real batches are resident where the step reads them: the real halves of the trainer's critic batches (the destination an
    # input pipeline copies its host batches to; the reference's loop reads each batch where `.to(device)` put it, no device copy)
    reals = trainer.real_slots(B, resolution=32, updates=5, device="cuda")
    for slot in reals:
        slot.copy_((torch.rand(B, 1, 32, 32, 32, generator=gen) * 2 - 1))
    # (the unit's five critic latent batches are one draw: WGANTrainer.step's grouped generator pass then reads them in place)
    zs = list(torch.randn(5, B, 128, generator=gen).cuda().unbind(0))
    zg = torch.randn(B, 128, generator=gen).cuda()
    info = {"optimizers": [trainer.c_opt, trainer.g_opt],
            "metric": "GAN train steps/sec @32^3 voxels (train_wgan.py 5 critic + 1 generator updates, batch 64/GPU)",
            "unit": "steps/s", "units_per_step": 1.0 / (world if strong else 1),
            "workload": "train_wgan.py 32^3 voxel WGAN, fp32, batch=64 synthetic SDF grids (BASELINE configs[1])",
            "batch": B, "extra": {"critic_updates_per_step": 5, "generator_updates_per_step": 1},
            "allreduce_bytes_per_step": 4 * (5 * trainer.c_opt.f.total + trainer.g_opt.f.total)}
    return (lambda: trainer.step(reals, zs, zg)), info, (reals, zs, zg, state)


def make_hybrid_progressive(rank, world=1, strong=False):
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridProgressiveGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(0)
    R, B = 64, _local_batch(16, world, strong, "train_hybrid_progressive_gan.py")
    g, d = SDFNet(), Discriminator().cuda()
    d.set_iteration(3)
    tr = HybridProgressiveGANTrainer(g, d, torch.tensor(get_voxel_coordinates(R)).cuda(), R)
    gen = torch.Generator().manual_seed(1000 + rank)
    reals = [(torch.rand(B, R, R, R, generator=gen) * 2 - 1).cuda() for _ in range(5)]
    zs = [torch.randn(B, 128, generator=gen).cuda() for _ in range(6)]
    alphas = [torch.rand(B, 1, 1, 1, generator=gen).cuda() for _ in range(5)]

    def step():   # train_hybrid_progressive_gan.py:120-166: generator on every 5th batch (first), discriminator on every batch
        tr.generator_step(zs[5])
        for real, z, alpha in zip(reals, zs, alphas):
            tr.discriminator_step(real, z, alpha)
    info = {"optimizers": [tr.d_opt, tr.g_opt],
            "metric": "hybrid progressive WGAN-GP train steps/sec @64^3 (1 generator + 5 discriminator updates, batch 16/GPU)",
            "unit": "steps/s", "units_per_step": 1.0 / (world if strong else 1),
            "workload": "train_hybrid_progressive_gan.py iteration=3 (64^3), SDFNet generator + progressive 3D-CNN discriminator, "
                        "WGAN-GP double backward, fp32, batch=16 (BASELINE configs[3])",
            "batch": B, "extra": {"discriminator_updates_per_step": 5, "generator_updates_per_step": 1},
            "allreduce_bytes_per_step": 4 * (5 * tr.d_opt.f.total + tr.g_opt.f.total),
            "sdfnet_points": {"forward": 6 * B * R ** 3, "backward": B * R ** 3, "latent": 128}}
    return step, info, None


def make_hybrid_wgan(rank, world=1, strong=False):
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridWGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(0)
    B = _local_batch(8, world, strong, "train_hybrid_wgan.py")
    tr = HybridWGANTrainer(SDFNet(), Discriminator(), torch.tensor(get_voxel_coordinates(32)).cuda())
    gen = torch.Generator().manual_seed(1000 + rank)
    reals = [(torch.rand(B, 32, 32, 32, generator=gen) * 0.2 - 0.1).cuda() for _ in range(5)]
    zs = [torch.randn(B, 128, generator=gen).cuda() for _ in range(6)]

    def step():   # train_hybrid_wgan.py:78-115: critic on every batch, generator on every 5th
        for i, (real, z) in enumerate(zip(reals, zs)):
            tr.critic_step(real, z)
            if i == 0:
                tr.generator_step(zs[5])
    info = {"optimizers": [tr.c_opt, tr.g_opt],
            "metric": "hybrid WGAN train steps/sec @32^3 (5 critic + 1 generator updates, batch 8/GPU)", "unit": "steps/s",
            "units_per_step": 1.0 / (world if strong else 1),
            "workload": "train_hybrid_wgan.py, SDFNet generator sampled to 32^3 + 3D-CNN critic, fp32, batch=8 (BASELINE configs[4])",
            "batch": B, "extra": {"critic_updates_per_step": 5, "generator_updates_per_step": 1},
            "allreduce_bytes_per_step": 4 * (5 * tr.c_opt.f.total + tr.g_opt.f.total),
            "sdfnet_points": {"forward": 6 * B * 32 ** 3, "backward": B * 32 ** 3, "latent": 128}}
    return step, info, None


def make_sdf(rank, world=1, strong=False):
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    torch.manual_seed(0)
    pc, shapes, lat = 200000, 64, 256
    npts = _local_batch(200000, world, strong, "train_sdf_autodecoder.py")
    gen = torch.Generator().manual_seed(1000 + rank)
    pts = (torch.rand(shapes * pc, 3, generator=gen) * 2 - 1).cuda()
    sdf = (torch.rand(shapes * pc, generator=gen) * 0.2 - 0.1).cuda()
    table = (torch.randn(shapes, lat, generator=torch.Generator().manual_seed(7)) * 1e-2).cuda()   # replicated table
    tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), table, pts, sdf, pointcloud_size=pc)
    idx = torch.randint(0, shapes * pc, (npts,), generator=gen).cuda()
    info = {"optimizers": [tr.net_opt, tr.lat_opt],
            "metric": "SDFNet auto-decoder training Mpoints/sec (train_sdf_autodecoder.py, latent 256, 200 000 points/step/GPU)",
            "unit": "Mpoints/s", "units_per_step": npts / 1e6,
            "workload": "train_sdf_autodecoder.py DeepSDF, latent=256, 200k (xyz,sdf) points/step, fp32 (BASELINE configs[2])",
            "batch": npts, "extra": {},
            "allreduce_bytes_per_step": 4 * (tr.net_opt.f.total + tr.lat_opt.f.total),
            "sdfnet_points": {"forward": npts, "backward": npts, "latent": lat}}
    return (lambda: tr.step(idx)), info, None


def other_configs(steps=4, warmup=3):
    """The remaining BASELINE workloads (configs[2], [3], [4]) on this GPU, a few timed steps each, so that they appear in the
    driver-run record next to the headline: `value` in the config's own unit, and the SDFNet share of the step (its generator
    evaluations / its fused training kernels at the FLOPs they EXECUTE, against the fp32 MFMA peak)."""
    out = {}
    for name in ("sdf", "hybrid_progressive", "hybrid_wgan"):
        torch.cuda.empty_cache()
        step, info, _ = WORKLOADS[name](0)
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3      # wall clock around synchronised steps, as the headline is timed
        pts = info["sdfnet_points"]
        _, exe = sdf_flops(pts["latent"])
        flop = (pts["forward"] + 2 * pts["backward"]) * exe       # forward everywhere, + input & weight gradients where it trains
        out[name] = {"workload": info["workload"], "metric": info["metric"], "unit": info["unit"], "steps": steps,
                     "ms_per_step": round(ms, 3), "value": round(info["units_per_step"] / (ms * 1e-3), 4),
                     "sdfnet_executed_tflop_per_step": round(flop / 1e12, 3),
                     "sdfnet_executed_flop_over_step_time_frac_of_f32_mfma_peak": round(flop / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
        del step
    torch.cuda.empty_cache()
    try:      # (a side measurement must never cost the headline its line)
        out["point_gan"] = point_gan_updates()
    except Exception as e:      # noqa: BLE001
        out["point_gan"] = {"error": "%s: %s" % (type(e).__name__, e)}
    torch.cuda.empty_cache()
    return out


def point_gan_updates(P=16384, B=12, steps=5, warmup=3):
    """SURVEY.md 8f rank 4 (train_point_gan.py:52-83 at its (num_points, batch) = (16 384, 12) stage): one critic update with the
    gradient penalty and one generator update of the PointNet GAN, in time and points per second.  FLOP rates in two arithmetics:
    the reference's dense autograd (4.58 / 3.40 MFLOP per point) and the one executed here (the max over a cloud has a sparse adjoint:
    everything behind the plain passes runs on 512 points per cloud, shapegan_amd/model/point_sdf_net.py)."""
    from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
    from shapegan_amd.train_steps import PointGANTrainer
    torch.manual_seed(0)
    tr = PointGANTrainer(SDFGenerator(128, 256, 8, True).cuda(), PointNet(1).cuda())
    u = torch.cat([torch.rand(B, P, 3) * 2 - 1, torch.rand(B, P, 1) * 0.2 - 0.1], -1).cuda()
    z, a = torch.randn(B, 128, device="cuda"), torch.rand(B, 1, 1, device="cuda")
    out = {"workload": "train_point_gan.py WGAN-GP, SDFGenerator(128, 256, 8) vs PointNet critic, %d clouds x %d points, fp32" % (B, P)}
    g, d, sp = 0.790e6, 0.345e6, 512.0 / P
    # (each update as one captured graph launch, like the 20 000-point auto-decoder step: ~110 launches of 5 - 60 us behind a dense
    #  pass of 3 ms are paced by the host when launched eagerly on a slow one)
    for name, fn, ref, exe in (("critic_update", lambda: tr.critic_step_graphed(u, z, a), g + 11 * d, g + 3 * d + sp * 11 * d),
                               ("generator_update", lambda: tr.generator_step_graphed(u, z), 3 * (g + d), g + d + sp * 3 * (g + d))):
        for _ in range(warmup + 1):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        out[name] = {"launch": "one captured graph per update", "ms": round(ms, 3), "mpoints_per_s": round(B * P / ms / 1e3, 2),
                     "reference_arithmetic_frac_of_f32_mfma_peak": round(ref * B * P / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                     "executed_arithmetic_frac_of_f32_mfma_peak": round(exe * B * P / (ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4)}
    return out


def replica_digests(optimizers, world):
    """[world][len(optimizers)] int64 digests of every rank's flat parameter buffers (the fp32 words summed as integers: equal
    digests <=> bit-identical replicas, up to a collision).  Data-parallel replicas start identical by seed and apply the same
    all-reduced gradient, so after any number of steps they must still be bit-identical; the line reports whether they are."""
    mine = torch.stack([o.f.flat.view(torch.int32).to(torch.int64).sum() for o in optimizers])
    everyone = [torch.zeros_like(mine) for _ in range(world)]
    torch.distributed.all_gather(everyone, mine)
    return [e.cpu().tolist() for e in everyone]


WORKLOADS = {"wgan": make_wgan, "hybrid_progressive": make_hybrid_progressive, "hybrid_wgan": make_hybrid_wgan, "sdf": make_sdf}


def self_launch(gpus, argv, script=None):
    """`python bench.py --gpus N` without a launcher: starts N copies of this script, one per GPU, with the environment
    torch.distributed.run would have given them (rendezvous on 127.0.0.1 and a free port), passes rank 0's stdout (the ONE
    JSON line) through, prefixes every rank's stderr lines with its rank, and returns the first non-zero exit code after stopping
    the remaining ranks (by the PIDs started here, never by pattern)."""
    import socket
    import subprocess
    import threading
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs, pumps = [], []

    def pump(stream, sink, prefix):
        for line in iter(stream.readline, ""):
            sink.write(prefix + line)
            sink.flush()
        stream.close()
    for r in range(gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(gpus), LOCAL_WORLD_SIZE=str(gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL's intra-node transport needs it on this driver
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
        p = subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + argv, env=env, text=True, bufsize=1,
                             stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE)
        procs.append(p)
        if r == 0:
            pumps.append(threading.Thread(target=pump, args=(p.stdout, sys.stdout, ""), daemon=True))
        pumps.append(threading.Thread(target=pump, args=(p.stderr, sys.stderr, "[rank %d] " % r), daemon=True))
    for t in pumps:
        t.start()
    rc, live = 0, set(range(gpus))
    try:
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is not None:
                    live.discard(r)
                    if code != 0:
                        rc = code
                        print("bench.py launcher: rank %d exited with code %d — stopping ranks %s" % (r, code, sorted(live)),
                              file=sys.stderr, flush=True)
                        break
            time.sleep(0.05)
    finally:
        for r in live:                       # only reached with ranks alive after a failure or an interrupt
            procs[r].terminate()
        for r in live:
            try:
                procs[r].wait(timeout=20)
            except subprocess.TimeoutExpired:
                procs[r].kill()
    for t in pumps:
        t.join(timeout=10)
    return rc


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-replica":
        return _cpu_replica_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(WORKLOADS), default="wgan")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default, the headline): every GPU runs the reference's own batch; strong: the reference's batch is "
                         "split over the GPUs (64 -> 64/N samples, 16 -> 16/N shapes, 200 000 -> 200 000/N points per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip roofline / SDFNet side measurements")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))        # this process becomes the launcher of N ranks

    from shapegan_amd import parallel
    rank, world, local = parallel.init_distributed()
    if world != args.gpus:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE=%d: either run `python bench.py --gpus N` with no WORLD_SIZE in the "
                 "environment (it launches its own ranks) or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`"
                 % (args.gpus, world))
    torch.cuda.set_device(local % torch.cuda.device_count())
    strong = args.scaling == "strong"
    step, info, wgan_data = WORKLOADS[args.config](rank, world, strong)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # untimed priming when fewer than 3 warm-up steps were asked for: the first steps of a process grow the caching allocator
    # (multi-GB activation images of the SDFNet configs: 22 ms instead of 4 ms per step) and set kernel attributes
    for _ in range(max(0, 3 - args.warmup)):
        step()
    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        parallel.EXPOSED.enable()      # event pairs around every wait for a gradient exchange (two event records each)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    digests, exposed = None, None
    if world > 1:
        t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
        digests = replica_digests(info["optimizers"], world)      # after the timed region
        stream_ms, host_ms, waits = parallel.EXPOSED.totals()
        e = torch.tensor([stream_ms, host_ms], device="cuda", dtype=torch.float64)
        torch.distributed.all_reduce(e, op=torch.distributed.ReduceOp.MAX)
        exposed = (float(e[0].item()), float(e[1].item()), waits)

    if rank == 0:
        config = {"workload": info["workload"], "global_batch": info["batch"] * world, "per_gpu_batch": info["batch"],
                  "parallelism": "dp%d" % world}
        config.update(info["extra"])
        # weak: every GPU does the reference's step -> N steps' worth of units per step time; strong: the N GPUs share ONE
        # reference step (a rank's units_per_step is then 1/N of a step, or its 1/N share of the points)
        units = world * args.steps * info["units_per_step"]
        line = {
            "metric": info["metric"] if not strong else info["metric"].replace("/GPU", " GLOBAL, split over the GPUs"),
            "value": round(units / elapsed, 4),
            "unit": info["unit"], "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": config,
        }
        if world > 1:
            # what carried the gradient exchange, read back from the communicator itself (ncclCommCount / ncclGetVersion)
            line["comm"] = dict(parallel.TRANSPORT, transport=parallel.TRANSPORT["name"], world=world,
                                allreduce_bytes_per_step=info.get("allreduce_bytes_per_step"))
            line["comm"].pop("name", None)
            # what the overlap did NOT hide: how long the compute stream stood still at the buckets' waits (max over ranks), and
            # how long the host sat in the wait calls (a gloo exchange blocks the host; RCCL only enqueues)
            line["comm"]["exposed_ms_per_step"] = round(exposed[0] / args.steps, 4)
            line["comm"]["host_wait_ms_per_step"] = round(exposed[1] / args.steps, 4)
            line["comm"]["exchanges_waited_per_step"] = round(exposed[2] / args.steps, 2)
            from shapegan_amd import lib as sglib
            try:
                sglib.load_comm()
                line["comm"].update({k: v for k, v in sglib.COMM_INFO.items()})
            except RuntimeError as e:       # noqa: BLE001 — a diagnostic field
                line["comm"]["rccl_bind_error"] = str(e)
            line["comm"]["replicas_bit_identical"] = all(d == digests[0] for d in digests)
            line["comm"]["parameter_digests_rank0"] = digests[0]
        if args.config == "wgan":
            line["critic_updates_per_s"] = round(world * args.steps * 5 / elapsed, 3)
            if world == 1 and not args.no_extras:
                # single-kernel and side measurements belong to the N = 1 line only: at N > 1 the other ranks would sit in the
                # closing barrier while rank 0 measures
                line["roofline"], line["kernels"] = roofline_and_kernels()
                line["sdfnet"] = sdfnet_numbers()
                line["other_configs"] = other_configs()
                line["dropin_loop"] = dropin_loop_numbers()
                line["dropin_loop"]["trainer_step_launches_per_step"] = _count_launches(step)
                for k in ("resident_batches", "host_batches_copied_in_the_loop"):
                    line["dropin_loop"][k]["fraction_of_trainer_step"] = round(line["dropin_loop"][k]["steps_per_s"] / line["value"], 4)
            if world == 1 and not args.no_cpu_baseline:
                reals, zs, zg, (g_state, c_state) = wgan_data
                line["cpu_baseline"] = cpu_baseline(reals, zs, zg, g_state, c_state)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
