"""GPU tier: the data-parallel path of the multi-GPU BASELINE configs with world size 2 (two processes sharing the one GPU of
the test box, gloo as the transport — RCCL cannot form a communicator of two ranks on one device; the code path through
GradBucket / the optimizers is the one the 8-GPU run takes over RCCL).

  * train_hybrid_progressive_gan.py (configs[3], the reference's only DataParallel site, :62-68,102-166): one discriminator
    update with gradient penalty — double backward with the early (tail) slice of the flat gradient buffer exchanged from
    inside backward — on two half batches must produce, on both ranks, the gradient of the full batch;
  * train_sdf_autodecoder.py (configs[2]): one shape-sorted step on two halves of a point batch: network gradient and dense
    latent-table gradient equal the full batch's.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _setup(rank, world, port, device):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SG_DIST_BACKEND="gloo")
    from shapegan_amd import parallel
    r, w, _ = parallel.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    if device == "cuda":
        torch.cuda.set_device(0)
    else:   # the same trainers on CPU tensors (libshapegan_cpu.so): pin the modules' default device
        import shapegan_amd.util as U
        import shapegan_amd.model.gan as G
        U.device = G.default_device = torch.device("cpu")
        torch.set_num_threads(4)
    return parallel


def _prog_worker(rank, world, port, out_dir, device="cuda"):
    parallel = _setup(rank, world, port, device)
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridProgressiveGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    it, R, B = 1, 16, 4
    grid = torch.tensor(get_voxel_coordinates(R)).to(device)
    gen = torch.Generator().manual_seed(5)
    real = (torch.rand(B, R, R, R, generator=gen) * 2 - 1).to(device)
    z, alpha = torch.randn(B, 128, generator=gen).to(device), torch.rand(B, 1, 1, 1, generator=gen).to(device)

    def build():
        torch.manual_seed(21)                  # identical replicas by seed
        g, d = SDFNet(device=device), Discriminator().to(device)
        d.set_iteration(it)
        d.fade_in_progress = 0.6               # the fade-in blend is on the gradient penalty's double-backward path too
        return g, d, HybridProgressiveGANTrainer(g, d, grid, R)

    g, d, tr = build()
    assert tr.d_bucket.tail is not None and tr.d_opt.grad_scale == 0.5
    lo, hi = rank * B // world, (rank + 1) * B // world
    tr.discriminator_step(real[lo:hi], z[lo:hi], alpha[lo:hi])
    np.save(os.path.join(out_dir, "prog_grad%d.npy" % rank), (tr.d_opt.flat_grad * tr.d_opt.grad_scale).cpu().numpy())
    np.save(os.path.join(out_dir, "prog_param%d.npy" % rank), tr.d_opt.f.flat.cpu().numpy())
    if rank == 0:   # the full batch in one piece, no exchange (the buckets are never armed)
        g2, d2, ref = build()
        with torch.no_grad():
            fake = ref.generate(z)
        loss = d2(fake).mean() - d2(real).mean() + ref.gradient_penalty(real, fake, alpha)
        ref.d_opt.zero_grad()
        loss.backward()
        ref.d_opt.f.adopt_grads()
        np.save(os.path.join(out_dir, "prog_full.npy"), ref.d_opt.flat_grad.cpu().numpy())
        np.save(os.path.join(out_dir, "prog_slices.npy"), np.array([(o, p.numel()) for p, o in zip(ref.d_opt.f.params, ref.d_opt.f.offsets)]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _sdf_worker(rank, world, port, out_dir, device="cuda", n=131072):
    parallel = _setup(rank, world, port, device)
    from shapegan_amd import ops
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    pc, shapes, L = 4000, 8, 128
    gen = torch.Generator().manual_seed(6)
    pts = (torch.rand(shapes * pc, 3, generator=gen) * 2 - 1).to(device)
    sdf = (torch.rand(shapes * pc, generator=gen) * 0.3 - 0.15).to(device)
    table = (torch.randn(shapes, L, generator=gen) * 1e-2).to(device)
    idx = torch.randint(0, shapes * pc, (n,), generator=gen).to(device)

    def build():
        torch.manual_seed(22)
        net = SDFNet(latent_code_size=L, device=device)
        return net, SDFAutoDecoderTrainer(net, table.clone(), pts, sdf, pointcloud_size=pc)

    net, tr = build()
    half = n // world
    tr.step_sorted(idx[rank * half:(rank + 1) * half])
    np.save(os.path.join(out_dir, "sdf_grad%d.npy" % rank), (tr.net_opt.flat_grad * tr.net_opt.grad_scale).cpu().numpy())
    np.save(os.path.join(out_dir, "sdf_lat%d.npy" % rank), (tr.lat_opt.flat_grad * tr.lat_opt.grad_scale).cpu().numpy())
    np.save(os.path.join(out_dir, "sdf_table%d.npy" % rank), tr.latent_codes.detach().cpu().numpy())
    if rank == 0:
        net2, ref = build()
        ref.net_opt.zero_grad()
        ref.lat_opt.zero_grad()
        model_indices = torch.div(idx, pc, rounding_mode='floor')
        batch_latent = ops.gather_rows(ref.latent_codes, model_indices)
        out = net2(ops.gather_rows(pts, idx), batch_latent)
        loss = torch.mean(torch.abs(out - ref.sdf[idx])) + 0.01 * torch.mean(torch.pow(batch_latent, 2))   # train_sdf_autodecoder.py:88
        loss.backward()
        ref.net_opt.f.adopt_grads()
        ref.lat_opt.f.adopt_grads()
        np.save(os.path.join(out_dir, "sdf_full.npy"), ref.net_opt.flat_grad.cpu().numpy())
        np.save(os.path.join(out_dir, "sdf_lat_full.npy"), ref.lat_opt.flat_grad.cpu().numpy())
        np.save(os.path.join(out_dir, "sdf_slices.npy"), np.array([(o, p.numel()) for p, o in zip(ref.net_opt.f.params, ref.net_opt.f.offsets)]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def _used(tmp_path, name, total):
    used = np.zeros(total, dtype=bool)
    for o, n in np.load(tmp_path / name):
        used[o:o + n] = True
    return used


def _mostly_close(a, b, rtol, max_bad=2e-3, what=""):
    scale = np.abs(b).mean()
    bad = np.abs(a - b) > rtol * (scale + np.abs(b))
    assert bad.mean() <= max_bad, "%s: %.3f%% of entries differ (max err %.3e, scale %.3e)" % (what, 100 * bad.mean(), np.abs(a - b).max(), scale)


def check_prog(tmp_path):
    full = np.load(tmp_path / "prog_full.npy")
    used = _used(tmp_path, "prog_slices.npy", full.size)
    g0, g1 = np.load(tmp_path / "prog_grad0.npy")[used], np.load(tmp_path / "prog_grad1.npy")[used]
    np.testing.assert_array_equal(g0, g1)                                    # both ranks hold the same reduced gradient
    _mostly_close(g0, full[used], 2e-4, what="shard-averaged vs full-batch discriminator gradient")
    np.testing.assert_array_equal(np.load(tmp_path / "prog_param0.npy"), np.load(tmp_path / "prog_param1.npy"))   # replicas stay identical


def check_sdf(tmp_path):
    full = np.load(tmp_path / "sdf_full.npy")
    used = _used(tmp_path, "sdf_slices.npy", full.size)
    g0, g1 = np.load(tmp_path / "sdf_grad0.npy")[used], np.load(tmp_path / "sdf_grad1.npy")[used]
    np.testing.assert_array_equal(g0, g1)
    _mostly_close(g0, full[used], 3e-4, what="shard-averaged vs full-batch network gradient")
    l0, l1, lf = np.load(tmp_path / "sdf_lat0.npy"), np.load(tmp_path / "sdf_lat1.npy"), np.load(tmp_path / "sdf_lat_full.npy")
    np.testing.assert_array_equal(l0, l1)
    _mostly_close(l0, lf, 3e-4, what="dense latent-table gradient")
    np.testing.assert_array_equal(np.load(tmp_path / "sdf_table0.npy"), np.load(tmp_path / "sdf_table1.npy"))


def test_hybrid_progressive_discriminator_step_world2(tmp_path):
    mp.spawn(_prog_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    check_prog(tmp_path)


def test_sdf_autodecoder_sorted_step_world2(tmp_path):
    mp.spawn(_sdf_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    check_sdf(tmp_path)


def test_nn_dataparallel_of_the_shells_on_a_duplicated_device():
    """with device_count() > 1 the unmodified train_hybrid_progressive_gan.py wraps both modules in
    nn.DataParallel (:62-68) — replicate() makes shallow copies whose parameters are broadcast copies, parallel_apply drives them from
    one thread per device.  The one GPU of a test box stands in for two (device_ids=[0, 0]: two replicas, two threads, one device):
    outputs equal the unwrapped module's, and the gradients that the reduce-add brings back to the wrapped module equal the plain
    backward's.  What this exercises in the shells: no state keyed on the module object survives into a replica (kept weight images
    are not taken for the replicas' broadcast weights — nobody announces writes to them —, the SDFNet pack is rebuilt per call),
    and two host threads inside the library at once on one device."""
    import torch.nn as nn
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    torch.manual_seed(12)
    net = SDFNet()
    pts = (torch.rand(8192, 3, device="cuda") * 2 - 1)
    lat = torch.randn(8192, 128, device="cuda") * 0.1
    want = net(pts, lat)
    want.square().mean().backward()
    ref_grads = [p.grad.clone() for p in net.parameters()]
    net.zero_grad()
    try:
        dp = nn.DataParallel(net, device_ids=[0, 0])
        got = dp(pts, lat)
    except (RuntimeError, AssertionError) as e:           # a torch build that refuses a repeated device id
        pytest.skip("nn.DataParallel on a duplicated device is not possible here: %s" % e)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
    got.square().mean().backward()
    for p, r in zip(net.parameters(), ref_grads):
        torch.testing.assert_close(p.grad, r, rtol=1e-4, atol=1e-6 * float(r.abs().max() + 1e-30))
    disc = Discriminator().cuda()
    disc.set_iteration(2)
    x = (torch.rand(8, 32, 32, 32, device="cuda") * 2 - 1).requires_grad_()
    want = disc(x)
    want.mean().backward()
    ref_x, ref_grads = x.grad.clone(), [None if p.grad is None else p.grad.clone() for p in disc.parameters()]
    disc.zero_grad()
    x.grad = None
    got = nn.DataParallel(disc, device_ids=[0, 0])(x)
    torch.testing.assert_close(got, want, rtol=1e-5, atol=1e-6)
    got.mean().backward()
    torch.testing.assert_close(x.grad, ref_x, rtol=1e-4, atol=1e-7)
    for p, r in zip(disc.parameters(), ref_grads):
        if r is not None:
            torch.testing.assert_close(p.grad, r, rtol=1e-4, atol=1e-6 * float(r.abs().max() + 1e-30))
