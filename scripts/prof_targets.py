"""Workloads for the rocprofv3 counter passes of a round (run one mode per pass; see scripts/profile_round.sh).

    mfma    : SDFNet fused forward / training steps (sdfnet_fwd, sdfnet_bwd, gemm_nt_bigk batched), the three halo conv forms and
              the one-channel layers at the critic's 128-sample pass
              -> SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE
    hbm     : two WGAN 5+1 steps (covers the Cin=1 / Cout=1 edge kernels) + calibration streams of a known byte count
    configs : BASELINE configs[2] / [3] / [4] (their kernels run at other shapes than the WGAN's: kept out of the two passes
              above so that those stay per-shape medians), used with both counter groups
           (1 GiB read + 1 GiB write through b32 loads (sg_axpby) and through b128 loads (sg_act_fwd)) -> FETCH_SIZE / WRITE_SIZE
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops  # noqa: E402
from shapegan_amd.lib import check, ptr, stream  # noqa: E402

mode = sys.argv[1]
torch.manual_seed(0)
if mode == "mfma":
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    from shapegan_amd.util import get_voxel_coordinates
    net = SDFNet()
    grid = torch.tensor(get_voxel_coordinates(32)).cuda().repeat((8, 1))
    z = torch.randn(8, 128, device="cuda")
    with torch.no_grad():
        for _ in range(4):
            net.forward_shapes(grid, z, 32768)
    pc, shapes = 200000, 64
    pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
    sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
    for npts, lat in ((200000, 256),):     # BASELINE configs[2]; the per-kernel medians are then those of this step
        tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), torch.randn(shapes, lat, device="cuda") * 1e-2, pts, sdf,
                                   pointcloud_size=pc)
        idx = torch.randint(0, shapes * pc, (npts,), device="cuda")
        for _ in range(6):
            tr.step(idx)
    x = torch.randn(128, 64, 16, 16, 16, device="cuda")
    w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
    b = torch.zeros(128, device="cuda")
    for _ in range(4):
        y = ops.conv_fwd_raw(x, w, b, 1, 0.2)
        ops.conv_dgrad_raw(y, w, None, 64)
        ops.conv_wgrad_raw(y, x, 64)
    # the one-channel layers at the critic's shapes (28 flop/B: their matrix-pipe share is what bounds them next to HBM)
    x32 = torch.randn(128, 1, 32, 32, 32, device="cuda")
    w1 = torch.randn(64, 1, 4, 4, 4, device="cuda") * 0.1
    b1 = torch.zeros(64, device="cuda")
    y16, ya = torch.randn(128, 64, 16, 16, 16, device="cuda"), torch.randn(128, 64, 16, 16, 16, device="cuda")
    for _ in range(4):
        ops.conv_fwd_raw(x32, w1, b1, 1, 0.2)
        ops.conv_wgrad_raw(y16, x32, 1)
        ops.conv_wgrad_act_raw(y16, ya, x32, 1, 0.2)
        ops.conv_dgrad_raw(y16[:64].contiguous(), w1, None, 1)
elif mode == "sdfstep":
    # six 200 000-point / latent-256 auto-decoder steps alone (scripts/kernel_pmc.sh sdfnet ...)
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    pc, shapes, lat = 200000, 64, 256
    pts = torch.rand(shapes * pc, 3, device="cuda") * 2 - 1
    sdf = torch.rand(shapes * pc, device="cuda") * 0.2 - 0.1
    tr = SDFAutoDecoderTrainer(SDFNet(latent_code_size=lat), torch.randn(shapes, lat, device="cuda") * 1e-2, pts, sdf, pointcloud_size=pc)
    idx = torch.randint(0, shapes * pc, (200000,), device="cuda")
    for _ in range(6):
        tr.step(idx)
elif mode == "hbm":
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    tr = WGANTrainer(Generator(), Discriminator())
    reals = [(torch.rand(64, 32, 32, 32) * 2 - 1).cuda() for _ in range(5)]
    zs = [torch.randn(64, 128).cuda() for _ in range(5)]
    zg = torch.randn(64, 128).cuda()
    for _ in range(2):
        tr.step(reals, zs, zg)
    n = 1 << 28                                                   # 1 GiB of floats: past the 256 MiB Infinity Cache
    a, o = torch.randn(n, device="cuda"), torch.empty(n, device="cuda")
    lib = ops.L.load()
    for _ in range(3):
        check(lib.sg_axpby(ptr(a), None, ptr(o), n, 2.0, 0.0, stream()), "axpby")      # b32 loads / stores
        check(lib.sg_act_fwd(ptr(a), ptr(o), n, 1, 0.2, stream()), "act_fwd")          # b128 loads / stores
elif mode == "configs":
    # BASELINE configs[2] / [3] / [4]: two steps / 5+1 units each (the auto-decoder step at 200 000 points, 16 x 64^3 SDFNet
    # evaluations + the progressive discriminator with the gradient penalty's double backward, the batch-8 hybrid WGAN)
    sys.argv = [sys.argv[0]]
    import bench
    for cfg in ("sdf", "hybrid_progressive", "hybrid_wgan"):
        step, _, _ = bench.WORKLOADS[cfg](0)
        for _ in range(2):
            step()
        del step
        torch.cuda.empty_cache()
    # two critic and two generator updates of the PointNet GAN at 12 x 16 384 points (the LayerNorm form of
    # the fused MLP kernels, the fused selection pass of the critic)
    from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
    from shapegan_amd.train_steps import PointGANTrainer
    tr = PointGANTrainer(SDFGenerator(128, 256, 8, True).cuda(), PointNet(1).cuda())
    u = torch.cat([torch.rand(12, 16384, 3) * 2 - 1, torch.rand(12, 16384, 1) * 0.2 - 0.1], -1).cuda()
    z, a = torch.randn(12, 128, device="cuda"), torch.rand(12, 1, 1, device="cuda")
    for _ in range(3):
        tr.critic_step(u, z, a)
        tr.generator_step(u, z)
    # ... and the dense form of the generator's training kernels (what the reference's own loop runs: every point recorded)
    g = tr.generator
    for _ in range(3):
        g.zero_grad()
        g(u[..., :3], z).sum().backward()
torch.cuda.synchronize()
