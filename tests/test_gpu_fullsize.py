"""GPU tier: BASELINE-size parity through the PUBLIC, auto-dispatched entry points (the LDS-halo kernels are
selected by grid size, so small-N tests never reach them through the public entries).  Every case here runs the layer /
step at the batch the BASELINE configs use and compares directly with the CPU oracle (torch fp32 ops = the reference's own
arithmetic engine): F.conv3d / F.conv_transpose3d for the layers, WGANOracle / SDFAutoDecoderOracle for the steps."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def close(a, b, rtol=RTOL, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    atol = rtol * max(float(b.abs().mean()), 1e-30)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: what + ": " + m)


def close_mostly(a, b, rtol=RTOL, max_bad_frac=1e-3, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    atol = rtol * max(float(b.abs().mean()), 1e-30)
    bad = (a - b).abs() > (atol + rtol * b.abs())
    frac = float(bad.float().mean())
    assert frac <= max(max_bad_frac, 2.0 / bad.numel()), "%s: %.4f%% of elements out of tolerance (max err %.3e, scale %.3e)" % (
        what, 100 * frac, float((a - b).abs().max()), float(b.abs().mean()))
    if bad.any():
        assert float((a - b).abs().max()) <= 0.5 * float(b.abs().max()), what + ": outlier larger than the tensor scale"


@pytest.mark.parametrize("N,Ci,Co,R", [(128, 64, 128, 16), (128, 128, 256, 8), (64, 64, 128, 16), (16, 32, 64, 32), (16, 128, 256, 8),
                                       (3, 128, 256, 8), (5, 64, 96, 8)])
def test_conv3d_full_size_vs_aten(N, Ci, Co, R):
    """gan.Discriminator layers 2 / 3 at the critic's concatenated fake+real batch (2 x 64), the generator-step batch (64),
    the progressive discriminator's 32 -> 64 stage at batch 16 and the 4^3-output layer at the hybrid GANs' small batches (the halo
    kernel with its input channels split over 4 / 8 / 2 workgroups + the split-K finalize): forward (+ LeakyReLU epilogue), input gradient, weight
    gradient and bias gradient through ops.conv3d_k4s2p1, i.e. whatever kernel the dispatcher picks at this size."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(N + Ci + Co + R)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Co, Ci, 4, 4, 4) / (Ci * 64) ** 0.5
    b = torch.randn(Co) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, br, stride=2, padding=1)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    # the fused LeakyReLU epilogue: forward only (a mask flip at an output within rounding of 0 is legitimate in fp32 and
    # moves a 64 x 64 block of the weight gradient by a finite amount, so gradients are compared on the linear op;
    # the mask itself is covered at small sizes in test_gpu_ops.py)
    with torch.no_grad():
        close(ops.conv3d_k4s2p1(xg, wg, bg, ACT_LEAKY, 0.2), F.leaky_relu(y_ref, 0.2), what="forward + LeakyReLU")
    y = ops.conv3d_k4s2p1(xg, wg, bg)
    close(y, y_ref, what="forward")
    y.backward(dy.cuda())
    close(xg.grad, xr.grad, what="input gradient")
    close(wg.grad, wr.grad, what="weight gradient")
    close(bg.grad, br.grad, what="bias gradient")


@pytest.mark.parametrize("N,Ci,Co,R", [(64, 256, 128, 4), (64, 128, 64, 8), (64, 64, 1, 16)])
def test_conv_transpose3d_full_size_vs_aten(N, Ci, Co, R):
    """gan.Generator layers 2 / 3 / 4 at batch 64 through ops.conv_transpose3d_k4s2p1 (model/gan.py:13,17,21)."""
    from shapegan_amd import ops
    torch.manual_seed(N + Ci + Co + R)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Ci, Co, 4, 4, 4) / (Ci * 8) ** 0.5
    b = torch.randn(Co) * 0.1
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv_transpose3d(xr, wr, br, stride=2, padding=1)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, wg, bg = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    y = ops.conv_transpose3d_k4s2p1(xg, wg, bg)
    close(y, y_ref, what="forward")
    y.backward(dy.cuda())
    close(xg.grad, xr.grad, what="input gradient")
    close(wg.grad, wr.grad, what="weight gradient")
    close(bg.grad, br.grad, what="bias gradient")


def _state(module):
    return {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}


def _state64(module):
    return {k: (v.detach().cpu().double() if v.is_floating_point() else v.detach().cpu().clone())
            for k, v in module.state_dict().items()}


def test_wgan_batch64_update_vs_oracle():
    """BASELINE configs[1]: one critic update and one generator update of train_wgan.py:60-84 at batch 64 against WGANOracle
    on the same inputs — critic loss, the 64 critic scores of each side, every gradient, the generator's BatchNorm running
    statistics.  Gradients are sums over up to 128 x 4096 products with heavy cancellation, so the comparison is the one the
    trajectory tests use: as close to the fp64 oracle as 1e-4 of the tensor's scale plus 4x the fp32 oracle's own error."""
    from test_gpu_modules import check_against_oracles
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    torch.manual_seed(0)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    g, c = Generator(), Discriminator()
    o32, o64 = O.WGANOracle(_state(g), _state(c)), O.WGANOracle(_state64(g), _state64(c))
    tr = WGANTrainer(g, c)
    gen = torch.Generator().manual_seed(1000)
    real = torch.rand(64, 32, 32, 32, generator=gen) * 2 - 1
    z, zg = torch.randn(64, 128, generator=gen), torch.randn(64, 128, generator=gen)
    loss, out_fake, out_real = tr.critic_step(real.cuda(), z.cuda())
    grads = {k: p.grad.detach().clone() for k, p in c.named_parameters()}
    r32, r64 = o32.critic_step(real, z), o64.critic_step(real.double(), z.double())
    np.testing.assert_allclose(loss.item(), r64[0].item(), rtol=RTOL, atol=1e-7)
    check_against_oracles(out_fake, r32[1], r64[1], "critic(fake)")
    check_against_oracles(out_real, r32[2], r64[2], "critic(real)")
    gscale = max(float(o64.C[k].grad.abs().mean()) for k in grads)
    # (128 samples x up to 64 x 4096 LeakyReLU outputs per layer: a few more kink flips than in the small cases -> 0.3 %)
    for k in grads:
        check_against_oracles(grads[k], o32.C[k].grad, o64.C[k].grad, "critic grad " + k, gscale=gscale, max_frac=3e-3)
    gl, out = tr.generator_step(zg.cuda())
    g32, g64 = o32.generator_step(zg), o64.generator_step(zg.double())
    np.testing.assert_allclose(gl.item(), g64[0].item(), rtol=RTOL, atol=1e-7)
    check_against_oracles(out, g32[1], g64[1], "critic(generator(z))")
    gscale = max(float(v.grad.abs().mean()) for k, v in o64.G.items() if v.requires_grad)
    for k, p in g.named_parameters():
        check_against_oracles(p.grad, o32.G[k].grad, o64.G[k].grad, "generator grad " + k, gscale=gscale, max_frac=3e-3)
    for k, v in g.state_dict().items():
        if "running_" in k:
            check_against_oracles(v, o32.G[k], o64.G[k], k)


def test_wgan_headline_step_vs_oracle(batch=64):
    """The exact code path bench.py's headline times: ONE full 5 + 1 `WGANTrainer.step` at batch 64 with the
    shipped defaults — real batches delivered into `real_slots`, the unit's latents as rows of one tensor, the generator
    evaluations of critic updates 2..5 as one grouped pass, the inference generator's last BatchNorm folded into the final ConvT —
    against `WGANOracle.step` (train_wgan.py:60-84 restated on torch CPU ops) in fp32 AND fp64 from the same state and inputs.
    Compared after the whole unit (six optimizer updates deep): the last critic update's loss and 2 x 64 scores and its
    gradients by the fp64-truth criterion of test_wgan_batch64_update_vs_oracle; the generator's BatchNorm running statistics and
    batch counters after its six evaluations; and every parameter's total movement over the unit, which RMSprop makes
    sign-like (|step| ~ 10 lr on the first update whatever the gradient's size): an entry counts as deviating when it is further
    from the fp64 oracle than 0.5 % of the tensor's typical movement, and the native path may deviate on at most three times the
    fraction of entries on which the fp32 oracle — the reference's own arithmetic — deviates, plus 0.1 % (or four entries of a small
    tensor: a gradient at the 1e-8 level of RMSprop's eps moves its entry by anything between 0 and 10 lr)."""
    from test_gpu_modules import check_against_oracles
    from shapegan_amd.model import stack
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    assert stack.FUSE_BN_INTO_LAST_CONV_TRANSPOSE is True         # the shipped default is what is under test
    torch.manual_seed(0)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    g, c = Generator(), Discriminator()
    g0, c0 = _state64(g), _state64(c)
    o32, o64 = O.WGANOracle(_state(g), _state(c)), O.WGANOracle(g0, c0)
    tr = WGANTrainer(g, c)
    gen = torch.Generator().manual_seed(1000)
    reals_host = [torch.rand(batch, 32, 32, 32, generator=gen) * 2 - 1 for _ in range(5)]
    zs_host = torch.randn(5, batch, 128, generator=gen)
    zg = torch.randn(batch, 128, generator=gen)
    slots = tr.real_slots(batch, resolution=32, updates=5)
    for slot, r in zip(slots, reals_host):
        slot.copy_(r.reshape(batch, 1, 32, 32, 32))
    zs = list(zs_host.cuda().unbind(0))
    loss, out_fake, out_real = tr.step(slots, zs, zg.cuda())
    assert g.layers[1].num_batches_tracked.item() == 6            # five critic updates + the generator update evaluated it
    r32 = o32.step(reals_host, list(zs_host.unbind(0)), zg)
    r64 = o64.step([r.double() for r in reals_host], list(zs_host.double().unbind(0)), zg.double())
    # the fifth critic update, five critic steps and one generator step behind the common start
    noise = abs(r32[0].item() - r64[0].item())
    assert abs(loss.item() - r64[0].item()) <= RTOL * abs(r64[0].item()) + 4 * noise + 1e-7
    check_against_oracles(out_fake, r32[1], r64[1], "critic(fake), update 5")
    check_against_oracles(out_real, r32[2], r64[2], "critic(real), update 5")
    gscale = max(float(v.grad.abs().mean()) for v in o64.C.values() if v.requires_grad)
    for k, p in c.named_parameters():
        check_against_oracles(p.grad, o32.C[k].grad, o64.C[k].grad, "critic grad, update 5: " + k, gscale=gscale, max_frac=3e-3)
    for k, v in g.state_dict().items():
        if "running_" in k:
            check_against_oracles(v, o32.G[k], o64.G[k], k)
        elif "num_batches_tracked" in k:
            assert int(v) == int(o32.G[k])
    for name, mod, start, ora32, ora64 in (("generator", g, g0, o32.G, o64.G), ("critic", c, c0, o32.C, o64.C)):
        for k, p in mod.named_parameters():
            d = p.detach().double().cpu() - start[k]
            d32, d64 = ora32[k].detach().double() - start[k], ora64[k].detach() - start[k]
            unit = 5e-3 * max(float(d64.abs().mean()), 1e-12)
            f_native, f_ref = float(((d - d64).abs() > unit).double().mean()), float(((d32 - d64).abs() > unit).double().mean())
            assert f_native <= 3 * f_ref + max(1e-3, 4.0 / d.numel()), "%s %s: %.3f %% of the entries moved differently from the fp64 oracle (fp32 oracle: %.3f %%)" % (
                name, k, 100 * f_native, 100 * f_ref)


def test_sdf_autodecoder_200k_L256_step_vs_oracle():
    """BASELINE configs[2]: one auto-decoder step (train_sdf_autodecoder.py:77-91) at 200 000 points, latent 256, through the
    shape-sorted data flow, against SDFAutoDecoderOracle in fp32 AND fp64: loss, network gradients, dense latent-table gradient,
    by the same criterion as the WGAN update above (1e-4 of the tensor's scale around the fp64 truth plus 4x the fp32 oracle's
    own error; sums over 200 000 points through 7 ReLU layers: up to 0.2 % kink outliers)."""
    from test_gpu_modules import check_against_oracles
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    torch.manual_seed(2)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    pc, shapes, L, n = 20000, 64, 256, 200000
    pts = torch.rand(shapes * pc, 3) * 2 - 1
    sdf = torch.rand(shapes * pc) * 0.3 - 0.15
    table = torch.randn(shapes, L) * 1e-2
    idx = torch.randint(0, shapes * pc, (n,))
    net = SDFNet(latent_code_size=L)
    o32 = O.SDFAutoDecoderOracle(_state(net), table, pts, sdf, pointcloud_size=pc)
    o64 = O.SDFAutoDecoderOracle(_state64(net), table.double(), pts.double(), sdf.double(), pointcloud_size=pc)
    tr = SDFAutoDecoderTrainer(net, table.clone().cuda(), pts.cuda(), sdf.cuda(), pointcloud_size=pc)
    loss = tr.step(idx.cuda())
    l32, l64 = o32.step(idx), o64.step(idx)
    np.testing.assert_allclose(loss.item(), l64.item(), rtol=RTOL)
    gscale = max(float(v.grad.abs().mean()) for v in o64.P.values() if v.requires_grad)
    for k, p in net.named_parameters():
        check_against_oracles(p.grad, o32.P[k].grad, o64.P[k].grad, "net grad " + k, gscale=gscale, max_frac=2e-3)
    check_against_oracles(tr.latent_codes.grad, o32.latent_codes.grad, o64.latent_codes.grad, "latent table grad", max_frac=2e-3)


def _inject_fake(oracle, fake_cpu, dtype):
    """The oracle's generator replaced by a given batch of generated grids: at 16 x 64^3 = 4.2 M points the SDFNet generator
    (3.9 TFLOP forward) is out of the CPU's reach, so the discriminator side of configs[3] is checked on the grids the native
    generator produced (its own parity at this size: test_gpu_modules.py::test_full_size_hybrid_progressive_config and the
    masked-subset check below)."""
    oracle.generate = lambda z, f=fake_cpu.to(dtype): f


@pytest.mark.parametrize("fade", [1.0, 0.5])
def test_hybrid_progressive_it3_b16_steps_vs_oracle(fade):
    hybrid_progressive_case(3, 16, fade, 50000)


def hybrid_progressive_case(iteration, B, fade, subset):
    """BASELINE configs[3] at its benchmarked size — train_hybrid_progressive_gan.py iteration=3 (64^3), batch 16 — against
    HybridProgressiveGANOracle in fp32 and fp64 (train_hybrid_progressive_gan.py:102-111,134-166):
      (1) one discriminator update with gradient penalty (D(fake), D(real), lerp, double backward): loss, penalty and every
          parameter gradient, both sides fed the same generated grids;
      (2) the generator update's discriminator half: loss -mean D(G(z)) and dLoss/dfake [16,64,64,64];
      (3) its SDFNet half: the 4.2 M-point backward with the upstream gradient restricted to a 50 000-point subset (every other
          entry zero), against the oracle evaluated on exactly those points (reference semantics: tiled latents, cat + Linear)."""
    from test_gpu_modules import check_against_oracles
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridProgressiveGANTrainer, frozen
    from shapegan_amd.util import get_voxel_coordinates
    from shapegan_amd import ops
    torch.manual_seed(31)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    R = (8, 16, 32, 64)[iteration]
    g, d = SDFNet(), Discriminator().cuda()
    d.set_iteration(iteration)
    d.fade_in_progress = fade
    grid = torch.tensor(get_voxel_coordinates(R))
    o32 = O.HybridProgressiveGANOracle(_state(g), _state(d), grid, iteration, fade)
    o64 = O.HybridProgressiveGANOracle(_state64(g), _state64(d), grid.double(), iteration, fade)
    tr = HybridProgressiveGANTrainer(g, d, grid.cuda(), R)
    gen = torch.Generator().manual_seed(3100)
    real = torch.rand(B, R, R, R, generator=gen) * 2 - 1
    z1, z2 = torch.randn(B, 128, generator=gen), torch.randn(B, 128, generator=gen)
    alpha = torch.rand(B, 1, 1, 1, generator=gen)

    # (2) + (3) first: the generator update sees the initial discriminator on both sides
    fake = tr.generate(z1.cuda())                       # [16,64,64,64] with its autograd graph (4.2 M points)
    fake.retain_grad()
    with frozen(d):
        loss_g = ops.neg_mean(d(fake))
    loss_g.backward()
    g_fake = fake.grad.detach().clone()
    refs = []
    for o, dt in ((o32, torch.float32), (o64, torch.float64)):
        f = fake.detach().cpu().to(dt).requires_grad_(True)
        lo = -o.disc(f).mean()
        refs.append((lo.detach(), torch.autograd.grad(lo, f)[0]))
    np.testing.assert_allclose(loss_g.item(), refs[1][0].item(), rtol=RTOL, atol=1e-7)
    check_against_oracles(g_fake, refs[0][1], refs[1][1], "dLoss/dfake through the 64^3 discriminator", max_frac=3e-3)

    sel = torch.randperm(B * R ** 3, generator=gen)[:subset]
    pts_s, lat_s = grid[sel % R ** 3], z1[sel // R ** 3]
    fragile = O.sdfnet_min_preactivation(o64.G, pts_s.double(), lat_s.double()) < 1e-6     # points on a ReLU kink (oracle docstring): zero upstream gradient
    assert float(fragile.float().mean()) < 0.05
    up = refs[1][1].reshape(-1)[sel].float() * (~fragile).float()
    u = torch.zeros(B * R ** 3)
    u[sel] = up
    for p in g.parameters():
        p.grad = None
    fake2 = tr.generate(z1.cuda())
    fake2.backward(u.reshape(B, R, R, R).cuda())
    sub = []
    for o, dt in ((o32, torch.float32), (o64, torch.float64)):
        out = O.sdfnet_forward(o.G, pts_s.to(dt), lat_s.to(dt))
        out.backward(up.to(dt))
        sub.append(out.detach())
    check_against_oracles(fake2.detach().reshape(-1).cpu()[sel], sub[0], sub[1], "generated sdf at the subset")
    gscale = max(float(v.grad.abs().mean()) for v in o64.G.values() if v.requires_grad)
    for k, p in g.named_parameters():
        check_against_oracles(p.grad, o32.G[k].grad, o64.G[k].grad, "generator grad (subset upstream) " + k, gscale=gscale,
                              max_frac=2e-3)

    # (1) the discriminator update
    with torch.no_grad():
        fake_d = tr.generate(z2.cuda()).cpu()
    _inject_fake(o32, fake_d, torch.float32)
    _inject_fake(o64, fake_d, torch.float64)
    dl, gp = tr.discriminator_step(real.cuda(), z2.cuda(), alpha.cuda())
    r32 = o32.discriminator_step(real, z2, alpha)
    r64 = o64.discriminator_step(real.double(), z2.double(), alpha.double())
    np.testing.assert_allclose([dl.item(), gp.item()], [r64[0].item(), r64[1].item()], rtol=RTOL, atol=1e-6)
    named = dict(d.named_parameters())
    used = [k for k in named if o64.D[k].grad is not None]
    assert sorted(k for k in named if named[k].grad is not None) == sorted(used)     # the head and the stages up to `iteration`
    gscale = max(float(o64.D[k].grad.abs().mean()) for k in used)
    for k in used:
        check_against_oracles(named[k].grad, o32.D[k].grad, o64.D[k].grad, "discriminator grad " + k, gscale=gscale, max_frac=3e-3)


def test_hybrid_wgan_32_b8_steps_vs_oracle():
    hybrid_wgan_case(8)


def hybrid_wgan_case(B):
    """BASELINE configs[4] at its benchmarked size — train_hybrid_wgan.py, SDFNet generator sampled to 32^3, batch 8 (262 144
    points per generator evaluation), critic with weight clipping — one critic update and one generator update against
    HybridWGANOracle in fp32 and fp64 (train_hybrid_wgan.py:83-115), the generator on the CPU at full size: losses, the 8 scores
    of each side, every gradient of both networks."""
    from test_gpu_modules import check_against_oracles
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridWGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(41)
    torch.set_num_threads(min(32, torch.get_num_threads()))
    g, c = SDFNet(), Discriminator()
    grid = torch.tensor(get_voxel_coordinates(32))
    o32 = O.HybridWGANOracle(_state(g), _state(c), grid)
    o64 = O.HybridWGANOracle(_state64(g), _state64(c), grid.double())
    tr = HybridWGANTrainer(g, c, grid.cuda())
    gen = torch.Generator().manual_seed(4100)
    real = torch.rand(B, 32, 32, 32, generator=gen) * 0.2 - 0.1          # rescale_sdf = False (train_hybrid_wgan.py:31)
    z1, z2 = torch.randn(B, 128, generator=gen), torch.randn(B, 128, generator=gen)
    loss, out_fake, out_real = tr.critic_step(real.cuda(), z1.cuda())
    c_grads = {k: p.grad.detach().clone() for k, p in c.named_parameters()}
    r32, r64 = o32.critic_step(real, z1), o64.critic_step(real.double(), z1.double())
    np.testing.assert_allclose(loss.item(), r64[0].item(), rtol=RTOL, atol=1e-7)
    check_against_oracles(out_fake, r32[1], r64[1], "critic(fake)")
    check_against_oracles(out_real, r32[2], r64[2], "critic(real)")
    gscale = max(float(o64.C[k].grad.abs().mean()) for k in c_grads)
    for k in c_grads:
        check_against_oracles(c_grads[k], o32.C[k].grad, o64.C[k].grad, "critic grad " + k, gscale=gscale, max_frac=3e-3)
    gl, out = tr.generator_step(z2.cuda())
    g32, g64 = o32.generator_step(z2), o64.generator_step(z2.double())
    np.testing.assert_allclose(gl.item(), g64[0].item(), rtol=RTOL, atol=1e-7)
    check_against_oracles(out, g32[1], g64[1], "critic(generator(grid, z))")
    gscale = max(float(v.grad.abs().mean()) for v in o64.G.values() if v.requires_grad)
    for k, p in g.named_parameters():
        check_against_oracles(p.grad, o32.G[k].grad, o64.G[k].grad, "generator grad " + k, gscale=gscale, max_frac=3e-3)


def test_sdfnet_backward_beyond_2m_points_equals_its_halves():
    """configs[3]'s generator step evaluates 16 x 64^3 = 4.2 M points in one call: per-layer images of more than 2 GiB.  The
    backward's addressing is 32-bit lane offset + scalar row offset from a per-wave base; this checks a 2.46 M-point call
    (every image 2.5 GB) against the same work done as two 1.2 M-point calls (the size class the oracle tests cover), everything to 1e-4 / 2e-6 absolute on the outputs."""
    from shapegan_amd.model.sdf_net import SDFNet
    torch.manual_seed(5)
    net = SDFNet(latent_code_size=32).cuda()
    S, pps = 10, 245760
    pts = (torch.rand(S * pps, 3, device="cuda") * 2 - 1)
    z = torch.randn(S, 32, device="cuda") * 0.3
    w = torch.randn(S * pps, device="cuda")

    def run(lo, hi):
        p = pts[lo * pps:hi * pps].clone().requires_grad_(True)
        zz = z[lo:hi].clone().requires_grad_(True)
        out = net.forward_shapes(p, zz, pps)
        (out * w[lo * pps:hi * pps]).sum().backward()
        grads = [q.grad.clone() for q in net.parameters()]
        for q in net.parameters():
            q.grad = None
        return out.detach(), p.grad, zz.grad, grads

    out, dp, dz, g = run(0, S)
    out_a, dp_a, dz_a, g_a = run(0, S // 2)
    out_b, dp_b, dz_b, g_b = run(S // 2, S)
    # (not bit-equal: the [S,L] x [L,256] GEMM that folds the latents into per-shape biases splits K differently for 10 and 5 rows)
    torch.testing.assert_close(out, torch.cat([out_a, out_b]), rtol=0, atol=2e-6)
    close(dp, torch.cat([dp_a, dp_b]), what="d points")
    close(dz, torch.cat([dz_a, dz_b]), what="d latent")
    for (name, _), full, ha, hb in zip(net.named_parameters(), g, g_a, g_b):
        close_mostly(full, ha + hb, what="grad " + name)      # sums over 2.46 M points: isolated entries differ at rounding level


def test_sdfnet_sorted_batches_at_random_sizes_equal_their_pieces():
    """Fuzz of the SDFNet training kernels at ragged sizes (scripts/fuzz_sdf.py runs more of it): a shape-sorted batch evaluated in
    one call against the same batch evaluated as two calls cut at a random point — other tile plans and partial-sum layouts, the
    same per-point arithmetic, so no ReLU kink can flip between the two: outputs bit-equal, latent-table and parameter gradients
    equal to summation-order rounding."""
    import random
    from shapegan_amd.model.sdf_net import SDFNet
    random.seed(7)
    torch.manual_seed(0)
    nets = {L: SDFNet(latent_code_size=L).cuda() for L in (16, 128)}
    for it in range(24):
        L = random.choice((16, 128))
        net = nets[L]
        N = random.choice((1, 31, 33, 63, 64, 65, 127, 129)) if it % 4 == 0 else random.randint(2, random.choice((300, 5000, 40000, 70000)))
        S = random.randint(1, min(40, N))
        sid = torch.sort(torch.randint(0, S, (N,), device="cuda"))[0]
        pts = torch.rand(N, 3, device="cuda") * 2 - 1
        table = torch.randn(S, L, device="cuda") * 0.5
        w = torch.randn(N, device="cuda")

        def run(lo, hi):
            t = table.clone().requires_grad_(True)
            for p in net.parameters():
                p.grad = None
            sg = torch.zeros(S + 1, dtype=torch.int64, device="cuda")
            sg[1:] = torch.cumsum(torch.bincount(sid[lo:hi], minlength=S), 0)
            out = net.forward_segments(pts[lo:hi].contiguous(), t, sid[lo:hi].int().contiguous(), sg)
            (out * w[lo:hi]).sum().backward()
            return out.detach(), t.grad.clone(), [p.grad.clone() for p in net.parameters()]

        full = run(0, N)
        if N == 1:
            continue
        cut = random.randint(1, N - 1)
        a, b = run(0, cut), run(cut, N)
        what = "N=%d cut=%d S=%d L=%d" % (N, cut, S, L)
        assert torch.equal(full[0], torch.cat([a[0], b[0]])), what + ": outputs depend on the tiling"
        close(full[1], a[1] + b[1], rtol=1e-3, what=what + " d latent table")
        for (name, _), g, ga, gb in zip(net.named_parameters(), full[2], a[2], b[2]):
            close(g, ga + gb, rtol=1e-3, what=what + " grad " + name)


def test_point_gan_updates_at_12x16384_points_sparse_vs_dense():
    """SURVEY.md 8f rank 4 at the (num_points, batch) = (16 384, 12) stage of train_point_gan.py:29-34: the trainer's updates —
    fused selection pass, recorded passes on the 512 selected points per cloud, generator update on the selected points — against
    the same trainer forced onto the dense path (every point recorded, fused LayerNorm-MLP backward over all 196 608 points):
    losses and every parameter gradient.  Size-independent property on top: moving points that hold NO maximum (without letting them
    take one over) changes neither the critic's output nor any gradient."""
    import copy
    from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
    from shapegan_amd.train_steps import PointGANTrainer
    torch.manual_seed(21)
    g, d = SDFGenerator(128, 256, 8, True, dropout=0.0).cuda(), PointNet(out_channels=1).cuda()
    B, P = 12, 16384
    uniform = torch.cat([torch.rand(B, P, 3) * 2 - 1, torch.rand(B, P, 1) * 0.2 - 0.1], -1).cuda()
    z1, z2, alpha = torch.randn(B, 128).cuda(), torch.randn(B, 128).cuda(), torch.rand(B, 1, 1).cuda()
    runs = {}
    for name in ("sparse", "dense"):
        gg, dd = copy.deepcopy(g), copy.deepcopy(d)
        if name == "dense":
            dd.SPARSE_MIN_POINTS = 10 ** 9
        tr = PointGANTrainer(gg, dd)
        dl, gp = tr.critic_step(uniform, z1, alpha)
        dgr = {k: p.grad.detach().clone() for k, p in dd.named_parameters()}
        gl = tr.generator_step(uniform, z2)
        ggr = {k: p.grad.detach().clone() for k, p in gg.named_parameters() if p.grad is not None}
        runs[name] = ([dl.item(), gp.item(), gl.item()], dgr, ggr)
    np.testing.assert_allclose(runs["sparse"][0], runs["dense"][0], rtol=2e-4, atol=1e-6)
    for which, what in ((1, "critic"), (2, "generator")):
        for k, ref in runs["dense"][which].items():
            close_mostly(runs["sparse"][which][k], ref, rtol=2e-4, max_bad_frac=2e-3, what="%s grad %s, sparse vs dense" % (what, k))
    # the critic sees a cloud through the selected points only
    x = uniform[:2].clone()
    idx = d.selected_points(x)
    keep = torch.zeros(2, P, dtype=torch.bool, device="cuda")
    keep.scatter_(1, idx, True)
    moved = x.clone()
    for b in range(2):                         # every other point becomes a copy of a selected one of ITS cloud: it can tie a maximum, not beat it
        moved[b][~keep[b]] = x[b, idx[b, 0]]
    with torch.no_grad():
        a, b = d(x[..., :3], x[..., 3:]), d(moved[..., :3], moved[..., 3:])
    close(a, b, rtol=1e-5, what="critic output after moving unselected points")
