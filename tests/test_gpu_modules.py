"""GPU tier: the drop-in modules and training steps against the golden fixtures generated from the real reference
(tests/golden/, oracle/make_golden.py) and against the CPU oracle on the same seeded inputs."""
import numpy as np
import pytest
import torch

from conftest import assert_summary_close
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4


def close(a, b, rtol=RTOL, atol=None, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if atol is None:
        atol = rtol * max(float(b.abs().mean()), 1e-30)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: what + ": " + m)


def _wts(shape):
    return torch.randn(shape, generator=torch.Generator().manual_seed(99))


def _check_module(golden, tag, seed, build, forward, grad_rtol=RTOL):
    torch.manual_seed(seed)
    m = build()
    m.train()
    ins = [golden.t("%s/in%d" % (tag, i)).cuda() for i in range(3) if ("%s/in%d" % (tag, i)) in golden.z.files]
    out = forward(m, *ins)
    first = out[0] if isinstance(out, tuple) else out
    close(first, golden.t(tag + "/out"), what=tag + " forward")
    loss = (first * _wts(first.shape).cuda()).sum()
    np.testing.assert_allclose(loss.item(), golden[tag + "/loss"], rtol=RTOL,
                               atol=RTOL * float(np.abs(golden[tag + "/out"]).sum()) * 0.05)
    loss.backward()
    grads = golden.sub(tag + "/grad")
    for k, p in m.named_parameters():
        if k in grads:
            assert p.grad is not None, k
            assert_summary_close(p.grad, grads[k], grad_rtol, 1e-9, tag + " grad " + k)
        else:
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
    for k, ref in golden.sub(tag + "/buffers_after").items():
        close(m.state_dict()[k].double(), torch.from_numpy(ref), rtol=1e-5, what=tag + " buffer " + k)
    return m


def test_generator(golden_modules):
    from shapegan_amd.model.gan import Generator
    _check_module(golden_modules, "generator", 11, Generator, lambda m, z: m(z))


def test_discriminator(golden_modules):
    from shapegan_amd.model.gan import Discriminator

    def build():
        d = Discriminator()
        d.use_sigmoid = False
        return d
    _check_module(golden_modules, "discriminator", 12, build, lambda m, x: m(x))
    _check_module(golden_modules, "discriminator_sigmoid", 13, Discriminator, lambda m, x: m(x))


def test_autoencoder_classic(golden_modules):
    """BASELINE config 1's network (classic AE, batch 4).  The 256-channel BN layers normalise 4 values per channel,
    which amplifies fp32 summation-order noise: gradients are compared at 1e-3."""
    from shapegan_amd.model.autoencoder import Autoencoder
    _check_module(golden_modules, "autoencoder", 14, lambda: Autoencoder(is_variational=False), lambda m, x: m(x),
                  grad_rtol=1e-3)


def test_vae_forward(golden_modules, monkeypatch):
    from shapegan_amd.model import autoencoder as ae_mod
    torch.manual_seed(15)
    vae = ae_mod.Autoencoder(is_variational=True)
    vae.train()
    eps = golden_modules.t("vae/eps")

    class FixedNormal(object):
        def sample(self, shape):
            return eps
    monkeypatch.setattr(ae_mod, "standard_normal_distribution", FixedNormal())
    out, mean, logvar = vae(golden_modules.t("vae/in0").cuda())
    close(mean, golden_modules.t("vae/mean"), rtol=1e-3, what="vae mean")
    close(logvar, golden_modules.t("vae/logvar"), rtol=1e-3, what="vae logvar")
    close(out, golden_modules.t("vae/out"), rtol=1e-3, what="vae out")


@pytest.mark.parametrize("it,fade", [(0, 1.0), (1, 0.4), (2, 0.3), (3, 1.0), (3, 0.5)])
def test_progressive_discriminator(golden_modules, it, fade):
    from shapegan_amd.model.progressive_gan import Discriminator

    def build():
        d = Discriminator()
        d.set_iteration(it)
        d.fade_in_progress = fade
        return d.cuda()
    _check_module(golden_modules, "progressive_it%d_fade%02d" % (it, int(fade * 10)), 20 + it, build, lambda m, x: m(x))


@pytest.mark.parametrize("latent", [128, 256])
def test_sdfnet_module(golden_modules, latent):
    from shapegan_amd.model.sdf_net import SDFNet
    _check_module(golden_modules, "sdfnet_L%d" % latent, 30, lambda: SDFNet(latent_code_size=latent),
                  lambda m, p, l: m(p, l))


def test_gradient_penalty_double_backward(golden_modules):
    """WGAN-GP through the HIP kernels: autograd.grad(create_graph=True) then backward
    (train_hybrid_progressive_gan.py:102-111) vs the reference's value, input gradient and parameter gradients."""
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridProgressiveGANTrainer
    torch.manual_seed(41)
    d = Discriminator().cuda()
    d.set_iteration(2)
    d.fade_in_progress = 0.3
    tr = HybridProgressiveGANTrainer(SDFNet(), d, None, 32)
    real, fake, alpha = (golden_modules.t("gp/" + k).cuda() for k in ("real", "fake", "alpha"))
    a = alpha.expand(real.shape)
    xi = (a * real + (1 - a) * fake).requires_grad_(True)
    out = d(xi)
    (g,) = torch.autograd.grad(outputs=out, inputs=xi, grad_outputs=torch.ones_like(out), create_graph=True)
    close(g, golden_modules.t("gp/dx"), what="dD/dx")
    tr.d_opt.zero_grad()
    gp = tr.gradient_penalty(real, fake, alpha)
    np.testing.assert_allclose(gp.item(), golden_modules["gp/value"], rtol=RTOL)
    gp.backward()
    grads = golden_modules.sub("gp/grad")
    for k, p in d.named_parameters():
        if k in grads:
            assert_summary_close(p.grad, grads[k], 2e-4, 1e-9, "gp grad " + k)
    assert float(d.optional_layers[3][0].weight.grad.abs().sum()) == 0.0      # unused stage untouched


def _check_final(module, golden, prefix, rtol=1e-4):
    for k, ref in golden.sub(prefix).items():
        assert_summary_close(module.state_dict()[k].float(), ref, rtol, 1e-8, prefix + " " + k)


def test_wgan_trajectory(golden_steps):
    """train_wgan.py steps (2 critic + 1 generator) from the reference's seed-51 init."""
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    torch.manual_seed(51)
    tr = WGANTrainer(Generator(), Discriminator())
    losses = []
    for i in range(2):
        losses.append(tr.critic_step(golden_steps.t("wgan/real%d" % i).cuda(), golden_steps.t("wgan/z%d" % i).cuda())[0].item())
        if i == 0:
            losses.append(tr.generator_step(golden_steps.t("wgan/zg").cuda())[0].item())
    np.testing.assert_allclose(losses, golden_steps["wgan/losses"], rtol=1e-4, atol=1e-6)
    _check_final(tr.critic, golden_steps, "wgan/c_final")
    _check_final(tr.generator, golden_steps, "wgan/g_final")


def test_autoencoder_trajectory(golden_steps):
    """BASELINE config 1: train_autoencoder.py classic, batch 4, three Adam steps."""
    from shapegan_amd.model.autoencoder import Autoencoder
    from shapegan_amd.train_steps import AutoencoderTrainer
    torch.manual_seed(52)
    tr = AutoencoderTrainer(Autoencoder(is_variational=False))
    losses = [tr.step(golden_steps.t("ae/batch%d" % i).cuda())[0].item() for i in range(3)]
    np.testing.assert_allclose(losses, golden_steps["ae/losses"], rtol=1e-3)
    _check_final(tr.autoencoder, golden_steps, "ae/final", rtol=1e-3)


def test_sdf_autodecoder_trajectory(golden_steps):
    """train_sdf_autodecoder.py: three steps incl. the latent-table gather / scatter-add and both Adam updates."""
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    torch.manual_seed(53)
    net = SDFNet()
    lat = golden_steps.t("sdf/lat0").cuda()
    tr = SDFAutoDecoderTrainer(net, lat, golden_steps.t("sdf/points").cuda(), golden_steps.t("sdf/sdf").cuda(),
                               pointcloud_size=500)
    losses = [tr.step(golden_steps.t("sdf/idx%d" % i).cuda()).item() for i in range(3)]
    np.testing.assert_allclose(losses, golden_steps["sdf/losses"], rtol=1e-4)
    _check_final(net, golden_steps, "sdf/final")
    close(tr.latent_codes, golden_steps.t("sdf/lat_final"), rtol=1e-4, atol=1e-7, what="latent table")


def test_hybrid_wgan_trajectory(golden_steps):
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridWGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(54)
    g, c = SDFNet(), Discriminator()
    tr = HybridWGANTrainer(g, c, torch.tensor(get_voxel_coordinates(32)).cuda())
    cl = tr.critic_step(golden_steps.t("hybrid/real").cuda(), golden_steps.t("hybrid/z1").cuda())[0].item()
    gl = tr.generator_step(golden_steps.t("hybrid/z2").cuda())[0].item()
    np.testing.assert_allclose([cl, gl], golden_steps["hybrid/losses"], rtol=1e-4, atol=1e-6)
    _check_final(c, golden_steps, "hybrid/c_final")
    _check_final(g, golden_steps, "hybrid/g_final")


def test_hybrid_progressive_trajectory(golden_steps):
    """Generator step + discriminator step with gradient penalty at iteration 1 (16^3), fade-in 0.6."""
    from shapegan_amd.model.progressive_gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import HybridProgressiveGANTrainer
    from shapegan_amd.util import get_voxel_coordinates
    torch.manual_seed(55)
    g, d = SDFNet(), Discriminator().cuda()
    d.set_iteration(1)
    d.fade_in_progress = 0.6
    tr = HybridProgressiveGANTrainer(g, d, torch.tensor(get_voxel_coordinates(16)).cuda(), 16)
    gl = tr.generator_step(golden_steps.t("prog/z1").cuda()).item()
    dl, gp = tr.discriminator_step(golden_steps.t("prog/real").cuda(), golden_steps.t("prog/z2").cuda(),
                                   golden_steps.t("prog/alpha").cuda())
    np.testing.assert_allclose([gl, dl.item(), gp.item()], golden_steps["prog/losses"], rtol=2e-4, atol=1e-6)
    _check_final(d, golden_steps, "prog/d_final")
    _check_final(g, golden_steps, "prog/g_final")


def test_dp_shards_sum_to_full_batch_gradient():
    """Distributed math on one GPU: averaged shard gradients == full-batch gradient for the BN-free
    critic, i.e. what one RCCL all-reduce of the flat buffers + grad_scale 1/G produces."""
    from shapegan_amd.model.gan import Discriminator
    torch.manual_seed(3)
    d = Discriminator()
    d.use_sigmoid = False
    x = torch.rand(8, 32, 32, 32, device="cuda") * 2 - 1
    d(x).mean().backward()
    full = [p.grad.clone() for p in d.parameters()]
    d.zero_grad()
    for s in range(4):
        (d(x[2 * s:2 * s + 2]).mean() / 4).backward()
    for f, p in zip(full, d.parameters()):
        close(p.grad, f, rtol=2e-4)
