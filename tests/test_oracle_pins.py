"""CPU tier: pins the oracle (oracle/) against the golden fixtures and, when /root/reference is present, against
the real reference classes.  Nothing here touches shapegan_amd's product path."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_summary_close, summarize
from oracle import c_oracle, ref_import
from oracle import torch_oracle as O

have_ref = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present (GPU box)")


def test_sdfnet_known_answers_chairs(golden_sdf, chairs_state):
    """examples/gan_generator_voxels_chairs.to known answers: the only pinned artefact the
    reference ships."""
    from shapegan_amd.util import get_voxel_coordinates
    pts = torch.tensor(get_voxel_coordinates(32))
    z = torch.from_numpy(golden_sdf["z"])
    with torch.no_grad():
        out = O.sdfnet_forward(chairs_state, pts, z.repeat(32768, 1))
    stats = golden_sdf["chairs/stats"]
    assert abs(out.mean().item() - 0.072827) < 2e-6 and abs(stats[0] - 0.072827) < 2e-6
    assert abs(out.min().item() - (-0.137464)) < 2e-6 and abs(out.max().item() - 0.120349) < 2e-6
    assert int((out < 0).sum()) == 4187 == int(stats[3])
    assert abs(out[0].item() - 0.098810) < 2e-6 and abs(out[16912].item() - 0.086498) < 2e-6
    np.testing.assert_allclose(out[:4096].numpy(), golden_sdf["chairs/out_head"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(out[16384:16384 + 4096].numpy(), golden_sdf["chairs/out_mid"], rtol=1e-5, atol=1e-6)


def test_sdfnet_known_answers_table(golden_sdf):
    want = {"chairs": (0.072827, -0.137464, 0.120349, 4187, 0.098810, 0.086498),
            "airplanes": (0.097593, -0.068043, 0.107420, 179, 0.101230, 0.036896),
            "sofas": (0.073069, -0.106731, 0.112226, 4252, 0.100641, -0.080480)}
    for name, w in want.items():
        np.testing.assert_allclose(golden_sdf[name + "/stats"], np.array(w), rtol=0, atol=2e-6)


def test_c_oracle_sdfnet_matches_torch_oracle(chairs_state):
    torch.manual_seed(3)
    pts = torch.rand(300, 3) * 2 - 1
    lat = torch.randn(300, 128) * 0.5
    keys = ["layers%d.%d.%s" % (s, i, n) for s in (1, 2) for i in (0, 2, 4, 6) for n in ("weight", "bias")]
    out_c = c_oracle.sdfnet_fwd(pts.numpy(), lat.numpy(), [chairs_state[k].numpy() for k in keys])
    with torch.no_grad():
        out_t = O.sdfnet_forward(chairs_state, pts, lat).numpy()
    np.testing.assert_allclose(out_c, out_t, rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("N,Ci,Co,R", [(2, 3, 5, 8), (1, 1, 4, 6), (2, 8, 1, 4), (1, 2, 2, 2)])
def test_c_oracle_conv_matches_aten(N, Ci, Co, R):
    """Plain-C defining sums == the ATen ops the reference's nn.Conv3d / nn.ConvTranspose3d call."""
    torch.manual_seed(N * 100 + Ci * 10 + Co)
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Co, Ci, 4, 4, 4) * 0.2
    b = torch.randn(Co)
    y = F.conv3d(x, w, b, stride=2, padding=1)
    np.testing.assert_allclose(c_oracle.conv_fwd(x.numpy(), w.numpy(), b.numpy()), y.numpy(), rtol=1e-5, atol=1e-5)
    dy = torch.randn_like(y)
    xg = x.clone().requires_grad_(True)
    wg = w.clone().requires_grad_(True)
    F.conv3d(xg, wg, b, stride=2, padding=1).backward(dy)
    np.testing.assert_allclose(c_oracle.conv_dgrad(dy.numpy(), w.numpy()), xg.grad.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(c_oracle.conv_wgrad(dy.numpy(), x.numpy()), wg.grad.numpy(), rtol=1e-4, atol=1e-4)
    # ConvTranspose3d forward == dgrad form with weight [Cin_T=Co, Cout_T=Ci] and bias over Cout_T
    bt = torch.randn(Ci)
    yt = F.conv_transpose3d(dy, w, bt, stride=2, padding=1)
    np.testing.assert_allclose(c_oracle.conv_dgrad(dy.numpy(), w.numpy(), bt.numpy()), yt.numpy(), rtol=1e-5, atol=1e-5)


def test_c_oracle_bn_stats():
    torch.manual_seed(5)
    x = torch.randn(3, 4, 5, 5, 5) * 2 + 7
    m, v = c_oracle.bn_stats(x.numpy())
    np.testing.assert_allclose(m, x.double().mean(dim=(0, 2, 3, 4)).numpy(), rtol=1e-9)
    np.testing.assert_allclose(v, x.double().var(dim=(0, 2, 3, 4), unbiased=False).numpy(), rtol=1e-9)


def _module_cases():
    return ["generator", "discriminator", "discriminator_sigmoid", "autoencoder", "progressive_it0_fade10",
            "progressive_it1_fade04", "progressive_it2_fade03", "progressive_it3_fade10", "progressive_it3_fade05",
            "sdfnet_L128", "sdfnet_L256"]


@have_ref
@pytest.mark.parametrize("tag", _module_cases())
def test_oracle_reproduces_reference_modules(tag, golden_modules):
    """Rebuilds the reference module under the fixture's seed, checks the init summary, then checks that the
    oracle's functional forward equals the fixture output (make_golden.py asserted equality with the reference)."""
    ref = ref_import.load()
    seeds = {"generator": 11, "discriminator": 12, "discriminator_sigmoid": 13, "autoencoder": 14, "sdfnet_L128": 30,
             "sdfnet_L256": 30}
    if tag.startswith("progressive"):
        it = int(tag[len("progressive_it")])
        fade = int(tag[-2:]) / 10.0
        torch.manual_seed(20 + it)
        m = ref.ProgressiveDiscriminator()
        fwd = lambda P, x: O.progressive_forward(P, x, it, fade)  # noqa: E731
    else:
        torch.manual_seed(seeds[tag])
        if tag == "generator":
            m, fwd = ref.Generator(), lambda P, z: O.generator_forward(P, z, True)
        elif tag == "discriminator":
            m, fwd = ref.Discriminator(), lambda P, x: O.discriminator_forward(P, x, False)
        elif tag == "discriminator_sigmoid":
            m, fwd = ref.Discriminator(), lambda P, x: O.discriminator_forward(P, x, True)
        elif tag == "autoencoder":
            m, fwd = ref.Autoencoder(is_variational=False), lambda P, x: O.autoencoder_forward(P, x, True, False)
        else:
            lat = int(tag.split("L")[1])
            m, fwd = ref.SDFNet(latent_code_size=lat, device="cpu"), lambda P, p, l: O.sdfnet_forward(P, p, l)
    init = golden_modules.sub(tag + "/init")
    sd = m.state_dict()
    assert set(init) == set(sd)
    for k, v in sd.items():
        np.testing.assert_array_equal(summarize(v.float()), init[k], err_msg=k)
    ins = [golden_modules.t("%s/in%d" % (tag, i)) for i in range(3) if ("%s/in%d" % (tag, i)) in golden_modules.z.files]
    P = O.clone_state(sd)
    out = fwd(P, *ins)
    np.testing.assert_array_equal(out.detach().numpy(), golden_modules[tag + "/out"])


def test_oracle_steps_match_golden_wgan(golden_steps, golden_modules):
    """The WGAN trajectory stored in steps.npz is reproduced by the oracle from the same seed-derived init.
    (Init comes from torch's own constructors, which the shapegan_amd shells share — see test_host_logic.)"""
    from shapegan_amd.model.gan import Discriminator, Generator
    torch.manual_seed(51)
    G, C = Generator(), Discriminator()
    orc = O.WGANOracle(G.state_dict(), C.state_dict())
    reals = [golden_steps.t("wgan/real%d" % i) for i in range(2)]
    zs = [golden_steps.t("wgan/z%d" % i) for i in range(2)]
    losses = []
    for i in range(2):
        losses.append(orc.critic_step(reals[i], zs[i])[0].item())
        if i == 0:
            losses.append(orc.generator_step(golden_steps.t("wgan/zg"))[0].item())
    np.testing.assert_allclose(losses, golden_steps["wgan/losses"], rtol=1e-5, atol=1e-7)
    for k, ref in golden_steps.sub("wgan/c_final").items():
        assert_summary_close(orc.C[k].float(), ref, 1e-5, 1e-8, k)
    for k, ref in golden_steps.sub("wgan/g_final").items():
        assert_summary_close(orc.G[k].float(), ref, 1e-5, 1e-8, k)


def test_oracle_steps_match_golden_classic_gan(golden_steps_f2):
    """train_gan.py's three updates per batch, reproduced from the seed-derived init."""
    from shapegan_amd.model.gan import Discriminator, Generator
    g = golden_steps_f2
    torch.manual_seed(61)
    G, D = Generator(), Discriminator()
    orc = O.ClassicGANOracle(G.state_dict(), D.state_dict())
    losses = [orc.generator_step(g.t("gan/zg")).item()]
    fl, of = orc.discriminator_fake_step(g.t("gan/zd"))
    vl, ov = orc.discriminator_real_step(g.t("gan/real"))
    np.testing.assert_allclose(losses + [fl.item(), vl.item()], g["gan/losses"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(of.numpy(), g["gan/out_fake"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ov.numpy(), g["gan/out_real"], rtol=1e-5, atol=1e-7)
    for k, ref in g.sub("gan/d_final").items():
        assert_summary_close(orc.D[k].float(), ref, 1e-5, 1e-8, k)
    for k, ref in g.sub("gan/g_final").items():
        assert_summary_close(orc.G[k].float(), ref, 1e-5, 1e-8, k)


def test_oracle_steps_match_golden_hybrid_gan(golden_steps_f2):
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.util import get_voxel_coordinates
    g = golden_steps_f2
    torch.manual_seed(62)
    G, D = SDFNet(device="cpu"), Discriminator()
    orc = O.HybridGANOracle(G.state_dict(), D.state_dict(), torch.tensor(get_voxel_coordinates(32)))
    losses = [orc.generator_step(g.t("hgan/zg")).item(), orc.discriminator_fake_step(g.t("hgan/zd"))[0].item(),
              orc.discriminator_real_step(g.t("hgan/real"))[0].item()]
    np.testing.assert_allclose(losses, g["hgan/losses"], rtol=1e-5, atol=1e-7)
    for k, ref in g.sub("hgan/g_final").items():
        assert_summary_close(orc.G[k].float(), ref, 1e-5, 1e-8, k)


def test_oracle_steps_match_golden_vae(golden_steps_f2):
    from shapegan_amd.model.autoencoder import Autoencoder
    g = golden_steps_f2
    torch.manual_seed(63)
    A = Autoencoder(is_variational=True)
    orc = O.AutoencoderOracle(A.state_dict(), True)
    recs = [orc.step(g.t("vae/batch%d" % i), g.t("vae/eps%d" % i))[0].item() for i in range(2)]
    np.testing.assert_allclose(recs, g["vae/losses"][0::2], rtol=1e-5, atol=1e-7)
    for k, ref in g.sub("vae/final").items():
        assert_summary_close(orc.P[k].float(), ref, 1e-5, 1e-8, k)


def test_oracle_point_gan_matches_golden(golden_steps_f4):
    """PointNet / SDFGenerator forwards and the train_point_gan.py steps, from seed-derived init."""
    from shapegan_amd.model.point_sdf_net import PointNet, SDFGenerator
    g = golden_steps_f4
    torch.manual_seed(81)
    G = SDFGenerator(128, 256, 8, True, dropout=0.0)
    out = O.sdf_generator_forward(O.clone_state(G.state_dict()), g.t("gen/pos"), g.t("gen/z"))
    np.testing.assert_array_equal(out.detach().numpy(), g["gen/out"])
    torch.manual_seed(82)
    D = PointNet(out_channels=1)
    out = O.pointnet_forward(O.clone_state(D.state_dict()), g.t("disc/pos"), g.t("disc/dist"))
    np.testing.assert_array_equal(out.detach().numpy(), g["disc/out"])
    torch.manual_seed(83)
    G, D = SDFGenerator(128, 256, 8, True, dropout=0.0), PointNet(out_channels=1)
    orc = O.PointGANOracle(G.state_dict(), D.state_dict())
    dl, gp = orc.critic_step(g.t("step/uniform"), g.t("step/z1"), g.t("step/alpha"))
    gl = orc.generator_step(g.t("step/uniform"), g.t("step/z2"))
    np.testing.assert_allclose([dl.item(), gp.item(), gl.item()], g["step/losses"], rtol=1e-5, atol=1e-7)
    for k, ref in g.sub("step/d_final").items():
        assert_summary_close(orc.D[k].float(), ref, 1e-5, 1e-8, k)
    for k, ref in g.sub("step/g_final").items():
        assert_summary_close(orc.G[k].float(), ref, 1e-5, 1e-8, k)


def test_oracle_gradient_penalty_golden(golden_modules):
    from shapegan_amd.model.progressive_gan import Discriminator
    torch.manual_seed(41)
    d = Discriminator()
    orc = O.HybridProgressiveGANOracle({"w": torch.zeros(1)}, d.state_dict(), None, 2, 0.3)
    gp = orc.gradient_penalty(golden_modules.t("gp/real"), golden_modules.t("gp/fake"), golden_modules.t("gp/alpha"))
    np.testing.assert_allclose(gp.item(), golden_modules["gp/value"], rtol=1e-6)
    gp.backward()
    for k, ref in golden_modules.sub("gp/grad").items():
        assert_summary_close(orc.D[k].grad, ref, 1e-4, 1e-9, k)
    # stages above the active iteration never receive a gradient (RMSprop must skip them)
    assert orc.D["optional_layers.3.0.weight"].grad is None
