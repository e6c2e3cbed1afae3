"""TEST INFRASTRUCTURE — generates tests/golden/*.npz from the REAL reference (authoring container only).

    python oracle/make_golden.py

The reference ships no tests or golden vectors; its only pinned artefacts are the pretrained SDFNet checkpoints
in examples/.  This script (1) reproduces the known answers of those checkpoints, (2) runs every hot-path module of
the reference (imported from /root/reference, CPU) under fixed seeds on small inputs and records outputs, losses and
gradient summaries, (3) runs short training trajectories with the reference's modules inside the restated step bodies
and records them — after asserting that oracle/torch_oracle.py reproduces each of them, which is what pins the oracle.
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, torch_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def summarize(t):
    a = t.detach().double().reshape(-1)
    return np.concatenate([[a.sum().item(), a.abs().sum().item()], a[:4].numpy(), a[-4:].numpy()])


def grads_summary(named):
    return {k: summarize(g) for k, g in named if g is not None}


def put(d, prefix, sub):
    for k, v in sub.items():
        d[prefix + "/" + k] = np.asarray(v)


def main():
    ref = ref_import.load()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)

    # ---- 1. pretrained SDFNet checkpoints: known answers ---------------------------------------
    ex = {}
    pts = torch.tensor(ref.get_voxel_coordinates(32))
    torch.manual_seed(1234)
    z = torch.randn(128)
    ex["z"] = z.numpy()
    for name in ("chairs", "airplanes", "sofas"):
        path = os.path.join(ref.root, "examples", "gan_generator_voxels_%s.to" % name)
        sd = torch.load(path, map_location="cpu")
        net = ref.SDFNet(device="cpu")
        missing = net.load_state_dict(sd, strict=False)
        assert not missing.missing_keys and not missing.unexpected_keys
        with torch.no_grad():
            out = net(pts, z.repeat(32768, 1))
            out_o = O.sdfnet_forward(O.clone_state(sd, requires_grad=False), pts, z.repeat(32768, 1))
        assert torch.equal(out, out_o), "oracle SDFNet differs from the reference"
        ex[name + "/sha256"] = np.frombuffer(hashlib.sha256(open(path, "rb").read()).digest(), dtype=np.uint8)
        ex[name + "/stats"] = np.array([out.mean().item(), out.min().item(), out.max().item(), (out < 0).sum().item(),
                                        out[0].item(), out[16912].item()])
        ex[name + "/out_head"] = out[:4096].numpy()
        ex[name + "/out_mid"] = out[16384:16384 + 4096].numpy()
        if name == "chairs":
            np.savez(os.path.join(OUT, "sdfnet_chairs_weights.npz"), **{k: v.numpy() for k, v in sd.items()})
    np.savez(os.path.join(OUT, "sdfnet_examples.npz"), **ex)

    # ---- 2. module-level fixtures --------------------------------------------------------------------------------
    mod = {}

    def run_module(tag, seed, build, make_inputs, fwd_ref, fwd_oracle, loss_of):
        torch.manual_seed(seed)
        m = build()
        m.train()
        sd0 = {k: v.clone() for k, v in m.state_dict().items()}
        put(mod, tag + "/init", {k: summarize(v.float()) for k, v in sd0.items()})
        inputs = make_inputs()
        out = fwd_ref(m, *inputs)
        loss = loss_of(out)
        loss.backward()
        P = O.clone_state(sd0)
        out_o = fwd_oracle(P, *inputs)
        loss_o = loss_of(out_o)
        loss_o.backward()
        first = out[0] if isinstance(out, tuple) else out
        first_o = out_o[0] if isinstance(out_o, tuple) else out_o
        assert torch.equal(first, first_o), tag + ": oracle forward differs from the reference"
        for (k, p) in m.named_parameters():
            if p.grad is None:
                assert P[k].grad is None or float(P[k].grad.abs().sum()) == 0.0, (tag, k)
                continue
            assert torch.allclose(p.grad, P[k].grad, rtol=1e-5, atol=1e-7), (tag, k)
        for i, t in enumerate(inputs):
            mod["%s/in%d" % (tag, i)] = t.detach().numpy()
        mod[tag + "/out"] = first.detach().numpy()
        mod[tag + "/loss"] = np.array(loss.item())
        put(mod, tag + "/grad", grads_summary((k, p.grad) for k, p in m.named_parameters()))
        after = m.state_dict()
        put(mod, tag + "/buffers_after", {k: v.double().numpy() for k, v in after.items()
                                          if "running_" in k or "num_batches" in k})
        return m

    wsum = lambda out: ((out[0] if isinstance(out, tuple) else out) * wts(out)).sum()  # noqa: E731

    def wts(out):
        t = out[0] if isinstance(out, tuple) else out
        g = torch.Generator().manual_seed(99)
        return torch.randn(t.shape, generator=g)

    run_module("generator", 11, ref.Generator, lambda: (torch.randn(3, 128),), lambda m, z: m(z),
               lambda P, z: O.generator_forward(P, z, True), wsum)

    def build_d():
        d = ref.Discriminator()
        d.use_sigmoid = False
        return d
    run_module("discriminator", 12, build_d, lambda: (torch.rand(3, 32, 32, 32) * 2 - 1,), lambda m, x: m(x),
               lambda P, x: O.discriminator_forward(P, x, False), wsum)
    run_module("discriminator_sigmoid", 13, ref.Discriminator, lambda: (torch.rand(2, 32, 32, 32) * 2 - 1,),
               lambda m, x: m(x), lambda P, x: O.discriminator_forward(P, x, True), wsum)
    run_module("autoencoder", 14, lambda: ref.Autoencoder(is_variational=False),
               lambda: (torch.rand(4, 32, 32, 32) * 2 - 1,), lambda m, x: m(x),
               lambda P, x: O.autoencoder_forward(P, x, True, False), wsum)

    for it, fade, bsz in ((0, 1.0, 3), (1, 0.4, 3), (2, 0.3, 2), (3, 1.0, 2), (3, 0.5, 2)):
        def build_p(it=it, fade=fade):
            d = ref.ProgressiveDiscriminator()
            d.set_iteration(it)
            d.fade_in_progress = fade
            return d
        r = O.RESOLUTIONS[it]
        run_module("progressive_it%d_fade%02d" % (it, int(fade * 10)), 20 + it, build_p,
                   lambda r=r, bsz=bsz: (torch.rand(bsz, r, r, r) * 2 - 1,), lambda m, x: m(x),
                   lambda P, x, it=it, fade=fade: O.progressive_forward(P, x, it, fade), wsum)

    for lat in (128, 256):
        run_module("sdfnet_L%d" % lat, 30, lambda lat=lat: ref.SDFNet(latent_code_size=lat, device="cpu"),
                   lambda lat=lat: (torch.rand(200, 3) * 2 - 1, torch.randn(200, lat) * 0.3),
                   lambda m, p, l: m(p, l), lambda P, p, l: O.sdfnet_forward(P, p, l), wsum)

    # VAE (eps injected: the reference draws it at autoencoder.py:79 from the global CPU RNG)
    torch.manual_seed(15)
    vae = ref.Autoencoder(is_variational=True)
    vae.train()
    sd0 = {k: v.clone() for k, v in vae.state_dict().items()}
    x = torch.rand(4, 32, 32, 32) * 2 - 1
    torch.manual_seed(1515)
    out, mean, logvar = vae(x)
    torch.manual_seed(1515)
    eps = torch.distributions.normal.Normal(0, 1).sample(mean.shape)
    P = O.clone_state(sd0)
    out_o, mean_o, logvar_o = O.autoencoder_forward(P, x, True, True, eps)
    assert torch.equal(out, out_o) and torch.equal(mean, mean_o)
    mod["vae/in0"], mod["vae/eps"] = x.numpy(), eps.numpy()
    mod["vae/out"], mod["vae/mean"], mod["vae/logvar"] = out.detach().numpy(), mean.detach().numpy(), logvar.detach().numpy()
    put(mod, "vae/init", {k: summarize(v.float()) for k, v in sd0.items()})

    # gradient penalty (train_hybrid_progressive_gan.py:102-111) on the reference ProgD, iteration 2, fade 0.3
    torch.manual_seed(41)
    d = ref.ProgressiveDiscriminator()
    d.set_iteration(2)
    d.fade_in_progress = 0.3
    sd0 = {k: v.clone() for k, v in d.state_dict().items()}
    real, fake = torch.rand(3, 32, 32, 32) * 2 - 1, torch.rand(3, 32, 32, 32) * 0.2 - 0.1
    alpha = torch.rand(3, 1, 1, 1)
    a = alpha.expand(real.shape)
    xi = (a * real + (1 - a) * fake).requires_grad_(True)
    o = d(xi)
    g = torch.autograd.grad(outputs=o, inputs=xi, grad_outputs=torch.ones(o.shape), create_graph=True,
                            retain_graph=True, only_inputs=True)[0]
    gp = ((g.norm(2, dim=(1, 2, 3)) - 1) ** 2).mean() * 10
    gp.backward()
    orc = O.HybridProgressiveGANOracle({"layers1.0.weight": torch.zeros(1)}, sd0, None, 2, 0.3)
    gp_o = orc.gradient_penalty(real, fake, alpha)
    gp_o.backward()
    assert torch.allclose(gp, gp_o, rtol=1e-6)
    for k, p in d.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, orc.D[k].grad, rtol=1e-5, atol=1e-7), k
    mod["gp/real"], mod["gp/fake"], mod["gp/alpha"] = real.numpy(), fake.numpy(), alpha.numpy()
    mod["gp/value"] = np.array(gp.item())
    mod["gp/dx"] = g.detach().numpy()
    put(mod, "gp/init", {k: summarize(v.float()) for k, v in sd0.items()})
    put(mod, "gp/grad", grads_summary((k, p.grad) for k, p in d.named_parameters()))
    np.savez_compressed(os.path.join(OUT, "modules.npz"), **mod)

    # ---- 3. short training trajectories (reference modules inside the restated steps == oracle) --------------------
    st = {}
    # WGAN (train_wgan.py): 2 critic steps + 1 generator step at B=4
    torch.manual_seed(51)
    G, C = ref.Generator(), ref.Discriminator()
    C.use_sigmoid = False
    g_sd, c_sd = {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in C.state_dict().items()}
    g_opt = torch.optim.RMSprop(G.parameters(), lr=0.00005)
    c_opt = torch.optim.RMSprop(C.parameters(), lr=0.00005)
    reals = [torch.rand(4, 32, 32, 32) * 2 - 1 for _ in range(2)]
    zs = [torch.randn(4, 128) for _ in range(2)]
    zg = torch.randn(4, 128)
    orc = O.WGANOracle(g_sd, c_sd)
    losses = []
    for i in range(2):
        G.zero_grad(); C.zero_grad()
        fake = G(zs[i]).detach()
        of, orr = C(fake), C(reals[i])
        loss = torch.mean(of) - torch.mean(orr)
        loss.backward(); c_opt.step(); C.clip_weights(0.01)
        lo = orc.critic_step(reals[i], zs[i])[0]
        assert torch.allclose(loss, lo, rtol=1e-5, atol=1e-7)
        losses.append(loss.item())
        if i == 0:
            G.zero_grad(); C.zero_grad()
            out = C(G(zg))
            gl = -torch.mean(out)
            gl.backward(); g_opt.step()
            glo = orc.generator_step(zg)[0]
            assert torch.allclose(gl, glo, rtol=1e-5, atol=1e-7)
            losses.append(gl.item())
    for k, v in G.state_dict().items():
        assert torch.allclose(v.float(), orc.G[k].float(), rtol=1e-4, atol=1e-6), k
    for k, v in C.state_dict().items():
        assert torch.allclose(v.float(), orc.C[k].float(), rtol=1e-4, atol=1e-6), k
    st["wgan/losses"] = np.array(losses)
    for i in range(2):
        st["wgan/real%d" % i], st["wgan/z%d" % i] = reals[i].numpy(), zs[i].numpy()
    st["wgan/zg"] = zg.numpy()
    put(st, "wgan/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "wgan/c_final", {k: summarize(v.float()) for k, v in C.state_dict().items()})

    # Autoencoder classic (train_autoencoder.py), config 1: B=4, 3 Adam steps
    torch.manual_seed(52)
    A = ref.Autoencoder(is_variational=False)
    a_sd = {k: v.clone() for k, v in A.state_dict().items()}
    opt = torch.optim.Adam(A.parameters(), lr=0.00005)
    orc = O.AutoencoderOracle(a_sd, False)
    batches = [(torch.rand(4, 32, 32, 32) * 0.4 - 0.2).clamp(-0.1, 0.1) / 0.1 for _ in range(3)]
    losses = []
    for b in batches:
        A.zero_grad(); A.train()
        out = A(b)
        rec = O.reconstruction_loss(out, b)
        rec.backward(); opt.step()
        ro = orc.step(b)[0]
        assert torch.allclose(rec, ro, rtol=1e-5, atol=1e-7)
        losses.append(rec.item())
    st["ae/losses"] = np.array(losses)
    for i, b in enumerate(batches):
        st["ae/batch%d" % i] = b.numpy()
    put(st, "ae/final", {k: summarize(v.float()) for k, v in A.state_dict().items()})

    # DeepSDF auto-decoder (train_sdf_autodecoder.py): 4 shapes x 500 points, 3 steps of 256 points
    torch.manual_seed(53)
    S = ref.SDFNet(device="cpu")
    s_sd = {k: v.clone() for k, v in S.state_dict().items()}
    pc = 500
    points = torch.rand(4 * pc, 3) * 2 - 1
    sdf = torch.rand(4 * pc) * 0.3 - 0.15
    lat0 = torch.distributions.normal.Normal(0, 0.0001).sample((4, 128))
    idxs = [torch.randint(0, 4 * pc, (256,)) for _ in range(3)]
    orc = O.SDFAutoDecoderOracle(s_sd, lat0, points, sdf, pointcloud_size=pc)
    lat = lat0.clone().requires_grad_(True)
    sdf_c = sdf.clamp(-0.1, 0.1)
    n_opt, l_opt = torch.optim.Adam(S.parameters(), lr=1e-5), torch.optim.Adam([lat], lr=1e-5)
    losses = []
    for idx in idxs:
        mi = idx // pc
        bl, bp, bs = lat[mi, :], points[idx, :], sdf_c[idx]
        S.zero_grad()
        if lat.grad is not None:
            lat.grad.data.zero_()
        out = S.forward(bp, bl)
        loss = torch.mean(torch.abs(out - bs)) + 0.01 * torch.mean(torch.pow(bl, 2))
        loss.backward(); n_opt.step(); l_opt.step()
        lo = orc.step(idx)
        assert torch.allclose(loss, lo, rtol=1e-5, atol=1e-8)
        losses.append(loss.item())
    st["sdf/losses"] = np.array(losses)
    st["sdf/points"], st["sdf/sdf"], st["sdf/lat0"] = points.numpy(), sdf.numpy(), lat0.numpy()
    for i, idx in enumerate(idxs):
        st["sdf/idx%d" % i] = idx.numpy()
    st["sdf/lat_final"] = lat.detach().numpy()
    put(st, "sdf/final", {k: summarize(v.float()) for k, v in S.state_dict().items()})

    # Hybrid WGAN (train_hybrid_wgan.py) at 8^3... the critic needs 32^3; B=2, 1 critic + 1 generator step
    torch.manual_seed(54)
    G, C = ref.SDFNet(device="cpu"), ref.Discriminator()
    C.use_sigmoid = False
    g_sd, c_sd = {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in C.state_dict().items()}
    grid = torch.tensor(ref.get_voxel_coordinates(32))
    orc = O.HybridWGANOracle(g_sd, c_sd, grid)
    g_opt = torch.optim.Adam(G.parameters(), lr=0.00001)
    c_opt = torch.optim.RMSprop(C.parameters(), lr=0.00001)
    real = torch.rand(2, 32, 32, 32) * 0.2 - 0.1
    z1, z2 = torch.randn(2, 128), torch.randn(2, 128)
    pts = grid.repeat((2, 1))
    c_opt.zero_grad()
    fake = G(pts, O.tile_latents(z1, 32768)).reshape(-1, 32, 32, 32)
    cl = torch.mean(C(fake)) - torch.mean(C(real))
    cl.backward(); c_opt.step(); C.clip_weights(0.01)
    clo = orc.critic_step(real, z1)[0]
    assert torch.allclose(cl, clo, rtol=1e-5, atol=1e-7)
    g_opt.zero_grad(); C.zero_grad()
    fake = G(pts, O.tile_latents(z2, 32768)).reshape(-1, 32, 32, 32)
    gl = torch.mean(-C(fake))
    gl.backward(); g_opt.step()
    glo = orc.generator_step(z2)[0]
    assert torch.allclose(gl, glo, rtol=1e-5, atol=1e-7)
    st["hybrid/losses"] = np.array([cl.item(), gl.item()])
    st["hybrid/real"], st["hybrid/z1"], st["hybrid/z2"] = real.numpy(), z1.numpy(), z2.numpy()
    put(st, "hybrid/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "hybrid/c_final", {k: summarize(v.float()) for k, v in C.state_dict().items()})

    # Hybrid progressive GAN (train_hybrid_progressive_gan.py) iteration 1 (16^3), fade 0.6, B=2: G step + D step
    torch.manual_seed(55)
    G, D = ref.SDFNet(device="cpu"), ref.ProgressiveDiscriminator()
    D.set_iteration(1)
    D.fade_in_progress = 0.6
    g_sd, d_sd = {k: v.clone() for k, v in G.state_dict().items()}, {k: v.clone() for k, v in D.state_dict().items()}
    grid = torch.tensor(ref.get_voxel_coordinates(16))
    orc = O.HybridProgressiveGANOracle(g_sd, d_sd, grid, 1, 0.6)
    g_opt = torch.optim.RMSprop(G.parameters(), lr=0.0001)
    d_opt = torch.optim.RMSprop(D.parameters(), lr=0.0001)
    real = torch.rand(2, 16, 16, 16) * 2 - 1
    z1, z2 = torch.randn(2, 128), torch.randn(2, 128)
    alpha = torch.rand(2, 1, 1, 1)
    pts = grid.repeat((2, 1))
    g_opt.zero_grad()
    fake = G(pts, O.tile_latents(z1, 4096)).reshape(-1, 16, 16, 16)
    gl = -D(fake).mean()
    gl.backward(); g_opt.step()
    glo = orc.generator_step(z1)
    assert torch.allclose(gl, glo, rtol=1e-5, atol=1e-7)
    d_opt.zero_grad()
    fake = G(pts, O.tile_latents(z2, 4096)).reshape(-1, 16, 16, 16)
    of, orr = D(fake), D(real)
    a = alpha.expand(real.shape)
    xi = (a * real.detach() + (1 - a) * fake.detach()).requires_grad_(True)
    o = D(xi)
    g = torch.autograd.grad(outputs=o, inputs=xi, grad_outputs=torch.ones(o.shape), create_graph=True, retain_graph=True,
                            only_inputs=True)[0]
    gp = ((g.norm(2, dim=(1, 2, 3)) - 1) ** 2).mean() * 10
    dl = of.mean() - orr.mean() + gp
    dl.backward(); d_opt.step()
    dlo, gpo = orc.discriminator_step(real, z2, alpha)
    assert torch.allclose(dl, dlo, rtol=1e-5, atol=1e-7) and torch.allclose(gp, gpo, rtol=1e-5)
    st["prog/losses"] = np.array([gl.item(), dl.item(), gp.item()])
    st["prog/real"], st["prog/z1"], st["prog/z2"], st["prog/alpha"] = real.numpy(), z1.numpy(), z2.numpy(), alpha.numpy()
    put(st, "prog/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "prog/d_final", {k: summarize(v.float()) for k, v in D.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, "steps.npz"), **st)
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        print("  %-32s %8d bytes" % (f, os.path.getsize(os.path.join(OUT, f))))


if __name__ == "__main__":
    main()
