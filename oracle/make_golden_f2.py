"""Fixtures for SURVEY.md 8f rank 2 (classic GAN / hybrid GAN / VAE training steps): tests/golden/steps_f2.npz.

Run in THIS container (needs /root/reference).  As in make_golden.py, every step is executed with the REAL reference
modules inside the restated loop body and asserted equal to oracle/torch_oracle.py before anything is written.
TEST INFRASTRUCTURE ONLY."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import, torch_oracle as O            # noqa: E402
from oracle.make_golden import summarize, put               # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def state(m):
    return {k: v.clone() for k, v in m.state_dict().items()}


def assert_state(m, P, what):
    for k, v in m.state_dict().items():
        assert torch.allclose(v.float(), P[k].float(), rtol=1e-4, atol=1e-6), (what, k)


def main():
    ref = ref_import.load()
    st = {}

    # ---- classic GAN (train_gan.py:57-86), B=4: generator update, discriminator on fakes, discriminator on reals ----
    torch.manual_seed(61)
    G, D = ref.Generator(), ref.Discriminator()          # use_sigmoid stays True
    orc = O.ClassicGANOracle(state(G), state(D))
    g_opt = torch.optim.Adam(G.parameters(), lr=0.001)
    d_opt = torch.optim.Adam(D.parameters(), lr=0.00001)
    real = torch.rand(4, 32, 32, 32) * 2 - 1
    zg, zd = torch.randn(4, 128), torch.randn(4, 128)
    g_opt.zero_grad()
    gl = -torch.mean(torch.log(D(G(zg))))
    gl.backward(); g_opt.step()
    assert torch.allclose(gl, orc.generator_step(zg), rtol=1e-5, atol=1e-7)
    d_opt.zero_grad()
    of = D(G(zd).detach())
    fl = F.binary_cross_entropy(of, torch.zeros(4))
    fl.backward(); d_opt.step()
    assert torch.allclose(fl, orc.discriminator_fake_step(zd)[0], rtol=1e-5, atol=1e-7)
    d_opt.zero_grad()
    ov = D(real)
    vl = F.binary_cross_entropy(ov, torch.ones(4))
    vl.backward(); d_opt.step()
    assert torch.allclose(vl, orc.discriminator_real_step(real)[0], rtol=1e-5, atol=1e-7)
    assert_state(G, orc.G, "gan G"); assert_state(D, orc.D, "gan D")
    st["gan/losses"] = np.array([gl.item(), fl.item(), vl.item()])
    st["gan/real"], st["gan/zg"], st["gan/zd"] = real.numpy(), zg.numpy(), zd.numpy()
    st["gan/out_fake"], st["gan/out_real"] = of.detach().numpy(), ov.detach().numpy()
    put(st, "gan/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "gan/d_final", {k: summarize(v.float()) for k, v in D.state_dict().items()})

    # ---- hybrid GAN (train_hybrid_gan.py:77-116), B=2 ----
    torch.manual_seed(62)
    G, D = ref.SDFNet(device="cpu"), ref.Discriminator()
    grid = torch.tensor(ref.get_voxel_coordinates(32))
    orc = O.HybridGANOracle(state(G), state(D), grid)
    g_opt = torch.optim.Adam(G.parameters(), lr=0.001)
    d_opt = torch.optim.Adam(D.parameters(), lr=0.00001)
    real = torch.rand(2, 32, 32, 32) * 0.2 - 0.1
    zg, zd = torch.randn(2, 128), torch.randn(2, 128)
    pts = grid.repeat((2, 1))
    g_opt.zero_grad()
    fake = G(pts, O.tile_latents(zg, 32768)).reshape(-1, 32, 32, 32)
    gl = torch.mean(-torch.log(D(fake)))
    gl.backward(); g_opt.step()
    assert torch.allclose(gl, orc.generator_step(zg), rtol=1e-5, atol=1e-7)
    d_opt.zero_grad()
    fake = G(pts, O.tile_latents(zd, 32768)).reshape(-1, 32, 32, 32)
    of = D(fake)
    fl = F.binary_cross_entropy(of, torch.zeros(2))
    fl.backward(); d_opt.step()
    assert torch.allclose(fl, orc.discriminator_fake_step(zd)[0], rtol=1e-5, atol=1e-7)
    d_opt.zero_grad()
    ov = D(real)
    vl = F.binary_cross_entropy(ov, torch.ones(2))
    vl.backward(); d_opt.step()
    assert torch.allclose(vl, orc.discriminator_real_step(real)[0], rtol=1e-5, atol=1e-7)
    assert_state(G, orc.G, "hybrid gan G"); assert_state(D, orc.D, "hybrid gan D")
    st["hgan/losses"] = np.array([gl.item(), fl.item(), vl.item()])
    st["hgan/real"], st["hgan/zg"], st["hgan/zd"] = real.numpy(), zg.numpy(), zd.numpy()
    put(st, "hgan/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "hgan/d_final", {k: summarize(v.float()) for k, v in D.state_dict().items()})

    # ---- VAE branch of train_autoencoder.py (:98-117 with :54-55), B=4, two Adam steps; eps drawn as the reference
    # does (autoencoder.py:79, global CPU RNG) and recorded ----
    torch.manual_seed(63)
    A = ref.Autoencoder(is_variational=True)
    orc = O.AutoencoderOracle(state(A), True)
    opt = torch.optim.Adam(A.parameters(), lr=0.00005)
    batches = [(torch.rand(4, 32, 32, 32) * 0.4 - 0.2).clamp(-0.1, 0.1) / 0.1 for _ in range(2)]
    losses = []
    for i, b in enumerate(batches):
        A.zero_grad(); A.train()
        torch.manual_seed(6300 + i)
        out, mean, logvar = A(b)
        torch.manual_seed(6300 + i)
        eps = torch.distributions.normal.Normal(0, 1).sample(mean.shape)
        rec = O.reconstruction_loss(out, b)
        kld = O.kld_loss(mean, logvar)
        (rec + kld).backward(); opt.step()
        ro = orc.step(b, eps)[0]
        assert torch.allclose(rec, ro, rtol=1e-5, atol=1e-7)
        losses += [rec.item(), kld.item()]
        st["vae/batch%d" % i], st["vae/eps%d" % i] = b.numpy(), eps.numpy()
    assert_state(A, orc.P, "vae")
    st["vae/losses"] = np.array(losses)
    put(st, "vae/final", {k: summarize(v.float()) for k, v in A.state_dict().items()})

    # ---- input pipeline: the reference's VoxelDataset on real files, and its create_batches
    # generator executed from the reference's own source text (the script cannot be imported: it trains at import) ----
    import ast
    import tempfile
    from shapegan_amd import datasets as native
    sys.path.insert(0, ref_import.REFERENCE_ROOT)
    import datasets as ref_datasets
    rng = np.random.RandomState(64)
    raw = (rng.rand(5, 8, 8, 8).astype(np.float32) * 0.4 - 0.2)
    raw[0, 0, 0, :6] = [np.nan, np.inf, -np.inf, 0.1, -0.1, np.float32(0.1) + np.float32(1e-8)]
    with tempfile.TemporaryDirectory() as tmp:
        names = []
        for i in range(5):
            names.append(os.path.join(tmp, "%c.npy" % "cadbe"[i]))
            np.save(names[-1], raw[i])
        for kw in (dict(), dict(rescale_sdf=False), dict(clamp=None)):
            theirs, ours = ref_datasets.VoxelDataset(names, **kw), native.VoxelDataset(names, **kw)
            for i in range(5):
                assert np.array_equal(theirs[i].numpy(), ours[i].numpy(), equal_nan=True)
        assert ref_datasets.VoxelDataset.glob(tmp + "/**.npy").files == native.VoxelDataset.glob(tmp + "/**.npy").files
        st["vox/raw"] = raw
        st["vox/rescaled"] = torch.stack([ref_datasets.VoxelDataset(names)[i] for i in range(5)]).numpy()
        ds = ref_datasets.VoxelDataset(names)
        ds.rescale_sdf = False
        st["vox/clamped"] = torch.stack([ds[i] for i in range(5)]).numpy()
    src = open(os.path.join(ref_import.REFERENCE_ROOT, "train_sdf_autodecoder.py")).read()
    fn = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "create_batches"][0]
    code = compile(ast.Module(body=[fn], type_ignores=[]), "train_sdf_autodecoder.py", "exec")
    for case, (count, batch) in enumerate(((1000, 64), (777, 50), (4096, 512))):
        signs = np.random.RandomState(70 + case).rand(count) > (0.5, 0.3, 0.8)[case]
        env = {"np": np, "signs": signs, "BATCH_SIZE": batch}
        exec(code, env)
        np.random.seed(700 + case)
        theirs = [b.copy() for b in env["create_batches"]()]
        np.random.seed(700 + case)
        ours = list(native.create_batches(signs, batch))
        assert len(theirs) == len(ours) and all(np.array_equal(a, b) for a, b in zip(theirs, ours))
        st["batches/%d/signs" % case] = signs
        st["batches/%d/flat" % case] = np.concatenate(theirs)
        st["batches/%d/sizes" % case] = np.array([len(b) for b in theirs])

    np.savez_compressed(os.path.join(OUT, "steps_f2.npz"), **st)
    print("wrote steps_f2.npz with %d arrays" % len(st))


if __name__ == "__main__":
    main()
