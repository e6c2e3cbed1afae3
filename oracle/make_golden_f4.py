"""Fixtures for SURVEY.md 8f rank 4 (PointNet-discriminator GAN, model/point_sdf_net.py + train_point_gan.py):
tests/golden/steps_f4.npz.  Run in THIS container (needs /root/reference).  Every quantity is produced by the REAL
reference classes and asserted equal to oracle/torch_oracle.py before it is written.  TEST INFRASTRUCTURE ONLY."""
import importlib.util
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_import, torch_oracle as O            # noqa: E402
from oracle.make_golden import grads_summary, put, summarize  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def state(m):
    return {k: v.clone() for k, v in m.state_dict().items()}


def main():
    spec = importlib.util.spec_from_file_location("ref_point_sdf_net",
                                                  os.path.join(ref_import.REFERENCE_ROOT, "model", "point_sdf_net.py"))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    st = {}

    # ---- module forwards / gradients ----
    torch.manual_seed(81)
    G = ref.SDFGenerator(128, 256, 8, True, dropout=0.0)
    pos, z = torch.rand(2, 96, 3) * 2 - 1, torch.randn(2, 128)
    out = G(pos, z)
    Pg = O.clone_state(state(G))
    out_o = O.sdf_generator_forward(Pg, pos, z)
    assert torch.equal(out, out_o)
    w = torch.randn_like(out)
    (out * w).sum().backward()
    (out_o * w).sum().backward()
    for k, p in G.named_parameters():
        if p.grad is not None:
            assert torch.allclose(p.grad, Pg[k].grad, rtol=1e-5, atol=1e-7), k
    st["gen/pos"], st["gen/z"], st["gen/w"], st["gen/out"] = pos.numpy(), z.numpy(), w.numpy(), out.detach().numpy()
    put(st, "gen/grad", grads_summary((k, p.grad) for k, p in G.named_parameters() if p.grad is not None))

    torch.manual_seed(82)
    D = ref.PointNet(out_channels=1)
    pos, dist = torch.rand(3, 50, 3) * 2 - 1, torch.rand(3, 50, 1) * 0.2 - 0.1
    out = D(pos, dist)
    Pd = O.clone_state(state(D))
    assert torch.equal(out, O.pointnet_forward(Pd, pos, dist))
    out.sum().backward()
    O.pointnet_forward(Pd, pos, dist).sum().backward()
    for k, p in D.named_parameters():
        assert torch.allclose(p.grad, Pd[k].grad, rtol=1e-5, atol=1e-7), k
    st["disc/pos"], st["disc/dist"], st["disc/out"] = pos.numpy(), dist.numpy(), out.detach().numpy()
    put(st, "disc/grad", grads_summary((k, p.grad) for k, p in D.named_parameters()))

    # ---- training steps (train_point_gan.py:52-83), B=2, P=128: critic update with gradient penalty, generator update ----
    torch.manual_seed(83)
    G, D = ref.SDFGenerator(128, 256, 8, True, dropout=0.0), ref.PointNet(out_channels=1)
    orc = O.PointGANOracle(state(G), state(D))
    g_opt = torch.optim.RMSprop(G.parameters(), lr=0.0001)
    d_opt = torch.optim.RMSprop(D.parameters(), lr=0.0001)
    uniform = torch.cat([torch.rand(2, 128, 3) * 2 - 1, torch.rand(2, 128, 1) * 0.2 - 0.1], dim=-1)
    z1, z2, alpha = torch.randn(2, 128), torch.randn(2, 128), torch.rand(2, 1, 1)
    u_pos, u_dist = uniform[..., :3], uniform[..., 3:]
    d_opt.zero_grad()
    fake = G(u_pos, z1)
    d_loss = D(u_pos, fake).mean() - D(u_pos, u_dist).mean()
    interpolated = alpha * u_dist + (1 - alpha) * fake
    interpolated.requires_grad_(True)
    o = D(u_pos, interpolated)
    grad = torch.autograd.grad(o, interpolated, grad_outputs=torch.ones_like(o), create_graph=True, retain_graph=True,
                               only_inputs=True)[0]
    gp = 10 * ((grad.view(grad.size(0), -1).norm(dim=-1, p=2) - 1).pow(2).mean())
    (d_loss + gp).backward()
    d_opt.step()
    dl_o, gp_o = orc.critic_step(uniform, z1, alpha)
    assert torch.allclose(d_loss, dl_o, rtol=1e-5, atol=1e-7) and torch.allclose(gp, gp_o, rtol=1e-5, atol=1e-7)
    g_opt.zero_grad()
    g_loss = -D(u_pos, G(u_pos, z2)).mean()
    g_loss.backward()
    g_opt.step()
    assert torch.allclose(g_loss, orc.generator_step(uniform, z2), rtol=1e-5, atol=1e-7)
    for k, v in G.state_dict().items():
        assert torch.allclose(v, orc.G[k], rtol=1e-4, atol=1e-6), k
    for k, v in D.state_dict().items():
        assert torch.allclose(v, orc.D[k], rtol=1e-4, atol=1e-6), k
    st["step/uniform"], st["step/z1"], st["step/z2"], st["step/alpha"] = uniform.numpy(), z1.numpy(), z2.numpy(), alpha.numpy()
    st["step/losses"] = np.array([d_loss.item(), gp.item(), g_loss.item()])
    put(st, "step/g_final", {k: summarize(v.float()) for k, v in G.state_dict().items()})
    put(st, "step/d_final", {k: summarize(v.float()) for k, v in D.state_dict().items()})

    np.savez_compressed(os.path.join(OUT, "steps_f4.npz"), **st)
    print("wrote steps_f4.npz with %d arrays" % len(st))


if __name__ == "__main__":
    main()
