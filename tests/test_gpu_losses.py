"""GPU tier: loss / blend / scatter_max kernels against torch CPU compositions of the reference
expressions, SDFNet inference helpers (8f rank 1), clip_weights (a5), and double backward through tanh / sigmoid."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"   # tests/test_cpu_twin.py re-runs these bodies on the CPU twin with DEV = "cpu"
RTOL = 1e-4


def close(a, b, rtol=RTOL, atol=None, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if atol is None:
        atol = rtol * max(float(b.abs().mean()), 1e-30)
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: what + ": " + m)


def close_mostly(a, b, rtol=RTOL, max_bad_frac=1e-3, what=""):
    """For gradients that pass through LeakyReLU / ReLU masks: an activation within rounding of 0 may take the other
    branch in two correct fp32 implementations; all but a sliver of the entries must agree, outliers stay bounded."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    atol = rtol * max(float(b.abs().mean()), 1e-30)
    bad = (a - b).abs() > (atol + rtol * b.abs())
    frac = float(bad.float().mean())
    assert frac <= max(max_bad_frac, 2.0 / bad.numel()), "%s: %.4f%% of elements out of tolerance" % (what, 100 * frac)
    if bad.any():
        assert float((a - b).abs().max()) <= 0.5 * float(b.abs().max()), what + ": outlier larger than the tensor scale"


# ---- voxel_difference: the integer, bit-exact loss of SURVEY.md 8 row a13 ----------------------------------------------------
def _ref_voxel_difference(input, target):
    """train_autoencoder.py:50-52, verbatim."""
    wrong_signs = (input * target) < 0
    return torch.sum(wrong_signs).item() / wrong_signs.nelement()


@pytest.mark.parametrize("shape", [(64, 32, 32, 32), (4, 32, 32, 32), (3, 7, 5), (1,), (2049,), (16, 64, 64, 64)])
def test_voxel_difference_bit_exact(shape):
    """sg_count_sign_mismatch against the reference expression on the CPU: the COUNT must be equal (integer work), hence the
    returned fraction is the same double.  Edge values: +-0, NaN, +-inf, denormals, products that underflow to -0 (not
    counted: the rounded fp32 product is what is compared with 0) and products that are denormal (counted)."""
    from shapegan_amd import ops
    from shapegan_amd.train_steps import voxel_difference
    torch.manual_seed(len(shape) * 7 + shape[0])
    a = torch.randn(shape)
    b = torch.randn(shape)
    fa, fb = a.view(-1), b.view(-1)
    n = fa.numel()
    special = [(0.0, -1.0), (-0.0, 1.0), (float("nan"), -1.0), (1.0, float("nan")), (float("inf"), -1.0),
               (float("-inf"), float("-inf")), (float("inf"), -0.0), (1e-30, -1e-30), (1e-20, -1e-20), (-1e-45, 1.0),
               (1e-45, -1e-45), (3.0, -2.0), (-3.0, -2.0)]
    gen = torch.Generator().manual_seed(n)
    for (x, y), pos in zip(special, torch.randperm(n, generator=gen)[:len(special)].tolist()):
        fa[pos], fb[pos] = x, y
    ref_count = int(torch.sum((a * b) < 0).item())
    got = ops.count_sign_mismatch(a.to(DEV), b.to(DEV))
    assert got.dtype == torch.int64 and int(got.item()) == ref_count
    assert voxel_difference(a.to(DEV), b.to(DEV)) == _ref_voxel_difference(a, b)
    # all-agree / all-disagree extremes
    c = torch.rand(shape) + 1
    assert voxel_difference(c.to(DEV), (c * 2).to(DEV)) == 0.0
    assert voxel_difference(c.to(DEV), (-c).to(DEV)) == 1.0


def test_voxel_difference_on_autoencoder_output(golden_modules):
    """The call site: `voxel_difference(output, test_set)` in train_autoencoder.py:64-77 (eval-mode autoencoder output against
    its input batch) — native module output, native count, against the reference expression on the same tensors."""
    from shapegan_amd.model.autoencoder import Autoencoder
    from shapegan_amd.train_steps import voxel_difference
    torch.manual_seed(11)
    ae = Autoencoder(is_variational=False).to(DEV)
    ae.eval()
    batch = (torch.rand(6, 32, 32, 32) * 2 - 1).to(DEV)
    with torch.no_grad():
        out = ae(batch)
    assert voxel_difference(out, batch) == _ref_voxel_difference(out.cpu(), batch.cpu())


# ---- reconstruction / KLD / DeepSDF losses --------------------------------------------------------------------------
def _ref_reconstruction_loss(output, target):
    """train_autoencoder.py:57-62, verbatim semantics (in-place masked scale)."""
    difference = output - target
    wrong_signs = target < 0
    difference[wrong_signs] *= 32
    return torch.mean(torch.abs(difference))


@pytest.mark.parametrize("shape", [(4, 32, 32, 32), (3, 7, 5), (1, 1)])
def test_weighted_l1_matches_reconstruction_loss(shape):
    from shapegan_amd import ops
    torch.manual_seed(sum(shape))
    out = torch.randn(shape)
    target = torch.randn(shape)
    out.view(-1)[0] = target.view(-1)[0]          # an exact zero difference: sign(0) = 0 in both
    target.view(-1)[-1] = 0.0                     # target == 0 is "not negative"
    o_ref = out.clone().requires_grad_(True)
    ref = _ref_reconstruction_loss(o_ref, target)
    ref.backward()
    o = out.cuda().requires_grad_(True)
    loss = ops.weighted_l1(o, target.cuda(), 32.0)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    (loss * 3.0).backward()
    close(o.grad, 3.0 * o_ref.grad, rtol=1e-6, atol=0.0, what="d loss / d output")
    # neg_weight 1: DeepSDF data term mean|out - sdf| (train_sdf_autodecoder.py:88)
    o2 = out.cuda().requires_grad_(True)
    l1 = ops.weighted_l1(o2, target.cuda())
    np.testing.assert_allclose(l1.item(), torch.mean(torch.abs(out - target)).item(), rtol=1e-5)
    l1.backward()
    close(o2.grad, torch.sign(out - target) / out.numel(), rtol=1e-6, atol=0.0)


def test_kld_matches_reference():
    from shapegan_amd import ops
    torch.manual_seed(3)
    mean, lv = torch.randn(32, 128), torch.randn(32, 128) * 0.5
    m_ref, l_ref = mean.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    ref = -0.5 * torch.sum(1 + l_ref - m_ref.pow(2) - l_ref.exp()) / m_ref.nelement()     # train_autoencoder.py:54-55
    ref.backward()
    m, l = mean.cuda().requires_grad_(True), lv.cuda().requires_grad_(True)
    loss = ops.kld(m, l)
    np.testing.assert_allclose(loss.item(), ref.item(), rtol=1e-5)
    loss.backward()
    close(m.grad, m_ref.grad, rtol=1e-5)
    close(l.grad, l_ref.grad, rtol=1e-5)


def test_mean_sq_plain_and_row_weighted():
    from shapegan_amd import ops
    torch.manual_seed(4)
    table = torch.randn(64, 256) * 0.01
    model_indices = torch.randint(0, 64, (5000,))
    rows = table[model_indices]
    t_ref = table.clone().requires_grad_(True)
    ref = 0.01 * torch.mean(torch.pow(t_ref[model_indices, :], 2))            # train_sdf_autodecoder.py:88
    ref.backward()
    # gathered rows
    r = rows.cuda().requires_grad_(True)
    got = ops.mean_sq(r, None, r.numel() / 0.01)
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-5)
    got.backward()
    close(r.grad, 0.01 * 2 * rows / rows.numel(), rtol=1e-5)
    # the same regulariser through shape counts on the table itself
    counts = torch.bincount(model_indices, minlength=64).float()
    t = table.cuda().requires_grad_(True)
    got2 = ops.mean_sq(t, counts.cuda(), rows.numel() / 0.01)
    np.testing.assert_allclose(got2.item(), ref.item(), rtol=1e-5)
    got2.backward()
    close(t.grad, t_ref.grad, rtol=1e-5)


@pytest.mark.parametrize("n,shapes,width,counted", [(20000, 64, 128, True), (200000, 64, 256, True), (5000, 5000, 16, False),
                                                    (1, 1, 1, True), (777, 3, 5, True)])
def test_deepsdf_loss_is_the_sum_of_its_two_ops_bit_for_bit(n, shapes, width, counted):
    """train_sdf_autodecoder.py:88 `l1(out, sdf) + SIGMA * mean(z_batch^2)`: the one-op form against the reference expression in
    torch (value, both gradients) and, bit for bit, against weighted_l1 + mean_sq + the fp32 add it replaces."""
    from shapegan_amd import ops
    g = torch.Generator().manual_seed(n + width)
    out = (torch.rand(n, generator=g) * 0.4 - 0.2)
    sdf = (torch.rand(n, generator=g) * 0.2 - 0.1)
    out[::7] = sdf[::7]                                           # exact zeros of the difference: sign(0) = 0
    table = torch.randn(shapes, width, generator=g) * 0.01
    sigma = 0.01
    if counted:
        model_indices = torch.randint(0, shapes, (n,), generator=g)
        counts = torch.bincount(model_indices, minlength=shapes).float()
        z, rw, denom = table, counts, n * width / sigma
        o_ref, t_ref = out.clone().requires_grad_(True), table.clone().requires_grad_(True)
        ref = torch.nn.functional.l1_loss(o_ref, sdf) + sigma * torch.mean(torch.pow(t_ref[model_indices, :], 2))
    else:
        z, rw, denom = table, None, table.numel() / sigma
        o_ref, t_ref = out.clone().requires_grad_(True), table.clone().requires_grad_(True)
        ref = torch.nn.functional.l1_loss(o_ref, sdf) + sigma * torch.mean(torch.pow(t_ref, 2))
    (ref * 1.3).backward()
    o1, z1 = out.clone().to(DEV).requires_grad_(True), z.clone().to(DEV).requires_grad_(True)
    rwd = None if rw is None else rw.to(DEV)
    fused = ops.deepsdf_loss(o1, sdf.to(DEV), z1, rwd, denom)
    (fused * 1.3).backward()
    o2, z2 = out.clone().to(DEV).requires_grad_(True), z.clone().to(DEV).requires_grad_(True)
    split = ops.weighted_l1(o2, sdf.to(DEV)) + ops.mean_sq(z2, rwd, denom)
    (split * 1.3).backward()
    assert torch.equal(fused.detach(), split.detach())
    assert torch.equal(o1.grad, o2.grad) and torch.equal(z1.grad, z2.grad)
    np.testing.assert_allclose(fused.item(), ref.item(), rtol=2e-6)
    close(o1.grad, o_ref.grad, rtol=1e-6, atol=0.0 if n > 1 else 1e-12, what="d loss / d out")
    # (torch's reference accumulates a row's n / shapes gathered contributions one by one in fp32 — that sum is the noisy side)
    close(z1.grad, t_ref.grad, rtol=2e-4 if counted else 1e-5, what="d loss / d latent codes")


# ---- gradient penalty pieces -----------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,n_first", [(128, 64), (8, 3), (1, 1), (5000, 1), (7, 7), (7, 0)])
def test_mean_difference_matches_torch(n, n_first):
    """train_wgan.py:68 / :82: mean(fake scores) - mean(real scores) on the concatenated critic batch, and -mean(scores), as
    one launch each way; value and gradient against torch."""
    from shapegan_amd import ops
    torch.manual_seed(n + n_first)
    x = torch.randn(n, 1) * 3 + 1
    xr = x.clone().requires_grad_(True)
    if 0 < n_first < n:
        ref = torch.mean(xr[:n_first]) - torch.mean(xr[n_first:])
    elif n_first == n:
        ref = torch.mean(xr)
    else:
        ref = -torch.mean(xr)
    (ref * 1.7).backward()
    xg = x.clone().to(DEV).requires_grad_(True)
    got = ops.mean_difference(xg, n_first)
    (got * 1.7).backward()
    close(got, ref.detach(), rtol=1e-6, atol=1e-6, what="mean difference")
    close(xg.grad, xr.grad, rtol=1e-6, atol=1e-9, what="d mean difference")
    xg2 = x.clone().to(DEV).requires_grad_(True)
    neg = ops.neg_mean(xg2)
    neg.backward()
    close(neg, -x.mean(), rtol=1e-6, atol=1e-6, what="neg mean")
    close(xg2.grad, torch.full_like(x, -1.0 / n), rtol=1e-6, atol=1e-9, what="d neg mean")
    # as the root of lib.backward() (what the trainers do) the gradient comes out of the forward launch: the same values
    from shapegan_amd import lib
    xg3 = x.clone().to(DEV).requires_grad_(True)
    lib.backward(ops.mean_difference(xg3, n_first))
    xg4 = x.clone().to(DEV).requires_grad_(True)
    ops.mean_difference(xg4, n_first).backward()
    assert torch.equal(xg3.grad, xg4.grad)
    close(xg3.grad, xr.grad / 1.7, rtol=1e-6, atol=1e-9, what="d mean difference, unit upstream gradient")


def test_lerp_rows_bit_exact():
    from shapegan_amd import ops
    torch.manual_seed(5)
    real, fake = torch.randn(16, 16, 16, 16), torch.randn(16, 16, 16, 16)
    alpha = torch.rand(16, 1, 1, 1)
    a = alpha.expand(real.shape)
    ref = a * real + ((1 - a) * fake)                                         # train_hybrid_progressive_gan.py:104-105
    got = ops.lerp_rows(real.cuda(), fake.cuda(), alpha.cuda())
    assert got.shape == real.shape and not got.requires_grad
    diff = (got.cpu() - ref).abs()
    assert torch.equal(got.cpu(), ref), "%d entries differ, max %.3e" % (int((diff > 0).sum()), float(diff.max()))


@pytest.mark.parametrize("B,shape", [(16, (32, 32, 32)), (5, (7, 3)), (3, (1000,))])
def test_gradient_penalty_value_and_gradient(B, shape):
    from shapegan_amd import ops
    torch.manual_seed(B)
    g = torch.randn((B,) + shape) * 0.01
    g[1] = 0.0                                                                # a zero-norm row: torch.norm's subgradient is 0
    g_ref = g.clone().requires_grad_(True)
    dims = tuple(range(1, g.dim()))
    ref = ((g_ref.norm(2, dim=dims) - 1) ** 2).mean() * 10.0                  # train_hybrid_progressive_gan.py:111
    ref.backward()
    gg = g.cuda().requires_grad_(True)
    got = ops.gradient_penalty(gg, 10.0)
    np.testing.assert_allclose(got.item(), ref.item(), rtol=1e-5)
    got.backward()
    close(gg.grad, g_ref.grad, rtol=1e-5, what="d gp / d gradients")
    assert float(gg.grad[1].abs().sum()) == 0.0


# ---- fade-in blend ----------------------------------------------------------------------------------------------------
def test_subsample2_bit_exact_and_adjoint():
    from shapegan_amd import ops
    torch.manual_seed(6)
    x = torch.randn(3, 16, 16, 16)
    xg = x.cuda().requires_grad_(True)
    half = ops.Subsample2.apply(xg)
    assert torch.equal(half.cpu(), x[:, ::2, ::2, ::2])
    w = torch.randn(3, 8, 8, 8)
    (half * w.cuda()).sum().backward()
    ref = torch.zeros_like(x)
    ref[:, ::2, ::2, ::2] = w
    assert torch.equal(xg.grad.cpu(), ref)


@pytest.mark.parametrize("it,fade", [(1, 0.3), (3, 0.75)])
def test_fade_blend_first_and_second_order(it, fade):
    """fade*x + (1-fade)*from_SDF(x_in[:, ::2, ::2, ::2]) (model/progressive_gan.py:48-50) with a quadratic head, so
    that the double backward (the gradient penalty's path through the blend) is exercised."""
    from shapegan_amd import ops
    res = [8, 16, 32, 64][it]
    C = [128, 64, 32, 1][it - 1]
    B, r = 2, res // 2
    torch.manual_seed(it)
    x = torch.randn(B, C, r, r, r)
    x_in = torch.randn(B, res, res, res)
    w = torch.randn(B, C, r, r, r)

    def reference(x, x_in):
        half = x_in[:, ::2, ::2, ::2]
        x2 = O.from_sdf(half, it - 1)
        return fade * x + (1.0 - fade) * x2

    xr, ir = x.clone().requires_grad_(True), x_in.clone().requires_grad_(True)
    yr = reference(xr, ir)
    (gi_ref,) = torch.autograd.grad((yr * yr * w).sum(), ir, create_graph=True)
    (gi_ref.pow(2).sum()).backward()
    xg, ig = x.cuda().requires_grad_(True), x_in.cuda().requires_grad_(True)
    y = ops.fade_blend(xg, ig, fade)
    close(y, yr, rtol=1e-6, atol=1e-7, what="blend forward")
    (gi,) = torch.autograd.grad((y * y * w.cuda()).sum(), ig, create_graph=True)
    close(gi, gi_ref, rtol=1e-5, what="d/dx_in")
    (gi.pow(2).sum()).backward()
    close(xg.grad, xr.grad, rtol=1e-5, what="double backward wrt x")
    close(ig.grad, ir.grad, rtol=1e-5, what="double backward wrt x_in")


# ---- scatter_max over ragged batch vectors ----------------------------------------------------------------------------
@pytest.mark.parametrize("N,B,C,sorted_batch", [(1000, 7, 64, True), (4096, 16, 512, False), (10, 12, 5, False)])
def test_scatter_max_ragged(N, B, C, sorted_batch):
    from shapegan_amd import ops
    torch.manual_seed(N + B)
    x = torch.randn(N, C)
    x[N // 2] = x[N // 3]                    # duplicated rows: ties (when both fall into the same segment)
    batch = torch.randint(0, B, (N,))
    if N <= 10:
        batch[:] = torch.tensor([0, 3, 3, 5, 5, 5, 11, 0, 3, 11])            # segments 1,2,4,6..10 are empty -> 0
    if sorted_batch:
        batch = torch.sort(batch).values
    xr = x.clone().requires_grad_(True)
    idx = batch.unsqueeze(1).expand(N, C)
    ref = torch.zeros(B, C).scatter_reduce(0, idx, xr, reduce="amax", include_self=False)
    empty = torch.bincount(batch, minlength=B) == 0
    ref = torch.where(empty.unsqueeze(1), torch.zeros_like(ref), ref)
    xg = x.cuda().requires_grad_(True)
    out = ops.scatter_max(xg, batch.cuda(), B)
    assert torch.equal(out.cpu(), ref.detach())
    w = torch.randn(B, C)
    (out * w.cuda()).sum().backward()
    # first-occurrence argmax (torch's amax backward splits ties evenly, so build the expected gradient by hand)
    expect = torch.zeros(N, C)
    for b in range(B):
        rows = torch.nonzero(batch == b).flatten()
        if rows.numel():
            am = x[rows].argmax(dim=0)        # first maximal row within the segment
            expect[rows[am], torch.arange(C)] = w[b]
    assert torch.equal(xg.grad.cpu(), expect)
    # adjoint of the adjoint (double backward under a gradient penalty)
    xg2 = x.cuda().requires_grad_(True)
    out2 = ops.scatter_max(xg2, batch.cuda(), B)
    (g,) = torch.autograd.grad((out2 * out2).sum(), xg2, create_graph=True)
    (g * torch.randn(N, C, generator=torch.Generator().manual_seed(1)).cuda()).sum().backward()
    assert torch.isfinite(xg2.grad).all()


def test_pointnet_ragged_batch_equals_dense():
    """PointNet(pos, dist, batch) on the concatenation of equal-size clouds == the dense [B,P,...] call
    (model/point_sdf_net.py:39-43)."""
    from shapegan_amd.model.point_sdf_net import PointNet
    torch.manual_seed(11)
    net = PointNet(out_channels=1).cuda()
    B, P = 4, 256
    pos, dist = torch.rand(B, P, 3, device=DEV) * 2 - 1, torch.randn(B, P, 1, device=DEV) * 0.1
    dense = net(pos, dist)
    batch = torch.arange(B, device=DEV).view(-1, 1).repeat(1, P).view(-1)
    perm = torch.randperm(B * P, device=DEV)
    ragged = net(pos.reshape(-1, 3)[perm], dist.reshape(-1, 1)[perm], batch[perm])
    close(ragged, dense.reshape(B, -1), rtol=1e-5)


# ---- tanh / sigmoid second-order terms ---------------------------------------------------------------------------------
@pytest.mark.parametrize("use_sigmoid", [True, False])
def test_double_backward_through_sigmoid_discriminator(use_sigmoid):
    """A gradient penalty on gan.Discriminator with its sigmoid on: autograd.grad(create_graph=True) then backward must
    carry the sigmoid's second-order term (ActBwd.backward's d/dy)."""
    from shapegan_amd.model.gan import Discriminator
    torch.manual_seed(12)
    d = Discriminator()
    d.use_sigmoid = use_sigmoid
    sd = {k: v.detach().cpu().clone() for k, v in d.state_dict().items()}
    x = torch.rand(3, 32, 32, 32) * 2 - 1
    P = O.clone_state(sd)
    xr = x.clone().requires_grad_(True)
    out_ref = O.discriminator_forward(P, xr, use_sigmoid)
    (g_ref,) = torch.autograd.grad(out_ref.sum(), xr, create_graph=True)
    ((g_ref.norm(2, dim=(1, 2, 3)) - 1) ** 2).mean().backward()
    xg = x.cuda().requires_grad_(True)
    out = d(xg)
    close(out, out_ref, what="forward")
    (g,) = torch.autograd.grad(out.sum(), xg, create_graph=True)
    close_mostly(g, g_ref, what="input gradient")
    ((g.norm(2, dim=(1, 2, 3)) - 1) ** 2).mean().backward()
    for k, p in d.named_parameters():
        if P[k].grad is None:                      # the last bias never enters dD/dx
            assert p.grad is None or float(p.grad.abs().sum()) == 0.0, k
            continue
        close_mostly(p.grad, P[k].grad, rtol=2e-4, what="double-backward grad " + k)


@pytest.mark.parametrize("act_mod", [torch.nn.Tanh, torch.nn.Sigmoid, lambda: torch.nn.LeakyReLU(0.2)])
def test_stack_conv_bn_act_gradients(act_mod):
    """[ConvTranspose3d, BatchNorm3d, act] with tanh / sigmoid fused into the BatchNorm pass (model/stack.py): gradients
    must match torch for every activation, not only LeakyReLU / ReLU."""
    from shapegan_amd.model.stack import run_stack
    torch.manual_seed(13)
    seq = torch.nn.Sequential(torch.nn.ConvTranspose3d(6, 5, 4, 2, 1), torch.nn.BatchNorm3d(5), act_mod())
    x = torch.randn(3, 6, 4, 4, 4)
    w = torch.randn(3, 5, 8, 8, 8)
    xr = x.clone().requires_grad_(True)
    (seq(xr) * w).sum().backward()
    ref = {k: p.grad.clone() for k, p in seq.named_parameters()}
    seq.zero_grad()
    seq_g = seq.cuda()
    xg = x.cuda().requires_grad_(True)
    y = run_stack(seq_g, xg, True)
    (y * w.cuda()).sum().backward()
    close_mostly(xg.grad, xr.grad, rtol=2e-4, what="dx")
    for k, p in seq_g.named_parameters():
        if k == "0.bias":
            continue                                   # mathematically zero in front of a training-mode BatchNorm
        close_mostly(p.grad, ref[k], rtol=2e-4, what=k)


# ---- a5: Discriminator.clip_weights ------------------------------------------------------------------------------------
def test_discriminator_clip_weights():
    """model/gan.py:67-69 through the drop-in method (sg_clamp on each parameter's own storage)."""
    from shapegan_amd.model.gan import Discriminator
    torch.manual_seed(14)
    d = Discriminator()
    before = {k: v.detach().cpu().clone() for k, v in d.named_parameters()}
    ptrs = {k: v.data_ptr() for k, v in d.named_parameters()}
    d.clip_weights(0.01)
    for k, p in d.named_parameters():
        assert torch.equal(p.detach().cpu(), before[k].clamp(-0.01, 0.01)), k
        assert p.data_ptr() == ptrs[k]
    assert any(float(v.abs().max()) > 0.01 for v in before.values())


# ---- 8f rank 1: SDFNet inference helpers --------------------------------------------------------------------------------
def _chairs_net(chairs_state):
    from shapegan_amd.model.sdf_net import SDFNet
    net = SDFNet()
    net.load_state_dict(chairs_state)
    net.eval()
    return net


def _oracle_sdf_and_normals(state, points, latent):
    P = O.clone_state(state, requires_grad=False)
    p = points.clone().requires_grad_(True)
    sdf = O.sdfnet_forward(P, p, latent.reshape(1, -1).repeat(p.shape[0], 1))
    sdf.backward(torch.ones_like(sdf))
    normals = p.grad / torch.norm(p.grad, dim=1).unsqueeze(1)
    return sdf.detach(), normals, p.grad


def test_get_normals(chairs_state):
    """model/sdf_net.py:118-128: d sdf / d points, normalised; also the API contract (raises for tensors that require grad)."""
    net = _chairs_net(chairs_state)
    torch.manual_seed(15)
    z = torch.randn(128) * 0.5
    pts = torch.rand(3000, 3) * 2 - 1
    n = net.get_normals(z.cuda(), pts.clone().cuda())
    _, n_ref, raw = _oracle_sdf_and_normals(chairs_state, pts, z)
    keep = raw.norm(dim=1) > 1e-4                              # a vanishing gradient has no direction to compare
    assert float(keep.float().mean()) > 0.95
    cos = (n.cpu()[keep] * n_ref[keep]).sum(dim=1)
    assert float((cos < 1 - 1e-4).float().mean()) <= 2e-3      # ReLU-kink flips: isolated points only
    np.testing.assert_allclose(n.cpu()[keep].norm(dim=1).numpy(), 1.0, rtol=1e-5)
    with pytest.raises(Exception, match="require grad"):
        net.get_normals(z.cuda(), pts.cuda().requires_grad_(True))


def test_get_surface_points_seeded(chairs_state):
    """model/sdf_net.py:130-156 with the device RNG seeded: the same draws, projected with the oracle's autograd on the CPU."""
    from shapegan_amd.util import get_points_in_unit_sphere
    net = _chairs_net(chairs_state)
    z = torch.randn(128, generator=torch.Generator().manual_seed(16)) * 0.5
    for use_unit_sphere in (True, False):
        torch.manual_seed(17)
        if use_unit_sphere:
            pts = get_points_in_unit_sphere(n=2000, device=net.device) * 1.1
        else:
            pts = torch.rand((2000, 3), device=net.device) * 2.2 - 1
        pts = pts.cpu()
        torch.manual_seed(17)
        got_p, got_n = net.get_surface_points(z.cuda(), sample_size=2000, sdf_cutoff=0.1, return_normals=True,
                                              use_unit_sphere=use_unit_sphere)
        sdf, n_ref, _ = _oracle_sdf_and_normals(chairs_state, pts, z)
        proj = pts - n_ref * sdf.unsqueeze(1)
        mask = (sdf.abs() < 0.1) & torch.all(torch.isfinite(proj), dim=1)
        # the mask is a threshold on fp32 values: allow a few borderline points to fall on the other side
        assert abs(int(mask.sum()) - got_p.shape[0]) <= 3
        if int(mask.sum()) == got_p.shape[0]:
            err = (got_p.cpu() - proj[mask]).norm(dim=1)
            assert float((err > 1e-4).float().mean()) <= 5e-3
            cos = (got_n.cpu() * n_ref[mask]).sum(dim=1)
            assert float((cos < 1 - 1e-4).float().mean()) <= 5e-3
    out = net.get_surface_points_in_batches(z.cuda(), amount=500)
    assert out.shape == (500, 3) and torch.isfinite(out).all()


def test_get_voxels_full_grid_and_padding(chairs_state):
    """model/sdf_net.py:90-93: sphere_only=False evaluates the whole grid, pad=True wraps it in a layer of 1s."""
    from shapegan_amd.util import get_voxel_coordinates
    net = _chairs_net(chairs_state)
    z = torch.randn(128, generator=torch.Generator().manual_seed(18)) * 0.5
    R = 16
    padded = net.get_voxels(z.cuda(), R, sphere_only=False, pad=True)
    plain = net.get_voxels(z.cuda(), R, sphere_only=False, pad=False)
    assert padded.shape == (R + 2,) * 3 and plain.shape == (R,) * 3
    np.testing.assert_array_equal(padded[1:-1, 1:-1, 1:-1], plain)
    shell = padded.copy()
    shell[1:-1, 1:-1, 1:-1] = 1
    assert (shell == 1).all()
    P = O.clone_state(chairs_state, requires_grad=False)
    pts = torch.tensor(get_voxel_coordinates(R))
    ref = O.sdfnet_forward(P, pts, z.reshape(1, -1).repeat(pts.shape[0], 1)).detach().reshape(R, R, R)
    np.testing.assert_allclose(plain, ref.numpy(), rtol=1e-4, atol=2e-6)
    sphere = net.get_voxels(z.cuda(), R, sphere_only=True)
    inside = np.linalg.norm(get_voxel_coordinates(R), axis=1).reshape(R, R, R) < 1.1
    np.testing.assert_array_equal(sphere[inside], plain[inside])            # same kernel, same points: bit-equal
    assert (sphere[~inside] == 1).all()


def test_graph_replay_invalidates_weight_packs():
    """After step_graphed() replays, eager SDFNet calls must see the updated weights (the captured Adam kernel
    changes parameters through raw pointers, so the pack cache has to be invalidated by the replay)."""
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    pc, shapes, L = 1000, 4, 128
    torch.manual_seed(19)
    pts = torch.rand(shapes * pc, 3, device=DEV) * 2 - 1
    sdf = torch.rand(shapes * pc, device=DEV) * 0.3 - 0.15
    net = SDFNet(latent_code_size=L)
    tr = SDFAutoDecoderTrainer(net, torch.randn(shapes, L, device=DEV) * 1e-2, pts, sdf, pointcloud_size=pc, lr=1e-3,
                               capturable=True)
    probe, z = pts[:512].contiguous(), torch.randn(1, L, device=DEV)
    with torch.no_grad():
        net.forward_shapes(probe, z, 512)                       # primes the pack cache with the initial weights
    for _ in range(6):
        tr.step_graphed(torch.randint(0, shapes * pc, (2048,), device=DEV))
    with torch.no_grad():
        after = net.forward_shapes(probe, z, 512)
    fresh = SDFNet(latent_code_size=L)
    fresh.load_state_dict({k: v.detach().clone() for k, v in net.state_dict().items()})
    with torch.no_grad():
        expect = fresh.forward_shapes(probe, z, 512)
    assert torch.equal(after, expect)


@pytest.mark.parametrize("graphed,late_polls", [(False, 0), (True, 0), (False, 2)])
def test_bad_batch_index_leaves_the_state_before_the_bad_batch(graphed, late_polls, monkeypatch):
    """train_sdf_autodecoder.py:79 raises an IndexError BEFORE any update.  Here the error is noticed without a host synchronisation,
    possibly some steps late — but the optimizers are guarded by the device word the sort kernel sets (sg_adam_step_guarded), so when
    the IndexError arrives, parameters, latent table, both Adam moments and the step counters (host lists, or the device counter
    inside a replayed graph) are bit-for-bit those of the step before the bad batch; training then continues normally.
    late_polls: the host's look at the pinned word is made to miss the flag that many times (a host running ahead of the device)."""
    from shapegan_amd import ops
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    if graphed and DEV == "cpu":
        pytest.skip("graph capture needs the GPU")
    pc, shapes, L, n = 3000, 4, 64, 8192
    torch.manual_seed(23)
    pts = torch.rand(shapes * pc, 3, device=DEV) * 2 - 1
    sdf = torch.rand(shapes * pc, device=DEV) * 0.3 - 0.15
    net = SDFNet(latent_code_size=L)
    tr = SDFAutoDecoderTrainer(net, torch.randn(shapes, L, device=DEV) * 1e-2, pts, sdf, pointcloud_size=pc, lr=1e-3,
                               capturable=graphed)
    step = tr.step_graphed if graphed else tr.step
    good = lambda: torch.randint(0, shapes * pc, (n,), device=DEV)

    def state():
        if DEV != "cpu":
            torch.cuda.synchronize()
        out = [p.detach().clone() for p in net.parameters()] + [tr.latent_codes.detach().clone()]
        for o in (tr.net_opt, tr.lat_opt):
            out += [o.exp_avg.clone(), o.exp_avg_sq.clone()]
            out.append(o.step_dev.clone() if graphed else torch.tensor(o.steps))
        return out

    for _ in range(4):
        step(good())
    before, host_steps = state(), list(tr.net_opt.steps)
    real_poll, misses = tr._words.raise_if_bad, [late_polls]

    def poll(**kw):
        # ONE look at the pinned word per poll (two looks race with the sort kernel: "not pending" for the miss counter, then
        # "pending" inside the real poll a microsecond later raised at once, whatever late_polls said — seen when the step got shorter)
        if not tr._words.pending():
            return
        if misses[0] > 0:
            misses[0] -= 1
            return
        real_poll(**kw)
    monkeypatch.setattr(tr._words, "raise_if_bad", poll)
    bad = good()
    bad[n // 3] = shapes * pc + 3
    raised, calls = False, 0
    for idx in [bad] + [good() for _ in range(4)]:       # the host is not synchronised: the error may arrive a few calls late
        calls += 1
        try:
            step(idx)
        except IndexError as e:
            raised = True
            assert e.sort_sequence > 0
            break
    assert raised, "the out-of-range index was never reported"
    assert calls > late_polls
    after = state()
    for a, b in zip(before, after):
        assert torch.equal(a, b), "state changed by (or after) the bad batch, %d call(s) before the IndexError" % calls
    step(good())                                          # the words were cleared: training goes on
    moved = state()
    assert not torch.equal(moved[0], before[0])
    if not graphed:
        assert tr.net_opt.steps == [s + 1 for s in host_steps]


def test_bad_batch_index_words_belong_to_one_trainer():
    """the bad-index words used to be one pair per DEVICE.  A second trainer on the same device then had its updates
    no-op'd by the first one's bad batch without its host counters knowing, and whoever polled first cleared the other's error; a
    gathered step (below the 8192-point threshold) behind a bad sorted batch was dropped on the device but not counted.  Each
    trainer owns its pair now: B trains on undisturbed while A's error is pending, A's IndexError arrives at A — also when A's next
    step is a gathered one — with A's state bit-for-bit that of the step before the bad batch."""
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.train_steps import SDFAutoDecoderTrainer
    pc, shapes, L, n = 3000, 4, 64, 8192
    torch.manual_seed(29)
    pts = torch.rand(shapes * pc, 3, device=DEV) * 2 - 1
    sdf = torch.rand(shapes * pc, device=DEV) * 0.3 - 0.15

    def make():
        return SDFAutoDecoderTrainer(SDFNet(latent_code_size=L), torch.randn(shapes, L, device=DEV) * 1e-2, pts, sdf,
                                     pointcloud_size=pc, lr=1e-3)
    A, B = make(), make()
    assert A._words is not B._words and A.net_opt.guard.data_ptr() != B.net_opt.guard.data_ptr()
    good = lambda k=n: torch.randint(0, shapes * pc, (k,), device=DEV)

    def state(tr):
        if DEV != "cpu":
            torch.cuda.synchronize()
        return [p.detach().clone() for p in tr.net.parameters()] + [tr.latent_codes.detach().clone(), torch.tensor(tr.net_opt.steps)]
    for _ in range(3):
        A.step(good())
        B.step(good())
    a0, b0 = state(A), state(B)
    bad = good()
    bad[7] = -1
    raised = False
    try:
        A.step(bad)                              # GPU: noticed late (no synchronisation on this path); the synchronous twin: at once
    except IndexError:
        raised = True
    B.step(good())                               # B is not A: its update is applied and counted, nothing raises
    b1 = state(B)
    assert not torch.equal(b1[0], b0[0]) and torch.equal(b1[-1], b0[-1] + 1)
    if not raised:
        with pytest.raises(IndexError):
            A.step(good(500))                    # a GATHERED step: polls A's words first
    for x, y in zip(a0, state(A)):
        assert torch.equal(x, y), "A's state is not that of the step before its bad batch"
    B.step(good())                               # B never sees A's error
    A.step(good(500))                            # and A goes on
    assert torch.equal(state(A)[-1], a0[-1] + 1) and torch.equal(state(B)[-1], b0[-1] + 2)


# ---- C-ABI RCCL exchange (sg_allreduce_*) -----------------------------------------------------------------------
def test_native_allreduce_single_rank_stream_order():
    """libshapegan_comm.so on the one GPU of this box: communicator of world size 1 (all a single-GPU box can form), the
    exchange runs on the communicator's stream AFTER the producer kernel on the compute stream, and the compute stream sees the
    result after sg_allreduce_wait.  Sum over one rank = identity."""
    from shapegan_amd.parallel import NativeComm
    comm = NativeComm(rank=0, world=1)
    n = 1 << 24
    x = torch.zeros(n, device=DEV)
    for it in range(3):
        x.add_(1.5)                    # producer on the compute stream
        comm.launch(x[: n // 2])       # two slices of the flat buffer, like the tail / head exchange
        comm.launch(x[n // 2:])
        comm.wait()
        y = x * 2.0                    # consumer on the compute stream
        assert float(y[0]) == 3.0 * (it + 1) and float(y[-1]) == 3.0 * (it + 1)
    torch.cuda.synchronize()
    comm.close()


# ---- round 4: the critic's tail as one node, the classic-GAN losses, the VAE reparameterisation ---------------------------------
@pytest.mark.parametrize("N,C,act", [(128, 256, 1), (64, 256, 1), (5, 24, 2), (17, 8, 0), (1, 4, 1)])
def test_head_dot_forward_and_backward(N, C, act):
    """sg_head_dot_fwd / _bwd (model/gan.py:54-55: LeakyReLU -> Conv3d(C -> 1, k4 s1) on the 4^3 grid) against the torch
    composition in fp64: scores, gradient w.r.t. the pre-activation, head weight / bias gradients and the bias gradient of the
    layer below (channel sums of gz) — at the critic's shapes (128 and 64 samples x 256 channels) and ragged ones."""
    from shapegan_amd import lib as L
    from shapegan_amd.ops import check, ptr, stream
    torch.manual_seed(N * 31 + C)
    slope = 0.2
    z = torch.randn(N, C, 4, 4, 4)
    w = torch.randn(1, C, 4, 4, 4) * 0.05
    b = torch.randn(1)
    gy = torch.randn(N)
    zd = z.double().requires_grad_()
    wd, bd = w.double().requires_grad_(), b.double().requires_grad_()
    a = F.leaky_relu(zd, slope) if act == 1 else (F.relu(zd) if act == 2 else zd)
    yref = F.conv3d(a, wd, bd).reshape(N)
    (yref * gy.double()).sum().backward()
    lib = L.load()
    zg, wg, bg, gyg = z.to(DEV), w.to(DEV), b.to(DEV), gy.to(DEV)
    y = torch.empty(N, device=DEV)
    check(lib.sg_head_dot_fwd(ptr(zg), ptr(wg), ptr(bg), ptr(y), N, C * 64, act, slope, stream()), "head_dot_fwd")
    close(y, yref, rtol=1e-5, what="head scores")
    gz, gw, gb, gbz = torch.empty_like(zg), torch.empty_like(wg), torch.empty(1, device=DEV), torch.empty(C, device=DEV)
    check(lib.sg_head_dot_bwd(ptr(zg), ptr(wg), ptr(gyg), ptr(gz), ptr(gw), ptr(gb), ptr(gbz), None, N, C, 64, act, slope, stream()),
          "head_dot_bwd")
    close(gz, zd.grad, rtol=1e-6, what="d / d pre-activation")
    close(gw, wd.grad, rtol=1e-5, what="head weight gradient")
    close(gb, bd.grad, rtol=1e-5, what="head bias gradient")
    close(gbz, zd.grad.sum(dim=(0, 2, 3, 4)), rtol=1e-5, atol=1e-5 * float(zd.grad.abs().sum(dim=(0, 2, 3, 4)).mean()),
          what="bias gradient of the layer below")
    # optional outputs (a frozen critic in the generator update): only gz
    gz2 = torch.empty_like(zg)
    check(lib.sg_head_dot_bwd(ptr(zg), ptr(wg), ptr(gyg), ptr(gz2), None, None, None, None, N, C, 64, act, slope, stream()), "head_dot_bwd")
    assert torch.equal(gz2, gz)


def test_conv_head_node_matches_the_two_layer_composition():
    """ops.conv_head (what model/stack.py dispatches gan.Discriminator's last two layers to) against ConvFwd + LinearAct, the
    path it replaces: outputs and every gradient, plain backward and the create_graph form (gradient penalty)."""
    from shapegan_amd import ops
    torch.manual_seed(11)
    N = 6
    x = torch.randn(N, 16, 8, 8, 8)
    w, b = torch.randn(32, 16, 4, 4, 4) * 0.05, torch.randn(32) * 0.1
    wh, bh = torch.randn(1, 32, 4, 4, 4) * 0.05, torch.randn(1)
    gy = torch.randn(N)

    def run(fused, create_graph):
        leaves = [t.clone().to(DEV).requires_grad_() for t in (x, w, b, wh, bh)]
        xx, ww, bb, wwh, bbh = leaves
        if fused:
            y = ops.conv_head(xx, ww, bb, 1, 0.2, wwh, bbh)
        else:
            h = ops.conv3d_k4s2p1(xx, ww, bb, 1, 0.2)
            y = ops.LinearAct.apply(h.reshape(N, -1), wwh.reshape(1, -1), bbh, 0, 0.0, False, 0).reshape(N)
        if not create_graph:
            (y * gy.to(DEV)).sum().backward()
            return [y] + [t.grad for t in leaves]
        (gx,) = torch.autograd.grad(y, xx, grad_outputs=torch.ones_like(y), create_graph=True)
        pen = ((gx.reshape(N, -1).norm(dim=1) - 1) ** 2).mean()
        pen.backward()
        return [pen, gx] + [leaves[i].grad for i in (1, 3)]     # the penalty reaches the two weights
    for cg in (False, True):
        for got, ref, name in zip(run(True, cg), run(False, cg), "abcdefg"):
            close_mostly(got, ref, rtol=2e-5, max_bad_frac=2e-3, what="conv_head create_graph=%s output %s" % (cg, name))


@pytest.mark.parametrize("n", [64, 8, 1, 777])
def test_bce_and_neg_mean_log_match_torch(n):
    """sg_loss_bce_* / sg_loss_neg_mean_log_* against torch.nn.functional.binary_cross_entropy with constant targets and
    -torch.mean(torch.log(p)) (train_gan.py:30,65,78,84), values and gradients, including saturated scores (the -100 clamp of the
    logarithm and the 1e-12 floor of the backward's denominator)."""
    from shapegan_amd import ops
    torch.manual_seed(n)
    p = torch.rand(n).clamp(1e-4, 1 - 1e-4)
    if n >= 8:
        p[0], p[1], p[2] = 1.0, 0.0, 1e-30       # saturated discriminator outputs
    for target in (0.0, 1.0):
        pr = p.clone().requires_grad_()
        ref = F.binary_cross_entropy(pr, torch.full_like(pr, target))
        (ref * 1.7).backward()
        pg = p.clone().to(DEV).requires_grad_()
        got = ops.bce_const(pg, target)
        (got * 1.7).backward()
        close(got, ref, rtol=1e-6, what="bce target %g" % target)
        torch.testing.assert_close(pg.grad.cpu(), pr.grad, rtol=1e-5, atol=1e-30, msg=lambda m: "bce grad target %g: %s" % (target, m))
    q = torch.rand(n).clamp(1e-3, 1.0)
    qr = q.clone().requires_grad_()
    ref = -torch.mean(torch.log(qr))
    ref.backward()
    qg = q.clone().to(DEV).requires_grad_()
    got = ops.neg_mean_log(qg)
    got.backward()
    close(got, ref, rtol=1e-6, what="-mean(log)")
    close(qg.grad, qr.grad, rtol=1e-5, what="-mean(log) grad")


def test_vae_reparameterisation_matches_torch():
    """sg_vae_reparam_* against `mean + torch.exp(log_variance * 0.5) * eps` (model/autoencoder.py:77-82) and its autograd."""
    from shapegan_amd import ops
    torch.manual_seed(3)
    for shape in ((32, 128), (4, 128), (128,)):
        mu, lv, eps = torch.randn(shape), torch.randn(shape) * 2, torch.randn(shape)
        g = torch.randn(shape)
        mr, lr = mu.clone().requires_grad_(), lv.clone().requires_grad_()
        ref = mr + torch.exp(lr * 0.5) * eps
        (ref * g).sum().backward()
        mg, lg = mu.clone().to(DEV).requires_grad_(), lv.clone().to(DEV).requires_grad_()
        got = ops.vae_reparam(mg, lg, eps.to(DEV))
        (got * g.to(DEV)).sum().backward()
        torch.testing.assert_close(got.detach().cpu(), ref.detach(), rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(mg.grad.cpu(), mr.grad, rtol=0, atol=0)
        torch.testing.assert_close(lg.grad.cpu(), lr.grad, rtol=1e-5, atol=1e-7)
