"""GPU tier: every HIP kernel family against the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 relative fp32 (BASELINE.json north star); integer/index work bit-exact."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import c_oracle
from oracle import torch_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"   # tests/test_cpu_twin.py re-runs these bodies on the CPU twin with DEV = "cpu"

RTOL = 1e-4


def close(a, b, rtol=RTOL, atol=None, what=""):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    if atol is None:
        atol = rtol * max(float(b.abs().mean()), 1e-30)   # relative to the tensor's typical magnitude
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol, msg=lambda m: what + ": " + m)


def close_mostly(a, b, rtol=RTOL, max_bad_frac=5e-4, what=""):
    """Gradients that pass through ReLU masks: a pre-activation within rounding of 0 can land on the other side of
    the (discontinuous) derivative in two correct implementations, so a handful of isolated elements may differ by a
    finite amount.  Require all but a tiny fraction within tolerance, and the outliers bounded by the tensor scale."""
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    atol = rtol * max(float(b.abs().mean()), 1e-30)
    bad = (a - b).abs() > (atol + rtol * b.abs())
    frac = float(bad.float().mean())
    assert frac <= max_bad_frac, "%s: %.4f%% of elements out of tolerance" % (what, 100 * frac)
    if bad.any():
        assert float((a - b).abs().max()) <= 0.5 * float(b.abs().max()), what + ": outlier larger than the tensor scale"


def dev(t):
    return t.cuda()


# ---- convolution ------------------------------------------------------------------------------------------------
CONV_CASES = [
    # N, Cin, Cout, R
    (2, 3, 5, 8), (1, 1, 4, 6), (3, 8, 1, 4), (1, 2, 2, 2),       # tiny + ragged (R=6 -> 3^3, R=2 -> 1^3)
    (2, 64, 128, 16), (2, 128, 256, 8), (4, 1, 64, 32),            # gan.Discriminator layers
    (2, 24, 48, 16), (2, 48, 96, 8),                               # autoencoder encoder
    (1, 32, 64, 32), (1, 1, 32, 64),                               # progressive D at 64^3
    (2, 70, 130, 8),                                               # non-multiple-of-tile channels
]


@pytest.mark.parametrize("N,Ci,Co,R", CONV_CASES)
def test_conv3d_fwd_dgrad_wgrad(N, Ci, Co, R):
    from shapegan_amd import ops
    torch.manual_seed(N * 1000 + Ci * 10 + Co + R)
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Co, Ci, 4, 4, 4) / (Ci * 64) ** 0.5
    b = torch.randn(Co)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.conv3d(xr, wr, br, stride=2, padding=1)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.conv3d_k4s2p1(xg, wg, bg)
    close(y, y_ref, what="fwd")
    y.backward(dev(dy))
    close(xg.grad, xr.grad, what="dgrad")
    close(wg.grad, wr.grad, what="wgrad")
    close(bg.grad, br.grad, what="bias grad")
    if N * Ci * Co * R ** 3 <= 2 * 3 * 5 * 512:
        close(y, torch.from_numpy(c_oracle.conv_fwd(x.numpy(), w.numpy(), b.numpy())), what="fwd vs C oracle")


@pytest.mark.parametrize("N,Ci,Co,R", [(2, 256, 128, 4), (2, 128, 64, 8), (3, 64, 1, 16), (2, 96, 48, 4), (2, 24, 1, 16),
                                       (1, 5, 3, 3), (9, 8, 1, 8), (10, 40, 1, 4), (70, 64, 1, 2), (2, 3, 1, 12)])
def test_conv_transpose3d(N, Ci, Co, R):
    """nn.ConvTranspose3d(k4,s2,p1) with fused LeakyReLU / Tanh epilogues (model/gan.py:13-22)."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY, ACT_NONE, ACT_TANH
    torch.manual_seed(N + Ci + Co + R)
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Ci, Co, 4, 4, 4) / (Ci * 8) ** 0.5
    b = torch.randn(Co)
    for act, fn in ((ACT_NONE, lambda t: t), (ACT_LEAKY, lambda t: F.leaky_relu(t, 0.2)), (ACT_TANH, torch.tanh)):
        xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y_ref = fn(F.conv_transpose3d(xr, wr, br, stride=2, padding=1))
        dy = torch.randn_like(y_ref)
        y_ref.backward(dy)
        xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
        y = ops.conv_transpose3d_k4s2p1(xg, wg, bg, act, 0.2)
        close(y, y_ref, what="convT fwd act%d" % act)
        y.backward(dev(dy))
        close(xg.grad, xr.grad, what="convT dgrad act%d" % act)
        close(wg.grad, wr.grad, what="convT wgrad act%d" % act)
        close(bg.grad, br.grad, what="convT bias grad act%d" % act)


def test_conv_transpose_keeps_its_weight_image_until_the_weight_changes(monkeypatch):
    """sg_conv3d_k4s2p1_dgrad_keep: the packed weight image of a ConvTranspose3d forward is reused while the weight is unchanged
    (the WGAN generator: six evaluations per update, train_wgan.py:60-84) and rebuilt after every kind of change — a flat-buffer
    optimizer's raw-pointer step (its own buffer's epoch), an in-place torch update (tensor version), a new tensor at the same
    address — but NOT after another optimizer's step."""
    from shapegan_amd import ops, optim
    flags = []
    real_get = ops._KEPT.get

    def spy(w, nbytes, shape_key, **kw):
        ws, unchanged = real_get(w, nbytes, shape_key, **kw)
        flags.append(unchanged)
        return ws, unchanged
    monkeypatch.setattr(ops._KEPT, "get", spy)
    torch.manual_seed(11)
    for Ci, Co, R in ((128, 64, 8), (256, 128, 4), (24, 12, 4)):      # halo kernel (two modes), tile-GEMM path
        x = torch.randn(4, Ci, R, R, R)
        w = torch.nn.Parameter((torch.randn(Ci, Co, 4, 4, 4) / (Ci * 8) ** 0.5).cuda())
        other = torch.nn.Parameter(torch.randn(1000).cuda())
        opt, opt_other = optim.RMSprop([w], lr=0.05), optim.RMSprop([other], lr=0.05)

        def run():
            with torch.no_grad():
                return ops.conv_transpose3d_k4s2p1(x.cuda(), w, None)

        def ref():
            return F.conv_transpose3d(x, w.detach().cpu(), None, stride=2, padding=1)
        del flags[:]
        y1, y2 = run(), run()
        assert flags == [False, True] and torch.equal(y1, y2)
        close(y1, ref(), what="convT, fresh image")
        other.grad = torch.ones_like(other)
        opt_other.step()                                  # somebody else's parameters moved: image still valid
        y3 = run()
        assert flags[-1] is True and torch.equal(y3, y1)
        w.grad = torch.randn_like(w)
        opt.step()                                        # raw-pointer update of w
        y4 = run()
        assert flags[-1] is False and not torch.equal(y4, y1)
        close(y4, ref(), what="convT after the optimizer step")
        assert torch.equal(run(), y4) and flags[-1] is True
        with torch.no_grad():
            w.mul_(0.5)                                   # in-place torch update
        y5 = run()
        assert flags[-1] is False
        close(y5, ref(), what="convT after an in-place update")
        with torch.enable_grad():                         # grad mode shares the image (generator update after a critic update)
            y6 = ops.conv_transpose3d_k4s2p1(x.cuda(), w, None)
        assert flags[-1] is True and torch.equal(y6.detach(), y5)
    # a different weight at a recycled address is not mistaken for the old one
    w1 = torch.nn.Parameter(torch.randn(128, 64, 4, 4, 4).cuda() * 0.03)
    opt1 = optim.RMSprop([w1], lr=0.05)                    # (images are kept for flat-buffer parameters only)
    x = torch.randn(2, 128, 8, 8, 8).cuda()
    with torch.no_grad():
        ops.conv_transpose3d_k4s2p1(x, w1, None)
        addr = w1.data_ptr()
        del w1, opt1
        w2 = torch.nn.Parameter(torch.randn(128, 64, 4, 4, 4).cuda() * 0.03)
        opt2 = optim.RMSprop([w2], lr=0.05)
        del flags[:]
        y = ops.conv_transpose3d_k4s2p1(x, w2, None)
    if w2.data_ptr() == addr:
        assert flags == [False]
    close(y, F.conv_transpose3d(x.cpu(), w2.detach().cpu(), None, stride=2, padding=1), what="convT, recycled address")
    del opt2


def test_a_write_through_data_is_never_served_a_stale_weight_image(monkeypatch):
    """`p.data.mul_()` / `p.data.clamp_()` (the reference's own clip_weights idiom, model/gan.py:67-69) move neither
    `tensor._version` nor a parameter epoch.  Images are therefore kept only for parameters that live in a shapegan_amd.optim flat
    buffer (every writer of those announces itself); any other weight is packed on every call — and a `.data` writer of a
    flat-buffer parameter calls ops.invalidate_weight_images()."""
    from shapegan_amd import ops, optim
    from shapegan_amd.model.gan import Discriminator
    from shapegan_amd.model.sdf_net import SDFNet
    calls = []
    real_get = ops._KEPT.get
    monkeypatch.setattr(ops._KEPT, "get", lambda *a, **kw: (calls.append(1), real_get(*a, **kw))[1])
    torch.manual_seed(5)
    x = torch.randn(4, 128, 8, 8, 8)
    # (a) a bare parameter, no flat buffer: both convolution directions see the new values at once, nothing is kept
    wt = torch.nn.Parameter((torch.randn(128, 64, 4, 4, 4) * 0.03).cuda())
    wc = torch.nn.Parameter((torch.randn(64, 128, 4, 4, 4) * 0.03).cuda())
    with torch.no_grad():
        y1 = ops.conv_transpose3d_k4s2p1(x.cuda(), wt, None)
        c1 = ops.conv3d_k4s2p1(x.cuda(), wc, None)
        wt.data.mul_(0.5)
        wc.data.mul_(0.5)
        assert wt._version == 0                                        # the write left no trace ...
        y2 = ops.conv_transpose3d_k4s2p1(x.cuda(), wt, None)
        c2 = ops.conv3d_k4s2p1(x.cuda(), wc, None)
    assert not calls
    close(y2, F.conv_transpose3d(x, wt.detach().cpu(), None, stride=2, padding=1), what="convT after p.data.mul_")   # ... and is seen
    close(c2, F.conv3d(x, wc.detach().cpu(), None, stride=2, padding=1), what="conv after p.data.mul_")
    close(y2, 0.5 * y1, what="convT halves")
    close(c2, 0.5 * c1, what="conv halves")
    # (b) the reference's clip_weights body on a critic that no flat optimizer owns (stock torch.optim in the scripts)
    d = Discriminator()
    d.use_sigmoid = False
    v = (torch.rand(8, 32, 32, 32) * 2 - 1).cuda()
    with torch.no_grad():
        d(v)
        for p in d.parameters():
            p.data.clamp_(-0.01, 0.01)
        got = d(v)
        fresh = Discriminator()
        fresh.load_state_dict({k: t.clone() for k, t in d.state_dict().items()})
        fresh.use_sigmoid = False
        assert torch.equal(got, fresh(v))
    # (c) a flat-buffer parameter: kept — a `.data` writer says so
    opt = optim.RMSprop([wt], lr=0.05)
    with torch.no_grad():
        ops.conv_transpose3d_k4s2p1(x.cuda(), wt, None)
        assert calls or DEV != "cuda"
        wt.data.mul_(2.0)
        ops.invalidate_weight_images()
        y3 = ops.conv_transpose3d_k4s2p1(x.cuda(), wt, None)
    close(y3, F.conv_transpose3d(x, wt.detach().cpu(), None, stride=2, padding=1), what="convT after invalidate_weight_images")
    del opt
    # (d) the SDFNet pack follows the same rule
    net = SDFNet()
    pts = (torch.rand(4096, 3) * 2 - 1).cuda()
    z = torch.randn(4096, 128).cuda() * 0.1
    with torch.no_grad():
        s1 = net(pts, z)
        net.layers2[6].weight.data.mul_(0.25)
        s2 = net(pts, z)
    close(torch.atanh(s2.clamp(-0.999, 0.999)).cpu() - net.layers2[6].bias.detach().cpu(),
          0.25 * (torch.atanh(s1.clamp(-0.999, 0.999)).cpu() - net.layers2[6].bias.detach().cpu()), what="SDFNet after p.data.mul_", rtol=2e-3)


@pytest.mark.parametrize("N,Co,O,act", [(16, 64, 16, 1), (17, 24, 16, 2), (128, 64, 16, 1)])
def test_conv_wgrad_through_activation(N, Co, O, act):
    """sg_conv3d_k4s2p1_wgrad_act: weight and bias gradient of act(conv(x) + b) for a one-channel input straight from dLoss/dy (the
    activation backward rides in the weight-gradient kernel) == autograd of ATen conv3d + LeakyReLU / ReLU; and the module path
    (ops.conv3d_k4s2p1 on an input that needs no gradient) takes it."""
    from shapegan_amd import ops
    torch.manual_seed(N + Co + O)
    x = torch.rand(N, 1, 2 * O, 2 * O, 2 * O) * 2 - 1
    w = (torch.randn(Co, 1, 4, 4, 4) * 0.2).requires_grad_(True)
    b = (torch.randn(Co) * 0.1).requires_grad_(True)
    fn = (lambda t: F.leaky_relu(t, 0.2)) if act == 1 else F.relu
    pre = F.conv3d(x, w, b, stride=2, padding=1)
    y_ref = fn(pre)
    dy = torch.randn_like(y_ref)
    dy[pre.detach().abs() < 1e-5] = 0      # a pre-activation within rounding of the kink may take either branch on the GPU
    y_ref.backward(dy)
    dw, db = ops.conv_wgrad_act_raw(dev(dy), dev(y_ref.detach()), dev(x), act, 0.2)
    close(dw, w.grad, what="dw through activation")
    close(db, b.grad, what="db through activation")
    if DEV == "cuda":
        wg, bg = dev(w.detach()).requires_grad_(True), dev(b.detach()).requires_grad_(True)
        y = ops.conv3d_k4s2p1(dev(x), wg, bg, act, 0.2)
        y.backward(dev(dy))
        close(wg.grad, w.grad, what="module path dw")
        close(bg.grad, b.grad, what="module path db")


def test_conv_wgrad_act_falls_back_when_scratch_exceeds_cap(monkeypatch):
    """when the fused weight-gradient + activation-backward kernel's scratch does not fit under the workspace
    cap, the first critic layer's backward takes the two-pass path (activation backward + plain weight gradient) instead of
    raising.  The cap is lowered so that the BASELINE shape triggers the refusal; both paths must give the ATen gradients."""
    from shapegan_amd import ops
    torch.manual_seed(21)
    N, Co, O = 16, 64, 16
    x = torch.rand(N, 1, 2 * O, 2 * O, 2 * O) * 2 - 1
    w = (torch.randn(Co, 1, 4, 4, 4) * 0.2).requires_grad_(True)
    b = (torch.randn(Co) * 0.1).requires_grad_(True)
    pre = F.conv3d(x, w, b, stride=2, padding=1)
    y_ref = F.leaky_relu(pre, 0.2)
    dy = torch.randn_like(y_ref)
    dy[pre.detach().abs() < 1e-5] = 0
    y_ref.backward(dy)
    calls = []
    real = ops.conv_wgrad_act_raw
    monkeypatch.setattr(ops, "conv_wgrad_act_raw", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    for cap, fused in ((ops._WGRAD_WS_CAP, True), (1 << 20, False)):
        monkeypatch.setattr(ops, "_WGRAD_WS_CAP", cap)
        del calls[:]
        wg, bg = dev(w.detach()).requires_grad_(True), dev(b.detach()).requires_grad_(True)
        y = ops.conv3d_k4s2p1(dev(x), wg, bg, 1, 0.2)
        y.backward(dev(dy))
        assert bool(calls) == fused
        close(wg.grad, w.grad, what="dw (fused path %s)" % fused)
        close(bg.grad, b.grad, what="db (fused path %s)" % fused)


def test_conv_transpose3d_to_one_channel_random_shapes():
    """ConvTranspose3d(C -> 1) through the fused tap-group kernel at random channel counts / odd plane sizes / batch sizes that
    are no multiple of the XCD count (partial position tiles, clamped channel reads, idle workgroups) == ATen."""
    import random
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_TANH
    random.seed(5)
    for it in range(14):
        N, C, R = random.randint(1, 19), random.choice((1, 2, 7, 24, 33, 63, 64)), random.choice((1, 2, 3, 5, 6, 9, 12, 13, 16))
        torch.manual_seed(it)
        x = torch.randn(N, C, R, R, R)
        w = torch.randn(C, 1, 4, 4, 4) / (C * 8) ** 0.5
        b = torch.randn(1)
        ref = torch.tanh(F.conv_transpose3d(x, w, b, stride=2, padding=1))
        got = ops.conv_transpose3d_k4s2p1(dev(x), dev(w), dev(b), ACT_TANH, 0.0)
        close(got, ref, what="convT C->1 N=%d C=%d R=%d" % (N, C, R))


def test_conv_transpose3d_to_one_channel_streaming_kernel_random_shapes():
    """The plane-streaming ConvTranspose3d(C -> 1) kernel (four workgroups per sample, one output parity pair each): through the
    input-transform entry sg_convT3d_k4s2p1_to1_pre — act_in(x * scale[c] + shift[c]) folded into the loads, as the inference-mode
    generator uses it for its last BatchNorm3d + LeakyReLU — at random channel counts, odd plane sizes (8-byte loads at 4-byte
    alignment, partial blocks), batch sizes that are no multiple of the XCD count, every output activation form; and through the
    plain entry at batch sizes that take it (>= 48 samples)."""
    import random
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH
    random.seed(7)
    for it in range(16):
        N, C, R = random.randint(1, 19), random.choice((1, 2, 7, 24, 33, 63, 64)), random.choice((1, 2, 3, 5, 6, 9, 12, 13, 16))
        in_act, in_slope = random.choice(((ACT_LEAKY, 0.2), (ACT_RELU, 0.0), (ACT_NONE, 0.0)))
        act = random.choice((ACT_TANH, ACT_NONE, ACT_SIGMOID))
        torch.manual_seed(100 + it)
        x = torch.randn(N, C, R, R, R)
        w = torch.randn(C, 1, 4, 4, 4) / (C * 8) ** 0.5
        b = torch.randn(1)
        scale, shift = torch.randn(C), torch.randn(C) * 0.3
        t = x * scale.view(1, C, 1, 1, 1) + shift.view(1, C, 1, 1, 1)
        t = F.leaky_relu(t, in_slope) if in_act == ACT_LEAKY else (F.relu(t) if in_act == ACT_RELU else t)
        ref = F.conv_transpose3d(t, w, b, stride=2, padding=1)
        ref = torch.tanh(ref) if act == ACT_TANH else (torch.sigmoid(ref) if act == ACT_SIGMOID else ref)
        assert ops.convT_to1_pre_served(dev(x), dev(w))
        got = ops.conv_transpose3d_to1_pre_raw(dev(x), dev(scale), dev(shift), in_act, in_slope, dev(w), dev(b), act, 0.0)
        close(got, ref, what="convT C->1 with input transform N=%d C=%d R=%d in_act=%d act=%d" % (N, C, R, in_act, act))
    for N, C, R in ((48, 64, 4), (50, 24, 6), (67, 64, 16), (53, 5, 3)):
        torch.manual_seed(N)
        x = torch.randn(N, C, R, R, R)
        w = torch.randn(C, 1, 4, 4, 4) / (C * 8) ** 0.5
        b = torch.randn(1)
        ref = torch.tanh(F.conv_transpose3d(x, w, b, stride=2, padding=1))
        got = ops.conv_transpose3d_k4s2p1(dev(x), dev(w), dev(b), ACT_TANH, 0.0)
        close(got, ref, what="convT C->1 (streaming kernel) N=%d C=%d R=%d" % (N, C, R))
        out = torch.full((N + 1, 1, 2 * R, 2 * R, 2 * R), 7.0)
        out = dev(out)
        with torch.no_grad():
            got2 = ops.conv_transpose3d_k4s2p1(dev(x), dev(w), dev(b), ACT_TANH, 0.0, out=out[:N])
        assert got2.data_ptr() == out.data_ptr() and torch.equal(got2, got.detach()) and bool((out[N] == 7.0).all())
    # the kernel forms of sg_convT3d_k4s2p1_to1_pre_impl (selected through the ABI; the library reads no environment variable):
    # 1 = one h parity per workgroup, 2 = the same with two plane walks per (sample, pd, ph) (the second workgroup recomputes the
    # plane in front of its half for the carried sum), 3 / 4 = both h parities per workgroup (convT_c1_stream2_kernel, the default
    # from 192 samples on) with one / two walks, 5 - 8 = all four (pd, ph) pairs per workgroup (convT_c1_all_kernel): the same sums
    # in the same order, bit for bit, at any batch size
    if DEV == "cuda":
        for N, C, R in ((9, 64, 16), (5, 24, 6), (3, 7, 4), (50, 64, 8)):
            torch.manual_seed(N + R)
            x, w, b = dev(torch.randn(N, C, R, R, R)), dev(torch.randn(C, 1, 4, 4, 4) / (C * 8) ** 0.5), dev(torch.randn(1))
            scale, shift = dev(torch.randn(C)), dev(torch.randn(C) * 0.3)
            one = ops.conv_transpose3d_to1_pre_raw(x, scale, shift, ACT_LEAKY, 0.2, w, b, ACT_TANH, 0.0, form=1)
            assert torch.equal(one, ops.conv_transpose3d_to1_pre_raw(x, scale, shift, ACT_LEAKY, 0.2, w, b, ACT_TANH, 0.0))
            for form in (2, 3, 4, 5, 6, 7, 8):     # (5 - 8: all 64 taps per workgroup, plane ranges chosen / 1 / 2 / 4 per sample)
                other = ops.conv_transpose3d_to1_pre_raw(x, scale, shift, ACT_LEAKY, 0.2, w, b, ACT_TANH, 0.0, form=form)
                assert torch.equal(one, other), "form %d differs from form 1 at N=%d C=%d R=%d" % (form, N, C, R)
        # 200 samples: the dispatch rule takes the both-parity form; directly against ATen and against form 1
        xc = torch.randn(200, 64, 16, 16, 16)
        wc, bc = torch.randn(64, 1, 4, 4, 4) / 23.0, torch.randn(1)
        x, w, b = dev(xc), dev(wc), dev(bc)
        got = ops.conv_transpose3d_k4s2p1(x, w, b, ACT_TANH, 0.0)
        close(got, torch.tanh(F.conv_transpose3d(xc, wc, bc, stride=2, padding=1)), what="convT 64->1 at 200 samples (default: both-parity form) vs ATen")
        ident_s, ident_h = dev(torch.ones(64)), dev(torch.zeros(64))
        ref1 = ops.conv_transpose3d_to1_pre_raw(x, ident_s, ident_h, ACT_NONE, 1.0, w, b, ACT_TANH, 0.0, form=1)
        ref3 = ops.conv_transpose3d_to1_pre_raw(x, ident_s, ident_h, ACT_NONE, 1.0, w, b, ACT_TANH, 0.0, form=3)
        assert torch.equal(ref1, ref3)
        close(ref3, got, what="both-parity form through the input-transform entry vs the plain entry")


def test_pack_group_rebuilds_every_stale_image_in_one_launch_and_only_then(monkeypatch):
    """gan.Discriminator registers its conv weights as a pack group (ops.register_pack_group): after an optimizer step the first
    convolution that finds its kept image stale rebuilds ALL stale images of the group (forward and input-gradient forms, shapes
    known from the previous update) with one sg_conv3d_k4s2p1_pack_images call; every later call of the update is told its image is
    in place.  Outputs and gradients must equal those of a fresh critic (no kept state) holding the same weights — after one and
    after two optimizer steps, and when the batch size changes in between."""
    from shapegan_amd import ops, optim
    from shapegan_amd import lib as L
    from shapegan_amd.model.gan import Discriminator
    if DEV != "cuda":
        pytest.skip("kept images exist on the GPU only")
    torch.manual_seed(31)
    d = Discriminator()
    d.use_sigmoid = False
    opt = optim.RMSprop(d.parameters(), lr=1e-3)
    calls = []
    real = L.load().sg_conv3d_k4s2p1_pack_images
    monkeypatch.setattr(L.load(), "sg_conv3d_k4s2p1_pack_images", lambda n, *a: (calls.append(n), real(n, *a))[1])
    for it, nb in enumerate((8, 8, 8, 4, 8)):
        x = (torch.rand(nb, 32, 32, 32) * 2 - 1).cuda().requires_grad_()
        opt.zero_grad()
        del calls[:]
        out = d(x)
        L.backward(out.mean())
        if it >= 2 and nb == 8:
            assert calls == [4], "update %d: expected one launch for the four stale images, got %r" % (it, calls)
        fresh = Discriminator()
        fresh.load_state_dict({k: v.clone() for k, v in d.state_dict().items()})
        fresh.use_sigmoid = False
        xf = x.detach().clone().requires_grad_()
        of = fresh(xf)
        of.mean().backward()
        assert torch.equal(out.detach(), of.detach()), "update %d: outputs differ from a fresh critic's" % it
        assert torch.equal(x.grad, xf.grad), "update %d: input gradient" % it
        for (k, p), (_, q) in zip(d.named_parameters(), fresh.named_parameters()):
            assert torch.equal(p.grad, q.grad), "update %d: gradient of %s" % (it, k)
        opt.step()


def test_kept_images_serve_another_batch_size(monkeypatch):
    """The LDS-halo kernels' weight images do not depend on the batch size (sg_conv3d_k4s2p1_image_layout): train_wgan.py's critic
    runs at 128 samples in its own updates and at 64 in the generator update — the images built for one serve the other, nothing
    is re-packed until the weights change.  Results must equal a fresh critic's (no kept state) at every pass."""
    from shapegan_amd import ops, optim
    from shapegan_amd import lib as L
    from shapegan_amd.model.gan import Discriminator
    if DEV != "cuda":
        pytest.skip("kept images exist on the GPU only")
    lay = lambda kind, dims: L.load().sg_conv3d_k4s2p1_image_layout(kind, (ctypes.c_int * 8)(*dims), 1 << 28)
    assert lay(0, (128, 64, 64, 64, 128, 16, 16, 16)) == lay(0, (64, 64, 64, 64, 128, 16, 16, 16)) != 0
    assert lay(1, (128, 128, 128, 128, 256, 8, 8, 8)) == lay(1, (64, 128, 128, 128, 256, 8, 8, 8)) != 0
    assert lay(0, (128, 64, 64, 64, 128, 16, 16, 16)) != lay(1, (128, 64, 64, 64, 128, 16, 16, 16))
    assert lay(0, (128, 64, 64, 64, 128, 16, 16, 16)) != lay(0, (128, 128, 128, 128, 256, 8, 8, 8))
    assert lay(0, (2, 64, 64, 64, 128, 16, 16, 16)) == 0 and lay(0, (128, 1, 1, 1, 64, 32, 32, 32)) == 0
    torch.manual_seed(32)
    d = Discriminator()
    d.use_sigmoid = False
    opt = optim.RMSprop(d.parameters(), lr=1e-3)
    flags, packs = [], []
    real_get = ops._KEPT.get
    monkeypatch.setattr(ops._KEPT, "get", lambda w, nb, key, **kw: (lambda r: (flags.append(r[1]), r)[1])(real_get(w, nb, key, **kw)))
    real_pack = L.load().sg_conv3d_k4s2p1_pack_images
    monkeypatch.setattr(L.load(), "sg_conv3d_k4s2p1_pack_images", lambda n, *a: (packs.append(n), real_pack(n, *a))[1])
    changed = True
    for it, (nb, step) in enumerate(((128, True), (64, False), (128, True), (64, False), (128, False), (64, True), (128, False))):
        x = (torch.rand(nb, 32, 32, 32) * 2 - 1).cuda().requires_grad_()
        opt.zero_grad()
        del flags[:], packs[:]
        out = d(x)
        L.backward(out.mean())
        if it >= 1 and not changed:
            assert flags and all(flags) and not packs, "pass %d (batch %d, weights unchanged): re-packed (%r, %r)" % (it, nb, flags, packs)
        if it >= 2 and changed:
            assert packs == [4], "pass %d: expected one launch for the four stale images, got %r" % (it, packs)
        fresh = Discriminator()
        fresh.load_state_dict({k: v.clone() for k, v in d.state_dict().items()})
        fresh.use_sigmoid = False
        xf = x.detach().clone().requires_grad_()
        of = fresh(xf)
        of.mean().backward()
        assert torch.equal(out.detach(), of.detach()), "pass %d: outputs differ from a fresh critic's" % it
        assert torch.equal(x.grad, xf.grad), "pass %d: input gradient" % it
        for (k, p), (_, q) in zip(d.named_parameters(), fresh.named_parameters()):
            assert torch.equal(p.grad, q.grad), "pass %d: gradient of %s" % (it, k)
        changed = step
        if step:
            opt.step()


@pytest.mark.parametrize("N", [128, 64, 37])
def test_producer_written_dy_images_equal_the_packing_pass(N):
    """sg_act_bwd_rowsum_pack8 (8^3 grids) and sg_head_dot_bwd (4^3 grids) write the incoming gradient in the LDS-halo
    weight-gradient kernel's fragment order themselves; sg_conv3d_k4s2p1_wgrad_prepacked then skips pack_wgrad_dy(4)_kernel.  The
    weight gradients must be BIT-equal to sg_conv3d_k4s2p1_wgrad on the same gradient (same kernel, same image), dz / row sums /
    bias gradients equal to the unfused ops, at the critic's shapes (128 / 64 samples) and a ragged batch."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    if DEV != "cuda":
        pytest.skip("the image is a property of the HIP kernels")
    torch.manual_seed(N)
    # layer 2 of the critic: y [N,128,8^3] = lrelu(conv(x [N,64,16^3]))
    x = torch.randn(N, 64, 16, 16, 16, device="cuda")
    y = torch.randn(N, 128, 8, 8, 8, device="cuda")
    gy = torch.randn(N, 128, 8, 8, 8, device="cuda")
    image = ops._wgrad_dy_image(N, 64, 128, 8, x.device)
    if N >= 64:
        assert image is not None
    if image is not None:
        gz_ref, gb_ref = ops.act_bwd_rowsum_raw(y, gy, ACT_LEAKY, 0.2)
        dw_ref = ops.conv_wgrad_raw(gz_ref, x, 64)
        gz, gb = ops.act_bwd_rowsum_pack8_raw(y, gy, ACT_LEAKY, 0.2, image[0], image[2])
        dw = ops.conv_wgrad_prepacked_raw(gz, x, 64, image[0])
        assert torch.equal(gz, gz_ref) and torch.equal(dw, dw_ref)
        close(gb, gb_ref, rtol=1e-5, what="bias gradient")
    # layer 3 + head: z [N,256,4^3] = conv(x8 [N,128,8^3]); gz from sg_head_dot_bwd
    x8 = torch.randn(N, 128, 8, 8, 8, device="cuda")
    z = torch.randn(N, 256, 4, 4, 4, device="cuda")
    wh = torch.randn(1, 256, 4, 4, 4, device="cuda") * 0.05
    g = torch.randn(N, device="cuda")
    lib = ops.L.load()
    image = ops._wgrad_dy_image(N, 128, 256, 4, x8.device)
    if N >= 64:
        assert image is not None
    if image is not None:
        gz0, gz1 = torch.empty_like(z), torch.empty_like(z)
        ops.check(lib.sg_head_dot_bwd(ops.ptr(z), ops.ptr(wh), ops.ptr(g), ops.ptr(gz0), None, None, None, None, N, 256, 64, 1, 0.2,
                                      ops.stream()), "head_dot_bwd")
        dw_ref = ops.conv_wgrad_raw(gz0, x8, 128)
        ops.check(lib.sg_head_dot_bwd(ops.ptr(z), ops.ptr(wh), ops.ptr(g), ops.ptr(gz1), None, None, None, ops.ptr(image[0]), N, 256, 64,
                                      1, 0.2, ops.stream()), "head_dot_bwd")
        dw = ops.conv_wgrad_prepacked_raw(gz1, x8, 128, image[0])
        assert torch.equal(gz0, gz1) and torch.equal(dw, dw_ref)


def test_conv_from_sdf_zero_channels():
    """First progressive stage: conv over [x, 0, ..., 0] == conv over channel 0 only; dW of the zero channels is 0."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(0)
    x = torch.randn(2, 16, 16, 16)
    w = torch.randn(64, 32, 4, 4, 4) * 0.05
    b = torch.randn(64)
    wr = w.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    padded = torch.cat((xr.reshape(2, 1, 16, 16, 16), torch.zeros(2, 31, 16, 16, 16)), 1)   # from_SDF, iteration 2
    y_ref = F.leaky_relu(F.conv3d(padded, wr, b, stride=2, padding=1), 0.2)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, wg = dev(x).requires_grad_(True), dev(w).requires_grad_(True)
    y = ops.conv3d_k4s2p1(xg.reshape(2, 1, 16, 16, 16), wg, dev(b), ACT_LEAKY, 0.2)
    close(y, y_ref)
    y.backward(dev(dy))
    close(xg.grad, xr.grad)
    close(wg.grad, wr.grad)
    assert float(wg.grad[:, 1:].abs().sum()) == 0.0


def test_conv_linearity_and_adjoint_full_size():
    """Size-independent properties at the BASELINE config-2 size (B=64): linearity of the forward and
    <conv(x), y> == <x, conv^T(y)> between the fwd and dgrad kernels."""
    from shapegan_amd import ops
    torch.manual_seed(1)
    w = (torch.randn(128, 64, 4, 4, 4) / 64).cuda()
    x1, x2 = torch.randn(64, 64, 16, 16, 16, device=DEV), torch.randn(64, 64, 16, 16, 16, device=DEV)
    y1, y2 = ops.conv_fwd_raw(x1, w, None), ops.conv_fwd_raw(x2, w, None)
    y12 = ops.conv_fwd_raw(0.5 * x1 - 2.0 * x2, w, None)
    close(y12, 0.5 * y1 - 2.0 * y2, rtol=2e-4)
    g = torch.randn_like(y1)
    dx = ops.conv_dgrad_raw(g, w, None, 64)
    lhs, rhs = (y1.double() * g.double()).sum().item(), (x1.double() * dx.double()).sum().item()
    assert abs(lhs - rhs) <= 1e-4 * max(abs(lhs), (y1.double() * g.double()).abs().sum().item() * 1e-3)
    dw = ops.conv_wgrad_raw(g, x1, 64)
    lhs_w = (dw.double() * w.double()).sum().item()
    assert abs(lhs - lhs_w) <= 1e-4 * max(abs(lhs), (y1.double() * g.double()).abs().sum().item() * 1e-3)


# ---- GEMM / Linear / 1^3<->4^3 convs --------------------------------------------------------------------------------
@pytest.mark.parametrize("M,N,K", [(64, 128, 256), (4, 128, 256), (16, 128, 16384), (64, 1, 16384), (5, 7, 3),
                                   (200, 300, 130), (64, 16384, 128),
                                   # shapes that take gemm128_kernel (round 5): the PointNet GAN's Linear layers at many points —
                                   # forward k-major x k-major, input gradient k-major x row-major, weight gradient row-major x
                                   # row-major with a K split; ragged row counts, a K that ends inside a stage
                                   (20000, 256, 256), (16388, 384, 320), (40004, 128, 100),
                                   # more than 512 tiles: the persistent form (gemm128p_kernel) for forward and input gradient
                                   (70004, 256, 128)])
def test_linear_fwd_bwd(M, N, K):
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(M + N + K)
    x, w, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = F.leaky_relu(F.linear(xr, wr, br), 0.2)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, wg, bg = dev(x).requires_grad_(True), dev(w).requires_grad_(True), dev(b).requires_grad_(True)
    y = ops.linear(xg, wg, bg, ACT_LEAKY, 0.2)
    close(y, y_ref, what="linear fwd")
    y.backward(dev(dy))
    close(xg.grad, xr.grad, what="linear dx")
    close(wg.grad, wr.grad, what="linear dw")
    close(bg.grad, br.grad, what="linear db")


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K", [(70004, 256, 128), (66000, 384, 64), (131072, 128, 320), (70004, 256, 100)])     # (K = 100: a tile's last stage ends inside a 16-byte piece)
def test_gemm128_persistent_all_layouts(M, N, K):
    """gemm128p_kernel (a workgroup walks tiles as one stream of stages, > 512 tiles, even stage count) in its three operand layouts,
    with a per-column bias and LeakyReLU in the epilogue, against torch in fp64; ragged last row tile."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(M + N + K)
    a = torch.randn(M, K)
    b = torch.randn(N, K) / K ** 0.5
    bias = torch.randn(N)
    ref = F.leaky_relu(a.double() @ b.double().t() + bias.double(), 0.2).float()
    got = ops.gemm_raw(dev(a), False, dev(b), True, bias_j=dev(bias), act=ACT_LEAKY, slope=0.2)                 # k-major x k-major
    close(got, ref, what="A [M,K] x B [N,K]^T")
    got = ops.gemm_raw(dev(a), False, dev(b.t().contiguous()), False, bias_j=dev(bias), act=ACT_LEAKY, slope=0.2)   # k-major x row-major
    close(got, ref, what="A [M,K] x B [K,N]")
    got = ops.gemm_raw(dev(a.t().contiguous()), True, dev(b.t().contiguous()), False, bias_j=dev(bias), act=ACT_LEAKY, slope=0.2)
    close(got, ref, what="A [K,M]^T x B [K,N]")                                                                   # row-major x row-major


def test_gemm_double_backward():
    from shapegan_amd import ops
    torch.manual_seed(2)
    a, b = torch.randn(6, 10, dtype=torch.float32), torch.randn(7, 10, dtype=torch.float32)
    for ta, tb in ((False, True), (False, False), (True, False), (True, True)):
        aa = (a.t() if ta else a).contiguous()
        bb = (b if tb else b.t()).contiguous()
        ar, br = aa.clone().requires_grad_(True), bb.clone().requires_grad_(True)
        ag, bg = dev(aa).requires_grad_(True), dev(bb).requires_grad_(True)

        def f(p, q, mm):
            c = mm(p, q)
            (g,) = torch.autograd.grad(c.pow(2).sum(), p, create_graph=True)
            return (g.pow(2).sum() + c.sum())

        ref = f(ar, br, lambda p, q: (p.t() if ta else p) @ (q.t() if tb else q))
        ref.backward()
        got = f(ag, bg, lambda p, q: ops.Gemm.apply(p, q, ta, tb))
        got.backward()
        close(got, ref, what="gemm dbl value %s%s" % (ta, tb))
        close(ag.grad, ar.grad, what="gemm dbl ga")
        close(bg.grad, br.grad, what="gemm dbl gb")


# ---- BatchNorm --------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,C,S", [(64, 256, 64), (8, 128, 512), (4, 64, 4096), (4, 24, 4096), (4, 256, 1), (3, 7, 27)])
def test_batchnorm_train_fwd_bwd(N, C, S):
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(N + C + S)
    shape = (N, C, S) if S > 1 else (N, C)
    x = torch.randn(shape) * 1.7 + 3.0
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C)
    rm, rv = torch.randn(C), torch.rand(C) + 0.5
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_r, rv_r = rm.clone(), rv.clone()
    y_ref = F.leaky_relu(F.batch_norm(xr, rm_r, rv_r, gr, br, True, 0.1, 1e-5), 0.2)
    dy = torch.randn_like(y_ref)
    y_ref.backward(dy)
    xg, gg, bg = dev(x).requires_grad_(True), dev(gamma).requires_grad_(True), dev(beta).requires_grad_(True)
    rm_g, rv_g, nbt = dev(rm), dev(rv), torch.zeros((), dtype=torch.long, device=DEV)
    y = ops.BatchNormAct.apply(xg, gg, bg, rm_g, rv_g, nbt, True, 1e-5, 0.1, ACT_LEAKY, 0.2)
    close(y, y_ref, what="bn fwd")
    close(rm_g, rm_r, what="running_mean")
    close(rv_g, rv_r, what="running_var")
    assert int(nbt.item()) == 1
    y.backward(dev(dy))
    tol = 1e-3 if N * S <= 8 else RTOL     # 4 values per channel: torch's own CPU/GPU BN differ at this level
    close(xg.grad, xr.grad, rtol=tol, what="bn dx")
    close(gg.grad, gr.grad, rtol=tol, what="bn dgamma")
    close(bg.grad, br.grad, rtol=tol, what="bn dbeta")
    # eval mode uses the running statistics
    y_eval_ref = F.batch_norm(x, rm_r, rv_r, gamma, beta, False, 0.1, 1e-5)
    y_eval = ops.BatchNormAct.apply(dev(x), dev(gamma), dev(beta), rm_g, rv_g, None, False, 1e-5, 0.1, 0, 0.0)
    close(y_eval, y_eval_ref, what="bn eval")


# ---- activations, optimizers, rows --------------------------------------------------------------------------------------
def test_activations():
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY, ACT_RELU, ACT_SIGMOID, ACT_TANH
    torch.manual_seed(3)
    x = torch.randn(1001) * 2
    for act, fn in ((ACT_LEAKY, lambda t: F.leaky_relu(t, 0.2)), (ACT_RELU, F.relu), (ACT_TANH, torch.tanh),
                    (ACT_SIGMOID, torch.sigmoid)):
        xr, xg = x.clone().requires_grad_(True), dev(x).requires_grad_(True)
        yr, yg = fn(xr), ops.Act.apply(xg, act, 0.2)
        close(yg, yr, atol=1e-6)
        dy = torch.randn(1001)
        yr.backward(dy)
        yg.backward(dev(dy))
        close(xg.grad, xr.grad, atol=1e-6)


def test_rmsprop_adam_clamp_match_torch():
    from shapegan_amd import optim
    torch.manual_seed(4)
    shapes = [(64, 1, 4, 4, 4), (64,), (33, 7), (5,)]
    for kind in ("rmsprop", "adam", "rmsprop_clip"):
        ref_p = [torch.randn(s).requires_grad_(True) for s in shapes]
        gpu_p = [torch.nn.Parameter(p.detach().clone().cuda()) for p in ref_p]
        if kind == "adam":
            ro, go = torch.optim.Adam(ref_p, lr=1e-3), optim.Adam(gpu_p, lr=1e-3)
        else:
            ro = torch.optim.RMSprop(ref_p, lr=1e-3)
            go = optim.RMSprop(gpu_p, lr=1e-3, clip=0.5 if kind == "rmsprop_clip" else 0.0)
        for step in range(4):
            go.zero_grad()
            for i, (rp, gp) in enumerate(zip(ref_p, gpu_p)):
                g = torch.randn(rp.shape)
                rp.grad = g.clone()
                if step == 2 and i == 3:
                    rp.grad = None          # a parameter without gradient is skipped (unused progressive stage)
                    gp.grad = None
                else:
                    gp.grad.copy_(g.cuda()) if gp.grad is not None else setattr(gp, "grad", g.cuda())
            ro.step()
            go.step()
            if kind == "rmsprop_clip":
                with torch.no_grad():
                    for rp in ref_p:
                        rp.clamp_(-0.5, 0.5)
        for rp, gp in zip(ref_p, gpu_p):
            close(gp.data, rp.data, rtol=1e-5, atol=1e-6, what=kind)


def test_gather_scatter_rows_bit_exact():
    from shapegan_amd import ops
    torch.manual_seed(5)
    table = torch.randn(37, 128)
    idx = torch.randint(0, 37, (1000,))
    tg = dev(table).requires_grad_(True)
    rows = ops.gather_rows(tg, dev(idx))
    assert torch.equal(rows.cpu(), table[idx])                       # index work: bit-exact
    g = torch.randn(1000, 128)
    rows.backward(dev(g))
    ref = torch.zeros_like(table).index_add_(0, idx, g)
    close(tg.grad, ref, rtol=1e-5, atol=1e-5)                        # float atomics: order differs


def test_mean_reduction():
    from shapegan_amd import ops
    torch.manual_seed(6)
    x = torch.randn(64) + 3
    xg = dev(x).requires_grad_(True)
    m = ops.mean(xg)
    close(m, x.mean(), rtol=1e-6)
    m.backward()
    close(xg.grad, torch.full((64,), 1 / 64.0), rtol=1e-6)
    big = torch.randn(3_000_001)
    close(ops.mean(dev(big)), big.double().mean().float(), rtol=1e-5, atol=1e-7)


# ---- SDFNet fused MLP ------------------------------------------------------------------------------------------------------
def _sdf_state(seed, latent=128):
    from shapegan_amd.model.sdf_net import SDFNet
    torch.manual_seed(seed)
    return SDFNet(latent_code_size=latent)


@pytest.mark.parametrize("N,latent", [(1, 128), (63, 128), (64, 128), (1000, 128), (20000, 128), (777, 256), (130, 16),
                                      (33100, 32)])   # 33100: the backward's last round is cut into 32-point tiles
def test_sdfnet_points_mode(N, latent):
    net = _sdf_state(8, latent)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.manual_seed(N)
    pts, lat = torch.rand(N, 3) * 2 - 1, torch.randn(N, latent) * 0.5
    P = O.clone_state(sd)
    pr, lr_ = pts.clone().requires_grad_(True), lat.clone().requires_grad_(True)
    out_ref = O.sdfnet_forward(P, pr, lr_)
    out_ref = out_ref.reshape(-1)
    pg, lg = dev(pts).requires_grad_(True), dev(lat).requires_grad_(True)
    out = net(pg, lg).reshape(-1)
    close(out, out_ref, atol=2e-6, what="sdfnet fwd")
    dy = torch.randn(N)
    fragile = O.sdfnet_min_preactivation(P, pts, lat) < 1e-6      # points sitting on a ReLU kink (see oracle docstring)
    assert N < 64 or float(fragile.float().mean()) < 0.1
    dy[fragile] = 0
    out_ref.backward(dy)
    out.backward(dev(dy))
    close(pg.grad, pr.grad, what="d points")
    close(lg.grad, lr_.grad, what="d latent")
    for k, p in net.named_parameters():
        close(p.grad, P[k].grad, rtol=2e-4, what="grad " + k)


@pytest.mark.parametrize("S,pps", [(1, 128), (3, 512), (2, 4096), (1, 1000), (1, 37), (5, 6656)])   # 5 x 6656: 512 tiles + 16 small ones
def test_sdfnet_shapes_mode(S, pps):
    """Per-shape latents (folded biases) == the reference's tiled-latent forward."""
    net = _sdf_state(9)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.manual_seed(S * pps)
    pts, z = torch.rand(S * pps, 3) * 2 - 1, torch.randn(S, 128)
    P = O.clone_state(sd)
    pr, zr = pts.clone().requires_grad_(True), z.clone().requires_grad_(True)
    out_ref = O.sdfnet_forward(P, pr, O.tile_latents(zr, pps)).reshape(-1)
    pg, zg = dev(pts).requires_grad_(True), dev(z).requires_grad_(True)
    out = net.forward_shapes(pg, zg, pps)
    close(out, out_ref, atol=2e-6, what="shapes fwd")
    dy = torch.randn(S * pps)
    fragile = O.sdfnet_min_preactivation(P, pts, O.tile_latents(z, pps)) < 1e-6
    assert float(fragile.float().mean()) < 0.1
    dy[fragile] = 0
    out_ref.backward(dy)
    out.backward(dev(dy))
    close(pg.grad, pr.grad, what="d points")
    close(zg.grad, zr.grad, rtol=2e-4, what="d z")
    for k, p in net.named_parameters():
        close(p.grad, P[k].grad, rtol=2e-4, what="grad " + k)


def test_sdfnet_chairs_known_answers(golden_sdf, chairs_state):
    """HIP path on the reference's shipped checkpoint reproduces its known answers."""
    from shapegan_amd.model.sdf_net import SDFNet
    from shapegan_amd.util import get_voxel_coordinates
    net = SDFNet()
    net.load_state_dict(chairs_state)
    pts = torch.tensor(get_voxel_coordinates(32)).cuda()
    z = torch.from_numpy(golden_sdf["z"]).cuda()
    with torch.no_grad():
        a = net(pts, z.repeat(32768, 1)).cpu()
        b = net.forward_shapes(pts, z.reshape(1, -1), 32768).cpu()
        c = net.evaluate_in_batches(pts, z, batch_size=10000)
    for out in (a, b, c):
        assert abs(out.mean().item() - 0.072827) < 5e-6 and int((out < 0).sum()) == 4187
        assert abs(out[0].item() - 0.098810) < 5e-6 and abs(out[16912].item() - 0.086498) < 5e-6
        np.testing.assert_allclose(out[:4096].numpy(), golden_sdf["chairs/out_head"], rtol=1e-4, atol=2e-6)
    voxels = net.get_voxels(z, 32)                                   # sphere mask scatter: index work is bit-exact
    mask = np.linalg.norm(get_voxel_coordinates(32), axis=1) < 1.1
    assert voxels.shape == (32, 32, 32) and np.all(voxels.reshape(-1)[~mask] == 1.0)
    np.testing.assert_allclose(voxels.reshape(-1)[mask], a.numpy()[mask], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("N,Ci,Co,R", [(2, 64, 128, 16), (1, 8, 32, 16), (2, 24, 48, 16), (1, 32, 64, 32), (3, 16, 96, 16),
                                       (1, 128, 256, 16), (1, 12, 40, 16), (2, 4 * 17, 130, 16),
                                       (3, 128, 256, 8), (2, 16, 32, 8), (5, 24, 100, 8), (1, 40, 64, 8)])
def test_conv_fwd_halo_kernel(N, Ci, Co, R):
    """The LDS-halo forward (forced, whatever the grid size) == the gather kernel == the oracle."""
    from shapegan_amd import ops
    torch.manual_seed(N + Ci + Co + R)
    x = torch.randn(N, Ci, R, R, R)
    w = torch.randn(Co, Ci, 4, 4, 4) / (Ci * 64) ** 0.5
    b = torch.randn(Co)
    ref = F.leaky_relu(F.conv3d(x, w, b, stride=2, padding=1), 0.2)
    y_halo = ops.conv_fwd_impl_raw(dev(x), dev(w), dev(b), 1, 0.2, impl=1)
    y_gather = ops.conv_fwd_impl_raw(dev(x), dev(w), dev(b), 1, 0.2, impl=0)
    close(y_halo, ref, what="halo vs oracle")
    close(y_gather, ref, what="gather vs oracle")
    if R // 2 >= 8:          # both tile heights of the 8x8-position kernel (64-row tiles: chosen for small grids; 128: debug bit 7)
        close(ops.conv_fwd_impl_raw(dev(x), dev(w), dev(b), 1, 0.2, impl=1, debug=48), ref, what="halo, 64-row tiles")
        close(ops.conv_fwd_impl_raw(dev(x), dev(w), dev(b), 1, 0.2, impl=1, debug=128), ref, what="halo, 128-row tiles")


@pytest.mark.parametrize("N,Ci,Co,O", [(2, 64, 128, 8), (1, 32, 16, 8), (1, 64, 32, 16), (2, 128, 64, 8), (1, 40, 48, 8),
                                       (4, 128, 256, 4), (3, 96, 32, 4), (1, 64, 16, 4), (2, 32, 64, 16), (3, 32, 64, 4)])
def test_conv_dgrad_halo_kernel(N, Ci, Co, O):
    """The LDS-halo dgrad / ConvTranspose3d forward (forced) == ATen conv_transpose3d (Ci = 32: the 32-row form, both grid modes,
    single and paired parity stores)."""
    from shapegan_amd import ops
    from shapegan_amd.lib import ACT_LEAKY
    torch.manual_seed(N + Ci + Co + O)
    dy = torch.randn(N, Co, O, O, O)
    w = torch.randn(Co, Ci, 4, 4, 4) / (Co * 8) ** 0.5
    b = torch.randn(Ci)
    ref = F.leaky_relu(F.conv_transpose3d(dy, w, b, stride=2, padding=1), 0.2)
    got = ops.conv_dgrad_halo_raw(dev(dy), dev(w), dev(b), Ci, ACT_LEAKY, 0.2)
    close(got, ref, what="dgrad halo vs oracle")
    if (Co // 16) % 2 == 0:   # several output parities per workgroup (the large-grid configuration), forced
        for ppw in (2, 4, 8):
            got = ops.conv_dgrad_halo_raw(dev(dy), dev(w), dev(b), Ci, ACT_LEAKY, 0.2, impl=1 + 4 * ppw)
            close(got, ref, what="dgrad halo, %d parities per workgroup" % ppw)


@pytest.mark.parametrize("S,N", [(5, 700), (64, 20000), (3, 129), (300, 1000), (7, 33100)])
def test_sdfnet_segments_mode(S, N):
    """Ragged per-shape latents (auto-decoder batches sorted by shape) == reference forward on gathered latents,
    including the dense latent-table gradient and shapes that receive no point at all."""
    net = _sdf_state(10)
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    torch.manual_seed(S + N)
    table = torch.randn(S, 128) * 0.5
    sid = torch.sort(torch.randint(0, S, (N,)))[0]
    if S == 300:
        sid = torch.sort(torch.randint(0, S // 2, (N,)) * 2)[0]          # odd shapes are absent
    pts = torch.rand(N, 3) * 2 - 1
    counts = torch.bincount(sid, minlength=S)
    seg_off = torch.zeros(S + 1, dtype=torch.int64)
    seg_off[1:] = torch.cumsum(counts, 0)
    P = O.clone_state(sd)
    tr = table.clone().requires_grad_(True)
    out_ref = O.sdfnet_forward(P, pts, tr[sid]).reshape(-1)
    tg = dev(table).requires_grad_(True)
    out = net.forward_segments(dev(pts), tg, dev(sid).int(), dev(seg_off))
    close(out, out_ref, atol=2e-6, what="segments fwd")
    dy = torch.randn(N)
    fragile = O.sdfnet_min_preactivation(P, pts, table[sid]) < 1e-6
    dy[fragile] = 0
    out_ref.backward(dy)
    out.backward(dev(dy))
    close(tg.grad, tr.grad, rtol=2e-4, what="d latent table")
    for k, p in net.named_parameters():
        close(p.grad, P[k].grad, rtol=2e-4, what="grad " + k)


@pytest.mark.parametrize("S,pc,N", [(64, 2000, 200000), (5, 40, 700), (3, 1000, 1), (300, 7, 1000), (1, 513, 1025),
                                    (4097, 3, 9000)])
def test_sdf_batch_sort(S, pc, N):
    """The one-pass batch assembly == a stable sort of the batch on indices // pointcloud_size followed by the reference's
    gathers (train_sdf_autodecoder.py:78-85); bit-exact, including shapes that receive no entry."""
    from shapegan_amd import ops
    torch.manual_seed(S + N)
    points = torch.rand(S * pc, 3) * 2 - 1
    sdf = torch.rand(S * pc) * 0.2 - 0.1
    idx = torch.randint(0, S * pc, (N,))
    if S == 300:
        idx = (torch.randint(0, S // 2, (N,)) * 2) * pc + torch.randint(0, pc, (N,))      # odd shapes are absent
    shape = torch.div(idx, pc, rounding_mode="floor")
    order = torch.sort(shape, stable=True)[1]
    bp, bs, sid, seg_off, counts = ops.sdf_batch_sort(dev(idx), pc, S, dev(points), dev(sdf))
    assert torch.equal(bp.cpu(), points[idx[order]])
    assert torch.equal(bs.cpu(), sdf[idx[order]])
    assert sid.dtype == torch.int32 and torch.equal(sid.cpu().long(), shape[order])
    cnt = torch.bincount(shape, minlength=S)
    assert torch.equal(counts.cpu(), cnt.float())
    assert seg_off.dtype == torch.int64 and torch.equal(seg_off.cpu()[1:], torch.cumsum(cnt, 0)) and int(seg_off[0]) == 0
    ops.check_batch_indices()
    bad = idx.clone()
    bad[N // 2] = S * pc + 5
    ops.sdf_batch_sort(dev(bad), pc, S, dev(points), dev(sdf))
    with pytest.raises(IndexError):
        ops.check_batch_indices()
    ops.check_batch_indices()          # the flag was cleared


@pytest.mark.parametrize("N,Ci,Co,O", [(2, 64, 128, 8), (1, 3, 32, 8), (3, 16, 40, 8), (1, 8, 160, 16), (5, 2, 64, 8),
                                       (6, 128, 256, 4), (1, 3, 40, 4), (9, 16, 130, 4), (4, 2, 32, 4)])
def test_conv_wgrad_halo_kernel(N, Ci, Co, O):
    """The LDS-halo weight gradient (forced) == autograd of ATen conv3d."""
    from shapegan_amd import ops
    torch.manual_seed(N + Ci + Co + O)
    x = torch.randn(N, Ci, 2 * O, 2 * O, 2 * O)
    dy = torch.randn(N, Co, O, O, O)
    w = torch.zeros(Co, Ci, 4, 4, 4, requires_grad=True)
    F.conv3d(x, w, None, stride=2, padding=1).backward(dy)
    got = ops.conv_wgrad_halo_raw(dev(dy), dev(x), Ci)
    close(got, w.grad, what="wgrad halo vs oracle")


# ---- input pipeline ---------------------------------------------------------------------------
@pytest.mark.gpu
def test_voxel_prepare_bit_exact(golden_steps_f2):
    """sg_voxel_prepare == the reference's CPU clamp_ + /= (datasets.py:19-22) bit for bit, NaN / inf included."""
    from shapegan_amd import ops
    g = golden_steps_f2
    raw = torch.from_numpy(g["vox/raw"]).cuda()
    got = ops.voxel_prepare(raw.clone(), 0.1, 0.1).cpu().numpy()
    assert np.array_equal(got, g["vox/rescaled"], equal_nan=True)
    got = ops.voxel_prepare(raw.clone(), 0.1, 0.0).cpu().numpy()
    assert np.array_equal(got, g["vox/clamped"], equal_nan=True)
    # full-size property: a 64^3 batch against torch's own CPU arithmetic
    torch.manual_seed(0)
    big = torch.rand(16, 64, 64, 64) * 0.5 - 0.25
    want = big.clone().clamp_(-0.1, 0.1)
    want /= 0.1
    out = torch.empty_like(big, device=DEV)
    ops.voxel_prepare(big.cuda(), 0.1, 0.1, out=out)
    assert torch.equal(out.cpu(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["resident", "stream"])
def test_voxel_batches_equal_dataloader(tmp_path, mode):
    """Resident (HBM gather) and streaming (pinned double buffer) batch paths deliver exactly the batches
    DataLoader(reference-style VoxelDataset, shuffle=True) delivers under the same seed: order, ragged last batch,
    values bit for bit."""
    from shapegan_amd.datasets import VoxelDataset
    rng = np.random.RandomState(3)
    for i in range(21):
        np.save(str(tmp_path / ("m%02d.npy" % i)), (rng.rand(8, 8, 8).astype(np.float32) * 0.6 - 0.3))
    ds = VoxelDataset.glob(str(tmp_path) + "/**.npy")
    torch.manual_seed(77)
    want = [b for b in torch.utils.data.DataLoader(ds, shuffle=True, batch_size=4)]
    torch.manual_seed(77)
    got = list(ds.resident().batches(4)) if mode == "resident" else list(ds.stream(4))
    assert [tuple(b.shape) for b in got] == [tuple(b.shape) for b in want] and got[-1].shape[0] == 1
    for a, b in zip(got, want):
        assert a.is_cuda and torch.equal(a.cpu(), b)
    torch.manual_seed(78)
    dropped = list(ds.resident().batches(4, drop_last=True))
    assert len(dropped) == 5 and all(b.shape[0] == 4 for b in dropped)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["resident", "stream"])
def test_loaders_deliver_into_the_wgan_trainers_slots(tmp_path, mode):
    """`WGANTrainer.real_slots` is where the step reads its real batches without a device copy — and the product's
    loaders can deliver there: `ResidentVoxels.batches(into=slots)` gathers each batch straight into a slot, `VoxelStream(into=)`
    lands its async H2D copy there.  Batches, order and values equal DataLoader's; every full batch IS its slot (same storage);
    two 5 + 1 units fed this way leave the same parameters, bit for bit, as units fed ordinary tensors of the same batches."""
    from shapegan_amd.datasets import VoxelDataset
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.train_steps import WGANTrainer
    rng = np.random.RandomState(5)
    B = 4
    for i in range(10 * B + 1):                                  # two units of five batches + a short batch of one
        np.save(str(tmp_path / ("m%02d.npy" % i)), (rng.rand(32, 32, 32).astype(np.float32) * 0.6 - 0.3))
    ds = VoxelDataset.glob(str(tmp_path) + "/**.npy")
    torch.manual_seed(77)
    want = [b for b in torch.utils.data.DataLoader(ds, shuffle=True, batch_size=B)]
    gen = torch.Generator().manual_seed(3)
    lat = [(list(torch.randn(5, B, 128, generator=gen).cuda().unbind(0)), torch.randn(B, 128, generator=gen).cuda()) for _ in range(2)]

    def run(in_place):
        torch.manual_seed(80)
        g, c = Generator(), Discriminator()
        tr = WGANTrainer(g, c)
        slots = tr.real_slots(B, device="cuda") if in_place else None
        torch.manual_seed(77)
        it = iter(ds.resident().batches(B, into=slots) if mode == "resident" else ds.stream(B, into=slots))
        for u in range(2):
            reals = [next(it) for _ in range(5)]
            for k, r in enumerate(reals):
                assert torch.equal(r.reshape(B, 32, 32, 32).cpu(), want[5 * u + k])
                if in_place:
                    assert r.data_ptr() == slots[k].data_ptr(), "batch %d was not delivered into its slot" % k
            tr.step(reals, lat[u][0], lat[u][1])
        last = next(it)                                           # the short batch: a tensor of its own
        assert last.shape[0] == 1 and torch.equal(last.reshape(1, 32, 32, 32).cpu(), want[10])
        torch.cuda.synchronize()
        return {k: v.detach().cpu().clone() for m in (g, c) for k, v in m.state_dict().items()}
    a, b = run(True), run(False)
    for k in a:
        assert torch.equal(a[k], b[k]), k


# ---- PointNet-discriminator GAN family ---------------------------------------------------------
@pytest.mark.parametrize("R,C,rps,tail,act", [(7, 256, 7, 0, 2), (300, 256, 100, 3, 2), (130, 64, 13, 0, 0), (64, 200, 64, 0, 2),
                                              (4096, 256, 1024, 3, 2), (1000, 512, 250, 0, 2), (33, 260, 11, 0, 2), (50, 130, 10, 0, 0),
                                              (8192, 256, 2048, 0, 2)])
def test_layernorm_act(R, C, rps, tail, act):
    """y = act(LN(x + zrow[r // rps])) (+ tail columns) and its backward vs torch fp64."""
    from shapegan_amd import ops
    torch.manual_seed(R + C)
    B = R // rps
    x = torch.randn(R, C) * 2 + 0.5
    zb = torch.randn(B, C)
    gamma, beta = torch.rand(C) + 0.5, torch.randn(C)
    t = torch.randn(R, tail) if tail else None
    w = torch.randn(R, C + tail)

    def ref(dt):
        xs, zs, g, b = (v.to(dt).requires_grad_(True) for v in (x, zb, gamma, beta))
        y = F.layer_norm(xs + zs.repeat_interleave(rps, 0), (C,), g, b, 1e-5)
        y = F.relu(y) if act == 2 else y
        if tail:
            y = torch.cat([y, t.to(dt)], 1)
        (y * w.to(dt)).sum().backward()
        return y.detach(), xs.grad, zs.grad, g.grad, b.grad
    want = ref(torch.float64)
    xs, zs, g, b = (v.cuda().requires_grad_(True) for v in (x, zb, gamma, beta))
    y = ops.layernorm_act(xs, zs, rps, g, b, 1e-5, act, None if t is None else t.cuda())
    (y * w.cuda()).sum().backward()
    got = (y.detach(), xs.grad, zs.grad, g.grad, b.grad)
    for name, a, e in zip(("y", "dx", "dz", "dgamma", "dbeta"), got, want):
        scale = float(e.abs().max()) + 1e-12
        err = float((a.double().cpu() - e).abs().max()) / scale
        assert err < 2e-5, (name, err)


@pytest.mark.parametrize("B,P,C", [(3, 50, 512), (1, 1, 7), (2, 1000, 130), (6, 32768, 512), (32, 1024, 512), (4, 200, 64),
                                   (2, 130, 260), (1, 4097, 8)])
def test_segmax_and_adjoints(B, P, C):
    """max over points == torch.max(dim=-2) bit for bit (values AND first-occurrence indices, ties included); scatter
    and gather are each other's adjoints; double backward through the pair works."""
    from shapegan_amd import ops
    torch.manual_seed(B * P + C)
    x = torch.randn(B, P, C)
    if P > 4:
        x[:, 3] = x[:, 1]                      # exact ties: the first occurrence must win
        x[0, 2, :] = x.max() + 1
    want_v, want_i = x.max(dim=-2)
    xs = x.cuda().requires_grad_(True)
    out, idx = ops.SegMax.apply(xs)
    assert torch.equal(out.cpu(), want_v)
    if P > 4:
        first = (x == want_v.unsqueeze(1)).float().argmax(dim=1)     # first index attaining the max
        assert torch.equal(idx.cpu().long(), first)
    w = torch.randn(B, C, device=DEV)
    (out * w).sum().backward()
    ref = torch.zeros(B, P, C)
    ref.scatter_(1, idx.cpu().long().unsqueeze(1), w.cpu().unsqueeze(1))
    assert torch.equal(xs.grad.cpu(), ref)
    # adjointness <scatter(dy), u> == <dy, gather(u)> and double backward
    u = torch.randn(B, P, C, device=DEV)
    dy = torch.randn(B, C, device=DEV, requires_grad=True)
    sc = ops.SegMaxScatter.apply(dy, idx, P)
    lhs = (sc * u).sum()
    rhs = (dy * ops.SegMaxGather.apply(u, idx)).sum()
    np.testing.assert_allclose(lhs.item(), rhs.item(), rtol=1e-5)
    (g,) = torch.autograd.grad(lhs, dy)
    assert torch.equal(g, ops.SegMaxGather.apply(u, idx))


@pytest.mark.parametrize("B,P,C", [(3, 50, 512), (6, 32768, 512), (4, 200, 64), (2, 1000, 130)])
def test_segmax_nan_and_inf_follow_torch_max(B, P, C):
    """torch.max semantics on special values, in the one-pass kernel and in the chunked two-stage form alike: a NaN wins and
    the FIRST NaN's index is reported; -inf columns report index 0; +inf ties report the first."""
    from shapegan_amd import ops
    torch.manual_seed(B + P + C)
    x = torch.randn(B, P, C)
    x[:, :, 1] = float("-inf")
    x[0, P // 2, 2] = float("inf")
    x[0, P - 1, 2] = float("inf")
    x[B - 1, P - 1, 3] = float("nan")                      # last point of the last chunk
    x[0, min(P - 1, 70), 5] = float("nan")
    x[0, min(P - 1, 7), 5] = float("nan")                  # an earlier NaN in the same column: this one is reported
    want_v, want_i = x.max(dim=-2)
    out, idx = ops.SegMax.apply(x.to(DEV))
    out, idx = out.cpu(), idx.cpu().long()
    assert torch.equal(torch.isnan(out), torch.isnan(want_v))
    assert torch.equal(out[~torch.isnan(out)], want_v[~torch.isnan(want_v)])
    assert idx[0, 5] == min(P - 1, 7) and idx[B - 1, 3] == P - 1
    assert int(idx[0, 1]) == 0 and int(idx[0, 2]) == P // 2
    finite = torch.isfinite(want_v)
    first = (x == want_v.unsqueeze(1)).float().argmax(dim=1)
    assert torch.equal(idx[finite], first[finite])


def test_colsum_tall():
    from shapegan_amd import ops
    torch.manual_seed(5)
    x = torch.randn(5, 3000, 70, device=DEV)
    got = ops.colsum_tall_raw(x, 5, 3000 * 70, 3000, 70, 70)
    np.testing.assert_allclose(got.cpu().numpy(), x.double().sum(1).cpu().numpy(), rtol=1e-5, atol=1e-4)
    for batch, rows, cols, ld in ((3, 4099, 512, 516), (1, 7, 256, 256), (2, 131072, 64, 64)):      # the b128 form
        y = torch.randn(batch, rows, ld, device=DEV)
        got = ops.colsum_tall_raw(y, batch, rows * ld, rows, cols, ld)
        np.testing.assert_allclose(got.cpu().numpy(), y[:, :, :cols].double().sum(1).cpu().numpy(), rtol=1e-5, atol=3e-3)
    g = torch.randn(100000, 256, device=DEV, requires_grad=True)
    s = ops.ColSum.apply(g)
    np.testing.assert_allclose(s.detach().cpu().numpy(), g.detach().double().sum(0).cpu().numpy(), rtol=1e-5, atol=2e-3)
    s.sum().backward()
    assert torch.equal(g.grad, torch.ones_like(g))


@pytest.mark.parametrize("M,N,K,lda", [(256, 256, 20000, 20000), (256, 256, 4099, 4100), (70, 130, 1000, 1003), (256, 256, 31, 40),
                                       (32, 256, 262144, 262144), (256, 256, 33, 33)])
def test_gemm_nt_bigk(M, N, K, lda):
    """C = A B^T for K-contiguous operands (SDFNet weight gradients): ragged M / N / K, K not a multiple of the stage or of
    4, leading dimensions with padding, row-strided output."""
    from shapegan_amd import ops
    torch.manual_seed(M + N + K)
    a = torch.randn(M, lda, device=DEV)
    b = torch.randn(N, lda, device=DEV)
    out = torch.full((M, N + 5), 7.0, device=DEV)
    ops.gemm_nt_raw(a, b, out, M, N, K, lda, lda, N + 5)
    want = a[:, :K].double() @ b[:, :K].double().t()
    err = float((out[:, :N].double() - want).abs().max()) / float(want.abs().max())
    assert err < 2e-5, err
    assert float((out[:, N:] - 7.0).abs().max()) == 0.0      # nothing written beyond the N columns
