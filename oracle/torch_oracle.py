"""TEST INFRASTRUCTURE — CPU oracle for the shapegan hot path (never imported by shapegan_amd/).

A functional restatement, on torch CPU fp32 ops, of the reference's five hot-path modules and of the step bodies of
its five in-scope training scripts.  The reference's arithmetic lives in PyTorch itself (un-vendored, unpinned; the
code style implies torch ~1.3, this container has torch 2.10) — it has no native code and no tests, so the oracle is
"the same torch.nn.functional calls the reference's nn.Modules make", driven by a state_dict, and it is pinned
against the real reference classes imported from /root/reference (tests/test_oracle_pins.py) and against the golden
fixtures generated from them (tests/golden/, oracle/make_golden.py).

Every function cites the reference lines it follows.  Parameters are plain dicts name -> tensor using the
reference's state_dict keys.
"""
import copy

import torch
import torch.nn.functional as F

LATENT_CODE_SIZE = 128  # model/__init__.py:10


class kink_control(object):
    """Context manager around oracle forwards that deals with LeakyReLU / ReLU kinks.  At a kink the derivative is
    discontinuous: two correct fp32 implementations whose pre-activations differ in the last bits take different branches,
    and their gradients then differ by a finite amount (one flipped element of a conv layer moves a whole [Cin x taps] block
    of the weight gradient below it and, through the input gradient, everything further down).

    record_below=t : `fragile` collects (call index, flat element indices) of the pre-activations with |z| < t * mean|z| of
                     their layer — the elements whose branch fp32 rounding can decide either way.
    flips={call: indices} : those elements take the OTHER branch (value and derivative), i.e. the oracle evaluates the
                     network for the sign pattern an implementation with slightly different rounding would see."""

    def __init__(self, record_below=None, flips=None):
        self.record_below, self.flips = record_below, flips or {}
        self.fragile = []

    def __enter__(self):
        self._saved = (F.leaky_relu, F.relu)
        self._call = 0

        def watch(fn, leaky):
            def wrapped(z, *a, **k):
                i = self._call
                self._call += 1
                if self.record_below is not None:
                    with torch.no_grad():
                        near = (z.abs() < self.record_below * z.abs().mean()).reshape(-1).nonzero().flatten()
                    if near.numel():
                        self.fragile.append((i, near))
                y = fn(z, *a, **k)
                if i in self.flips:
                    slope = (a[0] if a else k.get("negative_slope", 0.01)) if leaky else 0.0
                    other = torch.where(z > 0, z * slope, z)
                    m = torch.zeros(z.numel(), dtype=z.dtype)
                    m[self.flips[i]] = 1
                    y = y + (other - y) * m.reshape(z.shape)
                return y
            return wrapped
        F.leaky_relu, F.relu = watch(F.leaky_relu, True), watch(F.relu, False)
        return self

    def __exit__(self, *exc):
        F.leaky_relu, F.relu = self._saved
        return False


def clone_state(sd, requires_grad=True, device="cpu"):
    """Deep copy of a state_dict onto `device`; float tensors become leaves that require grad (buffers do not)."""
    out = {}
    for k, v in sd.items():
        t = v.detach().clone().to(device)
        is_buffer = k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")
        if requires_grad and t.is_floating_point() and not is_buffer:
            t.requires_grad_(True)
        out[k] = t
    return out


def params_of(P):
    return [v for k, v in P.items() if v.requires_grad]


def _bn(P, prefix, x, training):
    """nn.BatchNorm{1,3}d forward incl. running-stat update (momentum 0.1, eps 1e-5)."""
    if training:
        P[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, P[prefix + ".running_mean"], P[prefix + ".running_var"], P[prefix + ".weight"],
                        P[prefix + ".bias"], training, 0.1, 1e-5)


# ---- model/gan.py ---------------------------------------------------------------------------------------------
def generator_forward(P, z, training=True):
    """Generator.forward, model/gan.py:8-29."""
    x = z.reshape((-1, LATENT_CODE_SIZE, 1, 1, 1))
    x = F.conv_transpose3d(x, P["layers.0.weight"], P["layers.0.bias"], stride=1)            # :9
    x = F.leaky_relu(_bn(P, "layers.1", x, training), 0.2)                                   # :10-11
    x = F.conv_transpose3d(x, P["layers.3.weight"], P["layers.3.bias"], stride=2, padding=1)  # :13
    x = F.leaky_relu(_bn(P, "layers.4", x, training), 0.2)
    x = F.conv_transpose3d(x, P["layers.6.weight"], P["layers.6.bias"], stride=2, padding=1)  # :17
    x = F.leaky_relu(_bn(P, "layers.7", x, training), 0.2)
    x = F.conv_transpose3d(x, P["layers.9.weight"], P["layers.9.bias"], stride=2, padding=1)  # :21
    return torch.tanh(x)                                                                     # :22


def discriminator_forward(P, x, use_sigmoid=True):
    """Discriminator.forward, model/gan.py:48-65."""
    if len(x.shape) < 5:
        x = x.unsqueeze(dim=1)
    x = F.leaky_relu(F.conv3d(x, P["layers.0.weight"], P["layers.0.bias"], stride=2, padding=1), 0.2)
    x = F.leaky_relu(F.conv3d(x, P["layers.2.weight"], P["layers.2.bias"], stride=2, padding=1), 0.2)
    x = F.leaky_relu(F.conv3d(x, P["layers.4.weight"], P["layers.4.bias"], stride=2, padding=1), 0.2)
    x = F.conv3d(x, P["layers.6.weight"], P["layers.6.bias"], stride=1)
    if use_sigmoid:
        x = torch.sigmoid(x)
    return x.squeeze()


def clip_weights(P, value):
    """Discriminator.clip_weights, model/gan.py:67-69."""
    with torch.no_grad():
        for p in params_of(P):
            p.clamp_(-value, value)


# ---- model/autoencoder.py -------------------------------------------------------------------------------------
def autoencoder_encode(P, x, training, variational, eps=None):
    """Autoencoder.encode, model/autoencoder.py:67-89.  `eps` replaces the N(0,1) draw at :79."""
    x = x.reshape((-1, 1, 32, 32, 32))
    for conv, bn in (("encoder.0", "encoder.1"), ("encoder.3", "encoder.4"), ("encoder.6", "encoder.7")):
        x = F.conv3d(x, P[conv + ".weight"], P[conv + ".bias"], stride=2, padding=1)
        x = F.leaky_relu(_bn(P, bn, x, training), 0.2)
    x = F.conv3d(x, P["encoder.9.weight"], P["encoder.9.bias"], stride=1)                     # :28
    x = F.leaky_relu(_bn(P, "encoder.10", x, training), 0.2)
    x = x.reshape(x.shape[0], -1)
    x = F.linear(x, P["encoder.13.weight"], P["encoder.13.bias"])                             # :34
    if not variational:
        return x
    x = F.leaky_relu(_bn(P, "encoder.vae-bn", x, training), 0.2)                              # :38-39
    mean = F.linear(x, P["encode_mean.weight"], P["encode_mean.bias"]).squeeze()
    log_variance = F.linear(x, P["encode_log_variance.weight"], P["encode_log_variance.bias"]).squeeze()
    if training:
        z = mean + torch.exp(log_variance * 0.5) * eps
    else:
        z = mean
    return z, mean, log_variance


def autoencoder_decode(P, z, training):
    """Autoencoder.decode, model/autoencoder.py:91-95."""
    if len(z.shape) == 1:
        z = z.unsqueeze(dim=0)
    x = F.linear(z, P["decoder.0.weight"], P["decoder.0.bias"])
    x = F.leaky_relu(_bn(P, "decoder.1", x, training), 0.2)
    x = x.reshape(-1, 2 * LATENT_CODE_SIZE, 1, 1, 1)
    x = F.conv_transpose3d(x, P["decoder.4.weight"], P["decoder.4.bias"], stride=1)           # :51
    x = F.leaky_relu(_bn(P, "decoder.5", x, training), 0.2)
    x = F.conv_transpose3d(x, P["decoder.7.weight"], P["decoder.7.bias"], stride=2, padding=1)
    x = F.leaky_relu(_bn(P, "decoder.8", x, training), 0.2)
    x = F.conv_transpose3d(x, P["decoder.10.weight"], P["decoder.10.bias"], stride=2, padding=1)
    x = F.leaky_relu(_bn(P, "decoder.11", x, training), 0.2)
    x = F.conv_transpose3d(x, P["decoder.13.weight"], P["decoder.13.bias"], stride=2, padding=1)
    return x.squeeze()


def autoencoder_forward(P, x, training=True, variational=False, eps=None):
    """Autoencoder.forward, model/autoencoder.py:97-104."""
    if not variational:
        return autoencoder_decode(P, autoencoder_encode(P, x, training, False), training)
    z, mean, log_variance = autoencoder_encode(P, x, training, True, eps)
    return autoencoder_decode(P, z, training), mean, log_variance


# ---- model/progressive_gan.py ---------------------------------------------------------------------------------
RESOLUTIONS = [8, 16, 32, 64]
FEATURE_COUNTS = [128, 64, 32, 1]


def from_sdf(x, iteration):
    """from_SDF, model/progressive_gan.py:9-16 (zero-channel padding)."""
    r, c = RESOLUTIONS[iteration], FEATURE_COUNTS[iteration]
    x = x.reshape((-1, 1, r, r, r))
    return torch.cat((x, torch.zeros((x.shape[0], c - 1, r, r, r), device=x.device, dtype=x.dtype)), dim=1)


def progressive_forward(P, x, iteration, fade_in_progress=1.0):
    """progressive_gan.Discriminator.forward, model/progressive_gan.py:44-57."""
    def stage(i, t):
        k = "optional_layers.%d.0" % i
        return F.leaky_relu(F.conv3d(t, P[k + ".weight"], P[k + ".bias"], stride=2, padding=1), 0.2)

    x_in = x
    x = stage(iteration, from_sdf(x, iteration))
    if fade_in_progress < 1.0 and iteration > 0:
        x2 = from_sdf(x_in[:, ::2, ::2, ::2], iteration - 1)
        x = fade_in_progress * x + (1.0 - fade_in_progress) * x2
    for i in range(iteration - 1, -1, -1):
        x = stage(i, x)
    x = x.reshape(-1, 64 * 256)
    x = F.leaky_relu(F.linear(x, P["head.1.weight"], P["head.1.bias"]), 0.2)
    return F.linear(x, P["head.3.weight"], P["head.3.bias"]).squeeze()


# ---- model/sdf_net.py -----------------------------------------------------------------------------------------
def sdfnet_forward(P, points, latent_codes):
    """SDFNet.forward, model/sdf_net.py:56-61."""
    inp = torch.cat((points, latent_codes), dim=1)
    x = inp
    for i in (0, 2, 4, 6):
        x = F.relu(F.linear(x, P["layers1.%d.weight" % i], P["layers1.%d.bias" % i]))
    x = torch.cat((x, inp), dim=1)
    for i in (0, 2, 4):
        x = F.relu(F.linear(x, P["layers2.%d.weight" % i], P["layers2.%d.bias" % i]))
    x = torch.tanh(F.linear(x, P["layers2.6.weight"], P["layers2.6.bias"]))
    return x.squeeze()


def sdfnet_min_preactivation(P, points, latent_codes):
    """min over the 7 hidden layers and 256 units of |pre-activation| per point.  A point whose smallest
    |pre-activation| is within fp32 rounding of 0 sits on a ReLU kink: two correct implementations may take different
    sides, so parity tests give such points zero upstream gradient (they are reported, not hidden)."""
    with torch.no_grad():
        inp = torch.cat((points, latent_codes), dim=1)
        x = inp
        m = torch.full((points.shape[0],), float("inf"), dtype=points.dtype)
        for i in (0, 2, 4, 6):
            z = F.linear(x, P["layers1.%d.weight" % i], P["layers1.%d.bias" % i])
            m = torch.minimum(m, z.abs().min(dim=1).values)
            x = F.relu(z)
        x = torch.cat((x, inp), dim=1)
        for i in (0, 2, 4):
            z = F.linear(x, P["layers2.%d.weight" % i], P["layers2.%d.bias" % i])
            m = torch.minimum(m, z.abs().min(dim=1).values)
            x = F.relu(z)
    return m


def tile_latents(z, points_per_shape):
    """sample_latent_codes tiling, train_hybrid_wgan.py:69 / train_hybrid_progressive_gan.py:92: row s*pps+q = z[s]."""
    return z.repeat((1, 1, points_per_shape)).reshape(-1, z.shape[1])


# ---- step bodies ----------------------------------------------------------------------------------------------
class WGANOracle(object):
    """train_wgan.py:37-46,60-84."""

    def __init__(self, g_state, c_state, lr=0.00005, clip=0.01):
        self.G, self.C = clone_state(g_state), clone_state(c_state)
        self.g_opt = torch.optim.RMSprop(params_of(self.G), lr=lr)
        self.c_opt = torch.optim.RMSprop(params_of(self.C), lr=lr)
        self.clip = clip

    def critic_step(self, real, z):
        self.g_opt.zero_grad()
        self.c_opt.zero_grad()
        fake = generator_forward(self.G, z, True).detach()
        out_fake = discriminator_forward(self.C, fake, False)
        out_real = discriminator_forward(self.C, real, False)
        loss = torch.mean(out_fake) - torch.mean(out_real)
        loss.backward()
        self.c_opt.step()
        clip_weights(self.C, self.clip)
        return loss.detach(), out_fake.detach(), out_real.detach()

    def generator_step(self, z):
        self.g_opt.zero_grad()
        self.c_opt.zero_grad()
        fake = generator_forward(self.G, z, True)
        out = discriminator_forward(self.C, fake, False)
        loss = -torch.mean(out)
        loss.backward()
        self.g_opt.step()
        return loss.detach(), out.detach()

    def step(self, reals, zs_critic, z_gen):
        last = None
        for i, (real, z) in enumerate(zip(reals, zs_critic)):
            last = self.critic_step(real, z)
            if i == 0:
                self.generator_step(z_gen)
        return last


def reconstruction_loss(output, target):
    """get_reconstruction_loss, train_autoencoder.py:57-62 (in-place x32 where target < 0)."""
    difference = output - target
    wrong_signs = target < 0
    difference[wrong_signs] *= 32
    return torch.mean(torch.abs(difference))


def kld_loss(mean, log_variance):
    """train_autoencoder.py:54-55."""
    return -0.5 * torch.sum(1 + log_variance - mean.pow(2) - log_variance.exp()) / mean.nelement()


class AutoencoderOracle(object):
    """train_autoencoder.py:35,98-117."""

    def __init__(self, state, variational=False, lr=0.00005):
        self.P = clone_state(state)
        self.variational = variational
        self.opt = torch.optim.Adam(params_of(self.P), lr=lr)

    def step(self, batch, eps=None):
        self.opt.zero_grad()
        if self.variational:
            output, mean, log_variance = autoencoder_forward(self.P, batch, True, True, eps)
            kld = kld_loss(mean, log_variance)
        else:
            output = autoencoder_forward(self.P, batch, True, False)
            kld = 0
        rec = reconstruction_loss(output, batch)
        loss = rec + kld
        loss.backward()
        self.opt.step()
        return rec.detach(), output.detach()


class SDFAutoDecoderOracle(object):
    """train_sdf_autodecoder.py:20-45,77-91 (with integer floor division at :78)."""

    def __init__(self, net_state, latent_codes, points, sdf, pointcloud_size=200000, lr=1e-5, sigma=0.01, cutoff=0.1):
        self.P = clone_state(net_state)
        self.latent_codes = latent_codes.detach().clone().requires_grad_(True)
        self.points, self.sdf = points, sdf.clamp(-cutoff, cutoff)
        self.pointcloud_size, self.sigma = pointcloud_size, sigma
        self.net_opt = torch.optim.Adam(params_of(self.P), lr=lr)
        self.lat_opt = torch.optim.Adam([self.latent_codes], lr=lr)

    def step(self, indices):
        model_indices = indices // self.pointcloud_size
        batch_latent = self.latent_codes[model_indices, :]
        batch_points = self.points[indices, :]
        batch_sdf = self.sdf[indices]
        self.net_opt.zero_grad()
        if self.latent_codes.grad is not None:
            self.latent_codes.grad.data.zero_()
        output = sdfnet_forward(self.P, batch_points, batch_latent)
        loss = torch.mean(torch.abs(output - batch_sdf)) + self.sigma * torch.mean(torch.pow(batch_latent, 2))
        loss.backward()
        self.net_opt.step()
        self.lat_opt.step()
        return loss.detach()


class HybridWGANOracle(object):
    """train_hybrid_wgan.py:23-26,53-56,67-72,83-115."""

    def __init__(self, g_state, c_state, grid_points, resolution=32, lr=0.00001, clip=0.01):
        self.G, self.C = clone_state(g_state), clone_state(c_state)
        self.res, self.grid, self.clip = resolution, grid_points, clip
        self.g_opt = torch.optim.Adam(params_of(self.G), lr=lr)
        self.c_opt = torch.optim.RMSprop(params_of(self.C), lr=lr)

    def generate(self, z):
        pts = self.grid.repeat((z.shape[0], 1))
        out = sdfnet_forward(self.G, pts, tile_latents(z, self.res ** 3))
        return out.reshape(-1, self.res, self.res, self.res)

    def critic_step(self, real, z):
        self.c_opt.zero_grad()
        fake = self.generate(z)
        out_fake = discriminator_forward(self.C, fake, False)
        out_real = discriminator_forward(self.C, real, False)
        loss = torch.mean(out_fake) - torch.mean(out_real)
        loss.backward()
        self.c_opt.step()
        clip_weights(self.C, self.clip)
        return loss.detach(), out_fake.detach(), out_real.detach()

    def generator_step(self, z):
        self.g_opt.zero_grad()
        self.c_opt.zero_grad()
        fake = self.generate(z)
        out = discriminator_forward(self.C, fake, False)
        loss = torch.mean(-out)
        loss.backward()
        self.g_opt.step()
        return loss.detach(), out.detach()


class HybridProgressiveGANOracle(object):
    """train_hybrid_progressive_gan.py:36-38,81-82,90-111,134-166."""

    def __init__(self, g_state, d_state, grid_points, iteration, fade_in_progress=1.0, lr=0.0001, gp_weight=10.0):
        self.G, self.D = clone_state(g_state), clone_state(d_state)
        self.it, self.fade = iteration, fade_in_progress
        self.res, self.grid, self.gp_weight = RESOLUTIONS[iteration], grid_points, gp_weight
        self.g_opt = torch.optim.RMSprop(params_of(self.G), lr=lr)
        self.d_opt = torch.optim.RMSprop(params_of(self.D), lr=lr)

    def disc(self, x):
        return progressive_forward(self.D, x, self.it, self.fade)

    def generate(self, z):
        pts = self.grid.repeat((z.shape[0], 1))
        out = sdfnet_forward(self.G, pts, tile_latents(z, self.res ** 3))
        return out.reshape(-1, self.res, self.res, self.res)

    def gradient_penalty(self, real, fake, alpha):
        alpha = alpha.expand(real.shape)
        interpolated = alpha * real + ((1 - alpha) * fake)
        interpolated.requires_grad = True
        out = self.disc(interpolated)
        gradients = torch.autograd.grad(outputs=out, inputs=interpolated, grad_outputs=torch.ones_like(out),
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
        return ((gradients.norm(2, dim=(1, 2, 3)) - 1) ** 2).mean() * self.gp_weight

    def generator_step(self, z):
        self.g_opt.zero_grad()
        fake = self.generate(z)
        loss = -self.disc(fake).mean()
        loss.backward()
        self.g_opt.step()
        return loss.detach()

    def discriminator_step(self, real, z, alpha):
        self.d_opt.zero_grad()
        fake = self.generate(z)
        out_fake = self.disc(fake)
        out_real = self.disc(real)
        gp = self.gradient_penalty(real.detach(), fake.detach(), alpha)
        loss = out_fake.mean() - out_real.mean() + gp
        loss.backward()
        self.d_opt.step()
        return loss.detach(), gp.detach()


class ClassicGANOracle(object):
    """train_gan.py:28-31,57-88: Adam 1e-3 generator, Adam 1e-5 discriminator with sigmoid + BCE;
    per batch: one generator update, one discriminator update on fakes, one on reals."""

    def __init__(self, g_state, d_state, g_lr=0.001, d_lr=0.00001):
        self.G, self.D = clone_state(g_state), clone_state(d_state)
        self.g_opt = torch.optim.Adam(params_of(self.G), lr=g_lr)
        self.d_opt = torch.optim.Adam(params_of(self.D), lr=d_lr)

    def generate(self, z):
        return generator_forward(self.G, z, True)

    def generator_step(self, z):
        self.g_opt.zero_grad()
        out = discriminator_forward(self.D, self.generate(z), True)
        loss = -torch.mean(torch.log(out))
        loss.backward()
        self.g_opt.step()
        return loss.detach()

    def discriminator_fake_step(self, z):
        self.d_opt.zero_grad()
        out = discriminator_forward(self.D, self.generate(z).detach(), True)
        loss = F.binary_cross_entropy(out, torch.zeros_like(out))
        loss.backward()
        self.d_opt.step()
        return loss.detach(), out.detach()

    def discriminator_real_step(self, real):
        self.d_opt.zero_grad()
        out = discriminator_forward(self.D, real, True)
        loss = F.binary_cross_entropy(out, torch.ones_like(out))
        loss.backward()
        self.d_opt.step()
        return loss.detach(), out.detach()


class HybridGANOracle(ClassicGANOracle):
    """train_hybrid_gan.py:43-46,64-67,77-125: the same cadence with an SDFNet generator sampled on the voxel grid.
    (The reference keeps the generator graph in the fake-discriminator update and discards its gradients.)"""

    def __init__(self, g_state, d_state, grid_points, resolution=32, g_lr=0.001, d_lr=0.00001):
        ClassicGANOracle.__init__(self, g_state, d_state, g_lr, d_lr)
        self.res, self.grid = resolution, grid_points

    def generate(self, z):
        pts = self.grid.repeat((z.shape[0], 1))
        out = sdfnet_forward(self.G, pts, tile_latents(z, self.res ** 3))
        return out.reshape(-1, self.res, self.res, self.res)


# ---- PointNet-discriminator GAN family (model/point_sdf_net.py) -----------------------------------------------------
def pointnet_forward(P, pos, dist):
    """PointNet.forward, point_sdf_net.py:33-47 (dense path, batch=None)."""
    dist = dist.unsqueeze(-1) if dist.size(-1) != 1 else dist
    x = torch.cat([pos, dist], dim=-1)
    for i in (0, 2, 4, 6):
        x = F.linear(x, P["nn1.%d.weight" % i], P["nn1.%d.bias" % i])
        if i < 6:
            x = F.relu(x)
    x = x.max(dim=-2)[0]
    for i in (0, 2, 4):
        x = F.linear(x, P["nn2.%d.weight" % i], P["nn2.%d.bias" % i])
        if i < 4:
            x = F.relu(x)
    return x


def sdf_generator_forward(P, pos, z, num_layers=8):
    """SDFGenerator.forward, point_sdf_net.py:83-119 (norm=True, dropout 0)."""
    x = pos
    for i in range(num_layers):
        if i == num_layers // 2:
            x = torch.cat([x, pos], dim=-1)
        x = F.linear(x, P["lins.%d.weight" % i], P["lins.%d.bias" % i])
        if i == 0:
            x = F.linear(z, P["z_lin1.weight"], P["z_lin1.bias"]).unsqueeze(1) + x
        if i == num_layers // 2:
            x = F.linear(z, P["z_lin2.weight"], P["z_lin2.bias"]).unsqueeze(1) + x
        if i < num_layers - 1:
            w = P["norms.%d.weight" % i]
            x = F.relu(F.layer_norm(x, (w.shape[0],), w, P["norms.%d.bias" % i], 1e-5))
    return x


class PointGANOracle(object):
    """train_point_gan.py:15-26,52-83."""

    def __init__(self, g_state, d_state, lr=0.0001, gp_weight=10.0):
        self.G, self.D = clone_state(g_state), clone_state(d_state)
        self.g_opt = torch.optim.RMSprop(params_of(self.G), lr=lr)
        self.d_opt = torch.optim.RMSprop(params_of(self.D), lr=lr)
        self.gp_weight = gp_weight

    def gradient_penalty(self, pos, dist, fake, alpha):
        interpolated = alpha * dist + (1 - alpha) * fake
        interpolated.requires_grad_(True)
        out = pointnet_forward(self.D, pos, interpolated)
        grad = torch.autograd.grad(out, interpolated, grad_outputs=torch.ones_like(out), create_graph=True,
                                   retain_graph=True, only_inputs=True)[0]
        grad_norm = grad.view(grad.size(0), -1).norm(dim=-1, p=2)
        return self.gp_weight * ((grad_norm - 1).pow(2).mean())

    def critic_step(self, uniform, z, alpha):
        pos, dist = uniform[..., :3], uniform[..., 3:]
        self.d_opt.zero_grad()
        fake = sdf_generator_forward(self.G, pos, z)
        out_real = pointnet_forward(self.D, pos, dist)
        out_fake = pointnet_forward(self.D, pos, fake)
        d_loss = out_fake.mean() - out_real.mean()
        gp = self.gradient_penalty(pos, dist, fake, alpha)
        (d_loss + gp).backward()
        self.d_opt.step()
        return d_loss.detach(), gp.detach()

    def generator_step(self, uniform, z):
        pos = uniform[..., :3]
        self.g_opt.zero_grad()
        fake = sdf_generator_forward(self.G, pos, z)
        loss = -pointnet_forward(self.D, pos, fake).mean()
        loss.backward()
        self.g_opt.step()
        return loss.detach()


def snapshot(P):
    return copy.deepcopy({k: v.detach().clone() for k, v in P.items()})
