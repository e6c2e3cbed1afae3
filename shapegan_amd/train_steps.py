"""The inner-loop bodies of the five in-scope training scripts, restated as step methods on the native modules.

Python owns the loops (north star); every heavy op inside a step is a HIP kernel behind shapegan_amd.ops /
shapegan_amd.optim.  Each method cites the reference lines it restates; random draws (latents, GP alpha) are
arguments so that parity tests can inject the reference's values.  Data-parallel runs insert exactly one flat
gradient all-reduce per optimizer step (shapegan_amd.parallel.GradBucket); single-process runs skip it.
"""
import collections
import contextlib

import torch

from . import lib, ops, optim
from .parallel import GradBucket, allreduce_tensor_, world_size


@contextlib.contextmanager
def frozen(module):
    """Skip weight-gradient kernels for a network whose gradients the step discards anyway."""
    flags = [p.requires_grad for p in module.parameters()]
    for p in module.parameters():
        p.requires_grad_(False)
    try:
        yield
    finally:
        for p, f in zip(module.parameters(), flags):
            p.requires_grad_(f)


class _Graphed(object):
    """`fn(*tensors)` as ONE captured graph launch (single process): the first `warm` calls run eagerly (lazy initialisations,
    workspaces), the next is captured — into static copies of its tensor arguments — and every later call copies its arguments in
    and replays.  Everything `fn` does must be on the device (no .item(), no host-side step counters); returned tensors are
    overwritten by the next call.  Weight images are rebuilt inside the graph (ops._capturing), so replays see the parameters
    other graphs or eager steps have written in between."""

    def __init__(self, fn, warm=2):
        self.fn, self.warm, self.calls = fn, warm, 0
        self.graph, self.static, self.out = None, None, None

    def __call__(self, *tensors):
        self.calls += 1
        if self.calls <= self.warm:
            return self.fn(*tensors)
        if self.graph is None or any(s.shape != t.shape for s, t in zip(self.static, tensors)):
            self.static = [t.clone() for t in tensors]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.out = self.fn(*self.static)
            self.graph = graph
        else:
            for s, t in zip(self.static, tensors):
                s.copy_(t)
        self.graph.replay()
        lib.bump_param_epoch()      # the captured optimizer kernels rewrote parameters through raw pointers
        return self.out


class WGANTrainer(object):
    """train_wgan.py: RMSprop(lr 5e-5) for both nets, n_critic 5, weight clipping 0.01, batch 64."""

    def __init__(self, generator, critic, lr=0.00005, clip=0.01):
        self.generator, self.critic = generator, critic
        critic.use_sigmoid = False                                   # train_wgan.py:31
        self.g_opt = optim.RMSprop(generator.parameters(), lr=lr)    # :45
        self.c_opt = optim.RMSprop(critic.parameters(), lr=lr, clip=clip)  # :46 + clip_weights :71 fused
        self.g_bucket, self.c_bucket = GradBucket(self.g_opt), GradBucket(self.c_opt)
        self._batches = None      # the unit's critic batches [updates, n_fake + n_real, 1, R, R, R], kept between units

    def _unit_batches(self, updates, n_fake, n_real, res, device):
        shape = (updates, n_fake + n_real, 1, res, res, res)
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:          # "cuda" names the current device: compare like with like
            device = torch.device("cuda", torch.cuda.current_device())
        if self._batches is None or tuple(self._batches.shape) != shape or self._batches.device != device:
            self._batches = torch.empty(shape, dtype=torch.float32, device=device)
        return self._batches

    def real_slots(self, n_real, n_fake=None, resolution=32, updates=5, device=None):
        """Where `step` wants the unit's real batches: the real halves of its critic batches, `updates` tensors [n_real, 1, R, R, R].
        critic(fake) and critic(real) are one pass over a concatenated batch here (see `critic_step`), which costs one device copy
        of every real batch that the reference does not have (its loader's batch is read where `.to(device)` put it,
        train_wgan.py:56-66).  An input pipeline that delivers its batches INTO these slots (`slot.copy_(host_batch,
        non_blocking=True)`) and passes the slots as `reals` gets that copy back: `critic_step` notices the batch is in place."""
        n_fake = n_real if n_fake is None else n_fake
        device = device if device is not None else next(self.critic.parameters()).device
        batches = self._unit_batches(updates, n_fake, n_real, resolution, device)
        return [batches[g, n_fake:] for g in range(updates)]

    def critic_step(self, real, z, both=None, fake_ready=False):
        """train_wgan.py:60-71.  `z` [B,128] replaces generator.generate's CPU draw.  `both` (optional): the [n_fake + n_real, 1,
        R, R, R] batch to use; fake_ready: its fake half already holds generator(z) (see `step`)."""
        self.c_opt.zero_grad()
        # The critic has no batch statistics, so critic(fake) and critic(real) (train_wgan.py:64-66) are one pass over
        # the concatenated batch: same outputs and gradients, half the launches, one weight-gradient reduction.  The generator
        # writes its samples straight into the fake half of that batch.
        n_fake, n_real = z.shape[0], real.shape[0]
        res = real.shape[-1]
        with torch.no_grad():                      # == generate(...).detach(); BN running stats still update
            if both is None:
                both = torch.empty((n_fake + n_real, 1, res, res, res), dtype=torch.float32, device=real.device)
            if not fake_ready:
                fake = self.generator(z, out=both[:n_fake])
                if fake.data_ptr() != both.data_ptr():
                    both[:n_fake].copy_(fake)
            if not (real.data_ptr() == both[n_fake:].data_ptr() and real.is_contiguous() and real.numel() == both[n_fake:].numel()):
                both[n_fake:].copy_(real.reshape(n_real, 1, res, res, res))     # (else: delivered in place, `real_slots`)
        out = self.critic(both)
        out_fake, out_real = out[:n_fake], out[n_fake:]
        loss = ops.mean_difference(out, n_fake)        # mean(out_fake) - mean(out_real), one launch
        self.c_bucket.arm()
        lib.backward(loss)
        self.c_bucket.finish()
        self.c_opt.step()
        return loss.detach(), out_fake.detach(), out_real.detach()

    def generator_step(self, z):
        """train_wgan.py:74-84 (always BATCH_SIZE fresh latents).  The reference also accumulates critic weight
        gradients here and throws them away at the next critic.zero_grad(); they are not computed."""
        self.g_opt.zero_grad()
        fake = self.generator(z)
        with frozen(self.critic):
            out = self.critic(fake)
        loss = ops.neg_mean(out)
        self.g_bucket.arm()
        lib.backward(loss)
        self.g_bucket.finish()
        self.g_opt.step()
        return loss.detach(), out.detach()

    def step(self, reals, zs_critic, z_gen):
        """One 5+1 unit: batch_index % 5 == 0 trains the critic and the generator, the next four batches the critic.

        The generator changes once per unit (right after the first critic update), so the samples of the remaining critic updates
        depend on nothing those updates compute: they are generated in ONE pass behind the generator update
        (Generator.forward_groups): every transposed convolution runs once on 4 x B latents, every BatchNorm3d normalises each of the
        four batches with its own statistics and updates its running buffers batch after batch — the same samples and buffers as
        four separate evaluations (tests/test_gpu_modules.py), a third of the launches and the convolutions at a batch size they are
        more efficient at."""
        same = len({(tuple(r.shape), tuple(z.shape)) for r, z in zip(reals[1:], zs_critic[1:])}) == 1
        if len(reals) < 3 or not same or not self.generator.training:
            last = None
            for i, (real, z) in enumerate(zip(reals, zs_critic)):
                last = self.critic_step(real, z)
                if i == 0:
                    self.generator_step(z_gen)
            return last
        same = same and reals[0].shape == reals[1].shape and zs_critic[0].shape == zs_critic[1].shape
        n_fake, n_real, res = zs_critic[1].shape[0], reals[1].shape[0], reals[1].shape[-1]
        batches = self._unit_batches(len(reals), n_fake, n_real, res, reals[1].device) if same else None
        self.critic_step(reals[0], zs_critic[0], batches[0] if same else None)
        self.generator_step(z_gen)
        if not same:
            batches = self._unit_batches(len(reals), n_fake, n_real, res, reals[1].device)
        with torch.no_grad():
            self.generator.forward_groups(list(zs_critic[1:]), [batches[g, :n_fake] for g in range(1, len(reals))])
        last = None
        for g, (real, z) in enumerate(zip(reals[1:], zs_critic[1:]), start=1):
            last = self.critic_step(real, z, batches[g], fake_ready=True)
        return last


def reconstruction_loss(output, target):
    """train_autoencoder.py:57-62: mean |d|, d = output - target, d *= 32 where target < 0."""
    return ops.weighted_l1(output, target.reshape(output.shape), 32.0)     # one streaming pass (sg_loss_weighted_l1)


def kld_loss(mean, log_variance):
    """train_autoencoder.py:54-55."""
    return ops.kld(mean, log_variance)


def voxel_difference(output, target):
    """train_autoencoder.py:50-52: fraction of sign mismatches (integer count, bit-exact)."""
    target = target.reshape(output.shape) if target.numel() == output.numel() else target.expand_as(output).contiguous()
    return ops.count_sign_mismatch(output, target).item() / output.numel()


class AutoencoderTrainer(object):
    """train_autoencoder.py: Adam(lr 5e-5); classic (AE) or variational."""

    def __init__(self, autoencoder, lr=0.00005, capturable=False):
        self.autoencoder = autoencoder
        # capturable: Adam keeps its step counter on the device, so step_graphed() can replay a captured step
        self.opt = optim.Adam(autoencoder.parameters(), lr=lr, capturable=capturable)       # :35
        self.bucket = GradBucket(self.opt)
        self.capturable = capturable
        self._graph, self._graph_in, self._graph_out, self._graph_calls = None, None, None, 0

    def step_graphed(self, batch):
        """The classic-autoencoder step as ONE captured graph launch (single process).  At the script's small batches the step
        is bound by the host: ~180 kernel launches of a few microseconds each behind ~26 us of Python / ctypes / autograd per
        launch (batch 4: 4.8 ms per step with the GPU busy for 0.9 ms).  Everything the step does is on the device — losses,
        BatchNorm running statistics and their counters, Adam with a device-side step count — so the third call captures it and
        every later call copies the batch into the static input and replays.  Returned tensors are overwritten by the next
        call.  (The variational branch draws eps on the host, model/autoencoder.py:74-82, and is not capturable.)"""
        if not self.capturable or world_size() > 1 or self.autoencoder.is_variational:
            raise RuntimeError("step_graphed needs AutoencoderTrainer(capturable=True), one process and the classic autoencoder")
        self._graph_calls += 1
        if self._graph_calls <= 2:
            return self.step(batch)
        if self._graph is None or self._graph_in.shape != batch.shape:
            self._graph_in = batch.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._graph_out = self.step(self._graph_in)
            self._graph = graph
        else:
            self._graph_in.copy_(batch)
        self._graph.replay()
        lib.bump_param_epoch()      # the captured optimizer kernels rewrote the parameters through raw pointers
        return self._graph_out

    def step(self, batch):
        """train_autoencoder.py:98-117."""
        ae = self.autoencoder
        self.opt.zero_grad()
        ae.train()
        if ae.is_variational:
            output, mean, log_variance = ae(batch)
            kld = kld_loss(mean, log_variance)
        else:
            output = ae(batch)
            kld = 0
        rec = reconstruction_loss(output, batch)
        loss = rec + kld
        self.bucket.arm()
        lib.backward(loss)
        self.bucket.finish()
        self.opt.step()
        return rec.detach(), output.detach()


class SDFAutoDecoderTrainer(object):
    """train_sdf_autodecoder.py: DeepSDF auto-decoder, Adam(lr 1e-5) for the net and for the latent table."""

    def __init__(self, sdf_net, latent_codes, points, sdf, pointcloud_size=200000, lr=1e-5, sigma=0.01, cutoff=0.1,
                 capturable=False):
        self.net, self.latent_codes = sdf_net, latent_codes
        self.points = points
        self.sdf = sdf.clamp(-cutoff, cutoff)                         # :27
        self.pointcloud_size, self.sigma = pointcloud_size, sigma
        latent_codes.requires_grad = True                             # :42
        # capturable: Adam keeps its step counter on the device, so step_graphed() can replay a captured step
        self.net_opt = optim.Adam(sdf_net.parameters(), lr=lr, capturable=capturable)     # :44
        self.lat_opt = optim.Adam([latent_codes], lr=lr, capturable=capturable)           # :45
        self.net_bucket, self.lat_bucket = GradBucket(self.net_opt), GradBucket(self.lat_opt)
        self.capturable = capturable
        self._graph, self._graph_idx, self._graph_loss, self._graph_calls = None, None, None, 0
        self._sorted_calls = 0
        # An out-of-range batch index is an IndexError BEFORE any update in the reference (train_sdf_autodecoder.py:79).  Here it is
        # noticed without a host synchronisation, one step late (`_poll_indices`) — so both optimizers are guarded by the device
        # word the sort kernel sets in the same stream: the bad batch's update is a no-op on parameters, moments and step counters,
        # and the error leaves the state of the step before it (also inside a replayed graph).
        # The pair of words is THIS trainer's (ops.BadIndexWords): another trainer on the same device neither skips nor clears it.
        self._words = ops.BadIndexWords(latent_codes.device)
        self.net_opt.guard = self.lat_opt.guard = self._words.guard
        self._updates = collections.deque(maxlen=4096)     # the sort call number behind every update issued so far

    def _poll_indices(self, synchronise=False):
        try:
            self._words.raise_if_bad(synchronise_first=synchronise, synchronise_before_raise=True)
        except IndexError as e:
            # the host may be several steps ahead of the device: every update issued behind the first bad sort was a no-op on the
            # device (the guard word is sticky until the host clears it) — take the host-side step counters of exactly those back
            skipped = sum(1 for s in self._updates if s >= getattr(e, "sort_sequence", 0) > 0)
            self.net_opt.unstep(skipped)
            self.lat_opt.unstep(skipped)
            self._updates.clear()
            raise

    def step_graphed(self, indices):
        """`step` as ONE captured graph launch (single process only): the shape-sorted flow from 8192 points on, the gathered
        flow below.  The reference's 20 000-point batch is launch-bound (27 kernels of a few microseconds on 313 tiles); the
        first two calls run eagerly (lazy initialisations, workspaces), the third is captured and every call from then on is a
        replay.  Nothing in the step needs the host: the batch-index check is a word in pinned host memory that the sort kernel
        sets and this method reads before each replay (an out-of-range index raises at the NEXT call, one step late).  The
        returned loss tensor is overwritten by the next call."""
        if not self.capturable or world_size() > 1:
            raise RuntimeError("step_graphed needs SDFAutoDecoderTrainer(capturable=True) in a single process")
        self._graph_calls += 1
        if self._graph_calls <= 2:
            return self.step(indices)
        self._poll_indices()           # the pinned host word an earlier replay's sort kernel may have set
        if self._graph is None or self._graph_idx.shape != indices.shape:
            self._graph_idx = indices.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._graph_loss = self.step(self._graph_idx)
            self._graph = graph
        else:
            self._graph_idx.copy_(indices)
        self._graph.replay()
        # the captured Adam kernels rewrite the parameters through raw pointers: neither tensor._version nor the
        # epoch moved, so derived weight images (ops._PackCache) would otherwise survive the replay
        lib.bump_param_epoch()
        return self._graph_loss

    def step(self, indices):
        """train_sdf_autodecoder.py:77-91 (with the integer floor division `:78` intends).

        The batch is re-ordered by shape (a permutation of the batch changes neither the loss nor any gradient, only
        fp32 summation order): the fused kernel then reads each point's latent contribution as a per-shape bias row
        and the latent-table gradient is assembled from per-shape sums, so `latent_codes[model_indices]` ([N,L]) and
        its scatter-add backward never exist.  The regulariser mean(z_batch^2) is evaluated through shape counts:
        sum_s count_s |z_s|^2 / (N L)."""
        # tiny batches: the per-shape bookkeeping costs more than it saves (measured: sorted 0.72 ms, gathered 0.80 ms at the
        # reference's 20 000-point batch of 64 shapes)
        if indices.numel() < 8192 or self.latent_codes.shape[0] > ops.sdf_batch_sort_max_shapes():
            return self.step_gathered(indices)
        return self.step_sorted(indices)

    def step_sorted(self, indices):
        """The shape-sorted data flow described in `step`: the batch is grouped by one native counting sort
        (ops.sdf_batch_sort: keys, gathers of points / sdf, run bounds and counts; no host round trip)."""
        shapes = self.latent_codes.shape[0]
        if indices.numel() >= ops._OVERLAP_MIN_POINTS:
            # GPU-bound steps: the weight pack + latent fold (they depend on the parameters only) on a side stream next to the sort
            with ops._SideStream(self.latent_codes.device) as side:
                with side.run():
                    self.net.prepare_latents(self.latent_codes)
                batch_points, batch_sdf, model_indices, seg_off, counts = ops.sdf_batch_sort(
                    indices, self.pointcloud_size, shapes, self.points, self.sdf, words=self._words)
        else:
            batch_points, batch_sdf, model_indices, seg_off, counts = ops.sdf_batch_sort(
                indices, self.pointcloud_size, shapes, self.points, self.sdf, words=self._words)
        self._sorted_calls += 1
        if self._sorted_calls == 1:
            self._poll_indices(synchronise=True)    # first call: synchronous (a systematically wrong index source fails at once)
        else:
            self._poll_indices()                    # every later call: no host sync; an out-of-range index raises one step late,
                                                    # behind a guarded (skipped) update
        self.net_opt.zero_grad()
        self.lat_opt.zero_grad()
        n, width = indices.shape[0], self.latent_codes.shape[1]
        # the regulariser's gradient w.r.t. the latent table, 2 sigma count_s z_s / (n L), is added by the backward of the latent
        # fold (one gradient contribution for the table, written where its flat slice lives; the loss sees a detached table)
        reg = (counts, float(2.0 / (n * width / self.sigma))) if self.sigma != 0 else None
        output = self.net.forward_segments(batch_points, self.latent_codes, model_indices, seg_off, latent_reg=reg)
        if self.sigma != 0:
            # data term + sigma * mean(z_batch^2) through shape counts in one op; sigma rides in the denominator (sigma 0: the
            # term is 0, as in the reference's formula)
            loss = ops.deepsdf_loss(output, batch_sdf, self.latent_codes.detach(), counts, n * width / self.sigma)
        else:
            loss = ops.weighted_l1(output, batch_sdf)
        lib.backward(loss)
        self.net_bucket.allreduce()
        self.lat_bucket.allreduce()
        optim.step_together((self.net_opt, self.lat_opt))      # one launch when both are capturable (the graphed step)
        self._updates.append(self._words.seq)
        return loss.detach()

    def step_gathered(self, indices):
        """The same step through the reference's data flow (materialised latent_codes[model_indices], per-point
        latent kernel mode) — kept for A/B and parity."""
        # a sorted step's bad batch may still be pending on the device: this update would then be a guarded no-op too — it is polled
        # for and counted like the sorted ones (mixed batch sizes around the 8192-point threshold)
        self._poll_indices()
        model_indices = torch.div(indices, self.pointcloud_size, rounding_mode='floor')
        self.net_opt.zero_grad()
        self.lat_opt.zero_grad()
        batch_latent = ops.gather_rows(self.latent_codes, model_indices)
        batch_points = ops.gather_rows(self.points, indices)
        batch_sdf = ops.gather_rows(self.sdf.unsqueeze(1), indices).squeeze(1)
        output = self.net(batch_points, batch_latent)
        if self.sigma != 0:
            loss = ops.deepsdf_loss(output, batch_sdf, batch_latent, None, batch_latent.numel() / self.sigma)
        else:
            loss = ops.weighted_l1(output, batch_sdf)
        lib.backward(loss)
        self.net_bucket.allreduce()
        self.lat_bucket.allreduce()
        optim.step_together((self.net_opt, self.lat_opt))      # one launch when both are capturable (the graphed step)
        self._updates.append(self._words.seq)
        return loss.detach()


class HybridWGANTrainer(object):
    """train_hybrid_wgan.py: SDFNet generator sampled on a fixed 32^3 grid (Adam 1e-5) + gan.Discriminator critic
    (RMSprop 1e-5, clip 0.01), batch 8, n_critic 5."""

    def __init__(self, generator, critic, grid_points, resolution=32, lr=0.00001, clip=0.01):
        self.generator, self.critic = generator, critic
        critic.use_sigmoid = False                                    # :40
        self.res = resolution
        self.grid = grid_points                                       # [R^3, 3], get_voxel_coordinates(R)  (:72)
        self.g_opt = optim.Adam(generator.parameters(), lr=lr)        # :53
        self.c_opt = optim.RMSprop(critic.parameters(), lr=lr, clip=clip)  # :56 + :94
        self.g_bucket, self.c_bucket = GradBucket(self.g_opt), GradBucket(self.c_opt)
        self._tiled = {}

    def _points(self, count):
        if count not in self._tiled:
            self._tiled[count] = self.grid.repeat((count, 1))
        return self._tiled[count]

    def generate(self, z):
        """generator(grid_points, tiled latents).reshape(-1,R,R,R) (:84-86) without tiling the latents."""
        r3 = self.res ** 3
        sdf = self.generator.forward_shapes(self._points(z.shape[0]), z, r3)
        return sdf.reshape(-1, self.res, self.res, self.res)

    def critic_step(self, real, z):
        """train_hybrid_wgan.py:83-94: the generator graph is kept in the reference (its grads are discarded by the
        next generator_optimizer.zero_grad()); they are not computed here."""
        self.c_opt.zero_grad()
        with torch.no_grad():
            fake = self.generate(z)
        # one critic pass over the concatenated fake+real batch (train_hybrid_wgan.py:87-89), as in WGANTrainer
        n_fake = fake.shape[0]
        out = self.critic(torch.cat([fake, real.reshape((-1,) + tuple(fake.shape[1:]))]))
        out_fake, out_real = out[:n_fake], out[n_fake:]
        loss = ops.mean_difference(out, n_fake)        # mean(out_fake) - mean(out_real), one launch
        self.c_bucket.arm()
        lib.backward(loss)
        self.c_bucket.finish()
        self.c_opt.step()
        return loss.detach(), out_fake.detach(), out_real.detach()

    def generator_step(self, z):
        """train_hybrid_wgan.py:97-115."""
        self.g_opt.zero_grad()
        fake = self.generate(z)
        with frozen(self.critic):
            out = self.critic(fake)
        loss = ops.neg_mean(out)
        self.g_bucket.arm()
        lib.backward(loss)
        self.g_bucket.finish()
        self.g_opt.step()
        return loss.detach(), out.detach()


class HybridProgressiveGANTrainer(object):
    """train_hybrid_progressive_gan.py: SDFNet generator + progressive discriminator, WGAN-GP (lambda 10),
    RMSprop(lr 1e-4) for both, batch 16; the DataParallel wrap (:62-68) becomes process-per-GPU + GradBucket."""

    def __init__(self, generator, discriminator, grid_points, resolution, lr=0.0001, gp_weight=10.0):
        self.generator, self.discriminator = generator, discriminator
        self.res, self.gp_weight = resolution, gp_weight
        self.grid = grid_points
        self.g_opt = optim.RMSprop(generator.parameters(), lr=lr)           # :81
        self.d_opt = optim.RMSprop(discriminator.parameters(), lr=lr)       # :82
        self.g_bucket, self.d_bucket = GradBucket(self.g_opt), GradBucket(self.d_opt)
        self._tiled = {}

    def _points(self, count):
        if count not in self._tiled:
            self._tiled[count] = self.grid.repeat((count, 1))
        return self._tiled[count]

    def generate(self, z):
        r3 = self.res ** 3
        sdf = self.generator.forward_shapes(self._points(z.shape[0]), z, r3)
        return sdf.reshape(-1, self.res, self.res, self.res)

    def gradient_penalty(self, real, fake, alpha):
        """train_hybrid_progressive_gan.py:102-111; `alpha` [B,1,1,1] replaces the on-device torch.rand."""
        interpolated = ops.lerp_rows(real, fake, alpha)
        interpolated.requires_grad = True
        out = self.discriminator(interpolated)
        gradients = torch.autograd.grad(outputs=out, inputs=interpolated, grad_outputs=torch.ones_like(out),
                                        create_graph=True, retain_graph=True, only_inputs=True)[0]
        return ops.gradient_penalty(gradients, self.gp_weight)

    def generator_step(self, z):
        """:135-149."""
        self.g_opt.zero_grad()
        fake = self.generate(z)
        with frozen(self.discriminator):
            out = self.discriminator(fake)
        loss = ops.neg_mean(out)
        self.g_bucket.arm()
        lib.backward(loss)
        self.g_bucket.finish()
        self.g_opt.step()
        return loss.detach()

    def discriminator_step(self, real, z, alpha):
        """:153-166.  The reference back-propagates into the generator here too and discards the result."""
        self.d_opt.zero_grad()
        with torch.no_grad():
            fake = self.generate(z)
        # the progressive discriminator has no batch statistics, so D(fake) and D(real) (:146-150) are one pass over the
        # concatenated batch: same outputs and gradients, half the launches, the batch-32 kernels instead of the batch-16 ones
        out = self.discriminator(torch.cat([fake, real.reshape(fake.shape)]))
        gp = self.gradient_penalty(real.detach(), fake.detach(), alpha)
        loss = ops.mean_difference(out, fake.shape[0]) + gp          # mean(out_fake) - mean(out_real) + penalty
        self.d_bucket.arm()
        lib.backward(loss)
        self.d_bucket.finish()
        self.d_opt.step()
        return loss.detach(), gp.detach()


class ClassicGANTrainer(object):
    """train_gan.py: gan.Generator (Adam 1e-3) against gan.Discriminator with its sigmoid
    (Adam 1e-5, binary cross-entropy), batch 64; per batch one generator update (:57-67), one discriminator update on
    fakes (:75-80) and one on reals (:82-86).  BCE and -mean(log) are native single-launch ops on the [B] score vector."""

    def __init__(self, generator, discriminator, g_lr=0.001, d_lr=0.00001):
        self.generator, self.discriminator = generator, discriminator
        discriminator.use_sigmoid = True
        self.g_opt = optim.Adam(generator.parameters(), lr=g_lr)          # :28
        self.d_opt = optim.Adam(discriminator.parameters(), lr=d_lr)      # :31
        self.g_bucket, self.d_bucket = GradBucket(self.g_opt), GradBucket(self.d_opt)

    def generate(self, z):
        return self.generator(z)

    def generator_step(self, z):
        """:57-67 (the discriminator gradients the reference accumulates here are zeroed at :75; not computed)."""
        self.g_opt.zero_grad()
        fake = self.generate(z)
        with frozen(self.discriminator):
            out = self.discriminator(fake)
        loss = ops.neg_mean_log(out)                 # -torch.mean(torch.log(out)), train_gan.py:65
        self.g_bucket.arm()
        lib.backward(loss)
        self.g_bucket.finish()
        self.g_opt.step()
        return loss.detach()

    def _discriminator_update(self, sample, target_value):
        self.d_opt.zero_grad()
        out = self.discriminator(sample)
        loss = ops.bce_const(out, target_value)      # binary_cross_entropy against a constant target, train_gan.py:78,84
        self.d_bucket.arm()
        lib.backward(loss)
        self.d_bucket.finish()
        self.d_opt.step()
        return loss.detach(), out.detach()

    def discriminator_fake_step(self, z):
        """:75-80."""
        with torch.no_grad():
            fake = self.generate(z)
        return self._discriminator_update(fake, 0.0)

    def discriminator_real_step(self, real):
        """:82-86."""
        return self._discriminator_update(real, 1.0)

    def step(self, real, z_gen, z_disc):
        self.generator_step(z_gen)
        fake = self.discriminator_fake_step(z_disc)
        return fake, self.discriminator_real_step(real)


class HybridGANTrainer(ClassicGANTrainer):
    """train_hybrid_gan.py: the same cadence with an SDFNet generator sampled on the 32^3 grid, batch 8 (:55).  The
    reference keeps the generator graph in the fake-discriminator update (:103-108) and discards its gradients; they
    are not computed here.  Per-shape latents replace sample_latent_codes' [B*R^3, L] tiling (:64-67)."""

    def __init__(self, generator, discriminator, grid_points, resolution=32, g_lr=0.001, d_lr=0.00001):
        ClassicGANTrainer.__init__(self, generator, discriminator, g_lr, d_lr)
        self.res, self.grid = resolution, grid_points
        self._tiled = {}

    def generate(self, z):
        count = z.shape[0]
        if count not in self._tiled:
            self._tiled[count] = self.grid.repeat((count, 1))
        sdf = self.generator.forward_shapes(self._tiled[count], z, self.res ** 3)
        return sdf.reshape(-1, self.res, self.res, self.res)


class PointGANTrainer(object):
    """train_point_gan.py: SDFGenerator (LayerNorm MLP, latent 128) against a PointNet critic on
    [B, P, 4] point clouds, WGAN-GP (lambda 10) on the DISTANCE channel only, RMSprop 1e-4 for both; the critic is
    updated on every batch, the generator on every 5th (:52-83)."""

    def __init__(self, generator, critic, lr=0.0001, gp_weight=10.0):
        self.generator, self.critic = generator, critic
        self.g_opt = optim.RMSprop(generator.parameters(), lr=lr)    # :25
        self.d_opt = optim.RMSprop(critic.parameters(), lr=lr)       # :26
        self.g_bucket, self.d_bucket = GradBucket(self.g_opt), GradBucket(self.d_opt)
        self.gp_weight = gp_weight
        self._critic_graph, self._generator_graph = _Graphed(self.critic_step), _Graphed(self.generator_step)

    def critic_step_graphed(self, uniform, z, alpha):
        """`critic_step` as one captured graph launch (single process): the update is ~110 launches of 5 - 60 us behind a dense pass
        of 3 ms — eagerly it is paced by the host on a slow one (3.9 - 5.5 ms measured over the round's boxes)."""
        if world_size() > 1:
            raise RuntimeError("critic_step_graphed: single process only")
        return self._critic_graph(uniform, z, alpha)

    def generator_step_graphed(self, uniform, z):
        if world_size() > 1:
            raise RuntimeError("generator_step_graphed: single process only")
        return self._generator_graph(uniform, z)

    def gradient_penalty(self, pos, dist, fake, alpha):
        """:61-70; `alpha` [B,1,1] replaces the on-device torch.rand."""
        interpolated = ops.lerp_rows(dist, fake, alpha)
        interpolated.requires_grad_(True)
        out = self.critic(pos, interpolated)
        grad = torch.autograd.grad(out, interpolated, grad_outputs=torch.ones_like(out), create_graph=True,
                                   retain_graph=True, only_inputs=True)[0]
        return ops.gradient_penalty(grad, self.gp_weight)

    def critic_step(self, uniform, z, alpha):
        """:52-74.  The reference keeps the generator graph here and discards its gradients (G_optimizer.zero_grad at
        :77 precedes the only G_optimizer.step); they are not computed.  Large clouds: the three critic evaluations (real,
        generated, interpolated) are ONE plain pass of the per-point network over 3 B clouds that finds the points holding the
        maxima, and one recorded pass over those 512 points per cloud (PointNet.forward_selected) — the outputs, the penalty's
        gradient with respect to the interpolated distances and its double backward are those of three separate calls."""
        pos, dist = uniform[..., :3], uniform[..., 3:]
        self.d_opt.zero_grad()
        with torch.no_grad():
            fake = self.generator(pos, z)
        if pos.dim() == 3 and pos.shape[-2] >= self.critic.SPARSE_MIN_POINTS:
            B = pos.shape[0]
            interpolated = ops.lerp_rows(dist, fake, alpha)
            interpolated.requires_grad_(True)
            x = torch.cat([torch.cat([pos, d], dim=-1) for d in (dist, fake, interpolated)], dim=0)      # [3B,P,4]
            xs = self.critic.gather_points(x, self.critic.selected_points(x.detach()))
            out = self.critic.forward_selected(xs)
            out_real, out_fake, out_i = out[:B], out[B:2 * B], out[2 * B:]
            d_loss = ops.mean(out_fake) - ops.mean(out_real)
            grad = torch.autograd.grad(out_i, interpolated, grad_outputs=torch.ones_like(out_i), create_graph=True,
                                       retain_graph=True, only_inputs=True)[0]
            gp = ops.gradient_penalty(grad, self.gp_weight)
        else:
            out_real = self.critic(pos, dist)
            out_fake = self.critic(pos, fake)
            d_loss = ops.mean(out_fake) - ops.mean(out_real)
            gp = self.gradient_penalty(pos, dist, fake, alpha)
        loss = d_loss + gp
        self.d_bucket.arm()
        lib.backward(loss)
        self.d_bucket.finish()
        self.d_opt.step()
        return d_loss.detach(), gp.detach()

    def generator_step(self, uniform, z):
        """:76-83.  For large clouds the update is evaluated on the points that matter: the critic sees the generated cloud
        through a max over its points, so only the (at most 512) points of a shape that hold a channel's maximum pass a gradient
        back — and both networks treat points independently.  One plain evaluation of generator and per-point critic over all
        points (nothing recorded) finds those points; the recorded evaluation, its backward and the parameter gradients then run
        on 512 points per shape instead of P.  Loss and gradients are those of the dense evaluation (the points left out
        contribute exact zeros)."""
        pos = uniform[..., :3]
        self.g_opt.zero_grad()
        if pos.shape[-2] >= self.critic.SPARSE_MIN_POINTS:
            with torch.no_grad():
                fake = self.generator(pos, z)
                idx = self.critic.selected_points(torch.cat([pos, fake], dim=-1))
            pos_s = self.critic.gather_points(pos, idx)                           # [B,512,3]
            fake_s = self.generator(pos_s, z)
            with frozen(self.critic):
                out = self.critic.forward_selected(torch.cat([pos_s, fake_s], dim=-1))
        else:
            fake = self.generator(pos, z)
            with frozen(self.critic):
                out = self.critic(pos, fake)
        loss = ops.neg_mean(out)
        self.g_bucket.arm()
        lib.backward(loss)
        self.g_bucket.finish()
        self.g_opt.step()
        return loss.detach()
