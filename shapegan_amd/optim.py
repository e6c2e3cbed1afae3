"""Fused flat-buffer optimizers (K11): torch.optim.RMSprop / torch.optim.Adam semantics with torch defaults, one
HIP launch per step over a flat fp32 parameter buffer (instead of ~5 ATen launches per tensor), WGAN weight
clipping folded into the RMSprop step, and the same flat gradient buffer serving as the RCCL all-reduce bucket.

Reference call sites: optim.RMSprop(lr) in train_wgan.py:45-46, train_hybrid_progressive_gan.py:81-82,
train_hybrid_wgan.py:56; optim.Adam(lr) in train_autoencoder.py:35, train_sdf_autodecoder.py:44-45,
train_hybrid_wgan.py:53; critic.clip_weights(0.01) in train_wgan.py:71 / train_hybrid_wgan.py:94.

The constructor re-points every parameter's `.data` (and `.grad`) at slices of two flat buffers; Parameter objects,
state_dict keys and shapes are unchanged.  Parameters whose `.grad` is None at step time are skipped exactly like
torch.optim does (unused progressive-GAN stages).
"""
import collections

import torch

from . import lib as L
from .lib import check, ptr, stream


class _Flat(object):
    def __init__(self, params):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("optimizer got an empty parameter list")
        dev = self.params[0].device
        sizes = [p.numel() for p in self.params]
        # 4-float alignment of every slice keeps float4 access legal for any consumer
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 3) // 4 * 4
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=dev)
        # The gradient buffer is preceded by a small header, one float per parameter ("this rank has a gradient for it"): a
        # data-parallel exchange of the head slice carries it along for free, so the ranks can tell whether they all skip the
        # same parameters (shapegan_amd.parallel.GradBucket).  Kernels and the optimizer only ever see `grad`.
        self.header_len = (len(self.params) + 3) // 4 * 4
        self.grad_store = torch.zeros(self.header_len + off, dtype=torch.float32, device=dev)
        self.header, self.grad = self.grad_store[:self.header_len], self.grad_store[self.header_len:]
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            self.flat[o:o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + n].view(p.shape)
            if p.grad is not None:
                self.grad[o:o + n].copy_(p.grad.reshape(-1))
            p.grad = self.grad[o:o + n].view(p.shape)
        self.slots = [L.register_grad_slot(p, self.grad, o) for p, o in zip(self.params, self.offsets)]
        self.range = L.register_param_range(self.flat.data_ptr(), 4 * self.total)   # this buffer's own parameter epoch
        L.bump_param_epoch()

    def grad_view_ok(self, i):
        p = self.params[i]
        return p.grad is not None and p.grad.data_ptr() == self.grad.data_ptr() + 4 * self.offsets[i] \
            and p.grad.is_contiguous()

    def coherent(self):
        return all(self.grad_view_ok(i) for i in range(len(self.params)))

    def adopt_grads(self, indices=None, attach_missing=False):
        """Copies the live .grad of every parameter (or of those in `indices`) into its flat slice and re-attaches the view, so
        that the flat buffer is what the exchange and the update both see.  A parameter WITHOUT a gradient contributes zeros
        to the wire but keeps p.grad None, so step() skips it exactly as torch.optim does (Adam's moments of an unused
        progressive-GAN stage must not keep moving it); attach_missing=True attaches the zero slice instead (callers that
        must update the whole buffer in one fixed launch)."""
        for i in (range(len(self.params)) if indices is None else indices):
            p, o = self.params[i], self.offsets[i]
            if self.grad_view_ok(i):
                continue
            n = p.numel()
            if p.grad is None:
                self.grad[o:o + n].zero_()
                if not attach_missing:
                    continue
            else:
                self.grad[o:o + n].copy_(p.grad.reshape(-1))
            p.grad = self.grad[o:o + n].view(p.shape)

    def check_storage(self):
        """module.to()/.cuda()/.float() after the optimizer was built re-allocates p.data: the flat buffer would be
        updated while the module computes with the orphaned copy.  Fail instead."""
        base = self.flat.data_ptr()
        for p, o in zip(self.params, self.offsets):
            if p.data_ptr() != base + 4 * o:
                raise RuntimeError("optimizer: a parameter no longer lives in the flat buffer (module.to()/.float() "
                                   "after constructing the optimizer?); rebuild the optimizer")

    def zero_grad(self):
        """Gradients -> None (torch's set_to_none semantics) and every slice of the flat buffer open for a direct write:
        the next `lib.backward(loss)`'s weight-gradient kernels store into the slices and autograd adopts those views as p.grad
        (lib.grad_destination), so neither a memset of the buffer nor a `p.grad += g` pass per parameter is launched.  A
        parameter that receives no gradient keeps p.grad None and is skipped by step(), as in torch.optim.

        Aliasing (differs from torch): p.grad of a parameter owned by this optimizer IS its slice of the flat buffer.  A
        reference to p.grad kept across zero_grad() is overwritten by the next backward (torch would leave the old tensor
        intact); clone it if it must survive.  Direct writes happen only inside `shapegan_amd.lib.backward(loss)` (what the
        trainers call): gradients RETURNED by torch.autograd.grad(loss, params), and those of a plain `loss.backward()`, are
        ordinary tensors (never slices), as in torch; step() / the exchange copy them in on demand."""
        for p, slot in zip(self.params, self.slots):
            p.grad = None
            slot.written = False

    def __del__(self):
        try:
            L.unregister_grad_slots(self.slots)
            L.unregister_param_range(self.range)
        except Exception:       # interpreter shutdown
            pass


class _Base(object):
    def __init__(self, params):
        self.f = _Flat(params)
        self.grad_scale = 1.0  # set to 1/world_size by shapegan_amd.parallel for data-parallel averaging
        self.param_groups = [{"params": self.f.params}]

    def zero_grad(self, set_to_none=False):
        self.f.zero_grad()

    @property
    def flat_grad(self):
        return self.f.grad

    def _segments(self, keys=None):
        """[(offset, length, grad_ptr[, keep-alive])] of what to update: the whole buffer when every grad is the flat view;
        otherwise maximal runs of consecutive parameters whose gradient is their flat slice (one launch per run; the alignment
        gaps between slices hold zeros), one segment per parameter whose gradient lives elsewhere, nothing for a parameter
        without a gradient (torch.optim skips those).  `keys[i]` (optional) must also be equal within a run."""
        f = self.f
        if f.coherent() and (keys is None or len(set(keys)) <= 1):
            return [(0, f.total, f.grad.data_ptr())]
        segs, run = [], None
        for i, (p, o) in enumerate(zip(f.params, f.offsets)):
            if p.grad is None:
                run = None
                continue
            if f.grad_view_ok(i):
                key = None if keys is None else keys[i]
                if run is not None and run[2] == key:
                    run[1] = o + p.numel()
                else:
                    run = [o, o + p.numel(), key]
                    segs.append(run)
                continue
            run = None
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            if g.dtype != torch.float32:
                g = g.float()
            segs.append((o, p.numel(), g.data_ptr(), g))
        base = f.grad.data_ptr()
        return [(s[0], s[1] - s[0], base + 4 * s[0]) if isinstance(s, list) else s for s in segs]


class RMSprop(_Base):
    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, clip=0.0):
        super().__init__(params)
        self.lr, self.alpha, self.eps, self.clip = lr, alpha, eps, clip
        self.square_avg = torch.zeros_like(self.f.flat)

    def step(self):
        lib = L.load()
        self.f.check_storage()
        base_p, base_s = self.f.flat.data_ptr(), self.square_avg.data_ptr()
        for seg in self._segments():
            o, n, g = seg[0], seg[1], seg[2]
            L.note_device(self.f.flat)
            check(lib.sg_rmsprop_step(base_p + 4 * o, g, base_s + 4 * o, n, self.lr, self.alpha, self.eps,
                                      self.grad_scale, self.clip, stream()), "rmsprop_step")
        L.bump_param_epoch(self.f.range)


class Adam(_Base):
    """torch.optim.Adam defaults.  capturable=True keeps the step counter on the device (sg_adam_step_dev) so that the
    whole update can sit inside a captured graph; it needs every parameter's gradient in the flat buffer on every step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        super().__init__(params)
        self.lr, self.betas, self.eps = lr, betas, eps
        self.capturable = capturable
        if capturable:
            self.step_dev = torch.zeros(1, dtype=torch.int64, device=self.f.flat.device)
            self.corr_dev = torch.zeros(544, dtype=torch.float32, device=self.f.flat.device)   # SG_ADAM_DEV_WORDS: corrections [2] + arrival tickets
        self.exp_avg = torch.zeros_like(self.f.flat)
        self.exp_avg_sq = torch.zeros_like(self.f.flat)
        # torch keeps one step counter per parameter; a parameter that never had a grad never advances
        self.steps = [0] * len(self.f.params)
        # optional int32 device word: while it is non-zero, step() leaves parameters, moments and the device step counter untouched
        # (ops.batch_index_guard — the auto-decoder's out-of-range batch index); the host counters are taken back by `unstep`
        self.guard = None
        self._advanced = collections.deque(maxlen=4096)     # per step(): the parameters whose host counter it advanced

    def unstep(self, count=1):
        """Takes back the host-side step counters of the last `count` step() calls — for a caller that learned afterwards that the
        guard word was set while those steps' kernels ran (they did nothing).  The capturable variant has nothing to take back: its
        counter is on the device and was not advanced."""
        for _ in range(min(count, len(self._advanced))):
            for i in self._advanced.pop():
                self.steps[i] -= 1

    def _prepare_capturable(self):
        f = self.f
        if not f.coherent():
            if any(p.grad is None for p in f.params):
                raise RuntimeError("Adam(capturable=True) updates the whole flat buffer in one fixed launch: every "
                                   "parameter needs a gradient on every step (torch.optim would skip the missing ones)")
            f.adopt_grads()      # gradients that arrived as ordinary tensors: copied into their slices (capturable too)

    def step(self):
        lib = L.load()
        f = self.f
        f.check_storage()
        base_p, base_m, base_v = f.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr()
        uniform = f.coherent() and len(set(self.steps)) == 1
        guard = self.guard.data_ptr() if self.guard is not None else None
        if self.capturable:
            self._prepare_capturable()
            L.note_device(f.flat)
            check(lib.sg_adam_step_dev_guarded(base_p, f.grad.data_ptr(), base_m, base_v, f.total, self.lr, self.betas[0],
                                               self.betas[1], self.eps, self.step_dev.data_ptr(), self.corr_dev.data_ptr(),
                                               self.grad_scale, guard, stream()), "adam_step_dev")
        elif uniform:
            self.steps = [s + 1 for s in self.steps]
            self._advanced.append(range(len(self.steps)))
            L.note_device(f.flat)
            check(lib.sg_adam_step_guarded(base_p, f.grad.data_ptr(), base_m, base_v, f.total, self.lr, self.betas[0],
                                           self.betas[1], self.eps, self.steps[0], self.grad_scale, guard, stream()), "adam_step")
        else:
            # per-parameter step counters, as torch: a parameter without a gradient neither moves nor ages
            self._advanced.append([i for i, p in enumerate(f.params) if p.grad is not None])
            for i in self._advanced[-1]:
                self.steps[i] += 1
            first = {o: i for i, o in enumerate(f.offsets)}
            for seg in self._segments(keys=self.steps):
                o, n, g = seg[0], seg[1], seg[2]
                L.note_device(f.flat)
                check(lib.sg_adam_step_guarded(base_p + 4 * o, g, base_m + 4 * o, base_v + 4 * o, n, self.lr, self.betas[0],
                                               self.betas[1], self.eps, self.steps[first[o]], self.grad_scale, guard, stream()),
                      "adam_step")
        L.bump_param_epoch(f.range)


def step_together(optimizers):
    """`for o in optimizers: o.step()` — as ONE launch when they are two to four capturable Adam optimizers on one device that share
    their guard word (sg_adam_step_dev_multi: the network's and the latent table's optimizer of train_sdf_autodecoder.py:44-45,
    90-91 inside a captured step), one after the other otherwise."""
    import ctypes
    opts = list(optimizers)
    same = (2 <= len(opts) <= 4 and all(isinstance(o, Adam) and o.capturable for o in opts)
            and len({o.f.flat.device for o in opts}) == 1
            and len({None if o.guard is None else o.guard.data_ptr() for o in opts}) == 1)
    if not same:
        for o in opts:
            o.step()
        return
    lib = L.load()
    for o in opts:
        o.f.check_storage()
        o._prepare_capturable()
    n = len(opts)
    ptrs = lambda vals: (ctypes.c_void_p * n)(*vals)
    floats = lambda vals: (ctypes.c_float * n)(*vals)
    L.note_device(opts[0].f.flat)
    guard = opts[0].guard.data_ptr() if opts[0].guard is not None else None
    check(lib.sg_adam_step_dev_multi(n, ptrs([o.f.flat.data_ptr() for o in opts]), ptrs([o.f.grad.data_ptr() for o in opts]),
                                     ptrs([o.exp_avg.data_ptr() for o in opts]), ptrs([o.exp_avg_sq.data_ptr() for o in opts]),
                                     (ctypes.c_long * n)(*[o.f.total for o in opts]), floats([o.lr for o in opts]),
                                     floats([o.betas[0] for o in opts]), floats([o.betas[1] for o in opts]),
                                     floats([o.eps for o in opts]), ptrs([o.step_dev.data_ptr() for o in opts]),
                                     ptrs([o.corr_dev.data_ptr() for o in opts]), floats([o.grad_scale for o in opts]), guard,
                                     stream()), "adam_step_dev_multi")
    for o in opts:
        L.bump_param_epoch(o.f.range)
