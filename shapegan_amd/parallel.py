"""Data parallelism for the hot path: one process per GPU, replicated parameters, batch axis sharded, ONE
all-reduce of the flat fp32 gradient buffer per network per optimizer step (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU for tests).

Replaces nn.DataParallel in train_hybrid_progressive_gan.py:62-68, which re-broadcasts 19.4 MB of parameters and
scatters up to 2.2 GB of inputs on every forward.  Here parameters stay resident (identical init by seed or load(),
identical updates afterwards), inputs are generated/loaded per rank, and the only exchange is the gradient sum; the
1/world factor is applied inside the optimizer kernel (grad_scale), not as a separate pass.
"""
import atexit
import json
import os
import sys
import threading
import time
import weakref

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialises torch.distributed from the torchrun environment (RANK/WORLD_SIZE/LOCAL_RANK/MASTER_*).
    Returns (rank, world, local_rank).  No-op single-process fallback when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # SG_DIST_BACKEND=gloo lets the N>1 code path be exercised on a single-GPU box (all ranks on cuda:0)
            backend = os.environ.get("SG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


class CommInitTimeout(RuntimeError):
    """ncclCommInitRank did not return: the helper thread is still inside RCCL holding the device, so nothing that follows on
    this device (in particular torch.distributed's own NCCL collectives) can be trusted not to hang.  Fatal — never a fallback."""


def _close_weak(ref):
    comm = ref()
    if comm is not None:
        comm.close()


def _rank_state(rank, world, device):
    """What this rank knows about itself when a communicator bring-up goes wrong (printed as one JSON object per rank)."""
    env = {k: v for k, v in os.environ.items()
           if k.startswith(("NCCL_", "RCCL_", "HSA_", "HIP_", "ROCR_", "MASTER_", "SG_")) or k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    state = {"rank": rank, "world": world, "device": device, "pid": os.getpid(), "env": env,
             "torch_distributed": dist.get_backend() if dist.is_initialized() else None}
    try:
        state["visible_devices"] = torch.cuda.device_count()
        state["device_name"] = torch.cuda.get_device_name(device)
        free, total = torch.cuda.mem_get_info(device)
        state["hbm_free_gb"], state["hbm_total_gb"] = round(free / 2 ** 30, 1), round(total / 2 ** 30, 1)
    except Exception as e:       # noqa: BLE001 — a diagnostic must not raise
        state["device_query_error"] = str(e)
    return state


class NativeComm(object):
    """The C-ABI gradient exchange (include/shapegan_hip.h: sg_allreduce_*, libshapegan_comm.so): an RCCL communicator with
    its own stream; `launch` enqueues an in-place all-reduce(sum) of a contiguous fp32 CUDA tensor ordered after the current
    stream's work, `wait` makes the current stream wait for everything launched — no torch.distributed on the hot path.
    torch.distributed is used once, as the out-of-band channel for the 128-byte RCCL unique id."""

    def __init__(self, rank=None, world=None, device=None, unique_id=None, init_timeout=None):
        import ctypes
        from . import lib as L
        self.L, self.lib = L, L.load_comm()
        rank = (dist.get_rank() if dist.is_initialized() else 0) if rank is None else rank
        world = world_size() if world is None else world
        device = torch.cuda.current_device() if device is None else device
        nbytes = self.lib.sg_allreduce_unique_id_bytes()
        if unique_id is None:
            buf = ctypes.create_string_buffer(nbytes)
            if rank == 0:
                L.check_comm(self.lib.sg_allreduce_unique_id(buf, nbytes), "unique_id")
            box = [buf.raw]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            unique_id = box[0]
        handle = ctypes.c_void_p()
        self.handle = None
        # ncclCommInitRank blocks until every rank has joined: run it on a helper thread so that a rank that never shows up
        # becomes an error here (after `init_timeout` seconds) instead of a hang
        result = {}

        def init():
            rc = self.lib.sg_allreduce_init(ctypes.byref(handle), rank, world, unique_id, len(unique_id), device)
            msg = self.lib.sg_comm_last_error() if rc != 0 else b""
            result["rc"], result["msg"] = rc, (msg or b"").decode()
        if init_timeout is None:
            init_timeout = float(os.environ.get("SG_COMM_INIT_TIMEOUT", "300"))
        th = threading.Thread(target=init, name="sg_allreduce_init", daemon=True)
        t0 = time.perf_counter()
        th.start()
        th.join(init_timeout)
        self.init_seconds = time.perf_counter() - t0
        if th.is_alive():
            # everything a post-mortem of a first multi-GPU bring-up needs, from THIS rank, before the job dies
            print("shapegan_amd.parallel: rank %d of %d STILL INSIDE ncclCommInitRank after %.0f s — %s"
                  % (rank, world, init_timeout, json.dumps(_rank_state(rank, world, device))), file=sys.stderr, flush=True)
            raise CommInitTimeout("shapegan_comm: ncclCommInitRank did not return within %.0f s (rank %d of %d)" % (init_timeout, rank, world))
        if os.environ.get("SG_COMM_VERBOSE", "1") != "0" and world > 1:
            print("shapegan_amd.parallel: rank %d of %d: ncclCommInitRank on device %d returned %d after %.2f s"
                  % (rank, world, device, result["rc"], self.init_seconds), file=sys.stderr, flush=True)
        if result["rc"] != 0:
            raise RuntimeError("shapegan_comm init failed (%d): %s" % (result["rc"], result["msg"]))
        self.handle, self.rank, self.world = handle, rank, world
        atexit.register(_close_weak, weakref.ref(self))      # a weak reference: the registry must not pin the communicator

    def info(self):
        """What the communicator reports about itself: {"ranks": ncclCommCount, "rank", "device", "rccl_version": "2.26.6"}."""
        import ctypes
        vals = [ctypes.c_int(0) for _ in range(4)]
        self.L.check_comm(self.lib.sg_allreduce_info(self.handle, *[ctypes.byref(v) for v in vals]), "info")
        code = vals[3].value
        return {"ranks": vals[0].value, "rank": vals[1].value, "device": vals[2].value,
                "rccl_version": "%d.%d.%d" % (code // 10000, code // 100 % 100, code % 100) if code >= 10000 else str(code)}

    def launch(self, t):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise RuntimeError("NativeComm.launch needs a contiguous fp32 CUDA tensor")
        self.L.check_comm(self.lib.sg_allreduce_launch(self.handle, t.data_ptr(), t.numel(),
                                                       torch.cuda.current_stream().cuda_stream), "launch")

    def wait(self):
        self.L.check_comm(self.lib.sg_allreduce_wait(self.handle, torch.cuda.current_stream().cuda_stream), "wait")

    def close(self):
        if self.handle is not None:
            handle, self.handle = self.handle, None
            self.lib.sg_allreduce_destroy(handle)


_native = None
_native_tried = False
TRANSPORT = {"name": "none", "reason": "single process"}     # what carries the gradient exchange (bench.py reports it)


def _all_ranks_agree(ok):
    """True iff `ok` holds on every rank (one tiny torch.distributed all-reduce; also the rendezvous after an init attempt)."""
    flag = torch.tensor([0.0 if ok else 1.0], device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.SUM)
    return float(flag.item()) == 0.0


def negotiate_native(make_comm, verify, agree=None, log=None):
    """Brings the C-ABI exchange up on every rank or on none.  `make_comm()` builds this rank's communicator (may raise / time
    out), `verify(comm)` checks one exchange against the reference transport, `agree(ok)` tells whether every rank succeeded.
    Any failure anywhere makes ALL ranks close what they built and returns (None, reason) — loudly (`log`), never a hang and
    never a mixed transport."""
    agree = agree or _all_ranks_agree
    log = log or (lambda m: print(m, file=sys.stderr, flush=True))
    comm, reason = None, ""
    try:
        comm = make_comm()
    except CommInitTimeout as e:
        # the helper thread is still blocked inside ncclCommInitRank with the device: agree() (a torch.distributed collective on
        # the same device) could hang behind it.  A rank that never joins is a broken job, not a reason to switch transports.
        log("shapegan_amd.parallel: FATAL — %s" % e)
        raise
    except Exception as e:       # noqa: BLE001 — any other failure means "fall back"
        reason = "init: %s" % e
    if not agree(comm is not None):
        if comm is not None:
            comm.close()
        reason = reason or "another rank could not initialise its communicator"
        log("shapegan_amd.parallel: C-ABI RCCL exchange unavailable (%s) — FALLING BACK to torch.distributed all-reduce" % reason)
        return None, reason
    ok = False
    try:
        ok = bool(verify(comm))
        if not ok:
            reason = "verification all-reduce disagreed with torch.distributed"
    except Exception as e:       # noqa: BLE001
        reason = "verify: %s" % e
    if not agree(ok):
        comm.close()
        reason = reason or "another rank failed the verification exchange"
        log("shapegan_amd.parallel: C-ABI RCCL exchange unavailable (%s) — FALLING BACK to torch.distributed all-reduce" % reason)
        return None, reason
    return comm, ""


def _verify_against_torch(comm):
    """One 4096-float exchange through the new communicator against torch.distributed's all-reduce of the same data."""
    gen = torch.Generator().manual_seed(1234 + comm.rank)
    x = torch.randn(4096, generator=gen).cuda()
    ref = x.clone()
    comm.launch(x)
    comm.wait()
    dist.all_reduce(ref, op=dist.ReduceOp.SUM)
    torch.cuda.synchronize()
    return bool(torch.allclose(x, ref, rtol=1e-5, atol=1e-5)) and comm.info()["ranks"] == comm.world


def native_comm():
    """The process-wide NativeComm, or None when torch.distributed carries the exchange.  With the "nccl" backend (= RCCL on
    ROCm), GPUs and more than one rank the C-ABI exchange is the DEFAULT; SG_NATIVE_ALLREDUCE=0 is the escape hatch.  It is
    negotiated once (negotiate_native): every rank must build its communicator and pass a verification exchange, otherwise all
    ranks fall back to torch.distributed with a message on stderr.  "gloo" runs (CPU tests, several ranks on one GPU) always use
    torch.distributed: RCCL cannot form a communicator with two ranks on one device."""
    global _native, _native_tried
    if _native is not None or _native_tried:
        return _native
    if world_size() < 2:
        return None
    _native_tried = True
    mode = os.environ.get("SG_NATIVE_ALLREDUCE", "1")
    if mode == "0":
        TRANSPORT.update(name="torch-" + dist.get_backend(), reason="SG_NATIVE_ALLREDUCE=0")
        return None
    # SG_NATIVE_ALLREDUCE=force: negotiate the C-ABI exchange under any backend (torch.distributed stays the out-of-band and
    # fallback channel).  With several ranks on ONE device (the gloo rehearsal on a 1-GPU box) RCCL refuses the communicator
    # ("duplicate GPU") on every rank — which is exactly the loud, uniform fallback the negotiation exists for, and how it is tested.
    if not torch.cuda.is_available() or (dist.get_backend() != "nccl" and mode != "force"):
        TRANSPORT.update(name="torch-" + dist.get_backend(), reason="backend is not nccl")
        return None
    _native, reason = negotiate_native(NativeComm, _verify_against_torch)
    if _native is None:
        TRANSPORT.update(name="torch-" + dist.get_backend(), reason="fallback: " + reason)
    else:
        from . import lib as L
        TRANSPORT.update(name="native-rccl", reason="", comm_init_seconds=round(_native.init_seconds, 3), **_native.info())
        TRANSPORT.update(L.COMM_INFO)     # which RCCL file was bound, header version vs runtime version
    return _native


class _ExposedCommTime(object):
    """How long the COMPUTE stream actually stood still for gradient exchanges (bench.py: comm.exposed_ms_per_step): one event on
    the compute stream where it reaches a bucket's wait, one right behind the wait — their distance is what the overlap did not
    hide (0 when the all-reduce had finished before backward did).  Next to it the time the HOST spent inside the wait calls
    (a gloo exchange blocks the host, an RCCL one only enqueues).  Off unless `enable()`d: two event records per exchange."""

    def __init__(self):
        self.enabled = False
        self.pairs = []
        self.host_s = 0.0

    def enable(self, on=True):
        self.enabled = bool(on)
        self.reset()

    def reset(self):
        self.pairs, self.host_s = [], 0.0

    def begin(self):
        if not self.enabled:
            return None
        a = torch.cuda.Event(enable_timing=True) if torch.cuda.is_available() else None
        if a is not None:
            a.record()
        return (a, time.perf_counter())

    def end(self, probe):
        if probe is None:
            return
        a, t0 = probe
        self.host_s += time.perf_counter() - t0
        if a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            self.pairs.append((a, b))

    def totals(self):
        """(compute-stream milliseconds, host milliseconds, exchanges waited for) since the last reset; synchronises the device."""
        if self.pairs:
            torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self.pairs), self.host_s * 1e3, len(self.pairs)


EXPOSED = _ExposedCommTime()


class GradBucket(object):
    """Gradient exchange of one optimizer: its flat fp32 gradient buffer is summed across ranks in (at most) two
    contiguous slices.  The TAIL slice holds the parameters whose gradients autograd finishes FIRST (the deepest
    layers, which come last in parameter order): its all-reduce is launched from a post-accumulate-grad hook as soon
    as those gradients are complete and runs on RCCL's stream while the remaining backward kernels execute; the HEAD
    slice goes out when backward returns.  `wait()` blocks the compute stream on both.  With one process everything
    is a no-op.

    Call pattern per optimizer step:   bucket.arm(); loss.backward(); bucket.finish(); optimizer.step()
    (`allreduce()` = arm-less variant: one exchange of the whole buffer after backward.)"""

    def __init__(self, optimizer, overlap=True, tail_fraction=0.5):
        self.opt = optimizer
        self.works = []
        self.armed = False
        self.pending = 0
        self.tail = None            # (start, end) element range of the early slice
        self.tail_params = []
        self.hooks = []
        self.native = native_comm()
        self._pattern = None        # which parameters had a gradient at the last exchange (this rank)
        self._pattern_dev = None    # the same as floats on the device: copied into the header before every exchange
        optimizer.grad_scale = 1.0 / world_size()
        if overlap and world_size() > 1:
            self._plan(tail_fraction)

    def _plan(self, tail_fraction):
        f = self.opt.f
        total = f.total
        # the largest suffix of the parameter list that is at least `tail_fraction` of the bytes (but not everything)
        start_idx = len(f.params)
        for i in range(len(f.params) - 1, 0, -1):
            start_idx = i
            if total - f.offsets[i] >= tail_fraction * total:
                break
        if start_idx >= len(f.params) or start_idx == 0:
            return
        self.tail = (f.offsets[start_idx], total)
        self.tail_index = list(range(start_idx, len(f.params)))
        self.tail_params = f.params[start_idx:]
        for p in self.tail_params:
            self.hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, param):
        """Post-accumulate hook of a tail parameter (autograd runs it once per backward and parameter, after all of the
        parameter's contributions have been summed)."""
        if not self.armed:
            return
        self.pending -= 1
        if self.pending == 0:
            self.armed = False
            f = self.opt.f
            # a gradient that did not come out of a direct write (stock autograd tensors, several contributions summed by
            # the engine) is copied into its slice now, so that the slice is what goes on the wire
            f.adopt_grads(self.tail_index)
            a, b = self.tail
            self._exchange(self.opt.flat_grad[a:b])
            self.tail_done = True

    def arm(self):
        """Call right before loss.backward(): the tail slice will be exchanged from inside backward."""
        self.tail_done = False
        if self.tail is not None:
            self.pending = sum(1 for p in self.tail_params if p.requires_grad)
            self.armed = self.pending > 0

    def finish(self):
        """Call right after loss.backward(): exchanges whatever has not gone out yet and waits."""
        self.armed = False
        check = False
        if world_size() > 1:
            self._check_previous_header()     # what the LAST exchange's summed header said, before entering another collective
            f = self.opt.f
            first = self._pattern is None
            # Only the FIRST exchange of a bucket is validated synchronously: every rank is at its first exchange at the same
            # program point, so every rank reads the same summed header and raises (or not) together.  A pattern that changes
            # later on ONE rank must not make that rank raise alone — with two buckets per trainer the others would already be
            # inside the next bucket's collective, which the raising rank never joins.  Every rank finds such a
            # disagreement in the snapshot of the summed header at the top of this bucket's NEXT finish(): identical data on
            # every rank, identical decision at the identical program point, nobody is left behind in a collective.
            check = self._stage_pattern() and first
            h = f.header_len
            if getattr(self, "tail_done", False):
                a, _ = self.tail
                f.adopt_grads(range(0, self.tail_index[0]))
                # tail parameters must still be the views that were reduced (a later out-of-place accumulation would have
                # left the reduced slice behind)
                if not all(f.grad_view_ok(i) for i in self.tail_index):
                    raise RuntimeError("GradBucket: a gradient changed after its slice was exchanged")
                self._exchange(f.grad_store[:h + a])          # header + head slice: one contiguous piece
            else:
                # module.zero_grad() (grads -> None), stock autograd tensors or an out-of-place accumulation leave p.grad
                # outside the flat buffer: the optimizer would step on p.grad while the exchange summed a stale slice
                f.adopt_grads()
                self._exchange(f.grad_store)
        self.tail_done = False
        self.wait()
        if check:
            self._check_pattern()
        elif world_size() > 1:
            self._snapshot_header()

    def _stage_pattern(self):
        """Writes "this rank has a gradient for parameter i" into the header that travels with the head slice.  A parameter
        without a gradient is skipped by step() (torch.optim semantics); if another rank HAS a gradient for it, that rank would
        step it with the summed gradient and the replicas would diverge silently.  The summed header says whether the
        ranks agree: every entry must come back as 0 or world_size.  Steady state costs one tiny device copy per exchange and one
        asynchronous copy of the summed header to pinned host memory.  The first exchange is read back at once (a host
        synchronisation on every rank, at the same program point); from then on EVERY rank — also the one whose own pattern
        just changed, e.g. at a progressive-GAN stage switch — validates the snapshot of the summed header at the start of this
        bucket's next finish(), before it enters another collective: the decision is taken from identical data at an identical
        program point on all ranks, so they raise together whatever other buckets' collectives lie in between.
        Returns whether this rank's pattern differs from its previous exchange."""
        f = self.opt.f
        pattern = tuple(p.grad is not None for p in f.params)
        changed = pattern != self._pattern
        if changed:
            host = torch.zeros(f.header_len, dtype=torch.float32)
            host[:len(pattern)] = torch.tensor(pattern, dtype=torch.float32)
            self._pattern_dev = host.to(f.header.device)
            self._pattern = pattern
        f.header.copy_(self._pattern_dev)
        return changed

    def _snapshot_header(self):
        """The summed header of the exchange that just completed, on its way to the host without blocking it."""
        f = self.opt.f
        if f.header.is_cuda:
            if getattr(self, "_header_host", None) is None:
                self._header_host = torch.empty(f.header_len, dtype=torch.float32).pin_memory()
                self._header_event = torch.cuda.Event()
            self._header_host.copy_(f.header, non_blocking=True)
            self._header_event.record()
        else:
            self._header_host = f.header.clone()
            self._header_event = None
        self._header_pending = True

    def _check_previous_header(self):
        if getattr(self, "_header_pending", False):
            self._header_pending = False
            if self._header_event is not None:
                self._header_event.synchronize()       # recorded a whole step ago
            self._validate(self._header_host[:len(self.opt.f.params)].tolist(), late=True)

    def _check_pattern(self):
        f = self.opt.f
        self._validate(f.header[:len(f.params)].cpu().tolist(), late=False)

    def _validate(self, got, late):
        w = float(world_size())
        bad = [i for i, v in enumerate(got) if v != 0.0 and v != w]
        if bad:
            raise RuntimeError("GradBucket: the ranks disagree on which parameters have a gradient (parameter indices %s: %s of %d "
                               "ranks have one); a rank without it would skip the update the others apply and the replicas "
                               "would diverge%s" % (bad[:8], [int(got[i]) for i in bad[:8]], int(w),
                                                    " — found in the header of the PREVIOUS exchange: that update has been "
                                                    "applied on this rank" if late else ""))

    def _exchange(self, t):
        """In-place SUM of a slice of the flat gradient buffer across ranks, asynchronous w.r.t. the compute stream."""
        if self.native is not None:
            self.native.launch(t)
            self.works.append(None)
        else:
            self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        if not self.works:
            return
        probe = EXPOSED.begin()
        if self.native is not None:
            self.native.wait()
        for w in self.works:
            if w is not None:
                w.wait()
        EXPOSED.end(probe)
        self.works = []

    def allreduce(self):
        self.finish()


def allreduce_tensor_(t):
    """In-place SUM of an arbitrary tensor across ranks (e.g. the dense latent-table gradient of the auto-decoder)."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def broadcast_parameters(module, src=0):
    """Makes replicas bit-identical once at start-up (only needed when ranks were not seeded identically)."""
    if world_size() > 1:
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src=src)
        # a write through .data moves neither tensor._version nor a parameter epoch: weight images derived before the broadcast
        # (kept ConvTranspose images, the SDFNet pack) would keep serving the pre-broadcast weights on the non-source ranks
        from . import lib as L
        L.bump_param_epoch()


def _on_src_then_tell_everyone(action, src, what):
    """Runs `action` on rank `src` only and broadcasts whether it worked BEFORE anybody enters the payload collective: a missing
    file or a full disk on `src` raises on every rank instead of leaving the others inside a broadcast / barrier that `src`
    never joins."""
    status, error = [None], None
    if dist.get_rank() == src:
        try:
            action()
        except Exception as e:       # noqa: BLE001 — reported on every rank below
            error = e
            status = ["%s: %s" % (type(e).__name__, e)]
    dist.broadcast_object_list(status, src=src)
    if status[0] is not None:
        if error is not None:
            raise error
        raise RuntimeError("%s failed on rank %d: %s" % (what, src, status[0]))


def save_checkpoint(module, epoch=None, src=0):
    """`SavableModule.save(epoch)` under process-per-GPU data parallelism: rank `src` writes the file, everybody waits for it.

    Which replica a checkpoint carries: parameters are bit-identical on every rank (same start, same all-reduced gradient, same
    update), so any rank's would do.  BatchNorm running statistics are NOT — every rank normalises its own shard and updates its
    own buffers (DataParallel semantics) — and the file carries those of rank `src` = 0: exactly what the reference's
    `nn.DataParallel` leaves in the module it wraps (train_hybrid_progressive_gan.py:62-68; replica 0 shares its buffers with the
    wrapped module, the other replicas' updates are dropped with the replicas).  Training-mode forwards never read them; an
    evaluation after `load_checkpoint` sees the same statistics on every rank."""
    if world_size() == 1:
        module.save(epoch=epoch)
        return
    _on_src_then_tell_everyone(lambda: module.save(epoch=epoch), src, "save_checkpoint")


def load_checkpoint(module, epoch=None, src=0):
    """`SavableModule.load(epoch)` under process-per-GPU data parallelism: rank `src` reads the file and broadcasts parameters AND
    buffers (ranks need not see the same file system, and nobody reads a file another rank is still writing); every rank ends
    up bit-identical to the file and with its derived weight images invalidated."""
    if world_size() == 1:
        module.load(epoch=epoch)
        return
    _on_src_then_tell_everyone(lambda: module.load(epoch=epoch), src, "load_checkpoint")
    broadcast_parameters(module, src=src)
