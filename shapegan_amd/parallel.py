"""Data parallelism for the hot path: one process per GPU, replicated parameters, batch axis sharded, ONE
all-reduce of the flat fp32 gradient buffer per network per optimizer step (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" on CPU for tests).

Replaces nn.DataParallel in train_hybrid_progressive_gan.py:62-68, which re-broadcasts 19.4 MB of parameters and
scatters up to 2.2 GB of inputs on every forward.  Here parameters stay resident (identical init by seed or load(),
identical updates afterwards), inputs are generated/loaded per rank, and the only exchange is the gradient sum; the
1/world factor is applied inside the optimizer kernel (grad_scale), not as a separate pass.
"""
import os

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


class NativeComm(object):
    """The C-ABI gradient exchange (include/shapegan_hip.h: sg_allreduce_*, libshapegan_comm.so): an RCCL communicator with
    its own stream; `launch` enqueues an in-place all-reduce(sum) of a contiguous fp32 CUDA tensor ordered after the current
    stream's work, `wait` makes the current stream wait for everything launched — no torch.distributed on the hot path.
    torch.distributed is used once, as the out-of-band channel for the 128-byte RCCL unique id."""

    def __init__(self, rank=None, world=None, device=None, unique_id=None):
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
        L.check_comm(self.lib.sg_allreduce_init(ctypes.byref(handle), rank, world, unique_id, len(unique_id), device), "init")
        self.handle, self.rank, self.world = handle, rank, world

    def launch(self, t):
        if not (t.is_cuda and t.is_contiguous() and t.dtype == torch.float32):
            raise RuntimeError("NativeComm.launch needs a contiguous fp32 CUDA tensor")
        self.L.check_comm(self.lib.sg_allreduce_launch(self.handle, t.data_ptr(), t.numel(),
                                                       torch.cuda.current_stream().cuda_stream), "launch")

    def wait(self):
        self.L.check_comm(self.lib.sg_allreduce_wait(self.handle, torch.cuda.current_stream().cuda_stream), "wait")

    def close(self):
        if self.handle is not None:
            self.lib.sg_allreduce_destroy(self.handle)
            self.handle = None


_native = None


def native_comm():
    """The process-wide NativeComm when SG_NATIVE_ALLREDUCE=1 and the tensors live on GPUs, else None (torch.distributed
    carries the exchange: "nccl" = RCCL on ROCm, "gloo" in the CPU tests)."""
    global _native
    if os.environ.get("SG_NATIVE_ALLREDUCE", "0") != "1" or not torch.cuda.is_available() or world_size() < 2:
        return None
    if dist.get_backend() != "nccl":
        return None          # several ranks on one GPU (gloo functional checks): RCCL cannot form that communicator
    if _native is None:
        _native = NativeComm()
    return _native


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
        if world_size() > 1:
            f = self.opt.f
            if getattr(self, "tail_done", False):
                a, _ = self.tail
                f.adopt_grads(range(0, self.tail_index[0]))
                # tail parameters must still be the views that were reduced (a later out-of-place accumulation would have
                # left the reduced slice behind)
                if not all(f.grad_view_ok(i) for i in self.tail_index):
                    raise RuntimeError("GradBucket: a gradient changed after its slice was exchanged")
                self._exchange(self.opt.flat_grad[:a])
            else:
                # module.zero_grad() (grads -> None), stock autograd tensors or an out-of-place accumulation leave p.grad
                # outside the flat buffer: the optimizer would step on p.grad while the exchange summed a stale slice
                f.adopt_grads()
                self._exchange(self.opt.flat_grad)
        self.tail_done = False
        self.wait()

    def _exchange(self, t):
        """In-place SUM of a slice of the flat gradient buffer across ranks, asynchronous w.r.t. the compute stream."""
        if self.native is not None:
            self.native.launch(t)
            self.works.append(None)
        else:
            self.works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True))

    def wait(self):
        if self.native is not None and self.works:
            self.native.wait()
        for w in self.works:
            if w is not None:
                w.wait()
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
