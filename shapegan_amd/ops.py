"""torch.autograd.Function shells over the C ABI (include/shapegan_hip.h).

Every Function's backward is written in terms of the other Functions here, so the set is closed under
differentiation: `autograd.grad(..., create_graph=True)` (WGAN-GP, train_hybrid_progressive_gan.py:102-111) works
without any extra kernel — convolution and matmul are bilinear, LeakyReLU's derivative is a mask.

PyTorch is used for storage (device tensors), the autograd tape and stream handles only; all arithmetic on the
hot path runs in libshapegan_hip.so.  There is no CPU fallback: tensors must live on the GPU.
"""
import ctypes
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import lib as L
from .lib import ACT_LEAKY, ACT_NONE, ACT_RELU, ACT_SIGMOID, ACT_TANH, check, f32c, ptr, stream, workspace  # noqa: F401


def _lib():
    return L.load()


# --------------------------------------------------------------------------------------------------------------
# raw launches (no autograd)
# --------------------------------------------------------------------------------------------------------------
def conv_fwd_raw(x, w, bias, act=ACT_NONE, slope=0.0, keep=False):
    """x [N,Cx,D,H,W], w [Co,Ct,4,4,4] -> act(conv_k4s2p1(x, w[:, :Cx]) + bias) [N,Co,D/2,H/2,W/2].
    keep: `w` is a layer's own weight — its packed image lives in a workspace of its own (_KeptWeightImages) and is rebuilt only
    when the weight changed, together with the other stale images of its pack group (register_pack_group)."""
    N, Cx, D, H, W = x.shape
    Co, Ct = w.shape[0], w.shape[1]
    if Cx > Ct:
        raise RuntimeError("conv: input has %d channels, weight expects %d" % (Cx, Ct))
    ptr(x)  # fail loudly on CPU tensors before anything else
    L.reset_call_state()
    y = torch.empty((N, Co, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
    lib = _lib()
    nb = lib.sg_conv3d_k4s2p1_fwd_workspace_bytes(N, Cx, Co, D // 2, H // 2, W // 2)
    if keep and Cx > 1 and x.is_cuda and L.writers_known(w) and not torch.cuda.is_current_stream_capturing():
        ws, unchanged = _KEPT.get(w, nb, (N, Cx, Ct, Cx, Co, D, H, W), kind=0)
        check(lib.sg_conv3d_k4s2p1_fwd_keep(ptr(x), ptr(w), ptr(bias), ptr(y), N, Cx, Ct, Cx, Co, D, H, W, act, slope, ptr(ws),
                                            ws.numel(), int(unchanged), stream()), "conv3d_fwd_keep")
        return y
    ws = workspace("splitk", nb, x.device) if nb else None
    check(lib.sg_conv3d_k4s2p1_fwd(ptr(x), ptr(w), ptr(bias), ptr(y), N, Cx, Ct, Cx, Co, D, H, W, act, slope, ptr(ws),
                                   ws.numel() if ws is not None else 0, stream()), "conv3d_fwd")
    return y


def conv_fwd_impl_raw(x, w, bias, act, slope, impl, debug=0):
    """Forward through a forced implementation (0 gather, 1 LDS-halo) — tests and tuning only."""
    N, Cx, D, H, W = x.shape
    Co, Ct = w.shape[0], w.shape[1]
    y = torch.empty((N, Co, D // 2, H // 2, W // 2), dtype=torch.float32, device=x.device)
    lib = _lib()
    nb = max(lib.sg_conv3d_k4s2p1_fwd_workspace_bytes(N, Cx, Co, D // 2, H // 2, W // 2), 1 << 20)
    ws = workspace("splitk", nb, x.device)
    check(lib.sg_conv3d_k4s2p1_fwd_impl(ptr(x), ptr(w), ptr(bias), ptr(y), N, Cx, Ct, Cx, Co, D, H, W, act, slope,
                                        ptr(ws), ws.numel(), impl, debug, stream()), "conv3d_fwd_impl")
    return y


class _KeptWeightImages(object):
    """Workspaces dedicated to one weight and one packed form each (sg_conv3d_k4s2p1_fwd_keep / _dgrad_keep): the image a call
    leaves there serves the next call as long as the weight is unchanged — same storage, same tensor version, same parameter epoch
    of the optimizer buffer it lives in (lib.param_epoch_of) — and the call's shapes are the same.  The WGAN generator is evaluated
    six times per 5+1 unit (train_wgan.py:60-84) and updated once.  At most `cap` images are remembered (oldest dropped).

    Pack groups (register_pack_group): the weights of one network change together (one optimizer step), so when a call finds its
    image stale, every OTHER stale image of the group whose call shapes are known from earlier calls is rebuilt in the same launch
    (sg_conv3d_k4s2p1_pack_images) — a critic update of train_wgan.py then packs its four images (two weights, forward and
    input-gradient form) with one launch instead of four."""

    def __init__(self, cap=64):
        self.cap, self.entries, self.groups, self.layouts = cap, {}, {}, {}

    def _key(self, w, kind, dims, nbytes):
        """What an image is valid FOR: the library's layout number of the call where a kept image serves it
        (sg_conv3d_k4s2p1_image_layout: the batch size does not enter the LDS-halo kernels' images, so the critic's images serve its
        128-sample update passes and the 64-sample pass of the generator update alike), the full call shape otherwise."""
        if w.device.type != "cuda":
            return dims
        q = (kind, dims, int(nbytes))
        key = self.layouts.get(q)
        if key is None:
            if len(self.layouts) > 4096:
                self.layouts.clear()
            layout = _lib().sg_conv3d_k4s2p1_image_layout(kind, (ctypes.c_int * 8)(*[int(d) for d in dims]), int(nbytes))
            key = self.layouts[q] = ("layout", layout) if layout else dims
        return key

    def _ident(self, w, kind):
        return (w.device.index, stream(), w.data_ptr(), kind)

    def _state(self, w, shape_key):
        return (w._version, L.param_epoch_of(w), shape_key)

    def get(self, w, nbytes, shape_key, kind=1):
        """-> (workspace, unchanged).  kind: 0 forward image, 1 input-gradient image; shape_key = (batch, Cin, Cin_total, Cx, Cout,
        ID, IH, IW) of the call.  An entry belongs to one tensor OBJECT (weak reference): a new tensor that the allocator places at
        a freed weight's address, with the same version and epoch, is a different weight."""
        ident = self._ident(w, kind)
        ent = self.entries.pop(ident, None)
        if ent is None or ent[0].numel() < nbytes or ent[2]() is not w:
            ent = [torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=w.device), None, weakref.ref(w), None]
        # (the layout is asked for the workspace the call is really given: an entry sized by an earlier, larger call keeps its size)
        state = self._state(w, self._key(w, kind, shape_key, ent[0].numel()))
        unchanged = ent[1] == state
        if not unchanged:
            unchanged = self._pack_group(w, kind, ent, shape_key)
        ent[1], ent[3] = state, shape_key        # ent[3]: the shapes of the entry's latest call (what a group launch plans with)
        self.entries[ident] = ent            # (re-inserted last: dict order = age)
        while len(self.entries) > self.cap:
            self.entries.pop(next(iter(self.entries)))
        return ent[0], unchanged

    def _pack_group(self, w, kind, ent, shape_key):
        """The stale image (w, kind) and every other stale image of w's pack group with known call shapes, in one launch.  Returns
        whether (w, kind) itself is in place now."""
        group = self.groups.get(id(w))
        if group is None or group[1]() is not w:
            return False
        jobs = [(w, kind, ent, shape_key)]
        for ref in group[0]:
            v = ref()
            if v is None or v.device != w.device:
                continue
            for k2 in (0, 1):
                if v is w and k2 == kind:
                    continue
                e2 = self.entries.get(self._ident(v, k2))
                if e2 is None or e2[2]() is not v or e2[1] is None:
                    continue
                if e2[1] != self._state(v, e2[1][2]):          # stale, shapes known from its last call
                    jobs.append((v, k2, e2, e2[3]))
        if len(jobs) < 2:
            return False                     # nothing to share the launch with: the call packs its own image as before
        jobs = jobs[:8]
        n = len(jobs)
        kinds = (ctypes.c_int * n)(*[j[1] for j in jobs])
        wptr = (ctypes.c_void_p * n)(*[j[0].data_ptr() for j in jobs])
        wsp = (ctypes.c_void_p * n)(*[j[2][0].data_ptr() for j in jobs])
        wsb = (ctypes.c_size_t * n)(*[j[2][0].numel() for j in jobs])
        dims = (ctypes.c_int * (8 * n))(*[int(d) for j in jobs for d in j[3]])
        served = (ctypes.c_int * n)()
        L.note_device(w)
        check(_lib().sg_conv3d_k4s2p1_pack_images(n, kinds, wptr, wsp, wsb, dims, served, stream()), "conv3d_pack_images")
        for j, ok in zip(jobs[1:], list(served)[1:]):
            j[2][1] = self._state(j[0], self._key(j[0], j[1], j[3], j[2][0].numel())) if ok else None
        return bool(served[0])

    def register_group(self, weights):
        refs = [weakref.ref(w) for w in weights]
        for w in weights:
            self.groups[id(w)] = (refs, weakref.ref(w))


_KEPT = _KeptWeightImages()


def invalidate_weight_images():
    """For code that writes parameters behind the library's back — `p.data.mul_(...)`, `p.data.copy_(...)` on a parameter that
    lives in a shapegan_amd.optim flat buffer (such writes move neither `tensor._version` nor a parameter epoch): every image
    derived from any weight (kept conv images, the SDFNet pack) is rebuilt at its next use.  Parameters outside a flat buffer
    (stock torch.optim, the reference's scripts) need no call: their images are rebuilt on every call."""
    L.bump_param_epoch()


def register_pack_group(weights):
    """Declares conv weights that change together (the parameters of one network): their kept images are rebuilt in one launch."""
    _KEPT.register_group(list(weights))


def conv_dgrad_raw(dy, w, bias, cin, act=ACT_NONE, slope=0.0, keep=False, out=None):
    """dy [N,Co,O,O,O], w [Co,Ct,4,4,4] -> act(conv^T(dy, w[:, :cin]) + bias) [N,cin,2O,2O,2O].
    keep: `w` is a layer's own weight (ConvTranspose3d forward) — its packed image is kept for the next call (_KeptWeightImages).
    out: where to write the result (a contiguous fp32 tensor of the result's shape, e.g. one half of the critic's batch)."""
    N, Co, OD, OH, OW = dy.shape
    Ct = w.shape[1]
    if w.shape[0] != Co or cin > Ct:
        raise RuntimeError("conv dgrad: shape mismatch")
    shape = (N, cin, 2 * OD, 2 * OH, 2 * OW)
    if out is not None:
        if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dy.device:
            raise RuntimeError("conv dgrad: `out` must be a contiguous fp32 tensor of shape %s on %s" % (shape, dy.device))
        dx = out
    else:
        dx = torch.empty(shape, dtype=torch.float32, device=dy.device)
    lib = _lib()
    nb = lib.sg_conv3d_k4s2p1_dgrad_workspace_bytes_for(N, cin, Co, OD, OH, OW)
    # (one-channel layers pack nothing; a graph under capture must contain its own packing launch: its replays see new weights)
    if keep and cin > 1 and dy.is_cuda and L.writers_known(w) and not torch.cuda.is_current_stream_capturing():
        ws, unchanged = _KEPT.get(w, nb, (N, cin, Ct, cin, Co, 2 * OD, 2 * OH, 2 * OW), kind=1)
        check(lib.sg_conv3d_k4s2p1_dgrad_keep(ptr(dy), ptr(w), ptr(bias), ptr(dx), N, cin, Ct, cin, Co, 2 * OD, 2 * OH, 2 * OW,
                                              act, slope, ptr(ws), ws.numel(), int(unchanged), stream()), "conv3d_dgrad_keep")
        return dx
    ws = workspace("dgrad", nb, dy.device)
    check(lib.sg_conv3d_k4s2p1_dgrad(ptr(dy), ptr(w), ptr(bias), ptr(dx), N, cin, Ct, cin, Co, 2 * OD, 2 * OH, 2 * OW,
                                     act, slope, ptr(ws), ws.numel(), stream()), "conv3d_dgrad")
    return dx


def conv_dgrad_halo_raw(dy, w, bias, cin, act=ACT_NONE, slope=0.0, impl=1):
    """dgrad through the forced LDS-halo kernel (impl 3: one output parity per workgroup) — tests and tuning only."""
    N, Co, OD, OH, OW = dy.shape
    Ct = w.shape[1]
    dx = torch.empty((N, cin, 2 * OD, 2 * OH, 2 * OW), dtype=torch.float32, device=dy.device)
    lib = _lib()
    ws = workspace("dgrad", lib.sg_conv3d_k4s2p1_dgrad_workspace_bytes(Co, cin), dy.device)
    check(lib.sg_conv3d_k4s2p1_dgrad_impl(ptr(dy), ptr(w), ptr(bias), ptr(dx), N, cin, Ct, cin, Co, 2 * OD, 2 * OH,
                                          2 * OW, act, slope, ptr(ws), ws.numel(), impl, stream()), "conv3d_dgrad_impl")
    return dx


_WGRAD_WS_CAP = 256 << 20


def _param_grad_out(param, shape, device):
    """Where a parameter gradient of a plain backward goes: the parameter's slice of its optimizer's flat gradient buffer
    when that is open (lib.grad_destination: autograd then adopts the view as p.grad, no accumulation pass), else a new
    tensor."""
    dst = L.grad_destination(param, shape) if param is not None else None
    return dst if dst is not None else torch.empty(shape, dtype=torch.float32, device=device)


def conv_wgrad_raw(dy, x, ct, out=None):
    """dy [N,Co,O,O,O], x [N,Cx,2O,2O,2O] -> dw [Co,ct,4,4,4] (channels >= Cx are zero)."""
    N, Co, OD, OH, OW = dy.shape
    Cx = x.shape[1]
    if x.shape[0] != N or x.shape[2] != 2 * OD:
        raise RuntimeError("conv wgrad: shape mismatch")
    if out is not None:
        dw = out
        if Cx < ct:
            dw.zero_()
    elif Cx < ct:
        dw = torch.zeros((Co, ct, 4, 4, 4), dtype=torch.float32, device=dy.device)
    else:
        dw = torch.empty((Co, ct, 4, 4, 4), dtype=torch.float32, device=dy.device)
    lib = _lib()
    nb = min(lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(N, Cx, Co, OD, OH, OW), _WGRAD_WS_CAP)
    ws = workspace("splitk", nb, dy.device)
    check(lib.sg_conv3d_k4s2p1_wgrad(ptr(dy), ptr(x), ptr(dw), N, Cx, ct, Cx, Co, 2 * OD, 2 * OH, 2 * OW, ptr(ws),
                                     ws.numel(), stream()), "conv3d_wgrad")
    return dw


def _wgrad_dy_image(N, Cx, Co, O, device):
    """(workspace, mt_total, nslice) if the weight-gradient call dy [N, Co, O^3] x [N, Cx, (2O)^3] is served with a packed-dy image
    its PRODUCER writes (sg_conv3d_k4s2p1_wgrad_dy_image), else None."""
    if device.type != "cuda":
        return None
    lib = _lib()
    nb = min(lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(N, Cx, Co, O, O, O), _WGRAD_WS_CAP)
    ws = workspace("splitk", nb, device)
    mt, ns = ctypes.c_int(0), ctypes.c_long(0)
    if not lib.sg_conv3d_k4s2p1_wgrad_dy_image(N, Cx, Co, O, O, O, ws.numel(), ctypes.byref(mt), ctypes.byref(ns)):
        return None
    return ws, mt.value, ns.value


def conv_wgrad_prepacked_raw(dy, x, ct, ws, out=None):
    """conv_wgrad_raw for a call whose packed-dy image is already at the start of `ws` (written by dy's producer)."""
    N, Co = dy.shape[0], dy.shape[1]
    Cx = x.shape[1]
    dw = out if out is not None else torch.empty((Co, ct, 4, 4, 4), dtype=torch.float32, device=dy.device)
    if Cx < ct:
        raise RuntimeError("conv_wgrad_prepacked: partial input channels are not served")
    check(_lib().sg_conv3d_k4s2p1_wgrad_prepacked(ptr(dy), ptr(x), ptr(dw), N, Cx, ct, Cx, Co, x.shape[2], x.shape[3], x.shape[4],
                                                  ptr(ws), ws.numel(), stream()), "conv3d_wgrad_prepacked")
    return dw


def act_bwd_rowsum_pack8_raw(y, dy, act, slope, ws, nslice, gb_out=None):
    """act_bwd_rowsum_raw for [N, C, 8, 8, 8] tensors that also writes dz in the weight-gradient kernel's fragment order to `ws`."""
    y, dy = f32c(y), f32c(dy)
    N, C = y.shape[0], y.shape[1]
    lib = _lib()
    dz = torch.empty_like(y)
    rows = torch.empty(N * C, dtype=torch.float32, device=y.device)
    check(lib.sg_act_bwd_rowsum_pack8(ptr(y), ptr(dy), ptr(dz), ptr(rows), ptr(ws), N, C, nslice, act, slope, stream()),
          "act_bwd_rowsum_pack8")
    gb = torch.empty(C, dtype=torch.float32, device=y.device) if gb_out is None else gb_out
    check(lib.sg_colsum(ptr(rows), ptr(gb), N, C, C, stream()), "colsum")
    return dz, gb


def conv_wgrad_act_raw(dy, y, x, act, slope, dw_out=None, db_out=None):
    """(dw, db) of y = act(conv(x) + b) from dy = dLoss/dy in one pass (sg_conv3d_k4s2p1_wgrad_act)."""
    N, Co, OD, OH, OW = dy.shape
    Cx = x.shape[1]
    dw = dw_out if dw_out is not None else torch.empty((Co, Cx, 4, 4, 4), dtype=torch.float32, device=dy.device)
    db = db_out if db_out is not None else torch.empty(Co, dtype=torch.float32, device=dy.device)
    lib = _lib()
    nb = min(lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(N, Cx, Co, OD, OH, OW), _WGRAD_WS_CAP)
    ws = workspace("splitk", nb, dy.device)
    check(lib.sg_conv3d_k4s2p1_wgrad_act(ptr(dy), ptr(y), ptr(x), ptr(dw), ptr(db), N, Cx, Cx, Cx, Co, 2 * OD, 2 * OH, 2 * OW, act,
                                         slope, ptr(ws), ws.numel(), stream()), "conv3d_wgrad_act")
    return dw, db


def _wgrad_act_served(x, w, y, act):
    """Whether sg_conv3d_k4s2p1_wgrad_act takes this layer: the shape is served AND its scratch (padded grid + partials) fits
    under the workspace cap conv_wgrad_act_raw allocates with — otherwise the two-pass path (activation backward, then the plain
    weight gradient, which falls through to the halo / gather kernels) is used."""
    lib = _lib()
    N, Cx, Co = x.shape[0], x.shape[1], w.shape[0]
    if not lib.sg_conv3d_k4s2p1_wgrad_act_eligible(N, Cx, Co, y.shape[2], y.shape[3], y.shape[4], act):
        return False
    return lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(N, Cx, Co, y.shape[2], y.shape[3], y.shape[4]) <= _WGRAD_WS_CAP


def conv_wgrad_halo_raw(dy, x, ct):
    """wgrad through the forced LDS-halo kernel — tests and tuning only."""
    N, Co, OD, OH, OW = dy.shape
    Cx = x.shape[1]
    dw = torch.zeros((Co, ct, 4, 4, 4), dtype=torch.float32, device=dy.device)
    lib = _lib()
    ws = workspace("splitk", lib.sg_conv3d_k4s2p1_wgrad_workspace_bytes(N, Cx, Co, OD, OH, OW), dy.device)
    check(lib.sg_conv3d_k4s2p1_wgrad_impl(ptr(dy), ptr(x), ptr(dw), N, Cx, ct, Cx, Co, 2 * OD, 2 * OH, 2 * OW, ptr(ws),
                                          ws.numel(), 1, stream()), "conv3d_wgrad_impl")
    return dw


def gemm_raw(a, ta, b, tb, bias_j=None, bias_shift=0, act=ACT_NONE, slope=0.0, out=None, a_off=0, b_off=0, M=None,
             N=None, K=None, lda=None, ldb=None, ldc=None, c_off=0):
    """out = act(op(a) @ op(b) + bias_j[j >> shift]);  a, b are 2-D row-major (leading dims lda/ldb),
    op = transpose when ta/tb.  Offsets (elements) and explicit M/N/K allow column slices of bigger matrices."""
    lda = a.shape[-1] if lda is None else lda
    ldb = b.shape[-1] if ldb is None else ldb
    if M is None:
        M = a.shape[1] if ta else a.shape[0]
    if K is None:
        K = a.shape[0] if ta else a.shape[1]
    if N is None:
        N = b.shape[0] if tb else b.shape[1]
    sai, sak = (1, lda) if ta else (lda, 1)
    sbk, sbj = (1, ldb) if tb else (ldb, 1)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    ldc = out.shape[-1] if ldc is None else ldc
    lib = _lib()
    nb = min(lib.sg_gemm_workspace_bytes(M, N), _WGRAD_WS_CAP)
    ws = workspace("splitk", nb, a.device)
    check(lib.sg_gemm(ptr(a) + 4 * a_off, sai, sak, ptr(b) + 4 * b_off, sbk, sbj, ptr(out) + 4 * c_off, ldc, 1, None,
                      ptr(bias_j), bias_shift, M, N, K, act, slope, ptr(ws), ws.numel(), stream()), "gemm")
    return out


def gemm_nt_raw(a, b, out, M, N, K, lda, ldb, ldc, a_off=0, b_off=0, c_off=0):
    """out[i, j] = sum_k a[i, k] * b[j, k] for K-contiguous operands with K >> M, N (the SDFNet weight gradients over the
    points of a batch): the dedicated LDS-staged split-K kernel."""
    lib = _lib()
    ws = workspace("gemm_nt", lib.sg_gemm_nt_workspace_bytes(M, N, K), a.device)
    check(lib.sg_gemm_nt(ptr(a) + 4 * a_off, lda, ptr(b) + 4 * b_off, ldb, ptr(out) + 4 * c_off, ldc, M, N, K, ptr(ws),
                         ws.numel(), stream()), "gemm_nt")
    return out


def act_bwd_raw(y, dy, act, slope):
    dx = torch.empty_like(y)
    check(_lib().sg_act_bwd(ptr(y), ptr(dy), ptr(dx), y.numel(), act, slope, stream()), "act_bwd")
    return dx


def act_bwd_rowsum_raw(y, dy, act, slope, gb_out=None):
    """dz = dy * act'(y) for y [N,C,*S] together with the bias gradient sum over (N, S) of dz (one pass + a tiny column sum)."""
    y, dy = f32c(y), f32c(dy)
    N, C = y.shape[0], y.shape[1]
    S = y.numel() // (N * C)
    lib = _lib()
    dz = torch.empty_like(y)
    rows = torch.empty(N * C, dtype=torch.float32, device=y.device)
    check(lib.sg_act_bwd_rowsum(ptr(y), ptr(dy), ptr(dz), ptr(rows), N * C, S, act, slope, stream()), "act_bwd_rowsum")
    gb = torch.empty(C, dtype=torch.float32, device=y.device) if gb_out is None else gb_out
    check(lib.sg_colsum(ptr(rows), ptr(gb), N, C, C, stream()), "colsum")
    return dz, gb


def act_fwd_raw(x, act, slope):
    y = torch.empty_like(x)
    check(_lib().sg_act_fwd(ptr(x), ptr(y), x.numel(), act, slope, stream()), "act_fwd")
    return y


def channel_sum_raw(g, out=None):
    """g [N,C,*S] -> [C]  (bias gradient of a convolution)."""
    N, C = g.shape[0], g.shape[1]
    S = g.numel() // (N * C)
    lib = _lib()
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=g.device)
    if S == 1:
        check(lib.sg_colsum(ptr(g), ptr(out), N, C, C, stream()), "colsum")
        return out
    tmp = torch.empty(N * C, dtype=torch.float32, device=g.device)
    check(lib.sg_rowsum(ptr(g), ptr(tmp), N * C, S, S, stream()), "rowsum")
    check(lib.sg_colsum(ptr(tmp), ptr(out), N, C, C, stream()), "colsum")
    return out


# --------------------------------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------------------------------
class ActBwd(Function):
    """dx = dy * act'(.) with the derivative read off the activation OUTPUT y.  Linear in dy, so its own
    backward w.r.t. dy is itself (this is LeakyReluBackwardBackward of the reference's GP graph)."""

    @staticmethod
    def forward(ctx, y, dy, act, slope):
        y, dy = f32c(y), f32c(dy)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        ctx.dy = dy if act in (ACT_TANH, ACT_SIGMOID) else None
        return act_bwd_raw(y, dy, act, slope)

    @staticmethod
    def backward(ctx, ggx):
        (y,) = ctx.saved_tensors
        g_dy = ActBwd.apply(y, ggx, ctx.act, ctx.slope) if ctx.needs_input_grad[1] else None
        g_y = None
        if ctx.needs_input_grad[0] and ctx.act in (ACT_TANH, ACT_SIGMOID):
            # d/dy of dy*(1-y^2) resp. dy*y*(1-y): the second-order term of a double backward through tanh / sigmoid
            # (zero almost everywhere for LeakyReLU / ReLU).  Needs dy, which forward did not have to keep otherwise.
            dy = ctx.dy
            g_y = torch.empty_like(y)
            check(_lib().sg_act_bwd_dy(ptr(y), ptr(f32c(dy.detach())), ptr(f32c(ggx.detach())), ptr(g_y), y.numel(), ctx.act,
                                       stream()), "act_bwd_dy")
        return g_y, g_dy, None, None


class Act(Function):
    """Stand-alone activation (used where no producer kernel exists to fuse into)."""

    @staticmethod
    def forward(ctx, x, act, slope):
        x = f32c(x)
        y = act_fwd_raw(x, act, slope)
        ctx.act, ctx.slope = act, slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return ActBwd.apply(y, gy, ctx.act, ctx.slope), None, None


class ChannelSum(Function):
    @staticmethod
    def forward(ctx, g):
        g = f32c(g)
        ctx.shape = g.shape
        return channel_sum_raw(g)

    @staticmethod
    def backward(ctx, gg):
        shape = ctx.shape
        view = [1, shape[1]] + [1] * (len(shape) - 2)
        return gg.reshape(view).expand(shape)


# --------------------------------------------------------------------------------------------------------------
# convolution k4 s2 p1: three primitives closed under differentiation
# --------------------------------------------------------------------------------------------------------------
class ConvFwd(Function):
    """y = act(conv3d_k4s2p1(x, w[:, :Cx]) + b) — nn.Conv3d forward and nn.ConvTranspose3d input-gradient."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope):
        x, w = f32c(x), f32c(w)
        y = conv_fwd_raw(x, w, b, act, slope, keep=True)
        ctx.act, ctx.slope, ctx.has_b = act, slope, b is not None
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, b = ctx.saved_tensors
        gb = None
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        plain = not torch.is_grad_enabled()      # no create_graph: raw kernels, parameter gradients straight into their slices
        if (ctx.act != ACT_NONE and want_b and plain and ctx.needs_input_grad[1] and not ctx.needs_input_grad[0]
                and w.shape[1] == x.shape[1] and _wgrad_act_served(x, w, y, ctx.act)):
            # the input needs no gradient (first layer of the critic): the activation backward rides in the weight-gradient
            # kernel, dz = dy * act'(y) is never written
            gw, gb = conv_wgrad_act_raw(f32c(gy), y, x, ctx.act, ctx.slope, L.grad_destination(w, w.shape),
                                        L.grad_destination(b, b.shape))
            return None, gw, gb, None, None
        image = None
        if (ctx.act in (ACT_LEAKY, ACT_RELU) and want_b and plain and ctx.needs_input_grad[1] and tuple(y.shape[2:]) == (8, 8, 8)
                and w.shape[1] == x.shape[1] and y.shape[1] % 128 == 0):
            image = _wgrad_dy_image(x.shape[0], x.shape[1], w.shape[0], 8, y.device)
        if image is not None:
            # activation backward + bias sums + the weight-gradient kernel's packed image of gz in one pass
            gz, gb = act_bwd_rowsum_pack8_raw(y, gy, ctx.act, ctx.slope, image[0], image[2], L.grad_destination(b, b.shape))
        elif ctx.act != ACT_NONE and want_b and plain and y.shape[2] * y.shape[3] * y.shape[4] >= 512:
            # activation + bias sums in one pass
            gz, gb = act_bwd_rowsum_raw(y, gy, ctx.act, ctx.slope, L.grad_destination(b, b.shape))
        else:
            gz = ActBwd.apply(y, gy, ctx.act, ctx.slope) if ctx.act != ACT_NONE else gy
        gx = None
        if ctx.needs_input_grad[0]:
            # plain backward: the raw kernel with this layer's kept input-gradient image (nothing differentiates it again)
            gx = conv_dgrad_raw(f32c(gz), w, None, x.shape[1], keep=True) if plain else ConvDgrad.apply(gz, w, None, x.shape[1], ACT_NONE, 0.0)
        gw = None
        if ctx.needs_input_grad[1]:
            if image is not None:
                gw = conv_wgrad_prepacked_raw(gz, x, w.shape[1], image[0], L.grad_destination(w, w.shape))
            else:
                gw = conv_wgrad_raw(f32c(gz), x, w.shape[1], L.grad_destination(w, w.shape)) if plain else ConvWgrad.apply(gz, x, w.shape[1])
        if want_b and gb is None:
            gb = channel_sum_raw(f32c(gz), L.grad_destination(b, b.shape)) if plain else ChannelSum.apply(gz)
        return gx, gw, gb, None, None


class ConvDgrad(Function):
    """dx = act(conv^T(dy, w[:, :cin]) + b) — nn.Conv3d input-gradient and nn.ConvTranspose3d forward."""

    @staticmethod
    def forward(ctx, dy, w, b, cin, act, slope, keep=False):
        dy, w = f32c(dy), f32c(w)
        dx = conv_dgrad_raw(dy, w, b, cin, act, slope, keep)
        ctx.act, ctx.slope, ctx.has_b = act, slope, b is not None
        ctx.save_for_backward(dy, w, dx if act != ACT_NONE else None, b)
        return dx

    @staticmethod
    def backward(ctx, gdx):
        dy, w, dx, b = ctx.saved_tensors
        g_b = None
        want_b = ctx.has_b and ctx.needs_input_grad[2]
        plain = not torch.is_grad_enabled()
        if ctx.act != ACT_NONE and want_b and plain and dx.shape[2] * dx.shape[3] * dx.shape[4] >= 512:
            gz, g_b = act_bwd_rowsum_raw(dx, gdx, ctx.act, ctx.slope, L.grad_destination(b, b.shape))
        else:
            gz = ActBwd.apply(dx, gdx, ctx.act, ctx.slope) if ctx.act != ACT_NONE else gdx
        g_dy = ConvFwd.apply(gz, w, None, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
        g_w = None
        if ctx.needs_input_grad[1]:
            g_w = conv_wgrad_raw(dy, f32c(gz), w.shape[1], L.grad_destination(w, w.shape)) if plain else ConvWgrad.apply(dy, gz, w.shape[1])
        if want_b and g_b is None:
            g_b = channel_sum_raw(f32c(gz), L.grad_destination(b, b.shape)) if plain else ChannelSum.apply(gz)
        return g_dy, g_w, g_b, None, None, None, None


class ConvWgrad(Function):
    """dw[Co,ct,4,4,4] = sum_{n,o} dy[n,co,o] * x[n,ci,2o+tap-1] (channels >= Cx stay zero)."""

    @staticmethod
    def forward(ctx, dy, x, ct):
        dy, x = f32c(dy), f32c(x)
        ctx.save_for_backward(dy, x)
        return conv_wgrad_raw(dy, x, ct)

    @staticmethod
    def backward(ctx, gdw):
        dy, x = ctx.saved_tensors
        g_dy = ConvFwd.apply(x, gdw, None, ACT_NONE, 0.0) if ctx.needs_input_grad[0] else None
        g_x = ConvDgrad.apply(dy, gdw, None, x.shape[1], ACT_NONE, 0.0) if ctx.needs_input_grad[1] else None
        return g_dy, g_x, None


def conv3d_k4s2p1(x, w, b, act=ACT_NONE, slope=0.0):
    return ConvFwd.apply(x, w, b, act, slope)


def conv_transpose3d_k4s2p1(x, w, b, act=ACT_NONE, slope=0.0, out=None):
    """nn.ConvTranspose3d(k4,s2,p1): weight [Cin_T, Cout_T, 4,4,4] is the adjoint conv's [Cout, Cin] layout.
    out (only without grad mode): the tensor the result is written to — the generator's samples go straight into the fake half
    of the critic's batch (train_wgan.py:62-66) instead of being copied there."""
    if out is not None and not torch.is_grad_enabled():
        return conv_dgrad_raw(f32c(x), f32c(w), b, w.shape[1], act, slope, True, out)
    return ConvDgrad.apply(x, w, b, w.shape[1], act, slope, True)


def convT_to1_pre_eligible(N, C, D, H, W):
    """The same question for a shape alone (model/stack.py plans a grouped pass before launching anything)."""
    return bool(_lib().sg_convT3d_k4s2p1_to1_pre_eligible(int(N), int(C), int(D), int(H), int(W)))


def convT_to1_pre_served(x, w):
    """Whether sg_convT3d_k4s2p1_to1_pre takes ConvTranspose3d(C -> 1) on x [N,C,D,H,W] (weight [C,1,4,4,4])."""
    return (w.shape[1] == 1 and x.dim() == 5 and
            bool(_lib().sg_convT3d_k4s2p1_to1_pre_eligible(x.shape[0], x.shape[1], x.shape[2], x.shape[3], x.shape[4])))


def bn_train_stats_affine(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, groups=1):
    """Batch statistics of x [N,C,*S] with torch's running-statistics update, WITHOUT writing the normalised tensor: returns
    (scale, shift) with batch_norm(x)[:, c] == x[:, c] * scale[c] + shift[c] (sg_bn_train_stats).  No autograd: inference-mode
    generator evaluations only.  groups > 1: x holds `groups` independent batches of N / groups samples (statistics per batch,
    running statistics updated batch after batch); scale / shift are [groups, C]."""
    x = f32c(x)
    N, C = x.shape[0], x.shape[1]
    if N % groups:
        raise RuntimeError("bn_train_stats_affine: %d samples in %d groups" % (N, groups))
    S = x.numel() // (N * C)
    lib = _lib()
    mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    invstd, scale, shift = torch.empty_like(mean), torch.empty_like(mean), torch.empty_like(mean)
    ws = workspace("bn", groups * lib.sg_bn_workspace_bytes(C), x.device)
    check(lib.sg_bn_train_stats_grouped(ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(running_mean), ptr(running_var),
                                        ptr(num_batches_tracked), ptr(scale), ptr(shift), groups, N // groups, C, S, eps, momentum,
                                        ptr(ws), ws.numel(), stream()), "bn_train_stats")
    return (scale, shift) if groups > 1 else (scale[0], shift[0])


def bn_train_fwd_grouped_raw(x, gamma, beta, running_mean, running_var, num_batches_tracked, eps, momentum, act, slope, groups):
    """act(batch_norm(x)) for x = `groups` independent batches stacked along dim 0, each normalised with its own batch statistics
    (sg_bn_train_fwd_grouped; running statistics updated batch after batch).  No autograd."""
    x = f32c(x)
    N, C = x.shape[0], x.shape[1]
    if N % groups:
        raise RuntimeError("bn_train_fwd_grouped: %d samples in %d groups" % (N, groups))
    S = x.numel() // (N * C)
    lib = _lib()
    y = torch.empty_like(x)
    mean = torch.empty((groups, C), dtype=torch.float32, device=x.device)
    invstd = torch.empty_like(mean)
    ws = workspace("bn", groups * lib.sg_bn_workspace_bytes(C), x.device)
    check(lib.sg_bn_train_fwd_grouped(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean),
                                      ptr(running_var), ptr(num_batches_tracked), groups, N // groups, C, S, eps, momentum, act,
                                      slope, ptr(ws), ws.numel(), stream()), "bn_train_fwd_grouped")
    return y


def conv_transpose3d_to1_pre_raw(x, scale, shift, in_act, in_slope, w, b, act=ACT_NONE, slope=0.0, out=None, outs=None, form=0):
    """act(conv_transpose3d_k4s2p1(act_in(x * scale[c] + shift[c]), w) + b) for w [C,1,4,4,4]: the BatchNorm + activation between
    the producing layer and the last transposed convolution ride in this kernel's loads (no autograd).
    outs (with scale / shift [groups, C]): x holds `groups` independent batches; batch g is written to outs[g] (equally spaced
    contiguous fp32 tensors [N / groups, 1, 2D, 2H, 2W], e.g. the fake halves of consecutive critic batches).
    form (GPU only, tests / tuning): a kernel form of sg_convT3d_k4s2p1_to1_pre_impl instead of the dispatch rule."""
    x, w = f32c(x), f32c(w)
    N, C, D, H, W = x.shape
    if outs is not None:
        groups = len(outs)
        per = N // groups
        want = (per, 1, 2 * D, 2 * H, 2 * W)
        stride = (outs[1].data_ptr() - outs[0].data_ptr()) // 4 if groups > 1 else per * 8 * D * H * W
        for g, o in enumerate(outs):
            if (tuple(o.shape) != want or o.dtype != torch.float32 or not o.is_contiguous() or o.device != x.device
                    or o.data_ptr() != outs[0].data_ptr() + 4 * stride * g):
                raise RuntimeError("conv_transpose3d_to1_pre: `outs` must be equally spaced contiguous fp32 tensors of shape %s" % (want,))
        if N % groups or tuple(scale.shape) != (groups, C) or stride < per * 8 * D * H * W:
            raise RuntimeError("conv_transpose3d_to1_pre: %d samples, %d groups, scale %s" % (N, groups, tuple(scale.shape)))
        check(_lib().sg_convT3d_k4s2p1_to1_pre_grouped(ptr(x), ptr(w), ptr(b), ptr(outs[0]), ptr(scale), ptr(shift), in_act, in_slope,
                                                       N, C, D, H, W, act, slope, per, stride, stream()), "convT3d_to1_pre_grouped")
        return outs
    shape = (N, 1, 2 * D, 2 * H, 2 * W)
    if out is not None:
        if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != x.device:
            raise RuntimeError("conv_transpose3d_to1_pre: `out` must be a contiguous fp32 tensor of shape %s on %s" % (shape, x.device))
        y = out
    else:
        y = torch.empty(shape, dtype=torch.float32, device=x.device)
    if form:
        args = (ptr(x), ptr(w), ptr(b), ptr(y), ptr(scale), ptr(shift), in_act, in_slope, N, C, D, H, W, act, slope, int(form), stream())
        L.reset_call_state()      # (an *_impl entry has no twin: it is called on the HIP library directly)
        if not x.is_cuda:
            raise RuntimeError("conv_transpose3d_to1_pre: kernel forms exist on the GPU only")
        check(L._load_hip().sg_convT3d_k4s2p1_to1_pre_impl(*args), "convT3d_to1_pre_impl")
        return y
    check(_lib().sg_convT3d_k4s2p1_to1_pre(ptr(x), ptr(w), ptr(b), ptr(y), ptr(scale), ptr(shift), in_act, in_slope, N, C, D, H, W,
                                           act, slope, stream()), "convT3d_to1_pre")
    return y


# --------------------------------------------------------------------------------------------------------------
# the critic's tail: Conv3d(k4 s2 p1) -> activation -> Conv3d(C -> 1, k4 s1) on the 4^3 grid (model/gan.py:53-55)
# --------------------------------------------------------------------------------------------------------------
def head_dot_served(x, w_head, act):
    """Whether sg_head_dot_* takes the layer pair: the producing convolution leaves a 4^3 grid, the head has ONE output channel
    over all of it, and the activation between them is one the kernels apply on load."""
    return (w_head.shape[0] == 1 and tuple(w_head.shape[2:]) == (4, 4, 4) and tuple(x.shape[2:]) == (8, 8, 8)
            and act in (ACT_NONE, ACT_LEAKY, ACT_RELU))


class ConvHead(Function):
    """y[n] = bh + sum_k act(z[n, k]) * wh[k],  z = conv3d_k4s2p1(x, w) + b  — the last two layers of gan.Discriminator as one
    node.  Forward: the convolution stores its PRE-activation and sg_head_dot_fwd applies the activation on load.  Backward
    (plain): ONE streaming pass (sg_head_dot_bwd) yields the gradient w.r.t. z, the head's weight and bias gradients and the
    convolution's bias gradient; then the convolution's input / weight gradients as usual.  With create_graph the backward is
    composed of the differentiable Functions of this module instead (closed under differentiation like everything here)."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, wh, bh):
        x, w, wh = f32c(x), f32c(w), f32c(wh)
        z = conv_fwd_raw(x, w, b, ACT_NONE, 0.0, keep=True)
        N, C = z.shape[0], z.shape[1]
        y = torch.empty(N, dtype=torch.float32, device=x.device)
        check(_lib().sg_head_dot_fwd(ptr(z), ptr(wh), ptr(bh), ptr(y), N, C * 64, act, slope, stream()), "head_dot_fwd")
        ctx.cfg = (act, slope, b is not None, bh is not None)
        ctx.save_for_backward(x, w, b, wh, bh, z)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, b, wh, bh, z = ctx.saved_tensors
        act, slope, has_b, has_bh = ctx.cfg
        need_x, need_w, need_b, need_wh, need_bh = (ctx.needs_input_grad[i] for i in (0, 1, 2, 5, 6))
        need_b, need_bh = need_b and has_b, need_bh and has_bh
        N, C = z.shape[0], z.shape[1]
        if torch.is_grad_enabled():
            # create_graph: gradients that can be differentiated again.  The pre-activation is recomputed as a graph node (the
            # saved z is an intermediate without history): the activation mask then hangs on the convolution's inputs, and a
            # parameter the penalty reaches only through that mask (this layer's bias) gets the zero gradient torch gives it
            z = ConvFwd.apply(x, w, b, ACT_NONE, 0.0)
            zf = z.reshape(N, C * 64)
            gcol = gy.reshape(N, 1)
            gz = ActBwd.apply(z, Gemm.apply(gcol, wh.reshape(1, C * 64), False, False).reshape(z.shape), act, slope) \
                if act != ACT_NONE else Gemm.apply(gcol, wh.reshape(1, C * 64), False, False).reshape(z.shape)
            gwh = Gemm.apply(gcol, Act.apply(zf, act, slope) if act != ACT_NONE else zf, True, False).reshape(wh.shape) if need_wh else None
            gbh = ColSum.apply(gcol) if need_bh else None
            gx = ConvDgrad.apply(gz, w, None, x.shape[1], ACT_NONE, 0.0) if need_x else None
            gw = ConvWgrad.apply(gz, x, w.shape[1]) if need_w else None
            gb = ChannelSum.apply(gz) if need_b else None
            return gx, gw, gb, None, None, gwh, gbh
        gy = f32c(gy)
        dev = z.device
        gz = torch.empty_like(z)
        gwh = _param_grad_out(wh, wh.shape, dev) if need_wh else None
        gbh = _param_grad_out(bh, bh.shape, dev) if need_bh else None
        gb = _param_grad_out(b, b.shape, dev) if need_b else None
        # with the convolution's weight gradient to come, gz is also written in that kernel's fragment order (no packing pass)
        image = _wgrad_dy_image(N, x.shape[1], C, 4, dev) if (need_w and C % 128 == 0 and w.shape[1] == x.shape[1]) else None
        check(_lib().sg_head_dot_bwd(ptr(z), ptr(wh), ptr(gy), ptr(gz), ptr(gwh), ptr(gbh), ptr(gb),
                                     ptr(image[0]) if image is not None else None, N, C, 64, act, slope, stream()), "head_dot_bwd")
        gx = conv_dgrad_raw(gz, w, None, x.shape[1], keep=True) if need_x else None
        gw = None
        if need_w:
            gw = conv_wgrad_prepacked_raw(gz, x, w.shape[1], image[0], L.grad_destination(w, w.shape)) if image is not None \
                else conv_wgrad_raw(gz, x, w.shape[1], L.grad_destination(w, w.shape))
        return gx, gw, gb, None, None, gwh, gbh


def conv_head(x, w, b, act, slope, wh, bh):
    return ConvHead.apply(x, w, b, act, slope, wh, bh)


# --------------------------------------------------------------------------------------------------------------
# GEMM / Linear
# --------------------------------------------------------------------------------------------------------------
class Gemm(Function):
    """C = op(a) @ op(b) for 2-D a, b."""

    @staticmethod
    def forward(ctx, a, b, ta, tb):
        a, b = f32c(a), f32c(b)
        ctx.ta, ctx.tb = ta, tb
        ctx.save_for_backward(a, b)
        return gemm_raw(a, ta, b, tb)

    @staticmethod
    def backward(ctx, gc):
        a, b = ctx.saved_tensors
        ta, tb = ctx.ta, ctx.tb
        ga = gb = None
        if ctx.needs_input_grad[0]:
            if not ta:
                ga = Gemm.apply(gc, b, False, not tb)       # gC op(b)^T
            else:
                ga = Gemm.apply(b, gc, tb, True)            # op(b) gC^T
        if ctx.needs_input_grad[1]:
            if not tb:
                gb = Gemm.apply(a, gc, not ta, False)       # op(a)^T gC
            else:
                gb = Gemm.apply(gc, a, True, ta)            # gC^T op(a)
        return ga, gb, None, None


def colsum_tall_raw(x, batch, batch_stride, rows, cols, ld):
    """out[b][c] = sum_r x[b*batch_stride + r*ld + c] (two deterministic passes; rows = points)."""
    lib = _lib()
    out = torch.empty((batch, cols), dtype=torch.float32, device=x.device)
    ws = workspace("colsum", lib.sg_colsum_tall_workspace_bytes(batch, rows, cols), x.device)
    check(lib.sg_colsum_tall(ptr(x), ptr(out), batch, batch_stride, rows, cols, ld, ptr(ws), ws.numel(), stream()),
          "colsum_tall")
    return out


def _colsum_raw(g, out=None):
    """Column sums of a 2-D tensor (bias gradient of a Linear layer)."""
    if g.shape[0] > 256:   # per-point layers: one thread per column walking every row would serialise
        res = colsum_tall_raw(g, 1, 0, g.shape[0], g.shape[1], g.shape[1])[0]
        if out is not None:
            out.copy_(res)
            return out
        return res
    if out is None:
        out = torch.empty(g.shape[1], dtype=torch.float32, device=g.device)
    check(_lib().sg_colsum(ptr(g), ptr(out), g.shape[0], g.shape[1], g.shape[1], stream()), "colsum")
    return out


class ColSum(Function):
    @staticmethod
    def forward(ctx, g):
        g = f32c(g)
        ctx.rows = g.shape[0]
        return _colsum_raw(g)

    @staticmethod
    def backward(ctx, gg):
        return gg.unsqueeze(0).expand(ctx.rows, gg.shape[0])


class LinearAct(Function):
    """y = act(x @ W' + bias) with W' = w^T (nn.Linear layout [out,in]) or w (layout [in,out], w_kn=True);
    bias[j >> bias_shift] lets one bias entry cover 64 taps of a 1^3 -> 4^3 ConvTranspose (model/gan.py:9)."""

    @staticmethod
    def forward(ctx, x, w, b, act, slope, w_kn, bias_shift):
        x, w = f32c(x), f32c(w)
        y = gemm_raw(x, False, w, not w_kn, bias_j=b, bias_shift=bias_shift, act=act, slope=slope)
        ctx.cfg = (act, slope, w_kn, bias_shift, b is not None)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y, b = ctx.saved_tensors
        act, slope, w_kn, bias_shift, has_b = ctx.cfg
        gz = ActBwd.apply(y, gy, act, slope) if act != ACT_NONE else gy
        gx = gw = gb = None
        plain = not torch.is_grad_enabled()
        if ctx.needs_input_grad[0]:
            gx = Gemm.apply(gz, w, False, w_kn)             # gz @ w  (or gz @ w^T when w is [in,out])
        if ctx.needs_input_grad[1]:
            if plain:
                gzc = f32c(gz)
                a, bm = (x, gzc) if w_kn else (gzc, x)
                gw = gemm_raw(a, True, bm, False, out=L.grad_destination(w, w.shape))
            else:
                gw = Gemm.apply(x, gz, True, False) if w_kn else Gemm.apply(gz, x, True, False)
        if has_b and ctx.needs_input_grad[2]:
            if plain:
                if bias_shift:   # one bias entry per group of 2^shift columns: column sums, then sums over each group
                    cols = _colsum_raw(f32c(gz))
                    gb = _colsum_raw(f32c(cols.reshape(-1, 1 << bias_shift).t()), L.grad_destination(b, b.shape))
                else:
                    gb = _colsum_raw(f32c(gz), L.grad_destination(b, b.shape))
            else:
                gb = ColSum.apply(gz)
                if bias_shift:
                    gb = ColSum.apply(gb.reshape(-1, 1 << bias_shift).t())
        return gx, gw, gb, None, None, None, None


def linear(x, w, b, act=ACT_NONE, slope=0.0):
    return LinearAct.apply(x, w, b, act, slope, False, 0)


# --------------------------------------------------------------------------------------------------------------
# BatchNorm (+ fused activation)
# --------------------------------------------------------------------------------------------------------------
class BatchNormAct(Function):
    """act(batch_norm(x)) for x [N,C,*S]; training updates running stats in place exactly like torch
    (momentum 0.1, unbiased running_var, num_batches_tracked += 1)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches_tracked, training, eps, momentum, act,
                slope):
        x = f32c(x)
        N, C = x.shape[0], x.shape[1]
        S = x.numel() // (N * C)
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        invstd = torch.empty(C, dtype=torch.float32, device=x.device)
        lib = _lib()
        if training:
            nb = lib.sg_bn_workspace_bytes(C)
            ws = workspace("bn", nb, x.device)
            check(lib.sg_bn_train_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(invstd), ptr(running_mean),
                                      ptr(running_var), ptr(num_batches_tracked), N, C, S, eps, momentum, act, slope,
                                      ptr(ws), ws.numel(), stream()), "bn_train_fwd")
        else:
            check(lib.sg_bn_eval_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(running_mean), ptr(running_var),
                                     ptr(mean), ptr(invstd), N, C, S, eps, act, slope, stream()), "bn_eval_fwd")
        ctx.cfg = (N, C, S, bool(training), act, slope)
        ctx.save_for_backward(x, gamma, beta, mean, invstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, gamma, beta, mean, invstd = ctx.saved_tensors
        N, C, S, training, act, slope = ctx.cfg
        gy = f32c(gy)
        dx = torch.empty_like(x)
        dgamma = _param_grad_out(gamma, gamma.shape, x.device)
        dbeta = _param_grad_out(beta, beta.shape, x.device)
        lib = _lib()
        ws = workspace("bn", lib.sg_bn_workspace_bytes(C), x.device)
        check(lib.sg_bn_bwd(ptr(gy), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(invstd), ptr(dx), ptr(dgamma),
                            ptr(dbeta), N, C, S, 1 if training else 0, act, slope, ptr(ws), ws.numel(), stream()),
              "bn_bwd")
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


# --------------------------------------------------------------------------------------------------------------
# SDFNet fused MLP
# --------------------------------------------------------------------------------------------------------------
_H = 256


def _capturing(t):
    """A graph under capture must contain the weight pack even when the image happens to be current at capture time: the replays
    run after optimizer steps (possibly captured in OTHER graphs) that the capture-time check cannot see."""
    return t.is_cuda and torch.cuda.is_current_stream_capturing()


class _PackCache(object):
    """MFMA-fragment image of the 16 SDFNet tensors, rebuilt only when a parameter changed (in-place optimizer
    steps bump tensor._version).  One entry per device: nn.DataParallel replicas (train_hybrid_progressive_gan.py:62-68) are
    shallow copies that share this object and call it from one thread per device, so an entry is published as ONE tuple
    assignment and a replica never sees another device's image."""

    def __init__(self):
        self.entries = {}

    def _lookup(self, params, latent, kin_used):
        """(key, the device's image if it is still the image of these parameters, else None)."""
        ptrs = [p.data_ptr() for p in params]
        key = (kin_used, latent, L.param_epoch_of_ptrs(ptrs)) + tuple(ptrs) + tuple(p._version for p in params)
        entry = self.entries.get(params[0].device)
        # (parameters outside a flat optimizer buffer can be written through `.data` without a trace: never reuse their image)
        if entry is None or entry[0] != key or not L.writers_known(*params) or _capturing(params[0]):
            return key, None
        return key, entry[1]

    def get(self, params, latent, kin_used):
        key, packed = self._lookup(params, latent, kin_used)
        if packed is None:
            lib = _lib()
            dev = params[0].device
            packed = torch.empty(lib.sg_sdfnet_packed_floats(kin_used), dtype=torch.float32, device=dev)
            arr = (ctypes.c_void_p * 16)(*[ptr(f32c(p.detach())) for p in params])
            check(lib.sg_sdfnet_pack(arr, latent, kin_used, ptr(packed), stream()), "sdfnet_pack")
            self.entries[dev] = (key, packed)
        return packed

    def get_with_fold(self, params, z):
        """Per-shape mode: (image, zb1, zb5) for latents z [S, L] — the image rebuilt when stale, the fold zb = b + z W[:, latent
        columns]^T of layers1.0 / layers2.0 always; ONE launch when the image is stale (sg_sdfnet_pack_shape_bias: behind every
        optimizer step both are due), the fold alone otherwise, nothing when a fold made ahead (prepare_fold) is still current."""
        z = f32c(z)
        S, Lz = z.shape
        dev = z.device
        lib = _lib()
        key, packed = self._lookup(params, Lz, 3)
        if packed is not None:
            ready = self.take_fold(z, packed)
            if ready is not None:
                return (packed,) + ready
        zb1 = torch.empty((S, _H), dtype=torch.float32, device=dev)
        zb5 = torch.empty((S, _H), dtype=torch.float32, device=dev)
        if packed is None:
            packed = torch.empty(lib.sg_sdfnet_packed_floats(3), dtype=torch.float32, device=dev)
            arr = (ctypes.c_void_p * 16)(*[ptr(f32c(p.detach())) for p in params])
            check(lib.sg_sdfnet_pack_shape_bias(arr, Lz, ptr(packed), ptr(z), S, ptr(zb1), ptr(zb5), stream()), "sdfnet_pack_shape_bias")
            self.entries[dev] = (key, packed)
            self.fold = None
        else:
            check(lib.sg_sdfnet_shape_bias(ptr(z), S, Lz, ptr(f32c(params[0])), ptr(f32c(params[1])), ptr(f32c(params[8])),
                                           ptr(f32c(params[9])), ptr(zb1), ptr(zb5), stream()), "sdfnet_shape_bias")
        return packed, zb1, zb5

    def prepare_fold(self, params, z):
        """Pack (if stale) and the per-shape latent fold of `z` NOW, on the current stream, for the forward_segments / forward_shapes
        call that follows with the same latents: a trainer runs this on a side stream next to its batch sort (both are short
        launches that depend on nothing of each other).  The forward takes the result when latents and weights are still the ones
        it was made from, and computes its own otherwise."""
        z = f32c(z)
        S, Lz = z.shape
        packed = self.get(params, Lz, 3)
        w1, b1, w5, b5 = f32c(params[0]), f32c(params[1]), f32c(params[8]), f32c(params[9])
        zb1 = torch.empty((S, _H), dtype=torch.float32, device=z.device)
        zb5 = torch.empty((S, _H), dtype=torch.float32, device=z.device)
        check(_lib().sg_sdfnet_shape_bias(ptr(z), S, Lz, ptr(w1), ptr(b1), ptr(w5), ptr(b5), ptr(zb1), ptr(zb5), stream()),
              "sdfnet_shape_bias")
        # (the latents may live in a flat optimizer buffer whose kernels write through raw pointers: its parameter epoch is part of the key)
        self.fold = (z.device, z.data_ptr(), z._version, tuple(z.shape), packed.data_ptr(), L.param_epoch_of_ptrs([z.data_ptr()]), zb1, zb5)

    def take_fold(self, z, packed):
        fold, self.fold = getattr(self, "fold", None), None
        if fold is not None and fold[:6] == (z.device, z.data_ptr(), z._version, tuple(z.shape), packed.data_ptr(),
                                             L.param_epoch_of_ptrs([z.data_ptr()])):
            return fold[6], fold[7]
        return None


_PART_ROW = 14 * _H + 32      # SG_SDFNET_PARTIAL_ROW: floats per tile of the backward's partial sums

_side_streams = {}
_OVERLAP_MIN_POINTS = 65536      # side-stream overlap of short launches only in GPU-bound steps


class _SideStream(object):
    """`with _SideStream(dev) as side:` — a second HIP stream of `dev` that has waited for everything enqueued on the current one;
    leaving the block makes the current stream wait for it.  Small reductions of a backward (the finishing launch of the tile
    partials, the backward of the latent fold: 30 - 75 us) run there UNDER the weight-gradient GEMM batch instead of in front of /
    behind it; their outputs are distinct elements of the gradient tensors.  Capturable (the side stream joins and leaves the
    capture inside the block).  CPU tensors: no-op (`side` is None)."""

    def __init__(self, dev):
        self.dev = dev
        self.side = None

    def __enter__(self):
        if self.dev.type != "cuda":
            return self
        key = self.dev.index if self.dev.index is not None else torch.cuda.current_device()
        side = _side_streams.get(key)
        if side is None:
            side = _side_streams[key] = torch.cuda.Stream(device=self.dev)
        self.side, self.main = side, torch.cuda.current_stream(self.dev)
        side.wait_stream(self.main)
        return self

    def run(self):
        """Context in which launches go to the side stream (a plain no-op context for CPU tensors)."""
        import contextlib
        return torch.cuda.stream(self.side) if self.side is not None else contextlib.nullcontext()

    def __exit__(self, *exc):
        if self.side is not None:
            self.main.wait_stream(self.side)
        return False


def _sdf_partials(N, extended, dev):
    """[tiles][_PART_ROW] partial sums of one fused backward (include/shapegan_hip.h: bias_partials), tile-major."""
    return torch.empty((_lib().sg_sdfnet_bwd_blocks(N), _PART_ROW), dtype=torch.float32, device=dev)


def _sdf_param_grads(ctx_params, needs, dz, dz8, acts, ldn, N, x_parts, kin_total, bsum=None, extended=False, seg=None, side=None):
    """Weight/bias gradients from the saved dZ_l / H_l images.  x_parts: list of (tensor [N,w], col_offset, w)
    blocks of the per-point input X that are materialised row-major (points, and latents in per-point mode).
    seg: (seg_off, S, t1, t5) — the per-segment sums of dZ1 / dZ5 come out of the same finishing launch."""
    lib = _lib()
    dev = dz.device
    grads = [None] * 16

    def wgrad_from_acts(layer_dz, act_idx, out, c_off=0, ldc=_H):
        # out[o, c_off + k] = sum_p dZ[o,p] * H[k,p]
        gemm_nt_raw(dz, acts, out, _H, _H, N, ldn, ldn, ldc, a_off=layer_dz * _H * ldn, b_off=act_idx * _H * ldn, c_off=c_off)

    def wgrad_from_rows(layer_dz, rows, width, out, c_off, ldc):
        # out[o, c_off + k] = sum_p dZ[o,p] * rows[p,k]
        gemm_raw(dz, False, rows, False, out=out, a_off=layer_dz * _H * ldn, M=_H, N=width, K=N, lda=ldn,
                 ldb=rows.shape[1], ldc=ldc, c_off=c_off)

    # layers1.0 / layers2.0 take X; pieces not materialised per point are filled by the caller afterwards
    # (every column of both is written: points / latent columns below or by the caller, the hidden block by the batch)
    w1 = _param_grad_out(ctx_params[0], (_H, kin_total), dev)
    w5 = _param_grad_out(ctx_params[8], (_H, _H + kin_total), dev)
    # bias gradients: [tiles][_PART_ROW] partial sums from the fused backward -> ONE finishing launch (sg_sdfnet_bwd_finish):
    # the seven bias gradients, the layers2.6 bias gradient (sum of dz8) and, with the extended partials (sg_sdfnet_bwd given
    # the points), the layers2.6 weight gradient and the three point columns of dW1 / dW5 — each written where that
    # parameter's gradient lives — and the per-segment sums of the shape-sorted step.
    bias_idx = (1, 3, 5, 7, 9, 11, 13)                       # parameter index of the bias of dZ layer 0..6
    bouts = [_param_grad_out(ctx_params[pi], (_H,), dev) for pi in bias_idx]
    w8 = _param_grad_out(ctx_params[14], (1, _H), dev)
    b8 = _param_grad_out(ctx_params[15], (1,), dev)
    if bsum is not None:
        import contextlib
        with (side.run() if side is not None else contextlib.nullcontext()):      # (under the GEMM batch below when a side stream is given)
            arr = (ctypes.c_void_p * 7)(*[ptr(t) for t in bouts])
            ws = workspace("sdf_finish", lib.sg_sdfnet_bwd_finish_workspace_bytes(N), dev)
            seg_off, S, t1, t5 = seg if seg is not None else (None, 0, None, None)
            check(lib.sg_sdfnet_bwd_finish(ptr(dz), ptr(bsum), ldn, N, 1 if extended else 0, arr, ptr(w8), ptr(b8), ptr(w1),
                                           kin_total, ptr(w5) + 4 * _H if extended else None, _H + kin_total, ptr(seg_off), S,
                                           ptr(t1), ptr(t5), ptr(ws), ws.numel(), ptr(L.tickets("sdf_finish", dev)), stream()),
                  "sdfnet_bwd_finish")
    else:
        for layer_dz, out in enumerate(bouts):
            check(lib.sg_rowsum(ptr(dz) + 4 * layer_dz * _H * ldn, ptr(out), _H, N, ldn, stream()), "rowsum")
        rws = workspace("reduce", lib.sg_reduce_workspace_bytes(), dev)
        check(lib.sg_reduce_sum(ptr(dz8), ptr(b8), N, 1.0, ptr(rws), rws.numel(), stream()), "reduce_sum")

    def bgrad(layer_dz):
        return bouts[layer_dz]

    for rows, off, width in x_parts:
        if extended and off == 0 and width == 3:
            continue
        wgrad_from_rows(0, rows, width, w1, off, kin_total)
        wgrad_from_rows(4, rows, width, w5, _H + off, _H + kin_total)
    # the six 256 x 256 x N products dZ_l H_{l-1}^T in ONE launch (+ one finalize): layers2.0's hidden block, then layers
    # 1.2/1.4/1.6/2.2/2.4
    hidden = [_param_grad_out(ctx_params[pi], (_H, _H), dev) for pi in (2, 4, 6, 10, 12)]
    pairs = ((4, 3), (1, 0), (2, 1), (3, 2), (5, 4), (6, 5))
    outs = [(w5, _H + kin_total)] + [(hidden[i], _H) for i in range(5)]
    arr = ctypes.c_long * 6
    a_off = arr(*[ldz * _H * ldn for ldz, _ in pairs])
    b_off = arr(*[ai * _H * ldn for _, ai in pairs])
    c_off = arr(*[(o.data_ptr() - w5.data_ptr()) // 4 for o, _ in outs])
    ldcs = arr(*[ld for _, ld in outs])
    ws = workspace("gemm_nt", lib.sg_gemm_nt_batched_workspace_bytes(6, _H, _H, N), dev)
    check(lib.sg_gemm_nt_batched(ptr(dz), a_off, ldn, ptr(acts), b_off, ldn, ptr(w5), c_off, ldcs, 6, _H, _H, N, ptr(ws),
                                 ws.numel(), stream()), "gemm_nt_batched")
    grads[0], grads[8] = w1, w5
    grads[1], grads[9] = bgrad(0), bgrad(4)
    for i, (pi, (ldz, _)) in enumerate(zip((2, 4, 6, 10, 12), pairs[1:])):
        grads[pi] = hidden[i]
        grads[pi + 1] = bgrad(ldz)
    # layers2.6: W8 [1,256] (from the partials when extended), b8 [1] (from the partials, or a two-stage sum of dz8)
    if not extended:
        gemm_raw(dz8, False, acts, True, out=w8, b_off=6 * _H * ldn, M=1, N=_H, K=N, lda=N, ldb=ldn, ldc=_H)
    grads[14], grads[15] = w8, b8
    return grads


class SDFNetPoints(Function):
    """SDFNet.forward(points[N,3], latent_codes[N,L]) with reference semantics (model/sdf_net.py:56-61)."""

    @staticmethod
    def forward(ctx, cache, points, latent, grad_mode, *params):
        """grad_mode: torch.is_grad_enabled() of the CALLER.  Inside a Function's forward grad mode is always off and
        ctx.needs_input_grad is True for every parameter even under torch.no_grad(), so without it every no_grad evaluation (the
        discriminator updates of the hybrid GANs: 5 of 6 generator evaluations) would write the seven activation images
        (30 GB at 16 x 64^3 points) for a backward that never comes."""
        points, latent = f32c(points), f32c(latent)
        N, Lz = latent.shape
        kin = 3 + Lz
        lib = _lib()
        packed = cache.get(params, Lz, kin)
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad[1:])
        out = torch.empty(N, dtype=torch.float32, device=points.device)
        acts = torch.empty(lib.sg_sdfnet_acts_floats(N), dtype=torch.float32, device=points.device) if need_grad else None   # H1..H7 + sign masks
        check(lib.sg_sdfnet_fwd(ptr(points), 0, ptr(latent), None, Lz, ptr(packed), kin, None, None, 0, None, ptr(out),
                                ptr(acts), N, N, stream()), "sdfnet_fwd")
        ctx.cache, ctx.Lz = cache, Lz
        ctx.save_for_backward(points, latent, out, acts, packed, *params)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        points, latent, out, acts, packed = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        if acts is None:
            raise RuntimeError("SDFNet: backward through a forward that ran without grad mode")
        N, Lz = latent.shape
        kin = 3 + Lz
        gout = f32c(gout)
        lib = _lib()
        dev = out.device
        dz = torch.empty(7 * _H * N + 32, dtype=torch.float32, device=dev)[:7 * _H * N].view(7, _H, N)   # (+128 B: vector loads of the last row's tail)
        dz8 = torch.empty(N, dtype=torch.float32, device=dev)
        need_x = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]
        dx = torch.empty((N, kin), dtype=torch.float32, device=dev) if need_x else None
        need_p = any(ctx.needs_input_grad[4:])
        bsum = _sdf_partials(N, True, dev) if need_p else None
        check(lib.sg_sdfnet_bwd(ptr(gout), ptr(out), ptr(acts), ptr(dz), ptr(dz8), ptr(bsum), ptr(points) if need_p else None,
                                0, ptr(dx), kin, ptr(packed), kin, N, N, stream()), "sdfnet_bwd")
        grads = [None] * 16
        if need_p:
            grads = _sdf_param_grads(params, ctx.needs_input_grad[4:], dz, dz8, acts, N, N,
                                     [(points, 0, 3), (latent, 3, Lz)], kin, bsum, extended=True)
        gp = dx[:, :3] if ctx.needs_input_grad[1] else None
        gl = dx[:, 3:] if ctx.needs_input_grad[2] else None
        return (None, gp, gl, None) + tuple(grads)


_FOLD_MAX_SHAPES = 1024   # the one-launch latent fold (sg_sdfnet_shape_bias_*) walks the shapes serially per thread


class SDFNetShapes(Function):
    """Per-shape latents: out[s*pps + q] = SDFNet(points[s*pps + q], z[s]) without tiling z per point.
    The latent columns of layers1.0 / layers2.0 become per-shape biases (same products, summed in a different
    order than cat+Linear).  Replaces sample_latent_codes + generator(...) in train_hybrid_wgan.py:67-72,84-86 and
    train_hybrid_progressive_gan.py:90-96,138-139."""

    @staticmethod
    def forward(ctx, cache, points, z, pps, sid, seg_off, reg, grad_mode, *params):
        """(grad_mode: the caller's torch.is_grad_enabled(), see SDFNetPoints.forward.)  Uniform segments: sid is None, row s*pps+q uses z[s].  Ragged segments: sid[N] (int32) names each point's
        latent row and seg_off[S+1] (int64) bounds the contiguous run of every shape (points sorted by shape).
        reg: None, or (row_weight [S] or None, scale): the backward adds row_weight[s] * scale * z[s] to the latent gradient — the
        gradient of a quadratic latent regulariser evaluated elsewhere on a detached z (SDFAutoDecoderTrainer: one gradient
        contribution for the latent table, written in place, instead of two that autograd adds)."""
        points, z = f32c(points), f32c(z)
        S, Lz = z.shape
        N = points.shape[0] if sid is not None else S * pps
        if points.shape[0] != N:
            raise RuntimeError("SDFNetShapes: need points for all %d x %d samples" % (S, pps))
        kin_total = 3 + Lz
        lib = _lib()
        # zb1[s,o] = b1[o] + sum_k z[s,k] W1[o,3+k];  zb5[s,o] = b5[o] + sum_k z[s,k] W5[o,259+k]: both folds in one launch,
        # accumulated in double — the same launch as the weight pack when that is stale (behind every optimizer step), none at all
        # when _PackCache.prepare_fold made them ahead for exactly these latents and weights
        packed, zb1, zb5 = cache.get_with_fold(params, z)
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad[1:])
        out = torch.empty(N, dtype=torch.float32, device=points.device)
        acts = torch.empty(lib.sg_sdfnet_acts_floats(N), dtype=torch.float32, device=points.device) if need_grad else None   # H1..H7 + sign masks
        check(lib.sg_sdfnet_fwd(ptr(points), 0, None, None, Lz, ptr(packed), 3, ptr(zb1), ptr(zb5), pps, ptr(sid),
                                ptr(out), ptr(acts), N, N, stream()), "sdfnet_fwd")
        ctx.pps = pps
        ctx.seg_off = seg_off
        ctx.reg = None if reg is None else (None if reg[0] is None else f32c(reg[0]), float(reg[1]))
        ctx.save_for_backward(points, z, out, acts, packed, *params)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        points, z, out, acts, packed = ctx.saved_tensors[:5]
        params = ctx.saved_tensors[5:]
        if acts is None:
            raise RuntimeError("SDFNet: backward through a forward that ran without grad mode")
        S, Lz = z.shape
        pps = ctx.pps
        N = out.shape[0]
        kin_total = 3 + Lz
        gout = f32c(gout)
        lib = _lib()
        dev = out.device
        dz = torch.empty(7 * _H * N + 32, dtype=torch.float32, device=dev)[:7 * _H * N].view(7, _H, N)   # (+128 B: vector loads of the last row's tail)
        dz8 = torch.empty(N, dtype=torch.float32, device=dev)
        dx = torch.empty((N, 3), dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        need_p = any(ctx.needs_input_grad[8:])
        need_z = ctx.needs_input_grad[2]
        # (the tile partials are also what the per-shape sums of a shape-sorted batch are assembled from)
        want_partials = need_p or (need_z and ctx.seg_off is not None)
        bsum = _sdf_partials(N, need_p, dev) if want_partials else None
        check(lib.sg_sdfnet_bwd(ptr(gout), ptr(out), ptr(acts), ptr(dz), ptr(dz8), ptr(bsum), ptr(points) if need_p else None,
                                0, ptr(dx), 3, ptr(packed), 3, N, N, stream()), "sdfnet_bwd")
        grads = [None] * 16
        gz = None
        seg = None
        if need_p or need_z:
            # per-shape sums of dZ1 / dZ5: T[o, s] = sum_{p in shape s} dZ[o, p]   (each shape's points are contiguous)
            t1 = torch.empty((_H, S), dtype=torch.float32, device=dev)
            t5 = torch.empty((_H, S), dtype=torch.float32, device=dev)
            seg_off = ctx.seg_off
            if seg_off is None and need_p:
                # uniform segments (the hybrid GANs' shapes of pps grid points each): the same path as the sorted batch — the
                # per-shape sums come out of the tile partials in the finishing launch instead of two passes over the dZ1 / dZ5
                # images (2 x 42 us per generator update of train_hybrid_wgan.py:84-91)
                seg_off = torch.arange(S + 1, dtype=torch.int64, device=dev) * pps
            if seg_off is None:
                check(lib.sg_rowsum(ptr(dz), ptr(t1), _H * S, pps, pps, stream()), "rowsum")
                check(lib.sg_rowsum(ptr(dz) + 4 * 4 * _H * N, ptr(t5), _H * S, pps, pps, stream()), "rowsum")
            elif need_p:
                # interior tiles of a segment come from the backward's own per-tile partials, in the launch that also finishes
                # the bias gradients (sg_sdfnet_bwd_finish inside _sdf_param_grads): no pass over the images
                seg = (seg_off, S, t1, t5)
            else:
                ws = workspace("sdf_finish", lib.sg_sdfnet_bwd_finish_workspace_bytes(N), dev)
                check(lib.sg_sdfnet_bwd_finish(ptr(dz), ptr(bsum), N, N, 0, None, None, None, None, 0, None, 0, ptr(seg_off), S,
                                               ptr(t1), ptr(t5), ptr(ws), ws.numel(), ptr(L.tickets("sdf_finish", dev)), stream()),
                      "sdfnet_bwd_finish")
        fold = (need_p or need_z) and S <= _FOLD_MAX_SHAPES
        if fold and need_z:
            gz = _param_grad_out(z, (S, Lz), dev)       # (allocated on the main stream)
        # The finishing launch of the partials and the backward of the latent fold (both short, few workgroups) run on a side
        # stream UNDER the weight-gradient GEMM batch: they write the bias gradients, w8, the point and latent columns of dW1 / dW5
        # and gz, the GEMMs the hidden blocks — distinct elements.  Joined before this backward returns.
        # (worth it where the step is GPU-bound: at the reference's 20 000-point batch the launches are host-paced — the stream
        # switches cost the host more than the overlap saves — and inside a captured graph every fork / join edge is 6 - 12 us)
        overlap = _SideStream(dev) if (need_p and seg is not None and fold and N >= _OVERLAP_MIN_POINTS) else None
        if overlap is not None:
            overlap.__enter__()
        try:
            if need_p:
                grads = _sdf_param_grads(params, ctx.needs_input_grad[8:], dz, dz8, acts, N, N, [(points, 0, 3)], kin_total,
                                         bsum, extended=True, seg=seg, side=overlap)
            if fold:
                # backward of the latent fold in one launch: latent columns dW1[:, 3:] = T1 @ z, dW5[:, 259:] = T5 @ z and the
                # latent gradient gz = T1^T W1[:, 3:] + T5^T W5[:, 259:]
                w1, w5 = f32c(params[0]), f32c(params[8])
                import contextlib
                with (overlap.run() if overlap is not None else contextlib.nullcontext()):
                    rw, rs = ctx.reg if (ctx.reg is not None and need_z) else (None, 0.0)
                    check(lib.sg_sdfnet_shape_bias_bwd(ptr(t1), ptr(t5), S, ptr(z), Lz, ptr(w1), ptr(w5),
                                                       ptr(grads[0]) if need_p else None, ptr(grads[8]) if need_p else None,
                                                       ptr(gz) if need_z else None, ptr(rw), rs, stream()), "sdfnet_shape_bias_bwd")
        finally:
            if overlap is not None:
                overlap.__exit__(None, None, None)
        if fold:
            pass
        else:
            if need_p:
                gemm_raw(t1, False, z, False, out=grads[0], M=_H, N=Lz, K=S, lda=S, ldb=Lz, ldc=kin_total, c_off=3)
                gemm_raw(t5, False, z, False, out=grads[8], M=_H, N=Lz, K=S, lda=S, ldb=Lz, ldc=_H + kin_total,
                         c_off=_H + 3)
            if need_z:
                w1, w5 = f32c(params[0]), f32c(params[8])
                g1 = gemm_raw(t1, True, w1, False, b_off=3, M=S, N=Lz, K=_H, lda=S, ldb=kin_total)
                g5 = gemm_raw(t5, True, w5, False, b_off=_H + 3, M=S, N=Lz, K=_H, lda=S, ldb=_H + kin_total)
                gz = _param_grad_out(z, g1.shape, dev)
                check(lib.sg_axpby(ptr(g1), ptr(g5), ptr(gz), g1.numel(), 1.0, 1.0, stream()), "axpby")
                if ctx.reg is not None:      # (more shapes than the one-launch fold takes: the regulariser's gradient as a torch op)
                    rw, rs = ctx.reg
                    gz += (z * rs) if rw is None else (z * (rw * rs).unsqueeze(1))
        return (None, dx, gz, None, None, None, None, None) + tuple(grads)


class _GenPackCache(object):
    """MFMA-fragment image of an SDFGenerator's 30 tensors (lins.{0..7}.{weight,bias}, norms.{0..6}.{weight,bias}), rebuilt when a
    parameter changed; one entry per device (see _PackCache)."""

    def __init__(self):
        self.entries = {}

    def get(self, params):
        dev = params[0].device
        ptrs = [p.data_ptr() for p in params]
        key = (L.param_epoch_of_ptrs(ptrs),) + tuple(ptrs) + tuple(p._version for p in params)
        entry = self.entries.get(dev)
        if entry is None or entry[0] != key or not L.writers_known(*params) or _capturing(params[0]):
            lib = _lib()
            packed = torch.empty(lib.sg_sdfnet_packed_floats(3), dtype=torch.float32, device=dev)
            lins = (ctypes.c_void_p * 16)(*[ptr(f32c(p.detach())) for p in params[:16]])
            norms = (ctypes.c_void_p * 14)(*[ptr(f32c(p.detach())) for p in params[16:]])
            check(lib.sg_sdfgen_pack(lins, norms, ptr(packed), stream()), "sdfgen_pack")
            entry = (key, packed)
            self.entries[dev] = entry
        return entry[1]


_GEN_PART_ROW = _PART_ROW + 14 * _H      # SG_SDFGEN_PARTIAL_ROW


class SDFGenFused(Function):
    """SDFGenerator.forward (model/point_sdf_net.py:89-119) for hidden_channels 256 / num_layers 8 as ONE launch, and its backward
    as the fused backward-data kernel + one finishing launch + one weight-gradient GEMM batch: out[s * pps + q] for pos [S * pps, 3]
    and the per-shape rows zb1 = z_lin1(z) + lins.0.bias, zb5 = z_lin2(z) + lins.4.bias ([S, 256]; their two small Linear layers stay
    with the caller and autograd).  params: lins.{0..7}.{weight,bias}, norms.{0..6}.{weight,bias}; the gradients of lins.0.bias /
    lins.4.bias arrive through zb1 / zb5."""

    @staticmethod
    def forward(ctx, cache, pos, zb1, zb5, pps, eps, grad_mode, *params):
        pos, zb1, zb5 = f32c(pos), f32c(zb1), f32c(zb5)
        S = zb1.shape[0]
        N = pos.shape[0]
        if N != S * pps or zb1.shape != (S, _H) or zb5.shape != (S, _H):
            raise RuntimeError("SDFGenFused: need pos for all %d x %d samples and [S, 256] rows" % (S, pps))
        lib = _lib()
        dev = pos.device
        packed = cache.get(params)
        sid = None
        if not (pps % 128 == 0 or pps >= N):      # (tiles may straddle shapes: every point names its row)
            sid = torch.arange(S, dtype=torch.int32, device=dev).repeat_interleave(pps)
        need_grad = bool(grad_mode) and any(ctx.needs_input_grad[2:4] + ctx.needs_input_grad[7:])
        out = torch.empty(N, dtype=torch.float32, device=dev)
        acts = torch.empty(lib.sg_sdfgen_acts_floats(N), dtype=torch.float32, device=dev) if need_grad else None
        check(lib.sg_sdfgen_fwd(ptr(pos), ptr(packed), ptr(zb1), ptr(zb5), pps, ptr(sid), float(eps), ptr(out), ptr(acts), N, N,
                                stream()), "sdfgen_fwd")
        ctx.pps, ctx.S = pps, S
        ctx.save_for_backward(pos, acts, packed, *params)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        pos, acts, packed = ctx.saved_tensors[:3]
        params = ctx.saved_tensors[3:]
        if acts is None:
            raise RuntimeError("SDFGenerator: backward through a forward that ran without grad mode")
        lib = _lib()
        dev = pos.device
        N, S, pps = pos.shape[0], ctx.S, ctx.pps
        gout = f32c(gout).reshape(-1)
        dz = torch.empty(7 * _H * N + 32, dtype=torch.float32, device=dev)[:7 * _H * N].view(7, _H, N)
        dz8 = torch.empty(N, dtype=torch.float32, device=dev)
        bsum = torch.empty((lib.sg_sdfgen_bwd_blocks(N), _GEN_PART_ROW), dtype=torch.float32, device=dev)
        check(lib.sg_sdfgen_bwd(ptr(gout), ptr(acts), ptr(dz), ptr(dz8), ptr(bsum), ptr(pos), ptr(packed), N, N, stream()), "sdfgen_bwd")
        grads = [None] * 30
        w1 = _param_grad_out(params[0], (_H, 3), dev)
        w5 = _param_grad_out(params[8], (_H, _H + 3), dev)
        scratch = torch.empty((2, _H), dtype=torch.float32, device=dev)     # row sums of dZ1 / dZ5: the zb rows carry those gradients
        bouts = [scratch[0] if pi == 1 else scratch[1] if pi == 9 else _param_grad_out(params[pi], (_H,), dev)
                 for pi in (1, 3, 5, 7, 9, 11, 13)]
        w8 = _param_grad_out(params[14], (1, _H), dev)
        b8 = _param_grad_out(params[15], (1,), dev)
        ngr = [_param_grad_out(params[16 + 2 * l], (_H,), dev) for l in range(7)] + \
              [_param_grad_out(params[17 + 2 * l], (_H,), dev) for l in range(7)]
        seg_off = torch.arange(S + 1, dtype=torch.int64, device=dev) * pps
        t1 = torch.empty((_H, S), dtype=torch.float32, device=dev)
        t5 = torch.empty((_H, S), dtype=torch.float32, device=dev)
        barr = (ctypes.c_void_p * 7)(*[ptr(t) for t in bouts])
        narr = (ctypes.c_void_p * 14)(*[ptr(t) for t in ngr])
        ws = workspace("sdfgen_finish", lib.sg_sdfgen_bwd_finish_workspace_bytes(N), dev)
        check(lib.sg_sdfgen_bwd_finish(ptr(dz), ptr(bsum), N, N, barr, ptr(w8), ptr(b8), ptr(w1), 3, ptr(w5) + 4 * _H, _H + 3, narr,
                                       ptr(seg_off), S, ptr(t1), ptr(t5), ptr(ws), ws.numel(),
                                       ptr(L.tickets("sdfgen_finish", dev, 32)), stream()), "sdfgen_bwd_finish")
        # the six 256 x 256 x N products dZ_l relu(gamma xhat_{l-1} + beta)^T in one launch (+ one finalize)
        hidden = [_param_grad_out(params[pi], (_H, _H), dev) for pi in (2, 4, 6, 10, 12)]
        pairs = ((4, 3), (1, 0), (2, 1), (3, 2), (5, 4), (6, 5))
        outs = [(w5, _H + 3)] + [(hidden[i], _H) for i in range(5)]
        arr = ctypes.c_long * 6
        a_off = arr(*[ldz * _H * N for ldz, _ in pairs])
        b_off = arr(*[ai * _H * N for _, ai in pairs])
        g_off = arr(*[ai * _H for _, ai in pairs])
        c_off = arr(*[(o.data_ptr() - w5.data_ptr()) // 4 for o, _ in outs])
        ldcs = arr(*[ld for _, ld in outs])
        gws = workspace("gemm_nt", lib.sg_gemm_nt_batched_workspace_bytes(6, _H, _H, N), dev)
        check(lib.sg_gemm_nt_batched_lnrelu(ptr(dz), a_off, N, ptr(acts), b_off, N,
                                            ptr(packed) + 4 * lib.sg_sdfgen_packed_norm_offset(0),
                                            ptr(packed) + 4 * lib.sg_sdfgen_packed_norm_offset(1), g_off, ptr(w5), c_off, ldcs, 6,
                                            _H, _H, N, ptr(gws), gws.numel(), stream()), "gemm_nt_batched_lnrelu")
        grads[0], grads[8] = w1, w5
        for i, pi in enumerate((2, 4, 6, 10, 12)):
            grads[pi] = hidden[i]
        for i, pi in enumerate((1, 3, 5, 7, 9, 11, 13)):
            grads[pi] = None if pi in (1, 9) else bouts[i]
        grads[14], grads[15] = w8, b8
        for l in range(7):
            grads[16 + 2 * l], grads[17 + 2 * l] = ngr[l], ngr[7 + l]
        needs = ctx.needs_input_grad[7:]
        grads = [g if needs[i] else None for i, g in enumerate(grads)]
        gzb1 = t1.t() if ctx.needs_input_grad[2] else None
        gzb5 = t5.t() if ctx.needs_input_grad[3] else None
        return (None, None, gzb1, gzb5, None, None, None) + tuple(grads)


class GatherRowsGrouped(Function):
    """x [N, K] -> x[rows] for rows [B * C] (C rows per group; the rows of a group may repeat): PointNet.gather_points.  Its adjoint
    (ScatterRowsGrouped) adds duplicates in a fixed order without atomics, and the two are each other's adjoints — the gradient
    penalty's double backward stays deterministic."""

    @staticmethod
    def forward(ctx, x, rows, C):
        x = f32c(x)
        out = torch.empty((rows.numel(), x.shape[1]), dtype=torch.float32, device=x.device)
        check(_lib().sg_gather_rows(ptr(x), ptr(rows), ptr(out), rows.numel(), x.shape[1], stream()), "gather_rows")
        ctx.N, ctx.C = x.shape[0], C
        ctx.save_for_backward(rows)
        return out

    @staticmethod
    def backward(ctx, g):
        (rows,) = ctx.saved_tensors
        return ScatterRowsGrouped.apply(g, rows, ctx.C, ctx.N), None, None


class ScatterRowsGrouped(Function):
    @staticmethod
    def forward(ctx, g, rows, C, N):
        g = f32c(g)
        K = g.shape[1]
        dx = torch.zeros((N, K), dtype=torch.float32, device=g.device)
        check(_lib().sg_scatter_rows_grouped(ptr(g), ptr(rows), ptr(dx), rows.numel() // C, C, K, stream()), "scatter_rows_grouped")
        ctx.C = C
        ctx.save_for_backward(rows)
        return dx

    @staticmethod
    def backward(ctx, gg):
        (rows,) = ctx.saved_tensors
        return GatherRowsGrouped.apply(gg, rows, ctx.C), None, None, None


def gather_rows_grouped(x, rows, C):
    return GatherRowsGrouped.apply(x, rows, C)


class RowDot(Function):
    """out[b, c] = bias[c] + h[b, c, :] . w[c, :] — the diagonal of `h @ w.T + bias` for h [B, C, K], w [C, K]: the last layer of
    PointNet's selected-points pass (only output c of row c is used).  RowDot / RowScale / RowOuter are each other's adjoints, so
    the gradient penalty's double backward stays on these three kernels."""

    @staticmethod
    def forward(ctx, h, w, bias):
        h, w = f32c(h), f32c(w)
        B, C, K = h.shape
        out = torch.empty((B, C), dtype=torch.float32, device=h.device)
        check(_lib().sg_rowdot(ptr(h), ptr(w), ptr(f32c(bias)) if bias is not None else None, ptr(out), B, C, K, stream()), "rowdot")
        ctx.save_for_backward(h, w)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g):
        h, w = ctx.saved_tensors
        gh = RowScale.apply(g, w) if ctx.needs_input_grad[0] else None
        gw = RowOuter.apply(g, h) if ctx.needs_input_grad[1] else None
        gb = _colsum_any(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return gh, gw, gb


class RowScale(Function):
    """out[b, c, k] = g[b, c] w[c, k]."""

    @staticmethod
    def forward(ctx, g, w):
        g, w = f32c(g), f32c(w)
        B, C = g.shape
        K = w.shape[1]
        out = torch.empty((B, C, K), dtype=torch.float32, device=g.device)
        check(_lib().sg_rowscale(ptr(g), ptr(w), ptr(out), B, C, K, stream()), "rowscale")
        ctx.save_for_backward(g, w)
        return out

    @staticmethod
    def backward(ctx, G):
        g, w = ctx.saved_tensors
        return (RowDot.apply(G, w, None) if ctx.needs_input_grad[0] else None,
                RowOuter.apply(g, G) if ctx.needs_input_grad[1] else None)


class RowOuter(Function):
    """out[c, k] = sum_b g[b, c] h[b, c, k]."""

    @staticmethod
    def forward(ctx, g, h):
        g, h = f32c(g), f32c(h)
        B, C, K = h.shape
        out = torch.empty((C, K), dtype=torch.float32, device=g.device)
        check(_lib().sg_rowouter(ptr(g), ptr(h), ptr(out), B, C, K, stream()), "rowouter")
        ctx.save_for_backward(g, h)
        return out

    @staticmethod
    def backward(ctx, G):
        g, h = ctx.saved_tensors
        return (RowDot.apply(h, G, None) if ctx.needs_input_grad[0] else None,
                RowScale.apply(g, G) if ctx.needs_input_grad[1] else None)


def _colsum_any(g):
    """Column sums of a [B, C] gradient as a differentiable op (ColSum is linear: its own adjoint is a broadcast)."""
    return ColSum.apply(g)


def rowdot(h, w, bias=None):
    return RowDot.apply(h, w, bias)


class _PointPackCache(object):
    """MFMA-fragment image of PointNet.nn1's four weight matrices, rebuilt when one of them changed (see _PackCache)."""

    def __init__(self):
        self.entries = {}

    def get(self, weights):
        dev = weights[0].device
        ptrs = [p.data_ptr() for p in weights]
        key = (L.param_epoch_of_ptrs(ptrs),) + tuple(ptrs) + tuple(p._version for p in weights)
        entry = self.entries.get(dev)
        if entry is None or entry[0] != key or not L.writers_known(*weights) or _capturing(weights[0]):
            lib = _lib()
            packed = torch.empty(lib.sg_pointnet_packed_floats(), dtype=torch.float32, device=dev)
            arr = (ctypes.c_void_p * 4)(*[ptr(f32c(p.detach())) for p in weights])
            check(lib.sg_pointnet_pack(arr, ptr(packed), stream()), "pointnet_pack")
            entry = (key, packed)
            self.entries[dev] = entry
        return entry[1]


def pointnet_select(cache, x, weights, biases):
    """x [B,P,4] (P a multiple of 32) -> (max over the cloud of nn1(x) [B,512], the point that holds it [B,512] int32): one fused
    launch + the merge of its tiles; nothing is recorded (model/point_sdf_net.py:14-23,40)."""
    x = f32c(x.detach())
    B, P = x.shape[0], x.shape[1]
    lib = _lib()
    packed = cache.get(weights)
    out = torch.empty((B, 512), dtype=torch.float32, device=x.device)
    idx = torch.empty((B, 512), dtype=torch.int32, device=x.device)
    ws = workspace("pointnet_select", lib.sg_pointnet_select_workspace_bytes(B, P), x.device)
    barr = (ctypes.c_void_p * 4)(*[ptr(f32c(b.detach())) for b in biases])
    check(lib.sg_pointnet_select(ptr(x), ptr(packed), barr, B, P, ptr(out), ptr(idx), ptr(ws), ws.numel(), stream()), "pointnet_select")
    return out, idx


def sdfgen_fused(cache, pos, zb1, zb5, pps, eps, params):
    return SDFGenFused.apply(cache, pos, zb1, zb5, pps, eps, torch.is_grad_enabled(), *params)


# --------------------------------------------------------------------------------------------------------------
# reductions and latent-table rows (K9 / K10)
# --------------------------------------------------------------------------------------------------------------
class Mean(Function):
    """torch.mean over all elements as a two-stage deterministic HIP reduction (train_wgan.py:68,82)."""

    @staticmethod
    def forward(ctx, x):
        x = f32c(x)
        ctx.shape, ctx.n = x.shape, x.numel()
        out = torch.empty((), dtype=torch.float32, device=x.device)
        lib = _lib()
        ws = workspace("reduce", lib.sg_reduce_workspace_bytes(), x.device)
        check(lib.sg_reduce_sum(ptr(x), ptr(out), x.numel(), 1.0 / x.numel(), ptr(ws), ws.numel(), stream()), "reduce_sum")
        return out

    @staticmethod
    def backward(ctx, g):
        return (g / ctx.n).expand(ctx.shape)


def mean(x):
    return Mean.apply(x)


class MeanSplit(Function):
    """w_first * mean(x[:n_first]) + w_rest * mean(x[n_first:]) over the flattened x in one launch (and one for the backward):
    the WGAN losses on the critic's concatenated fake+real batch (train_wgan.py:68,82)."""

    @staticmethod
    def forward(ctx, x, n_first, w_first, w_rest):
        x = f32c(x)
        ctx.shape, ctx.args = x.shape, (x.numel(), int(n_first), float(w_first), float(w_rest))
        out = torch.empty((), dtype=torch.float32, device=x.device)
        # the gradient for an upstream 1 comes out of the same launch: the loss is the root of lib.backward() in every trainer
        ctx.dx_unit = torch.empty(x.shape, dtype=torch.float32, device=x.device) if ctx.needs_input_grad[0] else None
        check(_lib().sg_loss_mean_split_fwd(ptr(x), x.numel(), int(n_first), float(w_first), float(w_rest), ptr(out),
                                            ptr(ctx.dx_unit), stream()), "loss_mean_split_fwd")
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        n, n_first, w_first, w_rest = ctx.args
        if ctx.dx_unit is not None and L.is_unit_gradient(g):
            return ctx.dx_unit, None, None, None
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(_lib().sg_loss_mean_split_bwd(ptr(f32c(g)), ptr(dx), n, n_first, w_first, w_rest, stream()), "loss_mean_split_bwd")
        return dx, None, None, None


def mean_difference(x, n_first):
    """mean(x[:n_first]) - mean(x[n_first:])  (critic loss: scores of the fake half minus scores of the real half)."""
    return MeanSplit.apply(x, n_first, 1.0, -1.0)


def neg_mean(x):
    """-mean(x)  (generator loss)."""
    return MeanSplit.apply(x, x.numel(), -1.0, 0.0)


class GatherRows(Function):
    """table[idx] for a 2-D table (latent_codes[model_indices], train_sdf_autodecoder.py:80); backward is the
    index_add scatter into a zero table-shaped gradient."""

    @staticmethod
    def forward(ctx, table, idx):
        table = f32c(table)
        n, width = idx.numel(), table.shape[1]
        out = torch.empty((n, width), dtype=torch.float32, device=table.device)
        check(_lib().sg_gather_rows(ptr(table), ptr(idx), ptr(out), n, width, stream()), "gather_rows")
        ctx.rows = table.shape[0]
        ctx.save_for_backward(idx, table)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        idx, table = ctx.saved_tensors
        width = g.shape[1]
        if g.stride(1) != 1:
            g = g.contiguous()
        tg = L.grad_destination(table, (ctx.rows, width))     # the latent table's own slice of its optimizer's flat buffer
        if tg is not None:
            tg.zero_()
        else:
            tg = torch.zeros((ctx.rows, width), dtype=torch.float32, device=g.device)
        check(_lib().sg_scatter_add_rows(g.data_ptr(), g.stride(0), ptr(idx), ptr(tg), idx.numel(), width, stream()),
              "scatter_add_rows")
        return tg, None


def gather_rows(table, idx, out=None):
    """table[idx].  out: where to write the rows (no-grad data movement only — e.g. a loader delivering a batch into a trainer's
    slot): a contiguous fp32 tensor of idx.numel() * table.shape[1] elements on the table's device."""
    if out is None:
        return GatherRows.apply(table, idx)
    table = f32c(table)
    n, width = idx.numel(), table.shape[1]
    if out.dtype != torch.float32 or not out.is_contiguous() or out.numel() != n * width or out.device != table.device:
        raise RuntimeError("gather_rows: `out` must be a contiguous fp32 tensor of %d x %d elements on %s" % (n, width, table.device))
    if idx.dtype != torch.int64 or not idx.is_contiguous():
        idx = idx.to(torch.int64).contiguous()
    check(_lib().sg_gather_rows(ptr(table), ptr(idx), ptr(out), n, width, stream()), "gather_rows")
    return out


class BadIndexWords(object):
    """The out-of-range-batch-index state of ONE consumer of sdf_batch_sort (a trainer): a host word in PINNED HOST memory
    (device-accessible under unified addressing: the sort kernel touches it only on error, the host reads it with a plain load — no
    copy, no launch, no synchronisation per step, nothing to capture), a device word that guarded optimizer kernels read
    (`optim.Adam.guard`), and the number of the latest sort call that used them.  Both words are sticky and set together; they
    carry the number of the FIRST bad sort call.  Every trainer owns a pair (with one pair per device a second trainer —
    or a step that does not go through the sort — had its updates dropped by the other one's bad batch without being counted, and
    whoever polled first cleared the other's error); `default_bad_index_words(dev)` serves direct callers of ops.sdf_batch_sort."""

    def __init__(self, dev):
        dev = torch.device(dev)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev = dev
        self.flag = torch.zeros(1, dtype=torch.int32)
        if dev.type == "cuda":
            self.flag = self.flag.pin_memory()
        self.guard = torch.zeros(1, dtype=torch.int32, device=dev)
        self.seq = 0
        _all_bad_index_words.add(self)

    def next_seq(self):
        self.seq = self.seq % 0x7fffffff + 1        # this call's number (never 0)
        return self.seq

    def pending(self):
        return int(self.flag[0]) != 0

    def raise_if_bad(self, synchronise_first=False, synchronise_before_raise=False):
        capturing = self.dev.type == "cuda" and torch.cuda.is_current_stream_capturing()
        if synchronise_first and self.dev.type == "cuda" and not capturing:
            torch.cuda.synchronize(self.dev)
        if not self.pending():
            return
        if synchronise_before_raise and self.dev.type == "cuda" and not capturing:
            torch.cuda.synchronize(self.dev)
        err = IndexError("sdf_batch_sort: a batch index was outside [0, shapes * pointcloud_size)")
        err.sort_sequence = int(self.flag[0])
        self.flag[0] = 0
        if not capturing:
            self.guard.zero_()
        raise err


import weakref as _weakref  # noqa: E402
_all_bad_index_words = _weakref.WeakSet()
_bad_index_flags = {}          # device -> the default pair of that device


def default_bad_index_words(dev):
    dev = torch.device(dev)
    if dev.type == "cuda" and dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    words = _bad_index_flags.get(dev)
    if words is None:
        words = _bad_index_flags[dev] = BadIndexWords(dev)
    return words


def sdf_batch_sort(indices, pointcloud_size, shapes, points, sdf, words=None):
    """The auto-decoder batch grouped by shape in one stable counting sort (train_sdf_autodecoder.py:78-85): returns
    points[indices] ([N,3]), sdf[indices] ([N]), the shape id of every entry (int32 [N]), the run bounds of every shape
    (int64 [shapes+1]) and the run lengths (float32 [shapes]) — all in shape order; no host synchronisation.  An index
    outside the tables sets `words` (a BadIndexWords; default: the device's shared pair) and raises at the owner's next check
    (`words.raise_if_bad()` / `check_batch_indices()`; the reference raises an IndexError at once)."""
    lib = _lib()
    if shapes > sdf_batch_sort_max_shapes():
        raise RuntimeError("sdf_batch_sort: %d shapes (at most %d)" % (shapes, sdf_batch_sort_max_shapes()))
    if indices.dtype != torch.int64 or not indices.is_contiguous():
        indices = indices.to(torch.int64).contiguous()
    points, sdf = f32c(points), f32c(sdf)
    if points.shape[0] != sdf.shape[0] or points.shape[0] < shapes * pointcloud_size:
        raise RuntimeError("sdf_batch_sort: tables hold %d / %d rows, need %d" % (points.shape[0], sdf.shape[0],
                                                                                  shapes * pointcloud_size))
    dev, n = points.device, indices.numel()
    out_points = torch.empty((n, 3), dtype=torch.float32, device=dev)
    out_sdf = torch.empty(n, dtype=torch.float32, device=dev)
    out_shape = torch.empty(n, dtype=torch.int32, device=dev)
    seg_off = torch.empty(shapes + 1, dtype=torch.int64, device=dev)
    counts = torch.empty(shapes, dtype=torch.float32, device=dev)
    words = default_bad_index_words(dev) if words is None else words
    seq = words.next_seq()
    ws = workspace("sdf_batch_sort", lib.sg_sdf_batch_sort_workspace_bytes(n, shapes), dev)
    check(lib.sg_sdf_batch_sort(ptr(indices), n, pointcloud_size, shapes, ptr(points), ptr(sdf), ptr(out_points), ptr(out_sdf),
                                ptr(out_shape), ptr(seg_off), ptr(counts), words.flag.data_ptr(), ptr(words.guard), seq, ptr(ws),
                                ws.numel(), stream()), "sdf_batch_sort")
    return out_points, out_sdf, out_shape, seg_off, counts


def batch_sort_sequence(dev):
    """The number of the latest sdf_batch_sort call on `dev`'s default words (1, 2, …; 0 before the first).  The IndexError of a bad
    batch carries the number of the FIRST call that saw one (`.sort_sequence`)."""
    return default_bad_index_words(dev).seq


def batch_index_guard(dev):
    """The device word of `dev`'s default pair: non-zero from the moment a sort kernel met an out-of-range batch index until the
    host has raised for it.  optim.Adam(…).guard = such a tensor makes the update of that batch a no-op (sg_adam_step_guarded): the
    reference raises its IndexError before any update (train_sdf_autodecoder.py:79), here the error surfaces one step late but the
    state it leaves behind is the one before the bad batch.  (Trainers own a pair each: BadIndexWords.)"""
    return default_bad_index_words(dev).guard


def sdf_batch_sort_max_shapes():
    return L.load().sg_sdf_batch_sort_max_shapes()


def check_batch_indices():
    """Synchronises (once) and raises if any sdf_batch_sort call on a device's DEFAULT pair of words since the last check saw an index
    outside its tables.  Pairs a trainer owns (BadIndexWords passed as `words=`) are the trainer's to poll: its handler also takes the
    host-side Adam counters of the skipped updates back, which a raise from here would bypass."""
    if torch.cuda.is_available() and any(d.type == "cuda" for d in _bad_index_flags):
        torch.cuda.synchronize()
    for words in list(_bad_index_flags.values()):
        words.raise_if_bad()


def poll_batch_indices():
    """The same check without a synchronisation, for use on EVERY step: a plain read of the pinned host words the sort kernels set.
    An out-of-range index is reported as soon as the kernel that saw it has run — in practice at the next step (the reference
    raises at once; the batch in question was computed on clamped rows, and an optimizer guarded by the pair's device word did not
    apply it).  Costs nothing on the device and is safe inside a stream capture (there is nothing to record)."""
    for words in list(_bad_index_flags.values()):       # (the default pairs only: see check_batch_indices)
        words.raise_if_bad(synchronise_before_raise=True)


# --------------------------------------------------------------------------------------------------------------
# PointNet-discriminator GAN family (model/point_sdf_net.py)
# --------------------------------------------------------------------------------------------------------------
class LayerNormAct(Function):
    """y = act(layer_norm(x + rowbias[row // rows_per_shape])) for x [R,C] (point_sdf_net.py:104-116: `lin(x)`, the
    z-injection `z_lin(z).unsqueeze(1) + x`, `norm`, `relu` in one pass).  With `tail` [R,T] the output is [R,C+T]
    whose last T columns are a copy of tail: the skip concat `cat([x, pos])` (:100) without a second pass over x."""

    @staticmethod
    def forward(ctx, x, rowbias, rows_per_shape, gamma, beta, eps, act, tail):
        x = f32c(x)
        R, C = x.shape
        T = 0 if tail is None else tail.shape[1]
        y = torch.empty((R, C + T), dtype=torch.float32, device=x.device)
        mean = torch.empty(R, dtype=torch.float32, device=x.device)
        rstd = torch.empty(R, dtype=torch.float32, device=x.device)
        rb = None if rowbias is None else f32c(rowbias)
        check(_lib().sg_layernorm_fwd(ptr(x), C, ptr(rb), rows_per_shape, ptr(gamma), ptr(beta), ptr(y), C + T, ptr(mean),
                                      ptr(rstd), R, C, eps, act, stream()), "layernorm_fwd")
        if T:
            y[:, C:] = tail
        ctx.cfg = (R, C, T, rows_per_shape, act)
        ctx.save_for_backward(x, rb, gamma, y, mean, rstd)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, rb, gamma, y, mean, rstd = ctx.saved_tensors
        R, C, T, rps, act = ctx.cfg
        gy = f32c(gy)
        lib = _lib()
        dz = torch.empty((R, C), dtype=torch.float32, device=x.device)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = workspace("layernorm", lib.sg_layernorm_bwd_workspace_bytes(R, C), x.device)
        check(lib.sg_layernorm_bwd(ptr(x), C, ptr(rb), rps, ptr(gamma), ptr(y), C + T, ptr(gy), C + T, ptr(mean), ptr(rstd),
                                   ptr(dz), C, ptr(dgamma), ptr(dbeta), R, C, act, ptr(ws), ws.numel(), stream()),
              "layernorm_bwd")
        grb = None
        if rb is not None and ctx.needs_input_grad[1]:
            grb = colsum_tall_raw(dz, R // rps, rps * C, rps, C, C)
        gtail = gy[:, C:] if (T and ctx.needs_input_grad[7]) else None
        return dz, grb, None, dgamma, dbeta, None, None, gtail


class SegMaxScatter(Function):
    """dx[b,p,c] = dy[b,c] where p == idx[b,c], else 0 (the adjoint of the max over points); its own backward is the
    gather at idx, so the gradient penalty's double backward (train_point_gan.py:61-70) stays on HIP kernels."""

    @staticmethod
    def forward(ctx, dy, idx, P):
        dy = f32c(dy)
        B, C = dy.shape
        dx = torch.empty((B, P, C), dtype=torch.float32, device=dy.device)
        check(_lib().sg_segmax_scatter(ptr(dy), ptr(idx), ptr(dx), B, P, C, stream()), "segmax_scatter")
        ctx.save_for_backward(idx)
        return dx

    @staticmethod
    def backward(ctx, gdx):
        (idx,) = ctx.saved_tensors
        return SegMaxGather.apply(gdx, idx), None, None


class SegMaxGather(Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = f32c(x)
        B, P, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        check(_lib().sg_segmax_gather(ptr(x), ptr(idx), ptr(out), B, P, C, stream()), "segmax_gather")
        ctx.P = P
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return SegMaxScatter.apply(g, idx, ctx.P), None


class SegMax(Function):
    """x [B,P,C] -> max over the P points of each shape (`x.max(dim=-2)[0]`, point_sdf_net.py:40)."""

    @staticmethod
    def forward(ctx, x):
        x = f32c(x)
        B, P, C = x.shape
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        idx = torch.empty((B, C), dtype=torch.int32, device=x.device)
        lib = _lib()
        nb = lib.sg_segmax_workspace_bytes(B, P, C)
        ws = workspace("segmax", nb, x.device) if nb else None
        check(lib.sg_segmax_fwd(ptr(x), ptr(out), ptr(idx), B, P, C, ptr(ws), ws.numel() if ws is not None else 0, stream()),
              "segmax_fwd")
        ctx.P = P
        ctx.save_for_backward(idx)
        ctx.mark_non_differentiable(idx)
        return out, idx

    @staticmethod
    def backward(ctx, g, _gidx):
        (idx,) = ctx.saved_tensors
        return SegMaxScatter.apply(g, idx, ctx.P)


def segmax(x):
    return SegMax.apply(x)[0]


def layernorm_act(x, rowbias, rows_per_shape, gamma, beta, eps=1e-5, act=ACT_NONE, tail=None):
    return LayerNormAct.apply(x, rowbias, rows_per_shape, gamma, beta, eps, act, tail)


def voxel_prepare(x, clamp, divisor, out=None):
    """out = clamp(x, -clamp, clamp) / divisor on the device (VoxelDataset.__getitem__, datasets.py:19-22); divisor <= 0
    skips the division.  In place when out is None."""
    x = f32c(x)
    out = x if out is None else out
    check(_lib().sg_voxel_prepare(ptr(x), ptr(out), x.numel(), float(clamp), float(divisor), stream()), "voxel_prepare")
    return out


# --------------------------------------------------------------------------------------------------------------
# loss compositions, gradient-penalty pieces, fade-in blend (K8 / K9)
# --------------------------------------------------------------------------------------------------------------
def _loss_ws(device):
    return workspace("loss", _lib().sg_loss_workspace_bytes(), device)


class WeightedL1(Function):
    """mean(|d|), d = out - target, d *= neg_weight where target < 0 (get_reconstruction_loss, train_autoencoder.py:57-62;
    neg_weight 1: the DeepSDF data term mean|out - sdf|, train_sdf_autodecoder.py:88)."""

    @staticmethod
    def forward(ctx, out, target, neg_weight):
        out, target = f32c(out), f32c(target.detach())
        if out.shape != target.shape:
            raise RuntimeError("weighted_l1: shape mismatch %s vs %s" % (tuple(out.shape), tuple(target.shape)))
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        ws = _loss_ws(out.device)
        check(_lib().sg_loss_weighted_l1_fwd(ptr(out), ptr(target), out.numel(), neg_weight, ptr(loss), ptr(ws), ws.numel(),
                                             stream()), "loss_weighted_l1_fwd")
        ctx.neg_weight = neg_weight
        ctx.save_for_backward(out, target)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        out, target = ctx.saved_tensors
        d = torch.empty_like(out)
        check(_lib().sg_loss_weighted_l1_bwd(ptr(out), ptr(target), ptr(f32c(g)), ptr(d), out.numel(), ctx.neg_weight,
                                             stream()), "loss_weighted_l1_bwd")
        return d, None, None


def weighted_l1(out, target, neg_weight=1.0):
    return WeightedL1.apply(out, target, float(neg_weight))


def count_sign_mismatch(a, b):
    """#{(a * b) < 0} as an int64 device scalar (voxel_difference, train_autoencoder.py:50-52): integer work, bit-exact."""
    a, b = f32c(a.detach()), f32c(b.detach())
    if a.numel() != b.numel():
        raise RuntimeError("count_sign_mismatch: %d vs %d elements" % (a.numel(), b.numel()))
    count = torch.empty((), dtype=torch.int64, device=a.device)
    ws = _loss_ws(a.device)
    check(_lib().sg_count_sign_mismatch(ptr(a), ptr(b), a.numel(), ptr(count), ptr(ws), ws.numel(), stream()),
          "count_sign_mismatch")
    return count


class KLD(Function):
    """-0.5 * sum(1 + lv - mu^2 - exp(lv)) / numel (kld_loss, train_autoencoder.py:54-55)."""

    @staticmethod
    def forward(ctx, mean, log_variance):
        mean, log_variance = f32c(mean), f32c(log_variance)
        loss = torch.empty((), dtype=torch.float32, device=mean.device)
        ws = _loss_ws(mean.device)
        check(_lib().sg_loss_kld_fwd(ptr(mean), ptr(log_variance), mean.numel(), ptr(loss), ptr(ws), ws.numel(), stream()),
              "loss_kld_fwd")
        ctx.save_for_backward(mean, log_variance)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mean, lv = ctx.saved_tensors
        dm, dl = torch.empty_like(mean), torch.empty_like(lv)
        check(_lib().sg_loss_kld_bwd(ptr(mean), ptr(lv), ptr(f32c(g)), ptr(dm), ptr(dl), mean.numel(), stream()),
              "loss_kld_bwd")
        return dm, dl


def kld(mean, log_variance):
    return KLD.apply(mean, log_variance)


class BCEConst(Function):
    """torch.nn.functional.binary_cross_entropy(p, full_like(p, target)) on the [B] vector of discriminator outputs
    (train_gan.py:30,78,84), logarithms clamped at -100 and the backward exactly as torch computes them."""

    @staticmethod
    def forward(ctx, p, target):
        p = f32c(p)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        check(_lib().sg_loss_bce_fwd(ptr(p), p.numel(), float(target), ptr(loss), stream()), "loss_bce_fwd")
        ctx.target = float(target)
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        dp = torch.empty_like(p)
        check(_lib().sg_loss_bce_bwd(ptr(p), ptr(f32c(g)), ptr(dp), p.numel(), ctx.target, stream()), "loss_bce_bwd")
        return dp, None


def bce_const(p, target):
    return BCEConst.apply(p, target)


class NegMeanLog(Function):
    """-torch.mean(torch.log(p))  (the classic GAN's generator loss, train_gan.py:65)."""

    @staticmethod
    def forward(ctx, p):
        p = f32c(p)
        loss = torch.empty((), dtype=torch.float32, device=p.device)
        check(_lib().sg_loss_neg_mean_log_fwd(ptr(p), p.numel(), ptr(loss), stream()), "loss_neg_mean_log_fwd")
        ctx.save_for_backward(p)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        dp = torch.empty_like(p)
        check(_lib().sg_loss_neg_mean_log_bwd(ptr(p), ptr(f32c(g)), ptr(dp), p.numel(), stream()), "loss_neg_mean_log_bwd")
        return dp


def neg_mean_log(p):
    return NegMeanLog.apply(p)


class VAEReparam(Function):
    """z = mean + exp(0.5 * log_variance) * eps  (model/autoencoder.py:77-82; eps is drawn by the caller)."""

    @staticmethod
    def forward(ctx, mean, log_variance, eps):
        mean, log_variance, eps = f32c(mean), f32c(log_variance), f32c(eps)
        z = torch.empty_like(mean)
        check(_lib().sg_vae_reparam_fwd(ptr(mean), ptr(log_variance), ptr(eps), ptr(z), mean.numel(), stream()), "vae_reparam_fwd")
        ctx.save_for_backward(log_variance, eps)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, gz):
        lv, eps = ctx.saved_tensors
        gz = f32c(gz)
        dl = None
        if ctx.needs_input_grad[1]:
            dl = torch.empty_like(lv)
            check(_lib().sg_vae_reparam_bwd(ptr(lv), ptr(eps), ptr(gz), ptr(dl), lv.numel(), stream()), "vae_reparam_bwd")
        return (gz if ctx.needs_input_grad[0] else None), dl, None


def vae_reparam(mean, log_variance, eps):
    return VAEReparam.apply(mean, log_variance, eps)


class MeanSq(Function):
    """sum_r w_r |x_r|^2 / denom for x [rows, L] (w = 1 without row_weight): the latent regulariser
    mean(batch_latent_codes^2) of train_sdf_autodecoder.py:88 — over the gathered rows (row_weight None, denom = numel) or
    over the latent table with row_weight = how often each shape occurs in the batch (denom = batch * L)."""

    @staticmethod
    def forward(ctx, x, row_weight, denom):
        x = f32c(x)
        rows, width = x.shape
        rw = None if row_weight is None else f32c(row_weight)
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        ws = _loss_ws(x.device)
        check(_lib().sg_loss_meansq_fwd(ptr(x), ptr(rw), rows, width, float(denom), ptr(loss), ptr(ws), ws.numel(),
                                        stream()), "loss_meansq_fwd")
        ctx.denom = float(denom)
        ctx.save_for_backward(x, rw)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, rw = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(_lib().sg_loss_meansq_bwd(ptr(x), ptr(rw), ptr(f32c(g)), ptr(dx), x.shape[0], x.shape[1], ctx.denom,
                                        stream()), "loss_meansq_bwd")
        return dx, None, None


def mean_sq(x, row_weight=None, denom=None):
    x2 = x.reshape(x.shape[0], -1) if x.dim() != 2 else x
    return MeanSq.apply(x2, row_weight, float(x.numel() if denom is None else denom))


class DeepSDFLoss(Function):
    """mean|out - sdf| + sum_r w_r |z_r|^2 / denom, the loss of train_sdf_autodecoder.py:88 as one op: one pass + the finishing
    wave forward, one launch backward (weighted_l1 + mean_sq + their fp32 add, bit for bit, in 3 launches instead of 7)."""

    @staticmethod
    def forward(ctx, out, target, z, row_weight, denom):
        out, target, z = f32c(out), f32c(target.detach()), f32c(z)
        if out.shape != target.shape:
            raise RuntimeError("deepsdf_loss: shape mismatch %s vs %s" % (tuple(out.shape), tuple(target.shape)))
        rows, width = z.shape
        rw = None if row_weight is None else f32c(row_weight)
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        ws = _loss_ws(out.device)
        # the gradient for an upstream 1 comes out of the same launch (the loss is the root of lib.backward() in the trainer)
        ctx.d_unit = torch.empty_like(out) if ctx.needs_input_grad[0] else None
        ctx.dz_unit = torch.empty_like(z) if ctx.needs_input_grad[2] else None
        check(_lib().sg_loss_deepsdf_fused(ptr(out), ptr(target), out.numel(), ptr(z), ptr(rw), rows, width, float(denom),
                                           ptr(loss), ptr(ctx.d_unit), ptr(ctx.dz_unit), ptr(ws), ws.numel(),
                                           ptr(L.tickets("deepsdf", out.device)), stream()), "loss_deepsdf_fused")
        ctx.denom = float(denom)
        ctx.save_for_backward(out, target, z, rw)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        out, target, z, rw = ctx.saved_tensors
        if L.is_unit_gradient(g) and (ctx.d_unit is not None or ctx.dz_unit is not None):
            return ctx.d_unit, None, ctx.dz_unit, None, None
        d, dz = torch.empty_like(out), torch.empty_like(z)
        check(_lib().sg_loss_deepsdf_bwd(ptr(out), ptr(target), out.numel(), ptr(z), ptr(rw), z.shape[0], z.shape[1],
                                         ctx.denom, ptr(f32c(g)), ptr(d), ptr(dz), stream()), "loss_deepsdf_bwd")
        return d, None, dz, None, None


def deepsdf_loss(out, target, z, row_weight=None, denom=None):
    z2 = z.reshape(z.shape[0], -1) if z.dim() != 2 else z
    return DeepSDFLoss.apply(out, target, z2, row_weight, float(z.numel() if denom is None else denom))


class GradientPenalty(Function):
    """((||g_b||_2 - 1)^2).mean() * weight over the per-sample rows of `gradients`
    (train_hybrid_progressive_gan.py:110-111, train_point_gan.py:68-70)."""

    @staticmethod
    def forward(ctx, gradients, weight):
        B = gradients.shape[0]
        g2 = f32c(gradients).reshape(B, -1)
        norms = torch.empty(B, dtype=torch.float32, device=g2.device)
        loss = torch.empty((), dtype=torch.float32, device=g2.device)
        check(_lib().sg_gradient_penalty_fwd(ptr(g2), B, g2.shape[1], weight, ptr(norms), ptr(loss), stream()),
              "gradient_penalty_fwd")
        ctx.weight, ctx.shape = weight, gradients.shape
        ctx.save_for_backward(g2, norms)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g2, norms = ctx.saved_tensors
        dg = torch.empty_like(g2)
        check(_lib().sg_gradient_penalty_bwd(ptr(g2), ptr(norms), ptr(f32c(g)), ptr(dg), g2.shape[0], g2.shape[1],
                                             ctx.weight, stream()), "gradient_penalty_bwd")
        return dg.reshape(ctx.shape), None


def gradient_penalty(gradients, weight):
    return GradientPenalty.apply(gradients, float(weight))


def lerp_rows(a, b, alpha):
    """alpha[b] * a[b] + (1 - alpha[b]) * b[b] per sample (train_hybrid_progressive_gan.py:103-105); no autograd: the
    reference detaches both operands and makes the result a leaf."""
    B = a.shape[0]
    a2, b2 = f32c(a.detach()).reshape(B, -1), f32c(b.detach()).reshape(B, -1)
    al = f32c(alpha.detach()).reshape(-1)
    if al.numel() != B or a2.shape != b2.shape:
        raise RuntimeError("lerp_rows: need one alpha per sample and equal shapes")
    out = torch.empty_like(a2)
    check(_lib().sg_lerp_rows(ptr(a2), ptr(b2), ptr(al), ptr(out), B, a2.shape[1], stream()), "lerp_rows")
    return out.reshape(a.shape)


class Scale(Function):
    @staticmethod
    def forward(ctx, x, a):
        x = f32c(x)
        ctx.a = a
        out = torch.empty_like(x)
        check(_lib().sg_axpby(ptr(x), None, ptr(out), x.numel(), a, 0.0, stream()), "axpby")
        return out

    @staticmethod
    def backward(ctx, g):
        return Scale.apply(g, ctx.a), None


class Chan0(Function):
    """out[b, s] = a * g[b, 0, s] for g [B, C, S...]; adjoint of the channel-0 embedding."""

    @staticmethod
    def forward(ctx, g, a):
        g = f32c(g)
        B, C = g.shape[0], g.shape[1]
        S = g.numel() // (B * C)
        out = torch.empty((B,) + tuple(g.shape[2:]), dtype=torch.float32, device=g.device)
        check(_lib().sg_channel0(ptr(g), ptr(out), B, C, S, a, stream()), "channel0")
        ctx.a, ctx.C = a, C
        return out

    @staticmethod
    def backward(ctx, gg):
        return FadeBlend.apply(None, gg, ctx.C, 0.0, ctx.a), None


class FadeBlend(Function):
    """fade * x + half_scale * from_SDF(half) (model/progressive_gan.py:48-50) with x [B,C,r,r,r], half [B,r,r,r]: the
    C-1 zero channels of from_SDF are never built.  x None: the embedding alone (the adjoint of Chan0).  Linear in both
    operands, backward written with Scale / Chan0, so the gradient penalty's double backward passes through."""

    @staticmethod
    def forward(ctx, x, half, C, fade, half_scale):
        half = f32c(half)
        B = half.shape[0]
        S = half.numel() // B
        if x is not None:
            x = f32c(x)
            if x.shape[0] != B or x.shape[1] != C or x.numel() != B * C * S:
                raise RuntimeError("fade_blend: x %s does not match half %s" % (tuple(x.shape), tuple(half.shape)))
        out = torch.empty((B, C) + tuple(half.shape[1:]), dtype=torch.float32, device=half.device)
        check(_lib().sg_fade_blend(ptr(x), ptr(half), ptr(out), B, C, S, fade, half_scale, stream()), "fade_blend")
        ctx.cfg = (x is not None, fade, half_scale)
        return out

    @staticmethod
    def backward(ctx, g):
        has_x, fade, hs = ctx.cfg
        gx = Scale.apply(g, fade) if (has_x and ctx.needs_input_grad[0]) else None
        gh = Chan0.apply(g, hs) if ctx.needs_input_grad[1] else None
        return gx, gh, None, None, None


class Subsample2(Function):
    """x[:, ::2, ::2, ::2] of [B,R,R,R] grids (model/progressive_gan.py:49; index work, bit-exact)."""

    @staticmethod
    def forward(ctx, x):
        x = f32c(x)
        B, R = x.shape[0], x.shape[-1]
        out = torch.empty((B, R // 2, R // 2, R // 2), dtype=torch.float32, device=x.device)
        check(_lib().sg_subsample2(ptr(x), ptr(out), B, R, stream()), "subsample2")
        return out

    @staticmethod
    def backward(ctx, g):
        return Subsample2Adjoint.apply(g)


class Subsample2Adjoint(Function):
    @staticmethod
    def forward(ctx, g):
        g = f32c(g)
        B, h = g.shape[0], g.shape[-1]
        out = torch.empty((B, 2 * h, 2 * h, 2 * h), dtype=torch.float32, device=g.device)
        check(_lib().sg_subsample2_adjoint(ptr(g), ptr(out), B, 2 * h, stream()), "subsample2_adjoint")
        return out

    @staticmethod
    def backward(ctx, gg):
        return Subsample2.apply(gg)


def fade_blend(x, x_in, fade):
    """The progressive discriminator's fade-in (model/progressive_gan.py:48-50): x [B,C,r,r,r] from the new stage, x_in
    [B,2r,2r,2r] the stage's input grid."""
    half = Subsample2.apply(x_in)
    return FadeBlend.apply(x, half, x.shape[1], float(fade), float(1.0 - fade))


class ScatterMaxScatter(Function):
    @staticmethod
    def forward(ctx, dy, arg, N):
        dy = f32c(dy)
        B, C = dy.shape
        dx = torch.empty((N, C), dtype=torch.float32, device=dy.device)
        check(_lib().sg_scatter_max_scatter(ptr(dy), ptr(arg), ptr(dx), N, B, C, stream()), "scatter_max_scatter")
        ctx.save_for_backward(arg)
        return dx

    @staticmethod
    def backward(ctx, gdx):
        (arg,) = ctx.saved_tensors
        return ScatterMaxGather.apply(gdx, arg), None, None


class ScatterMaxGather(Function):
    @staticmethod
    def forward(ctx, x, arg):
        x = f32c(x)
        N, C = x.shape
        B = arg.shape[0]
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        check(_lib().sg_scatter_max_gather(ptr(x), ptr(arg), ptr(out), N, B, C, stream()), "scatter_max_gather")
        ctx.N = N
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, g):
        (arg,) = ctx.saved_tensors
        return ScatterMaxScatter.apply(g, arg, ctx.N), None


class ScatterMax(Function):
    """torch_scatter.scatter_max(x, batch, dim=-2)[0] for x [N,C] and a ragged int64 `batch` vector
    (model/point_sdf_net.py:42-43): out [B,C], B = dim_size; empty segments give 0."""

    @staticmethod
    def forward(ctx, x, batch, B):
        x = f32c(x)
        N, C = x.shape
        lib = _lib()
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
        ws = workspace("scatter_max", lib.sg_scatter_max_workspace_bytes(B, C), x.device)
        check(lib.sg_scatter_max_fwd(ptr(x), ptr(batch), ptr(out), ptr(arg), N, B, C, ptr(ws), ws.numel(), stream()),
              "scatter_max_fwd")
        ctx.N = N
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, g, _garg):
        (arg,) = ctx.saved_tensors
        return ScatterMaxScatter.apply(g, arg, ctx.N), None, None


def scatter_max(x, batch, dim_size=None):
    if batch.dtype != torch.int64:
        batch = batch.long()
    if dim_size is None:
        dim_size = int(batch.max().item()) + 1
    return ScatterMax.apply(x, batch.contiguous(), int(dim_size))[0]
