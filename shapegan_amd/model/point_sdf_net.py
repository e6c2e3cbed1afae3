"""PointNet-discriminator GAN family (model/point_sdf_net.py): `PointNet` critic and `SDFGenerator` on HIP kernels.

Same constructor arguments and state_dict keys as the reference (`nn1.{0,2,4,6}`, `nn2.{0,2,4}`; `lins.i`, `norms.i`,
`z_lin1`, `z_lin2`, including the never-used last LayerNorm(1)), so checkpoints load either way.  The torch layer
objects are parameter containers; every forward below runs through shapegan_amd.ops:
  * per-point Linear (+ReLU): MFMA GEMM with bias / activation epilogue (`ops.linear`), closed under double backward;
  * `lin(x) + z_lin(z).unsqueeze(1)`, LayerNorm, ReLU (:104-116): one `sg_layernorm_fwd` pass, the per-shape z row is
    added on load — the [B,P,256] broadcast sum never exists; the skip concat `cat([x, pos])` (:100) is written by the
    same pass as the tail of a 259-float row;
  * `x.max(dim=-2)[0]` (:40): `sg_segmax_fwd` (+ scatter / gather adjoints for backward and double backward).
The training configuration of train_point_gan.py:21 (hidden_channels 256, num_layers 8, norm) has the layer shapes of an SDFNet
without latent columns: on the GPU its whole forward is ONE launch of the fused MLP kernel in its LayerNorm form (`ops.sdfgen_fused`,
csrc/sdfnet.hip), its backward the fused backward-data kernel + one finishing launch + one weight-gradient GEMM batch.
"""
import torch
import torch.nn as nn

from .. import ops
from ..lib import ACT_NONE, ACT_RELU


def _mlp(sizes):
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(sizes[i], sizes[i + 1]))
        if i + 2 < len(sizes):
            mods.append(nn.ReLU())
    return nn.Sequential(*mods)


def _run_mlp(seq, x):
    """Linear / ReLU chain of an nn.Sequential container on the GEMM kernel (activation fused into the epilogue)."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
        x = ops.linear(x, mods[i].weight, mods[i].bias, ACT_RELU if relu else ACT_NONE)
        i += 2 if relu else 1
    return x


class PointNet(nn.Module):
    """point_sdf_net.py:11-47: per-point MLP 4 -> 64 -> 128 -> 256 -> 512, max over the points of a shape, MLP
    512 -> 256 -> 128 -> out_channels.

    The adjoint of the max is sparse: of a shape's P points only the (at most 512) that hold a channel's maximum pass a
    gradient — to the parameters of nn1, to the input (the gradient penalty of train_point_gan.py:61-70 differentiates with
    respect to the distance channel) and in the double backward.  For large clouds the forward therefore runs nn1 twice: once
    over all points WITHOUT recording anything (only the argmax indices are kept: no activation is saved, no dense backward
    ever runs), and once, recorded, over the 512 selected points of every shape, whose c-th row's c-th output IS the maximum of
    channel c.  Same function, same (sub)gradient as `x.max(dim=-2)[0]` (autograd routes the gradient to the argmax row
    as well), closed under double backward because it is made of the same Linear / ReLU operators on a gathered batch; the cost
    of everything behind the plain forward drops by P / 512."""

    SPARSE_MIN_POINTS = 1024       # below: the selected rows would be more than half of the cloud

    def __init__(self, out_channels):
        super(PointNet, self).__init__()
        self.nn1 = _mlp([4, 64, 128, 256, 512])
        self.nn2 = _mlp([512, 256, 128, out_channels])
        self._pack = ops._PointPackCache()

    def selected_points(self, x):
        """x [B,P,4] -> [B,512] int64: for every channel of nn1 the point of the shape that holds its maximum (nothing recorded).
        Clouds of a multiple of 32 points: one fused launch that never writes the per-point layers (ops.pointnet_select)."""
        B, P = x.shape[0], x.shape[1]
        if P % 32 == 0:
            lins = [m for m in self.nn1 if isinstance(m, nn.Linear)]
            return ops.pointnet_select(self._pack, x, [l.weight for l in lins], [l.bias for l in lins])[1].long()
        with torch.no_grad():
            h = _run_mlp(self.nn1, x.reshape(-1, 4))
            return ops.SegMax.apply(h.reshape(B, P, h.shape[-1]))[1].long()

    def forward_selected(self, xs):
        """xs [B,512,4], row c = the point that holds the maximum of channel c: nn2(max over the cloud of nn1), recorded."""
        B, C = xs.shape[0], xs.shape[1]
        mods = list(self.nn1)
        h3 = _run_mlp(mods[:-1], xs.reshape(-1, 4))              # [B*512, 256], through the ReLU in front of the last Linear
        # the last Linear's output c of row c only: a row-wise dot product with the row's own weight row (ops.RowDot) instead of
        # the [B*512, 512] product whose diagonal it is
        m = ops.rowdot(h3.reshape(B, C, h3.shape[-1]), mods[-1].weight, mods[-1].bias)
        return _run_mlp(self.nn2, m)

    @staticmethod
    def gather_points(x, idx):
        """Rows idx [B,C] of x [B,P,K] -> [B,C,K] (differentiable: the backward is a deterministic scatter-add of the few selected
        rows — several channels may have selected the same point —, closed under double backward: ops.GatherRowsGrouped)."""
        B, P, C = x.shape[0], x.shape[1], idx.shape[1]
        rows = (idx + torch.arange(B, device=x.device).unsqueeze(1) * P).reshape(-1)
        return ops.gather_rows_grouped(x.reshape(B * P, -1), rows, C).reshape(B, C, -1)

    def forward(self, pos, dist, batch=None):
        dist = dist.unsqueeze(-1) if dist.size(-1) != 1 else dist
        x = torch.cat([pos, dist], dim=-1)                      # [B,P,4] (16 B per point)
        if batch is not None:
            # ragged point sets (train_point_gan_ref.py:31-52): x [N,4], batch [N] names each point's shape
            h = _run_mlp(self.nn1, x.reshape(-1, 4))
            h = ops.scatter_max(h, batch.reshape(-1))           # [B,512], torch_scatter.scatter_max(...)[0]  (:42)
            return _run_mlp(self.nn2, h)
        lead, P = x.shape[:-2], x.shape[-2]
        if P >= self.SPARSE_MIN_POINTS and torch.is_grad_enabled() and (
                x.requires_grad or any(p.requires_grad for p in self.nn1.parameters())):
            x3 = x.reshape(-1, P, 4)
            out = self.forward_selected(self.gather_points(x3, self.selected_points(x3)))
            return out.reshape(tuple(lead) + (out.shape[-1],))
        needs_graph = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.nn1.parameters()))
        if not needs_graph and P % 32 == 0 and P >= 32:
            # nothing to record: nn1 and the max in one fused launch (the per-point layers are never written)
            lins = [m for m in self.nn1 if isinstance(m, nn.Linear)]
            h = ops.pointnet_select(self._pack, x.reshape(-1, P, 4), [l.weight for l in lins], [l.bias for l in lins])[0]
        else:
            h = _run_mlp(self.nn1, x.reshape(-1, 4))
            h = ops.segmax(h.reshape(-1, P, h.shape[-1]))       # [B,512]
        out = _run_mlp(self.nn2, h)
        return out.reshape(tuple(lead) + (out.shape[-1],))


class SDFGenerator(nn.Module):
    """point_sdf_net.py:49-119: `num_layers` Linear layers of width `hidden_channels` with LayerNorm + ReLU, latent
    injected as a per-shape bias after layers 0 and num_layers/2, the input positions concatenated again at
    num_layers/2, last layer -> 1 channel without norm / activation."""

    def __init__(self, latent_channels, hidden_channels, num_layers, norm=True, dropout=0.0):
        super(SDFGenerator, self).__init__()
        assert num_layers % 2 == 0
        self.layers1 = None
        self.layers2 = None
        self.latent_channels = latent_channels
        self.hidden_channels = hidden_channels
        self.num_layers = num_layers
        self.norm = norm
        self.dropout = dropout
        self.lins = nn.ModuleList()
        self.norms = nn.ModuleList()
        fan_in, fan_out = 3, hidden_channels
        for i in range(num_layers):
            self.lins.append(nn.Linear(fan_in, fan_out))
            self.norms.append(nn.LayerNorm(fan_out))
            fan_in = hidden_channels + 3 if i == num_layers // 2 - 1 else hidden_channels
            if i == num_layers - 2:
                fan_out = 1
        self.z_lin1 = nn.Linear(latent_channels, hidden_channels)
        self.z_lin2 = nn.Linear(latent_channels, hidden_channels)
        self._pack = ops._GenPackCache()

    def _fused(self, pos):
        """The one-launch form covers the script's configuration; every other one runs layer by layer."""
        return (self.hidden_channels == 256 and self.num_layers == 8
                and all(n.elementwise_affine and n.eps == self.norms[0].eps for n in self.norms))

    def forward(self, pos, z):
        pos = pos.unsqueeze(0) if pos.dim() == 2 else pos
        assert pos.dim() == 3 and pos.size(-1) == 3
        z = z.unsqueeze(0) if z.dim() == 1 else z
        assert z.dim() == 2 and z.size(-1) == self.latent_channels
        assert pos.size(0) == z.size(0)
        if not self.norm or (self.dropout > 0.0 and self.training):
            raise NotImplementedError("the native path covers the training configuration (norm=True, dropout=0.0)")
        B, P = pos.shape[0], pos.shape[1]
        half = self.num_layers // 2
        pos2 = pos.reshape(B * P, 3)
        if self._fused(pos):
            # the latent enters layers 0 and num_layers/2 as one bias row per shape (:104-111)
            zb1 = ops.linear(z, self.z_lin1.weight, self.z_lin1.bias + self.lins[0].bias)
            zb5 = ops.linear(z, self.z_lin2.weight, self.z_lin2.bias + self.lins[half].bias)
            params = [t for lin in self.lins for t in (lin.weight, lin.bias)] + \
                     [t for norm in self.norms[:7] for t in (norm.weight, norm.bias)]
            out = ops.sdfgen_fused(self._pack, pos2, zb1, zb5, P, self.norms[0].eps, params)
            return out.reshape(B, P, 1)
        x = pos2
        for i, (lin, norm) in enumerate(zip(self.lins, self.norms)):
            last = i == self.num_layers - 1
            x = ops.linear(x, lin.weight, lin.bias)
            if last:
                zrow = None
                if i == 0 or i == half:   # degenerate 2-layer nets only
                    zrow = ops.linear(z, (self.z_lin1 if i == 0 else self.z_lin2).weight,
                                      (self.z_lin1 if i == 0 else self.z_lin2).bias)
                    x = x.reshape(B, P, -1) + zrow.unsqueeze(1)
                break
            zrow = None
            if i == 0:
                zrow = ops.linear(z, self.z_lin1.weight, self.z_lin1.bias)
            elif i == half:
                zrow = ops.linear(z, self.z_lin2.weight, self.z_lin2.bias)
            tail = pos2 if i == half - 1 else None        # the next layer reads cat([x, pos])
            x = ops.layernorm_act(x, zrow, P, norm.weight, norm.bias, norm.eps, ACT_RELU, tail)
        return x.reshape(B, P, -1)
