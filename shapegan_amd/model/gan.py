"""Voxel GAN: `Generator` (z[128] -> 32^3 SDF) and `Discriminator` (32^3 -> score), the reference's
model/gan.py:4-69 surface on HIP kernels.

state_dict keys match the reference (`layers.{0,3,6,9}.{weight,bias}` + BatchNorm3d at `layers.{1,4,7}` for the
generator; `layers.{0,2,4,6}.{weight,bias}` for the discriminator), so checkpoints interchange.
"""
import torch
import torch.nn as nn

from .. import ops
from ..lib import ACT_SIGMOID
from ..util import device as default_device
from ..util import standard_normal_distribution
from . import LATENT_CODE_SIZE, Lambda, SavableModule
from .stack import run_stack, run_stack_groups

# (in, out, stride, padding) of the four transposed convolutions, model/gan.py:9-21
_G_CONVS = ((LATENT_CODE_SIZE, 256, 1, 0), (256, 128, 2, 1), (128, 64, 2, 1), (64, 1, 2, 1))
# model/gan.py:49-55
_D_CONVS = ((1, 64, 2, 1), (64, 128, 2, 1), (128, 256, 2, 1), (256, 1, 1, 0))


class Generator(SavableModule):
    def __init__(self):
        super().__init__(filename="generator.to")
        layers = []
        for idx, (cin, cout, stride, pad) in enumerate(_G_CONVS):
            layers.append(nn.ConvTranspose3d(in_channels=cin, out_channels=cout, kernel_size=4, stride=stride,
                                             padding=pad))
            if idx + 1 < len(_G_CONVS):
                layers += [nn.BatchNorm3d(cout), nn.LeakyReLU(negative_slope=0.2)]
            else:
                layers.append(nn.Tanh())
        self.layers = nn.Sequential(*layers)
        self.to(default_device)

    def forward(self, x, out=None):
        """out (optional, without grad mode): a contiguous fp32 [B,1,32,32,32] tensor the samples are written to (and which is
        then returned) — e.g. the fake half of the critic's batch."""
        x = x.reshape((-1, LATENT_CODE_SIZE, 1, 1, 1))
        return run_stack(self.layers, x, self.training, out)

    def forward_groups(self, zs, outs):
        """Several independent evaluations in one pass, without grad mode and in training mode (batch statistics): zs[g] [B,128]
        -> outs[g] ([B,1,32,32,32], equally spaced contiguous fp32 tensors), each evaluation with its own BatchNorm statistics and
        the running buffers updated evaluation after evaluation — the same results as `for z, o in zip(zs, outs): self(z, out=o)`
        with every convolution launched once (model/stack.py:run_stack_groups).  Falls back to exactly that loop when the fused
        path does not apply (grad mode, eval mode, CPU tensors of odd shapes, ...)."""
        if (not torch.is_grad_enabled() and self.training and len(zs) > 1 and len({tuple(z.shape) for z in zs}) == 1):
            x = _stacked_view(zs)                  # latent batches drawn as one tensor: no copy
            if x is None:
                x = torch.cat([z.reshape((-1, LATENT_CODE_SIZE)) for z in zs])
            x = x.reshape((-1, LATENT_CODE_SIZE, 1, 1, 1))
            if run_stack_groups(self.layers, x, len(zs), list(outs)):
                return outs
        for z, o in zip(zs, outs):
            y = self(z, out=o)
            if y.data_ptr() != o.data_ptr():
                o.copy_(y)
        return outs

    def generate(self, sample_size=1):
        # latents are drawn on the CPU and moved (model/gan.py:31-34): reproducible across backends
        z = standard_normal_distribution.sample(torch.Size((sample_size, LATENT_CODE_SIZE))).to(self.device)
        return self(z)

    def copy_autoencoder_weights(self, autoencoder):
        raise Exception("Not implemented.")  # as in the reference (model/gan.py:36-40)


def _stacked_view(zs):
    """[len(zs) * B, 128] view of the latent batches if they are consecutive contiguous slices of one tensor (e.g. the rows of one
    torch.randn(K, B, 128) draw), else None."""
    z0 = zs[0]
    if z0.dim() != 2 or not all(z.is_contiguous() and z.dtype == z0.dtype and z.device == z0.device for z in zs):
        return None
    step = z0.numel()
    for i, z in enumerate(zs):
        if (z.untyped_storage().data_ptr() != z0.untyped_storage().data_ptr()
                or z.storage_offset() != z0.storage_offset() + i * step):
            return None
    return z0.as_strided((len(zs) * z0.shape[0], z0.shape[1]), (z0.shape[1], 1), z0.storage_offset())


class Discriminator(SavableModule):
    def __init__(self):
        super().__init__(filename="discriminator.to")
        self.use_sigmoid = True
        layers = []
        for idx, (cin, cout, stride, pad) in enumerate(_D_CONVS):
            layers.append(nn.Conv3d(in_channels=cin, out_channels=cout, kernel_size=4, stride=stride, padding=pad))
            if idx + 1 < len(_D_CONVS):
                layers.append(nn.LeakyReLU(negative_slope=0.2))
        # `use_sigmoid` is read at call time, exactly like the reference's closure (model/gan.py:56)
        layers.append(Lambda(lambda t: ops.Act.apply(t, ACT_SIGMOID, 0.0) if self.use_sigmoid else t))
        self.layers = nn.Sequential(*layers)
        self.to(default_device)
        # the critic's weights change together (one optimizer step per update): their packed images are rebuilt in one launch
        ops.register_pack_group([m.weight for m in self.layers if isinstance(m, nn.Conv3d) and m.in_channels > 1])

    def forward(self, x):
        if len(x.shape) < 5:
            x = x.unsqueeze(dim=1)  # channel axis
        return run_stack(self.layers, x, self.training).squeeze()

    def clip_weights(self, value):
        """WGAN weight clipping (model/gan.py:67-69) — ONE clamp launch for all parameter tensors (sg_clamp_multi; it was one per
        tensor: eight launches per critic update of the reference's own loop), in place on the real storage.
        shapegan_amd.optim.RMSprop(clip=value) fuses it into the optimizer step instead."""
        import ctypes
        lib = ops.L.load()
        params = [p.data for p in self.parameters()]
        ptrs = (ctypes.c_void_p * len(params))(*[ops.ptr(t) for t in params])
        counts = (ctypes.c_long * len(params))(*[t.numel() for t in params])
        ops.check(lib.sg_clamp_multi(ptrs, counts, len(params), -value, value, ops.stream()), "clamp_multi")
        ops.L.bump_param_epoch()
