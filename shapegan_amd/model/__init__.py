"""Drop-in mirror of the reference's `model` package surface (model/__init__.py:1-47).

Same names and behaviour: MODEL_PATH / CHECKPOINT_PATH / LATENT_CODES_FILENAME / LATENT_CODE_SIZE constants, the
`Lambda` wrapper and `SavableModule` (filename handling, `load(epoch)` with strict=False, `save(epoch)`, `.device`).
The modules built on top (gan.py, autoencoder.py, progressive_gan.py, sdf_net.py) keep the reference's class
names, constructor arguments, attributes and state_dict keys, but their forward passes run hand-written HIP
kernels through shapegan_amd.ops instead of ATen.
"""
import os

import torch
import torch.nn as nn

MODEL_PATH = "models"
CHECKPOINT_PATH = os.path.join(MODEL_PATH, 'checkpoints')
LATENT_CODES_FILENAME = os.path.join(MODEL_PATH, "sdf_net_latent_codes.to")
LATENT_CODE_SIZE = 128


def checkpoint_file(name, epoch=None):
    """Where a module called `name` lives on disk: `models/<name>`, or for an epoch snapshot
    `models/checkpoints/<stem>-epoch-%05d.<ext>` where <stem> is everything before the last dot
    (model/__init__.py:25-34)."""
    if epoch is None:
        return os.path.join(MODEL_PATH, name)
    stem, dot, ext = name.rpartition('.')
    if not dot:
        raise IndexError("checkpoint names need an extension: %r" % name)   # the reference indexes parts[-2]
    return os.path.join(CHECKPOINT_PATH, "%s-epoch-%05d.%s" % (stem, epoch, ext))


class Lambda(nn.Module):
    """Parameter-free module around a callable (model/__init__.py:12-18); shows up in state_dict as nothing."""

    def __init__(self, function):
        nn.Module.__init__(self)
        self.function = function

    def forward(self, x):
        return self.function(x)

    def extra_repr(self):
        return getattr(self.function, "__name__", "callable")


class SavableModule(nn.Module):
    """nn.Module with the reference's checkpoint protocol (model/__init__.py:20-47): a mutable `.filename`,
    `get_filename(epoch, filename)`, `load(epoch)` (strict=False), `save(epoch)`, `.device`.  Optimizer state is never
    part of a checkpoint, exactly as in the reference."""

    def __init__(self, filename):
        nn.Module.__init__(self)
        self.filename = filename

    def get_filename(self, epoch=None, filename=None):
        return checkpoint_file(self.filename if filename is None else filename, epoch)

    def load(self, epoch=None):
        # map_location: a file written from cuda:0 must not allocate on cuda:0 when rank r loads it onto cuda:r (process-per-GPU
        # DP); load_state_dict copies into the parameters where they live either way
        try:
            target = self.device
        except StopIteration:
            target = None
        state = torch.load(self.get_filename(epoch=epoch), map_location=target)
        self.load_state_dict(state, strict=False)
        from ..lib import bump_param_epoch
        bump_param_epoch()   # packed weight images derived from the old values are stale now

    def save(self, epoch=None):
        target = self.get_filename(epoch=epoch)
        os.makedirs(os.path.dirname(target), exist_ok=True)
        # per-tensor clones: parameters may be views into a flat optimizer buffer, the file should not drag it along
        torch.save({key: value.detach().clone() for key, value in self.state_dict().items()}, target)

    @property
    def device(self):
        for parameter in self.parameters():
            return parameter.device
        raise StopIteration("module without parameters has no device")
