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


class Lambda(nn.Module):
    """Parameter-free module around a callable (model/__init__.py:12-18)."""

    def __init__(self, function):
        super().__init__()
        self.function = function

    def forward(self, x):
        return self.function(x)


class SavableModule(nn.Module):
    """nn.Module with the reference's checkpoint naming (model/__init__.py:20-47):
    models/<filename>, models/checkpoints/<stem>-epoch-%05d.<ext>; optimizer state is never saved."""

    def __init__(self, filename):
        super().__init__()
        self.filename = filename

    def get_filename(self, epoch=None, filename=None):
        name = self.filename if filename is None else filename
        if epoch is None:
            return os.path.join(MODEL_PATH, name)
        parts = name.split('.')
        parts[-2] += '-epoch-{:05d}'.format(epoch)
        return os.path.join(CHECKPOINT_PATH, '.'.join(parts))

    def load(self, epoch=None):
        self.load_state_dict(torch.load(self.get_filename(epoch=epoch)), strict=False)
        from ..lib import bump_param_epoch
        bump_param_epoch()

    def save(self, epoch=None):
        os.makedirs(CHECKPOINT_PATH if epoch is not None else MODEL_PATH, exist_ok=True)
        # clone: parameters may be views into a flat optimizer buffer; keep the file per-tensor like the reference's
        torch.save({k: v.detach().clone() for k, v in self.state_dict().items()}, self.get_filename(epoch=epoch))

    @property
    def device(self):
        return next(self.parameters()).device
