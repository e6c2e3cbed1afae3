"""Progressively grown 3-D CNN discriminator (model/progressive_gan.py:4-61) on HIP kernels.

Four optional Conv3d(k4,s2,p1)+LeakyReLU stages for 8^3/16^3/32^3/64^3 inputs and a Linear head; both registrations
of every stage (`optional_layers.i.*` and `optional_layer_i.*`) are kept so state_dict keys match the reference.
`from_SDF`'s C-1 zero channels are never materialised: the first stage's kernel reads channel 0 of its weight only,
its weight gradient for the other input channels is exactly zero (as in the reference, whose inputs there are 0).
"""
import torch
import torch.nn as nn

from ..lib import ACT_LEAKY
from .. import ops
from . import LATENT_CODE_SIZE, Lambda, SavableModule  # noqa: F401  (LATENT_CODE_SIZE re-exported like the reference)
from .stack import run_stack

RESOLUTIONS = [8, 16, 32, 64]
FEATURE_COUNTS = [128, 64, 32, 1]
FINAL_LAYER_FEATURES = 256


def from_SDF(x, iteration):
    """Reference semantics (model/progressive_gan.py:9-16, "fromRGB"): [B,R,R,R] -> [B,C,R,R,R], the grid on channel 0
    and C-1 zero channels behind it.  Kept for callers that want the materialised tensor and for the fade-in blend;
    the discriminator's own first stage never builds it."""
    side, channels = RESOLUTIONS[iteration], FEATURE_COUNTS[iteration]
    grid = x.reshape((-1, 1, side, side, side))
    return torch.nn.functional.pad(grid, (0, 0, 0, 0, 0, 0, 0, channels - 1))   # zeros appended along the channel axis


def _stage(cin, cout):
    return nn.Sequential(nn.Conv3d(cin, cout, kernel_size=4, stride=2, padding=1), nn.LeakyReLU(negative_slope=0.2))


class Discriminator(SavableModule):
    """model/progressive_gan.py:18-61.  Module creation order (head, then stages 0..3) is the reference's, so the default
    initialisation under a given seed is bit-identical."""

    def __init__(self):
        self.filename_base = "hybrid_progressive_gan_discriminator_{:d}.to"
        SavableModule.__init__(self, filename=self.filename_base.format(0))
        self.iteration = 0
        self.fade_in_progress = 1

        flat = 64 * FINAL_LAYER_FEATURES
        self.head = nn.Sequential(Lambda(lambda t: t.reshape(-1, flat)), nn.Linear(flat, 128),
                                  nn.LeakyReLU(negative_slope=0.2), nn.Linear(128, 1))
        widths = [FINAL_LAYER_FEATURES] + FEATURE_COUNTS          # stage i maps widths[i + 1] -> widths[i] channels
        self.optional_layers = nn.ModuleList()
        for index in range(len(FEATURE_COUNTS)):
            stage = _stage(widths[index + 1], widths[index])
            self.optional_layers.append(stage)
            self.add_module('optional_layer_{:d}'.format(index), stage)    # second registration: both key sets exist

    def forward(self, x):
        it = self.iteration
        res = RESOLUTIONS[it]
        x_in = x
        # stage `it` on the single real channel (the conv kernel skips the zero-padded channels of from_SDF)
        x = x.reshape((-1, 1, res, res, res))
        conv = self.optional_layers[it][0]
        x = ops.conv3d_k4s2p1(x, conv.weight, conv.bias, ACT_LEAKY, 0.2)
        if (self.fade_in_progress < 1.0) and it > 0:
            # blend with the nearest-neighbour downsampled input injected on channel 0 (progressive_gan.py:48-50)
            # one fused pass: the nearest-neighbour subsample lands on channel 0, from_SDF's zero channels are not built
            x = ops.fade_blend(x, x_in.reshape((-1, res, res, res)), self.fade_in_progress)
        for i in range(it - 1, -1, -1):
            x = run_stack(self.optional_layers[i], x, self.training)
        return run_stack(self.head, x, self.training).squeeze()

    def set_iteration(self, value):
        self.iteration = value
        self.filename = self.filename_base.format(value)
