"""Voxel (variational) autoencoder, the reference's model/autoencoder.py:7-104 surface on HIP kernels.

Encoder: 3 x [Conv3d(k4,s2,p1) BN3d LeakyReLU], Conv3d(96->256,k4,s1) BN3d LeakyReLU, flatten, Linear(256,128)
[VAE: BN1d LeakyReLU + mean / log-variance heads].  Decoder: Linear(128,256) BN1d LeakyReLU, ConvT(256->96,k4,s1)
BN3d LeakyReLU, 2 x [ConvT(k4,s2,p1) BN3d LeakyReLU], ConvT(24->1,k4,s2,p1).  Module names/indices (and therefore
state_dict keys, including `encoder.vae-bn.*`) are the reference's.
"""
import torch
import torch.nn as nn

from ..util import device as default_device
from ..util import standard_normal_distribution
from . import LATENT_CODE_SIZE, Lambda, SavableModule
from .. import ops
from .stack import run_stack

AUTOENCODER_MODEL_COMPLEXITY_MULTIPLIER = 24
amcm = AUTOENCODER_MODEL_COMPLEXITY_MULTIPLIER


def _bn_lrelu(channels, dims):
    bn = nn.BatchNorm3d(channels) if dims == 3 else nn.BatchNorm1d(channels)
    return [bn, nn.LeakyReLU(negative_slope=0.2, inplace=True)]


class Autoencoder(SavableModule):
    def __init__(self, is_variational=True):
        super().__init__(filename="autoencoder-{:d}.to".format(LATENT_CODE_SIZE))
        self.is_variational = is_variational
        if is_variational:
            self.filename = 'variational-' + self.filename
        z = LATENT_CODE_SIZE

        enc = []
        for cin, cout in ((1, amcm), (amcm, 2 * amcm), (2 * amcm, 4 * amcm)):
            enc.append(nn.Conv3d(in_channels=cin, out_channels=cout, kernel_size=4, stride=2, padding=1))
            enc += _bn_lrelu(cout, 3)
        enc.append(nn.Conv3d(in_channels=4 * amcm, out_channels=2 * z, kernel_size=4, stride=1))
        enc += _bn_lrelu(2 * z, 3)
        enc.append(Lambda(lambda t: t.reshape(t.shape[0], -1)))
        enc.append(nn.Linear(in_features=2 * z, out_features=z))
        self.encoder = nn.Sequential(*enc)
        if is_variational:
            self.encoder.add_module('vae-bn', nn.BatchNorm1d(z))
            self.encoder.add_module('vae-lr', nn.LeakyReLU(negative_slope=0.2, inplace=True))
            self.encode_mean = nn.Linear(in_features=z, out_features=z)
            self.encode_log_variance = nn.Linear(in_features=z, out_features=z)

        dec = [nn.Linear(in_features=z, out_features=2 * z)] + _bn_lrelu(2 * z, 1)
        dec.append(Lambda(lambda t: t.reshape(-1, 2 * z, 1, 1, 1)))
        dec.append(nn.ConvTranspose3d(in_channels=2 * z, out_channels=4 * amcm, kernel_size=4, stride=1))
        dec += _bn_lrelu(4 * amcm, 3)
        for cin, cout in ((4 * amcm, 2 * amcm), (2 * amcm, amcm)):
            dec.append(nn.ConvTranspose3d(in_channels=cin, out_channels=cout, kernel_size=4, stride=2, padding=1))
            dec += _bn_lrelu(cout, 3)
        dec.append(nn.ConvTranspose3d(in_channels=amcm, out_channels=1, kernel_size=4, stride=2, padding=1))
        self.decoder = nn.Sequential(*dec)
        self.to(default_device)

    def encode(self, x, return_mean_and_log_variance=False):
        x = x.reshape((-1, 1, 32, 32, 32))
        x = run_stack(self.encoder, x, self.training)
        if not self.is_variational:
            return x
        mean = run_stack([self.encode_mean], x, self.training).squeeze()
        if self.training or return_mean_and_log_variance:
            log_variance = run_stack([self.encode_log_variance], x, self.training).squeeze()
            eps = standard_normal_distribution.sample(mean.shape).to(x.device)  # CPU draw, as the reference
        # mean + exp(0.5 * log_variance) * eps (model/autoencoder.py:77-82) as one native op (sg_vae_reparam_*)
        x = ops.vae_reparam(mean, log_variance, eps) if self.training else mean
        if return_mean_and_log_variance:
            return x, mean, log_variance
        return x

    def decode(self, x):
        if len(x.shape) == 1:
            x = x.unsqueeze(dim=0)
        return run_stack(self.decoder, x, self.training).squeeze()

    def forward(self, x):
        if not self.is_variational:
            return self.decode(self.encode(x))
        z, mean, log_variance = self.encode(x, return_mean_and_log_variance=True)
        return self.decode(z), mean, log_variance
