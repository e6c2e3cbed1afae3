"""DeepSDF decoder `SDFNet` (model/sdf_net.py:23-61 + inference helpers :63-95,118-156) on the fused MFMA kernel.

state_dict keys: `layers1.{0,2,4,6}.{weight,bias}`, `layers2.{0,2,4,6}.{weight,bias}` — the shipped
examples/gan_generator_voxels_*.to checkpoints load unchanged.
"""
import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..util import get_points_in_unit_sphere, get_voxel_coordinates
from . import LATENT_CODE_SIZE, LATENT_CODES_FILENAME, SavableModule  # noqa: F401  (re-exported: train_sdf_autodecoder.py:13)

SDF_NET_BREADTH = 256


class SDFVoxelizationHelperData(object):
    """Cached sample grid (and unit-sphere mask, |p| < 1.1) per resolution — model/sdf_net.py:7-19."""

    def __init__(self, device, voxel_resolution, sphere_only=True):
        sample_points = get_voxel_coordinates(voxel_resolution)
        if sphere_only:
            unit_sphere_mask = np.linalg.norm(sample_points, axis=1) < 1.1
            sample_points = sample_points[unit_sphere_mask, :]
            self.unit_sphere_mask = unit_sphere_mask.reshape(voxel_resolution, voxel_resolution, voxel_resolution)
        self.sample_points = torch.tensor(sample_points, device=device)
        self.point_count = self.sample_points.shape[0]


sdf_voxelization_helper = dict()


def _mlp(sizes, last_act):
    mods = []
    for i in range(len(sizes) - 1):
        mods.append(nn.Linear(in_features=sizes[i], out_features=sizes[i + 1]))
        mods.append(nn.ReLU(inplace=True) if i + 2 < len(sizes) else last_act)
    return nn.Sequential(*mods)


class SDFNet(SavableModule):
    def __init__(self, latent_code_size=LATENT_CODE_SIZE, device='cuda'):
        super().__init__(filename="sdf_net.to")
        b = SDF_NET_BREADTH
        self.latent_code_size = latent_code_size
        # parameter containers only; the forward below never calls them (model/sdf_net.py:26-52)
        self.layers1 = _mlp([3 + latent_code_size, b, b, b, b], nn.ReLU(inplace=True))
        self.layers2 = _mlp([b + latent_code_size + 3, b, b, b, 1], nn.Tanh())
        self._pack_points = ops._PackCache()
        self._pack_shapes = ops._PackCache()
        if device == 'cuda' and not torch.cuda.is_available():
            device = 'cpu'  # construct-only (state_dict / checkpoint handling); forward needs the GPU
        self.to(device)

    def _params(self):
        out = []
        for seq in (self.layers1, self.layers2):
            for i in (0, 2, 4, 6):
                out += [seq[i].weight, seq[i].bias]
        return out

    def forward(self, points, latent_codes):
        """points [N,3], latent_codes [N,L] -> sdf [N]   (model/sdf_net.py:56-61)."""
        return ops.SDFNetPoints.apply(self._pack_points, points, latent_codes, torch.is_grad_enabled(), *self._params()).squeeze()

    def forward_shapes(self, points, latent_codes, points_per_shape):
        """points [S*pps,3], latent_codes [S,L] -> sdf [S*pps]: row s*pps+q uses latent s.  Same function as
        forward(points, latent.repeat_interleave(pps)) without materialising the tiled latents."""
        return ops.SDFNetShapes.apply(self._pack_shapes, points, latent_codes, int(points_per_shape), None, None, None,
                                      torch.is_grad_enabled(), *self._params())

    def prepare_latents(self, latent_table):
        """Weight pack (if stale) and per-shape latent fold for the forward_shapes / forward_segments call that follows with the same
        table — lets a trainer run them next to its batch assembly (ops._PackCache.prepare_fold)."""
        self._pack_shapes.prepare_fold(self._params(), latent_table)

    def forward_segments(self, points, latent_table, shape_index, segment_offsets, latent_reg=None):
        """points [N,3] grouped by shape, latent_table [S,L], shape_index [N] (int32), segment_offsets [S+1] (int64):
        sdf[i] = SDFNet(points[i], latent_table[shape_index[i]]).  The auto-decoder's latent_codes[model_indices]
        gather (train_sdf_autodecoder.py:80) without the [N,L] tensor: latent columns fold into per-shape biases and
        the latent-table gradient comes out dense, [S,L], from per-shape sums.
        latent_reg: None or (row_weight [S] | None, scale): the backward adds row_weight[s] * scale * latent_table[s] to the latent
        gradient (see ops.SDFNetShapes)."""
        return ops.SDFNetShapes.apply(self._pack_shapes, points, latent_table, 0, shape_index, segment_offsets, latent_reg,
                                      torch.is_grad_enabled(), *self._params())

    # ---- inference helpers (reference signatures) ----
    def evaluate_in_batches(self, points, latent_code, batch_size=100000, return_cpu_tensor=True):
        """One latent for all points (model/sdf_net.py:63-75).  The fused kernel streams any N in one launch, so
        `batch_size` only bounds the launch size."""
        n = points.shape[0]
        with torch.no_grad():
            z = latent_code.reshape(1, -1)
            result = torch.empty(n, dtype=torch.float32, device=points.device)
            for begin in range(0, n, batch_size):
                chunk = points[begin:begin + batch_size]
                result[begin:begin + chunk.shape[0]] = self.forward_shapes(chunk, z, chunk.shape[0])
        return result.cpu() if return_cpu_tensor else result

    def get_voxels(self, latent_code, voxel_resolution, sphere_only=True, pad=True):
        """SDF voxel grid; outside the |p|<1.1 sphere the grid is 1 (model/sdf_net.py:77-95)."""
        key = (voxel_resolution, sphere_only)
        if key not in sdf_voxelization_helper:
            sdf_voxelization_helper[key] = SDFVoxelizationHelperData(self.device, voxel_resolution, sphere_only)
        helper = sdf_voxelization_helper[key]
        distances = self.evaluate_in_batches(helper.sample_points, latent_code).numpy()
        if sphere_only:
            voxels = np.ones((voxel_resolution,) * 3, dtype=np.float32)
            voxels[helper.unit_sphere_mask] = distances
        else:
            voxels = distances.reshape(voxel_resolution, voxel_resolution, voxel_resolution)
            if pad:
                voxels = np.pad(voxels, 1, mode='constant', constant_values=1)
        return voxels

    def get_mesh(self, *args, **kwargs):
        raise NotImplementedError("marching cubes (skimage/trimesh, model/sdf_net.py:97-112) is outside the hot path")

    def get_normals(self, latent_code, points):
        """d sdf / d points, normalised (model/sdf_net.py:118-128)."""
        if latent_code.requires_grad or points.requires_grad:
            raise Exception('get_normals may only be called with tensors that don\'t require grad.')
        points.requires_grad = True
        sdf = self.forward_shapes(points, latent_code.reshape(1, -1), points.shape[0])
        sdf.backward(torch.ones(sdf.shape[0], device=self.device))
        normals = points.grad
        normals /= torch.norm(normals, dim=1).unsqueeze(dim=1)
        return normals

    def get_surface_points(self, latent_code, sample_size=100000, sdf_cutoff=0.1, return_normals=False,
                           use_unit_sphere=True):
        """Project random samples onto the zero level set along the SDF gradient (model/sdf_net.py:130-156)."""
        if use_unit_sphere:
            points = get_points_in_unit_sphere(n=sample_size, device=self.device) * 1.1
        else:
            points = torch.rand((sample_size, 3), device=self.device) * 2.2 - 1
        points.requires_grad = True
        sdf = self.forward_shapes(points, latent_code.reshape(1, -1), points.shape[0])
        sdf.backward(torch.ones((sdf.shape[0]), device=self.device))
        normals = points.grad
        normals /= torch.norm(normals, dim=1).unsqueeze(dim=1)
        points.requires_grad = False
        points -= normals * sdf.detach().unsqueeze(dim=1)
        mask = (torch.abs(sdf) < sdf_cutoff) & torch.all(torch.isfinite(points), dim=1)
        points, normals = points[mask, :], normals[mask, :]
        return (points, normals) if return_normals else points

    def get_surface_points_in_batches(self, latent_code, amount=1000):
        result = torch.zeros((amount, 3), device=self.device)
        position, tries = 0, 20
        while position < amount and tries > 0:
            points = self.get_surface_points(latent_code, sample_size=amount * 6)
            used = min(amount - position, points.shape[0])
            result[position:position + used, :] = points[:used, :]
            position += used
            tries -= 1
        return result
