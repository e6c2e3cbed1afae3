"""Host-side helpers with the reference's `util` surface (util.py:1-85), restated.

Only the pieces the hot path touches are here: `device`, `standard_normal_distribution`,
`get_voxel_coordinates`, `get_points_in_unit_sphere`, `ensure_directory`, `create_text_slice`.
Unlike the reference, importing this module does not create plots/ models/ data/ in the CWD (util.py:11-13 does);
`SavableModule.save` creates `models/` on demand instead.
"""
import os

import numpy as np
import torch

# util.py:2-3
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
standard_normal_distribution = torch.distributions.normal.Normal(0, 1)


def ensure_directory(directory):
    if not os.path.exists(directory):
        os.makedirs(directory)


def get_voxel_coordinates(resolution=32, size=1, center=0, return_torch_tensor=False):
    """[R^3, 3] grid, flat index = x*R^2 + y*R + z, coordinates linspace(c-size, c+size, R) computed in float64 and
    cast to float32 last — the same operation order as util.py:60-74, so the grid is bit-identical."""
    if isinstance(center, int):
        center = (center, center, center)
    axes = [np.linspace(center[a] - size, center[a] + size, resolution) for a in range(3)]
    # meshgrid default 'xy' indexing followed by swapaxes(1, 2) == 'ij' indexing of (x, y, z)
    grid = np.stack(np.meshgrid(axes[0], axes[1], axes[2], indexing="ij"))
    points = grid.reshape(3, -1).transpose()
    if return_torch_tensor:
        return torch.tensor(points, dtype=torch.float32, device=device)
    return points.astype(np.float32)


def get_points_in_unit_sphere(n, device):
    """Rejection-sample n points in the unit ball (util.py:32-39)."""
    x = torch.rand(int(n * 2.5), 3, device=device) * 2 - 1
    keep = (torch.norm(x, dim=1) < 1).nonzero().squeeze()
    x = x[keep[:n], :]
    if x.shape[0] < n:
        print("Warning: Did not find enough points.")
    return x


_CHARACTERS = '      `.-:/+osyhdmm###############'


def create_text_slice(voxels):
    """ASCII rendering of one voxel slice (util.py:17-29)."""
    res = voxels.shape[-1]
    data = voxels[res // 4, :, :]
    data = (torch.clamp(data * -0.5 + 0.5, 0, 1) * (len(_CHARACTERS) - 1)).type(torch.int).cpu()
    lines = ['|' + ''.join(_CHARACTERS[i] for i in row) + '|' for row in data]
    picked = []
    for i in range(res):
        if len(picked) < i / 2.2:
            picked.append(lines[i])
    frame = '+' + '—' * res + '+\n'
    return frame + '\n'.join(reversed(picked)) + '\n' + frame
