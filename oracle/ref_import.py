"""Imports the REAL reference (marian42/shapegan at /root/reference) on CPU.  Authoring-container only.

Shims: model/sdf_net.py:2-3 import trimesh / skimage.measure at top level (not installed,
only used by get_mesh) -> empty stub modules; gan.py:25,59 / autoencoder.py:65 call self.cuda() -> no-op on a
machine without a GPU; `import util` creates plots/ models/ data/ in the CWD (util.py:11-13) -> chdir to a temp dir.
"""
import os
import sys
import tempfile
import types

REFERENCE_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "model"))


_cache = {}


def load():
    """Returns a namespace with the reference's classes and helpers."""
    if _cache:
        return _cache["ns"]
    if not available():
        raise RuntimeError("reference not present at %s" % REFERENCE_ROOT)
    import torch
    import torch.nn as nn
    for m in ("trimesh", "skimage", "skimage.measure"):
        sys.modules.setdefault(m, types.ModuleType(m))
    if not torch.cuda.is_available():
        nn.Module.cuda = lambda self, *a, **k: self
    cwd = os.getcwd()
    scratch = tempfile.mkdtemp(prefix="shapegan_ref_")
    os.chdir(scratch)
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "model" or k.startswith("model.") or k == "util"}
    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import model.gan as gan
        import model.autoencoder as autoencoder
        import model.progressive_gan as progressive_gan
        import model.sdf_net as sdf_net
        import util
    finally:
        sys.path.remove(REFERENCE_ROOT)
        os.chdir(cwd)
    ns = types.SimpleNamespace(
        Generator=gan.Generator, Discriminator=gan.Discriminator, Autoencoder=autoencoder.Autoencoder,
        ProgressiveDiscriminator=progressive_gan.Discriminator, SDFNet=sdf_net.SDFNet, util=util,
        get_voxel_coordinates=util.get_voxel_coordinates, root=REFERENCE_ROOT)
    # keep the reference's `model`/`util` out of the way of anything else named like that
    for k in [k for k in list(sys.modules) if k == "model" or k.startswith("model.") or k == "util"]:
        ns.__dict__.setdefault("_modules", {})[k] = sys.modules.pop(k)
    sys.modules.update(saved)
    _cache["ns"] = ns
    return ns
