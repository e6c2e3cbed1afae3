"""Shared by oracle/make_golden_dropin.py (reference modules) and tests/test_dropin.py (native modules): which reference
scripts are executed, on what seeded synthetic data, and what is recorded / compared afterwards.

A case runs ONE epoch of the script's own `train()` loop in a scratch working directory that `prepare` fills with the files
the script expects (`data/chairs/voxels_32/*.npy`, `data/sdf_points.to`, `data/sdf_values.to`), after seeding the three RNGs
the scripts draw from (python `random`, numpy — create_batches' shuffles —, torch — module init, DataLoader shuffle,
generator.generate()'s latent draws).  `replace` are literal substitutions of module-level constants in the script text.
"""
import collections
import os
import random

import numpy as np
import torch

STRIDE = 101          # large tensors are compared on every STRIDE-th element
WHOLE_BELOW = 1 << 18  # floats; tensors below 1 MB are stored whole

Case = collections.namedtuple("Case", "name script argv epochs replace data seed device")

SDF_FIX = {"model_indices = indices / POINTCLOUD_SIZE": "model_indices = indices // POINTCLOUD_SIZE"}

CASES = [
    # BASELINE configs[0]: train_autoencoder.py classic, batch 4, 16 synthetic shapes (CPU plumbing; also run on the GPU)
    Case("ae_classic_b4", "train_autoencoder.py", ["classic", "nogui"], 1, {"BATCH_SIZE = 32": "BATCH_SIZE = 4"},
         ("voxels", 16), 0, "any"),
    Case("wgan_b4", "train_wgan.py", ["nogui"], 1, {"BATCH_SIZE = 64": "BATCH_SIZE = 4"}, ("voxels", 12), 1, "any"),
    Case("sdf_small", "train_sdf_autodecoder.py", ["nogui"], 1,
         dict(SDF_FIX, **{"POINTCLOUD_SIZE = 200000": "POINTCLOUD_SIZE = 2000", "BATCH_SIZE = 20000": "BATCH_SIZE = 1000"}),
         ("sdf", 4, 2000), 2, "any"),
    # the scripts' own constants (BASELINE configs[1] / [2] shapes): GPU only
    Case("wgan_b64", "train_wgan.py", ["nogui"], 1, None, ("voxels", 128), 3, "gpu"),
    Case("vae_b32", "train_autoencoder.py", ["nogui"], 1, None, ("voxels", 64), 4, "gpu"),
    Case("sdf_b20000", "train_sdf_autodecoder.py", ["nogui"], 1, SDF_FIX, ("sdf", 4, 200000), 5, "gpu"),
]
BY_NAME = {c.name: c for c in CASES}


def prepare(case):
    """Fills the CWD with the case's data files and seeds every RNG the script draws from."""
    rng = np.random.RandomState(1000 + case.seed)
    if case.data[0] == "voxels":
        os.makedirs("data/chairs/voxels_32", exist_ok=True)
        for i in range(case.data[1]):
            # blobby signed fields in [-0.2, 0.2]: the dataset clamps to +-0.1 and rescales (datasets.py:19-22)
            grid = (rng.rand(32, 32, 32).astype(np.float32) * 0.4 - 0.2)
            np.save("data/chairs/voxels_32/shape%03d.npy" % i, grid)
    else:
        shapes, pc = case.data[1], case.data[2]
        os.makedirs("data", exist_ok=True)
        pts = (rng.rand(shapes * pc, 3).astype(np.float32) * 2 - 1)
        sdf = (rng.rand(shapes * pc).astype(np.float32) * 0.3 - 0.15)
        torch.save(torch.from_numpy(pts), "data/sdf_points.to")
        torch.save(torch.from_numpy(sdf), "data/sdf_values.to")
    random.seed(case.seed)
    np.random.seed(case.seed)
    torch.manual_seed(case.seed)


def sample(t):
    """What is stored / compared for one tensor: (values, sum, abs-sum)."""
    a = t.detach().double().cpu().reshape(-1)
    vals = a if a.numel() < WHOLE_BELOW else a[::STRIDE]
    return vals.float().numpy(), np.array([a.sum().item(), a.abs().sum().item()])


def _record_module(rec, prefix, module):
    for k, v in module.state_dict().items():
        if v.is_floating_point():
            vals, sums = sample(v)
            rec["%s/%s" % (prefix, k)] = vals
            rec["%s/%s#sums" % (prefix, k)] = sums
        else:
            rec["%s/%s" % (prefix, k)] = v.detach().cpu().numpy()


def collect(case, ns):
    """Arrays describing what the script left behind: final module states (from the script's own objects), which files it
    saved, and the losses it logged."""
    rec = {}
    if case.script == "train_wgan.py":
        _record_module(rec, "generator", ns["generator"])
        _record_module(rec, "critic", ns["critic"])
        files = ["models/wgan-generator.to", "models/wgan-critic.to", "models/checkpoints/wgan-generator-epoch-00000.to",
                 "models/checkpoints/wgan-critic-epoch-00000.to", "plots/wgan_training.csv"]
        log = open("plots/wgan_training.csv").read().split()
        rec["log"] = np.array([float(log[2]), float(log[3])])           # mean critic value on fakes / reals (2 decimals)
    elif case.script == "train_autoencoder.py":
        ae = ns["autoencoder"]
        _record_module(rec, "autoencoder", ae)
        files = ["models/" + ae.filename, "models/checkpoints/" + ae.filename.replace(".to", "-epoch-00000.to")]
        rec["reconstruction_loss"] = np.array(list(ns["reconstruction_error_history"]), dtype=np.float64)
        rec["kld_loss"] = np.array(list(ns["kld_error_history"]), dtype=np.float64)
    else:
        _record_module(rec, "sdf_net", ns["sdf_net"])
        rec["latent_codes"] = ns["latent_codes"].detach().cpu().numpy()
        files = ["models/sdf_net.to", "models/sdf_net_latent_codes.to", "models/checkpoints/sdf_net-epoch-00000.to",
                 "models/checkpoints/sdf_net_latent_codes-epoch-00000.to", "plots/sdf_net_training.csv"]
        log = open("plots/sdf_net_training.csv").read().split()
        rec["log"] = np.array([float(log[2]), float(log[3])])           # mean loss of the epoch, latent std (6 decimals)
    for f in files:
        assert os.path.exists(f), "the script did not write " + f
    # the saved file equals the in-memory module (SavableModule.save, model/__init__.py:43-47)
    saved = torch.load(files[0], map_location="cpu")
    rec["saved_keys"] = np.array(sorted(saved.keys()))
    return rec


def initial_states(case, classes):
    """The modules as the script constructs them under the case's seed, sampled like the recorded finals.  `classes` maps
    the names the script imports (Generator, Discriminator, Autoencoder, SDFNet) to the implementation under test: the
    reference's and the native ones initialise bit-identically (tests/test_host_logic.py::test_state_dict_contract)."""
    out = {}
    if case.script == "train_wgan.py":
        torch.manual_seed(case.seed)
        mods = {"generator": classes["Generator"](), "critic": classes["Discriminator"]()}
    elif case.script == "train_autoencoder.py":
        torch.manual_seed(0)                                      # the script seeds itself (train_autoencoder.py:10-11)
        mods = {"autoencoder": classes["Autoencoder"](is_variational="classic" not in case.argv)}
    else:
        torch.manual_seed(case.seed)
        mods = {"sdf_net": classes["SDFNet"]()}
        out["latent_codes"] = torch.distributions.normal.Normal(0, 0.0001).sample((case.data[1], 128)).numpy()
    for prefix, m in mods.items():
        for k, v in m.state_dict().items():
            if v.is_floating_point():
                out["%s/%s" % (prefix, k)] = sample(v)[0]
    return out


def gradient_free(key):
    """Conv / ConvTranspose / Linear biases directly in front of a training-mode BatchNorm (gan.Generator layers 0/3/6,
    the Autoencoder's convs and its decoder's first Linear): their gradient is mathematically zero."""
    if not key.endswith(".bias"):
        return False
    if key.startswith("generator/layers."):
        return key.split(".")[1] in ("0", "3", "6")
    if key.startswith("autoencoder/encoder."):
        return key.split(".")[1] in ("0", "3", "6", "9") or key.endswith("encoder.13.bias")
    if key.startswith("autoencoder/decoder."):
        return key.split(".")[1] in ("0", "4", "7", "10")
    return False


META = ("saved_keys", "log", "reconstruction_loss", "kld_loss")


def update_disagreement(rec, ref, init, rel=0.1):
    """Per trained tensor: (fraction of entries whose UPDATE u = final - initial differs between the two runs by more than
    rel * mean|u_ref|, mean|u_ref|).  Skips buffers, integer state, gradient-free biases and untouched tensors."""
    out = {}
    for k, r in ref.items():
        if "#" in k or k in META or not np.issubdtype(r.dtype, np.floating) or "running_" in k or gradient_free(k):
            continue
        u_ref, u_got = r.astype(np.float64) - init[k], rec[k].astype(np.float64) - init[k]
        scale = float(np.abs(u_ref).mean())
        if scale == 0.0:
            continue
        out[k] = (float((np.abs(u_got - u_ref) > rel * scale).mean()), scale)
    return out
