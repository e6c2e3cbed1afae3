"""The executed drop-in: the REFERENCE's OWN training scripts (`/root/reference/train_wgan.py`, `train_autoencoder.py`,
`train_sdf_autodecoder.py` — their loops, stock torch.optim optimizers, zero_grad / clip_weights / generate / save, the
DataLoader over VoxelDataset) run unmodified on the shapegan_amd modules registered under the names the scripts import
(shapegan_amd.dropin), and what they leave behind is compared with what the same scripts left behind when run on the
reference's own modules (tests/golden/dropin.npz, written by oracle/make_golden_dropin.py in the authoring container).

The scripts live only in the reference checkout: without it (SHAPEGAN_REFERENCE_DIR, default /root/reference) these tests
skip.  Cases marked "any" run wherever the native modules can compute: on the GPU, and on CPU tensors through the
plain-C++ twin library (BASELINE configs[0] is exactly `ae_classic_b4` on the CPU); the "gpu" cases use the scripts' own
batch sizes and need the GPU.

Comparison.  One epoch is 3-40 optimizer steps.  Adam / RMSprop turn a gradient of any magnitude into a step of about lr
(the first RMSprop step is 10*lr*sign(g)), so for the few weights whose gradient is at fp32 rounding level two correct
runs step in opposite directions; everything else must agree closely.  Per tensor, with u = final - initial:
|u_native - u_reference| <= 0.1 * mean|u_reference| for all but a small fraction of the entries: 3 %, or three times the
fraction by which the REFERENCE differs from ITSELF when the same script runs on one thread instead of eight (recorded per
tensor in the fixture as `#noise`; it is < 0.1 % everywhere except the batch-4 autoencoder, whose last BatchNorm
normalises over 4 values per channel and amplifies rounding noise in the reference itself to ~9 % for the first conv).  Tensors the reference
leaves mathematically gradient-free (conv biases in front of a training-mode BatchNorm) are pure rounding noise in the
reference itself and are only required to stay within the step size; logged losses must agree to the printed precision.
"""
import os

import numpy as np
import pytest
import torch

import dropin_cases as cases
from conftest import GOLDEN

REF = os.environ.get("SHAPEGAN_REFERENCE_DIR", "/root/reference")
HAVE_REF = os.path.exists(os.path.join(REF, "train_wgan.py"))
HAVE_GOLDEN = os.path.exists(os.path.join(GOLDEN, "dropin.npz"))

needs_reference = pytest.mark.skipif(not (HAVE_REF and HAVE_GOLDEN),
                                     reason="reference scripts not present (set SHAPEGAN_REFERENCE_DIR)")


def _golden(case):
    z = np.load(os.path.join(GOLDEN, "dropin.npz"))
    p = case.name + "/"
    return {k[len(p):]: z[k] for k in z.files if k.startswith(p)}


def _native_classes():
    from shapegan_amd.model.autoencoder import Autoencoder
    from shapegan_amd.model.gan import Discriminator, Generator
    from shapegan_amd.model.progressive_gan import Discriminator as ProgressiveDiscriminator
    from shapegan_amd.model.sdf_net import SDFNet
    return {"Generator": Generator, "Discriminator": Discriminator, "Autoencoder": Autoencoder, "SDFNet": SDFNet,
            "ProgressiveDiscriminator": ProgressiveDiscriminator}


def compare(case, rec, gold, report=None):
    init = cases.initial_states(case, _native_classes())          # (a continued run: read from models/ in the CWD)
    # the reference run's own starting point, recorded in the fixture when the run was continued from saved files
    init_ref = {k[:-len("#init")]: v for k, v in gold.items() if k.endswith("#init")} or init
    assert sorted(rec["saved_keys"]) == sorted(gold["saved_keys"])
    for k, ref in gold.items():
        if "#" in k or k in cases.META:
            continue
        got = rec[k]
        assert got.shape == ref.shape, k
        if not np.issubdtype(ref.dtype, np.floating):
            np.testing.assert_array_equal(got, ref, err_msg=k)     # num_batches_tracked
        elif "running_" in k:
            # BatchNorm running statistics inherit the random walk of the gradient-free bias in front of them (a few
            # optimizer steps of +-lr each, entering with momentum 0.1): 2 % of the typical entry
            np.testing.assert_allclose(got, ref, rtol=2e-2, atol=2e-2 * float(np.abs(ref).mean()) + 1e-6, err_msg=k)
        elif cases.gradient_free(k):
            u_ref, u_got = ref.astype(np.float64) - init_ref[k], got.astype(np.float64) - init[k]
            assert float(np.abs(u_got).max()) <= 3.0 * float(np.abs(u_ref).max()) + 1e-12, k
        elif float(np.abs(ref.astype(np.float64) - init_ref[k]).mean()) == 0.0:
            assert np.array_equal(got, init[k].astype(got.dtype)), k + ": the reference leaves this tensor untouched"
    worst = 0.0
    for k, (frac, scale) in cases.update_disagreement(rec, gold, init, init_ref=init_ref).items():
        noise = float(gold.get(k + "#noise", 0.0))
        worst = max(worst, frac)
        if report is not None:
            report.append((k, frac, noise, scale))
        bound = max(0.03, 3.0 * noise) + 2.0 / gold[k].size
        assert frac <= bound, "%s: %.2f%% of the updates differ from the reference run (the reference differs from itself by %.2f%%)" % (
            k, 100 * frac, 100 * noise)
    if "log" in gold:
        atol = {"train_wgan.py": 0.011, "train_hybrid_wgan.py": 1.1e-4, cases.PROG: 1.1e-4}.get(case.script, 2e-6)   # printed decimals
        # the hybrid scripts log means of discriminator outputs taken AFTER 1-6 RMSprop updates whose first steps are
        # 10 * lr * sign(g) = 1e-3 per weight (lr 1e-4): weights whose gradient is at rounding level step the other way in two
        # correct fp32 runs and the later outputs inherit that (measured on the MI355X: 4e-4 / 1.2e-3 relative after the
        # continued 16^3 run, 6e-5 after the first three updates)
        rtol = 3e-3 if case.script in ("train_hybrid_wgan.py", cases.PROG) else 2e-4
        np.testing.assert_allclose(rec["log"], gold["log"], rtol=rtol, atol=atol)
    if "fade_in_progress" in gold:
        assert float(rec["fade_in_progress"]) == float(gold["fade_in_progress"])
    if "reconstruction_loss" in gold:
        np.testing.assert_allclose(rec["reconstruction_loss"], gold["reconstruction_loss"], rtol=2e-4)
        np.testing.assert_allclose(rec["kld_loss"], gold["kld_loss"], rtol=2e-4, atol=1e-7)
    return worst


def run_case(case, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    rec, ns = cases.run(case, REF, aliases=True)
    # the classes the script imported by the reference's names are the native ones
    for name in ("Generator", "Discriminator", "Autoencoder", "SDFNet"):
        if name in ns:
            assert ns[name].__module__.startswith("shapegan_amd.model"), name
    return rec, ns


@needs_reference
@pytest.mark.gpu
@pytest.mark.parametrize("name", [c.name for c in cases.CASES])
def test_reference_script_on_native_modules_gpu(name, tmp_path, monkeypatch):
    case = cases.BY_NAME[name]
    rec, ns = run_case(case, tmp_path, monkeypatch)
    for key in ("generator", "critic", "discriminator", "autoencoder", "sdf_net"):
        if key in ns:
            assert next(ns[key].parameters()).is_cuda
    compare(case, rec, _golden(case))


@needs_reference
@pytest.mark.parametrize("name", [c.name for c in cases.CASES if c.device == "any"])
def test_reference_script_on_native_modules_cpu(name, tmp_path, monkeypatch):
    """BASELINE configs[0] as written — `train_autoencoder.py classic`, batch 4, 16 synthetic shapes, on the CPU, no GPU — plus
    train_wgan.py and train_sdf_autodecoder.py at CPU-sized constants: the reference's own script text on the native modules
    with every tensor on the CPU, i.e. through libshapegan_cpu.so (the plain-C++ twin of the C ABI).  On a GPU box the
    scripts' `device` is pinned to the CPU for this test; the HIP library is not asked to compute."""
    import shapegan_amd.util as U
    import shapegan_amd.model.gan as G
    import shapegan_amd.model.autoencoder as A
    import shapegan_amd.model.sdf_net as S
    import shapegan_amd.lib as L
    cpu = torch.device("cpu")
    monkeypatch.setattr(U, "device", cpu)
    monkeypatch.setattr(G, "default_device", cpu)
    monkeypatch.setattr(A, "default_device", cpu)
    if torch.cuda.is_available():
        monkeypatch.setattr(S.SDFNet.__init__, "__defaults__", (128, "cpu"))
    L.load_cpu()
    case = cases.BY_NAME[name]
    rec, ns = run_case(case, tmp_path, monkeypatch)
    for key in ("generator", "critic", "discriminator", "autoencoder", "sdf_net"):
        if key in ns:
            assert not next(ns[key].parameters()).is_cuda
    compare(case, rec, _golden(case))
