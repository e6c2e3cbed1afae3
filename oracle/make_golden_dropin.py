"""TEST INFRASTRUCTURE — runs the REFERENCE's OWN training scripts on the REFERENCE's OWN modules (CPU, authoring
container only) and records what they leave behind, as the expected result of the executed drop-in tests.

    python oracle/make_golden_dropin.py            # -> tests/golden/dropin.npz

Each case (tests/dropin_cases.py) builds a seeded synthetic data directory in a scratch CWD, seeds the RNGs and executes
`/root/reference/train_*.py` through shapegan_amd.dropin.run_script(aliases=False): `model`, `util`, `datasets` resolve to
the reference's files.  tests/test_dropin.py then executes the SAME script text with the names aliased to shapegan_amd
and compares checkpoints and logged losses with what is recorded here.  Large tensors are stored as a strided sample
(every STRIDE-th element) plus (sum, abs-sum); tensors below 1 MB are stored whole.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dropin_cases as cases  # noqa: E402
from shapegan_amd import dropin  # noqa: E402

REFERENCE_ROOT = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden", "dropin.npz")


def main(only=None):
    import torch.nn as nn
    for m in ("trimesh", "skimage", "skimage.measure"):                 # model/sdf_net.py:2-3 (only get_mesh uses them)
        sys.modules.setdefault(m, types.ModuleType(m))
    if not torch.cuda.is_available():
        nn.Module.cuda = lambda self, *a, **k: self                     # gan.py:25,59 / autoencoder.py:65 call self.cuda()
    sys.path.insert(0, REFERENCE_ROOT)
    out = {}
    if only and os.path.exists(OUT):
        out = dict(np.load(OUT))
    home = os.getcwd()

    def run_reference(case, threads):
        os.chdir(tempfile.mkdtemp(prefix="dropin_ref_"))
        saved_threads = torch.get_num_threads()
        torch.set_num_threads(threads)
        try:
            for k in [k for k in sys.modules if k in ("model", "util", "datasets") or k.startswith("model.")]:
                del sys.modules[k]                                       # fresh import per run (util.py:11-13 makes plots/ ...)
            if not torch.cuda.is_available():
                import model.sdf_net as ref_sdf_net                      # SDFNet(latent_code_size=128, device='cuda'):
                ref_sdf_net.SDFNet.__init__.__defaults__ = (128, 'cpu')  # no GPU in the authoring container
            rec, ns = cases.run(case, REFERENCE_ROOT, aliases=False)
            classes = {n: ns[n] for n in ("Generator", "Discriminator", "Autoencoder", "SDFNet") if n in ns}
            if case.script == cases.PROG:
                classes["ProgressiveDiscriminator"] = classes.pop("Discriminator")
            return rec, cases.initial_states(case, classes)              # (in the run's directory: a continued run reads models/)
        finally:
            torch.set_num_threads(saved_threads)
            os.chdir(home)

    for case in cases.CASES:
        if only and case.name not in only:
            continue
        for k in [k for k in out if k.startswith(case.name + "/")]:
            del out[k]                                                   # regenerated: nothing of the old record survives
        rec, init = run_reference(case, torch.get_num_threads())
        # the reference against itself: the same run on ONE thread (different fp32 summation order).  How far its updates
        # move is the noise floor the native run is allowed (tests/test_dropin.py::compare).
        rec1, init1 = run_reference(case, 1)
        for k, (frac, _) in cases.update_disagreement(rec1, rec, init1, init_ref=init).items():
            rec[k + "#noise"] = np.array(frac)
        if case.pre:
            for k, v in init.items():
                rec[k + "#init"] = v                                     # where the recorded run started from
        for k, v in rec.items():
            out["%s/%s" % (case.name, k)] = v
        print(case.name, "->", len(rec), "arrays; largest self-disagreement %.4f" %
              max([float(v) for k, v in rec.items() if k.endswith("#noise")] + [0.0]), flush=True)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "%.1f MB" % (os.path.getsize(OUT) / 1e6))


if __name__ == "__main__":
    main(set(sys.argv[1:]) or None)
