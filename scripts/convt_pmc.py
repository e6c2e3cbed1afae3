"""Counter target: the ConvTranspose3d(64 -> 1) forward alone (scripts/convt_pmc.sh): every kernel form of
sg_convT3d_k4s2p1_to1_pre_impl at 256 and 64 samples (identity input transform + tanh)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
wt = torch.randn(64, 1, 4, 4, 4, device="cuda") / 23.0
bt = torch.randn(1, device="cuda")
sc, sh = torch.ones(64, device="cuda"), torch.zeros(64, device="cuda")
forms = [int(f) for f in sys.argv[1:]] or [1, 3, 5]
for nb in (256, 64):
    y = torch.randn(nb, 64, 16, 16, 16, device="cuda")
    for form in forms:
        for _ in range(6):
            ops.conv_transpose3d_to1_pre_raw(y, sc, sh, 1, 0.2, wt, bt, 3, 0.0, form=form)
    torch.cuda.synchronize()
