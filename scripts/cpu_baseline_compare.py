"""The CPU baseline of bench.py timed twice on ONE box with the same thread count: on the reference's own module classes
(`kind: "reference"`, imported from the checkout by oracle/ref_import.py) and on oracle/torch_oracle.py's restatement
(`kind: "port"`, what the driver's GPU box — which has no reference checkout — can time).  Needs no GPU; run in the authoring
container:   python scripts/cpu_baseline_compare.py > profiles/rNN_cpu_baseline_reference_vs_port.json
The same 5+1 WGAN step at batch 64 from the same state_dict and inputs as bench.py's `cpu_baseline` leg."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

threads = min(32, os.cpu_count() or 1)
torch.set_num_threads(threads)
ref = ref_import.load()
torch.manual_seed(0)
g, c = ref.Generator().cpu(), ref.Discriminator().cpu()
g_state = {k: v.detach().clone() for k, v in g.state_dict().items()}
c_state = {k: v.detach().clone() for k, v in c.state_dict().items()}
gen = torch.Generator().manual_seed(1000)
B = bench.BATCH
reals = [torch.rand(B, 32, 32, 32, generator=gen) * 2 - 1 for _ in range(5)]
zs = [torch.randn(B, 128, generator=gen) for _ in range(5)]
zg = torch.randn(B, 128, generator=gen)
out = {"host_cores": os.cpu_count(), "cores": threads, "steps_timed": 2,
       "what": "train_wgan.py 5 critic + 1 generator updates at batch 64, torch CPU fp32, same inputs and initial state"}
losses = {}
for kind in ("reference", "port"):
    orc = bench._ReferenceWGAN(ref, g_state, c_state) if kind == "reference" else O.WGANOracle(g_state, c_state)
    first = orc.critic_step(reals[0], zs[0])       # warm-up, and the value both must agree on
    losses[kind] = float(first[0])
    t0 = time.perf_counter()
    for _ in range(out["steps_timed"]):
        orc.step(reals, zs, zg)
    dt = time.perf_counter() - t0
    out[kind] = {"steps_per_s": round(out["steps_timed"] / dt, 4), "s_per_step": round(dt / out["steps_timed"], 3),
                 "first_critic_loss": losses[kind]}
out["port_over_reference"] = round(out["port"]["steps_per_s"] / out["reference"]["steps_per_s"], 3)
out["first_critic_loss_equal"] = losses["reference"] == losses["port"]
print(json.dumps(out, indent=1))
