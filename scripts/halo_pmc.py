import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
dbg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
x = torch.randn(int(os.environ.get('SG_PMC_BATCH', '128')), 64, 16, 16, 16, device="cuda"); w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
b = torch.zeros(128, device="cuda")
for _ in range(5): ops.conv_fwd_impl_raw(x, w, b, 1, 0.2, 1, dbg)
torch.cuda.synchronize()
