"""Consistency fuzz of the SDFNet training kernels at random ragged sizes: a shape-sorted batch (per-shape bias fold, tile partials,
segment sums, fold backward) evaluated in one call against the same batch evaluated as two calls cut at a random point (other tile
plans, other partial-sum layouts; identical per-point arithmetic, so no ReLU kink can flip between the two) — outputs bit-equal,
latent-table and parameter gradients equal to summation-order rounding.  Run under `timeout`; prints the worst deviation."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd.model.sdf_net import SDFNet
random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
torch.manual_seed(0)
worst = 0.0
nets = {L: SDFNet(latent_code_size=L).cuda() for L in (16, 128)}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    L = random.choice((16, 128))
    net = nets[L]
    N = random.choice((1, 31, 33, 63, 64, 65, 127, 129)) if it % 5 == 0 else random.randint(1, random.choice((300, 5000, 40000, 70000)))
    S = random.randint(1, min(40, N))
    sid = torch.sort(torch.randint(0, S, (N,), device="cuda"))[0]
    counts = torch.bincount(sid, minlength=S)
    seg = torch.zeros(S + 1, dtype=torch.int64, device="cuda"); seg[1:] = torch.cumsum(counts, 0)
    pts = torch.rand(N, 3, device="cuda") * 2 - 1
    table = torch.randn(S, L, device="cuda") * 0.5
    w = torch.randn(N, device="cuda")
    def run(lo, hi):
        t = table.clone().requires_grad_(True)
        for p in net.parameters(): p.grad = None
        sg = torch.zeros(S + 1, dtype=torch.int64, device="cuda")
        sg[1:] = torch.cumsum(torch.bincount(sid[lo:hi], minlength=S), 0)
        out = net.forward_segments(pts[lo:hi].contiguous(), t, sid[lo:hi].int().contiguous(), sg)
        (out * w[lo:hi]).sum().backward()
        return out.detach(), t.grad.clone(), [p.grad.clone() for p in net.parameters()]
    full = run(0, N)
    if N > 1:
        cut = random.randint(1, N - 1)
        a, b = run(0, cut), run(cut, N)
        halves = (torch.cat([a[0], b[0]]), a[1] + b[1], [x + y for x, y in zip(a[2], b[2])])
    else:
        halves = full
    res = [full, halves]
    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().mean() + 1e-30))
    errs = [float((res[0][0] - res[1][0]).abs().max()), rel(res[0][1], res[1][1])] + [rel(a, b) for a, b in zip(res[0][2], res[1][2])]
    if not all(torch.isfinite(torch.tensor(errs))) or errs[0] != 0.0 or max(errs[1:]) > 2e-3:
        print("MISMATCH N=%d S=%d L=%d" % (N, S, L), errs); sys.exit(1)
    worst = max(worst, max(errs[1:]))
print("fuzz ok, worst relative gradient deviation %.2e" % worst)
