"""Debug: sg_sdfnet_fwd / sg_sdfnet_bwd (per-point mode) against a plain torch evaluation on the GPU, tensor by tensor."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
from shapegan_amd.lib import check, ptr, stream
from shapegan_amd.model.sdf_net import SDFNet
N = int(sys.argv[1]) if len(sys.argv) > 1 else 63
Lz = 128
torch.manual_seed(N)
net = SDFNet(latent_code_size=Lz).cuda()
P = [p.detach() for p in net.parameters()]
pts = (torch.rand(N, 3, device="cuda") * 2 - 1).requires_grad_(True)
lat = (torch.randn(N, Lz, device="cuda") * 0.5).requires_grad_(True)
x = torch.cat([pts, lat], 1)
hs, h = [], x
for l in range(4):
    h = torch.relu(h @ P[2 * l].t() + P[2 * l + 1]); h.retain_grad(); hs.append(h)
h = torch.cat([h, x], 1)
for l in range(4, 7):
    h = torch.relu(h @ P[2 * l].t() + P[2 * l + 1]); h.retain_grad(); hs.append(h)
out_ref = torch.tanh(h @ P[14].t() + P[15]).reshape(-1)
dy = torch.randn(N, device="cuda")
out_ref.backward(dy)
lib = ops._lib()
kin = 3 + Lz
packed = net._pack_points.get(list(net.parameters()), Lz, kin) if hasattr(net, "_pack_points") else None
if packed is None:
    cache = ops._PackCache(); packed = cache.get(list(net.parameters()), Lz, kin)
out = torch.empty(N, device="cuda"); acts = torch.full((lib.sg_sdfnet_acts_floats(N),), 7.0, device="cuda")
check(lib.sg_sdfnet_fwd(ptr(pts.detach()), 0, ptr(lat.detach()), None, Lz, ptr(packed), kin, None, None, 0, None, ptr(out), ptr(acts), N, N, stream()), "fwd")
print("out err", float((out - out_ref).abs().max()))
for l in range(7):
    print("acts", l, float((acts[l] - hs[l].detach().t()).abs().max()))
dz = torch.full((7, 256, N), 9.0, device="cuda"); dz8 = torch.empty(N, device="cuda"); dx = torch.full((N, kin), 5.0, device="cuda")
nb = lib.sg_sdfnet_bwd_blocks(N)
bsum = torch.full((14 * 256, nb), 3.0, device="cuda")
check(lib.sg_sdfnet_bwd(ptr(dy), ptr(out), ptr(acts), ptr(dz), ptr(dz8), ptr(bsum), ptr(pts.detach()), 0, ptr(dx), kin, ptr(packed), kin, N, N, stream()), "bwd")
torch.cuda.synchronize()
for l in range(7):
    ref = (hs[l].grad * (hs[l].detach() > 0)).t()
    print("dz", l, float((dz[l] - ref).abs().max()), "scale", float(ref.abs().max()), " bsum err", float((bsum[l * 256:(l + 1) * 256].sum(1) - ref.sum(1)).abs().max()))
print("dx pts err", float((dx[:, :3] - pts.grad).abs().max()), "scale", float(pts.grad.abs().max()))
print("dx lat err", float((dx[:, 3:] - lat.grad).abs().max()), "scale", float(lat.grad.abs().max()))
bad = ((dx[:, :3] - pts.grad).abs() > 1e-5).nonzero()
print("bad rows", sorted(set(bad[:, 0].tolist()))[:70])
