"""Debug (GPU box): the 16-bit sign masks sg_sdfnet_fwd writes behind the H images against (H > 0) recomputed from the images."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import lib as L, ops
from shapegan_amd.lib import check, ptr, stream
from shapegan_amd.model.sdf_net import SDFNet

lib = L.load()
torch.manual_seed(0)
for N, Lz, S in ((200000, 256, 64), (20000, 128, 64), (777, 16, 3), (64, 16, 1)):
    net = SDFNet(latent_code_size=Lz)
    pts = torch.rand(N, 3, device="cuda") * 2 - 1
    sid = torch.sort(torch.randint(0, S, (N,), device="cuda"))[0].int()
    z = torch.randn(S, Lz, device="cuda") * 0.1
    params = net._params()
    packed = net._pack_shapes.get(params, Lz, 3)
    kin = 3 + Lz
    zb1 = ops.gemm_raw(z, False, params[0].detach(), True, bias_j=params[1].detach(), b_off=3, M=S, N=256, K=Lz, lda=Lz, ldb=kin)
    zb5 = ops.gemm_raw(z, False, params[8].detach(), True, bias_j=params[9].detach(), b_off=259, M=S, N=256, K=Lz, lda=Lz, ldb=256 + kin)
    out = torch.empty(N, device="cuda")
    acts = torch.full((lib.sg_sdfnet_acts_floats(N),), 7.0, device="cuda")
    check(lib.sg_sdfnet_fwd(ptr(pts), 0, None, None, Lz, ptr(packed), 3, ptr(zb1), ptr(zb5), 0, ptr(sid), ptr(out), ptr(acts), N, N, stream()), "fwd")
    torch.cuda.synchronize()
    H = acts[:7 * 256 * N].reshape(7, 256, N)
    mask = acts[7 * 256 * N:].view(torch.int16).reshape(7, 16, N).to(torch.int32) & 0xffff
    # expected: bit q of mask[l][2w+kh][p] = H[l][32w + (q&3) + 8(q>>2) + 4kh][p] > 0
    exp = torch.zeros_like(mask)
    for w in range(8):
        for kh in range(2):
            for q in range(16):
                row = 32 * w + (q & 3) + 8 * (q >> 2) + 4 * kh
                exp[:, 2 * w + kh, :] |= (H[:, row, :] > 0).to(torch.int32) << q
    bad = (mask != exp)
    print("N=%d L=%d: mask mismatches %d of %d words" % (N, Lz, int(bad.sum()), bad.numel()))
    if bad.any():
        idx = bad.nonzero()[:10]
        for l, g, p in idx.tolist():
            print("   layer %d group %d point %d: got %04x expected %04x" % (l, g, p, int(mask[l, g, p]), int(exp[l, g, p])))
        per_layer = bad.reshape(7, -1).sum(1).tolist()
        print("   per layer:", per_layer, " per group:", bad.sum((0, 2)).tolist())
        pb = bad.any(0).any(0).nonzero().flatten()
        print("   bad points from %d to %d, count %d" % (int(pb.min()), int(pb.max()), pb.numel()))
    # ---- backward: dZ_l must vanish exactly where H_l == 0 and (almost surely) nowhere else
    dout = torch.randn(N, device="cuda")
    dz = torch.full((7, 256, N), 3.0, device="cuda")
    dz8 = torch.empty(N, device="cuda")
    nblk = lib.sg_sdfnet_bwd_blocks(N)
    bsum = torch.empty((14 * 256, nblk), device="cuda")
    check(lib.sg_sdfnet_bwd(ptr(dout), ptr(out), ptr(acts), ptr(dz), ptr(dz8), ptr(bsum), ptr(pts), 0, None, 3, ptr(packed), 3, N, N, stream()), "bwd")
    torch.cuda.synchronize()
    for l in range(7):
        off_but_nonzero = int(((H[l] == 0) & (dz[l] != 0)).sum())
        on_but_zero = int(((H[l] > 0) & (dz[l] == 0)).sum())
        print("   layer %d: H==0 & dz!=0: %d   H>0 & dz==0: %d   (of %d)" % (l, off_but_nonzero, on_but_zero, H[l].numel()))
        if off_but_nonzero:
            badp = ((H[l] == 0) & (dz[l] != 0)).any(0).nonzero().flatten()
            badr = ((H[l] == 0) & (dz[l] != 0)).any(1).nonzero().flatten()
            print("      points %d..%d (%d), rows %s" % (int(badp.min()), int(badp.max()), badp.numel(), badr[:40].tolist()))

    # ---- backward values against a plain torch chain on the same H images
    P_ = [p.detach() for p in params]
    W = {2: P_[2], 3: P_[4], 4: P_[6], 5: P_[8][:, :256], 6: P_[10], 7: P_[12], 8: P_[14]}
    ref8 = dout * (1 - out * out)
    print("   dz8 max err %.3e" % float((dz8 - ref8).abs().max()))
    dH = W[8].t() @ ref8[None, :]                      # [256, N]
    for l in range(6, -1, -1):
        ref = dH * (H[l] > 0)
        err = float((dz[l] - ref).abs().max()) / max(float(ref.abs().mean()), 1e-30)
        print("   dZ%d max err / mean |ref| = %.3e" % (l + 1, err))
        if l > 0:
            dH = W[l + 1].t() @ ref                    # W of layer l+1 maps H_l -> Z_{l+1}
