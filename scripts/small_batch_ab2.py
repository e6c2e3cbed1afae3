"""Threshold scan for two dispatch rules (GPU box): Conv3d 64->128 forward at 16^3 (LDS-halo, 128- and 64-row tiles, against the
split-K gather kernel) and the 4^3 -> 8^3 input gradient 256 -> 128 (LDS-halo sample-pair kernel against the gather kernel)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shapegan_amd import ops
from small_batch_ab import t_us  # noqa: E402  (runs that script's table first when imported; cheap)

out = {}
w = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
for n in (4, 8, 12, 16, 20, 24, 32, 40, 48):
    x = torch.randn(n, 64, 16, 16, 16, device="cuda")
    out["fwd64to128_n%d" % n] = {"auto": t_us(lambda: ops.conv_fwd_raw(x, w, None)), "gather": t_us(lambda: ops.conv_fwd_impl_raw(x, w, None, 0, 0.0, 0)),
                                 "halo_auto_rows": t_us(lambda: ops.conv_fwd_impl_raw(x, w, None, 0, 0.0, 1)),
                                 "halo128": t_us(lambda: ops.conv_fwd_impl_raw(x, w, None, 0, 0.0, 1, debug=128)),
                                 "halo64": t_us(lambda: ops.conv_fwd_impl_raw(x, w, None, 0, 0.0, 1, debug=48))}
w2 = torch.randn(64, 32, 4, 4, 4, device="cuda") * 0.02
for n in (4, 8, 16, 32):
    x = torch.randn(n, 32, 32, 32, 32, device="cuda")
    out["fwd32to64_n%d" % n] = {"auto": t_us(lambda: ops.conv_fwd_raw(x, w2, None)), "gather": t_us(lambda: ops.conv_fwd_impl_raw(x, w2, None, 0, 0.0, 0)),
                                "halo": t_us(lambda: ops.conv_fwd_impl_raw(x, w2, None, 0, 0.0, 1))}
w4 = torch.randn(256, 128, 4, 4, 4, device="cuda") * 0.02
for n in (2, 4, 8, 12, 16, 24, 32, 48, 64):
    dy = torch.randn(n, 256, 4, 4, 4, device="cuda")
    out["dgrad256to128_n%d" % n] = {"auto": t_us(lambda: ops.conv_dgrad_raw(dy, w4, None, 128)),
                                    "halo_ppw1": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w4, None, 128, impl=3)),
                                    "halo_auto_ppw": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w4, None, 128, impl=1))}
    x8 = torch.randn(n, 128, 8, 8, 8, device="cuda")
    out["fwd128to256_n%d" % n] = {"auto": t_us(lambda: ops.conv_fwd_raw(x8, w4, None)), "halo4": t_us(lambda: ops.conv_fwd_impl_raw(x8, w4, None, 0, 0.0, 1))}
    out["wgrad128to256_n%d" % n] = {"auto": t_us(lambda: ops.conv_wgrad_raw(dy, x8, 128)), "halo4": t_us(lambda: ops.conv_wgrad_halo_raw(dy, x8, 128))}
w3 = torch.randn(128, 64, 4, 4, 4, device="cuda") * 0.02
for n in (2, 4, 8, 16):
    dy = torch.randn(n, 128, 8, 8, 8, device="cuda")
    out["dgrad128to64_n%d" % n] = {"auto": t_us(lambda: ops.conv_dgrad_raw(dy, w3, None, 64)),
                                   "halo_ppw1": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w3, None, 64, impl=3)),
                                   "halo_auto_ppw": t_us(lambda: ops.conv_dgrad_halo_raw(dy, w3, None, 64, impl=1))}
for k, v in out.items():
    print(k, json.dumps(v))
