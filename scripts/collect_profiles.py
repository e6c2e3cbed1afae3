"""gpurun_out/prof_<tag>/ (scripts/profile_round.sh on the GPU box) -> profiles/<tag>_*: the kernel-stat CSVs as they are, plus
two derived tables: MFMA utilisation per kernel (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs); the counter
CSV sums GRBM_GUI_ACTIVE over the 8 XCDs) and HBM traffic per kernel (FETCH_SIZE x 2 — calibrated in the same pass on 1 GiB
streams through b32 and b128 loads, both read exactly half — plus WRITE_SIZE, kilobytes -> bytes)."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", "prof_" + tag), os.path.join(root, "profiles")
for name in ("bench", "wgan_step", "sdf_train", "hybrid_progressive", "hybrid_wgan", "point_gan"):
    f = os.path.join(src, "%s_%s_kernel_stats.csv" % (tag, name))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for name in ("stream_calibration.json", "edge_kernels_by_batch.json", "point_gan_bench.txt", "wgan_step_timeline.txt", "sdf200k_step_timeline.txt",
             "sdf20k_step_timeline.txt"):
    f = os.path.join(src, "%s_%s" % (tag, name))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))


def table(path):
    by = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        by[r["kernel"]][r["counter"]] = (float(r["median_value"]), int(r["median_dispatch_ns"]), int(r["launches"]))
    return by


def mfma_table(name):
    path = os.path.join(src, "%s_%s_counters.csv" % (tag, name))
    if not os.path.exists(path):
        return []
    shutil.copy(path, os.path.join(dst, os.path.basename(path)))
    out = []
    for k, v in table(path).items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v["SQ_VALU_MFMA_BUSY_CYCLES"][0] > 1e6:
            busy, gui = v["SQ_VALU_MFMA_BUSY_CYCLES"][0], v["GRBM_GUI_ACTIVE"][0]
            ns = v["GRBM_GUI_ACTIVE"][1]
            out.append({"kernel": k, "launches": v["GRBM_GUI_ACTIVE"][2], "dispatch_us": round(ns / 1e3, 1),
                        "mfma_busy_cycles": busy, "gui_active_cycles_all_xcd": gui,
                        "mfma_util": round(busy / (gui / 8.0 * 1024.0), 4), "clock_ghz_while_profiled": round(gui / 8.0 / ns, 3)})
    out.sort(key=lambda r: -r["mfma_busy_cycles"])
    return out


def hbm_table(name):
    path = os.path.join(src, "%s_%s_counters.csv" % (tag, name))
    if not os.path.exists(path):
        return []
    shutil.copy(path, os.path.join(dst, os.path.basename(path)))
    rows = []
    for k, v in table(path).items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v and (v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) > 1000:
            fetch, write, ns = v["FETCH_SIZE"][0] * 1024 * 2, v["WRITE_SIZE"][0] * 1024, v["FETCH_SIZE"][1]
            rows.append({"kernel": k, "launches": v["FETCH_SIZE"][2], "dispatch_us": round(ns / 1e3, 1), "hbm_read_bytes": fetch,
                         "hbm_write_bytes": write, "hbm_gb_per_s": round((fetch + write) / ns, 1)})
    rows.sort(key=lambda r: -(r["hbm_read_bytes"] + r["hbm_write_bytes"]))
    return rows


out = mfma_table("mfma")
json.dump(out, open(os.path.join(dst, tag + "_mfma_utilisation.json"), "w"), indent=1)
rows = hbm_table("hbm")
json.dump(rows, open(os.path.join(dst, tag + "_hbm_traffic.json"), "w"), indent=1)
cfg_m, cfg_h = mfma_table("configs_mfma"), hbm_table("configs_hbm")
if cfg_m:
    json.dump(cfg_m, open(os.path.join(dst, tag + "_configs_mfma_utilisation.json"), "w"), indent=1)
if cfg_h:
    json.dump(cfg_h, open(os.path.join(dst, tag + "_configs_hbm_traffic.json"), "w"), indent=1)
dom = [r for r in rows if "conv_dgrad_halo_kernel<0>" in r["kernel"]]
if dom:
    json.dump({"kernel": dom[0]["kernel"], "hbm_bytes_per_launch": dom[0]["hbm_read_bytes"] + dom[0]["hbm_write_bytes"],
               "read": dom[0]["hbm_read_bytes"], "write": dom[0]["hbm_write_bytes"],
               "note": "median over the launches of two WGAN steps (128- and 64-sample shapes mixed; the median launch is the "
                       "128-sample critic shape: WRITE = dx [128,64,16^3] fp32 = 134 MB)"},
              open(os.path.join(dst, tag + "_dominant_kernel_hbm.json"), "w"), indent=1)
for r in out:
    print("%-60s util %.3f  %.0f us" % (r["kernel"][:60], r["mfma_util"], r["dispatch_us"]))
for r in rows[:14]:
    print("%-60s R %.1f MB W %.1f MB  %.0f GB/s" % (r["kernel"][:60], r["hbm_read_bytes"] / 1e6, r["hbm_write_bytes"] / 1e6, r["hbm_gb_per_s"]))
