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
for name in ("bench", "wgan_step", "sdf_train", "hybrid_progressive", "hybrid_wgan", "point_gan", "point_gan_critic", "point_gan_generator"):
    f = os.path.join(src, "%s_%s_kernel_stats.csv" % (tag, name))
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, os.path.basename(f)))
for name in ("stream_calibration.json", "edge_kernels_by_batch.json", "point_gan_bench.txt", "wgan_step_timeline.txt", "sdf200k_step_timeline.txt",
             "sdf20k_step_timeline.txt", "sdf20k_graphed_step_timeline.txt", "convT_c1_counters.txt", "bench_line.json", "bench_line_2ranks_gloo_one_gpu.json",
             "pytest_gpu.log", "dropin_gpu.log", "write_pattern.jsonl", "edge_kernels_cold.json", "fwd_c1_ablation.json",
             "convT_c1_ablation.json", "dgrad_paired_stores.json", "convT_forms.json", "sdfnet_bwd_ablation.txt", "sdfnet_counters.txt",
             "dropin_loop_kernel_stats.txt", "bench_line_2ranks_strong_gloo_one_gpu.json"):
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
            row = {"kernel": k, "launches": v["GRBM_GUI_ACTIVE"][2], "dispatch_us": round(ns / 1e3, 1),
                   "mfma_busy_cycles": busy, "gui_active_cycles_all_xcd": gui, "mfma_util": round(busy / (gui / 8.0 * 1024.0), 4)}
            if ns >= 200000:      # GUI_ACTIVE / dispatch time is no clock estimate for short kernels (2.98 "GHz" at 30 us)
                row["clock_ghz_while_profiled"] = round(gui / 8.0 / ns, 3)
            out.append(row)
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
# the kernel with the largest share of the WGAN step (the same rule bench.py's `roofline` follows)
dom_name = None
stats = os.path.join(dst, tag + "_wgan_step_kernel_stats.csv")
if os.path.exists(stats):
    top = max(csv.DictReader(open(stats)), key=lambda r: float(r["Percentage"]))
    dom_name = top["Name"].split("(")[0].replace("void ", "").strip()
dom = [r for r in rows if dom_name and dom_name.replace(" ", "") in r["kernel"].replace(" ", "")]
if dom:
    json.dump({"kernel": dom[0]["kernel"], "hbm_bytes_per_launch": dom[0]["hbm_read_bytes"] + dom[0]["hbm_write_bytes"],
               "read": dom[0]["hbm_read_bytes"], "write": dom[0]["hbm_write_bytes"],
               "note": "median launch of two WGAN steps = the critic's 128-sample shape ([128,128,8^3] -> [128,64,16^3]: 134 MB of output, "
                       "35.6 MB of operands); the step also runs the kernel at 64 and 256 samples.  Per-shape figures: "
                       "<tag>_dgrad_paired_stores.json"},
              open(os.path.join(dst, tag + "_dominant_kernel_hbm.json"), "w"), indent=1)


# ---- markdown tables generated from the files above: profiles/README.md and DESIGN.md quote THESE, nothing is typed by hand ------
def md_tables():
    L = ["<!-- generated by scripts/collect_profiles.py %s from the files named in each heading; do not edit -->" % tag, ""]
    for name, what in (("_pytest_gpu.log", "`python -m pytest tests -q -m gpu` on the GPU box"),
                       ("_dropin_gpu.log", "`SHAPEGAN_REFERENCE_DIR=<scratch copy of the five reference scripts> python -m pytest "
                                           "tests/test_dropin.py -v` on the GPU box")):
        f = os.path.join(dst, tag + name)
        if os.path.exists(f):
            tail = [ln.strip() for ln in open(f).read().splitlines() if " passed" in ln or " failed" in ln or " error" in ln]
            L += ["### `%s%s` (%s)" % (tag, name, what), "", tail[-1].strip("= ") if tail else "(no summary line)", ""]
    bl = os.path.join(dst, tag + "_bench_line.json")
    if os.path.exists(bl):
        d = json.loads([ln for ln in open(bl).read().splitlines() if ln.startswith("{")][-1])
        L += ["### `%s_bench_line.json` (`python bench.py`, N = 1)" % tag, "",
              "| quantity | value |", "|---|---|",
              "| %s | **%.2f %s** (%.3f ms per step) |" % (d["metric"], d["value"], d["unit"], d["ms_per_step"]),
              "| roofline kernel | %s: %.1f %s of %.1f = **%.3f** |" % (d["roofline"]["kernel"].split(" (")[0], d["roofline"]["achieved"],
                                                                       d["roofline"]["unit"], d["roofline"]["peak"], d["roofline"]["frac"])]
        s_ = d["sdfnet"]
        L += ["| SDFNet fused forward, 8 x 32^3, no grad | %.1f Mpoints/s = %.3f of the fp32 MFMA peak (executed FLOPs) |"
              % (s_["fwd_mpoints_per_s"], s_["fwd_frac_of_f32_mfma_peak_executed"])]
        for key, what in (("train_ref_20k_L128", "auto-decoder step, 20 000 points, L = 128, one captured graph"),
                          ("train_ref_20k_L128_eager", "the same launched eagerly"),
                          ("train_cfg_200k_L256", "auto-decoder step, 200 000 points, L = 256 (configs[2])")):
            L += ["| %s | %.3f ms = %.2f Mpoints/s = %.3f executed |" % (what, s_[key]["ms_per_step"], s_[key]["mpoints_per_s"],
                                                                       s_[key]["frac_of_f32_mfma_peak_executed"])]
        for key, v in d.get("other_configs", {}).items():
            if key == "point_gan" and "error" in v:
                L += ["| other_configs.point_gan | failed: %s |" % v["error"]]
                continue
            if key == "point_gan":
                for upd in ("critic_update", "generator_update"):
                    u = v[upd]
                    L += ["| other_configs.point_gan %s (12 x 16 384 points) | %.3f ms = %.2f Mpoints/s; %.3f of the fp32 MFMA peak in the "
                          "reference's dense arithmetic, %.3f in the arithmetic executed |"
                          % (upd.replace("_", " "), u["ms"], u["mpoints_per_s"], u["reference_arithmetic_frac_of_f32_mfma_peak"],
                             u["executed_arithmetic_frac_of_f32_mfma_peak"])]
                continue
            L += ["| other_configs.%s | %.4g %s (%.3f ms per step) |" % (key, v["value"], v["unit"], v["ms_per_step"])]
        cb = d.get("cpu_baseline")
        if cb:
            L += ["| cpu_baseline | %.3f %s on %d of %d host threads, kind `%s` |" % (cb["value"], cb["unit"], cb["cores"],
                                                                                    cb.get("host_cores", 0), cb["kind"])]
        L += ["", "| conv form (`kernels`) | kernel | us | rate | fraction of its roofline | us warm (replay on one buffer set) | us inside the profiled step |",
              "|---|---|---|---|---|---|---|"]
        for k in d["kernels"]:
            rate = "%.1f TFLOP/s" % k["tflops"] if k["bound"] == "mfma" else "%.0f GB/s" % k["gb_per_s"]
            L += ["| %s | `%s` | %.1f%s | %s | %.3f (%s) | %s | %s |" % (
                k["name"], k["kernel"], k["us"], " (cold)" if k.get("timing", "").startswith("cold") else "", rate, k["frac"],
                "fp32 MFMA 157.3 TF" if k["bound"] == "mfma" else "HBM 8 TB/s", k.get("us_warm", ""), k.get("us_in_step", ""))]
        L += [""]
        dl = d.get("dropin_loop")
        if dl:
            L += ["`dropin_loop` (%s):" % dl["what"], "",
                  "| loop | steps/s | ms per step | fraction of `WGANTrainer.step` | kernel launches per step |", "|---|---|---|---|---|",
                  "| `WGANTrainer.step` (the headline) | %.2f | %.3f | 1 | %s |" % (d["value"], d["ms_per_step"], dl.get("trainer_step_launches_per_step")),
                  "| module-level loop, batches resident on the device | %.2f | %.3f | %.3f | %s |" % (
                      dl["resident_batches"]["steps_per_s"], dl["resident_batches"]["ms_per_step"],
                      dl["resident_batches"].get("fraction_of_trainer_step", 0), dl.get("launches_per_step")),
                  "| module-level loop, pageable host batches copied in the loop (`batch.to(device)`) | %.2f | %.3f | %.3f | |" % (
                      dl["host_batches_copied_in_the_loop"]["steps_per_s"], dl["host_batches_copied_in_the_loop"]["ms_per_step"],
                      dl["host_batches_copied_in_the_loop"].get("fraction_of_trainer_step", 0)), ""]
    b2 = os.path.join(dst, tag + "_bench_line_2ranks_gloo_one_gpu.json")
    if os.path.exists(b2):
        lines = [ln for ln in open(b2).read().splitlines() if ln.startswith("{")]
        if lines:
            d = json.loads(lines[-1])
            L += ["### `%s_bench_line_2ranks_gloo_one_gpu.json` (the driver's multi-GPU command with two ranks on ONE GPU, gloo)" % tag, "",
                  "n_gpus %d, parallelism %s, %.2f %s aggregate (both ranks share one GPU: a rehearsal of the code path, not a scaling "
                  "number); comm: %s" % (d["n_gpus"], d["config"]["parallelism"], d["value"], d["unit"], json.dumps(d.get("comm"))), ""]
    tl = os.path.join(dst, tag + "_wgan_step_timeline.txt")
    if os.path.exists(tl):
        import re
        rows_ = [(float(m.group(1)), m.group(2).strip()) for m in
                 (re.match(r"\s*[\d.]+\s+\+\s+[-\d.]+ gap\s+([\d.]+) us\s+(.*)", ln) for ln in open(tl)) if m]
        halo = sum(d_ for d_, n in rows_ if "halo_kernel" in n or "halo4_kernel" in n)
        tot = sum(d_ for d_, n in rows_)
        agg = collections.defaultdict(lambda: [0, 0.0])
        for d_, n in rows_:
            agg[n[:70]][0] += 1
            agg[n[:70]][1] += d_
        L += ["### `%s_wgan_step_timeline.txt` (one 5+1 WGAN step under `rocprofv3 --kernel-trace`)" % tag, "",
              "%d launches, %.2f ms of kernel time: LDS-halo MFMA convolutions %.2f ms, everything else (\"tail\") %.2f ms" %
              (len(rows_), tot / 1e3, halo / 1e3, (tot - halo) / 1e3), "", "| kernel | launches | us per step |", "|---|---|---|"]
        for n, (c_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
            L += ["| `%s` | %d | %.1f |" % (n, c_, t_)]
        L += [""]
    for name, title in (("_mfma_utilisation.json", "MFMA utilisation (`SQ_VALU_MFMA_BUSY_CYCLES` / (`GRBM_GUI_ACTIVE` per XCD x 1024 SIMDs), median launch)"),
                        ("_configs_mfma_utilisation.json", "the same for the kernels of configs[2] / [3] / [4]")):
        f = os.path.join(dst, tag + name)
        if os.path.exists(f):
            L += ["### `%s%s`: %s" % (tag, name, title), "", "| kernel | launches | dispatch us | MFMA busy |", "|---|---|---|---|"]
            for r in json.load(open(f)):
                L += ["| `%s` | %d | %.1f | %.3f |" % (r["kernel"][:90], r["launches"], r["dispatch_us"], r["mfma_util"])]
            L += [""]
    for name, title in (("_hbm_traffic.json", "HBM bytes per launch (`FETCH_SIZE` x 2 + `WRITE_SIZE`, separate passes, median launch)"),
                        ("_configs_hbm_traffic.json", "the same for the kernels of configs[2] / [3] / [4]")):
        f = os.path.join(dst, tag + name)
        if os.path.exists(f):
            L += ["### `%s%s`: %s" % (tag, name, title), "", "| kernel | launches | dispatch us | read MB | written MB | GB/s |", "|---|---|---|---|---|---|"]
            for r in json.load(open(f))[:22]:
                L += ["| `%s` | %d | %.1f | %.1f | %.1f | %.0f |" % (r["kernel"][:90], r["launches"], r["dispatch_us"], r["hbm_read_bytes"] / 1e6,
                                                                  r["hbm_write_bytes"] / 1e6, r["hbm_gb_per_s"])]
            L += [""]
    f = os.path.join(dst, tag + "_stream_calibration.json")
    if os.path.exists(f):
        L += ["### `%s_stream_calibration.json` (`python scripts/stream_calibration.py`)" % tag, "", "| pass | kernel | us | TB/s |", "|---|---|---|---|"]
        for r in json.load(open(f))["passes"]:
            L += ["| %s | %s | %.1f | %.2f |" % (r["pass"], r["kernel"], r["us"], r["tb_per_s"])]
        L += [""]
    f = os.path.join(dst, tag + "_edge_kernels_by_batch.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_edge_kernels_by_batch.json` (`python scripts/edge_ab.py`: one-channel kernels, us per launch)" % tag, "",
              "| entry | " + " | ".join(k for k in d if k != "lib") + " |", "|---|" + "---|" * (len(d) - 1),
              "| us | " + " | ".join(str(v) for k, v in d.items() if k != "lib") + " |", ""]
    f = os.path.join(dst, tag + "_edge_kernels_cold.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_edge_kernels_cold.json` (`python scripts/edge_cold.py all`: every call on the next of K operand sets, > 640 MB in rotation)" % tag, "",
              "| kernel @ samples | cold us | warm us | cold fraction of 8 TB/s |", "|---|---|---|---|"]
        for k, v in d.items():
            if isinstance(v, dict):
                L += ["| %s | %.1f | %.1f | %.3f |" % (k, v["cold_us"], v["warm_us"], v["cold_frac_8TBs"])]
        L += [""]
    f = os.path.join(dst, tag + "_fwd_c1_ablation.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_fwd_c1_ablation.json` (Conv3d(1 -> 64) forward, 32^3 -> 16^3: what each part of the kernel costs, cold us)" % tag, "",
              "| build | 128 samples | 64 samples | 16 samples |", "|---|---|---|---|"]
        for k, v in d.items():
            L += ["| %s | %.1f | %.1f | %.1f |" % (k, v["fwd_128"]["cold_us"], v["fwd_64"]["cold_us"], v["fwd_16"]["cold_us"])]
        L += [""]
    f = os.path.join(dst, tag + "_convT_c1_ablation.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_convT_c1_ablation.json` (ConvTranspose3d(64 -> 1) forward, 16^3 -> 32^3: `convT_c1_stream_kernel` builds, cold us; below 48 "
              "samples the per-plane kernel serves the call and the builds do not differ)" % tag, "",
              "| build | 256 samples | 64 samples | 32 samples | 16 samples |", "|---|---|---|---|---|"]
        for k, v in d.items():
            L += ["| %s | %.1f | %.1f | %.1f | %.1f |" % (k, v["convT_256"]["cold_us"], v["convT_64"]["cold_us"], v["convT_32"]["cold_us"],
                                                       v["convT_16"]["cold_us"])]
        L += [""]
    f = os.path.join(dst, tag + "_write_pattern.jsonl")
    if os.path.exists(f):
        rows_ = [json.loads(ln) for ln in open(f) if ln.startswith("{")]
        names = {0: "the forward's pattern: a wave writes 128 B to each of 64 rows 16 KB apart", 1: "512 B runs, a wave per 64 rows",
                 2: "512 B runs, a wave per 16 rows", 3: "plain fill (1 KB per wave instruction)"}
        L += ["### `%s_write_pattern.jsonl` (`scripts/micro/write_pattern`: 134 MB written with nothing else going on)" % tag, "",
              "| workgroups | pattern | cold us | TB/s | warm us |", "|---|---|---|---|---|"]
        cold = {(r["grid"], r["mode"]): r for r in rows_ if "us" in r}
        warm = {(r["grid"], r["mode"]): r for r in rows_ if "warm_us" in r}
        for (g_, m_), r in sorted(cold.items()):
            if r["us"] > 10:
                L += ["| %d | %s | %.1f | %.2f | %.1f |" % (g_, names.get(m_, m_), r["us"], r["tb_per_s"], warm.get((g_, m_), {}).get("warm_us", 0))]
        L += [""]
    f = os.path.join(dst, tag + "_dgrad_paired_stores.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_dgrad_paired_stores.json` (`conv_dgrad_halo_kernel`: 8-byte (pw0, pw1) stores against the 4-byte stores of rounds 1-4)" % tag, "",
              "| kernel, grid | build | written MB | fetched MB (x2) | MFMA busy |", "|---|---|---|---|---|"]
        for v in ("paired", "unpaired"):
            for k, r in d[v].items():
                if k != "time" and "written_MB" in r:
                    L += ["| %s | %s | %.1f | %.1f | %s |" % (k, v, r["written_MB"], r["fetched_MB_x2"], r.get("mfma_busy", ""))]
        L += ["", "| shape | paired us | unpaired us |", "|---|---|---|"]
        for k in d["paired"]["time"]:
            L += ["| %s | %.1f | %.1f |" % (k, d["paired"]["time"][k]["us"], d["unpaired"]["time"].get(k, {}).get("us", 0))]
        L += ["", d.get("note", ""), ""]
    f = os.path.join(dst, tag + "_cpu_baseline_reference_vs_port.json")
    if os.path.exists(f):
        d = json.load(open(f))
        L += ["### `%s_cpu_baseline_reference_vs_port.json` (`python scripts/cpu_baseline_compare.py`, authoring container, no GPU)" % tag, "",
              "reference modules %.4f steps/s, port (oracle/torch_oracle.py) %.4f steps/s on the same %d threads: ratio %.3f; first critic "
              "loss bit-equal: %s" % (d["reference"]["steps_per_s"], d["port"]["steps_per_s"], d["cores"], d["port_over_reference"],
                                      d["first_critic_loss_equal"]), ""]
    open(os.path.join(dst, tag + "_tables.md"), "w").write("\n".join(L) + "\n")


md_tables()
for r in out:
    print("%-60s util %.3f  %.0f us" % (r["kernel"][:60], r["mfma_util"], r["dispatch_us"]))
for r in rows[:14]:
    print("%-60s R %.1f MB W %.1f MB  %.0f GB/s" % (r["kernel"][:60], r["hbm_read_bytes"] / 1e6, r["hbm_write_bytes"] / 1e6, r["hbm_gb_per_s"]))
