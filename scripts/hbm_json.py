"""Builds profiles/r01_conv_fwd_halo_hbm.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over halo_pmc.py."""
import csv, json, sys
def median_kb(path, counter):
    v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(path))
               if "conv_fwd_halo_kernel" in r["Kernel_Name"] and r["Counter_Name"] == counter)
    name = [r["Kernel_Name"] for r in csv.DictReader(open(path)) if "conv_fwd_halo_kernel" in r["Kernel_Name"]][0]
    return v[len(v) // 2], name
fetch, name = median_kb(sys.argv[1], "FETCH_SIZE")
write, _ = median_kb(sys.argv[2], "WRITE_SIZE")
batch = int(sys.argv[3])
alg = batch * (64 * 16 ** 3 + 128 * 8 ** 3) * 4 + 128 * 64 * 64 * 4
out = {"kernel": name, "config": "Conv3d 64->128 k4 s2 p1 forward, x [%d,64,16,16,16] fp32" % batch,
       "FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write,
       "correction": "MI355X_MICROARCH.md HBM section: gfx950 FETCH_SIZE counts 128-B requests as 64 B -> x2; WRITE_SIZE as reported; KB = 1024 B",
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0, "algorithmic_bytes_per_launch": alg,
       "note": "separate --pmc passes (FETCH_SIZE, WRITE_SIZE), 5 dispatches each, scripts/halo_pmc.py; algorithmic = x + y + W once"}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(out)
