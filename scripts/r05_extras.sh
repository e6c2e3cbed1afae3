#!/bin/bash
# Round-5 evidence that scripts/profile_round.sh does not produce (run on the GPU box by scripts/round_evidence.sh):
#   <tag>_write_pattern.jsonl             what the memory side takes when 134 MB are written the way conv_fwd_c1 writes them (cold / warm)
#   <tag>_edge_kernels_cold.json          the four one-channel kernels timed cold AND warm (scripts/edge_cold.py)
#   <tag>_fwd_c1_ablation.json            conv_fwd_c1_lds_kernel without stores / MFMAs / staging loads, and the gather form it replaced
#   <tag>_convT_c1_ablation.json         convT_c1_stream_kernel without plane loads / MFMAs / global stores (cold, 256 .. 16 samples)
#   <tag>_dgrad_paired_stores.json        conv_dgrad_halo_kernel with 8-byte (pw0, pw1) stores against the 4-byte stores of rounds 1-4:
#                                         time, WRITE_SIZE, FETCH_SIZE, MFMA busy per shape
tag=${1:-rXX}
repo=$(pwd); out=$repo/gpurun_out/prof_$tag; mkdir -p $out
[ -x scripts/micro/write_pattern ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/micro/write_pattern.hip -o scripts/micro/write_pattern 2>/dev/null
scripts/micro/write_pattern > $out/${tag}_write_pattern.jsonl 2>/dev/null
[ -f scripts/_abl/nopair.so ] || bash scripts/ab_build.sh nopair conv3d_halo.hip -DSG_DGRAD_NO_PAIR > /dev/null 2>&1
for v in 1 2 4 3; do [ -f scripts/_abl/fwdc1_abl$v.so ] || bash scripts/ab_build.sh fwdc1_abl$v conv3d_edge.hip -DSG_FWDC1_ABL=$v > /dev/null 2>&1; done
for v in 1 2 4 3; do [ -f scripts/_abl/convt_abl$v.so ] || bash scripts/ab_build.sh convt_abl$v conv3d_edge.hip -DSG_CONVT_ABL=$v > /dev/null 2>&1; done
python scripts/edge_cold.py all > $out/${tag}_edge_kernels_cold.json 2> $out/edge_cold.err
{
  echo '{'
  echo '"shipped": '; python scripts/edge_cold.py convT 2>/dev/null; echo ','
  echo '"no_plane_loads": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/convt_abl1.so python scripts/edge_cold.py convT 2>/dev/null; echo ','
  echo '"no_mfma": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/convt_abl2.so python scripts/edge_cold.py convT 2>/dev/null; echo ','
  echo '"no_global_stores": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/convt_abl4.so python scripts/edge_cold.py convT 2>/dev/null; echo ','
  echo '"no_plane_loads_no_mfma": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/convt_abl3.so python scripts/edge_cold.py convT 2>/dev/null
  echo '}'
} > $out/${tag}_convT_c1_ablation.json
{
  echo '{'
  echo '"lds": '; python scripts/edge_cold.py fwd 2>/dev/null; echo ','
  echo '"gather_form_of_rounds_3_4": '; SG_FWD_C1_LDS=0 python scripts/edge_cold.py fwd 2>/dev/null; echo ','
  echo '"lds_no_global_stores": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/fwdc1_abl1.so python scripts/edge_cold.py fwd 2>/dev/null; echo ','
  echo '"lds_no_mfma": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/fwdc1_abl2.so python scripts/edge_cold.py fwd 2>/dev/null; echo ','
  echo '"lds_no_staging_loads": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/fwdc1_abl4.so python scripts/edge_cold.py fwd 2>/dev/null; echo ','
  echo '"lds_no_stores_no_mfma": '; SHAPEGAN_HIP_LIB=$repo/scripts/_abl/fwdc1_abl3.so python scripts/edge_cold.py fwd 2>/dev/null
  echo '}'
} > $out/${tag}_fwd_c1_ablation.json
for v in paired unpaired; do
  lib=$repo/shapegan_amd/libshapegan_hip.so; [ $v = unpaired ] && lib=$repo/scripts/_abl/nopair.so
  SHAPEGAN_HIP_LIB=$lib python scripts/dgrad_target.py time > $out/dgrad_time_$v.json 2>/dev/null
  ( cd /tmp && export TMPDIR=/tmp
    for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
      n=${c%% *}
      SHAPEGAN_HIP_LIB=$lib rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/dg/$v$n -o x -- python $repo/scripts/dgrad_target.py > $out/pmc_$v$n.log 2>&1
      f=$(find /tmp/dg/$v$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && cp $f $out/dgrad_${v}_${n}.csv
    done )
done
python - $out $tag <<'PY'
import collections, csv, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {}
for v in ("paired", "unpaired"):
    rec = {"time": json.load(open(os.path.join(out, "dgrad_time_%s.json" % v)))}
    for f in sorted(glob.glob(os.path.join(out, "dgrad_%s_*.csv" % v))):
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(f)):
            if "conv_dgrad_halo" not in r["Kernel_Name"]:
                continue
            key = ("conv_dgrad_halo_kernel<%s>" % r["Kernel_Name"].split("<")[1][0], int(r["Grid_Size"]), r["Counter_Name"])
            agg.setdefault(key, []).append(float(r["Counter_Value"]))
        for (k, grid, c), vals in agg.items():
            vals.sort()
            rec.setdefault("%s grid %d" % (k, grid), {})[c] = vals[len(vals) // 2]
    for k, d in rec.items():
        if "WRITE_SIZE" in d:
            d["written_MB"], d["fetched_MB_x2"] = round(d["WRITE_SIZE"] * 1024 / 1e6, 1), round(d.get("FETCH_SIZE", 0) * 2048 / 1e6, 1)
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and d.get("GRBM_GUI_ACTIVE"):
            d["mfma_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
    res[v] = rec
res["note"] = ("grids: <0> 131072 threads = 64 AND 128 samples of [N,128,8^3] -> [N,64,16^3] (512 workgroups either way, 4 or 8 parities "
               "each; the median launch is a 128-sample one: 134 MB of output), 262144 = 256 samples; <1> 32768 / 65536 / 131072 = 16-32 / 64 / "
               "128-256 samples of [N,256,4^3] -> [N,128,8^3].  paired = this round's library, unpaired = the same source built with "
               "-DSG_DGRAD_NO_PAIR (the 4-byte stores of rounds 1-4)")
json.dump(res, open(os.path.join(out, tag + "_dgrad_paired_stores.json"), "w"), indent=1)
PY
ls -la $out | grep -E "write_pattern|edge_kernels_cold|fwd_c1_ablation|convT_c1_ablation|dgrad_paired" 
