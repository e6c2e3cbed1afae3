mkdir -p gpurun_out/r03
for v in 0 1 2 3; do
  echo "=== variant $v" >> gpurun_out/r03/debug_var.log
  if [ $v = 0 ]; then unset SHAPEGAN_HIP_LIB; else export SHAPEGAN_HIP_LIB=$PWD/scripts/_abl/lib_v$v.so; fi
  timeout 300 python scripts/debug_sdf_mask.py 2>&1 | grep -v "H==0" | head -12 >> gpurun_out/r03/debug_var.log
done
cat gpurun_out/r03/debug_var.log
