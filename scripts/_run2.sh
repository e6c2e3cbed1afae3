bash scripts/kernel_pmc.sh sdfnet_bwd python scripts/prof_targets.py sdfstep > gpurun_out/pmc_new.txt 2>&1
SHAPEGAN_HIP_LIB=$PWD/scripts/_abl/v1.so bash scripts/kernel_pmc.sh sdfnet_bwd python scripts/prof_targets.py sdfstep > gpurun_out/pmc_v1.txt 2>&1
tail -3 gpurun_out/kernel_pmc/tcp.log gpurun_out/kernel_pmc/mfma.log
