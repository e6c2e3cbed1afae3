mkdir -p gpurun_out/r03
L=gpurun_out/r03/c.log; rm -f $L
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_modules.py -q --tb=short -x -k "conv or discriminator or wgan or autoencoder or progressive" 2>&1 | tail -8 >> $L
python scripts/edge_ab.py >> $L 2>&1
SG_EDGE_DEBUG=1 python scripts/edge_ab.py >> $L 2>&1
cat $L | grep -v amdgpu.ids
