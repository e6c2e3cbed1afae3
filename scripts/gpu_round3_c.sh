for i in 1 2; do
python scripts/sdf_ab.py 2>&1 | tail -1
for t in a b c nomask; do SHAPEGAN_HIP_LIB=$PWD/scripts/_abl/lib_$t.so python scripts/sdf_ab.py 2>&1 | tail -1; done
done
