set -eu
for pass in 1 2; do
  for lib in ab/libdxtlt_bc1t128.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
    echo "=== pass $pass: $lib"
    DXTLT_LIB_PATH=$PWD/$lib timeout -k 10 400 python3 tools/batch_nosplit_probe.py --settings "1,1" --more
  done
done
