#!/bin/bash
# Same-box A/B of two library versions: builds csrc/<file> as of git revision REV into ab/libdxtlt_old.so (every other
# object taken from the current build), to be run beside the current library through DXTLT_LIB_PATH on ONE gpurun box --
# box-to-box spread (+-0.02 of peak) is larger than most kernel changes.  ab/ is git-ignored; remove it afterwards (it
# travels with every gpurun push).
#     tools/ab_build.sh HEAD~1 bcn_kernels.hip
#     gpurun -- 'for lib in ab/libdxtlt_old.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
#                  DXTLT_LIB_PATH=$GRAFT_REPO_ROOT/$lib python tools/shift_probe.py; done'
set -eu
REV=${1:?git revision}; FILE=${2:-bcn_kernels.hip}
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/dxt-lossless-transform_amd/csrc
mkdir -p $R/ab
python3 -c "import sys; sys.path.insert(0, '$R'); import dxt_lossless_transform_amd as p; p.build(force=False)"
git -C $R show $REV:dxt-lossless-transform_amd/csrc/$FILE > $C/_ab_old_$FILE
trap 'rm -f $C/_ab_old_$FILE' EXIT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-command-line-argument -x hip -c $C/_ab_old_$FILE -o $R/ab/old.o
objs=$(ls $R/build/obj/*.o | grep -v "/$FILE.o")
hipcc --offload-arch=gfx950 -shared -fPIC $objs $R/ab/old.o -o $R/ab/libdxtlt_old.so -lpthread
echo "built $R/ab/libdxtlt_old.so ($FILE as of $REV)"
