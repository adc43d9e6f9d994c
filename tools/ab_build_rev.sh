#!/bin/bash
# Same-box A/B of whole library versions: builds the library as of git revision REV (all sources) into ab/libdxtlt_NAME.so,
# to be run beside the current one through DXTLT_LIB_PATH on ONE gpurun box (box-to-box spread is +-0.02 of peak).
#     tools/ab_build_rev.sh HEAD~3 r3
#     DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_EXPERIMENTS tools/ab_build_rev.sh WORKTREE exp     (the experiments side build)
#     gpurun -- 'for lib in ab/libdxtlt_r3.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
#                  DXTLT_LIB_PATH=$GRAFT_REPO_ROOT/$lib python tools/batch_shape_probe.py; done'
# ab/ is git-ignored; remove it afterwards (it travels with every gpurun push).
set -eu
REV=${1:?git revision}; NAME=${2:?name}
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/ab_rev_$NAME
rm -rf $W && mkdir -p $W $R/ab
if [ "$REV" = WORKTREE ]; then      # the working tree as it is (with DXTLT_EXTRA_HIPCC_FLAGS: a compiler-flag experiment)
  mkdir -p $W/dxt-lossless-transform_amd && cp -r $R/dxt-lossless-transform_amd/csrc $R/dxt-lossless-transform_amd/_build.py $W/dxt-lossless-transform_amd/ && cp -r $R/include $W/
else
  git -C $R archive $REV dxt-lossless-transform_amd include | tar -x -C $W
fi
BUILT=$(python3 - <<PY
import importlib.util, sys
spec = importlib.util.spec_from_file_location("_build", "$W/dxt-lossless-transform_amd/_build.py")
m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import os
extra = os.environ.get("DXTLT_EXTRA_HIPCC_FLAGS", "").split()   # honoured HERE only (the package's own build() never reads it)
print(m.build(force=True, extra_flags=extra))     # the shipped path of the copy under /tmp, or build/side-<hash>/ with extra flags
PY
)
BUILT=$(echo "$BUILT" | tail -1)
cp "$BUILT" $R/ab/libdxtlt_$NAME.so
rm -rf $W
echo "built $R/ab/libdxtlt_$NAME.so (as of $REV)"
