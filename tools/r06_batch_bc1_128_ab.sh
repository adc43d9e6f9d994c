#!/bin/bash
# Same-box A/B: 128 lanes for EVERY BC1 halo / shifted / edge / batch tile against the shipped 256, split setting, by shape and by size
# class of the corpus (is there a size below which the smaller tile wins?  No: profiles/r06_batch_bc1_nosplit.txt, item 2).  Build first:
#     DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_BC1_SHIFT_THREADS=128 tools/ab_build_rev.sh WORKTREE bc1t128
set -eu
for pass in 1 2; do
  for lib in ab/libdxtlt_bc1t128.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
    echo "=== pass $pass: $lib"
    DXTLT_LIB_PATH=$PWD/$lib timeout -k 10 400 python3 tools/batch_nosplit_probe.py --settings "1,1" --more
  done
done
