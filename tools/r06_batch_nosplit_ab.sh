#!/bin/bash
# Same-box A/B of the batch kernel's lane count for BC1 without the colour split, forward: the shipped library
# (128 lanes, batch_tile_threads) against a side build with the round-5 shape (256 lanes).  Build the side library FIRST, here:
#     DXTLT_EXTRA_HIPCC_FLAGS=-DDXTLT_BATCH_BC1_NOSPLIT_FWD_THREADS=256 tools/ab_build_rev.sh WORKTREE nosplit256
# then:  gpurun --timeout 900 -- 'bash tools/r06_batch_nosplit_ab.sh > gpurun_out/r06_batch_nosplit_ab.txt 2>&1'
set -eu
cd "$(dirname "$0")/.."
for pass in 1 2; do
  for lib in ab/libdxtlt_nosplit256.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
    echo "=== pass $pass: $lib"
    DXTLT_LIB_PATH=$PWD/$lib timeout -k 10 300 python3 tools/batch_nosplit_probe.py
  done
done
