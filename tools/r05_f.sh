#!/bin/bash
# 32-byte halo windows for BC1 only?  single-call cases of tools/shift_probe.py, BC1, halo32 side build against the tree, ABAB
set -u
OUT=gpurun_out/r05_f; mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do
  for lib in ab/libdxtlt_halo32.so dxt-lossless-transform_amd/libdxtlt_gfx950.so; do
    echo "== round $round $lib" >> $OUT/shift_bc1.txt
    for only in bc1_odd bc1_plus; do
      DXTLT_LIB_PATH=$R/$lib PROBE_FMT=bc1 PROBE_ONLY=${only#bc1_} timeout -k 10 300 python tools/shift_probe.py 2>&1 | grep "^bc1" | grep -v "WRONG\|forced\|generic\|remap\|policy\|shared\|nt_\|first_form\|identity\|contiguous\|16_only" >> $OUT/shift_bc1.txt
    done
  done
done
