#!/bin/bash
# Round 5, third box: GPU suite on the 128-lane BC1 shift tiles, BC1 timing, PMC of the aligned 256-lane shape, A/B against 256-lane BC1 tiles.
set -u
OUT=gpurun_out/r05_c; mkdir -p $OUT
(timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?; echo "pytest rc $rc" >> $OUT/tests.txt; [ $rc -le 1 ]) || exit 3
grep -q "pytest rc 0" $OUT/tests.txt || exit 3
timeout -k 10 300 python tools/bc1_nosplit_probe.py time > $OUT/nosplit_time.txt 2>&1 || exit 5
for sc in 1 0; do
  PMC_PASS_TIMEOUT=200 timeout -k 10 600 python3 tools/pmc_passes.py $OUT/pmc256_sc$sc tools/nosplit256_pmc_sets.txt -- python3 tools/bc1_nosplit_probe.py pmc256 $sc > $OUT/pmc256_sc$sc.txt 2>&1 || exit 6
done
SHIFT_PROBE=1 bash tools/r05_ab.sh $OUT bc1x256 new
