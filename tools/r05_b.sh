#!/bin/bash
# Round 5, second box: GPU suite, the placement probe on the BC3 corpus, BC1 no-split (timing + PMC passes), A/B of side libraries.
set -u
OUT=gpurun_out/r05_b; mkdir -p $OUT gpurun_out/r05_vmm
if [ "${SKIP_TESTS:-0}" != 1 ]; then
  (timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/tests.txt 2>&1; rc=$?; echo "pytest rc $rc" >> $OUT/tests.txt; [ $rc -le 1 ]) || exit 3
fi
# (the placement probe's corpus pass ran here once: its first VMM arm ended in a GPU memory access fault inside a freshly mapped range --
#  profiles/r05_placement_vmm.txt item 5 -- and no VMM mapping has been made on the pool since)
timeout -k 10 300 python tools/bc1_nosplit_probe.py time > $OUT/nosplit_time.txt 2>&1 || exit 5
for sc in 1 0; do
  PMC_PASS_TIMEOUT=200 timeout -k 10 900 python3 tools/pmc_passes.py $OUT/pmc_nosplit_sc$sc tools/nosplit_pmc_sets.txt -- python3 tools/bc1_nosplit_probe.py pmc $sc > $OUT/pmc_nosplit_sc$sc.txt 2>&1 || exit 6
done
bash tools/r05_ab.sh $OUT wave0 halo32 new
