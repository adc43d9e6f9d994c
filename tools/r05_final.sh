#!/bin/bash
# Round 5, last box: the measurement set on the final source (tools/r03_measure.sh, ROUND=r05) and N = 4 rehearsals on the one GPU
# (gloo carries the barrier and the reductions; four ranks share the device): the driver's N > 1 code path with more than two ranks.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=r05 bash $R/tools/r03_measure.sh > $R/gpurun_out/r05_final_measure.log 2>&1 || exit 3
O=$R/gpurun_out/r05_final
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --size-gib 2 --steps 10 > $O/bench_gpus4_weak_gloo.json 2> $O/bench_gpus4_weak.err || exit 4
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --scaling strong --size-gib 8 --steps 10 > $O/bench_gpus4_strong_gloo.json 2> $O/bench_gpus4_strong.err || exit 5
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --workload archive --size-gib 2 --steps 5 > $O/bench_gpus4_archive_gloo.json 2> $O/bench_gpus4_archive.err || exit 6
echo done
