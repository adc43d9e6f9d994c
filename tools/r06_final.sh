#!/bin/bash
# Round 6, the measurement set on the final source (tools/r03_measure.sh, ROUND=r06), the N = 4 rehearsals on the one GPU (gloo carries
# the barrier and the reductions; four ranks share the device) -- now with every rank's own exactness flags in per_rank -- and a fuzz
# campaign over the changed batch path.   gpurun --timeout 1200 -- 'bash tools/r06_final.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
ROUND=r06 bash $R/tools/r03_measure.sh > $R/gpurun_out/r06_final_measure.log 2>&1 || exit 3
O=$R/gpurun_out/r06_final
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --size-gib 2 --steps 10 > $O/bench_gpus4_weak_gloo.json 2> $O/bench_gpus4_weak.err || exit 4
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --scaling strong --size-gib 8 --steps 10 > $O/bench_gpus4_strong_gloo.json 2> $O/bench_gpus4_strong.err || exit 5
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --workload archive --size-gib 2 --steps 5 > $O/bench_gpus4_archive_gloo.json 2> $O/bench_gpus4_archive.err || exit 6
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 4 --format bc7 --size-gib 1 --steps 5 > $O/bench_gpus4_bc7_gloo.json 2> $O/bench_gpus4_bc7.err || exit 7
timeout -k 10 200 python3 $R/tools/fuzz_gpu.py --seconds 120 --seed 606 > $O/fuzz_all.txt 2>&1 || exit 8
timeout -k 10 200 python3 $R/tools/fuzz_gpu.py --seconds 120 --seed 607 --batch-only > $O/fuzz_batch.txt 2>&1 || exit 9
echo done
