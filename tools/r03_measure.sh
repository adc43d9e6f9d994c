#!/bin/bash
# Round-3 measurement set on the GPU box (run through gpurun from the repo root): the driver's bench command un-profiled and
# under rocprofv3 --kernel-trace --stats (every BASELINE single-GPU config is a leg of that one line), PMC traffic passes
# (FETCH_SIZE and WRITE_SIZE in separate runs: 3 + 2 of the 4 TCC slots), and the stand-alone lines of the other
# formats.  Output under gpurun_out/r03_final/; tools/summarize_round.py condenses it into profiles/r03_z_*.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${ROUND:-r03}_final
mkdir -p $O
step() { echo "== $*"; }

step "bench, driver command"
timeout -k 10 400 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_bc1.json 2> $O/bench_bc1.err || exit 1
step "kernel trace"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 $R/bench.py --steps 20 --warmup 5 --host-array-gib 0 --no-cpu-baseline > $O/prof_kt.json 2> $O/prof_kt.err || exit 1
for pass in "prof_fetch FETCH_SIZE" "prof_write WRITE_SIZE"; do
  set -- $pass; d=$1; shift
  step "pmc $*"
  timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$d -- python3 $R/bench.py --steps 3 --warmup 1 --leg-steps 2 --no-cpu-baseline --host-array-gib 0 > $O/$d.json 2> $O/$d.err || exit 1
done
step "stand-alone lines"
timeout -k 10 300 python3 $R/bench.py --format bc2 > $O/bench_bc2.json 2> $O/bench_bc2.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc3 > $O/bench_bc3.json 2> $O/bench_bc3.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc7 > $O/bench_bc7.json 2> $O/bench_bc7.err || exit 1
timeout -k 10 300 python3 $R/bench.py --workload archive > $O/bench_archive.json 2> $O/bench_archive.err || exit 1
step "N = 2 rehearsals on one GPU (gloo): self-launch, weak and strong"
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --size-gib 4 --steps 10 > $O/bench_gpus2_weak_gloo.json 2> $O/bench_gpus2_weak.err || exit 1
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --scaling strong --steps 10 > $O/bench_gpus2_strong_gloo.json 2> $O/bench_gpus2_strong.err || exit 1
step done
