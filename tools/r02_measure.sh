#!/bin/bash
# Per-round measurement set on the GPU box (ROUND=r03 names the output directory; default r02) (run through gpurun from the repo root): bench lines, rocprofv3 kernel stats
# and PMC traffic passes for the BC1 headline workload and the BC7 workload.  Output under gpurun_out/r02_final/;
# tools/summarize_profiles.py condenses it into profiles/.
set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/${ROUND:-r02}_final
mkdir -p $O
step() { echo "== $*"; }

for fmt in bc1 bc7; do
  D=$O/$fmt; mkdir -p $D
  extra=""; [ $fmt = bc7 ] && extra="--format bc7"
  step "kernel trace $fmt"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $D/prof_kt -- python3 $R/bench.py --steps 20 --warmup 3 --host-array-gib 0 $extra > $D/prof_kt.log 2> $D/prof_kt.err || exit 1
  for pass in "prof_fetch FETCH_SIZE" "prof_write WRITE_SIZE" "prof_req TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_32B_sum"; do
    set -- $pass; d=$1; shift
    step "pmc $fmt $*"
    timeout -k 10 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $D/$d -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --host-array-gib 0 $extra > $D/$d.log 2> $D/$d.err || exit 1
  done
done
step "bench lines"
timeout -k 10 400 python3 $R/bench.py > $O/bench_bc1.json 2> $O/bench_bc1.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc2 > $O/bench_bc2.json 2> $O/bench_bc2.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc3 > $O/bench_bc3.json 2> $O/bench_bc3.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc3 --settings 0,0,0 --host-array-gib 0 > $O/bench_bc3_standard.json 2> $O/bench_bc3_standard.err || exit 1
timeout -k 10 300 python3 $R/bench.py --format bc7 > $O/bench_bc7.json 2> $O/bench_bc7.err || exit 1
timeout -k 10 300 python3 $R/bench.py --workload archive > $O/bench_archive.json 2> $O/bench_archive.err || exit 1
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --workload archive --archive-split range --gpus 2 --size-gib 2 --steps 10 > $O/bench_archive_range_gpus2_gloo.json 2> $O/bench_archive_range.err || exit 1
step "N = 2 rehearsals on one GPU (gloo): self-launch, weak and strong"
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --size-gib 4 --steps 10 > $O/bench_gpus2_weak_gloo.json 2> $O/bench_gpus2_weak.err || exit 1
DXTLT_BENCH_BACKEND=gloo timeout -k 10 300 python3 $R/bench.py --gpus 2 --scaling strong --steps 10 > $O/bench_gpus2_strong_gloo.json 2> $O/bench_gpus2_strong.err || exit 1
step done
