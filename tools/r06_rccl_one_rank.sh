#!/bin/bash
# The N > 1 code path over RCCL itself (backend nccl) with ONE rank under the launcher (two ranks cannot share a device under RCCL):
# process-group init with device_id, the gloo side group, the early barrier, MAX all-reduce, and the per_rank all-gather that now
# carries every rank's exactness flags -- for the buffer, BC7 and archive workloads.   gpurun -- 'bash tools/r06_rccl_one_rank.sh'
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06_rccl
mkdir -p $O
export DXTLT_BENCH_FORCE_DIST=1
run() { name=$1; shift
  timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
      $R/bench.py --gpus 1 "$@" > $O/$name.json 2> $O/$name.err || { echo "$name failed"; tail -5 $O/$name.err; exit 1; }
  python3 - "$O/$name.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{"metric"')][0])
print(sys.argv[1].split("/")[-1], d["value"], d["backend"], d["world_size_seen"], d["per_rank"])
PY
}
run buffer --size-gib 2 --steps 5 --warmup 2 --host-array-gib 0 --no-cpu-baseline
run bc7 --format bc7 --size-gib 1 --steps 5 --warmup 2 --no-cpu-baseline
run archive --workload archive --size-gib 1 --steps 3 --warmup 1
echo done
