#!/bin/bash
# fwd / inv roofline fraction for every settings combination of one format (run on the GPU box)
f=${1:-bc3}
for v in 0 1 2 3; do for sa in 0 1; do for sc in 0 1; do
  if [ "$f" != "bc3" ] && [ "$sa" = "1" ]; then continue; fi
  python bench.py --steps 10 --warmup 2 --format $f --no-cpu-baseline --settings $v,$sa,$sc 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$f settings $v,$sa,$sc fwd', d['roofline']['frac'], 'inv', d['roofline']['inverse_kernel']['frac'])"
done; done; done
