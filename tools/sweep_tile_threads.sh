#!/bin/bash
# fwd / inv roofline fraction per format and tile workgroup size (run on the GPU box)
for f in bc1 bc2 bc3; do for t in 64 128 256 512; do
  python bench.py --steps 15 --warmup 3 --format $f --no-cpu-baseline --tile-threads $t 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$f threads $t fwd', d['roofline']['frac'], 'inv', d['roofline']['inverse_kernel']['frac'], 'value', d['value'])"
done; done
