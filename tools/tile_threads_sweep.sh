#!/bin/bash
# Aligned tiles, lanes per workgroup (64 / 128 / 256 / 512) per format at the bench size: value, forward and inverse fraction
# of the HBM peak.  Run through gpurun from the repo root; profiles/r02_n_tile_threads.txt holds the round-2 result.
for fmt in bc1 bc2 bc3; do for th in 64 128 256 512; do
  python3 bench.py --format $fmt --tile-threads $th --steps 10 --warmup 2 --no-cpu-baseline --host-array-gib 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('$fmt', $th, d['value'], r['frac'], r['inverse_kernel']['frac'])" || exit 1
done; done
