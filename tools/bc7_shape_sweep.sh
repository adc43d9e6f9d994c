#!/bin/bash
# BC7 kernel shape sweep on one box: workgroup lanes x mode mix, steady state (tools/bc7_bench.py warms up for 150 ms).
set -u
for lanes in 256 512 1024; do
  for mix in uniform skewed reserved mode6; do
    echo "LANES=$lanes $(DXTLT_BC7_LANES=$lanes python tools/bc7_bench.py 4 $mix 2>/dev/null | cut -c1-330)"
  done
done
