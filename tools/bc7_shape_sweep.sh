#!/bin/bash
# BC7 kernel shape sweep on one box: workgroup lanes, uniform / skewed / reserved mixes (tools/bc7_bench.py).
set -u
for lanes in 256 512; do
  for mix in uniform skewed reserved; do
    echo "LANES=$lanes $(DXTLT_BC7_LANES=$lanes python tools/bc7_bench.py 4 $mix 2>/dev/null | cut -c1-330)"
  done
done
