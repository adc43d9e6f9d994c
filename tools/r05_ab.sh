#!/bin/bash
# Round 5, one box: the GPU suite on the new library, then same-box A/B of whole libraries (ab/libdxtlt_*.so, tools/ab_build_rev.sh)
# on the default bench's legs, twice in alternation, and on the odd-count cases of tools/shift_probe.py.
#   gpurun -- 'bash tools/r05_ab.sh OUTDIR lib1 lib2 ...'     (lib = a name under ab/, or "new" for the in-tree library)
set -u
OUT=$1; shift
mkdir -p $OUT
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
lib_path() { if [ "$1" = new ]; then echo $R/dxt-lossless-transform_amd/libdxtlt_gfx950.so; else echo $R/ab/libdxtlt_$1.so; fi; }
for round in 1 2; do
  for lib in "$@"; do
    echo "== round $round $lib" >> $OUT/ab_legs.txt
    DXTLT_LIB_PATH=$(lib_path $lib) timeout -k 10 400 python tools/ab_bench_legs.py >> $OUT/ab_legs.txt 2>&1 || exit 3
  done
done
[ "${SHIFT_PROBE:-0}" = 1 ] || exit 0
for lib in "$@"; do
  echo "== $lib" >> $OUT/ab_shift_probe.txt
  for only in aligned128 odd plus; do
    DXTLT_LIB_PATH=$(lib_path $lib) PROBE_ONLY=$only timeout -k 10 300 python tools/shift_probe.py 2>&1 | grep -v "WRONG\|forced\|generic\|remap\|policy\|shared\|nt_\|first_form\|identity\|contiguous\|16_only" >> $OUT/ab_shift_probe.txt
  done
done
