#!/bin/bash
# AddressSanitizer + UBSan over the HOST side of libdxtlt_gfx950.so, on the CPU (GPU sanitizers are not available on
# the pool): every .cpp of the product is rebuilt with clang -fsanitize=address,undefined, linked with the normal
# (uninstrumented) .hip objects into build/asan/libdxtlt_gfx950_asan.so, and the CPU test suites that drive the C ABI
# without a device (argument validation, DDS parser and header bits, BC7 shard placement, batch planning, no-device
# error paths) run against it with the sanitizer runtime preloaded -- together with a sanitizer build of the CPU oracle
# and its own suites (golden vectors, AVX2 / AVX-512 ports against the scalar statement, BC7 statements).  usage: tools/asan_host_check.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CLANG=/opt/rocm/lib/llvm/bin/clang++
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
SRC="$ROOT/dxt-lossless-transform_amd/csrc"
OUT="$ROOT/build/asan"
mkdir -p "$OUT"
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import dxt_lossless_transform_amd as p; p.build()"   # .hip objects
objs=()
for f in "$SRC"/*.cpp; do
    o="$OUT/$(basename "$f").o"
    "$CLANG" -x c++ -std=c++17 -O1 -g -fPIC -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined \
        -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -c "$f" -o "$o" &
    objs+=("$o")
done
wait
for f in "$ROOT"/build/obj/*.hip.o; do objs+=("$f"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan "${objs[@]}" \
    -o "$OUT/libdxtlt_gfx950_asan.so" -lpthread
echo "built $OUT/libdxtlt_gfx950_asan.so"
# the oracle too (test infrastructure, but every parity claim and the cpu_baseline legs stand on it): same sources, same flags
# as oracle/Makefile apart from the optimisation level and the sanitizers
/opt/rocm/lib/llvm/bin/clang -O1 -g -fPIC -std=gnu11 -fno-strict-aliasing -fno-omit-frame-pointer \
    -fsanitize=address,undefined -fno-sanitize-recover=undefined -shared-libsan -shared \
    "$ROOT"/oracle/dxtlt_oracle.c "$ROOT"/oracle/dxtlt_oracle_bc7.c "$ROOT"/oracle/dxtlt_oracle_avx2.c "$ROOT"/oracle/dxtlt_oracle_norm.c \
    -o "$OUT/libdxtlt_oracle_asan.so" -lpthread
echo "built $OUT/libdxtlt_oracle_asan.so"
cd "$ROOT"
export DXTLT_ORACLE_SO="$OUT/libdxtlt_oracle_asan.so"
DXTLT_LIB_PATH="$OUT/libdxtlt_gfx950_asan.so" LD_PRELOAD="$RT" \
    ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
    python3 -m pytest -q -m "not gpu" -p no:cacheprovider \
    tests/test_cabi_load.py tests/test_cabi_reference_surface.py tests/test_file_formats.py tests/test_bc7_sharded.py \
    tests/test_batch.py tests/test_numa_affinity.py tests/test_bc7.py tests/test_oracle.py tests/test_compression_gain.py tests/test_color565.py tests/test_decode.py tests/test_normalize.py tests/test_normalize_bc23.py "$@"
