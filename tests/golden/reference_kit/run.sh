#!/bin/bash
# One command for whoever has `cargo` and a checkout of the reference: produces tests/golden/reference_out/, the reference's own
# transformed bytes for the inputs of tests/golden/make_reference_inputs.py, which turn the dormant tests of
# tests/test_reference_vectors.py on (oracle == reference on the CPU, HIP == reference with -m gpu).
#
#     tests/golden/reference_kit/run.sh /path/to/dxt-lossless-transform          # the directory that holds src/Cargo.toml
#
# Needs: cargo (a toolchain recent enough for the reference itself), python3 with numpy (the inputs), network access for cargo to fetch
# the reference's crates.io dependencies.  No GPU, no ROCm.
#
# Nothing of the reference is copied into this repository: the kit is built in a scratch directory and links the reference's
# crates where they lie; only its OUTPUT (data) lands under tests/golden/reference_out/.  Commit MANIFEST.txt, REFERENCE_REV.txt and
# the 288 small .out files `make_reference_inputs.py --commit-list` names; every output is pinned by its length + CRC-32 line.
set -euo pipefail
REF=$(cd "${1:?path of a checkout of Sewer56/dxt-lossless-transform}" && pwd)
test -f "$REF/src/core/dxt-lossless-transform-bc1/Cargo.toml" || { echo "$REF does not look like the reference" >&2; exit 2; }
HERE=$(cd "$(dirname "$0")" && pwd)
GOLDEN=$(dirname "$HERE")
WORK=$(mktemp -d "${TMPDIR:-/tmp}/dxtlt-reference-kit.XXXXXX")
trap 'rm -rf "$WORK"' EXIT
mkdir -p "$WORK/kit/src" "$WORK/in"
sed "s|@REF@|$REF|g" "$HERE/Cargo.toml.in" > "$WORK/kit/Cargo.toml"
cp "$HERE/src/main.rs" "$WORK/kit/src/main.rs"
mkdir -p "$WORK/kit/src/bin" && cp "$HERE/src/bin/auto.rs" "$WORK/kit/src/bin/auto.rs"
cp "$REF/src/Cargo.lock" "$WORK/kit/Cargo.lock" 2>/dev/null || true     # the reference's pinned dependency versions, where they apply
python3 "$GOLDEN/make_reference_inputs.py" "$WORK/in"
( cd "$WORK/kit" && cargo run --release --bin dxtlt-reference-kit -- "$WORK/in" "$WORK/out" ) || {
  echo "retrying without the reference's Cargo.lock" >&2
  rm -f "$WORK/kit/Cargo.lock" && rm -rf "$WORK/out"
  ( cd "$WORK/kit" && cargo run --release --bin dxtlt-reference-kit -- "$WORK/in" "$WORK/out" )
}
# second program: the choices of transform_bcN_auto.  Optional -- if it does not build, the transform outputs above stand alone.
( cd "$WORK/kit" && cargo run --release --bin auto -- "$WORK/in" "$WORK/out" ) || echo "auto kit failed: AUTO_MANIFEST.txt not written (the transform outputs are unaffected)" >&2
rm -rf "$GOLDEN/reference_out" && mkdir -p "$GOLDEN/reference_out"
cp "$WORK/out/MANIFEST.txt" "$GOLDEN/reference_out/"
cp "$WORK/out/AUTO_MANIFEST.txt" "$GOLDEN/reference_out/" 2>/dev/null || true
( git -C "$REF" rev-parse HEAD 2>/dev/null || echo unknown ) > "$GOLDEN/reference_out/REFERENCE_REV.txt"
if [ "${KEEP_ALL:-0}" = 1 ]; then        # every output as a whole file (about 50 MB; not for committing)
  cp "$WORK"/out/*.out "$GOLDEN/reference_out/"
else
  python3 "$GOLDEN/make_reference_inputs.py" --commit-list | while read -r f; do cp "$WORK/out/$f" "$GOLDEN/reference_out/"; done
fi
echo "wrote $(ls "$GOLDEN/reference_out" | wc -l) files to $GOLDEN/reference_out; now: python -m pytest tests/test_reference_vectors.py"
