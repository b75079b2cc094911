#!/bin/bash
# One command that turns "GPU == oracle" into "oracle == rs_pbrt" for every scene of tests/golden/ref_scenes (oracle/REFERENCE_FIXTURES.md), on a machine
# that HAS a Rust toolchain (this image does not: SURVEY.md section 8c).
#   RS_PBRT=<checkout of wahn/rs_pbrt v0.9.12 with rust_shim/rs_pbrt.patch applied and rust_shim/{gpu,ffi,refdump}.rs under src/gpu/> tools/make_reference_fixtures.sh [scene ...]
# For each scene: RSPT_REF_DUMP=<dir> RSPT_REF_DUMP_LI=1 cargo run --release --bin rs_pbrt -- -t 1 <scene>.pbrt   (refdump.rs writes bvh_nodes.bin,
# bvh_prims.bin, film.bin, li.bin, hits.bin, occluded.bin, meta.json), then tools/ref_to_npz.py packs them into tests/golden/ref_<scene>.npz.
# Afterwards: python -m pytest tests/test_reference_fixtures.py   (the skip disappears; a fabricated dump is refused).
set -eu
repo=$(cd "$(dirname "$0")/.." && pwd)
: "${RS_PBRT:?set RS_PBRT to the patched rs_pbrt checkout}"
command -v cargo > /dev/null || { echo "cargo not found: run this where a Rust toolchain exists" >&2; exit 2; }
scenes=("$@")
if [ ${#scenes[@]} -eq 0 ]; then
  for f in "$repo"/tests/golden/ref_scenes/*.pbrt; do scenes+=("$(basename "$f" .pbrt)"); done
fi
work=$(mktemp -d)
for s in "${scenes[@]}"; do
  out="$work/$s"; mkdir -p "$out"
  cp "$repo/tests/golden/ref_scenes/rays.bin" "$out/"
  (cd "$RS_PBRT" && RSPT_REF_DUMP="$out" RSPT_REF_DUMP_LI=1 cargo run --release --bin rs_pbrt -- -t 1 "$repo/tests/golden/ref_scenes/$s.pbrt")
  python3 "$repo/tools/ref_to_npz.py" "$out" "$s"
done
echo "fixtures written to $repo/tests/golden/ref_*.npz; now: python -m pytest tests/test_reference_fixtures.py"
