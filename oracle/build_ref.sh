#!/bin/bash
# Builds the REFERENCE fast_surf Fortran (unmodified, where it lies under
# /root/reference) into oracle/_ref/libfast_surf_ref.so with AMD flang.
# TEST INFRASTRUCTURE ONLY: the result is a checker / CPU baseline, never a
# product path.  Nothing is copied into the repo; oracle/_ref/ is git-ignored.
# Flags: -O2, no FMA contraction (defect 10: fixtures come
# from the -O2 no-FMA build).
set -euo pipefail
REF=${REFERENCE_ROOT:-/root/reference}/fast_surf_src
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
FLANG=${FLANG:-/opt/rocm/lib/llvm/bin/flang}
if [ ! -d "$REF" ]; then
  echo "build_ref.sh: $REF absent (GPU box?) - keeping prebuilt $OUT" >&2
  exit 0
fi
mkdir -p "$OUT"
"$FLANG" -O2 -ffp-contract=off -fPIC -shared -ffixed-line-length-none \
  -o "$OUT/libfast_surf_ref.so" \
  "$REF/fast_surf.f" "$REF/flat1.f" "$REF/init.f" "$REF/calcul.f" \
  "$REF/surfa.f" "$REF/mchdepsun.f"
# second build of the same unmodified sources with FMA contraction: only used by
# tests/golden/make_golden_spread.py to record how far the reference moves from ITSELF under another
# legal compilation (defect 10) - the yardstick for ill-conditioned entries
"$FLANG" -O2 -ffp-contract=fast -march=native -fPIC -shared -ffixed-line-length-none \
  -o "$OUT/libfast_surf_ref_fma.so" \
  "$REF/fast_surf.f" "$REF/flat1.f" "$REF/init.f" "$REF/calcul.f" \
  "$REF/surfa.f" "$REF/mchdepsun.f"
"$FLANG" --version | head -1 > "$OUT/BUILD_INFO.txt"
echo "flags: -O2 -ffp-contract=off -fPIC -shared -ffixed-line-length-none" >> "$OUT/BUILD_INFO.txt"
echo "built $OUT/libfast_surf_ref.so"
