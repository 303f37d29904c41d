#!/bin/bash
# usage: bash scripts/build_variant.sh <tag> <source.hip> [-DFLAG ...]
# A diagnostic build of libmisplat with ONE source compiled under extra defines, the other objects as built:
# collab_splats_amd/_exp/libmisplat_<tag>.so (git-ignored; delete after use -- it ships to the GPU box while it exists).
# Run with MISPLAT_LIB=collab_splats_amd/_exp/libmisplat_<tag>.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
TAG=$1; SRC=$2; shift 2
python -m collab_splats_amd.build > /dev/null
mkdir -p collab_splats_amd/_exp
EXTRA=""
[ "$SRC" = "project.hip" ] && EXTRA="-ffp-contract=off"
[ "$SRC" = "blend.hip" ] && [ -z "$MISPLAT_VARIANT_SRC" ] && EXTRA="-fno-slp-vectorize"
# MISPLAT_VARIANT_SRC: compile THIS file in place of csrc/$SRC (an older revision, `git show REV:path > file`)
SRCPATH=${MISPLAT_VARIANT_SRC:-collab_splats_amd/csrc/$SRC}
O=collab_splats_amd/_exp/${TAG}_${SRC%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I include -I collab_splats_amd/csrc -Wall -Wno-unused-function -fno-fast-math $EXTRA "$@" \
    -c $SRCPATH -o $O
OBJS=""
for f in collab_splats_amd/_obj/*.o; do
    [ "$(basename $f)" = "${SRC%.hip}.o" ] && OBJS="$OBJS $O" || OBJS="$OBJS $f"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o collab_splats_amd/_exp/libmisplat_$TAG.so $OBJS
rm -f $O
ls -la collab_splats_amd/_exp/libmisplat_$TAG.so
