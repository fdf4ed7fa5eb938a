#!/bin/bash
# A/B aid: build tools/_ab/lib_<name>.so from the in-tree objects with ONE source recompiled (default gemm.hip), optionally
# from another copy of that source and with extra hipcc flags.   tools/build_variant.sh <name> [src.hip] [flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/multi-modal-retrieval-system-image-search-and-data-governance_amd/csrc
NAME=$1; SRC=${2:-$CS/gemm.hip}; shift; shift || true
BASE=$(basename "$SRC" .hip)
mkdir -p "$ROOT/tools/_ab" /tmp/mmr_variants
EXTRA=""
if [ "$BASE" = "vit_ops" ]; then EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -fno-honor-nans"; fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wno-unused-result -Wno-unused-value -I"$CS" $EXTRA "$@" \
    -c "$SRC" -o /tmp/mmr_variants/${NAME}_${BASE}.o
OBJS=""
for o in api_common comm search gemm vit_ops tower preprocess; do
  if [ "$o" = "$BASE" ]; then OBJS="$OBJS /tmp/mmr_variants/${NAME}_${BASE}.o"; else OBJS="$OBJS $CS/_obj/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/_ab/lib_${NAME}.so" $OBJS -ldl
echo "$ROOT/tools/_ab/lib_${NAME}.so"
