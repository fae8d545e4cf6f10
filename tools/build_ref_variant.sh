#!/bin/bash
# Build the library of another git revision into variants/ for same-box A/B timing:
#     bash tools/build_ref_variant.sh <git-ref> <name>     ->  variants/libgoldilocks_amd_<name>.so
set -eu
REF=$1; NAME=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WT=/tmp/wt_$NAME
rm -rf "$WT"; git -C "$ROOT" worktree prune; git -C "$ROOT" worktree add -f --detach "$WT" "$REF" > /dev/null
(cd "$WT" && python3 -c "import __graft_entry__ as g; g.build_lib(force=True)" > /dev/null)
mkdir -p "$ROOT/variants"; cp "$WT/libgoldilocks_amd/libgoldilocks_amd.so" "$ROOT/variants/libgoldilocks_amd_$NAME.so"
git -C "$ROOT" worktree remove --force "$WT"
echo "built variants/libgoldilocks_amd_$NAME.so from $REF"
