#!/bin/bash
# Builds libphnrec_lcrc.so of another commit (or of this tree with extra compiler flags) into phnrec_amd/lib/ab/
# for tools/ab_kernel.py:   tools/build_ab_lib.sh <git-ref|WORKTREE> <name> [extra HIPFLAGS...]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="$1"; NAME="$2"; shift 2
TMP="$(mktemp -d /tmp/ablib.XXXXXX)"
if [ "$REF" = "WORKTREE" ]; then
    mkdir -p "$TMP/phnrec_amd" "$TMP/include"
    cp -r "$ROOT/phnrec_amd/csrc" "$TMP/phnrec_amd/csrc"
    cp "$ROOT"/include/*.h "$TMP/include/"
else
    git -C "$ROOT" archive "$REF" phnrec_amd/csrc include | tar -x -C "$TMP"
fi
make -s -C "$TMP/phnrec_amd/csrc" ../lib/libphnrec_lcrc.so HIPFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-function --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form $*" 2>&1 | grep -E "error" || true
mkdir -p "$ROOT/phnrec_amd/lib/ab"
cp "$TMP/phnrec_amd/lib/libphnrec_lcrc.so" "$ROOT/phnrec_amd/lib/ab/lib$NAME.so"
rm -rf "$TMP"
echo "phnrec_amd/lib/ab/lib$NAME.so"
