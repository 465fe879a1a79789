#!/bin/bash
# Build ab/libbase.so from the csrc of a git revision (default HEAD) for tools/ab.sh
set -eu
cd "$(dirname "$0")/.."
REV=${1:-HEAD}
TMP=$(mktemp -d)
git archive "$REV" point-gnn_amd/csrc include | tar -x -C "$TMP"
mkdir -p ab
(cd "$TMP/point-gnn_amd/csrc" && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
   -ffp-contract=off -I"$TMP/include" -I. -shared -o "$OLDPWD/ab/libbase.so" *.hip)
rm -rf "$TMP"
ls -la ab/libbase.so
