#!/bin/bash
# Dev helper for A/B runs: build libcrg_hip.so from a git revision's kernel sources into tools/ab/libcrg_<tag>.so
# (the ctypes layer picks it up through CRG_LIB=...).  Usage: tools/build_rev.sh <git-rev> <tag>
set -eo pipefail
REV=$1; TAG=$2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/crg_rev_$TAG
rm -rf "$W"; mkdir -p "$W" "$ROOT/tools/ab"
git -C "$ROOT" archive "$REV" cremage_amd/csrc include | tar -x -C "$W"
cd "$W/cremage_amd/csrc"
for src in *.hip; do
  f=${src%.hip}
  extra=""; [ "$f" = attention ] && grep -q "fno-honor-nans" "$ROOT/cremage_amd/build.py" && extra="-fno-honor-nans"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$W/include" $extra -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libcrg_$TAG.so" *.o
echo "built tools/ab/libcrg_$TAG.so from $REV"
