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
for f in crg_api gemm_conv norms attention small_ops; do
  extra=""; [ "$f" = attention ] && grep -q "fno-honor-nans" "$ROOT/cremage_amd/build.py" && extra="-fno-honor-nans"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra -c $f.hip -o $f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libcrg_$TAG.so" crg_api.o gemm_conv.o norms.o attention.o small_ops.o
echo "built tools/ab/libcrg_$TAG.so from $REV"
