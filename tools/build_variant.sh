#!/bin/bash
# Dev helper: build the WORKING TREE's kernels with extra compiler flags into tools/ab/libcrg_<tag>.so (A/B of compile-time knobs).
# Usage: tools/build_variant.sh <tag> [-DNAME=VALUE ...]
set -eo pipefail
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/crg_var_$TAG
rm -rf "$W"; mkdir -p "$W" "$ROOT/tools/ab"
cd "$ROOT/cremage_amd/csrc"
for f in crg_api gemm_conv conv_pp gemm_ring lngemm norms attention small_ops; do
  extra=""; [ "$f" = attention ] && extra="-fno-honor-nans"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" $extra "$@" -c $f.hip -o "$W/$f.o" &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libcrg_$TAG.so" "$W"/*.o
echo "built tools/ab/libcrg_$TAG.so ($*)"
