#!/bin/bash
# Dev helper: rebuild ONE source of the working tree with extra compiler flags and link it against the other (default-build) objects
# into tools/ab/libcrg_<tag>.so.  Usage: tools/build_one_variant.sh <tag> <source stem, e.g. conv_pp> [-DNAME=VALUE ...]
set -eo pipefail
TAG=$1; SRC=$2; shift; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python -c "import sys; sys.path.insert(0, '$ROOT'); from cremage_amd import build; build.build(verbose=False)"
mkdir -p "$ROOT/tools/ab" /tmp/crg_one_$TAG
extra=""; [ "$SRC" = attention ] && extra="-fno-honor-nans"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$ROOT/include" $extra "$@" -c "$ROOT/cremage_amd/csrc/$SRC.hip" -o /tmp/crg_one_$TAG/$SRC.o
OBJS=$(ls "$ROOT"/cremage_amd/csrc/_obj/*.o | grep -v _f16.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libcrg_$TAG.so" $OBJS /tmp/crg_one_$TAG/$SRC.o
echo "built tools/ab/libcrg_$TAG.so ($SRC: $*)"
