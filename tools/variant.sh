#!/bin/bash
# Builds raytracing_amd/lib/libv_<name>.so with extra compiler flags: tools/variant.sh <name> [-DRT_X=..]...
root="$(cd "$(dirname "$0")/.." && pwd)"; cd "$root/raytracing_amd/csrc" || exit 1
name=$1; shift
make OUT=../lib/libv_$name.so CXXFLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -w $*" ../lib/libv_$name.so > /tmp/mk_$name.log 2>&1 || { grep -i error /tmp/mk_$name.log; exit 1; }
