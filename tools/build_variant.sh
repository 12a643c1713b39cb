#!/bin/bash
# bash tools/build_variant.sh NAME "EXTRA_FLAGS": a variant build of the library (same sources, extra compiler flags)
# as indirect_learning_pose-shape_amd/lib_NAME.so, objects in a scratch directory; the default library is left alone.
# On the GPU box: bash tools/ab_run.sh "python bench.py ..." keep NAME   ("keep" = the default build)
set -e
cd "$(dirname "$0")/.."
NAME=$1; FLAGS=$2
PKG=indirect_learning_pose-shape_amd; OUT=${TMPDIR:-/tmp}/smplr_${NAME}_build
rm -rf "$OUT"; mkdir -p "$OUT/pkg/csrc" "$OUT/include"
cp $PKG/csrc/*.hip $PKG/csrc/*.h $PKG/csrc/*.cpp $PKG/csrc/Makefile "$OUT/pkg/csrc/"; cp include/smplraster.h "$OUT/include/"
make -C "$OUT/pkg/csrc" -j8 LIB=../lib_$NAME.so ../lib_$NAME.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function $FLAGS" > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
cp "$OUT/pkg/lib_$NAME.so" $PKG/lib_$NAME.so
echo "built $PKG/lib_$NAME.so"
