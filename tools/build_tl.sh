#!/bin/bash
# bash tools/build_tl.sh: the timeline build of the library (-DSMPLR_TL, see csrc/common.h) as
# indirect_learning_pose-shape_amd/lib_tl.so, objects in a scratch directory; the default library is left alone.
# On the GPU box: bash tools/ab_run.sh "python tools/probes/segbwd_timeline.py" tl   (likewise skinbwd_, segbin_)
set -e
cd "$(dirname "$0")/.."
PKG=indirect_learning_pose-shape_amd; OUT=${TMPDIR:-/tmp}/smplr_tl_build
rm -rf "$OUT"; mkdir -p "$OUT/pkg/csrc" "$OUT/include"
cp $PKG/csrc/*.hip $PKG/csrc/*.h $PKG/csrc/*.cpp $PKG/csrc/Makefile "$OUT/pkg/csrc/"; cp include/smplraster.h "$OUT/include/"
make -C "$OUT/pkg/csrc" LIB=../lib_tl.so ../lib_tl.so CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wall -Wno-unused-function -DSMPLR_TL" > "$OUT/build.log" 2>&1 || { tail -20 "$OUT/build.log"; exit 1; }
cp "$OUT/pkg/lib_tl.so" $PKG/lib_tl.so
echo "built $PKG/lib_tl.so"
