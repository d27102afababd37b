#!/bin/bash
# Builds an experimental variant of libhvo.so next to the product library (never replaces it):
#   bash tools/build_variant.sh NAME "-DHVO_CLUSTER_WPE=3"   ->  build_variants/NAME/libhvo.so   (use with HVO_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/build_variants/$1
rm -rf $D; mkdir -p $D/pkg/csrc $D/include
cp $R/a-low-texture-robust-hybrid-feature-based-visual-odometry_amd/csrc/*.hip $R/a-low-texture-robust-hybrid-feature-based-visual-odometry_amd/csrc/*.hpp $R/a-low-texture-robust-hybrid-feature-based-visual-odometry_amd/csrc/*.inc $R/a-low-texture-robust-hybrid-feature-based-visual-odometry_amd/csrc/Makefile $D/pkg/csrc/ 2>/dev/null || true
cp $R/include/* $D/include/
make -s -j8 -C $D/pkg/csrc DEFS="$2"
cp $D/pkg/csrc/libhvo.so $D/libhvo.so
rm -rf $D/pkg $D/include
echo built $D/libhvo.so
