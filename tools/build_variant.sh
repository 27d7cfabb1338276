#!/bin/bash
# Developer tool: build a variant of librrt_hip.so for A/B timing (tools/ab_variants.sh, tools/ab_multi.py):
#   tools/build_variant.sh <name> "<extra hipcc flags>" [alternative render.hip]     -> rust-ray-tracer_amd/librrt_hip_<name>.so
# (variants are *_<name>.so; only librrt_hip.so is the product).  One compilation unit, the frame kernels' code-generation switches (csrc/Makefile:
# KFLAGS) unless KFLAGS is set in the environment: the per-ray kernels of a variant are therefore built with the frame kernels' switches.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
cd "$HERE/../rust-ray-tracer_amd/csrc"
NAME=$1; EXTRA=$2; RENDER=${3:-render.hip}
KF=${KFLAGS-$(python3 "$HERE/_kflags.py")}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math $KF -I. $EXTRA -shared -o ../librrt_hip_$NAME.so api.cpp octree.cpp clusters.cpp obj_loader.cpp image_decode.cpp scene_build.hip $RENDER -lz 2>/tmp/build_variant_$NAME.log || { tail -5 /tmp/build_variant_$NAME.log; echo "FAILED librrt_hip_$NAME.so"; exit 1; }
echo built librrt_hip_$NAME.so
