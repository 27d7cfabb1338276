#!/bin/bash
# Developer tool: build a variant of librrt_hip.so for A/B timing (tools/ab_multi.py): tools/build_variant.sh <name> "<extra hipcc flags>" [alternative render.hip]
# -> rust-ray-tracer_amd/librrt_hip_<name>.so   (variants are *_<name>.so; only librrt_hip.so is the product)
set -e
cd "$(dirname "$0")/../rust-ray-tracer_amd/csrc"
NAME=$1; EXTRA=$2; RENDER=${3:-render.hip}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math ${KFLAGS--mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -simplifycfg-sink-common=false -mllvm -enable-pre=false -mllvm -join-splitedges} -I. $EXTRA -shared -o ../librrt_hip_$NAME.so api.cpp octree.cpp clusters.cpp obj_loader.cpp image_decode.cpp $RENDER -lz 2>/dev/null
echo built librrt_hip_$NAME.so
