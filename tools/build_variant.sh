#!/bin/bash
# Builds a variant of libdgnn_hip.so into dgnn_amd/variants/<name>.so (git-ignored, travels with gpurun).
#   tools/build_variant.sh <name> [extra hipcc flags applied to fused_mfma.hip / fused_bf16.hip / fused.hip / decoder.hip]
set -e
cd "$(dirname "$0")/../dgnn_amd/csrc"
name=$1; shift
out=../variants/$name.so
mkdir -p ../variants build_$name
for f in plan aggregate gemm norm sampler mesh ingest reorder infer halo adam train chain loss; do
  [ build/$f.o -nt $f.hip ] || make -s build/$f.o
  cp build/$f.o build_$name/$f.o
done
for f in fused fused_mfma fused_bf16 decoder; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -fno-slp-vectorize "$@" -c $f.hip -o build_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_$name/*.o -o $out
rm -rf build_$name
echo built $out
