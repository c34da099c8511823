#!/bin/bash
# Builds libshowo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
OUT=../libshowo_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -Wno-unused-value"
OBJS=""
for f in basic gemm attention attention_bwd train_kernels decode prompting sampler vq_kernels engine vq_engine train_engine clip_engine image_ops; do
  if [ ! -f _build/$f.o ] || [ $f.hip -nt _build/$f.o ] || [ common.h -nt _build/$f.o ] || [ engine.h -nt _build/$f.o ] || [ prof.h -nt _build/$f.o ] || [ ../../include/showo_hip.h -nt _build/$f.o ]; then
    mkdir -p _build
    hipcc $FLAGS -c $f.hip -o _build/$f.o &
  fi
  OBJS="$OBJS _build/$f.o"
done
if [ ! -f _build/errors.o ] || [ errors.cpp -nt _build/errors.o ]; then
  mkdir -p _build
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -c errors.cpp -o _build/errors.o &
fi
wait
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS _build/errors.o -o $OUT
echo "built $(realpath $OUT)"
