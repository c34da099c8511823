#!/bin/bash
# Builds libshowo_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU).
# A failed compile stops the build: the stale object is removed before compiling and every job's exit status is checked.
set -e
cd "$(dirname "$0")"
OUT=${SHOWO_BUILD_OUT:-../libshowo_hip.so}   # SHOWO_BUILD_OUT / SHOWO_BUILD_DIR / SHOWO_BUILD_FLAGS: a second library for same-box A/B runs
BD=${SHOWO_BUILD_DIR:-_build}
# -Wno-inline-asm: the LDS-DMA helpers (common.h glds16_untracked, gemm_tn.hip glds16_sa) list "m0" as clobbered so that the compiler
# never assumes M0 survives them; the backend notes for every inlined copy that m0 is a reserved register (a remark, not a defect)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=on -Wno-unused-result -Wno-unused-value -Wno-inline-asm $SHOWO_BUILD_FLAGS"
OBJS=""
PIDS=""
mkdir -p $BD
for f in gemm2p gemm2p_f16 gemm3w gemm3w_f16 basic gemm gemm_tn attention attention_bwd train_kernels decode decode_batch prompting sampler vq_kernels engine vq_engine train_engine clip_engine image_ops precise; do
  if [ ! -f $BD/$f.o ] || [ $f.hip -nt $BD/$f.o ] || [ common.h -nt $BD/$f.o ] || [ engine.h -nt $BD/$f.o ] || [ gemm_common.h -nt $BD/$f.o ] || [ gemm2p_kernel.h -nt $BD/$f.o ] || [ gemm3w_kernel.h -nt $BD/$f.o ] || [ decode_common.h -nt $BD/$f.o ] || [ prof.h -nt $BD/$f.o ] || [ ../../include/showo_hip.h -nt $BD/$f.o ]; then
    rm -f $BD/$f.o
    hipcc $FLAGS -c $f.hip -o $BD/$f.o &
    PIDS="$PIDS $!"
  fi
  OBJS="$OBJS $BD/$f.o"
done
if [ ! -f $BD/errors.o ] || [ errors.cpp -nt $BD/errors.o ]; then
  rm -f $BD/errors.o
  hipcc --offload-arch=gfx950 -O2 -std=c++17 -fPIC -c errors.cpp -o $BD/errors.o &
  PIDS="$PIDS $!"
fi
FAIL=0
for p in $PIDS; do
  wait $p || FAIL=1
done
if [ $FAIL -ne 0 ]; then
  echo "build failed: a hipcc job returned non-zero" >&2
  exit 1
fi
rm -f $OUT
hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $BD/errors.o -o $OUT
echo "built $(realpath $OUT)"
