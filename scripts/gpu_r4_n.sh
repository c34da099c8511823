#!/bin/bash
# round-4 pass n (no torch, ~15 s): is the full-chip k-tile time a clock / power bound or a shared-bandwidth bound?  (tools/gemm_bench mode 8)
mkdir -p gpurun_out
O=gpurun_out/r4n_power_vs_bandwidth.txt
{ echo "== random normal operands: ring kernel at 256-row tiles (2256), plain 2-phase kernel (256), tuner";
  timeout 40 tools/gemm_bench 0:4:0:2256,0:4:0:256,0:4:0:0 8;
  echo "== all-zero operands (GEMM_BENCH_ZERO=1), same binaries, same launches";
  GEMM_BENCH_ZERO=1 timeout 40 tools/gemm_bench 0:4:0:2256,0:4:0:256,0:4:0:0 8; } > $O 2>&1
cat $O
