#!/bin/bash
# round-4 pass m (no torch, ~20 s): row-range split of the dense|fc2 launch measured in the native harness (tools/gemm_bench mode 7)
mkdir -p gpurun_out
O=gpurun_out/r4m_rowsplit.txt
{ echo "== production dispatch (tuner on) and forced 256-row tiles; cooperative split-K reduction (default)";
  timeout 60 tools/gemm_bench 0:4:0:0,0:4:0:256 7;
  echo "== SHOWO_GEMM_COOP=0 (last-arriver reduction)";
  SHOWO_GEMM_COOP=0 timeout 40 tools/gemm_bench 0:4:0:0 7;
  echo "== SHOWO_GEMM_SPLITK=0";
  SHOWO_GEMM_SPLITK=0 timeout 40 tools/gemm_bench 0:4:0:0 7; } > $O 2>&1
cat $O
