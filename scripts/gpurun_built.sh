#!/bin/bash
# local wrapper: make sure libshowo_hip.so and tools/gemm_bench match the sources, then hand the command to gpurun
# usage: scripts/gpurun_built.sh <timeout_s> '<command>'
set -e -o pipefail
cd "$(dirname "$0")/.."
bash show-o_amd/csrc/build.sh | tail -1
if [ ! -f tools/gemm_bench ] || [ tools/gemm_bench.cpp -nt tools/gemm_bench ] || [ show-o_amd/libshowo_hip.so -nt tools/gemm_bench ]; then
  hipcc --offload-arch=gfx950 -O2 tools/gemm_bench.cpp -o tools/gemm_bench -Lshow-o_amd -lshowo_hip -Wl,-rpath,'$ORIGIN/../show-o_amd' 2>&1 | grep -v warning | grep -i error || true
fi
python - <<'PY'
import sys; sys.path.insert(0, '.')
import showo_amd
showo_amd._lib.load()  # every prototype resolves against the fresh library
print("lib ok")
PY
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
