#!/bin/bash
# round 5, pass T: the 3-tap-reuse split conv kernel: parity (new test + the existing conv / GroupNorm tests), then the A/B on the VQGAN's launches
mkdir -p gpurun_out/r5t
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "conv" > gpurun_out/r5t/t_conv.log 2>&1; tail -5 gpurun_out/r5t/t_conv.log | cut -c1-300
grep -E "^E  " gpurun_out/r5t/t_conv.log | head -8 | cut -c1-300
timeout 300 python tools/conv_ab.py > gpurun_out/r5t/ab_on.txt 2>&1; cat gpurun_out/r5t/ab_on.txt | grep -v amdgpu.ids
SHOWO_CONV_3TAP=0 timeout 300 python tools/conv_ab.py > gpurun_out/r5t/ab_off.txt 2>&1; cat gpurun_out/r5t/ab_off.txt | grep -v amdgpu.ids
