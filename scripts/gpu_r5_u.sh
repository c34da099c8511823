#!/bin/bash
# round 5, pass U: everything that runs through the VQGAN convolutions after the 3-tap-reuse kernel: reference goldens (ids, pixels), image tests,
# training parity, then the VQ / training benches
mkdir -p gpurun_out/r5u
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_modules_gpu.py tests/test_image_gpu.py tests/test_kernels_gpu.py -q -k "magvit or vq or image or conv or gn or lfq" > gpurun_out/r5u/t_vq.log 2>&1; tail -4 gpurun_out/r5u/t_vq.log | cut -c1-300
grep -E "^E  |^FAILED" gpurun_out/r5u/t_vq.log | head -10 | cut -c1-300
timeout 900 python -m pytest tests/test_train_gpu.py -q -x > gpurun_out/r5u/t_train.log 2>&1; tail -3 gpurun_out/r5u/t_train.log | cut -c1-300
timeout 600 python bench.py --workload train > gpurun_out/r5u/train_bench.json 2> gpurun_out/r5u/train_bench.err
grep '"metric"' gpurun_out/r5u/train_bench.json | cut -c1-400
