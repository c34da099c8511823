#!/bin/bash
# round 5, pass R: decode GEMVs on v_dot2c_f32_bf16 (+ the VALU-only reductions and the one-round-trip attention prologue): parity + speed
R=$(pwd)
mkdir -p gpurun_out/r5r
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "wave_reductions or decode or gemm or gemv" > gpurun_out/r5r/t_kernels.log 2>&1; tail -3 gpurun_out/r5r/t_kernels.log
timeout 1200 python -m pytest tests/test_decode_batch_gpu.py tests/test_batch_gpu.py -q -x > gpurun_out/r5r/t_batch.log 2>&1; tail -3 gpurun_out/r5r/t_batch.log
timeout 1200 python -m pytest tests/test_modules_gpu.py -q -k "mmu or decode_layer or cache or forward" > gpurun_out/r5r/t_modules.log 2>&1; tail -5 gpurun_out/r5r/t_modules.log
timeout 600 python tools/decode_sweep.py --configs pf=0:0:0 > gpurun_out/r5r/sweep.txt 2> gpurun_out/r5r/sweep.err
cat gpurun_out/r5r/sweep.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r5r/prof -o prof -- python $R/tools/decode_sweep.py --configs pf=0:0:0 --reps 1 > $R/gpurun_out/r5r/prof.log 2>&1
f=$(find $R/gpurun_out/r5r/prof -name "*kernel_stats.csv" | head -1)
grep -E "gemv|attn_decode|y2|seam" $f | cut -c1-160
cd $R
find gpurun_out/r5r -type f ! -name "*stats*" -size +2M -delete
