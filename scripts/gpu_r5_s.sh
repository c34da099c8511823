#!/bin/bash
# round 5, pass S: grid knobs of the decode launches after the dot2 rewrite (one process per line of configurations)
mkdir -p gpurun_out/r5s
export TMPDIR=/tmp
timeout 900 python tools/decode_sweep.py --configs "pf=0:0:0;co_blocks=128,ln_blocks=1024,batch_ln_blocks=1024;co_blocks=128,ln_blocks=1024,batch_ln_blocks=1024,out_blocks=512;co_blocks=144,ln_blocks=1024,batch_ln_blocks=896;co_blocks=128,ln_blocks=896,batch_ln_blocks=1280;co_blocks=192,ln_blocks=1195,batch_ln_blocks=1024;co_blocks=224;co_blocks=128,ln_blocks=1024,batch_ln_blocks=1024" > gpurun_out/r5s/sweep2.txt 2> gpurun_out/r5s/sweep2.err
cat gpurun_out/r5s/sweep2.txt; tail -2 gpurun_out/r5s/sweep2.err
SHOWO_DECODE_LNR=4 timeout 900 python tools/decode_sweep.py --configs "co_blocks=128;co_blocks=128,ln_blocks=896;co_blocks=128,ln_blocks=512" > gpurun_out/r5s/sweep3.txt 2> gpurun_out/r5s/sweep3.err
cat gpurun_out/r5s/sweep3.txt; tail -2 gpurun_out/r5s/sweep3.err
