#!/bin/bash
# per-kernel time of ONE 20-step beam decode (graph replay), bf16 weights, batch 64
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_dec -o dec -- python3 $R/tools/bench_decode.py --bf16 --decode-only 4 > $R/gpurun_out/r06/decode_only.log 2>&1
grep decode_20 $R/gpurun_out/r06/decode_only.log | tail -3
python3 $R/tools/decode_kernel_profile.py /tmp/prof_dec | cut -c1-170 | tee $R/gpurun_out/r06/decode_kernel_stats.txt
