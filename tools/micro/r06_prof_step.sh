#!/bin/bash
# kernel trace of the graph-replayed step: per-kernel ms/step, idle gaps, ordered kernel list of the decoder phase
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06; mkdir -p $O
rm -rf /tmp/step_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/step_trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-analysis > $O/step_trace${TAG}.log 2>&1
GRIT_PROFILE_STEP_SEQUENCE=$O/step_sequence${TAG}.txt GRIT_PROFILE_SEQUENCE=$O/decoder_phase_sequence${TAG}.txt python3 $R/tools/steady_profile.py /tmp/step_trace > $O/bench_bs32_steady_state${TAG}.txt 2>&1
head -60 $O/bench_bs32_steady_state${TAG}.txt | cut -c1-200
grep -n "idle\|decoder phase" $O/bench_bs32_steady_state${TAG}.txt | head
tail -3 $O/step_trace${TAG}.log | cut -c1-300
