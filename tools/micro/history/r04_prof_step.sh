# kernel trace of the graph-replayed step: per-kernel ms/step, idle gaps
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
rm -rf /tmp/step_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/step_trace -- python3 $R/bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-analysis > $O/step_trace.log 2>&1
python3 $R/tools/steady_profile.py /tmp/step_trace > $O/bench_bs32_steady_state.txt 2>&1
head -75 $O/bench_bs32_steady_state.txt | cut -c1-190
grep -n "idle\|decoder phase" $O/bench_bs32_steady_state.txt | head
tail -3 $O/step_trace.log | cut -c1-300
