cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03; mkdir -p $O
rm -rf $O/step_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/step_trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-analysis > $O/step_trace.log 2>&1
python3 $R/tools/steady_profile.py $O/step_trace > $O/steady.txt 2>&1
rm -rf $O/step_trace
head -70 $O/steady.txt | cut -c1-200
