cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/step_trace -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/step_trace.log 2>&1
python3 $R/tools/steady_profile.py $R/gpurun_out/step_trace > $R/gpurun_out/steady_fused.txt 2>&1
head -50 $R/gpurun_out/steady_fused.txt | cut -c1-200
