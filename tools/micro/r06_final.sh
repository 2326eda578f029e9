#!/bin/bash
# round-6 final records: the driver's command, the default command, rocprofv3 --stats of the default command, the steady-state breakdown, PMC
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/stats_run
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/stats_run -- python3 $R/bench.py --no-cpu-baseline --no-analysis > $O/bench_default_under_rocprof.json 2> $O/rocprof.err
cp /tmp/stats_run/*/*_kernel_stats.csv $O/bench_default_command_kernel_stats.csv 2>/dev/null
cd $R
TAG=_final bash tools/micro/r06_prof_step.sh > $O/prof_step_final.log 2>&1
bash tools/micro/r06_pmc.sh > $O/pmc.log 2>&1
for f in bench_driver_command bench_default bench_default_under_rocprof; do python3 - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1].split('/')[-1], round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms', 'roofline', d.get('roofline',{}).get('kernel','')[:40], d.get('roofline',{}).get('frac'))
except Exception as e: print(sys.argv[1], 'NO LINE', e)
PY
done
head -5 $O/bench_default_command_kernel_stats.csv | cut -c1-200
head -40 $O/pmc_in_step.txt | cut -c1-180
