R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "--- fresh process, batch 16, graph"
timeout 600 python -X faulthandler bench.py --batch 16 --steps 6 --warmup 4 --no-cpu-baseline --no-analysis > $O/b16.json 2> $O/b16.err
echo rc=$?; grep -v Warning $O/b16.err | tail -12 | cut -c1-200; cut -c1-300 $O/b16.json
echo "--- second capture at batch 32"
GRIT_BENCH_C3B=32 timeout 600 python -X faulthandler bench.py --steps 6 --warmup 4 --no-cpu-baseline > $O/b32b.json 2> $O/b32b.err
echo rc=$?; grep -v Warning $O/b32b.err | tail -12 | cut -c1-200; python -c "
import json;d=json.loads(open('$O/b32b.json').read().strip().splitlines()[-1]);print(d['value'],d['config3_bs16'])"
