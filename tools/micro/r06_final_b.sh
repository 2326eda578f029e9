R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd $R
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command_b.json 2> $O/bench_driver_command_b.err
python3 bench.py > $O/bench_default_b.json 2> $O/bench_default_b.err
python3 __graft_entry__.py 2>&1 | tail -3
for f in bench_driver_command_b bench_default_b; do python3 - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d.get('roofline') or {}
print(sys.argv[1].split('/')[-1], round(d['value'],1),'img/s', round(d['ms_per_step'],2),'ms', 'roofline', r.get('kernel','')[:30], r.get('frac'), r.get('traffic'), r.get('traffic_source'), d.get('decode_config5',{}).get('tokens_match_fixture'), d.get('cpu_baseline',{}).get('value'))
PY
done
