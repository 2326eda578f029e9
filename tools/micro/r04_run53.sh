R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_winattn_gpu.py -x -q 2>&1 | tail -6 | cut -c1-250
for i in 1 2 3; do
timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_waskip_$i.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_waskip_$i.json').read().strip().splitlines()[-1]);print('winattn zero-skip', round(d['value'],1), round(d['ms_per_step'],2), d['final_loss'])"
done
