R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_stream_kernels_gpu.py -x -q -k "patch_merging" 2>&1 | tail -6
timeout 900 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -4
for v in 0 1 0 1; do
GRIT_MERGE_LN=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_mergeln_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_mergeln_$v.json').read().strip().splitlines()[-1]);print('MERGE_LN=$v', round(d['value'],1), round(d['ms_per_step'],2), d['final_loss'])"
done
