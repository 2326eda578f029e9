R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
echo "--- all workgroups run the by-product MFMAs"
SHAPES=51200x2048x512,51200x512x2048,51200x1536x512,12800x4096x1024,204800x1024x256 timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | cut -c1-230
echo "--- only k-tile 0"
GRIT_WGRAD_TN_DBG=2 SHAPES=51200x2048x512,51200x512x2048,51200x1536x512,12800x4096x1024,204800x1024x256 timeout 600 python tools/micro/bench_wgrad_tn.py 2>&1 | grep "^M" | cut -c1-230
for v in 2 0 2 0; do
GRIT_WGRAD_TN_DBG=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_csall_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_csall_$v.json').read().strip().splitlines()[-1]);print('DBG=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done
