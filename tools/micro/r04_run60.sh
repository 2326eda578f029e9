R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r04
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_driver_command_b.json 2>/dev/null
python3 -c "
import json;d=json.loads(open('gpurun_out/r04/bench_driver_command_b.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config']['step_graph'],d['config3_bs16']['images_per_sec'])"
