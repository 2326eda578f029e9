R=$GRAFT_REPO_ROOT
cd $R
( time timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" ) 2>&1 | grep -v Warn | tail -5
( time timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > /tmp/b.json 2>/tmp/b.err ) 2>&1 | tail -4
python3 -c "
import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['config3_bs16']['images_per_sec'])"
