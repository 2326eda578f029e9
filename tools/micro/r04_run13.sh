R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04; mkdir -p $O
cd $R
for v in 0 1 0 1; do
GRIT_WGRAD_TN_W4=$v timeout 600 python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-analysis > $O/ab_tn4_$v.json 2>/dev/null
python -c "
import json;d=json.loads(open('$O/ab_tn4_$v.json').read().strip().splitlines()[-1]);print('TN_W4=$v', round(d['value'],1), round(d['ms_per_step'],2))"
done
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["config"]["step_graph"], d["config"]["step_graph_error"])
r=d["roofline"]; print("roofline:", r["kernel"][:60], r["ms_per_step"], r["bound"], r["frac"], r.get("frac_mfma"), r.get("frac_hbm"))
for k,v in d["gemm"].items():
    if isinstance(v,dict) and "frac" in v: print(k, "%.1f l/step %.2f ms %.1f us  mfma %.3f hbm %.3f"%(v["launches_per_step"],v["ms_per_step"],v["avg_launch_us"],v["frac_mfma"],v["frac_hbm"]))
    elif isinstance(v,dict): print(k,v)
print("msda", d["roofline_msda"]["frac_touched"], d["roofline_msda"]["frac_compulsory_8d"])
print("config3", d["config3_bs16"]); print("decode", d.get("decode_config5",{}).get("captions_per_sec")); print("cpu", d["cpu_baseline"]["value"])
PY
tail -3 $O/bench_default.err | cut -c1-300
