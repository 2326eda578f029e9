# The bench lines profiles/r02 records next to the headline (run on the GPU box from the repo root).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout 240 python bench.py > $O/bench_default.json 2> $O/bench_default.err
GRIT_MSDA_BWD_F32ACC=1 timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 > $O/bench_msda_f32acc.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --points spread > $O/bench_points_spread.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 30 --warmup 10 --ragged > $O/bench_ragged.json 2>/dev/null
timeout 240 python bench.py --no-cpu-baseline --steps 6 --warmup 3 --fp32 > $O/bench_fp32.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/default_stats -- python3 $R/bench.py --no-cpu-baseline --no-analysis > $O/bench_default_under_rocprof.json 2>/dev/null
cp /tmp/default_stats/*/*_kernel_stats.csv $O/bench_default_command_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/steady -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 $R/tools/steady_profile.py /tmp/steady > $O/bench_bs32_steady_state.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 4 --warmup 3 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 4 --warmup 3 > /dev/null 2>&1
python3 - <<'PY' > $O/pmc_in_step.txt
import csv, glob, collections
print("HBM-side traffic of the hand-written kernels INSIDE the benchmark step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,")
print("python3 bench.py --no-cpu-baseline --no-analysis --steps 4 --warmup 3; mean per launch; FETCH_SIZE doubled per the gfx950 note in")
print("MI355X_MICROARCH.md; KiB)")
tot = {}
for name, d in (("FETCH_SIZE", "/tmp/pmc_fetch"), ("WRITE_SIZE", "/tmp/pmc_write")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"].split("(")[0][:70]
            if not any(s in k for s in ("msda", "winattn", "gemm_nt", "ln_", "adam_flat", "gn_")): continue
            a = tot.setdefault(k, {}).setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(tot.items()):
    f = v.get("FETCH_SIZE", [0, 1]); w = v.get("WRITE_SIZE", [0, 1])
    fm, wm = f[0] / max(f[1], 1), w[0] / max(w[1], 1)
    print(f"{k:72s} launches {f[1]:5d}  FETCH_SIZE {fm:12.1f}  WRITE_SIZE {wm:12.1f}  HBM-side bytes/launch {int((2 * fm + wm) * 1024):12d}")
PY
for f in $O/*.json; do echo "== $f"; tail -1 $f | cut -c1-200; done
