# HBM-side traffic of the hand-written kernels inside the training step (separate rocprofv3 --pmc passes; run from the repo root on
# the GPU box).  Eager launches (GRIT_STEP_GRAPH=0): the same kernels as the replayed graph, dispatched one by one for the counters.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export GRIT_STEP_GRAPH=0
timeout 500 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2 > /dev/null 2>&1
echo "fetch pass rc=$?"
timeout 500 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2 > /dev/null 2>&1
echo "write pass rc=$?"
python3 - <<'PY' > $O/pmc_in_step.txt
import csv, glob, re
print("HBM-side traffic of the hand-written kernels INSIDE the benchmark step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,")
print("GRIT_STEP_GRAPH=0 python3 bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2; mean per launch; FETCH_SIZE doubled per the")
print("gfx950 note in MI355X_MICROARCH.md; counters in KiB).  `family:` rows = all instantiations of a kernel family, launch-weighted.")
tot, fam = {}, {}
FAMILY = (("family:gemm_nt_bf16", "gemm_nt_bf16<256,"), ("family:gemm_short", "gemm_nt_bf16<64,"), ("family:gemm_w4", "gemm_w4_bf16<"), ("family:wgrad_tn", "wgrad_tn"), ("family:wgrad_small", "wgrad_small"), ("family:gemm_lib_long", "Custom_Cijk"))
for name, d in (("FETCH_SIZE", "/tmp/pmc_fetch"), ("WRITE_SIZE", "/tmp/pmc_write")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"]
            for fn, sub in FAMILY:
                if sub in k:
                    a = fam.setdefault(fn, {}).setdefault(name, [0.0, 0])
                    a[0] += float(r["Counter_Value"]); a[1] += 1
            m = re.search(r"(msda_\w+(?:<\d>)?|winattn_\w+|gemm_nt_bf16<[^>]*>|gemm_w4_bf16<\d, \d>|wgrad_tn4?_256\w*|wgrad_small\w*|ln_fwd|ln_bwd|adam_flat|gn_\w+|colsum_kernel|slab_sum_grouped_kernel|slab_sum_kernel|relu_dropout|attn_mfma_\w+|Custom_Cijk\w{0,60})", k)
            if not m: continue
            a = tot.setdefault(m.group(1), {}).setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
for table in (fam, tot):
    for k, v in sorted(table.items()):
        f = v.get("FETCH_SIZE", [0, 0]); w = v.get("WRITE_SIZE", [0, 0])
        fm = f[0] / f[1] if f[1] else float('nan'); wm = w[0] / w[1] if w[1] else float('nan')
        print(f"{k[:60]:60s} launches {max(f[1], w[1]):5d}  FETCH_SIZE {fm:12.1f}  WRITE_SIZE {wm:12.1f}  HBM-side bytes/launch {(2 * fm + wm) * 1024:14.0f}")
PY
cat $O/pmc_in_step.txt | cut -c1-200 | head -50
