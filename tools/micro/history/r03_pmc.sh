# HBM-side traffic of the hand-written kernels inside the training step (separate rocprofv3 --pmc passes; run from the repo root on the GPU box)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2 > /dev/null 2>&1
echo "fetch pass rc=$?"
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_write -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2 > /dev/null 2>&1
echo "write pass rc=$?"
python3 - <<'PY' > $O/pmc_in_step.txt
import csv, glob, re
print("HBM-side traffic of the hand-written kernels INSIDE the benchmark step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes,")
print("python3 bench.py --no-cpu-baseline --no-analysis --steps 3 --warmup 2; mean per launch; FETCH_SIZE doubled per the gfx950 note in")
print("MI355X_MICROARCH.md; counters in KiB)")
tot = {}
for name, d in (("FETCH_SIZE", "/tmp/pmc_fetch"), ("WRITE_SIZE", "/tmp/pmc_write")):
    for f in glob.glob(d + "/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name: continue
            k = r["Kernel_Name"]
            m = re.search(r"(msda_bwd_d64_pk<true>|msda_bwd_d64_pk<false>|msda_\w+|winattn_\w+|gemm_nt_bf16<[^>]*>|gemm_pp_bf16<[^>]*>|ln_fwd|ln_bwd|adam_flat|gn_\w+|colsum_kernel|slab_sum_grouped_kernel|slab_sum_kernel|relu_dropout|gate_bwd_\w|msda_geometry_\w+|box_refine|topk_rows)", k)
            if not m: continue
            a = tot.setdefault(m.group(1), {}).setdefault(name, [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
for k, v in sorted(tot.items()):
    f = v.get("FETCH_SIZE", [0, 0]); w = v.get("WRITE_SIZE", [0, 0])
    fm = f[0] / f[1] if f[1] else float('nan'); wm = w[0] / w[1] if w[1] else float('nan')
    print(f"{k:44s} launches {max(f[1], w[1]):5d}  FETCH_SIZE {fm:12.1f}  WRITE_SIZE {wm:12.1f}  HBM-side bytes/launch {(2 * fm + wm) * 1024:14.0f}")
PY
cat $O/pmc_in_step.txt | cut -c1-180 | head -40
