# LDS bank conflicts and wait share of every hand-written kernel INSIDE the training step (one rocprofv3 --pmc pass, kernel trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq_step
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d /tmp/sq_step -- python3 $R/bench.py --no-cpu-baseline --no-analysis --steps 2 --warmup 2 > /dev/null 2>&1
echo "rc=$?"
python3 - <<'PY' > $O/sq_in_step.txt
import csv, glob, re
tot = {}
for f in glob.glob("/tmp/sq_step/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        m = re.search(r"(wgrad_tn_256_grouped|wgrad_tn_256|wgrad_small\w*|msda_\w+|winattn_\w+|gemm_nt_bf16<[^>]*>|ln_fwd|ln_bwd|adam_flat|gn_\w+|colsum\w*|slab_sum_grouped_kernel|attn_mfma_\w+|relbias_\w+|transpose_grouped_kernel)", k)
        if not m: continue
        a = tot.setdefault(m.group(1), {}).setdefault(r["Counter_Name"], [0.0, 0])
        a[0] += float(r["Counter_Value"]); a[1] += 1
print("SQ counters of the hand-written kernels inside the training step (python3 bench.py --no-cpu-baseline --no-analysis --steps 2 --warmup 2,")
print("one rocprofv3 --pmc pass; mean per launch; LDS_BANK_CONFLICT and ACTIVE_INST_LDS in the same units, so their ratio is the share of")
print("LDS-instruction time lost to conflicts)")
print("%-34s %8s %16s %16s %10s %12s" % ("kernel", "launches", "LDS_BANK_CONFLICT", "ACTIVE_INST_LDS", "conflict %", "wait / wave"))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    n = max(x[1] for x in v.values())
    c = v.get("SQ_LDS_BANK_CONFLICT", [0, 1]); l = v.get("SQ_ACTIVE_INST_LDS", [0, 1]); w = v.get("SQ_WAVE_CYCLES", [0, 1]); wa = v.get("SQ_WAIT_ANY", [0, 1])
    print("%-34s %8d %16.0f %16.0f %10.1f %12.3f" % (k, n, c[0] / max(c[1], 1), l[0] / max(l[1], 1), 100.0 * c[0] / max(l[0], 1), wa[0] / max(w[0], 1)))
PY
cat $O/sq_in_step.txt
