# SQ counters of the window-attention kernels: wave cycles, wait cycles, busy cycles, MFMA busy, VALU / LDS instruction
# counts -- DMA-staged kernels (default) and the register-staged ones (GRIT_WINATTN_{FWD,BWD}_DMA=0).  One counter set per pass.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in dma reg; do
  if [ $mode = reg ]; then export GRIT_WINATTN_FWD_DMA=0 GRIT_WINATTN_BWD_DMA=0; fi
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf /tmp/sq_${mode}_$tag
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d /tmp/sq_${mode}_$tag -- python3 $R/tools/bench_kernels.py winattn --iters 4 > /dev/null 2>&1
    echo "$mode $tag rc=$?"
  done
done
python3 - <<'PY' > $O/winattn_sq_counters.txt
import csv, glob, re
print("SQ counters of the window-attention kernels, summed over the launches of `tools/bench_kernels.py winattn --iters 4` (all four Swin stages,")
print("shift 0 and 6; rocprofv3 --pmc, one counter set per pass; quad-cycle units for the cycle counters, see MI355X_MICROARCH.md), per launch:")
for mode, label in (("dma", "DMA-staged (default)"), ("reg", "register-staged (GRIT_WINATTN_*_DMA=0)")):
    tot = {}
    for f in glob.glob("/tmp/sq_%s_*/*/*_counter_collection.csv" % mode):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(winattn_fwd_dma|winattn_bwd_dma|winattn_fwd|winattn_bwd)", r["Kernel_Name"])
            if not m: continue
            a = tot.setdefault(m.group(1), {}).setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
    print("\n" + label)
    for k, v in sorted(tot.items()):
        print("  " + k)
        for c in sorted(v):
            print("    %-28s %16.0f  (%d launches)" % (c, v[c][0] / v[c][1], v[c][1]))
        if "SQ_WAVE_CYCLES" in v and "SQ_WAIT_ANY" in v:
            print("    wait / wave cycles           %16.3f" % (v["SQ_WAIT_ANY"][0] / v["SQ_WAVE_CYCLES"][0]))
        if "SQ_BUSY_CYCLES" in v and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            print("    MFMA busy / busy cycles      %16.3f" % (v["SQ_VALU_MFMA_BUSY_CYCLES"][0] / v["SQ_BUSY_CYCLES"][0]))
PY
cat $O/winattn_sq_counters.txt | head -70
