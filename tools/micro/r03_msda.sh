#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_msda_gpu.py tests/test_det_rows.py -q -m gpu -x 2>&1 | tail -4
timeout 600 python tools/micro/bench_msda_bwd.py 2>&1 | grep -v amdgpu.ids | tee $O/msda_bwd_methods.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $O/msda_trace
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/msda_trace -- python3 $R/tools/micro/bench_msda_bwd.py > /dev/null 2>&1
f=$(find $O/msda_trace -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee -a $O/msda_bwd_methods.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "msda" in r["Name"]:
        print("%-70s calls %5s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
rm -rf $O/msda_trace
