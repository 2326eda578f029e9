# which op launches a given library kernel: its neighbours in the replayed step (kernel trace), tools/micro/r05_where_is_kernel.sh PATTERN...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/step_trace
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/step_trace -- python3 $R/bench.py --steps 4 --warmup 4 --no-cpu-baseline --no-analysis > /dev/null 2>&1
python3 - "$@" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/step_trace/*/*_kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
ms = [i for i, r in enumerate(rows) if 'msda_fwd' in r['Kernel_Name']]
# last full step: from the previous step's last msda_fwd group to the end
starts = [i for n, i in enumerate(ms) if n == 0 or int(rows[i]['Start_Timestamp']) - int(rows[ms[n - 1]]['Start_Timestamp']) > 30e6]
lo, hi = starts[-2], starts[-1]
t0 = int(rows[lo]['Start_Timestamp'])
for pat in sys.argv[1:]:
    print('=== ', pat)
    for i in range(lo, hi):
        if pat in rows[i]['Kernel_Name']:
            for j in range(max(lo, i - 3), min(hi, i + 3)):
                r = rows[j]
                print('%s %9.3f ms +%7.1f us  %s' % ('->' if j == i else '  ', (int(r['Start_Timestamp']) - t0) / 1e6, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:100]))
            print()
PY
