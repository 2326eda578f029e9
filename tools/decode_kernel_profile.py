"""Per-kernel time of ONE replay of the beam-search graph, from a `rocprofv3 --kernel-trace --output-format csv` trace of
`tools/bench_decode.py --bf16 --decode-only N`: the trace is cut into bursts at idle gaps (the host synchronises and prints
between decodes) and the last burst is summarised.

    python tools/decode_kernel_profile.py <dir with *_kernel_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    files = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    bursts, cur = [], [rows[0]]
    for a, b in zip(rows, rows[1:]):
        if b[0] - a[1] > 300_000:
            bursts.append(cur)
            cur = []
        cur.append(b)
    bursts.append(cur)
    last = bursts[-1]
    wall = (last[-1][1] - last[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in last) / 1e6
    print("bursts %d; last burst: %d kernels, wall %.2f ms, kernel time %.2f ms, idle between kernels %.2f ms" %
          (len(bursts), len(last), wall, busy, wall - busy))
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in last:
        agg[n][0] += e - s
        agg[n][1] += 1
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:50]:
        print("%7.3f ms %6d calls %8.1f us  %s" % (t / 1e6, c, t / c / 1e3, n[:120]))


if __name__ == "__main__":
    main()
