"""Intrinsic GPU dispatch gaps of the training step: analyse a rocprofv3 kernel trace of tools/timeline.py.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 tools/timeline.py --steps 3
    python tools/primed_gaps.py OUT

In timeline.py's 'primed' steps a ~250 ms spin kernel runs first, so the host has queued the whole step before the GPU
starts it: every gap between consecutive kernels of such a step is GPU-side (command processor, barriers, cache
write-back), not host launch latency."""
import collections
import csv
import glob
import sys


def main(out_dir):
    f = glob.glob(out_dir + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    spins = [i for i, r in enumerate(rows) if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) > 100e6]
    if not spins:
        print("no spin kernels found")
        return
    tot = collections.Counter()
    per_kernel = collections.defaultdict(lambda: [0, 0, 0])  # gap-before sum, count, duration sum
    nsteps = 0
    for a, b in zip(spins, spins[1:] + [len(rows)]):
        seg = rows[a + 1:b]
        if len(seg) < 500:
            continue
        # the step ends where the trace goes quiet for > 5 ms (synchronize + next step's setup)
        end = len(seg)
        for i in range(1, len(seg)):
            if int(seg[i]["Start_Timestamp"]) - int(seg[i - 1]["End_Timestamp"]) > 5e6:
                end = i
                break
        seg = seg[:end]
        nsteps += 1
        busy_until = int(seg[0]["End_Timestamp"])
        tot["busy"] += int(seg[0]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
        for prev, cur in zip(seg, seg[1:]):
            busy_until = max(busy_until, int(prev["End_Timestamp"]))
            s, e = int(cur["Start_Timestamp"]), int(cur["End_Timestamp"])
            g = max(0, s - busy_until)
            tot["gap"] += g
            tot["busy"] += e - s
            tot["n"] += 1
            k = per_kernel[cur["Kernel_Name"][:90]]
            k[0] += g; k[1] += 1; k[2] += e - s
            tot["gap_lt2"] += g if g < 2e3 else 0
            tot["gap_2_5"] += g if 2e3 <= g < 5e3 else 0
            tot["gap_5_20"] += g if 5e3 <= g < 20e3 else 0
            tot["gap_gt20"] += g if g >= 20e3 else 0
        tot["span"] += int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])
    n = max(nsteps, 1)
    print(f"primed steps: {nsteps}; per step: {tot['n'] / n:.0f} kernels, span {tot['span'] / 1e6 / n:.2f} ms, "
          f"kernel time {tot['busy'] / 1e6 / n:.2f} ms, gaps {tot['gap'] / 1e6 / n:.2f} ms "
          f"(<2us {tot['gap_lt2'] / 1e6 / n:.2f}, 2-5us {tot['gap_2_5'] / 1e6 / n:.2f}, 5-20us {tot['gap_5_20'] / 1e6 / n:.2f}, "
          f">20us {tot['gap_gt20'] / 1e6 / n:.2f})")
    print("gap before kernel, by kernel (ms/step, launches/step, avg gap us, avg duration us):")
    for name, (g, c, d) in sorted(per_kernel.items(), key=lambda kv: -kv[1][0])[:40]:
        print(f"  {g / 1e6 / n:7.3f}  x{c / n:7.1f}  gap {g / c / 1e3:6.2f}  dur {d / c / 1e3:8.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
