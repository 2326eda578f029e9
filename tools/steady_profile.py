"""Per-step kernel breakdown of the steady state from a rocprofv3 --kernel-trace CSV of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 bench.py --steps 4 --warmup 3 --no-cpu-baseline
    python tools/steady_profile.py OUT [n_steps_in_window=2] > profiles/rNN/steady.txt

Steps are delimited by the first MSDeformAttn forward launch of each step (6 per step), which skips the warm-up
(MIOpen / hipBLASLt first-call work) that a whole-run --stats summary mixes in."""
import collections
import csv
import glob
import sys


def main(out_dir, nwin=2):
    f = glob.glob(out_dir + "/*/*_kernel_trace.csv")[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ms = [int(r["Start_Timestamp"]) for r in rows if "msda_fwd" in r["Kernel_Name"]]
    steps = []
    for t in ms:
        if not steps or t - steps[-1][-1] > 30e6:
            steps.append([t])
        else:
            steps[-1].append(t)
    start, end = steps[-1 - nwin][0], steps[-1][0]
    agg = collections.defaultdict(lambda: [0, 0])
    for r in rows:
        s = int(r["Start_Timestamp"])
        if start <= s < end:
            a = agg[r["Kernel_Name"]]
            a[0] += int(r["End_Timestamp"]) - s
            a[1] += 1
    tot = sum(v[0] for v in agg.values())
    gemm = sum(v[0] for k, v in agg.items() if "Cijk" in k)
    print(f"steps seen {len(steps)}; window {nwin} steps: wall {(end - start) / 1e6 / nwin:.2f} ms/step, "
          f"GPU busy {tot / 1e6 / nwin:.2f} ms/step, GEMM (hipBLASLt/rocBLAS) {gemm / 1e6 / nwin:.2f} ms/step")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:60]:
        print(f"{v[0] / 1e6 / nwin:8.2f} ms/step  calls/step {v[1] / nwin:7.1f}  avg {v[0] / v[1] / 1e3:9.1f} us  {k[:140]}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
