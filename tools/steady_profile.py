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
    import os
    if os.environ.get("GRIT_PROFILE_STEP_SEQUENCE"):  # the ordered kernel list of ONE whole step (first msda_fwd to the next step's)
        a, b = steps[-2][0], steps[-1][0]
        with open(os.environ["GRIT_PROFILE_STEP_SEQUENCE"], "w") as fh:
            for r in rows:
                s0 = int(r["Start_Timestamp"])
                if a <= s0 < b:
                    fh.write("%9.1f us  +%7.1f  %s\n" % ((s0 - a) / 1e3, (int(r["End_Timestamp"]) - s0) / 1e3, r["Kernel_Name"][:110]))
    # the decoder phase of a step: from the first MSDeformAttn forward launch (decoder layer 0) to the end of the last MSDeformAttn
    # backward work (decoder layer 0 again): deformable decoder forward after its first sampling, grid net, caption decoder, loss,
    # and their backward passes -- thousands of small dependent kernels.  Busy time inside it, kernel count, and how much of the
    # busy time belongs to kernels shorter than 20 us.
    win_rows = [r for r in rows if start <= int(r["Start_Timestamp"]) < end]
    spans, cur_first, last_bwd = [], None, None
    for r in win_rows:
        nm, s0, e0 = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "msda_fwd" in nm and (cur_first is None or s0 - cur_first[0] > 30e6):
            if cur_first is not None and last_bwd is not None:
                spans.append((cur_first[0], last_bwd))
            cur_first, last_bwd = (s0,), None
        if "msda_bwd" in nm or "msda_stage_flush" in nm:
            last_bwd = e0
    if cur_first is not None and last_bwd is not None:
        spans.append((cur_first[0], last_bwd))
    if spans:
        tot_span = tot_busy = tot_small = n_k = 0
        for a, b in spans:
            ks = [r for r in win_rows if a <= int(r["Start_Timestamp"]) < b]
            tot_span += b - a
            tot_busy += sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks)
            tot_small += sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ks
                             if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < 20000)
            n_k += len(ks)
        n = len(spans)
        print(f"\ndecoder phase (first msda_fwd .. last msda_bwd), mean of {n}: span {tot_span / n / 1e6:.2f} ms, GPU busy "
              f"{tot_busy / n / 1e6:.2f} ms in {n_k / n:.0f} kernels, of which {tot_small / n / 1e6:.2f} ms in kernels < 20 us")
        import os
        if os.environ.get("GRIT_PROFILE_SEQUENCE"):  # the ordered kernel list of ONE decoder phase (what runs after what, how long)
            a, b = spans[-1]
            with open(os.environ["GRIT_PROFILE_SEQUENCE"], "w") as fh:
                for r in win_rows:
                    s0 = int(r["Start_Timestamp"])
                    if a <= s0 < b:
                        fh.write("%9.1f us  +%7.1f  %s\n" % ((s0 - a) / 1e3, (int(r["End_Timestamp"]) - s0) / 1e3, r["Kernel_Name"][:110]))
        inside = collections.defaultdict(lambda: [0, 0])
        for a, b in spans:
            for r in win_rows:
                if a <= int(r["Start_Timestamp"]) < b:
                    e = inside[r["Kernel_Name"]]
                    e[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                    e[1] += 1
        for k, v in sorted(inside.items(), key=lambda kv: -kv[1][1])[:40]:
            print(f"    in phase: calls/step {v[1] / n:6.1f}  {v[0] / n / 1e6:6.3f} ms/step  avg {v[0] / v[1] / 1e3:7.1f} us  {k[:150]}")
    # where the GPU waits for the host: idle gaps between consecutive kernels inside the window
    win = [r for r in rows if start <= int(r["Start_Timestamp"]) < end]
    gaps, busy_until = [], int(win[0]["End_Timestamp"])
    for prev, cur in zip(win, win[1:]):
        busy_until = max(busy_until, int(prev["End_Timestamp"]))
        g = int(cur["Start_Timestamp"]) - busy_until
        if g > 0:
            gaps.append((g, int(cur["Start_Timestamp"]) - start, prev["Kernel_Name"][:70], cur["Kernel_Name"][:70]))
    tot_gap = sum(g[0] for g in gaps)
    print(f"\nidle: {tot_gap / 1e6 / nwin:.2f} ms/step in {len(gaps) / nwin:.0f} gaps; gaps > 20 us: "
          f"{sum(g[0] for g in gaps if g[0] > 20e3) / 1e6 / nwin:.2f} ms/step; > 100 us: "
          f"{sum(g[0] for g in gaps if g[0] > 100e3) / 1e6 / nwin:.2f} ms/step")
    # idle time per 5-ms slice of the step (position inside the window modulo the step length)
    step_len = (end - start) / nwin
    slices = collections.defaultdict(float)
    for g, pos, _, _ in gaps:
        slices[int((pos % step_len) / 5e6)] += g
    print("idle per 5-ms slice of the step [ms/step]: " + " ".join(f"{slices[i] / 1e6 / nwin:.2f}" for i in range(int(step_len / 5e6) + 1)))
    for g, pos, a, b in sorted(gaps, reverse=True)[:25]:
        print(f"  gap {g / 1e3:8.1f} us at +{(pos % step_len) / 1e6:6.2f} ms   after {a}   before {b}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2)
