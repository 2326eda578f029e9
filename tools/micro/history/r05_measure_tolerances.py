"""Measured noise behind the config-3 / config-4 / DropPath-on tolerances (VERDICT r04 'weak' 3: tighten to measured noise x 2): the bf16
step against the fp32-kernel step (per picked tensor: relative L2 distance of the gradients; relative loss distance), three repetitions of the
bf16 run to see its own run-to-run floor (MSDeformAttn backward sums in LDS-counter order), at 16 and 32 images."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_configs_gpu import PICKS, _config4_worker, _free_port, _run  # noqa: E402


def rel(a, b):
    return float(torch.linalg.norm(a - b) / torch.linalg.norm(b))


def main():
    out = {}
    for n_img in (16, 32):
        fp32 = _run(_config4_worker, _free_port(), 'fp32', n_img)
        runs = [_run(_config4_worker, _free_port(), 'plain', n_img) for _ in range(3)]
        out[str(n_img)] = {
            "loss_rel_vs_fp32": [abs(r["loss"] - fp32["loss"]) / fp32["loss"] for r in runs],
            "grad_rel_vs_fp32": {n: [rel(r["grads"][n], fp32["grads"][n]) for r in runs] for n in PICKS},
            "grad_rel_run_to_run": {n: [rel(runs[i]["grads"][n], runs[0]["grads"][n]) for i in (1, 2)] for n in PICKS},
        }
        print(json.dumps({n_img: out[str(n_img)]}), flush=True)
    with open("gpurun_out/r05_tolerances.json", "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
