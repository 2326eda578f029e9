"""bench.py's stand-alone MSDeformAttn-forward roofline measurement (config-2 point distribution, rotated value maps) alone."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
torch.cuda.set_device(0)
r = bench.msda_spread_microbench(torch.device("cuda", 0), iters=60)
for k, v in r.items():
    print(k, "avg_us", round(v["avg_launch_us"], 2), "median_us", round(v["median_launch_us"], 2), "frac", round(v["frac"], 3),
          "MB", round(v["algorithmic_bytes_per_launch"] / 1e6, 1))
