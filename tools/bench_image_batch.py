"""Device image pipeline: time per batch (HIP events) and bytes moved, COCO-like sizes -> 384x640 (maxwh) or a fixed
640x640 batch.  `python tools/bench_image_batch.py [--batch 32] [--iters 50]`"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from grit_amd.datasets.caption.transforms import MaxWHResize  # noqa: E402
from grit_amd.ops.image_batch import image_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=32)
    ap.add_argument('--iters', type=int, default=50)
    a = ap.parse_args()
    rng = np.random.default_rng(0)
    coco = [(480, 640), (427, 640), (640, 480), (333, 500), (375, 500), (500, 375), (640, 428), (480, 640)]
    for name, shapes, policy in (('coco->maxwh(384,640)', [coco[i % 8] for i in range(a.batch)], MaxWHResize((384, 640))),
                                 ('1280x1280->640x640', [(1280, 1280)] * a.batch, MaxWHResize((640, 640)))):
        host = [torch.from_numpy(rng.integers(0, 256, s + (3,), dtype=np.uint8)) for s in shapes]
        dev = [h.cuda() for h in host]
        sizes = [policy.output_size(*s) for s in shapes]
        H, W = max(s[0] for s in sizes), max(s[1] for s in sizes)
        src = sum(3 * h * w for h, w in shapes)
        tmp = sum(3 * h * s[1] for (h, w), s in zip(shapes, sizes))
        algo = src + a.batch * H * W * 13
        pinned = [h.pin_memory() for h in host]
        for feed, label in ((dev, 'device-resident'), (pinned, 'host, pinned by the producer'), (host, 'host, pageable (staged)')):
            for _ in range(3):
                image_batch(feed, sizes, device='cuda')
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            for _ in range(a.iters):
                image_batch(feed, sizes, device='cuda')
            e1.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / a.iters
            ms = e0.elapsed_time(e1) / a.iters
            print(json.dumps({'case': name, 'input': label, 'batch': a.batch, 'out': [H, W], 'ms_per_batch': round(ms, 3),
                              'wall_ms_per_batch': round(wall * 1e3, 3), 'images_per_s': round(a.batch / wall),
                              'algorithmic_MB': round(algo / 1e6, 1), 'scratch_MB': round(2 * tmp / 1e6, 1),
                              'GB_per_s_algorithmic': round(algo / ms / 1e6, 1)}))


if __name__ == '__main__':
    main()
