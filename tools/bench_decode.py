"""BASELINE config 5 on one GPU: beam-search (beam 5, 20 steps) captions/sec at batch 64 on synthetic 640x640 images.

    python tools/bench_decode.py [--batch 64] [--iters 3] [--bf16]

Reports the detector (backbone + deformable decoder + grid net) and the 20-step decode loop separately."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--bf16", action="store_true", help="bf16 weights (fp32 logits); default fp32 weights")
    ap.add_argument("--decode-only", type=int, default=0, help="run the detector once, then N decodes only (for rocprofv3)")
    ap.add_argument("--inference-mode", action="store_true", help="torch.inference_mode instead of torch.no_grad")
    a = ap.parse_args()
    from grit_amd.config import default_config
    from grit_amd.data import synthetic_batch
    from grit_amd.models.caption import Transformer
    from grit_amd.models.caption.detector import build_detector
    cfg = default_config()
    from grit_amd.tuning import load_tuned_gemms
    load_tuned_gemms()
    torch.manual_seed(0)
    model = Transformer(build_detector(cfg), cfg).cuda().eval()
    if a.bf16:
        model.to(torch.bfloat16)
    batch = synthetic_batch(a.batch, 640, 640, device="cuda", seed=1)

    def sync_time(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    mode = torch.inference_mode if a.inference_mode else torch.no_grad
    if a.decode_only:
        with mode():
            vis = model.detector(batch['samples'])
            model.cached_features = True
            for it in range(a.decode_only + 2):
                (tokens, _), t_dec = sync_time(lambda: model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3,
                                                              beam_size=5, out_size=1))
                print(json.dumps({"decode_20_steps_ms": t_dec * 1e3}))
        return
    with mode():
        for it in range(a.iters + 1):
            vis, t_det = sync_time(lambda: model.detector(batch['samples']))
            model.cached_features = True
            (tokens, _), t_dec = sync_time(lambda: model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3,
                                                          beam_size=5, out_size=1))
            model.cached_features = False
            if it:
                print(json.dumps({"batch": a.batch, "dtype": "bf16" if a.bf16 else "fp32", "detector_ms": t_det * 1e3,
                                  "decode_20_steps_ms": t_dec * 1e3, "captions_per_sec": a.batch / (t_det + t_dec)}))
    assert tokens.shape == (a.batch, 20)
    # pipelined: detector of batch i+1 on its own stream under the (launch-bound) beam search of batch i
    from inference_caption import caption_stream
    n = 6
    with torch.no_grad():
        list(caption_stream(model, [batch['samples']] * 2, cfg, 5))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = list(caption_stream(model, [batch['samples']] * n, cfg, 5))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    # same kernels on the same data as a sequential run WITHOUT the tuned GEMM table (caption_stream switches it off, see there)
    import torch.cuda.tunable as tunable
    was = tunable.is_enabled()
    tunable.enable(False)
    with torch.no_grad():
        ref_tokens = model(batch['samples'], seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)[0]
    tunable.enable(was)
    assert all(torch.equal(o[0], ref_tokens) for o in outs)
    print(json.dumps({"batch": a.batch, "dtype": "bf16" if a.bf16 else "fp32", "pipelined_batches": n,
                      "ms_per_batch": dt / n * 1e3, "captions_per_sec": a.batch * n / dt}))


if __name__ == "__main__":
    main()
