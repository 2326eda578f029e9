"""List (default) or tune (--tune) the GEMM shapes of the training step that grit_amd/tunableop_gfx950.csv does not cover.

    python tools/tune_gemms.py            # records untuned shapes to gpurun_out/tunableop_untuned.csv
    python tools/tune_gemms.py --tune     # tunes them (PyTorch TunableOp) and writes the merged table to gpurun_out/
    python tools/tune_gemms.py --inference [--tune]      # same for captioning at batch 64 (detector forward + beam search)

The merged table is copied over grit_amd/tunableop_gfx950.csv by hand after review; bench.py only ever reads it."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tune", action="store_true")
    ap.add_argument("--max-ms", type=int, default=40)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--fresh", action="store_true", help="ignore the shipped table: tune every shape from scratch")
    ap.add_argument("--inference", action="store_true", help="the shapes of captioning at batch 64 (detector forward + eager beam "
                    "search, bf16 weights) instead of the training step")
    args = ap.parse_args()
    import torch.cuda.tunable as tunable
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    tunable.enable(True)
    tunable.tuning_enable(bool(args.tune))
    tunable.set_max_tuning_duration(args.max_ms)
    tunable.set_max_tuning_iterations(args.iters)
    tunable.set_filename(os.path.join(out_dir, "tunableop_merged.csv"))
    if hasattr(tunable, "write_file_on_exit"):
        tunable.write_file_on_exit(bool(args.tune))
    if not args.fresh:
        tunable.read_file(os.path.join(ROOT, "grit_amd", "tunableop_gfx950.csv"))
    if not args.tune:
        os.environ["PYTORCH_TUNABLEOP_UNTUNED_FILENAME"] = os.path.join(out_dir, "tunableop_untuned.csv")
        tunable.record_untuned_enable(True)
    if args.inference:
        os.environ["GRIT_GRAPH_DECODE"] = "0"  # tuning launches candidates: not inside a graph capture
        from grit_amd.config import default_config
        from grit_amd.data import synthetic_batch
        from grit_amd.models.caption import Transformer
        from grit_amd.models.caption.detector import build_detector
        cfg = default_config()
        torch.manual_seed(0)
        model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
        batch = synthetic_batch(64, 640, 640, device="cuda", seed=1)
        with torch.no_grad():
            for _ in range(2):
                model(batch['samples'], seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
        torch.cuda.synchronize()
        if args.tune and hasattr(tunable, "write_file"):
            tunable.write_file()
        print("results:", len(tunable.get_results()))
        return
    import bench
    from grit_amd.amp import Bf16Compute
    from grit_amd.config import default_config
    from grit_amd.data import synthetic_batch
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    device = torch.device("cuda", 0)
    config = default_config()
    model = bench.build(device, config).train()
    wrapped = Bf16Compute(model, bucket_mb=64)
    opts = build_optimizers(wrapped, config, mode="xe")
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    batch = synthetic_batch(32, 640, 640, 20, device=device, seed=0)
    for _ in range(2):
        train_xe_step(wrapped, batch, opts, loss_fn)
    torch.cuda.synchronize()
    if args.tune and hasattr(tunable, "write_file"):
        tunable.write_file()
    print("results:", len(tunable.get_results()))


if __name__ == "__main__":
    main()
