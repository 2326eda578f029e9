"""Library GEMM solution choices for this model's shapes on gfx950.

`grit_amd/tunableop_gfx950.csv` holds, per GEMM signature (layouts, M, N, K, dtypes), the hipBLASLt / rocBLAS solution that was
fastest on an MI355X (PyTorch TunableOp, tuned offline by tools/tune_gemms.py: training step at batch 32 and inference at batch 64,
640 x 640).  `load_tuned_gemms()` makes torch read it; nothing is tuned at run time.  The library's default heuristics are up to
1.8x slower on the short-K, long-M GEMMs of the Swin stages (171 vs 93 us for 51 200 x 2 048 x 512).  torch validates the file
against its PyTorch / ROCm / hipBLASLt versions and the gfx arch: on any mismatch it is ignored and the defaults run.
Every entry point calls this once per process (train_caption.main, inference_caption, bench.py); GRIT_TUNED_GEMMS=0 skips it."""
import os
import sys

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tunableop_gfx950.csv")
_state = {"loaded": None}


def load_tuned_gemms(path=TABLE):
    """True when the table was read (idempotent).  Never raises: a problem with the table must not stop training."""
    if _state["loaded"] is not None:
        return _state["loaded"]
    ok = False
    if os.environ.get("GRIT_TUNED_GEMMS", "1") == "1" and os.path.exists(path):
        try:
            import torch
            if torch.cuda.is_available():
                import torch.cuda.tunable as tunable
                tunable.enable(True)
                tunable.tuning_enable(False)
                if hasattr(tunable, "record_untuned_enable"):
                    tunable.record_untuned_enable(False)
                if hasattr(tunable, "write_file_on_exit"):
                    tunable.write_file_on_exit(False)
                tunable.set_filename(os.path.join("/tmp", "grit_tunableop_scratch_%d.csv" % os.getpid()))
                ok = bool(tunable.read_file(path))
        except Exception as e:
            print("[grit_amd] tuned GEMM table not loaded: %s" % e, file=sys.stderr)
    _state["loaded"] = ok
    return ok
