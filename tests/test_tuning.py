"""The shipped GEMM solution table (grit_amd/tunableop_gfx950.csv) and its loader."""
import csv

from grit_amd import tuning


def test_table_is_well_formed():
    rows = list(csv.reader(open(tuning.TABLE)))
    validators = [r for r in rows if r[0] == "Validator"]
    entries = [r for r in rows if r[0] != "Validator"]
    assert {v[1] for v in validators} >= {"PT_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME"}
    assert any("gfx950" in v[2] for v in validators)
    assert len(entries) > 100
    keys = set()
    for r in entries:
        assert len(r) == 4 and r[0].endswith(("_TN", "_NT", "_NN", "_TT")) and float(r[3]) > 0, r
        assert (r[0], r[1]) not in keys, ("duplicate signature", r)
        keys.add((r[0], r[1]))


def test_loader_is_a_no_op_without_a_gpu_and_never_raises(monkeypatch):
    import torch
    monkeypatch.setitem(tuning._state, "loaded", None)
    got = tuning.load_tuned_gemms()
    assert got is (False if not torch.cuda.is_available() else got)
    assert tuning.load_tuned_gemms() is got  # idempotent
    monkeypatch.setitem(tuning._state, "loaded", None)
    assert tuning.load_tuned_gemms("/nonexistent/table.csv") is False
