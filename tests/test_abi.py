"""The C-ABI library builds, loads without a GPU and exports exactly what include/grit_hip.h declares."""
import ctypes
import os
import re

from grit_amd import build, lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "grit_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(grit_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported_and_bound():
    build.build_hip()
    names = _declared()
    assert "grit_msda_fwd_f32" in names and "grit_msda_bwd_f32" in names
    dll = ctypes.CDLL(lib.LIB_PATH)
    for n in names:
        assert hasattr(dll, n), "missing export " + n
        assert n in lib.SIGNATURES, "no ctypes signature for " + n
    assert sorted(lib.SIGNATURES) == names, "binding declares symbols the header does not"


def test_load_and_status_strings():
    l = lib.load()
    assert l.grit_abi_version() == lib.ABI_VERSION
    assert l.grit_status_string(0) == b"ok"
    assert b"launch" in l.grit_status_string(3)


def test_header_cites_reference_interfaces():
    text = open(os.path.join(ROOT, "include", "grit_hip.h")).read()
    for cite in ("ms_deform_attn.h:20", "vision.cpp", "swin_model.py:155", "attention.py:51"):
        assert cite in text


def test_cpu_tensors_are_rejected_like_the_reference():
    import pytest
    import torch
    from grit_amd.ops.msda import ms_deform_attn_forward
    shapes = torch.tensor([[2, 2]]); lsi = torch.tensor([0])
    with pytest.raises(RuntimeError, match="Not implemented on the CPU"):
        ms_deform_attn_forward(torch.zeros(1, 4, 1, 4), shapes, lsi, torch.zeros(1, 1, 1, 1, 1, 2),
                               torch.zeros(1, 1, 1, 1, 1))


def test_one_hip_runtime_whatever_is_loaded_first():
    """libgrit_hip.so loaded BEFORE torch must end up on torch's HIP runtime (two runtimes in one process do not share streams):
    the library finds `libamdhip64.so.7` next to itself, a link to torch's copy made by the build."""
    import subprocess
    import sys
    from grit_amd import lib
    code = (
        "import ctypes\n"
        "l = ctypes.CDLL(%r)\n"
        "assert l.grit_abi_version() == %d\n"
        "import torch, os\n"
        "maps = open('/proc/self/maps').read()\n"
        "libs = sorted({x.split()[-1] for x in maps.split('\\n') if 'libamdhip64' in x})\n"
        "print(len(libs), os.path.realpath(libs[0]).startswith(os.path.dirname(torch.__file__)))\n" % (lib.LIB_PATH, lib.ABI_VERSION))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["1", "True"], out.stdout
