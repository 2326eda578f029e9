"""Build the native pieces in-tree: libgrit_hip.so (hipcc, gfx950 only) and the C oracle (gcc).

`python -m grit_amd.build` or `__graft_entry__.build()`.  hipcc cross-compiles without a GPU.  The
shared objects stay next to their sources (git-ignored) so they travel with the tree to the GPU box.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "grit_amd", "csrc")
LIB = os.path.join(CSRC, "libgrit_hip.so")
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libmsda_oracle.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def hip_sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def build_hip(force=False, verbose=False):
    srcs = hip_sources()
    deps = srcs + [os.path.join(ROOT, "include", "grit_hip.h")] + \
        [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not force and _newer(LIB, deps) and os.path.lexists(os.path.join(CSRC, "libamdhip64.so.7")):
        return LIB
    # one object per source in parallel, then link
    objs, procs = [], []
    for s in srcs:
        o = s[:-4] + ".o"
        objs.append(o)
        if not force and _newer(o, deps):
            continue
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode()))
    # One HIP runtime per process, by construction: PyTorch-ROCm ships its own libamdhip64 (torch/lib, SONAME libamdhip64.so.7)
    # and two runtimes in one process do not share streams.  The library looks for its runtime next to itself first ($ORIGIN),
    # where `libamdhip64.so.7` is a link to torch's copy: whichever of torch / this library is loaded first, the loader ends up
    # with the same file (it de-duplicates by inode), so there is no import-order rule any more.
    _link_torch_hip_runtime()
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,$ORIGIN", "-Wl,--disable-new-dtags", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    return LIB


def _link_torch_hip_runtime():
    """csrc/libamdhip64.so.7 -> <torch>/lib/libamdhip64.so (when PyTorch-ROCm is installed; otherwise the system runtime is
    found through hipcc's default run path)."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        torch_lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so") if spec and spec.origin else None
    except Exception:
        torch_lib = None
    link = os.path.join(CSRC, "libamdhip64.so.7")
    if torch_lib and os.path.exists(torch_lib):
        if os.path.islink(link) or os.path.exists(link):
            if os.path.realpath(link) == os.path.realpath(torch_lib):
                return
            os.remove(link)
        os.symlink(torch_lib, link)


def build_oracle(force=False):
    """The checker (oracle/*.c).  Building it is not using it: only tests/smoke/bench's cpu leg load it."""
    for name in ("msda", "image"):
        src = os.path.join(ORACLE_DIR, name + "_oracle.c")
        lib = os.path.join(ORACLE_DIR, "lib%s_oracle.so" % name)
        if not force and _newer(lib, [src]):
            continue
        cmd = ["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-std=c99", "-o", lib, src, "-lm"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("gcc failed:\n" + r.stdout.decode())
    return ORACLE_LIB


def build_all(force=False, verbose=False):
    return build_hip(force, verbose), build_oracle(force)


if __name__ == "__main__":
    print(build_all(force="--force" in sys.argv, verbose=True))
