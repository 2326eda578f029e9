"""The contract command `python bench.py --gpus N` must start N ranks by itself (the flag used to be ignored).
CPU-side checks of the launcher: gloo rendezvous on 127.0.0.1 with the training step mocked (GRIT_BENCH_MOCK=1), refusal when
fewer devices than ranks are visible, refusal when --gpus disagrees with WORLD_SIZE."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(kw)
    return env


def test_gpus_2_spawns_two_ranks_on_gloo():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env=_env(GRIT_BENCH_BACKEND="gloo", GRIT_BENCH_MOCK="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # rank 0 only
    # ... and nothing else: native libraries of the workers (RCCL's version banner) get stderr as their fd 1
    assert r.stdout.strip().splitlines() == lines, r.stdout
    out = json.loads(lines[0])
    assert out["mock"] is True and out["n_gpus"] == 2 and out["steps"] == 3
    devs = out["config"]["rank_devices"]
    assert [d["rank"] for d in devs] == [0, 1] and devs[0]["pid"] != devs[1]["pid"]
    assert out["config"]["ranks"] == 2


def test_refuses_more_ranks_than_devices():
    """No GPU here: the nccl (contract) backend with --gpus 8 must fail loudly, not print a 1-GPU line."""
    import torch
    if torch.cuda.device_count() >= 8:
        return
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 2
    assert "device(s) visible" in r.stderr and not any(l.startswith("{") for l in r.stdout.splitlines())


def test_refuses_world_size_mismatch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"], env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=2" in r.stderr


def test_native_stdout_of_a_worker_goes_to_stderr():
    """A C-level write to fd 1 inside a worker (what RCCL's banner is) must not land on the stdout the JSON line goes to."""
    code = ("import os, sys; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench; bench._keep_stdout_for_the_json_line(); "
            "os.write(1, b'native banner\\n'); print('{\"json\": 1}')" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == '{"json": 1}' and "native banner" in r.stderr
