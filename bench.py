"""Contract benchmark: GRIT cross-entropy training throughput on synthetic 640x640 batches.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: before any GPU call it checks that N
devices are visible (exit code 2 otherwise -- it never prints an `n_gpus: 1` line for `--gpus 8`), starts N fresh worker processes
of this file (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT in their environment, one per GPU), waits
for them and exits with the first non-zero worker code.  Under torchrun (WORLD_SIZE set) the process IS a worker; `--gpus` must
then equal WORLD_SIZE.

Workload (BASELINE.json metric, configs[3] per GPU): full GRIT (Swin-B window 12 + 6 deformable decoder layers +
3-layer grid net + 3-layer caption decoder, 161 M parameters, random init), 32 images of 3x640x640 per GPU,
captions of 20 tokens, train mode (dropout / DropPath on), bf16 compute copies over fp32 master weights
(grit_amd.amp.Bf16Compute; softmax / LayerNorm statistics / sampling locations / logits / loss in fp32), one step =
forward + backward + gradient all-reduce (RCCL, bucketed, overlapped with backward) + two fused Adam steps, exactly
the order of reference engine/caption_engine.py:312-350.  Weak scaling: the per-GPU batch is fixed.

The step runs as ONE captured HIP graph on one rank (grit_amd/engine/graph_step.py; GRIT_STEP_GRAPH=0 = eager launches; N > 1 runs
eager launches with the bucketed RCCL all-reduce): two eager warm-up steps, capture, every later step is a copy of the batch into the
graph's input buffers + one graph launch.  `config.step_graph` says which one ran.

One JSON line on rank 0.  Besides the contract keys:
  roofline      the hand-written kernel FAMILY with the most ms/step among {gemm_nt_bf16 (fused Mlp GEMMs), wgrad_tn (weight gradients),
                gemm_w4 (persistent long-map GEMM), window-attention backward / forward}, chosen from the measured per-launch times of
                this run, against the larger of its two bounds (algorithmic flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s); both
                fractions are in it (`frac_mfma`, `frac_hbm`).  Launch times: HIP events on the launch stream in eager steps right
                behind the timed region (a replayed graph runs no Python to record events).  `traffic` = PMC bytes per launch of the
                same kernel from profiles/r04/pmc_in_step.txt where present.
  roofline_gemm_nt_bf16 / _gemm_w4 / _wgrad_tn   the three own GEMM families, same fields, always reported.
  gemm          EVERY GEMM family of the step incl. `gemm_lib` (hipBLASLt / rocBLAS launches): launches/step, ms/step, average launch,
                TFLOP/s + frac_mfma, GB/s + frac_hbm, bound, frac.
  config3_bs16  (N = 1 only) SURVEY 8(d) config 3: the same step at 16 images, img/s and ms/step over 10 steps.
  roofline_msda MSDeformAttn forward kernel (HBM-bound gather) as it runs INSIDE the step: algorithmic bytes per launch /
                average launch time measured with HIP events on the launch stream inside the timed region, against 8 TB/s.
                Algorithmic bytes = the distinct 128-byte lines of the value map that the launch's sampling points touch (counted
                from the recorded sampling locations after the timed region) + locations + weights + output, each once -- so the
                fraction cannot exceed 1 by construction (charging the whole map, as round 1 did, gave 1.29).  Both denominators
                are reported: `frac_touched` (= frac) and `frac_compulsory_8d` (SURVEY 8d's whole-value-map figure).
  roofline_msda_spread   the same kernel(s) stand-alone on SURVEY 8d config 2's point distribution (learned-offset-like spread
                points), value maps rotated through > 512 MB so the 256 MB Infinity Cache cannot serve them: bf16 B = 32 and the
                fp32 kernel at B = 8 (config 2 itself).
  roofline_winattn_bwd / _fwd   window attention against its compulsory HBM bytes (7 resp. 4 bf16 [144, 32] slices per window-head).
  cpu_baseline  (N = 1 only) the same training step on the host CPU: this repo's modules with the oracle ops
                (oracle/torch_ref.py) injected -- a port, not the reference -- on a bounded sample: batch 1
                over a thread sweep (8 / 16 / 32 / 64), headline = the best point, plus batch 4 there; budgeted at 60 s.
  decode_config5  (N = 1 only) BASELINE config 5 on one GPU after the timed region: beam-5 x 20-step captions/s at batch 64.
Diagnostic flags (recorded in config, never the default): --points spread (decoder sampling locations replaced by config 2's
distribution), --ragged (images of different sizes: the general padding-mask path), --fp32, GRIT_MSDA_BWD_F32ACC=0 (value gradient of the
deformable attention accumulated in bf16 by packed atomics instead of f32).
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_PEAK_BF16 = 2.5e15
MSDA_FWD_KERNEL = {"fwd": "msda_fwd_vec4<16,4> (MSDeformAttn forward, fp32 value map)",
                   "fwd_bf16": "msda_fwd_bf16_rows4<2> (MSDeformAttn forward, bf16 value map, fp32 sampling geometry)"}
# HBM-side bytes per launch come from the committed PMC summary of THIS command (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
# passes, FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md), read at run time: no figure is typed in here
PMC_SUMMARIES = ("profiles/r06/pmc_in_step.txt", "profiles/r05/pmc_in_step.txt", "profiles/r04/pmc_in_step.txt", "profiles/r03/pmc_in_step.txt", "profiles/r02/pmc_in_step.txt")


def pmc_traffic(kernel_substring):
    """(bytes per launch, source file) of the first PMC summary row whose kernel name contains `kernel_substring`; (None, None)
    when no summary has it.  Rows: `<kernel> launches N  FETCH_SIZE f  WRITE_SIZE w  HBM-side bytes/launch b`."""
    for rel in PMC_SUMMARIES:
        try:
            with open(os.path.join(ROOT, rel)) as f:
                for line in f:
                    if kernel_substring in line and "HBM-side bytes/launch" in line:
                        return float(line.rsplit("HBM-side bytes/launch", 1)[1].split()[0]), rel
        except OSError:
            continue
    return None, None


FLOP_PER_IMAGE_FWD_BWD = 955.8e9  # measured on the reference with torch.utils.flop_counter (640^2, T = 20)


def _baseline_metric():
    """The metric name exactly as BASELINE.json spells it."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except Exception:
        return "images/sec (train fwd+bwd) at 640\u00d7640 bs=32/GPU, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)  # >= 20 warm-up + >= 50 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU (metric is defined at 32)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--caption-len", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed steps of the cpu_baseline headline sample (median)")
    ap.add_argument("--fp32", action="store_true", help="diagnostic: no autocast (not the metric's dtype)")
    ap.add_argument("--points", choices=("model", "spread"), default="model",
                    help="spread: MSDeformAttn sampling locations replaced by SURVEY 8d config 2's distribution (diagnostic)")
    ap.add_argument("--ragged", action="store_true", help="diagnostic: images of different sizes (general padding-mask path)")
    ap.add_argument("--no-analysis", action="store_true", help="skip the post-run analysis steps / micro-benchmarks")
    return ap.parse_args()


def spread_locations(B, Lq, M, L, P, device, seed=0):
    """ref ~ U(0,1) per (b, q); loc = clamp(ref + 0.05 N(0,1), -0.05, 1.05) per sampling point."""
    g = torch.Generator(device=device).manual_seed(seed)
    ref = torch.rand(B, Lq, 1, 1, 1, 2, device=device, generator=g)
    return (ref + 0.05 * torch.randn(B, Lq, M, L, P, 2, device=device, generator=g)).clamp_(-0.05, 1.05)


class _SpreadOverride(object):
    """loc -> config-2 locations of the same shape (one fixed draw per shape)."""

    def __init__(self):
        self.cache = {}

    def __call__(self, loc):
        key = (tuple(loc.shape), loc.device)
        if key not in self.cache:
            B, Lq, M, L, P, _ = loc.shape
            self.cache[key] = spread_locations(B, Lq, M, L, P, loc.device)
        return self.cache[key]


def _event_us(pairs):
    return [a.elapsed_time(b) * 1e3 for a, b in pairs]


def msda_forward_bytes(cells, cell_bytes, B, Lq, M, D, L, P, out_esize):
    """Algorithmic bytes of one forward launch: distinct value cells touched + sampling locations (2 f32) + weights (1 f32)
    per point + the output rows."""
    return cells * cell_bytes + 3 * 4 * B * Lq * M * L * P + out_esize * B * Lq * M * D


def config3_child(args):
    """SURVEY 8(d) config 3 (full GRIT fwd+bwd, ONE GPU, 16 images, bf16): this file again as a CHILD process at --batch 16 (5 warm-up
    + 10 timed steps, graph step like the headline's), after this process released its training state.  A child because a second
    capture on a wrapper whose first graph was released dies inside hipStreamEndCapture on this ROCm (known limits)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--batch", "16", "--size", str(args.size), "--caption-len",
           str(args.caption_len), "--steps", "10", "--warmup", "5", "--no-analysis", "--no-cpu-baseline"]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300,
                           env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
        lines = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "child exited with %d: %s" % (r.returncode, r.stderr.decode()[-300:])}
        d = json.loads(lines[-1])
        return {"workload": "config 3: the same XE training step at 16 images of %dx%d on one GPU, bf16 (child process of this run)"
                            % (args.size, args.size),
                "images_per_sec": d["value"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"],
                "step_graph": d["config"]["step_graph"], "final_loss": d["final_loss"]}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def msda_spread_microbench(device, iters=30):
    """MSDeformAttn forward stand-alone on config 2's distribution with the value maps rotated so that consecutive launches
    cannot find their map in the Infinity Cache (total > 512 MB)."""
    from grit_amd.ops import msda as msda_op
    shapes = torch.tensor([[80, 80], [40, 40], [20, 20], [10, 10]], device=device)
    lsi = torch.cat((shapes.new_zeros(1), shapes.prod(1).cumsum(0)[:-1]))
    S, M, D, Lq, L, P = 8500, 8, 64, 150, 4, 4
    out = {}
    for name, B, dtype in (("bf16_B32", 32, torch.bfloat16), ("fp32_B8_config2", 8, torch.float32)):
        esize = 2 if dtype == torch.bfloat16 else 4
        nmaps = max(3, -(-(600 << 20) // (B * S * M * D * esize)))
        maps = [torch.randn(B, S, M, D, device=device, dtype=dtype) for _ in range(nmaps)]
        loc = spread_locations(B, Lq, M, L, P, device)
        aw = torch.softmax(torch.randn(B, Lq, M, L * P, device=device), -1).view(B, Lq, M, L, P)
        for i in range(3):
            msda_op.ms_deform_attn_forward(maps[i % nmaps], shapes, lsi, loc, aw)
        pairs = []
        for i in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            msda_op.ms_deform_attn_forward(maps[i % nmaps], shapes, lsi, loc, aw)
            b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        us = sorted(_event_us(pairs))
        avg_us = sum(us) / len(us)
        cells = msda_op.unique_lines_touched(loc, shapes.cpu(), lsi.cpu(), S, M)
        nbytes = msda_forward_bytes(cells, 64 * esize, B, Lq, M, D, L, P, esize)
        whole_map = esize * B * S * M * D
        out[name] = {"kernel": "msda_fwd_bf16_rows4<2>" if esize == 2 else "msda_fwd_vec4<16,4>", "bound": "hbm",
                     "avg_launch_us": avg_us, "median_launch_us": us[len(us) // 2], "launches": iters,
                     "rotating_value_maps": nmaps, "rotated_bytes": nmaps * whole_map,
                     "algorithmic_bytes_per_launch": int(nbytes), "value_cells_touched_frac": cells * 64 * esize / whole_map,
                     "achieved": nbytes / avg_us / 1e3, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": nbytes / avg_us / 1e3 / HBM_PEAK_GBPS}
        del maps
    return out


def build(device, config):
    from grit_amd.models.caption import Transformer
    from grit_amd.models.caption.detector import build_detector
    torch.manual_seed(config.exp.seed + int(os.environ.get('GRIT_BENCH_SEED', '0')))  # (GRIT_BENCH_SEED: diagnostic -- other weights / masks)
    model = Transformer(build_detector(config), config).to(device)
    model.cached_features = False
    return model


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo; os.cpu_count() when that cannot be read."""
    try:
        pairs, phys = set(), None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    pairs.add((phys, line.split(":")[1].strip()))
        return len(pairs) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_baseline(config, size, caption_len, steps, budget_s=60.0):
    """Bounded CPU sample of the same step: this repo's modules with the oracle ops injected (kind =
    'port'), fp32, dropout on.  Batch 1 at 8 / 16 / 32 / 64 threads (1 warm-up + 2 timed steps each), headline = the best point;
    batch 4 at that thread count; points are skipped once the leg has used most of `budget_s` seconds."""
    from grit_amd.data import synthetic_batch
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    from grit_amd.ops.backend import use_reference_ops
    from oracle import torch_ref  # checker / CPU baseline only
    logical, physical = os.cpu_count() or 1, _physical_cores()
    model = build(torch.device("cpu"), config).train()
    opts = build_optimizers(model, config, mode="xe")
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    t_leg = time.perf_counter()

    def sample(bs, threads, timed):
        torch.set_num_threads(threads)
        batch = synthetic_batch(bs, size, size, caption_len, device="cpu", seed=0)
        times = []
        with use_reference_ops(torch_ref):
            for i in range(timed + 1):
                if i > 1 and time.perf_counter() - t_leg > budget_s:
                    break
                t0 = time.perf_counter()
                train_xe_step(model, batch, opts, loss_fn)
                times.append(time.perf_counter() - t0)
        timed_steps = sorted(times[1:])
        if not timed_steps:
            return None
        med = timed_steps[len(timed_steps) // 2]
        return {"batch": bs, "threads": threads, "timed_steps": len(timed_steps), "median_s_per_step": med,
                "images_per_sec": bs / med}

    # Thread sweep at batch 1 (1 warm-up + 2 timed steps each): torch's CPU kernels do NOT scale with the cores of these hosts --
    # measured on a 128-core / 256-cpu box: 0.55 images/s with 8 threads, 0.21 with 64, 0.10 with 128 (profiles/r03/README.md) --
    # so the headline is the BEST point of the sweep (the baseline most favourable to the CPU), with its thread count in `cores`.
    # Round 6 (BASELINE.md section 3 asks for N = physical cores and a median of 5): the sweep takes ONE timed step per point and includes
    # the N = physical-cores point; the headline is then re-measured at the best thread count as the median of `steps` (5) timed steps.
    matrix = []
    for threads in sorted({min(8, logical), min(16, physical), min(32, physical), min(64, physical), physical}):
        if time.perf_counter() - t_leg < 0.6 * budget_s:
            r = sample(1, threads, 1)
            if r is not None:
                matrix.append(dict(r, role="sweep"))
    best = max(matrix, key=lambda r: r["images_per_sec"])
    head = sample(1, best["threads"], steps) if time.perf_counter() - t_leg < 0.8 * budget_s else None
    if head is None:
        head = best
    else:
        matrix.append(dict(head, role="headline"))
    if time.perf_counter() - t_leg < 0.9 * budget_s:
        r = sample(4, head["threads"], 1)
        if r is not None:
            matrix.append(dict(r, role="batch 4"))
    return {"value": head["images_per_sec"], "unit": "images/sec", "cores": head["threads"], "kind": "port",
            "sample": f"batch 1, {size}x{size}, T={caption_len}, fp32: thread sweep (8 / 16 / 32 / 64 / {physical} = physical cores; 1 warm-up + "
                      f"1 timed step each), then 1 warm-up + median of {head['timed_steps']} timed steps at the best thread count "
                      f"({head['threads']}) on a host with {physical} physical cores / {logical} cpus; `matrix` holds every point plus "
                      f"batch 4 at that thread count; the leg is budgeted at {budget_s:.0f} s (steps are dropped past it)",
            "matrix": matrix, "physical_cores": physical, "leg_seconds": time.perf_counter() - t_leg}


def decode_config5(device, config, batch=64, iters=2):
    """BASELINE config 5 on this GPU (the code path of tools/bench_decode.py): beam search, beam 5, 20 steps, batch 64 of
    synthetic 640x640 images, eval mode, bf16 weights with fp32 logits / log-softmax / beam arithmetic; detector and decode
    timed separately between device syncs, sequentially (no overlap between batches).  Tokens are hashed so that runs can be
    compared bit for bit."""
    import hashlib
    from grit_amd.data import synthetic_batch
    torch.manual_seed(config.exp.seed)
    model = build(device, config).eval().to(torch.bfloat16)
    data = synthetic_batch(batch, 640, 640, device=device, seed=1)

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        return out, time.perf_counter() - t0

    det, dec, tokens = [], [], None
    with torch.no_grad():
        for it in range(iters + 1):  # iteration 0 warms up and captures the decode graph
            vis, t_det = timed(lambda: model.detector(data['samples']))
            model.cached_features = True
            (tokens, _), t_dec = timed(lambda: model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5,
                                                     out_size=1))
            model.cached_features = False
            if it:
                det.append(t_det)
                dec.append(t_dec)
    t_det, t_dec = min(det), min(dec)
    sha1 = hashlib.sha1(tokens.cpu().numpy().tobytes()).hexdigest()
    try:  # the recorded hash of this exact workload (tests/golden/config5_tokens.json): a changed kernel that changes a token shows here
        with open(os.path.join(ROOT, "tests", "golden", "config5_tokens.json")) as f:
            fixture = json.load(f)
        match = bool(batch == 64 and fixture["tokens_sha1"] == sha1 and fixture["tokens_shape"] == list(tokens.shape))
    except Exception:
        match = None
    return {"workload": "beam search, beam 5, 20 steps, batch %d, synthetic 640x640, eval, bf16 weights / fp32 decoder tail" % batch,
            "captions_per_sec_sequential": batch / (t_det + t_dec), "detector_ms": t_det * 1e3, "decode_20_steps_ms": t_dec * 1e3,
            "batch": batch, "beam": 5, "iterations": iters, "n_gpus": 1,
            "tokens_sha1": sha1, "tokens_shape": list(tokens.shape), "tokens_match_fixture": match}


def _enable_tuned_gemms():
    """hipBLASLt / rocBLAS solution choices for this model's GEMM shapes, tuned once on an MI355X and shipped as data: the same
    call every entry point of the package makes (grit_amd/tuning.py).  No tuning happens inside the benchmark."""
    from grit_amd.tuning import load_tuned_gemms
    return load_tuned_gemms()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(args):
    """`--gpus N` without a launcher around it: start N worker processes of this file, one per GPU, and wait for them.
    Nothing here touches the GPU (torch.cuda.device_count() only counts devices): the workers are fresh processes, not re-execs
    of one that initialised HIP.  Exit code: 2 when fewer than N devices are visible (the contract backend needs one device per
    rank), otherwise the first non-zero worker code."""
    import subprocess
    backend = os.environ.get("GRIT_BENCH_BACKEND", "nccl")
    visible = torch.cuda.device_count()
    if backend == "nccl" and visible < args.gpus:
        sys.stderr.write("bench.py: --gpus %d asked for, %d device(s) visible: refusing to run (an RCCL rank needs its own GPU)\n"
                         % (args.gpus, visible))
        return 2
    port = _free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    code, left = 0, set(range(args.gpus))
    while left:
        for r in sorted(left):
            rc = procs[r].poll()
            if rc is None:
                continue
            left.discard(r)
            if rc != 0 and code == 0:
                code = rc
                for o in left:  # a dead rank leaves the others waiting in a collective: end exactly the processes started here
                    procs[o].terminate()
        time.sleep(0.05)
    return code


def mock_worker(args, rank, world, backend):
    """GRIT_BENCH_MOCK=1: the launcher / rendezvous / timing / JSON plumbing with the training step replaced by a sleep and a
    small all-reduce, so the N > 1 contract command can be tested where there is no GPU (tests/test_bench_launcher.py).  The
    line it prints is labelled as a mock and carries no throughput claim."""
    dist.init_process_group(backend, rank=rank, world_size=world)
    t = torch.zeros(4)
    for _ in range(args.warmup):
        dist.all_reduce(t)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.002)
        t += 1
        dist.all_reduce(t)
    dist.barrier()
    tmax = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    ids = [None] * world
    dist.all_gather_object(ids, {"rank": rank, "pid": os.getpid(), "device": "cpu"})
    if rank == 0:
        print(json.dumps({"metric": "MOCK (launcher plumbing test, no training step ran)", "mock": True, "value": None,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(tmax) / args.steps * 1e3,
                          "config": {"backend": backend, "ranks": dist.get_world_size(), "rank_devices": ids}}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0


def _keep_stdout_for_the_json_line():
    """RCCL prints a version banner with C stdio (seen: five lines that reach stdout when the process exits, i.e. AFTER the JSON
    line).  Whoever reads this program's stdout wants exactly one line: native code gets stderr as its fd 1, Python's sys.stdout
    keeps the real one."""
    sys.stdout.flush()
    real = os.dup(1)
    os.dup2(2, 1)
    sys.stdout = os.fdopen(real, "w", buffering=1)


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch(args)
    _keep_stdout_for_the_json_line()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the line would misreport n_gpus; refusing\n" % (args.gpus, world))
        return 2
    # GRIT_BENCH_BACKEND=gloo lets the N>1 code path be exercised on a 1-GPU box (ranks share the device); the
    # contract run uses 'nccl' (= RCCL over xGMI on ROCm), one rank per GPU
    backend = os.environ.get("GRIT_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if os.environ.get("GRIT_BENCH_MOCK") == "1":
        return mock_worker(args, rank, world, backend)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: a GPU is required"
    if backend == "nccl" and torch.cuda.device_count() < world:
        sys.stderr.write("bench.py: %d ranks but %d device(s) visible\n" % (world, torch.cuda.device_count()))
        return 2
    device = torch.device("cuda", local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    # RCCL channel budget: every channel is one workgroup (one CU) taken from the backward GEMMs while a collective is in flight.
    # GRIT_RCCL_MAX_CHANNELS=n caps it (RCCL reads NCCL_MAX_NCHANNELS at communicator creation); unset = RCCL's own choice.
    # Recorded in config.rccl_env together with every NCCL_* / RCCL_* variable the run saw.
    if os.environ.get("GRIT_RCCL_MAX_CHANNELS"):
        os.environ["NCCL_MAX_NCHANNELS"] = os.environ["GRIT_RCCL_MAX_CHANNELS"]
    rccl_env = {k: v for k, v in os.environ.items() if k.startswith(("NCCL_", "RCCL_"))}
    # GRIT_GRAD_SYNC=allreduce (default): bucketed all-reduce, every rank steps every master.  GRIT_GRAD_SYNC=shard: reduce-scatter,
    # every rank's FlatAdam steps its 1/N slice, the bf16 compute weights are all-gathered (grit_amd.amp shard_optimizer)
    grad_sync = os.environ.get("GRIT_GRAD_SYNC", "allreduce")
    if grad_sync not in ("allreduce", "shard"):
        sys.stderr.write("bench.py: GRIT_GRAD_SYNC must be allreduce or shard\n")
        return 2
    # GRIT_BENCH_SELF_COLLECTIVES=1 (N = 1 only): a one-rank process group whose gradient sync is still issued collective by
    # collective (grit_amd/ddp.py GRIT_DDP_SELF_COLLECTIVES) -- what the RCCL path costs on the one GPU a box has, wire excluded
    self_coll = world == 1 and os.environ.get("GRIT_BENCH_SELF_COLLECTIVES") == "1"
    if self_coll:
        os.environ["GRIT_DDP_SELF_COLLECTIVES"] = "1"
    if (world > 1 or self_coll) and not dist.is_initialized():
        dist.init_process_group(backend, rank=rank, world_size=world)
    rank_devices = [{"rank": rank, "device": device.index, "pid": os.getpid()}]
    if world > 1:
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, {"rank": rank, "device": device.index, "pid": os.getpid()})
    _enable_tuned_gemms()

    from grit_amd.config import default_config
    from grit_amd.data import synthetic_batch
    from grit_amd.amp import Bf16Compute
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.engine.caption_engine import build_optimizers, train_xe_step
    from grit_amd.ops import msda as msda_op
    from grit_amd.ops import profiling
    from grit_amd.ops import window_attention as wa_op

    config = default_config()
    model = build(device, config).train()
    if args.fp32:
        wrapped = BucketedDataParallel(model, bucket_mb=64)
    else:  # bf16 compute copies + fp32 master weights; gradients are produced, all-reduced and unscaled in flat bf16 buckets
        wrapped = Bf16Compute(model, bucket_mb=64, shard_optimizer=(grad_sync == "shard" and (world > 1 or self_coll)))
    optimizers = build_optimizers(wrapped, config, mode="xe")
    loss_fn = torch.nn.NLLLoss(ignore_index=1)
    # inputs resident in HBM before the timed region; 4 distinct batches per rank, cycled
    batches = [synthetic_batch(args.batch, args.size, args.size, args.caption_len, device=device, seed=1000 * rank + i,
                               ragged=args.ragged) for i in range(4)]
    if args.points == "spread":
        msda_op.LOC_OVERRIDE = _SpreadOverride()

    def eager_step(i):
        return train_xe_step(wrapped, batches[i % len(batches)], optimizers, loss_fn)

    # The step as ONE HIP graph (grit_amd/engine/graph_step.py; default on one rank, GRIT_STEP_GRAPH=0 = eager launches): the first
    # two warm-up steps run eagerly and a third pass is captured, every later step -- the rest of the warm-up and the whole timed
    # region -- is a device-to-device copy of the batch into the graph's input buffers, the optimizers' per-step scalars and one
    # hipGraphLaunch.  What is timed is the same work (forward, backward, gradient buckets, both Adam steps), enqueued differently.
    from grit_amd.engine import graph_step
    graphed, graph_error, graph_plan = None, None, None
    graph_reason = ("GRIT_STEP_GRAPH=0" if not graph_step.ENABLED else "--fp32" if args.fp32 else "--warmup 0" if args.warmup < 1
                    else graph_step.why_not(wrapped, optimizers))
    want_graph = graph_reason is None
    eager_warmup = min(2, args.warmup) if want_graph else args.warmup
    for i in range(eager_warmup):
        loss = eager_step(i)
    if want_graph:
        try:
            graphed = graph_step.GraphedXEStep(wrapped, optimizers, loss_fn, batches[eager_warmup % len(batches)], eager_steps=0)
            graph_plan = None if graphed.plan is None else [k for k, _ in graphed.plan]  # (release() below forgets it)
        except Exception as e:  # stay on the eager HIP path and say so in the line (config.step_graph_error)
            graph_error = "%s: %s" % (type(e).__name__, str(e)[:300])
            sys.stderr.write("bench.py: step graph not captured (%s); running eager launches\n" % graph_error)
            # (GraphedXEStep leaves nothing of the failed capture behind: graph_step.abandon_capture)

    def step(i):
        if graphed is not None:
            return graphed(batches[i % len(batches)])
        return eager_step(i)

    for i in range(eager_warmup, args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    msda_op.PROFILE_EVENTS = []
    wa_op.PROFILE_EVENTS = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(args.warmup + i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events, msda_op.PROFILE_EVENTS = msda_op.PROFILE_EVENTS, None
    wa_events, wa_op.PROFILE_EVENTS = wa_op.PROFILE_EVENTS, None
    final_loss = float(loss)
    if world > 1:
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    # ---- after the timed region: analysis steps (every rank takes them: the gradient all-reduce is collective) -----------
    geom_events, gemm_events = [], []
    event_steps = args.steps  # steps the per-launch events of the MSDA / window-attention kernels cover
    if graphed is not None:
        # a replayed graph runs no Python, so no per-launch HIP events were recorded inside the timed region: the same launches are
        # timed in eager steps right behind it (the graph is dropped first; the optimizers go back to launch-argument scalars)
        graphed.release()
        torch.cuda.synchronize()
        if not args.no_analysis:
            msda_op.PROFILE_EVENTS, wa_op.PROFILE_EVENTS = [], []
            event_steps = 3
            for i in range(event_steps):
                eager_step(args.warmup + args.steps + i)
            torch.cuda.synchronize()
            events, msda_op.PROFILE_EVENTS = msda_op.PROFILE_EVENTS, None
            wa_events, wa_op.PROFILE_EVENTS = wa_op.PROFILE_EVENTS, None
    if not args.no_analysis:
        msda_op.PROFILE_EVENTS, msda_op.PROFILE_RECORD_GEOMETRY, profiling.EVENTS = geom_events, True, gemm_events
        for i in range(2):
            eager_step(args.warmup + args.steps + 3 + i)
        torch.cuda.synchronize()
        msda_op.PROFILE_EVENTS, msda_op.PROFILE_RECORD_GEOMETRY, profiling.EVENTS = None, False, None

    # what an event pair reads with NOTHING between its two markers: the part of every per-launch figure below that is
    # marker / dispatch latency, not kernel time (reported, not subtracted: the roofline figures stay conservative)
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    for a, b in pairs:
        a.record()
        b.record()
    torch.cuda.synchronize()
    empty_pair_us = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)[len(pairs) // 2]

    if rank == 0:
        images = world * args.batch * args.steps
        value = images / elapsed
        roof, msda_bwd = None, None
        fwd_kind = "fwd" if args.fp32 else "fwd_bf16"
        fwd = [(a.elapsed_time(b) * 1e-3, n) for kind, a, b, n, _ in events if kind == fwd_kind]
        bwd = [(a.elapsed_time(b) * 1e-3, n) for kind, a, b, n, _ in events if kind == fwd_kind.replace("fwd", "bwd")]
        # distinct value-map cells the in-step launches touch, from the sampling locations recorded in the analysis steps
        cells, whole_map = [], None
        for kind, a, b, n, geo in geom_events:
            if kind == fwd_kind and geo is not None:
                loc, shapes, lsi, B, S, M, esize = geo
                cells.append(msda_op.unique_lines_touched(loc, shapes.cpu(), lsi.cpu(), S, M))
                _, Lq, _, L, P, _ = loc.shape
                whole_map = (B, S, M, 64, L, Lq, P, esize)
        if fwd:
            avg_t = sum(t for t, _ in fwd) / len(fwd)
            whole_bytes = sum(n for _, n in fwd) / len(fwd)  # every tensor once, whole value map (round-1 convention)
            if cells:
                B, S, M, D, L, Lq, P, esize = whole_map
                nbytes = msda_forward_bytes(sum(cells) / len(cells), D * esize, B, Lq, M, D, L, P, esize)
                basis = ("distinct value-map cells touched by the recorded sampling locations (mean over %d launches: %.1f %% of the "
                         "map) + locations + weights + output" % (len(cells), 100.0 * sum(cells) / len(cells) * D * esize
                                                                  / (esize * B * S * M * D)))
            else:  # --no-analysis: no recorded geometry; fall back to PMC traffic so the fraction stays <= 1
                nbytes, basis = None, "not counted (--no-analysis)"
            # HBM-side bytes per launch of this kernel at this shape from the committed PMC profile (separate
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 note); PMC collection
            # cannot run inside the timed benchmark, so the figure is quoted only for the shape it was measured on
            traffic, traffic_src = (pmc_traffic("msda_fwd_bf16_rows4") if (not args.fp32 and args.batch == 32 and args.size == 640
                                                                           and args.points == "model" and not args.ragged)
                                    else (None, None))
            achieved = (nbytes if nbytes is not None else (traffic or 0)) / avg_t / 1e9
            roof = {"bound": "hbm", "kernel": MSDA_FWD_KERNEL[fwd_kind] + ", in the training step", "achieved": achieved,
                    "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                    "traffic_source": traffic_src,
                    "traffic_frac": (traffic / avg_t / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                    "launches": len(fwd), "avg_launch_us": avg_t * 1e6,
                    "algorithmic_bytes_per_launch": int(nbytes) if nbytes is not None else None,
                    "algorithmic_bytes_basis": basis, "whole_map_bytes_per_launch": int(whole_bytes),
                    "points": args.points, "empty_event_pair_us": empty_pair_us}
        if bwd:  # informational: the backward is bound by the chip-wide memory-side atomic rate, not by HBM
            avg_b = sum(t for t, _ in bwd) / len(bwd)
            msda_bwd = {"kernel": "msda_bwd_d64 (f32 maps, f32 atomics)" if args.fp32 else
                        (("msda_bwd_value + msda_bwd_rows4 (gather form: per (image, head) the corner contributions are binned by cell "
                          "in one CU's LDS and every cell is summed on the matrix cores in f32, rounded to bf16 once; no atomics on "
                          "memory, dense output)"
                          if msda_op.F32_METHOD == "sorted" else
                          "msda_bwd_d64_pk<stage> + msda_stage_flush (f32 atomics into a staging map, one rounding to bf16 per touched cell)")
                         if msda_op.F32_ACCUMULATE else "msda_bwd_d64_pk (packed-bf16 atomics, same-cell merges)"),
                        "launches": len(bwd), "avg_launch_us": avg_b * 1e6,
                        "whole_map_bytes_per_launch": int(bwd[0][1])}
            if not args.fp32 and msda_op.F32_ACCUMULATE and msda_op.F32_METHOD == "sorted":
                # the gather form IS an HBM-shaped kernel pair: its compulsory bytes are the dense gradient slice it writes plus
                # locations, weights, grad_out (twice: both kernels) and the gathered corner rows of the row walk
                ta, tb = pmc_traffic("msda_bwd_value"), pmc_traffic("msda_bwd_rows4")
                on_config = args.batch == 32 and args.size == 640 and args.points == "model" and not args.ragged
                traffic_b = (ta[0] + tb[0]) if (on_config and ta[0] and tb[0]) else None
                Bm, Sm, Mm, Lm, Lqm, Pm = (int(v) for v in msda_op.LAST_BWD_SHAPE)
                rows_m, lp_m = Bm * Lqm * Mm, Lm * Pm
                compulsory = Bm * Sm * Mm * 64 * 2 + rows_m * (lp_m * 12 * 2 + 64 * 2 * 2)
                msda_bwd.update({"bound": "hbm", "achieved": compulsory / avg_b / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                 "frac": compulsory / avg_b / 1e9 / HBM_PEAK_GBPS,
                                 "algorithmic_bytes_per_launch": int(compulsory),
                                 "algorithmic_bytes_basis": "dense bf16 gradient slice written once + locations, weights and their "
                                                            "gradients + grad_out read by both kernels; the gathered corner rows of the "
                                                            "row walk are NOT counted (a lower bound on the bytes)",
                                 "traffic": traffic_b, "traffic_source": ta[1] if traffic_b else None,
                                 "traffic_frac": (traffic_b / avg_b / 1e9 / HBM_PEAK_GBPS) if traffic_b else None})
        # window attention against its compulsory HBM bytes (memory-bound at 72 flop/B; the MFMA fraction is informational)
        window_attention = {}
        for kind, name, products, tensors in (("fwd", "winattn_fwd", 2, 4), ("bwd", "winattn_bwd", 5, 7)):
            ev = [(a.elapsed_time(b) * 1e-3, f) for k, a, b, f in wa_events if k == kind]
            if ev:
                tt, ff = sum(t for t, _ in ev), sum(f for _, f in ev)
                units = ff / (products * 2 * 144 * 144 * 32)             # (window, head) pairs over all launches
                nbytes = units * tensors * 144 * 32 * 2
                wa_traffic, wa_src = pmc_traffic(name) if (not args.fp32 and args.batch == 32 and args.size == 640
                                                           and not args.ragged) else (None, None)
                window_attention[name] = {"bound": "hbm", "kernel": name, "launches_per_step": len(ev) / event_steps,
                                          "traffic": wa_traffic, "traffic_source": wa_src,
                                          "traffic_note": "PMC bytes per launch, mean over the launches of a step (stages differ)",
                                          "algorithmic_bytes_per_launch": int(nbytes / len(ev)),
                                          "traffic_frac": (wa_traffic / (tt / len(ev)) / 1e9 / HBM_PEAK_GBPS) if wa_traffic else None,
                                          "ms_per_step": tt / event_steps * 1e3, "avg_launch_us": tt / len(ev) * 1e6,
                                          "algorithmic_bytes_per_step": int(nbytes / event_steps),
                                          "algorithmic_bytes_basis": "%d bf16 [144, 32] slices per (window, head)" % tensors,
                                          "achieved": nbytes / tt / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                          "frac": nbytes / tt / 1e9 / HBM_PEAK_GBPS,
                                          "achieved_TFLOPs": ff / tt / 1e12, "mfma_frac_bf16": ff / tt / MFMA_PEAK_BF16}
        # ---- GEMM families of the step, each against BOTH of its bounds (per-launch HIP events in 2 eager steps behind the timed region)
        FAMILIES = {
            "gemm_nt_bf16": "gemm_nt_bf16<256,256,64,...> (grit_amd/csrc/gemm.hip): the Swin Mlp's fused GEMMs -- fc1 + bias + GELU "
                            "writing pre-activation and activation, fc2 input gradient x GELU' + bias-gradient column sums -- and the stage-0 map's "
                            "narrow products (128 / 384 output columns, K <= 512: HBM streams)",
            "gemm_w4": "gemm_w4_bf16 (grit_amd/csrc/gemm_w4.hip): persistent four-wave kernel, 128 x 128 / 112 x 128 wave tiles -- every long-map "
                       "Linear of the step: qkv / proj / fc2 forward (proj and fc2 with the residual epilogue), the NT input gradients of the Swin "
                       "blocks, PatchMerging, the stacked value projection of the deformable decoder and its input gradient",
            "gemm_short": "gemm_nt_bf16<64,64,64,4,1,3,E> (grit_amd/csrc/gemm.hip, variant 12): the Linears of the two decoders and the grid net "
                          "(640 .. 4 800 rows) -- forward + bias and input gradients on transposed weight copies; three workgroups per CU",
            "wgrad_tn": "wgrad_tn_256 + wgrad_tn_256_grouped (grit_amd/csrc/wgrad_tn.hip): weight gradients dW = dY^T X of the long token "
                        "maps and, grouped, of the decoders' short ones; fp32 row-slice partials",
            "wgrad_small": "wgrad_small (grit_amd/csrc/wgrad.hip): 64 x 64-tile grouped weight gradients of shapes outside 256-multiples",
            "gemm_lib": "hipBLASLt / rocBLAS through torch (tuned table grit_amd/tunableop_gfx950.csv): the sum of gemm_lib_long and "
                        "gemm_lib_short below",
            "gemm_lib_long": "NOT an own kernel -- hipBLASLt through torch on the long token maps (>= 8192 rows): what the policies of "
                             "grit_amd/ops/gemm.py leave there -- the input gradients of input_proj (NN form), K >= 2 048 Linears of the grid net",
            "gemm_lib_short": "NOT an own kernel -- hipBLASLt / rocBLAS through torch on the short maps (< 8192 rows): what the short-map policy "
                              "(grit_amd/ops/gemm.py prefers_own_short) leaves of the decoders' Linears -- K >= 2 048, the gate GEMMs, the fp32 "
                              "vocabulary projection, bias + ReLU fused calls of the detached box refinement",
        }
        n_an = 2  # analysis steps the events cover
        gemm, families = None, {}
        skipped_flops = skippable_flops = 0.0
        if gemm_events:
            def fams_of(kind, pl):
                if kind == "gemm_lib":  # the aggregate AND its long-map / short-map halves (different regimes: MFMA-bound vs latency)
                    return ("gemm_lib", "gemm_lib_long" if pl.get("rows", 0) >= 8192 else "gemm_lib_short")
                return (pl.get("kernel", "gemm_nt_bf16"),)
            for kind, a, b, pl in gemm_events:
                dt = a.elapsed_time(b) * 1e-3
                scale = pl.get("row_scale")
                if scale is not None:  # drop path: 256-row tiles that lie inside dropped samples were not computed
                    rows, per = int(pl["rows"]), int(pl["rows_per_sample"])
                    zero = (scale.detach().float().cpu() == 0)
                    first = torch.arange(0, rows, 256)
                    last = torch.clamp(first + 255, max=rows - 1)
                    lo, hi = first // per, last // per
                    tile_skipped = (lo == hi) & zero[lo]  # the kernel's rule (gemm.hip): the tile lies inside ONE sample, and it is dropped
                    skipped_flops += pl.get("flops", 0.0) * float(tile_skipped.float().mean())
                    skippable_flops += pl.get("flops", 0.0)
                for name in fams_of(kind, pl):
                    f = families.setdefault(name, {"t": 0.0, "flops": 0.0, "bytes": 0.0, "n": 0})
                    f["t"] += dt
                    f["flops"] += pl.get("flops", 0.0)
                    f["bytes"] += pl.get("bytes", 0.0)
                    f["n"] += 1
            gemm = {"note": "every GEMM family of the step against BOTH bounds: `frac_mfma` = algorithmic flops / time / 2.5 PFLOP/s (dense bf16), "
                            "`frac_hbm` = algorithmic bytes (operands once + outputs once + fp32 split partials) / time / 8 TB/s; `bound` names "
                            "the larger one, `frac` is its value.  Per-launch HIP events on the launch stream in %d eager steps right behind "
                            "the timed region (an event pair adds ~5 us to a launch); `traffic` = PMC bytes per launch where "
                            "profiles/r06/pmc_in_step.txt (or an earlier round.s) has the kernel" % n_an}
            for name, f in families.items():
                tt = f["t"]
                fm, fh = f["flops"] / tt / MFMA_PEAK_BF16, f["bytes"] / tt / 1e9 / HBM_PEAK_GBPS
                traffic, tsrc = pmc_traffic("family:" + name)
                gemm[name] = {"kernel": FAMILIES.get(name, name), "launches_per_step": f["n"] / n_an, "ms_per_step": tt / n_an * 1e3,
                              "avg_launch_us": tt / f["n"] * 1e6, "algorithmic_flops_per_launch": f["flops"] / f["n"],
                              "algorithmic_bytes_per_launch": f["bytes"] / f["n"], "TFLOPs": f["flops"] / tt / 1e12,
                              "GBps": f["bytes"] / tt / 1e9, "frac_mfma": fm, "frac_hbm": fh, "bound": "mfma" if fm >= fh else "hbm",
                              "frac": max(fm, fh), "traffic": traffic, "traffic_source": tsrc}
            own = [v for k, v in gemm.items() if isinstance(v, dict) and not k.startswith("gemm_lib")]
            for label, sel in (("gemm_own", own), ("all", [v for k, v in gemm.items() if isinstance(v, dict)
                                                           and k not in ("gemm_lib_long", "gemm_lib_short")])):
                if sel:
                    tt = sum(v["ms_per_step"] for v in sel) * 1e-3
                    ff = sum(v["algorithmic_flops_per_launch"] * v["launches_per_step"] for v in sel)
                    gemm[label] = {"ms_per_step": tt * 1e3, "PFLOPs": ff / tt / 1e15, "mfma_frac_bf16": ff / tt / MFMA_PEAK_BF16}

        def family_roofline(name):
            v = gemm.get(name) if gemm else None
            if not v:
                return None
            mf = v["bound"] == "mfma"
            return {"bound": v["bound"], "kernel": v["kernel"], "launches_per_step": v["launches_per_step"], "ms_per_step": v["ms_per_step"],
                    "avg_launch_us": v["avg_launch_us"],
                    "achieved": v["TFLOPs"] if mf else v["GBps"], "peak": MFMA_PEAK_BF16 / 1e12 if mf else HBM_PEAK_GBPS,
                    "unit": "TFLOP/s" if mf else "GB/s", "frac": v["frac"], "frac_mfma": v["frac_mfma"], "frac_hbm": v["frac_hbm"],
                    "algorithmic_flops_per_launch": v["algorithmic_flops_per_launch"],
                    "algorithmic_bytes_per_launch": v["algorithmic_bytes_per_launch"],
                    "algorithmic_basis": "2 M N K flop per problem; bytes: both operands once, every output map once (two for fc1 + GELU; the "
                                         "GELU' GEMM also reads the pre-activation map), fp32 row-slice partials of the weight gradients",
                    "traffic": v["traffic"], "traffic_source": v["traffic_source"],
                    "timing": "per-launch HIP events on the launch stream in %d eager steps right behind the timed region" % n_an}
        roof_wgrad = family_roofline("wgrad_tn")
        roof_lib_long = family_roofline("gemm_lib_long")
        # the kernel family with the most time per step -- ALL of them compete: the own GEMM families, the two
        # window-attention kernels AND the library's long-map GEMMs, which the line marks as not an own kernel when they win
        for r, own_kernel in ((roof_lib_long, False), (roof_wgrad, True)):
            if r:
                r["own_kernel"] = own_kernel
        candidates = [r for r in (family_roofline("gemm_nt_bf16"), roof_wgrad, family_roofline("gemm_w4"), roof_lib_long,
                                  window_attention.get("winattn_bwd"), window_attention.get("winattn_fwd")) if r]
        for r in candidates:
            r.setdefault("own_kernel", True)
        dominant = max(candidates, key=lambda r: r["ms_per_step"]) if candidates else roof
        if roof:  # MSDeformAttn forward: both denominators side by side
            roof["frac_touched"] = roof["frac"]
            roof["frac_compulsory_8d"] = roof["whole_map_bytes_per_launch"] / (roof["avg_launch_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
            roof["denominators"] = ("frac / frac_touched: distinct value cells the recorded sampling locations reach + locations + weights + "
                                    "output; frac_compulsory_8d: every tensor once incl. the WHOLE value map (SURVEY 8d's figure -- counts "
                                    "cells the kernel never needs)")
        out = {
            "metric": _baseline_metric(),
            "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "fp32" if args.fp32 else "bf16", "data": "synthetic",
            "config": {"workload": f"GRIT XE training step (Swin-B w12 + 6 deformable decoder layers + 3-layer grid net + "
                                   f"3-layer caption decoder, 161M params, random init), {args.size}x{args.size} images, "
                                   f"caption length {args.caption_len}, Adam x2, dropout on",
                       "global_batch": world * args.batch, "per_gpu_batch": args.batch,
                       "parallelism": f"dp{world}",
                       "backend": backend if world > 1 else None,
                       "rccl_ranks": dist.get_world_size() if (world > 1 and backend == "nccl") else (1 if world == 1 else 0),
                       "rank_devices": rank_devices,
                       "grad_allreduce": (("RCCL" if backend == "nccl" else backend + " (plumbing run, not the contract backend)")
                                          + (" bucketed all-reduce" if grad_sync == "allreduce" or args.fp32 else
                                             " bucketed reduce-scatter + sharded FlatAdam + all-gather of the bf16 weights")
                                          + " (64 MiB flat bf16 buckets, 8 MiB tail), overlapped with backward")
                       if world > 1 else ("none (1 GPU)" if not self_coll else
                                          backend + " SELF-collectives of a one-rank group (" + grad_sync + "): the sync path "
                                          "without a wire, not a multi-GPU number"),
                       "grad_sync": grad_sync if (world > 1 or self_coll) else None, "self_collectives": bool(self_coll),
                       "step_graph": graphed is not None, "step_graph_error": graph_error, "step_graph_reason": graph_reason,
                       "step_graph_segments": (graph_plan.count('graph') if graph_plan is not None else 1) if graphed is not None else 0,
                       "step_enqueue": (("one captured HIP graph replayed per step" if graph_plan is None else
                                         "%d captured graph segments per step, the %d bucket all-reduces issued eagerly between them on the "
                                         "process group's stream" % (graph_plan.count('graph'), graph_plan.count('collective')))
                                        + " (grit_amd/engine/graph_step.py); per-launch kernel events come from %d eager steps behind "
                                          "the timed region" % event_steps) if graphed is not None
                       else "eager launches",
                       "rccl_env": rccl_env,
                       "points": args.points, "ragged": bool(args.ragged),
                       "msda_backward_accumulation": ("f32" if args.fp32 else ("f32 (" + msda_op.F32_METHOD + ")") if msda_op.F32_ACCUMULATE
                                                      else "bf16 (packed atomics)")},
            "mfma_roofline_frac_bf16": value / world * FLOP_PER_IMAGE_FWD_BWD / MFMA_PEAK_BF16,
            # ... the same with the flops the step EXECUTES: the fused Mlp GEMMs do not compute the 256-row tiles inside samples that drop path
            # removed from the branch (counted per launch from the drawn factors of the analysis steps)
            "flops_per_image": {"nominal": FLOP_PER_IMAGE_FWD_BWD,
                                "executed": FLOP_PER_IMAGE_FWD_BWD - skipped_flops / n_an / args.batch,
                                "skipped_fraction_of_skippable_gemms": (skipped_flops / skippable_flops) if skippable_flops else 0.0},
            "mfma_roofline_frac_bf16_executed": value / world * (FLOP_PER_IMAGE_FWD_BWD - skipped_flops / n_an / args.batch) / MFMA_PEAK_BF16,
            "final_loss": final_loss,
            # the hand-written kernel with the most time per step (profiles/r03/*steady_state.txt): the long-map weight-gradient GEMM
            # (MFMA-bound) once the analysis steps ran, else the window-attention backward (HBM-bound); both are always reported below
            "roofline": dominant,
            "roofline_gemm_nt_bf16": family_roofline("gemm_nt_bf16"),
            "roofline_gemm_w4": family_roofline("gemm_w4"),
            "roofline_wgrad_tn": roof_wgrad,
            "roofline_gemm_lib_long": roof_lib_long,
            "roofline_msda": roof,
            "msda_backward": msda_bwd,
            "roofline_winattn_bwd": window_attention.get("winattn_bwd"),
            "roofline_winattn_fwd": window_attention.get("winattn_fwd"),
            "gemm": gemm,
            "config3_bs16": None,
        }
        if not args.no_analysis:
            del wrapped, optimizers, model
            torch.cuda.empty_cache()
            out["roofline_msda_spread"] = msda_spread_microbench(device)
            if world == 1:  # config 5 on one GPU, driver-visible (after the timed region, training state released)
                out["decode_config5"] = decode_config5(device, config)
                torch.cuda.empty_cache()
            if world == 1 and not args.fp32 and args.batch == 32 and not args.ragged and args.points == "model":
                out["config3_bs16"] = config3_child(args)
        if world == 1 and not args.no_cpu_baseline:
            torch.cuda.empty_cache()
            out["cpu_baseline"] = cpu_baseline(config, args.size, args.caption_len, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
