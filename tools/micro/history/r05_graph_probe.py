"""Round-5 probes of HIP-graph behaviour on this stack (ROCm 7.2 / torch 2.10), run once on the GPU box:
 (1) do independent branches of a captured graph (forked streams) run CONCURRENTLY on replay?
 (2) can a capture be ended and the next one begun from an autograd hook (the engine's device thread) in relaxed mode?
 (3) timing of a replayed chain of tiny dependent kernels (the decoder phase's regime)."""
import sys
import time

import torch


def timed_replay(g, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def chain(x, w, steps):
    for _ in range(steps):
        x = torch.relu(x @ w)
    return x


def probe_branches(rows, steps=200):
    dev = 'cuda'
    w = (torch.randn(512, 512, device=dev) * 0.04).bfloat16()
    a = torch.randn(rows, 512, device=dev).bfloat16()
    b = torch.randn(rows, 512, device=dev).bfloat16()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        chain(a, w, 3), chain(b, w, 3)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        ya = chain(a, w, steps)
        yb = chain(b, w, steps)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            yb2 = chain(b, w, steps)
        ya2 = chain(a, w, steps)
        cur.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        ya3 = chain(a, w, steps)
    t1, t2, t3 = timed_replay(g1), timed_replay(g2), timed_replay(g3)
    ok = torch.equal(ya, ya2) and torch.equal(yb, yb2)
    print(f"branches rows={rows}: serial 2 chains {t1:.3f} ms, forked {t2:.3f} ms, one chain {t3:.3f} ms  ({2 * steps} kernels of "
          f"{t3 / (2 * steps) * 1e3:.1f} us each)  equal={ok}", flush=True)


def probe_hook_segments():
    dev = 'cuda'
    lin = [torch.nn.Linear(256, 256, device=dev) for _ in range(4)]
    x = torch.randn(64, 256, device=dev)

    def fwd_bwd():
        h = x
        for l in lin:
            h = torch.relu(l(h))
        h.sum().backward()

    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fwd_bwd()
    torch.cuda.synchronize()
    ref = [l.weight.grad.clone() for l in lin]
    for l in lin:
        l.weight.grad = None
        l.bias.grad = None
    graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
    state = {'k': 0, 'threads': []}
    import threading

    def cut(_p):
        state['threads'].append(threading.get_ident())
        k = state['k']
        graphs[k].capture_end()
        state['k'] = k + 1
        graphs[k + 1].capture_begin(pool=pool, capture_error_mode="relaxed")

    hooks = [lin[2].weight.register_post_accumulate_grad_hook(cut), lin[0].weight.register_post_accumulate_grad_hook(cut)]
    pool = torch.cuda.graph_pool_handle()
    main = threading.get_ident()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        graphs[0].capture_begin(pool=pool, capture_error_mode="relaxed")
        fwd_bwd()
        graphs[state['k']].capture_end()
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    print("hook threads", [t == main for t in state['threads']], "segments", state['k'] + 1, flush=True)
    for l in lin:
        l.weight.grad.zero_()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    print("segmented replay equals eager:", all(torch.allclose(l.weight.grad, r) for l, r in zip(lin, ref)), flush=True)
    for l in lin:
        l.weight.grad.zero_()
    for g in graphs:
        g.replay()
    torch.cuda.synchronize()
    print("second replay equals eager:", all(torch.allclose(l.weight.grad, r) for l, r in zip(lin, ref)), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ['branches', 'hooks']
    if 'branches' in what:
        for rows in (160, 640, 4800):
            probe_branches(rows)
    if 'hooks' in what:
        try:
            probe_hook_segments()
        except Exception as e:
            print("hook segments FAILED:", type(e).__name__, str(e)[:500], flush=True)
