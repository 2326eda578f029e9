"""Streaming kernels around the library GEMMs: LayerNorm fwd/bwd, column sums (bias gradients), the slab-sum second
stage, and the Linear module's re-posed backward -- each against torch evaluated in float64 on the same inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("groups,alloc,slabs,n", [(1, 1, 1, 8), (1, 3, 3, 128), (2, 512, 37, 128), (2, 512, 512, 4096),
                                                   (1, 64, 64, 2048 * 512), (1, 16, 16, 100), (3, 9, 5, 260), (3, 1024, 1024, 512),
                                                   (2, 1024, 700, 128), (1, 256, 256, 1536)])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_slab_sum(groups, alloc, slabs, n, out_dtype):
    """grit_slab_sum: out[g] = cast(sum of the first `slabs` slabs of group g); slabs beyond that are never read."""
    from grit_amd.ops.linear import slab_sum
    g = torch.Generator().manual_seed(n + slabs)
    part = torch.randn(groups, alloc, n, generator=g).to(DEV)
    part[:, slabs:] = float("nan")
    got = slab_sum(part, out_dtype, slabs=slabs)
    ref = part[:, :slabs].double().sum(1)
    assert got.shape == (groups, n) and got.dtype == out_dtype
    tol = 1e-5 if out_dtype == torch.float32 else 8e-3
    np.testing.assert_allclose(got.double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * slabs ** 0.5)


def test_grouped_slab_sum_equals_separate_launches():
    """grit_slab_sum_grouped: the slab sums a backward node owes in ONE launch -- the shapes of a Swin Mlp node (LayerNorm
    partial sums [3, 1024, C] with fewer live slabs, split-M weight gradients [1, 16, N, K], the GELU-epilogue column sums
    [1, 400, 4C]) plus odd ones, bf16 and f32 outputs mixed, 19 jobs (> GRIT_SLAB_GROUP_MAX: two launches).  Bit-identical to
    grit_slab_sum job by job (same per-job arithmetic)."""
    from grit_amd.ops.linear import SlabGroup, slab_sum
    g = torch.Generator().manual_seed(3)
    specs = [((3, 1024, 512), 400, torch.bfloat16), ((1, 16, 2048, 512), None, torch.bfloat16), ((1, 400, 2048), None, torch.bfloat16),
             ((1, 16, 512, 2048), None, torch.bfloat16), ((2, 7, 12), 5, torch.float32), ((1, 1, 4), None, torch.bfloat16),
             ((1, 64, 256, 1024), None, torch.float32), ((3, 1024, 128), 1024, torch.float32), ((1, 33, 1536), 33, torch.bfloat16)]
    specs = specs + specs + [((1, 4, 8), None, torch.float32)]
    parts = []
    for shape, slabs, dt in specs:
        p_ = torch.randn(*shape, generator=g).to(DEV)
        if slabs is not None:
            p_[:, slabs:] = float("nan")
        parts.append(p_)
    group = SlabGroup()
    outs = [group.add(p_, dt, slabs=sl) for p_, (_, sl, dt) in zip(parts, specs)]
    assert len(group.jobs) == 19
    group.run()
    assert not group.jobs
    for p_, (shape, sl, dt), out in zip(parts, specs, outs):
        want = slab_sum(p_, dt, slabs=sl)
        assert out.shape == want.shape and out.dtype == dt
        assert torch.equal(out, want), shape
        ref = p_[:, :(sl or shape[1])].double().sum(1)
        tol = 1e-5 if dt == torch.float32 else 8e-3
        np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol * shape[1] ** 0.5)


@pytest.mark.parametrize("C", [128, 256, 512, 1024, 2048, 4096])
@pytest.mark.parametrize("dtype,wdtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
def test_layer_norm_kernels_vs_torch(C, dtype, wdtype):
    """grit_layernorm_{fwd,bwd} against F.layer_norm evaluated in fp64 on the same (rounded) inputs."""
    from grit_amd.ops.layer_norm import layer_norm
    g = torch.Generator().manual_seed(C)
    rows = 1000 + C // 128  # not a multiple of the rows-per-block
    x = (torch.randn(rows, C, generator=g) * 2 + 0.5).to(dtype)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(wdtype)
    b = (0.1 * torch.randn(C, generator=g)).to(wdtype)
    cot = torch.randn(rows, C, generator=g).to(dtype)
    xr, wr, br = (z.double().requires_grad_(True) for z in (x, w, b))
    torch.nn.functional.layer_norm(xr, (C,), wr, br, 1e-5).backward(cot.double())
    xd, wd, bd = (z.to(DEV).requires_grad_(True) for z in (x, w, b))
    y = layer_norm(xd.view(4, -1, C) if rows % 4 == 0 else xd, wd, bd, 1e-5)
    y.backward(cot.to(DEV).view(y.shape))
    ref_y = torch.nn.functional.layer_norm(x.double(), (C,), w.double(), b.double(), 1e-5)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    np.testing.assert_allclose(y.detach().float().cpu().numpy().reshape(rows, C), ref_y.float().numpy(), rtol=tol, atol=tol)
    np.testing.assert_allclose(xd.grad.float().cpu().numpy(), xr.grad.float().numpy(), rtol=tol, atol=tol)
    for got, ref in ((wd.grad, wr.grad), (bd.grad, br.grad)):
        scale = ref.abs().max().item()
        assert (got.float().cpu() - ref.float()).abs().max().item() < (2e-2 if wdtype == torch.bfloat16 else 2e-3) * scale


@pytest.mark.parametrize("M,N,dtype", [(51200, 512, torch.bfloat16), (4097, 1536, torch.bfloat16), (5000, 2048, torch.float32),
                                       (204800, 256, torch.bfloat16), (4096, 8, torch.float32)])
def test_column_sum_and_linear_backward(M, N, dtype):
    """grit_colsum (bias gradient of the Swin Linears) against a float64 sum; Linear module grads vs nn.Linear."""
    from grit_amd.ops.linear import Linear, column_sum
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g).to(dtype).to(DEV)
    ref = x.double().sum(0)
    got = column_sum(x)
    scale = ref.abs().max().item()
    assert (got.double() - ref).abs().max().item() < 1e-5 * scale + 1e-3
    if N >= 256:
        return
    lin = Linear(N, 16).to(DEV).to(dtype)
    ref_lin = torch.nn.Linear(N, 16).to(DEV).to(dtype)
    ref_lin.load_state_dict(lin.state_dict())
    a = x.clone().requires_grad_(True)
    b_ = x.clone().requires_grad_(True)
    cot = torch.randn(M, 16, generator=g).to(dtype).to(DEV)
    lin(a).backward(cot)
    ref_lin(b_).backward(cot)
    for p, q in ((a.grad, b_.grad), (lin.weight.grad, ref_lin.weight.grad), (lin.bias.grad, ref_lin.bias.grad)):
        assert torch.allclose(p.float(), q.float(), rtol=2e-2, atol=2e-2 * q.float().abs().max().item())


@pytest.mark.parametrize("C,B,L", [(128, 3, 1601), (256, 2, 400), (512, 4, 100), (1024, 2, 25)])
@pytest.mark.parametrize("dtype,wdtype", [(torch.bfloat16, torch.bfloat16), (torch.float32, torch.float32)])
@pytest.mark.parametrize("drop_path", [False, True])
def test_add_layer_norm_fused_vs_composition(C, B, L, dtype, wdtype, drop_path):
    """grit_add_layernorm_{fwd,bwd}: x = shortcut + scale[b] * branch, y = LN(x) (the residual + norm pairs of a Swin
    block, swin_model.py:289-298) against the same composition in float64; the stored sum must be BIT-identical to
    torch's own add / addcmul in the tensor dtype, and both outputs carry gradient (x feeds the next skip path)."""
    from grit_amd.ops.layer_norm import add_layer_norm
    g = torch.Generator().manual_seed(C + L)
    sc, br = (torch.randn(B, L, C, generator=g) * 1.5).to(dtype), torch.randn(B, L, C, generator=g).to(dtype)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(wdtype)
    b = (0.1 * torch.randn(C, generator=g)).to(wdtype)
    scale = (torch.tensor([0.0, 1.25, 1.25, 0.0][:B]) if drop_path else None)
    cot_x, cot_y = torch.randn(B, L, C, generator=g).to(dtype), torch.randn(B, L, C, generator=g).to(dtype)
    # float64 reference on the same rounded inputs, with the sum rounded to the tensor dtype like the unfused op
    s64, b64, w64, bb64 = (z.double().requires_grad_(True) for z in (sc, br, w, b))
    sum64 = s64 + (b64 if scale is None else b64 * scale.double().view(-1, 1, 1))
    x_ref = sum64 + (sum64.detach().to(dtype).double() - sum64.detach())  # straight-through rounding
    y_ref = torch.nn.functional.layer_norm(x_ref, (C,), w64, bb64, 1e-5)
    (x_ref * cot_x.double()).sum().backward(retain_graph=True)
    (y_ref * cot_y.double()).sum().backward()
    sd, bd, wd, bbd = (z.to(DEV).requires_grad_(True) for z in (sc, br, w, b))
    x, y = add_layer_norm(sd, bd, None if scale is None else scale.to(DEV), wd, bbd, 1e-5)
    torch.autograd.backward([x, y], [cot_x.to(DEV), cot_y.to(DEV)])
    unfused = sc.to(DEV) + br.to(DEV) if scale is None else torch.addcmul(sc.to(DEV), br.to(DEV), scale.to(DEV).to(dtype).view(-1, 1, 1))
    if dtype == torch.bfloat16 or scale is None:
        assert torch.equal(x.detach(), unfused)
    else:  # fp32 addcmul contracts a + b*c into one fma inside torch's kernel: 1 ulp
        np.testing.assert_allclose(x.detach().cpu().numpy(), unfused.cpu().numpy(), rtol=3e-7, atol=5e-7)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-5
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), y_ref.detach().float().numpy(), rtol=tol, atol=tol)
    gtol = 4e-2 if dtype == torch.bfloat16 else 1e-4
    np.testing.assert_allclose(sd.grad.float().cpu().numpy(), s64.grad.float().numpy(), rtol=gtol, atol=gtol)
    np.testing.assert_allclose(bd.grad.float().cpu().numpy(), b64.grad.float().numpy(), rtol=gtol, atol=gtol * 1.25)
    for got, ref in ((wd.grad, w64.grad), (bbd.grad, bb64.grad)):
        assert (got.float().cpu() - ref.float()).abs().max().item() < (2e-2 if wdtype == torch.bfloat16 else 2e-3) * ref.abs().max().item()


@pytest.mark.parametrize("dtype,wdtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
@pytest.mark.parametrize("C,G,Ts,B", [(512, 32, (400, 100, 25, 7), 3), (256, 32, (64, 33), 2), (512, 16, (1601,), 2)])
def test_group_norm_levels_vs_torch(C, G, Ts, B, dtype, wdtype):
    """grit_groupnorm_tokens_{fwd,bwd}: per-level GroupNorm on token-major maps written into one flat [B, S, C] map,
    against torch.nn.functional.group_norm on the channels-first view + cat (detector.py:28-33, det_module.py:172-175)
    in float64 on the same inputs; gradients of inputs, weights and biases."""
    from grit_amd.ops.group_norm import group_norm_levels
    g = torch.Generator().manual_seed(C + len(Ts))
    xs = [(torch.randn(B, T, C, generator=g) * (1 + l) + 0.3 * l).to(dtype) for l, T in enumerate(Ts)]
    ws = [(1 + 0.2 * torch.randn(C, generator=g)).to(wdtype) for _ in Ts]
    bs = [(0.1 * torch.randn(C, generator=g)).to(wdtype) for _ in Ts]
    cot = torch.randn(B, sum(Ts), C, generator=g).to(dtype)
    xr, wr, br = ([z.double().requires_grad_(True) for z in zs] for zs in (xs, ws, bs))
    ref = torch.cat([torch.nn.functional.group_norm(x.transpose(1, 2), G, w, b, 1e-5).transpose(1, 2)
                     for x, w, b in zip(xr, wr, br)], 1)
    ref.backward(cot.double())
    xd, wd, bd = ([z.to(DEV).requires_grad_(True) for z in zs] for zs in (xs, ws, bs))
    out = group_norm_levels(xd, wd, bd, G, 1e-5)
    assert out.shape == (B, sum(Ts), C) and out.dtype == dtype and out.is_contiguous()
    out.backward(cot.to(DEV))
    tol = 3e-2 if dtype == torch.bfloat16 else 2e-4
    np.testing.assert_allclose(out.detach().float().cpu().numpy(), ref.detach().float().numpy(), rtol=tol, atol=tol)
    for got, want in zip(xd, xr):
        scale = want.grad.abs().max().item()
        assert (got.grad.double().cpu() - want.grad).abs().max().item() < (3e-2 if dtype == torch.bfloat16 else 2e-4) * scale
    for gots, wants in ((wd, wr), (bd, br)):
        for got, want in zip(gots, wants):
            scale = want.grad.abs().max().item()
            assert (got.grad.double().cpu() - want.grad).abs().max().item() < (2e-2 if wdtype == torch.bfloat16 else 3e-3) * scale


def test_flat_adam_matches_torch_adam_and_state_dict_roundtrip():
    """grit_adam_flat through grit_amd.amp.FlatAdam (bf16 gradient buckets -> fp32 masters + moments -> bf16 compute
    copy, two optimizers splitting the flat space as build_optimizers does) against torch.optim.Adam stepping fp32
    clones with the same (bf16-rounded) gradients, 6 steps; then state_dict() -> load_state_dict() into a fresh wrapper
    continues identically."""
    import copy
    from grit_amd.amp import Bf16Compute

    def make():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(24, 50), torch.nn.GELU(), torch.nn.Linear(50, 2), torch.nn.Tanh(),
                                   torch.nn.Linear(2, 37), torch.nn.LayerNorm(37), torch.nn.Linear(37, 8)).to(DEV)

    def build(module):
        w = Bf16Compute(module, bucket_mb=0.002)  # several tiny buckets: runs cross bucket boundaries
        named = w.named_master_parameters()
        head = [p for n, p in named if n.startswith(("0.", "2."))]
        tail = [p for n, p in named if not n.startswith(("0.", "2."))]
        opts = [w.flat_adam([{'params': head[:2]}, {'params': head[2:]}], 1e-2, (0.9, 0.99)), w.flat_adam(tail, 3e-3, (0.9, 0.99))]
        return w, named, opts

    w, named, opts = build(make())
    assert w.flat_optimizer and len(w.ddp.buckets) >= 2
    ref_params = [p.detach().clone().requires_grad_(True) for _, p in named]
    n_head = sum(len(g['params']) for g in opts[0].param_groups)
    names = [n for n, _ in named]
    head_idx = [i for i, n in enumerate(names) if n.startswith(("0.", "2."))]
    tail_idx = [i for i in range(len(names)) if i not in head_idx]
    ref_opts = [torch.optim.Adam([ref_params[i] for i in head_idx], lr=1e-2, betas=(0.9, 0.99)),
                torch.optim.Adam([ref_params[i] for i in tail_idx], lr=3e-3, betas=(0.9, 0.99))]
    assert n_head == len(head_idx)
    compute = dict(w.module.named_parameters())
    gen = torch.Generator(device=DEV).manual_seed(0)

    def one_step(wrapper, optimizers, check=True):
        x = torch.randn(64, 24, device=DEV, generator=gen).bfloat16()
        wrapper(x).float().square().mean().backward()
        wrapper.finish_gradient_sync()
        if check:
            for i, n in enumerate(names):
                ref_params[i].grad = compute[n].grad.detach().float().clone()
        for o in optimizers:
            o.step()
        wrapper.after_optimizer_step()
        if check:
            for o in ref_opts:
                o.step()

    for _ in range(6):
        one_step(w, opts)
    for i, (n, m) in enumerate(named):
        np.testing.assert_allclose(m.detach().cpu().numpy(), ref_params[i].detach().cpu().numpy(), rtol=2e-6, atol=2e-7, err_msg=n)
        assert torch.equal(compute[n].detach(), m.detach().bfloat16()), n
    sd = [copy.deepcopy(o.state_dict()) for o in opts]
    assert sd[0]['state'][0]['exp_avg'].shape == named[head_idx[0]][1].shape and float(sd[0]['state'][0]['step']) == 6.0
    # resume: fresh wrapper from the exported masters + optimizer state, one more identical step on both
    state = w.master_state_dict()
    fresh = make()
    fresh.load_state_dict(state)
    w2, named2, opts2 = build(fresh)
    for o, s in zip(opts2, sd):
        o.load_state_dict(s)
    gen_state = gen.get_state()
    one_step(w, opts, check=False)
    gen.set_state(gen_state)
    one_step(w2, opts2, check=False)
    for (n, a), (_, b) in zip(named, named2):
        assert torch.equal(a.detach(), b.detach()), n


@pytest.mark.parametrize("C,K,B,L,drop_path", [(128, 128, 2, 2500, True), (512, 2048, 3, 1400, False), (256, 1024, 2, 2100, True)])
def test_linear_add_layer_norm_node(C, K, B, L, drop_path):
    """linear_add_layer_norm: x = shortcut + scale * (inp @ W^T + b), y = LN(x) as ONE autograd node (attn.proj / mlp.fc2 +
    residual + the following norm of a Swin block, swin_model.py:289-298) against the float64 composition: outputs and
    the gradients of inp, W, b (delivered by the LayerNorm backward kernel's extra column sums), shortcut, gamma, beta."""
    from grit_amd.ops.layer_norm import linear_add_layer_norm
    from grit_amd.ops.linear import Linear
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(C + K)
    inp = torch.randn(B, L, K, generator=g).to(dtype)
    sc = (torch.randn(B, L, C, generator=g) * 1.5).to(dtype)
    lin = Linear(K, C)
    lin.weight.data = (torch.randn(C, K, generator=g) / K ** 0.5)
    lin.bias.data = 0.3 * torch.randn(C, generator=g)
    lin = lin.to(dtype)
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(dtype)
    b = (0.1 * torch.randn(C, generator=g)).to(dtype)
    scale = torch.tensor([1.25, 0.0, 1.25][:B]) if drop_path else None
    cot_x, cot_y = torch.randn(B, L, C, generator=g).to(dtype), torch.randn(B, L, C, generator=g).to(dtype)
    i64, s64, W64, B64, w64, b64 = (z.detach().double().requires_grad_(True) for z in (inp, sc, lin.weight, lin.bias, w, b))
    branch = torch.nn.functional.linear(i64, W64, B64)
    branch = branch + (branch.detach().to(dtype).double() - branch.detach())  # the GEMM output is stored in bf16
    x_ref = s64 + (branch if scale is None else branch * scale.double().view(-1, 1, 1))
    y_ref = torch.nn.functional.layer_norm(x_ref + (x_ref.detach().to(dtype).double() - x_ref.detach()), (C,), w64, b64, 1e-5)
    torch.autograd.backward([x_ref, y_ref], [cot_x.double(), cot_y.double()])
    lin_d = lin.to(DEV)
    i_d, s_d, w_d, b_d = (z.to(DEV).requires_grad_(True) for z in (inp, sc, w, b))
    x, y = linear_add_layer_norm(i_d, lin_d, s_d, None if scale is None else scale.to(DEV), w_d, b_d, 1e-5)
    torch.autograd.backward([x, y], [cot_x.to(DEV), cot_y.to(DEV)])
    np.testing.assert_allclose(x.detach().float().cpu().numpy(), x_ref.detach().float().numpy(), rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(y.detach().float().cpu().numpy(), y_ref.detach().float().numpy(), rtol=3e-2, atol=3e-2)
    for name, got, want in (("d_inp", i_d.grad, i64.grad), ("d_shortcut", s_d.grad, s64.grad), ("dW", lin_d.weight.grad, W64.grad),
                            ("db", lin_d.bias.grad, B64.grad), ("dgamma", w_d.grad, w64.grad), ("dbeta", b_d.grad, b64.grad)):
        scale_ = want.abs().max().item()
        err = (got.double().cpu() - want).abs().max().item()
        assert err < 4e-2 * scale_, (name, err, scale_)


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_linear_add_layer_norm_dropout_mask_is_consistent(p):
    """Element dropout inside the fused tail (nn.Dropout between a projection and LayerNorm(x + .), post-norm decoder
    layers): the keep mask is regenerated from the device seed in the backward.  fp32, identity projection: the mask is
    read back from the stored sum, its rate is checked, and outputs / every gradient are compared with the same mask
    applied through torch ops."""
    from grit_amd.ops.layer_norm import linear_add_layer_norm
    C, B, L = 512, 4, 700
    g = torch.Generator().manual_seed(11)
    inp = (torch.randn(B, L, C, generator=g) + 3.0).to(DEV).requires_grad_(True)  # away from 0: dropped <=> exactly 0
    sc = torch.randn(B, L, C, generator=g).to(DEV).requires_grad_(True)
    lin = torch.nn.Linear(C, C).to(DEV)
    with torch.no_grad():
        lin.weight.copy_(torch.eye(C))
        lin.bias.zero_()
    w = (1 + 0.2 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    b = (0.1 * torch.randn(C, generator=g)).to(DEV).requires_grad_(True)
    cot_x, cot_y = torch.randn(B, L, C, generator=g).to(DEV), torch.randn(B, L, C, generator=g).to(DEV)
    torch.manual_seed(5)
    x, y = linear_add_layer_norm(inp, lin, sc, None, w, b, 1e-5, dropout_p=p, training=True)
    torch.autograd.backward([x, y], [cot_x, cot_y])
    keep = ((x.detach() - sc.detach()).abs() > 1e-6)
    rate = 1.0 - keep.float().mean().item()
    assert abs(rate - p) < 0.01, rate
    got = [t_.grad.clone() for t_ in (inp, sc, lin.weight, lin.bias, w, b)]
    for t_ in (inp, sc, lin.weight, lin.bias, w, b):
        t_.grad = None
    branch = torch.nn.functional.linear(inp, lin.weight, lin.bias) * keep.float() / (1.0 - p)
    x_ref = sc + branch
    y_ref = torch.nn.functional.layer_norm(x_ref, (C,), w, b, 1e-5)
    torch.autograd.backward([x_ref, y_ref], [cot_x, cot_y])
    np.testing.assert_allclose(x.detach().cpu().numpy(), x_ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(y.detach().cpu().numpy(), y_ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    for name, a, t_ in zip(("d_inp", "d_shortcut", "dW", "db", "dgamma", "dbeta"), got, (inp, sc, lin.weight, lin.bias, w, b)):
        scale_ = t_.grad.abs().max().item()
        assert (a - t_.grad).abs().max().item() < 2e-3 * scale_, name
    # a second call draws a different mask
    x2, _ = linear_add_layer_norm(inp, lin, sc, None, w, b, 1e-5, dropout_p=p, training=True)
    from grit_amd.ops import backend
    assert not torch.equal((x2.detach() - sc.detach()).abs() > 1e-6, keep), (backend._seeds.used, backend._seeds.buf[:6].tolist())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_shared_input_linears_match_separate_linears(dtype):
    """Six Linear layers on one input as one node (input gradient accumulated by the GEMMs) against the same six layers
    applied separately through autograd; one output is left unused (its gradient is None in the node's backward)."""
    from grit_amd.ops.linear import Linear, shared_input_linears
    torch.manual_seed(0)
    x = torch.randn(2, 4250, 256, device=DEV, dtype=dtype, requires_grad=True)
    lins = [Linear(256, 256).to(DEV, dtype) for _ in range(6)]
    cots = [torch.randn(2, 4250, 256, device=DEV, dtype=dtype) for _ in range(6)]
    ys = shared_input_linears(x, lins)
    loss = sum((y.float() * c.float()).sum() for i, (y, c) in enumerate(zip(ys, cots)) if i != 3)
    loss.backward()
    got = [x.grad.clone()] + [l.weight.grad.clone() if l.weight.grad is not None else None for l in lins] + \
          [l.bias.grad.clone() if l.bias.grad is not None else None for l in lins]
    x.grad = None
    for l in lins:
        l.weight.grad = l.bias.grad = None
    ref_ys = [torch.nn.functional.linear(x, l.weight, l.bias) for l in lins]
    for y, r in zip(ys, ref_ys):
        assert torch.equal(y, r)
    loss = sum((y.float() * c.float()).sum() for i, (y, c) in enumerate(zip(ref_ys, cots)) if i != 3)
    loss.backward()
    want = [x.grad] + [l.weight.grad for l in lins] + [l.bias.grad for l in lins]
    assert got[1 + 3] is None and got[7 + 3] is None and want[1 + 3] is None
    tol = dict(rtol=2e-2, atol=2e-1) if dtype == torch.bfloat16 else dict(rtol=1e-4, atol=1e-3)
    for g, w in zip(got, want):
        if w is not None:
            torch.testing.assert_close(g.float(), w.float(), **tol)


def test_small_map_weight_gradients_beside_the_chain_equal_the_inline_ones(monkeypatch):
    """GRIT_WGRAD_STREAM_SMALL: inside a gradient-bucket wrapper's scope the weight / bias gradients of single-use Linears on
    small maps run on a side stream and the main stream waits only when the wrapper packs them.  Same numbers as with the
    knob off, bit for bit; a Linear applied twice (not declared single-use) is untouched by the mechanism; after
    finish_gradient_sync nothing is pending."""
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.ops import linear as L
    from grit_amd.ops.layer_norm import linear_add_layer_norm

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = torch.nn.ModuleList(L.Linear(512, 1024) for _ in range(3))
            self.b = torch.nn.ModuleList(torch.nn.Linear(1024, 512) for _ in range(3))
            self.norm = torch.nn.LayerNorm(512)
            self.twice = L.Linear(512, 512)
            L.mark_single_use(self.a, self.b)

        def forward(self, x):
            for a, b in zip(self.a, self.b):
                h = torch.relu(a(x))
                x = linear_add_layer_norm(h, b, x, None, self.norm.weight, self.norm.bias, 1e-5, 0.0, True)[1]
                x = self.twice(self.twice(x))
            return x

    torch.manual_seed(0)
    net = Net().to(DEV).bfloat16()
    x = torch.randn(8, 600, 512, device=DEV).bfloat16()
    cot = torch.randn(8, 600, 512, device=DEV).bfloat16()
    ddp = BucketedDataParallel(net, bucket_mb=1)
    monkeypatch.setattr(L, "WGRAD_DEFER", False)  # this test is about the side-stream variant of the per-node path
    results = []
    for knob in (False, True, True):
        monkeypatch.setattr(L, "WGRAD_STREAM_SMALL", knob)
        forks = []
        real_fork = L.fork
        monkeypatch.setattr(L, "fork", lambda *a, **k: (forks.append(real_fork(*a, **k)) or forks[-1]))
        import grit_amd.ops.layer_norm as LNmod
        monkeypatch.setattr(LNmod, "fork", L.fork)
        (ddp(x).float() * cot.float()).sum().backward()
        deferred = sum(1 for f in forks if f is not None and getattr(f, "deferred", False))
        ddp.finish_gradient_sync()
        assert not L._deferral["pending"] and not L._deferral["active"]
        torch.cuda.synchronize()
        results.append(({n: p.grad.clone() for n, p in net.named_parameters()}, deferred))
        monkeypatch.setattr(L, "fork", real_fork)
        monkeypatch.setattr(LNmod, "fork", real_fork)
    # b[0..2], a[1..2] (a[0] reads the input, which needs no gradient: nothing to overlap with); `twice` never
    assert results[0][1] == 0 and results[1][1] == 5 and results[2][1] == 5
    for n, g in results[0][0].items():
        assert torch.equal(g, results[1][0][n]) and torch.equal(g, results[2][0][n]), n


def test_deferred_grouped_weight_gradients(monkeypatch):
    """GRIT_WGRAD_DEFER: inside a gradient-bucket wrapper's scope the single-use Linears of short maps return EMPTY weight / bias
    gradients that the wrapper fills with one grouped launch before it packs them (grit_wgrad_small_grouped + one grouped slab
    sum).  Same gradients as the per-node library path within bf16 accumulation-order noise; nothing pending afterwards; a
    Linear that is (wrongly) declared single-use but applied twice stops the run with an error instead of a wrong gradient."""
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.lib import GritHipError
    from grit_amd.ops import linear as L
    from grit_amd.ops.layer_norm import linear_add_layer_norm

    class Net(torch.nn.Module):
        def __init__(self, lie=False):
            super().__init__()
            self.a = L.Linear(512, 1024)
            self.b = torch.nn.Linear(1024, 512)
            self.c = L.Linear(512, 128)
            self.norm = torch.nn.LayerNorm(512)
            self.twice = L.Linear(512, 512)
            L.mark_single_use(self.a, self.b, self.c)
            if lie:
                L.mark_single_use(self.twice)

        def forward(self, x):
            h = torch.relu(self.a(x))
            x = linear_add_layer_norm(h, self.b, x, None, self.norm.weight, self.norm.bias, 1e-5, 0.0, True)[1]
            x = self.twice(torch.tanh(self.twice(x)))
            return x, self.c(x)

    torch.manual_seed(0)
    net = Net().to(DEV).bfloat16()
    x = torch.randn(8, 600, 512, device=DEV).bfloat16().requires_grad_(True)
    cot = torch.randn(8, 600, 512, device=DEV).bfloat16()
    cot2 = torch.randn(8, 600, 128, device=DEV).bfloat16()
    ddp = BucketedDataParallel(net, bucket_mb=1)
    results = []
    for knob in (False, True):
        monkeypatch.setattr(L, "WGRAD_DEFER", knob)
        x.grad = None
        y, z = ddp(x)
        ((y.float() * cot.float()).sum() + (z.float() * cot2.float()).sum()).backward()
        if knob:
            assert len(L._deferral["jobs"]) + sum(b.packed for b in ddp.buckets) > 0  # something was deferred (or already flushed)
        ddp.finish_gradient_sync()
        assert not L._deferral["jobs"] and not L._deferral["slabs"] and not L._deferral["unverified"] and not L._deferral["active"]
        torch.cuda.synchronize()
        results.append({n: p.grad.float().clone() for n, p in net.named_parameters()})
    for n, g in results[0].items():
        scale = g.abs().max().item()
        assert (g - results[1][n]).abs().max().item() <= 2e-2 * scale + 1e-6, n
    # the lie: `twice` is applied twice but declared single-use
    monkeypatch.setattr(L, "WGRAD_DEFER", True)
    bad = BucketedDataParallel(Net(lie=True).to(DEV).bfloat16(), bucket_mb=1)
    y, z = bad(x)
    with pytest.raises(GritHipError, match="second gradient"):
        ((y.float() * cot.float()).sum() + (z.float() * cot2.float()).sum()).backward()
        bad.finish_gradient_sync()
    L.abandon_deferred()
    L.end_deferral()
    assert not L._deferral["unverified"] and not L._deferral["slabs"]


def test_packed_in_projection_node(monkeypatch):
    """ops.linear.packed_in_proj (the decoder layers' nn.MultiheadAttention in-projection with q = k = tgt + pos, v = tgt, reference
    models/detection/det_module.py:313-326): one node over the packed [3E, E] parameter.  Inside a gradient-bucket scope its two row
    ranges are two problems of the scope's grouped weight-gradient launch, written into the parameter's bucket slot; outside they are
    computed by the node.  Both against the split-weights form (two Linear nodes): outputs bit for bit, gradients within bf16
    accumulation-order noise; nothing pending afterwards."""
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.ops import linear as L

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mha = torch.nn.MultiheadAttention(512, 8)
            self.packed = True

        def forward(self, t, pos):
            w, b, E = self.mha.in_proj_weight, self.mha.in_proj_bias, 512
            if self.packed:
                return L.packed_in_proj(t + pos, t, w, b)
            w_qk, w_v = w.split([2 * E, E])
            b_qk, b_v = b.split([2 * E, E])
            return L.linear(t + pos, w_qk, b_qk), L.linear(t, w_v, b_v)

    torch.manual_seed(1)
    net = Net().to(DEV).bfloat16()
    with torch.no_grad():
        net.mha.in_proj_bias.normal_(0, 0.1)
    t = torch.randn(32, 150, 512, device=DEV).bfloat16().requires_grad_(True)
    pos = torch.randn(32, 150, 512, device=DEV).bfloat16()
    c_qk = torch.randn(32, 150, 1024, device=DEV).bfloat16()
    c_v = torch.randn(32, 150, 512, device=DEV).bfloat16()
    results = {}
    for name, packed, wrap in (("split", False, False), ("node", True, False), ("bucket", True, True)):
        net.packed = packed
        for p_ in net.parameters():
            p_.grad = None
        t.grad = None
        model = BucketedDataParallel(net, bucket_mb=1) if wrap else net
        qk, v = model(t, pos)
        ((qk.float() * c_qk.float()).sum() + (v.float() * c_v.float()).sum()).backward()
        if wrap:
            assert len(L._deferral["jobs"]) + sum(b.packed for b in model.buckets) > 0
            model.finish_gradient_sync()
        assert not L._deferral["jobs"] and not L._deferral["unverified"] and not L._deferral["active"]
        torch.cuda.synchronize()
        results[name] = (qk.detach(), v.detach(), t.grad.float().clone(), net.mha.in_proj_weight.grad.float().clone(),
                         net.mha.in_proj_bias.grad.float().clone())
    for name in ("node", "bucket"):
        assert torch.equal(results[name][0], results["split"][0]) and torch.equal(results[name][1], results["split"][1])
        for got, ref in zip(results[name][2:], results["split"][2:]):
            assert (got - ref).abs().max().item() <= 2e-2 * ref.abs().max().item() + 1e-6, name


def test_long_map_weight_gradients_are_written_into_their_bucket_slots(monkeypatch):
    """Inside a gradient-bucket scope the weight gradients of long maps (own GEMM + slab sum) are summed straight into the parameter's
    slot of the flat bucket (ops.linear.grad_slot): autograd adopts the view, the bucket pack finds the gradient in place and does
    not copy it.  Same gradients as with the knob off; fewer elements through _pack's multi-tensor copy."""
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.ops import linear as L

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.a = L.Linear(512, 512)
            self.b = L.Linear(512, 256)

        def forward(self, x):
            return self.b(torch.tanh(self.a(x)))

    torch.manual_seed(0)
    net = Net().to(DEV).bfloat16()
    x = torch.randn(16, 1024, 512, device=DEV).bfloat16()          # 16 384 rows: a long map
    cot = torch.randn(16, 1024, 256, device=DEV).bfloat16()
    ddp = BucketedDataParallel(net, bucket_mb=64)
    copied, results = [], []
    real = torch._foreach_copy_

    def spy(dst, src, *a, **k):
        copied[-1] += sum(t.numel() for t in dst)
        return real(dst, src, *a, **k)

    monkeypatch.setattr(torch, "_foreach_copy_", spy)
    for knob in (False, True):
        monkeypatch.setattr(L, "GRAD_IN_PLACE", knob)
        copied.append(0)
        y = ddp(x)
        (y.float() * cot.float()).sum().backward()
        if knob:
            slot = net.a.weight._grit_grad_slot
            assert net.a.weight.grad.data_ptr() == slot[0].data_ptr() + slot[1] * 2  # already in its bucket before the pack
        ddp.finish_gradient_sync()
        torch.cuda.synchronize()
        results.append({n: p.grad.float().clone() for n, p in net.named_parameters()})
    for n, g in results[0].items():
        assert torch.equal(g, results[1][n]), n
    assert copied[1] <= copied[0] - net.a.weight.numel() - net.b.weight.numel()


@pytest.mark.gpu
def test_parked_weight_gradient_runs_in_its_partners_launch(monkeypatch):
    """park_weight_grad_for_partner: inside a gradient-bucket scope the long-map weight gradient of the first Linear (single-use) is
    not launched by its own node; the partner's node, reached next by backward, computes both in one grouped launch of the long-map
    kernel.  Same gradients as with the knob off (fp32 slice sums: a different number of row slices, so bf16-rounding close, not
    bit-equal); nothing stays parked; a parked job whose partner never runs is computed by the scope's flush."""
    from grit_amd.ddp import BucketedDataParallel
    from grit_amd.ops import linear as L
    from grit_amd.ops.layer_norm import linear_add_layer_norm

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.qkv = L.Linear(512, 1536)
            self.proj = L.Linear(512, 512)
            self.norm = torch.nn.LayerNorm(512)
            L.mark_single_use(self.proj)
            L.park_weight_grad_for_partner(self.proj, self.qkv)
            self.skip_partner = False

        def forward(self, x):
            h = x if self.skip_partner else torch.tanh(self.qkv(x)[..., :512])
            s, n = linear_add_layer_norm(h, self.proj, x, None, self.norm.weight, self.norm.bias, self.norm.eps)
            return s + n

    torch.manual_seed(0)
    net = Net().to(DEV).bfloat16()
    x = torch.randn(16, 1024, 512, device=DEV).bfloat16()          # 16 384 rows: a long map
    cot = torch.randn(16, 1024, 512, device=DEV).bfloat16()
    ddp = BucketedDataParallel(net, bucket_mb=64, tail_mb=0)  # one bucket: no pack (= flush) between the two nodes
    launches, results = [], []
    lib = L._lib.load()
    real = lib.grit_wgrad_tn_grouped

    class Spy(object):
        def __call__(self, table, n, stream):
            launches[-1].append(n)
            return real(table, n, stream)

    for knob, skip in ((False, False), (True, False), (False, True), (True, True)):
        monkeypatch.setattr(L, "WGRAD_PARK", knob)
        net.skip_partner = skip
        launches.append([])
        monkeypatch.setattr(lib, "grit_wgrad_tn_grouped", Spy(), raising=False)
        y = ddp(x)
        (y.float() * cot.float()).sum().backward()
        assert not L._deferral["parked"] or skip
        ddp.finish_gradient_sync()
        monkeypatch.setattr(lib, "grit_wgrad_tn_grouped", real, raising=False)
        assert not L._deferral["parked"]
        torch.cuda.synchronize()
        results.append({n: (None if p.grad is None else p.grad.float().clone()) for n, p in net.named_parameters()})
        for p in net.parameters():
            p.grad = None
    assert launches == [[], [2], [], [1]]   # partner + parked together; a leftover alone at the scope's flush
    for a, b in ((0, 1), (2, 3)):
        for n, g in results[a].items():
            if g is None:
                assert results[b][n] is None and n.startswith("qkv"), n
                continue
            scale = g.abs().max().clamp_min(1e-6)
            assert (g - results[b][n]).abs().max() <= 0.02 * scale, n


@pytest.mark.gpu
def test_grouped_weight_transposes_and_their_cache():
    """grit_transpose_bf16_grouped against torch (bit-exact: a copy), and ops.transposed: a copy is served only for the weight value
    it was made from (tensor version, data pointer, weights_epoch) -- anything else falls back to the node's own transpose."""
    from grit_amd.ops import transposed as T
    from grit_amd.ops import weights_epoch
    torch.manual_seed(0)
    ws = [torch.randn(r, c, device=DEV).bfloat16() for r, c in ((512, 2048), (128, 512), (1024, 4096), (64, 64), (256, 1024))]
    ws.append(torch.randn(100, 64, device=DEV).bfloat16())           # rows not a multiple of 64: never cached
    assert all(T.lookup(w) is None for w in ws)
    T.refresh(ws)
    for w in ws[:-1]:
        assert torch.equal(T.lookup(w), w.t().contiguous())
    assert T.lookup(ws[-1]) is None
    kept = T.lookup(ws[0])
    ws[0].add_(1.0)                                                   # visible in-place write: the copy is stale
    assert T.lookup(ws[0]) is None
    T.refresh(ws)
    assert T.lookup(ws[0]).data_ptr() == kept.data_ptr() and torch.equal(T.lookup(ws[0]), ws[0].t().contiguous())  # buffer re-used
    weights_epoch.bump()                                              # a raw-kernel write (flat optimizer step): everything is stale
    assert all(T.lookup(w) is None for w in ws)
    T.refresh(ws)
    assert torch.equal(T.lookup(ws[2]), ws[2].t().contiguous())


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 8, 12, 128, torch.bfloat16), (3, 20, 20, 256, torch.bfloat16), (1, 10, 6, 512, torch.float32),
                                           (32, 80, 80, 128, torch.bfloat16), (2, 4, 2, 1024, torch.bfloat16)])
def test_patch_merging_layer_norm_through_the_view(B, H, W, C, dtype):
    """grit_merge_layernorm_{fwd,bwd} (reference models/common/swin_model.py:279-288): LayerNorm(4C) of the 2 x 2 patch-merged view
    addressed directly against the materialised permute + reshape followed by the plain kernels: forward bit-identical (same rows,
    same arithmetic), input gradient equal element for element in the token-map layout, dgamma / dbeta equal within the order of the
    partial sums; odd sizes are not taken (the caller pads and copies)."""
    from grit_amd.ops.layer_norm import layer_norm, merge_layer_norm
    g = torch.Generator(device=DEV).manual_seed(B * H + C)
    x = torch.randn(B, H * W, C, device=DEV, generator=g).to(dtype)
    w = (1.0 + 0.1 * torch.randn(4 * C, device=DEV, generator=g)).to(dtype)
    b = (0.1 * torch.randn(4 * C, device=DEV, generator=g)).to(dtype)
    cot = torch.randn(B, (H // 2) * (W // 2), 4 * C, device=DEV, generator=g).to(dtype)
    outs = []
    for fused in (True, False):
        xi, wi, bi = (t_.clone().requires_grad_(True) for t_ in (x, w, b))
        if fused:
            y = merge_layer_norm(xi, H, W, wi, bi, 1e-5)
            assert y is not None
        else:
            v = xi.view(B, H // 2, 2, W // 2, 2, C).permute(0, 1, 3, 4, 2, 5).reshape(B, (H // 2) * (W // 2), 4 * C)
            y = layer_norm(v, wi, bi, 1e-5)
        (y.float() * cot.float()).sum().backward()
        outs.append((y.detach(), xi.grad, wi.grad, bi.grad))
    (y1, dx1, dw1, db1), (y0, dx0, dw0, db0) = outs
    assert torch.equal(y1, y0)
    assert torch.equal(dx1, dx0)
    tol = 2e-2 if dtype == torch.bfloat16 else 1e-4
    assert float((dw1.float() - dw0.float()).abs().max()) <= tol * float(dw0.float().abs().max()) + 1e-6
    assert float((db1.float() - db0.float()).abs().max()) <= tol * float(db0.float().abs().max()) + 1e-6
    assert merge_layer_norm(x[:, :(H - 1) * W].contiguous(), H - 1, W, w, b, 1e-5) is None  # odd height: not taken


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W,C,img_dtype", [(2, 64, 128, 128, torch.float32), (1, 128, 64, 96, torch.bfloat16), (3, 32, 192, 192, torch.float32),
                                               (32, 640, 640, 128, torch.float32)])
def test_patch_embedding_in_one_pass(B, H, W, C, img_dtype):
    """grit_patch_embed_ln_fwd (reference models/common/swin_model.py:336-365): conv 4 x 4 / 4 + bias + LayerNorm against the
    reference composition evaluated in fp32 on the same bf16-rounded image and weights, and against this repository's unfused path
    (cast, im2col copy, GEMM, LayerNorm kernels): the bf16 results may differ by the rounding of differently ordered fp32 sums
    (one bf16 ulp of the conv output moves the normalised value by ~2^-8 of its scale)."""
    from grit_amd.models.common.swin_model import PatchEmbed
    from grit_amd.ops.layer_norm import LayerNorm
    import grit_amd.models.common.swin_model as SM
    torch.manual_seed(C + H)
    pe = PatchEmbed(patch_size=4, in_chans=3, embed_dim=C, norm_layer=LayerNorm).to(DEV).to(torch.bfloat16)
    with torch.no_grad():
        pe.norm.weight.copy_(1.0 + 0.1 * torch.randn(C, device=DEV))
        pe.norm.bias.copy_(0.1 * torch.randn(C, device=DEV))
    for p in pe.parameters():
        p.requires_grad_(False)
    img = torch.randn(B, 3, H, W, device=DEV).to(img_dtype)
    with torch.no_grad():
        fused, Wh, Ww = pe.tokens(img)
        assert (Wh, Ww) == (H // 4, W // 4) and fused.shape == (B, Wh * Ww, C) and fused.dtype == torch.bfloat16
        unfused = None
        if C == 128:  # (the streaming LayerNorm kernels of the unfused path cover the widths GRIT uses: 128 ... 4096)
            old = SM._PATCH_EMBED_FUSED
            SM._PATCH_EMBED_FUSED = False
            try:
                unfused, _, _ = pe.tokens(img)
            finally:
                SM._PATCH_EMBED_FUSED = old
        x32 = img.to(torch.bfloat16).float()
        conv = torch.nn.functional.conv2d(x32, pe.proj.weight.float(), pe.proj.bias.float(), stride=4)
        tok = conv.flatten(2).transpose(1, 2).to(torch.bfloat16).float()
        ref = torch.nn.functional.layer_norm(tok, (C,), pe.norm.weight.float(), pe.norm.bias.float(), pe.norm.eps)
    err = (fused.float() - ref).abs()
    assert float(err.max()) <= 0.06 and float(err.mean()) <= 4e-3, (float(err.max()), float(err.mean()))
    if unfused is not None:
        d = (fused.float() - unfused.float()).abs()
        assert float(d.max()) <= 0.06 and float((d > 0).float().mean()) < 0.2, (float(d.max()), float((d > 0).float().mean()))
    # a width outside the kernel's tiling keeps the GEMM path
    with torch.no_grad():
        assert pe._fused_tokens(img[..., :W - 4].contiguous()) is None
