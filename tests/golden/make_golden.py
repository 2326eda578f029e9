"""Generate the golden fixtures in tests/golden/ from the *imported Python reference*.

Runs ONLY in the build container (needs /root/reference; the GPU box never sees it):

    python tests/golden/make_golden.py [--only g1,g2,...]

The reference cannot travel, so what is committed is data only: seeded inputs + the outputs the
reference's own code produced for them.  Import-time stubs: the CUDA extension
`MultiScaleDeformableAttention` is replaced by the reference's own pure-PyTorch statement
`ms_deform_attn_core_pytorch` (models/ops/functions/ms_deform_attn_func.py:41-61); `timm.models.layers`
by DropPath=identity / to_2tuple / trunc_normal_; `torchvision` by a version object.

Fixtures
  msda_g1.npz   reference test shapes + seed (models/ops/test.py:21-36,85): fwd double/float, grads D in {30,32,64,71}
  msda_g2.npz   GRIT-shaped mini case with out-of-range and exactly-on-border points, fwd + grads
  (model-level fixtures g3..g8 are added by the functions further down)
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def import_reference():
    """Put the reference on sys.path behind the three import-time stubs; returns nothing."""
    if "models" in sys.modules and getattr(sys.modules["models"], "__file__", "").startswith(REF):
        return
    assert os.path.isdir(REF), "reference tree not present: golden vectors can only be made in the build container"
    for k in [k for k in sys.modules if k.split(".")[0] in ("models", "engine", "utils", "datasets")]:
        del sys.modules[k]
    sys.path.insert(0, REF)

    msda = types.ModuleType("MultiScaleDeformableAttention")
    sys.modules["MultiScaleDeformableAttention"] = msda

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):  # eval-time identity; fixtures are made in eval()/p=0
        def __init__(self, p=0.0):
            super().__init__()
            self.drop_prob = p

        def forward(self, x):
            return x

    timm_layers.DropPath = DropPath
    timm_layers.to_2tuple = lambda x: tuple(x) if isinstance(x, (tuple, list)) else (x, x)
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models = timm_models
    timm_models.layers = timm_layers
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers})

    tv = types.ModuleType("torchvision")
    tv.__version__ = "0.15.0"
    sys.modules["torchvision"] = tv

    # route the autograd Function through the reference's own differentiable PyTorch statement
    import models.ops.functions.ms_deform_attn_func as f

    class _Fn:
        @staticmethod
        def apply(value, shapes, lsi, loc, aw, im2col_step):
            return f.ms_deform_attn_core_pytorch(value, shapes, loc, aw)

    import models.ops.modules.ms_deform_attn as m
    m.MSDeformAttnFunction = _Fn
    f.MSDeformAttnFunction = _Fn


def _lsi(shapes):
    return torch.cat((shapes.new_zeros((1,)), shapes.prod(1).cumsum(0)[:-1]))


def _ref_fwd_bwd(value, shapes, loc, aw, cot):
    """Reference forward + autograd grads in the dtype of `value` (grid_sample is differentiable)."""
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    v = value.clone().requires_grad_(True)
    l = loc.clone().requires_grad_(True)
    a = aw.clone().requires_grad_(True)
    out = ms_deform_attn_core_pytorch(v, shapes, l, a)
    gv, gl, ga = torch.autograd.grad(out, (v, l, a), cot)
    return out.detach(), gv, gl, ga


def make_g1():
    """models/ops/test.py: N=1 M=2 D=2 Lq=2 L=2 P=2, shapes (6,4),(3,2), torch.manual_seed(3), same draw order."""
    import_reference()
    from models.ops.functions.ms_deform_attn_func import ms_deform_attn_core_pytorch
    N, M, D, Lq, L, P = 1, 2, 2, 2, 2, 2
    shapes = torch.as_tensor([(6, 4), (3, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    torch.manual_seed(3)
    out = {"shapes": shapes.numpy(), "lsi": _lsi(shapes).numpy()}

    def draw(d):
        value = torch.rand(N, S, M, d) * 0.01
        loc = torch.rand(N, Lq, M, L, P, 2)
        aw = torch.rand(N, Lq, M, L, P) + 1e-5
        aw /= aw.sum(-1, keepdim=True).sum(-2, keepdim=True)
        return value, loc, aw

    # check_forward_equal_with_pytorch_double (test.py:31-44)
    value, loc, aw = draw(D)
    out.update(dbl_value=value.numpy(), dbl_loc=loc.numpy(), dbl_aw=aw.numpy(),
               dbl_out=ms_deform_attn_core_pytorch(value.double(), shapes, loc.double(), aw.double()).numpy())
    # check_forward_equal_with_pytorch_float (test.py:47-60)
    value, loc, aw = draw(D)
    out.update(flt_value=value.numpy(), flt_loc=loc.numpy(), flt_aw=aw.numpy(),
               flt_out=ms_deform_attn_core_pytorch(value, shapes, loc, aw).numpy())
    # check_gradient_numerical for the four channel counts (test.py:63-86): same draws, analytic grads
    # of <out, cot> in double through the reference's PyTorch statement
    cg = torch.Generator().manual_seed(1234)
    for d in (30, 32, 64, 71):
        value, loc, aw = draw(d)
        cot = torch.randn(N, Lq, M * d, generator=cg, dtype=torch.float64)
        o, gv, gl, ga = _ref_fwd_bwd(value.double(), shapes, loc.double(), aw.double(), cot)
        out.update({f"g{d}_value": value.numpy(), f"g{d}_loc": loc.numpy(), f"g{d}_aw": aw.numpy(),
                    f"g{d}_cot": cot.numpy(), f"g{d}_out": o.numpy(), f"g{d}_gv": gv.numpy(),
                    f"g{d}_gl": gl.numpy(), f"g{d}_ga": ga.numpy()})
    np.savez_compressed(os.path.join(HERE, "msda_g1.npz"), **out)
    print("g1 first outputs", out["dbl_out"].ravel()[:4])


def make_g2():
    """GRIT-shaped mini case: M=8, D=64, L=P=4; points outside the map and exactly on every border."""
    import_reference()
    g = torch.Generator().manual_seed(20)
    B, M, D, Lq, L, P = 2, 8, 64, 20, 4, 4
    shapes = torch.as_tensor([(6, 7), (4, 4), (3, 2), (2, 2)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    value = torch.randn(B, S, M, D, generator=g)
    loc = torch.rand(B, Lq, M, L, P, 2, generator=g) * 1.2 - 0.1
    aw = torch.softmax(torch.randn(B, Lq, M, L * P, generator=g), -1).view(B, Lq, M, L, P)
    # hand-placed coordinates: h_im/w_im in {-1, -1+eps, -0.5, 0, H-1, H-0.5, H-eps, H} per level
    for l in range(L):
        H, W = [int(x) for x in shapes[l]]
        xs = [-0.5 / W, (-0.5 + 1e-3) / W, 0.0, 0.5 / W, (W - 0.5) / W, 1.0, (W + 0.5 - 1e-3) / W, (W + 0.5) / W]
        ys = [-0.5 / H, (-0.5 + 1e-3) / H, 0.0, 0.5 / H, (H - 0.5) / H, 1.0, (H + 0.5 - 1e-3) / H, (H + 0.5) / H]
        for i, (x, y) in enumerate(zip(xs, ys)):
            loc[0, i, :, l, 0] = torch.tensor([x, y])          # both on the border
            loc[0, i, :, l, 1] = torch.tensor([x, 0.37])       # x on the border only
            loc[1, i, :, l, 2] = torch.tensor([0.61, y])       # y on the border only
    loc = loc.float()
    cot = torch.randn(B, Lq, M * D, generator=g)
    o, gv, gl, ga = _ref_fwd_bwd(value.double(), shapes, loc.double(), aw.double(), cot.double())
    np.savez_compressed(os.path.join(HERE, "msda_g2.npz"), shapes=shapes.numpy(), lsi=_lsi(shapes).numpy(),
                        value=value.numpy(), loc=loc.numpy(), aw=aw.numpy(), cot=cot.numpy(),
                        out=o.float().numpy(), gv=gv.float().numpy(), gl=gl.float().numpy(),
                        ga=ga.float().numpy())
    print("g2 out", tuple(o.shape), float(o.abs().mean()))


MAKERS = {"g1": make_g1, "g2": make_g2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    torch.set_num_threads(8)
    names = [n for n in args.only.split(",") if n] or list(MAKERS)
    for n in names:
        MAKERS[n]()


if __name__ == "__main__":
    main()
