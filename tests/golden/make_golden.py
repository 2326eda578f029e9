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


# =====================================================================================================
# model-level fixtures (g3..g8) -- appended makers; MAKERS is extended at the bottom
# =====================================================================================================
def _fill():
    sys.path.insert(0, HERE)
    import fill
    return fill.deterministic_fill_


def _cfg(**over):
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from grit_amd.config import default_config
    return default_config(**over)


def make_keys():
    """State-dict surface: key -> shape of the reference model at 3 and 2 decoder layers."""
    import json
    import_reference()
    from models.caption import Transformer
    from models.caption.detector import build_detector
    out = {}
    for n in (3, 2):
        cfg = _cfg(**{'model.cap_generator.n_layers': n})
        model = Transformer(build_detector(cfg), cfg)
        out[str(n)] = {k: list(v.shape) for k, v in model.state_dict().items()}
        out[f'{n}_trainable'] = sorted(k for k, p in model.named_parameters() if p.requires_grad)
    with open(os.path.join(HERE, 'state_dict_keys.json'), 'w') as f:
        json.dump(out, f)
    print('keys', {k: len(v) for k, v in out.items()})


def make_g3():
    """MSDeformAttn module, 2-d and 4-d reference points, padding mask (ms_deform_attn.py:73-119)."""
    import_reference()
    from models.ops.modules import MSDeformAttn
    g = torch.Generator().manual_seed(33)
    mod = _fill()(MSDeformAttn(d_model=128, n_levels=3, n_heads=4, n_points=4), 'g3.').double()
    shapes = torch.as_tensor([(7, 9), (4, 5), (2, 3)], dtype=torch.long)
    S = int(shapes.prod(1).sum())
    B, Lq = 2, 11
    query = torch.randn(B, Lq, 128, generator=g)
    src = torch.randn(B, S, 128, generator=g)
    ref2 = torch.rand(B, Lq, 3, 2, generator=g)
    ref4 = torch.cat([torch.rand(B, Lq, 3, 2, generator=g), 0.1 + 0.4 * torch.rand(B, Lq, 3, 2, generator=g)], -1)
    pad = torch.zeros(B, S, dtype=torch.bool)
    pad[1, -9:] = True
    with torch.no_grad():
        out2 = mod(query.double(), ref2.double(), src.double(), shapes, _lsi(shapes), None)
        out4 = mod(query.double(), ref4.double(), src.double(), shapes, _lsi(shapes), pad)
    np.savez_compressed(os.path.join(HERE, 'msda_module_g3.npz'), shapes=shapes.numpy(), lsi=_lsi(shapes).numpy(),
                        query=query.numpy(), src=src.numpy(), ref2=ref2.numpy(), ref4=ref4.numpy(), pad=pad.numpy(),
                        out2=out2.float().numpy(), out4=out4.float().numpy())
    print('g3', float(out2.abs().mean()), float(out4.abs().mean()))


def make_g4():
    """WindowAttention (dim 128, 4 heads, window 12) without / with the BasicLayer shift mask at H=W=20 (-> 24), and a
    whole 2-block BasicLayer (no shift + shift, pad + crop, PatchMerging) on the same 20x20 map (covers G5)."""
    import_reference()
    from models.common.swin_model import BasicLayer, PatchMerging, WindowAttention, window_partition
    fill = _fill()
    g = torch.Generator().manual_seed(44)
    layer = BasicLayer(dim=128, depth=2, num_heads=4, window_size=12, drop_path=[0.0, 0.1], downsample=PatchMerging)
    fill(layer, 'g4.')
    layer.eval()
    attn = layer.blocks[1].attn
    xw = torch.randn(4, 144, 128, generator=g)  # 1 image x 4 windows
    # the mask exactly as BasicLayer.forward builds it (swin_model.py:424-441)
    Hp = Wp = 24
    img_mask = torch.zeros((1, Hp, Wp, 1))
    cnt = 0
    for h in (slice(0, -12), slice(-12, -6), slice(-6, None)):
        for w in (slice(0, -12), slice(-12, -6), slice(-6, None)):
            img_mask[:, h, w, :] = cnt
            cnt += 1
    mw = window_partition(img_mask, 12).view(-1, 144)
    am = mw.unsqueeze(1) - mw.unsqueeze(2)
    am = am.masked_fill(am != 0, float(-100.0)).masked_fill(am == 0, float(0.0))
    x = torch.randn(2, 400, 128, generator=g)
    with torch.no_grad():
        o_nomask = attn(xw, None)
        o_mask = attn(xw, am)
        x_out, H, W, x_down, Wh, Ww = layer(x, 20, 20)
    np.savez_compressed(os.path.join(HERE, 'win_g4.npz'), xw=xw.numpy(), attn_mask=am.numpy(), o_nomask=o_nomask.numpy(),
                        o_mask=o_mask.numpy(), x=x.numpy(), x_out=x_out.numpy(), x_down=x_down.numpy(),
                        dims=np.array([H, W, Wh, Ww]))
    print('g4', float(o_nomask.abs().mean()), float(x_out.abs().mean()), (H, W, Wh, Ww))


def make_g6():
    """ParallelAttentionLayer with PAD tokens: pins the fc_alpha1-twice quirk (cap_generator.py:40-56) and
    MultiHeadAttention / FeedForward (attention.py:166-184, pos_embed.py:44-48)."""
    import_reference()
    from models.caption.cap_generator import ParallelAttentionLayer
    g = torch.Generator().manual_seed(66)
    layer = _fill()(ParallelAttentionLayer(512, 8, 2048, dropout=0.1), 'g6.').eval()
    B, T, N1, N2 = 2, 7, 9, 12
    x = torch.randn(B, T, 512, generator=g)
    y1 = torch.randn(B, N1, 512, generator=g)
    y2 = torch.randn(B, N2, 512, generator=g)
    tokens = torch.tensor([[2, 5, 6, 7, 8, 9, 3], [2, 11, 12, 3, 1, 1, 1]])
    pad = tokens == 1
    mask_pad = (~pad).unsqueeze(-1).float()
    mask_x = (torch.triu(torch.ones(T, T, dtype=torch.uint8), diagonal=1)[None, None] + pad[:, None, None].byte()).gt(0)
    mask_y1 = torch.zeros(B, 1, 1, N1, dtype=torch.bool)
    mask_y1[1, ..., -3:] = True
    mask_y2 = torch.zeros(B, 1, 1, N2, dtype=torch.bool)
    with torch.no_grad():
        out = layer(x, y1, y2, mask_pad, mask_x, mask_y1, mask_y2)
    np.savez_compressed(os.path.join(HERE, 'attn_g6.npz'), x=x.numpy(), y1=y1.numpy(), y2=y2.numpy(),
                        tokens=tokens.numpy(), mask_pad=mask_pad.numpy(), mask_x=mask_x.numpy(),
                        mask_y1=mask_y1.numpy(), mask_y2=mask_y2.numpy(), out=out.numpy())
    print('g6', float(out.abs().mean()))


def _ref_model(n_layers, **over):
    import_reference()
    from models.caption import Transformer
    from models.caption.detector import build_detector
    cfg = _cfg(**{'model.cap_generator.n_layers': n_layers, **over})
    model = Transformer(build_detector(cfg), cfg)
    _fill()(model)
    return model, cfg


def make_g7():
    """BASELINE config 1: one 224x224 image, 2-layer decoder, deterministic-fill weights, eval mode, CPU.
    Teacher-forcing log-probs (top-16 per position), greedy (beam 1) and beam-5 tokens + per-step top-6 candidate
    scores (margin record), and the detector outputs for decoder-only checks."""
    import_reference()
    from engine.utils import NestedTensor
    model, cfg = _ref_model(2)
    model.eval()
    g = torch.Generator().manual_seed(0)
    image = torch.randn(1, 3, 224, 224, generator=g)
    samples = NestedTensor(image, torch.zeros(1, 224, 224, dtype=torch.bool))
    seq = torch.randint(4, 10201, (1, 20), generator=g)
    seq[:, 0], seq[:, -1] = 2, 3
    out = {'image': image.numpy(), 'seq': seq.numpy()}
    with torch.no_grad():
        vis = model.detector(samples)
        out.update(gri_feat=vis['gri_feat'].numpy(), reg_feat=vis['reg_feat'].numpy(),
                   gri_mask=vis['gri_mask'].numpy(), reg_mask=vis['reg_mask'].numpy())
        lp = model(samples, seq)
        top = lp.topk(16, -1)
        out.update(tf_top_val=top.values.numpy(), tf_top_idx=top.indices.numpy(),
                   tf_target_lp=lp[0, :-1].gather(1, seq[0, 1:, None]).numpy(), tf_row_mean=lp.mean(-1).numpy())
        for beam in (1, 5):
            record = []
            orig = model.select

            def select(t, cand, beam_size, _orig=orig, _rec=record, **kw):
                flat = cand.reshape(cand.shape[0], -1)
                _rec.append(torch.sort(flat, -1, descending=True)[0][:, :beam_size + 1].clone())
                return _orig(t, cand, beam_size, **kw)

            model.select = select
            toks, lps = model(samples, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=beam, out_size=1)
            model.select = orig
            top6 = torch.stack(record, 1)  # [B, steps, beam+1]
            margin = (top6[..., :-1] - top6[..., 1:]).abs().min().item()
            out.update({f'beam{beam}_tokens': toks.numpy(), f'beam{beam}_logprobs': lps.numpy(),
                        f'beam{beam}_top': top6.numpy()})
            print(f'g7 beam{beam}', toks.tolist(), 'min margin between consecutive candidates', margin)
    np.savez_compressed(os.path.join(HERE, 'model_g7.npz'), **out)


def make_g8():
    """One XE step (train mode, every dropout p = 0, DropPath = identity): loss, per-module gradient norms, the set of
    parameters that receive no gradient (static unused set), ragged batch with PAD tokens and masks."""
    import json
    import_reference()
    from engine.utils import NestedTensor
    model, cfg = _ref_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    g = torch.Generator().manual_seed(8)
    B, T = 2, 12
    images = torch.randn(B, 3, 224, 224, generator=g)
    mask = torch.zeros(B, 224, 224, dtype=torch.bool)
    images[1, :, 192:, :] = 0
    images[1, :, :, 160:] = 0
    mask[1, 192:, :] = True
    mask[1, :, 160:] = True
    caps = torch.randint(4, 10201, (B, T), generator=g)
    caps[:, 0] = 2
    caps[0, -1] = 3
    caps[1, 7] = 3
    caps[1, 8:] = 1
    out = model(NestedTensor(images, mask), caps)
    loss = torch.nn.NLLLoss(ignore_index=1)(out[:, :-1].reshape(-1, out.shape[-1]), caps[:, 1:].reshape(-1))
    loss.backward()
    norms, nograd = {}, []
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if p.grad is None:
            nograd.append(n)
            continue
        top = '.'.join(n.split('.')[:2]) if n.startswith('detector') else n.split('.')[0]
        norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    norms = {k: v**0.5 for k, v in norms.items()}
    sel = {}
    for n in ('cap_generator.fc.weight', 'grid_net.fc.weight', 'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight',
              'detector.backbone.layers.2.blocks.17.attn.relative_position_bias_table',
              'detector.backbone.layers.1.blocks.1.attn.qkv.bias', 'detector.input_proj.0.0.weight'):
        p = dict(model.named_parameters())[n]
        sel[n] = p.grad.flatten()[:64].numpy()
    np.savez_compressed(os.path.join(HERE, 'step_g8.npz'), images=images.numpy(), mask=mask.numpy(), caps=caps.numpy(),
                        loss=np.array(loss.item()), **{'grad:' + k: v for k, v in sel.items()})
    with open(os.path.join(HERE, 'step_g8.json'), 'w') as f:
        json.dump({'loss': loss.item(), 'grad_norms': norms, 'no_grad': sorted(nograd)}, f, indent=1)
    print('g8 loss', loss.item(), 'unused', len(nograd), norms)


def make_g9():
    """One self-critical step (reference engine/caption_engine.py:421-443) on the reference model: train mode, every
    dropout p = 0, beam search WITH gradient (beam 3, 6 steps, out_size = beam), a fixed reward tensor in place of the
    host-side CIDEr, loss = mean(-mean_t(log_probs) * (reward - mean_beam(reward))), backward.  Stored: beam tokens,
    their log-probs, the loss, per-module gradient norms and slices of a few gradients."""
    import json
    import_reference()
    from engine.utils import NestedTensor
    model, cfg = _ref_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    g = torch.Generator().manual_seed(9)
    B, beam, T = 2, 3, 6
    images = torch.randn(B, 3, 224, 224, generator=g)
    mask = torch.zeros(B, 224, 224, dtype=torch.bool)
    reward = torch.rand(B, beam, generator=g)
    outs, log_probs = model(NestedTensor(images, mask), seq=None, use_beam_search=True, max_len=T, eos_idx=3,
                            beam_size=beam, out_size=beam, return_probs=False)
    baseline = torch.mean(reward, -1, keepdim=True)
    loss = (-torch.mean(log_probs, -1) * (reward - baseline)).mean()
    loss.backward()
    norms = {}
    for n, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            top = '.'.join(n.split('.')[:2]) if n.startswith('detector') else n.split('.')[0]
            norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    norms = {k: v**0.5 for k, v in norms.items()}
    sel = {}
    for n in ('cap_generator.fc.weight', 'grid_net.fc.weight', 'cap_generator.layers.0.vis_att1.attention.fc_k.weight',
              'detector.det_module.decoder_layers.5.cross_attn.value_proj.weight', 'detector.input_proj.0.0.weight'):
        sel[n] = dict(model.named_parameters())[n].grad.flatten()[:64].numpy()
    np.savez_compressed(os.path.join(HERE, 'sc_g9.npz'), images=images.numpy(), reward=reward.numpy(),
                        tokens=outs.numpy(), log_probs=log_probs.detach().numpy(), loss=np.array(loss.item()),
                        **{'grad:' + k: v for k, v in sel.items()})
    with open(os.path.join(HERE, 'sc_g9.json'), 'w') as f:
        json.dump({'loss': loss.item(), 'grad_norms': norms}, f, indent=1)
    print('g9 loss', loss.item(), 'tokens', outs.tolist(), norms)


def make_g10():
    """Caption strings (BASELINE config 1 / reference inference_caption.py:66-67): the reference's own TextField.decode
    (datasets/caption/field.py:258-283) applied to the G7 token ids with the reference's data/vocab.json.  The module
    is loaded from its file behind a synthetic package (its package __init__ pulls in torchvision / pycocotools) with
    empty stand-ins for the two imports it never uses while decoding (h5py, spacy).  Committed: the `itos` word list
    (data) and the expected strings."""
    import importlib
    import json
    pkg = types.ModuleType('refcap')
    pkg.__path__ = [os.path.join(REF, 'datasets', 'caption')]
    sys.modules['refcap'] = pkg
    sys.modules.setdefault('h5py', types.ModuleType('h5py'))
    sp = types.ModuleType('spacy')
    sp.load = lambda name: None
    sys.modules.setdefault('spacy', sp)
    field = importlib.import_module('refcap.field')
    tf = field.TextField(vocab_path=os.path.join(REF, 'data', 'vocab.json'))
    g = np.load(os.path.join(HERE, 'model_g7.npz'))
    extra = torch.tensor([[2, 4, 5, 3, 7, 7], [3, 9, 9, 9, 9, 9], [10200, 0, 1, 2, 4, 3]])  # eos first / unk / pad / bos
    out = {'itos': list(tf.vocab.itos), 'eos_token': tf.eos_token,
           'beam1': tf.decode(torch.from_numpy(g['beam1_tokens']), join_words=True),
           'beam5': tf.decode(torch.from_numpy(g['beam5_tokens']), join_words=True),
           'extra_tokens': extra.tolist(), 'extra': tf.decode(extra, join_words=True)}
    with open(os.path.join(HERE, 'vocab_g10.json'), 'w') as f:
        json.dump(out, f)
    print('g10', out['beam1'], out['extra'])


def make_g11():
    """Image side of the batch contract.  The reference's own resize classes
    (datasets/caption/transforms/utils.py, loaded from the file: the package __init__ needs torchvision, absent here) are
    applied to seeded PIL images; ToTensor / Normalize are torchvision's published definitions restated with torch ops
    (`uint8 HWC -> CHW float32 .div(255)`, `.sub_(mean).div_(std)`, constants of transforms/__init__.py:6-7), and the
    padding + mask come from the reference's nested_tensor_from_tensor_list (engine/utils.py:278-295)."""
    import importlib.util
    from PIL import Image
    spec = importlib.util.spec_from_file_location('ref_transform_utils', os.path.join(REF, 'datasets/caption/transforms/utils.py'))
    tu = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tu)
    import_reference()
    from engine.utils import nested_tensor_from_tensor_list
    mean = torch.as_tensor([0.485, 0.456, 0.406], dtype=torch.float32)[:, None, None]
    std = torch.as_tensor([0.229, 0.224, 0.225], dtype=torch.float32)[:, None, None]
    rng = np.random.default_rng(11)
    shapes = [(60, 80), (75, 50), (64, 64), (31, 97), (120, 160)]
    out = {}
    for name, policy in (('maxwh', tu.MaxWHResize((48, 64))), ('minmax', tu.MinMaxResize((64, 128)))):
        tensors, sizes = [], []
        for i, (h, w) in enumerate(shapes):
            if i % 2 == 0:  # noise
                img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            else:  # smooth ramps with a sharp edge (overshoot -> exercises the clamp)
                y, x = np.mgrid[0:h, 0:w]
                img = np.stack([x * 255 // (w - 1), y * 255 // (h - 1), np.where(x > w // 2, 255, 0)], -1).astype(np.uint8)
            out['%s_in%d' % (name, i)] = img
            small = policy(Image.fromarray(img, 'RGB'))
            u8 = torch.from_numpy(np.asarray(small).copy())
            out['%s_u8_%d' % (name, i)] = u8.numpy()
            sizes.append(u8.shape[:2])
            tensors.append(u8.permute(2, 0, 1).to(torch.float32).div(255).sub_(mean).div_(std))
        nt = nested_tensor_from_tensor_list(tensors)
        out[name + '_sizes'] = np.asarray(sizes, np.int64)
        out[name + '_tensors'] = nt.tensors.numpy()
        out[name + '_mask'] = nt.mask.numpy()
        print('g11', name, sizes, tuple(nt.tensors.shape))
    np.savez_compressed(os.path.join(HERE, 'image_g11.npz'), **out)


MAKERS.update({'g10': make_g10, 'g11': make_g11})
MAKERS.update({'keys': make_keys, 'g3': make_g3, 'g4': make_g4, 'g6': make_g6, 'g7': make_g7, 'g8': make_g8, 'g9': make_g9})


def make_g12():
    """DetectionModule (models/detection/det_module.py:135-213) row level: 6 deformable decoder layers with box refinement
    (reference points are 4-d from the first layer on), ragged padding masks (valid ratios != 1, masked_fill of the value maps),
    from STORED level maps -> hs (all 7) and the reference boxes.  Dropout 0, eval mode; sampling offsets are filled ~ N(0,1) so
    the points leave the maps as well."""
    import_reference()
    from models.detection.det_module import build_det_module_with_config
    cfg = _cfg(**{'model.detector.dropout': 0.0})
    mod = build_det_module_with_config(cfg.model.detector)
    _fill()(mod, 'g12.')
    mod.eval()
    g = torch.Generator().manual_seed(12)
    B = 2
    shapes = [(12, 10), (6, 5), (3, 3), (2, 2)]
    valid = [[(12, 10), (6, 5), (3, 3), (2, 2)], [(9, 7), (5, 4), (3, 2), (2, 1)]]  # image 1 is smaller: padding on the right / bottom
    srcs, masks = [], []
    for l, (h, w) in enumerate(shapes):
        src = torch.randn(B, 512, h, w, generator=g)
        mask = torch.zeros(B, h, w, dtype=torch.bool)
        for b in range(B):
            vh, vw = valid[b][l]
            mask[b, vh:, :] = True
            mask[b, :, vw:] = True
        srcs.append(src)
        masks.append(mask)
    with torch.no_grad():
        hs, init_ref, inter_refs = mod(srcs, masks)
    out = {'hs_first': hs[1].numpy(), 'hs_last': hs[-1].numpy(), 'init_ref': init_ref.numpy(), 'inter_refs': inter_refs.numpy()}
    for l in range(4):
        out['src%d' % l] = srcs[l].numpy()
        out['mask%d' % l] = masks[l].numpy()
    np.savez_compressed(os.path.join(HERE, 'det_g12.npz'), **out)
    print('g12 hs', tuple(hs.shape), float(hs[-1].abs().mean()), 'refs', tuple(inter_refs.shape))


MAKERS.update({'g12': make_g12})


def make_g13():
    """One self-critical step from CACHED features (reference train_sc with model.cached_features = True,
    engine/caption_engine.py:421-443 + models/caption/transformer.py:138-142): the reference detector's outputs for the G9
    images are stored, the decoder (grid net + caption generator) runs the beam search with gradient on them.  On any device
    with fp32 decoder arithmetic the beams are the same, so loss and gradients can be asserted unconditionally."""
    import json
    import_reference()
    from engine.utils import NestedTensor
    model, cfg = _ref_model(3, **{'model.dropout': 0.0, 'model.detector.dropout': 0.0})
    model.train()
    g9 = np.load(os.path.join(HERE, 'sc_g9.npz'))
    images = torch.from_numpy(g9['images'])
    reward = torch.from_numpy(g9['reward'])
    B, beam, T = images.shape[0], reward.shape[1], 6
    with torch.no_grad():
        vis = model.detector(NestedTensor(images, torch.zeros(B, 224, 224, dtype=torch.bool)))
    feats = {k: v.clone() for k, v in vis.items()}
    model.cached_features = True
    outs, log_probs = model({k: v.clone() for k, v in feats.items()}, seq=None, use_beam_search=True, max_len=T, eos_idx=3,
                            beam_size=beam, out_size=beam, return_probs=False)
    baseline = torch.mean(reward, -1, keepdim=True)
    loss = (-torch.mean(log_probs, -1) * (reward - baseline)).mean()
    loss.backward()
    norms = {}
    for n, p in model.named_parameters():
        if p.requires_grad and p.grad is not None:
            top = n.split('.')[0]
            norms[top] = norms.get(top, 0.0) + float(p.grad.double().pow(2).sum())
    norms = {k: v**0.5 for k, v in norms.items()}
    sel = {}
    for n in ('cap_generator.fc.weight', 'grid_net.fc.weight', 'cap_generator.layers.0.vis_att1.attention.fc_k.weight',
              'cap_generator.layers.2.fc_alpha1.weight', 'grid_net.layers.2.pwff.fc2.bias'):
        sel[n] = dict(model.named_parameters())[n].grad.flatten()[:64].numpy()
    np.savez_compressed(os.path.join(HERE, 'sc_g13.npz'), reward=reward.numpy(), tokens=outs.numpy(),
                        log_probs=log_probs.detach().numpy(), loss=np.array(loss.item()),
                        **{'feat:' + k: v.numpy() for k, v in feats.items()}, **{'grad:' + k: v for k, v in sel.items()})
    with open(os.path.join(HERE, 'sc_g13.json'), 'w') as f:
        json.dump({'loss': loss.item(), 'grad_norms': norms}, f, indent=1)
    print('g13 loss', loss.item(), 'tokens', outs.tolist(), norms)


MAKERS.update({'g13': make_g13})


def make_g14():
    """CIDEr-D reward (reference datasets/caption/metrics/cider/*.py, pure Python: loaded from its files) on seeded synthetic
    captions: corpus statistics from a "training set" of 40 images x 5 references, then (a) the self-critical call form --
    5 beams per image against the image's references repeated per beam -- and (b) a call without corpus statistics."""
    import importlib
    import json
    pkg = types.ModuleType('refcider')
    pkg.__path__ = [os.path.join(REF, 'datasets', 'caption', 'metrics', 'cider')]
    sys.modules['refcider'] = pkg
    cider_mod = importlib.import_module('refcider.cider')
    rng = np.random.default_rng(14)
    vocab = ['a', 'the', 'man', 'woman', 'dog', 'cat', 'sitting', 'standing', 'on', 'in', 'of', 'with', 'red', 'blue', 'bench',
             'street', 'table', 'plate', 'food', 'two', 'and', 'next', 'to', 'holding', 'umbrella', 'riding', 'bike', 'field']

    def sentence():
        return ' '.join(rng.choice(vocab, size=int(rng.integers(4, 13))))

    train = {i: [sentence() for _ in range(5)] for i in range(40)}
    cider = cider_mod.Cider(train)
    B, beam = 6, 5
    refs = [[sentence() for _ in range(5)] for _ in range(B)]
    gts, res = {}, {}
    for b in range(B):
        for j in range(beam):
            hyp = refs[b][j % 5].split()
            if j:  # perturb: drop / repeat / swap words so that clipping and the length penalty matter
                hyp = hyp[: max(2, len(hyp) - j)] + list(rng.choice(vocab, size=j - 1))
            gts[b * beam + j] = refs[b]
            res[b * beam + j] = [' '.join(hyp)]
    mean, scores = cider.compute_score(gts, res)
    mean2, scores2 = cider_mod.Cider().compute_score(gts, res)
    out = {'train': {str(k): v for k, v in train.items()}, 'gts': {str(k): v for k, v in gts.items()},
           'res': {str(k): v for k, v in res.items()}, 'mean': float(mean), 'scores': [float(x) for x in scores],
           'mean_nocorpus': float(mean2), 'scores_nocorpus': [float(x) for x in scores2]}
    with open(os.path.join(HERE, 'cider_g14.json'), 'w') as f:
        json.dump(out, f)
    print('g14 mean', mean, 'first scores', scores[:4], 'no corpus', mean2)


MAKERS.update({'g14': make_g14})


def make_h5():
    """tests/golden/features_ref.h5: a miniature of the reference's feature cache (tools/extract_features.py:66-155) written by
    the HDF5 C library itself -- the same calls h5py's create_dataset issues (contiguous layout, int64 / float32 / the int8 enum
    {FALSE, TRUE} numpy bool maps to) -- through the generator tests/golden/make_features_ref_h5.c.  h5py is not installed here
    but an HDF5 1.10 library happens to sit under /opt/conda in the build image; the same library then reads a file written by
    grit_amd.datasets.caption.hdf5_min (h5diff: 0 differences)."""
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from grit_amd.datasets.caption import hdf5_min
    conda = '/opt/conda'
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, 'gen')
        subprocess.check_call(['gcc', os.path.join(HERE, 'make_features_ref_h5.c'), '-I' + conda + '/include', '-L' + conda + '/lib',
                               '-lhdf5', '-Wl,-rpath,' + conda + '/lib', '-o', exe])
        out = os.path.join(HERE, 'features_ref.h5')
        subprocess.check_call([exe, out])
        ref = hdf5_min.H5File(out)
        mine = os.path.join(tmp, 'mine.h5')
        w = hdf5_min.create(mine, {k: (v['shape'], np.bool_ if v['bool'] else v['dtype']) for k, v in ref.datasets.items()})
        for k in ref.keys():
            w[k][...] = ref[k]
        w.close()
        r = subprocess.run([conda + '/bin/h5diff', '-v', out, mine], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        print('h5: library file read back;', r.stdout.count('0 differences found'), 'objects identical under h5diff')


MAKERS.update({'h5': make_h5})



def make_g15():
    """Beam search that REACHES EOS, with a real batch (reference models/caption/transformer.py:204-254: the `seq_mask`
    product, the -999 fill, the index-0 survivor of a finished beam; :184-188 the full descending sort).  The closed-form fill
    never emits token 3 (G7 / G9 / G13 contain none), so row 3 of `cap_generator.fc.weight` is scaled: x6 makes the eight images
    finish at different steps (one never, two with all five beams finished), x10 adds EOS as the very first word.  Decoder only
    (model.cached_features = True) from seeded features with RAGGED grid masks (padded grid tokens zeroed, as the detector's
    padded positions are masked keys); 3-layer decoder = BASELINE config 5's model, beam 5 x 20 steps, out_size 5 (every beam,
    best first).  Stored per variant: tokens, log-probs and the per-step top-6 candidate scores (the margin record)."""
    import_reference()
    model, cfg = _ref_model(3)
    model.eval()
    B, Ng, beam, T = 8, 100, 5, 20
    gen = torch.Generator().manual_seed(15)
    gri = torch.randn(B, Ng, 1024, generator=gen)
    reg = torch.randn(B, 150, 512, generator=gen)
    gri_mask = torch.zeros(B, 1, 1, Ng, dtype=torch.bool)
    for i, n in enumerate([100, 100, 80, 70, 100, 49, 90, 64]):
        gri_mask[i, ..., n:] = True
        gri[i, n:] = 0
    reg_mask = torch.zeros(B, 1, 1, 150, dtype=torch.bool)
    out = {'gri_feat': gri.numpy(), 'reg_feat': reg.numpy(), 'gri_mask': gri_mask.numpy(), 'reg_mask': reg_mask.numpy(),
           'eos_row_scales': np.array([6.0, 10.0], dtype=np.float32)}
    w0 = model.cap_generator.fc.weight.data.clone()
    model.cached_features = True
    for scale in (6, 10):
        model.cap_generator.fc.weight.data.copy_(w0)
        model.cap_generator.fc.weight.data[3] *= float(scale)
        record = []
        orig = model.select

        def select(t, cand, beam_size, _orig=orig, _rec=record, **kw):
            flat = cand.reshape(cand.shape[0], -1)
            _rec.append(torch.sort(flat, -1, descending=True)[0][:, :beam_size + 1].clone())
            return _orig(t, cand, beam_size, **kw)

        model.select = select
        with torch.no_grad():
            toks, lps = model({'gri_feat': gri.clone(), 'gri_mask': gri_mask.clone(), 'reg_feat': reg.clone(),
                               'reg_mask': reg_mask.clone()}, seq=None, use_beam_search=True, max_len=T, eos_idx=3,
                              beam_size=beam, out_size=beam)
        model.select = orig
        top6 = torch.stack(record, 1)  # [B, steps, beam + 1]
        out.update({f's{scale}_tokens': toks.numpy(), f's{scale}_logprobs': lps.numpy(), f's{scale}_top': top6.numpy()})
        first = [[int((toks[b, k] == 3).nonzero()[0]) if (toks[b, k] == 3).any() else -1 for k in range(beam)] for b in range(B)]
        d = (top6[..., :-1] - top6[..., 1:]).abs()
        print(f'g15 x{scale}: first EOS per (image, beam)', first, 'min margin per image', d.amin((1, 2)).tolist())
    np.savez_compressed(os.path.join(HERE, 'beam_g15.npz'), **out)


MAKERS.update({'g15': make_g15})

if __name__ == "__main__":
    main()
