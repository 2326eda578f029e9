"""hipGraph capture of the launch-bound part of the training step.

Profile (profiles/r01/bench_bs32_v8_steady_state.txt + gap analysis): the Swin backbone keeps the GPU > 97 % busy,
but input_proj + the 6 deformable decoder layers + grid net + caption decoder are ~900 small kernels forward and
backward (150 queries, 20 caption tokens) during which the GPU idles more than half of the time waiting for the Python
dispatcher: ~12 ms of a ~100 ms step.  `CaptionHead` packages exactly that region as one callable of tensors and
`GraphedHead` captures its forward and backward with torch.cuda.make_graphed_callables (hipGraph under ROCm): the
kernels -- including this repo's HIP kernels, which are launched on the capturing stream like any other -- replay
with no host work in between.

Conditions (otherwise the head runs eagerly, same arithmetic): training mode, device tensors, bf16 compute weights,
no image padding (masks all False, known on the host), static shapes per cache entry.  Dropout stays random under
replay: torch's dropout uses the graph-safe device generator, and the attention kernels read their seed from device
memory refreshed by a captured RNG kernel (include/grit_hip.h, seed_dev).
"""
import warnings

import torch
import torch.nn.functional as F
from torch import nn


class CaptionHead(nn.Module):
    """features (4 NCHW maps) + caption tokens -> log-probs; everything after the backbone, no padding."""

    def __init__(self, transformer):
        super().__init__()
        det = transformer.detector
        self.input_proj = det.input_proj
        self.det_module = det.det_module
        self.grid_net = transformer.grid_net
        self.cap_generator = transformer.cap_generator

    def forward(self, f0, f1, f2, f3, captions):
        feats = (f0, f1, f2, f3)
        B = f0.shape[0]
        masks = [torch.zeros((B,) + tuple(f.shape[-2:]), dtype=torch.bool, device=f.device) for f in feats]
        srcs = []
        for (conv, gn), f in zip(self.input_proj, feats):  # Detector.project_level
            _, C, H, W = f.shape
            y = F.linear(f.permute(0, 2, 3, 1).reshape(B, H * W, C), conv.weight.view(conv.out_channels, C), conv.bias)
            srcs.append(gn(y.transpose(1, 2).reshape(B, conv.out_channels, H, W)))
        hs, _, _ = self.det_module(srcs, masks, no_padding=True)
        reg_feat = hs[-1]
        gri_feat = f3.flatten(2).transpose(1, 2)
        gri_mask = masks[-1].flatten(1)[:, None, None, :]
        grid, _ = self.grid_net(gri_feat, gri_mask)
        vis = {'gri_feat': grid[:, -1], 'gri_mask': gri_mask, 'reg_feat': reg_feat,
               'reg_mask': torch.zeros((B, 1, 1, reg_feat.shape[1]), dtype=torch.bool, device=f0.device)}
        return self.cap_generator(captions, vis)


class GraphedHead(object):
    """Lazily captured CaptionHead, one graph pair per input signature (shapes / dtypes); eager on any failure."""

    MAX_GRAPHS = 4

    def __init__(self, transformer):
        self.head = CaptionHead(transformer)
        self.graphs = {}
        self.disabled = False

    @staticmethod
    def _signature(tensors):
        return tuple((tuple(t.shape), t.dtype, t.requires_grad) for t in tensors)

    def __call__(self, features, captions):
        args = tuple(f.contiguous() if not f.is_contiguous() and False else f for f in features) + (captions,)
        if self.disabled or not all(a.is_cuda for a in args):
            return self.head(*args)
        key = self._signature(args)
        fn = self.graphs.get(key)
        if fn is None:
            if len(self.graphs) >= self.MAX_GRAPHS:
                return self.head(*args)
            try:
                samples = tuple(a.detach().clone().requires_grad_(a.requires_grad) for a in args)
                fn = torch.cuda.make_graphed_callables(self.head, samples, allow_unused_input=True)
                self.graphs[key] = fn
            except Exception as e:  # capture is an optimisation: never fail the step because of it
                warnings.warn(f"hipGraph capture of the caption head failed, running it eagerly: {e}")
                self.disabled = True
                return self.head(*args)
        return fn(*args)
