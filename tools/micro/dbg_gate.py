"""Isolate which stage of grit_gate_fuse deviates from torch's bf16 chain."""
import numpy as np
import torch
from grit_amd.ops import gate
torch.manual_seed(0)
R, d = 320, 512
dt = torch.bfloat16
e1 = (torch.randn(R, 1, d, device='cuda') * 2).to(dt)
e2 = (torch.randn(R, 1, d, device='cuda') * 2).to(dt)
m = torch.ones(R, 1, 1, device='cuda').to(dt)
G = (torch.randn(2 * R, d, device='cuda') * 2).to(dt)
z = torch.zeros_like(e1)
big = torch.full_like(G, 60.0)
zero = torch.zeros_like(G)


def ref(a, b, g):
    gv = g.view(2, R, 1, d)
    return ((a * torch.sigmoid(gv[0]) + b * torch.sigmoid(gv[1])) / np.sqrt(2)) * m


def cmp(name, a, b, g):
    got, want = gate.fuse(a, b, g, m), ref(a, b, g)
    bad = (got != want)
    print("%-40s mismatches %7d / %d   max ulp-ish %s" % (name, bad.sum().item(), got.numel(),
          (got.view(torch.int16).int() - want.view(torch.int16).int()).abs().max().item()))


with torch.no_grad():
    cmp("gates=+60 (sigmoid 1): (e1+e2)*inv", e1, e2, big)
    cmp("gates=0 (sigmoid .5): (e1/2+e2/2)*inv", e1, e2, zero)
    cmp("e2=0, gates=+60: e1*inv", e1, z, big)
    cmp("e2=0: (e1*s1)*inv", e1, z, G)
    cmp("full", e1, e2, G)
    s = torch.sigmoid(G)
    print("sigmoid(G) distinct from 1/(1+exp(-f32)):", (s != (1 / (1 + torch.exp(-G.float()))).to(dt)).sum().item())
