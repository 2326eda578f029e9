"""Debug: where does grit_msda_bwd_bf16_staged differ from the dense f32 kernel + one rounding?"""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from grit_amd import lib as _lib
from grit_amd.ops import msda as msda_op
sys.path.insert(0, "tests")
from tests.test_msda_gpu import _config2
DEV = "cuda"
for B in (2, 32, 32, 8, 32):
    value, shapes, lsi, loc, aw = _config2(B=B)
    v16 = value.bfloat16().to(DEV)
    shapes, lsi, loc, aw = shapes.to(DEV), lsi.to(DEV), loc.to(DEV), aw.to(DEV)
    cot = torch.randn(B, 150, 512, generator=torch.Generator().manual_seed(5)).bfloat16().to(DEV)
    for rep in range(3):
        gv, gl, ga = msda_op.ms_deform_attn_backward(v16, shapes, lsi, loc, aw, cot)
        Bv, S, M, D = v16.shape
        dense = torch.zeros(v16.shape, dtype=torch.float32, device=DEV)
        gl2, ga2 = torch.empty_like(loc), torch.empty_like(aw)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        st = _lib.load().grit_msda_bwd_bf16(p(v16), p(shapes), p(lsi), p(loc), p(aw), p(cot), Bv, S, M, D, 4, 150, 4, p(dense), p(gl2), p(ga2), _lib.current_stream_ptr())
        torch.cuda.synchronize()
        want = dense.to(torch.bfloat16)
        bad = (dense == 0) & (gv != 0)
        miss = (dense != 0) & (gv == 0) & (want != 0)
        ent = msda_op._STAGE[(str(v16.device), Bv, S, M)]
        cells_bad = bad.any(-1)
        print("B", B, "rep", rep, "spurious elems", int(bad.sum()), "in cells", int(cells_bad.sum()), "missing elems", int(miss.sum()),
              "stage nz", int(ent[0].count_nonzero()), "flags nz", int(ent[1].count_nonzero()),
              "max spurious", float(gv[bad].float().abs().max()) if bad.any() else 0.0)
        if bad.any():
            idx = cells_bad.nonzero()[:5]
            for b, s, m in idx.tolist():
                print("   cell", b, s, m, "gv", gv[b, s, m, :4].float().tolist(), "dense", dense[b, s, m, :4].tolist(), "chan bad", int(bad[b, s, m].sum()))
