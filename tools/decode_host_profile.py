"""Host-side cost of the beam-search loop (cProfile, batch 64, bf16): where the ~2.7 ms per step go."""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from grit_amd.config import default_config
    from grit_amd.models.caption import Transformer
    from grit_amd.models.caption.detector import build_detector
    cfg = default_config()
    torch.manual_seed(0)
    model = Transformer(build_detector(cfg), cfg).cuda().eval().to(torch.bfloat16)
    B = 64
    vis = {"gri_feat": torch.randn(B, 100, 1024, device="cuda").bfloat16(), "reg_feat": torch.randn(B, 150, 512, device="cuda").bfloat16(),
           "gri_mask": torch.zeros(B, 1, 1, 100, dtype=torch.bool, device="cuda"),
           "reg_mask": torch.zeros(B, 1, 1, 150, dtype=torch.bool, device="cuda")}
    model.cached_features = True
    with torch.inference_mode():
        for _ in range(2):
            model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(3):
            model(vis, seq=None, use_beam_search=True, max_len=20, eos_idx=3, beam_size=5, out_size=1)
        pr.disable()
        torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(30)


if __name__ == "__main__":
    main()
